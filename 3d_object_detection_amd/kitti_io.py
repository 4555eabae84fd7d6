"""Data formats either side of the hot path: the reference's evaluation loop reads KITTI-style
`velodyne/*.bin` clouds listed in `data_info.pkl` info files and writes the detections as one pickle
(`train.py:204-222`, `:258-265`; infos are produced by `create_info.py:33-113`).

  read_velodyne(path)            f32[N,4] cloud                                   (train.py:221-222)
  load_infos(root, info_paths)   concatenated info dicts                          (train.py:208-214)
  remap_classes(infos)           the reference's label clean-up, in place         (train.py:164-184)
  annos_from_records(...)        device detection records -> KITTI-style annos    (inference.py:124-138,724-737)
  run_sequence(...)              clouds -> annos through pp_infer_batch, B frames per pass, one D2H per pass
  save_detections / load_detections   the `dt_info` pickle                        (train.py:264-265)
  pointcloud2_to_points(msg)     sensor_msgs/PointCloud2 payload -> f32[N,4] on the device   (ros_node.py:55-59)

Host-side glue only: the compute stays in libpp_hip.so (engine.Engine).
"""
import os
import pickle

import numpy as np
import torch

from .engine import engine_for
from .framework.inference import get_start_result_anno

# train.py:171-184: dataset labels -> the three detection classes
CLASS_REMAP = {"car": "vehicle", "truck": "vehicle", "bus": "vehicle", "person": "pedestrian",
               "bicycle": "cyclist", "motorbike": "cyclist"}


def read_velodyne(path, num_features=4):
    """`np.fromfile(v_path, dtype=np.float32).reshape([-1, 4])` (train.py:222); a truncated file is an error here
    instead of a silent reshape failure."""
    raw = np.fromfile(os.fspath(path), dtype=np.float32)
    if raw.size % num_features:
        raise ValueError(f"{path}: {raw.size} floats is not a whole number of {num_features}-feature points")
    return raw.reshape(-1, num_features)


def load_infos(data_root, info_paths):
    """Concatenate the pickled info lists (train.py:208-214). Each info carries at least `velodyne_path` and,
    for evaluation, `annos` (name, location, dimensions, rotation_y, num_points, ...)."""
    infos = []
    for rel in info_paths:
        with open(os.path.join(os.fspath(data_root), rel), "rb") as f:
            infos += pickle.load(f)
    return infos


def remap_classes(infos):
    """In-place label clean-up of train.py:164-184: drop boxes without points, then map the dataset's labels onto
    vehicle / pedestrian / cyclist. Returns the number of 'person' boxes seen (the reference counts them)."""
    persons = 0
    for info in infos:
        annos = info.get("annos")
        if annos is None or len(annos["name"]) == 0:
            continue
        keep = np.asarray(annos["num_points"]) > 0
        for key in annos:
            annos[key] = annos[key][keep]
        names = annos["name"]
        persons += int((names == "person").sum())
        out = names.astype(object)
        for src, dst in CLASS_REMAP.items():
            out[names == src] = dst
        # a '<U7' source array would truncate 'pedestrian': widen like the reference's in-place writes cannot
        annos["name"] = out.astype("<U10")
    return persons


def annos_from_records(det, cnt, class_names):
    """det f32[nb,rows,9] (x,y,z,l,w,h,r,score,class), cnt i32[nb,1+C] on the HOST -> list of nb annos in the
    reference's result layout (inference.py:124-138 on top of get_start_result_anno :724-737)."""
    out = []
    for b in range(det.shape[0]):
        k = int(cnt[b, 0])
        anno = get_start_result_anno()
        if k > 0:
            rows = det[b, :k]
            anno["name"] = np.array([class_names[int(c)] for c in rows[:, 8]], dtype="<U10")
            anno["location"] = rows[:, :3].copy()
            anno["dimensions"] = rows[:, 3:6].copy()
            anno["rotation_y"] = rows[:, 6].copy()
            anno["score"] = rows[:, 7].copy()
        out.append(anno)
    return out


def run_sequence(config, clouds, class_names, batch=16, nms_mode=0):
    """The body of the reference's `infer()` loop (train.py:218-242) for a list of clouds (numpy f32[N,4] or paths
    of .bin files): frames go through pp_infer_batch `batch` at a time, one pinned D2H of the fixed-size records
    per pass. Returns the annos in frame order. The engine must hold committed weights (PointPillars.load_state_dict
    or Engine.load_state_dict)."""
    eng = engine_for(config)
    batch = max(1, min(int(batch), eng.max_batch))
    rows = eng.cfg.num_classes * eng.cfg.nms_post_max
    det_h = torch.zeros((batch, rows, 9), dtype=torch.float32).pin_memory()
    cnt_h = torch.zeros((batch, det_count_stride()), dtype=torch.int32).pin_memory()
    annos = []
    for i0 in range(0, len(clouds), batch):
        group = []
        for c in clouds[i0:i0 + batch]:
            pts = read_velodyne(c) if isinstance(c, (str, os.PathLike)) else np.ascontiguousarray(c, dtype=np.float32)
            group.append(torch.from_numpy(pts).to(eng.device, non_blocking=True))
        det, cnt = eng.infer_batch(group, nms_mode=nms_mode)
        nb = len(group)
        det_h[:nb].copy_(det, non_blocking=True)
        cnt_h[:nb].copy_(cnt, non_blocking=True)
        torch.cuda.current_stream(eng.device).synchronize()
        annos += annos_from_records(det_h[:nb].numpy(), cnt_h[:nb].numpy(), class_names)
    return annos


def det_count_stride():
    from . import _lib
    return 1 + _lib.PP_MAX_CLASSES


def save_detections(path, dt_annos):
    """`pickle.dump(dt_annos, f)` into <data_root>/<result_path>/<experiment>/<dt_info> (train.py:258-265)."""
    d = os.path.dirname(os.fspath(path))
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(dt_annos, f)


def load_detections(path):
    with open(path, "rb") as f:
        return pickle.load(f)


# sensor_msgs/PointField datatype codes
_PC2_SIZES = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 4, 8: 8}


def pointcloud2_to_points(msg, device=None):
    """`np.asarray(list(pc2.read_points(msg)))[:, :4].astype(np.float32)` of the reference's ROS callback
    (ros_node.py:55-59) as one H2D copy of the raw payload + one unpack kernel: the first four fields of the message
    (in the message's field order, whatever their types, offsets and padding) become f32[N,4] on the device.
    `msg` is any object with the PointCloud2 attributes (data, fields[name/offset/datatype/count], point_step, row_step,
    width, height, is_bigendian); rospy itself is not needed."""
    import ctypes
    from . import _lib
    fields = sorted(msg.fields, key=lambda f: f.offset)  # read_points orders the struct by offset
    if len(fields) < 4:
        raise ValueError("PointCloud2 message has fewer than 4 fields")
    first = fields[:4]
    for f in first:
        if getattr(f, "count", 1) != 1:
            raise ValueError(f"field {f.name}: count != 1 is not supported")
        if f.datatype not in _PC2_SIZES:
            raise ValueError(f"field {f.name}: unknown datatype {f.datatype}")
    n = int(msg.width) * int(msg.height)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    out = torch.empty((n, 4), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    raw = np.frombuffer(bytes(msg.data), dtype=np.uint8)
    need = (int(msg.height) - 1) * int(msg.row_step) + int(msg.width) * int(msg.point_step)
    if raw.size < need:
        raise ValueError(f"PointCloud2 data holds {raw.size} bytes, {need} needed")
    buf = torch.from_numpy(raw.copy()).to(dev, non_blocking=True)
    offs = (ctypes.c_int32 * 4)(*[int(f.offset) for f in first])
    dts = (ctypes.c_int32 * 4)(*[int(f.datatype) for f in first])
    with torch.cuda.device(dev):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.load().pp_unpack_points(buf.data_ptr(), n, int(msg.width), int(msg.row_step), int(msg.point_step), offs, dts,
                                                int(bool(msg.is_bigendian)), out.data_ptr(), s), None, "pp_unpack_points")
    return out
