"""Data formats either side of the path: .bin clouds, info pickles, the dt_info pickle.
Host logic on CPU; the sequence runner against the drop-in loop on the GPU."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import golden, load_pkg


def _info(path, names, npts):
    n = len(names)
    return {"velodyne_path": path, "image_idx": 0, "pointcloud_num_features": 4,
            "annos": {"name": np.array(names, dtype="<U9"), "num_points": np.array(npts, dtype=np.int32),
                      "location": np.arange(3 * n, dtype=np.float32).reshape(n, 3),
                      "dimensions": np.ones((n, 3), np.float32), "rotation_y": np.zeros(n, np.float32)}}


def test_velodyne_roundtrip_and_truncated_file(tmp_path, synth):
    io = load_pkg("kitti_io")
    pts = synth.lidar_cloud("eight_20cm", seed=3, n_points=777)
    f = tmp_path / "000003.bin"
    pts.tofile(f)
    got = io.read_velodyne(f)
    assert got.dtype == np.float32 and got.shape == (777, 4) and np.array_equal(got, pts)
    (tmp_path / "bad.bin").write_bytes(pts.tobytes()[:-4])
    with pytest.raises(ValueError):
        io.read_velodyne(tmp_path / "bad.bin")
    (tmp_path / "empty.bin").write_bytes(b"")
    assert io.read_velodyne(tmp_path / "empty.bin").shape == (0, 4)


def test_infos_and_class_remap(tmp_path):
    """train.py:164-184: boxes without points are dropped, then car/truck/bus -> vehicle, person -> pedestrian,
    bicycle/motorbike -> cyclist; other labels pass through."""
    io = load_pkg("kitti_io")
    a = [_info("seq/velodyne/0.bin", ["car", "person", "truck", "bicycle", "cone"], [5, 0, 3, 9, 2]),
         _info("seq/velodyne/1.bin", [], [])]
    b = [_info("seq/velodyne/2.bin", ["motorbike", "bus", "person"], [1, 1, 7])]
    os.makedirs(tmp_path / "x")
    with open(tmp_path / "x" / "a.pkl", "wb") as f:
        pickle.dump(a, f)
    with open(tmp_path / "b.pkl", "wb") as f:
        pickle.dump(b, f)
    infos = io.load_infos(tmp_path, ["x/a.pkl", "b.pkl"])
    assert [i["velodyne_path"] for i in infos] == ["seq/velodyne/0.bin", "seq/velodyne/1.bin", "seq/velodyne/2.bin"]
    persons = io.remap_classes(infos)
    assert persons == 1  # the person without points was dropped before the count, as in the reference
    assert list(infos[0]["annos"]["name"]) == ["vehicle", "vehicle", "cyclist", "cone"]
    assert infos[0]["annos"]["location"].shape == (4, 3) and np.array_equal(infos[0]["annos"]["num_points"], [5, 3, 9, 2])
    assert len(infos[1]["annos"]["name"]) == 0
    assert list(infos[2]["annos"]["name"]) == ["cyclist", "vehicle", "pedestrian"]


def test_records_to_annos_and_detection_pickle(tmp_path):
    io = load_pkg("kitti_io")
    det = np.zeros((2, 6, 9), np.float32)
    det[0, :3] = [[1, 2, 3, 4, 5, 6, 0.5, 0.9, 0], [7, 8, 9, 1, 1, 2, -0.5, 0.8, 2], [0, 0, 0, 1, 1, 1, 0, 0.7, 1]]
    cnt = np.zeros((2, 9), np.int32)
    cnt[0, 0] = 3
    annos = io.annos_from_records(det, cnt, ["vehicle", "pedestrian", "cyclist"])
    assert list(annos[0]["name"]) == ["vehicle", "cyclist", "pedestrian"]
    assert np.allclose(annos[0]["location"][1], [7, 8, 9]) and np.allclose(annos[0]["dimensions"][0], [4, 5, 6])
    assert np.allclose(annos[0]["rotation_y"], [0.5, -0.5, 0]) and np.allclose(annos[0]["score"], [0.9, 0.8, 0.7])
    # an empty frame keeps the reference's empty result layout (inference.py:724-737)
    assert annos[1]["location"].shape == (0, 3) and annos[1]["bbox"].shape == (0, 4) and annos[1]["name"].shape == (0,)
    p = tmp_path / "results" / "exp" / "dt_info.pkl"
    io.save_detections(p, annos)
    back = io.load_detections(p)
    assert len(back) == 2 and np.array_equal(back[0]["score"], annos[0]["score"]) and list(back[0]["name"]) == list(annos[0]["name"])


def test_remap_and_anno_layout_vs_reference_fixture(tmp_path):
    """tests/golden/kitti_io_ref.npz = the reference itself run by make_goldens.py --only io: train.py:164-184 `changeInfo` on a
    seeded 40-frame info list (200 boxes, 142 left), and the anno dicts `Inference.infer_gpu` returns (inference.py:124-138,
    724-737) for a frame with 764 detections and an empty one -- what train.py:258-265 pickles as dt_info."""
    io = load_pkg("kitti_io")
    g = golden("kitti_io_ref")
    infos, o = [], 0
    for f, n in enumerate(g["in_n"]):
        sl = slice(o, o + int(n))
        infos.append({"velodyne_path": f"seq/velodyne/{f:06d}.bin",
                      "annos": {"name": g["in_name"][sl].copy(), "num_points": g["in_num_points"][sl].copy(), "location": g["in_location"][sl].copy(),
                                "bbox": g["in_bbox"][sl].copy()}})
        o += int(n)
    persons = io.remap_classes(infos)
    assert persons == int((g["in_name"][g["in_num_points"] > 0] == "person").sum())
    o = 0
    for f, n in enumerate(g["out_n"]):
        a, sl = infos[f]["annos"], slice(o, o + int(n))
        assert len(a["name"]) == n, f
        if n or len(g["in_name"]):  # the reference leaves frames without boxes untouched
            assert list(a["name"]) == list(g["out_name"][sl]), f
        assert np.array_equal(a["num_points"], g["out_num_points"][sl]) and np.array_equal(a["location"], g["out_location"][sl])
        assert np.array_equal(a["bbox"], g["out_bbox"][sl])
        if g["in_n"][f]:
            assert a["name"].dtype.str == g["out_name_dtype"][f]
        o += int(n)
    # the anno layout: key order, dtypes, shapes and values of a frame with detections and of an empty frame
    k = g["det_score"].shape[0]
    det = np.zeros((2, k + 5, 9), np.float32)
    det[0, :k] = np.concatenate([g["det_location"], g["det_dimensions"], g["det_rotation_y"][:, None], g["det_score"][:, None],
                                 g["det_cls"][:, None].astype(np.float32)], axis=1)
    cnt = np.zeros((2, 4), np.int32)
    cnt[0, 0] = k
    annos = io.annos_from_records(det, cnt, list(g["class_names"]))
    layout = lambda a: [f"{key}|{np.asarray(v).dtype.str}|{','.join(map(str, np.asarray(v).shape))}" for key, v in a.items()]
    assert layout(annos[0]) == list(g["anno_layout"])
    assert layout(annos[1]) == list(g["empty_layout"])
    assert list(annos[0]["name"]) == list(g["det_name"])
    for key in ("location", "dimensions", "rotation_y", "score"):
        assert np.array_equal(annos[0][key], g["det_" + key])
    io.save_detections(tmp_path / "r" / "dt_info.pkl", annos)
    back = io.load_detections(tmp_path / "r" / "dt_info.pkl")
    assert layout(back[0]) == list(g["anno_layout"]) and layout(back[1]) == list(g["empty_layout"])


@pytest.mark.gpu
def test_run_sequence_vs_oracle(tmp_path, synth):
    """.bin files -> run_sequence (pp_infer_batch, 2 frames per pass, ragged tail) against the CPU ORACLE frame by frame
    (tests/frame_check.compare_frame: mask bit-exact, every logit bounded, the selection exact on the GPU's own logits, every oracle row
    matched by anchor id within 1e-3 or explained) -- the body of train.py:218-242 on the product path."""
    from frame_check import compare_frame, gpu_logits, oracle_frame
    from oracle import pp_oracle as O
    pkg = load_pkg()
    pkg.install()
    import framework.voxel_generator as vg
    import networks.pointpillars8_shared as shared
    io = load_pkg("kitti_io")
    eng_mod = load_pkg("engine")
    cfg = synth.load_config("nuscene")
    cfg["device"] = torch.device("cuda:0")
    cfg["max_batch"] = 2
    vg.VoxelGenerator(cfg)
    net = shared.PointPillars(cfg)
    net.to(cfg["device"])
    sd = synth.seeded_state_dict(4, cls_bias=-3.0)
    net.load_state_dict(sd)
    net.eval()
    paths, clouds = [], []
    for i in range(3):
        p = tmp_path / f"{i:06d}.bin"
        clouds.append(synth.lidar_cloud("nuscene", seed=40 + i, n_points=9000 + 1000 * i))
        clouds[-1].tofile(p)
        paths.append(p)
    names = list(O.make_anchors(O.voxel_setup(synth.load_config("nuscene")))["class_masks"].keys())
    eng = eng_mod.engine_for(cfg)

    def rows_of(anno):
        k = len(anno["score"])
        det = np.concatenate([anno["location"].reshape(k, 3), anno["dimensions"].reshape(k, 3), np.asarray(anno["rotation_y"]).reshape(k, 1),
                              np.asarray(anno["score"]).reshape(k, 1), np.array([names.index(n) for n in anno["name"]], np.float32).reshape(k, 1)],
                             axis=1).astype(np.float32)
        cnt = np.array([k] + [int((det[:, 8] == c).sum()) for c in range(len(names))], np.int32)
        return det, cnt

    got = io.run_sequence(cfg, paths, names, batch=2)         # passes: frames (0, 1), then the ragged tail (2)
    assert len(got) == 3 and all(len(g["score"]) > 0 for g in got)
    det, cnt = rows_of(got[2])                                 # the engine still holds the last pass: frame 2 at index 0
    compare_frame(oracle_frame(synth, "nuscene", clouds[2], sd), gpu_logits(eng, 0), det, cnt, "aabb", "run_sequence nuscene frame 2 (tail pass)")
    again = io.run_sequence(cfg, paths[:2], names, batch=2)   # now the engine holds frames 0 and 1
    for f in range(2):
        det, cnt = rows_of(again[f])
        compare_frame(oracle_frame(synth, "nuscene", clouds[f], sd), gpu_logits(eng, f), det, cnt, "aabb", f"run_sequence nuscene frame {f}")
        assert list(again[f]["name"]) == list(got[f]["name"])
        for key in ("location", "dimensions", "rotation_y", "score"):
            np.testing.assert_allclose(again[f][key], got[f][key], rtol=0, atol=1e-4)
