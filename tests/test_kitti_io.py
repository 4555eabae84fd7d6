"""Data formats either side of the path: .bin clouds, info pickles, the dt_info pickle.
Host logic on CPU; the sequence runner against the drop-in loop on the GPU."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import load_pkg


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


@pytest.mark.gpu
def test_run_sequence_matches_dropin_loop(tmp_path, synth):
    """.bin files -> run_sequence (pp_infer_batch, 2 frames per pass, ragged tail) must give the annos the
    reference-style per-frame loop gives on the drop-in classes."""
    pkg = load_pkg()
    pkg.install()
    import framework.voxel_generator as vg
    import framework.anchor_assigner as aa
    import framework.dataset as ds
    import framework.inference as inf
    import networks.pointpillars8_shared as shared
    io = load_pkg("kitti_io")
    cfg = synth.load_config("nuscene")
    cfg["device"] = torch.device("cuda:0")
    cfg["max_batch"] = 2
    voxel_generator = vg.VoxelGenerator(cfg)
    anchor_assigner = aa.AnchorAssigner(cfg)
    inference = inf.Inference(cfg, anchor_assigner)
    infer_data = ds.InferData(cfg, voxel_generator, anchor_assigner, torch.float32)
    net = shared.PointPillars(cfg)
    net.to(cfg["device"])
    net.load_state_dict(synth.seeded_state_dict(4, cls_bias=-3.0))
    net.eval()
    paths = []
    for i in range(3):
        p = tmp_path / f"{i:06d}.bin"
        synth.lidar_cloud("nuscene", seed=40 + i, n_points=9000 + 1000 * i).tofile(p)
        paths.append(p)
    names = list(anchor_assigner.class_masks.keys())
    got = io.run_sequence(cfg, paths, names, batch=2)
    assert len(got) == 3
    for p, g in zip(paths, got):
        example = infer_data.get(io.read_velodyne(p))
        with torch.no_grad():
            ref = inference.infer_gpu(example, net(example))[0]
        assert len(ref["score"]) > 0
        assert list(g["name"]) == list(ref["name"])
        for key in ("location", "dimensions", "rotation_y", "score"):
            np.testing.assert_allclose(g[key], ref[key], rtol=0, atol=1e-3)  # dense-canvas loop vs fused sparse path
