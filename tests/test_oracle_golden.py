"""CPU: the oracle (oracle/pp_oracle.py + pp_oracle.c) against the golden vectors that
tests/golden/make_goldens.py captured from the reference itself.
Integer / index stages must be bit-exact, float stages <= 1e-5."""
import hashlib

import numpy as np
import pytest

from conftest import golden
from oracle import c_oracle as C
from oracle import pp_oracle as O

CONFIGS = ("eight_20cm", "ntusl_10cm", "nuscene")


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", CONFIGS)
def test_voxel_setup(name, synth):
    g = golden(f"setup_{name}")
    s = O.voxel_setup(synth.load_config(name))
    for k in ("voxel_size", "offset", "grid_size", "detection_range", "range_diff"):
        assert np.array_equal(s[k], g[k]), k
        assert s[k].dtype == g[k].dtype


@pytest.mark.parametrize("impl", [O, C], ids=["numpy", "c"])
@pytest.mark.parametrize("name", CONFIGS)
def test_voxelize_small_break(name, impl, synth):
    g = golden(f"voxel_small_{name}")
    s = O.voxel_setup(synth.load_config(name))
    v, c, n = impl.points_to_voxels(g["points"], s["voxel_size"], s["offset"], s["grid_size"],
                                    int(g["max_voxels"]), int(g["max_num_points"]))
    assert v.shape[0] == 900  # the (max_voxels+1)-th pillar break fired
    assert np.array_equal(c, g["coors"]) and np.array_equal(n, g["num"]) and np.array_equal(v, g["voxels"])


@pytest.mark.parametrize("impl", [O, C], ids=["numpy", "c"])
@pytest.mark.parametrize("name", CONFIGS)
def test_voxelize_full(name, impl, synth):
    g = golden(f"voxel_full_{name}")
    cfg = synth.load_config(name)
    s = O.voxel_setup(cfg)
    pts = synth.lidar_cloud(name, seed=1000)
    assert sha(pts) == str(g["points_sha"]), "synthetic cloud generator drifted"
    v, c, n = impl.points_to_voxels(pts, s["voxel_size"], s["offset"], s["grid_size"], cfg["max_voxels"],
                                    cfg["max_num_points"])
    assert np.array_equal(c, g["coors"]) and np.array_equal(n, g["num"])
    assert sha(v) == str(g["voxels_sha"])


@pytest.mark.parametrize("impl", [O, C], ids=["numpy", "c"])
def test_voxelize_edges(impl, synth):
    g = golden("voxel_edge")
    cfg = synth.load_config("eight_20cm")
    s = O.voxel_setup(cfg)
    v, c, n = impl.points_to_voxels(g["points"], s["voxel_size"], s["offset"], s["grid_size"], 16000, 15)
    assert np.array_equal(c, g["coors"]) and np.array_equal(n, g["num"]) and np.array_equal(v, g["voxels"])
    v, c, n = impl.points_to_voxels(np.zeros((0, 4), np.float32), s["voxel_size"], s["offset"], s["grid_size"], 16000, 15)
    assert v.shape[0] == int(g["empty_p"]) == 0


@pytest.fixture(scope="module")
def anchors_e20(synth):
    s = O.voxel_setup(synth.load_config("eight_20cm"))
    return s, O.make_anchors(s)


def test_anchor_table(anchors_e20):
    s, a = anchors_e20
    g = golden("anchors_eight_20cm")
    assert a["anchors"].shape == (1440000, 7)
    assert sha(a["anchors"]) == str(g["anchors_sha"])
    assert sha(a["anchors_coors"]) == str(g["coors_sha"])
    rows = g["rows"]
    assert np.array_equal(a["anchors"][rows], g["anchors_rows"])
    assert np.array_equal(a["anchors_bv"][rows], g["bv_rows"])
    assert np.array_equal(a["anchors_coors"][rows], g["coors_rows"])
    assert list(a["class_masks"].keys()) == [str(x) for x in g["class_names"]]
    assert np.array_equal(np.array(list(a["class_masks"].values())), g["class_ranges"])


@pytest.mark.parametrize("impl", [O, C], ids=["numpy", "c"])
def test_anchor_mask(impl, anchors_e20):
    s, a = anchors_e20
    g = golden("anchors_eight_20cm")
    coors = golden("voxel_full_eight_20cm")["coors"]
    m = impl.create_mask(coors, s["grid_size"], a["anchors_coors"])
    assert int(m.sum()) == int(g["mask_count"])
    assert np.array_equal(np.packbits(m), g["mask_bits"])


def test_pfn(synth):
    g = golden("pfn_eight_20cm")
    s = O.voxel_setup(synth.load_config("eight_20cm"))
    out = O.pfn(g["voxels"], g["num"], g["coors"], synth.seeded_state_dict(0), s)
    assert set(np.unique(g["num"])) >= {1, 14, 15}
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=1e-5)


def test_scatter():
    g = golden("scatter_small")
    assert np.array_equal(O.scatter(g["feat"], g["coors"], (24, 40, 1)), g["canvas"])


@pytest.mark.parametrize("norm", ["instance", "batch"])
def test_backbone_small(norm, synth):
    g = golden(f"backbone_small_{norm}")
    y = O.backbone(g["x"], synth.seeded_state_dict(0, norm=norm), norm=norm)
    assert y.shape == g["y"].shape == (1, 320, 32, 24)
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1e-5)


def test_head_layout(synth):
    g = golden("head_small")
    cls, box, dr = O.head(g["x"], synth.seeded_state_dict(0))
    np.testing.assert_allclose(cls, g["cls"], atol=1e-5)
    np.testing.assert_allclose(box, g["box"], atol=1e-5)
    np.testing.assert_allclose(dr, g["dir"], atol=1e-5)


def test_box_math():
    g = golden("boxmath")
    dec = O.box_decode(g["enc"], g["anchors"])
    assert np.array_equal(dec, g["dec_np"])
    np.testing.assert_allclose(dec, g["dec_t"], rtol=1e-6, atol=1e-6)
    cor = O.center_to_corner_box2d(dec[:, :2], dec[:, 3:5], dec[:, 6])
    np.testing.assert_allclose(cor, g["cor_np"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(cor, g["cor_t"], rtol=0, atol=1e-5)
    assert np.array_equal(O.corner_to_standup_nd(g["cor_np"]), g["st_np"])
    assert np.array_equal(O.corner_to_standup_nd(g["cor_np"]), g["st_t"])
    assert np.array_equal(O.limit_period(g["lp_in"], 0.5, 2 * np.pi), g["lp_out"])


@pytest.mark.parametrize("impl", [O, C], ids=["numpy", "c"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 300, 1000])
def test_nms_aabb(n, impl):
    g = golden("nms_aabb")
    assert impl.nms_aabb(g[f"dets_{n}"], 0.1) == [int(v) for v in g[f"keep_{n}"]]
    assert impl.nms_aabb(np.zeros((0, 5), np.float32), 0.1) == []


def test_rotated_iou_matrix():
    g = golden("nms_rotated")
    b = g["boxes"]
    m = np.array([[O.rotated_iou(b[i], b[j]) for j in range(b.shape[0])] for i in range(b.shape[0])], dtype=np.float32)
    mc = np.array([[C.rotated_iou(b[i], b[j]) for j in range(b.shape[0])] for i in range(b.shape[0])], dtype=np.float32)
    # the reference's devRotateIoU was run as plain Python (libm double sin/cos rounded to f32);
    # C uses sinf/cosf: equal to a few ulp except on degenerate (coincident-edge) pairs
    ok = np.isfinite(g["iou"])
    np.testing.assert_allclose(m[ok], g["iou"][ok], rtol=0, atol=1e-5)
    assert np.mean(np.abs(mc[ok] - g["iou"][ok]) < 1e-4) > 0.995


@pytest.mark.parametrize("impl", [O, C], ids=["numpy", "c"])
def test_nms_rotated(impl):
    g = golden("nms_rotated")
    assert impl.nms_rotated(g["dets"], 0.1) == [int(v) for v in g["keep"]]
    assert impl.nms_rotated(g["dets200"], 0.1) == [int(v) for v in g["keep200"]]
