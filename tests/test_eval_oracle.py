"""The evaluation oracle (oracle/eval_oracle.py + the C rotated overlap) against the goldens the reference's own
eval/eval.py produced (tests/golden/eval_ap.npz)."""
import numpy as np

from conftest import golden
from oracle import c_oracle as C
from oracle import eval_oracle as E

CLASSES = ["vehicle", "pedestrian", "cyclist"]


def unpack(g, prefix, keys):
    cnt = g[prefix + "count"]
    off = np.concatenate([[0], np.cumsum(cnt)])
    return [{k: g[prefix + k][off[i]:off[i + 1]] for k in keys} for i in range(len(cnt))]


def load_sets():
    g = golden("eval_ap")
    gts = unpack(g, "gt_", ["name", "location", "dimensions", "rotation_y", "num_points"])
    dts = unpack(g, "dt_", ["name", "location", "dimensions", "rotation_y", "score"])
    return g, gts, dts


def test_rotated_overlap_criteria():
    g = golden("eval_ap")
    rb = g["rb"]
    for c, key in ((-1, "crit_m1"), (0, "crit_0"), (1, "crit_1"), (2, "crit_2")):
        got = C.rotated_iou_eval(rb[:10], rb[8:], c)
        np.testing.assert_allclose(got, g[key], rtol=0, atol=2e-6)
    assert C.rotated_iou_eval(rb[:0], rb, -1).shape == (0, 24)


def test_frame_overlaps_bev_and_3d():
    g, gts, dts = load_sets()
    for f in (0, 5):
        np.testing.assert_allclose(E.frame_overlaps(dts[f], gts[f], "bev"), g[f"ov_bev_{f}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(E.frame_overlaps(dts[f], gts[f], "3d"), g[f"ov_3d_{f}"], rtol=0, atol=2e-6)


def test_official_result_matches_reference():
    g, gts, dts = load_sets()
    for rt in (80.0, 40.0):
        res, text = E.official_result(gts, dts, CLASSES, rt)
        np.testing.assert_allclose(res[0], g[f"map_bev_{int(rt)}"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(res[1], g[f"map_3d_{int(rt)}"], rtol=0, atol=1e-9)
        assert text == str(g[f"eval_str_{int(rt)}"])
    ret = E.eval_class_ap(gts, dts, CLASSES, "3d", E.MIN_OVERLAPS, 5, 80.0)
    np.testing.assert_allclose(ret["precision"], g["precision_3d_80"], rtol=0, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(ret["recall"], g["recall_3d_80"], rtol=0, atol=1e-12, equal_nan=True)
