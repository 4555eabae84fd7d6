"""KITTI-style AP evaluation of the package (eval/eval.py, eval/iou.py mirror).
CPU: the native host statistics and the Python bookkeeping against the oracle.  GPU: the device overlaps and the
whole `get_official_eval_result` against the goldens the reference's eval.py produced."""
import numpy as np
import pytest

from conftest import golden, load_pkg
from oracle import eval_oracle as E
from test_eval_oracle import CLASSES, load_sets


def test_native_statistics_equal_oracle():
    ev = load_pkg("eval.eval")
    rng = np.random.default_rng(0)
    for _ in range(300):
        nd, ng = int(rng.integers(0, 12)), int(rng.integers(0, 12))
        ov = rng.uniform(0, 1, (nd, ng))
        ig = rng.integers(-1, 2, ng)
        idt = rng.integers(-1, 1, nd)
        sc = rng.uniform(0, 1, nd).astype(np.float32)
        if nd > 2:
            sc[1] = sc[0]  # score ties: the first detection in order wins (strict >)
        for fp in (False, True):
            a = ev.compute_statistics_jit(ov, ig, idt, sc, 0.5, 0.3, fp)
            b = E.compute_statistics(ov, ig, idt, sc, 0.5, 0.3, fp)
            assert a[:3] == b[:3] and np.array_equal(a[3], b[3])


def test_fused_statistics_walks_the_diagonal_blocks():
    ev = load_pkg("eval.eval")
    rng = np.random.default_rng(1)
    gt_nums = np.array([3, 0, 5, 2])
    dt_nums = np.array([4, 2, 0, 6])
    ov = rng.uniform(0, 1, (dt_nums.sum(), gt_nums.sum()))
    ig = rng.integers(-1, 2, gt_nums.sum())
    idt = rng.integers(-1, 1, dt_nums.sum())
    sc = rng.uniform(0, 1, dt_nums.sum()).astype(np.float32)
    th = np.array([0.1, 0.4, 0.8])
    pr = np.zeros((3, 4))
    pr[0, 0] = 7.0  # accumulates
    ev.fused_compute_statistics(ov, pr, gt_nums, dt_nums, ig, idt, sc, 0.5, th)
    want = np.zeros((3, 4))
    want[0, 0] = 7.0
    g0 = d0 = 0
    for f in range(4):
        for t, x in enumerate(th):
            tp, fp, fn, _ = E.compute_statistics(ov[d0:d0 + dt_nums[f], g0:g0 + gt_nums[f]], ig[g0:g0 + gt_nums[f]], idt[d0:d0 + dt_nums[f]],
                                                 sc[d0:d0 + dt_nums[f]], 0.5, x, True)
            want[t, :3] += (tp, fp, fn)
        g0 += gt_nums[f]
        d0 += dt_nums[f]
    assert np.array_equal(pr, want)


def test_clean_data_thresholds_and_map_equal_oracle():
    ev = load_pkg("eval.eval")
    _, gts, dts = load_sets()
    for f in range(len(gts)):
        for cls in CLASSES + ["Vehicle"]:
            v, ig, idt = ev.clean_data(gts[f], dts[f], cls, 5, 60.0)
            v2, ig2, idt2 = E.clean_data(gts[f], dts[f], cls, 5, 60.0)
            assert v == v2 and list(ig) == list(ig2) and list(idt) == list(idt2)
    rng = np.random.default_rng(2)
    for n in (1, 2, 17, 200):
        s = rng.uniform(0, 1, n)
        assert ev.get_thresholds(s.copy(), 37) == E.get_thresholds(s.copy(), 37)
    p = rng.uniform(0, 1, (3, 2, 41))
    assert np.array_equal(ev.get_mAP(p), E.get_map(p))
    assert ev.get_split_parts(53, 50) == [1] * 50 + [3] and ev.get_split_parts(100, 50) == [2] * 50


def test_native_class_ap_equals_oracle():
    """pp_eval_class_ap (one call per (class, overlap threshold) cell over the flattened data set) against the oracle's
    restatement of the reference's loop nest, on oracle-computed overlaps (no device needed): parts of 1, of 7 and one
    part for everything, ragged / empty frames included."""
    ev = load_pkg("eval.eval")
    _, gts, dts = load_sets()
    gts, dts = gts[:23], dts[:23]
    want = E.eval_class_ap(gts, dts, CLASSES, "bev", ev.MIN_OVERLAPS, 5, 60.0)
    ovs = [np.ascontiguousarray(E.frame_overlaps(dts[i], gts[i], "bev")) for i in range(len(gts))]
    scene = ev._Scene(gts, dts)
    for per in (1, 7, 23):
        parts, frames = [], []
        for f0 in range(0, len(gts), per):
            fs = range(f0, min(f0 + per, len(gts)))
            nd, ng = sum(ovs[f].shape[0] for f in fs), sum(ovs[f].shape[1] for f in fs)
            m = np.full((nd, ng), 0.123)  # off-diagonal blocks hold overlaps of boxes of DIFFERENT frames: must never be read
            d0 = g0 = 0
            for f in fs:
                m[d0:d0 + ovs[f].shape[0], g0:g0 + ovs[f].shape[1]] = ovs[f]
                d0 += ovs[f].shape[0]
                g0 += ovs[f].shape[1]
            parts.append(np.ascontiguousarray(m))
            frames.append(len(fs))
        for mi, cls in enumerate(CLASSES):
            for k, mo in enumerate(ev.MIN_OVERLAPS[cls]):
                prec, rec = scene.class_ap(parts, np.array(frames, np.int64), cls, mo, 5, 60.0)
                np.testing.assert_allclose(prec, want["precision"][mi, k], rtol=0, atol=1e-12, equal_nan=True)
                np.testing.assert_allclose(rec, want["recall"][mi, k], rtol=0, atol=1e-12, equal_nan=True)


@pytest.mark.gpu
def test_rotate_iou_gpu_eval_criteria():
    iou = load_pkg("eval.iou")
    g = golden("eval_ap")
    rb = g["rb"]
    for c, key in ((-1, "crit_m1"), (0, "crit_0"), (1, "crit_1"), (2, "crit_2")):
        got = iou.rotate_iou_gpu_eval(rb[:10], rb[8:], c)
        assert got.dtype == np.float32 and got.shape == (10, 16)
        np.testing.assert_allclose(got, g[key], rtol=0, atol=2e-6)
    assert iou.rotate_iou_gpu_eval(rb[:0], rb, -1).shape == (0, 24)
    assert iou.rotate_iou_gpu_eval(rb.astype(np.float64), rb[:3].astype(np.float64), 2).dtype == np.float64


@pytest.mark.gpu
def test_official_eval_result_matches_reference():
    """53 frames, 3 classes, two ranges: mAP tables, report strings and the precision/recall curves of the reference."""
    ev = load_pkg("eval.eval")
    g, gts, dts = load_sets()
    ov, parts, ngt, ndt = ev.calculate_iou_partly_lidar(dts, gts, "bev", 50)
    assert len(ov) == 53 and len(parts) == 51
    for f in (0, 5):
        np.testing.assert_allclose(ov[f], g[f"ov_bev_{f}"], rtol=0, atol=2e-6)
    ov3, _, _, _ = ev.calculate_iou_partly_lidar(dts, gts, "3d", 50)
    for f in (0, 5):
        np.testing.assert_allclose(ov3[f], g[f"ov_3d_{f}"], rtol=0, atol=2e-6)
    for rt in (80.0, 40.0):
        res, text = ev.get_official_eval_result(gts, dts, CLASSES, rt)
        np.testing.assert_allclose(res[0], g[f"map_bev_{int(rt)}"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(res[1], g[f"map_3d_{int(rt)}"], rtol=0, atol=1e-9)
        assert text == str(g[f"eval_str_{int(rt)}"])
    ret = ev.eval_class_AP(gts, dts, CLASSES, "3d", ev.MIN_OVERLAPS, "lidar", 5, range_thresh=80.0)
    np.testing.assert_allclose(ret["precision"], g["precision_3d_80"], rtol=0, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(ret["recall"], g["recall_3d_80"], rtol=0, atol=1e-12, equal_nan=True)
    # fewer frames than parts: the reference fails in np.concatenate (eval.py:249); here empty parts are skipped
    res7, _ = ev.get_official_eval_result(gts[:7], dts[:7], CLASSES, 80.0)
    want7, _ = E.official_result(gts[:7], dts[:7], CLASSES, 80.0)
    np.testing.assert_allclose(res7[0], want7[0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(res7[1], want7[1], rtol=0, atol=1e-9)
