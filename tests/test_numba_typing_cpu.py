"""CPU: robustness of the NMS keep lists against numba's typing of the reference's device functions.

The golden vectors were produced by running the reference's `@jit` / `@cuda.jit` functions as plain Python (numba is absent
from the container), where every intermediate stays float32.  Real numba types `float32 + 1` (an int64 literal,
framework/nms.py:111-115) and `/ 2.0` (eval/iou.py:170-177) as float64: the '+1' widths, the areas, the triangle-fan sum and
the final division then run in double.  Only a pair whose IoU lies within ~1e-7 of the threshold can tell the two typings
apart.  This test runs BOTH typings of the oracle (oracle/pp_oracle.c: `orc_set_numba_typing`) over every NMS golden and
over the candidate sets of the end-to-end golden frames, requires identical keep lists, and reports the smallest
|IoU - threshold| met on the way (the safety margin of the statement "keep lists bit-exact vs the reference")."""
import numpy as np
import pytest

from conftest import golden
from oracle import c_oracle as C
from oracle import pp_oracle as O

THR = 0.1


def both_typings(fn, dets, thr):
    with C.numba_typing(False) as plain:
        k32 = fn(dets, thr)
    with C.numba_typing(True) as typed:
        k64 = fn(dets, thr)
    return k32, k64, min(plain.margin, typed.margin)


def test_aabb_goldens_keep_lists_do_not_depend_on_the_typing():
    g = golden("nms_aabb")
    worst = 1e30
    for n in (1, 63, 64, 65, 300, 1000):
        k32, k64, m = both_typings(C.nms_aabb, g[f"dets_{n}"], THR)
        assert k32 == k64 == [int(v) for v in g[f"keep_{n}"]], n
        worst = min(worst, m)
    print(f"[numba typing] AABB goldens: keep lists identical under both typings; smallest |IoU - thr| = {worst:.3e}")
    assert worst > 1e-6


def test_rotated_goldens_keep_lists_do_not_depend_on_the_typing():
    g = golden("nms_rotated")
    worst = 1e30
    for dk, kk in (("dets", "keep"), ("dets200", "keep200")):
        k32, k64, m = both_typings(C.nms_rotated, g[dk], THR)
        assert k32 == k64 == [int(v) for v in g[kk]], dk
        worst = min(worst, m)
    print(f"[numba typing] rotated goldens: keep lists identical under both typings; smallest |IoU - thr| = {worst:.3e}")
    assert worst > 1e-6


@pytest.mark.parametrize("tag", ["rand", "trained"])
def test_e2e_frames_keep_lists_do_not_depend_on_the_typing(tag, synth):
    """The candidate sets of the end-to-end golden frames (<= 1000 boxes per class after score filter + top-k), AABB and
    rotated: same survivors under both typings, and the detections are still the reference's annos."""
    from frame_check import oracle_frame
    bias = None if tag == "rand" else -4.6
    sd = synth.seeded_state_dict(0, cls_bias=bias)
    r = oracle_frame(synth, "eight_20cm", synth.lidar_cloud("eight_20cm", seed=1000), sd)
    g = golden(f"e2e_eight_20cm_{tag}")
    ref = np.concatenate([g["location"], g["dimensions"], g["rotation_y"][:, None], g["score"][:, None], g["cls_idx"][:, None].astype(np.float32)], axis=1)
    for mode, fn in (("aabb", C.nms_aabb), ("rotated", C.nms_rotated)):
        out = {}
        margins = []
        for typed in (False, True):
            with C.numba_typing(typed) as t:
                det, counts, info = O.postprocess(r["cls"], r["box"], r["dir"], r["mask"], r["anchors"], r["class_masks"], r["center_limit"], mode,
                                                  detail=True, nms_fn=fn)
            margins.append(t.margin)
            out[typed] = (det, counts, [list(map(int, i["idx"][i["keep"]])) for i in info])
        assert out[False][1] == out[True][1] and out[False][2] == out[True][2], (tag, mode)
        np.testing.assert_array_equal(out[False][0], out[True][0])
        if mode == "aabb":
            np.testing.assert_allclose(out[True][0], ref, rtol=0, atol=1e-5)
        print(f"[numba typing] e2e eight_20cm {tag} {mode}: {sum(out[True][1])} detections identical under both typings; "
              f"smallest |IoU - thr| = {min(margins):.3e}")
        assert min(margins) > 1e-7
