"""CPU: the whole-frame checker (tests/frame_check.py) itself -- logit noise at the GPU's level must pass with
every differing row explained as a near-tie; a planted, non-tie difference must be caught."""
import numpy as np
import pytest

from frame_check import compare_frame, oracle_frame
from oracle import c_oracle as C
from oracle import pp_oracle as O


@pytest.fixture(scope="module")
def frame(synth):
    sd = synth.seeded_state_dict(1, cls_bias=-3.0)
    return oracle_frame(synth, "nuscene", synth.lidar_cloud("nuscene", seed=77), sd)


def fake_gpu(ref, noise, seed, nms_mode="aabb"):
    rng = np.random.default_rng(seed)
    g = {k: (ref[k] + rng.uniform(-noise, noise, ref[k].shape)).astype(np.float32) for k in ("cls", "box", "dir")}
    g["mask"] = ref["mask"]
    det, counts = O.postprocess(g["cls"], g["box"], g["dir"], ref["mask"], ref["anchors"], ref["class_masks"], ref["center_limit"], nms_mode,
                                nms_fn=C.nms_rotated if nms_mode == "rotated" else C.nms_aabb)
    return g, det, np.array([det.shape[0]] + counts, np.int32)


@pytest.mark.parametrize("noise,seed", [(0.0, 0), (5e-6, 1), (2e-5, 2), (1e-4, 3), (8e-4, 4), (8e-4, 5)])
def test_noise_is_explained(frame, noise, seed):
    g, det, cnt = fake_gpu(frame, noise, seed)
    # the cap on explained rows is for GPU-level deviations (<= 2e-5); the larger noise levels test the explanations themselves
    rep = compare_frame(frame, g, det, cnt, "aabb", f"noise {noise}", max_explained=None if noise <= 2e-5 else 10 ** 6)
    assert rep["n_ref"] > 50 and rep["matched"] >= rep["n_ref"] - rep["differing"]
    if noise == 0.0:
        assert rep["differing"] == 0 and rep["max_matched_dev"] == 0.0


def test_rotated_mode(frame):
    g, det, cnt = fake_gpu(frame, 2e-5, 3, "rotated")
    rep = compare_frame(frame, g, det, cnt, "rotated", "rotated noise 2e-5")
    assert isinstance(rep["reasons"], dict) and sum(rep["reasons"].values()) >= rep["differing"]


def test_planted_difference_is_caught(frame):
    g, det, cnt = fake_gpu(frame, 0.0, 0)
    # drop one confident detection from the "GPU" result: not a near-tie -> must fail at step 3 (selection differs)
    with pytest.raises(AssertionError):
        compare_frame(frame, g, det[1:], np.array([det.shape[0] - 1, cnt[1] - 1, cnt[2], cnt[3]], np.int32), "aabb", "planted drop")
    # a logit moved by 0.5 on one strong anchor: caught by the logit bound
    g2 = {k: v.copy() for k, v in g.items()}
    a = int(np.argmax(np.where(frame["mask"], frame["cls"], -1e9)))
    g2["cls"][a] -= 0.5
    with pytest.raises(AssertionError):
        compare_frame(frame, g2, det, cnt, "aabb", "planted logit")
    # matched row moved by 5e-3 while logits pretend to be fine: caught by the matched-row bound or the self check
    det3 = det.copy()
    det3[0, 0] += 5e-3
    with pytest.raises(AssertionError):
        compare_frame(frame, g, det3, cnt, "aabb", "planted box shift")


def test_explained_rows_are_capped(frame):
    """Noise far above the GPU's level produces many (individually explainable) flips: the default cap must reject the frame."""
    g, det, cnt = fake_gpu(frame, 8e-4, 4)
    rep = compare_frame(frame, g, det, cnt, "aabb", "noise 8e-4 uncapped", max_explained=10 ** 6)
    if rep["differing"] > max(4, int(np.ceil(0.005 * rep["n_ref"]))):
        with pytest.raises(AssertionError):
            compare_frame(frame, g, det, cnt, "aabb", "noise 8e-4 capped")
