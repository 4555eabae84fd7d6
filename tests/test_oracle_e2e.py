"""CPU: the whole oracle pipeline on the eight_20cm frame (seed 1000) against the annos the
reference produced for the same frame and weights (tests/golden/e2e_eight_20cm_*.npz)."""
import numpy as np
import pytest

from conftest import golden
from oracle import c_oracle as C
from oracle import pp_oracle as O


def run_oracle_frame(synth, name, seed, sd, norm="instance", nms_mode="aabb"):
    cfg = synth.load_config(name)
    s = O.voxel_setup(cfg)
    a = O.make_anchors(s)
    pts = synth.lidar_cloud(name, seed=seed)
    v, c, n = C.points_to_voxels(pts, s["voxel_size"], s["offset"], s["grid_size"], cfg["max_voxels"], cfg["max_num_points"])
    mask = C.create_mask(c, s["grid_size"], a["anchors_coors"])
    feat = O.pfn(v, n, c, sd, s)
    canvas = O.scatter(feat, c, s["grid_size"])
    rpn = O.backbone(canvas, sd, norm)
    cls, box, dr = O.head(rpn, sd)
    det, counts = O.postprocess(cls, box, dr, mask, a["anchors"], a["class_masks"], cfg["center_limit"], nms_mode)
    return dict(feat=feat, rpn=rpn, cls=cls, box=box, dir=dr, det=det, counts=counts, mask=mask)


def match_dets(det, ref, tol=1e-3):
    """Greedy nearest match on (class, x, y); returns fraction of ref rows matched within tol."""
    if ref.shape[0] == 0:
        return 1.0 if det.shape[0] == 0 else 0.0
    used = np.zeros(det.shape[0], dtype=bool)
    ok = 0
    for r in ref:
        d = np.abs(det[:, :8] - r[None, :8]).max(axis=1) + (det[:, 8] != r[8]) * 1e3 + used * 1e3
        j = int(np.argmin(d)) if det.shape[0] else -1
        if j >= 0 and d[j] <= tol:
            used[j] = True
            ok += 1
    return ok / ref.shape[0]


@pytest.mark.parametrize("tag,cls_bias", [("rand", None), ("trained", -4.6)])
def test_oracle_frame_vs_reference(tag, cls_bias, synth):
    g = golden(f"e2e_eight_20cm_{tag}")
    r = run_oracle_frame(synth, "eight_20cm", 1000, synth.seeded_state_dict(0, cls_bias=cls_bias))
    np.testing.assert_allclose(r["feat"][:64], g["pfn_rows"], atol=1e-5)
    np.testing.assert_allclose(r["rpn"].reshape(-1)[g["rpn_idx"]], g["rpn_vals"], atol=1e-5)
    np.testing.assert_allclose(r["cls"].reshape(-1)[g["pred_idx"]], g["cls_vals"], atol=1e-5)
    np.testing.assert_allclose(r["box"].reshape(-1, 7)[g["pred_idx"]], g["box_vals"], atol=1e-5)
    np.testing.assert_allclose(r["dir"].reshape(-1, 2)[g["pred_idx"]], g["dir_vals"], atol=1e-5)
    ref = np.concatenate([g["location"], g["dimensions"], g["rotation_y"][:, None], g["score"][:, None],
                          g["cls_idx"][:, None].astype(np.float32)], axis=1)
    assert r["det"].shape[0] == ref.shape[0]
    # same rows in the same order (the oracle fixes tie order; scores here are tie-free)
    np.testing.assert_allclose(r["det"], ref, rtol=0, atol=1e-5)
