"""Deterministic synthetic LiDAR clouds and seeded weights for tests and bench.py.

The reference reads KITTI-layout ``*.bin`` files (train.py:222) from private
datasets; none exist here, so every test and benchmark runs on seeded synthetic
clouds of the same shape.  Uniform-random clouds would give ~1 point per pillar
and trip the ``max_voxels`` break of voxel_generator.py:96-97, so the clouds are
LiDAR-shaped: spinning multi-beam sensor, ground plane, a few boxes, emitted in
sensor (azimuth-major) order.  SURVEY.md section 8(d) fixes the parameters.
"""
import json
import os

import numpy as np

_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")

# name -> (beams, elev_lo_deg, elev_hi_deg, n_points, sensor_height)
CLOUD_SHAPES = {
    "nuscene_10class": (32, -30.0, 10.0, 34000, 1.8),
    "eight_20cm": (64, -24.8, 2.0, 20000, 1.7),
    "ntusl_10cm": (64, -24.8, 2.0, 60000, 1.7),
    "nuscene": (32, -30.0, 10.0, 34000, 1.8),
}


def load_config(name):
    """Load one of the shipped hot-path configs (same values as the reference's
    configs/<name>.json; nuscene.json there is invalid JSON, ours is valid)."""
    path = name if os.path.isfile(name) else os.path.join(_CFG_DIR, name + ".json")
    with open(path, "r") as f:
        txt = f.read()
    try:
        return json.loads(txt)
    except json.JSONDecodeError:
        # tolerate trailing commas (the reference's nuscene.json:24 has one)
        import re
        return json.loads(re.sub(r",\s*([}\]])", r"\1", txt))


def _ray_box(o, d, lo, hi):
    """Slab ray/AABB intersection, vectorised over rays. Returns t (inf = miss)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (lo - o) * inv
        t1 = (hi - o) * inv
    tmin = np.minimum(t0, t1).max(axis=1)
    tmax = np.maximum(t0, t1).min(axis=1)
    hit = (tmax >= np.maximum(tmin, 0.0))
    return np.where(hit, np.maximum(tmin, 0.0), np.inf)


def lidar_cloud(shape="eight_20cm", seed=1000, n_points=None, max_range=75.0):
    """points f32[N,4] (x, y, z, intensity) in sensor order.

    64/32 beams, 0.17 deg azimuth step, ground plane at z=-h, 12 boxes
    (4.5 x 1.9 x 1.6 m) at 5-60 m, sigma=2 cm range noise, subsampled to N.
    """
    beams, e_lo, e_hi, n_def, h = CLOUD_SHAPES[shape]
    n_points = n_def if n_points is None else n_points
    rng = np.random.default_rng(seed)
    az = np.deg2rad(np.arange(0.0, 360.0, 0.17))
    el = np.deg2rad(np.linspace(e_lo, e_hi, beams))
    A, E = np.meshgrid(az, el, indexing="ij")  # azimuth-major
    A = A.ravel()
    E = E.ravel()
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=1)
    o = np.zeros(3)
    with np.errstate(divide="ignore"):
        t = np.where(d[:, 2] < 0, -h / d[:, 2], np.inf)
    nbox = 12
    r = rng.uniform(5.0, 60.0, nbox)
    th = rng.uniform(0, 2 * np.pi, nbox)
    for i in range(nbox):
        c = np.array([r[i] * np.cos(th[i]), r[i] * np.sin(th[i]), -h + 0.8])
        half = np.array([2.25, 0.95, 0.8]) if rng.uniform() < 0.5 else np.array([0.95, 2.25, 0.8])
        t = np.minimum(t, _ray_box(o, d, c - half, c + half))
    ok = np.isfinite(t) & (t <= max_range) & (t > 1.0)
    idx = np.nonzero(ok)[0]
    if idx.size > n_points:
        idx = np.sort(rng.choice(idx, size=n_points, replace=False))
    t = t[idx] + rng.normal(0.0, 0.02, idx.size)
    pts = d[idx] * t[:, None]
    inten = rng.uniform(0.0, 1.0, idx.size)
    out = np.concatenate([pts, inten[:, None]], axis=1).astype(np.float32)
    if out.shape[0] < n_points:  # top up (merged second sweep) so N is exact
        extra = lidar_cloud(shape, seed + 7919, n_points - out.shape[0], max_range)
        out = np.concatenate([out, extra], axis=0)
    return np.ascontiguousarray(out)


def seeded_state_dict(seed=0, norm="instance", cls_bias=None, num_anchor_per_loc=9):
    """Seeded random weights for tests / bench.py: the network's own initialiser (networks/init.py) with a chosen seed."""
    from .networks.init import init_state_dict
    return init_state_dict(seed, norm=norm, cls_bias=cls_bias, num_anchor_per_loc=num_anchor_per_loc)
