"""ctypes loader for oracle/pp_oracle.c (TEST INFRASTRUCTURE ONLY, see its header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_rotated_iou.restype = ctypes.c_float
        _lib.orc_min_margin.restype = ctypes.c_double
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def points_to_voxels(points, voxel_size, offset, grid_size, max_voxels, max_num_points):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n, f = pts.shape
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    off = np.ascontiguousarray(offset, dtype=np.float32)
    g = np.ascontiguousarray(grid_size, dtype=np.int32)
    voxels = np.empty((max_voxels, max_num_points, f), dtype=np.float32)
    coors = np.empty((max_voxels, 3), dtype=np.int32)
    num = np.empty((max_voxels,), dtype=np.int32)
    scratch = np.empty(int(g[0]) * int(g[1]) * int(g[2]), dtype=np.int32)
    nv = lib().orc_points_to_voxels(_p(pts, ctypes.c_float), n, f, _p(vs, ctypes.c_float), _p(off, ctypes.c_float),
                                    _p(g, ctypes.c_int32), int(max_voxels), int(max_num_points),
                                    _p(voxels, ctypes.c_float), _p(coors, ctypes.c_int32), _p(num, ctypes.c_int32),
                                    _p(scratch, ctypes.c_int32))
    return voxels[:nv], coors[:nv], num[:nv]


def create_mask(coors, grid_size, anchors_coors):
    co = np.ascontiguousarray(coors, dtype=np.int32)
    rects = np.ascontiguousarray(anchors_coors, dtype=np.int32)
    gx, gy = int(grid_size[0]), int(grid_size[1])
    mask = np.empty(rects.shape[0], dtype=np.uint8)
    scratch = np.empty(gx * gy, dtype=np.int32)
    lib().orc_anchor_mask(_p(co, ctypes.c_int32), co.shape[0], gx, gy, _p(rects, ctypes.c_int32), rects.shape[0],
                          _p(mask, ctypes.c_uint8), _p(scratch, ctypes.c_int32))
    return mask.astype(bool)


def nms_aabb(dets, thresh):
    d = np.ascontiguousarray(dets, dtype=np.float32)
    keep = np.empty(max(d.shape[0], 1), dtype=np.int32)
    n = lib().orc_nms_aabb(_p(d, ctypes.c_float), d.shape[0], ctypes.c_float(thresh), _p(keep, ctypes.c_int32))
    return [int(v) for v in keep[:n]]


def nms_rotated(dets, thresh):
    d = np.ascontiguousarray(dets, dtype=np.float32)
    keep = np.empty(max(d.shape[0], 1), dtype=np.int32)
    n = lib().orc_nms_rotated(_p(d, ctypes.c_float), d.shape[0], ctypes.c_float(thresh), _p(keep, ctypes.c_int32))
    return [int(v) for v in keep[:n]]


def rotated_iou(r1, r2):
    a = np.ascontiguousarray(r1, dtype=np.float32)
    b = np.ascontiguousarray(r2, dtype=np.float32)
    return float(lib().orc_rotated_iou(_p(a, ctypes.c_float), _p(b, ctypes.c_float)))


def rotated_iou_eval(boxes, qboxes, criterion=-1):
    """rotate_iou_gpu_eval (eval/iou.py:606-638): f32 in/out, result cast back to the input dtype."""
    dt = np.asarray(boxes).dtype
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    q = np.ascontiguousarray(qboxes, dtype=np.float32)
    out = np.zeros((b.shape[0], q.shape[0]), dtype=np.float32)
    if b.shape[0] and q.shape[0]:
        lib().orc_rotated_iou_eval(_p(b, ctypes.c_float), b.shape[0], _p(q, ctypes.c_float), q.shape[0], int(criterion), _p(out, ctypes.c_float))
    return out.astype(dt)


class numba_typing:
    """Context manager: the NMS routines follow numba's typing of the reference's device functions (`float32 + 1` and
    `/ 2.0` promote to float64 -- see pp_oracle.c) instead of the all-fp32 arithmetic the golden vectors pin.  `.margin`
    after the block = the smallest |IoU - threshold| any NMS run inside it has seen."""

    def __init__(self, on=True):
        self.on = on
        self.margin = None

    def __enter__(self):
        lib().orc_set_numba_typing(1 if self.on else 0)
        lib().orc_reset_min_margin()
        return self

    def __exit__(self, *exc):
        self.margin = float(lib().orc_min_margin())
        lib().orc_set_numba_typing(0)
        return False
