"""framework.box_torch_ops (reference box_torch_ops.py:18-77): CUDA tensors go through the HIP
kernels of libpp_hip.so; CPU tensors are rejected (no CPU fallback in the product path)."""
import ctypes

import torch

from .. import _lib


def _prep(t):
    if not t.is_cuda:
        raise RuntimeError("box_torch_ops: CUDA tensors only (HIP path, no CPU fallback)")
    return t.contiguous().float()


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def box_decode(box_encodings, anchors):
    enc, anc = _prep(box_encodings), _prep(anchors)
    out = torch.empty_like(enc)
    n = enc.numel() // 7
    with torch.cuda.device(enc.device):
        _lib.check(_lib.load().pp_box_decode(enc.data_ptr(), anc.data_ptr(), out.data_ptr(), n, _s()), None, "pp_box_decode")
    return out


def center_to_corner_box2d(centers, dims, angles=None, origin=0.5):
    if origin != 0.5:
        raise NotImplementedError("only origin=0.5 is on the inference path")
    c, d = _prep(centers), _prep(dims)
    a = _prep(angles) if angles is not None else None
    n = c.shape[0]
    out = torch.empty((n, 4, 2), dtype=torch.float32, device=c.device)
    with torch.cuda.device(c.device):
        _lib.check(_lib.load().pp_corners2d(c.data_ptr(), d.data_ptr(), a.data_ptr() if a is not None else None,
                                            out.data_ptr(), n, _s()), None, "pp_corners2d")
    return out


def corners_nd(dims, origin=0.5):
    d = _prep(dims)
    return center_to_corner_box2d(torch.zeros_like(d), d, None, origin)


def rotation_2d(points, angles):
    rot_sin, rot_cos = torch.sin(angles), torch.cos(angles)
    rot = torch.stack([torch.stack([rot_cos, rot_sin]), torch.stack([-rot_sin, rot_cos])])
    return torch.einsum('aij,jka->aik', (points, rot))


def corner_to_standup_nd(boxes_corner):
    c = _prep(boxes_corner)
    n = c.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=c.device)
    with torch.cuda.device(c.device):
        _lib.check(_lib.load().pp_standup2d(c.data_ptr(), out.data_ptr(), n, _s()), None, "pp_standup2d")
    return out
