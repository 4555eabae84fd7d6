"""framework.nms.nms_gpu (reference nms.py:6-40) and eval.iou.rotate_nms_gpu (eval/iou.py:438-473):
sort, 64x64 bitmask tiles and the greedy sweep all run on the device (pp_nms)."""
import ctypes

import numpy as np
import torch

from .. import _lib


def _run(dets, thresh, rotate, device_id):
    stride = 6 if rotate else 5
    dev = torch.device("cuda", device_id)
    if isinstance(dets, torch.Tensor):
        d = dets.to(dev, torch.float32).contiguous()
    else:
        d = torch.from_numpy(np.ascontiguousarray(dets, dtype=np.float32)).to(dev)
    n = int(d.shape[0])
    if n == 0:
        return []
    if d.shape[1] != stride:
        raise ValueError(f"dets must be [n,{stride}]")
    keep = torch.empty((n,), dtype=torch.int32, device=dev)
    nk = torch.zeros((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.load().pp_nms(d.data_ptr(), n, stride, float(thresh), keep.data_ptr(), nk.data_ptr(), int(rotate), s),
                   None, "pp_nms")
    k = int(nk.item())
    return [int(v) for v in keep[:k].cpu().numpy()]


def nms_gpu(dets, nms_overlap_thresh, device_id=0):
    """dets f32[n,5] (x1,y1,x2,y2,score) -> list of kept indices into dets, best score first."""
    return _run(dets, nms_overlap_thresh, False, device_id)


def rotate_nms_gpu(dets, nms_overlap_thresh, device_id=0):
    """dets f32[n,6] (cx,cy,dx,dy,angle,score)."""
    return _run(dets, nms_overlap_thresh, True, device_id)


def rotate_iou_gpu(boxes, query_boxes, device_id=0):
    """N x K rotated IoU (eval/iou.py:509-560 with criterion -1). boxes [N,5], query_boxes [K,5]."""
    dev = torch.device("cuda", device_id)
    a = torch.as_tensor(np.ascontiguousarray(boxes, dtype=np.float32)).to(dev)
    b = torch.as_tensor(np.ascontiguousarray(query_boxes, dtype=np.float32)).to(dev)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.load().pp_rotated_iou(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], b.shape[0], s), None, "pp_rotated_iou")
    return out.cpu().numpy()
