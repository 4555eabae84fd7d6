"""eval.iou of the reference: rotate_iou_gpu_eval (eval/iou.py:606-638), rotate_iou_gpu (:509-560) and
rotate_nms_gpu (:438-473) with the reference's numpy-in / numpy-out signatures; the work runs in libpp_hip.so."""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..framework.nms import rotate_iou_gpu, rotate_nms_gpu  # noqa: F401  (same signatures as the reference)


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """boxes [N,5], query_boxes [K,5] (cx,cy,dx,dy,angle; clockwise positive) -> [N,K] in boxes' dtype.
    criterion -1: IoU, 0: intersection / area(query), 1: intersection / area(box), 2: intersection area
    (the reference's kernel passes the query box first to devRotateIoUEval, eval/iou.py:600-603)."""
    boxes = np.asarray(boxes)
    query_boxes = np.asarray(query_boxes)
    box_dtype = boxes.dtype
    n, k = int(boxes.shape[0]), int(query_boxes.shape[0])
    if n == 0 or k == 0:
        return np.zeros((n, k), dtype=np.float32).astype(box_dtype)
    dev = torch.device("cuda", device_id)
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32)).to(dev)
    q = torch.from_numpy(np.ascontiguousarray(query_boxes, dtype=np.float32)).to(dev)
    out = torch.empty((n, k), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.load().pp_rotated_iou_eval(b.data_ptr(), q.data_ptr(), out.data_ptr(), n, k, int(criterion), s), None,
                   "pp_rotated_iou_eval")
    return out.cpu().numpy().astype(box_dtype)
