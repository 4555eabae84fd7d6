"""framework.inference.Inference (reference inference.py:9-256) and the module functions nms / nms_torch (:689-721).

infer_gpu  -- the path the reference's loop calls (train.py:230): per-class score filter, top-k, decode, NMS, direction
              flip, range mask as ONE pp_postprocess call on the device and one D2H of <= ncls*300 rows.
infer_torch -- the reference's second, staged path (:140-256) as a composition of the C ABI's stage entry points:
              pp_select_candidates (mask gather + sigmoid + threshold + top-k for all classes in one call), pp_box_decode,
              pp_corners2d, pp_standup2d, pp_nms; torch only indexes the <= ncls*1000 selected rows.
Both advance the reference's accumulators p1..p4 (train.py:244-258 prints them per frame)."""
import time

import numpy as np
import torch

from ..engine import engine_for
from . import box_torch_ops
from .nms import nms_gpu


def get_start_result_anno():
    """inference.py:724-737."""
    return {
        'name': np.array([]), 'truncated': np.array([]), 'occluded': np.array([]), 'alpha': np.array([]),
        'bbox': np.zeros([0, 4]), 'dimensions': np.zeros([0, 3]), 'location': np.zeros([0, 3]),
        'rotation_y': np.array([]), 'score': np.array([]),
    }


def nms(bboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
    """inference.py:689-703: numpy boxes [n,4] + scores [n] -> kept indices (int64, best first, cut to post_max_size) or None."""
    dets_np = np.concatenate([bboxes, scores[:, np.newaxis]], axis=1)
    if len(dets_np) == 0:
        keep = np.array([], dtype=np.int64)
    else:
        keep = np.array(nms_gpu(dets_np, iou_threshold), dtype=np.int64)[:post_max_size]
    return None if keep.shape[0] == 0 else keep


def nms_torch(bboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
    """inference.py:706-721: tensor boxes [n,4] + scores [n] -> LongTensor of kept indices or None.  The boxes stay on the
    device (nms_gpu takes tensors); the reference copies them to the host first."""
    dets = torch.cat([bboxes, scores.unsqueeze(-1).float()], dim=1)
    if dets.shape[0] == 0:
        keep = np.array([], dtype=np.int64)
    else:
        dev_id = dets.device.index if dets.is_cuda and dets.device.index is not None else 0
        keep = np.array(nms_gpu(dets, iou_threshold, device_id=dev_id), dtype=np.int64)[:post_max_size]
    return None if keep.shape[0] == 0 else torch.from_numpy(keep).long()


class Inference:
    def __init__(self, config, anchor_assigner, nms_mode="aabb"):
        self.device = config['device']
        self._config = config
        eng = engine_for(config)
        self.anchors = torch.from_numpy(anchor_assigner.anchors).to(eng.device)
        self._nms_pre_max_size = 1000
        self._nms_post_max_size = 300
        self._nms_iou_threshold = 0.1
        self._box_code_size = 7
        self._num_class = 1
        self._use_direction_classifier = True
        self._nms_score_threshold = torch.tensor([0.05]).to(eng.device)
        self.center_limit = config['center_limit']
        self.detect_class = np.array(config['detect_class'])
        self.class_masks = anchor_assigner.class_masks
        self.nms_mode = 1 if nms_mode in ("rotate", "rotated", 1) else 0
        self.profile_stages = True  # like the reference, which synchronises around every bucket
        self.p1, self.p2, self.p3, self.p4, self.p5 = 0.0, 0.0, 0.0, 0.0, 0.0

    def infer_device(self, example, preds_dict):
        """Device tensors out: det f32[ncls*300,9] (x,y,z,l,w,h,r,score,class), cnt i32[1+PP_MAX_CLASSES]; no sync."""
        eng = engine_for(self._config)
        return eng.postprocess(preds_dict["cls_preds"].contiguous(), preds_dict["box_preds"].contiguous(),
                               preds_dict["dir_preds"].contiguous(), example["anchors_mask"].reshape(-1).contiguous(),
                               self.nms_mode)

    def infer_gpu(self, example, preds_dict):
        """inference.py:26-138.  The buckets come from HIP events between the kernels of the fused call:
        p1 mask / sigmoid / threshold / candidate gather, p2 exact top-k (box decode runs in the same kernel, so p3 -- the
        reference's host decode -- stays 0 here), p4 NMS + direction flip + range mask + the D2H of the result."""
        eng = engine_for(self._config)
        start = time.time()
        if self.profile_stages:
            eng.stage_profile_begin()
        det, cnt = self.infer_device(example, preds_dict)
        cnt = cnt.cpu().numpy()
        k = int(cnt[0])
        rows = det[:k].cpu().numpy()
        if self.profile_stages:
            ms = eng.stage_profile_end()
            self.p1 += ms["post_filter"] * 1e-3
            self.p2 += ms["post_topk_decode"] * 1e-3
            self.p4 += max(time.time() - start - (ms["post_filter"] + ms["post_topk_decode"]) * 1e-3, 0.0)
        else:
            self.p4 += time.time() - start
        anno = get_start_result_anno()
        if k > 0:
            names = list(self.class_masks.keys())
            anno["name"] = np.array([names[int(c)] for c in rows[:, 8]], dtype='<U10')
            anno["location"] = rows[:, :3]
            anno["dimensions"] = rows[:, 3:6]
            anno["rotation_y"] = rows[:, 6]
            anno["score"] = rows[:, 7]
        return [anno]

    def infer_torch(self, example, preds_dict):
        """The reference's staged post-processing (inference.py:140-256) as a composition of the engine's stage entry points,
        all classes at once where the stage allows it:
          p1  pp_select_candidates -- per class mask gather, sigmoid, score >= 0.05, exact top-1000 in ONE device call
              (the reference's p1 and p2 buckets; its boolean-mask indexing over up to 960 k rows per class is gone);
          p2  gather of the <= ncls*1000 selected rows (box / dir logits, anchors);
          p3  pp_box_decode, pp_corners2d, pp_standup2d on the concatenated candidates of all classes;
          p4  pp_nms per class (NMS is per class by definition), then direction flip, the range-mask quirk
              (:107-109 compares dims with the upper limits) and limit_period on the kept rows, one D2H."""
        eng = engine_for(self._config)
        dev = eng.device
        tick = (lambda: (torch.cuda.synchronize(), time.time())[1]) if self.profile_stages else time.time
        t0 = tick()
        box_all = preds_dict["box_preds"].contiguous().view(-1, 7)
        dir_all = preds_dict["dir_preds"].contiguous().view(-1, 2)
        idx, score, count = eng.select_candidates(preds_dict["cls_preds"].contiguous(), box_all, dir_all,
                                                  example["anchors_mask"].reshape(-1).contiguous())
        counts = count.cpu().tolist()
        t1 = tick()
        live = torch.arange(idx.shape[1], device=dev)[None, :] < count[:, None]
        rows = idx[live].long()            # class-major, descending score inside a class
        scores = score[live]
        enc, anc = box_all[rows], self.anchors[rows]
        dir_label = dir_all[rows, 1] > dir_all[rows, 0]
        t2 = tick()
        dec = box_torch_ops.box_decode(enc, anc) if rows.numel() else enc
        standup = (box_torch_ops.corner_to_standup_nd(box_torch_ops.center_to_corner_box2d(dec[:, :2], dec[:, 3:5], dec[:, 6]))
                   if rows.numel() else dec[:, :4])
        t3 = tick()
        kept, kept_cls, off = [], [], 0
        for ci, n in enumerate(counts):
            sel = nms_torch(standup[off:off + n], scores[off:off + n], pre_max_size=self._nms_pre_max_size,
                            post_max_size=self._nms_post_max_size, iou_threshold=self._nms_iou_threshold) if n else None
            if sel is not None:
                kept.append(sel.to(dev) + off)
                kept_cls.append(torch.full((sel.numel(),), ci, dtype=torch.int64))
            off += n
        anno = get_start_result_anno()
        if kept:
            k = torch.cat(kept)
            out = dec[k].clone()
            out[:, 6] += torch.where((out[:, 6] > 0) ^ dir_label[k], torch.tensor(np.pi, dtype=out.dtype, device=dev), torch.zeros((), dtype=out.dtype, device=dev))
            out = out.cpu().numpy()
            sc = scores[k].cpu().numpy()
            ci = torch.cat(kept_cls).numpy()
            lim = np.asarray(self.center_limit)
            ok = np.any(out[:, :3] > lim[:3], axis=1) & np.any(out[:, 3:6] < lim[3:], axis=1)
            out, sc, ci = out[ok], sc[ok], ci[ok]
            out[:, 6] -= np.floor(out[:, 6] / (2 * np.pi) + 0.5) * (2 * np.pi)  # box_np_ops.limit_period(r, 0.5, 2*pi)
            if out.shape[0]:
                names = np.array(list(self.class_masks.keys()), dtype='<U10')
                anno.update(name=names[ci], location=out[:, :3], dimensions=out[:, 3:6], rotation_y=out[:, 6], score=sc)
        t4 = tick()
        self.p1 += t1 - t0
        self.p2 += t2 - t1
        self.p3 += t3 - t2
        self.p4 += t4 - t3
        return [anno]
