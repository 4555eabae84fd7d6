"""framework.inference.Inference (reference inference.py:9-256) and the module functions nms / nms_torch (:689-721).

infer_gpu  -- the path the reference's loop calls (train.py:230): per-class score filter, top-k, decode, NMS, direction
              flip, range mask as ONE pp_postprocess call on the device and one D2H of <= ncls*300 rows.
infer_torch -- the reference's second path (:140-256), stage by stage on the device like the original: torch indexing /
              sigmoid / topk as the container plumbing the reference itself uses there, box decode / corners / stand-up
              boxes / NMS through the HIP entry points (pp_box_decode, pp_corners2d, pp_standup2d, pp_nms).
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
        """inference.py:140-256, stage by stage with the same four synchronised buckets per class."""
        cls_all = preds_dict["cls_preds"].squeeze(0)
        box_all = preds_dict["box_preds"].squeeze(0)
        dir_all = preds_dict["dir_preds"].squeeze(0)
        anchors_mask = example["anchors_mask"].squeeze(0)
        name_list, location_list, dimensions_list, rotation_y_list, score_list = [], [], [], [], []
        sync = torch.cuda.synchronize if cls_all.is_cuda else (lambda: None)
        for cls, a_range in self.class_masks.items():
            sync()
            start = time.time()
            a_mask = anchors_mask[a_range[0]: a_range[1]]
            box_preds = box_all[a_range[0]: a_range[1]][a_mask]
            cls_preds = cls_all[a_range[0]: a_range[1]][a_mask]
            dir_preds = dir_all[a_range[0]: a_range[1]][a_mask]
            anchors = self.anchors[a_range[0]: a_range[1]][a_mask]
            cls_scores = torch.sigmoid(cls_preds)
            top_scores = torch.max(cls_scores, dim=-1)[0]
            dir_labels = torch.max(dir_preds, dim=-1)[1]
            selected = None
            p1 = p2 = p3 = start
            keep = top_scores >= self._nms_score_threshold
            if keep.any():
                top_scores, box_preds, dir_labels, anchors = top_scores[keep], box_preds[keep], dir_labels[keep], anchors[keep]
                pre_max_size = min(top_scores.shape[0], self._nms_pre_max_size)
                sync()
                p1 = time.time()
                top_scores, indices = torch.topk(top_scores, k=pre_max_size)
                box_preds, dir_labels, anchors = box_preds[indices], dir_labels[indices], anchors[indices]
                sync()
                p2 = time.time()
                box_preds = box_torch_ops.box_decode(box_preds, anchors)
                boxes_for_nms = box_preds[:, [0, 1, 3, 4, 6]]
                corners = box_torch_ops.center_to_corner_box2d(boxes_for_nms[:, :2], boxes_for_nms[:, 2:4], boxes_for_nms[:, 4])
                boxes_for_nms = box_torch_ops.corner_to_standup_nd(corners)
                sync()
                p3 = time.time()
                selected = nms_torch(boxes_for_nms, top_scores, pre_max_size=self._nms_pre_max_size,
                                     post_max_size=self._nms_post_max_size, iou_threshold=self._nms_iou_threshold)
                if selected is not None:
                    selected = selected.to(box_preds.device)
            sync()
            p4 = time.time()
            if selected is not None:
                box_preds = box_preds[selected]
                scores_preds = top_scores[selected]
                opp_labels = (box_preds[..., -1] > 0) ^ dir_labels[selected].bool()
                box_preds[..., -1] += torch.where(opp_labels, torch.tensor(np.pi).type_as(box_preds), torch.tensor(0.0).type_as(box_preds))
                scores_preds = scores_preds.detach().cpu().numpy()
                box_preds = box_preds.detach().cpu().numpy()
                limit_range = self.center_limit
                range_mask = np.any(box_preds[:, :3] > limit_range[:3], axis=1) & np.any(box_preds[:, 3:6] < limit_range[3:], axis=1)
                box_preds = box_preds[range_mask]
                r = box_preds[..., -1]
                box_preds[..., -1] = r - np.floor(r / (2 * np.pi) + 0.5) * (2 * np.pi)  # box_np_ops.limit_period(r, 0.5, 2*pi)
                scores_preds = scores_preds[range_mask]
                dt_num = box_preds.shape[0]
                if dt_num > 0:
                    name_list.append(np.full(dt_num, cls, dtype='<U10'))
                    location_list.append(box_preds[:, :3])
                    dimensions_list.append(box_preds[:, 3:6])
                    rotation_y_list.append(box_preds[:, 6])
                    score_list.append(scores_preds)
            self.p1 += p1 - start
            self.p2 += p2 - p1
            self.p3 += p3 - p2
            self.p4 += p4 - p3
        anno = get_start_result_anno()
        if len(name_list) > 0:
            anno["name"] = np.concatenate(name_list)
            anno["location"] = np.concatenate(location_list)
            anno["dimensions"] = np.concatenate(dimensions_list)
            anno["rotation_y"] = np.concatenate(rotation_y_list)
            anno["score"] = np.concatenate(score_list)
        return [anno]
