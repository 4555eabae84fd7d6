"""framework.inference.Inference (reference inference.py:9-138): per-class score filter, top-k,
decode, NMS, direction flip, range mask -- one pp_postprocess call, one D2H of <= 900 rows."""
import time

import numpy as np
import torch

from ..engine import engine_for


def get_start_result_anno():
    """inference.py:724-737."""
    return {
        'name': np.array([]), 'truncated': np.array([]), 'occluded': np.array([]), 'alpha': np.array([]),
        'bbox': np.zeros([0, 4]), 'dimensions': np.zeros([0, 3]), 'location': np.zeros([0, 3]),
        'rotation_y': np.array([]), 'score': np.array([]),
    }


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
        self.p1, self.p2, self.p3, self.p4, self.p5 = 0.0, 0.0, 0.0, 0.0, 0.0

    def infer_device(self, example, preds_dict):
        """Device tensors out: det f32[900,9] (x,y,z,l,w,h,r,score,class), cnt i32[1+ncls]; no sync."""
        eng = engine_for(self._config)
        return eng.postprocess(preds_dict["cls_preds"].contiguous(), preds_dict["box_preds"].contiguous(),
                               preds_dict["dir_preds"].contiguous(), example["anchors_mask"].reshape(-1).contiguous(),
                               self.nms_mode)

    def infer_gpu(self, example, preds_dict):
        start = time.time()
        det, cnt = self.infer_device(example, preds_dict)
        cnt = cnt.cpu().numpy()
        k = int(cnt[0])
        rows = det[:k].cpu().numpy()
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

    infer_torch = infer_gpu
