"""framework.anchor_assigner.AnchorAssigner, inference half (reference anchor_assigner.py:220-335).
Target assignment (`assign`, :337-425) is training-only and out of scope."""
import numpy as np
import torch

from ..engine import class_table_of, engine_for


class AnchorAssigner:
    def __init__(self, config):
        # the reference overwrites detect_class and adds the per-class dicts (:222-245); a config with its own
        # `class_table` (build-side extension, engine.class_table_of) keeps its classes
        names, table = class_table_of(config)
        config['detect_class'] = list(names)
        self.detect_class = config['detect_class']
        fm = [int(config['grid_size'][0]) // 2, int(config['grid_size'][1]) // 2, 1]
        for name in names:
            t = table[name]
            config[name] = dict(sizes=[list(s) for s in t['sizes']], rotations=list(t['rotations']),
                                feature_map_size=[list(fm) for _ in t['sizes']],
                                matched_threshold=t.get('matched_threshold', 0.6), unmatched_threshold=t.get('unmatched_threshold', 0.45))
        self.anchor_offsets = config['detection_offset']
        self.grid_size = config['grid_size']
        self.box_code_size = config['box_code_size']
        self._config = config
        eng = engine_for(config)
        self.anchors = eng.anchors_np
        self.anchors_bv = eng.anchors_bv
        self.anchors_coors = eng.rects_np
        self.class_masks = eng.class_masks
        self.matched_threshold = np.concatenate(
            [np.full(e - s, table[n].get('matched_threshold', 0.6), np.float32) for n, (s, e) in self.class_masks.items()])
        self.unmatched_threshold = np.concatenate(
            [np.full(e - s, table[n].get('unmatched_threshold', 0.45), np.float32) for n, (s, e) in self.class_masks.items()])

    def create_mask_device(self, coors, num):
        """coors i32[>=P,3] cuda, num i32[1] cuda -> bool[A] cuda."""
        return engine_for(self._config).anchor_mask(coors, num).view(torch.bool)

    def create_mask(self, coors, grid_size=None, voxel_size=None, offset=None, gpu=True):
        """numpy coors in, numpy bool[A] out (the reference's signature; gpu flag is ignored --
        there is only the device path)."""
        eng = engine_for(self._config)
        if isinstance(coors, torch.Tensor):
            co = coors.to(eng.device, torch.int32).contiguous()
        else:
            co = torch.from_numpy(np.ascontiguousarray(coors, dtype=np.int32)).to(eng.device)
        if co.shape[0] > eng.max_voxels:
            raise ValueError("more pillars than max_voxels")
        mask = eng.anchor_mask(co, eng.num_tensor(co.shape[0])).view(torch.bool)
        return mask if isinstance(coors, torch.Tensor) else mask.cpu().numpy()
