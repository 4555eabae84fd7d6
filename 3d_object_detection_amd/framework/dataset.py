"""framework.dataset.InferData (reference dataset.py:199-231): voxelise + anchor mask -> example dict.
Everything stays on the device; the only host traffic is the cloud upload and a 4-byte pillar count."""
import time

import numpy as np
import torch

from ..engine import engine_for


class InferData:
    def __init__(self, config, voxel_generator, anchor_assigner, dtype=torch.float32):
        self.voxel_generator = voxel_generator
        self.anchor_assigner = anchor_assigner
        self.grid_size = config['grid_size']
        self.create_mask_gpu = config.get('create_mask_gpu', 1) == 1
        self.dtype = dtype
        self.voxel_time = 0.0
        self.mask_time = 0.0
        self.convert_time = 0.0
        self.device = config['device']
        self._config = config
        self.profile_stages = True  # voxel_time / mask_time / convert_time are per-stage wall times like the reference's (:208-229)

    def get(self, points, toTorch=True):
        eng = engine_for(self._config)
        sync = torch.cuda.synchronize if self.profile_stages else (lambda: None)
        start = time.time()
        # the reference converts to device tensors LAST (example_convert_to_torch, utils.py:7-20); here the cloud goes up
        # first and everything after it stays on the device -- its upload is what convert_time measures
        if isinstance(points, torch.Tensor):
            pts = points.to(eng.device, torch.float32).contiguous()
        else:
            pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(eng.device)
        sync()
        convert_time = time.time()
        voxels, coors, npts, num = eng.voxelize(pts)
        p = int(num.item())  # the one unavoidable sync: the reference's example carries exact-length tensors
        voxel_time = time.time()
        mask = eng.anchor_mask(coors, num).view(torch.bool)
        sync()
        mask_time = time.time()
        example = {'voxels': voxels[:p], 'coordinates': coors[:p], 'num_points_per_voxel': npts[:p],
                   'anchors_mask': mask[None, :]}
        self.convert_time += convert_time - start
        self.voxel_time += voxel_time - convert_time
        self.mask_time += mask_time - voxel_time
        if not toTorch:
            example = {k: v.cpu().numpy() for k, v in example.items()}
        return example
