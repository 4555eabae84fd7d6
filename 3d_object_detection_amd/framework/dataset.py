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

    def get(self, points, toTorch=True):
        eng = engine_for(self._config)
        start = time.time()
        if isinstance(points, torch.Tensor):
            pts = points.to(eng.device, torch.float32).contiguous()
        else:
            pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(eng.device)
        voxels, coors, npts, num = eng.voxelize(pts)
        mask = eng.anchor_mask(coors, num).view(torch.bool)
        p = int(num.item())  # the one sync: the reference's example carries exact-length tensors
        voxel_time = time.time()
        example = {'voxels': voxels[:p], 'coordinates': coors[:p], 'num_points_per_voxel': npts[:p],
                   'anchors_mask': mask[None, :]}
        self.voxel_time += voxel_time - start
        if not toTorch:
            example = {k: v.cpu().numpy() for k, v in example.items()}
        return example
