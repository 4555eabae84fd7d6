"""framework.voxel_generator.VoxelGenerator (reference voxel_generator.py:5-40) on the HIP voxeliser."""
import numpy as np
import torch

from ..engine import engine_for, snap_geometry


class VoxelGenerator:
    def __init__(self, config):
        vs, offset, grid, range_diff, det_range = snap_geometry(config)
        self.voxel_size = vs
        self.detection_range = det_range
        self.offset = offset
        self.grid_size = grid
        self.max_num_points = config['max_num_points']
        self.max_voxels = config['max_voxels']
        # the reference writes these four keys back for AnchorAssigner / PointPillars (:23-26)
        config['detection_range'] = det_range
        config['detection_offset'] = offset
        config['detection_range_diff'] = range_diff
        config['grid_size'] = grid
        self._config = config

    def __getstate__(self):  # picklable for DataLoader workers: the GPU engine is re-created lazily
        d = dict(self.__dict__)
        d['_config'] = {k: v for k, v in self._config.items() if not k.startswith('_pp_')}
        return d

    def generate_device(self, points, sync=True):
        """points: f32[N,F] cuda tensor.  Returns device tensors; with sync=True sliced to the pillar
        count (one 4-byte D2H), else the full-capacity buffers plus the device count."""
        eng = engine_for(self._config)
        voxels, coors, npts, num = eng.voxelize(points.contiguous())
        if not sync:
            return voxels, coors, npts, num
        p = int(num.item())
        return voxels[:p], coors[:p], npts[:p]

    def generate(self, points):
        """numpy in, numpy out -- same contract as the reference (caller owns the outputs)."""
        eng = engine_for(self._config)
        pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(eng.device)
        if pts.dim() != 2:
            pts = pts.reshape(-1, eng.F)
        v, c, n = self.generate_device(pts)
        return v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy()


class VoxelGenerator_trt(VoxelGenerator):
    """voxel_generator.py:43-79: same, also returns voxel_num."""

    def generate(self, points):
        v, c, n = super().generate(points)
        return v, c, n, v.shape[0]
