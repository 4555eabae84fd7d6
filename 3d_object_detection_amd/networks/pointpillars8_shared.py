"""networks.pointpillars8_shared.PointPillars (reference pointpillars8_shared.py:346-382): the eager
network with the InstanceNorm backbone, running PFN / scatter / RPN / head as HIP kernels."""
import time

import numpy as np
import torch

from ..engine import engine_for
from .init import init_state_dict


class PointPillars:
    _norm = "instance"

    def __init__(self, config):
        self.device = config['device']
        self._config = config
        config['_pp_norm'] = self._norm
        self._eng = engine_for(config, self._norm)
        self._sd = None
        self.profile_stages = True  # the reference synchronises after every stage (:365-374)
        self.pfn_time, self.rpn_time, self.scatter_time, self.heads_time = 0.0, 0.0, 0.0, 0.0
        # like nn.Module construction, start from random initial weights
        self.load_state_dict(init_state_dict(0, norm=self._norm))

    # nn.Module-style surface used by train.py:196-205
    def to(self, device):
        return self

    def eval(self):
        return self

    def half(self):
        """The reference deploys FP16 TensorRT engines (framework/trt_utils.py:30, networks/pointpillars8_trt.py:208-223,295-314).
        Here: fp16 MFMA operands for every convolution, upsampler and the head, fp32 accumulation, activations still fp32 in
        HBM; tolerance table in DESIGN.md.  The parity contract (<= 1e-3 vs the fp32 reference) is stated for float() / the
        default (and is also met by precision("bf16x3"))."""
        self._eng.set_precision("fp16")
        return self

    def float(self):
        self._eng.set_precision("fp32")
        return self

    def precision(self, mode):
        """ "fp32" | "bf16x3" (split-bf16: fp32-equivalent on the bf16 MFMAs) | "fp16" | "bf16" | "fp16s" (fp16 operands and fp16 tensors)."""
        self._eng.set_precision(mode)
        return self

    def state_dict(self):
        return dict(self._sd)

    def load_state_dict(self, sd, strict=True):
        self._sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in sd.items()}
        self._eng.load_state_dict(self._sd)
        return self

    def _sync(self):
        if self.profile_stages:
            torch.cuda.synchronize()
        return time.time()

    def forward(self, example):
        eng = self._eng
        voxels = example["voxels"].contiguous()
        npts = example["num_points_per_voxel"].contiguous()
        coors = example["coordinates"].contiguous()
        num = eng.num_tensor(voxels.shape[0])
        start = time.time()
        feat = eng.pfn(voxels, coors, npts, num)
        pfn_time = self._sync()
        canvas = eng.scatter(feat, coors, num)
        scatter_time = self._sync()
        rpn = eng.backbone(canvas)
        rpn_time = self._sync()
        cls, box, dr = eng.head(rpn)
        heads_time = self._sync()
        self.pfn_time += pfn_time - start
        self.scatter_time += scatter_time - pfn_time
        self.rpn_time += rpn_time - scatter_time
        self.heads_time += heads_time - rpn_time
        return {"cls_preds": cls, "box_preds": box, "dir_preds": dr}

    __call__ = forward

    # sub-stages, named as the reference's sub-modules, for stage-wise parity tests
    def pillar_point_net(self, voxels, num_point_per_voxel, coors):
        return self._eng.pfn(voxels.contiguous(), coors.contiguous(), num_point_per_voxel.contiguous(),
                             self._eng.num_tensor(voxels.shape[0]))[:voxels.shape[0]]

    def middle_feature_extractor(self, voxel_features, coords):
        return self._eng.scatter(voxel_features.contiguous(), coords.contiguous(), self._eng.num_tensor(coords.shape[0]))

    def rpn(self, x):
        return self._eng.backbone(x.contiguous())

    def heads(self, x):
        cls, box, dr = self._eng.head(x.contiguous())
        return {"cls_preds": cls, "box_preds": box, "dir_preds": dr}
