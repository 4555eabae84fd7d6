"""networks.pointpillars8_export.PointPillars (reference pointpillars8_export.py:163-226): same network
with the BatchNorm2d(eps 1e-3) backbone, folded into the conv prologues.  `.export` (ONNX/TensorRT
tooling, :228-278) has no ROCm counterpart and is out of scope."""
from .pointpillars8_shared import PointPillars as _Base


class PointPillars(_Base):
    _norm = "batch"
