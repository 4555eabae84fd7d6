"""networks.pointpillars8_trt.PointPillars (reference pointpillars8_trt.py:370-446): the TensorRT-engine
runner is replaced by the same HIP kernels as pointpillars8_export (BatchNorm backbone, fp32)."""
from .pointpillars8_export import PointPillars  # noqa: F401
