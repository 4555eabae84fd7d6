"""HIP-backed mirrors of the reference's networks/pointpillars8_* modules."""
__all__ = ["pointpillars8_shared", "pointpillars8_export", "pointpillars8_trt"]
