"""3d_object_detection_amd -- MI355X-native PointPillars inference hot path.

Drop-in for the reference's call surface (train.py:192-196,224-230):

    import importlib; pp = importlib.import_module("3d_object_detection_amd"); pp.install()
    from framework.voxel_generator import VoxelGenerator      # now the HIP-backed classes
    from networks.pointpillars8_shared import PointPillars

install() aliases this package's framework/, networks/ and eval/ sub-packages under the reference's
top-level module names (eval.iou.rotate_iou_gpu_eval, eval.eval.get_official_eval_result, ...).  All compute runs in csrc/libpp_hip.so (hand-written HIP, gfx950).
"""
import importlib
import sys

__all__ = ["install"]


def install():
    """Expose framework.* / networks.* / eval.* under the reference's import paths."""
    pkg = __name__
    for top in ("framework", "networks", "eval"):
        mod = importlib.import_module(f"{pkg}.{top}")
        sys.modules[top] = mod
        for sub in getattr(mod, "__all__", []):
            sys.modules[f"{top}.{sub}"] = importlib.import_module(f"{pkg}.{top}.{sub}")
    return sys.modules["framework"], sys.modules["networks"]
