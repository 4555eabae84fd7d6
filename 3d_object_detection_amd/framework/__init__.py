"""HIP-backed mirrors of the reference's framework/ modules on the inference hot path."""
__all__ = ["voxel_generator", "anchor_assigner", "dataset", "utils", "inference", "nms", "box_torch_ops"]
