"""framework.utils (reference utils.py:7-20): host->device conversion of an example dict."""
import torch


def example_convert_to_torch(example, dtype=torch.float32, device=torch.device("cuda:0")):
    out = {}
    for k, v in example.items():
        if k in ["voxels"]:
            out[k] = torch.as_tensor(v, dtype=dtype, device=device)
        elif k in ["coordinates", "num_points_per_voxel", "voxel_num"]:
            out[k] = torch.as_tensor(v, dtype=torch.int32, device=device)
        elif k in ["anchors_mask"]:
            out[k] = torch.as_tensor(v, dtype=torch.bool, device=device)
        else:
            out[k] = v
    return out
