"""Initial weights of the PointPillars network with the reference's state_dict key names / shapes
(networks/pointpillars8_shared.py:346-357): what constructing the reference's nn.Module
gives before load_state_dict -- uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) conv weights -- plus non-trivial BatchNorm
running statistics, so the PFN padded-slot term and the BN folding are exercised by untrained weights too."""
import numpy as np


def init_state_dict(seed=0, norm="instance", cls_bias=None, num_anchor_per_loc=9):
    """Random-init weights with the reference's state_dict key names/shapes
    (networks/pointpillars8_shared.py:346-357), with
    non-trivial BatchNorm running stats so the PFN padded-slot term and BN folding
    are exercised.  norm='batch' adds the per-norm-layer BN tensors of
    pointpillars8_export.py:65.  Returns dict[str, np.ndarray f32]."""
    rng = np.random.default_rng(seed)

    def conv_w(co, ci, k):
        bound = 1.0 / np.sqrt(ci * k * k)
        return rng.uniform(-bound, bound, (co, ci, k, k)).astype(np.float32)

    sd = {}
    b = 1.0 / np.sqrt(9.0)
    sd["pillar_point_net.pfn_layers.0.weight"] = rng.uniform(-b, b, (64, 9, 1)).astype(np.float32)
    sd["pillar_point_net.pfn_layers.1.weight"] = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    sd["pillar_point_net.pfn_layers.1.bias"] = rng.normal(0, 0.1, 64).astype(np.float32)
    sd["pillar_point_net.pfn_layers.1.running_mean"] = rng.normal(0, 0.1, 64).astype(np.float32)
    sd["pillar_point_net.pfn_layers.1.running_var"] = rng.uniform(0.5, 1.5, 64).astype(np.float32)

    def bn(prefix, c):
        if norm == "batch":
            sd[prefix + ".weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            sd[prefix + ".bias"] = rng.normal(0, 0.1, c).astype(np.float32)
            sd[prefix + ".running_mean"] = rng.normal(0, 0.1, c).astype(np.float32)
            sd[prefix + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    cin = 64
    for bi, (c, nres) in enumerate([(64, [1, 0]), (128, [1, 1, 0]), (256, [1, 1, 0])], start=1):
        sd[f"rpn.block{bi}.0.weight"] = conv_w(c, cin, 3)
        bn(f"rpn.block{bi}.1", c)
        for j, nl in enumerate(nres):
            base = f"rpn.block{bi}.{3 + j}.conv_block"
            bn(base + ".0", c)
            sd[base + ".2.weight"] = conv_w(c, c, 3)
            if nl == 1:
                bn(base + ".3", c)
                sd[base + ".5.weight"] = conv_w(c, c, 3)
        cin = c
    for di, (ci, co, k) in enumerate([(64, 64, 1), (128, 128, 2), (256, 128, 4)], start=1):
        bound = 1.0 / np.sqrt(co * k * k)
        sd[f"rpn.deconv{di}.0.weight"] = rng.uniform(-bound, bound, (ci, co, k, k)).astype(np.float32)
        bn(f"rpn.deconv{di}.1", co)
    bound = 1.0 / np.sqrt(320.0)
    na = int(num_anchor_per_loc)  # head rows: cls na | box 7 na | dir 2 na (reference: 9 anchors per location)
    for name, co in (("cls", na), ("box", 7 * na), ("dir", 2 * na)):
        sd[f"heads.conv_{name}.weight"] = rng.uniform(-bound, bound, (co, 320, 1, 1)).astype(np.float32)
        sd[f"heads.conv_{name}.bias"] = rng.uniform(-bound, bound, co).astype(np.float32)
    if cls_bias is not None:  # "trained-like": few anchors pass the 0.05 score threshold
        sd["heads.conv_cls.bias"][:] = cls_bias
    return sd
