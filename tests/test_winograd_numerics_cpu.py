"""CPU: how far fp32 Winograd moves the logits of the whole backbone + head -- the numerical side of DESIGN.md's choice of
F(2x2,3x3) for the stride-1 3x3 convolutions and of its verdict on F(4x4,3x3) (VERDICT r1 task 4: "reject it if any golden
moves past 1e-3": it does not -- F(4x4) is turned down for its register footprint, not for its arithmetic).  The oracle's backbone runs three times on the reference-generated small canvas: direct convolution (the
reference's arithmetic), and with every stride-1 3x3 convolution replaced by an fp32 emulation of Winograd F(2x2) / F(4x4)
(weights transformed in fp64 and rounded once, input / output transforms and the per-position contraction in fp32, as the HIP
kernels do)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib

from oracle import pp_oracle as O

synth = importlib.import_module("3d_object_detection_amd.synth")

MATS = {
    2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
        np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)),
    4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64),
        np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)),
}


def wino_conv(x, w, m):
    """3x3, stride 1, padding 1 convolution of x [1,C,H,W] with w [K,C,3,3] by Winograd F(m x m, 3x3) in fp32."""
    Bt, G, At = MATS[m]
    a = m + 2
    U = torch.from_numpy(np.einsum("ai,kcij,bj->kcab", G, w.double().numpy(), G)).float()      # fp64 on the host, rounded once
    Btf, Atf = torch.from_numpy(Bt).float(), torch.from_numpy(At).float()
    _, C, H, W = x.shape
    nh, nw = -(-H // m), -(-W // m)
    xp = Fn.pad(x, (1, nw * m - W + 1, 1, nh * m - H + 1))
    t = xp.unfold(2, a, m).unfold(3, a, m)[0]                                                   # [C, nh, nw, a, a]
    V = torch.einsum("ai,cnwij,bj->cnwab", Btf, t, Btf)
    M = torch.einsum("kcab,cnwab->knwab", U, V)
    Y = torch.einsum("ia,knwab,jb->knwij", Atf, M, Atf)                                         # [K, nh, nw, m, m]
    return Y.permute(0, 1, 3, 2, 4).reshape(1, -1, nh * m, nw * m)[:, :, :H, :W].contiguous()


def backbone_with(conv, x, sd, monkeypatch):
    real = Fn.conv2d

    def patched(inp, weight, bias=None, stride=1, padding=0, *args, **kw):
        if conv is not None and tuple(weight.shape[2:]) == (3, 3) and stride == 1 and padding == 1 and bias is None:
            return conv(inp, weight)
        return real(inp, weight, bias, stride, padding, *args, **kw)

    monkeypatch.setattr(Fn, "conv2d", patched)
    try:
        rpn = O.backbone(x, sd)
    finally:
        monkeypatch.setattr(Fn, "conv2d", real)
    return rpn, O.head(rpn, sd)


def test_winograd_logit_deviation_f2_and_f4(monkeypatch, capsys):
    g = np.load(os.path.join(ROOT, "tests", "golden", "backbone_small_instance.npz"))
    sd = synth.seeded_state_dict(0)
    x = g["x"]
    with torch.no_grad():
        ref_rpn, ref_head = backbone_with(None, x, sd, monkeypatch)
        np.testing.assert_allclose(ref_rpn, g["y"], rtol=0, atol=2e-5)                          # the oracle against the reference's own output
        dev = {}
        for m in (2, 4):
            rpn, head = backbone_with(lambda i, w, m=m: wino_conv(i, w, m), x, sd, monkeypatch)
            dev[m] = (float(np.abs(rpn - ref_rpn).max()), max(float(np.abs(a - b).max()) for a, b in zip(head, ref_head)))
    with capsys.disabled():
        print(f"\n[winograd numerics] 13 stacked stride-1 3x3 convs, fp32: F(2x2) moves the backbone output by {dev[2][0]:.2e} and the logits by "
              f"{dev[2][1]:.2e}; F(4x4) by {dev[4][0]:.2e} / {dev[4][1]:.2e}")
    assert dev[2][1] <= 2e-5                  # F(2x2): two orders of magnitude inside the 1e-3 parity bar
    assert dev[2][1] < dev[4][1] <= 1e-4      # F(4x4): several times more (the transforms' dynamic range), still far inside the bar
