#!/usr/bin/env python3
"""VERDICT r3 #2 asked for the error of an fp16-operand Winograd F(2x2,3x3) tower at BASELINE configs[4]'s depth (19x19,
tower 20) against float64 before anybody builds the kernel.  CPU emulation in torch float64 with explicit half roundings
at the points where such a kernel would round:

  direct   today's fp16 tower (agz_conv16.hip): weights and stored activations are IEEE half, products summed exactly
           (f32 accumulation error is not modelled: it is below the roundings studied here)
  f23      F(2x2,3x3): activations stored half; V = B^T d B formed in f32/f64 from them and ROUNDED TO HALF (the MFMA
           operand); U = G k G^T formed in float64 and rounded to half; M = sum_cin U.V exact; Y = A^T M A; BatchNorm,
           residual, ReLU as in the direct form; result stored half
  f23s     the same with U and V pre-scaled per plane by powers of two so that their entries use the half range
  f43(s)   F(4x4,3x3) with half operands the same way (the transform agz_wino4.hip runs in f32): 3.6x fewer multiplies

Network: the synthetic network of tests/test_gpu_configs.py (glorot weights, BatchNorm statistics randomised), B positions
of random stones.  Prints max |d pi|, |d v| against the exact float64 network.  Needs oracle/liboracle.so (tests/orc.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from test_oracle_nn import get_param, randomize_bn  # noqa: E402

L = orc.lib()
N, TOWER, B = 19, int(os.environ.get("TOWER", "20")), 4
dt = torch.float64
h16 = lambda t: t.to(torch.float16).to(dt)      # round to IEEE half, keep working in float64

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=dt)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=dt)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=dt)
# F(4x4,3x3), points {0, +-1, +-2, inf} (agz_wino4.hip)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=dt)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1]], dtype=dt)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=dt)


def params(net, l, cin, cout, k):
    w = torch.tensor(get_param(net, l, orc.K_WEIGHT).reshape(cout, cin, k, k), dtype=dt).permute(0, 1, 3, 2)
    w = torch.flip(w, dims=(2, 3))
    g = torch.tensor(get_param(net, l, orc.K_BN_GAMMA), dtype=dt)
    be = torch.tensor(get_param(net, l, orc.K_BN_BETA), dtype=dt)
    mu = torch.tensor(get_param(net, l, orc.K_BN_MEAN), dtype=dt)
    var = torch.tensor(get_param(net, l, orc.K_BN_VAR), dtype=dt)
    eps = float(get_param(net, l, orc.K_BN_EPS)[0])
    b = torch.tensor(get_param(net, l, orc.K_BIAS), dtype=dt)
    sc = g / torch.sqrt(var + eps)
    return w, sc, (b - mu) * sc + be


def conv_direct(x, w):
    return torch.nn.functional.conv2d(x, w, None, padding=1)


def conv_f23(x, w, mode):
    """x [B, C, 19, 19] (already half-rounded where the mode says so), w [O, C, 3, 3] correlation kernel"""
    Bn, C, H, W = x.shape
    big = mode.startswith("f43")
    m, a_ = (4, 6) if big else (2, 4)
    bt, g, at = (BT4, G4, AT4) if big else (BT, G, AT)
    T = (H + m - 1) // m
    xp = torch.nn.functional.pad(x, (1, m * T - W + 1, 1, m * T - H + 1))
    d = xp.unfold(2, a_, m).unfold(3, a_, m)                    # [B, C, T, T, a, a]
    V = torch.einsum("iu,bctsuv,jv->bctsij", bt, d, bt)
    U = torch.einsum("ia,ocab,jb->ocij", g, w, g)               # [O, C, a, a]
    if mode.endswith("s"):
        s = U.abs().amax(dim=(0, 1), keepdim=True)               # per-plane scale to use the half range
        s = 2.0 ** torch.floor(torch.log2(1.0 / s))
    else:
        s = torch.ones(1, 1, a_, a_, dtype=dt)
    if mode.endswith("s"):
        # ... and V per plane too: a power of two that brings the plane's largest entry of this batch near 1 (a kernel
        # would use a fixed per-plane constant; the transform's row sums bound it)
        sv = 2.0 ** torch.floor(torch.log2(1.0 / V.abs().amax(dim=(0, 1, 2, 3), keepdim=True).clamp_min(1e-30)))
    else:
        sv = torch.ones(1, 1, 1, 1, a_, a_, dtype=dt)
    Uh, Vh = h16(U * s), h16(V * sv)
    M = torch.einsum("ocij,bctsij->botsij", Uh, Vh) / (s.reshape(1, 1, 1, 1, a_, a_) * sv)
    Y = torch.einsum("pi,botsij,qj->botspq", at, M, at)          # [B, O, T, T, m, m]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, -1, m * T, m * T)
    return y[:, :, :H, :W]


def forward(net, x, mode):
    """mode: exact | direct | f23 | f23s"""
    P = N * N
    half = mode != "exact"
    w, sc, sh = params(net, 0, 17, 256, 3)
    h = torch.relu(conv_direct(x, w) * sc[None, :, None, None] + sh[None, :, None, None])       # the stem is f32 in every mode
    res = h                                                        # block 0's residual is the unrounded stem output
    a = h16(h) if half else h
    for blk in range(TOWER):
        w1, sc1, sh1 = params(net, 1 + 2 * blk, 256, 256, 3)
        w2, sc2, sh2 = params(net, 2 + 2 * blk, 256, 256, 3)
        if mode in ("f23", "f23s", "f43", "f43s"):
            c1 = conv_f23(a, w1, mode)
        else:
            c1 = conv_direct(a, h16(w1) if half else w1)
        t = torch.relu(c1 * sc1[None, :, None, None] + sh1[None, :, None, None])
        t = h16(t) if half else t
        if mode in ("f23", "f23s", "f43", "f43s"):
            c2 = conv_f23(t, w2, mode)
        else:
            c2 = conv_direct(t, h16(w2) if half else w2)
        o = torch.relu(c2 * sc2[None, :, None, None] + sh2[None, :, None, None] + res)
        last = blk + 1 == TOWER
        a = o if (last or not half) else h16(o)                   # the last block's output stays f32 for the heads
        res = a
    hfin = a
    wv, scv, shv = params(net, orc.L_VALUE_CONV, 256, 1, 1)
    wp_, scp, shp = params(net, orc.L_POLICY_CONV, 256, 2, 1)
    vh = torch.relu(torch.nn.functional.conv2d(hfin, wv) * scv[None, :, None, None] + shv[None, :, None, None])
    ph = torch.relu(torch.nn.functional.conv2d(hfin, wp_) * scp[None, :, None, None] + shp[None, :, None, None])
    Bn = x.shape[0]
    vflat = vh.permute(0, 1, 3, 2).reshape(Bn, P)
    pflat = ph.permute(0, 1, 3, 2).reshape(Bn, 2 * P)
    w1 = torch.tensor(get_param(net, orc.L_VALUE_FC1, orc.K_WEIGHT).reshape(P, 256).T.copy(), dtype=dt)
    b1 = torch.tensor(get_param(net, orc.L_VALUE_FC1, orc.K_BIAS), dtype=dt)
    w2 = torch.tensor(get_param(net, orc.L_VALUE_FC2, orc.K_WEIGHT).reshape(256, 1).T.copy(), dtype=dt)
    b2 = torch.tensor(get_param(net, orc.L_VALUE_FC2, orc.K_BIAS), dtype=dt)
    wp = torch.tensor(get_param(net, orc.L_POLICY_FC, orc.K_WEIGHT).reshape(2 * P, P + 1).T.copy(), dtype=dt)
    bp = torch.tensor(get_param(net, orc.L_POLICY_FC, orc.K_BIAS), dtype=dt)
    v = torch.tanh(torch.relu(vflat @ w1.T + b1) @ w2.T + b2)[:, 0]
    pi = torch.softmax(pflat @ wp.T + bp, dim=1)
    return pi.numpy(), v.numpy()


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rng = np.random.RandomState(42)
    net = L.or_net_new(N, TOWER)
    L.or_net_init_synthetic(net, 3)
    randomize_bn(net, list(range(0, 1 + 2 * TOWER)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], rng)
    x = (rng.rand(B, 17, N, N) < 0.25).astype(np.float64)
    x[:, 16] = np.where(rng.rand(B, 1, 1) < 0.5, 1.0, -1.0)
    x = torch.tensor(x, dtype=dt)
    with torch.no_grad():
        pi0, v0 = forward(net, x, "exact")
        # sanity: the F(2x2,3x3) algebra itself (no half rounding) reproduces the direct convolution
        w = params(net, 1, 256, 256, 3)[0]
        a = torch.tensor(rng.randn(1, 256, N, N), dtype=dt)
        U_ = conv_direct(a, w)
        global h16
        keep = h16
        h16 = lambda t: t
        alg = (conv_f23(a, w, "f23") - U_).abs().max().item()
        h16 = keep
        print(f"F(2x2,3x3) identity without roundings: max |diff| {alg:.2e}")
        print(f"19x19, tower {TOWER}, {B} positions; pi max {pi0.max():.3e}, |v| max {np.abs(v0).max():.3f}")
        for mode in ("direct", "f23", "f23s", "f43", "f43s"):
            pi, v = forward(net, x, mode)
            print(f"{mode:7s} vs exact f64: max |d pi| {np.abs(pi - pi0).max():.2e}   max |d v| {np.abs(v - v0).max():.2e}")
    L.or_net_free(net)


if __name__ == "__main__":
    main()
