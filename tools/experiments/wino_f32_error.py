#!/usr/bin/env python3
"""Which Winograd tiles hold the 1e-4 bar in f32?  (HISTORY.md 10: the 9x9 layer is on the board's power limit, so only
fewer multiplies move it: F(5x5,3x3) over a 10-wide cover needs 196 per channel pair and board against F(3x3,3x3)'s 225,
an exact cover 9 = 5 + 4 with four tile shapes 169.)  CPU emulation in torch: the transforms B^T d B and A^T M A, the
plane products summed over the input channels and the BatchNorm / residual / ReLU all in float32, U = G k G^T formed in
float64 and rounded to float32 -- where agz_wino.hip / agz_wino4.hip round -- against the float64 network.  The sum over
channels is torch's float32 einsum (blocked pairwise-ish order), not the MFMA's running order: same error class.

  f33     F(3x3,3x3), points {0, +-1, 2, inf}              (agz_wino.hip; 9x9: 3x3 tiles, 225 multiplies)
  f43     F(4x4,3x3), points {0, +-1, +-2, inf}            (agz_wino4.hip; 9x9: 3x3 tiles of 4 over 12, 324)
  f53     F(5x5,3x3), points {0, +-1, +-2, 1/2, inf}       (9x9: 2x2 tiles over 10, 196)
  f53b    F(5x5,3x3), points {0, +-1, +-1/2, 2, inf}
  mix54   rows and columns split 5 + 4: tiles 5x5, 5x4, 4x5, 4x4 with F(5,3) / F(4,3) per axis (9x9: exact cover, 169)

usage: wino_f32_error.py [board 9] [tower 10] [positions 8]      Needs oracle/liboracle.so (tests/orc.py)."""
import os
import sys
from fractions import Fraction as Fr

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from test_oracle_nn import get_param, randomize_bn  # noqa: E402

L = orc.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 9
TOWER = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
f64, f32 = torch.float64, torch.float32


def toom_cook(m, pts):
    """1-D F(m, 3) with the finite points `pts` (m + 1 of them) and infinity: exact A^T [m x n], G [n x 3], B^T [n x n],
    n = m + 2, such that y = A^T ((G g) * (B^T d)) is the correlation y_k = sum_t d_{k+t} g_t."""
    r, n = 3, m + 2
    pts = [Fr(p) for p in pts]
    assert len(pts) == n - 1
    AT = [[(pts[j] ** i if j < n - 1 else Fr(int(i == m - 1))) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        f = Fr(1)
        for l in range(n - 1):
            if l != j:
                f *= pts[j] - pts[l]
        G.append([pts[j] ** k / f for k in range(r)])
    G.append([Fr(0), Fr(0), Fr(1)])
    # B^T from the bilinear identity  sum_j AT[k][j] G[j][t] BT[j][s] = [s == k + t]:  for every s an (m r) x n system
    BT = [[Fr(0)] * n for _ in range(n)]
    for s in range(n):
        rows = [[AT[k][j] * G[j][t] for j in range(n)] + [Fr(int(s == k + t))] for k in range(m) for t in range(r)]
        # Gaussian elimination over the rationals (consistent, rank n)
        piv = []
        rr = 0
        for c in range(n):
            p = next((i for i in range(rr, len(rows)) if rows[i][c] != 0), None)
            if p is None:
                continue
            rows[rr], rows[p] = rows[p], rows[rr]
            inv = 1 / rows[rr][c]
            rows[rr] = [v * inv for v in rows[rr]]
            for i in range(len(rows)):
                if i != rr and rows[i][c] != 0:
                    fac = rows[i][c]
                    rows[i] = [a - fac * b for a, b in zip(rows[i], rows[rr])]
            piv.append(c)
            rr += 1
        assert len(piv) == n and all(all(v == 0 for v in row) for row in rows[n:]), "points do not give a valid algorithm"
        for i, c in enumerate(piv):
            BT[c][s] = rows[i][n]
    t = lambda M_: torch.tensor([[float(v) for v in row] for row in M_], dtype=f64)
    return t(AT), t(G), t(BT)


ALGS = {      # outputs per axis, finite interpolation points (infinity is added)
    "f33": (3, [0, 1, -1, 2]),                    # agz_wino.hip
    "f43": (4, [0, 1, -1, 2, -2]),                # agz_wino4.hip
    "f53": (5, [0, 1, -1, 2, -2, Fr(1, 2)]),
    "f53b": (5, [0, 1, -1, Fr(1, 2), Fr(-1, 2), 2]),
}


def params(net, l, cin, cout, k):
    w = torch.tensor(get_param(net, l, orc.K_WEIGHT).reshape(cout, cin, k, k), dtype=f64).permute(0, 1, 3, 2)
    w = torch.flip(w, dims=(2, 3))
    g = torch.tensor(get_param(net, l, orc.K_BN_GAMMA), dtype=f64)
    be = torch.tensor(get_param(net, l, orc.K_BN_BETA), dtype=f64)
    mu = torch.tensor(get_param(net, l, orc.K_BN_MEAN), dtype=f64)
    var = torch.tensor(get_param(net, l, orc.K_BN_VAR), dtype=f64)
    eps = float(get_param(net, l, orc.K_BN_EPS)[0])
    b = torch.tensor(get_param(net, l, orc.K_BIAS), dtype=f64)
    sc = g / torch.sqrt(var + eps)
    return w, sc, (b - mu) * sc + be


def conv_direct(x, w):
    return torch.nn.functional.conv2d(x, w, None, padding=1)


def axis_plan(mode):
    """segments of one board axis: (start, outputs m, algorithm)"""
    if mode == "mix54":
        assert N == 9, "the 5 + 4 split is the 9x9 cover"
        return [(0, 5, "f53"), (5, 4, "f43")]
    m = ALGS[mode][0]
    return [(s, m, mode) for s in range(0, N, m)]


MATS = {}


def conv_wino(x, w, mode, dt):
    """x [B, C, N, N] in dt, w [O, C, 3, 3] float64 correlation kernel; every product of the algorithm in dt"""
    Bn, C = x.shape[:2]
    O = w.shape[0]
    plan = axis_plan(mode)
    y = torch.zeros(Bn, O, N, N, dtype=dt)
    for (r0, mr, ar) in plan:
        ATr, Gr, BTr = MATS[ar]
        for (c0, mc, ac) in plan:
            ATc, Gc, BTc = MATS[ac]
            nr, nc = mr + 2, mc + 2
            # input patch rows r0 - 1 .. r0 + mr, zero outside the board
            d = torch.zeros(Bn, C, nr, nc, dtype=dt)
            rs, re = max(r0 - 1, 0), min(r0 + mr + 1, N)
            cs, ce = max(c0 - 1, 0), min(c0 + mc + 1, N)
            d[:, :, rs - (r0 - 1):re - (r0 - 1), cs - (c0 - 1):ce - (c0 - 1)] = x[:, :, rs:re, cs:ce]
            V = torch.einsum("iu,bcuv,jv->bcij", BTr.to(dt), d, BTc.to(dt))
            U = torch.einsum("ia,ocab,jb->ocij", Gr, w, Gc).to(dt)            # float64, rounded once
            M = torch.einsum("ocij,bcij->boij", U, V)
            Y = torch.einsum("pi,boij,qj->bopq", ATr.to(dt), M, ATc.to(dt))
            y[:, :, r0:min(r0 + mr, N), c0:min(c0 + mc, N)] = Y[:, :, :min(mr, N - r0), :min(mc, N - c0)]
    return y


def forward(net, x, mode):
    """mode "exact": float64 direct; otherwise the tower's convolutions by `mode` in float32 (stem and heads direct f32)"""
    dt = f64 if mode == "exact" else f32
    P = N * N
    cast = lambda *ts: [t.to(dt) for t in ts]
    w, sc, sh = params(net, 0, 17, 256, 3)
    w, sc, sh = cast(w, sc, sh)
    a = torch.relu(conv_direct(x.to(dt), w) * sc[None, :, None, None] + sh[None, :, None, None])
    for blk in range(TOWER):
        w1, sc1, sh1 = params(net, 1 + 2 * blk, 256, 256, 3)
        w2, sc2, sh2 = params(net, 2 + 2 * blk, 256, 256, 3)
        sc1, sh1, sc2, sh2 = cast(sc1, sh1, sc2, sh2)
        cv = (lambda t, ww: conv_direct(t, ww)) if mode == "exact" else (lambda t, ww: conv_wino(t, ww, mode, dt))
        t = torch.relu(cv(a, w1) * sc1[None, :, None, None] + sh1[None, :, None, None])
        a = torch.relu(cv(t, w2) * sc2[None, :, None, None] + sh2[None, :, None, None] + a)
    wv, scv, shv = cast(*params(net, orc.L_VALUE_CONV, 256, 1, 1))
    wp_, scp, shp = cast(*params(net, orc.L_POLICY_CONV, 256, 2, 1))
    vh = torch.relu(torch.nn.functional.conv2d(a, wv) * scv[None, :, None, None] + shv[None, :, None, None])
    ph = torch.relu(torch.nn.functional.conv2d(a, wp_) * scp[None, :, None, None] + shp[None, :, None, None])
    Bn = x.shape[0]
    vflat = vh.permute(0, 1, 3, 2).reshape(Bn, P)
    pflat = ph.permute(0, 1, 3, 2).reshape(Bn, 2 * P)
    g = lambda l, k, shape: torch.tensor(get_param(net, l, k).reshape(*shape).T.copy(), dtype=dt)
    w1 = g(orc.L_VALUE_FC1, orc.K_WEIGHT, (P, 256))
    b1 = torch.tensor(get_param(net, orc.L_VALUE_FC1, orc.K_BIAS), dtype=dt)
    w2 = g(orc.L_VALUE_FC2, orc.K_WEIGHT, (256, 1))
    b2 = torch.tensor(get_param(net, orc.L_VALUE_FC2, orc.K_BIAS), dtype=dt)
    wp = g(orc.L_POLICY_FC, orc.K_WEIGHT, (2 * P, P + 1))
    bp = torch.tensor(get_param(net, orc.L_POLICY_FC, orc.K_BIAS), dtype=dt)
    v = torch.tanh(torch.relu(vflat @ w1.T + b1) @ w2.T + b2)[:, 0]
    pi = torch.softmax(pflat @ wp.T + bp, dim=1)
    return pi.to(f64).numpy(), v.to(f64).numpy()


def multiplies(mode):
    plan = axis_plan(mode)
    return sum((mr + 2) * (mc + 2) for (_, mr, _) in plan for (_, mc, _) in plan)


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for name, (m, pts) in ALGS.items():
        MATS[name] = toom_cook(m, pts)
    rng = np.random.RandomState(42)
    net = L.or_net_new(N, TOWER)
    L.or_net_init_synthetic(net, 3)
    randomize_bn(net, list(range(0, 1 + 2 * TOWER)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], rng)
    x = (rng.rand(B, 17, N, N) < 0.25).astype(np.float64)
    x[:, 16] = np.where(rng.rand(B, 1, 1) < 0.5, 1.0, -1.0)
    x = torch.tensor(x, dtype=f64)
    modes = ["f33", "f43", "f53", "f53b"] + (["mix54"] if N == 9 else [])
    with torch.no_grad():
        w = params(net, 1, 256, 256, 3)[0]
        a = torch.tensor(rng.randn(1, 256, N, N), dtype=f64)
        ref = conv_direct(a, w)
        for mode in modes:
            e64 = (conv_wino(a, w, mode, f64) - ref).abs().max().item()
            e32 = (conv_wino(a.to(f32), w, mode, f32).to(f64) - ref).abs().max().item()
            print(f"{mode:6s} one layer on N(0,1) input: float64 algebra {e64:.1e}, float32 {e32:.2e} (|y| max {ref.abs().max():.1f}); "
                  f"multiplies per channel pair and board {multiplies(mode)}")
        pi0, v0 = forward(net, x, "exact")
        print(f"{N}x{N}, tower {TOWER}, {B} positions; pi max {pi0.max():.3e}, |v| max {np.abs(v0).max():.3f}")
        for mode in modes:
            pi, v = forward(net, x, mode)
            print(f"{mode:6s} f32 vs exact f64: max |d pi| {np.abs(pi - pi0).max():.2e}   max |d v| {np.abs(v - v0).max():.2e}")
    L.or_net_free(net)


if __name__ == "__main__":
    main()
