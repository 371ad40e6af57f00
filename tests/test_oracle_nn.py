"""The reference's tests never exercise the network ("parity unpinned"), so
the oracle's forward is checked against the documented Flux/NNlib layer semantics written
independently with torch in float64: Conv = conv2d on spatially FLIPPED kernels, BatchNorm
in inference form, Dense = W*x + b, column softmax.  CPU only."""
import ctypes as C

import numpy as np
import pytest
import torch

import orc

L = orc.lib()


def get_param(net, layer, kind):
    n = L.or_net_param_count(net, layer, kind)
    out = np.zeros(n, np.float32)
    assert L.or_net_get(net, layer, kind, orc.fptr(out), n) == orc.OK
    return out


def randomize_bn(net, layers, rng):
    for l in layers:
        n = L.or_net_param_count(net, l, orc.K_BIAS)
        for kind, lo, hi in ((orc.K_BIAS, -0.2, 0.2), (orc.K_BN_BETA, -0.3, 0.3), (orc.K_BN_GAMMA, 0.5, 1.5),
                             (orc.K_BN_MEAN, -0.3, 0.3), (orc.K_BN_VAR, 0.5, 2.0)):
            v = rng.uniform(lo, hi, n).astype(np.float32)
            assert L.or_net_set(net, l, kind, orc.fptr(v), n) == orc.OK
    for l in (orc.L_VALUE_FC1, orc.L_VALUE_FC2, orc.L_POLICY_FC):
        n = L.or_net_param_count(net, l, orc.K_BIAS)
        v = rng.uniform(-0.2, 0.2, n).astype(np.float32)
        L.or_net_set(net, l, orc.K_BIAS, orc.fptr(v), n)


def torch_forward(net, N, tower, x_whcn):
    """x_whcn: (B, 17, N, N) indexed [b, c, j(col), i(row)] is awkward; build [b,c,i,j]."""
    P = N * N
    B = x_whcn.shape[0]
    dt = torch.float64
    x = torch.tensor(x_whcn, dtype=dt)   # [B, 17, i, j]

    def conv(l, k, cin, cout, inp):
        w = get_param(net, l, orc.K_WEIGHT).reshape(cout, cin, k, k)   # col-major [a,b,ci,o] -> [o,ci,b,a]
        w = torch.tensor(w, dtype=dt).permute(0, 1, 3, 2)               # [o, ci, a, b]
        w = torch.flip(w, dims=(2, 3))                                  # true convolution
        b = torch.tensor(get_param(net, l, orc.K_BIAS), dtype=dt)
        y = torch.nn.functional.conv2d(inp, w, b, padding=k // 2)
        g = torch.tensor(get_param(net, l, orc.K_BN_GAMMA), dtype=dt)
        be = torch.tensor(get_param(net, l, orc.K_BN_BETA), dtype=dt)
        mu = torch.tensor(get_param(net, l, orc.K_BN_MEAN), dtype=dt)
        var = torch.tensor(get_param(net, l, orc.K_BN_VAR), dtype=dt)
        eps = float(get_param(net, l, orc.K_BN_EPS)[0])
        return torch.nn.functional.batch_norm(y, mu, var, g, be, training=False, eps=eps)

    h = torch.relu(conv(0, 3, 17, 256, x))
    for blk in range(tower):
        t = torch.relu(conv(1 + 2 * blk, 3, 256, 256, h))
        h = torch.relu(conv(2 + 2 * blk, 3, 256, 256, t) + h)
    vh = torch.relu(conv(orc.L_VALUE_CONV, 1, 256, 1, h))     # [B,1,i,j]
    ph = torch.relu(conv(orc.L_POLICY_CONV, 1, 256, 2, h))    # [B,2,i,j]
    # Julia reshape of W x H x C x B (column-major) => index i + N*j + P*c
    vflat = vh.permute(0, 1, 3, 2).reshape(B, P)
    pflat = ph.permute(0, 1, 3, 2).reshape(B, 2 * P)
    w1 = torch.tensor(get_param(net, orc.L_VALUE_FC1, orc.K_WEIGHT).reshape(P, 256).T.copy(), dtype=dt)  # [out,in]
    b1 = torch.tensor(get_param(net, orc.L_VALUE_FC1, orc.K_BIAS), dtype=dt)
    w2 = torch.tensor(get_param(net, orc.L_VALUE_FC2, orc.K_WEIGHT).reshape(256, 1).T.copy(), dtype=dt)
    b2 = torch.tensor(get_param(net, orc.L_VALUE_FC2, orc.K_BIAS), dtype=dt)
    wp = torch.tensor(get_param(net, orc.L_POLICY_FC, orc.K_WEIGHT).reshape(2 * P, P + 1).T.copy(), dtype=dt)
    bp = torch.tensor(get_param(net, orc.L_POLICY_FC, orc.K_BIAS), dtype=dt)
    v = torch.tanh(torch.relu(vflat @ w1.T + b1) @ w2.T + b2)[:, 0]
    pi = torch.softmax(pflat @ wp.T + bp, dim=1)
    return pi.numpy(), v.numpy()


@pytest.mark.parametrize("N,tower", [(5, 1), (9, 2)])
def test_forward_matches_torch_fp64(N, tower):
    rng = np.random.RandomState(3)
    P, A = N * N, N * N + 1
    net = L.or_net_new(N, tower)
    L.or_net_init_synthetic(net, 5)
    randomize_bn(net, list(range(0, 1 + 2 * tower)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], rng)
    B = 3
    # feature-like input in WHCN order: x[i + N*(j + N*(c + 17*b))]
    x_bcij = rng.choice([-1.0, 0.0, 1.0], size=(B, 17, N, N))
    x_whcn = np.ascontiguousarray(x_bcij.transpose(0, 1, 3, 2)).astype(np.float32).reshape(-1)
    pi64 = np.zeros(B * A)
    v64 = np.zeros(B)
    xd = x_whcn.astype(np.float64)
    L.or_net_forward_feats_f64(net, xd.ctypes.data_as(C.POINTER(C.c_double)), B,
                               pi64.ctypes.data_as(C.POINTER(C.c_double)),
                               v64.ctypes.data_as(C.POINTER(C.c_double)))
    tpi, tv = torch_forward(net, N, tower, x_bcij)
    assert np.abs(pi64.reshape(B, A) - tpi).max() < 1e-10
    assert np.abs(v64 - tv).max() < 1e-10
    # fp32 oracle vs fp64 oracle: the 1e-4 bar of BASELINE.json north_star
    pi32 = np.zeros(B * A, np.float32)
    v32 = np.zeros(B, np.float32)
    L.or_net_forward_feats(net, orc.fptr(x_whcn), B, orc.fptr(pi32), orc.fptr(v32), 32)
    assert np.abs(pi32 - pi64).max() < 1e-5
    assert np.abs(v32 - v64).max() < 1e-5
    assert np.allclose(pi32.reshape(B, A).sum(1), 1, atol=1e-5)
    L.or_net_free(net)


def test_flip_matters():
    """A cross-correlation reading of the same weights must NOT agree -- guards the kernel
    flip that weight import has to perform."""
    N, tower = 5, 0
    rng = np.random.RandomState(0)
    net = L.or_net_new(N, tower)
    L.or_net_init_synthetic(net, 9)
    x_bcij = rng.choice([-1.0, 0.0, 1.0], size=(1, 17, N, N))
    x = np.ascontiguousarray(x_bcij.transpose(0, 1, 3, 2)).astype(np.float32).reshape(-1)
    pi = np.zeros(N * N + 1, np.float32)
    v = np.zeros(1, np.float32)
    L.or_net_forward_feats(net, orc.fptr(x), 1, orc.fptr(pi), orc.fptr(v), 32)
    # flip the stem kernel spatially and forward again
    w = get_param(net, 0, orc.K_WEIGHT).reshape(256, 17, 3, 3)
    wf = np.ascontiguousarray(w[:, :, ::-1, ::-1]).reshape(-1)
    L.or_net_set(net, 0, orc.K_WEIGHT, orc.fptr(wf), wf.size)
    pi2 = np.zeros_like(pi)
    L.or_net_forward_feats(net, orc.fptr(x), 1, orc.fptr(pi2), orc.fptr(v), 32)
    assert np.abs(pi - pi2).max() > 1e-6
    L.or_net_free(net)


def test_synthetic_init_statistics():
    net = L.or_net_new(9, 1)
    L.or_net_init_synthetic(net, 0)
    w = get_param(net, 1, orc.K_WEIGHT)
    limit = np.sqrt(6.0 / (2304 + 2304))
    assert abs(w.max() - limit) < 1e-3 * limit * 10 and abs(w.min() + limit) < 1e-2 * limit
    assert abs(w.mean()) < 1e-4
    assert abs(w.var() - limit ** 2 / 3) < 1e-2 * limit ** 2
    assert (get_param(net, 1, orc.K_BIAS) == 0).all()
    assert (get_param(net, 1, orc.K_BN_GAMMA) == 1).all()
    assert get_param(net, 1, orc.K_BN_EPS)[0] == np.float32(1e-5)
    L.or_net_free(net)
