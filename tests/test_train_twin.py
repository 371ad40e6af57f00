"""The float64 autograd twin that pins agz_train_step (tests/train_twin.py) is tied to the pinned oracle: in
inference mode its forward must equal the oracle's float64 forward on the same parameters.  CPU only."""
import ctypes as C

import numpy as np

import orc
from test_hostsim_go import random_positions
from test_oracle_nn import get_param, randomize_bn
from train_twin import Twin

L = orc.lib()


def test_twin_inference_forward_equals_the_oracle():
    N, tower, B = 5, 1, 4
    net = L.or_net_new(N, tower)
    L.or_net_init_synthetic(net, 5)
    randomize_bn(net, list(range(0, 1 + 2 * tower)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], np.random.RandomState(3))
    twin = Twin(N, tower, lambda l, k: get_param(net, l, k))
    feats = np.stack([orc.feats(p).reshape(-1) for p in random_positions(N, 2, 20, seed=1)[:B]]).astype(np.float64)
    pi64, v64 = np.zeros((B, N * N + 1)), np.zeros(B)
    L.or_net_forward_feats_f64(net, feats.ctypes.data_as(C.POINTER(C.c_double)), B, pi64.ctypes.data_as(C.POINTER(C.c_double)),
                               v64.ctypes.data_as(C.POINTER(C.c_double)))
    logp, v = twin.forward(feats, False)
    assert np.abs(np.exp(logp.detach().numpy()) - pi64).max() < 1e-12 and np.abs(v.detach().numpy() - v64).max() < 1e-12
    L.or_net_free(net)


def test_twin_step_decreases_the_loss_on_a_fixed_batch():
    N, tower, B = 5, 1, 6
    net = L.or_net_new(N, tower)
    L.or_net_init_synthetic(net, 2)
    twin = Twin(N, tower, lambda l, k: get_param(net, l, k))
    rng = np.random.RandomState(0)
    feats = np.stack([orc.feats(p).reshape(-1) for p in random_positions(N, 2, 20, seed=2)[:B]]).astype(np.float64)
    pi = rng.dirichlet(np.full(N * N + 1, 0.3), size=B)
    z = rng.choice([-1.0, 1.0], size=B)
    losses = [twin.step(feats, pi, z)[0] for _ in range(5)]
    assert losses[-1] < losses[0]
    L.or_net_free(net)
