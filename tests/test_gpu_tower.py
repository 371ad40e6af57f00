"""The f32 Winograd tower as ONE persistent launch (k_wino_tower: opt-in, agz_net_set_tower_persistent) against the same tower
as one launch per layer: the same device function runs each (layer, tile block, cout block) either way, so the
network's outputs must agree BIT FOR BIT -- any difference is a scheduling / visibility bug of the persistent kernel
(a tile block read before its producer's stores arrived), not rounding.  Repeated, on warm caches, at batch sizes
from one tile block to many rounds of all 64 quads, and against the float64 oracle."""
import ctypes as C

import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from gpu_common import copy_weights_from_oracle
from test_oracle_nn import randomize_bn

pytestmark = pytest.mark.gpu
L = orc.lib()
TOL = 1e-4


def _feats(rng, B, N):
    """Synthetic but legal-looking feature planes: 16 binary stone planes and a +-1 colour plane."""
    P = N * N
    f = (rng.rand(B, 17, P) < 0.3).astype(np.float32)
    f[:, 16, :] = np.where(rng.rand(B, 1) < 0.5, 1.0, -1.0)
    return f.reshape(B, 17 * P)


@pytest.mark.parametrize("precision", ["f32", "f32s"])
@pytest.mark.parametrize("N,tower,Bs", [(9, 3, [1, 7, 8, 57, 500, 3000]), (5, 2, [1, 16, 17, 2100]), (12, 1, [4, 5, 333]),
                                        (9, 10, [8192])])
def test_persistent_tower_is_bit_identical_to_per_layer_launches(N, tower, Bs, precision):
    rng = np.random.RandomState(100 * N + tower)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(5)
    eng.set_precision(precision)
    for B in Bs:
        feats = _feats(rng, B, N)
        eng.set_tower_persistent(False)
        pi0, v0 = eng.forward_features(feats)
        eng.set_tower_persistent(True)
        for rep in range(3):          # warm L1 / L2, scheduler words reused
            pi1, v1 = eng.forward_features(feats)
            assert (pi1 == pi0).all() and (v1 == v0).all(), (N, tower, B, rep, np.abs(pi1 - pi0).max(), np.abs(v1 - v0).max())
        # and different data through the same buffers right behind it
        feats2 = _feats(rng, B, N)
        pi2, v2 = eng.forward_features(feats2)
        eng.set_tower_persistent(False)
        pi3, v3 = eng.forward_features(feats2)
        assert (pi2 == pi3).all() and (v2 == v3).all(), (N, tower, B)
    eng.close()


def test_persistent_tower_matches_the_oracle():
    N, tower, B = 9, 4, 91
    A = N * N + 1
    rng = np.random.RandomState(3)
    onet = L.or_net_new(N, tower)
    L.or_net_init_synthetic(onet, 3)
    randomize_bn(onet, list(range(0, 1 + 2 * tower)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], rng)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    copy_weights_from_oracle(eng, onet, tower)
    eng.set_tower_persistent(True)
    feats = _feats(rng, B, N)
    pi = np.zeros((B, A))
    v = np.zeros(B)
    x = feats.astype(np.float64)
    L.or_net_forward_feats_f64(onet, x.ctypes.data_as(C.POINTER(C.c_double)), B, pi.ctypes.data_as(C.POINTER(C.c_double)),
                               v.ctypes.data_as(C.POINTER(C.c_double)))
    gpi, gv = eng.forward_features(feats)
    assert np.abs(gpi - pi).max() <= TOL and np.abs(gv - v).max() <= TOL, (np.abs(gpi - pi).max(), np.abs(gv - v).max())
    L.or_net_free(onet)
    eng.close()


def test_sustained_mfma_rate_is_a_sane_number():
    """bench.py's roofline.sustained_mfma: between a third of and the whole nominal f32 MFMA peak; argument checked."""
    eng = ag.Engine(board_size=5, games=1, tower_height=1, num_readouts=8, max_nodes_per_game=16)
    tf = eng.mfma_sustained_tflops(100)
    assert 50.0 < tf <= 160.0, tf
    with pytest.raises(ag.AgzError):
        eng.mfma_sustained_tflops(10)
    eng.close()


@pytest.mark.parametrize("N,tower,B,chains", [(19, 2, 2048, 2), (19, 3, 1500, 2), (17, 2, 1400, 2), (19, 2, 1311, 2), (19, 2, 2700, 4)])
def test_two_tower_chains_are_bit_identical_to_one(N, tower, B, chains):
    """The F(4x4,3x3) tower of a large batch runs as two independent layer chains -- the two halves of its tile blocks, cut at
    a board boundary -- on two streams (agz_net_set_tower_streams, the default).  Same kernels, same rows: the outputs
    must equal the one-chain form bit for bit, repeatedly (any difference is a missing dependency between the streams),
    also when the batch leaves the second chain nearly empty, and the conv profile must still count every layer."""
    rng = np.random.RandomState(N + B)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(4)
    feats = _feats(rng, B, N)
    eng.set_tower_streams(1)
    pi0, v0 = eng.forward_features(feats)
    eng.set_tower_streams(chains)
    eng.profile_conv(True)
    for rep in range(3):
        pi1, v1 = eng.forward_features(feats)
        assert (pi1 == pi0).all() and (v1 == v0).all(), (rep, np.abs(pi1 - pi0).max())
    ms, flop, n = eng.profile_conv_read()
    eng.profile_conv(False)
    assert n == 3 * 2 * tower and ms > 0 and flop > 0, (ms, flop, n)
    # a batch that fills only the first chain's range, through the same (larger) buffers
    small = B // 3
    eng.set_tower_streams(1)
    spi0, sv0 = eng.forward_features(feats[:small])
    eng.set_tower_streams(chains)
    spi1, sv1 = eng.forward_features(feats[:small])
    assert (spi1 == spi0).all() and (sv1 == sv0).all() and (spi0 == pi0[:small]).all()
    eng.close()


@pytest.mark.parametrize("precision", ["f32", "f32s"])
def test_tower_chains_9x9_are_bit_identical(precision):
    """the same for the F(3x3,3x3) tower with whole-board tile blocks (the 9x9 headline: 8192 positions, 1171 tile blocks):
    1, 2 and 4 chains, per-layer launches and the persistent launch all give the same bits"""
    N, tower, B = 9, 4, 8192
    rng = np.random.RandomState(9)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(6)
    eng.set_precision(precision)
    feats = _feats(rng, B, N)
    eng.set_tower_streams(1)
    pi0, v0 = eng.forward_features(feats)
    for chains in (2, 4, 3):
        eng.set_tower_streams(chains)
        for rep in range(2):
            pi1, v1 = eng.forward_features(feats)
            assert (pi1 == pi0).all() and (v1 == v0).all(), (chains, rep, np.abs(pi1 - pi0).max())
    eng.set_tower_persistent(True)
    pi2, v2 = eng.forward_features(feats)
    assert (pi2 == pi0).all() and (v2 == v0).all()
    eng.close()
