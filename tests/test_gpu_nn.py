"""HIP network (features + MFMA tower + heads, through the C ABI) against the oracle.
Tolerance: |d pi|, |d v| <= 1e-4 versus the float64 oracle (BASELINE.json north_star); feature
planes are integer-valued and must match exactly."""
import ctypes as C

import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from gpu_common import copy_weights_from_oracle, pos_soa
from test_hostsim_go import random_positions
from test_oracle_nn import randomize_bn

pytestmark = pytest.mark.gpu
L = orc.lib()
TOL = 1e-4


def oracle_forward64(onet, feats, A):
    B = feats.shape[0]
    pi = np.zeros((B, A))
    v = np.zeros(B)
    x = feats.astype(np.float64)
    L.or_net_forward_feats_f64(onet, x.ctypes.data_as(C.POINTER(C.c_double)), B,
                               pi.ctypes.data_as(C.POINTER(C.c_double)), v.ctypes.data_as(C.POINTER(C.c_double)))
    return pi, v


@pytest.mark.parametrize("N", [5, 9, 19])
def test_features_exact(N):
    positions = random_positions(N, 6, 60, seed=N)[::3]
    eng = ag.Engine(board_size=N, games=1, tower_height=0, num_readouts=8, max_nodes_per_game=16)
    got = eng.features(*pos_soa(positions))
    for b, p in enumerate(positions):
        assert (got[b].reshape(17, N * N) == orc.feats(p)).all(), b
    eng.close()


@pytest.mark.parametrize("winograd", [1, 0])
@pytest.mark.parametrize("N,tower,B", [(5, 1, 7), (9, 2, 37), (9, 10, 16), (19, 3, 5), (7, 2, 70)])
def test_forward_matches_oracle(N, tower, B, winograd):
    A = N * N + 1
    rng = np.random.RandomState(N + tower)
    onet = L.or_net_new(N, tower)
    L.or_net_init_synthetic(onet, 3)
    randomize_bn(onet, list(range(0, 1 + 2 * tower)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], rng)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    eng.set_winograd(winograd)      # Winograd F(3x3,3x3) and the direct implicit GEMM must both hold 1e-4
    copy_weights_from_oracle(eng, onet, tower)
    positions = random_positions(N, 4, 80, seed=7)
    positions = [positions[i] for i in rng.choice(len(positions), B, replace=False)]
    feats = np.stack([orc.feats(p).reshape(-1) for p in positions])
    pi64, v64 = oracle_forward64(onet, feats, A)
    gpi, gv = eng.forward(*pos_soa(positions))
    assert np.abs(gpi - pi64).max() <= TOL, np.abs(gpi - pi64).max()
    assert np.abs(gv - v64).max() <= TOL, np.abs(gv - v64).max()
    assert np.allclose(gpi.sum(1), 1, atol=1e-5)
    # the feature-tensor entry point agrees bit for bit with the position entry point
    fpi, fv = eng.forward_features(feats)
    assert (fpi == gpi).all() and (fv == gv).all()
    # a position's output does not depend on its batch neighbours (the tree parity tests rely on it)
    perm = rng.permutation(B)
    ppi, pv = eng.forward(*pos_soa([positions[i] for i in perm]))
    assert (ppi == gpi[perm]).all() and (pv == gv[perm]).all()
    spi, sv = eng.forward(*pos_soa(positions[:1]))
    assert (spi[0] == gpi[0]).all() and sv[0] == gv[0]
    L.or_net_free(onet)
    eng.close()


@pytest.mark.parametrize("precision", ["f32", "f32s"])
@pytest.mark.parametrize("N", [3, 4, 6, 8, 10, 12, 13, 15, 16, 17, 18, 19])
def test_every_tiling_class_of_the_winograd_tower(N, precision):
    """Tile blocks hold whole boards for N <= 12 (T*T = 1, 4, 9, 16 tiles per board: 64, 64, 63, 64 rows used,
    next layer's input transform fused into the GEMM epilogue) and are packed densely above (N = 13..15: 25
    tiles, 16..18: 36, 19: 49 -- the f32s rows; exact f32 runs F(4x4,3x3) from 13x13 on: 16 tiles per board and four
    whole boards per block for N = 13..16, 25 tiles and five boards per block PAIR for N = 17..19; the epilogue emits the V of every
    tile whose input patch lies inside its block and a fix-up transform does the block ends); boards whose side is not
    a multiple of the tile have tiles hanging over the edge.  One parity check per class, batch sizes that leave a
    partial last block; a position's output must not depend on its batch row (which decides, for dense blocks, which
    of the two producers of V a tile gets)."""
    tower, B = 2, 23
    A = N * N + 1
    rng = np.random.RandomState(N)
    onet = L.or_net_new(N, tower)
    L.or_net_init_synthetic(onet, 3)
    randomize_bn(onet, list(range(0, 1 + 2 * tower)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], rng)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    copy_weights_from_oracle(eng, onet, tower)
    eng.set_precision(precision)
    feats = (rng.rand(B, 17 * N * N) < 0.3).astype(np.float32)
    feats[:, 16 * N * N:] = np.where(rng.rand(B, 1) < 0.5, 1.0, -1.0)
    pi64, v64 = oracle_forward64(onet, feats, A)
    gpi, gv = eng.forward_features(feats)
    assert np.abs(gpi - pi64).max() <= TOL and np.abs(gv - v64).max() <= TOL, (np.abs(gpi - pi64).max(), np.abs(gv - v64).max())
    spi, sv = eng.forward_features(feats[5:6])
    assert (spi[0] == gpi[5]).all() and sv[0] == gv[5]
    for lo in (1, 2, 7, 11):                        # the same positions at other block rows
        spi, sv = eng.forward_features(feats[lo:])
        assert (spi == gpi[lo:]).all() and (sv == gv[lo:]).all(), lo
    L.or_net_free(onet)
    eng.close()


def test_synthetic_init_matches_oracle_stream():
    """agz_net_init_synthetic and the oracle draw the same tensors from the shared draw stream"""
    N, tower = 9, 1
    onet = L.or_net_new(N, tower)
    L.or_net_init_synthetic(onet, 11)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(11)
    positions = random_positions(N, 1, 30, seed=2)[:8]
    feats = np.stack([orc.feats(p).reshape(-1) for p in positions])
    pi64, v64 = oracle_forward64(onet, feats, N * N + 1)
    gpi, gv = eng.forward(*pos_soa(positions))
    assert np.abs(gpi - pi64).max() <= 1e-5 and np.abs(gv - v64).max() <= 1e-5
    L.or_net_free(onet)
    eng.close()


def test_weight_shape_errors():
    eng = ag.Engine(board_size=9, games=1, tower_height=1, num_readouts=8, max_nodes_per_game=16)
    with pytest.raises(ag.AgzError) as ei:
        eng.set_weights(0, 0, np.zeros(10, np.float32))
    assert ei.value.status == ag._lib.BAD_SHAPE
    assert eng.param_count(0, 0) == 3 * 3 * 17 * 256
    assert eng.param_count(ag._lib.L_POLICY_FC, 0) == 162 * 82
    eng.close()


@pytest.mark.parametrize("N,tower,B,precision", [
    (9, 4, 8192, "f32"), (9, 4, 8192, "f16"), (9, 4, 8192, "f32s"),      # the bench's batch (configs[1])
    (19, 2, 2048, "f32"), (19, 2, 2048, "f32s"),                         # configs[3]'s shard: dense tile blocks
    (19, 2, 4096, "f16"),                                                # configs[4]'s shard
])
def test_full_batch_is_deterministic_and_matches_small_batches(N, tower, B, precision):
    """race screen at the full batch sizes (every CU busy, DMA rings full): repeated forwards are bit-identical, and so
    is a slice evaluated as a small batch.  At 19x19 tile blocks are packed densely: a tile's V comes from the GEMM
    epilogue of the layer before or from the fix-up transform depending on where its batch row puts it in a block, so
    slices that start at different rows (every residue of the 49- / 25-tile board stride against the 64-row blocks)
    check that the two producers agree to the bit (VERDICT r3 #3b)."""
    rng = np.random.RandomState(1)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(2)
    eng.set_precision(precision)
    feats = (rng.rand(B, 17 * N * N) < 0.25).astype(np.float32)
    feats[:, 16 * N * N:] = np.where(rng.rand(B, 1) < 0.5, 1.0, -1.0)
    pi0, v0 = eng.forward_features(feats)
    for _ in range(6 if N == 9 else 3):
        pi, v = eng.forward_features(feats)
        assert (pi == pi0).all() and (v == v0).all()
    n = 300 if N == 9 else 67
    for lo in (0, 3000 % (B - n), B - n, 1, 977 % (B - n)):
        spi, sv = eng.forward_features(feats[lo:lo + n])
        assert (spi == pi0[lo:lo + n]).all() and (sv == v0[lo:lo + n]).all(), lo
    assert np.isfinite(pi0).all() and np.allclose(pi0.sum(1), 1, atol=1e-5)
    eng.close()
