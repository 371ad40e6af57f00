"""The five-pass 64-tile x 128-cout form of the F(3x3,3x3) tower layer (alphago.jl_amd/csrc/agz_wino5.hip,
agz_net_set_winograd(3)) against the float64 oracle (/root/reference/src/resnet.jl:11-32, neural_net.jl:57-68) and
against the one-pass kernel it replaces.

  * <= 1e-4 against the float64 network at every whole-board tiling class (N = 3 .. 12: 1, 4, 9, 16 tiles per board, boards
    whose side is not a multiple of 3 included) and at tower 10 (north_star's tolerance; measured ~1e-8);
  * a position's output does not depend on its batch row or its neighbours (tree parity rests on it);
  * the device-built U image equals the host pack bit for bit;
  * at the bench's batch (8192 positions of 9x9) repeated forwards and 1 / 2 / 4 layer chains are bit-identical, slices
    evaluated alone too, and the result agrees with k_wino_gemm4's to rounding (the two sum the same products in a
    different order);
  * a whole self-play game on it equals the oracle's game (moves, pi, q bit for bit, the oracle calling the same forward).
"""
import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from gpu_common import GpuNetForOracle, copy_weights_from_oracle, pos_soa
from test_gpu_nn import oracle_forward64
from test_hostsim_go import random_positions
from test_oracle_nn import randomize_bn

pytestmark = pytest.mark.gpu
L = orc.lib()
TOL = 1e-4


def _net(N, tower, seed):
    rng = np.random.RandomState(seed)
    onet = L.or_net_new(N, tower)
    L.or_net_init_synthetic(onet, 3)
    randomize_bn(onet, list(range(0, 1 + 2 * tower)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], rng)
    return onet, rng


def _feats(rng, B, N):
    f = (rng.rand(B, 17 * N * N) < 0.3).astype(np.float32)
    f[:, 16 * N * N:] = np.where(rng.rand(B, 1) < 0.5, 1.0, -1.0)
    return f


@pytest.mark.parametrize("N,tower,B", [(3, 2, 23), (4, 2, 40), (5, 1, 7), (6, 2, 23), (7, 2, 70), (8, 2, 23), (9, 2, 37), (9, 10, 16),
                                       (10, 2, 23), (11, 1, 9), (12, 2, 23)])
def test_five_pass_tower_matches_the_float64_oracle(N, tower, B):
    A = N * N + 1
    onet, rng = _net(N, tower, 10 * N + tower)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    eng.set_winograd(3)
    copy_weights_from_oracle(eng, onet, tower)
    feats = _feats(rng, B, N)
    pi64, v64 = oracle_forward64(onet, feats, A)
    gpi, gv = eng.forward_features(feats)
    dpi, dv = np.abs(gpi - pi64).max(), np.abs(gv - v64).max()
    print(f"five-pass F(3x3,3x3) {N}x{N} tower {tower} B={B}: max|dpi| {dpi:.2e} max|dv| {dv:.2e}")
    assert dpi <= TOL and dv <= TOL, (dpi, dv)
    assert np.allclose(gpi.sum(1), 1, atol=1e-5)
    assert eng.debug_pack_diff(6) == 0                      # the U image the kernel read == the host pack
    # the same positions at other block rows, alone, and permuted
    spi, sv = eng.forward_features(feats[5:6])
    assert (spi[0] == gpi[5]).all() and sv[0] == gv[5]
    for lo in (1, 2, 7):
        if lo < B:
            spi, sv = eng.forward_features(feats[lo:])
            assert (spi == gpi[lo:]).all() and (sv == gv[lo:]).all(), lo
    perm = rng.permutation(B)
    ppi, pv = eng.forward_features(feats[perm])
    assert (ppi == gpi[perm]).all() and (pv == gv[perm]).all()
    # and the one-pass kernel computes the same network to rounding
    eng.set_winograd(1)
    opi, ov = eng.forward_features(feats)
    assert np.abs(opi - gpi).max() <= 1e-5 and np.abs(ov - gv).max() <= 1e-5
    L.or_net_free(onet)
    eng.close()


def test_five_pass_full_batch_is_deterministic_and_chain_independent():
    N, tower, B = 9, 4, 8192
    rng = np.random.RandomState(1)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(2)
    eng.set_winograd(3)
    feats = _feats(rng, B, N)
    eng.set_tower_streams(1)
    pi0, v0 = eng.forward_features(feats)
    assert np.isfinite(pi0).all() and np.allclose(pi0.sum(1), 1, atol=1e-5)
    for chains in (1, 2, 4, 3):
        eng.set_tower_streams(chains)
        for _ in range(3):
            pi, v = eng.forward_features(feats)
            assert (pi == pi0).all() and (v == v0).all(), chains
    n = 300
    for lo in (0, 3000, B - n, 1, 977):
        spi, sv = eng.forward_features(feats[lo:lo + n])
        assert (spi == pi0[lo:lo + n]).all() and (sv == v0[lo:lo + n]).all(), lo
    eng.set_winograd(1)
    pi1, v1 = eng.forward_features(feats)
    d = max(np.abs(pi1 - pi0).max(), np.abs(v1 - v0).max())
    print(f"five-pass vs one-pass F(3x3,3x3), 8192 positions of 9x9, tower 4: max difference {d:.2e}")
    assert d <= 1e-5
    eng.close()


def test_whole_games_on_the_five_pass_tower_equal_the_oracle_games():
    from test_gpu_selfplay import check_against_oracle, run
    N, tower, R, games = 9, 2, 32, 3
    eng = ag.Engine(board_size=N, tower_height=tower, games=games, num_readouts=R, seed=2, record_capacity_games=games + 8)
    eng.init_synthetic(0)
    eng.set_winograd(3)
    recs, st = run(eng, games)
    assert len(recs) == games and st["pool_exhausted"] == 0
    fwd = ag.Engine(board_size=N, tower_height=tower, games=1, num_readouts=8, max_nodes_per_game=16)
    fwd.init_synthetic(0)
    fwd.set_winograd(3)
    moves, evals = check_against_oracle(recs, GpuNetForOracle(fwd), N, R, 2)
    assert st["positions"] == moves and st["evals"] == evals
    fwd.close()
    eng.close()
