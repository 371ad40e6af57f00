"""AGZ_PRECISION_F32S: the f32 network with the Winograd GEMM operands carried as two IEEE halves each (fp16 MFMA,
all four cross products, f32 accumulate) against the float64 oracle.  Bar: the same 1e-4 as the exact-f32 path
(BASELINE.json north_star); the measured error is printed.  Opt-in, not what bench.py measures by default."""
import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from gpu_common import GpuNetForOracle, copy_weights_from_oracle, pos_soa
from test_gpu_nn import oracle_forward64
from test_gpu_selfplay import check_against_oracle, run
from test_hostsim_go import random_positions
from test_oracle_nn import randomize_bn

pytestmark = pytest.mark.gpu
L = orc.lib()
TOL = 1e-4


@pytest.mark.parametrize("N,tower,B", [(5, 1, 7), (9, 2, 37), (9, 10, 16), (19, 3, 5), (7, 2, 70), (19, 20, 8)])
def test_f32s_forward_matches_oracle(N, tower, B):
    A = N * N + 1
    rng = np.random.RandomState(N + tower)
    onet = L.or_net_new(N, tower)
    L.or_net_init_synthetic(onet, 3)
    randomize_bn(onet, list(range(0, 1 + 2 * tower)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], rng)
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    copy_weights_from_oracle(eng, onet, tower)
    positions = random_positions(N, 4, 80, seed=7)
    positions = [positions[i] for i in rng.choice(len(positions), B, replace=False)]
    feats = np.stack([orc.feats(p).reshape(-1) for p in positions])
    pi64, v64 = oracle_forward64(onet, feats, A)
    pi32, v32 = eng.forward(*pos_soa(positions))
    eng.set_precision("f32s")
    gpi, gv = eng.forward(*pos_soa(positions))
    dpi, dv = np.abs(gpi - pi64).max(), np.abs(gv - v64).max()
    print(f"N={N} tower={tower}: f32s vs f64 |dpi| {dpi:.2e} |dv| {dv:.2e}   (exact f32: {np.abs(pi32 - pi64).max():.2e} "
          f"{np.abs(v32 - v64).max():.2e})")
    assert dpi <= TOL and dv <= TOL, (dpi, dv)
    assert np.allclose(gpi.sum(1), 1, atol=1e-5)
    if tower >= 2:
        assert (gpi != pi32).any() or (gv != v32).any()          # it really is a different arithmetic
    perm = rng.permutation(B)                                    # batch-row independence (tree parity relies on it)
    ppi, pv = eng.forward(*pos_soa([positions[i] for i in perm]))
    assert (ppi == gpi[perm]).all() and (pv == gv[perm]).all()
    eng.set_precision("f32")
    bpi, bv = eng.forward(*pos_soa(positions))
    assert (bpi == pi32).all() and (bv == v32).all()             # the switch is not sticky
    L.or_net_free(onet)
    eng.close()


def test_f32s_selfplay_games_match_oracle():
    N, tower, readouts, games = 9, 2, 24, 2
    eng = ag.Engine(board_size=N, tower_height=tower, games=games, num_readouts=readouts, seed=6, record_capacity_games=games + 8)
    eng.init_synthetic(0)
    eng.set_precision("f32s")
    recs, st = run(eng, games)
    assert len(recs) == games and st["pool_exhausted"] == 0
    fwd = ag.Engine(board_size=N, tower_height=tower, games=1, num_readouts=8, max_nodes_per_game=16)
    fwd.init_synthetic(0)
    fwd.set_precision("f32s")
    moves, evals = check_against_oracle(recs, GpuNetForOracle(fwd), N, readouts, 6)
    assert st["positions"] == moves and st["evals"] == evals
    fwd.close()
    eng.close()
