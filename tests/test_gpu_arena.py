"""evaluate() arena on the GPU (src/neural_net.jl:103-158): the paired-slot kernels with (1) two CPU
networks supplied from outside and (2) the engine's own two HIP networks, against the oracle's
or_evaluate_game; plus the host-level evaluate() wrapper."""
import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from gpu_common import GpuNetForOracle
from test_hostsim_arena import oracle_eval_game
from test_hostsim_selfplay import OracleNet, bits_equal

pytestmark = pytest.mark.gpu


def check(recs, black, white, N, readouts, seed, threshold=-0.9):
    evals = 0
    for r in recs:
        o = oracle_eval_game(N, black, white, readouts, seed, int(r["game_id"]) // 2, threshold)
        assert r["num_moves"] == o["num_moves"], r["game_id"]
        assert (r["moves"] == o["moves"]).all()
        assert bits_equal(r["qs"], o["qs"])
        assert r["result"] == o["result"] and bool(r["was_resign"]) == bool(o["was_resign"])
        assert np.float32(r["final_score"]) == np.float32(o["final_score"])
        evals += sum(o["evals"])
    return evals


@pytest.mark.parametrize("N,readouts,games,slots,thr", [(5, 16, 6, 4, -0.9), (5, 12, 9, 6, -0.05), (9, 16, 2, 4, -0.9)])
def test_arena_external_networks_match_oracle(N, readouts, games, slots, thr):
    black, white = OracleNet(N, 1, seed=0), OracleNet(N, 1, seed=5)
    eng = ag.Engine(board_size=N, tower_height=0, games=slots, num_readouts=readouts, seed=1, arena_mode=1,
                    external_network=1, record_capacity_games=games + 8, resign_threshold=thr)
    eng.start(games)
    for _ in range(200000):
        eng.step_external(black.on_feats, white.on_feats)
        if eng.stats()["games_finished"] >= games:
            break
    recs, st = eng.records(), eng.stats()
    assert len(recs) == games and st["pool_exhausted"] == 0
    assert sorted(int(r["game_id"]) // 2 for r in recs) == list(range(games))
    assert check(recs, black, white, N, readouts, 1, thr) == st["evals"]
    eng.close()
    black.close()
    white.close()


@pytest.mark.parametrize("N,tower,readouts,games,slots", [(5, 1, 16, 6, 4), (9, 2, 24, 3, 4)])
def test_arena_internal_networks_match_oracle(N, tower, readouts, games, slots):
    eng = ag.Engine(board_size=N, tower_height=tower, games=slots, num_readouts=readouts, seed=3, arena_mode=1,
                    record_capacity_games=games + 8)
    eng.init_synthetic(0)
    eng.net_select(1)
    eng.init_synthetic(5)
    eng.net_select(0)
    eng.start(games)
    for _ in range(100000):
        eng.step(8)
        if eng.stats()["games_finished"] >= games:
            break
    recs, st = eng.records(), eng.stats()
    assert len(recs) == games and st["pool_exhausted"] == 0
    fb = ag.Engine(board_size=N, tower_height=tower, games=1, num_readouts=8, max_nodes_per_game=16)
    fw = ag.Engine(board_size=N, tower_height=tower, games=1, num_readouts=8, max_nodes_per_game=16)
    fb.init_synthetic(0)
    fw.init_synthetic(5)
    assert check(recs, GpuNetForOracle(fb), GpuNetForOracle(fw), N, readouts, 3) == st["evals"]
    # network 1 really is a different network, and net_select addresses it for forward too
    x = (np.random.RandomState(0).rand(1, 17 * N * N) < 0.3).astype(np.float32)
    p0, _ = eng.forward_features(x)
    eng.net_select(1)
    p1, _ = eng.forward_features(x)
    assert (p1 == fw.forward_features(x)[0]).all() and (p0 == fb.forward_features(x)[0]).all() and (p0 != p1).any()
    for e in (eng, fb, fw):
        e.close()


def test_evaluate_wrapper_and_errors():
    env = ag.GoEnv(5)
    a, b = ag.NeuralNet(env, tower_height=1, seed=0), ag.NeuralNet(env, tower_height=1, seed=5)
    ok, st = ag.evaluate(env, a, b, num_games=8, ro=16, seed=2, return_stats=True)
    assert st.num_games == 8 and 0 <= st.games_won <= 8 and ok == (st.games_won / 8 >= 0.55)
    assert len(st.records) == 8 and st.moves > 8
    ok2, st2 = ag.evaluate(env, a, b, num_games=8, ro=16, seed=2, slots=3, return_stats=True)     # slot count is invisible
    assert st2.games_won == st.games_won and st2.moves == st.moves
    with pytest.raises(ValueError):
        ag.evaluate(env, a, ag.NeuralNet(env, tower_height=2), num_games=2, ro=8)
    with pytest.raises(ag.AgzError):
        ag.Engine(board_size=5, games=3, num_readouts=8, arena_mode=1)
    e1 = ag.Engine(board_size=5, games=2, num_readouts=8)
    with pytest.raises(ag.AgzError):
        e1.net_select(1)                       # no second network outside arena mode
    e1.close()
