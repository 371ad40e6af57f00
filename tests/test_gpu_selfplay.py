"""Whole self-play games on the GPU against the oracle.

(1) external-network mode: the HIP search kernels drive a CPU network (the oracle's); every
    finished game must equal the oracle's own selfplay() move for move, pi bit for bit.
(2) the engine's own HIP network: the oracle's tree search is run with agz_net_forward as its
    network callable, so both searches see the same numbers; games must again be identical.
    This also proves that features built from ancestor boards equal features built from the
    reference's delta history, and that NN outputs do not depend on batch composition.
(3) size-independent properties on a larger run (BASELINE configs[1] shape, shortened)."""
import ctypes as C

import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from gpu_common import GpuNetForOracle
from test_hostsim_selfplay import OracleNet, bits_equal, oracle_game

pytestmark = pytest.mark.gpu
L = orc.lib()


def run(eng, games, network=None, max_steps=100000):
    eng.start(games)
    steps = 0
    while steps < max_steps:
        if network is None:
            eng.step(8)
            steps += 8
        else:
            eng.step_external(network)
            steps += 1
        if eng.stats()["games_finished"] >= games:
            break
    return eng.records(), eng.stats()


def check_against_oracle(recs, net_for_oracle, N, readouts, seed, **kw):
    moves = evals = 0
    for r in recs:
        o = oracle_game(N, net_for_oracle, readouts, seed, int(r["game_id"]), kw.get("resign_threshold", -0.9),
                        kw.get("resign_disable_fraction", 0.05))
        assert r["num_moves"] == o["num_moves"], r["game_id"]
        assert (r["moves"] == o["moves"][: r["num_moves"]]).all()
        assert r["result"] == o["result"]
        assert r["was_resign"] == (o["result_string"] in (b"B+R", b"W+R"))
        assert bits_equal(r["qs"], o["qs"]) and bits_equal(r["pis"], o["pis"])
        moves += o["num_moves"]
        evals += o["evals"]
    return moves, evals


@pytest.mark.parametrize("N,tower,readouts,games,slots", [(5, 1, 16, 4, 4), (5, 1, 16, 9, 3), (9, 1, 24, 2, 2)])
def test_external_network_games_match_oracle(N, tower, readouts, games, slots):
    net = OracleNet(N, tower, seed=0)
    eng = ag.Engine(board_size=N, tower_height=0, games=slots, num_readouts=readouts, seed=1,
                    external_network=1, record_capacity_games=games + 8)
    recs, st = run(eng, games, network=net.on_feats)
    assert len(recs) == games and st["pool_exhausted"] == 0
    moves, evals = check_against_oracle(recs, net, N, readouts, 1)
    assert st["positions"] == moves and st["evals"] == evals
    eng.close()
    net.close()


@pytest.mark.parametrize("N,tower,readouts,games,slots,kw", [
    (5, 1, 16, 4, 4, {}),                                   # BASELINE.json configs[0]
    (5, 2, 16, 12, 5, dict(resign_threshold=-0.05, resign_disable_fraction=0.5)),
    (9, 2, 32, 3, 3, {}),
    (19, 1, 8, 2, 2, {}),                                   # full-size board (BASELINE configs[3] shape, shortened)
    (9, 10, 400, 1, 1, {}),                                 # BASELINE configs[1] network and readout budget, one whole game
])
def test_internal_network_games_match_oracle(N, tower, readouts, games, slots, kw):
    eng = ag.Engine(board_size=N, tower_height=tower, games=slots, num_readouts=readouts, seed=2,
                    record_capacity_games=games + 8, **kw)
    eng.init_synthetic(0)
    recs, st = run(eng, games)
    assert len(recs) == games and st["pool_exhausted"] == 0
    fwd = ag.Engine(board_size=N, tower_height=tower, games=1, num_readouts=8, max_nodes_per_game=16)
    fwd.init_synthetic(0)
    moves, evals = check_against_oracle(recs, GpuNetForOracle(fwd), N, readouts, 2, **kw)
    assert st["positions"] == moves and st["evals"] == evals
    # replay_position on the device reproduces the features the oracle rebuilds by replaying
    r = recs[0]
    if r["num_moves"]:
        feats = eng.record_features(r["index"], r["num_moves"])
        pos = orc.make_pos(N)
        for k in range(r["num_moves"]):
            assert (feats[k].reshape(17, N * N) == orc.feats(pos)).all(), k
            _, pos = orc.play(pos, int(r["moves"][k]))
    fwd.close()
    eng.close()


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_19x19_games_to_their_end_replay_on_the_oracle(precision):
    """Whole 19x19 games through recycled slots (the F(4x4,3x3) tower with its layer chains / the fp16 tower; 16 readouts):
    every game is filed once, every record replays legally on the oracle's rules and -- unless it was resigned -- ends by
    two passes or at max_game_length = 505 with the result the oracle's Tromp-Taylor count gives for the final board.
    On the DEFAULT node pool (16 R + 256 + 16 max_game_length since round 5: a sharp policy over a 500-move game keeps
    more of its tree than 16 R covers when R is small)."""
    N, G, TOTAL = 19, 32, 40
    eng = ag.Engine(board_size=N, tower_height=1, games=G, num_readouts=16, seed=33, record_capacity_games=TOTAL + 8)
    eng.init_synthetic(0)
    eng.set_precision(precision)
    recs, st = run(eng, TOTAL, max_steps=60000)
    assert len(recs) == TOTAL and st["pool_exhausted"] == 0 and st["pool_short_searches"] == 0
    assert sorted(r["game_id"] for r in recs) == list(range(TOTAL))
    print(f"19x19 {precision}: peak nodes per game {st['peak_nodes_per_game']} of {st['node_capacity']}")
    ended_by_length = 0
    for r in recs:
        pos = orc.make_pos(N)
        for a in r["moves"]:
            rc, pos = orc.play(pos, int(a))
            assert rc == orc.OK
        assert pos.n == r["num_moves"] <= 505
        if not r["was_resign"]:
            assert pos.done or pos.n >= 505
            ended_by_length += not pos.done
            assert r["result"] == orc.lib().or_result(C.byref(pos))
    print(f"19x19 {precision}: {TOTAL} games, {sum(r['num_moves'] for r in recs)} moves, "
          f"{sum(bool(r['was_resign']) for r in recs)} resigned, {ended_by_length} at max_game_length")
    eng.close()


def test_properties_at_scale():
    """9x9 / tower 2 / 64 games / 32 readouts: invariants that need no oracle"""
    N, A = 9, 82
    eng = ag.Engine(board_size=N, tower_height=2, games=64, num_readouts=32, seed=9, record_capacity_games=80)
    eng.init_synthetic(0)
    recs, st = run(eng, 64)
    assert len(recs) == 64 and st["pool_exhausted"] == 0
    assert sorted(r["game_id"] for r in recs) == list(range(64))
    for r in recs:
        n = r["num_moves"]
        assert 1 <= n <= 113
        assert r["result"] in (-1, 0, 1)
        pis = r["pis"]
        ok = ~np.isnan(pis).any(axis=1)
        assert np.allclose(pis[ok].sum(1), 1, atol=1e-4) and (pis[ok] >= 0).all()
        assert (np.abs(r["qs"]) <= 1.0 + 1e-6).all()
        # every recorded game replays legally on the oracle and ends where the record says
        pos = orc.make_pos(N)
        for k in range(n):
            rcode, pos = orc.play(pos, int(r["moves"][k]))
            assert rcode == orc.OK
        if not r["was_resign"]:
            assert pos.done or pos.n >= 113
            assert r["final_score"] == L.or_score(C.byref(pos))
            assert r["result"] == L.or_result(C.byref(pos))
    # same seed => same games (determinism across runs and across slot counts)
    eng2 = ag.Engine(board_size=N, tower_height=2, games=16, num_readouts=32, seed=9, record_capacity_games=80)
    eng2.init_synthetic(0)
    recs2, _ = run(eng2, 24)
    for r2 in recs2:
        r1 = recs[int(r2["game_id"])]
        assert (r1["moves"] == r2["moves"]).all() and bits_equal(r1["pis"], r2["pis"])
    eng.close()
    eng2.close()


def test_full_size_config_invariants():
    """BASELINE.json configs[1] at full size (GoEnv(9), tower 10, 400 readouts, 1024 concurrent games,
    batches of up to 8192 leaves) for 60 steps: properties that need no oracle."""
    N, A, G, R = 9, 82, 1024, 400
    eng = ag.Engine(board_size=N, tower_height=10, games=G, num_readouts=R, seed=11, stagger_moves=40)
    eng.init_synthetic(0)
    eng.start(0)
    eng.step(60)
    st = eng.stats()
    assert st["pool_exhausted"] == 0 and st["steps"] == 60
    assert st["evals"] <= 60 * 8 * G and st["evals"] >= 0.9 * 59 * 8 * G           # the batch stays full
    assert st["root_visits"] >= st["evals"] - G                                     # every collected leaf is a visit
    assert st["positions"] > 0 and st["games_started"] >= G     # (a game's shortened first move is not counted)
    rng = np.random.RandomState(0)
    for g in rng.choice(G, 24, replace=False):
        g = int(g)
        root = eng.tree_root(g)
        info = eng.node_info(g, root)
        assert eng.pending_vlosses(g) == 0                                           # mcts_play.jl:92: all reverted
        cn = eng.node_floats(g, root, 0)
        assert (cn >= 0).all() and (cn == np.round(cn)).all()
        if info.is_expanded:
            assert info.N == 1 + cn.sum() or info.N == cn.sum()                      # the root's own first visit
            legal = eng.go_legal(eng.node_board(g, root)[None], [info.pos.to_play], [info.pos.ko])[0]
            assert not (cn[legal == 0] > 0).any()                                    # test_mcts.jl:146-167 at scale
            prior = eng.node_floats(g, root, 2)
            assert abs(prior.sum() - 1.0) < 1e-3                                     # 0.75 p + 0.25 dirichlet
    eng.close()


def test_configs1_full_shard_first_move_matches_oracle():
    """BASELINE.json configs[1] at its full size -- GoEnv(9), tower 10, 400 readouts, 1024 concurrent games, every
    network call a batch of 8192 leaves, no stagger -- held against the oracle where the invariants above only look at
    properties (VERDICT r3 #3c): for 8 sampled game slots the FIRST recorded move (the move, pi = children_as_pi of
    the root's visit counts, q) is compared bit for bit with the oracle's selfplay of the same game id, whose network
    callable is the HIP forward; and the complete tree of the slot (every node's child_N / child_W / priors / children,
    tests/test_gpu_tree.py) with the oracle twin advanced by the same number of tree_search! calls behind that move."""
    from test_gpu_tree import compare_trees
    N, A, G, R, seed = 9, 82, 1024, 400, 11
    eng = ag.Engine(board_size=N, tower_height=10, games=G, num_readouts=R, seed=seed)
    eng.init_synthetic(0)
    eng.start(0)
    rng = np.random.RandomState(3)
    sample = sorted({0, G - 1} | set(int(g) for g in rng.choice(G, 6, replace=False)))
    moved_at, steps = {}, 0
    eng.step(R // 8 - 2)                      # nobody can have moved yet: a step adds at most 8 readouts to a root
    steps = R // 8 - 2
    assert all(eng.debug_live_record(g)[1] == 0 for g in sample)
    while len(moved_at) < len(sample) and steps < R // 8 + 40:
        eng.step(1)
        steps += 1
        for g in sample:
            if g not in moved_at and eng.debug_live_record(g)[1] >= 1:
                moved_at[g] = steps
    assert len(moved_at) == len(sample), moved_at
    st = eng.stats()
    assert st["pool_exhausted"] == 0 and st["evals"] >= 0.9 * (steps - 1) * 8 * G      # the batches really were full
    fwd = ag.Engine(board_size=N, tower_height=10, games=1, num_readouts=8, max_nodes_per_game=16)
    fwd.init_synthetic(0)
    net = GpuNetForOracle(fwd)
    env = orc.env(N)
    for g in sample:
        gid, nm, mv, pi, q = eng.debug_live_record(g, 0)
        assert nm == 1                              # (the second move needs another 50 steps)
        p = L.or_selfplay_ex(N, net.cb, None, R, seed, gid, 1, -0.9, 0.05)
        assert L.or_player_num_moves(p) == 1 and L.or_player_result(p) == 0
        oroot = L.or_player_root(p)
        opos = L.or_node_pos(oroot).contents
        assert mv == opos.recent_move[0], (g, gid)
        assert bits_equal(pi, orc.node_arr(L.or_player_search_pi(p, 0), A)), (g, gid)
        assert bits_equal(np.float32(q), np.float32(L.or_player_q(p, 0))), (g, gid)
        # the step in which a game moves also injects the new root's noise and runs one tree_search! on it
        d = orc.ODraw(seed, gid, opos.n, 0)
        L.or_inject_noise(C.byref(env), oroot, C.byref(d))
        for _ in range(steps - moved_at[g] + 1):
            L.or_player_tree_search(p, 8)
        assert compare_trees(eng, g, eng.tree_root(g), L.or_player_root(p)) > 8
        L.or_player_free(p)
    fwd.close()
    eng.close()
