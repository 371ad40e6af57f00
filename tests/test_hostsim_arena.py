"""evaluate() arena (src/neural_net.jl:103-158): the paired-slot state machine
of agz_search.h (arena_pre / arena_move_phase under the host wave simulator) against the oracle's
one-game-at-a-time or_evaluate_game: same two networks, same draw streams -> identical moves,
mover Q's, result, resign flag and final score for every game.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import hs
import orc
from test_hostsim_selfplay import OracleNet, bits_equal

L = orc.lib()


def oracle_eval_game(N, black, white, readouts, seed, game, threshold=-0.9):
    mgl = (N * N * 7) // 5
    moves = np.zeros(mgl + 2, np.int16)
    qs = np.zeros(mgl + 2, np.float32)
    out = orc.OEvalGame()
    L.or_evaluate_game(N, black.cb, None, white.cb, None, readouts, threshold, seed, game,
                       moves.ctypes.data_as(C.POINTER(C.c_int16)), orc.fptr(qs), C.byref(out))
    n = out.num_moves
    return dict(num_moves=n, moves=moves[:n].copy(), qs=qs[:n].copy(), result=out.result, was_resign=out.was_resign,
                black_won=out.black_won, final_score=out.final_score, evals=(out.evals_black, out.evals_white))


def run_arena(N, black, white, readouts, seed, games, slots, max_steps=400000, **cfg):
    sim = hs.Sim(board_size=N, games=slots, num_readouts=readouts, seed=seed, arena_mode=1,
                 record_capacity_games=games + 8, **cfg)
    sim.start(games)
    steps = 0
    while sim.counters()["finished"] < games and steps < max_steps:
        sim.step(black.on_feats, white.on_feats)
        steps += 1
    recs, ct = sim.records(), sim.counters()
    sim.close()
    return recs, ct, steps


def check_arena(N, towers, readouts, seed, games, slots, **cfg):
    black, white = OracleNet(N, towers[0], seed=0), OracleNet(N, towers[1], seed=5)
    recs, ct, steps = run_arena(N, black, white, readouts, seed, games, slots, **cfg)
    assert len(recs) == games and ct["pool_exhausted"] == 0
    assert sorted(int(r["game_id"]) // 2 for r in recs) == list(range(games))
    won = resigned = evals = 0
    for r in recs:
        o = oracle_eval_game(N, black, white, readouts, seed, int(r["game_id"]) // 2, cfg.get("resign_threshold", -0.9))
        assert r["num_moves"] == o["num_moves"], r["game_id"]
        assert (r["moves"] == o["moves"]).all()
        assert bits_equal(r["qs"], o["qs"])
        assert r["result"] == o["result"] and bool(r["was_resign"]) == bool(o["was_resign"])
        assert np.float32(r["final_score"]) == np.float32(o["final_score"])
        assert (r["final_score"] > 0) == bool(o["black_won"])           # neural_net.jl:147
        if r["num_moves"]:
            assert not np.asarray(r["pis"]).any()                      # two_player_mode records no pi
        won += o["black_won"]
        resigned += o["was_resign"]
        evals += sum(o["evals"])
    assert ct["evals"] == evals
    black.close()
    white.close()
    return dict(won=won, resigned=resigned, steps=steps)


def test_arena_5x5_two_networks():
    st = check_arena(5, (1, 1), 16, seed=1, games=6, slots=4)
    assert st["steps"] > 10


def test_arena_more_games_than_pairs_and_resigns():
    st = check_arena(5, (1, 2), 12, seed=2, games=9, slots=6, resign_threshold=-0.05)
    assert st["resigned"] > 0


def test_arena_9x9():
    check_arena(9, (1, 1), 16, seed=3, games=2, slots=4)


def test_arena_needs_even_slots_on_the_device_path():
    # the simulator shares fill_dims with the engine; the engine's validate() is covered in test_abi
    sim = hs.Sim(board_size=5, games=2, num_readouts=8, seed=1, arena_mode=1)
    sim.start(1)
    for _ in range(2000):
        sim.step(lambda f: (np.full((len(f), 26), 1 / 26, np.float32), np.zeros(len(f), np.float32)),
                 lambda f: (np.full((len(f), 26), 1 / 26, np.float32), np.zeros(len(f), np.float32)))
        if sim.counters()["finished"] >= 1:
            break
    assert sim.counters()["finished"] == 1
    sim.close()


def test_arena_pool_exhaustion_is_visible_and_files_every_game_once():
    """node-pool exhaustion in the arena.  (a) A pool that runs out now and then: every game is
    still filed exactly once (a side that cannot follow its partner's move raises the pair's abort word and
    the side that moved files the game as void instead of waiting for ever).  (b) A pool too small to search
    at all cannot make progress -- that must show in `pool_exhausted` from the first steps on, which is what
    the selfplay()/evaluate() wrappers poll inside their stepping loops to raise instead of hanging."""
    N, games = 5, 6
    black, white = OracleNet(N, 1, seed=0), OracleNet(N, 1, seed=5)
    recs, ct, steps = run_arena(N, black, white, 24, 4, games, 4, max_steps=20000, max_nodes_per_game=30)
    assert ct["pool_exhausted"] > 0
    assert steps < 20000 and len(recs) == games
    assert sorted(int(r["game_id"]) // 2 for r in recs) == list(range(games))
    recs, ct, steps = run_arena(N, black, white, 24, 4, games, 4, max_steps=40, max_nodes_per_game=6)
    assert ct["pool_exhausted"] > 0 and len(recs) == 0
    black.close()
    white.close()
