"""agz_config.pool_policy on the GPU, through the C ABI (the CPU twin of these checks runs the same agz_search.h source
on the host wave simulator: tests/test_pool_policy.py).  The reference's tree is unbounded
(/root/reference/src/mcts.jl:140-147, src/mcts_play.jl:48); here a full pool either ends the current move's search early
(AGZ_POOL_MOVE_EARLY, default; counted) or parks the slot until the host abandons the game (AGZ_POOL_STALL) -- and the
other slots keep stepping either way."""
import ctypes as C

import numpy as np
import pytest

import alphago_jl_amd as ag
import orc

pytestmark = pytest.mark.gpu
L = orc.lib()
POOL_EXHAUSTED = ag._lib.POOL_EXHAUSTED


def engine(policy, cap, slots=8, total=16):
    eng = ag.Engine(board_size=5, tower_height=1, games=slots, num_readouts=16, seed=3, record_capacity_games=total + 8,
                    max_nodes_per_game=cap, pool_policy=policy, resign_threshold=-2.0)
    eng.init_synthetic(0)
    eng.start(total)
    return eng


def test_move_early_finishes_every_game_and_counts_the_short_searches():
    eng = engine(0, 24)
    for _ in range(4000):
        eng.step(8)
        if eng.records_count() >= 16:
            break
    st, recs = eng.stats(), eng.records()
    assert len(recs) == 16 and st["stalled_games"] == 0
    assert st["pool_exhausted"] > 0 and st["pool_short_searches"] > 0 and st["peak_nodes_per_game"] <= 24
    assert sum(r["short_searches"] for r in recs) == st["pool_short_searches"]
    for r in recs:
        pos = orc.make_pos(5)
        for a in r["moves"]:
            rc, pos = orc.play(pos, int(a))
            assert rc == orc.OK
        assert pos.done or pos.n >= 35
        assert r["result"] == L.or_result(C.byref(pos)) and abs(r["final_score"] - L.or_score(C.byref(pos))) < 1e-6
    eng.close()


def test_games_with_room_equal_the_games_of_an_unbounded_pool():
    """the same game ids with the default pool (never full) and with 56 nodes: every game the small pool never touched
    (short_searches == 0) is bit-identical; the batch around a starved game is not disturbed by it"""
    big = engine(0, 0)
    small = engine(0, 56)
    for eng in (big, small):
        for _ in range(4000):
            eng.step(8)
            if eng.records_count() >= 16:
                break
    want = {r["game_id"]: r for r in big.records()}
    assert big.stats()["pool_exhausted"] == 0
    clean = [r for r in small.records() if r["short_searches"] == 0]
    assert clean, "56 nodes starve some 16-readout games, not all"
    for r in clean:
        w = want[r["game_id"]]
        assert r["num_moves"] == w["num_moves"] and (r["moves"] == w["moves"]).all() and r["result"] == w["result"]
        assert r["pis"].tobytes() == w["pis"].tobytes() and r["qs"].tobytes() == w["qs"].tobytes()
    big.close()
    small.close()


def test_stall_policy_reports_per_slot_and_abandon_frees_the_slot():
    eng = engine(1, 24, slots=8, total=0)                   # total 0: slots recycle for ever
    stalled = []
    for _ in range(400):
        eng.step(8)
        status, nodes, moves = eng.slot_status()
        stalled = [g for g in range(8) if status[g] == POOL_EXHAUSTED]
        if len(stalled) >= 2:
            break
    st = eng.stats()
    assert len(stalled) >= 2 and st["stalled_games"] >= 2 and st["pool_short_searches"] == 0
    assert all(nodes[g] == 24 for g in stalled)
    g0 = stalled[0]
    before = int(moves[g0])
    started = st["games_started"]
    fin0 = st["games_finished"]
    eng.step(40)
    status, nodes, moves2 = eng.slot_status()
    assert status[g0] == POOL_EXHAUSTED and moves2[g0] == before, "a stalled game does not move"
    eng.slot_abandon(g0)
    eng.step(4)
    status, nodes, moves3 = eng.slot_status()
    st2 = eng.stats()
    assert status[g0] == 0 and st2["games_started"] > started and nodes[g0] < 24      # the slot plays a new game
    assert st2["games_finished"] >= fin0                                             # and nobody else was held up
    with pytest.raises(ag.AgzError):
        eng.slot_abandon(99)
    eng.close()
