"""What a game does when its node pool is full (agz_config.pool_policy; include/agz.h).  The reference's tree is
garbage-collected and unbounded (/root/reference/src/mcts.jl:140-147, src/mcts_play.jl:48); the engine's pools are
fixed.  Checked here on the host wave simulator (the same agz_search.h source as the HIP kernels; CPU only):
  * with a pool that is large enough nothing changes -- parity with the oracle is test_hostsim_selfplay.py's job;
  * AGZ_POOL_MOVE_EARLY (default): a starved game still ends, every move it made is legal (its record replays on the
    oracle's rules to the same final score), its shortened searches are counted in the header and in the counters, and
    the OTHER games of the batch -- whose pools never filled -- equal the oracle's games bit for bit;
  * AGZ_POOL_STALL: a starved game never shortens a search: it waits with err = AGZ_POOL_EXHAUSTED while others finish."""
import ctypes as C

import numpy as np

import hs
import orc
from test_hostsim_selfplay import OracleNet, bits_equal, oracle_game

L = orc.lib()
POOL_EXHAUSTED = 8


def replay_on_oracle(N, moves):
    pos = orc.make_pos(N)
    for a in moves:
        rcode, pos = orc.play(pos, int(a))
        assert rcode == orc.OK, f"illegal move {a} in a recorded game"
    return pos


def run(policy, cap, games=6, slots=3, N=5, R=16, max_steps=6000):
    net = OracleNet(N, 1, seed=0)
    sim = hs.Sim(board_size=N, games=slots, num_readouts=R, seed=3, game_id_base=0, game_id_stride=1,
                 record_capacity_games=games + 8, max_nodes_per_game=cap, pool_policy=policy, resign_threshold=-2.0)
    sim.start(games)
    steps = 0
    while sim.counters()["finished"] < games and steps < max_steps:
        sim.step(net.on_feats)
        steps += 1
    return net, sim, steps


def test_default_pool_size_has_a_term_in_the_game_length():
    sim = hs.Sim(board_size=9, games=1, num_readouts=400)
    assert sim.cap == 16 * 400 + 256 + 16 * sim.mgl and sim.mgl == 113
    sim.close()
    sim = hs.Sim(board_size=19, games=1, num_readouts=16)
    assert sim.cap == 16 * 16 + 256 + 16 * 505                       # the 19x19 / 16-readout soak needed ~8000
    sim.close()


def test_move_early_keeps_a_starved_game_going_and_counts_it():
    net, sim, steps = run(policy=0, cap=24)                           # 16 readouts + noise: ~24 nodes is one move's worth
    ct = sim.counters()
    recs = sim.records()
    assert ct["finished"] == 6 and len(recs) == 6, "a starved game must still end"
    assert ct["pool_exhausted"] > 0 and ct["pool_short"] > 0
    assert sum(r["short_searches"] for r in recs) == ct["pool_short"]
    assert ct["peak_nodes"] <= 24
    for r in recs:
        assert 0 < r["num_moves"] <= sim.mgl
        pos = replay_on_oracle(5, r["moves"])
        if not r["was_resign"]:
            sc = L.or_score(C.byref(pos))
            assert abs(sc - r["final_score"]) < 1e-6 and np.sign(sc) == r["result"]
        assert np.allclose(r["pis"].sum(axis=1), 1.0, atol=1e-5)
    sim.close()
    net.close()


def test_games_whose_pool_never_filled_are_still_the_oracles_games():
    """a pool that only the longest trees outgrow: the games it never touched are bit-equal to the oracle's"""
    net, sim, steps = run(policy=0, cap=56, games=8, slots=4)
    recs = sim.records()
    ct = sim.counters()
    assert ct["finished"] == 8
    clean = [r for r in recs if r["short_searches"] == 0]
    assert len(clean) >= 1 and len(clean) < 8 or ct["pool_short"] == 0
    for r in clean:
        o = oracle_game(5, net, 16, 3, int(r["game_id"]), -2.0, 0.05)
        assert r["num_moves"] == o["num_moves"] and (r["moves"] == o["moves"][: r["num_moves"]]).all()
        assert bits_equal(r["pis"], o["pis"]) and bits_equal(r["qs"], o["qs"]) and r["result"] == o["result"]
    sim.close()
    net.close()


def test_stall_policy_waits_instead_of_shortening():
    net, sim, steps = run(policy=1, cap=24, games=6, slots=3, max_steps=1500)
    ct = sim.counters()
    assert ct["pool_short"] == 0, "AGZ_POOL_STALL never plays a move on a shortened search"
    assert ct["pool_exhausted"] > 0
    stalled = [g for g in range(3) if sim.game(g).err == POOL_EXHAUSTED]
    assert stalled, "with 24 nodes a 16-readout game must run into its pool"
    for g in stalled:                                                 # a stalled game keeps its state: nothing was played for it
        before = sim.game(g).move_count
        for _ in range(20):
            sim.step(net.on_feats)
        assert sim.game(g).move_count == before and sim.game(g).err == POOL_EXHAUSTED and sim.game(g).stalled == 1
    assert all(r["short_searches"] == 0 for r in sim.records())
    sim.close()
    net.close()


def test_move_early_games_are_never_reported_as_stalled_between_steps():
    """G.err stays AGZ_POOL_EXHAUSTED from a refused allocation until the NEXT step's pre phase
    plays the early move, so `err == EXHAUSTED and phase == SEARCH` -- what agz_stats.stalled_games counted -- was true
    between steps for games that were not waiting at all, and api.selfplay / check_pool / bench.py aborted in exactly
    the case the default policy survives.  `stalled` is set by game_pre only when the game could NOT move early: polled
    after every step of a starved run it is never set under AGZ_POOL_MOVE_EARLY, while err == EXHAUSTED is seen."""
    net = OracleNet(5, 1, seed=0)
    sim = hs.Sim(board_size=5, games=3, num_readouts=16, seed=3, game_id_base=0, game_id_stride=1,
                 record_capacity_games=16, max_nodes_per_game=24, pool_policy=0, resign_threshold=-2.0)
    sim.start(6)
    err_seen = stalled_seen = steps = 0
    while sim.counters()["finished"] < 6 and steps < 6000:
        sim.step(net.on_feats)
        steps += 1
        for g in range(3):
            G = sim.game(g)
            err_seen += G.err == POOL_EXHAUSTED and G.phase == 3
            stalled_seen += G.stalled != 0
    assert sim.counters()["finished"] == 6 and sim.counters()["pool_short"] > 0
    assert err_seen > 0, "the run did hit the ambiguous state (err set between steps)"
    assert stalled_seen == 0, "no game of a move-early run ever waits"
    sim.close()
    net.close()
