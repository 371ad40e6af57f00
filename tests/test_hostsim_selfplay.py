"""Whole-game parity of the batched self-play state machine (agz_search.h game_pre / game_post
under the host wave simulator) against the oracle's one-game-at-a-time selfplay(): with the
same network function and the same draw stream every game must produce identical moves, search
distributions pi, root values q, result and resign flag -- "identical visit counts and selected
moves under a fixed RNG" (BASELINE.json north_star).  Includes BASELINE.json configs[0]
(GoEnv(5), tower 1, 16 readouts, 4 games).  CPU only."""
import ctypes as C

import numpy as np
import pytest

import hs
import orc

L = orc.lib()


class OracleNet:
    def __init__(self, N, tower, seed):
        self.N, self.P, self.A = N, N * N, N * N + 1
        self.net = L.or_net_new(N, tower)
        L.or_net_init_synthetic(self.net, seed)
        self.cb = orc.NET_FN(lambda ctx, pos, B, pi, v: L.or_net_callable(self.net, pos, B, pi, v))

    def on_feats(self, feats):
        B = feats.shape[0]
        pi = np.zeros((B, self.A), np.float32)
        v = np.zeros(B, np.float32)
        L.or_net_forward_feats(self.net, orc.fptr(feats), B, orc.fptr(pi), orc.fptr(v), 32)
        return pi, v

    def close(self):
        L.or_net_free(self.net)


def oracle_game(N, net, readouts, seed, game, threshold=-0.9, disable=0.05):
    p = L.or_selfplay_ex(N, net.cb, None, readouts, seed, game, 0, threshold, disable)
    A = N * N + 1
    n = L.or_player_num_moves(p)
    pos = L.or_node_pos(L.or_player_root(p)).contents
    rec = dict(num_moves=n, result=L.or_player_result(p), result_string=L.or_player_result_string(p),
               moves=np.array([pos.recent_move[k] for k in range(pos.recent_len)], np.int16),
               pis=np.stack([orc.node_arr(L.or_player_search_pi(p, k), A).copy() for k in range(n)]) if n else None,
               qs=np.array([L.or_player_q(p, k) for k in range(n)], np.float32),
               evals=L.or_player_evals(p))
    L.or_player_free(p)
    return rec


def run_engine(N, net, readouts, seed, games, slots, max_steps=200000, **cfg):
    sim = hs.Sim(board_size=N, games=slots, num_readouts=readouts, seed=seed, game_id_base=0, game_id_stride=1,
                 record_capacity_games=games + 8, **cfg)
    sim.start(games)
    steps = 0
    while sim.counters()["finished"] < games and steps < max_steps:
        sim.step(net.on_feats)
        steps += 1
    recs = sim.records()
    ct = sim.counters()
    sim.close()
    return recs, ct, steps


def bits_equal(a, b):
    if a is None or b is None:      # a game that resigned before its first move records nothing
        return (a is None or np.size(a) == 0) and (b is None or np.size(b) == 0)
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and (nan_a == nan_b).all() and (a[~nan_a] == b[~nan_b]).all()


def check_games(N, tower, readouts, seed, games, slots, **cfg):
    net = OracleNet(N, tower, seed=0)
    recs, ct, steps = run_engine(N, net, readouts, seed, games, slots, **cfg)
    assert len(recs) == games and ct["pool_exhausted"] == 0
    total_moves = total_evals = resigned = 0
    for r in recs:
        o = oracle_game(N, net, readouts, seed, int(r["game_id"]), cfg.get("resign_threshold", -0.9),
                        cfg.get("resign_disable_fraction", 0.05))
        assert r["num_moves"] == o["num_moves"], r["game_id"]
        assert (r["moves"] == o["moves"][: r["num_moves"]]).all()
        assert r["result"] == o["result"]
        assert r["was_resign"] == (o["result_string"] in (b"B+R", b"W+R"))
        assert bits_equal(r["qs"], o["qs"])
        assert bits_equal(r["pis"], o["pis"])
        total_moves += o["num_moves"]
        total_evals += o["evals"]
        resigned += r["was_resign"]
    assert ct["positions"] == total_moves
    assert ct["evals"] == total_evals
    net.close()
    return dict(moves=total_moves, evals=total_evals, resigned=resigned, steps=steps)


def test_config0_5x5_tower1_r16_4games():
    """BASELINE.json configs[0]"""
    st = check_games(5, 1, 16, seed=1, games=4, slots=4)
    assert st["moves"] >= 4


def test_more_games_than_slots_recycles():
    st = check_games(5, 1, 16, seed=2, games=7, slots=3)
    assert st["moves"] > 7


def test_5x5_resign_paths():
    """a threshold that is easy to hit exercises should_resign / B+R / W+R and the 5 % disable coin"""
    st = check_games(5, 1, 16, seed=3, games=12, slots=5, resign_threshold=-0.05, resign_disable_fraction=0.5)
    assert st["resigned"] > 0


@pytest.mark.parametrize("seed", [4])
def test_9x9_tower1_r24(seed):
    st = check_games(9, 1, 24, seed=seed, games=2, slots=2)
    assert st["moves"] > 20
