#!/usr/bin/env python3
"""Regenerates the committed fixtures in tests/golden/.

The reference (tejank10/AlphaGo.jl) is Julia and cannot be imported or run in this image, so these
vectors are NOT reference outputs: they are outputs of the pinned CPU oracle (oracle/, itself
checked against every known answer of the reference's own tests -- tests/test_oracle_*.py) on
seeded inputs, frozen so that (a) the oracle cannot drift silently and (b) the HIP path can be
checked on the GPU box against numbers that were produced in a different process on a different
machine.  Data only: inputs and expected outputs.

  python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc  # noqa: E402
from gpu_common import pos_soa  # noqa: E402
from test_hostsim_go import random_positions  # noqa: E402
from test_hostsim_selfplay import OracleNet, oracle_game  # noqa: E402
from test_oracle_go import ALMOST_DONE  # noqa: E402

L = orc.lib()

SELFPLAY = {  # name -> (N, tower, readouts, seed, game ids, resign threshold, disable fraction)
    "selfplay_c1_5x5_t1_r16": (5, 1, 16, 1, [0, 1, 2, 3], -0.9, 0.05),          # BASELINE.json configs[0]
    "selfplay_5x5_t1_r16_resign": (5, 1, 16, 3, list(range(8)), -0.05, 0.5),
    "selfplay_9x9_t1_r24": (9, 1, 24, 4, [0, 1], -0.9, 0.05),
}
NN = {"nn_5x5_t1": (5, 1, 12), "nn_9x9_t2": (9, 2, 12), "nn_19x19_t1": (19, 1, 4)}
GO = {"go_5x5": (5, 40, 40), "go_9x9": (9, 24, 140), "go_19x19": (19, 4, 420)}


def selfplay():
    for name, (N, tower, R, seed, games, thr, dis) in SELFPLAY.items():
        net = OracleNet(N, tower, seed=0)
        out = {"config": np.array([N, tower, R, seed], np.int64), "games": np.array(games, np.int64),
               "resign": np.array([thr, dis], np.float64)}
        for g in games:
            o = oracle_game(N, net, R, seed, g, thr, dis)
            n = o["num_moves"]
            out[f"g{g}_moves"] = o["moves"][:n]
            out[f"g{g}_pis"] = o["pis"] if n else np.zeros((0, N * N + 1), np.float32)
            out[f"g{g}_qs"] = o["qs"]
            out[f"g{g}_result"] = np.array([o["result"], o["result_string"] in (b"B+R", b"W+R"), o["evals"]], np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        net.close()


def nn():
    for name, (N, tower, B) in NN.items():
        positions = random_positions(N, B, 3 * N, seed=7 + N)[:B]
        boards, deltas, nd, tp = pos_soa(positions)
        net = L.or_net_new(N, tower)
        L.or_net_init_synthetic(net, 0)
        A = N * N + 1
        x = np.stack([orc.feats(p).astype(np.float32).reshape(-1) for p in positions])
        pi64, v64 = np.zeros((B, A), np.float32), np.zeros(B, np.float32)
        pi32, v32 = np.zeros((B, A), np.float32), np.zeros(B, np.float32)
        L.or_net_forward_feats(net, orc.fptr(x), B, orc.fptr(pi64), orc.fptr(v64), 64)
        L.or_net_forward_feats(net, orc.fptr(x), B, orc.fptr(pi32), orc.fptr(v32), 32)
        L.or_net_free(net)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), config=np.array([N, tower, 0], np.int64), boards=boards,
                            deltas=deltas, ndeltas=nd, to_play=tp, feats=x.astype(np.int8), pi_f64=pi64, v_f64=v64,
                            pi_f32=pi32, v_f32=v32)


def go():
    for name, (N, games, moves) in GO.items():
        A = N * N + 1
        positions = random_positions(N, games, moves, seed=300 + N)
        B = len(positions)
        rng = np.random.RandomState(5)
        mv = rng.randint(0, A, size=B).astype(np.int32)
        boards = np.stack([p.board_np() for p in positions])
        legal = np.stack([orc.legal_moves(p) for p in positions])
        score = np.array([L.or_score(C.byref(p)) for p in positions], np.float32)
        nb, nko, ncap, st = boards.copy(), np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        for b, p in enumerate(positions):
            rcode, nxt = orc.play(p, int(mv[b]))
            st[b] = 0 if rcode == orc.OK else 1
            if rcode == orc.OK:
                nb[b], nko[b] = nxt.board_np(), nxt.ko
                ncap[b] = (nxt.caps[0] - p.caps[0]) + (nxt.caps[1] - p.caps[1])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), N=np.array([N]), boards=boards,
                            to_play=np.array([p.to_play for p in positions], np.int8),
                            ko=np.array([p.ko for p in positions], np.int32), legal=legal, score=score, move=mv,
                            next_board=nb, next_ko=nko, captured=ncap, status=st)


def tree():
    """root statistics after 20 x tree_search!(8) of the reference's `dont_pass_if_losing` scenario
    (test/test_mcts_player.jl:139-165) under draw seeds 0 and 1"""
    N, A = 9, 82
    from test_hostsim_tree import almost_done_net, send_two_return_one
    out = {}
    for seed in (0, 1):
        net = almost_done_net()
        p = L.or_player_new(N, net.cb, None, 800, 0, -0.9, seed, 0)
        L.or_player_initialize_game(p, C.byref(send_two_return_one()))
        for _ in range(20):
            L.or_player_tree_search(p, 8)
        root = L.or_player_root(p)
        out[f"seed{seed}_child_N"] = orc.node_arr(L.or_node_child_N(root), A).copy()
        out[f"seed{seed}_child_W"] = orc.node_arr(L.or_node_child_W(root), A).copy()
        L.or_player_free(p)
    np.savez_compressed(os.path.join(HERE, "tree_dont_pass_if_losing.npz"), **out)


class OracleNetSink:
    """duck-typed stand-in for Engine in bson_weights.apply_param_lists: pushes into an oracle net"""

    def __init__(self, N, tower):
        self.N, self.tower_height = N, tower
        self.net = L.or_net_new(N, tower)
        L.or_net_init_synthetic(self.net, 0)

    def set_weights(self, layer, kind, data):
        a = np.ascontiguousarray(np.asarray(data, np.float32).ravel())
        assert L.or_net_set(self.net, layer, kind, orc.fptr(a), a.size) == 0, (layer, kind)


def shipped(model_dir="/root/reference/models"):
    """the 9x9 / tower-0 network shipped with the reference (models/weights/agz_*.bson, BatchNorm
    statistics from the Flux <= 0.7 struct dumps models/agz_*.bson, whose 5th BatchNorm field is the moving
    STANDARD DEVIATION -- stored here as variance = sigma^2 with eps = 0, plus the raw field and eps
    as `*_field_*` / `*_fieldeps_*`), decoded by alphago.jl_amd.bson_weights:
    its parameters (data, 76 k floats) + float64 oracle outputs on seeded positions.  Exercises what
    synthetic weights cannot: non-identity BatchNorm folding, non-zero biases, Flux kernel flip."""
    import alphago_jl_amd as ag
    bw = ag.bson_weights
    ck = bw.read_checkpoint(model_dir)
    N, tower, B = 9, bw.tower_height_of(ck["base"]), 16
    sink = OracleNetSink(N, tower)
    bw.apply_param_lists(sink, ck["base"], ck["value"], ck["policy"], ck["base_stats"], ck["value_stats"], ck["policy_stats"])
    positions = random_positions(N, B, 60, seed=77)[:B]
    boards, deltas, nd, tp = pos_soa(positions)
    x = np.stack([orc.feats(p).astype(np.float32).reshape(-1) for p in positions])
    pi64, v64 = np.zeros((B, N * N + 1), np.float32), np.zeros(B, np.float32)
    L.or_net_forward_feats(sink.net, orc.fptr(x), B, orc.fptr(pi64), orc.fptr(v64), 64)
    out = dict(boards=boards, deltas=deltas, ndeltas=nd, to_play=tp, feats=x.astype(np.int8), pi_f64=pi64, v_f64=v64)
    for part in ("base", "value", "policy"):
        for i, a in enumerate(ck[part]):
            out[f"{part}_{i}"] = a
        raw = bw.read_batchnorm_stats(os.path.join(model_dir, f"agz_{part}.bson"), "var")
        for i, (m, v, e) in enumerate(ck[part + "_stats"]):
            out[f"{part}_mu_{i}"], out[f"{part}_var_{i}"], out[f"{part}_eps_{i}"] = m, v, np.array([e])
            out[f"{part}_field_{i}"], out[f"{part}_fieldeps_{i}"] = raw[i][1], np.array([raw[i][2]])
    np.savez_compressed(os.path.join(HERE, "shipped_9x9_t0.npz"), **out)


if __name__ == "__main__":
    selfplay(); nn(); go(); tree()
    if os.path.isdir("/root/reference/models"):
        shipped()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
