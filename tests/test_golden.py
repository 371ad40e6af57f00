"""Committed fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the pinned
oracle on seeded inputs) checked two ways:

  * CPU (-m "not gpu"): the oracle still reproduces them bit for bit (drift guard), and so does
    the engine logic under the host wave simulator;
  * GPU (-m gpu): the HIP path, called through the C ABI, reproduces them -- bit-exact for rules,
    tree statistics, moves and search distributions; |d pi|, |d v| <= 1e-4 for the network
    (BASELINE.json north_star tolerance) against the float64 evaluation.

The reference itself (Julia) cannot be run in this image, so these are oracle outputs, not
reference outputs; what pins the oracle to the reference is tests/test_oracle_*.py."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from test_hostsim_selfplay import OracleNet, bits_equal, oracle_game, run_engine

L = orc.lib()
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4

SELFPLAY = ["selfplay_c1_5x5_t1_r16", "selfplay_5x5_t1_r16_resign", "selfplay_9x9_t1_r24"]
NN = ["nn_5x5_t1", "nn_9x9_t2", "nn_19x19_t1"]
GO = ["go_5x5", "go_9x9", "go_19x19"]


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def check_record(g, d, rec):
    """rec: a game record dict (oracle_game / engine records) against golden game g"""
    n = len(d[f"g{g}_moves"])
    assert rec["num_moves"] == n
    assert (np.asarray(rec["moves"][:n]) == d[f"g{g}_moves"]).all()
    assert bits_equal(rec["qs"], d[f"g{g}_qs"])
    assert bits_equal(rec["pis"], d[f"g{g}_pis"])
    assert rec["result"] == d[f"g{g}_result"][0]


# ------------------------------------------------------------------ CPU: oracle and host simulator

@pytest.mark.parametrize("name", SELFPLAY)
def test_oracle_reproduces_selfplay_golden(name):
    d = load(name)
    N, tower, R, seed = (int(x) for x in d["config"])
    thr, dis = (float(x) for x in d["resign"])
    net = OracleNet(N, tower, seed=0)
    for g in d["games"]:
        o = oracle_game(N, net, R, seed, int(g), thr, dis)
        check_record(int(g), d, o)
        assert o["evals"] == d[f"g{g}_result"][2]
        assert (o["result_string"] in (b"B+R", b"W+R")) == bool(d[f"g{g}_result"][1])
    net.close()


@pytest.mark.parametrize("name", SELFPLAY[:2])
def test_hostsim_reproduces_selfplay_golden(name):
    d = load(name)
    N, tower, R, seed = (int(x) for x in d["config"])
    thr, dis = (float(x) for x in d["resign"])
    net = OracleNet(N, tower, seed=0)
    games = len(d["games"])
    recs, ct, _ = run_engine(N, net, R, seed, games, min(games, 4), resign_threshold=thr, resign_disable_fraction=dis)
    assert sorted(int(r["game_id"]) for r in recs) == [int(g) for g in d["games"]]
    for r in recs:
        check_record(int(r["game_id"]), d, r)
        assert bool(r["was_resign"]) == bool(d[f"g{int(r['game_id'])}_result"][1])
    net.close()


@pytest.mark.parametrize("name", NN)
def test_oracle_reproduces_nn_golden(name):
    d = load(name)
    N, tower, _ = (int(x) for x in d["config"])
    net = L.or_net_new(N, tower)
    L.or_net_init_synthetic(net, 0)
    x = d["feats"].astype(np.float32)
    B, A = x.shape[0], N * N + 1
    for prec, tol in ((64, 1e-12), (32, 1e-6)):
        pi, v = np.zeros((B, A), np.float32), np.zeros(B, np.float32)
        L.or_net_forward_feats(net, orc.fptr(x), B, orc.fptr(pi), orc.fptr(v), prec)
        assert np.abs(pi - d[f"pi_f{prec}"]).max() <= tol and np.abs(v - d[f"v_f{prec}"]).max() <= tol
    # f32 and f64 evaluations agree far inside the 1e-4 tolerance the HIP path is held to
    assert np.abs(d["pi_f32"] - d["pi_f64"]).max() < 1e-5 and np.abs(d["v_f32"] - d["v_f64"]).max() < 1e-5
    L.or_net_free(net)


@pytest.mark.parametrize("name", GO)
def test_oracle_reproduces_go_golden(name):
    d = load(name)
    N = int(d["N"][0])
    for b in range(len(d["boards"])):
        pos = orc.make_pos(N, board=d["boards"][b], to_play=int(d["to_play"][b]), ko=int(d["ko"][b]))
        assert (orc.legal_moves(pos) == d["legal"][b]).all()
        assert np.float32(L.or_score(C.byref(pos))) == d["score"][b]
        rcode, nxt = orc.play(pos, int(d["move"][b]))
        assert (rcode != orc.OK) == bool(d["status"][b])
        if rcode == orc.OK:
            assert (nxt.board_np() == d["next_board"][b]).all() and nxt.ko == d["next_ko"][b]


def test_oracle_reproduces_tree_golden():
    from test_hostsim_tree import almost_done_net, send_two_return_one
    d = load("tree_dont_pass_if_losing")
    for seed in (0, 1):
        net = almost_done_net()
        p = L.or_player_new(9, net.cb, None, 800, 0, -0.9, seed, 0)
        L.or_player_initialize_game(p, C.byref(send_two_return_one()))
        for _ in range(20):
            L.or_player_tree_search(p, 8)
        root = L.or_player_root(p)
        assert bits_equal(orc.node_arr(L.or_node_child_N(root), 82), d[f"seed{seed}_child_N"])
        assert bits_equal(orc.node_arr(L.or_node_child_W(root), 82), d[f"seed{seed}_child_W"])
        assert int(np.argmax(d[f"seed{seed}_child_N"])) == orc.from_kgs("D9", 9)   # test_mcts_player.jl:155
        L.or_player_free(p)


# ------------------------------------------------------------------ GPU: the HIP path through the C ABI

@pytest.mark.gpu
@pytest.mark.parametrize("name", SELFPLAY)
def test_gpu_selfplay_matches_golden(name):
    import alphago_jl_amd as ag
    d = load(name)
    N, tower, R, seed = (int(x) for x in d["config"])
    thr, dis = (float(x) for x in d["resign"])
    games = len(d["games"])
    net = OracleNet(N, tower, seed=0)     # the CPU network the golden games were played with
    eng = ag.Engine(board_size=N, tower_height=0, games=min(games, 4), num_readouts=R, seed=seed, external_network=1,
                    record_capacity_games=games + 8, resign_threshold=thr, resign_disable_fraction=dis)
    eng.start(games)
    for _ in range(100000):
        eng.step_external(net.on_feats)
        if eng.stats()["games_finished"] >= games:
            break
    recs = eng.records()
    assert sorted(int(r["game_id"]) for r in recs) == [int(g) for g in d["games"]]
    for r in recs:
        check_record(int(r["game_id"]), d, r)
        assert bool(r["was_resign"]) == bool(d[f"g{int(r['game_id'])}_result"][1])
    assert eng.stats()["evals"] == sum(int(d[f"g{g}_result"][2]) for g in d["games"])
    eng.close()
    net.close()


@pytest.mark.gpu
@pytest.mark.parametrize("wino", [1, 0])
@pytest.mark.parametrize("name", NN)
def test_gpu_nn_matches_golden(name, wino):
    import alphago_jl_amd as ag
    d = load(name)
    N, tower, _ = (int(x) for x in d["config"])
    eng = ag.Engine(board_size=N, tower_height=tower, games=1, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(0)
    eng.set_winograd(wino)
    feats = eng.features(d["boards"], d["deltas"], d["ndeltas"], d["to_play"])
    assert (feats.reshape(len(d["boards"]), -1) == d["feats"]).all()          # integer work: exact
    pi, v = eng.forward(d["boards"], d["deltas"], d["ndeltas"], d["to_play"])
    assert np.abs(pi - d["pi_f64"]).max() <= TOL and np.abs(v - d["v_f64"]).max() <= TOL
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", GO)
def test_gpu_go_matches_golden(name):
    import alphago_jl_amd as ag
    d = load(name)
    N = int(d["N"][0])
    eng = ag.Engine(board_size=N, games=1, tower_height=0, num_readouts=8, max_nodes_per_game=16)
    assert (eng.go_legal(d["boards"], d["to_play"], d["ko"]) == d["legal"]).all()
    assert (eng.go_score(d["boards"], np.full(len(d["boards"]), 7.5, np.float32)) == d["score"]).all()
    bo, ko, nc, st = eng.go_play(d["boards"], d["to_play"], d["ko"], d["move"])
    assert (st == d["status"]).all()
    ok = d["status"] == 0
    assert (bo[ok] == d["next_board"][ok]).all() and (ko[ok] == d["next_ko"][ok]).all() and (nc[ok] == d["captured"][ok]).all()
    assert (bo[~ok] == d["boards"][~ok]).all()
    eng.close()


@pytest.mark.gpu
def test_gpu_tree_matches_golden():
    import alphago_jl_amd as ag
    from test_hostsim_tree import almost_done_net, send_two_return_one
    d = load("tree_dont_pass_if_losing")
    for seed in (0, 1):
        net, pos = almost_done_net(), send_two_return_one()
        eng = ag.Engine(board_size=9, games=1, tower_height=0, num_readouts=800, seed=seed, max_nodes_per_game=4096,
                        external_network=1)
        eng.tree_init(0, pos.board_np(), n=pos.n, to_play=pos.to_play, ko=pos.ko, caps=tuple(pos.caps),
                      last_move=pos.recent_move[pos.recent_len - 1], komi=pos.komi)
        eng.set_draw(0, 0, 0)
        fn = lambda feats: (np.tile(net.priors, (feats.shape[0], 1)).astype(np.float32),
                            np.full(feats.shape[0], net.value, np.float32))
        for _ in range(20):
            eng.tree_search(0, 8, network=fn)
        root = eng.tree_root(0)
        assert bits_equal(eng.node_floats(0, root, 0), d[f"seed{seed}_child_N"])
        assert bits_equal(eng.node_floats(0, root, 1), d[f"seed{seed}_child_W"])
        eng.close()
