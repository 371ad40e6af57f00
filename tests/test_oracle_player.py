"""Pins the oracle's MCTSPlayer / selfplay against the known answers of the reference's
test/test_mcts_player.jl.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import orc
from orc import BLACK, WHITE, DummyNet, from_kgs, load_board, make_pos
from test_oracle_go import ALMOST_DONE, TT_FTW

N = 9
A = N * N + 1
L = orc.lib()
ENV = orc.env(N)
PASS = N * N


def send_two_return_one():  # test_mcts_player.jl:48-57
    return make_pos(N, board=load_board(ALMOST_DONE, N), n=70, komi=2.5, caps=(1, 4),
                    recent=[(BLACK, orc.rc(1, 2, N)), (WHITE, orc.rc(1, 9, N))], to_play=BLACK)


def new_player(net, **kw):
    return L.or_player_new(N, net.cb, None, kw.get("num_readouts", 800), kw.get("two_player_mode", 0),
                           kw.get("resign_threshold", -0.9), kw.get("seed", 11), kw.get("game", 0))


def basic_player():  # test_mcts_player.jl:59-66
    net = DummyNet(A)
    p = new_player(net)
    L.or_player_initialize_game(p, None)
    root = L.or_player_root(p)
    d = orc.ODraw(11, 0, 0, 0)
    first = L.or_select_leaf(C.byref(ENV), root, C.byref(d))
    L.or_incorporate_results(C.byref(ENV), first, orc.fptr(net.priors), A, float(net.value), root)
    return p, net


def almost_done_player(seed=11):  # test_mcts_player.jl:68-77 (1-based 3:5 -> 0-based 2:5)
    probs = np.ones(A) * 0.001
    probs[2:5] = 0.2
    probs[-1] = 0.2
    net = DummyNet(A, fake_priors=probs)
    p = new_player(net, seed=seed)
    pos = send_two_return_one()
    L.or_player_initialize_game(p, C.byref(pos))
    return p, net


def test_inject_noise():  # test_mcts_player.jl:93-109
    p, net = basic_player()
    root = L.or_player_root(p)
    prior = orc.node_arr(L.or_node_child_prior(root), A)
    s0 = prior.sum()
    assert s0 == pytest.approx(1)
    cas = np.zeros(A)
    L.or_child_action_score(C.byref(ENV), root, cas.ctypes.data_as(C.POINTER(C.c_double)))
    # uniform priors, value 0 => Q = 0 and U identical everywhere
    assert (cas == cas[0]).all()
    d = orc.ODraw(11, 0, 0, 0)
    L.or_inject_noise(C.byref(ENV), root, C.byref(d))
    assert prior.sum() == pytest.approx(s0, abs=1e-5)
    assert prior.max() > 3 / A
    L.or_player_free(p)


def test_pick_moves():  # test_mcts_player.jl:111-137
    p, net = basic_player()
    root = L.or_player_root(p)
    cn = orc.node_arr(L.or_node_child_N(root), A)
    cn[orc.rc(3, 1, N)] = 10
    cn[orc.rc(2, 1, N)] = 5
    cn[orc.rc(4, 1, N)] = 1
    L.or_node_pos_mut(root).contents.n = A   # endgame
    assert L.or_node_pos(root).contents.n > L.or_player_tau_threshold(p)
    a = C.c_int()
    assert L.or_player_pick_move(p, C.byref(a)) == orc.OK
    assert a.value == orc.rc(3, 1, N)
    # early game: soft pick proportional to visits; pass excluded by construction
    L.or_node_pos_mut(root).contents.n = 3
    assert L.or_node_pos(root).contents.n <= L.or_player_tau_threshold(p)
    assert L.or_player_pick_move(p, C.byref(a)) == orc.OK
    assert a.value in (orc.rc(3, 1, N), orc.rc(2, 1, N), orc.rc(4, 1, N))
    L.or_player_free(p)


def test_soft_pick_distribution():
    """mcts_play.jl:63-67 completed from the minigo TODO at test_mcts_player.jl:129-136:
    u = .5 must give (3,1)... in this column-major indexing the cdf walks (2,1)=5 first."""
    counts = {}
    for g in range(300):
        net = DummyNet(A)
        p = new_player(net, seed=3, game=g)
        L.or_player_initialize_game(p, None)
        root = L.or_player_root(p)
        cn = orc.node_arr(L.or_node_child_N(root), A)
        cn[orc.rc(3, 1, N)] = 10
        cn[orc.rc(2, 1, N)] = 5
        cn[orc.rc(4, 1, N)] = 1
        L.or_node_pos_mut(root).contents.n = 3
        a = C.c_int()
        assert L.or_player_pick_move(p, C.byref(a)) == orc.OK
        counts[a.value] = counts.get(a.value, 0) + 1
        L.or_player_free(p)
    assert set(counts) <= {orc.rc(3, 1, N), orc.rc(2, 1, N), orc.rc(4, 1, N)}
    assert counts[orc.rc(3, 1, N)] > counts[orc.rc(2, 1, N)] > counts.get(orc.rc(4, 1, N), 0)


@pytest.mark.parametrize("seed", range(6))
def test_dont_pass_if_losing(seed):  # test_mcts_player.jl:139-165
    p, net = almost_done_player(seed)
    root = L.or_player_root(p)
    assert L.or_score(L.or_node_pos(root)) == -0.5
    for _ in range(20):
        L.or_player_tree_search(p, 8)
    flat = from_kgs("D9", N)
    cn = orc.node_arr(L.or_node_child_N(root), A)
    cw = orc.node_arr(L.or_node_child_W(root), A)
    assert int(np.argmax(cn)) == flat
    assert L.or_node_Q(L.or_node_child(root, flat)) > 0
    assert L.or_node_N(root) >= 20
    assert cw[-1] / (1 + cn[-1]) < 0
    assert L.or_tree_pending_vlosses(root) == 0
    L.or_player_free(p)


def test_parallel_tree_search():  # test_mcts_player.jl:167-191
    p, net = almost_done_player()
    root = L.or_player_root(p)
    L.or_player_tree_search(p, 1)
    for _ in range(6):
        L.or_player_tree_search(p, 10)
    flat = from_kgs("D9", N)
    cn = orc.node_arr(L.or_node_child_N(root), A)
    assert cn[flat] == cn.max()
    assert L.or_node_Q(L.or_node_child(root, flat)) > 0
    assert L.or_node_N(root) >= 20
    assert L.or_tree_pending_vlosses(root) == 0
    L.or_player_free(p)


def test_ridiculously_parallel_tree_search():  # test_mcts_player.jl:193-202
    p, net = almost_done_player()
    for _ in range(10):
        L.or_player_tree_search(p, 50)
    assert L.or_tree_pending_vlosses(L.or_player_root(p)) == 0
    L.or_player_free(p)


def test_long_game_tree_search():  # test_mcts_player.jl:204-225
    net = DummyNet(A)
    p = new_player(net)
    endgame = make_pos(N, board=load_board(TT_FTW, N), n=ENV.max_game_length - 2, komi=2.5,
                       recent=[(BLACK, orc.rc(1, 2, N)), (WHITE, orc.rc(1, 9, N))], to_play=BLACK)
    L.or_player_initialize_game(p, C.byref(endgame))
    for _ in range(10):
        L.or_player_tree_search(p, 8)
    root = L.or_player_root(p)
    assert L.or_tree_pending_vlosses(root) == 0
    assert L.or_node_Q(root) > 0
    L.or_player_free(p)


def test_cold_start_parallel_tree_search():  # test_mcts_player.jl:227-240
    net = DummyNet(A, fake_value=0.17)
    p = new_player(net)
    L.or_player_initialize_game(p, None)
    root = L.or_player_root(p)
    assert L.or_node_N(root) == 0
    assert not L.or_node_is_expanded(root)
    L.or_player_tree_search(p, 4)
    assert L.or_tree_pending_vlosses(root) == 0
    assert L.or_node_N(root) == 1
    assert L.or_node_Q(root) == pytest.approx(0.085)
    assert net.positions == 4      # the network saw the same root four times
    L.or_player_free(p)


def test_tree_search_failsafe():  # test_mcts_player.jl:242-252
    probs = np.ones(A) * 0.001
    probs[-1] = 1
    net = DummyNet(A, fake_priors=probs)
    p = new_player(net)
    start = make_pos(N)
    passed = orc.OPos()
    L.or_pass_move(C.byref(start), C.byref(passed))
    L.or_player_initialize_game(p, C.byref(passed))
    L.or_player_tree_search(p, 1)
    assert L.or_tree_pending_vlosses(L.or_player_root(p)) == 0
    L.or_player_free(p)


def test_only_check_game_end_once():  # test_mcts_player.jl:254-283
    pos = make_pos(N)
    for a in (orc.rc(4, 4, N), orc.rc(4, 5, N), orc.rc(5, 4, N), PASS):
        rcode, pos = orc.play(pos, a)
        assert rcode == orc.OK
    net = DummyNet(A)
    p = new_player(net)
    L.or_player_initialize_game(p, C.byref(pos))
    for _ in range(15):
        L.or_player_tree_search(p, 8)
    root = L.or_player_root(p)
    cn = orc.node_arr(L.or_node_child_N(root), A)
    assert L.or_node_N(L.or_node_child(root, PASS)) == 1
    assert cn[PASS] == 1
    L.or_player_tree_search(p, 8)
    assert cn[PASS] == 1
    L.or_player_free(p)


def test_extract_data_normal_end():  # test_mcts_player.jl:285-301
    net = DummyNet(A)
    p = new_player(net)
    L.or_player_initialize_game(p, None)
    L.or_player_tree_search(p, 8)
    assert L.or_player_play_move(p, PASS) == 1
    L.or_player_tree_search(p, 8)
    assert L.or_player_play_move(p, PASS) == 1
    root = L.or_player_root(p)
    assert L.or_node_is_done(C.byref(ENV), root)
    L.or_player_set_result(p, L.or_result(L.or_node_pos(root)), 0)
    positions = (orc.OPos * 2)()
    pis = np.zeros(2 * A, np.float32)
    results = (C.c_int * 2)()
    n = L.or_player_extract_data(p, positions, orc.fptr(pis), results)
    assert n == 2
    assert results[0] == WHITE
    assert L.or_player_result_string(p) == b"W+7.5"
    assert positions[0].n == 0 and positions[1].n == 1 and positions[1].to_play == WHITE
    # one tree_search! on a fresh root only expands it: child_N is all zero and the reference's
    # children_as_pi (mcts.jl:241-252) yields 0/0 = NaN -- reproduced, not papered over
    assert np.isnan(pis[:A]).all()
    L.or_player_free(p)


def test_extract_data_resign_end():  # test_mcts_player.jl:303-321
    net = DummyNet(A)
    p = new_player(net)
    L.or_player_initialize_game(p, None)
    L.or_player_tree_search(p, 8)
    L.or_player_play_move(p, orc.rc(1, 1, N))
    L.or_player_tree_search(p, 8)
    L.or_player_play_move(p, PASS)
    L.or_player_tree_search(p, 8)
    root = L.or_player_root(p)
    assert L.or_result(L.or_node_pos(root)) == BLACK
    L.or_player_set_result(p, WHITE, 1)
    results = (C.c_int * 2)()
    assert L.or_player_extract_data(p, None, None, results) == 2
    assert results[0] == WHITE
    assert L.or_player_result_string(p) == b"W+R"
    L.or_player_free(p)


def test_play_move_illegal_returns_false():  # mcts_play.jl:39-46
    net = DummyNet(A)
    p = new_player(net)
    pos = make_pos(N, board=load_board(ALMOST_DONE, N), to_play=BLACK)
    L.or_player_initialize_game(p, C.byref(pos))
    L.or_player_tree_search(p, 8)
    assert L.or_player_play_move(p, 1) == 0       # occupied point
    assert L.or_player_num_moves(p) == 0 and L.or_player_nqs(p) == 0
    L.or_player_free(p)


def test_selfplay_5x5_plumbing():
    """BASELINE.json configs[0]: GoEnv(5), tower_height=1, 16 readouts, 4 games on the CPU."""
    n5 = 5
    net = L.or_net_new(n5, 1)
    L.or_net_init_synthetic(net, 0)
    cb = orc.NET_FN(lambda ctx, pos, B, pi, v: L.or_net_callable(net, pos, B, pi, v))
    total = 0
    for g in range(4):
        p = L.or_selfplay(n5, cb, None, 16, 1, g, 0)
        n = L.or_player_num_moves(p)
        assert 1 <= n <= 35
        assert L.or_player_result(p) in (-1, 1) or L.or_player_result_string(p) == b"DRAW"
        pis = np.zeros(n * 26, np.float32)
        res = (C.c_int * n)()
        root = L.or_player_root(p)
        if L.or_node_pos(root).contents.n == n:
            assert L.or_player_extract_data(p, None, orc.fptr(pis), res) == n
            assert np.allclose(pis.reshape(n, 26).sum(1), 1, atol=1e-5)
            assert all(r == L.or_player_result(p) for r in res)
        total += n
        # reproducible
        p2 = L.or_selfplay(n5, cb, None, 16, 1, g, 0)
        assert L.or_player_num_moves(p2) == n
        assert L.or_player_result_string(p2) == L.or_player_result_string(p)
        L.or_player_free(p)
        L.or_player_free(p2)
    assert total > 4
    L.or_net_free(net)


def test_softpick_assertion_fails_when_no_board_move_was_visited():
    """mcts_play.jl:63-67: below the temperature threshold the move is drawn from the visit counts of the BOARD moves
    (`cdf /= cdf[end - 1]`: "prevents passing via softpick") and `@assert child_N[fcoord] != 0`.  With visits on the pass
    only -- or none at all -- the normaliser is 0, the cdf is NaN (or 0/0), searchsortedfirst runs off the end and the
    assertion fails: the oracle reports OR_ASSERT_SOFTPICK (the reference would throw; its self-play loop would die, ours
    plays a pass: DESIGN.md 8, deviations).  A visited board move makes it succeed again."""
    net = DummyNet(A)
    p = new_player(net)
    L.or_player_initialize_game(p, None)
    L.or_player_tree_search(p, 8)                       # expands the root; n = 0 < tau_threshold
    assert L.or_player_tau_threshold(p) > 0
    root = L.or_player_root(p)
    cn = orc.node_arr(L.or_node_child_N(root), A)
    a = C.c_int(-7)
    cn[:] = 0
    assert L.or_player_pick_move(p, C.byref(a)) == orc.ASSERT_SOFTPICK
    cn[PASS] = 3
    assert L.or_player_pick_move(p, C.byref(a)) == orc.ASSERT_SOFTPICK
    cn[17] = 1
    assert L.or_player_pick_move(p, C.byref(a)) == orc.OK and a.value == 17      # the only board move with visits
    L.or_player_free(p)
