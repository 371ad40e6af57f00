"""Pins the oracle's search tree against the known answers of the reference's
test/test_mcts.jl and its features against test/test_features.jl.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import orc
from orc import BLACK, WHITE, from_kgs, load_board, make_pos, fptr
from test_oracle_go import ALMOST_DONE

N = 9
A = N * N + 1
L = orc.lib()
ENV = orc.env(N)
ALMOST_DONE_BOARD = load_board(ALMOST_DONE, N)


def send_two_return_one():  # test_mcts.jl:34-43
    return make_pos(N, board=ALMOST_DONE_BOARD, n=75, komi=0.5, caps=(0, 0),
                    recent=[(BLACK, orc.rc(1, 2, N)), (WHITE, orc.rc(1, 9, N)), (BLACK, orc.rc(2, 1, N))],
                    to_play=WHITE)


def new_root(pos):
    return L.or_node_new(C.byref(ENV), C.byref(pos))


def draw(seed=1, game=0, move=0):
    return orc.ODraw(seed, game, move, 0)


def incorporate(node, probs, value, up_to):
    p = np.asarray(probs, dtype=np.float32)
    return L.or_incorporate_results(C.byref(ENV), node, fptr(p), len(p), float(value), up_to)


def select(root, d):
    return L.or_select_leaf(C.byref(ENV), root, C.byref(d))


def test_env_constants():  # mcts.jl:15-25, mcts_play.jl:19 and SURVEY.md section 8 header
    for n, mgl, tau, alpha in ((5, 35, 2, 0.4165), (9, 113, 6, 0.13207), (19, 505, 30, 0.029917)):
        e = orc.env(n)
        assert e.max_game_length == mgl
        assert abs(e.dirichlet_alpha - alpha) < 1e-4
        p = L.or_player_new(n, orc.NET_FN(lambda *a: None), None, 8, 0, -0.9, 0, 0)
        assert L.or_player_tau_threshold(p) == tau
        L.or_player_free(p)


def test_action_flipping():  # test_mcts.jl:45-59
    rng = np.random.RandomState(1)
    probs = 0.02 * np.ones(A) + rng.rand(A) * 0.001
    d = draw()
    black_root = new_root(make_pos(N))
    white_root = new_root(make_pos(N, to_play=WHITE))
    incorporate(select(black_root, d), probs, 0, black_root)
    incorporate(select(white_root, d), probs, 0, white_root)
    bl = select(black_root, d)
    wl = select(white_root, d)
    assert L.or_node_fmove(bl) == L.or_node_fmove(wl)
    sb = np.zeros(A)
    sw = np.zeros(A)
    L.or_child_action_score(C.byref(ENV), black_root, sb.ctypes.data_as(C.POINTER(C.c_double)))
    L.or_child_action_score(C.byref(ENV), white_root, sw.ctypes.data_as(C.POINTER(C.c_double)))
    assert (sb == sw).all()
    L.or_node_free_tree(black_root)
    L.or_node_free_tree(white_root)


def test_select_leaf():  # test_mcts.jl:61-70
    flat = from_kgs("D9", N)
    probs = 0.02 * np.ones(A)
    probs[flat] = 0.4
    root = new_root(send_two_return_one())
    d = draw()
    incorporate(select(root, d), probs, 0, root)
    assert L.or_node_pos(root).contents.to_play == WHITE
    leaf = select(root, d)
    assert leaf == L.or_node_child(root, flat)
    L.or_node_free_tree(root)


def test_backup_incorporate_results():  # test_mcts.jl:72-114
    probs = 0.02 * np.ones(A)
    root = new_root(send_two_return_one())
    d = draw()
    incorporate(select(root, d), probs, 0, root)
    leaf = select(root, d)
    incorporate(leaf, probs, -1, root)  # white wins!
    assert L.or_node_N(root) == 2
    assert L.or_node_Q(root) == pytest.approx(-1 / 3)
    fm = L.or_node_fmove(leaf)
    cn = orc.node_arr(L.or_node_child_N(root), A)
    cw = orc.node_arr(L.or_node_child_W(root), A)
    assert cn[fm] == 1
    assert L.or_node_N(leaf) == 1
    assert cw[fm] / (1 + cn[fm]) == -0.5
    assert L.or_node_Q(leaf) == pytest.approx(-0.5)
    assert L.or_node_pos(root).contents.to_play == WHITE
    leaf2 = select(root, d)
    assert L.or_node_parent(leaf2) == leaf
    incorporate(leaf2, probs, -0.2, root)
    assert L.or_node_N(root) == 3
    assert L.or_node_Q(root) == pytest.approx(-0.3)
    assert L.or_node_N(leaf) == 2
    assert L.or_node_N(leaf2) == 1
    assert L.or_node_Q(leaf) == pytest.approx(cw[fm] / (1 + cn[fm]))
    assert L.or_node_Q(leaf) == pytest.approx(-0.4)
    lcw = orc.node_arr(L.or_node_child_W(leaf), A)
    lcn = orc.node_arr(L.or_node_child_N(leaf), A)
    f2 = L.or_node_fmove(leaf2)
    assert lcw[f2] / (1 + lcn[f2]) == pytest.approx(-0.6)
    assert L.or_node_Q(leaf2) == pytest.approx(-0.6)
    L.or_node_free_tree(root)


def test_do_not_explore_past_finish():  # test_mcts.jl:116-127
    probs = 0.02 * np.ones(A, dtype=np.float32)
    root = new_root(make_pos(N))
    d = draw()
    incorporate(select(root, d), probs, 0, root)
    first_pass = C.c_void_p()
    assert L.or_maybe_add_child(C.byref(ENV), root, N * N, C.byref(first_pass)) == orc.OK
    incorporate(first_pass, probs, 0, root)
    second_pass = C.c_void_p()
    assert L.or_maybe_add_child(C.byref(ENV), first_pass, N * N, C.byref(second_pass)) == orc.OK
    assert incorporate(second_pass, probs, 0, root) == orc.ASSERT_DONE_NODE
    node = select(second_pass, d)
    assert node == second_pass.value
    L.or_node_free_tree(root)


def test_add_child_and_idempotency():  # test_mcts.jl:129-144 (1-based 17 -> 0-based 16)
    root = new_root(make_pos(N))
    child = C.c_void_p()
    L.or_maybe_add_child(C.byref(ENV), root, 16, C.byref(child))
    assert L.or_node_child(root, 16) == child.value
    assert L.or_node_parent(child) == root
    assert L.or_node_fmove(child) == 16
    child2 = C.c_void_p()
    L.or_maybe_add_child(C.byref(ENV), root, 16, C.byref(child2))
    assert child.value == child2.value
    assert L.or_tree_count_nodes(root) == 2
    L.or_node_free_tree(root)


def test_never_select_illegal_moves():  # test_mcts.jl:146-167 (1-based flat 2 -> 0-based 1)
    probs = 0.02 * np.ones(A)
    probs[1] = 0.99
    root = new_root(send_two_return_one())
    incorporate(root, probs, 0, root)
    L.or_node_set_N(root, 10000.0)
    legal = orc.legal_moves(L.or_node_pos(root).contents).astype(bool)
    cn = orc.node_arr(L.or_node_child_N(root), A)
    cn[legal] = 10000
    d = draw()
    leaf = select(root, d)
    assert L.or_node_fmove(leaf) != 1
    for i in range(10):
        dd = orc.ODraw(7, 3, i, 0)
        L.or_inject_noise(C.byref(ENV), root, C.byref(dd))
        leaf = select(root, d)
        assert L.or_node_fmove(leaf) != 1
    L.or_node_free_tree(root)


def test_dont_pick_unexpanded_child():  # test_mcts.jl:169-183 (1-based 18 -> 0-based 17)
    probs = 0.02 * np.ones(A)
    probs[17] = 0.999
    root = new_root(make_pos(N))
    incorporate(root, probs, 0, root)
    d = draw()
    leaf1 = select(root, d)
    assert L.or_node_fmove(leaf1) == 17
    L.or_add_virtual_loss(leaf1, root)
    leaf2 = select(root, d)
    assert leaf1 == leaf2
    L.or_node_free_tree(root)


def test_tie_break_uses_draw_stream():
    """Uniform priors => all 82 children tie at the root; the pick must be the draw-stream
    index among the legal moves, reproducibly."""
    probs = np.ones(A) / A
    picks = set()
    for sel in range(20):
        root = new_root(make_pos(N))
        incorporate(root, probs, 0, root)
        d = orc.ODraw(5, 9, 0, sel)
        leaf = select(root, d)
        fm = L.or_node_fmove(leaf)
        picks.add(fm)
        root2 = new_root(make_pos(N))
        incorporate(root2, probs, 0, root2)
        d2 = orc.ODraw(5, 9, 0, sel)
        assert L.or_node_fmove(select(root2, d2)) == fm
        L.or_node_free_tree(root)
        L.or_node_free_tree(root2)
    assert len(picks) > 5


# ------------------------------------------------------------------ features

EMPTY_ROW = "." * N + "\n"


def test_stone_features():  # test_features.jl:39-79
    pos = make_pos(N)
    for (r, c) in ((1, 1), (1, 2), (1, 3), (1, 4), (2, 2)):
        rcode, pos = orc.play(pos, orc.rc(r, c, N))
        assert rcode == orc.OK
    assert pos.to_play == WHITE
    f = orc.feats(pos)
    assert f.shape == (17, 81)
    exp = [
        "...X.....\n.........\n",
        "X.X......\n.X.......\n",
        ".X.X.....\n.........\n",
        "X.X......\n.........\n",
        ".X.......\n.........\n",
        "X.X......\n.........\n",
    ]
    for k, txt in enumerate(exp):
        assert (f[k] == load_board(txt + EMPTY_ROW * 7, N)).all(), k
    for k in range(10, 16):
        assert (f[k] == 0).all()
    # plane 17 is the colour to play as +1/-1 (features.jl:22), not pinned by the reference
    assert (f[16] == -1).all()


def test_features_repeat_oldest_board():
    """features.jl:14 -- a position built from a bare board has no deltas, so all eight
    history slots repeat the current board."""
    b = load_board(".X.....OO\nX........\n" + EMPTY_ROW * 7, N)
    pos = make_pos(N, board=b, n=3, to_play=BLACK)
    f = orc.feats(pos)
    for k in range(8):
        assert (f[2 * k] == (b == 1)).all()
        assert (f[2 * k + 1] == (b == -1)).all()
    assert (f[16] == 1).all()
    # one move later: slot 0 is the new board, slots 1..7 the old one
    _, p2 = orc.play(pos, orc.rc(5, 5, N))
    f2 = orc.feats(p2)
    nb = p2.board_np()
    assert (f2[0] == (nb == -1)).all() and (f2[1] == (nb == 1)).all()
    for k in range(1, 8):
        assert (f2[2 * k] == (b == -1)).all()
        assert (f2[2 * k + 1] == (b == 1)).all()


def test_features_capture_history():
    """deltas carry +color where an opponent stone vanished (board.jl:479-481) so that
    B_prev = B - delta restores captured stones."""
    sb = load_board(EMPTY_ROW * 5 + "XXXX.....\nXOOX.....\nO.OX.....\nOOXX.....\n", N)
    pos = make_pos(N, board=sb, to_play=BLACK)
    _, p2 = orc.play(pos, from_kgs("B2", N))
    f = orc.feats(p2)   # white to play
    assert (f[2] == (sb == -1)).all()   # previous board, white stones incl. the six captured
    assert (f[3] == (sb == 1)).all()
