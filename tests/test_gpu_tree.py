"""HIP search kernels, one reference-style call at a time through the C ABI, against the oracle:
after every operation the two trees must be identical bit for bit.  Same scenarios as
tests/test_hostsim_tree.py (which are the reference's own test_mcts.jl / test_mcts_player.jl)."""
import ctypes as C

import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from orc import BLACK, WHITE, load_board
from test_oracle_go import ALMOST_DONE, TT_FTW

pytestmark = pytest.mark.gpu
N = 9
P = N * N
A = P + 1
L = orc.lib()
ENV = orc.env(N)


def compare_trees(eng, g, snode, onode, depth=0):
    opos = L.or_node_pos(onode).contents
    info = eng.node_info(g, snode)
    assert (info.pos.n, info.pos.to_play, info.pos.ko) == (opos.n, opos.to_play, opos.ko)
    assert (info.pos.caps_black, info.pos.caps_white) == tuple(opos.caps)
    assert bool(info.is_expanded) == bool(L.or_node_is_expanded(onode))
    assert info.losses_applied == L.or_node_losses_applied(onode)
    assert bool(info.done) == bool(opos.done)
    assert (eng.node_board(g, snode) == opos.board_np()).all()
    assert np.float32(info.N) == np.float32(L.or_node_N(onode))
    assert np.float32(info.W) == np.float32(L.or_node_W(onode))
    for field, getter in ((0, L.or_node_child_N), (1, L.or_node_child_W), (2, L.or_node_child_prior)):
        a = eng.node_floats(g, snode, field)
        b = orc.node_arr(getter(onode), A)
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), (field, depth)
    ch = eng.node_children(g, snode)
    count = 1
    for a in range(A):
        oc = L.or_node_child(onode, a)
        assert (ch[a] >= 0) == bool(oc), (a, depth)
        if oc:
            count += compare_trees(eng, g, int(ch[a]), oc, depth + 1)
    return count


class Twin:
    def __init__(self, pos, net, seed=11, game=0, par=8, readouts=800, resign=-0.9):
        self.net = net
        self.eng = ag.Engine(board_size=pos.N, games=1, tower_height=0, num_readouts=readouts,
                             parallel_readouts=max(par, 8), seed=seed, resign_threshold=resign,
                             max_nodes_per_game=4096, external_network=1)
        self.op = L.or_player_new(pos.N, net.cb, None, readouts, 0, resign, seed, game)
        L.or_player_initialize_game(self.op, C.byref(pos))
        last = pos.recent_move[pos.recent_len - 1] if 0 < pos.recent_len <= orc.MAXRECENT else -1
        self.eng.tree_init(0, pos.board_np(), n=pos.n, to_play=pos.to_play, ko=pos.ko, caps=tuple(pos.caps),
                           last_move=last, komi=pos.komi)
        self.eng.set_draw(0, game, 0)

    @property
    def oroot(self):
        return L.or_player_root(self.op)

    def check(self):
        return compare_trees(self.eng, 0, self.eng.tree_root(0), self.oroot)

    def _dummy(self, feats):
        n = feats.shape[0]
        return np.tile(self.net.priors, (n, 1)).astype(np.float32), np.full(n, self.net.value, np.float32)

    def tree_search(self, par=8):
        no = L.or_player_tree_search(self.op, par)
        ns = self.eng.tree_search(0, par, network=self._dummy)
        assert ns == no
        return ns

    def play(self, a):
        ro = L.or_player_play_move(self.op, a)
        rs = self.eng.play_move(0, a)
        assert rs == ro
        return rs

    def pick(self):
        a = C.c_int()
        so = L.or_player_pick_move(self.op, C.byref(a))
        st, rs = self.eng.pick_move(0)
        assert st == so
        if so == 0:
            assert rs == a.value
        return rs

    def close(self):
        L.or_player_free(self.op)
        self.eng.close()


def send_two_return_one():
    return orc.make_pos(N, board=load_board(ALMOST_DONE, N), n=70, komi=2.5, caps=(1, 4),
                        recent=[(BLACK, orc.rc(1, 2, N)), (WHITE, orc.rc(1, 9, N))], to_play=BLACK)


def almost_done_net():
    probs = np.ones(A) * 0.001
    probs[2:5] = 0.2
    probs[-1] = 0.2
    return orc.DummyNet(A, fake_priors=probs)


@pytest.mark.parametrize("seed", [0, 1])
def test_dont_pass_if_losing_twin(seed):  # test_mcts_player.jl:139-165
    t = Twin(send_two_return_one(), almost_done_net(), seed=seed)
    for _ in range(20):
        t.tree_search(8)
    assert t.check() > 20
    root = t.eng.tree_root(0)
    assert int(np.argmax(t.eng.node_floats(0, root, 0))) == orc.from_kgs("D9", N)
    assert t.eng.pending_vlosses(0) == 0
    t.close()


@pytest.mark.parametrize("par", [1, 10, 50])
def test_parallel_tree_search_twin(par):  # test_mcts_player.jl:167-202
    t = Twin(send_two_return_one(), almost_done_net(), par=par)
    for _ in range(8):
        t.tree_search(par)
    t.check()
    assert t.eng.pending_vlosses(0) == 0
    t.close()


def test_cold_start_twin():  # test_mcts_player.jl:227-240
    t = Twin(orc.make_pos(N), orc.DummyNet(A, fake_value=0.17))
    t.tree_search(4)
    t.check()
    info = t.eng.node_info(0, t.eng.tree_root(0))
    assert info.N == 1 and info.Q == pytest.approx(0.085)
    t.close()


def test_uniform_priors_tie_breaks_twin():
    t = Twin(orc.make_pos(N), orc.DummyNet(A), seed=5, game=3)
    for _ in range(30):
        t.tree_search(8)
    assert t.check() > 100
    t.close()


def test_long_game_failsafe_and_pass_first_twin():  # test_mcts_player.jl:204-283
    endgame = orc.make_pos(N, board=load_board(TT_FTW, N), n=ENV.max_game_length - 2, komi=2.5,
                           recent=[(BLACK, orc.rc(1, 2, N)), (WHITE, orc.rc(1, 9, N))], to_play=BLACK)
    t = Twin(endgame, orc.DummyNet(A))
    for _ in range(10):
        t.tree_search(8)
    t.check()
    assert t.eng.node_info(0, t.eng.tree_root(0)).Q > 0
    t.close()
    pos = orc.make_pos(N)
    for a in (orc.rc(4, 4, N), orc.rc(4, 5, N), orc.rc(5, 4, N), P):
        _, pos = orc.play(pos, a)
    t = Twin(pos, orc.DummyNet(A))
    for _ in range(16):
        t.tree_search(8)
    assert t.eng.node_floats(0, t.eng.tree_root(0), 0)[P] == 1
    t.check()
    t.close()


def test_play_pick_noise_and_reroot_twin():
    t = Twin(orc.make_pos(N), orc.DummyNet(A), seed=21, game=7)
    for move in range(10):
        for _ in range(6):
            t.tree_search(8)
        d = orc.ODraw(21, 7, L.or_node_pos(t.oroot).contents.n, 0)
        L.or_inject_noise(C.byref(ENV), t.oroot, C.byref(d))
        t.eng.inject_noise(0, t.eng.tree_root(0))
        t.check()
        a = t.pick()
        assert t.play(a) == 1
        t.check()
        assert t.eng.should_resign(0) == L.or_player_should_resign(t.op)
    t.close()


def test_node_level_reference_scenarios():
    """test_mcts.jl:72-183 driven through agz_tree_* alone"""
    eng = ag.Engine(board_size=N, games=1, tower_height=0, num_readouts=8, max_nodes_per_game=256)
    pos = orc.make_pos(N, board=load_board(ALMOST_DONE, N), n=75, komi=0.5,
                       recent=[(BLACK, orc.rc(1, 2, N)), (WHITE, orc.rc(1, 9, N)), (BLACK, orc.rc(2, 1, N))],
                       to_play=WHITE)
    root = eng.tree_init(0, pos.board_np(), n=75, to_play=WHITE, komi=0.5, last_move=orc.rc(2, 1, N))
    probs = np.full(A, 0.02, np.float32)
    assert eng.incorporate_results(0, eng.select_leaf(0, root), probs, 0, root) == 0
    leaf = eng.select_leaf(0, root)
    assert eng.incorporate_results(0, leaf, probs, -1, root) == 0
    ri, li = eng.node_info(0, root), eng.node_info(0, leaf)
    assert ri.N == 2 and ri.Q == pytest.approx(-1 / 3) and li.N == 1 and li.Q == pytest.approx(-0.5)
    leaf2 = eng.select_leaf(0, root)
    assert eng.node_info(0, leaf2).parent == leaf
    assert eng.incorporate_results(0, leaf2, probs, -0.2, root) == 0
    assert eng.node_info(0, root).Q == pytest.approx(-0.3)
    assert eng.node_info(0, leaf).Q == pytest.approx(-0.4)
    assert eng.node_info(0, leaf2).Q == pytest.approx(-0.6)
    # wrong-length priors: @assert size(move_probs) == (A,)
    assert eng.incorporate_results(0, leaf2, probs[:-1], 0, root) == ag._lib.BAD_SHAPE
    # done node: AssertionError; select_leaf stops at the end position (test_mcts.jl:116-127)
    root = eng.tree_init(0, np.zeros(P, np.int8))
    eng.incorporate_results(0, eng.select_leaf(0, root), probs, 0, root)
    p1 = eng.maybe_add_child(0, root, P)
    eng.incorporate_results(0, p1, probs, 0, root)
    p2 = eng.maybe_add_child(0, p1, P)
    assert eng.incorporate_results(0, p2, probs, 0, root) == ag._lib.ASSERT_DONE_NODE
    assert eng.select_leaf(0, p2) == p2
    assert eng.is_done(0, p2) == 1
    # virtual loss keeps the search on the unexpanded favourite (test_mcts.jl:169-183)
    root = eng.tree_init(0, np.zeros(P, np.int8))
    pr = np.full(A, 0.02, np.float32)
    pr[17] = 0.999
    eng.incorporate_results(0, root, pr, 0, root)
    leaf1 = eng.select_leaf(0, root)
    assert eng.node_info(0, leaf1).fmove == 17
    eng.add_virtual_loss(0, leaf1, root)
    assert eng.select_leaf(0, root) == leaf1
    # illegal move through maybe_add_child raises IllegalMove
    with pytest.raises(ag.IllegalMove):
        root = eng.tree_init(0, load_board(ALMOST_DONE, N))
        eng.maybe_add_child(0, root, 1)
    eng.close()
