"""Engine search logic (agz_search.h under the host wave simulator) against the oracle, one
reference-style call at a time: after every operation the two trees must be identical bit for
bit -- visit counts, W, priors, expansion flags, virtual losses, boards, ko, legality.  The
scenarios are the reference's own (test/test_mcts.jl, test/test_mcts_player.jl).  CPU only;
tests/test_gpu_tree.py replays the same scenarios through the HIP kernels."""
import ctypes as C

import numpy as np
import pytest

import hs
import orc
from orc import BLACK, WHITE, load_board
from test_oracle_go import ALMOST_DONE, TT_FTW

N = 9
P = N * N
A = P + 1
L = orc.lib()
ENV = orc.env(N)


def compare_trees(sim, g, snode, onode, depth=0):
    """recursive bit-exact comparison; returns the number of nodes compared"""
    A_ = sim.A
    opos = L.or_node_pos(onode).contents
    m = sim.meta(g, snode)
    assert m.n == opos.n and m.to_play == opos.to_play and m.ko == opos.ko
    assert (m.caps_b, m.caps_w) == tuple(opos.caps)
    assert bool(m.flags & 1) == bool(L.or_node_is_expanded(onode))
    assert m.losses == L.or_node_losses_applied(onode)
    assert bool(m.flags & 2) == bool(opos.done)
    assert (sim.board(g, snode) == opos.board_np()).all()
    assert np.float32(sim.N_(g, snode)) == np.float32(L.or_node_N(onode))
    assert np.float32(sim.W_(g, snode)) == np.float32(L.or_node_W(onode))
    for field, getter in ((0, L.or_node_child_N), (1, L.or_node_child_W), (2, L.or_node_child_prior)):
        a = sim.row(g, snode, field)
        b = orc.node_arr(getter(onode), A_)
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), (field, depth)
    assert (sim.legal(g, snode) == orc.legal_moves(opos)).all()
    ch = sim.children(g, snode)
    count = 1
    for a in range(A_):
        oc = L.or_node_child(onode, a)
        assert (ch[a] >= 0) == bool(oc), (a, depth)
        if oc:
            count += compare_trees(sim, g, int(ch[a]), oc, depth + 1)
    return count


class Twin:
    """the same player driven through the oracle and through the engine logic"""

    def __init__(self, pos, net, seed=11, game=0, par=8, readouts=800, resign=-0.9):
        self.net = net
        self.sim = hs.Sim(board_size=pos.N, games=1, num_readouts=readouts, parallel_readouts=max(par, 8),
                          seed=seed, resign_threshold=resign, max_nodes_per_game=4096)
        self.op = L.or_player_new(pos.N, net.cb, None, readouts, 0, resign, seed, game)
        L.or_player_initialize_game(self.op, C.byref(pos))
        last = pos.recent_move[pos.recent_len - 1] if 0 < pos.recent_len <= orc.MAXRECENT else -1
        self.sroot = self.sim.tree_init(0, pos.board_np(), n=pos.n, to_play=pos.to_play, ko=pos.ko,
                                        caps=tuple(pos.caps), last_move=last, komi=pos.komi)
        self.sim.L.hs_game_set(self.sim.h, 0, 0, float(game))

    @property
    def oroot(self):
        return L.or_player_root(self.op)

    def check(self):
        return compare_trees(self.sim, 0, self.sim.game(0).root, self.oroot)

    def tree_search(self, par=8):
        no = L.or_player_tree_search(self.op, par)
        st, ns = self.sim.op(hs.TOP_SEARCH_SELECT, par=par)
        assert st == 0 and ns == no
        if ns:
            feats = np.zeros((ns, 17 * self.sim.P), np.float32)
            self.sim.L.hs_tree_leaf_features(self.sim.h, 0, hs.pf(feats))
            pi = np.tile(self.net.priors, (ns, 1)).astype(np.float32)
            v = np.full(ns, self.net.value, np.float32)
            self.sim.L.hs_set_batch_outputs(self.sim.h, hs.pf(pi), hs.pf(v), ns)
            st, _ = self.sim.op(hs.TOP_SEARCH_POST)
            assert st == 0
        return ns

    def play(self, a):
        ro = L.or_player_play_move(self.op, a)
        st, rs = self.sim.op(hs.TOP_PLAY, a=a)
        assert st == 0 and rs == ro
        return rs

    def pick(self):
        a = C.c_int()
        so = L.or_player_pick_move(self.op, C.byref(a))
        st, rs = self.sim.op(hs.TOP_PICK)
        assert st == so
        if so == 0:
            assert rs == a.value
        return rs

    def close(self):
        L.or_player_free(self.op)
        self.sim.close()


def send_two_return_one():
    return orc.make_pos(N, board=load_board(ALMOST_DONE, N), n=70, komi=2.5, caps=(1, 4),
                        recent=[(BLACK, orc.rc(1, 2, N)), (WHITE, orc.rc(1, 9, N))], to_play=BLACK)


def almost_done_net():
    probs = np.ones(A) * 0.001
    probs[2:5] = 0.2
    probs[-1] = 0.2
    return orc.DummyNet(A, fake_priors=probs)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_dont_pass_if_losing_twin(seed):  # test_mcts_player.jl:139-165
    t = Twin(send_two_return_one(), almost_done_net(), seed=seed)
    for _ in range(20):
        t.tree_search(8)
        t.check()
    root = t.sim.game(0).root
    assert int(np.argmax(t.sim.row(0, root, 0))) == orc.from_kgs("D9", N)
    st, pending = t.sim.op(hs.TOP_PENDING)
    assert pending == 0
    assert t.check() > 20
    t.close()


@pytest.mark.parametrize("par", [1, 4, 10, 50])
def test_parallel_tree_search_twin(par):  # test_mcts_player.jl:167-202
    t = Twin(send_two_return_one(), almost_done_net(), par=par)
    for _ in range(8):
        t.tree_search(par)
    t.check()
    assert t.sim.op(hs.TOP_PENDING)[1] == 0
    t.close()


def test_cold_start_twin():  # test_mcts_player.jl:227-240
    t = Twin(orc.make_pos(N), orc.DummyNet(A, fake_value=0.17))
    t.tree_search(4)
    t.check()
    root = t.sim.game(0).root
    assert t.sim.N_(0, root) == 1
    assert t.sim.Q_(0, root) == pytest.approx(0.085)
    t.close()


def test_uniform_priors_tie_breaks_twin():
    """DummyNet's uniform priors make every PUCT comparison a tie: exercises the draw stream"""
    t = Twin(orc.make_pos(N), orc.DummyNet(A), seed=5, game=3)
    for _ in range(30):
        t.tree_search(8)
    assert t.check() > 100
    t.close()


def test_long_game_and_failsafe_twin():  # test_mcts_player.jl:204-252
    endgame = orc.make_pos(N, board=load_board(TT_FTW, N), n=ENV.max_game_length - 2, komi=2.5,
                           recent=[(BLACK, orc.rc(1, 2, N)), (WHITE, orc.rc(1, 9, N))], to_play=BLACK)
    t = Twin(endgame, orc.DummyNet(A))
    for _ in range(10):
        t.tree_search(8)
    t.check()
    assert t.sim.Q_(0, t.sim.game(0).root) > 0
    t.close()
    probs = np.ones(A) * 0.001
    probs[-1] = 1
    start = orc.make_pos(N)
    passed = orc.OPos()
    L.or_pass_move(C.byref(start), C.byref(passed))
    t = Twin(passed, orc.DummyNet(A, fake_priors=probs))
    t.tree_search(1)
    t.tree_search(8)
    t.check()
    t.close()


def test_only_check_game_end_once_twin():  # test_mcts_player.jl:254-283
    pos = orc.make_pos(N)
    for a in (orc.rc(4, 4, N), orc.rc(4, 5, N), orc.rc(5, 4, N), P):
        _, pos = orc.play(pos, a)
    t = Twin(pos, orc.DummyNet(A))
    for _ in range(15):
        t.tree_search(8)
    root = t.sim.game(0).root
    assert t.sim.row(0, root, 0)[P] == 1
    t.tree_search(8)
    assert t.sim.row(0, root, 0)[P] == 1
    t.check()
    t.close()


def test_play_pick_and_reroot_twin():
    """a short game driven move by move: pick_move (soft-pick early, argmax late), play_move!
    (pi / Q recording, subtree reuse), noise injection"""
    t = Twin(orc.make_pos(N), orc.DummyNet(A), seed=21, game=7)
    rng = np.random.RandomState(0)
    for move in range(12):
        for _ in range(6):
            t.tree_search(8)
        t.check()
        d = orc.ODraw(21, 7, L.or_node_pos(t.oroot).contents.n, 0)
        L.or_inject_noise(C.byref(ENV), t.oroot, C.byref(d))
        t.sim.op(hs.TOP_NOISE, node=t.sim.game(0).root)
        t.check()
        a = t.pick()
        assert t.play(a) == 1
        t.check()
        assert t.sim.op(hs.TOP_RESIGN)[1] == L.or_player_should_resign(t.op)
    # recorded pi / q agree exactly
    nm = L.or_player_num_moves(t.op)
    g = t.sim.game(0)
    assert g.move_count == nm == 12
    t.close()


def test_illegal_play_returns_false_twin():  # mcts_play.jl:39-46
    pos = orc.make_pos(N, board=load_board(ALMOST_DONE, N), to_play=BLACK)
    t = Twin(pos, orc.DummyNet(A))
    t.tree_search(8)
    assert t.play(1) == 0
    t.check()
    t.close()


def test_incorporate_into_done_node():  # test_mcts.jl:116-127
    sim = hs.Sim(board_size=N, games=1, num_readouts=8, max_nodes_per_game=64)
    root = sim.tree_init(0, np.zeros(P, np.int8))
    probs = np.full(A, 0.02, np.float32)
    st, leaf = sim.op(hs.TOP_SELECT, node=root)
    assert leaf == root
    assert sim.op(hs.TOP_INCORPORATE, node=root, up_to=root, probs=probs, value=0.0)[0] == 0
    st, p1 = sim.op(hs.TOP_ADD_CHILD, node=root, a=P)
    assert sim.op(hs.TOP_INCORPORATE, node=p1, up_to=root, probs=probs, value=0.0)[0] == 0
    st, p2 = sim.op(hs.TOP_ADD_CHILD, node=p1, a=P)
    assert sim.op(hs.TOP_INCORPORATE, node=p2, up_to=root, probs=probs, value=0.0)[0] == 2  # AGZ_ASSERT_DONE_NODE
    st, leaf = sim.op(hs.TOP_SELECT, node=p2)
    assert leaf == p2
    # add_child idempotency (test_mcts.jl:129-144)
    st, c1 = sim.op(hs.TOP_ADD_CHILD, node=root, a=16)
    st, c2 = sim.op(hs.TOP_ADD_CHILD, node=root, a=16)
    assert c1 == c2 and sim.meta(0, c1).parent == root and sim.meta(0, c1).fmove == 16
    sim.close()


def test_never_select_illegal_moves():  # test_mcts.jl:146-167
    pos = orc.make_pos(N, board=load_board(ALMOST_DONE, N), n=75, komi=0.5,
                       recent=[(BLACK, orc.rc(1, 2, N)), (WHITE, orc.rc(1, 9, N)), (BLACK, orc.rc(2, 1, N))],
                       to_play=WHITE)
    sim = hs.Sim(board_size=N, games=1, num_readouts=8, max_nodes_per_game=256)
    root = sim.tree_init(0, pos.board_np(), n=75, to_play=WHITE, komi=0.5, last_move=orc.rc(2, 1, N))
    probs = np.full(A, 0.02, np.float32)
    probs[1] = 0.99
    assert sim.op(hs.TOP_INCORPORATE, node=root, up_to=root, probs=probs, value=0.0)[0] == 0
    sim.L.hs_node_set_N(sim.h, 0, root, 10000.0)
    legal = sim.legal(0, root).astype(bool)
    sim.row(0, root, 0)[legal] = 10000
    st, leaf = sim.op(hs.TOP_SELECT, node=root)
    assert sim.meta(0, leaf).fmove != 1
    for i in range(10):
        sim.op(hs.TOP_NOISE, node=root)
        st, leaf = sim.op(hs.TOP_SELECT, node=root)
        assert sim.meta(0, leaf).fmove != 1
    sim.close()


def test_pool_exhaustion_is_flagged():
    t = hs.Sim(board_size=N, games=1, num_readouts=8, max_nodes_per_game=6)
    root = t.tree_init(0, np.zeros(P, np.int8))
    probs = np.ones(A, np.float32) / A
    t.op(hs.TOP_INCORPORATE, node=root, up_to=root, probs=probs, value=0.0)
    for a in range(5):
        st, c = t.op(hs.TOP_ADD_CHILD, node=root, a=a)
        assert st == 0
    st, c = t.op(hs.TOP_ADD_CHILD, node=root, a=7)
    assert st == 8   # AGZ_POOL_EXHAUSTED
    assert t.counters()["pool_exhausted"] >= 1
    t.close()
