"""ctypes binding of tests/hostsim/libhostsim.so -- the host wave simulator over the engine's
search templates (TEST INFRASTRUCTURE; see tests/hostsim/hostsim.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CT = ["steps", "positions", "started", "finished", "evals", "dup", "terminal", "rootvisits",
      "pool_exhausted", "resigned", "claimed", "recorded",
      "t_free", "t_pick", "t_child", "t_reroot", "t_noise", "t_move_select", "n_move", "t_select", "n_select", "t_move_max",
      "t_create", "n_create", "pool_short", "peak_nodes"]          # enum Counter of agz_state.h, in order


class AgzConfig(C.Structure):
    _fields_ = [
        ("board_size", C.c_int32), ("tower_height", C.c_int32), ("games", C.c_int32),
        ("num_readouts", C.c_int32), ("parallel_readouts", C.c_int32), ("two_player_mode", C.c_int32),
        ("komi", C.c_float), ("reserved0", C.c_float),
        ("c_puct", C.c_double), ("dirichlet_noise_weight", C.c_double), ("resign_threshold", C.c_double),
        ("resign_disable_fraction", C.c_double),
        ("seed", C.c_uint64), ("game_id_base", C.c_uint64), ("game_id_stride", C.c_uint64),
        ("max_nodes_per_game", C.c_int32), ("device", C.c_int32), ("external_network", C.c_int32),
        ("pool_policy", C.c_int32), ("record_capacity_games", C.c_int32), ("arena_mode", C.c_int32),
    ]


def default_config(**kw):
    c = AgzConfig()
    c.board_size = 19
    c.tower_height = 19
    c.games = 1
    c.num_readouts = 800
    c.parallel_readouts = 8
    c.two_player_mode = 0
    c.komi = 7.5
    c.c_puct = 0.96
    c.dirichlet_noise_weight = 0.25
    c.resign_threshold = -0.9
    c.resign_disable_fraction = 0.05
    c.seed = 0
    c.game_id_base = 0
    c.game_id_stride = 1
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class PositionInfo(C.Structure):
    _fields_ = [("n", C.c_int32), ("to_play", C.c_int32), ("ko", C.c_int32), ("caps_black", C.c_int32),
                ("caps_white", C.c_int32), ("last_move", C.c_int32), ("prev_move", C.c_int32),
                ("history_len", C.c_int32), ("komi", C.c_float)]


class GameHeader(C.Structure):
    _fields_ = [("game_id", C.c_uint64), ("num_moves", C.c_int32), ("result", C.c_int32),
                ("was_resign", C.c_int32), ("resign_disabled", C.c_int32), ("final_score", C.c_float),
                ("short_searches", C.c_int32)]


class NodeMeta(C.Structure):
    _fields_ = [("parent", C.c_int32), ("n", C.c_int32), ("ko", C.c_int32), ("caps_b", C.c_int32),
                ("caps_w", C.c_int32), ("fmove", C.c_int16), ("last_move", C.c_int16), ("losses", C.c_int16),
                ("to_play", C.c_int8), ("flags", C.c_uint8), ("pad", C.c_int32)]


class GameState(C.Structure):
    _fields_ = [("game_id", C.c_uint64), ("resign_threshold", C.c_double), ("rootN", C.c_float),
                ("rootW", C.c_float), ("target", C.c_float), ("komi", C.c_float), ("root", C.c_int32),
                ("phase", C.c_int32), ("sel", C.c_int32), ("move_count", C.c_int32), ("nqs", C.c_int32),
                ("hist_len", C.c_int32), ("free_top", C.c_int32), ("nleaves", C.c_int32),
                ("leaf_base", C.c_int32), ("resign_disabled", C.c_int32), ("err", C.c_int32),
                ("result", C.c_int32), ("was_resign", C.c_int32), ("nodes_used", C.c_int32),
                ("short_first", C.c_int32), ("arena_k", C.c_int32), ("garbage", C.c_int32), ("npend", C.c_int32),
                ("short_searches", C.c_int32), ("stalled", C.c_int32)]


class TreeArgs(C.Structure):
    _fields_ = [("op", C.c_int32), ("g", C.c_int32), ("node", C.c_int32), ("a", C.c_int32),
                ("up_to", C.c_int32), ("par", C.c_int32), ("value", C.c_float), ("info", PositionInfo),
                ("probs", C.POINTER(C.c_float)), ("board", C.POINTER(C.c_int8)),
                ("history", C.POINTER(C.c_int8)), ("iout", C.POINTER(C.c_int32)),
                ("dout", C.POINTER(C.c_double))]


(TOP_INIT, TOP_SELECT, TOP_ADD_CHILD, TOP_VLOSS_ADD, TOP_VLOSS_REVERT, TOP_INCORPORATE, TOP_NOISE,
 TOP_SEARCH_SELECT, TOP_SEARCH_POST, TOP_PICK, TOP_PLAY, TOP_RESIGN, TOP_SCORES, TOP_PENDING) = range(14)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    d = os.path.join(HERE, "hostsim")
    if os.path.exists("/root/reference") or not os.path.exists(os.path.join(d, "libhostsim.so")):
        subprocess.run(["make", "-s", "-C", d], check=True)
    L = C.CDLL(os.path.join(d, "libhostsim.so"))
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    P = C.POINTER
    sig = {
        "hs_create": (vp, [P(AgzConfig)]), "hs_destroy": (None, [vp]), "hs_dims": (None, [vp, P(C.c_int32)]),
        "hs_start": (None, [vp, C.c_int64]), "hs_pre": (i, [vp]), "hs_leaf_features": (None, [vp, P(f)]),
        "hs_post": (None, [vp, P(f), P(f)]), "hs_counters": (None, [vp, P(C.c_ulonglong)]),
        "hs_arena_counts": (None, [vp, P(C.c_int32)]),
        "hs_live_games": (i, [vp]), "hs_records_count": (C.c_long, [vp]),
        "hs_record_header": (None, [vp, C.c_long, P(GameHeader)]),
        "hs_record_game": (None, [vp, C.c_long, P(C.c_int16), P(f), P(f)]),
        "hs_go_play": (None, [vp, P(C.c_int8), P(C.c_int8), P(C.c_int32), P(C.c_int32), i, P(C.c_int8),
                              P(C.c_int32), P(C.c_int32), P(C.c_int32)]),
        "hs_go_legal": (None, [vp, P(C.c_int8), P(C.c_int8), P(C.c_int32), i, P(C.c_int8)]),
        "hs_go_score": (None, [vp, P(C.c_int8), P(f), i, P(f)]),
        "hs_tree_op": (i, [vp, P(TreeArgs), P(C.c_int32)]),
        "hs_set_batch_outputs": (None, [vp, P(f), P(f), i]),
        "hs_tree_leaf_features": (None, [vp, i, P(f)]),
        "hs_game_state": (None, [vp, i, P(GameState)]), "hs_game_set": (None, [vp, i, i, C.c_double]),
        "hs_node_meta": (None, [vp, i, i, P(NodeMeta)]), "hs_node_set_n": (None, [vp, i, i, i]),
        "hs_node_N": (f, [vp, i, i]), "hs_node_W": (f, [vp, i, i]), "hs_node_set_N": (None, [vp, i, i, f]),
        "hs_node_row": (P(f), [vp, i, i, i]), "hs_node_children": (P(C.c_int32), [vp, i, i]),
        "hs_node_board": (P(C.c_int8), [vp, i, i]), "hs_node_legal": (None, [vp, i, i, P(C.c_int8)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def p8(a):
    return a.ctypes.data_as(C.POINTER(C.c_int8))


def p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Sim:
    """A hostsim engine plus the helpers the parity tests need."""

    def __init__(self, **cfg):
        self.L = lib()
        self.cfg = default_config(**cfg)
        self.h = self.L.hs_create(C.byref(self.cfg))
        d = (C.c_int32 * 10)()
        self.L.hs_dims(self.h, d)
        (self.N, self.P, self.A, self.AP, self.cap, self.games, self.par, self.mgl, self.tau, self.maxd) = list(d)

    def close(self):
        if self.h:
            self.L.hs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- batched self-play
    def start(self, total_games):
        self.L.hs_start(self.h, total_games)

    def step(self, net, white_net=None):
        """one self-play step; net(feats [B,17*P] float32) -> (pi [B,A], v [B]).  Arena mode:
        `net` answers the Black players' leaves, `white_net` the White players'."""
        B = self.L.hs_pre(self.h)
        if B == 0:
            if white_net is not None:      # a step may consist of terminal leaves only: moves happen in post
                z = np.zeros((1, self.A), np.float32)
                self.L.hs_post(self.h, pf(z), pf(np.zeros(1, np.float32)))
            return 0
        feats = np.zeros((B, 17 * self.P), np.float32)
        self.L.hs_leaf_features(self.h, pf(feats))
        if white_net is not None:
            cnt = (C.c_int32 * 2)()
            self.L.hs_arena_counts(self.h, cnt)
            pi, v = np.zeros((B, self.A), np.float32), np.zeros(B, np.float32)
            if cnt[0]:
                pi[:cnt[0]], v[:cnt[0]] = net(feats[:cnt[0]])
            if cnt[1]:
                pi[cnt[0]:], v[cnt[0]:] = white_net(feats[cnt[0]:])
        else:
            pi, v = net(feats)
        pi = np.ascontiguousarray(pi, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        self.L.hs_post(self.h, pf(pi), pf(v))
        return B

    def counters(self):
        out = (C.c_ulonglong * 64)()            # enum Counter of agz_state.h is longer than the names kept here
        self.L.hs_counters(self.h, out)
        return dict(zip(CT, list(out)))

    def records(self):
        out = []
        for k in range(self.L.hs_records_count(self.h)):
            hd = GameHeader()
            self.L.hs_record_header(self.h, k, C.byref(hd))
            nm = hd.num_moves
            moves = np.zeros(max(nm, 1), np.int16)
            pis = np.zeros((max(nm, 1), self.A), np.float32)
            qs = np.zeros(max(nm, 1), np.float32)
            self.L.hs_record_game(self.h, k, moves.ctypes.data_as(C.POINTER(C.c_int16)), pf(pis), pf(qs))
            out.append(dict(game_id=hd.game_id, num_moves=nm, result=hd.result, was_resign=hd.was_resign,
                            resign_disabled=hd.resign_disabled, final_score=hd.final_score,
                            short_searches=hd.short_searches, moves=moves[:nm].copy(), pis=pis[:nm].copy(), qs=qs[:nm].copy()))
        return sorted(out, key=lambda r: r["game_id"])

    # ---- Go rules
    def go_play(self, boards, to_play, ko, moves):
        B = len(moves)
        boards = np.ascontiguousarray(boards, np.int8)
        to_play = np.ascontiguousarray(to_play, np.int8)
        ko = np.ascontiguousarray(ko, np.int32)
        moves = np.ascontiguousarray(moves, np.int32)
        bo = np.zeros_like(boards)
        ko_o = np.zeros(B, np.int32)
        nc = np.zeros(B, np.int32)
        st = np.zeros(B, np.int32)
        self.L.hs_go_play(self.h, p8(boards), p8(to_play), p32(ko), p32(moves), B, p8(bo), p32(ko_o), p32(nc), p32(st))
        return bo, ko_o, nc, st

    def go_legal(self, boards, to_play, ko):
        B = len(to_play)
        boards = np.ascontiguousarray(boards, np.int8)
        out = np.zeros((B, self.A), np.int8)
        self.L.hs_go_legal(self.h, p8(boards), p8(np.ascontiguousarray(to_play, np.int8)),
                           p32(np.ascontiguousarray(ko, np.int32)), B, p8(out))
        return out

    def go_score(self, boards, komi):
        B = len(komi)
        out = np.zeros(B, np.float32)
        self.L.hs_go_score(self.h, p8(np.ascontiguousarray(boards, np.int8)),
                           pf(np.ascontiguousarray(komi, np.float32)), B, pf(out))
        return out

    # ---- single-tree ops
    def op(self, op, g=0, node=0, a=0, up_to=-1, par=0, value=0.0, probs=None, board=None, info=None,
           history=None, dout=None):
        T = TreeArgs()
        T.op, T.g, T.node, T.a, T.up_to, T.par, T.value = op, g, node, a, up_to, par, value
        keep = []
        if probs is not None:
            pr = np.ascontiguousarray(probs, np.float32)
            keep.append(pr)
            T.probs = pf(pr)
        if board is not None:
            bd = np.ascontiguousarray(board, np.int8)
            keep.append(bd)
            T.board = p8(bd)
        if history is not None:
            hh = np.ascontiguousarray(history, np.int8)
            keep.append(hh)
            T.history = p8(hh)
        if info is not None:
            T.info = info
        if dout is not None:
            T.dout = dout.ctypes.data_as(C.POINTER(C.c_double))
        r0 = C.c_int32()
        st = self.L.hs_tree_op(self.h, C.byref(T), C.byref(r0))
        return st, r0.value

    def tree_init(self, g, board, n=0, to_play=1, ko=-1, caps=(0, 0), last_move=-1, komi=7.5, history=None):
        info = PositionInfo()
        info.n, info.to_play, info.ko = n, to_play, ko
        info.caps_black, info.caps_white = caps
        info.last_move, info.prev_move = last_move, -1
        info.history_len = 0 if history is None else len(history)
        info.komi = komi
        st, root = self.op(TOP_INIT, g=g, board=board, info=info, history=history)
        assert st == 0
        return root

    def game(self, g):
        s = GameState()
        self.L.hs_game_state(self.h, g, C.byref(s))
        return s

    def meta(self, g, node):
        m = NodeMeta()
        self.L.hs_node_meta(self.h, g, node, C.byref(m))
        return m

    def row(self, g, node, field):
        return np.ctypeslib.as_array(self.L.hs_node_row(self.h, g, node, field), shape=(self.AP,))[: self.A]

    def children(self, g, node):
        return np.ctypeslib.as_array(self.L.hs_node_children(self.h, g, node), shape=(self.AP,))[: self.A]

    def board(self, g, node):
        return np.ctypeslib.as_array(self.L.hs_node_board(self.h, g, node), shape=(self.P,)).copy()

    def legal(self, g, node):
        out = np.zeros(self.A, np.int8)
        self.L.hs_node_legal(self.h, g, node, p8(out))
        return out

    def N_(self, g, node):
        return self.L.hs_node_N(self.h, g, node)

    def W_(self, g, node):
        return self.L.hs_node_W(self.h, g, node)

    def Q_(self, g, node):
        return np.float32(self.W_(g, node)) / (np.float32(1) + np.float32(self.N_(g, node)))
