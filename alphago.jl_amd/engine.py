"""Engine: a numpy-friendly handle on one libagz engine (one MI355X, one HIP stream).

Thin by design: every method is one C-ABI call of include/agz.h plus array marshalling.  All
indices are 0-based (point p = row + N*col, action N*N = pass)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import AgzError, IllegalMove


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _i8(a):
    return np.ascontiguousarray(a, dtype=np.int8)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Engine:
    def __init__(self, **cfg):
        self.L = _lib.load()
        self.cfg = _lib.default_config(**cfg)
        h = C.c_void_p()
        st = self.L.agz_engine_create(C.byref(self.cfg), C.byref(h))
        if st != _lib.OK:
            raise AgzError(st, self.L.agz_last_error(None).decode())
        self.h = h
        self.N = self.cfg.board_size
        self.P = self.N * self.N
        self.A = self.P + 1
        self.max_game_length = (self.P * 7) // 5
        self.tower_height = self.cfg.tower_height

    def close(self):
        if getattr(self, "h", None):
            self.L.agz_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, st, allow=()):
        if st != _lib.OK and st not in allow:
            msg = self.L.agz_last_error(self.h).decode()
            raise (IllegalMove if st == _lib.ILLEGAL_MOVE else AgzError)(st, msg)
        return st

    def sync(self):
        self._ck(self.L.agz_engine_sync(self.h))

    # ------------------------------------------------------------ network
    def set_weights(self, layer, kind, data):
        d = _f32(np.asarray(data).reshape(-1, order="A") if isinstance(data, np.ndarray) else data)
        self._ck(self.L.agz_net_set_weights(self.h, layer, kind, _p(d, C.c_float), d.size))

    def get_weights(self, layer, kind):
        n = self.param_count(layer, kind)
        out = np.zeros(n, np.float32)
        self._ck(self.L.agz_net_get_weights(self.h, layer, kind, _p(out, C.c_float), n))
        return out

    def layers(self):
        """every (layer, kind) pair of this network, in a stable order"""
        t = self.tower_height
        out = [(l, k) for l in list(range(0, 1 + 2 * t)) + [_lib.L_VALUE_CONV, _lib.L_POLICY_CONV] for k in range(7)]
        out += [(l, k) for l in (_lib.L_VALUE_FC1, _lib.L_VALUE_FC2, _lib.L_POLICY_FC) for k in (0, 1)]
        return out

    def copy_weights_to(self, other):
        for l, k in self.layers():
            other.set_weights(l, k, self.get_weights(l, k))

    def param_count(self, layer, kind):
        return self.L.agz_net_param_count(self.h, layer, kind)

    def init_synthetic(self, seed=0):
        self._ck(self.L.agz_net_init_synthetic(self.h, seed))

    def forward(self, boards, deltas, ndeltas, to_play):
        """positions SoA -> (pi [B, A], v [B])"""
        boards, deltas, ndeltas, to_play = _i8(boards), _i8(deltas), _i32(ndeltas), _i8(to_play)
        B = len(to_play)
        pi = np.zeros((B, self.A), np.float32)
        v = np.zeros(B, np.float32)
        self._ck(self.L.agz_net_forward(self.h, _p(boards, C.c_int8), _p(deltas, C.c_int8), _p(ndeltas, C.c_int32),
                                        _p(to_play, C.c_int8), B, _p(pi, C.c_float), _p(v, C.c_float)))
        return pi, v

    def forward_features(self, feats):
        """feats [B, 17*P] (N x N x 17 per position, column-major) -> (pi [B, A], v [B])"""
        feats = _f32(feats)
        B = feats.shape[0]
        pi = np.zeros((B, self.A), np.float32)
        v = np.zeros(B, np.float32)
        self._ck(self.L.agz_net_forward_features(self.h, _p(feats, C.c_float), B, _p(pi, C.c_float), _p(v, C.c_float)))
        return pi, v

    def features(self, boards, deltas, ndeltas, to_play):
        boards, deltas, ndeltas, to_play = _i8(boards), _i8(deltas), _i32(ndeltas), _i8(to_play)
        B = len(to_play)
        out = np.zeros((B, 17 * self.P), np.float32)
        self._ck(self.L.agz_features(self.h, _p(boards, C.c_int8), _p(deltas, C.c_int8), _p(ndeltas, C.c_int32),
                                     _p(to_play, C.c_int8), B, _p(out, C.c_float)))
        return out

    def time_forward(self, B, iters):
        ms = C.c_float()
        self._ck(self.L.agz_net_time_forward(self.h, B, iters, C.byref(ms)))
        return ms.value

    def time_conv(self, B, iters):
        ms = C.c_float()
        self._ck(self.L.agz_net_time_conv(self.h, B, iters, C.byref(ms)))
        return ms.value

    def set_winograd(self, on=True):
        self._ck(self.L.agz_net_set_winograd(self.h, 1 if on else 0))

    def set_precision(self, precision="f32"):
        """tower arithmetic: "f32" (default, exact) or "f16" (fp16 operands, f32 accumulate)"""
        code = {"f32": 0, "f16": 1, 0: 0, 1: 1}[precision]
        self._ck(self.L.agz_net_set_precision(self.h, code))

    def profile_conv(self, on=True):
        self._ck(self.L.agz_profile_conv_enable(self.h, 1 if on else 0))

    def profile_conv_read(self):
        """(total ms, total algorithmic flop, launches) of the timed tower-conv launches"""
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        self._ck(self.L.agz_profile_conv_read(self.h, C.byref(ms), C.byref(fl), C.byref(n)))
        return ms.value, fl.value, n.value

    # ------------------------------------------------------------ Go rules
    def go_play(self, boards, to_play, ko, moves):
        boards, to_play, ko, moves = _i8(boards), _i8(to_play), _i32(ko), _i32(moves)
        B = len(moves)
        bo = np.zeros_like(boards)
        ko_o, nc, st = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        self._ck(self.L.agz_go_play(self.h, _p(boards, C.c_int8), _p(to_play, C.c_int8), _p(ko, C.c_int32),
                                    _p(moves, C.c_int32), B, _p(bo, C.c_int8), _p(ko_o, C.c_int32),
                                    _p(nc, C.c_int32), _p(st, C.c_int32)))
        return bo, ko_o, nc, st

    def go_legal(self, boards, to_play, ko):
        boards, to_play, ko = _i8(boards), _i8(to_play), _i32(ko)
        B = len(to_play)
        out = np.zeros((B, self.A), np.int8)
        self._ck(self.L.agz_go_legal(self.h, _p(boards, C.c_int8), _p(to_play, C.c_int8), _p(ko, C.c_int32), B,
                                     _p(out, C.c_int8)))
        return out

    def go_score(self, boards, komi):
        boards, komi = _i8(boards), _f32(komi)
        B = len(komi)
        out = np.zeros(B, np.float32)
        self._ck(self.L.agz_go_score(self.h, _p(boards, C.c_int8), _p(komi, C.c_float), B, _p(out, C.c_float)))
        return out

    # ------------------------------------------------------------ batched self-play
    def start(self, total_games=0):
        self._ck(self.L.agz_selfplay_start(self.h, total_games))

    def step(self, nsteps=1):
        self._ck(self.L.agz_selfplay_step(self.h, nsteps))

    def stats(self):
        s = _lib.Stats()
        self._ck(self.L.agz_engine_stats(self.h, C.byref(s)))
        return s.as_dict()

    def select(self):
        n = C.c_int32()
        self._ck(self.L.agz_selfplay_select(self.h, C.byref(n)))
        return n.value

    def leaf_features(self, nleaves):
        out = np.zeros((max(nleaves, 1), 17 * self.P), np.float32)
        self._ck(self.L.agz_selfplay_leaf_features(self.h, _p(out, C.c_float)))
        return out[:nleaves]

    def incorporate(self, pi, v):
        pi, v = _f32(pi), _f32(v)
        self._ck(self.L.agz_selfplay_incorporate(self.h, _p(pi, C.c_float), _p(v, C.c_float)))

    def net_select(self, which):
        """arena_mode: address network 0 (Black's) or 1 (White's) with the weight / forward calls"""
        self._ck(self.L.agz_net_select(self.h, which))

    def arena_counts(self):
        out = (C.c_int32 * 2)()
        self._ck(self.L.agz_arena_counts(self.h, out))
        return int(out[0]), int(out[1])

    def step_external(self, network, white_network=None):
        """one step with a caller-supplied network(feats [B,17P]) -> (pi [B,A], v [B]); in arena
        mode `network` answers the Black players' leaves and `white_network` the White players'"""
        n = self.select()
        if n > 0 and white_network is not None:
            n0, n1 = self.arena_counts()
            feats = self.leaf_features(n)
            pi, v = np.zeros((n, self.A), np.float32), np.zeros(n, np.float32)
            if n0:
                pi[:n0], v[:n0] = network(feats[:n0])
            if n1:
                pi[n0:], v[n0:] = white_network(feats[n0:])
            self.incorporate(pi, v)
        elif n > 0:
            pi, v = network(self.leaf_features(n))
            self.incorporate(pi, v)
        else:
            self.incorporate(np.zeros((1, self.A), np.float32), np.zeros(1, np.float32))
        return n

    def records_count(self):
        return self.L.agz_records_count(self.h)

    def records(self):
        """finished games, sorted by game id: dicts with moves [n], pis [n, A], qs [n], result..."""
        out = []
        for k in range(self.records_count()):
            hd = _lib.GameHeader()
            self._ck(self.L.agz_records_header(self.h, k, C.byref(hd)))
            nm = hd.num_moves
            moves = np.zeros(max(nm, 1), np.int16)
            pis = np.zeros((max(nm, 1), self.A), np.float32)
            qs = np.zeros(max(nm, 1), np.float32)
            self._ck(self.L.agz_records_game(self.h, k, _p(moves, C.c_int16), _p(pis, C.c_float), _p(qs, C.c_float)))
            out.append(dict(index=k, game_id=hd.game_id, num_moves=nm, result=hd.result, was_resign=hd.was_resign,
                            resign_disabled=hd.resign_disabled, final_score=hd.final_score,
                            moves=moves[:nm].copy(), pis=pis[:nm].copy(), qs=qs[:nm].copy()))
        return sorted(out, key=lambda r: r["game_id"])

    def record_features(self, k, num_moves):
        out = np.zeros((max(num_moves, 1), 17 * self.P), np.float32)
        self._ck(self.L.agz_records_features(self.h, k, _p(out, C.c_float)))
        return out[:num_moves]

    def replay_features(self, moves, game_offset, ply, out=None):
        """features of sampled (game, ply) pairs by device replay (agz_replay_features).  `out` may be
        a CUDA float32 torch tensor [B, 17*P] (filled in place, no host copy) or None -> numpy."""
        moves = np.ascontiguousarray(moves, np.int16)
        off = np.ascontiguousarray(game_offset, np.int32)
        ply = np.ascontiguousarray(ply, np.int32)
        B = len(off)
        assert len(ply) == B
        if out is None:
            res = np.zeros((B, 17 * self.P), np.float32)
            ptr, dev = res.ctypes.data_as(C.c_void_p), 0
        else:
            assert out.is_cuda and out.is_contiguous() and out.numel() == B * 17 * self.P and out.element_size() == 4
            res, ptr, dev = out, C.c_void_p(out.data_ptr()), 1
        self._ck(self.L.agz_replay_features(self.h, _p(moves, C.c_int16), moves.size, _p(off, C.c_int32),
                                            _p(ply, C.c_int32), B, ptr, dev))
        return res

    def records_packed(self):
        n = C.c_int64()
        self._ck(self.L.agz_records_packed_size(self.h, C.byref(n)))
        buf = np.zeros(max(n.value, 1), np.uint8)
        self._ck(self.L.agz_records_export_packed(self.h, buf.ctypes.data_as(C.c_void_p), n.value, 0))
        return buf[: n.value]

    def records_packed_device(self):
        """the same packed records written straight into a CUDA uint8 tensor (no host staging):
        the send buffer of the RCCL replay all-gather"""
        import torch
        n = C.c_int64()
        self._ck(self.L.agz_records_packed_size(self.h, C.byref(n)))
        buf = torch.empty(max(n.value, 1), dtype=torch.uint8, device=torch.device("cuda", self.cfg.device))
        self._ck(self.L.agz_records_export_packed(self.h, C.c_void_p(buf.data_ptr()), n.value, 1))
        return buf[: n.value]

    def records_clear(self):
        self._ck(self.L.agz_records_clear(self.h))

    # ------------------------------------------------------------ single-tree compat (slot g)
    def tree_init(self, g, board, n=0, to_play=1, ko=-1, caps=(0, 0), last_move=-1, komi=7.5, history=None):
        info = _lib.PositionInfo()
        info.n, info.to_play, info.ko = n, to_play, ko
        info.caps_black, info.caps_white = caps
        info.last_move, info.prev_move = last_move, -1
        info.history_len = 0 if history is None else len(history)
        info.komi = komi
        b = _i8(board).reshape(-1)
        hp = None
        if history is not None and len(history):
            hh = _i8(history).reshape(len(history), -1)
            hp = _p(hh, C.c_int8)
        self._ck(self.L.agz_tree_init(self.h, g, _p(b, C.c_int8), C.byref(info), hp))
        return self.tree_root(g)

    def tree_root(self, g):
        r = C.c_int32()
        self._ck(self.L.agz_tree_root(self.h, g, C.byref(r)))
        return r.value

    def select_leaf(self, g, node):
        r = C.c_int32()
        self._ck(self.L.agz_tree_select_leaf(self.h, g, node, C.byref(r)))
        return r.value

    def maybe_add_child(self, g, node, a):
        r = C.c_int32()
        self._ck(self.L.agz_tree_maybe_add_child(self.h, g, node, a, C.byref(r)))
        return r.value

    def add_virtual_loss(self, g, node, up_to):
        self._ck(self.L.agz_tree_add_virtual_loss(self.h, g, node, up_to))

    def revert_virtual_loss(self, g, node, up_to):
        self._ck(self.L.agz_tree_revert_virtual_loss(self.h, g, node, up_to))

    def incorporate_results(self, g, node, probs, value, up_to):
        pr = _f32(probs)
        return self.L.agz_tree_incorporate(self.h, g, node, _p(pr, C.c_float), pr.size, float(value), up_to)

    def inject_noise(self, g, node):
        self._ck(self.L.agz_tree_inject_noise(self.h, g, node))

    def tree_search(self, g, par=8, network=None):
        """tree_search!(player, parallel_readouts); network(feats)->(pi, v) or None for the engine's own"""
        n = C.c_int32()
        self._ck(self.L.agz_tree_search_select(self.h, g, par, C.byref(n)))
        if network is None:
            self._ck(self.L.agz_tree_search_incorporate(self.h, g, None, None))
        else:
            if n.value > 0:
                feats = np.zeros((n.value, 17 * self.P), np.float32)
                self._ck(self.L.agz_tree_leaf_features(self.h, g, _p(feats, C.c_float)))
                pi, v = network(feats)
                pi, v = _f32(pi), _f32(v)
                self._ck(self.L.agz_tree_search_incorporate(self.h, g, _p(pi, C.c_float), _p(v, C.c_float)))
            else:
                z = np.zeros(self.A, np.float32)
                self._ck(self.L.agz_tree_search_incorporate(self.h, g, _p(z, C.c_float), _p(z, C.c_float)))
        return n.value

    def pick_move(self, g):
        a = C.c_int32()
        st = self.L.agz_tree_pick_move(self.h, g, C.byref(a))
        return st, a.value

    def play_move(self, g, a):
        ok = C.c_int32()
        self._ck(self.L.agz_tree_play_move(self.h, g, a, C.byref(ok)))
        return ok.value

    def should_resign(self, g):
        r = C.c_int32()
        self._ck(self.L.agz_tree_should_resign(self.h, g, C.byref(r)))
        return r.value

    def is_done(self, g, node):
        r = C.c_int32()
        self._ck(self.L.agz_tree_is_done(self.h, g, node, C.byref(r)))
        return r.value

    def node_info(self, g, node):
        info = _lib.NodeInfo()
        self._ck(self.L.agz_tree_node_info(self.h, g, node, C.byref(info)))
        return info

    def node_floats(self, g, node, field):
        out = np.zeros(self.A, np.float32)
        self._ck(self.L.agz_tree_node_floats(self.h, g, node, field, _p(out, C.c_float)))
        return out

    def node_scores(self, g, node):
        out = np.zeros(self.A, np.float64)
        self._ck(self.L.agz_tree_node_scores(self.h, g, node, _p(out, C.c_double)))
        return out

    def node_set_floats(self, g, node, field, values):
        v = _f32(values)
        self._ck(self.L.agz_tree_node_set_floats(self.h, g, node, field, _p(v, C.c_float)))

    def node_set_N(self, g, node, value):
        self._ck(self.L.agz_tree_node_set_N(self.h, g, node, float(value)))

    def node_set_n(self, g, node, n):
        self._ck(self.L.agz_tree_node_set_n(self.h, g, node, n))

    def node_children(self, g, node):
        out = np.zeros(self.A, np.int32)
        self._ck(self.L.agz_tree_node_children(self.h, g, node, _p(out, C.c_int32)))
        return out

    def node_board(self, g, node):
        out = np.zeros(self.P, np.int8)
        self._ck(self.L.agz_tree_node_board(self.h, g, node, _p(out, C.c_int8)))
        return out

    def pending_vlosses(self, g):
        r = C.c_int32()
        self._ck(self.L.agz_tree_pending_vlosses(self.h, g, C.byref(r)))
        return r.value

    def set_draw(self, g, game_id, sel=0):
        self._ck(self.L.agz_tree_set_draw(self.h, g, game_id, sel))

    # ------------------------------------------------------------ diagnostics
    def debug_draws(self, seed, game, move, n, alpha):
        out = np.zeros(n, np.float64)
        self._ck(self.L.agz_debug_draws(self.h, seed, game, move, n, alpha, _p(out, C.c_double)))
        return out

    def debug_math(self, op, x, y=None):
        x = np.ascontiguousarray(x, np.float64)
        y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float64)
        out = np.zeros_like(x)
        self._ck(self.L.agz_debug_math(self.h, op, _p(x, C.c_double), _p(y, C.c_double), x.size, _p(out, C.c_double)))
        return out
