"""ctypes binding of the CPU oracle (oracle/liboracle.so) plus the small fixtures helpers the
reference's test-suite uses (test/test_utils.jl).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

MAXP = 361
MAXA = 362
MAXRECENT = 1024
BLACK, WHITE, EMPTY = 1, -1, 0

OK, ILLEGAL_MOVE, ASSERT_DONE_NODE, HISTORY_INCOMPLETE, BAD_SHAPE, ASSERT_SOFTPICK = range(6)

L_VALUE_CONV, L_POLICY_CONV, L_VALUE_FC1, L_VALUE_FC2, L_POLICY_FC = -1, -2, -3, -4, -5
K_WEIGHT, K_BIAS, K_BN_BETA, K_BN_GAMMA, K_BN_MEAN, K_BN_VAR, K_BN_EPS = range(7)


class OPos(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("A", C.c_int),
        ("board", C.c_int8 * MAXP),
        ("n", C.c_int), ("komi", C.c_float), ("caps", C.c_int * 2), ("ko", C.c_int),
        ("to_play", C.c_int), ("done", C.c_int), ("ndeltas", C.c_int),
        ("deltas", (C.c_int8 * MAXP) * 7),
        ("recent_len", C.c_int),
        ("recent_move", C.c_int16 * MAXRECENT),
        ("recent_color", C.c_int8 * MAXRECENT),
    ]

    def board_np(self):
        N = self.N
        return np.frombuffer(self.board, dtype=np.int8, count=N * N).copy()

    def copy(self):
        out = OPos()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(OPos))
        return out


class OEnv(C.Structure):
    _fields_ = [("N", C.c_int), ("A", C.c_int), ("max_game_length", C.c_int),
                ("dirichlet_alpha", C.c_float), ("c_puct", C.c_double), ("noise_weight", C.c_double)]


class OEvalGame(C.Structure):
    _fields_ = [("num_moves", C.c_int), ("result", C.c_int), ("was_resign", C.c_int), ("black_won", C.c_int),
                ("final_score", C.c_float), ("evals_black", C.c_uint64), ("evals_white", C.c_uint64)]


class ODraw(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("game", C.c_uint64), ("move", C.c_uint32), ("sel", C.c_uint32)]


NET_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.POINTER(OPos)), C.c_int,
                     C.POINTER(C.c_float), C.POINTER(C.c_float))

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("GOMP_SPINCOUNT", "0")
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path) or os.path.exists("/root/reference"):
        # in the build container always rebuild (cheap, make is incremental)
        try:
            build()
        except Exception:
            if not os.path.exists(path):
                raise
    L = C.CDLL(path)
    vp, i, f, d, u64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_uint64
    P = C.POINTER
    sig = {
        "or_env_init": (None, [P(OEnv), i]),
        "or_pos_init": (None, [P(OPos), i, f]),
        "or_pos_from_board": (None, [P(OPos), i, P(C.c_int8), i, f, i, i, i, i, i, P(C.c_int16), P(C.c_int8)]),
        "or_play_move": (i, [P(OPos), i, P(OPos)]),
        "or_play_move_color": (i, [P(OPos), i, i, P(OPos)]),
        "or_pass_move": (None, [P(OPos), P(OPos)]),
        "or_flip_playerturn": (None, [P(OPos), P(OPos)]),
        "or_is_koish": (i, [i, P(C.c_int8), i]),
        "or_is_eyeish": (i, [i, P(C.c_int8), i]),
        "or_is_move_suicidal": (i, [P(OPos), i]),
        "or_is_move_legal": (i, [P(OPos), i]),
        "or_all_legal_moves": (None, [P(OPos), P(C.c_int8)]),
        "or_score": (f, [P(OPos)]),
        "or_result": (i, [P(OPos)]),
        "or_result_string": (None, [P(OPos), C.c_char_p]),
        "or_group_info": (i, [i, P(C.c_int8), i, P(C.c_int8), P(C.c_int8)]),
        "or_count_groups": (i, [i, P(C.c_int8)]),
        "or_get_feats": (None, [P(OPos), P(f)]),
        "or_get_feats_f64": (None, [P(OPos), P(d)]),
        "or_node_new": (vp, [P(OEnv), P(OPos)]),
        "or_node_free_tree": (None, [vp]),
        "or_select_leaf": (vp, [P(OEnv), vp, P(ODraw)]),
        "or_maybe_add_child": (i, [P(OEnv), vp, i, P(vp)]),
        "or_add_virtual_loss": (None, [vp, vp]),
        "or_revert_virtual_loss": (None, [vp, vp]),
        "or_revert_visits": (None, [vp, vp]),
        "or_incorporate_results": (i, [P(OEnv), vp, P(f), i, f, vp]),
        "or_backup_value": (None, [vp, f, vp]),
        "or_node_is_done": (i, [P(OEnv), vp]),
        "or_inject_noise": (None, [P(OEnv), vp, P(ODraw)]),
        "or_children_as_pi": (None, [vp, i, P(f)]),
        "or_child_action_score": (None, [P(OEnv), vp, P(d)]),
        "or_node_N": (f, [vp]), "or_node_W": (f, [vp]), "or_node_Q": (f, [vp]),
        "or_node_set_N": (None, [vp, f]),
        "or_node_fmove": (i, [vp]), "or_node_is_expanded": (i, [vp]),
        "or_node_losses_applied": (i, [vp]),
        "or_node_child": (vp, [vp, i]), "or_node_parent": (vp, [vp]),
        "or_node_pos": (P(OPos), [vp]), "or_node_pos_mut": (P(OPos), [vp]),
        "or_node_child_N": (P(f), [vp]), "or_node_child_W": (P(f), [vp]),
        "or_node_child_prior": (P(f), [vp]), "or_node_original_prior": (P(f), [vp]),
        "or_tree_pending_vlosses": (i, [vp]), "or_tree_count_nodes": (i, [vp]),
        "or_player_new": (vp, [i, NET_FN, vp, i, i, d, u64, u64]),
        "or_player_free": (None, [vp]),
        "or_player_initialize_game": (None, [vp, P(OPos)]),
        "or_player_tree_search": (i, [vp, i]),
        "or_player_pick_move": (i, [vp, P(i)]),
        "or_player_play_move": (i, [vp, i]),
        "or_player_should_resign": (i, [vp]),
        "or_player_is_done": (i, [vp]),
        "or_player_set_result": (None, [vp, i, i]),
        "or_player_root": (vp, [vp]),
        "or_player_env": (P(OEnv), [vp]),
        "or_player_result": (i, [vp]),
        "or_player_result_string": (C.c_char_p, [vp]),
        "or_player_tau_threshold": (i, [vp]),
        "or_player_num_moves": (i, [vp]),
        "or_player_search_pi": (P(f), [vp, i]),
        "or_player_q": (f, [vp, i]),
        "or_player_nqs": (i, [vp]),
        "or_player_evals": (u64, [vp]),
        "or_player_extract_data": (i, [vp, P(OPos), P(f), P(i)]),
        "or_selfplay": (vp, [i, NET_FN, vp, i, u64, u64, i]),
        "or_selfplay_ex": (vp, [i, NET_FN, vp, i, u64, u64, i, d, d]),
        "or_evaluate_game": (None, [i, NET_FN, vp, NET_FN, vp, i, d, u64, u64, P(C.c_int16), P(f), P(OEvalGame)]),
        "or_net_new": (vp, [i, i]),
        "or_net_free": (None, [vp]),
        "or_net_set": (i, [vp, i, i, P(f), C.c_int64]),
        "or_net_get": (i, [vp, i, i, P(f), C.c_int64]),
        "or_net_param_count": (C.c_int64, [vp, i, i]),
        "or_net_init_synthetic": (None, [vp, u64]),
        "or_net_forward_feats": (None, [vp, P(f), i, P(f), P(f), i]),
        "or_quant_half": (C.c_float, [C.c_float]),
        "or_net_forward_feats_f64": (None, [vp, P(d), i, P(d), P(d)]),
        "or_net_callable": (None, [vp, P(P(OPos)), i, P(f), P(f)]),
        "or_set_num_threads": (None, [i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    # The CPU network is OpenMP code; the GPU box reports 256 hardware threads but the container may
    # own far fewer, and libgomp's spinning barriers then stall for minutes (seen: a 20 s suite taking
    # > 15 min).  Tests are small: cap the team and never spin.
    L.or_set_num_threads(default_threads())
    _lib = L
    return L


def default_threads(cap=8):
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = min(avail, int(float(q) / float(per) + 0.5))
    except Exception:
        pass
    return max(1, min(cap, avail))


# ---------------------------------------------------------------- fixtures helpers

_KGS = "ABCDEFGHJKLMNOPQRST"


def load_board(text, N):
    """test/test_utils.jl:1-18 -- 'X' black, 'O' white, '.' empty; text is row-major; the
    result is the flat board with p = row + N*col."""
    chars = [ch for ch in text if ch in "XO.#"]
    assert len(chars) == N * N, (len(chars), N)
    m = {"X": 1, "O": -1, ".": 0, "#": 2}
    b = np.zeros(N * N, dtype=np.int8)
    for k, ch in enumerate(chars):
        r, c = divmod(k, N)
        b[r + N * c] = m[ch]
    return b


def from_kgs(s, N):
    """coords.jl:26-34; returns the flat 0-based action (N*N for 'pass')."""
    if s == "pass":
        return N * N
    col = _KGS.index(s[0].upper())
    row = N - int(s[1:])
    return row + N * col


def to_kgs(a, N):
    if a == N * N:
        return "pass"
    row, col = a % N, a // N
    return f"{_KGS[col]}{N - row}"


def from_sgf(s, N):
    sg = "abcdefghijklmnopqrstuvwxyz"
    if not s:
        return N * N
    return sg.index(s[1]) + N * sg.index(s[0])


def rc(row1, col1, N):
    """1-based (row, col) of the reference -> flat 0-based point."""
    return (row1 - 1) + N * (col1 - 1)


def pc_set(s, N):
    return {from_kgs(t, N) for t in s.split()}


def make_pos(N, board=None, n=0, komi=7.5, caps=(0, 0), ko=-1, recent=(), to_play=BLACK):
    """Position(env; board, n, komi, caps, ko, recent, to_play); recent = [(color, action)]"""
    L = lib()
    pos = OPos()
    nb = None
    if board is not None:
        nb = np.ascontiguousarray(board, dtype=np.int8)
    rm = (C.c_int16 * max(1, len(recent)))(*[m for _, m in recent])
    rcol = (C.c_int8 * max(1, len(recent)))(*[c for c, _ in recent])
    L.or_pos_from_board(C.byref(pos), N,
                        nb.ctypes.data_as(C.POINTER(C.c_int8)) if nb is not None else None,
                        n, komi, caps[0], caps[1], ko, to_play, len(recent), rm, rcol)
    return pos


def env(N):
    e = OEnv()
    lib().or_env_init(C.byref(e), N)
    return e


def play(pos, a):
    out = OPos()
    rcode = lib().or_play_move(C.byref(pos), a, C.byref(out))
    return rcode, out


def legal_moves(pos):
    out = np.zeros(pos.A, dtype=np.int8)
    lib().or_all_legal_moves(C.byref(pos), out.ctypes.data_as(C.POINTER(C.c_int8)))
    return out


def feats(pos):
    P = pos.N * pos.N
    out = np.zeros(17 * P, dtype=np.float32)
    lib().or_get_feats(C.byref(pos), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out.reshape(17, P)


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def node_arr(ptr, A):
    return np.ctypeslib.as_array(ptr, shape=(A,))


class DummyNet:
    """test/test_mcts_player.jl:10-32 -- constant priors and value."""

    def __init__(self, A, fake_priors=None, fake_value=0.0):
        self.A = A
        self.priors = (np.ones(A) / A if fake_priors is None else np.asarray(fake_priors)).astype(np.float32)
        self.value = np.float32(fake_value)
        self.calls = 0
        self.positions = 0

        def _fn(ctx, positions, B, pi, v):
            self.calls += 1
            self.positions += B
            for b in range(B):
                for a in range(self.A):
                    pi[b * self.A + a] = self.priors[a]
                v[b] = self.value

        self.cb = NET_FN(_fn)
