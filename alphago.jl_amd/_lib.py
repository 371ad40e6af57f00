"""ctypes binding of libagz.so (include/agz.h).  The library is built in-tree by
__graft_entry__.build() / `make -C alphago.jl_amd/csrc`; there is NO fallback: a missing
library or a missing gfx950 device raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libagz.so")

OK, ILLEGAL_MOVE, ASSERT_DONE_NODE, HISTORY_INCOMPLETE, BAD_SHAPE, ASSERT_SOFTPICK = range(6)
BAD_ARGUMENT, HIP_ERROR, POOL_EXHAUSTED, RCCL_ERROR, NOT_READY = 6, 7, 8, 9, 10

L_VALUE_CONV, L_POLICY_CONV, L_VALUE_FC1, L_VALUE_FC2, L_POLICY_FC = -1, -2, -3, -4, -5
K_WEIGHT, K_BIAS, K_BN_BETA, K_BN_GAMMA, K_BN_MEAN, K_BN_VAR, K_BN_EPS = range(7)
F_CHILD_N, F_CHILD_W, F_CHILD_PRIOR = 0, 1, 2


class AgzError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"agz status {status}: {message}")
        self.status = status


class IllegalMove(AgzError):
    """IllegalMove, src/AlphaGo.jl:8"""


class Config(C.Structure):
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


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "steps", "positions", "games_started", "games_finished", "evals", "duplicate_evals",
        "terminal_visits", "root_visits", "nodes_in_use", "pool_exhausted", "resigned_games", "live_games",
        "records_dropped", "pool_short_searches", "peak_nodes_per_game", "stalled_games", "node_capacity",
        "abandoned_games")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class GameHeader(C.Structure):
    _fields_ = [("game_id", C.c_uint64), ("num_moves", C.c_int32), ("result", C.c_int32),
                ("was_resign", C.c_int32), ("resign_disabled", C.c_int32), ("final_score", C.c_float),
                ("short_searches", C.c_int32)]


class PositionInfo(C.Structure):
    _fields_ = [("n", C.c_int32), ("to_play", C.c_int32), ("ko", C.c_int32), ("caps_black", C.c_int32),
                ("caps_white", C.c_int32), ("last_move", C.c_int32), ("prev_move", C.c_int32),
                ("history_len", C.c_int32), ("komi", C.c_float)]


class NodeInfo(C.Structure):
    _fields_ = [("N", C.c_float), ("W", C.c_float), ("Q", C.c_float), ("parent", C.c_int32),
                ("fmove", C.c_int32), ("is_expanded", C.c_int32), ("losses_applied", C.c_int32),
                ("done", C.c_int32), ("pos", PositionInfo)]


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (+ HSA runtime, comgr).  If libagz pulls
    in /opt/rocm's copy first and torch is imported later in the same process, torch finds a
    half-foreign runtime and reports "No HIP GPUs are available"; the other order works (libagz binds
    to the runtime torch loaded).  So when torch is installed but not yet imported, load ITS
    libamdhip64 first -- by path, without importing torch (that costs seconds to minutes).
    AGZ_SYSTEM_HIP=1 keeps the system runtime (torch.cuda is then unusable in this process)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("AGZ_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else None
        if path and os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """dlopen libagz.so and declare every prototype of include/agz.h"""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    E = C.c_void_p
    P = C.POINTER
    i32, i64, u32, u64, f32, f64 = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double
    i8p, i16p, i32p, f32p, f64p = P(C.c_int8), P(C.c_int16), P(i32), P(f32), P(f64)
    sig = {
        "agz_version": (i32, []),
        "agz_config_default": (None, [P(Config)]),
        "agz_engine_create": (i32, [P(Config), P(E)]),
        "agz_engine_destroy": (None, [E]),
        "agz_last_error": (C.c_char_p, [E]),
        "agz_engine_sync": (i32, [E]),
        "agz_net_set_weights": (i32, [E, i32, i32, f32p, i64]),
        "agz_net_param_count": (i64, [E, i32, i32]),
        "agz_net_get_weights": (i32, [E, i32, i32, f32p, i64]),
        "agz_net_init_synthetic": (i32, [E, u64]),
        "agz_net_forward": (i32, [E, i8p, i8p, i32p, i8p, i32, f32p, f32p]),
        "agz_net_forward_features": (i32, [E, f32p, i32, f32p, f32p]),
        "agz_features": (i32, [E, i8p, i8p, i32p, i8p, i32, f32p]),
        "agz_net_time_forward": (i32, [E, i32, i32, f32p]),
        "agz_net_time_conv": (i32, [E, i32, i32, f32p]),
        "agz_net_set_winograd": (i32, [E, i32]),
        "agz_net_set_tower_persistent": (i32, [E, i32]),
        "agz_net_set_tower_streams": (i32, [E, i32]),
        "agz_net_set_precision": (i32, [E, i32]),
        "agz_profile_conv_enable": (i32, [E, i32]),
        "agz_profile_conv_read": (i32, [E, f64p, f64p, P(i64)]),
        "agz_profile_search_enable": (i32, [E, i32]),
        "agz_profile_search_read": (i32, [E, f64p, P(i64)]),
        "agz_go_play": (i32, [E, i8p, i8p, i32p, i32p, i32, i8p, i32p, i32p, i32p]),
        "agz_go_legal": (i32, [E, i8p, i8p, i32p, i32, i8p]),
        "agz_go_score": (i32, [E, i8p, f32p, i32, f32p]),
        "agz_selfplay_start": (i32, [E, i64]),
        "agz_selfplay_step": (i32, [E, i32]),
        "agz_engine_stats": (i32, [E, P(Stats)]),
        "agz_selfplay_select": (i32, [E, i32p]),
        "agz_selfplay_leaf_features": (i32, [E, f32p]),
        "agz_selfplay_incorporate": (i32, [E, f32p, f32p]),
        "agz_records_count": (i64, [E]),
        "agz_records_header": (i32, [E, i64, P(GameHeader)]),
        "agz_records_game": (i32, [E, i64, i16p, f32p, f32p]),
        "agz_records_packed_size": (i32, [E, P(i64)]),
        "agz_records_export_packed": (i32, [E, C.c_void_p, i64, i32]),
        "agz_records_clear": (i32, [E]),
        "agz_slot_status": (i32, [E, i32p, i32p, i32p]),
        "agz_slot_abandon": (i32, [E, i32]),
        "agz_arena_counts": (i32, [E, i32p]),
        "agz_net_select": (i32, [E, i32]),
        "agz_records_features": (i32, [E, i64, f32p]),
        "agz_replay_features": (i32, [E, i16p, i64, i32p, i32p, i32, C.c_void_p, i32]),
        "agz_replay_ingest_packed": (i32, [E, C.c_void_p, i64, i32, P(i64)]),
        "agz_replay_ingest_gathered": (i32, [E, C.c_void_p, i32, i32, i64, P(i64), P(i64)]),
        "agz_replay_count": (i64, [E]),
        "agz_replay_positions": (i64, [E]),
        "agz_replay_header": (i32, [E, i64, P(GameHeader)]),
        "agz_replay_game": (i32, [E, i64, i16p, f32p, f32p]),
        "agz_replay_trim": (i32, [E, i64]),
        "agz_replay_clear": (i32, [E]),
        "agz_replay_batch": (i32, [E, P(i64), i32p, i32, C.c_void_p, C.c_void_p, C.c_void_p, i32]),
        "agz_train_step": (i32, [E, C.c_void_p, C.c_void_p, C.c_void_p, i32, i32, f32, f32, f32p]),
        "agz_train_reset": (i32, [E]),
        "agz_comm_unique_id": (i32, [P(C.c_uint8)]),
        "agz_comm_create": (i32, [E, i32, i32, P(C.c_uint8), P(E)]),
        "agz_comm_destroy": (None, [E]),
        "agz_allgather_records": (i32, [E, E, P(i64)]),
        "agz_broadcast_weights": (i32, [E, E, i32, P(i64)]),
        "agz_gather_plan": (i32, [P(i64), i32, P(i64), P(i64)]),
        "agz_abi_layout": (i32, [C.c_char_p, i32p, i32]),
        "agz_tree_init": (i32, [E, i32, i8p, P(PositionInfo), i8p]),
        "agz_tree_root": (i32, [E, i32, i32p]),
        "agz_tree_select_leaf": (i32, [E, i32, i32, i32p]),
        "agz_tree_maybe_add_child": (i32, [E, i32, i32, i32, i32p]),
        "agz_tree_add_virtual_loss": (i32, [E, i32, i32, i32]),
        "agz_tree_revert_virtual_loss": (i32, [E, i32, i32, i32]),
        "agz_tree_incorporate": (i32, [E, i32, i32, f32p, i32, f32, i32]),
        "agz_tree_inject_noise": (i32, [E, i32, i32]),
        "agz_tree_search": (i32, [E, i32, i32, i32p]),
        "agz_tree_search_select": (i32, [E, i32, i32, i32p]),
        "agz_tree_leaf_features": (i32, [E, i32, f32p]),
        "agz_tree_leaf_positions": (i32, [E, i32, i32p, i8p, i8p, i32p, i8p, P(PositionInfo)]),
        "agz_tree_search_incorporate": (i32, [E, i32, f32p, f32p]),
        "agz_tree_pick_move": (i32, [E, i32, i32p]),
        "agz_tree_play_move": (i32, [E, i32, i32, i32p]),
        "agz_tree_should_resign": (i32, [E, i32, i32p]),
        "agz_tree_is_done": (i32, [E, i32, i32, i32p]),
        "agz_tree_node_info": (i32, [E, i32, i32, P(NodeInfo)]),
        "agz_tree_node_floats": (i32, [E, i32, i32, i32, f32p]),
        "agz_tree_node_scores": (i32, [E, i32, i32, f64p]),
        "agz_tree_node_set_floats": (i32, [E, i32, i32, i32, f32p]),
        "agz_tree_node_set_N": (i32, [E, i32, i32, f32]),
        "agz_tree_node_set_n": (i32, [E, i32, i32, i32]),
        "agz_tree_node_children": (i32, [E, i32, i32, i32p]),
        "agz_tree_node_board": (i32, [E, i32, i32, i8p]),
        "agz_tree_pending_vlosses": (i32, [E, i32, i32p]),
        "agz_tree_set_draw": (i32, [E, i32, u64, u32]),
        "agz_debug_draws": (i32, [E, u64, u64, u32, i32, f64, f64p]),
        "agz_debug_math": (i32, [E, i32, f64p, f64p, i32, f64p]),
        "agz_debug_counters": (i32, [E, C.POINTER(C.c_uint64), i32]),
        "agz_debug_set_stagger": (i32, [E, i32]),
        "agz_debug_live_record": (i32, [E, i32, i32, P(u64), i32p, i32p, f32p, f32p]),
        "agz_debug_mfma_sustained": (i32, [E, i32, C.POINTER(C.c_float)]),
        "agz_debug_pack_diff": (i32, [E, i32, P(i64)]),
        "agz_debug_mfma_sustained_data": (i32, [E, i32, i32, C.POINTER(C.c_float)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._agz_signatures = sig
    _lib = L
    return L


def default_config(**kw):
    c = Config()
    load().agz_config_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise TypeError(f"unknown agz_config field {k!r}")
        setattr(c, k, v)
    return c
