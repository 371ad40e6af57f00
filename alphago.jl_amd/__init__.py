"""alphago.jl_amd -- MI355X-native self-play engine for AlphaGo.jl's hot path.

The directory name contains a dot, so it cannot be imported by name; use the repo-root shim
`import alphago_jl_amd` (which loads this package) or importlib.  Everything computational is
in libagz.so (csrc/, HIP for gfx950); this package is the host-side mirror of the reference's
Julia call surface over the C ABI of include/agz.h."""
from . import _lib
from ._lib import AgzError, IllegalMove, load
from .engine import Engine, comm_unique_id
from .api import (BLACK, EMPTY, WHITE, GameRecord, GoEnv, LeafPosition, MCTSPlayer, Momentum, NeuralNet, PlayerMove,
                  Position, SelfPlayPlayer, _train, evaluate, extract_data, get_replay_batch, seed, from_flat, from_kgs, from_sgf, get_feats, load_model, save_model, selfplay, to_flat,
                  to_kgs, to_sgf)
from . import bson_weights
from . import distributed
from .replay import ReplayBuffer

__all__ = ["Engine", "comm_unique_id", "AgzError", "IllegalMove", "load", "_lib", "GoEnv", "Position", "PlayerMove", "NeuralNet",
           "MCTSPlayer", "selfplay", "extract_data", "GameRecord", "SelfPlayPlayer", "LeafPosition", "get_replay_batch",
           "Momentum", "_train", "seed", "get_feats", "to_flat", "from_flat",
           "from_kgs", "to_kgs", "from_sgf", "to_sgf", "BLACK", "WHITE", "EMPTY", "load_model", "save_model", "evaluate",
           "bson_weights", "ReplayBuffer"]
