"""alphago.jl_amd -- MI355X-native self-play engine for AlphaGo.jl's hot path.

The directory name contains a dot, so it cannot be imported by name; use the repo-root shim
`import alphago_jl_amd` (which loads this package) or importlib.  Everything computational is
in libagz.so (csrc/, HIP for gfx950); this package is the host-side mirror of the reference's
Julia call surface over the C ABI of include/agz.h."""
from . import _lib
from ._lib import AgzError, IllegalMove, load
from .engine import Engine

__all__ = ["Engine", "AgzError", "IllegalMove", "load", "_lib"]
