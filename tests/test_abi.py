"""The C-ABI shared library loads and exports every symbol include/agz.h declares; without a
GPU it must refuse to create an engine (no CPU fallback).  CPU only (no compute calls)."""
import ctypes as C
import os
import re

import pytest

import alphago_jl_amd as ag

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "agz.h")).read() + open(os.path.join(ROOT, "include", "agz_debug.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:agz_status|int32_t|int64_t|void|const char\*)\s+(agz_[a-z_0-9]+)\s*\(", hdr, flags=re.M)))


def test_exports_every_declared_symbol():
    L = ag.load()
    names = declared_functions()
    assert len(names) >= 55
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # and the python binding declares a prototype for each of them
    unbound = [n for n in names if n not in L._agz_signatures]
    assert not unbound, unbound
    assert L.agz_version() == 100


def test_config_default_mirrors_reference_defaults():
    c = ag._lib.default_config()
    assert (c.board_size, c.tower_height, c.num_readouts, c.parallel_readouts) == (19, 19, 800, 8)
    assert c.komi == 7.5 and c.c_puct == 0.96 and c.dirichlet_noise_weight == 0.25
    assert c.resign_threshold == -0.9 and c.resign_disable_fraction == 0.05
    assert C.sizeof(ag._lib.Config) == 112


def _c_layout(name, nfields):
    L = ag.load()
    out = (C.c_int32 * 64)()
    n = L.agz_abi_layout(name.encode(), out, 64)
    assert n == nfields, (name, n)
    return out[0], list(out[1:1 + n])


JL_SIZES = {"Int32": 4, "UInt32": 4, "Float32": 4, "Int64": 8, "UInt64": 8, "Float64": 8}


def _julia_struct_layout(text, name, nested=()):
    """field offsets of a Julia `struct` whose fields are C-compatible bits types, by the C layout rules Julia
    applies to isbits structs (natural alignment, size rounded up to the largest alignment)"""
    body = re.search(r"^struct " + name + r"\b(.*?)^end", text, flags=re.S | re.M).group(1)
    body = re.sub(r"#.*", "", body)
    fields = re.findall(r"(\w+)::(\w+)", body)
    off, offs, maxal = 0, [], 1
    for _, t in fields:
        if t in JL_SIZES:
            sz = al = JL_SIZES[t]
        else:
            sz, al = dict(nested)[t]
        off = (off + al - 1) // al * al
        offs.append(off)
        off += sz
        maxal = max(maxal, al)
    return (off + maxal - 1) // maxal * maxal, offs, maxal


def test_pod_layouts_agree_between_c_ctypes_and_julia():
    """the ctypes Structures and the Julia structs of AlphaGoMI.jl (which has never been
    run here: no julia binary) are mirrors of include/agz.h by hand.  The library reports sizeof/offsetof of
    every POD as compiled (agz_abi_layout); both mirrors must reproduce them field for field."""
    lib = ag._lib
    jl = open(os.path.join(ROOT, "alphago.jl_amd", "julia", "AlphaGoMI.jl")).read()
    nested = []
    for cname, ct, jname in (("agz_config", lib.Config, "AgzConfig"), ("agz_stats", lib.Stats, "AgzStats"),
                             ("agz_game_header", lib.GameHeader, "AgzGameHeader"),
                             ("agz_position_info", lib.PositionInfo, "AgzPositionInfo"),
                             ("agz_node_info", lib.NodeInfo, "AgzNodeInfo")):
        size, offs = _c_layout(cname, len(ct._fields_))
        assert C.sizeof(ct) == size, cname
        assert [getattr(ct, f[0]).offset for f in ct._fields_] == offs, cname
        jsize, joffs, jal = _julia_struct_layout(jl, jname, nested)
        assert jsize == size and joffs == offs, (jname, jsize, size, joffs, offs)
        nested.append((jname, (jsize, jal)))
    assert _c_layout("agz_config", 21)[0] == 112
    assert ag.load().agz_abi_layout(b"no_such_struct", (C.c_int32 * 4)(), 4) == -1


def test_comm_entry_points_fail_loudly_without_a_gpu():
    """the RCCL exchange is bound at run time: on a GPU-less host the library still loads and exports the
    symbols (checked above); creating a communicator needs an engine, which does not exist here"""
    L = ag.load()
    assert L.agz_replay_count(None) == -1
    assert L.agz_allgather_records(None, None, None) == ag._lib.BAD_ARGUMENT
    out = C.c_void_p()
    idb = (C.c_uint8 * 128)()
    assert L.agz_comm_create(None, 0, 1, idb, C.byref(out)) == ag._lib.BAD_ARGUMENT and not out.value


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is for GPU-less hosts")
    with pytest.raises(ag.AgzError) as ei:
        ag.Engine(board_size=9, games=1, tower_height=1)
    assert ei.value.status == ag._lib.HIP_ERROR
    assert "no CPU fallback" in str(ei.value)


def test_draw_header_is_shared_not_copied():
    """the engine and the oracle must include the same draw-stream header"""
    for rel in ("alphago.jl_amd/csrc/agz_search.h", "oracle/agz_oracle_mcts.c"):
        assert "include/agz_draws.h" in open(os.path.join(ROOT, rel)).read()
