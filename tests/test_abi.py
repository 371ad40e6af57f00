"""The C-ABI shared library loads and exports every symbol include/agz.h declares; without a
GPU it must refuse to create an engine (no CPU fallback).  CPU only (no compute calls)."""
import ctypes as C
import os
import re

import pytest

import alphago_jl_amd as ag

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "agz.h")).read()
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
