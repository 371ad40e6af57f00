"""Compiler-output invariants the hot kernels' performance rests on, checked without a GPU (hipcc cross-compiles gfx950).

The Winograd GEMMs give one wave per SIMD the CU's whole register file and place their s_waitcnt vmcnt by hand around
LDS-DMA streams.  A register spilled to scratch is reloaded behind an `s_waitcnt vmcnt(0)` -- a wait for every DMA piece
in flight -- and a single such reload per K-loop pass costs percents (HISTORY.md 4g: one extra comparison in the F(4x4,3x3)
kernel's point table compiled to 416 bytes of scratch and +8 % per step).  hipcc decides this per build, so the build is
checked: no scratch in the product instantiations of the two GEMM kernels."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _scratch_sizes(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                            os.path.join(ROOT, "alphago.jl_amd", "csrc", src), "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read()
    sizes = {}
    for m in re.finditer(r"^(_ZN3agz\w+):[^\n]*\n.*?^; ScratchSize: (\d+)", text, flags=re.M | re.S):
        sizes[m.group(1)] = int(m.group(2))
    return sizes


@pytest.mark.skipif(not shutil.which(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src,kernel,bound", [("agz_wino4.hip", "k_wino4_gemm", 0), ("agz_wino.hip", "k_wino_gemm4", 16)])
def test_winograd_gemm_kernels_use_no_scratch(src, kernel, bound):
    """bound: bytes of scratch per lane tolerated -- 0 for the F(4x4,3x3) kernel; k_wino_gemm4 has carried two to four
    dwords of prologue spill (outside its K loop) since round 2, and nothing more may join them"""
    sizes = {k: v for k, v in _scratch_sizes(src).items() if kernel in k}
    assert len(sizes) >= 3, sizes
    assert all(v <= bound for v in sizes.values()), {k: v for k, v in sizes.items() if v > bound}


@pytest.mark.skipif(not shutil.which(HIPCC), reason="no hipcc")
def test_fp16_tower_kernel_uses_no_scratch_in_any_product_form():
    """k_conv3x3_f16_w2 sits at 512 of 512 registers with its weight fragments 17 k-steps ahead in flight: a reload is a wait
    behind all of them.  Round 5: the two direct-epilogue forms of a tower (f32 residual in / f32 out) spilled 29-44
    registers with the 18-deep weight ring and take a 9-deep one; every product form (DBG = 0, RB = 7, DM = 0) now compiles
    to at most one spilled dword (the half-in / half-out form with a residual: 8 bytes, outside its chunk loop)."""
    every = _scratch_sizes("agz_conv16.hip")
    sizes = {k: v for k, v in every.items() if "k_conv3x3_f16_w2ILi0E" in k and "ELi7ELb0ELb0ELi0E" in k}
    assert len(sizes) == 6, sizes
    assert all(v <= 8 for v in sizes.values()), {k: v for k, v in sizes.items() if v > 8}
    # the 2 x 2 form (half-in / half-out layers since round 5): no scratch without a residual, <= 16 bytes with one (two
    # dwords outside its chunk loop); its weight ring is 6 deep because 9 spills 69-138 registers
    quad = {k: v for k, v in every.items() if "k_conv3x3_f16_q" in k and "ELb0EEEv" in k}      # (ZB = false: the product forms)
    assert len(quad) == 2 and all(v <= 16 for v in quad.values()), quad
