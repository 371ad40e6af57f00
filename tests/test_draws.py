"""The injected draw stream (include/agz_draws.h): accuracy of its deterministic log/exp/pow,
uniformity of its integers, and the Gamma/Dirichlet sampler's moments.  CPU only; the GPU
bit-equality check lives in tests/test_gpu_draws.py."""
import ctypes as C
import math

import numpy as np
import pytest

import orc

L = orc.lib()
for name, res, args in (
    ("or_draw_u64", C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64]),
    ("or_draw_u01", C.c_double, [C.c_uint64]),
    ("or_draw_index", C.c_uint32, [C.c_uint64, C.c_uint32]),
    ("or_det_log", C.c_double, [C.c_double]),
    ("or_det_exp", C.c_double, [C.c_double]),
    ("or_det_pow", C.c_double, [C.c_double, C.c_double]),
    ("or_det_sqrt", C.c_double, [C.c_double]),
    ("or_dirichlet_gamma", C.c_double, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double]),
):
    getattr(L, name).restype = res
    getattr(L, name).argtypes = args


def test_det_math_accuracy():
    rng = np.random.RandomState(0)
    for x in np.concatenate([10.0 ** rng.uniform(-300, 300, 2000), rng.uniform(0.5, 2, 2000), [1.0, 2.0, 5e-324]]):
        assert L.or_det_log(x) == pytest.approx(math.log(x), rel=4e-16, abs=4e-16)
    for x in np.concatenate([rng.uniform(-700, 700, 2000), rng.uniform(-1, 1, 2000), [0.0]]):
        assert L.or_det_exp(x) == pytest.approx(math.exp(x), rel=1e-14)
    assert L.or_det_exp(-800.0) == 0.0
    for n in range(1, 2000):
        assert L.or_det_pow(float(n), 0.98) == pytest.approx(n ** 0.98, rel=1e-14)
    assert L.or_det_pow(0.0, 0.98) == 0.0
    for x in 10.0 ** rng.uniform(-20, 20, 1000):
        assert L.or_det_sqrt(x) == pytest.approx(math.sqrt(x), rel=1e-15)


def test_u01_and_index():
    us = np.array([L.or_draw_u01(L.or_draw_u64(1, 2, 3, 1, i)) for i in range(20000)])
    assert 0 < us.min() and us.max() < 1
    assert abs(us.mean() - 0.5) < 0.01 and abs(us.var() - 1 / 12) < 0.005
    idx = np.array([L.or_draw_index(L.or_draw_u64(1, 2, 3, 2, i), 7) for i in range(14000)])
    counts = np.bincount(idx, minlength=7)
    assert counts.min() > 1800 and counts.max() < 2200
    # keys matter
    assert L.or_draw_u64(1, 2, 3, 1, 0) != L.or_draw_u64(1, 2, 3, 2, 0)
    assert L.or_draw_u64(1, 2, 3, 1, 0) != L.or_draw_u64(1, 3, 3, 1, 0)
    assert L.or_draw_u64(1, 2, 3, 1, 0) == L.or_draw_u64(1, 2, 3, 1, 0)


@pytest.mark.parametrize("alpha", [0.029917, 0.13207, 0.4165, 1.7])
def test_gamma_moments(alpha):
    g = np.array([L.or_dirichlet_gamma(9, k // 400, 0, k % 400, alpha) for k in range(40000)])
    assert (g >= 0).all()
    assert g.mean() == pytest.approx(alpha, rel=0.06)
    assert g.var() == pytest.approx(alpha, rel=0.12)
    from scipy import stats
    # Kolmogorov-Smirnov against the exact law (on a log scale the tiny-alpha mass is at ~0)
    ks = stats.kstest(g[g > 1e-300], lambda x: stats.gamma.cdf(x, alpha))
    assert ks.statistic < 0.03
