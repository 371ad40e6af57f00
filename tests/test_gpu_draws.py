"""gfx950 and the host must produce BIT-IDENTICAL draws and Float32/Float64 PUCT arithmetic:
the whole "identical visit counts under a fixed RNG" claim rests on it."""
import ctypes as C

import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
import test_draws  # declares the or_draw_* prototypes

pytestmark = pytest.mark.gpu
L = orc.lib()


@pytest.fixture(scope="module")
def eng():
    e = ag.Engine(board_size=9, games=1, tower_height=0, num_readouts=8, max_nodes_per_game=16)
    yield e
    e.close()


@pytest.mark.parametrize("alpha", [0.029917, 0.13207, 0.4165])
def test_dirichlet_gammas_bit_equal(eng, alpha):
    alpha = float(np.float32(alpha))
    for game, move in ((0, 0), (17, 5), (123456789, 112)):
        g = eng.debug_draws(7, game, move, 362, alpha)
        ref = np.array([L.or_dirichlet_gamma(7, game, move, a, alpha) for a in range(362)])
        assert (g.view(np.uint64) == ref.view(np.uint64)).all()


def test_det_math_bit_equal(eng):
    rng = np.random.RandomState(0)
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 4000), rng.uniform(0.5, 2, 4000), np.arange(1, 2000.0)])
    for op, fn in ((0, L.or_det_log), (2, lambda t: L.or_det_pow(t, 0.98))):
        got = eng.debug_math(op, x)
        ref = np.array([fn(float(t)) for t in x])
        assert (got.view(np.uint64) == ref.view(np.uint64)).all(), op
    xe = np.concatenate([rng.uniform(-745, 709, 4000), rng.uniform(-1, 1, 4000)])
    got = eng.debug_math(1, xe)
    ref = np.array([L.or_det_exp(float(t)) for t in xe])
    assert (got.view(np.uint64) == ref.view(np.uint64)).all()


def test_f32_div_sqrt_and_puct_are_ieee(eng):
    """Float32 sqrt / divide must be correctly rounded on the device (numpy float32 is), and the
    mixed-precision PUCT score must match the host formula bit for bit (no FMA contraction)."""
    rng = np.random.RandomState(1)
    x = rng.uniform(0, 5000, 20000).astype(np.float32).astype(np.float64)
    y = rng.uniform(0.5, 3000, 20000).astype(np.float32).astype(np.float64)
    got = eng.debug_math(3, x)
    assert (got == np.sqrt(x.astype(np.float32)).astype(np.float64)).all()
    got = eng.debug_math(4, x, y)
    assert (got == (x.astype(np.float32) / y.astype(np.float32)).astype(np.float64)).all()
    # op 5: W=x (signed), N=y integer counts
    w = (x - 2500).astype(np.float32)
    n = np.floor(y).astype(np.float32)
    got = eng.debug_math(5, w.astype(np.float64), n.astype(np.float64))
    denom = np.float32(1) + n
    qs = (w / denom) * np.float32(-1)
    scale = 0.96 * np.sqrt(np.float32(1) + (n + np.float32(7))).astype(np.float64)
    ref = qs.astype(np.float64) + (scale * np.float64(np.float32(0.25))) / denom.astype(np.float64)
    assert (got == ref).all()
