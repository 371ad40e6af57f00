"""agz_train_step (the training step of /root/reference/src/neural_net.jl:75-101 on the
device) against the float64 autograd twin tests/train_twin.py.  The reference cannot run this step (it is
broken at HEAD), so the twin is the pin; the twin itself is tied to the pinned oracle by its
inference-mode forward (tests/test_train_twin.py, CPU).

Bars: losses within 1e-5 relative; BatchNorm running statistics 1e-5; every parameter's update (theta_new -
theta_old), as a fraction of the largest update of its tensor and after two f32 ulps of the parameter itself (it is
stored in f32): within 1e-5 -- or, where f32 arithmetic itself cannot do that, within 4x the error a plain PyTorch
float32 autograd of the same step makes on the same tensor.  Measured on MI355X (round 3): <= 1e-6 at the toy shapes
(round 2's bar was 2e-3, which would have hidden a dropped tap on a small tensor); at the reference's shape (9x9, 21
stacked convolutions, batch 32) 40 of the 98 tensors are above 1e-5, the worst 4.8e-2 where torch float32 is 2.8e-2 on
the same tensor (worst ratio 2.8): the gradient reaches the first layers through 20 BatchNorm backward passes, each a
cancellation (dy - mean(dy) - xhat mean(dy xhat)) -- f32 roundoff class, not a kernel defect.  At batch 128 (9x9, tower
2: launch_conv3x3_direct and the row-split k_wgrad3x3 + k_sum_parts) the worst is 4.2e-4 on one conv weight tensor
(an f32 chain of 10,368 products per element, where torch's blocked summation stays within the parameter's ulp): floor
2e-3 there."""
import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from test_hostsim_go import random_positions
from train_twin import K_MEAN, K_VAR, Twin

pytestmark = pytest.mark.gpu


def batch(N, B, seed):
    rng = np.random.RandomState(seed)
    positions = random_positions(N, max(3, (B + 29) // 30), 40, seed=seed)
    positions = [positions[i] for i in rng.choice(len(positions), B, replace=False)]
    feats = np.stack([orc.feats(p).reshape(-1) for p in positions]).astype(np.float32)
    pi = rng.dirichlet(np.full(N * N + 1, 0.3), size=B).astype(np.float32)
    pi[0, : N] = 0.0                                   # exact zeros in a target are legal (0 * log p = 0)
    pi[0] /= pi[0].sum()
    z = rng.choice([-1.0, 1.0], size=B).astype(np.float32)
    return feats, pi, z


# (N, tower, B): the two toy shapes of round 2; tower 3 (two stacked residual backward paths, `dsc` accumulation);
# the reference's own `_train` shape -- 9x9 board, tower_height = 10 here as in BASELINE configs[1] (train.jl:38-40 has
# batch_size = 32; its tower_height = 19 default differs only in depth); and B = 128 at 9x9, where the forward / dgrad
# take launch_conv3x3_direct (no tap split) and the weight gradient the row-split k_wgrad3x3 + k_sum_parts path.
CASES = [(5, 1, 8), (9, 2, 6), (5, 3, 8), (9, 10, 32), (9, 2, 128)]
# floor of the update bar per case = 10x the worst measured on MI355X (printed by the test): toy shapes measure <= 1e-6
ABS_BAR = {(5, 1, 8): 1e-5, (9, 2, 6): 1e-5, (5, 3, 8): 1e-5, (9, 10, 32): 1e-5, (9, 2, 128): 2e-3}
F32_FACTOR = 4.0      # above the floor: no further from float64 than 4x what torch float32 autograd is on that tensor


@pytest.mark.parametrize("N,tower,B", CASES)
def test_train_step_matches_float64_twin(N, tower, B):
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(7)
    rng = np.random.RandomState(1)
    for l in list(range(1 + 2 * tower)) + [-1, -2]:    # non-trivial biases / BatchNorm parameters
        n = eng.param_count(l, 1)
        eng.set_weights(l, 1, rng.uniform(-0.2, 0.2, n).astype(np.float32))
        eng.set_weights(l, 2, rng.uniform(-0.3, 0.3, n).astype(np.float32))
        eng.set_weights(l, 3, rng.uniform(0.5, 1.5, n).astype(np.float32))
    import torch
    twin = Twin(N, tower, eng.get_weights)
    twin32 = Twin(N, tower, eng.get_weights, dtype=torch.float32)      # the plain-PyTorch-fp32 yardstick
    before = {key: eng.get_weights(*key).copy() for key in eng.layers()}
    worst, worst32 = {}, {}                            # (layer, kind) -> worst |d update| / largest update of the tensor
    floor, bad = ABS_BAR[(N, tower, B)], []
    for it in range(2):                                # the second step exercises the Momentum velocity
        feats, pi, z = batch(N, B, 10 + it)
        got = eng.train_step(feats, pi, z)
        want = twin.step(feats, pi, z)
        twin32.step(feats, pi, z)
        assert np.allclose(got, want, rtol=1e-5, atol=1e-9), (it, got, want)
        for (l, k) in eng.layers():
            new = eng.get_weights(l, k)
            if k == 6:
                continue
            ref = twin.param(l, k)
            if k in (K_MEAN, K_VAR):
                # running statistics: 1e-5, or (second step, after parameters that already differ in their last bits) what
                # the float32 twin itself is away from float64
                tol = 1e-6 + 1e-5 * np.abs(ref)
                d, d32 = np.abs(new - ref), np.abs(twin32.param(l, k).astype(np.float64) - ref)
                if not (d <= np.maximum(tol, F32_FACTOR * d32.max())).all():
                    bad.append((it, l, k, float(d.max()), float(d32.max())))
                continue
            upd, upd_ref = new.astype(np.float64) - before[(l, k)], ref - before[(l, k)]
            scale = np.abs(upd_ref).max()
            ulp = 2.0 ** -23 * max(np.abs(ref).max(), 1e-30)        # the parameter itself is stored in f32
            err = np.abs(upd - upd_ref).max()
            err32 = np.abs(twin32.param(l, k).astype(np.float64) - before[(l, k)] - upd_ref).max()
            rel = max(err - 2 * ulp, 0.0) / max(scale, 1e-300)
            rel32 = max(err32 - 2 * ulp, 0.0) / max(scale, 1e-300)
            worst[(l, k)] = max(worst.get((l, k), 0.0), rel)
            worst32[(l, k)] = max(worst32.get((l, k), 0.0), rel32)
            if rel > max(floor, F32_FACTOR * rel32):
                bad.append((it, l, k, rel, rel32))
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"\n[train parity {N}x{N} tower {tower} B {B}] worst update error / largest update (torch-f32 autograd on the "
          f"same tensor): " + ", ".join(f"layer {l} kind {k}: {e:.2e} ({worst32[(l, k)]:.2e})" for (l, k), e in top)
          + f"; tensors above {floor:g}: {sum(e > floor for e in worst.values())} of {len(worst)}; worst ratio to torch-f32 "
          f"among those: {max([e / max(worst32[key], 1e-300) for key, e in worst.items() if e > floor] or [0.0]):.2f}")
    assert not bad, bad[:8]
    # the step really moved the network, and inference now runs with the new parameters
    assert any(np.abs(eng.get_weights(*key) - before[key]).max() > 0 for key in eng.layers() if key[1] == 0)
    feats, _, _ = batch(N, B, 99)
    gpi, gv = eng.forward_features(feats)
    with np.errstate(all="ignore"):
        import torch
        logp, v = twin.forward(feats, False)
    assert np.abs(gpi - np.exp(logp.detach().numpy())).max() <= 1e-4 and np.abs(gv - v.detach().numpy()).max() <= 1e-4
    eng.close()


def test_train_step_argument_checks_and_reset():
    eng = ag.Engine(board_size=5, games=1, tower_height=1, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(0)
    feats, pi, z = batch(5, 4, 3)
    with pytest.raises(ag.AgzError):
        eng.train_step(feats[:1], pi[:1], z[:1])       # BatchNorm needs a batch
    a = eng.train_step(feats, pi, z)
    w1 = eng.get_weights(1, 0).copy()
    eng.train_reset()
    eng.init_synthetic(0)
    b = eng.train_step(feats, pi, z)                   # same start, fresh optimiser: the same step
    assert (a == b).all() and (eng.get_weights(1, 0) == w1).all()
    assert a[0] == pytest.approx(a[1] + a[2] + a[3], rel=1e-6) and a[3] > 0
    eng.close()
