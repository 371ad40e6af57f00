#!/usr/bin/env python3
"""agz_train_step timing on one MI355X at the reference's batch size (train.jl:40 batch_size = 32): wall time per
step (host-synchronous call) for BASELINE's two network sizes.  Not a bench line: SURVEY.md 8f row 4."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alphago_jl_amd as ag  # noqa: E402

CONFIGS = ((9, 10, 32), (19, 20, 32), (9, 10, 256))
if len(sys.argv) > 1:                      # train_micro.py <index>: one configuration only (for rocprofv3 --stats)
    CONFIGS = (CONFIGS[int(sys.argv[1])],)
for N, tower, B in CONFIGS:
    eng = ag.Engine(board_size=N, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(0)
    rng = np.random.RandomState(0)
    feats = (rng.rand(B, 17 * N * N) < 0.3).astype(np.float32)
    pi = rng.dirichlet(np.full(N * N + 1, 0.3), size=B).astype(np.float32)
    z = rng.choice([-1.0, 1.0], size=B).astype(np.float32)
    losses = [eng.train_step(feats, pi, z) for _ in range(2)]
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        l = eng.train_step(feats, pi, z)
    dt = (time.perf_counter() - t0) / K
    fwd = 2.0 * B * N * N * (9 * 17 * 256 + tower * 2 * 9 * 256 * 256)
    print(json.dumps({"board": N, "tower": tower, "batch": B, "ms_per_step": 1e3 * dt, "positions_per_s": B / dt,
                      "approx_TFLOPs(3x forward conv flops)": 3 * fwd / dt / 1e12,
                      "loss_first": float(losses[0][0]), "loss_after_7_steps": float(l[0])}))
    eng.close()
