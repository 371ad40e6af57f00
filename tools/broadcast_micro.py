#!/usr/bin/env python3
"""What new weights cost before the next forward (/root/reference/src/train.jl:67-74 changes the weights every
iteration): a world of one rank on this GPU, real RCCL through the C ABI.  Times agz_broadcast_weights (round 5: one
ncclBroadcast of the device master copy, in place), the first forward after one array was set through
agz_net_set_weights, and the first forward after an agz_train_step -- in every case all inference images (direct,
F(3x3,3x3) / F(4x4,3x3) U in float64, folded affines) are rebuilt by kernels from the device master."""
import json
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import alphago_jl_amd as ag  # noqa: E402

out = []
for N, tower in ((9, 10), (19, 20)):
    eng = ag.Engine(board_size=N, tower_height=tower, games=1, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(1)
    feats = np.zeros((8, 17 * N * N), np.float32)
    eng.forward_features(feats)
    comm = eng.comm_create(0, 1, ag.comm_unique_id())
    eng.broadcast_weights(comm, 0)
    t0 = time.perf_counter()
    n = eng.broadcast_weights(comm, 0)
    t1 = time.perf_counter()
    w = eng.get_weights(1, 0)
    eng.set_weights(1, 0, w)                       # marks every pack stale
    t2 = time.perf_counter()
    eng.forward_features(feats)
    t3 = time.perf_counter()
    eng.forward_features(feats)
    t4 = time.perf_counter()
    rng = np.random.RandomState(0)
    B = 8
    pi = rng.dirichlet(np.full(N * N + 1, 0.3), size=B).astype(np.float32)
    z = rng.choice([-1.0, 1.0], size=B).astype(np.float32)
    eng.train_step(feats, pi, z)                   # (allocations, first-use packs of the trainer)
    eng.forward_features(feats)
    t5 = time.perf_counter()
    eng.train_step(feats, pi, z)
    t6 = time.perf_counter()
    eng.forward_features(feats)
    t7 = time.perf_counter()
    t8 = time.perf_counter()
    w_back = eng.get_weights(1, 0)                 # the host copies catch up only when somebody asks
    t9 = time.perf_counter()
    out.append({"board": N, "tower": tower, "parameters": n, "MB": 4e-6 * n, "broadcast_ms_root_world1": 1e3 * (t1 - t0),
                "first_forward_after_new_weights_ms": 1e3 * (t3 - t2), "forward_ms_warm": 1e3 * (t4 - t3),
                "train_step_ms_B8": 1e3 * (t6 - t5), "first_forward_after_train_step_ms": 1e3 * (t7 - t6),
                "first_get_weights_after_train_step_ms": 1e3 * (t9 - t8)})
    eng.comm_destroy(comm)
    eng.close()
print(json.dumps(out))
