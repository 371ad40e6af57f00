#!/usr/bin/env python3
"""What the weight broadcast costs ("replace the host staging ... or state the measured ms"): a world of one
rank on this GPU, real RCCL through the C ABI.  Times agz_broadcast_weights (flat pack on the host, H2D, ncclBroadcast,
and on a receiver D2H + unpack) and the first forward after the parameters changed (every inference image rebuilt on the
host and uploaded: Winograd U in float64, folded affines, ...)."""
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
    out.append({"board": N, "tower": tower, "parameters": n, "MB": 4e-6 * n, "broadcast_ms_root_world1": 1e3 * (t1 - t0),
                "first_forward_after_new_weights_ms": 1e3 * (t3 - t2), "forward_ms_warm": 1e3 * (t4 - t3)})
    eng.comm_destroy(comm)
    eng.close()
print(json.dumps(out))
