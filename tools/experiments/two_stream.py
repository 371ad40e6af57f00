#!/usr/bin/env python3
"""Does running two half-batch forwards on two streams hide each other's last workgroup round?  Two engines (two streams,
two copies of the same weights), 4096 positions each, forwards issued from two threads at once, against one engine at
8192.  Wall clock around `iters` forwards, device synchronised inside time_forward."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import alphago_jl_amd as ag  # noqa: E402

N, t, iters = (int(sys.argv[1]), int(sys.argv[2]), 10) if len(sys.argv) > 2 else (9, 10, 10)
FULL = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
HALF = FULL // 2


def mk():
    e = ag.Engine(board_size=N, tower_height=t, games=1, num_readouts=1, max_nodes_per_game=8)
    e.init_synthetic(0)
    e.set_winograd(1)
    return e


e0, e1, e2 = mk(), mk(), mk()
for e, b in ((e0, FULL), (e1, HALF), (e2, HALF)):
    e.time_forward(b, 2)
for rep in range(3):
    t0 = time.perf_counter()
    ms = e0.time_forward(FULL, iters)
    w0 = (time.perf_counter() - t0) * 1e3 / iters
    out = {}
    bar = threading.Barrier(2)

    def run(e, k):
        bar.wait()
        out[k] = e.time_forward(HALF, iters)

    th = [threading.Thread(target=run, args=(e1, 1)), threading.Thread(target=run, args=(e2, 2))]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    w2 = (time.perf_counter() - t0) * 1e3 / iters
    a = e1.time_forward(HALF, iters)
    print("one stream, 8192: %.3f ms per forward (events %.3f) | two streams, 4096 + 4096 at once: %.3f ms per pair (events %.3f / %.3f) | 4096 alone: %.3f"
          % (w0, ms, w2, out[1], out[2], a))
for e in (e0, e1, e2):
    e.close()
