#!/usr/bin/env python3
"""Micro-benchmarks of the network kernels on one MI355X (HIP events on the engine's stream):
the dominant 3x3 256->256 conv launch and the whole forward, at the batch sizes of the bench."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alphago_jl_amd as ag  # noqa: E402

PEAK = 157.3

ap = argparse.ArgumentParser()
ap.add_argument("--board", type=int, default=9)
ap.add_argument("--tower", type=int, default=10)
ap.add_argument("--batches", type=int, nargs="+", default=[1024, 4096, 8192])
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--algos", type=int, nargs="+", default=[1, 0], help="1 = winograd, 0 = direct")
ap.add_argument("--precision", default="f32", choices=["f32", "f16", "f32s"], help="f16 = fp16-operand tower (algos ignored)")
ap.add_argument("--tower-persistent", type=int, default=0, help="1 = the whole tower as one persistent launch (k_wino_tower)")
args = ap.parse_args()

N, t = args.board, args.tower
P = N * N
eng = ag.Engine(board_size=N, tower_height=t, games=1, num_readouts=1, max_nodes_per_game=8)
eng.init_synthetic(0)
eng.set_precision(args.precision)
eng.set_tower_persistent(bool(args.tower_persistent))
if args.precision == "f16":
    args.algos = [1]
    PEAK = 2500.0
fe = 2.0 * P * (9 * 17 * 256 + t * 2 * 9 * 256 * 256) + 2.0 * P * 256 * 3 + 2.0 * (2 * P * (P + 1) + P * 256 + 256)
for wino, B in [(w, b) for w in args.algos for b in args.batches]:
    eng.set_winograd(wino)
    conv_ms = eng.time_conv(B, args.iters * 4)
    fwd_ms = eng.time_forward(B, args.iters)
    conv_tf = 2.0 * B * P * 9 * 256 * 256 / (conv_ms * 1e-3) / 1e12
    fwd_tf = B * fe / (fwd_ms * 1e-3) / 1e12
    print(json.dumps({"forward_ms": fwd_ms, "conv_ms_same_layer_loop": conv_ms, "board": N, "tower": t, "B": B, "algo": "fp16 implicit GEMM" if args.precision == "f16" else "winograd F(3x3,3x3)" if wino else "direct implicit GEMM", "conv_TFLOPs": conv_tf,
                      "conv_frac_of_peak(algorithmic flops / MFMA peak of the precision)": conv_tf / PEAK, "forward_TFLOPs": fwd_tf,
                      "forward_frac_of_peak": fwd_tf / PEAK, "evals_per_s": B / (fwd_ms * 1e-3)}))
eng.close()
