#!/usr/bin/env python3
"""What this board sustains on MFMAs alone, by operand data (agz_debug_mfma_sustained / _data): a matrix pipe's power follows
its operands' toggling, so the ceiling of a layer depends on what it multiplies.  Prints TFLOP/s, with socket W and sclk."""
import importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alphago_jl_amd as ag
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
eng = ag.Engine(board_size=9, tower_height=1, games=1, num_readouts=1, max_nodes_per_game=8)
names = {0: "f32 MFMA, constant operands (1 + 1e-6 tid, 0.5)", 1: "f32 MFMA, random A, half-zero B", 3: "f32 MFMA, random A, dense random B",
         2: "fp16 MFMA 32x32x16, random A, half-zero B", 4: "fp16 MFMA 32x32x16, random A, dense random B"}
out = []
for rep in range(2):
    for mode in (0, 1, 3, 2, 4):
        s = bench.PowerSampler(0).start()
        tf = eng.mfma_sustained_data_tflops(1500, mode)
        pw = s.stop()
        row = {"mode": mode, "what": names[mode], "TFLOP/s": tf, "W": (pw.get("socket_power_w") or {}).get("mean"), "sclk_MHz": (pw.get("sclk_mhz") or {}).get("mean")}
        out.append(row)
        print(json.dumps(row))
