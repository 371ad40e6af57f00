#!/usr/bin/env python3
"""BASELINE.md section 4 from a bench.py line (the driver's command): python tools/fill_baseline_table.py profiles/r06_bench_default.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
d = json.loads(open(src).read().strip().splitlines()[-1])
cb, c3, c4, g = d["cpu_baseline"], d["config3"], d["config4"], d["generation"]
steady = d.get("steady_state", {}).get("value", d["value"])
rows = f"""## 4. Results table (filled by the build, round 6; one MI355X, `{os.path.relpath(src, ROOT)}` = `python bench.py --steps {d['steps']} --warmup {d['warmup']}`, the driver's command)

| config | backend | GPUs / cores | positions/s | evals/position | roofline.achieved | notes |
|---|---|---|---|---|---|---|
| C1 5×5 t1 R16 G4 | CPU restatement (`oracle/`) and HIP | — / 1 core; 1 GPU | plumbing only | 16 | — | parity: `tests/test_gpu_selfplay.py` (4 games == the oracle's, moves, π, q bit for bit), `tests/test_gpu_train_loop.py` (`train.jl:56-74` loop body) |
| C2 9×9 t10 R400 G1024 | HIP exact f32 (Winograd F(3×3,3×3) on the f32 MFMA) | 1 | **{d['value']:.1f}** = SURVEY 8d's generation rate ({g['games_finished']} games to their natural end in {g['wall_s']:.0f} s after a warm-up generation, `capped: {str(g['capped']).lower()}`); {steady:.1f} in the {d['steps']}-step steady-state window | {g['evals_per_position']:.0f} | {d['roofline']['achieved']:.1f} TFLOP/s executed = {d['roofline']['frac']:.3f} of the 157.3 f32 MFMA peak ({d['roofline']['achieved_algorithmic']:.0f} algorithmic); layer {d['roofline']['avg_launch_ms']:.3f} ms; HBM {d['roofline']['traffic'] / 1e9:.2f} GB per layer | {d['power']['socket_power_w']['mean']:.0f} W at {d['power']['sclk_mhz']['mean']:.0f} MHz: power-limited (DESIGN §4) |
| C2' same | CPU restatement, OpenMP C network | — / {cb['cores']} threads ({cb['cpu']}) | {cb['value']:.2f} | 400 | — | `cpu_baseline` of the same line; the Julia/Flux reference cannot run here (no `julia`) |
| C3 9×9 t10 R400 G8192 | HIP exact f32 + RCCL replay all-gather | 8 (also 2, 4) | not measured: one GPU per lease | | | code path: `bench.py --gpus N`; 8 ranks × 1024 games rehearsed on ONE GPU with a consistent exchange (`profiles/r05_bench_8rank_fullsize_single_device.json`); games are independent: the expected value is N × C2 |
| C4 19×19 t20 R800 G2048 | HIP exact f32 (Winograd F(4×4,3×3)) | 8 → one GPU's shard (256 games) measured | {c3['value']:.1f} per GPU (steady-state window) | — | {c3['roofline']['achieved']:.1f} TFLOP/s executed = {c3['roofline']['frac']:.3f}; layer {c3['layer_ms']:.3f} ms | `config3` of the line; {c3['ms_per_step']:.1f} ms per step |
| C5 19×19 t20 R1600 G4096 | HIP fp16 MFMA tower | 8 → one GPU's shard (512 games) measured | {c4['value']:.1f} per GPU (steady-state window) | — | {c4['roofline']['achieved']:.0f} TFLOP/s = {c4['roofline']['frac']:.3f} of the 2.5 PFLOP/s fp16 peak; layer {c4['layer_ms']:.3f} ms | `config4` of the line; {c4['ms_per_step']:.1f} ms per step; what fp16 does to the selected moves: DESIGN §4, `tests/test_gpu_c5_acceptance.py` |
"""
p = os.path.join(ROOT, "BASELINE.md")
s = open(p).read()
s = s[: s.index("## 4. Results table")] + rows
open(p, "w").write(s)
print(rows)
