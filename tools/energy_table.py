#!/usr/bin/env python3
"""ms, shader clock and socket power of the dominant tower-conv launch, per timing variant ("a table of
variants with ms, GHz and watts").  Each variant runs its layer back to back for --seconds under bench.py's PowerSampler
(amdgpu hwmon: power1_input, freq1_input); a variant is a fresh process, because the kernels read their debug switch once.

  python tools/energy_table.py --precision f16 --board 19 --batch 4096 --env AGZ_C16_DEBUG --variants 0 32 1 3 5 9 15
  python tools/energy_table.py --precision f32 --board 19 --batch 2048 --env AGZ_WINO4_X --variants 0 1 2 3 4 5 8

needs gpurun_ab/libagz_T.so (tools/build_timing_lib.sh, ALSO=agz_conv16 for the fp16 kernel) in place of libagz.so --
--swap does that around the run.  Results of variants other than 0 are WRONG by construction; only their cost is read."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import importlib.util

    import alphago_jl_amd as ag
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    eng = ag.Engine(board_size=args.board, tower_height=4, games=1, num_readouts=1, max_nodes_per_game=8)
    eng.init_synthetic(0)
    eng.set_precision(args.precision)
    if args.winograd != 1:
        eng.set_winograd(args.winograd)                # (3: the five-pass 64 x 128 form of F(3x3,3x3), agz_wino5.hip)
    eng.time_conv(args.batch, 20)                      # warm-up (packs, allocations, clocks)
    sampler = bench.PowerSampler(0).start()
    t0, ms, n = time.perf_counter(), [], 0
    while time.perf_counter() - t0 < args.seconds:
        ms.append(eng.time_conv(args.batch, 200))
        n += 200
    pw = sampler.stop()
    idle = bench.PowerSampler(0)
    time.sleep(1.0)
    idle.start()
    time.sleep(0.5)
    pidle = idle.stop()
    print(json.dumps({"ms": sum(ms) / len(ms), "launches": n, "power": pw, "idle_after": pidle}))
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--board", type=int, default=19)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--env", default="AGZ_C16_DEBUG")
    ap.add_argument("--variants", nargs="+", default=["0"])
    ap.add_argument("--extra-env", nargs="*", default=[], help="NAME=VALUE pairs set for every variant")
    ap.add_argument("--swap", action="store_true", help="run with gpurun_ab/libagz_T.so in place of libagz.so")
    ap.add_argument("--winograd", type=int, default=1)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    lib, keep = os.path.join(ROOT, "alphago.jl_amd", "libagz.so"), "/tmp/libagz_energy_keep.so"
    if args.swap:
        shutil.copy(lib, keep)
        shutil.copy(os.path.join(ROOT, "gpurun_ab", "libagz_T.so"), lib)
    rows = []
    try:
        for v in args.variants:
            env = dict(os.environ, **{args.env: str(v)}, **dict(kv.split("=", 1) for kv in args.extra_env))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--precision", args.precision, "--board",
                                str(args.board), "--batch", str(args.batch), "--seconds", str(args.seconds), "--winograd", str(args.winograd)],
                               env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                rows.append({"variant": v, "error": r.stderr[-300:]})
                continue
            d = json.loads(line[-1])
            pw = d["power"]
            w = (pw.get("socket_power_w") or {}).get("mean")
            mhz = (pw.get("sclk_mhz") or {}).get("mean")
            rows.append({"variant": v, "ms": d["ms"], "sclk_mhz": mhz, "sclk_min": (pw.get("sclk_mhz") or {}).get("min"),
                         "sclk_max": (pw.get("sclk_mhz") or {}).get("max"), "watts_max": (pw.get("socket_power_w") or {}).get("max"), "watts": w,
                         "mJ_per_launch": w * d["ms"] if w else None, "Mcycles": mhz * d["ms"] * 1e-3 if mhz else None,
                         "idle_watts_after": (d["idle_after"].get("socket_power_w") or {}).get("mean")})
    finally:
        if args.swap:
            shutil.copy(keep, lib)
    print(json.dumps({"precision": args.precision, "board": args.board, "batch": args.batch, "switch": args.env, "winograd": args.winograd,
                      "extra_env": args.extra_env, "rows": rows}))
    for r in rows:
        if "error" in r:
            print(f"# {args.env}={r['variant']}: {r['error']}", file=sys.stderr)
        else:
            print(f"# {args.env}={r['variant']:>7}: {r['ms']:.3f} ms  {r['sclk_mhz'] or 0:.0f} MHz [{r['sclk_min'] or 0:.0f}..{r['sclk_max'] or 0:.0f}]  {r['watts'] or 0:.0f} W (max {r['watts_max'] or 0:.0f})  "
                  f"{r['mJ_per_launch'] or 0:.0f} mJ/launch  {r['Mcycles'] or 0:.2f} Mcycles (idle after: {r['idle_watts_after'] or 0:.0f} W)",
                  file=sys.stderr)


if __name__ == "__main__":
    main()
