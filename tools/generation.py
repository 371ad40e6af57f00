#!/usr/bin/env python3
"""One full self-play generation at BASELINE.json configs[1] size, games played to their natural end
("timing over >= 1 full generation ... positions = sum over finished games of position.n";
the loop it models: /root/reference/src/selfplay.jl:22-43).

bench.py times K steps of the steady state (slots recycled, every game mid-search at a random phase) and
counts the moves completed inside the window.  This tool checks that a real generation sustains that rate:
the slots recycle for ever, the clock starts after a warm-up (stagger prelude + W steps, as in bench.py) and
stops when G more games have FINISHED; it reports
  generation_rate   = sum of num_moves of the games that finished inside the window / wall time   (8d's definition)
  steady_state_rate = moves played inside the window / wall time                                 (bench.py's definition)
together with evaluations per position, duplicate and terminal counts, game-length and result statistics.
Writes one JSON object (stdout, and --out)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--tower", type=int, default=10)
    ap.add_argument("--readouts", type=int, default=400)
    ap.add_argument("--games", type=int, default=1024, help="concurrent game slots = games per generation")
    ap.add_argument("--stagger", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--chunk", type=int, default=50, help="steps between polls of the finished-game counter")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16"])
    ap.add_argument("--max-seconds", type=float, default=1500.0)
    ap.add_argument("--no-warmup-generation", dest="warmup_generation", action="store_false")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import numpy as np

    import alphago_jl_amd as ag

    N, R, G = args.board, args.readouts, args.games
    eng = ag.Engine(board_size=N, tower_height=args.tower, games=G, num_readouts=R, parallel_readouts=8, seed=1,
                    stagger_moves=args.stagger, record_capacity_games=2 * G + 64)
    eng.init_synthetic(0)
    eng.set_precision(args.precision)
    eng.start(0)
    prelude = (R + 7) // 8 + 5 if args.stagger > 0 else 0
    eng.step(prelude + args.warmup)
    eng.sync()
    # warm-up generation ("timing over >= 1 full generation after a warm-up generation"): the first G
    # games to finish are the staggered start-up population (random opening prefixes that were never searched, all
    # born at the same moment); the measured window opens when the G-th of them has finished
    tw = time.perf_counter()
    if args.warmup_generation:
        while eng.stats()["games_finished"] < G and time.perf_counter() - tw < args.max_seconds:
            eng.step(args.chunk)
    warm_s = time.perf_counter() - tw
    eng.sync()
    eng.records_clear()
    s0 = eng.stats()
    t0 = time.perf_counter()
    steps = 0
    while True:
        eng.step(args.chunk)
        steps += args.chunk
        st = eng.stats()                                   # synchronises
        if st["games_finished"] - s0["games_finished"] >= G or time.perf_counter() - t0 > args.max_seconds:
            break
    t1 = time.perf_counter()
    s1 = st
    wall = t1 - t0
    recs = eng.records()
    nm = np.array([r["num_moves"] for r in recs], np.int64)
    res = np.array([r["result"] for r in recs], np.int64)
    d = {k: s1[k] - s0[k] for k in ("positions", "evals", "duplicate_evals", "terminal_visits", "root_visits",
                                    "games_finished", "games_started", "resigned_games", "steps")}
    out = {
        "what": "one generation played to natural end, slots recycled (tools/generation.py)",
        "workload": f"GoEnv({N}), tower_height={args.tower}, {R} readouts, {G} concurrent games, precision {args.precision}",
        "wall_s": wall, "steps": d["steps"], "ms_per_step": 1e3 * wall / max(d["steps"], 1),
        "games_finished": d["games_finished"], "games_started": d["games_started"], "resigned_games": d["resigned_games"],
        "records_read": len(recs), "records_dropped": s1["records_dropped"],
        "generation_positions": int(nm.sum()), "generation_rate": float(nm.sum()) / wall,
        "steady_state_positions": d["positions"], "steady_state_rate": d["positions"] / wall,
        "evals": d["evals"], "evals_per_position": d["evals"] / max(d["positions"], 1),
        "duplicate_evals": d["duplicate_evals"], "terminal_visits": d["terminal_visits"],
        "batch_fill": d["evals"] / max(d["steps"] * 8 * G, 1),
        "game_length": {"mean": float(nm.mean()) if len(nm) else None, "min": int(nm.min()) if len(nm) else None,
                        "max": int(nm.max()) if len(nm) else None},
        "results": {"black": int((res > 0).sum()), "white": int((res < 0).sum()), "draw": int((res == 0).sum())},
        "pool_exhausted": s1["pool_exhausted"], "timed_out": wall > args.max_seconds,
        "setup": {"prelude_steps": prelude, "warmup_steps": args.warmup, "stagger_moves": args.stagger,
                  "warmup_generation": bool(args.warmup_generation), "warmup_generation_s": warm_s},
    }
    txt = json.dumps(out)
    print(txt, flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")
    eng.close()


if __name__ == "__main__":
    main()
