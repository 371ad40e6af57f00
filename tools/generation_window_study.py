#!/usr/bin/env python3
"""How long must the measured window of SURVEY.md 8d's generation rate be?

generation_rate = sum of the lengths of the games that ENDED in the window / the window's wall time.  bench.py's window
closes when as many games have ended as there are slots (one generation): the games that manage to end inside a window
as long as one game's life are the short ones (the inspection paradox), so the rate reads below the rate at which
positions are actually produced.  This tool plays a warm-up generation and then K more on the headline workload and
reports the count-defined rate for windows of 1 .. K generations, the time-defined rate for the same durations, and the
steady-state rate (moves PLAYED / time) -- all from one run, one box.
  python tools/generation_window_study.py --generations 3 > profiles/r06_generation_window_study.json"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--generations", type=int, default=3)
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=25)
    args = ap.parse_args()
    import numpy as np

    import alphago_jl_amd as ag

    N, tower, R, G, K = 9, 10, 400, args.games, args.generations
    eng = ag.Engine(board_size=N, tower_height=tower, games=G, num_readouts=R, parallel_readouts=8, seed=1,
                    stagger_moves=60, record_capacity_games=(K + 1) * G + 256)
    eng.init_synthetic(0)
    eng.start(0)
    eng.step((R + 7) // 8 + 5 + 10)
    eng.sync()
    while eng.stats()["games_finished"] < G:          # warm-up generation
        eng.step(args.chunk)
    eng.records_clear()
    s0 = eng.stats()
    t0 = time.perf_counter()
    polls = []                                        # (seconds, games finished, positions played) since the window opened
    while True:
        eng.step(args.chunk)
        s = eng.stats()
        polls.append((time.perf_counter() - t0, s["games_finished"] - s0["games_finished"], s["positions"] - s0["positions"]))
        if polls[-1][1] >= K * G:
            break
    recs = eng.records()                               # finish order
    lens = np.array([r["num_moves"] for r in recs], np.int64)
    cum = np.concatenate([[0], np.cumsum(lens)])       # moves of the first k games to end
    t = np.array([p[0] for p in polls])
    fin = np.array([p[1] for p in polls])
    pos = np.array([p[2] for p in polls])
    out = {"workload": f"GoEnv({N}), tower {tower}, {R} readouts, {G} slots; warm-up generation, then {K} generations in one run",
           "games_recorded": int(len(recs)), "mean_game_length": float(lens.mean()), "windows": []}
    for k in range(1, K + 1):
        i = int(np.argmax(fin >= k * G))               # first poll at which k generations' worth of games have ended
        w = {"generations": k, "wall_s": float(t[i]), "games_ended": int(fin[i]),
             "count_defined_generation_rate": float(cum[min(fin[i], len(lens))] / t[i]),
             "steady_state_rate": float(pos[i] / t[i]),
             "mean_length_of_games_ended": float(cum[min(fin[i], len(lens))] / max(fin[i], 1))}
        out["windows"].append(w)
    out["note"] = ("count_defined_generation_rate = sum of lengths of the games that ended in the window / wall time (bench.py's `value`, "
                   "window = 1 generation); steady_state_rate = moves played in the same window / wall time.  The two converge as the "
                   "window grows: the shortfall of the one-generation window is the short games it selects")
    print(json.dumps(out, indent=1))
    eng.close()


if __name__ == "__main__":
    main()
