#!/usr/bin/env python3
"""VERDICT r3 #7 (hide the search phase at 19x19 behind the other half's tower): priced at the process level before
building it into the engine.  Two engines of G/2 game slots each, stepped concurrently from two host threads on their own
streams (the C ABI releases the GIL), against one engine of G slots: the same games, the same kernels, and whatever the
hardware scheduler can overlap between the two streams it overlaps.  Prints ms per step-pair and positions/s of both forms.

  python tools/experiments/two_cohorts.py --board 19 --tower 20 --readouts 800 --games 256 [--precision f16]
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import alphago_jl_amd as ag  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--board", type=int, default=19)
ap.add_argument("--tower", type=int, default=20)
ap.add_argument("--readouts", type=int, default=800)
ap.add_argument("--games", type=int, default=256)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--precision", default="f32")
args = ap.parse_args()


def make(games, base, stride):
    e = ag.Engine(board_size=args.board, tower_height=args.tower, games=games, num_readouts=args.readouts, seed=1,
                  game_id_base=base, game_id_stride=stride, stagger_moves=60)
    e.init_synthetic(0)
    e.set_precision(args.precision)
    e.start(0)
    e.step((args.readouts + 7) // 8 + 5)
    e.sync()
    return e


def timed(engines):
    def run(e):
        e.step(args.steps)
        e.sync()
    for e in engines:
        e.step(3)
        e.sync()
    p0 = sum(e.stats()["positions"] for e in engines)
    ths = [threading.Thread(target=run, args=(e,)) for e in engines]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    p1 = sum(e.stats()["positions"] for e in engines)
    return 1e3 * dt / args.steps, (p1 - p0) / dt


one = make(args.games, 0, 1)
ms1, r1 = timed([one])
ms1b, r1b = timed([one])
one.close()
two = [make(args.games // 2, 0, 2), make(args.games // 2, 1, 2)]
ms2, r2 = timed(two)
ms2b, r2b = timed(two)
for e in two:
    e.close()
print(json.dumps({"config": vars(args), "one_engine": {"ms_per_step": [ms1, ms1b], "positions_per_s": [r1, r1b]},
                  "two_cohorts_two_streams": {"ms_per_step_pair": [ms2, ms2b], "positions_per_s": [r2, r2b]},
                  "note": "two engines of half the slots, stepped concurrently from two threads: every kernel of one cohort may overlap any kernel of the other"}))
