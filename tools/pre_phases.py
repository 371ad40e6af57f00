#!/usr/bin/env python3
"""Where k_pre's time goes.  Needs libagz.so built with `make EXTRA=-DAGZ_TIMING_EXPERIMENTS`: the search
code then stamps the 100 MHz wall clock at its phase boundaries (agz_search.h: AGZ_STAMP) and sums the differences
into device counters.  Runs the bench workload (BASELINE configs[1]) for K steps after the stagger prelude and prints,
per game that went through the phase, the average time of
  free      deferred release of the previous move's subtrees (every game, every step, budgeted)
  pick      pick_move + children_as_pi (record pi, Q)
  child     node_create_child (only when the chosen child was never expanded)
  reroot    reroot: history ring, garbage stack, root statistics
  noise     inject_noise: Dirichlet draw over the new root's children
  select    the select phase of games that did NOT move this step / of those that did
and the worst single (move phase + select) seen, which is what the kernel's duration follows."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alphago_jl_amd as ag  # noqa: E402

NAMES = ["STEPS", "POSITIONS", "STARTED", "FINISHED", "EVALS", "DUP", "TERMINAL", "ROOTVISITS", "POOL_EXHAUSTED", "RESIGNED",
         "CLAIMED", "RECORDED", "T_FREE", "T_PICK", "T_CHILD", "T_REROOT", "T_NOISE", "T_MOVE_SELECT", "N_MOVE", "T_SELECT",
         "N_SELECT", "T_MOVE_MAX", "T_CREATE", "N_CREATE"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--tower", type=int, default=10)
    ap.add_argument("--readouts", type=int, default=400)
    args = ap.parse_args()
    eng = ag.Engine(board_size=args.board, tower_height=args.tower, games=args.games, num_readouts=args.readouts, seed=1,
                    stagger_moves=60)
    eng.init_synthetic(0)
    eng.start(0)
    eng.step(args.readouts // 8 + 15)
    c0 = dict(zip(NAMES, eng.debug_counters().astype(float)))
    eng.step(args.steps)
    c1 = dict(zip(NAMES, eng.debug_counters().astype(float)))
    d = {k: c1[k] - c0[k] for k in NAMES}
    if d["N_SELECT"] + d["N_MOVE"] == 0:
        sys.exit("no phase clocks: libagz.so was not built with -DAGZ_TIMING_EXPERIMENTS")
    us = 0.01   # 100 MHz ticks -> microseconds
    nm, ns = max(d["N_MOVE"], 1), max(d["N_SELECT"], 1)
    print(f"steps {int(d['STEPS'])}, games in a move phase per step {d['N_MOVE'] / d['STEPS']:.1f} of {args.games}")
    print(f"free (all games)      {d['T_FREE'] * us / (nm + ns):8.1f} us")
    for k in ("PICK", "CHILD", "REROOT", "NOISE"):
        print(f"{k.lower():<21} {d['T_' + k] * us / nm:8.1f} us per moving game")
    print(f"select after a move   {d['T_MOVE_SELECT'] * us / nm:8.1f} us")
    print(f"select (no move)      {d['T_SELECT'] * us / ns:8.1f} us")
    print(f"node_create_child     {d['T_CREATE'] * us / max(d['N_CREATE'], 1):8.1f} us x {d['N_CREATE'] / (nm + ns):.2f} per game and step")
    print(f"worst move + select   {c1['T_MOVE_MAX'] * us:8.1f} us")
    eng.close()


if __name__ == "__main__":
    main()
