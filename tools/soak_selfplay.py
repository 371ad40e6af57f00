#!/usr/bin/env python3
"""Soak run on one MI355X: 3000 complete 9x9 self-play games through 1024 recycled slots (tower 2,
32 readouts); every game id finishes exactly once, no pool exhaustion, a sample of the records replays
legally on the oracle and ends where the record says.  ~15 s.  (tools/, not a pytest: it is a long run.)"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, alphago_jl_amd as ag, orc
N=9; G=1024; TOTAL=3000
eng = ag.Engine(board_size=N, tower_height=2, games=G, num_readouts=32, seed=21, record_capacity_games=TOTAL+64)
eng.init_synthetic(0); eng.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f32"); eng.start(TOTAL)
t=time.time(); steps=0
while eng.records_count() < TOTAL and steps < 20000:
    eng.step(50); steps += 50
st = eng.stats(); print("steps", steps, "sec", round(time.time()-t,1), {k: st[k] for k in ("games_finished","positions","pool_exhausted","resigned_games","evals")})
recs = eng.records()
assert len(recs) == TOTAL and sorted(r["game_id"] for r in recs) == list(range(TOTAL))
rng = np.random.RandomState(0)
for r in [recs[i] for i in rng.choice(TOTAL, 40, replace=False)]:
    pos = orc.make_pos(N)
    for a in r["moves"]:
        rc, pos = orc.play(pos, int(a)); assert rc == orc.OK
    if not r["was_resign"]:
        assert pos.done or pos.n >= 113
        assert r["result"] == orc.lib().or_result(__import__("ctypes").byref(pos))
print("soak OK; moves/game avg", np.mean([r["num_moves"] for r in recs]))
eng.close()
