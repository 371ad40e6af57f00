#!/usr/bin/env python3
"""Soak run on one MI355X: 3000 complete 9x9 self-play games through 1024 recycled slots (tower 2,
32 readouts); every game id finishes exactly once, no pool exhaustion, a sample of the records replays
legally on the oracle and ends where the record says.  ~15 s.  (tools/, not a pytest: it is a long run.)"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, alphago_jl_amd as ag, orc
# usage: soak_selfplay.py [precision [board [slots [total games]]]]   e.g.  f32 19 128 300  (the F(4x4,3x3) tower, two layer chains)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 9
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
TOTAL = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
# (the default pool: 16 * readouts + 256 + 16 * max_game_length nodes per game; the peak a game reached is printed)
eng = ag.Engine(board_size=N, tower_height=2, games=G, num_readouts=32, seed=21, record_capacity_games=TOTAL+64,
                max_nodes_per_game=0)
eng.init_synthetic(0); eng.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f32"); eng.start(TOTAL)
t=time.time(); steps=0
while eng.records_count() < TOTAL and steps < (20000 if N == 9 else 200000):
    eng.step(50); steps += 50
    if steps % 1000 == 0 and eng.stats()["pool_exhausted"]:
        sys.exit("node pool exhausted: raise max_nodes_per_game")
st = eng.stats(); print("steps", steps, "sec", round(time.time()-t,1), {k: st[k] for k in ("games_finished","positions","pool_exhausted","pool_short_searches","peak_nodes_per_game","node_capacity","resigned_games","evals")})
recs = eng.records()
assert len(recs) == TOTAL and sorted(r["game_id"] for r in recs) == list(range(TOTAL))
rng = np.random.RandomState(0)
for r in [recs[i] for i in rng.choice(TOTAL, min(40, TOTAL), replace=False)]:
    pos = orc.make_pos(N)
    for a in r["moves"]:
        rc, pos = orc.play(pos, int(a)); assert rc == orc.OK
    if not r["was_resign"]:
        assert pos.done or pos.n >= (N * N * 7) // 5
        assert r["result"] == orc.lib().or_result(__import__("ctypes").byref(pos))
print("soak OK; moves/game avg", np.mean([r["num_moves"] for r in recs]))
eng.close()
