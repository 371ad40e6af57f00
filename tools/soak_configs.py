#!/usr/bin/env python3
"""BASELINE configs[3] and configs[4] played to the very end at their own readout budget, resignation DISABLED: 19x19,
tower 20, 800 readouts (exact f32, F(4x4,3x3) tower) and 1600 readouts (fp16 tower), 16 slots each on the DEFAULT node
pool, until WANT games of each have ended by two passes or at max_game_length = 505 (/root/reference/src/selfplay.jl:22-43,
src/mcts.jl:15-25).  The two engines step side by side (own streams).  Checked per finished game: no refused allocation,
no shortened search, the record replays legally on the oracle's rules, ends by two passes or at move 505, result and score
equal the oracle's Tromp-Taylor count.  Prints the peak tree size against the pool.  ~10 min on one MI355X; not a pytest
(tests/test_gpu_configs.py plays the same configs until the first games end, by resignation)."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np

import alphago_jl_amd as ag
import orc

L = orc.lib()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
WANT = int(sys.argv[2]) if len(sys.argv) > 2 else 8
runs = []
for precision, R in (("f32", 800), ("f16", 1600)):
    eng = ag.Engine(board_size=19, tower_height=20, games=G, num_readouts=R, seed=17, record_capacity_games=2 * G + 8,
                    resign_disable_fraction=1.0)
    eng.init_synthetic(0)
    eng.set_precision(precision)
    eng.start(0)
    runs.append(dict(precision=precision, R=R, eng=eng, steps=0, done=False, t_done=None))
t0 = time.time()
while not all(r["done"] for r in runs) and time.time() - t0 < 1500:
    for r in runs:
        if not r["done"]:
            r["eng"].step((r["R"] + 7) // 8)           # asynchronous: the other engine's step overlaps
            r["steps"] += (r["R"] + 7) // 8
    for r in runs:
        if not r["done"] and r["eng"].records_count() >= WANT:
            r["done"], r["t_done"] = True, time.time() - t0
for r in runs:
    eng, st, recs = r["eng"], r["eng"].stats(), r["eng"].records()
    assert len(recs) >= WANT, f"{r['precision']}: {len(recs)} games over after {r['steps']} steps"
    assert st["pool_exhausted"] == 0 and st["pool_short_searches"] == 0 and st["stalled_games"] == 0
    ended = {"passes": 0, "length": 0}
    for rec in recs:
        assert not rec["was_resign"] and rec["resign_disabled"] and rec["short_searches"] == 0
        pos = orc.make_pos(19)
        for k, a in enumerate(rec["moves"]):
            legal = orc.legal_moves(pos)
            assert legal[int(a)] == 1 and not (rec["pis"][k][legal == 0] > 0).any()
            rc, pos = orc.play(pos, int(a))
            assert rc == orc.OK
        assert pos.n == rec["num_moves"] <= 505 and (pos.done or pos.n >= 505)
        ended["passes" if pos.done else "length"] += 1
        assert rec["result"] == L.or_result(C.byref(pos)) and abs(rec["final_score"] - L.or_score(C.byref(pos))) < 1e-6
    print(f"configs soak {r['precision']} R={r['R']}: {len(recs)} games over in {r['steps']} steps / {r['t_done']:.0f} s, "
          f"moves {[int(x['num_moves']) for x in recs]}, ended {ended}, peak nodes per game {st['peak_nodes_per_game']} of "
          f"{st['node_capacity']} ({st['peak_nodes_per_game'] / st['node_capacity']:.2f}), evals {st['evals']}, "
          f"positions {st['positions']}")
    eng.close()
print("soak_configs OK")
