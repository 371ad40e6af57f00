#!/usr/bin/env python3
"""One generation of the reference's train() loop (src/train.jl:38-92) on the MI355X:
self-play -> replay arena -> training batches -> optimiser step -> arena match -> checkpoint.

    python examples/generation_loop.py [--board 9] [--tower 2] [--games 32] [--readouts 64]

What runs where:
  selfplay          G concurrent games on the device (one wave per tree, one network batch per step)
  extract_data      finished games come back as (moves, pi, result); multi-GPU: all-gathered over RCCL
  get_replay_batch  (game, ply) pairs are sampled on the host; the device replays the move lists of the replay
                    arena and emits the N x N x 17 x B feature tensor, pi and z (agz_replay_batch)
  _train            agz_train_step: training-mode forward, 0.01 CE + 0.01 MSE + 1e-4 L2, backward, Momentum(0.02)
  evaluate          candidate vs incumbent, both networks resident in one arena engine
  save_model        BSON parameter lists readable by Flux.loadparams!
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (before the engine: one HIP runtime per process)

import alphago_jl_amd as ag  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--tower", type=int, default=2)
    ap.add_argument("--games", type=int, default=32)
    ap.add_argument("--readouts", type=int, default=64)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--eval-games", type=int, default=8)
    args = ap.parse_args(argv)

    env = ag.GoEnv(args.board)
    cur = ag.NeuralNet(env, tower_height=args.tower, seed=0)        # Flux-default-equivalent init
    prev = ag.NeuralNet(env, tower_height=args.tower, seed=1)

    records = ag.selfplay(env, cur, args.readouts, games=args.games, seed=1)
    buf = ag.ReplayBuffer(env, memory_size=500000)
    buf.extend(records)
    print(f"self-play: {len(records)} games, {len(buf)} positions, "
          f"results B/W/draw = {sum(r.result == 1 for r in records)}/{sum(r.result == -1 for r in records)}/"
          f"{sum(r.result == 0 for r in records)}")

    B = min(args.batch_size, len(buf))
    feats = torch.empty((B, 17 * env.N * env.N), dtype=torch.float32, device="cuda")
    _, pi, z = buf.sample(B, np.random.default_rng(0), cur.engine, out=feats)       # host-side buffer, device replay
    # the same through the device replay arena + the training step (train.jl:56-70)
    eng = cur.engine
    eng.replay_ingest(ag.distributed.pack_records(
        [dict(game_id=r.game_id, result=r.result, was_resign=r.was_resign, moves=[ag.to_flat(c, env) for c in r.moves],
              pis=r.searches_pi, qs=r.qs) for r in records], env.action_space))
    rng = np.random.default_rng(1)
    lens = np.array([eng.replay_record(k)["num_moves"] for k in range(eng.replay_count())])
    losses = []
    for it in range(3):
        flat = rng.choice(int(lens.sum()), size=B, replace=False)                 # sample(1:n, B, replace=false)
        game = np.searchsorted(np.cumsum(lens), flat, side="right")
        ply = flat - (np.cumsum(lens) - lens)[game]
        f, p, zz = eng.replay_batch(game, ply)
        losses.append(eng.train_step(f, p, zz)[0])
    print("training: loss", " -> ".join(f"{x:.5f}" for x in losses))

    ok, st = ag.evaluate(env, cur, prev, num_games=args.eval_games, ro=args.readouts, seed=2, return_stats=True)
    print(f"evaluate: Black (candidate) won {st.games_won}/{st.num_games} -> {'keep' if ok else 'revert'}")

    with tempfile.TemporaryDirectory() as d:
        ag.save_model(cur, d)
        back = ag.load_model(d, env)
        x = (np.random.RandomState(0).rand(1, 17 * env.N * env.N) < 0.3).astype(np.float32)
        same = (back.engine.forward_features(x)[0] == cur.engine.forward_features(x)[0]).all()
        print("checkpoint round trip:", sorted(os.listdir(os.path.join(d, "weights"))), "identical outputs:", bool(same))
        back.engine.close()
    return len(buf), st.num_games, bool(same)


if __name__ == "__main__":
    main()
