"""Replay buffer contract of train() (src/train.jl:4-12,47-66): FIFO window per
entry, sampling without replacement, (features, pi A x B, results) batches.  CPU tests use an
oracle-backed stand-in for the device replay; the GPU test runs agz_replay_features itself."""
import numpy as np
import pytest

import alphago_jl_amd as ag
import orc


class OracleReplay:
    """what Engine.replay_features computes, restated with the oracle (replay_position, board.jl:557-578)"""

    def __init__(self, N):
        self.N = N

    def replay_features(self, moves, game_offset, ply, out=None):
        res = []
        for off, j in zip(game_offset, ply):
            pos = orc.make_pos(self.N)
            for k in range(j):
                rc, pos = orc.play(pos, int(moves[off + k]))
                assert rc == orc.OK
            res.append(orc.feats(pos).astype(np.float32).reshape(-1))
        return np.stack(res)


def random_game(N, n, seed):
    rng = np.random.RandomState(seed)
    pos, moves = orc.make_pos(N), []
    while len(moves) < n:
        legal = np.flatnonzero(orc.legal_moves(pos)[:-1])
        a = int(rng.choice(legal)) if len(legal) else N * N
        rc, pos = orc.play(pos, a)
        assert rc == orc.OK
        moves.append(a)
    A = N * N + 1
    pis = rng.rand(n, A).astype(np.float32)
    pis /= pis.sum(1, keepdims=True)
    return dict(moves=np.array(moves, np.int16), pis=pis, result=int(rng.choice([-1, 1])), game_id=seed)


def test_fifo_window_trims_per_entry():
    env = ag.GoEnv(5)
    buf = ag.ReplayBuffer(env, memory_size=25)
    games = [random_game(5, n, s) for s, n in enumerate([10, 12, 9])]
    buf.extend(games[:2])
    assert len(buf) == 22 and buf.positions()[0] == (0, 0)
    buf.push_record(games[2])                    # 31 entries -> the 6 oldest go (train.jl:52: vec[end-memory_size+1:end])
    assert len(buf) == 25
    pos = buf.positions()
    assert pos[0] == (0, 6) and pos[-1] == (2, 8) and len(pos) == 25
    buf.push_record(random_game(5, 20, 7))       # drops the rest of game 0, all of game 1, the first 4 plies of game 2
    assert len(buf) == 25 and buf.positions()[0] == (0, 4) and len(buf._games) == 2
    buf.push_record(dict(moves=[], pis=[], result=1))        # a game resigned before its first move adds nothing
    assert len(buf) == 25


def test_sample_without_replacement_and_batch_shapes():
    N, A = 5, 26
    env = ag.GoEnv(N)
    buf = ag.ReplayBuffer(env, memory_size=1000)
    games = [random_game(N, 8 + s, 10 + s) for s in range(5)]
    buf.extend(games)
    rng = np.random.default_rng(0)
    pairs, _ = buf.sample_indices(len(buf), rng)
    assert sorted(pairs) == buf.positions()                   # the whole buffer exactly once
    with pytest.raises(AssertionError):
        buf.sample_indices(len(buf) + 1, rng)
    feats, pi, res = buf.sample(16, np.random.default_rng(3), OracleReplay(N))
    assert feats.shape == (16, 17 * N * N) and pi.shape == (A, 16) and res.shape == (16,)
    pairs, gl = buf.sample_indices(16, np.random.default_rng(3))           # same draws -> same batch
    for b, (g, j) in enumerate(pairs):
        assert (pi[:, b] == gl[g]["pis"][j]).all() and res[b] == gl[g]["result"]
        want = OracleReplay(N).replay_features(gl[g]["moves"], [0], [j])[0]
        assert (feats[b] == want).all()
    assert len(set(pairs)) == 16


def test_trimmed_game_still_replays_from_move_zero():
    N = 5
    buf = ag.ReplayBuffer(ag.GoEnv(N), memory_size=6)
    g = random_game(N, 14, 3)
    buf.push_record(g)
    assert buf.positions() == [(0, j) for j in range(8, 14)]
    feats, pi, res = buf.sample(6, np.random.default_rng(1), OracleReplay(N))
    pairs, _ = buf.sample_indices(6, np.random.default_rng(1))
    for b, (_, j) in enumerate(pairs):
        assert (feats[b] == OracleReplay(N).replay_features(g["moves"], [0], [j])[0]).all()


def test_game_record_objects_are_accepted():
    env = ag.GoEnv(5)
    g = random_game(5, 6, 4)
    rec = ag.GameRecord(0, [ag.from_flat(int(a), env) for a in g["moves"]], list(g["pis"]), np.zeros(6), g["result"], "B+1.5", False)
    buf = ag.ReplayBuffer(env)
    buf.push_record(rec)
    assert (buf._games[0]["moves"] == g["moves"]).all() and len(buf) == 6


@pytest.mark.gpu
@pytest.mark.parametrize("N", [5, 9, 19])
def test_gpu_replay_features_match_oracle(N):
    import torch
    eng = ag.Engine(board_size=N, games=1, tower_height=0, num_readouts=8, max_nodes_per_game=16)
    games = [random_game(N, n, 20 + s) for s, n in enumerate([3 * N, 2 * N + 1, 1, (N * N * 7) // 5])]
    buf = ag.ReplayBuffer(ag.GoEnv(N))
    buf.extend(games)
    B = min(24, len(buf))
    feats, pi, res = buf.sample(B, np.random.default_rng(5), eng)
    want, wpi, wres = buf.sample(B, np.random.default_rng(5), OracleReplay(N))
    assert (feats == want).all() and (pi == wpi).all() and (res == wres).all()
    out = torch.empty((B, 17 * N * N), dtype=torch.float32, device="cuda")          # device-resident batch
    buf.sample(B, np.random.default_rng(5), eng, out=out)
    assert (out.cpu().numpy() == want).all()
    # ply 0 is the empty board, Black to play; bad samples are rejected with an error, not garbage
    f0 = eng.replay_features(games[0]["moves"], [0], [0])[0].reshape(17, -1)
    assert (f0[:16] == 0).all() and (f0[16] == 1).all()
    with pytest.raises(ag.AgzError):
        eng.replay_features(games[0]["moves"], [0], [len(games[0]["moves"]) + 1])
    eng.close()
