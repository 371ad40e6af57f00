"""Replay buffer of train() (src/train.jl:4-12,47-66) -- SURVEY.md section 8f row 1.

The reference keeps three parallel vectors (pos_buffer, pi_buffer, res_buffer), appends the
`extract_data` tuples of every finished game, trims to the newest `memory_size` entries
(`shrink`, train.jl:52) and draws training batches of `batch_size` distinct entries
(`sample(1:n, batch_size, replace=false)`, train.jl:5).

Here a game is kept the way the engine exports it -- its action list, its search distributions
and its result -- not as materialised Position objects: positions are rebuilt on demand, on the
device, by replaying the move list (`agz_replay_features`, the counterpart of `replay_position`,
board.jl:557-578).  Entry k of the buffer is (game g, ply j); trimming is per entry exactly like
the reference (the oldest game may be partially trimmed: its late plies stay sampleable, the move
list is kept whole because replay needs it).  Records gathered from other ranks
(distributed.allgather_records) go through the same `push_record`.
"""
import collections

import numpy as np


class ReplayBuffer:
    def __init__(self, env, memory_size=500000):
        self.env = env
        self.memory_size = int(memory_size)
        self._games = collections.deque()      # dicts: moves int16[n], pis f32[n][A], result, first (oldest live ply)
        self._len = 0

    def __len__(self):
        return self._len

    # -- push_data + shrink, train.jl:51-52,60-64
    def push_record(self, rec):
        """rec: mapping with moves (action indices), pis [n][A], result -- an element of
        Engine.records() / distributed.unpack_records(), or an api.GameRecord"""
        if hasattr(rec, "searches_pi"):        # api.GameRecord (moves are board coords / None there)
            from .api import to_flat
            moves = np.array([to_flat(c, self.env) for c in rec.moves], np.int16)
            pis = np.asarray(rec.searches_pi, np.float32).reshape(len(moves), -1)
            result = rec.result
        else:
            moves = np.asarray(rec["moves"], np.int16)
            n = len(moves)
            pis = np.asarray(rec["pis"], np.float32).reshape(n, -1) if n else np.zeros((0, self.env.action_space), np.float32)
            result = rec["result"]
        n = len(moves)
        if n == 0:
            return
        assert pis.shape == (n, self.env.action_space), "searches_pi does not match the move list"   # mcts_play.jl:127
        self._games.append(dict(moves=moves, pis=pis, result=int(result), first=0))
        self._len += n
        while self._len > self.memory_size:
            g = self._games[0]
            live = len(g["moves"]) - g["first"]
            drop = min(live, self._len - self.memory_size)
            g["first"] += drop
            self._len -= drop
            if g["first"] == len(g["moves"]):
                self._games.popleft()

    def extend(self, records):
        for r in records:
            self.push_record(r)

    # -- get_replay_batch, train.jl:4-12
    def sample_indices(self, batch_size, rng):
        """`batch_size` distinct (game slot, ply) pairs, uniform over the live entries"""
        assert batch_size <= self._len, "Cannot take a larger sample than the buffer without replacement"
        idx = np.sort(rng.choice(self._len, size=batch_size, replace=False))
        out, base, gi = [], 0, 0
        games = list(self._games)
        for k in idx:
            while k >= base + len(games[gi]["moves"]) - games[gi]["first"]:
                base += len(games[gi]["moves"]) - games[gi]["first"]
                gi += 1
            out.append((gi, games[gi]["first"] + int(k - base)))
        order = rng.permutation(batch_size)                       # sample() returns them in random order
        return [out[i] for i in order], games

    def sample(self, batch_size, rng, engine, out=None):
        """-> (features [B][17*N*N] (numpy, or the CUDA tensor `out`), pi [A][B], results [B]) like
        get_replay_batch's (pos_replay, pi_replay = hcat(...), res_replay), the positions already
        turned into the network's input planes by device replay"""
        pairs, games = self.sample_indices(batch_size, rng)
        used = sorted({g for g, _ in pairs})
        offs, chunks, o = {}, [], 0
        for g in used:
            offs[g] = o
            chunks.append(games[g]["moves"])
            o += len(games[g]["moves"])
        moves = np.concatenate(chunks) if chunks else np.zeros(0, np.int16)
        feats = engine.replay_features(moves, [offs[g] for g, _ in pairs], [j for _, j in pairs], out=out)
        pi = np.stack([games[g]["pis"][j] for g, j in pairs], axis=1)
        res = np.array([games[g]["result"] for g, _ in pairs], np.int64)
        return feats, pi, res

    def positions(self):
        """the live entries as (game slot, ply) in buffer order -- oldest first"""
        return [(gi, j) for gi, g in enumerate(self._games) for j in range(g["first"], len(g["moves"]))]
