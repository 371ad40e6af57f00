"""The N>1 path on CPU: two gloo ranks shard self-play games by id (rank r plays r, r+2, ...),
with no data-path collective, then all-gather their finished-game records.  The games come from
the host wave simulator over the engine's search templates (no GPU here); the union must equal a
single-rank run of the same ids.  CPU only."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import alphago_jl_amd as ag
from alphago_jl_amd.distributed import allgather_records, pack_records, unpack_records

HERE = os.path.dirname(os.path.abspath(__file__))


def play(rank, world, games_total, slots, seed=5):
    sys.path.insert(0, HERE)
    import hs
    from test_hostsim_selfplay import OracleNet

    net = OracleNet(5, 1, seed=0)
    mine = len(range(rank, games_total, world))
    sim = hs.Sim(board_size=5, games=slots, num_readouts=16, seed=seed, game_id_base=rank, game_id_stride=world,
                 record_capacity_games=mine + 4)
    sim.start(mine)
    while sim.counters()["finished"] < mine:
        sim.step(net.on_feats)
    recs = sim.records()
    sim.close()
    net.close()
    return recs


def worker(rank, world, port, games_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs = play(rank, world, games_total, slots=2)
    assert all(r["game_id"] % world == rank for r in recs)
    allr = allgather_records(recs, 26)
    dist.barrier()
    q.put((rank, [(r["game_id"], r["num_moves"], r["result"], r["moves"].tolist(), float(np.nansum(r["pis"]))) for r in allr]))
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_pack_unpack_roundtrip():
    rng = np.random.RandomState(0)
    recs = []
    for g, n in enumerate((0, 1, 7, 30)):
        recs.append(dict(game_id=g * 3 + 1, result=int(rng.choice([-1, 0, 1])), was_resign=int(g % 2), resign_disabled=0,
                         final_score=float(g) - 0.5, moves=rng.randint(0, 26, n).astype(np.int16),
                         pis=rng.rand(n, 26).astype(np.float32), qs=rng.rand(n).astype(np.float32)))
    buf = pack_records(recs, 26)
    assert buf.size % 8 == 0
    back = unpack_records(buf, 26)
    assert len(back) == len(recs)
    for a, b in zip(recs, back):
        assert a["game_id"] == b["game_id"] and a["result"] == b["result"] and a["was_resign"] == b["was_resign"]
        assert (a["moves"] == b["moves"]).all() and (a["pis"] == b["pis"]).all() and (a["qs"] == b["qs"]).all()
        assert b["final_score"] == np.float32(a["final_score"])


def test_two_rank_gloo_shard_and_allgather():
    world, games_total = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, games_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0] == got[1]                      # every rank ends with the same replay set
    assert [g[0] for g in got[0]] == list(range(games_total))
    single = play(0, 1, games_total, slots=3)    # one rank playing all ids gives the same games
    assert [(r["game_id"], r["num_moves"], r["result"], r["moves"].tolist(), float(np.nansum(r["pis"]))) for r in single] == got[0]


class StubWeights:
    """the three methods broadcast_weights needs, over plain numpy (no GPU in this container)"""

    def __init__(self, seed):
        rng = np.random.RandomState(seed)
        self.w = {(l, k): rng.randn(n).astype(np.float32) for (l, k), n in {(0, 0): 3 * 3 * 17 * 8, (0, 1): 8, (-3, 0): 50, (-5, 1): 26}.items()}

    def layers(self):
        return sorted(self.w)

    def get_weights(self, l, k):
        return self.w[(l, k)]

    def set_weights(self, l, k, data):
        assert len(data) == len(self.w[(l, k)])
        self.w[(l, k)] = np.array(data, np.float32)


def bcast_worker(rank, world, port, q):
    from alphago_jl_amd.distributed import broadcast_weights
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = StubWeights(seed=10 + rank)              # every rank starts with different parameters
    n = broadcast_weights(net, src=1)
    q.put((rank, n, {str(k): v.tolist() for k, v in net.w.items()}))
    dist.destroy_process_group()


def test_two_rank_gloo_weight_broadcast():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: (n, w) for r, n, w in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = {str(k): v.tolist() for k, v in StubWeights(seed=11).w.items()}      # rank 1's parameters
    assert got[0][1] == want and got[1][1] == want
    assert got[0][0] == got[1][0] == sum(len(v) for v in want.values())
