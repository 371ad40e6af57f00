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


# ---- Engine.allgather_records_hosted (the host-carried exchange) over a stub of the C ABI --------------------------------
class StubAbi:
    """The entry points Engine.allgather_records_hosted calls, over numpy: `records` is this rank's packed export.
    agz_gather_plan is the REAL library's (pure host code); everything else is a stand-in for an engine on a GPU."""

    def __init__(self, packed, nrec, fail_status=0):
        import alphago_jl_amd as ag_
        self.real = ag_.load()
        self.packed, self.nrec, self.fail = np.ascontiguousarray(packed, np.uint8), nrec, fail_status
        self.ingested = None

    def agz_records_packed_size(self, h, out):
        if self.fail:
            return self.fail
        out._obj.value = self.packed.size
        return 0

    def agz_records_count(self, h):
        return self.nrec

    def agz_engine_sync(self, h):          # (the real method orders the engine's stream against torch's around the pack)
        return 0

    def agz_gather_plan(self, *a):
        return self.real.agz_gather_plan(*a)

    def agz_last_error(self, h):
        return self.real.agz_last_error(None) if h is None else b"stub engine error"

    def agz_records_export_packed(self, h, dst, nbytes, is_device):
        raise AssertionError("replaced per test")

    def agz_replay_ingest_gathered(self, h, buf, is_device, world, stride, counts, added):
        import ctypes as C
        n = world * stride
        raw = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(n,)).copy()
        cnt = np.ctypeslib.as_array(counts, shape=(2 * world,)).copy()
        self.ingested = (raw, int(stride), cnt)
        added._obj.value = int(cnt[0::2].sum())
        return 0


def hosted_worker(rank, world, port, sizes, fail_rank, q):
    import ctypes as C

    import torch

    import alphago_jl_amd as ag_
    from alphago_jl_amd.engine import Engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nrec, nbytes = sizes[rank]
    packed = (np.arange(nbytes, dtype=np.int64) * 7 + rank * 31).astype(np.uint8)
    eng = Engine.__new__(Engine)                       # no engine behind it: the protocol logic only
    eng.L = StubAbi(packed, nrec, ag_._lib.HIP_ERROR if rank == fail_rank else 0)
    eng.h, eng.cfg = None, type("Cfg", (), {"device": 0})()

    # the export writes into a CUDA tensor in the real method; on this GPU-less host route it through a CPU tensor
    real_zeros = torch.zeros

    def zeros_cpu(*a, **kw):
        kw["device"] = "cpu"
        return real_zeros(*a, **kw)

    torch.zeros = zeros_cpu

    def export(h, dst, nb, is_device):
        C.memmove(dst, packed.ctypes.data, int(nb))
        return 0

    eng.L.agz_records_export_packed = export
    out = {"rank": rank}
    try:
        out["added"] = eng.allgather_records_hosted()
        raw, stride, cnt = eng.L.ingested if eng.L.ingested else (None, 0, None)
        out["stride"], out["counts"] = stride, None if cnt is None else cnt.tolist()
        if raw is not None:
            out["chunks"] = [raw[r * stride: r * stride + sizes[r][1]].tolist() for r in range(world)]
    except ag_.AgzError as ex:
        out["error"] = (int(ex.status), str(ex))
    torch.zeros = real_zeros
    dist.barrier()
    q.put(out)
    dist.destroy_process_group()


def run_hosted(sizes, fail_rank=-1):
    world = len(sizes)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=hosted_worker, args=(r, world, port, sizes, fail_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        o = q.get(timeout=300)
        got[o["rank"]] = o
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_hosted_exchange_protocol_two_gloo_ranks():
    """counts -> agz_gather_plan -> padded payload -> ingest, as Engine.allgather_records_hosted runs it (the GPU version of
    this test, with real engines, is tests/test_gpu_multirank.py): unequal ranks, stride = the largest padded to 256"""
    sizes = [(3, 3 * 32 + 1000), (1, 40)]
    got = run_hosted(sizes)
    for r in (0, 1):
        assert got[r]["added"] == 4 and got[r]["stride"] == 1280 and got[r]["counts"] == [3, 1096, 1, 40]
    want = [((np.arange(nb, dtype=np.int64) * 7 + r * 31).astype(np.uint8)).tolist() for r, (_, nb) in enumerate(sizes)]
    assert got[0]["chunks"] == want and got[1]["chunks"] == want


def test_hosted_exchange_of_nothing_and_failure_sentinel():
    got = run_hosted([(0, 0), (0, 0)])
    assert got[0]["added"] == got[1]["added"] == 0
    got = run_hosted([(2, 128), (1, 64)], fail_rank=0)
    assert got[0]["error"][0] == ag._lib.HIP_ERROR                       # the failing rank reports its own failure
    assert got[1]["error"][0] == ag._lib.RCCL_ERROR and "rank 0 failed before the exchange" in got[1]["error"][1]


def test_bench_self_launcher_starts_n_ranks_and_propagates_their_status():
    """`python bench.py --gpus 2` with WORLD_SIZE unset is its own launcher.  Without a GPU every rank must
    stop at the same place with the same message -- 'no CPU fallback' -- and the launcher must return their status; with
    --gpus above the visible device count (and no --single-device-test) it refuses before starting anything."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the GPU suite runs the launcher for real (tests/test_gpu_multirank.py)")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-device-test", "--steps", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 1 and r.stdout.strip() == ""
    assert r.stderr.count("bench.py needs an MI355X: the engine has no CPU fallback") == 2, r.stderr[-500:]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 2 and "--gpus 2 but 0 GPU(s) visible" in r.stderr
