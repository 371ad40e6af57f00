"""N > 1 before the driver's first SCALE run: several ranks, one process each, ALL on cuda:0 of this 1-GPU box (RCCL
refuses two ranks on one device, so gloo carries the two collectives), every other step of the replay exchange
through the C ABI -- agz_records_count / agz_records_packed_size -> agz_gather_plan -> agz_records_export_packed ->
agz_replay_ingest_gathered (Engine.allgather_records_hosted), the route include/agz.h gives a host that brings its
own communication library.  Checked: every rank's device arena holds the same games in rank order, bit-identical to
the records of a single-rank engine playing the same ids; unequal and empty ranks; a rank that fails before the
exchange takes every rank down together; and bench.py's own N > 1 leg (fd swap, thread, watchdog, exchange object)
under torch.distributed.run.  Caller served: /root/reference/src/train.jl:56-66 across ranks (BASELINE configs[2])."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def digest(r):
    h = hashlib.sha256()
    for a in (np.asarray(r["moves"], np.int16), np.asarray(r["pis"], np.float32), np.asarray(r["qs"], np.float32),
              np.asarray([r["final_score"]], np.float32)):
        h.update(np.ascontiguousarray(a).tobytes())
    return (int(r["game_id"]), int(r["num_moves"]), int(r["result"]), int(r["was_resign"]), h.hexdigest())


def play(rank, world, games, seed=7):
    import alphago_jl_amd as ag
    eng = ag.Engine(board_size=5, tower_height=1, games=4, num_readouts=16, seed=seed, game_id_base=rank,
                    game_id_stride=world, record_capacity_games=games + 8, device=0)
    eng.init_synthetic(0)
    if games > 0:
        eng.start(games)
        while eng.records_count() < games:
            eng.step(8)
    return eng


class FailingLib:
    """the C ABI with one entry point failing: what a rank whose pack step dies looks like to the exchange"""

    def __init__(self, L, status):
        self._L, self._status = L, status

    def __getattr__(self, name):
        if name == "agz_records_packed_size":
            return lambda *a: self._status
        return getattr(self._L, name)


def worker(rank, world, port, per_rank, fail_rank, q):
    for p in (HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import alphago_jl_amd as ag
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = play(rank, world, per_rank[rank])
    out = {"rank": rank, "own": [digest(r) for r in eng.records()]}
    if rank == fail_rank:
        eng.L = FailingLib(eng.L, ag._lib.HIP_ERROR)
    try:
        out["added"] = eng.allgather_records_hosted()
    except ag.AgzError as ex:
        out["error"] = (int(ex.status), str(ex))
    if isinstance(eng.L, FailingLib):
        eng.L = eng.L._L
    if "added" in out:
        out["arena"] = [digest(eng.replay_record(k)) for k in range(eng.replay_count())]
        out["positions"] = eng.replay_positions()
        eng.records_clear()
        out["again"] = eng.allgather_records_hosted()          # nothing new: an exchange of nothing, on every rank
        out["count_after"] = eng.replay_count()
    dist.barrier()
    q.put(out)
    eng.close()
    dist.destroy_process_group()


def run_world(per_rank, fail_rank=-1):
    import torch.multiprocessing as mp
    world = len(per_rank)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, per_rank, fail_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        o = q.get(timeout=900)
        got[o["rank"]] = o
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_two_ranks_on_one_gpu_fill_identical_arenas_through_the_c_abi():
    per_rank = [5, 3]                                       # unequal: rank 1's chunk is padded up to rank 0's
    got = run_world(per_rank)
    a0, a1 = got[0]["arena"], got[1]["arena"]
    assert got[0]["added"] == got[1]["added"] == sum(per_rank) == len(a0)
    assert a0 == a1 and got[0]["positions"] == got[1]["positions"] == sum(d[1] for d in a0)
    assert [d[0] % 2 for d in a0] == [0] * 5 + [1] * 3       # rank order, rank 0's games first
    assert a0[:5] == got[0]["own"] or sorted(a0[:5]) == sorted(got[0]["own"])
    assert sorted(a0[5:]) == sorted(got[1]["own"])
    # the same ids played by ONE rank: ids 0, 2, 4, 6, 8 (rank 0's) and 1, 3, 5 (rank 1's) -- games depend on the id only
    single = play(0, 1, 9)
    want = {d[0]: d for d in (digest(r) for r in single.records())}
    single.close()
    assert all(d == want[d[0]] for d in a0), "a sharded game differs from the single-rank game with the same id"
    for r in (0, 1):
        assert got[r]["again"] == 0 and got[r]["count_after"] == sum(per_rank)


def test_three_ranks_one_of_them_with_nothing_to_send():
    per_rank = [2, 0, 4]
    got = run_world(per_rank)
    arenas = [got[r]["arena"] for r in range(3)]
    assert arenas[0] == arenas[1] == arenas[2] and len(arenas[0]) == 6
    assert [d[0] % 3 for d in arenas[0]] == [0, 0, 2, 2, 2, 2]
    assert all(got[r]["added"] == 6 for r in range(3))


def test_eight_ranks_with_sixteen_or_more_games_each():
    """BASELINE configs[2]'s world size (8 ranks; here all on one GPU) with unequal, non-trivial loads -- 16 to 23 finished
    games per rank and one rank with none: every arena is the same 129 games in rank order"""
    per_rank = [17, 16, 20, 0, 19, 16, 23, 18]
    got = run_world(per_rank)
    arenas = [got[r]["arena"] for r in range(8)]
    assert all(a == arenas[0] for a in arenas) and len(arenas[0]) == sum(per_rank)
    assert [d[0] % 8 for d in arenas[0]] == [r for r, n in enumerate(per_rank) for _ in range(n)]
    assert all(got[r]["added"] == sum(per_rank) and got[r]["positions"] == got[0]["positions"] for r in range(8))
    for r in range(8):
        assert sorted(d for d in arenas[0] if d[0] % 8 == r) == sorted(got[r]["own"])
        assert got[r]["again"] == 0 and got[r]["count_after"] == sum(per_rank)


def test_a_rank_failing_before_the_exchange_fails_every_rank():
    got = run_world([2, 2], fail_rank=1)
    import alphago_jl_amd as ag
    assert "error" in got[0] and "error" in got[1]
    assert got[1]["error"][0] == ag._lib.HIP_ERROR                       # its own failure
    assert got[0]["error"][0] == ag._lib.RCCL_ERROR and "rank 1 failed before the exchange" in got[0]["error"][1]


def test_bench_two_ranks_single_device_runs_the_whole_exchange_leg():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --single-device-test`: the contract line with
    an `exchange` object whose arena holds every rank's games"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device-test",
           "--board", "5", "--tower", "1", "--readouts", "16", "--games", "16", "--steps", "30", "--warmup", "2",
           "--stagger", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 30
    ex = d["exchange"]
    assert "error" not in ex, ex
    assert ex["consistent"] and ex["games_in_arena"] == sum(ex["own_games_by_rank"]) > 0
    assert min(ex["own_games_by_rank"]) >= 16 and ex["positions_in_arena"] > 0


def test_bench_eight_ranks_single_device_exchanges_sixteen_games_per_rank():
    """the same leg at BASELINE configs[2]'s world size: 8 ranks under torch.distributed.run (all on cuda:0), every rank
    with >= 16 finished games before the exchange"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--single-device-test",
           "--board", "5", "--tower", "1", "--readouts", "16", "--games", "32", "--steps", "30", "--warmup", "2",
           "--stagger", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    ex = d["exchange"]
    assert "error" not in ex, ex
    assert ex["consistent"] and ex["games_in_arena"] == sum(ex["own_games_by_rank"])
    assert len(ex["own_games_by_rank"]) == 8 and min(ex["own_games_by_rank"]) >= 16, ex


def test_bench_starts_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 2 --single-device-test` with WORLD_SIZE unset (how the driver starts the N = 1 line): bench.py
    is its own launcher -- two ranks, one JSON line from rank 0, per-rank rates and the exchange object"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device-test", "--board", "5", "--tower", "1",
           "--readouts", "16", "--games", "16", "--steps", "5", "--warmup", "2", "--stagger", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["value"] > 0 and d["scaling"] == "weak"
    assert [p["rank"] for p in d["per_rank"]] == [0, 1] and all(p["positions_per_s"] > 0 for p in d["per_rank"])
    assert abs(sum(p["positions"] for p in d["per_rank"]) - d["positions"]) < 0.5
    ex = d["exchange"]
    assert "error" not in ex, ex
    assert ex["consistent"] is True and ex["ms"] > 0 and ex["games_in_arena"] == sum(ex["own_games_by_rank"]) > 0
    # the line describes itself: bytes exchanged, the same box's one-rank rate, the efficiency against it
    assert ex["bytes"] == sum(ex["bytes_by_rank"]) > 0 and len(ex["bytes_by_rank"]) == 2
    solo = d["single_rank_same_box"]
    assert solo["positions_per_s"] > 0
    assert abs(d["weak_scaling_efficiency"] - min(p["positions_per_s"] for p in d["per_rank"]) / solo["positions_per_s"]) < 1e-9
    assert "search_kernels" in d and set(d["search_kernels"]["kernels"]) == {"k_pre", "k_expand", "k_scan", "k_leaf_features", "k_post"}


# ---------------------------------------------------------------------------------------------------------------------
# The real thing: RCCL, one rank per GPU, as many ranks as this box has devices (up to 8).  A 1-GPU box skips these; the
# driver's multi-GPU box runs them without anyone asking.  Same checks as the gloo rehearsals above.

def visible_devices():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def rccl_worker(rank, world, port, per_rank, q):
    for p in (HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    import alphago_jl_amd as ag
    dist.init_process_group("gloo", rank=rank, world_size=world)          # carries the 128-byte RCCL id, nothing else
    eng = ag.Engine(board_size=5, tower_height=1, games=4, num_readouts=16, seed=7, game_id_base=rank,
                    game_id_stride=world, record_capacity_games=per_rank[rank] + 8, device=rank)
    eng.init_synthetic(rank)                                              # every rank starts from DIFFERENT weights ...
    ids = [ag.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    comm = eng.comm_create(rank, world, ids[0])
    n = eng.broadcast_weights(comm, 0)                                    # ... and plays with rank 0's (ncclBroadcast)
    out = {"rank": rank, "bcast": n,
           "weights": hashlib.sha256(b"".join(np.ascontiguousarray(eng.get_weights(l, k), np.float32).tobytes()
                                              for l, k in eng.layers())).hexdigest()}
    if per_rank[rank] > 0:
        eng.start(per_rank[rank])
        while eng.records_count() < per_rank[rank]:
            eng.step(8)
    out["own"] = [digest(r) for r in eng.records()]
    out["added"] = eng.allgather_records(comm)                            # ncclAllGather x 2, device to device
    out["arena"] = [digest(eng.replay_record(k)) for k in range(eng.replay_count())]
    out["positions"] = eng.replay_positions()
    eng.records_clear()
    out["again"] = eng.allgather_records(comm)
    dist.barrier()
    q.put(out)
    eng.comm_destroy(comm)
    eng.close()
    dist.destroy_process_group()


def run_rccl_world(per_rank):
    import torch.multiprocessing as mp
    world = len(per_rank)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=rccl_worker, args=(r, world, port, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        o = q.get(timeout=900)
        got[o["rank"]] = o
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_rccl_worker_with_a_world_of_one():
    """the worker of the multi-GPU test below, on the one GPU every box has: its code path cannot rot unseen"""
    got = run_rccl_world([3])
    assert got[0]["added"] == 3 == len(got[0]["arena"]) and got[0]["again"] == 0 and got[0]["bcast"] > 0
    assert sorted(got[0]["arena"]) == sorted(got[0]["own"])


@pytest.mark.skipif(visible_devices() < 2, reason="RCCL refuses two ranks on one device: needs >= 2 visible GPUs")
def test_rccl_exchange_and_broadcast_with_one_rank_per_visible_gpu():
    """agz_comm_create + agz_broadcast_weights + agz_allgather_records over RCCL/xGMI with world = min(devices, 8): every
    rank ends with rank 0's weights and with the same arena, rank order, equal to a single-rank engine's games of the same
    ids (BASELINE configs[2]'s exchange; caller /root/reference/src/train.jl:56-66)"""
    world = min(visible_devices(), 8)
    per_rank = [5, 3, 4, 0, 6, 2, 3, 4][:world]
    got = run_rccl_world(per_rank)
    assert len({got[r]["weights"] for r in range(world)}) == 1 and all(got[r]["bcast"] > 0 for r in range(world))
    arenas = [got[r]["arena"] for r in range(world)]
    assert all(a == arenas[0] for a in arenas) and len(arenas[0]) == sum(per_rank)
    assert [d[0] % world for d in arenas[0]] == [r for r, n in enumerate(per_rank) for _ in range(n)]
    for r in range(world):
        assert got[r]["added"] == sum(per_rank) and got[r]["again"] == 0 and got[r]["positions"] == got[0]["positions"]
        assert sorted(d for d in arenas[0] if d[0] % world == r) == sorted(got[r]["own"])
    import alphago_jl_amd as ag
    single = ag.Engine(board_size=5, tower_height=1, games=4, num_readouts=16, seed=7, record_capacity_games=64, device=0)
    single.init_synthetic(0)
    want_ids = sorted(d[0] for d in arenas[0])
    single.start(max(want_ids) + 1)
    while single.records_count() < max(want_ids) + 1:
        single.step(8)
    want = {d[0]: d for d in (digest(r) for r in single.records())}
    single.close()
    assert all(d == want[d[0]] for d in arenas[0]), "a sharded game differs from the single-rank game with the same id"


@pytest.mark.skipif(visible_devices() < 2, reason="needs >= 2 visible GPUs")
def test_bench_self_launched_over_rccl_on_every_visible_gpu():
    """`python bench.py --gpus W` (no launcher, no --single-device-test): W ranks on W GPUs, torch's nccl(=RCCL) group for
    the scalars, libagz's own RCCL communicator for the replay exchange"""
    world = min(visible_devices(), 8)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--board", "5", "--tower", "1",
           "--readouts", "16", "--games", "16", "--steps", "20", "--warmup", "2", "--stagger", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and len(d["per_rank"]) == world and d["value"] > 0
    ex = d["exchange"]
    assert "error" not in ex, ex
    assert ex["consistent"] and "ncclAllGather" in ex["collective"] and ex["games_in_arena"] == sum(ex["own_games_by_rank"])
