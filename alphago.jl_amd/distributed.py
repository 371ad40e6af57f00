"""Replay all-gather of finished self-play games.

Self-play shards by game id with no collective on the data path; the ONE exchange step is the
all-gather of completed (state, pi, z) records into every rank's replay buffer.  Records travel
in the packed layout of agz_records_export_packed:

    per game:  agz_game_header (32 B) | moves int16[n] | pad to 4 | pis float32[n][A] | qs float32[n] | pad to 8

(the board states are rebuilt from the move list, exactly as the reference's extract_data does
through replay_position, board.jl:557-578).  Sizes differ per rank, so the counts are exchanged
first and the payload is padded to the maximum -- one small and one large collective.  Backend
"nccl" is RCCL over xGMI on the GPU box; "gloo" is what the CPU tests use."""
import struct

import numpy as np

HEADER = struct.Struct("<Qiiiifi")   # game_id, num_moves, result, was_resign, resign_disabled, final_score, short_searches
assert HEADER.size == 32


def pack_records(records, A):
    """records: iterable of dicts (game_id, result, was_resign, resign_disabled, final_score, moves, pis, qs)"""
    out = bytearray()
    for r in records:
        n = int(len(r["moves"]))
        out += HEADER.pack(int(r["game_id"]), n, int(r["result"]), int(r["was_resign"]),
                           int(r.get("resign_disabled", 0)), float(r.get("final_score", 0.0)), int(r.get("short_searches", 0)))
        out += np.ascontiguousarray(r["moves"], np.int16).tobytes()
        out += b"\0" * (-len(out) % 4)
        pis = np.ascontiguousarray(r["pis"], np.float32).reshape(n, A) if n else np.zeros((0, A), np.float32)
        out += pis.tobytes()
        out += np.ascontiguousarray(r["qs"], np.float32).tobytes()
        out += b"\0" * (-len(out) % 8)
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def unpack_records(buf, A):
    buf = np.ascontiguousarray(buf, np.uint8)
    raw = buf.tobytes()
    out, off = [], 0
    while off + HEADER.size <= len(raw):
        game_id, n, result, was_resign, resign_disabled, final_score, short = HEADER.unpack_from(raw, off)
        off += HEADER.size
        moves = np.frombuffer(raw, np.int16, n, off).copy()
        off += 2 * n
        off += -off % 4
        pis = np.frombuffer(raw, np.float32, n * A, off).reshape(n, A).copy()
        off += 4 * n * A
        qs = np.frombuffer(raw, np.float32, n, off).copy()
        off += 4 * n
        off += -off % 8
        out.append(dict(game_id=game_id, num_moves=n, result=result, was_resign=was_resign,
                        resign_disabled=resign_disabled, final_score=final_score, short_searches=short, moves=moves, pis=pis,
                        qs=qs))
    return out


def allgather_packed(packed, group=None, device=None, force_collective=False):
    """all-gather one uint8 buffer per rank; returns the concatenation in rank order (numpy uint8).
    `packed` is a numpy array or a torch tensor (a CUDA tensor from Engine.records_packed_device()
    goes into the collective as it is -- device to device over RCCL/xGMI)."""
    import torch
    import torch.distributed as dist

    is_tensor = isinstance(packed, torch.Tensor)
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective):
        return packed.cpu().numpy() if is_tensor else np.ascontiguousarray(packed, np.uint8)
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = packed.to(device) if is_tensor else torch.from_numpy(np.ascontiguousarray(packed, np.uint8)).to(device)
    size = torch.tensor([mine.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    padded = torch.zeros(mx, dtype=torch.uint8, device=device)
    padded[: mine.numel()] = mine
    parts = [torch.empty(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return np.concatenate([p[:s].cpu().numpy() for p, s in zip(parts, sizes)]) if sum(sizes) else np.zeros(0, np.uint8)


def allgather_records(engine_or_records, A, group=None, force_collective=False):
    """engine or a list of record dicts -> every rank's records, by game id.  With the nccl (RCCL)
    backend an engine's records go from its HBM arena into the collective without touching the host."""
    import torch.distributed as dist

    if hasattr(engine_or_records, "records_packed"):
        on_device = dist.is_initialized() and dist.get_backend(group) == "nccl" and hasattr(engine_or_records, "records_packed_device")
        packed = engine_or_records.records_packed_device() if on_device else engine_or_records.records_packed()
    else:
        packed = pack_records(engine_or_records, A)
    recs = unpack_records(allgather_packed(packed, group, force_collective=force_collective), A)
    return sorted(recs, key=lambda r: r["game_id"])


def broadcast_weights(engine, src=0, group=None):
    """The optional second collective of SURVEY.md 8e: after a training step on rank `src`, every
    rank's weight replica is overwritten with that rank's parameters (one flat float32 broadcast --
    12-24 M parameters, < 100 MB: a single RCCL call).  `engine` needs layers() / get_weights() /
    set_weights() (alphago.jl_amd.Engine, or NeuralNet.engine)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    keys = list(engine.layers())
    parts = [np.ascontiguousarray(engine.get_weights(l, k), np.float32).ravel() for l, k in keys]
    sizes = [p.size for p in parts]
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    flat = torch.from_numpy(np.concatenate(parts)).to(device)
    dist.broadcast(flat, src=src, group=group)
    if dist.get_rank(group) != src:
        host = flat.cpu().numpy()
        off = 0
        for (l, k), n in zip(keys, sizes):
            engine.set_weights(l, k, host[off:off + n])
            off += n
    return int(flat.numel())
