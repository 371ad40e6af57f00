"""SURVEY.md 8e behind the C ABI: agz_comm_* / agz_allgather_records / agz_broadcast_weights (RCCL bound at
run time inside libagz.so) and the device replay arena agz_replay_* the gathered games land in.

One GPU on this box, so the RCCL communicator has one rank -- that still runs the real ncclCommInitRank /
ncclAllGather / ncclBroadcast calls and every step of the exchange (pack kernel, count exchange, padded gather,
on-device record indexing, compaction).  The arithmetic of several ranks' chunks is exercised by ingesting the
packed exports of a second engine that plays the other half of the game ids ("rank 1" of a 2-rank shard)."""
import ctypes as C

import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from test_hostsim_selfplay import bits_equal

pytestmark = pytest.mark.gpu


def play(rank, world, games, N=5, tower=1, readouts=16, seed=7):
    eng = ag.Engine(board_size=N, tower_height=tower, games=4, num_readouts=readouts, seed=seed,
                    game_id_base=rank, game_id_stride=world, record_capacity_games=games + 8)
    eng.init_synthetic(0)
    eng.start(games)
    while eng.records_count() < games:
        eng.step(8)
    return eng


def same_record(a, b):
    return (a["game_id"] == b["game_id"] and a["num_moves"] == b["num_moves"] and a["result"] == b["result"]
            and a["was_resign"] == b["was_resign"] and (a["moves"] == b["moves"]).all()
            and bits_equal(a["pis"], b["pis"]) and bits_equal(a["qs"], b["qs"])
            and np.float32(a["final_score"]) == np.float32(b["final_score"]))


def test_allgather_records_through_the_c_abi_over_rccl():
    e0 = play(0, 2, 6)
    e1 = play(1, 2, 5)
    want0, want1 = e0.records(), e1.records()
    comm = e0.comm_create(0, 1, ag.comm_unique_id())
    assert e0.replay_count() == 0
    assert e0.allgather_records(comm) == 6                       # RCCL all-gather, world of one rank
    assert e0.replay_count() == 6 and e0.replay_positions() == sum(r["num_moves"] for r in want0)
    by_id = {int(r["game_id"]): r for r in want0 + want1}        # (records() sorts by game id, the arena keeps arrival order)
    for k in range(6):
        got = e0.replay_record(k)
        assert same_record(got, by_id[int(got["game_id"])]), k
    # "rank 1": its packed export enters the same arena, once from host memory and once more from device memory
    assert e0.replay_ingest(e1.records_packed()) == 5
    dev = e1.records_packed_device()
    assert e0.replay_ingest((dev.data_ptr(), dev.numel())) == 5
    assert e0.replay_count() == 16
    for k in range(5):
        a, b = e0.replay_record(6 + k), e0.replay_record(11 + k)
        assert int(a["game_id"]) % 2 == 1 and same_record(a, by_id[int(a["game_id"])]) and same_record(b, a), k
    ids = sorted(e0.replay_record(k)["game_id"] for k in range(11))
    assert ids == list(range(11))                                # game ids r, r + W, ...: the shards interleave
    # records this call has filed once are not sent again (a second call used to re-ingest every game) ...
    assert e0.allgather_records(comm) == 0 and e0.replay_count() == 16
    assert e0.allgather_records(None) == 0 and e0.replay_count() == 16          # ... nor by the communicator-less form
    # a second generation appends; the engine's own ring is the caller's to clear
    e0.records_clear()
    assert e0.allgather_records(comm) == 0 and e0.replay_count() == 16
    # get_replay_batch from the arena == replaying the oracle (features), searches_pi[ply], result
    rng = np.random.RandomState(0)
    games = rng.randint(0, 16, size=40)
    plies = np.array([rng.randint(0, max(1, e0.replay_record(int(g))["num_moves"])) for g in games], np.int32)
    ok = np.array([e0.replay_record(int(g))["num_moves"] > 0 for g in games])
    games, plies = games[ok], plies[ok]
    feats, pi, z = e0.replay_batch(games, plies)
    for b, (g, j) in enumerate(zip(games, plies)):
        r = e0.replay_record(int(g))
        pos = orc.make_pos(5)
        for k in range(j):
            _, pos = orc.play(pos, int(r["moves"][k]))
        assert (feats[b].reshape(17, 25) == orc.feats(pos)).all(), (g, j)
        assert bits_equal(pi[b], r["pis"][j]) and z[b] == r["result"]
    # errors are reported, not swallowed
    with pytest.raises(ag.AgzError):
        e0.replay_batch([0], [e0.replay_record(0)["num_moves"]])          # ply == num_moves has no searches_pi
    with pytest.raises(ag.AgzError):
        e0.replay_batch([16], [0])
    bad = e1.records_packed().copy()
    bad[8:12] = 255                                                       # num_moves = -1 in the first header
    with pytest.raises(ag.AgzError):
        e0.replay_ingest(bad)
    assert e0.replay_count() == 16
    # `shrink` (train.jl:52): the window is measured in positions, whole oldest games go
    total = e0.replay_positions()
    first = e0.replay_record(0)["num_moves"]
    second = e0.replay_record(1)
    e0.replay_trim(total - 1)
    assert e0.replay_count() == 15 and e0.replay_positions() == total - first
    assert same_record(e0.replay_record(0), second)
    f2, p2, z2 = e0.replay_batch([0], [0])
    assert z2[0] == second["result"] and bits_equal(p2[0], second["pis"][0])
    e0.replay_clear()
    assert e0.replay_count() == 0 and e0.replay_positions() == 0
    e0.comm_destroy(comm)
    e0.close()
    e1.close()


def test_ingest_of_a_padded_multi_rank_gather_buffer():
    """the second half of agz_allgather_records for a world of three ranks (one of them with nothing to send): the
    receive buffer of the padded all-gather is laid out by hand from three engines' packed exports"""
    engs = [play(r, 3, n) for r, n in ((0, 4), (1, 3))]
    packs = [e.records_packed() for e in engs] + [np.zeros(0, np.uint8)]
    counts = [(e.records_count(), p.size) for e, p in zip(engs, packs)] + [(0, 0)]
    stride = (max(p.size for p in packs) + 255) // 256 * 256
    buf = np.full(3 * stride, 0xAB, np.uint8)                    # the padding is garbage and must never be read
    for r, p in enumerate(packs):
        buf[r * stride: r * stride + p.size] = p
    dst = engs[0]
    assert dst.replay_ingest_gathered(buf, 3, stride, counts) == 7 and dst.replay_count() == 7
    want = {int(r["game_id"]): r for e in engs for r in e.records()}
    got = [dst.replay_record(k) for k in range(7)]
    assert [int(r["game_id"]) % 3 for r in got] == [0, 0, 0, 0, 1, 1, 1]          # rank order
    assert all(same_record(r, want[int(r["game_id"])]) for r in got)
    bad = list(counts)
    bad[1] = (counts[1][0] + 1, counts[1][1])                    # a rank that announces one record too many
    with pytest.raises(ag.AgzError):
        dst.replay_ingest_gathered(buf, 3, stride, bad)
    assert dst.replay_count() == 7
    for e in engs:
        e.close()


def test_broadcast_weights_through_the_c_abi_over_rccl():
    eng = ag.Engine(board_size=5, tower_height=1, games=1, num_readouts=8, max_nodes_per_game=16)
    eng.init_synthetic(4)
    before = {(l, k): eng.get_weights(l, k).copy() for l, k in eng.layers()}
    comm = eng.comm_create(0, 1, ag.comm_unique_id())
    n = eng.broadcast_weights(comm, 0)
    assert n == sum(v.size for v in before.values())
    for (l, k), v in before.items():
        assert (eng.get_weights(l, k) == v).all()
    with pytest.raises(ag.AgzError) as ei:
        eng.broadcast_weights(comm, 3)
    assert ei.value.status == ag._lib.BAD_ARGUMENT
    eng.comm_destroy(comm)
    eng.close()


def test_comm_create_rejects_bad_ranks():
    eng = ag.Engine(board_size=5, tower_height=0, games=1, num_readouts=8, max_nodes_per_game=16)
    with pytest.raises(ag.AgzError):
        eng.comm_create(2, 2, ag.comm_unique_id())
    eng.close()
