"""Two generations of the reference's train() loop (/root/reference/src/train.jl:56-74,86-89) closed on the device, at
5x5 / tower 1 sizes:

    self-play (G games, weights theta0)  ->  replay arena (agz_allgather_records, world of one)
    -> 3 x agz_train_step on batches sampled from the arena (agz_replay_batch)
    -> self-play again on theta1 -- the weights the trainer left in the device master, no host repack in between

Checked: generation 2 is the ORACLE's self-play on theta1 -- theta1 read back through agz_net_get_weights, handed to a
second engine through agz_net_set_weights, and that engine's forward given to the oracle's own tree search: every game
move for move, pi and q bit for bit; generation 1 likewise on theta0; theta1 != theta0; and a BSON save -> load round trip
of theta1 (weights/agz_*.bson as Flux.loadparams! reads them, train.jl:14-35) reproduces generation 2's games as well."""
import os
import tempfile

import numpy as np
import pytest

import alphago_jl_amd as ag
from alphago_jl_amd import bson_weights as bw
from gpu_common import GpuNetForOracle
from test_hostsim_selfplay import bits_equal, oracle_game

pytestmark = pytest.mark.gpu
N, TOWER, R, G, SEED = 5, 1, 16, 6, 11


def play_generation(eng):
    eng.start(G)
    for _ in range(20000):
        eng.step(8)
        if eng.records_count() >= G:
            break
    recs = eng.records()
    assert len(recs) == G and eng.stats()["pool_short_searches"] == 0
    return recs


def assert_oracle_plays_the_same(recs, fwd_engine, what):
    net = GpuNetForOracle(fwd_engine)
    for r in recs:
        o = oracle_game(N, net, R, SEED, int(r["game_id"]))
        assert r["num_moves"] == o["num_moves"] and (r["moves"] == o["moves"][: r["num_moves"]]).all(), what
        assert bits_equal(r["pis"], o["pis"]) and bits_equal(r["qs"], o["qs"]) and r["result"] == o["result"], what


def test_selfplay_train_selfplay_second_generation_is_the_oracles_on_the_new_weights():
    eng = ag.Engine(board_size=N, tower_height=TOWER, games=3, num_readouts=R, seed=SEED, record_capacity_games=G + 8)
    eng.init_synthetic(0)
    fwd = ag.Engine(board_size=N, tower_height=TOWER, games=1, num_readouts=8, max_nodes_per_game=16)
    eng.copy_weights_to(fwd)
    theta0 = {lk: eng.get_weights(*lk).copy() for lk in eng.layers()}

    # ---- generation 1 on theta0
    gen1 = play_generation(eng)
    assert_oracle_plays_the_same(gen1, fwd, "generation 1")

    # ---- finished games -> replay arena -> three optimisation steps (train.jl:56-70)
    added = eng.allgather_records(None)                    # a world of one: the engine files its own records
    eng.records_clear()
    assert added == G and eng.replay_count() == G
    lens = np.array([eng.replay_record(k)["num_moves"] for k in range(G)])
    assert lens.sum() == sum(r["num_moves"] for r in gen1) == eng.replay_positions()
    rng = np.random.default_rng(1)
    B = int(min(16, lens.sum()))
    losses = []
    for _ in range(3):
        flat = rng.choice(int(lens.sum()), size=B, replace=False)             # sample(1:n, B, replace = false), train.jl:5
        game = np.searchsorted(np.cumsum(lens), flat, side="right")
        ply = flat - (np.cumsum(lens) - lens)[game]
        f, p, z = eng.replay_batch(game, ply)
        losses.append(eng.train_step(f, p, z)[0])
    assert np.isfinite(losses).all()

    # ---- generation 2 on theta1, straight from the device master
    gen2 = play_generation(eng)
    theta1 = {lk: eng.get_weights(*lk).copy() for lk in eng.layers()}
    changed = [lk for lk in theta0 if not np.array_equal(theta0[lk], theta1[lk])]
    assert (0, 0) in changed and (1, 4) in changed and (1, 5) in changed, "weights and running statistics moved"
    for lk, w in theta1.items():                           # theta1 through the C ABI into the second engine
        fwd.set_weights(lk[0], lk[1], w)
    assert_oracle_plays_the_same(gen2, fwd, "generation 2")
    assert any(a["num_moves"] != b["num_moves"] or not np.array_equal(a["moves"], b["moves"]) for a, b in zip(gen1, gen2)), \
        "the same game ids played on different weights"

    # ---- BSON round trip of theta1 (save_model / load_model): a third engine, same games
    with tempfile.TemporaryDirectory() as d:
        bw.write_checkpoint(d, bw.extract_param_lists(eng))
        assert sorted(os.listdir(os.path.join(d, "weights")))[:3] == ["agz_base.bson", "agz_base_bnstats.bson", "agz_policy.bson"]
        ck = bw.read_checkpoint(d)
        third = ag.Engine(board_size=N, tower_height=TOWER, games=1, num_readouts=8, max_nodes_per_game=16)
        bw.apply_param_lists(third, ck["base"], ck["value"], ck["policy"], ck.get("base_stats"), ck.get("value_stats"),
                             ck.get("policy_stats"))
    for lk, w in theta1.items():
        assert np.array_equal(third.get_weights(*lk), w), f"checkpoint changed {lk}"
    assert_oracle_plays_the_same(gen2[:2], third, "generation 2 from the checkpoint")
    third.close()
    fwd.close()
    eng.close()
