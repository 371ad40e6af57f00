"""BASELINE.json configs[3] and configs[4] at their real network depth and shard shape.

configs[3]: GoEnv(19), tower_height=20, 800 readouts, 2048 games on 8 GPUs  -> 256 games per GPU,
            batches of up to 2048 leaves, exact-f32 tower (Winograd F(3x3,3x3): a 19x19 board is 7x7
            tiles over a 21x21 padded board, 40 convolutions deep -- where transform error would
            accumulate if it did).
configs[4]: same network on the fp16 MFMA path, 1600 readouts, 4096 games -> 512 games per GPU,
            batches of up to 4096 leaves.

Three kinds of check (VERDICT r1 "Next round" #1):
  (i)   forward parity at (N=19, tower=20, B=16) against the float64 oracle: f32 <= 1e-4 with both
        convolution algorithms; fp16 tower against the oracle's restatement of that arithmetic and
        against f64, measured error printed, bars stated below;
  (ii)  size-independent invariants of ~30 steps of the real shard shape (no oracle needed);
  (iii) the first moves of a whole 19x19 / tower-20 game, engine vs oracle tree, bit for bit
        (the oracle's network callable is the same HIP forward, see tests/gpu_common.py)."""
import ctypes as C

import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from gpu_common import GpuNetForOracle, copy_weights_from_oracle, pos_soa
from test_gpu_nn import oracle_forward64
from test_gpu_tree import compare_trees
from test_hostsim_go import random_positions
from test_oracle_nn import randomize_bn

pytestmark = pytest.mark.gpu
L = orc.lib()
N19, T20 = 19, 20
TOL = 1e-4                    # north_star: policy/value tensors within 1e-4 fp32
TOLMIX = 1e-2                 # fp16 tower vs the exact f64 network (test_gpu_nn16.py); measured 8.2e-3 here


def _deep_net(seed):
    rng = np.random.RandomState(seed)
    onet = L.or_net_new(N19, T20)
    L.or_net_init_synthetic(onet, 3)
    randomize_bn(onet, list(range(0, 1 + 2 * T20)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], rng)
    return onet, rng


def _positions(rng, B):
    positions = random_positions(N19, 3, 150, seed=5)
    return [positions[i] for i in rng.choice(len(positions), B, replace=False)]


@pytest.mark.parametrize("winograd", [1, 2, 0])      # 1: F(4x4,3x3) at this size; 2: F(3x3,3x3) only (dense blocks); 0: direct
def test_c4_forward_f32_tower20_19x19(winograd):
    B, A = 16, N19 * N19 + 1
    onet, rng = _deep_net(41)
    eng = ag.Engine(board_size=N19, games=1, tower_height=T20, num_readouts=8, max_nodes_per_game=16)
    eng.set_winograd(winograd)
    copy_weights_from_oracle(eng, onet, T20)
    positions = _positions(rng, B)
    feats = np.stack([orc.feats(p).reshape(-1) for p in positions])
    pi64, v64 = oracle_forward64(onet, feats, A)
    gpi, gv = eng.forward(*pos_soa(positions))
    dpi, dv = np.abs(gpi - pi64).max(), np.abs(gv - v64).max()
    print(f"19x19 tower 20 f32 winograd={winograd}: max|dpi|={dpi:.2e} max|dv|={dv:.2e} "
          f"(pi max {pi64.max():.3f}, |v| max {np.abs(v64).max():.3f})")
    assert dpi <= TOL and dv <= TOL, (dpi, dv)
    assert np.allclose(gpi.sum(1), 1, atol=1e-5)
    # the answer for a position does not depend on its batch row, also at this size
    perm = rng.permutation(B)
    ppi, pv = eng.forward(*pos_soa([positions[i] for i in perm]))
    assert (ppi == gpi[perm]).all() and (pv == gv[perm]).all()
    L.or_net_free(onet)
    eng.close()


def test_c5_forward_f16_tower20_19x19():
    B, A = 16, N19 * N19 + 1
    onet, rng = _deep_net(42)
    eng = ag.Engine(board_size=N19, games=1, tower_height=T20, num_readouts=8, max_nodes_per_game=16)
    copy_weights_from_oracle(eng, onet, T20)
    eng.set_precision("f16")
    positions = _positions(rng, B)
    feats = np.stack([orc.feats(p).reshape(-1) for p in positions]).astype(np.float32)
    gpi, gv = eng.forward(*pos_soa(positions))
    pi16, v16 = np.zeros((B, A), np.float32), np.zeros(B, np.float32)
    L.or_net_forward_feats(onet, orc.fptr(feats), B, orc.fptr(pi16), orc.fptr(v16), 16)
    pi64, v64 = oracle_forward64(onet, feats, A)
    # 40 convolutions deep the 5e-4 bar "GPU vs the oracle's restatement of the fp16 arithmetic" of the shallow
    # tests does not exist any more: f32-vs-f64 accumulation moves an activation across a half rounding boundary
    # now and then (one half ulp), and those flips propagate through the remaining layers exactly like the fp16
    # storage error itself does (measured: GPU vs restatement 1.1e-2, each of them vs f64 8e-3).  What can be
    # held is (a) the stated bar against the exact network and (b) that the GPU's fp16 arithmetic is no further
    # from the exact network than the oracle's restatement of it is.
    d16 = max(np.abs(gpi - pi16).max(), np.abs(gv - v16).max())
    dmix = max(np.abs(gpi - pi64).max(), np.abs(gv - v64).max())
    dref = max(np.abs(pi16 - pi64).max(), np.abs(v16 - v64).max())
    print(f"19x19 tower 20 fp16 tower: GPU vs f64 {dmix:.2e} (bar {TOLMIX}); oracle fp16 restatement vs f64 {dref:.2e}; "
          f"GPU vs restatement {d16:.2e}")
    assert dmix <= TOLMIX, dmix
    assert dmix <= 2.0 * dref + 1e-3, (dmix, dref)
    assert d16 <= dmix + dref + 1e-6
    assert np.allclose(gpi.sum(1), 1, atol=1e-5)
    L.or_net_free(onet)
    eng.close()


TOL16_BLOCK = 1e-3            # fp16 tower, ONE residual block, GPU vs the oracle's restatement of the same arithmetic


def test_c5_one_block_f16_full_batch_vs_restatement():
    """configs[4]'s shape without its depth (VERDICT r3 #3a): 19x19, B = 4096 positions in one call (the shard's batch:
    6601 tiles of 224 board-point rows over 256 persistent workgroups), ONE residual block -- two fp16 convolutions,
    nothing for half rounding to amplify through -- against the oracle's restatement of this arithmetic (half rounding
    where the GPU stores, float64 sums) at <= 1e-3.  The oracle evaluates a sample of the batch (first and last
    positions, the ones around the middle, random others: 0.9 GFLOP per position on the CPU); the rest of the batch is
    tied to small-batch evaluations bit for bit by test_gpu_nn.py's full-batch screen at this very shape.  This is what
    separates "the fp16 kernel is correct at full size" from "depth amplifies half rounding" (the tower-20 test above)."""
    B, A, tower = 4096, N19 * N19 + 1, 1
    rng = np.random.RandomState(77)
    onet = L.or_net_new(N19, tower)
    L.or_net_init_synthetic(onet, 3)
    randomize_bn(onet, list(range(0, 1 + 2 * tower)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV], rng)
    eng = ag.Engine(board_size=N19, games=1, tower_height=tower, num_readouts=8, max_nodes_per_game=16)
    copy_weights_from_oracle(eng, onet, tower)
    eng.set_precision("f16")
    pool = random_positions(N19, 6, 150, seed=9)
    feats_pool = np.stack([orc.feats(p).reshape(-1) for p in pool]).astype(np.float32)
    feats = feats_pool[rng.randint(0, len(pool), B)]
    gpi, gv = eng.forward_features(feats)
    assert np.isfinite(gpi).all() and np.allclose(gpi.sum(1), 1, atol=1e-5)
    sample = sorted({0, 1, 2, B // 2 - 1, B // 2, B - 2, B - 1} | set(int(i) for i in rng.choice(B, 41, replace=False)))
    sf = np.ascontiguousarray(feats[sample])
    n = len(sample)
    pi16, v16 = np.zeros((n, A), np.float32), np.zeros(n, np.float32)
    L.or_net_forward_feats(onet, orc.fptr(sf), n, orc.fptr(pi16), orc.fptr(v16), 16)
    pi64, v64 = oracle_forward64(onet, sf, A)
    d16 = max(np.abs(gpi[sample] - pi16).max(), np.abs(gv[sample] - v16).max())
    dmix = max(np.abs(gpi[sample] - pi64).max(), np.abs(gv[sample] - v64).max())
    print(f"19x19, one block, B={B}, fp16 tower: GPU vs the oracle's fp16 restatement {d16:.2e} (bar {TOL16_BLOCK}), vs f64 {dmix:.2e}")
    assert d16 <= TOL16_BLOCK, d16
    assert dmix <= TOLMIX, dmix
    L.or_net_free(onet)
    eng.close()


def _invariants(eng, G, R, steps, par=8):
    st = eng.stats()
    assert st["pool_exhausted"] == 0 and st["steps"] == steps
    assert st["evals"] <= steps * par * G and st["evals"] >= 0.9 * (steps - 1) * par * G     # the batch stays full
    assert st["root_visits"] >= st["evals"] - G
    assert st["games_started"] >= G
    rng = np.random.RandomState(0)
    for g in rng.choice(G, 12, replace=False):
        g = int(g)
        root = eng.tree_root(g)
        info = eng.node_info(g, root)
        assert eng.pending_vlosses(g) == 0                                           # mcts_play.jl:92
        cn = eng.node_floats(g, root, 0)
        assert (cn >= 0).all() and (cn == np.round(cn)).all()
        if info.is_expanded:
            assert info.N == 1 + cn.sum() or info.N == cn.sum()
            legal = eng.go_legal(eng.node_board(g, root)[None], [info.pos.to_play], [info.pos.ko])[0]
            assert not (cn[legal == 0] > 0).any()                                    # test_mcts.jl:146-167 at scale
            prior = eng.node_floats(g, root, 2)
            assert np.isfinite(prior).all() and abs(prior.sum() - 1.0) < 1e-3
    return st


@pytest.mark.parametrize("precision,R,G", [("f32", 800, 256), ("f16", 1600, 512)])
def test_c4_c5_full_size_shard_invariants(precision, R, G):
    """one GPU's shard of configs[3] (f32, 800 readouts, 256 games) / configs[4] (fp16 tower, 1600
    readouts, 512 games) for 30 steps"""
    steps = 30
    eng = ag.Engine(board_size=N19, tower_height=T20, games=G, num_readouts=R, seed=13, stagger_moves=120)
    eng.init_synthetic(0)
    eng.set_precision(precision)
    eng.start(0)
    eng.step(steps)
    st = _invariants(eng, G, R, steps)
    print(f"configs shard {precision}: {st['evals']} evals in {steps} steps, {st['positions']} positions")
    eng.close()


def test_c4_opening_of_a_whole_game_matches_oracle_tree():
    """19x19 / tower 20: the first 3 moves of a self-play game (48 readouts each, 8 leaves per
    tree_search!) on the engine's tree and on the oracle's tree, both fed by the HIP network:
    the complete trees are compared bit for bit after every search and every move."""
    R, seed, game = 48, 5, 3
    eng = ag.Engine(board_size=N19, games=1, tower_height=T20, num_readouts=R, seed=seed, max_nodes_per_game=4096)
    eng.init_synthetic(0)
    fwd = ag.Engine(board_size=N19, tower_height=T20, games=1, num_readouts=8, max_nodes_per_game=16)
    fwd.init_synthetic(0)
    net = GpuNetForOracle(fwd)
    pos = orc.make_pos(N19)
    op = L.or_player_new(N19, net.cb, None, R, 0, -0.9, seed, game)
    L.or_player_initialize_game(op, C.byref(pos))
    eng.tree_init(0, pos.board_np(), n=0, to_play=1, komi=pos.komi)
    eng.set_draw(0, game, 0)
    env = orc.env(N19)
    A = N19 * N19 + 1
    for move in range(3):
        for _ in range(R // 8):
            no = L.or_player_tree_search(op, 8)
            ns = eng.tree_search(0, 8)
            assert ns == no
        oroot = L.or_player_root(op)
        d = orc.ODraw(seed, game, L.or_node_pos(oroot).contents.n, 0)
        L.or_inject_noise(C.byref(env), oroot, C.byref(d))
        eng.inject_noise(0, eng.tree_root(0))
        nodes = compare_trees_19(eng, oroot, A)
        assert nodes > R // 2
        a = C.c_int()
        so = L.or_player_pick_move(op, C.byref(a))
        st, rs = eng.pick_move(0)
        assert st == so == 0 and rs == a.value
        assert eng.play_move(0, rs) == L.or_player_play_move(op, rs) == 1
    L.or_player_free(op)
    fwd.close()
    eng.close()


def compare_trees_19(eng, oroot, A):
    """test_gpu_tree.compare_trees is written for the module-level 9x9 action count; same walk here"""
    import test_gpu_tree as tt
    old = tt.A
    tt.A = A
    try:
        return compare_trees(eng, 0, eng.tree_root(0), oroot)
    finally:
        tt.A = old


@pytest.mark.parametrize("precision,R", [("f32", 800), ("f16", 1600)])
def test_c4_c5_whole_games_at_their_own_readout_budget(precision, R):
    """configs[3] / configs[4] played to the END once: 19x19, tower 20, 800 readouts (exact f32,
    F(4x4,3x3) tower) / 1600 readouts (fp16 tower), 16 concurrent games on the DEFAULT node pool until 4 of them are
    over -- by resignation, two passes or at move 505 (/root/reference/src/selfplay.jl:22-43, src/mcts.jl:15-25).  (With the
    synthetic tower-20 network the first games to end resign around move 130; the same run with resignation disabled, every
    game to two passes or move 505, is tools/soak_configs.py: ~10 min, log under profiles/.)
    Asserted: no allocation was ever refused and no search shortened (the default pool holds these trees; the peak
    is printed); every record replays legally on the oracle's rules; a game that was not resigned ends by two passes or
    at max_game_length with the oracle's Tromp-Taylor result and score; resigned games have Q(root) below the
    threshold on their last recorded move; every pi is a distribution over moves that were legal."""
    G, WANT = 16, 4
    eng = ag.Engine(board_size=N19, tower_height=T20, games=G, num_readouts=R, seed=17, record_capacity_games=G + 8)
    eng.init_synthetic(0)
    eng.set_precision(precision)
    eng.start(0)                                           # slots recycle: the batch stays 16 games wide
    import time
    t0, steps = time.time(), 0
    per_move = (R + 7) // 8
    while eng.records_count() < WANT and steps < 520 * per_move * 2:
        eng.step(per_move)
        steps += per_move
    st = eng.stats()
    recs = eng.records()
    assert len(recs) >= WANT, f"{len(recs)} games over after {steps} steps"
    assert st["pool_exhausted"] == 0 and st["pool_short_searches"] == 0 and st["stalled_games"] == 0
    assert st["node_capacity"] == 16 * R + 256 + 16 * 505
    ended = {"resign": 0, "passes": 0, "length": 0}
    for r in recs:
        assert r["short_searches"] == 0
        pos = orc.make_pos(N19)
        for k, a in enumerate(r["moves"]):
            legal = orc.legal_moves(pos)
            assert legal[int(a)] == 1 and not (r["pis"][k][legal == 0] > 0).any()
            assert abs(float(r["pis"][k].sum()) - 1.0) < 1e-4
            rc, pos = orc.play(pos, int(a))
            assert rc == orc.OK
        assert pos.n == r["num_moves"] <= 505
        if r["was_resign"]:
            ended["resign"] += 1
            assert not r["resign_disabled"]
        else:
            assert pos.done or pos.n >= 505
            ended["passes" if pos.done else "length"] += 1
            assert r["result"] == L.or_result(C.byref(pos)) and abs(r["final_score"] - L.or_score(C.byref(pos))) < 1e-6
    print(f"configs whole games {precision} R={R}: {len(recs)} games over in {steps} steps / {time.time() - t0:.0f} s, "
          f"{sum(r['num_moves'] for r in recs)} moves, ended {ended}, peak nodes per game {st['peak_nodes_per_game']} of "
          f"{st['node_capacity']} ({st['peak_nodes_per_game'] / st['node_capacity']:.2f}), evals {st['evals']}")
    eng.close()
