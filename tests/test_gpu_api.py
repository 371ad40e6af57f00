"""The reference's own test scenarios written against the host mirror of its API
(alphago.jl_amd/api.py): GoEnv / Position / MCTSPlayer / NeuralNet / selfplay / extract_data.
Each test cites the reference test it transcribes."""
import numpy as np
import pytest

import alphago_jl_amd as ag
from alphago_jl_amd import BLACK, WHITE, GoEnv, MCTSPlayer, NeuralNet, PlayerMove, Position, from_kgs, to_flat
from orc import load_board
from test_oracle_go import ALMOST_DONE

pytestmark = pytest.mark.gpu
N = 9


def board2d(text):
    return load_board(text, N).reshape(N, N).T     # flat p = row + N*col  ->  [row, col]


class DummyNet:  # test/test_mcts_player.jl:10-32
    def __init__(self, env, fake_priors=None, fake_value=0.0):
        self.p = np.ones(env.action_space) / env.action_space if fake_priors is None else np.asarray(fake_priors)
        self.v = fake_value

    def __call__(self, positions):
        if positions is None or len(positions) == 0:
            raise ValueError("No positions passed!")
        B = len(positions)
        return np.tile(self.p[:, None], (1, B)), np.full(B, self.v)


def test_go_rules_through_api():  # test_go.jl:380-516
    env = GoEnv(N)
    EMPTY_ROW = "." * N + "\n"
    start = Position(env, board=board2d(".X.....OO\nX........\n" + EMPTY_ROW * 7), komi=6.5, caps=(1, 2))
    p1 = start.play_move(from_kgs("C9", env))
    assert (p1.board == board2d(".XX....OO\nX........\n" + EMPTY_ROW * 7)).all()
    assert p1.n == 1 and p1.to_play == WHITE and p1.caps == (1, 2)
    kb = Position(env, board=board2d(".OX......\nOX.......\n" + EMPTY_ROW * 7), komi=6.5, caps=(1, 2))
    k1 = kb.play_move(from_kgs("A9", env))
    assert k1.ko == from_kgs("B9", env) and k1.caps == (2, 2)
    with pytest.raises(ag.IllegalMove):
        k1.play_move(from_kgs("B9", env))
    k2 = k1.pass_move().pass_move().play_move(from_kgs("B9", env))
    assert k2.caps == (2, 3) and k2.ko == from_kgs("A9", env) and k2.n == 4
    root = Position(env)
    assert not root.done and not root.play_move(None).done and root.play_move(None).play_move(None).done
    assert root.result_string() == "W+7.5"
    # features.jl / test_features.jl:39-79
    pos = Position(env)
    for c in ((0, 0), (0, 1), (0, 2), (0, 3), (1, 1)):
        pos = pos.play_move(c)
    f = ag.get_feats(pos)
    assert f.shape == (17, N, N) and pos.to_play == WHITE
    assert (f[0] == board2d("...X.....\n" + EMPTY_ROW * 8)).all()
    assert (f[1] == board2d("X.X......\n.X.......\n" + EMPTY_ROW * 7)).all()
    assert (f[10:16] == 0).all() and (f[16] == -1).all()


def almost_done_player(env):  # test_mcts_player.jl:68-77
    probs = np.ones(env.action_space) * 0.001
    probs[2:5] = 0.2
    probs[-1] = 0.2
    player = MCTSPlayer(env, DummyNet(env, fake_priors=probs))
    pos = Position(env, board=board2d(ALMOST_DONE), n=70, komi=2.5, caps=(1, 4),
                   recent=[PlayerMove(BLACK, (0, 1)), PlayerMove(WHITE, (0, 8))], to_play=BLACK)
    player.initialize_game(pos)
    return player


def test_dont_pass_if_losing():  # test_mcts_player.jl:139-165
    env = GoEnv(N)
    player = almost_done_player(env)
    assert player.root.position.score() == -0.5
    for _ in range(20):
        player.tree_search()
    flat = to_flat(from_kgs("D9", env), env)
    root = player.root
    assert int(np.argmax(root.child_N)) == flat
    assert root.children[flat].Q > 0
    assert root.N >= 20
    assert root.child_Q[-1] < 0
    assert player.engine.pending_vlosses(0) == 0


def test_cold_start_and_extract_data():  # test_mcts_player.jl:227-240, 285-321
    env = GoEnv(N)
    player = MCTSPlayer(env, DummyNet(env, fake_value=0.17))
    player.initialize_game()
    assert player.root.N == 0 and not player.root.is_expanded
    player.tree_search(4)
    assert player.root.N == 1 and player.root.Q == pytest.approx(0.085)
    player = MCTSPlayer(env, DummyNet(env))
    player.initialize_game()
    player.tree_search()
    assert player.play_move(None)
    player.tree_search()
    assert player.play_move(None)
    assert player.is_done()
    player.set_result(player.root.position.result(), False)
    positions, pis, results = player.extract_data()
    assert len(positions) == len(pis) == len(results) == 2
    assert results[0] == WHITE and player.result_string == "W+7.5"
    player = MCTSPlayer(env, DummyNet(env))
    player.initialize_game()
    player.tree_search()
    player.play_move((0, 0))
    player.tree_search()
    player.play_move(None)
    player.tree_search()
    assert player.root.position.result() == BLACK
    player.set_result(WHITE, True)
    assert player.extract_data()[2][0] == WHITE and player.result_string == "W+R"


def test_neuralnet_and_selfplay_api():  # src/neural_net.jl:57-73, src/selfplay.jl, mcts_play.jl:126-139
    env = GoEnv(5)
    nn = NeuralNet(env, tower_height=1, seed=0)
    pos = Position(env).play_move((2, 2))
    pi, v = nn([pos, pos.play_move((1, 1))])
    assert pi.shape == (26, 2) and np.allclose(pi.sum(0), 1, atol=1e-5) and abs(v).max() < 1
    p1, v1 = nn(pos)
    assert (p1 == pi[:, 0]).all() and v1 == v[0]
    # an MCTSPlayer whose network is the HIP NeuralNet itself
    player = MCTSPlayer(env, nn, num_readouts=16)
    player.initialize_game()
    for _ in range(4):
        player.tree_search()
    assert player.root.N >= 4 and player.engine.pending_vlosses(0) == 0
    recs = ag.selfplay(env, nn, 16, games=3, seed=1)
    assert [r.game_id for r in recs] == [0, 1, 2]
    for r in recs:
        positions, pis, results = ag.extract_data(env, r)
        assert len(positions) == len(pis) == len(results) == len(r.moves) >= 1
        assert all(z == r.result for z in results)
        assert positions[0].n == 0 and positions[-1].n == len(r.moves) - 1


def test_replay_allgather_over_rccl_from_device_records():
    """SURVEY.md 8e exchange step on the real backend: finished games leave the engine's HBM arena
    as one packed CUDA buffer and go through torch.distributed's nccl (= RCCL) all-gather.  One GPU
    here, so the group has one rank; the collective path is forced."""
    import os
    import torch
    import torch.distributed as dist
    from alphago_jl_amd.distributed import allgather_records, unpack_records

    eng = ag.Engine(board_size=5, tower_height=1, games=4, num_readouts=16, seed=7, record_capacity_games=16)
    eng.init_synthetic(0)
    eng.start(6)
    while eng.records_count() < 6:
        eng.step(8)
    want = eng.records()
    dev = eng.records_packed_device()
    assert dev.is_cuda and (dev.cpu().numpy() == eng.records_packed()).all()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = allgather_records(eng, eng.A, force_collective=True)
    finally:
        dist.destroy_process_group()
    assert [r["game_id"] for r in got] == sorted(r["game_id"] for r in want)
    by_id = {r["game_id"]: r for r in want}
    for r in got:
        w = by_id[r["game_id"]]
        assert (r["moves"] == w["moves"]).all() and r["result"] == w["result"]
        assert (np.nan_to_num(r["pis"]) == np.nan_to_num(w["pis"])).all()
    eng.close()


def test_generation_loop_example():
    """examples/generation_loop.py: self-play -> replay batches on the device -> arena -> BSON checkpoint"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "generation_loop", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "generation_loop.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    positions, eval_games, same = mod.main(["--board", "5", "--tower", "1", "--games", "6", "--readouts", "16",
                                            "--batch-size", "8", "--eval-games", "4"])
    assert positions > 6 and eval_games == 4 and same


def test_initialize_game_from_a_midgame_position_keeps_its_history():
    """initialize_game!(player, pos) keeps pos.board_deltas (board.jl:505-506, mcts_play.jl:110-118): the history
    planes of the first leaves after starting from a mid-game position equal get_feats of the positions the
    reference would build (the history was dropped and the oldest board repeated instead)."""
    env = GoEnv(N)
    pos = Position(env)
    for c in [(2, 2), (6, 6), (2, 6), (6, 2), (4, 4), (3, 3), (5, 5), (1, 1), (7, 7)]:
        pos = pos.play_move(c)
    assert pos.board_deltas.shape[0] == 7
    seen = []

    class Spy(DummyNet):
        def __call__(self, leaves):
            seen.extend(np.asarray(l.feats, np.float32).copy() for l in leaves)
            return super().__call__(leaves)

    player = MCTSPlayer(env, Spy(env))
    player.initialize_game(pos)
    player.tree_search(1)                        # the root itself is the first leaf
    player.tree_search(2)                        # then two children
    want_root = ag.get_feats(pos).transpose(0, 2, 1).reshape(-1)      # [plane, row, col] -> plane-major, p = row + N*col
    assert (seen[0] == want_root).all()
    assert len(seen) >= 2
    # every child leaf: features of pos.play_move(its move); identify the move by the one extra stone of plane pair 1
    legal = pos.all_legal_moves()
    for f in seen[1:]:
        cands = [a for a in range(N * N) if legal[a]]
        match = [a for a in cands
                 if (ag.get_feats(pos.play_move(ag.from_flat(a, env))).transpose(0, 2, 1).reshape(-1) == f).all()]
        assert len(match) == 1, "leaf features are not those of any child position with the reference's history"


def test_node_level_surface_and_suggest_move():
    """The names test/test_mcts.jl:2-5 and test_mcts_player.jl:3-6 import, on the host mirror's node handle (the Julia
    wrapper binds the same entry points: tests/test_abi.py checks its ccalls): select_leaf / maybe_add_child /
    add_virtual_loss / incorporate_results / inject_noise / child_U / child_action_score / set_N as methods of a node,
    and suggest_move / get_position (mcts_play.jl:141-151) on the player."""
    env = GoEnv(N)
    probs = np.ones(env.action_space, np.float32) * 0.02
    probs[17] = 0.4
    probs /= probs.sum()
    player = MCTSPlayer(env, DummyNet(env, fake_priors=probs, fake_value=0.1), num_readouts=24)
    player.initialize_game()
    root = player.root
    leaf = root.select_leaf()                                   # the unexpanded root is its own leaf (mcts.jl:114-118)
    assert leaf == root and not root.is_expanded
    leaf.incorporate_results(probs, 0.1, root)
    assert root.is_expanded and root.N == 1
    # child_action_score = child_Q * to_play + child_U (mcts.jl:86-92), Float64; child_U in the reference's mixed precision
    u = root.child_U
    assert np.allclose(root.child_action_score, root.child_Q.astype(np.float64) * root.position.to_play + u, rtol=0, atol=1e-12)
    assert int(np.argmax(root.child_action_score)) == 17
    child = root.maybe_add_child(17)
    assert root.children[17] == child and child.fmove == 17 and child.position.n == 1
    assert root.maybe_add_child(17) == child                    # idempotent (mcts.jl:141-147)
    w0, rootw0 = root.child_W[17], root.W
    assert w0 == np.float32(0.1)                                # child_W is seeded with the parent's value (mcts.jl:209)
    child.add_virtual_loss(root)                                # W += position.to_play along the path, up_to included (mcts.jl:151-163)
    assert child.losses_applied == 1 and root.losses_applied == 1
    assert root.child_W[17] == w0 + np.float32(WHITE) and root.W == np.float32(rootw0 + BLACK) and root.child_N[17] == 0
    child.revert_virtual_loss(root)
    # (Float32, as the reference: (0.1 - 1) + 1 is 0.100000024)
    assert child.losses_applied == 0 and root.losses_applied == 0
    assert root.child_W[17] == (w0 + np.float32(WHITE)) - np.float32(WHITE)
    assert root.W == np.float32(np.float32(rootw0 + BLACK) - np.float32(BLACK))
    before = root.child_prior.copy()
    root.inject_noise()
    after = root.child_prior
    assert abs(after.sum() - 1.0) < 1e-3 and (after != before).any()
    child.set_N(5.0)
    assert child.N == 5.0 and root.child_N[17] == 5.0
    child.set_N(0.0)
    with pytest.raises(AssertionError):                        # mcts.jl:190: a 10-entry probability vector is a shape error
        child.incorporate_results(np.ones(10, np.float32), 0.0, root)
    # suggest_move: searches until the root has num_readouts more visits, then picks (mcts_play.jl:144-151)
    n0 = player.root.N
    mv = player.suggest_move()
    assert player.root.N >= n0 + 24 and player.engine.pending_vlosses(0) == 0
    assert mv is None or (0 <= mv[0] < N and 0 <= mv[1] < N)
    pos = player.get_position()
    assert pos.n == 0 and pos.to_play == BLACK and (np.asarray(pos.board) == 0).all()


def test_softpick_assertion_matches_the_reference_failure():
    """mcts_play.jl:63-67 on the device: visits on the pass only (or none) below the temperature threshold fail the
    soft pick's assertion -- AGZ_ASSERT_SOFTPICK through the C ABI, AssertionError in the host mirror (the reference's
    @assert) -- and one visited board move makes it succeed; the same cases as tests/test_oracle_player.py."""
    env = GoEnv(N)
    player = MCTSPlayer(env, DummyNet(env), num_readouts=8)
    player.initialize_game()
    player.tree_search()
    e, root = player.engine, player.engine.tree_root(0)
    cn = np.zeros(env.action_space, np.float32)
    e.node_set_floats(0, root, ag._lib.F_CHILD_N, cn)
    assert e.pick_move(0)[0] == ag._lib.ASSERT_SOFTPICK
    cn[-1] = 3
    e.node_set_floats(0, root, ag._lib.F_CHILD_N, cn)
    assert e.pick_move(0)[0] == ag._lib.ASSERT_SOFTPICK
    with pytest.raises(AssertionError):
        player.pick_move()
    cn[17] = 1
    e.node_set_floats(0, root, ag._lib.F_CHILD_N, cn)
    st, a = e.pick_move(0)
    assert st == ag._lib.OK and a == 17
    assert player.pick_move() == ag.from_flat(17, env)


def test_external_network_receives_a_vector_of_positions_like_the_reference():
    """mcts_play.jl:89: `mcts_player.network([leaf.position for leaf in leaves])` -- the duck-typed network gets B
    Position objects (len(positions) = B, test/test_mcts_player.jl:25-32), each with the GoPosition fields of the leaf:
    they equal the positions the host builds by playing the leaf's moves from the root position."""
    env = GoEnv(N)
    pos = Position(env)
    for c in [(2, 2), (6, 6), (2, 6), (6, 2), (4, 4), (3, 3), (5, 5), (1, 1), (7, 7), (0, 1), (1, 0)]:
        pos = pos.play_move(c)
    seen = []

    class Spy(DummyNet):
        def __call__(self, positions):
            assert isinstance(positions, list) and all(isinstance(p, Position) for p in positions)
            seen.append(positions)
            return super().__call__(positions)            # sized by len(positions), like the reference's DummyNet

    player = MCTSPlayer(env, Spy(env))
    player.initialize_game(pos)
    for _ in range(6):
        leaves = player.tree_search(8)
        assert len(leaves) == len(seen[-1]) and all(l == p.node for l, p in zip(leaves, seen[-1]))
    assert sum(len(b) for b in seen) >= 20 and max(len(b) for b in seen) == 8
    root = player.root
    for batch in seen:
        for leaf in batch:
            path, node = [], leaf.node
            while node != root:                              # the leaf's moves, root first
                path.append(node.fmove)
                node = NodeViewParent(node)
            want = pos
            for a in reversed(path):
                want = want.play_move(ag.from_flat(int(a), env))
            assert (leaf.board == want.board).all() and leaf.n == want.n and leaf.to_play == want.to_play
            assert leaf.caps == want.caps and leaf.ko == want.ko and leaf.komi == want.komi
            assert leaf.board_deltas.shape == want.board_deltas.shape and (leaf.board_deltas == want.board_deltas).all()
            assert 1 <= len(leaf.recent) <= 2 and leaf.recent == want.recent[-len(leaf.recent):]     # the last move(s) the tree holds
            assert (ag.get_feats(leaf) == ag.get_feats(want)).all()


def NodeViewParent(node):
    return ag.api.NodeView(node._p, node._info.parent)


def test_a_wrapped_neuralnet_as_external_network_builds_the_same_tree_as_the_resident_one():
    """the Vector{Position} seam end to end: MCTSPlayer(env, nn) evaluates leaves on the device; MCTSPlayer(env, f) with
    f = positions -> nn(positions) goes leaf positions -> host -> agz_net_forward -> host -> incorporate.  Same tree."""
    env = GoEnv(5)
    nn = NeuralNet(env, tower_height=1, seed=3)
    calls = []

    def wrapped(positions):
        calls.append(len(positions))
        return nn(positions)

    a = MCTSPlayer(env, nn, num_readouts=32)
    b = MCTSPlayer(env, wrapped, num_readouts=32)
    for p in (a, b):
        p.initialize_game()
    for _ in range(10):
        la, lb = a.tree_search(), b.tree_search()
        assert [x.id for x in la] == [x.id for x in lb]
    assert calls and (a.root.child_N == b.root.child_N).all() and (a.root.child_W == b.root.child_W).all()
    assert a.root.N == b.root.N and a.pick_move() == b.pick_move()


def test_tracker_style_network_results_are_unwrapped():
    """mcts_play.jl:90: `move_probs, values = move_probs.data, values.data` -- a network may answer with objects carrying
    `.data` (the reference's DummyNet returns `param(...)`, test/test_mcts_player.jl:22-32)"""
    env = GoEnv(N)

    class Tracked:
        def __init__(self, data):
            self.data = data

    class TrackedNet(DummyNet):
        def __call__(self, positions):
            p, v = super().__call__(positions)
            return Tracked(p), Tracked(v)

    player = MCTSPlayer(env, TrackedNet(env, fake_value=0.17))
    player.initialize_game()
    player.tree_search(4)
    assert player.root.N == 1 and player.root.Q == pytest.approx(0.085)

    class WrongShape(DummyNet):                              # a network sized by anything but len(positions) fails loudly
        def __call__(self, positions):
            return np.ones((env.action_space, 3)) / env.action_space, np.zeros(3)

    bad = MCTSPlayer(env, WrongShape(env))
    bad.initialize_game()
    with pytest.raises(AssertionError):
        bad.tree_search(1)
