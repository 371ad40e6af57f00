"""The body of the reference's train() loop (/root/reference/src/train.jl:47-74) transliterated LINE FOR LINE through the
host mirror (alphago.jl_amd/api.py) -- the drop-in claim of BASELINE.json's north_star ("keeping the MCTSPlayer / NeuralNet /
GoEnv API surface so it drops into train()/selfplay"):

    player  = selfplay(env, cur_nn, readouts)                    train.jl:57   ONE player-like object, no `games`
    p, π, v = extract_data(player)                               train.jl:58   ONE argument
    push_data / shrink on the three buffers                      train.jl:60-65
    get_replay_batch(pos_buffer, π_buffer, res_buffer; ...)      train.jl:68
    _train(cur_nn, (replay_pos, replay_π, replay_res), opt)      train.jl:70
    player.result_string, player.root.position.n                 train.jl:71-72

for 3 games at 5x5 / tower 1 / 16 readouts.  Every game's record must be the ORACLE's game on the weights the loop held at
that moment (moves, π and q bit for bit, result, result string), i.e. the training step between games feeds the next
selfplay like it does in the reference (cur_nn is updated in place)."""
import numpy as np
import pytest

import alphago_jl_amd as ag
from alphago_jl_amd import GoEnv, Momentum, NeuralNet, Position, _train, extract_data, get_replay_batch, selfplay
from gpu_common import GpuNetForOracle
from test_hostsim_selfplay import bits_equal, oracle_game

pytestmark = pytest.mark.gpu
N, TOWER, R, SEED = 5, 1, 16, 3


def test_train_loop_body_runs_unchanged_and_every_game_is_the_oracles():
    env = GoEnv(N)
    num_games, memory_size, batch_size, epochs, readouts, start_training_after = 3, 40, 8, 1, R, 8
    ag.seed(SEED)                                               # Random.seed!(SEED)
    rng = np.random.default_rng(0)

    cur_nn = NeuralNet(env, tower_height=TOWER)                 # train.jl:43
    fwd = GpuNetForOracle(cur_nn.engine)                        # the oracle's tree search on cur_nn's forward

    pos_buffer = []                                             # train.jl:47-49
    pi_buffer = []
    res_buffer = []

    push_data = lambda vec, data: vec + list(data)              # train.jl:51  vcat(vec, data)
    shrink = lambda vec: vec[len(vec) - memory_size:]           # train.jl:52  vec[end-memory_size+1:end]

    opt = Momentum(2e-2)                                        # train.jl:54
    losses, trained = [], 0

    for i in range(1, num_games + 1):                           # train.jl:56
        want = oracle_game(N, fwd, readouts, SEED, i - 1)       # (the checker: the oracle on the weights of this moment)

        player = selfplay(env, cur_nn, readouts)                # train.jl:57
        p, pi, v = extract_data(player)                         # train.jl:58

        pos_buffer, pi_buffer, res_buffer = (push_data(b, d) for b, d in
                                             zip((pos_buffer, pi_buffer, res_buffer), (p, pi, v)))      # train.jl:60-61

        if len(pos_buffer) > memory_size:                       # train.jl:63-65
            pos_buffer, pi_buffer, res_buffer = (shrink(b) for b in (pos_buffer, pi_buffer, res_buffer))

        # ---- parity of what the loop just consumed
        n = want["num_moves"]
        assert player.game_id == i - 1 and player.short_searches == 0
        assert player.root.position.n == n == len(p) == len(pi) == len(v) == len(player.searches_pi) == len(player.qs)
        assert [ag.to_flat(c, env) for c in player.moves] == list(want["moves"][:n])
        assert bits_equal(np.stack(pi), want["pis"]) and bits_equal(np.asarray(player.qs), want["qs"])
        assert player.result == want["result"] and all(z == want["result"] for z in v)
        assert player.result_string == want["result_string"].decode()
        assert all(isinstance(q, Position) for q in p) and [q.n for q in p] == list(range(n))
        assert [ag.to_flat(m.move, env) for m in player.root.position.recent] == list(want["moves"][:n])

        if len(pos_buffer) >= start_training_after:             # train.jl:67
            replay_pos, replay_pi, replay_res = get_replay_batch(pos_buffer, pi_buffer, res_buffer,
                                                                 batch_size=batch_size, rng=rng)         # train.jl:68-69
            loss = _train(cur_nn, (replay_pos, replay_pi, replay_res), opt, epochs=epochs)               # train.jl:70
            result = player.result_string                       # train.jl:71
            num_moves = player.root.position.n                  # train.jl:72
            print(f"Episode {i} over. Loss: {loss}. Winner: {result}. Moves: {num_moves}.")              # train.jl:73
            assert replay_pi.shape == (env.action_space, batch_size) and len(replay_pos) == len(replay_res) == batch_size
            assert np.isfinite(loss) and isinstance(result, str) and num_moves == n
            losses.append(loss)
            trained += 1

    assert trained >= 2, "the loop trained between games: later games ran on updated weights"
    assert len(pos_buffer) == len(pi_buffer) == len(res_buffer) <= memory_size


def test_selfplay_with_games_returns_the_same_player_objects_and_successive_calls_continue_the_stream():
    env = GoEnv(N)
    nn = NeuralNet(env, tower_height=TOWER)
    ag.seed(7)
    a = selfplay(env, nn, R)
    b = selfplay(env, nn, R)
    ag.seed(7)
    both = selfplay(env, nn, R, games=2)
    assert isinstance(both, list) and [q.game_id for q in both] == [0, 1] == [a.game_id, b.game_id]
    for one, many in zip((a, b), both):
        assert one.moves == many.moves and one.result_string == many.result_string
        assert bits_equal(np.stack(one.searches_pi), np.stack(many.searches_pi))
        pos, pis, res = extract_data(many)
        assert len(pos) == len(pis) == len(res) == many.root.position.n
