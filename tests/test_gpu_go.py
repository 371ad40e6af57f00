"""HIP Go-rule kernels (through the C ABI) against the oracle: legality, play / capture / ko and
Tromp-Taylor scoring must agree exactly on random games and on the reference's fixtures."""
import ctypes as C

import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from orc import load_board
from test_hostsim_go import random_positions
from test_oracle_go import ALMOST_DONE, LEGAL_BOARD, TT_FTW

pytestmark = pytest.mark.gpu
L = orc.lib()


@pytest.mark.parametrize("N,games,moves", [(5, 60, 40), (9, 40, 140), (19, 6, 420)])
def test_rules_match_oracle_on_random_games(N, games, moves):
    P, A = N * N, N * N + 1
    positions = random_positions(N, games, moves, seed=100 + N)
    eng = ag.Engine(board_size=N, games=1, tower_height=0, num_readouts=8, max_nodes_per_game=16)
    B = len(positions)
    boards = np.stack([p.board_np() for p in positions])
    tp = np.array([p.to_play for p in positions], np.int8)
    ko = np.array([p.ko for p in positions], np.int32)
    legal = eng.go_legal(boards, tp, ko)
    score = eng.go_score(boards, np.full(B, 7.5, np.float32))
    rng = np.random.RandomState(1)
    moves_ = rng.randint(0, A, size=B).astype(np.int32)
    bo, ko_o, nc, st = eng.go_play(boards, tp, ko, moves_)
    n_illegal = n_capture = 0
    for b, pos in enumerate(positions):
        assert (legal[b] == orc.legal_moves(pos)).all(), b
        assert score[b] == L.or_score(C.byref(pos)), b
        rcode, nxt = orc.play(pos, int(moves_[b]))
        if rcode != orc.OK:
            assert st[b] == 1 and (bo[b] == boards[b]).all()
            n_illegal += 1
            continue
        assert st[b] == 0
        assert (bo[b] == nxt.board_np()).all(), b
        assert ko_o[b] == nxt.ko, b
        caps = (nxt.caps[0] - pos.caps[0]) + (nxt.caps[1] - pos.caps[1])
        assert nc[b] == caps, b
        n_capture += caps > 0
    assert n_illegal > 0 and n_capture > 0
    eng.close()


def test_reference_fixtures():
    N = 9
    eng = ag.Engine(board_size=N, games=1, tower_height=0, num_readouts=8, max_nodes_per_game=16)
    board = load_board(LEGAL_BOARD, N)
    for b, tp in ((board, 1), (-board, -1)):       # test_go.jl:338-378
        pos = orc.make_pos(N, board=b, to_play=tp)
        got = eng.go_legal(b[None], [tp], [-1])[0]
        assert (got == orc.legal_moves(pos)).all() and int(got.sum()) == 45
    ad, tt = load_board(ALMOST_DONE, N), load_board(TT_FTW, N)
    assert list(eng.go_score(np.stack([ad, ad, tt]), [2.5, 0.5, 2.5])) == [-0.5, 1.5, -5.5]
    sb = load_board(".OX......\nOX.......\n" + ("." * 9 + "\n") * 7, N)   # test_go.jl:461-507
    bo, ko_o, nc, st = eng.go_play(sb[None], [1], [-1], [orc.from_kgs("A9", N)])
    assert st[0] == 0 and nc[0] == 1 and ko_o[0] == orc.from_kgs("B9", N)
    assert eng.go_play(bo, [-1], ko_o, [orc.from_kgs("B9", N)])[3][0] == 1
    eng.close()
