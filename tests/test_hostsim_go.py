"""Engine rules logic (alphago.jl_amd/csrc/agz_search.h, run lane-serially by the host wave
simulator) against the oracle: legality, play/capture/ko and area scoring must agree exactly on
random games and on the reference's own fixtures.  CPU only; the same comparisons run against
the real HIP kernels in tests/test_gpu_go.py."""
import ctypes as C

import numpy as np
import pytest

import hs
import orc
from orc import load_board
from test_oracle_go import ALMOST_DONE, LEGAL_BOARD, TT_FTW

L = orc.lib()


def random_positions(N, games, max_moves, seed):
    """positions reached by uniformly random legal play (passes allowed but rare)"""
    rng = np.random.RandomState(seed)
    out = []
    P = N * N
    for _ in range(games):
        pos = orc.make_pos(N)
        for _ in range(max_moves):
            legal = np.nonzero(orc.legal_moves(pos))[0]
            pts = legal[legal < P]
            if len(pts) == 0 or rng.rand() < 0.02:
                a = P
            else:
                a = int(rng.choice(pts))
            rcode, pos = orc.play(pos, a)
            assert rcode == orc.OK
            out.append(pos.copy())
            if pos.done:
                break
    return out


@pytest.mark.parametrize("N,games,moves", [(5, 30, 40), (9, 12, 140), (19, 2, 420)])
def test_rules_match_oracle_on_random_games(N, games, moves):
    P, A = N * N, N * N + 1
    positions = random_positions(N, games, moves, seed=N)
    sim = hs.Sim(board_size=N, games=1, num_readouts=8, max_nodes_per_game=16)
    B = len(positions)
    boards = np.stack([p.board_np() for p in positions])
    tp = np.array([p.to_play for p in positions], np.int8)
    ko = np.array([p.ko for p in positions], np.int32)
    komi = np.full(B, 7.5, np.float32)
    legal = sim.go_legal(boards, tp, ko)
    score = sim.go_score(boards, komi)
    rng = np.random.RandomState(1)
    moves_ = rng.randint(0, A, size=B).astype(np.int32)
    bo, ko_o, nc, st = sim.go_play(boards, tp, ko, moves_)
    n_illegal = n_capture = n_ko = 0
    for b, pos in enumerate(positions):
        assert (legal[b] == orc.legal_moves(pos)).all(), b
        assert score[b] == L.or_score(C.byref(pos)), b
        rcode, nxt = orc.play(pos, int(moves_[b]))
        if rcode != orc.OK:
            assert st[b] == 1
            assert (bo[b] == boards[b]).all()
            n_illegal += 1
            continue
        assert st[b] == 0
        assert (bo[b] == nxt.board_np()).all(), b
        assert ko_o[b] == nxt.ko, b
        caps = (nxt.caps[0] - pos.caps[0]) + (nxt.caps[1] - pos.caps[1])
        assert nc[b] == caps, b
        n_capture += caps > 0
        n_ko += nxt.ko >= 0
    assert n_illegal > 0 and n_capture > 0
    sim.close()


def test_reference_fixtures():
    """test_go.jl:338-378 legality fixture (both colours), scoring fixtures, ko sequence"""
    N = 9
    sim = hs.Sim(board_size=N, games=1, num_readouts=8, max_nodes_per_game=16)
    board = load_board(LEGAL_BOARD, N)
    for b, tp in ((board, 1), (-board, -1)):
        pos = orc.make_pos(N, board=b, to_play=tp)
        got = sim.go_legal(b[None], [tp], [-1])[0]
        assert (got == orc.legal_moves(pos)).all()
        assert int(got.sum()) == 45
        for s in ("A9", "E9", "J9"):
            assert got[orc.from_kgs(s, N)] == 0
        for s in ("A4", "G1", "J1", "H7"):
            assert got[orc.from_kgs(s, N)] == 1
    ad = load_board(ALMOST_DONE, N)
    tt = load_board(TT_FTW, N)
    sc = sim.go_score(np.stack([ad, ad, tt]), [2.5, 0.5, 2.5])
    assert list(sc) == [-0.5, 1.5, -5.5]
    # ko (test_go.jl:461-507): capture at A9 sets ko at B9; retake is illegal
    sb = load_board(".OX......\nOX.......\n" + ("." * 9 + "\n") * 7, N)
    bo, ko_o, nc, st = sim.go_play(sb[None], [1], [-1], [orc.from_kgs("A9", N)])
    assert st[0] == 0 and nc[0] == 1 and ko_o[0] == orc.from_kgs("B9", N)
    bo2, ko2, nc2, st2 = sim.go_play(bo, [-1], ko_o, [orc.from_kgs("B9", N)])
    assert st2[0] == 1
    # 6-stone capture (test_go.jl:426-459)
    cb = load_board(("." * 9 + "\n") * 5 + "XXXX.....\nXOOX.....\nO.OX.....\nOOXX.....\n", N)
    bo, ko_o, nc, st = sim.go_play(cb[None], [1], [-1], [orc.from_kgs("B2", N)])
    assert st[0] == 0 and nc[0] == 6 and ko_o[0] == -1
    sim.close()
