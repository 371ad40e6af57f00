"""Pins the oracle's Go rules against every known answer in the reference's
test/test_go.jl (file:line cited per test).  CPU only."""
import ctypes as C

import numpy as np
import pytest

import orc
from orc import BLACK, WHITE, EMPTY, from_kgs, load_board, make_pos, pc_set

N = 9
EMPTY_ROW = "." * N + "\n"
TEST_BOARD = load_board(".X.....OO\nX........\n" + EMPTY_ROW * 7, N)


def i8(a):
    return np.ascontiguousarray(a, dtype=np.int8).ctypes.data_as(C.POINTER(C.c_int8))


def group(board, p):
    stones = np.zeros(N * N, np.int8)
    libs = np.zeros(N * N, np.int8)
    nl = orc.lib().or_group_info(N, i8(board), p, i8(stones), i8(libs))
    # i8() of a fresh contiguous array aliases it, so the outputs were written in place
    return set(np.nonzero(stones)[0].tolist()), set(np.nonzero(libs)[0].tolist()), nl


def group_arrays(board, p):
    stones = np.zeros(N * N, np.int8)
    libs = np.zeros(N * N, np.int8)
    sp = stones.ctypes.data_as(C.POINTER(C.c_int8))
    lp = libs.ctypes.data_as(C.POINTER(C.c_int8))
    b = np.ascontiguousarray(board, np.int8)
    nl = orc.lib().or_group_info(N, b.ctypes.data_as(C.POINTER(C.c_int8)), p, sp, lp)
    return set(np.nonzero(stones)[0].tolist()), set(np.nonzero(libs)[0].tolist()), nl


def add_stone(board, color, p):
    """add_stone! on a bare board (board.jl:227-269): place, capture, return (board, captured)."""
    pos = make_pos(N, board=board, to_play=color)
    out = orc.OPos()
    rcode = orc.lib().or_play_move_color(C.byref(pos), p, color, C.byref(out))
    assert rcode == orc.OK
    nb = out.board_np()
    captured = {q for q in range(N * N) if board[q] == -color and nb[q] == EMPTY}
    return nb, captured


def ngroups(board):
    b = np.ascontiguousarray(board, np.int8)
    return orc.lib().or_count_groups(N, b.ctypes.data_as(C.POINTER(C.c_int8)))


def test_parsing():  # test_go.jl:24-30
    assert from_kgs("A9", N) == orc.rc(1, 1, N)
    assert orc.from_sgf("aa", N) == orc.rc(1, 1, N)
    assert from_kgs("A3", N) == orc.rc(7, 1, N)
    assert orc.from_sgf("ac", N) == orc.rc(3, 1, N)
    assert from_kgs("D4", N) == orc.from_sgf("df", N)
    # derived in SURVEY.md 8c: D9 <-> coord (1,4) <-> 1-based flat 28
    assert from_kgs("D9", N) == 27


def test_is_koish():  # test_go.jl:42-47
    L = orc.lib()
    assert L.or_is_koish(N, i8(TEST_BOARD), from_kgs("A9", N)) == BLACK
    assert L.or_is_koish(N, i8(TEST_BOARD), from_kgs("B8", N)) == 0
    assert L.or_is_koish(N, i8(TEST_BOARD), from_kgs("B9", N)) == 0
    assert L.or_is_koish(N, i8(TEST_BOARD), from_kgs("E5", N)) == 0


def test_is_eyeish():  # test_go.jl:49-73
    board = load_board("""
              .XX...XXX
              X.X...X.X
              XX.....X.
              ........X
              XXXX.....
              OOOX....O
              X.OXX.OO.
              .XO.X.O.O
              XXO.X.OO.
          """, N)
    L = orc.lib()
    for p in pc_set("A2 A9 B8 J7 H8", N):
        assert L.or_is_eyeish(N, i8(board), p) == BLACK
    for p in pc_set("H2 J1 J3", N):
        assert L.or_is_eyeish(N, i8(board), p) == WHITE
    for p in pc_set("B3 E5", N):
        assert L.or_is_eyeish(N, i8(board), p) == 0


def test_lib_tracker_init():  # test_go.jl:74-86
    board = load_board("X........" + EMPTY_ROW * 8, N)
    assert ngroups(board) == 1
    stones, libs, nl = group_arrays(board, from_kgs("A9", N))
    assert nl == 2 and stones == pc_set("A9", N) and libs == pc_set("B9 A8", N)


def test_place_stone():  # test_go.jl:88-101
    board = load_board("X........" + EMPTY_ROW * 8, N)
    nb, cap = add_stone(board, BLACK, from_kgs("B9", N))
    assert ngroups(nb) == 1 and cap == set()
    for s in ("A9", "B9"):
        stones, libs, nl = group_arrays(nb, from_kgs(s, N))
        assert nl == 3
    assert stones == pc_set("A9 B9", N) and libs == pc_set("C9 A8 B8", N)


def test_place_stone_opposite_color():  # test_go.jl:103-122
    board = load_board("X........" + EMPTY_ROW * 8, N)
    nb, cap = add_stone(board, WHITE, from_kgs("B9", N))
    assert ngroups(nb) == 2
    bs, bl, bn = group_arrays(nb, from_kgs("A9", N))
    ws, wl, wn = group_arrays(nb, from_kgs("B9", N))
    assert bn == 1 and wn == 2
    assert bs == pc_set("A9", N) and bl == pc_set("A8", N)
    assert ws == pc_set("B9", N) and wl == pc_set("C9 B8", N)


def test_merge_multiple_groups():  # test_go.jl:124-143
    board = load_board(".X.......\nX.X......\n.X.......\n" + EMPTY_ROW * 6, N)
    nb, cap = add_stone(board, BLACK, from_kgs("B8", N))
    assert ngroups(nb) == 1
    stones, libs, nl = group_arrays(nb, from_kgs("B8", N))
    assert stones == pc_set("B9 A8 B8 C8 B7", N)
    assert libs == pc_set("A9 C9 D8 A7 C7 B6", N)
    assert nl == 6


def test_capture_stone():  # test_go.jl:145-156
    board = load_board(".X.......\nXO.......\n.X.......\n" + EMPTY_ROW * 6, N)
    nb, cap = add_stone(board, BLACK, from_kgs("C8", N))
    assert ngroups(nb) == 4
    assert nb[from_kgs("B8", N)] == EMPTY
    assert cap == pc_set("B8", N)


def test_capture_many():  # test_go.jl:158-204
    board = load_board(".XX......\nXOO......\n.XX......\n" + EMPTY_ROW * 6, N)
    nb, cap = add_stone(board, BLACK, from_kgs("D8", N))
    assert ngroups(nb) == 4
    assert cap == pc_set("B8 C8", N)
    s, l, n_ = group_arrays(nb, from_kgs("A8", N))
    assert s == pc_set("A8", N) and l == pc_set("A9 B8 A7", N) and n_ == 3
    s, l, n_ = group_arrays(nb, from_kgs("D8", N))
    assert s == pc_set("D8", N) and l == pc_set("D9 C8 E8 D7", N) and n_ == 4
    s, l, n_ = group_arrays(nb, from_kgs("B9", N))
    assert s == pc_set("B9 C9", N) and l == pc_set("A9 D9 B8 C8", N) and n_ == 4
    s, l, n_ = group_arrays(nb, from_kgs("B7", N))
    assert s == pc_set("B7 C7", N) and l == pc_set("B8 C8 A7 D7 B6 C6", N) and n_ == 6


def test_capture_multiple_groups():  # test_go.jl:206-232
    board = load_board(".OX......\nOXX......\nXX.......\n" + EMPTY_ROW * 6, N)
    nb, cap = add_stone(board, BLACK, from_kgs("A9", N))
    assert ngroups(nb) == 2
    assert cap == pc_set("B9 A8", N)
    s, l, n_ = group_arrays(nb, from_kgs("A9", N))
    assert s == pc_set("A9", N) and l == pc_set("B9 A8", N) and n_ == 2
    s, l, n_ = group_arrays(nb, from_kgs("C9", N))
    assert s == pc_set("C9 B8 C8 A7 B7", N)
    assert l == pc_set("B9 D9 A8 D8 C7 A6 B6", N) and n_ == 7


def test_same_friendly_group_neighboring_twice():  # test_go.jl:234-247
    board = load_board("XX.......\nX........\n" + EMPTY_ROW * 7, N)
    nb, cap = add_stone(board, BLACK, from_kgs("B8", N))
    assert ngroups(nb) == 1 and cap == set()
    s, l, _ = group_arrays(nb, from_kgs("A9", N))
    assert s == pc_set("A9 B9 A8 B8", N) and l == pc_set("C9 C8 A7 B7", N)


def test_same_opponent_group_neighboring_twice():  # test_go.jl:249-266
    board = load_board("XX.......\nX........\n" + EMPTY_ROW * 7, N)
    nb, cap = add_stone(board, WHITE, from_kgs("B8", N))
    assert ngroups(nb) == 2 and cap == set()
    s, l, _ = group_arrays(nb, from_kgs("A9", N))
    assert s == pc_set("A9 B9 A8", N) and l == pc_set("C9 A7", N)
    s, l, _ = group_arrays(nb, from_kgs("B8", N))
    assert s == pc_set("B8", N) and l == pc_set("C8 B7", N)


def pos_equal(a, b, check_recent=True):
    """test_utils.jl:62-74 (the liberty tracker is derived from the board)"""
    assert (a.board_np() == b.board_np()).all()
    assert a.n == b.n
    assert tuple(a.caps) == tuple(b.caps)
    assert a.ko == b.ko
    r = min(a.recent_len, b.recent_len)
    if check_recent and r > 0:
        ra = [(a.recent_color[k], a.recent_move[k]) for k in range(a.recent_len - r, a.recent_len)]
        rb = [(b.recent_color[k], b.recent_move[k]) for k in range(b.recent_len - r, b.recent_len)]
        assert ra == rb
    assert a.to_play == b.to_play


def test_passing():  # test_go.jl:264-285
    start = make_pos(N, board=TEST_BOARD, n=0, komi=6.5, caps=(1, 2), ko=from_kgs("A1", N), to_play=BLACK)
    expected = make_pos(N, board=TEST_BOARD, n=1, komi=6.5, caps=(1, 2), ko=-1,
                        recent=[(BLACK, N * N)], to_play=WHITE)
    out = orc.OPos()
    orc.lib().or_pass_move(C.byref(start), C.byref(out))
    pos_equal(out, expected)


def test_flipturn():  # test_go.jl:287-308
    start = make_pos(N, board=TEST_BOARD, n=0, komi=6.5, caps=(1, 2), ko=from_kgs("A1", N), to_play=BLACK)
    expected = make_pos(N, board=TEST_BOARD, n=0, komi=6.5, caps=(1, 2), ko=-1, to_play=WHITE)
    out = orc.OPos()
    orc.lib().or_flip_playerturn(C.byref(start), C.byref(out))
    pos_equal(out, expected)


def test_is_move_suicidal():  # test_go.jl:310-336
    board = load_board("""
        ...O.O...
        ....O....
        XO.....O.
        OXO...OXO
        O.XO.OX.O
        OXO...OOX
        XO.......
        ......XXO
        .....XOO.
    """, N)
    pos = make_pos(N, board=board, to_play=BLACK)
    L = orc.lib()
    for p in pc_set("E9 H5", N):
        assert board[p] == EMPTY
        assert L.or_is_move_suicidal(C.byref(pos), p) == 1
    for p in pc_set("B5 J1 A9", N):
        assert board[p] == EMPTY
        assert L.or_is_move_suicidal(C.byref(pos), p) == 0


LEGAL_BOARD = """
        .O.O.XOX.
        O..OOOOOX
        ......O.O
        OO.....OX
        XO.....X.
        .O.......
        OX.....OO
        XX...OOOX
        .....O.X.
    """


def test_legal_moves():  # test_go.jl:338-378
    board = load_board(LEGAL_BOARD, N)
    L = orc.lib()
    for b, tp in ((board, BLACK), (-board, WHITE)):
        pos = make_pos(N, board=b, to_play=tp)
        for p in pc_set("A9 E9 J9", N):
            assert L.or_is_move_legal(C.byref(pos), p) == 0
        for p in pc_set("A4 G1 J1 H7", N):
            assert L.or_is_move_legal(C.byref(pos), p) == 1
        bulk = orc.legal_moves(pos)
        for a in range(N * N + 1):
            assert L.or_is_move_legal(C.byref(pos), a) == bulk[a]
        # SURVEY.md 8c probe: 45 legal entries including the pass
        assert int(bulk.sum()) == 45


def test_move():  # test_go.jl:380-424
    start = make_pos(N, board=TEST_BOARD, n=0, komi=6.5, caps=(1, 2), to_play=BLACK)
    eb = load_board(".XX....OO\nX........\n" + EMPTY_ROW * 7, N)
    expected = make_pos(N, board=eb, n=1, komi=6.5, caps=(1, 2),
                        recent=[(BLACK, from_kgs("C9", N))], to_play=WHITE)
    rcode, actual = orc.play(start, from_kgs("C9", N))
    assert rcode == orc.OK
    pos_equal(actual, expected)
    eb2 = load_board(".XX....OO\nX.......O\n" + EMPTY_ROW * 7, N)
    expected2 = make_pos(N, board=eb2, n=2, komi=6.5, caps=(1, 2),
                         recent=[(BLACK, from_kgs("C9", N)), (WHITE, from_kgs("J8", N))], to_play=BLACK)
    rcode, actual2 = orc.play(actual, from_kgs("J8", N))
    assert rcode == orc.OK
    pos_equal(actual2, expected2)


def test_move_with_capture():  # test_go.jl:426-459
    sb = load_board(EMPTY_ROW * 5 + "XXXX.....\nXOOX.....\nO.OX.....\nOOXX.....\n", N)
    start = make_pos(N, board=sb, n=0, komi=6.5, caps=(1, 2), to_play=BLACK)
    eb = load_board(EMPTY_ROW * 5 + "XXXX.....\nX..X.....\n.X.X.....\n..XX.....\n", N)
    expected = make_pos(N, board=eb, n=1, komi=6.5, caps=(7, 2),
                        recent=[(BLACK, from_kgs("B2", N))], to_play=WHITE)
    rcode, actual = orc.play(start, from_kgs("B2", N))
    assert rcode == orc.OK
    pos_equal(actual, expected)


def test_ko_move():  # test_go.jl:461-507
    sb = load_board(".OX......\nOX.......\n" + EMPTY_ROW * 7, N)
    start = make_pos(N, board=sb, n=0, komi=6.5, caps=(1, 2), to_play=BLACK)
    eb = load_board("X.X......\nOX.......\n" + EMPTY_ROW * 7, N)
    expected = make_pos(N, board=eb, n=1, komi=6.5, caps=(2, 2), ko=from_kgs("B9", N),
                        recent=[(BLACK, from_kgs("A9", N))], to_play=WHITE)
    rcode, actual = orc.play(start, from_kgs("A9", N))
    assert rcode == orc.OK
    pos_equal(actual, expected)
    # retaking the ko is illegal until two intervening moves
    rcode, _ = orc.play(actual, from_kgs("B9", N))
    assert rcode == orc.ILLEGAL_MOVE
    _, p1 = orc.play(actual, N * N)
    _, p2 = orc.play(p1, N * N)
    rcode, retake = orc.play(p2, from_kgs("B9", N))
    assert rcode == orc.OK
    expected = make_pos(N, board=sb, n=4, komi=6.5, caps=(2, 3), ko=from_kgs("A9", N),
                        recent=[(BLACK, from_kgs("A9", N)), (WHITE, N * N), (BLACK, N * N),
                                (WHITE, from_kgs("B9", N))], to_play=BLACK)
    pos_equal(retake, expected)


def test_is_game_over():  # test_go.jl:509-516
    root = make_pos(N)
    assert not root.done
    _, first = orc.play(root, N * N)
    assert not first.done
    _, second = orc.play(first, N * N)
    assert second.done


def test_scoring():  # test_go.jl:518-564
    board = load_board("""
        .XX......
        OOXX.....
        OOOX...X.
        OXX......
        OOXXXXXX.
        OOOXOXOXX
        .O.OOXOOX
        .O.O.OOXX
        ......OOO
    """, N)
    pos = make_pos(N, board=board, n=54, komi=6.5, caps=(2, 5), to_play=BLACK)
    assert orc.lib().or_score(C.byref(pos)) == 1.5
    board = load_board("""
        XXX......
        OOXX.....
        OOOX...X.
        OXX......
        OOXXXXXX.
        OOOXOXOXX
        .O.OOXOOX
        .O.O.OOXX
        ......OOO
      """, N)
    pos = make_pos(N, board=board, n=55, komi=6.5, caps=(2, 5), to_play=WHITE)
    assert orc.lib().or_score(C.byref(pos)) == 2.5


ALMOST_DONE = """
    .XO.XO.OO
    X.XXOOOO.
    XXXXXOOOO
    XXXXXOOOO
    .XXXXOOO.
    XXXXXOOOO
    .XXXXOOO.
    XXXXXOOOO
    XXXXOOOOO
"""
TT_FTW = """
    .XXOOOOOO
    X.XOO...O
    .XXOO...O
    X.XOO...O
    .XXOO..OO
    X.XOOOOOO
    .XXOOOOOO
    X.XXXXXXX
    XXXXXXXXX
"""


def test_fixture_scores_and_legal_sets():
    """test_mcts_player.jl:143 (score == -0.5) and the derived vectors of SURVEY.md 8c."""
    b = load_board(ALMOST_DONE, N)
    pos = make_pos(N, board=b, komi=2.5, to_play=BLACK)
    assert orc.lib().or_score(C.byref(pos)) == -0.5
    pos05 = make_pos(N, board=b, komi=0.5, to_play=BLACK)
    assert orc.lib().or_score(C.byref(pos05)) == 1.5
    tt = make_pos(N, board=load_board(TT_FTW, N), komi=2.5)
    assert orc.lib().or_score(C.byref(tt)) == -5.5
    legal_b = (np.nonzero(orc.legal_moves(pos))[0] + 1).tolist()
    assert legal_b == [1, 5, 7, 11, 28, 82]
    posw = make_pos(N, board=b, komi=2.5, to_play=WHITE)
    legal_w = (np.nonzero(orc.legal_moves(posw))[0] + 1).tolist()
    assert legal_w == [28, 55, 74, 77, 79, 82]


def test_result_strings():  # board.jl:546-555
    pos = make_pos(N)
    s = C.create_string_buffer(16)
    orc.lib().or_result_string(C.byref(pos), s)
    assert s.value == b"W+7.5"
    assert orc.lib().or_result(C.byref(pos)) == -1
