"""The Go board, the ladder reader and the V1 feature planes (scope row f-3) through the C ABI, pinned to the reference's own
tests: src/libdg_go/board.rs:280-411 (capture, capture_group, suicide_corner, suicide_middle, ko,
double_liberty_subtraction, black_starts, alternate_turns), src/libdg_go/board_fast.rs:547-562 (check_get_n_liberty_if),
src/libdg_go/utils/ladder.rs:187-350 (all six).  The feature planes have no expected vectors in the reference; they are
checked plane by plane against their definition (features.rs:99-251) on positions small enough to reason about."""
import os

import numpy as np

B, W = 0, 1


def test_capture(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    b.place(B, 9, 9)
    for x, y in ((8, 9), (10, 9), (9, 8), (9, 10)):
        b.place(W, x, y)
    assert b.at(9, 9) is None


def test_capture_group(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    for x, y in ((0, 1), (1, 0), (0, 0), (1, 1)):
        b.place(B, x, y)
    for x, y in ((2, 0), (2, 1), (0, 2), (1, 2)):
        b.place(W, x, y)
    assert all(b.at(x, y) is None for x, y in ((0, 0), (0, 1), (1, 0), (1, 1)))


def test_suicide_corner(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    b.place(W, 0, 0); b.place(B, 1, 0); b.place(B, 0, 1)
    assert b.at(0, 0) is None
    assert not b.is_valid(W, 0, 0) and b.is_valid(B, 0, 0)


def test_suicide_middle(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    b.place(B, 9, 9)
    for x, y in ((8, 9), (10, 9), (9, 8), (9, 10)):
        b.place(W, x, y)
    assert b.at(9, 9) is None and not b.is_valid(B, 9, 9) and b.is_valid(W, 9, 9)


def test_ko(api):
    b = api.Board(7.5)
    b.place(B, 0, 0); b.place(B, 0, 2); b.place(B, 1, 1); b.place(W, 1, 0); b.place(W, 0, 1)
    assert not b.is_valid(B, 0, 0)


def test_double_liberty_subtraction(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    for x, y in ((1, 1), (1, 2), (2, 1), (0, 2), (2, 0)):
        b.place(B, x, y)
    for x, y in ((0, 3), (3, 0), (1, 3), (3, 1), (2, 2)):
        b.place(W, x, y)
    assert b.is_valid(W, 0, 1) and b.is_valid(W, 1, 0)
    b.place(W, 0, 1)
    assert b.at(0, 1) == W
    assert all(b.at(x, y) == B for x, y in ((1, 1), (1, 2), (2, 1), (0, 2), (2, 0)))


def test_turns(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(0.5)
    assert b.to_move() == B
    b.place(B, 0, 0); assert b.to_move() == W
    b.place(W, 1, 1); assert b.to_move() == B
    b.place(W, 2, 2); assert b.to_move() == B            # "if black pass, it should not mess everything up"


def test_check_get_n_liberty_if(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    b.place(B, 0, 1); b.place(B, 1, 0); b.place(W, 1, 1); b.place(W, 2, 0)
    assert b.get_n_liberty_if(W, 0, 0) == 1
    assert b.get_n_liberty_if(W, 2, 1) == 4
    assert b.get_n_liberty_if(W, 0, 2) == 2
    assert b.get_n_liberty_if(W, 9, 9) == 4
    b.place(W, 0, 2)
    assert b.get_n_liberty_if(W, 0, 0) == 2
    # the bench position of board_fast.rs:564-580
    c = api.Board(7.5)
    for x, y in ((1, 0), (1, 1), (1, 2), (0, 2)):
        c.place(B, x, y)
    for x, y in ((0, 1), (2, 0), (2, 1), (2, 2), (0, 3), (1, 3)):
        c.place(W, x, y)
    assert c.get_n_liberty_if(W, 0, 0) == 3


def test_ladder_corner_capture(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    for x, y in ((0, 0), (0, 18), (18, 0), (18, 18)):
        b.place(B, x, y)
    ladders = {(1, 0), (0, 1), (18, 17), (17, 18), (1, 18), (18, 1), (0, 17), (17, 0)}
    for y in range(19):
        for x in range(19):
            if b.is_valid(W, x, y):
                assert b.is_ladder_capture(W, x, y) == ((x, y) in ladders), (x, y)


def test_ladder_capture(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    b.place(W, 3, 3); b.place(B, 2, 3); b.place(B, 3, 2); b.place(B, 4, 2)
    for y in range(19):
        for x in range(19):
            if b.is_valid(B, x, y):
                assert b.is_ladder_capture(B, x, y) == ((x, y) == (3, 4)), (x, y)


def test_ladder_escape(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    b.place(W, 3, 3); b.place(W, 15, 15)                 # ladder breaker on the diagonal
    for x, y in ((2, 3), (3, 2), (4, 2), (3, 4)):
        b.place(B, x, y)
    for y in range(19):
        for x in range(19):
            if b.is_valid(W, x, y):
                assert not b.is_ladder_capture(B, x, y), (x, y)
                assert b.is_ladder_escape(W, x, y) == ((x, y) == (4, 3)), (x, y)


NOT_LADDER_GAME = [
    (B, 15, 3), (W, 3, 15), (B, 16, 15), (W, 3, 2), (B, 14, 16), (W, 2, 4), (B, 3, 5), (W, 3, 4), (B, 4, 5), (W, 4, 4), (B, 5, 5), (W, 1, 6),
    (B, 12, 2), (W, 16, 9), (B, 16, 7), (W, 16, 12), (B, 6, 3), (W, 14, 9), (B, 15, 10), (W, 15, 9), (B, 14, 7), (W, 16, 2), (B, 16, 3), (W, 15, 2),
    (B, 14, 2), (W, 14, 1), (B, 13, 1), (W, 14, 3), (B, 13, 2), (W, 17, 3), (B, 17, 4), (W, 17, 1), (B, 18, 3), (W, 17, 2), (B, 14, 4), (W, 15, 1),
    (B, 17, 11), (W, 17, 12), (B, 17, 10), (W, 15, 11), (B, 17, 9), (W, 14, 10), (B, 12, 8), (W, 5, 16), (B, 12, 10), (W, 14, 14), (B, 12, 15), (W, 12, 14),
    (B, 11, 14), (W, 12, 13), (B, 11, 15), (W, 15, 16), (B, 15, 15), (W, 13, 16), (B, 13, 15), (W, 14, 15), (B, 14, 17), (W, 15, 17), (B, 13, 17), (W, 17, 16),
    (B, 13, 14), (W, 15, 14), (B, 13, 13), (W, 14, 12), (B, 5, 2), (W, 11, 9), (B, 11, 8), (W, 8, 8), (B, 8, 6), (W, 4, 1), (B, 1, 14), (W, 1, 15),
    (B, 2, 15), (W, 2, 14), (B, 2, 16), (W, 1, 16), (B, 3, 14), (W, 2, 13), (B, 3, 16), (W, 4, 15), (B, 1, 13), (W, 1, 17), (B, 2, 12), (W, 3, 13),
    (B, 4, 12), (W, 3, 12), (B, 3, 11),
]


def test_not_ladder(api_full):
    """ladder.rs:282-317: a real game that once sent the reference's reader into an infinite loop."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    for color, x, y in NOT_LADDER_GAME:
        b.place(color, x, y)
    assert b.is_ladder_escape(W, 4, 13) is True


def test_not_ladder_due_to_self_atari(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    for first in ((1, 2), (3, 4)):                       # ladder.rs:319-350, both variants
        b = api.Board(7.5)
        for color, x, y in ((B,) + first, (W, 2, 4), (B, 2, 3), (W, 1, 5), (B, 1, 4)):
            b.place(color, x, y)
        assert b.is_ladder_capture(W, 1, 3) is False


# ---- feature planes: features.rs:99-251, plane by plane ----

def test_features_of_the_empty_board(api):
    f = api.Board(7.5).get_features(B).astype(np.float32)
    assert f.shape == (361, 32)
    assert np.all(f[:, 0] == np.float32(np.float16(1.0)))          # komi 7.5: clamp(0.5 + 0.5 * 7.5 / 7.5) = 1 on "we are black"
    assert np.all(f[:, 1] == 0) and np.all(f[:, 2] == 0) and np.all(f[:, 3:5] == 0)
    assert np.all(f[:, 5:11] == 0) and np.all(f[:, 17:23] == 0)    # no stones: no "our" / "opponent" liberties
    # liberties after a move: 2 in a corner, 3 on an edge, 4 inside, for either colour
    after = f[:, 11:17].sum(axis=1).reshape(19, 19)
    assert after[0, 0] == 2 and after[0, 9] == 3 and after[9, 9] == 4
    assert np.array_equal(f[:, 11:17], f[:, 23:29])
    assert np.all(f[:, 29:32] == 0)
    w = api.Board(0.0).get_features(W).astype(np.float32)
    assert np.all(w[:, 0] == 0) and np.all(w[:, 1] == 0.5)         # komi 0: 0.5 on "we are white"


def test_features_history_liberties_ko_and_ladder_planes(api):
    b = api.Board(7.5)
    b.place(W, 3, 3); b.place(B, 2, 3); b.place(B, 3, 2); b.place(B, 4, 2)   # the ladder_capture position, black to move
    f = b.get_features(B).astype(np.float32).reshape(19, 19, 32)
    assert f[2, 4, 3] == 1 and f[:, :, 3].sum() == 1               # most recent move (4, 2): plane index 3 at [y][x]
    assert f[2, 3, 4] == 1 and f[:, :, 4].sum() == 1               # the one before
    assert f[3, 3, 17:23].tolist() == [1, 1, 0, 0, 0, 0]           # the white stone has 2 liberties: "opponent liberties >= 1, >= 2"
    assert f[3, 2, 5:11].tolist() == [1, 1, 1, 0, 0, 0]            # black (2, 3): 3 liberties
    assert f[2, 3, 5:11].tolist() == [1, 1, 1, 1, 1, 0]            # black (3, 2) + (4, 2): 5 liberties, the block's count on both stones
    assert f[2, 4, 5:11].tolist() == [1, 1, 1, 1, 1, 0]
    assert f[4, 3, 30] == 1 and f[:, :, 30].sum() == 1             # (3, 4) is the ladder capture
    assert f[:, :, 29].sum() == 0 and f[:, :, 2].sum() == 0
    k = api.Board(7.5)                                             # the ko position: black may not retake at (0, 0)
    k.place(B, 0, 0); k.place(B, 0, 2); k.place(B, 1, 1); k.place(W, 1, 0); k.place(W, 0, 1)
    g = k.get_features(B).astype(np.float32).reshape(19, 19, 32)
    assert g[0, 0, 29] == 1 and np.all(g[:, :, 2] == 1)            # "is super-ko" at the point, "any move is super-ko" everywhere


def test_features_follow_the_symmetry(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    b.place(B, 3, 2); b.place(W, 15, 16); b.place(B, 4, 4)
    base = b.get_features(W).reshape(361, 32)
    for t in api.Transform.ALL:
        moved = b.get_features(W, t).reshape(361, 32)
        table = np.array([api.Transform.apply(t, i) for i in range(361)])
        assert np.array_equal(moved[table], base)                  # features[o.index(c, symmetry_table[point])] = plane c at point
    assert b.is_symmetric(api.Transform.Identity) and not b.is_symmetric(api.Transform.Rot90)
    assert all(api.Board(7.5).is_symmetric(t) for t in api.Transform.ALL)


def test_search_and_selfplay_on_the_real_board(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    b.place(B, 3, 3)
    prior = np.full(362, 1.0 / 361, dtype=np.float32); prior[3 * 19 + 3] = -np.inf; prior[361] = -np.inf
    root = api.Node(W, 0.5, prior)
    evals, _ = api.search_run(root, b, num_rollouts=60, num_threads=1, seed=4)
    counts = np.array([root.with_(i).count() for i in range(362)])
    # the search ends at the limit or as soon as the runner-up cannot catch up any more (EARLY-C, time_control/mod.rs:86-101)
    assert evals == counts.sum() <= 60 and root.is_done(60) and counts[3 * 19 + 3] == 0 and counts[361] == 0
    assert b.count() == 1                                          # the search works on clones
    # symmetric duplicates are eliminated below the root: after black (3, 3) white's replies come in mirror pairs
    # (transpose), so no child of the root's children... (checked through the prior of an expanded child)
    stats = api.selfplay_go(num_games=2, num_moves=6, num_rollouts=12, threads_per_search=2, seed=9, return_moves=True)
    record = stats["record"]
    assert stats["moves"] == 12 and np.all(record >= 0) and np.all(record < 361)
    again = api.selfplay_go(num_games=2, num_moves=6, num_rollouts=12, threads_per_search=1, seed=9, return_moves=True)
    once_more = api.selfplay_go(num_games=2, num_moves=6, num_rollouts=12, threads_per_search=1, seed=9, return_moves=True)
    assert np.array_equal(again["record"], once_more["record"])     # one search thread: the games are a function of the seed
    for game in record:                                             # every recorded move was legal when it was played
        replay = api.Board(7.5)
        for m, index in enumerate(game):
            assert replay.is_valid(m % 2, index % 19, index // 19)
            replay.place(m % 2, index % 19, index // 19)


def test_full_forward_of_the_empty_board(api_full):
    """full_forward (src/libdg_mcts/lib.rs:83-130): every symmetry is evaluated, the candidates are what the initial policy
    leaves (symmetric duplicates removed: on the empty board one eighth of the points, incl. the diagonals and the centre,
    = 55, plus pass), the result sums to one, and a second call is served from the transposition cache."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    value, prior, n = api.full_forward(b, B)
    assert n == 8 and 0.0 <= value <= 1.0
    finite = np.isfinite(prior)
    assert finite.sum() == 56 and finite[361]
    assert np.isclose(prior[finite].sum(), 1.0, atol=1e-5)
    cache = api.LruCache(1000)
    first = api.full_forward(b, B, cache=cache)
    second = api.full_forward(b, B, cache=cache)
    assert first[2] == 8 and second[2] == 0                        # the count is the number of predictor calls
    assert np.array_equal(first[1], second[1]) and first[0] == second[0]
    assert np.array_equal(first[1], prior)                         # the cache does not change the answer
    b.place(B, 3, 4)                                               # nothing symmetric about this one
    _, prior, n = api.full_forward(b, W)
    assert n == 8 and np.isfinite(prior).sum() == 361 and prior[4 * 19 + 3] == -np.inf


def test_cached_search_reports_hits_and_same_tree(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5)
    b.place(B, 3, 3)
    prior = np.full(362, 1.0 / 361, dtype=np.float32); prior[3 * 19 + 3] = -np.inf; prior[361] = -np.inf
    cache = api.LruCache(10_000)
    counts = []
    for _ in range(2):
        root = api.Node(W, 0.5, prior)
        evals, conflicts, hits = api.search_run(root, b, num_rollouts=60, num_threads=1, seed=4, cache=cache)
        counts.append((evals, hits, np.array([root.with_(i).count() for i in range(362)])))
    assert counts[0][0] + counts[0][1] == counts[0][2].sum() <= 60                            # (EARLY-C may stop before the limit)
    assert counts[1][0] == 0 and counts[1][1] == counts[0][2].sum()                           # second search: all from the cache
    assert np.array_equal(counts[0][2], counts[1][2])
    plain = api.Node(W, 0.5, prior)
    api.search_run(plain, b, num_rollouts=60, num_threads=1, seed=4)
    assert np.array_equal(counts[0][2], np.array([plain.with_(i).count() for i in range(362)]))


def test_selfplay_records_games_in_the_reference_format(api):
    out = api.selfplay_go(num_games=2, num_moves=5, num_rollouts=10, threads_per_search=1, seed=3, cache=api.LruCache(5000),
                          return_moves=True, return_sgf=True)
    assert len(out["sgf"]) == 2
    for game, record in zip(out["sgf"], out["record"]):
        assert game.startswith("(;GM[1]FF[4]DT[") and "]SZ[19]RU[Chinese]KM[" in game and game.endswith(")")
        assert game.count(";B[") + game.count(";W[") == 5
        letters = "abcdefghijklmnopqrs"
        for m, index in enumerate(record):                         # the SGF vertex is the move that was played (CGoban, sgf.rs:36-43)
            x, y = index % 19, index // 19
            assert ";%s[%s%s]" % ("BW"[m % 2], letters[x], letters[y]) in game


def test_real_games_reach_the_reference_positions(api_full):
    """src/dg_tests/tests/real_games.rs:27-117 replays three professional games (one with a triple ko) and pins the final
    zobrist hash.  The build keeps its own zobrist constants, so the pin is on the final POSITION: tests/golden/real_games.json
    holds the positions that reproduced the reference's expected hashes under the reference's table when the fixture was
    made (tests/golden/make_real_games.py); when the reference checkout is present the hash is recomputed here as well."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    import importlib.util
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_real_games", os.path.join(here, "golden", "make_real_games.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    import re
    games = json.load(open(os.path.join(here, "golden", "real_games.json")))["games"]
    assert [g["zobrist_hash"] for g in games] == [8306876591293505520, 6611710067991275858, 8830202173104103970]
    table = gen.reference_table() if os.path.exists(gen.REF_TABLE) else None
    for g in games:
        board = gen.replay(api, re.findall(r";([BW])\[([a-z]*)\]", g["moves"]))          # asserts that every move is valid
        final = gen.position(board)
        assert final == g["final_position"], g["name"]
        if table is not None:
            assert gen.hash_of(final, table) == g["zobrist_hash"]


# ---- Benson, score, stone status (utils/benson.rs, utils/score.rs) -------------------------------------------------------

def _senseis_example():
    """The blocks and regions of benson.rs:375-566 (FakeAllBlocks / FakeAllRegions): a 6x6 diagram from Sensei's Library."""
    blocks = [
        ([(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 3), (1, 4), (2, 4), (2, 5)], [(1, 1), (1, 2), (2, 3), (3, 4), (3, 5), (1, 5), (0, 4)]),
        ([(0, 5)], [(0, 4), (1, 5)]),
        ([(2, 1)], [(2, 0), (2, 2), (1, 1), (3, 1)]),
        ([(4, 5)], [(3, 5), (5, 5), (4, 4)]),
        ([(3, 0), (4, 0), (4, 1), (5, 1), (5, 2), (5, 3), (5, 4)], [(2, 0), (5, 0), (5, 5), (3, 1), (4, 2), (4, 3), (4, 4)]),
    ]
    regions = [
        ([(0, 4)], [(0, 5), (0, 3), (1, 4)]),
        ([(1, 5)], [(0, 5), (2, 5), (1, 4)]),
        ([(2, 0)], [(1, 0), (3, 0), (2, 1)]),
        ([(5, 0)], [(5, 1), (4, 0)]),
        ([(5, 5)], [(5, 4), (4, 5)]),
        ([(1, 1), (1, 2), (2, 2), (2, 3), (3, 1), (3, 2), (3, 3), (3, 4), (3, 5), (4, 2), (4, 3), (4, 4)],
         [(0, 1), (0, 2), (1, 0), (1, 3), (1, 4), (2, 1), (2, 4), (2, 5), (3, 0), (4, 1), (4, 5), (5, 2), (5, 3), (5, 4)]),
    ]
    return blocks, regions


def test_benson_restatement_reproduces_the_reference_example():
    """benson.rs:571-599: exactly these twelve points are not `valid` (alive stones and the eyes they live on)."""
    from oracle import benson as restated
    status = restated.benson(*_senseis_example())
    invalid = {(0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (1, 0), (1, 3), (1, 4), (1, 5), (2, 4), (2, 5)}
    assert {p for p, s in status.items() if s != restated.NONE} == invalid


def test_benson_matches_the_restatement_on_random_positions(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    from oracle import benson as restated
    rng = np.random.default_rng(11)
    interesting = 0
    for trial in range(40):
        board, stones = api.Board(0.5), {}
        density = (0.5, 0.75, 0.9, 0.97)[trial % 4]                # dense boards are where alive groups appear
        for index in rng.permutation(361):
            if rng.random() > density:
                continue
            x, y = int(index % 19), int(index // 19)
            color = int(rng.random() < (0.5 if trial % 3 else 0.8))
            if board.is_valid(color, x, y):
                board.place(color, x, y)
        for y in range(19):
            for x in range(19):
                if board.at(x, y) is not None:
                    stones[(x, y)] = board.at(x, y)
        for color in (B, W):
            got = board.benson(color)
            assert np.array_equal(got, np.array(restated.benson_of_position(stones, color), dtype=np.uint8)), (trial, color)
            interesting += int(got.any())
    assert interesting >= 10                                       # the comparison was not all-zero boards


def test_benson_eyes_and_empty(api_full):
    """benson.rs:601-609 (empty_is_all_valid) and :655-676 (eyes)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    assert not api.Board(0.5).benson(B).any()
    b = api.Board(0.5)
    for x, y in ((0, 1), (1, 1), (2, 0), (2, 1), (3, 1), (4, 0), (4, 1)):
        b.place(W, x, y)
    b.place(B, 0, 0)
    status = b.benson(W)
    for y in range(19):
        for x in range(19):
            if b.at(x, y) == W:
                assert status[y, x] == 1
            elif (x, y) in ((0, 0), (1, 0), (3, 0)):
                assert status[y, x] == 2


def test_scores(api_full):
    """score.rs:289-353: score_black, score_white, score_black_white, checker_board_black / white."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(7.5); b.place(B, 0, 0)
    assert not b.is_scorable() and b.get_score() == (361, 0)
    b = api.Board(7.5); b.place(W, 0, 0)
    assert not b.is_scorable() and b.get_score() == (0, 361)
    b = api.Board(7.5)
    for x, y in ((1, 0), (0, 1), (1, 1), (1, 2), (0, 3), (1, 3)):
        b.place(W, x, y)
    for x, y in ((2, 0), (2, 1), (2, 2), (2, 3), (0, 4), (1, 4), (2, 4)):
        b.place(B, x, y)
    assert not b.is_scorable() and b.get_score() == (353, 8)
    for color, expected in ((B, (361, 0)), (W, (0, 361))):
        b = api.Board(0.5)
        for y in range(19):
            for x in range(1, 19, 2):
                b.place(color, x, y)
        assert b.is_scorable() and b.get_score() == expected
    assert api.Board(7.5).get_score() == (0, 0)                    # score.rs:131-135: nothing played


def test_eyes_should_be_territory(api_full):
    """score.rs:355-405, and the RE[] / TB / TW text GameResult::Ended makes of it (game_result.rs:46-93)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(0.5)
    for x, y in ((0, 1), (1, 1), (2, 0), (2, 1), (3, 1), (4, 0), (4, 1)):
        b.place(W, x, y)
    b.place(B, 0, 0); b.place(B, 9, 9)
    status = b.get_stone_status()
    for y in range(19):
        for x in range(19):
            if b.at(x, y) == W:
                assert status[y, x] == api.STONE_ALIVE | api.TERRITORY_WHITE
            elif (x, y) == (0, 0):
                assert status[y, x] == api.STONE_DEAD | api.TERRITORY_WHITE
            elif (x, y) == (9, 9):
                assert status[y, x] == api.STONE_SEKI | api.TERRITORY_BLACK
            else:
                assert status[y, x] == api.TERRITORY_WHITE
    winner, territory = b.result_sgf()
    assert winner == "W+359.5"                                      # 360 points + 0.5 komi against the one seki stone
    assert territory.startswith("TB[jj]TW[aa][ba]") and territory.count("[") == 361


def test_scoring_search_candidates(api_full):
    """ScoringPolicyChecker (options.rs:110-139): no pass, no Benson eye of either colour, no own simple eye (is_eye :192-213,
    tests :221-258 corner / side / middle)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Board(0.5)
    for x, y in ((0, 1), (1, 1), (2, 0), (2, 1), (3, 1), (4, 0), (4, 1)):
        b.place(W, x, y)
    _, prior, _ = api.full_forward(b, B, scoring=True)
    assert prior[361] == -np.inf and prior[0 * 19 + 1] == -np.inf and prior[0 * 19 + 3] == -np.inf    # white's two eyes
    assert np.isfinite(prior[5 * 19 + 5])
    _, standard, _ = api.full_forward(b, B)
    assert np.isfinite(standard[361])
    for stones, eye in ((((1, 0), (0, 1), (1, 1)), (0, 0)),                                          # corner
                        (((3, 0), (5, 0), (4, 1), (3, 1), (5, 1)), (4, 0)),                           # side
                        (((3, 4), (5, 4), (4, 3), (4, 5), (3, 3), (5, 3), (3, 5)), (4, 4))):          # middle
        b = api.Board(0.5)
        for x, y in stones:
            b.place(B, x, y)
        _, prior, _ = api.full_forward(b, B, scoring=True)
        assert prior[eye[1] * 19 + eye[0]] == -np.inf
        _, prior, _ = api.full_forward(b, W, scoring=True)
        # for white the point is either a suicide or a legal move: it is not excluded for being an eye
        assert np.isfinite(prior[eye[1] * 19 + eye[0]]) == b.is_valid(W, *eye)


def test_choose_on_a_policy(api_full):
    """choose.rs:54-88 on real weights: -inf entries are ignored by the total and never chosen."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    p = np.array([0.1, -np.inf, 0.6, 0.3, -np.inf])
    assert api.choose_f64(p, 0.5, 1.0, 0.0)[0] == 2
    picks = {api.choose_f64(p, 0.05, 1.0, at)[0] for at in np.linspace(0.0, 1.0, 21)}
    assert picks == {0, 2, 3}
    assert api.choose_f64(p, 0.5, 1.0, 1.0)[0] == 2               # the lower half of the mass (0.1, 0.3) is cut off


def test_selfplay_plays_games_to_the_end(api_full):
    """self_play_one (self_play.rs:423-459) with the deterministic hash predictor: tiny budgets, but whole games: every record
    is a GameResult::Ended record, every stone was legal, and a game that ended did so on a scorable board after two passes."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    out = api.selfplay_go(num_games=2, num_moves=0, num_rollouts=12, threads_per_search=1, seed=5, cache=api.LruCache(20000),
                          return_moves=True, return_sgf=True)
    assert out["games"] == 2 and len(out["sgf"]) == 2
    import re
    for game, record in zip(out["sgf"], out["record"]):
        assert re.match(r"^\(;GM\[1\]FF\[4\]DT\[\d{4}-\d\d-\d\dT\d\d:\d\d:\d\d\+0000\]SZ\[19\]RU\[Chinese\]KM\[-?\d+\.5\]RE\[(0|[BW]\+\d+\.\d)\];[BW]\[", game)
        assert game.endswith(")")
        replay = api.Board(float(re.search(r"KM\[(-?\d+\.5)\]", game).group(1)))
        moves = re.findall(r";([BW])\[([a-s]{2})?\]", game)
        stones = [m for m in moves if m[1]]
        assert len(stones) == int((record >= 0).sum())
        for (colour, vertex), index in zip(stones, record):
            x, y = ord(vertex[0]) - 97, ord(vertex[1]) - 97
            assert index == 19 * y + x and replay.is_valid("BW".index(colour), x, y)
            replay.place("BW".index(colour), x, y)
        assert [c for c, _ in moves] == ["B", "W"] * (len(moves) // 2) + ["B"] * (len(moves) % 2)     # turns alternate, passes included
        if len(stones) < 722:                                       # ended by two passes
            assert moves[-1][1] == "" and moves[-2][1] == "" and replay.is_scorable()
            winner, territory = replay.result_sgf()
            assert "RE[%s]" % winner in game and game.endswith(territory + ")")
    assert out["games_ended"] == sum(int((r >= 0).sum()) < 722 for r in out["record"])


def test_incremental_bookkeeping_equals_a_flood_fill(api_full):
    """place() keeps blocks and liberties incrementally (merges, captures of several blocks at once, snap-backs); after every
    stone of dense random games -- legal moves and, now and then, an illegal one (place() does not check) -- the state must
    equal a from-scratch flood fill, and the liberty counts must equal a brute-force count."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng(5)
    for game in range(6):
        b = api.Board(0.5)
        color = 0
        for move in range(500):
            points = rng.permutation(361)
            for index in points:
                x, y = int(index % 19), int(index // 19)
                if b.is_valid(color, x, y) or (b.at(x, y) is None and rng.random() < 0.002):
                    b.place(color, x, y)
                    break
            else:
                break
            assert b.check(), (game, move)
            color ^= 1
        stones = {(x, y): b.at(x, y) for y in range(19) for x in range(19) if b.at(x, y) is not None}
        for (x, y), c in list(stones.items())[::7]:                 # brute-force liberties of a sample of blocks
            todo, block, liberties = [(x, y)], {(x, y)}, set()
            while todo:
                u, v = todo.pop()
                for s, t in ((u + 1, v), (u - 1, v), (u, v + 1), (u, v - 1)):
                    if not (0 <= s < 19 and 0 <= t < 19):
                        continue
                    if (s, t) not in stones:
                        liberties.add((s, t))
                    elif stones[(s, t)] == c and (s, t) not in block:
                        block.add((s, t)); todo.append((s, t))
            assert b.get_n_liberty(x, y) == len(liberties)


def test_selfplay_with_fewer_games_in_flight_plays_the_same_games(api):
    """self_play (self_play.rs:478-495) keeps num_parallel games in flight and starts the next one when a game ends; a game is a
    function of (seed, game id), so the records do not depend on how many run at a time."""
    kw = dict(num_games=5, num_moves=6, num_rollouts=10, threads_per_search=1, seed=13, return_moves=True)
    all_at_once = api.selfplay_go(**kw)
    two_at_a_time = api.selfplay_go(num_parallel=2, **kw)
    assert np.array_equal(all_at_once["record"], two_at_a_time["record"]) and two_at_a_time["moves"] == 30


def test_policy_play_and_expert_iteration(api):
    """BASELINE.json configs[4] (--ex-it --num-rollout 1): with a budget of one rollout every move is drawn from the 8-symmetry
    policy (self_play.rs:369-405; Played::from_forward has one rollout, which Played::fmt prints without TV[] / P[]); with expert iteration about one move in twenty carries a full search instead
    (self_play.rs:277-319: TV[] = that search's size, TR[] = its prior's best move), and the move sequence is NOT changed by it."""
    import re
    kw = dict(num_games=2, num_moves=40, num_rollouts=1, threads_per_search=1, seed=17, return_moves=True, return_sgf=True)
    plain = api.selfplay_go(**kw)
    assert plain["ex_it_searches"] == 0 and plain["conflicts"] == 0
    for game in plain["sgf"]:
        assert "TV[" not in game and "TR[" not in game and game.count("V[") == 40
    assert plain["evals"] <= 2 * 40 * 8 + 16                       # at most eight evaluations per move: no search ran
    ex_it = api.selfplay_go(ex_it_rollouts=30, **kw)
    assert ex_it["ex_it_searches"] >= 1
    sizes = [int(v) for game in ex_it["sgf"] for v in re.findall(r"TV\[(\d+)\]", game)]
    assert len(sizes) == ex_it["ex_it_searches"] and all(1 < n <= 30 for n in sizes)     # EARLY-C may stop a search before its limit
    assert sum(game.count("TR[") for game in ex_it["sgf"]) == ex_it["ex_it_searches"]
    # the expert's search draws from the game's random stream, so only the moves before the first such search must agree
    first = min(len(re.findall(r";[BW]\[", game.split("TV[")[0])) for game in ex_it["sgf"] if "TV[" in game)
    for a, b in zip(plain["record"], ex_it["record"]):
        assert np.array_equal(a[:max(first - 1, 0)], b[:max(first - 1, 0)])


def test_records_leave_over_the_channel_as_their_games_end(api):
    """self_play.rs:471-497: every game thread SENDS its GameResult and the caller prints it and lets go of it.  `on_game` is that
    channel: one call per game as it leaves, the record the same bytes `sgf_out` would hold at the end, the calls serialised, the
    global game id with each -- and nothing is kept for a caller who did not ask for all records at the end (a job that replaces its
    games for hours, dgnn_selfplay_window, would otherwise hold every record it ever produced)."""
    import threading
    kw = dict(num_games=6, num_moves=30, num_rollouts=1, threads_per_search=1, seed=23, first_game=2, game_stride=3, num_parallel=3)
    kept = api.selfplay_go(return_sgf=True, **kw)
    streamed, inside = {}, []
    busy = threading.Lock()

    def on_game(game_id, record, whole):
        inside.append(busy.acquire(blocking=False))
        streamed[game_id] = (record.decode("ascii"), whole)
        busy.release()
    both = api.selfplay_go(return_sgf=True, on_game=on_game, **kw)
    only = {}
    api.selfplay_go(on_game=lambda g, r, w: only.__setitem__(g, r.decode("ascii")), **kw)
    strip = lambda text: __import__("re").sub(r"DT\[[^\]]*\]", "", text)               # (the header carries the time of day)
    assert sorted(streamed) == [2 + 3 * i for i in range(6)] == sorted(only) and all(inside)
    assert [strip(streamed[2 + 3 * i][0]) for i in range(6)] == [strip(g) for g in both["sgf"]] == [strip(g) for g in kept["sgf"]]
    assert [strip(only[2 + 3 * i]) for i in range(6)] == [strip(g) for g in kept["sgf"]]
    assert all(whole for _, whole in streamed.values())                                 # 30 stones is this call's cap: whole games


# ---- round 3: the bit-board forms the search calls (Benson, candidate moves) against the flood-fill forms they replaced

def _simple_eye(board, color, x, y):
    """is_eye, src/libdg_mcts/options.rs:192-213."""
    own = lambda u, v: 0 <= u < 19 and 0 <= v < 19 and board.at(u, v) == color
    cross = own(x + 1, y) + own(x - 1, y) + own(x, y + 1) + own(x, y - 1)
    diagonal = own(x + 1, y + 1) + own(x + 1, y - 1) + own(x - 1, y + 1) + own(x - 1, y - 1)
    edge_x, edge_y = x in (0, 18), y in (0, 18)
    if edge_x and edge_y:
        return cross >= 2 and diagonal >= 1
    if edge_x or edge_y:
        return cross >= 3 and diagonal >= 2
    return cross >= 4 and diagonal >= 3


def _check_fast_forms(api, board):
    status = {}
    for color in (0, 1):
        status[color] = board.benson(color)
        assert np.array_equal(status[color], board.benson_flood(color)), (board.count(), color)
    for color in (0, 1):
        standard, scoring = board.candidates(color), board.candidates(color, scoring=True)
        assert standard[361] and not scoring[361]
        for p in range(361):
            x, y = p % 19, p // 19
            valid = board.is_valid(color, x, y)
            assert standard[p] == valid, (board.count(), color, x, y)
            expect = valid and status[0][y, x] != 2 and status[1][y, x] != 2 and not _simple_eye(board, color, x, y)
            assert scoring[p] == expect, (board.count(), color, x, y)
    return int((status[0] == 1).any() or (status[1] == 1).any())


def test_bitboard_benson_and_candidates_equal_the_flood_fill_forms(api_full):
    """go.cpp: benson_sets / valid_moves / go_candidates_scoring (bit boards, what every rollout calls) against benson_flood (the
    reference's steps, utils/benson.rs:62-190) and the per-point definitions (options.rs:44-214, board.rs:140-142): every 7th position
    of the three reference games, and random games played to the end with the ScoringSearch rule (no own eyes filled), which leave
    boards full of small eyes, unconditionally alive groups, captures and kos."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    import json as _json, re as _re
    checked = alive = 0
    for game in _json.load(open(os.path.join(os.path.dirname(__file__), "golden", "real_games.json")))["games"]:
        board = api.Board(7.5)
        for n, (color, vertex) in enumerate(_re.findall(r";([BW])\[([a-s]{2})\]", game["moves"])):
            board.place(0 if color == "B" else 1, ord(vertex[0]) - 97, ord(vertex[1]) - 97)
            if n % 7 == 0:
                alive += _check_fast_forms(api, board)
                checked += 1
    rng = np.random.default_rng(5)
    for _ in range(3):
        board, color, passes = api.Board(0.5), 0, 0
        while passes < 2 and board.count() < 500:
            options = np.nonzero(board.candidates(color, scoring=True)[:361])[0]
            if len(options) == 0:
                passes += 1
            else:
                passes = 0
                p = int(rng.choice(options))
                board.place(color, p % 19, p // 19)
                if board.count() % 9 == 0 or board.count() > 300:
                    alive += _check_fast_forms(api, board)
                    checked += 1
            color ^= 1
        assert board.check()
    print(f"{checked} positions, {alive} with an unconditionally alive block")
    assert checked > 150 and alive > 20
