"""The INDEPENDENT oracle of the input contract (oracle/features.py: features::V1 by flood fill, simulated moves and whole-position
super-ko) -- pinned to the reference, then used as the checker of the product's host statement of the planes
(csrc/go.cpp features_v1).  The -m gpu twin (tests/test_features_gpu.py) holds the DEVICE kernel to the same committed fixture.

Pins of the oracle itself: the reference's own unit tests of its parts, restated (board_fast.rs:545-580 `check_get_n_liberty_if`;
ladder.rs:173-351 `ladder_corner_capture`, `ladder_capture`, `ladder_escape`, `not_ladder`, `not_ladder_due_to_self_atari{,_2}`), and
the three `real_games.rs` games: its replay reaches, move for move legally, the final positions whose zobrist hashes the reference
expects (tests/golden/real_games.json holds them, verified with the reference's table by make_real_games.py)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import features as F
from test_go_host import NOT_LADDER_GAME

B, W = F.BLACK, F.WHITE


def at(x, y):
    return y * 19 + x


def stones_of(moves):
    s = F.Stones()
    for color, x, y in moves:
        s.place(color, at(x, y))
    return s


# ---- the reference's unit tests of the parts

def test_oracle_get_n_liberty_if_like_the_reference():
    s = stones_of([(B, 0, 1), (B, 1, 0), (W, 1, 1), (W, 2, 0)])                      # board_fast.rs:545-560
    assert [s.num_liberties_if(W, at(*p)) for p in ((0, 0), (2, 1), (0, 2), (9, 9))] == [1, 4, 2, 4]
    s.place(W, at(0, 2))
    assert s.num_liberties_if(W, at(0, 0)) == 2
    s = stones_of([(B, 1, 0), (B, 1, 1), (B, 1, 2), (B, 0, 2), (W, 0, 1), (W, 2, 0), (W, 2, 1), (W, 2, 2), (W, 0, 3), (W, 1, 3)])
    assert s.num_liberties_if(W, at(0, 0)) == 3                                       # board_fast.rs:564-580


def test_oracle_ladders_like_the_reference():
    s = stones_of([(B, 0, 0), (B, 0, 18), (B, 18, 0), (B, 18, 18)])                   # ladder.rs:173-207
    ladders = {(1, 0), (0, 1), (18, 17), (17, 18), (1, 18), (18, 1), (0, 17), (17, 0)}
    for p in range(361):
        if s.is_valid(W, p):
            assert F.is_ladder_capture(s, W, p) == ((p % 19, p // 19) in ladders), p
    s = stones_of([(W, 3, 3), (B, 2, 3), (B, 3, 2), (B, 4, 2)])                       # ladder.rs:209-236
    for p in range(361):
        if s.is_valid(B, p):
            assert F.is_ladder_capture(s, B, p) == (p == at(3, 4)), p
    s = stones_of([(W, 3, 3), (W, 15, 15), (B, 2, 3), (B, 3, 2), (B, 4, 2), (B, 3, 4)])   # ladder.rs:251-277
    for p in range(361):
        if s.is_valid(W, p):
            assert not F.is_ladder_capture(s, B, p), p
            assert F.is_ladder_escape(s, W, p) == (p == at(4, 3)), p
    s = stones_of([(B if c == 0 else W, x, y) for c, x, y in NOT_LADDER_GAME])        # ladder.rs:282-317
    assert F.is_ladder_escape(s, W, at(4, 13)) is True
    for first in ((1, 2), (3, 4)):                                                    # ladder.rs:319-350
        s = stones_of([(B,) + first, (W, 2, 4), (B, 2, 3), (W, 1, 5), (B, 1, 4)])
        assert F.is_ladder_capture(s, W, at(1, 3)) is False


def test_oracle_replay_reaches_the_positions_the_reference_hashes():
    for game in json.load(open(os.path.join(GOLDEN, "real_games.json")))["games"]:
        board = None
        for board, _, _ in F.replay(game["moves"]):        # (asserts that every move is legal: no suicide, no super-ko)
            pass
        assert F.position_text(board) == game["final_position"], game["name"]


def test_oracle_empty_board_and_komi_constant():
    board = F.Board(7.5)
    mask, komi = F.plane_masks(board, B)
    # nothing but the colour plane and "our / their liberties after a move >= 1 .. 4" on an empty board (2 .. 4 at edges and corners)
    assert komi == int(np.float16(1.0).view(np.uint16))
    centre, corner, edge = int(mask[at(9, 9)]), int(mask[at(0, 0)]), int(mask[at(0, 9)])
    after = lambda n: sum(1 << (11 + i) for i in range(n)) | sum(1 << (23 + i) for i in range(n))     # noqa: E731
    assert centre == 1 | after(4) and corner == 1 | after(2) and edge == 1 | after(3)
    assert F.plane_masks(F.Board(0.5), W)[1] == int(np.float16(np.float32(0.5) + np.float32(0.25) / np.float32(7.5)).view(np.uint16))
    assert int(F.plane_masks(F.Board(-20.0), B)[0][0]) & 3 == 0                     # the clamp: a constant 0 is an empty plane


# ---- the product's host statement against the oracle

def product_masks(api, board, color, symmetry):
    f = board.get_features(color, symmetry).reshape(361, 32).view(np.uint16)
    mask = (((f != 0).astype(np.uint32)) << np.arange(32, dtype=np.uint32)[None, :]).sum(axis=1).astype(np.uint32)
    return mask, int(f[:, color].max())


def replay_both(api, moves):
    """The product's board and the oracle's, stone by stone, the way the reference's harness replays (x, 18 - y)."""
    board = api.Board(7.5)
    for oracle_board, c, p in F.replay(moves):
        board.place(c - 1, p % 19, p // 19)
        yield board, oracle_board


def fixture():
    return np.load(os.path.join(GOLDEN, "planes_real_games.npz"))


def test_go_cpp_features_equal_the_committed_oracle_planes(api):
    """csrc/go.cpp features_v1 against tests/golden/planes_real_games.npz (made by tests/golden/make_planes.py from oracle/features.py):
    every 5th position of the three games x both colours x eight symmetries, every plane, the komi constant."""
    fx = fixture()
    games = json.load(open(os.path.join(GOLDEN, "real_games.json")))["games"]
    wanted = {}
    for i in range(len(fx["game"])):
        wanted.setdefault((int(fx["game"][i]), int(fx["stones"][i])), []).append(i)
    checked = 0
    for g, game in enumerate(games):
        for n, (board, _) in enumerate(replay_both(api, game["moves"])):
            for i in wanted.get((g, n + 1), ()):
                color, symmetry = int(fx["color"][i]), int(fx["symmetry"][i])
                mask, komi = product_masks(api, board, color - 1, symmetry)
                diff = np.nonzero(mask != fx["masks"][i])[0]
                assert diff.size == 0, (game["name"], n + 1, color, symmetry, int(diff[0]), hex(int(mask[diff[0]])), hex(int(fx["masks"][i][diff[0]])))
                assert komi == int(fx["komi"][i])
                checked += 1
    assert checked == len(fx["game"]) == 2064
    planes = [int(((fx["masks"] >> c) & 1).sum()) for c in range(32)]
    assert all(count > 0 for count in planes)              # every plane occurs in the fixture: ko (plane 29), both ladder planes


def test_go_cpp_features_equal_the_live_oracle_on_every_position(api):
    """The same comparison with the oracle run HERE on every position of the three games (both colours; symmetries 0 and one other
    per position) -- five times the fixture's positions, and the proof that the committed fixture is what the oracle computes."""
    fx = fixture()
    index = {(int(fx["game"][i]), int(fx["stones"][i]), int(fx["color"][i]), int(fx["symmetry"][i])): i for i in range(len(fx["game"]))}
    games = json.load(open(os.path.join(GOLDEN, "real_games.json")))["games"]
    positions = reproduced = 0
    for g, game in enumerate(games):
        for n, (board, oracle_board) in enumerate(replay_both(api, game["moves"])):
            for color in (B, W):
                base, komi = F.plane_masks(oracle_board, color)
                for symmetry in {0, 1 + (n + color) % 7}:
                    want = F.apply_symmetry(base, symmetry)
                    mask, product_komi = product_masks(api, board, color - 1, symmetry)
                    assert np.array_equal(mask, want), (game["name"], n + 1, color, symmetry)
                    assert product_komi == komi
                    i = index.get((g, n + 1, color, symmetry))
                    if i is not None:
                        assert np.array_equal(fx["masks"][i], want) and int(fx["komi"][i]) == komi
                        reproduced += 1
            positions += 1
    assert positions == 648 and reproduced > 250


def test_a_misread_plane_order_fails_the_comparison(api):
    """What the fixture is FOR: a restatement that swapped two planes (here: ladder capture <-> ladder escape, and "our liberties" <->
    "their liberties") does not pass -- made with the generator's own --wrong-order switch on the first 60 stones of the games."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_planes", os.path.join(GOLDEN, "make_planes.py"))
    make_planes = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(make_planes)
    games = json.load(open(os.path.join(GOLDEN, "real_games.json")))["games"]
    boards = {}
    for g, game in enumerate(games):
        for n, (board, _) in enumerate(replay_both(api, game["moves"])):
            if n + 1 > 60:
                break
            if (n + 1) % 5 == 0:
                boards[(g, n + 1)] = board.clone()
    for swap in ((30, 31), (5, 17), None):
        mismatches = 0
        for g, stones, color, symmetry, mask, komi in make_planes.entries(swap=swap, limit=60):
            got, _ = product_masks(api, boards[(g, stones)], color - 1, symmetry)
            mismatches += int(not np.array_equal(got, mask))
        assert (mismatches > 0) == (swap is not None), (swap, mismatches)
