#!/usr/bin/env python
"""Generates tests/golden/real_games.json: the three full games the reference replays in src/dg_tests/tests/real_games.rs
(:27-49 Lee Sedol - AlphaGo game 4, :53-74 Ke Jie - AlphaGo game 2, :80-117 Park Taehee - Kim Dayoung, the triple-ko game),
the zobrist hash each of those tests expects (:49, :74, :117), and the final position this build's board reaches -- written
ONLY after that position, hashed with the reference's own table (src/libdg_go/zobrist.rs, indexed
[colour 1|2][20*(y+1) + x+1], board_fast.rs:409,450 / point.rs:26-33), reproduces the expected hash.  The table is read from
/root/reference when this script runs and is NOT stored: the fixture holds the games (test data), the expected hashes and
the verified final positions.  Run in the build container (needs /root/reference and the built library):

    python tests/golden/make_real_games.py
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_TESTS = "/root/reference/src/dg_tests/tests/real_games.rs"
REF_TABLE = "/root/reference/src/libdg_go/zobrist.rs"


def reference_table():
    words = re.findall(r"0x([0-9a-f]{16})", open(REF_TABLE).read())
    assert len(words) == 3 * 420
    return [[int(w, 16) for w in words[420 * c:420 * (c + 1)]] for c in range(3)]


def replay(api, moves):
    """common/mod.rs:37-77: SGF vertex -> Point::new(x, 18 - y); every move must be valid; empty / out of range = pass."""
    board = api.Board(7.5)
    for n, (colour, vertex) in enumerate(moves):
        if len(vertex) != 2:
            continue
        x, y = ord(vertex[0]) - 97, ord(vertex[1]) - 97
        if x >= 19 or y >= 19:
            continue
        c = 0 if colour == "B" else 1
        assert board.is_valid(c, x, 18 - y), "invalid move %d: %s %s" % (n + 1, colour, vertex)
        board.place(c, x, 18 - y)
        assert board.check(), "block / liberty bookkeeping broken after move %d" % (n + 1)
    return board


def position(board):
    return "".join("." if board.at(x, y) is None else "XO"[board.at(x, y)] for y in range(19) for x in range(19))


def hash_of(position, table):
    h = 0
    for i, ch in enumerate(position):
        if ch != ".":
            x, y = i % 19, i // 19
            h ^= table[1 if ch == "X" else 2][20 * (y + 1) + x + 1]
    return h


def main():
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api
    text = open(REF_TESTS).read()
    table = reference_table()
    games = []
    for name, body, expected in re.findall(r'fn (\w+)\(\) \{\s*let board = playout_game\(r#"(.*?)"#, None\);\s*assert_eq!\(board\.zobrist_hash\(\), (\d+)',
                                           text, re.S):
        moves = re.findall(r";([BW])\[([a-z]*)\]", body)
        final = position(replay(api, moves))
        assert hash_of(final, table) == int(expected), name
        games.append({"name": name, "moves": "".join(";%s[%s]" % m for m in moves), "zobrist_hash": int(expected), "final_position": final})
        print(name, len(moves), "moves: final position reproduces", expected)
    assert len(games) == 3
    with open(os.path.join(HERE, "real_games.json"), "w") as f:
        json.dump({"source": "src/dg_tests/tests/real_games.rs:27-117; positions verified against the expected hashes with the "
                             "reference's zobrist table at generation time (make_real_games.py)", "games": games}, f, indent=1)


if __name__ == "__main__":
    main()
