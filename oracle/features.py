"""CPU restatement of the reference's input contract: the 32 feature planes of `features::V1` -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` and the fixture generator ``tests/golden/make_planes.py`` import this module, and only as the checker of the two
product statements of the planes (``csrc/go.cpp: features_v1`` on the host, ``csrc/features.hip: board_records_kernel`` on the
device).  Nothing under ``dream-go_amd/`` imports it, and it shares no code, table or data structure with them: the product keeps
incremental blocks, bit boards and a zobrist ring; this file keeps a list of 361 colours, recomputes every block by FLOOD FILL
whenever it is asked about one, simulates "liberties after a move" by playing the move on a copy, and decides super-ko by
comparing whole positions.  Slow (about half a second per position and colour) and meant to be.

What it restates (paths relative to the reference checkout, kblomdahl/dream-go):
  src/libdg_go/utils/features.rs:99-251      V1::get_features (plane order, HWC index, validity gating, the komi constant)
  src/libdg_go/board.rs:120-176              Board::_is_ko / is_valid / _place (history of moves, ring of the last 16 positions)
  src/libdg_go/board_fast.rs:169-241         get_n_liberty, has_n_liberty, is_valid (no suicide)
  src/libdg_go/board_fast.rs:430-541         place (captures; own stones are never cleared), get_n_liberty_if
  src/libdg_go/utils/ladder.rs:26-170        _can_escape_with_capture, _is_ladder_capture, is_ladder_capture, is_ladder_escape
  src/libdg_go/utils/symmetry.rs:19-62       the eight transformations
  src/libdg_go/iter/adjacent_iter.rs:38-57   neighbour order (+x, -y, -x, +y): the ladder reader takes the FIRST group it finds
  src/libdg_go/circular_buf.rs, small_set.rs the two history containers (8 most recent moves; 16 most recent position hashes,
                                             initially sixteen zeros = sixteen empty boards)

PINNING: the reference holds no expected plane vector (features.rs:476-492 assert lengths only).  What pins this restatement to
the reference: its own unit tests of the parts, restated in tests/test_features_oracle.py -- `check_get_n_liberty_if`
(board_fast.rs:545-560), `ladder_corner_capture`, `ladder_capture` and the other ladder cases (ladder.rs:173-351) -- and the
super-ko positions of the three `real_games.rs` games (their final zobrist hashes are reproduced by the product's replay, the
stone configurations of which this file's replay must equal move for move).
"""
import numpy as np

EMPTY, BLACK, WHITE = 0, 1, 2
SIZE = 19
NUM_POINTS = 361
NUM_FEATURES = 32
PASS = -1

# src/libdg_go/iter/adjacent_iter.rs:42-43
_DX = (1, 0, -1, 0)
_DY = (0, -1, 0, 1)


def _neighbours_of(p):
    x, y = p % SIZE, p // SIZE
    out = []
    for dx, dy in zip(_DX, _DY):
        nx, ny = x + dx, y + dy
        if 0 <= nx < SIZE and 0 <= ny < SIZE:      # ValidIter: off-board vertices are skipped (board_fast.rs:115-120)
            out.append(ny * SIZE + nx)
    return tuple(out)


NEIGHBOURS = tuple(_neighbours_of(p) for p in range(NUM_POINTS))


def opposite(color):
    return BLACK + WHITE - color


class Stones:
    """BoardFast without its bookkeeping: the colours, nothing else.  Blocks and liberties are recomputed on demand."""

    def __init__(self, colors=None):
        self.c = list(colors) if colors is not None else [EMPTY] * NUM_POINTS

    def copy(self):
        return Stones(self.c)

    def block(self, p):
        """(stones, liberties) of the block at p, by flood fill."""
        color = self.c[p]
        assert color != EMPTY
        stones, liberties, stack = {p}, set(), [p]
        while stack:
            q = stack.pop()
            for n in NEIGHBOURS[q]:
                if self.c[n] == EMPTY:
                    liberties.add(n)
                elif self.c[n] == color and n not in stones:
                    stones.add(n)
                    stack.append(n)
        return stones, liberties

    def num_liberties(self, p):                     # get_n_liberty, board_fast.rs:169-173 (the incremental count IS the number of
        return len(self.block(p)[1])                # distinct liberties: join_blocks / capture only ever add points that are new)

    def has_n_liberty(self, p, n):                  # board_fast.rs:202-206
        return self.num_liberties(p) >= n

    def is_valid(self, color, p):
        """board_fast.rs:213-241: empty, and not suicide -- a direct liberty, a friendly neighbour with at least two liberties, or
        an enemy neighbour with fewer than two."""
        if self.c[p] != EMPTY:
            return False
        for n in NEIGHBOURS[p]:
            if self.c[n] == EMPTY:
                return True
            if (self.c[n] == color) == self.has_n_liberty(n, 2):
                return True
        return False

    def place(self, color, p):
        """board_fast.rs:430-471: the stone goes down whether legal or not; enemy blocks without liberties are taken off; one's own
        blocks never are."""
        self.c[p] = color
        for n in NEIGHBOURS[p]:
            if self.c[n] == opposite(color):
                stones, liberties = self.block(n)
                if not liberties:
                    for s in stones:
                        self.c[s] = EMPTY

    def num_liberties_if(self, color, p):
        """get_n_liberty_if (board_fast.rs:484-541), as what it MEANS: the liberties of the block the stone would belong to once it
        has been played and its captures taken off -- computed by doing exactly that on a copy."""
        other = self.copy()
        other.place(color, p)
        return other.num_liberties(p)

    def a_liberty(self, p):                         # get_a_liberty, board_fast.rs:182-192 (only ever asked of a block in atari:
        liberties = self.block(p)[1]                # the one liberty it has)
        return min(liberties) if liberties else None


# ---- ladders (src/libdg_go/utils/ladder.rs) ----------------------------------------------------------------------------------------

def _can_escape_with_capture(stones, color, p):
    """ladder.rs:26-34: some stone of the block at p touches an enemy block that is itself in atari."""
    for s in stones.block(p)[0]:
        for n in NEIGHBOURS[s]:
            if stones.c[n] == opposite(color) and not stones.has_n_liberty(n, 2):
                return True
    return False


def _is_ladder_capture(stones, color, p):
    """ladder.rs:36-103.  `stones` is consumed."""
    stones.place(color, p)
    enemy = opposite(color)
    # the FIRST neighbouring enemy block (neighbour order!) that is now in atari, cannot counter-capture, and whose one liberty is
    # a legal move for it: it extends there
    extend = None
    for n in NEIGHBOURS[p]:
        if stones.c[n] == enemy and not stones.has_n_liberty(n, 2) and not _can_escape_with_capture(stones, enemy, n):
            liberty = stones.a_liberty(n)
            if liberty is not None and stones.is_valid(enemy, liberty):
                extend = liberty
                break
    if extend is None:
        return False
    stones.place(enemy, extend)
    if not stones.has_n_liberty(extend, 2):         # one liberty: captured
        return True
    if stones.has_n_liberty(extend, 3):             # more than two: got away
        return False
    # the extension put one of OUR stones into atari: not a ladder
    for n in NEIGHBOURS[extend]:
        if stones.c[n] == color and not stones.has_n_liberty(n, 2):
            return False
    # keep chasing from any neighbour of the extension (ladder.rs:96-102: its neighbours, not the block's liberties)
    for n in NEIGHBOURS[extend]:
        if stones.is_valid(color, n) and _is_ladder_capture(stones.copy(), color, n):
            return True
    return False


def is_ladder_capture(stones, color, p):            # ladder.rs:106-110
    return _is_ladder_capture(stones.copy(), color, p)


def is_ladder_escape(stones, color, p):
    """ladder.rs:112-146: p connects to a block of ours that is in atari, the result has exactly two liberties, and the enemy has
    no ladder capture at any NEIGHBOUR of p."""
    if not any(stones.c[n] == color and not stones.has_n_liberty(n, 2) for n in NEIGHBOURS[p]):
        return False
    after = stones.copy()
    after.place(color, p)
    if after.num_liberties(p) != 2:
        return False
    enemy = opposite(color)
    for n in NEIGHBOURS[p]:
        if after.is_valid(enemy, n) and _is_ladder_capture(after.copy(), enemy, n):
            return False
    return True


# ---- the board with its history (src/libdg_go/board.rs) ----------------------------------------------------------------------------

class Board:
    def __init__(self, komi=7.5):
        self.stones = Stones()
        self.komi = float(komi)
        self.visited = [False] * NUM_POINTS         # point_state `visited`: a stone has stood here (board_fast.rs:446)
        self.history = []                           # most recent move first (CircularBuf, eight deep; V1 looks at two)
        # SmallSet64: the hashes of the 16 most recent positions, initially sixteen zeros -- the hash of the EMPTY board
        self.ring = [tuple([EMPTY] * NUM_POINTS)] * 16
        self.ring_at = 0

    def place(self, color, p):                      # board.rs:151-176
        self.stones.place(color, p)
        self.visited[p] = True
        self.history.insert(0, p)
        del self.history[8:]
        self.ring = list(self.ring)
        self.ring[self.ring_at] = tuple(self.stones.c)
        self.ring_at = (self.ring_at + 1) % 16

    def is_ko(self, color, p):
        """board.rs:120-130: only where a stone has stood before, and only if the position after the move (captures included) is
        one of the sixteen remembered ones.  The reference compares zobrist hashes of positions; this compares the positions."""
        if not self.visited[p]:
            return False
        after = self.stones.copy()
        after.place(color, p)
        return tuple(after.c) in self.ring


# ---- symmetries (src/libdg_go/utils/symmetry.rs:19-62) -----------------------------------------------------------------------------

_TRANSFORMS = (
    (lambda x, y: x, lambda x, y: y),               # Identity
    (lambda x, y: -x, lambda x, y: y),              # FlipLR
    (lambda x, y: x, lambda x, y: -y),              # FlipUD
    (lambda x, y: y, lambda x, y: x),               # Transpose
    (lambda x, y: -y, lambda x, y: -x),             # TransposeAnti
    (lambda x, y: y, lambda x, y: -x),              # Rot90
    (lambda x, y: -x, lambda x, y: -y),             # Rot180
    (lambda x, y: -y, lambda x, y: x),              # Rot270
)


def symmetry_table(symmetry):
    ax, ay = _TRANSFORMS[symmetry]
    table = []
    for p in range(NUM_POINTS):
        x, y = p % SIZE - 9, p // SIZE - 9
        table.append((ay(x, y) + 9) * SIZE + (ax(x, y) + 9))
    return table


# ---- the planes (src/libdg_go/utils/features.rs:99-251) ----------------------------------------------------------------------------

def plane_masks(board, to_move):
    """-> (mask [361] uint32 with bit c set where plane c is non-zero, fp16 bits of the komi constant), identity orientation."""
    stones, enemy = board.stones, opposite(to_move)
    mask = [0] * NUM_POINTS
    for i, p in enumerate(board.history[:2]):                            # :173-180 planes 3, 4: the two most recent moves
        if p != PASS:
            mask[p] |= 1 << (3 + i)
    any_ko = False
    for p in range(NUM_POINTS):
        if stones.c[p] != EMPTY:                                         # :186-192 planes 5..10 ours / 17..22 theirs: liberties >= 1..6
            start = 5 if stones.c[p] == to_move else 17
            for i in range(min(stones.num_liberties(p), 6)):
                mask[p] |= 1 << (start + i)
            continue
        if stones.is_valid(to_move, p):                                  # :194-200 planes 11..16: our liberties after the move
            for i in range(min(stones.num_liberties_if(to_move, p), 6)):
                mask[p] |= 1 << (11 + i)
            if board.is_ko(to_move, p):                                  # :218-224 plane 29 (and the board-wide plane 2)
                any_ko = True
                mask[p] |= 1 << 29
            if is_ladder_capture(stones, to_move, p):                    # :226-229 plane 30
                mask[p] |= 1 << 30
            if is_ladder_escape(stones, to_move, p):                     # :231-234 plane 31
                mask[p] |= 1 << 31
        if stones.is_valid(enemy, p):                                    # :202-208 planes 23..28: theirs after THEIR move
            for i in range(min(stones.num_liberties_if(enemy, p), 6)):
                mask[p] |= 1 << (23 + i)
    board_wide = (1 << (0 if to_move == BLACK else 1)) | ((1 << 2) if any_ko else 0)     # :236-250
    komi = np.float32(0.5) + (np.float32(0.5) * np.float32(board.komi)) / np.float32(7.5)
    komi = np.float16(min(max(komi, np.float32(0.0)), np.float32(1.0)))
    if komi == 0:
        board_wide &= ~3                                                 # a plane "filled with" the constant 0 is empty
    return np.array([m | board_wide for m in mask], dtype=np.uint64).astype(np.uint32), int(komi.view(np.uint16))


def apply_symmetry(mask, symmetry):
    """features[o.index(c, symmetry_table[point])] = value at point (features.rs:176,184)."""
    out = np.zeros_like(mask)
    out[np.array(symmetry_table(symmetry))] = mask
    return out


def features_v1(board, to_move, symmetry=0):
    """The reference's tensor: fp16 bit patterns [361 * 32], HWC (index = 32 * (19 y + x) + c, features.rs:52-60)."""
    mask, komi_bits = plane_masks(board, to_move)
    mask = apply_symmetry(mask, symmetry)
    bits = ((mask[:, None] >> np.arange(NUM_FEATURES, dtype=np.uint32)[None, :]) & 1).astype(np.uint16)
    out = bits * np.uint16(0x3c00)
    colour_plane = 0 if to_move == BLACK else 1
    out[:, colour_plane] = bits[:, colour_plane] * np.uint16(komi_bits)
    return out.reshape(-1)


def replay(moves_text, komi=7.5):
    """Positions of an SGF move list like ';B[pd];W[dp]...' (the form tests/golden/real_games.json holds), replayed the way the
    reference's test harness does (src/dg_tests/tests/common/mod.rs:37-77): vertex 'xy' -> Point::new(x, 18 - y), empty or
    out-of-range vertices are passes and skipped.  Yields (board, colour that just played, point) after every stone."""
    import re
    board = Board(komi)
    for color, vertex in re.findall(r";([BW])\[([a-z]*)\]", moves_text):
        if len(vertex) != 2:
            continue
        x, y = ord(vertex[0]) - 97, ord(vertex[1]) - 97
        if x >= SIZE or y >= SIZE:
            continue
        c = BLACK if color == "B" else WHITE
        p = (18 - y) * SIZE + x
        assert board.stones.is_valid(c, p) and not board.is_ko(c, p), (color, vertex)      # common/mod.rs: every move must be valid
        board.place(c, p)
        yield board, c, p


def position_text(board):
    """'.', 'X' (black), 'O' (white) per point, row by row: the form of `final_position` in tests/golden/real_games.json."""
    return "".join(".XO"[c] for c in board.stones.c)
