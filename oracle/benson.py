"""TEST INFRASTRUCTURE ONLY (like everything under oracle/): a pure-Python restatement of the reference's Benson pass
(src/libdg_go/utils/benson.rs:55-190) over ABSTRACT blocks and regions, the way the reference's own unit test drives it
(FakeAllBlocks / FakeAllRegions, :375-566), plus the extraction of blocks and regions from a position
(AllBlocksImpl :231-253, AllRegionsImpl :300-333).  It is pinned by the reference's Sensei's Library example (:571-599, the
expected `invalid` list) in tests/test_go_host.py and then used as the checker for go.cpp's implementation on random
positions.  Nothing in the product imports it."""

NONE, BLOCK, REGION = 0, 1, 2


def benson(blocks, regions):
    """blocks: [(points, liberties)], regions: [(points, neighbours)] with points as hashable ids -> {point: status}."""
    status = {}
    blocks = [(list(p), set(l)) for p, l in blocks]
    regions = [(list(p), list(n)) for p, n in regions]

    def mark(points, value):
        for p in points:
            status[p] = value

    def is_vital(region, block):                                    # :207-209
        return all(p in block[1] for p in region[0])

    for b in blocks:                                                # mark_all_blocks
        mark(b[0], BLOCK)
    for r in regions:                                               # mark_all_regions
        mark(r[0], REGION)
    kept = []                                                       # remove_non_vital_regions
    for r in regions:
        if any(is_vital(r, b) for b in blocks):
            kept.append(r)
        else:
            mark(r[0], NONE)
    regions = kept
    while True:
        changed = False
        kept = []                                                   # remove_non_alive_blocks
        for b in blocks:
            if sum(1 for r in regions if is_vital(r, b)) >= 2:
                kept.append(b)
            else:
                mark(b[0], NONE)
                changed = True
        blocks = kept
        kept = []                                                   # remove_non_surrounded_regions
        for r in regions:
            if all(status.get(p, NONE) == BLOCK for p in r[1]):
                kept.append(r)
            else:
                mark(r[0], NONE)
                changed = True
        regions = kept
        if not changed:
            return status


def _adjacent(p):
    x, y = p
    return [(u, v) for u, v in ((x + 1, y), (x, y - 1), (x - 1, y), (x, y + 1)) if 0 <= u < 19 and 0 <= v < 19]


def blocks_and_regions(stones, color):
    """stones: {(x, y): colour}; the blocks of `color` with every adjacent point as "liberty" (adjacencies_of yields points of
    any colour, board_fast.rs:154-159) and the regions reachable from an empty point without crossing `color`."""
    seen, blocks, regions = set(), [], []
    for y in range(19):
        for x in range(19):
            p = (x, y)
            if stones.get(p) != color or p in seen:
                continue
            points, adjacent, todo = [], set(), [p]
            seen.add(p)
            while todo:
                q = todo.pop()
                points.append(q)
                for r in _adjacent(q):
                    adjacent.add(r)
                    if stones.get(r) == color and r not in seen:
                        seen.add(r)
                        todo.append(r)
            blocks.append((points, adjacent))
    seen = set()
    for y in range(19):
        for x in range(19):
            p = (x, y)
            if p in stones or p in seen:
                continue
            points, neighbours, todo = [], [], [p]
            seen.add(p)
            while todo:
                q = todo.pop()
                points.append(q)
                for r in _adjacent(q):
                    if stones.get(r) == color:
                        neighbours.append(r)
                    elif r not in seen:
                        seen.add(r)
                        todo.append(r)
            if neighbours:
                regions.append((points, neighbours))
    return blocks, regions


def benson_of_position(stones, color):
    status = benson(*blocks_and_regions(stones, color))
    return [[status.get((x, y), NONE) for x in range(19)] for y in range(19)]
