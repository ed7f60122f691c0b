"""CPU tests of the compact board record (include/dgnn.h: dgnn_forward_boards): the record of a position, expanded by the HOST
statement of the device algorithm (go.cpp: record_to_packed), gives exactly the planes of features::V1
(src/libdg_go/utils/features.rs:99-251 as go.cpp's features_v1 states it) -- on every position of the three committed
reference games (tests/golden/real_games.json; src/dg_tests/tests/real_games.rs), all eight symmetries, both colours to move, and
on dense random positions with many captures, ataris and kos.  The -m gpu twin (tests/test_features_gpu.py) checks the DEVICE kernel
the same way."""
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN


def game_positions(api, stride=1):
    for game in json.load(open(os.path.join(GOLDEN, "real_games.json")))["games"]:
        board = api.Board(7.5)
        for n, (color, vertex) in enumerate(re.findall(r";([BW])\[([a-s]{2})\]", game["moves"])):
            board.place(0 if color == "B" else 1, ord(vertex[0]) - 97, ord(vertex[1]) - 97)
            if n % stride == 0:
                yield board


def random_positions(api, seed, games, stones, komi=0.5):
    """Dense random legal play: captures, ataris, snapbacks and kos in quantity (the three professional games hold few)."""
    rng = np.random.default_rng(seed)
    for _ in range(games):
        board, color = api.Board(komi), 0
        for _ in range(stones):
            for index in rng.permutation(361)[:60]:
                if board.is_valid(color, int(index % 19), int(index // 19)):
                    board.place(color, int(index % 19), int(index // 19))
                    break
            color ^= 1
            yield board


def expected_and_records(api, boards):
    features, records = [], []
    for board in boards:
        for color in (0, 1):
            for symmetry in range(8):
                features.append(board.get_features(color, symmetry).reshape(-1).view(np.uint16))
                records.append(board.record(color, symmetry))
    n = len(records)
    return api.features_pack(np.concatenate(features), n).reshape(n, api.PACKED_WORDS), np.concatenate(records), n


def describe_mismatch(expected, got):
    bad = np.argwhere(expected != got)
    i, w = bad[0]
    return (f"{len(bad)} words differ; first: variant {i} (colour {(i // 8) % 2}, symmetry {i % 8}) word {w}: "
            f"expected {expected[i, w]:#010x} got {got[i, w]:#010x} (planes {[b for b in range(32) if ((expected[i, w] ^ got[i, w]) >> b) & 1]})")


def test_record_planes_equal_features_v1_on_the_reference_games(api):
    total = 0
    for board in game_positions(api):
        expected, records, n = expected_and_records(api, [board])
        got = api.board_records_to_packed(records, n).reshape(n, api.PACKED_WORDS)
        assert np.array_equal(expected, got), f"after {board.count()} stones: " + describe_mismatch(expected, got)
        total += n
    assert total == 648 * 16


@pytest.mark.parametrize("komi", [0.5, 7.5, -7.5, -20.0])
def test_record_planes_equal_features_v1_on_dense_random_positions(api_full, komi):
    """komi -7.5 and below clamp the colour plane's constant to 0 (features.rs:236-239): plane 0 / 1 is then all zero."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    total, with_ko, with_ladder = 0, 0, 0
    for board in random_positions(api, seed=int(komi * 10) + 300, games=2, stones=330, komi=komi):
        if board.count() % 3:
            continue
        expected, records, n = expected_and_records(api, [board])
        got = api.board_records_to_packed(records, n).reshape(n, api.PACKED_WORDS)
        assert np.array_equal(expected, got), f"after {board.count()} stones: " + describe_mismatch(expected, got)
        with_ko += int(((expected[:, :361] >> 29) & 1).any())
        with_ladder += int(((expected[:, :361] >> 30) & 3).any())
        total += n
    print(f"komi {komi}: {total} variants, {with_ko} positions with a ko plane, {with_ladder} with a ladder plane")
    assert total > 2000 and with_ladder > 0


def test_record_layout_and_argument_checks(api):
    board = api.Board(6.5)
    board.place(0, 3, 3)
    board.place(1, 15, 15)
    board.place(0, 3, 4)
    r = board.record(1, 5)
    assert r.size == api.BOARD_RECORD_BYTES == 896
    cell = r[:736].view(np.uint16)
    assert cell[3 * 19 + 3] & 3 == 1 and cell[15 * 19 + 15] & 3 == 2 and cell[0] == 0
    assert cell[3 * 19 + 3] >> 2 == cell[4 * 19 + 3] >> 2                     # one block, one id
    assert r[880] == 1 and r[881] == 5                                        # to_move, symmetry
    assert r[884:888].view(np.int16).tolist() == [4 * 19 + 3, 15 * 19 + 15]   # the two most recent moves
    assert np.all(r[888:] == 0)
    bad = r.copy()
    bad[880] = 2
    with pytest.raises(api.Error):
        api.board_records_to_packed(bad, 1)
    bad = r.copy()
    bad[0] = 3                                                                # a cell that is neither empty, black nor white
    with pytest.raises(api.Error):
        api.board_records_to_packed(bad, 1)


def test_selfplay_through_a_pool_of_board_records_plays_the_same_games(api_full):
    """The search worker hands the pool 896-byte records instead of 23 KB of planes (search.cpp: Predictor::encode); a pool whose
    lanes are callbacks expands them with the host statement of the device kernel.  One search thread and a transposition table per
    game make a game a function of (seed, network outputs): the records pool must reproduce the games of the planes pool, move
    for move and record for record."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    kw = dict(num_games=4, num_moves=14, num_rollouts=60, threads_per_search=1, leaves_per_thread=4, seed=77, cache_per_game=4096,
              return_moves=True, return_sgf=True)
    out = {}
    for boards in (False, True):
        pool = api.LeafPool.over_callbacks(api.lib.dgnn_hash_predictor(), None, num_lanes=1, max_batch_size=32, boards=boards)
        assert pool.takes_boards == boards
        out[boards] = api.selfplay_go(pool=pool, **kw)
        stats = pool.stats()
        pool.close()
        assert stats["boards"] == out[boards]["evals"] > 0
    assert np.array_equal(out[False]["record"], out[True]["record"])
    strip = lambda sgf: re.sub(r"DT\[[^\]]*\]", "", sgf)          # the date stamp is the wall clock
    assert [strip(g) for g in out[False]["sgf"]] == [strip(g) for g in out[True]["sgf"]]


def test_pools_refuse_the_other_input_form(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    planes = api.LeafPool.over_callbacks(api.lib.dgnn_hash_predictor(), None, num_lanes=1, max_batch_size=8)
    records = api.LeafPool.over_callbacks(api.lib.dgnn_hash_predictor(), None, num_lanes=1, max_batch_size=8, boards=True)
    board = api.Board(7.5)
    board.place(0, 3, 3)
    with pytest.raises(api.Error):
        planes.predict_boards(board.record(1), 1)
    with pytest.raises(api.Error):
        records.predict(board.get_features(1).reshape(-1), 1)
    a = planes.predict(board.get_features(1, 3).reshape(-1), 1)[0]
    b = records.predict_boards(board.record(1, 3), 1)[0]
    assert a.value() == b.value() and np.array_equal(a.policy(), b.policy())
    value, prior, n = api.full_forward(board, 1, pool=records)
    value2, prior2, n2 = api.full_forward(board, 1, pool=planes)
    assert n == n2 == 8 and value == value2 and np.array_equal(prior, prior2)
    planes.close()
    records.close()


def test_games_as_fibers_play_the_games_of_one_thread_per_game(api_full):
    """csrc/fiber.h: with host_threads > 0 the games in flight are fibers on that many OS threads, switched in user space where a
    game waits for the leaf pool.  A game is a function of its seed and the network's outputs, so the records are those of the
    thread-per-game run -- with one, two and three scheduler threads, more games than fit in flight, and both pool forms."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    kw = dict(num_games=10, num_moves=10, num_rollouts=48, threads_per_search=1, leaves_per_thread=4, seed=91, cache_per_game=2048,
              num_parallel=6, return_moves=True, return_sgf=True)
    strip = lambda sgf: re.sub(r"DT\[[^\]]*\]", "", sgf)
    reference = None
    for boards in (True, False):
        for host_threads in (0, 1, 2, 3):
            pool = api.LeafPool.over_callbacks(api.lib.dgnn_hash_predictor(), None, num_lanes=1, max_batch_size=16, boards=boards)
            out = api.selfplay_go(pool=pool, host_threads=host_threads, **kw)
            pool.close()
            games = [strip(g) for g in out["sgf"]]
            if reference is None:
                reference = (out["record"].copy(), games, out["evals"])
            assert np.array_equal(out["record"], reference[0]) and games == reference[1] and out["evals"] == reference[2], (boards, host_threads)


def test_fibers_over_the_dry_device_and_a_shared_table(api):
    """256 games as fibers on two threads against the stand-in device (a batch takes base + per_board * n of wall time): the whole
    host pipeline of the bench's self-play leg without a GPU.  With a table shared between games the trajectories depend on
    timing; the totals must still add up."""
    pool = api.LeafPool.over_dry_device(base_us=200.0, per_board_us=2.0, max_batch_size=256)
    cache = api.LruCache(50_001, num_shards=4)
    out = api.selfplay_go(num_games=256, num_moves=2, num_rollouts=64, threads_per_search=1, leaves_per_thread=8, seed=5, pool=pool, cache=cache,
                          host_threads=2, return_moves=True)
    stats = pool.stats()
    pool.close()
    assert out["moves"] == 512 and (out["record"][:, :2] >= 0).all()
    assert stats["boards"] == out["evals"] and out["cache_hits"] > 0
    print(f"256 fibers / 2 threads: {out['evals']} evals in {stats['batches']} batches (mean {stats['boards'] / stats['batches']:.0f}), {out['cache_hits']} table hits")


def test_private_tables_on_a_coalescing_pool_are_refused(api):
    """a table per game promises games that are a function of their seed; a pool that lets requests follow positions in
    flight hands one game an evaluation another game asked for.  The C API refuses the combination instead of mixing silently."""
    pool = api.LeafPool.over_dry_device(base_us=0.0, per_board_us=0.0, max_batch_size=32, batches_per_lane=2, workers_per_lane=2)
    pool.set_coalescing(True)
    with pytest.raises(api.Error) as refused:
        api.selfplay_go(2, 4, 16, 1, seed=3, pool=pool, cache_per_game=2000)
    assert refused.value.status == api.BAD_ARGUMENT
    pool.set_coalescing(False)
    assert api.selfplay_go(2, 4, 16, 1, seed=3, pool=pool, cache_per_game=2000)["moves"] == 8
    pool.close()
