"""Host side of the boundary (scope row f-1) through the C ABI, no GPU needed: policy symmetries, the transposition
cache and the leaf batcher's queueing rules, pinned to the reference's own tests and arithmetic:

    src/libdg_mcts/predictor.rs:98-108     Prediction::with_transform identities
    src/libdg_go/utils/symmetry.rs:19-128  the eight tables and their inverses
    src/libdg_mcts/lru_cache.rs:152-183    insert_only_20000, mixed_insert
    src/libdg_mcts/pool/batch.rs:91-121    get_batch: newest max_batch_size entries, max_batches cap
"""
import threading
import time

import numpy as np
import pytest


def _reference_table(ax, ay):
    """get_transformation of symmetry.rs:19-37 restated on packed indices."""
    out = np.empty(362, dtype=np.int64)
    for i in range(361):
        x, y = i % 19 - 9, i // 19 - 9
        out[i] = (ay(x, y) + 9) * 19 + (ax(x, y) + 9)
    out[361] = 361
    return out


REFERENCE_TABLES = [                                   # symmetry.rs:39-62, order of symmetry::ALL (:119-128)
    _reference_table(lambda x, y: x, lambda x, y: y),      # Identity
    _reference_table(lambda x, y: -x, lambda x, y: y),     # FlipLR
    _reference_table(lambda x, y: x, lambda x, y: -y),     # FlipUD
    _reference_table(lambda x, y: y, lambda x, y: x),      # Transpose
    _reference_table(lambda x, y: -y, lambda x, y: -x),    # TransposeAnti
    _reference_table(lambda x, y: y, lambda x, y: -x),     # Rot90
    _reference_table(lambda x, y: -x, lambda x, y: -y),    # Rot180
    _reference_table(lambda x, y: -y, lambda x, y: x),     # Rot270
]


def test_symmetry_tables_and_inverses(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    T = api.Transform
    for t in T.ALL:
        got = np.array([T.apply(t, i) for i in range(362)])
        assert np.array_equal(got, REFERENCE_TABLES[t])
        inv = T.inverse(t)
        assert all(T.apply(inv, T.apply(t, i)) == i for i in range(362))
    assert [T.inverse(t) for t in T.ALL] == [0, 1, 2, 3, 4, 7, 6, 5]          # symmetry.rs:78-90
    assert T.apply(8, 0) == -1 and T.apply(0, 362) == -1 and T.inverse(-1) == -1


def test_check_with_transform(api_full):
    """predictor.rs:98-108, verbatim."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    original = api.Prediction(0.0, np.arange(362, dtype=np.float32))
    T = api.Transform
    assert original.policy()[0] == api.Prediction.with_transform(original, T.Identity).policy()[0]
    assert original.policy()[0] == api.Prediction.with_transform(original, T.Rot180).policy()[360]
    assert original.policy()[361] == api.Prediction.with_transform(original, T.Rot180).policy()[361]
    # every transform is a permutation of the 361 points that fixes the pass move
    for t in T.ALL:
        p = api.Prediction.with_transform(original, t).policy()
        assert sorted(p[:361].tolist()) == list(range(361)) and p[361] == 361
        assert np.array_equal(p[REFERENCE_TABLES[t]], original.policy())        # out[t(i)] = in[i]


def test_lru_insert_only_20000(api_full):
    """lru_cache.rs:157-165."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    lru = api.LruCache(capacity=1000, num_shards=1)
    p = api.Prediction(0.5, np.zeros(362))
    for i in range(20000):
        lru.cache(i, 0, 0, p)
    assert len(lru) == 1000


def test_lru_mixed_insert(api_full):
    """lru_cache.rs:167-182: a get refreshes an entry, the least recently used ones are dropped."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    lru = api.LruCache(capacity=10, num_shards=1)
    p = api.Prediction(0.25, np.zeros(362))
    for i in range(10):
        lru.cache(i, 0, 0, p)
    for i in range(2):
        assert lru.fetch(i, 0, 0) is not None
    for i in range(6):
        lru.cache(i + 20, 0, 0, p)
    for i in (0, 1, 8, 9):
        assert lru.fetch(i, 0, 0) is not None, i
    for i in range(2, 8):
        assert lru.fetch(i, 0, 0) is None, i


def test_cache_is_keyed_by_colour_and_canonicalises_the_symmetry(api_full):
    """nn.rs:69-82: cache() stores with_transform(response, symmetry.inverse()), fetch() returns with_transform(entry, symmetry)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    T = api.Transform
    lru = api.LruCache(capacity=100)
    rng = np.random.default_rng(3)
    response = api.Prediction(-0.75, rng.random(362).astype(np.float16))
    lru.cache(0xDEADBEEFCAFEF00D, 1, T.Rot90, response)
    assert lru.fetch(0xDEADBEEFCAFEF00D, 2, T.Rot90) is None                    # other colour to move
    same = lru.fetch(0xDEADBEEFCAFEF00D, 1, T.Rot90)
    assert same.value() == -0.75 and np.array_equal(same.policy(), response.policy())
    for s in T.ALL:
        got = lru.fetch(0xDEADBEEFCAFEF00D, 1, s)
        canonical = api.Prediction.with_transform(response, T.inverse(T.Rot90))
        assert np.array_equal(got.policy(), api.Prediction.with_transform(canonical, s).policy())
    # a present key is left untouched (lru_cache.rs:121)
    lru.cache(0xDEADBEEFCAFEF00D, 1, T.Identity, api.Prediction(0.1, np.zeros(362)))
    assert lru.fetch(0xDEADBEEFCAFEF00D, 1, T.Rot90).value() == -0.75
    assert lru.stats()["hits"] >= 9 and lru.stats()["misses"] == 1


def test_sharded_cache_keeps_its_capacity_under_threads(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    lru = api.LruCache(capacity=4096, num_shards=16)
    p = api.Prediction(0.0, np.zeros(362))

    def worker(base):
        for i in range(3000):
            lru.cache(base * 1_000_003 + i * 7919, i & 1, i & 7, p)
            lru.fetch(base * 1_000_003 + (i // 2) * 7919, (i // 2) & 1, 0)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert 0 < len(lru) <= 4096


def _board(tag):
    f = np.zeros(11552, dtype=np.float16)
    f[:8] = np.frombuffer(np.uint64(tag).tobytes(), dtype=np.uint8).astype(np.float16)
    return f


def test_batcher_hands_out_the_newest_entries(api_full):
    """batch.rs:91-121 with max_batch_size 4: the split index is size - max when size >= max, else 0."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Batcher(None, max_batch_size=4, max_batches=2)
    assert b.get_batch(1) is None                                               # nothing queued
    for tag in range(10):
        b.push(tag, _board(tag))
    assert b.get_batch(11) is None                                              # fewer than min_batch_size queued
    first = b.get_batch(4)
    assert first.tags().tolist() == [6, 7, 8, 9]                                # split_off(size - max): the NEWEST four
    assert np.array_equal(first.features().reshape(4, 11552)[:, :8], np.stack([_board(t)[:8] for t in (6, 7, 8, 9)]))
    assert b.pending() == (6, 1)
    second = b.get_batch(1)
    assert second.tags().tolist() == [2, 3, 4, 5]
    assert b.pending() == (2, 2)
    assert b.get_batch(1) is None                                               # max_batches alive (batch.rs:96-98)
    first.destroy()                                                             # Batch::forward decrements (batch.rs:37); so does a drop here
    third = b.get_batch(1)
    assert third.tags().tolist() == [0, 1]                                      # size < max: split index 0, everything
    assert b.pending() == (0, 2)
    with pytest.raises(api.Error):                                              # no predictor behind this batcher
        third.forward()


def test_push_and_get_batch_triggers_at_max_batch_size(api_full):
    """batch.rs:86-89: push_and_get_batch(event) == push + get_batch(max_batch_size)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Batcher(None, max_batch_size=3, max_batches=4)
    assert b.push_and_get_batch(100, _board(100)) is None
    assert b.push_and_get_batch(101, _board(101)) is None
    batch = b.push_and_get_batch(102, _board(102))
    assert batch is not None and batch.tags().tolist() == [100, 101, 102]
    assert b.pending() == (0, 1)


def test_batcher_conserves_every_event_under_threads(api_full):
    """Many game threads push, several workers drain: every tag comes out exactly once, no batch exceeds the maximum."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    b = api.Batcher(None, max_batch_size=16, max_batches=3)
    produced = 8 * 250
    seen, lock, done = [], threading.Lock(), threading.Event()

    def game(g):
        for i in range(250):
            b.push(g * 1000 + i, _board(g * 1000 + i))

    def drain():
        while True:
            batch = b.get_batch(1)
            if batch is None:
                if done.is_set() and b.pending()[0] == 0:
                    return
                continue
            assert 1 <= len(batch) <= 16
            tags = batch.tags().tolist()
            feats = batch.features().reshape(len(tags), 11552)
            for t, f in zip(tags, feats):
                assert np.array_equal(f[:8], _board(t)[:8])                     # features travel with their tag
            with lock:
                seen.extend(tags)
            batch.destroy()

    workers = [threading.Thread(target=drain) for _ in range(3)]
    games = [threading.Thread(target=game, args=(g,)) for g in range(8)]
    [t.start() for t in workers + games]
    [t.join() for t in games]
    done.set()
    [t.join() for t in workers]
    assert sorted(seen) == sorted(g * 1000 + i for g in range(8) for i in range(250))
    assert len(seen) == produced and b.pending() == (0, 0)


# ---- round 2: the pool over lanes (devices in production, predictor callbacks here), NnPredictor's process model

def test_pool_over_callback_lanes_coalesces_and_spreads(api):
    """dgnn_pool_create_callbacks: the multi-device pool's queueing with the deterministic CPU predictor in place of the
    devices.  Results equal direct predictor calls whatever lane evaluated them; every lane gets work under load; the
    counters add up (predictors/nn.rs:84-108: one queue, batches spread over Device::all())."""
    import ctypes
    import threading
    fn = api.lib.dgnn_hash_predictor()
    lanes, callers, per_call, rounds = 3, 12, 5, 20
    pool = api.LeafPool.over_callbacks(fn, None, num_lanes=lanes, max_batch_size=16, batches_per_lane=2, workers_per_lane=1)
    rng = np.random.default_rng(5)
    feats = (rng.random((callers, per_call, api.FEATURE_SIZE)) < 0.2).astype(np.float16)

    # the predictor called directly
    predict = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint16), ctypes.c_int,
                               ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint16))(fn)
    want = []
    for c in range(callers):
        v = np.empty(per_call, np.uint16); p = np.empty(per_call * api.NUM_MOVES, np.uint16)
        f = np.ascontiguousarray(feats[c]).view(np.uint16).reshape(-1)
        assert predict(None, f.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), per_call, v.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                       p.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))) == api.OK
        want.append((v.copy(), p.copy()))

    errors = []

    def caller(c):
        try:
            for _ in range(rounds):
                got = pool.predict(feats[c], per_call)
                for i, pr in enumerate(got):
                    assert np.asarray(pr._value).view(np.uint16) == want[c][0][i]
                    assert np.array_equal(pr._policy.view(np.uint16), want[c][1][i * api.NUM_MOVES:(i + 1) * api.NUM_MOVES])
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=caller, args=(c,)) for c in range(callers)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors[0]
    st = pool.stats()
    assert st["lanes"] == lanes and st["calls"] == callers * rounds
    assert st["boards"] == callers * rounds * per_call == sum(st["lane_boards"])
    assert st["batches"] == sum(st["lane_batches"]) <= st["boards"]
    assert st["batches"] < st["boards"]                      # boards of different callers shared device batches
    assert all(n > 0 for n in st["lane_batches"])            # every lane took part
    pool.close()


def test_pool_outlives_its_batcher_handle(api):
    """LeafPool held a raw Batcher*; destroying the batcher first left the workers polling freed memory.  The
    pool now holds a reference: either order is fine (the old order, batcher first, runs under the sanitizers in
    tools/sanitize_host.sh)."""
    fn = api.lib.dgnn_hash_predictor()
    pool = api.LeafPool.over_callbacks(fn, None, num_lanes=1, max_batch_size=4)
    feats = np.zeros(api.FEATURE_SIZE, np.float16)
    assert len(pool.predict(feats, 1)) == 1
    pool.close()
    pool.close()                                            # idempotent

    with pytest.raises(api.Error) as e:
        api.LeafPool.over_callbacks(None, None)
    assert e.value.status == api.BAD_ARGUMENT
    with pytest.raises(api.Error) as e:
        api.LeafPool.over_callbacks(fn, None, num_lanes=0)
    assert e.value.status == api.BAD_ARGUMENT


def test_idle_pool_workers_sleep(api):
    """idle workers woke every 200 us.  They now sleep on a condition variable until something is queued AND a
    batch slot is free: an idle pool burns (almost) no CPU, and a request is still served at once."""
    import time
    fn = api.lib.dgnn_hash_predictor()
    pool = api.LeafPool.over_callbacks(fn, None, num_lanes=2, max_batch_size=8, workers_per_lane=2)
    feats = np.zeros(api.FEATURE_SIZE, np.float16)
    pool.predict(feats, 1)
    t0, c0 = time.perf_counter(), time.process_time()
    time.sleep(0.5)
    cpu = time.process_time() - c0
    wall = time.perf_counter() - t0
    assert cpu < 0.05 * wall, (cpu, wall)                    # four workers polling every 200 us cost about 10 % of a core each
    t1 = time.perf_counter()
    pool.predict(feats, 1)
    assert time.perf_counter() - t1 < 0.015                  # woken by the push, not by the 20 ms safety timeout
    pool.close()


def test_adaptive_batches_alive_follow_the_unique_positions_per_second(api_full):
    """Round 4: between 2 and 3 batches alive per lane, chosen by the pool from what it measures.  Here over the
    CPU stand-in of a device: the rule can only be switched on for a pool of board records that was created with enough batches and
    workers; it counts repeated positions (same stones, same colour to move, ANY symmetry) and never leaves [lo, hi]."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    pool = api.LeafPool.over_dry_device(base_us=50.0, per_board_us=0.5, max_batch_size=64, batches_per_lane=3, workers_per_lane=3)
    import pytest
    with pytest.raises(api.Error):
        pool.set_adaptive_batches(2, 4)                      # more than the pool has workers for
    with pytest.raises(api.Error):
        pool.set_adaptive_batches(0, 3)
    pool.set_adaptive_batches(2, 3)
    assert pool.adaptive_stats()["alive_per_lane"] == 2
    board = api.Board(7.5)
    board.place(0, 3, 3)
    same = np.concatenate([board.record(1, s) for s in range(8)])               # one position, eight symmetries
    other = api.Board(7.5)
    other.place(0, 15, 15)
    records = np.concatenate([same, other.record(1, 0)])
    for _ in range(8000):                                                       # 72 000 boards: two windows of 32 768
        pool.predict_boards(records, 9)
    stats = pool.adaptive_stats()
    assert stats["boards_hashed"] == 72000 and stats["boards_repeated"] >= 71990        # everything but the first sighting of each
    # (the two first sightings fell into the first window, measured under `lo`; the second window saw repeats only)
    assert stats["alive_per_lane"] in (2, 3) and stats["unique_per_sec_lo"] > 0 and stats["unique_per_sec_hi"] == 0
    pool.set_adaptive_batches(0, 0)
    assert pool.adaptive_stats()["alive_per_lane"] == 3
    pool.close()
    features_pool = api.LeafPool.over_callbacks(api.lib.dgnn_hash_predictor(), None, num_lanes=1, max_batch_size=8, batches_per_lane=3, workers_per_lane=3)
    with pytest.raises(api.Error):
        features_pool.set_adaptive_batches(2, 3)             # fp16 planes are not hashed: records only
    features_pool.close()


def test_requests_for_a_position_in_flight_follow_it(api_full):
    """Round 4: dgnn_pool_set_coalescing.  A slow stand-in device; one thread asks for a position in symmetry 0, seven more ask for the
    SAME position in the other symmetries while it is out: ONE board reaches the device, and every follower receives the leader's
    answer re-oriented to its own symmetry -- value as is, policy = with_transform(with_transform(p, inverse(leader)), own): what
    NnPredictor::fetch returns for a table hit (predictors/nn.rs:72-81).  A different position is evaluated on its own."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    import threading
    import time
    board = api.Board(7.5)
    for i, (x, y) in enumerate([(3, 3), (15, 15), (3, 15), (16, 3), (9, 10)]):
        board.place(i % 2, x, y)
    other = board.clone()
    other.place(1, 10, 10)
    plain = api.LeafPool.over_dry_device(base_us=0.0, per_board_us=0.0, max_batch_size=16, batches_per_lane=1, workers_per_lane=1)
    direct = plain.predict_boards(board.record(1, 0), 1)                       # what the stand-in answers for the leader's record
    direct_other = plain.predict_boards(other.record(0, 5), 1)
    plain.close()
    pool = api.LeafPool.over_dry_device(base_us=150000.0, per_board_us=0.0, max_batch_size=16, batches_per_lane=1, workers_per_lane=1)
    pool.set_coalescing(True)
    answers = {}

    def ask(symmetry, rec):
        answers[symmetry] = pool.predict_boards(rec, 1)

    leader = threading.Thread(target=ask, args=(0, board.record(1, 0)))
    leader.start()
    time.sleep(0.03)                                                           # the leader is on the "device" (150 ms) by now
    rest = [threading.Thread(target=ask, args=(s, board.record(1, s))) for s in range(1, 8)]
    rest.append(threading.Thread(target=ask, args=("other", other.record(0, 5))))
    [t.start() for t in rest]
    [t.join() for t in rest + [leader]]
    assert pool.coalesced() == 7 and pool.stats()["boards"] == 2               # the position once, the other position once
    bits = lambda prediction: prediction._policy.view(np.uint16)              # noqa: E731  (the 362 fp16 bit patterns)
    for s in range(8):                                                         # leader symmetry 0 = the canonical orientation
        got = answers[s][0]
        assert float(got.value()) == float(direct[0].value())
        want = api.Prediction.with_transform(direct[0], s)
        assert np.array_equal(bits(got), bits(want)), s
    assert np.array_equal(bits(answers["other"][0]), bits(direct_other[0]))
    # nothing stays behind: the same position asked again is evaluated again (the table, not the pool, remembers answers)
    again = pool.predict_boards(board.record(1, 3), 1)
    assert pool.stats()["boards"] == 3 and pool.coalesced() == 7 and again is not None
    # the eight symmetries of ONE request are eight evaluations (the root ensemble of lib.rs:83-130), never followers of each other
    ensemble = pool.predict_boards(np.concatenate([board.record(1, s) for s in range(8)]), 8)
    assert pool.stats()["boards"] == 11 and pool.coalesced() == 7 and len(ensemble) == 8
    pool.close()
    features_pool = api.LeafPool.over_callbacks(api.lib.dgnn_hash_predictor(), None, num_lanes=1, max_batch_size=8)
    import pytest
    with pytest.raises(api.Error):
        features_pool.set_coalescing(True)
    features_pool.close()


def test_a_third_batch_only_while_enough_boards_wait(api_full):
    """dgnn_pool_set_extra_batch_queue: two batches alive per lane, a third only while at least `min_queued` boards are waiting."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    import pytest
    pool = api.LeafPool.over_dry_device(base_us=0.0, per_board_us=0.0, max_batch_size=64, batches_per_lane=3, workers_per_lane=3)
    with pytest.raises(api.Error):
        pool.set_extra_batch_queue(2, 4, 64)                 # more than the pool has workers for
    with pytest.raises(api.Error):
        pool.set_extra_batch_queue(0, 3, 64)
    pool.set_extra_batch_queue(2, 3, 64)
    assert pool.adaptive_stats()["alive_per_lane"] == 2     # nothing waits: the reference's rule
    board = api.Board(7.5)
    board.place(0, 3, 3)
    for _ in range(50):
        assert len(pool.predict_boards(np.concatenate([board.record(1, s) for s in range(8)]), 8)) == 8
    pool.set_extra_batch_queue(0, 0, 0)
    assert pool.adaptive_stats()["alive_per_lane"] == 3     # back to the constant the pool was created with
    pool.close()


def test_the_table_behaves_like_a_textbook_lru_under_random_traffic(api_full):
    """The arena / open-addressing form of the table (batcher.cpp CacheShard) against an OrderedDict that restates
    lru_cache.rs:95-147 -- get moves an entry to the front, an insert of a present key changes nothing, an insert into a full
    table drops the least recently used entry -- over 60 000 random operations on few enough keys that buckets collide, entries
    are evicted and re-inserted, and deletions shift probe chains; then a clear, and the same again on the re-used arena."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    from collections import OrderedDict
    rng = np.random.default_rng(11)
    for capacity, keys in ((7, 40), (64, 400), (500, 1200)):
        lru = api.LruCache(capacity=capacity, num_shards=1)
        for _round in range(2):
            model = OrderedDict()
            for _ in range(15000 if capacity < 500 else 30000):
                key, colour = int(rng.integers(keys)) * 0x9E3779B97F4A7C15 % (1 << 64), int(rng.integers(2))
                if rng.random() < 0.5:
                    got = lru.fetch(key, colour, 0)
                    want = model.get((key, colour))
                    assert (got is None) == (want is None)
                    if want is not None:
                        assert got.value() == want
                        model.move_to_end((key, colour))
                else:
                    value = float(np.float16(rng.integers(-64, 64) / 64))
                    lru.cache(key, colour, 0, api.Prediction(value, np.zeros(362)))
                    if (key, colour) not in model:
                        model[(key, colour)] = value
                        if len(model) > capacity:
                            model.popitem(last=False)
                assert len(lru) == len(model)
            # every survivor is there with its value, nothing else is (checked from the oldest: the fetches re-order both alike)
            for (key, colour), value in list(model.items()):
                assert lru.fetch(key, colour, 0).value() == value
            lru.clear()
            assert len(lru) == 0 and lru.fetch(next(iter(model))[0], 0, 0) is None and lru.fetch(next(iter(model))[0], 1, 0) is None


def test_back_to_back_holds_partial_batches_until_the_device_is_about_to_be_free(api):
    """dgnn_pool_set_back_to_back over the CPU stand-in of a device that takes 3 ms per batch: sixteen callers with four boards each.
    With the reference's rule a slot that frees takes the few boards that are queued at that moment; back to back a partial batch
    waits until shortly before the running batch is due and leaves with everything that has arrived: fewer, fuller batches, the
    same answers (those of the stand-in called directly), nobody left behind.  With snapping no batch has 33 .. 47 boards."""
    import pytest
    boards = []
    for k in range(64):
        b = api.Board(7.5)
        b.place(0, k % 19, k // 19)
        b.place(1, 18 - k % 19, 18 - k // 19)
        boards.append(b.record(0, k % 8))
    direct = api.LeafPool.over_dry_device(base_us=0.0, per_board_us=0.0, max_batch_size=64, batches_per_lane=1, workers_per_lane=1)
    want = [direct.predict_boards(r, 1)[0] for r in boards]
    direct.close()

    def drive(pool, seconds=0.6):
        stop, errors = threading.Event(), []

        def caller(t):
            mine = list(range(4 * t, 4 * t + 4))
            records = np.concatenate([boards[i] for i in mine])
            while not stop.is_set():
                got = pool.predict_boards(records, 4)
                for j, i in enumerate(mine):
                    if got[j].value() != want[i].value() or not np.array_equal(got[j].policy(), want[i].policy()):
                        errors.append(i)
        threads = [threading.Thread(target=caller, args=(t,)) for t in range(16)]
        [t.start() for t in threads]
        time.sleep(seconds)
        stop.set()
        [t.join() for t in threads]
        assert not errors
        s = pool.stats()
        return s["boards"] / max(s["batches"], 1), s["boards"]

    plain = api.LeafPool.over_dry_device(base_us=3000.0, per_board_us=0.0, max_batch_size=64, batches_per_lane=2, workers_per_lane=2)
    plain_mean, plain_boards = drive(plain)
    plain.close()
    pool = api.LeafPool.over_dry_device(base_us=3000.0, per_board_us=0.0, max_batch_size=64, batches_per_lane=2, workers_per_lane=2)
    with pytest.raises(api.Error):
        pool.set_back_to_back(300, 32, 32)                   # an empty snapping range
    pool.set_back_to_back(300, 32, 48)
    mean, total = drive(pool)
    print(f"mean batch {plain_mean:.1f} -> {mean:.1f} boards, {plain_boards} -> {total} boards evaluated in 0.6 s")
    assert mean > 1.15 * plain_mean and total > 0.8 * plain_boards          # (measured: 22.5 -> 31.5 boards, 4 300 -> 6 100 in 0.6 s; the same under full CPU load)
    times = pool.batch_times()                               # the pool's own clock: a batch of the stand-in takes 3 ms (+ the wait for its turn)
    assert times and all(3000 <= t <= 20000 for t in times.values()) and max(times) <= 64
    pool.set_back_to_back(0, 0, 0)
    drive(pool, 0.1)
    pool.close()
    lanes = api.LeafPool.over_callbacks(api.lib.dgnn_hash_predictor(), None, num_lanes=2, max_batch_size=8)
    with pytest.raises(api.Error):
        lanes.set_back_to_back(300)                          # one device, one clock
    lanes.close()


def test_a_leader_that_could_not_be_queued_leaves_no_position_behind(api_full):
    """with coalescing a slot registers as the LEADER of its position before it is pushed into the queue.  A push that
    fails (an allocation) used to leave the entry: every later request for the position followed a leader nobody would ever answer
    and blocked for ever.  The hook makes one push fail like a failed allocation: the call reports it, the boards of the request
    that WERE queued are still evaluated, and the position can be asked for again."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    import threading
    board = api.Board(7.5)
    for i, (x, y) in enumerate([(3, 3), (15, 15), (3, 15), (16, 3)]):
        board.place(i % 2, x, y)
    others = []
    for k in range(3):
        b = board.clone()
        b.place(0, 9 + k, 9)
        others.append(b)
    pool = api.LeafPool.over_dry_device(base_us=2000.0, per_board_us=0.0, max_batch_size=16, batches_per_lane=1, workers_per_lane=1)
    pool.set_coalescing(True)
    try:
        api.lib.dgnn_test_fail_push_after(0)                       # the very next push
        with pytest.raises(api.Error) as failed:
            pool.predict_boards(board.record(0, 0), 1)
        assert failed.value.status == api.OUT_OF_MEMORY
        # three boards, the THIRD push fails: the call reports the failure after the two queued boards have come back
        api.lib.dgnn_test_fail_push_after(2)
        before = pool.stats()["boards"]
        with pytest.raises(api.Error):
            pool.predict_boards(np.concatenate([b.record(1, 0) for b in others]), 3)
        assert pool.stats()["boards"] - before == 2
    finally:
        api.lib.dgnn_test_fail_push_after(-1)
    # every position involved can be asked for again and is EVALUATED (a stale leader entry would make these calls wait for ever)
    answers = []
    worker = threading.Thread(target=lambda: answers.extend([pool.predict_boards(board.record(0, 3), 1), pool.predict_boards(others[2].record(1, 1), 1)]))
    worker.start()
    worker.join(timeout=20.0)
    assert not worker.is_alive() and len(answers) == 2 and pool.coalesced() == 0
    pool.close()
