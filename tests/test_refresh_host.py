"""Weight refresh inside a running job, the host side: the leaf pool quiesces (dgnn_pool_pause),
the transposition table is emptied (dgnn_cache_clear), the pool resumes and every answer from then on comes from the new
weights.  Here with a Python predictor standing in for the network (its "weights" are a version number); the -m gpu twin swaps
real weight images (tests/test_failure_paths_gpu.py)."""
import threading
import time

import numpy as np
import pytest


def test_cache_clear_drops_every_entry(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    cache = api.LruCache(100, num_shards=4)
    policy = np.linspace(0, 1, 362).astype(np.float16)
    for h in range(40):
        cache.cache(1000 + h, h % 2, h % 8, api.Prediction(np.float16(0.25), policy))
    assert len(cache) == 40 and cache.fetch(1003, 1, 0) is not None
    cache.clear()
    assert len(cache) == 0 and cache.fetch(1003, 1, 0) is None
    cache.cache(1003, 1, 2, api.Prediction(np.float16(-0.5), policy))       # usable afterwards
    assert float(cache.fetch(1003, 1, 2).value()) == -0.5


@pytest.mark.parametrize("boards", [False, True])
def test_pause_refresh_resume_under_pool_traffic(api_full, boards):
    """Eight caller threads hammer the pool; the main thread pauses it, checks that nothing is being evaluated, swaps the "weights",
    resumes.  No caller may see an old-version answer that was REQUESTED after resume returned, no evaluation may overlap the
    swap, and callers that were waiting during the pause are answered by the new version."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    state = {"version": 1, "inside": 0, "overlap": False, "swapping": False}
    lock = threading.Lock()

    def predict(features, n):
        with lock:
            state["inside"] += 1
            if state["swapping"]:
                state["overlap"] = True
            version = state["version"]
        time.sleep(0.002)
        with lock:
            state["inside"] -= 1
        return np.full(n, np.float16(version / 8)), np.full(n * 362, np.float16(1 / 362))

    fn = api.python_predictor(predict)
    pool = api.LeafPool.over_callbacks(fn, None, num_lanes=1, max_batch_size=16, batches_per_lane=2, workers_per_lane=2, boards=boards)
    board = api.Board(7.5)
    board.place(0, 3, 3)
    item = board.record(1, 0) if boards else board.get_features(1, 0).reshape(-1)
    call = (lambda: pool.predict_boards(item, 1)) if boards else (lambda: pool.predict(item, 1))
    stop, seen = threading.Event(), []

    def caller():
        while not stop.is_set():
            t_request = time.perf_counter()
            seen.append((t_request, float(call()[0].value()) * 8))

    threads = [threading.Thread(target=caller) for _ in range(8)]
    [t.start() for t in threads]
    time.sleep(0.1)
    pool.pause()
    with lock:
        assert state["inside"] == 0                      # quiesced: no batch in flight
        state["swapping"] = True
    time.sleep(0.05)                                     # callers keep queueing meanwhile
    with lock:
        state["version"], state["swapping"] = 2, False
    pool.resume()
    t_resumed = time.perf_counter()
    time.sleep(0.1)
    stop.set()
    [t.join() for t in threads]
    pool.close()
    assert not state["overlap"]
    versions_before = {v for t, v in seen if t < t_resumed - 0.06}
    versions_after = {v for t, v in seen if t >= t_resumed}
    assert 1.0 in versions_before and versions_after == {2.0}, (versions_before, versions_after)
    assert len(seen) > 50


def test_an_answer_requested_before_a_clear_is_not_stored_after_it(api_full):
    """a request in flight across pause -> set weights -> clear -> resume used to put the OLD network's output back
    into the emptied table.  The table carries a generation: the miss is seen at one, the insert names it."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    cache = api.LruCache(100, num_shards=4)
    policy = np.linspace(0, 1, 362).astype(np.float16)
    seen = cache.generation()
    cache.clear()                                                      # the refresh happens while the evaluation is in flight
    assert cache.generation() == seen + 1
    cache.cache_at(seen, 77, 0, 3, api.Prediction(np.float16(0.125), policy))
    assert len(cache) == 0 and cache.fetch(77, 0, 3) is None           # dropped
    cache.cache_at(cache.generation(), 77, 0, 3, api.Prediction(np.float16(0.25), policy))
    assert float(cache.fetch(77, 0, 3).value()) == 0.25                # a miss seen after the clear is stored
    cache.cache(78, 1, 0, api.Prediction(np.float16(0.5), policy))     # the unconditional form (lru_cache.rs:121) still is
    assert len(cache) == 2


def test_the_search_does_not_cache_what_it_requested_before_a_refresh(api_full):
    """The search driver with a shared table: the predictor itself plays the refresh in the middle of call 3 (clears the table, bumps its
    "weights") and answers THAT call with the old version -- what a pool does to a request that was on the device during the pause.  No
    position may be left in the table with the old version's value once the search is over."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    board = api.Board(7.5)
    board.place(0, 3, 3)
    cache = api.LruCache(10_000, num_shards=2)
    state = {"calls": 0, "version": 1}

    def predict(features, n):
        state["calls"] += 1
        version = state["version"]
        if state["calls"] == 3:
            state["version"] = 2
            cache.clear()
        return np.full(n, np.float16(version / 8)), np.full(n * 362, np.float16(1 / 362))

    fn = api.python_predictor(predict)
    value, prior, _ = api.full_forward(board, 1, pool=fn, cache=None)
    root = api.Node(1, value, prior)
    evals, conflicts, hits = api.search_run(root, board, 48, num_threads=1, seed=5, pool=fn, cache=cache, leaves_per_thread=4)
    assert state["calls"] >= 4 and state["version"] == 2
    versions = []
    for y in range(19):
        for x in range(19):
            if not board.is_valid(1, x, y):
                continue
            child = board.clone()
            child.place(1, x, y)
            hit = cache.fetch(child.zobrist_hash(), 0, 0)
            if hit is not None:
                versions.append(float(hit.value()) * 8)
    assert len(versions) >= 8 and set(versions) == {2.0}, sorted(set(versions))


def test_games_with_a_table_of_their_own_drop_it_at_a_refresh(api):
    """cache_entries_per_game > 0: dgnn_cache_clear cannot reach a game's private table, so the game compares the pool's generation
    before every move.  Two games, policy play through a pool over a versioned predictor; the pool is paused / 'refreshed' / resumed
    while they run: after the refresh no move is played from a version-1 value (V[] of the record = the eight-symmetry value)."""
    import re
    state = {"version": 1}

    def predict(features, n):
        time.sleep(0.004)
        return np.full(n, np.float16(state["version"] / 4 - 0.5)), np.full(n * 362, np.float16(1 / 362))

    fn = api.python_predictor(predict)
    pool = api.LeafPool.over_callbacks(fn, None, num_lanes=1, max_batch_size=16, batches_per_lane=2, workers_per_lane=2)
    assert pool.generation() == 0
    result = {}

    def play():
        result["out"] = api.selfplay_go(2, 60, 1, threads_per_search=1, seed=9, pool=pool, cache=None, cache_per_game=500, return_sgf=True)

    t = threading.Thread(target=play)
    t.start()
    time.sleep(0.15)
    pool.pause()
    state["version"] = 2
    pool.resume()
    assert pool.generation() == 1
    t.join()
    pool.close()
    for sgf in result["out"]["sgf"]:
        values = [float(v) for v in re.findall(r"V\[([-+0-9.eE]+)\]", sgf)]
        assert len(values) >= 30
        changed = [i for i, v in enumerate(values) if abs(v) < 1e-3]                 # version 2 answers 0, version 1 -0.25
        old = [i for i, v in enumerate(values) if abs(abs(v) - 0.25) < 1e-3]
        mixed = [i for i in range(len(values)) if i not in changed and i not in old]   # the move whose eight symmetries straddled the pause
        assert changed and old, (values[:5], values[-5:])
        assert max(old) < min(changed)              # once a move has seen the new network, none after it is served version-1 entries
        assert all(max(old) < i < min(changed) for i in mixed), (mixed, max(old), min(changed))
