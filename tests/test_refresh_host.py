"""Weight refresh inside a running job, the host side: the leaf pool quiesces (dgnn_pool_pause),
the transposition table is emptied (dgnn_cache_clear), the pool resumes and every answer from then on comes from the new
weights.  Here with a Python predictor standing in for the network (its "weights" are a version number); the -m gpu twin swaps
real weight images (tests/test_failure_paths_gpu.py)."""
import threading
import time

import numpy as np
import pytest


def test_cache_clear_drops_every_entry(api):
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
def test_pause_refresh_resume_under_pool_traffic(api, boards):
    """Eight caller threads hammer the pool; the main thread pauses it, checks that nothing is being evaluated, swaps the "weights",
    resumes.  No caller may see an old-version answer that was REQUESTED after resume returned, no evaluation may overlap the
    swap, and callers that were waiting during the pause are answered by the new version."""
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
