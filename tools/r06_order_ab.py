"""A/B on one box: split launches ordered behind / in front of whole-board towers (DGNN_TOWER_ORDER, test build) or only among themselves.

    python tools/r06_order_ab.py            # runs itself twice (order on / off) and prints one JSON line per case

Cases: k 256-board submissions and one small one (20x256), submitted together from one thread, 200 rounds -- the shape of
tests/test_round6_gpu.py -- and the leaf pool's own mix through dgnn_forward_pipelined_timed_n is left to tools/bench_pool.py."""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api as shipped, synth
    api = shipped.load_full()
    u16p = ctypes.POINTER(ctypes.c_uint16)
    net = api.Network.from_json(synth.to_json(synth.make_tensors(20, 256, seed=6100)))
    for small, k in [(32, 1), (64, 1), (96, 1), (128, 1), (64, 2), (96, 2), (16, 2), (64, 0), (96, 0)]:
        f_big = [np.ascontiguousarray(synth.make_features(256, seed=6101 + i)).view(np.uint16).reshape(-1) for i in range(k)]
        copies = 1 if k else 3                      # k = 0: three small ones alive, nothing whole-board (the chain alone)
        f_small = np.ascontiguousarray(synth.make_features(small, seed=6110)).view(np.uint16).reshape(-1)
        ws_big = [net.get_workspace(256) for _ in range(k)]
        ws_small = [net.get_workspace(small) for _ in range(copies)]

        def run(ws, f, n):
            assert api.lib.dgnn_forward_submit_n(ws.handle, f.ctypes.data_as(u16p), n) == 0

        def wait(ws, n):
            v, p = np.empty(n, dtype=np.uint16), np.empty(n * 362, dtype=np.uint16)
            assert api.lib.dgnn_forward_wait(ws.handle, v.ctypes.data_as(u16p), p.ctypes.data_as(u16p)) == 0

        for w, f in zip(ws_big, f_big):
            run(w, f, 256); wait(w, 256)
        for w in ws_small:
            run(w, f_small, small); wait(w, small)
        rounds, t0 = 200, time.perf_counter()
        for _ in range(rounds):
            for w, f in zip(ws_big, f_big):
                run(w, f, 256)
            for w in ws_small:
                run(w, f_small, small)
            for w in ws_big:
                wait(w, 256)
            for w in ws_small:
                wait(w, small)
        seconds = time.perf_counter() - t0
        boards = rounds * (256 * k + small * copies)
        print(json.dumps({"order": os.environ.get("DGNN_TOWER_ORDER", "1"), "whole_board_towers": k, "small": small, "small_alive": copies,
                          "ms_per_round": round(1e3 * seconds / rounds, 3), "boards_per_s": round(boards / seconds), "towers": net.tower_stats()}), flush=True)
        for w in ws_big + ws_small:
            w.release()
    net.destroy()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for order in ("1", "0", "1", "0"):      # (DGNN_TOWER_ORDER was taken out of the library after this script had run: profiles/r06_final/order_ab.jsonl)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, DGNN_TOWER_ORDER=order), check=True)
