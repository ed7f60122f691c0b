"""The AVX2 / F16C forms of the host's hot loops (PUCT, arg-max, finite sum, fp16 -> fp32, stone sets, weak blocks) against their
scalar forms, bit for bit: the same seeded self-play job -- whole searches on real boards, candidate rules, Benson,
feature planes -- in a child process with DGNN_HOST_SIMD=0 (read once per process) and in one with the default, records compared as
text.  On a host without AVX2 both children run the scalar forms and the test says nothing (it still passes)."""
import json
import os
import subprocess
import sys

from conftest import ROOT

CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import __graft_entry__ as graft
graft.package()
from dream_go_amd import api
out = api.selfplay_go(3, 40, 48, threads_per_search=1, seed=31, pool=None, cache=None, cache_per_game=2000, leaves_per_thread=4,
                      return_moves=True, return_sgf=True)
api = api.load_full()          # full_forward and the candidate lists on their own are test-build entry points (include/dgnn_test.h)
board = api.Board(7.5)
for i, (x, y) in enumerate([(3, 3), (15, 15), (3, 15), (15, 3), (9, 9), (9, 10), (10, 9)]):
    board.place(i %% 2, x, y)
value, prior, _ = api.full_forward(board, 1)
print(json.dumps({"record": out["record"].tolist(), "sgf": out["sgf"], "evals": out["evals"], "value": value, "prior": [float(v) for v in prior],
                  "candidates": [board.candidates(1).tolist(), board.candidates(1, scoring=True).tolist()]}))
"""


def run(simd):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("DGNN_HOST_SIMD", None)
    if not simd:
        env["DGNN_HOST_SIMD"] = "0"
    done = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert done.returncode == 0, done.stderr[-2000:]
    return json.loads(done.stdout.strip().splitlines()[-1])


def test_scalar_and_vector_host_paths_play_the_same_games():
    vector, scalar = run(True), run(False)
    assert vector["evals"] == scalar["evals"] > 1000
    assert vector["record"] == scalar["record"]
    import re
    strip = lambda games: [re.sub(r"DT\[[^\]]*\]", "", g) for g in games]      # noqa: E731  (the date stamp of game_result.rs:30)
    assert strip(vector["sgf"]) == strip(scalar["sgf"])         # visit distributions (P[b85]), values, C[] text: every bit
    assert vector["value"] == scalar["value"] and vector["prior"] == scalar["prior"]
    assert vector["candidates"] == scalar["candidates"]
