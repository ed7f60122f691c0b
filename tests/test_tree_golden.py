"""The search tree's sparse children (tree.cpp), the arena transposition table and the block pools against games recorded with the
DENSE tree (tests/golden/selfplay_dense_tree.json, written by tests/golden/make_selfplay_dense_tree.py at the last commit that had
it): three seeded self-play jobs on the CPU stand-in predictor -- every move, the number of evaluations and of probe conflicts, and
the SHA-256 of every game record (visit distributions P[], values, the C[] text of tree::to_pretty) -- with the AVX2 host paths and
with the scalar ones (DGNN_HOST_SIMD=0 is read once per process: child processes)."""
import json
import os
import subprocess
import sys

from conftest import GOLDEN, ROOT


def _run(simd):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("DGNN_HOST_SIMD", None)
    if not simd:
        env["DGNN_HOST_SIMD"] = "0"
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"dgnn_tree_golden_{os.getpid()}_{int(simd)}.json")
    done = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_selfplay_dense_tree.py"), out], capture_output=True, text=True,
                          timeout=600, env=env, cwd=ROOT)
    assert done.returncode == 0, done.stderr[-2000:]
    try:
        return json.load(open(out))
    finally:
        os.remove(out)


def test_the_sparse_tree_plays_the_games_of_the_dense_tree():
    want = json.load(open(os.path.join(GOLDEN, "selfplay_dense_tree.json")))
    for simd in (True, False):
        got = _run(simd)
        assert got.keys() == want.keys()
        for name in want:
            assert got[name]["record"] == want[name]["record"], (name, simd)
            assert (got[name]["evals"], got[name]["conflicts"]) == (want[name]["evals"], want[name]["conflicts"]), (name, simd)
            assert got[name]["sgf_sha256"] == want[name]["sgf_sha256"], (name, simd)
    assert sum(j["evals"] for j in want.values()) > 25000 and any(j["conflicts"] for j in want.values())
