import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

graft.package()   # registers dream-go_amd/ as `dream_go_amd`

GOLDEN = os.path.join(ROOT, "tests", "golden")

# tests/ablations/ exercises kernels that only the DGNN_ABLATIONS build of the library carries (csrc/ablations/): collected when that
# library is the one loaded, otherwise not at all -- no skips in the default run
collect_ignore = [] if "ablations" in os.environ.get("DGNN_LIBRARY", "") else ["ablations"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as module
    module.build()
    return module


@pytest.fixture(scope="session")
def api():
    """The product path: ctypes mirror of dg_nn over libdgnn_hip.so.  Raises if the library is missing."""
    graft.build()
    from dream_go_amd import api as module
    return module


@pytest.fixture(scope="session")
def api_full(api):
    """The same bindings over the TEST build of the library (lib/libdgnn_hip_test.so: the shipped library's kernel objects, the host
    sources with the entry points of include/dgnn_test.h exported as well).  Tests that pin single layers, the tree's / board's
    accessors or fault paths take this fixture; everything that only needs the product surface takes `api`.
    (Both libraries may be live in this process and each chains its own split launches per device -- ADVICE r05.  Harmless since
    round 6: a split launch makes progress with whatever compute units it is given, conv_board_tower.hip.)"""
    return api.load_full()


@pytest.fixture(scope="session")
def synth():
    from dream_go_amd import synth as module
    return module


def f16_to_f32(bits):
    return np.asarray(bits).view(np.float16).astype(np.float32)


def policy_kl(p_ref, p):
    """KL(p_ref || p) per board, on fp16 probabilities promoted to fp64, with a floor against log(0)."""
    p_ref = np.asarray(p_ref, dtype=np.float64).reshape(-1, 362)
    p = np.asarray(p, dtype=np.float64).reshape(-1, 362)
    eps = 1e-12
    return np.sum(p_ref * (np.log(p_ref + eps) - np.log(p + eps)), axis=1)
