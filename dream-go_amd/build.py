"""Builds libdgnn_hip.so (the C-ABI library, include/dgnn.h) for gfx950 with hipcc, in-tree.

    python dream-go_amd/build.py [--force]

The library depends only on the HIP runtime (libamdhip64); it does not link torch.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdgnn_hip.so")

SOURCES = ["loader.cpp", "network.cpp", "engine.cpp", "rccl.cpp", "batcher.cpp", "formats.cpp", "tree.cpp", "search.cpp", "go.cpp", "fiber.cpp", "kernels.hip", "conv_board.hip", "conv_board_tower.hip", "heads.hip", "features.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
if os.environ.get("DGNN_ABLATIONS", "0") == "1":        # the timing-only ablated kernels of tools/ablate*.sh and the NHWC tile kernels of
    FLAGS.append("-DDGNN_ABLATIONS")                    # round 1 (conv_tile.hip): A/B baselines, not part of the shipped library
    SOURCES.append("conv_tile.hip")


def hipcc():
    for candidate in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if candidate and (os.path.isabs(candidate) and os.path.exists(candidate) or not os.path.isabs(candidate)):
            return candidate
    raise RuntimeError("hipcc not found")


STAMP_PATH = os.path.join(LIB_DIR, "build.stamp")


def _stamp():
    """What an object file depends on besides its sources: the flags and the compiler.  Toggling DGNN_ABLATIONS=1 (or any flag)
    must not link objects built for the other setting."""
    try:
        version = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout.strip().splitlines()[0]
    except Exception:   # noqa: BLE001
        version = "unknown hipcc"
    return " ".join(FLAGS) + "\n" + version + "\n"


def _stamp_matches():
    try:
        return open(STAMP_PATH).read() == _stamp()
    except OSError:
        return False


def needs_build():
    if not os.path.exists(LIB_PATH) or not _stamp_matches():
        return True
    built = os.path.getmtime(LIB_PATH)
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for name in os.listdir(root):
            if os.path.getmtime(os.path.join(root, name)) > built:
                return True
    return False


def _stale(src_path, obj):
    """An object is rebuilt when its source, any header beside it or the public header is newer."""
    if not os.path.exists(obj):
        return True
    built = os.path.getmtime(obj)
    deps = [src_path] + [os.path.join(CSRC, n) for n in os.listdir(CSRC) if n.endswith(".h")]
    inc = os.path.join(HERE, "..", "include")
    deps += [os.path.join(inc, n) for n in os.listdir(inc)]
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    if not _stamp_matches():
        force = True                                       # other flags or another compiler: every object is stale
    objects, running = [], []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.rsplit(".", 1)[0] + ".o")
        objects.append(obj)
        if not force and not _stale(os.path.join(CSRC, src), obj):
            continue
        cmd = [hipcc(), *FLAGS, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        running.append((cmd, subprocess.Popen(cmd)))      # the translation units are independent: compile them side by side
    for cmd, proc in running:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objects]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(STAMP_PATH, "w") as f:
        f.write(_stamp())
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
