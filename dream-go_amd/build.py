"""Builds the C-ABI library for gfx950 with hipcc, in-tree.

    python dream-go_amd/build.py [--force]

    lib/libdgnn_hip.so        the shipped library: exports what include/dgnn.h and include/dgnn_host.h declare, nothing else
    lib/libdgnn_hip_test.so   the SAME kernel objects + the host sources compiled with -DDGNN_TEST_ENTRIES: additionally exports the
                              test-only entry points of include/dgnn_test.h (layer-level calls, accessors the reference's unit tests
                              look at, fault injection).  The tests that need them load this one (tests/conftest.py: api_full).
    DGNN_ABLATIONS=1          one library (lib/ablations/libdgnn_hip.so) with what measurements need and the shipped library does not carry:
                              the NHWC tile kernels of round 1 (csrc/ablations/conv_tile.hip, an A/B baseline), the timing-only ablated
                              board kernels, the stamp / trace hooks -- and every test entry.  Its tests: tests/ablations/ (collected only
                              when that library is the one loaded, DGNN_LIBRARY=dream-go_amd/lib/ablations/libdgnn_hip.so)

Both depend only on the HIP runtime (libamdhip64); neither links torch.
"""
import os
import subprocess
import time
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ABLATIONS = os.environ.get("DGNN_ABLATIONS", "0") == "1"
LIB_DIR = os.path.join(HERE, "lib", "ablations") if ABLATIONS else os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdgnn_hip.so")
TEST_LIB_PATH = LIB_PATH if ABLATIONS else os.path.join(LIB_DIR, "libdgnn_hip_test.so")

HOST_SOURCES = ["loader.cpp", "network.cpp", "engine.cpp", "rccl.cpp", "batcher.cpp", "formats.cpp", "tree.cpp", "search.cpp", "go.cpp", "fiber.cpp"]
KERNEL_SOURCES = ["conv_board.hip", "conv_board_b.hip", "conv_board_tower.hip", "kernels.hip", "heads.hip", "features.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
if ABLATIONS:
    FLAGS.append("-DDGNN_ABLATIONS")
    KERNEL_SOURCES.append(os.path.join("ablations", "conv_tile.hip"))
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic-functions"]    # calls inside the library bind inside it: the shipped
                                                                                           # and the test library may live in one process


def hipcc():
    for candidate in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if candidate and (os.path.isabs(candidate) and os.path.exists(candidate) or not os.path.isabs(candidate)):
            return candidate
    raise RuntimeError("hipcc not found")


STAMP_PATH = os.path.join(LIB_DIR, "build.stamp")


def _stamp():
    """What an object file depends on besides its sources: the flags and the compiler."""
    try:
        version = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout.strip().splitlines()[0]
    except Exception:   # noqa: BLE001
        version = "unknown hipcc"
    return " ".join(FLAGS + LINK_FLAGS) + "\n" + version + "\n"


def _stamp_matches():
    try:
        return open(STAMP_PATH).read() == _stamp()
    except OSError:
        return False


def _headers():
    inc = os.path.join(HERE, "..", "include")
    return [os.path.join(CSRC, n) for n in os.listdir(CSRC) if n.endswith(".h")] + [os.path.join(inc, n) for n in os.listdir(inc)]


def _kernel_headers():
    """The kernel translation units include the csrc headers only (kernels.h, conv_board_body.h ...), never the public ones: a change of
    include/*.h recompiles the host sources (seconds), not the kernels (minutes)."""
    return [os.path.join(CSRC, n) for n in os.listdir(CSRC) if n.endswith(".h") and n != "dgnn_internal.h"]


def needs_build():
    if not os.path.exists(LIB_PATH) or not os.path.exists(TEST_LIB_PATH) or not _stamp_matches():
        return True
    built = min(os.path.getmtime(LIB_PATH), os.path.getmtime(TEST_LIB_PATH))
    return any(os.path.getmtime(d) > built for d in _headers() + [os.path.join(CSRC, n) for n in HOST_SOURCES + KERNEL_SOURCES])


def _stale(src_path, obj, headers):
    if not os.path.exists(obj):
        return True
    built = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > built for d in [src_path] + headers)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(os.path.join(LIB_DIR, "test"), exist_ok=True)
    if not _stamp_matches():
        force = True                                       # other flags or another compiler: every object is stale
    if not ABLATIONS:
        stale = os.path.join(LIB_DIR, "conv_tile.o")       # (an object of an ablations build of rounds 1-4 that used to share this directory)
        if os.path.exists(stale):
            os.remove(stale)
    jobs, running = [], []
    kernel_objects, host_objects, test_objects = [], [], []
    for src in KERNEL_SOURCES:
        obj = os.path.join(LIB_DIR, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        kernel_objects.append(obj)
        jobs.append((src, obj, [], _kernel_headers()))
    for src in HOST_SOURCES:
        obj = os.path.join(LIB_DIR, src.rsplit(".", 1)[0] + ".o")
        host_objects.append(obj)
        jobs.append((src, obj, [], _headers()))
        if not ABLATIONS:
            tobj = os.path.join(LIB_DIR, "test", src.rsplit(".", 1)[0] + ".o")
            test_objects.append(tobj)
            jobs.append((src, tobj, ["-DDGNN_TEST_ENTRIES"], _headers()))
    # The translation units are independent: compile them side by side, as many at a time as there are CPUs, the kernel units (the
    # long ones: the critical path of a clean build) first, so that they never share a core with the twenty short host units.
    pending = []
    for src, obj, extra, headers in jobs:
        if not force and not _stale(os.path.join(CSRC, src), obj, headers):
            continue
        pending.append([hipcc(), *FLAGS, *extra, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj])
    slots = max(2, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2))
    while pending or running:
        while pending and len(running) < slots:
            cmd = pending.pop(0)
            if verbose:
                print(" ".join(cmd))
            running.append((cmd, subprocess.Popen(cmd)))
        for entry in running:
            if entry[1].poll() is not None:
                if entry[1].returncode != 0:
                    for _, other in running:
                        if other.poll() is None:
                            other.kill()
                    raise subprocess.CalledProcessError(entry[1].returncode, entry[0])
                running.remove(entry)
                break
        else:
            time.sleep(0.05)
    links = [(LIB_PATH, host_objects + kernel_objects)]
    if not ABLATIONS:
        links.append((TEST_LIB_PATH, test_objects + kernel_objects))
    for path, objects in links:
        cmd = [hipcc(), *LINK_FLAGS, "-o", path, *objects]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    with open(STAMP_PATH, "w") as f:
        f.write(_stamp())
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
