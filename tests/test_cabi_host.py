"""CPU-side tests of the C-ABI library: it loads, exports exactly the symbols its product headers declare, ingests the
reference's weights format identically to the oracle's restatement of loader.rs, and fails with status codes
(never a crash, never a silent fallback) when there is no GPU.  No compute is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dgnn_[a-z0-9_]+)\s*\(", text)))


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("dgnn_")})


def test_the_shipped_library_exports_the_two_product_headers_and_nothing_else(api):
    """include/dgnn.h is the drop-in boundary, include/dgnn_host.h the C++ host rows (8f); what the
    tests need beyond them is in include/dgnn_test.h and is exported by the TEST build only.  The shipped .so exports exactly the
    product surface: at most 110 entry points (round 4: 170 from one 749-line header)."""
    boundary, host, test = _declared_symbols("dgnn.h"), _declared_symbols("dgnn_host.h"), _declared_symbols("dgnn_test.h")
    assert len(boundary) >= 40 and not set(boundary) & set(host) and not (set(boundary) | set(host)) & set(test)
    exported = _exported(api.LIB_PATH)
    assert exported == sorted(set(boundary) | set(host)), set(exported) ^ (set(boundary) | set(host))
    assert len(exported) <= 110
    assert set(test) == set(api.TEST_ONLY)
    bound = {name for name, _, _ in api.SYMBOLS}
    assert bound == set(boundary) | set(host) | set(test), bound ^ (set(boundary) | set(host) | set(test))
    # a test-only entry point is not reachable through the shipped library: it fails loudly at the call
    with pytest.raises(AttributeError):
        api.lib.dgnn_layer_conv3x3


def test_the_test_library_exports_all_three_headers(api_full):
    assert api_full.LIB_PATH != __import__("dream_go_amd").api.LIB_PATH or os.environ.get("DGNN_LIBRARY")
    exported = set(_exported(api_full.LIB_PATH))
    for header in ("dgnn.h", "dgnn_host.h", "dgnn_test.h"):
        missing = [name for name in _declared_symbols(header) if name not in exported]
        assert not missing, (header, missing)


def test_status_strings(api):
    for status in (0, -1, -2, -3, -4, -5, -6, -99):
        assert api.lib.dgnn_status_string(status)
    assert b"gfx950" in api.lib.dgnn_version()


def test_loader_kats_through_the_c_abi(api):
    # loader.rs:123-129 empty_json
    with pytest.raises(api.Error) as e:
        api.Network.from_json("")
    assert e.value.status == api.MISSING_WEIGHTS
    with pytest.raises(api.Error) as e:
        api.Network.from_json("{}")
    assert e.value.status == api.MISSING_WEIGHTS
    # loader.rs:131-141 load_json parses, but a single offset tensor is not a network
    with pytest.raises(api.Error) as e:
        api.Network.from_json('{"11v_value/linear_2/offset:0": {"s": "(^d>V", "t": "f2", "v": "(^d>V"}}')
    assert e.value.status == api.MALFORMED_WEIGHTS
    for bad in ('{"a": {"t": "f8"}}', '{"a": {"q": "00000"}}', '{"a": {"v": "00 00"}}', '{"a": 5}', '[1]', '{"a": {"t": "f2"}'):
        with pytest.raises(api.Error) as e:
            api.Network.from_json(bad)
        assert e.value.status == api.MALFORMED_WEIGHTS, bad


def test_missing_file_and_default_search(api, tmp_path, monkeypatch):
    with pytest.raises(api.Error) as e:
        api.Network.load(str(tmp_path / "nope.json"))
    assert e.value.status == api.MISSING_WEIGHTS
    monkeypatch.chdir(tmp_path)
    assert api.Network.new() is None                      # Network::new() -> None (network.rs:108-127)


def test_default_search_order(api, synth, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    os.makedirs(tmp_path / "models")
    synth.write_network(str(tmp_path / "models" / "dream_go.json"), 1, 32, num_samples=2, seed=2)
    net = api.Network.new()
    assert net is not None and net.info()["num_blocks"] == 1
    synth.write_network(str(tmp_path / "dream_go.json"), 2, 32, num_samples=2, seed=2)
    assert api.Network.new().info()["num_blocks"] == 2     # "dream_go.json" precedes "models/dream_go.json"
    synth.write_network(str(tmp_path / "engine.json"), 3, 32, num_samples=2, seed=2)
    assert api.Network.new(exe_path=str(tmp_path / "engine")).info()["num_blocks"] == 3   # <exe>.json first
    (tmp_path / "engine.json").write_text('{"a": {"t": "zz"}}')
    with pytest.raises(api.Error) as e:                    # a malformed file is fatal, not skipped (network.rs:114-116)
        api.Network.new(exe_path=str(tmp_path / "engine"))
    assert e.value.status == api.MALFORMED_WEIGHTS


@pytest.mark.parametrize("blocks,channels,samples", [(0, 32, 8), (2, 64, 8), (3, 96, 4)])
def test_topology_discovery(api, synth, blocks, channels, samples):
    net = api.Network.from_json(synth.to_json(synth.make_tensors(blocks, channels, num_samples=samples, seed=1)))
    assert net.info() == {"num_blocks": blocks, "num_channels": channels, "num_samples": samples}


def test_topology_defaults_and_probing_stops_at_gap(api, synth):
    tensors = synth.make_tensors(3, 128, num_samples=8, seed=1)
    for name in ("num_channels:0", "num_samples:0", "num_blocks:0"):
        del tensors[name]                                  # layers/common.rs:22-50 defaults 128 / 8
    assert api.Network.from_json(synth.to_json(tensors)).info() == {"num_blocks": 3, "num_channels": 128, "num_samples": 8}

    del tensors["03_residual/conv_2:0"]                     # graph.rs:76-96 stops at the first missing block ...
    # ... and the heads are then looked up as "{2 + 1}p_policy" (graph.rs:54-55), which this file does not hold:
    # the reference panics ("no filter available", conv2d.rs:62-63); the C ABI reports malformed weights.
    with pytest.raises(api.Error) as e:
        api.Network.from_json(synth.to_json(tensors))
    assert e.value.status == api.MALFORMED_WEIGHTS

    # re-number the heads to follow the single remaining block and the same file loads as a 1-block network
    renamed = {name.replace("05p_policy", "03p_policy").replace("05v_value", "03v_value"): t for name, t in tensors.items()}
    assert api.Network.from_json(synth.to_json(renamed)).info()["num_blocks"] == 1


def test_unsupported_channel_count(api, synth):
    with pytest.raises(api.Error) as e:
        api.Network.from_json(synth.to_json(synth.make_tensors(1, 48, seed=1)))
    assert e.value.status == api.UNSUPPORTED


def test_weight_image_layout_and_prescaling(api, synth, oracle):
    blocks, c, s = 1, 32, 2
    text = synth.to_json(synth.make_tensors(blocks, c, num_samples=s, seed=4, alpha=0.25))
    net = api.Network.from_json(text)
    image = net.weight_image_host().copy()
    tensors, _ = oracle.load_weights(text)

    # the image starts with the zero page (off-board taps), then the header: everything a receiver of a broadcast needs to
    # launch besides tensors -- architecture, packing, the residual gates and the temperature (network.cpp: ImageHeader)
    assert not image[:256].any()
    assert image[256:264].tobytes() == b"DGNNIMG2"
    header_bytes, layouts = image[264:272].view(np.uint32)
    assert image[272:280].view(np.uint64)[0] == image.size
    assert tuple(image[280:292].view(np.int32)) == (blocks, c, s)
    assert image[292:296].view(np.float32)[0] == np.float32(0.709888)
    assert header_bytes == 40 + 4 * blocks and image[296:300].view(np.float32)[0] == np.float32(0.25)   # alpha of block 0
    assert layouts == 1                                   # C = 32: only the generic layout exists for it

    # first tensor behind the header: up-conv filter, [9 * cin / 32][cout_pad][32] from KRSC [cout][3][3][cin]
    w = tensors["01_upsample/conv_1:0"].view(np.uint16)[: c * 9 * 32].reshape(c, 9, 32)
    expect = np.transpose(w, (1, 0, 2)).reshape(-1)
    got = image[512: 512 + expect.size * 2].view(np.uint16)
    assert np.array_equal(got, expect)

    # alpha * b2 rounded to fp16 (residual_block.rs:72-74) and tau * b_l (policy_head.rs:87-89) appear in the image
    b2 = tensors["02_residual/conv_2/offset:0"].astype(np.float32)[:c]
    scaled = (np.float32(0.25) * b2).astype(np.float16).view(np.uint16)
    words = image.view(np.uint16)
    assert _find(words, scaled) >= 0
    tau = np.float32(1.0) / np.float32(0.709888)
    bl = (tau * tensors["03p_policy/linear_1/offset:0"].astype(np.float32)[:362]).astype(np.float16).view(np.uint16)
    assert _find(words, bl) >= 0

    # temperature is baked at build time and can be changed before the first workspace (policy_head.rs:46)
    before = net.weight_checksum()
    net.set_softmax_temperature(1.0)
    assert net.weight_checksum() != before
    bl1 = tensors["03p_policy/linear_1/offset:0"][:362].view(np.uint16)
    assert _find(net.weight_image_host().view(np.uint16), bl1) >= 0


def test_image_carries_only_the_layouts_its_kernels_read(api_full, synth, monkeypatch):
    """the image held every filter three ways (148.5 MB for 20 x 256).  By default only the board layout is
    packed where the board kernel covers the network; the NHWC layouts are packed when DGNN_CONV selects those kernels."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    blocks, c = 2, 256
    text = synth.to_json(synth.make_tensors(blocks, c, seed=8))
    filters = (2 * blocks * 9 * c * c + 9 * 32 * c) * 2                      # residual + up-layer filter bytes
    monkeypatch.delenv("DGNN_CONV", raising=False)
    monkeypatch.delenv("DGNN_PACK", raising=False)
    board_only = api.Network.from_json(text).weight_image_host().size
    assert filters < board_only < filters + 6_000_000                       # + heads (2 x 2.1 MB dense, 147 KB head filter), padding
    monkeypatch.setenv("DGNN_CONV", "tile2")
    nhwc = api.Network.from_json(text).weight_image_host().size
    assert 2 * filters - 1_000_000 < nhwc < 2 * filters + 6_000_000          # generic + tile
    monkeypatch.setenv("DGNN_CONV", "v1")
    assert api.Network.from_json(text).weight_image_host().size < nhwc - filters + 1_000_000
    monkeypatch.delenv("DGNN_CONV")
    monkeypatch.setenv("DGNN_PACK", "all")
    assert api.Network.from_json(text).weight_image_host().size > 3 * filters


def test_tile_filter_layout(api, synth, oracle, monkeypatch):
    """conv_tile.hip consumes [cout/256][chunk * 9 + tap][256 rows][8 slots][8] with slot ^= (row >> 1) & 7."""
    monkeypatch.setenv("DGNN_CONV", "tile2")               # the tile layout is packed when the tile kernel is selected
    c = 256
    text = synth.to_json(synth.make_tensors(1, c, num_samples=2, seed=6))
    net = api.Network.from_json(text)                      # keep alive: the image view borrows its memory
    words = net.weight_image_host().view(np.uint16)
    tensors, _ = oracle.load_weights(text)
    w = tensors["02_residual/conv_1:0"].view(np.uint16)[: c * 9 * c].reshape(c, 9, c // 64, 8, 8)   # [co][tap][chunk][piece][e]
    rows = np.arange(c)
    expect = np.empty((c // 64, 9, c, 8, 8), dtype=np.uint16)                                        # [chunk][tap][row][slot][e]
    for slot in range(8):
        piece = slot ^ ((rows >> 1) & 7)
        expect[:, :, rows, slot, :] = np.transpose(w[rows, :, :, piece, :], (2, 1, 0, 3))
    at = _find(words, expect.reshape(-1)[:4096])
    assert at >= 0
    assert np.array_equal(words[at: at + expect.size], expect.reshape(-1))


def _find(haystack, needle):
    first = np.flatnonzero(haystack == needle[0])
    for i in first:
        if i + needle.size <= haystack.size and np.array_equal(haystack[i:i + needle.size], needle):
            return int(i)
    return -1


def test_softmax_temperature_env(api, synth, monkeypatch):
    text = synth.to_json(synth.make_tensors(0, 32, num_samples=2, seed=4))
    a = api.Network.from_json(text).weight_checksum()
    monkeypatch.setenv("SOFTMAX_TEMPERATURE", "1.5")      # config.rs:176-177
    b = api.Network.from_json(text).weight_checksum()
    monkeypatch.setenv("SOFTMAX_TEMPERATURE", "0.709888")
    c = api.Network.from_json(text).weight_checksum()
    assert a != b and a == c


def test_checksum_is_deterministic(api, synth):
    t1 = synth.to_json(synth.make_tensors(1, 32, num_samples=2, seed=1))
    t2 = synth.to_json(synth.make_tensors(1, 32, num_samples=2, seed=2))
    assert api.Network.from_json(t1).weight_checksum() == api.Network.from_json(t1).weight_checksum()
    assert api.Network.from_json(t1).weight_checksum() != api.Network.from_json(t2).weight_checksum()


def test_no_gpu_is_an_error_not_a_fallback(api_full, synth):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    try:
        count = api.Device.len()
    except api.Error as e:
        assert e.status == api.HIP_ERROR and e.hip_error != 0
        count = 0
    if count > 0:
        pytest.skip("a GPU is present; the no-device path is covered on the CPU box")
    net = api.Network.from_json(synth.to_json(synth.make_tensors(0, 32, num_samples=2, seed=1)))
    with pytest.raises(api.Error) as e:
        net.get_workspace(1)
    assert e.value.status == api.HIP_ERROR                 # the product path fails loudly without its device
    x = np.zeros(361 * 32, dtype=np.uint16)
    with pytest.raises(api.Error):
        api.layer_conv3x3(x, 1, 19, 32, np.zeros(16 * 9 * 32, np.uint16), np.zeros(16, np.uint16), 16)


def test_bad_arguments(api):
    assert api.lib.dgnn_network_info(None, None, None, None) == api.BAD_ARGUMENT
    assert api.lib.dgnn_forward(None, None, 0, None, None) == api.BAD_ARGUMENT
    assert api.lib.dgnn_device_count(None) == api.BAD_ARGUMENT
    assert api.lib.dgnn_network_get_workspace(None, 0, 1, None) == api.BAD_ARGUMENT
    api.lib.dgnn_workspace_release(None)
    api.lib.dgnn_network_destroy(None)


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/dgnn.h must be usable from C (the boundary is a C ABI): the example host compiles as strict C99 and links
    against the library with nothing but the header."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "forward_example"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "forward.c"), "-L" + os.path.join(root, "dream-go_amd", "lib"),
                    "-ldgnn_hip", "-Wl,-rpath," + os.path.join(root, "dream-go_amd", "lib"), "-o", str(exe)], check=True)
    # without a weights file it reports the reference's MissingWeights and exits non-zero (no GPU needed for that)
    run = subprocess.run([str(exe), str(tmp_path / "nothing.json")], capture_output=True, text=True)
    assert run.returncode == 1 and "missing weights" in run.stderr


def test_board_record_example_in_c(tmp_path):
    """examples/forward_boards.c (strict C99): the record of a position, expanded by the host statement of the device kernel, equals
    features::V1's planes; without a weights file the device half is skipped."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "forward_boards_example"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "forward_boards.c"), "-L" + os.path.join(root, "dream-go_amd", "lib"),
                    "-ldgnn_hip", "-Wl,-rpath," + os.path.join(root, "dream-go_amd", "lib"), "-o", str(exe)], check=True)
    run = subprocess.run([str(exe), str(tmp_path / "nothing.json")], capture_output=True, text=True)
    assert run.returncode == 0 and "planes from the record equal the planes of features::V1" in run.stdout, run.stdout + run.stderr


def test_selfplay_example_in_c(tmp_path):
    """examples/selfplay.c: the host rows (board, search, cache, game records) from strict C99 through the same header; without
    a weights file it plays on the CPU stand-in predictor, so it runs here."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "selfplay_example"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "selfplay.c"), "-L" + os.path.join(root, "dream-go_amd", "lib"),
                    "-ldgnn_hip", "-Wl,-rpath," + os.path.join(root, "dream-go_amd", "lib"), "-o", str(exe)], check=True)
    class Run:          # bytes, not text mode: a record holds carriage returns (the C[] text of every searched move, self_play.rs:192)
        def __init__(self):
            done = subprocess.run([str(exe)], capture_output=True, check=True)
            self.stdout, self.stderr = done.stdout.decode("ascii"), done.stderr.decode("ascii")
    first, again = Run(), Run()
    games = [g for g in first.stdout.split("\n") if g]
    assert all(g.count("C[Nodes: ") == 30 for g in games)
    assert len(games) == 2 and all(g.startswith("(;GM[1]FF[4]DT[") and g.endswith(")") for g in games)
    assert all(len(re.findall(r";[BW]\[[a-s]{2}\]", g)) == 30 for g in games)
    strip = lambda text: re.sub(r"DT\[[^\]]*\]", "", text)          # the date is the only thing that may differ between two runs
    assert sorted(strip(first.stdout).split("\n")) == sorted(strip(again.stdout).split("\n"))    # (printed as the games end: in that order)
    assert "2 games, 60 stones" in first.stderr


# ---- round 2: packed features, RCCL entry points (status paths without a GPU)

def test_packed_features_round_trip(api, synth):
    """dgnn_features_pack / _unpack: 361 plane masks + the two komi constants describe a features::V1 tensor exactly."""
    rng = np.random.default_rng(12)
    count = 3
    f = (rng.random((count, 361, 32)) < 0.2).astype(np.float16)
    komi = np.float16(0.5 + 0.5 * 6.5 / 7.5)
    f[0, :, 0] = komi; f[0, :, 1] = 0                        # black to move
    f[1, :, 0] = 0; f[1, :, 1] = komi                        # white to move
    f[2, :, 0] = 0; f[2, :, 1] = 0                           # komi -7.5 clamps to 0 (features.rs:236-239)
    packed = api.features_pack(f, count)
    assert packed.size == count * api.PACKED_WORDS
    words = packed.reshape(count, api.PACKED_WORDS)
    assert words[0, 361] == int(komi.view(np.uint16)) and words[1, 361] == int(komi.view(np.uint16)) << 16 and words[2, 361] == 0
    assert not words[:, 362:].any()
    for c in range(32):
        assert np.array_equal(((words[:, :361] >> c) & 1).astype(bool), f[:, :, c] != 0)
    back = api.features_unpack(packed, count)
    assert np.array_equal(back.view(np.uint16), f.reshape(-1).view(np.uint16))

    bad = f.copy(); bad[1, 5, 7] = 0.5                       # a binary plane holding something else
    with pytest.raises(api.Error) as e:
        api.features_pack(bad, count)
    assert e.value.status == api.BAD_ARGUMENT
    bad = f.copy(); bad[0, 9, 0] = 1.0                       # two different constants in channel 0
    with pytest.raises(api.Error) as e:
        api.features_pack(bad, count)
    assert e.value.status == api.BAD_ARGUMENT


def test_real_board_features_are_packable(api):
    """Every tensor dgnn_go_features produces has the packed form (all 8 symmetries, komi planes, ko, ladders)."""
    board = api.Board(komi=6.5)
    moves = [(3, 3), (15, 15), (3, 15), (15, 3), (9, 9), (2, 5), (5, 2), (16, 13), (13, 16), (10, 10)]
    for i, (x, y) in enumerate(moves):
        board.place(i % 2, x, y)
    for sym in range(8):
        f = board.get_features(len(moves) % 2, sym)
        back = api.features_unpack(api.features_pack(f, 1), 1)
        assert np.array_equal(back.view(np.uint16), np.asarray(f).reshape(-1).view(np.uint16))


def test_rccl_entry_points_without_a_gpu(api):
    """The collective lives in the library (dgnn_broadcast_weights); librccl is opened on demand.  Argument and
    status paths that need no device."""
    import ctypes
    assert api.lib.dgnn_rccl_available() in (0, 1)
    assert api.lib.dgnn_rccl_unique_id(None) == api.BAD_ARGUMENT
    out = ctypes.c_void_p()
    ident = (ctypes.c_uint8 * api.RCCL_UNIQUE_ID_BYTES)()
    assert api.lib.dgnn_rccl_comm_create(0, 0, 0, ident, ctypes.byref(out)) == api.BAD_ARGUMENT      # no ranks
    assert api.lib.dgnn_rccl_comm_create(0, 2, 2, ident, ctypes.byref(out)) == api.BAD_ARGUMENT      # rank outside the job
    assert api.lib.dgnn_rccl_comm_create(0, 2, 0, None, ctypes.byref(out)) == api.BAD_ARGUMENT
    assert api.lib.dgnn_rccl_comm_wrap(None, 0, ctypes.byref(out)) == api.BAD_ARGUMENT
    assert api.lib.dgnn_broadcast_weights(None, None, 0) == api.BAD_ARGUMENT
    assert api.lib.dgnn_broadcast_weights_devices(None, None, 2) == api.BAD_ARGUMENT
    assert api.lib.dgnn_rccl_comm_info(None, None, None, None) == api.BAD_ARGUMENT
    api.lib.dgnn_rccl_comm_destroy(None)
    assert api.lib.dgnn_status_string(api.COMM_ERROR) == b"RCCL error"
    assert api.lib.dgnn_rccl_error_string(0)                                                         # never NULL
