"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle, on a real MI355X.

Tolerances are the ones BASELINE.json's north_star states: value abs diff <= 1e-3, policy max-abs diff <= 1e-3
(both on fp16 outputs promoted to fp32); policy KL(oracle || hip) is asserted small and reported.
Layer level: the convolution output is compared after fp16 rounding; the oracle accumulates k in order in
fp32, the MFMA kernel accumulates 32-wide k-groups in fp32, so results agree to a few fp16 ulps of the output.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, f16_to_f32, policy_kl

pytestmark = pytest.mark.gpu

VALUE_TOL = 1e-3
POLICY_TOL = 1e-3


def _close_fp16(got_bits, ref_bits, ulps=2.0, atol=2e-3):
    """|got - ref| <= ulps * ulp_fp16(|ref|) + tiny; returns (ok, worst absolute difference)."""
    got, ref = f16_to_f32(got_bits), f16_to_f32(ref_bits)
    diff = np.abs(got - ref)
    bound = ulps * np.maximum(np.abs(ref), 2.0 ** -14) * 2.0 ** -10 + atol * 0
    return bool(np.all(diff <= bound + 1e-7)), float(diff.max()), float((diff / np.maximum(np.abs(ref), 1e-3)).max())


# --------------------------------------------------------------------------------------------- layer level

def test_conv2d_pick_middle_and_sum_through_hip(api_full):
    """The reference's own layer test (conv2d.rs:252-291) through the HIP kernel; cin padded 1 -> 32 with zeros."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    k = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))["conv2d_pick_middle_and_sum"]
    x = np.zeros((9, 32), dtype=np.float16); x[:, 0] = k["input"]
    w = np.zeros((2, 9, 32), dtype=np.float16); w[:, :, 0] = np.asarray(k["filter_kcrs_as_listed"], dtype=np.float16).reshape(2, 9)
    y = api.layer_conv3x3(x, 1, 3, 32, w, np.zeros(2, np.float16), 2, relu=True)
    assert f16_to_f32(y).tolist() == [float(v) for v in k["expected_nhwc"]]


def test_conv_asymmetric_taps_and_channels(api_full, oracle):
    """A transpose-detecting case: every (tap, cin, cout) weight distinct, single hot input pixels."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    cin, cout = 32, 16
    w = (np.arange(cout * 9 * cin, dtype=np.float32).reshape(cout, 9, cin) % 251) / 64.0
    x = np.zeros((1, 19, 19, cin), dtype=np.float32)
    x[0, 0, 0, 3] = 1.0; x[0, 7, 11, 17] = 2.0; x[0, 18, 18, 31] = -1.0; x[0, 18, 0, 0] = 0.5
    b = np.linspace(-1, 1, cout).astype(np.float32)
    args = (oracle.f16(x), 1, 19, cin, oracle.f16(w), oracle.f16(b), cout)
    ref = oracle.conv3x3(*args, relu=False)
    got = api.layer_conv3x3(*args, relu=False)
    assert np.array_equal(got, ref)      # every output is one exact product (+ bias): bit-exact


@pytest.mark.parametrize("batch,cin,cout,alpha,relu", [
    (1, 32, 64, None, True),       # up-layer shape (K1)
    (3, 32, 128, None, True),      # up-layer shape (K2), ragged tile (3 * 361 = 1083 positions)
    (2, 64, 64, 0.5, True),        # residual conv_2 with gated skip (K1)
    (2, 128, 128, 0.25, True),     # K2
    (1, 256, 256, 0.5, True),      # K3
    (2, 256, 256, None, False),    # identity activation
    (2, 96, 96, 0.5, True),        # channel count that is not a power of two (partial N tile)
    (2, 128, 8, None, True),       # policy-head conv (fp32 compute in the reference, policy_head.rs:51)
    (2, 128, 2, None, True),       # value-head conv (value_head.rs:47), scalar epilogue
    (5, 64, 2, None, True),
])
def test_conv_layer_against_oracle(api_full, oracle, batch, cin, cout, alpha, relu):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng([batch, cin, cout])
    x = np.maximum(rng.normal(0, 1, (batch, 19, 19, cin)), 0).astype(np.float16)     # post-ReLU like activations
    w = rng.normal(0, np.sqrt(2.0 / (9 * cin)), (cout, 3, 3, cin)).astype(np.float16)
    b = rng.normal(0, 0.1, (cout,)).astype(np.float16)
    z = np.maximum(rng.normal(0, 1, (batch, 19, 19, cout)), 0).astype(np.float16) if alpha is not None else None
    a1, a2 = (alpha, 1.0 - alpha) if alpha is not None else (1.0, 0.0)
    ref = oracle.conv3x3(x, batch, 19, cin, w, b, cout, alpha1=a1, alpha2=a2, z=z, relu=relu)
    got = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=a1, alpha2=a2, z=z, relu=relu)
    ok, worst_abs, worst_rel = _close_fp16(got, ref, ulps=2.0)
    mismatched = int(np.count_nonzero(got != ref))
    print(f"conv B={batch} {cin}->{cout}: {mismatched}/{got.size} outputs differ, max abs {worst_abs:.3e}, max rel {worst_rel:.3e}")
    assert np.abs(f16_to_f32(got) - f16_to_f32(ref)).max() <= 4e-3 * max(1.0, float(np.abs(f16_to_f32(ref)).max()))
    assert mismatched <= 0.02 * got.size            # different fp32 summation order flips few fp16 roundings


# ---- board kernel (conv_board.hip): padded slot-major layout, one board per workgroup

def test_board_kernel_asymmetric_taps(api_full, oracle, monkeypatch):
    """Transpose- and shift-detecting: distinct weight per (tap, cin, cout), hot pixels in corners, on edges and in
    the interior; every output is one exact product (+ bias), so the board path must be bit-exact."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    cin, cout = 32, 256
    w = ((np.arange(cout * 9 * cin, dtype=np.float32).reshape(cout, 9, cin) * 7) % 251) / 64.0
    x = np.zeros((2, 19, 19, cin), dtype=np.float32)
    x[0, 0, 0, 3] = 1.0; x[0, 7, 11, 17] = 2.0; x[0, 18, 18, 31] = -1.0; x[0, 18, 0, 0] = 0.5; x[0, 0, 18, 9] = 4.0
    x[1, 9, 0, 1] = 1.0; x[1, 9, 18, 30] = 2.0; x[1, 3, 4, 15] = -2.0
    b = np.linspace(-1, 1, cout).astype(np.float32)
    args = (oracle.f16(x), 2, 19, cin, oracle.f16(w), oracle.f16(b), cout)
    ref = oracle.conv3x3(*args, relu=False)
    monkeypatch.setenv("DGNN_CONV", "board")
    got = api.layer_conv3x3(*args, relu=False)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("batch,cin,cout,alpha,relu", [
    (1, 32, 256, None, True),      # up-layer shape of configs[2]: one chunk
    (3, 256, 256, 0.5, True),      # residual conv_2 with gated skip
    (2, 256, 256, None, False),    # identity activation, negative outputs
    (2, 64, 512, 0.25, True),      # two 256-channel blocks (grid.y = 2), two chunks
    (2, 128, 128, 0.5, True),      # configs[1] residual shape (9 x 128)
    (3, 64, 192, 0.375, True),     # 6 slices: only 2 slices per workgroup divide it
    (1, 32, 64, None, True),       # configs[0] up-layer shape (6 x 64)
    (2, 64, 64, 0.5, False),
])
@pytest.mark.parametrize("skip_mode", ["0", "1"])     # skip folded into the K loop (default) / added in the epilogue
@pytest.mark.parametrize("ns", ["16", "4", "2", "1"])  # 32-channel slices per workgroup (ignored when it does not divide C_out / 32); 16 = eight on four waves, 1 = twelve waves
def test_board_kernel_against_oracle(api_full, oracle, monkeypatch, batch, cin, cout, alpha, relu, skip_mode, ns):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    monkeypatch.setenv("DGNN_BOARD_SKIP", skip_mode)
    monkeypatch.setenv("DGNN_BOARD_NS", ns)
    rng = np.random.default_rng([7, batch, cin, cout])
    x = np.maximum(rng.normal(0, 1, (batch, 19, 19, cin)), 0).astype(np.float16)
    w = rng.normal(0, np.sqrt(2.0 / (9 * cin)), (cout, 3, 3, cin)).astype(np.float16)
    b = rng.normal(0, 0.1, (cout,)).astype(np.float16)
    z = np.maximum(rng.normal(0, 1, (batch, 19, 19, cout)), 0).astype(np.float16) if alpha is not None else None
    a1, a2 = (alpha, 1.0 - alpha) if alpha is not None else (1.0, 0.0)
    ref = oracle.conv3x3(x, batch, 19, cin, w, b, cout, alpha1=a1, alpha2=a2, z=z, relu=relu)
    monkeypatch.setenv("DGNN_CONV", "board")
    got = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=a1, alpha2=a2, z=z, relu=relu)
    mismatched = int(np.count_nonzero(got != ref))
    print(f"board conv B={batch} {cin}->{cout}: {mismatched}/{got.size} outputs differ")
    assert np.abs(f16_to_f32(got) - f16_to_f32(ref)).max() <= 4e-3 * max(1.0, float(np.abs(f16_to_f32(ref)).max()))
    assert mismatched <= 0.02 * got.size


@pytest.mark.parametrize("ns", ["16", "4", "2", "1"])
def test_board_kernel_exact_under_full_load(api_full, monkeypatch, ns):
    """configs[2] layer shape at full size (256 boards = 256 workgroups, every CU busy): integer-valued data make the
    generic kernel a bit-exact reference; three runs must be identical (a DMA / ring race shows up as run-to-run
    differences under load)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng(29)
    batch, cin, cout = 256, 256, 256
    x = rng.integers(0, 4, (batch, 19, 19, cin)).astype(np.float16)
    w = (rng.integers(-3, 4, (cout, 3, 3, cin)) / 16.0).astype(np.float16)
    b = (rng.integers(-8, 9, (cout,)) / 8.0).astype(np.float16)
    z = rng.integers(0, 4, (batch, 19, 19, cout)).astype(np.float16)
    monkeypatch.setenv("DGNN_CONV", "v1")
    want = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=0.5, alpha2=0.5, z=z, relu=False)
    monkeypatch.setenv("DGNN_CONV", "board")
    monkeypatch.setenv("DGNN_BOARD_NS", ns)
    for run in range(3):
        got = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=0.5, alpha2=0.5, z=z, relu=False)
        assert np.array_equal(got, want), (run, int(np.count_nonzero(got != want)))


def test_board_path_whole_network(api, oracle, synth, monkeypatch):
    """Whole forward pass on the board path (layout conversion, up layer, tower, heads) at a small batch."""
    monkeypatch.setenv("DGNN_CONV", "board")
    text = synth.to_json(synth.make_tensors(3, 256, seed=303))
    features = synth.make_features(5, seed=3030)
    v1, p1 = _compare(api, oracle, text, features, 5, "board path 3x256 B=5")
    monkeypatch.setenv("DGNN_CONV", "tile2")
    net = api.Network.from_json(text)
    with net.get_workspace(5) as ws:
        v2, p2 = api.forward(ws, features).unwrap()
    # two different kernels, same fp32-accumulate contract: fp16 outputs agree to the tolerance
    assert np.abs(v1.astype(np.float32) - v2.astype(np.float32)).max() <= VALUE_TOL
    assert np.abs(p1.astype(np.float32) - p2.astype(np.float32)).max() <= POLICY_TOL


def test_conv_does_not_read_skip_when_alpha2_is_zero(api_full, oracle):
    """Conv2d::forward aliases z to the uninitialised output with alpha2 = 0 (conv2d.rs:188-190): z must be ignored."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1, (1, 19, 19, 32)).astype(np.float16)
    w = rng.normal(0, 0.1, (32, 3, 3, 32)).astype(np.float16)
    b = np.zeros(32, np.float16)
    nan = np.full((1, 19, 19, 32), np.nan, dtype=np.float16)
    a = api.layer_conv3x3(x, 1, 19, 32, w, b, 32, alpha1=1.0, alpha2=0.0, z=nan)
    c = api.layer_conv3x3(x, 1, 19, 32, w, b, 32, alpha1=1.0, alpha2=0.0, z=None)
    assert np.array_equal(a, c) and not np.isnan(f16_to_f32(a)).any()


def test_conv_linearity(api_full):
    """Size-independent property: conv(x1 + x2) == conv(x1) + conv(x2) for exactly representable small integers."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng(8)
    cin, cout, batch = 64, 64, 4
    x1 = rng.integers(0, 3, (batch, 19, 19, cin)).astype(np.float16)
    x2 = rng.integers(0, 3, (batch, 19, 19, cin)).astype(np.float16)
    w = (rng.integers(-2, 3, (cout, 3, 3, cin)) / 8.0).astype(np.float16)
    b = np.zeros(cout, np.float16)
    run = lambda x: f16_to_f32(api.layer_conv3x3(x, batch, 19, cin, w, b, cout, relu=False))
    assert np.array_equal(run(x1 + x2), run(x1) + run(x2))     # all sums are exact in fp32 and fp16 here


# --------------------------------------------------------------------------------------------- whole network

def _compare(api, oracle, text, features, batch, label, oracle_boards=None):
    net = api.Network.from_json(text) if text.lstrip().startswith("{") else api.Network.load(text)
    with net.get_workspace(batch) as ws:
        value, policy = api.forward(ws, features).unwrap()
    tensors, _ = oracle.load_weights(text)
    onet = oracle.OracleNetwork(tensors)
    boards = list(range(batch)) if oracle_boards is None else oracle_boards
    f = np.ascontiguousarray(features).view(np.uint16).reshape(batch, 11552)
    rv, rp = onet.forward(f[boards].reshape(-1), len(boards))
    rv, rp = f16_to_f32(rv), f16_to_f32(rp).reshape(len(boards), 362)
    gv, gp = value.astype(np.float32)[boards], policy.astype(np.float32).reshape(batch, 362)[boards]
    dv, dp = float(np.abs(gv - rv).max()), float(np.abs(gp - rp).max())
    kl = policy_kl(rp, gp)
    print(f"{label}: max|dvalue| {dv:.3e}  max|dpolicy| {dp:.3e}  max KL {kl.max():.3e}  (boards checked: {len(boards)})")
    assert dv <= VALUE_TOL and dp <= POLICY_TOL, (dv, dp)
    assert kl.max() <= 1e-4
    sums = policy.astype(np.float32).reshape(batch, 362).sum(axis=1)
    assert np.all(np.abs(sums - 1.0) < 5e-3)
    assert np.all(np.abs(value.astype(np.float32)) <= 1.0)
    net.destroy()
    return value, policy


@pytest.mark.parametrize("name", ["net_2x32_s2", "net_1x128_s2"])
def test_golden_fixture(api, name):
    """HIP path against the committed fixture (no oracle needed at run time)."""
    case = np.load(os.path.join(GOLDEN, name + "_case.npz"))
    batch = int(case["batch"])
    net = api.Network.load(os.path.join(GOLDEN, name + ".json"))
    with net.get_workspace(batch) as ws:
        value, policy = api.forward(ws, case["features"]).unwrap()
    dv = np.abs(value.astype(np.float32) - f16_to_f32(case["value"])).max()
    dp = np.abs(policy.astype(np.float32) - f16_to_f32(case["policy"])).max()
    print(f"{name}: max|dvalue| {dv:.3e} max|dpolicy| {dp:.3e}")
    assert dv <= VALUE_TOL and dp <= POLICY_TOL


@pytest.mark.parametrize("blocks,channels,batch", [
    (6, 64, 8),      # BASELINE.json configs[0] shape (the "plumbing" net), batch 8
    (6, 64, 16),
    (9, 128, 32),    # configs[1]: 9-block / 128-filter, batch 32
    (2, 256, 4),     # a slice of configs[2]
    (0, 32, 1),      # no residual block at all, batch 1 (edge case: empty tower)
    (1, 32, 7),      # ragged: 7 * 361 positions is not a multiple of any tile
])
def test_network_against_oracle(api, oracle, synth, blocks, channels, batch):
    text = synth.to_json(synth.make_tensors(blocks, channels, seed=100 + blocks))
    features = synth.make_features(batch, seed=blocks * 1000 + batch)
    _compare(api, oracle, text, features, batch, f"{blocks}x{channels} B={batch}")


@pytest.mark.parametrize("blocks,channels,samples,batch", [
    (2, 192, 8, 6),      # 6 slices of 32 channels: only 2 slices per workgroup divide it
    (1, 64, 16, 3),      # the widest policy head the loader accepts (S = 16: 18 of the 32 head-conv rows in use)
    (1, 64, 4, 2),       # S = 4: dense input 1444, not a multiple of the GEMM's k-step
    (3, 128, 8, 300),    # more boards than CUs, not a multiple of anything
])
def test_network_shapes_on_the_board_path(api, oracle, synth, blocks, channels, samples, batch):
    text = synth.to_json(synth.make_tensors(blocks, channels, num_samples=samples, seed=500 + channels + samples))
    features = synth.make_features(batch, seed=channels * 7 + batch)
    boards = None if batch <= 8 else [0, 1, batch // 2, batch - 1]
    _compare(api, oracle, text, features, batch, f"{blocks}x{channels} S={samples} B={batch}", oracle_boards=boards)


def test_full_size_network_sampled_boards(api, oracle, synth):
    """BASELINE.json configs[2] at full size (20 x 256, batch 256).  Leaf evaluations are independent, so the
    oracle evaluates a sample of the boards (first, last, two in the middle) and every one must match."""
    blocks, channels, batch = 20, 256, 256
    text = synth.to_json(synth.make_tensors(blocks, channels, seed=2020))
    features = synth.make_features(batch, seed=77)
    value, policy = _compare(api, oracle, text, features, batch, "20x256 B=256", oracle_boards=[0, 101, 200, 255])
    # size-independent property on ALL boards: a board's result does not depend on its batch neighbours
    f = features.reshape(batch, 11552)
    perm = np.random.default_rng(1).permutation(batch)
    net = api.Network.from_json(text)
    with net.get_workspace(batch) as ws:
        v2, p2 = api.forward(ws, f[perm].reshape(-1)).unwrap()
    assert np.array_equal(v2.view(np.uint16), value.view(np.uint16)[perm])
    assert np.array_equal(p2.view(np.uint16).reshape(batch, 362), policy.view(np.uint16).reshape(batch, 362)[perm])


def test_every_batch_size_up_to_sixteen_and_pooling(api, oracle, synth):
    """The pool sees any batch size 1..max (worker_thread.rs:131-133, batch.rs:109-117): same board, any batch."""
    text = synth.to_json(synth.make_tensors(2, 64, seed=31))
    net = api.Network.from_json(text)
    one = synth.make_features(1, seed=5)
    with net.get_workspace(1) as ws:
        v1, p1 = api.forward(ws, one).unwrap()
    for batch in list(range(1, 17)) + [33]:
        with net.get_workspace(batch) as ws:
            v, p = api.forward(ws, np.tile(one, batch)).unwrap()
            handle = ws.handle.value
        assert np.all(v.view(np.uint16) == v1.view(np.uint16)[0])
        assert np.all(p.view(np.uint16).reshape(batch, 362) == p1.view(np.uint16)[None, :])
        with net.get_workspace(batch) as ws:          # a released workspace is reused (network.rs:137-140)
            assert ws.handle.value == handle
    net.synchronize()                                  # network.rs:145-159: drains the pool
    with net.get_workspace(4) as ws:
        api.forward(ws, np.tile(one, 4))
    net.destroy()


def test_forward_argument_checks(api, synth):
    net = api.Network.from_json(synth.to_json(synth.make_tensors(0, 32, seed=1)))
    with net.get_workspace(2) as ws:
        with pytest.raises(api.Error) as e:            # graph.rs:124-125 debug_assert
            api.forward(ws, synth.make_features(3))
        assert e.value.status == api.BAD_ARGUMENT
        api.forward(ws, synth.make_features(2))
    with pytest.raises(api.Error) as e:
        net.get_workspace(1, device=api.Device.len())
    assert e.value.status == api.BAD_ARGUMENT


def test_softmax_temperature_changes_policy_only(api, oracle, synth):
    text = synth.to_json(synth.make_tensors(1, 32, seed=9))
    features = synth.make_features(2, seed=3)
    tensors, _ = oracle.load_weights(text)
    for temperature in (0.709888, 1.0, 2.5):
        net = api.Network.from_json(text)
        net.set_softmax_temperature(temperature)
        with net.get_workspace(2) as ws:
            value, policy = api.forward(ws, features).unwrap()
        rv, rp = oracle.OracleNetwork(tensors, softmax_temperature=temperature).forward(features, 2)
        assert np.abs(value.astype(np.float32) - f16_to_f32(rv)).max() <= VALUE_TOL
        assert np.abs(policy.astype(np.float32) - f16_to_f32(rp)).max() <= POLICY_TOL
        with pytest.raises(api.Error):                 # baked into the device image: too late to change
            net.set_softmax_temperature(1.0)


def test_concurrent_workspaces_from_threads(api, synth):
    """predict() is called concurrently from pool workers and game threads (predictors/nn.rs:64-67, lib.rs:114)."""
    import threading
    text = synth.to_json(synth.make_tensors(2, 64, seed=13))
    net = api.Network.from_json(text)
    feats = {b: synth.make_features(b, seed=40 + b) for b in (1, 3, 8)}
    with net.get_workspace(8) as ws:
        expect = {b: None for b in feats}
    for b in feats:
        with net.get_workspace(b) as ws:
            expect[b] = [a.view(np.uint16).copy() for a in api.forward(ws, feats[b]).unwrap()]
    errors = []

    def worker(b, rounds):
        try:
            for _ in range(rounds):
                with net.get_workspace(b) as ws:
                    v, p = api.forward(ws, feats[b]).unwrap()
                if not (np.array_equal(v.view(np.uint16), expect[b][0]) and np.array_equal(p.view(np.uint16), expect[b][1])):
                    errors.append(f"mismatch at batch {b}")
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(b, 20)) for b in (1, 3, 8, 1, 3, 8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors[:3]
    net.destroy()


def test_resident_and_timed_paths_match_host_path(api, synth):
    text = synth.to_json(synth.make_tensors(2, 64, seed=17))
    net = api.Network.from_json(text)
    features = synth.make_features(5, seed=1)
    with net.get_workspace(5) as ws:
        v0, p0 = api.forward(ws, features).unwrap()
        ws.upload(features)
        ws.forward_resident(); ws.sync()
        out = ws.download()
        assert np.array_equal(out.value.view(np.uint16), v0.view(np.uint16))
        assert np.array_equal(out.policy.view(np.uint16), p0.view(np.uint16))
        total, tower, launches = ws.forward_timed(3)
        # one launch per pass where the whole tower is one kernel (64 channels: a workgroup owns a board), else one per convolution
        assert launches == (3 if ws.conv_kernel() in ("tower_board_kernel", "tower_split_kernel") else 2 * 2 * 3) and 0 < tower <= total
        out = ws.download()
        assert np.array_equal(out.policy.view(np.uint16), p0.view(np.uint16))


# --------------------------------------------------------------------------------------------- host side (f-1) on the GPU

def test_partial_batches_from_one_workspace(api, synth):
    """dgnn_forward_n: every count 1..capacity from ONE workspace gives the results of a dedicated workspace."""
    text = synth.to_json(synth.make_tensors(2, 64, seed=41))
    net = api.Network.from_json(text)
    feats = synth.make_features(9, seed=410).reshape(9, 11552)
    with net.get_workspace(9) as ws:
        full_v, full_p = api.forward(ws, feats.reshape(-1)).unwrap()
        for count in (1, 2, 5, 8, 9):
            v, p = api.forward_n(ws, feats[:count].reshape(-1), count).unwrap()
            assert np.array_equal(v.view(np.uint16), full_v.view(np.uint16)[:count])
            assert np.array_equal(p.view(np.uint16), full_p.view(np.uint16)[:count * 362])
        with pytest.raises(api.Error):
            api.forward_n(ws, feats.reshape(-1)[:0], 0)
        again_v, again_p = api.forward(ws, feats.reshape(-1)).unwrap()          # the full size still works afterwards
        assert np.array_equal(again_p.view(np.uint16), full_p.view(np.uint16))
    net.destroy()


def test_batcher_forward_matches_direct_forward(api_full, synth):
    """Batch::forward (batch.rs:35-40): boards evaluated through the batcher, in batches of whatever size the queue
    had, give bit-identical results to evaluating each board alone."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    text = synth.to_json(synth.make_tensors(2, 64, seed=43))
    net = api.Network.from_json(text)
    feats = synth.make_features(23, seed=430).reshape(23, 11552)
    with net.get_workspace(1) as ws:
        alone = [api.forward(ws, feats[i]).unwrap() for i in range(23)]
    b = api.Batcher(net, max_batch_size=8, max_batches=2)
    for i in range(23):
        b.push(1000 + i, feats[i])
    got = {}
    while True:
        batch = b.get_batch(1)
        if batch is None:
            break
        tags, out = batch.forward()
        v, p = out.unwrap()
        for j, tag in enumerate(tags.tolist()):
            got[tag] = (v[j], p[j * 362:(j + 1) * 362])
    assert sorted(got) == list(range(1000, 1023))
    for i in range(23):
        assert got[1000 + i][0].view(np.uint16) == alone[i][0].view(np.uint16)[0]
        assert np.array_equal(got[1000 + i][1].view(np.uint16), alone[i][1].view(np.uint16))
    assert b.stats() == {"boards": 23, "batches": 3} and b.pending() == (0, 0)


def test_leaf_pool_coalesces_callers(api_full, synth):
    """Many threads call predict() with <= 8 boards each (the shape of full_forward, lib.rs:83-130): everyone gets the
    result of evaluating its boards alone, and the device saw fewer, larger batches than there were calls."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    import threading
    text = synth.to_json(synth.make_tensors(2, 64, seed=47))
    net = api.Network.from_json(text)
    feats = synth.make_features(8, seed=470).reshape(8, 11552)
    with net.get_workspace(8) as ws:
        ref_v, ref_p = api.forward(ws, feats.reshape(-1)).unwrap()
    batcher = api.Batcher(net, max_batch_size=64, max_batches=2)
    pool = api.LeafPool(batcher, num_workers=2)
    errors, calls = [], 24 * 10

    def game(g):
        try:
            for i in range(10):
                n = 1 + (g + i) % 8
                preds = pool.predict(feats[:n].reshape(-1), n)
                for j, pr in enumerate(preds):
                    if not (np.float16(pr.value()).view(np.uint16) == ref_v.view(np.uint16)[j]
                            and np.array_equal(pr.policy().astype(np.float16).view(np.uint16), ref_p.view(np.uint16)[j * 362:(j + 1) * 362])):
                        errors.append((g, i, j))
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=game, args=(g,)) for g in range(24)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    stats = batcher.stats()
    pool.close()
    assert not errors, errors[:3]
    assert stats["boards"] == sum(1 + (g + i) % 8 for g in range(24) for i in range(10))
    print(f"leaf pool: {calls} calls, {stats['boards']} boards in {stats['batches']} device batches")
    # how much is coalesced depends on timing (a worker may even take a batch between two boards of one call); correctness does not
    assert stats["batches"] <= stats["boards"]


# --------------------------------------------------------------------------------------------- search (f-2) on the GPU

def test_search_visit_counts_hip_vs_oracle(api_full, oracle, synth):
    """north_star: MCTS visit counts under a fixed seed, HIP predictor against the CPU oracle as the predictor.  Single
    search thread, same seed, same synthetic game: the searches see the same positions as long as they choose the same
    children, so identical network outputs give identical counts; the outputs agree to fp16 rounding (<= 1e-3)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    text = synth.to_json(synth.make_tensors(2, 64, seed=71))
    net = api.Network.from_json(text)
    tensors, _ = oracle.load_weights(text)
    onet = oracle.OracleNetwork(tensors)
    oracle_fn = api.python_predictor(lambda f, n: onet.forward(f, n))

    def run(predictor, seed):
        root = api.Node(api.BLACK, 0.5, np.full(362, 1.0 / 362, dtype=np.float32))
        api.search_run(root, api.SyntheticBoard(99), num_rollouts=64, num_threads=1, seed=seed, pool=predictor)
        return np.array([root.with_(i).count() for i in range(362)])

    batcher = api.Batcher(net, max_batch_size=8, max_batches=1)
    pool = api.LeafPool(batcher, num_workers=1)
    hip_a, hip_b = run(pool, 3), run(pool, 3)
    ref = run(oracle_fn, 3)
    pool.close()
    assert np.array_equal(hip_a, hip_b)                    # same engine, same seed: bit-identical visit counts
    moved = int(np.abs(hip_a - ref).sum()) // 2
    print(f"search 64 rollouts: {moved} of 64 visits differ between the HIP predictor and the oracle predictor")
    assert 0 < hip_a.sum() <= 64 and 0 < ref.sum() <= 64   # a search may end before its limit (EARLY-C, time_control/mod.rs:86-101)
    assert moved == 0                                      # north_star: bit-identical visit counts (more seeds: tests/test_round2_gpu.py)


def test_synthetic_selfplay_through_the_pool(api_full, synth):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    text = synth.to_json(synth.make_tensors(2, 64, seed=73))
    net = api.Network.from_json(text)
    batcher = api.Batcher(net, max_batch_size=64, max_batches=2)
    pool = api.LeafPool(batcher, num_workers=2)
    stats = api.selfplay_synthetic(num_games=8, num_moves=3, num_rollouts=40, threads_per_search=4, seed=5, pool=pool)
    pooled = batcher.stats()
    pool.close()
    assert stats["moves"] == 24 and stats["evals"] == pooled["boards"]
    assert pooled["batches"] <= pooled["boards"]           # how much is coalesced depends on timing
    print(f"synthetic self-play: {stats['evals']} evals in {pooled['batches']} device batches, {stats['conflicts']} probe conflicts")


# --------------------------------------------------------------------------------------------- the Go rows (f-3) on the GPU

def test_real_board_features_through_the_engine(api, oracle, synth):
    """The input contract end to end: feature planes of real positions (all eight symmetries), not synthetic
    bits, through the HIP engine and through the oracle."""
    text = synth.to_json(synth.make_tensors(2, 64, seed=81))
    rng = np.random.default_rng(2)
    board, color, planes = api.Board(6.5), 0, []
    for move in range(96):
        for index in rng.permutation(361):
            if board.is_valid(color, int(index % 19), int(index // 19)):
                board.place(color, int(index % 19), int(index // 19))
                break
        color ^= 1
        if move % 12 == 11:
            planes.append(board.get_features(color, symmetry=len(planes) % 8))
    features = np.stack(planes).astype(np.float16).reshape(-1)
    assert len(planes) == 8
    _compare(api, oracle, text, features, len(planes), "real-board feature planes (8 positions, 8 symmetries)")


def test_go_selfplay_through_the_pool(api_full, synth):
    """Board + ladders + feature planes + search + transposition cache + pool + HIP engine in one piece: every stone legal,
    every evaluation accounted for by the batcher."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    text = synth.to_json(synth.make_tensors(2, 64, seed=83))
    net = api.Network.from_json(text)
    batcher = api.Batcher(net, max_batch_size=64, max_batches=2)
    pool = api.LeafPool(batcher, num_workers=2)
    out = api.selfplay_go(num_games=6, num_moves=10, num_rollouts=40, threads_per_search=4, seed=5, pool=pool,
                          cache=api.LruCache(50_000, 4), return_moves=True, return_sgf=True)
    pooled = batcher.stats()
    pool.close()
    assert out["moves"] == 60 and out["evals"] == pooled["boards"] and out["cache_hits"] > 0
    for game, record in zip(out["sgf"], out["record"]):
        replay = api.Board(7.5)
        for m, index in enumerate(record):
            assert index >= 0 and replay.is_valid(m % 2, int(index % 19), int(index // 19))
            replay.place(m % 2, int(index % 19), int(index // 19))
        assert game.count("TV[") == 10 and game.count("P[") == 10
    print(f"go self-play: {out['evals']} evals in {pooled['batches']} device batches, {out['cache_hits']} cache hits")


def test_selfplay_on_the_engine_is_a_function_of_the_seed(api_full, synth):
    """north_star: visit counts bit-identical under a fixed seed -- here for whole self-play games on the HIP engine: one search
    thread per game with 8 leaves in flight, games coalesced by the pool into device batches whose composition depends on timing
    and on how many games run at a time.  The engine's result for a board does not depend on what else is in the batch (a board is
    one workgroup, the contraction order is fixed), so the records must not depend on it either."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    text = synth.to_json(synth.make_tensors(3, 64, seed=85))
    net = api.Network.from_json(text)
    records = []
    for parallel, max_batch in ((0, 64), (2, 16), (0, 7)):
        batcher = api.Batcher(net, max_batch_size=max_batch, max_batches=2)
        pool = api.LeafPool(batcher, num_workers=2)
        out = api.selfplay_go(num_games=6, num_moves=12, num_rollouts=48, threads_per_search=1, leaves_per_thread=8, seed=23, pool=pool,
                              num_parallel=parallel, return_moves=True)
        stats = batcher.stats()
        pool.close()
        records.append(out["record"])
        print(f"games in flight {parallel or 6}, max batch {max_batch}: {out['evals']} evals in {stats['batches']} device batches")
    assert np.array_equal(records[0], records[1]) and np.array_equal(records[0], records[2])


# --------------------------------------------------------------------------------------------- the tower as one launch

@pytest.mark.parametrize("blocks,channels,batch", [(3, 64, 5), (2, 128, 200), (2, 256, 192), (3, 64, 256)])
def test_tower_kernel_equals_one_launch_per_layer(api, synth, blocks, channels, batch, monkeypatch):
    """tower_board_kernel walks all layers of a board in one workgroup (no grid-wide synchronisation is needed: a board's
    convolutions depend on that board only).  It runs the same code per layer as conv3x3_board_kernel, so the outputs must be
    bit-identical to the per-layer launches (DGNN_TOWER=0), and it must only be chosen where a workgroup owns a whole board."""
    text = synth.to_json(synth.make_tensors(blocks, channels, seed=91))
    features = synth.make_features(batch, seed=92)
    outputs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DGNN_TOWER", mode)
        net = api.Network.from_json(text)
        with net.get_workspace(batch) as ws:
            outputs[mode] = (ws.conv_kernel(), api.forward(ws, features).unwrap())
        net.destroy()
    assert outputs["1"][0] == "tower_board_kernel" and outputs["0"][0] == "conv3x3_board_kernel"
    assert np.array_equal(outputs["1"][1][0].view(np.uint16), outputs["0"][1][0].view(np.uint16))
    assert np.array_equal(outputs["1"][1][1].view(np.uint16), outputs["0"][1][1].view(np.uint16))


def test_small_batches_split_a_board_inside_the_tower_launch(api, synth, monkeypatch):
    net = api.Network.from_json(synth.to_json(synth.make_tensors(2, 256, seed=93)))
    with net.get_workspace(8) as ws:                       # 8 boards of 256 channels: eight workgroups per board, ONE launch
        assert ws.conv_kernel() == "tower_split_kernel"
    with net.get_workspace(160) as ws:                     # a split would need more workgroups than CUs: a workgroup per board
        assert ws.conv_kernel() == "tower_board_kernel"
    net.destroy()
    monkeypatch.setenv("DGNN_TOWER_SPLIT", "0")
    net = api.Network.from_json(synth.to_json(synth.make_tensors(2, 256, seed=93)))
    with net.get_workspace(8) as ws:
        assert ws.conv_kernel() == "conv3x3_board_kernel"
    net.destroy()


@pytest.mark.parametrize("blocks,channels,batch,split", [
    (3, 256, 128, "1"), (3, 256, 100, "1"), (2, 256, 64, "1"), (2, 256, 37, "1"), (3, 256, 20, "1"), (2, 256, 1, "1"),
    (4, 128, 128, "1"), (4, 128, 70, "1"), (3, 128, 40, "1"), (3, 128, 3, "1"), (3, 64, 50, "2"), (2, 64, 128, "2"),
    # the row split (a channel block's rows over two workgroups: batch * C / 32 * 2 <= 256 CUs) at and around its limits
    (3, 256, 16, "1"), (2, 256, 9, "1"), (3, 256, 17, "1"), (3, 128, 32, "1"), (4, 128, 17, "1"), (3, 128, 33, "1")])
def test_split_tower_equals_one_launch_per_layer(api, synth, blocks, channels, batch, split, monkeypatch):
    """tower_split_kernel: the workgroups of a board hand their channel blocks to each other inside one launch (per-board
    flags, write-through stores, L1-bypassing loads).  Same per-layer code, same contraction order as conv3x3_board_kernel:
    the outputs must be bit-identical to one launch per layer, pass after pass (a hand-off that let a workgroup read a
    channel block too early would show up as a mismatch on some pass).  Round 3: where one 32-channel slice per workgroup still
    leaves half the CUs idle the rows are split too (tower_split_kernel<32>: flags per (board, block, row half), a consumer waits for
    its sibling's 21 halo rows as well)."""
    text = synth.to_json(synth.make_tensors(blocks, channels, seed=191))
    monkeypatch.setenv("DGNN_TOWER", "0")
    net = api.Network.from_json(text)
    reference = []
    with net.get_workspace(batch) as ws:
        assert ws.conv_kernel() == "conv3x3_board_kernel"
        for i in range(4):
            reference.append(api.forward(ws, synth.make_features(batch, seed=300 + i)).unwrap())
    net.destroy()
    monkeypatch.setenv("DGNN_TOWER", "1")
    monkeypatch.setenv("DGNN_TOWER_SPLIT", split)
    net = api.Network.from_json(text)
    with net.get_workspace(batch) as ws:
        assert ws.conv_kernel() == "tower_split_kernel"
        for rep in range(3):
            for i in range(4):
                value, policy = api.forward(ws, synth.make_features(batch, seed=300 + i)).unwrap()
                assert np.array_equal(value.view(np.uint16), reference[i][0].view(np.uint16)), (rep, i)
                assert np.array_equal(policy.view(np.uint16), reference[i][1].view(np.uint16)), (rep, i)
    net.destroy()


def test_one_workspace_serves_every_fill_level_with_the_same_bits(api, synth, monkeypatch):
    """A workspace of capacity 256 is handed 256, 150, 100, 40 and 5 boards (dgnn_forward_n): a workgroup per board, then a
    board over two, four and eight workgroups.  A board's outputs do not depend on which of these ran it."""
    text = synth.to_json(synth.make_tensors(2, 256, seed=195))
    features = synth.make_features(256, seed=196)
    net = api.Network.from_json(text)
    with net.get_workspace(256) as ws:
        full_value, full_policy = api.forward(ws, features).unwrap()
        for count in (150, 100, 40, 5):
            value, policy = api.forward_n(ws, features[: count * 361 * 32], count).unwrap()
            assert np.array_equal(value.view(np.uint16), full_value.view(np.uint16)[:count]), count
            assert np.array_equal(policy.view(np.uint16).reshape(count, 362), full_policy.view(np.uint16).reshape(256, 362)[:count]), count
    net.destroy()
