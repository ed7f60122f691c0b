"""Pins the CPU oracle to every known-answer test the reference holds for this path.

The vectors live in tests/golden/reference_kats.json (data only, with the reference file:line of each).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, f16_to_f32

with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
    KATS = json.load(f)


def test_conv2d_pick_middle_and_sum(oracle):
    k = KATS["conv2d_pick_middle_and_sum"]                       # conv2d.rs:252-291
    y = oracle.conv3x3(oracle.f16(k["input"]), k["batch"], k["board_size"], k["cin"],
                       oracle.f16(k["filter_kcrs_as_listed"]), oracle.f16(k["offset"]), k["cout"], relu=True)
    assert f16_to_f32(y).tolist() == [float(v) for v in k["expected_nhwc"]]


def test_dense_matmul_identity_8x8(oracle):
    k = KATS["dense_matmul_identity_8x8"]                        # dense.rs:243-275
    y = oracle.dense(oracle.f16(k["input"]), 1, 8, oracle.f16(np.eye(8).reshape(-1)), oracle.f16(k["offset"]), 8, relu=True)
    assert f16_to_f32(y).tolist() == [float(v) for v in k["expected"]]


def test_dense_matmul_identity_1x8(oracle):
    k = KATS["dense_matmul_identity_1x8"]                        # dense.rs:277-299
    y = oracle.dense(oracle.f16(k["input"]), 1, 8, oracle.f16(k["filter"]), oracle.f16(k["offset"]), 1, relu=False)
    assert f16_to_f32(y).tolist() == [0.0]


def test_dense_is_transposed_from_in_out(oracle):
    # dense.rs:79-96: the stored matrix is [in][out]; an asymmetric matrix detects a missing transpose
    w = np.arange(6, dtype=np.float32).reshape(3, 2)             # [in=3][out=2]
    x = np.array([1.0, 10.0, 100.0], dtype=np.float32)
    y = oracle.dense(oracle.f16(x), 1, 3, oracle.f16(w.reshape(-1)), oracle.f16([0, 0]), 2, relu=False)
    assert f16_to_f32(y).tolist() == (x @ w).tolist()


def test_b85(oracle):
    k = KATS["b85"]                                              # b85.rs:171-219
    for case in k["decode_f16"]:
        got = np.frombuffer(oracle.b85_decode(case["text"]), dtype="<f2").astype(np.float32)
        assert got.tolist() == [float(v) for v in case["expected"]]
    for case in k["decode_f32"]:
        assert np.frombuffer(oracle.b85_decode(case["text"]), dtype="<f4").tolist() == case["expected"]
    for case in k["encode_f16"]:
        assert oracle.b85_encode_f16(case["values"]) == case["expected"]
    for values in k["roundtrip_f16"]:
        got = np.frombuffer(oracle.b85_decode(oracle.b85_encode_f16(values)), dtype="<f2")
        assert got.tolist() == np.asarray(values, dtype=np.float32).astype(np.float16).tolist()
    assert oracle.b85_decode("NJ4N\"") is None                   # invalid character -> None (b85.rs:116-118)
    assert oracle.b85_decode("NJ4NyNJ") == oracle.b85_decode("NJ4Ny")   # partial trailing group dropped


def test_f16(oracle):
    k = KATS["f16"]                                              # fp16.rs:77-92
    L = oracle.lib()
    for bits, value in k["from_bits"]:
        assert np.float32(L.dgo_f16_to_f32(bits)) == np.float32(value)
    for value, bits in k["to_bits"]:
        assert L.dgo_f32_to_f16(np.float32(value)) == bits


def test_f16_conversion_matches_ieee_for_all_halfs(oracle):
    L = oracle.lib()
    all_bits = np.arange(65536, dtype=np.uint16)
    as_f32 = all_bits.view(np.float16).astype(np.float32)
    rng = np.random.default_rng(0)
    for bits in rng.choice(65536, size=3000, replace=False):
        v = L.dgo_f16_to_f32(int(bits))
        if np.isnan(as_f32[bits]):
            assert np.isnan(v)
        else:
            assert np.float32(v) == as_f32[bits]
    # f32 -> f16: random values, halfway cases, subnormals, overflow
    samples = np.concatenate([
        rng.normal(0, 1, 2000), rng.normal(0, 1e-6, 500), rng.normal(0, 7e4, 200),
        (as_f32[1:2000].astype(np.float64) + as_f32[0:1999].astype(np.float64)) / 2,    # exact ties
        [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, 6e-8],
    ]).astype(np.float32)
    with np.errstate(over="ignore"):
        expect = samples.astype(np.float16).view(np.uint16)
    for v, e in zip(samples, expect):
        assert L.dgo_f32_to_f16(v) == e, (v, e)


def test_loader(oracle):
    k = KATS["loader"]                                           # loader.rs:123-141
    with pytest.raises(oracle.MissingWeights):
        oracle.load_weights(k["empty_document_is_error"] + " ")
    tensors, scales = oracle.load_weights(k["document"])
    assert len(tensors) == k["expected_tensors"]
    name = "11v_value/linear_2/offset:0"
    assert np.float32(scales[name]) == np.float32(k["expected_scale"])
    assert tensors[name].nbytes == k["expected_size_in_bytes"]
    with pytest.raises(oracle.MalformedWeights):
        oracle.load_weights('{"a": {"t": "f8", "v": "00000"}}')
    with pytest.raises(oracle.MalformedWeights):
        oracle.load_weights('{"a": {"x": "00000"}}')


def test_scale_and_softmax_and_tanh(oracle):
    L = oracle.lib()
    import ctypes
    b = oracle.f16([1.0, -2.5, 0.333]).copy()
    L.dgo_scale(b.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), 3, 0.5)
    assert f16_to_f32(b).tolist() == (np.float32(0.5) * f16_to_f32(oracle.f16([1.0, -2.5, 0.333]))).astype(np.float16).astype(np.float32).tolist()
    x = oracle.f16(np.linspace(-4, 4, 362))
    y = np.empty(362, dtype=np.uint16)
    L.dgo_softmax(x.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), 1, 362, y.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
    xf = f16_to_f32(x).astype(np.float64)
    ref = np.exp(xf - xf.max()); ref /= ref.sum()
    assert np.all(np.abs(f16_to_f32(y) - ref) <= 1e-3 * ref + 1e-7)      # fp16 output rounding
    t = oracle.f16([-3.0, -0.5, 0.0, 0.25, 2.0]).copy()
    L.dgo_tanh(t.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), 5)
    assert np.abs(f16_to_f32(t) - np.tanh([-3.0, -0.5, 0.0, 0.25, 2.0])).max() < 1e-3


def test_conv_matches_torch_conv2d(oracle):
    """Cross-check of the C restatement against torch.nn.functional.conv2d (CPU, fp32), SURVEY.md 8c."""
    import torch
    rng = np.random.default_rng(3)
    batch, cin, cout = 2, 32, 24
    x = rng.normal(0, 1, (batch, 19, 19, cin)).astype(np.float16)
    w = rng.normal(0, 0.1, (cout, 3, 3, cin)).astype(np.float16)          # KRSC
    b = rng.normal(0, 0.1, (cout,)).astype(np.float16)
    z = rng.normal(0, 1, (batch, 19, 19, cout)).astype(np.float16)
    y = oracle.conv3x3(x, batch, 19, cin, w, b, cout, alpha1=0.5, alpha2=0.5, z=z, relu=True)
    xt = torch.from_numpy(x.astype(np.float32)).permute(0, 3, 1, 2)
    wt = torch.from_numpy(w.astype(np.float32)).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xt, wt, padding=1).permute(0, 2, 3, 1).numpy()
    ref = np.maximum(0.5 * ref + b.astype(np.float32) + 0.5 * z.astype(np.float32), 0.0)
    got = f16_to_f32(y).reshape(ref.shape)
    assert np.abs(got - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())


def test_c_restatement_matches_numpy_statement(oracle, synth):
    """Whole network: C restatement vs the independent numpy/GEMM statement of the same dataflow."""
    batch = 2
    text = synth.to_json(synth.make_tensors(2, 32, num_samples=4, seed=5))
    tensors, _ = oracle.load_weights(text)
    net = oracle.OracleNetwork(tensors)
    assert (net.num_blocks, net.num_channels, net.num_samples) == (2, 32, 4)
    feats = synth.make_features(batch, seed=9)
    v_c, p_c = net.forward(feats, batch)
    v_n, p_n = oracle.forward_numpy(net, feats, batch)
    assert np.abs(f16_to_f32(v_c) - f16_to_f32(v_n)).max() <= 1e-3
    assert np.abs(f16_to_f32(p_c) - f16_to_f32(p_n)).max() <= 1e-3
    sums = f16_to_f32(p_c).reshape(batch, 362).sum(axis=1)
    assert np.all(np.abs(sums - 1) < 5e-3)


def test_fast_convolution_is_the_plain_one(oracle):
    """oracle/dgnn_oracle.c states the convolution twice: as plain loops (dgo_conv3x3_plain: what one reads against conv2d.rs:171-220
    and convolution_bias_activation.rs:21-41) and register-tiled with hardware fp16 conversions on a zero-padded copy (dgo_conv3x3:
    what every parity test and the CPU baseline run).  Bit for bit the same on everything that distinguishes them: every fp16 value
    as an input (subnormals, -0, the largest finite ones), board edges, channel counts that are not multiples of 16 or 8, outputs
    that overflow to infinity or round into the subnormals, with and without the skip input and ReLU, two calls of different shapes
    in a row (the padded scratch buffer is reused), boards of 3 x 3 and 19 x 19."""
    rng = np.random.default_rng(606)
    finite = np.array([h for h in range(65536) if (h >> 10) & 31 != 31], dtype=np.uint16)        # every finite half
    cases = [(2, 19, 32, 48, 0.5, True), (1, 3, 8, 2, None, True), (3, 19, 16, 8, 0.25, False), (1, 19, 64, 40, None, False),
             (2, 5, 24, 17, 0.75, True), (1, 19, 256, 32, 0.5, True)]
    for batch, size, cin, cout, alpha, relu in cases + cases[:2]:
        n = batch * size * size * cin
        x = rng.choice(finite, size=n).astype(np.uint16)
        x[rng.random(n) < 0.5] = 0                                                                   # post-ReLU sparsity, and taps on real zeros
        scale = rng.choice([1e-3, 0.05, 1.0, 30.0])
        w = (rng.normal(0, scale, (cout, 3, 3, cin))).astype(np.float16).view(np.uint16)
        w.reshape(-1)[: min(200, w.size)] = rng.choice(finite, size=min(200, w.size))                # a sprinkle of extreme weights
        b = rng.choice(finite, size=cout).astype(np.uint16)
        z = rng.choice(finite, size=batch * size * size * cout).astype(np.uint16) if alpha is not None else None
        a1, a2 = (alpha, 1.0 - alpha) if alpha is not None else (1.0, 0.0)
        fast = oracle.conv3x3(x, batch, size, cin, w, b, cout, alpha1=a1, alpha2=a2, z=z, relu=relu)
        plain = oracle.conv3x3(x, batch, size, cin, w, b, cout, alpha1=a1, alpha2=a2, z=z, relu=relu, plain=True)
        same = fast == plain
        # (a NaN is a NaN in both -- infinity minus infinity among the extreme operands -- whatever its payload)
        nan = lambda v: ((v >> 10) & 31 == 31) & (v & 1023 != 0)
        assert np.all(same | (nan(fast) & nan(plain))), (batch, size, cin, cout, int(np.count_nonzero(~same)))
        assert np.count_nonzero((fast >> 10) & 31 == 31) < fast.size                                   # not everything overflowed

