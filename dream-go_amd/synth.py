"""Synthetic ("random-init") networks and inputs in the reference's own formats.

There is no network access for trained weights, so the harness generates a network of the requested
architecture and writes it in exactly the schema the reference's trainer dumps
(``contrib/trainer/dream_tf/hooks/dump.py:34-67``): one JSON object, per tensor
``{"s": b85(f32 max-abs), "t": dtype tag, "v": b85(raw little-endian bytes)}`` with RFC-1924 base85 and
zero padding to a multiple of four bytes, so the engine's loader is exercised on the real file format.

Tensor names and layouts follow the trainer's definition of the network:
``layers/tower.py:31-71`` (scopes ``01_upsample``, ``{02+i}_residual``, ``{2+blocks}p_policy``,
``{2+blocks}v_value``; meta tensors ``num_blocks:0``/``num_channels:0``/``num_samples:0`` as ``i4``),
``layers/batch_norm.py:55-75`` (conv filters transposed to [out, h, w, in], offsets ``.../offset:0``),
``layers/residual_block.py:42-45`` (``alpha:0`` as ``f4``, initial value 0.5) and ``layers/dense.py:28-39``
(matrices stored [in, out]).

Input features follow the distribution of the reference's own forward benchmark
(``src/dg_tests/benches/batch_sizes.rs:44-52``): every element i.i.d. Bernoulli(0.2) in {0, 1}, fp16, HWC.
"""
import base64
import json
import zlib

import numpy as np

NUM_FEATURES = 32
NUM_POINTS = 361
NUM_MOVES = 362

FEATURE_SEED = 0xD6EA60


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode("ascii"))])


def _entry(array, tag):
    dtype = {"f2": "<f2", "f4": "<f4", "i4": "<i4"}[tag]
    array = np.asarray(array)
    max_abs = np.asarray(np.max(np.abs(array)) if array.size else 0.0).astype("<f4").tobytes()
    return {
        "s": base64.b85encode(max_abs, pad=True).decode("ascii"),
        "t": tag,
        "v": base64.b85encode(array.astype(dtype).tobytes(), pad=True).decode("ascii"),
    }


def make_tensors(num_blocks, num_channels, num_samples=8, seed=1, alpha=0.5):
    """Returns {name: (np.ndarray, tag)} for a random-init network.

    Conv filters ~ N(0, 2 / (9 * Cin)) (He), conv offsets ~ N(0, 0.01), dense matrices ~ N(0, 1 / in),
    policy offset ~ N(-6.5, 0.5) (log-probability scale, like ``policy_head.py:54-56``),
    value offset -0.00502319782 (``value_head.py:62-63``).
    """
    c, s = int(num_channels), int(num_samples)
    out = {
        "num_blocks:0": (np.array([num_blocks]), "i4"),
        "num_channels:0": (np.array([c]), "i4"),
        "num_samples:0": (np.array([s]), "i4"),
    }

    def conv(name, cout, cin):
        rng = _rng(seed, name)
        w = rng.normal(0.0, np.sqrt(2.0 / (9 * cin)), size=(cout, 3, 3, cin))   # [out, h, w, in]
        b = rng.normal(0.0, 0.01, size=(cout,))
        out[f"{name}:0"] = (w.reshape(-1), "f2")
        out[f"{name}/offset:0"] = (b, "f2")

    def dense(name, num_in, num_out, offset):
        rng = _rng(seed, name)
        w = rng.normal(0.0, np.sqrt(1.0 / num_in), size=(num_in, num_out))       # [in, out]
        out[f"{name}:0"] = (w.reshape(-1), "f2")
        out[f"{name}/offset:0"] = (np.asarray(offset, dtype=np.float64), "f2")

    conv("01_upsample/conv_1", c, NUM_FEATURES)
    for i in range(num_blocks):
        scope = f"{2 + i:02d}_residual"
        conv(f"{scope}/conv_1", c, c)
        conv(f"{scope}/conv_2", c, c)
        out[f"{scope}/alpha:0"] = (np.array([alpha]), "f4")

    head = 2 + num_blocks
    conv(f"{head:02d}p_policy/conv_1", s, c)
    dense(f"{head:02d}p_policy/linear_1", NUM_POINTS * s, NUM_MOVES,
          _rng(seed, "policy_offset").normal(-6.5, 0.5, size=(NUM_MOVES,)))
    conv(f"{head:02d}v_value/conv_1", 2, c)
    dense(f"{head:02d}v_value/linear_2", NUM_POINTS * 2, 1, [-0.00502319782])
    return out


def make_sharp_tensors(num_blocks, num_channels, num_samples=8, seed=1, alpha=0.1, policy_gain=24.0, value_gain=16.0):
    """A random-init network whose OUTPUTS discriminate between positions: the plain He-init net of
    make_tensors with alpha = 0.5 forgets its input over 20 blocks (policy maxima 0.013-0.03 on every board, boards of a batch 1e-4
    apart), so a kernel that mixed up boards or lost accuracy on peaked soft-maxes / saturated tanh would pass a 1e-3 test.  Here
    the residual gate is small (the skip path carries the input through the tower) and the two dense layers are scaled
    (`p_policy/linear_1` by policy_gain, `v_value/linear_2` by value_gain): on real positions the policy maximum then spans
    about 0.05 ... 0.99 with many different arg-maxima and the value reaches both ends of tanh (measured with the oracle:
    20x256 with (24, 16), 9x128 with (32, 6))."""
    out = make_tensors(num_blocks, num_channels, num_samples, seed, alpha)
    head = 2 + num_blocks
    for name, gain in ((f"{head:02d}p_policy/linear_1:0", policy_gain), (f"{head:02d}v_value/linear_2:0", value_gain)):
        array, tag = out[name]
        out[name] = (array * gain, tag)
    return out


def make_centered_tensors(num_blocks, num_channels, num_samples=8, seed=1, alpha=0.5, conv_gain=1.5, policy_gain=6.0, value_gain=4.0):
    """A discriminating random-init network AT THE REFERENCE'S DEFAULT GATE (alpha = 0.5, residual_block.rs:50):
    make_sharp_tensors keeps the input alive by making nine tenths of every block the skip path -- which also attenuates the
    convolution errors a parity test is after by ten per block.  What makes the He-init tower forget its input is the COMMON MODE:
    activations behind a ReLU are all positive, a random filter answers their mean with a large input-independent offset, and
    twenty blocks later that offset is all there is.  Here every residual filter is made zero-mean over its 9 * Cin inputs (the
    common mode is not amplified) and scaled by conv_gain (about 1.5 keeps the variance of the differential part stationary through
    `relu(0.5 conv + 0.5 t)`; measured with the oracle on the committed games: 20x256 with 1.5 / gains 6, 4: policy maxima
    0.06 ... 0.55, values -0.84 ... 0.88; with 1.55: maxima up to 0.99, values at both ends of tanh; 9x128 with 1.6 / gains 8, 1.5:
    0.10 ... 0.81, -0.99 ... 0.75).  Half of every block is convolution: its errors reach the outputs undamped."""
    out = make_tensors(num_blocks, num_channels, num_samples, seed, alpha)
    for name in list(out):
        if "_residual/conv_" in name and name.endswith(":0") and "offset" not in name:
            array, tag = out[name]
            w = array.reshape(num_channels, -1)
            out[name] = (((w - w.mean(axis=1, keepdims=True)) * conv_gain).reshape(-1), tag)
    head = 2 + num_blocks
    for name, gain in ((f"{head:02d}p_policy/linear_1:0", policy_gain), (f"{head:02d}v_value/linear_2:0", value_gain)):
        array, tag = out[name]
        out[name] = (array * gain, tag)
    return out


def to_json(tensors, model_name="synthetic"):
    doc = {name: _entry(array, tag) for name, (array, tag) in tensors.items()}
    doc["model_name:0"] = model_name     # string-valued entry, ignored by the loader (loader.rs:47)
    return json.dumps(doc, sort_keys=True)


def write_network(path, num_blocks, num_channels, num_samples=8, seed=1, alpha=0.5):
    text = to_json(make_tensors(num_blocks, num_channels, num_samples, seed, alpha),
                   model_name=f"synthetic-{num_blocks}x{num_channels}-seed{seed}")
    with open(path, "w") as f:
        f.write(text)
    return path


def make_features(batch, seed=FEATURE_SEED, p=0.2):
    """[batch * 361 * 32] fp16 bit patterns, HWC, i.i.d. Bernoulli(p) (batch_sizes.rs:44-52)."""
    rng = np.random.default_rng([int(seed), int(batch)])
    bits = rng.random(batch * NUM_POINTS * NUM_FEATURES) < p
    return np.where(bits, np.float16(1.0), np.float16(0.0)).astype(np.float16).view(np.uint16)


def flops_per_eval(num_blocks, num_channels, num_samples=8):
    """Algorithmic FLOPs of one leaf evaluation."""
    c, s = num_channels, num_samples
    tower = 361 * 2 * 9 * (NUM_FEATURES * c + 2 * num_blocks * c * c)
    heads = 361 * 2 * 9 * (s * c + 2 * c) + 2 * (NUM_POINTS * s) * NUM_MOVES + 2 * 722
    return tower + heads
