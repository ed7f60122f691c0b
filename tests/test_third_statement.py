"""A THIRD statement of the network's composition, from a different source than the oracle's.

The oracle (oracle/dgnn_oracle.c) restates the reference's Rust inference path (src/libdg_nn/graph.rs:123-158 and the layer
files).  The reference holds no whole-network golden vector and its arithmetic is inside closed-source cuDNN, so the oracle's
reading of residual gating, flatten order and the heads is pinned by layer KATs only ("the pin is layer-deep").
This file states the SAME network a second way, from the TRAINER's definition of it -- the TensorFlow graph that produced the
weights the Rust path consumes -- in plain torch-CPU fp32, and checks the two readings against each other:

    contrib/trainer/dream_tf/layers/tower.py:54-71          up-layer, blocks, heads, and the variable scopes (tensor names)
    contrib/trainer/dream_tf/layers/batch_norm.py:27-75      conv 'SAME' NHWC; folded filter dumped as transpose(HWIO, [3, 0, 1, 2]) = OHWI
    contrib/trainer/dream_tf/layers/residual_block.py:48-59  relu(alpha * (conv_2 + b_2) + (1 - alpha) * x)
    contrib/trainer/dream_tf/layers/policy_head.py:43-51     relu(conv) -> reshape(-1, 361 * S) -> matmul [361 S, 362] + offset
    contrib/trainer/dream_tf/layers/value_head.py:44-57      relu(conv, 2 planes) -> reshape(-1, 722) -> matmul [722, 1] + offset -> tanh
    contrib/trainer/dream_tf/layers/dense.py:28-39           x @ W + b with W stored [in, out]

Only the inference-time soft-max temperature is not in the trainer's graph; it is taken from src/libdg_utils/config.rs:176-177.
A misreading shared by oracle and kernels (which alpha multiplies which operand, HWC vs CHW flatten, tap transposition, which
of b_2 / x is gated) would show up here as an O(0.1) difference; fp16 rounding of the layer outputs accounts for ~1e-3.
Nothing here reads /root/reference at run time.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

SOFTMAX_TEMPERATURE = 0.709888      # src/libdg_utils/config.rs:176-177


def trainer_graph_forward(tensors, features, batch, round_layers):
    """The trainer's tower() in inference mode on dumped (BN-folded) weights.  round_layers: round every layer output to fp16
    like the fp16 compute type does (cast_to_compute_type); False = the pure fp32 mathematics of the graph."""
    f32 = lambda name: torch.from_numpy(np.asarray(tensors[name]).astype(np.float32))
    c = int(tensors["num_channels:0"][0])
    s = int(tensors["num_samples:0"][0])
    r = (lambda t: t.half().float()) if round_layers else (lambda t: t)

    def conv(x, scope, cout, cin):
        w = f32(scope + ":0")[: cout * 9 * cin].reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)      # OHWI as dumped -> OIHW
        b = f32(scope + "/offset:0")[:cout]
        return F.conv2d(x, w, bias=None, padding=1), b.reshape(1, cout, 1, 1)                      # padding='SAME', stride 1

    x = torch.from_numpy(np.asarray(features).view(np.float16).astype(np.float32)).reshape(batch, 19, 19, 32).permute(0, 3, 1, 2)
    y, b = conv(x, "01_upsample/conv_1", c, 32)
    y = r(torch.relu(y + b))
    i = 2
    while f"{i:02d}_residual/conv_1:0" in tensors:
        scope = f"{i:02d}_residual"
        alpha = float(np.asarray(tensors[scope + "/alpha:0"]).view(np.float32)[0])
        t, b1 = conv(y, scope + "/conv_1", c, c)
        t = r(torch.relu(t + b1))
        t, b2 = conv(t, scope + "/conv_2", c, c)
        y = r(torch.relu(alpha * (t + b2) + (1.0 - alpha) * y))       # residual_block.py:56
        i += 1
    head = i
    p, pb = conv(y, f"{head:02d}p_policy/conv_1", s, c)
    p = r(torch.relu(p + pb)).permute(0, 2, 3, 1).reshape(batch, 361 * s)                           # NHWC reshape(-1, 361 * S)
    w = f32(f"{head:02d}p_policy/linear_1:0")[: 361 * s * 362].reshape(361 * s, 362)
    logits = p @ w + f32(f"{head:02d}p_policy/linear_1/offset:0")[:362]
    policy = torch.softmax(logits / SOFTMAX_TEMPERATURE, dim=1)
    v, vb = conv(y, f"{head:02d}v_value/conv_1", 2, c)
    v = r(torch.relu(v + vb)).permute(0, 2, 3, 1).reshape(batch, 722)
    w = f32(f"{head:02d}v_value/linear_2:0")[:722].reshape(722, 1)
    value = torch.tanh(v @ w + f32(f"{head:02d}v_value/linear_2/offset:0")[:1])
    return value.reshape(batch).numpy(), policy.numpy()


@pytest.mark.parametrize("blocks,channels,samples,batch,alpha", [
    (2, 32, 2, 3, 0.25),       # asymmetric gate: alpha and 1 - alpha cannot be swapped unnoticed
    (3, 64, 8, 2, 0.7),
    (6, 64, 8, 2, 0.5),        # BASELINE.json configs[0] shape
    (0, 32, 4, 2, 0.5),        # no residual block
])
def test_oracle_agrees_with_the_trainers_statement(oracle, synth, blocks, channels, samples, batch, alpha):
    torch.set_num_threads(4)
    text = synth.to_json(synth.make_tensors(blocks, channels, num_samples=samples, seed=900 + blocks + channels, alpha=alpha))
    tensors, _ = oracle.load_weights(text)
    rng = np.random.default_rng([blocks, channels, batch])
    f = (rng.random((batch, 361, 32)) < 0.2).astype(np.float16)
    f[:, :, 0] = np.float16(0.9333)                                      # a non-binary plane like the komi plane
    features = f.reshape(-1).view(np.uint16)

    ov, op = oracle.OracleNetwork(tensors).forward(features, batch)
    ov = ov.view(np.float16).astype(np.float32)
    op = op.view(np.float16).astype(np.float32).reshape(batch, 362)

    tv, tp = trainer_graph_forward(tensors, features, batch, round_layers=True)
    assert np.abs(ov - tv).max() <= 1e-3, np.abs(ov - tv).max()          # the north star's tolerance
    assert np.abs(op - tp).max() <= 1e-3, np.abs(op - tp).max()
    assert np.array_equal(op.argmax(axis=1), tp.argmax(axis=1))

    pv, pp = trainer_graph_forward(tensors, features, batch, round_layers=False)     # the graph's pure fp32 mathematics
    assert np.abs(ov - pv).max() <= 4e-3 and np.abs(op - pp).max() <= 4e-3


def test_the_statement_detects_a_misreading(oracle, synth):
    """The check has teeth: swapping the gate's operands moves the outputs an order of magnitude further from the oracle than
    the faithful statement is (a random-init policy is nearly uniform, ~3e-3 per move, so the comparison is relative)."""
    text = synth.to_json(synth.make_tensors(2, 32, num_samples=2, seed=77, alpha=0.25))
    tensors, _ = oracle.load_weights(text)
    features = synth.make_features(2, seed=78)
    ov, op = oracle.OracleNetwork(tensors).forward(features, 2)
    op = op.view(np.float16).astype(np.float32).reshape(2, 362)
    ov = ov.view(np.float16).astype(np.float32)
    fv, fp = trainer_graph_forward(tensors, features, 2, round_layers=True)
    faithful = max(np.abs(op - fp).max() / op.max(), np.abs(ov - fv).max())
    swapped = dict(tensors)
    for i in (2, 3):
        swapped[f"{i:02d}_residual/alpha:0"] = np.array([0.75], np.float32)         # alpha <-> 1 - alpha
    sv, sp = trainer_graph_forward(swapped, features, 2, round_layers=True)
    misread = max(np.abs(op - sp).max() / op.max(), np.abs(ov - sv).max())
    print(f"relative distance from the oracle: faithful {faithful:.2e}, gate operands swapped {misread:.2e}")
    assert faithful < 5e-3 and misread > 10 * faithful and misread > 5e-2
