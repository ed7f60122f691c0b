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

from oracle.torch_statement import trainer_graph_forward      # (the statement itself: oracle/torch_statement.py, test infrastructure)


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
