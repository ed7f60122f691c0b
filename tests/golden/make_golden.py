"""Regenerates the whole-network fixtures under tests/golden/ (run from the repo root):

    python tests/golden/make_golden.py

The reference cannot be executed (Rust + closed-source cuDNN), and it holds no golden vector
for a whole forward pass, so these fixtures are NOT reference outputs: they are outputs of the CPU oracle
(oracle/dgnn_oracle.c) on seeded synthetic networks written in the reference's weights format.  They pin the
oracle against regressions and give the GPU box (which has no /root/reference and runs the oracle only as a
checker) a committed input/expected-output pair.  Reference-held known answers are in reference_kats.json.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, blocks, channels, samples, weight seed, batch, feature seed, alpha
    ("net_2x32_s2", 2, 32, 2, 3, 3, 21, 0.5),
    ("net_1x128_s2", 1, 128, 2, 4, 2, 22, 0.375),
]


def main():
    for name, blocks, channels, samples, seed, batch, fseed, alpha in CASES:
        path = os.path.join(HERE, name + ".json")
        synth.write_network(path, blocks, channels, num_samples=samples, seed=seed, alpha=alpha)
        tensors, _ = oracle.load_weights(path)
        net = oracle.OracleNetwork(tensors)
        features = synth.make_features(batch, seed=fseed)
        # make the input less uniform than pure Bernoulli noise: board 0 gets constant planes like real features
        f = features.view(np.float16).reshape(batch, 361, 32)
        f[0, :, 0] = np.float16(0.75)
        f[0, :, 1] = np.float16(1.0)
        value, policy = net.forward(features, batch)
        np.savez_compressed(os.path.join(HERE, name + "_case.npz"), features=features, value=value, policy=policy,
                            batch=batch, blocks=blocks, channels=channels, samples=samples)
        print(name, os.path.getsize(path), "bytes;", "value", value.view(np.float16), "policy max",
              policy.view(np.float16).reshape(batch, 362).max(axis=1))


if __name__ == "__main__":
    main()
