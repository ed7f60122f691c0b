"""Committed whole-network fixtures (tests/golden/make_golden.py): the oracle must reproduce them bit for bit,
and the C-ABI loader must ingest the same weight files."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = ["net_2x32_s2", "net_1x128_s2"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_fixture(oracle, name):
    case = np.load(os.path.join(GOLDEN, name + "_case.npz"))
    tensors, _ = oracle.load_weights(os.path.join(GOLDEN, name + ".json"))
    net = oracle.OracleNetwork(tensors)
    assert (net.num_blocks, net.num_channels, net.num_samples) == (int(case["blocks"]), int(case["channels"]), int(case["samples"]))
    value, policy = net.forward(case["features"], int(case["batch"]))
    assert np.array_equal(value, case["value"])
    assert np.array_equal(policy, case["policy"])


@pytest.mark.parametrize("name", CASES)
def test_c_abi_loads_fixture(api, name):
    case = np.load(os.path.join(GOLDEN, name + "_case.npz"))
    net = api.Network.load(os.path.join(GOLDEN, name + ".json"))
    assert net.info() == {"num_blocks": int(case["blocks"]), "num_channels": int(case["channels"]),
                          "num_samples": int(case["samples"])}
