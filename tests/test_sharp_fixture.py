"""tests/golden/sharp_2x256_b8.npz: the parity contract on peaked outputs as a committed fixture.

  * CPU: the oracle reproduces both statements bit for bit from the regenerated weights (whose CRC the fixture holds);
  * -m gpu: the HIP path is held to the fixture WITHOUT the oracle at run time -- logits and pre-tanh value within ONE fp16 ulp
    (at the scale of the sum) of the fp32 oracle's, as near to the exact-arithmetic statement as the oracle is, probabilities and
    values within the per-entry allowance that bound implies (tests/parity_bounds.py).
"""
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, f16_to_f32
from parity_bounds import MAX_LOGIT_ULPS, MAX_PRE_TANH_ULPS, allowances, logit_unit, policy_kl, pre_tanh_unit


def load_case(synth):
    case = np.load(os.path.join(GOLDEN, "sharp_2x256_b8.npz"))
    text = synth.to_json(synth.make_sharp_tensors(int(case["blocks"]), int(case["channels"]), seed=int(case["seed"]),
                                                   policy_gain=float(case["policy_gain"]), value_gain=float(case["value_gain"])))
    crc = zlib.crc32(text.encode("ascii") if isinstance(text, str) else text)
    assert crc == int(case["weights_crc32"]), "synth.make_sharp_tensors no longer produces the fixture's weights (numpy generator changed?): regenerate with tests/golden/make_sharp_fixture.py"
    return case, text


def test_oracle_reproduces_the_sharpened_fixture(oracle, synth):
    case, text = load_case(synth)
    batch = int(case["batch"])
    tensors, _ = oracle.load_weights(text)
    net = oracle.OracleNetwork(tensors)
    for prefix, exact in (("", False), ("exact_", True)):
        value, policy, logits, pre = net.forward_ex(case["features"], batch, accumulate_f64=exact)
        assert np.array_equal(value, case[prefix + "value"]) and np.array_equal(policy, case[prefix + "policy"])
        assert np.array_equal(logits, case[prefix + "logits"]) and np.array_equal(pre, case[prefix + "value_pre"])
    # the fixture is worth the name: peaked and flat boards, several best moves, and the two statements do differ
    p = f16_to_f32(case["policy"]).reshape(batch, 362)
    assert p.max(axis=1).max() >= 3 * p.max(axis=1).min() and len(set(p.argmax(axis=1).tolist())) >= 5
    assert int((case["logits"] != case["exact_logits"]).sum()) > 100


@pytest.mark.gpu
def test_hip_meets_the_contract_of_the_sharpened_fixture(api_full, synth):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    case, text = load_case(synth)
    batch = int(case["batch"])
    net = api.Network.from_json(text)
    with net.get_workspace(batch) as ws:
        value, policy = api.forward(ws, case["features"]).unwrap()
        logits, pre = ws.download_pre()
    net.destroy()
    gv, gp = value.astype(np.float32), policy.astype(np.float32).reshape(batch, 362)
    gl, gpre = logits.astype(np.float32).reshape(batch, 362), pre.astype(np.float32)
    rv, rp = f16_to_f32(case["value"]), f16_to_f32(case["policy"]).reshape(batch, 362)
    rl, rpre = f16_to_f32(case["logits"]).reshape(batch, 362), f16_to_f32(case["value_pre"])
    xl, xpre = f16_to_f32(case["exact_logits"]).reshape(batch, 362), f16_to_f32(case["exact_value_pre"])
    lu, pu = logit_unit(rl), pre_tanh_unit(rpre)
    dl, dpre = np.abs(gl - rl) / lu, np.abs(gpre - rpre) / pu
    hip_x, ora_x = np.abs(gl - xl) / lu, np.abs(rl - xl) / lu
    print(f"logits HIP vs oracle: max {dl.max():.2f} ulp, {100 * (dl > 0).mean():.1f} % differ; pre-tanh max {dpre.max():.2f} ulp; vs exact arithmetic: "
          f"HIP mean {hip_x.mean():.4f} max {hip_x.max():.2f}, oracle mean {ora_x.mean():.4f} max {ora_x.max():.2f} ulp; "
          f"max |dp| {np.abs(gp - rp).max():.2e} max |dv| {np.abs(gv - rv).max():.2e} max KL {policy_kl(rp, gp).max():.2e}")
    assert dl.max() <= MAX_LOGIT_ULPS and dpre.max() <= MAX_PRE_TANH_ULPS
    assert hip_x.mean() <= 1.25 * ora_x.mean() + 0.01 and hip_x.max() <= 1.5 * ora_x.max() + 1.0
    allowed_p, allowed_v, held = allowances(rp, rv, rl, rpre)
    assert np.all(np.abs(gp - rp) <= allowed_p) and np.all(np.abs(gv - rv) <= allowed_v) and policy_kl(rp, gp).max() <= 1e-3
    assert held.mean() > 0.9                                            # nine tenths of the entries are held to north_star's 1e-3 outright
