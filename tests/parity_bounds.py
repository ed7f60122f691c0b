"""The parity contract on peaked outputs, in one place (tests/test_round3_gpu.py, tests/test_round4_gpu.py, tests/test_sharp_fixture*.py).

Both implementations store every tensor as fp16 and differ by single-ulp flips from the order of their fp32 sums.  In front of the
soft-max sit fp16 LOGITS, in front of tanh an fp16 pre-activation; the contract is stated THERE: at most MAX_LOGIT_ULPS /
MAX_PRE_TANH_ULPS fp16 ulps at the scale of the sum (the board's largest |logit|; the batch's largest |pre-tanh|).  What north_star's
1e-3 in probability / value space becomes follows per entry:

    dp_i = p_i (d_i - sum_j p_j d_j)  <=  2 p_i (1 - p_i) D + ulp(p_i)        D = the logit bound on that board
    dv   = (1 - v^2) d                <=  (1 - v^2) Dv + ulp(v)

and wherever that is below 1e-3 the entry is held to 1e-3 (98.8 % of the probabilities of the peaked 20x256 workload).
"""
import numpy as np

NORTH_STAR_TOL = 1e-3
MAX_LOGIT_ULPS = 1.0           # measured on the well-conditioned workloads (gate 0.1): exactly 1.00 at 20x256 B=256 and at 9x128 B=32
MAX_PRE_TANH_ULPS = 1.0


def ulp16(x):
    """Spacing of fp16 at |x| (fp32 array): 2^(floor(log2 |x|) - 10), 2^-24 in the subnormal range."""
    a = np.maximum(np.abs(np.asarray(x, dtype=np.float64)), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(a)) - 10)


def logit_unit(ref_logits):
    """One fp16 ulp at the scale of each board's largest |logit| -> [B, 1]."""
    return ulp16(np.abs(ref_logits).max(axis=1, keepdims=True))


def pre_tanh_unit(ref_pre):
    return ulp16(np.abs(ref_pre).max())


def allowances(ref_policy, ref_value, ref_logits, ref_pre, logit_ulps=MAX_LOGIT_ULPS, pre_ulps=MAX_PRE_TANH_ULPS):
    """Per-entry tolerances derived from the ulp bounds -> (allowed |dp| [B, 362], allowed |dv| [B], entries held to 1e-3 [B, 362] bool)."""
    derived = 2.0 * ref_policy * (1.0 - ref_policy) * logit_ulps * logit_unit(ref_logits) + ulp16(ref_policy)
    allowed_p = np.maximum(NORTH_STAR_TOL, derived)
    allowed_v = np.maximum(NORTH_STAR_TOL, (1.0 - ref_value * ref_value) * pre_ulps * pre_tanh_unit(ref_pre) + ulp16(ref_value))
    return allowed_p, allowed_v, derived <= NORTH_STAR_TOL


def policy_kl(p_ref, p):
    """KL(p_ref || p) per board, on fp16 probabilities promoted to fp64, with a floor against log(0)."""
    p_ref = np.asarray(p_ref, dtype=np.float64).reshape(-1, 362)
    p = np.asarray(p, dtype=np.float64).reshape(-1, 362)
    eps = 1e-12
    return np.sum(p_ref * (np.log(p_ref + eps) - np.log(p + eps)), axis=1)
