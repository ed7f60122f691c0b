"""Formats either side of the path (scope row f-4) through the C ABI, pinned to the reference's own tests:
src/libdg_utils/b85.rs:171-219 (pi_e, _1234567, _f32, _encode, decode_encode) and
src/libdg_mcts/self_play.rs:542-590 (played_from_policy, played_pass, played_fixed)."""
import numpy as np


def test_b85_pi_e(api):
    assert api.b85_decode("NJ4Ny").tolist() == [3.140625, 2.71875]


def test_b85_1234567_padding(api):
    assert api.b85_decode("06YLd073vn07U>s07n1-").astype(np.float32).tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 0.0]


def test_b85_f32(api):
    assert api.b85_decode("000<4", dtype=np.float32).tolist() == [9.5]


def test_b85_encode(api):
    assert api.b85_encode([2.7578125, 3.67382812]) == "gh5$D"


def test_b85_decode_encode(api):
    examples = [
        [3.140625, 2.71875],
        [5.3203125, 9.9765625, 3.28320312, 8.15625, 7.109375, 1.81640625, 1.69921875, 4.4296875],
        [6.37890625, 9.6171875, 2.2890625, 9.4609375, 7.8984375, 9.3125, 4.10546875, 9.390625],
    ]
    for example in examples:
        want = np.asarray(example, dtype=np.float32).astype(np.float16)
        assert np.array_equal(api.b85_decode(api.b85_encode(example)), want)


def test_b85_rejects_characters_outside_the_alphabet(api):
    assert api.b85_decode('NJ4N"') is None
    assert api.b85_decode("").size == 0


def test_b85_agrees_with_python_base64_on_random_policies(api):
    """RFC 1924 is what Python's base64.b85 implements: an independent second opinion."""
    import base64
    rng = np.random.default_rng(5)
    p = rng.random(362).astype(np.float32)
    text = api.b85_encode(p)
    assert np.array_equal(np.frombuffer(base64.b85decode(text), dtype="<f2").astype(np.float16), p.astype(np.float16))
    assert np.array_equal(api.b85_decode(text), p.astype(np.float16))


def test_played_from_policy(api):
    """self_play.rs:543-552: from_forward has num_rollout = 1, so no TV / P; value 0.5 for black is 0.0000."""
    policy = np.zeros(362, dtype=np.float32); policy[0] = 1.0
    assert api.sgf_played(0, 0, value=0.5, num_rollout=1, softmax=policy) == ";B[aa]V[0.0000]"


def test_played_pass_and_fixed(api):
    """self_play.rs:575-589."""
    assert api.sgf_played(0, 361) == ";B[]"
    assert api.sgf_played(0, 0) == ";B[aa]"


def test_played_from_mcts_shape(api):
    """self_play.rs:554-573 checks the presence of TR / V / P; here the exact text for known inputs."""
    policy = np.full(362, 1.0 / 362, dtype=np.float32)
    text = api.sgf_played(1, 19 * 3 + 15, value=0.25, num_rollout=800, softmax=policy, prior_index=1, explain="a\nb")
    assert text.startswith(";W[pd]C[a\rb]TR[ba]TV[800]P[")
    assert text.endswith("]V[0.5000]")                                       # white to move at 25 %: black at 75 % -> 0.5
    body = text[text.index("P[") + 2:text.rindex("]V[")]
    assert len(body) == 181 * 5 and np.array_equal(api.b85_decode(body), policy.astype(np.float16))
