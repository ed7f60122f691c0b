"""The Winograd F(2x2, 3x3) emulation behind tools/winograd_numerics.py is a correct convolution: with
exactly representable data (small integers: every transform, product and sum is exact in fp16 / fp32) it gives the direct form's
result BIT FOR BIT, on a full 19 x 19 board including the odd last row / column of output tiles, with a skip input and without.
What the study then measures on real networks is rounding, not an indexing slip."""
import numpy as np


def test_winograd_emulation_is_exact_on_integers(oracle):
    rng = np.random.default_rng(3)
    b, cin, cout = 2, 8, 16
    x = rng.integers(-3, 4, size=(b, 19, 19, cin)).astype(np.float32)
    w = (rng.integers(-2, 3, size=(cout, 3, 3, cin)) * 4).astype(np.float32)       # multiples of 4: G g G^T stays integral
    bias = rng.integers(-5, 6, size=(cout,)).astype(np.float32)
    z = rng.integers(-8, 9, size=(b, 19, 19, cout)).astype(np.float32)
    for kw in (dict(), dict(alpha1=0.5, alpha2=0.5, z=z), dict(relu=False)):
        direct = oracle._conv3x3_numpy(x, w, bias, **kw)
        winograd = oracle._conv3x3_winograd(x, w, bias, **kw)
        assert np.array_equal(direct, winograd), kw.keys()


def test_winograd_whole_network_stays_close_on_a_small_net(oracle, synth):
    tensors, _ = oracle.load_weights(synth.to_json(synth.make_tensors(3, 64, seed=9)))
    net = oracle.OracleNetwork(tensors)
    features = synth.make_features(3, seed=10)
    f32 = lambda bits: np.asarray(bits).view(np.float16).astype(np.float32)
    dv, dp = oracle.forward_numpy(net, features, 3)
    wv, wp = oracle.forward_numpy(net, features, 3, tower_conv=oracle._conv3x3_winograd)
    assert np.abs(f32(dv) - f32(wv)).max() < 1e-3 and np.abs(f32(dp) - f32(wp)).max() < 1e-3
