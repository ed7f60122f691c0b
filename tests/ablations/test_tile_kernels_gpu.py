"""The NHWC tile kernels of round 1 (csrc/ablations/conv_tile.hip): an A/B baseline that only the DGNN_ABLATIONS build of the library
carries.  This directory is collected only when that library is the one loaded (tests/conftest.py):

    DGNN_ABLATIONS=1 python dream-go_amd/build.py
    DGNN_LIBRARY=dream-go_amd/lib/ablations/libdgnn_hip.so python -m pytest tests/ablations -m gpu

The default run neither collects nor skips them (rounds 2-5 carried 43 skips here)."""
import numpy as np
import pytest

from conftest import f16_to_f32

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", ["tile1", "tile2", "tile3"])
@pytest.mark.parametrize("fm", [4, 6, 8])
@pytest.mark.parametrize("batch,cin,cout,alpha", [
    (1, 256, 256, 0.5),      # one board: 361 positions, ragged in every tile height
    (3, 64, 256, None),      # a single 64-channel chunk per tap
    (2, 128, 512, 0.25),     # two output-channel blocks (grid.y = 2)
    (8, 256, 256, 0.5),      # several tiles
])
def test_tile_kernel_against_oracle_and_generic_kernel(api_full, oracle, monkeypatch, variant, fm, batch, cin, cout, alpha):
    """The tuned tile kernels (conv_tile.hip; lock-step and staggered pipeline, every position-tile height) against
    the oracle and against the generic kernel on the same inputs."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng([fm, batch, cin, cout])
    x = np.maximum(rng.normal(0, 1, (batch, 19, 19, cin)), 0).astype(np.float16)
    w = rng.normal(0, np.sqrt(2.0 / (9 * cin)), (cout, 3, 3, cin)).astype(np.float16)
    b = rng.normal(0, 0.1, (cout,)).astype(np.float16)
    z = np.maximum(rng.normal(0, 1, (batch, 19, 19, cout)), 0).astype(np.float16) if alpha is not None else None
    a1, a2 = (alpha, 1.0 - alpha) if alpha is not None else (1.0, 0.0)
    ref = oracle.conv3x3(x, batch, 19, cin, w, b, cout, alpha1=a1, alpha2=a2, z=z, relu=True)
    monkeypatch.setenv("DGNN_CONV_FM", str(fm))
    monkeypatch.setenv("DGNN_CONV", variant)
    tile = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=a1, alpha2=a2, z=z, relu=True)
    monkeypatch.setenv("DGNN_CONV", "v1")
    generic = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=a1, alpha2=a2, z=z, relu=True)
    scale = max(1.0, float(np.abs(f16_to_f32(ref)).max()))
    for name, got in (("tile", tile), ("generic", generic)):
        diff = np.abs(f16_to_f32(got) - f16_to_f32(ref))
        print(f"{name} {variant} fm={fm} B={batch} {cin}->{cout}: {int(np.count_nonzero(got != ref))}/{got.size} differ, max abs {diff.max():.3e}")
        assert diff.max() <= 4e-3 * scale
        assert np.count_nonzero(got != ref) <= 0.02 * got.size


def test_tile_kernel_exact_on_integers(api_full, monkeypatch):
    """All products and sums exactly representable: the tile kernel must equal the generic kernel bit for bit
    (catches any wrong row / slot / tap in the swizzled LDS image, which a tolerance could hide)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng(11)
    batch, cin, cout = 5, 128, 256
    x = rng.integers(0, 4, (batch, 19, 19, cin)).astype(np.float16)
    w = (rng.integers(-3, 4, (cout, 3, 3, cin)) / 16.0).astype(np.float16)
    b = (rng.integers(-8, 9, (cout,)) / 8.0).astype(np.float16)
    z = rng.integers(0, 4, (batch, 19, 19, cout)).astype(np.float16)
    monkeypatch.setenv("DGNN_CONV", "v1")
    want = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=0.5, alpha2=0.5, z=z, relu=False)
    for variant in ("tile1", "tile2", "tile3"):
        monkeypatch.setenv("DGNN_CONV", variant)
        for fm in (4, 6, 8):
            monkeypatch.setenv("DGNN_CONV_FM", str(fm))
            got = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=0.5, alpha2=0.5, z=z, relu=False)
            assert np.array_equal(got, want), (variant, fm)


@pytest.mark.parametrize("variant", ["tile1", "tile2", "tile3"])
def test_tile_kernel_exact_under_full_load(api_full, monkeypatch, variant):
    """BASELINE.json configs[2] layer shape at full size (256 boards, 256 -> 256 channels, every CU busy, two rounds
    of workgroups): integer-valued data make the generic kernel a bit-exact reference; three runs must be identical
    (a pipeline race would show up as run-to-run differences under load)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng(23)
    batch, cin, cout = 256, 256, 256
    x = rng.integers(0, 4, (batch, 19, 19, cin)).astype(np.float16)
    w = (rng.integers(-3, 4, (cout, 3, 3, cin)) / 16.0).astype(np.float16)
    b = (rng.integers(-8, 9, (cout,)) / 8.0).astype(np.float16)
    z = rng.integers(0, 4, (batch, 19, 19, cout)).astype(np.float16)
    monkeypatch.setenv("DGNN_CONV", "v1")
    want = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=0.5, alpha2=0.5, z=z, relu=False)
    monkeypatch.setenv("DGNN_CONV", variant)
    monkeypatch.delenv("DGNN_CONV_FM", raising=False)
    for run in range(3):
        got = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=0.5, alpha2=0.5, z=z, relu=False)
        assert np.array_equal(got, want), (variant, run, int(np.count_nonzero(got != want)))
