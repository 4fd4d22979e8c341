"""CPU tier: the product's kernel bodies + host orchestration, executed by the
work-group emulator (tests/emul), against the reference's golden vectors and
the oracle.  No GPU needed; sizes are the golden fixtures' (<= 448 x 256)."""
import pytest
import torch

from tests import parity_checks as pc
from tests.golden import inputs as gi


@pytest.fixture(scope="module")
def engine():
    from tests.emul.loader import emul_engine
    return emul_engine()


@pytest.mark.parametrize("case", gi.FFT_CASES, ids=lambda c: c["id"])
def test_fft(engine, golden, case):
    pc.check_fft(engine, golden, case)


@pytest.mark.parametrize("case", gi.INTERP_CASES, ids=lambda c: c["id"])
def test_interpolate(engine, golden, case):
    pc.check_interp(engine, golden, case)


@pytest.mark.parametrize("case", gi.PAIR_CASES, ids=lambda c: c["id"])
def test_pair_slerp(engine, golden, case):
    pc.check_pair(engine, golden, case)


@pytest.mark.parametrize("case", gi.ARITH_CASES, ids=lambda c: c["id"])
def test_pair_arith(engine, golden, case):
    pc.check_arith(engine, golden, case)


@pytest.mark.parametrize("case", gi.LAYER_CASES, ids=lambda c: c["id"])
def test_layer(engine, golden, case):
    pc.check_layer(engine, golden, case)


def test_unsupported_length_is_loud(engine):
    x = torch.randn(4, 17 * 19)          # 323 = 17*19: no radix for it
    with pytest.raises(NotImplementedError):
        engine.fft_transform(x)
    assert engine.lib.length_supported(14336) and engine.lib.length_supported(28672)
    assert not engine.lib.length_supported(17) and not engine.lib.length_supported(65536)


def test_inf_raises_like_reference(engine):
    base = torch.zeros(8, 8, dtype=torch.bfloat16)
    ft = base.clone()
    ft[0, 0] = float("inf")
    with pytest.raises(ValueError, match="Inf in merged tensor for model.layers.3.w"):
        engine.merge_layer([ft], [base], [1.0], base, layer_name="model.layers.3.w")
    a = torch.randn(8, 8)
    b = torch.randn(8, 8)
    a[1, 1] = 3e38
    b[1, 1] = -3e38
    # the overflow in the transform gives Inf/NaN; the reference zeroes NaN and raises on Inf
    try:
        engine.merge_tensors_fft2_slerp(a, b, 0.5)
    except ValueError as e:
        assert "Inf in ifft output" in str(e)


def test_nan_is_zeroed(engine):
    base = torch.randn(16, 16).to(torch.bfloat16)
    ft = (base.float() + 0.01 * torch.randn(16, 16)).to(torch.bfloat16)
    ft2 = ft.clone()
    ft2[2, 3] = float("nan")
    out, rep = engine.merge_layer([ft2], [base], [1.0], base)
    assert rep.nan_final == 1 and out[2, 3] == 0 and not torch.isnan(out.float()).any()
