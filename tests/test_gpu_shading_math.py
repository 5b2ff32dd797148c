"""The library's own pow / sin / cos (trace3_device.h: rtPow, rtSinCos — the functions whose precision GLSL leaves to the
implementation, random.glsl:94-95,133-141, trace3.cs:335) against double precision, through the C-ABI (rt_shading_math).
Parity note: the reference's pinned platform evaluates these with polynomials of its own ("within about 1 ulp of libm,
not bit-identical"), so there is no bit pattern to match: the bar is an error bound, stated here."""
import numpy as np
import pytest

from raytracer_amd import hiplib

pytestmark = pytest.mark.gpu

ULP1 = 2.0 ** -23  # one unit in the last place of values in [1, 2)


@pytest.fixture(scope="module")
def ctx():
    c = hiplib.Context(0)
    yield c
    c.close()


def test_sin_cos_of_every_random_number(ctx):
    """All 2^23 values a random number can take (random.glsl:68-75: k / 2^23): the argument is the ROUNDED product
    2 pi * rp, as the shader forms it; error of sin and cos <= 0.8 ulp(1)."""
    worst = 0.0
    for lo in range(0, 1 << 23, 1 << 21):
        k = np.arange(lo, lo + (1 << 21), dtype=np.float64)
        rp = (k / 2.0 ** 23).astype(np.float32)
        p = (np.float32(6.28318530718) * rp).astype(np.float64)
        s, c, _ = ctx.shading_math(rp, np.ones_like(rp), np.ones_like(rp))
        worst = max(worst, float(np.abs(s - np.sin(p)).max()), float(np.abs(c - np.cos(p)).max()))
    assert worst <= 0.8 * ULP1, worst / ULP1


def test_pow_against_double_precision(ctx):
    rng = np.random.default_rng(7)
    n = 1 << 20
    # bases as the call sites have them: random numbers and clamped cosines in [0, 1]; exponents: Ns (up to a few
    # hundred), a / (a + 1) and 1 / (a + 1)
    x = rng.random(n).astype(np.float32)
    y = np.where(rng.random(n) < 0.5, rng.random(n), rng.random(n) * 400.0).astype(np.float32)
    _, _, got = ctx.shading_math(np.zeros(n, np.float32), x, y)
    want = np.power(x.astype(np.float64), y.astype(np.float64))
    ok = want > 1e-30   # (results that underflow fp32's normal range come out as 0 or a denormal either way)
    # v_exp_f32(y * v_log_f32(x)): the product carries the logarithm's last-place error times |y log2 x|
    bound = ULP1 * (2.0 + np.abs(y.astype(np.float64) * np.log2(np.maximum(x.astype(np.float64), 1e-300))))
    rel = np.abs(got[ok] - want[ok]) / want[ok]
    assert (rel <= bound[ok]).all(), float((rel / bound[ok]).max())
    assert (got[~ok] <= 1e-29).all()


def test_pow_edge_cases(ctx):
    x = np.array([0.0, 0.0, 1.0, 0.5, 0.25, 1.0, 0.0], np.float32)
    y = np.array([0.0, 2.5, 0.0, 0.0, 0.5, 123.0, 1.0], np.float32)
    _, _, got = ctx.shading_math(np.zeros_like(x), x, y)
    assert got.tolist() == [1.0, 0.0, 1.0, 1.0, 0.5, 1.0, 0.0]   # pow(x, 0) = 1 also for x = 0, as powf


def test_bad_arguments(ctx):
    with pytest.raises(hiplib.RtError):
        hiplib.Context._check(ctx, hiplib.lib().rt_shading_math(ctx._h, -1, None, None), "rt_shading_math")
