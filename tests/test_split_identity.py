"""The f16 hi/lo split of the shipped kernels (pesto_amd/csrc/pesto_mfma_common.h: split8): hi = f16(x) rounded to nearest even, lo = f16(x - hi).
Round 6 takes the residual in fp32 (v_fma_mix_f32) and converts a pair at once (v_cvt_pk_f16_f32) instead of one fused f16-destination
v_fma_mix{lo,hi}_f16 per element. Both round ONCE, to f16, provided x - hi is exact in fp32 - checked here over every binade the split can
see (the range guard keeps |x| <= 65504), plus the claim the kernels' accuracy rests on: hi + lo carries x to 2^-22 relative."""
import numpy as np


def _samples():
    rng = np.random.default_rng(7)
    e = rng.integers(-40, 15, size=2_000_000)                      # 2^-40 .. 2^15: f16 normals, subnormals and values that flush to zero
    m = rng.random(2_000_000, dtype=np.float32) + np.float32(1.0)
    x = (m * np.exp2(e).astype(np.float32)) * rng.choice(np.array([-1.0, 1.0], dtype=np.float32), size=e.size)
    edge = np.array([0.0, -0.0, 65504.0, -65504.0, 65503.9, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25, 1.0 + 2.0 ** -11, 1.0 + 2.0 ** -12, 1.0 + 3 * 2.0 ** -12,
                     6.1e-5, 5.96e-8, 1e-10], dtype=np.float32)
    return np.concatenate([x.astype(np.float32), edge])


def test_the_residual_of_the_split_is_exact_in_fp32():
    x = _samples()
    hi = x.astype(np.float16)
    r32 = x - hi.astype(np.float32)                                # what v_fma_mix_f32 computes (fma(hi, -1, x): one rounding to fp32)
    r64 = x.astype(np.float64) - hi.astype(np.float64)             # the real number
    assert np.array_equal(r32.astype(np.float64), r64)             # exact: rounding it to f16 afterwards = the fused form's single rounding
    lo = r32.astype(np.float16)
    assert np.array_equal(lo, r64.astype(np.float16))


def test_hi_plus_lo_carries_two_to_the_minus_22():
    x = _samples()
    x = x[np.abs(x) >= 2.0 ** -2]                                  # (the relative bound needs lo to be a normal f16: |lo| ~ 2^-12 |x| >= 2^-14)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - x.astype(np.float64))
    assert (err <= np.abs(x.astype(np.float64)) * 2.0 ** -22).all()
