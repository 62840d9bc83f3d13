"""The arithmetic claim behind the fp16 hi/lo-split MFMA paths (csrc/siren16_kernels.h, csrc/full_conv16_kernels.h),
checked in numpy on the shipped student weights: with v = hi + lo (fp16 halves) and fp32 accumulation,
W_hi x_hi + W_hi x_lo + W_lo x_hi reproduces the fp64 product to fp32-class accuracy, a single fp16 pass does not,
and one SIREN layer of sin(30 (W x + b)) amplifies the difference exactly as SURVEY.md §0.4 states."""
import numpy as np
import pytest

f16, f32, f64 = np.float16, np.float32, np.float64


def split(a):
    a = a.astype(f32)
    hi = a.astype(f16)
    lo = (a - hi.astype(f32)).astype(f16)
    return hi.astype(f32), lo.astype(f32)


def pow2_scale(w):
    e = int(np.floor(np.log2(16384.0 / np.abs(w).max())))
    return f32(2.0 ** e)


@pytest.fixture(scope="module")
def layer(golden_weights):
    w = golden_weights["body.siren_layers.0.1.linear.weight"].reshape(360, 360).astype(f32)
    b = golden_weights["body.siren_layers.0.1.linear.bias"].astype(f32)
    rng = np.random.default_rng(3)
    x = np.sin(rng.uniform(-40, 40, (360, 512))).astype(f32)          # activations are sines
    return w, b, x


def test_three_term_split_is_fp32_class(layer):
    w, b, x = layer
    w30 = (w * f32(30.0)).astype(f32)
    s = pow2_scale(w30)
    wh, wl = split(w30 * s)
    xh, xl = split(x)
    acc = (wh @ xh + wh @ xl + wl @ xh).astype(f32)                  # fp16 products are exact in fp32; numpy sums in fp32
    u_split = acc / s + (f32(30.0) * b)[:, None]
    u_ref = (w30.astype(f64) @ x.astype(f64)) + (30.0 * b.astype(f64))[:, None]
    u_f32 = (w30 @ x).astype(f32) + (f32(30.0) * b)[:, None]
    scale = np.abs(u_ref).max()
    err_split = np.abs(u_split - u_ref).max() / scale
    err_f32 = np.abs(u_f32 - u_ref).max() / scale
    assert err_split < 2.0 ** -20, err_split                         # ~22 significant bits survive
    assert err_split < 8 * max(err_f32, 2.0 ** -24)                  # the same class as a plain fp32 GEMM
    one_pass = (wh @ xh) / s + (f32(30.0) * b)[:, None]              # single fp16 pass: 11 bits
    err_one = np.abs(one_pass - u_ref).max() / scale
    assert err_one > 50 * err_split


def test_sine_layer_amplification(layer):
    w, b, x = layer
    w30 = (w * f32(30.0)).astype(f32)
    s = pow2_scale(w30)
    wh, wl = split(w30 * s)
    xh, xl = split(x)
    ref = np.sin((w30.astype(f64) @ x.astype(f64)) + (30.0 * b.astype(f64))[:, None])
    three = np.sin(((wh @ xh + wh @ xl + wl @ xh) / s + (f32(30.0) * b)[:, None]).astype(f64))
    one = np.sin(((wh @ xh) / s + (f32(30.0) * b)[:, None]).astype(f64))
    plain = np.sin(((w30 @ x).astype(f32) + (f32(30.0) * b)[:, None]).astype(f64))     # an ordinary fp32 GEMM
    e3, e1, e32 = np.abs(three - ref).max(), np.abs(one - ref).max(), np.abs(plain - ref).max()
    assert e3 < 5e-5 and e3 < 3 * e32                # the class of a plain fp32 layer (|u| reaches ~100: 1 ulp is 8e-6)
    assert e1 > 5e-3                                 # fp16 weights/activations in one pass: visible after a single layer


def test_unscaled_lo_halves_of_small_weights_need_the_scale(layer):
    """without the per-layer power-of-two scale the lo halves of small weights fall into the fp16 subnormal range"""
    w, _, _ = layer
    small = w[np.abs(w) < 1e-3]
    assert small.size > 100
    hi = small.astype(f16).astype(f32)
    lo_unscaled = (small - hi).astype(f16).astype(f32)
    s = pow2_scale(w * f32(30.0)) * f32(30.0)
    his = (small * s).astype(f16).astype(f32)
    lo_scaled = ((small * s) - his).astype(f16).astype(f32)
    err_unscaled = np.abs((hi + lo_unscaled) - small).max() / np.abs(small).max()
    err_scaled = np.abs((his + lo_scaled) / s - small).max() / np.abs(small).max()
    assert err_scaled < 2.0 ** -18 and err_unscaled > 4 * err_scaled
