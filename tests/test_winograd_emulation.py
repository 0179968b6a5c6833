"""CPU pins of the arithmetic behind conv_h3w_kernel (csrc/nbe_kernels_wino.h), emulated in NumPy (tools/wino_emulation.py):
the F(2,3) identity along z with the two-phase accumulation order and the negated U3, the 2^14 weight scale with an
unscaled lo part, and the 2^-11 of the lo(x) product on the weight operand -- held to the float32 tolerances of the GPU
layer tests (tests/test_gpu_layers.py) against an exact float64 convolution."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import wino_emulation as W          # noqa: E402


def _case(cin, cout, dims, seed):
    rng = np.random.default_rng(seed)
    x = W.f32(rng.standard_normal((cin,) + dims))
    w = W.f32(W.unit_rows(rng.standard_normal((cout, cin, 3, 3, 3))))
    return x, w


def test_winograd_z_form_is_exact_in_float64():
    """With exact arithmetic the form is the convolution (identity, stage order, signs): checked with the weights left in
    float64 and activations that are exactly representable in f16 (so that the hi / lo split is exact)."""
    rng = np.random.default_rng(1)
    x = rng.integers(-8, 9, size=(16, 6, 5, 7)).astype(np.float64)
    w = rng.integers(-4, 5, size=(8, 16, 3, 3, 3)).astype(np.float64) / 64.0
    assert np.array_equal(W.conv_winograd_z(x, w, S=2.0 ** 6), W.conv_exact(x, w))


def test_winograd_z_emulation_meets_the_float32_tolerances():
    x, w = _case(32, 16, (6, 7, 9), 2)
    ye = W.conv_exact(x, w)
    e_w, e_d = W.rel(W.conv_winograd_z(x, w), ye), W.rel(W.conv_f16x3_direct(x, w), ye)
    assert e_w <= 2e-6 and e_d <= 2e-6, (e_w, e_d)
    assert e_w <= 4 * e_d + 1e-7, (e_w, e_d)                    # the transform amplifies rounding by a small factor only


def test_winograd_z_emulation_follows_the_direct_form_over_the_f16_range():
    """The transform works on joined float32 values and re-splits them like a producer's epilogue: no fixed-point floor
    of its own.  (Below ~2^-14 the hi parts themselves go subnormal, for both forms: the range shift of include/nbe.h
    keeps calls away from there.)"""
    x, w = _case(16, 8, (4, 5, 6), 3)
    for s in (2.0 ** -10, 1.0, 2.0 ** 12):
        xs = W.f32(x * s)
        ye = W.conv_exact(xs, w)
        e_w, e_d = W.rel(W.conv_winograd_z(xs, w), ye), W.rel(W.conv_f16x3_direct(xs, w), ye)
        assert e_w <= 2e-6 and e_w <= 4 * e_d + 1e-7, (s, e_w, e_d)


def test_packed_f16_transform_matches_the_float32_one():
    """xf_step's transform (round 3): s = a hi +- b hi in f16, TwoSum's exact error and the lo parts into the new lo part.
    It represents V = a +- b to within 2^-21 (|a| + |b|) -- the float32 transform it replaces: 2^-23; both below the 2^-22 the
    operands themselves carry per part -- also where a and b cancel; the layer's error moves from 3.4e-7 to 3.5e-7."""
    x, w = _case(32, 16, (6, 7, 9), 4)
    ye = W.conv_exact(x, w)
    e16, e32 = W.rel(W.conv_winograd_z(x, w), ye), W.rel(W.conv_winograd_z(x, w, transform='f32'), ye)
    assert e16 <= 2e-6 and e16 <= 1.5 * e32 + 1e-7, (e16, e32)
    rng = np.random.default_rng(5)
    a = W.f32(rng.standard_normal(20000))
    b = W.f32(a * (1 + 1e-3 * rng.standard_normal(20000)))      # nearly equal: a - b cancels
    ah, al = W.split_scaled(a)
    bh, bl = W.split_scaled(b)
    for sb in (-1.0, 1.0):
        s, lo = W.transform_f16(ah, al, bh, bl, sb)
        exact = (ah + al / 2048.0) + sb * (bh + bl / 2048.0)
        err = np.abs(s + lo / 2048.0 - exact) / (np.abs(a) + np.abs(b))
        assert err.max() <= 2.0 ** -21, (sb, err.max())


def test_unscaled_lo_part_of_the_transformed_planes():
    """The default build keeps the lo part of V = a +- b unscaled (NBE_WINO_LOU: V lo = err + (a lo +- b lo) 2^-11), so that the
    product hi(w) . lo(V) needs no 2^-11 copy of the weights.  Below 2^-14 that part is a subnormal f16 number (2^-25 absolute);
    the engine's range shift (H3_RANGE_UP = 6: the input's maximum in [32, 64)) keeps activations where that does not show:
    same error as the scaled form from an activation RMS of 64 down to 1, within 10 % at 1/8 (2^-9 of the input's scale)."""
    x, w = _case(32, 16, (6, 7, 9), 5)
    for rms, slack in ((64.0, 1.02), (8.0, 1.02), (1.0, 1.02), (0.125, 1.10)):
        xs = W.f32(x * rms)
        ye = W.conv_exact(xs, w)
        e_s, e_u = W.rel(W.conv_winograd_z(xs, w), ye), W.rel(W.conv_winograd_z(xs, w, unscaled_lo=True), ye)
        assert e_u <= slack * e_s + 1e-9 and e_u <= 6e-7, (rms, e_s, e_u)
    # exact arithmetic: the form is still the convolution
    rng = np.random.default_rng(6)
    xi = rng.integers(-8, 9, size=(16, 6, 5, 7)).astype(np.float64) * 64.0
    wi = rng.integers(-4, 5, size=(8, 16, 3, 3, 3)).astype(np.float64) / 64.0
    assert np.array_equal(W.conv_winograd_z(xi, wi, S=2.0 ** 6, unscaled_lo=True), W.conv_exact(xi, wi))


def test_float16_model_winograd_form_rounds_its_operands_once_more():
    """conv_h3w_kernel<., ., F16> (the float16 model): on the SAME f16 operands a direct evaluation is left with the store's
    one rounding (2^-12 / sqrt 3 = 1.4e-4 .. 2.1e-4 relative L2), the Winograd-z form also rounds the transformed weights
    U_xi and the transformed planes V = a +- b to f16: about twice that -- the 7e-4 / 7e-3 of tests/test_gpu_layers.py
    (measured on the GPU: 4.6e-4 against 2.1e-4)."""
    x, w = _case(32, 16, (6, 7, 9), 7)
    x16, w16 = W.f16(x), W.f16(w)
    ye = W.conv_exact(x16, w16)
    e_d, e_w = W.rel(W.f16(ye), ye), W.rel(W.f16(W.conv_winograd_z_f16(x16, w16)), ye)
    assert 1.0e-4 <= e_d <= 2.5e-4 and e_d < e_w <= 6e-4, (e_d, e_w)
    # exact arithmetic: 32-channel stages, two-phase order, negated U3 -- still the convolution
    rng = np.random.default_rng(8)
    xi = rng.integers(-8, 9, size=(32, 6, 5, 7)).astype(np.float64)
    wi = rng.integers(-4, 5, size=(8, 32, 3, 3, 3)).astype(np.float64) / 64.0
    assert np.array_equal(W.conv_winograd_z_f16(xi, wi), W.conv_exact(xi, wi))
