"""Oracle: layer arithmetic (test infrastructure only).

Follows the reference layer files:
  style_layers_vel.py:62-105   style -> s_mod, ds_mod; weight modulation,
                               demodulation, d/dDz of the demodulated weight,
                               first-layer rule (dx is None -> + w_n / Dz)
  style_layers_vel.py:107-141  y = conv(x, w_n) + b ; dy = conv(x, dw) [+ conv(dx, w_n)]
                               VALID cross-correlation, NCDHW / OIDHW
  style_layers_vel.py:234-269  upsample: lhs_dilation 2, padding 1, k = 2
  style_layers.py:59-92        displacement-only twin (no tangent)
  layers_vel.py:182-186        LeakyReLUVel
  nbody_emulator.py:131-148, :189-219  the same algebra as premodulation

All functions work on UNBATCHED arrays (C, D, H, W) in a caller-chosen dtype
(float64 = truth, float32 = envelope).  Convolutions are evaluated tap by tap
as (voxels x Cin) @ (Cin x Cout) matrix products.
"""

import numpy as np
from numpy.lib.stride_tricks import as_strided

NEG_SLOPE = 0.01


# --------------------------------------------------------------------------
# weight algebra
# --------------------------------------------------------------------------

def style_vector(Om, Dz, dtype=np.float64):
    """s = ((Om - 0.3) * 5, Dz - 1)   -- style_nbody_emulator_vel_core.py:126-128."""
    return np.array([(float(Om) - 0.3) * 5.0, float(Dz) - 1.0], dtype=dtype)


def modulate_weights(style_weight, style_bias, weight, s, eps=1e-8):
    """w_n  (style_layers.py:59-84 / nbody_emulator.py:131-148)."""
    dt = s.dtype
    sw = np.asarray(style_weight, dtype=dt)
    sb = np.asarray(style_bias, dtype=dt)
    w0 = np.asarray(weight, dtype=dt)
    s_mod = sw @ s + sb                                   # (Cin,)
    w = w0 * s_mod[None, :, None, None, None]
    norm = np.sqrt(np.sum(w * w, axis=(1, 2, 3, 4), keepdims=True) + dt.type(eps))
    return w / norm


def modulate_weights_vel(style_weight, style_bias, weight, s, first_layer, eps=1e-8):
    """(w_n, dw_tot)  (style_layers_vel.py:62-101 / nbody_emulator.py:189-219)."""
    dt = s.dtype
    sw = np.asarray(style_weight, dtype=dt)
    sb = np.asarray(style_bias, dtype=dt)
    w0 = np.asarray(weight, dtype=dt)
    s_mod = sw @ s + sb                                   # (Cin,)
    ds_mod = sw[:, 1]                                     # d s_mod / d s[1]
    w = w0 * s_mod[None, :, None, None, None]
    dw_style = w0 * ds_mod[None, :, None, None, None]
    norm = np.sqrt(np.sum(w * w, axis=(1, 2, 3, 4), keepdims=True) + dt.type(eps))
    dnorm = -np.sum(w * dw_style, axis=(1, 2, 3, 4), keepdims=True) / norm ** 3
    w_n = w / norm
    dw_n = dw_style / norm + w * dnorm
    if first_layer:
        Dz = s[1] + dt.type(1.0)
        dw_n = dw_n + w_n / Dz
    return w_n, dw_n


# --------------------------------------------------------------------------
# convolutions
# --------------------------------------------------------------------------

def _conv_valid_s1(x, w, zblock=8):
    """Stride-1 VALID cross-correlation.  In channels-last storage the k taps
    along W and the Cin channels of one (dz, dy) row are contiguous, so each
    (dz, dy) pair is ONE (voxels x k*Cin) @ (k*Cin x Cout) product."""
    Cin, D, H, W = x.shape
    Cout, Cin2, k = w.shape[0], w.shape[1], w.shape[2]
    assert Cin == Cin2, (x.shape, w.shape)
    Do, Ho, Wo = D - k + 1, H - k + 1, W - k + 1
    xl = np.ascontiguousarray(np.moveaxis(x, 0, -1))      # (D,H,W,Cin)
    sz, sy, sx, sc = xl.strides
    v = as_strided(xl, shape=(D, H, Wo, k * Cin), strides=(sz, sy, sx, sc), writeable=False)
    wt = np.ascontiguousarray(np.transpose(w, (2, 3, 4, 1, 0))).reshape(k, k, k * Cin, Cout)
    out = np.empty((Do, Ho, Wo, Cout), dtype=x.dtype)
    for z0 in range(0, Do, zblock):
        z1 = min(Do, z0 + zblock)
        acc = np.zeros(((z1 - z0) * Ho * Wo, Cout), dtype=x.dtype)
        for a in range(k):
            for b in range(k):
                acc += v[z0 + a:z1 + a, b:b + Ho].reshape(-1, k * Cin) @ wt[a, b]
        out[z0:z1] = acc.reshape(z1 - z0, Ho, Wo, Cout)
    return np.ascontiguousarray(np.moveaxis(out, -1, 0))


def _conv_valid(x, w, stride=1, zblock=16):
    """VALID cross-correlation, x (Cin,D,H,W), w (Cout,Cin,k,k,k) -> (Cout,Do,Ho,Wo)."""
    if stride == 1:
        return _conv_valid_s1(x, w)
    Cin, D, H, W = x.shape
    Cout, Cin2, k = w.shape[0], w.shape[1], w.shape[2]
    assert Cin == Cin2, (x.shape, w.shape)
    Do, Ho, Wo = (D - k) // stride + 1, (H - k) // stride + 1, (W - k) // stride + 1
    xl = np.ascontiguousarray(np.moveaxis(x, 0, -1))      # (D,H,W,Cin)
    out = np.empty((Do, Ho, Wo, Cout), dtype=x.dtype)
    wt = [[[np.ascontiguousarray(w[:, :, a, b, c].T) for c in range(k)] for b in range(k)] for a in range(k)]
    for z0 in range(0, Do, zblock):
        z1 = min(Do, z0 + zblock)
        nz = z1 - z0
        acc = np.zeros((nz * Ho * Wo, Cout), dtype=x.dtype)
        for a in range(k):
            zs = slice(z0 * stride + a, z0 * stride + a + (nz - 1) * stride + 1, stride)
            for b in range(k):
                ys = slice(b, b + (Ho - 1) * stride + 1, stride)
                for c in range(k):
                    xs = slice(c, c + (Wo - 1) * stride + 1, stride)
                    acc += xl[zs, ys, xs, :].reshape(-1, Cin) @ wt[a][b][c]
        out[z0:z1] = acc.reshape(nz, Ho, Wo, Cout)
    return np.ascontiguousarray(np.moveaxis(out, -1, 0))


def conv3(x, w):
    return _conv_valid(x, w, 1)


def conv1(x, w):
    return _conv_valid(x, w, 1)


def down2(x, w):
    return _conv_valid(x, w, 2)


def up2_literal(x, w):
    """Exactly what the reference asks XLA for (style_layers_vel.py:236-244):
    zero-stuff the input by 2 (lhs_dilation), pad 1 each side, VALID k=2 conv."""
    Cin, D, H, W = x.shape
    xd = np.zeros((Cin, 2 * D + 1, 2 * H + 1, 2 * W + 1), dtype=x.dtype)
    xd[:, 1:2 * D:2, 1:2 * H:2, 1:2 * W:2] = x
    return _conv_valid(xd, w, 1)


def up2(x, w):
    """Parity form of up2_literal: y[:, 2i+p] = W[:, :, 1-p] . x[:, i]
    (checked against up2_literal in tests/test_oracle_pins.py)."""
    Cin, D, H, W = x.shape
    Cout = w.shape[0]
    xl = np.moveaxis(x, 0, -1).reshape(-1, Cin)
    out = np.empty((Cout, 2 * D, 2 * H, 2 * W), dtype=x.dtype)
    for pz in range(2):
        for py in range(2):
            for px in range(2):
                y = xl @ w[:, :, 1 - pz, 1 - py, 1 - px].T           # (vox, Cout)
                out[:, pz::2, py::2, px::2] = np.moveaxis(y.reshape(D, H, W, Cout), -1, 0)
    return out


_CONV = {'conv3': conv3, 'skip': conv1, 'down': down2, 'up': up2}


def conv_layer(kind, x, w, b):
    """Displacement-only layer: y = conv(x, w) + b   (style_layers.py:86-99)."""
    return _CONV[kind](x, w) + b[:, None, None, None]


def conv_layer_vel(kind, x, dx, w, dw, b):
    """y = conv(x,w)+b ; dy = conv(x,dw) [+ conv(dx,w)]  (style_layers_vel.py:129-141).
    The bias enters y only."""
    f = _CONV[kind]
    y = f(x, w) + b[:, None, None, None]
    dy = f(x, dw)
    if dx is not None:
        dy = dy + f(dx, w)
    return y, dy


# --------------------------------------------------------------------------
# activation
# --------------------------------------------------------------------------

def leaky_relu(x, slope=NEG_SLOPE):
    """layers.py:127-133 (jax.nn.leaky_relu: x >= 0 keeps x)."""
    return np.where(x >= 0, x, x.dtype.type(slope) * x)


def leaky_relu_vel(x, dx, slope=NEG_SLOPE):
    """layers_vel.py:182-186: the tangent takes the slope branch at x == 0."""
    sl = x.dtype.type(slope)
    return np.where(x >= 0, x, sl * x), np.where(x > 0, dx, sl * dx)
