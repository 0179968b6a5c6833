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

Two interchangeable evaluators of the stride-1 VALID cross-correlation:
  'numpy'  (default) the tap-wise GEMM below -- the definition every small test uses;
  'torch'  torch.nn.functional.conv3d on the CPU, in z-chunks that bound its im2col buffer -- an independent
           implementation of the same sum (pinned against the NumPy one in tests/test_oracle_pins.py), 6x faster
           in float64: used for the production-width fixtures (224^3 sub-boxes), the kink-aware parity checks that
           run the oracle at test time, and bench.py's cpu_baseline ("torch-CPU/oneDNN conv core").
Select with `with layers.backend('torch'): ...`.
"""

import contextlib

import numpy as np
from numpy.lib.stride_tricks import as_strided

NEG_SLOPE = 0.01

_BACKEND = ['numpy']


@contextlib.contextmanager
def backend(name):
    """Evaluate the stride-1 convolutions with 'numpy' (tap-wise GEMM) or 'torch' (CPU conv3d) inside the block."""
    if name not in ('numpy', 'torch'):
        raise ValueError("backend must be 'numpy' or 'torch'")
    _BACKEND.append(name)
    try:
        yield
    finally:
        _BACKEND.pop()


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


def _conv_valid_s1_torch(x, w, im2col_bytes=1.5e9):
    """The same sum through torch.nn.functional.conv3d (CPU).  float64 goes in z-chunks: torch's float64 path builds an
    im2col buffer of k^3 * Cin values per output voxel; float32 (oneDNN) takes the tensor whole."""
    import torch
    Cin, D, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    Do, Ho, Wo = D - k + 1, H - k + 1, W - k + 1
    xt = torch.from_numpy(np.ascontiguousarray(x))[None]
    wt = torch.from_numpy(np.ascontiguousarray(w))
    with torch.no_grad():
        if x.dtype == np.float32:
            return torch.nn.functional.conv3d(xt, wt)[0].numpy()
        out = np.empty((Cout, Do, Ho, Wo), dtype=x.dtype)
        ot = torch.from_numpy(out)
        nz = max(1, int(im2col_bytes / (k ** 3 * Cin * Ho * Wo * x.dtype.itemsize)))
        for z0 in range(0, Do, nz):
            z1 = min(Do, z0 + nz)
            ot[:, z0:z1] = torch.nn.functional.conv3d(xt[:, :, z0:z1 + k - 1], wt)[0]
    return out


def _iadd(a, b):
    """a += b (multi-threaded through torch when that backend is selected: the production-width tensors are GBs)."""
    if _BACKEND[-1] == 'torch':
        import torch
        torch.from_numpy(a).add_(torch.from_numpy(np.ascontiguousarray(b)) if b.shape == a.shape else torch.from_numpy(np.ascontiguousarray(b)).expand(a.shape))
    else:
        a += b
    return a


def _conv_valid(x, w, stride=1, zblock=16):
    """VALID cross-correlation, x (Cin,D,H,W), w (Cout,Cin,k,k,k) -> (Cout,Do,Ho,Wo)."""
    if stride == 1:
        if _BACKEND[-1] == 'torch' and w.shape[2] > 1:
            return _conv_valid_s1_torch(x, w)
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
    if _BACKEND[-1] == 'torch':
        # 1x1x1: one (Cout x Cin) @ (Cin x voxels) product in the tensor's own channels-first layout
        import torch
        xt = torch.from_numpy(x) if x.flags.writeable else torch.from_numpy(np.array(x))
        Cin = x.shape[0]
        with torch.no_grad():
            y = torch.from_numpy(np.ascontiguousarray(w[:, :, 0, 0, 0])) @ xt.reshape(Cin, -1)
        return y.numpy().reshape((w.shape[0],) + x.shape[1:])
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
    y = _CONV[kind](x, w)
    return _iadd(y, b[:, None, None, None])


def conv_layer_vel(kind, x, dx, w, dw, b):
    """y = conv(x,w)+b ; dy = conv(x,dw) [+ conv(dx,w)]  (style_layers_vel.py:129-141).
    The bias enters y only.  (Sums are formed in place: the production-width fixtures hold 5 GB tensors.)"""
    f = _CONV[kind]
    if kind == 'conv3' and _BACKEND[-1] == 'torch':
        # conv(x, w) and conv(x, dw) as one convolution with 2 Cout outputs: x is unfolded once
        co = w.shape[0]
        ydy = f(x, np.concatenate([w, dw], axis=0))
        y, dy = ydy[:co], ydy[co:]
    else:
        y = f(x, w)
        dy = f(x, dw)
    _iadd(y, b[:, None, None, None])
    if dx is not None:
        _iadd(dy, f(dx, w))
    return y, dy


# --------------------------------------------------------------------------
# activation
# --------------------------------------------------------------------------

def leaky_relu(x, slope=NEG_SLOPE):
    """layers.py:127-133 (jax.nn.leaky_relu: x >= 0 keeps x)."""
    return np.where(x >= 0, x, x.dtype.type(slope) * x)


def leaky_relu_vel(x, dx, slope=NEG_SLOPE, branch=None):
    """layers_vel.py:182-186: the tangent takes the slope branch at x == 0.

    branch: optional boolean array, True where the TANGENT is to take the identity branch instead of the
    reference's `x > 0` -- the kink-aware parity checks (tests/kink.py) evaluate the oracle with the branch
    decisions another evaluation took; the primal always follows the reference."""
    sl = x.dtype.type(slope)
    up = (x > 0) if branch is None else branch
    return np.where(x >= 0, x, sl * x), np.where(up, dx, sl * dx)


def leaky_relu_vel_(x, dx, slope=NEG_SLOPE, branch=None):
    """leaky_relu_vel written into its arguments (x and dx must be arrays nobody else reads)."""
    sl = x.dtype.type(slope)
    if _BACKEND[-1] == 'torch' and x.flags.c_contiguous and dx.flags.c_contiguous:
        import torch
        xt, dxt = torch.from_numpy(x), torch.from_numpy(dx)
        up = (xt > 0) if branch is None else torch.from_numpy(np.ascontiguousarray(branch))
        torch.where(up, dxt, dxt * float(sl), out=dxt)
        torch.where(xt >= 0, xt, xt * float(sl), out=xt)
        return x, dx
    up = (x > 0) if branch is None else branch
    np.multiply(dx, sl, out=dx, where=~up)
    np.multiply(x, sl, out=x, where=x < 0)
    return x, dx


def leaky_relu_(x, slope=NEG_SLOPE):
    np.multiply(x, x.dtype.type(slope), out=x, where=x < 0)
    return x
