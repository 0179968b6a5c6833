"""The four emulator cores with the reference's constructor and `apply` signatures.

  StyleNBodyEmulatorVelCore.apply(params, x, Om, Dz, vel_fac) -> (disp, vel)   style_nbody_emulator_vel_core.py:105-195
  StyleNBodyEmulatorCore.apply(params, x, Om, Dz)             -> disp          style_nbody_emulator_core.py:101-175
  NBodyEmulatorVelCore.apply(params, x, Dz, vel_fac)          -> (disp, vel)   nbody_emulator_vel_core.py:102-183
  NBodyEmulatorCore.apply(params, x, Dz)                      -> disp          nbody_emulator_core.py:98-166

They hold no arithmetic: `apply` hands the parameter tree and the input to the HIP
engine through the C ABI (engine.py -> libnbe.so).  `x` is (B, C, D, H, W), a NumPy
array (host, copied over PCIe) or a CUDA torch tensor (stays resident); outputs come
back as the same kind, in x's dtype.  The arithmetic follows x's dtype as in the reference
(style_layers_vel.py:103-105): float32 input -> the float32-equivalent engine (NBE_PRECISION,
default "f16x3"), float16 input -> the float16 engine ("f16": f16 operands, float32
accumulation).  bfloat16 input is rounded on the way in and out of the float32 engine.
"""

from dataclasses import dataclass

import numpy as np

from . import engine as _engine

_ENGINES = {}


def precision_for(dtype):
    """Engine arithmetic for a model dtype: float16 -> "f16", everything else -> NBE_PRECISION / "f16x3"."""
    import os
    is_f16 = False
    if dtype is not None:
        if _engine.torch is not None and isinstance(dtype, _engine.torch.dtype):
            is_f16 = dtype == _engine.torch.float16
        else:
            is_f16 = np.dtype(dtype) == np.float16
    return "f16" if is_f16 else os.environ.get("NBE_PRECISION", "f16x3")


def get_engine(model, device=None, precision=None):
    """One engine (context + weights + workspace) per (device, architecture, variant, arithmetic)."""
    if device is None:
        device = 0
    if precision is None:
        precision = precision_for(None)
    key = (int(device), model.in_chan, model.out_chan, model.mid_chan, float(model.eps), model._compute_vel, precision)
    if getattr(model, 'style_size', 2) != 2:
        raise ValueError("style_size must be 2: the style vector is ((Om-0.3)*5, Dz-1)")
    eng = _ENGINES.get(key)
    if eng is None:
        eng = _engine.Engine(device=device, in_chan=model.in_chan, out_chan=model.out_chan,
                             mid_chan=model.mid_chan, eps=model.eps, compute_vel=model._compute_vel,
                             precision=precision)
        _ENGINES[key] = eng
    return eng


def release_engines():
    for e in _ENGINES.values():
        e.close()
    _ENGINES.clear()


def _layer_table(in_chan, out_chan, mid_chan):
    """(block, layer, cout, cin, k) for the 33 convolution layers
    (style_nbody_emulator_vel_core.py:45-103, channel rule style_blocks_vel.py:126-134)."""
    m1, m2 = mid_chan, 2 * mid_chan
    blocks = [('conv_l00', in_chan, m1, 'CACA'), ('conv_l01', m1, m1, 'CACA'), ('down_l0', m1, m1, 'DA'),
              ('conv_l1', m1, m1, 'CACA'), ('down_l1', m1, m1, 'DA'), ('conv_l2', m1, m1, 'CACA'),
              ('down_l2', m1, m1, 'DA'), ('conv_c', m1, m1, 'CACA'), ('up_r2', m1, m1, 'UA'),
              ('conv_r2', m2, m1, 'CACA'), ('up_r1', m1, m1, 'UA'), ('conv_r1', m2, m1, 'CACA'),
              ('up_r0', m1, m1, 'UA'), ('conv_r00', m2, m1, 'CACA'), ('conv_r01', m1, out_chan, 'CAC')]
    out = []
    for name, ci, co, seq in blocks:
        if 'D' in seq or 'U' in seq:
            out.append((name, 'conv_0', co, ci, 2))
            continue
        mid = max(ci, co)
        out.append((name, 'skip', co, ci, 1))
        n = seq.count('C')
        for i in range(n):
            out.append((name, 'conv_%d' % i, co if i == n - 1 else mid, ci if i == 0 else mid, 3))
    return out


def _trunc_normal(rng, shape, std):
    # variance_scaling(..., 'truncated_normal'): N(0,1) truncated to [-2, 2], rescaled to unit variance
    v = rng.standard_normal(shape)
    bad = np.abs(v) > 2
    while bad.any():
        v[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(v) > 2
    return (v * (std / 0.87962566103423978)).astype(np.float32)


def _seed_of(key):
    if isinstance(key, (int, np.integer)):
        return int(key)
    try:
        return int(np.asarray(key).ravel()[-1])
    except Exception:
        return abs(hash(key)) % (2 ** 31)


class _Core:
    # concrete cores are frozen dataclasses declaring the reference's fields in the reference's order
    _premodulate = False
    _compute_vel = True

    # ---- parameters ---------------------------------------------------------------------------
    def init(self, key, *example_args):
        """Fresh parameter tree with the reference's initialisers (lecun_normal weights, ones
        style_bias, zero bias: style_layers_vel.py:55-75; premodulated layers also draw `dweight`,
        layers_vel.py:44-52).  `key` is an int seed (a jax PRNGKey-like array also works);
        the random stream is NumPy's, so values differ from JAX's for the same key."""
        rng = np.random.default_rng(_seed_of(key))
        tree = {}
        for blk, lay, co, ci, k in _layer_table(self.in_chan, self.out_chan, self.mid_chan):
            std = 1.0 / np.sqrt(ci * k ** 3)
            leaf = {'weight': _trunc_normal(rng, (co, ci, k, k, k), std), 'bias': np.zeros(co, np.float32)}
            if self._premodulate:
                if self._compute_vel:
                    leaf['dweight'] = _trunc_normal(rng, (co, ci, k, k, k), std)
            else:
                leaf['style_weight'] = _trunc_normal(rng, (ci, getattr(self, 'style_size', 2)), 1.0 / np.sqrt(ci))
                leaf['style_bias'] = np.ones(ci, np.float32)
            tree.setdefault(blk, {})[lay] = leaf
        return {'params': tree}

    # ---- forward ------------------------------------------------------------------------------
    def _run(self, params, x, Om, Dz, vel_fac, device=None):
        is_t = _engine._is_torch(x)
        if x.ndim != 5:
            raise ValueError("x must be (B, C, D, H, W); got shape %s" % (tuple(x.shape),))
        if is_t and not x.is_cuda:
            x = x.numpy()
            is_t = False
        if is_t and device is None:
            device = x.device.index or 0
        prec = precision_for(x.dtype)
        try:
            return self._run_on(get_engine(self, device, prec), params, x, Om, Dz, vel_fac, is_t)
        except _engine.NBERangeError as e:
            # an activation left the f16 range (include/nbe.h, "Range"): the strict float32 engine has float32's range
            import warnings
            warnings.warn("%s -- recomputing this call with the strict float32 engine" % e, RuntimeWarning)
            return self._run_on(get_engine(self, device, "f32"), params, x, Om, Dz, vel_fac, is_t)

    def _run_on(self, eng, params, x, Om, Dz, vel_fac, is_t):
        eng.ensure_params(params, self._premodulate)
        B = x.shape[0]
        bc = lambda v: None if v is None else np.broadcast_to(np.atleast_1d(np.asarray(v, dtype=np.float32)).ravel(), (B,))
        Om_, Dz_, vf_ = bc(Om), bc(Dz), bc(vel_fac)
        in_dtype = x.dtype
        ds, vs = [], []
        for i in range(B):
            if not self._premodulate:
                eng.set_cosmology(Om_[i], Dz_[i])
            xi = x[i]
            xi = xi.float() if is_t else np.asarray(xi).astype(np.float32, copy=False)
            r = eng.forward(xi, Dz_[i], 0.0 if vf_ is None else vf_[i])
            if self._compute_vel:
                ds.append(r[0]); vs.append(r[1])
            else:
                ds.append(r)
        if is_t:
            import torch
            stack = lambda l: torch.stack(l).to(in_dtype)
        else:
            stack = lambda l: np.stack(l).astype(in_dtype, copy=False)
        if self._compute_vel:
            return stack(ds), stack(vs)
        return stack(ds)


@dataclass(frozen=True)
class StyleNBodyEmulatorVelCore(_Core):
    style_size: int = 2
    in_chan: int = 3
    out_chan: int = 3
    mid_chan: int = 64
    eps: float = 1e-8
    _premodulate = False
    _compute_vel = True

    def apply(self, params, x, Om, Dz, vel_fac):
        return self._run(params, x, Om, Dz, vel_fac)

    __call__ = apply


@dataclass(frozen=True)
class StyleNBodyEmulatorCore(_Core):
    style_size: int = 2
    in_chan: int = 3
    out_chan: int = 3
    mid_chan: int = 64
    eps: float = 1e-8
    _premodulate = False
    _compute_vel = False

    def apply(self, params, x, Om, Dz):
        return self._run(params, x, Om, Dz, None)

    __call__ = apply


@dataclass(frozen=True)
class NBodyEmulatorVelCore(_Core):
    in_chan: int = 3
    out_chan: int = 3
    mid_chan: int = 64
    eps: float = 1e-8
    _premodulate = True
    _compute_vel = True

    def apply(self, params, x, Dz, vel_fac):
        return self._run(params, x, None, Dz, vel_fac)

    __call__ = apply


@dataclass(frozen=True)
class NBodyEmulatorCore(_Core):
    in_chan: int = 3
    out_chan: int = 3
    mid_chan: int = 64
    eps: float = 1e-8
    _premodulate = True
    _compute_vel = False

    def apply(self, params, x, Dz):
        return self._run(params, x, None, Dz, None)

    __call__ = apply
