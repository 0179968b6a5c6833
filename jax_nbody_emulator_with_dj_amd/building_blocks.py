"""Tier-3 building blocks (layers and blocks for custom architectures) on the HIP engine.

The reference exposes its layers and blocks through eight submodules (`layers`, `layers_vel`, `style_layers`,
`style_layers_vel`, `blocks`, `blocks_vel`, `style_blocks`, `style_blocks_vel`; reference `__init__.py:52-63`).  They
are flax modules there; here each is a small class with the same constructor fields, the same parameter leaves and the
same call signatures -- `init(key, *example_args) -> {'params': ...}` and `apply(params, *args)` -- whose arithmetic runs
one layer at a time through the production kernels (`nbe_test_layer` / `nbe_test_modulate` of the C ABI).  They are for
experimenting with architectures, not for speed: every call moves its tensors over PCIe.  The emulator itself never
goes through this module.

  layer classes        params leaves                               call
  StyleConvBase3DVel   weight, bias, style_weight, style_bias      (x, s, dx=None) -> (y, dy)     style_layers_vel.py:20-147
  StyleConvBase3D      ''                                          (x, s) -> y                    style_layers.py:19-105
  ConvBase3DVel        weight, dweight, bias                       (x, dx=None) -> (y, dy)        layers_vel.py:20-96
  ConvBase3D           weight, bias                                (x) -> y                       layers.py:19-69
  ...Transpose...      the k = 2, lhs_dilation 2 up-sampling twins                                style_layers_vel.py:150-275 ...
  LeakyReLU[Vel]       --                                          (x[, dx])                      layers_vel.py:178-186
  blocks               {conv_0, conv_1, ..., skip}                 as their layers                style_blocks_vel.py:31-166 ...

x is (B, C, D, H, W) or (C, D, H, W); s is (B, style_size) or (style_size,), style_size = 2.
"""

from dataclasses import dataclass
from functools import partial

import numpy as np

from . import engine as _engine
from .models import _seed_of, _trunc_normal

_ENG = {}


def _eng():
    """One engine for single-layer calls, in the package's default arithmetic (NBE_PRECISION, default f16x3)."""
    import os
    prec = os.environ.get("NBE_PRECISION", "f16x3")
    if prec not in _ENG:
        _ENG[prec] = _engine.Engine(device=0, mid_chan=8, compute_vel=True, precision=prec)
    return _ENG[prec]


def release():
    for e in _ENG.values():
        e.close()
    _ENG.clear()


def _batched(x, dx=None, s=None):
    x = np.asarray(x, dtype=np.float32)
    un = x.ndim == 4
    if un:
        x = x[None]
        dx = None if dx is None else np.asarray(dx, np.float32)[None]
    elif dx is not None:
        dx = np.asarray(dx, np.float32)
    if s is not None:
        s = np.asarray(s, dtype=np.float32)
        if s.ndim == 1:
            s = np.broadcast_to(s[None], (x.shape[0], s.shape[0]))
    return x, dx, s, un


def _kind(kernel_size, stride, transpose):
    if transpose:
        if kernel_size != 2:
            raise ValueError("the up-sampling layer has kernel_size 2 (style_layers_vel.py:159)")
        return "up"
    k = {(3, 1): "conv3", (1, 1): "skip", (2, 2): "down"}.get((kernel_size, stride))
    if k is None:
        raise ValueError("supported (kernel_size, stride): (3,1), (1,1), (2,2); got (%d,%d)" % (kernel_size, stride))
    return k


def _run_layer(kind, x, dx, w, dw, b, vel):
    """All batch elements of one layer; returns y or (y, dy)."""
    e = _eng()
    ys, dys = [], []
    for i in range(x.shape[0]):
        wi = w[i] if w.ndim == 6 else w
        dwi = None if dw is None else (dw[i] if dw.ndim == 6 else dw)
        if vel:
            y, dy = e.test_layer(kind, x[i], wi, b, dx=None if dx is None else dx[i], dw=dwi)
            ys.append(y); dys.append(dy)
        else:
            ys.append(e.test_layer(kind, x[i], wi, b))
    return (np.stack(ys), np.stack(dys)) if vel else np.stack(ys)


@dataclass(frozen=True)
class _ConvLayer:
    in_chan: int
    out_chan: int
    kernel_size: int = 3
    stride: int = 1
    _style = False
    _vel = False
    _transpose = False

    # ---- parameters (reference initialisers: lecun_normal weights, ones style_bias, zero bias)
    def init(self, key, *example_args):
        rng = np.random.default_rng(_seed_of(key))
        k, ci, co = self.kernel_size, self.in_chan, self.out_chan
        std = 1.0 / np.sqrt(ci * k ** 3)
        leaf = {"weight": _trunc_normal(rng, (co, ci, k, k, k), std), "bias": np.zeros(co, np.float32)}
        if self._style:
            leaf["style_weight"] = _trunc_normal(rng, (ci, getattr(self, "style_size", 2)), 1.0 / np.sqrt(ci))
            leaf["style_bias"] = np.ones(ci, np.float32)
        elif self._vel:
            leaf["dweight"] = _trunc_normal(rng, (co, ci, k, k, k), std)
        return {"params": leaf}

    def _weights(self, lp, s, first):
        """Per-sample (w, dw): the style layers modulate on the device (style_layers_vel.py:62-105)."""
        if not self._style:
            return np.asarray(lp["weight"], np.float32), (np.asarray(lp["dweight"], np.float32) if self._vel else None)
        e = _eng()
        ws, dws = [], []
        for si in s:
            r = e.test_modulate(lp["weight"], lp["style_weight"], lp["style_bias"], si, first, eps=self.eps, vel=self._vel)
            if self._vel:
                ws.append(r[0]); dws.append(r[1])
            else:
                ws.append(r)
        return np.stack(ws), (np.stack(dws) if self._vel else None)

    def _call(self, params, x, s=None, dx=None):
        lp = params["params"] if "params" in params else params
        x, dx, s, un = _batched(x, dx, s)
        if x.shape[1] != self.in_chan:
            raise ValueError("input has %d channels, layer expects %d" % (x.shape[1], self.in_chan))
        if self._style and (s is None or s.shape[-1] != 2):
            raise ValueError("style vector must have 2 entries ((Om-0.3)*5, Dz-1)")
        w, dw = self._weights(lp, s, dx is None)
        out = _run_layer(_kind(self.kernel_size, self.stride, self._transpose), x, dx, w, dw,
                         np.asarray(lp["bias"], np.float32), self._vel)
        if self._vel:
            return (out[0][0], out[1][0]) if un else out
        return out[0] if un else out


@dataclass(frozen=True)
class StyleConvBase3DVel(_ConvLayer):
    style_size: int = 2
    eps: float = 1e-8
    _style, _vel = True, True

    def apply(self, params, x, s, dx=None):
        return self._call(params, x, s, dx)
    __call__ = apply


@dataclass(frozen=True)
class StyleTransposeBase3DVel(_ConvLayer):
    kernel_size: int = 2
    style_size: int = 2
    eps: float = 1e-8
    _style, _vel, _transpose = True, True, True

    def apply(self, params, x, s, dx=None):
        return self._call(params, x, s, dx)
    __call__ = apply


@dataclass(frozen=True)
class StyleConvBase3D(_ConvLayer):
    style_size: int = 2
    eps: float = 1e-8
    _style = True

    def apply(self, params, x, s):
        return self._call(params, x, s)
    __call__ = apply


@dataclass(frozen=True)
class StyleConvTransposeBase3D(_ConvLayer):
    kernel_size: int = 2
    style_size: int = 2
    eps: float = 1e-8
    _style, _transpose = True, True

    def apply(self, params, x, s):
        return self._call(params, x, s)
    __call__ = apply


@dataclass(frozen=True)
class ConvBase3DVel(_ConvLayer):
    eps: float = 1e-8
    _vel = True

    def apply(self, params, x, dx=None):
        return self._call(params, x, None, dx)
    __call__ = apply


@dataclass(frozen=True)
class ConvTransposeBase3DVel(_ConvLayer):
    kernel_size: int = 2
    eps: float = 1e-8
    _vel, _transpose = True, True

    def apply(self, params, x, dx=None):
        return self._call(params, x, None, dx)
    __call__ = apply


@dataclass(frozen=True)
class ConvBase3D(_ConvLayer):
    def apply(self, params, x):
        return self._call(params, x)
    __call__ = apply


@dataclass(frozen=True)
class ConvTransposeBase3D(_ConvLayer):
    kernel_size: int = 2
    _transpose = True

    def apply(self, params, x):
        return self._call(params, x)
    __call__ = apply


@dataclass(frozen=True)
class LeakyReLU:
    negative_slope: float = 0.01

    def init(self, key, *a):
        return {"params": {}}

    def apply(self, params, x):
        x = np.asarray(x)
        return np.where(x >= 0, x, x * np.asarray(self.negative_slope, x.dtype))
    __call__ = apply


@dataclass(frozen=True)
class LeakyReLUVel:
    negative_slope: float = 0.01

    def init(self, key, *a):
        return {"params": {}}

    def apply(self, params, x, dx):
        """layers_vel.py:182-186: the tangent takes the slope branch at exactly 0."""
        x, dx = np.asarray(x), np.asarray(dx)
        sl = np.asarray(self.negative_slope, x.dtype)
        return np.where(x >= 0, x, x * sl), np.where(x > 0, dx, dx * sl)
    __call__ = apply


# ---- the reference's partials ------------------------------------------------------------------------------
StyleConv3DVel = partial(StyleConvBase3DVel, kernel_size=3, stride=1)
StyleSkip3DVel = partial(StyleConvBase3DVel, kernel_size=1, stride=1)
StyleDownSample3DVel = partial(StyleConvBase3DVel, kernel_size=2, stride=2)
StyleUpSample3DVel = StyleTransposeBase3DVel
StyleConv3D = partial(StyleConvBase3D, kernel_size=3, stride=1)
StyleSkip3D = partial(StyleConvBase3D, kernel_size=1, stride=1)
StyleDownSample3D = partial(StyleConvBase3D, kernel_size=2, stride=2)
StyleUpSample3D = StyleConvTransposeBase3D
Conv3DVel = partial(ConvBase3DVel, kernel_size=3, stride=1)
Skip3DVel = partial(ConvBase3DVel, kernel_size=1, stride=1)
DownSample3DVel = partial(ConvBase3DVel, kernel_size=2, stride=2)
UpSample3DVel = ConvTransposeBase3DVel
Conv3D = partial(ConvBase3D, kernel_size=3, stride=1)
Skip3D = partial(ConvBase3D, kernel_size=1, stride=1)
DownSample3D = partial(ConvBase3D, kernel_size=2, stride=2)
UpSample3D = ConvTransposeBase3D


# ---- blocks ------------------------------------------------------------------------------------------------
class _Block:
    """Blocks compose the layers above exactly as style_blocks_vel.py:40-85 (resample) and :96-166 (ResNet): channel
    rule mid = max(in, out), first conv in -> mid, last conv mid -> out; layer names conv_<i> / skip."""
    _style = False
    _vel = False
    _LAYERS = None        # (conv3, skip, down, up) constructors, set by the concrete classes

    def _layer_table(self):
        mid = max(self.in_chan, self.out_chan)
        convs = [c for c in self.seq if c in "CUD"]
        n = len(convs)
        tab = []
        for i, ch in enumerate(convs):
            ci = self.in_chan if i == 0 else mid
            co = self.out_chan if i == n - 1 else mid
            tab.append(("conv_%d" % i, ch, ci, co))
        return tab

    def _make(self, ch, ci, co):
        conv3, skip, down, up = self._LAYERS
        ctor = {"C": conv3, "S": skip, "D": down, "U": up}[ch]
        kw = dict(in_chan=ci, out_chan=co)
        if self._style:
            kw.update(style_size=self.style_size, eps=self.eps)
        elif self._vel:
            kw.update(eps=self.eps)
        return ctor(**kw)

    def init(self, key, *example_args):
        seed = _seed_of(key)
        tree = {}
        for i, (name, ch, ci, co) in enumerate(self._layer_table()):
            tree[name] = self._make(ch, ci, co).init(seed + 1000 * (i + 1))["params"]
        if self._resnet:
            tree["skip"] = self._make("S", self.in_chan, self.out_chan).init(seed + 999)["params"]
        return {"params": tree}

    def _call(self, params, x, s=None, dx=None):
        tree = params["params"] if "params" in params else params
        act = LeakyReLUVel() if self._vel else LeakyReLU()

        def run(layer, lp, x, dx):
            args = (x,) + ((s,) if self._style else ()) + ((dx,) if self._vel else ())
            r = layer.apply({"params": lp}, *args)
            return r if self._vel else (r, None)

        def activate(x, dx):
            return act.apply({}, x, dx) if self._vel else (act.apply({}, x), None)

        if not self._resnet:
            tab = {ch_i: t for ch_i, t in enumerate(self._layer_table())}
            ci = 0
            for ch in self.seq:
                if ch in "UD":
                    name, _, cin, cout = tab[ci]
                    x, dx = run(self._make(ch, cin, cout), tree[name], x, dx)
                    ci += 1
                elif ch == "A":
                    x, dx = activate(x, dx)
                else:
                    raise ValueError(f'Layer type "{ch}" not supported.')
            return (x, dx) if self._vel else x
        last_act = self.seq[-1] == "A"
        main = self.seq[:-1] if last_act else self.seq
        y, dy = run(self._make("S", self.in_chan, self.out_chan), tree["skip"], x, dx)
        ncv = main.count("C")
        if ncv > 0:
            crop = (Ellipsis,) + (slice(ncv, -ncv),) * 3
            y = y[crop]
            dy = None if dy is None else dy[crop]
        tab = self._layer_table()
        ci = 0
        for ch in main:
            if ch == "C":
                name, _, cin, cout = tab[ci]
                x, dx = run(self._make("C", cin, cout), tree[name], x, dx)
                ci += 1
            elif ch == "A":
                x, dx = activate(x, dx)
            else:
                raise ValueError(f'Layer type "{ch}" not supported. Use C (conv) or A (activation).')
        x = x + y
        if self._vel:
            dx = dx + dy
        if last_act:
            x, dx = activate(x, dx)
        return (x, dx) if self._vel else x


def _block(name, style, vel, resnet, layers):
    fields = [("seq", str)] + ([("style_size", int)] if style else []) + [("in_chan", int), ("out_chan", int)]

    def apply_style_vel(self, params, x, s, dx=None): return self._call(params, x, s, dx)
    def apply_style(self, params, x, s): return self._call(params, x, s)
    def apply_vel(self, params, x, dx=None): return self._call(params, x, None, dx)
    def apply_plain(self, params, x): return self._call(params, x)
    ap = {(True, True): apply_style_vel, (True, False): apply_style, (False, True): apply_vel, (False, False): apply_plain}[(style, vel)]
    ns = {"__annotations__": dict(fields + [("eps", float)]), "eps": 1e-8, "_style": style, "_vel": vel, "_resnet": resnet,
          "_LAYERS": layers, "apply": ap, "__call__": ap, "__doc__": _Block.__doc__}
    return dataclass(frozen=True)(type(name, (_Block,), ns))


_SV = (StyleConv3DVel, StyleSkip3DVel, StyleDownSample3DVel, StyleUpSample3DVel)
_S = (StyleConv3D, StyleSkip3D, StyleDownSample3D, StyleUpSample3D)
_V = (Conv3DVel, Skip3DVel, DownSample3DVel, UpSample3DVel)
_P = (Conv3D, Skip3D, DownSample3D, UpSample3D)
StyleResampleBlock3DVel = _block("StyleResampleBlock3DVel", True, True, False, _SV)
StyleResNetBlock3DVel = _block("StyleResNetBlock3DVel", True, True, True, _SV)
StyleResampleBlock3D = _block("StyleResampleBlock3D", True, False, False, _S)
StyleResNetBlock3D = _block("StyleResNetBlock3D", True, False, True, _S)
ResampleBlock3DVel = _block("ResampleBlock3DVel", False, True, False, _V)
ResNetBlock3DVel = _block("ResNetBlock3DVel", False, True, True, _V)
ResampleBlock3D = _block("ResampleBlock3D", False, False, False, _P)
ResNetBlock3D = _block("ResNetBlock3D", False, False, True, _P)
