"""Oracle: blocks and the four emulator cores (test infrastructure only).

Follows
  style_blocks_vel.py:40-85    StyleResampleBlock3DVel  ('DA' / 'UA')
  style_blocks_vel.py:96-166   StyleResNetBlock3DVel    (skip 1x1x1, crop by #convs,
                               C/A sequence, residual add, optional final act)
  style_nbody_emulator_vel_core.py:45-103, :105-195   U-Net wiring, crops 48/40/16/4,
                               concat [skip, up], head
  nbody_emulator_vel_core.py:102-183, style_nbody_emulator_core.py:101-175,
  nbody_emulator_core.py:98-166   the premodulated / displacement-only twins
  (identical wiring; they differ in where the weights come from and whether a
  tangent is carried).

`forward` evaluates any of the four variants:
    premodulated=False, compute_vel=True   StyleNBodyEmulatorVelCore.apply(params, x, Om, Dz, vel_fac)
    premodulated=False, compute_vel=False  StyleNBodyEmulatorCore.apply(params, x, Om, Dz)
    premodulated=True,  compute_vel=True   NBodyEmulatorVelCore.apply(params, x, Dz, vel_fac)
    premodulated=True,  compute_vel=False  NBodyEmulatorCore.apply(params, x, Dz)
"""

import numpy as np

from . import layers as L

# (block name, kind, seq) in execution order -- style_nbody_emulator_vel_core.py:50-103
RESNET_BLOCKS = ('conv_l00', 'conv_l01', 'conv_l1', 'conv_l2', 'conv_c',
                 'conv_r2', 'conv_r1', 'conv_r00', 'conv_r01')
RESAMPLE_BLOCKS = ('down_l0', 'down_l1', 'down_l2', 'up_r2', 'up_r1', 'up_r0')
BLOCK_SEQ = {
    'conv_l00': 'CACA', 'conv_l01': 'CACA', 'conv_l1': 'CACA', 'conv_l2': 'CACA',
    'conv_c': 'CACA', 'conv_r2': 'CACA', 'conv_r1': 'CACA', 'conv_r00': 'CACA',
    'conv_r01': 'CAC',
    'down_l0': 'DA', 'down_l1': 'DA', 'down_l2': 'DA',
    'up_r2': 'UA', 'up_r1': 'UA', 'up_r0': 'UA',
}


class _Weights:
    """Hands out (w, dw, b) per layer in the working dtype."""

    def __init__(self, params, premodulated, compute_vel, s, dtype, eps):
        self.p = params['params'] if 'params' in params else params
        self.premod = premodulated
        self.vel = compute_vel
        self.s = s
        self.dt = np.dtype(dtype)
        self.eps = eps

    def get(self, block, layer, first_layer):
        lp = self.p[block][layer]
        dt = self.dt
        b = np.asarray(lp['bias'], dtype=dt)
        if self.premod:
            w = np.asarray(lp['weight'], dtype=dt)
            dw = np.asarray(lp['dweight'], dtype=dt) if self.vel else None
            return w, dw, b
        # the reference modulates in float32 and then casts to x.dtype
        # (style_layers_vel.py:103-105); the oracle modulates in its working dtype.
        if self.vel:
            w, dw = L.modulate_weights_vel(lp['style_weight'], lp['style_bias'], lp['weight'],
                                           self.s, first_layer, self.eps)
            return w, dw, b
        w = L.modulate_weights(lp['style_weight'], lp['style_bias'], lp['weight'], self.s, self.eps)
        return w, None, b


def _layer(W, vel, block, layer, kind, x, dx):
    first = dx is None
    w, dw, b = W.get(block, layer, first)
    if vel:
        return L.conv_layer_vel(kind, x, dx, w, dw, b)
    return L.conv_layer(kind, x, w, b), None


def _act(W, vel, x, dx, name):
    """LeakyReLU[Vel] on freshly computed arrays, in place.  `name` is 'block/layer' of the convolution whose output
    is activated; W.branch_hook(name, x), when set, supplies the branch the tangent takes (layers.leaky_relu_vel)."""
    if vel:
        hook = getattr(W, 'branch_hook', None)
        return L.leaky_relu_vel_(x, dx, branch=None if hook is None else hook(name, x))
    return L.leaky_relu_(x), None


def resnet_block(W, vel, name, x, dx):
    """style_blocks_vel.py:96-166.  The reference evaluates the 1x1x1 skip on the whole input and crops it by the
    number of convolutions; a 1x1x1 layer commutes with cropping, so the skip is evaluated on the cropped input here
    (same values), after the main branch, and added in place -- at production width a level-0 tensor pair is 11 GB."""
    seq = BLOCK_SEQ[name]
    last_act = seq[-1] == 'A'
    main = seq[:-1] if last_act else seq
    ncv = main.count('C')
    x_in, dx_in = x, dx
    ci = 0
    for ch in main:
        if ch == 'C':
            x, dx = _layer(W, vel, name, 'conv_%d' % ci, 'conv3', x, dx)
            ci += 1
        elif ch == 'A':
            x, dx = _act(W, vel, x, dx, '%s/conv_%d' % (name, ci - 1))
        else:
            raise ValueError('Layer type "%s" not supported. Use C (conv) or A (activation).' % ch)
    c = ncv
    y, dy = _layer(W, vel, name, 'skip', 'skip', _crop(x_in, c) if c else x_in, _crop(dx_in, c) if c else dx_in)
    L._iadd(x, y)
    if vel:
        L._iadd(dx, dy)
    if last_act:
        x, dx = _act(W, vel, x, dx, '%s/conv_%d' % (name, ci - 1))
    return x, dx


def resample_block(W, vel, name, x, dx):
    """style_blocks_vel.py:40-85."""
    ci = 0
    for ch in BLOCK_SEQ[name]:
        if ch == 'U':
            x, dx = _layer(W, vel, name, 'conv_%d' % ci, 'up', x, dx)
            ci += 1
        elif ch == 'D':
            x, dx = _layer(W, vel, name, 'conv_%d' % ci, 'down', x, dx)
            ci += 1
        elif ch == 'A':
            x, dx = _act(W, vel, x, dx, '%s/conv_%d' % (name, ci - 1))
        else:
            raise ValueError('Layer type "%s" not supported.' % ch)
    return x, dx


def _crop(a, c):
    return None if a is None else a[:, c:-c, c:-c, c:-c]


def _crop_copy(a, c):
    """A skip connection's centre crop as an array of its own, so that the uncropped tensor can be released."""
    return None if a is None else np.ascontiguousarray(a[:, c:-c, c:-c, c:-c])


def _cat(a, b):
    return None if a is None else np.concatenate([a, b], axis=0)


def forward_single(params, x, Om, Dz, vel_fac, premodulated, compute_vel,
                   dtype=np.float64, eps=1e-8, branch_hook=None):
    """One batch element.  x (C, D, H, W).  Returns disp or (disp, vel).

    branch_hook(name, pre_activation) -> boolean array or None: called at every LeakyReLUVel with the 'block/layer'
    name of the convolution it follows; a returned array replaces `pre_activation > 0` as the branch of the TANGENT
    (tests/kink.py: the oracle evaluated with the branch decisions of the evaluation under test)."""
    dt = np.dtype(dtype)
    vel = compute_vel
    Dz = dt.type(Dz)
    s = None if premodulated else L.style_vector(Om, Dz, dt)
    W = _Weights(params, premodulated, compute_vel, s, dt, eps)
    W.branch_hook = branch_hook

    x = np.asarray(x, dtype=dt) * (Dz / dt.type(6.0))          # core :132-134
    dx = None
    x0 = x[:, 48:-48, 48:-48, 48:-48]                          # core :139

    x0 = np.ascontiguousarray(x0)
    x, dx = resnet_block(W, vel, 'conv_l00', x, dx)
    y0, dy0 = resnet_block(W, vel, 'conv_l01', x, dx)
    x, dx = resample_block(W, vel, 'down_l0', y0, dy0)
    y0, dy0 = _crop_copy(y0, 40), _crop_copy(dy0, 40)

    y1, dy1 = resnet_block(W, vel, 'conv_l1', x, dx)
    x, dx = resample_block(W, vel, 'down_l1', y1, dy1)
    y1, dy1 = _crop_copy(y1, 16), _crop_copy(dy1, 16)

    y2, dy2 = resnet_block(W, vel, 'conv_l2', x, dx)
    x, dx = resample_block(W, vel, 'down_l2', y2, dy2)
    y2, dy2 = _crop_copy(y2, 4), _crop_copy(dy2, 4)

    x, dx = resnet_block(W, vel, 'conv_c', x, dx)

    x, dx = resample_block(W, vel, 'up_r2', x, dx)
    x, dx = _cat(y2, x), _cat(dy2, dx)
    x, dx = resnet_block(W, vel, 'conv_r2', x, dx)

    x, dx = resample_block(W, vel, 'up_r1', x, dx)
    x, dx = _cat(y1, x), _cat(dy1, dx)
    x, dx = resnet_block(W, vel, 'conv_r1', x, dx)

    x, dx = resample_block(W, vel, 'up_r0', x, dx)
    x, dx = _cat(y0, x), _cat(dy0, dx)
    x, dx = resnet_block(W, vel, 'conv_r00', x, dx)
    x, dx = resnet_block(W, vel, 'conv_r01', x, dx)

    disp = (x + x0) * dt.type(6.0)                             # core :187
    if not vel:
        return disp
    vf = dt.type(vel_fac)
    velocity = dx * (vf * dt.type(6.0)) + x0 * (vf * dt.type(6.0) / Dz)   # core :190-193
    return disp, velocity


def forward(params, x, Om=None, Dz=None, vel_fac=None, premodulated=False, compute_vel=True,
            dtype=np.float64, eps=1e-8, branch_hook=None):
    """Batched front end: x (B, C, D, H, W); Om, Dz, vel_fac scalars or (B,)."""
    x = np.asarray(x)
    B = x.shape[0]
    bc = lambda v: None if v is None else np.broadcast_to(np.atleast_1d(np.asarray(v, dtype=np.float64)), (B,))
    Om_, Dz_, vf_ = bc(Om), bc(Dz), bc(vel_fac)
    outs = [forward_single(params, x[i],
                           None if Om_ is None else Om_[i], Dz_[i],
                           None if vf_ is None else vf_[i],
                           premodulated, compute_vel, dtype, eps, branch_hook) for i in range(B)]
    if compute_vel:
        return np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs])
    return np.stack(outs)
