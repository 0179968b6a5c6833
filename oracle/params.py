"""Oracle: parameter trees (test infrastructure only).

  * `synthetic_params`  -- a seeded stand-in for the absent pretrained blob
    (/root/reference/.MISSING_LARGE_BLOBS).  Leaf names and shapes follow
    style_layers_vel.py:55-75 / tests/test_style_nbody_emulator_vel_core.py:408-419:
    weight (Cout,Cin,k,k,k), bias (Cout,), style_weight (Cin,2), style_bias (Cin,).
    Reference init is lecun_normal / ones / zeros; here the bias is small and
    non-zero and style_bias is jittered around 1 so that every term of the
    algebra is exercised.
  * `premodulate` / `premodulate_vel` -- tree walkers of
    nbody_emulator.py:150-187 and :221-266 (first-layer rule :243-246).
"""

import numpy as np

from . import layers as L
from .cosmology import growth_factor
from .model import RESNET_BLOCKS, RESAMPLE_BLOCKS, BLOCK_SEQ


def layer_table(in_chan=3, out_chan=3, mid_chan=64):
    """[(block, layer, Cout, Cin, k)] for all 33 conv layers, execution order."""
    m1, m2 = mid_chan, 2 * mid_chan
    io = {
        'conv_l00': (in_chan, m1), 'conv_l01': (m1, m1), 'down_l0': (m1, m1),
        'conv_l1': (m1, m1), 'down_l1': (m1, m1), 'conv_l2': (m1, m1), 'down_l2': (m1, m1),
        'conv_c': (m1, m1), 'up_r2': (m1, m1), 'conv_r2': (m2, m1), 'up_r1': (m1, m1),
        'conv_r1': (m2, m1), 'up_r0': (m1, m1), 'conv_r00': (m2, m1), 'conv_r01': (m1, out_chan),
    }
    order = ['conv_l00', 'conv_l01', 'down_l0', 'conv_l1', 'down_l1', 'conv_l2', 'down_l2', 'conv_c',
             'up_r2', 'conv_r2', 'up_r1', 'conv_r1', 'up_r0', 'conv_r00', 'conv_r01']
    out = []
    for blk in order:
        cin, cout = io[blk]
        if blk in RESAMPLE_BLOCKS:
            out.append((blk, 'conv_0', cout, cin, 2))
            continue
        mid = max(cin, cout)                                  # style_blocks_vel.py:126
        out.append((blk, 'skip', cout, cin, 1))
        ncv = BLOCK_SEQ[blk].count('C')
        for i in range(ncv):
            ci = cin if i == 0 else mid
            co = cout if i == ncv - 1 else mid
            out.append((blk, 'conv_%d' % i, co, ci, 3))
    return out


def synthetic_params(seed=1234, in_chan=3, out_chan=3, mid_chan=64, dtype=np.float32):
    rng = np.random.default_rng(seed)
    tree = {}
    for blk, lay, co, ci, k in layer_table(in_chan, out_chan, mid_chan):
        fan_in = ci * k ** 3
        tree.setdefault(blk, {})[lay] = {
            'weight': (rng.standard_normal((co, ci, k, k, k)) / np.sqrt(fan_in)).astype(dtype),
            'bias': (0.05 * rng.standard_normal(co)).astype(dtype),
            'style_weight': (rng.standard_normal((ci, 2)) / np.sqrt(ci)).astype(dtype),
            'style_bias': (1.0 + 0.1 * rng.standard_normal(ci)).astype(dtype),
        }
    return {'params': tree}


def premodulate(params, z, Om, eps=1e-8, dtype=np.float64):
    """nbody_emulator.py:150-187."""
    Dz = growth_factor(z, Om)
    s = L.style_vector(Om, Dz, np.dtype(dtype))
    out = {'params': {}}
    for blk, bp in params['params'].items():
        out['params'][blk] = {}
        for lay, lp in bp.items():
            if 'style_weight' in lp:
                w = L.modulate_weights(lp['style_weight'], lp['style_bias'], lp['weight'], s, eps)
                out['params'][blk][lay] = {'weight': w, 'bias': np.asarray(lp['bias'])}
            else:
                out['params'][blk][lay] = lp
    return out


def premodulate_vel(params, z, Om, eps=1e-8, dtype=np.float64):
    """nbody_emulator.py:221-266 (first-layer rule :243-246)."""
    Dz = growth_factor(z, Om)
    s = L.style_vector(Om, Dz, np.dtype(dtype))
    out = {'params': {}}
    for blk, bp in params['params'].items():
        out['params'][blk] = {}
        for lay, lp in bp.items():
            if 'style_weight' in lp:
                first = blk == 'conv_l00' and lay in ('conv_0', 'skip')
                w, dw = L.modulate_weights_vel(lp['style_weight'], lp['style_bias'], lp['weight'],
                                               s, first, eps)
                out['params'][blk][lay] = {'weight': w, 'dweight': dw, 'bias': np.asarray(lp['bias'])}
            else:
                out['params'][blk][lay] = lp
    return out
