#!/usr/bin/env python
"""Generates tests/golden/*.npz from the float64 oracle (oracle/), with seeded synthetic inputs.

Why the oracle and not the reference: the reference is pure Python on JAX/Flax, neither of which is
installed here (no network), so it cannot be imported to produce vectors; its own tests hold no golden
vectors and the pretrained weights are absent.  These fixtures therefore pin the ORACLE (regression +
cross-machine reproducibility) and give the GPU tests expected outputs without re-running the oracle at
full width.  The reference's own value-level pins are asserted separately in tests/test_oracle_pins.py.

Run from the repository root:  python tests/golden/make_golden.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import cosmology as C, layers as L, model as M, params as P, subbox as S  # noqa: E402

Z, OM = 0.5, 0.3
DZ, VF = float(C.growth_factor(Z, OM)), float(C.vel_norm(Z, OM))


def whole_net(seed_p, seed_x, mid, shape):
    p = P.synthetic_params(seed=seed_p, mid_chan=mid)
    x = np.random.default_rng(seed_x).standard_normal((1, 3) + shape).astype(np.float32)
    d, v = M.forward(p, x, OM, DZ, VF)
    return d[0], v[0]


def main():
    out = {}
    # 1. cosmology scalars
    zs = np.array([0.0, 0.5, 1.0, 2.0, 3.0])
    oms = np.array([0.1, 0.3, 0.5])
    zz, oo = np.meshgrid(zs, oms, indexing="ij")
    out["cosmo_z"], out["cosmo_Om"] = zz, oo
    out["cosmo_D"], out["cosmo_H"] = C.growth_factor(zz, oo), C.hubble_rate(zz, oo)
    out["cosmo_f"], out["cosmo_vel"] = C.growth_rate(zz, oo), C.vel_norm(zz, oo)
    out["cosmo_acc"], out["cosmo_dlogH"] = C.acc_norm(zz, oo), C.dlogH_dloga(zz, oo)

    # 2. whole network, narrow (mid_chan=8) and production width (mid_chan=64), smallest legal input
    d, v = whole_net(11, 5, 8, (104, 104, 112))
    out["net8_disp"], out["net8_vel"] = d, v
    out["net8_meta"] = np.array([11, 5, 8, 104, 104, 112])
    d, v = whole_net(1234, 6, 64, (104, 104, 104))
    out["net64_disp"], out["net64_vel"] = d, v
    out["net64_meta"] = np.array([1234, 6, 64, 104, 104, 104])

    # 3. process_box on a tiny periodic box (crops wrap the box many times over)
    p = P.synthetic_params(seed=21, mid_chan=8)
    box = np.random.default_rng(22).standard_normal((3, 16, 8, 8)).astype(np.float32)
    dis, vel = S.process_box(p, box, Z, OM, (16, 8, 8), (2, 1, 1))
    out["pbox_disp"], out["pbox_vel"] = dis, vel
    out["pbox_meta"] = np.array([21, 22, 8, 16, 8, 8, 2, 1, 1])

    # 4. weight modulation of one layer (first-layer rule on and off)
    rng = np.random.default_rng(31)
    w = (rng.standard_normal((16, 8, 3, 3, 3)) / np.sqrt(8 * 27)).astype(np.float32)
    sw = (rng.standard_normal((8, 2)) / np.sqrt(8)).astype(np.float32)
    sb = (1 + 0.1 * rng.standard_normal(8)).astype(np.float32)
    s = L.style_vector(OM, DZ)
    out["mod_w"], out["mod_sw"], out["mod_sb"] = w, sw, sb
    out["mod_wn"], out["mod_dw"] = L.modulate_weights_vel(sw, sb, w, s, False)
    _, out["mod_dw_first"] = L.modulate_weights_vel(sw, sb, w, s, True)

    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote golden_v1.npz:", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
