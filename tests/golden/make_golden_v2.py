#!/usr/bin/env python
"""Second batch of golden vectors (float64 oracle, seeded synthetic inputs): the edge cases the reference's
tests walk through -- asymmetric divisions, z = 0, high redshift, extreme Omega_m (tests/test_subbox.py:865-1000),
batched apply with per-sample cosmology (style_layers_vel.py:129-141).  See make_golden.py for why the oracle and
not the reference produces them.   Run from the repository root:  python tests/golden/make_golden_v2.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import cosmology as C, model as M, params as P, subbox as S  # noqa: E402


def main():
    out = {}
    p = P.synthetic_params(seed=41, mid_chan=8)
    # 1. process_box with asymmetric divisions and non-cubic crops
    box = np.random.default_rng(42).standard_normal((3, 32, 16, 8)).astype(np.float32)
    dis, vel = S.process_box(p, box, 1.0, 0.25, (32, 16, 8), (2, 2, 1))
    out["asym_disp"], out["asym_vel"] = dis, vel
    out["asym_meta"] = np.array([41, 42, 8, 32, 16, 8, 2, 2, 1])
    out["asym_cosmo"] = np.array([1.0, 0.25])
    # 2. one sub-box at several cosmologies (z, Om): z = 0, high z, extreme Om
    x = np.random.default_rng(43).standard_normal((1, 3, 104, 104, 104)).astype(np.float32)
    cosmos = np.array([[0.0, 0.3], [3.0, 0.3], [0.5, 0.1], [0.5, 0.5]])
    ds, vs = [], []
    for z, Om in cosmos:
        Dz, vf = float(C.growth_factor(z, Om)), float(C.vel_norm(z, Om))
        d, v = M.forward(p, x, Om, Dz, vf)
        ds.append(d[0]); vs.append(v[0])
    out["cosmo_grid"] = cosmos
    out["cosmo_disp"], out["cosmo_vel"] = np.stack(ds), np.stack(vs)
    out["cosmo_meta"] = np.array([41, 43, 8, 104, 104, 104])
    np.savez_compressed(os.path.join(HERE, "golden_v2.npz"), **out)
    print("wrote golden_v2.npz:", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
