#!/usr/bin/env python
"""Fourth batch of golden vectors: ONE SUB-BOX AT THE REFERENCE'S OWN SHAPE AND WIDTH -- the unit of BASELINE configs 2-5.

  t224_*   StyleNBodyEmulatorVelCore.apply on a (1,3,224,224,224) sub-box, mid_chan 64, z = 0.5, Om = 0.3 ->
           (1,3,128,128,128) displacement and velocity: the deeper levels at their real sizes (108 / 52 / 24-voxel
           tensors, the 40 / 16 / 4 crops of the skip connections).  Float64 oracle (torch-CPU convolution core, pinned
           against the NumPy tap-wise GEMM in tests/test_oracle_pins.py); stored as float32:
             t224_disp_s4 / t224_vel_s4     every fourth voxel along each axis  [:, ::4, ::4, ::4]   (3, 32, 32, 32)
             t224_disp_c16 / t224_vel_c16   the central 16^3                    [:, 56:72, 56:72, 56:72]
           See make_golden.py for why the oracle and not the reference produces them (no JAX here: parity unpinned).

Run from the repository root:  python tests/golden/make_golden_v4.py      (about 15 minutes and 40 GB on 8 cores)
"""

import os
import resource
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import cosmology as C, layers as L, model as M, params as P  # noqa: E402

Z, OM = 0.5, 0.3
DZ, VF = float(C.growth_factor(Z, OM)), float(C.vel_norm(Z, OM))
SEED_PARAMS, SEED_INPUT, N = 1234, 224, 224


def main():
    t = time.time()
    p = P.synthetic_params(seed=SEED_PARAMS, mid_chan=64)
    x = np.random.default_rng(SEED_INPUT).standard_normal((1, 3, N, N, N)).astype(np.float32)
    with L.backend('torch'):
        d, v = M.forward(p, x, OM, DZ, VF)
    d, v = d[0], v[0]
    out = {
        "t224_disp_s4": d[:, ::4, ::4, ::4].astype(np.float32), "t224_vel_s4": v[:, ::4, ::4, ::4].astype(np.float32),
        "t224_disp_c16": d[:, 56:72, 56:72, 56:72].astype(np.float32), "t224_vel_c16": v[:, 56:72, 56:72, 56:72].astype(np.float32),
        "t224_rms": np.array([np.sqrt(np.mean(d * d)), np.sqrt(np.mean(v * v))]),
        "t224_meta": np.array([SEED_PARAMS, SEED_INPUT, 64, N, N, N]),
    }
    np.savez_compressed(os.path.join(HERE, "golden_v4.npz"), **out)
    print("wrote golden_v4.npz:", {k: np.asarray(a).shape for k, a in out.items()},
          "%.0f s, peak %.1f GB" % (time.time() - t, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))


if __name__ == "__main__":
    main()
