#!/usr/bin/env python
"""Third batch of golden vectors (float64 oracle, seeded synthetic inputs).  See make_golden.py for why the oracle
and not the reference produces them.

  c1_*     BASELINE config 1 at production width: StyleNBodyEmulatorVelCore.apply on one (1,3,128,128,128) sub-box,
           mid_chan 64, z = 0.5, Om = 0.3 -> (1,3,32,32,32) displacement and velocity.
  scale_*  one (1,3,104,104,104) sub-box at mid_chan 8 with the input multiplied by 10^k, k in SCALE_EXP: the range
           test of the split-f16 arithmetic (tests/test_gpu_range.py).  The network is not scale-free (biases), so
           every scale has its own expected fields.

Run from the repository root:  python tests/golden/make_golden_v3.py      (about 5 minutes on 8 cores)
"""

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import cosmology as C, model as M, params as P  # noqa: E402

Z, OM = 0.5, 0.3
DZ, VF = float(C.growth_factor(Z, OM)), float(C.vel_norm(Z, OM))
SCALE_EXP = (-12, -8, -4, -2, 0, 2, 4, 8, 12)


def main():
    out = {}
    t = time.time()
    p = P.synthetic_params(seed=1234, mid_chan=64)
    x = np.random.default_rng(0).standard_normal((1, 3, 128, 128, 128)).astype(np.float32)
    d, v = M.forward(p, x, OM, DZ, VF)
    out["c1_disp"], out["c1_vel"] = d[0], v[0]
    out["c1_meta"] = np.array([1234, 0, 64, 128, 128, 128])
    print("c1: %.0f s" % (time.time() - t), flush=True)

    p8 = P.synthetic_params(seed=51, mid_chan=8)
    x8 = np.random.default_rng(52).standard_normal((1, 3, 104, 104, 104)).astype(np.float32)
    ds, vs = [], []
    for k in SCALE_EXP:
        xs = (x8 * np.float32(10.0 ** k)).astype(np.float32)
        d, v = M.forward(p8, xs, OM, DZ, VF)
        ds.append(d[0]); vs.append(v[0])
    out["scale_exp"] = np.array(SCALE_EXP)
    out["scale_disp"], out["scale_vel"] = np.stack(ds), np.stack(vs)
    out["scale_meta"] = np.array([51, 52, 8, 104, 104, 104])
    np.savez_compressed(os.path.join(HERE, "golden_v3.npz"), **out)
    print("wrote golden_v3.npz:", {k: np.asarray(v).shape for k, v in out.items()}, "%.0f s" % (time.time() - t))


if __name__ == "__main__":
    main()
