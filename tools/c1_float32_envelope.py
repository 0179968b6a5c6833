"""The float32 envelope of BASELINE config 1: the float32 NumPy oracle against the float64 fixture
(tests/golden/golden_v3.npz) on the same (1,3,128,128,128) input.  Shows what ANY float32 evaluation differs by from
the float64 one: the displacement by rounding (rel-L2 2.7e-7), the velocity additionally by LeakyReLU activations
that change branch (sparse outliers).  CPU only, about 2.5 minutes on 8 cores.  Result: profiles/r02_c1_float32_envelope.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model as M, params as P, cosmology as C
g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v3.npz"))
p = P.synthetic_params(seed=1234, mid_chan=64)
x = np.random.default_rng(0).standard_normal((1, 3, 128, 128, 128)).astype(np.float32)
Dz, vf = float(C.growth_factor(0.5, 0.3)), float(C.vel_norm(0.5, 0.3))
t = time.time()
d, v = M.forward(p, x, 0.3, Dz, vf, dtype=np.float32)
print("float32 oracle: %.0f s" % (time.time() - t))
for name, a, b in (("disp", d[0], g["c1_disp"]), ("vel", v[0], g["c1_vel"])):
    a = a.astype(np.float64); rms = np.sqrt(np.mean(b * b)); e = np.abs(a - b) / rms
    print(name, "rel_l2 %.3e  max/rms %.3e  frac>2e-4 %.4f  frac>1e-3 %.5f  inlier(<=1e-3) rel_l2 %.3e  median %.2e  p99 %.2e  p99.9 %.2e"
          % (np.linalg.norm(a - b) / np.linalg.norm(b), e.max(), (e > 2e-4).mean(), (e > 1e-3).mean(),
             np.sqrt(np.sum(((a - b) ** 2)[e <= 1e-3])) / np.linalg.norm(b), np.median(e), np.percentile(e, 99), np.percentile(e, 99.9)))
