"""Does the alignment of a tile's 512-byte row pieces matter to conv_h3w_kernel?  One 64 -> 64 layer through the gauged test hook
(dense tensors: a row of the output is (W - 2) * 16 B), W chosen so that output rows start 128-byte aligned or not.
Prints the kernel time per output voxel.  python tools/gpu/align_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jax_nbody_emulator_with_dj_amd.engine import Engine

e = Engine(device=0, compute_vel=True, precision="f16x3")
rng = np.random.default_rng(0)
cin = cout = 64
w = rng.standard_normal((cout, cin, 3, 3, 3)).astype(np.float32)
w /= np.sqrt((w.astype(np.float64) ** 2).sum(axis=(1, 2, 3, 4), keepdims=True)).astype(np.float32)
beta = (0.3 * rng.standard_normal(cout)).astype(np.float32)
b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
for W in (514, 515, 516, 518, 520, 522):
    D, H = 18, 258
    x = rng.standard_normal((cin, D, H, W)).astype(np.float32)
    dx = rng.standard_normal((cin, D, H, W)).astype(np.float32)
    for wino in ("1", "0"):
        os.environ["NBE_WINO"] = wino
        e.test_layer_gauged(x, dx, w, beta, b, act=True)           # warm
        e.profile_reset(); e.profile_enable(True)
        for _ in range(3):
            e.test_layer_gauged(x, dx, w, beta, b, act=True)
        e.profile_enable(False)
        k = [p for p in e.profile_read() if p["kernel"].startswith("conv_h3")][0]
        nvox = (D - 2) * (H - 2) * (W - 2)
        print("W_in %d  out row %5d B (mod 128 = %3d)  in row mod 128 = %3d  wino %s  %-24s %.3f ms  %.4f ns/voxel" % (
            W, (W - 2) * 16, ((W - 2) * 16) % 128, (W * 16) % 128, wino, k["kernel"], k["ms"] / k["launches"], 1e6 * k["ms"] / k["launches"] / nvox), flush=True)
e.close()
