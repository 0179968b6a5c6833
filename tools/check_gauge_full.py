"""Production width (mid_chan 64), whole process_box path: the default two-product ("gauged") tangent against the general
three-product kernels (NBE_GAUGE=0) on the same box, for the three arithmetic modes.  Prints relative L2 differences."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import torch
import jax_nbody_emulator_with_dj_amd as J
from jax_nbody_emulator_with_dj_amd import models

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
size, ndiv = (N,) * 3, (N // 128,) * 3
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
box = torch.randn((3,) + size, device="cuda", generator=gen)
m = J.StyleNBodyEmulatorVelCore()
p = m.init(7)
rel = lambda a, b: float((a.float() - b.float()).pow(2).sum().sqrt() / b.float().pow(2).sum().sqrt())
for dtype, name in ((np.float32, "f16x3 (default)"), (np.float16, "f16")):
    out = {}
    for gauge in ("1", "0"):
        os.environ["NBE_GAUGE"] = gauge
        models.release_engines()
        proc = J.SubboxProcessor(m, p, J.SubboxConfig(size=size, ndiv=ndiv, dtype=dtype))
        out[gauge] = proc.process_box(box, 0.5, 0.3, show_progress=False)
    d1, v1 = out["1"]; d0, v0 = out["0"]
    print("%-16s %d^3 ndiv %d: gauged vs general rel-L2 disp %.2e vel %.2e; finite %s" % (
        name, N, ndiv[0], rel(d1, d0), rel(v1, v0), bool(torch.isfinite(v1).all())), flush=True)
os.environ.pop("NBE_GAUGE", None)
models.release_engines()
