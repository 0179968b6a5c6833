import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import jax_nbody_emulator_with_dj_amd as J
from jax_nbody_emulator_with_dj_amd.models import get_engine, release_engines
from jax_nbody_emulator_with_dj_amd.engine import NBEError
from oracle import params as P
Z, OM = 0.5, 0.3
pad = ((48, 48),) * 3
Dz, vf = float(J.growth_factor(Z, OM)), float(J.vel_norm(Z, OM))
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
nfail = 0
for rep in range(12):
    m = J.StyleNBodyEmulatorVelCore(mid_chan=8)
    p = P.synthetic_params(seed=29, mid_chan=8)
    eng = get_engine(m, 0)
    eng.ensure_params(p, False)
    for size, ndiv, mt in (((64, 48, 56), (2, 1, 1), 512), ((32, 16, 24), (2, 1, 1), 0)):
        eng.set_max_tile(mt)
        gen = torch.Generator(device="cuda"); gen.manual_seed(31 + size[0])
        box = torch.randn((3,) + size, device="cuda", generator=gen)
        out = (torch.zeros_like(box), torch.zeros_like(box))
        for (om, dz) in ((OM, Dz), (0.25, 0.9)):
            eng.set_cosmology(om, dz)
            for i in range(3):
                try:
                    eng.process_box(box, size, ndiv, pad, dz, vf, out=out)
                except NBEError as e:
                    nfail += 1
                    d, v = out
                    bd, bv = ~torch.isfinite(d), ~torch.isfinite(v)
                    print(mode, "rep", rep, size, "cosmo", om, "iter", i, "non-finite disp", int(bd.sum()), "vel", int(bv.sum()),
                          "planes z:", torch.nonzero(bv.any(dim=(0, 2, 3))).flatten().tolist()[:20],
                          "ch:", bv.any(dim=(1, 2, 3)).tolist(), flush=True)
    release_engines()
print(mode, "failures:", nfail)
