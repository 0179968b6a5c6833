import sys, numpy as np
sys.path.insert(0, '.')
import jax_nbody_emulator_with_dj_amd as J
from jax_nbody_emulator_with_dj_amd.engine import Engine
from oracle import params as P
Z, OM = 0.5, 0.3
size, ndiv = (128, 128, 64), (4, 2, 1)
p = P.synthetic_params(seed=61, mid_chan=8)
full = np.random.default_rng(62).standard_normal((3,) + size).astype(np.float32)
Dz, vf = float(J.growth_factor(Z, OM)), float(J.vel_norm(Z, OM))
res = {}
for prec in ("f32", "f16x3"):
    for mt in (0, 256):
        e = Engine(device=0, mid_chan=8, compute_vel=True, precision=prec)
        e.load_params(p, premodulated=False); e.set_cosmology(OM, Dz); e.set_max_tile(mt)
        print(prec, mt, "plan", e.plan_tiles(size, ndiv), flush=True)
        d, v = e.process_box(full, size, ndiv, ((48, 48),) * 3, Dz, vf)
        res[(prec, mt)] = (d, v)
        e.close()
ref = res[("f32", 0)]
for k, (d, v) in res.items():
    dd = np.abs(d - ref[0])
    bad = dd > 1e-3
    print(k, "max abs disp diff vs f32/mt0: %.3e" % dd.max(), "bad frac %.3f" % bad.mean(),
          "bad bbox", [(int(ix.min()), int(ix.max())) for ix in np.nonzero(bad)[1:]] if bad.any() else None, flush=True)
