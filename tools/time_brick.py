"""One rank's share of an N-GPU run of the 512^3 box, timed on one card, for candidate rank grids.  A brick is haloed
only along the axes its rank grid splits; along the others the region is the periodic box itself (periodic mode)."""
import sys, time
sys.path.insert(0, ".")
import torch
from jax_nbody_emulator_with_dj_amd.engine import Engine
from jax_nbody_emulator_with_dj_amd import StyleNBodyEmulatorVelCore

e = Engine(device=0)
e.load_params(StyleNBodyEmulatorVelCore().init(1), False)
e.set_cosmology(0.3, 0.77)
N = 512
cases = [("N=2 (2,1,1)", (2, 1, 1)), ("N=4 (2,2,1)", (2, 2, 1)), ("N=4 (4,1,1)", (4, 1, 1)),
         ("N=8 (2,2,2)", (2, 2, 2)), ("N=8 (4,2,1)", (4, 2, 1))]
for name, grid in cases:
    b = tuple(N // g for g in grid)
    pa = tuple(48 if g > 1 else 0 for g in grid)
    H = torch.randn((3,) + tuple(bb + 2 * p for bb, p in zip(b, pa)), device="cuda")
    disp = torch.zeros((3,) + b, device="cuda")
    vel = torch.zeros_like(disp)
    nd_local = tuple(4 // g for g in grid)
    nd = e.plan_tiles(b, nd_local, periodic_box=False)
    order = list(range(nd[0] * nd[1] * nd[2]))
    for it in range(2):
        e.profile_reset(); e.profile_enable(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.process_region(H, pa, b, nd, 0.77, 50.0, disp, vel, order=order)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        e.profile_enable(False)
        fl = sum(p["flops"] for p in e.profile_read()) / 1e12
    print("%s: brick %s tiles %s: %.3f s, %.0f TFLOP per rank -> %.1f Mvox/s for the job" % (name, b, nd, dt, fl, N ** 3 / dt / 1e6), flush=True)
    del H, disp, vel
