"""One rank's share of a 2-GPU run on one card: a (256, 512, 512) brick haloed along z only (the axis the rank grid
splits); along y and x the region is the periodic box itself, so the engine runs its periodic mode."""
import sys, time
sys.path.insert(0, ".")
import torch
from jax_nbody_emulator_with_dj_amd.engine import Engine
from jax_nbody_emulator_with_dj_amd import StyleNBodyEmulatorVelCore

e = Engine(device=0)
e.load_params(StyleNBodyEmulatorVelCore().init(1), False)
e.set_cosmology(0.3, 0.77)
for name, shape, origin in (("z-haloed brick (periodic y/x)", (3, 352, 512, 512), (48, 0, 0)),
                            ("fully haloed brick (padded)", (3, 352, 608, 608), (48, 48, 48))):
    H = torch.randn(shape, device="cuda")
    disp = torch.zeros((3, 256, 512, 512), device="cuda")
    vel = torch.zeros_like(disp)
    for it in range(2):
        e.profile_reset(); e.profile_enable(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.process_region(H, origin, (256, 512, 512), (1, 1, 1), 0.77, 50.0, disp, vel, order=[0])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        e.profile_enable(False)
        fl = sum(p["flops"] for p in e.profile_read()) / 1e12
    print("%s: %.3f s, %.0f TFLOP, finite %s" % (name, dt, fl, bool(torch.isfinite(disp).all())), flush=True)
    del H, disp, vel
