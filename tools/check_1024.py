"""BASELINE config 5 on one card: process_box 1024^3, ndiv (8,8,8), disp+vel, resident tensors.  Size-independent
property: translation equivariance on the periodic box -- rolling the input by a multiple of 8 voxels rolls both
fields by the same amount (checked on the whole arrays)."""
import os as _os
# schedules are compared bit for bit: on the direct gauged kernel (the Winograd-z kernel's rounding depends on how a launch
# pairs its planes, i.e. on the schedule -- tests/conftest.py::direct_kernels)
_os.environ.setdefault("NBE_WINO", "0")
import sys, time
sys.path.insert(0, ".")
import torch
import jax_nbody_emulator_with_dj_amd as J
from jax_nbody_emulator_with_dj_amd.models import get_engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = J.StyleNBodyEmulatorVelCore()
p = m.init(1234)
eng = get_engine(m, 0)
eng.ensure_params(p, False)
Dz, vf = float(J.growth_factor(0.5, 0.3)), float(J.vel_norm(0.5, 0.3))
eng.set_cosmology(0.3, Dz)
size, ndiv = (N,) * 3, (N // 128,) * 3
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
box = torch.randn((3,) + size, device="cuda", generator=gen)
print("plan", eng.plan_tiles(size, ndiv), flush=True)
pad = ((48, 48),) * 3
t0 = time.perf_counter(); d1, v1 = eng.process_box(box, size, ndiv, pad, Dz, vf); torch.cuda.synchronize()
print("first call %.2f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); d1, v1 = eng.process_box(box, size, ndiv, pad, Dz, vf); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("second call %.2f s = %.1f Mvox/s, workspace %.0f GB" % (dt, N ** 3 / dt / 1e6, eng.workspace_bytes() / 1e9), flush=True)
sh = (136, 264, 72)
# the workspace was sized to the memory that was free at the first call: park the first result on the host and roll the
# input in place of the original before the second run, then compare channel by channel
ok = bool(torch.isfinite(d1).all()) and bool(torch.isfinite(v1).all())
rms_d, rms_v = float(d1.pow(2).mean().sqrt()), float(v1.pow(2).mean().sqrt())
d1c, v1c = d1.cpu(), v1.cpu()
del d1, v1
box = torch.roll(box, sh, dims=(1, 2, 3))
torch.cuda.empty_cache()
d2, v2 = eng.process_box(box, size, ndiv, pad, Dz, vf)
torch.cuda.synchronize()
ed = ev = 0.0
for c in range(3):
    ed = max(ed, float((torch.roll(d1c[c].cuda(), sh, dims=(0, 1, 2)) - d2[c]).abs().max()))
    ev = max(ev, float((torch.roll(v1c[c].cuda(), sh, dims=(0, 1, 2)) - v2[c]).abs().max()))
print("finite %s; translation equivariance: max|delta|/rms disp %.2e vel %.2e" % (ok, ed / rms_d, ev / rms_v), flush=True)
