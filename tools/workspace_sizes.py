"""Workspace of the 512^3 / ndiv 4 box under the tiling / schedule knobs (fresh engine each: the workspace never shrinks)."""
import sys
sys.path.insert(0, ".")
import torch
from jax_nbody_emulator_with_dj_amd.engine import Engine
from jax_nbody_emulator_with_dj_amd import StyleNBodyEmulatorVelCore

p = StyleNBodyEmulatorVelCore().init(1)
x = torch.zeros((3, 512, 512, 512), device="cuda")
for prec in ("f16x3", "f16"):
    for mt, sl in ((512, -1), (512, 0), (512, 64), (256, 0), (0, 0)):
        e = Engine(device=0, precision=prec)
        e.load_params(p, False)
        e.set_cosmology(0.3, 0.77)
        e.set_max_tile(mt)
        e.set_slab(sl)
        plan = e.plan_tiles((512,) * 3, (4,) * 3)
        e.process_box(x, (512,) * 3, (4,) * 3, ((48, 48),) * 3, 0.77, 50.0)
        torch.cuda.synchronize()
        print("%s max_tile %3d slab %3d -> tiles %s workspace %.1f GB" % (prec, mt, sl, plan, e.workspace_bytes() / 1e9), flush=True)
        e.close()
