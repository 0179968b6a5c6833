"""Default engine schedule (merged tiles, z-slabs, periodic mode) against the reference-shaped one (the caller's grid,
whole tensors, padded) on a handful of box shapes / models / arithmetic modes.  mid_chan 8 and 16 for speed (16: the
decoder's concat is read from two tensors, which needs mid_chan % 16 == 0)."""
import os as _os, sys as _sys
# schedules are compared bit for bit: on the direct gauged kernel (the Winograd-z kernel's rounding depends on how a launch
# pairs its planes, i.e. on the schedule -- tests/conftest.py::direct_kernels).  With --winograd the default kernels run and the
# schedules are held to float32 rounding instead: displacement max|delta| <= 5e-5 RMS, velocity median <= 5e-6 RMS and 95 % of the
# voxels within 1e-3 RMS (LeakyReLU kinks, tests/test_gpu_range.py::_kink_robust_vel).
WINO = "--winograd" in _sys.argv
_os.environ["NBE_WINO"] = "1" if WINO else _os.environ.get("NBE_WINO", "0")
import os, sys, itertools
import numpy as np
sys.path.insert(0, ".")
import jax_nbody_emulator_with_dj_amd as J
from jax_nbody_emulator_with_dj_amd.models import get_engine, release_engines
from oracle import params as P

Z, OM = 0.5, 0.3
cases = [((64, 48, 56), (2, 1, 1)), ((96, 64, 48), (3, 1, 1)), ((48, 48, 48), (1, 1, 1)), ((128, 64, 64), (1, 2, 1)),
         ((72, 56, 48), (1, 1, 1)), ((40, 48, 64), (1, 1, 2)), ((160, 48, 48), (5, 1, 1)), ((56, 48, 48), (1, 1, 1))]
worst = 0.0
for prec, mid in itertools.product(("f16x3", "f32"), (8, 16)):
    os.environ["NBE_PRECISION"] = prec
    for vel, premod in itertools.product((True, False), (False, True)):
        p = P.synthetic_params(seed=5, mid_chan=mid)
        if premod:
            p = (J.modulate_emulator_parameters_vel if vel else J.modulate_emulator_parameters)(p, Z, OM)
        cls = {(True, False): J.StyleNBodyEmulatorVelCore, (False, False): J.StyleNBodyEmulatorCore,
               (True, True): J.NBodyEmulatorVelCore, (False, True): J.NBodyEmulatorCore}[(vel, premod)]
        m = cls(mid_chan=mid)
        for size, ndiv in cases:
            box = np.random.default_rng(sum(size)).standard_normal((3,) + size).astype(np.float32)
            proc = J.SubboxProcessor(m, p, J.SubboxConfig(size=size, ndiv=ndiv))
            eng = get_engine(m, 0)
            eng.set_max_tile(0); eng.set_slab(0); eng.set_periodic(False)
            ref = proc.process_box(box, Z, OM, show_progress=False)
            eng.set_max_tile(512); eng.set_slab(-1); eng.set_periodic(True)
            plan = eng.plan_tiles(size, ndiv)
            got = proc.process_box(box, Z, OM, show_progress=False)
            eng.set_slab(32)
            got2 = proc.process_box(box, Z, OM, show_progress=False)
            eng.set_slab(-1)
            def close(name, r, g, what):
                global worst
                rms = max(np.sqrt(np.mean(r.astype(np.float64) ** 2)), 1e-30)
                d = np.abs(g.astype(np.float64) - r) / rms
                assert np.isfinite(g).all(), (what, prec, vel, premod, size, ndiv, name)
                if WINO and name == "vel":
                    assert np.median(d) < 5e-6 and (d > 1e-3).mean() < 0.05, (what, prec, vel, premod, size, ndiv, name, float(np.median(d)), float((d > 1e-3).mean()))
                else:
                    worst = max(worst, float(d.max()))
                    assert d.max() < (5e-5 if WINO else 1e-5), (what, prec, vel, premod, size, ndiv, name, float(d.max()))
            for name, r, g in zip(("disp", "vel"), ref if vel else (ref,), got if vel else (got,)):
                close(name, r, g, "default vs reference-shaped")
            for name, r, g in zip(("disp", "vel"), got if vel else (got,), got2 if vel else (got2,)):
                if WINO:
                    close(name, r, g, "slab 32")
                else:
                    assert np.array_equal(r, g), ("slab 32 differs", prec, vel, premod, size, ndiv)
            print(prec, "mid", mid, "vel" if vel else "novel", "premod" if premod else "style", size, ndiv, "->", plan, "ok", flush=True)
        release_engines()
print("all schedules agree; worst max|delta|/rms %.2e" % worst)
