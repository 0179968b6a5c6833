"""PCIe-inclusive timing of the reference-shaped call: NumPy box in, NumPy fields out."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import jax_nbody_emulator_with_dj_amd as J

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = J.SubboxConfig(size=(N,) * 3, ndiv=(N // 128,) * 3)
m = J.StyleNBodyEmulatorVelCore()
p = m.init(1234)
emu = J.create_emulator(load_params=False, processor_config=cfg)
emu.processor.params = p
box = np.random.default_rng(0).standard_normal((3, N, N, N), dtype=np.float32)
for i in range(3):
    t0 = time.perf_counter()
    d, v = emu.process_box(box, 0.5, 0.3, show_progress=False)
    print("call %d: %.3f s  (%.2f Mvox/s)  finite=%s" % (i, time.perf_counter() - t0, N ** 3 / (time.perf_counter() - t0) / 1e6,
                                                        bool(np.isfinite(d).all() and np.isfinite(v).all())), flush=True)
