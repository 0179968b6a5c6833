"""Timing of the reference-shaped call -- NumPy box in, NumPy fields out (subbox.py:139-219) -- against the resident
call, on one box in one process: the reference's default call (show_progress=True: a tqdm bar fed from a host thread),
pipelined (default), un-pipelined (NBE_HOST_PIPE=0), pinned input, CUDA tensors.  N = 1024 runs eight tiles of 512^3, pipelined
tile by tile (tile k + 1's planes go up and tile k - 1's fields come down under tile k).
NBE_PIPE_TRACE=1 prints the host-side timeline of every pipelined call.   python tools/time_host_path.py [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
os.environ.setdefault("NBE_PIPE_TRACE", "1")
import torch
import jax_nbody_emulator_with_dj_amd as J
from jax_nbody_emulator_with_dj_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = J.SubboxConfig(size=(N,) * 3, ndiv=(N // 128,) * 3)
m = J.StyleNBodyEmulatorVelCore()
p = m.init(1234)
emu = J.create_emulator(load_params=False, processor_config=cfg)
emu.processor.params = p
box = np.random.default_rng(0).standard_normal((3, N, N, N), dtype=np.float32)


def run(tag, x, n=3):
    for i in range(n):
        t0 = time.perf_counter()
        d, v = emu.process_box(x, 0.5, 0.3)                         # default arguments, as the reference is called
        if isinstance(d, torch.Tensor):
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-28s call %d: %.3f s  (%.2f Mvox/s)" % (tag, i, dt, N ** 3 / dt / 1e6), flush=True)
        del d, v


run("pageable in, pipelined", box, 4)
os.environ["NBE_HOST_PIPE"] = "0"
run("pageable in, un-pipelined", box, 2)
del os.environ["NBE_HOST_PIPE"]
pin = _lib.pinned_empty(box.shape, np.float32)
pin[...] = box
run("pinned in, pipelined", pin, 3)
t = torch.from_numpy(box).cuda()
run("CUDA tensors (resident)", t, 3)
