"""GPU parity, edge cases the reference's tests walk through (tests/test_subbox.py:865-1000,
tests/test_nbody_emulator.py:775-863), against tests/golden/golden_v2.npz (float64 oracle):
asymmetric divisions with non-cubic crops, z = 0, high redshift, extreme Omega_m, batched apply with
per-sample cosmology, float16 inputs.  Same tolerances as tests/test_gpu_api.py."""

import os

import numpy as np
import pytest

import jax_nbody_emulator_with_dj_amd as J
from conftest import rel_l2, max_over_rms

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v2.npz"))


def _close(got, want, l2, mx, tag):
    assert got.shape == want.shape and np.all(np.isfinite(got)), tag
    e = rel_l2(got, want), max_over_rms(got, want)
    assert e[0] <= l2 and e[1] <= mx, "%s: rel_l2=%.3e max/rms=%.3e" % (tag, *e)


def _params(seed, mid):
    from oracle import params as P
    return P.synthetic_params(seed=seed, mid_chan=mid)


def test_asymmetric_divisions_noncubic_crops():
    seed_p, seed_x, mid, s0, s1, s2, n0, n1, n2 = (int(v) for v in GOLD["asym_meta"])
    z, Om = (float(v) for v in GOLD["asym_cosmo"])
    p = _params(seed_p, mid)
    box = np.random.default_rng(seed_x).standard_normal((3, s0, s1, s2)).astype(np.float32)
    cfg = J.SubboxConfig(size=(s0, s1, s2), ndiv=(n0, n1, n2))
    assert cfg.crop_size == (16, 8, 8) and cfg.n_subboxes == 4
    emu = J.create_emulator(load_params=False, processor_config=cfg, mid_chan=mid)
    emu.processor.params = p
    dis, vel = emu.process_box(box, z, Om, show_progress=False)
    _close(dis, GOLD["asym_disp"], 2e-5, 2e-4, "asymmetric disp")
    _close(vel, GOLD["asym_vel"], 5e-5, 2e-4, "asymmetric vel")
    # the caller's grid, run exactly, gives the same field
    from jax_nbody_emulator_with_dj_amd.models import get_engine
    eng = get_engine(emu.model, 0)
    try:
        eng.set_max_tile(0)
        dis0, vel0 = emu.process_box(box, z, Om, show_progress=False)
    finally:
        eng.set_max_tile(512)
    _close(dis0, GOLD["asym_disp"], 2e-5, 2e-4, "asymmetric disp, caller's grid")
    _close(vel0, GOLD["asym_vel"], 5e-5, 2e-4, "asymmetric vel, caller's grid")


def test_cosmology_sweep_and_batched_apply():
    """z = 0, z = 3, Om = 0.1, Om = 0.5: once one by one through NBodyEmulator.apply, once as ONE batched
    call with per-sample (Om, Dz, vel_fac) -- the reference vmaps the modulation over the batch."""
    seed_p, seed_x, mid, d0, d1, d2 = (int(v) for v in GOLD["cosmo_meta"])
    p = _params(seed_p, mid)
    x = np.random.default_rng(seed_x).standard_normal((1, 3, d0, d1, d2)).astype(np.float32)
    grid = GOLD["cosmo_grid"]
    emu = J.create_emulator(load_params=False, mid_chan=mid)
    emu.params = p
    for i, (z, Om) in enumerate(grid):
        d, v = emu.apply(x, z, Om)
        _close(d[0], GOLD["cosmo_disp"][i], 2e-5, 2e-4, "z=%g Om=%g disp" % (z, Om))
        _close(v[0], GOLD["cosmo_vel"][i], 5e-5, 2e-4, "z=%g Om=%g vel" % (z, Om))
    xb = np.repeat(x, len(grid), axis=0)
    d, v = emu(xb, grid[:, 0], grid[:, 1])                       # __call__ = apply
    assert d.shape == (len(grid), 3, 8, 8, 8)
    for i in range(len(grid)):
        _close(d[i], GOLD["cosmo_disp"][i], 2e-5, 2e-4, "batched disp %d" % i)
        _close(v[i], GOLD["cosmo_vel"][i], 5e-5, 2e-4, "batched vel %d" % i)


def test_reduced_precision_io_dtype():
    """dtype=float16 (tests/test_subbox.py:598-625): the model runs on the float16 engine (float16 operands and
    activations, float32 accumulation) and returns float16, as the reference does for a float16 input."""
    seed_p, seed_x, mid, d0, d1, d2 = (int(v) for v in GOLD["cosmo_meta"])
    p = _params(seed_p, mid)
    x = np.random.default_rng(seed_x).standard_normal((1, 3, d0, d1, d2)).astype(np.float32)
    emu16 = J.create_emulator(load_params=False, mid_chan=mid, dtype=np.float16)
    emu16.params = p
    d, v = emu16.apply(x, 0.0, 0.3)
    assert d.dtype == np.float16 and v.dtype == np.float16
    ref = GOLD["cosmo_disp"][0]
    assert rel_l2(d[0].astype(np.float64), ref) < 5e-3          # 11-bit operands through ~21 layers
    from jax_nbody_emulator_with_dj_amd import models
    assert any(k[-1] == "f16" for k in models._ENGINES)         # it really ran on the float16 engine


def test_device_tensors_are_ordered_with_torch_default_stream(engine_factory):
    """CUDA tensors in and out: the engine must enqueue on torch's current stream -- including the default
    (null) stream, which a private non-blocking stream is NOT ordered with.  The input is written by a copy
    queued behind ~100 ms of torch work; an engine running ahead of the null stream would read zeros."""
    import torch
    p = _params(7, 8)
    e = engine_factory(mid_chan=8, compute_vel=True)
    e.load_params(p, premodulated=False)
    e.set_cosmology(0.3, 0.8)
    x_host = np.random.default_rng(3).standard_normal((3, 104, 104, 104)).astype(np.float32)
    d_ref, v_ref = e.forward(x_host, 0.8, 50.0)                 # host path: synchronous
    dev = torch.device("cuda:0")
    src = torch.from_numpy(x_host).to(dev)
    x = torch.zeros_like(src)
    a = torch.randn(4096, 4096, device=dev)
    a /= a.norm()
    torch.cuda.synchronize()
    for _ in range(60):
        a = a @ a                                               # keeps the default stream busy
    x.copy_(src)                                                # ... and only then fills the input
    d, v = e.forward(x, 0.8, 50.0)
    d_host, v_host = d.cpu().numpy(), v.cpu().numpy()           # torch's copy must see the finished result
    assert np.array_equal(d_host, d_ref) and np.array_equal(v_host, v_ref)
