"""GPU: kink-aware (causal) parity of the velocity -- tests/kink.py explains the method.

Narrow models (mid_chan 8) walk the branch probe through every schedule the engine has (whole tensors, z-slabs, periodic
tiles, merged and caller's tiles, blocks whose cones wrap around the periodic box or straddle slab boundaries); the
production-width cases are BASELINE config 1 over its WHOLE output, one 224^3 -> 128^3 sub-box (the unit of configs 2-5,
with the float64 fixture tests/golden/golden_v4.npz beside it) and the timed configuration itself: config 3 on the
default kernels and the default plan."""

import os

import numpy as np
import pytest

import jax_nbody_emulator_with_dj_amd as J
from conftest import rel_l2, max_over_rms
import kink

pytestmark = pytest.mark.gpu

Z, OM = 0.5, 0.3
DZ, VF = 0.7731811501855036, 50.537651303131064
HERE = os.path.dirname(__file__)


def _synthetic(seed, mid):
    from oracle import params as P
    return P.synthetic_params(seed=seed, mid_chan=mid)


def _block(a, o, n):
    return np.asarray(a)[:, o[0]:o[0] + n, o[1]:o[1] + n, o[2]:o[2] + n]


def _probe_box(eng, box, size, ndiv, origin, nout, vf=VF):
    eng.probe_begin(origin, nout)
    try:
        d, v = eng.process_box(box, size, ndiv, ((48, 48),) * 3, DZ, vf)
        br = eng.probe_read()
    finally:
        eng.probe_end()
    return d, v, br


# ---- the probe's bookkeeping, every schedule, narrow model ---------------------------------------------------------
SCHEDULES = [
    # name, ndiv, max_tile, periodic, slab, block origins
    ("one periodic tile, slabs of 32", (2, 1, 1), 512, 1, 32, [(120, 56, 0), (40, 8, 24)]),
    ("one periodic tile, one slab", (2, 1, 1), 512, 1, -1, [(56, 0, 56)]),
    ("one padded tile, slabs of 32", (2, 1, 1), 512, 0, 32, [(64, 16, 8)]),
    ("two periodic-yx tiles (caller's grid)", (2, 1, 1), 0, 1, 32, [(72, 0, 56), (8, 24, 16)]),
    ("eight padded tiles, whole tensors", (2, 2, 2), 0, 1, -1, [(72, 8, 48), (0, 0, 0)]),
]


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
@pytest.mark.parametrize("sched", SCHEDULES, ids=[s[0] for s in SCHEDULES])
def test_probe_and_cone_oracle_through_every_schedule(engine_factory, prec, sched):
    _, ndiv, max_tile, periodic, slab, origins = sched
    if prec == "f32" and sched is not SCHEDULES[0]:
        pytest.skip("the probe's frames do not depend on the arithmetic: strict float32 walks the first schedule only")
    mid, size = 8, (128, 64, 64)
    p = _synthetic(81, mid)
    box = np.random.default_rng(82).standard_normal((3,) + size).astype(np.float32)
    e = engine_factory(mid_chan=mid, compute_vel=True, precision=prec)
    e.load_params(p, premodulated=False)
    e.set_cosmology(OM, DZ)
    e.set_max_tile(max_tile); e.set_periodic(periodic); e.set_slab(slab)
    try:
        for o in (origins if prec == "f16x3" else origins[:1]):
            d, v, br = _probe_box(e, box, size, ndiv, o, 8)
            xin = kink.cone_input_periodic(box, o, 8)
            d_o, v_o, st = kink.oracle_cone(p, xin, OM, DZ, VF, br, backend='numpy')
            # a frame error would flip about half of a tensor's branches
            kink.assert_cone("%s %s block %s" % (prec, sched[0], o), _block(d, o, 8), _block(v, o, 8), d_o, v_o, st, max_flip_frac=1e-4)
    finally:
        e.set_max_tile(512); e.set_periodic(1); e.set_slab(-1)


def test_probe_on_a_single_input_and_a_larger_block(engine_factory):
    """nbe_forward (whole tensors of one padded input) with a 16^3 block."""
    mid = 8
    p = _synthetic(83, mid)
    x = np.random.default_rng(84).standard_normal((3, 136, 120, 128)).astype(np.float32)
    for prec in ("f16x3",):
        e = engine_factory(mid_chan=mid, compute_vel=True, precision=prec)
        e.load_params(p, premodulated=False)
        e.set_cosmology(OM, DZ)
        o = (24, 8, 16)
        e.probe_begin(o, 16)
        try:
            d, v = e.forward(x, DZ, VF)
            br = e.probe_read()
        finally:
            e.probe_end()
        d_o, v_o, st = kink.oracle_cone(p, kink.cone_input_valid(x, o, 16), OM, DZ, VF, br, backend='numpy')
        kink.assert_cone("%s single input" % prec, _block(d, o, 16), _block(v, o, 16), d_o, v_o, st, max_flip_frac=1e-4)


def test_probe_refuses_a_block_that_straddles_tiles(engine_factory):
    from jax_nbody_emulator_with_dj_amd.engine import NBEError
    mid, size = 8, (128, 64, 64)
    e = engine_factory(mid_chan=mid, compute_vel=True, precision="f16x3")
    e.load_params(_synthetic(81, mid), premodulated=False)
    e.set_cosmology(OM, DZ)
    e.set_max_tile(0)
    box = np.zeros((3,) + size, np.float32)
    try:
        e.probe_begin((56, 0, 0), 16)                               # tiles are 64 planes deep: 56 .. 72 lies in two of them
        e.process_box(box, size, (2, 1, 1), ((48, 48),) * 3, DZ, VF)
        with pytest.raises(NBEError, match="inside one tile"):
            e.probe_read()
    finally:
        e.probe_end(); e.set_max_tile(512)


# ---- BASELINE config 1 at production width: the WHOLE output, causally --------------------------------------------
@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_config1_velocity_is_exact_given_the_branches(engine_factory, prec):
    """StyleNBodyEmulatorVelCore.apply on (1,3,128,128,128), mid_chan 64 (BASELINE config 1): the cone of the whole 32^3
    output is the whole input.  With the library's own branch decisions the float64 oracle must reproduce displacement AND
    velocity at the plain tolerances on every voxel (no medians, no quotas), and every branch that differs from the
    float64 oracle's must sit on a pre-activation within 1e-4 RMS of zero."""
    gold = np.load(os.path.join(HERE, "golden", "golden_v3.npz"))
    seed_p, seed_x, mid, d0, d1, d2 = (int(v) for v in gold["c1_meta"])
    p = _synthetic(seed_p, mid)
    x = np.random.default_rng(seed_x).standard_normal((3, d0, d1, d2)).astype(np.float32)
    e = engine_factory(mid_chan=mid, compute_vel=True, precision=prec)
    e.load_params(p, premodulated=False)
    e.set_cosmology(OM, DZ)
    e.probe_begin((0, 0, 0), 32)
    try:
        d, v = e.forward(x, DZ, VF)
        br = e.probe_read()
    finally:
        e.probe_end()
    d_o, v_o, st = kink.oracle_cone(p, x, OM, DZ, VF, br)
    r = kink.assert_cone("config 1 %s" % prec, d, v, d_o, v_o, st)
    # the displacement does not depend on the branches: the committed fixture (natural branches) holds it as well
    assert rel_l2(d, gold["c1_disp"]) <= 2e-5 and max_over_rms(d, gold["c1_disp"]) <= 2e-4
    # and the branches explain the whole difference to the fixture's velocity
    print("config 1 %s: against the fixture's own branches the velocity differs by %.2e (max %.2e RMS), with the library's by %.2e"
          % (prec, rel_l2(v, gold["c1_vel"]), max_over_rms(v, gold["c1_vel"]), r["vel"][0]))


# ---- one sub-box at the reference's shape: 224^3 -> 128^3, production width -----------------------------------------
@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_subbox_224_matches_float64_fixture_and_cone_oracle(engine_factory, prec):
    """The unit of BASELINE configs 2-5 (the deeper levels at their real sizes, the 40 / 16 / 4 crops): displacement against
    the float64 fixture on every stored voxel; velocity causally on a cone, and against the fixture in the median."""
    gold = np.load(os.path.join(HERE, "golden", "golden_v4.npz"))
    seed_p, seed_x, mid, n = (int(v) for v in gold["t224_meta"][:4])
    p = _synthetic(seed_p, mid)
    x = np.random.default_rng(seed_x).standard_normal((1, 3, n, n, n)).astype(np.float32)[0]
    e = engine_factory(mid_chan=mid, compute_vel=True, precision=prec)
    e.load_params(p, premodulated=False)
    e.set_cosmology(OM, DZ)
    o = (56, 104, 16)
    e.probe_begin(o, 8)
    try:
        d, v = e.forward(x, DZ, VF)
        br = e.probe_read()
    finally:
        e.probe_end()
    assert d.shape == (3, 128, 128, 128)
    rms_d, rms_v = (float(r) for r in gold["t224_rms"])
    for name, got in (("s4", d[:, ::4, ::4, ::4]), ("c16", d[:, 56:72, 56:72, 56:72])):
        want = gold["t224_disp_" + name]
        ed = rel_l2(got, want), float(np.abs(got - want).max() / rms_d)
        print("224^3 sub-box %s disp %s: %.2e / %.2e" % (prec, name, *ed))
        assert ed[0] <= 2e-5 and ed[1] <= 2e-4, (name, ed)
    ev = np.abs(np.concatenate([(v[:, ::4, ::4, ::4] - gold["t224_vel_s4"]).ravel(), (v[:, 56:72, 56:72, 56:72] - gold["t224_vel_c16"]).ravel()])) / rms_v
    print("224^3 sub-box %s vel against the fixture's own branches: median %.2e, %.2f %% beyond 2e-4 RMS, max %.2e" % (prec, np.median(ev), 100 * (ev > 2e-4).mean(), ev.max()))
    assert np.median(ev) <= 5e-6
    # the displacement-only twin (style_nbody_emulator_core.py:101-175; f16x3: conv_h3w_kernel<., NOVEL> with fused skips) against
    # the same fixture: vel primal == non-vel (tests/test_style_nbody_emulator_vel_core.py), at the reference's sub-box shape
    en = engine_factory(mid_chan=mid, compute_vel=False, precision=prec)
    en.load_params(p, premodulated=False)
    en.set_cosmology(OM, DZ)
    en.profile_enable(True)
    dn = en.forward(x, DZ)
    en.profile_enable(False)
    if prec == "f16x3":
        assert any(k["kernel"].startswith("conv_h3w<FLAT3,novel>") for k in en.profile_read())
    for name, got in (("s4", dn[:, ::4, ::4, ::4]), ("c16", dn[:, 56:72, 56:72, 56:72])):
        want = gold["t224_disp_" + name]
        ed = rel_l2(got, want), float(np.abs(got - want).max() / rms_d)
        print("224^3 sub-box %s displacement-only %s: %.2e / %.2e" % (prec, name, *ed))
        assert ed[0] <= 2e-5 and ed[1] <= 2e-4, (name, ed)
    en.close()                                                  # (its workspace with it)
    if prec == "f16x3":         # (strict float32 takes the same schedule through the same probe: config 1 covers it causally)
        d_o, v_o, st = kink.oracle_cone(p, kink.cone_input_valid(x, o, 8), OM, DZ, VF, br)
        kink.assert_cone("224^3 sub-box %s block %s" % (prec, o), _block(d, o, 8), _block(v, o, 8), d_o, v_o, st, rms_d=rms_d, rms_v=rms_v)


# ---- the timed configuration: config 3 on the default kernels and the default plan ---------------------------------
def test_config3_default_kernels_and_plan_against_cone_oracle_and_callers_grid():
    """BASELINE config 3 (512^3, ndiv 4, disp + vel, production width) exactly as bench.py times it -- default kernels
    (conv_h3w_kernel, Winograd F(2,3) along z), the plan the card's memory gives (a free card: the whole box as one periodic
    tile in z-slabs of 128) -- checked (a) causally against the float64 oracle on two cones, one that wraps around the box
    in x and straddles a slab boundary in z and one in the interior, and (b) against the caller's own grid of 64 padded
    224^3 sub-boxes on the direct kernel: displacement at the plain tolerances on every voxel of the box, velocity causally
    on the same cones (both runs against the oracle with their own branches)."""
    import torch
    from jax_nbody_emulator_with_dj_amd.models import get_engine, release_engines
    m = J.StyleNBodyEmulatorVelCore()
    p = m.init(1234)
    size, ndiv = (512,) * 3, (4,) * 3
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    box = torch.randn((3,) + size, device="cuda", generator=gen)
    hbox = box.cpu().numpy()
    eng = get_engine(m, 0)
    eng.ensure_params(p, False)
    eng.set_cosmology(OM, DZ)
    free_gb = torch.cuda.mem_get_info()[0] / 1e9
    plan = eng.plan_tiles(size, ndiv)
    if free_gb > 240:
        assert plan == (1, 1, 1), (plan, free_gb)
    cones = [(160, 248, 504), (320, 64, 128)]         # encoder slabs start at plane 40 + 128 k of the padded frame
    runs = {}
    eng.profile_enable(True); eng.profile_reset()
    for o in cones:
        d, v, br = _probe_box(eng, box, size, ndiv, o, 8)
        runs[o] = (_block(d.cpu().numpy(), o, 8).copy(), _block(v.cpu().numpy(), o, 8).copy(), br)
    names = [k["kernel"] for k in eng.profile_read()]
    eng.profile_enable(False)
    if plan == (1, 1, 1):
        assert (eng.query("slab"), eng.query("periodic_yx"), eng.query("periodic_z"), eng.query("gauge_active")) == (128.0, 1.0, 1.0, 1.0)
        assert any(n.startswith("conv_h3w") for n in names), names       # the kernel bench.py's roofline line is about
    d1, v1 = d, v
    rms_d, rms_v = float(d1.pow(2).mean().sqrt()), float(v1.pow(2).mean().sqrt())
    for o in cones:
        d_o, v_o, st = kink.oracle_cone(p, kink.cone_input_periodic(hbox, o, 8), OM, DZ, VF, runs[o][2])
        kink.assert_cone("config 3 default plan %s, cone %s" % (plan, o), runs[o][0], runs[o][1], d_o, v_o, st, rms_d=rms_d, rms_v=rms_v)
    # (b) the caller's grid on the direct kernel
    os.environ["NBE_WINO"] = "0"
    try:
        eng.set_max_tile(0)
        assert eng.plan_tiles(size, ndiv) == (4, 4, 4)
        o = cones[1]
        d0, v0, br0 = _probe_box(eng, box, size, ndiv, o, 8)
    finally:
        eng.set_max_tile(512)
        del os.environ["NBE_WINO"]
    ed = float((d1 - d0).norm() / d0.norm()), float((d1 - d0).abs().max()) / rms_d
    e = (v1 - v0).abs_() / rms_v
    print("config 3: default plan vs caller's grid on the direct kernel: disp %.2e / %.2e; vel median %.2e, %.3f %% beyond 2e-4 RMS, max %.2e"
          % (*ed, float(e.flatten()[::61].median()), 100 * float((e > 2e-4).float().mean()), float(e.max())))
    assert ed[0] <= 3e-6 and ed[1] <= 5e-5
    d_o, v_o, st = kink.oracle_cone(p, kink.cone_input_periodic(hbox, o, 8), OM, DZ, VF, br0)
    kink.assert_cone("config 3 caller's grid, direct kernel, cone %s" % (o,), _block(d0.cpu().numpy(), o, 8), _block(v0.cpu().numpy(), o, 8),
                     d_o, v_o, st, rms_d=rms_d, rms_v=rms_v)
    del d0, v0, d1, v1, box, e
    release_engines()
    torch.cuda.empty_cache()
