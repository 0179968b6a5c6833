"""GPU parity of the whole network (engine level, through the C ABI) against the float64 oracle.

Tolerance: relative L2 <= 2e-5 (displacement), <= 5e-5 (velocity); max|delta|/RMS <= 2e-4.
Rationale: ~21 sequential float32 contractions with K <= 3456; the float32-vs-float64 envelope of
the oracle itself on these inputs is ~1e-6 (tests/test_oracle_pins.py)."""

import numpy as np
import pytest

from conftest import rel_l2, max_over_rms

pytestmark = pytest.mark.gpu

DZ, VF, OM = 0.7731811501855036, 50.537651303131064, 0.3


def _check(d, v, d_o, v_o, tag):
    assert np.all(np.isfinite(d)) and (v is None or np.all(np.isfinite(v)))
    e = rel_l2(d, d_o), max_over_rms(d, d_o)
    assert e[0] <= 2e-5 and e[1] <= 2e-4, "%s disp rel_l2=%.3e max/rms=%.3e" % (tag, *e)
    if v is not None:
        e = rel_l2(v, v_o), max_over_rms(v, v_o)
        assert e[0] <= 5e-5 and e[1] <= 2e-4, "%s vel rel_l2=%.3e max/rms=%.3e" % (tag, *e)


@pytest.fixture(scope="module")
def small():
    from oracle import params as P, model as M
    rng = np.random.default_rng(5)
    p = P.synthetic_params(seed=11, mid_chan=8)
    x = rng.standard_normal((3, 104, 104, 112)).astype(np.float32)
    d_o, v_o = M.forward(p, x[None], OM, DZ, VF)
    return p, x, d_o[0], v_o[0]


PRECS = ["f32", "f16x3"]


@pytest.mark.parametrize("prec", PRECS)
def test_style_vel_small(engine_factory, small, prec):
    p, x, d_o, v_o = small
    e = engine_factory(mid_chan=8, compute_vel=True, precision=prec)
    e.load_params(p, premodulated=False)
    e.set_cosmology(OM, DZ)
    d, v = e.forward(x, DZ, VF)
    assert d.shape == (3, 8, 8, 16)
    _check(d, v, d_o, v_o, "style-vel mid8")
    # determinism: bit-identical on repeat (tests/test_nbody_emulator.py:842-863)
    d2, v2 = e.forward(x, DZ, VF)
    assert np.array_equal(d, d2) and np.array_equal(v, v2)


@pytest.mark.parametrize("prec", PRECS)
def test_style_novel_small(engine_factory, small, prec):
    p, x, d_o, _ = small
    e = engine_factory(mid_chan=8, compute_vel=False, precision=prec)
    e.load_params(p, premodulated=False)
    e.set_cosmology(OM, DZ)
    d = e.forward(x, DZ)
    _check(d, None, d_o, None, "style-novel mid8")


@pytest.mark.parametrize("prec", PRECS)
def test_premod_vel_small(engine_factory, small, prec):
    from oracle import params as P
    p, x, d_o, v_o = small
    pp = P.premodulate_vel(p, 0.5, OM)
    e = engine_factory(mid_chan=8, compute_vel=True, precision=prec)
    e.load_params(pp, premodulated=True)
    d, v = e.forward(x, DZ, VF)
    _check(d, v, d_o, v_o, "premod-vel mid8")


@pytest.mark.parametrize("prec", PRECS)
def test_gauged_tangent_paths(engine_factory, small, monkeypatch, prec):
    """Style path with velocity: the two-product gauged tangent (default; conv_h3g_kernel / conv_mfma_kernel<G6>) against the
    three-product general kernels (NBE_GAUGE=0), and the fall-back to those when a style factor is exactly zero
    (alpha = s'/s does not exist then).  Same tolerances as everywhere: both forms are float32-equivalent."""
    import copy
    from oracle import model as M
    p, x, d_o, v_o = small

    def run(params):
        e = engine_factory(mid_chan=8, compute_vel=True, precision=prec)
        e.load_params(params, premodulated=False)
        e.set_cosmology(OM, DZ)
        e.profile_enable(True)
        d, v = e.forward(x, DZ, VF)
        e.profile_enable(False)
        return d, v, any(k["kernel"].startswith(("conv_h3g", "conv_mfma_g")) for k in e.profile_read())

    d1, v1, g1 = run(p)
    monkeypatch.setenv("NBE_GAUGE", "0")
    d0, v0, g0 = run(p)
    monkeypatch.delenv("NBE_GAUGE")
    assert g1 and not g0
    _check(d1, v1, d_o, v_o, "gauged")
    _check(d0, v0, d_o, v_o, "general")
    assert rel_l2(d1, d0) <= 2e-6 and rel_l2(v1, v0) <= 5e-6

    pz = copy.deepcopy(p)
    pz["params"]["conv_l1"]["conv_1"]["style_weight"][3] = 0
    pz["params"]["conv_l1"]["conv_1"]["style_bias"][3] = 0
    dz_o, vz_o = M.forward(pz, x[None], OM, DZ, VF)
    dz, vz, gz = run(pz)
    assert not gz, "a zero style factor must switch the gauged kernels off"
    _check(dz, vz, dz_o[0], vz_o[0], "zero style factor")

    # a style factor that is merely tiny: alpha = s'/s = 1000 would push dx + alpha x towards the float16 range
    pt = copy.deepcopy(p)
    pt["params"]["conv_r1"]["conv_0"]["style_weight"][5] = (0.0, 1.0)
    pt["params"]["conv_r1"]["conv_0"]["style_bias"][5] = -(DZ - 1.0) + 1e-3
    dt_o, vt_o = M.forward(pt, x[None], OM, DZ, VF)
    dt, vt, gt = run(pt)
    assert not gt, "|alpha| > 64 must switch the gauged kernels off"
    _check(dt, vt, dt_o[0], vt_o[0], "tiny style factor")


@pytest.mark.parametrize("switch", ["NBE_H3G_TALL", "NBE_STEM", "NBE_UP8", "NBE_NARROW", "NBE_HEAD4", "NBE_WINO"])
def test_kernel_ab_switches_keep_parity(engine_factory, small, monkeypatch, switch):
    """Every A/B switch of the f16x3 velocity path selects kernels that stay held to the oracle: the 2 x 4 wave tile of
    conv_h3g_kernel (its zero-select once sat next to an asm MFMA, tests/test_mfma_hazards.py), the general first-layer kernel,
    eight up-sampling launches, the wide tile for the head, one output plane per workgroup on the narrow tile, and the direct gauged
    kernel in place of the Winograd-z one (conv_h3w_kernel)."""
    p, x, d_o, v_o = small
    monkeypatch.setenv(switch, "0")
    e = engine_factory(mid_chan=8, compute_vel=True, precision="f16x3")
    e.load_params(p, premodulated=False)
    e.set_cosmology(OM, DZ)
    e.profile_enable(True)
    d, v = e.forward(x, DZ, VF)
    e.profile_enable(False)
    names = [k["kernel"] for k in e.profile_read()]
    _check(d, v, d_o, v_o, switch + "=0")
    if switch == "NBE_STEM":
        assert not any(n.startswith("stem_h3") for n in names)
    if switch == "NBE_UP8":
        assert not any(n.startswith("up_h3") for n in names)
    if switch == "NBE_WINO":
        assert not any(n.startswith("conv_h3w") for n in names)
    monkeypatch.delenv(switch)
    e = engine_factory(mid_chan=8, compute_vel=True, precision="f16x3")     # (the head's tile is wired when the weights load)
    e.load_params(p, premodulated=False)
    e.set_cosmology(OM, DZ)
    e.profile_enable(True)
    d1, v1 = e.forward(x, DZ, VF)
    e.profile_enable(False)
    names = [k["kernel"] for k in e.profile_read()]
    _check(d1, v1, d_o, v_o, switch + " default")
    assert any(n.startswith("stem_h3") for n in names) and any(n.startswith("up_h3") for n in names)
    assert any(n.startswith("conv_h3w") for n in names)          # the conv_0 layers of the blocks run on the Winograd-z kernel
    assert rel_l2(d1, d) <= 2e-6 and rel_l2(v1, v) <= 5e-6


@pytest.mark.parametrize("prec", PRECS)
def test_premodulated_pairs_are_recognised(engine_factory, small, prec):
    """Premodulated (W, dW) pairs made by modulate_emulator_parameters_vel factorise as dW = W (alpha[ci] + beta[co]); the
    engine recognises that from the numbers and runs the two-product kernels.  A pair that does not factorise (one
    perturbed dweight element) must leave the network on the general kernels."""
    import copy
    from oracle import params as P
    p, x, d_o, v_o = small
    pp = P.premodulate_vel(p, 0.5, OM)

    def run(params):
        e = engine_factory(mid_chan=8, compute_vel=True, precision=prec)
        e.load_params(params, premodulated=True)
        e.profile_enable(True)
        d, v = e.forward(x, DZ, VF)
        e.profile_enable(False)
        return d, v, any(k["kernel"].startswith(("conv_h3g", "conv_mfma_g")) for k in e.profile_read())

    d, v, g = run(pp)
    assert g, "factorising premodulated weights should run the gauged kernels"
    _check(d, v, d_o, v_o, "premod gauged")
    pq = copy.deepcopy(pp)
    dw = pq["params"]["conv_l1"]["conv_0"]["dweight"]
    dw = np.array(dw, copy=True)
    dw[1, 2, 0, 1, 2] += 1e-3 * np.abs(dw).max()
    pq["params"]["conv_l1"]["conv_0"]["dweight"] = dw
    d2, v2, g2 = run(pq)
    assert not g2, "a dweight that does not factorise must keep the general kernels"
    assert rel_l2(d2, d) <= 2e-6 and rel_l2(v2, v_o) <= 1e-3


def test_float16_mode(engine_factory, small, monkeypatch):
    """The float16 engine ("f16": float16 operands and stored activations, float32 accumulation -- the
    arithmetic of the reference's dtype=float16 configuration) against the float64 oracle.  Every one of the
    ~21 sequential layers rounds its operands to 11 significant bits, and the velocity is a tangent through
    all of them; a NumPy emulation of exactly that rounding gives rel-L2 4e-4 (disp) / 1.4e-2 (vel) at
    mid_chan=64.  Tolerance: rel-L2 <= 2e-3 (disp), <= 4e-2 (vel)."""
    import os
    from oracle import params as P
    p, x, d_o, v_o = small
    e = engine_factory(mid_chan=8, compute_vel=True, precision="f16")
    e.load_params(p, premodulated=False)
    e.set_cosmology(OM, DZ)
    d, v = e.forward(x, DZ, VF)
    assert np.all(np.isfinite(d)) and np.all(np.isfinite(v))
    print("f16 mid8: disp rel_l2 %.3e vel rel_l2 %.3e" % (rel_l2(d, d_o), rel_l2(v, v_o)))
    assert rel_l2(d, d_o) <= 2e-3 and rel_l2(v, v_o) <= 4e-2
    d2, v2 = e.forward(x, DZ, VF)
    assert np.array_equal(d, d2) and np.array_equal(v, v2)
    e0 = engine_factory(mid_chan=8, compute_vel=False, precision="f16")
    e0.load_params(p, premodulated=False)
    e0.set_cosmology(OM, DZ)
    assert np.array_equal(e0.forward(x, DZ), d)                 # the primal does not depend on the tangent path
    monkeypatch.setenv("NBE_GAUGE", "0")                        # three-product tangent (default: gauged, two products)
    eg = engine_factory(mid_chan=8, compute_vel=True, precision="f16")
    eg.load_params(p, premodulated=False)
    eg.set_cosmology(OM, DZ)
    dg, vg = eg.forward(x, DZ, VF)
    monkeypatch.delenv("NBE_GAUGE")
    print("f16 mid8, general tangent: vel rel_l2 %.3e; gauged vs general %.3e" % (rel_l2(vg, v_o), rel_l2(v, vg)))
    assert np.array_equal(dg, d) and rel_l2(vg, v_o) <= 4e-2

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
    seed_p, seed_x, mid, d0, d1, d2_ = (int(t) for t in gold["net64_meta"])
    p = P.synthetic_params(seed=seed_p, mid_chan=mid)
    x = np.random.default_rng(seed_x).standard_normal((1, 3, d0, d1, d2_)).astype(np.float32)[0]
    e = engine_factory(mid_chan=mid, compute_vel=True, precision="f16")
    e.load_params(p, premodulated=False)
    e.set_cosmology(OM, DZ)
    e.profile_reset(); e.profile_enable(True)
    d, v = e.forward(x, DZ, VF)
    e.profile_enable(False)
    prof = e.profile_read()
    names = [k["kernel"] for k in prof]
    skips = sum(k["launches"] for k in prof if k["kernel"].startswith("conv_h1<FLAT1,vel,dx>"))
    # production width: the gauged 3x3x3 layers run the Winograd-z form (conv_h3w_kernel<., ., F16>) wherever a launch has an
    # even number of planes, with the block's 1x1x1 skip fused into conv_1 there (conv_l00, whose skip reads three channels,
    # and a conv_c with an odd number of planes keep theirs as launches of their own); NBE_WINO=0 puts everything on
    # conv_h2q_kernel and the general 1x1x1 kernel -- both within the float16 tolerances
    assert any(n.startswith("conv_h1w<FLAT3") for n in names), names
    print("f16 mid64: disp rel_l2 %.3e vel rel_l2 %.3e" % (rel_l2(d, gold["net64_disp"]), rel_l2(v, gold["net64_vel"])))
    assert rel_l2(d, gold["net64_disp"]) <= 2e-3 and rel_l2(v, gold["net64_vel"]) <= 4e-2
    monkeypatch.setenv("NBE_WINO", "0")
    e.profile_reset(); e.profile_enable(True)
    d0_, v0_ = e.forward(x, DZ, VF)
    e.profile_enable(False)
    monkeypatch.delenv("NBE_WINO")
    assert not any(k["kernel"].startswith("conv_h1w") for k in e.profile_read())
    skips0 = sum(k["launches"] for k in e.profile_read() if k["kernel"].startswith("conv_h1<FLAT1,vel,dx>"))
    print("f16 mid64: %d skip launches with fusion, %d without" % (skips, skips0))
    assert skips0 == 9 - 1 and skips <= 2        # (conv_l00's skip has no input tangent: another kernel name)
    print("f16 mid64, direct kernels: disp rel_l2 %.3e vel rel_l2 %.3e; Winograd-z vs direct %.3e / %.3e" % (
        rel_l2(d0_, gold["net64_disp"]), rel_l2(v0_, gold["net64_vel"]), rel_l2(d, d0_), rel_l2(v, v0_)))
    assert rel_l2(d0_, gold["net64_disp"]) <= 2e-3 and rel_l2(v0_, gold["net64_vel"]) <= 4e-2


def test_full_width_c1_slice(engine_factory):
    """mid_chan=64 (production width) on the smallest legal input, against the committed golden
    fixture (float64 oracle output, tests/golden/make_golden.py), both arithmetic modes."""
    import os
    from oracle import params as P
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
    seed_p, seed_x, mid, d0, d1, d2 = (int(v) for v in gold["net64_meta"])
    p = P.synthetic_params(seed=seed_p, mid_chan=mid)
    x = np.random.default_rng(seed_x).standard_normal((1, 3, d0, d1, d2)).astype(np.float32)[0]
    for prec in PRECS:
        e = engine_factory(mid_chan=mid, compute_vel=True, precision=prec)
        e.load_params(p, premodulated=False)
        e.set_cosmology(OM, DZ)
        d, v = e.forward(x, DZ, VF)
        print("mid64 %s: disp rel_l2 %.3e vel rel_l2 %.3e" % (prec, rel_l2(d, gold["net64_disp"]), rel_l2(v, gold["net64_vel"])))
        _check(d, v, gold["net64_disp"], gold["net64_vel"], "style-vel mid64 " + prec)
