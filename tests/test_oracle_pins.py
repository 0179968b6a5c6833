"""CPU: pin the oracle.

(1) every value-level known answer the reference's own tests/README hold for this path (SURVEY.md
    section 4 table), (2) the committed golden fixtures, (3) self-consistency that needs no reference:
    tangent == d(disp)/d(Dz) by float64 central differences, style == premodulated, conv core ==
    torch.nn.functional.conv3d, float32-vs-float64 envelope.
"""

import os

import numpy as np
import pytest

from oracle import cosmology as C, layers as L, model as M, params as P, subbox as S

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
Z, OM = 0.5, 0.3
DZ, VF = float(C.growth_factor(Z, OM)), float(C.vel_norm(Z, OM))


# ---- reference pins: LeakyReLU (tests/test_layers_vel.py:268-334, tests/test_layers.py:147-184) ------
def test_leaky_relu_values():
    x = np.array([-2.0, -1.0, 0.0, 1.0, 2.0])
    np.testing.assert_allclose(L.leaky_relu(x), [-0.02, -0.01, 0.0, 1.0, 2.0], rtol=1e-12)
    y, dy = L.leaky_relu_vel(x, np.ones_like(x), slope=0.1)
    np.testing.assert_allclose(y, [-0.2, -0.1, 0.0, 1.0, 2.0], rtol=1e-12)
    np.testing.assert_allclose(dy, [0.1, 0.1, 0.1, 1.0, 1.0], rtol=1e-12)      # x == 0 takes the slope branch
    _, dy2 = L.leaky_relu_vel(x, 3.0 * np.ones_like(x), slope=0.1)
    np.testing.assert_allclose(dy2, 3.0 * dy, rtol=1e-12)                      # dy scales with dx


# ---- reference pins: cosmology (tests/test_cosmology.py:18-23,34-38,92-108,126-139,173-194; README.md:175-180)
def test_cosmology_identities():
    assert abs(C.growth_factor(0.0, 0.3) - 1.0) < 1e-12
    assert abs(C.hubble_rate(0.0, 0.3) - 100.0) < 1e-12
    assert abs(C.growth_factor(1.0, 1.0 - 1e-12) - 0.5) < 1e-9                 # EdS: D = a
    assert abs(C.growth_rate(1.0, 1.0 - 1e-12) - 1.0) < 1e-9                   # EdS: f = 1
    for z in (5.0, 8.0):
        Omz = 0.3 * (1 + z) ** 3 / (0.3 * (1 + z) ** 3 + 0.7)
        assert abs(C.growth_rate(z, 0.3) / Omz ** 0.55 - 1) < 0.01
    h = 1e-5
    for z in (0.0, 0.5, 2.0):
        fd = -(1 + z) * (np.log(C.growth_factor(z + h, 0.3)) - np.log(C.growth_factor(z - h, 0.3))) / (2 * h)
        assert abs(fd - C.growth_rate(z, 0.3)) < 1e-8


def test_cosmology_readme_table():
    zs = np.array([0.0, 0.5, 1.0])
    np.testing.assert_allclose(C.growth_factor(zs, 0.3), [1.0, 0.77, 0.61], atol=0.006)
    np.testing.assert_allclose(C.hubble_rate(zs, 0.3), [100, 131, 176], atol=0.6)
    np.testing.assert_allclose(C.growth_rate(zs, 0.3), [0.51, 0.75, 0.87], atol=0.006)
    assert abs(DZ - 0.773181) < 1e-6 and abs(VF - 50.5377) < 1e-4


def test_cosmology_golden():
    zz, oo = GOLD["cosmo_z"], GOLD["cosmo_Om"]
    np.testing.assert_allclose(C.growth_factor(zz, oo), GOLD["cosmo_D"], rtol=1e-12)
    np.testing.assert_allclose(C.vel_norm(zz, oo), GOLD["cosmo_vel"], rtol=1e-12)
    np.testing.assert_allclose(C.acc_norm(zz, oo), GOLD["cosmo_acc"], rtol=1e-12)


# ---- convolution semantics ------------------------------------------------------------------------------
def test_conv_core_matches_torch():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 9, 10, 11))
    w3, w1, w2 = rng.standard_normal((7, 5, 3, 3, 3)), rng.standard_normal((7, 5, 1, 1, 1)), rng.standard_normal((7, 5, 2, 2, 2))
    tx = torch.tensor(x)[None]
    np.testing.assert_allclose(L.conv3(x, w3), F.conv3d(tx, torch.tensor(w3))[0].numpy(), atol=1e-12)
    np.testing.assert_allclose(L.conv1(x, w1), F.conv3d(tx, torch.tensor(w1))[0].numpy(), atol=1e-12)
    x2 = rng.standard_normal((5, 8, 10, 12))
    np.testing.assert_allclose(L.down2(x2, w2), F.conv3d(torch.tensor(x2)[None], torch.tensor(w2), stride=2)[0].numpy(), atol=1e-12)


def test_upsample_parity_form_equals_lhs_dilation():
    """style_layers_vel.py:236-244: lhs_dilation=2, padding 1, k=2  ==  y[2i+p] = W[..., 1-p] x[i]
    == stride-2 transposed conv with flipped kernel and swapped in/out."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(1)
    x, w = rng.standard_normal((5, 4, 5, 6)), rng.standard_normal((7, 5, 2, 2, 2))
    lit = L.up2_literal(x, w)
    assert lit.shape == (7, 8, 10, 12)
    np.testing.assert_allclose(L.up2(x, w), lit, atol=1e-12)
    wt = torch.tensor(w).flip(2, 3, 4).permute(1, 0, 2, 3, 4)
    np.testing.assert_allclose(F.conv_transpose3d(torch.tensor(x)[None], wt, stride=2)[0].numpy(), lit, atol=1e-12)


def test_modulation_golden_and_first_layer_rule():
    s = L.style_vector(OM, DZ)
    wn, dw = L.modulate_weights_vel(GOLD["mod_sw"], GOLD["mod_sb"], GOLD["mod_w"], s, False)
    np.testing.assert_allclose(wn, GOLD["mod_wn"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(dw, GOLD["mod_dw"], rtol=1e-12, atol=1e-15)
    _, dwf = L.modulate_weights_vel(GOLD["mod_sw"], GOLD["mod_sb"], GOLD["mod_w"], s, True)
    np.testing.assert_allclose(dwf, dw + wn / DZ, rtol=1e-12, atol=1e-15)      # style_layers_vel.py:94-101
    # unit filters after demodulation (style_layers_vel.py:87-90)
    np.testing.assert_allclose(np.sum(wn ** 2, axis=(1, 2, 3, 4)), 1.0, rtol=1e-6)
    # dw is the derivative of w_n w.r.t. s[1]
    h = 1e-6
    wp = L.modulate_weights(GOLD["mod_sw"], GOLD["mod_sb"], GOLD["mod_w"], s + np.array([0, h]))
    wm = L.modulate_weights(GOLD["mod_sw"], GOLD["mod_sb"], GOLD["mod_w"], s - np.array([0, h]))
    np.testing.assert_allclose((wp - wm) / (2 * h), dw, atol=1e-8)


# ---- whole network --------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def net8():
    seed_p, seed_x, mid, d0, d1, d2 = (int(v) for v in GOLD["net8_meta"])
    p = P.synthetic_params(seed=seed_p, mid_chan=mid)
    x = np.random.default_rng(seed_x).standard_normal((1, 3, d0, d1, d2)).astype(np.float32)
    d, v = M.forward(p, x, OM, DZ, VF)
    return p, x, d, v


def test_whole_net_golden(net8):
    _, _, d, v = net8
    np.testing.assert_allclose(d[0], GOLD["net8_disp"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(v[0], GOLD["net8_vel"], rtol=1e-9, atol=1e-9)


def test_param_tree_shapes():
    """tests/test_style_nbody_emulator_vel_core.py:408-419, tests/test_style_layers_vel.py:392-436."""
    p = P.synthetic_params(mid_chan=64)["params"]
    assert sorted(p) == sorted(M.RESNET_BLOCKS + M.RESAMPLE_BLOCKS) and len(p) == 15
    assert p["conv_l00"]["conv_0"]["weight"].shape == (64, 3, 3, 3, 3)
    assert p["conv_r2"]["conv_0"]["weight"].shape == (128, 128, 3, 3, 3)        # mid = max(in, out)
    assert p["conv_r2"]["conv_1"]["weight"].shape == (64, 128, 3, 3, 3)
    assert p["conv_r01"]["conv_1"]["weight"].shape == (3, 64, 3, 3, 3)
    assert p["down_l0"]["conv_0"]["weight"].shape == (64, 64, 2, 2, 2)
    assert p["conv_l1"]["skip"]["style_weight"].shape == (64, 2) and p["conv_l1"]["skip"]["style_bias"].shape == (64,)
    n = sum(a.size for b in p.values() for l in b.values() for a in l.values())
    assert n == 3354776


def test_velocity_properties(net8):
    """vel proportional to vel_fac, disp independent of it, vel_fac=0 => vel=0
    (tests/test_nbody_emulator_vel_core.py:190-221, :575-591); vel primal == non-vel model
    (tests/test_style_layers_vel.py:641-651)."""
    p, x, d, v = net8
    d2, v2 = M.forward(p, x, OM, DZ, 2 * VF)
    np.testing.assert_allclose(v2, 2 * v, rtol=1e-12)
    np.testing.assert_array_equal(d2, d)
    d0, v0 = M.forward(p, x, OM, DZ, 0.0)
    assert np.all(v0 == 0)
    dn = M.forward(p, x, OM, DZ, None, compute_vel=False)
    np.testing.assert_array_equal(dn, d)


def test_tangent_is_dDz_derivative(net8):
    """What the manual JVP encodes (style_layers_vel.py:64-101, core :190-193):
    vel / vel_fac == d disp / d Dz at fixed Om.  The reference only asserts corr > 0.9
    (tests/test_nbody_emulator_vel_core.py:679-710); here: float64 central differences."""
    p, x, d, v = net8
    h = 1e-6
    dp = M.forward(p, x, OM, DZ + h, None, compute_vel=False)
    dm = M.forward(p, x, OM, DZ - h, None, compute_vel=False)
    fd = (dp - dm) / (2 * h)
    err = np.linalg.norm(fd - v / VF) / np.linalg.norm(fd)
    assert err < 1e-5, err                     # LeakyReLU kinks crossed inside +-h bound the agreement
    assert np.corrcoef(fd.ravel(), v.ravel())[0, 1] > 0.999999


def test_style_equals_premodulated(net8):
    p, x, d, v = net8
    pp = P.premodulate_vel(p, Z, OM)
    assert "style_weight" not in pp["params"]["conv_l1"]["conv_0"]             # tests/test_nbody_emulator.py:699-706
    assert pp["params"]["conv_l1"]["conv_0"]["dweight"].shape == pp["params"]["conv_l1"]["conv_0"]["weight"].shape
    d2, v2 = M.forward(pp, x, None, DZ, VF, premodulated=True)
    np.testing.assert_allclose(d2, d, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(v2, v, rtol=1e-12, atol=1e-11)
    pn = P.premodulate(p, Z, OM)
    d3 = M.forward(pn, x, None, DZ, None, premodulated=True, compute_vel=False)
    np.testing.assert_allclose(d3, d, rtol=1e-12, atol=1e-13)


def test_float32_envelope(net8):
    """float32 evaluation of the same restatement: the rounding envelope the GPU tolerances are set against."""
    p, x, d, v = net8
    d32, v32 = M.forward(p, x, OM, DZ, VF, dtype=np.float32)
    e_d = np.linalg.norm(d32 - d) / np.linalg.norm(d)
    e_v = np.linalg.norm(v32 - v) / np.linalg.norm(v)
    assert e_d < 5e-6 and e_v < 1e-5, (e_d, e_v)


# ---- sub-box loop ------------------------------------------------------------------------------------------
def test_subbox_indices():
    """tests/test_subbox.py:86-95 (anchors), :121-134 (periodic wrap), :184-204 (paste covers once)."""
    size, ndiv = (256, 256, 256), (2, 2, 2)
    cs = S.crop_size(size, ndiv)
    assert cs == (128, 128, 128)
    assert S.get_anchor(0, ndiv, cs) == (0, 0, 0) and S.get_anchor(1, ndiv, cs) == (0, 0, 128)
    assert S.get_anchor(2, ndiv, cs) == (0, 128, 0) and S.get_anchor(7, ndiv, cs) == (128, 128, 128)
    crop, add = S.compute_indices(0, size, ndiv)
    assert crop[1].shape == (224, 1, 1) and crop[2].shape == (224, 1) and crop[3].shape == (224,)
    assert crop[3][0] == 208 and crop[3][47] == 255 and crop[3][48] == 0       # wraps to >= 208
    cover = np.zeros(size, np.int32)
    for idx in range(8):
        _, add = S.compute_indices(idx, size, ndiv)
        cover[add[1:]] += 1
    assert np.all(cover == 1)


def test_process_box_golden_and_ndiv_independence():
    seed_p, seed_x, mid, s0, s1, s2, n0, n1, n2 = (int(v) for v in GOLD["pbox_meta"])
    p = P.synthetic_params(seed=seed_p, mid_chan=mid)
    box = np.random.default_rng(seed_x).standard_normal((3, s0, s1, s2)).astype(np.float32)
    dis, vel = S.process_box(p, box, Z, OM, (s0, s1, s2), (n0, n1, n2))
    np.testing.assert_allclose(dis, GOLD["pbox_disp"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(vel, GOLD["pbox_vel"], rtol=1e-9, atol=1e-9)
    # crop_size % 8 == 0 => the result does not depend on ndiv
    dis1, vel1 = S.process_box(p, box, Z, OM, (s0, s1, s2), (1, 1, 1))
    np.testing.assert_allclose(dis1, dis, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(vel1, vel, rtol=1e-9, atol=1e-9)


def test_style_tangent_weight_factorises():
    """The identity behind the engine's two-product ("gauged") tangent, checked on the oracle in float64:
    dw[o,i,k] = w_n[o,i,k] * (alpha[i] + beta[o]) with alpha = s'/s, beta = -sum (w s)(w s') / norm^2
    (style_layers_vel.py:62-101), hence W.dx + dW.x = W.(dx + alpha*x) + beta*(W.x) for any x, dx."""
    from oracle import layers as L
    rng = np.random.default_rng(21)
    co, ci, k = 6, 5, 3
    w0 = rng.standard_normal((co, ci, k, k, k))
    sw = rng.standard_normal((ci, 2)) / np.sqrt(ci)
    sb = 1.0 + 0.1 * rng.standard_normal(ci)
    s = L.style_vector(0.31, 0.77)
    for first in (False, True):
        w_n, dw = L.modulate_weights_vel(sw, sb, w0, s, first)
        smod = sw @ s + sb
        alpha = sw[:, 1] / smod
        wmod = w0 * smod[None, :, None, None, None]
        norm2 = np.sum(wmod * wmod, axis=(1, 2, 3, 4)) + 1e-8
        beta = -np.sum(wmod * (w0 * sw[:, 1][None, :, None, None, None]), axis=(1, 2, 3, 4)) / norm2
        if first:
            beta = beta + 1.0 / (s[1] + 1.0)                 # the first layer's input scale is one more [o]-independent term
        fact = w_n * (alpha[None, :, None, None, None] + beta[:, None, None, None, None])
        assert np.max(np.abs(fact - dw)) <= 1e-12 * np.max(np.abs(dw))
        x, dx = rng.standard_normal((2, ci, k, k, k))
        lhs = np.einsum("oizyx,izyx->o", w_n, dx) + np.einsum("oizyx,izyx->o", dw, x)
        xg = dx + alpha[:, None, None, None] * x
        rhs = np.einsum("oizyx,izyx->o", w_n, xg) + beta * np.einsum("oizyx,izyx->o", w_n, x)
        assert np.max(np.abs(lhs - rhs)) <= 1e-12 * np.max(np.abs(lhs))


def test_torch_backend_and_branch_hook_of_the_oracle():
    """The oracle's second evaluator of the stride-1 convolutions (torch-CPU conv3d, used for the production-width fixtures,
    the kink-aware GPU checks and bench.py's cpu_baseline) is the NumPy tap-wise GEMM to float64 rounding; and the branch
    hook of tests/kink.py: handing back the oracle's own branches changes nothing, flipping one branch changes the velocity
    behind it and never the displacement (the primal follows the reference: layers_vel.py:184-185)."""
    from oracle import layers as L, model as M, params as P
    rng = np.random.default_rng(5)
    x = rng.standard_normal((16, 7, 9, 8))
    w = rng.standard_normal((8, 16, 3, 3, 3))
    with L.backend('torch'):
        yt = L.conv3(x, w)
        y1 = L.conv1(x, w[:, :, :1, :1, :1])
    np.testing.assert_allclose(yt, L.conv3(x, w), rtol=0, atol=1e-12)
    np.testing.assert_allclose(y1, L.conv1(x, w[:, :, :1, :1, :1]), rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        with L.backend('jax'):
            pass
    p = P.synthetic_params(seed=3, mid_chan=8)
    xin = rng.standard_normal((3, 104, 104, 104)).astype(np.float32)
    Om, Dz, vf = 0.3, 0.7731811501855036, 50.537651303131064
    d0, v0 = M.forward_single(p, xin, Om, Dz, vf, False, True)
    with L.backend('torch'):
        d1, v1 = M.forward_single(p, xin, Om, Dz, vf, False, True)
    assert np.abs(d1 - d0).max() <= 1e-12 and np.abs(v1 - v0).max() <= 1e-10
    seen = []
    d2, v2 = M.forward_single(p, xin, Om, Dz, vf, False, True, branch_hook=lambda name, pre: seen.append(name) or (pre > 0))
    assert np.array_equal(d2, d0) and np.array_equal(v2, v0)
    assert len(seen) == 23 and seen[0] == 'conv_l00/conv_0' and seen[-1] == 'conv_r01/conv_0' and 'conv_r01/conv_1' not in seen

    def flip(name, pre):
        b = pre > 0
        if name == 'conv_r01/conv_0':
            b = b.copy()
            b[3, 5, 5, 5] = ~b[3, 5, 5, 5]
        return b
    d3, v3 = M.forward_single(p, xin, Om, Dz, vf, False, True, branch_hook=flip)
    assert np.array_equal(d3, d0)
    changed = np.argwhere(np.abs(v3 - v0).sum(axis=0) > 0)
    # conv_r01/conv_0's voxel (5,5,5) feeds the 3^3 outputs around (4,4,4) of the 8^3 block (one VALID convolution later)
    assert len(changed) > 0 and changed.min() >= 3 and changed.max() <= 5
