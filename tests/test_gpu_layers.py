"""GPU parity, layer by layer: production kernels (through the C ABI test hook nbe_test_layer)
against the float64 oracle on the same seeded inputs.

Tolerances: the float32 MFMA is a k-ordered fmaf chain whose error against float64 is about
3.5e-7 * sum|a*b| at K = 4096 (MI355X guide); the tangent adds two such chains (K up to 2 x 3456).
Per layer output: relative L2 <= 5e-6 and max|delta|/RMS <= 1e-4 (measured: <= 1.5e-6 and <= 2.2e-5).

The float16 mode ("f16") is checked against the oracle run on the SAME float16-rounded operands
(activations, tangents, weights, residuals), so what remains is the float32 accumulation plus the
single float16 rounding of the stored result (half an ulp = 2^-11 relative per element):
relative L2 <= 5e-4, max|delta|/RMS <= 5e-3."""

import numpy as np
import pytest

from conftest import rel_l2, max_over_rms

pytestmark = pytest.mark.gpu

RTOL_L2 = 5e-6
RTOL_MAX = 1e-4


RTOL_L2_F16 = 5e-4
RTOL_MAX_F16 = 5e-3


def _chk(got, want, what, half=False):
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.all(np.isfinite(got)), what
    e2, em = rel_l2(got, want), max_over_rms(got, want)
    t2, tm = (RTOL_L2_F16, RTOL_MAX_F16) if half else (RTOL_L2, RTOL_MAX)
    assert e2 <= t2 and em <= tm, "%s: rel_l2=%.3e max/rms=%.3e" % (what, e2, em)


# the float16 model's Winograd-z form: U_xi and V = a +- b are rounded to float16 once more (measured: 4.6e-4 / 3.9e-3;
# the direct kernel on the same operands 2.1e-4)
RTOL_L2_F16W = 7e-4
RTOL_MAX_F16W = 7e-3


def _chk_f16w(got, want, what, pre=None, dpre=None):
    """pre / dpre: the oracle's pre-activation value and tangent.  The tangent of LeakyReLU jumps by a factor of 100 where
    the value changes sign, and this form's value carries ~3e-4 of rounding before the activation: voxels whose oracle
    pre-activation lies within 1e-2 RMS of zero may take either branch (each within the plain tolerance of that branch);
    every other voxel meets the plain tolerances."""
    assert got.shape == want.shape and np.all(np.isfinite(got)), what
    if pre is not None:
        rms = float(np.sqrt(np.mean(want.astype(np.float64) ** 2)))
        near = np.abs(pre) <= 1e-2 * float(np.sqrt(np.mean(pre ** 2)))
        either = np.minimum(np.abs(got - dpre), np.abs(got - 0.01 * dpre))
        assert float(either[near].max(initial=0.0)) <= RTOL_MAX_F16W * rms, "%s: a voxel at the kink on neither branch" % what
        got, want = got[~near], want[~near]
    e2, em = rel_l2(got, want), max_over_rms(got, want)
    assert e2 <= RTOL_L2_F16W and em <= RTOL_MAX_F16W, "%s: rel_l2=%.3e max/rms=%.3e" % (what, e2, em)
    return e2, em


def _h(a, half):
    """operand as the float16 engine sees it"""
    return a if (a is None or not half) else a.astype(np.float16).astype(np.float32)


def _rand(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


@pytest.fixture(scope="module", params=["f32", "f16x3", "f16"])
def eng(engine_factory, request):
    """Strict float32 MFMA and the float32-equivalent f16x3 split (operands carry 22 significant bits,
    float32 accumulation) must meet the same float32 tolerances; "f16" the float16 ones above."""
    return engine_factory(precision=request.param)


@pytest.mark.parametrize("cout,cin,k,first", [(64, 64, 3, False), (64, 3, 3, True), (3, 64, 1, False),
                                               (128, 128, 3, False), (64, 64, 2, False)])
def test_modulate(eng, cout, cin, k, first):
    from oracle import layers as L
    rng = np.random.default_rng(10 + cout + cin + k)
    w = _rand(rng, cout, cin, k, k, k) / np.sqrt(cin * k ** 3)
    sw = _rand(rng, cin, 2) / np.sqrt(cin)
    sb = (1 + 0.1 * _rand(rng, cin)).astype(np.float32)
    s = np.array([0.2, -0.226819], dtype=np.float64)
    wn, dw = eng.test_modulate(w, sw, sb, s, first)
    wn_o, dw_o = L.modulate_weights_vel(sw, sb, w, s, first)
    _chk(wn, wn_o, "w_n")
    _chk(dw, dw_o, "dw_tot")


CASES = [
    # kind, cin, cout, (D,H,W), has_dx, act, res, crop
    ("conv3", 64, 64, (9, 13, 21), True, True, False, 0),
    ("conv3", 64, 64, (7, 10, 40), True, True, True, 0),
    ("conv3", 3, 64, (10, 11, 19), False, True, False, 0),      # first layer: no input tangent
    ("conv3", 3, 64, (20, 40, 100), False, True, False, 0),     # ... more tiles than persistent workgroups (stem_h3_kernel)
    ("conv3", 3, 8, (6, 9, 35), False, False, False, 0),        # ... of a narrow test model, ragged tile edges
    ("conv3", 2, 16, (5, 7, 37), False, True, False, 0),
    ("conv3", 64, 3, (8, 9, 18), True, False, True, 0),         # head: Cout=3, residual, no act
    ("conv3", 128, 128, (6, 7, 20), True, True, False, 0),
    ("conv3", 128, 64, (6, 9, 15), True, True, True, 0),
    ("conv3", 8, 8, (12, 12, 12), True, True, True, 0),
    ("conv3", 16, 24, (9, 8, 33), True, False, False, 0),
    ("skip", 64, 64, (9, 12, 17), True, False, False, 2),
    ("skip", 3, 64, (10, 10, 22), False, False, False, 2),
    ("skip", 128, 64, (8, 8, 16), True, False, False, 2),
    ("skip", 64, 3, (8, 9, 18), True, False, False, 2),
    ("skip", 8, 8, (9, 9, 9), True, False, False, 2),
    ("down", 64, 64, (8, 12, 20), True, True, False, 0),
    ("down", 8, 8, (16, 8, 12), True, True, False, 0),
    ("up", 64, 64, (5, 6, 9), True, True, False, 0),
    ("up", 8, 8, (6, 6, 6), True, True, False, 0),
]


@pytest.mark.parametrize("kind,cin,cout,dims,has_dx,act,res,crop", CASES)
def test_layer_vel(eng, kind, cin, cout, dims, has_dx, act, res, crop):
    from oracle import layers as L
    rng = np.random.default_rng(1000 + cin * 7 + cout * 13 + dims[0] * 31 + dims[1] * 17 + dims[2])
    k = {"conv3": 3, "skip": 1, "down": 2, "up": 2}[kind]
    x = _rand(rng, cin, *dims)
    dx = _rand(rng, cin, *dims) if has_dx else None
    w = _rand(rng, cout, cin, k, k, k) / np.sqrt(cin * k ** 3)
    dw = _rand(rng, cout, cin, k, k, k) / np.sqrt(cin * k ** 3)
    b = 0.1 * _rand(rng, cout)
    half = eng.precision == "f16"
    x64 = _h(x, half).astype(np.float64)
    dx64 = None if dx is None else _h(dx, half).astype(np.float64)
    y_o, dy_o = L.conv_layer_vel(kind, x64, dx64, _h(w, half).astype(np.float64), _h(dw, half).astype(np.float64),
                                 b.astype(np.float64))
    if kind == "skip" and crop:
        y_o, dy_o = y_o[:, crop:-crop, crop:-crop, crop:-crop], dy_o[:, crop:-crop, crop:-crop, crop:-crop]
    r = dr = None
    if res:
        r, dr = _rand(rng, *y_o.shape), _rand(rng, *y_o.shape)
        y_o, dy_o = y_o + _h(r, half), dy_o + _h(dr, half)
    if act:
        y_o, dy_o = L.leaky_relu_vel(y_o, dy_o)
    y, dy = eng.test_layer(kind, x, w, b, dx=dx, dw=dw, crop=crop, act=act, res=r, dres=dr)
    _chk(y, y_o, "%s primal" % kind, half)
    _chk(dy, dy_o, "%s tangent" % kind, half)


@pytest.mark.parametrize("kind,cin,cout,dims", [("conv3", 64, 64, (8, 9, 23)), ("skip", 64, 64, (8, 9, 12)),
                                                 ("down", 64, 64, (8, 8, 12)), ("up", 64, 64, (4, 5, 7)),
                                                 ("conv3", 3, 64, (9, 9, 17)), ("conv3", 64, 3, (9, 9, 17))])
def test_layer_disp_only(eng, kind, cin, cout, dims):
    """displacement-only twins (style_layers.py:86-99): same kernels, primal accumulators only."""
    from oracle import layers as L
    rng = np.random.default_rng(77 + cin + cout)
    k = {"conv3": 3, "skip": 1, "down": 2, "up": 2}[kind]
    x = _rand(rng, cin, *dims)
    w = _rand(rng, cout, cin, k, k, k) / np.sqrt(cin * k ** 3)
    b = 0.1 * _rand(rng, cout)
    half = eng.precision == "f16"
    y_o = L.leaky_relu(L.conv_layer(kind, _h(x, half).astype(np.float64), _h(w, half).astype(np.float64),
                                    b.astype(np.float64)))
    y = eng.test_layer(kind, x, w, b, act=True)
    _chk(y, y_o, "%s disp-only" % kind, half)


def test_leaky_relu_pins(eng):
    """tests/test_layers_vel.py:268-334 pins: [-2,-1,0,1,2] -> [-0.02,-0.01,0,1,2]; tangent at x == 0
    takes the slope branch.  Realised through a 1x1x1 identity layer + activation epilogue."""
    cin = cout = 8
    vals = np.array([-2, -1, 0, 1, 2], np.float32)
    x = np.zeros((cin, 1, 1, 8), np.float32)
    x[0, 0, 0, :5] = vals
    dx = np.ones_like(x)
    w = np.eye(cout, cin, dtype=np.float32).reshape(cout, cin, 1, 1, 1)
    dw = np.zeros_like(w)
    y, dy = eng.test_layer("skip", x, w, np.zeros(cout, np.float32), dx=dx, dw=dw, act=True)
    rtol = 2.0 ** -11 if eng.precision == "f16" else 1e-6          # stored as float16: half an ulp
    np.testing.assert_allclose(y[0, 0, 0, :5], [-0.02, -0.01, 0.0, 1.0, 2.0], rtol=rtol, atol=0)
    np.testing.assert_allclose(dy[0, 0, 0, :5], [0.01, 0.01, 0.01, 1.0, 1.0], rtol=rtol, atol=0)


@pytest.mark.parametrize("prec", ["f16x3", "f16"])
def test_dma_addressing_with_bit31_of_the_address_set(engine_factory, prec, monkeypatch):
    """The 16x16x32 kernels feed their global -> LDS DMA with a wave-uniform 64-bit base that is split into 32-bit halves
    (pinned in SGPRs with readfirstlane) and joined again, plus a 32-bit per-lane offset (nbe_kernels_h3.hip: dma16s).
    A join that sign-extends the LOW half is wrong exactly when bit 31 of the address is set -- the intermittent memory
    access fault of round 1 (address 0xffffbf6e4000 = an address ending in 0xbf6e4000 with the upper word all ones; it
    came and went with where the allocator put the arena).  NBE_TEST_ADDR_BIT31 places the test tensors on either side
    of that bit: same results, bit for bit, and within the layer tolerance of the oracle."""
    from oracle import layers as L
    half = prec == "f16"
    e = engine_factory(precision=prec)
    rng = np.random.default_rng(99)
    cin, cout = 32, 64
    x, dx = _rand(rng, cin, 6, 20, 40), _rand(rng, cin, 6, 20, 40)
    w = _rand(rng, cout, cin, 3, 3, 3) / np.sqrt(cin * 27)
    dw = _rand(rng, cout, cin, 3, 3, 3) / np.sqrt(cin * 27)
    b = (0.05 * _rand(rng, cout)).astype(np.float32)
    out = {}
    for bit in ("0", "1"):
        monkeypatch.setenv("NBE_TEST_ADDR_BIT31", bit)
        out[bit] = e.test_layer("conv3", x, w, b, dx=dx, dw=dw, act=True)
    monkeypatch.delenv("NBE_TEST_ADDR_BIT31")
    assert np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1])
    y, dy = L.conv_layer_vel("conv3", _h(x, half).astype(np.float64), _h(dx, half).astype(np.float64),
                             _h(w, half).astype(np.float64), _h(dw, half).astype(np.float64), b.astype(np.float64))
    y, dy = L.leaky_relu_vel(y, dy)
    _chk(out["1"][0], y, "conv3 bit31 y", half)
    _chk(out["1"][1], dy, "conv3 bit31 dy", half)


GAUGED = [
    # cin, cout, (D, H, W), act       (Dv = D - 2: even -> the Winograd-z kernel on f16x3, odd -> the direct gauged kernel)
    (64, 64, (10, 13, 21), True),
    (64, 64, (6, 10, 40), False),
    (128, 64, (8, 12, 37), True),        # conv_r00/conv_0: eight chunks, 32 stages
    (64, 128, (4, 9, 18), True),         # two cout tiles
    (16, 24, (12, 8, 33), True),         # one chunk, ragged cout
    (8, 8, (14, 12, 12), False),         # padded chunk of a narrow test model
    (64, 64, (9, 11, 19), True),         # odd number of output planes: no Winograd form
    (32, 64, (26, 20, 70), True),        # more plane pairs and patches than one wave of workgroups per XCD
]


@pytest.mark.parametrize("cin,cout,dims,act", GAUGED)
def test_layer_gauged(eng, cin, cout, dims, act, monkeypatch):
    """The two-product form of a style-modulated 3x3x3 layer: y = W.x + b, dy = W.dx~ + beta (.) (W.x),
    through nbe_test_layer_gauged -- conv_h3w_kernel (Winograd F(2,3) along z, one accumulator per output) where it
    applies, conv_h3g_kernel otherwise and with NBE_WINO=0; the float32 and float16 engines run their own gauged kernels.
    Weights are unit vectors per output channel, as the modulation leaves them (style_layers_vel.py:84-96)."""
    from oracle import layers as L
    rng = np.random.default_rng(4000 + cin * 3 + cout * 5 + dims[0] * 7 + dims[2])
    x, dx = _rand(rng, cin, *dims), _rand(rng, cin, *dims)
    w = _rand(rng, cout, cin, 3, 3, 3)
    w /= np.sqrt((w.astype(np.float64) ** 2).sum(axis=(1, 2, 3, 4), keepdims=True)).astype(np.float32)
    beta = (0.3 * _rand(rng, cout)).astype(np.float32)
    b = 0.1 * _rand(rng, cout)
    half = eng.precision == "f16"
    w64 = _h(w, half).astype(np.float64)
    y_o, dy_o = L.conv_layer_vel("conv3", _h(x, half).astype(np.float64), _h(dx, half).astype(np.float64), w64,
                                 w64 * beta.astype(np.float64)[:, None, None, None, None], b.astype(np.float64))
    y_pre, dy_pre = y_o, dy_o
    if act:
        y_o, dy_o = L.leaky_relu_vel(y_o, dy_o)
    y, dy = eng.test_layer_gauged(x, dx, w, beta, b, act=act)
    if half and cin % 32 == 0 and dims[0] % 2 == 0:
        # the float16 model's Winograd-z form (conv_h3w_kernel<., ., F16>): the transformed weights U_xi and the transformed
        # planes V = a +- b are rounded to float16 once more (half an ulp each on top of the operands' own), and NBE_WINO=0
        # runs conv_h2q_kernel on the same operands
        ey = _chk_f16w(y, y_o, "gauged primal, float16 Winograd-z")
        ed = _chk_f16w(dy, dy_o, "gauged tangent, float16 Winograd-z", *((y_pre, dy_pre) if act else ()))
        monkeypatch.setenv("NBE_WINO", "0")
        y0, dy0 = eng.test_layer_gauged(x, dx, w, beta, b, act=act)
        monkeypatch.delenv("NBE_WINO")
        print("f16 wino vs f64, rel-L2 (max/rms): y %.2e (%.2e) dy away from the kink %.2e (%.2e) | direct: %.2e %.2e" % (
            ey + ed + (rel_l2(y0, y_o), rel_l2(dy0, dy_o))))
        y, dy = y0, dy0
    _chk(y, y_o, "gauged primal", half)
    _chk(dy, dy_o, "gauged tangent", half)
    if eng.precision == "f16x3":
        monkeypatch.setenv("NBE_WINO", "0")
        y0, dy0 = eng.test_layer_gauged(x, dx, w, beta, b, act=act)
        _chk(y0, y_o, "gauged primal, direct kernel", half)
        _chk(dy0, dy_o, "gauged tangent, direct kernel", half)
        print("wino vs direct: y %.2e dy %.2e | vs f64: wino %.2e %.2e direct %.2e %.2e" % (
            rel_l2(y, y0), rel_l2(dy, dy0), rel_l2(y, y_o), rel_l2(dy, dy_o), rel_l2(y0, y_o), rel_l2(dy0, dy_o)))


@pytest.mark.parametrize("cin,cout,dims,act", [(64, 64, (8, 13, 21), True), (128, 64, (6, 12, 37), False), (64, 3, (10, 9, 40), False)])
def test_layer_gauged_with_residual_float16(engine_factory, cin, cout, dims, act, monkeypatch):
    """conv_1 of the float16 model: the gauged layer plus the block's skip as a residual added before the activation.  The
    Winograd-z form adds it in its epilogue (the only conv_h3w_kernel variant with F_RES); NBE_WINO=0 runs conv_h2q_kernel."""
    from oracle import layers as L
    e = engine_factory(precision="f16")
    rng = np.random.default_rng(4100 + cin + cout + dims[2])
    x, dx = _rand(rng, cin, *dims), _rand(rng, cin, *dims)
    w = _rand(rng, cout, cin, 3, 3, 3)
    w /= np.sqrt((w.astype(np.float64) ** 2).sum(axis=(1, 2, 3, 4), keepdims=True)).astype(np.float32)
    beta, b = (0.3 * _rand(rng, cout)).astype(np.float32), 0.1 * _rand(rng, cout)
    w64 = _h(w, True).astype(np.float64)
    y_o, dy_o = L.conv_layer_vel("conv3", _h(x, True).astype(np.float64), _h(dx, True).astype(np.float64), w64,
                                 w64 * beta.astype(np.float64)[:, None, None, None, None], b.astype(np.float64))
    r, dr = _rand(rng, *y_o.shape), _rand(rng, *y_o.shape)
    y_o, dy_o = y_o + _h(r, True), dy_o + _h(dr, True)
    y_pre, dy_pre = y_o, dy_o
    if act:
        y_o, dy_o = L.leaky_relu_vel(y_o, dy_o)
    e.profile_reset(); e.profile_enable(True)
    y, dy = e.test_layer_gauged(x, dx, w, beta, b, act=act, res=r, dres=dr)
    e.profile_enable(False)
    assert any(k["kernel"].startswith("conv_h1w") for k in e.profile_read()), e.profile_read()
    ey = _chk_f16w(y, y_o, "primal")
    ed = _chk_f16w(dy, dy_o, "tangent", *((y_pre, dy_pre) if act else ()))
    monkeypatch.setenv("NBE_WINO", "0")
    y0, dy0 = e.test_layer_gauged(x, dx, w, beta, b, act=act, res=r, dres=dr)
    _chk(y0, y_o, "primal, direct kernel", True)
    _chk(dy0, dy_o, "tangent, direct kernel", True)
    print("f16 wino+res vs f64, rel-L2 (max/rms): y %.2e (%.2e) dy away from the kink %.2e (%.2e) | direct: %.2e %.2e" % (
        ey + ed + (rel_l2(y0, y_o), rel_l2(dy0, dy_o))))


def test_winograd_pack_refuses_weights_beyond_its_scale(engine_factory):
    """conv_h3w_kernel scales the weights by 2^14 (unit rows fit with two binades to spare); a weight that would leave
    the f16 range there clears the context's Winograd flag and the launch runs on the direct kernel: same tolerances."""
    from oracle import layers as L
    e = engine_factory(precision="f16x3")
    rng = np.random.default_rng(5)
    cin, cout, dims = 32, 64, (6, 9, 20)
    x, dx = _rand(rng, cin, *dims), _rand(rng, cin, *dims)
    w = _rand(rng, cout, cin, 3, 3, 3) / np.sqrt(cin * 27.0).astype(np.float32)
    w[3, 5, 1, 1, 1] = 9.0
    beta, b = (0.3 * _rand(rng, cout)).astype(np.float32), 0.1 * _rand(rng, cout)
    w64 = w.astype(np.float64)
    y_o, dy_o = L.conv_layer_vel("conv3", x.astype(np.float64), dx.astype(np.float64), w64,
                                 w64 * beta.astype(np.float64)[:, None, None, None, None], b.astype(np.float64))
    y, dy = e.test_layer_gauged(x, dx, w, beta, b)
    _chk(y, y_o, "primal")
    _chk(dy, dy_o, "tangent")
