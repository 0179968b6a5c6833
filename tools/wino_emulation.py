"""NumPy emulation of conv_h3w_kernel's arithmetic (csrc/nbe_kernels_wino.h) against an exact float64 convolution:
Winograd F(2,3) along z with the two-phase K order (xi = 1 -> A, xi = 2 -> B; butterfly; xi = 0 -> A, xi = 3 -> B with U3
negated), weights scaled by 2^14 and split into f16 hi + UNSCALED f16 lo, hi * 2^-11 as the weight operand of the lo(x)
product, activations transformed in packed f16 on their (hi, lo) parts (round 3: s = a hi +- b hi, TwoSum's exact rounding
error of s into the lo part; `transform='f32'` is the round-2 form: joined, transformed and re-split in float32), one
float32 accumulator per output rounded after every 16-channel tap product.  Beside it the direct f16x3 form of conv_h3g_kernel (separate main / correction accumulators).

  python tools/wino_emulation.py            # 64 -> 64 channels, prints relative L2 errors at a few input scales
"""
import numpy as np

f16 = lambda a: a.astype(np.float16).astype(np.float64)
f32 = lambda a: a.astype(np.float32).astype(np.float64)


def split_scaled(a):
    hi = f16(a)
    return hi, f16((a - hi) * 2048.0)


def conv_exact(x, w):
    cout, cin = w.shape[:2]
    D, H, W = x.shape[1:]
    y = np.zeros((cout, D - 2, H - 2, W - 2))
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                y += np.einsum('oi,izyx->ozyx', w[:, :, dz, dy, dx], x[:, dz:dz + D - 2, dy:dy + H - 2, dx:dx + W - 2])
    return y


def conv_f16x3_direct(x, w):
    cout, cin = w.shape[:2]
    D, H, W = x.shape[1:]
    Do, Ho, Wo = D - 2, H - 2, W - 2
    xh, xl = split_scaled(x)
    wh, wl = split_scaled(w)
    ym = np.zeros((cout, Do, Ho, Wo))
    yc = np.zeros_like(ym)
    for c0 in range(0, cin, 16):
        for dz in range(3):
            for dy in range(3):
                for dx in range(3):
                    sl = (slice(c0, c0 + 16), slice(dz, dz + Do), slice(dy, dy + Ho), slice(dx, dx + Wo))
                    Wh, Wl = wh[:, c0:c0 + 16, dz, dy, dx], wl[:, c0:c0 + 16, dz, dy, dx]
                    ym = f32(ym + np.einsum('oi,izyx->ozyx', Wh, xh[sl]))
                    yc = f32(yc + np.einsum('oi,izyx->ozyx', Wh, xl[sl]))
                    yc = f32(yc + np.einsum('oi,izyx->ozyx', Wl, xh[sl]))
    return f32(ym + yc / 2048.0)


def transform_f16(ah, al, bh, bl, sb, unscaled=False):
    """V = a + sb b on (hi, lo * 2^11) parts in f16 arithmetic, as xf_step does it: returns (s, lo).
    unscaled (NBE_WINO_LOU, the default build): lo = err + (a lo + sb b lo) 2^-11, NOT times 2^11 -- subnormal below 2^-14."""
    s = f16(ah + sb * bh)
    bb = f16(s - ah)
    err = f16(f16(ah - f16(s - bb)) + f16(sb * bh - bb))          # exact (TwoSum)
    if unscaled:
        return s, f16(f16(al + sb * bl) / 2048.0 + err)           # fma: one rounding (gradual underflow)
    lo = f16(err * 2048.0 + f16(al + sb * bl))                    # fma: one rounding
    return s, lo


def conv_winograd_z(x, w, S=2.0 ** 14, transform='f16', unscaled_lo=False):
    """unscaled_lo: the default build's form -- the lo part of V is kept unscaled and meets the weights' hi part as it is."""
    cout, cin = w.shape[:2]
    D, H, W = x.shape[1:]
    Do, Ho, Wo = D - 2, H - 2, W - 2
    assert Do % 2 == 0
    U = np.stack([w[:, :, 0], (w[:, :, 0] + w[:, :, 1] + w[:, :, 2]) / 2, (w[:, :, 0] - w[:, :, 1] + w[:, :, 2]) / 2,
                  -w[:, :, 2]], axis=0) * S                       # [xi][o][i][dy][dx], pack_h3w_kernel
    Uh = f16(U)
    Ul = f16(U - Uh)                                              # unscaled remainder
    Uhp = f16(Uh / 2048.0)                                        # v_pk_mul_f16 by 2^-11
    xh, xl = split_scaled(x)                                      # the stored (hi, lo) planes
    xj = f32(xh + xl / 2048.0)                                    # ... joined in float32 (round-2 form)
    PA, PB, SB = (0, 1, 2, 1), (2, 2, 1, 3), (-1.0, 1.0, -1.0, -1.0)   # launch_h3w: V_xi = d[PA] + SB d[PB]
    y = np.zeros((cout, Do, Ho, Wo))
    for p in range(Do // 2):
        d = [xj[:, 2 * p + k] for k in range(4)]
        V = [f32(d[0] - d[2]), f32(d[1] + d[2]), f32(d[2] - d[1]), f32(d[1] - d[3])]

        def run(acc, xi):
            if transform == 'f16':
                Vh, Vl = transform_f16(xh[:, 2 * p + PA[xi]], xl[:, 2 * p + PA[xi]], xh[:, 2 * p + PB[xi]], xl[:, 2 * p + PB[xi]], SB[xi],
                                       unscaled=unscaled_lo)
            else:
                Vh, Vl = split_scaled(V[xi])
            Wlo = Uh if unscaled_lo else Uhp                      # what multiplies the lo part of V
            for c0 in range(0, cin, 16):
                for dy in range(3):
                    for dx in range(3):
                        sl = (slice(c0, c0 + 16), slice(dy, dy + Ho), slice(dx, dx + Wo))
                        acc = f32(acc + np.einsum('oi,iyx->oyx', Wlo[xi][:, c0:c0 + 16, dy, dx], Vl[sl]))
                        acc = f32(acc + np.einsum('oi,iyx->oyx', Uh[xi][:, c0:c0 + 16, dy, dx], Vh[sl]))
                        acc = f32(acc + np.einsum('oi,iyx->oyx', Ul[xi][:, c0:c0 + 16, dy, dx], Vh[sl]))
            return acc
        A = run(np.zeros((cout, Ho, Wo)), 1)
        B = run(np.zeros((cout, Ho, Wo)), 2)
        A, B = f32(A + B), f32(A - B)
        A, B = run(A, 0), run(B, 3)
        y[:, 2 * p], y[:, 2 * p + 1] = f32(A / S), f32(B / S)
    return y


def conv_winograd_z_f16(x, w, S=256.0):
    """The float16 model's form (conv_h3w_kernel<., ., F16>): operands are plain f16 numbers (x as stored, w as given), the
    transformed weights U_xi * 2^8 and the transformed planes V = a +- b are each rounded to f16 once more, float32 accumulation
    per 32-channel tap-pair product, same two-phase K order.  Returns float32-rounded outputs BEFORE the store's f16 rounding."""
    cout, cin = w.shape[:2]
    D, H, W = x.shape[1:]
    Do, Ho, Wo = D - 2, H - 2, W - 2
    assert Do % 2 == 0 and cin % 32 == 0
    U = f16(np.stack([w[:, :, 0], (w[:, :, 0] + w[:, :, 1] + w[:, :, 2]) / 2, (w[:, :, 0] - w[:, :, 1] + w[:, :, 2]) / 2,
                      -w[:, :, 2]], axis=0) * S)
    PA, PB, SB = (0, 1, 2, 1), (2, 2, 1, 3), (-1.0, 1.0, -1.0, -1.0)
    y = np.zeros((cout, Do, Ho, Wo))
    for p in range(Do // 2):
        def run(acc, xi):
            V = f16(x[:, 2 * p + PA[xi]] + SB[xi] * x[:, 2 * p + PB[xi]])
            for c0 in range(0, cin, 32):
                for dy in range(3):
                    for dx in range(3):
                        sl = (slice(c0, c0 + 32), slice(dy, dy + Ho), slice(dx, dx + Wo))
                        acc = f32(acc + np.einsum('oi,iyx->oyx', U[xi][:, c0:c0 + 32, dy, dx], V[sl]))
            return acc
        A = run(np.zeros((cout, Ho, Wo)), 1)
        B = run(np.zeros((cout, Ho, Wo)), 2)
        A, B = f32(A + B), f32(A - B)
        A, B = run(A, 0), run(B, 3)
        y[:, 2 * p], y[:, 2 * p + 1] = f32(A / S), f32(B / S)
    return y


def unit_rows(w):
    return w / np.sqrt((w ** 2).sum(axis=(1, 2, 3, 4), keepdims=True))


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    cin = cout = 64
    x = f32(rng.standard_normal((cin, 10, 12, 12)))
    w = f32(unit_rows(rng.standard_normal((cout, cin, 3, 3, 3))))
    ye = conv_exact(x, w)
    print("direct f16x3      rel-L2 %.2e" % rel(conv_f16x3_direct(x, w), ye))
    print("Winograd-z merged rel-L2 %.2e (packed-f16 TwoSum transform), %.2e (float32 transform, round 2)"
          % (rel(conv_winograd_z(x, w), ye), rel(conv_winograd_z(x, w, transform='f32'), ye)))
    for s in (1e-3, 30.0, 1e3):
        xs = f32(x * s)
        print("  input scale %g: %.2e" % (s, rel(conv_winograd_z(xs, w), conv_exact(xs, w))))
    x16, w16 = f16(x), f16(w)                                      # the float16 model: operands as the engine stores them
    y16 = conv_exact(x16, w16)
    print("float16 model on the same f16 operands, after the store's f16 rounding: direct %.2e, Winograd-z form %.2e"
          % (rel(f16(y16), y16), rel(f16(conv_winograd_z_f16(x16, w16)), y16)))
    print("unscaled lo part of V (default build), activation RMS as the range shift leaves it:")
    for s in (2.0 ** 6, 8.0, 1.0, 2.0 ** -3, 2.0 ** -6):
        xs = f32(x * s)
        print("  activation RMS %-8g rel-L2 %.2e" % (s, rel(conv_winograd_z(xs, w, unscaled_lo=True), conv_exact(xs, w))))
