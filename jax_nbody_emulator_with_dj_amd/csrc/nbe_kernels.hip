// gfx950 (MI355X, CDNA4) kernels of the N-body emulator forward.
//
// What they replace in the reference (all of it XLA-generated code there):
//   conv_mfma_kernel   jax.lax.conv_general_dilated call sites
//                      style_layers_vel.py:109-141 (k=3, k=1, k=2 s=2) and :236-269 (lhs_dilation
//                      up-sample), plus the bias add, LeakyReLUVel (layers_vel.py:182-186), the ResNet
//                      residual add (style_blocks_vel.py:158-164) as epilogues
//   modulate_kernel    weight modulation / demodulation / d/dDz (style_layers_vel.py:62-105,
//                      nbody_emulator.py:189-219)
//   gather/head        subbox.py:197-215 crop + paste, core :132-139 (input scale, x0) and :187-193
//
// Design: activations are stored as C/4 planes of float4 per voxel.  A 3x3x3 VALID
// convolution over a dense (D,H,W) volume is a sum of 27 "flat shifts": output flat position q reads
// input flat positions q + dz*H*W + dy*W + dx.  The kernel sweeps 256 consecutive flat positions per
// workgroup (positions whose x/y run past the valid extent are computed and discarded, ~2/W+2/H waste),
// stages one (dz,dy) row segment of 8 channels at a time into LDS with direct global->LDS DMA, serves the
// three dx taps from the same segment by shifting the LDS read address, and contracts with fp32 MFMA
// (v_mfma_f32_32x32x2_f32: A = weights [cout x k], B = activations [k x voxel]) so that each lane ends up
// with 4 consecutive output channels of one voxel = one float4 store into the output plane.
// The tangent (velocity) path shares every staged operand: y += W.X, dy += dW.X + W.dX.

#include "nbe_kernels_internal.h"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace nbe {

#define LDS_AS NBE_LDS_AS
#define GLB_AS NBE_GLB_AS

extern __shared__ __attribute__((aligned(16))) f32x4 lds_dyn[];

// G6 (3x3x3 with input tangent only): the input tangent arrives in this layer's gauge, dy = W.dx~ + beta * (W.x) --
// no dW operand and two products per tap instead of three (see conv_h3g_kernel in nbe_kernels_h3.hip).
template <int MODE, bool VEL, bool HAS_DX, int NI, bool G6 = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvKArgs a) {
    constexpr int CK = mode_ck(MODE), GL = CK / 4, TAPS = mode_taps(MODE);
    constexpr int COUT_T = 32 * NI;
    constexpr int XV = (MODE == MODE_FLAT3) ? 320 : 256;
    constexpr int WP = TAPS * GL * COUT_T;           // 16-byte pieces of one weight stage
    constexpr int XP = GL * XV;                      // 16-byte pieces of one activation stage
    constexpr bool DX = VEL && HAS_DX;
    constexpr bool DWT = VEL && !G6;                 // a tangent weight operand exists
    static_assert(!G6 || (DX && MODE == MODE_FLAT3), "gauged form: 3x3x3 layers with an input tangent");
    constexpr int OFF_W = 0, OFF_DW = WP, OFF_X = OFF_DW + (DWT ? WP : 0), OFF_DXX = OFF_X + XP;
    constexpr int BUF = OFF_DXX + (DX ? XP : 0);
    constexpr int NIW = WP / 64, NIX = XP / 64;
    constexpr int NW_TOT = NIW * (DWT ? 2 : 1);
    constexpr int NINSTR = NW_TOT + NIX * (DX ? 2 : 1);
    static_assert(WP % 64 == 0 && XP % 64 == 0, "stage arrays must be whole wave-instructions");

    f32x4* lds = lds_dyn;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    const int tile = xcd_tile(blockIdx.x, a.ntiles);
    const int ct = blockIdx.y;
    const long q0 = (long)tile * TILE_VOX;
    const int nchunk = a.nchunk;
    const int nstage = mode_nseg(MODE) * nchunk;
    const long HW = (long)a.H * a.W;

    int* inbase = (int*)(lds + 2 * BUF);             // DOWN mode: input voxel of each tile slot
    if (MODE == MODE_DOWN) {
        long o = q0 + tid;
        if (o > a.Q - 1) o = a.Q - 1;
        const int hw = a.Hv * a.Wv;
        const int zo = (int)(o / hw), rem = (int)(o - (long)zo * hw);
        const int yo = rem / a.Wv, xo = rem - yo * a.Wv;
        inbase[tid] = (int)((2L * zo * a.H + 2 * yo) * a.W + 2 * xo);
        __syncthreads();
    }

    auto issue = [&](int s, int b) {
        // channel chunk outer, (dz,dy) segment inner: the 9 segments of one 8-channel slab are fetched back to
        // back, so the rows shared between segments and between neighbouring tiles are still in the XCD's L2
        const int chunk = s / mode_nseg(MODE), seg = s - chunk * mode_nseg(MODE);
        long segoff;
        if (MODE == MODE_FLAT3) segoff = (seg / 3) * HW + (seg % 3) * a.W;
        else if (MODE == MODE_DOWN) segoff = (seg >> 2) * HW + ((seg >> 1) & 1) * a.W + (seg & 1);
        else segoff = 0;
        const long wbase = ((long)(ct * nstage + s) * WP) * 4;
        f32x4* buf = lds + b * BUF;
#pragma unroll
        for (int t = 0; t < (NINSTR + 3) / 4; ++t) {
            const int n = wave + 4 * t;              // wave-uniform instruction slot
            if (n < NIW) {
                dma16(a.w + wbase + (long)(n * 64 + lane) * 4, buf + OFF_W + n * 64);
            } else if (DWT && n < NW_TOT) {
                const int m = n - NIW;
                dma16(a.dw + wbase + (long)(m * 64 + lane) * 4, buf + OFF_DW + m * 64);
            } else if (n < NINSTR) {
                const bool tang = DX && n >= NW_TOT + NIX;
                const int m = n - NW_TOT - (tang ? NIX : 0);
                const int gl = (m * 64) / XV;
                const int vl = (m * 64) % XV + lane;
                long v;
                if (MODE == MODE_DOWN) v = (long)inbase[vl] + segoff;
                else {
                    v = q0 + a.in_off + segoff + vl;
                    if (v > a.P - 1) v = a.P - 1;
                }
                const long off = ((long)(chunk * GL + gl) * a.in_pstride + v) * 4;
                dma16((tang ? a.dx : a.x) + off, buf + (tang ? OFF_DXX : OFF_X) + m * 64);
            }
        }
    };

    f32x16 accy[NI][2];
    f32x16 accd[NI][2];
#pragma unroll
    for (int it = 0; it < NI; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accy[it][jt][e] = 0.f; accd[it][jt][e] = 0.f; }

    auto compute = [&](int b) {
        const f32x4* buf = lds + b * BUF;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
            for (int kg = 0; kg < CK / 8; ++kg) {
                f32x4 wv[NI], dwv[NI], xv[2], dxv[2];
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    const int o = (tap * GL + 2 * kg + lh) * COUT_T + 32 * it + li;
                    wv[it] = buf[OFF_W + o];
                    if (DWT) dwv[it] = buf[OFF_DW + o];
                }
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    const int o = (2 * kg + lh) * XV + wave * 64 + 32 * jt + li + (MODE == MODE_FLAT3 ? tap : 0);
                    xv[jt] = buf[OFF_X + o];
                    if (DX) dxv[jt] = buf[OFF_DXX + o];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int it = 0; it < NI; ++it) {
#pragma unroll
                        for (int jt = 0; jt < 2; ++jt) {
                            accy[it][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[it][r], xv[jt][r], accy[it][jt], 0, 0, 0);
                            if (DWT) accd[it][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dwv[it][r], xv[jt][r], accd[it][jt], 0, 0, 0);
                            if (DX) accd[it][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[it][r], dxv[jt][r], accd[it][jt], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };

    issue(0, 0);
    __syncthreads();                                  // hipcc drains the DMA (vmcnt(0)) at the barrier
    for (int s = 0; s < nstage; ++s) {
        if (s + 1 < nstage) issue(s + 1, (s + 1) & 1);
        compute(s & 1);
        __syncthreads();
    }

    // ---- epilogue: bias, residual, LeakyReLU (+tangent), float4 stores ----------------------------
    const bool act = a.flags & F_ACT, res = a.flags & F_RES;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const long q = q0 + wave * 64 + 32 * jt + li;
        bool valid = q < a.Q;
        long o;
        if (MODE == MODE_DOWN) {
            o = q;
        } else {
            const int z = (int)(q / HW), rem = (int)(q - (long)z * HW);
            const int yy = rem / a.W, xx = rem - yy * a.W;
            valid = valid && xx < a.Wv && yy < a.Hv && z < a.Dv;
            o = ((long)(z * a.osz + a.oz) * a.Ho + (yy * a.osz + a.oy)) * a.Wo + (xx * a.osz + a.ox);
        }
        if (!valid) continue;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cgl = 8 * it + 2 * k + lh;                  // cout group inside the tile
                const int cg = ct * (COUT_T / 4) + cgl;
                if (cg >= a.cout_groups) continue;
                const f32x4 bv = *(const f32x4*)(a.bias + cg * 4);
                f32x4 v, dv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = accy[it][jt][4 * k + e] + bv[e]; dv[e] = accd[it][jt][4 * k + e]; }
                if (G6) {
                    const f32x4 be = *(const f32x4*)(a.beta + cg * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[e] += be[e] * accy[it][jt][4 * k + e];
                }
                if (res) {
                    const long ro = ((long)cg * a.res_pstride + o) * 4;
                    const f32x4 rv = *(const f32x4*)(a.r + ro);
                    v += rv;
                    if (VEL) { const f32x4 drv = *(const f32x4*)(a.dr + ro); dv += drv; }
                }
                if (act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (VEL) dv[e] = v[e] > 0.f ? dv[e] : 0.01f * dv[e];
                        v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
                    }
                }
                if (VEL && a.gout) {                                  // the stored tangent is dy + gout * y
                    const f32x4 gv = *(const f32x4*)(a.gout + cg * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[e] += gv[e] * v[e];
                }
                const long oo = ((long)(a.out_g0 + cg) * a.out_pstride + o) * 4;
                *(f32x4*)(a.y + oo) = v;
                if (VEL) *(f32x4*)(a.dy + oo) = dv;
            }
        }
    }
}

template <int MODE, bool VEL, bool HAS_DX, int NI, bool G6 = false>
static void launch_conv_t(const ConvKArgs& ka, int ctiles, hipStream_t s) {
    constexpr int CK = mode_ck(MODE), GL = CK / 4, TAPS = mode_taps(MODE);
    constexpr int XV = (MODE == MODE_FLAT3) ? 320 : 256;
    constexpr int WP = TAPS * GL * 32 * NI, XP = GL * XV;
    constexpr int BUF = WP * ((VEL && !G6) ? 2 : 1) + XP * ((VEL && HAS_DX) ? 2 : 1);
    constexpr size_t smem = (size_t)2 * BUF * 16 + 256 * sizeof(int);
    auto kern = conv_mfma_kernel<MODE, VEL, HAS_DX, NI, G6>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    dim3 grid(ka.ntiles, ctiles, 1), block(256, 1, 1);
    hipLaunchKernelGGL(kern, grid, block, smem, s, ka);
}

int launch_conv(const PackedW& pw, const ConvLaunch& L, bool vel, bool has_dx, hipStream_t s) {
    ConvKArgs ka;
    ka.x = L.in.x; ka.dx = L.in.dx; ka.in_pstride = L.in.pstride;
    ka.D = L.in.D; ka.H = L.in.H; ka.W = L.in.W; ka.P = L.in.vox(); ka.in_off = L.in_off;
    ka.Dv = L.Dv; ka.Hv = L.Hv; ka.Wv = L.Wv;
    if (pw.mode == MODE_DOWN) ka.Q = (long)L.Dv * L.Hv * L.Wv;
    else ka.Q = ((long)(L.Dv - 1) * L.in.H + (L.Hv - 1)) * L.in.W + L.Wv;
    ka.y = L.out.x; ka.dy = L.out.dx; ka.out_pstride = L.out.pstride; ka.out_g0 = L.out_g0;
    ka.Ho = L.out.H; ka.Wo = L.out.W; ka.osz = L.osz; ka.oz = L.oz; ka.oy = L.oy; ka.ox = L.ox;
    ka.r = L.res.x; ka.dr = L.res.dx; ka.res_pstride = L.res.pstride;
    ka.bias = L.bias ? L.bias : pw.bias;
    const int set = L.set < 0 ? 0 : L.set;
    ka.up8 = L.set < 0 ? 1 : 0; ka.set_stride = (long)pw.floats * 4;
    ka.stem_w = pw.stem;
    ka.ww = L.wino ? pw.ww : nullptr;
    ka.wws = (L.wino && L.skw) ? L.skw->ww : nullptr; ka.wws_set_floats = L.skw ? L.skw->floats : 0;
    ka.w = pw.w + (size_t)set * pw.floats;
    ka.dw = pw.dw ? pw.dw + (size_t)set * pw.floats : nullptr;
    ka.nchunk = pw.cin_pad / prec_ck(pw.prec, pw.mode);
    ka.cout_groups = prec_is_half(pw.prec) ? (pw.cout + 7) / 8 : (pw.cout + 3) / 4;
    ka.flags = L.flags;
    ka.gout = L.gout; ka.beta = L.beta;
    // second input segment and fused skip (conv_h3g_kernel<false>; every other kernel ignores them, and the launcher
    // refuses them where they would be ignored)
    ka.x2 = L.in2.x; ka.dx2 = L.in2.dx; ka.in2_pstride = L.in2.pstride;
    if (L.csplit_ch % 16 != 0 || L.sk_split_ch % 16 != 0) return 1;          // sources switch between 16-channel chunks
    ka.csplit = L.csplit_ch > 0 ? L.csplit_ch / 16 : (1 << 30);
    ka.xs = ka.dxs = ka.xs2 = ka.dxs2 = nullptr; ka.s_pstride = ka.s2_pstride = 0; ka.s_csplit = 1 << 30;
    ka.ws = ka.dws = nullptr; ka.nskip = 0; ka.dws_delta = 0;
    if (L.skw) {
        if (L.sk.H != L.in.H || L.sk.W != L.in.W) return 1;                   // the fused skip shares the input's pitch
        // patch origin of output tile (z, y0, x0) = the skip's voxel minus one row and one column (centre tap)
        const int64_t o = (L.sk_off - L.sk.W - 1) * 4;                        // floats: one voxel of a plane is 16 bytes
        ka.xs = L.sk.x + o; ka.dxs = (L.flags & F_SKIP_NODX) ? ka.xs : L.sk.dx + o; ka.s_pstride = L.sk.pstride;
        if (L.sk_split_ch > 0) { ka.xs2 = L.sk2.x + o; ka.dxs2 = L.sk2.dx + o; ka.s2_pstride = L.sk2.pstride; ka.s_csplit = L.sk_split_ch / 16; }
        ka.ws = L.skw->w; ka.dws = L.skw->dw; ka.nskip = L.skw->cin_pad / 16;
    }
    ka.ntiles = (int)((ka.Q + TILE_VOX - 1) / TILE_VOX);
    if (prec_is_half(pw.prec)) return launch_conv_h3(pw, ka, vel, has_dx, s);
    const int ct = pw.ctiles;
    if (ka.beta) {                                               // gauged input tangent: 3x3x3 layers only
        if (!(pw.mode == MODE_FLAT3 && vel && has_dx)) return 1;   // no gauged kernel for this layer: the caller reports it
        if (pw.ni == 2) launch_conv_t<MODE_FLAT3, true, true, 2, true>(ka, ct, s);
        else launch_conv_t<MODE_FLAT3, true, true, 1, true>(ka, ct, s);
        return 0;
    }
#define NBE_DISPATCH(MODE)                                                                   \
    if (vel) {                                                                               \
        if (has_dx) { if (pw.ni == 2) launch_conv_t<MODE, true, true, 2>(ka, ct, s);        \
                      else launch_conv_t<MODE, true, true, 1>(ka, ct, s); }                 \
        else        { if (pw.ni == 2) launch_conv_t<MODE, true, false, 2>(ka, ct, s);       \
                      else launch_conv_t<MODE, true, false, 1>(ka, ct, s); }                \
    } else {                                                                                 \
        if (pw.ni == 2) launch_conv_t<MODE, false, false, 2>(ka, ct, s);                     \
        else launch_conv_t<MODE, false, false, 1>(ka, ct, s);                                \
    }
    if (pw.mode == MODE_FLAT3) { NBE_DISPATCH(MODE_FLAT3) }
    else if (pw.mode == MODE_FLAT1) { NBE_DISPATCH(MODE_FLAT1) }
    else { NBE_DISPATCH(MODE_DOWN) }
#undef NBE_DISPATCH
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight preparation
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ double block_sum(double v, double* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += scratch[i];
    return t;
}

// one workgroup per output filter: reduction over Cin*k^3 with wavefront shuffles
__global__ __launch_bounds__(256) void modulate_kernel(const float* __restrict__ weight,
                                                       const float* __restrict__ sw,
                                                       const float* __restrict__ sb,
                                                       int cin, int k3, float s0, float s1, float eps,
                                                       int first_layer, float* __restrict__ w_n,
                                                       float* __restrict__ dw_tot, const float* __restrict__ a_in,
                                                       float* __restrict__ beta_out, const float* __restrict__ b_sub) {
    __shared__ double scratch[4];
    const int co = blockIdx.x;
    const int n = cin * k3;
    const float* wr = weight + (size_t)co * n;
    double sww = 0.0, swd = 0.0;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int ci = e / k3;
        const float smod = sw[2 * ci] * s0 + sw[2 * ci + 1] * s1 + sb[ci];
        const float w = wr[e] * smod;
        const float dws = wr[e] * sw[2 * ci + 1];
        sww += (double)w * w;
        swd += (double)w * dws;
    }
    sww = block_sum(sww, scratch);
    swd = block_sum(swd, scratch);
    const float norm = sqrtf((float)sww + eps);
    const float dnorm = -(float)swd / (norm * norm * norm);
    const float inv_dz = 1.0f / (s1 + 1.0f);
    // dw_tot[co, ci, :] = w_n[co, ci, :] * (alpha[ci] + beta[co]) with alpha = ds/s (launch_style_alpha)
    if (beta_out && threadIdx.x == 0) beta_out[co] = dnorm * norm;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int ci = e / k3;
        const float smod = sw[2 * ci] * s0 + sw[2 * ci + 1] * s1 + sb[ci];
        const float w = wr[e] * smod;
        const float dws = wr[e] * sw[2 * ci + 1];
        const float wn = w / norm;
        w_n[(size_t)co * n + e] = wn;
        if (dw_tot) {
            float d = dws / norm + w * dnorm;
            if (first_layer) d += wn * inv_dz;
            if (a_in) d -= wn * a_in[ci];                        // the input's tangent is stored as dx + a_in * x
            if (b_sub) d -= wn * b_sub[co];                      // a skip fused into a gauged conv_1 (conv_h3g_kernel)
            dw_tot[(size_t)co * n + e] = d;
        }
    }
}

void launch_modulate(const float* weight, const float* style_weight, const float* style_bias,
                     int cout, int cin, int k3, float s0, float s1, float eps, int first_layer,
                     float* w_n, float* dw_tot, hipStream_t s, const float* a_in, float* beta_out, const float* b_sub) {
    hipLaunchKernelGGL(modulate_kernel, dim3(cout), dim3(256), 0, s, weight, style_weight, style_bias,
                       cin, k3, s0, s1, eps, first_layer, w_n, dw_tot, a_in, beta_out, b_sub);
}

__global__ void style_alpha_kernel(const float* __restrict__ sw, const float* __restrict__ sb, int cin, float s0,
                                   float s1, float* __restrict__ alpha, int* __restrict__ flag) {
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= cin) return;
    const float smod = sw[2 * ci] * s0 + sw[2 * ci + 1] * s1 + sb[ci];
    const float al = sw[2 * ci + 1] / smod;
    // |alpha| is capped as well: the f16-based formats store dx + alpha * x, which must stay far inside the f16 range
    const bool ok = fabsf(smod) > 1e-20f && isfinite(al) && fabsf(al) <= 64.f;
    alpha[ci] = ok ? al : 0.f;
    if (!ok) atomicOr(flag, 1);
}

void launch_style_alpha(const float* style_weight, const float* style_bias, int cin, float s0, float s1,
                        float* alpha, int* flag, hipStream_t s) {
    hipLaunchKernelGGL(style_alpha_kernel, dim3((cin + 63) / 64), dim3(64), 0, s, style_weight, style_bias, cin, s0, s1,
                       alpha, flag);
}

// packed layout: [set][ct][stage = chunk*nseg + seg][tap][gl][co][e]; channel = chunk*CK + gl*4 + e
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, int cout, int cin, int kind,
                                                   int mode, int ni, int nchunk, long floats_per_set,
                                                   int nsets, float* __restrict__ dst) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= floats_per_set * nsets) return;
    const int CK = mode_ck(mode), GL = CK / 4, TAPS = mode_taps(mode), COUT_T = 32 * ni;
    const int nstage = mode_nseg(mode) * nchunk;
    long r = idx;
    const int e = (int)(r % 4); r /= 4;
    const int co = (int)(r % COUT_T); r /= COUT_T;
    const int gl = (int)(r % GL); r /= GL;
    const int tap = (int)(r % TAPS); r /= TAPS;
    const int stage = (int)(r % nstage); r /= nstage;
    const long per_set_ct = floats_per_set / ((long)nstage * TAPS * GL * COUT_T * 4);
    const int ct = (int)(r % per_set_ct); r /= per_set_ct;
    const int set = (int)r;
    const int chunk = stage / mode_nseg(mode), seg = stage - chunk * mode_nseg(mode);
    const int ci = chunk * CK + gl * 4 + e;
    const int oc = ct * COUT_T + co;
    int k, kz, ky, kx;
    if (kind == 0) { k = 3; kz = seg / 3; ky = seg % 3; kx = tap; }
    else if (kind == 1) { k = 1; kz = ky = kx = 0; }
    else if (kind == 2) { k = 2; kz = seg >> 2; ky = (seg >> 1) & 1; kx = seg & 1; }
    else { k = 2; kz = 1 - ((set >> 2) & 1); ky = 1 - ((set >> 1) & 1); kx = 1 - (set & 1); }
    float v = 0.f;
    if (oc < cout && ci < cin) v = w[(((size_t)oc * cin + ci) * k + kz) * k * k + ky * k + kx];
    dst[idx] = v;
}

void launch_pack(const float* w_oidhw, int cout, int cin, int kind, const PackedW& pw, float* dst, hipStream_t s) {
    if (prec_is_half(pw.prec)) { launch_pack_h3(w_oidhw, cout, cin, kind, pw, dst, s); return; }
    const long total = pw.floats * pw.nsets;
    const int nchunk = pw.cin_pad / mode_ck(pw.mode);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_oidhw, cout, cin,
                       kind, pw.mode, pw.ni, nchunk, pw.floats, pw.nsets, dst);
}

// ------------------------------------------------------------------------------------------------
// data movement
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ box, int C, int Db, int Hb, int Wb,
                                                     int a0, int a1, int a2, float* __restrict__ dst,
                                                     long pstride, int G, int D, int H, int W, float scale) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long V = (long)D * H * W;
    if (v >= V) return;
    const int z = (int)(v / ((long)H * W)), rem = (int)(v - (long)z * H * W);
    const int y = rem / W, x = rem - y * W;
    const int bz = ((a0 + z) % Db + Db) % Db, by = ((a1 + y) % Hb + Hb) % Hb, bx = ((a2 + x) % Wb + Wb) % Wb;
    const long bo = ((long)bz * Hb + by) * Wb + bx;
    const long bstride = (long)Db * Hb * Wb;
    for (int g = 0; g < G; ++g) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * g + e;
            o[e] = c < C ? box[c * bstride + bo] * scale : 0.f;
        }
        *(f32x4*)(dst + ((long)g * pstride + v) * 4) = o;
    }
}

void launch_gather(const float* box, int C, int Db, int Hb, int Wb, int a0, int a1, int a2,
                   const Planes& dst, float scale, int prec, hipStream_t s) {
    if (prec_is_half(prec)) { launch_gather_h8(box, C, Db, Hb, Wb, a0, a1, a2, dst.x, dst, scale, prec_parts(prec), s); return; }
    const long V = dst.vox();
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, s, box, C, Db, Hb, Wb,
                       a0, a1, a2, dst.x, dst.pstride, dst.G, dst.D, dst.H, dst.W, scale);
}

void launch_to_planes(const float* src, int C, const Planes& dst, bool tangent, float scale, int prec, hipStream_t s) {
    if (prec_is_half(prec)) { launch_gather_h8(src, C, dst.D, dst.H, dst.W, 0, 0, 0, tangent ? dst.dx : dst.x, dst, scale, prec_parts(prec), s); return; }
    // a dense (C,D,H,W) array is a "box" of the same size gathered at origin 0
    const long V = dst.vox();
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, s, src, C, dst.D, dst.H,
                       dst.W, 0, 0, 0, tangent ? dst.dx : dst.x, dst.pstride, dst.G, dst.D, dst.H, dst.W, scale);
}

__global__ __launch_bounds__(256) void from_planes_kernel(const float* __restrict__ src, long pstride, long V, int C,
                                                          float* __restrict__ dst) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    for (int g = 0; 4 * g < C; ++g) {
        const f32x4 t = *(const f32x4*)(src + ((long)g * pstride + v) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * g + e < C) dst[(long)(4 * g + e) * V + v] = t[e];
    }
}

void launch_from_planes(const Planes& src, bool tangent, int C, float* dst, int prec, hipStream_t s) {
    if (prec_is_half(prec)) { launch_from_planes_h8(tangent ? src.dx : src.x, src, C, dst, prec_parts(prec), s); return; }
    const long V = src.vox();
    hipLaunchKernelGGL(from_planes_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, s,
                       tangent ? src.dx : src.x, src.pstride, V, C, dst);
}

__global__ __launch_bounds__(256) void crop_kernel(const float* __restrict__ sx, const float* __restrict__ sdx,
                                                   long spstride, int SH, int SW, int c, float* __restrict__ dx_,
                                                   float* __restrict__ ddx, long dpstride, int g0, int G, int D,
                                                   int H, int W, int cz) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long V = (long)D * H * W;
    if (v >= V) return;
    const int g = blockIdx.y;
    const int z = (int)(v / ((long)H * W)), rem = (int)(v - (long)z * H * W);
    const int y = rem / W, x = rem - y * W;
    const long sv = ((long)(z + cz) * SH + (y + c)) * SW + (x + c);
    const long so = ((long)g * spstride + sv) * 4, dof = ((long)(g0 + g) * dpstride + v) * 4;
    *(f32x4*)(dx_ + dof) = *(const f32x4*)(sx + so);
    if (sdx) *(f32x4*)(ddx + dof) = *(const f32x4*)(sdx + so);
}

void launch_crop(const Planes& src, int c, const Planes& dst, int g0, bool vel, hipStream_t s, int cz) {
    const long V = dst.vox();
    hipLaunchKernelGGL(crop_kernel, dim3((unsigned)((V + 255) / 256), src.G), dim3(256), 0, s, src.x,
                       vel ? src.dx : nullptr, src.pstride, src.H, src.W, c, dst.x, dst.dx, dst.pstride, g0,
                       src.G, dst.D, dst.H, dst.W, cz < 0 ? c : cz);
}

// Periodic y/x halo of a tensor whose interior [pad, H - pad) x [pad, W - pad) has been written: every halo voxel
// takes the interior voxel one period away.  16-byte units, all channel planes, primal and tangent.
__global__ __launch_bounds__(256) void fill_yx_kernel(float* __restrict__ x, float* __restrict__ dx, long pstride,
                                                      int G, int D, int H, int W, int pad) {
    const int Hi = H - 2 * pad, Wi = W - 2 * pad;
    const long per = (long)H * W - (long)Hi * Wi;                 // halo voxels of one z plane
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * D) return;
    const int z = (int)(idx / per);
    long r = idx - (long)z * per;
    int y, xx;
    if (r < (long)pad * W) { y = (int)(r / W); xx = (int)(r - (long)y * W); }                           // top rows
    else if (r < 2L * pad * W) { r -= (long)pad * W; y = H - pad + (int)(r / W); xx = (int)(r % W); }  // bottom rows
    else { r -= 2L * pad * W; y = pad + (int)(r / (2 * pad)); const int k = (int)(r % (2 * pad)); xx = k < pad ? k : W - 2 * pad + k; }
    const int ys = pad + ((y - pad) % Hi + Hi) % Hi, xs = pad + ((xx - pad) % Wi + Wi) % Wi;
    const long d = ((long)z * H + y) * W + xx, sidx = ((long)z * H + ys) * W + xs;
    const int g = blockIdx.y;
    *(f32x4*)(x + ((long)g * pstride + d) * 4) = *(const f32x4*)(x + ((long)g * pstride + sidx) * 4);
    if (dx) *(f32x4*)(dx + ((long)g * pstride + d) * 4) = *(const f32x4*)(dx + ((long)g * pstride + sidx) * 4);
}

void launch_fill_yx(const Planes& t, int pad, bool vel, hipStream_t s) {
    const long per = (long)t.H * t.W - (long)(t.H - 2 * pad) * (t.W - 2 * pad);
    const long n = per * t.D;
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_yx_kernel, dim3((unsigned)((n + 255) / 256), t.G), dim3(256), 0, s, t.x, vel ? t.dx : nullptr,
                       t.pstride, t.G, t.D, t.H, t.W, pad);
}

// dst (D, Hs + 2*pad, Ws + 2*pad) = src (Ds, Hs, Ws) extended periodically in y and x by `pad` voxels and, when
// padz > 0, in z by padz planes (D = Ds + 2*padz; padz = 0: D = Ds, planes map one to one)
__global__ __launch_bounds__(256) void wrap_pad_kernel(const float* __restrict__ sx, const float* __restrict__ sdx,
                                                       long spstride, int Ds, int Hs, int Ws, float* __restrict__ dx_,
                                                       float* __restrict__ ddx, long dpstride, int D, int H, int W,
                                                       int pad, int padz) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= (long)D * H * W) return;
    const int z = (int)(v / ((long)H * W)), rem = (int)(v - (long)z * H * W);
    const int y = rem / W, x = rem - y * W;
    const int ys = ((y - pad) % Hs + Hs) % Hs, xs = ((x - pad) % Ws + Ws) % Ws;
    const int zs = ((z - padz) % Ds + Ds) % Ds;
    const long sv = ((long)zs * Hs + ys) * Ws + xs;
    const int g = blockIdx.y;
    *(f32x4*)(dx_ + ((long)g * dpstride + v) * 4) = *(const f32x4*)(sx + ((long)g * spstride + sv) * 4);
    if (sdx) *(f32x4*)(ddx + ((long)g * dpstride + v) * 4) = *(const f32x4*)(sdx + ((long)g * spstride + sv) * 4);
}

void launch_wrap_pad(const Planes& src, const Planes& dst, int pad, bool vel, hipStream_t s, int padz) {
    const long V = dst.vox();
    hipLaunchKernelGGL(wrap_pad_kernel, dim3((unsigned)((V + 255) / 256), src.G), dim3(256), 0, s, src.x,
                       vel ? src.dx : nullptr, src.pstride, src.D, src.H, src.W, dst.x, dst.dx, dst.pstride, dst.D, dst.H,
                       dst.W, pad, padz);
}

template <typename OT>
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                   long ypstride, int D, int H, int W,
                                                   const float* __restrict__ xin, long xpstride, int XH, int XW,
                                                   int c0, int C, float k_disp, float k_dy, float k_x0,
                                                   OT* __restrict__ disp, OT* __restrict__ velo, int Db, int Hb, int Wb,
                                                   int a0, int a1, int a2, int pad, int* __restrict__ bad) {
    // pad > 0: y and xin carry a periodic y/x halo of `pad` voxels; the loop runs over the interior
    const int Hi = H - 2 * pad, Wi = W - 2 * pad, c1 = pad > 0 ? pad : c0;
    const long vi = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vi >= (long)D * Hi * Wi) return;
    const int z = (int)(vi / ((long)Hi * Wi)), rem = (int)(vi - (long)z * Hi * Wi);
    const int yy = rem / Wi, x = rem - yy * Wi;
    const long v = ((long)z * H + yy + pad) * W + x + pad;
    const long xv = ((long)(z + c0) * XH + (yy + c1)) * XW + (x + c1);
    const long bo = ((long)(a0 + z) * Hb + (a1 + yy)) * Wb + (a2 + x);
    const long bstride = (long)Db * Hb * Wb;
    bool nf = false;
    for (int g = 0; 4 * g < C; ++g) {
        const f32x4 yv = *(const f32x4*)(y + ((long)g * ypstride + v) * 4);
        const f32x4 x0 = *(const f32x4*)(xin + ((long)g * xpstride + xv) * 4);
        f32x4 dv = {0.f, 0.f, 0.f, 0.f};
        if (dy) dv = *(const f32x4*)(dy + ((long)g * ypstride + v) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * g + e;
            if (c < C) {
                const float dsp = (yv[e] + x0[e]) * k_disp;
                disp[c * bstride + bo] = (OT)dsp;
                nf = nf || !isfinite(dsp);
                if (dy) {
                    const float vv = dv[e] * k_dy + x0[e] * k_x0;
                    velo[c * bstride + bo] = (OT)vv;
                    nf = nf || !isfinite(vv);
                }
            }
        }
    }
    if (nf && bad) atomicOr(bad, 1);
}

void launch_head(const Planes& y, const Planes& xin, int c0, int C, const HeadScale& hs, bool vel,
                 void* disp, void* velo, int out_dtype, int Db, int Hb, int Wb, int a0, int a1, int a2,
                 int prec, hipStream_t s, int pad) {
    if (prec_is_half(prec)) { launch_head_h8(y, xin, c0, C, hs, vel, disp, velo, out_dtype, Db, Hb, Wb, a0, a1, a2, prec_parts(prec), s, pad); return; }
    const long V = (long)y.D * (y.H - 2 * pad) * (y.W - 2 * pad);
    dim3 grid((unsigned)((V + 255) / 256)), block(256);
    if (out_dtype == 0)
        hipLaunchKernelGGL(head_kernel<float>, grid, block, 0, s, y.x, vel ? y.dx : nullptr, y.pstride, y.D, y.H,
                           y.W, xin.x, xin.pstride, xin.H, xin.W, c0, C, hs.k_disp, hs.k_dy, hs.k_x0, (float*)disp,
                           (float*)velo, Db, Hb, Wb, a0, a1, a2, pad, hs.bad);
    else
        hipLaunchKernelGGL(head_kernel<_Float16>, grid, block, 0, s, y.x, vel ? y.dx : nullptr, y.pstride, y.D,
                           y.H, y.W, xin.x, xin.pstride, xin.H, xin.W, c0, C, hs.k_disp, hs.k_dy, hs.k_x0,
                           (_Float16*)disp, (_Float16*)velo, Db, Hb, Wb, a0, a1, a2, pad, hs.bad);
}

// max |x| of a float array as a bit pattern (grid-stride, 16 B per lane, wave shuffle reduction, one atomic per wave)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ src, long n, unsigned* __restrict__ out) {
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned m = 0;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint4 v = *(const uint4*)(src + 4 * i);
        m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(src[(n4 << 2) + threadIdx.x]) & 0x7fffffffu);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_down((int)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

void launch_absmax(const float* src, int64_t n, unsigned* out_bits, hipStream_t s) {
    if (n <= 0) return;
    // 16-byte loads need a 16-byte aligned base: hipMalloc / torch allocations are; anything else goes scalar
    if (((uintptr_t)src & 15) != 0) {
        const long head = std::min<long>(n, (16 - ((uintptr_t)src & 15)) / 4);
        hipLaunchKernelGGL(absmax_kernel, dim3(1), dim3(256), 0, s, src, head, out_bits);   // 1..3 elements: the tail path
        if (head >= n) return;
        src += head; n -= head;
    }
    const unsigned blocks = (unsigned)std::min<long>(2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, src, (long)n, out_bits);
}

// zero `units` 16-byte units (a kernel rather than hipMemsetAsync: it also runs inside captured graphs, where the
// schedule should consist of kernel nodes only)
__global__ __launch_bounds__(256) void zero_kernel(f32x4* __restrict__ dst, long units) {
    const long stride = (long)gridDim.x * blockDim.x;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += stride) dst[i] = z;
}

void launch_zero(void* dst, int64_t bytes, hipStream_t s) {
    const long units = bytes / 16;
    if (units <= 0) return;
    const unsigned blocks = (unsigned)std::min<long>(4096, (units + 255) / 256);
    hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, s, (f32x4*)dst, units);
}

__global__ void scale_kernel(const float* __restrict__ src, const float* __restrict__ src2, float* __restrict__ dst, int n, float f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (src[i] + (src2 ? src2[i] : 0.f)) * f;
}

void launch_scale(const float* src, float* dst, int n, float f, hipStream_t s, const float* src2) {
    if (n > 0) hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, src2, dst, n, f);
}

// one word at the end of an exchange buffer: written by the sender, compared by the receiver (brick mode: every rank must
// have computed its faces with the same range shift)
__global__ void tag_word_kernel(unsigned* dst, unsigned v, const unsigned* expect_at, unsigned* flag, unsigned bit) {
    if (dst) *dst = v;
    if (expect_at && *expect_at != v) atomicOr(flag, bit);
}
void launch_tag_word(unsigned* dst, unsigned v, const unsigned* expect_at, unsigned* flag, unsigned bit, hipStream_t s) {
    hipLaunchKernelGGL(tag_word_kernel, dim3(1), dim3(1), 0, s, dst, v, expect_at, flag, bit);
}

// ---- branch probe (test instrumentation: include/nbe.h, "Branch probe") ---------------------------------------------
// One thread per 32-voxel word of the probe tensor (C, n, n, nw): bit = "the LeakyReLU behind this stored activation took
// the identity branch" = stored value > 0 (LeakyReLU keeps the sign; exactly zero is the slope branch, layers_vel.py:184-185).
__global__ __launch_bounds__(256) void probe_signs_kernel(ProbeLaunch a) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.C * a.n * a.n * a.nw;
    if (t >= total) return;
    const int w = (int)(t % a.nw), jy = (int)((t / a.nw) % a.n), jz = (int)((t / ((long)a.nw * a.n)) % a.n), ch = (int)(t / ((long)a.nw * a.n * a.n));
    // z: a tensor that exists for the planes [zlo, zhi) of a box that is periodic along z holds plane i as i, i - zper or i + zper
    int zi = a.o[0] + jz;
    if (a.zper > 0 && (zi < a.zlo || zi >= a.zhi)) {
        if (zi - a.zper >= a.zlo && zi - a.zper < a.zhi) zi -= a.zper;
        else if (zi + a.zper >= a.zlo && zi + a.zper < a.zhi) zi += a.zper;
    }
    int iz = zi - a.org[0], iy = a.o[1] + jy - a.org[1];
    if (a.per[1] > 0) iy = ((iy % a.per[1]) + a.per[1]) % a.per[1];
    if (iz < 0 || iz >= a.ext[0] || iy < 0 || iy >= a.ext[1]) return;
    unsigned bits = 0; int covered = 0, want = 0;
    for (int b = 0; b < 32; ++b) {
        const int jx = 32 * w + b;
        if (jx >= a.n) break;
        ++want;
        int ix = a.o[2] + jx - a.org[2];
        if (a.per[2] > 0) ix = ((ix % a.per[2]) + a.per[2]) % a.per[2];
        if (ix < 0 || ix >= a.ext[2]) continue;
        ++covered;
        const long vox = ((long)iz * a.H + iy) * a.W + ix;
        bool up;
        if (a.prec == PREC_F32) {
            up = a.x[((long)(a.g0 + ch / 4) * a.pstride + vox) * 4 + (ch & 3)] > 0.f;
        } else {
            const int plane = a.prec == PREC_F16X3 ? a.g0 + 2 * (ch / 8) : a.g0 + ch / 8;
            const _Float16* hp = (const _Float16*)(a.x + ((long)plane * a.pstride + vox) * 4);
            const float hi = (float)hp[ch & 7];
            up = hi > 0.f;
            if (a.prec == PREC_F16X3 && hi == 0.f) up = (float)((const _Float16*)(a.x + ((long)(plane + 1) * a.pstride + vox) * 4))[ch & 7] > 0.f;
        }
        bits |= (up ? 1u : 0u) << b;
    }
    if (covered == 0) return;
    a.bits[t] = bits;
    atomicAdd(a.count, covered == want ? 1u : 0x40000000u);      // a partly covered row is an error of the caller's bookkeeping
}

void launch_probe_signs(const ProbeLaunch& a, hipStream_t s) {
    const long total = (long)a.C * a.n * a.n * a.nw;
    if (total > 0) hipLaunchKernelGGL(probe_signs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

}  // namespace nbe
