// gfx950 kernels of the float32-equivalent split-precision path (PREC_F16X3).
//
// Same algorithm, schedule and tiling as nbe_kernels.hip (flat-shift implicit GEMM, LDS staging by
// global->LDS DMA, XCD-aware tiles); what changes is the arithmetic: every operand is a pair of f16 numbers
// (hi = f16(x), lo = f16((x - hi) * 2^11)) and one float32 product becomes three f16 MFMAs with float32
// accumulation,   a*b  ~=  a_hi*b_hi  +  2^-11 * (a_hi*b_lo + a_lo*b_hi),
// i.e. 22 significant bits per operand.  v_mfma_f32_32x32x16_f16 runs at 16x the rate of the f32 MFMA, so the
// 3-term split is ~5.3x the strict-float32 matrix rate.  Accuracy is held to the float32 tolerances by the GPU
// tests, which are parametrised over both precisions (tests/test_gpu_layers.py, test_gpu_model.py, test_gpu_api.py).
//
// Two convolution kernels: conv_h3p_kernel (2-D patches, the 3x3x3 layers = 98 % of the FLOPs) and
// conv_h3_kernel (flat positions: 1x1x1 skips, up-sample parities, stride-2 down-sample; 3x3x3 only for A/B).
//
// Storage: plane 2g holds hi, plane 2g+1 holds lo of channels 8g..8g+7 (8 x f16 = 16 B per voxel): one
// 16-byte unit is exactly the A/B operand of one lane for one MFMA (k = 8*(lane>>5) + j).

#include "nbe_kernels_internal.h"
#include <cstdlib>
#include <cstdio>
#include <algorithm>

#ifndef NBE_DBG
#define NBE_DBG 0          // 1: compile the timing-experiment switches (python: NBE_BUILD_DBG=1)
#endif
#ifndef NBE_DBG_SHAPE16
#define NBE_DBG_SHAPE16 0  // 1: MFMA-shape timing probe in conv_h3p_kernel (results invalid)
#endif

namespace nbe {

extern __shared__ __attribute__((aligned(16))) f32x4 lds_h3[];

__device__ __forceinline__ void split4(const f32x4 v, half4& hi, half4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (_Float16)v[e];
        // (v - hi) 2^11 as one mixed-precision FMA on the f16 half where it lies: v 2^11 is exact, hi 2^11 is exact and so is
        // their difference -- the same bits as (v - float(hi)) * 2^11, two vector operations fewer per element
        lo[e] = (_Float16)__builtin_fmaf((float)hi[e], -H3_SCALE, v[e] * H3_SCALE);
    }
}
__device__ __forceinline__ f32x4 join4(const half4 hi, const half4 lo) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)hi[e] + (float)lo[e] * H3_INV;
    return v;
}

// Epilogue of the two 32x32 accumulator tiles of a wave (column tiles jt = 0, 1; this lane's output voxels o[jt]):
// join main + correction, bias, residual, LeakyReLU (+ tangent) in float32, then split to hi/lo and store 4 channels
// (8 B) per part.  it = cout half of the wave, lh = lane half; register 4k+e of a tile is cout 32*it + 8k + 4*lh + e.
// Every global load (bias, residuals of all 8 (jt, k) pieces) is issued before the first use: written piece by
// piece, each piece's loads wait for the previous piece's stores (vmcnt counts both) and the epilogue serialises on
// memory latency.  Lanes without a valid voxel (ok[jt] false) must pass a valid dummy o[jt]; they skip the stores.
template <bool VEL, bool SPLIT>
__device__ __forceinline__ void h3_store2(const ConvKArgs& a, int ct, int it, int lh, const long (&o)[2],
                                          const bool (&ok)[2], const f32x16 (&ym)[2], const f32x16 (&yc)[2],
                                          const f32x16 (&dm)[2], const f32x16 (&dc)[2]) {
    constexpr int PARTS = SPLIT ? 2 : 1;
    const bool act = a.flags & F_ACT, res = a.flags & F_RES;
    int unit[4];
    bool uok[4];
    f32x4 bv[4], gv[4];
    const bool gauge = VEL && a.gout;                            // stored tangent = dy + gout[o] * y (see conv_h3g_kernel)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unit[k] = ct * 8 + 4 * it + k;                           // 8-channel group of the output
        uok[k] = unit[k] < a.cout_groups;
        if (!uok[k]) unit[k] = a.cout_groups - 1;
        bv[k] = *(const f32x4*)(a.bias + unit[k] * 8 + 4 * lh);
        if (gauge) gv[k] = *(const f32x4*)(a.gout + unit[k] * 8 + 4 * lh);
    }
    half4 rh[8], rl[8], dh[8], dl[8];
    if (res) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const long rb = ((long)(PARTS * unit[p & 3]) * a.res_pstride + o[p >> 2]) * 16 + 8 * lh;
            const long rlo = rb + a.res_pstride * 16;
            rh[p] = *(const half4*)((const char*)a.r + rb);
            if (SPLIT) rl[p] = *(const half4*)((const char*)a.r + rlo);
            if (VEL) {
                dh[p] = *(const half4*)((const char*)a.dr + rb);
                if (SPLIT) dl[p] = *(const half4*)((const char*)a.dr + rlo);
            }
        }
    }
    // (16-byte stores through v_permlane32_swap pairs -- guide T21 -- were measured on the first layer, the up-sampling and
    // the stride-2 launches, which are write-bound: no change; the 8-byte form below keeps 24 registers fewer alive.)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int jt = p >> 2, k = p & 3;
        f32x4 v, dv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = ym[jt][4 * k + e] + (SPLIT ? yc[jt][4 * k + e] * H3_INV : 0.f) + bv[k][e];
            dv[e] = dm[jt][4 * k + e] + (SPLIT ? dc[jt][4 * k + e] * H3_INV : 0.f);
        }
        if (res) {
            if (SPLIT) {
                v += join4(rh[p], rl[p]);
                if (VEL) dv += join4(dh[p], dl[p]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] += (float)rh[p][e];
                    if (VEL) dv[e] += (float)dh[p][e];
                }
            }
        }
        if (act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (VEL) dv[e] = v[e] > 0.f ? dv[e] : 0.01f * dv[e];
                v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
            }
        }
        if (gauge) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dv[e] += gv[k][e] * v[e];
        }
        if (uok[k] && ok[jt]) {
            const long ob = ((long)(a.out_g0 + PARTS * unit[k]) * a.out_pstride + o[jt]) * 16 + 8 * lh;
            const long ol = ob + a.out_pstride * 16;
            half4 hi, lo;
            split4(v, hi, lo);
            *(half4*)((char*)a.y + ob) = hi;
            if (SPLIT) *(half4*)((char*)a.y + ol) = lo;
            if (VEL) {
                split4(dv, hi, lo);
                *(half4*)((char*)a.dy + ob) = hi;
                if (SPLIT) *(half4*)((char*)a.dy + ol) = lo;
            }
        }
    }
}

// Workgroup: 512 threads = 8 waves; tile = 64 output channels x 256 flat positions.
// wave w: it = w & 1 (32 couts), jq = w >> 1 (64 positions = 2 MFMA column tiles).
//
// Pipeline: weights live in a 2-deep LDS ring, activation row segments in a ring of XDEPTH (2 or 3) stages.
// During stage s the DMA of W(s+1) and X(s+XDEPTH-1) is issued BETWEEN the MFMAs of the first tap.
//   XDEPTH 2 (default): every stage ends with vmcnt(0) + barrier, one stage of DMA in flight.
//   XDEPTH 3: a counted s_waitcnt leaves this wave's X(s+2) pieces in flight across a raw s_barrier (vmcnt
//             counts in issue order: X(s+1), issued a whole stage earlier, and W(s+1) are complete).
// Same-device A/B (env NBE_H3_DEPTH): depth 2 = 343, depth 3 = 332 TFLOP/s-equivalent on the 512^3 bench --
// DMA latency is not what limits these kernels (see the power note at conv_h3p_kernel).

// SPLIT = true: f16x3 (hi and lo planes, three MFMAs per product); SPLIT = false: plain f16 (PREC_F16: hi planes
// only, one MFMA per product -- the arithmetic of the reference's dtype=float16 configuration).
// A 16-channel chunk is UN = 2*PARTS units of 8 channels: unit = PARTS*h + part (h = MFMA lane half, part 0 = hi).
template <int MODE, bool SPLIT>
struct H3Geom {
    static constexpr int PARTS = SPLIT ? 2 : 1, UN = 2 * PARTS;
    static constexpr int TAPS = mode_taps(MODE);
    static constexpr int XV = (MODE == MODE_FLAT3) ? 288 : 256;   // 256 + 2 halo voxels, rounded to 32
    static constexpr int WP = TAPS * UN * 64;                     // 16-byte units: weights of one stage
    static constexpr int XP = UN * XV;                            // 16-byte units: activations of one stage
};

template <int MODE, bool VEL, bool HAS_DX, int H3_XDEPTH, bool SPLIT>
__global__ __launch_bounds__(512, 2) void conv_h3_kernel(ConvKArgs a) {
    typedef H3Geom<MODE, SPLIT> G;
    constexpr int TAPS = G::TAPS, XV = G::XV, WP = G::WP, XP = G::XP, PARTS = G::PARTS, UN = G::UN;
    constexpr bool DX = VEL && HAS_DX;
    constexpr int WB = WP * (VEL ? 2 : 1);           // one weight buffer  (W [, dW])
    constexpr int XB = XP * (DX ? 2 : 1);            // one activation buffer (X [, dX])
    constexpr int OFF_DW = WP, OFF_DXX = XP;
    constexpr int XRING = 2 * WB;                    // start of the activation ring
    constexpr int NW_TOT = WB / 64, NX_TOT = (XB + 63) / 64;
    constexpr int NWS = (NW_TOT + 7) / 8, NXS = (NX_TOT + 7) / 8, NX_REM = NX_TOT % 8;
    static_assert(WB % 64 == 0, "weight stage must be whole wave-instructions");
    static_assert(XP % 32 == 0, "activation planes must be half-wave multiples");
    static_assert(NWS + NXS <= 9, "more DMA slots per wave than MFMA pairs in one tap");

    f32x4* lds = lds_h3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int it = wave & 1, jq = wave >> 1;

    const int tile = xcd_tile(blockIdx.x, a.ntiles);
    const int ct = blockIdx.y;
    const long q0 = (long)tile * TILE_VOX;
    const int nstage = mode_nseg(MODE) * a.nchunk;
    const long HW = (long)a.H * a.W;

    int* inbase = (int*)(lds + XRING + H3_XDEPTH * XB);
    if (MODE == MODE_DOWN) {
        if (tid < TILE_VOX) {
            long o = q0 + tid;
            if (o > a.Q - 1) o = a.Q - 1;
            const int hw = a.Hv * a.Wv;
            const int zo = (int)(o / hw), rem = (int)(o - (long)zo * hw);
            const int yo = rem / a.Wv, xo = rem - yo * a.Wv;
            inbase[tid] = (int)((2L * zo * a.H + 2 * yo) * a.W + 2 * xo);
        }
        __syncthreads();
    }

    // ---- DMA slots of this wave.  Everything that does not depend on the stage is fixed here; per stage one
    // wave-uniform byte offset is added (weights: stage*WP*16; activations: chunk planes + row segment).
    // Activation reads may run past the end of a plane by < 2*H*W + 2*W + 400 voxels for flat positions whose
    // outputs are discarded: every tensor is allocated with that much slack (engine: ws_planes).
    // An activation wave-instruction covers 64 consecutive 16-byte units of the [plane][voxel] stage image; with
    // XV = 288 it may straddle two planes (a half-wave each), which the per-lane source address handles.
    const char* wsrc[NWS];
    int wdst[NWS];
    const char* xsrc[NXS];
    int xdst[NXS];
#pragma unroll
    for (int t = 0; t < NWS; ++t) {
        const int n = wave + 8 * t;
        wsrc[t] = nullptr; wdst[t] = 0;
        if (n < NW_TOT) {
            const bool d = VEL && n >= WP / 64;
            const int m = n - (d ? WP / 64 : 0);
            wsrc[t] = (const char*)(d ? a.dw : a.w) + ((long)ct * nstage * WP + m * 64 + lane) * 16;
            wdst[t] = (d ? OFF_DW : 0) + m * 64;
        }
    }
#pragma unroll
    for (int t = 0; t < NXS; ++t) {
        const int n = wave + 8 * t;
        xsrc[t] = nullptr; xdst[t] = 0;
        if (n < NX_TOT) {
            const int u = n * 64 + lane;             // unit inside [X planes 0..3][dX planes 0..3]
            const bool tang = DX && u >= XP;
            const int uu = u - (tang ? XP : 0);
            const int pl = uu / XV, vl = uu - pl * XV;
            long v;
            if (MODE == MODE_DOWN) v = (long)inbase[vl];
            else v = q0 + a.in_off + vl;
            xsrc[t] = (const char*)(tang ? a.dx : a.x) + ((long)pl * a.in_pstride + v) * 16;
            xdst[t] = (tang ? OFF_DXX : 0) + uu;     // lane-linear: dst base = unit of lane 0
            xdst[t] = __builtin_amdgcn_readfirstlane(xdst[t]);
        }
    }

    auto seg_offset = [&](int s) -> long {
        const int chunk = s / mode_nseg(MODE), seg = s - chunk * mode_nseg(MODE);
        long segoff;
        if (MODE == MODE_FLAT3) segoff = (seg / 3) * HW + (seg % 3) * a.W;
        else if (MODE == MODE_DOWN) segoff = (seg >> 2) * HW + ((seg >> 1) & 1) * a.W + (seg & 1);
        else segoff = 0;
        return ((long)chunk * UN * a.in_pstride + segoff) * 16;
    };
    auto dma_w = [&](int t, int s) {                 // weights of stage s -> weight buffer s & 1
        if (wave + 8 * t < NW_TOT) dma16((const float*)(wsrc[t] + (long)s * WP * 16), lds + (s & 1) * WB + wdst[t]);
    };
    auto dma_x = [&](int t, long xoff, int ring) {   // activations (offset xoff) -> ring slot
        if (wave + 8 * t < NX_TOT) dma16((const float*)(xsrc[t] + xoff), lds + XRING + ring * XB + xdst[t]);
    };
    // leave this wave's activation pieces of ONE stage in flight (its count is NXS or NXS-1 by wave)
    auto wait_keep_x = [&]() {
        if (NX_REM == 0 || wave < NX_REM) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NXS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NXS - 1) : "memory");
    };
    auto wait_all = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // main / correction accumulators of y and dy: 2 column tiles each
    f32x16 ym[2], yc[2], dm[2], dc[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { ym[jt][e] = 0.f; yc[jt][e] = 0.f; dm[jt][e] = 0.f; dc[jt][e] = 0.f; }

    // operands of one dx tap
    struct Ops { half8 wh, wl, dwh, dwl, xh[2], xl[2], dxh[2], dxl[2]; };
    auto load_ops = [&](const half8* wb, const half8* xb, int tap, Ops& o) {
        const int wo = (tap * UN + PARTS * lh) * 64 + 32 * it + li;
        o.wh = wb[wo];
        if (SPLIT) o.wl = wb[wo + 64];
        if (VEL) { o.dwh = wb[OFF_DW + wo]; if (SPLIT) o.dwl = wb[OFF_DW + wo + 64]; }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int xo = (PARTS * lh) * XV + jq * 64 + 32 * jt + li + (MODE == MODE_FLAT3 ? tap : 0);
            o.xh[jt] = xb[xo];
            if (SPLIT) o.xl[jt] = xb[xo + XV];
            if (DX) { o.dxh[jt] = xb[OFF_DXX + xo]; if (SPLIT) o.dxl[jt] = xb[OFF_DXX + xo + XV]; }
        }
    };
    // the i-th of the 18 MFMAs of one tap (i is a compile-time constant after unrolling): jt = i / 9
    auto mfma1 = [&](const Ops& o, int i) {
        const int jt = i / 9, k = i % 9;
        if (k == 0) ym[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wh, o.xh[jt], ym[jt], 0, 0, 0);
        if (SPLIT && k == 1) yc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wh, o.xl[jt], yc[jt], 0, 0, 0);
        if (SPLIT && k == 2) yc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wl, o.xh[jt], yc[jt], 0, 0, 0);
        if (VEL) {
            if (k == 3) dm[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.dwh, o.xh[jt], dm[jt], 0, 0, 0);
            if (SPLIT && k == 4) dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.dwh, o.xl[jt], dc[jt], 0, 0, 0);
            if (SPLIT && k == 5) dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.dwl, o.xh[jt], dc[jt], 0, 0, 0);
        }
        if (DX) {
            if (k == 6) dm[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wh, o.dxh[jt], dm[jt], 0, 0, 0);
            if (SPLIT && k == 7) dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wh, o.dxl[jt], dc[jt], 0, 0, 0);
            if (SPLIT && k == 8) dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wl, o.dxh[jt], dc[jt], 0, 0, 0);
        }
    };
    auto mfma_ops = [&](const Ops& o) {
#pragma unroll
        for (int i = 0; i < 18; ++i) mfma1(o, i);
    };

    // ---- prologue: X(0), W(0), X(1) -------------------------------------------------------------------
    {
        const long x0 = seg_offset(0);
#pragma unroll
        for (int t = 0; t < NXS; ++t) dma_x(t, x0, 0);
#pragma unroll
        for (int t = 0; t < NWS; ++t) dma_w(t, 0);
        if (H3_XDEPTH == 3 && nstage > 1) {
            const long x1 = seg_offset(1);
#pragma unroll
            for (int t = 0; t < NXS; ++t) dma_x(t, x1, 1);
            wait_keep_x();
        } else {
            wait_all();
        }
        barrier();
    }

    int ring = 0;                                    // ring slot of X(s)
    for (int s = 0; s < nstage; ++s) {
        // activations prefetched during this stage: X(s + XDEPTH - 1) into the ring slot freed by stage s-1
        const bool pw = s + 1 < nstage, px = s + H3_XDEPTH - 1 < nstage;
        const long xoff2 = px ? seg_offset(s + H3_XDEPTH - 1) : 0;
        const int ring2 = H3_XDEPTH == 3 ? (ring >= 1 ? ring - 1 : 2) : (ring ^ 1);
        const half8* wb = (const half8*)(lds + (s & 1) * WB);
        const half8* xb = (const half8*)(lds + XRING + ring * XB);
        Ops o0, o1;
        load_ops(wb, xb, 0, o0);
        if (TAPS > 1) load_ops(wb, xb, 1, o1);
        __builtin_amdgcn_sched_barrier(0);
        // first tap: one DMA instruction after every second MFMA -- W(s+1) first, then X(s+2)
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            mfma1(o0, i);
            if (i & 1) {
                const int t = i >> 1;
                if (t < NWS) { if (pw) dma_w(t, s + 1); }
                else if (t - NWS < NXS) { if (px) dma_x(t - NWS, xoff2, ring2); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (TAPS > 1) {
            load_ops(wb, xb, 2, o0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_ops(o1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_ops(o0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (H3_XDEPTH == 3 && px) wait_keep_x(); else wait_all();
        barrier();
        ring = ring == H3_XDEPTH - 1 ? 0 : ring + 1;
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    long o[2];
    bool ok[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const long q = q0 + jq * 64 + 32 * jt + li;
        ok[jt] = q < a.Q;
        if (MODE == MODE_DOWN) {
            o[jt] = q;
        } else {
            const int z = (int)(q / HW), rem = (int)(q - (long)z * HW);
            const int yy = rem / a.W, xx = rem - yy * a.W;
            ok[jt] = ok[jt] && xx < a.Wv && yy < a.Hv && z < a.Dv;
            o[jt] = ((long)(z * a.osz + a.oz) * a.Ho + (yy * a.osz + a.oy)) * a.Wo + (xx * a.osz + a.ox);
        }
        if (!ok[jt]) o[jt] = 0;
    }
    h3_store2<VEL, SPLIT>(a, ct, it, lh, o, ok, ym, yc, dm, dc);
}

// ------------------------------------------------------------------------------------------------
// 2x up-sampling as ONE launch (f16x3, velocity)
// ------------------------------------------------------------------------------------------------
// StyleTransposeBase3DVel (style_layers_vel.py:234-269) is eight 1x1x1 GEMMs on the same input, one per output parity
// (weight set p = 4 oz + 2 oy + ox).  Run as eight launches of conv_h3_kernel<MODE_FLAT1> the input was read eight
// times and every launch wrote every other 16-byte unit of the output rows.  Here a workgroup keeps its 256 input
// positions x all Cin <= 64 channels (X and dX, hi and lo: 128 KB) resident in LDS and streams the weights.
// A wave accumulates the TWO x parities of 32 positions side by side (the two accumulator tiles that conv_h3_kernel
// gives to two column tiles): voxels 2x and 2x + 1 of an output row are then stored by the same wave within one
// epilogue, and the row leaves L2 as whole lines.  (Parity after parity the second half of every line arrived ~100 us
// after the first, long after L2 had turned over: WRITE_SIZE 1.5 x the algorithmic bytes, and the kernel ran at the
// write bandwidth that pattern allows -- tools/micro/write_bw.hip: 4.7 TB/s for 32 interleaved plane streams.)
// Stage = (position half, parity pair, chunk): weights of both parities, 16 KB, double-buffered.
// SPLIT = false: the float16 model (one plane per eight channels, one MFMA per product): half the LDS image, two workgroups per
// CU, a third of the MFMAs -- instead of eight launches of the general 1x1x1 kernel that each read the whole input.
constexpr int UP_XV = 256;                           // positions per workgroup
constexpr int UP_MAXCH = 4;                          // Cin <= 64
template <bool SPLIT> struct UpGeom {
    static constexpr int UN = SPLIT ? 4 : 2;             // 16-byte units per position and 16-channel chunk (2 h + part | h)
    static constexpr int XC = 2 * UN * UP_XV;            // units of one resident chunk: (X, dX) x UN units x 256
    static constexpr int WS = 2 * 2 * UN * 64;           // units of one weight stage: 2 parities x (W, dW) x UN units x 64 couts
    static constexpr int WBASE = UP_MAXCH * XC;
    static constexpr int LDS_UNITS = WBASE + 2 * WS;     // f16x3: 10240 units = 163,840 B, all of the CU's LDS; float16: half
};

// VEL = false: the displacement-only models (no tangent tensors, no dW): the same stages with the primal's products only.
template <bool SPLIT, bool VEL = true>
__global__ __launch_bounds__(512, 2) void up_h3_kernel(ConvKArgs a) {
    constexpr int UN = UpGeom<SPLIT>::UN, UP_XC = UpGeom<SPLIT>::XC, UP_WS = UpGeom<SPLIT>::WS, UP_WBASE = UpGeom<SPLIT>::WBASE;
    constexpr int PS = SPLIT ? 2 : 1;                        // planes per channel half: hi, lo | one
    f32x4* lds = lds_h3;
    const half8* L8 = (const half8*)lds_h3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int it = wave & 1, jq = wave >> 1;                 // 32 couts, 32 of the 128 positions of a half
    const int tile = xcd_tile(blockIdx.x, a.ntiles);
    const int ct = blockIdx.y;
    const long q0 = (long)tile * UP_XV;
    const int nchunk = a.nchunk, nstage = 2 * 4 * nchunk;    // stage = (half * 4 + pair) * nchunk + chunk
    const long HW = (long)a.H * a.W;

    // ---- resident activations: chunk c, tensor t, unit u (2 h + part), position v: lds[c * UP_XC + (t * 4 + u) * 256 + v]
    // 32 wave-instructions per chunk, 4 per wave
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
        for (int k = 0; k < (VEL ? UN : UN / 2); ++k) {
            const int n = wave + 8 * k;                       // 0..8 UN - 1: tensor, unit, quarter (n & 3)  (!VEL: X only)
            const int t = n / (4 * UN), u = (n >> 2) % UN, qd = n & 3;
            const long v = q0 + a.in_off + qd * 64 + lane;
            const char* src = (const char*)(t ? a.dx : a.x) + (((long)c * UN + u) * a.in_pstride + v) * 16;
            dma16((const float*)src, lds + c * UP_XC + (t * UN + u) * UP_XV + qd * 64);
        }
    }
    // ---- weights of stage st: parities 2 pp and 2 pp + 1 of chunk c: [parity][W | dW][unit 4][64 couts] = 16 wave-instructions
    auto dma_w = [&](int st) {
        const int c = st % nchunk, pp = (st / nchunk) & 3;
#pragma unroll
        for (int k = 0; k < UN / 2; ++k) {
            const int n = wave + 8 * k;                       // 0..4 UN - 1: parity, set (W | dW), unit
            const int par = n / (2 * UN), d = (n / UN) & 1, m = n % UN;
            if (!VEL && d) continue;                          // (no dW: its rows of the stage stay unused)
            const char* src = (const char*)(d ? a.dw : a.w) + (long)(2 * pp + par) * a.set_stride +
                              (((long)ct * nchunk + c) * (UN * 64) + m * 64 + lane) * 16;
            dma16((const float*)src, lds + UP_WBASE + (st & 1) * UP_WS + n * 64);
        }
    };
    dma_w(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 ym[2], yc[2], dm[2], dc[2];                       // [x parity]
    auto zero_acc = [&]() {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { ym[jt][e] = 0.f; yc[jt][e] = 0.f; dm[jt][e] = 0.f; dc[jt][e] = 0.f; }
    };
    zero_acc();

    for (int st = 0; st < nstage; ++st) {
        const int c = st % nchunk, pp = (st / nchunk) & 3, half = st / (4 * nchunk);
        if (st + 1 < nstage) dma_w(st + 1);
        const half8* wb = L8 + UP_WBASE + (st & 1) * UP_WS;
        const half8* xb = L8 + c * UP_XC;
        const int xo = (PS * lh) * UP_XV + half * 128 + jq * 32 + li;
        const half8 xh = xb[xo];
        half8 dxh = xh, xl = xh, dxl = xh;
        if (VEL) dxh = xb[UN * UP_XV + xo];
        if (SPLIT) { xl = xb[xo + UP_XV]; if (VEL) dxl = xb[UN * UP_XV + xo + UP_XV]; }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {                      // x parity jt: its own weights, the same activations
            const int wo = jt * (2 * UN * 64) + (PS * lh) * 64 + 32 * it + li;
            const half8 wh = wb[wo];
            half8 dwh = wh;
            if (VEL) dwh = wb[UN * 64 + wo];
            if (!SPLIT) {                                    // float16: y += w.x, dy += dw.x + w.dx
                ym[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, ym[jt], 0, 0, 0);
                if (VEL) {
                    dm[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dwh, xh, dm[jt], 0, 0, 0);
                    dm[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, dxh, dm[jt], 0, 0, 0);
                }
                continue;
            }
            const half8 wl = wb[wo + 64];
            ym[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, ym[jt], 0, 0, 0);
            yc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, yc[jt], 0, 0, 0);
            yc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, yc[jt], 0, 0, 0);
            if (!VEL) continue;
            const half8 dwl = wb[UN * 64 + wo + 64];
            dm[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dwh, xh, dm[jt], 0, 0, 0);
            dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dwh, xl, dc[jt], 0, 0, 0);
            dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dwl, xh, dc[jt], 0, 0, 0);
            dm[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, dxh, dm[jt], 0, 0, 0);
            dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, dxl, dc[jt], 0, 0, 0);
            dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, dxh, dc[jt], 0, 0, 0);
        }
        const bool epi = c == nchunk - 1;
        if (epi) {                                           // the parity pair (oz, oy) = pp of this half is complete
            // W(st + 1) has had the whole stage to land: wait for it BEFORE the stores, so that the barrier below does not
            // have to drain them (vmcnt counts loads and stores alike; eight such drains per workgroup otherwise)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long q = q0 + half * 128 + jq * 32 + li;
            const int z = (int)(q / HW), rem = (int)(q - (long)z * HW);
            const int yy = rem / a.W, xx = rem - yy * a.W;
            const bool okq = q < a.Q && xx < a.Wv && yy < a.Hv && z < a.Dv;
            const long o0 = okq ? ((long)(2 * z + (pp >> 1)) * a.Ho + (2 * yy + (pp & 1))) * a.Wo + 2 * xx : 0;
            const long o[2] = {o0, o0 + (okq ? 1 : 0)};
            const bool ok[2] = {okq, okq};
            h3_store2<VEL, SPLIT>(a, ct, it, lh, o, ok, ym, yc, dm, dc);
            zero_acc();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // W(st + 1) has landed
            __syncthreads();
        }
    }
}

template <bool SPLIT, bool VEL = true>
static int launch_up_h3(ConvKArgs ka, int ctiles, hipStream_t s) {
    constexpr size_t smem = (size_t)UpGeom<SPLIT>::LDS_UNITS * 16;
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    if (ka.nchunk > UP_MAXCH) return 1;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)up_h3_kernel<SPLIT, VEL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    dim3 grid(ka.ntiles, ctiles, 1), block(512, 1, 1);
    hipLaunchKernelGGL((up_h3_kernel<SPLIT, VEL>), grid, block, smem, s, ka);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// 3x3x3 convolution on 2-D patches (the production kernel for MODE_FLAT3 layers)
// ------------------------------------------------------------------------------------------------
// The flat tiling above re-fetches every activation row segment once per (dz,dy) pair: 95 B of activation DMA
// per MFMA on top of 57 B of weights, and the kernel is bound by L2->LDS throughput.  Here a workgroup owns an
// 8-row x 32-column patch of ONE output plane; for each 16-channel chunk and each dz it stages the 10 x 34
// input patch once and serves all nine (dy,dx) taps from it (LDS address = row*34 + col, taps are address
// shifts).  Weights stream exactly as before: one 24.6 KB stage per (chunk, dz, dy), same packed layout.
// DMA per MFMA drops to ~34 B (activations) + 57 B (weights).  Tiles are ordered z-fastest so that the 32
// workgroups of an XCD work on neighbouring planes of the same (y,x) patch and share two of their three input
// planes through that XCD's L2.
constexpr int HP_ROWS = 8, HP_COLS = 32;
constexpr int HP_RS = HP_COLS + 2;                   // LDS row stride (units)
constexpr int HP_PL = (HP_ROWS + 2) * HP_RS;         // units per plane of the patch image: 340

// SCHED selects how the DMA of one stage is issued (A/B on one device: env NBE_H3_SCHED):
//   0  burst: W(s+1) and a third of X(g+1) right after the barrier, interleaved with the first tap's MFMAs;
//      every stage ends with vmcnt(0) (__syncthreads).
//   1  de-bursted: W(s+1) in the first tap; X(g+1) in halves during the dy = 0 and dy = 1 stages, issued in the
//      second/third tap and left in flight across the barrier with a counted vmcnt (they are needed only when the
//      (chunk,dz) group changes).
// Measured (same device, 512^3 bench, TFLOP/s-equivalent of this kernel): SCHED 0 = 354, SCHED 1 = 345.
// Debug-build probes (NBE_BUILD_DBG=1, NBE_DEBUG_FLAGS): no DMA at all 482; weight DMA only 481; activation DMA
// only 475; both 352 -- any mix of the two streams costs 20-30 % whatever its size (2W+1X slots: 394, 3W+1X: 385,
// 2W+2X: 387), and neither de-bursting (SCHED 1), nor leaving the DMA in flight across the barrier, nor a deeper
// ring, nor 40 % fewer bytes (this kernel vs the flat one) changes that.  Explanation (profiles/
// r01_clock_vs_dma_conv_h3p.txt): in CYCLES all variants are the same kernel (matrix pipe busy 64.6-66.1 % of SIMD
// cycles); the chip holds 2.35-2.38 GHz without the combined DMA streams and 1.81 GHz with them.  The kernel is
// power-limited, so what pays is less energy per MFMA (bytes, LDS reads), not a tighter issue stream.
template <bool VEL, bool HAS_DX, int SCHED, bool SPLIT>
__global__ __launch_bounds__(512, 2) void conv_h3p_kernel(ConvKArgs a) {
    constexpr bool DX = VEL && HAS_DX;
    constexpr int PARTS = SPLIT ? 2 : 1, UN = 2 * PARTS;         // units of 8 channels per 16-channel chunk
    constexpr int HP_XP = UN * HP_PL;                            // valid units of one tensor's patch: 1360 (680)
    constexpr int HP_XPP = (HP_XP + 63) / 64 * 64;               // padded to whole wave-instructions: 1408 (704)
    constexpr int WP = 3 * UN * 64;
    constexpr int WB = WP * (VEL ? 2 : 1);
    constexpr int XB = HP_XPP * (DX ? 2 : 1);
    constexpr int OFF_DW = WP, OFF_DXX = HP_XPP;
    constexpr int XBASE = 2 * WB;
    constexpr int NW_TOT = WB / 64, NX_TOT = XB / 64;            // 24 and 44 wave-instructions
    constexpr int NWS = (NW_TOT + 7) / 8;                        // weight slots per wave and stage
    constexpr int XPARTS = SCHED == 0 ? 3 : 2;                   // stages of a group that issue activation DMA
    constexpr int NXS = (NX_TOT + 8 * XPARTS - 1) / (8 * XPARTS);   // activation slots per wave in such a stage
    static_assert(NWS + NXS <= 9 && NXS <= 4, "DMA slots per wave");

    f32x4* lds = lds_h3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int it = wave & 1, jq = wave >> 1;

    // tile -> (ty, tx, z), z fastest
    const int tile = xcd_tile(blockIdx.x, a.ntiles);
    const int ct = blockIdx.y;
    const int z = tile % a.Dv, tyx = tile / a.Dv;
    const int ty = tyx / a.tnx, tx = tyx - ty * a.tnx;
    const int y0 = ty * HP_ROWS, x0 = tx * HP_COLS;
    const int nstage = 9 * a.nchunk;

    // ---- DMA slots.  Weights: instruction n = wave + 8t of every stage.  Activations: the NX_TOT instructions of
    // the NEXT (chunk,dz) patch are spread over XPARTS stages of the current one: instruction
    // n = (p*NXS + t)*8 + wave.  Per-lane source = patch-relative voxel (row*W + col) in plane pl; per patch a
    // wave-uniform offset (chunk planes + patch origin) is added.
    const char* wsrc[NWS];
    int wdst[NWS];
    const char* xsrc[XPARTS][NXS];
    int xdst[XPARTS][NXS];
    int kx[XPARTS];                                              // activation instructions of this wave per part
#pragma unroll
    for (int t = 0; t < NWS; ++t) {
        const int n = wave + 8 * t;
        wsrc[t] = nullptr; wdst[t] = 0;
        if (n < NW_TOT) {
            const bool d = VEL && n >= WP / 64;
            const int m = n - (d ? WP / 64 : 0);
            wsrc[t] = (const char*)(d ? a.dw : a.w) + ((long)ct * nstage * WP + m * 64 + lane) * 16;
            wdst[t] = (d ? OFF_DW : 0) + m * 64;
        }
    }
#pragma unroll
    for (int p = 0; p < XPARTS; ++p) {
        kx[p] = 0;
#pragma unroll
        for (int t = 0; t < NXS; ++t) {
            const int n = (p * NXS + t) * 8 + wave;
            xsrc[p][t] = nullptr; xdst[p][t] = 0;
            if (n < NX_TOT) {
                ++kx[p];
                const int u = n * 64 + lane;
                const bool tang = DX && u >= HP_XPP;
                int uu = u - (tang ? HP_XPP : 0);
                const int ud = uu;                               // LDS position (padding lanes land in the pad)
                if (uu >= HP_XP) uu = HP_XP - 1;
                const int pl = uu / HP_PL, rem = uu - pl * HP_PL;
                const int row = rem / HP_RS, col = rem - row * HP_RS;
                xsrc[p][t] = (const char*)(tang ? a.dx : a.x) + ((long)pl * a.in_pstride + (long)row * a.W + col) * 16;
                xdst[p][t] = __builtin_amdgcn_readfirstlane((tang ? OFF_DXX : 0) + ud);
            }
        }
    }
    auto patch_offset = [&](int g) -> long {                     // g = chunk*3 + dz
        const int chunk = g / 3, dz = g - chunk * 3;
        return ((long)chunk * UN * a.in_pstride + ((long)(z + dz) * a.H + y0) * a.W + x0) * 16;
    };
    // NBE_DBG builds only (timing experiments, results invalid): flags bit 8 = no weight DMA after the prologue,
    // bit 9 = no activation DMA after the prologue, bit 10 / 11 = drop the third weight slot / the second activation slot
    const bool dbg_now = NBE_DBG && (a.flags & 256), dbg_nox = NBE_DBG && (a.flags & 512);
    const bool dbg_w2 = NBE_DBG && (a.flags & 1024), dbg_x1 = NBE_DBG && (a.flags & 2048);   // drop one slot each
    bool dbg_pro = true;
    auto dma_w = [&](int t, int s) {
        if (NBE_DBG && (dbg_now || (dbg_w2 && t == 2)) && !dbg_pro) return;
        if (wave + 8 * t < NW_TOT) dma16((const float*)(wsrc[t] + (long)s * WP * 16), lds + (s & 1) * WB + wdst[t]);
    };
    auto dma_x = [&](int p, int t, long xoff, int xb) {
        if (NBE_DBG && (dbg_nox || (dbg_x1 && t == 1)) && !dbg_pro) return;
        if ((p * NXS + t) * 8 + wave < NX_TOT) dma16((const float*)(xsrc[p][t] + xoff), lds + XBASE + xb * XB + xdst[p][t]);
    };
    // wait until at most k of this wave's DMA instructions are outstanding (vmcnt counts in issue order)
    auto wait_keep = [&](int k) {
        if (k <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (k == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (k == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (k == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };

    f32x16 ym[2], yc[2], dm[2], dc[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { ym[jt][e] = 0.f; yc[jt][e] = 0.f; dm[jt][e] = 0.f; dc[jt][e] = 0.f; }

    struct Ops { half8 wh, wl, dwh, dwl, xh[2], xl[2], dxh[2], dxl[2]; };
    // operands of tap (dy, dx): weights of the current stage, activations from the resident patch
    auto load_ops = [&](const half8* wb, const half8* xb, int dy, int dx, Ops& o) {
        const int wo = (dx * UN + PARTS * lh) * 64 + 32 * it + li;
        o.wh = wb[wo];
        if (SPLIT) o.wl = wb[wo + 64];
        if (VEL) { o.dwh = wb[OFF_DW + wo]; if (SPLIT) o.dwl = wb[OFF_DW + wo + 64]; }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int xo = (PARTS * lh) * HP_PL + (2 * jq + jt + dy) * HP_RS + li + dx;
            o.xh[jt] = xb[xo];
            if (SPLIT) o.xl[jt] = xb[xo + HP_PL];
            if (DX) { o.dxh[jt] = xb[OFF_DXX + xo]; if (SPLIT) o.dxl[jt] = xb[OFF_DXX + xo + HP_PL]; }
        }
    };
    auto mfma1 = [&](const Ops& o, int i) {
        const int jt = i / 9, k = i % 9;
#if NBE_DBG_SHAPE16
        // timing probe only (results invalid): the same operand registers fed to two 16x16x32 MFMAs per 32x32x16 one
        // (same cycles, same LDS/DMA traffic) -- measures what the MFMA shape alone does to the clock the chip holds
        auto two = [&](const half8& A, const half8& B, f32x16& acc) {
            f32x4 c0 = {acc[0], acc[1], acc[2], acc[3]}, c1 = {acc[4], acc[5], acc[6], acc[7]};
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c1, 0, 0, 0);
            acc[0] = c0[0]; acc[1] = c0[1]; acc[2] = c0[2]; acc[3] = c0[3];
            acc[4] = c1[0]; acc[5] = c1[1]; acc[6] = c1[2]; acc[7] = c1[3];
        };
        if (k == 0) two(o.wh, o.xh[jt], ym[jt]);
        if (SPLIT && k == 1) two(o.wh, o.xl[jt], yc[jt]);
        if (SPLIT && k == 2) two(o.wl, o.xh[jt], yc[jt]);
        if (VEL) {
            if (k == 3) two(o.dwh, o.xh[jt], dm[jt]);
            if (SPLIT && k == 4) two(o.dwh, o.xl[jt], dc[jt]);
            if (SPLIT && k == 5) two(o.dwl, o.xh[jt], dc[jt]);
        }
        if (DX) {
            if (k == 6) two(o.wh, o.dxh[jt], dm[jt]);
            if (SPLIT && k == 7) two(o.wh, o.dxl[jt], dc[jt]);
            if (SPLIT && k == 8) two(o.wl, o.dxh[jt], dc[jt]);
        }
        return;
#endif
        if (k == 0) ym[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wh, o.xh[jt], ym[jt], 0, 0, 0);
        if (SPLIT && k == 1) yc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wh, o.xl[jt], yc[jt], 0, 0, 0);
        if (SPLIT && k == 2) yc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wl, o.xh[jt], yc[jt], 0, 0, 0);
        if (VEL) {
            if (k == 3) dm[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.dwh, o.xh[jt], dm[jt], 0, 0, 0);
            if (SPLIT && k == 4) dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.dwh, o.xl[jt], dc[jt], 0, 0, 0);
            if (SPLIT && k == 5) dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.dwl, o.xh[jt], dc[jt], 0, 0, 0);
        }
        if (DX) {
            if (k == 6) dm[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wh, o.dxh[jt], dm[jt], 0, 0, 0);
            if (SPLIT && k == 7) dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wh, o.dxl[jt], dc[jt], 0, 0, 0);
            if (SPLIT && k == 8) dc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wl, o.dxh[jt], dc[jt], 0, 0, 0);
        }
    };
    auto mfma_ops = [&](const Ops& o) {
#pragma unroll
        for (int i = 0; i < 18; ++i) mfma1(o, i);
    };

    // ---- prologue: patch of group 0 (all its instructions) and weights of stage 0
    {
        const long x0off = patch_offset(0);
#pragma unroll
        for (int p = 0; p < XPARTS; ++p)
#pragma unroll
            for (int t = 0; t < NXS; ++t) dma_x(p, t, x0off, 0);
#pragma unroll
        for (int t = 0; t < NWS; ++t) dma_w(t, 0);
        __syncthreads();
    }

    dbg_pro = false;
    int g = 0, dy = 0;                                           // stage s = 3*g + dy
    for (int s = 0; s < nstage; ++s) {
        const bool pw = s + 1 < nstage, px = 3 * (g + 1) < nstage;
        const long xoff = px ? patch_offset(g + 1) : 0;
        const half8* wb = (const half8*)(lds + (s & 1) * WB);
        const half8* xb = (const half8*)(lds + XBASE + (g & 1) * XB);
        const int nb = (g + 1) & 1;
        Ops o0, o1;
        load_ops(wb, xb, dy, 0, o0);
        load_ops(wb, xb, dy, 1, o1);
        __builtin_amdgcn_sched_barrier(0);
        // first tap: one DMA instruction after every second MFMA -- W(s+1) [SCHED 0: then a third of X(g+1)]
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            mfma1(o0, i);
            if (i & 1) {
                const int t = i >> 1;
                if (t < NWS) { if (pw) dma_w(t, s + 1); }
                else if (SCHED == 0 && t - NWS < NXS) {
                    if (px) {
                        if (dy == 0) dma_x(0, t - NWS, xoff, nb);
                        else if (dy == 1) dma_x(1, t - NWS, xoff, nb);
                        else dma_x(XPARTS - 1, t - NWS, xoff, nb);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_ops(wb, xb, dy, 2, o0);
        __builtin_amdgcn_sched_barrier(0);
        if (SCHED == 0) {
            mfma_ops(o1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_ops(o0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                     // vmcnt(0): W(s+1) and this stage's X pieces landed
        } else {
            // second and third tap: this stage's half of X(g+1), one instruction every sixth MFMA
            const bool xs = px && dy < 2;
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                if (i < 18) mfma1(o1, i); else mfma1(o0, i - 18);
                if (i % 6 == 2 && i / 6 < NXS) {
                    if (xs) { if (dy == 0) dma_x(0, i / 6, xoff, nb); else dma_x(XPARTS - 1, i / 6, xoff, nb); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // W(s+1) must have landed; the X pieces issued in this stage may stay in flight unless the group ends
            wait_keep(xs ? (dy == 0 ? kx[0] : kx[XPARTS - 1]) : 0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (++dy == 3) { dy = 0; ++g; }
    }

    // ---- epilogue: rows y0 + 2*jq + jt, column x0 + li of output plane z
    long o[2];
    bool ok[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int yy = y0 + 2 * jq + jt, xx = x0 + li;
        ok[jt] = yy < a.Hv && xx < a.Wv;
        o[jt] = ok[jt] ? ((long)z * a.Ho + yy) * a.Wo + xx : (long)z * a.Ho * a.Wo;
    }
    h3_store2<VEL, SPLIT>(a, ct, it, lh, o, ok, ym, yc, dm, dc);
}

// ------------------------------------------------------------------------------------------------
// The same 2-D patch convolution on the 16x16x32 MFMA shape (f16x3 with velocity and input tangent: the
// production kernel of the 3x3x3 layers with Cin >= 16)
// ------------------------------------------------------------------------------------------------
// Why: this loop is power-limited (see above), and the chip holds a higher clock on v_mfma_f32_16x16x32_f16 than
// on 32x32x16 at equal cycles per FLOP (MI355X guide, DVFS give-back item 7).  A timing probe that fed the 32x32x16
// kernel's operand registers to pairs of 16x16x32 MFMAs (build -DNBE_DBG_SHAPE16=1, same LDS and DMA traffic) ran
// the 512^3 bench 14.5 % faster on the same device (413 vs 361 TFLOP/s-equivalent).
//
// K = 32 is TWO TAPS x 16 channels: lane group q = lane >> 4 supplies channels 8*(q&1).. of tap (q>>1) of the pair,
// for A (weights, row = cout) and B (activations, column = position) alike -- per-lane LDS addresses make that free,
// and the nine products per tile and their operand reuse are exactly those of the 32x32x16 kernel.  The nine taps
// of a (chunk, dz) group are four pairs and one single, (0,1) (2,3) [4] (5,6) (7,8) with tap = 3*dy + dx.  The single
// tap pairs PARTS instead of taps: yc += [wh|wl].[xl|xh], dc += [dwh|dwl].[xl|xh] + [wh|wl].[dxl|dxh],
// dm += [dwh|wh].[xh|dxh], ym += [0|wh].[xl|xh]: 5 MFMAs per tile where 4.5 would be ideal (1.2 % of a group).
// A group is two stages, taps 0-4 (weight buffer A, 40 KB) and taps 5-8 (buffer B, 32 KB): two barriers per group
// instead of three, and the packed weight layout is unchanged (the nine taps of a group are contiguous).
// Wave tile 32 couts x 64 positions = 2 x 4 MFMA tiles, tile t = 4*mt + nt, nt = 2*jt + nh (row jt, column half nh).
constexpr int HQ_TAPU = 4 * 64;                           // 16-byte units per tap and set
constexpr int HQ_WA = 5 * HQ_TAPU, HQ_WB = 4 * HQ_TAPU;   // units per set in weight buffer A (taps 0-4) / B (taps 5-8)
constexpr int HQ_OFF_B = 2 * HQ_WA;                       // buffer B follows buffer A
// LDS pitch of a patch plane: 340 valid units padded to a multiple of 8, so that the planes 2*kh a ds_read_b128 mixes in
// one LDS cycle (lanes of two 16-lane rows) start 0 mod 16 units apart and hit 16 distinct bank quads.  With 340 the
// rows were 8 mod 16 apart and every activation read was a 2-way bank conflict (SQ_LDS_BANK_CONFLICT = 40 % of
// SQ_LDS_IDX_ACTIVE).
constexpr int HQ_PP = (HP_PL + 7) / 8 * 8;                // 344
constexpr int HQ_XT = 4 * HQ_PP;                          // units of one tensor's patch (hi/lo x two channel halves)
constexpr int HQ_XB = 2 * HQ_XT;                          // one patch buffer: X, dX
constexpr int HQ_XBASE = HQ_OFF_B + 2 * HQ_WB;
constexpr int HQ_LDS_UNITS = HQ_XBASE + 2 * HQ_XB;        // 10048 units = 160,768 B

// read one accumulator element where it is used (operand constraint "a": it stays in its AGPR until then)
__device__ __forceinline__ float acc_read(const float& acc) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc));
    return v;
}

// global -> LDS DMA with a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit per-lane pointers
__device__ __forceinline__ void dma16s(const char* ubase, unsigned voff, f32x4* dst_wave_base) {
    // pin the whole uniform address in SGPRs: otherwise its loop-invariant part is folded into a per-lane pointer
    const unsigned long ub = (unsigned long)ubase;
    ubase = (const char*)(((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ub >> 32)) << 32) |
                          (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ub));
    asm volatile("" : "+v"(voff));          // keep the zero-extension next to the load (saddr + 32-bit voffset form)
    __builtin_amdgcn_global_load_lds((const NBE_GLB_AS void*)(ubase + voff), (NBE_LDS_AS void*)dst_wave_base, 16, 0, 0);
}

// NBE_DBG builds: per-wave cycle totals (s_memtime) of the phases of conv_h3q_kernel, stamped only where the wave
// drains its counters anyway (around the barriers): 0 prologue; 1 / 4 compute of the first / second stage of a
// group, 2 / 5 the wait for its own DMA (vmcnt) and 3 / 6 the wait at the barrier that ends the stage; 7 epilogue;
// 8 waves.  Written to memory nothing else reads.
__device__ unsigned long long h3q_stamps[16];

void h3q_read_stamps(double* out, hipStream_t s) {
    unsigned long long h[16] = {0};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(h3q_stamps), sizeof h, 0, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) out[i] = (double)h[i];
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(h3q_stamps), z, sizeof z, 0, hipMemcpyHostToDevice);
}

__global__ __launch_bounds__(512, 2) void conv_h3q_kernel(ConvKArgs a) {
#if NBE_DBG
    unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define NBE_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tk[i] += t_ - tk0; tk0 = t_; }
// the barrier that ends a stage, its two waits stamped separately (i: compute, i+1: own DMA, i+2: other waves)
#define NBE_STAGE_END(i, keep) { NBE_STAMP(i) wait_keep(keep); NBE_STAMP(i + 1) \
                                 asm volatile("s_barrier" ::: "memory"); NBE_STAMP(i + 2) }
#else
#define NBE_STAMP(i)
#define NBE_STAGE_END(i, keep) { wait_keep(keep); asm volatile("s_barrier" ::: "memory"); }
#endif
    f32x4* lds = lds_h3;
    const half8* L8 = (const half8*)lds_h3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4, kh = q & 1, ks = q >> 1;
    const int it = wave & 1, jq = wave >> 1;

    // tile -> (ty, tx, z), z fastest: the 32 workgroups an XCD runs at the same time work on neighbouring planes of one
    // patch column and share two of their three input planes through that XCD's L2.  (Persistent workgroups were
    // measured on one device: a run of planes per workgroup loses that sharing, -12 %; a run of patches along x keeps
    // it but gains < 1 % over the same code with runs of one, and carrying the epilogue inside a loop costs 6 %.)
    const int nct = (a.cout_groups + 7) / 8;                     // cout tiles of one patch side by side (see conv_h3g_kernel)
    const int vt = xcd_tile(blockIdx.x, a.ntiles * nct);
    const int tile = vt / nct, ct = vt - tile * nct;
    const int z = tile % a.Dv, tyx = tile / a.Dv;
    const int ty = tyx / a.tnx, tx = tyx - ty * a.tnx;
    const int y0 = ty * HP_ROWS, x0 = tx * HP_COLS;
    const int ngroups = 3 * a.nchunk;

    const unsigned lane16 = (unsigned)lane * 16u;
    // ---- DMA.  Weights: wave-instruction n = wave + 8t of a stage (set w first, then dw).  Activations: the x
    // patch of group g+1 during the first stage of group g, the dx patch during the second.
    const long wct = (long)ct * ngroups * 9 * HQ_TAPU;
    auto dma_w = [&](int g, int second, int t) {
        const int per = (second ? HQ_WB : HQ_WA) / 64;
        const int n = wave + 8 * t;
        if (n >= 2 * per) return;
        const int set = n >= per ? 1 : 0, m = n - set * per;
        const long src = wct + (long)g * 9 * HQ_TAPU + (second ? 5 * HQ_TAPU : 0) + m * 64;
        dma16s((const char*)(set ? a.dw : a.w) + src * 16, lane16,
               lds + (second ? HQ_OFF_B : 0) + set * (second ? HQ_WB : HQ_WA) + m * 64);
    };
    // Patch DMA: 6 wave-instructions per plane of 340 units (the 6th covers 20 lanes), 24 per tensor, 3 per wave;
    // the plane goes into the uniform base, the per-lane offset is the position inside the plane (32 bits).
    unsigned xoff[3];
    bool xval[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int k = (wave + 8 * t) % 6;
        const int u = k * 64 + lane;
        xval[t] = u < HP_PL;
        const int uu = xval[t] ? u : HP_PL - 1;
        const int row = uu / HP_RS, col = uu - row * HP_RS;
        xoff[t] = (unsigned)(row * a.W + col) * 16u;
    }
    auto patch_offset = [&](int g) -> long {                     // g = chunk*3 + dz
        const int chunk = g / 3, dz = g - chunk * 3;
        return ((long)chunk * 4 * a.in_pstride + ((long)(z + dz) * a.H + y0) * a.W + x0) * 16;
    };
    auto dma_x = [&](int tensor, int t, long xo, int buf) {
        const int n = wave + 8 * t, pl = n / 6, k = n - 6 * pl;
        if (xval[t])
            dma16s((const char*)(tensor ? a.dx : a.x) + xo + (long)pl * a.in_pstride * 16, xoff[t],
                   lds + HQ_XBASE + buf * HQ_XB + tensor * HQ_XT + pl * HQ_PP + k * 64);
    };

    f32x4 ym[8], yc[8], dm[8], dc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ym[t][e] = 0.f; yc[t][e] = 0.f; dm[t][e] = 0.f; dc[t][e] = 0.f; }
    // The 32 accumulator tiles live in AGPRs and are updated in place: 128 AGPRs + at most 128 VGPRs for operands and
    // addresses.  (Left to the register allocator the 4-register tiles wander through one 256-register file and the
    // kernel spills; a scratch reload in the loop waits on vmcnt(0) and with it on every DMA in flight.)  Dependent
    // MFMAs on one tile are always >= 8 MFMAs apart.
    auto mm = [&](f32x4& acc, const half8& A, const half8& B) {
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
    };
    // the same behind a VALU select: the compiler does not see an MFMA inside an asm statement and leaves out the two wait
    // states the ISA wants between a VALU write of a VGPR and an MFMA that reads it (tools/check_mfma_hazards.py)
    auto mmz = [&](f32x4& acc, const half8& A, const half8& B) {
        asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
    };

    // ---- operands.  A (2 MFMA row tiles): unit (tap, 2*kh + part) of the stage's weight buffer, row 32*it + 16*mt + c.
    // B (4 column tiles): patch plane 2*kh + part, position (2*jq + jt, 16*nh + c) shifted by this lane group's tap.
    const int aP = (ks * 4 + 2 * kh) * 64 + 32 * it + c;
    const int bB = (2 * kh) * HQ_PP + (2 * jq) * HP_RS + c;
    const int bP1 = bB + ks, bP32 = bB + 32 * ks;               // second tap of the pair: one column / 32 units further
    auto LA = [&](half8 (&r)[2], int idx) {
        r[0] = L8[idx];
        r[1] = L8[idx + 16];
    };
    auto LB = [&](half8 (&r)[4], int idx) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) r[nt] = L8[idx + (nt >> 1) * HP_RS + 16 * (nt & 1)];
    };
    // one product on the wave tile: 8 MFMAs.  kind 1 / 2 (first / second stage of a group) issues the DMA slots
    // `slot` and `slot + 1` after the 4th and the 8th MFMA: slots 0-4 weights of the next stage, 5-7 the x / dx
    // patch of the next group.  (Issuing both patches in the first stage and leaving them in flight across its
    // barrier with a counted vmcnt was measured on one device: dx only -0.6 %, x and dx -2.5 %.)
    auto MM8 = [&](f32x4 (&acc)[8], const half8 (&A)[2], const half8 (&B)[4], int kind, int slot, int g, int gn,
                   long xo, int nb, bool px, bool zsel = false /* A comes out of a VALU select */) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (zsel && (t & 3) == 0) mmz(acc[t], A[t >> 2], B[t & 3]); else
            mm(acc[t], A[t >> 2], B[t & 3]);
            if (kind != 0 && (t & 3) == 3) {
                const int k = slot + (t >> 2);
                if (k < 5) { if (kind == 1) dma_w(g, 1, k); else if (px) dma_w(gn + 1, 0, k); }
                else if (k < 8 && px) dma_x(kind == 1 ? 0 : 1, k - 5, xo, nb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#define NBE_SB __builtin_amdgcn_sched_barrier(0)
    half8 wh[2], wl[2], dwh[2], dwl[2], xh[4], xl[4], dxh[4], dxl[4];
    // A tap pair: nine products in an order that keeps at most 80 operand registers alive, every operand requested
    // two or three products (256-384 cycles) before its first use.  On entry dwh, xl and xh of the pair are loaded
    // (or in flight); pre6 / pre7 request those of whatever follows.
    auto pair = [&](int kind, int g, int gn, long xo, int nb, bool px, int wa /* buffer + tap offset */, int wset,
                    int xp /* patch + shift + lane base */, auto&& pre6, auto&& pre7) {
        LA(wh, wa + aP); LA(dwl, wa + wset + 64 + aP);
        NBE_SB; MM8(dc, dwh, xl, kind, 0, g, gn, xo, nb, px); NBE_SB;
        LA(wl, wa + 64 + aP);
        NBE_SB; MM8(dm, dwh, xh, kind, 2, g, gn, xo, nb, px); NBE_SB;
        MM8(yc, wh, xl, kind, 4, g, gn, xo, nb, px); NBE_SB;
        LB(dxh, xp + HQ_XT);
        NBE_SB; MM8(dc, dwl, xh, kind, 6, g, gn, xo, nb, px); NBE_SB;
        MM8(yc, wl, xh, 0, 0, g, gn, xo, nb, px); NBE_SB;
        LB(dxl, xp + HQ_XT + HQ_PP);
        NBE_SB; MM8(ym, wh, xh, 0, 0, g, gn, xo, nb, px); NBE_SB;
        pre6();
        NBE_SB; MM8(dc, wl, dxh, 0, 0, g, gn, xo, nb, px); NBE_SB;
        pre7();
        NBE_SB; MM8(dm, wh, dxh, 0, 0, g, gn, xo, nb, px); NBE_SB;
        MM8(dc, wh, dxl, 0, 0, g, gn, xo, nb, px); NBE_SB;
    };

    // wait until at most `keep` of this wave's DMA instructions are outstanding (vmcnt counts in issue order), and for
    // its LDS reads
    auto wait_keep = [&](int keep) {
        if (keep >= 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else if (keep == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };

    // ---- prologue: both patches of group 0 and the weights of its first stage
    {
        const long x0off = patch_offset(0);
#pragma unroll
        for (int t = 0; t < 3; ++t) { dma_x(0, t, x0off, 0); dma_x(1, t, x0off, 0); }
#pragma unroll
        for (int t = 0; t < 5; ++t) dma_w(0, 0, t);
        __syncthreads();
        NBE_STAMP(0)
        LB(xl, HQ_XBASE + bP1 + HQ_PP);
        LB(xh, HQ_XBASE + bP1);
    }

    constexpr int SH4 = HP_RS + 1, SH5 = HP_RS + 2, SH7 = 2 * HP_RS + 1;   // tap shifts: 3*dy + dx -> dy*34 + dx
    for (int g = 0; g < ngroups; ++g) {
        const bool px = g + 1 < ngroups;
        const int gn = g;                                        // the second stage loads the weights of group gn + 1
        const long xo = px ? patch_offset(g + 1) : 0;
        const int nb = (g + 1) & 1;
        const int xb = HQ_XBASE + (g & 1) * HQ_XB, xbn = HQ_XBASE + nb * HQ_XB;
        half8 a1w[2], a1d[2], a2[2], a0[2], b1x[4], b1d[4], b2[4];
        // single tap 4 = (dy 1, dx 1): lane group halves select the PART (a1*, b1*) or the TENSOR / weight set (a2, b2)
        const int aS1 = 4 * HQ_TAPU + (2 * kh + ks) * 64 + 32 * it + c;
        const int aS2 = 4 * HQ_TAPU + (2 * kh) * 64 + (ks ? 0 : HQ_WA) + 32 * it + c;
        const int bS1 = xb + (2 * kh + 1 - ks) * HQ_PP + (2 * jq) * HP_RS + c + SH4;
        const int bS2 = xb + (2 * kh) * HQ_PP + (ks ? HQ_XT : 0) + (2 * jq) * HP_RS + c + SH4;

        // ======== first stage: taps (0,1) (2,3) [4] from weight buffer A
        LA(dwh, HQ_WA + aP);
        pair(1, g, gn, xo, nb, px, 0, HQ_WA, xb + bP1,
             [&] { LA(dwh, 2 * HQ_TAPU + HQ_WA + aP); LB(xl, xb + 2 + bP32 + HQ_PP); },
             [&] { LB(xh, xb + 2 + bP32); });
        pair(0, g, gn, xo, nb, px, 2 * HQ_TAPU, HQ_WA, xb + 2 + bP32,
             [&] { LA(a1d, aS1 + HQ_WA); LB(b1x, bS1); },
             [&] { LA(a1w, aS1); });
        LA(a2, aS2);
        NBE_SB; MM8(dc, a1d, b1x, 0, 0, g, gn, xo, nb, px); NBE_SB;          // dwh.xl + dwl.xh
        LB(b2, bS2);
        {
            const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            a0[0] = ks ? a2[0] : zero;                                   // [0 | wh]  (VALU, one product ahead of its MFMAs)
            a0[1] = ks ? a2[1] : zero;
        }
        NBE_SB; MM8(yc, a1w, b1x, 0, 0, g, gn, xo, nb, px); NBE_SB;          // wh.xl + wl.xh
        LB(b1d, bS1 + HQ_XT);
        NBE_SB; MM8(ym, a0, b1x, 0, 0, g, gn, xo, nb, px, true); NBE_SB;     // wh.xh
        LB(xl, xb + SH5 + bP32 + HQ_PP);                                 // x of taps (5,6): this group's patch
        NBE_SB; MM8(dm, a2, b2, 0, 0, g, gn, xo, nb, px); NBE_SB;            // dwh.xh + wh.dxh
        LB(xh, xb + SH5 + bP32);
        NBE_SB; MM8(dc, a1w, b1d, 0, 0, g, gn, xo, nb, px); NBE_SB;          // wh.dxl + wl.dxh
        NBE_STAGE_END(1, 0)                                      // buffer B and the x patch of g+1 have landed

        // ======== second stage: taps (5,6) (7,8) from weight buffer B
        LA(dwh, HQ_OFF_B + HQ_WB + aP);
        pair(2, g, gn, xo, nb, px, HQ_OFF_B, HQ_WB, xb + SH5 + bP32,
             [&] { LA(dwh, HQ_OFF_B + 2 * HQ_TAPU + HQ_WB + aP); LB(xl, xb + SH7 + bP1 + HQ_PP); },
             [&] { LB(xh, xb + SH7 + bP1); });
        pair(0, g, gn, xo, nb, px, HQ_OFF_B + 2 * HQ_TAPU, HQ_WB, xb + SH7 + bP1,
             [&] { if (px) LB(xl, xbn + bP1 + HQ_PP); },         // x of taps (0,1) of the next group: landed above
             [&] { if (px) LB(xh, xbn + bP1); });
        NBE_STAGE_END(4, 0)                                      // buffer A and both patches of g+1 have landed
    }
#undef NBE_SB
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // MFMA results -> VALU reads of the epilogue

    // ---- epilogue: tile t = 4*mt + nt covers couts 32*it + 16*mt + 4*q .. +3 of position (2*jq + jt, 16*nh + c).
    // All global loads (bias, residuals of the 8 tiles) are issued before the first use: done tile by tile every
    // store would be waited for by the next tile's loads (vmcnt counts both) -- 12 % of the kernel's wave cycles
    // when it was written that way.  Out-of-range lanes load from a valid dummy address and skip the store.
    {
        const bool act = a.flags & F_ACT, res = a.flags & F_RES;
        int unit[2];
        bool uok[2];
        f32x4 bv[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            unit[mt] = ct * 8 + 4 * it + 2 * mt + ks;
            uok[mt] = unit[mt] < a.cout_groups;
            if (!uok[mt]) unit[mt] = a.cout_groups - 1;
            bv[mt] = *(const f32x4*)(a.bias + unit[mt] * 8 + 4 * kh);
        }
        long o[4];
        bool ook[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int yy = y0 + 2 * jq + (nt >> 1), xx = x0 + 16 * (nt & 1) + c;
            ook[nt] = yy < a.Hv && xx < a.Wv;
            o[nt] = ook[nt] ? ((long)z * a.Ho + yy) * a.Wo + xx : (long)z * a.Ho * a.Wo;
        }
        half4 rh[8], rl[8], dh[8], dl[8];
        if (res) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const long rb = ((long)(2 * unit[t >> 2]) * a.res_pstride + o[t & 3]) * 16 + 8 * kh;
                const long rl_ = rb + a.res_pstride * 16;
                rh[t] = *(const half4*)((const char*)a.r + rb);
                rl[t] = *(const half4*)((const char*)a.r + rl_);
                dh[t] = *(const half4*)((const char*)a.dr + rb);
                dl[t] = *(const half4*)((const char*)a.dr + rl_);
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int mt = t >> 2, nt = t & 3;
            f32x4 v, dv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = ym[t][e] + yc[t][e] * H3_INV + bv[mt][e];
                dv[e] = dm[t][e] + dc[t][e] * H3_INV;
            }
            if (res) { v += join4(rh[t], rl[t]); dv += join4(dh[t], dl[t]); }
            if (act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dv[e] = v[e] > 0.f ? dv[e] : 0.01f * dv[e];
                    v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
                }
            }
            if (uok[mt] && ook[nt]) {
                const long ob = ((long)(a.out_g0 + 2 * unit[mt]) * a.out_pstride + o[nt]) * 16 + 8 * kh;
                const long ol = ob + a.out_pstride * 16;
                half4 hi, lo;
                split4(v, hi, lo);
                *(half4*)((char*)a.y + ob) = hi;
                *(half4*)((char*)a.y + ol) = lo;
                split4(dv, hi, lo);
                *(half4*)((char*)a.dy + ob) = hi;
                *(half4*)((char*)a.dy + ol) = lo;
            }
        }
    }
#if NBE_DBG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NBE_STAMP(7)
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&h3q_stamps[i], tk[i]);
        atomicAdd(&h3q_stamps[8], 1ull);
    }
#endif
#undef NBE_STAMP
#undef NBE_STAGE_END
}

static int launch_h3q(ConvKArgs ka, int ctiles, hipStream_t s) {
    constexpr size_t smem = (size_t)HQ_LDS_UNITS * 16;
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_h3q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    ka.tny = (ka.Hv + HP_ROWS - 1) / HP_ROWS;
    ka.tnx = (ka.Wv + HP_COLS - 1) / HP_COLS;
    ka.ntiles = ka.Dv * ka.tny * ka.tnx;
    if (ctiles != (ka.cout_groups + 7) / 8) return 1;            // cout tiling mismatch: reported by the caller
    dim3 grid(ka.ntiles * ctiles, 1, 1), block(512, 1, 1);
    hipLaunchKernelGGL(conv_h3q_kernel, grid, block, smem, s, ka);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The gauged-tangent kernel: two float32 products per tap instead of three (f16x3, velocity, style modulation)
// ------------------------------------------------------------------------------------------------
// A style-modulated weight is w_n[o,i,k] = w[o,i,k] s[i] / norm[o] (style_layers_vel.py:62-105), so its derivative
// along Dz factorises: dw[o,i,k] = w_n[o,i,k] * (alpha[i] + beta[o]), alpha = s'/s, beta = norm' ... (modulate_kernel).
// Hence
//     dy = W.dx + dW.x = W.(dx + alpha (.) x) + beta (.) (W.x):
// if the PRODUCER of x stores the tangent as dx~ = dx + alpha (.) x (its epilogue knows alpha as `gout`, one FMA per
// element), this layer needs W.x and W.dx~ only -- six f16 MFMAs per tile and tap pair where conv_h3q_kernel issues
// nine, no dW operand at all, and beta[o] * (W.x) is one FMA per output in the epilogue.  The engine assigns every
// tensor the gauge of its one 3x3x3 consumer; its other consumers (1x1x1 skips, down-sampling) keep the general
// kernels with dW - W (.) alpha as their tangent weight (launch_modulate a_in), which is the same identity.
//
// Without dW a whole (chunk, dz) group of weights is 36 KB: weights AND patches are double-buffered by group
// (2 x 36 KB + 2 x 44 KB = the same 160 KB), every DMA is issued a full group (~2 us) ahead of its use, and a group
// ends with ONE barrier.  Operand layout, tap pairing, MFMA shape and the wave tile are those of conv_h3q_kernel.
//
// NARROW: the same kernel for layers with at most 16 output channels (conv_r01/conv_1, 64 -> 3: a full-resolution
// layer that the 64-cout tile computes at 64 / 3 times its cost -- 9 % of the wide kernel's time per box).  One
// 16-cout MFMA row tile; wave w owns row w of the 8 x 32 patch (two column tiles); weight rows are 16 couts wide
// ([group][tap][unit][16 couts][8 ch], packed with cout_t = 16), a group of weights is 9 KB.  Each activation operand
// then feeds one MFMA instead of two and the patch DMA is spread over a quarter of the MFMAs: the variant is bound by
// the L2 -> LDS stream (about 2650 cycles per group against 1800 of MFMA), i.e. ~3.8 x faster than the wide tile.
// TALL (wide tile only): the wave tile is 64 couts x one patch row (4 x 2 MFMA tiles) instead of 32 couts x two rows
// (2 x 4).  A product has two weight operands (hi, lo) but four activation operands (x, dx~, hi and lo each), so per tap
// pair a wave reads 2 * MT + 4 * NT 16-byte operand sets from LDS: 20 for 2 x 4, 16 for 4 x 2 -- and on this chip, which
// runs the kernel against its power limit, an LDS operand read costs 0.58 of an MFMA's energy
// (tools/micro/mfma_power.hip, profiles/r02_mfma_power.txt), so bytes per MFMA are time.
// BIG (wide tile only): 256 threads, ONE wave per SIMD with all 512 registers: wave tile 64 couts x two patch rows (4 x 4 MFMA
// tiles, 256 accumulator registers), 24 operand reads per 96 MFMAs (0.25 per MFMA against 0.33 for 4 x 2).  Same LDS image,
// same DMA -- issued by four waves instead of eight, four slots per product.
template <bool NARROW, bool TALL = false, bool BIG = false>
struct HGGeom {
    static_assert(!(NARROW && TALL) && !(NARROW && BIG) && !(TALL && BIG), "one geometry at a time");
    static constexpr int CT = NARROW ? 16 : 64;             // couts per tile = rows of one weight unit in LDS
    static constexpr int TAPU = 4 * CT;                     // 16-byte units per tap
    static constexpr int WG = 9 * TAPU;                     // units of one group's weights (set w)
    static constexpr int XBASE = 2 * WG;                    // the two patch buffers follow the two weight buffers
    static constexpr int LDS_UNITS = XBASE + 2 * HQ_XB;
    static constexpr int MT = NARROW ? 1 : ((TALL || BIG) ? 4 : 2), NT = (NARROW || TALL) ? 2 : 4, NTILE = MT * NT;   // MFMA tiles of a wave
    static constexpr int NW = BIG ? 4 : 8;                  // waves per workgroup
    static constexpr int XS = 24 / NW;                      // patch DMA slots per wave and tensor (24 wave-instructions per tensor)
    static constexpr int SPP = NTILE == 16 ? 4 : 2;         // DMA slots a product can carry (one after every fourth MFMA)
    static constexpr bool ROWW = NARROW || TALL;            // a wave owns ONE row of the 8 x 32 patch (else two)
    static constexpr int NWI = WG / 64;                     // weight DMA wave-instructions per group: 36 / 9
    static constexpr int NWS = (NWI + NW - 1) / NW;         // ... slots per wave: 5 / 2 / 9
    static_assert(NWS + 2 * XS <= 6 * SPP, "the first tap pair carries the whole DMA of the next group");
};
static_assert(HGGeom<false>::XBASE == HQ_XBASE, "the two weight buffers fill exactly what conv_h3q_kernel uses for four");

template <bool NARROW, bool TALL, bool BIG>
__global__ __launch_bounds__(BIG ? 256 : 512, BIG ? 1 : 2) void conv_h3g_kernel(ConvKArgs a) {
    typedef HGGeom<NARROW, TALL, BIG> G;
    constexpr bool ROWW = G::ROWW;
    constexpr int NW = G::NW, XS = G::XS, SPP = G::SPP;
    constexpr int CT = G::CT, TAPU = G::TAPU, WGU = G::WG, XBASE = G::XBASE, MT = G::MT, NT = G::NT, NTILE = G::NTILE;
    constexpr int NWS = G::NWS;
#if NBE_DBG   // phases as h3q_stamps: 0 prologue, 1 group up to its barrier, 2 own DMA, 3 barrier, 4 the three products after it, 7 epilogue
    unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define NBE_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tk[i] += t_ - tk0; tk0 = t_; }
#else
#define NBE_STAMP(i)
#endif
    f32x4* lds = lds_h3;
    const half8* L8 = (const half8*)lds_h3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4, kh = q & 1, ks = q >> 1;
    const int it = wave & 1, jq = wave >> 1;

    // cout tile fastest, then z (as in conv_h3q_kernel): the workgroups of one patch that differ only in their 64 couts
    // run side by side on one XCD and share the patch through its L2 instead of fetching it from HBM once per cout tile
    const int nct = (a.cout_groups + CT / 8 - 1) / (CT / 8);
    const int vt = xcd_tile(blockIdx.x, a.ntiles * nct);
    const int tile = vt / nct, ct = vt - tile * nct;
    const int z = tile % a.Dv, tyx = tile / a.Dv;
    const int ty = tyx / a.tnx, tx = tyx - ty * a.tnx;
    const int y0 = ty * HP_ROWS, x0 = tx * HP_COLS;
    const int ngroups = 3 * a.nchunk;

    const unsigned lane16 = (unsigned)lane * 16u;
    // ---- DMA of group g into buffers g & 1: 36 wave-instructions of weights (5 slots per wave), 24 + 24 of the
    // x and dx~ patches (3 + 3 slots per wave).
    // Groups [0, ngroups) are the (chunk, dz) groups of the 3x3x3 layer; chunks >= csplit come from a second tensor of
    // the same geometry (the decoder's concat([skip, up]) without a materialised concat tensor, core :168-169).
    // Groups [ngroups, ngroups + nskip) are the block's 1x1x1 skip (style_blocks_vel.py:108-123) fused into this,
    // the block's last, convolution: one 16-channel chunk of the BLOCK INPUT each (its patch placed so that the centre
    // tap is the skip's voxel), weights [W_s | dW_s~] of that chunk -- see the skip body below.
    const int nskip = a.nskip;
    // Per group the host has prepared {x, dx, w, plane stride} (ConvKArgs::gs, in the kernel-argument segment): all that
    // depends on the group but not on the tile.  The tile adds its patch origin and its cout tile.  (Computing the
    // sources here from the dozen pointers and strides they derive from kept ~80 more SGPRs alive through the loop.)
    // (the skip's input has the row and plane pitch of the layer's input -- the engine allocates the block's hidden
    // tensor with the pitch of the block input -- so one patch origin and one set of per-lane offsets serve both)
    const long to = (((long)z * a.H + y0) * a.W + x0) * 16;
    const long wcm = (long)ct * ngroups * WGU * 16, wcs = (long)ct * nskip * TAPU * 16;   // a skip chunk: one "tap" of W_s (and of dW_s~)
    struct Nxt { const char *x, *dx, *w0; long psb; bool sk; } nx;   // sources of the group being fetched
    auto set_next = [&](int g) {
        const ConvGroupSrc e = a.gs[g];
        const bool sk = g >= ngroups;
        nx.x = e.x + to; nx.dx = e.dx + to; nx.psb = e.psb; nx.w0 = e.w + (sk ? wcs : wcm); nx.sk = sk;
    };
    auto dma_w = [&](int buf, int t) {
        const int n = wave + NW * t;
        constexpr int PER = TAPU / 64;                           // wave-instructions per skip weight set: 4 / 1
        if (!nx.sk) { if (n < G::NWI) dma16s(nx.w0 + (long)n * 1024, lane16, lds + buf * WGU + n * 64); }
        else if (n < 2 * PER) dma16s(nx.w0 + (n < PER ? 0 : a.dws_delta) + (long)(n % PER) * 1024, lane16, lds + buf * WGU + n * 64);
    };
    unsigned xoff[XS];                                           // per-lane byte offset inside a patch plane
    bool xval[XS];
#pragma unroll
    for (int t = 0; t < XS; ++t) {
        const int k = (wave + NW * t) % 6;
        const int u = k * 64 + lane;
        xval[t] = u < HP_PL;
        const int uu = xval[t] ? u : HP_PL - 1;
        const int row = uu / HP_RS, col = uu - row * HP_RS;
        xoff[t] = (unsigned)(row * a.W + col) * 16u;
    }
    auto dma_x = [&](int tensor, int t, int buf) {
        const int n = wave + NW * t, pl = n / 6, k = n - 6 * pl;
        if (xval[t])
            dma16s((tensor ? nx.dx : nx.x) + (long)pl * nx.psb, xoff[t],
                   lds + XBASE + buf * HQ_XB + tensor * HQ_XT + pl * HQ_PP + k * 64);
    };
    auto dma_slot = [&](int k, int buf) {                        // slot k of the NWS + 2 XS (11 / 8 / 21) of the group in `nx`
        if (k < NWS) dma_w(buf, k);
        else if (k < NWS + XS) dma_x(0, k - NWS, buf);
        else if (k < NWS + 2 * XS) dma_x(1, k - NWS - XS, buf);
    };

    f32x4 ym[NTILE], yc[NTILE], dm[NTILE], dc[NTILE];            // AGPRs, updated in place (see conv_h3q_kernel)
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ym[t][e] = 0.f; yc[t][e] = 0.f; dm[t][e] = 0.f; dc[t][e] = 0.f; }
    auto mm = [&](f32x4& acc, const half8& A, const half8& B) {
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
    };
    // the same behind a VALU select: the compiler does not see an MFMA inside an asm statement and leaves out the two wait
    // states the ISA wants between a VALU write of a VGPR and an MFMA that reads it (tools/check_mfma_hazards.py)
    auto mmz = [&](f32x4& acc, const half8& A, const half8& B) {
        asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
    };

    const int rowc = (ROWW || BIG) ? 0 : 32 * it;                // first cout row of this wave inside the tile
    const int rowp = ROWW ? wave : (BIG ? 2 * wave : 2 * jq);    // first patch row of this wave
    const int aP = (ks * 4 + 2 * kh) * CT + rowc + c;
    const int bB = (2 * kh) * HQ_PP + rowp * HP_RS + c;
    const int bP1 = bB + ks, bP32 = bB + 32 * ks;
    auto LA = [&](half8 (&r)[MT], int idx) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) r[mt] = L8[idx + 16 * mt];
    };
    auto LB = [&](half8 (&r)[NT], int idx) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) r[nt] = L8[idx + (ROWW ? 0 : (nt >> 1)) * HP_RS + 16 * (ROWW ? nt : (nt & 1))];
    };
    // one product on the wave tile: NTILE MFMAs; slot >= 0: DMA slots `slot`, `slot + 1` of group gn, one after each
    // MFMA row (wide) or both after the product (narrow)
    auto MM8 = [&](f32x4 (&acc)[NTILE], const half8 (&A)[MT], const half8 (&B)[NT], int slot, int nb, bool px,
                   bool zsel = false /* A comes out of a VALU select */) {
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            if (zsel && (t % NT) == 0) mmz(acc[t], A[t / NT], B[t % NT]); else
            mm(acc[t], A[t / NT], B[t % NT]);
            if (slot >= 0 && (NTILE >= 8 ? (t % 4) == 3 : t == NTILE - 1)) {
                if (px) {
                    if (NTILE >= 8) dma_slot(slot + t / 4, nb);
                    else { dma_slot(slot, nb); dma_slot(slot + 1, nb); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#define NBE_SB __builtin_amdgcn_sched_barrier(0)
    half8 wh[MT], wl[MT], xh[NT], xl[NT], dxh[NT], dxl[NT];
    // A tap pair: six products.  On entry wh, xl and xh of the pair are loaded (or in flight); preXl / preW / preXh
    // request those of whatever follows as soon as the registers are free.  Dependent MFMAs are >= 8 MFMAs apart.
    // (mid: after the third product every LDS read of the pair has been issued -- the group's barrier goes there)
    auto pair = [&](int slot0, int nb, bool px, int wa, int xp, auto&& preXl, auto&& preW, auto&& preXh, auto&& mid) {
        LA(wl, wa + CT + aP); LB(dxh, xp + HQ_XT);
        NBE_SB; MM8(yc, wh, xl, slot0, nb, px); NBE_SB;
        LB(dxl, xp + HQ_XT + HQ_PP);
        NBE_SB; MM8(ym, wh, xh, slot0 < 0 ? -1 : slot0 + SPP, nb, px); NBE_SB;
        MM8(dm, wh, dxh, slot0 < 0 ? -1 : slot0 + 2 * SPP, nb, px); NBE_SB;
        mid();
        preXl();
        NBE_SB; MM8(dc, wh, dxl, slot0 < 0 ? -1 : slot0 + 3 * SPP, nb, px); NBE_SB;
        preW();
        NBE_SB; MM8(yc, wl, xh, slot0 < 0 ? -1 : slot0 + 4 * SPP, nb, px); NBE_SB;
        preXh();
        NBE_SB; MM8(dc, wl, dxh, slot0 < 0 ? -1 : slot0 + 5 * SPP, nb, px); NBE_SB;
    };

    // ---- prologue: group 0
    {
        set_next(0);
#pragma unroll
        for (int k = 0; k < NWS + 2 * XS; ++k) dma_slot(k, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        NBE_STAMP(0)
        LA(wh, aP); LB(xl, XBASE + bP1 + HQ_PP); LB(xh, XBASE + bP1);
    }

    constexpr int SH4 = HP_RS + 1, SH5 = HP_RS + 2, SH7 = 2 * HP_RS + 1;   // tap shifts: 3*dy + dx -> dy*34 + dx
    for (int g = 0; g < ngroups; ++g) {
        const bool px = g + 1 < ngroups + nskip;
        if (px) set_next(g + 1);
        const int nb = (g + 1) & 1;
        const int wb = (g & 1) * WGU, xb = XBASE + (g & 1) * HQ_XB;
        const int wbn = WGU - wb, xbn = XBASE + nb * HQ_XB;
        half8 a1w[MT], a0[MT], b1x[NT], b1d[NT];
        // single tap 4 = (dy 1, dx 1): the lane-group halves select the PART: [wh|wl].[xl|xh] and [0|wh].[xl|xh]
        const int aS1 = wb + 4 * TAPU + (2 * kh + ks) * CT + rowc + c;
        const int aS0 = wb + 4 * TAPU + (2 * kh) * CT + rowc + c;
        const int bS1 = xb + (2 * kh + 1 - ks) * HQ_PP + rowp * HP_RS + c + SH4;

        pair(0, nb, px, wb, xb + bP1,                                                  // taps (0,1) + the DMA of group g+1
             [&] { LB(xl, xb + 2 + bP32 + HQ_PP); }, [&] { LA(wh, wb + 2 * TAPU + aP); }, [&] { LB(xh, xb + 2 + bP32); },
             [&] {});
        pair(-1, nb, px, wb + 2 * TAPU, xb + 2 + bP32,                                 // taps (2,3)
             [&] { LB(b1x, bS1); }, [&] { LA(a1w, aS1); LA(a0, aS0); }, [&] { LB(b1d, bS1 + HQ_XT); }, [&] {});
        {
            const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a0[mt] = ks ? a0[mt] : zero;       // [0 | wh]
        }
        NBE_SB; MM8(yc, a1w, b1x, -1, 0, false); NBE_SB;                     // wh.xl + wl.xh
        LB(xl, xb + SH5 + bP32 + HQ_PP);
        NBE_SB; MM8(ym, a0, b1x, -1, 0, false, true); NBE_SB;                // wh.xh
        LB(xh, xb + SH5 + bP32); LA(wh, wb + 5 * TAPU + aP);
        NBE_SB; MM8(dc, a1w, b1d, -1, 0, false); NBE_SB;                     // wh.dxl + wl.dxh
        MM8(dm, a0, b1d, -1, 0, false, true); NBE_SB;                        // wh.dxh
        pair(-1, nb, px, wb + 5 * TAPU, xb + SH5 + bP32,                               // taps (5,6)
             [&] { LB(xl, xb + SH7 + bP1 + HQ_PP); }, [&] { LA(wh, wb + 7 * TAPU + aP); }, [&] { LB(xh, xb + SH7 + bP1); },
             [&] {});
        // taps (7,8).  The group's one barrier sits after the third product: by then this wave has read everything it
        // needs from the buffers of group g, and all of group g+1 has landed once every wave has waited for its own
        // DMA -- the first operands of group g+1 are requested under the last three products.
        const bool pm = g + 1 < ngroups;                             // the next group is a 3x3x3 group (not a skip group)
        pair(-1, nb, px, wb + 7 * TAPU, xb + SH7 + bP1,
             [&] { if (pm) LB(xl, xbn + bP1 + HQ_PP); }, [&] { if (pm) LA(wh, wbn + aP); }, [&] { if (pm) LB(xh, xbn + bP1); },
             [&] { NBE_STAMP(1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); NBE_STAMP(2)
                   asm volatile("s_barrier" ::: "memory"); NBE_STAMP(3) });
        NBE_STAMP(4)
    }
    // ---- the block's 1x1x1 skip, one 16-channel chunk of the block input per group.  With the block input x stored
    // in the gauge a of its 3x3x3 reader (dx~ = dx + a x) the skip's tangent W_s.dx + dW_s.x is
    //     W_s.dx~ + (dW_s - W_s (.) a) . x,
    // and this kernel's epilogue adds beta[o] * (everything accumulated in y), so the weight the host packs as dW_s~ is
    //     dW_s - W_s (.) a[i] - beta[o] W_s
    // (launch_modulate a_in / b_sub): y += W_s.x,  dy += W_s.dx~ + dW_s~.x -- six products on the centre tap, parts
    // paired in K as for tap 4 above: [wh|wl].[xl|xh] -> correction, [0|wh].[xl|xh] -> main.
    {
        for (int sc = 0; sc < nskip; ++sc) {
            const int g = ngroups + sc;
            const bool px = sc + 1 < nskip;
            if (px) set_next(g + 1);
            const int nb = (g + 1) & 1;
            const int wb = (g & 1) * WGU, xb = XBASE + (g & 1) * HQ_XB;
            half8 a1w[MT], a0[MT], a1d[MT], a0d[MT], b1x[NT], b1d[NT];
            const int aS1 = wb + (2 * kh + ks) * CT + rowc + c, aS0 = wb + (2 * kh) * CT + rowc + c;
            const int bS1 = xb + (2 * kh + 1 - ks) * HQ_PP + rowp * HP_RS + c + SH4;
            const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            LA(a1w, aS1); LB(b1x, bS1); LA(a0, aS0); LB(b1d, bS1 + HQ_XT);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a0[mt] = ks ? a0[mt] : zero;
            NBE_SB; MM8(yc, a1w, b1x, 0, nb, px); NBE_SB;                    // W_s.x, correction terms
            LA(a1d, aS1 + 4 * CT);                                           // (the dW_s~ operands follow as registers free up)
            NBE_SB; MM8(ym, a0, b1x, SPP, nb, px, true); NBE_SB;             //        main term
            if (!(a.flags & F_SKIP_NODX)) {                                  // (conv_l00: the input field has no tangent)
                MM8(dc, a1w, b1d, 2 * SPP, nb, px); NBE_SB;                  // W_s.dx~
                LA(a0d, aS0 + 4 * CT);
                NBE_SB; MM8(dm, a0, b1d, 3 * SPP, nb, px, true); NBE_SB;
            } else {
                LA(a0d, aS0 + 4 * CT);
                if (px) {
#pragma unroll
                    for (int k = 2 * SPP; k < 4 * SPP; ++k) dma_slot(k, nb);
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a0d[mt] = ks ? a0d[mt] : zero;
            NBE_SB; MM8(dc, a1d, b1x, 4 * SPP, nb, px); NBE_SB;              // dW_s~.x
            MM8(dm, a0d, b1x, 5 * SPP, nb, px, true); NBE_SB;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
#undef NBE_SB
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // MFMA results -> VALU reads of the epilogue

    // ---- epilogue (layout and load-first order of conv_h3q_kernel): y = W.x + b, dy = W.dx~ + beta * (W.x)
    {
        const bool act = a.flags & F_ACT, res = a.flags & F_RES, gauge = a.gout != nullptr;
        int o[NT];                                               // voxel index in the output planes (< 2^31: tiles are <= 608^3)
        bool ook[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int yy = y0 + rowp + (ROWW ? 0 : (nt >> 1)), xx = x0 + 16 * (ROWW ? nt : (nt & 1)) + c;
            ook[nt] = yy < a.Hv && xx < a.Wv;
            o[nt] = ook[nt] ? (z * a.Ho + yy) * a.Wo + xx : z * a.Ho * a.Wo;
        }
        // one row of MFMA tiles (16 couts = 2 units) at a time: its per-channel vectors, the residuals of its NT tiles
        // (loads first), then the tiles -- all MT rows at once do not fit the registers of the 4 x 2 wave tile
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int unit = ct * (CT / 8) + ((ROWW || BIG) ? 0 : 4 * it) + 2 * mt + ks;
            const bool uok = unit < a.cout_groups;
            if (!uok) unit = a.cout_groups - 1;
            const f32x4 bv = *(const f32x4*)(a.bias + unit * 8 + 4 * kh);
            const f32x4 be = *(const f32x4*)(a.beta + unit * 8 + 4 * kh);
            f32x4 gv = {0.f, 0.f, 0.f, 0.f};
            if (gauge) gv = *(const f32x4*)(a.gout + unit * 8 + 4 * kh);
            half4 rh[NT], rl[NT], dh[NT], dl[NT];
            if (res) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const long rb = ((long)(2 * unit) * a.res_pstride + (long)o[nt]) * 16 + 8 * kh;
                    const long rl_ = rb + a.res_pstride * 16;
                    rh[nt] = *(const half4*)((const char*)a.r + rb);
                    rl[nt] = *(const half4*)((const char*)a.r + rl_);
                    dh[nt] = *(const half4*)((const char*)a.dr + rb);
                    dl[nt] = *(const half4*)((const char*)a.dr + rl_);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int t = mt * NT + nt;
                f32x4 v, dv;
                // the accumulators leave their AGPRs tile by tile, here: copied out wholesale at the top of the epilogue
                // (what the compiler does by itself) they do not fit beside the residuals and spill, and a scratch
                // reload among the stores waits for every store before it
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float yp = acc_read(ym[t][e]) + acc_read(yc[t][e]) * H3_INV;
                    v[e] = yp + bv[e];
                    dv[e] = acc_read(dm[t][e]) + acc_read(dc[t][e]) * H3_INV + be[e] * yp;
                }
                if (res) { v += join4(rh[nt], rl[nt]); dv += join4(dh[nt], dl[nt]); }
                if (act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dv[e] = v[e] > 0.f ? dv[e] : 0.01f * dv[e];
                        v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
                    }
                }
                if (gauge) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[e] += gv[e] * v[e];
                }
                if (uok && ook[nt]) {
                    const long ob = ((long)(a.out_g0 + 2 * unit) * a.out_pstride + (long)o[nt]) * 16 + 8 * kh;
                    const long ol = ob + a.out_pstride * 16;
                    half4 hi, lo;
                    split4(v, hi, lo);
                    *(half4*)((char*)a.y + ob) = hi;
                    *(half4*)((char*)a.y + ol) = lo;
                    split4(dv, hi, lo);
                    *(half4*)((char*)a.dy + ob) = hi;
                    *(half4*)((char*)a.dy + ol) = lo;
                }
            }
        }
    }
#if NBE_DBG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NBE_STAMP(7)
    if (lane == 0 && (blockIdx.x & 63) == 0) {                   // a sample: the atomics of every wave would dominate the run
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&h3q_stamps[i], tk[i]);
        atomicAdd(&h3q_stamps[8], 1ull);
    }
#endif
#undef NBE_STAMP
}

template <bool NARROW, bool TALL, bool BIG = false>
static int launch_h3g(ConvKArgs ka, int ctiles, hipStream_t s) {
    typedef HGGeom<NARROW, TALL, BIG> G;
    constexpr size_t smem = (size_t)G::LDS_UNITS * 16;
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_h3g_kernel<NARROW, TALL, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    ka.tny = (ka.Hv + HP_ROWS - 1) / HP_ROWS;
    ka.tnx = (ka.Wv + HP_COLS - 1) / HP_COLS;
    ka.ntiles = ka.Dv * ka.tny * ka.tnx;
    if (ctiles != (ka.cout_groups + G::CT / 8 - 1) / (G::CT / 8)) return 1;
    const int ngroups = 3 * ka.nchunk, nskip = ka.nskip;
    if (ngroups + nskip > NBE_MAX_GROUPS) return 1;              // the engine does not wire such a network for this kernel
    for (int g = 0; g < ngroups; ++g) {
        const int chunk = g / 3, dz = g - chunk * 3;
        const bool second = chunk >= ka.csplit;
        const long ps = second ? ka.in2_pstride : ka.in_pstride;
        const long off = ((long)(second ? chunk - ka.csplit : chunk) * 4 * ps + (long)dz * ka.H * ka.W) * 16;
        ka.gs[g] = {(const char*)(second ? ka.x2 : ka.x) + off, (const char*)(second ? ka.dx2 : ka.dx) + off,
                    (const char*)ka.w + (long)g * G::WG * 16, ps * 16};
    }
    for (int sc = 0; sc < nskip; ++sc) {
        const bool second = sc >= ka.s_csplit;
        const long ps = second ? ka.s2_pstride : ka.s_pstride;
        const long off = (long)(second ? sc - ka.s_csplit : sc) * 4 * ps * 16;
        ka.gs[ngroups + sc] = {(const char*)(second ? ka.xs2 : ka.xs) + off, (const char*)(second ? ka.dxs2 : ka.dxs) + off,
                               (const char*)ka.ws + (long)sc * G::TAPU * 16, ps * 16};
    }
    ka.dws_delta = nskip ? (const char*)ka.dws - (const char*)ka.ws : 0;
    dim3 grid(ka.ntiles * ctiles, 1, 1), block(G::NW * 64, 1, 1);
    hipLaunchKernelGGL((conv_h3g_kernel<NARROW, TALL, BIG>), grid, block, smem, s, ka);
    return 0;
}

#include "nbe_kernels_wino.h"
#include "nbe_kernels_head.h"

// ------------------------------------------------------------------------------------------------
// The two-accumulator variants on the 16x16x32 shape: f16x3 without velocity and plain f16 with velocity
// ------------------------------------------------------------------------------------------------
// Both have one product for the first accumulator set and two for the second:
//   SPLIT  (f16x3, displacement only): a0 = w hi, a1 = w lo, b0 = x hi, b1 = x lo:  ym += a0.b0, yc += a0.b1 + a1.b0
//   !SPLIT (f16, velocity):            a0 = w,    a1 = dw,   b0 = x,    b1 = dx:    ym += a0.b0, dm += a0.b1 + a1.b0
// so one kernel serves both; what differs is where a1 and b1 live (the lo part / the tangent tensor) and the epilogue.
// With half the accumulators of conv_h3q_kernel the workgroup tile doubles to 64 couts x 512 positions (16 rows x 32
// columns of one plane): per MFMA the same bytes of weights and activations as there, instead of 1.5 x as many in the
// 32x32x16 patch kernel, which left these two modes bound by the L2 -> LDS stream (0.35 of their peaks).
// K = 32 is two taps x 16 channels as in conv_h3q_kernel; the single tap of a group pairs a0/a1 with b1/b0:
// acc1 += [a0|a1].[b1|b0] and acc0 += [0|a0].[b1|b0].  Wave tile 32 couts x 128 positions (4 rows) = 2 x 8 MFMA tiles,
// tile t = 8*mt + nt, nt = 2*row + column half; the rolling operand pipeline works on half tiles (2 rows).
// LDS: weight buffer A (taps 0-4) / B (taps 5-8) with rows [tap][r][64 couts], r = 2*h + part (SPLIT) or 2*set + h;
// patch planes p = 2*h + part (SPLIT) or 2*tensor + h, 18 x 34 units each.
constexpr int H2_ROWS = 16;
constexpr int H2_PL = (H2_ROWS + 2) * HP_RS;              // units per patch plane: 612
// LDS pitch of a patch plane, padded so that the two planes a ds_read_b128 mixes in one LDS cycle start 0 mod 16 units
// apart (no bank conflicts): planes 2*kh apart in the SPLIT variant (pitch multiple of 8), kh apart otherwise (of 16)
constexpr int H2_PP = 624;
constexpr int H2_XB = 4 * H2_PP;                          // one patch buffer: 4 planes
constexpr int H2_TAPU = 4 * 64;                           // units per tap (all rows)
constexpr int H2_WA = 5 * H2_TAPU, H2_WB = 4 * H2_TAPU;
constexpr int H2_OFF_B = H2_WA;
constexpr int H2_XBASE = H2_WA + H2_WB;
constexpr int H2_LDS_UNITS = H2_XBASE + 2 * H2_XB;        // 7200 units = 115,200 B
constexpr int H2_NPI = (H2_PL + 63) / 64;                 // DMA instructions per patch plane: 10 (the last one 36 lanes)

// G6 (!SPLIT only): the input tangent arrives in this layer's gauge (see conv_h3g_kernel): dm += a0.b1 only, no dw operand,
// dy = dm + beta * ym in the epilogue -- two products per tap instead of three.
template <bool SPLIT, bool G6 = false>
__global__ __launch_bounds__(512, 2) void conv_h2q_kernel(ConvKArgs a) {
    static_assert(!(SPLIT && G6), "the gauged form belongs to the velocity variant");
    f32x4* lds = lds_h3;
    const half8* L8 = (const half8*)lds_h3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4, kh = q & 1, ks = q >> 1;
    const int it = wave & 1, jq = wave >> 1;

    const int nct = (a.cout_groups + 7) / 8;                     // cout tile fastest, then z (see conv_h3g_kernel)
    const int vt = xcd_tile(blockIdx.x, a.ntiles * nct);
    const int tile = vt / nct, ct = vt - tile * nct;
    const int z = tile % a.Dv, tyx = tile / a.Dv;
    const int ty = tyx / a.tnx, tx = tyx - ty * a.tnx;
    const int y0 = ty * H2_ROWS, x0 = tx * HP_COLS;
    const int ngroups = 3 * a.nchunk;
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- DMA.  Weight rows: instruction n = wave + 8t of a stage covers (tap, r) = (n / 4, n % 4).
    auto dma_w = [&](int g, int second, int t) {
        const int n = wave + 8 * t;
        if (n >= (second ? 16 : 20)) return;
        const int tap = (second ? 5 : 0) + (n >> 2), r = n & 3;
        if (G6 && (r >> 1)) return;                              // no dw rows
        const char* src;
        if (SPLIT) {
            src = (const char*)a.w + (((long)ct * ngroups + g) * 9 * 4 + tap * 4 + r) * 64 * 16;
        } else {
            src = (const char*)((r >> 1) ? a.dw : a.w) + (((long)ct * ngroups + g) * 9 * 2 + tap * 2 + (r & 1)) * 64 * 16;
        }
        dma16s(src, lane16, lds + (second ? H2_OFF_B : 0) + n * 64);
    };
    // Patch: 10 instructions per plane, instruction n = wave + 8t (t < 5) -> plane n / 10, piece n % 10
    unsigned xoff[5];
    bool xval[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int k = (wave + 8 * t) % H2_NPI;
        const int u = k * 64 + lane;
        xval[t] = u < H2_PL;
        const int uu = xval[t] ? u : H2_PL - 1;
        const int row = uu / HP_RS, col = uu - row * HP_RS;
        xoff[t] = (unsigned)(row * a.W + col) * 16u;
    }
    auto patch_offset = [&](int g) -> long {                     // g = chunk*3 + dz
        const int chunk = g / 3, dz = g - chunk * 3;
        return ((long)chunk * (SPLIT ? 4 : 2) * a.in_pstride + ((long)(z + dz) * a.H + y0) * a.W + x0) * 16;
    };
    auto dma_x = [&](int t, long xo, int buf) {
        const int n = wave + 8 * t, pl = n / H2_NPI, k = n - H2_NPI * pl;
        if (xval[t]) {
            const char* base = SPLIT ? (const char*)a.x + (long)pl * a.in_pstride * 16
                                     : (const char*)((pl >> 1) ? a.dx : a.x) + (long)(pl & 1) * a.in_pstride * 16;
            dma16s(base + xo, xoff[t], lds + H2_XBASE + buf * H2_XB + pl * H2_PP + k * 64);
        }
    };

    f32x4 acc0[16], acc1[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc0[t][e] = 0.f; acc1[t][e] = 0.f; }
    auto mm = [&](f32x4& acc, const half8& A, const half8& B) {   // accumulators pinned in AGPRs, updated in place
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
    };
    auto mmz = [&](f32x4& acc, const half8& A, const half8& B) {  // behind a VALU select: see conv_h3q_kernel
        asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
    };

    // ---- operands
    constexpr int A1OFF = SPLIT ? 64 : 128;                      // a1 = a0 + one row (lo part) / two rows (dw set)
    constexpr int B1OFF = SPLIT ? H2_PP : 2 * H2_PP;             // b1 = b0 + one plane (lo part) / two planes (dx tensor)
    const int rowA = SPLIT ? 2 * kh : kh;
    const int aP = (ks * 4 + rowA) * 64 + 32 * it + c;           // pair: tap ks of the pair, row of a0
    const int bB = rowA * H2_PP + (4 * jq) * HP_RS + c;          // plane of b0 (same index rule as the weight row)
    const int bP1 = bB + ks, bP32 = bB + 32 * ks;
    auto LA = [&](half8 (&r)[2], int idx) {
        r[0] = L8[idx];
        r[1] = L8[idx + 16];
    };
    // B operands of one half tile (rows 2*hf, 2*hf + 1): 4 column tiles
    auto LB = [&](half8 (&r)[4], int idx, int hf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = L8[idx + (2 * hf + (j >> 1)) * HP_RS + 16 * (j & 1)];
    };
    // one product on one half tile: 8 MFMAs on tiles t = 8*mt + 4*hf + j; kind 1 / 2: DMA slots `slot`, `slot + 1`
    auto MM8 = [&](f32x4 (&acc)[16], const half8 (&A)[2], const half8 (&B)[4], int hf, int kind, int slot, int g,
                   int gn, long xo, int nb, bool px, bool zsel = false /* A comes out of a VALU select */) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (zsel && (i & 3) == 0) mmz(acc[8 * (i >> 2) + 4 * hf + (i & 3)], A[i >> 2], B[i & 3]); else
            mm(acc[8 * (i >> 2) + 4 * hf + (i & 3)], A[i >> 2], B[i & 3]);
            if (kind != 0 && (i & 3) == 3) {
                const int k = slot + (i >> 2);
                if (kind == 1) {                                 // first stage: weights of the second, patch pieces 0-2
                    if (k < 2) dma_w(g, 1, k);
                    else if (k < 5 && px) dma_x(k - 2, xo, nb);
                } else if (px) {                                 // second stage: weights of the next group, pieces 3-4
                    if (k < 3) dma_w(gn + 1, 0, k);
                    else if (k < 5) dma_x(k, xo, nb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#define NBE_SB __builtin_amdgcn_sched_barrier(0)
    half8 a0x[2], a0y[2], a1[2], b1[4], b0a[4], b0b[4];
    // A tap pair on both half tiles: six products; every operand is requested two products before its first use and
    // at most 64 operand registers are alive.  On entry a0 (A0), b1 and b0a of the first half are loaded or in
    // flight; pre5 / pre6 request those of whatever follows (a0 and b1 in pre5, b0a in pre6).
    auto pair = [&](half8 (&A0)[2], int kind, int g, int gn, long xo, int nb, bool px, int wa, int xp,
                    auto&& pre5, auto&& pre6) {
        if (!G6) LA(a1, wa + A1OFF + aP);
        NBE_SB; MM8(acc1, A0, b1, 0, kind, 0, g, gn, xo, nb, px); NBE_SB;      // a0.b1, first half
        LB(b1, xp + B1OFF, 1);
        NBE_SB; MM8(acc0, A0, b0a, 0, kind, 2, g, gn, xo, nb, px); NBE_SB;     // a0.b0
        LB(b0b, xp, 1);
        if (!G6) { NBE_SB; MM8(acc1, a1, b0a, 0, kind, 4, g, gn, xo, nb, px); NBE_SB; }    // a1.b0
        MM8(acc1, A0, b1, 1, G6 ? kind : 0, 4, g, gn, xo, nb, px); NBE_SB;     // second half (G6: the last DMA slots)
        pre5();
        NBE_SB; MM8(acc0, A0, b0b, 1, 0, 0, g, gn, xo, nb, px); NBE_SB;
        pre6();
        if (!G6) { NBE_SB; MM8(acc1, a1, b0b, 1, 0, 0, g, gn, xo, nb, px); NBE_SB; }
    };

    // ---- prologue: the patch of group 0 and the weights of its first stage
    {
        const long x0off = patch_offset(0);
#pragma unroll
        for (int t = 0; t < 5; ++t) dma_x(t, x0off, 0);
#pragma unroll
        for (int t = 0; t < 3; ++t) dma_w(0, 0, t);
        __syncthreads();
    }

    constexpr int SH4 = HP_RS + 1, SH5 = HP_RS + 2, SH7 = 2 * HP_RS + 1;
    for (int g = 0; g < ngroups; ++g) {
        const bool px = g + 1 < ngroups;
        const int gn = g;
        const long xo = px ? patch_offset(g + 1) : 0;
        const int nb = (g + 1) & 1;
        const int xb = H2_XBASE + (g & 1) * H2_XB;
        half8 as[2], am[2], bsa[4], bsb[4];
        const int aS = 4 * H2_TAPU + (SPLIT ? (2 * kh + ks) : (kh + 2 * ks)) * 64 + 32 * it + c;     // [a0 | a1]
        const int aM = 4 * H2_TAPU + rowA * 64 + 32 * it + c;                                       // a0 for both halves
        const int bS = xb + (rowA + (1 - ks) * (SPLIT ? 1 : 2)) * H2_PP + (4 * jq) * HP_RS + c + SH4;   // [b1 | b0]

        // ======== first stage: taps (0,1) (2,3) [4] from weight buffer A
        LA(a0x, aP); LB(b1, xb + bP1 + B1OFF, 0); LB(b0a, xb + bP1, 0);
        pair(a0x, 1, g, gn, xo, nb, px, 0, xb + bP1,
             [&] { LA(a0y, 2 * H2_TAPU + aP); LB(b1, xb + 2 + bP32 + B1OFF, 0); },
             [&] { LB(b0a, xb + 2 + bP32, 0); });
        pair(a0y, 0, g, gn, xo, nb, px, 2 * H2_TAPU, xb + 2 + bP32,
             [&] { LA(as, aS); LB(bsa, bS, 0); },
             [&] { LB(bsb, bS, 1); });
        LA(am, aM);
        if (G6) {
            const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            as[0] = ks ? zero : as[0];                                       // [a0 | 0]: a0.b1 only
            as[1] = ks ? zero : as[1];
        }
        NBE_SB; MM8(acc1, as, bsa, 0, 0, 0, g, gn, xo, nb, px); NBE_SB;        // a0.b1 + a1.b0
        MM8(acc1, as, bsb, 1, 0, 0, g, gn, xo, nb, px); NBE_SB;
        {
            const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            am[0] = ks ? am[0] : zero;                                       // [0 | a0]
            am[1] = ks ? am[1] : zero;
        }
        NBE_SB; MM8(acc0, am, bsa, 0, 0, 0, g, gn, xo, nb, px, true); NBE_SB;  // a0.b0
        MM8(acc0, am, bsb, 1, 0, 0, g, gn, xo, nb, px, true); NBE_SB;
        __syncthreads();                                         // weight buffer B has landed

        // ======== second stage: taps (5,6) (7,8) from weight buffer B
        LA(a0x, H2_OFF_B + aP); LB(b1, xb + SH5 + bP32 + B1OFF, 0); LB(b0a, xb + SH5 + bP32, 0);
        pair(a0x, 2, g, gn, xo, nb, px, H2_OFF_B, xb + SH5 + bP32,
             [&] { LA(a0y, H2_OFF_B + 2 * H2_TAPU + aP); LB(b1, xb + SH7 + bP1 + B1OFF, 0); },
             [&] { LB(b0a, xb + SH7 + bP1, 0); });
        pair(a0y, 0, g, gn, xo, nb, px, H2_OFF_B + 2 * H2_TAPU, xb + SH7 + bP1, [] {}, [] {});
        __syncthreads();                                         // weight buffer A and the patch of g+1 have landed
    }
#undef NBE_SB
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // MFMA results -> VALU reads of the epilogue

    // ---- epilogue: tile t = 8*mt + nt covers couts 32*it + 16*mt + 4*q .. +3 of position (4*jq + (nt >> 1),
    // 16*(nt & 1) + c).  All global loads first (see conv_h3q_kernel).
    {
        const bool act = a.flags & F_ACT, res = a.flags & F_RES;
        int unit[2];
        bool uok[2];
        f32x4 bv[2], be[2], gv[2];
        const bool gauge = !SPLIT && a.gout != nullptr;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            unit[mt] = ct * 8 + 4 * it + 2 * mt + ks;
            uok[mt] = unit[mt] < a.cout_groups;
            if (!uok[mt]) unit[mt] = a.cout_groups - 1;
            bv[mt] = *(const f32x4*)(a.bias + unit[mt] * 8 + 4 * kh);
            if (G6) be[mt] = *(const f32x4*)(a.beta + unit[mt] * 8 + 4 * kh);
            if (gauge) gv[mt] = *(const f32x4*)(a.gout + unit[mt] * 8 + 4 * kh);
        }
        long o[8];
        bool ook[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const int yy = y0 + 4 * jq + (nt >> 1), xx = x0 + 16 * (nt & 1) + c;
            ook[nt] = yy < a.Hv && xx < a.Wv;
            o[nt] = ook[nt] ? ((long)z * a.Ho + yy) * a.Wo + xx : (long)z * a.Ho * a.Wo;
        }
        constexpr int PARTS = SPLIT ? 2 : 1;
        half4 r0[16], r1[16];                                    // SPLIT: residual hi, lo;  !SPLIT: residual, its tangent
        if (res) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const long rb = ((long)(PARTS * unit[t >> 3]) * a.res_pstride + o[t & 7]) * 16 + 8 * kh;
                r0[t] = *(const half4*)((const char*)a.r + rb);
                r1[t] = SPLIT ? *(const half4*)((const char*)a.r + rb + a.res_pstride * 16)
                              : *(const half4*)((const char*)a.dr + rb);
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int mt = t >> 3, nt = t & 7;
            f32x4 v, dv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc0[t][e] + (SPLIT ? acc1[t][e] * H3_INV : 0.f) + bv[mt][e];
                dv[e] = SPLIT ? 0.f : acc1[t][e];
                if (G6) dv[e] += be[mt][e] * acc0[t][e];
            }
            if (res) {
                if (SPLIT) v += join4(r0[t], r1[t]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += (float)r0[t][e]; dv[e] += (float)r1[t][e]; }
                }
            }
            if (act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!SPLIT) dv[e] = v[e] > 0.f ? dv[e] : 0.01f * dv[e];
                    v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
                }
            }
            if (gauge) {
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[e] += gv[mt][e] * v[e];
            }
            if (uok[mt] && ook[nt]) {
                const long ob = ((long)(a.out_g0 + PARTS * unit[mt]) * a.out_pstride + o[nt]) * 16 + 8 * kh;
                half4 hi, lo;
                split4(v, hi, lo);
                *(half4*)((char*)a.y + ob) = hi;
                if (SPLIT) *(half4*)((char*)a.y + ob + a.out_pstride * 16) = lo;
                else {
                    split4(dv, hi, lo);
                    *(half4*)((char*)a.dy + ob) = hi;
                }
            }
        }
    }
}

template <bool SPLIT, bool G6 = false>
static int launch_h2q(ConvKArgs ka, int ctiles, hipStream_t s) {
    constexpr size_t smem = (size_t)H2_LDS_UNITS * 16;
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    auto kern = conv_h2q_kernel<SPLIT, G6>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    ka.tny = (ka.Hv + H2_ROWS - 1) / H2_ROWS;
    ka.tnx = (ka.Wv + HP_COLS - 1) / HP_COLS;
    ka.ntiles = ka.Dv * ka.tny * ka.tnx;
    if (ctiles != (ka.cout_groups + 7) / 8) return 1;
    dim3 grid(ka.ntiles * ctiles, 1, 1), block(512, 1, 1);
    hipLaunchKernelGGL(kern, grid, block, smem, s, ka);
    return 0;
}

template <int MODE, bool VEL, bool HAS_DX, int XDEPTH, bool SPLIT>
static void launch_h3_t(const ConvKArgs& ka, int ctiles, hipStream_t s) {
    typedef H3Geom<MODE, SPLIT> G;
    constexpr int WB = G::WP * (VEL ? 2 : 1), XB = G::XP * ((VEL && HAS_DX) ? 2 : 1);
    constexpr size_t smem = (size_t)(2 * WB + XDEPTH * XB) * 16 + TILE_VOX * sizeof(int);
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    auto kern = conv_h3_kernel<MODE, VEL, HAS_DX, XDEPTH, SPLIT>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    dim3 grid(ka.ntiles, ctiles, 1), block(512, 1, 1);
    hipLaunchKernelGGL(kern, grid, block, smem, s, ka);
}

template <bool VEL, bool HAS_DX, int SCHED, bool SPLIT>
static void launch_h3p_t(ConvKArgs ka, int ctiles, hipStream_t s) {
    constexpr int UN = SPLIT ? 4 : 2;
    constexpr int WB = 3 * UN * 64 * (VEL ? 2 : 1), XB = ((UN * HP_PL + 63) / 64 * 64) * ((VEL && HAS_DX) ? 2 : 1);
    constexpr size_t smem = (size_t)(2 * WB + 2 * XB) * 16;
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    auto kern = conv_h3p_kernel<VEL, HAS_DX, SCHED, SPLIT>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    ka.tny = (ka.Hv + HP_ROWS - 1) / HP_ROWS;
    ka.tnx = (ka.Wv + HP_COLS - 1) / HP_COLS;
    ka.ntiles = ka.Dv * ka.tny * ka.tnx;
    dim3 grid(ka.ntiles, ctiles, 1), block(512, 1, 1);
    hipLaunchKernelGGL(kern, grid, block, smem, s, ka);
}

template <int SCHED, bool SPLIT, bool VEL, bool HAS_DX>
static void launch_h3p_v(const ConvKArgs& ka, int ct, hipStream_t s) { launch_h3p_t<VEL, HAS_DX, SCHED, SPLIT>(ka, ct, s); }
template <int MODE, int XDEPTH, bool SPLIT, bool VEL, bool HAS_DX>
static void launch_h3_v(const ConvKArgs& ka, int ct, hipStream_t s) { launch_h3_t<MODE, VEL, HAS_DX, XDEPTH, SPLIT>(ka, ct, s); }

// ------------------------------------------------------------------------------------------------
// The first layer (conv_l00/conv_0: Cin = 3, no input tangent; style_nbody_emulator_vel_core.py:132-143) -- f16x3, velocity
// ------------------------------------------------------------------------------------------------
// On the general kernels the three input channels are a 16-channel chunk: K = 27 taps x 16 = 432 where 81 are real, and the
// layer was bound by those MFMAs (43 % matrix-busy, 9.1 ms per launch for a layer that only has to write its output).
// Here K = (tap, channel) = 81, padded to 96 = three k-steps of v_mfma_f32_16x16x32_f16: the workgroup keeps the weights
// ([W hi | W lo | dW hi | dW lo] x 96 x 64 couts = 48 KB) in LDS for its whole life, stages the 3 x 130 x 3 input patch of a
// 1 x 128 output tile as nine small f16 planes per part (part, channel, dz), and every lane GATHERS its B operand
// (8 consecutive k of one position) with 16-bit LDS reads at offsets it computed once.  18 MFMAs per 16 x 16 output tile
// where the general kernel issued 81.  Wave w owns 32 columns of the tile: 64 couts x 32 positions, 128 accumulator registers.
// Workgroups are persistent (grid = 2 per CU, tile = blockIdx.x + n gridDim.x: the tiles in flight are neighbours), patches
// double-buffered: one barrier per tile, the next tile's 12 bytes per patch voxel are fetched under the MFMAs.
// y = W.x + b, dy = dW.x; LeakyReLU, output gauge, hi/lo split and stores as in conv_h3g_kernel.
// Tile shape: ST_ROWS x ST_COLS = 128 positions, four waves of 32 columns each.  The layer is bound by its stores -- 32 output
// planes (64 couts x hi / lo x y / dy) per tile -- and a tile writes ST_COLS * 16 B contiguous bytes per row and plane: the
// longer the runs, the faster (same-device A/B, -DNBE_STEM_ROWS=4 / 2 / 1: 26.9 / 22.9 / 20.6 ms per box, i.e. 3.0 / 3.5 /
// 3.9 TB/s of output; profiles/r02_ab_stem_rows.txt).  One row of 128 columns: 2 KB runs.
#ifndef NBE_STEM_ROWS
#define NBE_STEM_ROWS 1
#endif
constexpr int ST_ROWS = NBE_STEM_ROWS, ST_COLS = 128 / ST_ROWS, ST_RS = ST_COLS + 2, ST_WPR = ST_COLS / 32;   // waves per row
constexpr int ST_PL = (ST_ROWS + 2) * ST_RS;                 // halves of one (part, channel, dz) patch plane: 3 x 130 = 390
constexpr int ST_PV = 3 * ST_PL;                             // patch voxels of a tile: 1170
constexpr int ST_NPV = (ST_PV + 255) / 256;                  // ... per thread
constexpr int ST_PATCH = (6 * ST_PV + 7) / 8 * 8;            // 2 parts x 3 channels x 3 dz planes (7020 halves), rounded to 16 B
constexpr int ST_WU = 4 * 3 * 4 * 64;                        // weight units: [set 4][k-step 3][k-block 4][64 couts] x 8 halves
constexpr int ST_VEC = ST_WU * 4 + ST_PATCH;                  // floats: after the two patch buffers, bias[64] and gout[64]
constexpr size_t ST_LDS = (size_t)ST_WU * 16 + 2 * ST_PATCH * 2 + 2 * 64 * 4;   // 77,760 B: two workgroups per CU

// VEL = false: the displacement-only models' first layer (the W sets only, y only: half the MFMAs and half the stores).
// SPLIT = false: the float16 model (one part: sets [W | dW], three input planes per patch, one MFMA per product and k-step).
template <bool VEL, bool SPLIT = true>
__global__ __launch_bounds__(256, 2) void stem_h3_kernel(ConvKArgs a) {
    constexpr int NSETW = (SPLIT ? 2 : 1) * (VEL ? 2 : 1) * 6;   // weight DMA wave-instructions per wave (6 per set over four waves)
    f32x4* lds = lds_h3;
    const half8* L8 = (const half8*)lds_h3;
    _Float16* P = (_Float16*)(lds_h3 + ST_WU);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4, kh = q & 1, ks = q >> 1;

    // weights: 48 wave-instructions of 1 KB, 12 per wave (displacement only: the first 24, the W sets)
#pragma unroll
    for (int k = 0; k < NSETW / 2; ++k) {
        const int n = wave * (NSETW / 2) + k;
        dma16(a.stem_w + ((long)n * 64 + lane) * 4, lds + n * 64);
    }

    // this lane's B-operand gather: element j of k-step s is k = 32 s + 8 q + j = 3 tap + channel
    int off[3][8];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * s + 8 * q + j;
            const int tap = k / 3, ch = k - 3 * tap, dz = tap / 9, r = tap - 9 * dz, dy = r / 3, dx = r - 3 * dy;
            off[s][j] = k < 81 ? (ch * 3 + dz) * ST_PL + dy * ST_RS + dx : 0;     // k >= 81: zero weights, any finite value
        }

    const int tnx = a.tnx, tny = a.tny, ntiles = a.ntiles;
    // patch voxels of this thread: pv = tid, tid + 256, tid + 512 (< 612)
    int prow[ST_NPV], pcol[ST_NPV], pdz[ST_NPV];
#pragma unroll
    for (int i = 0; i < ST_NPV; ++i) {
        const int pv = min(tid + 256 * i, ST_PV - 1);
        pdz[i] = pv / ST_PL;
        const int r = pv - pdz[i] * ST_PL;
        prow[i] = r / ST_RS; pcol[i] = r - prow[i] * ST_RS;
    }
    half4 ph[ST_NPV], pl[ST_NPV];
    auto fetch = [&](int tile) {                                 // global -> registers: 4 halves (3 channels) of hi and of lo
        const int tx = tile % tnx, t2 = tile / tnx, ty = t2 % tny, z = t2 / tny;
#pragma unroll
        for (int i = 0; i < ST_NPV; ++i) {
            const int yy = min(ty * ST_ROWS + prow[i], a.H - 1), xx = min(tx * ST_COLS + pcol[i], a.W - 1);
            const long v = ((long)(z + pdz[i]) * a.H + yy) * a.W + xx;
            ph[i] = *(const half4*)(a.x + v * 4);
            if (SPLIT) pl[i] = *(const half4*)(a.x + (a.in_pstride + v) * 4);
        }
    };
    auto stage = [&](int buf) {                                  // registers -> LDS planes [part][channel][dz][row][col]
        _Float16* Pb = P + buf * ST_PATCH;
#pragma unroll
        for (int i = 0; i < ST_NPV; ++i) {
            const int pv = tid + 256 * i;
            if (pv < ST_PV) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    Pb[ch * ST_PV + pv] = ph[i][ch];
                    if (SPLIT) Pb[(3 + ch) * ST_PV + pv] = pl[i][ch];
                }
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    const bool act = a.flags & F_ACT, gauge = a.gout != nullptr;
    // the per-channel vectors live in LDS for the life of the (persistent) workgroup: fetched from memory inside the
    // epilogue, each load waits for the stores issued before it (vmcnt counts both) -- four store drains per tile
    float* Lvec = (float*)lds_h3 + ST_VEC;
    if (tid < 64) Lvec[tid] = tid < 8 * a.cout_groups ? a.bias[tid] : 0.f;
    else if (tid < 128) Lvec[tid] = (gauge && tid - 64 < 8 * a.cout_groups) ? a.gout[tid - 64] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the weights (and the first patch)
    const int wrow = wave / ST_WPR, wcol = (wave % ST_WPR) * 32;
    const int rb = wrow * ST_RS + wcol + c;
    for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        stage(buf);
        __syncthreads();
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) fetch(nxt);

        f32x4 ym[8], yc[8], dm[8], dc[8];                        // tile t = 2 mt + nt
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) { ym[t][e] = 0.f; yc[t][e] = 0.f; dm[t][e] = 0.f; dc[t][e] = 0.f; }
        const _Float16* Pb = P + buf * ST_PATCH + rb;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            half8 xh[2], xl[2], A[4];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    xh[nt][j] = Pb[off[s][j] + 16 * nt];
                    if (SPLIT) xl[nt][j] = Pb[off[s][j] + 16 * nt + 3 * ST_PV];
                }
            auto LA = [&](int set) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) A[mt] = L8[((set * 3 + s) * 4 + q) * 64 + 16 * mt + c];
            };
            auto MM = [&](f32x4 (&acc)[8], const half8 (&B)[2]) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t >> 1], B[t & 1], acc[t], 0, 0, 0);
            };
            if (!SPLIT) {                                         // float16 model: sets W, dW
                LA(0); MM(ym, xh);
                if (VEL) { LA(1); MM(dm, xh); }
                continue;
            }
            LA(0); MM(yc, xl); MM(ym, xh);                        // W hi
            LA(1); MM(yc, xh);                                    // W lo
            if (VEL) {
                LA(2); MM(dc, xl); MM(dm, xh);                    // dW hi
                LA(3); MM(dc, xh);                                // dW lo
            }
        }

        // epilogue: lane (c, q) holds couts 16 mt + 4 q + e of position (row wave, col 16 nt + c)
        const int tx = tile % tnx, t2 = tile / tnx, ty = t2 % tny, z = t2 / tny;
        const int yy = ty * ST_ROWS + wrow;
        long o[2];
        bool ook[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int xx = tx * ST_COLS + wcol + 16 * nt + c;
            ook[nt] = yy < a.Hv && xx < a.Wv;
            o[nt] = ook[nt] ? ((long)z * a.Ho + yy) * a.Wo + xx : 0;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int unit = 2 * mt + ks;
            const bool uok = unit < a.cout_groups;
            if (!uok) unit = a.cout_groups - 1;
            const f32x4 bv = *(const f32x4*)(Lvec + unit * 8 + 4 * kh);
            const f32x4 gv = *(const f32x4*)(Lvec + 64 + unit * 8 + 4 * kh);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int t = 2 * mt + nt;
                f32x4 v, dv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = ym[t][e] + (SPLIT ? yc[t][e] * H3_INV : 0.f) + bv[e];
                    dv[e] = dm[t][e] + (SPLIT ? dc[t][e] * H3_INV : 0.f);
                }
                if (act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dv[e] = v[e] > 0.f ? dv[e] : 0.01f * dv[e];
                        v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
                    }
                }
                if (gauge) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[e] += gv[e] * v[e];
                }
                if (!SPLIT) {                                    // one f16 plane per unit
                    if (uok && ook[nt]) {
                        const long ob = ((long)(a.out_g0 + unit) * a.out_pstride + o[nt]) * 16 + 8 * kh;
                        half4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
                        *(half4*)((char*)a.y + ob) = h;
                        if (VEL) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[e] = (_Float16)dv[e];
                            *(half4*)((char*)a.dy + ob) = h;
                        }
                    }
                } else
                if (uok && ook[nt]) {
                    const long ob = ((long)(a.out_g0 + 2 * unit) * a.out_pstride + o[nt]) * 16 + 8 * kh;
                    const long ol = ob + a.out_pstride * 16;
                    half4 hi, lo;
                    split4(v, hi, lo);
                    *(half4*)((char*)a.y + ob) = hi;
                    *(half4*)((char*)a.y + ol) = lo;
                    if (VEL) {
                        split4(dv, hi, lo);
                        *(half4*)((char*)a.dy + ob) = hi;
                        *(half4*)((char*)a.dy + ol) = lo;
                    }
                }
            }
        }
    }
}

template <bool VEL, bool SPLIT = true>
static int launch_stem(ConvKArgs ka, hipStream_t s) {
    ka.tny = (ka.Hv + ST_ROWS - 1) / ST_ROWS;
    ka.tnx = (ka.Wv + ST_COLS - 1) / ST_COLS;
    const long nt = (long)ka.Dv * ka.tny * ka.tnx;
    if (nt <= 0 || nt >= (1L << 31) || ka.cout_groups > 8 || ka.nchunk != 1) return 1;
    ka.ntiles = (int)nt;
    const int grid = (int)std::min<long>(nt, 512);
    static_assert(2 * ST_LDS <= 160 * 1024, "two workgroups per CU");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)stem_h3_kernel<VEL, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ST_LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL((stem_h3_kernel<VEL, SPLIT>), dim3(grid), dim3(256), ST_LDS, s, ka);
    return 0;
}

// stem_w: [set: W hi, W lo, dW hi, dW lo][k-step 3][k-block 4][cout 64][j 8], k = 32 s + 8 q + j = 3 tap + channel
__global__ __launch_bounds__(256) void pack_stem_kernel(const float* __restrict__ w, int cout, int cin, _Float16* __restrict__ dst, int parts) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // over [part 2 (float16 model: 1)][s 3][q 4][co 64][j 8]
    if (idx >= parts * 3 * 4 * 64 * 8) return;
    const int j = idx & 7, co = (idx >> 3) & 63, qq = (idx >> 9) & 3, sp = idx >> 11, st = sp % 3, part = sp / 3;
    const int k = 32 * st + 8 * qq + j, tap = k / 3, ch = k - 3 * tap;
    float v = 0.f;
    if (k < 81 && ch < cin && co < cout) v = w[((size_t)co * cin + ch) * 27 + tap];
    const _Float16 hi = (_Float16)v;
    dst[idx] = part == 0 ? hi : (_Float16)((v - (float)hi) * H3_SCALE);
}

// Which kernel runs a layer.  Generations that no default or fall-back reaches any more -- the flat tiling of the 3x3x3
// layers, its 3-deep ring, the de-bursted DMA issue, the flat first layer, the 4 x 4 wave tile with one wave per SIMD
// (DESIGN.md sections 4 and 4c have their measurements) -- are instantiated in timing-probe builds (-DNBE_DBG=1) only, behind
// their old switches (NBE_H3_FLAT, NBE_H3_DEPTH, NBE_H3_SCHED, NBE_L0_FLAT, NBE_H3G_BIG).
int launch_conv_h3(const PackedW& pw, const ConvKArgs& ka, bool vel, bool has_dx, hipStream_t s) {
    const int ct = pw.ctiles;
#if NBE_DBG
    static const int depth = (getenv("NBE_H3_DEPTH") && atoi(getenv("NBE_H3_DEPTH")) == 3) ? 3 : 2;
    static const bool flat3 = getenv("NBE_H3_FLAT") && atoi(getenv("NBE_H3_FLAT")) != 0;   // A/B: flat 3x3x3 tiling
    static const int sched = (getenv("NBE_H3_SCHED") && atoi(getenv("NBE_H3_SCHED")) == 1) ? 1 : 0;
    static const bool l0_flat = getenv("NBE_L0_FLAT") && atoi(getenv("NBE_L0_FLAT")) == 1;
#else
    constexpr int depth = 2, sched = 0;
    constexpr bool flat3 = false, l0_flat = false;
#endif
    const bool split = pw.prec == PREC_F16X3;
    const bool stem_on = !(getenv("NBE_STEM") && atoi(getenv("NBE_STEM")) == 0);   // A/B switch, default on (read per launch: tests flip it)
    if (stem_on && ka.stem_w && !(vel && has_dx) && pw.mode == MODE_FLAT3 && ka.in_off == 0 && ka.osz == 1 &&
        !(ka.flags & F_RES) && ka.nskip == 0 && !ka.beta) {
        if (split) return vel ? launch_stem<true>(ka, s) : launch_stem<false>(ka, s);
        return vel ? launch_stem<true, false>(ka, s) : launch_stem<false, false>(ka, s);
    }
    // a second input segment / a fused skip exist only in the wide gauged f16x3 kernel and in conv_h3w_kernel's displacement-only form
    if ((ka.nskip > 0 || ka.csplit < ka.nchunk) && !(ka.beta && split)) {
        if (split && !vel && ka.ww && pw.mode == MODE_FLAT3 && ka.in_off == 0 && ka.osz == 1)
            return launch_h3w(ka, ka.ww, ka.wws, ka.wws_set_floats, ct, s, true);     // 1: no such form for this launch (an engine bug)
        if (!split && vel && has_dx && ka.beta && ka.ww && pw.mode == MODE_FLAT3 && ka.in_off == 0 && ka.osz == 1)
            return launch_h3w(ka, ka.ww, ka.wws, ka.wws_set_floats, ct, s, false, true);   // the float16 model's Winograd-z form
        return 1;
    }
#define NBE_VD(F, ...)                                                          \
    if (vel) { if (has_dx) F<__VA_ARGS__, true, true>(ka, ct, s); else F<__VA_ARGS__, true, false>(ka, ct, s); } \
    else F<__VA_ARGS__, false, false>(ka, ct, s);
    static const bool shape32 = getenv("NBE_H3_SHAPE") && atoi(getenv("NBE_H3_SHAPE")) == 32;   // A/B: 32x32x16 MFMAs
    if (ka.beta) {                                               // gauged input tangent: only conv_h3g_kernel reads it
        if (!(pw.mode == MODE_FLAT3 && vel && has_dx && ka.in_off == 0 && ka.osz == 1)) return 1;   // no gauged kernel
        if (split) {
            const bool tall = !(getenv("NBE_H3G_TALL") && atoi(getenv("NBE_H3G_TALL")) == 0);   // A/B switch, default on (read per launch)
            if (pw.cout_t == 16) {                              // the head convolution: four output planes per workgroup where the launch allows
                const bool head4 = !(getenv("NBE_HEAD4") && atoi(getenv("NBE_HEAD4")) == 0);   // A/B switch, default on (read per launch)
                if (head4 && launch_h3n4(ka, ct, s) == 0) return 0;
                return launch_h3g<true, false>(ka, ct, s);
            }
#if NBE_DBG
            if (getenv("NBE_H3G_BIG") && atoi(getenv("NBE_H3G_BIG")) == 1) return launch_h3g<false, false, true>(ka, ct, s);   // 4 x 4 wave tile, one wave per SIMD
#endif
            if (ka.ww && launch_h3w(ka, ka.ww, ka.wws, ka.wws_set_floats, ct, s) == 0) return 0;     // Winograd along z; 1: no such form for this launch
            return tall ? launch_h3g<false, true>(ka, ct, s) : launch_h3g<false, false>(ka, ct, s);
        }
        if (ka.ww && launch_h3w(ka, ka.ww, ka.wws, ka.wws_set_floats, ct, s, false, true) == 0) return 0;   // float16 model: Winograd along z
        return launch_h2q<false, true>(ka, ct, s);
    }
    const bool first_flat = l0_flat && split && vel && !has_dx && pw.mode == MODE_FLAT3 && ka.nchunk == 1;
    if (pw.mode == MODE_FLAT3 && !flat3 && !first_flat) {
        if (!(ka.in_off == 0 && ka.osz == 1)) return 1;          // 3x3x3 layers are never cropped or strided
        if (split && vel && has_dx && !shape32 && sched == 0) return launch_h3q(ka, ct, s);
        if (split && !vel && ka.ww && launch_h3w(ka, ka.ww, ka.wws, ka.wws_set_floats, ct, s, true) == 0) return 0;   // Winograd along z, displacement only
        if (split && !vel && !shape32) return launch_h2q<true>(ka, ct, s);
        if (!split && vel && has_dx && !shape32) return launch_h2q<false>(ka, ct, s);
        if (!split) { NBE_VD(launch_h3p_v, 0, false) }
#if NBE_DBG
        else if (sched == 1) { NBE_VD(launch_h3p_v, 1, true) }
#endif
        else { NBE_VD(launch_h3p_v, 0, true) }
        return 0;
    }
    if (pw.mode == MODE_FLAT3) {
#if NBE_DBG
        if (!split) { NBE_VD(launch_h3_v, MODE_FLAT3, 2, false) }
        else if (depth == 3) { NBE_VD(launch_h3_v, MODE_FLAT3, 3, true) }
        else { NBE_VD(launch_h3_v, MODE_FLAT3, 2, true) }
#else
        return 1;
#endif
    } else if (pw.mode == MODE_FLAT1) {
        if (ka.up8) {                                            // all eight parities in one launch
            if (vel && has_dx) return split ? launch_up_h3<true>(ka, ct, s) : launch_up_h3<false>(ka, ct, s);
            if (!vel) return split ? launch_up_h3<true, false>(ka, ct, s) : launch_up_h3<false, false>(ka, ct, s);
            return 1;
        }
        if (!split) { NBE_VD(launch_h3_v, MODE_FLAT1, 2, false) } else { NBE_VD(launch_h3_v, MODE_FLAT1, 2, true) }
    } else {
        if (!split) { NBE_VD(launch_h3_v, MODE_DOWN, 2, false) } else { NBE_VD(launch_h3_v, MODE_DOWN, 2, true) }
    }
#undef NBE_VD
    (void)depth;
    return 0;
}

// packed layout: [set][ct][stage = chunk*nseg + seg][tap][u = 2*h + part][co cout_t = 64 (16: narrow tiles)][j 8];
// channel = chunk*16 + 8*h + j; part 0 = hi, 1 = lo * 2^11
__global__ __launch_bounds__(256) void pack_h3_kernel(const float* __restrict__ w, int cout, int cin, int kind,
                                                      int mode, int nchunk, long halves_per_set, int nsets,
                                                      int parts, int cout_t, _Float16* __restrict__ dst,
                                                      float wscale = 0.f, int* __restrict__ flag = nullptr, int f16w = 0) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= halves_per_set * nsets) return;
    const int TAPS = mode_taps(mode), nseg = mode_nseg(mode);
    const int nstage = nseg * nchunk;
    long r = idx;
    const int j = (int)(r % 8); r /= 8;
    const int co = (int)(r % cout_t); r /= cout_t;
    const int un = 2 * parts;
    const int u = (int)(r % un); r /= un;
    const int tap = (int)(r % TAPS); r /= TAPS;
    const int stage = (int)(r % nstage); r /= nstage;
    const long per_set_ct = halves_per_set / ((long)nstage * TAPS * un * cout_t * 8);
    const int ct = (int)(r % per_set_ct); r /= per_set_ct;
    const int set = (int)r;
    const int chunk = stage / nseg, seg = stage - chunk * nseg;
    const int h = u / parts, part = u - h * parts;
    // f16w: conv_h3w_kernel's float16 form -- a chunk is 32 channels, unit u holds channels 8 u .. 8 u + 7 (no lo part)
    const int ci = f16w ? chunk * 32 + 8 * u + j : chunk * 16 + 8 * h + j;
    const int oc = ct * cout_t + co;
    int k, kz, ky, kx;
    if (kind == 0) { k = 3; kz = seg / 3; ky = seg % 3; kx = tap; }
    else if (kind == 1) { k = 1; kz = ky = kx = 0; }
    else if (kind == 2) { k = 2; kz = seg >> 2; ky = (seg >> 1) & 1; kx = seg & 1; }
    else { k = 2; kz = 1 - ((set >> 2) & 1); ky = 1 - ((set >> 1) & 1); kx = 1 - (set & 1); }
    float v = 0.f;
    if (oc < cout && ci < cin) v = w[(((size_t)oc * cin + ci) * k + kz) * k * k + ky * k + kx];
    if (wscale != 0.f) {                                         // conv_h3w_kernel's form: value * 2^14, lo = the unscaled remainder
        v *= wscale;
        if (!(fabsf(v) <= 60000.f)) { atomicOr(flag, 1); v = 0.f; }
        const _Float16 h = (_Float16)v;
        dst[idx] = (part == 0 || f16w) ? h : (_Float16)(v - (float)h);
        return;
    }
    const _Float16 hi = (_Float16)v;
    dst[idx] = part == 0 ? hi : (_Float16)((v - (float)hi) * H3_SCALE);
}

// a fused skip's weights for conv_h3w_kernel: the FLAT1 packing of `pw` with the kernel's 2^14 scale (dst: pw.floats floats)
void launch_pack_h3w_skip(const float* w_oidhw, int cout, int cin, const PackedW& pw, float* dst, int* flag, hipStream_t s) {
    const long halves = pw.floats * 2;
    if (pw.prec == PREC_F16) {                                   // 32-channel chunks of four units, scale 2^8 (WINO_WSCALE_F16)
        hipLaunchKernelGGL(pack_h3_kernel, dim3((unsigned)((halves + 255) / 256)), dim3(256), 0, s, w_oidhw, cout, cin,
                           1, pw.mode, pw.cin_pad / 32, halves, 1, 2, pw.cout_t, (_Float16*)dst, 256.0f, flag, 1);
        return;
    }
    hipLaunchKernelGGL(pack_h3_kernel, dim3((unsigned)((halves + 255) / 256)), dim3(256), 0, s, w_oidhw, cout, cin,
                       1, pw.mode, pw.cin_pad / 16, halves, 1, 2, pw.cout_t, (_Float16*)dst, 16384.0f, flag);
}

void launch_pack_h3(const float* w_oidhw, int cout, int cin, int kind, const PackedW& pw, float* dst, hipStream_t s) {
    if (pw.stem && kind == 0 && (dst == pw.w || dst == pw.dw)) {    // the first layer's own format beside the general one
        const int parts = pw.prec == PREC_F16X3 ? 2 : 1;
        hipLaunchKernelGGL(pack_stem_kernel, dim3(48), dim3(256), 0, s, w_oidhw, cout, cin,
                           (_Float16*)pw.stem + (dst == pw.dw ? parts * 3 * 4 * 64 * 8 : 0), parts);
    }
    const long halves = pw.floats * 2;
    const long total = halves * pw.nsets;
    hipLaunchKernelGGL(pack_h3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_oidhw, cout, cin,
                       kind, pw.mode, pw.cin_pad / 16, halves, pw.nsets, pw.prec == PREC_F16X3 ? 2 : 1, pw.cout_t,
                       (_Float16*)dst);
}

// ------------------------------------------------------------------------------------------------
// data movement in the 8 x f16 plane formats: parts = 2 (f16x3: plane 2g = hi, 2g+1 = lo) or 1 (plain f16)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ half8 ld8(const float* base, long plane, long pstride, long v) {
    return *(const half8*)(base + (plane * pstride + v) * 4);
}
__device__ __forceinline__ float join1(const half8& hi, const half8& lo, int e, int parts) {
    return parts == 2 ? (float)hi[e] + (float)lo[e] * H3_INV : (float)hi[e];
}

__global__ __launch_bounds__(256) void gather_h8_kernel(const float* __restrict__ box, int C, int Db, int Hb, int Wb,
                                                        int a0, int a1, int a2, float* __restrict__ dst, long pstride,
                                                        int G, int D, int H, int W, float scale, int parts) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long V = (long)D * H * W;
    if (v >= V) return;
    const int z = (int)(v / ((long)H * W)), rem = (int)(v - (long)z * H * W);
    const int y = rem / W, x = rem - y * W;
    const int bz = ((a0 + z) % Db + Db) % Db, by = ((a1 + y) % Hb + Hb) % Hb, bx = ((a2 + x) % Wb + Wb) % Wb;
    const long bo = ((long)bz * Hb + by) * Wb + bx;
    const long bstride = (long)Db * Hb * Wb;
    for (int g = 0; parts * g < G; ++g) {
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * g + e;
            const float f = c < C ? box[c * bstride + bo] * scale : 0.f;
            hi[e] = (_Float16)f;
            lo[e] = (_Float16)((f - (float)hi[e]) * H3_SCALE);
        }
        *(half8*)(dst + ((long)(parts * g) * pstride + v) * 4) = hi;
        if (parts == 2) *(half8*)(dst + ((long)(2 * g + 1) * pstride + v) * 4) = lo;
    }
}

void launch_gather_h8(const float* box, int C, int Db, int Hb, int Wb, int a0, int a1, int a2,
                      float* dst, const Planes& geom, float scale, int parts, hipStream_t s) {
    const long V = geom.vox();
    hipLaunchKernelGGL(gather_h8_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, s, box, C, Db, Hb, Wb,
                       a0, a1, a2, dst, geom.pstride, geom.G, geom.D, geom.H, geom.W, scale, parts);
}

__global__ __launch_bounds__(256) void from_planes_h8_kernel(const float* __restrict__ src, long pstride, long V, int C,
                                                             float* __restrict__ dst, int parts) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    for (int g = 0; 8 * g < C; ++g) {
        const half8 hi = ld8(src, parts * g, pstride, v);
        const half8 lo = parts == 2 ? ld8(src, 2 * g + 1, pstride, v) : hi;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (8 * g + e < C) dst[(long)(8 * g + e) * V + v] = join1(hi, lo, e, parts);
    }
}

void launch_from_planes_h8(const float* src, const Planes& geom, int C, float* dst, int parts, hipStream_t s) {
    const long V = geom.vox();
    hipLaunchKernelGGL(from_planes_h8_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, s, src, geom.pstride, V, C,
                       dst, parts);
}

template <typename OT>
__global__ __launch_bounds__(256) void head_h8_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                      long ypstride, int D, int H, int W,
                                                      const float* __restrict__ xin, long xpstride, int XH, int XW,
                                                      int c0, int C, float k_disp, float k_dy, float k_x0,
                                                      OT* __restrict__ disp, OT* __restrict__ velo, int Db, int Hb,
                                                      int Wb, int a0, int a1, int a2, int parts, int pad,
                                                      int* __restrict__ bad) {
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
    for (int g = 0; 8 * g < C; ++g) {
        const half8 yh = ld8(y, parts * g, ypstride, v), xh = ld8(xin, parts * g, xpstride, xv);
        const half8 yl = parts == 2 ? ld8(y, 2 * g + 1, ypstride, v) : yh;
        const half8 xl = parts == 2 ? ld8(xin, 2 * g + 1, xpstride, xv) : xh;
        half8 dh = yh, dl = yh;
        if (dy) { dh = ld8(dy, parts * g, ypstride, v); dl = parts == 2 ? ld8(dy, 2 * g + 1, ypstride, v) : dh; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * g + e;
            if (c < C) {
                const float yv = join1(yh, yl, e, parts), x0 = join1(xh, xl, e, parts);
                const float dsp = (yv + x0) * k_disp;
                disp[c * bstride + bo] = (OT)dsp;
                nf = nf || !isfinite(dsp);
                if (dy) {
                    const float vv = join1(dh, dl, e, parts) * k_dy + x0 * k_x0;
                    velo[c * bstride + bo] = (OT)vv;
                    nf = nf || !isfinite(vv);
                }
            }
        }
    }
    // an operand beyond the f16 range became an infinity on its way through the network (see nbe.h, "Range")
    if (nf && bad) atomicOr(bad, 1);
}

void launch_head_h8(const Planes& y, const Planes& xin, int c0, int C, const HeadScale& hs, bool vel,
                    void* disp, void* velo, int out_dtype, int Db, int Hb, int Wb, int a0, int a1, int a2,
                    int parts, hipStream_t s, int pad) {
    const long V = (long)y.D * (y.H - 2 * pad) * (y.W - 2 * pad);
    dim3 grid((unsigned)((V + 255) / 256)), block(256);
    if (out_dtype == 0)
        hipLaunchKernelGGL(head_h8_kernel<float>, grid, block, 0, s, y.x, vel ? y.dx : nullptr, y.pstride, y.D, y.H,
                           y.W, xin.x, xin.pstride, xin.H, xin.W, c0, C, hs.k_disp, hs.k_dy, hs.k_x0, (float*)disp,
                           (float*)velo, Db, Hb, Wb, a0, a1, a2, parts, pad, hs.bad);
    else
        hipLaunchKernelGGL(head_h8_kernel<_Float16>, grid, block, 0, s, y.x, vel ? y.dx : nullptr, y.pstride, y.D,
                           y.H, y.W, xin.x, xin.pstride, xin.H, xin.W, c0, C, hs.k_disp, hs.k_dy, hs.k_x0,
                           (_Float16*)disp, (_Float16*)velo, Db, Hb, Wb, a0, a1, a2, parts, pad, hs.bad);
}

}  // namespace nbe
