// Included by nbe_kernels_h3.hip (uses its patch constants, dma16s, split4, xcd_tile).
//
// conv_h3n4_kernel: the gauged 3x3x3 layer on the narrow tile (16 couts: the head convolution conv_r01/conv_1, 64 -> 3,
// style_nbody_emulator_vel_core.py:178-186) with FOUR OUTPUT PLANES PER WORKGROUP.
//
// conv_h3g_kernel<NARROW> gives a workgroup one output plane: per 16-channel chunk it stages three input patches (dz = 0, 1,
// 2; 44 KB each) for 2 x 28 MFMAs per wave -- the variant is bound by the L2 -> LDS stream, not by its MFMAs (34 % matrix-busy,
// 54 GB read per launch).  Here a workgroup owns planes z0 .. z0 + 3 of an 8 x 32 patch: the SIX input
// planes z0 .. z0 + 5 of a chunk are staged once each and input plane p feeds output plane p - dz for every dz in range --
// twelve (plane, dz) products from six patches instead of twelve, and the three dz weight groups of a chunk (27 KB) are staged
// once per chunk instead of once per plane.  Arithmetic, operand layout in LDS, tap pairing and epilogue are those of
// conv_h3g_kernel<NARROW> (main / correction accumulators of the f16x3 split, beta and gauge in the epilogue), so the fields
// are the same to the last bit; the block's fused 1x1x1 skip runs as one stage per (chunk, output plane) after the 3x3x3
// stages.  The MFMAs are compiler intrinsics here (no hand-pinned accumulators: 128 accumulator registers and ~40 operand
// registers leave the allocator room), so the VALU -> MFMA hazards are the compiler's to handle.
//
// A stage is (chunk, input plane); stage s + 1 is fetched by global -> LDS DMA at the start of stage s into the other patch
// buffer (and, when a new chunk begins, its weights into the other weight buffer); one barrier per stage.
constexpr int HN4_ZB = 4;                                       // output planes per workgroup
constexpr int HN4_CT = 16, HN4_TAPU = 4 * HN4_CT;               // couts per tile; 16-byte units per tap (2 channel halves x hi/lo)
constexpr int HN4_WG = 9 * HN4_TAPU;                            // one (chunk, dz) group: 576 units
constexpr int HN4_WC = 3 * HN4_WG;                              // a chunk's three groups: 1728 units = 27 KB
constexpr int HN4_XBASE = 2 * HN4_WC;
constexpr int HN4_LDS_UNITS = HN4_XBASE + 2 * HQ_XB;            // 8960 units = 143,360 B

__global__ __launch_bounds__(512, 1) void conv_h3n4_kernel(ConvKArgs a) {
    constexpr int ZB = HN4_ZB, CT = HN4_CT, TAPU = HN4_TAPU, WG = HN4_WG, WC = HN4_WC, XBASE = HN4_XBASE, NT = 2, NW = 8;
    f32x4* lds = lds_h3;
    const half8* L8 = (const half8*)lds_h3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4, kh = q & 1, ks = q >> 1;

    // tiles: blocks of ZB planes fastest (neighbours along z share two of their six input planes through the XCD's L2)
    const int nzb = a.Dv / ZB;
    const int tile = xcd_tile(blockIdx.x, a.ntiles);
    const int zb = tile % nzb, tyx = tile / nzb;
    const int ty = tyx / a.tnx, tx = tyx - ty * a.tnx;
    const int y0 = ty * HP_ROWS, x0 = tx * HP_COLS, z0 = zb * ZB;
    const int nchunk = a.nchunk, nskip = a.nskip;
    const int nmain = nchunk * (ZB + 2), nstage = nmain + nskip * ZB;
    const long plane = (long)a.H * a.W * 16;
    const long to = (((long)z0 * a.H + y0) * a.W + x0) * 16;
    const unsigned lane16 = (unsigned)lane * 16u;

    // per-lane offsets of the patch DMA: 24 wave-instructions per tensor (4 planes x 6), three per wave
    unsigned xoff[3];
    bool xval[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int k = (wave + NW * t) % 6;
        const int u = k * 64 + lane;
        xval[t] = u < HP_PL;
        const int uu = xval[t] ? u : HP_PL - 1;
        const int row = uu / HP_RS, col = uu - row * HP_RS;
        xoff[t] = (unsigned)(row * a.W + col) * 16u;
    }
    // stage s: sources (a.gs[chunk] = the chunk's planes at z = 0 of the layer input; a.gs[nchunk + sc] = the skip's)
    auto fetch = [&](int s) {
        const int buf = s & 1;
        int chunk, p;
        bool skip = s >= nmain;
        if (!skip) { chunk = s / (ZB + 2); p = s - chunk * (ZB + 2); }
        else { const int r = s - nmain; chunk = r / ZB; p = r - chunk * ZB; }
        const ConvGroupSrc e = a.gs[skip ? nchunk + chunk : chunk];
        const char* xs = e.x + to + (long)p * plane;
        const char* dxs = e.dx + to + (long)p * plane;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int n = wave + NW * t, pl = n / 6, k = n - 6 * pl;
            if (xval[t]) {
                dma16s(xs + (long)pl * e.psb, xoff[t], lds + XBASE + buf * HQ_XB + pl * HQ_PP + k * 64);
                dma16s(dxs + (long)pl * e.psb, xoff[t], lds + XBASE + buf * HQ_XB + HQ_XT + pl * HQ_PP + k * 64);
            }
        }
        if (p == 0) {                                            // a new chunk: its weights into the other weight buffer
            const int wbuf = (skip ? nchunk + chunk : chunk) & 1;
            if (!skip) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {                    // 27 wave-instructions
                    const int n = wave + NW * t;
                    if (n < WC / 64) dma16s(e.w + (long)n * 1024, lane16, lds + wbuf * WC + n * 64);
                }
            } else if (wave < 2) {                               // W_s (wave 0) and dW_s~ (wave 1) of the chunk: 64 units each
                dma16s(e.w + (wave ? a.dws_delta : 0), lane16, lds + wbuf * WC + wave * 64);
            }
        }
    };

    f32x4 ym[ZB][NT], yc[ZB][NT], dm[ZB][NT], dc[ZB][NT];
#pragma unroll
    for (int z = 0; z < ZB; ++z)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) { ym[z][t][e] = 0.f; yc[z][t][e] = 0.f; dm[z][t][e] = 0.f; dc[z][t][e] = 0.f; }
    auto mm = [&](f32x4& acc, const half8& A, const half8& B) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc, 0, 0, 0); };

    const int rowp = wave;                                       // this wave's row of the 8 x 32 patch
    const int aP = (ks * 4 + 2 * kh) * CT + c;
    const int bB = (2 * kh) * HQ_PP + rowp * HP_RS + c;
    const int bP1 = bB + ks, bP32 = bB + 32 * ks;
    constexpr int SH4 = HP_RS + 1, SH5 = HP_RS + 2, SH7 = 2 * HP_RS + 1;
    const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};

    // the nine taps of one (chunk, dz) group into the accumulators of one output plane: four tap pairs and the odd tap
    auto group = [&](f32x4 (&Ym)[NT], f32x4 (&Yc)[NT], f32x4 (&Dm)[NT], f32x4 (&Dc)[NT], int wb, int xb) {
        auto pair = [&](int wa, int xp) {
            const half8 wh = L8[wa + aP], wl = L8[wa + CT + aP];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const half8 xh = L8[xp + 16 * nt], xl = L8[xp + HQ_PP + 16 * nt];
                const half8 dxh = L8[xp + HQ_XT + 16 * nt], dxl = L8[xp + HQ_XT + HQ_PP + 16 * nt];
                mm(Yc[nt], wh, xl); mm(Ym[nt], wh, xh); mm(Dm[nt], wh, dxh);
                mm(Dc[nt], wh, dxl); mm(Yc[nt], wl, xh); mm(Dc[nt], wl, dxh);
            }
            __builtin_amdgcn_sched_barrier(0);                   // one pair's operands at a time (or the scheduler hoists every LDS read of a stage)
        };
        pair(wb, xb + bP1);                                      // taps (0,1)
        pair(wb + 2 * TAPU, xb + 2 + bP32);                      // taps (2,3)
        {                                                        // tap 4: the K halves select the PART: [wh|wl].[xl|xh], [0|wh].[xl|xh]
            const half8 a1w = L8[wb + 4 * TAPU + (2 * kh + ks) * CT + c];
            half8 a0 = L8[wb + 4 * TAPU + (2 * kh) * CT + c];
            a0 = ks ? a0 : zero;
            const int bS1 = xb + (2 * kh + 1 - ks) * HQ_PP + rowp * HP_RS + c + SH4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const half8 b1x = L8[bS1 + 16 * nt], b1d = L8[bS1 + HQ_XT + 16 * nt];
                mm(Yc[nt], a1w, b1x); mm(Ym[nt], a0, b1x); mm(Dc[nt], a1w, b1d); mm(Dm[nt], a0, b1d);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pair(wb + 5 * TAPU, xb + SH5 + bP32);                    // taps (5,6)
        pair(wb + 7 * TAPU, xb + SH7 + bP1);                     // taps (7,8)
    };
    // the fused skip's chunk on the centre tap of the block input's patch (conv_h3g_kernel's skip body):
    // y += W_s.x, dy += W_s.dx~ + dW_s~.x
    auto skipgroup = [&](f32x4 (&Ym)[NT], f32x4 (&Yc)[NT], f32x4 (&Dm)[NT], f32x4 (&Dc)[NT], int wb, int xb) {
        const half8 a1w = L8[wb + (2 * kh + ks) * CT + c], a1d = L8[wb + 4 * CT + (2 * kh + ks) * CT + c];
        half8 a0 = L8[wb + (2 * kh) * CT + c], a0d = L8[wb + 4 * CT + (2 * kh) * CT + c];
        a0 = ks ? a0 : zero; a0d = ks ? a0d : zero;
        const int bS1 = xb + (2 * kh + 1 - ks) * HQ_PP + rowp * HP_RS + c + SH4;
        const bool nodx = a.flags & F_SKIP_NODX;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const half8 b1x = L8[bS1 + 16 * nt], b1d = L8[bS1 + HQ_XT + 16 * nt];
            mm(Yc[nt], a1w, b1x); mm(Ym[nt], a0, b1x);
            if (!nodx) { mm(Dc[nt], a1w, b1d); mm(Dm[nt], a0, b1d); }
            mm(Dc[nt], a1d, b1x); mm(Dm[nt], a0d, b1x);
        }
    };

    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int wb = (chunk & 1) * WC;
#pragma unroll
        for (int p = 0; p < ZB + 2; ++p) {
            const int s = chunk * (ZB + 2) + p;
            if (s + 1 < nstage) fetch(s + 1);
            const int xb = XBASE + (s & 1) * HQ_XB;
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                const int zo = p - dz;                           // compile-time after unrolling
                if (zo >= 0 && zo < ZB) group(ym[zo], yc[zo], dm[zo], dc[zo], wb + dz * WG, xb);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    for (int sc = 0; sc < nskip; ++sc) {
        const int wb = ((nchunk + sc) & 1) * WC;
#pragma unroll
        for (int zo = 0; zo < ZB; ++zo) {
            const int s = nmain + sc * ZB + zo;
            if (s + 1 < nstage) fetch(s + 1);
            skipgroup(ym[zo], yc[zo], dm[zo], dc[zo], wb, XBASE + (s & 1) * HQ_XB);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- epilogue (conv_h3g_kernel's): y = W.x + b, dy = W.dx~ + beta * (W.x), LeakyReLU (+ tangent), gauge, split, store
    const bool act = a.flags & F_ACT, gauge = a.gout != nullptr;
    int unit = ks;                                               // one cout tile: units 0 / 1 (8 couts each)
    const bool uok = unit < a.cout_groups;
    if (!uok) unit = a.cout_groups - 1;
    const f32x4 bv = *(const f32x4*)(a.bias + unit * 8 + 4 * kh);
    const f32x4 be = *(const f32x4*)(a.beta + unit * 8 + 4 * kh);
    f32x4 gv = {0.f, 0.f, 0.f, 0.f};
    if (gauge) gv = *(const f32x4*)(a.gout + unit * 8 + 4 * kh);
#pragma unroll
    for (int zo = 0; zo < ZB; ++zo) {
        const int z = z0 + zo;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int yy = y0 + rowp, xx = x0 + 16 * nt + c;
            const bool ook = yy < a.Hv && xx < a.Wv;
            const long o = ook ? ((long)z * a.Ho + yy) * a.Wo + xx : 0;
            f32x4 v, dv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yp = ym[zo][nt][e] + yc[zo][nt][e] * H3_INV;
                v[e] = yp + bv[e];
                dv[e] = dm[zo][nt][e] + dc[zo][nt][e] * H3_INV + be[e] * yp;
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
            if (uok && ook) {
                const long ob = ((long)(a.out_g0 + 2 * unit) * a.out_pstride + o) * 16 + 8 * kh;
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

// 0: launched; 1: this launch has no such form (the caller takes conv_h3g_kernel<NARROW>)
static int launch_h3n4(ConvKArgs ka, int ctiles, hipStream_t s) {
    constexpr size_t smem = (size_t)HN4_LDS_UNITS * 16;
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    if (ctiles != 1 || ka.cout_groups > 2 || (ka.flags & F_RES) || !ka.beta || ka.Dv % HN4_ZB != 0) return 1;
    if (ka.nchunk + ka.nskip > NBE_MAX_GROUPS) return 1;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_h3n4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    ka.tny = (ka.Hv + HP_ROWS - 1) / HP_ROWS;
    ka.tnx = (ka.Wv + HP_COLS - 1) / HP_COLS;
    ka.ntiles = (ka.Dv / HN4_ZB) * ka.tny * ka.tnx;
    for (int chunk = 0; chunk < ka.nchunk; ++chunk) {
        const bool second = chunk >= ka.csplit;
        const long ps = second ? ka.in2_pstride : ka.in_pstride;
        const long off = (long)(second ? chunk - ka.csplit : chunk) * 4 * ps * 16;
        ka.gs[chunk] = {(const char*)(second ? ka.x2 : ka.x) + off, (const char*)(second ? ka.dx2 : ka.dx) + off,
                        (const char*)ka.w + (long)chunk * HN4_WC * 16, ps * 16};
    }
    for (int sc = 0; sc < ka.nskip; ++sc) {
        const bool second = sc >= ka.s_csplit;
        const long ps = second ? ka.s2_pstride : ka.s_pstride;
        const long off = (long)(second ? sc - ka.s_csplit : sc) * 4 * ps * 16;
        ka.gs[ka.nchunk + sc] = {(const char*)(second ? ka.xs2 : ka.xs) + off, (const char*)(second ? ka.dxs2 : ka.dxs) + off,
                                 (const char*)ka.ws + (long)sc * HN4_TAPU * 16, ps * 16};
    }
    ka.dws_delta = ka.nskip ? (const char*)ka.dws - (const char*)ka.ws : 0;
    hipLaunchKernelGGL(conv_h3n4_kernel, dim3(ka.ntiles), dim3(512), smem, s, ka);
    return 0;
}
