// Included by nbe_kernels_h3.hip (uses its HGGeom / dma16s / acc_read / patch constants).
//
// conv_h3w_kernel: the gauged 3x3x3 layer (conv_h3g_kernel, wide tile) with a Winograd F(2,3) transform ALONG Z.
//
// Two output planes z0, z0 + 1 at the same (y, x) come from four input planes d0..d3 = z0..z0 + 3 with four plane-wise
// 2-D convolutions (nine (dy, dx) taps each) instead of six:
//     M0 = U0 * (d0 - d2)   M1 = U1 * (d1 + d2)   M2 = U2 * (d2 - d1)   M3 = U3 * (d1 - d3)
//     U0 = w[dz=0]   U1 = (w0 + w1 + w2) / 2   U2 = (w0 - w1 + w2) / 2   U3 = w[dz=2]
//     y(z0) = M0 + M1 + M2          y(z0 + 1) = M1 - M2 - M3
// (Lavin & Gray's F(2,3), applied to the plane index only: the transform matrices have entries 0, +-1, +-1/2, and the
// one-dimensional form amplifies rounding by < 2: tests hold the kernel to the tolerances of the direct one.)
// That is 2/3 of the MFMAs of conv_h3g_kernel -- on a chip that runs this loop against its power limit, 2/3 of the time
// the MFMAs cost.  What makes it fit:
//
// * Accumulators.  M1 and M2 enter both planes, so a naive form needs four accumulator sets per plane pair where the
//   direct kernel has two.  Ordering K removes that: phase 1 runs xi = 1 into set A and xi = 2 into set B over all input
//   chunks, then (A, B) := (A + B, A - B) in registers, then phase 2 adds xi = 0 to A and xi = 3 (with U3 negated by the
//   packer) to B: A ends as plane z0, B as plane z0 + 1 -- two sets for two planes, as before.
// * One accumulator per output instead of a (main, correction) pair: the weights are scaled by 2^14 (style-modulated
//   weights are unit vectors per cout, so |U| <= 1) and their lo part is kept UNSCALED (|lo| <= 2^3, exact to 2^-24 of
//   the largest weight); the product hi(w) * lo(x), which the direct kernel sums in a separate accumulator because lo(x)
//   is stored times 2^11, meets an UNSCALED lo part here: the transformed planes are made by this kernel, and their lo part
//   is written as err + (a lo +- b lo) 2^-11 (NBE_WINO_LOU below; until then the 2^-11 sat on a copy of every weight
//   operand).  All three products of a float32 product then land in the same float32 accumulator; the epilogue
//   multiplies by 2^-14.
//   => 128 accumulator registers hold 64 couts x 32 positions x 2 planes x (y, dy): the wave tile, LDS image and operand
//   reads per MFMA of conv_h3g_kernel<false, TALL>.
// * The transformed planes V = a +- b are built by the workgroup itself: each wave loads its share of the two raw
//   patches into registers (16 B per lane and part), adds them part-wise in packed f16 (TwoSum for the hi parts' rounding
//   error: xf_step) and writes the result into the LDS patch buffer of the NEXT stage in the layout the B operands are read in.  Weights still arrive by
//   global -> LDS DMA.  LDS: 2 x 36 KB of weights + 2 x 44 KB of patches, as in conv_h3g_kernel.
//
// A stage is (phase, chunk, xi); stage s accumulates into set s & 1.  Per 16 input channels and 512 outputs: 4 stages of
// 28 MFMAs per tile where the direct kernel runs 6.

// NBE_WINO_LOU (default 1): the transformed patch carries its lo part UNSCALED (V lo = err + (a lo +- b lo) 2^-11), so the
// product hi(w) . lo(V) takes the weights as they are -- no 2^-11 copy of every weight operand (96 of ~340 non-matrix vector
// operations per stage and wave), no wp registers, no wait states behind those writes.  What makes that safe is the engine's
// range shift (H3_RANGE_UP, nbe_kernels.h): activations live at 2^6 x (input max in [0.5, 1)), where the lo part of anything
// that matters is a normal f16 number and the gradual underflow of the rest costs 2^-25 absolute, 2^-31 of the input's
// scale.  (f16 subnormals enter the MFMA at their value: tools/micro/mfma_denorm.hip.)  -DNBE_WINO_LOU=0 keeps round 3's
// first form (lo scaled by 2^11 like every stored lo part, weights scaled on the way) for same-device A/Bs.
// NBE_WINO_ZROW (default 1): the [lo(w) | 0] operand of the single tap reads its zero half from a zeroed kilobyte of LDS behind
// the patch buffers instead of being cleared by sixteen v_cndmask per stage (and their wait states before the MFMA);
// the weight DMA's lane offset lives in a register (one of those the unscaled lo part freed).
#ifndef NBE_EPI_MAX
#define NBE_EPI_MAX 1
#endif
#ifndef NBE_WINO_ZROW
#define NBE_WINO_ZROW 1
#endif
#ifndef NBE_WINO_ITEMREG
#define NBE_WINO_ITEMREG 1
#endif
#ifndef NBE_WINO_LOU
#define NBE_WINO_LOU (NBE_XF_F32 ? 0 : 1)
#endif
constexpr int NBE_MAX_WSTAGES = 32;                // 4 * Cin / 16: Cin <= 128
constexpr int NBE_MAX_WSKIP = 16;                  // fused skip: 2 planes x Cin_block / 16 raw stages after the transformed ones
constexpr float WINO_WSCALE = 16384.0f;            // 2^14
constexpr float WINO_WSCALE_F16 = 256.0f;          // the float16 form: no lo part to keep out of the subnormals, 2^8 for the small weights

struct WinoSrc { const char* xa; const char* xb; long dxd; const char* w; long psb; float sb; int pad_; };

struct WinoKArgs {
    int H, W, Dv, Hv, Wv, Ho, Wo;
    int nchunk, cout_groups, flags, ntiles, tny, tnx;
    int zblock;                // tile order: plane pairs fastest in blocks of this many (>= 1)
    float* y; float* dy; long out_pstride; int out_g0;
    const float* bias; const float* gout; const float* beta;
    const float* r; const float* dr; long res_pstride;          // F16 with F_RES: the residual (geometry of the output)
    float inv_scale;
    int nskip;                 // SKIP: 16-channel chunks of the block input (two raw stages each, plane z0 and plane z0 + 1)
    long dws_delta;            // SKIP: bytes from the scaled W_s to the scaled dW_s~ of a chunk
    WinoSrc st[NBE_MAX_WSTAGES + NBE_MAX_WSKIP];
};

// SKIP: the block's 1x1x1 skip runs inside this, its last, convolution as in conv_h3g_kernel: after the transformed stages, two
// RAW stages per 16-channel chunk of the block input (its plane under z0 into set A, the next one into set B), patches by
// global -> LDS DMA (nothing to transform), weights [W_s | dW_s~] scaled by 2^14 like the layer's own:
//     y += W_s.x      dy += W_s.dx~ + dW_s~.x          (dW_s~ = dW_s - W_s (.) a - beta_1 W_s, see conv_h3g_kernel)
// on the centre tap, the K halves selecting the part: [wh | wh 2^-11] . [xh | xl] and [wl | 0] . [xh | xl].
//
// NOVEL: the displacement-only layer (style_layers.py:86-99: y = conv(x, w) + b, no tangent) on the SAME instruction stream.
// Without a tangent the second accumulator set and the second patch tensor are free, so they carry a second block of
// eight output rows: "dx" is x eight rows further down (WinoSrc::dxd = 8 rows), "DY" the outputs of rows y0 + 8 .. y0 + 15.
// A workgroup then owns 16 rows x 32 columns x 2 planes for the same weights in LDS -- half the weight stream per output,
// which is what the one-accumulator-set form lacks (conv_h2q_kernel<SPLIT> doubles its tile for the same reason).  Only
// the epilogue (bias and LeakyReLU for both sets, both stored to y) and the fused skip's products (W_s.x for both row
// blocks, no dW_s~) differ.
//
// F16: the plain-float16 model (NBE_PREC_F16: one f16 plane per eight channels, one MFMA per product) on the same LDS image
// and stage skeleton.  What is the (hi, lo) pair of a 16-channel chunk above is here a pair of 16-channel chunks: a stage
// covers 32 input channels (plane u = 2 h + part of the stage <-> channels 8 u .. 8 u + 7), the transform is V = a +- b on
// each plane by itself (one packed operation per register), a tap pair runs four products (w0.x0 + w1.x1 into Y, w0.dx0~ +
// w1.dx1~ into DY) and the single tap one product per set with K = [part 0 | part 1] fully used: 144 MFMAs per stage and wave,
// none wasted.  Weights are scaled by 2^8 only (no lo part to protect).  The epilogue adds the residual (F_RES: the block's
// skip as computed by its own launch -- the float16 model has no fused skip), rounds once and stores one plane per unit.
template <bool SKIP, bool NOVEL, bool F16 = false>
__global__ __launch_bounds__(512, 2) void conv_h3w_kernel(WinoKArgs a) {
    static_assert(!(F16 && NOVEL), "the float16 form has no displacement-only variant");
    typedef HGGeom<false, true, false> G;
    constexpr int NW = 8, CT = 64, TAPU = G::TAPU, WGU = G::WG, XBASE = G::XBASE, MT = 4, NT = 2, NTILE = 8;
    constexpr int ZROW = G::LDS_UNITS;                           // 64 zeroed units behind the patch buffers (NBE_WINO_ZROW)
    f32x4* lds = lds_h3;
    const half8* L8 = (const half8*)lds_h3;
    half8* L8w = (half8*)lds_h3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4, kh = q & 1, ks = q >> 1;

    const int nct = (a.cout_groups + 7) / 8;
    // (persistent workgroups looping over tiles -- grids of 256, 512, 1024 -- were measured: +-0.3 % per box; what a tile
    // spends outside its stages, ~4.5 stage times, is its own epilogue, butterfly and first staging, not the dispatch)
    const int vt = xcd_tile(blockIdx.x, a.ntiles * nct);
    const int tile = vt / nct, ct = vt - tile * nct;
    const int npair = a.Dv >> 1;
    // tile order: plane pairs fastest in blocks of a.zblock (then x, y, the next block of pairs): the workgroups running side
    // by side on an XCD share input planes through its L2 along z and pages (TLB reach, of the stores above all) along x
    int zp, ty, tx;
    {
        const int zb = a.zblock;
        const int full = (npair / zb) * zb;                      // plane pairs in whole blocks
        const int per_blk = zb * a.tny * a.tnx;
        if (tile < (npair / zb) * per_blk) {
            const int blk = tile / per_blk, r = tile - blk * per_blk;
            const int zi = r % zb, tyx = r / zb;
            zp = blk * zb + zi; ty = tyx / a.tnx; tx = tyx - ty * a.tnx;
        } else {                                                 // the remaining pairs: z fastest
            const int r = tile - (npair / zb) * per_blk, rem = npair - full;
            const int zi = r % rem, tyx = r / rem;
            zp = full + zi; ty = tyx / a.tnx; tx = tyx - ty * a.tnx;
        }
    }
    const int y0 = ty * (NOVEL ? 2 * HP_ROWS : HP_ROWS), x0 = tx * HP_COLS, z0 = 2 * zp;
    const int nst = 4 * a.nchunk;

    const long to = (((long)z0 * a.H + y0) * a.W + x0) * 16;
    const long wcm = (long)ct * nst * WGU * 16, wcs = (long)ct * a.nskip * TAPU * 16;

    // ---- sources of the stage being prepared
    struct Nxt { const char *xa, *xb, *w0; long dxd, psb; float sb; } nx;
    auto set_next = [&](int s) {
        const WinoSrc e = a.st[s];
        nx.xa = e.xa + to; nx.xb = e.xb + to; nx.dxd = e.dxd; nx.psb = e.psb; nx.w0 = e.w + (SKIP && s >= nst ? wcs : wcm); nx.sb = e.sb;
    };
    unsigned lane16 = (unsigned)lane << 4;
    asm volatile("" : "+v"(lane16));
    auto dma_w = [&](int buf, int t) {                           // 36 wave-instructions of weights, 5 slots per wave
        const int n = wave + NW * t;
        unsigned l16 = (unsigned)lane;
        if (NBE_WINO_ZROW && !F16) l16 = lane16;
        else asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(l16));   // recomputed at every use: held in a register it is spilled
        if (n < G::NWI) dma16s(nx.w0 + (long)n * 1024, l16, lds + buf * WGU + n * 64);
    };
    // ---- a raw stage (SKIP): 8 wave-instructions of weights (W_s, dW_s~ of the chunk: one per wave) and 24 + 24 of the x and
    // dx~ patches of ONE plane (three + three per wave), straight into the buffers of the stage
    auto dma_raw = [&](int buf) {
        unsigned l16 = (unsigned)lane;
        asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(l16));
        dma16s(nx.w0 + (wave < 4 ? 0 : a.dws_delta) + (long)(wave & 3) * 1024, l16, lds + buf * WGU + wave * 64);
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int tensor = t / 3, n = wave + NW * (t % 3), pl = n / 6, k = n - 6 * pl;
            const int u = k * 64 + lane;
            const int uu = u < HP_PL ? u : HP_PL - 1;
            const int row = (uu * 241) >> 13, col = uu - row * HP_RS;
            if (u < HP_PL)
                dma16s(nx.xa + (tensor ? nx.dxd : 0) + (long)pl * nx.psb, (unsigned)(row * a.W + col) * 16u,
                       lds + XBASE + buf * HQ_XB + tensor * HQ_XT + pl * HQ_PP + k * 64);
        }
    };
    // ---- the transformed patch of the next stage: 24 wave-items (tensor, channel half, 64 units of the 340) of a hi and a
    // lo plane each, three per wave; an item is four 16-byte loads per lane (a hi, a lo, b hi, b lo), 8 channels of
    // V = a + sb * b in float32, and two 16-byte LDS stores.
    // (per-lane offsets are recomputed at every use -- a dozen VALU operations per item -- instead of held in registers)
    // ITEMREG (since the unscaled lo part freed the scaled-weight registers): the three items' unit and plane offset live in
    // six registers for the whole workgroup -- before, a dozen VALU operations per item and stage recomputed them.  (The
    // float16 form has no registers to spare: there they are recomputed.)
    constexpr bool ITEMREG = NBE_WINO_ITEMREG && !F16;
    int it_uu[3] = {0, 0, 0};
    unsigned it_go[3] = {0u, 0u, 0u};
    if (ITEMREG) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int u = ((wave + NW * j) % 6) * 64 + lane;
            it_uu[j] = u < HP_PL ? u : HP_PL - 1;
            const int row = (it_uu[j] * 241) >> 13, col = it_uu[j] - row * HP_RS;   // uu / 34 for uu < 384
            it_go[j] = (unsigned)(row * a.W + col) * 16u;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(it_uu[j]), "+v"(it_go[j]));   // held, not rematerialised
    }
    auto item_unit = [&](int j, bool& valid) {                   // unit 0..339 of the 10 x 34 patch plane this lane handles in item j
        if (ITEMREG) {
            valid = lane < HP_PL - ((wave + NW * j) % 6) * 64;
            return it_uu[j];
        }
        int l = lane;
        asm volatile("" : "+v"(l));                              // (or the compiler hoists the offsets out of the loop and spills them)
        const int u = ((wave + NW * j) % 6) * 64 + l;
        valid = u < HP_PL;
        return valid ? u : HP_PL - 1;
    };
    auto item_goff_j = [&](int j) {                              // byte offset of that unit in an input plane
        if (ITEMREG) return it_go[j];
        bool v;
        const int uu = item_unit(j, v);
        const int row = (uu * 241) >> 13, col = uu - row * HP_RS;   // uu / 34 for uu < 384
        return (unsigned)(row * a.W + col) * 16u;
    };
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // (the staging registers belong to one stage: declared there, so that nothing is carried across stage boundaries)
    struct Stg { half8 ah, al, bh, bl; float sb; };
    auto st_load = [&](int j, Stg& g) {
        const int n = wave + NW * j;
        const long po = ((n / 12) ? nx.dxd : 0) + (long)(((n % 12) / 6) * 2) * nx.psb;
        const char* pa = nx.xa + po;
        const char* pb = nx.xb + po;
        const unsigned go = item_goff_j(j);
        // buffer loads: wave-uniform base in a resource descriptor (SGPRs) + 32-bit lane offset (+ the lo plane's distance as
        // the scalar offset) -- the compiler's own global loads keep a 64-bit address per lane and load.  Unlike loads
        // issued from asm statements these are visible to the compiler's vmcnt bookkeeping (spill-safe, counted waits).
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)pa, 0, 0xffffffffu, 0x00027000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)pb, 0, 0xffffffffu, 0x00027000);
        const unsigned pso = (unsigned)nx.psb;                   // < 2^32 - patch extent: checked by the launcher
        auto ld = [&](half8& d, __amdgpu_buffer_rsrc_t r, unsigned so) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, go, so, 0);
            d = __builtin_bit_cast(half8, v);
        };
        ld(g.ah, ra, 0); ld(g.al, ra, pso); ld(g.bh, rb, 0); ld(g.bl, rb, pso);
        g.sb = nx.sb;
    };
    // V = (a hi + 2^-11 a lo) + sb (b hi + 2^-11 b lo), two channels per register, entirely in packed f16 (round 3; before:
    // joined to float32 with the mixed-precision FMA and split again, 11 operations per channel pair):
    //     s   = a hi + sb b hi                          the new hi part (one rounding)
    //     err = TwoSum's exact rounding error of s      (Knuth: bb = s - a hi; err = (a hi - (s - bb)) + (sb b hi - bb))
    //     lo  = err 2^11 + (a lo + sb b lo)             the new lo part, scaled by 2^11 like every lo part
    // 8 packed operations.  (s, lo) is not the canonical split of V -- s is the rounding of a hi + sb b hi, not of V -- but it
    // represents V to the same absolute accuracy, 2^-23 (|a| + |b|): what the float32 form delivers too, since a and b carry
    // 22 bits each; tools/wino_emulation.py reproduces both.  sb = +-1 is a packed constant of the stage (SGPR).
    // Sixteen micro-steps of two operations each: in the loop one follows every MFMA of two products, where it issues in the
    // MFMA's shadow -- all waves of a workgroup reach the same point of a stage together, so a block of VALU work stalls the
    // matrix pipe of its SIMD for its whole length (measured: 3.2 VALU per MFMA and 56 % matrix-busy with the transform in blocks).
#if NBE_XF_F32   // A/B build: the round-2 transform (joined to float32 with the mixed-precision FMA, 11 operations per channel pair)
    struct Xf { u32x4 HI, LO; float t0, t1; unsigned h; };
    auto xf_step = [&](const Stg& g, Xf& x, int k) {
        const int r = k >> 2, m = k & 3;
        const unsigned AH = __builtin_bit_cast(u32x4, g.ah)[r], AL = __builtin_bit_cast(u32x4, g.al)[r];
        const unsigned BH = __builtin_bit_cast(u32x4, g.bh)[r], BL = __builtin_bit_cast(u32x4, g.bl)[r];
        const float inv = H3_INV, sinv = g.sb * H3_INV, sb = g.sb, k2048 = H3_SCALE;
        if (m == 0) {
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(x.t0) : "v"(AL), "s"(inv), "v"(AH));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x.t0) : "v"(BL), "s"(sinv));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x.t0) : "v"(BH), "s"(sb));
        } else if (m == 1) {
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(x.t1) : "v"(AL), "s"(inv), "v"(AH));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x.t1) : "v"(BL), "s"(sinv));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x.t1) : "v"(BH), "s"(sb));
        } else if (m == 2) {
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x.h) : "v"(x.t0), "v"(x.t1));
            asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x.t0) : "v"(x.h));
            asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x.t1) : "v"(x.h));
        } else {
            unsigned l;
            asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(l) : "v"(x.t0), "s"(k2048));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(l) : "v"(x.t1), "s"(k2048));
            x.HI[r] = x.h; x.LO[r] = l;
        }
    };
#else
    struct Xf { u32x4 HI, LO; unsigned s, bb, e1, e2, l1; };
    auto xf_step = [&](const Stg& g, Xf& x, int k) {
        const int r = k >> 2, m = k & 3;
        const unsigned AH = __builtin_bit_cast(u32x4, g.ah)[r], AL = __builtin_bit_cast(u32x4, g.al)[r];
        const unsigned BH = __builtin_bit_cast(u32x4, g.bh)[r], BL = __builtin_bit_cast(u32x4, g.bl)[r];
        const unsigned sbp = g.sb < 0.f ? 0xBC00BC00u : 0x3C003C00u, k2048 = 0x68006800u;      // packed (sb, sb), (2048, 2048)
        if (F16) {                                               // two independent planes: V = a + sb b, one rounding each
            if (m == 0) asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(x.s) : "v"(BH), "s"(sbp), "v"(AH));
            else if (m == 1) asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(x.l1) : "v"(BL), "s"(sbp), "v"(AL));
            else if (m == 3) { x.HI[r] = x.s; x.LO[r] = x.l1; }
            (void)k2048;
        } else
        if (m == 0) {
            asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(x.s) : "v"(BH), "s"(sbp), "v"(AH));
            asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x.bb) : "v"(x.s), "v"(AH));
        } else if (m == 1) {
            asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x.e1) : "v"(x.s), "v"(x.bb));          // s - bb
            asm("v_pk_fma_f16 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(x.e2) : "v"(BH), "s"(sbp), "v"(x.bb));   // sb b hi - bb
        } else if (m == 2) {
            asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x.e1) : "v"(AH), "v"(x.e1));            // a hi - (s - bb)
            asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(x.l1) : "v"(BL), "s"(sbp), "v"(AL));
        } else {
            unsigned err, l;
            asm("v_pk_add_f16 %0, %1, %2" : "=v"(err) : "v"(x.e1), "v"(x.e2));
#if NBE_WINO_LOU
            const unsigned kinv = 0x10001000u;                   // packed (2^-11, 2^-11)
            asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(l) : "v"(x.l1), "s"(kinv), "v"(err));     // unscaled: err + (a lo + sb b lo) 2^-11
            (void)k2048;
#else
            asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(l) : "v"(err), "s"(k2048), "v"(x.l1));
#endif
            x.HI[r] = x.s; x.LO[r] = l;
        }
    };
#endif
    auto st_write = [&](int j, int buf, const Xf& x) {
        bool valid;
        const int n = wave + NW * j;
        const int lo_ = XBASE + buf * HQ_XB + (n / 12) * HQ_XT + (((n % 12) / 6) * 2) * HQ_PP + item_unit(j, valid);
        if (valid) {
            L8w[lo_] = __builtin_bit_cast(half8, x.HI);
            L8w[lo_ + HQ_PP] = __builtin_bit_cast(half8, x.LO);
        }
    };
    auto st_store = [&](int j, int buf, Stg& g) {                // the same in one block (prologue)
        Xf x;
#pragma unroll
        for (int k = 0; k < 16; ++k) xf_step(g, x, k);
        st_write(j, buf, x);
    };

    f32x4 YA[NTILE], DA[NTILE], YB[NTILE], DB[NTILE];            // AGPRs, updated in place (see conv_h3q_kernel)
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { YA[t][e] = 0.f; DA[t][e] = 0.f; YB[t][e] = 0.f; DB[t][e] = 0.f; }
    auto mm = [&](f32x4& acc, const half8& A, const half8& B) {
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
    };
    // behind a VALU write of A: the two wait states the ISA asks for travel with the MFMA (tools/check_mfma_hazards.py)
    auto mmz = [&](f32x4& acc, const half8& A, const half8& B) {
        asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
    };

    const int rowp = wave;                                       // this wave's row of the 8 x 32 patch
    const int aP = (ks * 4 + 2 * kh) * CT + c;
    const int bB = (2 * kh) * HQ_PP + rowp * HP_RS + c;
    const int bP1 = bB + ks, bP32 = bB + 32 * ks;
    auto LA = [&](half8 (&r)[MT], int idx) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) r[mt] = L8[idx + 16 * mt];
    };
    auto LB = [&](half8 (&r)[NT], int idx) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) r[nt] = L8[idx + 16 * nt];
    };
    const _Float16 kInv = (_Float16)H3_INV;
    // one product on the wave tile; slot >= 0: weight-DMA slots `slot`, `slot + 1` after the two halves of the product
    // The eight tiles of a product are walked in a snake -- (m0,n0) (m0,n1) (m1,n1) (m1,n0) (m2,n0) ... -- so that consecutive
    // MFMAs share an operand: eight operand changes per product instead of twelve (this kernel's time is its energy)
#define SNAKE(i) ((((i) >> 1) & 1) ? ((i) ^ 1) : (i))
    auto nohook = [](int) {};
    // (hook(t) runs after MFMA t: a micro-step of VALU work that issues while the matrix pipe is busy with that MFMA)
    auto MM8h = [&](f32x4 (&acc)[NTILE], const half8 (&A)[MT], const half8 (&B)[NT], int slot, int nb, bool px, bool zsel,
                    bool hooked, auto&& hook) {
#pragma unroll
        for (int i = 0; i < NTILE; ++i) {
            const int t = SNAKE(i);
            if (zsel && (i % NT) == 0) mmz(acc[t], A[t / NT], B[t % NT]); else
            mm(acc[t], A[t / NT], B[t % NT]);
            if (hooked) { hook(i); __builtin_amdgcn_sched_barrier(0); }
            if (slot >= 0 && (i % 4) == 3) {
                if (px) dma_w(nb, slot + i / 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto MM8 = [&](f32x4 (&acc)[NTILE], const half8 (&A)[MT], const half8 (&B)[NT], int slot, int nb, bool px, bool zsel = false) {
        MM8h(acc, A, B, slot, nb, px, zsel, false, nohook);
    };
    // the first product of a pair, hi(w) 2^-11 . lo(x): the scaled operands are made on the way (one row of tiles ahead of
    // its MFMAs) and kept in wp for the pair's fourth product
    auto MM8s = [&](f32x4 (&acc)[NTILE], half8 (&P)[MT], const half8 (&A)[MT], const half8 (&B)[NT], int slot, int nb, bool px) {
        P[0] = A[0] * kInv;
#pragma unroll
        for (int i = 0; i < NTILE; ++i) {
            const int t = SNAKE(i);
            if ((i % NT) == 0) { if (t / NT + 1 < MT) P[t / NT + 1] = A[t / NT + 1] * kInv; mmz(acc[t], P[t / NT], B[t % NT]); }
            else mm(acc[t], P[t / NT], B[t % NT]);
            if (slot >= 0 && (i % 4) == 3) {
                if (px) dma_w(nb, slot + i / 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#define NBE_SB __builtin_amdgcn_sched_barrier(0)
    half8 wh[MT], wl[MT], wp[MT], xh[NT], xl[NT], dxh[NT], dxl[NT];
    // A tap pair: six products into (Y, DY).  On entry wh, xl and xh of the pair are loaded (or in flight); preXl / preW /
    // preXh request those of whatever follows as soon as the registers are free.  Dependent MFMAs are >= 8 MFMAs apart.
    // (mid: after the third product every LDS read of the pair has been issued -- the stage's barrier goes there)
    auto pair = [&](f32x4 (&Y)[NTILE], f32x4 (&DY)[NTILE], int slot0, int nb, bool px, int wa, int xp,
                    auto&& preXl, auto&& preW, auto&& preXh, auto&& mid, bool hooked, auto&& hook, auto&& after3) {
        if (F16) {
            // four products: w0.x0 + w1.x1 -> Y, w0.dx0~ + w1.dx1~ -> DY (wh / wl: the weights of the stage's two 16-channel
            // halves, xh / xl and dxh / dxl their planes); every LDS read of the pair is issued before the second product
            LB(dxh, xp + HQ_XT); LA(wl, wa + CT + aP);
            NBE_SB; MM8h(Y, wh, xh, slot0, nb, px, false, hooked, [&](int t) { hook(t); }); NBE_SB;
            LB(dxl, xp + HQ_XT + HQ_PP);
            NBE_SB; MM8h(Y, wl, xl, slot0 < 0 ? -1 : slot0 + 2, nb, px, false, hooked, [&](int t) { hook(8 + t); }); NBE_SB;
            after3();
            mid();
            preXl(); preXh();
            NBE_SB; MM8(DY, wh, dxh, slot0 < 0 ? -1 : slot0 + 4, nb, px); NBE_SB;
            preW();
            NBE_SB; MM8(DY, wl, dxl, -1, nb, px); NBE_SB;
            return;
        }
        LB(dxh, xp + HQ_XT);
#if NBE_WINO_LOU
        NBE_SB; MM8(Y, wh, xl, slot0, nb, px); NBE_SB;                                    // hi(w) . lo(V), lo unscaled
#else
        NBE_SB; MM8s(Y, wp, wh, xl, slot0, nb, px); NBE_SB;                               // hi(w) 2^-11 . lo(x)
#endif
        LB(dxl, xp + HQ_XT + HQ_PP);
        NBE_SB; MM8h(Y, wh, xh, slot0 < 0 ? -1 : slot0 + 2, nb, px, false, hooked, [&](int t) { hook(t); }); NBE_SB;       // hi . hi
        LA(wl, wa + CT + aP);
        NBE_SB; MM8h(DY, wh, dxh, slot0 < 0 ? -1 : slot0 + 4, nb, px, false, hooked, [&](int t) { hook(8 + t); }); NBE_SB;
        after3();
        mid();
#if NBE_WINO_LOU
        NBE_SB; MM8(DY, wh, dxl, -1, nb, px); NBE_SB;
#else
        NBE_SB; MM8(DY, wp, dxl, -1, nb, px); NBE_SB;
#endif
        preXl(); preW();                                                                  // (not earlier: registers)
        NBE_SB; MM8(Y, wl, xh, -1, nb, px); NBE_SB;                                       // lo(w) . hi(x)
        preXh();
        NBE_SB; MM8(DY, wl, dxh, -1, nb, px); NBE_SB;
    };

    constexpr int SH4 = HP_RS + 1, SH5 = HP_RS + 2, SH7 = 2 * HP_RS + 1;   // tap shifts: 3*dy + dx -> dy*34 + dx
    // (pxc: a stage follows -- a compile-time constant, so that the fetch of the next stage costs no branches; only the
    // last stage of a workgroup runs the other instantiation)
    auto stage = [&](f32x4 (&Y)[NTILE], f32x4 (&DY)[NTILE], int s, auto nxt) {
        constexpr int NXT = decltype(nxt)::value;                // what follows: 0 nothing, 1 a transformed stage, 2 a raw stage (SKIP)
        constexpr bool px = NXT == 1;
        // the first operands of stage s+1 are requested under the last products of stage s -- except across the phase
        // boundary, where they would only be carried through the butterfly (registers): it requests them itself
        const bool pre = px && s + 1 != 2 * a.nchunk;
        if (NXT) set_next(s + 1);
        // (the buffer parity is a compile-time constant of each instantiation: hidden from the compiler, which would
        // otherwise precompute one address register per LDS read of the stage -- dozens, spilled)
        int par = s & 1;
        asm volatile("" : "+s"(par));
        int nb = 1 - par;
        asm volatile("" : "+s"(nb));
        const int wb = par * WGU, xb = XBASE + par * HQ_XB;
        const int wbn = nb * WGU, xbn = XBASE + nb * HQ_XB;
        // the transformed patch of stage s+1: item j is loaded, transformed under the second and third product of a later
        // pair (one micro-step per MFMA) and written after that pair's third product; the next item's loads follow at once
        Stg g;
        Xf xf;
        auto hk = [&](int k) { xf_step(g, xf, k); };
        auto none = [] {};
        if (NXT == 2) dma_raw(nb);                               // (its first operands are requested by the raw stage itself)
        if (px) st_load(0, g);
        pair(Y, DY, 0, nb, px, wb, xb + bP1,                                              // taps (0,1) + the weight DMA of stage s+1
             [&] { LB(xl, xb + 2 + bP32 + HQ_PP); }, [&] { LA(wh, wb + 2 * TAPU + aP); }, [&] { LB(xh, xb + 2 + bP32); },
             none, false, nohook, none);
        half8 a1[MT], a2[MT], b1x[NT], b1d[NT];
        // single tap 4 = (dy 1, dx 1): the K halves select the PART: [wh | wh 2^-11] . [xh | xl] and [wl | 0] . [xh | xl]
        const int aS = wb + 4 * TAPU + (2 * kh + (F16 ? ks : 0)) * CT + c;
        const int bS = xb + (2 * kh + ks) * HQ_PP + rowp * HP_RS + c + SH4;
        pair(Y, DY, -1, nb, px, wb + 2 * TAPU, xb + 2 + bP32,                             // taps (2,3)
             [&] { LB(b1x, bS); }, [&] { LA(a1, aS); }, [&] { LB(b1d, bS + HQ_XT); }, none,
             px, hk, [&] { if (px) { st_write(0, nb, xf); st_load(1, g); } });
        const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        if (F16) {                                               // [w0 | w1] . [x0 | x1]: one product per set, K fully used
            NBE_SB; MM8(Y, a1, b1x, -1, 0, false); NBE_SB;
            LB(xl, xb + SH5 + bP32 + HQ_PP); LB(xh, xb + SH5 + bP32); LA(wh, wb + 5 * TAPU + aP);
            NBE_SB; MM8(DY, a1, b1d, -1, 0, false); NBE_SB;
        } else {
#if NBE_WINO_LOU
        NBE_SB; MM8(Y, a1, b1x, -1, 0, false); NBE_SB;                                    // [wh | wh] . [V hi | V lo]
#else
        {
            const _Float16 m1 = ks ? kInv : (_Float16)1.0f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a1[mt] = a1[mt] * m1;
        }
        NBE_SB; MM8(Y, a1, b1x, -1, 0, false, true); NBE_SB;
#endif
#if NBE_WINO_ZROW
        LA(a2, ks ? ZROW + c : aS + CT);                                                  // [wl | 0]: the K half of the lo plane reads zeros
#else
        LA(a2, aS + CT);                                                                  // (only now: four A operand sets at once do not fit)
#endif
        LB(xl, xb + SH5 + bP32 + HQ_PP);
        NBE_SB; MM8(DY, a1, b1d, -1, 0, false); NBE_SB;
        LB(xh, xb + SH5 + bP32); LA(wh, wb + 5 * TAPU + aP);
#if NBE_WINO_ZROW
        NBE_SB; MM8(Y, a2, b1x, -1, 0, false); NBE_SB;
#else
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a2[mt] = ks ? zero : a2[mt];
        // (transforming item 1 under the single tap's last products instead, 7 / 5 / 7 products between loads and first use
        // rather than 6 / 8 / 4, was measured: +-0: the loads' latency is not exposed)
        NBE_SB; MM8(Y, a2, b1x, -1, 0, false, true); NBE_SB;
#endif
        MM8(DY, a2, b1d, -1, 0, false); NBE_SB;
        }
        pair(Y, DY, -1, nb, px, wb + 5 * TAPU, xb + SH5 + bP32,                           // taps (5,6)
             [&] { LB(xl, xb + SH7 + bP1 + HQ_PP); }, [&] { LA(wh, wb + 7 * TAPU + aP); }, [&] { LB(xh, xb + SH7 + bP1); },
             none, px, hk, [&] { if (px) { st_write(1, nb, xf); st_load(2, g); } });
        // taps (7,8); the stage's one barrier sits after the third product: by then this wave has read everything it
        // needs from the buffers of stage s, and the patch and weights of stage s+1 are complete once every wave has
        // waited for its own stores and DMA
        pair(Y, DY, -1, nb, px, wb + 7 * TAPU, xb + SH7 + bP1,
             [&] { if (pre) LB(xl, xbn + bP1 + HQ_PP); }, [&] { if (pre) LA(wh, wbn + aP); }, [&] { if (pre) LB(xh, xbn + bP1); },
             [&] { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); },
             px, hk, [&] { if (px) st_write(2, nb, xf); });
    };

    // ---- a raw stage of the fused skip: six products on the centre tap of the block input's patch
    auto raw_stage = [&](f32x4 (&Y)[NTILE], f32x4 (&DY)[NTILE], int s, bool px) {
        if (px) set_next(s + 1);
        int par = s & 1;
        asm volatile("" : "+s"(par));
        int nb = 1 - par;
        asm volatile("" : "+s"(nb));
        const int wb = par * WGU, xb = XBASE + par * HQ_XB;
        if (px) dma_raw(nb);
        half8 a1[MT], a2[MT], d1[MT], d2[MT], bx[NT], bd[NT];
        const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        const _Float16 m1 = ks ? kInv : (_Float16)1.0f;
        const int aS = wb + (2 * kh) * CT + c;
        const int bS = xb + (2 * kh + ks) * HQ_PP + rowp * HP_RS + c + SH4;
        if (F16) {
            // a chunk of the float16 model's skip is 32 channels, the K halves their two 16-channel halves: three products,
            // [W_s0 | W_s1] . [x0 | x1] -> Y,  the same . [dx~0 | dx~1] -> DY,  [dW_s~0 | dW_s~1] . [x0 | x1] -> DY
            const int aF = wb + (2 * kh + ks) * CT + c;
            LA(a1, aF); LB(bx, bS); LB(bd, bS + HQ_XT); LA(d1, aF + TAPU);
            NBE_SB; MM8(Y, a1, bx, -1, 0, false); NBE_SB;
            MM8(DY, a1, bd, -1, 0, false); NBE_SB;
            MM8(DY, d1, bx, -1, 0, false); NBE_SB;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            return;
        }
        LA(a1, aS); LB(bx, bS); LA(a2, aS + CT); LB(bd, bS + HQ_XT);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a1[mt] = a1[mt] * m1;
        NBE_SB; MM8(Y, a1, bx, -1, 0, false, true); NBE_SB;                               // [W_s hi | W_s hi 2^-11] . [x hi | x lo]
        if (!NOVEL) LA(d1, aS + TAPU);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a2[mt] = ks ? zero : a2[mt];
        NBE_SB; MM8(Y, a2, bx, -1, 0, false, true); NBE_SB;                               // [W_s lo | 0] . [x hi | x lo]
        if (!NOVEL) LA(d2, aS + TAPU + CT);
        if (NOVEL || !(a.flags & F_SKIP_NODX)) {                                         // (conv_l00: the input field has no tangent)
            NBE_SB; MM8(DY, a1, bd, -1, 0, false); NBE_SB;                                // NOVEL: W_s . x of the second row block
            MM8(DY, a2, bd, -1, 0, false); NBE_SB;
        }
        if (!NOVEL) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { d1[mt] = d1[mt] * m1; d2[mt] = ks ? zero : d2[mt]; }
            NBE_SB; MM8(DY, d1, bx, -1, 0, false, true); NBE_SB;                          // dW_s~ . x
            MM8(DY, d2, bx, -1, 0, false, true); NBE_SB;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    // ---- prologue: stage 0
    {
        if (NBE_WINO_ZROW && !F16 && tid < 64) L8w[ZROW + tid] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        set_next(0);
#pragma unroll
        for (int k = 0; k < G::NWS; ++k) dma_w(0, k);
        {
            // all three items in flight at once (nothing else is live yet): one memory latency exposed instead of three
            Stg g0, g1, g2;
            st_load(0, g0); st_load(1, g1); st_load(2, g2);
            st_store(0, 0, g0); st_store(1, 0, g1); st_store(2, 0, g2);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        LA(wh, aP); LB(xl, XBASE + bP1 + HQ_PP); LB(xh, XBASE + bP1);
    }

    auto butterfly = [&] {
            // end of phase 1: A = M1, B = M2  ->  A = M1 + M2, B = M1 - M2
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // MFMA results -> accumulator reads
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
                // one tile at a time, and back in its accumulator registers before the next one is touched
                f32x4 p, m;
#pragma unroll
                for (int e = 0; e < 4; ++e) { p[e] = acc_read(YA[t][e]); m[e] = acc_read(YB[t][e]); }
                YA[t] = p + m; YB[t] = p - m;
                asm volatile("" : "+a"(YA[t]), "+a"(YB[t]));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) { p[e] = acc_read(DA[t][e]); m[e] = acc_read(DB[t][e]); }
                DA[t] = p + m; DB[t] = p - m;
                asm volatile("" : "+a"(DA[t]), "+a"(DB[t]));
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_nop 7" ::: "memory");                 // accumulator writes -> the next MFMAs' C operands
            LA(wh, aP); LB(xl, XBASE + bP1 + HQ_PP); LB(xh, XBASE + bP1);   // stage 2 * nchunk reads buffers 0
    };
    // the last pair of stages runs outside the loop: its second stage fetches nothing (the other instantiation), and a
    // branch between the two instantiations inside the loop would merge two versions of every accumulator
    const int npairs = 2 * a.nchunk;
    for (int s2 = 0; s2 + 1 < npairs; ++s2) {
        if (s2 == a.nchunk) butterfly();
        stage(YA, DA, 2 * s2, std::integral_constant<int, 1>());
        stage(YB, DB, 2 * s2 + 1, std::integral_constant<int, 1>());
    }
    if (npairs - 1 == a.nchunk) butterfly();
    stage(YA, DA, 2 * npairs - 2, std::integral_constant<int, 1>());
    stage(YB, DB, 2 * npairs - 1, std::integral_constant<int, SKIP ? 2 : 0>());
    if (SKIP) {
        for (int sc = 0; sc < a.nskip; ++sc) {
            raw_stage(YA, DA, nst + 2 * sc, true);
            raw_stage(YB, DB, nst + 2 * sc + 1, sc + 1 < a.nskip);
        }
    }
#undef NBE_SB
#undef SNAKE
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // MFMA results -> VALU reads of the epilogue

    // ---- epilogue: y = W.x / 2^14 + b, dy = W.dx~ / 2^14 + beta * (W.x), LeakyReLU (+ tangent), gauge, split, store
    // The per-channel vectors of the four row tiles are loaded ONCE, before anything is stored, and serve both planes: vmcnt
    // counts loads and stores alike, so a load among the stores waits for every store issued before it -- with the vectors
    // fetched per row tile the epilogue was eight rounds of (store, drain, load, wait); worth 2 % on the 64-channel layers.
    // (Builds without the stores, or without the epilogue, run a box 13-16 % faster -- but they leave every activation at the
    // workspace's zeros, and on this power-limited chip MFMAs on zeros are cheap: such probes price the data, not the stores.)
    const bool act = a.flags & F_ACT, gauge = !NOVEL && a.gout != nullptr;
    f32x4 bv[MT], be[MT], gv[MT];
    int unit[MT];
    bool uok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        unit[mt] = ct * (CT / 8) + 2 * mt + ks;
        uok[mt] = unit[mt] < a.cout_groups;
        if (!uok[mt]) unit[mt] = a.cout_groups - 1;
        bv[mt] = *(const f32x4*)(a.bias + unit[mt] * 8 + 4 * kh);
        be[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!NOVEL) be[mt] = *(const f32x4*)(a.beta + unit[mt] * 8 + 4 * kh);
        gv[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!NOVEL && gauge) gv[mt] = *(const f32x4*)(a.gout + unit[mt] * 8 + 4 * kh);
    }
    // (a use of all of them here: the compiler waits for the loads now, once -- left to their first uses it waits with a count
    // that only the draining of the stores issued in between can satisfy)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(bv[mt]), "+v"(be[mt]), "+v"(gv[mt]));
    // (the 64-bit part of every store address -- unit, plane pitch, K half -- once per workgroup; per tile only + 16 o)
    long ob_mt[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ob_mt[mt] = (long)(a.out_g0 + (F16 ? 1 : 2) * unit[mt]) * a.out_pstride * 16 + 8 * kh;
    const long ol_off = a.out_pstride * 16;
    auto epilogue = [&](f32x4 (&Y)[NTILE], f32x4 (&DY)[NTILE], int z) {
        int o[NT];
        bool ook[NT], ook2[NT];                                  // NOVEL: the second row block, eight rows further down
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int yy = y0 + rowp, xx = x0 + 16 * nt + c;
            ook[nt] = yy < a.Hv && xx < a.Wv;
            ook2[nt] = NOVEL && yy + HP_ROWS < a.Hv && xx < a.Wv;
            o[nt] = ook[nt] ? (z * a.Ho + yy) * a.Wo + xx : z * a.Ho * a.Wo;
        }
        // F16 with a residual: the plane's sixteen residual vectors are loaded before anything of the plane is stored (vmcnt
        // counts loads and stores alike, see above)
        half4 r0[F16 ? NTILE : 1], r1[F16 ? NTILE : 1];
        const bool res = F16 && (a.flags & F_RES);
        if (F16 && res) {
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
                const int mt = t / NT, nt = t % NT;
                const long rb = ((long)unit[mt] * a.res_pstride + (long)o[nt]) * 16 + 8 * kh;
                r0[t] = *(const half4*)((const char*)a.r + rb);
                r1[t] = *(const half4*)((const char*)a.dr + rb);
            }
#pragma unroll
            for (int t = 0; t < NTILE; ++t) asm volatile("" : "+v"(r0[t]), "+v"(r1[t]));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int t = mt * NT + nt;
                f32x4 v, dv;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float yp = acc_read(Y[t][e]) * a.inv_scale;
                    v[e] = yp + bv[mt][e];
                    dv[e] = acc_read(DY[t][e]) * a.inv_scale + be[mt][e] * yp;
                }
                if (F16 && res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += (float)r0[t][e]; dv[e] += (float)r1[t][e]; }
                }
                if (act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dv[e] = v[e] > 0.f ? dv[e] : 0.01f * dv[e];
                        // max(v, 0.01 v) = (v >= 0 ? v : 0.01 v) without the compare -> vcc -> select chain and its wait states
                        // (the float16 form keeps the select: there the compiler pairs the compares)
                        v[e] = (F16 || !NBE_EPI_MAX) ? (v[e] >= 0.f ? v[e] : 0.01f * v[e]) : __builtin_fmaxf(v[e], 0.01f * v[e]);
                    }
                }
                if (gauge) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[e] += gv[mt][e] * v[e];
                }
                if (F16) {                                       // one f16 plane per unit; the residual was added before the activation
                    if (uok[mt] && ook[nt]) {
                        const long ob = ob_mt[mt] + (long)o[nt] * 16;
                        half4 h, dh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { h[e] = (_Float16)v[e]; dh[e] = (_Float16)dv[e]; }
                        *(half4*)((char*)a.y + ob) = h;
                        *(half4*)((char*)a.dy + ob) = dh;
                    }
                } else
                if (NOVEL) {                                     // two row blocks of y: bias and LeakyReLU on both sets
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dv[e] = acc_read(DY[t][e]) * a.inv_scale + bv[mt][e];
                        if (act) dv[e] = dv[e] >= 0.f ? dv[e] : 0.01f * dv[e];
                    }
                    const long ob = ob_mt[mt] + (long)o[nt] * 16;
                    const long ol = ob + ol_off;
                    const long r8 = (long)HP_ROWS * a.Wo * 16;
                    half4 hi, lo;
                    if (uok[mt] && ook[nt]) {
                        split4(v, hi, lo);
                        *(half4*)((char*)a.y + ob) = hi;
                        *(half4*)((char*)a.y + ol) = lo;
                    }
                    if (uok[mt] && ook2[nt]) {
                        split4(dv, hi, lo);
                        *(half4*)((char*)a.y + ob + r8) = hi;
                        *(half4*)((char*)a.y + ol + r8) = lo;
                    }
                } else
                if (uok[mt] && ook[nt]) {
                    const long ob = ob_mt[mt] + (long)o[nt] * 16;
                    const long ol = ob + ol_off;
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
    };
    epilogue(YA, DA, z0);
    epilogue(YB, DB, z0 + 1);
}

// Winograd packing of a 3x3x3 layer's weights (OIDHW float32) for conv_h3w_kernel:
// [ct][stage = chunk*4 + xi][tap = 3*dy + dx][u = 2*h + part][co 64][j 8], channel = chunk*16 + 8*h + j;
// value = U_xi[oc, ci, dy, dx] * 2^14 (U3 negated), part 0 = its f16 rounding, part 1 = the remainder, unscaled.
// *flag |= 1 when a scaled weight leaves the f16 range (the layer then stays on the direct kernel).
// f16 (the float16 model, conv_h3w_kernel<., ., true>): a stage is 32 channels, channel = chunk*32 + 8*u + j, every u holds
// the f16 rounding of U_xi * 2^8.
__global__ __launch_bounds__(256) void pack_h3w_kernel(const float* __restrict__ w, int cout, int cin, int nchunk, long total,
                                                       _Float16* __restrict__ dst, int* __restrict__ flag, int f16) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    long r = idx;
    const int j = (int)(r % 8); r /= 8;
    const int co = (int)(r % 64); r /= 64;
    const int u = (int)(r % 4); r /= 4;
    const int tap = (int)(r % 9); r /= 9;
    const int stage = (int)(r % (4 * nchunk)); r /= 4 * nchunk;
    const int ct = (int)r;
    const int chunk = stage >> 2, xi = stage & 3;
    const int h = u >> 1, part = u & 1;
    const int ci = f16 ? chunk * 32 + 8 * u + j : chunk * 16 + 8 * h + j, oc = ct * 64 + co;
    float v = 0.f;
    if (oc < cout && ci < cin) {
        const float* p = w + (((size_t)oc * cin + ci) * 3) * 9 + tap;        // [dz][dy][dx]
        const float w0 = p[0], w1 = p[9], w2 = p[18];
        v = xi == 0 ? w0 : xi == 1 ? 0.5f * (w0 + w1 + w2) : xi == 2 ? 0.5f * (w0 - w1 + w2) : -w2;
    }
    v *= f16 ? WINO_WSCALE_F16 : WINO_WSCALE;
    if (!(fabsf(v) <= 60000.f)) { atomicOr(flag, 1); v = 0.f; }
    const _Float16 hi = (_Float16)v;
    dst[idx] = (part == 0 || f16) ? hi : (_Float16)(v - (float)hi);
}

void launch_pack_h3w(const float* w_oidhw, int cout, int cin, int cin_pad, int ctiles, float* dst, int* flag, hipStream_t s, int prec) {
    const bool f16 = prec == PREC_F16;
    const int nchunk = cin_pad / (f16 ? 32 : 16);
    const long total = (long)ctiles * 4 * nchunk * 9 * 4 * 64 * 8;
    hipLaunchKernelGGL(pack_h3w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_oidhw, cout, cin, nchunk, total,
                       (_Float16*)dst, flag, f16 ? 1 : 0);
}

// 0: launched; 1: this launch has no Winograd form (the caller falls back to conv_h3g_kernel)
// wws: the fused skip's weights [W_s | dW_s~] in the kernel's scaling (PackedW::ww of the skip layer), needed when ka.nskip > 0
// novel: the displacement-only form (conv_h3w_kernel<SKIP, true>): no tangent tensors, two row blocks per workgroup
// f16: the float16 model's form (32-channel stages, no fused skip, residual in the epilogue)
static int launch_h3w(const ConvKArgs& ka_in, const float* ww, const float* wws, long wws_set_floats, int ctiles, hipStream_t s,
                      bool novel = false, bool f16 = false) {
    typedef HGGeom<false, true, false> G;
    constexpr size_t smem = (size_t)(G::LDS_UNITS + 64) * 16;   // + the zeroed kilobyte (NBE_WINO_ZROW)
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    ConvKArgs ka = ka_in;
    if (f16) {                                                   // 16-channel chunks -> 32-channel stages
        if (novel || (ka.nchunk & 1) || (ka.csplit < ka.nchunk && (ka.csplit & 1))) return 1;
        if (ka.nskip > 0 && ((ka.nskip & 1) || (ka.s_csplit < ka.nskip && (ka.s_csplit & 1)) || (ka.flags & F_SKIP_NODX))) return 1;
        ka.nchunk /= 2; if (ka.csplit < (1 << 29)) ka.csplit /= 2;
        ka.nskip /= 2; if (ka.s_csplit < (1 << 29)) ka.s_csplit /= 2;
    }
    if (!ww || (!f16 && (ka.flags & F_RES)) || (ka.Dv & 1) || 4 * ka.nchunk > NBE_MAX_WSTAGES || (!novel && !ka.beta)) return 1;
    if (ka.nskip > 0 && (!wws || 2 * ka.nskip > NBE_MAX_WSKIP)) return 1;
    if (ctiles != (ka.cout_groups + 7) / 8) return 1;
    // the raw planes are fetched with buffer loads (32-bit offsets): lane offset + hi -> lo plane distance stay below 2^32
    if (std::max(ka.in_pstride, ka.csplit < ka.nchunk ? ka.in2_pstride : 0L) * 16 + 16L * (HP_ROWS + 2) * ka.W >= (1L << 32) - (1L << 20)) return 1;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_h3w_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv_h3w_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv_h3w_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv_h3w_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv_h3w_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv_h3w_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    WinoKArgs wa;
    wa.H = ka.H; wa.W = ka.W; wa.Dv = ka.Dv; wa.Hv = ka.Hv; wa.Wv = ka.Wv; wa.Ho = ka.Ho; wa.Wo = ka.Wo;
    wa.nchunk = ka.nchunk; wa.cout_groups = ka.cout_groups; wa.flags = ka.flags;
    const int trows = novel ? 2 * HP_ROWS : HP_ROWS;
    wa.tny = (ka.Hv + trows - 1) / trows;
    wa.tnx = (ka.Wv + HP_COLS - 1) / HP_COLS;
    wa.ntiles = (ka.Dv / 2) * wa.tny * wa.tnx;
    wa.y = ka.y; wa.dy = ka.dy; wa.out_pstride = ka.out_pstride; wa.out_g0 = ka.out_g0;
    wa.bias = ka.bias; wa.gout = novel ? nullptr : ka.gout; wa.beta = novel ? nullptr : ka.beta;
    const long rows8 = (long)HP_ROWS * ka.W * 16;                // novel: the "tangent" patch is the input eight rows further down
    wa.inv_scale = 1.0f / (f16 ? WINO_WSCALE_F16 : WINO_WSCALE);
    wa.r = ka.r; wa.dr = ka.dr; wa.res_pstride = ka.res_pstride;
    {
        // A/B (profiles/r02_ab_wino_zblock.txt): 0 = all pairs of the launch (z fastest) 1494 ms per box, 1 (x fastest) 1496,
        // 2 / 4 / 8 / 16: 1479 / 1481 / 1477 / 1482
        static const int zb = getenv("NBE_WINO_ZBLOCK") ? atoi(getenv("NBE_WINO_ZBLOCK")) : 8;
        wa.zblock = zb > 0 ? std::min(zb, ka.Dv / 2) : ka.Dv / 2;
    }
    wa.nskip = ka.nskip; wa.dws_delta = wws_set_floats * 4;
    // stage order: phase 0 = (xi 1 -> A, xi 2 -> B) per chunk, phase 1 = (xi 0 -> A, xi 3 -> B) per chunk
    static const int XI[2][2] = {{1, 2}, {0, 3}};
    static const int PA[4] = {0, 1, 2, 1}, PB[4] = {2, 2, 1, 3};
    static const float SB[4] = {-1.f, 1.f, -1.f, -1.f};
    const long plane = (long)ka.H * ka.W * 16;
    for (int ph = 0; ph < 2; ++ph)
        for (int chunk = 0; chunk < ka.nchunk; ++chunk)
            for (int ab = 0; ab < 2; ++ab) {
                const int xi = XI[ph][ab];
                const bool second = chunk >= ka.csplit;
                const long ps = second ? ka.in2_pstride : ka.in_pstride;
                const char* x = (const char*)(second ? ka.x2 : ka.x);
                const char* dx = (const char*)(second ? ka.dx2 : ka.dx);
                const long off = (long)(second ? chunk - ka.csplit : chunk) * 4 * ps * 16;
                WinoSrc& e = wa.st[(ph * ka.nchunk + chunk) * 2 + ab];
                e.xa = x + off + PA[xi] * plane; e.xb = x + off + PB[xi] * plane; e.dxd = novel ? rows8 : dx - x;
                e.w = (const char*)ww + (long)(chunk * 4 + xi) * G::WG * 16; e.psb = ps * 16; e.sb = SB[xi]; e.pad_ = 0;
            }
    // raw stages of the fused skip: chunk sc of the block input, plane z0 (-> A) and plane z0 + 1 (-> B); ka.xs is already
    // offset so that the centre tap of the patch of output tile (z, y0, x0) is the skip's voxel
    for (int sc = 0; sc < ka.nskip; ++sc)
        for (int ab = 0; ab < 2; ++ab) {
            const bool second = sc >= ka.s_csplit;
            const long ps = second ? ka.s2_pstride : ka.s_pstride;
            const char* x = (const char*)(second ? ka.xs2 : ka.xs);
            const char* dx = (const char*)(second ? ka.dxs2 : ka.dxs);
            const long off = (long)(second ? sc - ka.s_csplit : sc) * 4 * ps * 16 + ab * plane;
            WinoSrc& e = wa.st[4 * ka.nchunk + 2 * sc + ab];
            e.xa = x + off; e.xb = e.xa; e.dxd = novel ? rows8 : dx - x; e.w = (const char*)wws + (long)sc * G::TAPU * 16; e.psb = ps * 16; e.sb = 0.f; e.pad_ = 0;
        }
    dim3 grid(wa.ntiles * ctiles, 1, 1), block(512, 1, 1);
    if (f16 && ka.nskip > 0) hipLaunchKernelGGL((conv_h3w_kernel<true, false, true>), grid, block, smem, s, wa);
    else if (f16) hipLaunchKernelGGL((conv_h3w_kernel<false, false, true>), grid, block, smem, s, wa);
    else if (novel) {
        wa.dws_delta = 0;
        if (ka.nskip > 0) hipLaunchKernelGGL((conv_h3w_kernel<true, true>), grid, block, smem, s, wa);
        else hipLaunchKernelGGL((conv_h3w_kernel<false, true>), grid, block, smem, s, wa);
    } else if (ka.nskip > 0) hipLaunchKernelGGL((conv_h3w_kernel<true, false>), grid, block, smem, s, wa);
    else hipLaunchKernelGGL((conv_h3w_kernel<false, false>), grid, block, smem, s, wa);
    return 0;
}
