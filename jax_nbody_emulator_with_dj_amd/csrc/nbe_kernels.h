// Host-visible launch interface of the gfx950 kernels (nbe_kernels.hip).
// Internal to the library: the public C-ABI is include/nbe.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbe {

// Activation storage ("C/4-blocked"): a tensor of C channels over V = D*H*W
// voxels is C/4 planes; plane g holds float4 {c=4g..4g+3} per voxel, voxels in
// row-major (z,y,x) order.  `pstride` is the distance between planes in voxels.
struct Planes {
    float* x = nullptr;      // primal
    float* dx = nullptr;     // tangent (d/dDz); nullptr in displacement-only mode
    int64_t pstride = 0;     // voxels between consecutive planes
    int G = 0;               // number of planes (= padded channels / 4)
    int D = 0, H = 0, W = 0; // geometry
    int64_t vox() const { return (int64_t)D * H * W; }
};

// Arithmetic of the convolutions:
//   PREC_F32    strict float32: v_mfma_f32_32x32x2_f32 on float4 planes (C/4 planes of 4 x f32 per voxel)
//   PREC_F16X3  float32-equivalent split: every operand x is stored as two f16 numbers
//               hi = f16(x), lo = f16((x - hi) * 2^11); a product a*b is evaluated as
//               a_hi*b_hi + 2^-11 * (a_hi*b_lo + a_lo*b_hi) on v_mfma_f32_32x32x16_f16 with float32
//               accumulation (22 significant bits per operand; the dropped lo*lo term is 2^-22 relative).
//               Planes hold 8 x f16 per voxel: plane 2g = hi, 2g+1 = lo of channels 8g..8g+7 -- the same
//               bytes and plane count as PREC_F32, so workspace planning is identical.
//   PREC_F16    plain float16 operands, one f16 MFMA per product, float32 accumulation: what the reference
//               computes with SubboxConfig.dtype = float16.  Planes hold 8 x f16 per voxel, plane g = channels
//               8g..8g+7 (half the bytes and planes of the other two).
enum Precision { PREC_F32 = 0, PREC_F16X3 = 1, PREC_F16 = 2 };
inline constexpr bool prec_is_half(int prec) { return prec == PREC_F16X3 || prec == PREC_F16; }
inline constexpr int prec_parts(int prec) { return prec == PREC_F16X3 ? 2 : 1; }
constexpr float H3_SCALE = 2048.0f, H3_INV = 1.0f / 2048.0f;
// f16x3: the range shift of a call places max(|x| Dz / 6, max |b|) in [2^5, 2^6) instead of [0.5, 1): conv_h3w_kernel keeps the
// lo part of its transformed planes unscaled (nbe_kernels_wino.h, NBE_WINO_LOU), which is a normal f16 number for every
// |value| >= 2^-3 there -- 2^-9 of the input's scale -- and leaves 2^10 of headroom above the input's scale.
constexpr int H3_RANGE_UP = 6;

// conv_h3g_kernel reads its per-group sources from a table in its arguments: 3 * Cin / 16 groups of the layer
// + Cin_skip / 16 of a fused skip must fit
constexpr int NBE_MAX_GROUPS = 64;

enum ConvMode { MODE_FLAT3 = 0, MODE_FLAT1 = 1, MODE_DOWN = 2 };
enum ConvFlags { F_ACT = 1, F_RES = 2, F_SKIP_NODX = 4 };   // F_SKIP_NODX: the fused skip's input has no tangent (conv_l00)

// Tile constants shared by the weight packer and the conv kernel.
constexpr int TILE_VOX = 256;                 // voxels per workgroup tile
inline constexpr int mode_taps(int mode) { return mode == MODE_FLAT3 ? 3 : 1; }
inline constexpr int mode_nseg(int mode) { return mode == MODE_FLAT3 ? 9 : (mode == MODE_DOWN ? 8 : 1); }
inline constexpr int mode_ck(int mode) { return mode == MODE_FLAT3 ? 8 : 16; }
inline constexpr int prec_ck(int prec, int mode) { return prec_is_half(prec) ? 16 : mode_ck(mode); }

// One packed weight set (see pack_index in nbe_kernels.hip for the layout).
struct PackedW {
    float* w = nullptr;      // [ct][stage = chunk*nseg + seg][tap][CK/4][COUT_T][4]
    float* dw = nullptr;
    float* bias = nullptr;   // padded to ct*COUT_T
    int prec = PREC_F32;
    int mode = 0, ni = 2;    // COUT_T = 32*ni (PREC_F16X3: always 64)
    int cout_t = 64;         // f16-based packing: couts per tile; 16 = the narrow tile of conv_h3g_kernel<true>
    int cin = 0, cout = 0;   // logical channels
    int cin_pad = 0;         // multiple of CK
    int ctiles = 0;          // cout tiles
    int64_t floats = 0;      // size of w (and dw) in floats, per parity set
    int nsets = 1;           // 8 for the up-sample layer (one set per output parity)
    float* ww = nullptr;     // f16x3 + velocity, gauged 3x3x3 layers on the wide tile: Winograd-z packing of w (conv_h3w_kernel), or NULL
    float* stem = nullptr;   // f16x3 + velocity, conv_l00/conv_0 only: [W hi | W lo | dW hi | dW lo] x 96 k x 64 couts (stem_h3_kernel)
};

struct ConvLaunch {
    Planes in;               // input planes (G*4 >= cin_pad)
    int64_t in_off = 0;      // FLAT modes: flat input offset of output position q (crop)
    int Dv = 0, Hv = 0, Wv = 0;   // FLAT: valid output extents along z,y,x in q coordinates; DOWN: output dims
    Planes out;              // output planes; written planes are out_g0 + cout group
    int out_g0 = 0;          // first output PLANE (concat offset)
    int osz = 1, oz = 0, oy = 0, ox = 0;   // output voxel = (z*osz+oz, y*osz+oy, x*osz+ox) in out geometry
    Planes res;              // residual (same geometry as out), used when flags & F_RES
    int flags = 0;
    int set = 0;             // weight set (parity) index; -1: all eight sets of an up-sampling layer in one launch (f16x3, velocity)
    const float* gout = nullptr;   // tangent gauge of the output (per cout), f16x3 kernels only
    const float* beta = nullptr;   // gauged input: two-product tangent with this per-cout factor (3x3x3 f16x3 only)
    const float* bias = nullptr;   // replaces the layer's own bias (a block's conv_1 with its skip fused: b_1 + b_s)
    // gauged f16x3 3x3x3 kernel (wide tile) only:
    Planes in2;                    // channels >= csplit_ch of the input come from here (same geometry and offsets as `in`)
    int csplit_ch = 0;             // 0: single input tensor
    Planes sk, sk2;                // fused 1x1x1 skip: its input (the block input), sk2 = its channels >= sk_split_ch
    int sk_split_ch = 0;
    int64_t sk_off = 0;            // flat offset in sk of the voxel aligned with output (0, 0, 0)
    const PackedW* skw = nullptr;  // the skip layer's packed weights (FLAT1): w = W_s, dw = dW_s~
    bool wino = false;             // run the Winograd-z kernel when the layer and the launch have that form (run_conv)
};

// 0 on success; 1 = the layer / flag combination has no kernel (an engine bug, reported through nbe_last_error)
int launch_conv(const PackedW& pw, const ConvLaunch& L, bool vel, bool has_dx, hipStream_t s);

// weight preparation -------------------------------------------------------
// (w_n, dw_tot) in OIDHW from raw style parameters (style_layers_vel.py:62-105)
void launch_modulate(const float* weight, const float* style_weight, const float* style_bias,
                     int cout, int cin, int k3, float s0, float s1, float eps, int first_layer,
                     float* w_n, float* dw_tot /*nullable*/, hipStream_t s,
                     const float* a_in = nullptr, float* beta_out = nullptr, const float* b_sub = nullptr);
// alpha[ci] = (ds/dDz) / s of the style modulation of a layer; *flag |= 1 where s is (numerically) zero or |alpha| > 64
void launch_style_alpha(const float* style_weight, const float* style_bias, int cin, float s0, float s1,
                        float* alpha, int* flag, hipStream_t s);
// OIDHW -> packed layout; `kind`: 0 conv3, 1 skip(1x1x1), 2 down(k2 s2), 3 up(k2, 8 parity sets)
void launch_pack(const float* w_oidhw, int cout, int cin, int kind, const PackedW& pw, float* dst, hipStream_t s);

// Winograd-z packing of a 3x3x3 layer's modulated weights for conv_h3w_kernel (nbe_kernels_wino.h): dst holds 4/3 of
// PackedW::floats; *flag |= 1 when a weight leaves the f16 range at the kernel's 2^14 scale
void launch_pack_h3w(const float* w_oidhw, int cout, int cin, int cin_pad, int ctiles, float* dst, int* flag, hipStream_t s,
                     int prec = PREC_F16X3);   // PREC_F16: the float16 model's form (32-channel stages, one part)
// a fused skip's weights (FLAT1 packing of pw) in that kernel's scaling: dst holds pw.floats floats
void launch_pack_h3w_skip(const float* w_oidhw, int cout, int cin, const PackedW& pw, float* dst, int* flag, hipStream_t s);

// data movement --------------------------------------------------------------
// periodic crop of a (C, Db, Hb, Wb) float box into input planes, scaled by `scale`
void launch_gather(const float* box, int C, int Db, int Hb, int Wb, int a0, int a1, int a2,
                   const Planes& dst, float scale, int prec, hipStream_t s);
// NCDHW (C,D,H,W) dense -> planes (+scale), and back (debug / apply path)
void launch_to_planes(const float* src, int C, const Planes& dst, bool tangent, float scale, int prec, hipStream_t s);
void launch_from_planes(const Planes& src, bool tangent, int C, float* dst, int prec, hipStream_t s);
// centre crop by c voxels per side into planes [g0, g0+src.G) of dst
// (cz >= 0: crop along z by cz instead of c -- the z-slab schedule copies plane ranges)
void launch_crop(const Planes& src, int c, const Planes& dst, int g0, bool vel, hipStream_t s, int cz = -1);
// head: disp = (y + x0)*6 ; vel = dy*(vf*6) + x0*(vf*6/Dz); x0 = input planes cropped by `c0`;
// written to a (C, Db, Hb, Wb) box at origin (a0,a1,a2); out_dtype 0 = f32, 1 = f16.
// The three factors arrive ready-made (k_disp = 6/s, k_dy = vf*6/s, k_x0 = vf*6/(Dz*s), s = the power-of-two range
// shift of the call); *bad |= 1 when a non-finite value is written (nullable).
struct HeadScale { float k_disp = 6.f, k_dy = 0.f, k_x0 = 0.f; int* bad = nullptr; };
void launch_head(const Planes& y, const Planes& xin, int c0, int C, const HeadScale& hs, bool vel,
                 void* disp, void* velo, int out_dtype, int Db, int Hb, int Wb, int a0, int a1, int a2,
                 int prec, hipStream_t s, int pad = 0);
// *out_bits = max(*out_bits, bit pattern of |src[i]|) over n floats (non-negative floats order like their bit
// patterns; a NaN or an infinity gives >= 0x7f800000)
void launch_absmax(const float* src, int64_t n, unsigned* out_bits, hipStream_t s);
// zero `bytes` (a multiple of 16) at a 16-byte aligned address
void launch_zero(void* dst, int64_t bytes, hipStream_t s);
// dst[i] = (src[i] + (src2 ? src2[i] : 0)) * f
void launch_scale(const float* src, float* dst, int n, float f, hipStream_t s, const float* src2 = nullptr);

// periodic y/x halo of width `pad` of a tensor whose interior has been written; dst = src extended periodically in y/x
void launch_fill_yx(const Planes& t, int pad, bool vel, hipStream_t s);
void launch_wrap_pad(const Planes& src, const Planes& dst, int pad, bool vel, hipStream_t s, int padz = 0);

// *dst = v (dst nullable); *flag |= bit when *expect_at != v (expect_at nullable)
void launch_tag_word(unsigned* dst, unsigned v, const unsigned* expect_at, unsigned* flag, unsigned bit, hipStream_t s);

// Branch probe (test instrumentation): sign bits of one stored activation tensor over a probe region.
// The launch's output is `ext` voxels starting at x (row pitch W, plane pitch H * W, plane stride pstride, first plane g0);
// its voxel (0,0,0) has index `org` in the frame of the oracle's tensor for this layer, periodic in y / x with `per` where > 0.
// The probe wants the n^3 voxels from index `o` on, all C channels, as (C, n, n, nw) 32-bit words.
struct ProbeLaunch {
    const float* x; int64_t pstride; int H, W, g0, prec, C;
    int ext[3], org[3], per[3];   // per[0] is not used: z wraps through (zlo, zhi, zper)
    int zlo, zhi, zper;           // zper > 0: the layer's tensor exists for planes [zlo, zhi) of a box periodic in z with that period
    int o[3], n, nw;
    unsigned* bits; unsigned* count;
};
void launch_probe_signs(const ProbeLaunch& a, hipStream_t s);

// NBE_DBG builds: per-phase cycle totals of the f16x3 3x3x3 kernel since the last call (zeros otherwise)
void h3q_read_stamps(double* out16, hipStream_t s);

}  // namespace nbe
