// Shared between nbe_kernels.hip (float32 path, data movement) and nbe_kernels_h3.hip (f16x3 path).
#pragma once
#include "nbe_kernels.h"

namespace nbe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define NBE_LDS_AS __attribute__((address_space(3)))
#define NBE_GLB_AS __attribute__((address_space(1)))

// conv_h3g_kernel: what group g of a launch reads, prepared by the launcher (everything that does not depend on the tile)
struct ConvGroupSrc { const char* x; const char* dx; const char* w; long psb; };

// kernel arguments of every convolution variant (POD, passed by value)
struct ConvKArgs {
    const float* x; const float* dx; long in_pstride;
    int D, H, W; long P; long in_off;
    int Dv, Hv, Wv; long Q;
    float* y; float* dy; long out_pstride; int out_g0;
    int Ho, Wo; int osz, oz, oy, ox;
    const float* r; const float* dr; long res_pstride;
    const float* bias; const float* w; const float* dw;
    int nchunk; int cout_groups; int flags; int ntiles;
    int tny, tnx;            // patch kernel: number of 8-row / 32-column tiles per output plane
    // tangent gauge (f16x3 style path, see conv_h3g_kernel): per-cout vectors, either may be NULL
    const float* gout;       // the stored tangent is dy + gout[o] * y
    const float* beta;       // the input tangent is in this layer's gauge: dy = W.dx~ + beta[o] * (W.x), no dW
    // conv_h3g_kernel<false> only.  Second K-segment of the input: chunks >= csplit are read from x2 / dx2 (same D, H, W
    // and offsets as x; its own plane stride) -- concat([skip, up]) without a concat tensor.  csplit >= nchunk: unused.
    const float* x2; const float* dx2; long in2_pstride; int csplit;
    // The block's 1x1x1 skip fused into its last 3x3x3 convolution: nskip 16-channel chunks of the block input xs / dxs
    // (chunks >= s_csplit from xs2 / dxs2), row and plane pitch (H, W) of x, pointers already offset so that the centre
    // tap of the patch of output tile (z, y0, x0) is the skip's voxel; ws / dws = the skip layer's FLAT1-packed W_s / dW_s~.
    const float* xs; const float* dxs; long s_pstride; const float* xs2; const float* dxs2; long s2_pstride; int s_csplit;
    const float* ws; const float* dws; int nskip;
    long dws_delta;          // dws - ws in bytes
    int up8; long set_stride; // up_h3_kernel: all eight parity sets of an up-sampling layer; bytes between weight sets
    const float* wws; long wws_set_floats;   // conv_h3w_kernel<SKIP>: the fused skip's [W_s | dW_s~] in that kernel's scaling, floats per set
    const float* ww;         // conv_h3w_kernel: the layer's Winograd-z packed weights when this launch may use them, or NULL
    const float* stem_w;     // stem_h3_kernel: the first layer's weights in its own packing (PackedW::stem), or NULL
    ConvGroupSrc gs[NBE_MAX_GROUPS];
};

__device__ __forceinline__ void dma16(const float* src, f32x4* dst_wave_base) {
    // 64 lanes x 16 B: LDS destination = wave-uniform base + lane*16 (hardware rule), source per lane.
    __builtin_amdgcn_global_load_lds((const NBE_GLB_AS void*)src, (NBE_LDS_AS void*)dst_wave_base, 16, 0, 0);
}

// XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run
// of tiles so that neighbouring tiles (which share input rows) hit the same L2.  Bijective form.
__device__ __forceinline__ int xcd_tile(int b, int nt) {
    const int qd = nt >> 3, rm = nt & 7, xcd = b & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (b >> 3);
}

// f16x3 path (nbe_kernels_h3.hip)
int launch_conv_h3(const PackedW& pw, const ConvKArgs& ka, bool vel, bool has_dx, hipStream_t s);
void launch_pack_h3(const float* w_oidhw, int cout, int cin, int kind, const PackedW& pw, float* dst, hipStream_t s);
void launch_gather_h8(const float* box, int C, int Db, int Hb, int Wb, int a0, int a1, int a2,
                      float* dst, const Planes& geom, float scale, int parts, hipStream_t s);
void launch_from_planes_h8(const float* src, const Planes& geom, int C, float* dst, int parts, hipStream_t s);
void launch_head_h8(const Planes& y, const Planes& xin, int c0, int C, const HeadScale& hs, bool vel,
                    void* disp, void* velo, int out_dtype, int Db, int Hb, int Wb, int a0, int a1, int a2,
                    int parts, hipStream_t s, int pad = 0);

}  // namespace nbe
