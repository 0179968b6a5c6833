// What the f16 matrix pipe sustains on gfx950 when nothing else is in the way: every SIMD issues v_mfma_f32_16x16x32_f16
// back to back for about a second (long enough for the power management to settle), with 0..8 ds_read_b128 per 8 MFMAs
// (conv_h3g_kernel: 20 per 48).  Wall time by HIP events -> PFLOP/s; the clock
// follows from the issue rate (16 cycles per instruction and SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float rnd(unsigned& s) {                  // uniform in (-1, 1): every mantissa bit toggles
    s = s * 1664525u + 1013904223u;
    return ((int)(s >> 8) - (1 << 23)) * (1.f / (1 << 23));
}

// RANDOM = 0: slowly varying small values (few bits toggle between consecutive operands); 1: random operands -- what a
// network's activations and weights look like to the multipliers.  The accumulators stay finite: a and b are O(1), zero mean.
template <int LDSR, int RANDOM>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ half8 sm[2048];
    unsigned seed = threadIdx.x * 9781u + blockIdx.x * 6271u + 1u;
    for (int i = threadIdx.x; i < 2048; i += 512)
        for (int e = 0; e < 8; ++e) sm[i][e] = RANDOM ? (_Float16)rnd(seed) : (_Float16)(0.001f * ((i + e) & 63));
    __syncthreads();
    half8 a8, b8;
    for (int i = 0; i < 8; ++i) {
        a8[i] = RANDOM ? (_Float16)rnd(seed) : (_Float16)(0.01f * ((threadIdx.x + i) & 31));
        b8[i] = RANDOM ? (_Float16)rnd(seed) : (_Float16)(0.02f * ((threadIdx.x - i) & 31));
    }
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
    int o = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t < LDSR) { b8 = sm[(o + 64 * t) & 2047]; }      // LDSR reads per 8 MFMAs
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[t], 0, 0, 0);
        }
        if (LDSR) o = (o + 512) & 2047;
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int LDSR, int RANDOM>
static void run(const char* name, float* out, int wgs_per_cu) {
    const int iters = 1500000 / wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<LDSR, RANDOM>), dim3(256 * wgs_per_cu), dim3(512), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)iters * 8 * 8 * 256 * wgs_per_cu;            // 8 waves per workgroup
        const double pf = inst * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e15;
        const double clk = inst / (256.0 * 4) * 16 / (ms * 1e-3) / 1e9;            // if the pipe never idles
        printf("%s, %d workgroup(s) of 8 waves per CU: %.1f ms  %.3f PFLOP/s f16  (= %.3f GHz if the pipe never idles)\n",
               name, wgs_per_cu, ms, pf, clk);
        fflush(stdout);
    }
}

int main() {
    float* out; hipMalloc(&out, 512 * 512 * 4 * 4);
    run<0, 0>("smooth operands, MFMA only", out, 1);
    run<0, 1>("random operands, MFMA only", out, 1);
    run<0, 1>("random operands, MFMA only", out, 2);
    run<2, 1>("random operands, 2 ds_read_b128 per 8 MFMAs", out, 1);
    run<3, 1>("random operands, 3 ds_read_b128 per 8 MFMAs (conv_h3g_kernel: 3.3)", out, 1);
    run<4, 1>("random operands, 4 ds_read_b128 per 8 MFMAs", out, 1);
    run<4, 1>("random operands, 4 ds_read_b128 per 8 MFMAs", out, 2);
    run<8, 1>("random operands, 8 ds_read_b128 per 8 MFMAs", out, 1);
    run<8, 1>("random operands, 8 ds_read_b128 per 8 MFMAs", out, 2);
    run<4, 0>("smooth operands, 4 ds_read_b128 per 8 MFMAs", out, 1);
    run<8, 0>("smooth operands, 8 ds_read_b128 per 8 MFMAs", out, 2);
    return 0;
}
