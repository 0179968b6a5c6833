// Issue rate of the f16 MFMA shapes on gfx950: cycles per instruction, one wave per SIMD, independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    half8 a8, b8; half4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (threadIdx.x + i)); b8[i] = (_Float16)(0.02f * (threadIdx.x - i)); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (SHAPE == 32) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[t], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int shape : {32, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
        m /= 256;
        printf("v_mfma_f32_16x16x%d_f16: %.2f cycles per instruction (one wave per SIMD, 8 independent accumulators)\n",
               shape, m / (iters * 8.0));
    }
    return 0;
}
