// Does v_mfma_f32_16x16x32_f16 take f16 SUBNORMAL A/B inputs at their value (no flush)?  conv_h3w_kernel's transformed lo part
// is unscaled since round 3 (V lo = err + (a lo +- b lo) 2^-11) and falls below 2^-14 for small activations: a flush there would
// cost 2^-14 absolute, gradual underflow costs 2^-25.    hipcc --offload-arch=gfx950 -O2 -o mfma_denorm mfma_denorm.hip && ./mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const _Float16* a, const _Float16* b, float* out) {
    const int lane = threadIdx.x;
    half8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = a[lane * 8 + j]; B[j] = b[lane * 8 + j]; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = acc[e];
}

int main() {
    _Float16 ha[512], hb[512];
    // A[m][k] = subnormal 3 * 2^-24 for every m, k;  B[k][n] = 1  ->  every output = 32 * 3 * 2^-24
    const _Float16 sub = (_Float16)(3.0f * 5.9604644775390625e-08f);
    for (int i = 0; i < 512; ++i) { ha[i] = sub; hb[i] = (_Float16)1.0f; }
    _Float16 *da, *db; float* dout; float ho[256];
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, sizeof ho);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    const float want = 32.0f * 3.0f * 5.9604644775390625e-08f;
    printf("subnormal A: got %.9g want %.9g -> %s\n", ho[0], want, ho[0] == want ? "kept (gradual underflow)" : "FLUSHED");
    // subnormal B as well
    for (int i = 0; i < 512; ++i) { ha[i] = (_Float16)1.0f; hb[i] = sub; }
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    printf("subnormal B: got %.9g want %.9g -> %s\n", ho[0], want, ho[0] == want ? "kept (gradual underflow)" : "FLUSHED");
    return 0;
}
