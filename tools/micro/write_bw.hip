// HBM write bandwidth by store pattern (gfx950): what the write-bound layers (first layer, up-sampling) can hope for.
//   A  one stream, 16 B per lane, contiguous (the textbook streaming store)
//   B  one stream, 8 B per lane, half-waves interleaved inside 16-byte units (the epilogue's store shape), contiguous
//   C  as B, but every workgroup writes 4 KB runs into 32 planes (hi/lo x y/dy x 8 channel groups), the layout of the
//      activation tensors: plane stride = voxels * 16 B
//   hipcc --offload-arch=gfx950 -O3 tools/micro/write_bw.hip -o tools/micro/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void kA(f4* dst, long units) {
    const long i = (long)blockIdx.x * 512 * 8 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const long j = i + k * 512; if (j < units) dst[j] = f4{1.f, 2.f, 3.f, 4.f}; }
}
__global__ __launch_bounds__(512) void kB(f2* dst, long units) {
    // lane (li, lh): unit = base + li, half lh -> 8 bytes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const long base = (long)blockIdx.x * 4096 + wave * 512;     // units; each wave 16 instructions x 32 units
#pragma unroll
    for (int k = 0; k < 16; ++k) { const long u = base + k * 32 + li; if (u < units) dst[2 * u + lh] = f2{1.f, 2.f}; }
}
__global__ __launch_bounds__(512) void kC(f2* dst, long vox, int planes) {
    // workgroup b owns voxels [256 b, 256 b + 256); wave w writes planes 4 w .. 4 w + 3 (x 8 waves = 32), 256 voxels each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const long v0 = (long)blockIdx.x * 256;
    for (int p = 0; p < 4; ++p) {
        const long pl = wave * 4 + p;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const long v = v0 + k * 32 + li; if (v < vox) dst[2 * (pl * vox + v) + lh] = f2{1.f, 2.f}; }
    }
}

int main() {
    const long bytes = 16l << 30;                          // 16 GiB
    void* d; hipMalloc(&d, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long units = bytes / 16;
    for (int pat = 0; pat < 3; ++pat) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (pat == 0) hipLaunchKernelGGL(kA, dim3((unsigned)(units / 4096)), dim3(512), 0, 0, (f4*)d, units);
            else if (pat == 1) hipLaunchKernelGGL(kB, dim3((unsigned)(units / 4096)), dim3(512), 0, 0, (f2*)d, units);
            else { const long vox = units / 32; hipLaunchKernelGGL(kC, dim3((unsigned)(vox / 256)), dim3(512), 0, 0, (f2*)d, vox, 32); }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("pattern %c: %.2f ms for %.1f GB -> %.2f TB/s\n", 'A' + pat, best, bytes / 1e9, bytes / 1e9 / best);
    }
    return 0;
}
