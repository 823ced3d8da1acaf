// Micro-benchmark: how many plain VALU ops issue "for free" beside a back-to-back stream of v_mfma_f32_16x16x4_f32 on one
// gfx950 SIMD?  Each loop iteration issues 8 MFMAs on 8 independent accumulators with V VALU fmas after each MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, bool MFMA>
__global__ __launch_bounds__(512) void body(float* out, float s, int iters) {
    f32x4 acc[8];
    float v[16];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i);
    const float a = s, b = s * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MFMA) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < V; ++k)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[(i * V + k) & 15]) : "v"(a), "v"(b));
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int V, bool MFMA>
float run(float* d, int threads) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((body<V, MFMA>), dim3(256), dim3(threads), 0, 0, d, 1.0001f, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e6f / (iters * 8);   // ns per MFMA slot
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    for (int threads = 256; threads <= 512; threads *= 2) {
        printf("waves per SIMD: %d   (ns per [1 MFMA + V VALU] group, per wave stream)\n", threads / 256);
        printf("  V   mfma+valu   valu-only   mfma-only\n");
        const float m = run<0, true>(d, threads);
#define ROW(V) printf("  %2d   %8.2f   %8.2f   %8.2f\n", V, run<V, true>(d, threads), run<V, false>(d, threads), m)
        ROW(1); ROW(2); ROW(4); ROW(6); ROW(8); ROW(12); ROW(16); ROW(24);
    }
    return 0;
}
