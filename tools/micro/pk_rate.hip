// Micro-benchmark: issue rate of v_pk_fma_f32 vs v_fma_f32 on gfx950 (same FLOPs, half the instructions).
// Result on MI355X (round 1): v_fma_f32 113.7 TFLOP/s, v_pk_fma_f32 128.7 TFLOP/s -> plain fp32 VALU already runs at the
// "packed" rate; packing buys ~13 % (fewer issue slots), not 2x.  Build: hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool PK>
__global__ __launch_bounds__(256) void rate(float* out, float s, int iters) {
    f32x2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x2{(float)threadIdx.x + i, (float)i};
    const f32x2 m = {s, s * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (PK) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
            } else {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][0]) : "v"(m[0]), "v"(a[(i + 1) & 7][0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][1]) : "v"(m[1]), "v"(a[(i + 1) & 7][1]));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * 8;
    for (int pk = 0; pk < 2; ++pk) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(rate<true>, dim3(grid), dim3(256), 0, 0, d, 1.0001f, iters);
            else hipLaunchKernelGGL(rate<false>, dim3(grid), dim3(256), 0, 0, d, 1.0001f, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 256 * iters * 16 * 2;
            if (rep) printf("%s: %.3f ms  %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, flop / ms * 1e-9);
        }
    }
    return 0;
}
