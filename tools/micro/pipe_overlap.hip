// Micro-benchmark (round 6): what do the two waves of ONE gfx950 SIMD share?  An 8-wave workgroup per CU (waves w and w + 4 sit on the
// same SIMD); waves 0..3 run role A, waves 4..7 role B, each its own instruction stream, timed per wave with s_memtime.
//   roles: Mb<N> = v_mfma_f32_16x16x32_bf16 on N independent accumulators (back-to-back), Mf<N> = v_mfma_f32_16x16x4_f32 likewise,
//          V = v_fma_f32 on 16 independent registers, P = v_pk_fma_f32 likewise, X = ds_read-free transcendental (v_exp_f32), - = idle
// Printed: cycles per instruction of each role alone and beside the other.
// Build: hipcc --offload-arch=gfx950 -O3 pipe_overlap.hip -o pipe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { IDLE = 0, MB2, MB8, MF2, MF8, VF, VP, VX, MIX_B, MIX_F, MB1, MF1, MIX_B1, MIX_B2, MIX_B6, MIX_B1D };
static const char* kNames[] = {"idle", "bf16 MFMA x2acc", "bf16 MFMA x8acc", "f32 MFMA x2acc", "f32 MFMA x8acc", "v_fma_f32", "v_pk_fma_f32", "v_exp_f32",
                               "1 bf16 MFMA + 3 v_fma (one stream)", "1 f32 MFMA + 3 v_fma (one stream)",
                               "bf16 MFMA x1acc (dependent)", "f32 MFMA x1acc (dependent)", "1 bf16 MFMA + 1 v_fma", "1 bf16 MFMA + 2 v_fma", "1 bf16 MFMA + 5 v_fma", "1 dependent bf16 MFMA + 3 v_fma"};
constexpr int kPerIter = 48;      // instructions per loop iteration of every role

template <int ROLE>
__device__ __forceinline__ void body(int iters, float s, float& sink) {
    f32x4 acc[8];
    float v[16];
    f32x2 pv[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (int i = 0; i < 8; ++i) pv[i] = f32x2{v[i], v[i + 8]};
    const float a = s, b = s * 0.5f;
    bf16x8 wa, wb;
    for (int i = 0; i < 8; ++i) { wa[i] = (__bf16)(s + i); wb[i] = (__bf16)(s - i); }
    const f32x2 pa = f32x2{a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kPerIter; ++i) {
            if constexpr (ROLE == MB2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 1]) : "v"(wa), "v"(wb));
            if constexpr (ROLE == MB8) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(wa), "v"(wb));
            if constexpr (ROLE == MF2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 1]) : "v"(a), "v"(b));
            if constexpr (ROLE == MF8) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a), "v"(b));
            if constexpr (ROLE == VF) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 15]) : "v"(a), "v"(b));
            if constexpr (ROLE == VP) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pv[i & 7]) : "v"(pa), "v"(pa));
            if constexpr (ROLE == VX) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 15]));
            if constexpr (ROLE == MIX_B) {
                if ((i & 3) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(i >> 2) & 1]) : "v"(wa), "v"(wb));
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 15]) : "v"(a), "v"(b));
            }
            if constexpr (ROLE == MB1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(wa), "v"(wb));
            if constexpr (ROLE == MF1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
            if constexpr (ROLE == MIX_B1) {
                if ((i & 1) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(i >> 1) & 1]) : "v"(wa), "v"(wb));
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 15]) : "v"(a), "v"(b));
            }
            if constexpr (ROLE == MIX_B2) {
                if ((i % 3) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(i / 3) & 1]) : "v"(wa), "v"(wb));
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 15]) : "v"(a), "v"(b));
            }
            if constexpr (ROLE == MIX_B6) {
                if ((i % 6) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(i / 6) & 1]) : "v"(wa), "v"(wb));
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 15]) : "v"(a), "v"(b));
            }
            if constexpr (ROLE == MIX_B1D) {
                if ((i & 3) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(wa), "v"(wb));
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 15]) : "v"(a), "v"(b));
            }
            if constexpr (ROLE == MIX_F) {
                if ((i & 3) == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[(i >> 2) & 1]) : "v"(a), "v"(b));
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 15]) : "v"(a), "v"(b));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3] + pv[i][0] + pv[i][1];
    for (int i = 0; i < 16; ++i) r += v[i];
    sink = r;
}

template <int RA, int RB, int PA = 0, int PB = 0>
__global__ __launch_bounds__(512) void pair_kernel(float* out, unsigned long long* cyc, float s, int iters) {
    const int wave = threadIdx.x >> 6;
    float sink = 0.f;
    if (wave < 4) __builtin_amdgcn_s_setprio(PA); else __builtin_amdgcn_s_setprio(PB);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) { if constexpr (RA != IDLE) body<RA>(iters, s, sink); }
    else { if constexpr (RB != IDLE) body<RB>(iters, s, sink); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if ((threadIdx.x & 63) == 0) atomicAdd(&cyc[wave < 4 ? 0 : 1], t1 - t0);
}

template <int RA, int RB, int PA = 0, int PB = 0>
void run(float* d, unsigned long long* c) {
    const int iters = 2000;
    (void)hipMemset(c, 0, 16);
    hipLaunchKernelGGL((pair_kernel<RA, RB, PA, PB>), dim3(256), dim3(512), 0, 0, d, c, 1.0001f, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemset(c, 0, 16);
    hipLaunchKernelGGL((pair_kernel<RA, RB, PA, PB>), dim3(256), dim3(512), 0, 0, d, c, 1.0001f, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double n = 256.0 * 4 * iters * kPerIter;
    printf("  A = %-36s (prio %d) B = %-36s (prio %d)  cycles / instruction:  A %6.2f   B %6.2f\n", kNames[RA], PA, kNames[RB], PB, h[0] / n, h[1] / n);
}

int main() {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    (void)hipMalloc(&c, 16);
    printf("two waves per SIMD (A = waves 0..3, B = waves 4..7 of an 8-wave workgroup, one per CU); s_memtime cycles per instruction of each stream\n");
    run<MB2, IDLE>(d, c); run<MB8, IDLE>(d, c); run<MF2, IDLE>(d, c); run<MF8, IDLE>(d, c);
    run<VF, IDLE>(d, c); run<VP, IDLE>(d, c); run<VX, IDLE>(d, c); run<MIX_B, IDLE>(d, c); run<MIX_F, IDLE>(d, c);
    run<MB2, MB2>(d, c); run<MB8, MB8>(d, c); run<MF8, MF8>(d, c); run<VF, VF>(d, c); run<VP, VP>(d, c);
    run<MB2, VF>(d, c); run<MB8, VF>(d, c); run<MB2, VP>(d, c); run<MB8, VP>(d, c); run<MB2, VX>(d, c);
    run<MF2, VF>(d, c); run<MF8, VF>(d, c); run<MF8, VP>(d, c);
    run<MB2, MF2>(d, c); run<MB8, MF8>(d, c);
    run<MIX_B, MIX_B>(d, c); run<MIX_F, MIX_F>(d, c); run<MIX_B, VF>(d, c); run<MIX_B, MF2>(d, c);
    printf("priorities (s_setprio):\n");
    run<MB2, VF, 0, 3>(d, c); run<MB2, VF, 3, 0>(d, c); run<VF, MB2, 0, 0>(d, c); run<VF, MB2, 3, 0>(d, c); run<VF, MB2, 0, 3>(d, c);
    run<MB2, VP, 0, 3>(d, c); run<MF2, VF, 0, 3>(d, c); run<MB2, VX, 0, 3>(d, c);
    run<MIX_B, MIX_B, 0, 3>(d, c); run<MIX_B, VF, 0, 3>(d, c); run<MB2, MIX_B, 0, 3>(d, c); run<MB2, MF2, 0, 3>(d, c);
    printf("dependent chains and in-stream mixes:\n");
    run<MB1, IDLE>(d, c); run<MF1, IDLE>(d, c); run<MIX_B1, IDLE>(d, c); run<MIX_B2, IDLE>(d, c); run<MIX_B6, IDLE>(d, c); run<MIX_B1D, IDLE>(d, c);
    run<MB1, VF>(d, c); run<VF, MB1>(d, c); run<MB1, MB1>(d, c); run<MB1, MB2>(d, c); run<MF1, VF>(d, c); run<MF1, MF1>(d, c); run<MB1, MF1>(d, c);
    run<MIX_B1D, MIX_B1D>(d, c); run<MIX_B1, MIX_B1>(d, c); run<MIX_B2, MIX_B2>(d, c); run<MIX_B6, MIX_B6>(d, c); run<MIX_B6, VF>(d, c);
    return 0;
}
