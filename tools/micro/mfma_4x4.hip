// v_mfma_f32_4x4x1_16B_f32 on gfx950: (1) which lane / register holds what (the layout rgl_mlp_chain.h's partial-tile product
// assumes), (2) issue rate against v_mfma_f32_16x16x4_f32.   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_4x4.hip -o tools/micro/mfma_4x4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

template <int KIND>
__global__ void rate_kernel(float* out, int iters) {
    const int l = threadIdx.x & 63;
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float x = 1.f + l * 1e-3f, y = 1.f - l * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND == 0) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[k], 0, 0, 0);
            else acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[k], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (float)(t1 - t0);
}

int main() {
    float *a, *b, *d;
    hipMalloc(&a, 64 * 4); hipMalloc(&b, 64 * 4); hipMalloc(&d, 64 * 16);
    std::vector<float> ha(64), hb(64), hd(256);
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = ldexpf(1.f, l); }            // a * b identifies (lane of a, lane of b): odd part x power of two
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            // assumed: block = l / 4; D[lane 4 blk + j][reg i] = A[blk][i] * B[blk][j] with A[blk][i] from lane 4 blk + i, B[blk][j] from lane 4 blk + j
            const int blk = l / 4, j = l % 4, i = r;
            const float want = ha[4 * blk + i] * hb[4 * blk + j];
            if (hd[l * 4 + r] != want) {
                if (bad < 8) {
                    int la = -1, lb = -1;
                    for (int x = 0; x < 64; ++x) for (int y = 0; y < 64; ++y) if (ha[x] * hb[y] == hd[l * 4 + r]) { la = x; lb = y; }
                    printf("lane %d reg %d: got %.0f = a[lane %d] * b[lane %d], assumed a[%d] * b[%d]\n", l, r, hd[l * 4 + r], la, lb, 4 * blk + i, 4 * blk + j);
                }
                ++bad;
            }
        }
    printf("4x4x1 16-block layout as assumed (block = lane / 4, D[4 blk + j][reg i] = A[4 blk + i] * B[4 blk + j]): %s (%d mismatches)\n",
           bad ? "NO" : "yes", bad);
    float* out;
    const int grid = 1024, thr = 256, iters = 20000;
    hipMalloc(&out, (grid * thr + 1) * 4);
    for (int kind = 0; kind < 2; ++kind) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(grid), dim3(thr), 0, 0, out, iters);
            else hipLaunchKernelGGL(rate_kernel<1>, dim3(grid), dim3(thr), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float cyc; hipMemcpy(&cyc, out + grid * thr, 4, hipMemcpyDeviceToHost);
        const double n = 8.0 * iters;                   // MFMAs per wave
        const double flop = (kind == 0 ? 2048.0 : 512.0) * n * grid * (thr / 64);
        printf("%s: %.3f ms, %.1f shader clocks per MFMA per wave (1 wave per SIMD), %.1f TFLOP/s\n",
               kind == 0 ? "v_mfma_f32_16x16x4_f32 " : "v_mfma_f32_4x4x1_16B_f32", ms, cyc / n, flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
