// Micro-benchmark: cost of feeding v_mfma_f32_16x16x4_f32 its A operand from LDS, as the MLP chains of the product do
// (rgl_mlp_chain.h layer_mfma), versus from registers, and a software-prefetched variant; plus relu instruction forms.
// Build: hipcc --offload-arch=gfx950 -O3 -I../../relationalgraphlearning_amd/csrc -I../../include mfma_lds.hip -o mfma_lds
#include "rgl_mlp_chain.h"
#include <cstdio>

namespace {

// prefetched variant of layer_mfma: the fragments of k-group (it+1) are loaded before the MFMAs of group it issue
template <int IN, int OUT>
__device__ __forceinline__ void layer_mfma_pf(const float* frags, const f32x4 (&in)[Tiles<IN>::v], f32x4 (&out)[Tiles<OUT>::v],
                                              int lane, const float* bias) {
    constexpr int IT = Tiles<IN>::v, OT = Tiles<OUT>::v;
    const int q = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) out[ot] = *reinterpret_cast<const f32x4*>(&bias[16 * ot + 4 * q]);
    float w[2][4][OT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) w[0][r][ot] = frags[((ot * IT + 0) * 4 + r) * 64 + lane];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        load_fence();
        if (it + 1 < IT) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) w[(it + 1) & 1][r][ot] = frags[((ot * IT + it + 1) * 4 + r) * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (it == IT - 1 && r >= LastTileSteps<IN>::v) continue;
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) out[ot] = mfma4(w[it & 1][r][ot], in[it][r], out[ot]);
        }
    }
    load_fence();
}

template <int MODE>   // 0: LDS frags as shipped, 1: prefetched, 2: A operand from a register (no LDS), 3: b128 frag reads
__global__ __launch_bounds__(1024) void chain(float* out, const float* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = 100;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 7 * 7 * 4 * 64 + 256; i += blockDim.x) lds[i] = in[i & 255];
    __syncthreads();
    const float* frags = lds;
    const float* bias = lds + 7 * 7 * 4 * 64;
    f32x4 a[7], b[7];
    for (int t = 0; t < 7; ++t) a[t] = f32x4{in[lane], in[lane + 1], in[lane + 2], in[lane + 3]};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            layer_mfma<D, D, true>(frags, a, b, lane, bias);
            layer_mfma<D, D, true>(frags, b, a, lane, bias);
        } else if (MODE == 1) {
            layer_mfma_pf<D, D>(frags, a, b, lane, bias);
            layer_mfma_pf<D, D>(frags, b, a, lane, bias);
        } else if (MODE == 2) {
            const float w = in[lane];
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
                for (int t = 0; t < 7; ++t) b[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 25; ++k)
#pragma unroll
                    for (int t = 0; t < 7; ++t) b[t] = mfma4(w, a[k % 7][k & 3], b[t]);
#pragma unroll
                for (int t = 0; t < 7; ++t) a[t] = b[t];
            }
        } else {
            // fragments stored so that one ds_read_b128 brings the 4 r-steps of (ot, it): [(ot*IT+it)*64 + lane][4]
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
                for (int t = 0; t < 7; ++t) b[t] = *reinterpret_cast<const f32x4*>(&bias[16 * t + 4 * (lane >> 4)]);
#pragma unroll
                for (int kt = 0; kt < 7; ++kt) {
                    load_fence();
                    f32x4 w[7];
#pragma unroll
                    for (int t = 0; t < 7; ++t) w[t] = *reinterpret_cast<const f32x4*>(&frags[((t * 7 + kt) * 64 + lane) * 4]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (kt == 6 && r >= 1) continue;
#pragma unroll
                        for (int t = 0; t < 7; ++t) b[t] = mfma4(w[t][r], a[kt][r], b[t]);
                    }
                }
                load_fence();
#pragma unroll
                for (int t = 0; t < 7; ++t) a[t] = b[t];
            }
        }
    }
    float r = 0;
    for (int t = 0; t < 7; ++t) r += a[t][0] + a[t][3];
    out[blockIdx.x * blockDim.x + tid] = r;
}

enum { MAXI, MAXF, MED3, ROW_NOW_I, ROW_NOW_F, ROW_ABS, ROW_ABS_PLAIN, ROW_FOLD, N_KIND };
const char* names[N_KIND] = {"v_max_i32 x,0", "v_max_f32 x,0", "v_med3_f32", "row: mul_dpp,fmac_dpp,max_i32,add", "row: mul_dpp,fmac_dpp,max_f32,add",
                             "row: mul_dpp,fmac_dpp,add|t|", "row: mul,fma,add|t| (no dpp)", "row: fmac_dpp, max_f32 acc-fold (2 op)"};

template <int KIND>
__global__ __launch_bounds__(1024) void valu(float* out, const float* in, int iters) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x & 255] + i;
    const float a = in[1], b = in[2];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == MAXI) asm volatile("v_max_i32 %0, %0, 0" : "+v"(v[i]));
            if (KIND == MAXF) asm volatile("v_max_f32 %0, %0, 0" : "+v"(v[i]));
            if (KIND == MED3) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(v[i]) : "v"(a));
            if (KIND == ROW_NOW_I || KIND == ROW_NOW_F) {
                float t;
                asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(b), "v"(v[i]));
                if (KIND == ROW_NOW_I) asm volatile("v_max_i32 %0, %0, 0" : "+v"(t));
                else asm volatile("v_max_f32 %0, %0, 0" : "+v"(t));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(i + 1) & 15]) : "v"(t));
            }
            if (KIND == ROW_ABS) {
                float t;
                asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(b), "v"(v[i]));
                asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(v[(i + 1) & 15]) : "v"(t));
            }
            if (KIND == ROW_ABS_PLAIN) {
                float t;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(t) : "v"(b), "v"(v[i]));
                asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(v[(i + 1) & 15]) : "v"(t));
            }
            if (KIND == ROW_FOLD) {
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(b), "v"(a));
                asm volatile("v_max_f32 %0, %0, 0" : "+v"(v[i]));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class K>
float run(K kern, float* d, const float* in, int threads, size_t lds, int iters, float slots) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, d, in, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e6f / (iters * slots) / (threads / 256.f);
}

template <int K>
void vrow(float* d, const float* in) {
    printf("  %-44s %8.3f %8.3f %8.3f\n", names[K], run(valu<K>, d, in, 256, 0, 2000, 16.f), run(valu<K>, d, in, 512, 0, 2000, 16.f),
           run(valu<K>, d, in, 1024, 0, 2000, 16.f));
    if constexpr (K + 1 < N_KIND) vrow<K + 1>(d, in);
}

}  // namespace

int main() {
    float *d, *in;
    (void)hipMalloc(&d, 256 * 1024 * sizeof(float));
    (void)hipMalloc(&in, 512 * sizeof(float));
    float h[512];
    for (int i = 0; i < 512; ++i) h[i] = 0.001f * ((i % 37) + 1);
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const size_t lds = (7 * 7 * 4 * 64 + 256) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chain<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chain<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chain<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("100->100 layer chain, ns per MFMA per SIMD (pure MFMA = 13.5); columns: 1, 2, 4 waves per SIMD\n");
    const float mf = 2 * 175.f;
    printf("  %-44s %8.3f %8.3f %8.3f\n", "LDS frags, as shipped (layer_mfma)", run(chain<0>, d, in, 256, lds, 200, mf),
           run(chain<0>, d, in, 512, lds, 200, mf), run(chain<0>, d, in, 1024, lds, 200, mf));
    printf("  %-44s %8.3f %8.3f %8.3f\n", "LDS frags, next k-group prefetched", run(chain<1>, d, in, 256, lds, 200, mf),
           run(chain<1>, d, in, 512, lds, 200, mf), run(chain<1>, d, in, 1024, lds, 200, mf));
    printf("  %-44s %8.3f %8.3f %8.3f\n", "A operand in a register", run(chain<2>, d, in, 256, lds, 200, mf),
           run(chain<2>, d, in, 512, lds, 200, mf), run(chain<2>, d, in, 1024, lds, 200, mf));
    printf("  %-44s %8.3f %8.3f %8.3f\n", "LDS frags as b128 (4 k-steps per read)", run(chain<3>, d, in, 256, lds, 200, mf),
           run(chain<3>, d, in, 512, lds, 200, mf), run(chain<3>, d, in, 1024, lds, 200, mf));
    printf("VALU, ns per slot per SIMD; columns: 1, 2, 4 waves per SIMD\n");
    vrow<0>(d, in);
    return 0;
}
