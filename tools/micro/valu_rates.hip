// Micro-benchmark: issue cost of the VALU instruction forms the rank-1 row phase can be written in (gfx950).
// Every kernel streams one instruction pattern over independent registers; reported: ns per wave-instruction per SIMD
// stream at 1, 2 and 4 waves per SIMD (256 workgroups = one per CU), and the "elements" rate of whole row-phase bodies.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { FMA, FMAC_DPP, MUL_DPP, MAXI, ADD, ADD_ABS, FMA_SGPR, PK_FMA, PK_FMA_SGPR, PK_MUL, PK_ADD, EXP, RCP, MFMA4, MFMA4_LDS,
       ROW_NOW, ROW_3OP, ROW_PK, ROW_ABS, N_KIND };
const char* names[N_KIND] = {"v_fma_f32", "v_fmac_f32_dpp bcast", "v_mul_f32_dpp bcast", "v_max_i32", "v_add_f32", "v_add_f32 |abs|",
                             "v_fma_f32 sgpr", "v_pk_fma_f32", "v_pk_fma_f32 sgpr", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32",
                             "v_rcp_f32", "mfma16x16x4f32", "mfma16x16x4f32+ds_read", "row body now (4 dpp ops/elt)",
                             "row body 3-op (fma sgpr,max,fmac)", "row body pk (pk_fma,2max,pk_fma /2elt)",
                             "row body abs (mul,fmac,add|t| /elt)"};

template <int KIND>
__global__ __launch_bounds__(1024) void body(float* out, const float* in, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = in[i & 255];
    __syncthreads();
    float v[16];
    f32x2 p[8];
    f32x4 acc[4];
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x & 255] + i;
    for (int i = 0; i < 8; ++i) p[i] = f32x2{v[i], v[i + 8]};
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = in[1], b = in[2];
    const f32x2 ab = {a, b};
    float s0 = __builtin_amdgcn_readfirstlane(in[3]), s1 = __builtin_amdgcn_readfirstlane(in[4]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == FMAC_DPP) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == MUL_DPP) asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(v[i]) : "v"(a), "v"(b));
            if (KIND == MAXI) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            if (KIND == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            if (KIND == ADD_ABS) asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(v[i]) : "v"(a));
            if (KIND == FMA_SGPR) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(a), "v"(b), "s"(s0));
            if (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i & 7]) : "v"(ab), "v"(ab));
            if (KIND == PK_FMA_SGPR) {
                unsigned long long sp = ((unsigned long long)__float_as_uint(s1) << 32) | __float_as_uint(s0);
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[i & 7]) : "v"(ab), "v"(ab), "s"(sp));
            }
            if (KIND == PK_MUL) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i & 7]) : "v"(ab), "v"(ab));
            if (KIND == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(ab));
            if (KIND == EXP) asm volatile("v_exp_f32 %0, %1" : "=v"(v[i]) : "v"(a));
            if (KIND == RCP) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[i]) : "v"(a));
            if (KIND == MFMA4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
            if (KIND == MFMA4_LDS) {
                const float w = lds[(i * 64 + threadIdx.x + it) & 4095];
                acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc[i & 3], 0, 0, 0);
            }
            // whole row-phase bodies: "elements" = 16 per unrolled pass (ROW_PK, ROW_ABS: see element count below)
            if (KIND == ROW_NOW) {       // t = b_i*y ; t += a_i*uw_i ; t = relu ; acc += t      (a_i, b_i by DPP broadcast)
                float t;
                asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(b), "v"(v[i]));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(t) : "v"(0));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(i + 1) & 15]) : "v"(t));
            }
            if (KIND == ROW_3OP) {       // t = r*y + u(sgpr) ; t = relu ; acc += a*t
                float t;
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "v"(b), "s"(s0));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(t) : "v"(0));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(t));
            }
            if (KIND == ROW_PK) {        // two elements: t2 = r2*y + u2(sgpr pair) ; relu x2 ; acc2 += a2*t2
                unsigned long long sp = ((unsigned long long)__float_as_uint(s1) << 32) | __float_as_uint(s0);
                f32x2 t;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(ab), "v"(ab), "s"(sp));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(t.x) : "v"(0));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(t.y) : "v"(0));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i & 7]) : "v"(ab), "v"(t));
            }
            if (KIND == ROW_ABS) {       // t = b_i*y ; t += a_i*uw_i ; acc += |t|   (sum t handled elsewhere)
                float t;
                asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(b), "v"(v[i]));
                asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(v[(i + 1) & 15]) : "v"(t));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
float run(float* d, const float* in, int threads) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((body<KIND>), dim3(256), dim3(threads), 0, 0, d, in, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // ns per unrolled slot per SIMD: a SIMD runs threads/256 waves, each issuing iters*16 slots
    return best * 1e6f / (iters * 16.f) / (threads / 256.f);
}

template <int K>
void row(float* d, const float* in) {
    printf("  %-44s %8.3f %8.3f %8.3f\n", names[K], run<K>(d, in, 256), run<K>(d, in, 512), run<K>(d, in, 1024));
    if constexpr (K + 1 < N_KIND) row<K + 1>(d, in);
}

int main() {
    float *d, *in;
    (void)hipMalloc(&d, 256 * 1024 * sizeof(float));
    (void)hipMalloc(&in, 256 * sizeof(float));
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 0.001f * (i + 1);
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    printf("ns per slot per SIMD (slot = 1 instruction, or one row body); columns: 1, 2, 4 waves per SIMD\n");
    row<0>(d, in);
    return 0;
}
