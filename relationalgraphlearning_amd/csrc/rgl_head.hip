// rgl_head.hip -- stage 2 of "value of the sibling children": rows [t_c | H_{L-1}[robot]] -> value.
//   h = relu(t W_last) (+skip), value head 32 -> D1 -> D2 -> D3 -> 1; one register-resident MFMA chain per 16 children,
//   weights as pre-permuted A fragments in LDS.
// Follows (reference paths): crowd_nav/policy/graph_model.py:124-127 (last layer), value_estimator.py:9,18-19.
#include "rgl_head_body.h"

namespace {

template <int D1, int D2, int D3>
__global__ __launch_bounds__(kHeadThreads, 2) void robot_head_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    head_load_image<D1, D2, D3>(lds, a, threadIdx.x);
    __syncthreads();
    head_rows_and_tail<D1, D2, D3>(lds, a, blockIdx.x, gridDim.x);
}

template <int D1, int D2, int D3>
int launch_head(const HeadArgs& ha, hipStream_t st) {
    auto kern = robot_head_kernel<D1, D2, D3>;
    const size_t lds_bytes = (size_t)HeadLds<D1, D2, D3>::total * sizeof(float);
    if (lds_bytes > 64 * 1024)
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds_bytes));
    int grid = (ha.n_tiles + kHeadWaves - 1) / kHeadWaves;
    const int cap = lds_bytes > 80 * 1024 ? 256 : 512;          // resident workgroups: 1 or 2 per CU
    if (grid > cap) grid = cap;
    if (ha.tail.enabled) grid = (ha.P + ha.parents_per_wg - 1) / ha.parents_per_wg;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kHeadThreads), lds_bytes, st, ha);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

// ------------------------------------------------------------------------------------------------
// stage 2 for ANY value head (value_network_dims of the reference's config, crowd_nav/configs: [32, 100, 100, 1] by default):
// x_dim -> d1 -> ... -> 1 with up to RGL_MAX_MLP_LAYERS layers of width <= RGL_MAX_WIDTH.  Same transposed product as above --
// out^T[o][child] = sum_k W[k][o] act^T[k][child] per 16-child tile -- but with run-time loops over the output tiles and k steps:
// the activations of a tile live in the wave's LDS ([feature][16 children], ping-pong), the weight fragments come straight from
// the k-major matrices in global memory (L1 / L2: every wave of the GPU reads the same few KB).  Not a speed-of-light kernel --
// it exists so that a non-default head keeps the MFMA stage-1 kernels instead of dropping the whole search to the general VALU
// kernel (1.8 % of peak, profiles/r01_a).
// ------------------------------------------------------------------------------------------------
struct HeadAnyArgs {
    const float* w_last;          // [32][32] last GCN layer
    RglMlp head;                  // k-major weights
    int skip;
    const float* rows;            // [M][64]
    float* value;                 // [M]
    int M, n_tiles;
};

constexpr int kAnyWaves = 4;
constexpr int kAnyRows = RGL_MAX_WIDTH;                // feature rows of an activation buffer
constexpr int kAnyBuf = kAnyRows * 16;                 // floats per buffer: [feature][16 children]

// act_in [IN][16] (rows >= IN unread) -> act_out [OUT rounded up to 16][16], out = W^T in + b, optional ReLU; padded output rows = 0
__device__ __forceinline__ void any_layer(const float* __restrict__ W, const float* __restrict__ bias, int IN, int OUT, bool relu,
                                          const float* act_in, float* act_out, int lane) {
    const int n = lane & 15, q = lane >> 4;
    const int k_steps = (IN + 3) >> 2, o_tiles = (OUT + 15) >> 4;
    for (int ot = 0; ot < o_tiles; ++ot) {
        const int o_a = 16 * ot + n;                   // A-operand row of this lane
        f32x4 acc = zero4();
        for (int s = 0; s < k_steps; ++s) {
            const int k = 4 * s + q;
            const float a = (k < IN && o_a < OUT) ? W[(size_t)k * OUT + o_a] : 0.f;
            const float b = k < IN ? act_in[k * 16 + n] : 0.f;
            acc = mfma4(a, b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 16 * ot + 4 * q + r;
            float v = o < OUT ? acc[r] + (bias ? bias[o] : 0.f) : 0.f;
            if (relu) v = fmaxf(v, 0.f);
            act_out[o * 16 + n] = o < OUT ? v : 0.f;
        }
    }
}

__global__ __launch_bounds__(kAnyWaves * 64) void robot_head_any_kernel(const HeadAnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    float* b0 = lds + wave * 2 * kAnyBuf;
    float* b1 = b0 + kAnyBuf;
    for (int tile = blockIdx.x * kAnyWaves + wave; tile < a.n_tiles; tile += gridDim.x * kAnyWaves) {
        const int row = 16 * tile + n;
        const int rc = row < a.M ? row : a.M - 1;
        const float* src = a.rows + (size_t)rc * 64;
        // t^T -> b0 rows 0..31, hprev^T -> b1 rows 32..63 (kept until the skip is added)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 tv = *reinterpret_cast<const f32x4*>(src + 16 * h + 4 * q);
            const f32x4 hv = *reinterpret_cast<const f32x4*>(src + 32 + 16 * h + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                b0[(16 * h + 4 * q + r) * 16 + n] = tv[r];
                b1[(XD + 16 * h + 4 * q + r) * 16 + n] = hv[r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        any_layer(a.w_last, nullptr, XD, XD, true, b0, b1, lane);           // h = relu(t W_last) -> b1 rows 0..31
        __builtin_amdgcn_wave_barrier();
        if (a.skip) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = 4 * i + q;                                      // 32 features x 16 children over 64 lanes
                b1[f * 16 + n] += b1[(XD + f) * 16 + n];
            }
            __builtin_amdgcn_wave_barrier();
        }
        float* cur = b1;
        float* nxt = b0;
        for (int l = 0; l < a.head.n_layers; ++l) {
            const bool last = l + 1 == a.head.n_layers;
            any_layer(a.head.weight[l], a.head.bias[l], a.head.dims[l], a.head.dims[l + 1], !last, cur, nxt, lane);
            __builtin_amdgcn_wave_barrier();
            float* t = cur; cur = nxt; nxt = t;
        }
        if (q == 0 && row < a.M) a.value[row] = cur[n];                      // output feature 0 of child n
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

namespace rgl {

// rows [M][64] (stage-1 hand-off) -> value;  1 = no kernel for this head (see head_variant)
int launch_head_rows(const RglGraph* g, const RglMlp* h, const float* rows, int M, float* value, hipStream_t stream,
                     const float* image, const void* tail, size_t tail_bytes, int* tail_done, int A) {
    if (tail_done) *tail_done = 0;
    const int hv = head_variant(*h);
    if (hv < 0) return 1;
    if (hv == 2) {
        HeadAnyArgs aa;
        aa.w_last = g->Ws[g->num_layer - 1];
        aa.head = *h;
        aa.skip = g->skip_connection;
        aa.rows = rows;
        aa.value = value;
        aa.M = M;
        aa.n_tiles = (M + 15) / 16;
        const size_t lds_bytes = (size_t)kAnyWaves * 2 * kAnyBuf * sizeof(float);      // 128 KB: one workgroup per CU
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(robot_head_any_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        int grid = (aa.n_tiles + kAnyWaves - 1) / kAnyWaves;
        if (grid > 256) grid = 256;
        hipLaunchKernelGGL(robot_head_any_kernel, dim3(grid), dim3(kAnyWaves * 64), lds_bytes, stream, aa);
        RGL_LAUNCH_CHECK();
        return RGL_OK;
    }
    HeadArgs ha;
    int chain = 0;
    head_args_for(g, h, hv, rows, M, value, image, tail, tail_bytes, A, hv == 0 ? 512 : 256, &ha, &chain);
    const int rc = hv == 0 ? launch_head<32, 100, 100>(ha, stream) : launch_head<150, 100, 100>(ha, stream);
    if (rc == RGL_OK && ha.tail.enabled && tail_done) *tail_done = chain ? 2 : 1;
    return rc;
}

}  // namespace rgl
