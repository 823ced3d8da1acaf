// rgl_head.hip -- stage 2 of "value of the sibling children": rows [t_c | H_{L-1}[robot]] -> value.
//   h = relu(t W_last) (+skip), value head 32 -> D1 -> D2 -> D3 -> 1; one register-resident MFMA chain per 16 children,
//   weights as pre-permuted A fragments in LDS.
// Follows (reference paths): crowd_nav/policy/graph_model.py:124-127 (last layer), value_estimator.py:9,18-19.
#include "rgl_mlp_chain.h"
#include "rgl_tail.h"

namespace {

// ------------------------------------------------------------------------------------------------
// stage 2:  rows [t | hprev] -> value
// ------------------------------------------------------------------------------------------------
struct HeadArgs {
    const float* w_last;          // [32][32] last GCN layer
    const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;   // value head, k-major
    int skip;
    const float* rows;            // [M][64]
    float* value;                 // [M]
    int M, n_tiles;
    const float* image;           // null, or the packed weight image (FusedLds<32,100,100> layout): its head vectors and fragments
    // round 3: with `tail.enabled` workgroup b owns the rows of parents [b k, (b + 1) k) (A rows each) and, behind a workgroup
    // barrier, runs the search's select / back-up / root steps for them (rgl_tail.h) -- as the fused children kernel does
    int A, parents_per_wg, P;
    TailArgs tail;
};

template <int D1, int D2, int D3>
struct HeadLds {
    static constexpr int f_last = 0;
    static constexpr int f1 = f_last + 2 * 2 * 4 * 64;
    static constexpr int f2 = f1 + Tiles<D1>::v * 2 * 4 * 64;
    static constexpr int f3 = f2 + Tiles<D2>::v * Tiles<D1>::v * 4 * 64;
    static constexpr int b1 = f3 + Tiles<D3>::v * Tiles<D2>::v * 4 * 64;
    static constexpr int b2 = b1 + Tiles<D1>::v * 16;
    static constexpr int b3 = b2 + Tiles<D2>::v * 16;
    static constexpr int w4 = b3 + Tiles<D3>::v * 16;
    static constexpr int total = w4 + Tiles<D3>::v * 16;
};

constexpr int kHeadThreads = 512;     // 8 waves share one weight image; two workgroups per CU -> 4 waves/SIMD
constexpr int kHeadWaves = kHeadThreads / 64;

template <int D1, int D2, int D3>
__global__ __launch_bounds__(kHeadThreads, 2) void robot_head_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using LO = HeadLds<D1, D2, D3>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    if constexpr (D1 == 32 && D2 == 100 && D3 == 100) {
        if (a.image) {
            // fragments and vectors are two contiguous blocks of the packed image, in this kernel's own order
            using FL = FusedLds<32, 100, 100>;
            static_assert(FL::scratch - FL::f_last == LO::b1 && FL::f_last - FL::b1 == LO::total - LO::b1, "same blocks in both layouts");
            copy_image<FL::scratch - FL::f_last, kHeadThreads>(lds + LO::f_last, a.image + FL::f_last, tid);
            copy_image<FL::f_last - FL::b1, kHeadThreads>(lds + LO::b1, a.image + FL::b1, tid);
        }
    }
    if (!(D1 == 32 && D2 == 100 && D3 == 100 && a.image)) {
    fill_frags<XD, XD, kHeadThreads>(lds + LO::f_last, a.w_last, tid);
    fill_frags<XD, D1, kHeadThreads>(lds + LO::f1, a.w1, tid);
    fill_frags<D1, D2, kHeadThreads>(lds + LO::f2, a.w2, tid);
    fill_frags<D2, D3, kHeadThreads>(lds + LO::f3, a.w3, tid);
    fill_bias<D1>(lds + LO::b1, a.b1, tid, kHeadThreads);
    fill_bias<D2>(lds + LO::b2, a.b2, tid, kHeadThreads);
    fill_bias<D3>(lds + LO::b3, a.b3, tid, kHeadThreads);
    fill_bias<D3>(lds + LO::w4, a.w4, tid, kHeadThreads);      // w4 is [D3][1]: same padded vector layout as a bias
    }
    __syncthreads();
    const float b4 = a.b4[0];
    // tile t -> workgroup t % grid, wave (t / grid) % 8: the tiles of the last, partial round land on DIFFERENT workgroups
    // (one extra tile per CU) instead of filling whole workgroups -- the kernel is MFMA-paced per SIMD, so a SIMD whose
    // four waves all carry an extra tile would set the kernel's time.  With a tail: the tiles of the workgroup's own rows.
    const bool owned = a.tail.enabled != 0;
    const int p_first = owned ? blockIdx.x * a.parents_per_wg : 0;
    const int k_b = owned ? (a.P - p_first < a.parents_per_wg ? a.P - p_first : a.parents_per_wg) : 0;
    const int row_lo = owned ? p_first * a.A : 0, row_hi = owned ? (p_first + k_b) * a.A : a.M;
    const int t_first = owned ? wave : blockIdx.x + gridDim.x * wave;
    const int t_step = owned ? kHeadWaves : gridDim.x * kHeadWaves;
    const int t_end = (row_hi - row_lo + 15) / 16;
    for (int tile = t_first; tile < t_end; tile += t_step) {
        const int row = row_lo + 16 * tile + n;
        const int rc = row < row_hi ? row : row_hi - 1;
        const float* src = a.rows + (size_t)rc * 64;
        f32x4 tin[2], hp[2];
        tin[0] = *reinterpret_cast<const f32x4*>(src + 4 * q);
        tin[1] = *reinterpret_cast<const f32x4*>(src + 16 + 4 * q);
        hp[0] = *reinterpret_cast<const f32x4*>(src + 32 + 4 * q);
        hp[1] = *reinterpret_cast<const f32x4*>(src + 48 + 4 * q);
        f32x4 h[2];
        layer_mfma<XD, XD, false>(lds + LO::f_last, tin, h, lane);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = relu1(h[ot][r]);
                if (a.skip) x += hp[ot][r];
                h[ot][r] = x;
            }
        f32x4 a1[Tiles<D1>::v];
        layer_mfma<XD, D1, true>(lds + LO::f1, h, a1, lane, lds + LO::b1);
        relu_tiles<D1>(a1);
        f32x4 a2[Tiles<D2>::v];
        layer_mfma<D1, D2, true>(lds + LO::f2, a1, a2, lane, lds + LO::b2);
        relu_tiles<D2>(a2);
        f32x4 a3[Tiles<D3>::v];
        layer_mfma<D2, D3, true>(lds + LO::f3, a2, a3, lane, lds + LO::b3);
        relu_tiles<D3>(a3);
        float v = 0.f;
#pragma unroll
        for (int ot = 0; ot < Tiles<D3>::v; ++ot) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&lds[LO::w4 + 16 * ot + 4 * q]);
#pragma unroll
            for (int r = 0; r < 4; ++r) v = fmaf(a3[ot][r], w[r], v);
        }
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (q == 0 && row < row_hi) a.value[row] = v + b4;
    }
    if (owned) {
        // every value of the owned parents was written by a wave of this workgroup: a barrier, then the select step (one wave per
        // parent; its tables go where the weight fragments were) and, at the deepest level, the back-up chain and the root decision
        __syncthreads();
        static_assert(LO::total >= kHeadWaves * kTailLdsInts, "the select step's tables fit where the fragments were");
        int* kl = reinterpret_cast<int*>(lds) + wave * kTailLdsInts;
        for (int lp = wave; lp < k_b; lp += kHeadWaves) tail_select(a.tail, p_first + lp, kl);
        if (a.tail.chain) {
            const int W = a.tail.W, lvl = a.tail.level;
            int per_deep = 1;
            for (int l = 0; l < lvl; ++l) per_deep *= W;
            const int r_first = p_first / per_deep, n_roots = k_b / per_deep;
            int per_l = per_deep;
            for (int l = lvl - 1; l >= 1; --l) {
                per_l /= W;
                __syncthreads();
                for (int i = tid; i < n_roots * per_l; i += kHeadThreads) tail_backup(a.tail, l, r_first * per_l + i);
            }
            __syncthreads();
            for (int base = 0; base < n_roots * kRootLanes; base += kHeadThreads) {
                const int i = base + tid, bl = i / kRootLanes;
                tail_root(a.tail, r_first + bl, i % kRootLanes, bl < n_roots);
            }
        }
    }
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
    ha.image = hv == 0 ? image : nullptr;
    ha.w_last = g->Ws[g->num_layer - 1];
    ha.w1 = h->weight[0]; ha.b1 = h->bias[0];
    ha.w2 = h->weight[1]; ha.b2 = h->bias[1];
    ha.w3 = h->weight[2]; ha.b3 = h->bias[2];
    ha.w4 = h->weight[3]; ha.b4 = h->bias[3];
    ha.skip = g->skip_connection;
    ha.rows = rows;
    ha.value = value;
    ha.M = M;
    ha.n_tiles = (M + 15) / 16;
    ha.tail = TailArgs{};
    ha.A = A; ha.P = A > 0 ? M / A : 0; ha.parents_per_wg = 1;
    const TailArgs* ta = (tail && tail_bytes == sizeof(TailArgs) && ((const TailArgs*)tail)->enabled && A > 0 && M % A == 0)
                             ? (const TailArgs*)tail : nullptr;
    static const bool tail_off = [] { const char* e = getenv("RGL_FUSED_NO_TAIL"); return e && e[0] == '1'; }();
    int chain = 0;
    if (ta && !tail_off) {
        // parents per workgroup: one workgroup slot per parent block, whole roots at the deepest level where that keeps >= half
        // of the slots busy (as in the fused children kernel)
        const int P = ha.P;
        const int slots = hv == 0 ? 512 : 256;
        int unit = 1;
        if (ta->chain) {
            long u = 1;
            for (int l = 0; l < ta->level && u <= P; ++l) u *= ta->W;
            if (u <= P && P % u == 0 && (P / u >= slots / 2 || P / u >= 128)) { unit = (int)u; chain = 1; }
            else if (u == 1) chain = 1;
        }
        int k = (P + slots - 1) / slots;
        if (k < 1) k = 1;
        k = ((k + unit - 1) / unit) * unit;
        ha.parents_per_wg = k;
        ha.tail = *ta;
        ha.tail.chain = chain;
    }
    const int rc = hv == 0 ? launch_head<32, 100, 100>(ha, stream) : launch_head<150, 100, 100>(ha, stream);
    if (rc == RGL_OK && ha.tail.enabled && tail_done) *tail_done = chain ? 2 : 1;
    return rc;
}

}  // namespace rgl
