// rgl_head_body.h -- stage 2 of "value of the sibling children" as device code: rows [t_c | H_{L-1}[robot]] -> value (+ the search's
// select / back-up / root steps for the parents a workgroup owns).  The body of robot_head_kernel (rgl_head.hip); children_deep_kernel
// (rgl_deep.hip) runs the same code as a trailing phase over its own parents' rows (round 4: no second launch, see there).
// Follows (reference paths): crowd_nav/policy/graph_model.py:124-127 (last layer), value_estimator.py:9,18-19.
#pragma once
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
    int own_rows;                 // 1: workgroup b takes the tiles of its parents' rows even without tail work (the trailing head phase of
                                  // children_deep_kernel: the rows are the workgroup's own)
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

// weight image of the head into LDS (every thread of an 8-wave workgroup; a barrier must follow)
template <int D1, int D2, int D3>
__device__ __forceinline__ void head_load_image(float* lds, const HeadArgs& a, int tid) {
    using LO = HeadLds<D1, D2, D3>;
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
}

// the tiles of the workgroup's rows (wave by wave), then -- owned rows only -- the tail steps.  `block` / `n_blocks`: the
// workgroup's index and count in the tile distribution (blockIdx.x / gridDim.x in robot_head_kernel).
template <int D1, int D2, int D3>
__device__ __forceinline__ void head_rows_and_tail(float* lds, const HeadArgs& a, int block, int n_blocks) {
    using LO = HeadLds<D1, D2, D3>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const float b4 = a.b4[0];
    // tile t -> workgroup t % grid, wave (t / grid) % 8: the tiles of the last, partial round land on DIFFERENT workgroups
    // (one extra tile per CU) instead of filling whole workgroups -- the kernel is MFMA-paced per SIMD, so a SIMD whose
    // four waves all carry an extra tile would set the kernel's time.  With a tail: the tiles of the workgroup's own rows.
    const bool owned = a.tail.enabled != 0 || a.own_rows != 0;
    const int p_first = owned ? block * a.parents_per_wg : 0;
    const int k_b = owned ? (a.P - p_first < a.parents_per_wg ? a.P - p_first : a.parents_per_wg) : 0;
    const int row_lo = owned ? p_first * a.A : 0, row_hi = owned ? (p_first + k_b) * a.A : a.M;
    const int t_first = owned ? wave : block + n_blocks * wave;
    const int t_step = owned ? kHeadWaves : n_blocks * kHeadWaves;
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
    if (a.tail.enabled) {
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

// HeadArgs of the shipped heads for the rows of P = M / A parents; `slots` = workgroups the launch keeps resident.  With the search's
// tail arguments: parents per workgroup and whether the launch runs the whole back-up chain (returned in *chain).
inline void head_args_for(const RglGraph* g, const RglMlp* h, int hv, const float* rows, int M, float* value, const float* image,
                          const void* tail, size_t tail_bytes, int A, int slots, HeadArgs* out, int* chain_out) {
    HeadArgs& ha = *out;
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
    ha.A = A; ha.P = A > 0 ? M / A : 0; ha.parents_per_wg = 1; ha.own_rows = 0;
    const TailArgs* ta = (tail && tail_bytes == sizeof(TailArgs) && ((const TailArgs*)tail)->enabled && A > 0 && M % A == 0)
                             ? (const TailArgs*)tail : nullptr;
    static const bool tail_off = [] { const char* e = getenv("RGL_FUSED_NO_TAIL"); return e && e[0] == '1'; }();
    int chain = 0;
    if (ta && !tail_off) {
        // parents per workgroup: one workgroup slot per parent block, whole roots at the deepest level where that keeps >= half
        // of the slots busy (as in the fused children kernel)
        const int P = ha.P;
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
    *chain_out = chain;
}

}  // namespace
