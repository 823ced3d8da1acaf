// rgl_backward_mfma.hip -- MFMA tile kernels for (1) the training path's backward pass on LARGE batches and (2) the forward of models
// outside the shipped shapes.
//
// (1) crowd_nav/utils/trainer.py:110-161,199-250 drive forward + backward over replay batches; the reference's own batch is 100, the
// vector explorer of this build feeds thousands.  rgl_backward.hip gives every scene a 1024-thread workgroup, a VALU loop per product
// and a gradient slab of its own; its time per scene does not shrink with the batch.  Here the same gradients are computed as a
// pipeline of tile kernels, every dense product an fp32 MFMA (v_mfma_f32_16x16x4_f32: exact products, fp32 accumulation -- the
// arithmetic of the VALU kernel up to summation order):
//
//   1. mlp rows (forward only)          X = [w_r(robot); w_h(humans)]              one wave per 16-row tile
//   2. graph_kernel<.., false>          H_L = layers(softmax(X Wa X^T), X)          one workgroup per scene, activations in its LDS
//   3. mlp rows (backward)              value head on H_L[:, 0] / motion head on H_L[:, 1:]  ->  dH_L, head gradients
//   4. graph_kernel<.., true>           recomputes 2., back-propagates  ->  dX, and dWa / dW_l accumulated in REGISTERS over the
//                                       workgroup's scenes (one slab per workgroup, not per scene)
//   5. mlp rows (backward)              w_r / w_h from dX (forward recomputed inside)
//   6. reduce_ranges_kernel             slabs summed in a fixed order (fixed tile -> wave -> slab assignment: deterministic)
//
// "mlp rows" is mlp2_rows_kernel<T0, T2> for the shipped narrow MLPs (in -> 64 -> out, in / out <= 32: weight gradients in
// registers) and mlp_rows_kernel for everything else (any MLP of the ABI; the value head: one workgroup per tile).  The intermediates
// X, H_L, dH_L, dX travel through HBM ([S][N][x_dim] floats each: 10 MB at 4096 scenes of 20 nodes, ~1.3 us of traffic apiece) so that
// each kernel keeps one job; everything else lives in LDS / registers.  Below RGL_BACKWARD_MFMA_MIN scenes of an eager step
// rgl_backward.hip runs.
//
// (2) launch_tiles_forward chains 1., 2. and head rows forward-only for models the shipped-shape kernels (rgl_scene.hip, rgl_fused.hip,
// ...) do not cover -- other embedding MLPs, x_dim = 64 -- instead of the general VALU kernel; sibling scenes of a rollout share
// their crowd's embedded rows.
//
// Envelope of both: embedded_gaussian, gaussian, squared, equal_attention or diagonal similarity (round 5: the three plain-weight
// normalisations), one adjacency for all layers -- or, round 6, one per layer (layerwise graphs) for the softmax and squared
// normalisations at x_dim 32 and N <= 32 --, x_dim 32 | 64, 1-3 layers, N <= 64; any embedding MLPs and heads within the ABI limits.  Outside it: return 1 (the caller falls back to rgl_backward.hip / the general kernel).
//
// Differentiated forward: graph_model.py:99-130, value_estimator.py:11-20, state_predictor.py:28-36, gcn.py:95-128.
#include "rgl_mfma.h"

#include <cstdlib>

// The launchers of this file answer 1 for "not this path" -- and hipErrorInvalidValue is 1 as well (a launch that asks for more LDS
// than a CU has): a runtime failure must never read as "not covered" and fall through to another kernel with the error still pending.
#undef RGL_HIP_TRY
#undef RGL_LAUNCH_CHECK
#define RGL_HIP_TRY(expr)                                                          \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) return e__ == hipErrorInvalidValue ? (int)hipErrorLaunchFailure : (int)e__; \
    } while (0)
#define RGL_LAUNCH_CHECK() RGL_HIP_TRY(hipGetLastError())

// -DRGL_PHASE_TIMING -DRGL_PHASE_HEAD: the phase counters belong to mlp_rows_kernel (the value head) instead of mlp2_rows_kernel
#ifdef RGL_PHASE_HEAD
#define HEAD_PHASE_START() PHASE_START()
#define HEAD_PHASE_MARK(i) PHASE_MARK(i)
#define HEAD_PHASE_FLUSH() PHASE_FLUSH()
#define ROWS2_PHASE_START() do { } while (0)
#define ROWS2_PHASE_MARK(i) do { } while (0)
#define ROWS2_PHASE_FLUSH() do { } while (0)
#else
#define HEAD_PHASE_START() do { } while (0)
#define HEAD_PHASE_MARK(i) do { } while (0)
#define HEAD_PHASE_FLUSH() do { } while (0)
#define ROWS2_PHASE_START() PHASE_START()
#define ROWS2_PHASE_MARK(i) PHASE_MARK(i)
#define ROWS2_PHASE_FLUSH() PHASE_FLUSH()
#endif

namespace {

__device__ __forceinline__ void wave_sync() {         // LDS written by some lanes of the wave, read by others
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier for waves that talk through LDS only: __syncthreads() also waits for every global store of the wave to be
// acknowledged (vmcnt(0): ~2.8 k cycles behind each layer's gradient stores in mlp_rows_kernel), this one orders LDS alone.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// C[MT*16][NTL*16] += A[.][k] B[k][.] over `ksteps` groups of four k: fa(row, k) / fb(k, col) fetch the operand elements
// (LDS or L1-resident weights; the callers clamp / zero what lies outside their matrices).  One A fragment per row tile and one B
// fragment per column tile feed MT*NTL MFMAs.  D row 4 (lane / 16) + r, column lane % 16 is element r of a lane's accumulator.
template <int MT, int NTL, int U, bool PIN, class FA, class FB>
__device__ __forceinline__ void mm_steps(f32x4 (&acc)[MT][NTL], int ks0, FA& fa, FB& fb) {
    const int l16 = threadIdx.x & 15, kk = (threadIdx.x & 63) >> 4;
    float av[U][MT], bv[U][NTL];
#pragma unroll
    for (int u = 0; u < U; ++u) {              // all operand fragments of U k steps in flight, then their MFMAs
        const int k = (ks0 + u) * 4 + kk;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[u][mt] = fa(mt * 16 + l16, k);
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) bv[u][nt] = fb(k, nt * 16 + l16);
    }
    // PIN: every fragment of the batch requested before its first MFMA (left alone, the scheduler sinks the loads to their uses and
    // reuses two registers: an LDS round trip per two k steps -- fine where other waves cover it, not for a workgroup that owns one
    // tile; pinned everywhere, the wide kernels of this file spill)
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) acc[mt][nt] = mfma4(av[u][mt], bv[u][nt], acc[mt][nt]);
    load_fence();            // the next batch's loads stay behind this batch (hoisting them all costs hundreds of VGPRs)
}
template <int MT, int NTL, int U, bool PIN, class FA, class FB>
__device__ __forceinline__ void mm_from(f32x4 (&acc)[MT][NTL], int& ks, int ksteps, FA& fa, FB& fb) {
    for (; ks + U <= ksteps; ks += U) mm_steps<MT, NTL, U, PIN>(acc, ks, fa, fb);
    if constexpr (U > 1) mm_from<MT, NTL, U / 2, PIN>(acc, ks, ksteps, fa, fb);        // the remainder in halving batches
}
template <int MT, int NTL, int UNROLL = 2, bool PIN = false, class FA, class FB>
__device__ __forceinline__ void mm(f32x4 (&acc)[MT][NTL], int ksteps, FA fa, FB fb) {
    int ks = 0;
    mm_from<MT, NTL, UNROLL, PIN>(acc, ks, ksteps, fa, fb);
}
template <int MT, int NTL>
__device__ __forceinline__ void clear(f32x4 (&acc)[MT][NTL]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) acc[mt][nt] = zero4();
}
// f(row, col, value, mt, nt, r) for every element of the block
template <int MT, int NTL, class F>
__device__ __forceinline__ void each(const f32x4 (&acc)[MT][NTL], F f) {
    const int l16 = threadIdx.x & 15, kk = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) f(mt * 16 + 4 * kk + r, nt * 16 + l16, acc[mt][nt][r], mt, nt, r);
}

// dst(idx, src(idx)) for idx = t0, t0 + step, .. < n, the loads U at a time (a plain loop pays one global-memory latency per
// iteration: the compiler does not pipeline across iterations)
template <int U, class Src, class Dst>
__device__ __forceinline__ void gather(int n, int t0, int step, Src src, Dst dst) {
    for (int base = t0; base < n; base += U * step) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * step;
            v[u] = idx < n ? src(idx) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * step;
            if (idx < n) dst(idx, v[u]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rows of an MLP: forward, and backward with parameter gradients
// ------------------------------------------------------------------------------------------------
// row r of a [groups][per][width] tensor embedded in a larger one: p + (r / per) * group_stride + (r % per) * row_stride
struct RowMap {
    float* p;
    int per, row_stride;
    long long group_stride;
};
__device__ __forceinline__ float* row_at(const RowMap& m, int r) {
    const int g = r / m.per;
    return m.p + g * m.group_stride + (long long)(r - g * m.per) * m.row_stride;
}

constexpr int kMaxRowJobs = 2;
struct RowsJob {
    RglMlp m;
    int w_off[RGL_MAX_MLP_LAYERS], b_off[RGL_MAX_MLP_LAYERS];   // inside the job's slab (torch layout [out][in], then the bias)
    int act_off[RGL_MAX_MLP_LAYERS + 1];                        // column of layer l's input inside a row of the activation tile
    int w_lds[RGL_MAX_MLP_LAYERS], w_ld[RGL_MAX_MLP_LAYERS];    // layer l's weights in the workgroup's LDS: float offset, row stride
    int b_lds[RGL_MAX_MLP_LAYERS];
    int weight_floats;                                          // weights + biases of all layers
    int act_ld, d_ld, n_params;
    int n_rows, n_tiles, n_waves;
    int wg_begin, n_wgs, waves_per_wg;
    int coop;                                                   // mlp_rows_kernel: one tile per WORKGROUP (n_waves counts workgroups)
    int wave_floats;                                            // LDS of one wave
    int kind;                                                   // 0: mlp_rows_kernel; 1: head_rows_kernel; 10 T0 + T2: mlp2_rows_kernel<T0, T2>
    int need_din, din_add;
    RowMap in, out, d_out, d_in;      // out: forward-only launches; d_out (null = zeros) / d_in: backward launches
    float* slabs;                     // [n_waves][n_params]
};
struct RowsArgs {
    RowsJob job[kMaxRowJobs];
    int n_jobs, backward;
};

// weights and biases of layers [l0, l1) of the job's MLP -> the workgroup's LDS.  The layers are cut into chunks of U elements per
// thread (element e of a layer: weight e of torch-transposed [in][out], then the biases) and the chunks run as a two-deep pipeline:
// the next chunk's loads are in flight while this one's values are stored.  (Round 5 staged a column at a time, eight loads deep:
// nine dependent trips to L2 / the Infinity Cache -- the weights were just written by the optimizer step -- for the value head.)
template <int U>
__device__ __forceinline__ void stage_layers(const RowsJob& J, float* lds, int l0, int l1) {
    const RglMlp& m = J.m;
    const int step = blockDim.x, span = U * step;
    auto issue = [&](int l, int base, float (&v)[U]) {
        const int out = m.dims[l + 1], nw = m.dims[l] * out;
        const float* __restrict__ W = m.weight[l];
        const float* __restrict__ B = m.bias[l];
#pragma unroll
        for (int u = 0; u < U; ++u) {       // one unconditional load per element (a guarded one is a branch, and the loads serialise)
            const int e = base + threadIdx.x + u * step;
            const float* p = e < nw ? W + e : B + min(e - nw, out - 1);
            v[u] = *p;
        }
    };
    auto store = [&](int l, int base, const float (&v)[U]) {
        const int out = m.dims[l + 1], nw = m.dims[l] * out, n = nw + out, ld = J.w_ld[l];
        const float rcp = 1.0f / (float)out;
        float* Wl = lds + J.w_lds[l];
        float* Bl = lds + J.b_lds[l];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + threadIdx.x + u * step;
            if (e < nw) {
                const int k = (int)(((float)e + 0.5f) * rcp);      // e / out, exact: e < 2^16 and the quotient is at most 256
                Wl[k * ld + (e - k * out)] = v[u];
            } else if (e < n) {
                Bl[e - nw] = v[u];
            }
        }
    };
    if (l0 >= l1) return;
    // chunk (l, base) -> the one behind it; false behind the last one
    auto advance = [&](int& l, int& base) {
        base += span;
        if (base >= (m.dims[l] + 1) * m.dims[l + 1]) { ++l; base = 0; }
        return l < l1;
    };
    float va[U], vb[U];                 // ping and pong (a copy between them would wait for the loads it copies)
    int l = l0, base = 0;
    issue(l, base, va);
    while (true) {
        int ln = l, bn = base;
        bool more = advance(ln, bn);
        if (more) issue(ln, bn, vb);
        store(l, base, va);
        if (!more) break;
        l = ln; base = bn;
        more = advance(ln, bn);
        if (more) issue(ln, bn, va);
        store(l, base, vb);
        if (!more) break;
        l = ln; base = bn;
    }
}

// Workgroups of up to four waves: the MLP's weights are staged once per workgroup in LDS (k-major rows of odd stride: the forward's
// B-operand reads and the transposed reads of the delta products both stay within two-way bank conflicts), then every wave works
// through its own 16-row tiles without further barriers: every layer's activations of the tile in LDS ([16][act_ld]) together with
// two delta buffers ([16][d_ld]).  A wave's first tile writes its gradient slab, later tiles add to it (L2-resident).
//
// Round 6: a workgroup that owns ONE tile (few tiles: the value head at the reference's batch) now runs head_rows_kernel below;
// what stays here is the many-tiles form and, under RGL_HEAD_ROWS_DIRECT=0, the one-tile form as the A/B partner
// (profiles/r06_head_rows.txt: 35.7 us in round 5 -> 30.3 us here -> 23.0 us there).  What this kernel gained on the way: the first
// tile's rows, its upstream gradient and (where they are added to) its input gradients are requested BEFORE the weights, which
// arrive as a pipeline of chunks; the ReLU masks are applied where a delta is produced instead of in passes of their own with a
// barrier each; the barriers order LDS only (__syncthreads() waits for the gradient stores' acknowledgements); a shared tile's
// products deal single column tiles to the eight waves with eight k steps of operands requested at once.
constexpr int kCoopWaves = 8;          // waves of a workgroup that shares one tile (mlp_rows_kernel, coop)
constexpr int kRowsPre = 8;            // elements per lane of a 16-row tile that are requested ahead (rows of up to 32 columns)
__global__ __launch_bounds__(kCoopWaves * 64) void mlp_rows_kernel(const RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l16 = lane & 15;
    const int ji = (a.n_jobs > 1 && (int)blockIdx.x >= a.job[1].wg_begin) ? 1 : 0;
    const RowsJob& J = a.job[ji];
    HEAD_PHASE_START();
    // coop (few tiles, wide layers: the value head): the eight waves of the workgroup share ONE tile -- the column tiles of every
    // product are dealt to them, with a workgroup barrier between phases -- instead of a tile each
    // (coop == 2: the MLP's weights do not fit LDS next to the tile -- path G's 150-100-100-1 head is 121 KB -- and every layer is
    // staged when it is needed, forward and again backward: a tile uses each weight once per direction anyway)
    const bool coop = J.coop != 0, per_layer = J.coop == 2;
    const int WV = coop ? kCoopWaves : 1, wv = coop ? wave : 0;
    const int w = coop ? (int)blockIdx.x - J.wg_begin : ((int)blockIdx.x - J.wg_begin) * J.waves_per_wg + wave;
    const bool mine = !((!coop && wave >= J.waves_per_wg) || w >= J.n_waves);
    const RglMlp& m = J.m;
    const int L = m.n_layers, ald = J.act_ld, dld = J.d_ld;
    // element-wise passes over the tile: lane -> (row rr = lane / 4, columns c4, c4 + 4, ..)
    const int rr = lane >> 2, c4 = lane & 3;
    const int d0 = m.dims[0], d0p = (d0 + 3) & ~3, dL = m.dims[L], dLp = (dL + 3) & ~3;
    const bool relu_top = m.last_relu != 0;
    // what the first tile reads from global memory, requested ahead of the weights (rows of up to 32 columns: the value head's 32)
    const bool pre_in = mine && d0p <= 4 * kRowsPre, pre_dout = mine && a.backward && dLp <= 4 * kRowsPre;
    const bool pre_din = mine && a.backward && J.need_din && J.din_add && d0 <= 4 * kRowsPre;
    float xin[kRowsPre], dov[kRowsPre], dinv[kRowsPre];
    {
        const int r0 = w * 16;
        const bool rok = mine && r0 + rr < J.n_rows;
        const int rrow = rok ? r0 + rr : (J.n_rows > 0 ? J.n_rows - 1 : 0);
        // unconditional loads from clamped addresses, the guards where the values are used, behind the barrier (a guarded load -- or
        // a guarded use the load can sink to -- is a branch with a wait inside: 24 round trips one after the other)
#pragma unroll
        for (int u = 0; u < kRowsPre; ++u) xin[u] = dov[u] = dinv[u] = 0.f;
        if (pre_in) {
            const float* src = row_at(J.in, rrow);
#pragma unroll
            for (int u = 0; u < kRowsPre; ++u) xin[u] = src[min(c4 + 4 * u, d0 - 1)];
        }
        if (pre_dout && J.d_out.p) {
            const float* src = row_at(J.d_out, rrow);
#pragma unroll
            for (int u = 0; u < kRowsPre; ++u) dov[u] = src[min(c4 + 4 * u, dL - 1)];
        }
        if (pre_din) {
            const float* src = row_at(J.d_in, rrow);
#pragma unroll
            for (int u = 0; u < kRowsPre; ++u) dinv[u] = src[min(c4 + 4 * u, d0 - 1)];
        }
    }
    if (!per_layer) stage_layers<16>(J, lds, 0, L);
    lds_barrier();
    HEAD_PHASE_MARK(0);
    if (!mine) {
        HEAD_PHASE_FLUSH();
        return;
    }
    auto sync = [&]() { if (coop) lds_barrier(); else wave_sync(); };
    float* acts = lds + J.weight_floats + (coop ? 0 : wave) * (16 * ald + 32 * dld);
    float* dcur = acts + 16 * ald;
    float* dnxt = dcur + 16 * dld;
    float* slab = J.slabs + (size_t)w * J.n_params;
    bool first = true;
    // C[16][NTL * 16] blocks of a product dealt to the waves that share the tile: one column tile each (coop) or pairs (a wave alone)
    auto product = [&](int cols, int ksteps, auto fa, auto fb, auto init, auto emit) {
        if (coop) {
            for (int jt = wv; jt * 16 < cols; jt += WV) {
                f32x4 acc[1][1];
                acc[0][0] = init(jt * 16 + l16);
                mm<1, 1, 8, true>(acc, ksteps, fa, [&](int k, int c) { return fb(k, jt * 16 + c); });
                each<1, 1>(acc, [&](int row, int c, float v, int, int, int) { emit(row, jt * 16 + c, v); });
            }
        } else {
            for (int jt = 0; jt * 16 < cols; jt += 2) {
                f32x4 acc[1][2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[0][nt] = init((jt + nt) * 16 + l16);
                mm<1, 2, 4, true>(acc, ksteps, fa, [&](int k, int c) { return fb(k, jt * 16 + c); });
                each<1, 2>(acc, [&](int row, int c, float v, int, int, int) { emit(row, jt * 16 + c, v); });
            }
        }
    };
    for (int t = w; t < J.n_tiles; t += J.n_waves, first = false) {
        const int r0 = t * 16;
        const bool rok = r0 + rr < J.n_rows;
        const int rrow = rok ? r0 + rr : J.n_rows - 1;
        // input rows (zero beyond the end, zero in the padding columns)
        if (first && pre_in) {
#pragma unroll
            for (int u = 0; u < kRowsPre; ++u)
                if (c4 + 4 * u < d0p) acts[rr * ald + c4 + 4 * u] = (rok && c4 + 4 * u < d0) ? xin[u] : 0.f;
        } else {
            const float* src = row_at(J.in, rrow);
            gather<8>(d0p, c4, 4, [&](int c) { return (rok && c < d0) ? src[c] : 0.f; }, [&](int c, float v) { acts[rr * ald + c] = v; });
        }
        sync();
        HEAD_PHASE_MARK(1);
        for (int l = 0; l < L; ++l) {
            const int in = m.dims[l], out = m.dims[l + 1], inp = (in + 3) & ~3, outp = (out + 3) & ~3;
            const int ioff = J.act_off[l], ooff = J.act_off[l + 1];
            const bool relu = (l != L - 1) || relu_top;
            const float* W = lds + J.w_lds[l];
            const float* b = lds + J.b_lds[l];
            const int wld = J.w_ld[l];
            if (per_layer) {              // the previous layer's reads of the region ended at its barrier
                stage_layers<16>(J, lds, l, l + 1);
                lds_barrier();
            }
            if (coop && out <= 4) {
                // up to four outputs (the value head's last layer): a dot product per row on the VALU -- lane (row rr, quarter c4) sums
                // every fourth input, the four lanes of a row add up -- instead of one wave's chain of in / 4 dependent MFMAs
                // while seven waves wait
                if (wv == 0) {
                    float sum[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int k0 = c4; k0 < inp; k0 += 32) {
                        float av[8], wv4[8][4];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int k = k0 + 4 * u;
                            // (a select between a load and a constant is turned into a branch around the load: a factor instead)
                            av[u] = acts[rr * ald + ioff + min(k, inp - 1)] * (k < inp ? 1.f : 0.f);
#pragma unroll
                            for (int o = 0; o < 4; ++o) wv4[u][o] = W[min(k, in - 1) * wld + min(o, out - 1)];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
#pragma unroll
                            for (int o = 0; o < 4; ++o) sum[o] = fmaf(av[u], wv4[u][o], sum[o]);
                    }
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        sum[o] += __shfl_xor(sum[o], 1);
                        sum[o] += __shfl_xor(sum[o], 2);
                        if (c4 == o && o < outp) {
                            const float v = sum[o] + b[min(o, out - 1)];
                            acts[rr * ald + ooff + o] = o < out ? (relu ? fmaxf(v, 0.f) : v) : 0.f;
                        }
                    }
                }
                sync();
                continue;
            }
            // clamped addresses, no guards (a guard becomes a branch and an LDS round trip per k step): the padding columns of the
            // activations are zero, and output columns past the end are never stored
            product(out, inp >> 2,
                    [&](int row, int k) { return acts[row * ald + ioff + k]; },
                    [&](int k, int col) { return W[min(k, in - 1) * wld + min(col, out - 1)]; },
                    [&](int col) { const float b0 = b[min(col, out - 1)], bv = col < out ? b0 : 0.f; return f32x4{bv, bv, bv, bv}; },
                    [&](int row, int col, float v) {
                        if (col < outp) acts[row * ald + ooff + col] = col < out ? (relu ? fmaxf(v, 0.f) : v) : 0.f;
                    });
            sync();
        }
        HEAD_PHASE_MARK(2);
        if (!a.backward) {
            const int ooff = J.act_off[L];
            float* dst = row_at(J.out, rrow);
            if (rok && wv == 0)
                for (int c = c4; c < dL; c += 4) dst[c] = acts[rr * ald + ooff + c];
            sync();
            continue;
        }
        {   // upstream gradient of the tile's rows, through the top layer's ReLU if it has one
            const int ooff = J.act_off[L];
            auto put = [&](int c, float v) {
                dcur[rr * dld + c] = (relu_top && !(acts[rr * ald + ooff + c] > 0.f)) ? 0.f : v;
            };
            if (first && pre_dout) {
#pragma unroll
                for (int u = 0; u < kRowsPre; ++u)
                    if (c4 + 4 * u < dLp) put(c4 + 4 * u, (J.d_out.p && rok && c4 + 4 * u < dL) ? dov[u] : 0.f);
            } else {
                const float* src = J.d_out.p ? row_at(J.d_out, rrow) : nullptr;
                gather<8>(dLp, c4, 4, [&](int c) { return (src && rok && c < dL) ? src[c] : 0.f; }, put);
            }
        }
        sync();
        HEAD_PHASE_MARK(3);
        for (int l = L - 1; l >= 0; --l) {
            const int in = m.dims[l], out = m.dims[l + 1], inp = (in + 3) & ~3, outp = (out + 3) & ~3;
            const int ioff = J.act_off[l];
            const float* W = lds + J.w_lds[l];
            const int wld = J.w_ld[l];
            if (per_layer && (l > 0 || J.need_din)) stage_layers<16>(J, lds, l, l + 1);      // read after the barrier below
            HEAD_PHASE_MARK(4);
            // dW^T[o][i] = sum_rows delta[row][o] act[row][i]   (M = outputs, N = inputs, K = the tile's 16 rows: the tile's columns
            // run along the contiguous dimension of torch's [out][in] layout, so a wave's stores are 64-byte runs)
            float* gW = slab + J.w_off[l];
            // blocks (ot, it) of 16 outputs x 32 inputs, dealt round-robin to the waves that share the tile
            const int nit = (in + 31) >> 5, nblocks = ((out + 15) >> 4) * nit;
            for (int blk = wv, ot = 0, itp = wv; blk < nblocks; blk += WV, itp += WV) {
                    while (itp >= nit) { itp -= nit; ++ot; }
                    const int it = 2 * itp;
                    f32x4 acc[1][2];
                    clear<1, 2>(acc);
                    if (!first) {                    // later tiles of the wave: the MFMAs accumulate on top of the slab's values
                        const int kq = lane >> 4;
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int o = ot * 16 + 4 * kq + r, i = (it + nt) * 16 + l16;
                                if (i < in && o < out) acc[0][nt][r] = gW[(size_t)o * in + i];
                            }
                    }
                    mm<1, 2, 4, true>(acc, 4,
                                [&](int mo, int k) { return dcur[k * dld + min(ot * 16 + mo, outp - 1)]; },
                                [&](int k, int c) { return acts[k * ald + ioff + min(it * 16 + c, inp - 1)]; });
                    each<1, 2>(acc, [&](int mo, int c, float v, int, int, int) {
                        const int o = ot * 16 + mo, i = it * 16 + c;
                        if (i < in && o < out) gW[(size_t)o * in + i] = v;
                    });
                }
            float* gb = slab + J.b_off[l];
            for (int c = lane + 64 * wv; c < out; c += 64 * WV) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += dcur[r * dld + c];
                gb[c] = first ? s : gb[c] + s;
            }
            HEAD_PHASE_MARK(5);
            if (l > 0 || J.need_din) {
                // delta_in[row][i] = sum_o delta[row][o] W[i][o], through the ReLU that produced input i (every layer below the top
                // one has it; the MLP's own input has none)
                if (per_layer) lds_barrier();
                const bool mask = l > 0;
                product(in, outp >> 2,
                        [&](int row, int k) { return dcur[row * dld + k]; },
                        [&](int k, int col) { return W[min(col, in - 1) * wld + min(k, out - 1)]; },
                        [&](int) { return zero4(); },
                        [&](int row, int i, float v) {
                            const float ai = acts[row * ald + ioff + min(i, inp - 1)];       // read unguarded, selected below
                            if (i < inp) dnxt[row * dld + i] = (i < in && !(mask && !(ai > 0.f))) ? v : 0.f;
                        });
            }
            sync();
            HEAD_PHASE_MARK(6);
            float* tmp = dcur;
            dcur = dnxt;
            dnxt = tmp;
        }
        if (J.need_din && wv == 0) {
            float* dst = row_at(J.d_in, rrow);
            if (first && (pre_din || (!J.din_add && d0 <= 4 * kRowsPre))) {
#pragma unroll
                for (int u = 0; u < kRowsPre; ++u) {
                    const int c = c4 + 4 * u;
                    if (rok && c < d0) dst[c] = (pre_din ? dinv[u] : 0.f) + dcur[rr * dld + c];
                }
            } else {
                gather<8>(d0, c4, 4, [&](int c) { return (J.din_add && rok) ? dst[c] : 0.f; },
                          [&](int c, float v) { if (rok) dst[c] = v + dcur[rr * dld + c]; });
            }
        }
        sync();
        HEAD_PHASE_MARK(7);
    }
    HEAD_PHASE_FLUSH();
}

// ------------------------------------------------------------------------------------------------
// rows of a wide MLP, a workgroup per 16-row tile, the weights straight from L2 (round 6)
// ------------------------------------------------------------------------------------------------
// The value head's rows at the reference's batch (100 scenes = 7 tiles) are latency, not work: with one tile per workgroup every
// weight is used ONCE per product, so staging the matrices in LDS (mlp_rows_kernel) only adds a round trip and an LDS read per
// MFMA operand -- 35.7 us for 0.6 us of MFMA time in round 5 (profiles/r06_head_rows.txt).  Here the B operands of every product come from
// global memory as the loads of a whole column tile in flight at once, LDS holds the tile's activations and deltas only (so any MLP
// of the ABI fits: path G's 150-100-100-1 head needs no per-layer staging), and the k slots of the MFMA are permuted -- slot
// (g, kk, u) <-> k = 16 g + 4 kk + u -- so that a lane's four k of a group are adjacent: one b128 read of its A row, and for the
// transposed products (delta_in: W[i][o] along o) one dwordx4 load per group.  Same arithmetic as mlp_rows_kernel up to the order
// of the k sum inside a product.  23.0 us against 30.3 for the batch-100 value head (7 workgroups); what remains is seven dependent
// products of ~2 us each -- one compute unit pulling a 40 KB matrix out of L2 per product -- and ~5 us of launch, row traffic and
// barriers (ablations in DESIGN §9).
constexpr int kDirectGroups = 8;       // 16-k groups of operands requested per batch (128 k: 32 + 32 registers)

// acc += A[16][16 K16] B[16 K16][16]: `arow` = this lane's A row (LDS, 16-byte aligned groups, zeros where k is past the end),
// fb(g, kk) = the four B values of k = 16 g + 4 kk .. + 3 for this lane's column
template <class FB>
__device__ __forceinline__ void direct_mm(f32x4& acc, const float* arow, int K16, FB fb) {
    const int kk = (threadIdx.x & 63) >> 4;
    for (int g0 = 0; g0 < K16; g0 += kDirectGroups) {
        f32x4 bv[kDirectGroups], av[kDirectGroups];
#pragma unroll
        for (int g = 0; g < kDirectGroups; ++g) bv[g] = fb(min(g0 + g, K16 - 1), kk);       // past the end: the last group again
#pragma unroll
        for (int g = 0; g < kDirectGroups; ++g) av[g] = *reinterpret_cast<const f32x4*>(arow + 16 * min(g0 + g, K16 - 1) + 4 * kk);
        __builtin_amdgcn_sched_barrier(0);             // every operand requested before the first MFMA
#pragma unroll
        for (int g = 0; g < kDirectGroups; ++g)
            if (g0 + g < K16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = mfma4(av[g][u], bv[g][u], acc);
            }
    }
}

__global__ __launch_bounds__(kCoopWaves * 64) void head_rows_kernel(const RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l16 = lane & 15, kq = lane >> 4;
    const int ji = (a.n_jobs > 1 && (int)blockIdx.x >= a.job[1].wg_begin) ? 1 : 0;
    const RowsJob& J = a.job[ji];
    HEAD_PHASE_START();
    const int w = (int)blockIdx.x - J.wg_begin;
    if (w >= J.n_waves) return;
    const RglMlp& m = J.m;
    const int L = m.n_layers, ald = J.act_ld, dld = J.d_ld;
    float* acts = lds;                         // [16][ald]: every layer's activations, each in a region of a multiple of 16 columns
    float* dcur = acts + 16 * ald;             // [16][dld] twice: the deltas of the layer at hand and of the one below
    float* dnxt = dcur + 16 * dld;
    float* slab = J.slabs + (size_t)w * J.n_params;
    const int rr = lane >> 2, c4 = lane & 3;   // element-wise passes: lane -> (row rr, columns c4, c4 + 4, ..)
    const int d0 = m.dims[0], dL = m.dims[L];
    const bool relu_top = m.last_relu != 0;
    bool first = true;
    for (int t = w; t < J.n_tiles; t += J.n_waves, first = false) {
        const int r0 = t * 16;
        const bool rok = r0 + rr < J.n_rows;
        const int rrow = rok ? r0 + rr : J.n_rows - 1;
        {   // input rows: every column of the region written (zeros beyond the row's end and beyond the last row)
            const float* src = row_at(J.in, rrow);
            const int d16 = (d0 + 15) & ~15;
            for (int c0 = c4; c0 < d16; c0 += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[min(c0 + 4 * u, d0 - 1)];          // unguarded loads, guarded values
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c0 + 4 * u < d16)        // a factor, not a select: a select around a load becomes a branch with a wait inside
                        acts[rr * ald + c0 + 4 * u] = v[u] * ((rok && c0 + 4 * u < d0) ? 1.f : 0.f);
            }
        }
        lds_barrier();
        HEAD_PHASE_MARK(1);
        for (int l = 0; l < L; ++l) {
            const int in = m.dims[l], out = m.dims[l + 1], K16 = (in + 15) >> 4;
            const int ioff = J.act_off[l], ooff = J.act_off[l + 1];
            const bool relu = (l != L - 1) || relu_top;
            const float* __restrict__ W = m.weight[l];
            const float* __restrict__ B = m.bias[l];
            {
                for (int jt = wv; jt * 16 < out; jt += kCoopWaves) {
                    const int col = jt * 16 + l16, colc = min(col, out - 1);
                    const float bv = B[colc];
                    f32x4 acc = f32x4{bv, bv, bv, bv};
                    // Addresses: a wave-uniform base (row 16 g + u of the matrix: scalar arithmetic) plus ONE 32-bit offset per lane
                    // that serves every load of the tile -- per-load 64-bit multiplies and clamps cost more VALU time than the
                    // product's MFMAs.  Only the last group of a ragged K clamps its rows (their A values are zeros).
                    const unsigned lane_off = (unsigned)(4 * kq * out + colc);
                    direct_mm(acc, acts + l16 * ald + ioff, K16, [&](int g, int kk) {
                        if (16 * g + 15 < in) {
                            const float* __restrict__ Wg = W + (size_t)(16 * g) * (size_t)out;
                            return f32x4{Wg[lane_off], (Wg + out)[lane_off], (Wg + 2 * out)[lane_off], (Wg + 3 * out)[lane_off]};
                        }
                        const int k = 16 * g + 4 * kk;
                        const float* __restrict__ Wc = W + colc;
                        return f32x4{Wc[(size_t)min(k, in - 1) * out], Wc[(size_t)min(k + 1, in - 1) * out],
                                     Wc[(size_t)min(k + 2, in - 1) * out], Wc[(size_t)min(k + 3, in - 1) * out]};
                    });
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acts[(4 * kq + r) * ald + ooff + col] = col < out ? (relu ? fmaxf(acc[r], 0.f) : acc[r]) : 0.f;
                }
            }
            lds_barrier();
        }
        HEAD_PHASE_MARK(2);
        if (!a.backward) {
            const int ooff = J.act_off[L];
            float* dst = row_at(J.out, rrow);
            if (rok && wv == 0)
                for (int c = c4; c < dL; c += 4) dst[c] = acts[rr * ald + ooff + c];
            lds_barrier();
            continue;
        }
        {   // upstream gradient of the tile's rows, through the top layer's ReLU if it has one; zeros up to a multiple of 16 columns
            const int ooff = J.act_off[L], d16 = (dL + 15) & ~15;
            const float* src = J.d_out.p ? row_at(J.d_out, rrow) : row_at(J.in, rrow);
            const bool have = J.d_out.p != nullptr;
            for (int c0 = c4; c0 < d16; c0 += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[have ? min(c0 + 4 * u, dL - 1) : 0];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + 4 * u;
                    if (c < d16) {
                        const float ac = acts[rr * ald + ooff + c];
                        dcur[rr * dld + c] = v[u] * ((have && rok && c < dL && !(relu_top && !(ac > 0.f))) ? 1.f : 0.f);
                    }
                }
            }
        }
        lds_barrier();
        HEAD_PHASE_MARK(3);
        for (int l = L - 1; l >= 0; --l) {
            const int in = m.dims[l], out = m.dims[l + 1], in16 = (in + 15) & ~15, out16 = (out + 15) & ~15;
            const int ioff = J.act_off[l];
            const float* __restrict__ W = m.weight[l];
            // dW^T[o][i] = sum_rows delta[row][o] act[row][i]   (M = outputs, N = inputs, K = the tile's 16 rows: the tile's columns
            // run along the contiguous dimension of torch's [out][in] layout, so a wave's stores are 64-byte runs); blocks (ot, it) of
            // 16 outputs x 32 inputs, dealt round-robin to the waves
            float* gW = slab + J.w_off[l];
            const int nit = (in + 31) >> 5, nblocks = (out16 >> 4) * nit;
            for (int blk = wv, ot = 0, itp = wv; blk < nblocks; blk += kCoopWaves, itp += kCoopWaves) {
                while (itp >= nit) { itp -= nit; ++ot; }
                const int it = 2 * itp;
                f32x4 acc[1][2];
                clear<1, 2>(acc);
                if (!first) {                    // later tiles of the workgroup: the MFMAs accumulate on top of the slab's values
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int o = ot * 16 + 4 * kq + r, i = (it + nt) * 16 + l16;
                            if (i < in && o < out) acc[0][nt][r] = gW[(size_t)o * in + i];
                        }
                }
                mm<1, 2, 4, true>(acc, 4,
                                  [&](int mo, int k) { return dcur[k * dld + ot * 16 + mo]; },
                                  [&](int k, int c) { return acts[k * ald + ioff + min(it * 16 + c, in16 - 1)]; });
                each<1, 2>(acc, [&](int mo, int c, float v, int, int, int) {
                    const int o = ot * 16 + mo, i = it * 16 + c;
                    if (i < in && o < out) gW[(size_t)o * in + i] = v;
                });
            }
            float* gb = slab + J.b_off[l];
            for (int c = lane + 64 * wv; c < out; c += 64 * kCoopWaves) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += dcur[r * dld + c];
                gb[c] = first ? s : gb[c] + s;
            }
            HEAD_PHASE_MARK(5);
            if (l > 0 || J.need_din) {
                // delta_in[row][i] = sum_o delta[row][o] W[i][o], through the ReLU that produced input i (every layer below the top
                // one has it; the MLP's own input has none).  W[i][.] is contiguous along the sum: a dwordx4 per group where the
                // row length allows the alignment
                const bool mask = l > 0, vec = (out & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
                const int K16 = out16 >> 4;
                for (int jt = wv; jt * 16 < in16; jt += kCoopWaves) {
                    const int i = jt * 16 + l16, ic = min(i, in - 1);
                    const float* __restrict__ Wi = W + (size_t)ic * out;
                    f32x4 acc = zero4();
                    const unsigned lane_off = (unsigned)(ic * out + 4 * kq);
                    if (vec)
                        direct_mm(acc, dcur + l16 * dld, K16, [&](int g, int kk) {
                            if (16 * g + 15 < out)           // uniform base + the lane's offset, as in the forward products
                                return *reinterpret_cast<const f32x4*>(W + 16 * g + lane_off);
                            // a quad past the row's end: the last one again (its delta values are zeros)
                            return *reinterpret_cast<const f32x4*>(Wi + min(16 * g + 4 * kk, out - 4));
                        });
                    else
                        direct_mm(acc, dcur + l16 * dld, K16, [&](int g, int kk) {
                            const int k = 16 * g + 4 * kk;
                            return f32x4{Wi[min(k, out - 1)], Wi[min(k + 1, out - 1)], Wi[min(k + 2, out - 1)], Wi[min(k + 3, out - 1)]};
                        });
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * kq + r;
                        const float ai = acts[row * ald + ioff + i];                   // i < in16: inside the region
                        dnxt[row * dld + i] = (i < in && !(mask && !(ai > 0.f))) ? acc[r] : 0.f;
                    }
                }
            }
            lds_barrier();
            HEAD_PHASE_MARK(6);
            float* tmp = dcur;
            dcur = dnxt;
            dnxt = tmp;
        }
        if (J.need_din && wv == 0) {
            float* dst = row_at(J.d_in, rrow);
            for (int c0 = c4; c0 < d0; c0 += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = J.din_add ? dst[min(c0 + 4 * u, d0 - 1)] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (rok && c0 + 4 * u < d0) dst[c0 + 4 * u] = v[u] + dcur[rr * dld + c0 + 4 * u];
            }
        }
        lds_barrier();
        HEAD_PHASE_MARK(7);
    }
    HEAD_PHASE_FLUSH();
}

// The shipped narrow MLPs -- in -> 64 -> out with in, out <= 32: w_r, w_h (9 | 5 | 6 | 7 -> 64 -> 32), the motion head
// (32 -> 64 -> 5) -- carry nearly all rows of a batch (every node of every scene).  Same organisation as mlp_rows_kernel, but the
// shapes are template parameters (T0 / T2 = 16-wide tiles of the input / output): every k loop is unrolled with its operand loads
// batched, and the weight gradients of BOTH layers stay in MFMA accumulators over all tiles of the wave -- (4 T0 + 4 T2) tiles, 48
// registers -- so a wave touches its slab once, at the end.
constexpr int kNarrowWaves = 4;       // waves per workgroup of mlp2_rows_kernel: three workgroups fit a CU (40 KB of LDS each, <= 170 VGPRs)
template <int T0, int T2>
__global__ __launch_bounds__(kNarrowWaves * 64, 3) void mlp2_rows_kernel(const RowsArgs a) {
    constexpr int HT = 4, IN_LD = T0 * 16 + 2, HLD = 66, OLD = T2 * 16 + 2;
    constexpr int kWaveFloats = 16 * (IN_LD + HLD + OLD);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15;
    const int ji = (a.n_jobs > 1 && (int)blockIdx.x >= a.job[1].wg_begin) ? 1 : 0;
    const RowsJob& J = a.job[ji];
    ROWS2_PHASE_START();
    const int w = ((int)blockIdx.x - J.wg_begin) * J.waves_per_wg + wave;
    const bool on = wave < J.waves_per_wg && w < J.n_waves;
    const RglMlp& m = J.m;
    const int in = m.dims[0], out = m.dims[2];
    const float* W0 = lds + J.w_lds[0];
    const float* W1 = lds + J.w_lds[1];
    const float* b0 = lds + J.b_lds[0];
    const float* b1 = lds + J.b_lds[1];
    constexpr int wld0 = 65, wld1 = T2 * 16 + 1;      // compile-time row strides of the two weight matrices in LDS (plan_rows_job)
    float* xin = lds + J.weight_floats + wave * kWaveFloats;      // [16][IN_LD]   input rows, zero beyond `in`; later the input deltas
    float* hid = xin + 16 * IN_LD;                                // [16][HLD]     hidden activations
    float* d1 = hid + 16 * HLD;                                   // [16][OLD]     outputs, then their deltas (zero beyond `out`)
    float* d0 = hid;                                              // the hidden deltas replace the hidden activations in place (each
                                                                  // element is read -- its ReLU mask -- and written by the same lane)
    const bool last_relu = m.last_relu != 0;
    f32x4 gW0[HT][T0], gW1[T2][HT];
    clear<HT, T0>(gW0);
    clear<T2, HT>(gW1);
    float gb0 = 0.f, gb1 = 0.f;
    const int rr = lane >> 2, c4 = lane & 3;
    // the rows of a tile (inputs, and upstream deltas in a backward launch) are fetched one tile ahead: lane -> (row lane / 4,
    // columns lane % 4 + 4 u); the first tile's before the weights, so that the two latencies overlap
    float vin[T0 * 4], vd[T2 * 4];
    auto fetch = [&](int t) {
        const bool ok = t < J.n_tiles && t * 16 + rr < J.n_rows;
        const int row = ok ? t * 16 + rr : 0;
        const float* src = row_at(J.in, row);
#pragma unroll
        for (int u = 0; u < T0 * 4; ++u) vin[u] = (ok && c4 + 4 * u < in) ? src[c4 + 4 * u] : 0.f;
        if (a.backward) {
            const float* dsrc = J.d_out.p ? row_at(J.d_out, row) : nullptr;
#pragma unroll
            for (int u = 0; u < T2 * 4; ++u) vd[u] = (dsrc && ok && c4 + 4 * u < out) ? dsrc[c4 + 4 * u] : 0.f;
        }
    };
    fetch(on ? w : J.n_tiles);
    {   // both layers' weights and biases -> LDS, padded with zeros to whole tiles (rows of W0 beyond `in`, columns of W1 / b1 beyond
        // `out`): the k loops then need neither guards nor clamps, and every LDS address is a base register plus a constant
        constexpr int n0 = T0 * 16 * 64, n1 = 64 * T2 * 16, nb = n0 + n1, total = nb + 64 + T2 * 16;
        const float* __restrict__ g0 = m.weight[0];
        const float* __restrict__ g1 = m.weight[1];
        const float* __restrict__ gb0p = m.bias[0];
        const float* __restrict__ gb1p = m.bias[1];
        gather<7>(total, threadIdx.x, kNarrowWaves * 64,
                  [&](int idx) {
                      if (idx < n0) return (idx >> 6) < in ? g0[idx] : 0.f;
                      if (idx < nb) { const int j = idx - n0, k = j / (T2 * 16), c = j % (T2 * 16); return c < out ? g1[k * out + c] : 0.f; }
                      if (idx < nb + 64) return gb0p[idx - nb];
                      return idx - nb - 64 < out ? gb1p[idx - nb - 64] : 0.f;
                  },
                  [&](int idx, float v) {
                      if (idx < n0) lds[J.w_lds[0] + (idx >> 6) * wld0 + (idx & 63)] = v;
                      else if (idx < nb) { const int j = idx - n0; lds[J.w_lds[1] + (j / (T2 * 16)) * wld1 + j % (T2 * 16)] = v; }
                      else if (idx < nb + 64) lds[J.b_lds[0] + idx - nb] = v;
                      else lds[J.b_lds[1] + idx - nb - 64] = v;
                  });
    }
    __syncthreads();
    ROWS2_PHASE_MARK(0);
    if (!on) return;
    for (int t = w; t < J.n_tiles; t += J.n_waves) {
        const int r0 = t * 16;
        const bool rok = r0 + rr < J.n_rows;
        const int rrow = rok ? r0 + rr : J.n_rows - 1;
        float vdc[T2 * 4];
#pragma unroll
        for (int u = 0; u < T0 * 4; ++u) xin[rr * IN_LD + c4 + 4 * u] = vin[u];
#pragma unroll
        for (int u = 0; u < T2 * 4; ++u) vdc[u] = vd[u];
        fetch(t + J.n_waves);
        wave_sync();
        ROWS2_PHASE_MARK(1);
        {   // hidden = relu(x W0 + b0)
            f32x4 acc[1][HT];
#pragma unroll
            for (int nt = 0; nt < HT; ++nt) {
                const float bv = b0[nt * 16 + l16];
                acc[0][nt] = f32x4{bv, bv, bv, bv};
            }
            mm<1, HT, 4>(acc, T0 * 4, [&](int row, int k) { return xin[row * IN_LD + k]; },
                              [&](int k, int c) { return W0[k * wld0 + c]; });
            each<1, HT>(acc, [&](int row, int c, float v, int, int, int) { hid[row * HLD + c] = fmaxf(v, 0.f); });
        }
        wave_sync();
        ROWS2_PHASE_MARK(2);
        {   // y = hidden W1 + b1 (ReLU when the MLP ends with one)
            f32x4 acc[1][T2];
#pragma unroll
            for (int nt = 0; nt < T2; ++nt) {
                const int col = nt * 16 + l16;
                const float bv = b1[col];
                acc[0][nt] = f32x4{bv, bv, bv, bv};
            }
            mm<1, T2, 4>(acc, 16, [&](int row, int k) { return hid[row * HLD + k]; },
                         [&](int k, int c) { return W1[k * wld1 + c]; });
            each<1, T2>(acc, [&](int row, int c, float v, int, int, int) { d1[row * OLD + c] = last_relu ? fmaxf(v, 0.f) : v; });
        }
        wave_sync();
        ROWS2_PHASE_MARK(3);
        if (!a.backward) {
            float* dst = row_at(J.out, rrow);
            if (rok)
                for (int c = c4; c < out; c += 4) dst[c] = d1[rr * OLD + c];
            wave_sync();
            continue;
        }
        {   // upstream deltas through the last ReLU; zero in the padding columns
#pragma unroll
            for (int u = 0; u < T2 * 4; ++u) {
                const int c = c4 + 4 * u;
                d1[rr * OLD + c] = (last_relu && !(d1[rr * OLD + c] > 0.f)) ? 0.f : vdc[u];
            }
        }
        wave_sync();
        ROWS2_PHASE_MARK(4);
        // dW1^T[o][h] += sum_rows delta1[row][o] hidden[row][h]
        mm<T2, HT, 4>(gW1, 4, [&](int mo, int k) { return d1[k * OLD + mo]; }, [&](int k, int c) { return hid[k * HLD + c]; });
        if (lane < T2 * 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gb1 += d1[r * OLD + lane];
        }
        {   // delta0 = (delta1 W1^T) where the hidden ReLU is open
            f32x4 acc[1][HT];
            clear<1, HT>(acc);
            mm<1, HT, 4>(acc, T2 * 4, [&](int row, int k) { return d1[row * OLD + k]; },
                              [&](int k, int c) { return W1[c * wld1 + k]; });
            each<1, HT>(acc, [&](int row, int c, float v, int, int, int) { d0[row * HLD + c] = hid[row * HLD + c] > 0.f ? v : 0.f; });
        }
        wave_sync();
        ROWS2_PHASE_MARK(5);
        // dW0^T[h][i] += sum_rows delta0[row][h] x[row][i]
        mm<HT, T0, 4>(gW0, 4, [&](int mh, int k) { return d0[k * HLD + mh]; }, [&](int k, int c) { return xin[k * IN_LD + c]; });
#pragma unroll
        for (int r = 0; r < 16; ++r) gb0 += d0[r * HLD + lane];
        ROWS2_PHASE_MARK(6);
        if (J.need_din) {
            f32x4 acc[1][T0];
            clear<1, T0>(acc);
            mm<1, T0, 4>(acc, 16, [&](int row, int k) { return d0[row * HLD + k]; },
                         [&](int k, int c) { return W0[c * wld0 + k]; });
            wave_sync();                          // every lane has read x for dW0
            each<1, T0>(acc, [&](int row, int c, float v, int, int, int) { xin[row * IN_LD + c] = v; });
            wave_sync();
            float* dst = row_at(J.d_in, rrow);
            float v[T0 * 4];
#pragma unroll
            for (int u = 0; u < T0 * 4; ++u) v[u] = (J.din_add && rok && c4 + 4 * u < in) ? dst[c4 + 4 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < T0 * 4; ++u) {
                const int c = c4 + 4 * u;
                if (rok && c < in) dst[c] = v[u] + xin[rr * IN_LD + c];
            }
        }
        wave_sync();
        ROWS2_PHASE_MARK(7);
    }
    ROWS2_PHASE_FLUSH();
    if (a.backward) {
        float* slab = J.slabs + (size_t)w * J.n_params;
        each<HT, T0>(gW0, [&](int h, int i, float v, int, int, int) { if (i < in) slab[J.w_off[0] + h * in + i] = v; });
        each<T2, HT>(gW1, [&](int o, int h, float v, int, int, int) { if (o < out) slab[J.w_off[1] + o * 64 + h] = v; });
        slab[J.b_off[0] + lane] = gb0;
        if (lane < out) slab[J.b_off[1] + lane] = gb1;
    }
}

// ------------------------------------------------------------------------------------------------
// the graph block: one workgroup per scene
// ------------------------------------------------------------------------------------------------
struct GraphArgs {
    const float* Xr;             // [S][X]            embedded robot rows
    const float* Xh;             // [S / spc][H][X]   embedded human rows; the spc sibling scenes of a rollout share their crowd
    const float* dHL;            // [S][N][X]   (backward)
    float* HL;                   // [S][N][X], or [S][X] (row 0 only: hl_row0)   (forward)
    float* dXr;                  // [S][X]      (backward)
    float* dXh;                  // [S][H][X]   (backward)
    const float* w_a;            // [X][X] or null (gaussian: S = X X^T)
    const float* Ws[3];
    float* slabs;                // [workgroups][(has w_a + L) * X * X]
    int S, N, skip, spc, hl_row0;
    int norm;                    // row normalisation of the similarity block (graph_model.py:63-93): 0 softmax(S) (embedded_gaussian,
                                 // gaussian); 1 squared: S^2 / sum_row S^2 (:86-89); 2 equal_attention: 1 / N (:90-91); 3 diagonal: I (:92-93);
                                 // 4 cosine: C_ij = S_ij / (m_i m_j), m_i = |row i of S| (:70-74); 5 cosine_softmax: softmax(C) (:75-79)
    // floats between the robot rows of consecutive scenes / the human rows of consecutive crowds, in Xr / Xh and in dXr / dXh: X and
    // (N - 1) X for the compact arrays above, N X for both when a scene's rows are one [N][X] block (Xh = Xr + X)
    int xr_stride, xh_stride;
    int lw;                      // layerwise graph (graph_model.py:118-122): an adjacency per layer, A_l = softmax(S(H_l))
};

// sum / max over the 16 lanes of a DPP row, every lane gets it
__device__ __forceinline__ float dpp_f(float x, int ctrl) {
    switch (ctrl) {
        case 0: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
        case 1: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
        case 2: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x141, 0xf, 0xf, true));    // row_half_mirror
        default: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x140, 0xf, 0xf, true));   // row_mirror
    }
}
__device__ __forceinline__ float row16_sum(float x) {
#pragma unroll
    for (int c = 0; c < 4; ++c) x += dpp_f(x, c);
    return x;
}
__device__ __forceinline__ float row16_maxf(float x) {
#pragma unroll
    for (int c = 0; c < 4; ++c) x = fmaxf(x, dpp_f(x, c));
    return x;
}

// LDS of a workgroup: weights  Wa | W_0 .. W_{L-1}  ([X][FLD] each, FLD = X + 2: row-indexed A-operand reads hit 32 distinct banks),
// then the scene's   X, dH, dZ, dT, T_0 .. T_{L-1}, H_1 .. H_{L-1}   ([NP][FLD], NP = N rounded up to 4: the padding rows stay zero,
// so node-indexed k loops need no guards)   and   A, dA   ([NP][ALD]).  A forward-only launch has no dH, dZ, dA.
template <int NT, int XT>
struct GraphLds {
    static constexpr int XW = XT * 16, FLD = XW + 2, ALD = NT * 16 + 2;
    static __host__ __device__ int scene_floats(int N, int L, bool bwd, bool lw = false) {     // lw: an adjacency per layer
        const int NP = (N + 3) & ~3;
        return ((bwd ? 4 : 2) + L + (L - 1)) * NP * FLD + ((lw ? L : 1) + (bwd ? 1 : 0)) * NP * ALD;
    }
    static __host__ __device__ int weight_floats(int L) { return (1 + L) * XW * FLD; }
};

// A scene is a chain of ~20 small products (8 MFMAs per 16 x 16 tile each), every one needing the whole result of the one before:
// a wave per scene spends its time in LDS round trips.  So the 2 NT waves of a workgroup share ONE scene: each product's output tiles
// are dealt to the waves -- an [N][X] result has NT x XT tiles (X = 16 XT features: 32 shipped, 64 supported), a 1 x XT/2 block per
// wave; [N][N] results NT x NT; the [X][X] weight gradients XT x XT over four waves -- with a workgroup barrier between phases, and
// several workgroups per CU overlap each other's barriers.  The weight gradients stay in the accumulators of the waves that own
// their tiles over all scenes of the workgroup (one slab per workgroup).
// COS: the build with the cosine family's passes (norm 4 / 5).  A build of its own, so that the shipped similarity functions keep
// the register allocation they had without them (the passes cost the L = 2 backward 36 more bytes of scratch per lane otherwise).
// LW (round 6): layerwise graphs (graph_model.py:118-122) -- the adjacency is recomputed from every layer's input, A_l =
// softmax(H_l Wa H_l^T): L adjacency buffers in LDS, and the backward pass goes through the similarity block inside the layer loop
// (dA_l, the softmax, dG_l = dS_l H_l, dH_l += dS_l^T G_l + dG_l Wa^T, dWa += H_l^T dG_l) before it forms the next layer's dZ.
// The softmax normalisations (embedded_gaussian, gaussian) and the squared one; equal_attention / diagonal adjacencies are
// constants, so their layerwise graphs ARE the one-adjacency form; the cosine family and the pair-MLP similarity of a layerwise
// graph stay on the per-scene kernel.
// (two workgroups per CU at least for the layerwise form: its backward carries the similarity block's accumulators through the
// layer loop and spills 50-88 registers under the four-workgroup budget of the one-adjacency form)
template <int NT, int XT, int L, bool BWD, bool COS, bool LW = false>
__global__ __launch_bounds__(NT * 128, LW ? 2 : ((NT == 2 && XT == 2) ? 4 : (NT == 1 ? 3 : 2))) void graph_kernel(const GraphArgs a) {
    static_assert(!(LW && COS), "layerwise graphs: the softmax and squared normalisations only");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using Lds = GraphLds<NT, XT>;
    constexpr int XW = Lds::XW, FLD = Lds::FLD, ALD = Lds::ALD;
    constexpr int W = 2 * NT;                        // waves
    constexpr int XTW = XT / 2;                      // column tiles of an [N][X] result per wave
    constexpr int NTW = NT >= 2 ? NT / 2 : 1;        // column tiles of an [N][N] result per wave
    constexpr int GM = XT / 2, GN = W >= 4 ? XT / 2 : XT;        // this wave's block of an [X][X] result, in tiles
    constexpr int PF = XW / 8;                       // elements per thread of a scene's [NP][X] rows (NP X / 64 W <= X / 8)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, kq = lane >> 4;
    const int N = a.N, NP = (N + 3) & ~3, NK = NP >> 2, last = NP - 1;
    const bool embedded = a.w_a != nullptr;
    float* Wa = lds;
    float* Wl = lds + XW * FLD;
    for (int idx = threadIdx.x; idx < XW * XW; idx += W * 64) {
        const int k = idx / XW, c = idx % XW;
        Wa[k * FLD + c] = embedded ? a.w_a[idx] : 0.f;
#pragma unroll
        for (int l = 0; l < L; ++l) Wl[l * XW * FLD + k * FLD + c] = a.Ws[l][idx];
    }
    float* base = lds + Lds::weight_floats(L);
    const int U = NP * FLD;
    float* X = base;
    float* dT = X + U;                 // G = X Wa lives here in the forward sweep
    float* T = dT + U;                 // [L]
    float* Hs = T + L * U;             // [L - 1]: H_1 ..
    float* A = Hs + (L - 1) * U;       // LW: A_0 .. A_{L-1}
    float* dA = A + (LW ? L : 1) * NP * ALD;          // backward only from here
    float* dH = dA + NP * ALD;
    float* dZ = dH + U;
    auto Hl = [&](int l) { return l == 0 ? X : Hs + (l - 1) * U; };
    // this wave's block of an [N][X] result, of an [N][N] result, of an [X][X] result
    const int fm = (wave >> 1) * 16, fn = (wave & 1) * XTW * 16;
    const int am = fm, an = (wave & 1) * NTW * 16;
    const bool a_on = (wave & 1) * NTW < NT;
    const int gm = (W >= 4 ? (wave >> 1) : wave) * GM * 16, gn = W >= 4 ? (wave & 1) * GN * 16 : 0;
    const bool g_on = W >= 4 ? wave < 4 : true;
    const int frow = fm + 4 * kq;      // element r of this lane in a tile of its [N][X] block: row frow + r, column fn + 16 nt + l16

    f32x4 gWa[GM][GN], gW[L][GM][GN];
    clear<GM, GN>(gWa);
#pragma unroll
    for (int l = 0; l < L; ++l) clear<GM, GN>(gW[l]);
    // X (and the upstream gradient) of a scene are fetched into registers one scene ahead
    float xp[PF], dp[PF];
    auto prefetch = [&](int s) {
        const bool ok = s < a.S;
        const int sc = ok ? s : 0;
        const float* xr = a.Xr + (size_t)sc * a.xr_stride;
        const float* xh = a.Xh + (size_t)(sc / a.spc) * a.xh_stride - XW;       // row i >= 1 at xh + i * XW
        const float* dg = BWD ? a.dHL + (size_t)sc * N * XW : nullptr;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int idx = threadIdx.x + u * W * 64;
            xp[u] = (ok && idx < N * XW) ? (idx < XW ? xr[idx] : xh[idx]) : 0.f;
            if constexpr (BWD) dp[u] = (ok && idx < N * XW) ? dg[idx] : 0.f;
        }
    };
    prefetch(blockIdx.x);
    __syncthreads();

    for (int s = blockIdx.x; s < a.S; s += gridDim.x) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {      // rows N .. NP-1 zero
            const int idx = threadIdx.x + u * W * 64;
            if (idx < NP * XW) {
                X[(idx / XW) * FLD + idx % XW] = xp[u];
                if constexpr (BWD) dH[(idx / XW) * FLD + idx % XW] = dp[u];
            }
        }
        prefetch(s + gridDim.x);
        __syncthreads();
        float* G = dT;
        // out[row][fn ..] of this wave's [N][X] block = v, zero in the padding rows
        auto put = [&](float* out, const f32x4 (&acc)[1][XTW]) {
#pragma unroll
            for (int nt = 0; nt < XTW; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (frow + r < NP) out[(frow + r) * FLD + fn + nt * 16 + l16] = frow + r < N ? acc[0][nt][r] : 0.f;
        };
        auto make_G = [&](const float* Hin) {          // G = H Wa of the rows the similarity is taken over (X; LW: H_l)
            f32x4 acc[1][XTW];
            clear<1, XTW>(acc);
            mm<1, XTW, 8>(acc, XW / 4, [&](int i, int k) { return Hin[min(fm + i, last) * FLD + k]; },
                          [&](int k, int j) { return Wa[k * FLD + fn + j]; });
            put(G, acc);
        };
        const int norm = a.norm;
        // A_l of a layerwise graph: S = (H_l Wa) H_l^T (gaussian, squared: H_l H_l^T), its row normalisation; ends with a barrier
        auto make_A_lw = [&](const float* Hc, float* Al) {
            if (embedded) {
                make_G(Hc);
                __syncthreads();
            }
            const float* GXl = embedded ? G : Hc;
            if (a_on) {
                f32x4 acc[1][NTW];
                clear<1, NTW>(acc);
                mm<1, NTW, 8>(acc, XW / 4, [&](int i, int k) { return GXl[min(am + i, last) * FLD + k]; },
                              [&](int k, int j) { return Hc[min(an + j, last) * FLD + k]; });
                each<1, NTW>(acc, [&](int i, int j, float v, int, int, int) {
                    const int row = am + i, col = an + j;
                    if (row < NP) Al[row * ALD + col] = (row < N && col < N) ? v : 0.f;
                });
            }
            __syncthreads();
            for (int row = wave * 4 + kq; norm == 1 && row < N; row += W * 4) {       // squared: as in the one-adjacency form below
                float* r = Al + row * ALD;
                float sv[NT], w[NT], sum = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    sv[j] = l16 + 16 * j < N ? r[l16 + 16 * j] : 0.f;
                    w[j] = sv[j] * sv[j];
                    sum += w[j];
                }
                sum = row16_sum(sum);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (l16 + 16 * j < N) r[l16 + 16 * j] = copysignf(w[j] / sum, sv[j]);
                if (l16 == 0) r[NT * 16] = sum;
            }
            for (int row = wave * 4 + kq; norm == 0 && row < N; row += W * 4) {
                float* r = Al + row * ALD;
                float v[NT], mx = -3.4e38f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    v[j] = l16 + 16 * j < N ? r[l16 + 16 * j] : -3.4e38f;
                    mx = fmaxf(mx, v[j]);
                }
                mx = row16_maxf(mx);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    v[j] = l16 + 16 * j < N ? expf(v[j] - mx) : 0.f;
                    sum += v[j];
                }
                sum = row16_sum(sum);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (l16 + 16 * j < N) r[l16 + 16 * j] = v[j] / sum;
            }
            __syncthreads();
        };
        const bool cosine = COS && norm >= 4;
        const bool from_s = norm <= 1 || cosine;        // the adjacency is a function of S (it is a constant otherwise)
        // squared similarity: A is kept SIGNED in LDS -- sign(S_ij) |A_ij| -- and the row's Z_i = sum_j S_ij^2 in the row's padding
        // column, so that the backward pass gets S_ij = sign sqrt(|A_ij| Z_i) back without a buffer of its own; every consumer of
        // A reads it through aval()
        auto aval = [&](float x) { return norm == 1 ? fabsf(x) : x; };
        if (embedded && !LW) {
            make_G(X);
            __syncthreads();
        }
        const float* GX = embedded ? G : X;
        if (!from_s) {
            const float c = norm == 2 ? 1.f / (float)N : 0.f;
            for (int idx = threadIdx.x; idx < NP * ALD; idx += W * 64) {
                const int row = idx / ALD, col = idx - row * ALD;
                A[idx] = (row < N && col < N) ? (norm == 2 ? c : (row == col ? 1.f : 0.f)) : 0.f;
            }
        }
        if (a_on && from_s && !LW) {   // S = G X^T   (graph_model.py:64-69)
            f32x4 acc[1][NTW];
            clear<1, NTW>(acc);
            mm<1, NTW, 8>(acc, XW / 4, [&](int i, int k) { return GX[min(am + i, last) * FLD + k]; },
                          [&](int k, int j) { return X[min(an + j, last) * FLD + k]; });
            each<1, NTW>(acc, [&](int i, int j, float v, int, int, int) {
                const int row = am + i, col = an + j;
                if (row < NP) A[row * ALD + col] = (row < N && col < N) ? v : 0.f;
            });
        }
        __syncthreads();
        // squared (graph_model.py:86-89): w = S^2 over its row sum
        for (int row = wave * 4 + kq; norm == 1 && row < N; row += W * 4) {
            float* r = A + row * ALD;
            float sv[NT], w[NT], sum = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                sv[j] = l16 + 16 * j < N ? r[l16 + 16 * j] : 0.f;
                w[j] = sv[j] * sv[j];
                sum += w[j];
            }
            sum = row16_sum(sum);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (l16 + 16 * j < N) r[l16 + 16 * j] = copysignf(w[j] / sum, sv[j]);
            if (l16 == 0) r[NT * 16] = sum;
        }
        // cosine family (graph_model.py:70-79): the norms of S's ROWS first (kept in the rows' padding column, like the squared sums:
        // the backward pass needs them again), then C_ij = S_ij / (m_i m_j) and, cosine_softmax, its row softmax in the same pass
        if constexpr (COS) {
        for (int row = wave * 4 + kq; cosine && row < N; row += W * 4) {
            float* r = A + row * ALD;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float v = l16 + 16 * j < N ? r[l16 + 16 * j] : 0.f;
                sum = fmaf(v, v, sum);
            }
            sum = row16_sum(sum);
            if (l16 == 0) r[NT * 16] = sqrtf(sum);
        }
        if (cosine) __syncthreads();
        for (int row = wave * 4 + kq; cosine && row < N; row += W * 4) {
            float* r = A + row * ALD;
            const float mi = r[NT * 16];
            float v[NT], mx = -3.4e38f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = l16 + 16 * j;
                v[j] = col < N ? r[col] / (mi * A[col * ALD + NT * 16]) : -3.4e38f;
                mx = fmaxf(mx, v[j]);
            }
            if (norm == 5) {
                mx = row16_maxf(mx);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    v[j] = l16 + 16 * j < N ? expf(v[j] - mx) : 0.f;
                    sum += v[j];
                }
                sum = row16_sum(sum);
#pragma unroll
                for (int j = 0; j < NT; ++j) v[j] /= sum;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (l16 + 16 * j < N) r[l16 + 16 * j] = v[j];
        }
        }
        // row softmax: 16 lanes per row, four rows per wave and pass
        for (int row = wave * 4 + kq; !LW && norm == 0 && row < N; row += W * 4) {
            float* r = A + row * ALD;
            float v[NT], mx = -3.4e38f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                v[j] = l16 + 16 * j < N ? r[l16 + 16 * j] : -3.4e38f;
                mx = fmaxf(mx, v[j]);
            }
            mx = row16_maxf(mx);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                v[j] = l16 + 16 * j < N ? expf(v[j] - mx) : 0.f;
                sum += v[j];
            }
            sum = row16_sum(sum);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (l16 + 16 * j < N) r[l16 + 16 * j] = v[j] / sum;
        }
        __syncthreads();
        unsigned mask[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const float* Hc = Hl(l);
            float* Tl = T + l * U;
            const float* Al = LW ? A + l * NP * ALD : A;
            if constexpr (LW) make_A_lw(Hc, A + l * NP * ALD);
            {   // T_l = A H_l
                f32x4 acc[1][XTW];
                clear<1, XTW>(acc);
                mm<1, XTW, 8>(acc, NK, [&](int i, int k) { return aval(Al[min(fm + i, last) * ALD + k]); },
                              [&](int k, int j) { return Hc[k * FLD + fn + j]; });
                put(Tl, acc);
            }
            __syncthreads();
            {   // H_{l+1} = relu(T_l W_l) (+ H_l)
                f32x4 acc[1][XTW];
                clear<1, XTW>(acc);
                const float* Wc = Wl + l * XW * FLD;
                mm<1, XTW, 8>(acc, XW / 4, [&](int i, int k) { return Tl[min(fm + i, last) * FLD + k]; },
                              [&](int k, int j) { return Wc[k * FLD + fn + j]; });
                unsigned bits = 0;
                const bool keep = l + 1 < L;                    // the next layer's input
                float* Hn = keep ? Hs + l * U : nullptr;
                float* out = (!BWD && !keep) ? (a.hl_row0 ? a.HL + (size_t)s * XW : a.HL + (size_t)s * N * XW) : nullptr;
#pragma unroll
                for (int nt = 0; nt < XTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = frow + r, col = fn + nt * 16 + l16;
                        const float v = acc[0][nt][r];
                        if (v > 0.f) bits |= 1u << (nt * 4 + r);
                        if (row < NP) {
                            const float h = row < N ? fmaxf(v, 0.f) + (a.skip ? Hc[row * FLD + col] : 0.f) : 0.f;
                            if (keep) Hn[row * FLD + col] = h;
                            else if (out && row < (a.hl_row0 ? 1 : N)) out[row * XW + col] = h;
                            // the top layer's dZ = dH_L where its ReLU is open, straight from here
                            if constexpr (BWD)
                                if (!keep) dZ[row * FLD + col] = (v > 0.f && row < N) ? dH[row * FLD + col] : 0.f;
                        }
                    }
                mask[l] = bits;
            }
            __syncthreads();
        }
        if constexpr (BWD) {
            f32x4 dAacc[1][NTW];
            clear<1, NTW>(dAacc);
#pragma unroll
            for (int l = L - 1; l >= 0; --l) {
                const float* Hc = Hl(l);
                const float* Tl = T + l * U;
                const float* Wc = Wl + l * XW * FLD;
                const float* Al = LW ? A + l * NP * ALD : A;
                // dW_l += T_l^T dZ
                if (g_on)
                    mm<GM, GN, 8>(gW[l], NK, [&](int mi, int k) { return Tl[k * FLD + gm + mi]; },
                                  [&](int k, int j) { return dZ[k * FLD + gn + j]; });
                {   // dT = dZ W_l^T
                    f32x4 acc[1][XTW];
                    clear<1, XTW>(acc);
                    mm<1, XTW, 8>(acc, XW / 4, [&](int i, int k) { return dZ[min(fm + i, last) * FLD + k]; },
                                  [&](int k, int j) { return Wc[(fn + j) * FLD + k]; });
                    put(dT, acc);
                }
                __syncthreads();
                // dA += dT H_l^T
                if (a_on)
                    mm<1, NTW, 8>(dAacc, XW / 4, [&](int i, int k) { return dT[min(am + i, last) * FLD + k]; },
                                  [&](int k, int j) { return Hc[min(an + j, last) * FLD + k]; });
                {   // dH_l = A^T dT (+ dH_{l+1} through the skip connection); the next layer's dZ right away
                    f32x4 acc[1][XTW];
                    clear<1, XTW>(acc);
                    mm<1, XTW, 8>(acc, NK, [&](int mi, int k) { return aval(Al[k * ALD + fm + mi]); },
                                  [&](int k, int j) { return dT[k * FLD + fn + j]; });
#pragma unroll
                    for (int nt = 0; nt < XTW; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = frow + r, col = fn + nt * 16 + l16;
                            if (row < N) {
                                const float d = acc[0][nt][r] + (a.skip ? dH[row * FLD + col] : 0.f);
                                dH[row * FLD + col] = d;
                                // (LW: dH_l is not complete yet -- the similarity block of this layer adds to it below)
                                if (!LW && l > 0) dZ[row * FLD + col] = ((mask[l > 0 ? l - 1 : 0] >> (nt * 4 + r)) & 1u) ? d : 0.f;
                            }
                        }
                }
                __syncthreads();
                if constexpr (LW) {
                    // ---- through A_l = softmax(S_l), S_l = G_l H_l^T, G_l = H_l Wa (gaussian: G_l = H_l)
                    if (a_on)
                        each<1, NTW>(dAacc, [&](int i, int j, float v, int, int, int) {
                            const int row = am + i, col = an + j;
                            if (row < NP) dA[row * ALD + col] = (row < N && col < N) ? v : 0.f;
                        });
                    clear<1, NTW>(dAacc);
                    if (embedded) make_G(Hc);            // into dT's buffer: its readers (dA_l, dH_l) are behind the barrier above
                    __syncthreads();
                    for (int row = wave * 4 + kq; norm == 1 && row < N; row += W * 4) {     // squared: dS = 2 S (dA - sum dA A) / Z
                        float* d = dA + row * ALD;
                        const float* p = Al + row * ALD;
                        const float zi = p[NT * 16];
                        float dv[NT], pv[NT], dot = 0.f;
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const bool ok = l16 + 16 * j < N;
                            dv[j] = ok ? d[l16 + 16 * j] : 0.f;
                            pv[j] = ok ? p[l16 + 16 * j] : 0.f;
                            dot = fmaf(dv[j], fabsf(pv[j]), dot);
                        }
                        dot = row16_sum(dot);
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            if (l16 + 16 * j < N) d[l16 + 16 * j] = 2.f * copysignf(sqrtf(fabsf(pv[j]) / zi), pv[j]) * (dv[j] - dot);
                    }
                    for (int row = wave * 4 + kq; norm == 0 && row < N; row += W * 4) {       // softmax: dS_ij = A_ij (dA_ij - sum_k dA_ik A_ik)
                        float* d = dA + row * ALD;
                        const float* p = Al + row * ALD;
                        float dv[NT], pv[NT], dot = 0.f;
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const bool ok = l16 + 16 * j < N;
                            dv[j] = ok ? d[l16 + 16 * j] : 0.f;
                            pv[j] = ok ? p[l16 + 16 * j] : 0.f;
                            dot = fmaf(dv[j], pv[j], dot);
                        }
                        dot = row16_sum(dot);
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            if (l16 + 16 * j < N) d[l16 + 16 * j] = pv[j] * (dv[j] - dot);
                    }
                    __syncthreads();
                    const float* GXl = embedded ? G : Hc;
                    float* dGl = dZ;                      // dZ_l is spent: dW_l and dT have read it
                    f32x4 dxl[1][XTW];
                    {
                        f32x4 acc[1][XTW];
                        clear<1, XTW>(acc);
                        mm<1, XTW, 8>(acc, NK, [&](int i, int k) { return dA[min(fm + i, last) * ALD + k]; },
                                      [&](int k, int j) { return Hc[k * FLD + fn + j]; });
                        put(dGl, acc);
                        clear<1, XTW>(dxl);
                        mm<1, XTW, 8>(dxl, NK, [&](int mi, int k) { return dA[k * ALD + fm + mi]; },
                                      [&](int k, int j) { return GXl[k * FLD + fn + j]; });
                        if (!embedded) {
#pragma unroll
                            for (int nt = 0; nt < XTW; ++nt)
#pragma unroll
                                for (int r = 0; r < 4; ++r) dxl[0][nt][r] += acc[0][nt][r];
                        }
                    }
                    __syncthreads();
                    if (embedded) {
                        if (g_on)
                            mm<GM, GN, 8>(gWa, NK, [&](int mi, int k) { return Hc[k * FLD + gm + mi]; },
                                          [&](int k, int j) { return dGl[k * FLD + gn + j]; });
                        mm<1, XTW, 8>(dxl, XW / 4, [&](int i, int k) { return dGl[min(fm + i, last) * FLD + k]; },
                                      [&](int k, int j) { return Wa[(fn + j) * FLD + k]; });
                    }
                    __syncthreads();                      // every read of dG_l (in dZ's buffer) is done
#pragma unroll
                    for (int nt = 0; nt < XTW; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = frow + r, col = fn + nt * 16 + l16;
                            if (row < NP) {
                                const float d = row < N ? dH[row * FLD + col] + dxl[0][nt][r] : 0.f;
                                if (row < N) dH[row * FLD + col] = d;
                                if (l > 0) dZ[row * FLD + col] = ((mask[l > 0 ? l - 1 : 0] >> (nt * 4 + r)) & 1u) ? d : 0.f;
                            }
                        }
                    __syncthreads();
                }
            }
            f32x4 dx[1][XTW];
            clear<1, XTW>(dx);
            if constexpr (!LW) {
            // through the row softmax: dS_ij = A_ij (dA_ij - sum_k dA_ik A_ik)
            if (a_on)
                each<1, NTW>(dAacc, [&](int i, int j, float v, int, int, int) {
                    const int row = am + i, col = an + j;
                    if (row < NP) dA[row * ALD + col] = (row < N && col < N && from_s) ? v : 0.f;      // constant adjacency: dS = 0
                });
            if (embedded) make_G(X);           // dT's buffer is free again
            __syncthreads();
            // through the squared normalisation: dS_ij = 2 S_ij (dA_ij - sum_k dA_ik A_ik) / Z_i, S_ij = sign sqrt(|A_ij| Z_i)
            for (int row = wave * 4 + kq; norm == 1 && row < N; row += W * 4) {
                float* d = dA + row * ALD;
                const float* p = A + row * ALD;
                const float zi = p[NT * 16];
                float dv[NT], pv[NT], dot = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bool ok = l16 + 16 * j < N;
                    dv[j] = ok ? d[l16 + 16 * j] : 0.f;
                    pv[j] = ok ? p[l16 + 16 * j] : 0.f;
                    dot = fmaf(dv[j], fabsf(pv[j]), dot);
                }
                dot = row16_sum(dot);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (l16 + 16 * j < N) d[l16 + 16 * j] = 2.f * copysignf(sqrtf(fabsf(pv[j]) / zi), pv[j]) * (dv[j] - dot);
            }
            for (int row = wave * 4 + kq; (norm == 0 || (COS && norm == 5)) && row < N; row += W * 4) {
                float* d = dA + row * ALD;
                const float* p = A + row * ALD;
                float dv[NT], pv[NT], dot = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bool ok = l16 + 16 * j < N;
                    dv[j] = ok ? d[l16 + 16 * j] : 0.f;
                    pv[j] = ok ? p[l16 + 16 * j] : 0.f;
                    dot = fmaf(dv[j], pv[j], dot);
                }
                dot = row16_sum(dot);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (l16 + 16 * j < N) d[l16 + 16 * j] = pv[j] * (dv[j] - dot);
            }
            if constexpr (COS) if (cosine) {
                // through C_ij = S_ij / (m_i m_j), m_i = sqrt(sum_k S_ik^2):
                //   dS_ij = dC_ij / (m_i m_j) - (r_i + c_i) S_ij / m_i^2,   r_i = sum_j dC_ij C_ij,  c_i = sum_j dC_ji C_ji
                // (m_i enters row i AND column i of C).  S is recomputed into A's buffer -- nothing reads the adjacency after the
                // layer loop -- next to the norms the forward sweep left in the padding column; dA holds dC.
                __syncthreads();
                if (a_on) {
                    f32x4 acc[1][NTW];
                    clear<1, NTW>(acc);
                    mm<1, NTW, 8>(acc, XW / 4, [&](int i, int k) { return X[min(am + i, last) * FLD + k]; },
                                  [&](int k, int j) { return X[min(an + j, last) * FLD + k]; });
                    each<1, NTW>(acc, [&](int i, int j, float v, int, int, int) {
                        const int row = am + i, col = an + j;
                        if (row < NP) A[row * ALD + col] = (row < N && col < N) ? v : 0.f;
                    });
                }
                __syncthreads();
                for (int row = wave * 4 + kq; row < N; row += W * 4) {          // r_i + c_i -> dA's padding column
                    const float mi = A[row * ALD + NT * 16];
                    float e = 0.f;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int col = l16 + 16 * j;
                        if (col < N) {
                            const float inv = 1.f / (mi * A[col * ALD + NT * 16]);
                            e = fmaf(dA[row * ALD + col], A[row * ALD + col] * inv, e);
                            e = fmaf(dA[col * ALD + row], A[col * ALD + row] * inv, e);
                        }
                    }
                    e = row16_sum(e);
                    if (l16 == 0) dA[row * ALD + NT * 16] = e;
                }
                __syncthreads();
                for (int row = wave * 4 + kq; row < N; row += W * 4) {
                    float* d = dA + row * ALD;
                    const float mi = A[row * ALD + NT * 16], back = d[NT * 16] / (mi * mi);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int col = l16 + 16 * j;
                        if (col < N) d[col] = d[col] / (mi * A[col * ALD + NT * 16]) - back * A[row * ALD + col];
                    }
                }
            }
            __syncthreads();
            // S = G X^T:  dG = dS X ;  dX += dS^T G        G = X Wa:  dWa += X^T dG ;  dX += dG Wa^T     (gaussian: G = X, dX += dG)
            float* dG = dZ;
            {
                f32x4 acc[1][XTW];
                clear<1, XTW>(acc);
                mm<1, XTW, 8>(acc, NK, [&](int i, int k) { return dA[min(fm + i, last) * ALD + k]; },
                              [&](int k, int j) { return X[k * FLD + fn + j]; });
                put(dG, acc);
                clear<1, XTW>(dx);
                mm<1, XTW, 8>(dx, NK, [&](int mi, int k) { return dA[k * ALD + fm + mi]; },
                              [&](int k, int j) { return GX[k * FLD + fn + j]; });
                if (!embedded) {
#pragma unroll
                    for (int nt = 0; nt < XTW; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) dx[0][nt][r] += acc[0][nt][r];
                }
            }
            __syncthreads();
            if (embedded) {
                if (g_on)
                    mm<GM, GN, 8>(gWa, NK, [&](int mi, int k) { return X[k * FLD + gm + mi]; },
                                  [&](int k, int j) { return dG[k * FLD + gn + j]; });
                mm<1, XTW, 8>(dx, XW / 4, [&](int i, int k) { return dG[min(fm + i, last) * FLD + k]; },
                              [&](int k, int j) { return Wa[(fn + j) * FLD + k]; });
            }
            }
            float* dxr = a.dXr + (size_t)s * a.xr_stride;
            float* dxh = a.dXh + (size_t)s * a.xh_stride - XW;
#pragma unroll
            for (int nt = 0; nt < XTW; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = frow + r, col = fn + nt * 16 + l16;
                    if (row < N) (row == 0 ? dxr : dxh + (size_t)row * XW)[col] = dx[0][nt][r] + dH[row * FLD + col];
                }
            __syncthreads();
        }
    }
    if constexpr (BWD) {
        if (g_on) {
            float* slab = a.slabs + (size_t)blockIdx.x * ((embedded ? 1 : 0) + L) * XW * XW;
            if (embedded) {
                each<GM, GN>(gWa, [&](int i, int j, float v, int, int, int) { slab[(gm + i) * XW + gn + j] = v; });
                slab += XW * XW;
            }
#pragma unroll
            for (int l = 0; l < L; ++l)
                each<GM, GN>(gW[l], [&](int i, int j, float v, int, int, int) { slab[l * XW * XW + (gm + i) * XW + gn + j] = v; });
        }
    }
}

// ------------------------------------------------------------------------------------------------
// slabs -> gradient vector
// ------------------------------------------------------------------------------------------------
constexpr int kMaxRanges = 8;
struct Range {
    const float* slabs;     // [count][n]; count = 0: the range is zero (detached parameters)
    int count, n, dst;
};
struct RangeArgs {
    Range r[kMaxRanges];
    int n_ranges, n_params;
};
// 256 threads = 32 parameters x 8 slab lanes: lane q sums slabs q, q + 8, .. in order (16 loads in flight), the eight partial sums
// are added in lane order -- a fixed summation tree, so the result is deterministic.
__global__ __launch_bounds__(256) void reduce_ranges_kernel(const RangeArgs a, float* __restrict__ out) {
    __shared__ float part[8][33];
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + c;
    float acc = 0.f;
    if (k < a.n_params) {
        int ri = 0;
        for (int i = 1; i < a.n_ranges; ++i)
            if (k >= a.r[i].dst) ri = i;
        const Range& R = a.r[ri];
        const float* src = R.slabs + (k - R.dst);
        int s = q;
        for (; s + 15 * 8 < R.count; s += 16 * 8) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(s + 8 * u) * R.n];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
        for (; s < R.count; s += 8) acc += src[(size_t)s * R.n];
    }
    part[q][c] = acc;
    __syncthreads();
    if (q == 0 && k < a.n_params) {
        float t = part[0][c];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += part[u][c];
        out[k] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
int mlp_params(const RglMlp& m) {
    int n = 0;
    for (int l = 0; l < m.n_layers; ++l) n += m.dims[l] * m.dims[l + 1] + m.dims[l + 1];
    return n;
}

void plan_rows_job(RowsJob& J, const RglMlp& m, int n_rows, int max_waves) {
    J = RowsJob{};
    J.m = m;
    int off = 0, col = 0, widest = 0, wl = 0;
    for (int l = 0; l < m.n_layers; ++l) {
        J.w_off[l] = off; off += m.dims[l] * m.dims[l + 1];
        J.b_off[l] = off; off += m.dims[l + 1];
        J.w_ld[l] = m.dims[l + 1] | 1;
        J.w_lds[l] = wl; wl += (m.dims[l] * J.w_ld[l] + 3) & ~3;
        J.b_lds[l] = wl; wl += (m.dims[l + 1] + 3) & ~3;
    }
    for (int l = 0; l <= m.n_layers; ++l) {
        J.act_off[l] = col;
        col += (m.dims[l] + 3) & ~3;
        widest = m.dims[l] > widest ? m.dims[l] : widest;
    }
    J.weight_floats = wl;
    J.n_params = off;
    J.act_ld = col + 2;                         // rows two banks apart
    J.d_ld = ((widest + 3) & ~3) + 2;
    J.n_rows = n_rows;
    J.n_tiles = (n_rows + 15) / 16;
    J.n_waves = J.n_tiles < max_waves ? J.n_tiles : max_waves;
    if (m.n_layers == 2 && m.dims[1] == 64 && m.dims[0] <= 32 && m.dims[2] <= 32) {
        const int T0 = (m.dims[0] + 15) / 16, T2 = (m.dims[2] + 15) / 16;
        J.kind = 10 * T0 + T2;
        J.w_ld[0] = 65; J.w_ld[1] = T2 * 16 + 1;           // the strides mlp2_rows_kernel<T0, T2> is compiled for
        J.w_lds[0] = 0; J.b_lds[0] = (T0 * 16 * 65 + 3) & ~3;        // W0 padded to T0 * 16 rows, W1 / b1 to T2 * 16 columns
        J.w_lds[1] = J.b_lds[0] + 64; J.b_lds[1] = J.w_lds[1] + ((64 * J.w_ld[1] + 3) & ~3);
        J.weight_floats = J.b_lds[1] + 32;
        J.waves_per_wg = kNarrowWaves;
        J.n_wgs = (J.n_waves + kNarrowWaves - 1) / kNarrowWaves;
        J.wave_floats = 16 * ((T0 * 16 + 2) + 66 + (T2 * 16 + 2));
        return;
    }
    static const bool direct_rows = [] { const char* e = getenv("RGL_HEAD_ROWS_DIRECT"); return !e || atoi(e) != 0; }();
    if (direct_rows && J.n_tiles <= 1024) {
        // few tiles: a workgroup per tile, the weights straight from L2 (head_rows_kernel): the tile's activations and deltas are all
        // that lives in LDS, every layer in a region of a multiple of 16 columns, rows 16 bytes aligned and four banks apart
        col = 0; widest = 0;
        for (int l = 0; l <= m.n_layers; ++l) {
            J.act_off[l] = col;
            col += (m.dims[l] + 15) & ~15;
            widest = m.dims[l] > widest ? m.dims[l] : widest;
        }
        J.act_ld = col + 4;
        J.d_ld = ((widest + 15) & ~15) + 4;
        J.wave_floats = 16 * J.act_ld + 32 * J.d_ld;
        J.weight_floats = 0;
        J.kind = 1;
        J.coop = 1;
        J.waves_per_wg = 1;
        J.n_wgs = J.n_waves;
        return;
    }
    J.wave_floats = 16 * J.act_ld + 32 * J.d_ld;
    if (J.n_tiles <= 1024 && ((size_t)wl + J.wave_floats) * sizeof(float) <= (size_t)rgl::kLdsBytesPerCu - 1024) {
        J.coop = 1;                             // few tiles: a workgroup per tile (the value head: one row per scene)
        J.waves_per_wg = 1;                     // LDS slices per workgroup
        J.n_wgs = J.n_waves;
        return;
    }
    {   // all layers do not fit next to a tile (path G's head): a workgroup per tile, one layer's weights in LDS at a time
        int wmax = 0, bmax = 0;
        for (int l = 0; l < m.n_layers; ++l) {
            const int wf = (m.dims[l] * J.w_ld[l] + 3) & ~3, bf = (m.dims[l + 1] + 3) & ~3;
            wmax = wf > wmax ? wf : wmax;
            bmax = bf > bmax ? bf : bmax;
        }
        if (((size_t)wl + J.wave_floats) * sizeof(float) > (size_t)rgl::kLdsBytesPerCu - 1024 &&
            ((size_t)wmax + bmax + J.wave_floats) * sizeof(float) <= (size_t)rgl::kLdsBytesPerCu - 1024) {
            J.coop = 2;
            for (int l = 0; l < m.n_layers; ++l) { J.w_lds[l] = 0; J.b_lds[l] = wmax; }
            J.weight_floats = wmax + bmax;
            J.waves_per_wg = 1;
            J.n_wgs = J.n_waves;
            return;
        }
    }
    // waves per workgroup: two workgroups per CU when a half of the LDS holds the weights and at least one wave's tile, else one
    // workgroup; 0 = the MLP does not fit this kernel (the caller takes another path)
    const long per_wave = (long)J.wave_floats * (long)sizeof(float), weights = (long)wl * (long)sizeof(float);
    const long lds = (long)rgl::kLdsBytesPerCu - 1024;
    long wpw = (lds / 2 - weights) / per_wave;
    if (wpw < 1) wpw = (lds - weights) / per_wave;
    J.waves_per_wg = wpw < 0 ? 0 : (wpw > 4 ? 4 : (int)wpw);
    J.n_wgs = J.waves_per_wg > 0 ? (J.n_waves + J.waves_per_wg - 1) / J.waves_per_wg : 0;
}
size_t rows_job_lds(const RowsJob& J) { return ((size_t)J.weight_floats + (size_t)J.waves_per_wg * J.wave_floats) * sizeof(float); }

template <class K>
int launch_rows_kernel(K kernel, RowsArgs& ra, hipStream_t st) {
    size_t lds = 0;
    int wgs = 0;
    for (int j = 0; j < ra.n_jobs; ++j) {
        const size_t b = rows_job_lds(ra.job[j]);
        lds = b > lds ? b : lds;
        ra.job[j].wg_begin = wgs;
        wgs += ra.job[j].n_wgs;
    }
    if (lds > 64 * 1024)
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    bool any_coop = false;
    for (int j = 0; j < ra.n_jobs; ++j) any_coop |= ra.job[j].coop != 0;
    hipLaunchKernelGGL(kernel, dim3(wgs), dim3(ra.job[0].kind >= 10 ? kNarrowWaves * 64 : (any_coop ? kCoopWaves * 64 : 256)), lds, st, ra);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

// The narrow jobs that share a launch (same kernel kind) get the same number of tiles per wave, chosen so that all their waves are
// resident at once (3 workgroups of kNarrowWaves waves per CU): a launch in one round with 3 tiles per wave beats 2 tiles per wave
// plus a second round for the overflow.
void balance_narrow(RowsJob* const* jobs, int n, int max_waves) {
    for (int j = 0; j < n; ++j) {
        if (!jobs[j] || jobs[j]->kind < 10) continue;
        bool first_of_kind = true;
        long tiles = 0;
        for (int i = 0; i < n; ++i)
            if (jobs[i] && jobs[i]->kind == jobs[j]->kind) {
                if (i < j) first_of_kind = false;
                tiles += jobs[i]->n_tiles;
            }
        if (!first_of_kind) continue;
        long capacity = 256L * 3 * kNarrowWaves;
        capacity = capacity < max_waves ? capacity : max_waves;
        const int per_wave = (int)((tiles + capacity - 1) / capacity);
        for (int i = 0; i < n; ++i)
            if (jobs[i] && jobs[i]->kind == jobs[j]->kind) {
                RowsJob& J = *jobs[i];
                J.n_waves = (J.n_tiles + per_wave - 1) / per_wave;
                J.n_wgs = (J.n_waves + kNarrowWaves - 1) / kNarrowWaves;
            }
    }
}

// the jobs of one pipeline stage: one launch per kernel kind among them
int launch_rows(RowsArgs& all, hipStream_t st) {
    bool done[kMaxRowJobs] = {};
    for (int j = 0; j < all.n_jobs; ++j) {
        if (done[j]) continue;
        RowsArgs ra{};
        ra.backward = all.backward;
        const int kind = all.job[j].kind;
        for (int i = j; i < all.n_jobs; ++i)
            if (!done[i] && all.job[i].kind == kind) { ra.job[ra.n_jobs++] = all.job[i]; done[i] = true; }
        int rc;
        switch (kind) {
            case 11: rc = launch_rows_kernel(mlp2_rows_kernel<1, 1>, ra, st); break;
            case 12: rc = launch_rows_kernel(mlp2_rows_kernel<1, 2>, ra, st); break;
            case 21: rc = launch_rows_kernel(mlp2_rows_kernel<2, 1>, ra, st); break;
            case 22: rc = launch_rows_kernel(mlp2_rows_kernel<2, 2>, ra, st); break;
            case 1: rc = launch_rows_kernel(head_rows_kernel, ra, st); break;
            default: rc = launch_rows_kernel(mlp_rows_kernel, ra, st); break;
        }
        if (rc) return rc;
    }
    return RGL_OK;
}

template <int NT, int XT, int L, bool BWD, bool COS, bool LW = false>
int launch_graph_kernel(const GraphArgs& ga, size_t lds, int grid, hipStream_t st) {
    auto kern = graph_kernel<NT, XT, L, BWD, COS, LW>;
    if (lds > 64 * 1024)
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT * 128), lds, st, ga);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}
template <int NT, int XT, int L>
int launch_graph_nxl(const GraphArgs& ga, bool bwd, size_t lds, int grid, hipStream_t st) {
    if (ga.lw) {        // layerwise graphs: the softmax / squared normalisations, up to 32 nodes of 32 features (tiles_cover)
        if constexpr (XT == 2 && NT <= 2)
            return bwd ? launch_graph_kernel<NT, XT, L, true, false, true>(ga, lds, grid, st)
                       : launch_graph_kernel<NT, XT, L, false, false, true>(ga, lds, grid, st);
        else
            return 1;
    }
    if (ga.norm >= 4)
        return bwd ? launch_graph_kernel<NT, XT, L, true, true>(ga, lds, grid, st) : launch_graph_kernel<NT, XT, L, false, true>(ga, lds, grid, st);
    return bwd ? launch_graph_kernel<NT, XT, L, true, false>(ga, lds, grid, st) : launch_graph_kernel<NT, XT, L, false, false>(ga, lds, grid, st);
}
template <int NT, int XT>
int launch_graph_nx(const GraphArgs& ga, int L, bool bwd, size_t lds, int grid, hipStream_t st) {
    switch (L) {
        case 1: return launch_graph_nxl<NT, XT, 1>(ga, bwd, lds, grid, st);
        case 2: return launch_graph_nxl<NT, XT, 2>(ga, bwd, lds, grid, st);
        default: return launch_graph_nxl<NT, XT, 3>(ga, bwd, lds, grid, st);
    }
}

struct GraphPlan { int grid; size_t lds; };
template <int NT, int XT>
GraphPlan plan_graph_nx(int S, int N, int L, bool bwd, bool lw) {
    GraphPlan p{};
    p.lds = (size_t)(GraphLds<NT, XT>::weight_floats(L) + GraphLds<NT, XT>::scene_floats(N, L, bwd, lw)) * sizeof(float);
    if (p.lds > (size_t)rgl::kLdsBytesPerCu) return p;
    // persistent workgroups: as many as are resident at once (LDS, and the waves the kernel's register budget allows), scenes dealt
    // round robin
    int per_cu = (int)((size_t)rgl::kLdsBytesPerCu / p.lds);
    const int by_waves = lw ? 2 : ((NT == 2 && XT == 2) ? 16 : (NT == 1 ? 12 : 8)) / (2 * NT);
    per_cu = per_cu > by_waves ? by_waves : per_cu;
    per_cu = per_cu < 1 ? 1 : per_cu;
    const int resident = 256 * per_cu;
    p.grid = S < resident ? S : resident;
    return p;
}
// x_dim 32 or 64
GraphPlan plan_graph(int S, int N, int X, int L, bool bwd, bool lw = false) {
    if (X == 32) return N <= 16 ? plan_graph_nx<1, 2>(S, N, L, bwd, lw) : (N <= 32 ? plan_graph_nx<2, 2>(S, N, L, bwd, lw) : plan_graph_nx<4, 2>(S, N, L, bwd, lw));
    return N <= 16 ? plan_graph_nx<1, 4>(S, N, L, bwd, lw) : (N <= 32 ? plan_graph_nx<2, 4>(S, N, L, bwd, lw) : plan_graph_nx<4, 4>(S, N, L, bwd, lw));
}
int launch_graph(const GraphArgs& ga, int X, int L, bool bwd, const GraphPlan& p, hipStream_t st) {
    if (X == 32) {
        if (ga.N <= 16) return launch_graph_nx<1, 2>(ga, L, bwd, p.lds, p.grid, st);
        if (ga.N <= 32) return launch_graph_nx<2, 2>(ga, L, bwd, p.lds, p.grid, st);
        return launch_graph_nx<4, 2>(ga, L, bwd, p.lds, p.grid, st);
    }
    if (ga.N <= 16) return launch_graph_nx<1, 4>(ga, L, bwd, p.lds, p.grid, st);
    if (ga.N <= 32) return launch_graph_nx<2, 4>(ga, L, bwd, p.lds, p.grid, st);
    return launch_graph_nx<4, 4>(ga, L, bwd, p.lds, p.grid, st);
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}

}  // namespace

#ifdef RGL_PHASE_TIMING
extern "C" int rgl_debug_read_backward_phase_cycles(unsigned long long* out16, int reset) {
    RGL_HIP_TRY(hipDeviceSynchronize());
    RGL_HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {0};
        RGL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)));
    }
    return 0;
}
#endif

namespace {

// dst[0..n) = src[0..n), or zero (src == null); grid-stride
__global__ __launch_bounds__(256) void init_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src ? src[i] : 0.f;
}

// what the tile kernels cover: embedded_gaussian / gaussian (softmax of S) and -- round 5 -- squared / equal_attention / diagonal
// (plain weights: graph_model.py:86-93) and cosine / cosine_softmax (:70-79), one adjacency for all layers, x_dim 32 or 64, 1-3
// layers, N <= 64; any embedding MLPs and heads within the ABI limits; layerwise graphs (round 6) with embedded_gaussian / gaussian
// / squared at x_dim 32, N <= 32 (and equal_attention / diagonal, whose adjacency is a constant).  The pair-MLP similarity
// (concatenation) and layerwise graphs of the cosine family stay on the per-scene kernels.
int tiles_norm(const RglGraph& g) {
    switch (g.similarity) {
        case RGL_SIM_EMBEDDED_GAUSSIAN: case RGL_SIM_GAUSSIAN: return 0;
        case RGL_SIM_SQUARED: return 1;
        case RGL_SIM_EQUAL_ATTENTION: return 2;
        case RGL_SIM_DIAGONAL: return 3;
        case RGL_SIM_COSINE: return 4;
        case RGL_SIM_COSINE_SOFTMAX: return 5;
        default: return -1;
    }
}
bool tiles_cover(const RglGraph& g, int H) {
    const int N = H + 1, L = g.num_layer;
    if ((g.x_dim != 32 && g.x_dim != 64) || L < 1 || L > 3 || N > 64 || H < 1) return false;
    // layerwise graphs (round 6): the softmax and the squared normalisations at the shipped feature width, up to 32 nodes; the
    // constant adjacencies (equal_attention, diagonal) do not depend on the layer's input at all -- the one-adjacency kernels
    if (g.layerwise_graph) {
        const int nm = tiles_norm(g);
        if (!(nm == 2 || nm == 3 || ((nm == 0 || nm == 1) && g.x_dim == 32 && N <= 32))) return false;
    }
    return tiles_norm(g) >= 0;
}

void graph_args(GraphArgs& ga, const RglGraph& g, int S, int N, int spc) {
    ga = GraphArgs{};
    ga.w_a = g.similarity == RGL_SIM_EMBEDDED_GAUSSIAN ? g.w_a : nullptr;
    for (int l = 0; l < g.num_layer; ++l) ga.Ws[l] = g.Ws[l];
    ga.S = S; ga.N = N; ga.skip = g.skip_connection ? 1 : 0; ga.spc = spc;
    ga.norm = tiles_norm(g);
    ga.lw = (g.layerwise_graph && ga.norm <= 1) ? 1 : 0;        // constant adjacencies: layerwise or not is the same graph
    ga.xr_stride = g.x_dim; ga.xh_stride = (N - 1) * g.x_dim;
}

struct Taker {                 // carves 256-byte aligned pieces out of a workspace
    char* base;
    size_t used = 0;
    template <class T>
    T* take(size_t count) {
        T* p = (T*)(base + used);
        used += (count * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

}  // namespace

namespace rgl {

// ------------------------------------------------------------------------------------------------
// forward on the same tile kernels: models the shipped-shape MFMA kernels (rgl_scene.hip, rgl_fused.hip, ..) do not cover -- other
// embedding MLPs (wr_dims / wh_dims), x_dim = 64 -- instead of the general VALU kernel.  0 bytes / 1 = not covered.
//   1. mlp rows (forward): robot rows [S][X], human rows [S / spc][H][X] (sibling scenes share their crowd's rows)
//   2. graph_kernel<.., false>: H_L -- all rows (motion head, H_out) or the robot row only (value head)
//   3. mlp rows (forward): value head on row 0 -> value_out, motion head on rows 1.. -> humans_next
// ------------------------------------------------------------------------------------------------
size_t tiles_forward_workspace_bytes(const RglGraph* g, const RglMlp* vh, const RglMlp* mh, int S, int spc, int H, int want_H) {
    if (!g || !tiles_cover(*g, H) || S < 1 || spc < 1 || S % spc) return 0;
    const bool has_m = mh && mh->n_layers > 0;
    const size_t X = g->x_dim, N = H + 1;
    const size_t hl = (has_m && !want_H) ? (size_t)S * N * X : ((!has_m && !want_H) ? (size_t)S * X : 0);     // H_out doubles as H_L
    return ((size_t)S * X + (size_t)(S / spc) * H * X + hl) * sizeof(float) + 3 * 256;
}

int launch_tiles_forward(const RglGraph* graph, const RglMlp* vh, const RglMlp* mh, const float* robot, const float* humans, int S,
                         int spc, int H, float* H_out, float* value_out, float* humans_next, void* workspace, size_t workspace_bytes,
                         hipStream_t st) {
    const bool has_v = vh && vh->n_layers > 0 && value_out, has_m = mh && mh->n_layers > 0 && humans_next;
    if (!graph || !workspace || !tiles_cover(*graph, H) || S < 1 || spc < 1 || S % spc) return 1;
    if (env_int("RGL_TILES_FORWARD", 1) == 0) return 1;          // measurements: the general kernel instead
    if (workspace_bytes < tiles_forward_workspace_bytes(graph, has_v ? vh : nullptr, has_m ? mh : nullptr, S, spc, H, H_out != nullptr))
        return 1;
    const RglGraph& g = *graph;
    const int N = H + 1, L = g.num_layer, X = g.x_dim, crowds = S / spc;
    const GraphPlan gp = plan_graph(S, N, X, L, false, g.layerwise_graph != 0 && tiles_norm(g) <= 1);
    if (gp.grid < 1) return 1;
    Taker ws{(char*)workspace};
    float* Xr = ws.take<float>((size_t)S * X);
    float* Xh = ws.take<float>((size_t)crowds * H * X);
    const bool row0 = !has_m && !H_out;
    float* HL = H_out ? H_out : ws.take<float>(row0 ? (size_t)S * X : (size_t)S * N * X);
    RowsJob j_wr, j_wh, j_v, j_m;
    plan_rows_job(j_wr, g.w_r, S, 2048);
    plan_rows_job(j_wh, g.w_h, crowds * H, 2048);
    if (has_v) plan_rows_job(j_v, *vh, S, 2048);
    if (has_m) plan_rows_job(j_m, *mh, S * H, 2048);
    {
        RowsJob* emb[2] = {&j_wr, &j_wh};
        RowsJob* heads[2] = {has_v ? &j_v : nullptr, has_m ? &j_m : nullptr};
        balance_narrow(emb, 2, 2048);
        balance_narrow(heads, 2, 2048);
    }
    for (const RowsJob* J : {&j_wr, &j_wh, has_v ? &j_v : nullptr, has_m ? &j_m : nullptr})
        if (J && J->waves_per_wg < 1) return 1;
    j_wr.in = RowMap{(float*)robot, 1, 0, (long long)g.w_r.dims[0]};
    j_wr.out = RowMap{Xr, 1, 0, (long long)X};
    j_wh.in = RowMap{(float*)humans, H, g.w_h.dims[0], (long long)H * g.w_h.dims[0]};
    j_wh.out = RowMap{Xh, H, X, (long long)H * X};
    {
        RowsArgs ra{};
        ra.job[0] = j_wr; ra.job[1] = j_wh; ra.n_jobs = 2; ra.backward = 0;
        const int rc = launch_rows(ra, st);
        if (rc) return rc;
    }
    GraphArgs ga;
    graph_args(ga, g, S, N, spc);
    ga.Xr = Xr; ga.Xh = Xh; ga.HL = HL; ga.hl_row0 = row0 ? 1 : 0;
    int rc = launch_graph(ga, X, L, false, gp, st);
    if (rc) return rc;
    if (has_v || has_m) {
        const long long hs = row0 ? X : (long long)N * X;
        RowsArgs ra{};
        ra.backward = 0;
        if (has_v) {
            j_v.in = RowMap{HL, 1, 0, hs};
            j_v.out = RowMap{value_out, 1, 0, 1};
            ra.job[ra.n_jobs++] = j_v;
        }
        if (has_m) {
            const int od = mh->dims[mh->n_layers];
            j_m.in = RowMap{HL + X, H, X, (long long)N * X};
            j_m.out = RowMap{humans_next, H, od, (long long)H * od};
            ra.job[ra.n_jobs++] = j_m;
        }
        rc = launch_rows(ra, st);
        if (rc) return rc;
    }
    return RGL_OK;
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// 1 = not this path (outside the envelope, below the batch threshold, or the caller's workspace cannot hold the intermediates):
// the per-scene VALU kernel of rgl_backward.hip runs.  Slab order of grad_out as documented in rgl_hip.h.
static int backward_tiles(const RglGraph* graph, const RglMlp* vh, const RglMlp* mh, const float* robot, const float* humans,
                         int S, int H, int detach_graph, const float* d_value, const float* d_humans_next, const float* d_H,
                         float* grad_out, void* workspace, size_t workspace_bytes, hipStream_t st, int only_choice) {
    // RGL_BACKWARD_MFMA = 0: never, 1: whenever the structure allows; default: by batch size, and whenever the per-scene kernel cannot
    // hold a scene in LDS (only_choice)
    const int mode = env_int("RGL_BACKWARD_MFMA", -1);
    if (mode == 0) return 1;
    if (mode < 1 && !only_choice && S < env_int("RGL_BACKWARD_MFMA_MIN", 256)) {
        // Below the threshold the pipeline's device time is still the shorter one (84 vs 94 us at 100 scenes of 6 nodes, 95 vs 125 us
        // at 20 nodes), but it is seven launches instead of two and an eager training step is bound by the host.  While the stream
        // is being captured into a hipGraph only the device time counts.
        // (never asked of the legacy NULL stream: querying it while another stream captures would invalidate that capture)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (!st || hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusActive) return 1;
    }
    const RglGraph& g = *graph;
    if (!tiles_cover(g, H)) return 1;
    const int N = H + 1, L = g.num_layer, X = g.x_dim;
    const bool has_v = vh && vh->n_layers > 0, has_m = mh && mh->n_layers > 0;
    const bool embedded = g.similarity == RGL_SIM_EMBEDDED_GAUSSIAN, lw = g.layerwise_graph != 0 && tiles_norm(g) <= 1;
    const GraphPlan gpf = plan_graph(S, N, X, L, false, lw), gp_full = plan_graph(S, N, X, L, true, lw);
    if (gp_full.grid < 1 || gpf.grid < 1) return 1;

    // gradient vector: w_r | w_h | w_a | Ws | value head | motion head
    const int n_wr = mlp_params(g.w_r), n_wh = mlp_params(g.w_h), n_graph = ((embedded ? 1 : 0) + L) * X * X;
    const int n_v = has_v ? mlp_params(*vh) : 0, n_m = has_m ? mlp_params(*mh) : 0;
    const int o_wr = 0, o_wh = n_wr, o_graph = o_wh + n_wh, o_v = o_graph + n_graph, o_m = o_v + n_v, n_params = o_m + n_m;

    // workspace: X | H_L | dH_L, each [S][N][X] | slabs of the row jobs and of the graph kernel.  dX replaces dH_L in place (a scene's
    // upstream rows are in the workgroup's LDS long before its dX rows are written, and no other workgroup touches them)
    Taker ws{(char*)workspace};
    const size_t feat = (size_t)S * N * X;
    float* Xs = ws.take<float>(feat);
    float* HL = ws.take<float>(feat);
    float* dHL = ws.take<float>(feat);
    float* dXs = dHL;
    // one slab per wave: fewer waves per row job (more tiles each) when the caller's workspace -- sized for the per-scene kernel's
    // slabs, n_scenes x n_params floats -- is short (few scenes of many nodes)
    RowsJob j_wr, j_wh, j_v, j_m;
    float* g_slabs = nullptr;
    const size_t used_feat = ws.used;
    GraphPlan gp = gp_full;
    for (int max_waves = 2048; max_waves >= 1; max_waves >>= 1) {
        ws.used = used_feat;
        gp.grid = gp_full.grid < max_waves ? gp_full.grid : max_waves;
        plan_rows_job(j_wr, g.w_r, S, max_waves);
        plan_rows_job(j_wh, g.w_h, S * H, max_waves);
        if (has_v) plan_rows_job(j_v, *vh, S, max_waves);
        if (has_m) plan_rows_job(j_m, *mh, S * H, max_waves);
        {   // the launches: (w_r, w_h) forward and backward, (value head, motion head)
            RowsJob* emb[2] = {&j_wr, &j_wh};
            RowsJob* heads[2] = {has_v ? &j_v : nullptr, has_m ? &j_m : nullptr};
            balance_narrow(emb, 2, max_waves);
            balance_narrow(heads, 2, max_waves);
        }
        auto slabs_for = [&](RowsJob& J) { J.slabs = ws.take<float>((size_t)J.n_waves * J.n_params); };
        if (!detach_graph) { slabs_for(j_wr); slabs_for(j_wh); }
        if (has_v) slabs_for(j_v);
        if (has_m) slabs_for(j_m);
        g_slabs = detach_graph ? nullptr : ws.take<float>((size_t)gp.grid * n_graph);
        if (ws.used <= workspace_bytes) break;
    }
    if (ws.used > workspace_bytes) return 1;
    for (const RowsJob* J : {&j_wr, &j_wh, has_v ? &j_v : nullptr, has_m ? &j_m : nullptr})
        if (J && J->waves_per_wg < 1) return 1;

    // 1. embeddings
    j_wr.in = RowMap{(float*)robot, 1, 0, (long long)g.w_r.dims[0]};
    j_wr.out = RowMap{Xs, 1, 0, (long long)N * X};
    j_wh.in = RowMap{(float*)humans, H, g.w_h.dims[0], (long long)H * g.w_h.dims[0]};
    j_wh.out = RowMap{Xs + X, H, X, (long long)N * X};
    {
        RowsArgs ra{};
        ra.job[0] = j_wr; ra.job[1] = j_wh; ra.n_jobs = 2; ra.backward = 0;
        const int rc = launch_rows(ra, st);
        if (rc) return rc;
    }
    // 2. graph forward
    GraphArgs ga;
    graph_args(ga, g, S, N, 1);
    ga.Xr = Xs; ga.Xh = Xs + X; ga.dHL = dHL; ga.HL = HL; ga.dXr = dXs; ga.dXh = dXs + X; ga.slabs = g_slabs;
    ga.xr_stride = ga.xh_stride = N * X;
    if (has_v || has_m) {
        const int rc = launch_graph(ga, X, L, false, gpf, st);
        if (rc) return rc;
    }
    // 3. heads: dH_L starts as the caller's d_H (or zero) and receives the heads' input gradients
    // (a kernel, not hipMemsetAsync / hipMemcpyAsync: recorded into a captured training step and replayed, the memset NODE was not
    // reliably ordered against the kernel nodes around it -- on some boxes the heads' `+=` into dH_L met garbage (NaN parameters), on
    // others the zeros arrived after it and a step lost the heads' gradient (sporadic 2e-3 deviations of the parameters; never in an
    // eager step, never with the per-scene kernel, which has no memset: tools/micro/captured_step_repeatability.py))
    {
        const size_t n4 = (feat + 3) / 4;
        const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(init_rows_kernel, dim3(blocks), dim3(256), 0, st, dHL, d_H, feat);
        RGL_LAUNCH_CHECK();
    }
    if (has_v || has_m) {
        RowsArgs ra{};
        ra.backward = 1;
        if (has_v) {
            j_v.in = RowMap{HL, 1, 0, (long long)N * X};
            j_v.d_out = RowMap{(float*)d_value, 1, 0, 1};
            j_v.d_in = RowMap{dHL, 1, 0, (long long)N * X};
            j_v.need_din = detach_graph ? 0 : 1; j_v.din_add = 1;
            ra.job[ra.n_jobs++] = j_v;
        }
        if (has_m) {
            const int od = mh->dims[mh->n_layers];
            j_m.in = RowMap{HL + X, H, X, (long long)N * X};
            j_m.d_out = RowMap{(float*)d_humans_next, H, od, (long long)H * od};
            j_m.d_in = RowMap{dHL + X, H, X, (long long)N * X};
            j_m.need_din = detach_graph ? 0 : 1; j_m.din_add = 1;
            ra.job[ra.n_jobs++] = j_m;
        }
        const int rc = launch_rows(ra, st);
        if (rc) return rc;
    }
    if (!detach_graph) {
        // 4. graph backward
        int rc = launch_graph(ga, X, L, true, gp, st);
        if (rc) return rc;
        // 5. embeddings backward
        j_wr.d_out = RowMap{dXs, 1, 0, (long long)N * X};
        j_wh.d_out = RowMap{dXs + X, H, X, (long long)N * X};
        RowsArgs ra{};
        ra.job[0] = j_wr; ra.job[1] = j_wh; ra.n_jobs = 2; ra.backward = 1;
        rc = launch_rows(ra, st);
        if (rc) return rc;
    }
    // 6. slabs -> grad_out
    RangeArgs rr{};
    auto range = [&](const float* slabs, int count, int n, int dst) {
        if (n > 0) rr.r[rr.n_ranges++] = Range{slabs, count, n, dst};
    };
    range(detach_graph ? nullptr : j_wr.slabs, detach_graph ? 0 : j_wr.n_waves, n_wr, o_wr);
    range(detach_graph ? nullptr : j_wh.slabs, detach_graph ? 0 : j_wh.n_waves, n_wh, o_wh);
    range(g_slabs, detach_graph ? 0 : gp.grid, n_graph, o_graph);
    if (has_v) range(j_v.slabs, j_v.n_waves, n_v, o_v);
    if (has_m) range(j_m.slabs, j_m.n_waves, n_m, o_m);
    rr.n_params = n_params;
    hipLaunchKernelGGL(reduce_ranges_kernel, dim3((n_params + 31) / 32), dim3(256), 0, st, rr, grad_out);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}


// 1 = not this path (outside the envelope, below the batch threshold, or the caller's workspace cannot hold the intermediates): the
// per-scene VALU kernel of rgl_backward.hip runs -- unless RGL_BACKWARD_MFMA=2 (tests), which turns "not this path" into an error.
int launch_backward_mfma(const RglGraph* graph, const RglMlp* vh, const RglMlp* mh, const float* robot, const float* humans,
                         int S, int H, int detach_graph, const float* d_value, const float* d_humans_next, const float* d_H,
                         float* grad_out, void* workspace, size_t workspace_bytes, hipStream_t st, int only_choice) {
    const int rc = backward_tiles(graph, vh, mh, robot, humans, S, H, detach_graph, d_value, d_humans_next, d_H, grad_out, workspace,
                                  workspace_bytes, st, only_choice);
    return (rc == 1 && env_int("RGL_BACKWARD_MFMA", -1) == 2) ? RGL_ERR_BAD_MODE : rc;
}

}  // namespace rgl
