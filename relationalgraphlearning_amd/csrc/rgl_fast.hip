// rgl_fast.hip -- f32-MFMA kernels for the shipped configuration of the relational graph
// (embedded_gaussian similarity -- and the other per-row-sum normalisations: gaussian, squared, equal_attention,
// diagonal, see rgl_mfma.h --, one adjacency for all layers, x_dim 32, embedding MLPs
// 9->64->32 / 5->64->32, value head 32->D1->D2->D3->1), specialised for the shape the rollout
// actually produces: the A sibling children of one parent share their crowd and differ only in
// the robot row.
//
//   stage 1  children_graph_kernel : one workgroup per parent.
//       prologue  human embeddings Xh, G = Xh*Wa, S_hh = G*Xh^T            (once per parent, VALU)
//       B1/B2     robot embeddings of 16 children at a time as an MFMA chain in "transposed" form
//                 (activations = B operand, kept in registers: the 4 D registers of one MFMA are
//                 the B operands of 4 k-steps of the next, with the k index permuted to match),
//                 then the robot row / robot column of every child's similarity matrix
//       B3        per 16 (child,node) columns: softmax computed in-lane directly in the MFMA
//                 B-operand layout, A*X as MFMA with the SHARED human rows as A operand plus a
//                 rank-1 update for the per-child robot row, *W by MFMA with W in registers,
//                 relu (+skip); node features staged in wave-private LDS; the last layer needs
//                 only the robot node: t_c = A_c[0,:] * H_c  (wave-level reduction)
//   stage 2  robot_head_kernel     : h = relu(t*W_last) (+skip), value head; one register-resident
//                 MFMA chain per 16 children, weights as pre-permuted A fragments in LDS.
//
// f32 MFMA (v_mfma_f32_16x16x4_f32) is exact fp32 at the vector-FMA rate; results differ from the
// general kernel only by summation order.  Anything outside the envelope above falls back to the
// general kernel (same numbers, lower speed) -- never to the CPU.
//
// Follows (reference paths): crowd_nav/policy/graph_model.py:99-130, value_estimator.py:11-20,
// model_predictive_rl.py:245-250 (the loop whose iterations these kernels run side by side).
#include "rgl_mfma.h"

namespace rgl {
int launch_generic_forward(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head,
                           const float* robot, const float* humans, int n_scenes, int scenes_per_crowd, int H,
                           float* H_out, float* A_out, float* value_out, float* humans_next, hipStream_t stream);
int launch_deep_children(const RglGraph* g, int P, int A, int H, const float* child_robot, const float* humans_next,
                         float* rows_out, int f16, hipStream_t stream);      // rgl_deep.hip; 1 = outside its envelope
}

namespace {

// ------------------------------------------------------------------------------------------------
// stage 1
// ------------------------------------------------------------------------------------------------
#ifndef STAGE1_THREADS
#define STAGE1_THREADS 512
#endif
constexpr int kThreads1 = STAGE1_THREADS;        // stage 1: 8 waves per parent, two workgroups per CU -> 4 waves/SIMD
constexpr int kWaves1 = kThreads1 / 64;
#define STAGE1_WAVES_PER_SIMD (STAGE1_THREADS / 128)
#ifndef STAGE1_STAGGER
#define STAGE1_STAGGER 0                         // x64 cycles
#endif

struct ChildArgs {
    const float *wr1, *br1, *wr2, *br2;   // robot embedding, k-major: [9][64], [64], [64][32], [32]
    const float *wh1, *bh1, *wh2, *bh2;   // human embedding:          [5][64], [64], [64][32], [32]
    const float* wa;                      // [32][32]
    const float* Ws[RGL_MAX_GCN_LAYERS];  // [32][32] each; the LAST layer's weight is applied in stage 2
    int L, skip;
    int mode;                             // 1: L == 1   2: L == 2 (streamed robot-row aggregation)   3: L >= 3 (staged)
    const float* child_robot;             // [P][A][9]
    const float* humans;                  // [P][H][5]
    int P, A, H;
    float* rows_out;                      // [P*A][64] = [ t_c (32) | H_{L-1}[robot] (32) ]
    // derived layout (float offsets into LDS)
    int N, SLD, NT, CT, CPC, G, tiles_per_group, n_groups, GC;
    unsigned magicN;                      // floor(2^32 / N) + 1
    int n_waves;                          // waves per workgroup (4..8), chosen to balance n_groups
    int off_wh1, off_bh1, off_wh2, off_bh2, off_wa, off_wr1, off_br1, off_wr2, off_br2;   // persistent weight image
    int off_xh, off_shh, off_s0, off_sc0, off_x0, off_wave, wave_stride;                  // per-parent data
};

// VAGG: robot-row aggregation on the VALU (valid when a tile holds at most two children, i.e. N >= 16) instead of
// the general MFMA selector product.
template <int KS, int MODE, bool VAGG, bool SKIP>
__global__ __launch_bounds__(kThreads1, STAGE1_WAVES_PER_SIMD) void children_graph_kernel(const ChildArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int nthreads = a.n_waves * 64;
    const int n = lane & 15, q = lane >> 4;
    const int N = a.N, H = a.H, A = a.A, SLD = a.SLD;
    const float* wh1 = lds + a.off_wh1;   // [5][HID]
    const float* bh1 = lds + a.off_bh1;
    const float* wh2 = lds + a.off_wh2;   // [HID][WLD]
    const float* bh2 = lds + a.off_bh2;
    const float* wa = lds + a.off_wa;     // [XD][WLD]
    const float* wr1 = lds + a.off_wr1;   // [12][W1LD], rows 9..11 zero
    const float* br1 = lds + a.off_br1;
    const float* wr2 = lds + a.off_wr2;   // [HID][WLD]
    const float* br2 = lds + a.off_br2;
    float* Xh = lds + a.off_xh;     // [16*NT][XLD]  node-indexed, rows 0 and >= N are zero
    float* Shh = lds + a.off_shh;   // [N][SLD]      node-indexed, columns >= N are -inf
    float* S0 = lds + a.off_s0;     // [16*CT][SLD]  S_c[0][j]
    float* Sc0 = lds + a.off_sc0;   // [16*CT][SLD]  S_c[i][0]
    float* X0 = lds + a.off_x0;     // [16*CT][XLD]  robot embedding of every child
    float* hid = X0;                // [H][HID]      (prologue only; dead before X0 is written)
    float* Gm = lds + a.off_wave;   // [16*NT][XLD]  (prologue + B1/B2 only; the wave-private area is idle until B3)
    float* wbase = lds + a.off_wave + wave * a.wave_stride;
    // wave-private area, by mode:   1: P0w[16][SLD]    2: P0w[G][SLD]    3: Hw[GC][XLD] | Hw2[GC][XLD] | P0w[G][SLD]
    float* Hw = wbase;
    float* Hw2 = wbase + a.GC * XLD;
    float* P0w = MODE == 3 ? wbase + 2 * a.GC * XLD : wbase;
    const float NEG_INF = -INFINITY;

    // ---------------- once per workgroup: weight image ------------------------------------------------
    {
        float* w = lds;
        for (int i = tid; i < 5 * HID; i += nthreads) w[a.off_wh1 + i] = a.wh1[i];
        for (int i = tid; i < HID; i += nthreads) { w[a.off_bh1 + i] = a.bh1[i]; w[a.off_br1 + i] = a.br1[i]; }
        for (int i = tid; i < XD; i += nthreads) { w[a.off_bh2 + i] = a.bh2[i]; w[a.off_br2 + i] = a.br2[i]; }
        for (int i = tid; i < HID * XD; i += nthreads) {
            const int r = i / XD, c = i - r * XD;
            w[a.off_wh2 + r * WLD + c] = a.wh2[i];
            w[a.off_wr2 + r * WLD + c] = a.wr2[i];
        }
        for (int i = tid; i < XD * XD; i += nthreads) {
            const int r = i / XD, c = i - r * XD;
            w[a.off_wa + r * WLD + c] = a.wa ? a.wa[i] : (r == c ? 1.f : 0.f);   // gaussian: Wa = I
        }
        for (int i = tid; i < 12 * HID; i += nthreads) {
            const int r = i / HID, c = i - r * HID;
            w[a.off_wr1 + r * W1LD + c] = r < 9 ? a.wr1[i] : 0.f;
        }
    }
    __syncthreads();

    PHASE_START();
    for (int p = blockIdx.x; p < a.P; p += gridDim.x) {
        PHASE_MARK(0);          // loop overhead / final barrier of the previous parent
        // ---------------- prologue: crowd-only quantities, shared by all children -----------------------
        const float* hsrc = a.humans + (size_t)p * H * 5;
        for (int idx = tid; idx < 16 * a.NT * XLD; idx += nthreads) { Xh[idx] = 0.f; Gm[idx] = 0.f; }
        for (int idx = tid; idx < H * HID; idx += nthreads) {
            const int j = idx / HID, u = idx - j * HID;
            float acc = bh1[u];
#pragma unroll
            for (int k = 0; k < 5; ++k) acc = fmaf(hsrc[j * 5 + k], wh1[k * HID + u], acc);
            hid[idx] = fmaxf(acc, 0.f);
        }
        __syncthreads();
        for (int idx = tid; idx < H * XD; idx += nthreads) {
            const int j = idx / XD, f = idx - j * XD;
            float acc = bh2[f];
#pragma unroll 8
            for (int u = 0; u < HID; ++u) acc = fmaf(hid[j * HID + u], wh2[u * WLD + f], acc);
            Xh[(j + 1) * XLD + f] = fmaxf(acc, 0.f);
        }
        __syncthreads();
        for (int idx = tid; idx < H * XD; idx += nthreads) {
            const int j = idx / XD, g = idx - j * XD;
            float acc = 0.f;
#pragma unroll 8
            for (int f = 0; f < XD; ++f) acc = fmaf(Xh[(j + 1) * XLD + f], wa[f * WLD + g], acc);
            Gm[(j + 1) * XLD + g] = acc;
        }
        __syncthreads();
        for (int idx = tid; idx < N * SLD; idx += nthreads) {
            const int i = idx / SLD, j = idx - i * SLD;
            float v = NEG_INF;
            if (i >= 1 && j >= 1 && j < N) {
                v = 0.f;
#pragma unroll 8
                for (int f = 0; f < XD; ++f) v = fmaf(Gm[i * XLD + f], Xh[j * XLD + f], v);
            }
            Shh[idx] = v;
        }
        // (no barrier needed before B1: it reads Xh/Gm, which were fenced above, and writes S0/Sc0/X0;
        //  X0 aliases `hid`, whose last readers finished before the barrier after the Xh loop)
        PHASE_MARK(1);          // prologue

        // ---------------- B1/B2: robot embedding and robot row/column of S for 16 children per pass -----
        for (int ct = wave; ct < a.CT; ct += a.n_waves) {
            const int c = 16 * ct + n;
            const int cc = c < A ? c : A - 1;
            const float* rr = a.child_robot + ((size_t)p * A + cc) * 9;
            f32x4 hacc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int k = 4 * s + q;
                const float b = k < 9 ? rr[k] : 0.f;
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) hacc[ht] = mfma4(wr1[k * W1LD + 16 * ht + n], b, hacc[ht]);
            }
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&br1[16 * ht + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) hacc[ht][r] = fmaxf(hacc[ht][r] + bb[r], 0.f);
            }
            f32x4 xacc[2] = {zero4(), zero4()};
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot)
                        xacc[ot] = mfma4(wr2[(16 * ht + 4 * q + r) * WLD + 16 * ot + n], hacc[ht][r], xacc[ot]);
            }
            load_fence();
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&br2[16 * ot + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) xacc[ot][r] = fmaxf(xacc[ot][r] + bb[r], 0.f);
                *reinterpret_cast<f32x4*>(&X0[c * XLD + 16 * ot + 4 * q]) = xacc[ot];
            }
            f32x4 gacc[2] = {zero4(), zero4()};
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gt = 0; gt < 2; ++gt)
                        gacc[gt] = mfma4(wa[(16 * ot + 4 * q + r) * WLD + 16 * gt + n], xacc[ot][r], gacc[gt]);
            }
            load_fence();
            float s00 = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) s00 = fmaf(gacc[t][r], xacc[t][r], s00);
            s00 += __shfl_xor(s00, 16);
            s00 += __shfl_xor(s00, 32);
            for (int nt = 0; nt < a.NT; ++nt) {
                load_fence();
                f32x4 sc = zero4(), s0 = zero4();
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const f32x4 gq = *reinterpret_cast<const f32x4*>(&Gm[(16 * nt + n) * XLD + 16 * ot + 4 * q]);
                    const f32x4 xq = *reinterpret_cast<const f32x4*>(&Xh[(16 * nt + n) * XLD + 16 * ot + 4 * q]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sc = mfma4(gq[r], xacc[ot][r], sc);    // S_c[node][0] = G[node] . x0_c
                        s0 = mfma4(xq[r], gacc[ot][r], s0);    // S_c[0][node] = (x0_c Wa) . Xh[node]
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int node = 16 * nt + 4 * q + r;
                    float vs = sc[r], v0 = s0[r];
                    if (node == 0) { vs = s00; v0 = s00; }
                    if (node >= N) { vs = NEG_INF; v0 = NEG_INF; }
                    if (node < SLD) { Sc0[c * SLD + node] = vs; S0[c * SLD + node] = v0; }
                }
            }
            // the row stride covers 4*KS entries; entries past the last node tile are padding too
            for (int k = 16 * a.NT + q; k < SLD; k += 4) { Sc0[c * SLD + k] = NEG_INF; S0[c * SLD + k] = NEG_INF; }
        }
        PHASE_MARK(2);          // B1/B2 work
        __syncthreads();      // Gm is dead from here on: its storage becomes the wave-private area
        PHASE_MARK(3);          // B1/B2 barrier wait

        // ---------------- B3: graph layers, G children per wave at a time --------------------------------
        float xh_a[2][KS];   // A operand of (A_c X): A[i = feature][k <-> node j = 4s+q], shared by every child
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int j = 4 * s + q;
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) xh_a[ft][s] = (j >= 1 && j < N) ? Xh[j * XLD + 16 * ft + n] : 0.f;
        }
        float w_a[2][8];     // W_l[in = 16ft+4q+r][out = 16ot+n]: A operand of W^T*T^T, or B operand of T*W
        if (MODE >= 2) {
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
                    w_a[ot][kk] = a.Ws[0][(16 * (kk >> 2) + 4 * q + (kk & 3)) * XD + 16 * ot + n];
        }
        // the two waves a workgroup places on each SIMD (w, w+4) would run the tile loop in lockstep, colliding on the
        // matrix pipe and idling it together; start the second one about half a tile later
        if (STAGE1_STAGGER > 0 && wave >= 4) __builtin_amdgcn_s_sleep(STAGE1_STAGGER);
        for (int g = wave; g < a.n_groups; g += a.n_waves) {
            const int c0 = g * a.G;
            const int Gv = (A - c0) < a.G ? (A - c0) : a.G;
            const int cols = Gv * a.CPC;
            float* cur = Hw;
            float* nxt = Hw2;
            f32x4 tacc[2] = {zero4(), zero4()};     // MODE 2: t_c accumulators, [child slot 4q+r][feature 16ot+n]
            float run_t[2] = {0.f, 0.f};            // MODE 2 / VAGG: running t_c of child slot run_cl (wave-uniform)
            int run_cl = 0;
            const int n_layers_here = MODE == 3 ? a.L - 1 : 1;
            for (int layer = 0; layer < n_layers_here; ++layer) {
                if (MODE == 3 && (layer >= 1 || g != wave)) {     // more than one full layer: the registers rotate
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk)
                            w_a[ot][kk] = a.Ws[layer][(16 * (kk >> 2) + 4 * q + (kk & 3)) * XD + 16 * ot + n];
                }
                for (int t = 0; t < a.tiles_per_group; ++t) {
                    const int m = 16 * t + n;
                    const bool valid = m < cols;
                    int cl = 0, i = 0;
                    if (MODE == 1) cl = valid ? m : 0;
                    else {
                        cl = div_small(m, a.magicN);
                        i = m - cl * N;
                        if (!valid) { cl = 0; i = 0; }
                    }
                    const int c = c0 + cl;
                    // similarity row of node i of child c, in B-operand order: lane (n,q) holds j = 4s+q
                    const float* rowp = (i == 0) ? &S0[c * SLD] : &Shh[i * SLD];
                    float v[KS];
#pragma unroll
                    for (int s = 0; s < KS; ++s) v[s] = rowp[4 * s + q];
                    if (i > 0 && q == 0) v[0] = Sc0[c * SLD + i];
                    float mx = v[0];
#pragma unroll
                    for (int s = 1; s < KS; ++s) mx = fmaxf(mx, v[s]);
                    mx = kgroups_max(mx);
                    float sum = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        v[s] = __expf(v[s] - mx);
                        sum += v[s];
                    }
                    sum = kgroups_sum(sum);
                    const float inv = valid ? __builtin_amdgcn_rcpf(sum) : 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s) v[s] *= inv;
                    if (layer == 0 && valid && i == 0) {
#pragma unroll
                        for (int s = 0; s < KS; ++s) P0w[cl * SLD + 4 * s + q] = v[s];
                    }
                    if (MODE == 1) continue;
                    // MODE 2 / VAGG: operands of the epilogue, fetched NOW so their LDS latency hides under the MFMAs.
                    // Register r of the swapped product holds column mr = 16t + 4q + r -> (child slot clr, node ir).
                    float e_sel[4], e_sk[2][4];
                    int e_cl[4];
                    if (MODE == 2 && VAGG) {
                        int mr = 16 * t + 4 * q;
                        int clr = div_small(mr, a.magicN);
                        int ir = mr - clr * N;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool vr = mr < cols;
                            const int cs = vr ? clr : 0, is = vr ? ir : 0;
                            const float pv = P0w[cs * SLD + is];                      // A_c[0][node]
                            e_sel[r] = vr ? pv : 0.f;
                            e_cl[r] = clr;
                            if (SKIP) {
                                const int off = (is == 0) ? a.off_x0 + (c0 + cs) * XLD : a.off_xh + is * XLD;
                                e_sk[0][r] = lds[off + n];
                                e_sk[1][r] = lds[off + 16 + n];
                            }
                            ++mr;
                            if (++ir == N) { ir = 0; ++clr; }
                        }
                    }
                    f32x4 acc[2] = {zero4(), zero4()};
                    f32x4 x0c[2];
                    x0c[0] = *reinterpret_cast<const f32x4*>(&X0[c * XLD + 4 * q]);
                    x0c[1] = *reinterpret_cast<const f32x4*>(&X0[c * XLD + 16 + 4 * q]);
                    if (layer == 0) {
                        // (A_c X_c)^T = Xh^T P  +  x0_c (x) P[robot column]
#pragma unroll
                        for (int s = 0; s < KS; ++s)
#pragma unroll
                            for (int ft = 0; ft < 2; ++ft) acc[ft] = mfma4(xh_a[ft][s], v[s], acc[ft]);
                        const float p0 = __shfl(v[0], n);
#pragma unroll
                        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[ft][r] = fmaf(p0, x0c[ft][r], acc[ft][r]);
                    } else {
                        // deeper layers: every child has its own node features -> one masked pass per child in the tile
                        const int first = (16 * t) / a.CPC;
                        int lastc = (16 * t + 15 < cols ? 16 * t + 15 : cols - 1) / a.CPC;
                        for (int cx = first; cx <= lastc; ++cx) {
#pragma unroll
                            for (int s = 0; s < KS; ++s) {
                                const int j = 4 * s + q;
                                const float b = (valid && cl == cx) ? v[s] : 0.f;
#pragma unroll
                                for (int ft = 0; ft < 2; ++ft) {
                                    const float av = j < N ? cur[(cx * N + j) * XLD + 16 * ft + n] : 0.f;
                                    acc[ft] = mfma4(av, b, acc[ft]);
                                }
                            }
                        }
                    }
                    if (MODE == 2 && VAGG) {
                        // H1pre = T * W with T^T's registers as the A operand: the result lands as [column 4q+r][feature 16ot+n]
                        f32x4 o[2] = {zero4(), zero4()};
#pragma unroll
                        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int ot = 0; ot < 2; ++ot) o[ot] = mfma4(acc[ft][r], w_a[ot][4 * ft + r], o[ot]);
                        const int lo = div_small(16 * t, a.magicN);        // slot of the tile's first column (wave-uniform)
                        float plo[2] = {0.f, 0.f}, phi[2] = {0.f, 0.f};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float wlo = (e_cl[r] == lo) ? e_sel[r] : 0.f, whi = (e_cl[r] == lo) ? 0.f : e_sel[r];
#pragma unroll
                            for (int ot = 0; ot < 2; ++ot) {
                                float hval = relu1(o[ot][r]);
                                if (SKIP) hval += e_sk[ot][r];
                                o[ot][r] = hval;
                                plo[ot] = fmaf(wlo, hval, plo[ot]);                            // t_c += A_c[0][node] * H[node]
                                phi[ot] = fmaf(whi, hval, phi[ot]);
                            }
                        }
                        // H_{L-1}[robot] of the children whose robot column (node 0) lies in this tile: wave-uniform walk
                        for (int k = div_small(16 * t + N - 1, a.magicN); k * N < 16 * t + 16 && k < Gv; ++k) {
                            const int ml = k * N - 16 * t, q0 = ml >> 2, r0 = ml & 3;
                            const float h0 = r0 == 0 ? o[0][0] : r0 == 1 ? o[0][1] : r0 == 2 ? o[0][2] : o[0][3];
                            const float h1 = r0 == 0 ? o[1][0] : r0 == 1 ? o[1][1] : r0 == 2 ? o[1][2] : o[1][3];
                            if (q == q0) {
                                float* hp = a.rows_out + ((size_t)p * A + c0 + k) * 64 + 32 + n;
                                hp[0] = h0;
                                hp[16] = h1;
                            }
                        }
                        // children are contiguous column ranges: the tile continues child `lo` and may start `lo+1`
                        if (lo != run_cl) {
                            if (q == 0 && run_cl < Gv) {
                                float* out = a.rows_out + ((size_t)p * A + c0 + run_cl) * 64 + n;
                                out[0] = run_t[0];
                                out[16] = run_t[1];
                            }
                            run_t[0] = run_t[1] = 0.f;
                            run_cl = lo;
                        }
                        run_t[0] += kgroups_sum(plo[0]);
                        run_t[1] += kgroups_sum(plo[1]);
                        if (lo + 1 < Gv && (lo + 1) * N < 16 * t + 16) {
                            if (q == 0) {
                                float* out = a.rows_out + ((size_t)p * A + c0 + run_cl) * 64 + n;
                                out[0] = run_t[0];
                                out[16] = run_t[1];
                            }
                            run_t[0] = kgroups_sum(phi[0]);
                            run_t[1] = kgroups_sum(phi[1]);
                            run_cl = lo + 1;
                        }
                    } else if (MODE == 2) {
                        // general selector form (any N): contraction over the tile's columns as one more MFMA product
                        f32x4 o[2] = {zero4(), zero4()};
#pragma unroll
                        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int ot = 0; ot < 2; ++ot) o[ot] = mfma4(acc[ft][r], w_a[ot][4 * ft + r], o[ot]);
                        int mr = 16 * t + 4 * q;
                        int clr = div_small(mr, a.magicN);
                        int ir = mr - clr * N;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool vr = mr < cols;
                            const int cr = c0 + (vr ? clr : 0);
                            const int irr = vr ? ir : 0;
                            const float* skp = (irr == 0) ? &X0[cr * XLD + n] : &Xh[irr * XLD + n];
                            const float pv = P0w[(vr ? clr : 0) * SLD + irr];
                            const float asel = (vr && clr == n) ? pv : 0.f;                    // selector row of slot n
                            float* hp = a.rows_out + ((size_t)p * A + cr) * 64 + 32 + n;
#pragma unroll
                            for (int ot = 0; ot < 2; ++ot) {
                                float hval = relu1(o[ot][r]);
                                if (SKIP) hval += skp[16 * ot];
                                if (vr && irr == 0) hp[16 * ot] = hval;                        // H_{L-1}[robot] for the skip of the last layer
                                tacc[ot] = mfma4(asel, hval, tacc[ot]);                        // t_c += A_c[0][node] * H[node]
                            }
                            ++mr;
                            if (++ir == N) { ir = 0; ++clr; }
                        }
                    } else {
                        f32x4 o[2] = {zero4(), zero4()};
#pragma unroll
                        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int ot = 0; ot < 2; ++ot) o[ot] = mfma4(w_a[ot][4 * ft + r], acc[ft][r], o[ot]);
                        float* dst = (layer == 0 ? Hw : nxt) + m * XLD;
#pragma unroll
                        for (int ot = 0; ot < 2; ++ot) {
                            f32x4 sk;
                            if (layer == 0) sk = (i == 0) ? x0c[ot] : *reinterpret_cast<const f32x4*>(&Xh[i * XLD + 16 * ot + 4 * q]);
                            else sk = *reinterpret_cast<const f32x4*>(&cur[m * XLD + 16 * ot + 4 * q]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float hval = fmaxf(o[ot][r], 0.f);
                                if (SKIP) hval += sk[r];
                                o[ot][r] = hval;
                            }
                            *reinterpret_cast<f32x4*>(&dst[16 * ot + 4 * q]) = o[ot];
                        }
                    }
                }
                if (layer >= 1) { float* tmp = cur; cur = nxt; nxt = tmp; }
            }
            // last layer, robot node only:  t_c = sum_j A_c[0][j] * H_c[j],   plus H_c[0] for the skip connection
            if (MODE == 2 && VAGG) {
                if (q == 0 && run_cl < Gv) {
                    float* out = a.rows_out + ((size_t)p * A + c0 + run_cl) * 64 + n;
                    out[0] = run_t[0];
                    out[16] = run_t[1];
                }
            } else if (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int slot = 4 * q + r;
                    if (slot < Gv) {
                        float* out = a.rows_out + ((size_t)p * A + c0 + slot) * 64 + n;
                        out[0] = tacc[0][r];
                        out[16] = tacc[1][r];
                    }
                }
            } else {
                for (int cl = lane >> 5; cl < Gv; cl += 2) {
                    const int f = lane & 31;
                    const int c = c0 + cl;
                    float tsum = 0.f, hprev;
                    if (MODE == 3) {
                        for (int j = 0; j < N; ++j) tsum = fmaf(P0w[cl * SLD + j], cur[(cl * N + j) * XLD + f], tsum);
                        hprev = cur[(cl * N) * XLD + f];
                    } else {
                        hprev = X0[c * XLD + f];
                        tsum = P0w[cl * SLD] * hprev;
                        for (int j = 1; j < N; ++j) tsum = fmaf(P0w[cl * SLD + j], Xh[j * XLD + f], tsum);
                    }
                    float* out = a.rows_out + ((size_t)p * A + c) * 64;
                    out[f] = tsum;
                    out[32 + f] = hprev;
                }
            }
        }
        PHASE_MARK(4);          // B3 work
        __syncthreads();
        PHASE_MARK(5);          // end-of-parent barrier wait
    }
    PHASE_FLUSH();
}

// ------------------------------------------------------------------------------------------------
// stage 1, rank-1 form (L == 2, N <= 32): on gfx950 the f32 MFMA and the VALU do not co-execute
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0), so what counts is MFMA cycles PLUS VALU cycles.  Siblings share every
// human row of X and of S except the robot column, hence for a human row i of child c
//     (A_c X_c)_i W1 = ( alpha_i * UW_i + beta_i * (x0_c W1) ) / Z_i ,
// with the crowd-only UW_i = (sum_{j>=1} e^{S_ij - msh_i} Xh_j) W1, Zsh_i = sum_{j>=1} e^{S_ij - msh_i},
// msh_i = max_{j>=1} S_ij, and per child m = max(msh_i, S_c[i][0]), alpha = e^{msh_i - m}, beta = e^{S_c[i][0] - m},
// Z = alpha*Zsh_i + beta (an exactly re-associated, overflow-safe softmax).  Because p >= 0,
// p * relu(x) = relu(p * x), so the robot-row aggregation t_c = sum_i A_c[0][i] H_c[i] folds into the same pass:
// 4 VALU ops per (row, feature) instead of 26 MFMAs per 16 columns.  The robot row itself costs two batched
// MFMA products (T_0 = p X_c, T_0 W1) per 16 children.
//
// Phases per parent (8 waves; waves 0..CT-1 own one 16-child MFMA tile each, waves CT..CT+NT-1 prepare the NEXT
// parent's crowd block meanwhile):   [x0, y = x0 W1, g0 = x0 Wa]  barrier  [robot row/column of S, p = softmax,
// p Xh, (a_i, b_i) table -- all in the MFMA D layout]  barrier  [row phase: all waves, lane = feature, the (a, b) pairs
// arrive as DPP row_newbcast operands]  barrier  [robot row: T_0 W1, relu, t_c, rows out -- registers and own rows only,
// so no barrier before the next parent].
// ------------------------------------------------------------------------------------------------
struct Rank1Args {
    const float *wr1, *br1, *wr2, *br2, *wh1, *bh1, *wh2, *bh2, *wa, *w1;
    const float* child_robot;             // [P][A][9]
    const float* humans;                  // [P][H][5]
    int P, A, H, N, CT, NT, SLD, n_waves;
    int sim;                              // SIM_* row normalisation
    float* rows_out;                      // [P*A][64]
    int off_wh1, off_bh1, off_wh2, off_bh2, off_wa, off_wr1, off_br1, off_wr2, off_br2, off_w1;   // weight image
    int off_crowd, crowd_stride;          // double-buffered crowd block: Xh | Gm | UW | msh | zsh
    int off_sc0, off_y0, off_tp;          // (a, b) table [16*CT][SLD][2], y = x0 W1 [16*CT][XLD], partial t_c [16*CT][XLD]
    int off_flag;                         // [4] ints: crowd-wave epochs
};

// HR >= N: human rows held in registers (padded rows contribute exactly 0); SOFT: softmax row normalisation (else sim)
template <int HR, int NT, bool SKIP, bool SOFT>
__global__ __launch_bounds__(512, 2) void children_rank1_kernel(const Rank1Args a) {
    const int sim = SOFT ? (int)SIM_SOFTMAX : a.sim;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int nthreads = a.n_waves * 64;
    const int n = lane & 15, q = lane >> 4;
    const int N = a.N, H = a.H, A = a.A, SLD = a.SLD;
    const float* wh1 = lds + a.off_wh1;   // [8][W1LD], rows 5..7 zero
    const float* bh1 = lds + a.off_bh1;
    const float* wh2 = lds + a.off_wh2;   // [HID][WLD]
    const float* bh2 = lds + a.off_bh2;
    const float* wa = lds + a.off_wa;     // [XD][WLD]
    const float* wr1 = lds + a.off_wr1;   // [12][W1LD], rows 9..11 zero
    const float* br1 = lds + a.off_br1;
    const float* wr2 = lds + a.off_wr2;
    const float* br2 = lds + a.off_br2;
    const float* w1 = lds + a.off_w1;     // [XD][WLD]
    float* AB = lds + a.off_sc0;          // [16*CT][SLD][2]  per child and row: (a, b) of the rank-1 form, p folded in
    float* Y0 = lds + a.off_y0;           // [16*CT][XLD]  x0 W1, later T_0
    float* TP = lds + a.off_tp;           // [16*CT][XLD]  t_c without the robot-row term
    const float NEG_INF = -INFINITY;
    // crowd block b: Xh[16*NT][XLD] | Gm[16*NT][XLD] | UW[16*NT][XLD] | msh[16*NT] | zsh[16*NT]
    auto crowd_xh = [&](int b) { return lds + a.off_crowd + b * a.crowd_stride; };
    auto crowd_gm = [&](int b) { return lds + a.off_crowd + b * a.crowd_stride + 16 * NT * XLD; };
    auto crowd_uw = [&](int b) { return lds + a.off_crowd + b * a.crowd_stride + 2 * 16 * NT * XLD; };
    auto crowd_msh = [&](int b) { return lds + a.off_crowd + b * a.crowd_stride + 3 * 16 * NT * XLD; };
    auto crowd_zsh = [&](int b) { return lds + a.off_crowd + b * a.crowd_stride + 3 * 16 * NT * XLD + 16 * NT; };

    {   // weight image, once per workgroup
        float* w = lds;
        for (int i = tid; i < 8 * HID; i += nthreads) {
            const int r = i / HID, c = i - r * HID;
            w[a.off_wh1 + r * W1LD + c] = r < 5 ? a.wh1[i] : 0.f;
        }
        for (int i = tid; i < HID; i += nthreads) { w[a.off_bh1 + i] = a.bh1[i]; w[a.off_br1 + i] = a.br1[i]; }
        for (int i = tid; i < XD; i += nthreads) { w[a.off_bh2 + i] = a.bh2[i]; w[a.off_br2 + i] = a.br2[i]; }
        for (int i = tid; i < HID * XD; i += nthreads) {
            const int r = i / XD, c = i - r * XD;
            w[a.off_wh2 + r * WLD + c] = a.wh2[i];
            w[a.off_wr2 + r * WLD + c] = a.wr2[i];
        }
        for (int i = tid; i < XD * XD; i += nthreads) {
            const int r = i / XD, c = i - r * XD;
            w[a.off_wa + r * WLD + c] = a.wa ? a.wa[i] : (r == c ? 1.f : 0.f);   // gaussian: Wa = I
            w[a.off_w1 + r * WLD + c] = a.w1[i];
        }
        for (int i = tid; i < 12 * HID; i += nthreads) {
            const int r = i / HID, c = i - r * HID;
            w[a.off_wr1 + r * W1LD + c] = r < 9 ? a.wr1[i] : 0.f;
        }
    }
    __syncthreads();

    // Wave roles in the embedding phase: waves [0, CT) embed 16 children each; waves [CT, CT+NT) run the crowd-only
    // prologue of the NEXT parent (one 16-node column tile each) into the other crowd buffer.
    const bool child_wave = wave < a.CT;
    const int pct = wave - a.CT;                       // prologue column tile
    const bool crowd_wave = pct >= 0 && pct < NT;
    f32x4 pg[2];                                       // prologue: G^T of my column tile, carried across the mid barrier
    bool node_ok = false;
    int node = 0;

    // crowd prologue, part 1: Xh = w_h(humans), G = Xh Wa   (transposed MFMA chain, 16 nodes per wave)
    auto prologue1 = [&](int pp, int b) {
        float* Xh = crowd_xh(b);
        float* Gm = crowd_gm(b);
        node = 16 * pct + n;
        node_ok = node >= 1 && node < N;
        const float* hsrc = a.humans + ((size_t)pp * H + (node_ok ? node - 1 : 0)) * 5;
        f32x4 hacc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = 4 * s + q;
            const float bv = (node_ok && k < 5) ? hsrc[k] : 0.f;
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) hacc[ht] = mfma4(wh1[k * W1LD + 16 * ht + n], bv, hacc[ht]);
        }
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(&bh1[16 * ht + 4 * q]);
#pragma unroll
            for (int r = 0; r < 4; ++r) hacc[ht][r] = relu1(hacc[ht][r] + bb[r]);
        }
        f32x4 xacc[2] = {zero4(), zero4()};
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            load_fence();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
                    xacc[ot] = mfma4(wh2[(16 * ht + 4 * q + r) * WLD + 16 * ot + n], hacc[ht][r], xacc[ot]);
        }
        load_fence();
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(&bh2[16 * ot + 4 * q]);
#pragma unroll
            for (int r = 0; r < 4; ++r) xacc[ot][r] = node_ok ? relu1(xacc[ot][r] + bb[r]) : 0.f;   // robot slot / padding rows are zero
            *reinterpret_cast<f32x4*>(&Xh[node * XLD + 16 * ot + 4 * q]) = xacc[ot];
        }
        pg[0] = zero4();
        pg[1] = zero4();
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            load_fence();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int gt = 0; gt < 2; ++gt)
                    pg[gt] = mfma4(wa[(16 * ot + 4 * q + r) * WLD + 16 * gt + n], xacc[ot][r], pg[gt]);
        }
        load_fence();
        *reinterpret_cast<f32x4*>(&Gm[node * XLD + 4 * q]) = pg[0];
        *reinterpret_cast<f32x4*>(&Gm[node * XLD + 16 + 4 * q]) = pg[1];
    };
    // part 2 (needs every Xh row): S_ij = G_i . Xh_j over humans j, msh/E/Zsh, U = E Xh, UW = U W1
    auto prologue2 = [&](int b) {
        const float* Xh = crowd_xh(b);
        float* UW = crowd_uw(b);
        f32x4 e[NT];
        float mx = NEG_INF;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            load_fence();
            f32x4 sacc = zero4();
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                const f32x4 xa = *reinterpret_cast<const f32x4*>(&Xh[(16 * jt + n) * XLD + 16 * ft + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc = mfma4(xa[r], pg[ft][r], sacc);      // [j = 16jt+4q+r][i = my node]
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * q + r;
                if (sim != SIM_SOFTMAX) sacc[r] = plain_weight(sim, sacc[r], node, j);
                if (j < 1 || j >= N) sacc[r] = sim == SIM_SOFTMAX ? NEG_INF : 0.f;
                mx = fmaxf(mx, sacc[r]);
            }
            e[jt] = sacc;
        }
        mx = kgroups_max(mx);
        if (!node_ok || sim != SIM_SOFTMAX) mx = 0.f;
        float z = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (sim == SIM_SOFTMAX) e[jt][r] = __expf(e[jt][r] - mx);
                if (!node_ok) e[jt][r] = 0.f;
                z += e[jt][r];
            }
        z = kgroups_sum(z);
        if (q == 0) {
            crowd_msh(b)[node] = mx;
            crowd_zsh(b)[node] = node_ok ? z : 1.f;
        }
        f32x4 u[2] = {zero4(), zero4()};
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            load_fence();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a0 = Xh[(16 * jt + 4 * q + r) * XLD + n];
                const float a1 = Xh[(16 * jt + 4 * q + r) * XLD + 16 + n];
                u[0] = mfma4(a0, e[jt][r], u[0]);                                         // U^T[f][i] = sum_j Xh[j][f] E[i][j]
                u[1] = mfma4(a1, e[jt][r], u[1]);
            }
        }
        f32x4 uw[2] = {zero4(), zero4()};
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            load_fence();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
                    uw[ot] = mfma4(w1[(16 * ft + 4 * q + r) * WLD + 16 * ot + n], u[ft][r], uw[ot]);
        }
        load_fence();
        *reinterpret_cast<f32x4*>(&UW[node * XLD + 4 * q]) = uw[0];
        *reinterpret_cast<f32x4*>(&UW[node * XLD + 16 + 4 * q]) = uw[1];
    };

    PHASE_START();
    int buf = 0;
    int* crowd_flag = reinterpret_cast<int*>(lds + a.off_flag);      // [NT] epoch reached by each crowd wave's part 1
    int crowd_epoch = 0;
    if (tid < NT) crowd_flag[tid] = 0;
    if ((int)blockIdx.x < a.P) {               // prime the pipeline: crowd block of the first parent
        if (crowd_wave) prologue1(blockIdx.x, 0);
        __syncthreads();
        if (crowd_wave) prologue2(0);
        __syncthreads();
    }
    for (int p = blockIdx.x; p < a.P; p += gridDim.x) {
        PHASE_MARK(0);
        const int pn = p + gridDim.x;
        const float* Xh = crowd_xh(buf);
        const float* Gm = crowd_gm(buf);
        // ---------------- embedding phase, first half: x0, y = x0 W1, g0 = x0 Wa  ||  prologue1(next parent) -----
        f32x4 xacc[2] = {zero4(), zero4()}, gacc[2] = {zero4(), zero4()};
        f32x4 t0h[2] = {zero4(), zero4()};        // child waves: (p_c Xh)^T of my 16 children, from embed-2 to the robot-row pass
        float p00 = 0.f;                          // A_c[0][0]
        const int c = 16 * wave + n;              // meaningful for child waves only
        float s00 = 0.f;
        if (child_wave) {
            const int cc = c < A ? c : A - 1;
            const float* rr = a.child_robot + ((size_t)p * A + cc) * 9;
            f32x4 hacc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int k = 4 * s + q;
                const float b = k < 9 ? rr[k] : 0.f;
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) hacc[ht] = mfma4(wr1[k * W1LD + 16 * ht + n], b, hacc[ht]);
            }
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&br1[16 * ht + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) hacc[ht][r] = relu1(hacc[ht][r] + bb[r]);
            }
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot)
                        xacc[ot] = mfma4(wr2[(16 * ht + 4 * q + r) * WLD + 16 * ot + n], hacc[ht][r], xacc[ot]);
            }
            load_fence();
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&br2[16 * ot + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) xacc[ot][r] = relu1(xacc[ot][r] + bb[r]);
            }
            f32x4 yacc[2] = {zero4(), zero4()};
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gt = 0; gt < 2; ++gt) {
                        gacc[gt] = mfma4(wa[(16 * ot + 4 * q + r) * WLD + 16 * gt + n], xacc[ot][r], gacc[gt]);
                        yacc[gt] = mfma4(w1[(16 * ot + 4 * q + r) * WLD + 16 * gt + n], xacc[ot][r], yacc[gt]);
                    }
            }
            load_fence();
            *reinterpret_cast<f32x4*>(&Y0[c * XLD + 4 * q]) = yacc[0];
            *reinterpret_cast<f32x4*>(&Y0[c * XLD + 16 + 4 * q]) = yacc[1];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) s00 = fmaf(gacc[t][r], xacc[t][r], s00);
            s00 = kgroups_sum(s00);
        } else if (crowd_wave && pn < a.P) {
            prologue1(pn, buf ^ 1);
        }
        PHASE_MARK(1);
        // No workgroup barrier here.  The child waves go straight on: their second half needs only their own registers and the
        // CURRENT parent's crowd block, finished an iteration ago.  Only the crowd waves depend on each other (part 2 reads
        // every Xh row of the block part 1 just wrote): they meet on a pair of LDS flags.
        if (NT > 1 && crowd_wave && pn < a.P) {
            ++crowd_epoch;
            if (lane == 0) __hip_atomic_store(&crowd_flag[pct], crowd_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (int w = 0; w < NT; ++w)
                while (__hip_atomic_load(&crowd_flag[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < crowd_epoch)
                    __builtin_amdgcn_s_sleep(2);
        }
        PHASE_MARK(2);
        // ---------------- embedding phase, second half: robot row / column of S  ||  prologue2(next parent) -------
        if (child_wave) {
            // robot row and column of S for my 16 children, then -- still in the MFMA D layout, lane (n, q) = child
            // 16 wave + n, nodes 16 nt + 4 q + r -- p = softmax(robot row) and the per-row scalars (a, b) of the rank-1
            // form with p folded in.  Nothing here crosses lanes except two permlane butterflies per child tile.
            f32x4 s0t[NT], sct[NT];
            float mx0 = NEG_INF;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                load_fence();
                f32x4 sc = zero4(), s0 = zero4();
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const f32x4 gq = *reinterpret_cast<const f32x4*>(&Gm[(16 * nt + n) * XLD + 16 * ot + 4 * q]);
                    const f32x4 xq = *reinterpret_cast<const f32x4*>(&Xh[(16 * nt + n) * XLD + 16 * ot + 4 * q]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sc = mfma4(gq[r], xacc[ot][r], sc);
                        s0 = mfma4(xq[r], gacc[ot][r], s0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nd = 16 * nt + 4 * q + r;
                    if (nd == 0) { sc[r] = s00; s0[r] = s00; }
                    if (sim != SIM_SOFTMAX) s0[r] = plain_weight(sim, s0[r], 0, nd);
                    if (nd >= N) { sc[r] = NEG_INF; s0[r] = sim == SIM_SOFTMAX ? NEG_INF : 0.f; }
                    mx0 = fmaxf(mx0, s0[r]);
                }
                s0t[nt] = s0;
                sct[nt] = sc;
            }
            mx0 = kgroups_max(mx0);
            float z0 = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (sim == SIM_SOFTMAX) s0t[nt][r] = __expf(s0t[nt][r] - mx0);
                    z0 += s0t[nt][r];
                }
            const float iz0 = __builtin_amdgcn_rcpf(kgroups_sum(z0));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s0t[nt][r] *= iz0;                       // p = A_c[0][:] (0 beyond row N-1)
            p00 = kgroups_sum(q == 0 ? s0t[0][0] : 0.f);
            // (p_c Xh)^T[f][c] = sum_j Xh^T[f][j] p_c[j]: the D registers of the robot-row product are already the B operand
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * nt + 4 * q + r;
                    t0h[0] = mfma4(Xh[j * XLD + n], s0t[nt][r], t0h[0]);
                    t0h[1] = mfma4(Xh[j * XLD + 16 + n], s0t[nt][r], t0h[1]);
                }
            }
            const float* mshp = crowd_msh(buf);
            const float* zshp = crowd_zsh(buf);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 ms = *reinterpret_cast<const f32x4*>(&mshp[16 * nt + 4 * q]);
                const f32x4 zs = *reinterpret_cast<const f32x4*>(&zshp[16 * nt + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nd = 16 * nt + 4 * q + r;
                    const float pv = s0t[nt][r];
                    float al, be;
                    if (sim == SIM_SOFTMAX) {
                        const float m = fmaxf(ms[r], sct[nt][r]);
                        al = __expf(ms[r] - m);
                        be = __expf(sct[nt][r] - m);
                    } else {
                        al = 1.f;
                        be = plain_weight(sim, sct[nt][r], nd, 0);        // diagonal: nd >= 1 here, so 0
                    }
                    const float piz = pv * __builtin_amdgcn_rcpf(fmaf(al, zs[r], be));
                    const bool rh = nd >= 1 && nd < N;
                    *reinterpret_cast<f32x2*>(&AB[(c * SLD + nd) * 2]) = f32x2{rh ? al * piz : 0.f, rh ? be * piz : 0.f};
                }
            }
        } else if (crowd_wave && pn < a.P) {
            prologue2(buf ^ 1);
        }
        PHASE_MARK(3);
        __syncthreads();
        PHASE_MARK(4);

        // ---------------- row phase: two children per pass (half-wave each), lane = feature -------------------
        {
            const float* UW = crowd_uw(buf);
            const int hh = lane >> 5, f = lane & 31;
            float uwr[HR];
#pragma unroll
            for (int i = 1; i < HR; ++i) uwr[i] = i < N ? UW[i * XLD + f] : 0.f;
            const int n_pairs = (A + 1) / 2;
            constexpr int HRV = HR < 16 * NT ? HR : 16 * NT;      // the tables hold 16*NT rows per child
            // (a_i, b_i) of my child: ONE b64 read per 16 rows, lane k of every 16-lane DPP row holding row 16*chunk + k;
            // each row's pair then reaches all lanes through row_newbcast operands of the mul / fmac themselves.
            // (Reading the pairs as per-row LDS broadcasts made this loop LDS-bound: 64 lanes x 8 B per row.)
            for (int pair = wave; pair < n_pairs; pair += a.n_waves) {
                const int ch = 2 * pair + hh;
                const bool cv = ch < A;
                const int cc = cv ? ch : A - 1;
                const float* sc_mine = AB + (cc * SLD + (lane & 15)) * 2;
                f32x2 ab[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) ab[t] = *reinterpret_cast<const f32x2*>(&sc_mine[32 * t]);
                const float yv = Y0[cc * XLD + f];
                float rp[4] = {0.f, 0.f, 0.f, 0.f};
                // four rows per step, stage by stage: each row is a mul -> fmac -> max -> add dependency chain, and with two
                // waves per SIMD the chain latency is exposed unless independent rows are interleaved in program order
                static_for<0, (HRV + 2) / 4>([&](auto gc) {
                    constexpr int i0 = 1 + 4 * decltype(gc)::value;
                    float t[4];
                    static_for<0, 4>([&](auto kc) {
                        constexpr int ii = i0 + decltype(kc)::value;
                        if constexpr (ii < HRV) t[ii - i0] = dpp_rowbcast_mul<(ii & 15)>(ab[ii >> 4][1], yv);
                    });
                    static_for<0, 4>([&](auto kc) {
                        constexpr int ii = i0 + decltype(kc)::value;
                        if constexpr (ii < HRV) t[ii - i0] = dpp_rowbcast_fmac<(ii & 15)>(ab[ii >> 4][0], uwr[ii], t[ii - i0]);
                    });
                    static_for<0, 4>([&](auto kc) {
                        constexpr int ii = i0 + decltype(kc)::value;
                        if constexpr (ii < HRV) rp[ii - i0] += relu1(t[ii - i0]);
                    });
                });
                if (cv) TP[ch * XLD + f] = (rp[0] + rp[1]) + (rp[2] + rp[3]);   // t_c without the robot-row / skip terms
            }
        }
        PHASE_MARK(5);
        __syncthreads();
        PHASE_MARK(6);

        // ---------------- robot row: H1_0 = relu(T_0 W1)(+x0), t_c += p00 * H1_0, rows out ---------------------
        if (child_wave) {
            // T_0 = (A_c X_c)[0] = p_c Xh + p_c[0] x0_c  (everything it needs is in this wave's registers)
            f32x4 o[2] = {zero4(), zero4()};
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float tb = fmaf(p00, xacc[ft][r], t0h[ft][r]);
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot)
                        o[ot] = mfma4(w1[(16 * ft + 4 * q + r) * WLD + 16 * ot + n], tb, o[ot]);
                }
            }
            if (c < A) {
                float* out = a.rows_out + ((size_t)p * A + c) * 64;
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const f32x4 tp = *reinterpret_cast<const f32x4*>(&TP[c * XLD + 16 * ot + 4 * q]);
                    f32x4 h, t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float hv = relu1(o[ot][r]);
                        if (SKIP) hv += xacc[ot][r];
                        h[r] = hv;
                        t[r] = fmaf(p00, hv, SKIP ? tp[r] + t0h[ot][r] : tp[r]);
                    }
                    *reinterpret_cast<f32x4*>(out + 16 * ot + 4 * q) = t;
                    *reinterpret_cast<f32x4*>(out + 32 + 16 * ot + 4 * q) = h;
                }
            }
        }
        PHASE_MARK(7);
        // No barrier here: the robot-row pass of parent p and the first embedding half of the next parent touch only rows
        // of the wave's own child tile (TP, Y0) and registers, and the crowd waves write the crowd buffer nobody reads
        // any more; the mid barrier of the next iteration orders everything else.
        buf ^= 1;
    }
    PHASE_FLUSH();
}

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
};

template <int D>
struct Tiles { static constexpr int v = (D + 15) / 16; };

// A-fragment image of W (k-major [IN][OUT]) for the transposed product: fragment (ot, it, r), lane (i = l&15, q):
//   W[in = 16*it + 4q + r][out = 16*ot + i]   (0 outside)
// Feature held by D row (4q + r) of tile t of a D-wide activation.  Full tiles use the plain order 16t + 4q + r.  A partial
// LAST tile uses 16t + 4r + q instead, so that as the next layer's k index its valid features sit in the first
// ceil(valid / 4) k steps and the remaining steps (all-zero padding) are skipped: 25 k steps instead of 28 at D = 100.
template <int D>
__device__ __forceinline__ int tile_feature(int t, int q, int r) {
    constexpr bool partial = (D % 16) != 0;
    return (partial && t == Tiles<D>::v - 1) ? 16 * t + 4 * r + q : 16 * t + 4 * q + r;
}
template <int D>
struct LastTileSteps { static constexpr int v = (D % 16) == 0 ? 4 : ((D % 16) + 3) / 4; };

template <int IN, int OUT>
__device__ __forceinline__ void fill_frags(float* dst, const float* __restrict__ W, int tid, int nthr = kThreads) {
    constexpr int IT = Tiles<IN>::v, OT = Tiles<OUT>::v;
    for (int idx = tid; idx < OT * IT * 4 * 64; idx += nthr) {
        const int l = idx & 63, fr = idx >> 6;
        const int r = fr & 3, it = (fr >> 2) % IT, ot = (fr >> 2) / IT;
        const int m = l & 15;                                        // A-operand row = D row of the output tile
        const int in = tile_feature<IN>(it, l >> 4, r), out = tile_feature<OUT>(ot, m >> 2, m & 3);
        dst[idx] = (in < IN && out < OUT) ? W[in * OUT + out] : 0.f;
    }
}

// per-feature vectors (bias, last-layer weights) in D-row order: dst[16 t + 4 q + r] belongs to tile_feature(t, q, r)
template <int OUT>
__device__ __forceinline__ void fill_bias(float* dst, const float* __restrict__ b, int tid, int nthr = kThreads) {
    for (int idx = tid; idx < Tiles<OUT>::v * 16; idx += nthr) {
        const int feat = tile_feature<OUT>(idx >> 4, (idx >> 2) & 3, idx & 3);
        dst[idx] = feat < OUT ? b[feat] : 0.f;
    }
}

// out = W^T in (+ bias): the accumulators START at the bias (no zero-init moves, no add afterwards).
template <int IN, int OUT>
__device__ __forceinline__ void layer_mfma(const float* frags, const f32x4 (&in)[Tiles<IN>::v], f32x4 (&out)[Tiles<OUT>::v],
                                           int lane, const float* bias = nullptr) {
    constexpr int IT = Tiles<IN>::v, OT = Tiles<OUT>::v;
    const int q = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) out[ot] = bias ? *reinterpret_cast<const f32x4*>(&bias[16 * ot + 4 * q]) : zero4();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        load_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (it == IT - 1 && r >= LastTileSteps<IN>::v) continue;          // k steps over padding only
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) out[ot] = mfma4(frags[((ot * IT + it) * 4 + r) * 64 + lane], in[it][r], out[ot]);
        }
    }
    load_fence();
}

template <int OUT>
__device__ __forceinline__ void relu_tiles(f32x4 (&x)[Tiles<OUT>::v]) {
#pragma unroll
    for (int ot = 0; ot < Tiles<OUT>::v; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[ot][r] = relu1(x[ot][r]);
}

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
    fill_frags<XD, XD>(lds + LO::f_last, a.w_last, tid, kHeadThreads);
    fill_frags<XD, D1>(lds + LO::f1, a.w1, tid, kHeadThreads);
    fill_frags<D1, D2>(lds + LO::f2, a.w2, tid, kHeadThreads);
    fill_frags<D2, D3>(lds + LO::f3, a.w3, tid, kHeadThreads);
    fill_bias<D1>(lds + LO::b1, a.b1, tid, kHeadThreads);
    fill_bias<D2>(lds + LO::b2, a.b2, tid, kHeadThreads);
    fill_bias<D3>(lds + LO::b3, a.b3, tid, kHeadThreads);
    fill_bias<D3>(lds + LO::w4, a.w4, tid, kHeadThreads);      // w4 is [D3][1]: same padded vector layout as a bias
    __syncthreads();
    const float b4 = a.b4[0];
    for (int tile = blockIdx.x * kHeadWaves + wave; tile < a.n_tiles; tile += gridDim.x * kHeadWaves) {
        const int row = 16 * tile + n;
        const int rc = row < a.M ? row : a.M - 1;
        const float* src = a.rows + (size_t)rc * 64;
        f32x4 tin[2], hp[2];
        tin[0] = *reinterpret_cast<const f32x4*>(src + 4 * q);
        tin[1] = *reinterpret_cast<const f32x4*>(src + 16 + 4 * q);
        hp[0] = *reinterpret_cast<const f32x4*>(src + 32 + 4 * q);
        hp[1] = *reinterpret_cast<const f32x4*>(src + 48 + 4 * q);
        f32x4 h[2];
        layer_mfma<XD, XD>(lds + LO::f_last, tin, h, lane);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = relu1(h[ot][r]);
                if (a.skip) x += hp[ot][r];
                h[ot][r] = x;
            }
        f32x4 a1[Tiles<D1>::v];
        layer_mfma<XD, D1>(lds + LO::f1, h, a1, lane, lds + LO::b1);
        relu_tiles<D1>(a1);
        f32x4 a2[Tiles<D2>::v];
        layer_mfma<D1, D2>(lds + LO::f2, a1, a2, lane, lds + LO::b2);
        relu_tiles<D2>(a2);
        f32x4 a3[Tiles<D3>::v];
        layer_mfma<D2, D3>(lds + LO::f3, a2, a3, lane, lds + LO::b3);
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
        if (q == 0 && row < a.M) a.value[row] = v + b4;
    }
}

struct ChildPlan {
    ChildArgs a;
    int ks_bucket;
    size_t lds_bytes;
    bool ok;
};

inline ChildPlan plan_children(const RglGraph& g, int P, int A, int H) {
    ChildPlan pl;
    pl.ok = false;
    if (!fast_path_enabled()) return pl;
    if (fast_similarity_mode(g) != SIM_SOFTMAX || g.layerwise_graph || g.x_dim != XD) return pl;
    if (g.num_layer < 1 || !mlp_is(g.w_r, 9, HID, XD, true) || !mlp_is(g.w_h, 5, HID, XD, true)) return pl;
    const int N = H + 1;
    if (N > 64 || A > 96 || A < 1) return pl;
    ChildArgs& a = pl.a;
    a.N = N;
    a.magicN = (unsigned)((1ull << 32) / (unsigned)N) + 1u;
    const int ks = (N + 3) / 4;
    pl.ks_bucket = ks <= 2 ? 2 : ks <= 5 ? 5 : ks <= 8 ? 8 : ks <= 13 ? 13 : 16;
    a.SLD = 4 * pl.ks_bucket + 1;
    a.NT = (N + 15) / 16;
    a.CT = (A + 15) / 16;
    a.L = g.num_layer;
    a.skip = g.skip_connection;
    a.mode = a.L == 1 ? 1 : (a.L == 2 ? 2 : 3);
    a.CPC = a.L >= 2 ? N : 1;
    // children per group: complete tiles when possible (G*N % 16 == 0), bounded wave-private storage
    int G;
    if (a.CPC == 1) G = 16;
    else {
        int gcd = 16, x = N;
        while (x) { int tmp = gcd % x; gcd = x; x = tmp; }
        G = 16 / gcd;
        if (a.mode == 3)
            while (G > 1 && ((G * N + 15) / 16) * 16 * XLD * 2 > 6144) G = (G + 1) / 2;   // staged layers: <= 24 KiB per wave
    }
    a.G = G;
    a.tiles_per_group = (G * a.CPC + 15) / 16;
    a.GC = a.tiles_per_group * 16;
    a.n_groups = (A + G - 1) / G;
    int off = 0;
    auto take = [&](int nfl) { int o = off; off += (nfl + 3) & ~3; return o; };
    a.off_wh1 = take(5 * HID);
    a.off_bh1 = take(HID);
    a.off_wh2 = take(HID * WLD);
    a.off_bh2 = take(XD);
    a.off_wa = take(XD * WLD);
    a.off_wr1 = take(12 * W1LD);
    a.off_br1 = take(HID);
    a.off_wr2 = take(HID * WLD);
    a.off_br2 = take(XD);
    a.off_xh = take(16 * a.NT * XLD);
    a.off_shh = take(N * a.SLD);
    a.off_s0 = take(16 * a.CT * a.SLD);
    a.off_sc0 = take(16 * a.CT * a.SLD);
    const int x0_floats = 16 * a.CT * XLD, hid_floats = H * HID;
    a.off_x0 = take(x0_floats > hid_floats ? x0_floats : hid_floats);
    int wave_floats;
    if (a.mode == 2) wave_floats = G * a.SLD;
    else if (a.mode == 3) wave_floats = 2 * a.GC * XLD + G * a.SLD;
    else wave_floats = 16 * a.SLD;
    a.wave_stride = (wave_floats + 3) & ~3;
    const int gm_floats = 16 * a.NT * XLD;                       // Gm borrows the (idle) wave-private area
    // waves per workgroup: 8 (two per SIMD; measured better than counts that balance n_groups exactly but load the
    // four SIMDs unevenly), 4 when there is too little work to share
    a.n_waves = a.n_groups >= 6 ? kWaves1 : 4;
    // staged modes carry node features per wave: give up waves (8 -> 4 -> 2) before giving up the MFMA path
    const int off_before_waves = off;
    for (;;) {
        off = off_before_waves;
        const int wave_total = a.n_waves * a.wave_stride > gm_floats ? a.n_waves * a.wave_stride : gm_floats;
        a.off_wave = take(wave_total);
        pl.lds_bytes = (size_t)off * sizeof(float);
        if (pl.lds_bytes <= (size_t)rgl::kLdsBytesPerCu || a.n_waves <= 2) break;
        a.n_waves /= 2;
    }
    if (pl.lds_bytes > (size_t)rgl::kLdsBytesPerCu) return pl;
    a.wr1 = g.w_r.weight[0]; a.br1 = g.w_r.bias[0]; a.wr2 = g.w_r.weight[1]; a.br2 = g.w_r.bias[1];
    a.wh1 = g.w_h.weight[0]; a.bh1 = g.w_h.bias[0]; a.wh2 = g.w_h.weight[1]; a.bh2 = g.w_h.bias[1];
    a.wa = bilinear_wa(g);
    for (int l = 0; l < RGL_MAX_GCN_LAYERS; ++l) a.Ws[l] = l < g.num_layer ? g.Ws[l] : nullptr;
    a.P = P; a.A = A; a.H = H;
    pl.ok = true;
    return pl;
}

template <int KS, int MODE, bool VAGG, bool SKIP>
int launch_children_skip(const ChildPlan& pl, hipStream_t st) {
    auto kern = children_graph_kernel<KS, MODE, VAGG, SKIP>;
    if (pl.lds_bytes > 64 * 1024)
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)pl.lds_bytes));
    const int per_cu = pl.lds_bytes * 2 <= (size_t)rgl::kLdsBytesPerCu ? 2 : 1;
    const int grid = pl.a.P < 256 * per_cu ? pl.a.P : 256 * per_cu;      // persistent: the weight image is built once
    hipLaunchKernelGGL(kern, dim3(grid), dim3(pl.a.n_waves * 64), pl.lds_bytes, st, pl.a);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

template <int KS, int MODE, bool VAGG>
int launch_children_mode(const ChildPlan& pl, hipStream_t st) {
    return pl.a.skip ? launch_children_skip<KS, MODE, VAGG, true>(pl, st) : launch_children_skip<KS, MODE, VAGG, false>(pl, st);
}

template <int KS>
int launch_children(const ChildPlan& pl, hipStream_t st) {
    switch (pl.a.mode) {
        case 1: return launch_children_mode<KS, 1, false>(pl, st);
        case 2: return pl.a.N >= 16 ? launch_children_mode<KS, 2, true>(pl, st) : launch_children_mode<KS, 2, false>(pl, st);
        default: return launch_children_mode<KS, 3, false>(pl, st);
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kHeadThreads), lds_bytes, st, ha);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}


// ------------------------------------------------------------------------------------------------
// state-predictor path: scenes with their own crowds (one graph forward per tree node)
//   row_mlp2_kernel   : batched 2-layer embedding MLP over rows (IN -> 64 -> 32, ReLU after both) as an MFMA chain
//   scene_graph_kernel: one wave per scene: S = (X Wa) X^T, softmax, L x relu(A H W)(+H), motion head 32->64->5
// ------------------------------------------------------------------------------------------------
struct RowMlpArgs {
    const float *w1, *b1, *w2, *b2;   // k-major [IN][64], [64], [64][32], [32]
    const float* rows;                // [M][IN]
    float* out;                       // [M][32]
    int M, n_tiles;
};

template <int IN>
__global__ __launch_bounds__(kThreads, 2) void row_mlp2_kernel(const RowMlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int F1 = 0, F2 = F1 + 4 * 1 * 4 * 64, B1 = F2 + 2 * 4 * 4 * 64, B2 = B1 + HID;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    fill_frags<IN, HID>(lds + F1, a.w1, tid);
    fill_frags<HID, XD>(lds + F2, a.w2, tid);
    fill_bias<HID>(lds + B1, a.b1, tid);
    fill_bias<XD>(lds + B2, a.b2, tid);
    __syncthreads();
    for (int tile = blockIdx.x * kWaves + wave; tile < a.n_tiles; tile += gridDim.x * kWaves) {
        const int row = 16 * tile + n;
        const int rc = row < a.M ? row : a.M - 1;
        const float* src = a.rows + (size_t)rc * IN;
        f32x4 in[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int feat = tile_feature<IN>(0, q, r);
            in[0][r] = feat < IN ? src[feat] : 0.f;
        }
        f32x4 h[4];
        layer_mfma<IN, HID>(lds + F1, in, h, lane, lds + B1);
        relu_tiles<HID>(h);
        f32x4 o[2];
        layer_mfma<HID, XD>(lds + F2, h, o, lane, lds + B2);
        relu_tiles<XD>(o);
        if (row < a.M) {
            float* dst = a.out + (size_t)row * XD;
            *reinterpret_cast<f32x4*>(dst + 4 * q) = o[0];
            *reinterpret_cast<f32x4*>(dst + 16 + 4 * q) = o[1];
        }
    }
}

struct SceneArgs {
    const float* xh_rows;              // [n_crowds][H][32]  human embeddings
    const float* x0_rows;              // [P][32]            robot embeddings
    int crowds_per;                    // scene s uses crowd s / crowds_per
    const float* wa;                   // [32][32]
    const float* Ws[RGL_MAX_GCN_LAYERS];
    int L, skip;
    int sim;                           // SIM_* row normalisation
    const float *wm1, *bm1, *wm2, *bm2;   // motion head, k-major [32][64], [64], [64][5], [5]
    float* humans_next;                // [P][H][5]
    int P, H, N;
    int off_wa, off_ws, off_wm1, off_bm1, off_wm2, off_bm2, off_wave, wave_stride;
};

constexpr int M2LD = 20;   // LDS row stride of the [64][5 -> 16] motion output layer (4*M2LD % 32 == 16)

template <int NT, bool SOFT>
__global__ __launch_bounds__(kThreads, 2) void scene_graph_kernel(const SceneArgs a) {
    const int sim = SOFT ? (int)SIM_SOFTMAX : a.sim;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int N = a.N, H = a.H;
    const float* wa = lds + a.off_wa;       // [32][WLD]
    const float* ws = lds + a.off_ws;       // [L][32][WLD]
    const float* wm1 = lds + a.off_wm1;     // [32][W1LD]
    const float* bm1 = lds + a.off_bm1;     // [64]
    const float* wm2 = lds + a.off_wm2;     // [64][M2LD], columns >= 5 zero
    const float* bm2 = lds + a.off_bm2;     // [16], entries >= 5 zero
    float* Hs = lds + a.off_wave + wave * a.wave_stride;   // [16*NT][XLD] node features of the wave's current scene
    {
        float* w = lds;
        for (int i = tid; i < XD * XD; i += kThreads) {
            const int r = i / XD, c = i - r * XD;
            w[a.off_wa + r * WLD + c] = a.wa ? a.wa[i] : (r == c ? 1.f : 0.f);   // gaussian: Wa = I
            for (int l = 0; l < a.L; ++l) w[a.off_ws + (l * XD + r) * WLD + c] = a.Ws[l][i];
        }
        for (int i = tid; i < XD * HID; i += kThreads) {
            const int r = i / HID, c = i - r * HID;
            w[a.off_wm1 + r * W1LD + c] = a.wm1[i];
        }
        for (int i = tid; i < HID * 16; i += kThreads) {
            const int r = i / 16, c = i - r * 16;
            w[a.off_wm2 + r * M2LD + c] = c < 5 ? a.wm2[r * 5 + c] : 0.f;
        }
        for (int i = tid; i < HID; i += kThreads) w[a.off_bm1 + i] = a.bm1[i];
        for (int i = tid; i < 16; i += kThreads) w[a.off_bm2 + i] = i < 5 ? a.bm2[i] : 0.f;
    }
    __syncthreads();
    for (int sc = blockIdx.x * kWaves + wave; sc < a.P; sc += gridDim.x * kWaves) {
        // node features of this scene: row 0 = robot, rows 1..H = its crowd, rows >= N zero
        const float* xr = a.x0_rows + (size_t)sc * XD;
        const float* xh = a.xh_rows + (size_t)(sc / a.crowds_per) * H * XD;
        for (int idx = lane; idx < 16 * NT * (XD / 4); idx += 64) {
            const int row = idx >> 3, c4 = (idx & 7) * 4;
            f32x4 val = zero4();
            if (row == 0) val = *reinterpret_cast<const f32x4*>(xr + c4);
            else if (row < N) val = *reinterpret_cast<const f32x4*>(xh + (size_t)(row - 1) * XD + c4);
            *reinterpret_cast<f32x4*>(&Hs[row * XLD + c4]) = val;
        }
        // G^T = Wa^T X^T   (per column tile: [g = 16gt+4q+r][col n])
        f32x4 gt_[NT][2];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            gt_[ct][0] = zero4();
            gt_[ct][1] = zero4();
            load_fence();
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                const f32x4 xb = *reinterpret_cast<const f32x4*>(&Hs[(16 * ct + n) * XLD + 16 * ft + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        gt_[ct][g] = mfma4(wa[(16 * ft + 4 * q + r) * WLD + 16 * g + n], xb[r], gt_[ct][g]);
            }
        }
        // S^T[j][col] = X[j] . G[col]  -> softmax over j, kept in B-operand order (k <-> j = 16jt+4q+r)
        f32x4 pr[NT][NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            float mx = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                load_fence();
                f32x4 sacc = zero4();
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    const f32x4 xa = *reinterpret_cast<const f32x4*>(&Hs[(16 * jt + n) * XLD + 16 * ft + 4 * q]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc = mfma4(xa[r], gt_[ct][ft][r], sacc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * jt + 4 * q + r;
                    if (sim != SIM_SOFTMAX) sacc[r] = plain_weight(sim, sacc[r], 16 * ct + n, j);
                    if (j >= N) sacc[r] = sim == SIM_SOFTMAX ? -INFINITY : 0.f;
                    mx = fmaxf(mx, sacc[r]);
                }
                pr[ct][jt] = sacc;
            }
            mx = kgroups_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (sim == SIM_SOFTMAX) pr[ct][jt][r] = __expf(pr[ct][jt][r] - mx);
                    sum += pr[ct][jt][r];
                }
            sum = kgroups_sum(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[ct][jt][r] *= inv;
        }
        // layers: H <- relu((A H) W_l) (+ H); every column tile's A*H is taken before any row is overwritten
        for (int l = 0; l < a.L; ++l) {
            f32x4 acc[NT][2];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                acc[ct][0] = zero4();
                acc[ct][1] = zero4();
            }
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a0 = Hs[(16 * jt + 4 * q + r) * XLD + n];
                    const float a1 = Hs[(16 * jt + 4 * q + r) * XLD + 16 + n];
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) {
                        acc[ct][0] = mfma4(a0, pr[ct][jt][r], acc[ct][0]);
                        acc[ct][1] = mfma4(a1, pr[ct][jt][r], acc[ct][1]);
                    }
                }
            }
            const float* wl = ws + l * XD * WLD;
            const bool last = (l == a.L - 1);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                load_fence();
                f32x4 o[2] = {zero4(), zero4()};
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int ot = 0; ot < 2; ++ot)
                            o[ot] = mfma4(wl[(16 * ft + 4 * q + r) * WLD + 16 * ot + n], acc[ct][ft][r], o[ot]);
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const f32x4 sk = *reinterpret_cast<const f32x4*>(&Hs[(16 * ct + n) * XLD + 16 * ot + 4 * q]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float hv = fmaxf(o[ot][r], 0.f);
                        if (a.skip) hv += sk[r];
                        o[ot][r] = hv;
                    }
                    if (!last) *reinterpret_cast<f32x4*>(&Hs[(16 * ct + n) * XLD + 16 * ot + 4 * q]) = o[ot];
                }
                if (last) {
                    // motion head on this tile's columns, straight from registers: 32 -> 64 (ReLU) -> 5
                    f32x4 hm[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot) {
                        load_fence();
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int ht = 0; ht < 4; ++ht)
                                hm[ht] = mfma4(wm1[(16 * ot + 4 * q + r) * W1LD + 16 * ht + n], o[ot][r], hm[ht]);
                    }
                    f32x4 om = zero4();
#pragma unroll
                    for (int ht = 0; ht < 4; ++ht) {
                        load_fence();
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(&bm1[16 * ht + 4 * q]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float hv = fmaxf(hm[ht][r] + bb[r], 0.f);
                            om = mfma4(wm2[(16 * ht + 4 * q + r) * M2LD + n], hv, om);
                        }
                    }
                    const int node = 16 * ct + n;
                    if (node >= 1 && node < N) {
                        float* dst = a.humans_next + ((size_t)sc * H + (node - 1)) * 5;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int oidx = 4 * q + r;
                            if (oidx < 5) dst[oidx] = om[r] + bm2[oidx];
                        }
                    }
                }
            }
        }
    }
}

template <int IN>
int launch_row_mlp2(const RglMlp& m, const float* rows, float* out, int M, hipStream_t st) {
    RowMlpArgs ra;
    ra.w1 = m.weight[0]; ra.b1 = m.bias[0]; ra.w2 = m.weight[1]; ra.b2 = m.bias[1];
    ra.rows = rows; ra.out = out; ra.M = M; ra.n_tiles = (M + 15) / 16;
    const size_t lds_bytes = (size_t)(4 * 4 * 64 + 2 * 4 * 4 * 64 + HID + XD) * sizeof(float);
    int grid = (ra.n_tiles + kWaves - 1) / kWaves;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(row_mlp2_kernel<IN>, dim3(grid), dim3(kThreads), lds_bytes, st, ra);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

template <int NT>
int launch_scene(const SceneArgs& sa, size_t lds_bytes, hipStream_t st) {
    auto kern = sa.sim == SIM_SOFTMAX ? scene_graph_kernel<NT, true> : scene_graph_kernel<NT, false>;
    if (lds_bytes > 64 * 1024)
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds_bytes));
    int grid = (sa.P + kWaves - 1) / kWaves;
    const int cap = 256 * (lds_bytes * 2 <= (size_t)rgl::kLdsBytesPerCu ? 2 : 1) * 2;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds_bytes, st, sa);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}


struct Rank1Plan {
    Rank1Args a;
    size_t lds_bytes;
    int hr;
    bool ok;
};

inline Rank1Plan plan_rank1(const RglGraph& g, int P, int A, int H) {
    Rank1Plan pl;
    pl.ok = false;
    if (!fast_path_enabled() || !rank1_enabled()) return pl;
    if (fast_similarity_mode(g) < 0 || g.layerwise_graph || g.x_dim != XD || g.num_layer != 2) return pl;
    if (!mlp_is(g.w_r, 9, HID, XD, true) || !mlp_is(g.w_h, 5, HID, XD, true)) return pl;
    const int N = H + 1;
    if (N > 32 || A > 96 || A < 1) return pl;
    Rank1Args& a = pl.a;
    a.N = N; a.H = H; a.A = A; a.P = P;
    pl.hr = N <= 8 ? 8 : (N <= 20 ? 20 : 32);
    a.SLD = 16 * ((N + 15) / 16) + 1;           // rows padded to whole MFMA tiles (unconditional access), odd stride
    a.NT = (N + 15) / 16;
    a.CT = (A + 15) / 16;
    a.n_waves = 8;                              // CT (<= 6) child waves + NT (<= 2) crowd waves
    int off = 0;
    auto take = [&](int nfl) { int o = off; off += (nfl + 3) & ~3; return o; };
    a.off_wh1 = take(8 * W1LD); a.off_bh1 = take(HID); a.off_wh2 = take(HID * WLD); a.off_bh2 = take(XD);
    a.off_wa = take(XD * WLD); a.off_wr1 = take(12 * W1LD); a.off_br1 = take(HID); a.off_wr2 = take(HID * WLD);
    a.off_br2 = take(XD); a.off_w1 = take(XD * WLD);
    a.crowd_stride = 3 * 16 * a.NT * XLD + 2 * 16 * a.NT;
    a.off_crowd = take(2 * a.crowd_stride);
    a.off_sc0 = take(2 * 16 * a.CT * a.SLD);
    a.off_y0 = take(16 * a.CT * XLD);
    a.off_tp = take(16 * a.CT * XLD);
    a.off_flag = take(4);
    pl.lds_bytes = (size_t)off * sizeof(float);
    if (pl.lds_bytes > (size_t)rgl::kLdsBytesPerCu) return pl;
    a.wr1 = g.w_r.weight[0]; a.br1 = g.w_r.bias[0]; a.wr2 = g.w_r.weight[1]; a.br2 = g.w_r.bias[1];
    a.wh1 = g.w_h.weight[0]; a.bh1 = g.w_h.bias[0]; a.wh2 = g.w_h.weight[1]; a.bh2 = g.w_h.bias[1];
    a.wa = bilinear_wa(g); a.w1 = g.Ws[0];
    a.sim = fast_similarity_mode(g);
    pl.ok = true;
    return pl;
}

template <int HR, int NT, bool SKIP, bool SOFT>
int launch_rank1_ts(const Rank1Plan& pl, hipStream_t st) {
    auto kern = children_rank1_kernel<HR, NT, SKIP, SOFT>;
    if (pl.lds_bytes > 64 * 1024)
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)pl.lds_bytes));
    const int grid = pl.a.P < 256 ? pl.a.P : 256;          // persistent: one 16-wave workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(pl.a.n_waves * 64), pl.lds_bytes, st, pl.a);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

template <int HR, int NT, bool SKIP>
int launch_rank1_t(const Rank1Plan& pl, hipStream_t st) {
    return pl.a.sim == SIM_SOFTMAX ? launch_rank1_ts<HR, NT, SKIP, true>(pl, st) : launch_rank1_ts<HR, NT, SKIP, false>(pl, st);
}

inline int launch_rank1(const Rank1Plan& pl, bool skip, hipStream_t st) {
    switch (pl.hr) {
        case 8: return skip ? launch_rank1_t<8, 1, true>(pl, st) : launch_rank1_t<8, 1, false>(pl, st);
        case 20: return pl.a.NT == 1 ? (skip ? launch_rank1_t<20, 1, true>(pl, st) : launch_rank1_t<20, 1, false>(pl, st))
                                     : (skip ? launch_rank1_t<20, 2, true>(pl, st) : launch_rank1_t<20, 2, false>(pl, st));
        default: return skip ? launch_rank1_t<32, 2, true>(pl, st) : launch_rank1_t<32, 2, false>(pl, st);
    }
}

}  // namespace

#ifdef RGL_PHASE_TIMING
extern "C" int rgl_debug_read_phase_cycles(unsigned long long* out16, int reset) {
    RGL_HIP_TRY(hipDeviceSynchronize());
    RGL_HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {0};
        RGL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)));
    }
    return 0;
}
#endif

namespace rgl {

size_t value_children_workspace_bytes(const MprlPlanner* pl, int P, int H) {
    const size_t children = (size_t)P * pl->num_actions * 64 * sizeof(float);
    const size_t predictor = (size_t)P * (H + 1) * XD * sizeof(float);         // embeddings of launch_predict_humans
    return children > predictor ? children : predictor;
}

// humans_next[s] = motion_head(RGL(robot[s], humans[s / crowds_per]))[1:]  for P scenes (StatePredictor.forward).
int launch_predict_humans(const MprlPlanner* pl, const float* robot, const float* humans, int crowds_per, int P, int H,
                          float* humans_next, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const RglGraph& g = pl->predictor_graph;
    const RglMlp& mh = pl->motion_head;
    const int N = H + 1;
    const bool ok = fast_path_enabled() && fast_similarity_mode(g) >= 0 && !g.layerwise_graph && g.x_dim == XD &&
                    g.num_layer >= 1 && g.num_layer <= 4 && mlp_is(g.w_r, 9, HID, XD, true) && mlp_is(g.w_h, 5, HID, XD, true) &&
                    mlp_is(mh, XD, HID, 5, false) && N <= 64 && workspace &&
                    workspace_bytes >= (size_t)P * N * XD * sizeof(float) && P % crowds_per == 0;
    if (!ok)
        return launch_generic_forward(&g, nullptr, &mh, robot, humans, P, crowds_per, H, nullptr, nullptr, nullptr,
                                      humans_next, stream);
    const int n_crowds = P / crowds_per;
    float* x0_rows = (float*)workspace;                      // [P][32]
    float* xh_rows = x0_rows + (size_t)P * XD;               // [n_crowds][H][32]
    int rc = launch_row_mlp2<9>(g.w_r, robot, x0_rows, P, stream);
    if (rc) return rc;
    rc = launch_row_mlp2<5>(g.w_h, humans, xh_rows, n_crowds * H, stream);
    if (rc) return rc;
    SceneArgs sa;
    sa.xh_rows = xh_rows; sa.x0_rows = x0_rows; sa.crowds_per = crowds_per;
    sa.wa = bilinear_wa(g);
    sa.sim = fast_similarity_mode(g);
    for (int l = 0; l < RGL_MAX_GCN_LAYERS; ++l) sa.Ws[l] = l < g.num_layer ? g.Ws[l] : nullptr;
    sa.L = g.num_layer; sa.skip = g.skip_connection;
    sa.wm1 = mh.weight[0]; sa.bm1 = mh.bias[0]; sa.wm2 = mh.weight[1]; sa.bm2 = mh.bias[1];
    sa.humans_next = humans_next;
    sa.P = P; sa.H = H; sa.N = N;
    const int NT = (N + 15) / 16;
    int off = 0;
    auto take = [&](int nfl) { int o = off; off += (nfl + 3) & ~3; return o; };
    sa.off_wa = take(XD * WLD);
    sa.off_ws = take(g.num_layer * XD * WLD);
    sa.off_wm1 = take(XD * W1LD);
    sa.off_bm1 = take(HID);
    sa.off_wm2 = take(HID * M2LD);
    sa.off_bm2 = take(16);
    sa.wave_stride = 16 * NT * XLD;
    sa.off_wave = take(kWaves * sa.wave_stride);
    const size_t lds_bytes = (size_t)off * sizeof(float);
    switch (NT) {
        case 1: return launch_scene<1>(sa, lds_bytes, stream);
        case 2: return launch_scene<2>(sa, lds_bytes, stream);
        case 3: return launch_scene<3>(sa, lds_bytes, stream);
        default: return launch_scene<4>(sa, lds_bytes, stream);
    }
}

// V(child) for the A children of each of P parents; children of one parent share humans_next[p].
int launch_value_children(const MprlPlanner* pl, const float* child_robot, const float* humans_next, int P, int H,
                          float* child_value, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const int A = pl->num_actions;
    const int hv = head_variant(pl->value_head);
    const bool want_f16 = pl->contraction_dtype == RGL_CONTRACT_F16;
    if (pl->contraction_dtype != RGL_CONTRACT_F32 && !want_f16) return RGL_ERR_BAD_MODE;
    const bool staged = hv >= 0 && workspace && workspace_bytes >= value_children_workspace_bytes(pl, P, H);
    // stage 1, in order of preference: rank-1 (L = 2, N <= 32), shared-crowd deep (L in {2,3}, N <= 60), tiles (softmax
    // similarities, any depth, N <= 64); everything else, or a head without a stage-2 kernel: the general kernel
    int rc = 1;                                            // 1 = no stage-1 kernel launched yet
    if (staged) {
        Rank1Plan rp = plan_rank1(pl->value_graph, P, A, H);
        if (rp.ok && !want_f16) {
            rp.a.child_robot = child_robot;
            rp.a.humans = humans_next;
            rp.a.rows_out = (float*)workspace;
            rc = launch_rank1(rp, pl->value_graph.skip_connection != 0, stream);
        } else {
            rc = launch_deep_children(&pl->value_graph, P, A, H, child_robot, humans_next, (float*)workspace,
                                      want_f16 && pl->value_graph.num_layer == 3, stream);
            if (want_f16 && (rc == 1 || pl->value_graph.num_layer != 3)) return RGL_ERR_BAD_MODE;
        }
        if (rc == 1) {
            ChildPlan cp = plan_children(pl->value_graph, P, A, H);
            if (cp.ok) {
                cp.a.child_robot = child_robot;
                cp.a.humans = humans_next;
                cp.a.rows_out = (float*)workspace;
                switch (cp.ks_bucket) {
                    case 2: rc = launch_children<2>(cp, stream); break;
                    case 5: rc = launch_children<5>(cp, stream); break;
                    case 8: rc = launch_children<8>(cp, stream); break;
                    case 13: rc = launch_children<13>(cp, stream); break;
                    default: rc = launch_children<16>(cp, stream); break;
                }
            }
        }
    }
    if (want_f16 && rc == 1) return RGL_ERR_BAD_MODE;
    if (rc == 1)
        return launch_generic_forward(&pl->value_graph, &pl->value_head, nullptr, child_robot, humans_next, P * A, A, H,
                                      nullptr, nullptr, child_value, nullptr, stream);
    if (rc) return rc;
    HeadArgs ha;
    const RglGraph& g = pl->value_graph;
    const RglMlp& h = pl->value_head;
    ha.w_last = g.Ws[g.num_layer - 1];
    ha.w1 = h.weight[0]; ha.b1 = h.bias[0];
    ha.w2 = h.weight[1]; ha.b2 = h.bias[1];
    ha.w3 = h.weight[2]; ha.b3 = h.bias[2];
    ha.w4 = h.weight[3]; ha.b4 = h.bias[3];
    ha.skip = g.skip_connection;
    ha.rows = (const float*)workspace;
    ha.value = child_value;
    ha.M = P * A;
    ha.n_tiles = (ha.M + 15) / 16;
    return hv == 0 ? launch_head<32, 100, 100>(ha, stream) : launch_head<150, 100, 100>(ha, stream);
}

}  // namespace rgl
