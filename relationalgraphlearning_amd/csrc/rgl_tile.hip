// rgl_tile.hip -- stage 1 of "value of the sibling children", tile form: every child's graph is evaluated in full on
// the MFMA, 16 (child, node) columns at a time; covers any depth and N <= 64 for the softmax similarities.  It is the
// fallback behind the shared-crowd kernels (rgl_rank1.hip: L = 2, N <= 32; rgl_deep.hip: L in {2, 3}, N <= 60) and can be
// forced with RGL_CHILDREN_TILE_KERNEL=1.
//
//   prologue  human embeddings Xh, G = Xh*Wa, S_hh = G*Xh^T            (once per parent, VALU)
//   B1/B2     robot embeddings of 16 children at a time as an MFMA chain in "transposed" form
//             (activations = B operand, kept in registers: the 4 D registers of one MFMA are
//             the B operands of 4 k-steps of the next, with the k index permuted to match),
//             then the robot row / robot column of every child's similarity matrix
//   B3        per 16 (child,node) columns: softmax computed in-lane directly in the MFMA
//             B-operand layout, A*X as MFMA with the SHARED human rows as A operand plus a
//             rank-1 update for the per-child robot row, *W by MFMA with W in registers,
//             relu (+skip); node features staged in wave-private LDS; the last layer needs
//             only the robot node: t_c = A_c[0,:] * H_c  (wave-level reduction)
//
// Follows (reference paths): crowd_nav/policy/graph_model.py:99-130, model_predictive_rl.py:245-250.
#include "rgl_mfma.h"

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
#pragma unroll 4
        for (int i = tid; i < 5 * HID; i += nthreads) w[a.off_wh1 + i] = a.wh1[i];
        for (int i = tid; i < HID; i += nthreads) { w[a.off_bh1 + i] = a.bh1[i]; w[a.off_br1 + i] = a.br1[i]; }
        for (int i = tid; i < XD; i += nthreads) { w[a.off_bh2 + i] = a.bh2[i]; w[a.off_br2 + i] = a.br2[i]; }
#pragma unroll 4
        for (int i = tid; i < HID * XD; i += nthreads) {
            const int r = i / XD, c = i - r * XD;
            w[a.off_wh2 + r * WLD + c] = a.wh2[i];
            w[a.off_wr2 + r * WLD + c] = a.wr2[i];
        }
#pragma unroll 4
        for (int i = tid; i < XD * XD; i += nthreads) {
            const int r = i / XD, c = i - r * XD;
            w[a.off_wa + r * WLD + c] = a.wa ? a.wa[i] : (r == c ? 1.f : 0.f);   // gaussian: Wa = I
        }
#pragma unroll 4
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

}  // namespace

namespace rgl {

// 1 = outside this kernel's envelope
int launch_tile_children(const RglGraph* g, int P, int A, int H, const float* child_robot, const float* humans_next,
                         float* rows_out, hipStream_t stream) {
    ChildPlan cp = plan_children(*g, P, A, H);
    if (!cp.ok) return 1;
    cp.a.child_robot = child_robot;
    cp.a.humans = humans_next;
    cp.a.rows_out = rows_out;
    switch (cp.ks_bucket) {
        case 2: return launch_children<2>(cp, stream);
        case 5: return launch_children<5>(cp, stream);
        case 8: return launch_children<8>(cp, stream);
        case 13: return launch_children<13>(cp, stream);
        default: return launch_children<16>(cp, stream);
    }
}

}  // namespace rgl
