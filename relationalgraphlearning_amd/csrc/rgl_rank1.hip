// rgl_rank1.hip -- stage 1 of "value of the sibling children", rank-1 (shared-crowd) form for the shipped shape:
// L = 2, N <= 32.  The dominant kernel of the headline workload.
//
// Follows (reference paths): crowd_nav/policy/graph_model.py:99-130, value_estimator.py:11-20,
// model_predictive_rl.py:245-250 (the loop whose iterations this kernel runs side by side).
#include "rgl_mlp_chain.h"

namespace {

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
    const float* image;                   // null, or the packed weight image (FusedLds layout, rgl_mlp_chain.h): its first FusedLds::b1 floats
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

    if (a.image) {
        // the caller has the packed image of these weights (once per parameter state): straight b128 copies, no index arithmetic
        copy_image<FusedLds<32, 100, 100>::b1, 512>(lds, a.image, tid);
    } else {
        // weight image, once per workgroup, in two phases -- every global load of the thread first, then the LDS stores -- so the
        // 34 KB image costs ONE L2 round trip (matrix by matrix it cost one per matrix: several us of a small launch)
        float* w = lds;
        constexpr int NTHR = 512;                      // the kernel is always launched with 8 waves
        constexpr int K2 = HID * XD / NTHR, KQ = XD * XD / NTHR;
        float v_wh2[K2], v_wr2[K2], v_wa[KQ], v_w1[KQ], v_wh1, v_wr1[2], v_b[4];
#pragma unroll
        for (int k = 0; k < K2; ++k) { v_wh2[k] = a.wh2[tid + k * NTHR]; v_wr2[k] = a.wr2[tid + k * NTHR]; }
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int i = tid + k * NTHR;
            v_wa[k] = a.wa ? a.wa[i] : ((i / XD) == (i % XD) ? 1.f : 0.f);       // gaussian: Wa = I
            v_w1[k] = a.w1[i];
        }
        v_wh1 = tid < 5 * HID ? a.wh1[tid] : 0.f;                                // [8][W1LD], rows 5..7 zero
#pragma unroll
        for (int k = 0; k < 2; ++k) { const int i = tid + k * NTHR; v_wr1[k] = i < 9 * HID ? a.wr1[i] : 0.f; }   // [12][W1LD], rows 9..11 zero
        v_b[0] = tid < HID ? a.bh1[tid] : 0.f; v_b[1] = tid < HID ? a.br1[tid] : 0.f;
        v_b[2] = tid < XD ? a.bh2[tid] : 0.f;  v_b[3] = tid < XD ? a.br2[tid] : 0.f;
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            const int i = tid + k * NTHR, r = i / XD, c = i - r * XD;
            w[a.off_wh2 + r * WLD + c] = v_wh2[k];
            w[a.off_wr2 + r * WLD + c] = v_wr2[k];
        }
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int i = tid + k * NTHR, r = i / XD, c = i - r * XD;
            w[a.off_wa + r * WLD + c] = v_wa[k];
            w[a.off_w1 + r * WLD + c] = v_w1[k];
        }
        { const int r = tid / HID, c = tid - r * HID; w[a.off_wh1 + r * W1LD + c] = v_wh1; }             // 8 x 64 = 512 entries
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * NTHR, r = i / HID, c = i - r * HID;
            if (i < 12 * HID) w[a.off_wr1 + r * W1LD + c] = v_wr1[k];
        }
        if (tid < HID) { w[a.off_bh1 + tid] = v_b[0]; w[a.off_br1 + tid] = v_b[1]; }
        if (tid < XD) { w[a.off_bh2 + tid] = v_b[2]; w[a.off_br2 + tid] = v_b[3]; }
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

    // embedding of my 16 children of parent p: x0 = w_r(robot'), y = x0 W1 -> Y0, g0 = x0 Wa, s00 = g0 . x0  (registers of the wave)
    f32x4 xacc[2], gacc[2];
    float s00 = 0.f;
    const int c = 16 * wave + n;              // meaningful for child waves only
    auto embed1 = [&](int p) {
        xacc[0] = xacc[1] = gacc[0] = gacc[1] = zero4();
        s00 = 0.f;
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
    };
    PHASE_START();
    int buf = 0;
    int* crowd_flag = reinterpret_cast<int*>(lds + a.off_flag);      // [NT] epoch reached by each crowd wave's part 1
    int crowd_epoch = 0;
    if (tid < NT) crowd_flag[tid] = 0;
    if ((int)blockIdx.x < a.P) {               // prime the pipeline: crowd block of the first parent || its children's embedding
        if (crowd_wave) prologue1(blockIdx.x, 0);
        else if (child_wave) embed1(blockIdx.x);
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
        f32x4 t0h[2] = {zero4(), zero4()};        // child waves: (p_c Xh)^T of my 16 children, from embed-2 to the robot-row pass
        float p00 = 0.f;                          // A_c[0][0]
        if (child_wave) {
            if (p != (int)blockIdx.x) embed1(p);       // the first parent's embedding ran under the crowd prologue that primed the pipeline
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

struct Rank1Plan {
    Rank1Args a;
    size_t lds_bytes;
    int hr;
    bool ok;
    bool image_layout;                    // the LDS offsets coincide with FusedLds: a packed image can be copied in
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
    // weight image in the order (and with the strides) of FusedLds: one copy from the packed image when the caller has one
    a.off_wr1 = take(12 * W1LD); a.off_br1 = take(HID); a.off_wr2 = take(HID * WLD); a.off_br2 = take(XD);
    a.off_wa = take(XD * WLD); a.off_w1 = take(XD * WLD);
    a.off_wh1 = take(8 * W1LD); a.off_bh1 = take(HID); a.off_wh2 = take(HID * WLD); a.off_bh2 = take(XD);
    using FL = FusedLds<32, 100, 100>;
    static_assert(FL::wr1 == 0, "image starts at the robot embedding");
    a.image = nullptr;
    pl.image_layout = a.off_wr1 == FL::wr1 && a.off_br1 == FL::br1 && a.off_wr2 == FL::wr2 && a.off_br2 == FL::br2 && a.off_wa == FL::wa &&
                      a.off_w1 == FL::w1 && a.off_wh1 == FL::wh1 && a.off_bh1 == FL::bh1 && a.off_wh2 == FL::wh2 && a.off_bh2 == FL::bh2;
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

// 1 = outside this kernel's envelope
int launch_rank1_children(const RglGraph* g, int P, int A, int H, const float* child_robot, const float* humans_next,
                          float* rows_out, hipStream_t stream, const float* image) {
    Rank1Plan rp = plan_rank1(*g, P, A, H);
    if (!rp.ok) return 1;
    rp.a.image = rp.image_layout ? image : nullptr;
    rp.a.child_robot = child_robot;
    rp.a.humans = humans_next;
    rp.a.rows_out = rows_out;
    return launch_rank1(rp, g->skip_connection != 0, stream);
}

}  // namespace rgl
