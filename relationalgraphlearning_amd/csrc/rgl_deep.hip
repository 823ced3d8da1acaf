// rgl_deep.hip -- stage 1 of "value of the sibling children" for dense crowds and the 3-layer graph
// (N <= 64 nodes, L in {2, 3}; BASELINE configs[4]: N = 50, L = 3), shared-crowd form.
//
// The A children of a parent share every human row of X and every S_ij with i, j >= 1; only the robot row and the
// robot column of the relation matrix differ.  With the crowd-only, max-shifted E_ij = e^{S_ij - msh_i} (i, j >= 1),
// Zsh_i = sum_j E_ij and the per-child scalars  m = max(msh_i, S_c[i][0]),  alpha = e^{msh_i - m},
// beta = e^{S_c[i][0] - m},  Z = alpha Zsh_i + beta,  a_i = alpha / Z,  b_i = beta / Z,  p = softmax(S_c[0][:]):
//
//   layer 0   H1_i = relu(a_i UW_i + b_i (x0_c W1)) (+ Xh_i)              UW = (E Xh) W1 is crowd-only  ("rank-1" form)
//             H1_0 = relu((p X_c) W1) (+ x0_c)                             batched over 16 children on the MFMA
//   layer 1   O    = H1 W2                                                 MFMA, W2 shared            (L == 3 only)
//             H2_i = relu(a_i (E O)_i + b_i O_0) (+ H1_i)                  MFMA, E shared: the SAME A-operand fragments
//             H2_0 = relu(p O) (+ H1_0)                                    (row 0 of E's first tile := p, per child)
//   last      t_c  = sum_i p_i H_{L-1,i}   ->  rows_out = [ t_c | H_{L-1,0} ];   stage 2 (robot_head_kernel) applies W_last.
//
// Per child the only O(N^2) work left is the E O product, whose left operand never leaves the registers.
// F16 = true feeds the two dense products of layer 1 to v_mfma_f32_16x16x32_f16 (f16 inputs, f32 accumulate;
// BASELINE configs[4] "fp16 MFMA XW path"); everything else, and F16 = false throughout, is exact fp32
// (v_mfma_f32_16x16x4_f32 + VALU).
//
// Follows (reference paths): crowd_nav/policy/graph_model.py:99-130, value_estimator.py:11-20,
// model_predictive_rl.py:245-250.
#include "rgl_head_body.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct DeepArgs {
    const float *wr1, *br1, *wr2, *br2, *wh1, *bh1, *wh2, *bh2, *wa, *w1, *w2;
    const float* child_robot;             // [P][A][9]
    const float* humans;                  // [P][H][5]
    int P, A, H, N, L, CT, TLD;           // TLD: row stride of the per-child tables (multiple of 4, >= N)
    int sim;                              // SIM_* row normalisation
    float* rows_out;                      // [P*A][64]
    int off_wh1, off_bh1, off_wh2, off_bh2, off_wa, off_wr1, off_br1, off_wr2, off_br2, off_w1, off_w2;   // weight image
    int off_xh, off_uw, off_gm, off_msh, off_zsh;       // crowd block
    int off_x0, off_y0, off_g0, off_s00;                // per-child rows; g0 is re-used for H1_0
    int off_tp, off_ta, off_tb;                         // tables [A][TLD]: p, S_c[.][0] -> a, (E exchange) -> b
    // round 4: stage 2 in the same launch.  Workgroup b owns the parents [b k, (b + 1) k) and, once their rows are written, runs the
    // value head over them (and the search's tail steps) where robot_head_kernel would in a launch of its own: head_rows_and_tail
    int fuse_head, parents_per_wg;
};

// The ReLUs of the per-child loop sit in the clamp bit of their FMAs (rgl_mfma.h): the row scalars a, b are held as 2^-64 a, 2^-64 b
// in the tables (|a UW + b y| < 2^64 = 1.8e19 stays exact), layer 0 scales back in the FMA that adds the skip term, layer 1 once per
// child after the p-weighted sum.
constexpr float kAbScale = 0x1p-64f, kAbUnscale = 0x1p64f;
constexpr int kDeepThreads = 512;
constexpr int kDeepWaves = 8;

// all-reduce over the 16 lanes of a DPP row, four values at once.  The DPP operand rides in the add itself (through the update_dpp
// builtin every step was v_mov 0 + v_mov_dpp + v_add), and the stages of the four sums interleave: three independent instructions
// between a write and its DPP read, so no wait states to pad beyond the first (a DPP read of a VGPR needs two after the VALU write
// of it, and the compiler's hazard recognizer does not look inside inline assembly; the one-value form spent an s_nop 1 per step)
__device__ __forceinline__ f32x4 row16_sum4(f32x4 v) {
    float a = v[0], b = v[1], c = v[2], d = v[3];
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    return f32x4{a, b, c, d};
}
__device__ __forceinline__ f16x8 pack8(f32x4 lo, f32x4 hi) {
    f16x8 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = (_Float16)lo[r]; v[4 + r] = (_Float16)hi[r]; }
    return v;
}

template <int NT, bool F16, bool SKIP, bool SOFT, bool T4 = false>
__global__ __launch_bounds__(kDeepThreads, 2) void children_deep_kernel(const DeepArgs a, const HeadArgs ha) {
    const int sim = SOFT ? (int)SIM_SOFTMAX : a.sim;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KT = (NT + 1) / 2;          // f16: k tiles of 32 nodes
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int N = a.N, H = a.H, A = a.A, TLD = a.TLD;
    // T4 kernels touch table rows 0 .. 16 LT + 3 only, all below TLD (plan_deep): the `row < TLD` guards of the per-child loop --
    // ~200 exec-mask / branch instructions per child in the general form -- are compile-time truths there
    // -- and in every form the rows of the node tiles before the last lie inside (N > 16 LT, TLD >= N): only the last tile is checked
    auto in_table = [&](int row, int tile) { return tile < NT - 1 ? true : row < TLD; };
    auto child_row = [&](int row, int tile) { return T4 ? true : in_table(row, tile); };     // the reads of the per-child loop
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
    const float* w2 = lds + a.off_w2;     // [XD][WLD]  (L == 3)
    float* Xh = lds + a.off_xh;           // [16*NT][XLD]   row 0 and rows >= N are zero
    float* UW = lds + a.off_uw;           // [16*NT][XLD]
    float* Gm = lds + a.off_gm;           // [16*NT][XLD]
    float* MSH = lds + a.off_msh;         // [16*NT]
    float* ZSH = lds + a.off_zsh;
    float* X0 = lds + a.off_x0;           // [A][XLD]
    float* Y0 = lds + a.off_y0;           // [A][XLD]   x0 W1
    float* G0 = lds + a.off_g0;           // [A][XLD]   x0 Wa, later H1_0
    float* S00 = lds + a.off_s00;         // [A]
    float* TP = lds + a.off_tp;           // [A][TLD]
    float* TA = lds + a.off_ta;
    float* TB = lds + a.off_tb;
    float* EX = TB;                       // E exchange: [NT][NT][64 lanes][4], dead before TB is written
    float* scr = Gm + wave * 32;          // per-wave transposition scratch of the per-child loop (Gm is dead by then)
    const float NEG_INF = -INFINITY;

    {   // weight image, once per workgroup
        float* w = lds;
#pragma unroll 4
        for (int i = tid; i < 8 * HID; i += kDeepThreads) {
            const int r = i / HID, c = i - r * HID;
            w[a.off_wh1 + r * W1LD + c] = r < 5 ? a.wh1[i] : 0.f;
        }
        for (int i = tid; i < HID; i += kDeepThreads) { w[a.off_bh1 + i] = a.bh1[i]; w[a.off_br1 + i] = a.br1[i]; }
        for (int i = tid; i < XD; i += kDeepThreads) { w[a.off_bh2 + i] = a.bh2[i]; w[a.off_br2 + i] = a.br2[i]; }
#pragma unroll 4
        for (int i = tid; i < HID * XD; i += kDeepThreads) {
            const int r = i / XD, c = i - r * XD;
            w[a.off_wh2 + r * WLD + c] = a.wh2[i];
            w[a.off_wr2 + r * WLD + c] = a.wr2[i];
        }
#pragma unroll 4
        for (int i = tid; i < XD * XD; i += kDeepThreads) {
            const int r = i / XD, c = i - r * XD;
            w[a.off_wa + r * WLD + c] = a.wa ? a.wa[i] : (r == c ? 1.f : 0.f);   // gaussian: Wa = I
            w[a.off_w1 + r * WLD + c] = a.w1[i];
            if (a.L == 3) w[a.off_w2 + r * WLD + c] = a.w2[i];
        }
#pragma unroll 4
        for (int i = tid; i < 12 * HID; i += kDeepThreads) {
            const int r = i / HID, c = i - r * HID;
            w[a.off_wr1 + r * W1LD + c] = r < 9 ? a.wr1[i] : 0.f;
        }
    }
    __syncthreads();

    // f16: W2 as B fragments (K = f = 8q + e, N = g = 16 gt + n), converted once
    f16x8 w2h[2];
    if (F16 && a.L == 3) {
#pragma unroll
        for (int gt = 0; gt < 2; ++gt)
#pragma unroll
            for (int e = 0; e < 8; ++e) w2h[gt][e] = (_Float16)w2[(8 * q + e) * WLD + 16 * gt + n];
    }

    // Node order inside a 16-row MFMA tile whose rows feed the NEXT product's k index.  f32 path: A-operand row m holds
    // node 16 jt + 4 (m & 3) + (m >> 2), so that the D rows 4q + r of a tile -- the k slots of lane group q in k step r --
    // are the consecutive nodes 16 jt + 4 r + q: k step s = 4 jt + r covers nodes 4s .. 4s+3 and steps >= ceil(N / 4)
    // are skipped.  f16 path (K = 32 per instruction, nothing to skip): plain order, node 16 jt + 4 q + r.
    constexpr bool PERM = !F16;
    const int jrow = PERM ? 4 * (n & 3) + (n >> 2) : n;
    auto knode = [&](int jt, int r) { return PERM ? 16 * jt + 4 * r + q : 16 * jt + 4 * q + r; };
    const int KS = (N + 3) >> 2;
    // T4 (round 4): the LAST node tile holds at most four valid nodes (configs[4]: N = 50 -> nodes 48, 49 in a tile of 16).  Its
    // rows of O = H1 W2 and of E O then come from v_mfma_f32_4x4x1_16B_f32 (rgl_mfma.h: 12 clocks per k step instead of 32): the
    // A operands -- the layer-0 rows of nodes 16 LT + (lane & 3), E's last row tile replicated over the quads once per parent --
    // pair with the B operands the 16 x 16 products read anyway, and a reduce(-scatter) over the k-groups leaves the results where
    // the next step wants them (O: node 16 LT + q in k-slot order; E O: row 16 LT + q in k-group q).
    // A template flag (launch_deep_t picks it from N and L): with both forms in one kernel the fourth tile's registers stay live.
    constexpr int LT = NT - 1;
    constexpr bool t4 = T4;
    static_assert(!T4 || (!F16 && NT >= 2), "T4: the f32 form, at least two node tiles");
    const bool crowd_wave = wave < NT;
    const int n_child_waves = kDeepWaves - NT;

    PHASE_START();
    const int p_first = a.fuse_head ? blockIdx.x * a.parents_per_wg : blockIdx.x;
    const int p_end = a.fuse_head ? (p_first + a.parents_per_wg < a.P ? p_first + a.parents_per_wg : a.P) : a.P;
    const int p_step = a.fuse_head ? 1 : gridDim.x;
    for (int p = p_first; p < p_end; p += p_step) {
        PHASE_MARK(0);
        // =========================== phase A: crowd embedding  ||  child embedding ===========================
        f32x4 pg[2] = {zero4(), zero4()};          // crowd waves: G^T of my column tile, carried into phase B
        const int node = 16 * wave + n;            // crowd waves only
        const bool node_ok = crowd_wave && node >= 1 && node < N;
        if (crowd_wave) {
            const float* hsrc = a.humans + ((size_t)p * H + (node_ok ? node - 1 : 0)) * 5;
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
                for (int r = 0; r < 4; ++r) xacc[ot][r] = node_ok ? relu1(xacc[ot][r] + bb[r]) : 0.f;
                *reinterpret_cast<f32x4*>(&Xh[node * XLD + 16 * ot + 4 * q]) = xacc[ot];
            }
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
        }
        // the 16-child tiles: the child waves first, the crowd waves join with the tiles beyond them once their crowd work is done
        for (int ct = crowd_wave ? n_child_waves + wave : wave - NT; ct < a.CT; ct += kDeepWaves) {
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
                        xacc[ot] = mfma4(wr2[(16 * ht + 4 * q + r) * WLD + 16 * ot + n], hacc[ht][r], xacc[ot]);
            }
            load_fence();
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&br2[16 * ot + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) xacc[ot][r] = relu1(xacc[ot][r] + bb[r]);
                if (c < A) *reinterpret_cast<f32x4*>(&X0[c * XLD + 16 * ot + 4 * q]) = xacc[ot];
            }
            f32x4 yacc[2] = {zero4(), zero4()}, gacc[2] = {zero4(), zero4()};
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
            float s00 = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (c < A) {
                    *reinterpret_cast<f32x4*>(&Y0[c * XLD + 16 * t + 4 * q]) = yacc[t];
                    *reinterpret_cast<f32x4*>(&G0[c * XLD + 16 * t + 4 * q]) = gacc[t];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) s00 = fmaf(gacc[t][r], xacc[t][r], s00);
            }
            s00 = kgroups_sum(s00);
            if (q == 0 && c < A) S00[c] = s00;
        }
        PHASE_MARK(1);
        __syncthreads();
        PHASE_MARK(2);

        // =========================== phase B: crowd relation block  ||  robot row / column of S ==============
        if (crowd_wave) {
            f32x4 e[NT];
            float mx = NEG_INF;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                load_fence();
                f32x4 sacc = zero4();
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    const f32x4 xa = *reinterpret_cast<const f32x4*>(&Xh[(16 * jt + jrow) * XLD + 16 * ft + 4 * q]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc = mfma4(xa[r], pg[ft][r], sacc);      // [j = 16jt+4r+q][i = my node]
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = knode(jt, r);
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
                MSH[node] = mx;
                ZSH[node] = node_ok ? z : 1.f;
            }
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
                *reinterpret_cast<f32x4*>(&EX[((wave * NT + jt) * 64 + lane) * 4]) = e[jt];
            f32x4 u[2] = {zero4(), zero4()};
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a0 = Xh[knode(jt, r) * XLD + n];
                    const float a1 = Xh[knode(jt, r) * XLD + 16 + n];
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
        }
        // the 16-child tiles: the child waves first, the crowd waves join with the tiles beyond them once their crowd work is done
        for (int ct = crowd_wave ? n_child_waves + wave : wave - NT; ct < a.CT; ct += kDeepWaves) {
            const int c = 16 * ct + n;
            const int cc = c < A ? c : A - 1;
            f32x4 xq[2], gq[2];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                xq[ot] = *reinterpret_cast<const f32x4*>(&X0[cc * XLD + 16 * ot + 4 * q]);
                gq[ot] = *reinterpret_cast<const f32x4*>(&G0[cc * XLD + 16 * ot + 4 * q]);
            }
            const float s00 = S00[cc];
            f32x4 s0t[NT];
            float mx0 = NEG_INF;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                load_fence();
                f32x4 sc = zero4(), s0 = zero4();
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(&Gm[(16 * nt + n) * XLD + 16 * ot + 4 * q]);
                    const f32x4 xh = *reinterpret_cast<const f32x4*>(&Xh[(16 * nt + n) * XLD + 16 * ot + 4 * q]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sc = mfma4(gm[r], xq[ot][r], sc);       // S_c[node][0]
                        s0 = mfma4(xh[r], gq[ot][r], s0);       // S_c[0][node]
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
                if (c < A && in_table(16 * nt + 4 * q, nt)) *reinterpret_cast<f32x4*>(&TA[c * TLD + 16 * nt + 4 * q]) = sc;
            }
            // p = softmax of the robot row, in the D layout: my lane holds nodes 16 nt + 4 q + r of child c
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
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s0t[nt][r] *= iz0;
                if (c < A && in_table(16 * nt + 4 * q, nt)) *reinterpret_cast<f32x4*>(&TP[c * TLD + 16 * nt + 4 * q]) = s0t[nt];
            }
        }
        PHASE_MARK(3);
        __syncthreads();

        // =========================== E fragments into registers (all waves) ====================================
        // A operand of the E O product: lane (n, q) of fragment [it][jt] holds E[i = 16 it + n][j = knode(jt, r)].
        f32x4 Ef[F16 ? 1 : NT][F16 ? 1 : NT];
        f16x8 Eh[F16 ? NT : 1][F16 ? KT : 1];
        if (a.L == 3) {
            if (F16) {
#pragma unroll
                for (int it = 0; it < NT; ++it)
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(&EX[((it * NT + 2 * t) * 64 + lane) * 4]);
                        const f32x4 hi = 2 * t + 1 < NT
                                             ? *reinterpret_cast<const f32x4*>(&EX[((it * NT + 2 * t + 1) * 64 + lane) * 4])
                                             : zero4();
                        Eh[F16 ? it : 0][F16 ? t : 0] = pack8(lo, hi);
                    }
            } else {
#pragma unroll
                for (int it = 0; it < NT; ++it)
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt)
                        Ef[F16 ? 0 : it][F16 ? 0 : jt] = *reinterpret_cast<const f32x4*>(&EX[((it * NT + jt) * 64 + lane) * 4]);
                if (t4) {                  // rows 16 LT .. 16 LT + 3 of E as the A operand of the 4 x 4 x 1 blocks (A row = lane % 4)
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            Ef[F16 ? 0 : LT][F16 ? 0 : jt][r] = quad0_bcast(Ef[F16 ? 0 : LT][F16 ? 0 : jt][r]);
                }
            }
        }
        __syncthreads();
        PHASE_MARK(4);
        PHASE_MARK(5);

        // =========================== phase C': row scalars a, b; robot row of layer 0, 16 children per MFMA tile =================
        for (int ct = wave; ct < a.CT; ct += kDeepWaves) {
            const int c = 16 * ct + n;
            const int cc = c < A ? c : A - 1;
            // row scalars a, b of my 16 children (D layout: nodes 16 nt + 4 q + r), over the raw S_c[node][0] left in TA
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (!in_table(16 * nt + 4 * q, nt)) continue;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(&TA[cc * TLD + 16 * nt + 4 * q]);
                const f32x4 ms = *reinterpret_cast<const f32x4*>(&MSH[16 * nt + 4 * q]);
                const f32x4 zs = *reinterpret_cast<const f32x4*>(&ZSH[16 * nt + 4 * q]);
                f32x4 av, bv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nd = 16 * nt + 4 * q + r;
                    float al, be;
                    if (sim == SIM_SOFTMAX) {
                        const float m = fmaxf(ms[r], sc[r]);
                        al = __expf(ms[r] - m);
                        be = __expf(sc[r] - m);
                    } else {
                        al = 1.f;
                        be = plain_weight(sim, sc[r], nd, 0);
                    }
                    const float iz = __builtin_amdgcn_rcpf(fmaf(al, zs[r], be));
                    const bool row_h = nd >= 1 && nd < N;
                    av[r] = row_h ? al * iz * kAbScale : (nd == 0 ? kAbScale : 0.f);
                    bv[r] = row_h ? be * iz * kAbScale : 0.f;
                }
                if (c < A) {
                    *reinterpret_cast<f32x4*>(&TA[c * TLD + 16 * nt + 4 * q]) = av;
                    *reinterpret_cast<f32x4*>(&TB[c * TLD + 16 * nt + 4 * q]) = bv;
                }
            }
            f32x4 t0[2] = {zero4(), zero4()};                 // T0^T[f = 16 ft + 4q + r][c]
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                load_fence();
                const f32x4 pb = in_table(16 * jt + 4 * q, jt) ? *reinterpret_cast<const f32x4*>(&TP[cc * TLD + 16 * jt + 4 * q]) : zero4();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    t0[0] = mfma4(Xh[(16 * jt + 4 * q + r) * XLD + n], pb[r], t0[0]);
                    t0[1] = mfma4(Xh[(16 * jt + 4 * q + r) * XLD + 16 + n], pb[r], t0[1]);
                }
            }
            const float p00 = TP[cc * TLD];
            f32x4 xv[2];
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                xv[ft] = *reinterpret_cast<const f32x4*>(&X0[cc * XLD + 16 * ft + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) t0[ft][r] = fmaf(p00, xv[ft][r], t0[ft][r]);
            }
            f32x4 o[2] = {zero4(), zero4()};
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot)
                        o[ot] = mfma4(w1[(16 * ft + 4 * q + r) * WLD + 16 * ot + n], t0[ft][r], o[ot]);
            }
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = SKIP ? relu1(o[ot][r]) + xv[ot][r] : relu1(o[ot][r]);
                if (c < A) *reinterpret_cast<f32x4*>(&G0[c * XLD + 16 * ot + 4 * q]) = h;            // H1_0
            }
        }
        __syncthreads();
        PHASE_MARK(6);

        // =========================== per child: layer 0 rows, layer 1, robot-row aggregation ===================
        const float* H10 = G0;
        for (int c = wave; c < A; c += kDeepWaves) {
            // feature index of my k slots: f32 path f = 16 fh + 4 q + r, f16 path f = 8 q + 4 fh + r
            const int fo0 = F16 ? 8 * q : 4 * q, fo1 = F16 ? 8 * q + 4 : 16 + 4 * q;
            f32x4 yv[2], h10v[2];
            yv[0] = *reinterpret_cast<const f32x4*>(&Y0[c * XLD + fo0]);
            yv[1] = *reinterpret_cast<const f32x4*>(&Y0[c * XLD + fo1]);
            h10v[0] = *reinterpret_cast<const f32x4*>(&H10[c * XLD + fo0]);
            h10v[1] = *reinterpret_cast<const f32x4*>(&H10[c * XLD + fo1]);
            f32x4 h1[NT][2];
            f32x4 tsk[2] = {zero4(), zero4()};
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                // T4: the last tile's rows in the 4 x 4 x 1 A layout -- node 16 LT + (n & 3), four copies of each (p counts one)
                const bool last4 = t4 && jt == LT;
                const int j = last4 ? 16 * LT + (n & 3) : 16 * jt + jrow;
                const bool jv = child_row(j, jt);
                const float aj = jv ? TA[c * TLD + j] : 0.f;
                const float bj = jv ? TB[c * TLD + j] : 0.f;
                const float pj = (jv && !(last4 && n >= 4)) ? TP[c * TLD + j] : 0.f;
#pragma unroll
                for (int fh = 0; fh < 2; ++fh) {
                    const int fo = fh ? fo1 : fo0;
                    const f32x4 uw = *reinterpret_cast<const f32x4*>(&UW[j * XLD + fo]);
                    const f32x4 xh = *reinterpret_cast<const f32x4*>(&Xh[j * XLD + fo]);
                    // two features per instruction (v_pk_mul / v_pk_fma / v_pk_add: same roundings as the scalar chain)
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) {
                        const f32x2 y2{yv[fh][2 * hp], yv[fh][2 * hp + 1]}, u2{uw[2 * hp], uw[2 * hp + 1]};
                        f32x2 v = pk_fma_lo_clamp(f32x2{aj, bj}, u2, f32x2{bj, bj} * y2);      // 2^-64 relu(fma(a, uw, round(b y)))
                        if (SKIP) v = __builtin_elementwise_fma(f32x2{kAbUnscale, kAbUnscale}, v, f32x2{xh[2 * hp], xh[2 * hp + 1]});
                        else v *= kAbUnscale;
                        if (jt == 0 && n == 0) v = f32x2{h10v[fh][2 * hp], h10v[fh][2 * hp + 1]};
                        h1[jt][fh][2 * hp] = v[0];
                        h1[jt][fh][2 * hp + 1] = v[1];
                        const f32x2 t2 = __builtin_elementwise_fma(f32x2{pj, pj}, v, f32x2{tsk[fh][2 * hp], tsk[fh][2 * hp + 1]});
                        tsk[fh][2 * hp] = t2[0];
                        tsk[fh][2 * hp + 1] = t2[1];
                    }
                }
            }
            if (SKIP || a.L == 2) {
#pragma unroll
                for (int fh = 0; fh < 2; ++fh) {
                    tsk[fh] = row16_sum4(tsk[fh]);
                    if (n == 0) *reinterpret_cast<f32x4*>(&scr[fh ? fo1 : fo0]) = tsk[fh];
                }
            }
            float* out = a.rows_out + ((size_t)p * A + c) * 64;
            if (a.L == 2) {
                // t_c = sum_j p_j H1_j ; H_{L-1,0} = H1_0
                if (q < 2) {
                    const int g = 16 * q + n;
                    out[g] = scr[g];
                    out[32 + g] = H10[c * XLD + g];
                }
                continue;
            }
            // ---- O = H1 W2 : tiles [jt][gt], lane (n, q) holds O[j = knode(jt, r)][g = 16 gt + n]
            f32x4 O[NT][2];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) { O[jt][0] = zero4(); O[jt][1] = zero4(); }
            if (F16) {
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) {
                    const f16x8 ah = pack8(h1[jt][0], h1[jt][1]);
#pragma unroll
                    for (int gt = 0; gt < 2; ++gt)
                        O[jt][gt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, w2h[gt], O[jt][gt], 0, 0, 0);
                }
            } else {
                f32x4 o4[2][2] = {{zero4(), zero4()}, {zero4(), zero4()}};      // T4: [gt][accumulator]
#pragma unroll
                for (int fh = 0; fh < 2; ++fh) {
                    load_fence();
                    float wb[4][2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        wb[r][0] = w2[(16 * fh + 4 * q + r) * WLD + n];
                        wb[r][1] = w2[(16 * fh + 4 * q + r) * WLD + 16 + n];
                    }
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt) {
                        if (t4 && jt == LT) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                o4[0][r & 1] = mfma4x4(h1[jt][fh][r], wb[r][0], o4[0][r & 1]);
                                o4[1][r & 1] = mfma4x4(h1[jt][fh][r], wb[r][1], o4[1][r & 1]);
                            }
                            continue;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            O[jt][0] = mfma4(h1[jt][fh][r], wb[r][0], O[jt][0]);
                            O[jt][1] = mfma4(h1[jt][fh][r], wb[r][1], O[jt][1]);
                        }
                    }
                }
                if (t4) {          // node 16 LT + q of the last tile in register 0 of lane (n, q): its only populated k step
                    O[LT][0] = f32x4{kgroups_reduce_scatter(o4[0][0] + o4[0][1]), 0.f, 0.f, 0.f};
                    O[LT][1] = f32x4{kgroups_reduce_scatter(o4[1][0] + o4[1][1]), 0.f, 0.f, 0.f};
                }
            }
            // row scalars in the D layout: i = 16 it + 4 q + r;  pn = p in k-slot order (row 0 of E's first tile)
            f32x4 pn[NT];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                if (PERM) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (T4 && jt == LT && r >= 1) pn[jt][r] = 0.f;          // nodes 16 LT + 4 r + q >= 16 LT + 4: beyond the crowd
                        else pn[jt][r] = child_row(knode(jt, r), jt) ? TP[c * TLD + knode(jt, r)] : 0.f;
                    }
                } else {
                    pn[jt] = in_table(16 * jt + 4 * q, jt) ? *reinterpret_cast<const f32x4*>(&TP[c * TLD + 16 * jt + 4 * q]) : zero4();
                }
            }
            // O_0[g]: held by the q == 0 lanes (r = 0 of tile jt = 0: node 0)
            float o0[2];
#pragma unroll
            for (int gt = 0; gt < 2; ++gt) o0[gt] = kgroups_sum(q == 0 ? O[0][gt][0] : 0.f);
            // ---- D2 = E O (+ robot row p O), then H2 = relu(a D2 + b O_0), t_c = sum_i p_i H2_i
            float tacc[2] = {0.f, 0.f};
            f32x2 tacc2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};      // F16 branch: two rows per packed instruction
            float hrow0[2] = {0.f, 0.f};
            if (F16) {
                f16x8 ob[KT][2];
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int gt = 0; gt < 2; ++gt)
                        ob[t][gt] = pack8(O[2 * t][gt], 2 * t + 1 < NT ? O[2 * t + 1 < NT ? 2 * t + 1 : 0][gt] : zero4());
#pragma unroll
                for (int it = 0; it < NT; ++it) {
                    load_fence();
                    const bool rv = in_table(16 * it + 4 * q, it);
                    const f32x4 pq = pn[it];               // plain node order: the k-slot order of p is its row order
                    const f32x4 aq = rv ? *reinterpret_cast<const f32x4*>(&TA[c * TLD + 16 * it + 4 * q]) : zero4();
                    const f32x4 bq = rv ? *reinterpret_cast<const f32x4*>(&TB[c * TLD + 16 * it + 4 * q]) : zero4();
                    f32x4 d[2] = {zero4(), zero4()};
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        f16x8 ea = Eh[F16 ? it : 0][F16 ? t : 0];
                        if (it == 0 && n == 0) ea = pack8(pn[2 * t], 2 * t + 1 < NT ? pn[2 * t + 1 < NT ? 2 * t + 1 : 0] : zero4());
                        d[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea, ob[t][0], d[0], 0, 0, 0);
                        d[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea, ob[t][1], d[1], 0, 0, 0);
                    }
                    // two rows per instruction; the two partial sums of t_c meet after the loop
#pragma unroll
                    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp) {
                            const f32x2 a2{aq[2 * hp], aq[2 * hp + 1]}, b2{bq[2 * hp], bq[2 * hp + 1]}, p2{pq[2 * hp], pq[2 * hp + 1]};
                            // (a D2 rounded first, then the FMA with b O_0: the inline-assembly FMA must not read the MFMA's result
                            // itself -- see pk_fma_clamp)
                            const f32x2 v = pk_fma_clamp(b2, f32x2{o0[gt], o0[gt]}, a2 * f32x2{d[gt][2 * hp], d[gt][2 * hp + 1]});
                            tacc2[gt] = __builtin_elementwise_fma(p2, v, tacc2[gt]);
                            if (it == 0 && hp == 0) hrow0[gt] = v[0];
                        }
                }
                tacc[0] = tacc2[0][0] + tacc2[0][1];
                tacc[1] = tacc2[1][0] + tacc2[1][1];
            } else {
#pragma unroll
                for (int it = 0; it < NT; ++it) {
                    load_fence();
                    if (t4 && it == LT) {
                        // rows 16 LT + i of E O on the 4 x 4 x 1 blocks; the reduce-scatter over the k-groups leaves row 16 LT + q in
                        // lane (n, q), column g = n: every row is finished (and weighted into t_c) by one k-group
                        f32x4 d4[2][2] = {{zero4(), zero4()}, {zero4(), zero4()}};
#pragma unroll
                        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (jt == LT && r >= 1) continue;             // T4: the last tile's one populated k step
                                const float ea = Ef[F16 ? 0 : LT][F16 ? 0 : jt][r];
                                d4[0][r & 1] = mfma4x4(ea, O[jt][0][r], d4[0][r & 1]);
                                d4[1][r & 1] = mfma4x4(ea, O[jt][1][r], d4[1][r & 1]);
                            }
                        const int i4 = 16 * LT + q;               // the reduce-scatter leaves row 16 LT + q in lane (n, q), column g = n
                        const float p1 = TP[c * TLD + i4], a1 = TA[c * TLD + i4], b1 = TB[c * TLD + i4];      // i4 <= 16 LT + 3 < TLD
#pragma unroll
                        for (int gt = 0; gt < 2; ++gt) {
                            const float x = kgroups_reduce_scatter(d4[gt][0] + d4[gt][1]);
                            const float v = fma_clamp(a1, x, b1 * o0[gt]);
                            tacc[gt] = fmaf(p1, v, tacc[gt]);
                        }
                        continue;
                    }
                    const bool rv = child_row(16 * it + 4 * q, it);
                    const f32x4 pq = rv ? *reinterpret_cast<const f32x4*>(&TP[c * TLD + 16 * it + 4 * q]) : zero4();
                    const f32x4 aq = rv ? *reinterpret_cast<const f32x4*>(&TA[c * TLD + 16 * it + 4 * q]) : zero4();
                    const f32x4 bq = rv ? *reinterpret_cast<const f32x4*>(&TB[c * TLD + 16 * it + 4 * q]) : zero4();
                    f32x4 d[2] = {zero4(), zero4()};
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (T4 ? (jt == LT && r >= 1) : (jt == NT - 1 && 4 * jt + r >= KS)) continue;   // k steps beyond the last node (uniform)
                            float ea = Ef[F16 ? 0 : it][F16 ? 0 : jt][r];
                            if (it == 0 && n == 0) ea = pn[jt][r];
                            d[0] = mfma4(ea, O[jt][0][r], d[0]);
                            d[1] = mfma4(ea, O[jt][1][r], d[1]);
                        }
#pragma unroll
                    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = fma_clamp(aq[r], d[gt][r], bq[r] * o0[gt]);
                            tacc[gt] = fmaf(pq[r], v, tacc[gt]);
                            if (it == 0 && r == 0) hrow0[gt] = v;
                        }
                }
            }
#pragma unroll
            for (int gt = 0; gt < 2; ++gt) {
                const int g = 16 * gt + n;
                float t = kgroups_sum(tacc[gt]) * kAbUnscale;       // the layer-1 ReLUs ran on 2^-64 of their values
                float hp = hrow0[gt] * kAbUnscale;          // valid in the q == 0 lanes (row i = 0)
                if (SKIP) {
                    t += scr[g];
                    hp += H10[c * XLD + g];
                }
                if (q == 0) {
                    out[g] = t;
                    out[32 + g] = hp;
                }
            }
        }
        PHASE_MARK(7);
        __syncthreads();
    }
    PHASE_FLUSH();
    if (a.fuse_head) {
        // stage 2 over the rows this workgroup has just written (behind the loop's closing barrier: every table above is dead, the
        // head's 73 KB of fragments take their place).  One launch and one weight image per level less than the two-stage form;
        // workgroups that finish their parents early run their heads under the others' children.
        head_load_image<32, 100, 100>(lds, ha, tid);
        __syncthreads();
        head_rows_and_tail<32, 100, 100>(lds, ha, blockIdx.x, gridDim.x);
    }
}

inline int deep_workgroups_per_cu(size_t lds_bytes) { return lds_bytes * 2 <= (size_t)rgl::kLdsBytesPerCu ? 2 : 1; }

struct DeepPlan {
    DeepArgs a;
    size_t lds_bytes;
    int NT;
    bool ok;
    bool t4;            // the T4 form (last node tile on the 4 x 4 x 1 MFMA, TLD = 16 NT)
};

inline DeepPlan plan_deep(const RglGraph& g, int P, int A, int H) {
    DeepPlan pl;
    pl.ok = false;
    if (!fast_path_enabled() || !rank1_enabled()) return pl;
    if (fast_similarity_mode(g) < 0 || g.layerwise_graph || g.x_dim != XD) return pl;
    if (g.num_layer != 2 && g.num_layer != 3) return pl;
    if (!mlp_is(g.w_r, 9, HID, XD, true) || !mlp_is(g.w_h, 5, HID, XD, true)) return pl;
    const int N = H + 1;
    if (N > 64 || A > 96 || A < 1) return pl;
    DeepArgs& a = pl.a;
    a.N = N; a.H = H; a.A = A; a.P = P; a.L = g.num_layer;
    a.sim = fast_similarity_mode(g);
    pl.NT = (N + 15) / 16;
    a.CT = (A + 15) / 16;
    a.TLD = (N + 3) & ~3;
    // last node tile with at most four valid nodes, three layers, softmax similarity: the T4 form of the kernel (4 x 4 x 1 MFMA for
    // that tile).  Its table rows end at node 16 LT + 3 (TLD = (N + 3) & ~3 >= 16 LT + 4): every row the T4 kernel touches is inside.
    static const bool t4_off = [] { const char* e = getenv("RGL_DEEP_T4"); return e && e[0] == '0'; }();
    const bool t4_wanted = !t4_off && a.L == 3 && pl.NT >= 2 && N - 16 * (pl.NT - 1) <= 4 && a.sim == SIM_SOFTMAX;
    pl.t4 = t4_wanted;
    int off = 0;
    auto take = [&](int nfl) { int o = off; off += (nfl + 3) & ~3; return o; };
    a.off_wh1 = take(8 * W1LD); a.off_bh1 = take(HID); a.off_wh2 = take(HID * WLD); a.off_bh2 = take(XD);
    a.off_wa = take(XD * WLD); a.off_wr1 = take(12 * W1LD); a.off_br1 = take(HID); a.off_wr2 = take(HID * WLD);
    a.off_br2 = take(XD); a.off_w1 = take(XD * WLD);
    a.off_w2 = a.L == 3 ? take(XD * WLD) : a.off_w1;
    a.off_xh = take(16 * pl.NT * XLD); a.off_uw = take(16 * pl.NT * XLD); a.off_gm = take(16 * pl.NT * XLD);
    a.off_msh = take(16 * pl.NT); a.off_zsh = take(16 * pl.NT);
    a.off_x0 = take(A * XLD); a.off_y0 = take(A * XLD); a.off_g0 = take(A * XLD);
    a.off_s00 = take(A);
    a.off_tp = take(A * a.TLD); a.off_ta = take(A * a.TLD);
    const int ex_floats = pl.NT * pl.NT * 256;
    a.off_tb = take(A * a.TLD > ex_floats ? A * a.TLD : ex_floats);
    pl.lds_bytes = (size_t)off * sizeof(float);
    if (pl.lds_bytes > (size_t)rgl::kLdsBytesPerCu) return pl;
    a.wr1 = g.w_r.weight[0]; a.br1 = g.w_r.bias[0]; a.wr2 = g.w_r.weight[1]; a.br2 = g.w_r.bias[1];
    a.wh1 = g.w_h.weight[0]; a.bh1 = g.w_h.bias[0]; a.wh2 = g.w_h.weight[1]; a.bh2 = g.w_h.bias[1];
    a.wa = bilinear_wa(g); a.w1 = g.Ws[0]; a.w2 = g.num_layer == 3 ? g.Ws[1] : g.Ws[0];
    pl.ok = true;
    return pl;
}

template <int NT, bool F16, bool SKIP, bool SOFT, bool T4 = false>
int launch_deep_t(const DeepPlan& pl, const HeadArgs& ha, hipStream_t st) {
    if constexpr (!T4 && !F16 && SOFT && NT >= 2) {
        if (pl.t4) return launch_deep_t<NT, F16, SKIP, SOFT, true>(pl, ha, st);      // plan_deep: eligible and its tables fit (RGL_DEEP_T4=0: never)
    }
    auto kern = children_deep_kernel<NT, F16, SKIP, SOFT, T4>;
    if (pl.lds_bytes > 64 * 1024)
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)pl.lds_bytes));
    const int per_cu = deep_workgroups_per_cu(pl.lds_bytes);
    int grid = pl.a.P < 256 * per_cu ? pl.a.P : 256 * per_cu;             // persistent workgroups
    if (pl.a.fuse_head) grid = (pl.a.P + pl.a.parents_per_wg - 1) / pl.a.parents_per_wg;      // contiguous parents per workgroup
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kDeepThreads), pl.lds_bytes, st, pl.a, ha);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

template <int NT>
int launch_deep_nt(const DeepPlan& pl, const HeadArgs& ha, bool f16, bool skip, hipStream_t st) {
    const bool soft = pl.a.sim == SIM_SOFTMAX;
    if (f16) {
        if (!soft) return 1;                      // f16 contractions are built for the softmax similarities only
        return skip ? launch_deep_t<NT, true, true, true>(pl, ha, st) : launch_deep_t<NT, true, false, true>(pl, ha, st);
    }
    if (soft) return skip ? launch_deep_t<NT, false, true, true>(pl, ha, st) : launch_deep_t<NT, false, false, true>(pl, ha, st);
    return skip ? launch_deep_t<NT, false, true, false>(pl, ha, st) : launch_deep_t<NT, false, false, false>(pl, ha, st);
}

}  // namespace

#ifdef RGL_PHASE_TIMING
extern "C" int rgl_debug_read_deep_phase_cycles(unsigned long long* out16, int reset) {
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

// Returns RGL_OK after launching, a negative / hip error code on failure, or 1 when the request is outside this
// kernel's envelope (the caller then picks another kernel).
// `head` .. `head_done` (optional): with the shipped value head and its packed image at hand the kernel also runs stage 2 -- and the
// search's tail steps, `tail` / `tail_done` as in launch_head_rows -- over the rows of the parents each workgroup owns; *head_done = 1
// then, and `value` holds the children's values (RGL_DEEP_FUSE_HEAD=0: never, the caller launches robot_head_kernel as before).
int launch_deep_children(const RglGraph* g, int P, int A, int H, const float* child_robot, const float* humans_next,
                         float* rows_out, int f16, hipStream_t stream, const RglMlp* head, float* value, const float* image,
                         const void* tail, size_t tail_bytes, int* tail_done, int* head_done) {
    if (head_done) *head_done = 0;
    DeepPlan pl = plan_deep(*g, P, A, H);
    if (!pl.ok) return 1;
    pl.a.child_robot = child_robot;
    pl.a.humans = humans_next;
    pl.a.rows_out = rows_out;
    pl.a.fuse_head = 0;
    pl.a.parents_per_wg = 1;
    HeadArgs ha{};
    static const bool fuse_off = [] { const char* e = getenv("RGL_DEEP_FUSE_HEAD"); return e && e[0] == '0'; }();
    const size_t head_lds = (size_t)HeadLds<32, 100, 100>::total * sizeof(float);
    if (head && head_done && value && image && !fuse_off && head_variant(*head) == 0) {
        const size_t lds = pl.lds_bytes > head_lds ? pl.lds_bytes : head_lds;
        int chain = 0;
        head_args_for(g, head, 0, rows_out, P * A, value, image, tail, tail_bytes, A, 256 * deep_workgroups_per_cu(lds), &ha, &chain);
        if (!ha.tail.enabled) {                   // stand-alone call: no tail, the same ownership of rows
            const int slots = 256 * deep_workgroups_per_cu(lds);
            ha.parents_per_wg = (P + slots - 1) / slots;
            ha.own_rows = 1;
        }
        pl.lds_bytes = lds;
        pl.a.fuse_head = 1;
        pl.a.parents_per_wg = ha.parents_per_wg;
        if (tail_done) *tail_done = ha.tail.enabled ? (chain ? 2 : 1) : 0;
        *head_done = 1;
    }
    const bool skip = g->skip_connection != 0;
    switch (pl.NT) {
        case 1: return launch_deep_nt<1>(pl, ha, f16 != 0, skip, stream);
        case 2: return launch_deep_nt<2>(pl, ha, f16 != 0, skip, stream);
        case 3: return launch_deep_nt<3>(pl, ha, f16 != 0, skip, stream);
        default: return launch_deep_nt<4>(pl, ha, f16 != 0, skip, stream);
    }
}

}  // namespace rgl
