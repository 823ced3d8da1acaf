// rgl_fused.hip -- "value of the sibling children" for the shipped shape (L = 2, N <= 32, value head 32-100-100-1) as ONE
// kernel over a stream of 16-child tiles.
//
// Follows (reference paths): crowd_nav/policy/graph_model.py:99-130 (RGL.forward), value_estimator.py:11-20,
// model_predictive_rl.py:245-250 (the per-action loop whose iterations run side by side here).
//
// Algebra: the rank-1 (shared-crowd) form of rgl_rank1.hip -- siblings share every human row of X and of S except the
// robot column, so a human row i of child c is (alpha_i UW_i + beta_i (x0_c W1)) / Z_i with crowd-only UW, msh, Zsh.
// What changes is the organisation (round 2):
//   * children_fused_kernel: every wave owns one 16-child tile from the robot embedding to the VALUE: embedding,
//     robot row/column of S, softmax scalars, the VALU row pass, the robot row, the last GCN layer on the robot row and the
//     value head all run on registers and a 5-6.5 KB wave-private LDS scratch.  No workgroup barrier after the weight image
//     is built, no [P*A][64] fp32 hand-off through HBM, no second launch.  Round 3: every workgroup OWNS a contiguous block
//     of parents; their work items (groups of tiles of one parent, plan_items) are dealt over the workgroup's 8 waves in
//     snake passes, heaviest first, so the four SIMDs of a CU carry the same load.  Because all 81 values of a parent are
//     produced inside one workgroup, the search's bookkeeping for that parent -- one-step values, top-w clipping, gather of the
//     next level's robot rows, and at the deepest level the whole back-up chain and the root decision (rgl_tail.h) -- runs in
//     the TAIL of this kernel behind a workgroup barrier: mprl_select / mprl_backup / mprl_root are no launches of their own.
//   * row pass (softmax similarity, N <= 20) in the MFMA D layout with packed fp32 math: 4 instructions per 2 elements.
//   * the crowd-only quantities of a parent (Xh, G = Xh Wa, UW, msh, Zsh: 208 MFMAs, ~9 % of a parent's work) are computed
//     by the wave that owns the work item, at the item's start, straight into the registers the tiles read them from
//     (the MFMA D layout of the crowd chain IS the A-operand layout of the robot row / column products; the two
//     transposed views go through the wave's own scratch).  Nothing crowd-sized crosses HBM: the kernel reads the child
//     robot rows and the parent's human rows and writes the values.  (Until the end of round 2 a crowd_block_kernel
//     produced 7.8-16.6 KB blocks per parent in global memory: one more launch and 5-10x the algorithmic HBM bytes.)
//   * a parent's partial last tile (A % 16 children; the single `stop` action of the 81-action table) runs everything but
//     the head and leaves its rows in a small buffer; the workgroup scores the rows of ITS partial tiles at the end of the
//     kernel, tile-packed over parents (16 `stop` children = one head tile), so the head never multiplies 15 columns of
//     padding and the whole path is one launch.
#include "rgl_mlp_chain.h"
#include "rgl_tail.h"

#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------
// the fused tile kernel
// ------------------------------------------------------------------------------------------------
// Wave-private LDS scratch of children_fused_kernel (floats).  Packed row pass (softmax, HR <= 20): UW [HRL][32] of the item's parent
// | (r, b) table [16][HRL + 2][2] of the tile; otherwise (a, b) table [16][16 NT + 1][2] | Y [16][XLD].  Between items the same
// space holds Xh [16 NT][XLD] | msh, Zsh [2][16 NT] of the crowd computation.
__host__ __device__ constexpr bool fused_packed_rows(int HR, bool SOFT) { return SOFT && HR <= 20; }
__host__ __device__ constexpr int fused_scratch_floats(int HR, int NT, bool SOFT) {
    const int NP = 16 * NT, HRL = HR < NP ? HR : NP;
    const int main_fl = fused_packed_rows(HR, SOFT) ? HRL * 32 + 16 * (HRL + 2) * 2 : 2 * 16 * (NP + 1) + 16 * XLD;
    const int crowd_fl = NP * XLD + 2 * NP;
    const int fl = main_fl > crowd_fl ? main_fl : crowd_fl;
    return ((fl > kTailLdsInts ? fl : kTailLdsInts) + 3) & ~3;      // ... and, after the last item, the tables of the select step
}

struct FusedArgs {
    const float *wr1, *br1, *wr2, *br2, *wa, *w1;                       // child-side weights (k-major)
    const float *wh1, *bh1, *wh2, *bh2;                                 // w_h (crowd side)
    const float* w_last;                                                // [32][32] last GCN layer
    const float *hw1, *hb1, *hw2, *hb2, *hw3, *hb3, *hw4, *hb4;         // value head, k-major
    const float* child_robot;     // [P][A][9]
    const float* humans;          // [P][H][5] the parents' crowds
    float* value;                 // [P][A]
    float* rows_left;             // [P][A % 16][64]  rows [t_c | H1_0] of the partial last tiles (null when A % 16 == 0)
    int P, A, N, H, SLD, sim;
    int n_full;                   // full tiles per parent = A / 16
    int rem;                      // A % 16
    const float* image;           // FusedLds weight image [0, FusedLds::scratch) prepared in global memory by pack_images_kernel
    int tiles_per_item;           // a work item = this many consecutive FULL tiles of one parent (crowd quantities computed once)
    int items_per_parent;         // = ceil(n_full / tiles_per_item); the last group also carries the partial tile
    int rot;                      // item order: group (o + rot) % items_per_parent at order position o (heaviest group first)
    int parents_per_wg;           // workgroup b owns parents [b k, (b + 1) k): every item of a parent runs on that workgroup's waves
    int image_sync;               // measurements: stage the whole weight image through registers before anything else (round 3)
    int inline_partial;           // few parents per workgroup: the partial tile is an ordinary (padded) tile with its own head
                                  // instead of a row hand-off to the tile-packed pass at the end (a barrier + a serial head)
    int phase_delay;              // measurements (RGL_FUSED_PHASE_DELAY): waves 4..7 start this many 512-cycle sleeps late
    int prio;                     // RGL_FUSED_PRIO: bit 1 (default) = s_setprio 2 inside the head: -0.6 % / -1.3 % at 2048 / 4096 parents; bit 0
                                  // (measurements) = static s_setprio 1 for waves 4..7: nothing (profiles/r06_c_phase_prio_ab.txt)
    TailArgs tail;                // tail.enabled: select (+ back-up chain + root step) for the owned parents at the end
};

constexpr int kFusedWaves = 8;
static_assert(B6Floats<HID, XD>::v == HID * XD * 3 / 2 && B6Floats<XD, XD>::v == XD * XD * 3 / 2, "FusedLds sizes its BX matrices so");
static_assert((FusedLds<32, 100, 100, true>::scratch + kFusedWaves * 1344 + 4) * 4 <= 160 * 1024, "the BX image + 8 wave scratches fit a CU");


// h = relu(t W_last)(+hprev), value head 32 -> D1 -> D2 -> D3 -> 1 for the 16 children of a tile (lane (n, q): child n, D-layout
// registers); returns the value of child n in every lane of its column (without the last bias)
template <class LO, int D1, int D2, int D3, bool SKIP, bool BX = false>
__device__ __forceinline__ float head_chain(const float* lds, const f32x4 (&tin)[2], const f32x4 (&hp)[2], int lane) {
    const int q = lane >> 4;
    f32x4 h[2];
    if constexpr (BX) layer_mfma_b6<XD, XD, false>(lds + LO::f_last, tin, h, lane);
    else layer_mfma<XD, XD, false>(lds + LO::f_last, tin, h, lane);
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = relu1(h[ot][r]);
            if (SKIP) x += hp[ot][r];
            h[ot][r] = x;
        }
    f32x4 a1[Tiles<D1>::v];
    if constexpr (BX && D1 == 32) layer_mfma_bx1<XD, 32, true>(lds + LO::f1, h, a1, lane, lds + LO::b1);
    else layer_mfma<XD, D1, true>(lds + LO::f1, h, a1, lane, lds + LO::b1);
    relu_tiles<D1>(a1);
    f32x4 a2[Tiles<D2>::v];
    if constexpr (BX) layer_mfma_bx1<D1, D2, true>(lds + LO::f2, a1, a2, lane, lds + LO::b2);
    else layer_mfma<D1, D2, true>(lds + LO::f2, a1, a2, lane, lds + LO::b2);
    relu_tiles<D2>(a2);
    f32x4 a3[Tiles<D3>::v];
    if constexpr (BX) layer_mfma_bx<D2, D3, true>(lds + LO::f3, a2, a3, lane, lds + LO::b3);
    else layer_mfma<D2, D3, true>(lds + LO::f3, a2, a3, lane, lds + LO::b3);
    relu_tiles<D3>(a3);
    float v = 0.f;
#pragma unroll
    for (int ot = 0; ot < Tiles<D3>::v; ++ot) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&lds[LO::w4 + 16 * ot + 4 * q]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v = fmaf(a3[ot][r], w[r], v);
    }
    return kgroups_sum(v);
}

// scale of the PK row pass (children_fused_kernel): its ReLU sits in the clamp bit of v_pk_fma_f32 (rgl_mfma.h), which clamps to [0, 1]
constexpr float kRowScale = 0x1p-110f, kRowUnscale = 0x1p110f;

// HR >= N: human rows of UW held in registers (padded rows contribute exactly 0); SOFT: softmax row normalisation
// BX: the D2 x D3 head matrix's first 64 input features as six bf16 terms on the matrix pipe (layer_mfma_bx; RGL_CONTRACT_BF16X6)
template <int HR, int NT, bool SKIP, bool SOFT, int D1, int D2, int D3, bool BX = false>
__global__ __launch_bounds__(kFusedWaves * 64) void children_fused_kernel(const FusedArgs a) {
    const int sim = SOFT ? (int)SIM_SOFTMAX : a.sim;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using LO = FusedLds<D1, D2, D3, BX>;
    constexpr int NP = 16 * NT;
    constexpr int nthreads = kFusedWaves * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    constexpr bool PK = fused_packed_rows(HR, SOFT);    // row pass in the D layout with packed fp32 math
    // 16 < N <= 20 (the shipped crowd): the second node tile holds four valid nodes.  Its rows of the robot row / column of S come
    // from v_mfma_f32_4x4x1_16B_f32 (rgl_mlp_chain.h: 12 clocks instead of 32 per k step) and land, after a reduce-scatter over
    // the k-groups, as node 16 + q in register 0 of lane (n, q): the tile's node order is 16 + 4 r + q, registers 1..3 are
    // padding -- their exps, table entries and k steps of p Xh are not computed at all.
    constexpr bool T1P = PK && NT == 2 && HR <= 20;
    constexpr int HRL = HR < NP ? HR : NP;              // rows of UW the row pass visits (rows >= N are zero)
    constexpr int SLDK = HRL + 2;
    const int N = a.N, A = a.A, SLD = PK ? SLDK : a.SLD;
    const float NEG_INF = -INFINITY;
    const float* wr1 = lds + LO::wr1;   // [12][W1LD], rows 9..11 zero (BX: [9][72], see FusedLds)
    const float* br1 = lds + LO::br1;
    const float* wr2 = lds + LO::wr2;   // [HID][WLD]
    const float* br2 = lds + LO::br2;
    const float* wa = lds + LO::wa;     // [XD][WLD]
    const float* w1 = lds + LO::w1;     // [XD][WLD]
    float* WS = lds + LO::scratch + wave * fused_scratch_floats(HR, NT, SOFT);
    float* UWs = WS;                                                    // PK: [HRL][32] UW rows of the item's parent
    float* AB = PK ? WS + HRL * 32 : WS;                                // [16][SLD][2]: per child and row (a, b) / (r, b), p folded in
    float* Y0 = AB + 2 * 16 * SLD;                                      // !PK: [16][XLD]: x0 W1, then t_c without the robot-row term
    const float hb4 = a.hb4[0];

    // Work items of the parents this workgroup owns, dealt over its 8 waves:
    //   item (j, p), group-major (all first groups, then all second ...): G consecutive full tiles of parent p; the LAST (short)
    //   group of a parent starts with its partial tile (A % 16 children: everything but the head, rows -> rows_left; scored at the
    //   end of the kernel, tile-packed over the workgroup's parents) -- or, with few parents per workgroup (inline_partial), the
    //   partial tile is one more ordinary tile with its own head.
    // The crowd quantities of a parent are computed once per item, and the NEXT item's loads (robot rows, crowd state rows) are
    // issued before the head of the current item's last tile: no item starts waiting on HBM.
    const int G = a.tiles_per_item;
    const int p_first = blockIdx.x * a.parents_per_wg;          // my workgroup's parents: p_first .. p_first + k_b - 1
    const int k_b = a.P - p_first < a.parents_per_wg ? a.P - p_first : a.parents_per_wg;
    const int n_items = k_b * a.items_per_parent;
    const int n_pass = (n_items + kFusedWaves - 1) / kFusedWaves;
    // pass k deals the workgroup's items 8 k .. 8 k + 7 over its waves, odd passes in reverse ("snake"): with the heavy items
    // first in the order, the waves that got a heavy item in one pass get a light one (or none) in the next; waves w and w + 4
    // share a SIMD
    auto item_at = [&](int k) { return k * kFusedWaves + ((k & 1) ? kFusedWaves - 1 - wave : wave); };
    const int n_tiles_inl = a.n_full + (a.inline_partial ? 1 : 0);      // tiles that run their own head
    float rin[3], hin[NT][2];
    // BX: w_h's second matrix [64][32] as the A-operand elements my lane feeds to its 32 MFMAs of every crowd computation --
    // fragment (ht, r, ot): row 16 ht + 4 q + r, column 16 ot + n -- held in registers for the whole kernel (see FusedLds::bh2)
    float wh2r[BX ? 32 : 1];
    if constexpr (BX) {
#pragma unroll
        for (int ht = 0; ht < 4; ++ht)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) wh2r[(ht * 4 + r) * 2 + ot] = a.wh2[(16 * ht + 4 * q + r) * XD + 16 * ot + n];
    }
    f32x4 gq[NT][2], xq[NT][2], ms[NT], zs[NT], xt[NT][2], uw4[HRL / 4];
    float xt1p[2] = {0.f, 0.f}, ms1 = 0.f, zs1 = 1.f;      // T1P: Xh^T[f][node 16 + q], msh / Zsh of node 16 + q
    // item wi = (order position o, local parent), o-major; group j = (o + rot) % items_per_parent covers the full tiles j G ..; the
    // LAST group (the short one) also carries the parent's partial tile, which it runs first.  Tile sequence ts .. t1-1, where
    // t < t0 means "the partial tile".
    auto item_tiles = [&](int wi, int& p, int& t0, int& ts, int& t1) {
        const int o = wi / k_b;
        p = p_first + (wi - o * k_b);
        int j = o + a.rot;
        if (j >= a.items_per_parent) j -= a.items_per_parent;
        t0 = j * G;
        t1 = t0 + G < n_tiles_inl ? t0 + G : n_tiles_inl;
        ts = (j == a.items_per_parent - 1 && a.rem && !a.inline_partial) ? t0 - 1 : t0;
    };
    auto robot_rows = [&](int p, int t) {                          // rows of child 16 t + n of parent p -> rin
        const int c0 = 16 * t + n;
        const float* rr = a.child_robot + ((size_t)p * A + (c0 < A ? c0 : A - 1)) * 9;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 4 * s + q;
            rin[s] = k < 9 ? rr[k] : 0.f;
        }
    };
    auto human_rows = [&](int p) {                                 // state rows of the parent's crowd (node 16 pct + n) -> hin
#pragma unroll
        for (int pct = 0; pct < NT; ++pct) {
            const int node = 16 * pct + n;
            const bool ok = node >= 1 && node < N;
            const float* hsrc = a.humans + ((size_t)p * a.H + (ok ? node - 1 : 0)) * 5;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int k = 4 * s + q;
                hin[pct][s] = (ok && k < 5) ? hsrc[k] : 0.f;
            }
        }
    };
    auto item_loads = [&](int wi) {                                // first robot rows + crowd state rows of item wi
        int p, t0, ts, t1;
        item_tiles(wi, p, t0, ts, t1);
        robot_rows(p, ts < t0 ? a.n_full : ts);
        human_rows(p);
    };
    // Crowd-only quantities of the item's parent, from hin, into the registers the tiles read (row 0 = robot slot and rows >= N
    // are zero rows / contribute nothing).  The MFMA D layout of the chain (lane (n, q): node 16 pct + n, features 16 ot + 4 q + r)
    // is the A-operand layout of the S products; Xh^T and the feature-major UW go through the wave's scratch (free between items).
    auto crowd_compute = [&]() {
        const float* wh1 = lds + LO::wh1;   // [8][W1LD], rows 5..7 zero (BX: [5][72])
        const float* bh1 = lds + LO::bh1;
        const float* wh2 = lds + LO::wh2;   // [HID][WLD]
        const float* bh2 = lds + LO::bh2;
        float* Xs = WS;                     // [NP][XLD]: Xh, later UW
        float* msz = WS + NP * XLD;         // [2][NP]: msh | Zsh
        // part 1: Xh = w_h(humans), G = Xh Wa   (transposed MFMA chain, 16 nodes per pass)
#pragma unroll
        for (int pct = 0; pct < NT; ++pct) {
            const int node = 16 * pct + n;
            const bool node_ok = node >= 1 && node < N;
            f32x4 hacc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int k = 4 * s + q;
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) hacc[ht] = mfma4(wh1[(k < LO::WH1ROWS ? k : LO::WH1ROWS - 1) * LO::WH1LD + 16 * ht + n], hin[pct][s], hacc[ht]);
            }
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&bh1[16 * ht + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) hacc[ht][r] = relu1(hacc[ht][r] + bb[r]);
            }
            f32x4 xa[2] = {zero4(), zero4()};
            {
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot)
                        xa[ot] = mfma4(BX ? wh2r[BX ? (ht * 4 + r) * 2 + ot : 0] : wh2[(16 * ht + 4 * q + r) * WLD + 16 * ot + n], hacc[ht][r], xa[ot]);
            }
            load_fence();
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&bh2[16 * ot + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) xa[ot][r] = node_ok ? relu1(xa[ot][r] + bb[r]) : 0.f;   // robot slot / padding rows: 0
                *reinterpret_cast<f32x4*>(&Xs[node * XLD + 16 * ot + 4 * q]) = xa[ot];
                xq[pct][ot] = xa[ot];
            }
            f32x4 pg[2] = {zero4(), zero4()};
            if constexpr (BX) layer_mfma_b6<XD, XD, false>(wa, xa, pg, lane);
            else {
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gt = 0; gt < 2; ++gt)
                        pg[gt] = mfma4(wa[(16 * ot + 4 * q + r) * WLD + 16 * gt + n], xa[ot][r], pg[gt]);
            }
            load_fence();
            }
            gq[pct][0] = pg[0];
            gq[pct][1] = pg[1];
            }
        }
        __builtin_amdgcn_wave_barrier();
        load_fence();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r)      // Xh^T for p Xh and U; T1P: only k step 0 of tile 1 is read by the tiles (node 16 + q)
                    xt[nt][ot][r] = Xs[(16 * nt + 4 * q + r) * XLD + 16 * ot + n];
        if constexpr (T1P) {
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) xt1p[ot] = Xs[(16 + q) * XLD + 16 * ot + n];
        }
        load_fence();
        __builtin_amdgcn_wave_barrier();
        
        // part 2 (every Xh row is in registers now): S_ij = G_i . Xh_j over humans j, msh / E / Zsh, U = E Xh, UW = U W1
#pragma unroll
        for (int pct = 0; pct < NT; ++pct) {
            const int node = 16 * pct + n;
            const bool node_ok = node >= 1 && node < N;
            f32x4 e[NT];
            float mx = NEG_INF;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                f32x4 sacc = zero4();
                {
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc = mfma4(xq[jt][ft][r], gq[pct][ft][r], sacc);      // [j = 16jt+4q+r][i = my node]
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
                msz[node] = mx;
                msz[NP + node] = node_ok ? z : 1.f;
            }
            f32x4 u[2] = {zero4(), zero4()};
            f32x4 uw[2] = {zero4(), zero4()};
            {
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    u[0] = mfma4(xt[jt][0][r], e[jt][r], u[0]);                               // U^T[f][i] = sum_j Xh[j][f] E[i][j]
                    u[1] = mfma4(xt[jt][1][r], e[jt][r], u[1]);
                }
            if constexpr (BX) layer_mfma_b6<XD, XD, false>(w1, u, uw, lane);
            else {
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
            }
            }
            if constexpr (PK) {                                                              // UW rows over the (consumed) Xh rows
                if (node < HRL) {
                    *reinterpret_cast<f32x4*>(&UWs[node * 32 + 4 * q]) = uw[0];
                    *reinterpret_cast<f32x4*>(&UWs[node * 32 + 16 + 4 * q]) = uw[1];
                }
            } else {
                *reinterpret_cast<f32x4*>(&Xs[node * XLD + 4 * q]) = uw[0];
                *reinterpret_cast<f32x4*>(&Xs[node * XLD + 16 + 4 * q]) = uw[1];
            }
        }
        __builtin_amdgcn_wave_barrier();
        load_fence();
#pragma unroll
        for (int nt = 0; nt < (T1P ? 1 : NT); ++nt) {
            ms[nt] = *reinterpret_cast<const f32x4*>(&msz[16 * nt + 4 * q]);
            zs[nt] = *reinterpret_cast<const f32x4*>(&msz[NP + 16 * nt + 4 * q]);
        }
        if constexpr (T1P) {
            ms1 = msz[16 + q];
            zs1 = msz[NP + 16 + q];
            // rows 16..19 of G and Xh as A operands of the 4 x 4 x 1 blocks: A row = lane % 4 (rows 20..31 were zero padding)
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gq[1][ot][r] = quad0_bcast(gq[1][ot][r]);
                    xq[1][ot][r] = quad0_bcast(xq[1][ot][r]);
                }
        }
        if constexpr (!PK) {
#pragma unroll
            for (int i4 = 0; i4 < HRL / 4; ++i4)
#pragma unroll
                for (int k = 0; k < 4; ++k) uw4[i4][k] = Xs[(4 * i4 + k) * XLD + (lane & 31)];  // lane = feature holds its nodes
        }
        load_fence();
        __builtin_amdgcn_wave_barrier();      // the tiles' AB / Y0 writes stay behind these reads
    };
    // Weight image -> LDS with LDS-direct loads (global_load_lds_dwordx4: a wave moves 1 KB per instruction, no registers in
    // between), in TWO parts (round 4): everything the crowd quantities and a tile need before its head -- the embedding and graph
    // matrices, the per-feature vectors: [0, f_last), a third of the image -- is waited for here; the value head's fragments (the
    // other two thirds) keep streaming in under the first item's crowd computation and its first tile's embedding / row pass, and
    // every wave waits for them once, in front of its first head (await_image: its own loads via vmcnt, the other waves' via an
    // LDS counter).  Until round 4 the whole image was staged through registers before anything else ran: ~4 us of every launch.
    static_assert(LO::scratch % 4 == 0 && LO::f_last % 4 == 0, "b128 granules");
    constexpr int kChunk = 256;                                            // floats per wave and instruction
    constexpr int kChunksA = (LO::f_last + kChunk - 1) / kChunk, kChunksAll = (LO::scratch + kChunk - 1) / kChunk;
    int* image_arrivals = reinterpret_cast<int*>(lds + LO::scratch + kFusedWaves * fused_scratch_floats(HR, NT, SOFT));
    auto image_chunk = [&](int c) {
        const int fl = c * kChunk + lane * 4;
        if (fl < LO::scratch)                                              // the last chunk is partial
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.image + fl),
                                             (__attribute__((address_space(3))) void*)(lds + c * kChunk), 16, 0, 0);
    };
    // (the first item's state rows are requested BEFORE the image: the wait in front of the barrier covers them, and no later
    // wait for them can hold the wave until the head fragments have landed as well -- vmcnt counts in order)
    if (wave < n_items) item_loads(item_at(0));
    if (tid == 0) *image_arrivals = 0;
    bool image_complete = false;                                           // wave-uniform
    if (a.image_sync) {                                                    // RGL_FUSED_IMAGE_SYNC=1 (measurements): the round-3 copy
        copy_image<LO::scratch, nthreads>(lds, a.image, tid);
        __syncthreads();
        image_complete = true;
    } else {
    for (int c = wave; c < kChunksA; c += kFusedWaves) image_chunk(c);
    __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0)
    __syncthreads();
    for (int c = kChunksA + wave; c < kChunksAll; c += kFusedWaves) image_chunk(c);
    }
    auto await_image = [&]() {
        if (image_complete) return;
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // my share of the head fragments has landed
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(image_arrivals, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(image_arrivals, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < kFusedWaves) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        image_complete = true;
    };
    if (wave >= 4) {
        for (int i = 0; i < a.phase_delay; ++i) __builtin_amdgcn_s_sleep(8);
        if (a.prio & 1) __builtin_amdgcn_s_setprio(1);
    }
    PHASE_START();
    for (int pass = 0; pass < n_pass; ++pass) {
        const int wi = item_at(pass);
        if (wi >= n_items) break;                          // only the last pass is short
        const int wi_next = pass + 1 < n_pass ? item_at(pass + 1) : n_items;
        PHASE_MARK(0);
        int p, t0, ts, t1;
        item_tiles(wi, p, t0, ts, t1);
        load_fence();
        crowd_compute();
        PHASE_MARK(1);

      for (int ti = ts; ti < t1; ++ti) {
        const int t = ti < t0 ? a.n_full : ti;             // the first group of a parent starts with the partial tile
        const bool full = t < a.n_full;
        const int c = 16 * t + n;                          // my child (column of the MFMA tiles)
        const int n_valid = full ? 16 : a.rem;

        // ---------------- embedding: x0 = w_r(robot'), y = x0 W1, g0 = x0 Wa (transposed MFMA chain) ----------------
        f32x4 xacc[2] = {zero4(), zero4()}, gacc[2] = {zero4(), zero4()}, yacc[2] = {zero4(), zero4()};
        float s00 = 0.f;
        {
            f32x4 hacc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int k = 4 * s + q;
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) hacc[ht] = mfma4(wr1[(k < LO::WR1ROWS ? k : LO::WR1ROWS - 1) * LO::WR1LD + 16 * ht + n], rin[s], hacc[ht]);
            }
            if (ti + 1 < t1) robot_rows(p, ti + 1);      // robot rows of my next tile: in flight under this tile's work
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&br1[16 * ht + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) hacc[ht][r] = relu1(hacc[ht][r] + bb[r]);
            }
            if constexpr (BX) {
                // round 6: the tile chain's weight products -- hidden -> x0 (64 x 32), x0 Wa, x0 W1 -- as six bf16 terms over three-piece
                // operands too (fragments in the image where the f32 matrices were; x0 is split once for both products)
                layer_mfma_b6<HID, XD, true>(wr2, hacc, xacc, lane, br2);
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xacc[ot][r] = relu1(xacc[ot][r]);
                const Split3 sx = split3_pair(xacc[0], xacc[1]);
                layer_mfma_b6_pre<XD, false>(wa, sx, gacc, lane);
                layer_mfma_b6_pre<XD, false>(w1, sx, yacc, lane);
            } else {
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
            }
            if constexpr (!PK) {
                *reinterpret_cast<f32x4*>(&Y0[n * XLD + 4 * q]) = yacc[0];
                *reinterpret_cast<f32x4*>(&Y0[n * XLD + 16 + 4 * q]) = yacc[1];
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s00 = fmaf(gacc[tt][r], xacc[tt][r], s00);
            s00 = kgroups_sum(s00);
        }

        PHASE_MARK(2);
        // ---------------- robot row / column of S, p = softmax(robot row), p Xh, the (a, b) table ---------------------
        // all in the MFMA D layout: lane (n, q) = child 16 t + n, registers = nodes 16 nt + 4 q + r
        f32x4 t0h[2] = {zero4(), zero4()};        // (p_c Xh)^T of my 16 children
        float p00;                                // A_c[0][0]
        {
            f32x4 s0t[NT], sct[NT];
            float mx0 = NEG_INF;
            float sc1 = NEG_INF, p1 = NEG_INF;           // T1P: S_c[16 + q][0], then S_c[0][16 + q] -> p of node 16 + q
            if constexpr (T1P) {
                f32x4 psc[2] = {zero4(), zero4()}, ps0[2] = {zero4(), zero4()};
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        psc[r & 1] = mfma4x4(gq[1][ot][r], xacc[ot][r], psc[r & 1]);
                        ps0[r & 1] = mfma4x4(xq[1][ot][r], gacc[ot][r], ps0[r & 1]);
                    }
                sc1 = kgroups_reduce_scatter(psc[0] + psc[1]);
                p1 = kgroups_reduce_scatter(ps0[0] + ps0[1]);
                if (16 + q >= N) { sc1 = NEG_INF; p1 = NEG_INF; }
                mx0 = p1;
            }
#pragma unroll
            for (int nt = 0; nt < (T1P ? 1 : NT); ++nt) {
                f32x4 sc = zero4(), s0 = zero4();
                {
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sc = mfma4(gq[nt][ot][r], xacc[ot][r], sc);
                        s0 = mfma4(xq[nt][ot][r], gacc[ot][r], s0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nd = 16 * nt + 4 * q + r;
                    if (nd == 0) { sc[r] = s00; s0[r] = s00; }
                    if (sim != SIM_SOFTMAX) s0[r] = plain_weight(sim, s0[r], 0, nd);
                    if (!T1P && nd >= N) { sc[r] = NEG_INF; s0[r] = sim == SIM_SOFTMAX ? NEG_INF : 0.f; }   // T1P: nd <= 15 < N here
                    mx0 = fmaxf(mx0, s0[r]);
                }
                s0t[nt] = s0;
                sct[nt] = sc;
            }
            mx0 = kgroups_max(mx0);
            float z0 = 0.f;
            if constexpr (T1P) {
                p1 = __expf(p1 - mx0);
                z0 = p1;
            }
#pragma unroll
            for (int nt = 0; nt < (T1P ? 1 : NT); ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (sim == SIM_SOFTMAX) s0t[nt][r] = __expf(s0t[nt][r] - mx0);
                    z0 += s0t[nt][r];
                }
            const float iz0 = __builtin_amdgcn_rcpf(kgroups_sum(z0));
            if constexpr (T1P) p1 *= iz0;
#pragma unroll
            for (int nt = 0; nt < (T1P ? 1 : NT); ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s0t[nt][r] *= iz0;                       // p = A_c[0][:] (0 beyond row N-1)
            p00 = kgroups_sum(q == 0 ? s0t[0][0] : 0.f);
            // (p_c Xh)^T[f][c] = sum_j Xh^T[f][j] p_c[j]: the D registers of the robot-row product are already the B operand
            {
#pragma unroll
            for (int nt = 0; nt < (T1P ? 1 : NT); ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    t0h[0] = mfma4(xt[nt][0][r], s0t[nt][r], t0h[0]);
                    t0h[1] = mfma4(xt[nt][1][r], s0t[nt][r], t0h[1]);
                }
            if constexpr (T1P) {                                 // nodes 16..19: ONE k step (k slot q = node 16 + q)
                t0h[0] = mfma4(xt1p[0], p1, t0h[0]);
                t0h[1] = mfma4(xt1p[1], p1, t0h[1]);
            }
            }
            if constexpr (PK) {
                // human row i of child c: relu((alpha UW_i + beta y_c) / Z) weighted by p_i = b_i relu(r_i UW_i + y_c) with
                // r_i = alpha / beta = exp(msh_i - S_i0), b_i = p_i beta / Z = p_i / (r_i Zsh_i + 1); r is capped at e^60 (beyond,
                // beta y is below fp32 resolution of the row and b r = p / Zsh is exact)
#pragma unroll
                for (int nt = 0; nt < (T1P ? 1 : NT); ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int nd = 16 * nt + 4 * q + r;
                        const float rr = __expf(fminf(ms[nt][r] - sct[nt][r], 60.f));
                        const float bb = s0t[nt][r] * __builtin_amdgcn_rcpf(fmaf(rr, zs[nt][r], 1.f));
                        // T1P: rows 1..15 are humans whatever N is (N > 16) and lie inside the table: no guards, no exec masks
                        const bool rh = T1P ? nd >= 1 : (nd >= 1 && nd < N);
                        if (T1P || nd < HRL) *reinterpret_cast<f32x2*>(&AB[(n * SLDK + nd) * 2]) = f32x2{rh ? rr * kRowScale : 0.f, rh ? bb : 0.f};
                    }
                if constexpr (T1P) {                             // node 16 + q (HRL == 20: rows 16..19 of the table)
                    const int nd = 16 + q;
                    const float rr = __expf(fminf(ms1 - sc1, 60.f));
                    const float bb = p1 * __builtin_amdgcn_rcpf(fmaf(rr, zs1, 1.f));
                    const bool rh = nd < N;
                    *reinterpret_cast<f32x2*>(&AB[(n * SLDK + nd) * 2]) = f32x2{rh ? rr * kRowScale : 0.f, rh ? bb : 0.f};      // nd <= 19 < HRL = 20
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int nd = 16 * nt + 4 * q + r;
                        const float pv = s0t[nt][r];
                        float al, be;
                        if (sim == SIM_SOFTMAX) {
                            const float m = fmaxf(ms[nt][r], sct[nt][r]);
                            al = __expf(ms[nt][r] - m);
                            be = __expf(sct[nt][r] - m);
                        } else {
                            al = 1.f;
                            be = plain_weight(sim, sct[nt][r], nd, 0);        // diagonal: nd >= 1 here, so 0
                        }
                        const float piz = pv * __builtin_amdgcn_rcpf(fmaf(al, zs[nt][r], be));
                        const bool rh = nd >= 1 && nd < N;
                        *reinterpret_cast<f32x2*>(&AB[(n * SLD + nd) * 2]) = f32x2{rh ? al * piz : 0.f, rh ? be * piz : 0.f};
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();
        load_fence();
        PHASE_MARK(3);

        // ---------------- row pass over my 16 children -------------------------------------------------------------------
        f32x4 tp4[2] = {zero4(), zero4()};           // PK: t_c without the robot-row / skip terms, D layout (child n, features)
        if constexpr (PK) {
            // D layout throughout: lane (n, q) = child n, features 4q..4q+3 and 16+4q..: y and the result stay in registers, the
            // child's (r_i, b_i) come as broadcast b128 reads (two nodes each), UW_i as two broadcast b128 reads; per node and feature
            // pair: v_pk_fma (r UW + y) with the ReLU in its clamp bit, v_pk_fma (acc += b relu) -- 2 instructions per 2 elements
            // (round 3: 4, with two integer-max relus; the lane = feature form spends 8: DPP broadcasts cannot feed packed operands).
            // The clamp is to [0, 1], so the pass runs on 2^-110 of its values: r (in the table) and y carry the factor, every
            // product and sum is the exact 2^-110 multiple of the unscaled one -- power-of-two scalings commute with rounding -- as
            // long as r UW + y stays below 2^110 = 1.3e33 (r <= e^60 = 1.1e26: UW up to 1e7) and terms below 2^-39 are not missed;
            // the sum is scaled back once per tile.
            f32x2 acc[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
            const f32x2 y2[4] = {f32x2{yacc[0][0], yacc[0][1]} * kRowScale, f32x2{yacc[0][2], yacc[0][3]} * kRowScale,
                                 f32x2{yacc[1][0], yacc[1][1]} * kRowScale, f32x2{yacc[1][2], yacc[1][3]} * kRowScale};
            const float* abrow = AB + n * SLDK * 2;
            // software pipeline by node pair: the loads of pair j + 1 are issued before the arithmetic of pair j; the register
            // fence after it keeps hipcc from stacking all 48 b128 loads (192 VGPRs) in front of the arithmetic
            f32x4 ab[2], ua[2][2], ub[2][2];                          // [buffer][node of the pair]
            auto load_pair = [&](int j2, int buf) {
                ab[buf] = *reinterpret_cast<const f32x4*>(&abrow[4 * j2]);              // (r, b) of nodes 2 j2, 2 j2 + 1
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * j2 + e;
                    ua[buf][e] = *reinterpret_cast<const f32x4*>(&UWs[i * 32 + 4 * q]);
                    ub[buf][e] = *reinterpret_cast<const f32x4*>(&UWs[i * 32 + 16 + 4 * q]);
                }
            };
            load_pair(0, 0);
            static_for<0, HRL / 2>([&](auto jc) {
                constexpr int j2 = decltype(jc)::value, buf = j2 & 1;
                if constexpr (j2 + 1 < HRL / 2) load_pair(j2 + 1, buf ^ 1);
                static_for<0, 2>([&](auto ec) {
                    constexpr int e = decltype(ec)::value, i = 2 * j2 + e;
                    if constexpr (i >= 1) {                                             // node 0 is the robot slot
                        const f32x2 rb = f32x2{ab[buf][2 * e], ab[buf][2 * e + 1]};       // (2^-110 r_i, b_i): one register pair
                        const f32x2 u2[4] = {f32x2{ua[buf][e][0], ua[buf][e][1]}, f32x2{ua[buf][e][2], ua[buf][e][3]},
                                             f32x2{ub[buf][e][0], ub[buf][e][1]}, f32x2{ub[buf][e][2], ub[buf][e][3]}};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            acc[k] = pk_fma_hi(rb, pk_fma_lo_clamp(rb, u2[k], y2[k]), acc[k]);
                        }
                    }
                });
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) :: "memory");
            });
            tp4[0] = f32x4{acc[0][0], acc[0][1], acc[1][0], acc[1][1]} * kRowUnscale;
            tp4[1] = f32x4{acc[2][0], acc[2][1], acc[3][0], acc[3][1]} * kRowUnscale;
        } else {
            const int hh = lane >> 5, f = lane & 31;
            constexpr int HRV = HR < 16 * NT ? HR : 16 * NT;      // the table holds 16*NT rows per child
            const int n_pairs = (n_valid + 1) >> 1;
            for (int pair = 0; pair < n_pairs; ++pair) {
                const int lc = 2 * pair + hh;                                      // child within the tile
                const float* sc_mine = AB + (lc * SLD + (lane & 15)) * 2;
                f32x2 ab[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) ab[tt] = *reinterpret_cast<const f32x2*>(&sc_mine[32 * tt]);
                const float yv = Y0[lc * XLD + f];
                float rp[4] = {0.f, 0.f, 0.f, 0.f};
                static_for<0, (HRV + 2) / 4>([&](auto gc) {
                    constexpr int i0 = 1 + 4 * decltype(gc)::value;
                    float tv[4];
                    static_for<0, 4>([&](auto kc) {
                        constexpr int ii = i0 + decltype(kc)::value;
                        if constexpr (ii < HRV) tv[ii - i0] = dpp_rowbcast_mul<(ii & 15)>(ab[ii >> 4][1], yv);
                    });
                    static_for<0, 4>([&](auto kc) {
                        constexpr int ii = i0 + decltype(kc)::value;
                        if constexpr (ii < HRV) tv[ii - i0] = dpp_rowbcast_fmac<(ii & 15)>(ab[ii >> 4][0], uw4[ii >> 2][ii & 3], tv[ii - i0]);
                    });
                    static_for<0, 4>([&](auto kc) {
                        constexpr int ii = i0 + decltype(kc)::value;
                        if constexpr (ii < HRV) rp[ii - i0] += relu1(tv[ii - i0]);
                    });
                });
                Y0[lc * XLD + f] = (rp[0] + rp[1]) + (rp[2] + rp[3]);             // t_c without the robot-row / skip terms
            }
        }
        __builtin_amdgcn_wave_barrier();
        load_fence();
        PHASE_MARK(4);

        // ---------------- robot row: H1_0 = relu(T_0 W1)(+x0), t_c += p00 * H1_0 ---------------------------------------
        f32x4 tin[2], hp[2];
        {
            f32x4 o[2] = {zero4(), zero4()};
            if constexpr (BX) {
                f32x4 tb[2];
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tb[ft][r] = fmaf(p00, xacc[ft][r], t0h[ft][r]);      // T_0 = p_c Xh + p_c[0] x0_c
                layer_mfma_b6<XD, XD, false>(w1, tb, o, lane);
            } else {
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float tb = fmaf(p00, xacc[ft][r], t0h[ft][r]);      // T_0 = p_c Xh + p_c[0] x0_c
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot)
                        o[ot] = mfma4(w1[(16 * ft + 4 * q + r) * WLD + 16 * ot + n], tb, o[ot]);
                }
            }
            load_fence();
            }
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                const f32x4 tp = PK ? tp4[ot] : *reinterpret_cast<const f32x4*>(&Y0[n * XLD + 16 * ot + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float hv = relu1(o[ot][r]);
                    if (SKIP) hv += xacc[ot][r];
                    hp[ot][r] = hv;
                    tin[ot][r] = fmaf(p00, hv, SKIP ? tp[r] + t0h[ot][r] : tp[r]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();      // the next tile's Y0 / AB writes stay behind this tile's reads
        PHASE_MARK(5);
        if (ti + 1 == t1 && wi_next < n_items) item_loads(wi_next);     // next item's loads fly under this tile's head

        if (!full && !a.inline_partial) {
            // partial last tile: leave the rows [t_c | H1_0] for the tile-packed head pass
            if (n < a.rem) {
                float* out = a.rows_left + ((size_t)p * a.rem + n) * 64;
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    *reinterpret_cast<f32x4*>(out + 16 * ot + 4 * q) = tin[ot];
                    *reinterpret_cast<f32x4*>(out + 32 + 16 * ot + 4 * q) = hp[ot];
                }
            }
            continue;
        }

        // ---------------- last GCN layer on the robot row + value head: one register-resident MFMA chain -------------------
        await_image();
        if (a.prio & 2) __builtin_amdgcn_s_setprio(2);
        const float v = head_chain<LO, D1, D2, D3, SKIP, BX>(lds, tin, hp, lane);
        if (a.prio & 2) { if ((a.prio & 1) && wave >= 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        if (q == 0 && c < A) a.value[(size_t)p * A + c] = v + hb4;
        PHASE_MARK(6);
      }
    }
    await_image();                       // waves without a head of their own (no items, partial tiles only)
    if (a.rem && !a.inline_partial) {
        // Rows of this workgroup's partial tiles (its own parents, all written by waves of THIS workgroup): scored here,
        // tile-packed over parents -- 16 parents' `stop` children fill one head tile exactly, so the head never multiplies
        // padding columns and the path needs no second launch.
        __syncthreads();
        const int n_rows = k_b * a.rem;
        for (int tt = wave; 16 * tt < n_rows; tt += kFusedWaves) {
            const int i = 16 * tt + n;
            const bool valid = i < n_rows;
            const int lp = valid ? i / a.rem : 0, k = valid ? i - lp * a.rem : 0;
            const int p = p_first + lp;
            const float* row = a.rows_left + ((size_t)p * a.rem + k) * 64;
            f32x4 tin[2], hp[2];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                tin[ot] = valid ? *reinterpret_cast<const f32x4*>(row + 16 * ot + 4 * q) : zero4();
                hp[ot] = valid ? *reinterpret_cast<const f32x4*>(row + 32 + 16 * ot + 4 * q) : zero4();
            }
            const float v = head_chain<LO, D1, D2, D3, SKIP, BX>(lds, tin, hp, lane);
            if (valid && q == 0) a.value[(size_t)p * A + 16 * a.n_full + k] = v + hb4;
        }
    }
    if (a.tail.enabled) {
        // The search's bookkeeping for the parents this workgroup owns (rgl_tail.h): every value of theirs was written by a wave
        // of this workgroup, so a workgroup barrier is all the synchronisation there is.  One wave per parent selects; at the
        // deepest level the workgroup owns the parents of whole roots and walks the back-up chain up to the root decision.
        __syncthreads();
        static_assert(fused_scratch_floats(HR, NT, SOFT) >= kTailLdsInts, "the wave's scratch holds the select step's tables");
        int* kl = reinterpret_cast<int*>(WS);                    // the wave's scratch is free now
        for (int lp = wave; lp < k_b; lp += kFusedWaves) tail_select(a.tail, p_first + lp, kl);
        if (a.tail.chain) {
            const int W = a.tail.W, lvl = a.tail.level;
            int per_deep = 1;
            for (int l = 0; l < lvl; ++l) per_deep *= W;        // parents of one root at this (the deepest) level
            const int r_first = p_first / per_deep, n_roots = k_b / per_deep;
            int per_l = per_deep;
            for (int l = lvl - 1; l >= 1; --l) {
                per_l /= W;
                __syncthreads();                                 // level l + 1's back-up values are in place
                for (int i = tid; i < n_roots * per_l; i += nthreads) tail_backup(a.tail, l, r_first * per_l + i);
            }
            __syncthreads();
            for (int base = 0; base < n_roots * kRootLanes; base += nthreads) {
                const int i = base + tid, bl = i / kRootLanes;
                tail_root(a.tail, r_first + bl, i % kRootLanes, bl < n_roots);
            }
        }
    }
    PHASE_FLUSH();
}

// Both weight images in global memory, in exactly the LDS layouts the kernels use: built ONCE per tree search (or per
// stand-alone call) by a grid of threads, then every workgroup of every level copies its image with b128 loads that are all in
// flight at once.  (Building the 97 KB image inside each persistent workgroup cost ~10 us of dependent L2 round trips per
// launch -- most of a small-batch launch.)
// One thread per image float (inverse map: which array, which element): every source load of the grid is in flight at once --
// one L2 round trip for both images instead of one per array (the array-by-array version took 7 us).
constexpr int kPackThreads = 256;

// element `idx` of the A-fragment image of W (k-major [IN][OUT]); same map as fill_frags
template <int IN, int OUT>
__device__ __forceinline__ float frag_element(const float* __restrict__ W, int idx) {
    constexpr int IT = Tiles<IN>::v;
    const int l = idx & 63, fr = idx >> 6;
    const int r = fr & 3, it = (fr >> 2) % IT, ot = (fr >> 2) / IT;
    const int in = tile_feature<IN>(it, l >> 4, r), out = frag_out_feature<OUT>(ot, l);
    return (in < IN && out < OUT) ? W[in * OUT + out] : 0.f;
}
template <int OUT>
__device__ __forceinline__ float bias_element(const float* __restrict__ b, int idx) {
    const int feat = tile_feature<OUT>(idx >> 4, (idx >> 2) & 3, idx & 3);
    return feat < OUT ? b[feat] : 0.f;
}
// element `e` of a [ROWS_PAD][LD] image of a k-major [ROWS][COLS] matrix (padding rows / columns: 0)
template <int ROWS, int COLS, int LD>
__device__ __forceinline__ float matrix_element(const float* __restrict__ W, int e) {
    const int r = e / LD, c = e - r * LD;
    return (r < ROWS && c < COLS) ? W[r * COLS + c] : 0.f;
}

// one float of the BX f3 region (BxLayout): a pair of bf16 pieces of the matrix-pipe fragments, or an f32 fragment element
template <int IN, int OUT>
__device__ __forceinline__ float bx_element(const float* __restrict__ W, int idx) {
    using BL = BxLayout<IN, OUT>;
    if (idx >= BL::p4) {                                           // partial tile, compact: row (q, lane % 4), k step = it * 4 + r
        using PC = P4Compact<BL::KP>;
        const int j = idx - BL::p4, row = j / PC::KPAD, ks = j - row * PC::KPAD;
        return ks < BL::KP ? frag_element<IN, OUT>(W, (BL::OTF * BL::IT * 4 + ks) * 64 + 16 * (row >> 2) + (row & 3)) : 0.f;
    }
    if (idx >= BL::f32) {
        const int j = idx - BL::f32, l = j & 63, slot = j >> 6;                          // slot: k steps of the output tiles, in order
        int ot, kf, it0;
        if (slot < BL::OT3 * BL::KF3) { ot = slot / BL::KF3; kf = slot - ot * BL::KF3; it0 = BL::ITB + 2; }
        else { const int s2 = slot - BL::OT3 * BL::KF3; ot = BL::OT3 + s2 / BL::KF; kf = s2 % BL::KF; it0 = BL::ITB; }
        return frag_element<IN, OUT>(W, ((ot * BL::IT + it0) * 4 + kf) * 64 + l);
    }
    int c, ot, pc, l, p;
    if (idx >= BL::b16c) {                                                               // third chunk: [ot < OT3][piece][lane]
        const int u = (idx - BL::b16c) >> 2, rest = u >> 6;
        p = idx & 3; l = u & 63; pc = rest % 3; ot = rest / 3; c = BL::NCB;
    } else {
        const int u = idx >> 2, rest = u >> 6;
        p = idx & 3; l = u & 63; pc = rest % 3; c = (rest / 3) % BL::NCB; ot = rest / 3 / BL::NCB;
    }
    const int q = l >> 4, out = frag_out_feature<OUT>(ot, l);
    bf16x2 v;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = 2 * p + k;
        const float w = W[tile_feature<IN>(2 * c + (e >> 2), q, e & 3) * OUT + out];
        const __bf16 hi = (__bf16)w;
        const float r1 = w - (float)hi;
        const __bf16 mid = (__bf16)r1;
        const __bf16 lo = (__bf16)(r1 - (float)mid);
        v[k] = pc == 0 ? hi : (pc == 1 ? mid : lo);
    }
    return __builtin_bit_cast(float, v);
}

// one float of the BX f2 region (Bx1Layout)
template <int IN, int OUT>
__device__ __forceinline__ float bx1_element(const float* __restrict__ W, int idx) {
    using BL = Bx1Layout<IN, OUT>;
    if (idx >= BL::p4) {                                           // partial tile, compact: row (q, lane % 4), k step = it * 4 + r
        using PC = P4Compact<BL::KP>;
        const int j = idx - BL::p4, row = j / PC::KPAD, ks = j - row * PC::KPAD;
        return ks < BL::KP ? frag_element<IN, OUT>(W, (BL::OTF * Tiles<IN>::v * 4 + ks) * 64 + 16 * (row >> 2) + (row & 3)) : 0.f;
    }
    const int u = idx >> 2, p = idx & 3, l = u & 63, rest = u >> 6;
    const int pc = rest % 3, ot = rest / 3, q = l >> 4, out = frag_out_feature<OUT>(ot, l);
    bf16x2 v;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = 2 * p + k;
        const float w = W[tile_feature<IN>(e >> 2, q, e & 3) * OUT + out];
        const __bf16 hi = (__bf16)w;
        const float r1 = w - (float)hi;
        const __bf16 mid = (__bf16)r1;
        const __bf16 lo = (__bf16)(r1 - (float)mid);
        v[k] = pc == 0 ? hi : (pc == 1 ? mid : lo);
    }
    return __builtin_bit_cast(float, v);
}

template <int D1, int D2, int D3, bool BX = false>
__global__ __launch_bounds__(kPackThreads) void pack_images_kernel(const FusedArgs a, float* img) {
    using LO = FusedLds<D1, D2, D3, BX>;
    const int e = blockIdx.x * kPackThreads + threadIdx.x;
    if (e >= LO::scratch) return;
    float v;
    if (e < LO::br1) v = matrix_element<9, HID, LO::WR1LD>(a.wr1, e - LO::wr1);
    else if (e < LO::wr2) v = a.br1[e - LO::br1];
    else if (e < LO::br2) {
        if constexpr (BX) v = frag_bf3_ld<HID, XD>(a.wr2, XD, XD, e - LO::wr2);
        else v = matrix_element<HID, XD, WLD>(a.wr2, e - LO::wr2);
    } else if (e < LO::wa) v = a.br2[e - LO::br2];
    else if (e < LO::w1) {
        const int k = e - LO::wa, r = k / WLD, col = k - r * WLD;
        if constexpr (BX) v = a.wa ? frag_bf3_ld<XD, XD>(a.wa, XD, XD, k) : frag_bf3_identity<XD>(k);      // gaussian: Wa = I
        else v = a.wa ? matrix_element<XD, XD, WLD>(a.wa, k) : ((col < XD && r == col) ? 1.f : 0.f);      // gaussian: Wa = I
    } else if (e < LO::wh1) {
        if constexpr (BX) v = frag_bf3_ld<XD, XD>(a.w1, XD, XD, e - LO::w1);
        else v = matrix_element<XD, XD, WLD>(a.w1, e - LO::w1);
    } else if (e < LO::bh1) v = matrix_element<5, HID, LO::WH1LD>(a.wh1, e - LO::wh1);
    else if (e < LO::wh2) v = a.bh1[e - LO::bh1];
    else if (e < LO::bh2) {
        v = matrix_element<HID, XD, WLD>(a.wh2, e - LO::wh2);
    }
    else if (e < LO::b1) v = a.bh2[e - LO::bh2];
    else if (e < LO::b2) v = bias_element<D1>(a.hb1, e - LO::b1);
    else if (e < LO::b3) v = bias_element<D2>(a.hb2, e - LO::b2);
    else if (e < LO::w4) v = bias_element<D3>(a.hb3, e - LO::b3);
    else if (e < LO::f_last) v = bias_element<D3>(a.hw4, e - LO::w4);          // w4 is [D3][1]: same padded vector layout as a bias
    else {
        if (e < LO::f1) {
            if constexpr (BX) v = frag_bf3_ld<XD, XD>(a.w_last, XD, XD, e - LO::f_last);
            else v = frag_element<XD, XD>(a.w_last, e - LO::f_last);
        }
        else if (e < LO::f2) {
            if constexpr (BX && D1 == 32) v = bx1_element<XD, 32>(a.hw1, e - LO::f1);
            else v = frag_element<XD, D1>(a.hw1, e - LO::f1);
        }
        else if (e < LO::f3) {
            if constexpr (BX) v = bx1_element<D1, D2>(a.hw2, e - LO::f2);
            else v = frag_element<D1, D2>(a.hw2, e - LO::f2);
        } else if constexpr (BX) v = bx_element<D2, D3>(a.hw3, e - LO::f3);
        else v = frag_element<D2, D3>(a.hw3, e - LO::f3);
    }
    img[e] = v;
}

// one packed image, in the layout of the kernel that will read it (mode 2: the bf16 /
// f32 hybrid of the last head matrix)
enum { kModeF32 = 0, kModeBx = 2 };      // (1 was the split-f16 mode of ABI 4..7)
inline int launch_pack_image(const FusedArgs& a, float* img, int mode, hipStream_t stream) {
    if (mode == kModeBx) {
        using LO = FusedLds<32, 100, 100, true>;
        hipLaunchKernelGGL((pack_images_kernel<32, 100, 100, true>), dim3((unsigned)((LO::scratch + kPackThreads - 1) / kPackThreads)),
                           dim3(kPackThreads), 0, stream, a, img);
    } else {
        using LO = FusedLds<32, 100, 100, false>;
        hipLaunchKernelGGL((pack_images_kernel<32, 100, 100, false>), dim3((unsigned)((LO::scratch + kPackThreads - 1) / kPackThreads)),
                           dim3(kPackThreads), 0, stream, a, img);
    }
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

// sized for the larger layout: one buffer size whatever the contraction mode
constexpr size_t max2(size_t x, size_t y) { return x > y ? x : y; }
constexpr size_t kImageFloats = max2(FusedLds<32, 100, 100, false>::scratch, FusedLds<32, 100, 100, true>::scratch);
constexpr size_t kImageBytes = (kImageFloats * sizeof(float) + 255) & ~(size_t)255;

struct FusedPlan {
    FusedArgs a;
    size_t lds_bytes;
    int grid;
    int hr, nt;
    bool bx;                       // bf16 six-term hybrid of the last head matrix (RGL_CONTRACT_BF16X6)
    int mode() const { return bx ? kModeBx : kModeF32; }
    bool ok;
};

// Which launches take the fused kernel.  Measured on MI355X (tools/kiter.py, profiles/r02_*): both organisations end up
// pipe-bound (MFMA + VALU issue, no co-execution) at a shader clock that drops with utilisation.  The fused kernel is one launch
// that reads the child rows and writes the values; from ~3 k tiles (P ~ 500 parents of 81 actions) upwards it is the faster one
// (P = 1024: 76 vs 79 us, 2048: 114 vs 131, 4096: 205 vs 233); below, the two-stage pair (8 waves share a parent, 4 waves per
// SIMD in the head) hides the per-tile latency better.  RGL_CHILDREN_FUSED=1 / RGL_CHILDREN_TWO_STAGE=1 force one (tests).
inline int fused_policy() {
    static const int pol = [] {
        const char* f = getenv("RGL_CHILDREN_FUSED");
        const char* t = getenv("RGL_CHILDREN_TWO_STAGE");
        return (f && f[0] == '1') ? 1 : ((t && t[0] == '1') ? -1 : 0);
    }();
    return pol;
}

// How the tiles of a launch are cut into work items.  A work item = G consecutive full tiles of one parent (the last group of a
// parent may be shorter, and also carries the parent's partial tile); its wave computes the parent's crowd quantities first, so
// small G repeats crowd work (~0.45 of a tile) while large G leaves waves idle when parents are few.  Two waves share a SIMD
// (pipe-bound: their loads add).
struct ItemPlan { int G, ipp, rot, grid, k_wg, inline_partial; };

inline int fused_cu_count() {
    static const int n_cu = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev);
                                 (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    return n_cu;
}

inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && e[0] ? atoi(e) : dflt;
}

// `unit`: parents are handed to workgroups in blocks that are multiples of this (the parents of one root at the deepest level when
// the kernel's tail walks the back-up chain; 1 otherwise).  Workgroup b owns parents [b k, (b + 1) k), k = the smallest multiple of
// `unit` that covers P with one workgroup per CU.  Within a workgroup the items are dealt over the 8 waves in snake passes (waves w
// and w + 4 share a SIMD); the plan simulates that dealing for one full workgroup and takes the cut with the smallest per-SIMD
// makespan (cached per shape: launches repeat).
inline ItemPlan plan_items(int P, int n_full, int rem, int unit) {
    static std::mutex mu;
    static std::unordered_map<unsigned long long, ItemPlan> cache;
    const unsigned long long key = ((unsigned long long)P << 32) ^ ((unsigned long long)unit << 16) ^
                                   ((unsigned long long)n_full << 8) ^ (unsigned long long)rem;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    // in full-tile units (instruction counts of the phases); a tile whose SIMD has no second wave to overlap with runs slower
    constexpr float kCrowd = 0.45f, kPartial = 0.45f, kHead = 0.65f, kSolo = 1.12f;
    const int n_cu = fused_cu_count();
    int k = (P + n_cu - 1) / n_cu;
    if (k < 1) k = 1;
    k = ((k + unit - 1) / unit) * unit;
    const int grid = (P + k - 1) / k;
    static const int force_g = env_int("RGL_FUSED_G", 0), force_inline = env_int("RGL_FUSED_INLINE_PARTIAL", -1);   // measurements
    ItemPlan best{1, n_full > 0 ? n_full : 1, 0, grid, k, 0};
    float best_cost = -1.f;
    for (int inl = 0; inl <= 1; ++inl) {
        if (inl && !rem) continue;
        if (force_inline >= 0 ? (inl != (force_inline && rem ? 1 : 0)) : (inl && (long)k * rem > 8)) continue;
        const int n_tiles = n_full + inl;                        // tiles that run their own head
        const int CT = n_tiles > 0 ? n_tiles : 1;
        for (int G = CT; G >= 1; --G) {
            if (force_g > 0 && G != (force_g < CT ? force_g : CT)) continue;
            const int ipp = (CT + G - 1) / G;
            if (G > 1 && (ipp - 1) * G >= CT) continue;
            const int last_tiles = n_tiles - (ipp - 1) * G;      // tiles of the last group (n_tiles == 0: 0)
            const float c_full = kCrowd + G;
            const float c_last = kCrowd + (last_tiles > 0 ? last_tiles : 0) + ((rem && !inl) ? kPartial : 0.f);
            const int rot = (ipp > 1 && c_last > c_full) ? ipp - 1 : 0;
            float wload[kFusedWaves] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const long n_items = (long)k * ipp;
            for (long i = 0; i < n_items; ++i) {
                int j = (int)(i / k) + rot;
                if (j >= ipp) j -= ipp;
                const long pass = i / kFusedWaves, r = i - pass * kFusedWaves;
                const int w = (int)((pass & 1) ? kFusedWaves - 1 - r : r);
                wload[w] += j == ipp - 1 ? c_last : c_full;
            }
            // a SIMD's two waves (w, w + 4) share the pipe while both run (their loads add) and the longer one finishes alone at
            // the solo rate: (3.45, 3.45) beats (5.45, 1.45) -- measured at P = 1024: 0.061 vs 0.063 ms (f32), 0.041 vs 0.043 (f16x3)
            float mk = 0.f;
            for (int sidx = 0; sidx < 4; ++sidx) {
                const float hi = wload[sidx] > wload[sidx + 4] ? wload[sidx] : wload[sidx + 4];
                const float lo = wload[sidx] > wload[sidx + 4] ? wload[sidx + 4] : wload[sidx];
                const float v = 2.f * lo + (hi - lo) * kSolo;
                mk = v > mk ? v : mk;
            }
            if (rem && !inl) {                                   // the tile-packed pass behind the barrier
                const long head_tiles = ((long)k * rem + 15) / 16;
                mk += kHead * (float)((head_tiles + kFusedWaves - 1) / kFusedWaves) * (head_tiles > 4 ? 2.f : 1.f) + 0.05f;
            }
            if (best_cost < 0.f || mk < best_cost - 1e-3f) { best_cost = mk; best = ItemPlan{G, ipp, rot, grid, k, inl}; }
        }
    }
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() > 8192) cache.clear();                    // callers with ever-changing parent counts: bounded memory
    cache.emplace(key, best);
    return best;
}

// mode kModeBx: the six-term bf16 products are wanted (RGL_CONTRACT_BF16X6): such a plan takes every launch size (the two-stage pair
// has no such head, and the packed image is in this kernel's layout only).
inline FusedPlan plan_fused(const RglGraph& g, const RglMlp& head, int P, int A, int H, int unit = 1, int mode = kModeF32) {
    FusedPlan pl;
    pl.ok = false;
    pl.bx = mode == kModeBx;                    // any similarity: the head does not depend on it
    if (!fast_path_enabled() || !rank1_enabled() || fused_policy() < 0) return pl;
    static const int min_tiles = env_int("RGL_FUSED_MIN_TILES", 1200);
    if (!pl.bx && fused_policy() == 0 && (long)P * ((A + 15) / 16) < min_tiles) return pl;
    if (fast_similarity_mode(g) < 0 || g.layerwise_graph || g.x_dim != XD || g.num_layer != 2) return pl;
    if (!mlp_is(g.w_r, 9, HID, XD, true) || !mlp_is(g.w_h, 5, HID, XD, true)) return pl;
    if (head_variant(head) != 0) return pl;
    const int N = H + 1;
    if (N > 32 || A < 1) return pl;
    FusedArgs& a = pl.a;
    a.N = N; a.A = A; a.P = P; a.H = H;
    pl.nt = (N + 15) / 16;
    pl.hr = N <= 8 ? 8 : (N <= 20 ? 20 : 32);
    a.SLD = 16 * pl.nt + 1;                      // rows padded to whole MFMA tiles (unconditional access), odd stride
    a.n_full = A / 16;
    a.rem = A % 16;
    a.sim = fast_similarity_mode(g);
    {
        const ItemPlan ip = plan_items(P, a.n_full, a.rem, unit);
        a.tiles_per_item = ip.G;
        a.items_per_parent = ip.ipp;
        a.rot = ip.rot;
        a.parents_per_wg = ip.k_wg;
        a.inline_partial = ip.inline_partial;
        a.tail = TailArgs{};
        pl.grid = ip.grid;
    }
    pl.lds_bytes = (size_t)((pl.bx ? FusedLds<32, 100, 100, true>::scratch : FusedLds<32, 100, 100, false>::scratch) +
                            kFusedWaves * fused_scratch_floats(pl.hr, pl.nt, a.sim == SIM_SOFTMAX) + 4) * sizeof(float);   // + the arrival counter of the image
    if (pl.lds_bytes > (size_t)rgl::kLdsBytesPerCu) return pl;
    a.wr1 = g.w_r.weight[0]; a.br1 = g.w_r.bias[0]; a.wr2 = g.w_r.weight[1]; a.br2 = g.w_r.bias[1];
    a.wa = bilinear_wa(g); a.w1 = g.Ws[0];
    a.w_last = g.Ws[1];
    a.hw1 = head.weight[0]; a.hb1 = head.bias[0]; a.hw2 = head.weight[1]; a.hb2 = head.bias[1];
    a.hw3 = head.weight[2]; a.hb3 = head.bias[2]; a.hw4 = head.weight[3]; a.hb4 = head.bias[3];
    a.wh1 = g.w_h.weight[0]; a.bh1 = g.w_h.bias[0]; a.wh2 = g.w_h.weight[1]; a.bh2 = g.w_h.bias[1];
    pl.ok = true;
    return pl;
}

template <int HR, int NT, bool SKIP, bool SOFT, bool BX = false>
int launch_fused_ts(const FusedPlan& pl, hipStream_t st) {
    auto kern = children_fused_kernel<HR, NT, SKIP, SOFT, 32, 100, 100, BX>;
    RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)pl.lds_bytes));
    const int grid = pl.grid;                        // persistent, one 8-wave workgroup per CU (LDS-bound)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kFusedWaves * 64), pl.lds_bytes, st, pl.a);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

template <int HR, int NT, bool SKIP>
int launch_fused_t(const FusedPlan& pl, hipStream_t st) {
    if (pl.bx) return pl.a.sim == SIM_SOFTMAX ? launch_fused_ts<HR, NT, SKIP, true, true>(pl, st)
                                              : launch_fused_ts<HR, NT, SKIP, false, true>(pl, st);
    return pl.a.sim == SIM_SOFTMAX ? launch_fused_ts<HR, NT, SKIP, true>(pl, st) : launch_fused_ts<HR, NT, SKIP, false>(pl, st);
}

inline int launch_fused(const FusedPlan& pl, bool skip, hipStream_t st) {
    switch (pl.hr) {
        case 8: return skip ? launch_fused_t<8, 1, true>(pl, st) : launch_fused_t<8, 1, false>(pl, st);
        case 20: return pl.nt == 1 ? (skip ? launch_fused_t<20, 1, true>(pl, st) : launch_fused_t<20, 1, false>(pl, st))
                                   : (skip ? launch_fused_t<20, 2, true>(pl, st) : launch_fused_t<20, 2, false>(pl, st));
        default: return skip ? launch_fused_t<32, 2, true>(pl, st) : launch_fused_t<32, 2, false>(pl, st);
    }
}

}  // namespace

#ifdef RGL_PHASE_TIMING
extern "C" int rgl_debug_read_fused_phase_cycles(unsigned long long* out16, int reset) {
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

// workspace: rows of the partial tiles [P][A % 16][64] | ... | weight image (at the END)
size_t fused_children_workspace_bytes(int P, int A, int H) {
    (void)H;
    const size_t main_bytes = (size_t)P * (A % 16) * 64 * sizeof(float);
    return ((main_bytes + 255) & ~(size_t)255) + kImageBytes;
}

static inline float* image_of(void* workspace, size_t workspace_bytes) {
    return reinterpret_cast<float*>((char*)workspace + ((workspace_bytes - kImageBytes) & ~(size_t)255));
}

const float* fused_workspace_image(const void* workspace, size_t workspace_bytes) {
    return image_of(const_cast<void*>(workspace), workspace_bytes);
}

// 1 = the fused kernel does not apply (or the workspace cannot hold its images)
int pack_children_images(const RglGraph* g, const RglMlp* head, int P, int A, int H, void* workspace, size_t workspace_bytes,
                         hipStream_t stream, int mode) {
    FusedPlan fp = plan_fused(*g, *head, P, A, H, 1, mode);                    // kModeF32 / kModeBx
    if (!fp.ok || !workspace || workspace_bytes < fused_children_workspace_bytes(P, A, H)) return 1;
    return launch_pack_image(fp.a, image_of(workspace, workspace_bytes), fp.mode(), stream);
}

// 1 = outside this kernel's envelope
int launch_fused_children(const RglGraph* g, const RglMlp* head, int P, int A, int H, const float* child_robot,
                          const float* humans_next, float* child_value, void* workspace, size_t workspace_bytes,
                          int image_ready, hipStream_t stream, const float* caller_image, const void* tail, size_t tail_bytes,
                          int* tail_done, int mode) {
    if (tail_done) *tail_done = 0;
    // the search's tail: selection always; the back-up chain + root step at the deepest level when handing whole roots to
    // workgroups does not starve the GPU (few roots with many parents each -- unclipped deep searches -- keep unit = 1)
    const TailArgs* ta = (tail && tail_bytes == sizeof(TailArgs) && ((const TailArgs*)tail)->enabled) ? (const TailArgs*)tail : nullptr;
    static const bool tail_off = env_int("RGL_FUSED_NO_TAIL", 0) != 0;          // measurements / tests: stand-alone kernels
    if (tail_off) ta = nullptr;
    int unit = 1, chain = 0;
    if (ta && ta->chain) {
        long u = 1;
        for (int l = 0; l < ta->level && u <= P; ++l) u *= ta->W;
        if (u <= P && P % u == 0 && P / u >= fused_cu_count() / 2) {
            unit = (int)u;
            chain = 1;
        } else if (u == 1) {
            chain = 1;
        }
    }
    FusedPlan fp = plan_fused(*g, *head, P, A, H, unit, mode);
    // the bf16 hybrid image is larger: crowds of 21..32 agents (lane = feature row pass, larger wave scratch) do not fit a CU
    // with it -- they run the f32 form of this kernel on an f32 image packed here (the caller's image is in the other layout)
    bool own_image = false;
    if (!fp.ok && mode) {
        fp = plan_fused(*g, *head, P, A, H, unit, kModeF32);
        own_image = fp.ok;
    }
    if (!fp.ok) return 1;
    if (!workspace || workspace_bytes < fused_children_workspace_bytes(P, A, H)) return 1;
    if (own_image || (!image_ready && !caller_image)) {
        int rc = launch_pack_image(fp.a, image_of(workspace, workspace_bytes), fp.mode(), stream);
        if (rc) return rc;
    }
    const float* image = (caller_image && !own_image) ? caller_image : image_of(workspace, workspace_bytes);
    float* rows_left = (float*)workspace;
    fp.a.child_robot = child_robot;
    fp.a.humans = humans_next;
    fp.a.value = child_value;
    fp.a.image = image;
    fp.a.rows_left = rows_left;
    static const int image_sync = env_int("RGL_FUSED_IMAGE_SYNC", 0);
    fp.a.image_sync = image_sync;
    static const int phase_delay = env_int("RGL_FUSED_PHASE_DELAY", 0), prio = env_int("RGL_FUSED_PRIO", 2);
    fp.a.phase_delay = phase_delay;
    fp.a.prio = prio;
    if (ta) {
        fp.a.tail = *ta;
        fp.a.tail.chain = chain;
    }
    const int rc = launch_fused(fp, g->skip_connection != 0, stream);
    if (rc == RGL_OK && ta && tail_done) *tail_done = chain ? 2 : 1;
    return rc;
}

}  // namespace rgl

// The image depends on the weights (and on the contraction mode: three-piece bf16 fragments for RGL_CONTRACT_BF16X6) only: a caller
// with fixed weights packs it once (MprlPlanner::children_image).
// Outside the fused kernel's envelope (three layers, crowds beyond 32 agents) the image still serves the stage-2 head -- launched
// on its own or run by children_deep_kernel behind its parents -- as long as the shipped head and embedding shapes are there: the
// same f32 layout, `w_last` = the graph's last layer.
static bool head_image_args(const RglGraph& g, const RglMlp& head, FusedArgs& a) {
    if (!fast_path_enabled() || g.x_dim != XD || g.num_layer < 2 || g.num_layer > RGL_MAX_GCN_LAYERS) return false;
    if (!mlp_is(g.w_r, 9, HID, XD, true) || !mlp_is(g.w_h, 5, HID, XD, true) || head_variant(head) != 0) return false;
    a = FusedArgs{};
    a.wr1 = g.w_r.weight[0]; a.br1 = g.w_r.bias[0]; a.wr2 = g.w_r.weight[1]; a.br2 = g.w_r.bias[1];
    a.wa = bilinear_wa(g); a.w1 = g.Ws[0];
    a.w_last = g.Ws[g.num_layer - 1];
    a.hw1 = head.weight[0]; a.hb1 = head.bias[0]; a.hw2 = head.weight[1]; a.hb2 = head.bias[1];
    a.hw3 = head.weight[2]; a.hb3 = head.bias[2]; a.hw4 = head.weight[3]; a.hb4 = head.bias[3];
    a.wh1 = g.w_h.weight[0]; a.bh1 = g.w_h.bias[0]; a.wh2 = g.w_h.weight[1]; a.bh2 = g.w_h.bias[1];
    return true;
}

static int image_mode_of(const MprlPlanner* planner) {
    return planner->contraction_dtype == RGL_CONTRACT_BF16X6 ? kModeBx : kModeF32;
}

extern "C" size_t mprl_children_image_bytes(const MprlPlanner* planner) {
    if (!planner) return 0;
    const int mode = image_mode_of(planner);
    if (plan_fused(planner->value_graph, planner->value_head, 4096, 16, 1, 1, mode).ok) return kImageBytes;      // architecture test only
    FusedArgs a;
    return (mode == kModeF32 && head_image_args(planner->value_graph, planner->value_head, a)) ? kImageBytes : 0;
}

extern "C" int mprl_pack_children_image_f32(const MprlPlanner* planner, float* image, size_t image_bytes, rgl_stream_t stream) {
    if (!planner || !image) return RGL_ERR_NULL;
    const int mode = image_mode_of(planner);
    FusedPlan fp = plan_fused(planner->value_graph, planner->value_head, 4096, 16, 1, 1, mode);
    if (!fp.ok) {
        if (mode != kModeF32 || !head_image_args(planner->value_graph, planner->value_head, fp.a)) return RGL_ERR_BAD_MODE;
        fp.bx = false;
    }
    if (image_bytes < kImageBytes) return RGL_ERR_WORKSPACE;
    return launch_pack_image(fp.a, image, fp.mode(), (hipStream_t)stream);
}
