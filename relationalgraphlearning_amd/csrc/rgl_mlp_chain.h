// MFMA chains for small MLPs in "transposed form" (activations = B operand, kept in registers; weights = pre-permuted A
// fragments in LDS): shared by the stage-2 head kernel (rgl_head.hip) and the row-embedding kernel (rgl_scene.hip).
#pragma once
#include "rgl_mfma.h"

namespace {

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

// A partial LAST OUTPUT tile of at most four features (the 100-wide layers of the value heads: 6 tiles + 4 features) is not worth a
// 16 x 16 x 4 MFMA per k step -- 12 of its 16 rows are padding.  Those features are produced by v_mfma_f32_4x4x1_16B_f32 instead
// (round 4; tools/micro/mfma_4x4.hip: 12 clocks against 32, layout checked on the hardware): 16 independent 4 x 4 x 1 blocks, block
// = lane / 4.  With the activations in the D layout (lane (n, q): child n, k slot q) the B operand is the very register the
// 16 x 16 product reads: block (q, n / 4) multiplies the four features (A row = lane % 4) with the children 4 (n / 4) .. + 3 for
// ITS k slot; the four k-group partial sums of feature i sit in register i of the lanes (n, q = 0..3) and meet in a
// reduce-scatter of three permlane swaps that leaves feature 16 t + q in register 0 of lane (n, q) -- exactly where
// tile_feature() puts a partial tile's features.
template <int OUT>
struct Partial4 { static constexpr bool v = (OUT % 16) != 0 && (OUT % 16) <= 4; };
// output feature multiplied by A-operand lane l of the fragments of output tile ot
template <int OUT>
__device__ __forceinline__ int frag_out_feature(int ot, int l) {
    if (Partial4<OUT>::v && ot == Tiles<OUT>::v - 1) return 16 * ot + (l & 3);
    const int m = l & 15;                                            // A-operand row = D row of the output tile
    return tile_feature<OUT>(ot, m >> 2, m & 3);
}
// NTHR > 0: the workgroup size as a compile-time constant -- the loops are fully unrolled and every global load of a thread is
// in flight at once (one L2 round trip for the whole image instead of one per batch of 8: the 74 KB image of the value head
// took ~10 us per workgroup to build, which is most of a small launch).
template <int IN, int OUT, int NTHR = 0>
__device__ __forceinline__ void fill_frags(float* dst, const float* __restrict__ W, int tid, int nthr = kThreads) {
    constexpr int IT = Tiles<IN>::v, OT = Tiles<OUT>::v;
    constexpr int TOTAL = OT * IT * 4 * 64;
    auto element = [&](int idx, float& w, bool& ok) {
        const int l = idx & 63, fr = idx >> 6;
        const int r = fr & 3, it = (fr >> 2) % IT, ot = (fr >> 2) / IT;
        const int in = tile_feature<IN>(it, l >> 4, r), out = frag_out_feature<OUT>(ot, l);
        w = W[(in < IN ? in : IN - 1) * OUT + (out < OUT ? out : OUT - 1)];   // unconditional load (batched), then mask
        ok = in < IN && out < OUT;
    };
    if constexpr (NTHR > 0) {
        constexpr int ITERS = (TOTAL + NTHR - 1) / NTHR;
        float w[ITERS];
        bool ok[ITERS];
#pragma unroll
        for (int k = 0; k < ITERS; ++k) {
            const int idx = tid + k * NTHR;
            element(idx < TOTAL ? idx : TOTAL - 1, w[k], ok[k]);
        }
#pragma unroll
        for (int k = 0; k < ITERS; ++k) {
            const int idx = tid + k * NTHR;
            if (idx < TOTAL) dst[idx] = ok[k] ? w[k] : 0.f;
        }
    } else {
#pragma unroll 8
        for (int idx = tid; idx < TOTAL; idx += nthr) {
            float w;
            bool ok;
            element(idx, w, ok);
            dst[idx] = ok ? w : 0.f;
        }
    }
}

// fill_frags in two halves, so that a caller can put the loads of SEVERAL images in flight before the first store (one L2 round trip
// for all of them: the scene kernel's embedding sets behind its weight image)
template <int IN, int OUT, int NTHR>
struct FragRegs {
    static constexpr int TOTAL = Tiles<OUT>::v * Tiles<IN>::v * 4 * 64;
    static constexpr int ITERS = (TOTAL + NTHR - 1) / NTHR;
    float w[ITERS];
    bool ok[ITERS];
};
template <int IN, int OUT, int NTHR>
__device__ __forceinline__ void frag_load(FragRegs<IN, OUT, NTHR>& r, const float* __restrict__ W, int tid) {
    constexpr int IT = Tiles<IN>::v;
    using R = FragRegs<IN, OUT, NTHR>;
#pragma unroll
    for (int k = 0; k < R::ITERS; ++k) {
        int idx = tid + k * NTHR;
        if (idx >= R::TOTAL) idx = R::TOTAL - 1;
        const int l = idx & 63, fr = idx >> 6;
        const int rr = fr & 3, it = (fr >> 2) % IT, ot = (fr >> 2) / IT;
        const int in = tile_feature<IN>(it, l >> 4, rr), out = frag_out_feature<OUT>(ot, l);
        r.w[k] = W[(in < IN ? in : IN - 1) * OUT + (out < OUT ? out : OUT - 1)];
        r.ok[k] = in < IN && out < OUT;
    }
}
template <int IN, int OUT, int NTHR>
__device__ __forceinline__ void frag_store(const FragRegs<IN, OUT, NTHR>& r, float* dst, int tid) {
    using R = FragRegs<IN, OUT, NTHR>;
#pragma unroll
    for (int k = 0; k < R::ITERS; ++k) {
        const int idx = tid + k * NTHR;
        if (idx < R::TOTAL) dst[idx] = r.ok[k] ? r.w[k] : 0.f;
    }
}
// one element of a fill_bias vector per thread (threads >= Tiles<OUT> * 16 hold nothing): load, then store
template <int OUT>
__device__ __forceinline__ float bias_load(const float* __restrict__ b, int tid) {
    const int idx = tid < Tiles<OUT>::v * 16 ? tid : 0;
    const int feat = tile_feature<OUT>(idx >> 4, (idx >> 2) & 3, idx & 3);
    const float v = b[feat < OUT ? feat : OUT - 1];
    return feat < OUT ? v : 0.f;
}
template <int OUT>
__device__ __forceinline__ void bias_store(float v, float* dst, int tid) {
    if (tid < Tiles<OUT>::v * 16) dst[tid] = v;
}

// per-feature vectors (bias, last-layer weights) in D-row order: dst[16 t + 4 q + r] belongs to tile_feature(t, q, r)
template <int OUT>
__device__ __forceinline__ void fill_bias(float* dst, const float* __restrict__ b, int tid, int nthr = kThreads) {
#pragma unroll 2
    for (int idx = tid; idx < Tiles<OUT>::v * 16; idx += nthr) {
        const int feat = tile_feature<OUT>(idx >> 4, (idx >> 2) & 3, idx & 3);
        const float v = b[feat < OUT ? feat : OUT - 1];
        dst[idx] = feat < OUT ? v : 0.f;
    }
}

// k-major [ROWS][COLS] matrix -> LDS image with row stride LD (rows >= ROWS of the ROWS_PAD-row image are zero), all loads of a
// thread in flight at once
template <int ROWS, int ROWS_PAD, int COLS, int LD, int NTHR>
__device__ __forceinline__ void fill_matrix(float* dst, const float* __restrict__ W, int tid) {
    constexpr int TOTAL = ROWS_PAD * COLS;
    constexpr int ITERS = (TOTAL + NTHR - 1) / NTHR;
    float w[ITERS];
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
        const int i = tid + k * NTHR;
        w[k] = W[i < ROWS * COLS ? i : 0];
    }
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
        const int i = tid + k * NTHR;
        const int r = i / COLS, c = i - r * COLS;
        if (i < TOTAL) dst[r * LD + c] = i < ROWS * COLS ? w[k] : 0.f;
    }
}

// The weight image of the shipped value estimator (w_r, w_h, Wa, W1, value-head vectors and A fragments) in the LDS layout of
// children_fused_kernel (rgl_fused.hip); pack_images_kernel writes it to global memory once per parameter state.  The two-stage pair
// takes its parts from the same image when one is at hand: children_rank1_kernel the first `b1` floats, robot_head_kernel the rest.
template <int IN, int OUT>
struct HeadFragFloats {
    static constexpr int v = Tiles<OUT>::v * Tiles<IN>::v * 4 * 64;
};

// BX (round 5, RGL_CONTRACT_BF16X6): the D2 x D3 matrix of the value head -- 40 % of a tile's MFMA cycles -- with its first 64 input
// features on the MATRIX pipe at full f32 operand width: every operand is three bf16 pieces by round-to-nearest (x = hi + mid + lo
// EXACTLY: 8 + 8 + 8 significand bits, bf16 has f32's exponent range, no scaling), a K = 32 block is the six terms
//   W_lo a_hi + W_mid a_mid + W_hi a_lo + W_mid a_hi + W_hi a_mid + W_hi a_hi          (v_mfma_f32_16x16x32_bf16, f32 accumulate)
// and the three dropped ones (W_mid a_lo + W_lo a_mid + W_lo a_lo) are at most 2^-23 |W||a| in the worst case (|mid| <= 2^-8 |x|,
// |lo| <= 2^-16 |x|), ~2^-25 typically -- the size of the rounding of one f32 product, and unbiased (signed pieces).  6 x 16 clocks replace 8 x 32 per 16 x 16 x 32 block, and unlike the f32 MFMA they do not
// occupy the vector ALUs.  Why not the whole head: three bf16 pieces are 6 bytes per weight, and the kernel's LDS
// (106 KB image + 42 KB wave scratch of 160 KB) had 11.5 KB to spare; w_h's second matrix moved into registers (FusedLds::bh2) frees
// 9 KB more.  That holds three K = 32 chunks of this matrix's six full output tiles and the 32 x 100 layer before it (Bx1Layout);
// input features 96.., the partial output tiles and the two 32 x 32 layers stay on the f32 MFMA
// (profiles/r05_micro_bf16x3_split.txt: the whole head in this form would be 1.67x, and needs +44 KB).
// Layout of the f3 region: [ot < OTF][chunk < 2][hi | mid | lo][lane] x 8 bf16  |  f32 fragments [ot < OTF][k step of input tiles
// 4..][lane]  |  the partial output tile's 4 x 4 x 1 fragments, compact (P4Compact).
// The A operand of a 4 x 4 x 1 block row holds 16 distinct values per k step -- A row = lane % 4, k slot = lane / 16; the four blocks
// n / 4 of a k-group repeat them -- so a partial output tile's fragments are stored COMPACT (round 6): row (q, lane % 4) of KPAD
// floats, k steps consecutive, read as b128 (four k steps per load, the rows of a 16-lane access group on disjoint banks).  A quarter of the 64-floats-per-k-step form and a quarter of its LDS reads.
template <int KP>
struct P4Compact {
    static constexpr int KV = (KP + 3) / 4;                         // b128 loads per lane
    // a 16-lane access group of ds_read_b128 meets 8 distinct rows: their 4-bank windows (row * KPAD mod 64) must not overlap
    __host__ __device__ static constexpr bool disjoint(int kpad) {
        for (int r = 0; r < 8; ++r)
            for (int t = r + 1; t < 8; ++t) {
                const int d = ((t - r) * kpad) % 64;
                if (d < 4 || d > 60) return false;
            }
        return true;
    }
    static constexpr int KPAD = disjoint(4 * KV) ? 4 * KV : (disjoint(4 * KV + 4) ? 4 * KV + 4 : 4 * KV + 8);
    static_assert(disjoint(KPAD), "bank-conflict-free row stride");
    static constexpr int floats = 16 * KPAD;
    __host__ __device__ static constexpr int row_of_lane(int lane) { return 4 * (lane >> 4) + (lane & 3); }
};

template <int IN, int OUT>
struct BxLayout {
    static constexpr int IT = Tiles<IN>::v, OT = Tiles<OUT>::v;
    static constexpr bool P4 = Partial4<OUT>::v;
    static constexpr int OTF = P4 ? OT - 1 : OT;
    static constexpr int NCB = 2, ITB = 2 * NCB;                                         // bf16 chunks every output tile gets; input tiles they cover
    // ... and a THIRD chunk (input tiles 4, 5) for the first OT3 output tiles (all of them since w_h's second matrix left the LDS)
    static constexpr int OT3 = (IT >= ITB + 3) ? OTF : 0;
    static constexpr int KF = (IT - ITB - 1) * 4 + LastTileSteps<IN>::v;                 // f32 k steps of an output tile without the third chunk
    static constexpr int KF3 = (IT - ITB - 3) * 4 + LastTileSteps<IN>::v;                // ... with it
    static constexpr int KP = (IT - 1) * 4 + LastTileSteps<IN>::v;                       // k steps of the partial output tile
    static constexpr int b16 = 0;                                                        // [ot][chunk < NCB][piece][lane] x 8 bf16
    static constexpr int b16c = b16 + OTF * NCB * 3 * 64 * 4;                            // [ot < OT3][piece][lane] x 8 bf16: the third chunk
    static constexpr int f32 = b16c + OT3 * 3 * 64 * 4;
    __host__ __device__ static constexpr int f32_of(int ot) { return f32 + (ot < OT3 ? ot * KF3 : OT3 * KF3 + (ot - OT3) * KF) * 64; }
    static constexpr int p4 = f32 + (OT3 * KF3 + (OTF - OT3) * KF) * 64;
    static constexpr int total = p4 + (P4 ? P4Compact<KP>::floats : 0);
    static_assert(IT > ITB, "input tiles 0..3 are full tiles and there is at least one tile beyond them");
};

// The same for a layer whose whole input is ONE K = 32 chunk (the 32 -> 100 head layer): [ot < OTF][hi | mid | lo][lane] x 8 bf16, then
// the partial output tile's 4 x 4 x 1 fragments, compact (f32).
template <int IN, int OUT>
struct Bx1Layout {
    static_assert(IN == 32, "one chunk");
    static constexpr int OT = Tiles<OUT>::v;
    static constexpr bool P4 = Partial4<OUT>::v;
    static constexpr int OTF = P4 ? OT - 1 : OT;
    static constexpr int KP = IN / 4;
    static constexpr int b16 = 0;
    static constexpr int p4 = b16 + OTF * 3 * 64 * 4;
    static constexpr int total = p4 + (P4 ? P4Compact<KP>::floats : 0);
};

template <int D1, int D2, int D3, bool BX = false>
struct FusedLds {
    // child-side weight image
    static constexpr int wr1 = 0;
    // BX (round 6): the first embedding matrices hold their VALID rows only (9 of 12, 5 of 8) at row stride 72 -- the k slots past the
    // input width multiply an input that is exactly 0, so their A operand may be any finite weight: the lanes re-read the last valid row
    // -- which frees the 512 floats W_last's three-piece bf16 fragments need beyond its f32 ones
    static constexpr int WR1LD = BX ? 72 : W1LD, WR1ROWS = BX ? 9 : 12;
    static constexpr int br1 = wr1 + WR1ROWS * WR1LD;
    static constexpr int wr2 = br1 + HID;
    // BX (round 6): wr2, wa, w1 as three-piece bf16 fragments (layer_mfma_b6's layout: 6 bytes per weight instead of 4.5 with the
    // padded f32 rows) -- the room comes from the compact partial-tile fragments of the head (P4Compact)
    static constexpr int br2 = wr2 + (BX ? HID * XD * 3 / 2 : HID * WLD);
    static constexpr int wa = br2 + XD;
    static constexpr int w1 = wa + (BX ? XD * XD * 3 / 2 : XD * WLD);
    // crowd side: w_h
    static constexpr int wh1 = w1 + (BX ? XD * XD * 3 / 2 : XD * WLD);
    static constexpr int WH1LD = BX ? 72 : W1LD;                    // BX: w_h's first matrix at row stride 72 (k-groups 8 banks apart: two
                                                                   // lanes per bank, as at 80) -- the 64 floats its image needs for ...
    static constexpr int WH1ROWS = BX ? 5 : 8;
    static constexpr int bh1 = wh1 + WH1ROWS * WH1LD;
    static constexpr int wh2 = bh1 + HID;
    static constexpr int bh2 = wh2 + (BX ? 0 : HID * WLD);          // BX: w_h's second matrix lives in REGISTERS (32 per lane, loaded once
                                                                   // per kernel: every crowd computation reads the same 32), its 9 KB
                                                                   // of LDS hold bf16 fragments of the head instead
    // value head: per-feature vectors, then the A fragments (layout of rgl_head.hip).  Everything addressed with many different
    // lane patterns sits below 64 KB (the reach of a ds instruction's immediate offset from a shared base register); the large
    // f3 image, addressed with one pattern, spans the boundary.
    static constexpr int b1 = bh2 + XD;
    static constexpr int b2 = b1 + Tiles<D1>::v * 16;
    static constexpr int b3 = b2 + Tiles<D2>::v * 16;
    static constexpr int w4 = b3 + Tiles<D3>::v * 16;
    static constexpr int f_last = w4 + Tiles<D3>::v * 16;
    static constexpr int f1 = f_last + (BX ? XD * XD * 3 / 2 : HeadFragFloats<XD, XD>::v);      // BX: W_last as three-piece bf16 fragments (layer_mfma_b6)
    static constexpr int f2 = f1 + ((BX && D1 == 32) ? Bx1Layout<XD, 32>::total : HeadFragFloats<XD, D1>::v);    // ... the XD x D1 layer as bf16 pieces
    static constexpr int f3 = f2 + (BX ? Bx1Layout<D1, D2>::total : HeadFragFloats<D1, D2>::v);
    static constexpr int scratch = f3 + (BX ? BxLayout<D2, D3>::total : HeadFragFloats<D2, D3>::v);      // per wave: fused_scratch_floats()
};

// LDS image <- the same image prepared in global memory: b128 copies, every load of a thread in flight at once
template <int NFLOATS, int NTHR>
__device__ __forceinline__ void copy_image(float* dst, const float* __restrict__ src, int tid) {
    static_assert(NFLOATS % 4 == 0, "images are whole float4s");
    constexpr int TOTAL = NFLOATS / 4;
    constexpr int ITERS = (TOTAL + NTHR - 1) / NTHR;
    f32x4 v[ITERS];
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
        const int i = tid + k * NTHR;
        v[k] = reinterpret_cast<const f32x4*>(src)[i < TOTAL ? i : TOTAL - 1];
    }
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
        const int i = tid + k * NTHR;
        if (i < TOTAL) reinterpret_cast<f32x4*>(dst)[i] = v[k];
    }
}

// out = W^T in (+ bias): the accumulators START at the bias (no zero-init moves, no add afterwards).  A partial last output tile
// of <= 4 features runs on the 4 x 4 x 1 16-block MFMA (Partial4 above; its fragments take the place of that tile's).
template <int IN, int OUT, bool BIAS>
__device__ __forceinline__ void layer_mfma(const float* frags, const f32x4 (&in)[Tiles<IN>::v], f32x4 (&out)[Tiles<OUT>::v],
                                           int lane, const float* bias = nullptr) {
    constexpr int IT = Tiles<IN>::v, OT = Tiles<OUT>::v;
    constexpr bool P4 = Partial4<OUT>::v;
    constexpr int OTF = P4 ? OT - 1 : OT;                 // output tiles on the 16 x 16 x 4 MFMA
    const int q = lane >> 4;
    // BIAS is a template flag, not a pointer test: with a run-time test hipcc zero-initialises every accumulator first
#pragma unroll
    for (int ot = 0; ot < OTF; ++ot) {
        if constexpr (BIAS) out[ot] = *reinterpret_cast<const f32x4*>(&bias[16 * ot + 4 * q]);
        else out[ot] = zero4();
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        load_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (it == IT - 1 && r >= LastTileSteps<IN>::v) continue;          // k steps over padding only
#pragma unroll
            for (int ot = 0; ot < OTF; ++ot) out[ot] = mfma4(frags[((ot * IT + it) * 4 + r) * 64 + lane], in[it][r], out[ot]);
        }
    }
    load_fence();
    if constexpr (P4) {
        // the <= 4 features of the partial tile: a pass of its own (in the k loop above its operand loads raised the kernel's
        // register pressure past 256), two accumulators so that consecutive 4 x 4 x 1 MFMAs do not wait for each other
        f32x4 p4[2] = {zero4(), zero4()};
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            load_fence();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (it == IT - 1 && r >= LastTileSteps<IN>::v) continue;
                p4[r & 1] = mfma4x4(frags[((OTF * IT + it) * 4 + r) * 64 + lane], in[it][r], p4[r & 1]);
            }
        }
        load_fence();
        float v = kgroups_reduce_scatter(p4[0] + p4[1]);
        if constexpr (BIAS) v += bias[16 * OTF + 4 * q];
        out[OTF] = f32x4{v, 0.f, 0.f, 0.f};
    }
}

// ---- BX: three bf16 pieces per operand, six terms (BxLayout above) -----------------------------------------------------------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Split3 { bf16x8 h, m, l; };

// pieces of the D-layout tiles 2C, 2C + 1 as one K = 32 B-operand chunk (slot (q, e): tile 2C + e / 4, register e % 4 -- the D
// registers of the previous layer, packed pairwise, ARE the operand; v_cvt_pk_bf16_f32 rounds to nearest even)
template <int IT, int C>
__device__ __forceinline__ Split3 split3_rn(const f32x4 (&in)[IT]) {
    u32x4 H, M, L;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int t = 2 * C + (p >> 1), r0 = 2 * (p & 1);
        const f32x2 x = f32x2{in[t][r0], in[t][r0 + 1]};
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        const f32x2 r1 = x - __builtin_convertvector(h, f32x2);             // exact
        const bf16x2 m = __builtin_convertvector(r1, bf16x2);
        const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);            // exact, at most 7 significant bits left
        const bf16x2 l = __builtin_convertvector(r2, bf16x2);               // exact
        H[p] = __builtin_bit_cast(unsigned, h);
        M[p] = __builtin_bit_cast(unsigned, m);
        L[p] = __builtin_bit_cast(unsigned, l);
    }
    Split3 s;
    s.h = __builtin_bit_cast(bf16x8, H);
    s.m = __builtin_bit_cast(bf16x8, M);
    s.l = __builtin_bit_cast(bf16x8, L);
    return s;
}

// pieces of ONE pair of D-layout tiles (ta, tb) -> a K = 32 chunk
__device__ __forceinline__ Split3 split3_pair(const f32x4& ta, const f32x4& tb) {
    const f32x4 pair[2] = {ta, tb};
    return split3_rn<2, 0>(pair);
}

// The <= 4 features of a partial output tile from compact fragments (P4Compact): lane (n, q) gets feature 16 OTF + q's pre-bias sum in
// its return value.  k step ks = 4 it + r (steps over the padding of a partial last INPUT tile are not stored); two accumulators so
// that consecutive 4 x 4 x 1 MFMAs do not wait for each other; the loads go out in batches of four b128 (16 VGPRs).
template <int IN, int KP>
__device__ __forceinline__ float partial4_compact(const float* frags, const f32x4 (&in)[Tiles<IN>::v], int lane) {
    using PC = P4Compact<KP>;
    constexpr int IT = Tiles<IN>::v;
    const f32x4* row = reinterpret_cast<const f32x4*>(frags + PC::row_of_lane(lane) * PC::KPAD);
    f32x4 p4[2] = {zero4(), zero4()};
#pragma unroll
    for (int v0 = 0; v0 < PC::KV; v0 += 4) {
        load_fence();
        f32x4 w[4];
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (v0 + v < PC::KV) w[v] = row[v0 + v];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ks = 4 * (v0 + v) + r;                      // == 4 it + r: full input tiles before the last take 4 steps each
                if (v0 + v < PC::KV && ks < KP) p4[r & 1] = mfma4x4(w[v][r], in[(v0 + v) < IT ? (v0 + v) : 0][r], p4[r & 1]);
            }
    }
    load_fence();
    return kgroups_reduce_scatter(p4[0] + p4[1]);
}

// pieces of one pair of D-layout elements (a "unit" of a K = 32 chunk: word p of the hi / mid / lo operands)
__device__ __forceinline__ void split3_unit(f32x2 x, unsigned& H, unsigned& M, unsigned& L) {
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    H = __builtin_bit_cast(unsigned, h);
    M = __builtin_bit_cast(unsigned, m);
    L = __builtin_bit_cast(unsigned, l);
}

// The D2 x D3 head layer (BxLayout) as a software pipeline (round 6).  Register budget: the kernel around it holds ~100 VGPRs of crowd
// quantities across the tile loop; the partial output tile goes FIRST (it reads every input tile), then one chunk at a time, two
// output tiles per fragment load, then the f32 k steps that are left.  tools/micro/pipe_overlap.hip: ONE wave that issues up to two plain VALU instructions
// behind each v_mfma_f32_16x16x32_bf16 gets them for free (16.7 / 18.8 cycles per group with one / two against 16.2 for the MFMA alone),
// while the partner wave of the SIMD cannot fill the matrix pipe's shadow (the older wave's next MFMA blocks the issue port).  So the
// split of chunk c + 1 (44 VALU) and the fragment loads of the next pair of output tiles are issued BETWEEN the twelve MFMAs of a
// pair of output tiles of chunk c (sched_group_barrier pins the order); only chunk 0's split stays in front.
template <int IN, int OUT, bool BIAS>
__device__ __forceinline__ void layer_mfma_bx(const float* frags, const f32x4 (&in)[Tiles<IN>::v], f32x4 (&out)[Tiles<OUT>::v],
                                                   int lane, const float* bias = nullptr) {
    using BL = BxLayout<IN, OUT>;
    constexpr int IT = BL::IT, OTF = BL::OTF, NCB = BL::NCB, ITB = BL::ITB;
    static_assert(BL::OT3 == OTF || BL::OT3 == 0, "every output tile takes the same chunks");
    constexpr int NC = NCB + (BL::OT3 > 0 ? 1 : 0);
    static_assert(OTF % 2 == 0, "pairs of output tiles");
    constexpr int NP = OTF / 2;                                    // pairs of output tiles = pipeline stages per chunk
    const int q = lane >> 4;
    typedef const __attribute__((address_space(3))) float* lds_f;
    typedef const __attribute__((address_space(3))) bf16x8* lds_b;
    unsigned o32 = (unsigned)(size_t)(lds_f)(frags + lane), o128 = (unsigned)(size_t)(lds_f)(frags + 4 * lane);
    asm volatile("" : "+v"(o32), "+v"(o128));
    const lds_f fl = (lds_f)(size_t)o32;
    const lds_b fq = (lds_b)(size_t)o128;
    float v4 = 0.f;
    if constexpr (BL::P4) {
        v4 = partial4_compact<IN, BL::KP>(frags + BL::p4, in, lane);
        if constexpr (BIAS) v4 += bias[16 * OTF + 4 * q];
    }
#pragma unroll
    for (int ot = 0; ot < OTF; ++ot) {
        if constexpr (BIAS) out[ot] = *reinterpret_cast<const f32x4*>(&bias[16 * ot + 4 * q]);
        else out[ot] = zero4();
    }
    auto frag_at = [&](int c, int ot, int pc) {
        return c < NCB ? fq[(BL::b16 / 4) + ((ot * NCB + c) * 3 + pc) * 64] : fq[(BL::b16c / 4) + (ot * 3 + pc) * 64];
    };
    auto unit_of = [&](int c, int p) { const int t = 2 * c + (p >> 1), r0 = 2 * (p & 1); return f32x2{in[t][r0], in[t][r0 + 1]}; };
    unsigned H[4], M[4], L[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) split3_unit(unit_of(0, p), H[p], M[p], L[p]);
    unsigned Hn_[4] = {0u, 0u, 0u, 0u}, Mn_[4] = {0u, 0u, 0u, 0u}, Ln_[4] = {0u, 0u, 0u, 0u};      // the next chunk's pieces, filled stage by stage
#ifndef RGL_BX_PIPE_WBUF
#define RGL_BX_PIPE_WBUF 1
#endif
    constexpr int WB = RGL_BX_PIPE_WBUF;                           // 2: the next stage's fragments are loaded under this stage's MFMAs (+24 VGPRs:
                                                                   // 11 spill in the children kernel; -1.8 % instead of -1.2 %, profiles/r06_f_*)
    bf16x8 w[WB][2][3];                                            // [buffer][tile of the pair][piece]
    if constexpr (WB == 2) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) w[0][g][pc] = frag_at(0, g, pc);
    }
    static_for<0, NC * NP>([&](auto stc) {
        constexpr int st = decltype(stc)::value, c = st / NP, pr = st % NP, buf = WB == 2 ? (st & 1) : 0;
        constexpr bool has_next = st + 1 < NC * NP;
        constexpr int cn = (st + 1) / NP, prn = (st + 1) % NP;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WB == 2 && has_next) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) w[WB == 2 ? buf ^ 1 : 0][g][pc] = frag_at(cn, 2 * prn + g, pc);
        }
        if constexpr (WB == 1) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) w[0][g][pc] = frag_at(c, 2 * pr + g, pc);
        }
        const bf16x8 sh = __builtin_bit_cast(bf16x8, u32x4{H[0], H[1], H[2], H[3]}), sm = __builtin_bit_cast(bf16x8, u32x4{M[0], M[1], M[2], M[3]}),
                     sl = __builtin_bit_cast(bf16x8, u32x4{L[0], L[1], L[2], L[3]});
#define RGL_BX_TERM(WP, AP)                                                                                         \
    _Pragma("unroll") for (int g = 0; g < 2; ++g)                                                                   \
        out[2 * pr + g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[buf][g][WP], AP, out[2 * pr + g], 0, 0, 0);
        RGL_BX_TERM(2, sh)          // small terms first
        RGL_BX_TERM(1, sm)
        RGL_BX_TERM(0, sl)
        RGL_BX_TERM(1, sh)
        RGL_BX_TERM(0, sm)
        RGL_BX_TERM(0, sh)
#undef RGL_BX_TERM
        // the next chunk's pieces, spread over this chunk's stages (four units over NP stages)
        if constexpr (c + 1 < NC) {
            constexpr int u0 = (4 * pr) / NP, u1 = (4 * (pr + 1)) / NP;
            static_for<u0, u1>([&](auto uc) {
                constexpr int p = decltype(uc)::value;
                split3_unit(unit_of(c + 1, p), Hn_[p], Mn_[p], Ln_[p]);
            });
        }
        // order: the next stage's six fragment reads early (their latency under this stage's MFMAs), then 1 MFMA : 2 VALU
        if constexpr (WB == 1) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (WB == 2) __builtin_amdgcn_sched_group_barrier(0x100, has_next ? 6 : 0, 0);
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if constexpr (pr == NP - 1 && c + 1 < NC) {
#pragma unroll
            for (int p = 0; p < 4; ++p) { H[p] = Hn_[p]; M[p] = Mn_[p]; L[p] = Ln_[p]; }
        }
    });
    __builtin_amdgcn_sched_barrier(0);
    // what is left on the f32 MFMA: input tiles beyond the chunks
#pragma unroll
    for (int it = ITB + (BL::OT3 > 0 ? 2 : 0); it < IT; ++it) {
        load_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (it == IT - 1 && r >= LastTileSteps<IN>::v) continue;
#pragma unroll
            for (int ot = 0; ot < OTF; ++ot) {
                const int k = (it - ITB - (ot < BL::OT3 ? 2 : 0)) * 4 + r;
                out[ot] = mfma4(fl[BL::f32_of(ot) + k * 64], in[it][r], out[ot]);
            }
        }
    }
    load_fence();
    if constexpr (BL::P4) out[OTF] = f32x4{v4, 0.f, 0.f, 0.f};
}

// The one-chunk layer (Bx1Layout): six terms per output tile, the partial tile on the f32 4 x 4 x 1 MFMA
template <int IN, int OUT, bool BIAS>
__device__ __forceinline__ void layer_mfma_bx1(const float* frags, const f32x4 (&in)[Tiles<IN>::v], f32x4 (&out)[Tiles<OUT>::v],
                                               int lane, const float* bias = nullptr) {
    using BL = Bx1Layout<IN, OUT>;
    constexpr int OTF = BL::OTF;
    const int q = lane >> 4;
    float v4 = 0.f;
    if constexpr (BL::P4) {
        v4 = partial4_compact<IN, BL::KP>(frags + BL::p4, in, lane);
        if constexpr (BIAS) v4 += bias[16 * OTF + 4 * q];
    }
#pragma unroll
    for (int ot = 0; ot < OTF; ++ot) {
        if constexpr (BIAS) out[ot] = *reinterpret_cast<const f32x4*>(&bias[16 * ot + 4 * q]);
        else out[ot] = zero4();
    }
    load_fence();
    const Split3 s = split3_pair(in[0], in[1]);
    constexpr int G = 2;
#pragma unroll
    for (int o0 = 0; o0 < OTF; o0 += G) {
        load_fence();
        bf16x8 w[G][3];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                if (o0 + g < OTF) w[g][pc] = *reinterpret_cast<const bf16x8*>(&frags[BL::b16 + (((o0 + g) * 3 + pc) * 64 + lane) * 4]);
#define RGL_BX_TERM(WP, AP)                                                                                         \
    _Pragma("unroll") for (int g = 0; g < G; ++g)                                                                   \
        if (o0 + g < OTF) out[o0 + g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g][WP], s.AP, out[o0 + g], 0, 0, 0);
        RGL_BX_TERM(2, h)
        RGL_BX_TERM(1, m)
        RGL_BX_TERM(0, l)
        RGL_BX_TERM(1, h)
        RGL_BX_TERM(0, m)
        RGL_BX_TERM(0, h)
#undef RGL_BX_TERM
    }
    load_fence();
    if constexpr (BL::P4) out[OTF] = f32x4{v4, 0.f, 0.f, 0.f};
}

// A whole layer in the six-term form (every input tile on the matrix pipe; full tiles only): the state predictor's scene kernel, whose
// 30 KB weight image leaves the LDS room the children kernel does not have.  Fragments: [ot][chunk][hi | mid | lo][lane] x 8 bf16 --
// chunk c covers the k slots of input tiles 2c, 2c + 1.
template <int IN, int OUT>
struct B6Floats { static constexpr int v = Tiles<OUT>::v * ((Tiles<IN>::v + 1) / 2) * 3 * 64 * 4; };

template <int IN, int OUT, bool BIAS>
__device__ __forceinline__ void layer_mfma_b6(const float* frags, const f32x4 (&in)[Tiles<IN>::v], f32x4 (&out)[Tiles<OUT>::v],
                                              int lane, const float* bias = nullptr) {
    constexpr int IT = Tiles<IN>::v, OT = Tiles<OUT>::v, NC = (IT + 1) / 2;
    static_assert(IN % 16 == 0, "full input tiles");
    const int q = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        if constexpr (BIAS) out[ot] = *reinterpret_cast<const f32x4*>(&bias[16 * ot + 4 * q]);
        else out[ot] = zero4();
    }
    // Blocks (chunk c, pair of output tiles) in order, their fragments double-buffered (round 6): block b + 1's six b128 reads are issued
    // in front of block b's twelve MFMAs, and the first block's in front of the first split -- the LDS latency sits under work
    // instead of in front of every block (ddddddw BBBBBBBBBBBB in the round-5 instruction stream).
    constexpr int G = 2, NOB = (OT + G - 1) / G, NB = NC * NOB;
    bf16x8 w[2][G][3];
    auto load_block = [&](int b, int buf) {
        const int c = b / NOB, o0 = G * (b % NOB);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                if (o0 + g < OT) w[buf][g][pc] = *reinterpret_cast<const bf16x8*>(&frags[((((o0 + g) * NC + c) * 3 + pc) * 64 + lane) * 4]);
    };
    load_fence();
    load_block(0, 0);
    Split3 s;
    static_for<0, NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value, c = b / NOB, o0 = G * (b % NOB), buf = b & 1;
        if constexpr (b % NOB == 0) s = split3_pair(in[2 * c], 2 * c + 1 < IT ? in[2 * c + 1 < IT ? 2 * c + 1 : 0] : zero4());
        load_fence();
        if constexpr (b + 1 < NB) load_block(b + 1, buf ^ 1);
#define RGL_B6_TERM(WP, AP)                                                                                         \
    _Pragma("unroll") for (int g = 0; g < G; ++g)                                                                   \
        if (o0 + g < OT) out[o0 + g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[buf][g][WP], s.AP, out[o0 + g], 0, 0, 0);
        RGL_B6_TERM(2, h)          // small terms first
        RGL_B6_TERM(1, m)
        RGL_B6_TERM(0, l)
        RGL_B6_TERM(1, h)
        RGL_B6_TERM(0, m)
        RGL_B6_TERM(0, h)
#undef RGL_B6_TERM
    });
    load_fence();
}

// The same with the ONE chunk of a 32-wide input already split (x0 of a tile feeds Wa and W1: one split, two products)
template <int OUT, bool BIAS>
__device__ __forceinline__ void layer_mfma_b6_pre(const float* frags, const Split3& s, f32x4 (&out)[Tiles<OUT>::v], int lane,
                                                  const float* bias = nullptr) {
    constexpr int OT = Tiles<OUT>::v;
    const int q = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        if constexpr (BIAS) out[ot] = *reinterpret_cast<const f32x4*>(&bias[16 * ot + 4 * q]);
        else out[ot] = zero4();
    }
    constexpr int G = 2;
#pragma unroll
    for (int o0 = 0; o0 < OT; o0 += G) {
        load_fence();
        bf16x8 w[G][3];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                if (o0 + g < OT) w[g][pc] = *reinterpret_cast<const bf16x8*>(&frags[(((o0 + g) * 3 + pc) * 64 + lane) * 4]);
#define RGL_B6_TERM(WP, AP)                                                                                         \
    _Pragma("unroll") for (int g = 0; g < G; ++g)                                                                   \
        if (o0 + g < OT) out[o0 + g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g][WP], s.AP, out[o0 + g], 0, 0, 0);
        RGL_B6_TERM(2, h)
        RGL_B6_TERM(1, m)
        RGL_B6_TERM(0, l)
        RGL_B6_TERM(1, h)
        RGL_B6_TERM(0, m)
        RGL_B6_TERM(0, h)
#undef RGL_B6_TERM
    }
    load_fence();
}

// float slot `idx` of the fragment image layer_mfma_b6 reads, for a k-major matrix W[in * ld + out] (columns >= n_out: 0): a pair of
// bf16 pieces.  Unit u = idx / 4 = ((ot NC + c) 3 + piece) 64 + lane; element e of it is W[in = 16 (2c + e / 4) + 4 q + e % 4][out of
// A-operand row lane % 16] -- hi = bf16(w), mid = bf16(w - hi), lo = w - hi - mid (exact).
template <int IN, int OUT>
__device__ __forceinline__ float frag_bf3_ld(const float* __restrict__ W, int ld, int n_out, int idx) {
    constexpr int IT = Tiles<IN>::v, NC = (IT + 1) / 2;
    const int u = idx >> 2, p = idx & 3;
    const int l = u & 63, rest = u >> 6;
    const int pc = rest % 3, c = (rest / 3) % NC, ot = rest / 3 / NC;
    const int m = l & 15, q = l >> 4;
    const int out = 16 * ot + m;
    bf16x2 v;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = 2 * p + k, t = 2 * c + (e >> 2);
        const int in = 16 * t + 4 * q + (e & 3);
        const float w = (t < IT && in < IN && out < n_out) ? W[in * ld + out] : 0.f;
        const __bf16 hi = (__bf16)w;
        const float r1 = w - (float)hi;
        const __bf16 mid = (__bf16)r1;
        const __bf16 lo = (__bf16)(r1 - (float)mid);
        v[k] = pc == 0 ? hi : (pc == 1 ? mid : lo);
    }
    return __builtin_bit_cast(float, v);
}

// the identity matrix in that layout (gaussian similarity: Wa = I): hi piece 1 on the diagonal
template <int D>
__device__ __forceinline__ float frag_bf3_identity(int idx) {
    constexpr int IT = Tiles<D>::v, NC = (IT + 1) / 2;
    const int u = idx >> 2, p = idx & 3;
    const int l = u & 63, rest = u >> 6;
    const int pc = rest % 3, c = (rest / 3) % NC, ot = rest / 3 / NC;
    bf16x2 v;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = 2 * p + k, in = 16 * (2 * c + (e >> 2)) + 4 * (l >> 4) + (e & 3), out = 16 * ot + (l & 15);
        v[k] = (__bf16)((pc == 0 && in == out) ? 1.f : 0.f);
    }
    return __builtin_bit_cast(float, v);
}

template <int OUT>
__device__ __forceinline__ void relu_tiles(f32x4 (&x)[Tiles<OUT>::v]) {
#pragma unroll
    for (int ot = 0; ot < Tiles<OUT>::v; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[ot][r] = relu1(x[ot][r]);
}

}  // namespace
