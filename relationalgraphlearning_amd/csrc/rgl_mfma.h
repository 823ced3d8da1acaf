// Device + host helpers shared by the MFMA translation units (rgl_fast.hip, rgl_deep.hip).  gfx950 only.
#pragma once
#include "rgl_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int XD = 32;    // x_dim
constexpr int HID = 64;   // embedding hidden width
constexpr int XLD = 36;   // LDS row stride of 32-wide feature rows (16-byte aligned, bank-skewed)
constexpr int kThreads = 256;
constexpr int kWaves = 4;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
// Compiler-only fence: stops hipcc from hoisting every operand-fragment load of a long unrolled MFMA
// chain to the top (which costs hundreds of VGPRs); each k-group loads its fragments right before use.
__device__ __forceinline__ void load_fence() { asm volatile("" ::: "memory"); }

// Butterfly reductions over the four 16-lane k-groups of a wave (lanes l, l^16, l^32, l^48), on the VALU
// (gfx950 v_permlane16_swap / v_permlane32_swap) instead of the LDS crossbar (ds_bpermute).
__device__ __forceinline__ float kgroups_max(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float kgroups_sum(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v_mfma_f32_4x4x1_16B_f32: sixteen independent 4 x 4 x 1 blocks, block = lane / 4, A row = B column = lane % 4, D register = row
// (tools/micro/mfma_4x4.hip: layout checked on the hardware, 12 clocks per instruction against 32 for the 16 x 16 x 4 form).
// Used where a 16-row MFMA tile would carry at most four valid rows (rgl_mlp_chain.h Partial4, rgl_fused.hip T1P, rgl_deep.hip T4).
__device__ __forceinline__ f32x4 mfma4x4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
// lane (n, q) <- sum over the four k-groups of register q (see above)
__device__ __forceinline__ float kgroups_reduce_scatter(const f32x4& p) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p[0]), __float_as_uint(p[1]), false, false);
    const float s01 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p[2]), __float_as_uint(p[3]), false, false);
    const float s23 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// lane n of every 16-lane row <- lane n & 3 of that row (three DPP row shifts, each writing one bank of four lanes)
__device__ __forceinline__ float quad0_bcast(float x) {
    int v = __float_as_int(x);
    v = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0x2, false);      // row_shr:4  -> lanes 4..7
    v = __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0x4, false);      // row_shr:8  -> lanes 8..11
    v = __builtin_amdgcn_update_dpp(v, v, 0x11C, 0xf, 0x8, false);      // row_shr:12 -> lanes 12..15
    return __int_as_float(v);
}

// max over the 16 lanes of a DPP row (every lane gets it); the DPP operand rides in the v_max itself
// (s_nop 1: a DPP read of a VGPR needs two wait states after the VALU write of it, and the compiler's hazard recognizer does
// not look inside inline assembly)
__device__ __forceinline__ float row16_max(float x) {
    float y;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(y));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(y));
    return x;
}
// m / N for 0 <= m < 65536 and 1 <= N <= 64 with magic = floor(2^32 / N) + 1
__device__ __forceinline__ int div_small(int m, unsigned magic) { return (int)__umulhi((unsigned)m, magic); }
// relu as ONE full-rate instruction: on the bit patterns, max(int(x), 0) maps every negative float (and -0) to +0
// and leaves positive floats untouched (v_max_i32).  fmaxf() / med3 cost a canonicalising v_max extra on MFMA results.
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// Optional phase timing (-DRGL_PHASE_TIMING, tools/phase_timing.py): per-wave s_memtime deltas summed per phase.
#ifdef RGL_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[16];
#define PHASE_START()                                                      \
    unsigned long long phase_acc__[8] = {0, 0, 0, 0, 0, 0, 0, 0};          \
    const unsigned long long phase_rt0__ = __builtin_amdgcn_s_memrealtime(); \
    const unsigned long long phase_ct0__ = __builtin_amdgcn_s_memtime();   \
    unsigned long long phase_t0__ = __builtin_amdgcn_s_memtime()
#define PHASE_MARK(idx)                                                    \
    do {                                                                   \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();     \
        phase_acc__[idx] += now__ - phase_t0__;                            \
        phase_t0__ = now__;                                                \
    } while (0)
#define PHASE_FLUSH()                                                      \
    do {                                                                   \
        if ((threadIdx.x & 63) == 0)                                       \
            for (int i__ = 0; i__ < 8; ++i__) atomicAdd(&g_phase_cycles[i__], phase_acc__[i__]); \
        if (threadIdx.x == 0 && blockIdx.x == 0) { /* shader clock: s_memtime ticks per 100 MHz s_memrealtime tick */ \
            atomicAdd(&g_phase_cycles[8], __builtin_amdgcn_s_memtime() - phase_ct0__);           \
            atomicAdd(&g_phase_cycles[9], __builtin_amdgcn_s_memrealtime() - phase_rt0__);       \
        }                                                                  \
    } while (0)
#else
#define PHASE_MARK(idx) do { } while (0)
#define PHASE_START() do { } while (0)
#define PHASE_FLUSH() do { } while (0)
#endif

constexpr int WLD = 36;    // LDS row stride of the 32-column weight images (k-major); 4*WLD % 32 == 16 keeps the
                           // four 16-lane k-groups of an MFMA A-operand read on disjoint banks
constexpr int W1LD = 80;   // same for the 64-column image of wr1 (rows differ by 1 between k-groups)


// gfx90a+ DPP row_newbcast:K -- every lane of a 16-lane DPP row reads lane K of its row -- fused into the consuming VOP2
// instruction: a per-row scalar held once per DPP row reaches all lanes with no LDS traffic and no extra instruction.
// (hipcc does not fold row_newbcast movs into their users, hence the asm.)
template <int K>
__device__ __forceinline__ float dpp_rowbcast_mul(float row_src, float other) {
    float d;
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(row_src), "v"(other), "n"(K));
    return d;
}
template <int K>
__device__ __forceinline__ float dpp_rowbcast_add(float row_src, float other) {
    float d;
    asm("v_add_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(row_src), "v"(other), "n"(K));
    return d;
}
template <int K>
__device__ __forceinline__ float dpp_rowbcast_fmac(float row_src, float other, float acc) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(row_src), "v"(other), "n"(K));
    return acc;
}
template <int I, int END, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, END>(f);
    }
}


// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
inline bool mlp_is(const RglMlp& m, int d0, int d1, int d2, bool last_relu) {
    return m.n_layers == 2 && m.dims[0] == d0 && m.dims[1] == d1 && m.dims[2] == d2 && (m.last_relu != 0) == last_relu;
}

// The MFMA kernels compute S = X Wa X^T and normalise its rows.  `gaussian`, `squared` (S = X X^T, graph_model.py:67-69,
// 86-89) are the same with Wa = I, which the kernels build in their LDS weight image when the pointer is null (x*1 + 0 is
// exact: same bits as X X^T); `equal_attention` / `diagonal` (:90-93) ignore S.  Row normalisations the shared-crowd
// kernels and the scene kernel implement: A_ij = w_ij / sum_j w_ij with
//   SIM_SOFTMAX  w = e^{S_ij - max_j S_ij}      SIM_SQUARED  w = S_ij^2      SIM_EQUAL  w = 1      SIM_DIAGONAL  w = [i == j]
// (`cosine*` scale every column by a child-dependent norm and `concatenation` is a pair MLP: general kernel.)
enum { SIM_SOFTMAX = 0, SIM_SQUARED = 1, SIM_EQUAL = 2, SIM_DIAGONAL = 3,
       SIM_COSINE = 4, SIM_COSINE_SOFTMAX = 5, SIM_CONCAT = 6 };      // the scene kernel only (scene_similarity_mode)
inline int fast_similarity_mode(const RglGraph& g) {
    switch (g.similarity) {
        case RGL_SIM_EMBEDDED_GAUSSIAN:
        case RGL_SIM_GAUSSIAN: return SIM_SOFTMAX;
        case RGL_SIM_SQUARED: return SIM_SQUARED;
        case RGL_SIM_EQUAL_ATTENTION: return SIM_EQUAL;
        case RGL_SIM_DIAGONAL: return SIM_DIAGONAL;
        default: return -1;
    }
}
// the one-wave-per-scene kernel (rgl_scene.hip) also normalises by the row norms of S: the cosine family
inline int scene_similarity_mode(const RglGraph& g) {
    if (g.similarity == RGL_SIM_COSINE) return SIM_COSINE;
    if (g.similarity == RGL_SIM_COSINE_SOFTMAX) return SIM_COSINE_SOFTMAX;
    if (g.similarity == RGL_SIM_CONCATENATION)      // pair MLP 2X -> 64 -> 1 with ReLU after both layers (graph_model.py:46-47)
        return (g.w_a_mlp.n_layers == 2 && g.w_a_mlp.dims[0] == 2 * XD && g.w_a_mlp.dims[1] == HID && g.w_a_mlp.dims[2] == 1 &&
                g.w_a_mlp.last_relu) ? (int)SIM_CONCAT : -1;
    return fast_similarity_mode(g);
}
inline const float* bilinear_wa(const RglGraph& g) { return g.similarity == RGL_SIM_EMBEDDED_GAUSSIAN ? g.w_a : nullptr; }
// un-normalised weight of a VALID entry under the non-softmax modes (entry s of row i, column j)
__device__ __forceinline__ float plain_weight(int sim, float s, int i, int j) {
    return sim == SIM_SQUARED ? s * s : (sim == SIM_EQUAL ? 1.f : (i == j ? 1.f : 0.f));
}

// 0 / 1: the two shipped value heads (register-resident MFMA chains, robot_head_kernel<D1,D2,D3>); 2: any other head
// x_dim -> ... -> 1 within the ABI limits (robot_head_any_kernel: MFMA with run-time tile loops, activations in LDS)
inline int head_variant(const RglMlp& h) {
    if (h.n_layers < 1 || h.n_layers > RGL_MAX_MLP_LAYERS || h.last_relu || h.dims[0] != XD || h.dims[h.n_layers] != 1) return -1;
    if (h.n_layers == 4 && h.dims[1] == 32 && h.dims[2] == 100 && h.dims[3] == 100) return 0;     // ValueEstimator default
    if (h.n_layers == 4 && h.dims[1] == 150 && h.dims[2] == 100 && h.dims[3] == 100) return 1;    // gcn.ValueNetwork default
    for (int l = 1; l < h.n_layers; ++l)
        if (h.dims[l] < 1 || h.dims[l] > RGL_MAX_WIDTH) return -1;
    return 2;
}

inline bool fast_path_enabled() {
    static const bool off = [] { const char* e = getenv("RGL_FORCE_GENERIC"); return e && e[0] == '1'; }();
    return !off;
}

// RGL_CHILDREN_TILE_KERNEL=1 disables the shared-crowd kernels (rank-1, deep) in favour of the tile kernel (tests)
inline bool rank1_enabled() {
    static const bool off = [] { const char* e = getenv("RGL_CHILDREN_TILE_KERNEL"); return e && e[0] == '1'; }();
    return !off;
}

// ReLU in the clamp bit of a packed FMA.  VOP3P has no packed f32 max, but every packed op can clamp its result to [0, 1] (DX10
// clamp mode: NaN -> 0): on operands scaled by a power of two 2^-k the clamp IS the ReLU -- clamp01(2^-k x) = 2^-k relu(x) for
// x < 2^k, and power-of-two scalings commute with the rounding of every product and sum (no overflow, and what underflows is
// below 2^(k-149) in unscaled terms) -- so the scaled pass computes exactly 2^-k times the unscaled values and is scaled back
// where the next exact operation happens anyway.
// RULE for the inline-assembly forms: no operand may be the result register of an MFMA.  hipcc's hazard recognizer pads the wait
// states between an MFMA's write and a VALU read of it only for instructions it can see; an operand of an asm statement it
// cannot (measured: children_deep_kernel<NT = 1, f16> read its E O accumulators three instructions after the MFMA and was 2e-2
// off).  Results of compiler-visible VALU instructions, LDS loads (their s_waitcnt is inserted for asm operands too) and other asm
// outputs are fine.
// clamp01(s.lo * u + y) on both halves, and s.hi * t + acc on both halves: a pair s = (r, b) is read in place by op_sel
__device__ __forceinline__ f32x2 pk_fma_lo_clamp(f32x2 s, f32x2 u, f32x2 y) {
    f32x2 t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] clamp" : "=v"(t) : "v"(s), "v"(u), "v"(y));
    return t;
}
__device__ __forceinline__ f32x2 pk_fma_hi(f32x2 s, f32x2 t, f32x2 acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(s), "v"(t));
    return acc;
}
// clamp01(a * b + c), element-wise
__device__ __forceinline__ f32x2 pk_fma_clamp(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t) : "v"(a), "v"(b), "v"(c));
    return t;
}
__device__ __forceinline__ float fma_clamp(float a, float b, float c) { return __builtin_amdgcn_fmed3f(fmaf(a, b, c), 0.f, 1.f); }

}  // namespace
