// The search's bookkeeping steps per tree node -- one-step values + top-w clipping (action_clip), the V_planning back-up and the
// root's first-maximum -- as device functions, shared by the stand-alone kernels of rgl_tree.hip (mprl_select_kernel,
// mprl_backup_kernel, mprl_root_kernel) and by the TAIL of children_fused_kernel (rgl_fused.hip), where the workgroup that scored
// all 81 children of a parent goes on to select / back up / decide for the parents (and, at the deepest level, the whole roots)
// it owns: no launch of their own, no second pass over HBM.
// Follows crowd_nav/policy/model_predictive_rl.py:228-231 (strict '>' argmax), :242-269 (action_clip), :271-302 (V_planning).
#pragma once
#include "rgl_common.h"
#include "rgl_mfma.h"

namespace {

constexpr int kMaxSparseWidth = 16;     // sparse (one action per group) searches: widest clipping the select step supports
constexpr int kRootLanes = 16;          // lanes that score the kept actions of one root side by side (tail_root)

struct TailLevel {
    const float* reward;      // [P][A]
    const float* child_value; // [P][A]   V(child)
    int* keep;                // [P][W]
    float* backup;            // [P][W]
    int* best_slot;           // [P]
    int P;
};

struct TailArgs {
    int enabled;              // 0: the kernel has no tail work (stand-alone value_children calls)
    int level, D, A, W, clip, sparse;
    float gamma_f;
    const int* groups;        // [A] or null
    const float* child_robot; // this level [P][A][9]
    float* value1;            // this level [P][A]
    const float* reward_sel;  // null, or [P][A]: the rewards THIS level's selection uses instead of lv[level].reward (joint-state
                              // roots: upstream's root action_clip reads the tensor state, the root values the JointState)
    float* next_robot;        // next level's robot rows [P*W][9]; null at the deepest level
    TailLevel lv[8];
    int chain;                // deepest level only: the launch also runs the back-up steps of the levels above and the root step
                              // (every workgroup owns the parents of whole roots)
    int B;
    int* best_action;         // [B]
    float* best_value;        // [B]
    float* root_values;       // [B][W] or null
    int* root_kept;           // [B][W] or null
};

__device__ __forceinline__ int tail_fallback(const int* kl, int k) { return k > 0 ? kl[k - 1] : 0; }

// One WAVE, parent p of level t.level: one-step values, top-w clipping (argpartition semantics; sparse: one action per group in
// descending value order), next level's robot states; at the deepest level also the leaf values V(kept child) and, below the
// root level, this parent's own back-up step into its parent's row.  Lane l owns actions l, l+64, l+128, l+192.
// `kl`: kTailLdsInts ints of LDS private to the wave (kept indices | V(child) | reward of the parent's actions).
// Written for LATENCY (in the fused kernel's tail nothing else hides it): every global load that does not depend on the selection
// -- rewards, child values, the parent's own value one level up -- is issued up front, and what the back-up needs of the kept
// actions comes from the LDS copy instead of a second trip to memory.
constexpr int kTailLdsInts = 3 * RGL_MAX_ACTIONS;

__device__ __forceinline__ void tail_select(const TailArgs& t, int p, int* kl) {
    const int lane = threadIdx.x & 63;
    const int A = t.A, W = t.W;
    const TailLevel& L = t.lv[t.level];
    const float* rw = (t.reward_sel ? t.reward_sel : L.reward) + (size_t)p * A;
    const float* cv = L.child_value + (size_t)p * A;
    float* v1 = t.value1 + (size_t)p * A;
    float* cvl = reinterpret_cast<float*>(kl) + RGL_MAX_ACTIONS;
    float* rwl = cvl + RGL_MAX_ACTIONS;
    const float gamma_f = t.gamma_f;
    const bool deepest = t.level + 1 == t.D, up = deepest && t.level >= 1;
    float val[4], cvr[4], rwr[4];
    bool avail[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = lane + 64 * k;
        avail[k] = a < A;
        cvr[k] = avail[k] ? cv[a] : 0.f;
        rwr[k] = avail[k] ? rw[a] : 0.f;
    }
    float v_up = 0.f;                                    // V of this parent as a child of ITS parent (back-up step below)
    if (up) {
        const TailLevel& U = t.lv[t.level - 1];
        const int q = p / W, slot = p - q * W;
        v_up = U.child_value[(size_t)q * A + U.keep[(size_t)q * W + slot]];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = lane + 64 * k;
        val[k] = 0.f;
        if (avail[k]) {
            val[k] = __fadd_rn(rwr[k], __fmul_rn(gamma_f, cvr[k]));
            v1[a] = val[k];
            if (deepest) { cvl[a] = cvr[k]; rwl[a] = rwr[k]; }
        }
    }
    int* kp = L.keep + (size_t)p * W;
    if (!t.clip) {
        for (int a = lane; a < A; a += 64) { kp[a] = a; kl[a] = a; }
    } else {
        // sparse search: one action per group, visited in descending one-step value; EXACTLY tied values are visited lower index
        // first.  (Upstream walks np.argsort(values)[::-1], model_predictive_rl.py:254: numpy's default sort is not stable and is
        // vectorised per CPU, so the order of exact ties is platform-defined there -- a stable sort would visit the HIGHER index
        // first.  Ties between different actions need bit-equal reward + gamma * V: documented deviation)
        // The groups taken so far are kept BY ID (any int32, as in the reference's python
        // set, model_predictive_rl.py:252-263) -- the width of a sparse search is at most kMaxSparseWidth (checked by the entry
        // point), so the set is a handful of wave-uniform registers and no id range has to be imposed on the caller.
        int seen[kMaxSparseWidth];
        int nkept = 0;
        int bi = -1;
        while (nkept < W) {
            // lane-local best: larger value first, then lower index; a NaN is only taken when nothing else is left
            bi = -1;
            float bv = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!avail[k]) continue;
                const float v = val[k];
                if (bi < 0 || v > bv || (bv != bv && v == v)) {
                    bi = lane + 64 * k;
                    bv = v;
                }
            }
            // wave argmax over (value, index) on the VALU: the maximum over the lanes that hold a number (DPP row steps + two
            // permlane swaps), then the lowest action index among the lanes that attain it by ballots, slot by slot (lane l's
            // candidates are l, l + 64, ...).  The ds_bpermute butterfly this replaces -- twelve trips through the LDS crossbar per
            // kept action, each waited for -- was most of the step's time (profiles/r06_z_tail_ab.md).
            {
                const bool have = bi >= 0, number = have && bv == bv;
                const bool numbers = __ballot(number) != 0ull;
                const float m = kgroups_max(row16_max(number ? bv : -INFINITY));
                const bool win = numbers ? (number && bv == m) : have;          // only NaNs left: the lowest index among them
                int wbi = -1;
#pragma unroll
                for (int k = 3; k >= 0; --k) {
                    const unsigned long long mk = __ballot(win && (bi >> 6) == k);
                    if (mk) wbi = __ffsll((long long)mk) - 1 + 64 * k;
                }
                bi = wbi;
            }
            if (bi < 0) break;                          // nothing left (wave-uniform)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (lane + 64 * k == bi) avail[k] = false;
            if (t.sparse) {
                const int gi = t.groups[bi];
                bool dup = false;
#pragma unroll
                for (int k = 0; k < kMaxSparseWidth; ++k) dup = dup || (k < nkept && seen[k] == gi);
                if (dup) continue;
#pragma unroll
                for (int k = 0; k < kMaxSparseWidth; ++k)
                    if (k == nkept) seen[k] = gi;
            }
            if (lane == 0) { kp[nkept] = bi; kl[nkept] = bi; }
            ++nkept;
        }
        if (lane == 0)
            for (int k = nkept; k < W; ++k) { kp[k] = tail_fallback(kl, k); kl[k] = kp[k]; }   // unreachable for validated inputs
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the wave's LDS writes (kl, cvl, rwl) are ordered before its reads below
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (t.next_robot) {
        for (int idx = lane; idx < W * 9; idx += 64) {      // same wave wrote kl: LDS operations of a wave execute in order
            const int k = idx / 9, i = idx - k * 9;
            const int a = kl[k];
            t.next_robot[((size_t)p * W + k) * 9 + i] = t.child_robot[((size_t)p * A + a) * 9 + i];
        }
    }
    if (deepest) {
        for (int k = lane; k < W; k += 64) L.backup[(size_t)p * W + k] = cvl[kl[k]];     // V_planning(child, 1) = V(child)
        if (up && lane == 0) {
            // ret_k = v/d + (d-1)/d * (gamma*nv_k + r_k), first maximum (model_predictive_rl.py:293,298-302); d = 2 here
            const TailLevel& U = t.lv[t.level - 1];
            const int d = 2;
            const int q = p / W, slot = p - q * W;
            const float v_over_d = __fdiv_rn(v_up, (float)d);
            const float c = (float)((double)(d - 1) / (double)d);
            float best = 0.f;
            int bk = -1;
            for (int k = 0; k < W; ++k) {
                const int a = kl[k];
                const float inner = __fadd_rn(__fmul_rn(gamma_f, cvl[a]), rwl[a]);
                const float ret = __fadd_rn(v_over_d, __fmul_rn(c, inner));
                if (bk < 0 || ret > best) {
                    best = ret;
                    bk = k;
                }
            }
            U.backup[(size_t)q * W + slot] = best;
            L.best_slot[p] = bk;
        }
    }
    __builtin_amdgcn_wave_barrier();                           // kl is reused by the wave's next parent
}

// Level l >= 1, one THREAD, parent p: ret_k = v/d + (d-1)/d * (gamma*nv_k + r_k); the max goes to the slot of p in its own
// parent's backup row (model_predictive_rl.py:293,298-302).  d = D - l + 1.
__device__ __forceinline__ void tail_backup(const TailArgs& t, int l, int p) {
    const TailLevel& L = t.lv[l];
    const TailLevel& U = t.lv[l - 1];
    const int A = t.A, W = t.W, d = t.D - l + 1;
    const int q = p / W, slot = p - q * W;
    const float v = U.child_value[(size_t)q * A + U.keep[(size_t)q * W + slot]];
    const float v_over_d = __fdiv_rn(v, (float)d);
    const float c = (float)((double)(d - 1) / (double)d);
    float best = 0.f;
    int bk = -1;
    for (int k = 0; k < W; ++k) {
        const float r = L.reward[(size_t)p * A + L.keep[(size_t)p * W + k]];
        const float inner = __fadd_rn(__fmul_rn(t.gamma_f, L.backup[(size_t)p * W + k]), r);
        const float ret = __fadd_rn(v_over_d, __fmul_rn(c, inner));
        if (bk < 0 || ret > best) {
            best = ret;
            bk = k;
        }
    }
    U.backup[(size_t)q * W + slot] = best;
    L.best_slot[p] = bk;
}

// kRootLanes consecutive lanes per root (all of them must call; `live` = the root exists): the W kept actions of a root are
// scored side by side (W = A = 81 without action clipping: a single thread walking them paid 81 dependent gather latencies),
// then a first-maximum reduction over the lanes.
__device__ __forceinline__ void tail_root(const TailArgs& t, int b, int sub, bool live) {
    const TailLevel& L = t.lv[0];
    const int A = t.A, W = t.W;
    float best = -INFINITY;
    int bk = -1;
    if (live) {
        for (int k = sub; k < W; k += kRootLanes) {
            const int a = L.keep[(size_t)b * W + k];
            const float val = __fadd_rn(L.reward[(size_t)b * A + a], __fmul_rn(t.gamma_f, L.backup[(size_t)b * W + k]));
            if (t.root_values) t.root_values[(size_t)b * W + k] = val;
            if (t.root_kept) t.root_kept[(size_t)b * W + k] = a;
            if (val > best) {                  // strict '>' keeps the first maximum (:228)
                best = val;
                bk = k;
            }
        }
    }
    // first maximum over the 16 lanes (one DPP row): a slot beats no slot, then the larger value, then the smaller slot index --
    // the row's maximum over the lanes that hold a slot, then the smallest slot among the lanes that attain it (slot indices are
    // exact in fp32), both as DPP row reductions instead of eight trips through the LDS crossbar.  A lane's `best` is a number
    // whenever it holds a slot (strict '>' above never takes a NaN).
    static_assert(kRootLanes == 16, "one DPP row per root");
    {
        const bool have = bk >= 0;
        const float m = row16_max(have ? best : -INFINITY);
        const float nk = row16_max(have && best == m ? -(float)bk : -INFINITY);
        bk = nk > -INFINITY ? (int)(-nk) : -1;
        best = m;
    }
    if (live && sub == 0) {
        t.best_action[b] = bk >= 0 ? L.keep[(size_t)b * W + bk] : -1;   // -1 <=> 'Value network is not well trained'
        t.best_value[b] = best;
        L.best_slot[b] = bk;
    }
}

}  // namespace
