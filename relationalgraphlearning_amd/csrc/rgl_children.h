// Next robot state + estimate_reward for one (parent, action) pair: the body of mprl_children_kernel, shared with the scene
// kernel, which runs it on extra workgroups beside the state predictor's graph forward (the two are independent).
// Follows crowd_nav/policy/state_predictor.py:41-60, model_predictive_rl.py:304-357, crowd_sim/envs/utils/utils.py:4-26.
#pragma once
#include "rgl_common.h"
#include "rgl_mfma.h"

namespace {

struct ChildrenArgs {
    const float* robot;        // [P][9]
    const float* humans;       // [P / humans_per][H][5]
    int humans_per;
    const double* actions;     // [A][2]
    int P, H, A, kinematics;
    double dt;
    int joint;                 // 1: position differences in float64 (JointState roots), 0: rounded to fp32 first
    float* child_robot;        // [P][A][9]
    float* reward;             // [P][A]
    const double* robot64;     // null, or the float64 states robot / humans were rounded from (joint roots): the reward reads these
    const double* humans64;
    float* reward_clip;        // null, or [P][A]: the same rewards read as a TENSOR-BORN state (joint = 0 on the fp32 rows) -- what
                               // upstream's root action_clip sees of a joint-state root (model_predictive_rl.py:216-218,246-248)
    int p_base, c_base;        // `robot` / `humans` start at parent p_base / crowd c_base (0 for whole-level arrays; a workgroup that staged
                               // its own parents' rows in LDS hands in its sub-range -- no pointer is ever rebased below its buffer:
                               // a flat LDS address that leaves the aperture faults, HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION)
    float v_max;               // > 0: an upper bound of the table's speeds (MprlPlanner::action_speed_bound, ABI 8); 0: every wave derives
                               // it from the table (two dependent float64 loads + a square root + a wave reduction: ~1.5 us of latency)
};

__device__ __forceinline__ double seg_point_dist_origin(double px, double py, double ex, double ey, bool f32_degenerate,
                                                        float fpx, float fpy) {
    // distance from the origin to the segment (px,py)-(ex,ey); utils.py:4-26 with (x3,y3) = 0
    const double sx = ex - px, sy = ey - py;
    if (sx == 0.0 && sy == 0.0) {
        if (f32_degenerate) return (double)sqrtf(__fadd_rn(__fmul_rn(fpx, fpx), __fmul_rn(fpy, fpy)));
        return sqrt(px * px + py * py);
    }
    double u = ((0.0 - px) * sx + (0.0 - py) * sy) / (sx * sx + sy * sy);
    u = u > 1.0 ? 1.0 : (u < 0.0 ? 0.0 : u);
    const double cx = px + u * sx, cy = py + u * sy;
    return sqrt(cx * cx + cy * cy);
}

// estimate_reward of one (parent, action) pair under one reading of the parent state (model_predictive_rl.py:304-357):
//   joint = 1: a JointState of python floats -- position differences in float64, of the float64 rows when `use64` (and the
//              planner carries them), of the fp32 rows widened otherwise;
//   joint = 0: a tensor-born state (tensor_to_joint_state, state.py:82-92) -- numpy float32 scalars, position differences
//              rounded to fp32 first, everything else widened by the float64 action / time step.
// `near_mask`: bit h clear = human h provably cannot influence any child of this parent (children_wave), skipped outright.
__device__ __forceinline__ double pair_reward(const ChildrenArgs& ca, int p, int a, unsigned long long near_mask, int joint,
                                              bool use64) {
    const float* __restrict__ r = ca.robot + (size_t)(p - ca.p_base) * 9;
    const float* __restrict__ hs = ca.humans + (size_t)(p / ca.humans_per - ca.c_base) * ca.H * 5;
    const double* r64 = use64 && ca.robot64 ? ca.robot64 + (size_t)p * 9 : nullptr;          // float64 roots (humans_per == 1 there)
    const double* hs64 = use64 && ca.humans64 ? ca.humans64 + (size_t)p * ca.H * 5 : nullptr;
    auto R = [&](int i) { return r64 ? r64[i] : (double)r[i]; };
    const double a0 = ca.actions[2 * a], a1 = ca.actions[2 * a + 1];
    const double dt = ca.dt;
    const int H = ca.H;
    double avx, avy, nx, ny;
    if (ca.kinematics == RGL_HOLONOMIC) {
        avx = a0;
        avy = a1;
        nx = R(0) + a0 * dt;
        ny = R(1) + a1 * dt;
    } else {
        // estimate_reward uses theta (slot 8) for the relative velocity and the goal test
        const double th = a1 + R(8);
        avx = a0 * cos(th);
        avy = a0 * sin(th);
        const double th2 = R(8) + a1;
        nx = R(0) + cos(th2) * a0 * dt;
        ny = R(1) + sin(th2) * a0 * dt;
    }
    bool collision = false;
    double dmin = INFINITY;
    const float favx = (float)avx, favy = (float)avy, fdt = (float)dt;
    // the near humans by index, then (crowds beyond 64, which carry no mask) the rest: one visit per human that can matter instead of
    // H mask tests (round 6: a crowd of 19 spread over the arena has 1-3 of them)
    auto visit = [&](int h) {
        const float* hu = hs + h * 5;
        {
            // fp32 pre-test of the exact shortcut below: with T = radii + 0.25, |p|^2 >= 2 (|e - p|^2 + T^2) proves that this
            // human's clearance is >= 0.25, i.e. that it cannot influence the reward.  Evaluated in fp32 with a 1e-3 relative
            // margin -- three orders above any fp32 rounding here (|p| >= 0.85 whenever the test passes, so the cancellation in
            // h - r is harmless) -- its YES implies the float64 YES; everything else takes the float64 path unchanged.
            const float qx = hu[0] - r[0], qy = hu[1] - r[1];
            const float sxf = (hu[2] - favx) * fdt, syf = (hu[3] - favy) * fdt;
            const float Tf = hu[4] + r[4] + 0.25f;
            if (qx * qx + qy * qy >= 2.002f * (sxf * sxf + syf * syf + Tf * Tf)) return;
        }
        const double* hu64 = hs64 ? hs64 + h * 5 : nullptr;
        auto HU = [&](int i) { return hu64 ? hu64[i] : (double)hu[i]; };
        double px, py;
        float fpx = 0.f, fpy = 0.f;
        if (joint) {
            px = HU(0) - R(0);
            py = HU(1) - R(1);
        } else {
            fpx = __fsub_rn(hu[0], r[0]);
            fpy = __fsub_rn(hu[1], r[1]);
            px = (double)fpx;
            py = (double)fpy;
        }
        const double vx = HU(2) - avx, vy = HU(3) - avy;
        const double ex = px + vx * dt, ey = py + vy * dt;
        // Exact shortcut: the outcome depends on this human only if its clearance d is < 0.2 (collision, or the minimum
        // when that is below the discomfort distance).  dist(origin, segment) >= |p| - |e - p|, so with
        // T = radii + 0.25 the test |p|^2 >= 2 (|e - p|^2 + T^2)  (=> |p| >= |e - p| + T) proves d >= 0.25 without the
        // float64 division and square root of the general case; the 0.05 margin dwarfs every rounding involved.
        {
            const double T = HU(4) + R(4) + 0.25;
            const double sx = ex - px, sy = ey - py;
            if (px * px + py * py >= 2.0 * (sx * sx + sy * sy + T * T)) return;
        }
        const double d = seg_point_dist_origin(px, py, ex, ey, !joint, fpx, fpy) - HU(4) - R(4);
        if (d < 0.0) collision = true;
        if (d < dmin) dmin = d;
    };
    unsigned long long todo = H >= 64 ? near_mask : (near_mask & ((1ull << H) - 1ull));
    while (todo) {
        const int h = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        visit(h);
    }
    for (int h = 64; h < H; ++h) visit(h);
    const double gx = nx - R(5), gy = ny - R(6);
    // norm < radius (model_predictive_rl.py:336-345): away from the boundary the squares decide -- a 1e-12 relative margin is four
    // orders above every rounding of x, r^2 and the root -- and the float64 square root runs only inside it
    const double gd2 = gx * gx + gy * gy, gr2 = R(4) * R(4);
    const bool reaching = (R(4) > 0.0 && gd2 <= gr2 * (1.0 - 1e-12)) ? true
                          : ((R(4) > 0.0 && gd2 >= gr2 * (1.0 + 1e-12)) ? false : sqrt(gd2) < R(4));
    if (collision) return -0.25;
    if (reaching) return 1.0;
    if (dmin < 0.2) return (dmin - 0.2) * 0.5 * dt;
    return 0.0;
}

// estimate_reward of a TENSOR-BORN state under the table's stop action.  Upstream builds that one entry from python ints --
// `ActionXY(0, 0)` / `ActionRot(0, 0)` (model_predictive_rl.py:166) -- and every other entry from numpy float64 products
// (:183-186).  The state's fields are numpy float32 scalars (state.py:82-92); under numpy >= 2 promotion (NEP 50: python scalars
// are weak) nothing in estimate_reward widens them when the action is made of python ints -- the whole function, the time step
// and the 0.2 / 0.5 constants included, runs in float32 -- whereas a float64 action component widens everything but the
// position differences (pair_reward, joint = 0).  Observed in the reference as it runs in this image (numpy 2.2.6): fixture
// root_clip.npz, action 0.  Each operation below is one float32 numpy scalar operation of utils.py:4-26 / :316-355, in order;
// with v = 0 both kinematics reduce to the same arithmetic (v * cos(..) = +-0).
__device__ __forceinline__ double stop_reward_f32(const ChildrenArgs& ca, int p, unsigned long long near_mask) {
    const float* __restrict__ r = ca.robot + (size_t)(p - ca.p_base) * 9;
    const float* __restrict__ hs = ca.humans + (size_t)(p / ca.humans_per - ca.c_base) * ca.H * 5;
    const float fdt = (float)ca.dt;
    bool collision = false;
    float dmin = INFINITY;
    auto visit = [&](int h) {
        const float* hu = hs + h * 5;
        const float px = __fsub_rn(hu[0], r[0]), py = __fsub_rn(hu[1], r[1]);
        const float ex = __fadd_rn(px, __fmul_rn(hu[2], fdt)), ey = __fadd_rn(py, __fmul_rn(hu[3], fdt));
        const float sx = __fsub_rn(ex, px), sy = __fsub_rn(ey, py);
        {
            // the same exact shortcut as pair_reward: provably >= 0.25 of clearance, the human cannot influence the reward
            const float Tf = hu[4] + r[4] + 0.25f;
            if (px * px + py * py >= 2.002f * (sx * sx + sy * sy + Tf * Tf)) return;
        }
        float dist;
        if (sx == 0.f && sy == 0.f) {
            dist = sqrtf(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)));
        } else {
            float u = __fdiv_rn(__fadd_rn(__fmul_rn(-px, sx), __fmul_rn(-py, sy)), __fadd_rn(__fmul_rn(sx, sx), __fmul_rn(sy, sy)));
            u = u > 1.f ? 1.f : (u < 0.f ? 0.f : u);
            const float cx = __fadd_rn(px, __fmul_rn(u, sx)), cy = __fadd_rn(py, __fmul_rn(u, sy));
            dist = sqrtf(__fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy)));
        }
        const float d = __fsub_rn(__fsub_rn(dist, hu[4]), r[4]);
        if (d < 0.f) collision = true;
        if (d < dmin) dmin = d;
    };
    unsigned long long todo = ca.H >= 64 ? near_mask : (near_mask & ((1ull << ca.H) - 1ull));
    while (todo) {
        const int h = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        visit(h);
    }
    for (int h = 64; h < ca.H; ++h) visit(h);
    const float gx = __fsub_rn(r[0], r[5]), gy = __fsub_rn(r[1], r[6]);
    const bool reaching = sqrtf(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy))) < r[4];
    if (collision) return -0.25;
    if (reaching) return 1.0;
    if (dmin < 0.2f) return (double)__fmul_rn(__fmul_rn(__fsub_rn(dmin, 0.2f), 0.5f), fdt);
    return 0.0;
}

// stop_reward_f32 by one WHOLE wave (H <= 64): lane h takes human h, the minimum and the collision flag meet in a wave reduction -- the
// same float32 operations per human, min / any over the same set, so the same bits.  Round 6: inside children_wave the one stop pair
// among a wave's 64 made EVERY lane wait through this second loop over the crowd (the two reward forms are divergent branches).
__device__ __forceinline__ double stop_reward_wave(const ChildrenArgs& ca, int p) {
    const int lane = threadIdx.x & 63;
    const float* __restrict__ r = ca.robot + (size_t)(p - ca.p_base) * 9;
    const float* __restrict__ hs = ca.humans + (size_t)(p / ca.humans_per - ca.c_base) * ca.H * 5;
    const float fdt = (float)ca.dt;
    float d = INFINITY;                                   // my human's clearance (inf: no such human, or provably >= 0.25)
    if (lane < ca.H) {
        const float* hu = hs + lane * 5;
        const float px = __fsub_rn(hu[0], r[0]), py = __fsub_rn(hu[1], r[1]);
        const float ex = __fadd_rn(px, __fmul_rn(hu[2], fdt)), ey = __fadd_rn(py, __fmul_rn(hu[3], fdt));
        const float sx = __fsub_rn(ex, px), sy = __fsub_rn(ey, py);
        const float Tf = hu[4] + r[4] + 0.25f;
        if (!(px * px + py * py >= 2.002f * (sx * sx + sy * sy + Tf * Tf))) {
            float dist;
            if (sx == 0.f && sy == 0.f) {
                dist = sqrtf(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)));
            } else {
                float u = __fdiv_rn(__fadd_rn(__fmul_rn(-px, sx), __fmul_rn(-py, sy)), __fadd_rn(__fmul_rn(sx, sx), __fmul_rn(sy, sy)));
                u = u > 1.f ? 1.f : (u < 0.f ? 0.f : u);
                const float cx = __fadd_rn(px, __fmul_rn(u, sx)), cy = __fadd_rn(py, __fmul_rn(u, sy));
                dist = sqrtf(__fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy)));
            }
            d = __fsub_rn(__fsub_rn(dist, hu[4]), r[4]);
        }
    }
    const bool collision = __ballot(d < 0.f) != 0ull;
    // wave minimum on the VALU (DPP row maxima + permlane swaps of the negated value: exact) instead of six ds_bpermute trips
    const float dmin = -kgroups_max(row16_max(-d));
    const float gx = __fsub_rn(r[0], r[5]), gy = __fsub_rn(r[1], r[6]);
    const bool reaching = sqrtf(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy))) < r[4];
    if (collision) return -0.25;
    if (reaching) return 1.0;
    if (dmin < 0.2f) return (double)__fmul_rn(__fmul_rn(__fsub_rn(dmin, 0.2f), 0.5f), fdt);
    return 0.0;
}

// A table row that is exactly (0, 0) is upstream's python-int stop action (no other entry of build_action_space can be: speeds
// are > 0); see stop_reward_f32.
__device__ __forceinline__ bool is_python_int_stop(const ChildrenArgs& ca, int a) {
    return ca.actions[2 * a] == 0.0 && ca.actions[2 * a + 1] == 0.0;
}

// One (parent, action) pair: next robot state (state_predictor.py:41-60) + estimate_reward -- under the reading the level asked
// for and, when `reward_clip` is set (joint-state roots of a clipped search), ALSO as the tensor-born state upstream's root
// action_clip is handed (model_predictive_rl.py:216-218 -> :246-248): the root's selection and the root's values price the same
// action with two different roundings there.
// `coop_stop`: the caller evaluates stop_reward_f32 itself (children_wave: by the whole wave), this call skips the outputs that need it
__device__ __forceinline__ void children_pa(const ChildrenArgs& ca, int p, int a, unsigned long long near_mask = ~0ull,
                                            bool coop_stop = false) {
    const long long idx = (long long)p * ca.A + a;
    const float* r = ca.robot + (size_t)(p - ca.p_base) * 9;
    const double a0 = ca.actions[2 * a], a1 = ca.actions[2 * a + 1];
    const double dt = ca.dt;
    float c[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = r[i];
    if (ca.kinematics == RGL_HOLONOMIC) {
        c[0] = __fadd_rn(r[0], (float)(a0 * dt));
        c[1] = __fadd_rn(r[1], (float)(a1 * dt));
        c[2] = (float)a0;
        c[3] = (float)a1;
    } else {
        // the reference rotates slot 7 (v_pref), not slot 8 (theta): kept (state_predictor.py:53-58)
        const float th7 = __fadd_rn(r[7], (float)a1);
        const float cs = cosf(th7), sn = sinf(th7);
        c[7] = th7;
        c[0] = __fadd_rn(r[0], (float)((double)cs * a0 * dt));
        c[1] = __fadd_rn(r[1], (float)((double)sn * a0 * dt));
        c[2] = (float)((double)cs * a0);
        c[3] = (float)((double)sn * a0);
    }
    // (the nine strided dword stores per lane are not what the launch waits for: without them the level-0 embedding + reward launch
    // of a 2048-root step takes 15.3 instead of 16.3 us -- profiles/r06_d_timeline_child_store_ab.md)
    float* co = ca.child_robot + (size_t)idx * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) co[i] = c[i];
    const bool stop = is_python_int_stop(ca, a);
    if (!(coop_stop && stop && !ca.joint))
        ca.reward[idx] = (float)(!ca.joint && stop ? stop_reward_f32(ca, p, near_mask) : pair_reward(ca, p, a, near_mask, ca.joint, true));
    if (ca.reward_clip && !(coop_stop && stop))
        ca.reward_clip[idx] = (float)(stop ? stop_reward_f32(ca, p, near_mask) : pair_reward(ca, p, a, near_mask, 0, false));
}

__device__ __forceinline__ void children_thread(const ChildrenArgs& ca, long long idx) {
    const int p = (int)(idx / ca.A);
    children_pa(ca, p, (int)(idx - (long long)p * ca.A));
}

// 64 consecutive pairs idx0 .. idx0 + 63 by one WHOLE wave (every lane must call it: lanes past `total` only help with the masks).
// With A >= 64 actions the pairs belong to at most two parents, pA and pA + 1.  Most humans of a crowd spread over an arena are far
// from the robot for EVERY action: with v_max >= |v_a| over the table, the relative displacement of human h over the step obeys
// |s| <= (|v_h| + v_max) dt, so |p|^2 >= 2 ((|v_h| + v_max)^2 dt^2 + T^2) (T = radii + 0.25) proves clearance >= 0.25 for all
// children of the parent -- the per-child pre-test of children_pa would pass for each of them.  Lanes 0..31 evaluate that test
// for the humans of pA, lanes 32..63 for pA + 1 (fp32, 1 % margin; the per-child test it implies keeps 0.1 %), one ballot makes the
// two masks, and every pair skips its parent's far humans outright instead of pre-testing them one by one: same results,
// H pre-tests per parent instead of per child.  Crowds of 33..64 humans (round 4: BASELINE configs[4], 49 humans, spent 6 us per
// level on per-child pre-tests) take one ballot per parent, every lane a human.  (Float64 roots and larger crowds: children_thread.)
__device__ __forceinline__ void children_wave(const ChildrenArgs& ca, long long idx0, long long total, float v_max) {
    const int lane = threadIdx.x & 63;
    const int A = ca.A, H = ca.H;
    const int pA = (int)(idx0 / A);
    const int p_end = (int)((total + A - 1) / A);        // parents with pairs below `total` (a caller may own a sub-range of the level and
                                                         // hand in row pointers that are valid for its own parents only)
    auto near_of = [&](int pm, int h) {
        if (h >= H || pm >= p_end || pm >= ca.P) return false;
        const float* r = ca.robot + (size_t)(pm - ca.p_base) * 9;
        const float* hu = ca.humans + ((size_t)(pm / ca.humans_per - ca.c_base) * H + h) * 5;
        const float qx = hu[0] - r[0], qy = hu[1] - r[1];
        const float sm = (sqrtf(hu[2] * hu[2] + hu[3] * hu[3]) + v_max) * (float)ca.dt;
        const float Tf = hu[4] + r[4] + 0.25f;
        return !(qx * qx + qy * qy >= 2.02f * (sm * sm + Tf * Tf));
    };
    unsigned long long mask_a, mask_b;
    if (H <= 32) {
        const unsigned long long bal = __ballot(near_of(pA + (lane >> 5), lane & 31));      // lanes 0..31: pA, lanes 32..63: pA + 1
        mask_a = bal & 0xffffffffull;
        mask_b = bal >> 32;
    } else {
        mask_a = __ballot(near_of(pA, lane));
        mask_b = __ballot(near_of(pA + 1, lane));
    }
    const long long idx = idx0 + lane;
    bool my_stop = false;
    if (idx < total) {
        const int p = (int)(idx / A), a = (int)(idx - (long long)p * A);
        children_pa(ca, p, a, p == pA ? mask_a : mask_b, true);
        my_stop = is_python_int_stop(ca, a) && (!ca.joint || ca.reward_clip != nullptr);
    }
    // the stop pairs among my 64 (at most two: one per parent), each by the whole wave over the humans
    unsigned long long stops = __ballot(my_stop);
    while (stops) {
        const int l = __ffsll((long long)stops) - 1;
        stops &= stops - 1ull;
        const long long sidx = idx0 + l;
        const float rv = (float)stop_reward_wave(ca, (int)(sidx / A));
        if (lane == l) {
            if (!ca.joint) ca.reward[sidx] = rv;
            if (ca.reward_clip) ca.reward_clip[sidx] = rv;
        }
    }
}

// largest action speed of the table (holonomic: |(vx, vy)|, unicycle: |v|), by one whole wave; slightly rounded up
__device__ __forceinline__ float table_speed_bound(const ChildrenArgs& ca) {
    if (ca.v_max > 0.f) return ca.v_max;
    const int lane = threadIdx.x & 63;
    float m = 0.f;
    for (int k = lane; k < ca.A; k += 64) {
        const double a0 = ca.actions[2 * k], a1 = ca.actions[2 * k + 1];
        m = fmaxf(m, ca.kinematics == RGL_HOLONOMIC ? (float)sqrt(a0 * a0 + a1 * a1) : (float)fabs(a0));
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    return m * 1.0001f;
}

}  // namespace
