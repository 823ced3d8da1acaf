// rgl_tree.hip -- the model-predictive rollout on device (path M) and the one-step search of
// path G.  Level-synchronous: every tree level is a handful of launches over ALL parents of
// that level, no host round trip, no allocation (caller-provided workspace) -> capturable in a
// hipGraph.
//
// Follows (reference paths): crowd_nav/policy/model_predictive_rl.py:192-357 (predict,
// action_clip, V_planning, estimate_reward), state_predictor.py:41-60,109-118,
// crowd_sim/envs/utils/utils.py:4-26, multi_human_rl.py:36-96, cadrl.py:113-138,241-276.
#include "rgl_children.h"
#include "rgl_tail.h"

#include <cstdlib>

namespace {

constexpr int kBlock = 256;

// One thread per (parent, action): next robot state + estimate_reward (rgl_children.h); a wave's 64 pairs share the far-human
// masks of their (at most two) parents when the table and the crowd allow it (children_wave).
__global__ void mprl_children_kernel(const ChildrenArgs ca) {
    const long long total = (long long)ca.P * ca.A;
    const long long idx0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) & ~63LL;      // my wave's first pair
    if (ca.A >= 64 && ca.H <= 64 && !ca.robot64) {
        if (idx0 < total) children_wave(ca, idx0, total, table_speed_bound(ca));
    } else {
        const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (idx < total) children_thread(ca, idx);
    }
}

__global__ void linear_humans_kernel(const float* __restrict__ humans, float* __restrict__ out, long long n_rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const float* h = humans + i * 5;
    float* o = out + i * 5;
    o[0] = __fadd_rn(h[0], h[2]);       // no time-step factor (state_predictor.py:115-116)
    o[1] = __fadd_rn(h[1], h[3]);
    o[2] = h[2];
    o[3] = h[3];
    o[4] = h[4];
}

__global__ void gather_parent_humans_kernel(const float* __restrict__ humans, int humans_per, int H, long long P,
                                            float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * H * 5) return;
    const long long p = i / (H * 5), rest = i - p * (H * 5);
    out[i] = humans[(p / humans_per) * H * 5 + rest];
}

// One WAVE per parent (tail_select, rgl_tail.h): one-step values, top-w clipping, next level's robot states; at the deepest
// level also the leaf values and the parent's own back-up step.
__global__ __launch_bounds__(256) void mprl_select_kernel(const TailArgs t) {
    __shared__ int kept_lds[4][kTailLdsInts];            // per wave: kept indices | V(child) | rewards (tail_select)
    const int p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= t.lv[t.level].P) return;
    tail_select(t, p, kept_lds[threadIdx.x >> 6]);
}

// value1 = reward + gamma_bar * V(child), each op rounded to fp32 like the reference's tensor arithmetic
__global__ void one_step_value_kernel(const float* __restrict__ r, const float* __restrict__ v, float g, long long n,
                                      float* __restrict__ o) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = __fadd_rn(r[i], __fmul_rn(g, v[i]));
}

// Level l >= 1, one thread per parent (tail_backup); 16 lanes per root (tail_root).
__global__ void mprl_backup_kernel(const TailArgs t, int l) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < t.lv[l].P) tail_backup(t, l, p);
}

__global__ void mprl_root_kernel(const TailArgs t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = i / kRootLanes;
    tail_root(t, b, i % kRootLanes, b < t.B);
}

// ------------------------------------------------------------------------------------------------
// path G
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rotate_row(const float* s, int unicycle, float* o) {
    // cadrl.py:241-276; every product/sum individually rounded like the chain of torch ops
    const float dx = __fsub_rn(s[5], s[0]), dy = __fsub_rn(s[6], s[1]);
    const float rot = atan2f(dy, dx);
    const float c = cosf(rot), sn = sinf(rot);
    o[0] = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    o[1] = s[7];
    o[2] = unicycle ? __fsub_rn(s[8], rot) : 0.f;
    o[3] = s[4];
    o[4] = __fadd_rn(__fmul_rn(s[2], c), __fmul_rn(s[3], sn));
    o[5] = __fsub_rn(__fmul_rn(s[3], c), __fmul_rn(s[2], sn));
    const float rx = __fsub_rn(s[9], s[0]), ry = __fsub_rn(s[10], s[1]);
    o[6] = __fadd_rn(__fmul_rn(rx, c), __fmul_rn(ry, sn));
    o[7] = __fsub_rn(__fmul_rn(ry, c), __fmul_rn(rx, sn));
    o[8] = __fadd_rn(__fmul_rn(s[11], c), __fmul_rn(s[12], sn));
    o[9] = __fsub_rn(__fmul_rn(s[12], c), __fmul_rn(s[11], sn));
    o[10] = s[13];
    const float ax = __fsub_rn(s[0], s[9]), ay = __fsub_rn(s[1], s[10]);
    o[11] = sqrtf(__fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay)));
    o[12] = __fadd_rn(s[4], s[13]);
}

__global__ void gcn_rotate_kernel(const float* __restrict__ in14, float* __restrict__ out13, int R, int unicycle) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    float s[14], o[13];
#pragma unroll
    for (int k = 0; k < 14; ++k) s[k] = in14[(size_t)i * 14 + k];
    rotate_row(s, unicycle, o);
#pragma unroll
    for (int k = 0; k < 13; ++k) out13[(size_t)i * 13 + k] = o[k];
}

// One thread per (scene, action, human): propagate, rotate; thread h == 0 also does compute_reward.
__global__ void gcn_prepare_kernel(const float* __restrict__ robot, const float* __restrict__ humans,
                                   const double* __restrict__ robot64, const double* __restrict__ humans64,
                                   const double* __restrict__ actions, int B, int H, int A, int kinematics, double dt,
                                   float* __restrict__ self6, float* __restrict__ hum7, float* __restrict__ reward) {
    // blockDim.x is a multiple of H: the H threads of a (root, action) pair sit in one workgroup, each contributes ITS human's
    // end-point clearance (one float64 sqrt per thread) and thread h == 0 takes the minimum -- it used to walk all H itself
    extern __shared__ double clearance[];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < (long long)B * A * H;
    const long long cidx = live ? idx : 0;
    const int h = (int)(cidx % H);
    const long long sa = cidx / H;
    const int a = (int)(sa % A), b = (int)(sa / A);
    const float* r = robot + (size_t)b * 9;
    const double* r64 = robot64 ? robot64 + (size_t)b * 9 : nullptr;        // the float64 state the fp32 row was rounded from
    auto R = [&](int i) { return r64 ? r64[i] : (double)r[i]; };
    const double a0 = actions[2 * a], a1 = actions[2 * a + 1];
    // CADRL.propagate in float64, as python floats
    double nr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) nr[i] = R(i);
    if (kinematics == RGL_HOLONOMIC) {
        nr[0] = R(0) + a0 * dt;
        nr[1] = R(1) + a1 * dt;
        nr[2] = a0;
        nr[3] = a1;
    } else {
        const double th = R(8) + a1;
        nr[2] = a0 * cos(th);
        nr[3] = a0 * sin(th);
        nr[0] = R(0) + nr[2] * dt;
        nr[1] = R(1) + nr[3] * dt;
        nr[8] = th;
    }
    const float* hu = humans + ((size_t)b * H + h) * 5;
    const double* hb64 = humans64 ? humans64 + (size_t)b * H * 5 : nullptr;
    auto HB = [&](int j, int i) { return hb64 ? hb64[j * 5 + i] : (double)humans[((size_t)b * H + j) * 5 + i]; };
    float s[14], o[13];
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = (float)nr[i];
    s[9] = (float)(HB(h, 0) + HB(h, 2) * dt);
    s[10] = (float)(HB(h, 1) + HB(h, 3) * dt);
    s[11] = hb64 ? (float)HB(h, 2) : hu[2];
    s[12] = hb64 ? (float)HB(h, 3) : hu[3];
    s[13] = hb64 ? (float)HB(h, 4) : hu[4];
    rotate_row(s, kinematics == RGL_UNICYCLE, o);
    if (live) {
        float* h7 = hum7 + ((size_t)sa * H + h) * 7;
#pragma unroll
        for (int i = 0; i < 7; ++i) h7[i] = o[6 + i];
    }
    {   // compute_reward (multi_human_rl.py:73-96): END-point distance of my human, float64
        const double hx = HB(h, 0) + HB(h, 2) * dt;
        const double hy = HB(h, 1) + HB(h, 3) * dt;
        const double ddx = nr[0] - hx, ddy = nr[1] - hy;
        clearance[threadIdx.x] = sqrt(ddx * ddx + ddy * ddy) - nr[4] - HB(h, 4);
    }
    __syncthreads();
    if (live && h == 0) {
        float* s6 = self6 + (size_t)sa * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) s6[i] = o[i];
        bool collision = false;
        double dmin = INFINITY;
        for (int j = 0; j < H; ++j) {
            const double d = clearance[threadIdx.x + j];
            if (d < 0.0) collision = true;
            if (d < dmin) dmin = d;
        }
        const double gx = nr[0] - nr[5], gy = nr[1] - nr[6];
        const bool reaching = sqrt(gx * gx + gy * gy) < nr[4];
        double rew;
        if (collision) rew = -0.25;
        else if (reaching) rew = 1.0;
        else if (dmin < 0.2) rew = (dmin - 0.2) * 0.5 * dt;
        else rew = 0.0;
        reward[sa] = (float)rew;
    }
}

// 16 lanes per root (kRootLanes): the A action values side by side, then a first-maximum reduction -- one thread walking 81 actions
// cost 25 us per call
__global__ void gcn_argmax_kernel(const float* __restrict__ robot, const float* __restrict__ reward,
                                  const float* __restrict__ value, int B, int A, double gamma, double dt,
                                  float* __restrict__ action_values, int* __restrict__ best_action, float* __restrict__ best_value) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t / kRootLanes, sub = t % kRootLanes;
    const bool live = b < B;
    double best = -INFINITY;
    int ba = -1;
    if (live) {
        const double disc = pow(gamma, dt * (double)robot[(size_t)b * 9 + 7]);
        for (int a = sub; a < A; a += kRootLanes) {
            const double v = (double)reward[(size_t)b * A + a] + disc * (double)value[(size_t)b * A + a];
            action_values[(size_t)b * A + a] = (float)v;
            if (v > best) {
                best = v;
                ba = a;
            }
        }
    }
#pragma unroll
    for (int m = kRootLanes / 2; m >= 1; m >>= 1) {
        const double ov = __shfl_xor(best, m);
        const int oa = __shfl_xor(ba, m);
        const bool take = oa >= 0 && (ba < 0 || ov > best || (ov == best && oa < ba));
        if (take) {
            best = ov;
            ba = oa;
        }
    }
    if (live && sub == 0) {
        best_action[b] = ba;
        if (best_value) best_value[b] = ba >= 0 ? (float)best : 0.f;      // = action_values[b][ba]: what callers used to gather
    }
}

inline dim3 grid_for(long long n, int block = kBlock) { return dim3((unsigned)((n + block - 1) / block)); }

// ------------------------------------------------------------------------------------------------
// workspace layout of the tree search
// ------------------------------------------------------------------------------------------------
struct LevelLayout {
    long long P;
    long long robot, humans, humans_next, child_robot, reward, child_value, value1, keep, backup, best_slot;
    long long reward_clip;      // level 0 only (-1 elsewhere): the root rewards as upstream's action_clip reads them
};

inline long long align_up(long long x) { return (x + 255) & ~255ll; }

inline int plan_levels(const MprlPlanner& pl, int B, int H, LevelLayout* lv, long long* total,
                       long long* scratch_off = nullptr, long long* scratch_bytes = nullptr) {
    const int A = pl.num_actions, D = pl.planning_depth;
    const int W = pl.do_action_clip ? pl.planning_width : A;
    long long off = 0, P = B;
    for (int l = 0; l < D; ++l) {
        LevelLayout& L = lv[l];
        L.P = P;
        L.robot = off;        off = align_up(off + (l == 0 ? 0 : P * 9 * 4));          // level 0 reads the caller's arrays
        L.humans = -1;                                                               // = previous level's humans_next
        L.humans_next = off;  off = align_up(off + P * H * 5 * 4);
        L.child_robot = off;  off = align_up(off + P * A * 9 * 4);
        L.reward = off;       off = align_up(off + P * A * 4);
        L.reward_clip = -1;
        if (l == 0) { L.reward_clip = off; off = align_up(off + P * A * 4); }
        L.child_value = off;  off = align_up(off + P * A * 4);
        L.value1 = off;       off = align_up(off + P * A * 4);
        L.keep = off;         off = align_up(off + P * W * 4);
        L.backup = off;       off = align_up(off + P * W * 4);
        L.best_slot = off;    off = align_up(off + P * 4);
        if (l > 0) lv[l].humans = lv[l - 1].humans_next;
        if (l + 1 < D) P *= W;
        if (P > (1ll << 31) / (A * 9)) return RGL_ERR_BAD_SHAPE;
    }
    // hand-off buffer of the two-stage value kernels, sized for the widest (= deepest) level
    const long long sb = (long long)rgl::value_children_workspace_bytes(&pl, (int)P, H);
    if (scratch_off) *scratch_off = off;
    if (scratch_bytes) *scratch_bytes = sb;
    off = align_up(off + sb);
    // split-f16 image of the state predictor's scene kernel, packed once per search unless the caller hands one in
    off = align_up(off + (long long)rgl::scene_image_bytes(&pl));
    *total = off;
    return RGL_OK;
}

inline int validate_planner(const MprlPlanner& pl, int H) {
    if (pl.num_actions < 1 || pl.num_actions > RGL_MAX_ACTIONS) return RGL_ERR_BAD_SHAPE;
    if (pl.planning_depth < 1 || pl.planning_depth > 8) return RGL_ERR_BAD_SHAPE;
    if (pl.do_action_clip && (pl.planning_width < 1 || pl.planning_width > pl.num_actions)) return RGL_ERR_BAD_SHAPE;
    if (pl.kinematics != RGL_HOLONOMIC && pl.kinematics != RGL_UNICYCLE) return RGL_ERR_BAD_MODE;
    if (!pl.actions) return RGL_ERR_NULL;
    if (pl.do_action_clip && pl.sparse_search && !pl.action_groups) return RGL_ERR_NULL;
    int rc = rgl::validate_graph(pl.value_graph, H);
    if (rc) return rc;
    rc = rgl::validate_mlp(pl.value_head, pl.value_graph.x_dim, 1);
    if (rc) return rc;
    if (pl.value_graph.w_r.dims[0] != 9 || pl.value_graph.w_h.dims[0] != 5) return RGL_ERR_BAD_SHAPE;
    if (!pl.linear_state_predictor) {
        rc = rgl::validate_graph(pl.predictor_graph, H);
        if (rc) return rc;
        rc = rgl::validate_mlp(pl.motion_head, pl.predictor_graph.x_dim, 5);
        if (rc) return rc;
        if (pl.predictor_graph.w_r.dims[0] != 9 || pl.predictor_graph.w_h.dims[0] != 5) return RGL_ERR_BAD_SHAPE;
    }
    return RGL_OK;
}

// One level: steps 1-3 of the header comment of mprl_expand_f32.
int expand_level(const MprlPlanner& pl, const float* robot, const float* humans, int humans_per, int P, int H, int joint,
                 float* humans_next, float* child_robot, float* reward, float* child_value, void* scratch,
                 size_t scratch_bytes, hipStream_t st, int image_ready = 0, const TailArgs* tail = nullptr,
                 int* tail_done = nullptr, const float* sp_image = nullptr, hipEvent_t before_children = nullptr,
                 float* reward_clip = nullptr) {
    const int A = pl.num_actions;
    ChildrenArgs ca;
    ca.reward_clip = reward_clip;
    ca.robot = robot; ca.humans = humans; ca.humans_per = humans_per; ca.actions = pl.actions;
    ca.P = P; ca.H = H; ca.A = A; ca.kinematics = pl.kinematics; ca.dt = pl.time_step; ca.joint = joint;
    ca.child_robot = child_robot; ca.reward = reward;
    ca.v_max = pl.action_speed_bound > 0.0 ? (float)pl.action_speed_bound * 1.0001f : 0.f;
    ca.p_base = ca.c_base = 0;
    const bool roots64 = joint && humans_per == 1 && pl.root_robot_f64 && pl.root_humans_f64;
    ca.robot64 = roots64 ? pl.root_robot_f64 : nullptr;
    ca.humans64 = roots64 ? pl.root_humans_f64 : nullptr;
    int children_done = 0;              // set when the state predictor's scene kernel ran them on its extra workgroups
    // (Round 6: the reward work inside the children launch -- every workgroup for the parents it owns, in its prologue under the
    // weight image's DMA, inputs staged in LDS -- was built and measured: the embedding launch drops from 13.6 / 15.6 to 7.9 / 7.3 us,
    // the children launches grow by 7.3 / 5.9: 277.6 against 279.0 us per 2048-root step, not worth a second home for that code;
    // profiles/r06_p_timeline_rewards_in_children_ab.md.  The step is ~2-3 us of divergent float64 arithmetic per wave wherever it runs.)
    // (Running mprl_children_kernel beside the state predictor on a side stream was measured: the two overlap -- 34 us together
    // instead of 27 + 15 -- but the event fork / join costs more than that on the critical path: 0.405 vs 0.398 ms per step.)
    if (pl.linear_state_predictor) {
        if (humans_per == 1) {
            hipLaunchKernelGGL(linear_humans_kernel, grid_for((long long)P * H), dim3(kBlock), 0, st, humans, humans_next,
                               (long long)P * H);
        } else {
            hipLaunchKernelGGL(gather_parent_humans_kernel, grid_for((long long)P * H * 5), dim3(kBlock), 0, st, humans,
                               humans_per, H, (long long)P, humans_next);
            hipLaunchKernelGGL(linear_humans_kernel, grid_for((long long)P * H), dim3(kBlock), 0, st, humans_next,
                               humans_next, (long long)P * H);
        }
        RGL_LAUNCH_CHECK();
    } else {
        int rc = rgl::launch_predict_humans(&pl, robot, humans, humans_per, P, H, humans_next, scratch, scratch_bytes, st,
                                            &ca, sizeof(ca), &children_done, sp_image);
        if (rc) return rc;
    }
    if (!children_done) {
        hipLaunchKernelGGL(mprl_children_kernel, grid_for((long long)P * A), dim3(kBlock), 0, st, ca);
        RGL_LAUNCH_CHECK();
    }
    if (before_children) RGL_HIP_TRY(hipEventRecord(before_children, st));     // traced searches: state predictor | children
    return rgl::launch_value_children(&pl, child_robot, humans_next, P, H, child_value, scratch, scratch_bytes, st, image_ready,
                                      tail, tail ? sizeof(TailArgs) : 0, tail_done);
}

}  // namespace

extern "C" int mprl_expand_f32(const MprlPlanner* planner, const float* robot, const float* humans, int P, int H,
                               int parents_are_joint_states, float* humans_next, float* child_robot, float* reward,
                               float* child_value, float* value1, void* workspace, size_t workspace_bytes,
                               rgl_stream_t stream) {
    if (!planner || !robot || !humans || !humans_next || !child_robot || !reward || !child_value) return RGL_ERR_NULL;
    if (P < 0) return RGL_ERR_BAD_SHAPE;
    int rc = validate_planner(*planner, H);
    if (rc) return rc;
    if (P == 0) return RGL_OK;
    hipStream_t st = (hipStream_t)stream;
    rc = expand_level(*planner, robot, humans, 1, P, H, parents_are_joint_states, humans_next, child_robot, reward,
                      child_value, workspace, workspace_bytes, st, 0, nullptr, nullptr, planner->predictor_image);
    if (rc) return rc;
    if (value1) {
        const long long n = (long long)P * planner->num_actions;
        hipLaunchKernelGGL(one_step_value_kernel, grid_for(n), dim3(kBlock), 0, st, reward, child_value,
                           (float)planner->gamma_bar, n, value1);
        RGL_LAUNCH_CHECK();
    }
    return RGL_OK;
}

extern "C" size_t mprl_value_children_workspace_bytes(const MprlPlanner* planner, int P, int H) {
    if (!planner || P < 1 || H < 1) return 0;
    return rgl::value_children_workspace_bytes(planner, P, H);
}

extern "C" int mprl_value_children_f32(const MprlPlanner* planner, const float* child_robot, const float* humans_next,
                                       int P, int H, float* child_value, void* workspace, size_t workspace_bytes,
                                       rgl_stream_t stream) {
    if (!planner || !child_robot || !humans_next || !child_value) return RGL_ERR_NULL;
    if (P < 0) return RGL_ERR_BAD_SHAPE;
    int rc = validate_planner(*planner, H);
    if (rc) return rc;
    if (P == 0) return RGL_OK;
    return rgl::launch_value_children(planner, child_robot, humans_next, P, H, child_value, workspace, workspace_bytes,
                                      (hipStream_t)stream);
}

extern "C" size_t mprl_predictor_image_bytes(const MprlPlanner* planner) { return rgl::scene_image_bytes(planner); }

extern "C" int mprl_pack_predictor_image_f32(const MprlPlanner* planner, float* image, size_t image_bytes, rgl_stream_t stream) {
    if (!planner || !image) return RGL_ERR_NULL;
    const size_t need = rgl::scene_image_bytes(planner);
    if (!need) return RGL_ERR_BAD_MODE;
    if (image_bytes < need) return RGL_ERR_WORKSPACE;
    return rgl::pack_scene_image(planner, image, (hipStream_t)stream);
}

extern "C" size_t mprl_tree_workspace_bytes(const MprlPlanner* planner, int B, int H) {
    if (!planner || B < 1 || planner->planning_depth < 1 || planner->planning_depth > 8) return 0;
    LevelLayout lv[8];
    long long total = 0;
    if (plan_levels(*planner, B, H, lv, &total)) return 0;
    return (size_t)total;
}

extern "C" int mprl_tree_level_view(const MprlPlanner* planner, int B, int H, int level, MprlLevelView* view) {
    if (!planner || !view) return RGL_ERR_NULL;
    if (planner->planning_depth < 1 || planner->planning_depth > 8 || level < 0 || level >= planner->planning_depth)
        return RGL_ERR_BAD_SHAPE;
    LevelLayout lv[8];
    long long total = 0;
    int rc = plan_levels(*planner, B, H, lv, &total);
    if (rc) return rc;
    const LevelLayout& L = lv[level];
    view->n_parents = L.P;
    view->robot_off = level == 0 ? -1 : L.robot;
    view->humans_off = L.humans;
    view->humans_next_off = L.humans_next;
    view->child_robot_off = L.child_robot;
    view->reward_off = L.reward;
    view->reward_clip_off = level == 0 && planner->do_action_clip ? L.reward_clip : -1;
    view->child_value_off = L.child_value;
    view->value1_off = L.value1;
    view->keep_off = L.keep;
    view->backup_off = L.backup;
    view->best_slot_off = L.best_slot;
    return RGL_OK;
}

namespace {
// `events`: null, or 3 * planning_depth + 1 hipEvent_t of the caller's (mprl_tree_search_traced_f32)
int tree_search(const MprlPlanner* planner, const float* robot, const float* humans, int B, int H,
                int roots_are_joint_states, void* workspace, size_t workspace_bytes,
                int* best_action, float* best_value, float* root_values, int* root_kept,
                rgl_stream_t stream, void* const* events) {
    if (!planner || !robot || !humans || !workspace || !best_action || !best_value) return RGL_ERR_NULL;
    if (B < 1) return RGL_ERR_BAD_SHAPE;
    int rc = validate_planner(*planner, H);
    if (rc) return rc;
    const MprlPlanner& pl = *planner;
    const int A = pl.num_actions, D = pl.planning_depth;
    const int W = pl.do_action_clip ? pl.planning_width : A;
    if (pl.do_action_clip && pl.sparse_search) {
        // the select kernel keeps the ids of the groups taken so far in kMaxSparseWidth registers
        if (W > kMaxSparseWidth) return RGL_ERR_BAD_MODE;
    }
    LevelLayout lv[8];
    long long total = 0, scratch_off = 0, scratch_bytes = 0;
    rc = plan_levels(pl, B, H, lv, &total, &scratch_off, &scratch_bytes);
    if (rc) return rc;
    if ((long long)workspace_bytes < total) return RGL_ERR_WORKSPACE;
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const float gamma_f = (float)pl.gamma_bar;

    // weight images of the value-of-children kernels: prepared once, every level copies them into LDS (0 = prepared)
    // (or handed in by the caller, packed once for fixed weights: MprlPlanner::children_image)
    const int image_mode = pl.contraction_dtype == RGL_CONTRACT_BF16X6 ? 2 : 0;
    const int image_ready = (pl.contraction_dtype == RGL_CONTRACT_F32 || image_mode != 0) &&
                            (pl.children_image != nullptr ||
                             rgl::pack_children_images(&pl.value_graph, &pl.value_head, (int)lv[D - 1].P, A, H, ws + scratch_off,
                                                       (size_t)scratch_bytes, st, image_mode) == 0);
    const float* sp_image = pl.predictor_image;
    if (!sp_image && rgl::scene_image_bytes(&pl)) {
        float* img = (float*)(ws + align_up(scratch_off + scratch_bytes));
        if (rgl::pack_scene_image(&pl, img, st) == RGL_OK) sp_image = img;
    }
    TailArgs tail{};
    tail.enabled = 1;
    tail.D = D; tail.A = A; tail.W = W; tail.clip = pl.do_action_clip; tail.sparse = pl.sparse_search;
    tail.gamma_f = gamma_f;
    tail.groups = pl.action_groups;
    for (int l = 0; l < D; ++l) {
        const LevelLayout& L = lv[l];
        tail.lv[l] = TailLevel{(const float*)(ws + L.reward), (const float*)(ws + L.child_value), (int*)(ws + L.keep),
                               (float*)(ws + L.backup), (int*)(ws + L.best_slot), (int)L.P};
    }
    tail.B = B;
    tail.best_action = best_action; tail.best_value = best_value; tail.root_values = root_values; tail.root_kept = root_kept;
    int chain_done = 0;              // the deepest level's children kernel also ran the back-up steps and the root step
    for (int l = 0; l < D; ++l) {
        const LevelLayout& L = lv[l];
        const int P = (int)L.P;
        const float* pr = l == 0 ? robot : (const float*)(ws + L.robot);
        const float* ph = l == 0 ? humans : (const float*)(ws + L.humans);
        const int humans_per = l == 0 ? 1 : W;
        const bool deepest = l + 1 == D;
        tail.level = l;
        tail.child_robot = (const float*)(ws + L.child_robot);
        tail.value1 = (float*)(ws + L.value1);
        tail.next_robot = deepest ? nullptr : (float*)(ws + lv[l + 1].robot);
        tail.chain = deepest;
        // Joint-state roots of a clipped search: upstream clips the root on the TENSOR state (model_predictive_rl.py:216-218 ->
        // :246-248, float32-born scalars) and prices the kept actions on the float64 JointState (:226) -- two reward arrays at
        // level 0, the first for the selection, the second for the root values.
        float* reward_clip = l == 0 && roots_are_joint_states && pl.do_action_clip ? (float*)(ws + L.reward_clip) : nullptr;
        tail.reward_sel = reward_clip;
        int tail_done = 0;           // 1: the children kernel selected for its parents; 2: ... and finished the search (deepest level)
        if (events) RGL_HIP_TRY(hipEventRecord((hipEvent_t)events[3 * l], st));
        rc = expand_level(pl, pr, ph, humans_per, P, H, l == 0 ? roots_are_joint_states : 0,
                          (float*)(ws + L.humans_next), (float*)(ws + L.child_robot), (float*)(ws + L.reward),
                          (float*)(ws + L.child_value), ws + scratch_off, (size_t)scratch_bytes, st, image_ready, &tail, &tail_done,
                          sp_image, events ? (hipEvent_t)events[3 * l + 1] : nullptr, reward_clip);
        if (rc) return rc;
        if (!tail_done) {
            // the deepest level's selection also writes the leaf values and (below the root) does its own back-up step
            hipLaunchKernelGGL(mprl_select_kernel, grid_for(P, 4), dim3(256), 0, st, tail);
            RGL_LAUNCH_CHECK();
        }
        if (events) RGL_HIP_TRY(hipEventRecord((hipEvent_t)events[3 * l + 2], st));
        if (deepest && tail_done == 2) chain_done = 1;
    }
    if (!chain_done) {
        for (int l = D - 2; l >= 1; --l) {
            hipLaunchKernelGGL(mprl_backup_kernel, grid_for(lv[l].P, 64), dim3(64), 0, st, tail, l);
            RGL_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(mprl_root_kernel, grid_for((long long)B * kRootLanes, 256), dim3(256), 0, st, tail);
        RGL_LAUNCH_CHECK();
    }
    if (events) RGL_HIP_TRY(hipEventRecord((hipEvent_t)events[3 * D], st));
    return RGL_OK;
}
}  // namespace

extern "C" int mprl_tree_search_f32(const MprlPlanner* planner, const float* robot, const float* humans, int B, int H,
                                    int roots_are_joint_states, void* workspace, size_t workspace_bytes,
                                    int* best_action, float* best_value, float* root_values, int* root_kept,
                                    rgl_stream_t stream) {
    return tree_search(planner, robot, humans, B, H, roots_are_joint_states, workspace, workspace_bytes, best_action, best_value,
                       root_values, root_kept, stream, nullptr);
}

extern "C" int mprl_tree_search_traced_f32(const MprlPlanner* planner, const float* robot, const float* humans, int B, int H,
                                           int roots_are_joint_states, void* workspace, size_t workspace_bytes,
                                           int* best_action, float* best_value, float* root_values, int* root_kept,
                                           rgl_stream_t stream, float* predictor_ms, float* children_ms, float* total_ms) {
    if (!planner || !predictor_ms || !children_ms) return RGL_ERR_NULL;
    const int D = planner->planning_depth;
    if (D < 1 || D > 8) return RGL_ERR_BAD_SHAPE;
    hipEvent_t ev[3 * 8 + 1];
    const int n = 3 * D + 1;
    for (int i = 0; i < n; ++i) {
        const hipError_t e = hipEventCreate(&ev[i]);
        if (e != hipSuccess) {                                   // give back what was created
            for (int k = 0; k < i; ++k) (void)hipEventDestroy(ev[k]);
            return (int)e;
        }
    }
    int rc = tree_search(planner, robot, humans, B, H, roots_are_joint_states, workspace, workspace_bytes, best_action, best_value,
                         root_values, root_kept, stream, reinterpret_cast<void* const*>(ev));
    if (rc == RGL_OK) rc = (int)hipEventSynchronize(ev[3 * D]);
    for (int l = 0; l < D && rc == RGL_OK; ++l) {
        rc = (int)hipEventElapsedTime(&predictor_ms[l], ev[3 * l], ev[3 * l + 1]);
        if (rc == RGL_OK) rc = (int)hipEventElapsedTime(&children_ms[l], ev[3 * l + 1], ev[3 * l + 2]);
    }
    if (rc == RGL_OK && total_ms) rc = (int)hipEventElapsedTime(total_ms, ev[0], ev[3 * D]);
    for (int i = 0; i < n; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

// estimate_reward + compute_next_state for every (parent, action) pair on their own (model_predictive_rl.py:304-357,
// state_predictor.py:41-60): the kernel every tree level runs, exported for ModelPredictiveRL.estimate_reward
extern "C" int mprl_estimate_reward_f32(const MprlPlanner* planner, const float* robot, const float* humans, int P, int H,
                                        int parents_are_joint_states, float* child_robot, float* reward, rgl_stream_t stream) {
    if (!planner || !robot || !humans || !child_robot || !reward) return RGL_ERR_NULL;
    if (P < 0 || H < 1 || H + 1 > RGL_MAX_NODES) return RGL_ERR_BAD_SHAPE;
    const MprlPlanner& pl = *planner;
    if (pl.num_actions < 1 || pl.num_actions > RGL_MAX_ACTIONS) return RGL_ERR_BAD_SHAPE;
    if (pl.kinematics != RGL_HOLONOMIC && pl.kinematics != RGL_UNICYCLE) return RGL_ERR_BAD_MODE;
    if (!pl.actions) return RGL_ERR_NULL;
    if (P == 0) return RGL_OK;
    ChildrenArgs ca;
    ca.reward_clip = nullptr;
    ca.robot = robot; ca.humans = humans; ca.humans_per = 1; ca.actions = pl.actions;
    ca.P = P; ca.H = H; ca.A = pl.num_actions; ca.kinematics = pl.kinematics; ca.dt = pl.time_step; ca.joint = parents_are_joint_states;
    ca.child_robot = child_robot; ca.reward = reward;
    ca.v_max = pl.action_speed_bound > 0.0 ? (float)pl.action_speed_bound * 1.0001f : 0.f;
    ca.p_base = ca.c_base = 0;
    const bool roots64 = parents_are_joint_states && pl.root_robot_f64 && pl.root_humans_f64;
    ca.robot64 = roots64 ? pl.root_robot_f64 : nullptr;
    ca.humans64 = roots64 ? pl.root_humans_f64 : nullptr;
    hipLaunchKernelGGL(mprl_children_kernel, grid_for((long long)P * pl.num_actions), dim3(kBlock), 0, (hipStream_t)stream, ca);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

// action_clip's selection on its own (model_predictive_rl.py:242-269): value1 = reward + gamma_bar * child_value, then the
// planning_width best actions per parent (argpartition semantics; sparse: one action per group) -- tail_select, the step every
// search level runs, without the gather of the next level's states
extern "C" int mprl_action_clip_f32(const MprlPlanner* planner, const float* reward, const float* child_value, int P,
                                    float* value1, int* keep, rgl_stream_t stream) {
    if (!planner || !reward || !child_value || !value1 || !keep) return RGL_ERR_NULL;
    if (P < 0) return RGL_ERR_BAD_SHAPE;
    const MprlPlanner& pl = *planner;
    const int A = pl.num_actions;
    if (A < 1 || A > RGL_MAX_ACTIONS) return RGL_ERR_BAD_SHAPE;
    if (pl.do_action_clip && (pl.planning_width < 1 || pl.planning_width > A)) return RGL_ERR_BAD_SHAPE;
    const int W = pl.do_action_clip ? pl.planning_width : A;
    if (pl.do_action_clip && pl.sparse_search) {
        if (!pl.action_groups) return RGL_ERR_NULL;
        if (W > kMaxSparseWidth) return RGL_ERR_BAD_MODE;
    }
    if (P == 0) return RGL_OK;
    TailArgs t{};
    t.enabled = 1;
    t.level = 0; t.D = 2;                       // "not the deepest level": no leaf values, no back-up step
    t.A = A; t.W = W; t.clip = pl.do_action_clip; t.sparse = pl.sparse_search;
    t.gamma_f = (float)pl.gamma_bar;
    t.groups = pl.action_groups;
    t.child_robot = nullptr; t.next_robot = nullptr;
    t.value1 = value1;
    t.lv[0] = TailLevel{reward, child_value, keep, nullptr, nullptr, P};
    t.B = P;
    hipLaunchKernelGGL(mprl_select_kernel, grid_for(P, 4), dim3(256), 0, (hipStream_t)stream, t);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

// ------------------------------------------------------------------------------------------------
// path G entry points
// ------------------------------------------------------------------------------------------------
extern "C" int gcn_rotate_f32(const float* joint14, float* rotated13, int n_rows, int kinematics,
                              rgl_stream_t stream) {
    if (!joint14 || !rotated13) return RGL_ERR_NULL;
    if (n_rows < 0) return RGL_ERR_BAD_SHAPE;
    if (kinematics != RGL_HOLONOMIC && kinematics != RGL_UNICYCLE) return RGL_ERR_BAD_MODE;
    if (n_rows == 0) return RGL_OK;
    hipLaunchKernelGGL(gcn_rotate_kernel, grid_for(n_rows), dim3(kBlock), 0, (hipStream_t)stream, joint14, rotated13,
                       n_rows, kinematics == RGL_UNICYCLE);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

// propagate + rotate + compute_reward for the B x A candidate scenes of a one-step search on their own (ABI 5): the first launch of
// gcn_predict_f32 (cadrl.py:113-138,241-276, multi_human_rl.py:46-51,73-96), exported for GCN.compute_reward / tests
extern "C" int gcn_prepare_f32(const GcnPlanner* planner, const float* robot, const float* humans, int B, int H,
                               float* self6, float* hum7, float* reward, rgl_stream_t stream) {
    if (!planner || !robot || !humans || !self6 || !hum7 || !reward) return RGL_ERR_NULL;
    if (B < 1 || H < 1 || H + 1 > RGL_MAX_NODES) return RGL_ERR_BAD_SHAPE;
    const GcnPlanner& pl = *planner;
    if (pl.num_actions < 1 || pl.num_actions > RGL_MAX_ACTIONS || !pl.actions) return RGL_ERR_BAD_SHAPE;
    if (pl.kinematics != RGL_HOLONOMIC && pl.kinematics != RGL_UNICYCLE) return RGL_ERR_BAD_MODE;
    const long long S = (long long)B * pl.num_actions;
    const int prep_threads = (H >= kBlock ? 1 : kBlock / H) * H;        // whole (root, action) groups per workgroup
    const bool r64 = pl.root_robot_f64 && pl.root_humans_f64;
    hipLaunchKernelGGL(gcn_prepare_kernel, grid_for(S * H, prep_threads), dim3(prep_threads), prep_threads * sizeof(double),
                       (hipStream_t)stream, robot, humans, r64 ? pl.root_robot_f64 : nullptr, r64 ? pl.root_humans_f64 : nullptr,
                       pl.actions, B, H, pl.num_actions, pl.kinematics, pl.time_step, self6, hum7, reward);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

constexpr long long kGcnImageBytes = 64 * 1024;      // >= scene_image_bytes_for(): Wa + 4 layer matrices + motion-head slots as 6-byte weights

extern "C" size_t gcn_predict_workspace_bytes(int B, int H, int A) {
    if (B < 1 || H < 1 || A < 1) return 0;
    const long long S = (long long)B * A;
    // rotated features, rewards, values + the scratch of the MFMA forward (embeddings [S][32], [S][H][32], value rows [S][64])
    // + room for the scene kernel's three-piece bf16 weight image (RGL_CONTRACT_BF16X6: packed per call, 4 layer matrices at most)
    return (size_t)(align_up(S * 6 * 4) + align_up(S * H * 7 * 4) + align_up(S * 4) + align_up(S * 4) +
                    align_up(S * (32 + (long long)H * 32 + 64) * 4) + kGcnImageBytes);
}

extern "C" int gcn_predict_f32(const GcnPlanner* planner, const float* robot, const float* humans, int B, int H,
                               void* workspace, size_t workspace_bytes, float* action_values, int* best_action,
                               float* best_value, rgl_stream_t stream) {
    if (!planner || !robot || !humans || !workspace || !action_values || !best_action) return RGL_ERR_NULL;
    if (B < 1) return RGL_ERR_BAD_SHAPE;
    const GcnPlanner& pl = *planner;
    if (pl.num_actions < 1 || pl.num_actions > RGL_MAX_ACTIONS || !pl.actions) return RGL_ERR_BAD_SHAPE;
    if (pl.kinematics != RGL_HOLONOMIC && pl.kinematics != RGL_UNICYCLE) return RGL_ERR_BAD_MODE;
    int rc = rgl::validate_graph(pl.graph, H);
    if (rc) return rc;
    if (pl.graph.w_r.dims[0] != 6 || pl.graph.w_h.dims[0] != 7) return RGL_ERR_BAD_SHAPE;
    rc = rgl::validate_mlp(pl.value_head, pl.graph.x_dim, 1);
    if (rc) return rc;
    if (pl.contraction_dtype != RGL_CONTRACT_F32 && pl.contraction_dtype != RGL_CONTRACT_BF16X6) return RGL_ERR_BAD_MODE;
    const int A = pl.num_actions;
    if (workspace_bytes < gcn_predict_workspace_bytes(B, H, A)) return RGL_ERR_WORKSPACE;
    const long long S = (long long)B * A;
    char* ws = (char*)workspace;
    float* self6 = (float*)ws;              ws += align_up(S * 6 * 4);
    float* hum7 = (float*)ws;               ws += align_up(S * H * 7 * 4);
    float* reward = (float*)ws;             ws += align_up(S * 4);
    float* value = (float*)ws;              ws += align_up(S * 4);
    void* fwd_ws = ws;
    const size_t fwd_bytes = (size_t)align_up(S * (32 + (long long)H * 32 + 64) * 4);
    hipStream_t st = (hipStream_t)stream;
    // RGL_CONTRACT_BF16X6: the graph's Wa / W_l as three-piece bf16 fragments for the scene kernel's value-rows mode, packed here
    // (one ~4 us launch beside a forward of hundreds; the weights may have changed since the last call)
    const float* rows_image = nullptr;
    if (pl.contraction_dtype == RGL_CONTRACT_BF16X6) {
        const size_t ib = rgl::scene_image_bytes_for(pl.graph, nullptr);
        if (ib && ib <= kGcnImageBytes) {
            float* img = (float*)(ws + fwd_bytes);
            if (rgl::pack_scene_image_for(pl.graph, nullptr, img, st) == RGL_OK) rows_image = img;
        }
    }
    const int prep_threads = (H >= kBlock ? 1 : kBlock / H) * H;        // whole (root, action) groups per workgroup
    hipLaunchKernelGGL(gcn_prepare_kernel, grid_for(S * H, prep_threads), dim3(prep_threads), prep_threads * sizeof(double), st, robot, humans,
                       pl.root_robot_f64 && pl.root_humans_f64 ? pl.root_robot_f64 : nullptr,
                       pl.root_robot_f64 && pl.root_humans_f64 ? pl.root_humans_f64 : nullptr, pl.actions, B, H, A,
                       pl.kinematics, pl.time_step, self6, hum7, reward);
    RGL_LAUNCH_CHECK();
    // the B x A rotated scenes: one wave per scene on the MFMA kernel where it covers the model (the shipped ValueNetwork: 6 / 7
    // inputs, 64-32 embeddings, head 150-100-100-1), else the general kernel
    rc = rgl::launch_scene_forward(&pl.graph, &pl.value_head, nullptr, self6, hum7, (int)S, 1, H, value, nullptr, fwd_ws, fwd_bytes,
                                   st, rows_image);
    if (rc == 1)      // other embedding MLPs (x_dim 32: what this workspace is sized for): the tile kernels of rgl_backward_mfma.hip
        rc = rgl::launch_tiles_forward(&pl.graph, &pl.value_head, nullptr, self6, hum7, (int)S, 1, H, nullptr, value, nullptr, fwd_ws,
                                       fwd_bytes, st);
    if (rc == 1) {
        const char* e = getenv("RGL_REQUIRE_MFMA_FORWARD");          // tests: refuse instead of running the general VALU kernel
        if (e && e[0] == '1') return RGL_ERR_BAD_MODE;
        rc = rgl::launch_generic_forward(&pl.graph, &pl.value_head, nullptr, self6, hum7, (int)S, 1, H, nullptr, nullptr, value,
                                         nullptr, st);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(gcn_argmax_kernel, grid_for((long long)B * kRootLanes, 256), dim3(256), 0, st, robot, reward, value, B, A, pl.gamma,
                       pl.time_step, action_values, best_action, best_value);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}
