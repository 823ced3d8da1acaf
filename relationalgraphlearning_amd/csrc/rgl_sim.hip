// rgl_sim.hip -- batched crowd-simulator step: B independent environments advance one time step on device.
// Per environment: human actions (linear-to-goal / constant velocity / supplied), robot-vs-human closest-approach
// test over the step, goal test, the reward / termination ladder with the configured constants, kinematic update of
// every agent, clock.  All arithmetic in float64 like the reference's python floats.
//
// Follows (reference paths): crowd_sim/envs/crowd_sim.py:252-368 (step), crowd_sim/envs/utils/agent.py:113-139
// (compute_position, step), crowd_sim/envs/policy/linear.py:16-22 (Linear.predict), crowd_sim/envs/utils/utils.py:4-26.
#include "rgl_common.h"

namespace {

__device__ __forceinline__ double seg_dist_origin(double px, double py, double ex, double ey) {
    const double sx = ex - px, sy = ey - py;
    if (sx == 0.0 && sy == 0.0) return sqrt(px * px + py * py);
    double u = ((0.0 - px) * sx + (0.0 - py) * sy) / (sx * sx + sy * sy);
    u = u > 1.0 ? 1.0 : (u < 0.0 ? 0.0 : u);
    const double cx = px + u * sx, cy = py + u * sy;
    return sqrt(cx * cx + cy * cy);
}

// one thread per environment
__global__ void crowd_step_kernel(const CrowdSimConfig cfg, double* __restrict__ robot, double* __restrict__ humans,
                                  const double* __restrict__ human_goals, const double* __restrict__ human_vpref,
                                  const double* __restrict__ robot_action, const double* __restrict__ human_actions,
                                  double* __restrict__ time, int* __restrict__ done, int B, int H, int update,
                                  float* __restrict__ reward, int* __restrict__ info, double* __restrict__ dmin_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (done[b]) {                       // finished episodes are frozen
        reward[b] = 0.f;
        info[b] = CROWD_INFO_DONE;
        dmin_out[b] = INFINITY;
        return;
    }
    double* r = robot + (size_t)b * 9;
    double* hs = humans + (size_t)b * H * 5;
    const double dt = cfg.time_step;
    const double a0 = robot_action[2 * b], a1 = robot_action[2 * b + 1];
    double avx, avy, npx, npy, ntheta = r[8];
    if (cfg.kinematics == RGL_HOLONOMIC) {
        avx = a0; avy = a1;
        npx = r[0] + a0 * dt; npy = r[1] + a1 * dt;
    } else {
        const double th = r[8] + a1;
        avx = a0 * cos(a1 + r[8]); avy = a0 * sin(a1 + r[8]);
        npx = r[0] + cos(th) * a0 * dt; npy = r[1] + sin(th) * a0 * dt;
        ntheta = fmod(th, 2.0 * M_PI);
        if (ntheta < 0.0) ntheta += 2.0 * M_PI;          // python's % returns a non-negative remainder
    }
    bool collision = false;
    double dmin = INFINITY;
    for (int h = 0; h < H; ++h) {
        const double px = hs[h * 5] - r[0], py = hs[h * 5 + 1] - r[1];
        const double vx = hs[h * 5 + 2] - avx, vy = hs[h * 5 + 3] - avy;
        const double d = seg_dist_origin(px, py, px + vx * dt, py + vy * dt) - hs[h * 5 + 4] - r[4];
        if (d < 0.0) { collision = true; break; }          // the reference stops at the first collision
        if (d < dmin) dmin = d;
    }
    const double gx = npx - r[5], gy = npy - r[6];
    const bool reaching = sqrt(gx * gx + gy * gy) < r[4];
    double rew = 0.0;
    int code = CROWD_INFO_NOTHING, fin = 0;
    if (time[b] >= cfg.time_limit - 1.0) { rew = 0.0; fin = 1; code = CROWD_INFO_TIMEOUT; }
    else if (collision) { rew = cfg.collision_penalty; fin = 1; code = CROWD_INFO_COLLISION; }
    else if (reaching) { rew = cfg.success_reward; fin = 1; code = CROWD_INFO_REACH_GOAL; }
    else if (dmin < cfg.discomfort_dist) { rew = (dmin - cfg.discomfort_dist) * cfg.discomfort_penalty_factor * dt; code = CROWD_INFO_DISCOMFORT; }
    reward[b] = (float)rew;
    info[b] = code;
    dmin_out[b] = collision ? -1.0 : dmin;
    if (!update) return;
    // humans act on the state BEFORE anyone moves (crowd_sim.py:257-268), then everybody steps
    for (int h = 0; h < H; ++h) {
        double hvx, hvy;
        if (cfg.human_policy == CROWD_HUMAN_LINEAR) {
            const double th = atan2(human_goals[((size_t)b * H + h) * 2 + 1] - hs[h * 5 + 1],
                                    human_goals[((size_t)b * H + h) * 2] - hs[h * 5]);
            const double vp = human_vpref[(size_t)b * H + h];
            hvx = cos(th) * vp; hvy = sin(th) * vp;
        } else if (cfg.human_policy == CROWD_HUMAN_CONSTANT_VELOCITY) {
            hvx = hs[h * 5 + 2]; hvy = hs[h * 5 + 3];
        } else {
            hvx = human_actions[((size_t)b * H + h) * 2]; hvy = human_actions[((size_t)b * H + h) * 2 + 1];
        }
        hs[h * 5] += hvx * dt;
        hs[h * 5 + 1] += hvy * dt;
        hs[h * 5 + 2] = hvx;
        hs[h * 5 + 3] = hvy;
    }
    r[0] = npx; r[1] = npy;
    if (cfg.kinematics == RGL_HOLONOMIC) { r[2] = a0; r[3] = a1; }
    else { r[8] = ntheta; r[2] = a0 * cos(ntheta); r[3] = a0 * sin(ntheta); }
    time[b] += dt;
    done[b] = fin;
}

__global__ void crowd_observe_kernel(const double* __restrict__ robot, const double* __restrict__ humans, int B, int H,
                                     float* __restrict__ robot32, float* __restrict__ humans32) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * 9) robot32[i] = (float)robot[i];
    if (i < B * H * 5) humans32[i] = (float)humans[i];
}

}  // namespace

extern "C" int crowd_step_f64(const CrowdSimConfig* cfg, double* robot, double* humans, const double* human_goals,
                              const double* human_vpref, const double* robot_action, const double* human_actions,
                              double* time, int* done, int B, int H, int update, float* reward, int* info, double* dmin,
                              rgl_stream_t stream) {
    if (!cfg || !robot || !humans || !robot_action || !time || !done || !reward || !info || !dmin) return RGL_ERR_NULL;
    if (B < 1 || H < 1) return RGL_ERR_BAD_SHAPE;
    if (cfg->kinematics != RGL_HOLONOMIC && cfg->kinematics != RGL_UNICYCLE) return RGL_ERR_BAD_MODE;
    if (cfg->human_policy == CROWD_HUMAN_LINEAR && (!human_goals || !human_vpref)) return RGL_ERR_NULL;
    if (cfg->human_policy == CROWD_HUMAN_GIVEN && update && !human_actions) return RGL_ERR_NULL;
    if (cfg->human_policy < CROWD_HUMAN_GIVEN || cfg->human_policy > CROWD_HUMAN_CONSTANT_VELOCITY) return RGL_ERR_BAD_MODE;
    hipLaunchKernelGGL(crowd_step_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, *cfg, robot, humans,
                       human_goals, human_vpref, robot_action, human_actions, time, done, B, H, update, reward, info, dmin);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

extern "C" int crowd_observe_f32(const double* robot, const double* humans, int B, int H, float* robot32, float* humans32,
                                 rgl_stream_t stream) {
    if (!robot || !humans || !robot32 || !humans32) return RGL_ERR_NULL;
    if (B < 1 || H < 1) return RGL_ERR_BAD_SHAPE;
    const int n = B * (H * 5 > 9 ? H * 5 : 9);
    hipLaunchKernelGGL(crowd_observe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, robot, humans, B, H,
                       robot32, humans32);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}
