// rgl_train.hip -- the two element-wise ends of an optimisation step that are neither a forward nor a backward of the networks
// (crowd_nav/utils/trainer.py:110-161: `for data in self.data_loader` ... `loss = self.criterion(outputs, target_values)` ...
// `loss.backward()` ... `v_losses += loss.data.item()`):
//   rgl_gather_rows_f32   the batch's rows of EVERY field of the replay memory in one launch (one index_select per field otherwise)
//   rgl_mse_step_f32      mean squared error, its gradient with respect to the prediction, the running sum of the reported losses and
//                         -- value update -- the bootstrapped target r + gamma V'(s') itself, in one launch (torch: mul, add,
//                         mse, mean, two fills, mse_backward, a cast and an add)
// A captured batch-100 step is launch-bound (43 nodes of a few microseconds, profiles/r05b_trainer_step_nodes.md): these two take 16
// of them away.  HBM-bound element-wise work; nothing here for the matrix pipe.
#include <hip/hip_runtime.h>

#include "rgl_common.h"
#include "rgl_hip.h"

namespace {

constexpr int kGatherJobs = 8;
struct GatherBatch {
    RglGatherJob job[kGatherJobs];
    int first_block[kGatherJobs + 1];
    int n, n_index;
    const long long* index;
};

// one thread per destination float; a field's rows are contiguous, so a wave reads and writes whole 256-byte runs of one or two rows
__global__ __launch_bounds__(256) void gather_rows_kernel(const GatherBatch b) {
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.first_block[j + 1]) ++j;
    const RglGatherJob J = b.job[j];
    const long long e = (long long)((int)blockIdx.x - b.first_block[j]) * 256 + threadIdx.x;
    if (e >= (long long)b.n_index * J.row_floats) return;
    const int r = (int)(e / J.row_floats), c = (int)(e - (long long)r * J.row_floats);
    const long long row = b.index[r];
    // an index outside the field is the caller's bug (torch's index_select traps on it): a NaN that no loss survives, not a wild read
    J.dst[e] = (row >= 0 && row < (long long)J.src_rows) ? J.src[row * J.row_floats + c] : __builtin_nanf("");
}

constexpr int kMseThreads = 1024;
constexpr int kMsePerBlock = 8 * kMseThreads;      // floats per workgroup before a launch takes a second one
constexpr int kMseMaxBlocks = RGL_MSE_WORKSPACE_BYTES / 8 - 1;
struct MseArgs {
    const float* out;
    const float* target;          // or null: target_i = reward_i + gamma * next_value_i
    const float* reward;
    const float* next_value;
    float gamma, grad_scale;      // grad_scale = (float)(2.0 / n): torch's mse_backward multiplies (out - target) by exactly this
    int n;
    float* grad;
    double* loss_sum;
    double* partial;              // workspace: [kMseMaxBlocks] partial sums, then the arrival counter (launches of several workgroups)
};

// Workgroup b owns the elements b * 1024 + t, + gridDim * 1024, ..: the partial sums, and the order in which the LAST workgroup to
// finish adds them up, do not depend on the scheduling -- the reported loss is the same number on every run.  A batch of values
// (n = 100) is one workgroup and touches no workspace.
__global__ __launch_bounds__(kMseThreads) void mse_step_kernel(const MseArgs a) {
    __shared__ double part[kMseThreads / 64];
    double s = 0.0;
    for (int i = blockIdx.x * kMseThreads + threadIdx.x; i < a.n; i += gridDim.x * kMseThreads) {
        // two roundings, as `rewards + gamma_bar * V` has upstream -- two kernels there -- so no contraction into a fused multiply-add
        // here (hipcc's default is -ffp-contract=fast, and HIP's __fmul_rn is a plain product that it contracts all the same)
#pragma clang fp contract(off)
        const float t = a.target ? a.target[i] : a.reward[i] + a.gamma * a.next_value[i];
        const float d = a.out[i] - t;
        a.grad[i] = a.grad_scale * d;
        s += (double)(d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tot = 0.0;
    for (int w = 0; w < kMseThreads / 64; ++w) tot += part[w];
    if (gridDim.x > 1) {
        unsigned* arrivals = reinterpret_cast<unsigned*>(a.partial + kMseMaxBlocks);
        a.partial[blockIdx.x] = tot;
        __threadfence();
        if (atomicAdd(arrivals, 1u) != gridDim.x - 1) return;
        __threadfence();
        tot = 0.0;
        for (unsigned b = 0; b < gridDim.x; ++b) tot += __hip_atomic_load(&a.partial[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *arrivals = 0u;                        // ready for the next launch (launches on one stream are ordered)
    }
    // the batch's loss as upstream reports it -- a float32 scalar -- added to the call's running float64 sum
    a.loss_sum[0] += (double)(float)(tot / (double)a.n);
}

}  // namespace

extern "C" int rgl_gather_rows_f32(const RglGatherJob* jobs, int n_jobs, const long long* index, int n_index, rgl_stream_t stream) {
    if (n_jobs < 0 || n_index < 0) return RGL_ERR_BAD_SHAPE;
    if (n_jobs == 0 || n_index == 0) return RGL_OK;
    if (!jobs || !index) return RGL_ERR_NULL;
    for (int j = 0; j < n_jobs; ++j) {
        if (!jobs[j].src || !jobs[j].dst) return RGL_ERR_NULL;
        if (jobs[j].row_floats < 1 || jobs[j].src_rows < 1 || (long long)jobs[j].row_floats * n_index > (1ll << 30)) return RGL_ERR_BAD_SHAPE;
    }
    for (int lo = 0; lo < n_jobs; lo += kGatherJobs) {
        GatherBatch b;
        b.n = n_jobs - lo < kGatherJobs ? n_jobs - lo : kGatherJobs;
        b.n_index = n_index;
        b.index = index;
        int blocks = 0;
        for (int j = 0; j < b.n; ++j) {
            b.job[j] = jobs[lo + j];
            b.first_block[j] = blocks;
            blocks += (int)(((long long)jobs[lo + j].row_floats * n_index + 255) / 256);
        }
        b.first_block[b.n] = blocks;
        hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
        RGL_LAUNCH_CHECK();
    }
    return RGL_OK;
}

extern "C" int rgl_mse_step_f32(const float* out, const float* target, const float* reward, const float* next_value, float gamma, int n,
                                float* grad, double* loss_sum, void* workspace, rgl_stream_t stream) {
    if (n < 1) return RGL_ERR_BAD_SHAPE;
    if (!out || !grad || !loss_sum || (!target && (!reward || !next_value))) return RGL_ERR_NULL;
    int blocks = (n + kMsePerBlock - 1) / kMsePerBlock;
    blocks = blocks > kMseMaxBlocks ? kMseMaxBlocks : blocks;
    if (blocks > 1 && !workspace) return RGL_ERR_WORKSPACE;
    MseArgs a;
    a.out = out; a.target = target; a.reward = reward; a.next_value = next_value;
    a.gamma = gamma; a.grad_scale = (float)(2.0 / (double)n); a.n = n;
    a.grad = grad; a.loss_sum = loss_sum; a.partial = (double*)workspace;
    hipLaunchKernelGGL(mse_step_kernel, dim3(blocks), dim3(kMseThreads), 0, (hipStream_t)stream, a);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}
