// rgl_generic.hip -- the general relational-graph forward: one 256-thread workgroup per scene,
// every similarity function / layerwise / skip / head combination of the reference, any widths
// within the ABI limits.  Activations live in LDS; weights are read k-major from global memory
// (coalesced across the output index, served by L1/L2).  This kernel is the complete,
// correctness-first path; rgl_fast.hip holds the MFMA kernels for the shipped configuration.
//
// Follows (reference paths): crowd_nav/policy/graph_model.py:63-130, value_estimator.py:11-20,
// state_predictor.py:28-36, gcn.py:95-128, helpers.py:5-13.
#include "rgl_common.h"

#include <cstdlib>

namespace {

constexpr int kThreads = 256;

struct SceneLds {           // float offsets into dynamic LDS
    int in_r, in_h;         // raw robot row, raw human rows
    int buf0, buf1, buf_ld; // MLP ping-pong, row stride
    int X, Hc, Hn, T;       // node features: embedding, current layer, next layer, A*H / X*Wa
    int S, s_ld;            // N x N similarity / adjacency (row stride s_ld)
    int U, V, uv_ld;        // concatenation mode: per-node halves of the pair MLP's first layer
    int total;
};

// rows x (dims[0] -> ... -> dims[n]) MLP on LDS rows; result written to dst (LDS or global).
__device__ __forceinline__ void mlp_rows(const RglMlp& m, const float* src, int src_ld, int rows,
                                         float* b0, float* b1, int buf_ld, float* dst, int dst_ld) {
    const float* cur = src;
    int cur_ld = src_ld;
#pragma unroll
    for (int l = 0; l < RGL_MAX_MLP_LAYERS; ++l) {
        if (l < m.n_layers) {
            const int in = m.dims[l], out = m.dims[l + 1];
            const bool last = (l == m.n_layers - 1);
            float* o = last ? dst : ((l & 1) ? b1 : b0);
            const int o_ld = last ? dst_ld : buf_ld;
            const bool relu = !last || m.last_relu;
            const float* __restrict__ W = m.weight[l];
            const float* __restrict__ bias = m.bias[l];
            for (int idx = threadIdx.x; idx < rows * out; idx += kThreads) {
                const int i = idx / out, j = idx - i * out;
                float acc = bias[j];
                const float* a = cur + i * cur_ld;
                for (int k = 0; k < in; ++k) acc = fmaf(a[k], W[k * out + j], acc);
                o[i * o_ld + j] = relu ? fmaxf(acc, 0.f) : acc;
            }
            __syncthreads();
            cur = o;
            cur_ld = o_ld;
        }
    }
}

// dst[i][j] = sum_k a[i][k] * W[k][j]   (W row-major [xd][xd] in global), rows i < rows
__device__ __forceinline__ void rows_times_w(const float* a, int rows, int xd, const float* __restrict__ W,
                                             float* dst) {
    for (int idx = threadIdx.x; idx < rows * xd; idx += kThreads) {
        const int i = idx / xd, j = idx - i * xd;
        float acc = 0.f;
        for (int k = 0; k < xd; ++k) acc = fmaf(a[i * xd + k], W[k * xd + j], acc);
        dst[idx] = acc;
    }
}

__device__ __forceinline__ void softmax_rows(float* S, int N, int ld, int rows) {
    for (int i = threadIdx.x; i < rows; i += kThreads) {
        float* r = S + i * ld;
        float mx = r[0];
        for (int j = 1; j < N; ++j) mx = fmaxf(mx, r[j]);
        float sum = 0.f;
        for (int j = 0; j < N; ++j) {
            const float e = expf(r[j] - mx);
            r[j] = e;
            sum += e;
        }
        for (int j = 0; j < N; ++j) r[j] = r[j] / sum;
    }
}

// Adjacency of node features F (N x xd) into S; `rows` = how many leading rows are needed.
__device__ void similarity(const RglGraph& g, const float* F, int N, float* T, float* S, int s_ld,
                           float* U, float* V, int uv_ld, int rows) {
    const int xd = g.x_dim;
    const int mode = g.similarity;
    if (mode == RGL_SIM_EQUAL_ATTENTION || mode == RGL_SIM_DIAGONAL) {
        const float inv = 1.0f / (float)N;
        for (int idx = threadIdx.x; idx < N * N; idx += kThreads) {
            const int i = idx / N, j = idx - i * N;
            S[i * s_ld + j] = (mode == RGL_SIM_DIAGONAL) ? (i == j ? 1.f : 0.f) : inv;
        }
        __syncthreads();
        return;
    }
    if (mode == RGL_SIM_CONCATENATION) {
        // pair MLP relu(w1 . relu(W0 [x_i | x_j] + b0) + b1): the first layer splits into a per-i
        // half U (bias folded in) and a per-j half V, so pairs cost O(hidden) instead of O(hidden*2X).
        const RglMlp& p = g.w_a_mlp;
        const int hid = p.dims[1];
        const float* __restrict__ W0 = p.weight[0];
        const float* __restrict__ b0 = p.bias[0];
        for (int idx = threadIdx.x; idx < N * hid; idx += kThreads) {
            const int i = idx / hid, h = idx - i * hid;
            float u = b0[h], v = 0.f;
            for (int k = 0; k < xd; ++k) {
                u = fmaf(F[i * xd + k], W0[k * hid + h], u);
                v = fmaf(F[i * xd + k], W0[(xd + k) * hid + h], v);
            }
            U[i * uv_ld + h] = u;
            V[i * uv_ld + h] = v;
        }
        __syncthreads();
        const float* __restrict__ w1 = p.weight[1];
        const float b1 = p.bias[1][0];
        for (int idx = threadIdx.x; idx < rows * N; idx += kThreads) {
            const int i = idx / N, j = idx - i * N;
            float acc = b1;
            for (int h = 0; h < hid; ++h) acc = fmaf(fmaxf(U[i * uv_ld + h] + V[j * uv_ld + h], 0.f), w1[h], acc);
            S[i * s_ld + j] = fmaxf(acc, 0.f);        // last_relu=True, and NOT normalised (reference behaviour)
        }
        __syncthreads();
        return;
    }
    const float* L = F;                                // left operand of L * F^T
    int lrows = rows;
    if (mode == RGL_SIM_EMBEDDED_GAUSSIAN) {
        rows_times_w(F, rows, xd, g.w_a, T);
        __syncthreads();
        L = T;
    } else if (mode == RGL_SIM_COSINE || mode == RGL_SIM_COSINE_SOFTMAX) {
        lrows = N;                                     // the normaliser needs every row's norm
    }
    for (int idx = threadIdx.x; idx < lrows * N; idx += kThreads) {
        const int i = idx / N, j = idx - i * N;
        float acc = 0.f;
        for (int k = 0; k < xd; ++k) acc = fmaf(L[i * xd + k], F[j * xd + k], acc);
        S[i * s_ld + j] = acc;
    }
    __syncthreads();
    if (mode == RGL_SIM_EMBEDDED_GAUSSIAN || mode == RGL_SIM_GAUSSIAN) {
        softmax_rows(S, N, s_ld, rows);
    } else if (mode == RGL_SIM_SQUARED) {
        for (int i = threadIdx.x; i < rows; i += kThreads) {
            float* r = S + i * s_ld;
            float sum = 0.f;
            for (int j = 0; j < N; ++j) {
                r[j] = r[j] * r[j];
                sum += r[j];
            }
            for (int j = 0; j < N; ++j) r[j] = r[j] / sum;
        }
    } else {  // cosine / cosine_softmax: S_ij / (|S_i| |S_j|) with |S_i| the 2-norm of ROW i of S
        float* nrm = T;                                // N scratch floats
        for (int i = threadIdx.x; i < N; i += kThreads) {
            float sq = 0.f;
            for (int j = 0; j < N; ++j) sq = fmaf(S[i * s_ld + j], S[i * s_ld + j], sq);
            nrm[i] = sqrtf(sq);
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < rows * N; idx += kThreads) {
            const int i = idx / N, j = idx - i * N;
            S[i * s_ld + j] = S[i * s_ld + j] / (nrm[i] * nrm[j]);
        }
        __syncthreads();
        if (mode == RGL_SIM_COSINE_SOFTMAX) softmax_rows(S, N, s_ld, rows);
    }
    __syncthreads();
}

struct ForwardArgs {
    RglGraph g;
    RglMlp vhead, mhead;
    int has_vhead, has_mhead;
    const float* robot;
    const float* humans;
    int n_scenes, scenes_per_crowd, H;
    float* H_out;
    float* A_out;
    float* value_out;
    float* humans_next;
    SceneLds L;
};

__global__ __launch_bounds__(kThreads) void rgl_scene_forward_kernel(const ForwardArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const RglGraph& g = a.g;
    const SceneLds& L = a.L;
    const int H = a.H, N = H + 1, xd = g.x_dim;
    const int rd = g.w_r.dims[0], hd = g.w_h.dims[0];
    float* in_r = lds + L.in_r;
    float* in_h = lds + L.in_h;
    float* b0 = lds + L.buf0;
    float* b1 = lds + L.buf1;
    float* X = lds + L.X;
    float* Hc = lds + L.Hc;
    float* Hn = lds + L.Hn;
    float* T = lds + L.T;
    float* S = lds + L.S;
    float* U = lds + L.U;
    float* V = lds + L.V;

    for (int s = blockIdx.x; s < a.n_scenes; s += gridDim.x) {
        const float* rsrc = a.robot + (size_t)s * rd;
        const float* hsrc = a.humans + (size_t)(s / a.scenes_per_crowd) * H * hd;
        for (int i = threadIdx.x; i < rd; i += kThreads) in_r[i] = rsrc[i];
        for (int i = threadIdx.x; i < H * hd; i += kThreads) in_h[i] = hsrc[i];
        __syncthreads();

        // X = [w_r(robot); w_h(humans)]
        mlp_rows(g.w_r, in_r, rd, 1, b0, b1, L.buf_ld, X, xd);
        mlp_rows(g.w_h, in_h, hd, H, b0, b1, L.buf_ld, X + xd, xd);

        // Which rows of the LAST layer are consumed?  Only node 0 when nothing but the value head reads it.
        const bool all_rows_last = (a.H_out != nullptr) || a.has_mhead || g.layerwise_graph;
        bool wrote_A = false;
        if (!g.layerwise_graph) {
            similarity(g, X, N, T, S, L.s_ld, U, V, L.uv_ld, N);
            if (a.A_out) {
                for (int idx = threadIdx.x; idx < N * N; idx += kThreads)
                    a.A_out[(size_t)s * N * N + idx] = S[(idx / N) * L.s_ld + (idx % N)];
                wrote_A = true;
            }
        }
        const float* cur = X;
        float* nxt = Hc;
        for (int l = 0; l < g.num_layer; ++l) {
            if (g.layerwise_graph) {
                similarity(g, cur, N, T, S, L.s_ld, U, V, L.uv_ld, N);
                if (a.A_out && !wrote_A) {
                    for (int idx = threadIdx.x; idx < N * N; idx += kThreads)
                        a.A_out[(size_t)s * N * N + idx] = S[(idx / N) * L.s_ld + (idx % N)];
                    wrote_A = true;
                }
            }
            const int rows = (l == g.num_layer - 1 && !all_rows_last) ? 1 : N;
            // T = A * cur
            for (int idx = threadIdx.x; idx < rows * xd; idx += kThreads) {
                const int i = idx / xd, c = idx - i * xd;
                float acc = 0.f;
                for (int j = 0; j < N; ++j) acc = fmaf(S[i * L.s_ld + j], cur[j * xd + c], acc);
                T[idx] = acc;
            }
            __syncthreads();
            // nxt = relu(T * W_l) (+ cur)
            const float* __restrict__ W = g.Ws[l];
            for (int idx = threadIdx.x; idx < rows * xd; idx += kThreads) {
                const int i = idx / xd, c = idx - i * xd;
                float acc = 0.f;
                for (int k = 0; k < xd; ++k) acc = fmaf(T[i * xd + k], W[k * xd + c], acc);
                acc = fmaxf(acc, 0.f);
                if (g.skip_connection) acc += cur[idx];
                nxt[idx] = acc;
            }
            __syncthreads();
            cur = nxt;
            nxt = (nxt == Hc) ? Hn : Hc;
        }
        if (a.A_out && !wrote_A) {      // num_layer == 0 with layerwise: still report an adjacency
            similarity(g, X, N, T, S, L.s_ld, U, V, L.uv_ld, N);
            for (int idx = threadIdx.x; idx < N * N; idx += kThreads)
                a.A_out[(size_t)s * N * N + idx] = S[(idx / N) * L.s_ld + (idx % N)];
        }
        if (a.H_out)
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) a.H_out[(size_t)s * N * xd + idx] = cur[idx];
        if (a.has_vhead) mlp_rows(a.vhead, cur, xd, 1, b0, b1, L.buf_ld, a.value_out + s, 1);
        if (a.has_mhead) {
            const int od = a.mhead.dims[a.mhead.n_layers];
            mlp_rows(a.mhead, cur + xd, xd, H, b0, b1, L.buf_ld, a.humans_next + (size_t)s * H * od, od);
        }
        __syncthreads();
    }
}

__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int r = idx / cols, c = idx - r * cols;
    dst[c * rows + r] = src[idx];
}

constexpr int kTransposeBatch = 32;
struct TransposeBatch {
    RglTransposeJob job[kTransposeBatch];
    int first_block[kTransposeBatch + 1];      // job j owns workgroups [first_block[j], first_block[j + 1])
    int n;
};

__global__ void transpose_many_kernel(const TransposeBatch b) {
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.first_block[j + 1]) ++j;
    const RglTransposeJob& t = b.job[j];
    const int idx = ((int)blockIdx.x - b.first_block[j]) * blockDim.x + threadIdx.x;
    if (idx >= t.rows * t.cols) return;
    const int r = idx / t.cols, c = idx - r * t.cols;
    t.dst[c * t.rows + r] = t.src[idx];
}

}  // namespace

namespace rgl {

// argument checks shared by every forward path
int validate_forward_call(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head, const float* robot,
                          const float* humans, int n_scenes, int scenes_per_crowd, int H, const float* value_out,
                          const float* humans_next) {
    if (!graph || !robot || !humans) return RGL_ERR_NULL;
    if (n_scenes < 0 || scenes_per_crowd < 1) return RGL_ERR_BAD_SHAPE;
    int rc = validate_graph(*graph, H);
    if (rc) return rc;
    if (value_head && value_head->n_layers > 0) {
        rc = validate_mlp(*value_head, graph->x_dim, 1);
        if (rc) return rc;
        if (!value_out) return RGL_ERR_NULL;
    }
    if (motion_head && motion_head->n_layers > 0) {
        rc = validate_mlp(*motion_head, graph->x_dim, 0);
        if (rc) return rc;
        if (!humans_next) return RGL_ERR_NULL;
    }
    return RGL_OK;
}

int launch_generic_forward(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head,
                           const float* robot, const float* humans, int n_scenes, int scenes_per_crowd, int H,
                           float* H_out, float* A_out, float* value_out, float* humans_next, hipStream_t stream) {
    int rc = validate_forward_call(graph, value_head, motion_head, robot, humans, n_scenes, scenes_per_crowd, H, value_out,
                                   humans_next);
    if (rc) return rc;
    if (n_scenes == 0) return RGL_OK;

    ForwardArgs a;
    a.g = *graph;
    a.has_vhead = (value_head && value_head->n_layers > 0) ? 1 : 0;
    a.has_mhead = (motion_head && motion_head->n_layers > 0) ? 1 : 0;
    if (a.has_vhead) a.vhead = *value_head; else a.vhead = RglMlp{};
    if (a.has_mhead) a.mhead = *motion_head; else a.mhead = RglMlp{};
    a.robot = robot;
    a.humans = humans;
    a.n_scenes = n_scenes;
    a.scenes_per_crowd = scenes_per_crowd;
    a.H = H;
    a.H_out = H_out;
    a.A_out = A_out;
    a.value_out = value_out;
    a.humans_next = humans_next;

    const int N = H + 1, xd = graph->x_dim;
    int wmax = mlp_max_hidden(graph->w_r);
    wmax = wmax > mlp_max_hidden(graph->w_h) ? wmax : mlp_max_hidden(graph->w_h);
    if (a.has_mhead) wmax = wmax > mlp_max_hidden(a.mhead) ? wmax : mlp_max_hidden(a.mhead);
    int vmax = a.has_vhead ? mlp_max_hidden(a.vhead) : 0;   // single-row MLP: needs vmax floats, not N*vmax
    SceneLds L;
    int off = 0;
    auto take = [&](int n) { int o = off; off += (n + 3) & ~3; return o; };
    L.in_r = take(graph->w_r.dims[0]);
    L.in_h = take(H * graph->w_h.dims[0]);
    L.buf_ld = wmax > 0 ? wmax : 1;
    int buf_floats = N * L.buf_ld;
    if (vmax > buf_floats) buf_floats = vmax;
    if (vmax > L.buf_ld) { /* single row may be wider than buf_ld; rows=1 so stride is irrelevant */ }
    // The MLP ping-pong rows and the N x N similarity matrix are never live together (embeddings come before the first
    // adjacency, the heads after the last layer): they share one region, which is what lets N = 128 fit a CU's LDS.
    L.s_ld = N + 1;
    {
        const int s_floats = N * L.s_ld;
        const int region = 2 * ((buf_floats + 3) & ~3) > s_floats ? 2 * ((buf_floats + 3) & ~3) : s_floats;
        L.buf0 = take(region);
        L.buf1 = L.buf0 + ((buf_floats + 3) & ~3);
        L.S = L.buf0;
    }
    L.X = take(N * xd);
    L.Hc = take(N * xd);
    L.Hn = take(N * xd);
    L.T = take(N * xd > N ? N * xd : N);
    if (graph->similarity == RGL_SIM_CONCATENATION) {
        L.uv_ld = graph->w_a_mlp.dims[1] + 1;
        L.U = take(N * L.uv_ld);
        L.V = take(N * L.uv_ld);
    } else {
        L.uv_ld = 1;
        L.U = L.V = 0;
    }
    L.total = off;
    const size_t lds_bytes = (size_t)L.total * sizeof(float);
    if (lds_bytes > (size_t)kLdsBytesPerCu) return RGL_ERR_LDS;
    a.L = L;

    if (lds_bytes > 64 * 1024) {
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(rgl_scene_forward_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    const int grid = n_scenes < 256 * 64 ? n_scenes : 256 * 64;
    hipLaunchKernelGGL(rgl_scene_forward_kernel, dim3(grid), dim3(kThreads), lds_bytes, stream, a);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

}  // namespace rgl

extern "C" size_t rgl_graph_forward_workspace_bytes(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head,
                                                    int n_scenes, int scenes_per_crowd, int H) {
    if (!graph || n_scenes < 1 || scenes_per_crowd < 1 || H < 1) return 0;
    const size_t scene = rgl::scene_forward_workspace_bytes(graph, value_head, motion_head, n_scenes, scenes_per_crowd, H);
    // outside the shipped shapes: the tile kernels (other embedding MLPs, x_dim = 64); RGL_TILES_FORWARD=2 (tests) runs them first
    // for the shipped shapes too
    const char* e = getenv("RGL_TILES_FORWARD");
    if ((scene && !(e && e[0] == '2')) || rgl::validate_graph(*graph, H)) return scene;
    const size_t tiles = rgl::tiles_forward_workspace_bytes(graph, value_head, motion_head, n_scenes, scenes_per_crowd, H, 0);
    return scene > tiles ? scene : tiles;
}

extern "C" int rgl_graph_forward_f32(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head,
                                     const float* robot, const float* humans, int n_scenes, int scenes_per_crowd,
                                     int H, float* H_out, float* A_out, float* value_out, float* humans_next,
                                     void* workspace, size_t workspace_bytes, rgl_stream_t stream) {
    // values / next humans only, and a workspace: the one-wave-per-scene MFMA kernel (rgl_scene.hip) where it covers the model
    if (workspace && !H_out && !A_out) {
        int rc = rgl::validate_forward_call(graph, value_head, motion_head, robot, humans, n_scenes, scenes_per_crowd, H, value_out,
                                            humans_next);
        if (rc) return rc;
        if (n_scenes == 0) return RGL_OK;
        {   // RGL_TILES_FORWARD=2 (tests): the tile kernels first, also for the shipped shapes
            const char* e = getenv("RGL_TILES_FORWARD");
            if (e && e[0] == '2') {
                rc = rgl::launch_tiles_forward(graph, value_head, motion_head, robot, humans, n_scenes, scenes_per_crowd, H, nullptr,
                                               value_out, humans_next, workspace, workspace_bytes, (hipStream_t)stream);
                if (rc != 1) return rc;
            }
        }
        rc = rgl::launch_scene_forward(graph, value_head, motion_head, robot, humans, n_scenes, scenes_per_crowd, H, value_out,
                                       humans_next, workspace, workspace_bytes, (hipStream_t)stream);
        if (rc != 1) return rc;
        rc = rgl::launch_tiles_forward(graph, value_head, motion_head, robot, humans, n_scenes, scenes_per_crowd, H, nullptr, value_out,
                                       humans_next, workspace, workspace_bytes, (hipStream_t)stream);
        if (rc != 1) return rc;
        // RGL_REQUIRE_MFMA_FORWARD=1 (tests): refuse instead of running the general VALU kernel
        const char* e = getenv("RGL_REQUIRE_MFMA_FORWARD");
        if (e && e[0] == '1') return RGL_ERR_BAD_MODE;
    }
    return rgl::launch_generic_forward(graph, value_head, motion_head, robot, humans, n_scenes, scenes_per_crowd, H,
                                       H_out, A_out, value_out, humans_next, (hipStream_t)stream);
}

extern "C" int rgl_transpose_f32(const float* src, float* dst, int rows, int cols, rgl_stream_t stream) {
    if (!src || !dst) return RGL_ERR_NULL;
    if (rows < 1 || cols < 1) return RGL_ERR_BAD_SHAPE;
    const int n = rows * cols;
    hipLaunchKernelGGL(transpose_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

extern "C" int rgl_transpose_many_f32(const RglTransposeJob* jobs, int n_jobs, rgl_stream_t stream) {
    if (n_jobs < 0) return RGL_ERR_BAD_SHAPE;
    if (n_jobs == 0) return RGL_OK;
    if (!jobs) return RGL_ERR_NULL;
    for (int j = 0; j < n_jobs; ++j) {
        if (!jobs[j].src || !jobs[j].dst) return RGL_ERR_NULL;
        if (jobs[j].rows < 1 || jobs[j].cols < 1 || (long long)jobs[j].rows * jobs[j].cols > (1ll << 30)) return RGL_ERR_BAD_SHAPE;
    }
    for (int lo = 0; lo < n_jobs; lo += kTransposeBatch) {
        TransposeBatch b;
        b.n = n_jobs - lo < kTransposeBatch ? n_jobs - lo : kTransposeBatch;
        int blocks = 0;
        for (int j = 0; j < b.n; ++j) {
            b.job[j] = jobs[lo + j];
            b.first_block[j] = blocks;
            blocks += (jobs[lo + j].rows * jobs[lo + j].cols + 255) / 256;
        }
        b.first_block[b.n] = blocks;
        hipLaunchKernelGGL(transpose_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
        RGL_LAUNCH_CHECK();
    }
    return RGL_OK;
}

extern "C" int rgl_abi_version(void) { return RGL_ABI_VERSION; }
extern "C" const char* rgl_build_target(void) { return "gfx950"; }
