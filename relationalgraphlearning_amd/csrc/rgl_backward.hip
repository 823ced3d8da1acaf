// rgl_backward.hip -- gradients of the relational-graph forward with respect to every parameter
// (training path: crowd_nav/utils/trainer.py:110-161,199-250 call forward+backward on batches of ~100
// scenes).  One 256-thread workgroup per scene recomputes the forward with every activation kept in LDS and
// back-propagates through  heads -> GCN layers -> row softmax -> similarity -> embedding MLPs ; each scene
// writes its own gradient slab (every element exactly once, by one thread, as a fixed-order sum), and
// reduce_slabs_kernel adds the slabs in scene order, so the result is deterministic.
//
// Supported structure: similarity embedded_gaussian, gaussian, squared, equal_attention or diagonal; one adjacency for all layers
// (layerwise_graph = 0), any depth / skip / MLP shapes within the ABI limits.  Anything else returns
// RGL_ERR_BAD_MODE (the Python side raises; it never falls back to another device).
//
// Forward being differentiated (reference: crowd_nav/policy/graph_model.py:99-130, value_estimator.py:11-20,
// state_predictor.py:28-36, gcn.py:95-128):
//   X = [w_r(robot); w_h(humans)]      G = X Wa (or X)      S = G X^T      A = softmax_rows(S)
//   H_0 = X ;  T_l = A H_l ;  R_l = relu(T_l W_l) ;  H_{l+1} = R_l (+ H_l)
//   value = value_head(H_L[0]) ;  humans_next = motion_head(H_L)[1:]
#include "rgl_common.h"

namespace {

constexpr int kThreads = 256;

struct MlpOffsets {                 // float offsets of W_l / b_l inside a gradient slab
    int w[RGL_MAX_MLP_LAYERS], b[RGL_MAX_MLP_LAYERS];
};

struct BackwardArgs {
    RglGraph g;
    RglMlp vhead, mhead;
    int has_vhead, has_mhead, detach_graph;
    const float* robot;             // [S][rd]
    const float* humans;            // [S][H][hd]
    int n_scenes, H;
    const float* d_value;           // [S] or null
    const float* d_humans_next;     // [S][H][od] or null
    const float* d_H;               // [S][N][xd] or null
    float* slabs;                   // [S][n_params]
    int n_params;
    MlpOffsets o_wr, o_wh, o_vh, o_mh;
    int o_wa, o_ws[RGL_MAX_GCN_LAYERS];
    // LDS layout (float offsets)
    int l_ar, ar_ld;                // robot MLP activations: [1][ar_ld]  (all layers' inputs/outputs back to back)
    int l_ah, ah_ld;                // human MLP activations: [H][ah_ld]
    int l_av, av_ld;                // value head activations: [1][av_ld]
    int l_am, am_ld;                // motion head activations: [H][am_ld]
    int l_G, l_A, a_ld, l_H, l_T, l_R;   // G[N][xd], A[N][a_ld], H[L+1][N][xd], T[L][N][xd], R[L][N][xd]
    int l_dH, l_dH2, l_dT, l_dA, l_dG, l_d0, l_d1, d_ld;   // deltas
    int total;
};

__device__ __forceinline__ int act_offset(const RglMlp& m, int layer) {   // offset of the input of `layer` (== output of layer-1)
    int o = 0;
    for (int l = 0; l < layer; ++l) o += m.dims[l];
    return o;
}

// forward of an MLP over `rows` rows with every layer's input/output stored: acts[r][act_offset(l)] ...
__device__ void mlp_forward_saved(const RglMlp& m, float* acts, int ld, int rows) {
    int in_off = 0;
    for (int l = 0; l < m.n_layers; ++l) {
        const int in = m.dims[l], out = m.dims[l + 1];
        const int out_off = in_off + in;
        const bool relu = (l != m.n_layers - 1) || m.last_relu;
        const float* __restrict__ W = m.weight[l];
        const float* __restrict__ b = m.bias[l];
        for (int idx = threadIdx.x; idx < rows * out; idx += kThreads) {
            const int r = idx / out, j = idx - r * out;
            float acc = b[j];
            const float* a = acts + r * ld + in_off;
            for (int k = 0; k < in; ++k) acc = fmaf(a[k], W[k * out + j], acc);
            acts[r * ld + out_off + j] = relu ? fmaxf(acc, 0.f) : acc;
        }
        __syncthreads();
        in_off = out_off;
    }
}

// backward of the same MLP.  d0 holds dL/d(output) [rows][d_ld] on entry; on exit the buffer returned holds
// dL/d(input) [rows][d_ld].  Weight/bias gradients go to the slab (k-major [in][out], like the forward weights).
__device__ float* mlp_backward(const RglMlp& m, const MlpOffsets& off, const float* acts, int ld, int rows, float* d0,
                               float* d1, int d_ld, float* slab, bool need_input_grad) {
    float* cur = d0;
    float* nxt = d1;
    for (int l = m.n_layers - 1; l >= 0; --l) {
        const int in = m.dims[l], out = m.dims[l + 1];
        const int in_off = act_offset(m, l), out_off = in_off + in;
        const bool relu = (l != m.n_layers - 1) || m.last_relu;
        if (relu) {
            for (int idx = threadIdx.x; idx < rows * out; idx += kThreads) {
                const int r = idx / out, j = idx - r * out;
                if (!(acts[r * ld + out_off + j] > 0.f)) cur[r * d_ld + j] = 0.f;
            }
            __syncthreads();
        }
        float* gW = slab + off.w[l];
        float* gb = slab + off.b[l];
        for (int idx = threadIdx.x; idx < in * out; idx += kThreads) {
            const int k = idx / out, j = idx - k * out;
            float acc = 0.f;
            for (int r = 0; r < rows; ++r) acc = fmaf(acts[r * ld + in_off + k], cur[r * d_ld + j], acc);
            gW[idx] = acc;
        }
        for (int j = threadIdx.x; j < out; j += kThreads) {
            float acc = 0.f;
            for (int r = 0; r < rows; ++r) acc += cur[r * d_ld + j];
            gb[j] = acc;
        }
        if (l > 0 || need_input_grad) {
            const float* __restrict__ W = m.weight[l];
            for (int idx = threadIdx.x; idx < rows * in; idx += kThreads) {
                const int r = idx / in, k = idx - r * in;
                float acc = 0.f;
                for (int j = 0; j < out; ++j) acc = fmaf(cur[r * d_ld + j], W[k * out + j], acc);
                nxt[r * d_ld + k] = acc;
            }
        }
        __syncthreads();
        float* t = cur;
        cur = nxt;
        nxt = t;
    }
    return cur;
}

__device__ void zero_mlp_grads(const RglMlp& m, const MlpOffsets& off, float* slab) {
    for (int l = 0; l < m.n_layers; ++l) {
        for (int i = threadIdx.x; i < m.dims[l] * m.dims[l + 1]; i += kThreads) slab[off.w[l] + i] = 0.f;
        for (int i = threadIdx.x; i < m.dims[l + 1]; i += kThreads) slab[off.b[l] + i] = 0.f;
    }
}

__global__ __launch_bounds__(kThreads) void rgl_scene_backward_kernel(const BackwardArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const RglGraph& g = a.g;
    const int H = a.H, N = H + 1, xd = g.x_dim, L = g.num_layer;
    const int rd = g.w_r.dims[0], hd = g.w_h.dims[0];
    float* AR = lds + a.l_ar;
    float* AH = lds + a.l_ah;
    float* AV = lds + a.l_av;
    float* AM = lds + a.l_am;
    float* G = lds + a.l_G;
    float* A = lds + a.l_A;
    float* Hs = lds + a.l_H;      // [L+1][N][xd]
    float* T = lds + a.l_T;       // [L][N][xd]
    float* R = lds + a.l_R;       // [L][N][xd]
    float* dH = lds + a.l_dH;
    float* dH2 = lds + a.l_dH2;
    float* dT = lds + a.l_dT;
    float* dA = lds + a.l_dA;
    float* dG = lds + a.l_dG;
    float* d0 = lds + a.l_d0;
    float* d1 = lds + a.l_d1;
    const int a_ld = a.a_ld, d_ld = a.d_ld;
    const bool embedded = g.similarity == RGL_SIM_EMBEDDED_GAUSSIAN;
    const int sim = g.similarity;

    for (int s = blockIdx.x; s < a.n_scenes; s += gridDim.x) {
        float* slab = a.slabs + (size_t)s * a.n_params;
        // ------------------------------ forward, everything kept ------------------------------------------
        for (int i = threadIdx.x; i < rd; i += kThreads) AR[i] = a.robot[(size_t)s * rd + i];
        for (int i = threadIdx.x; i < H * hd; i += kThreads) AH[(i / hd) * a.ah_ld + (i % hd)] = a.humans[(size_t)s * H * hd + i];
        __syncthreads();
        mlp_forward_saved(g.w_r, AR, a.ar_ld, 1);
        mlp_forward_saved(g.w_h, AH, a.ah_ld, H);
        const int xr_off = act_offset(g.w_r, g.w_r.n_layers), xh_off = act_offset(g.w_h, g.w_h.n_layers);
        float* X = Hs;                                          // H_0
        for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
            const int i = idx / xd, f = idx - i * xd;
            X[idx] = i == 0 ? AR[xr_off + f] : AH[(i - 1) * a.ah_ld + xh_off + f];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
            const int i = idx / xd, c = idx - i * xd;
            float acc;
            if (embedded) {
                acc = 0.f;
                for (int k = 0; k < xd; ++k) acc = fmaf(X[i * xd + k], g.w_a[k * xd + c], acc);
            } else acc = X[idx];
            G[idx] = acc;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < N * N; idx += kThreads) {
            const int i = idx / N, j = idx - i * N;
            float acc = 0.f;
            for (int k = 0; k < xd; ++k) acc = fmaf(G[i * xd + k], X[j * xd + k], acc);
            A[i * a_ld + j] = acc;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += kThreads) {
            float* r = A + i * a_ld;
            if (sim == RGL_SIM_SQUARED) {                       // S^2 / sum_j S^2   (graph_model.py:86-89)
                float sum = 0.f;
                for (int j = 0; j < N; ++j) { r[j] = r[j] * r[j]; sum += r[j]; }
                for (int j = 0; j < N; ++j) r[j] = r[j] / sum;
            } else if (sim == RGL_SIM_EQUAL_ATTENTION) {
                for (int j = 0; j < N; ++j) r[j] = 1.f / (float)N;
            } else if (sim == RGL_SIM_DIAGONAL) {
                for (int j = 0; j < N; ++j) r[j] = i == j ? 1.f : 0.f;
            } else {                                            // row softmax
                float mx = r[0];
                for (int j = 1; j < N; ++j) mx = fmaxf(mx, r[j]);
                float sum = 0.f;
                for (int j = 0; j < N; ++j) { r[j] = expf(r[j] - mx); sum += r[j]; }
                for (int j = 0; j < N; ++j) r[j] = r[j] / sum;
            }
        }
        __syncthreads();
        for (int l = 0; l < L; ++l) {
            const float* Hl = Hs + l * N * xd;
            float* Tl = T + l * N * xd;
            float* Rl = R + l * N * xd;
            float* Hn = Hs + (l + 1) * N * xd;
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
                const int i = idx / xd, c = idx - i * xd;
                float acc = 0.f;
                for (int j = 0; j < N; ++j) acc = fmaf(A[i * a_ld + j], Hl[j * xd + c], acc);
                Tl[idx] = acc;
            }
            __syncthreads();
            const float* __restrict__ W = g.Ws[l];
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
                const int i = idx / xd, c = idx - i * xd;
                float acc = 0.f;
                for (int k = 0; k < xd; ++k) acc = fmaf(Tl[i * xd + k], W[k * xd + c], acc);
                acc = fmaxf(acc, 0.f);
                Rl[idx] = acc;
                Hn[idx] = g.skip_connection ? acc + Hl[idx] : acc;
            }
            __syncthreads();
        }
        const float* HL = Hs + L * N * xd;
        if (a.has_vhead) {
            for (int f = threadIdx.x; f < xd; f += kThreads) AV[f] = HL[f];
            __syncthreads();
            mlp_forward_saved(a.vhead, AV, a.av_ld, 1);
        }
        if (a.has_mhead) {
            for (int idx = threadIdx.x; idx < H * xd; idx += kThreads)
                AM[(idx / xd) * a.am_ld + (idx % xd)] = HL[(idx / xd + 1) * xd + (idx % xd)];
            __syncthreads();
            mlp_forward_saved(a.mhead, AM, a.am_ld, H);
        }

        // ------------------------------ backward ------------------------------------------------------------
        for (int idx = threadIdx.x; idx < N * xd; idx += kThreads)
            dH[idx] = a.d_H ? a.d_H[(size_t)s * N * xd + idx] : 0.f;
        __syncthreads();
        if (a.has_vhead) {
            if (threadIdx.x == 0) d0[0] = a.d_value ? a.d_value[s] : 0.f;
            __syncthreads();
            const float* din = mlp_backward(a.vhead, a.o_vh, AV, a.av_ld, 1, d0, d1, d_ld, slab, true);
            for (int f = threadIdx.x; f < xd; f += kThreads) dH[f] += din[f];
            __syncthreads();
        }
        if (a.has_mhead) {
            const int od = a.mhead.dims[a.mhead.n_layers];
            for (int idx = threadIdx.x; idx < H * od; idx += kThreads)
                d0[(idx / od) * d_ld + (idx % od)] = a.d_humans_next ? a.d_humans_next[(size_t)s * H * od + idx] : 0.f;
            __syncthreads();
            const float* din = mlp_backward(a.mhead, a.o_mh, AM, a.am_ld, H, d0, d1, d_ld, slab, true);
            for (int idx = threadIdx.x; idx < H * xd; idx += kThreads)
                dH[(idx / xd + 1) * xd + (idx % xd)] += din[(idx / xd) * d_ld + (idx % xd)];
            __syncthreads();
        }
        if (a.detach_graph) {
            // StatePredictor(..., detach=True): the embedding is a constant; only the head learns
            zero_mlp_grads(g.w_r, a.o_wr, slab);
            zero_mlp_grads(g.w_h, a.o_wh, slab);
            if (embedded) for (int i = threadIdx.x; i < xd * xd; i += kThreads) slab[a.o_wa + i] = 0.f;
            for (int l = 0; l < L; ++l) for (int i = threadIdx.x; i < xd * xd; i += kThreads) slab[a.o_ws[l] + i] = 0.f;
            __syncthreads();
            continue;
        }
        for (int idx = threadIdx.x; idx < N * a_ld; idx += kThreads) dA[idx] = 0.f;
        __syncthreads();
        float* dcur = dH;
        float* dnxt = dH2;
        for (int l = L - 1; l >= 0; --l) {
            const float* Hl = Hs + l * N * xd;
            const float* Tl = T + l * N * xd;
            const float* Rl = R + l * N * xd;
            const float* __restrict__ W = g.Ws[l];
            // dZ = dH_{l+1} o (R_l > 0)   (kept in dG as scratch)
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) dG[idx] = Rl[idx] > 0.f ? dcur[idx] : 0.f;
            __syncthreads();
            float* gW = slab + a.o_ws[l];
            for (int idx = threadIdx.x; idx < xd * xd; idx += kThreads) {
                const int k = idx / xd, c = idx - k * xd;
                float acc = 0.f;
                for (int i = 0; i < N; ++i) acc = fmaf(Tl[i * xd + k], dG[i * xd + c], acc);
                gW[idx] = acc;
            }
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {      // dT = dZ W^T
                const int i = idx / xd, k = idx - i * xd;
                float acc = 0.f;
                for (int c = 0; c < xd; ++c) acc = fmaf(dG[i * xd + c], W[k * xd + c], acc);
                dT[idx] = acc;
            }
            __syncthreads();
            for (int idx = threadIdx.x; idx < N * N; idx += kThreads) {       // dA += dT H_l^T
                const int i = idx / N, j = idx - i * N;
                float acc = dA[i * a_ld + j];
                for (int k = 0; k < xd; ++k) acc = fmaf(dT[i * xd + k], Hl[j * xd + k], acc);
                dA[i * a_ld + j] = acc;
            }
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {      // dH_l = A^T dT (+ dH_{l+1})
                const int j = idx / xd, k = idx - j * xd;
                float acc = g.skip_connection ? dcur[idx] : 0.f;
                for (int i = 0; i < N; ++i) acc = fmaf(A[i * a_ld + j], dT[i * xd + k], acc);
                dnxt[idx] = acc;
            }
            __syncthreads();
            float* t = dcur;
            dcur = dnxt;
            dnxt = t;
        }
        // dcur = dL/dX from the layers.  softmax: dS_ij = A_ij (dA_ij - sum_k dA_ik A_ik)   (in place in dA)
        //                                  squared: dS_ij = (2 S_ij / sum_k S_ik^2) (dA_ij - sum_k dA_ik A_ik), S recomputed
        //                                  equal_attention / diagonal: A is constant, dS = 0
        for (int i = threadIdx.x; i < N; i += kThreads) {
            float dot = 0.f;
            for (int j = 0; j < N; ++j) dot = fmaf(dA[i * a_ld + j], A[i * a_ld + j], dot);
            if (sim == RGL_SIM_SQUARED) {
                float z = 0.f;
                for (int j = 0; j < N; ++j) {
                    float sij = 0.f;
                    for (int k = 0; k < xd; ++k) sij = fmaf(G[i * xd + k], X[j * xd + k], sij);
                    z = fmaf(sij, sij, z);
                }
                for (int j = 0; j < N; ++j) {
                    float sij = 0.f;
                    for (int k = 0; k < xd; ++k) sij = fmaf(G[i * xd + k], X[j * xd + k], sij);
                    dA[i * a_ld + j] = 2.f * sij / z * (dA[i * a_ld + j] - dot);
                }
            } else if (sim == RGL_SIM_EQUAL_ATTENTION || sim == RGL_SIM_DIAGONAL) {
                for (int j = 0; j < N; ++j) dA[i * a_ld + j] = 0.f;
            } else {
                for (int j = 0; j < N; ++j) dA[i * a_ld + j] = A[i * a_ld + j] * (dA[i * a_ld + j] - dot);
            }
        }
        __syncthreads();
        // (X = H_0 is still addressed through the pointer set up in the forward part)
        // S = G X^T :  dG = dS X ;  dX += dS^T G
        for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
            const int i = idx / xd, k = idx - i * xd;
            float acc = 0.f;
            for (int j = 0; j < N; ++j) acc = fmaf(dA[i * a_ld + j], X[j * xd + k], acc);
            dG[idx] = acc;
        }
        for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
            const int j = idx / xd, k = idx - j * xd;
            float acc = dcur[idx];
            for (int i = 0; i < N; ++i) acc = fmaf(dA[i * a_ld + j], G[i * xd + k], acc);
            dnxt[idx] = acc;
        }
        __syncthreads();
        if (embedded) {   // G = X Wa :  dWa = X^T dG ;  dX += dG Wa^T
            float* gWa = slab + a.o_wa;
            for (int idx = threadIdx.x; idx < xd * xd; idx += kThreads) {
                const int k = idx / xd, c = idx - k * xd;
                float acc = 0.f;
                for (int i = 0; i < N; ++i) acc = fmaf(X[i * xd + k], dG[i * xd + c], acc);
                gWa[idx] = acc;
            }
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
                const int i = idx / xd, k = idx - i * xd;
                float acc = dnxt[idx];
                for (int c = 0; c < xd; ++c) acc = fmaf(dG[i * xd + c], g.w_a[k * xd + c], acc);
                dnxt[idx] = acc;
            }
        } else {          // G = X
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) dnxt[idx] += dG[idx];
        }
        __syncthreads();
        // embeddings: row 0 -> w_r, rows 1..H -> w_h
        for (int f = threadIdx.x; f < xd; f += kThreads) d0[f] = dnxt[f];
        __syncthreads();
        mlp_backward(g.w_r, a.o_wr, AR, a.ar_ld, 1, d0, d1, d_ld, slab, false);
        for (int idx = threadIdx.x; idx < H * xd; idx += kThreads) d0[(idx / xd) * d_ld + (idx % xd)] = dnxt[(idx / xd + 1) * xd + (idx % xd)];
        __syncthreads();
        mlp_backward(g.w_h, a.o_wh, AH, a.ah_ld, H, d0, d1, d_ld, slab, false);
        __syncthreads();
    }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int n_scenes, int n_params, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_params) return;
    float acc = 0.f;
    for (int s = 0; s < n_scenes; ++s) acc += slabs[(size_t)s * n_params + k];     // fixed order: deterministic
    out[k] = acc;
}

int mlp_param_count(const RglMlp& m) {
    int n = 0;
    for (int l = 0; l < m.n_layers; ++l) n += m.dims[l] * m.dims[l + 1] + m.dims[l + 1];
    return n;
}

int mlp_act_width(const RglMlp& m) {
    int n = 0;
    for (int l = 0; l <= m.n_layers; ++l) n += m.dims[l];
    return n;
}

int assign_mlp(const RglMlp& m, MlpOffsets& o, int off) {
    for (int l = 0; l < m.n_layers; ++l) {
        o.w[l] = off; off += m.dims[l] * m.dims[l + 1];
        o.b[l] = off; off += m.dims[l + 1];
    }
    return off;
}

int plan_backward(const RglGraph* graph, const RglMlp* vh, const RglMlp* mh, int H, BackwardArgs& a) {
    if (!graph) return RGL_ERR_NULL;
    int rc = rgl::validate_graph(*graph, H);
    if (rc) return rc;
    if (graph->layerwise_graph) return RGL_ERR_BAD_MODE;
    if (graph->similarity != RGL_SIM_EMBEDDED_GAUSSIAN && graph->similarity != RGL_SIM_GAUSSIAN &&
        graph->similarity != RGL_SIM_SQUARED && graph->similarity != RGL_SIM_EQUAL_ATTENTION &&
        graph->similarity != RGL_SIM_DIAGONAL)
        return RGL_ERR_BAD_MODE;
    a.g = *graph;
    a.has_vhead = (vh && vh->n_layers > 0) ? 1 : 0;
    a.has_mhead = (mh && mh->n_layers > 0) ? 1 : 0;
    a.vhead = a.has_vhead ? *vh : RglMlp{};
    a.mhead = a.has_mhead ? *mh : RglMlp{};
    if (a.has_vhead && (rc = rgl::validate_mlp(a.vhead, graph->x_dim, 1))) return rc;
    if (a.has_mhead && (rc = rgl::validate_mlp(a.mhead, graph->x_dim, 0))) return rc;
    // slab order: w_r (W0,b0,W1,b1,..), w_h, w_a (embedded_gaussian only), Ws[0..L-1], value head, motion head
    int off = 0;
    off = assign_mlp(a.g.w_r, a.o_wr, off);
    off = assign_mlp(a.g.w_h, a.o_wh, off);
    a.o_wa = off;
    if (graph->similarity == RGL_SIM_EMBEDDED_GAUSSIAN) off += graph->x_dim * graph->x_dim;
    for (int l = 0; l < graph->num_layer; ++l) { a.o_ws[l] = off; off += graph->x_dim * graph->x_dim; }
    if (a.has_vhead) off = assign_mlp(a.vhead, a.o_vh, off);
    if (a.has_mhead) off = assign_mlp(a.mhead, a.o_mh, off);
    a.n_params = off;
    // LDS
    const int N = H + 1, xd = graph->x_dim, L = graph->num_layer;
    int lo = 0;
    auto take = [&](int n) { int o = lo; lo += (n + 3) & ~3; return o; };
    a.ar_ld = mlp_act_width(a.g.w_r); a.l_ar = take(a.ar_ld);
    a.ah_ld = mlp_act_width(a.g.w_h); a.l_ah = take(H * a.ah_ld);
    a.av_ld = a.has_vhead ? mlp_act_width(a.vhead) : 0; a.l_av = take(a.av_ld);
    a.am_ld = a.has_mhead ? mlp_act_width(a.mhead) : 0; a.l_am = take(H * a.am_ld);
    a.l_G = take(N * xd);
    a.a_ld = N + 1; a.l_A = take(N * a.a_ld);
    a.l_H = take((L + 1) * N * xd);
    a.l_T = take((L > 0 ? L : 1) * N * xd);
    a.l_R = take((L > 0 ? L : 1) * N * xd);
    a.l_dH = take(N * xd); a.l_dH2 = take(N * xd); a.l_dT = take(N * xd);
    a.l_dA = take(N * a.a_ld); a.l_dG = take(N * xd);
    int wmax = xd;
    auto widest = [&](const RglMlp& m) { for (int l = 0; l <= m.n_layers; ++l) wmax = m.dims[l] > wmax ? m.dims[l] : wmax; };
    widest(a.g.w_r); widest(a.g.w_h);
    if (a.has_vhead) widest(a.vhead);
    if (a.has_mhead) widest(a.mhead);
    a.d_ld = wmax;
    a.l_d0 = take((H > 1 ? H : 1) * wmax);
    a.l_d1 = take((H > 1 ? H : 1) * wmax);
    a.total = lo;
    if ((size_t)lo * sizeof(float) > (size_t)rgl::kLdsBytesPerCu) return RGL_ERR_LDS;
    return RGL_OK;
}

}  // namespace

extern "C" int rgl_graph_param_count(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head) {
    if (!graph) return RGL_ERR_NULL;
    BackwardArgs a;
    const int rc = plan_backward(graph, value_head, motion_head, 1, a);
    return rc ? rc : a.n_params;
}

extern "C" size_t rgl_graph_backward_workspace_bytes(const RglGraph* graph, const RglMlp* value_head,
                                                     const RglMlp* motion_head, int n_scenes) {
    const int n = rgl_graph_param_count(graph, value_head, motion_head);
    if (n <= 0 || n_scenes < 1) return 0;
    return (size_t)n_scenes * n * sizeof(float);
}

extern "C" int rgl_graph_backward_f32(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head,
                                      const float* robot, const float* humans, int n_scenes, int H, int detach_graph,
                                      const float* d_value, const float* d_humans_next, const float* d_H,
                                      float* grad_out, void* workspace, size_t workspace_bytes, rgl_stream_t stream) {
    if (!graph || !robot || !humans || !grad_out || !workspace) return RGL_ERR_NULL;
    if (n_scenes < 1) return RGL_ERR_BAD_SHAPE;
    BackwardArgs a;
    int rc = plan_backward(graph, value_head, motion_head, H, a);
    if (rc) return rc;
    if (workspace_bytes < (size_t)n_scenes * a.n_params * sizeof(float)) return RGL_ERR_WORKSPACE;
    a.detach_graph = detach_graph ? 1 : 0;
    a.robot = robot; a.humans = humans; a.n_scenes = n_scenes; a.H = H;
    a.d_value = d_value; a.d_humans_next = d_humans_next; a.d_H = d_H;
    a.slabs = (float*)workspace;
    const size_t lds_bytes = (size_t)a.total * sizeof(float);
    if (lds_bytes > 64 * 1024)
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(rgl_scene_backward_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipStream_t st = (hipStream_t)stream;
    const int grid = n_scenes < 4096 ? n_scenes : 4096;
    hipLaunchKernelGGL(rgl_scene_backward_kernel, dim3(grid), dim3(kThreads), lds_bytes, st, a);
    RGL_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((a.n_params + 255) / 256), dim3(256), 0, st, (const float*)workspace, n_scenes,
                       a.n_params, grad_out);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}
