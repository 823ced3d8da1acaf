// rgl_backward.hip -- gradients of the relational-graph forward with respect to every parameter
// (training path: crowd_nav/utils/trainer.py:110-161,199-250 call forward+backward on batches of ~100
// scenes).  One 256-thread workgroup per scene recomputes the forward with every activation kept in LDS and
// back-propagates through  heads -> GCN layers -> row softmax -> similarity -> embedding MLPs ; each scene
// writes its own gradient slab (every element exactly once, by one thread, as a fixed-order sum), and
// reduce_slabs_kernel adds the slabs in scene order, so the result is deterministic.
//
// Supported structure: all eight similarity functions of compute_similarity_matrix (graph_model.py:63-97), one adjacency for
// all layers or one per layer (layerwise_graph), any depth / skip / MLP shapes within the ABI limits; a configuration whose
// activations do not fit the 160 KB LDS of a CU returns RGL_ERR_LDS (the Python side raises; it never falls back to another
// device).  The similarity block is a pair of device functions (sim_forward / sim_backward) applied to X once, or to every
// H_l when the graph is layerwise; its internals (S, norms, pair-MLP halves) are recomputed from the saved H_l in the
// backward pass instead of being stored per layer.
//
// Forward being differentiated (reference: crowd_nav/policy/graph_model.py:99-130, value_estimator.py:11-20,
// state_predictor.py:28-36, gcn.py:95-128):
//   X = [w_r(robot); w_h(humans)]      A = sim(X)   [layerwise: A_l = sim(H_l)]
//   sim: softmax_rows(X Wa X^T) | softmax_rows(X X^T) | C = S / (m m^T), m_i = |S_i,:|_2 | softmax_rows(C) |
//        relu(w2 . relu(W1a x_i + W1b x_j + b1) + b2) | S^2 / rowsum(S^2) | 1/N | I
//   H_0 = X ;  T_l = A H_l ;  R_l = relu(T_l W_l) ;  H_{l+1} = R_l (+ H_l)
//   value = value_head(H_L[0]) ;  humans_next = motion_head(H_L)[1:]
#include "rgl_common.h"

namespace {

constexpr int kThreads = 1024;

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
    MlpOffsets o_wr, o_wh, o_vh, o_mh, o_wam;      // o_wam: the pair MLP of `concatenation`
    int o_wa, o_ws[RGL_MAX_GCN_LAYERS];
    // LDS layout (float offsets)
    int l_ar, ar_ld;                // robot MLP activations: [1][ar_ld]  (all layers' inputs/outputs back to back)
    int l_ah, ah_ld;                // human MLP activations: [H][ah_ld]
    int l_av, av_ld;                // value head activations: [1][av_ld]
    int l_am, am_ld;                // motion head activations: [H][am_ld]
    int l_G, l_A, a_ld, l_H, l_T, l_R;   // G[N][xd], A[nA][N][a_ld] (nA = L if layerwise else 1), H[L+1][N][xd], T[L][N][xd], R[L][N][xd]
    int l_S, l_C, l_m;                   // cosine*: S[N][a_ld], C[N][a_ld], m[N]          (scratch shared with G / the pair halves)
    int l_P, l_Q, l_dP, l_dQ, hid;       // concatenation: P = X W1a, Q = X W1b, their deltas, [N][hid]
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
#pragma unroll 16
            for (int k = 0; k < in; ++k) acc = fmaf(a[k], W[k * out + j], acc);
            acts[r * ld + out_off + j] = relu ? fmaxf(acc, 0.f) : acc;
        }
        __syncthreads();
        in_off = out_off;
    }
}

// backward of the same MLP.  d0 holds dL/d(output) [rows][d_ld] on entry; on exit the buffer returned holds
// dL/d(input) [rows][d_ld].  Weight/bias gradients go to the slab, weights in torch's Linear layout [out][in].
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
        for (int idx = threadIdx.x; idx < in * out; idx += kThreads) {     // torch layout [out][in]: autograd hands out contiguous views
            const int j = idx / in, k = idx - j * in;
            float acc = 0.f;
#pragma unroll 16
            for (int r = 0; r < rows; ++r) acc = fmaf(acts[r * ld + in_off + k], cur[r * d_ld + j], acc);
            gW[idx] = acc;
        }
        for (int j = threadIdx.x; j < out; j += kThreads) {
            float acc = 0.f;
#pragma unroll 16
            for (int r = 0; r < rows; ++r) acc += cur[r * d_ld + j];
            gb[j] = acc;
        }
        if (l > 0 || need_input_grad) {
            const float* __restrict__ W = m.weight[l];
            for (int idx = threadIdx.x; idx < rows * in; idx += kThreads) {
                const int r = idx / in, k = idx - r * in;
                float acc = 0.f;
#pragma unroll 16
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

// ------------------------------------------------------------------------------------------------
// similarity block:  A = sim(X)  and its backward.  All eight functions of graph_model.py:63-97.
// ------------------------------------------------------------------------------------------------
// scratch (LDS, per workgroup): G [N][xd] (embedded_gaussian) | S, C [N][a_ld] + m [N] (cosine, cosine_softmax) | P, Q [N][hid]
// (concatenation).  On return A [N][a_ld] holds the adjacency; the scratch holds what sim_backward needs for THIS X.
__device__ void sim_forward(const BackwardArgs& a, float* lds, const float* X, float* A) {
    const RglGraph& g = a.g;
    const int N = a.H + 1, xd = g.x_dim, a_ld = a.a_ld, sim = g.similarity;
    float* G = lds + a.l_G;
    if (sim == RGL_SIM_EQUAL_ATTENTION || sim == RGL_SIM_DIAGONAL) {
        for (int idx = threadIdx.x; idx < N * N; idx += kThreads) {
            const int i = idx / N, j = idx - i * N;
            A[i * a_ld + j] = sim == RGL_SIM_EQUAL_ATTENTION ? 1.f / (float)N : (i == j ? 1.f : 0.f);
        }
        __syncthreads();
        return;
    }
    if (sim == RGL_SIM_CONCATENATION) {
        // pair MLP 2X -> hid -> 1 with ReLU after both layers; the first layer split into the halves acting on x_i and x_j
        const RglMlp& m = g.w_a_mlp;
        const int hid = a.hid;
        float* P = lds + a.l_P;
        float* Q = lds + a.l_Q;
        const float* __restrict__ W1 = m.weight[0];      // [2 xd][hid], k-major
        for (int idx = threadIdx.x; idx < N * hid; idx += kThreads) {
            const int i = idx / hid, h = idx - i * hid;
            float p = 0.f, q = 0.f;
            for (int k = 0; k < xd; ++k) {
                p = fmaf(X[i * xd + k], W1[k * hid + h], p);
                q = fmaf(X[i * xd + k], W1[(xd + k) * hid + h], q);
            }
            P[idx] = p;
            Q[idx] = q;
        }
        __syncthreads();
        const float* __restrict__ b1 = m.bias[0];
        const float* __restrict__ w2 = m.weight[1];      // [hid][1]
        const float b2 = m.bias[1][0];
        for (int idx = threadIdx.x; idx < N * N; idx += kThreads) {
            const int i = idx / N, j = idx - i * N;
            float acc = b2;
#pragma unroll 16
            for (int h = 0; h < hid; ++h) acc = fmaf(fmaxf(P[i * hid + h] + Q[j * hid + h] + b1[h], 0.f), w2[h], acc);
            A[i * a_ld + j] = fmaxf(acc, 0.f);
        }
        __syncthreads();
        return;
    }
    const bool embedded = sim == RGL_SIM_EMBEDDED_GAUSSIAN;
    const bool cosine = sim == RGL_SIM_COSINE || sim == RGL_SIM_COSINE_SOFTMAX;
    if (embedded) {
        for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
            const int i = idx / xd, c = idx - i * xd;
            float acc = 0.f;
#pragma unroll 16
            for (int k = 0; k < xd; ++k) acc = fmaf(X[i * xd + k], g.w_a[k * xd + c], acc);
            G[idx] = acc;
        }
        __syncthreads();
    }
    const float* GX = embedded ? G : X;
    float* S = cosine ? lds + a.l_S : A;
    for (int idx = threadIdx.x; idx < N * N; idx += kThreads) {
        const int i = idx / N, j = idx - i * N;
        float acc = 0.f;
#pragma unroll 16
        for (int k = 0; k < xd; ++k) acc = fmaf(GX[i * xd + k], X[j * xd + k], acc);
        S[i * a_ld + j] = acc;
    }
    __syncthreads();
    if (cosine) {
        float* mnorm = lds + a.l_m;
        float* Cm = lds + a.l_C;
        for (int i = threadIdx.x; i < N; i += kThreads) {
            float z = 0.f;
            for (int j = 0; j < N; ++j) z = fmaf(S[i * a_ld + j], S[i * a_ld + j], z);
            mnorm[i] = sqrtf(z);
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < N * N; idx += kThreads) {
            const int i = idx / N, j = idx - i * N;
            const float c = S[i * a_ld + j] / (mnorm[i] * mnorm[j]);
            Cm[i * a_ld + j] = c;
            A[i * a_ld + j] = c;
        }
        __syncthreads();
        if (sim == RGL_SIM_COSINE) return;
    }
    for (int i = threadIdx.x; i < N; i += kThreads) {
        float* r = A + i * a_ld;
        if (sim == RGL_SIM_SQUARED) {                       // S^2 / sum_j S^2   (graph_model.py:86-89)
            float sum = 0.f;
            for (int j = 0; j < N; ++j) { r[j] = r[j] * r[j]; sum += r[j]; }
            for (int j = 0; j < N; ++j) r[j] = r[j] / sum;
        } else {                                            // row softmax
            float mx = r[0];
            for (int j = 1; j < N; ++j) mx = fmaxf(mx, r[j]);
            float sum = 0.f;
            for (int j = 0; j < N; ++j) { r[j] = expf(r[j] - mx); sum += r[j]; }
            for (int j = 0; j < N; ++j) r[j] = r[j] / sum;
        }
    }
    __syncthreads();
}

// dA [N][a_ld] = dL/dA on entry (destroyed).  dX [N][xd] += dL/dX through the similarity.  Parameter gradients of the block
// (w_a, or the pair MLP) are written to the slab when `first`, added otherwise (layerwise graphs apply the block L times; the
// order of the additions is fixed, so the result stays deterministic).  Requires sim_forward(X) to have run last.
__device__ void sim_backward(const BackwardArgs& a, float* lds, const float* X, const float* A, float* dA, float* dX,
                             float* slab, bool first) {
    const RglGraph& g = a.g;
    const int N = a.H + 1, xd = g.x_dim, a_ld = a.a_ld, sim = g.similarity;
    if (sim == RGL_SIM_EQUAL_ATTENTION || sim == RGL_SIM_DIAGONAL) return;           // A is a constant
    if (sim == RGL_SIM_CONCATENATION) {
        const RglMlp& m = g.w_a_mlp;
        const int hid = a.hid;
        const float* P = lds + a.l_P;
        const float* Q = lds + a.l_Q;
        float* dP = lds + a.l_dP;
        float* dQ = lds + a.l_dQ;
        const float* __restrict__ W1 = m.weight[0];
        const float* __restrict__ b1 = m.bias[0];
        const float* __restrict__ w2 = m.weight[1];
        // dz_ij = dA_ij where the output ReLU is open (A_ij > 0)
        for (int idx = threadIdx.x; idx < N * N; idx += kThreads) {
            const int i = idx / N, j = idx - i * N;
            if (!(A[i * a_ld + j] > 0.f)) dA[i * a_ld + j] = 0.f;
        }
        __syncthreads();
        // hidden_ij[h] = relu(P_i[h] + Q_j[h] + b1[h]);  d hidden = dz w2[h] where open
        for (int idx = threadIdx.x; idx < N * hid; idx += kThreads) {
            const int i = idx / hid, h = idx - i * hid;
            float accp = 0.f, accq = 0.f;
            for (int j = 0; j < N; ++j) {
                if (P[i * hid + h] + Q[j * hid + h] + b1[h] > 0.f) accp = fmaf(dA[i * a_ld + j], w2[h], accp);     // row i, pair (i, j)
                if (P[j * hid + h] + Q[i * hid + h] + b1[h] > 0.f) accq = fmaf(dA[j * a_ld + i], w2[h], accq);     // column i, pair (j, i)
            }
            dP[idx] = accp;
            dQ[idx] = accq;
        }
        float* gW1 = slab + a.o_wam.w[0];
        float* gb1 = slab + a.o_wam.b[0];
        float* gw2 = slab + a.o_wam.w[1];
        float* gb2 = slab + a.o_wam.b[1];
        for (int h = threadIdx.x; h < hid; h += kThreads) {              // d w2[h] = sum_ij dz_ij hidden_ij[h]
            float acc = 0.f;
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j)
                    acc = fmaf(dA[i * a_ld + j], fmaxf(P[i * hid + h] + Q[j * hid + h] + b1[h], 0.f), acc);
            gw2[h] = first ? acc : gw2[h] + acc;
        }
        if (threadIdx.x == 0) {
            float acc = 0.f;
            for (int i = 0; i < N; ++i)
#pragma unroll 16
                for (int j = 0; j < N; ++j) acc += dA[i * a_ld + j];
            gb2[0] = first ? acc : gb2[0] + acc;
        }
        __syncthreads();
        for (int h = threadIdx.x; h < hid; h += kThreads) {              // d b1 = sum_i dP_i  (= sum_j dQ_j)
            float acc = 0.f;
#pragma unroll 16
            for (int i = 0; i < N; ++i) acc += dP[i * hid + h];
            gb1[h] = first ? acc : gb1[h] + acc;
        }
        for (int idx = threadIdx.x; idx < 2 * xd * hid; idx += kThreads) {   // d W1a = X^T dP ; d W1b = X^T dQ   (torch layout [hid][2 xd])
            const int h = idx / (2 * xd), k = idx - h * (2 * xd);
            const float* D = k < xd ? dP : dQ;
            const int kk = k < xd ? k : k - xd;
            float acc = 0.f;
#pragma unroll 16
            for (int i = 0; i < N; ++i) acc = fmaf(X[i * xd + kk], D[i * hid + h], acc);
            gW1[idx] = first ? acc : gW1[idx] + acc;
        }
        for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {     // dX_i += W1a dP_i + W1b dQ_i
            const int i = idx / xd, k = idx - i * xd;
            float acc = dX[idx];
            for (int h = 0; h < hid; ++h) {
                acc = fmaf(dP[i * hid + h], W1[k * hid + h], acc);
                acc = fmaf(dQ[i * hid + h], W1[(xd + k) * hid + h], acc);
            }
            dX[idx] = acc;
        }
        __syncthreads();
        return;
    }
    const bool embedded = sim == RGL_SIM_EMBEDDED_GAUSSIAN;
    const bool cosine = sim == RGL_SIM_COSINE || sim == RGL_SIM_COSINE_SOFTMAX;
    float* G = lds + a.l_G;
    const float* GX = embedded ? G : X;
    // ---- through the row normalisation: dA -> dS (or dC for the cosine family), in place
    if (sim != RGL_SIM_COSINE) {
        for (int i = threadIdx.x; i < N; i += kThreads) {
            float dot = 0.f;
            for (int j = 0; j < N; ++j) dot = fmaf(dA[i * a_ld + j], A[i * a_ld + j], dot);
            if (sim == RGL_SIM_SQUARED) {       // dS_ij = (2 S_ij / sum_k S_ik^2) (dA_ij - sum_k dA_ik A_ik), S recomputed
                float z = 0.f;
                for (int j = 0; j < N; ++j) {
                    float sij = 0.f;
                    for (int k = 0; k < xd; ++k) sij = fmaf(GX[i * xd + k], X[j * xd + k], sij);
                    z = fmaf(sij, sij, z);
                }
                for (int j = 0; j < N; ++j) {
                    float sij = 0.f;
                    for (int k = 0; k < xd; ++k) sij = fmaf(GX[i * xd + k], X[j * xd + k], sij);
                    dA[i * a_ld + j] = 2.f * sij / z * (dA[i * a_ld + j] - dot);
                }
            } else {                            // softmax: dS_ij = A_ij (dA_ij - sum_k dA_ik A_ik)
                for (int j = 0; j < N; ++j) dA[i * a_ld + j] = A[i * a_ld + j] * (dA[i * a_ld + j] - dot);
            }
        }
        __syncthreads();
    }
    if (cosine) {
        // C_ij = S_ij / (m_i m_j), m_i = |S_i,:|_2.  dA holds dC.  m_i enters row i and column i:
        //   dm_i = -(1/m_i) (sum_j dC_ij C_ij + sum_j dC_ji C_ji) ;  dS_ij = dC_ij / (m_i m_j) + dm_i S_ij / m_i
        const float* S = lds + a.l_S;
        const float* Cm = lds + a.l_C;
        float* mnorm = lds + a.l_m;
        float* dm = lds + a.l_m + N;                        // [N] right behind m
        for (int i = threadIdx.x; i < N; i += kThreads) {
            float acc = 0.f;
            for (int j = 0; j < N; ++j) {
                acc = fmaf(dA[i * a_ld + j], Cm[i * a_ld + j], acc);
                acc = fmaf(dA[j * a_ld + i], Cm[j * a_ld + i], acc);
            }
            dm[i] = -acc / mnorm[i];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < N * N; idx += kThreads) {
            const int i = idx / N, j = idx - i * N;
            dA[i * a_ld + j] = dA[i * a_ld + j] / (mnorm[i] * mnorm[j]) + dm[i] * S[i * a_ld + j] / mnorm[i];
        }
        __syncthreads();
    }
    // ---- S = G X^T :  dG = dS X ;  dX += dS^T G   (dG kept in the dG buffer)
    float* dG = lds + a.l_dG;
    for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
        const int i = idx / xd, k = idx - i * xd;
        float acc = 0.f;
#pragma unroll 16
        for (int j = 0; j < N; ++j) acc = fmaf(dA[i * a_ld + j], X[j * xd + k], acc);
        dG[idx] = acc;
    }
    for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
        const int j = idx / xd, k = idx - j * xd;
        float acc = dX[idx];
#pragma unroll 16
        for (int i = 0; i < N; ++i) acc = fmaf(dA[i * a_ld + j], GX[i * xd + k], acc);
        dX[idx] = acc;
    }
    __syncthreads();
    if (embedded) {   // G = X Wa :  dWa = X^T dG ;  dX += dG Wa^T
        float* gWa = slab + a.o_wa;
        for (int idx = threadIdx.x; idx < xd * xd; idx += kThreads) {
            const int k = idx / xd, c = idx - k * xd;
            float acc = 0.f;
#pragma unroll 16
            for (int i = 0; i < N; ++i) acc = fmaf(X[i * xd + k], dG[i * xd + c], acc);
            gWa[idx] = first ? acc : gWa[idx] + acc;
        }
        for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
            const int i = idx / xd, k = idx - i * xd;
            float acc = dX[idx];
#pragma unroll 16
            for (int c = 0; c < xd; ++c) acc = fmaf(dG[i * xd + c], g.w_a[k * xd + c], acc);
            dX[idx] = acc;
        }
    } else {          // G = X
        for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) dX[idx] += dG[idx];
    }
    __syncthreads();
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
    float* Aall = lds + a.l_A;    // [nA][N][a_ld]
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
    const bool layerwise = g.layerwise_graph != 0;
    const bool embedded = g.similarity == RGL_SIM_EMBEDDED_GAUSSIAN;
    const bool concat = g.similarity == RGL_SIM_CONCATENATION;

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
        if (!layerwise) sim_forward(a, lds, X, Aall);
        for (int l = 0; l < L; ++l) {
            const float* Hl = Hs + l * N * xd;
            float* Tl = T + l * N * xd;
            float* Rl = R + l * N * xd;
            float* Hn = Hs + (l + 1) * N * xd;
            float* A = layerwise ? Aall + l * N * a_ld : Aall;
            if (layerwise) sim_forward(a, lds, Hl, A);
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
                const int i = idx / xd, c = idx - i * xd;
                float acc = 0.f;
#pragma unroll 16
                for (int j = 0; j < N; ++j) acc = fmaf(A[i * a_ld + j], Hl[j * xd + c], acc);
                Tl[idx] = acc;
            }
            __syncthreads();
            const float* __restrict__ W = g.Ws[l];
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {
                const int i = idx / xd, c = idx - i * xd;
                float acc = 0.f;
#pragma unroll 16
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
        if (a.detach_graph || L == 0) {
            // StatePredictor(..., detach=True): the embedding is a constant; only the head learns
            if (a.detach_graph || L == 0) {
                if (a.detach_graph) {
                    zero_mlp_grads(g.w_r, a.o_wr, slab);
                    zero_mlp_grads(g.w_h, a.o_wh, slab);
                }
                if (embedded) for (int i = threadIdx.x; i < xd * xd; i += kThreads) slab[a.o_wa + i] = 0.f;
                if (concat) zero_mlp_grads(g.w_a_mlp, a.o_wam, slab);
                for (int l = 0; l < L; ++l) for (int i = threadIdx.x; i < xd * xd; i += kThreads) slab[a.o_ws[l] + i] = 0.f;
                __syncthreads();
            }
            if (a.detach_graph) continue;
        }
        if (!layerwise) {
            for (int idx = threadIdx.x; idx < N * a_ld; idx += kThreads) dA[idx] = 0.f;
            __syncthreads();
        }
        float* dcur = dH;
        float* dnxt = dH2;
        bool first_sim = true;
        for (int l = L - 1; l >= 0; --l) {
            const float* Hl = Hs + l * N * xd;
            const float* Tl = T + l * N * xd;
            const float* Rl = R + l * N * xd;
            const float* A = layerwise ? Aall + l * N * a_ld : Aall;
            const float* __restrict__ W = g.Ws[l];
            // dZ = dH_{l+1} o (R_l > 0)   (kept in dG as scratch)
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) dG[idx] = Rl[idx] > 0.f ? dcur[idx] : 0.f;
            __syncthreads();
            float* gW = slab + a.o_ws[l];
            for (int idx = threadIdx.x; idx < xd * xd; idx += kThreads) {
                const int k = idx / xd, c = idx - k * xd;
                float acc = 0.f;
#pragma unroll 16
                for (int i = 0; i < N; ++i) acc = fmaf(Tl[i * xd + k], dG[i * xd + c], acc);
                gW[idx] = acc;
            }
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {      // dT = dZ W^T
                const int i = idx / xd, k = idx - i * xd;
                float acc = 0.f;
#pragma unroll 16
                for (int c = 0; c < xd; ++c) acc = fmaf(dG[i * xd + c], W[k * xd + c], acc);
                dT[idx] = acc;
            }
            __syncthreads();
            for (int idx = threadIdx.x; idx < N * N; idx += kThreads) {       // dA (+)= dT H_l^T
                const int i = idx / N, j = idx - i * N;
                float acc = layerwise ? 0.f : dA[i * a_ld + j];
#pragma unroll 16
                for (int k = 0; k < xd; ++k) acc = fmaf(dT[i * xd + k], Hl[j * xd + k], acc);
                dA[i * a_ld + j] = acc;
            }
            for (int idx = threadIdx.x; idx < N * xd; idx += kThreads) {      // dH_l = A^T dT (+ dH_{l+1})
                const int j = idx / xd, k = idx - j * xd;
                float acc = g.skip_connection ? dcur[idx] : 0.f;
#pragma unroll 16
                for (int i = 0; i < N; ++i) acc = fmaf(A[i * a_ld + j], dT[i * xd + k], acc);
                dnxt[idx] = acc;
            }
            __syncthreads();
            if (layerwise) {                  // A_l = sim(H_l): recompute the block's internals for this H_l, then go through it
                sim_forward(a, lds, Hl, Aall + l * N * a_ld);
                sim_backward(a, lds, Hl, A, dA, dnxt, slab, first_sim);
                first_sim = false;
            }
            float* t = dcur;
            dcur = dnxt;
            dnxt = t;
        }
        // dcur = dL/dX from the layers; one adjacency for all layers: its gradient was accumulated in dA
        if (!layerwise) sim_backward(a, lds, X, Aall, dA, dcur, slab, true);
        // embeddings: row 0 -> w_r, rows 1..H -> w_h
        for (int f = threadIdx.x; f < xd; f += kThreads) d0[f] = dcur[f];
        __syncthreads();
        mlp_backward(g.w_r, a.o_wr, AR, a.ar_ld, 1, d0, d1, d_ld, slab, false);
        for (int idx = threadIdx.x; idx < H * xd; idx += kThreads) d0[(idx / xd) * d_ld + (idx % xd)] = dcur[(idx / xd + 1) * xd + (idx % xd)];
        __syncthreads();
        mlp_backward(g.w_h, a.o_wh, AH, a.ah_ld, H, d0, d1, d_ld, slab, false);
        __syncthreads();
    }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int n_scenes, int n_params, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_params) return;
    // fixed order (deterministic), but 16 loads in flight at a time: a plain loop is one dependent-latency chain per scene
    // (25 us for 100 scenes)
    float acc = 0.f;
    int s = 0;
    for (; s + 16 <= n_scenes; s += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = slabs[(size_t)(s + u) * n_params + k];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; s < n_scenes; ++s) acc += slabs[(size_t)s * n_params + k];
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
    a.g = *graph;
    a.has_vhead = (vh && vh->n_layers > 0) ? 1 : 0;
    a.has_mhead = (mh && mh->n_layers > 0) ? 1 : 0;
    a.vhead = a.has_vhead ? *vh : RglMlp{};
    a.mhead = a.has_mhead ? *mh : RglMlp{};
    if (a.has_vhead && (rc = rgl::validate_mlp(a.vhead, graph->x_dim, 1))) return rc;
    if (a.has_mhead && (rc = rgl::validate_mlp(a.mhead, graph->x_dim, 0))) return rc;
    // slab order: w_r (W0,b0,W1,b1,..), w_h, w_a (embedded_gaussian: matrix; concatenation: its pair MLP), Ws[0..L-1],
    // value head, motion head
    int off = 0;
    off = assign_mlp(a.g.w_r, a.o_wr, off);
    off = assign_mlp(a.g.w_h, a.o_wh, off);
    a.o_wa = off;
    if (graph->similarity == RGL_SIM_EMBEDDED_GAUSSIAN) off += graph->x_dim * graph->x_dim;
    if (graph->similarity == RGL_SIM_CONCATENATION) off = assign_mlp(a.g.w_a_mlp, a.o_wam, off);
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
    a.a_ld = N + 1;
    // scratch of the similarity block: what one mode needs, the modes never coexist
    a.hid = graph->similarity == RGL_SIM_CONCATENATION ? graph->w_a_mlp.dims[1] : 0;
    a.l_G = a.l_S = a.l_C = a.l_m = a.l_P = a.l_Q = a.l_dP = a.l_dQ = lo;
    if (graph->similarity == RGL_SIM_CONCATENATION) {
        a.l_P = take(N * a.hid); a.l_Q = take(N * a.hid); a.l_dP = take(N * a.hid); a.l_dQ = take(N * a.hid);
    } else if (graph->similarity == RGL_SIM_COSINE || graph->similarity == RGL_SIM_COSINE_SOFTMAX) {
        a.l_S = take(N * a.a_ld); a.l_C = take(N * a.a_ld); a.l_m = take(2 * N);
    } else {
        a.l_G = take(N * xd);
    }
    a.l_A = take((graph->layerwise_graph && L > 0 ? L : 1) * N * a.a_ld);
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
    const bool too_big = rc == RGL_ERR_LDS;       // a scene's activations do not fit one CU's LDS: only the tile pipeline can run
    if (rc && !too_big) return rc;
    if (workspace_bytes < (size_t)n_scenes * a.n_params * sizeof(float)) return RGL_ERR_WORKSPACE;
    // large batches of the shipped structure: the tile pipeline on the matrix cores (rgl_backward_mfma.hip); 1 = not its case
    rc = rgl::launch_backward_mfma(graph, value_head, motion_head, robot, humans, n_scenes, H, detach_graph, d_value, d_humans_next,
                                   d_H, grad_out, workspace, workspace_bytes, (hipStream_t)stream, too_big ? 1 : 0);
    if (rc != 1) return rc;
    if (too_big) return RGL_ERR_LDS;
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
