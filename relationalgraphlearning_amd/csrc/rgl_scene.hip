// rgl_scene.hip -- state-predictor path of the rollout: scenes with their own crowds (one graph forward per tree node).
// Follows (reference paths): crowd_nav/policy/state_predictor.py:20-39, graph_model.py:99-130.
#include "rgl_children.h"
#include "rgl_mlp_chain.h"

namespace {

// ------------------------------------------------------------------------------------------------
// state-predictor path: scenes with their own crowds (one graph forward per tree node)
//   row_mlp2_kernel   : batched 2-layer embedding MLP over rows (IN -> 64 -> 32, ReLU after both) as an MFMA chain
//   scene_graph_kernel: one wave per scene: S = (X Wa) X^T, softmax, L x relu(A H W)(+H), motion head 32->64->5
// ------------------------------------------------------------------------------------------------
struct RowMlpArgs {
    const float *w1, *b1, *w2, *b2;   // k-major [IN][64], [64], [64][32], [32]
    const float* rows;                // [M][IN]
    float* out;                       // [M][32]
    int M, n_tiles;
};

// One wave per 16-row tile: IN -> 64 -> 32 with ReLU after both layers.
template <int IN>
__device__ __forceinline__ void row_mlp2_tiles(const RowMlpArgs& a, const float* lds_set, int first, int stride, int lane) {
    constexpr int F1 = 0, F2 = F1 + 4 * 1 * 4 * 64, B1 = F2 + 2 * 4 * 4 * 64, B2 = B1 + HID;
    const int n = lane & 15, q = lane >> 4;
    for (int tile = first; tile < a.n_tiles; tile += stride) {
        const int row = 16 * tile + n;
        const int rc = row < a.M ? row : a.M - 1;
        const float* src = a.rows + (size_t)rc * IN;
        f32x4 in[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int feat = tile_feature<IN>(0, q, r);
            in[0][r] = feat < IN ? src[feat] : 0.f;
        }
        f32x4 h[4];
        layer_mfma<IN, HID, true>(lds_set + F1, in, h, lane, lds_set + B1);
        relu_tiles<HID>(h);
        f32x4 o[2];
        layer_mfma<HID, XD, true>(lds_set + F2, h, o, lane, lds_set + B2);
        relu_tiles<XD>(o);
        if (row < a.M) {
            float* dst = a.out + (size_t)row * XD;
            *reinterpret_cast<f32x4*>(dst + 4 * q) = o[0];
            *reinterpret_cast<f32x4*>(dst + 16 + 4 * q) = o[1];
        }
    }
}

constexpr int kRowMlpSetFloats = 4 * 1 * 4 * 64 + 2 * 4 * 4 * 64 + HID + XD;

// Both embedding MLPs of a level in ONE launch (they are launch-latency sized): workgroups [0, grid_a) take the robot rows
// (INA inputs), the others the human rows (INB inputs).  (9, 5): path M's full / observable states; (6, 7): path G's rotated
// self / human features (gcn.py:34-47).
// `grid_mlp` < gridDim.x: the workgroups past the two MLPs run the level's independent next-robot-state / reward work (rgl_children.h)
// in the same launch -- with many scenes that work does not ride in the scene kernel (see scene_graph_kernel) and was a launch of its
// own between this one and the scene kernel.
template <int INA, int INB>
__global__ __launch_bounds__(kThreads, 2) void row_mlp2_pair_kernel(const RowMlpArgs ra, const RowMlpArgs rb, int grid_a, int grid_mlp,
                                                                    const ChildrenArgs ca) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x >= grid_mlp) {
        const long long total = (long long)ca.P * ca.A, stride = (long long)(gridDim.x - grid_mlp) * kThreads;
        const long long first_pair = (long long)(blockIdx.x - grid_mlp) * kThreads + threadIdx.x;
        if (ca.A >= 64 && ca.H <= 64 && !ca.robot64) {
            const float v_max = table_speed_bound(ca);
            for (long long base = first_pair & ~63LL; base < total; base += stride) children_wave(ca, base, total, v_max);
        } else {
            for (long long idx = first_pair; idx < total; idx += stride) children_thread(ca, idx);
        }
        return;
    }
    constexpr int F1 = 0, F2 = F1 + 4 * 1 * 4 * 64, B1 = F2 + 2 * 4 * 4 * 64, B2 = B1 + HID;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool first = (int)blockIdx.x < grid_a;
    const RowMlpArgs& a = first ? ra : rb;
    if (first) fill_frags<INA, HID, kThreads>(lds + F1, a.w1, tid); else fill_frags<INB, HID, kThreads>(lds + F1, a.w1, tid);
    fill_frags<HID, XD, kThreads>(lds + F2, a.w2, tid);
    fill_bias<HID>(lds + B1, a.b1, tid);
    fill_bias<XD>(lds + B2, a.b2, tid);
    __syncthreads();
    const int b = first ? blockIdx.x : blockIdx.x - grid_a, g = first ? grid_a : grid_mlp - grid_a;
    if (first) row_mlp2_tiles<INA>(a, lds, b + g * wave, g * kWaves, lane);       // partial round: one tile per workgroup
    else row_mlp2_tiles<INB>(a, lds, b + g * wave, g * kWaves, lane);
}

struct SceneArgs {
    // EMB kernels (few scenes: the embedding launch would cost more than its work): raw state rows + the two embedding MLPs
    const float* robot_rows;           // [P][9]
    const float* human_rows;           // [n_crowds][H][5]
    const float *er_w1, *er_b1, *er_w2, *er_b2;      // w_r, k-major [9][64], [64], [64][32], [32]
    const float *eh_w1, *eh_b1, *eh_w2, *eh_b2;      // w_h
    int off_er, off_eh;                // LDS: fragment sets of the two MLPs (kRowMlpSetFloats each)
    int bx;                            // launch the bf16 six-term (BX) form: the WEIGHT products (Wa, W_l, motion head) as layer_mfma_b6
                                       // over a packed image of three-piece fragments (RGL_CONTRACT_BF16X6); S and A H stay f32
    int ws_stride;                     // floats between the layer matrices in the LDS image
    int image_floats;                  // BX: floats of the packed image
    const float* image;                // BX: the weight image in this kernel's LDS layout (pack_scene_image)
    const float* xh_rows;              // [n_crowds][H][32]  human embeddings
    const float* x0_rows;              // [P][32]            robot embeddings
    int crowds_per;                    // scene s uses crowd s / crowds_per
    const float* wa;                   // [32][32]
    const float* Ws[RGL_MAX_GCN_LAYERS];
    int L, skip;
    int sim;                           // SIM_* row normalisation
    const float *wm1, *bm1, *wm2, *bm2;   // motion head, k-major [32][64], [64], [64][5], [5]
    float* humans_next;                // [P][H][5]   (state predictor)
    float* rows_out;                   // null, or [P][64]: value mode -- rows [ (A H_{L-1})[robot] | H_{L-1}[robot] ] for robot_head_kernel
    int layerwise;                     // adjacency recomputed from H_l in every layer (graph_model.py:119-122)
    const float *wc1, *bc1, *wc2, *bc2;   // concatenation: pair MLP, k-major [64][64], [64], [64][1], [1]
    int off_wc1, off_bc1, off_wc2;
    int P, H, N;
    int off_wa, off_ws, off_wm1, off_bm1, off_wm2, off_bm2, off_wave, wave_stride;
};

constexpr int M2LD = 20;   // LDS row stride of the [64][5 -> 16] motion output layer (4*M2LD % 32 == 16)
// N <= 32: 8 waves share one 30 KB weight image, two workgroups per CU = 4 waves per SIMD (round 1 ran 4-wave workgroups, 2 waves
// per SIMD: a scene is one serial chain of ~290 MFMAs with a softmax in the middle, and two waves did not cover its latencies).
// Larger crowds (3-4 column tiles: 200+ VGPRs, 9 KB of node features per wave) keep 4-wave workgroups, two per CU.

// One scene per wave.  CH = true: the level's independent next-robot-state / reward work (float64 VALU) rides in this launch on
// EXTRA workgroups [grid_scene, gridDim.x).  That pays while the scene workgroups leave LDS free (few scenes: the two halves
// overlap and a launch is saved); with many scenes the extra workgroups -- which reserve the same dynamic LDS -- only start when
// a scene workgroup retires, and the second code path costs the kernel a third of its occupancy in registers: the launcher then
// uses CH = false and the caller launches mprl_children_kernel (measured cross-over ~3 k scenes; an in-wave variant, lanes =
// actions with scalar crowd loads, was tried and was never better).
// SK: 0 = softmax of S (embedded_gaussian / gaussian), 1 = plain weights over their row sum (squared / equal_attention /
// diagonal), 2 = cosine family (cosine / cosine_softmax; graph_model.py:70-79), 3 = concatenation (pair MLP, :80-85)
// SPLIT: the NT column tiles of a scene go to NT waves (WAVES / NT scenes per workgroup pass, node features shared in LDS,
// workgroup barriers where a phase needs every row): a scene is one serial chain of ~70 MFMAs per column tile with a softmax in
// the middle, and with few scenes (the upper tree levels, dense crowds: 256-512 scenes of 50 agents on 256 CUs) nothing else hides
// that chain.  With thousands of scenes the unsplit form -- no barriers, the same MFMA count -- is as fast or faster.
// EMB: the wave computes the embeddings of its node tiles itself (w_r on the robot row, w_h on the human rows: the MFMA chains of
// row_mlp2_tiles) instead of reading rows a separate launch prepared -- with few scenes that launch is ~5 us of latency for ~1 us
// of work (and a round trip through HBM); sibling scenes repeat their crowd's human embeddings, which only matters when the
// kernel is throughput-bound (many scenes: the launcher keeps the two-launch form there).
// BX (RGL_CONTRACT_BF16X6, softmax similarity): the WEIGHT products as six bf16 MFMA terms over three-piece operands (layer_mfma_b6).
template <int NT, int SK, int WAVES, bool CH, bool SPLIT, bool EMB = false, bool BX = false>
__global__ __launch_bounds__(WAVES * 64, 2) void scene_graph_kernel(const SceneArgs a, const ChildrenArgs ca, int grid_scene) {
    static_assert(!BX || SK == 0, "six-term bf16 products: softmax similarity");
    static_assert(!SPLIT || (SK != 3 && NT > 1 && WAVES % NT == 0), "split scenes: whole scenes per workgroup, no pair-MLP similarity");
    constexpr int kSceneThreads = WAVES * 64;
    constexpr int kSlots = SPLIT ? WAVES / NT : WAVES;      // scenes in flight per workgroup
    constexpr int NCT = SPLIT ? 1 : NT;                     // column tiles of a scene this wave owns
    if constexpr (CH) {
        if ((int)blockIdx.x >= grid_scene) {
            const long long total = (long long)ca.P * ca.A, stride = (long long)(gridDim.x - grid_scene) * kSceneThreads;
            const long long first = (long long)(blockIdx.x - grid_scene) * kSceneThreads + threadIdx.x;
            if (ca.A >= 64 && ca.H <= 64 && !ca.robot64) {       // whole waves: far-human masks per parent (children_wave)
                const float v_max = table_speed_bound(ca);
                for (long long base = first & ~63LL; base < total; base += stride) children_wave(ca, base, total, v_max);
            } else {
                for (long long idx = first; idx < total; idx += stride) children_thread(ca, idx);
            }
            return;
        }
    }
    const int sim = SK == 0 ? (int)SIM_SOFTMAX : a.sim;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int N = a.N, H = a.H;
    const float* wa = lds + a.off_wa;       // [32][WLD]
    const float* ws = lds + a.off_ws;       // [L][32][WLD]
    const float* wm1 = lds + a.off_wm1;     // [32][W1LD]
    const float* bm1 = lds + a.off_bm1;     // [64]
    const float* wm2 = lds + a.off_wm2;     // [64][M2LD], columns >= 5 zero
    const float* bm2 = lds + a.off_bm2;     // [16], entries >= 5 zero
    const int slot = __builtin_amdgcn_readfirstlane(SPLIT ? wave / NT : wave);
    const int ctb = SPLIT ? __builtin_amdgcn_readfirstlane(wave % NT) : 0;      // my first (SPLIT: only) column tile
    float* Hs = lds + a.off_wave + slot * a.wave_stride;   // [16*NT][XLD] node features of the slot's current scene
    // EMB: the fragment sets of the two embedding MLPs.  Their loads go out together with the weight image's (f32 image: before its
    // stores), so that the whole prologue is ONE L2 round trip instead of one per fill (round 4)
    constexpr int kEmbThreads = EMB ? kSceneThreads : 64;
    FragRegs<9, HID, kEmbThreads> er1;
    FragRegs<HID, XD, kEmbThreads> er2, eh2;
    FragRegs<5, HID, kEmbThreads> eh1;
    float ebias[4] = {0.f, 0.f, 0.f, 0.f};
    auto emb_loads = [&]() {
        if constexpr (EMB) {
            frag_load(er1, a.er_w1, tid);
            frag_load(er2, a.er_w2, tid);
            frag_load(eh1, a.eh_w1, tid);
            frag_load(eh2, a.eh_w2, tid);
            ebias[0] = bias_load<HID>(a.er_b1, tid);
            ebias[1] = bias_load<XD>(a.er_b2, tid);
            ebias[2] = bias_load<HID>(a.eh_b1, tid);
            ebias[3] = bias_load<XD>(a.eh_b2, tid);
        }
    };
    if constexpr (BX) {
        // three-piece bf16 image, packed once per parameter state (or per search) in exactly this layout: b128 copies, every
        // load of a thread in flight at once.  (Converting the matrices here -- two L2 round trips per fragment element -- cost 10 us per launch.)
        // (round 6: LDS-direct loads, every chunk of a wave in flight at once.  The b128 copy loop this replaces compiled to load -> wait
        // -> store per iteration: 19 dependent L2 round trips, ~2.5 us of every launch.)
        constexpr int kChunk = 256;                                            // floats per wave and instruction
        const int n_chunks = (a.image_floats + kChunk - 1) / kChunk;
        for (int c = wave; c < n_chunks; c += WAVES) {
            const int fl = c * kChunk + lane * 4;
            if (fl < a.image_floats)                                           // the last chunk is partial (image_floats % 4 == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.image + fl),
                                                 (__attribute__((address_space(3))) void*)(lds + c * kChunk), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0): my chunks have landed (the barrier below: everyone's)
    } else
    {   // weight image in two phases -- every global load of the thread first, then the LDS stores -- so that the whole
        // 30 KB image costs ONE L2 round trip (filling matrix by matrix cost one per matrix: ~9 us of a ~35 us launch)
        float* w = lds;
        constexpr int NT_ = kSceneThreads;
        constexpr int KQ = XD * XD / NT_, KM1 = XD * HID / NT_, KM2 = HID * 16 / NT_;
        float vq[5][KQ];                                       // wa + up to 4 layer matrices
        float vm1[KM1], vm2[KM2], vb;
        const int Lc = a.L < 4 ? a.L : 4;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int i = tid + k * NT_;
            vq[0][k] = a.wa ? a.wa[i] : ((i / XD) == (i % XD) ? 1.f : 0.f);      // gaussian: Wa = I
#pragma unroll
            for (int l = 0; l < 4; ++l) vq[1 + l][k] = l < Lc ? a.Ws[l][i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KM1; ++k) vm1[k] = a.wm1 ? a.wm1[tid + k * NT_] : 0.f;        // value mode: no motion head
#pragma unroll
        for (int k = 0; k < KM2; ++k) {
            const int i = tid + k * NT_, r = i / 16, c = i - r * 16;
            vm2[k] = a.wm2 ? a.wm2[r * 5 + (c < 5 ? c : 0)] : 0.f;
        }
        vb = !a.bm1 ? 0.f : (tid < HID ? a.bm1[tid] : (tid < HID + 5 ? a.bm2[tid - HID] : 0.f));
        emb_loads();
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int i = tid + k * NT_, r = i / XD, c = i - r * XD;
            w[a.off_wa + r * WLD + c] = vq[0][k];
#pragma unroll
            for (int l = 0; l < 4; ++l)
                if (l < Lc) w[a.off_ws + (l * XD + r) * WLD + c] = vq[1 + l][k];
        }
#pragma unroll
        for (int k = 0; k < KM1; ++k) {
            const int i = tid + k * NT_, r = i / HID, c = i - r * HID;
            w[a.off_wm1 + r * W1LD + c] = vm1[k];
        }
#pragma unroll
        for (int k = 0; k < KM2; ++k) {
            const int i = tid + k * NT_, r = i / 16, c = i - r * 16;
            w[a.off_wm2 + r * M2LD + c] = c < 5 ? vm2[k] : 0.f;
        }
        if (tid < HID) w[a.off_bm1 + tid] = vb;
        else if (tid < HID + 16) w[a.off_bm2 + tid - HID] = tid < HID + 5 ? vb : 0.f;
        if (SK == 3) {
            fill_matrix<2 * XD, 2 * XD, HID, W1LD, kSceneThreads>(w + a.off_wc1, a.wc1, tid);
            for (int i = tid; i < HID; i += kSceneThreads) { w[a.off_bc1 + i] = a.bc1[i]; w[a.off_wc2 + i] = a.wc2[i]; }
        }
    }
    if constexpr (EMB) {
        float* w = lds;
        constexpr int F1 = 0, F2 = F1 + 4 * 1 * 4 * 64, B1 = F2 + 2 * 4 * 4 * 64, B2 = B1 + HID;
        if constexpr (BX) emb_loads();
        frag_store(er1, w + a.off_er + F1, tid);
        frag_store(er2, w + a.off_er + F2, tid);
        bias_store<HID>(ebias[0], w + a.off_er + B1, tid);
        bias_store<XD>(ebias[1], w + a.off_er + B2, tid);
        frag_store(eh1, w + a.off_eh + F1, tid);
        frag_store(eh2, w + a.off_eh + F2, tid);
        bias_store<HID>(ebias[2], w + a.off_eh + B1, tid);
        bias_store<XD>(ebias[3], w + a.off_eh + B2, tid);
    }
    __syncthreads();
    // scene of slot k in pass i: blockIdx + grid_scene * k + i * grid_scene * kSlots (partial round: one scene per workgroup).  SPLIT:
    // the loop is uniform over the workgroup (barriers inside); a slot past the end recomputes the last scene and writes nothing.
    for (int it = blockIdx.x + (SPLIT ? 0 : grid_scene * slot); it < a.P; it += grid_scene * kSlots) {
        const int sc_raw = SPLIT ? it + grid_scene * slot : it;
        const bool active = sc_raw < a.P;
        const int sc = active ? sc_raw : a.P - 1;
        // node features of this scene: row 0 = robot, rows 1..H = its crowd, rows >= N zero
        if constexpr (EMB) {
            constexpr int F1 = 0, F2 = F1 + 4 * 1 * 4 * 64, B1 = F2 + 2 * 4 * 4 * 64, B2 = B1 + HID;
            const float* er = lds + a.off_er;
            const float* eh = lds + a.off_eh;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int node = 16 * (ct + ctb) + n;
                const bool human = node >= 1 && node < N;
                const float* hsrc = a.human_rows + ((size_t)(sc / a.crowds_per) * H + (human ? node - 1 : 0)) * 5;
                f32x4 in[1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int feat = tile_feature<5>(0, q, r);
                    in[0][r] = (human && feat < 5) ? hsrc[feat] : 0.f;
                }
                f32x4 hh[4], oh[2];
                layer_mfma<5, HID, true>(eh + F1, in, hh, lane, eh + B1);
                relu_tiles<HID>(hh);
                layer_mfma<HID, XD, true>(eh + F2, hh, oh, lane, eh + B2);
                relu_tiles<XD>(oh);
                if (ct + ctb == 0) {                              // the tile that holds the robot: its row through w_r
                    const float* rsrc = a.robot_rows + (size_t)sc * 9;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int feat = tile_feature<9>(0, q, r);
                        in[0][r] = (n == 0 && feat < 9) ? rsrc[feat] : 0.f;
                    }
                    f32x4 orr[2];
                    layer_mfma<9, HID, true>(er + F1, in, hh, lane, er + B1);
                    relu_tiles<HID>(hh);
                    layer_mfma<HID, XD, true>(er + F2, hh, orr, lane, er + B2);
                    relu_tiles<XD>(orr);
                    if (n == 0) { oh[0] = orr[0]; oh[1] = orr[1]; }
                }
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
                    *reinterpret_cast<f32x4*>(&Hs[node * XLD + 16 * ot + 4 * q]) = node < N ? oh[ot] : zero4();
            }
            __builtin_amdgcn_wave_barrier();
        } else {
        const float* xr = a.x0_rows + (size_t)sc * XD;
        const float* xh = a.xh_rows + (size_t)(sc / a.crowds_per) * H * XD;
        for (int idx = lane; idx < 16 * NCT * (XD / 4); idx += 64) {
            const int row = (idx >> 3) + 16 * ctb, c4 = (idx & 7) * 4;
            f32x4 val = zero4();
            if (row == 0) val = *reinterpret_cast<const f32x4*>(xr + c4);
            else if (row < N) val = *reinterpret_cast<const f32x4*>(xh + (size_t)(row - 1) * XD + c4);
            *reinterpret_cast<f32x4*>(&Hs[row * XLD + c4]) = val;
        }
        }
        if (SPLIT) __syncthreads();
        // adjacency of the node features currently in Hs, transposed and in B-operand order: pr[ct][jt][r] = A[i][j] for
        // column i = 16 ct + n, j = 16 jt + 4 q + r.  Once per scene, or once per layer for layerwise graphs.
        // Node order inside the LAST 16-node tile (PERM: every similarity but concatenation): D row 4q + r of an S^T tile holds node
        // 16 jt + 4 r + q instead of 16 jt + 4 q + r, so that as the k index of A*H the valid nodes of a partial tile sit in its first
        // ceil(valid / 4) k steps and the steps over padding nodes are skipped (N = 20: 5 of 8 steps per layer, N = 5: 2 of 4).
        constexpr bool PERM = SK != 3;
        auto jnode = [&](int jt, int r) { return 16 * jt + ((PERM && jt == NT - 1) ? 4 * r + q : 4 * q + r); };
        const int last_steps = PERM ? (N - 16 * (NT - 1) + 3) >> 2 : 4;
        f32x4 pr[NT][NT];
        auto adjacency = [&]() {
            if constexpr (SK == 3) {
                // concatenation: A_ij = relu(w2 . relu(W1a x_i + W1b x_j + b1) + b2).  P^T = W1a^T X^T and Q^T = W1b^T X^T by MFMA
                // (lane (n, q) of column tile ct holds hidden units 16 ht + 4 q + r of node 16 ct + n); the pair sum over the 64
                // hidden units: Q_j of the same q-group arrives as a DPP row_newbcast operand, the four q-groups add up at the end.
                const float* wc1 = lds + a.off_wc1;      // [64][W1LD]: rows 0..31 act on x_i, rows 32..63 on x_j
                const float* bc1 = lds + a.off_bc1;
                const float* wc2 = lds + a.off_wc2;
                const float bc2 = a.bc2[0];
                f32x4 pt[NT][4], qt[NT][4];
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                    for (int ht = 0; ht < 4; ++ht) {
                        pt[ct][ht] = *reinterpret_cast<const f32x4*>(&bc1[16 * ht + 4 * q]);      // P carries b1
                        qt[ct][ht] = zero4();
                    }
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) {
                        load_fence();
                        const f32x4 xb = *reinterpret_cast<const f32x4*>(&Hs[(16 * ct + n) * XLD + 16 * ft + 4 * q]);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int ht = 0; ht < 4; ++ht) {
                                pt[ct][ht] = mfma4(wc1[(16 * ft + 4 * q + r) * W1LD + 16 * ht + n], xb[r], pt[ct][ht]);
                                qt[ct][ht] = mfma4(wc1[(XD + 16 * ft + 4 * q + r) * W1LD + 16 * ht + n], xb[r], qt[ct][ht]);
                            }
                    }
                }
                f32x4 w2h[4];
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) w2h[ht] = *reinterpret_cast<const f32x4*>(&wc2[16 * ht + 4 * q]);
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt) {
                        f32x4 res = zero4();
                        static_for<0, 16>([&](auto jc) {
                            constexpr int jl = decltype(jc)::value;
                            float acc = 0.f;
#pragma unroll
                            for (int ht = 0; ht < 4; ++ht)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    acc = fmaf(fmaxf(dpp_rowbcast_add<jl>(qt[jt][ht][r], pt[ct][ht][r]), 0.f), w2h[ht][r], acc);
                            acc = fmaxf(kgroups_sum(acc) + bc2, 0.f);
                            const int j = 16 * jt + jl;
                            if (j >= N || 16 * ct + n >= N) acc = 0.f;
                            if ((jl >> 2) == q) res[jl & 3] = acc;
                        });
                        pr[ct][jt] = res;
                    }
                return;
            }
            // G^T = Wa^T X^T   (per column tile: [g = 16gt+4q+r][col n])
            f32x4 gt_[NT][2];
            {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if constexpr (BX) {
                    load_fence();
                    f32x4 xin[2];
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft)
                        xin[ft] = *reinterpret_cast<const f32x4*>(&Hs[(16 * (ct + ctb) + n) * XLD + 16 * ft + 4 * q]);
                    layer_mfma_b6<XD, XD, false>(wa, xin, gt_[ct], lane);
                    continue;
                }
                gt_[ct][0] = zero4();
                gt_[ct][1] = zero4();
                load_fence();
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    const f32x4 xb = *reinterpret_cast<const f32x4*>(&Hs[(16 * (ct + ctb) + n) * XLD + 16 * ft + 4 * q]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int g = 0; g < 2; ++g)
                            gt_[ct][g] = mfma4(wa[(16 * ft + 4 * q + r) * WLD + 16 * g + n], xb[r], gt_[ct][g]);
                }
            }
            // S^T[j][col] = X[j] . G[col]
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) {
                    load_fence();
                    f32x4 sacc = zero4();
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) {
                        const int jrow = (PERM && jt == NT - 1) ? 4 * (n & 3) + (n >> 2) : n;       // D row m <-> node perm(m), see jnode
                        const f32x4 xa = *reinterpret_cast<const f32x4*>(&Hs[(16 * jt + jrow) * XLD + 16 * ft + 4 * q]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc = mfma4(xa[r], gt_[ct][ft][r], sacc);
                    }
                    pr[ct][jt] = sacc;
                }
            }
            if (SK == 2) {
                // cosine family (graph_model.py:70-79): C_ij = S_ij / (m_i m_j), m_i = |S_i,:|_2 (rows of S itself).  Row norm of
                // column i: over my registers and the four q-groups; m_j of the other index goes through the padding column
                // 32 of the node-feature rows (XLD = 36).  Padded nodes get 1/m = 0: their rows and columns stay exactly 0.
                load_fence();
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    float z = 0.f;
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) z = fmaf(pr[ct][jt][r], pr[ct][jt][r], z);
                    z = kgroups_sum(z);
                    const float im = (16 * (ct + ctb) + n < N && z > 0.f) ? 1.f / sqrtf(z) : 0.f;
                    if (q == 0) Hs[(16 * (ct + ctb) + n) * XLD + 32] = im;
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) pr[ct][jt][r] *= im;
                }
                if (SPLIT) __syncthreads(); else __builtin_amdgcn_wave_barrier();      // every column's 1/m is in place
                load_fence();
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float imj = Hs[jnode(jt, r) * XLD + 32];
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) pr[ct][jt][r] *= imj;
                    }
                if (sim == SIM_COSINE) return;                   // the cosine matrix itself is the adjacency (not normalised)
            }
            // row normalisation: softmax over j (kept in B-operand order), or the plain weights / their row sums
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float mx = -INFINITY;
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = jnode(jt, r);
                        float v = pr[ct][jt][r];
                        if (SK == 1) v = plain_weight(sim, v, 16 * (ct + ctb) + n, j);
                        if (j >= N) v = SK == 1 ? 0.f : -INFINITY;
                        mx = fmaxf(mx, v);
                        pr[ct][jt][r] = v;
                    }
                mx = kgroups_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (SK != 1) pr[ct][jt][r] = __expf(pr[ct][jt][r] - mx);
                        sum += pr[ct][jt][r];
                    }
                sum = kgroups_sum(sum);
                const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[ct][jt][r] *= inv;
            }
        };
        if (!a.layerwise) adjacency();
        // layers: H <- relu((A H) W_l) (+ H); every column tile's A*H is taken before any row is overwritten
        for (int l = 0; l < a.L; ++l) {
            if (a.layerwise) adjacency();
            const bool last = (l == a.L - 1);
            const bool rows_only = last && a.rows_out != nullptr;     // value rows: only (A H)[robot] of the last layer is needed
            f32x4 acc[NT][2];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                acc[ct][0] = zero4();
                acc[ct][1] = zero4();
            }
            {
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                load_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (PERM && jt == NT - 1 && r >= last_steps) continue;      // k steps over padding nodes only
                    const float a0 = Hs[jnode(jt, r) * XLD + n];
                    const float a1 = Hs[jnode(jt, r) * XLD + 16 + n];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        if (rows_only && ct + ctb > 0) continue;
                        acc[ct][0] = mfma4(a0, pr[ct][jt][r], acc[ct][0]);
                        acc[ct][1] = mfma4(a1, pr[ct][jt][r], acc[ct][1]);
                    }
                }
            }
            }
            if (SPLIT) __syncthreads();      // every wave of the scene has taken its A*H: rows may be overwritten
            if (rows_only) {
                // hand-off row of stage 2 (robot_head_kernel): [ (A H_{L-1})[robot] | H_{L-1}[robot] ]; column 0 of tile 0 = robot
                if (active && ctb == 0) {
                    float* out = a.rows_out + (size_t)sc * 64;
                    if (n == 0) {
                        *reinterpret_cast<f32x4*>(out + 4 * q) = acc[0][0];
                        *reinterpret_cast<f32x4*>(out + 16 + 4 * q) = acc[0][1];
                    }
                    if (lane < 8) *reinterpret_cast<f32x4*>(out + 32 + 4 * lane) = *reinterpret_cast<const f32x4*>(&Hs[4 * lane]);
                }
                break;
            }
            const float* wl = ws + l * a.ws_stride;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                load_fence();
                f32x4 o[2] = {zero4(), zero4()};
                if constexpr (BX) {
                    layer_mfma_b6<XD, XD, false>(wl, acc[ct], o, lane);
                } else {
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int ot = 0; ot < 2; ++ot)
                            o[ot] = mfma4(wl[(16 * ft + 4 * q + r) * WLD + 16 * ot + n], acc[ct][ft][r], o[ot]);
                }
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const f32x4 sk = *reinterpret_cast<const f32x4*>(&Hs[(16 * (ct + ctb) + n) * XLD + 16 * ot + 4 * q]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float hv = fmaxf(o[ot][r], 0.f);
                        if (a.skip) hv += sk[r];
                        o[ot][r] = 16 * (ct + ctb) + n < N ? hv : 0.f;          // padded node rows stay exactly zero (a softmax row of a
                    }                                                   // padded node is uniform, not zero; layerwise graphs re-read H)
                    if (!last) *reinterpret_cast<f32x4*>(&Hs[(16 * (ct + ctb) + n) * XLD + 16 * ot + 4 * q]) = o[ot];
                }
                if (last) {
                    // motion head on this tile's columns, straight from registers: 32 -> 64 (ReLU) -> 5
                    f32x4 hm[4] = {zero4(), zero4(), zero4(), zero4()};
                    f32x4 om = zero4();
                    if constexpr (BX) {
                        layer_mfma_b6<XD, HID, true>(wm1, o, hm, lane, bm1);
#pragma unroll
                        for (int ht = 0; ht < 4; ++ht)
#pragma unroll
                            for (int r = 0; r < 4; ++r) hm[ht][r] = fmaxf(hm[ht][r], 0.f);
                        f32x4 om1[1];
                        layer_mfma_b6<HID, 16, false>(wm2, hm, om1, lane);
                        om = om1[0];
                    } else {
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot) {
                        load_fence();
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int ht = 0; ht < 4; ++ht)
                                hm[ht] = mfma4(wm1[(16 * ot + 4 * q + r) * W1LD + 16 * ht + n], o[ot][r], hm[ht]);
                    }
#pragma unroll
                    for (int ht = 0; ht < 4; ++ht) {
                        load_fence();
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(&bm1[16 * ht + 4 * q]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float hv = fmaxf(hm[ht][r] + bb[r], 0.f);
                            om = mfma4(wm2[(16 * ht + 4 * q + r) * M2LD + n], hv, om);
                        }
                    }
                    }
                    const int node = 16 * (ct + ctb) + n;
                    if (active && node >= 1 && node < N) {
                        float* dst = a.humans_next + ((size_t)sc * H + (node - 1)) * 5;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int oidx = 4 * q + r;
                            if (oidx < 5) dst[oidx] = om[r] + bm2[oidx];
                        }
                    }
                }
            }
            if (SPLIT) __syncthreads(); else __builtin_amdgcn_wave_barrier();      // the next layer / adjacency reads the rows written above
        }
        if (SPLIT) __syncthreads();      // the slot's rows are free for the next scene
    }
}

// ---- weight image of the scene kernel: one layout for the LDS region and (BX) for its packed global copy ---------------------------
struct SceneImageLayout { int off_wa, off_ws, off_wm1, off_bm1, off_wm2, off_bm2, total, ws_stride; };
// bx: the matrices as three-piece bf16 fragments (layer_mfma_b6: 6 bytes per weight) instead of k-major f32 rows
inline SceneImageLayout scene_image_layout(int L, bool bx = false) {
    SceneImageLayout o;
    int off = 0;
    auto take = [&](int nfl) { int r = off; off += (nfl + 3) & ~3; return r; };
    o.ws_stride = bx ? B6Floats<XD, XD>::v : XD * WLD;
    o.off_wa = take(bx ? B6Floats<XD, XD>::v : XD * WLD);
    o.off_ws = take(L * o.ws_stride);
    o.off_wm1 = take(bx ? B6Floats<XD, HID>::v : XD * W1LD);
    o.off_bm1 = take(HID);
    o.off_wm2 = take(bx ? B6Floats<HID, 16>::v : HID * M2LD);
    o.off_bm2 = take(16);
    o.total = off;
    return o;
}

struct SceneImageArgs {
    const float* wa;                      // null: identity (gaussian)
    const float* Ws[4];
    const float *wm1, *bm1, *wm2, *bm2;   // null: no motion head (value rows)
    int L;
    SceneImageLayout lo;
};

// the BX image: three-piece bf16 fragments of Wa, W_l, wm1, wm2 (no scales: bf16 has f32's exponent range) + the two bias vectors
__global__ __launch_bounds__(256) void scene_pack_b6_kernel(const SceneImageArgs a, float* img) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const SceneImageLayout& lo = a.lo;
    if (e >= lo.total) return;
    float v = 0.f;
    if (e < lo.off_ws) {
        const int i = e - lo.off_wa;
        if (i < B6Floats<XD, XD>::v) {
            if (a.wa) v = frag_bf3_ld<XD, XD>(a.wa, XD, XD, i);
            else {                                              // gaussian: Wa = I -> hi piece 1 on the diagonal
                const int u = i >> 2, p = i & 3, l = u & 63, rest = u >> 6, pc = rest % 3, ot = rest / 3;
                bf16x2 d;
                for (int k = 0; k < 2; ++k) {
                    const int ee = 2 * p + k, in = 16 * (ee >> 2) + 4 * (l >> 4) + (ee & 3), out = 16 * ot + (l & 15);
                    d[k] = (__bf16)((pc == 0 && in == out) ? 1.f : 0.f);
                }
                v = __builtin_bit_cast(float, d);
            }
        }
    } else if (e < lo.off_wm1) {
        const int i = e - lo.off_ws, l = i / lo.ws_stride, j = i - l * lo.ws_stride;
        if (l < a.L && j < B6Floats<XD, XD>::v) v = frag_bf3_ld<XD, XD>(a.Ws[l], XD, XD, j);
    } else if (e < lo.off_bm1) {
        const int i = e - lo.off_wm1;
        if (a.wm1 && i < B6Floats<XD, HID>::v) v = frag_bf3_ld<XD, HID>(a.wm1, HID, HID, i);
    } else if (e < lo.off_wm2) {
        const int i = e - lo.off_bm1;
        if (a.bm1 && i < HID) v = a.bm1[i];
    } else if (e < lo.off_bm2) {
        const int i = e - lo.off_wm2;
        if (a.wm2 && i < B6Floats<HID, 16>::v) v = frag_bf3_ld<HID, 16>(a.wm2, 5, 5, i);
    } else {
        const int i = e - lo.off_bm2;
        if (a.bm2 && i < 5) v = a.bm2[i];
    }
    img[e] = v;
}

inline RowMlpArgs row_mlp_args(const RglMlp& m, const float* rows, float* out, int M) {
    RowMlpArgs ra;
    ra.w1 = m.weight[0]; ra.b1 = m.bias[0]; ra.w2 = m.weight[1]; ra.b2 = m.bias[1];
    ra.rows = rows; ra.out = out; ra.M = M; ra.n_tiles = (M + 15) / 16;
    return ra;
}

// robot rows [Ma][9] -> [Ma][32] with w_r, human rows [Mb][5] -> [Mb][32] with w_h
// `children` (optional): the level's reward / next-state work on extra workgroups of this launch
inline int launch_row_mlp2_pair(const RglMlp& wr, const float* robot_rows, float* x0_out, int Ma, const RglMlp& wh,
                                const float* human_rows, float* xh_out, int Mb, hipStream_t st, const ChildrenArgs* children = nullptr) {
    const RowMlpArgs ra = row_mlp_args(wr, robot_rows, x0_out, Ma), rb = row_mlp_args(wh, human_rows, xh_out, Mb);
    int grid_a = (ra.n_tiles + kWaves - 1) / kWaves, grid_b = (rb.n_tiles + kWaves - 1) / kWaves;
    if (grid_a > 256) grid_a = 256;
    if (grid_b > 1024) grid_b = 1024;
    ChildrenArgs ca{};
    int grid_c = 0;
    if (children) {
        ca = *children;
        const long long blocks = ((long long)ca.P * ca.A + kThreads - 1) / kThreads;
        grid_c = (int)(blocks < 4096 ? blocks : 4096);
    }
    const int grid_mlp = grid_a + grid_b;
    if (wr.dims[0] == 9)
        hipLaunchKernelGGL((row_mlp2_pair_kernel<9, 5>), dim3(grid_mlp + grid_c), dim3(kThreads), kRowMlpSetFloats * sizeof(float), st, ra,
                           rb, grid_a, grid_mlp, ca);
    else
        hipLaunchKernelGGL((row_mlp2_pair_kernel<6, 7>), dim3(grid_mlp + grid_c), dim3(kThreads), kRowMlpSetFloats * sizeof(float), st, ra,
                           rb, grid_a, grid_mlp, ca);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

// Split scenes (NT waves per scene) below this many scenes: env RGL_SCENE_SPLIT_BELOW overrides (0 = never, tests / measurements)
inline int scene_split_below(int nt) {
    static const int env = [] { const char* e = getenv("RGL_SCENE_SPLIT_BELOW"); return e ? atoi(e) : -1; }();
    if (env >= 0) return env;
    return nt == 2 ? 3072 : 4096;
}

template <int NT, int SK, int WAVES, bool SPLIT = false, bool EMB = false, bool BX = false>
int launch_scene_k(const SceneArgs& sa, size_t lds_bytes, const ChildrenArgs* children, hipStream_t st) {
    if constexpr (!SPLIT && SK != 3 && (NT == 2 || NT == 4)) {
        if (sa.P < scene_split_below(NT)) return launch_scene_k<NT, SK, 8, true, EMB, BX>(sa, lds_bytes, children, st);
    }
    constexpr int kSlots = SPLIT ? WAVES / NT : WAVES;
    int grid = (sa.P + kSlots - 1) / kSlots;
    const int cap = 256 * (lds_bytes * 2 <= (size_t)rgl::kLdsBytesPerCu ? 2 : 1) * (WAVES == 4 ? 2 : 1);
    if (grid > cap) grid = cap;
    ChildrenArgs ca{};
    int grid_children = 0;
    if (children) {
        ca = *children;
        const long long blocks = ((long long)ca.P * ca.A + WAVES * 64 - 1) / (WAVES * 64);
        grid_children = (int)(blocks < 2048 ? blocks : 2048);
    }
    auto kern = children ? scene_graph_kernel<NT, SK, WAVES, true, SPLIT, EMB, BX> : scene_graph_kernel<NT, SK, WAVES, false, SPLIT, EMB, BX>;
    if (lds_bytes > 64 * 1024)
        RGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(grid + grid_children), dim3(WAVES * 64), lds_bytes, st, sa, ca, grid);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

// 64 < N <= 128: eight column tiles, always split over the eight waves of a workgroup (one scene per workgroup pass; the
// unsplit form would hold an 8 x 8 block of adjacency tiles per wave: 256 VGPRs).  The pair-MLP similarity has no split form.
inline int launch_scene_wide(const SceneArgs& sa, size_t lds_bytes, const ChildrenArgs* children, hipStream_t st) {
    if (sa.sim == SIM_SOFTMAX) return launch_scene_k<8, 0, 8, true>(sa, lds_bytes, children, st);
    if (sa.sim == SIM_COSINE || sa.sim == SIM_COSINE_SOFTMAX) return launch_scene_k<8, 2, 8, true>(sa, lds_bytes, children, st);
    if (sa.sim == SIM_CONCAT) return 1;
    return launch_scene_k<8, 1, 8, true>(sa, lds_bytes, children, st);
}

template <int NT, int WAVES>
int launch_scene(const SceneArgs& sa, size_t lds_bytes, const ChildrenArgs* children, hipStream_t st) {
    if constexpr (NT == 2) {
        if (sa.bx)                                       // six-term bf16 weight products (softmax similarity, two node tiles)
            return sa.robot_rows ? launch_scene_k<2, 0, 8, true, true, true>(sa, lds_bytes, children, st)
                                 : launch_scene_k<2, 0, WAVES, false, false, true>(sa, lds_bytes, children, st);
    }
    if constexpr (NT == 1) {
        if (sa.robot_rows) return launch_scene_k<1, 0, 8, false, true>(sa, lds_bytes, children, st);           // embeddings inside
    }
    if constexpr (NT == 2 || NT == 4) {
        if (sa.robot_rows) return launch_scene_k<NT, 0, 8, true, true>(sa, lds_bytes, children, st);          // ... and split scenes
    }
    if (sa.sim == SIM_SOFTMAX) return launch_scene_k<NT, 0, WAVES>(sa, lds_bytes, children, st);
    if (sa.sim == SIM_COSINE || sa.sim == SIM_COSINE_SOFTMAX) return launch_scene_k<NT, 2, WAVES>(sa, lds_bytes, children, st);
    if (sa.sim == SIM_CONCAT) return launch_scene_k<NT, 3, WAVES>(sa, lds_bytes, children, st);
    return launch_scene_k<NT, 1, WAVES>(sa, lds_bytes, children, st);
}

}  // namespace

namespace {

static bool scene_kernel_covers(const RglGraph& g, int N) {
    const bool path_m = mlp_is(g.w_r, 9, HID, XD, true) && mlp_is(g.w_h, 5, HID, XD, true);
    const bool path_g = mlp_is(g.w_r, 6, HID, XD, true) && mlp_is(g.w_h, 7, HID, XD, true);        // gcn.ValueNetwork's inputs
    return fast_path_enabled() && scene_similarity_mode(g) >= 0 && g.x_dim == XD && g.num_layer >= 1 && g.num_layer <= 4 &&
           (path_m || path_g) && N <= 128 && (N <= 64 || scene_similarity_mode(g) != SIM_CONCAT);
}

// embeddings (one launch) + one-wave-per-scene graph forward; mh != null: motion head -> humans_next; rows_out != null: value rows
static int run_scene_kernels(const RglGraph& g, const RglMlp* mh, const float* robot, const float* humans, int crowds_per, int P,
                             int H, float* humans_next, float* rows_out, float* x0_rows, float* xh_rows,
                             const ChildrenArgs* ca, hipStream_t stream, const ChildrenArgs* embed_children = nullptr,
                             const float* sp_image = nullptr) {
    const int N = H + 1, n_crowds = P / crowds_per;
    const int NT0 = N > 64 ? 8 : (N + 15) / 16;
    // few scenes of the shipped shape: the scene kernel embeds its own node tiles (one launch for the level instead of two)
    static const bool emb_off = [] { const char* e = getenv("RGL_SCENE_EMBED_INSIDE"); return e && e[0] == '0'; }();
    const bool embed_inside = !emb_off && !embed_children && g.w_r.dims[0] == 9 && scene_similarity_mode(g) == SIM_SOFTMAX &&
                              (NT0 == 1 || NT0 == 2 || NT0 == 4) && P <= 512 &&     // measured: +1.5-3 % up to 512 scenes, -2 % at 1-2 k (sibling scenes repeat their crowd's rows)
                              (NT0 == 1 || P < scene_split_below(NT0));       // the split form (see launch_scene)
    if (!embed_inside) {
        int rc = launch_row_mlp2_pair(g.w_r, robot, x0_rows, P, g.w_h, humans, xh_rows, n_crowds * H, stream, embed_children);
        if (rc) return rc;
    }
    SceneArgs sa;
    sa.robot_rows = embed_inside ? robot : nullptr;
    sa.human_rows = embed_inside ? humans : nullptr;
    sa.er_w1 = g.w_r.weight[0]; sa.er_b1 = g.w_r.bias[0]; sa.er_w2 = g.w_r.weight[1]; sa.er_b2 = g.w_r.bias[1];
    sa.eh_w1 = g.w_h.weight[0]; sa.eh_b1 = g.w_h.bias[0]; sa.eh_w2 = g.w_h.weight[1]; sa.eh_b2 = g.w_h.bias[1];
    sa.off_er = sa.off_eh = 0;
    // (one node tile: measured slower than the f32 form -- 0.0615 vs 0.0496 ms per configs[1] step: too few MFMAs to pay for the splits)
    const bool split_ok = sp_image && scene_similarity_mode(g) == SIM_SOFTMAX && NT0 == 2 && g.num_layer <= 4;      // (mh == null: value rows)
    sa.bx = split_ok ? 1 : 0;
    sa.image = sp_image;
    sa.xh_rows = xh_rows; sa.x0_rows = x0_rows; sa.crowds_per = crowds_per;
    sa.wa = bilinear_wa(g);
    sa.sim = scene_similarity_mode(g);
    sa.layerwise = g.layerwise_graph;
    for (int l = 0; l < RGL_MAX_GCN_LAYERS; ++l) sa.Ws[l] = l < g.num_layer ? g.Ws[l] : nullptr;
    sa.L = g.num_layer; sa.skip = g.skip_connection;
    sa.wm1 = mh ? mh->weight[0] : nullptr; sa.bm1 = mh ? mh->bias[0] : nullptr;
    sa.wm2 = mh ? mh->weight[1] : nullptr; sa.bm2 = mh ? mh->bias[1] : nullptr;
    sa.humans_next = humans_next;
    sa.rows_out = rows_out;
    sa.P = P; sa.H = H; sa.N = N;
    const int NT = N > 64 ? 8 : (N + 15) / 16;
    int off = 0;
    auto take = [&](int nfl) { int o = off; off += (nfl + 3) & ~3; return o; };
    {   // the weight image: one layout for the LDS region and for its packed global copy
        const SceneImageLayout lo = scene_image_layout(g.num_layer, sa.bx != 0);
        sa.off_wa = lo.off_wa; sa.off_ws = lo.off_ws; sa.off_wm1 = lo.off_wm1; sa.off_bm1 = lo.off_bm1;
        sa.off_wm2 = lo.off_wm2; sa.off_bm2 = lo.off_bm2;
        sa.ws_stride = lo.ws_stride;
        sa.image_floats = lo.total;
        off = lo.total;
    }
    sa.wc1 = sa.bc1 = sa.wc2 = sa.bc2 = nullptr;
    sa.off_wc1 = sa.off_bc1 = sa.off_wc2 = 0;
    if (sa.sim == SIM_CONCAT) {
        sa.wc1 = g.w_a_mlp.weight[0]; sa.bc1 = g.w_a_mlp.bias[0]; sa.wc2 = g.w_a_mlp.weight[1]; sa.bc2 = g.w_a_mlp.bias[1];
        sa.off_wc1 = take(2 * XD * W1LD); sa.off_bc1 = take(HID); sa.off_wc2 = take(HID);
    }
    if (embed_inside) {
        sa.off_er = take(kRowMlpSetFloats);
        sa.off_eh = take(kRowMlpSetFloats);
    }
    sa.wave_stride = 16 * NT * XLD;
    // scene slots of a workgroup: one per wave, or -- the split form, which launch_scene always takes with the embeddings inside --
    // one per NT waves of its 8 (with the unsplit count a 50-agent scene workgroup held 95 KB instead of 76: one per CU, and the
    // launch's reward workgroups, which reserve the same LDS, waited for a scene workgroup to retire)
    static const bool wide_slots = [] { const char* e = getenv("RGL_SCENE_WIDE_SLOTS"); return e && e[0] == '1'; }();      // measurements
    const int slots = (embed_inside && (NT == 2 || NT == 4) && !wide_slots) ? 8 / NT : (NT <= 2 ? 8 : (NT <= 4 ? 4 : 1));
    sa.off_wave = take(slots * sa.wave_stride);
    const size_t lds_bytes = (size_t)off * sizeof(float);
    switch (NT) {
        case 1: return launch_scene<1, 8>(sa, lds_bytes, ca, stream);
        case 2: return launch_scene<2, 8>(sa, lds_bytes, ca, stream);
        case 3: return launch_scene<3, 4>(sa, lds_bytes, ca, stream);
        case 4: return launch_scene<4, 4>(sa, lds_bytes, ca, stream);
        default: return launch_scene_wide(sa, lds_bytes, ca, stream);
    }
}

}  // namespace

namespace rgl {

// The three-piece bf16 weight image of the state predictor's scene kernel (RGL_CONTRACT_BF16X6): depends on the weights only.
// (graph + optional motion head: the state predictor's form; mh == null: the value-rows form of path G / the module forwards)
size_t scene_image_bytes_for(const RglGraph& g, const RglMlp* mh) {
    if (!scene_kernel_covers(g, 20) || scene_similarity_mode(g) != SIM_SOFTMAX) return 0;
    if (mh && !mlp_is(*mh, XD, HID, 5, false)) return 0;
    return (((size_t)scene_image_layout(g.num_layer, true).total * sizeof(float)) + 255) & ~(size_t)255;
}

size_t scene_image_bytes(const MprlPlanner* pl) {
    if (!pl || pl->linear_state_predictor) return 0;
    if (pl->contraction_dtype != RGL_CONTRACT_BF16X6) return 0;
    return scene_image_bytes_for(pl->predictor_graph, &pl->motion_head);
}

int pack_scene_image(const MprlPlanner* pl, float* image, hipStream_t stream) {
    if (!scene_image_bytes(pl)) return 1;
    return pack_scene_image_for(pl->predictor_graph, &pl->motion_head, image, stream);
}

int pack_scene_image_for(const RglGraph& g, const RglMlp* mh, float* image, hipStream_t stream) {
    if (!scene_image_bytes_for(g, mh)) return 1;
    SceneImageArgs ia;
    ia.wa = bilinear_wa(g);
    for (int l = 0; l < 4; ++l) ia.Ws[l] = l < g.num_layer ? g.Ws[l] : nullptr;
    ia.wm1 = mh ? mh->weight[0] : nullptr; ia.bm1 = mh ? mh->bias[0] : nullptr;
    ia.wm2 = mh ? mh->weight[1] : nullptr; ia.bm2 = mh ? mh->bias[1] : nullptr;
    ia.L = g.num_layer;
    ia.lo = scene_image_layout(g.num_layer, true);
    hipLaunchKernelGGL(scene_pack_b6_kernel, dim3((ia.lo.total + 255) / 256), dim3(256), 0, stream, ia, image);
    RGL_LAUNCH_CHECK();
    return RGL_OK;
}

// humans_next[s] = motion_head(RGL(robot[s], humans[s / crowds_per]))[1:]  for P scenes (StatePredictor.forward).
int launch_predict_humans(const MprlPlanner* pl, const float* robot, const float* humans, int crowds_per, int P, int H,
                          float* humans_next, void* workspace, size_t workspace_bytes, hipStream_t stream,
                          const void* children, size_t children_bytes, int* children_done, const float* sp_image) {
    const ChildrenArgs* ca = (children && children_bytes == sizeof(ChildrenArgs)) ? (const ChildrenArgs*)children : nullptr;
    if (children_done) *children_done = 0;
    const RglGraph& g = pl->predictor_graph;
    const RglMlp& mh = pl->motion_head;
    const int N = H + 1;
    const bool ok = scene_kernel_covers(g, N) && mlp_is(mh, XD, HID, 5, false) && workspace &&
                    workspace_bytes >= (size_t)P * N * XD * sizeof(float) && P % crowds_per == 0;
    if (!ok) {
        // outside the shipped shapes: the tile kernels (rgl_backward_mfma.hip) where they cover the model, else the general kernel
        if (P % crowds_per == 0) {
            const int rc = launch_tiles_forward(&g, nullptr, &mh, robot, humans, P, crowds_per, H, nullptr, nullptr, humans_next, workspace,
                                                workspace_bytes, stream);
            if (rc != 1) return rc;
        }
        const char* e = getenv("RGL_REQUIRE_MFMA_FORWARD");          // tests: refuse instead of running the general VALU kernel
        if (e && e[0] == '1') return RGL_ERR_BAD_MODE;
        return launch_generic_forward(&g, nullptr, &mh, robot, humans, P, crowds_per, H, nullptr, nullptr, nullptr,
                                      humans_next, stream);
    }
    float* x0_rows = (float*)workspace;                      // [P][32]
    float* xh_rows = x0_rows + (size_t)P * XD;               // [n_crowds][H][32]
    // the level's reward / next-state work rides in the scene kernel's launch while the scene workgroups leave LDS free (few scenes),
    // in the embedding launch otherwise: never a launch of its own on this path
    // cross-over: ~3 k scenes for the f32 scene kernel (round 2), ~1.1 k once its weight products run on the matrix pipe (round 5:
    // 2048 roots 0.3022 -> 0.2988 ms, 1024 roots 0.1773 -> 0.1759, 512 roots unchanged); RGL_SCENE_CHILDREN_BELOW overrides (measurements)
    static const int below_env = [] { const char* e = getenv("RGL_SCENE_CHILDREN_BELOW"); return e ? atoi(e) : -1; }();
    const int children_in_scene_below = below_env >= 0 ? below_env : (pl->contraction_dtype == RGL_CONTRACT_BF16X6 ? 1100 : 3072);
    const ChildrenArgs* in_scene = (ca && P < children_in_scene_below) ? ca : nullptr;
    const ChildrenArgs* in_embed = (ca && !in_scene) ? ca : nullptr;
    const int rc = run_scene_kernels(g, &mh, robot, humans, crowds_per, P, H, humans_next, nullptr, x0_rows, xh_rows, in_scene, stream,
                                     in_embed, pl->contraction_dtype == RGL_CONTRACT_BF16X6 ? sp_image : nullptr);
    if (rc == RGL_OK && ca && children_done) *children_done = 1;
    return rc;
}

// Module forwards (ValueEstimator.forward / StatePredictor.forward on a batch: value_estimator.py:11-20, state_predictor.py:20-39)
// through the same kernels: embeddings + one wave per scene; values via the value-rows mode + robot_head_kernel.
// workspace: x0 [S][32] | xh [S / crowds_per][H][32] | rows [S][64] (value head only).
static bool scene_forward_covers(const RglGraph& g, const RglMlp* vh, const RglMlp* mh, int S, int crowds_per, int H) {
    const bool has_v = vh && vh->n_layers > 0, has_m = mh && mh->n_layers > 0;
    if (!(has_v || has_m) || crowds_per < 1 || S % crowds_per != 0) return false;
    if (!scene_kernel_covers(g, H + 1)) return false;
    if (has_v && head_variant(*vh) < 0) return false;
    if (has_m && !mlp_is(*mh, XD, HID, 5, false)) return false;
    return true;
}

size_t scene_forward_workspace_bytes(const RglGraph* g, const RglMlp* vh, const RglMlp* mh, int S, int crowds_per, int H) {
    if (!g || !scene_forward_covers(*g, vh, mh, S, crowds_per, H)) return 0;
    const bool has_v = vh && vh->n_layers > 0;
    return ((size_t)S * XD + (size_t)(S / crowds_per) * H * XD + (has_v ? (size_t)S * 64 : 0)) * sizeof(float);
}

int launch_scene_forward(const RglGraph* g, const RglMlp* vh, const RglMlp* mh, const float* robot, const float* humans, int S,
                         int crowds_per, int H, float* value_out, float* humans_next, void* workspace, size_t workspace_bytes,
                         hipStream_t stream, const float* value_rows_image) {
    if (!scene_forward_covers(*g, vh, mh, S, crowds_per, H)) return 1;
    if (!workspace || workspace_bytes < scene_forward_workspace_bytes(g, vh, mh, S, crowds_per, H)) return 1;
    const bool has_v = vh && vh->n_layers > 0, has_m = mh && mh->n_layers > 0;
    float* x0_rows = (float*)workspace;
    float* xh_rows = x0_rows + (size_t)S * XD;
    float* rows = xh_rows + (size_t)(S / crowds_per) * H * XD;
    if (has_m) {
        int rc = run_scene_kernels(*g, mh, robot, humans, crowds_per, S, H, humans_next, nullptr, x0_rows, xh_rows, nullptr, stream);
        if (rc) return rc;
    }
    if (has_v) {
        int rc = run_scene_kernels(*g, nullptr, robot, humans, crowds_per, S, H, nullptr, rows, x0_rows, xh_rows, nullptr, stream, nullptr,
                                   value_rows_image);
        if (rc) return rc;
        return launch_head_rows(g, vh, rows, S, value_out, stream);
    }
    return RGL_OK;
}

// Value of the children through the one-wave-per-scene kernel: every child's graph in full (no crowd sharing), any of the
// similarity functions it implements, layerwise graphs, 1..4 layers -- the MFMA path of everything the shared-crowd kernels do
// not cover (cosine / cosine_softmax scale the columns by child-dependent norms; layerwise graphs rebuild the adjacency from
// every H_l).  workspace: x0 [P*A][32] | xh [P*H][32] | rows [P*A][64].  1 = outside this kernel's envelope.
size_t scene_children_workspace_bytes(int P, int A, int H) {
    return ((size_t)P * A * (XD + 64) + (size_t)P * H * XD) * sizeof(float);
}

int launch_scene_children(const MprlPlanner* pl, const float* child_robot, const float* humans_next, int P, int H,
                          float* child_value, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const RglGraph& g = pl->value_graph;
    const int A = pl->num_actions, N = H + 1;
    if (!scene_kernel_covers(g, N) || head_variant(pl->value_head) < 0 || !workspace ||
        workspace_bytes < scene_children_workspace_bytes(P, A, H))
        return 1;
    float* x0_rows = (float*)workspace;                      // [P*A][32]
    float* xh_rows = x0_rows + (size_t)P * A * XD;           // [P][H][32]
    float* rows = xh_rows + (size_t)P * H * XD;              // [P*A][64]
    int rc = run_scene_kernels(g, nullptr, child_robot, humans_next, A, P * A, H, nullptr, rows, x0_rows, xh_rows, nullptr, stream);
    if (rc) return rc;
    return launch_head_rows(&g, &pl->value_head, rows, P * A, child_value, stream);
}

}  // namespace rgl
