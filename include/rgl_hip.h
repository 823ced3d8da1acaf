/*
 * rgl_hip.h -- C ABI of librgl_hip.so: the MI355X (gfx950) implementation of the RGL
 * relational-graph forward pass and the model-predictive rollout that calls it.
 *
 * The upstream project is pure Python + torch; it has no FFI of its own.  Each entry point
 * below therefore replaces a *Python* interface of the reference (cited per function, paths
 * relative to the upstream repo), and INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add to call it.
 *
 * Conventions (all entry points):
 *   - every `const float*` / `float*` / `int*` / `double*` argument that is documented as
 *     "device" is a DEVICE pointer owned by the caller (e.g. a contiguous fp32 torch tensor);
 *     the library never allocates, frees or retains device memory;
 *   - descriptor structs (RglMlp, RglGraph, MprlPlanner) live in HOST memory and are read
 *     during the call only; the device pointers inside them must stay valid until the work
 *     queued on `stream` has finished;
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream); no call synchronises the device, so all of them may be captured into a hipGraph
 *     (the one exception, by purpose: mprl_tree_search_traced_f32, a measurement call);
 *   - return value: 0 on success, a positive hipError_t if the runtime reported one, or one of
 *     the negative RGL_ERR_* codes below.  No exceptions cross this boundary;
 *   - thread safety: calls on distinct streams may run concurrently; there is no hidden
 *     global state.
 *
 * Matrix layouts:
 *   - MLP layer l maps dims[l] -> dims[l+1];  weight[l] is row-major [dims[l]][dims[l+1]]
 *     ("k-major", i.e. torch `linear.weight.t().contiguous()`); rgl_transpose_f32 produces it
 *     on device from the torch (out,in) layout.  bias[l] has dims[l+1] entries.
 *   - w_a and Ws[l] are row-major [x_dim][x_dim] exactly as the reference stores them
 *     (they multiply from the right: X @ w_a, (A H) @ Ws[l]).
 *   - agent states: robot rows are 9 floats [px,py,vx,vy,radius,gx,gy,v_pref,theta], human rows
 *     5 floats [px,py,vx,vy,radius] (crowd_sim/envs/utils/state.py:27-28,51-52).
 */
#ifndef RGL_HIP_H
#define RGL_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGL_ABI_VERSION 8

#define RGL_MAX_MLP_LAYERS 6
#define RGL_MAX_GCN_LAYERS 8
#define RGL_MAX_NODES 128   /* N = humans + 1 (64 until ABI 3) */
#define RGL_MAX_XDIM 64
#define RGL_MAX_WIDTH 256   /* widest MLP layer */
#define RGL_MAX_ACTIONS 256

#define RGL_OK 0
#define RGL_ERR_BAD_SHAPE (-1)      /* N, x_dim, widths or counts outside the limits above     */
#define RGL_ERR_BAD_MODE (-2)       /* unknown similarity / kinematics / inconsistent flags    */
#define RGL_ERR_NULL (-3)           /* a required pointer is NULL                              */
#define RGL_ERR_WORKSPACE (-4)      /* workspace too small (see mprl_tree_workspace_bytes)     */
#define RGL_CONTRACT_F32 0
#define RGL_CONTRACT_F16 1
/* 2 was RGL_CONTRACT_F16X3 (ABI 4..7: 22-23 operand bits as two f16 halves); superseded by RGL_CONTRACT_BF16X6 and removed in ABI 8:
 * a planner that asks for it gets RGL_ERR_BAD_MODE */
#define RGL_CONTRACT_BF16X6 3       /* ABI 6: f32-WIDTH products on the matrix pipe: three bf16 pieces per operand, six terms */

#define RGL_ERR_LDS (-5)            /* configuration does not fit the 160 KiB LDS of one CU    */

typedef void* rgl_stream_t;         /* hipStream_t */

/* similarity functions of RGL.compute_similarity_matrix (crowd_nav/policy/graph_model.py:63-97) */
enum {
    RGL_SIM_EMBEDDED_GAUSSIAN = 0,
    RGL_SIM_GAUSSIAN = 1,
    RGL_SIM_COSINE = 2,
    RGL_SIM_COSINE_SOFTMAX = 3,
    RGL_SIM_CONCATENATION = 4,
    RGL_SIM_SQUARED = 5,
    RGL_SIM_EQUAL_ATTENTION = 6,
    RGL_SIM_DIAGONAL = 7
};

enum { RGL_HOLONOMIC = 0, RGL_UNICYCLE = 1 };

/* Sequential[Linear, ReLU, ...]  (crowd_nav/policy/helpers.py:5-13) */
typedef struct RglMlp {
    int n_layers;                              /* 0 = absent                                   */
    int last_relu;                             /* ReLU after the last layer too                */
    int dims[RGL_MAX_MLP_LAYERS + 1];
    const float* weight[RGL_MAX_MLP_LAYERS];   /* device, [dims[l]][dims[l+1]]                 */
    const float* bias[RGL_MAX_MLP_LAYERS];     /* device, [dims[l+1]]                          */
} RglMlp;

/* parameters + flags of one relational graph model
 * (RGL.__init__, crowd_nav/policy/graph_model.py:11-61; gcn.ValueNetwork.__init__, gcn.py:11-47) */
typedef struct RglGraph {
    RglMlp w_r;                 /* robot/self-state embedding, dims[0] = robot_dim (9 | 6)     */
    RglMlp w_h;                 /* human embedding, dims[0] = human_dim (5 | 7)                */
    int x_dim;
    int num_layer;
    int similarity;             /* RGL_SIM_*                                                   */
    int layerwise_graph;
    int skip_connection;
    int reserved;
    const float* w_a;           /* device [x_dim][x_dim]; embedded_gaussian only               */
    RglMlp w_a_mlp;             /* concatenation only: 2*x_dim -> hidden -> 1, last_relu       */
    const float* Ws[RGL_MAX_GCN_LAYERS];   /* device, each [x_dim][x_dim]                      */
} RglGraph;

/* ---------------------------------------------------------------------------------------------
 * rgl_graph_forward_f32 -- batched relational-graph forward with optional heads.
 * Replaces: RGL.forward (graph_model.py:99-130), ValueEstimator.forward (value_estimator.py:11-20),
 *           the graph+motion part of StatePredictor.forward (state_predictor.py:20-39) and the
 *           network part of gcn.ValueNetwork.forward (gcn.py:95-128).
 *   robot      device [n_scenes][robot_dim]
 *   humans     device [n_scenes / scenes_per_crowd][H][human_dim]; scene s reads crowd
 *              s / scenes_per_crowd (sibling scenes of a rollout share their humans)
 *   value_head NULL or the MLP applied to node 0 of the last layer  -> value_out[n_scenes]
 *   motion_head NULL or the MLP applied to every node; rows 1..H    -> humans_next[n_scenes][H][out]
 *   H_out      NULL or device [n_scenes][N][x_dim]   (last-layer node features)
 *   A_out      NULL or device [n_scenes][N][N]       (first adjacency computed)
 *   workspace  NULL or device scratch of rgl_graph_forward_workspace_bytes(...) bytes (ABI 3).  With it, and with H_out = A_out =
 *              NULL, models the one-wave-per-scene MFMA kernel covers (the shipped path-M shapes: w_r 9-64-32, w_h 5-64-32,
 *              x_dim 32, <= 4 layers, N <= 128 (concatenation: N <= 64), every similarity function, layerwise graphs) run on
 *              it; other embedding MLPs and x_dim = 64 (embedded_gaussian / gaussian, one adjacency, 1-3 layers, N <= 64) run on
 *              the MFMA tile kernels (rgl_backward_mfma.hip); everything else, and every call without workspace, runs on the
 *              general kernel -- same numbers up to summation order.
 * Limits: N = H+1 <= RGL_MAX_NODES, x_dim <= RGL_MAX_XDIM, widths <= RGL_MAX_WIDTH.
 * ------------------------------------------------------------------------------------------- */
size_t rgl_graph_forward_workspace_bytes(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head,
                                         int n_scenes, int scenes_per_crowd, int H);   /* 0: no MFMA path for this call */
int rgl_graph_forward_f32(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head,
                          const float* robot, const float* humans,
                          int n_scenes, int scenes_per_crowd, int H,
                          float* H_out, float* A_out, float* value_out, float* humans_next,
                          void* workspace, size_t workspace_bytes, rgl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * rgl_graph_backward_f32 -- gradients of rgl_graph_forward_f32's outputs with respect to every
 * parameter, summed over the n_scenes scenes (deterministic: per-scene slabs reduced in scene order).
 * Replaces: torch autograd through RGL.forward / ValueEstimator.forward / StatePredictor.forward /
 * gcn.ValueNetwork.forward as driven by MPRLTrainer / VNRLTrainer (crowd_nav/utils/trainer.py:110-161,
 * 199-250).  Supported: all eight similarity functions, layerwise_graph 0 | 1.  Two implementations behind this entry point:
 * a per-scene kernel (one workgroup and one gradient slab per scene; RGL_ERR_LDS when a scene's activations exceed the 160 KB
 * LDS of a CU: N = 64 with deep MLPs) and, for embedded_gaussian / gaussian with one adjacency, x_dim 32 | 64, 1-3 layers,
 * N <= 64, a pipeline of MFMA tile kernels (rgl_backward_mfma.hip) -- taken from 256 scenes, whenever `stream` is being
 * captured into a hipGraph, and where the per-scene kernel does not fit.  Environment: RGL_BACKWARD_MFMA = 0 | 1 forces a
 * path (2: RGL_ERR_BAD_MODE instead of the per-scene kernel where the pipeline does not apply), RGL_BACKWARD_MFMA_MIN moves the
 * threshold.  Both are deterministic (fixed summation order per shape) -- but they are TWO summation orders: the same call below the
 * threshold runs the per-scene kernel eagerly and the tile pipeline while `stream` is being captured, so an eager step and its
 * captured replay agree to rounding (~4e-8 relative on the shipped shapes), not bit for bit.  Force one path with
 * RGL_BACKWARD_MFMA where bitwise equality between the two forms is wanted (the switches are read on every call).
 *   d_value [n_scenes], d_humans_next [n_scenes][H][out], d_H [n_scenes][N][x_dim]: upstream gradients of the
 *   corresponding forward outputs (device; NULL = zero).  detach_graph = 1 reproduces
 *   StatePredictor(..., detach=True): only the heads receive gradients.
 *   grad_out device [rgl_graph_param_count()]: w_r (W0,b0,W1,b1,..), w_h, w_a (embedded_gaussian: the matrix;
 *   concatenation: its pair MLP W0,b0,W1,b1), Ws[0..L-1], value head, motion head; Linear weight gradients in
 *   torch's nn.Linear layout [out][in] (ABI 3; the forward's RglMlp weights stay k-major [in][out]).
 *   workspace device, >= rgl_graph_backward_workspace_bytes().  Each scene has its own crowd here
 *   (scenes_per_crowd = 1: the training batches are independent transitions).
 * ------------------------------------------------------------------------------------------- */
int rgl_graph_param_count(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head);
size_t rgl_graph_backward_workspace_bytes(const RglGraph* graph, const RglMlp* value_head,
                                          const RglMlp* motion_head, int n_scenes);
int rgl_graph_backward_f32(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head,
                           const float* robot, const float* humans, int n_scenes, int H, int detach_graph,
                           const float* d_value, const float* d_humans_next, const float* d_H,
                           float* grad_out, void* workspace, size_t workspace_bytes, rgl_stream_t stream);

/* rgl_transpose_f32 -- dst[c][r] = src[r][c]; turns a torch Linear weight (out,in) into the
 * k-major layout RglMlp wants.  Host-side convenience of this ABI (no reference counterpart). */
int rgl_transpose_f32(const float* src, float* dst, int rows, int cols, rgl_stream_t stream);

/* rgl_transpose_many_f32 (ABI 4) -- the same for a LIST of matrices in one launch per 32 jobs: a module's descriptor holds
 * half a dozen Linear weights and is rebuilt after every optimizer step (crowd_nav/utils/trainer.py:110-161 steps two Adam
 * optimizers per batch), so one launch per weight was most of a training step's launch count.  `jobs` is a HOST array. */
typedef struct RglTransposeJob {
    const float* src;           /* device [rows][cols] */
    float* dst;                 /* device [cols][rows] */
    int rows, cols;
} RglTransposeJob;
int rgl_transpose_many_f32(const RglTransposeJob* jobs, int n_jobs, rgl_stream_t stream);

/* rgl_gather_rows_f32 (ABI 7) -- dst_j[i][:] = src_j[index[i]][:] for every field j of a replay memory in one launch per 8 fields:
 * the batch a trainer draws (crowd_nav/utils/trainer.py:120 `for data in self.data_loader`; crowd_nav/utils/memory.py keeps tuples,
 * this package's ReplayMemory mirrors them as one [capacity][row_floats] device array per field).  `jobs` is a HOST array; `index`
 * device int64 [n_index], every entry in [0, src_rows) -- a row of NaN is written for an entry outside (torch's index_select traps). */
typedef struct RglGatherJob {
    const float* src;           /* device [src_rows][row_floats] */
    float* dst;                 /* device [n_index][row_floats]  */
    int row_floats, src_rows;
} RglGatherJob;
int rgl_gather_rows_f32(const RglGatherJob* jobs, int n_jobs, const long long* index, int n_index, rgl_stream_t stream);

/* rgl_mse_step_f32 (ABI 7) -- the loss end of one optimisation step (crowd_nav/utils/trainer.py:130-137,145-153:
 * `loss = self.criterion(outputs, target_values)` with nn.MSELoss(), `loss.backward()`, `v_losses += loss.data.item()`):
 *   target_i  = target[i], or (target == NULL: the value update, :128-129) reward[i] + gamma * next_value[i] in two roundings
 *   grad[i]   = (float)(2 / n) * (out[i] - target_i)          -- d loss / d out, exactly torch's mse_backward arithmetic
 *   *loss_sum += (double)(float)(sum_i (out[i] - target_i)^2 / n)   -- float64 accumulation of the float32 loss upstream reports
 * all device pointers, n floats each (loss_sum: one double, read-modify-written: calls on one stream are ordered).
 * workspace: device, RGL_MSE_WORKSPACE_BYTES, ZEROED ONCE by the caller and then left to this function (partial sums and an arrival
 * counter that every launch leaves at zero); only touched when n > 8192 (several workgroups), may be NULL below that.  The partial
 * sums are added in a fixed order: the reported loss does not depend on scheduling. */
#define RGL_MSE_WORKSPACE_BYTES 2048
int rgl_mse_step_f32(const float* out, const float* target, const float* reward, const float* next_value, float gamma, int n,
                     float* grad, double* loss_sum, void* workspace, rgl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * gcn_rotate_f32 -- pairwise relation features: (R,14) [robot 9 | human 5] -> (R,13)
 * agent-centric rows [dg, v_pref, theta, radius, vx, vy, px1, py1, vx1, vy1, radius1, da,
 * radius_sum].  Replaces CADRL.rotate (crowd_nav/policy/cadrl.py:241-276).
 * ------------------------------------------------------------------------------------------- */
int gcn_rotate_f32(const float* joint14, float* rotated13, int n_rows, int kinematics,
                   rgl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * gcn_predict_f32 -- path G one-step lookahead for B scenes x A actions, fused on device:
 * propagate robot (action) and humans (constant velocity), compute_reward, rotate, ValueNetwork,
 * value = reward + gamma^(dt*v_pref) * V, first-max argmax.
 * Replaces the search loop of MultiHumanRL.predict (crowd_nav/policy/multi_human_rl.py:36-64),
 * compute_reward (:73-96) and CADRL.propagate (cadrl.py:113-138).
 *   robot  device [B][9], humans device [B][H][5], actions device [A][2] (float64; (vx,vy) or (v,r))
 *   workspace device, >= gcn_predict_workspace_bytes(B, H, A) bytes
 *   action_values device [B][A] (float32), best_action device [B] (int32), best_value device [B] (float32; may be NULL;
 *   ABI 8) = action_values[b][best_action[b]] (0 where no action was selected: best_action = -1)
 * ------------------------------------------------------------------------------------------- */
typedef struct GcnPlanner {
    RglGraph graph;             /* w_r.dims[0] = 6, w_h.dims[0] = 7                             */
    RglMlp value_head;          /* gcn.ValueNetwork.value_net                                   */
    int kinematics;
    int num_actions;
    double time_step;
    double gamma;               /* raw gamma; discount is gamma^(time_step * v_pref)            */
    const double* actions;      /* device [A][2]                                                */
    /* optional (NULL = absent): the float64 states robot[B][9] / humans[B][H][5] that the fp32 arrays passed to
     * gcn_predict_f32 were rounded from.  When set, propagate / compute_reward start from them -- the reference
     * works on the simulator's python floats (cadrl.py:113-138, multi_human_rl.py:73-96) and rounds once, at to_tensor. */
    const double* root_robot_f64;
    const double* root_humans_f64;
    int contraction_dtype;      /* ABI 8: RGL_CONTRACT_F32 (0, the reference's arithmetic) | RGL_CONTRACT_BF16X6: the weight products   *
                                 * (Wa, W_l) of the graph forward as six bf16 MFMA terms over three-piece operands where the scene kernel *
                                 * offers them (softmax similarities, 17..32 nodes); everything else plain f32; other values: BAD_MODE  */
    int reserved;
} GcnPlanner;

/* The first step of gcn_predict_f32 on its own (ABI 5): for every (root b, action a) CADRL.propagate (cadrl.py:113-138) of the
 * robot, constant-velocity humans, CADRL.rotate (:241-276) into the pairwise relation features and compute_reward
 * (multi_human_rl.py:73-96; float64, reading planner->root_*_f64 when set).  Reads planner->actions, num_actions, kinematics,
 * time_step only.  self6 device [B*A][6], hum7 device [B*A][H][7], reward device [B*A]. */
int gcn_prepare_f32(const GcnPlanner* planner, const float* robot, const float* humans, int B, int H,
                    float* self6, float* hum7, float* reward, rgl_stream_t stream);
size_t gcn_predict_workspace_bytes(int B, int H, int A);
int gcn_predict_f32(const GcnPlanner* planner, const float* robot, const float* humans, int B, int H,
                    void* workspace, size_t workspace_bytes,
                    float* action_values, int* best_action, float* best_value, rgl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Path M: model-predictive rollout (crowd_nav/policy/model_predictive_rl.py:192-357).
 * ------------------------------------------------------------------------------------------- */
typedef struct MprlPlanner {
    RglGraph value_graph;       /* ValueEstimator.graph_model                                   */
    RglMlp value_head;          /* ValueEstimator.value_network                                 */
    RglGraph predictor_graph;   /* StatePredictor.graph_model (may alias value_graph's weights) */
    RglMlp motion_head;         /* StatePredictor.human_motion_predictor                        */
    int linear_state_predictor; /* 1: humans' = humans + v (no dt; state_predictor.py:109-118)  */
    int kinematics;
    int num_actions;
    int planning_depth;
    int planning_width;
    int do_action_clip;
    int sparse_search;
    int contraction_dtype;      /* RGL_CONTRACT_F32 (reference arithmetic) | RGL_CONTRACT_F16: f16 inputs, f32   *
                                 * accumulate for the dense products of the middle GCN layer of the children's  *
                                 * value graph (BASELINE configs[4]); RGL_ERR_BAD_MODE when the configuration   *
                                 * has no such kernel (needs embedded_gaussian, L = 3, N <= 64)                 *
                                 * | RGL_CONTRACT_BF16X6 (ABI 6): products whose operands keep all 24 significand   *
                                 * bits -- x = hi + mid + lo exactly, three bf16 pieces by round-to-nearest, f32's  *
                                 * exponent range, no scaling -- as the six terms W_lo a_hi + W_mid a_mid + W_hi    *
                                 * a_lo + W_mid a_hi + W_hi a_mid + W_hi a_hi of v_mfma_f32_16x16x32_bf16 with f32  *
                                 * accumulation; the dropped terms W_mid a_lo + W_lo a_mid + W_lo a_lo are at most   *
                                 * 2^-23 |W||a| (|mid| <= 2^-8 |x|, |lo| <= 2^-16 |x| in the worst case; ~2^-25      *
                                 * typically) -- the size of one f32 rounding of the product, and unbiased.          *
                                 * Offered by the value-of-children kernel of the shipped shape for 13 312 of the  *
                                 * 14 224 products of its three value-head matrices (what its LDS holds) *
                                 * and by the state predictor's scene kernel (softmax similarity, 17..32 nodes) for *
                                 * its weight products (Wa, W_l, motion head); everything else computes plain f32. */
    double time_step;
    double gamma_bar;           /* gamma^(time_step * v_pref), get_normalized_gamma (:104-105)  */
    const double* actions;      /* device [A][2] float64, table of build_action_space (:155-190)*/
    const int* action_groups;   /* device [A], action_group_index (sparse search); may be NULL.  Any int32 ids: the
                                 * kernel compares ids (no range to validate); a sparse search supports planning_width
                                 * <= 16, wider requests return RGL_ERR_BAD_MODE                                  */
    /* optional (NULL = absent): the float64 JointStates robot[B][9] / humans[B][H][5] that the fp32 ROOT arrays were
     * rounded from.  When set and the roots are joint states, estimate_reward of the ROOT level reads them: the reference
     * evaluates it on the simulator's python floats (model_predictive_rl.py:226,304-357) while the networks see the
     * fp32 tensors.  Deeper levels are tensor-born in the reference too. */
    const double* root_robot_f64;
    const double* root_humans_f64;
    /* optional (NULL = absent; ABI 3): device buffer of mprl_children_image_bytes() bytes holding the value-of-children
     * kernel's LDS weight image, prepared by mprl_pack_children_image_f32 from THIS planner's value_graph / value_head
     * weights.  The image depends on the weights only, so a caller that keeps them fixed over many searches (inference,
     * a captured decision graph) packs once and re-packs after a parameter update; without it every search (or
     * stand-alone mprl_value_children_f32 call) packs the image into its workspace first (~5 us). */
    const float* children_image;
    /* optional (NULL = absent; ABI 4): device buffer of mprl_predictor_image_bytes() bytes holding the state predictor's scene
     * kernel weight image in the layout of the planner's split mode --
     * three-piece bf16 fragments for RGL_CONTRACT_BF16X6 (ABI 6: the scene kernel's weight products Wa, W_l and the motion head as
     * six bf16 MFMA terms; S and A H stay f32) -- prepared by mprl_pack_predictor_image_f32 from THIS planner's predictor_graph /
     * motion_head.  NULL: a search in such a mode packs it into its workspace itself. */
    const float* predictor_image;
    /* optional (0 = absent; ABI 8): an upper bound of the action table's speeds -- max |(vx, vy)| (holonomic) or max |v| (unicycle)
     * over `actions`, any value >= the true maximum.  The reward step uses it to rule out, per parent, the humans no action can bring
     * within reach; without it every wave of that step derives it from the table (same results, ~1.5 us more latency per launch). */
    double action_speed_bound;
} MprlPlanner;

/* Bytes of the weight image above; 0 when the configuration has no image-based children kernel (the searches then ignore
 * `children_image`).  Depends on the architecture only (not on the weights, P, A or H).  Round 4: also offered for three-layer
 * graphs and crowds beyond 32 agents with the shipped embedding / head shapes (f32 layout): there the
 * image feeds the stage-2 head, which with an image at hand runs inside the stage-1 launch (children_deep_kernel) instead of in
 * a launch of its own. */
size_t mprl_children_image_bytes(const MprlPlanner* planner);
/* Builds the image from the planner's current value_graph / value_head weights on `stream` (planner->children_image is not
 * read).  RGL_ERR_BAD_MODE when mprl_children_image_bytes() is 0, RGL_ERR_WORKSPACE when `image_bytes` is too small. */
int mprl_pack_children_image_f32(const MprlPlanner* planner, float* image, size_t image_bytes, rgl_stream_t stream);
/* The same for MprlPlanner::predictor_image (ABI 4).  0 bytes: the planner's mode / predictor has no such image. */
size_t mprl_predictor_image_bytes(const MprlPlanner* planner);
int mprl_pack_predictor_image_f32(const MprlPlanner* planner, float* image, size_t image_bytes, rgl_stream_t stream);

/* One tree level for P parent states (the unit `action_clip` evaluates, :242-269):
 *   humans_next[p]      = StatePredictor humans of parent p (or the linear approximation)
 *   child_robot[p][a]   = compute_next_state(robot[p], action a)        (state_predictor.py:41-60)
 *   reward[p][a]        = estimate_reward(parent p, action a)           (:304-357)
 *   child_value[p][a]   = ValueEstimator(child_robot[p][a], humans_next[p])
 *   value1[p][a]        = reward + gamma_bar * child_value
 * parents_are_joint_states: 1 for root states that came from float64 JointStates (position
 * differences taken in float64), 0 for tensor-born states (differences rounded to float32 first,
 * as tensor_to_joint_state + numpy scalars do).
 * All outputs device; child_robot [P][A][9], the others [P][A] / [P][H][5].  `workspace` as for
 * mprl_value_children_f32 (NULL allowed: general kernel). */
int mprl_expand_f32(const MprlPlanner* planner, const float* robot, const float* humans, int P, int H,
                    int parents_are_joint_states,
                    float* humans_next, float* child_robot, float* reward, float* child_value,
                    float* value1, void* workspace, size_t workspace_bytes, rgl_stream_t stream);

/* The dominant kernel of the rollout on its own: child_value[p][a] = ValueEstimator(child_robot[p][a],
 * humans_next[p]) for the A sibling children of each of P parents (siblings share their crowd).
 * Same code path mprl_expand_f32 / mprl_tree_search_f32 use; exported so it can be timed and
 * tested in isolation.  `workspace` (device, mprl_value_children_workspace_bytes) is the MFMA kernels' scratch: the rows of
 * the partial 16-child tiles and (unless planner->children_image is set) the weight image of the one-launch kernel, or the
 * [P*A][64] hand-off between the two stages of the small-launch pair; without it (NULL) the general kernel runs
 * (value_estimator.py:11-20 applied to model_predictive_rl.py:245-250's loop). */
size_t mprl_value_children_workspace_bytes(const MprlPlanner* planner, int P, int H);
int mprl_value_children_f32(const MprlPlanner* planner, const float* child_robot, const float* humans_next,
                            int P, int H, float* child_value, void* workspace, size_t workspace_bytes,
                            rgl_stream_t stream);

/* Whole depth-D search for B root scenes: level-synchronous expansion, top-w clipping
 * (argpartition / sparse grouped variant), V_planning back-up (:271-302), first-max argmax.
 *   best_action device [B] int32, best_value device [B] float32
 *   root_values NULL or device [B][W0] float32, root_kept NULL or device [B][W0] int32, with
 *   W0 = planning_width if do_action_clip else num_actions; kept actions are ordered by
 *   descending one-step value (ties: lower action index first).
 * roots_are_joint_states = 1 (ABI 6 semantics): the roots are JointStates (predict(), :192-240).  Upstream then prices every
 * root action TWICE: action_clip is handed the float32 TENSOR of the state (:216-218 -> :246-248 -> tensor_to_joint_state,
 * crowd_sim/envs/utils/state.py:82-92: float32-born scalars, position differences rounded to float32), while the values of the
 * kept actions read the float64 JointState (:226).  The search does the same: level 0's selection uses the tensor-born rewards
 * (kept in the workspace at MprlLevelView::reward_clip_off), root_values / best_value the float64 ones (reward_off; read from
 * planner->root_*_f64 when set).  Without do_action_clip only the second exists.  roots_are_joint_states = 0: the roots are
 * tensor states themselves (V_planning's view of a state), one reward array. */
size_t mprl_tree_workspace_bytes(const MprlPlanner* planner, int B, int H);
int mprl_tree_search_f32(const MprlPlanner* planner, const float* robot, const float* humans, int B, int H,
                         int roots_are_joint_states, void* workspace, size_t workspace_bytes,
                         int* best_action, float* best_value, float* root_values, int* root_kept,
                         rgl_stream_t stream);

/* The same search with timing marks (ABI 5) -- a MEASUREMENT call, the one entry point that synchronises: the library records
 * HIP events on `stream` around the launches of every level l and, after waiting for the last one, reports (host arrays of
 * planning_depth floats, milliseconds)
 *   predictor_ms[l]  the level's state-predictor / next-state / reward launches,
 *   children_ms[l]   its value-of-children launch(es) AS THEY RUN IN THE SEARCH: with the selection -- and at the deepest level
 *                    the back-up chain and the root decision -- in the kernel's tail where the fused kernel takes the level
 *                    (the stand-alone select kernel included where it does not): what bench.py's `roofline` prices,
 *   *total_ms        (NULL allowed) first record to last record.
 * Not capturable into a hipGraph (event records, a host wait). */
int mprl_tree_search_traced_f32(const MprlPlanner* planner, const float* robot, const float* humans, int B, int H,
                                int roots_are_joint_states, void* workspace, size_t workspace_bytes,
                                int* best_action, float* best_value, float* root_values, int* root_kept,
                                rgl_stream_t stream, float* predictor_ms, float* children_ms, float* total_ms);

/* estimate_reward (model_predictive_rl.py:304-357) and compute_next_state (state_predictor.py:41-60) for every (parent,
 * action) pair on their own (ABI 5) -- the float64 reward kernel every tree level runs:
 *   child_robot device [P][A][9], reward device [P][A]; robot [P][9], humans [P][H][5] (each parent its own crowd);
 *   parents_are_joint_states as for mprl_expand_f32 (with planner->root_*_f64 set the float64 states are read).
 * Reads planner->actions, num_actions, kinematics, time_step only. */
int mprl_estimate_reward_f32(const MprlPlanner* planner, const float* robot, const float* humans, int P, int H,
                             int parents_are_joint_states, float* child_robot, float* reward, rgl_stream_t stream);

/* action_clip's selection on its own (model_predictive_rl.py:242-269; ABI 5): value1[p][a] = reward + gamma_bar * child_value
 * (each op rounded to fp32), keep[p][0..W-1] = the planning_width best actions of parent p in descending one-step value
 * (ties: lower index first; sparse_search: one action per action_groups id) -- W = num_actions and keep = 0..A-1 without
 * do_action_clip.  value1 device [P][A], keep device [P][W] int32.  A sparse search whose table has FEWER distinct group ids than
 * planning_width fills the tail of a row by repeating the last action kept (rows have a fixed width; upstream returns the shorter
 * list there: the first <number of distinct groups> entries of the row are that list). */
int mprl_action_clip_f32(const MprlPlanner* planner, const float* reward, const float* child_value, int P,
                         float* value1, int* keep, rgl_stream_t stream);

/* Where level `level` keeps its arrays inside the workspace (byte offsets), so a host can read
 * back the best trajectory (ModelPredictiveRL.traj) or any intermediate without a second pass. */
typedef struct MprlLevelView {
    long long n_parents;
    long long robot_off, humans_off;          /* parent states of this level                   */
    long long humans_next_off, child_robot_off, reward_off, child_value_off, value1_off;
    long long keep_off;                       /* int32 [P][W]                                  */
    long long backup_off;                     /* float32 [P][W] returns of the kept children   */
    long long best_slot_off;                  /* int32 [P] first-max slot among the kept       */
    long long reward_clip_off;                /* ABI 6: float32 [B][A], level 0 of a clipped search, else -1: the root rewards as
                                               * upstream's root action_clip reads them (see mprl_tree_search_f32); written by
                                               * searches with roots_are_joint_states = 1                                 */
} MprlLevelView;
int mprl_tree_level_view(const MprlPlanner* planner, int B, int H, int level, MprlLevelView* view);

/* ---------------------------------------------------------------------------------------------
 * Batched crowd simulator (next row of the scope table): B independent environments, float64 state.
 * crowd_step_f64 replaces CrowdSim.step (crowd_sim/envs/crowd_sim.py:252-368) incl. Agent.step /
 * compute_position (crowd_sim/envs/utils/agent.py:113-139) and, for human_policy LINEAR, Linear.predict
 * (crowd_sim/envs/policy/linear.py:16-22).  ORCA humans (external rvo2) are out of scope: supply their
 * actions with human_policy GIVEN.
 *   robot [B][9], humans [B][H][5], time [B], done [B] (int, in/out): updated in place when update != 0
 *   (update = 0 is the reference's onestep_lookahead: outputs only); environments with done != 0 are frozen.
 *   human_goals [B][H][2], human_vpref [B][H] (LINEAR); human_actions [B][H][2] (GIVEN); robot_action [B][2]
 *   = (vx,vy) or (v,r).  Outputs: reward [B] fp32, info [B] CROWD_INFO_*, dmin [B] (closest approach; -1 on collision).
 * crowd_observe_f32: the fp32 (robot [B][9], humans [B][H][5]) view policies consume (JointState.to_tensor,
 * crowd_sim/envs/utils/state.py:64-79).
 * ------------------------------------------------------------------------------------------- */
enum { CROWD_INFO_NOTHING = 0, CROWD_INFO_DISCOMFORT = 1, CROWD_INFO_COLLISION = 2, CROWD_INFO_REACH_GOAL = 3,
       CROWD_INFO_TIMEOUT = 4, CROWD_INFO_DONE = 5 };
enum { CROWD_HUMAN_GIVEN = 0, CROWD_HUMAN_LINEAR = 1, CROWD_HUMAN_CONSTANT_VELOCITY = 2 };

typedef struct CrowdSimConfig {
    double time_step, time_limit;
    double success_reward, collision_penalty, discomfort_dist, discomfort_penalty_factor;
    int kinematics;             /* of the robot: RGL_HOLONOMIC | RGL_UNICYCLE */
    int human_policy;           /* CROWD_HUMAN_* */
} CrowdSimConfig;

int crowd_step_f64(const CrowdSimConfig* cfg, double* robot, double* humans, const double* human_goals,
                   const double* human_vpref, const double* robot_action, const double* human_actions,
                   double* time, int* done, int B, int H, int update, float* reward, int* info, double* dmin,
                   rgl_stream_t stream);
int crowd_observe_f32(const double* robot, const double* humans, int B, int H, float* robot32, float* humans32,
                      rgl_stream_t stream);

/* library identification: ABI version and the gfx target the device code was built for */
int rgl_abi_version(void);
const char* rgl_build_target(void);

#ifdef __cplusplus
}
#endif
#endif /* RGL_HIP_H */
