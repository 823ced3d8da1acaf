// Shared helpers of the librgl_hip translation units (host + device).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "rgl_hip.h"

#define RGL_HIP_TRY(expr)                          \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

#define RGL_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

namespace rgl {

constexpr int kLdsBytesPerCu = 160 * 1024;

inline int validate_mlp(const RglMlp& m, int want_in, int want_out) {
    if (m.n_layers < 1 || m.n_layers > RGL_MAX_MLP_LAYERS) return RGL_ERR_BAD_SHAPE;
    for (int l = 0; l <= m.n_layers; ++l)
        if (m.dims[l] < 1 || m.dims[l] > RGL_MAX_WIDTH) return RGL_ERR_BAD_SHAPE;
    for (int l = 0; l < m.n_layers; ++l)
        if (!m.weight[l] || !m.bias[l]) return RGL_ERR_NULL;
    if (want_in > 0 && m.dims[0] != want_in) return RGL_ERR_BAD_SHAPE;
    if (want_out > 0 && m.dims[m.n_layers] != want_out) return RGL_ERR_BAD_SHAPE;
    return RGL_OK;
}

inline int validate_graph(const RglGraph& g, int H) {
    if (H < 1 || H + 1 > RGL_MAX_NODES) return RGL_ERR_BAD_SHAPE;
    if (g.x_dim < 1 || g.x_dim > RGL_MAX_XDIM) return RGL_ERR_BAD_SHAPE;
    if (g.num_layer < 0 || g.num_layer > RGL_MAX_GCN_LAYERS) return RGL_ERR_BAD_SHAPE;
    if (g.similarity < RGL_SIM_EMBEDDED_GAUSSIAN || g.similarity > RGL_SIM_DIAGONAL) return RGL_ERR_BAD_MODE;
    int rc = validate_mlp(g.w_r, 0, g.x_dim);
    if (rc) return rc;
    rc = validate_mlp(g.w_h, 0, g.x_dim);
    if (rc) return rc;
    if (g.similarity == RGL_SIM_EMBEDDED_GAUSSIAN && !g.w_a) return RGL_ERR_NULL;
    if (g.similarity == RGL_SIM_CONCATENATION) {
        rc = validate_mlp(g.w_a_mlp, 2 * g.x_dim, 1);
        if (rc) return rc;
        if (g.w_a_mlp.n_layers != 2) return RGL_ERR_BAD_SHAPE;   // 2X -> hidden -> 1, as the reference builds it
    }
    for (int l = 0; l < g.num_layer; ++l)
        if (!g.Ws[l]) return RGL_ERR_NULL;
    return RGL_OK;
}

// ---- internal launchers (one translation unit each; a return value of 1 means "outside this kernel's envelope") -------------
int validate_forward_call(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head, const float* robot,
                          const float* humans, int n_scenes, int scenes_per_crowd, int H, const float* value_out,
                          const float* humans_next);                                                                // rgl_generic.hip
// module forwards (values and / or next humans of S scenes) through the one-wave-per-scene MFMA kernel; 0 bytes / 1 = the
// kernel does not cover the configuration
size_t scene_forward_workspace_bytes(const RglGraph* g, const RglMlp* value_head, const RglMlp* motion_head, int S, int crowds_per,
                                     int H);                                                                        // rgl_scene.hip
int launch_scene_forward(const RglGraph* g, const RglMlp* value_head, const RglMlp* motion_head, const float* robot,
                         const float* humans, int S, int crowds_per, int H, float* value_out, float* humans_next, void* workspace,
                         size_t workspace_bytes, hipStream_t stream, const float* value_rows_image = nullptr);      // rgl_scene.hip
// `value_rows_image`: the three-piece bf16 image of the graph's matrices (scene_image_bytes_for / pack_scene_image_for with no motion
// head): the value forward's weight products then run as six bf16 MFMA terms where the scene kernel offers them
int launch_generic_forward(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head,
                           const float* robot, const float* humans, int n_scenes, int scenes_per_crowd, int H,
                           float* H_out, float* A_out, float* value_out, float* humans_next, hipStream_t stream);   // rgl_generic.hip
// `image` (both launchers of the two-stage pair): null, or the packed weight image of these weights (FusedLds layout,
// pack_images_kernel) -- the kernels then copy their weight images instead of building them from the raw matrices
int launch_rank1_children(const RglGraph* g, int P, int A, int H, const float* child_robot, const float* humans_next,
                          float* rows_out, hipStream_t stream, const float* image = nullptr);                       // rgl_rank1.hip
int launch_deep_children(const RglGraph* g, int P, int A, int H, const float* child_robot, const float* humans_next,
                         float* rows_out, int f16, hipStream_t stream, const RglMlp* head = nullptr, float* value = nullptr,
                         const float* image = nullptr, const void* tail = nullptr, size_t tail_bytes = 0, int* tail_done = nullptr,
                         int* head_done = nullptr);                                                                 // rgl_deep.hip
int launch_tile_children(const RglGraph* g, int P, int A, int H, const float* child_robot, const float* humans_next,
                         float* rows_out, hipStream_t stream);                                                      // rgl_tile.hip
// `tail` (a TailArgs, opaque) with A = rows per parent: the head kernel's workgroups own whole parents and run the search's select /
// back-up / root steps for them in its tail (*tail_done = 1 / 2 as for the fused children kernel; the generic-dims head has none)
int launch_head_rows(const RglGraph* g, const RglMlp* head, const float* rows, int M, float* value,
                     hipStream_t stream, const float* image = nullptr, const void* tail = nullptr, size_t tail_bytes = 0,
                     int* tail_done = nullptr, int A = 0);                                                          // rgl_head.hip
// the fused tile kernel (rgl_fused.hip).  Its weight image is prepared in global memory by pack_images_kernel: by the caller once per
// parameter state (caller_image = MprlPlanner::children_image), else once per tree search (image_ready = 1 on the per-level calls)
// or by the call itself, at the END of the workspace it is given.
// `tail` (optional): a TailArgs (rgl_tail.h, passed opaquely with its size) -- the search's select / back-up / root steps for the
// parents of this launch; the kernel runs them in its tail and reports *tail_done = 1 (selection done) or 2 (the deepest level:
// back-up steps and root decision done as well).  0 = the caller launches the stand-alone kernels.
int launch_fused_children(const RglGraph* g, const RglMlp* head, int P, int A, int H, const float* child_robot,
                          const float* humans_next, float* child_value, void* workspace, size_t workspace_bytes,
                          int image_ready, hipStream_t stream, const float* caller_image = nullptr,
                          const void* tail = nullptr, size_t tail_bytes = 0, int* tail_done = nullptr, int mode = 0);
// mode 2: the six-term bf16 products (RGL_CONTRACT_BF16X6): the image then holds three-piece bf16 fragments for that kernel only
int pack_children_images(const RglGraph* g, const RglMlp* head, int P, int A, int H, void* workspace, size_t workspace_bytes,
                         hipStream_t stream, int mode = 0);   // P = the largest launch; 1 = the fused kernel does not apply
size_t fused_children_workspace_bytes(int P, int A, int H);
const float* fused_workspace_image(const void* workspace, size_t workspace_bytes);   // where pack_children_images put the image
int launch_value_children(const MprlPlanner* pl, const float* child_robot, const float* humans_next, int P, int H,
                          float* child_value, void* workspace, size_t workspace_bytes, hipStream_t stream,
                          int image_ready = 0, const void* tail = nullptr, size_t tail_bytes = 0,
                          int* tail_done = nullptr);                                                                // rgl_fast.hip
size_t value_children_workspace_bytes(const MprlPlanner* pl, int P, int H);                                        // rgl_fast.hip
// `children` (optional): a ChildrenArgs (rgl_children.h, passed opaquely with its size) describing the level's independent
// next-state / reward work; when the MFMA scene kernel runs, it executes that work on extra workgroups of the same launch
// and sets *children_done.
// `sp_image`: the three-piece bf16 weight image of the scene kernel (scene_image_bytes, pack_scene_image) when the planner's mode is
// RGL_CONTRACT_BF16X6; without one the f32 form of the kernel runs
int launch_predict_humans(const MprlPlanner* pl, const float* robot, const float* humans, int crowds_per, int P, int H,
                          float* humans_next, void* workspace, size_t workspace_bytes, hipStream_t stream,
                          const void* children = nullptr, size_t children_bytes = 0, int* children_done = nullptr,
                          const float* sp_image = nullptr);                                                           // rgl_scene.hip
size_t scene_image_bytes(const MprlPlanner* pl);                      // 0: no six-term bf16 scene kernel for this planner
int pack_scene_image(const MprlPlanner* pl, float* image, hipStream_t stream);
size_t scene_image_bytes_for(const RglGraph& g, const RglMlp* motion_head);          // the same for a graph (+ optional motion head)
int pack_scene_image_for(const RglGraph& g, const RglMlp* motion_head, float* image, hipStream_t stream);

int launch_scene_children(const MprlPlanner* pl, const float* child_robot, const float* humans_next, int P, int H,
                          float* child_value, void* workspace, size_t workspace_bytes, hipStream_t stream);         // rgl_scene.hip
size_t scene_children_workspace_bytes(int P, int A, int H);                                                        // rgl_scene.hip

// the backward pass of large batches as MFMA tile kernels; 1 = outside its envelope / below its batch threshold
int launch_backward_mfma(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head, const float* robot,
                         const float* humans, int n_scenes, int H, int detach_graph, const float* d_value,
                         const float* d_humans_next, const float* d_H, float* grad_out, void* workspace, size_t workspace_bytes,
                         hipStream_t stream, int only_choice);                                                                     // rgl_backward_mfma.hip

// forward of models outside the shipped shapes (other embedding MLPs, x_dim = 64) on the tile kernels of rgl_backward_mfma.hip instead
// of the general VALU kernel: embedded_gaussian / gaussian, one adjacency, 1-3 layers, N <= 64.  0 bytes / 1 = not covered.
size_t tiles_forward_workspace_bytes(const RglGraph* g, const RglMlp* value_head, const RglMlp* motion_head, int S, int crowds_per,
                                     int H, int want_H);
int launch_tiles_forward(const RglGraph* g, const RglMlp* value_head, const RglMlp* motion_head, const float* robot,
                         const float* humans, int S, int crowds_per, int H, float* H_out, float* value_out, float* humans_next,
                         void* workspace, size_t workspace_bytes, hipStream_t stream);                               // rgl_backward_mfma.hip

inline int mlp_max_hidden(const RglMlp& m) {
    int w = 0;
    for (int l = 1; l < m.n_layers; ++l) w = m.dims[l] > w ? m.dims[l] : w;
    return w;
}

}  // namespace rgl
