// rgl_fast.hip -- "value of the sibling children" (mprl_value_children_f32): choice of the stage-1 kernel and the two-stage
// launch.  The f32-MFMA kernels themselves live in
//   rgl_rank1.hip  shared-crowd rank-1 form, L = 2, N <= 32 (the shipped configuration)
//   rgl_deep.hip   shared-crowd form for L in {2, 3}, N <= 60, optional f16-input MFMA contractions
//   rgl_tile.hip   every child's graph in full, any depth, N <= 64 (softmax similarities)
//   rgl_head.hip   stage 2: last GCN layer on the robot row + value head
//   rgl_scene.hip  the state predictor's graph forward (one scene per wave)
// f32 MFMA (v_mfma_f32_16x16x4_f32) is exact fp32 at the vector-FMA rate; results differ from the general kernel
// (rgl_generic.hip) only by summation order.  Anything outside these envelopes runs on the general kernel (same
// numbers, lower speed) -- never on the CPU.
#include "rgl_mfma.h"

namespace rgl {

size_t value_children_workspace_bytes(const MprlPlanner* pl, int P, int H) {
    const size_t children = (size_t)P * pl->num_actions * 64 * sizeof(float);
    const size_t predictor = (size_t)P * (H + 1) * XD * sizeof(float);         // embeddings of launch_predict_humans
    const size_t fused = H + 1 <= 32 ? fused_children_workspace_bytes(P, pl->num_actions, H) : 0;
    const size_t scene = scene_children_workspace_bytes(P, pl->num_actions, H);
    // the tile kernels, for planners outside the shipped shapes: children of the value graph, scenes of the state predictor
    const size_t tiles_v = tiles_forward_workspace_bytes(&pl->value_graph, &pl->value_head, nullptr, P * pl->num_actions, pl->num_actions, H, 0);
    const size_t tiles_p = pl->linear_state_predictor ? 0 : tiles_forward_workspace_bytes(&pl->predictor_graph, nullptr, &pl->motion_head, P, 1, H, 0);
    const size_t tiles = tiles_v > tiles_p ? tiles_v : tiles_p;
    const size_t m0 = children > predictor ? children : predictor;
    const size_t m1 = m0 > scene ? m0 : scene;
    const size_t m = m1 > tiles ? m1 : tiles;
    // the fused kernel keeps its weight images at the END of the workspace: room for them behind every other use
    return H + 1 <= 32 ? ((m + 255) & ~(size_t)255) + fused_children_workspace_bytes(0, pl->num_actions, H) + (fused > m ? fused - m : 0) : m;
}

// V(child) for the A children of each of P parents; children of one parent share humans_next[p].
int launch_value_children(const MprlPlanner* pl, const float* child_robot, const float* humans_next, int P, int H,
                          float* child_value, void* workspace, size_t workspace_bytes, hipStream_t stream, int image_ready,
                          const void* tail, size_t tail_bytes, int* tail_done) {
    if (tail_done) *tail_done = 0;
    const int A = pl->num_actions;
    const int hv = head_variant(pl->value_head);
    const bool want_f16 = pl->contraction_dtype == RGL_CONTRACT_F16;
    // RGL_CONTRACT_BF16X6 (round 5): f32-WIDTH products (three bf16 pieces per operand, six terms) on the matrix pipe where a kernel
    // offers them -- the first 64 input features of the fused kernel's last head matrix; plain f32 everywhere else
    const bool want_b6 = pl->contraction_dtype == RGL_CONTRACT_BF16X6;
    const int fused_mode = want_b6 ? 2 : 0;           // kModeBx / kModeF32 (rgl_fused.hip)
    if (pl->contraction_dtype != RGL_CONTRACT_F32 && !want_f16 && !want_b6) return RGL_ERR_BAD_MODE;      // (2, the split-f16 mode of ABI 4..7, is gone)
    const bool staged = hv >= 0 && workspace && workspace_bytes >= value_children_workspace_bytes(pl, P, H);
    // stage 1, in order of preference: rank-1 (L = 2, N <= 32), shared-crowd deep (L in {2,3}, N <= 60), tiles (softmax
    // similarities, any depth, N <= 64); everything else, or a head without a stage-2 kernel: the general kernel
    int rc = 1;                                            // 1 = no stage-1 kernel launched yet
    if (staged && !want_f16) {
        // one fused kernel over 16-child tiles (L = 2, N <= 32, default head): values come out directly, no stage 2
        rc = launch_fused_children(&pl->value_graph, &pl->value_head, P, A, H, child_robot, humans_next, child_value, workspace,
                                   workspace_bytes, image_ready, stream, pl->children_image, tail, tail_bytes, tail_done, fused_mode);
        if (rc != 1) return rc;
    }
    // packed weight image of the value estimator (the caller's, or this search's at the end of the workspace): the two-stage pair
    // copies its weight images from it instead of building them from the raw matrices (most of a small launch)
    const float* image = (!staged || want_b6) ? nullptr          // (a three-piece bf16 image is in the fused kernel's layout only)
                         : pl->children_image ? pl->children_image
                         : image_ready ? fused_workspace_image(workspace, workspace_bytes) : nullptr;
    if (staged) {
        const RglGraph* g = &pl->value_graph;
        float* rows = (float*)workspace;
        if (!want_f16) rc = launch_rank1_children(g, P, A, H, child_robot, humans_next, rows, stream, image);
        if (rc == 1) {
            int head_done = 0;
            rc = launch_deep_children(g, P, A, H, child_robot, humans_next, rows, want_f16 && g->num_layer == 3, stream,
                                      &pl->value_head, child_value, image, tail, tail_bytes, tail_done, &head_done);
            if (want_f16 && (rc == 1 || g->num_layer != 3)) return RGL_ERR_BAD_MODE;
            if (rc != 1 && head_done) return rc;           // stage 2 (and the tail) ran inside the launch
            if (rc == 1 && tail_done) *tail_done = 0;
        }
        if (rc == 1) rc = launch_tile_children(g, P, A, H, child_robot, humans_next, rows, stream);
        if (rc == 1 && !want_f16) {
            // every child's graph in full, one wave per child: the remaining similarity functions and layerwise graphs
            rc = launch_scene_children(pl, child_robot, humans_next, P, H, child_value, workspace, workspace_bytes, stream);
            if (rc != 1) return rc;
        }
    }
    if (want_f16 && rc == 1) return RGL_ERR_BAD_MODE;
    if (rc == 1) {
        // outside the shipped shapes (other embedding MLPs, x_dim = 64): the tile kernels, the children of a parent sharing its crowd's rows
        rc = launch_tiles_forward(&pl->value_graph, &pl->value_head, nullptr, child_robot, humans_next, P * A, A, H, nullptr, child_value,
                                  nullptr, workspace, workspace_bytes, stream);
        if (rc != 1) return rc;
    }
    if (rc == 1) {
        // RGL_REQUIRE_MFMA_CHILDREN=1 (tests): refuse instead of running the general VALU kernel
        static const bool require = [] { const char* e = getenv("RGL_REQUIRE_MFMA_CHILDREN"); return e && e[0] == '1'; }();
        if (require) return RGL_ERR_BAD_MODE;
    }
    if (rc == 1)
        return launch_generic_forward(&pl->value_graph, &pl->value_head, nullptr, child_robot, humans_next, P * A, A, H,
                                      nullptr, nullptr, child_value, nullptr, stream);
    if (rc) return rc;
    return launch_head_rows(&pl->value_graph, &pl->value_head, (const float*)workspace, P * A, child_value, stream, image,
                            tail, tail_bytes, tail_done, A);
}

}  // namespace rgl
