// rgl_fast.hip -- MFMA kernels for the shipped configuration (embedded_gaussian, non-layerwise,
// skip connection, x_dim 32, hidden 64).  Until they are in place every call routes to the
// general kernel of rgl_generic.hip (same results, lower throughput).
#include "rgl_common.h"

namespace rgl {
int launch_generic_forward(const RglGraph* graph, const RglMlp* value_head, const RglMlp* motion_head,
                           const float* robot, const float* humans, int n_scenes, int scenes_per_crowd, int H,
                           float* H_out, float* A_out, float* value_out, float* humans_next, hipStream_t stream);

// V(child) for the A children of each of P parents; children of one parent share humans_next[p].
int launch_value_children(const MprlPlanner* pl, const float* child_robot, const float* humans_next, int P, int H,
                          float* child_value, hipStream_t stream) {
    return launch_generic_forward(&pl->value_graph, &pl->value_head, nullptr, child_robot, humans_next,
                                  P * pl->num_actions, pl->num_actions, H, nullptr, nullptr, child_value, nullptr, stream);
}
}  // namespace rgl
