"""ctypes binding of librgl_hip.so (the C ABI declared in include/rgl_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, the product
raises.  The CPU oracle under `oracle/` is test infrastructure and is never imported from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RGL_HIP_LIBRARY") or os.path.join(_HERE, "lib", "librgl_hip.so")   # override: debug builds only

MAX_MLP_LAYERS = 6
MAX_GCN_LAYERS = 8
MAX_NODES = 128
MAX_XDIM = 64
MAX_WIDTH = 256
MAX_ACTIONS = 256
ABI_VERSION = 8
MSE_WORKSPACE_BYTES = 2048

SIMILARITY = {"embedded_gaussian": 0, "gaussian": 1, "cosine": 2, "cosine_softmax": 3, "concatenation": 4,
              "squared": 5, "equal_attention": 6, "diagonal": 7}
KINEMATICS = {"holonomic": 0, "unicycle": 1}
CONTRACTION_DTYPES = {"f32": 0, "f16": 1, "bf16x6": 3}      # 2 was "f16x3" (ABI 4..7)

ERRORS = {-1: "RGL_ERR_BAD_SHAPE", -2: "RGL_ERR_BAD_MODE", -3: "RGL_ERR_NULL", -4: "RGL_ERR_WORKSPACE",
          -5: "RGL_ERR_LDS"}

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)


class RglMlp(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("last_relu", C.c_int), ("dims", C.c_int * (MAX_MLP_LAYERS + 1)),
                ("weight", C.c_void_p * MAX_MLP_LAYERS), ("bias", C.c_void_p * MAX_MLP_LAYERS)]


class RglTransposeJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int)]


class RglGatherJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_floats", C.c_int), ("src_rows", C.c_int)]


class RglGraph(C.Structure):
    _fields_ = [("w_r", RglMlp), ("w_h", RglMlp), ("x_dim", C.c_int), ("num_layer", C.c_int),
                ("similarity", C.c_int), ("layerwise_graph", C.c_int), ("skip_connection", C.c_int),
                ("reserved", C.c_int), ("w_a", C.c_void_p), ("w_a_mlp", RglMlp),
                ("Ws", C.c_void_p * MAX_GCN_LAYERS)]


class GcnPlanner(C.Structure):
    _fields_ = [("graph", RglGraph), ("value_head", RglMlp), ("kinematics", C.c_int), ("num_actions", C.c_int),
                ("time_step", C.c_double), ("gamma", C.c_double), ("actions", C.c_void_p),
                ("root_robot_f64", C.c_void_p), ("root_humans_f64", C.c_void_p), ("contraction_dtype", C.c_int), ("reserved", C.c_int)]


class MprlPlanner(C.Structure):
    _fields_ = [("value_graph", RglGraph), ("value_head", RglMlp), ("predictor_graph", RglGraph),
                ("motion_head", RglMlp), ("linear_state_predictor", C.c_int), ("kinematics", C.c_int),
                ("num_actions", C.c_int), ("planning_depth", C.c_int), ("planning_width", C.c_int),
                ("do_action_clip", C.c_int), ("sparse_search", C.c_int), ("contraction_dtype", C.c_int),
                ("time_step", C.c_double), ("gamma_bar", C.c_double), ("actions", C.c_void_p),
                ("action_groups", C.c_void_p), ("root_robot_f64", C.c_void_p), ("root_humans_f64", C.c_void_p),
                ("children_image", C.c_void_p), ("predictor_image", C.c_void_p), ("action_speed_bound", C.c_double)]


class CrowdSimConfig(C.Structure):
    _fields_ = [("time_step", C.c_double), ("time_limit", C.c_double), ("success_reward", C.c_double),
                ("collision_penalty", C.c_double), ("discomfort_dist", C.c_double),
                ("discomfort_penalty_factor", C.c_double), ("kinematics", C.c_int), ("human_policy", C.c_int)]


class MprlLevelView(C.Structure):
    _fields_ = [(n, C.c_longlong) for n in
                ("n_parents", "robot_off", "humans_off", "humans_next_off", "child_robot_off", "reward_off",
                 "child_value_off", "value1_off", "keep_off", "backup_off", "best_slot_off", "reward_clip_off")]


# name -> (restype, argtypes); every symbol include/rgl_hip.h declares
SIGNATURES = {
    "rgl_graph_forward_workspace_bytes": (C.c_size_t, [C.POINTER(RglGraph), C.POINTER(RglMlp), C.POINTER(RglMlp), C.c_int,
                                                       C.c_int, C.c_int]),
    "rgl_graph_forward_f32": (C.c_int, [C.POINTER(RglGraph), C.POINTER(RglMlp), C.POINTER(RglMlp), C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rgl_graph_param_count": (C.c_int, [C.POINTER(RglGraph), C.POINTER(RglMlp), C.POINTER(RglMlp)]),
    "rgl_graph_backward_workspace_bytes": (C.c_size_t, [C.POINTER(RglGraph), C.POINTER(RglMlp), C.POINTER(RglMlp), C.c_int]),
    "rgl_graph_backward_f32": (C.c_int, [C.POINTER(RglGraph), C.POINTER(RglMlp), C.POINTER(RglMlp), C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    "rgl_transpose_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rgl_transpose_many_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "rgl_gather_rows_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "rgl_mse_step_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "gcn_rotate_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "gcn_prepare_f32": (C.c_int, [C.POINTER(GcnPlanner), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "gcn_predict_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "gcn_predict_f32": (C.c_int, [C.POINTER(GcnPlanner), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                  C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mprl_expand_f32": (C.c_int, [C.POINTER(MprlPlanner), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.c_void_p]),
    "mprl_value_children_workspace_bytes": (C.c_size_t, [C.POINTER(MprlPlanner), C.c_int, C.c_int]),
    "mprl_children_image_bytes": (C.c_size_t, [C.POINTER(MprlPlanner)]),
    "mprl_pack_children_image_f32": (C.c_int, [C.POINTER(MprlPlanner), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mprl_predictor_image_bytes": (C.c_size_t, [C.POINTER(MprlPlanner)]),
    "mprl_pack_predictor_image_f32": (C.c_int, [C.POINTER(MprlPlanner), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mprl_value_children_f32": (C.c_int, [C.POINTER(MprlPlanner), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p]),
    "mprl_tree_workspace_bytes": (C.c_size_t, [C.POINTER(MprlPlanner), C.c_int, C.c_int]),
    "mprl_tree_search_f32": (C.c_int, [C.POINTER(MprlPlanner), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "mprl_tree_search_traced_f32": (C.c_int, [C.POINTER(MprlPlanner), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, c_float_p, c_float_p, c_float_p]),
    "mprl_estimate_reward_f32": (C.c_int, [C.POINTER(MprlPlanner), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "mprl_action_clip_f32": (C.c_int, [C.POINTER(MprlPlanner), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "mprl_tree_level_view": (C.c_int, [C.POINTER(MprlPlanner), C.c_int, C.c_int, C.c_int, C.POINTER(MprlLevelView)]),
    "crowd_step_f64": (C.c_int, [C.POINTER(CrowdSimConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "crowd_observe_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rgl_abi_version": (C.c_int, []),
    "rgl_build_target": (C.c_char_p, []),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library; raises NativeLibraryError (never falls back) when it cannot be used."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                "librgl_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C relationalgraphlearning_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise NativeLibraryError("librgl_hip.so lacks symbol %s" % name) from e
            fn.restype = res
            fn.argtypes = args
        if handle.rgl_abi_version() != ABI_VERSION:
            raise NativeLibraryError("librgl_hip.so ABI %d != expected %d" % (handle.rgl_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def poison_workspaces():
    """RGL_DEBUG_POISON_WORKSPACES=1 (tests): workspaces and output slabs are filled with NaN bit patterns before the kernels run, so
    that a read of memory no kernel has written shows up as NaN instead of as whatever the allocator handed out."""
    return os.environ.get("RGL_DEBUG_POISON_WORKSPACES", "0") == "1"


def check(rc, what):
    if rc == 0:
        return
    if rc in ERRORS:
        raise NativeLibraryError("%s failed: %s" % (what, ERRORS[rc]))
    raise NativeLibraryError("%s failed: hipError_t %d" % (what, rc))
