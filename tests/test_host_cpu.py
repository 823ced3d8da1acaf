"""CPU-side tests (-m "not gpu"): the C ABI library loads and exports every declared symbol, host
logic (action tables, sharding, configs, checkpoints) is right, and the product refuses to run
without the device path."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import relationalgraphlearning_amd as rga
from relationalgraphlearning_amd import _native as nat
from tests import golden_io as gio
from tests.helpers import make_mprl_policy, make_gcn_policy, JS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "rgl_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|size_t|const char\*)\s+(\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(nat.SIGNATURES), declared ^ set(nat.SIGNATURES)
    assert os.path.exists(nat.LIB_PATH), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(nat.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    lib = nat.lib()
    assert lib.rgl_abi_version() == nat.ABI_VERSION
    assert lib.rgl_build_target() == b"gfx950"


def test_struct_layout_matches_header_sizes():
    # sizes implied by include/rgl_hip.h on LP64 (ints 4 B, pointers 8 B, natural alignment)
    mlp = 4 + 4 + 4 * 7 + 4 + 8 * 6 + 8 * 6          # 4 B padding after dims[7] for pointer alignment
    assert ctypes.sizeof(nat.RglMlp) == mlp
    graph = 2 * mlp + 6 * 4 + 8 + mlp + 8 * 8
    assert ctypes.sizeof(nat.RglGraph) == graph
    assert ctypes.sizeof(nat.MprlLevelView) == 12 * 8               # ABI 6: + reward_clip_off
    assert ctypes.sizeof(nat.MprlPlanner) == 2 * graph + 2 * mlp + 8 * 4 + 2 * 8 + 2 * 8 + 2 * 8 + 8 + 8 + 8  # ABI 8: + action_speed_bound; ABI 2: + float64 root pointers; ABI 3: + children_image; ABI 4: + predictor_image
    assert ctypes.sizeof(nat.GcnPlanner) == graph + mlp + 2 * 4 + 2 * 8 + 8 + 2 * 8 + 2 * 4        # ABI 8: + contraction_dtype, reserved
    assert ctypes.sizeof(nat.RglTransposeJob) == 2 * 8 + 2 * 4 and ctypes.sizeof(nat.RglGatherJob) == 2 * 8 + 2 * 4      # ABI 7


def test_contraction_modes_and_level_view_match_the_header():
    """ABI 6 (7 adds entry points only): the contraction-mode constants of include/rgl_hip.h are the ones the Python side sends; the level view exposes the
    root's tensor-born reward array (reward_clip_off) at level 0 of a clipped search only."""
    hdr = open(os.path.join(ROOT, "include", "rgl_hip.h")).read()
    consts = {name: int(val) for name, val in re.findall(r"#define (RGL_CONTRACT_\w+)\s+(\d+)", hdr)}
    assert consts == {"RGL_CONTRACT_F32": 0, "RGL_CONTRACT_F16": 1, "RGL_CONTRACT_BF16X6": 3}      # 2 (F16X3) left with ABI 8
    assert nat.CONTRACTION_DTYPES == {"f32": 0, "f16": 1, "bf16x6": 3}
    assert int(re.search(r"#define RGL_ABI_VERSION (\d+)", hdr).group(1)) == nat.ABI_VERSION == 8
    lib = nat.lib()
    pl = nat.MprlPlanner()
    pl.planning_depth, pl.planning_width, pl.num_actions, pl.do_action_clip = 2, 2, 81, 1
    view = nat.MprlLevelView()
    assert lib.mprl_tree_level_view(ctypes.byref(pl), 64, 19, 0, ctypes.byref(view)) == 0 and view.reward_clip_off > 0
    assert view.reward_clip_off % 256 == 0 and view.reward_clip_off != view.reward_off
    assert lib.mprl_tree_level_view(ctypes.byref(pl), 64, 19, 1, ctypes.byref(view)) == 0 and view.reward_clip_off == -1
    pl.do_action_clip = 0
    assert lib.mprl_tree_level_view(ctypes.byref(pl), 64, 19, 0, ctypes.byref(view)) == 0 and view.reward_clip_off == -1


def test_host_side_argument_checks_without_gpu():
    lib = nat.lib()
    # NULL pointers and bad shapes are rejected on the host before any launch
    assert lib.rgl_transpose_f32(None, None, 4, 4, None) == -3
    assert lib.rgl_transpose_many_f32(None, 2, None) == -3 and lib.rgl_transpose_many_f32(None, 0, None) == 0
    assert ctypes.sizeof(nat.RglTransposeJob) == 24
    assert lib.gcn_rotate_f32(None, None, 4, 0, None) == -3
    assert lib.mprl_tree_workspace_bytes(None, 4, 5) == 0
    assert lib.gcn_predict_workspace_bytes(4, 5, 81) > 0
    pl = nat.MprlPlanner()
    pl.planning_depth, pl.planning_width, pl.num_actions, pl.do_action_clip = 2, 2, 81, 1
    n2 = lib.mprl_tree_workspace_bytes(ctypes.byref(pl), 2048, 19)
    pl.planning_depth = 3
    n3 = lib.mprl_tree_workspace_bytes(ctypes.byref(pl), 2048, 19)
    assert 0 < n2 < n3 < (1 << 31)
    view = nat.MprlLevelView()
    assert lib.mprl_tree_level_view(ctypes.byref(pl), 2048, 19, 2, ctypes.byref(view)) == 0
    assert view.n_parents == 2048 * 4
    assert lib.mprl_tree_level_view(ctypes.byref(pl), 2048, 19, 3, ctypes.byref(view)) == -1
    # ABI 3: the caller-owned weight image -- no image for an architecture without the image-based kernel (empty descriptors)
    assert lib.mprl_children_image_bytes(None) == 0
    assert lib.mprl_children_image_bytes(ctypes.byref(pl)) == 0
    assert lib.mprl_pack_children_image_f32(None, None, 0, None) == -3
    assert lib.mprl_pack_children_image_f32(ctypes.byref(pl), ctypes.c_void_p(16), 0, None) != 0


def test_product_refuses_cpu_tensors():
    pol = make_mprl_policy("trained", 1)
    pol.set_device(torch.device("cpu"))
    pl = gio.load("planning")
    js = JS(pl["plan.scene.s5.robot"][0], pl["plan.scene.s5.humans"][0])
    with pytest.raises(nat.NativeLibraryError):
        pol.predict(js)
    with torch.no_grad(), pytest.raises(nat.NativeLibraryError):
        pol.value_estimator((torch.zeros(1, 1, 9), torch.zeros(1, 2, 5)))
    g = make_gcn_policy()
    g.set_device(torch.device("cpu"))
    with pytest.raises(nat.NativeLibraryError):
        g.predict(js)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "relationalgraphlearning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "/root/reference" not in text, f


def test_action_tables_and_groups():
    ar = gio.load("actions_rewards")
    pol = make_mprl_policy("trained")
    pol.build_action_space(1.0)
    assert len(pol.action_space) == 81 and pol.action_space[0] == rga.ActionXY(0, 0)
    from relationalgraphlearning_amd.actions import as_array
    assert np.array_equal(as_array(pol.action_space), ar["act.mprl"])
    assert pol.action_group_index == ar["act.mprl_groups"].tolist()
    g = make_gcn_policy()
    g.build_action_space(1.0)
    assert np.array_equal(as_array(g.action_space), ar["act.gcn"])


def test_parameter_names_and_shapes_match_reference_checkpoints():
    pol = make_mprl_policy("rand", L=3)
    sd = pol.get_state_dict()
    m = gio.master("rand")
    for k, v in sd["graph_model1"].items():
        assert tuple(v.shape) == m["graph_model1." + k].shape, k
    assert [tuple(v.shape) for v in sd["value_network"].values()] == \
        [(32, 32), (32,), (100, 32), (100,), (100, 100), (100,), (1, 100), (1,)]
    assert [tuple(v.shape) for v in sd["motion_predictor"].values()] == [(64, 32), (64,), (5, 64), (5,)]
    shared = make_mprl_policy("rand", variant="shared")
    assert set(shared.get_state_dict()) == {"graph_model", "value_network", "motion_predictor"}
    assert shared.state_predictor.graph_model is shared.value_estimator.graph_model
    lin = make_mprl_policy("rand", variant="linear")
    assert set(lin.get_state_dict()) == {"graph_model", "value_network"} and not lin.state_predictor.trainable
    g = make_gcn_policy()
    assert set(g.get_state_dict()) == set(gio.path_g_sd())
    assert pol.get_model() is pol.value_estimator
    assert abs(pol.get_normalized_gamma() - 0.9 ** 0.25) < 1e-15


def test_logical_eval_counts():
    assert make_mprl_policy("rand", 1).tree_search().logical_value_evals_per_root() == 81
    assert make_mprl_policy("rand", 2, 2, True).tree_search().logical_value_evals_per_root() == 249
    assert make_mprl_policy("rand", 3, 2, True).tree_search().logical_value_evals_per_root() == 581
    assert make_mprl_policy("rand", 2, 1, False).tree_search().logical_value_evals_per_root() == 81 * 82


def test_registration_uses_reference_keys():
    factory = {}
    rga.register(factory)
    assert factory["model_predictive_rl"] is rga.ModelPredictiveRL and factory["gcn"] is rga.GCN
    p = factory["model_predictive_rl"]()
    for attr in ("trainable", "phase", "model", "device", "last_state", "time_step", "env", "name",
                 "multiagent_training", "kinematics", "planning_depth", "traj", "state_predictor"):
        assert hasattr(p, attr), attr
    for meth in ("configure", "set_phase", "set_device", "set_env", "set_time_step", "set_epsilon", "get_model",
                 "save_model", "load_model", "get_state_dict", "load_state_dict", "predict", "transform",
                 "get_normalized_gamma", "get_traj"):
        assert callable(getattr(p, meth)), meth
    q = factory["gcn"]()
    assert hasattr(q, "action_values") and callable(q.get_matrix_A)


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 8, 2048, 4097):
        for world in (1, 2, 3, 8):
            spans = [rga.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_modules_deepcopy_and_pickle_with_populated_descriptor_cache():
    """Trainer.update_target_model deep-copies the model after forwards have run (crowd_nav/utils/trainer.py:41,187;
    train.py:161,206): the descriptor caches hold ctypes structs with raw pointers, which must not travel with a copy."""
    import copy
    import io
    pol = make_mprl_policy("trained", D=1)
    ve, sp = pol.value_estimator, pol.state_predictor
    for cache in (ve._cache, ve.graph_model._cache, sp._cache, sp.graph_model._cache):
        cache.key, cache.value, cache.keep = ("stale",), nat.RglGraph(), [torch.zeros(3)]   # what a forward leaves behind
    ve2, sp2 = copy.deepcopy(ve), copy.deepcopy(sp)
    for cache in (ve2._cache, ve2.graph_model._cache, sp2._cache, sp2.graph_model._cache):
        assert cache.key is None and cache.value is None and cache.keep is None
    assert ve._cache.key == ("stale",)                                  # the original keeps its cache
    for (k, a), (_, b) in zip(ve.state_dict().items(), ve2.state_dict().items()):
        assert torch.equal(a, b) and a.data_ptr() != b.data_ptr(), k
    buf = io.BytesIO()
    torch.save(ve, buf)
    buf.seek(0)
    ve3 = torch.load(buf, weights_only=False)
    assert ve3._cache.key is None and set(ve3.state_dict()) == set(ve.state_dict())
    g = make_gcn_policy().model
    g._cache.key, g._cache.value = ("stale",), nat.RglGraph()
    g._head_cache.key, g._head_cache.value = ("stale",), nat.RglMlp()
    g2 = copy.deepcopy(g)
    assert g2._cache.key is None and g2._head_cache.key is None


def test_empty_crowd_takes_the_reference_greedy_action():
    """Fixture vnrl_trainer.npz (greedy.*): the reference MultiHumanRL.predict on JointStates without humans (multi_human_rl.py:27-31 ->
    CADRL.select_greedy_action).  No network is involved, so this runs without a GPU."""
    from tests import golden_io as gio
    from tests.helpers import make_gcn_policy, JS
    fx = gio.load("vnrl_trainer")
    pol = make_gcn_policy()
    pol.device = "cpu"                                   # only marks the policy as configured: nothing is evaluated on it
    for row, want in zip(fx["greedy.robot"], fx["greedy.action"]):
        a = pol.predict(JS(row, []))
        assert a == pol.action_space[int(want)]


def test_flat_params_follow_the_module_like_parameters_does():
    """nets._flat_params (the descriptor caches' view of a module's parameters: a cached sub-module list, parameters read from the
    sub-modules' dicts on every call) against nn.Module.parameters(): same objects in the same order for every module of both
    policies, after a deep copy (the copy must list ITS parameters), after a parameter object was replaced, and after pickling."""
    import copy
    import io
    from relationalgraphlearning_amd import nets
    pol = make_mprl_policy("trained", 1)
    gcn = make_gcn_policy()
    for mod in (pol.value_estimator, pol.state_predictor, pol.value_estimator.graph_model, gcn.model):
        assert [id(p) for p in nets._flat_params(mod)] == [id(p) for p in mod.parameters()]
        assert [id(p) for p in nets._flat_params(mod)] == [id(p) for p in mod.parameters()]          # from the cached list
        twin = copy.deepcopy(mod)
        assert [id(p) for p in nets._flat_params(twin)] == [id(p) for p in twin.parameters()]
        assert not set(id(p) for p in nets._flat_params(twin)) & set(id(p) for p in mod.parameters())
        buf = io.BytesIO()
        torch.save(mod, buf)
        buf.seek(0)
        back = torch.load(buf, weights_only=False)
        assert [id(p) for p in nets._flat_params(back)] == [id(p) for p in back.parameters()]
    ve = pol.value_estimator
    ve.value_network[0].weight = torch.nn.Parameter(torch.zeros_like(ve.value_network[0].weight))      # a replaced Parameter object
    assert [id(p) for p in nets._flat_params(ve)] == [id(p) for p in ve.parameters()]
    # a replaced SUB-MODULE (the kept sub-module list used to hide it for ever) and an added one
    old = [id(p) for p in nets._flat_params(ve)]
    ve.value_network[2] = torch.nn.Linear(ve.value_network[2].in_features, ve.value_network[2].out_features)
    assert [id(p) for p in nets._flat_params(ve)] == [id(p) for p in ve.parameters()] != old
    ve.value_network.add_module("extra", torch.nn.Linear(1, 1))
    assert [id(p) for p in nets._flat_params(ve)] == [id(p) for p in ve.parameters()]
    nets.invalidate_packed_weights(ve)
    assert "_rgl_submodules" not in ve.__dict__


def test_a_failed_pack_leaves_no_transposes_queued():
    """pack_mlp queues one transpose job per Linear weight; a descriptor that fails part-way (here: the module's
    parameters are CPU tensors) must take its jobs back out instead of leaving them for the next unrelated flush."""
    from relationalgraphlearning_amd import nets
    pol = make_mprl_policy("trained", 1)
    assert not nets._PENDING_TRANSPOSES
    with pytest.raises(nat.NativeLibraryError):
        pol.value_estimator.graph_model.descriptor()
    assert not nets._PENDING_TRANSPOSES
    with pytest.raises(nat.NativeLibraryError):
        with nets.batched_transposes():
            pol.value_estimator.graph_model.descriptor()
    assert not nets._PENDING_TRANSPOSES and nets._BATCH_DEPTH[0] == 0


def test_tree_search_image_caches_are_keyed_on_the_contraction_mode():
    """the children / predictor weight images are laid out per contraction mode; the cache key must carry it."""
    import inspect
    from relationalgraphlearning_amd import rollout
    for fn in (rollout.TreeSearch._children_image, rollout.TreeSearch._predictor_image):
        src = inspect.getsource(fn)
        assert "self.contraction_dtype" in src.split("ent = ")[0], fn.__name__


def test_planner_speed_bound_and_path_g_contraction_modes():
    """ABI 8 host side: MprlPlanner.action_speed_bound is the top speed of the search's own action table (holonomic: |(vx, vy)|,
    unicycle: |v|) -- the reward step prunes far humans with it, so it must never be below a table entry; path G's search accepts the
    two contraction modes GcnPlanner knows and refuses anything else before any device work."""
    import numpy as np
    from relationalgraphlearning_amd import actions as act
    from relationalgraphlearning_amd.rollout import GcnSearch, TreeSearch
    table = act.speed_major_table(1.0, 5, 16, "holonomic", np.pi / 3)[0]
    arr = act.as_array(table)
    ts = TreeSearch(None, None, arr, None, "holonomic")
    assert ts._speed_bound() == float(np.hypot(arr[:, 0], arr[:, 1]).max()) and ts._speed_bound() >= 1.0 - 1e-12
    uni = act.speed_major_table(1.0, 5, 16, "unicycle", np.pi / 3)[0]
    ua = act.as_array(uni)
    assert TreeSearch(None, None, ua, None, "unicycle")._speed_bound() == float(np.abs(ua[:, 0]).max())
    for mode in ("f32", "bf16x6"):
        assert GcnSearch(None, arr, contraction_dtype=mode).contraction_dtype == mode
    with pytest.raises(ValueError):
        GcnSearch(None, arr, contraction_dtype="f16")
