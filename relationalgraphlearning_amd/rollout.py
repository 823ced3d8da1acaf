"""Batched action-tree rollout on the MI355X and its multi-GPU sharding.

`TreeSearch` drives mprl_tree_search_f32 / mprl_expand_f32 (path M: the depth-D search of
crowd_nav/policy/model_predictive_rl.py:192-302 for B root scenes at once); `GcnSearch` drives
gcn_predict_f32 (path G: the one-step search of crowd_nav/policy/multi_human_rl.py:36-64).
`ShardedRollout` splits root scenes over the ranks of a torch.distributed group (one process per
GPU) and exchanges the per-shard results with ONE all-gather -- RCCL over xGMI on the GPU box,
gloo in the CPU tests.  Root trees are independent, so there is no other communication.
"""
import ctypes as C

import os

import numpy as np
import torch

from . import _native as nat
from .nets import _require_device_tensor, _stream, batched_transposes


def _check_roots64(roots64, B, H):
    """The float64 JointStates the fp32 roots were rounded from: contiguous float64 device tensors (B,9) / (B,H,5)."""
    r64, h64 = roots64
    if not (torch.is_tensor(r64) and torch.is_tensor(h64) and r64.dtype == torch.float64 and h64.dtype == torch.float64
            and r64.is_cuda and h64.is_cuda and r64.is_contiguous() and h64.is_contiguous()
            and tuple(r64.shape) == (B, 9) and tuple(h64.shape) == (B, H, 5)):
        raise ValueError("roots64 must be contiguous float64 device tensors (B,9) / (B,H,5)")
    return r64, h64


class _Workspace:
    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        if nat.poison_workspaces() and not torch.cuda.is_current_stream_capturing():
            self.buf.fill_(255)                  # tests: NaN bit patterns wherever a kernel reads what no kernel of THIS search wrote
        return self.buf


class TreeSearch:
    """Depth-D, width-w model-predictive search for batches of root scenes (path M)."""

    def __init__(self, value_estimator, state_predictor, actions, action_groups, kinematics="holonomic",
                 time_step=0.25, gamma_bar=0.9 ** 0.25, planning_depth=1, planning_width=1, do_action_clip=False,
                 sparse_search=False, contraction_dtype="f32"):
        self.value_estimator = value_estimator
        self.state_predictor = state_predictor            # StatePredictor module or LinearStatePredictor
        self.actions_np = np.ascontiguousarray(np.asarray(actions, dtype=np.float64))
        self.groups_np = None if action_groups is None else np.asarray(action_groups, dtype=np.int32)
        # group ids are arbitrary int32 values, as in the reference's python set (the select kernel compares ids, round 3)
        self.kinematics = kinematics
        self.time_step = float(time_step)
        self.gamma_bar = float(gamma_bar)
        self.planning_depth = int(planning_depth)
        self.planning_width = int(planning_width)
        self.do_action_clip = bool(do_action_clip)
        self.sparse_search = bool(sparse_search)
        self.contraction_dtype = contraction_dtype     # "f32" (reference arithmetic) | "f16" (BASELINE configs[4])
        self._dev_tables = {}
        self._ws = _Workspace()
        self._ws2 = _Workspace()    # hand-off buffer of the stand-alone expand / value_children calls
        self.last = None            # outputs of the most recent search (device tensors)
        self._decisions = {}        # (H, device) -> captured single-scene search (decide())
        self._images = {}           # device -> (parameter-state key, weight image of the value-of-children kernel)
        self._sp_images = {}        # device -> (parameter-state key, three-piece bf16 weight image of the state predictor's scene kernel)

    # -- descriptors -----------------------------------------------------------------------------
    @property
    def num_actions(self):
        return self.actions_np.shape[0]

    @property
    def kept_per_node(self):
        return self.planning_width if self.do_action_clip else self.num_actions

    def logical_value_evals_per_root(self):
        """Number of ValueEstimator forwards the reference performs for one decision."""
        A, w, D = self.num_actions, self.kept_per_node, self.planning_depth

        def V(d):                       # forwards inside V_planning(state, d)
            if d == 1:
                return 1
            clip = A if self.do_action_clip else 0
            return 1 + clip + w * V(d - 1)
        root_clip = A if self.do_action_clip else 0
        return root_clip + w * V(D)

    def _speed_bound(self):
        """MprlPlanner.action_speed_bound (ABI 8): max speed over the action table, once per table."""
        if getattr(self, "_speed_bound_cache", None) is None:
            a = self.actions_np
            v = np.hypot(a[:, 0], a[:, 1]) if self.kinematics == "holonomic" else np.abs(a[:, 0])
            self._speed_bound_cache = float(v.max()) if a.shape[0] else 0.0
        return self._speed_bound_cache

    def _tables(self, device):
        key = str(device)
        if key not in self._dev_tables:
            act = torch.tensor(self.actions_np, dtype=torch.float64, device=device)
            grp = None if self.groups_np is None else torch.tensor(self.groups_np, dtype=torch.int32, device=device)
            self._dev_tables[key] = (act, grp)
        return self._dev_tables[key]

    def planner(self, device):
        pl = nat.MprlPlanner()
        ve = self.value_estimator
        linear = not getattr(self.state_predictor, "trainable", False)
        with batched_transposes():                 # up to four descriptors: one transpose launch for all their Linear weights
            pl.value_graph = ve.graph_model.descriptor()
            pl.value_head = ve.head_descriptor()
            pl.linear_state_predictor = int(linear)
            if not linear:
                pl.predictor_graph = self.state_predictor.graph_model.descriptor()
                pl.motion_head = self.state_predictor.head_descriptor()
        pl.kinematics = nat.KINEMATICS[self.kinematics]
        pl.num_actions = self.num_actions
        pl.planning_depth = self.planning_depth
        pl.planning_width = self.planning_width
        pl.do_action_clip = int(self.do_action_clip)
        pl.sparse_search = int(self.sparse_search)
        # RGL_CONTRACT_F32_AS=bf16x6 (measurements / the admission run of the GPU suite): every search that asks for plain f32 runs
        # the 24-bit-operand bf16 mode instead, so that the whole suite's f32 bounds are held against it (profiles/r05_suite_under_bf16x6.txt)
        mode = self.contraction_dtype
        if mode == "f32" and os.environ.get("RGL_CONTRACT_F32_AS"):
            mode = os.environ["RGL_CONTRACT_F32_AS"]
        if mode not in nat.CONTRACTION_DTYPES:
            raise ValueError("contraction mode %r (contraction_dtype / RGL_CONTRACT_F32_AS): one of %s"
                             % (mode, sorted(nat.CONTRACTION_DTYPES)))
        pl.contraction_dtype = nat.CONTRACTION_DTYPES[mode]
        pl.time_step = self.time_step
        pl.gamma_bar = self.gamma_bar
        act, grp = self._tables(device)
        pl.actions = act.data_ptr()
        pl.action_groups = None if grp is None else grp.data_ptr()
        pl.action_speed_bound = self._speed_bound()
        image = self._children_image(pl, device)
        pl.children_image = None if image is None else image.data_ptr()
        if not linear and mode == "bf16x6":
            image = self._predictor_image(pl, device)
            pl.predictor_image = None if image is None else image.data_ptr()
        return pl

    def _predictor_image(self, pl, device):
        """MprlPlanner.predictor_image (ABI 4): the state predictor's scene-kernel weight image in the three-piece bf16 layout, packed
        when the predictor's descriptors were (re)built, into the same device buffer every time.  None when the mode or the
        predictor has no such kernel."""
        sp = self.state_predictor
        key = (sp.graph_model._cache.epoch, sp._cache.epoch, self.contraction_dtype, os.environ.get("RGL_CONTRACT_F32_AS"))   # the layout depends on the mode
        dkey = str(device)
        ent = self._sp_images.get(dkey)
        if ent is not None and ent[0] == key:
            return ent[1]
        lib = nat.lib()
        nbytes = lib.mprl_predictor_image_bytes(C.byref(pl))
        buf = None
        if nbytes:
            buf = ent[1] if ent is not None and ent[1] is not None and ent[1].numel() == nbytes else \
                torch.empty(nbytes, dtype=torch.uint8, device=device)
            nat.check(lib.mprl_pack_predictor_image_f32(C.byref(pl), buf.data_ptr(), nbytes, _stream()),
                      "mprl_pack_predictor_image_f32")
        self._sp_images[dkey] = (key, buf)
        return buf

    def _children_image(self, pl, device):
        """The value-of-children kernel's weight image for the CURRENT parameters (MprlPlanner.children_image): packed when the
        descriptors of the value estimator were (re)built, into the same device buffer every time (captured decision graphs keep
        reading it), so searches with unchanged weights skip the per-search packing launch.  None when the configuration has no
        image-based kernel."""
        ve = self.value_estimator
        gcache, hcache = ve.graph_model._cache, ve._cache
        # process-wide pack serials (nets._PACK_SERIAL) identify the parameter state; the image's LAYOUT depends on the
        # contraction mode (f32 matrices vs three-piece bf16 fragments, same byte size): a mode changed on a live object repacks
        key = (gcache.epoch, hcache.epoch, self.contraction_dtype, os.environ.get("RGL_CONTRACT_F32_AS"))
        dkey = str(device)
        ent = self._images.get(dkey)
        if ent is not None and ent[0] == key:
            return ent[1]
        lib = nat.lib()
        nbytes = lib.mprl_children_image_bytes(C.byref(pl))
        buf = None
        if nbytes:
            buf = ent[1] if ent is not None and ent[1] is not None and ent[1].numel() == nbytes else \
                torch.empty(nbytes, dtype=torch.uint8, device=device)
            nat.check(lib.mprl_pack_children_image_f32(C.byref(pl), buf.data_ptr(), nbytes, _stream()),
                      "mprl_pack_children_image_f32")
        self._images[dkey] = (key, buf)
        return buf

    # -- device calls ----------------------------------------------------------------------------
    def search(self, robot, humans, roots_are_joint_states=True, want_root_values=True, out=None, roots64=None, trace=False):
        """robot (B,9), humans (B,H,5) fp32 device tensors -> dict of device tensors:
        best_action (B,) int32, best_value (B,) fp32, root_values/root_kept (B,W0).
        `trace`: run mprl_tree_search_traced_f32 instead (a measurement call: it waits for the search) and leave
        {"predictor_ms": [...], "children_ms": [...], "total_ms": x} (per tree level, HIP events on the launch stream) in
        `self.last["trace"]`.
        `out` = (int32 (B,), fp32 (B,)) contiguous device tensors to receive best_action / best_value in place.
        `roots64` = (robot (B,9), humans (B,H,5)) float64 device tensors: the JointStates the fp32 roots were rounded from; the
        root level's estimate_reward reads them, as the reference does (model_predictive_rl.py:226)."""
        robot = _require_device_tensor(robot, "robot states")
        humans = _require_device_tensor(humans, "human states")
        B, H = robot.shape[0], humans.shape[1]
        dev = robot.device
        with torch.cuda.device(dev):
            pl = self.planner(dev)
            if roots64 is not None:
                r64, h64 = _check_roots64(roots64, B, H)
                pl.root_robot_f64, pl.root_humans_f64 = r64.data_ptr(), h64.data_ptr()
            lib = nat.lib()
            nbytes = lib.mprl_tree_workspace_bytes(C.byref(pl), B, H)
            if nbytes == 0:
                raise nat.NativeLibraryError("mprl_tree_workspace_bytes rejected the configuration")
            ws = self._ws.get(nbytes, dev)
            W0 = self.kept_per_node
            if out is not None:
                act_out, val_out = out
                if not (act_out.dtype == torch.int32 and val_out.dtype == torch.float32 and act_out.is_contiguous()
                        and val_out.is_contiguous() and act_out.numel() == B and val_out.numel() == B):
                    raise ValueError("out must be contiguous (int32 (B,), float32 (B,)) device tensors")
                out = {"best_action": act_out, "best_value": val_out}
            else:
                out = {"best_action": torch.empty(B, dtype=torch.int32, device=dev),
                       "best_value": torch.empty(B, dtype=torch.float32, device=dev)}
            rv = rk = None
            if want_root_values:
                out["root_values"] = torch.empty(B, W0, dtype=torch.float32, device=dev)
                out["root_kept"] = torch.empty(B, W0, dtype=torch.int32, device=dev)
                rv, rk = out["root_values"].data_ptr(), out["root_kept"].data_ptr()
            tr = None
            if trace:
                D = self.planning_depth
                sp_ms, ch_ms, tot = (C.c_float * D)(), (C.c_float * D)(), C.c_float()
                rc = lib.mprl_tree_search_traced_f32(C.byref(pl), robot.data_ptr(), humans.data_ptr(), B, H,
                                                     int(roots_are_joint_states), ws.data_ptr(), ws.numel(),
                                                     out["best_action"].data_ptr(), out["best_value"].data_ptr(), rv, rk, _stream(),
                                                     sp_ms, ch_ms, C.byref(tot))
                tr = {"predictor_ms": list(sp_ms), "children_ms": list(ch_ms), "total_ms": float(tot.value)}
            else:
                rc = lib.mprl_tree_search_f32(C.byref(pl), robot.data_ptr(), humans.data_ptr(), B, H,
                                              int(roots_are_joint_states), ws.data_ptr(), ws.numel(),
                                              out["best_action"].data_ptr(), out["best_value"].data_ptr(), rv, rk, _stream())
        nat.check(rc, "mprl_tree_search_traced_f32" if trace else "mprl_tree_search_f32")
        self.last = dict(out, B=B, H=H, robot=robot, humans=humans, planner=pl, workspace=ws, roots64=roots64, trace=tr,
                         roots_are_joint_states=bool(roots_are_joint_states))
        return out

    def decide(self, robot_row, human_rows):
        """One decision for ONE scene given as python floats (robot: 9 numbers, humans: H rows of 5): the whole search replayed
        from a hipGraph captured for this crowd size.  Returns the action index; `self.last` describes the search as after
        `search()`.  What `ModelPredictiveRL.predict` (model_predictive_rl.py:192-240) costs per call is then one host-to-device
        copy of the state, one graph launch and one 4-byte read-back instead of ~8 eager launches per tree level.
        The descriptors are refreshed before every replay: parameters updated in place (optimizer steps, load_state_dict) are
        re-transposed into the SAME device buffers the graph reads, and a parameter whose storage moved triggers a re-capture."""
        dev = next(self.value_estimator.parameters()).device
        if dev.type != "cuda":
            raise nat.NativeLibraryError("the policy's networks are on %s: the search runs only on the MI355X kernels (no CPU path)" % dev)
        H = len(human_rows)
        key = (H, str(dev))
        ent = self._decisions.get(key)
        with torch.cuda.device(dev):
            sig = bytes(self.planner(dev))                       # repacks (in place) if any parameter changed
            if ent is None or ent["sig"] != sig:
                ent = self._capture_decision(H, dev)
                ent["sig"] = bytes(self.planner(dev))
                self._decisions[key] = ent
            ent["host64"][:9] = torch.as_tensor(robot_row, dtype=torch.float64)
            ent["host64"][9:] = torch.as_tensor(human_rows, dtype=torch.float64).reshape(-1)
            ent["dev64"].copy_(ent["host64"], non_blocking=True)
            ent["robot"].copy_(ent["dev64"][:9].reshape(1, 9))                      # fp32 views the networks see (to_tensor)
            ent["humans"].copy_(ent["dev64"][9:].reshape(1, H, 5))
            ent["graph"].replay()
            self.last = ent["last"]
            return int(ent["last"]["best_action"][0])

    def _capture_decision(self, H, dev):
        host64 = torch.empty(9 + 5 * H, dtype=torch.float64).pin_memory()
        dev64 = torch.zeros(9 + 5 * H, dtype=torch.float64, device=dev)
        robot = torch.zeros(1, 9, dtype=torch.float32, device=dev)
        humans = torch.zeros(1, H, 5, dtype=torch.float32, device=dev)
        robot[0, 4], robot[0, 7], humans[0, :, 4] = 0.3, 1.0, 0.3                  # a harmless warm-up scene
        humans[0, :, 0] = torch.arange(H, device=dev, dtype=torch.float32) + 2.0
        dev64[:9], dev64[9:] = robot.reshape(-1).double(), humans.reshape(-1).double()
        roots64 = (dev64[:9].reshape(1, 9), dev64[9:].reshape(1, H, 5))
        shared_ws, self._ws = self._ws, _Workspace()                  # the graph bakes its workspace pointer: a private one
        try:
            self.search(robot, humans, True, roots64=roots64)       # warm-up: workspace, descriptors, function attributes
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.search(robot, humans, True, roots64=roots64)
            last = self.last
        finally:
            self._ws = shared_ws
        return {"graph": graph, "host64": host64, "dev64": dev64, "robot": robot, "humans": humans, "last": last}

    def capture(self, robot, humans, roots_are_joint_states=True, want_root_values=True, out=None, private_workspace=False):
        """Capture one whole search into a hipGraph (torch.cuda.CUDAGraph).  Returns (graph, outputs): copy new root
        states into `robot` / `humans` in place, call graph.replay(), read `outputs` -- no Python or launch overhead
        per decision.  The library allocates nothing and never synchronises, which is what makes this legal.
        The graph bakes device pointers: parameters may change IN PLACE (their transposed copies are refreshed in place by
        the next `planner()` / `search()` call -- call one of them before replaying after an optimizer step), but a parameter
        moved to new storage needs a new capture (`decide()` does both checks itself).  The workspace pointer is baked too:
        with `private_workspace` the graph owns its workspace (returned as outputs["workspace"]), so later searches of other
        sizes through this object cannot free it underneath the graph; `out` as in `search()`."""
        shared_ws = self._ws
        if private_workspace:
            self._ws = _Workspace()
        try:
            self.search(robot, humans, roots_are_joint_states, want_root_values, out=out)   # warm-up: workspace, descriptors
            torch.cuda.synchronize(robot.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                res = self.search(robot, humans, roots_are_joint_states, want_root_values, out=out)
            res = dict(res, workspace=self._ws.buf, last=self.last)
        finally:
            self._ws = shared_ws
        return graph, res

    def expand(self, robot, humans, parents_are_joint_states=True):
        """One tree level for P parents (what `action_clip` evaluates); all outputs device tensors."""
        robot = _require_device_tensor(robot, "robot states")
        humans = _require_device_tensor(humans, "human states")
        P, H, A = robot.shape[0], humans.shape[1], self.num_actions
        dev = robot.device
        o = {"humans_next": torch.empty(P, H, 5, device=dev), "child_robot": torch.empty(P, A, 9, device=dev),
             "reward": torch.empty(P, A, device=dev), "child_value": torch.empty(P, A, device=dev),
             "value1": torch.empty(P, A, device=dev)}
        with torch.cuda.device(dev):
            pl = self.planner(dev)
            ws = self._ws2.get(nat.lib().mprl_value_children_workspace_bytes(C.byref(pl), P, H), dev)
            rc = nat.lib().mprl_expand_f32(C.byref(pl), robot.data_ptr(), humans.data_ptr(), P, H,
                                           int(parents_are_joint_states), o["humans_next"].data_ptr(),
                                           o["child_robot"].data_ptr(), o["reward"].data_ptr(),
                                           o["child_value"].data_ptr(), o["value1"].data_ptr(), ws.data_ptr(),
                                           ws.numel(), _stream())
        nat.check(rc, "mprl_expand_f32")
        return o

    def estimate_reward(self, robot, humans, parents_are_joint_states=True, roots64=None, actions=None):
        """estimate_reward + compute_next_state for every (parent, action) pair (mprl_estimate_reward_f32: the float64 reward
        kernel of a tree level on its own; model_predictive_rl.py:304-357, state_predictor.py:41-60).  robot (P,9), humans
        (P,H,5) -> (child_robot (P,A,9), reward (P,A)).  `actions`: a float64 (A',2) table to use instead of the policy's."""
        robot = _require_device_tensor(robot, "robot states")
        humans = _require_device_tensor(humans, "human states")
        P, H = robot.shape[0], humans.shape[1]
        dev = robot.device
        with torch.cuda.device(dev):
            pl = self.planner(dev)
            table = None
            if actions is not None:
                table = torch.as_tensor(np.asarray(actions, dtype=np.float64).reshape(-1, 2), dtype=torch.float64).to(dev).contiguous()
                pl.actions, pl.num_actions = table.data_ptr(), table.shape[0]
            A = pl.num_actions
            if roots64 is not None:
                r64, h64 = _check_roots64(roots64, P, H)
                pl.root_robot_f64, pl.root_humans_f64 = r64.data_ptr(), h64.data_ptr()
            child = torch.empty(P, A, 9, dtype=torch.float32, device=dev)
            reward = torch.empty(P, A, dtype=torch.float32, device=dev)
            rc = nat.lib().mprl_estimate_reward_f32(C.byref(pl), robot.data_ptr(), humans.data_ptr(), P, H,
                                                    int(parents_are_joint_states), child.data_ptr(), reward.data_ptr(), _stream())
            nat.check(rc, "mprl_estimate_reward_f32")
            if table is not None:
                torch.cuda.current_stream(dev).synchronize()          # the temporary table must outlive the launch
        return child, reward

    def action_clip(self, reward, child_value, width=None):
        """action_clip's selection (mprl_action_clip_f32; model_predictive_rl.py:242-269): reward, child_value (P,A) ->
        (value1 (P,A) = reward + gamma_bar * child_value, keep (P,W) int32: the `width` (default: planning_width) best actions
        per parent in descending one-step value, one per group in a sparse search; every action when clipping is off)."""
        reward = _require_device_tensor(reward, "rewards")
        child_value = _require_device_tensor(child_value, "child values")
        P, A = reward.shape
        dev = reward.device
        with torch.cuda.device(dev):
            pl = self.planner(dev)
            if width is not None:
                pl.planning_width = int(width)
            W = pl.planning_width if pl.do_action_clip else A
            value1 = torch.empty(P, A, dtype=torch.float32, device=dev)
            keep = torch.empty(P, W, dtype=torch.int32, device=dev)
            rc = nat.lib().mprl_action_clip_f32(C.byref(pl), reward.data_ptr(), child_value.data_ptr(), P, value1.data_ptr(),
                                                keep.data_ptr(), _stream())
        nat.check(rc, "mprl_action_clip_f32")
        return value1, keep

    def value_children(self, child_robot, humans_next, out=None):
        """child_robot (P,A,9), humans_next (P,H,5) -> child_value (P,A): the dominant kernel alone."""
        child_robot = _require_device_tensor(child_robot, "child robot states")
        humans_next = _require_device_tensor(humans_next, "next human states")
        P, H = humans_next.shape[0], humans_next.shape[1]
        dev = child_robot.device
        if out is None:
            out = torch.empty(P, self.num_actions, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            pl = self.planner(dev)
            ws = self._ws2.get(nat.lib().mprl_value_children_workspace_bytes(C.byref(pl), P, H), dev)
            rc = nat.lib().mprl_value_children_f32(C.byref(pl), child_robot.data_ptr(), humans_next.data_ptr(), P, H,
                                                   out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        nat.check(rc, "mprl_value_children_f32")
        return out

    def level_arrays(self, level):
        """Views into the workspace of the last search for one tree level (device tensors)."""
        L = self.last
        view = nat.MprlLevelView()
        nat.check(nat.lib().mprl_tree_level_view(C.byref(L["planner"]), L["B"], L["H"], level, C.byref(view)),
                  "mprl_tree_level_view")
        ws, P, A, W, H = L["workspace"], int(view.n_parents), self.num_actions, self.kept_per_node, L["H"]

        def f32(off, *shape):
            n = int(np.prod(shape))
            return ws[off:off + 4 * n].view(torch.float32).reshape(*shape)

        def i32(off, *shape):
            n = int(np.prod(shape))
            return ws[off:off + 4 * n].view(torch.int32).reshape(*shape)
        arr = {"n_parents": P,
               "humans_next": f32(view.humans_next_off, P, H, 5), "child_robot": f32(view.child_robot_off, P, A, 9),
               "reward": f32(view.reward_off, P, A), "child_value": f32(view.child_value_off, P, A),
               "value1": f32(view.value1_off, P, A), "keep": i32(view.keep_off, P, W),
               "backup": f32(view.backup_off, P, W), "best_slot": i32(view.best_slot_off, P)}
        if level == 0:
            arr["robot"], arr["humans"], arr["humans_per"] = L["robot"], L["humans"], 1
            if view.reward_clip_off >= 0 and L.get("roots_are_joint_states"):
                # joint-state roots of a clipped search: the rewards the root's action_clip selected on (tensor-born reading,
                # model_predictive_rl.py:216-218,246-248); "reward" holds the float64 reading the root values use (:226)
                arr["reward_clip"] = f32(view.reward_clip_off, P, A)
        else:
            arr["robot"] = f32(view.robot_off, P, 9)
            arr["humans"] = f32(view.humans_off, P // W, H, 5)
            arr["humans_per"] = W
        return arr

    def best_trajectory(self, b=0):
        """[(robot (1,1,9), humans (1,H,5)), action index | None, reward | None] along the best branch
        of root scene `b` -- the content of ModelPredictiveRL.traj (model_predictive_rl.py:231,298-302)."""
        traj, p = [], b
        W = self.kept_per_node
        for lvl in range(self.planning_depth):
            arr = self.level_arrays(lvl)
            slot = int(arr["best_slot"][p])
            a = int(arr["keep"][p, slot])
            state = (arr["robot"][p].reshape(1, 1, 9).clone(),
                     arr["humans"][p // arr["humans_per"]].unsqueeze(0).clone())
            traj.append((state, a, float(arr["reward"][p, a])))
            last = (arr["child_robot"][p, a].reshape(1, 1, 9).clone(), arr["humans_next"][p].unsqueeze(0).clone())
            p = p * W + slot
        traj.append((last, None, None))
        return traj


class GcnSearch:
    """One-step lookahead over the rotation-major action table with the path-G ValueNetwork."""

    def __init__(self, value_network, actions, kinematics="holonomic", time_step=0.25, gamma=0.9, contraction_dtype="f32"):
        if contraction_dtype not in ("f32", "bf16x6"):
            raise ValueError("path G contraction mode %r: 'f32' (the reference's arithmetic) or 'bf16x6'" % (contraction_dtype,))
        self.contraction_dtype = contraction_dtype     # "bf16x6": the graph's weight products as six bf16 MFMA terms (GcnPlanner, ABI 8)
        self.model = value_network
        self.actions_np = np.ascontiguousarray(np.asarray(actions, dtype=np.float64))
        self.kinematics = kinematics
        self.time_step = float(time_step)
        self.gamma = float(gamma)
        self._dev_tables = {}
        self._ws = _Workspace()

    def search(self, robot, humans, roots64=None):
        """robot (B,9), humans (B,H,5) -> (action_values (B,A) fp32, best_action (B,) int32), device tensors.
        `roots64`: the float64 (robot, humans) the fp32 arrays were rounded from (propagate / compute_reward read them)."""
        robot = _require_device_tensor(robot, "robot states")
        humans = _require_device_tensor(humans, "human states")
        B, H, A = robot.shape[0], humans.shape[1], self.actions_np.shape[0]
        dev = robot.device
        key = str(dev)
        if key not in self._dev_tables:
            self._dev_tables[key] = torch.tensor(self.actions_np, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            pl = nat.GcnPlanner()
            with batched_transposes():
                pl.graph = self.model.descriptor()
                pl.value_head = self.model.head_descriptor()
            pl.kinematics = nat.KINEMATICS[self.kinematics]
            pl.num_actions = A
            pl.time_step = self.time_step
            pl.gamma = self.gamma
            pl.contraction_dtype = nat.CONTRACTION_DTYPES[self.contraction_dtype]
            pl.actions = self._dev_tables[key].data_ptr()
            if roots64 is not None:
                r64, h64 = _check_roots64(roots64, B, H)
                pl.root_robot_f64, pl.root_humans_f64 = r64.data_ptr(), h64.data_ptr()
            lib = nat.lib()
            ws = self._ws.get(lib.gcn_predict_workspace_bytes(B, H, A), dev)
            vals = torch.empty(B, A, dtype=torch.float32, device=dev)
            best = torch.empty(B, dtype=torch.int32, device=dev)
            best_value = torch.empty(B, dtype=torch.float32, device=dev)
            rc = lib.gcn_predict_f32(C.byref(pl), robot.data_ptr(), humans.data_ptr(), B, H, ws.data_ptr(), ws.numel(),
                                     vals.data_ptr(), best.data_ptr(), best_value.data_ptr(), _stream())
        nat.check(rc, "gcn_predict_f32")
        self.last_best_value = best_value        # value of the chosen action per root, from the argmax kernel itself (no gather launches)
        return vals, best


def rotate(joint14, kinematics="holonomic"):
    """(R,14) -> (R,13) pairwise relation features on device (CADRL.rotate, cadrl.py:241-276)."""
    joint14 = _require_device_tensor(joint14, "joint states")
    out = torch.empty(joint14.shape[0], 13, dtype=torch.float32, device=joint14.device)
    with torch.cuda.device(joint14.device):
        rc = nat.lib().gcn_rotate_f32(joint14.data_ptr(), out.data_ptr(), joint14.shape[0],
                                      nat.KINEMATICS[kinematics], _stream())
    nat.check(rc, "gcn_rotate_f32")
    return out


def prepare_scenes(robot, humans, actions, kinematics="holonomic", time_step=0.25, roots64=None):
    """gcn_prepare_f32: robot (B,9), humans (B,H,5), actions (A,2) float64 -> (self6 (B*A,6), hum7 (B*A,H,7), reward (B*A,)) --
    CADRL.propagate + constant-velocity humans + CADRL.rotate + compute_reward for every (root, action) pair
    (cadrl.py:113-138,241-276, multi_human_rl.py:46-51,73-96)."""
    robot = _require_device_tensor(robot, "robot states")
    humans = _require_device_tensor(humans, "human states")
    B, H = robot.shape[0], humans.shape[1]
    dev = robot.device
    table = torch.as_tensor(np.asarray(actions, dtype=np.float64).reshape(-1, 2), dtype=torch.float64).to(dev).contiguous()
    A = table.shape[0]
    pl = nat.GcnPlanner()
    pl.kinematics, pl.num_actions, pl.time_step, pl.actions = nat.KINEMATICS[kinematics], A, float(time_step), table.data_ptr()
    if roots64 is not None:
        r64, h64 = _check_roots64(roots64, B, H)
        pl.root_robot_f64, pl.root_humans_f64 = r64.data_ptr(), h64.data_ptr()
    self6 = torch.empty(B * A, 6, dtype=torch.float32, device=dev)
    hum7 = torch.empty(B * A, H, 7, dtype=torch.float32, device=dev)
    reward = torch.empty(B * A, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = nat.lib().gcn_prepare_f32(C.byref(pl), robot.data_ptr(), humans.data_ptr(), B, H, self6.data_ptr(), hum7.data_ptr(),
                                       reward.data_ptr(), _stream())
        nat.check(rc, "gcn_prepare_f32")
        torch.cuda.current_stream(dev).synchronize()              # the action table is a temporary
    return self6, hum7, reward


# --------------------------------------------------------------------------------------------------
# multi-GPU: shard the root scenes, one all-gather of the results
# --------------------------------------------------------------------------------------------------
def shard_bounds(n_items, world_size, rank):
    """Contiguous, balanced split: the first (n % world) ranks get one extra item."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedRollout:
    """Runs `search_fn(robot_shard, humans_shard) -> (best_action (b,), best_value (b,))` on this
    rank's contiguous slice of the roots and all-gathers `[action, value]` rows from every rank.

    The only collective is one `all_gather_into_tensor` of (ceil(B/world), 2) fp32 per rank -- a few
    KB, latency-bound; on MI355X this is RCCL over xGMI (`backend="nccl"`), in the CPU tests gloo."""

    def __init__(self, search_fn, group=None, search_into=None):
        """search_fn(robot, humans) -> (action (n,), value (n,));  optional search_into(robot, humans, act_out, val_out)
        writes int32 actions / fp32 values straight into the exchange buffer (no packing kernels)."""
        import torch.distributed as dist
        self.dist = dist
        self.search_fn = search_fn
        self.search_into = search_into
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self._static = None

    def use_static_buffers(self, total, device, depth=2):
        """Pre-allocate `depth` exchange buffers for steps of `total` roots and hand them out in turn, instead of a fresh
        buffer per step: a search captured in a hipGraph writes its results to FIXED addresses, so a pipelined loop (one
        exchange in flight under the next search) needs the buffers to exist before the capture.  Returns the list of
        (act_out (n,) int32, val_out (n,) fp32) views this rank's search must fill, in the order of use."""
        per = -(-total // self.world)
        lo, hi = shard_bounds(total, self.world, self.rank)
        n = hi - lo
        packed = [torch.zeros(2, per, dtype=torch.float32, device=device) for _ in range(depth)]
        gathered = [torch.empty(self.world * 2, per, dtype=torch.float32, device=device) for _ in range(depth)]
        self._static = {"total": total, "packed": packed, "gathered": gathered, "turn": 0}
        return [(p[0, :n].view(torch.int32), p[1, :n]) for p in packed]

    def run(self, robot, humans):
        """robot (B,9), humans (B,H,5): the FULL root batch (identical on every rank).
        Returns (best_action (B,) int64, best_value (B,) fp32) on the device of the inputs.  This convenience entry converts the
        indices once to torch's index type (gather / fancy indexing want int64); `run_local` / `launch_local` keep the search's
        own int32 indices, which is what travels through the exchange."""
        B = robot.shape[0]
        lo, hi = shard_bounds(B, self.world, self.rank)
        act, val = self.run_local(robot[lo:hi], humans[lo:hi], B)
        return act.long(), val

    def run_local(self, robot_shard, humans_shard, total):
        """Same, when each rank already holds only its shard (`total` = global root count)."""
        return self.launch_local(robot_shard, humans_shard, total).result()

    def launch_local(self, robot_shard, humans_shard, total):
        """Enqueue this rank's search and the exchange WITHOUT waiting for the exchange: returns a handle whose
        `.result()` gives (best_action, best_value) of all roots.  Calling `.result()` one step later lets the
        (latency-bound) all-gather of step i run on RCCL's stream underneath the search of step i+1."""
        per = -(-total // self.world)
        n = robot_shard.shape[0]
        if not self.active:                      # one rank: nothing to exchange, hand the search's outputs through
            if n == 0:
                return _Exchange(None, None, total, 1, (torch.zeros(0, dtype=torch.int32, device=robot_shard.device),
                                                        torch.zeros(0, dtype=torch.float32, device=robot_shard.device)))
            act, val = self.search_fn(robot_shard, humans_shard)
            return _Exchange(None, None, total, 1, (act, val))          # the search's own int32 indices: no conversion kernel
        # exchange buffer: row 0 = action indices (int32 bit patterns), row 1 = values
        st = self._static if self._static is not None and self._static["total"] == total else None
        if st is not None:
            turn = st["turn"]
            st["turn"] = (turn + 1) % len(st["packed"])
            packed = st["packed"][turn]                      # tail beyond n stays zero from allocation
        else:
            packed = torch.empty(2, per, dtype=torch.float32, device=robot_shard.device)
            if n < per:
                packed[:, n:].zero_()
        if n > 0:
            act_out, val_out = packed[0, :n].view(torch.int32), packed[1, :n]
            if self.search_into is not None:
                self.search_into(robot_shard, humans_shard, act_out, val_out)
            else:
                act, val = self.search_fn(robot_shard, humans_shard)
                act_out.copy_(act)
                val_out.copy_(val)
        gathered = st["gathered"][turn] if st is not None else \
            torch.empty(self.world * 2, per, dtype=torch.float32, device=packed.device)           # rank-major concatenation
        work = self.dist.all_gather_into_tensor(gathered, packed, group=self.group, async_op=True)
        return _Exchange(work, gathered, total, self.world, None, keep=packed)


class _Exchange:
    """Handle of one sharded step: the pending all-gather and how to unpack it."""

    def __init__(self, work, gathered, total, world, ready, keep=None):
        self.work, self.gathered, self.total, self.world, self.ready, self.keep = work, gathered, total, world, ready, keep

    def wait(self):
        """Order the caller's stream after the collective (the host does not block).  `result()` implies it."""
        if self.work is not None:
            self.work.wait()
            self.work = self.keep = None
        return self

    def result(self):
        if self.ready is None:
            self.wait()
            g = self.gathered.view(self.world, 2, -1)
            per = g.shape[2]
            if per * self.world == self.total:   # equal shards: rank r's block is already at its global position
                acts, vals = g[:, 0, :].reshape(-1), g[:, 1, :].reshape(-1)
            else:
                sizes = [shard_bounds(self.total, self.world, r) for r in range(self.world)]
                acts = torch.cat([g[r, 0, :hi - lo] for r, (lo, hi) in enumerate(sizes)])
                vals = torch.cat([g[r, 1, :hi - lo] for r, (lo, hi) in enumerate(sizes)])
            self.ready = (acts.contiguous().view(torch.int32), vals)
        return self.ready
