"""Network modules of the RGL hot path, with the reference's constructor signatures, parameter
names and shapes (so its checkpoints load unchanged) and HIP kernels behind `forward`.

Mirrors (reference paths): crowd_nav/policy/helpers.py:5-13 (`mlp`), graph_model.py:10-130 (`RGL`),
value_estimator.py:5-20 (`ValueEstimator`), state_predictor.py:7-118 (`StatePredictor`,
`LinearStatePredictor`), gcn.py:11-128 (`ValueNetwork`).

`forward` runs on the MI355X through librgl_hip.so and nowhere else: CPU tensors, a missing
library or a request for gradients raise -- there is no eager/CPU fallback to hide behind.
"""
import ctypes as C
import itertools

import torch
import torch.nn as nn

from . import _native as nat


def mlp(input_dim, mlp_dims, last_relu=False):
    """Sequential of Linear(+ReLU); ReLU after the last layer only when `last_relu`."""
    widths = [input_dim] + list(mlp_dims)
    mods = []
    for i in range(len(widths) - 1):
        mods.append(nn.Linear(widths[i], widths[i + 1]))
        if last_relu or i + 2 < len(widths):
            mods.append(nn.ReLU())
    return nn.Sequential(*mods)


# --------------------------------------------------------------------------------------------------
# parameter packing: k-major device copies of Linear weights + the ABI descriptor structs
# --------------------------------------------------------------------------------------------------
def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_device_tensor(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise nat.NativeLibraryError(
            "%s must be a CUDA(HIP) tensor: the RGL forward runs only on the MI355X kernels (no CPU path)" % what)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (what, t.dtype))
    return t.contiguous()


def _flat_params(module):
    """The parameters of `module` and its sub-modules, like list(module.parameters()) but without walking the module tree on every
    call (nn.Module.parameters() re-discovers the sub-modules each time: ~115 such walks were a fifth of an eager training step's
    host time).  The list of sub-modules is kept on the module; the Parameter objects are read from the sub-modules' own dicts each
    time, so replaced parameters and moved storages are seen.  A REPLACED SUB-MODULE (`seq[2] = nn.Linear(...)`, `add_module`) is
    seen too: the kept list carries a fingerprint -- the identity of every child in every sub-module's `_modules` dict -- that is
    checked on each call (a walk over a few small dicts, no generator recursion) and the list is rebuilt when it differs."""
    ent = module.__dict__.get("_rgl_submodules")
    if ent is not None:
        mods, print_ = ent
        if tuple(id(c) for m in mods for c in m._modules.values()) != print_:
            ent = None
    if ent is None:
        mods = list(module.modules())
        module.__dict__["_rgl_submodules"] = (mods, tuple(id(c) for m in mods for c in m._modules.values()))
    return [p for m in mods for p in m._parameters.values() if p is not None]


def _needs_grad(params):
    return torch.is_grad_enabled() and any(p.requires_grad for p in params)


class _Spec(object):
    """What autograd.GraphFunction needs to re-create the descriptors and to split the gradient slab."""

    def __init__(self, graph_module, value_seq=None, motion_seq=None, value_desc=None, motion_desc=None, want_H=False,
                 want_A=False, detach_graph=False):
        self.graph_module = graph_module
        self.value_desc, self.motion_desc = value_desc, motion_desc
        self.want_H, self.want_A, self.detach_graph = want_H, want_A, detach_graph
        gm = graph_module
        if gm.similarity_function not in nat.SIMILARITY:
            raise NotImplementedError(gm.similarity_function)
        self.params, self.param_shapes = [], []

        def add_mlp(seq):
            for m in seq:
                if isinstance(m, nn.Linear):
                    self.params += [m.weight, m.bias]
                    self.param_shapes += [("linear_w", tuple(m.weight.shape)), ("vector", tuple(m.bias.shape))]
        add_mlp(gm.w_r)
        add_mlp(gm.w_h)
        if gm.similarity_function == "embedded_gaussian":
            self.params.append(gm.w_a)
            self.param_shapes.append(("matrix", tuple(gm.w_a.shape)))
        elif gm.similarity_function == "concatenation":
            add_mlp(gm.w_a)                       # the pair MLP 2X -> 2X -> 1
        for w in gm._graph_weights():
            self.params.append(w)
            self.param_shapes.append(("matrix", tuple(w.shape)))
        self.n_graph_params = len(self.params)          # the graph model's share of `params` (the heads follow)
        if value_seq is not None:
            add_mlp(value_seq)
        if motion_seq is not None:
            add_mlp(motion_seq)

    def graph(self):
        return self.graph_module.descriptor()

    def value_head(self):
        return None if self.value_desc is None else self.value_desc()

    def motion_head(self):
        return None if self.motion_desc is None else self.motion_desc()


def _graph_apply(spec, robot, humans):
    from .autograd import GraphFunction
    robot = _require_device_tensor(robot, "robot states")
    humans = _require_device_tensor(humans, "human states")
    outs = GraphFunction.apply(spec, robot, humans, *spec.params)
    keys = [k for k in ("H", "value", "humans_next", "A") if
            (k == "H" and spec.want_H) or (k == "value" and spec.value_desc is not None) or
            (k == "humans_next" and spec.motion_desc is not None) or (k == "A" and spec.want_A)]
    return dict(zip(keys, outs))


def _linears(seq):
    return [m for m in seq if isinstance(m, nn.Linear)]


def _seq_last_relu(seq):
    mods = list(seq)
    return len(mods) > 0 and isinstance(mods[-1], nn.ReLU)


# (torch weight (out,in), k-major destination, rows, cols) collected by pack_mlp.  Process-global and unsynchronised, like torch's
# own current-stream state: descriptors are packed from one thread at a time (the thread that runs the forwards); a failing pack
# takes its own entries back out (_PackCache.get, batched_transposes.__exit__).
_PENDING_TRANSPOSES = []
_BATCH_DEPTH = [0]


class batched_transposes(object):
    """`with batched_transposes():` around the code that gathers SEVERAL descriptors (graph, value head, motion head of one
    forward; the four of a search): their Linear weights are transposed by one launch when the block ends instead of one per
    descriptor.  Nothing may consume a descriptor inside the block."""

    def __enter__(self):
        _BATCH_DEPTH[0] += 1
        return self

    def __exit__(self, exc_type, *exc):
        _BATCH_DEPTH[0] -= 1
        if _BATCH_DEPTH[0] == 0:
            if exc_type is None:
                flush_transposes()
            else:
                del _PENDING_TRANSPOSES[:]          # the block failed: nothing it queued may be launched later by somebody else
        return False



def flush_transposes():
    """One rgl_transpose_many_f32 launch for every Linear weight packed since the last flush (a descriptor holds several, and
    training rebuilds it after every optimizer step: one launch per weight was most of a training step's launches).  Called by
    _PackCache.get as soon as a descriptor is built, i.e. before anything can consume it, on the current stream."""
    if not _PENDING_TRANSPOSES:
        return
    jobs = (nat.RglTransposeJob * len(_PENDING_TRANSPOSES))()
    by_dev = {}
    for i, (w, wt, rows, cols) in enumerate(_PENDING_TRANSPOSES):
        by_dev.setdefault(w.device, []).append(i)
        jobs[i].src, jobs[i].dst, jobs[i].rows, jobs[i].cols = w.data_ptr(), wt.data_ptr(), rows, cols
    pending = list(_PENDING_TRANSPOSES)
    del _PENDING_TRANSPOSES[:]
    lib = nat.lib()
    for dev, idx in by_dev.items():
        sub = (nat.RglTransposeJob * len(idx))(*[jobs[i] for i in idx])
        with torch.cuda.device(dev):
            nat.check(lib.rgl_transpose_many_f32(sub, len(idx), _stream()), "rgl_transpose_many_f32")
    del pending


def pack_mlp(seq, keep, reuse=None):
    """nn.Sequential[Linear, ReLU...] -> RglMlp; transposed weights are appended to `keep`.  `reuse` (dict, optional) holds the
    transposed buffers of an earlier pack: they are overwritten in place, so device pointers handed out earlier (descriptors
    baked into a captured hipGraph) stay valid and see the new weights."""
    lins = _linears(seq)
    if not 1 <= len(lins) <= nat.MAX_MLP_LAYERS:
        raise ValueError("MLP depth %d outside 1..%d" % (len(lins), nat.MAX_MLP_LAYERS))
    m = nat.RglMlp()
    m.n_layers = len(lins)
    m.last_relu = int(_seq_last_relu(seq))
    m.dims[0] = lins[0].in_features
    for l, lin in enumerate(lins):
        w = _require_device_tensor(lin.weight.detach(), "MLP weight")
        b = _require_device_tensor(lin.bias.detach(), "MLP bias")
        rk = (id(lin), lin.in_features, lin.out_features, str(w.device))
        wt = None if reuse is None else reuse.get(rk)
        if wt is None:
            wt = torch.empty(lin.in_features, lin.out_features, device=w.device, dtype=torch.float32)
            if reuse is not None:
                reuse[rk] = wt
        _PENDING_TRANSPOSES.append((w, wt, lin.out_features, lin.in_features))     # launched together: flush_transposes()
        keep.extend([wt, b])
        m.dims[l + 1] = lin.out_features
        m.weight[l] = wt.data_ptr()
        m.bias[l] = b.data_ptr()
    return m


class _GraphCore(nn.Module):
    """What RGL and path G's ValueNetwork share: flags, parameter access, descriptor packing."""

    def _graph_weights(self):
        raise NotImplementedError

    def _pack_graph(self, keep, reuse=None):
        g = nat.RglGraph()
        g.w_r = pack_mlp(self.w_r, keep, reuse)
        g.w_h = pack_mlp(self.w_h, keep, reuse)
        g.x_dim = self.X_dim
        ws = self._graph_weights()
        g.num_layer = len(ws)
        if self.similarity_function not in nat.SIMILARITY:
            raise NotImplementedError(self.similarity_function)
        g.similarity = nat.SIMILARITY[self.similarity_function]
        g.layerwise_graph = int(bool(self.layerwise_graph))
        g.skip_connection = int(bool(self.skip_connection))
        if self.similarity_function == "embedded_gaussian":
            wa = _require_device_tensor(self.w_a.detach(), "w_a")
            keep.append(wa)
            g.w_a = wa.data_ptr()
        elif self.similarity_function == "concatenation":
            g.w_a_mlp = pack_mlp(self.w_a, keep, reuse)
        for l, w in enumerate(ws):
            wt = _require_device_tensor(w.detach(), "GCN weight")
            if tuple(wt.shape) != (self.X_dim, self.X_dim):
                raise ValueError("GCN layer weights must be (X_dim, X_dim) on the HIP path, got %s" % (tuple(wt.shape),))
            keep.append(wt)
            g.Ws[l] = wt.data_ptr()
        return g


_PACK_SERIAL = itertools.count(1)   # process-wide: a pack serial is never shared by two caches or two parameter states


class _PackCache:
    """Descriptor cache keyed on (storage pointer, version) of every parameter involved."""

    def __init__(self):
        self.key = None
        self.value = None
        self.keep = None
        self.buffers = {}         # transposed Linear weights, refreshed IN PLACE when parameters change (stable device pointers)
        self.epoch = 0            # serial of the latest (re)pack, unique in the process: dependants (TreeSearch's weight image)
                                  # compare it to notice a refresh -- or another module put in this one's place

    def get(self, modules, build):
        key = tuple((p.data_ptr(), p._version, p.device.index) for m in modules for p in _flat_params(m))
        if key != self.key:
            keep = []
            n_pending = len(_PENDING_TRANSPOSES)
            try:
                self.value = build(keep, self.buffers)
            except BaseException:
                # a descriptor that failed half-way (a CPU tensor in a later layer ...) must not leave its first layers queued: the
                # next unrelated flush would launch them, on whatever stream is current then, and the queue pins their tensors
                del _PENDING_TRANSPOSES[n_pending:]
                raise
            if _BATCH_DEPTH[0] == 0:
                flush_transposes()
            self.keep = keep
            self.key = key
            self.epoch = next(_PACK_SERIAL)
        return self.value

    # The cache holds ctypes descriptor structs (raw device pointers), which can be neither copied nor pickled and
    # would be wrong for a copy anyway (its parameters live elsewhere).  A copied / unpickled module starts with an
    # empty cache and repacks on its first forward: `copy.deepcopy(model)` (Trainer.update_target_model upstream,
    # crowd_nav/utils/trainer.py:41,187) and `torch.save(model)` work after any number of forwards.
    def __deepcopy__(self, memo):
        return _PackCache()

    def __copy__(self):
        return _PackCache()

    def __reduce__(self):
        return (_PackCache, ())


def invalidate_packed_weights(*modules):
    """Forget the packed descriptors (k-major weight copies, weight images) of `modules` and their sub-modules: the next forward
    repacks from the current parameters.  Needed after parameters were changed WITHOUT autograd's version counters noticing --
    i.e. after replaying a hipGraph that contains an optimizer step (a captured training step, tools/train_step_time.py --graph):
    replays mutate the parameters in place but do not bump `tensor._version`, which is what the caches key on."""
    for top in modules:
        if not isinstance(top, nn.Module):                   # LinearStatePredictor is a plain object (state_predictor.py:63), no weights
            continue
        for m in top.modules():
            m.__dict__.pop("_rgl_submodules", None)          # and the kept sub-module lists (_flat_params)
            for name in ("_cache", "_head_cache"):
                c = getattr(m, name, None)
                if isinstance(c, _PackCache):
                    c.key = None


def graph_forward(graph, value_head, motion_head, robot, humans, scenes_per_crowd=1, want_H=False, want_A=False):
    """Thin wrapper over rgl_graph_forward_f32.  robot (S,rd), humans (S/spc,H,hd) -> dict of outputs."""
    robot = _require_device_tensor(robot, "robot states")
    humans = _require_device_tensor(humans, "human states")
    S, H = robot.shape[0], humans.shape[1]
    if humans.shape[0] * scenes_per_crowd != S:
        raise ValueError("humans batch %d x %d != scenes %d" % (humans.shape[0], scenes_per_crowd, S))
    N, X = H + 1, graph.x_dim
    dev = robot.device
    out = {}
    Hp = Ap = vp = mp = None
    if want_H:
        out["H"] = torch.empty(S, N, X, device=dev, dtype=torch.float32)
        Hp = out["H"].data_ptr()
    if want_A:
        out["A"] = torch.empty(S, N, N, device=dev, dtype=torch.float32)
        Ap = out["A"].data_ptr()
    vh = mh = None
    if value_head is not None:
        out["value"] = torch.empty(S, 1, device=dev, dtype=torch.float32)
        vp = out["value"].data_ptr()
        vh = C.byref(value_head)
    if motion_head is not None:
        od = motion_head.dims[motion_head.n_layers]
        out["humans_next"] = torch.empty(S, H, od, device=dev, dtype=torch.float32)
        mp = out["humans_next"].data_ptr()
        mh = C.byref(motion_head)
    with torch.cuda.device(dev):
        lib = nat.lib()
        ws, wp, wbytes = None, None, 0
        if not (want_H or want_A) and S > 0:
            # values / next humans only: the one-wave-per-scene MFMA kernel where it covers the model (scratch for the embeddings)
            wbytes = lib.rgl_graph_forward_workspace_bytes(C.byref(graph), vh, mh, S, scenes_per_crowd, H)
            if wbytes:
                ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
                wp = ws.data_ptr()
        if nat.poison_workspaces():
            for t in [ws] + list(out.values()):
                if t is not None:
                    t.fill_(255) if t.dtype == torch.uint8 else t.fill_(float("nan"))
        rc = lib.rgl_graph_forward_f32(C.byref(graph), vh, mh, robot.data_ptr(), humans.data_ptr(), S,
                                       scenes_per_crowd, H, Hp, Ap, vp, mp, wp, wbytes, _stream())
    nat.check(rc, "rgl_graph_forward_f32")
    return out


# --------------------------------------------------------------------------------------------------
# path M modules
# --------------------------------------------------------------------------------------------------
class RGL(_GraphCore):
    def __init__(self, config, robot_state_dim, human_state_dim):
        super().__init__()
        gc = config.gcn
        self.multiagent_training = gc.multiagent_training
        self.num_layer = gc.num_layer
        self.X_dim = gc.X_dim
        self.similarity_function = gc.similarity_function
        self.layerwise_graph = gc.layerwise_graph
        self.skip_connection = gc.skip_connection
        self.robot_state_dim = robot_state_dim
        self.human_state_dim = human_state_dim
        self.w_r = mlp(robot_state_dim, gc.wr_dims, last_relu=True)
        self.w_h = mlp(human_state_dim, gc.wh_dims, last_relu=True)
        if self.similarity_function == "embedded_gaussian":
            self.w_a = nn.Parameter(torch.randn(self.X_dim, self.X_dim))
        elif self.similarity_function == "concatenation":
            self.w_a = mlp(2 * self.X_dim, [2 * self.X_dim, 1], last_relu=True)
        self.Ws = nn.ParameterList()
        for i in range(self.num_layer):
            out_dim = gc.final_state_dim if (i == self.num_layer - 1 and i > 0) else self.X_dim
            self.Ws.append(nn.Parameter(torch.randn(self.X_dim, out_dim)))
        self._A_dev = None
        self._cache = _PackCache()

    def _graph_weights(self):
        return list(self.Ws)

    def descriptor(self):
        return self._cache.get([self], self._pack_graph)

    @property
    def A(self):
        """Adjacency of the first scene of the last forward (host copy made on demand, not per call)."""
        return None if self._A_dev is None else self._A_dev.cpu().numpy()

    def forward(self, state):
        robot, humans = state
        if robot.dim() != 3 or humans.dim() != 3:
            raise AssertionError("states must be (batch, agents, features)")
        if _needs_grad(_flat_params(self)):
            out = _graph_apply(_Spec(self, want_H=True, want_A=True), robot.reshape(robot.shape[0], -1), humans)
        else:
            out = graph_forward(self.descriptor(), None, None, robot.reshape(robot.shape[0], -1), humans,
                                want_H=True, want_A=True)
        self._A_dev = out["A"][0].detach()
        return out["H"]


class ValueEstimator(nn.Module):
    def __init__(self, config, graph_model):
        super().__init__()
        self.graph_model = graph_model
        self.value_network = mlp(config.gcn.X_dim, config.model_predictive_rl.value_network_dims)
        self._cache = _PackCache()

    def head_descriptor(self):
        return self._cache.get([self.value_network], lambda keep, reuse: pack_mlp(self.value_network, keep, reuse))

    def forward(self, state):
        robot, humans = state
        assert len(robot.shape) == 3 and len(humans.shape) == 3
        if _needs_grad(_flat_params(self)):
            spec = _Spec(self.graph_model, value_seq=self.value_network, value_desc=self.head_descriptor)
            return _graph_apply(spec, robot.reshape(robot.shape[0], -1), humans)["value"]
        out = graph_forward(self.graph_model.descriptor(), self.head_descriptor(), None,
                            robot.reshape(robot.shape[0], -1), humans)
        return out["value"]


class StatePredictor(nn.Module):
    def __init__(self, config, graph_model, time_step):
        super().__init__()
        self.trainable = True
        self.kinematics = config.action_space.kinematics
        self.graph_model = graph_model
        self.human_motion_predictor = mlp(config.gcn.X_dim, config.model_predictive_rl.motion_predictor_dims)
        self.time_step = time_step
        self._cache = _PackCache()

    def head_descriptor(self):
        return self._cache.get([self.human_motion_predictor],
                               lambda keep, reuse: pack_mlp(self.human_motion_predictor, keep, reuse))

    def forward(self, state, action, detach=False):
        robot, humans = state
        assert len(robot.shape) == 3 and len(humans.shape) == 3
        if _needs_grad(_flat_params(self)):
            spec = _Spec(self.graph_model, motion_seq=self.human_motion_predictor, motion_desc=self.head_descriptor,
                         detach_graph=bool(detach))
            out = _graph_apply(spec, robot.reshape(robot.shape[0], -1), humans)
        else:
            out = graph_forward(self.graph_model.descriptor(), None, self.head_descriptor(),
                                robot.reshape(robot.shape[0], -1), humans)
        next_robot = None if action is None else self.compute_next_state(robot, action)
        return [next_robot, out["humans_next"]]

    def compute_next_state(self, robot_state, action):
        return next_robot_state(robot_state, action, self.kinematics, self.time_step)


def next_robot_state(robot_state, action, kinematics, time_step):
    """Kinematic update of the robot row; unlike the reference it is not limited to batch 1."""
    nxt = robot_state.clone()
    if kinematics == "holonomic":
        nxt[..., 0] = nxt[..., 0] + float(action.vx * time_step)
        nxt[..., 1] = nxt[..., 1] + float(action.vy * time_step)
        nxt[..., 2] = float(action.vx)
        nxt[..., 3] = float(action.vy)
    else:
        # the reference adds the rotation to slot 7 (v_pref) rather than slot 8 (theta); kept for parity
        nxt[..., 7] = nxt[..., 7] + float(action.r)
        nxt[..., 0] = nxt[..., 0] + torch.cos(nxt[..., 7]) * float(action.v * time_step)
        nxt[..., 1] = nxt[..., 1] + torch.sin(nxt[..., 7]) * float(action.v * time_step)
        nxt[..., 2] = torch.cos(nxt[..., 7]) * float(action.v)
        nxt[..., 3] = torch.sin(nxt[..., 7]) * float(action.v)
    return nxt


class LinearStatePredictor(object):
    def __init__(self, config, time_step):
        self.trainable = False
        self.kinematics = config.action_space.kinematics
        self.time_step = time_step

    def __call__(self, state, action):
        robot, humans = state
        assert len(robot.shape) == 3 and len(humans.shape) == 3
        return [next_robot_state(robot, action, self.kinematics, self.time_step),
                self.linear_motion_approximator(humans)]

    @staticmethod
    def linear_motion_approximator(human_states):
        nxt = human_states.clone()
        nxt[..., 0] = nxt[..., 0] + nxt[..., 2]      # no time-step factor, as upstream
        nxt[..., 1] = nxt[..., 1] + nxt[..., 3]
        return nxt


# --------------------------------------------------------------------------------------------------
# path G module
# --------------------------------------------------------------------------------------------------
class ValueNetwork(_GraphCore):
    def __init__(self, input_dim, self_state_dim, num_layer, X_dim, wr_dims, wh_dims, final_state_dim,
                 gcn2_w1_dim, planning_dims, similarity_function, layerwise_graph, skip_connection):
        super().__init__()
        self.similarity_function = similarity_function
        self.self_state_dim = self_state_dim
        self.human_state_dim = input_dim - self_state_dim
        self.num_layer = num_layer
        self.X_dim = X_dim
        self.layerwise_graph = layerwise_graph
        self.skip_connection = skip_connection
        self.w_r = mlp(self_state_dim, wr_dims, last_relu=True)
        self.w_h = mlp(self.human_state_dim, wh_dims, last_relu=True)
        if similarity_function == "embedded_gaussian":
            self.w_a = nn.Parameter(torch.randn(X_dim, X_dim))
        elif similarity_function == "concatenation":
            self.w_a = mlp(2 * X_dim, [2 * X_dim, 1], last_relu=True)
        if num_layer == 1:
            self.w1 = nn.Parameter(torch.randn(X_dim, final_state_dim))
        elif num_layer == 2:
            self.w1 = nn.Parameter(torch.randn(X_dim, gcn2_w1_dim))
            self.w2 = nn.Parameter(torch.randn(gcn2_w1_dim, final_state_dim))
        else:
            raise NotImplementedError
        self.value_net = mlp(final_state_dim, planning_dims)
        self._A_dev = None
        self._cache = _PackCache()
        self._head_cache = _PackCache()

    def _graph_weights(self):
        return [self.w1] if self.num_layer == 1 else [self.w1, self.w2]

    def _pack_graph(self, keep, reuse=None):
        g = super()._pack_graph(keep, reuse)
        if self.num_layer == 1:
            g.skip_connection = 0          # the one-layer variant never adds the skip (gcn.py:108-110)
        return g

    def descriptor(self):
        return self._cache.get([self], self._pack_graph)

    def head_descriptor(self):
        return self._head_cache.get([self.value_net], lambda keep, reuse: pack_mlp(self.value_net, keep, reuse))

    @property
    def A(self):
        return None if self._A_dev is None else self._A_dev.cpu().numpy()

    def forward(self, state_input):
        state = state_input[0] if isinstance(state_input, tuple) else state_input
        state = _require_device_tensor(state, "rotated joint states")
        d = self.self_state_dim
        if _needs_grad(_flat_params(self)):
            spec = _Spec(self, value_seq=self.value_net, value_desc=self.head_descriptor, want_A=True)
            out = _graph_apply(spec, state[:, 0, :d].contiguous(), state[:, :, d:].contiguous())
        else:
            out = graph_forward(self.descriptor(), self.head_descriptor(), None, state[:, 0, :d].contiguous(),
                                state[:, :, d:].contiguous(), want_A=True)
        self._A_dev = out["A"][0].detach()
        return out["value"]
