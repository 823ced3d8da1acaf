"""Trainers with the reference's contract whose optimisation step is ONE replayed hipGraph.

Mirrors crowd_nav/utils/trainer.py: `MPRLTrainer` (:10-161, path M: value update against a frozen target copy + state-predictor
update, two optimizers) and `VNRLTrainer` (:164-250, path G) -- same constructor arguments, `update_target_model`,
`set_learning_rate`, `optimize_epoch(num_epochs)`, `optimize_batch(num_batches, episode)`, same log lines / writer scalars, same
batch bookkeeping (including upstream's `batch_count > num_batches` off-by-one and its division by `num_batches`).
`register_trainers(module)` puts them into the namespace crowd_nav/train.py imports its trainers from.

What is different is how a step runs.  Upstream's step is ~60 launches of launch-sized work driven from Python (1.4-1.9 ms at batch
100 on this box, host-bound); here the whole step -- the batch gather, both forwards, the loss with its gradient and running sum
(rgl_mse_step_f32), both backwards (rgl_graph_backward_f32), the fused Adam updates, the repacking of the k-major weight copies -- is
captured once per batch shape into a hipGraph of 26 kernel nodes and replayed (0.21-0.23 ms): the library allocates only through
torch, never synchronises, repacks on the capture stream and records KERNELS only (a replayed hipMemsetAsync node was not reliably
ordered against its neighbours on ROCm 7.2: DESIGN.md section 4.3), which is what makes the capture legal and its replays
repeatable bit for bit.  Per batch the host copies the batch's indices (or the batch) into the graph's static input buffers and
launches the graph; losses accumulate on the device (float64, the sum upstream forms from `loss.data.item()`), read once per call.

Batches: upstream draws them with `DataLoader(memory, batch_size, shuffle=True)`.  A `memory` that offers `as_tensors()` (this
package's ReplayMemory: the experience as stacked device tensors) is sampled by index instead -- the SAME indices in the same
order, drawn from torch's global generator exactly as the DataLoader's RandomSampler would (`_ShuffledIndexBatches`, checked
against the real DataLoader on CPU) -- and gathered in one launch (rgl_gather_rows_f32); any other dataset, and a `data_loader` set
by the caller, goes through the DataLoader itself.  SGD and shapes that change from batch to batch run the same step eagerly.
"""
import copy
import ctypes as C
import logging
import os

import torch
import torch.nn as nn
import torch.optim as optim
from torch.utils.data import DataLoader

from . import _native as nat
from .nets import invalidate_packed_weights


class _ShuffledIndexBatches(object):
    """The index batches `DataLoader(dataset of n items, batch_size, shuffle=True)` would yield, consuming torch's global generator
    the way the loader does: one int64 draw when the iterator is made (_BaseDataLoaderIter's base seed), one when the sampler
    starts (RandomSampler's private generator), then `randperm(n)` from that generator cut into batches (the last one short)."""

    def __init__(self, n, batch_size):
        self.n, self.batch_size = int(n), int(batch_size)

    def __iter__(self):
        torch.empty((), dtype=torch.int64).random_()                                    # the loader iterator's base seed
        if self.n == 0:
            return
        g = torch.Generator()
        g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))         # RandomSampler.__iter__
        perm = torch.randperm(self.n, generator=g)
        for lo in range(0, self.n, self.batch_size):
            yield perm[lo:lo + self.batch_size]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gather_fields(fields, idx):
    """[f.index_select(0, idx) for f in fields]; device float32 fields of one device in ONE launch (rgl_gather_rows_f32: a replayed
    batch-100 step is launch-bound, and the gathers were five of its nodes), anything else through torch."""
    dev = idx.device
    if not (dev.type == "cuda" and idx.dtype == torch.int64 and idx.dim() == 1 and idx.is_contiguous() and len(fields) > 0 and
            all(f.device == dev and f.dtype == torch.float32 and f.is_contiguous() and f.dim() >= 1 and f.shape[0] > 0 and f[0].numel() > 0
                for f in fields)):
        return [f.index_select(0, idx) for f in fields]
    n = int(idx.shape[0])
    outs = [torch.empty((n,) + tuple(f.shape[1:]), dtype=torch.float32, device=dev) for f in fields]
    if n == 0:
        return outs
    jobs = (nat.RglGatherJob * len(fields))()
    for j, (f, o) in enumerate(zip(fields, outs)):
        jobs[j].src, jobs[j].dst, jobs[j].row_floats, jobs[j].src_rows = f.data_ptr(), o.data_ptr(), f[0].numel(), int(f.shape[0])
    with torch.cuda.device(dev):
        nat.check(nat.lib().rgl_gather_rows_f32(jobs, len(fields), idx.data_ptr(), n, _stream()), "rgl_gather_rows_f32")
    return outs


class _Batch(object):
    """The fields of one batch, as tensors (DataLoader path) or as (stacked memory fields, device indices) to be gathered."""

    def __init__(self, tensors=None, fields=None, idx=None):
        self._tensors, self._fields, self._idx = tensors, fields, idx

    def pick(self, which):
        if self._tensors is not None:
            return _Batch(tensors=[self._tensors[i] for i in which])
        return _Batch(fields=[self._fields[i] for i in which], idx=self._idx)

    def shapes(self):
        """Batch shapes, plus -- indexed batches -- the extent of the fields the gathers read from (a recorded index_select is
        only valid for the source extent it was recorded with)."""
        if self._tensors is not None:
            return tuple(tuple(t.shape) for t in self._tensors)
        return tuple((int(self._idx.shape[0]),) + tuple(f.shape[1:]) for f in self._fields) + (int(self._fields[0].shape[0]),)

    def tensors(self):
        if self._tensors is not None:
            return [t.contiguous() for t in self._tensors]
        return _gather_fields(self._fields, self._idx)

    def identity(self):
        """What a captured step bakes in besides the shapes: the addresses of the memory's stacked fields it gathers from."""
        return () if self._tensors is not None else tuple(f.data_ptr() for f in self._fields)

    def capture_inputs(self, fn):
        """(static input buffers, the function to record).  Tensor batches: the step's own inputs.  Indexed batches: ONE static
        buffer -- the batch's indices -- and the gathers become part of the recorded step, so a replayed batch costs the host one
        small copy and one graph launch."""
        if self._tensors is not None:
            return self.tensors(), fn
        fields = self._fields
        return [self._idx.clone()], (lambda idx: fn(*_gather_fields(fields, idx)))

    def fill(self, static):
        if self._tensors is not None:
            for dst, src in zip(static, self._tensors):
                dst.copy_(src, non_blocking=True)
        else:
            static[0].copy_(self._idx, non_blocking=True)


class _CapturedStep(object):
    """`fn(*static inputs)` captured into a hipGraph after ONE warm-up run on a side stream (a real step: the caller counts it as
    this batch's); `run(batch)` copies the batch into the static inputs and replays.  `signature` = what the capture baked in
    (shapes, optimizer / module identities, the addresses of the stacked memory fields)."""

    def __init__(self, fn, batch, signature):
        self.signature = signature
        self.static, fn = batch.capture_inputs(fn)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(*self.static)                     # a REAL step: code objects, workspaces, optimizer state; counted by the caller
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            fn(*self.static)                     # recorded, not executed

    def run(self, batch):
        batch.fill(self.static)
        self.graph.replay()


def _stack_field(x, device):
    return x.to(device) if torch.is_tensor(x) else x


# optimizer_str -> (class, keyword arguments of the value-side optimizer, of the predictor-side optimizer).  Upstream's choices
# (crowd_nav/utils/trainer.py:43-61,189-197): Adam with defaults for both; SGD with momentum 0.9 for the value side and plain SGD
# for the state predictor.  Adam is built `capturable` on a CUDA device so that its step can be recorded into the graph.
_OPTIMIZERS = {
    'Adam': (optim.Adam, {}, {}),
    'SGD': (optim.SGD, {'momentum': 0.9}, {}),
}


def _mark_parameters_changed(optimizer, args, kwargs):
    """Step post-hook of the fused Adam: torch's fused optimizer kernels update the parameters in place WITHOUT bumping autograd's
    version counters, and the descriptor caches (nets._PackCache: k-major weight copies, the search's weight images) key on those
    counters -- the forward after such a step, eager or being recorded into a captured step, would run on the packed weights of one
    step ago.  Host-side bookkeeping only; no kernel."""
    changed = [p for group in optimizer.param_groups for p in group['params'] if p.grad is not None]
    if changed:
        torch.autograd.graph.increment_version(changed)


def _new_optimizer(kind, module, learning_rate, capturable, predictor_side=False):
    """The optimizer upstream builds for `kind` (see _OPTIMIZERS).  Adam on a CUDA device is `capturable` -- its step can be recorded
    into the captured training step -- and `fused`: ONE multi-tensor kernel plus the step-counter increment per optimizer, where the
    default (foreach) capturable path is about fifty nodes of a few microseconds each, two of them PER PARAMETER (its divisions by
    the 0-dim bias corrections leave the multi-tensor fast path) -- 0.41 -> 0.24 ms per captured batch-100 step.  Same arithmetic up
    to rounding order (2.4e-7 on unit-scale parameters after three steps; tests hold both against the reference trainer's fixture).
    RGL_TRAINER_FUSED_ADAM=0 keeps the foreach path (A/B measurements)."""
    if kind not in _OPTIMIZERS:
        raise NotImplementedError
    cls, value_kw, predictor_kw = _OPTIMIZERS[kind]
    kw = dict(predictor_kw if predictor_side else value_kw)
    fused = False
    if cls is optim.Adam:
        kw['capturable'] = capturable
        fused = bool(capturable) and os.environ.get("RGL_TRAINER_FUSED_ADAM", "1") != "0"
        if fused:
            kw['fused'] = True
    optimizer = cls(module.parameters(), lr=learning_rate, **kw)
    if fused:
        optimizer.register_step_post_hook(_mark_parameters_changed)
    return optimizer


def _loss_signature(trainer):
    """What the loss end of a captured step bakes in: gamma_bar's ingredients (passed to rgl_mse_step_f32 by value when the step
    is recorded) and the criterion (its type and reduction select the fused loss or the torch one while recording)."""
    crit = trainer.criterion
    return (float(trainer.gamma), float(trainer.time_step) if trainer.time_step is not None else None,
            float(trainer.v_pref) if trainer.v_pref is not None else None, id(crit), type(crit), getattr(crit, "reduction", None))


def _log_learning_rate(learning_rate, modules, kind):
    """The line upstream logs from set_learning_rate (its format is part of the contract: logs are diffed against upstream runs)."""
    names = [name for m in modules for name, _ in m.named_parameters()]
    logging.info('Lr: {} for parameters {} with {} optimizer'.format(learning_rate, ' '.join(names), kind))


class _TrainerBase(object):
    """Graph bookkeeping shared by the two trainers."""

    capture = True                # class-level switch: False = every step eager (measurements, debugging)

    # what both upstream constructors set besides their own arguments (trainer.py:18-36,172-186): no target copy and no loader yet,
    # the MSE criterion on the device, and the constants of the value update target  r + gamma^(time_step * v_pref) V'
    _VALUE_UPDATE = (("gamma", 0.9), ("time_step", 0.25), ("v_pref", 1))

    def _adopt(self, device, memory, batch_size, optimizer_str, writer, **own):
        shared = dict(device=device, memory=memory, batch_size=batch_size, optimizer_str=optimizer_str, writer=writer,
                      target_model=None, data_loader=None, criterion=nn.MSELoss().to(device))
        for name, value in list(shared.items()) + list(own.items()) + list(self._VALUE_UPDATE):
            setattr(self, name, value)
        self._capturable = False
        self._init_graphs()

    @staticmethod
    def _require_learning_rate(optimizer):
        if optimizer is None:
            raise ValueError('Learning rate is not set!')

    def _init_graphs(self):
        self._steps = {}          # (kind, shapes...) -> _CapturedStep
        self._loss = None         # device float64 [2]: accumulated value / predictor loss of the current call
        self._stale = []          # modules whose packed weights a replay leaves behind the parameters
        self._replayed = False    # a replay ran since the caches were last invalidated

    def _can_capture(self, *optimizers):
        if not (self.capture and torch.cuda.is_available() and str(self.device).startswith("cuda")):
            return False
        return all(o is None or isinstance(o, optim.Adam) for o in optimizers)

    def _drop_graphs(self):
        self._steps = {}

    def _run(self, kind, fn, batch, modules, signature):
        """One optimisation step on `batch` (a _Batch): replay of the step captured for this kind / these shapes, or -- SGD,
        capture switched off -- the eager step."""
        key = (kind,) + batch.shapes()
        signature = signature + batch.identity()
        st = self._steps.get(key) if self._capturable else None
        if st is not None and st.signature != signature:
            st = None
        if st is None:
            # an EAGER run follows (the plain step, or the warm-up step of a new capture): replays moved the parameters without
            # bumping their version counters, so the descriptor caches would hand it the packed weights of one step ago
            if self._replayed:
                invalidate_packed_weights(*self._stale)
                self._replayed = False
            if not self._capturable:
                fn(*batch.tensors())
                return
            self._steps[key] = _CapturedStep(fn, batch, signature)            # its warm-up run WAS this batch's step
            for m in modules:
                if isinstance(m, nn.Module) and all(m is not x for x in self._stale):
                    self._stale.append(m)
            return
        st.run(batch)
        self._replayed = True

    def _run_eager(self, fn, tensors):
        """A step that is not captured (batches of mixed crowd sizes), after replays: the packed weights first."""
        if self._replayed:
            invalidate_packed_weights(*self._stale)
            self._replayed = False
        fn(*tensors)

    def _finish(self):
        """After the last replay of a call: replays do not bump autograd's version counters, so the descriptor caches of the
        trained modules (k-major weight copies, the search's weight images) must be told the parameters moved."""
        if self._stale and self._replayed:
            invalidate_packed_weights(*self._stale)
            self._replayed = False

    def _loss_begin(self):
        if self._loss is None:
            self._loss = torch.zeros(2, dtype=torch.float64, device=self.device)
            self._loss_ws = torch.zeros(nat.MSE_WORKSPACE_BYTES, dtype=torch.uint8, device=self.device)   # rgl_mse_step_f32's, zeroed once
        self._loss.zero_()

    def _loss_read(self):
        v, s = self._loss.tolist()               # the one host read of the call
        return v, s

    def _loss_backward(self, slot, outputs, target=None, bootstrap=None):
        """`loss = self.criterion(outputs, target); loss.backward()` and the loss added to the call's running sum
        (crowd_nav/utils/trainer.py:130-137,145-153,223-227).  `bootstrap` = (rewards, next values, gamma_bar) in place of `target`:
        the value update's `rewards + gamma_bar * target_model(next state)` (:128-129,221-222).
        On the device, with upstream's nn.MSELoss(): ONE launch (rgl_mse_step_f32: target, loss, its gradient -- torch's mse_backward
        arithmetic bit for bit -- and the running sum) and `outputs.backward(gradient)`, where torch runs nine launch-sized kernels.
        Any other criterion, dtype, broadcasting pair of shapes or device: torch, as upstream writes it."""
        others = [target] if target is not None else [bootstrap[0], bootstrap[1]]
        fused = (type(self.criterion) is nn.MSELoss and self.criterion.reduction == 'mean' and outputs.is_cuda and
                 outputs.dtype == torch.float32 and outputs.numel() > 0 and os.environ.get("RGL_TRAINER_FUSED_LOSS", "1") != "0" and
                 all(torch.is_tensor(t) and t.device == outputs.device and t.dtype == torch.float32 and t.shape == outputs.shape
                     and not t.requires_grad for t in others))
        if not fused:
            if target is None:
                target = bootstrap[0] + bootstrap[2] * bootstrap[1]
            loss = self.criterion(outputs, target)
            loss.backward()
            self._loss[slot].add_(loss.detach())            # float64 += float32 in one kernel (the iterator converts)
            return
        out = outputs.contiguous()
        grad = torch.empty_like(out)
        ptr = [t.contiguous() for t in others]
        with torch.cuda.device(out.device):
            rc = nat.lib().rgl_mse_step_f32(out.data_ptr(), ptr[0].data_ptr() if target is not None else None,
                                            None if target is not None else ptr[0].data_ptr(),
                                            None if target is not None else ptr[1].data_ptr(),
                                            0.0 if target is not None else float(bootstrap[2]), out.numel(), grad.data_ptr(),
                                            self._loss.data_ptr() + 8 * slot, self._loss_ws.data_ptr(), _stream())
        nat.check(rc, "rgl_mse_step_f32")
        out.backward(grad)


class MPRLTrainer(_TrainerBase):
    def __init__(self, value_estimator, state_predictor, memory, device, policy, writer, batch_size, optimizer_str, human_num,
                 reduce_sp_update_frequency, freeze_state_predictor, detach_state_predictor, share_graph_model):
        """Train the trainable models of a ModelPredictiveRL policy (crowd_nav/utils/trainer.py:10-36, same arguments; the attribute
        names are part of the contract -- train.py and user code read them)."""
        self._adopt(device, memory, batch_size, optimizer_str, writer,
                    value_estimator=value_estimator, state_predictor=state_predictor, target_policy=policy,
                    reduce_sp_update_frequency=reduce_sp_update_frequency, state_predictor_update_interval=human_num,
                    freeze_state_predictor=freeze_state_predictor, detach_state_predictor=detach_state_predictor,
                    share_graph_model=share_graph_model, v_optimizer=None, s_optimizer=None)

    # -- the reference's set-up calls ------------------------------------------------------------------------------------------
    def update_target_model(self, target_model):
        """A frozen copy of `target_model` (trainer.py:41).  The first call deep-copies; later calls copy the parameters INTO that
        copy, so captured steps -- which read the target's weights through fixed device pointers -- stay valid."""
        if self.target_model is not None and self._same_structure(self.target_model, target_model):
            with torch.no_grad():
                for dst, src in zip(self.target_model.parameters(), target_model.parameters()):
                    dst.copy_(src)
                for dst, src in zip(self.target_model.buffers(), target_model.buffers()):      # what deepcopy would snapshot as well
                    dst.copy_(src)
            if next(self.target_model.parameters()).is_cuda:
                invalidate_packed_weights(self.target_model)
                self._refresh_packed(self.target_model)
        else:
            self.target_model = copy.deepcopy(target_model)
            self._drop_graphs()

    @staticmethod
    def _same_structure(a, b):
        """Same module classes in the same order, same parameter / buffer shapes, dtypes and devices: only then may the in-place
        copy stand in for upstream's `copy.deepcopy` (anything else falls back to it and drops the captured steps)."""
        ma, mb = list(a.modules()), list(b.modules())
        if len(ma) != len(mb) or any(type(x) is not type(y) for x, y in zip(ma, mb)):
            return False
        for ta, tb in ((list(a.parameters()), list(b.parameters())), (list(a.buffers()), list(b.buffers()))):
            if len(ta) != len(tb) or any(x.shape != y.shape or x.dtype != y.dtype or x.device != y.device for x, y in zip(ta, tb)):
                return False
        return True

    @staticmethod
    def _refresh_packed(model):
        """Repack the descriptors of `model` now (in place: same device buffers), so that a captured step whose recording did not
        contain the repack reads the current weights."""
        with torch.no_grad():
            for m in model.modules():
                for name in ("descriptor", "head_descriptor"):
                    fn = getattr(m, name, None)
                    if callable(fn):
                        fn()

    def set_learning_rate(self, learning_rate):
        """Fresh optimizers at this rate (trainer.py:43-61); the state predictor gets one only while it is trainable."""
        capturable = self.capture and str(self.device).startswith("cuda")
        trained = [self.value_estimator] + ([self.state_predictor] if self.state_predictor.trainable else [])
        self.v_optimizer = _new_optimizer(self.optimizer_str, self.value_estimator, learning_rate, capturable)
        if self.state_predictor.trainable:
            self.s_optimizer = _new_optimizer(self.optimizer_str, self.state_predictor, learning_rate, capturable, predictor_side=True)
        self._capturable = self._can_capture(self.v_optimizer, self.s_optimizer)
        self._drop_graphs()                       # the optimizers (and their state tensors) are new
        _log_learning_rate(learning_rate, trained, self.optimizer_str)

    # -- batches ---------------------------------------------------------------------------------------------------------------
    def _batches(self):
        """Batches of (robot_states, human_states, values, rewards, next_robot_states, next_human_states), upstream's order."""
        fast = getattr(self.memory, "as_tensors", None)
        if self.data_loader is None and callable(fast):
            fields = fast()
            if fields is not None:
                # Gather from the FULL-CAPACITY stacked fields when the memory offers them: a captured step records its
                # index_select kernels with the source extent of the day of the capture, and the memory grows every episode --
                # replaying against `fields` (views of the first len(memory) rows at capture time) would index past the recorded
                # extent.  Indices are always < len(memory), so the rows beyond it are never read.
                whole = getattr(self.memory, "stacked_capacity_fields", None)
                if callable(whole):
                    fields = whole() or fields
                # the whole epoch's permutation goes to the device in ONE copy (a pageable host-to-device copy per batch would
                # stall the host behind the previous step every time); the batches are slices of it
                order = list(_ShuffledIndexBatches(len(self.memory), self.batch_size))
                if not order:
                    return
                perm = torch.cat(order).to(fields[0].device)
                lo = 0
                for idx in order:
                    yield _Batch(fields=fields, idx=perm[lo:lo + idx.shape[0]])
                    lo += idx.shape[0]
                return
        if self.data_loader is None:
            self.data_loader = DataLoader(self.memory, self.batch_size, shuffle=True)
        for data in self.data_loader:
            yield _Batch(tensors=[_stack_field(x, self.device) for x in data])

    # -- the two kinds of step ---------------------------------------------------------------------------------------------------
    def _signature(self):
        # everything a recorded step bakes in by value or by identity: upstream re-reads gamma / time_step / v_pref and the criterion
        # on every batch (crowd_nav/utils/trainer.py:128-131), so a change of any of them must force a new recording
        return (id(self.v_optimizer), id(self.s_optimizer), id(self.target_model), id(self.value_estimator), id(self.state_predictor),
                bool(self.detach_state_predictor)) + _loss_signature(self)

    def _value_step(self, robot_states, human_states, target=None, bootstrap=None):
        """One value update; `bootstrap()` -> (rewards, next values, gamma_bar) is evaluated after the forward, like upstream's
        target line."""
        self.v_optimizer.zero_grad()
        outputs = self.value_estimator((robot_states, human_states))
        self._loss_backward(0, outputs, target, bootstrap() if bootstrap is not None else None)
        self.v_optimizer.step()

    def _predictor_step(self, robot_states, human_states, next_human_states, detach):
        self.s_optimizer.zero_grad()
        if detach is None:
            _, next_human_states_est = self.state_predictor((robot_states, human_states), None)
        else:
            _, next_human_states_est = self.state_predictor((robot_states, human_states), None, detach=detach)
        self._loss_backward(1, next_human_states_est, next_human_states)
        self.s_optimizer.step()

    def optimize_epoch(self, num_epochs):
        self._require_learning_rate(self.v_optimizer)
        modules = [self.value_estimator, self.state_predictor]

        def il_step(update_sp):
            def fn(robot_states, human_states, values, next_human_states):
                self._value_step(robot_states, human_states, target=values)
                if update_sp:
                    self._predictor_step(robot_states, human_states, next_human_states, None)
            return fn
        steps = {False: il_step(False), True: il_step(True)}
        for epoch in range(num_epochs):
            self._loss_begin()
            logging.debug('{}-th epoch starts'.format(epoch))
            update_counter = 0
            for data in self._batches():
                update_sp = False
                if self.state_predictor.trainable:
                    update_sp = update_counter % self.state_predictor_update_interval == 0
                    update_counter += 1
                # robot_states, human_states, values, next_human_states of (robot, humans, value, reward, next robot, next humans)
                self._run(("il", update_sp), steps[update_sp], data.pick([0, 1, 2, 5]), modules, self._signature())
            epoch_v_loss, epoch_s_loss = self._loss_read()
            logging.debug('{}-th epoch ends'.format(epoch))
            self.writer.add_scalar('IL/epoch_v_loss', epoch_v_loss / len(self.memory), epoch)
            self.writer.add_scalar('IL/epoch_s_loss', epoch_s_loss / len(self.memory), epoch)
            logging.info('Average loss in epoch %d: %.2E, %.2E', epoch, epoch_v_loss / len(self.memory),
                         epoch_s_loss / len(self.memory))
        self._finish()
        return

    def optimize_batch(self, num_batches, episode):
        self._require_learning_rate(self.v_optimizer)
        gamma_bar = pow(self.gamma, self.time_step * self.v_pref)
        modules = [self.value_estimator, self.state_predictor]

        def rl_step(update_sp):
            def fn(robot_states, human_states, rewards, next_robot_states, next_human_states):
                def bootstrap():
                    with torch.no_grad():          # the frozen copy: upstream lets autograd walk it and never uses the result
                        return rewards, self.target_model((next_robot_states, next_human_states)), gamma_bar
                self._value_step(robot_states, human_states, bootstrap=bootstrap)
                if update_sp:
                    self._predictor_step(robot_states, human_states, next_human_states, self.detach_state_predictor)
            return fn
        steps = {False: rl_step(False), True: rl_step(True)}
        self._loss_begin()
        batch_count = 0
        for data in self._batches():
            update_sp = False
            if self.state_predictor.trainable:
                update_sp = True
                if self.freeze_state_predictor:
                    update_sp = False
                elif self.reduce_sp_update_frequency and batch_count % self.state_predictor_update_interval == 0:
                    update_sp = False
            # robot_states, human_states, rewards, next_robot_states, next_human_states
            self._run(("rl", update_sp), steps[update_sp], data.pick([0, 1, 3, 4, 5]), modules, self._signature())
            batch_count += 1
            if batch_count > num_batches:
                break
        v_losses, s_losses = self._loss_read()
        self._finish()
        average_v_loss = v_losses / num_batches
        average_s_loss = s_losses / num_batches
        logging.info('Average loss : %.2E, %.2E', average_v_loss, average_s_loss)
        self.writer.add_scalar('RL/average_v_loss', average_v_loss, episode)
        self.writer.add_scalar('RL/average_s_loss', average_s_loss, episode)
        return average_v_loss, average_s_loss


def _pad_longest_first(sequences):
    """Variable-length (L_i, D) sequences -> ((n, L_max, D) zero-padded tensor, int64 lengths), LONGEST FIRST; equally long ones
    keep their order (a stable sort on the negated length)."""
    order = sorted(range(len(sequences)), key=lambda i: -sequences[i].shape[0])
    picked = [sequences[i] for i in order]
    lengths = torch.tensor([t.shape[0] for t in picked], dtype=torch.int64)
    return torch.nn.utils.rnn.pad_sequence(picked, batch_first=True), lengths


def pad_batch(batch):
    """Collate function of the path-G trainer (the role of crowd_nav/utils/trainer.py:253-272): items (state (L, D), value (1,),
    reward (1,), next state (L', D)) -> ((states, lengths), values (n, 1), rewards (n, 1), (next states, lengths)).
    As upstream: each of the two state lists is ordered longest first ON ITS OWN while values and rewards keep the batch order --
    only consistent when every crowd of a batch has one size, which is the only case the graph network accepts anyway."""
    columns = list(zip(*batch))
    values = torch.cat(columns[1]).unsqueeze(1)
    rewards = torch.cat(columns[2]).unsqueeze(1)
    return _pad_longest_first(columns[0]), values, rewards, _pad_longest_first(columns[3])


class VNRLTrainer(_TrainerBase):
    def __init__(self, model, memory, device, policy, batch_size, optimizer_str, writer):
        """Train the value network of a path-G policy (crowd_nav/utils/trainer.py:164-186, same arguments and attribute names)."""
        self._adopt(device, memory, batch_size, optimizer_str, writer, model=model, policy=policy, optimizer=None)

    update_target_model = MPRLTrainer.update_target_model
    _same_structure = staticmethod(MPRLTrainer._same_structure)
    _refresh_packed = staticmethod(MPRLTrainer._refresh_packed)

    def set_learning_rate(self, learning_rate):
        """A fresh optimizer at this rate (trainer.py:189-197)."""
        self.optimizer = _new_optimizer(self.optimizer_str, self.model, learning_rate,
                                        self.capture and str(self.device).startswith("cuda"))
        self._capturable = self._can_capture(self.optimizer)
        self._drop_graphs()
        _log_learning_rate(learning_rate, [self.model], self.optimizer_str)

    def _batches(self):
        """Batches of ((states, lengths), values, rewards, (next_states, lengths)) -- pad_batch's output."""
        if self.data_loader is None:
            self.data_loader = DataLoader(self.memory, self.batch_size, shuffle=True, collate_fn=pad_batch)
        for data in self.data_loader:
            yield data

    def _signature(self):
        return (id(self.optimizer), id(self.target_model), id(self.model)) + _loss_signature(self)

    def _step(self, inputs, lengths, target=None, bootstrap=None):
        self.optimizer.zero_grad()
        outputs = self.model((inputs, lengths))
        self._loss_backward(0, outputs, target, bootstrap() if bootstrap is not None else None)
        self.optimizer.step()

    def _full_lengths(self, lengths, width):
        """The graph network takes whole batches of equally long sequences (gcn.ValueNetwork.forward's `lengths` only names the
        crowd size); mixed lengths have no static shape to capture and run eagerly."""
        return bool((torch.as_tensor(lengths) == width).all())

    def optimize_epoch(self, num_epochs):
        self._require_learning_rate(self.optimizer)
        average_epoch_loss = 0
        for epoch in range(num_epochs):
            self._loss_begin()
            logging.debug('{}-th epoch starts'.format(epoch))
            for data in self._batches():
                (inputs, lengths), values, _, _ = data
                inputs, values = inputs.to(self.device), values.to(self.device)
                lengths = torch.as_tensor(lengths)

                def fn(x, v, lengths=lengths):
                    self._step(x, lengths, target=v)
                if self._full_lengths(lengths, inputs.shape[1]):
                    self._run(("il",), fn, _Batch(tensors=[inputs, values]), [self.model], self._signature())
                else:
                    self._run_eager(fn, [inputs, values])
            epoch_loss = self._loss_read()[0]
            logging.debug('{}-th epoch ends'.format(epoch))
            average_epoch_loss = epoch_loss / len(self.memory)
            self.writer.add_scalar('IL/average_epoch_loss', average_epoch_loss, epoch)
            logging.info('Average loss in epoch %d: %.2E', epoch, average_epoch_loss)
        self._finish()
        return average_epoch_loss

    def optimize_batch(self, num_batches, episode=None):
        self._require_learning_rate(self.optimizer)
        gamma_bar = pow(self.gamma, self.time_step * self.v_pref)
        self._loss_begin()
        batch_count = 0
        for data in self._batches():
            (inputs, lengths), _, rewards, (next_states, next_lengths) = data
            inputs, rewards, next_states = inputs.to(self.device), rewards.to(self.device), next_states.to(self.device)
            lengths, next_lengths = torch.as_tensor(lengths), torch.as_tensor(next_lengths)

            def fn(x, r, x2, lengths=lengths, next_lengths=next_lengths):
                def bootstrap():
                    with torch.no_grad():
                        return r, self.target_model((x2, next_lengths)), gamma_bar
                self._step(x, lengths, bootstrap=bootstrap)
            if self._full_lengths(lengths, inputs.shape[1]) and self._full_lengths(next_lengths, next_states.shape[1]):
                self._run(("rl",), fn, _Batch(tensors=[inputs, rewards, next_states]), [self.model], self._signature())
            else:
                self._run_eager(fn, [inputs, rewards, next_states])
            batch_count += 1
            if batch_count > num_batches:
                break
        losses = self._loss_read()[0]
        self._finish()
        average_loss = losses / num_batches
        logging.info('Average loss : %.2E', average_loss)
        return average_loss


def register_trainers(namespace):
    """Install the graph-replaying trainers under the names crowd_nav/train.py imports (`from crowd_nav.utils.trainer import
    VNRLTrainer, MPRLTrainer`): pass the module (before train.py is imported) or a dict."""
    target = namespace if isinstance(namespace, dict) else namespace.__dict__
    target['MPRLTrainer'] = MPRLTrainer
    target['VNRLTrainer'] = VNRLTrainer
    target['pad_batch'] = pad_batch
    return namespace
