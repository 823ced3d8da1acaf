"""Autograd for the HIP forward: `GraphFunction` runs rgl_graph_forward_f32 in forward and
rgl_graph_backward_f32 in backward, so crowd_nav's trainers (MPRLTrainer / VNRLTrainer,
crowd_nav/utils/trainer.py:110-161,199-250) can optimise the modules of `nets.py` with torch optimizers.
Gradients flow to parameters only (the states of a replay batch are data)."""
import ctypes as C

import torch

from . import _native as nat


def linear_params(seq):
    out = []
    for m in seq:
        if isinstance(m, torch.nn.Linear):
            out += [m.weight, m.bias]
    return out


class GraphFunction(torch.autograd.Function):
    """forward(spec, robot, humans, *params) -> tuple of requested outputs.

    `spec` (plain python object, not a tensor) provides:
       graph() / value_head() / motion_head()  -> ctypes descriptors (or None), built from the CURRENT parameters
       want_H, detach_graph                    -> flags
       param_shapes                            -> [(kind, shape)] in slab order, kind in {'linear_w','vector','matrix'}
    `params` are the parameter tensors in slab order; they are inputs only so that autograd tracks them."""

    @staticmethod
    def forward(ctx, spec, robot, humans, *params):
        from .nets import graph_forward, batched_transposes
        if robot.requires_grad or humans.requires_grad:
            raise NotImplementedError("gradients with respect to the agent states are not provided by the HIP path "
                                      "(the states of a replay batch are data): detach the inputs")
        with batched_transposes():                 # the descriptors of one forward: one transpose launch for all their weights
            graph, vh, mh = spec.graph(), spec.value_head(), spec.motion_head()
        out = graph_forward(graph, vh, mh, robot, humans, want_H=spec.want_H, want_A=spec.want_A)
        ctx.spec = spec
        ctx.save_for_backward(robot, humans)
        ctx.n_params = len(params)
        ctx.param_versions = [(p.data_ptr(), p._version) for p in params]
        ctx.keys = [k for k in ("H", "value", "humans_next", "A") if k in out]
        for k in ("A",):
            if k in out:
                ctx.mark_non_differentiable(out[k])
        return tuple(out[k] for k in ctx.keys)

    @staticmethod
    def backward(ctx, *grads):
        spec = ctx.spec
        robot, humans = ctx.saved_tensors
        # the backward kernel recomputes the forward from the CURRENT parameters: refuse if they changed since forward
        # (torch raises for the same situation on its own saved tensors)
        if [(p.data_ptr(), p._version) for p in spec.params] != ctx.param_versions:
            raise RuntimeError("a parameter of the graph model was modified in place between forward and backward "
                               "(e.g. an optimizer step before loss.backward()): gradients would belong to other weights")
        g = dict(zip(ctx.keys, grads))
        S, H = robot.shape[0], humans.shape[1]
        from .nets import batched_transposes
        with batched_transposes():
            graph, vh, mh = spec.graph(), spec.value_head(), spec.motion_head()
        lib = nat.lib()
        vhp = C.byref(vh) if vh is not None else None
        mhp = C.byref(mh) if mh is not None else None
        n = lib.rgl_graph_param_count(C.byref(graph), vhp, mhp)
        if n <= 0:
            nat.check(n, "rgl_graph_param_count")
        dev = robot.device

        def ptr(name):
            t = g.get(name)
            if t is None:
                return None, None
            t = t.contiguous().to(torch.float32)
            return t, t.data_ptr()
        dv_t, dv = ptr("value")
        dm_t, dm = ptr("humans_next")
        dh_t, dh = ptr("H")
        flat = torch.empty(n, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.rgl_graph_backward_workspace_bytes(C.byref(graph), vhp, mhp, S), dtype=torch.uint8, device=dev)
        if nat.poison_workspaces():              # tests: a kernel that reads what nobody wrote turns the gradients into NaN
            flat.fill_(float("nan"))
            ws.fill_(255)
        with torch.cuda.device(dev):
            rc = lib.rgl_graph_backward_f32(C.byref(graph), vhp, mhp, robot.data_ptr(), humans.data_ptr(), S, H,
                                            int(spec.detach_graph), dv, dm, dh, flat.data_ptr(), ws.data_ptr(), ws.numel(),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream))
        nat.check(rc, "rgl_graph_backward_f32")
        grads_out, off = [], 0
        for kind, shape in spec.param_shapes:
            cnt = 1
            for d in shape:
                cnt *= d
            piece = flat[off:off + cnt]
            off += cnt
            # contiguous views into the freshly allocated slab (no copy kernels, and the optimizers' fused kernels take contiguous
            # gradients: transposed views sent them down torch's strided slow path); Linear weights come in torch layout (out, in)
            grads_out.append(piece.view(*shape))
        assert off == n, (off, n)
        if spec.detach_graph:
            # StatePredictor(..., detach=True) cuts the graph model out of the autograd graph upstream (state_predictor.py:29-30): its
            # parameters receive NO gradient -- `.grad` stays None after zero_grad() and the optimizer skips them.  Zeros are not the
            # same thing: Adam keeps moving a parameter along its momentum (and counts a step) on a zero gradient, which is what the
            # graph model would do through a whole RL phase after imitation learning has trained it (configs/icra_benchmark/mp_detach.py).
            for i in range(spec.n_graph_params):
                grads_out[i] = None
        return (None, None, None) + tuple(grads_out)
