#!/usr/bin/env python
"""bench.py -- agent-graph forward evals/sec of the RGL model-predictive rollout on MI355X.

One "step" = one complete depth-D, width-w action-tree search (state predictor, 81-action
expansion, rewards, value estimator, top-w clipping, V_planning back-up, argmax) for a batch of
synthetic root scenes, inputs resident in HBM.  Workload = BASELINE.json configs[2]: N=20 agents
(19 humans + robot), 2-layer GCN, depth-2 rollout (width 2), 2048 root scenes.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus 8 --steps 50 --warmup 10          # launches its own 8 ranks (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 50 --warmup 10        # what the driver does; same line

Prints ONE JSON line on rank 0.  `value` counts reference-equivalent ValueEstimator forwards
(249 per root for D=2,w=2) completed per second over all ranks.  Multi-GPU:
roots are sharded and each step ends with the one real exchange of the path, an RCCL all-gather
of the per-shard (action, value) rows.  Scaling readings (`--scaling`):

    both   (default)  N = 1: the workload as is (reported as "weak": nothing is split).  N > 1: `value` is the
                      FIXED-TOTAL reading SURVEY.md §8(d) prescribes (--roots roots in total, split contiguously over
                      the ranks: "scaling": "strong"), and the same invocation also times --roots roots PER GPU and
                      reports it as `weak_value` / `weak_ms_per_step`.
    weak              --roots R roots PER GPU: per-GPU work fixed as N grows
    strong            --total-roots T roots in total, split contiguously over the ranks
                      (BASELINE configs[3]: --scaling strong --total-roots 4096 --depth 3;
                       configs[4]: --scaling strong --total-roots 2048 --humans 49 --layers 3 --contraction f16)

With more than one rank the line also carries `multi_gpu`: every rank's search time per step and the exchange time per
step, measured separately (un-pipelined) after the timed region, and `ranks_seen` = dist.get_world_size().

`--graph auto|on|off`: replay the rank's whole search from a captured hipGraph (auto: depth-1 searches of <= 512 roots per rank,
the one case where it measured faster); `step_ms_device` (HIP events between steps) against `ms_per_step` (wall clock) shows
whether the host is the limiter.

RGL_BENCH_STUB_SEARCH=1 replaces the device search by a trivial CPU function over gloo: a test switch for the launcher
and the exchange logic on machines without GPUs (tests/test_bench_launcher.py).  The line it prints says so and is not a
measurement of anything.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EXCHANGE_DIAG = os.environ.get("RGL_BENCH_EXCHANGE_DIAG", "")   # measurements only: "nowait" = the steps never order themselves after the gathers
INIT_MS = float(os.environ.get("RGL_BENCH_INIT_MS", "80"))  # least duration of the set-up phase (device clock ramp), see main()
STUB = os.environ.get("RGL_BENCH_STUB_SEARCH") == "1"       # launcher / exchange test switch: no device, no kernels, no measurement

import relationalgraphlearning_amd as rga  # noqa: E402
from relationalgraphlearning_amd.config import policy_config  # noqa: E402

FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: vector == f32-MFMA peak
F16_MFMA_PEAK_TFLOPS = 2500.0   # dense f16/bf16 MFMA peak


def children_flops_per_scene(N, L, A):
    """EXECUTED FLOPs per child scene of the value-of-children kernels (stage 1 + stage 2), from the terms of
    SURVEY.md §8(d) (2 per MAC, softmax ~5 per element), counting only what the algorithm needs (no MFMA
    padding).  Crowd-only work is done once per parent and amortised over its A children; the last GCN layer is
    evaluated for the robot node only.  For L == 2 and N <= 32 the rank-1 kernel runs: human
    rows cost ~7 FLOP per (row, feature) instead of a dense A*X and *W product."""
    H = N - 1
    head = 28648
    last_layer = 2 * 32 * 32 + 2 * 32                                   # t*W_last, relu/skip (stage 2)
    robot_embed = 5248
    if L == 2 and N <= 32:
        per_parent = H * 4736 + 2 * H * 32 * 32 + 2 * H * H * 32 + 5 * H * H + 2 * H * H * 32 + 2 * H * 32 * 32
        per_child = (robot_embed + 2 * 2 * 32 * 32                      # x0*Wa, x0*W1
                     + 4 * N * 32 + 5 * N + 12 * H                      # robot row/column of S, robot-row softmax, row scalars
                     + 7 * H * 32 + 2 * 32                              # rank-1 row pass, T_0
                     + 2 * 32 * 32 + 4 * 32)                            # T_0*W1, H1_0, t_c
        return (per_parent / A + per_child + last_layer + head,
                "rank-1 form: children_fused_kernel (one launch: crowd quantities, 16-child tiles from embedding to value) from ~3k "
                "child tiles per launch, children_rank1_kernel + robot_head_kernel below")
    if L in (2, 3) and N <= 60:
        # shared-crowd deep kernel (rgl_deep.hip): layer 0 in rank-1 form; for L == 3 one dense layer per child whose
        # aggregation is E*O with the crowd-only E shared by the siblings
        per_parent = H * 4736 + 2 * H * 32 * 32 + 2 * H * H * 32 + 5 * H * H + 2 * H * H * 32 + 2 * H * 32 * 32
        per_child = (robot_embed + 2 * 2 * 32 * 32 + 4 * N * 32 + 5 * N + 12 * H        # x0*Wa, x0*W1, S row/col, p, a/b
                     + 5 * H * 32                                                       # layer-0 rows
                     + 2 * N * 32 + 2 * 32 * 32 + 2 * 32)                               # T_0, T_0*W1, H1_0
        if L == 3:
            per_child += 2 * N * 32 + 2 * N * 32 * 32 + 2 * N * N * 32 + 6 * N * 32 + 2 * 32    # p*H1, H1*W2, E*O, H2 + t_c
        else:
            per_child += 2 * N * 32                                                     # t_c = p*H1
        return per_parent / A + per_child + last_layer + head, "shared-crowd deep (children_deep_kernel, stage 2 -- the value head -- as its trailing phase; robot_head_kernel with RGL_DEEP_FUSE_HEAD=0)"
    per_parent = H * 4736 + 2 * H * 32 * 32 + 2 * H * H * 32
    sim = 2 * 32 * 32 + 4 * N * 32 + 5 * N * N
    full_layer = 2 * N * N * 32 + 2 * N * 32 * 32 + 2 * N * 32
    return (per_parent / A + robot_embed + sim + (L - 1) * full_layer + 2 * N * 32 + last_layer + head,
            "tiles (children_graph_kernel + robot_head_kernel)")


def predictor_flops_per_scene(N, L):
    """Algorithmic FLOPs of one state-predictor graph forward (state_predictor.py:20-39: embeddings, similarity, L GCN layers on
    every node, motion head on the human rows; 2 per MAC, softmax ~5 per element) -- one per tree node."""
    H = N - 1
    return (5248 + H * 4736                                        # w_r on the robot row, w_h on the human rows
            + 2 * N * 32 * 32 + 2 * N * N * 32 + 5 * N * N         # X Wa, (X Wa) X^T, softmax
            + L * (2 * N * N * 32 + 2 * N * 32 * 32)               # A H, (A H) W
            + H * (2 * 32 * 64 + 2 * 64 * 5))                      # motion head 32 -> 64 -> 5


def workload_name(N, args, reading=None):
    key = (N, args.layers, args.depth, args.width)
    if key == (20, 2, 2, 2):
        return "BASELINE configs[2]"
    share = "" if (reading or getattr(args, "scaling", "weak")) == "strong" else "per-GPU share, "
    if key == (20, 2, 3, 2):
        return "BASELINE configs[3]" + (" (per-GPU share)" if share else "")
    if key == (50, 3, 2, 2):
        return "BASELINE configs[4] (%s%s contractions)" % (share, args.contraction)
    if args.depth == 1 and N in (5, 6):
        return "BASELINE configs[1]"
    return "custom"


def load_weights(flavour, L, device):
    """Fixture F1 'trained-like' weights (tests/golden/weights_*.npz: data, generated by the reference's
    own constructors) -> a configured ModelPredictiveRL on `device`."""
    m = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_%s.npz" % flavour)))

    def graph(which):
        pre = which + "."
        sd = {k[len(pre):]: torch.tensor(v) for k, v in m.items() if k.startswith(pre) and ".Ws." not in k}
        for l in range(L):
            sd["Ws.%d" % l] = torch.tensor(m[pre + "Ws.%d" % l])
        return sd

    def sub(which):
        pre = which + "."
        return {k[len(pre):]: torch.tensor(v) for k, v in m.items() if k.startswith(pre)}
    return {"graph_model1": graph("graph_model1"), "graph_model2": graph("graph_model2"),
            "value_network": sub("value_network"), "motion_predictor": sub("motion_predictor")}


CLEARANCE = 0.3 + 0.3 + 0.2        # radius + radius + discomfort distance (crowd_sim.py:131-139 re-draws below it)


def synth_scenes(seed, B, H, placement="clearance"):
    """Seeded synthetic crowd states: robot on the radius-4 circle heading to the antipode, humans in
    [-5,5]^2 with velocities in [-1,1]^2, radius 0.3.  placement="clearance" (default since round 3) places the humans
    one after the other and re-draws a position until it keeps 0.3 + 0.3 + 0.2 from the robot and from every human placed
    before it -- the rejection rule of the reference's scene generator (crowd_sim.py:131-139), vectorised over the B scenes;
    "uniform" is the round-1/2 generator (no re-draw; kept for one round so that r02 numbers stay reproducible)."""
    rng = np.random.RandomState(seed)
    robot = np.zeros((B, 9), np.float32)
    ang = rng.uniform(0, 2 * np.pi, B)
    robot[:, 0], robot[:, 1] = 4 * np.cos(ang), 4 * np.sin(ang)
    v = rng.uniform(-1, 1, (B, 2))
    v /= np.maximum(1.0, np.linalg.norm(v, axis=1, keepdims=True))
    robot[:, 2:4] = v
    robot[:, 4] = 0.3
    robot[:, 5], robot[:, 6] = -4 * np.cos(ang), -4 * np.sin(ang)
    robot[:, 7] = 1.0
    robot[:, 8] = np.pi / 2
    humans = np.zeros((B, H, 5), np.float32)
    if placement == "uniform":
        humans[:, :, 0:2] = rng.uniform(-5, 5, (B, H, 2))
    elif placement == "clearance":
        pos = np.zeros((B, H, 2))
        placed = np.concatenate([robot[:, None, 0:2].astype(np.float64), pos], axis=1)      # [robot, humans placed so far]
        for i in range(H):
            todo = np.arange(B)
            while todo.size:
                cand = rng.uniform(-5, 5, (todo.size, 2))
                d = np.linalg.norm(placed[todo, :i + 1] - cand[:, None, :], axis=2)
                ok = (d >= CLEARANCE).all(axis=1)
                placed[todo[ok], i + 1] = cand[ok]
                todo = todo[~ok]
        humans[:, :, 0:2] = placed[:, 1:]
    else:
        raise ValueError("placement must be 'clearance' or 'uniform'")
    humans[:, :, 2:4] = rng.uniform(-1, 1, (B, H, 2))
    humans[:, :, 4] = 0.3
    return torch.tensor(robot), torch.tensor(humans)


def make_policy(args, device):
    cfg = policy_config("model_predictive_rl", gcn__num_layer=args.layers,
                        model_predictive_rl__planning_depth=args.depth,
                        model_predictive_rl__planning_width=args.width,
                        model_predictive_rl__do_action_clip=(args.depth > 1))
    pol = rga.ModelPredictiveRL()
    pol.time_step = 0.25
    pol.configure(cfg)
    pol.load_state_dict(load_weights("trained", args.layers, device))
    pol.set_time_step(0.25)
    pol.set_phase("test")
    pol.set_device(device)
    pol.build_action_space(1.0)
    pol.contraction_dtype = getattr(args, "contraction", "f32")
    return pol


def cpu_baseline(args, robot, humans, budget_s, device_values=None):
    """The CPU oracle timed on this host: (i) the reference-order batch-1 walk, (ii) the batched restatement.
    `device_values`: {mode: best_value tensor of the first roots} -- their deviation from a FLOAT64 evaluation of the same search
    (the oracle on double tensors, 64 roots) is reported beside the timings (`float64_check`): the checker at work, not timed."""
    from oracle import rgl_oracle as orc          # test infrastructure: only this leg may touch it
    from tests import golden_io as gio
    P = gio.oracle_params("trained", args.layers)
    cfg = orc.OracleConfig(num_layer=args.layers, planning_depth=args.depth, planning_width=args.width,
                           do_action_clip=(args.depth > 1))
    threads = torch.get_num_threads()
    per_root = None
    torch.set_num_threads(1)                       # batch-1 forwards: intra-op threading only adds dispatch overhead
    n_seq, t0 = 0, time.time()
    with torch.no_grad():
        while True:
            tr = orc.SeqTrace()
            orc.mprl_predict_sequential([float(x) for x in robot[n_seq]],
                                        [[float(x) for x in row] for row in humans[n_seq]], P, cfg, tr)
            per_root = tr.n_value_forwards
            n_seq += 1
            if time.time() - t0 > budget_s * 0.5 or n_seq >= robot.shape[0]:
                break
    t_seq = time.time() - t0
    torch.set_num_threads(threads)
    nb = min(robot.shape[0], 256)
    with torch.no_grad():
        orc.mprl_predict_batched(robot[:8], humans[:8], P, cfg)      # warm-up
        t0 = time.time()
        reps = 0
        while True:
            orc.mprl_predict_batched(robot[:nb], humans[:nb], P, cfg)
            reps += 1
            if time.time() - t0 > budget_s * 0.5:
                break
    t_bat = time.time() - t0
    f64 = None
    if device_values:
        n64 = min(64, robot.shape[0])
        ck = gio.checkpoint("trained", args.layers)
        P64 = orc.MprlParams.from_checkpoint({k: {kk: vv.double() for kk, vv in v.items()} for k, v in ck.items()})
        with torch.no_grad():
            _, v64, _, _ = orc.mprl_predict_batched(robot[:n64].double(), humans[:n64].double(), P64, cfg)
            _, v32, _, _ = orc.mprl_predict_batched(robot[:n64], humans[:n64], P, cfg)
        f64 = {"roots": n64, "max_abs_dV_vs_float64": {k: float((v[:n64].double().cpu() - v64).abs().max())
                                                       for k, v in device_values.items()},
               "rms_dV_vs_float64": {k: float((v[:n64].double().cpu() - v64).pow(2).mean().sqrt()) for k, v in device_values.items()}}
        f64["max_abs_dV_vs_float64"]["f32 oracle (torch CPU)"] = float((v32.double() - v64).abs().max())
    return {"value": n_seq * per_root / t_seq, "unit": "evals/s", "cores": 1, "kind": "port", "float64_check": f64,
            "sample": "%d roots, reference-order batch-1 walk (%d value forwards/root), 1 thread, %.1f s"
                      % (n_seq, per_root, t_seq),
            "batched_value": reps * nb * per_root / t_bat, "batched_cores": threads,
            "batched_sample": "%d x %d roots, level-synchronous torch-CPU restatement, %d threads, %.1f s"
                              % (reps, nb, threads, t_bat),
            "host_cpus": os.cpu_count()}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--roots", type=int, default=2048,
                    help="root scenes of the workload: per GPU in the weak reading, in total in the fixed-total reading of "
                         "--scaling both")
    ap.add_argument("--scaling", choices=("both", "weak", "strong"), default="both")
    ap.add_argument("--total-roots", type=int, default=None,
                    help="strong scaling: root scenes in total, split contiguously over the ranks (default: --roots)")
    ap.add_argument("--humans", type=int, default=19)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--width", type=int, default=2)
    ap.add_argument("--contraction", choices=("auto", "f32", "f16", "bf16x6"), default="auto",
                    help="auto (default): bf16x6 where the value-of-children kernel offers it (2-layer graphs, N <= 32), f32 otherwise, "
                         "with the plain f32-MFMA line printed beside it at N = 1; "
                         "bf16x6: f32-WIDTH operands (three bf16 pieces each, six MFMA terms) for the first 64 input features of the "
                         "children kernel's 100 x 100 head matrix -- admitted to `value`: DESIGN.md 4; "
                         "f16: f16-input MFMA for the dense middle-layer products (BASELINE configs[4]; needs --layers 3)")
    ap.add_argument("--scenes", choices=("clearance", "uniform"), default="clearance",
                    help="human placement of the synthetic scenes: SURVEY 8(d)'s clearance re-draw (default) or the round-1/2 "
                         "uniform draw")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="replay the rank's search from a captured hipGraph (auto: depth-1 searches of <= 512 roots per rank)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline leg (0 = skip)")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed.run environment: start the N ranks ourselves (one process
    per GPU, rendezvous on 127.0.0.1) and pass their output through -- rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # the host driver only supports dmabuf IPC (RCCL needs it)
    return subprocess.call(cmd, env=env)


class _StubSearch:
    """RGL_BENCH_STUB_SEARCH=1: a stand-in with the TreeSearch surface bench.py touches.  No arithmetic of the path -- it
    exists so that the launcher, the sharding and the exchange can be run end to end on gloo without a GPU."""
    num_actions, kept_per_node = 81, 2

    def __init__(self, depth):
        self.depth = depth

    def logical_value_evals_per_root(self):
        return {1: 81, 2: 249, 3: 581}.get(self.depth, 81)

    def search(self, robot, humans, roots_are_joint_states=False, want_root_values=False, out=None):
        act = (robot[:, 0].abs() * 1000).to(torch.int32) % self.num_actions
        val = robot.sum(dim=1) + humans.sum(dim=(1, 2))
        if out is not None:
            out[0].copy_(act)
            out[1].copy_(val)
            return {"best_action": out[0], "best_value": out[1]}
        return {"best_action": act, "best_value": val}


class Leg:
    """One timed reading: this rank's share of `total_roots` root scenes, its search (direct C-ABI call or replay of a
    captured hipGraph) and the pipelined exchange."""

    def __init__(self, args, ts, device, world, rank, total_roots, split, dist):
        self.args, self.ts, self.device, self.world, self.rank, self.dist = args, ts, device, world, rank, dist
        H = args.humans
        if split:                                         # every rank slices the SAME seeded global batch
            lo, hi = rga.shard_bounds(total_roots, world, rank)
            robot_all, humans_all = synth_scenes(1000, total_roots, H, args.scenes)
            self.robot_cpu, self.humans_cpu = robot_all[lo:hi].contiguous(), humans_all[lo:hi].contiguous()
            self.roots_per_rank = [rga.shard_bounds(total_roots, world, r)[1] - rga.shard_bounds(total_roots, world, r)[0]
                                   for r in range(world)]
        else:
            per = total_roots // world
            self.robot_cpu, self.humans_cpu = synth_scenes(1000 + rank, per, H, args.scenes)
            self.roots_per_rank = [per] * world
        self.B, self.total_roots = self.robot_cpu.shape[0], total_roots
        self.robot, self.humans = self.robot_cpu.to(device), self.humans_cpu.to(device)
        # measured (profiles/r03_a_share_regime.jsonl): a replayed search is 3-5 % SLOWER than the direct C-ABI call at depth 2 / 3
        # (0.116 vs 0.111 ms at 256 roots -- the host is not the limiter: wall = device time in both forms) and 7 % faster only
        # for the depth-1 search (0.0645 vs 0.0693 ms at 512 roots): auto = depth 1 with <= 512 roots per rank
        self.use_graph = (not STUB) and (args.graph == "on" or (args.graph == "auto" and args.depth == 1
                                                                and 0 < max(self.roots_per_rank) <= 512))
        self.sharded = rga.ShardedRollout(self._search_fn, search_into=self._search_into)
        self.graphs, self.graph_last = {}, None
        if self.use_graph and self.B > 0:                 # capture BEFORE the first collective of this leg is enqueued
            if self.sharded.active:
                targets = self.sharded.use_static_buffers(total_roots, device)
            else:
                targets = [(torch.empty(self.B, dtype=torch.int32, device=device),
                            torch.empty(self.B, dtype=torch.float32, device=device))]
            for a, v in targets:
                g, out = ts.capture(self.robot, self.humans, roots_are_joint_states=False, want_root_values=False, out=(a, v),
                                    private_workspace=True)
                self.graphs[a.data_ptr()] = (g, out)
            self.single = targets[0]
            self.graph_last = self.graphs[targets[0][0].data_ptr()][1]["last"]
        self.pending = None

    # the two shapes ShardedRollout calls the rank's search in
    def _search_fn(self, r, h):
        if self.graphs:
            self.graphs[self.single[0].data_ptr()][0].replay()
            return self.single
        o = self.ts.search(r, h, roots_are_joint_states=False, want_root_values=False)
        return o["best_action"], o["best_value"]

    def _search_into(self, r, h, a, v):
        if self.graphs:
            self.graphs[a.data_ptr()][0].replay()
        else:
            self.ts.search(r, h, roots_are_joint_states=False, want_root_values=False, out=(a, v))

    # One exchange per step, pipelined by one step: the all-gather of step i is waited for after the search of step
    # i+1 has been enqueued, so it runs on RCCL's stream underneath that search.  drain() completes the last one;
    # every timed step's exchange finishes inside the timed region.
    def step(self):
        nxt = self.sharded.launch_local(self.robot, self.humans, self.total_roots)
        if self.pending is not None and EXCHANGE_DIAG != "nowait":
            self.pending.wait()              # exchange of the previous step complete (stream-ordered); .result() unpacks
        self.pending = nxt

    def drain(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None

    def fence(self):
        if not STUB:
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if not STUB:
            torch.cuda.synchronize()

    def timed(self, steps, warmup, init_steps):
        """`init_steps` set-up steps (code objects, workspace, RCCL channels), `warmup` untimed steps, then exactly `steps`
        timed steps between fences; returns (wall seconds MAX over ranks, sorted per-step device ms of this rank)."""
        # one event in front of each of the first (up to) 10 set-up steps and one behind the last of them: every event in the list
        # is recorded whatever `init_steps` is (with 4..10 set-up steps the last one never was, and elapsed_time raised)
        n_cold = min(init_steps, 10)
        cold = None if STUB or n_cold == 0 else [torch.cuda.Event(enable_timing=True) for _ in range(n_cold + 1)]
        t_init = time.perf_counter()
        for i in range(init_steps):
            if cold and i <= n_cold:
                cold[i].record()
            self.step()
        if cold and init_steps == n_cold:
            cold[n_cold].record()
        self.drain()
        self.fence()
        # ... and the set-up phase lasts INIT_MS of device work at least (the clock ramp is a matter of time, not of steps: with
        # the step at 0.26 ms, 40 of them end before the device holds its clock -- profiles/r06_init_sweep.txt)
        self.init_steps_run = init_steps
        if init_steps > 0:
            spent = (time.perf_counter() - t_init) * 1e3
            more = min(20000, int(max(0.0, INIT_MS - spent) / max(spent / init_steps, 1e-3)))
            if self.dist is not None:          # every rank runs the same number of steps (each one is an exchange)
                t = torch.tensor([more], dtype=torch.int64, device=self.device)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                more = int(t.item())
            for _ in range(more):
                self.step()
            self.drain()
            self.fence()
            self.init_steps_run += more
        # steps 2.. of the set-up phase (the very first ones also pay for code objects and workspaces): a cold device's step time
        self.cold_ms = (sorted(cold[i].elapsed_time(cold[i + 1]) for i in range(2, n_cold))
                        if cold and n_cold > 3 else None)
        for _ in range(warmup):
            self.step()
        self.drain()
        self.fence()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step()
        self.drain()
        self.fence()
        elapsed = time.perf_counter() - t0
        # per-step device times: a SECOND pass of the same `steps` steps with a HIP event between them, after the timed region (an
        # event record between two launches is a marker packet the next kernel waits for: measurement, not work -- round 4 took the
        # 21 of them out of the timed steps)
        marks = None if STUB else [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        if marks:
            for i in range(steps):
                marks[i].record()
                self.step()
            self.drain()
            marks[steps].record()
            self.fence()
        step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps)) if marks else [elapsed / steps * 1e3] * steps
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, step_ms

    def decisions_digest(self):
        """One more (untimed) step, gathered on every rank: sha256 over the int32 action indices of ALL roots in root order, and
        over the fp32 values.  In the fixed-total reading every world size searches the same seeded roots, so the action digest of
        an N-rank run can be held against the single-process run (tests/test_bench_launcher.py does, over gloo; on hardware a
        differing digest can only come from numerical ties, since the launch shapes -- and with them the summation order -- change
        with the shard size)."""
        import hashlib
        self.fence()
        act, val = self.sharded.launch_local(self.robot, self.humans, self.total_roots).result()
        self.fence()
        a = act.detach().cpu().numpy().astype(np.int32)
        v = val.detach().cpu().numpy().astype(np.float32)
        return {"roots": int(a.shape[0]), "actions_sha256": hashlib.sha256(a.tobytes()).hexdigest(),
                "values_sha256": hashlib.sha256(v.tobytes()).hexdigest()}

    def diagnose(self, n_diag=10):
        """Search and exchange timed SEPARATELY (no pipelining) after the timed region, every rank's medians gathered."""
        dist, dev = self.dist, self.device
        if STUB:
            srch = exch = 0.0
            for _ in range(2):
                self.fence()
                t0 = time.perf_counter()
                h = self.sharded.launch_local(self.robot, self.humans, self.total_roots)
                t1 = time.perf_counter()
                h.result()
                srch, exch = (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3
        else:
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_diag)]
            for i in range(n_diag):
                self.fence()
                ev[i][0].record()
                h = self.sharded.launch_local(self.robot, self.humans, self.total_roots)   # search enqueued, all-gather behind it
                ev[i][1].record()                                     # end of this rank's search on the compute stream
                h.wait()                                              # compute stream waits for the collective
                ev[i][2].record()
            self.fence()
            srch = sorted(ev[i][0].elapsed_time(ev[i][1]) for i in range(n_diag))[n_diag // 2]
            exch = sorted(ev[i][1].elapsed_time(ev[i][2]) for i in range(n_diag))[n_diag // 2]
        mine = torch.tensor([srch, exch], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(every, mine)
        per_rank = [[float(x[0]), float(x[1])] for x in every]
        return {"ranks_seen": dist.get_world_size(),
                "search_ms_per_step_by_rank": [p[0] for p in per_rank], "exchange_ms_per_step_by_rank": [p[1] for p in per_rank],
                "search_ms_per_step_slowest_rank": max(p[0] for p in per_rank),
                "search_ms_per_step_fastest_rank": min(p[0] for p in per_rank),
                "exchange_ms_per_step_slowest_rank": max(p[1] for p in per_rank),
                "note": "median of %d un-pipelined steps after the timed region; exchange = from the end of the rank's own "
                        "search to the completion of the all-gather on its stream (includes waiting for slower ranks)" % n_diag,
                "roots_per_rank": self.roots_per_rank}


def children_roofline(args, ts, device, N, H, last, robot=None, humans=None):
    """Roofline of the dominant kernels (value of the sibling children), HIP events on the launch stream.
    `achieved` / `frac` price the kernel AS IT RUNS IN THE SEARCH (round 4): mprl_tree_search_traced_f32 records events around
    every level's value-of-children launch -- with the selection, and at the deepest level the back-up chain and the root
    decision, in its tail -- and the mean launch duration over the levels of a step divides the executed FLOPs of a launch.
    `standalone_*`: the same kernels launched straight through mprl_value_children_f32 (no tail work) on the inputs the last
    search left in its workspace, one launch per tree level, in the same mix -- the figure rounds 1-3 reported."""
    import ctypes as C
    from relationalgraphlearning_amd import _native as nat
    if last is not None:
        ts.last = last
    A = ts.num_actions
    levels = [ts.level_arrays(l) for l in range(args.depth)]
    outs = [torch.empty(lv["n_parents"], A, device=device) for lv in levels]
    lib = nat.lib()
    pl_desc = ts.planner(device)
    p_max = max(lv["n_parents"] for lv in levels)
    hand_off = torch.empty(lib.mprl_value_children_workspace_bytes(C.byref(pl_desc), p_max, H), dtype=torch.uint8,
                           device=device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    calls = [(lv["child_robot"].data_ptr(), lv["humans_next"].data_ptr(), lv["n_parents"], o.data_ptr())
             for lv, o in zip(levels, outs)]

    def launch_children_once():
        for cr, hn, n_par, op in calls:
            rc = lib.mprl_value_children_f32(C.byref(pl_desc), cr, hn, n_par, H, op, hand_off.data_ptr(),
                                             hand_off.numel(), stream)
            assert rc == 0, rc
    for _ in range(3):
        launch_children_once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(5, min(args.steps, 20))
    e0.record()
    for _ in range(reps):
        launch_children_once()
    e1.record()
    torch.cuda.synchronize()
    kern_launches = reps * len(calls)
    kern_ms = e0.elapsed_time(e1) / kern_launches
    scenes_per_launch = sum(lv["n_parents"] for lv in levels) * A / len(levels)
    flop_per_scene, kernel_path = children_flops_per_scene(N, args.layers, A)
    standalone_ms = kern_ms
    in_search = None
    if robot is not None:
        n_tr = 10
        acc = np.zeros(args.depth)
        for i in range(n_tr + 2):
            ts.search(robot, humans, roots_are_joint_states=False, want_root_values=False, trace=True)
            if i >= 2:
                acc += np.asarray(ts.last["trace"]["children_ms"])
        in_search = (acc / n_tr).tolist()
        kern_ms = float(sum(in_search) / len(in_search))
    achieved = scenes_per_launch * flop_per_scene / (kern_ms * 1e-3) / 1e12
    peak, peak_note = FP32_PEAK_TFLOPS, "fp32 vector == f32-MFMA peak (the two do not co-execute on gfx950)"
    if args.contraction == "bf16x6":
        # input features 0..95 of the 100 x 100 head matrix, all 32 of the 32 x 100 one (onto their 96 full-tile outputs), the 32 x 32
        # layer before them and (round 6) the tile chain's weight products -- hidden -> x0 (64 x 32), x0 Wa, x0 W1, T_0 W1, W_last --
        # plus the crowd's Xh Wa / U W1 (H rows per parent, shared by its A children) run on the bf16 matrix pipe as SIX terms, i.e. at a sixth of
        # the dense bf16 MFMA peak; everything else at the fp32 rate: time-weighted peak
        dense = (2 * 96 * 96 + 2 * 32 * 96 + 2 * 32 * 32) + (2 * 64 * 32 + 4 * 2 * 32 * 32) + 2.0 * (N - 1) * 2 * 32 * 32 / A
        b6_peak = F16_MFMA_PEAK_TFLOPS / 6.0
        peak = flop_per_scene / (dense / b6_peak + (flop_per_scene - dense) / FP32_PEAK_TFLOPS)
        peak_note = "blend: %.0f%% of the FLOPs as 6 bf16 MFMA terms over three-piece (24-bit) operands (a sixth of the dense bf16 MFMA peak = %.0f), the rest at the fp32 peak (%.1f)" % (
            100.0 * dense / flop_per_scene, b6_peak, FP32_PEAK_TFLOPS)
    if args.contraction == "f16":
        # the two dense middle-layer products run on the f16 matrix pipe, the rest at the fp32 rate: time-weighted peak
        dense = 2 * N * 32 * 32 + 2 * N * N * 32
        peak = flop_per_scene / (dense / F16_MFMA_PEAK_TFLOPS + (flop_per_scene - dense) / FP32_PEAK_TFLOPS)
        peak_note = "blend: %.0f%% of the FLOPs at the dense f16 MFMA peak (%.0f), the rest at the fp32 peak (%.1f)" % (
            100.0 * dense / flop_per_scene, F16_MFMA_PEAK_TFLOPS, FP32_PEAK_TFLOPS)
    # HBM bytes per launch from the PMC passes committed under profiles/ (counters cannot be read from inside this process);
    # only quoted for the workload they were measured on, and stamped with the source revision they were measured at -- a
    # record from another revision of the kernels is stale by construction, and the line says so (`traffic_stale`)
    traffic, traffic_src, traffic_rev, stale = None, None, None, None
    try:
        import glob
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
        tj = json.load(open(latest))
        if (tj["workload"]["N"], tj["workload"]["L"], tj["workload"]["A"]) == (N, args.layers, A):
            traffic, traffic_src = tj["bytes_per_scene"] * scenes_per_launch, tj["source"]
            traffic_rev = tj.get("source_revision", "round 1 (unstamped)")
            now = kernel_sources_digest()
            stale = None if tj.get("kernel_sources_sha256") is None else (tj["kernel_sources_sha256"] != now)
    except (OSError, KeyError, ValueError, IndexError):
        pass
    return {"bound": "mfma", "kernel": "value of sibling children, mprl_value_children_f32: " + kernel_path,
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "peak_note": peak_note,
            "frac": achieved / peak, "frac_of_fp32_peak": achieved / FP32_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
            "traffic_source": traffic_src, "traffic_measured_at_revision": traffic_rev,
            "traffic_stale": stale,
            "algorithmic_bytes": scenes_per_launch * (9 * 4 + 4) + scenes_per_launch / A * H * 20,
            "launch_ms": kern_ms, "launch_ms_source": ("HIP events around the level's children launch inside mprl_tree_search_traced_f32 "
                                                      "(tail work included), mean over the levels of a step" if in_search else
                                                      "stand-alone mprl_value_children_f32 launches"),
            "in_search_children_ms_by_level": in_search,
            "standalone_launch_ms": standalone_ms,
            "standalone_frac": scenes_per_launch * flop_per_scene / (standalone_ms * 1e-3) / 1e12 / peak,
            "scenes_per_launch": scenes_per_launch,
            "flop_per_scene": flop_per_scene,
            "reference_flop_per_eval": {(20, 2): 328120, (50, 3): 1337660, (6, 2): 102300}.get((N, args.layers))}


def kernel_sources_digest():
    """sha256 over the HIP sources and headers the library is built from (sorted by name): what `profiles/r*_traffic.json`
    is stamped with, so that a counter record from another state of the kernels is recognisable without git."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "relationalgraphlearning_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))
                    + [os.path.join(ROOT, "include", "rgl_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def resolve_contraction(args):
    """`auto` -> the mode that carries `value` for this workload.  bf16x6 (RGL_CONTRACT_BF16X6) keeps every operand at f32's 24
    significand bits and drops <= 2^-23 |w||a| per product in the worst case (~2^-25 typically) -- it is the reference's arithmetic width on another pipe, admitted by the
    criteria of DESIGN.md 4 (suite under the mode, float64 deviation not above the f32 kernels') -- and exists where the fused
    children kernel runs: two GCN layers, N <= 32.  Deterministic in the workload (never in a timing), so that the lines of
    N = 1, 2, 4, 8 ranks are the same arithmetic."""
    args.contraction_requested = args.contraction
    if args.contraction == "auto":
        args.contraction = "bf16x6" if (args.layers == 2 and args.humans + 1 <= 32 and not STUB) else "f32"
    return args


def main():
    args = resolve_contraction(parse_args())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    if args.gpus != world:
        sys.exit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    dist = None
    if STUB:
        device = torch.device("cpu")
        backend = "gloo"
    else:
        assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
        if os.environ.get("RGL_BENCH_SINGLE_DEVICE") == "1":      # world-size-1 RCCL smoke on a one-GPU box
            local_rank = 0
        if local_rank >= torch.cuda.device_count():
            sys.exit("rank %d: --gpus %d needs %d GPUs on this node, found %d" % (rank, args.gpus, args.gpus,
                                                                                  torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        backend = "nccl"
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: one process per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if STUB:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)

    H, N = args.humans, args.humans + 1
    if STUB:
        ts = _StubSearch(args.depth)
    else:
        pol = make_policy(args, device)
        ts = pol.tree_search()

    # which readings: (label, total roots, split one global batch?)
    if args.scaling == "strong":
        main_leg = ("strong", args.total_roots if args.total_roots is not None else args.roots, True)
        extra_leg = None
    elif args.scaling == "weak" or world == 1:
        main_leg = ("weak", args.roots * world, False)
        extra_leg = None
    else:                                             # both, N > 1: fixed total is `value`, per-GPU fixed rides along
        main_leg = ("strong", args.roots, True)
        extra_leg = ("weak", args.roots * world, False)

    # Host hygiene: a generation-2 pass of CPython's cyclic GC over the heap torch leaves behind takes ~40 ms (measured:
    # it landed inside the timed region and starved the GPU for 60 steps' worth of time).  Freeze the set-up objects so
    # later collections only look at what the steps themselves allocate.
    import gc
    # Set-up, not warm-up: the first calls load the code objects, size the workspace and (N > 1) open the RCCL channels -- and
    # the device needs ~15 ms of work to reach the clock it then holds: measured on one box with `--steps 20 --warmup 5`
    # (profiles/r04_init_steps.txt), 3 set-up steps: 0.398 ms per timed step, 40 or 150: 0.375-0.377.  The line reports what
    # the first steps cost (`step_ms_device_cold`: HIP events around set-up steps 2..9) beside the steady figure.
    # Round 6: the ramp is a matter of TIME -- with the step at 0.26 ms, 40 set-up steps (10 ms) left the driver's 20 timed steps on a
    # device still below its clock: 0.268-0.276 ms against 0.260 with 40 ms or more of set-up, three repeats each on one box
    # (profiles/r06_init_sweep.txt) -- so the set-up phase now runs INIT_STEPS steps and at least RGL_BENCH_INIT_MS of them.
    INIT_STEPS = int(os.environ.get("RGL_BENCH_INIT_STEPS", "40"))
    leg = Leg(args, ts, device, world, rank, main_leg[1], main_leg[2], dist)
    gc.collect()
    gc.freeze()
    elapsed, step_ms = leg.timed(args.steps, args.warmup, INIT_STEPS)
    per_root = ts.logical_value_evals_per_root()
    total_roots, B = leg.total_roots, leg.B
    value = per_root * total_roots * args.steps / elapsed
    multi = leg.diagnose() if dist is not None else None
    digest = leg.decisions_digest()

    weak = None
    if extra_leg is not None:
        leg2 = Leg(args, ts, device, world, rank, extra_leg[1], extra_leg[2], dist)
        e2, s2 = leg2.timed(args.steps, args.warmup, INIT_STEPS)
        weak = {"weak_value": per_root * leg2.total_roots * args.steps / e2, "weak_ms_per_step": e2 / args.steps * 1e3,
                "weak_total_roots": leg2.total_roots, "weak_roots_per_gpu": leg2.B,
                "weak_step_ms_device_median": s2[len(s2) // 2], "weak_graph_replay": bool(leg2.graphs),
                "weak_multi_gpu": leg2.diagnose()}
        del leg2

    A, W = ts.num_actions, ts.kept_per_node
    roofline = roofline_step = None
    if not STUB and B > 0:
        if not leg.graphs:                                   # leave the main leg's levels in the shared workspace
            ts.search(leg.robot, leg.humans, roots_are_joint_states=False, want_root_values=False)
        roofline = children_roofline(args, ts, device, N, H, leg.graph_last, leg.robot, leg.humans)
        # the whole step against the same peak: executed (algorithmic) FLOPs of everything a step computes -- the children's
        # value forwards and the state predictor's graph forwards, one per tree node -- over the measured step time
        n_nodes = sum(W_ ** l for W_ in [ts.kept_per_node] for l in range(args.depth)) * B
        step_flops = n_nodes * (ts.num_actions * roofline["flop_per_scene"] + predictor_flops_per_scene(N, args.layers))
        step_tflops = step_flops / (elapsed / args.steps) / 1e12
        roofline_step = {"flops_per_step": step_flops, "achieved": step_tflops, "peak": roofline["peak"], "unit": "TFLOP/s",
                         "frac": step_tflops / roofline["peak"], "frac_of_fp32_peak": step_tflops / FP32_PEAK_TFLOPS,
                         "note": "children value forwards (%d per tree node) + state-predictor graph forwards (one per tree node, "
                                 "%.0f FLOP), %d tree nodes on this GPU, over ms_per_step" % (ts.num_actions,
                                                                                         predictor_flops_per_scene(N, args.layers), n_nodes)}

    # ---- the plain f32-MFMA line beside a bf16x6 `value` ("the f32-MFMA line still printed beside it")
    f32_line, ts32 = None, None
    if not STUB and world == 1 and args.contraction == "bf16x6" and B > 0 and os.environ.get("RGL_BENCH_NO_F32_LINE") != "1":
        import copy
        a32 = copy.copy(args)
        a32.contraction = "f32"
        pol32 = make_policy(a32, device)
        ts32 = pol32.tree_search()
        leg32 = Leg(a32, ts32, device, world, rank, main_leg[1], main_leg[2], None)
        e32, s32 = leg32.timed(args.steps, args.warmup, INIT_STEPS)
        o_main = ts.search(leg.robot, leg.humans, roots_are_joint_states=False, want_root_values=False)
        o_main = {k: v.clone() for k, v in o_main.items() if torch.is_tensor(v)}
        o32 = ts32.search(leg.robot, leg.humans, roots_are_joint_states=False, want_root_values=False)
        r32 = children_roofline(a32, ts32, device, N, H, None, leg.robot, leg.humans)
        f32_line = {"value": per_root * total_roots * args.steps / e32, "ms_per_step": e32 / args.steps * 1e3,
                    "step_ms_device_median": s32[len(s32) // 2], "dtype": "f32",
                    "max_abs_dV_vs_value_kernels": float((o_main["best_value"] - o32["best_value"]).abs().max()),
                    "identical_decisions": float((o_main["best_action"] == o32["best_action"]).float().mean()),
                    "roofline": {k: r32[k] for k in ("achieved", "peak", "frac", "peak_note", "launch_ms", "standalone_launch_ms", "unit")},
                    "note": "contraction_dtype f32: the same search with every product on v_mfma_f32_16x16x4_f32 / the VALU (what "
                            "carried `value` until round 4), timed the same way in the same process right after the main leg"}
        del leg32
        ts.search(leg.robot, leg.humans, roots_are_joint_states=False, want_root_values=False)

    metric = "agent-graph forward evals/sec (N=%d, %d-layer GCN, depth-%d tree)" % (N, args.layers, args.depth)
    result = {
        "metric": ("STUB SEARCH, NOT A MEASUREMENT: " if STUB else "") + metric,
        "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": main_leg[0],
        "step_ms_device": {"p10": step_ms[len(step_ms) // 10], "median": step_ms[len(step_ms) // 2],
                           "p90": step_ms[(len(step_ms) * 9) // 10], "note": "rank 0, HIP events between steps, a second pass of the same steps after the timed region"},
        "step_ms_device_cold": (None if not getattr(leg, "cold_ms", None) else
                                {"median": leg.cold_ms[len(leg.cold_ms) // 2], "max": leg.cold_ms[-1],
                                 "note": "set-up steps 2..9 of this process, before the device reached its steady clock (not timed into `value`)"}),
        "vs_baseline": None, "dtype": {"f32": "f32", "f16": "f16 inputs / f32 accumulate (middle-layer products only; everything else f32)",
                                       "bf16x6": "f32 (24-bit operands throughout; 13 312 of the 14 224 products of the children kernel's 32 x 32, 32 x 100 and "
                                                 "100 x 100 head matrices, its embedding / graph weight products (w_r's second layer, Wa, W1, W_last) "
                                                 "and the state predictor's weight products as six bf16 MFMA terms over three "
                                                 "round-to-nearest bf16 pieces per operand, f32 accumulate, dropped terms <= 2^-23 |w||a| (worst case); "
                                                 "everything else on the f32 MFMA / VALU)"}[args.contraction],
        "data": "synthetic",
        "config": {"workload": "%s: N=%d agents (H=%d humans), %d-layer GCN, depth-%d width-%d "
                               "action-tree rollout, %s" % (workload_name(N, args, main_leg[0]), N, H, args.layers, args.depth, args.width,
                                                            ("%d root scenes per GPU" % B) if main_leg[0] == "weak" else
                                                            ("%d root scenes in total over %d GPU(s)" % (total_roots, world))),
                   "roots_per_gpu": B, "total_roots": total_roots, "logical_value_evals_per_root": per_root, "init_steps": getattr(leg, "init_steps_run", INIT_STEPS), "init_ms": INIT_MS,
                   # what changed in HOW the line is measured, so that round-over-round deltas can be read: 1 = rounds
                   # 1-3 (3 set-up steps, event marks inside the timed loop); 2 = round 4 on (40 untimed set-up steps in front of
                   # --warmup so the device holds its steady clock; per-step event marks in a second pass after the timed region)
                   "methodology_revision": 2,
                   "executed_graph_forwards_per_root": sum(W ** l for l in range(args.depth)) * (A + 1),
                   "decisions_per_s": total_roots * args.steps / elapsed,
                   "weights": "fixture F1 trained-like (tests/golden/weights_trained.npz)",
                   "scenes": "human placement: " + args.scenes,
                   "graph_replay": bool(leg.graphs),
                   "exchange": "all_gather_into_tensor of (roots_per_gpu,2) fp32 per rank" if world > 1 else "none"},
        "roofline": roofline,
        "roofline_step": roofline_step,
        "decisions": digest,
    }
    if f32_line is not None:
        result["f32_mfma_line"] = f32_line
    if args.contraction == "bf16x6":
        result["admission"] = {
            "mode": "RGL_CONTRACT_BF16X6", "operand_bits": 24, "pieces_per_operand": "3 x bf16, round to nearest, hi + mid + lo = x exactly",
            "terms": "6 of 9 (lo*hi, mid*mid, hi*lo, mid*hi, hi*mid, hi*hi), f32 accumulate",
            "dropped_terms_bound": "w_mid a_lo + w_lo a_mid + w_lo a_lo <= (2^-24 + 2^-24 + 2^-32) |w||a| < 2^-22.99 |w||a| per product in the worst case (|mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|), ~2^-25 typically; unbiased",
            "where": "children_fused_kernel: the 32 x 32, 32 x 100 and 100 x 100 value-head matrices, input features 0..95 onto output "
                     "features 0..95, and (round 6) w_r's 64 x 32 layer, Wa, W1 and W_last in the tile chain and the crowd computation (what the CU's LDS "
                     "holds as three bf16 pieces with w_h's second matrix in registers and the 4 x 4 x 1 fragments stored compact); scene_graph_kernel: "
                     "Wa, W_l, motion head; everything else f32",
            "float64_check": "cpu_baseline.float64_check of this line (needs --cpu-seconds > 0); "
                             "tests/test_gpu_parity.py::test_bf16x6_head_matrix_at_size_and_in_other_shapes asserts it",
            "suite_under_the_mode": "profiles/r06_final_suite_under_bf16x6.txt (RGL_CONTRACT_F32_AS=bf16x6: every f32 search of the GPU suite in this mode)"}
    result["config"]["contraction"] = args.contraction
    result["config"]["contraction_requested"] = args.contraction_requested
    if STUB:
        result["stub_search"] = True
    if weak is not None:
        result.update(weak)
    if multi is not None:
        result["multi_gpu"] = multi
        result["ranks_seen"] = multi["ranks_seen"]
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and not STUB:
        dv = {("%s kernels" % args.contraction): ts.search(leg.robot[:64], leg.humans[:64], roots_are_joint_states=False,
                                                           want_root_values=False)["best_value"].clone()}
        if ts32 is not None:
            dv["f32 kernels"] = ts32.search(leg.robot[:64], leg.humans[:64], roots_are_joint_states=False,
                                            want_root_values=False)["best_value"].clone()
        result["cpu_baseline"] = cpu_baseline(args, leg.robot_cpu, leg.humans_cpu, args.cpu_seconds, dv)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
