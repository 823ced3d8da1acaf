#!/usr/bin/env python
"""Kernel-iteration harness: times the value-of-children pair (mprl_value_children_f32) at several parent counts and
whole searches of the BASELINE shapes, and checks the pair against the general kernel on a small odd-sized batch.

    [RGL_HIP_LIBRARY=.../librgl_hip_x.so] python tools/kiter.py [--humans 19] [--layers 2] [--quick]
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relationalgraphlearning_amd import _native as nat  # noqa: E402


def time_calls(fn, reps):
    """(best, median) milliseconds per call over five batches of `reps` calls.  The MEDIAN, with CPython's cyclic GC held off:
    round 3's table carried `pair P=2048: best 0.1052 ms mean 1.3292 ms` -- one of three batches had caught a full collection
    over torch's heap (tens of ms during which no kernel is issued; bench.py freezes the heap for the same reason), and a mean
    of three hands that to the reader as if it were kernel time."""
    import gc
    for _ in range(max(10, reps)):
        fn()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    try:
        ms = []
        for _ in range(9 if reps >= 50 else 5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / reps)
    finally:
        gc.enable()
    ms.sort()
    return ms[0], ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--humans", type=int, default=19)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--contraction", default="f32")
    ap.add_argument("--parents", type=int, nargs="*", default=[256, 512, 1024, 2048, 4096])
    ap.add_argument("--quick", action="store_true", help="skip the whole-search timings")
    ap.add_argument("--reps", type=int, default=20, help="calls per timed batch (>= 50: nine batches and as many warm-up calls)")
    args = ap.parse_args()
    args.depth, args.width = 2, 2
    dev = torch.device("cuda:0")
    pol = bench.make_policy(args, dev)
    ts = pol.tree_search()
    A, H = ts.num_actions, args.humans
    print("library:", nat.LIB_PATH)

    # ---- correctness of the pair vs the general kernel (module forward), odd parent count
    Pc = 37
    robot, humans = bench.synth_scenes(7, Pc, H)
    ex = ts.expand(robot.to(dev), humans.to(dev), parents_are_joint_states=False)
    got = ts.value_children(ex["child_robot"], ex["humans_next"])
    with torch.no_grad():
        want = pol.value_estimator((ex["child_robot"].reshape(Pc * A, 1, 9),
                                    ex["humans_next"][:, None].expand(Pc, A, H, 5).reshape(Pc * A, H, 5).contiguous()))
    err = float((got.reshape(-1) - want.reshape(-1)).abs().max())
    print("check P=%d: max |pair - general kernel| = %.3e   (max |v| = %.3f)" % (Pc, err, float(want.abs().max())))

    # ---- pair timing
    flop, path = bench.children_flops_per_scene(H + 1, args.layers, A)
    print("path:", path, " flop/scene %.0f" % flop)
    lib = nat.lib()
    for P in args.parents:
        robot, humans = bench.synth_scenes(5, P, H)
        ex = ts.expand(robot.to(dev), humans.to(dev), parents_are_joint_states=False)
        out = torch.empty(P, A, device=dev)
        pl = ts.planner(dev)
        ws = torch.empty(lib.mprl_value_children_workspace_bytes(C.byref(pl), P, H), dtype=torch.uint8, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        cr, hn = ex["child_robot"], ex["humans_next"]

        def call():
            rc = lib.mprl_value_children_f32(C.byref(pl), cr.data_ptr(), hn.data_ptr(), P, H, out.data_ptr(), ws.data_ptr(),
                                             ws.numel(), stream)
            assert rc == 0, rc
        best, mean = time_calls(call, args.reps)
        print("pair P=%5d: best %.4f ms  median %.4f ms   %.1f TFLOP/s  frac %.3f" % (
            P, best, mean, P * A * flop / (mean * 1e-3) / 1e12, P * A * flop / (mean * 1e-3) / 1e12 / bench.FP32_PEAK_TFLOPS))
    if args.quick:
        return
    # ---- whole searches
    for (Hh, D, B) in [(19, 2, 2048), (19, 3, 512), (4, 1, 512), (5, 1, 512)]:
        if args.layers != 2:
            break
        a2 = argparse.Namespace(layers=2, depth=D, width=2, contraction="f32")
        p2 = bench.make_policy(a2, dev)
        t2 = p2.tree_search()
        robot, humans = bench.synth_scenes(11, B, Hh)
        robot, humans = robot.to(dev), humans.to(dev)
        best, mean = time_calls(lambda: t2.search(robot, humans, roots_are_joint_states=False, want_root_values=False), 20)
        ev = t2.logical_value_evals_per_root() * B
        print("search H=%2d D=%d B=%4d: best %.4f ms  median %.4f ms   %.3e evals/s" % (Hh, D, B, best, mean, ev / (mean * 1e-3)))


if __name__ == "__main__":
    main()
