#!/usr/bin/env python
"""Launch the dominant kernels (mprl_value_children_f32: children_graph_kernel + robot_head_kernel) a few
times on rollout-shaped inputs -- the target of `rocprofv3 --pmc ...` counter passes.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA ... -- python tools/profile_children.py
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parents", type=int, default=4096)
    ap.add_argument("--humans", type=int, default=19)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--contraction", choices=("f32", "f16", "bf16x6"), default="bf16x6",
                    help="default: the mode bench.py's `value` runs in for this shape (--contraction auto)")
    ap.add_argument("--tree", action="store_true", help="run whole tree searches instead of the kernel pair")
    args = ap.parse_args()
    args.depth, args.width = 2, 2
    dev = torch.device("cuda:0")
    pol = bench.make_policy(args, dev)
    ts = pol.tree_search()
    robot, humans = bench.synth_scenes(5, args.parents, args.humans)
    robot, humans = robot.to(dev), humans.to(dev)
    if args.tree:
        for _ in range(args.reps):
            ts.search(robot[:2048], humans[:2048], roots_are_joint_states=False)
    else:
        ex = ts.expand(robot, humans, parents_are_joint_states=False)
        for _ in range(args.reps):
            ts.value_children(ex["child_robot"], ex["humans_next"])
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
