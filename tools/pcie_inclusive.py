#!/usr/bin/env python
"""Step time of the headline workload when the root states start in (pinned) host memory and the decisions are read back:
H2D of robot (B,9) + humans (B,H,5), one search, D2H of (action, value).  `bench.py`'s `value` is measured with the inputs
already resident in HBM; this is the PCIe-inclusive figure quoted in DESIGN.md."""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class A:
    layers, depth, width, humans, contraction = 2, 2, 2, 19, "f32"


def main():
    B = 2048
    dev = torch.device("cuda:0")
    pol = bench.make_policy(A, dev)
    ts = pol.tree_search()
    robot_h, humans_h = bench.synth_scenes(1000, B, A.humans)
    robot_h, humans_h = robot_h.pin_memory(), humans_h.pin_memory()
    robot_d, humans_d = torch.empty_like(robot_h, device=dev), torch.empty_like(humans_h, device=dev)
    act_h = torch.empty(B, dtype=torch.int32).pin_memory()
    val_h = torch.empty(B, dtype=torch.float32).pin_memory()

    def step(copy):
        if copy:
            robot_d.copy_(robot_h, non_blocking=True)
            humans_d.copy_(humans_h, non_blocking=True)
        out = ts.search(robot_d, humans_d, roots_are_joint_states=False, want_root_values=False)
        if copy:
            act_h.copy_(out["best_action"], non_blocking=True)
            val_h.copy_(out["best_value"], non_blocking=True)
    gc.collect()
    gc.freeze()
    for copy in (False, True):
        for _ in range(50):
            step(copy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            step(copy)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 300 * 1e3
        print("%-28s %.4f ms/step  %.3e evals/s" % ("inputs resident in HBM:" if not copy else "host -> HBM -> host (PCIe):", ms,
                                                     B * 249 / ms * 1e3))


if __name__ == "__main__":
    main()
