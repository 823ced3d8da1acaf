#!/usr/bin/env python
"""Value of the children for the similarity functions / graph structures that have no shared-crowd form, on the one-wave-per-scene
MFMA kernel and (RGL_FORCE_GENERIC=1, separate process) on the general VALU kernel.

    python tools/sim_modes_timing.py [parents]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tests.helpers import make_mprl_policy  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    print("general VALU kernel forced" if os.environ.get("RGL_FORCE_GENERIC") == "1" else "default dispatch", " P =", P, " H = 19, L = 2")
    for sim, lw in (("embedded_gaussian", False), ("embedded_gaussian", True), ("cosine", False), ("cosine_softmax", False),
                    ("concatenation", False), ("concatenation", True)):
        pol = make_mprl_policy("trained", 1, similarity=sim, layerwise=lw, device=dev)
        pol.build_action_space(1.0)
        ts = pol.tree_search()
        robot, humans = bench.synth_scenes(5, P, 19)
        ex = ts.expand(robot.to(dev), humans.to(dev), parents_are_joint_states=False)
        for _ in range(3):
            ts.value_children(ex["child_robot"], ex["humans_next"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            ts.value_children(ex["child_robot"], ex["humans_next"])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("  %-18s layerwise=%d : %8.3f ms per call   %.3e children/s" % (sim, lw, ms, P * ts.num_actions / ms * 1e3))


if __name__ == "__main__":
    main()
