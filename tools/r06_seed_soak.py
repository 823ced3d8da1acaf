#!/usr/bin/env python
"""RGL_CONTRACT_BF16X6 against the f32 kernels over many seeded scene sets: decisions (action indices) and best values of whole
searches, BASELINE configs[2] (2048 roots), configs[3]'s share (512 roots, depth 3) and configs[1] (N = 6, depth 1).

    gpurun -- 'python tools/r06_seed_soak.py [seeds]' -> gpurun_out/r06_bf16x6_seed_soak.txt
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def searches(humans, layers, depth, width, dev):
    out = []
    for mode in ("f32", "bf16x6"):
        a = types.SimpleNamespace(layers=layers, depth=depth, width=width, humans=humans, contraction=mode)
        out.append(bench.make_policy(a, dev).tree_search())
    return out


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    dev = torch.device("cuda:0")
    lines = []
    for name, humans, depth, width, roots in (("configs[2]: N = 20, depth 2, 2048 roots", 19, 2, 2, 2048),
                                              ("configs[3] share: N = 20, depth 3, 512 roots", 19, 3, 2, 512),
                                              ("configs[1]: N = 6, depth 1, 2048 roots", 5, 1, 1, 2048)):
        ts32, tsb = searches(humans, 2, depth, width, dev)
        tot = same = 0
        worst_dv, worst_flip_gap, far = 0.0, 0.0, 0
        notes = []
        for seed in range(1, n_seeds + 1):
            robot, hum = bench.synth_scenes(seed, roots, humans)
            robot, hum = robot.to(dev), hum.to(dev)
            o32 = {k: v.clone() for k, v in ts32.search(robot, hum, roots_are_joint_states=False, want_root_values=True).items() if torch.is_tensor(v)}
            ob = {k: v.clone() for k, v in tsb.search(robot, hum, roots_are_joint_states=False, want_root_values=True).items() if torch.is_tensor(v)}
            eq = o32["best_action"] == ob["best_action"]
            tot += eq.numel()
            same += int(eq.sum())
            dv = (o32["best_value"] - ob["best_value"]).abs()
            worst_dv = max(worst_dv, float(dv.max()))
            far += int((dv > 1e-7).sum())
            for i in (dv > 1e-7).nonzero().flatten().tolist():
                notes.append("    seed %d root %d: kept root actions f32 %s / bf16x6 %s, their backed-up values f32 %s / bf16x6 %s" % (
                    seed, i, o32["root_kept"][i].tolist(), ob["root_kept"][i].tolist(),
                    ["%.9f" % v for v in o32["root_values"][i].tolist()], ["%.9f" % v for v in ob["root_values"][i].tolist()]))
            if not bool(eq.all()) and "root_values" in o32:
                # a flipped decision: how far apart the f32 kernels saw the two actions (a tie up to rounding, or a real gap?)
                idx = (~eq).nonzero().flatten()
                rv, rk = o32["root_values"][idx], o32["root_kept"][idx].long()
                hit = rk == ob["best_action"][idx].long().unsqueeze(1)                  # bf16x6's choice among f32's kept actions
                other = torch.where(hit, rv, torch.full_like(rv, float("-inf"))).max(dim=1).values
                gap = (o32["best_value"][idx] - other).abs()
                worst_flip_gap = max(worst_flip_gap, float(gap.max()))
        lines.append("%-46s %d seeds, %d decisions: %d identical (%d differ, largest f32 value gap between the two choices %.2e); "
                     "max |dV| of the best values %.2e, %d beyond 1e-7" % (name, n_seeds, tot, same, tot - same, worst_flip_gap, worst_dv, far))
        lines.extend(notes)
        print("\n".join([lines[-1 - len(notes)]] + notes), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_bf16x6_seed_soak.txt"), "w") as f:
        f.write("bf16x6 kernels against the f32 kernels, whole searches over seeded clearance-drawn scene sets (tools/r06_seed_soak.py)\n")
        f.write("\n".join(lines) + "\n")
        f.write("(a best value beyond 1e-7 with the same root decision and the same kept root actions, one of whose backed-up values agrees to\n"
                " 1e-9 while the other does not: the two modes kept different children on a near-tie INSIDE that subtree, and the deeper\n"
                " returns of the two candidates differ -- the same exposure the f32 kernels have against the oracle's summation order)\n")


if __name__ == "__main__":
    main()
