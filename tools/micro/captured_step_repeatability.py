#!/usr/bin/env python
"""Run-to-run repeatability of the public trainer's captured steps: MPRLTrainer.optimize_epoch (imitation) followed by optimize_batch
(RL) on a seeded memory, repeated in one process -- every repetition must end at bit-identical parameters, and within float32 noise
of the eager trainer's.  (Found in round 5: a hipMemsetAsync NODE inside the replayed step was not reliably ordered against the
kernels around it -- NaN parameters on one box, sporadic 2e-3 deviations on another, nothing on a third.)
usage: captured_step_repeatability.py [repetitions]"""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import relationalgraphlearning_amd as rga  # noqa: E402
from tests.helpers import make_mprl_policy  # noqa: E402
from tests.test_gpu_parity import seeded_scenes, _Writer  # noqa: E402


def run(dev, data, epochs, batches, capture=True):
    robot, humans, robot2, humans2, rew = data
    n, H = robot.shape[0], humans.shape[1]
    pol = make_mprl_policy("trained", 1, device=dev)
    mem = rga.ReplayMemory(n)
    for i in range(n):
        mem.push((robot[i:i + 1].to(dev), humans[i].to(dev), rew[i:i + 1].to(dev), rew[i:i + 1].to(dev), robot2[i:i + 1].to(dev),
                  humans2[i].to(dev)))
    cls = rga.MPRLTrainer if capture else type("EagerTrainer", (rga.MPRLTrainer,), {"capture": False})
    t = cls(pol.value_estimator, pol.state_predictor, mem, dev, pol, _Writer(), 100, "Adam", H, reduce_sp_update_frequency=False,
            freeze_state_predictor=False, detach_state_predictor=True, share_graph_model=False)
    t.set_learning_rate(1e-3)
    t.update_target_model(pol.value_estimator)
    torch.manual_seed(3)
    if epochs:
        t.optimize_epoch(epochs)
    torch.manual_seed(4)
    if batches:
        t.optimize_batch(batches, 0)
    torch.cuda.synchronize()
    return torch.cat([p.detach().flatten() for p in list(pol.value_estimator.parameters()) + list(pol.state_predictor.parameters())]).cpu()


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    worst = 0
    for H in (5, 19):
        n = 300
        data = seeded_scenes(21, n, H) + seeded_scenes(22, n, H) + (torch.rand(n, generator=torch.Generator().manual_seed(5)),)
        for e, b in ((2, 2), (3, 3), (0, 3), (2, 0)):
            eager = run(dev, data, e, b, capture=False)
            outs = [run(dev, data, e, b) for _ in range(reps)]
            c = Counter("%.1e" % float((o - outs[0]).abs().max()) for o in outs)
            bad = sum(1 for o in outs if not torch.equal(o, outs[0]))
            worst += bad
            print("H %d: %d epochs + %d batch calls, %d repetitions: %d differ from the first %s; captured vs eager %.1e"
                  % (H, e, b, reps, bad, dict(c), float((outs[0] - eager).abs().max())), flush=True)
    print("REPEATABLE" if worst == 0 else "NOT REPEATABLE (%d)" % worst)
    return 0 if worst == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
