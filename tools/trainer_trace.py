#!/usr/bin/env python
"""One case of tools/trainer_time.py for a kernel trace: MPRLTrainer.optimize_batch at H humans, batch 100, captured step,
index-sampled batches.  `rocprofv3 --kernel-trace --stats -- python tools/trainer_trace.py [H] [calls]`: every kernel's call count
divided by the batches run (printed) is the node count of one captured step."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relationalgraphlearning_amd as rga  # noqa: E402
from tests.helpers import make_mprl_policy  # noqa: E402
from tests.test_gpu_parity import seeded_scenes, _Writer  # noqa: E402


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    B, n_mem, batches = 100, 3000, 29
    robot, humans = seeded_scenes(3, n_mem, H)
    robot2, humans2 = seeded_scenes(4, n_mem, H)
    rew = torch.rand(n_mem)
    mem = rga.ReplayMemory(n_mem)
    for i in range(n_mem):
        mem.push((robot[i:i + 1].to(dev), humans[i].to(dev), rew[i:i + 1].to(dev), rew[i:i + 1].to(dev), robot2[i:i + 1].to(dev),
                  humans2[i].to(dev)))
    pol = make_mprl_policy("trained", 1, device=dev)
    t = rga.MPRLTrainer(pol.value_estimator, pol.state_predictor, mem, dev, pol, _Writer(), B, "Adam", H, reduce_sp_update_frequency=False,
                        freeze_state_predictor=False, detach_state_predictor=True, share_graph_model=False)
    t.set_learning_rate(1e-3)
    t.update_target_model(pol.value_estimator)
    for e in range(calls):
        t.optimize_batch(batches - 1, e)
    torch.cuda.synchronize()
    print(json.dumps({"batches_run": calls * batches, "of_them_eager_warmup": 1, "graphs_captured": len(t._steps)}))


if __name__ == "__main__":
    main()
