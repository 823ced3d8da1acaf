#!/usr/bin/env python
"""The optimisation step THROUGH THE PUBLIC TRAINER (relationalgraphlearning_amd.MPRLTrainer, the reference's contract:
crowd_nav/utils/trainer.py:110-161): wall time per batch of optimize_batch() at the reference's batch size 100 and at 4096, from a
ReplayMemory of device tuples, with the captured-step replay (default) and with every step eager (capture switched off), and the
reference-style DataLoader path beside the index-sampled one.  One JSON line per case."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relationalgraphlearning_amd as rga  # noqa: E402
from tests.helpers import make_mprl_policy  # noqa: E402
from tests.test_gpu_parity import seeded_scenes, _ListDataset, _Writer  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for H in (5, 19):
        for B, n_mem, batches in ((100, 3000, 29), (4096, 8192 * 3, 5)):
            robot, humans = seeded_scenes(3, n_mem, H)
            robot2, humans2 = seeded_scenes(4, n_mem, H)
            rew = torch.rand(n_mem)
            items = [(robot[i:i + 1].to(dev), humans[i].to(dev), rew[i:i + 1].to(dev), rew[i:i + 1].to(dev), robot2[i:i + 1].to(dev),
                      humans2[i].to(dev)) for i in range(n_mem)]
            for kind in ("captured step, index-sampled batches (ReplayMemory.as_tensors)", "captured step, torch DataLoader batches",
                         "eager step, index-sampled batches"):
                pol = make_mprl_policy("trained", 1, device=dev)
                if "DataLoader" in kind:
                    mem = _ListDataset(items)
                else:
                    mem = rga.ReplayMemory(n_mem)
                    for it in items:
                        mem.push(it)
                cls = rga.MPRLTrainer if "captured" in kind else type("EagerTrainer", (rga.MPRLTrainer,), {"capture": False})
                t = cls(pol.value_estimator, pol.state_predictor, mem, dev, pol, _Writer(), B, "Adam", H, reduce_sp_update_frequency=False,
                        freeze_state_predictor=False, detach_state_predictor=True, share_graph_model=False)
                t.set_learning_rate(1e-3)
                t.update_target_model(pol.value_estimator)
                t.optimize_batch(batches - 1, 0)                     # captures (one shape: memory size is a multiple of the batch)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 3
                for e in range(reps):
                    t.optimize_batch(batches - 1, e + 1)             # upstream's off-by-one: `batches` steps
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / (reps * batches) * 1e3
                print(json.dumps({"workload": "MPRLTrainer.optimize_batch, H=%d, batch %d, %s" % (H, B, kind), "ms_per_batch": ms,
                                  "batches_per_call": batches, "graphs_captured": len(t._steps)}), flush=True)


if __name__ == "__main__":
    main()
