#!/usr/bin/env python
"""Time of one MPRLTrainer-style optimisation step (crowd_nav/utils/trainer.py:110-161) on the HIP path: value loss (TD target from
a frozen copy) + state-predictor loss, two Adam optimizers; batch 100 (the reference's) and 4096 (what the vector explorer can
feed).  Prints wall time per step and a roofline line: algorithmic FLOPs of the step's graph forwards and backwards (SURVEY 8d
terms: a forward = 328 120 / 394 192 FLOP at N = 20 for the value / predictor graph; a backward counted as two forwards) over the
step time, against the fp32 peak."""
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import make_mprl_policy  # noqa: E402
from tests.test_gpu_parity import seeded_scenes  # noqa: E402

PEAK = 157.3e12


def forward_flops(N, L=2, head=28648, motion=False):
    H = N - 1
    f = 5248 + H * 4736 + 2 * N * 32 * 32 + 2 * N * N * 32 + 5 * N * N + L * (2 * N * N * 32 + 2 * N * 32 * 32 + 2 * N * 32)
    return f + (N * 4736 if motion else head)


def main():
    dev = torch.device("cuda:0")
    graph_mode = "--graph" in sys.argv     # the whole optimisation step captured once into a hipGraph and replayed
    for H in (5, 19):
        for B in (100, 4096):
            pol = make_mprl_policy("trained", 1, device=dev)
            ve, sp = pol.value_estimator, pol.state_predictor
            target = copy.deepcopy(ve)
            v_opt = torch.optim.Adam(ve.parameters(), lr=1e-3, capturable=graph_mode)
            s_opt = torch.optim.Adam(sp.human_motion_predictor.parameters(), lr=1e-3, capturable=graph_mode)
            robot, humans = seeded_scenes(3, B, H)
            robot2, humans2 = seeded_scenes(4, B, H)
            r, h, r2, h2 = robot.unsqueeze(1).to(dev), humans.to(dev), robot2.unsqueeze(1).to(dev), humans2.to(dev)
            rew = torch.zeros(B, 1, device=dev)
            crit = torch.nn.MSELoss()

            def step():
                v_opt.zero_grad()
                out = ve((r, h))
                with torch.no_grad():
                    tgt = rew + 0.9 * target((r2, h2))
                loss = crit(out, tgt)
                loss.backward()
                v_opt.step()
                s_opt.zero_grad()
                _, nh = sp((r, h), None, detach=True)
                l2 = crit(nh, h2)
                l2.backward()
                s_opt.step()
            run = step
            if graph_mode:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        step()
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    step()
                run = g.replay
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            reps = 50 if B == 100 else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            N = H + 1
            # value forward + target forward + value backward (2 forwards) + predictor forward + predictor backward (2)
            flops = B * (4 * forward_flops(N) + 3 * forward_flops(N, motion=True))
            print(json.dumps({"workload": "MPRLTrainer-style optimisation step, H=%d, batch %d%s" % (H, B, ", replayed from a captured hipGraph" if graph_mode else ""), "ms_per_step": ms,
                              "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": PEAK / 1e12,
                                           "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / PEAK,
                                           "flops_per_step": flops,
                                           "note": "forwards on the one-wave-per-scene MFMA kernel; backward: the per-scene VALU kernel "
                                                   "below 256 scenes of an eager step (launch / host bound there), the MFMA tile "
                                                   "pipeline of rgl_backward_mfma.hip from 256 scenes and in every captured step"}}))


if __name__ == "__main__":
    main()
