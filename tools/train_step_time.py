#!/usr/bin/env python
"""Time of one MPRLTrainer-style optimisation step (crowd_nav/utils/trainer.py:110-161) on the HIP path: batch of 100
transitions, value loss (TD target from a frozen copy) + state-predictor loss, two Adam optimizers."""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import make_mprl_policy  # noqa: E402
from tests.test_gpu_parity import seeded_scenes  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for H in (5, 19):
        pol = make_mprl_policy("trained", 1, device=dev)
        ve, sp = pol.value_estimator, pol.state_predictor
        target = copy.deepcopy(ve)
        v_opt = torch.optim.Adam(ve.parameters(), lr=1e-3)
        s_opt = torch.optim.Adam(sp.human_motion_predictor.parameters(), lr=1e-3)
        B = 100
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
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        print("H=%d: %.2f ms per optimisation step (batch 100, value + state-predictor update)" % (H, (time.perf_counter() - t0) * 20))


if __name__ == "__main__":
    main()
