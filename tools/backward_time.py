#!/usr/bin/env python
"""Time of the backward pass alone (rgl_graph_backward_f32 through autograd) by batch size, per-scene VALU kernel
(RGL_BACKWARD_MFMA=0) versus the tile pipeline on the matrix cores (=1): value estimator and state predictor, H = 5 / 19."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import make_mprl_policy  # noqa: E402
from tests.test_gpu_parity import seeded_scenes  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    pol = make_mprl_policy("trained", 1, device=dev, layerwise=os.environ.get("BT_LAYERWISE") == "1")     # BT_LAYERWISE=1: an adjacency per layer
    ve, sp = pol.value_estimator, pol.state_predictor
    print("| H | batch | module | per-scene kernel ms | tile pipeline ms | ratio |")
    print("|---|---|---|---|---|---|")
    Hs = [int(x) for x in os.environ.get("BT_H", "5,19").split(",")]
    Bs = [int(x) for x in os.environ.get("BT_B", "100,256,512,1024,4096,16384").split(",")]
    for H in Hs:
        for B in Bs:
            robot, humans = seeded_scenes(3, B, H)
            r, h = robot.unsqueeze(1).to(dev), humans.to(dev)
            for name in ("value", "predictor"):
                ms = {}
                for mode in os.environ.get("BT_MODES", "0,1").split(","):
                    os.environ["RGL_BACKWARD_MFMA"] = mode

                    def fwd():
                        if name == "value":
                            return ve((r, h)).sum()
                        return sp((r, h), None, detach=False)[1].sum()
                    for _ in range(3):
                        fwd().backward()
                    reps = 20 if B <= 1024 else 5
                    tot = 0.0
                    for _ in range(reps):
                        loss = fwd()
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        loss.backward()
                        e1.record()
                        torch.cuda.synchronize()
                        tot += e0.elapsed_time(e1)
                    ms[mode] = tot / reps
                if len(ms) == 2:
                    print("| %d | %d | %s | %.3f | %.3f | %.1fx |" % (H, B, name, ms["0"], ms["1"], ms["0"] / ms["1"]))
                else:
                    print("| %d | %d | %s | %s |" % (H, B, name, ms))


if __name__ == "__main__":
    main()
