#!/usr/bin/env python
"""Single-decision latency of ModelPredictiveRL.predict(JointState) -- what Robot.act() sees once per simulated step --
eager launches vs. replay of a captured hipGraph (B = 1 root scene)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import relationalgraphlearning_amd as rga  # noqa: E402


class A:
    layers, depth, width, humans = 2, 2, 2, 5


def main():
    dev = torch.device("cuda:0")
    # wake the device first: the first configuration measured straight after process start read 0.9-1.0 ms per decision in rounds
    # 2 and 3 (profiles/r02_u_latency.txt) against 0.13 ms for every later one -- a fresh process's first ~0.3 s of GPU work
    # (code-object loads, clocks leaving the idle state) is not decision latency
    x = torch.randn(2048, 2048, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        x = (x @ x).clamp_(-1, 1)
    torch.cuda.synchronize()
    for H, D in ((5, 1), (5, 2), (19, 2), (19, 3)):
        A.humans, A.depth = H, D
        pol = bench.make_policy(A, dev)
        robot, humans = bench.synth_scenes(3, 1, H)
        js = rga.JointState(rga.FullState(*[float(x) for x in robot[0]]),
                            [rga.ObservableState(*[float(x) for x in row]) for row in humans[0]])
        for _ in range(40):                    # incl. the capture and the first replays of the fresh graph (they ran at ~1 ms
            pol.predict(js)                    # each for the first configuration of a process: the r02 / r03 "outlier")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            pol.predict(js)
        t_pred = (time.perf_counter() - t0) / n
        ts = pol.tree_search()
        r, h = robot.to(dev), humans.to(dev)
        for _ in range(5):                      # the eager path's own first calls (workspace, function attributes): not latency
            int(ts.search(r, h)["best_action"][0])
        t0 = time.perf_counter()
        for _ in range(n):
            out = ts.search(r, h)
            int(out["best_action"][0])
        t_search = (time.perf_counter() - t0) / n
        graph, out = ts.capture(r, h)
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
            int(out["best_action"][0])
        t_graph = (time.perf_counter() - t0) / n
        print("H=%2d D=%d  predict(JointState) %.3f ms | search+readback %.3f ms | graph replay+readback %.3f ms"
              % (H, D, t_pred * 1e3, t_search * 1e3, t_graph * 1e3))


if __name__ == "__main__":
    main()
