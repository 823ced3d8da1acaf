#!/usr/bin/env python
"""Models outside the shipped shapes (X_dim = 64, other wr_dims / wh_dims): time of a depth-2 search step and of a value-estimator
forward on the tile kernels of rgl_backward_mfma.hip against the general VALU kernel (RGL_TILES_FORWARD=0)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, time, torch
sys.path.insert(0, %r)
import relationalgraphlearning_amd as rga
from relationalgraphlearning_amd.config import policy_config
import bench
dev = torch.device("cuda:0")
for X, wr, wh, H, B in ((64, [64, 64], [64, 64], 19, 256), (32, [128, 64, 32], [48, 32], 19, 256), (64, [64, 64], [64, 64], 19, 2048)):
    cfgp = policy_config("model_predictive_rl", gcn__num_layer=2, gcn__X_dim=X, gcn__final_state_dim=X, gcn__wr_dims=wr, gcn__wh_dims=wh,
                         model_predictive_rl__planning_depth=2, model_predictive_rl__planning_width=2,
                         model_predictive_rl__do_action_clip=True, model_predictive_rl__value_network_dims=[X, 100, 100, 1])
    torch.manual_seed(1)
    pol = rga.ModelPredictiveRL()
    pol.time_step = 0.25
    pol.configure(cfgp)
    with torch.no_grad():
        for gm in (pol.value_estimator.graph_model, pol.state_predictor.graph_model):
            for n_, p_ in gm.named_parameters():
                if n_ == "w_a" or n_.startswith("Ws"):
                    p_.mul_(1.0 / X ** 0.5)
    pol.set_time_step(0.25); pol.set_phase("test"); pol.set_device(dev)
    robot, humans = bench.synth_scenes(7, B, H)
    r, h = robot.to(dev), humans.to(dev)
    for _ in range(5):
        pol.predict_batch(r, h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pol.predict_batch(r, h)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("X_dim %%d wr %%s wh %%s, N = %%d, depth 2, %%d roots: %%.3f ms per search step (%%.3g evals/s)" %% (X, wr, wh, H + 1, B, ms, B * 249 / ms * 1e3))
''' % ROOT


def main():
    for mode, name in (("1", "tile kernels (MFMA)"), ("0", "general VALU kernel")):
        env = dict(os.environ, RGL_TILES_FORWARD=mode)
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
        print("== " + name)
        print(out.stdout.strip() if out.returncode == 0 else out.stderr[-2000:])


if __name__ == "__main__":
    main()
