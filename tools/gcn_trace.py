"""Path G (MultiHumanRL.predict, multi_human_rl.py:36-64; BASELINE configs[0]'s network): time of GCN.predict_batch for 2048 roots
and a roofline line for it -- the B x 81 rotated scenes each take ONE full graph forward of gcn.ValueNetwork (gcn.py:85-128: no crowd
sharing on this path, every action re-rotates every human into the robot's frame) and the 150-100-100-1 value head.
FLOPs per rotated scene (2 per MAC, softmax ~5 per element; what the algorithm needs, no MFMA padding):
    embeddings   6 -> 64 -> 32 on the robot row, 7 -> 64 -> 32 on H human rows
    similarity   X Wa (2 N 32 32), (X Wa) X^T (2 N N 32), row softmax (5 N N)
    layer 1      A X (2 N N 32), (A X) W1 (2 N 32 32)       -- every row
    layer 2      (A H1)[robot] (2 N 32), .W2 (2 32 32)      -- the value head reads node 0 only
    head         32 -> 150 -> 100 -> 100 -> 1
"""
import json
import os
import sys
import time

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from tests.helpers import make_gcn_policy  # noqa: E402
from tests.test_gpu_parity import seeded_scenes  # noqa: E402

PEAK = 157.3e12


def scene_flops(N):
    H = N - 1
    emb = 2 * (6 * 64 + 64 * 32) + H * 2 * (7 * 64 + 64 * 32)
    sim = 2 * N * 32 * 32 + 2 * N * N * 32 + 5 * N * N
    l1 = 2 * N * N * 32 + 2 * N * 32 * 32
    l2 = 2 * N * 32 + 2 * 32 * 32
    head = 2 * (32 * 150 + 150 * 100 + 100 * 100 + 100)
    return emb + sim + l1 + l2 + head


dev = torch.device("cuda:0")
ONLY_H = os.environ.get("RGL_GCN_TRACE_ONLY_H")          # counter passes: one crowd size, so per-kernel averages mean one shape
for H, B, mode in ((5, 2048, "f32"), (19, 2048, "f32"), (19, 2048, "bf16x6")):
    if ONLY_H and int(ONLY_H) != H:
        continue
    pol = make_gcn_policy(device=dev)
    pol.contraction_dtype = mode          # "bf16x6" (ABI 8): the graph's Wa / W_0 products of the scene kernel as six bf16 MFMA terms
    robot, humans = seeded_scenes(11, B, H)
    r, h = robot.to(dev), humans.to(dev)
    for _ in range(10):
        pol.predict_batch(r, h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        pol.predict_batch(r, h)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    A = len(pol.action_space)
    fl = scene_flops(H + 1)
    print("path G H=%d B=%d %s: %.3f ms per predict_batch" % (H, B, mode, ms))
    print(json.dumps({"workload": "path G, GCN.predict_batch: %d roots x %d actions, N = %d, contraction %s" % (B, A, H + 1, mode), "ms_per_batch": ms,
                      "rotated_scenes": B * A, "flop_per_rotated_scene": fl,
                      "roofline": {"bound": "mfma", "achieved": B * A * fl / (ms * 1e-3) / 1e12, "peak": PEAK / 1e12, "unit": "TFLOP/s",
                                   "frac": B * A * fl / (ms * 1e-3) / PEAK,
                                   "note": "wall time of the whole call (prepare, embeddings, scene kernel, head, argmax) over the "
                                           "algorithmic FLOPs of the B x A graph forwards; fp32 vector == f32-MFMA peak"}}))
