import sys, os, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from tests.helpers import make_gcn_policy
from tests.test_gpu_parity import seeded_scenes
dev = torch.device("cuda:0")
for H, B in ((5, 2048), (19, 2048)):
    pol = make_gcn_policy(device=dev)
    robot, humans = seeded_scenes(11, B, H)
    r, h = robot.to(dev), humans.to(dev)
    for _ in range(3): pol.predict_batch(r, h)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pol.predict_batch(r, h)
    torch.cuda.synchronize(); print("path G H=%d B=%d: %.3f ms per predict_batch" % (H, B, (time.perf_counter() - t0) / 20 * 1e3))
