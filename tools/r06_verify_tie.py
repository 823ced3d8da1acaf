import os, sys, types, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
ts = {}
for mode in ("f32", "bf16x6"):
    a = types.SimpleNamespace(layers=2, depth=3, width=2, humans=19, contraction=mode)
    ts[mode] = bench.make_policy(a, dev).tree_search()
robot, hum = bench.synth_scenes(18, 512, 19)
robot, hum = robot.to(dev)[325:326].contiguous(), hum.to(dev)[325:326].contiguous()
def level(mode, r, h, tag):
    ex = ts[mode].expand(r, h, parents_are_joint_states=False)
    v1 = ex["value1"][0]
    top = torch.topk(v1, 4)
    print("%-8s %-28s top-4 one-step estimates: %s" % (mode, tag, ", ".join("a%d %.10f" % (i, v) for v, i in zip(top.values.tolist(), top.indices.tolist()))))
    return ex, top.indices.tolist()
for mode in ("f32", "bf16x6"):
    ex0, k0 = level(mode, robot, hum, "root")
    r1 = ex0["child_robot"][0, 50:51].contiguous(); h1 = ex0["humans_next"][0:1].contiguous()
    ex1, k1 = level(mode, r1, h1, "child 50 of the root")
    for c in k1[:3]:
        r2 = ex1["child_robot"][0, c:c + 1].contiguous(); h2 = ex1["humans_next"][0:1].contiguous()
        level(mode, r2, h2, "grandchild %d" % c)
