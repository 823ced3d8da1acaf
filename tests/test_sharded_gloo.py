"""world_size-2 test of the multi-GPU path on CPU (gloo): shard the roots, one all-gather.

The per-shard search is injected (here: the CPU oracle), so the test exercises exactly the code
bench.py runs under RCCL -- ShardedRollout -- without a GPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import golden_io as gio


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from relationalgraphlearning_amd import ShardedRollout
    from oracle import rgl_oracle as orc
    torch.set_num_threads(1)
    pl = gio.load("planning")
    robot = torch.tensor(np.tile(pl["plan.scene.s5.robot"], (3, 1))[:total].astype(np.float32))
    humans = torch.tensor(np.tile(pl["plan.scene.s5.humans"], (3, 1, 1))[:total].astype(np.float32))
    robot[:, 0] += torch.arange(total) * 0.01            # make every root distinct
    P = gio.oracle_params("trained")
    cfg = orc.OracleConfig(planning_depth=2, planning_width=2, do_action_clip=True)
    calls = []

    def search(r, h):
        # root by root: a root's result must not depend on which other roots share its batch (MKL's summation order does), or the
        # comparisons across different shard sizes below would only hold by luck
        calls.append(r.shape[0])
        with torch.no_grad():
            outs = [orc.mprl_predict_batched(r[i:i + 1], h[i:i + 1], P, cfg)[:2] for i in range(r.shape[0])]
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    sr = ShardedRollout(search)
    act, val = sr.run(robot, humans)
    assert act.dtype == torch.int64 and val.dtype == torch.float32        # run(): torch's index type
    act = act.int()                                                       # launch_local / run_local: the search's own int32
    # pipelined use (bench.py): two steps in flight, results collected one step late
    from relationalgraphlearning_amd import shard_bounds
    lo, hi = shard_bounds(total, world, rank)
    h1 = sr.launch_local(robot[lo:hi], humans[lo:hi], total)
    h2 = sr.launch_local(robot[lo:hi], humans[lo:hi], total)
    a1, v1 = h1.result()
    a2, v2 = h2.result()
    assert torch.equal(a1, act) and torch.equal(v1, val) and torch.equal(a2, act) and torch.equal(v2, val)
    assert h1.result()[0] is a1                          # idempotent
    # equal shards take the no-copy unpacking path
    even = total - total % world
    he = sr.launch_local(robot[:even][rank * (even // world):(rank + 1) * (even // world)],
                         humans[:even][rank * (even // world):(rank + 1) * (even // world)], even)
    ae, ve = he.result()
    assert torch.equal(ae, act[:even]) and torch.equal(ve, val[:even])
    # search_into: the search writes int32 actions / fp32 values straight into the exchange buffer (what bench.py uses)
    def search_into(r, h, act_out, val_out):
        a, v = search(r, h)
        assert act_out.dtype == torch.int32 and val_out.dtype == torch.float32 and act_out.is_contiguous()
        act_out.copy_(a)
        val_out.copy_(v)
    sr2 = ShardedRollout(search, search_into=search_into)
    hx = sr2.launch_local(robot[lo:hi], humans[lo:hi], total).wait()
    ax, vx = hx.result()
    assert torch.equal(ax, act) and torch.equal(vx, val)
    # static exchange buffers (what a hipGraph-captured search needs): fixed addresses handed out in turn, same results
    sr3 = ShardedRollout(search, search_into=search_into)
    views = sr3.use_static_buffers(total, robot.device)
    seen = []

    def search_into_static(r, h, act_out, val_out):
        seen.append(act_out.data_ptr())
        search_into(r, h, act_out, val_out)
    sr3.search_into = search_into_static
    hs = [sr3.launch_local(robot[lo:hi], humans[lo:hi], total) for _ in range(3)]
    for hnd in hs[:2]:
        hnd.wait()
    for hnd in hs[2:]:
        ay, vy = hnd.result()
        assert torch.equal(ay, act) and torch.equal(vy, val)
    assert seen == [views[0][0].data_ptr(), views[1][0].data_ptr(), views[0][0].data_ptr()]
    del calls[1:]
    torch.save({"act": act, "val": val, "calls": calls}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("total,world,shards", [(7, 2, [4, 3]),                     # uneven split: 4 + 3
                                                (9, 8, [2, 1, 1, 1, 1, 1, 1, 1]),      # the node the driver scales to: 8 ranks, uneven
                                                (8, 8, [1] * 8)])                      # ... and the equal-shard (no-copy) unpacking
def test_sharded_rollout_matches_single_process(total, world, shards, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    assert [r["calls"] for r in res] == [[n] for n in shards]
    for r in res[1:]:
        assert torch.equal(res[0]["act"], r["act"]) and torch.equal(res[0]["val"], r["val"])
    from oracle import rgl_oracle as orc
    pl = gio.load("planning")
    robot = torch.tensor(np.tile(pl["plan.scene.s5.robot"], (3, 1))[:total].astype(np.float32))
    humans = torch.tensor(np.tile(pl["plan.scene.s5.humans"], (3, 1, 1))[:total].astype(np.float32))
    robot[:, 0] += torch.arange(total) * 0.01
    cfg = orc.OracleConfig(planning_depth=2, planning_width=2, do_action_clip=True)
    with torch.no_grad():
        outs = [orc.mprl_predict_batched(robot[i:i + 1], humans[i:i + 1], gio.oracle_params("trained"), cfg)[:2] for i in range(total)]
    a, v = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    assert torch.equal(res[0]["act"], a) and torch.equal(res[0]["val"], v)
