#!/usr/bin/env python
"""Per-phase cycle breakdown of the stage-1 kernels, children_rank1_kernel or children_deep_kernel (debug build:
make -C relationalgraphlearning_amd/csrc timing; per-wave s_memtime deltas, i.e. core-clock cycles).

    RGL_HIP_LIBRARY=relationalgraphlearning_amd/lib/librgl_hip_timing.so python tools/phase_timing.py
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RGL_HIP_LIBRARY", os.path.join(ROOT, "relationalgraphlearning_amd", "lib", "librgl_hip_timing.so"))
import bench  # noqa: E402
from relationalgraphlearning_amd import _native as nat  # noqa: E402


class A:
    layers, depth, width, humans, contraction = 2, 2, 2, 19, "f32"


DEEP_NAMES = ["loop top (end barrier)", "phase A: embeddings", "barrier", "phase B: relation block / S row+col",
              "barrier + E load", "(unused)", "phase C: row scalars a, b + robot row", "per-child loop"]


def main():
    """usage: phase_timing.py [parents] [humans layers contraction]   (N > 32 or 3 layers -> the deep kernel's phases)"""
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    if len(sys.argv) > 4:
        A.humans, A.layers, A.contraction = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    deep = A.layers == 3 or A.humans + 1 > 32
    dev = torch.device("cuda:0")
    pol = bench.make_policy(A, dev)
    ts = pol.tree_search()
    robot, humans = bench.synth_scenes(5, P, A.humans)
    ex = ts.expand(robot.to(dev), humans.to(dev), parents_are_joint_states=False)
    lib = nat.lib()
    raw = C.CDLL(nat.LIB_PATH)
    buf = (C.c_ulonglong * 16)()
    fused = (not deep) and os.environ.get("RGL_CHILDREN_TWO_STAGE", "0") != "1"
    read = raw.rgl_debug_read_deep_phase_cycles if deep else (raw.rgl_debug_read_fused_phase_cycles if fused
                                                              else raw.rgl_debug_read_phase_cycles)
    read(buf, 1)
    reps = 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ts.value_children(ex["child_robot"], ex["humans_next"])
    e1.record()
    torch.cuda.synchronize()
    read(buf, 1)
    names = DEEP_NAMES if deep else ["loop top", "embed-1 (x0, y, g0) / crowd-1 work", "mid barrier",
                                     "embed-2 (S row+col, p, pXh, a/b) / crowd-2 work", "barrier", "row phase work",
                                     "row barrier", "robot-row pass"]
    waves = 8 * P * reps          # per (wave, parent); rank-1 kernel, 8 waves per workgroup
    if fused:                     # per tile: one wave, 16 children
        names = ["loop top / value store", "tile loads issued", "embedding (x0, y, g0)", "S row+col, p, pXh, a/b", "row pass",
                 "robot row", "head", "(unused)"]
        waves = P * 6 * reps
    tot = sum(buf[i] for i in range(8))
    print("P=%d  %.3f ms per call (stage 1+2)" % (P, e0.elapsed_time(e1) / reps))
    for i, nm in enumerate(names):
        print("  %-18s %9.0f cycles per wave per parent   %5.1f %%" % (nm, buf[i] / waves, 100.0 * buf[i] / tot))
    print("  total              %9.0f" % (tot / waves))
    if buf[9]:
        print("  s_memtime / s_memrealtime (100 MHz) over the kernel: %.1f -> counter clock %.0f MHz" % (buf[8] / buf[9], 100.0 * buf[8] / buf[9]))


if __name__ == "__main__":
    main()
