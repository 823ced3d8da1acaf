#!/usr/bin/env python
"""Per-phase cycles of the value head's row kernel in the tile backward -- head_rows_kernel, or mlp_rows_kernel under
RGL_HEAD_ROWS_DIRECT=0 (debug build: make -C relationalgraphlearning_amd/csrc timing_head; each mark costs ~400 cycles itself)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RGL_HIP_LIBRARY", os.path.join(ROOT, "relationalgraphlearning_amd", "lib", "librgl_hip_timing_head.so"))
os.environ["RGL_BACKWARD_MFMA"] = "1"
from relationalgraphlearning_amd import _native as nat  # noqa: E402
from tests.helpers import make_mprl_policy  # noqa: E402
from tests.test_gpu_parity import seeded_scenes  # noqa: E402

NAMES = ["weights -> LDS + barrier (staged kernel)", "input rows -> LDS", "forward layers", "upstream gradient -> LDS",
         "per-layer staging (staged kernel)", "dW, db (slab stores)", "delta_in", "d_in rows out"]


def main():
    H, B = int(sys.argv[1]), int(sys.argv[2])
    dev = torch.device("cuda:0")
    pol = make_mprl_policy("trained", 1, device=dev)
    ve = pol.value_estimator
    robot, humans = seeded_scenes(3, B, H)
    r, h = robot.unsqueeze(1).to(dev), humans.to(dev)
    nat.lib()
    raw = C.CDLL(nat.LIB_PATH)
    buf = (C.c_ulonglong * 16)()
    for _ in range(3):
        ve((r, h)).sum().backward()
    torch.cuda.synchronize()
    raw.rgl_debug_read_backward_phase_cycles(buf, 1)
    reps = 20
    for _ in range(reps):
        ve((r, h)).sum().backward()
    torch.cuda.synchronize()
    raw.rgl_debug_read_backward_phase_cycles(buf, 1)
    tiles = (B + 15) // 16
    waves = tiles * 8 * reps
    tot = sum(buf[i] for i in range(8))
    print("value head rows, H = %d, batch %d: %d tiles, cycles per wave and launch" % (H, B, tiles))
    for i, nm in enumerate(NAMES):
        print("  %-36s %9.0f cycles  %5.1f %%" % (nm, buf[i] / waves, 100.0 * buf[i] / tot))
    print("  total %.0f cycles per wave; kernel clock / 100 MHz = %.1f" % (tot / waves, buf[8] / max(1, buf[9])))


if __name__ == "__main__":
    main()
