#!/usr/bin/env python
"""Per-phase cycles of mlp2_rows_kernel (debug build: make -C relationalgraphlearning_amd/csrc timing)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RGL_HIP_LIBRARY", os.path.join(ROOT, "relationalgraphlearning_amd", "lib", "librgl_hip_timing.so"))
os.environ["RGL_BACKWARD_MFMA"] = "1"
from relationalgraphlearning_amd import _native as nat  # noqa: E402
from tests.helpers import make_mprl_policy  # noqa: E402
from tests.test_gpu_parity import seeded_scenes  # noqa: E402

NAMES = ["weights -> LDS + barrier", "rows -> LDS, next tile's fetch issued", "layer 0", "layer 1", "output rows / delta1 pass",
         "dW1, db1, delta0", "dW0, db0", "d_in"]


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
    ve((r, h)).sum().backward()
    raw.rgl_debug_read_backward_phase_cycles(buf, 1)
    ve((r, h)).sum().backward()
    raw.rgl_debug_read_backward_phase_cycles(buf, 1)
    tot = sum(buf[i] for i in range(8))
    for i, nm in enumerate(NAMES):
        print("  %-32s %12d wave-cycles  %5.1f %%" % (nm, buf[i], 100.0 * buf[i] / tot))
    print("  total %d wave-cycles; kernel clock / 100 MHz = %.1f" % (tot, buf[8] / max(1, buf[9])))


if __name__ == "__main__":
    main()
