#!/bin/bash
# round 4, first GPU call: the whole GPU suite (no -x: every failure in one pass), then the driver-style bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04a
python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/r04a/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04a/bench_driver_style.json 2> gpurun_out/r04a/bench_driver_style.err
python bench.py --cpu-seconds 0 > gpurun_out/r04a/bench_default.json 2> gpurun_out/r04a/bench_default.err
tail -5 gpurun_out/r04a/pytest.log
