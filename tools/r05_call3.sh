#!/bin/bash
# round 5, third GPU call: the whole GPU suite with every f32 search in the bf16x6 mode (admission run), the suite as it is,
# and the driver-style bench line with the new default (`--contraction auto`)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
{ echo "# RGL_CONTRACT_F32_AS=bf16x6 python -m pytest tests -m gpu -q   (every search that asks for f32 runs RGL_CONTRACT_BF16X6;"
  echo "# the f32 bounds of the suite -- north star 1e-4, regression level REG_F32 = 1e-6 -- are held against it), source revision $1"
  RGL_CONTRACT_F32_AS=bf16x6 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -120; } > $O/r05_suite_under_bf16x6.txt
tail -4 $O/r05_suite_under_bf16x6.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/r05_c_gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -3 $O/r05_c_gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $O/r05_c_bench_driver_style.json
python - <<PY
import json
r = json.load(open("$O/r05_c_bench_driver_style.json"))
print("value %.4g ms_per_step %.4f dtype %s" % (r["value"], r["ms_per_step"], r["dtype"][:60]))
print("roofline", {k: r["roofline"][k] for k in ("achieved", "peak", "frac", "launch_ms")})
print("f32 line", {k: r["f32_mfma_line"][k] for k in ("value", "ms_per_step", "max_abs_dV_vs_value_kernels", "identical_decisions")})
print("float64", r["cpu_baseline"]["float64_check"])
PY
