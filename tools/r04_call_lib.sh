cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04lib
timeout 1200 python -m pytest tests -x -q -m gpu -k "reward_kats or expand_level or deep_kernel_searches or f16_tree or baseline_workloads or beyond_64 or planner_step_methods or root_reward" 2>&1 | tail -3 > gpurun_out/r04lib/tests.txt
cat gpurun_out/r04lib/tests.txt
bash tools/r04_c4_ab.sh RGL_HIP_LIBRARY=$PWD/ab/librgl_prev.so
