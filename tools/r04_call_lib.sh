cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04lib
timeout 600 python -m pytest tests -x -q -m gpu -k "state_predictor or expand_level or scene_kernel or tree_vs_batched or f16x3_mode" 2>&1 | tail -3 > gpurun_out/r04lib/tests.txt
cat gpurun_out/r04lib/tests.txt
bash tools/r04_share_ab.sh RGL_HIP_LIBRARY=$PWD/ab/librgl_prev.so
