cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04lib
timeout 1200 python -m pytest tests -x -q -m gpu -k "state_predictor or expand_level or scene_kernel or tree_vs_batched or properties_of_the_other or beyond_64" 2>&1 | tail -3 > gpurun_out/r04lib/tests.txt
cat gpurun_out/r04lib/tests.txt
bash tools/r04_c4_ab.sh RGL_SCENE_WIDE_SLOTS=1
