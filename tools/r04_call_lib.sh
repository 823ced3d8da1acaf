cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04lib
[ -f tools/_dbg_f16.py ] && python tools/_dbg_f16.py 2>&1 | grep -v amdgpu.ids | cut -c1-60 > gpurun_out/r04lib/dbg.txt; cat gpurun_out/r04lib/dbg.txt
timeout 1200 python -m pytest tests -x -q -m gpu -k "value_children or fused or tile_kernel_variant or non_default or f16_contraction or deep_kernel or f16_tree or baseline_workloads or forced_kernel or other_similarities or tree_vs_batched or properties_of_the_other" 2>&1 | tail -5 > gpurun_out/r04lib/tests.txt
cat gpurun_out/r04lib/tests.txt
bash tools/r04_lib_ab.sh
