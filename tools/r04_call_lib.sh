cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04lib
timeout 1200 python -m pytest tests -x -q -m gpu -k "clamp_bit or value_children or tile_kernel_variant or non_default or f16_contraction or deep_kernel or f16_tree or baseline_workloads or other_similarities or tree_vs_batched or properties_of_the_other or beyond_64 or crowds" 2>&1 | tail -7 > gpurun_out/r04lib/tests.txt
cat gpurun_out/r04lib/tests.txt
bash tools/r04_c4_ab.sh RGL_DEEP_FUSE_HEAD=0
