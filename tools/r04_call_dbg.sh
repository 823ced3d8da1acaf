cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04lib; O=gpurun_out/r04lib/dbg.txt; : > $O
for lib in "" "$PWD/ab/librgl_head.so"; do
for f in "" "2"; do
  echo "== lib=[$lib] RGL_TILES_FORWARD=[$f]" >> $O
  env RGL_HIP_LIBRARY=$lib RGL_TILES_FORWARD=$f timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "non_default_value_heads" 2>&1 | grep -E "AssertionError: \(|passed|failed" >> $O
done; done
cat $O
