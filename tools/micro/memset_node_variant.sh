#!/bin/bash
# Builds the library with ONE line changed back -- dH_L zeroed by hipMemsetAsync instead of init_rows_kernel -- into ab/ (git-ignored,
# travels with gpurun), for the same-box A/B of tools/micro/captured_step_repeatability.py:
#   bash tools/micro/memset_node_variant.sh                       (CPU: cross-compiles, ~2 min)
#   gpurun -- 'bash tools/micro/memset_node_variant.sh run'       (GPU: the tree's library, then the variant)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = run ]; then
  cd ${GRAFT_REPO_ROOT:-$ROOT}; mkdir -p gpurun_out
  { echo "== the tree's library (dH_L initialised by a kernel)"; python tools/micro/captured_step_repeatability.py ${2:-12} 2>&1 | grep "^H \|REPEATABLE"
    echo; echo "== variant: the same sources with hipMemsetAsync(dHL, 0, ...) in backward_tiles (ab/librgl_memset_node.so)"
    RGL_HIP_LIBRARY=$PWD/ab/librgl_memset_node.so python tools/micro/captured_step_repeatability.py ${2:-12} 2>&1 | grep "^H \|REPEATABLE"; } | tee gpurun_out/memset_node_ab.txt
  exit 0
fi
T=$(mktemp -d); mkdir -p $T/relationalgraphlearning_amd $ROOT/ab
cp -r $ROOT/relationalgraphlearning_amd/csrc $T/relationalgraphlearning_amd/; cp -r $ROOT/include $T/
python - "$T/relationalgraphlearning_amd/csrc/rgl_backward_mfma.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = "        hipLaunchKernelGGL(init_rows_kernel, dim3(blocks), dim3(256), 0, st, dHL, d_H, feat);\n"
new = ("        (void)blocks;\n        if (d_H) RGL_HIP_TRY(hipMemcpyAsync(dHL, d_H, feat * sizeof(float), hipMemcpyDeviceToDevice, st));\n"
       "        else RGL_HIP_TRY(hipMemsetAsync(dHL, 0, feat * sizeof(float), st));\n")
assert s.count(old) == 1
open(p, "w").write(s.replace(old, new))
PY
make -C $T/relationalgraphlearning_amd/csrc clean > /dev/null
make -C $T/relationalgraphlearning_amd/csrc -j16 2>&1 | grep -i " error" 
cp $T/relationalgraphlearning_amd/lib/librgl_hip.so $ROOT/ab/librgl_memset_node.so && echo "built ab/librgl_memset_node.so"
rm -rf $T
