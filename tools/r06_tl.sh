#!/bin/bash
# timeline of the configs[2] step with the product library and (optionally) a variant library, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/${TAG}.md
rm -f $O
cd $R
bash tools/timeline.sh $O --roots 2048
for lib in "$@"; do
  echo "# variant library $lib" >> $O
  RGL_HIP_LIBRARY=$R/relationalgraphlearning_amd/lib/$lib bash tools/timeline.sh $O --roots 2048
done
cat $O
