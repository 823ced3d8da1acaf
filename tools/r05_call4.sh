#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gradients_on_the_mfma or non_default_similarity or reference_trainer_fixture or exact_ties or growing or tile_pipeline_forced" > $O/r05_d_tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/r05_d_tests.log
