#!/bin/bash
out=gpurun_out/${1:-r5_newtests}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_golden.py -x -q -m gpu -k "polynomial or one_workgroup" > $out/tests.log 2>&1
tail -15 $out/tests.log
