#!/bin/bash
out=gpurun_out/${1:-r5_t21}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_golden.py -x -q -m gpu > $out/tests.log 2>&1
tail -5 $out/tests.log
