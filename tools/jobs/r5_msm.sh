#!/bin/bash
out=gpurun_out/${1:-r5_msm}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_msm.py -x -q -m gpu > $out/tests_msm.log 2>&1
tail -5 $out/tests_msm.log
timeout 900 python3 tools/eig_probe.py 1000 2>&1 | grep -v amdgpu.ids | tee $out/eig_probe.log
