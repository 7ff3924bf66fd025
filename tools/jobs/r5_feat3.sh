#!/bin/bash
out=gpurun_out/${1:-r5_feat3}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_features.py -x -q -m gpu > $out/tests_features.log 2>&1
tail -15 $out/tests_features.log
