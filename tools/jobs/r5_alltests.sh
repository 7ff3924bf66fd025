#!/bin/bash
out=gpurun_out/${1:-r5_alltests}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
tail -8 $out/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
