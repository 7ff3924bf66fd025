#!/bin/bash
# Round 6, second call: hamming clustering, the 20 000-medoid sweep with the faster oracle,
# then configs[3] on one GPU (small, then whole).
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_features.py tests/test_gpu_bigk.py -q -m gpu -x --durations=8 -k "hamming or twenty_thousand_medoids or feature_pam" > $out/tests.log 2>&1
tail -14 $out/tests.log
bash tools/jobs/r6_c4.sh ${out#gpurun_out/}/c4_small --frames-per-shard 131072 --centers 3000 --check-centers 120 --templates 2000
if grep -q '"ok": true' $out/c4_small/c4_one_gpu.json 2>/dev/null; then
  bash tools/jobs/r6_c4.sh ${out#gpurun_out/}/c4_full
fi
