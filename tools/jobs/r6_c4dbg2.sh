#!/bin/bash
out=gpurun_out/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
run() { name=$1; shift; echo "== $name: $@"; timeout 150 python3 tools/c4_one_gpu.py --no-oracle "$@" > $out/$name.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $out/$name.log | grep -E "phase|shard [0-9]|failed" | cut -c1-420; }
B="--frames-per-shard 131072 --centers 3000 --check-centers 120 --templates 2000"
run repro $B
run repro2 $B
run cands16 $B --candidates 16
run cands8 $B --candidates 8
run cap4 $B --pick-cap 4
run ss0 $B --small-shards 0
run s4 $B --shards 4
run a100 $B --atoms 100
run onephase --frames-per-shard 131072 --centers 3000 --check-centers 3000 --templates 2000
