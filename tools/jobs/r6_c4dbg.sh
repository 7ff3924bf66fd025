#!/bin/bash
# debugging matrix for tools/c4_one_gpu.py (mailbox time-outs with eight big shards)
out=gpurun_out/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
run() { name=$1; shift; echo "== $name: $@"; timeout 120 python3 tools/c4_one_gpu.py --no-oracle "$@" > $out/$name.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $out/$name.log | grep -E "phase|failed|properties" | cut -c1-330; }
B="--frames-per-shard 131072 --centers 400 --check-centers 120 --templates 2000"
run repro $B
run cands16 $B --candidates 16
run cands8 $B --candidates 8
run a100 $B --atoms 100
run s4 $B --shards 4
run s2 $B --shards 2
run small --frames-per-shard 16384 --centers 400 --check-centers 120 --templates 2000
run onephase --frames-per-shard 131072 --centers 400 --check-centers 400 --templates 2000
ENSPARA_NO_TORCH=1 run notorch $B
GPU_MAX_HW_QUEUES=32 run q32 $B
