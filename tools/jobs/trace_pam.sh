#!/bin/bash
# rocprofv3 kernel trace of one PAM sweep (tools/lab_pam.py child); per-kernel
# table plus the timeline of one window in the middle of the sweep
# usage: trace_pam.sh <outdir-under-gpurun_out> [lib.so]
out=gpurun_out/$1; lib=$2
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 tools/lab_pam.py --centers 64 --reps 1 > /dev/null 2>&1     # makes the frames file
[ -n "$lib" ] && export ENSPARA_HIP_LIB=$GRAFT_REPO_ROOT/$lib
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/lab_pam.py --child /tmp/lab_frames_1000000_300.npy 5000 1 1 > $out/lab.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary.csv
python3 tools/summarize_profile.py window $f $out/window.txt
rm -rf $out/trace
tail -2 $out/lab.log
head -40 $out/kernel_summary.csv
