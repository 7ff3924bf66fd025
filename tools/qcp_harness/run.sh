#!/bin/sh
# Compiles enspara_amd/csrc/ek_qcp.h for the host (empty hip_runtime.h stub) and
# runs ek_rmsd_from_S_below against ek_rmsd_from_S over ~140 M random cases:
# "wrong" counts the cases where the early stop returned +inf although the full
# iteration ends below `cur`.  Takes a few minutes on one core.
set -e
here=$(cd "$(dirname "$0")" && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/hip" && : > "$tmp/hip/hip_runtime.h"
for t in generic near_collinear coincident_roots; do
    g++ -O2 -ffp-contract=off -I"$tmp" -I"$here/../../enspara_amd/csrc" \
        "$here/$t.cpp" -o "$tmp/$t"
    echo "== $t"; "$tmp/$t" | grep -v '^  '
done
rm -rf "$tmp"
