#!/bin/bash
# Kernel traces of one tile share and of the whole frame of a workload.  Usage (GPU box): bash tools/share_trace.sh <tag> <workload> [N] [options...]
set -o pipefail
tag=$1; wl=$2; n=${3:-8}; shift 3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for part in share whole; do
  nn=$n; [ $part = whole ] && nn=1
  out=gpurun_out/${tag}_$part
  echo "# rocprofv3 --kernel-trace --stats -- python3 tools/share_trace.py $wl $nn 0 10 $*" > $out.txt
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out -- python3 tools/share_trace.py $wl $nn 0 10 "$@" > $out.log 2>&1 || { tail -n 5 $out.log; exit 1; }
  grep "per frame" $out.log >> $out.txt
  db=$(find $out -name '*.db' | head -n 1)
  python3 tools/prof_summary.py "$db" | grep -v "^k_.* [0-9]* 0 [0-9]* [0-9]* [0-9]* [0-9]* [0-9]*$" >> $out.txt 2>&1
  rm -rf $out $out.log
  head -n 24 $out.txt | cut -c1-110
done
