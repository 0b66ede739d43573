#!/bin/bash
# Kernel trace of the default bench run.  Usage (GPU box): bash tools/trace_pass.sh <tag> <workload> [bench args]
set -o pipefail
tag=$1; wl=$2; shift 2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
# every frame's kernels in one launch each and into HBM, as rt355_render_device and the bench's kernel times have them
# (rt355_render by itself renders a one-kernel frame in six parts or straight into the caller's pixmap, and launches a
# searched frame's last stage in four parts, each read back while the next runs)
export RT355_TAIL_PARTS=1 RT355_READBACK_COPY=1 RT355_RENDER_PARTS=1
echo "# RT355_TAIL_PARTS=1 RT355_READBACK_COPY=1 RT355_RENDER_PARTS=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --workload $wl --no-others --no-scaling --no-allpairs --no-cpu-baseline $*" > gpurun_out/$tag.txt
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -- python3 bench.py --steps 3 --warmup 1 --workload $wl --no-others --no-scaling --no-allpairs --no-cpu-baseline "$@" > gpurun_out/$tag.log 2>&1 || { tail -n 5 gpurun_out/$tag.log; exit 1; }
db=$(find gpurun_out/$tag -name '*.db' | head -n 1)
python3 tools/prof_summary.py "$db" >> gpurun_out/$tag.txt 2>&1
[ -n "$KEEP_RAW" ] || rm -rf gpurun_out/$tag gpurun_out/$tag.log  # (the raw trace is tens of megabytes)
head -n 20 gpurun_out/$tag.txt
