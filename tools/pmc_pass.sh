#!/bin/bash
# One counter pass of a bench run.  Usage (GPU box): bash tools/pmc_pass.sh <tag> <workload> [bench args] -- COUNTER...
# (RT355_LIB in the environment selects a library variant.)
set -o pipefail
tag=$1; wl=$2; shift 2
extra=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do extra+=("$1"); shift; done
[ "$1" = "--" ] && shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
# every frame's kernels in one launch each and into HBM, as rt355_render_device and the bench's kernel times have them
# (rt355_render by itself renders a one-kernel frame in six parts or straight into the caller's pixmap, and launches a
# searched frame's last stage in four parts, each read back while the next runs)
export RT355_TAIL_PARTS=1 RT355_READBACK_COPY=1 RT355_RENDER_PARTS=1
echo "# RT355_TAIL_PARTS=1 RT355_READBACK_COPY=1 RT355_RENDER_PARTS=1 rocprofv3 --pmc $* -- python3 bench.py --steps 3 --warmup 1 --workload $wl --no-others --no-scaling --no-allpairs --no-cpu-baseline ${extra[*]}" > gpurun_out/$tag.txt
timeout -s KILL 120 rocprofv3 --pmc "$@" -d gpurun_out/$tag -- python3 bench.py --steps 3 --warmup 1 --workload $wl --no-others --no-scaling --no-allpairs --no-cpu-baseline "${extra[@]}" > gpurun_out/$tag.log 2>&1 || { tail -n 5 gpurun_out/$tag.log; exit 1; }
db=$(find gpurun_out/$tag -name '*.db' | head -n 1)
python3 tools/prof_summary.py "$db" >> gpurun_out/$tag.txt 2>&1
[ -n "$KEEP_RAW" ] || rm -rf gpurun_out/$tag gpurun_out/$tag.log
grep -E "k_shade_group|k_resolve_primary|k_search|k_render|k_gather" gpurun_out/$tag.txt | head -n 40
