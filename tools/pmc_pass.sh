#!/bin/bash
# One counter pass of the default bench run.  Usage (GPU box): bash tools/pmc_pass.sh <tag> <workload> COUNTER...
# (RT355_LIB in the environment selects a library variant.)
set -o pipefail
tag=$1; wl=$2; shift 2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" -d gpurun_out/$tag -- python3 bench.py --steps 3 --warmup 1 --workload $wl --no-others --no-allpairs --no-cpu-baseline > gpurun_out/$tag.log 2>&1 || { tail -n 5 gpurun_out/$tag.log; exit 1; }
db=$(find gpurun_out/$tag -name '*.db' | head -n 1)
python3 tools/prof_summary.py "$db" > gpurun_out/$tag.txt 2>&1
grep -E "k_shade_group|k_resolve_primary|k_search_packet|k_bin_scatter" gpurun_out/$tag.txt
