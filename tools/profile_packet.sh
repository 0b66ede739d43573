#!/bin/bash
# Kernel trace and counter passes of the default bench run (packet mode); summaries land in gpurun_out/.
# Usage (on the GPU box): bash tools/profile_packet.sh <tag> [workload]
set -o pipefail
tag=$1; wl=${2:-spheres10k_8lights_4k}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
args="bench.py --steps 5 --warmup 3 --workload $wl --no-allpairs --no-cpu-baseline"
run() { # <suffix> <rocprofv3 options...>
  local sfx=$1; shift
  timeout -s KILL 120 rocprofv3 "$@" -d gpurun_out/${tag}_$sfx -- python3 $args > gpurun_out/${tag}_$sfx.log 2>&1 || { tail -n 5 gpurun_out/${tag}_$sfx.log; return 1; }
  local db=$(find gpurun_out/${tag}_$sfx -name '*.db' | head -n 1)
  { echo "# rocprofv3 $* -- python3 $args"; grep -o '{"metric.*' gpurun_out/${tag}_$sfx.log | cut -c1-400; python3 tools/prof_summary.py "$db"; } > gpurun_out/${tag}_$sfx.txt 2>&1
}
run trace --kernel-trace --stats || exit 1
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
run pmc_mem --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
echo profiled
