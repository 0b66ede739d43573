#!/bin/bash
# Kernel trace and counter passes of the default (packet) bench run; summaries land in gpurun_out/.
# Usage (on the GPU box): bash tools/profile_packet.sh <tag> [workload]
set -o pipefail
tag=$1; wl=${2:-spheres10k_8lights_4k}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
args="bench.py --steps 3 --warmup 1 --workload $wl --no-allpairs --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -- python3 $args > gpurun_out/${tag}_trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d gpurun_out/${tag}_pmc_sq -- python3 $args > gpurun_out/${tag}_pmc_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE TA_BUSY_avr -d gpurun_out/${tag}_pmc_tcp -- python3 $args > gpurun_out/${tag}_pmc_tcp.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -- python3 $args > gpurun_out/${tag}_pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -- python3 $args > gpurun_out/${tag}_pmc_write.log 2>&1 || exit 1
for p in trace pmc_sq pmc_tcp pmc_fetch pmc_write; do
  db=$(find gpurun_out/${tag}_$p -name '*.db' | head -n 1)
  python3 tools/prof_summary.py "$db" > gpurun_out/${tag}_$p.txt 2>&1
done
echo profiled
