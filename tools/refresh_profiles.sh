#!/bin/bash
# Every committed profile of the bench workloads in one go (GPU box): kernel traces, the HBM traffic passes and
# the SQ passes of the unsearched kernels.  Usage: bash tools/refresh_profiles.sh
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for p in "trace_default default_4k" "trace_c3 spheres10k_8lights_4k" "trace_mesh mesh100k_2lights_4k" \
         "trace_bleed default_bleed162_4k" "trace_mirrors spheres10k_mirrors_8lights_4k" \
         "trace_glass glass_room_4k" "trace_default_mirrors default_mirrors_4k"; do
  set -- $p
  bash tools/trace_pass.sh $1 $2 > /dev/null || exit 1
  echo "done $1" >> gpurun_out/refresh_progress.txt
done
bash tools/traffic_passes.sh default_4k spheres10k_8lights_4k mesh100k_2lights_4k default_bleed162_4k glass_room_4k default_mirrors_4k > /dev/null || exit 1
echo "done traffic" >> gpurun_out/refresh_progress.txt
for p in "pmc_glass_sq glass_room_4k" "pmc_default_mirrors_sq default_mirrors_4k" "pmc_default_sq default_4k"; do
  set -- $p
  bash tools/pmc_pass.sh $1 $2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES > /dev/null
  echo "done $1" >> gpurun_out/refresh_progress.txt
done
exit 0
