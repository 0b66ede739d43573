#!/bin/bash
# Every committed profile of the bench workloads in one go (GPU box): kernel traces, the HBM traffic passes, the SQ
# passes of the dominant kernels and the all-pairs search.  Results land in gpurun_out/ named <round>_*; copy the
# .txt summaries and the traffic file into profiles/.  Usage: bash tools/refresh_profiles.sh r3 [traces|counters]
# (the two halves fit one GPU call each)
set -o pipefail
R=${1:-r3}
HALF=${2:-all}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
if [ "$HALF" != "counters" ]; then
for p in "trace_default default_4k" "trace_spheres10k spheres10k_8lights_4k" "trace_mesh100k mesh100k_2lights_4k" \
         "trace_bleed162 default_bleed162_4k" "trace_mirrors10k spheres10k_mirrors_8lights_4k" \
         "trace_glass glass_room_4k" "trace_default_mirrors default_mirrors_4k" "trace_mesh1m_8k mesh1m_2lights_8k" \
         "trace_mesh100k_bleed42 mesh100k_bleed42_4k" "trace_glass10k spheres10k_glass_8lights_4k"; do
  set -- $p
  bash tools/trace_pass.sh ${R}_$1 $2 > /dev/null || exit 1
  echo "done $1" >> gpurun_out/refresh_progress.txt
done
# the all-pairs search on the north-star scene: trace, then the counters its roofline is recomputed from
bash tools/trace_pass.sh ${R}_trace_mesh100k_allpairs mesh100k_2lights_4k --mode allpairs > /dev/null || exit 1
bash tools/pmc_pass.sh ${R}_pmc_mesh100k_allpairs_sq mesh100k_2lights_4k --mode allpairs -- SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU > /dev/null || exit 1
echo "done allpairs" >> gpurun_out/refresh_progress.txt
fi
[ "$HALF" = "traces" ] && exit 0
bash tools/traffic_passes.sh default_4k spheres10k_8lights_4k mesh100k_2lights_4k default_bleed162_4k glass_room_4k default_mirrors_4k spheres10k_mirrors_8lights_4k mesh100k_bleed42_4k spheres10k_glass_8lights_4k > /dev/null || exit 1
cp gpurun_out/traffic.json gpurun_out/${R}_traffic.json
echo "done traffic" >> gpurun_out/refresh_progress.txt
for p in "pmc_spheres10k_sq spheres10k_8lights_4k" "pmc_mesh100k_sq mesh100k_2lights_4k" "pmc_default_sq default_4k" \
         "pmc_glass_sq glass_room_4k" "pmc_bleed162_sq default_bleed162_4k" "pmc_mesh100k_bleed42_sq mesh100k_bleed42_4k" \
         "pmc_mirrors10k_sq spheres10k_mirrors_8lights_4k" "pmc_glass10k_sq spheres10k_glass_8lights_4k"; do
  set -- $p
  bash tools/pmc_pass.sh ${R}_$1 $2 -- SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES > /dev/null || exit 1
  echo "done $1" >> gpurun_out/refresh_progress.txt
done
for f in gpurun_out/traffic_*_SIZE.txt; do cp "$f" "gpurun_out/${R}_$(basename "$f")"; done
# the share of the vector-issue slots every kernel uses, for bench.py's "valu_issue" figures
python3 tools/issue_merge.py gpurun_out/${R}_pmc_*_sq.txt > gpurun_out/${R}_issue.json || exit 1
exit 0
