#!/bin/bash
# HBM-side traffic of every kernel of the bench workloads: one FETCH_SIZE pass and one WRITE_SIZE pass each
# (the two do not fit one pass), merged into gpurun_out/traffic.json by tools/traffic_merge.py.
# rt355_render is told to render every frame with one launch into the framebuffer in HBM and copy it back, so that
# every launch of a kernel moves one frame's bytes through HBM (by itself it may write a page-locked pixmap directly).
# Usage (GPU box): bash tools/traffic_passes.sh <workload> [...]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export RT355_READBACK_COPY=1 RT355_RENDER_PARTS=1 RT355_TAIL_PARTS=1
for wl in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    tag=traffic_${wl}_$c
    timeout -s KILL 150 rocprofv3 --pmc $c -d gpurun_out/$tag -- python3 bench.py --steps 3 --warmup 1 --workload $wl --no-others --no-scaling --no-allpairs --no-cpu-baseline > gpurun_out/$tag.log 2>&1 || { tail -n 5 gpurun_out/$tag.log; exit 1; }
    db=$(find gpurun_out/$tag -name '*.db' | head -n 1)
    python3 tools/prof_summary.py "$db" > gpurun_out/$tag.txt 2>&1 || exit 1
    [ -n "$KEEP_RAW" ] || rm -rf gpurun_out/$tag gpurun_out/$tag.log
  done
done
python3 tools/traffic_merge.py "$@" > gpurun_out/traffic.json && cat gpurun_out/traffic.json | head -c 1500
