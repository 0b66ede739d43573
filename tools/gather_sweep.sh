#!/bin/bash
# GPU box: frame time of the searched gather workload over pass layouts ("grid,per_cell,theta_limit").
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "$@"; do
    echo "== RT355_GATHER_PASSES=$cfg"
    RT355_GATHER_PASSES=$cfg timeout -k 10 120 python3 tools/frame_times.py mesh100k_bleed42_4k || exit 1
done
