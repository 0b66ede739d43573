#!/bin/bash
# GPU box: tools/frame_times.py under several values of one environment variable.
# usage: tools/env_sweep.sh VAR "<workloads>" value...
var=$1; wls=$2; shift 2
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
    echo "== $var=$v"
    env "$var=$v" timeout -k 10 200 python3 tools/frame_times.py $wls 2>&1 | grep -v amdgpu.ids || exit 1
done
