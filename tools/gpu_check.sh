#!/bin/bash
# The round's GPU checks in one call (GPU box): the full -m gpu suite, then a default bench run.
# Usage: bash tools/gpu_check.sh [extra pytest files...]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests "$@" -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -n 15 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
rc=$?
tail -c 600 gpurun_out/bench_default.err
head -c 1500 gpurun_out/bench_default.json
exit $rc
