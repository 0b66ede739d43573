#!/bin/bash
# The N > 1 path of bench.py on a one-GPU box: every rank on device 0, collectives over gloo (RCCL refuses two ranks
# on one device).  Writes gpurun_out/multirank_rehearsal.txt: the command lines and the JSON line of each run.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export RT355_BENCH_SHARE_GPU=1
out=gpurun_out/multirank_rehearsal.txt
: > $out
run() {
  echo "# RT355_BENCH_SHARE_GPU=1 python3 bench.py $*  (all ranks on one GPU, gloo)" >> $out
  timeout -k 10 280 python3 bench.py "$@" --steps 5 --warmup 1 2> gpurun_out/rehearsal.err | cut -c 1-2600 >> $out || { tail -n 5 gpurun_out/rehearsal.err; exit 1; }
}
run --gpus 2 --shard slabs --workload mesh100k_2lights_4k || exit 1
run --gpus 2 --shard tiles --workload mesh100k_2lights_4k || exit 1
run --gpus 2 --shard balanced --workload mesh100k_2lights_4k || exit 1
run --gpus 2 --workload mesh100k_sweep60 || exit 1
run --gpus 2 || exit 1
run --gpus 4 --workload mesh1m_8k_tiles || exit 1
run --gpus 4 --inproc --workload mesh1m_8k_tiles || exit 1
cat $out | cut -c 1-260
