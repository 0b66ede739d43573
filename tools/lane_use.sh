#!/bin/bash
# Lane utilisation of every kernel of some workloads (GPU box): SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU), the share
# of a wave's 64 lanes that take part in its average vector instruction.  usage: bash tools/lane_use.sh <workload> ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for wl in "$@"; do
  bash tools/pmc_pass.sh lanes_$wl $wl -- SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU > /dev/null || exit 1
  python3 - gpurun_out/lanes_$wl.txt $wl <<'PY'
import re, sys
ins, thr, ms = {}, {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"^(?:void )?(k_\w+)(?:<[^>]*>?)?\s+(SQ_\w+)\s+(\d+)\s+(\d+)\s+(\d+)\s*$", line)
    if m:
        d = ins if m.group(2) == "SQ_INSTS_VALU" else thr
        d[m.group(1)] = d.get(m.group(1), 0) + int(m.group(3))
        continue
    m = re.match(r"^(?:void )?(k_\w+)(?:<[^>]*>?)?\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
    if m:
        ms[m.group(1)] = ms.get(m.group(1), 0.0) + float(m.group(3))
print(sys.argv[2])
for k in sorted(ins, key=lambda k: -ms.get(k, 0)):
    if ins[k] and ms.get(k, 0) > 0.05:
        print(f"  {k:22s} {ms.get(k, 0):9.3f} ms total   lanes in use {thr.get(k, 0) / (64.0 * ins[k]):.3f}", flush=True)
PY
done
