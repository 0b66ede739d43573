#!/usr/bin/env python
"""Merge the FETCH_SIZE / WRITE_SIZE summaries written by tools/traffic_passes.sh into one JSON object:
{workload: {kernel: {"fetch_kb": per launch, "write_kb": per launch, "launches": n}}}.  rocprofv3 reports both
counters in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes, so a reader doubles it."""
import json
import re
import sys


def table(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"^(?:void )?(k_\w+)(?:<[^>]*>?)?\s+" + counter + r"\s+(\d+)\s+(\d+)\s+(\d+)\s*$", line)
        if m:
            name, total, n = m.group(1), int(m.group(2)), int(m.group(3))
            cur = out.setdefault(name, [0, 0])
            cur[0] += total; cur[1] += n
    return out


def main(workloads):
    res = {}
    for wl in workloads:
        f = table(f"gpurun_out/traffic_{wl}_FETCH_SIZE.txt", "FETCH_SIZE")
        w = table(f"gpurun_out/traffic_{wl}_WRITE_SIZE.txt", "WRITE_SIZE")
        res[wl] = {k: {"fetch_kb": f[k][0] / f[k][1], "write_kb": w.get(k, [0, 1])[0] / max(1, w.get(k, [0, 1])[1]),
                       "launches": f[k][1]} for k in sorted(f)}
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1:])
