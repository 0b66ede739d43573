#!/usr/bin/env python
"""Merge the SQ counter summaries written by tools/pmc_pass.sh into one JSON object:
{workload: {kernel: {"valu_inst_per_launch": wave-level vector instructions per launch (SQ_ACTIVE_INST_VALU),
"avg_ms": the launch's average duration in the counter pass, "launches": n}}}.  Template instances of a kernel are
merged under its name.  bench.py turns an entry into the share of the chip's vector-issue slots a launch uses
(instructions x 4 cycles over 1 024 SIMDs x clock x the duration it measures itself).
Usage: python tools/issue_merge.py gpurun_out/r3_pmc_*_sq.txt > gpurun_out/r3_issue.json"""
import json
import re
import sys


def main(paths):
    res = {}
    for path in paths:
        lines = open(path).read().splitlines()
        m = re.search(r"--workload (\S+)", lines[0])
        if not m or "--mode allpairs" in lines[0]:
            continue
        wl = res.setdefault(m.group(1), {})
        inst, ms = {}, {}
        for line in lines:
            t = re.match(r"^(?:void )?(k_\w+)(?:<[^>]*>?)?\s+(\d+)\s+(\d+\.\d+)\s+(\d+\.\d+)\s+(\d+\.\d+)\s+(\d+\.\d+)\s+(\d+\.\d+)\s*$", line)
            if t:  # the time table: calls, total ms
                cur = ms.setdefault(t.group(1), [0.0, 0])
                cur[0] += float(t.group(3)); cur[1] += int(t.group(2))
            c = re.match(r"^(?:void )?(k_\w+)(?:<[^>]*>?)?\s+SQ_ACTIVE_INST_VALU\s+(\d+)\s+(\d+)\s+(\d+)\s*$", line)
            if c:
                cur = inst.setdefault(c.group(1), [0, 0])
                cur[0] += int(c.group(2)); cur[1] += int(c.group(3))
        for k in sorted(inst):
            if k in ms and ms[k][1] and inst[k][1]:
                wl[k] = {"valu_inst_per_launch": inst[k][0] / inst[k][1], "avg_ms": round(ms[k][0] / ms[k][1], 4), "launches": inst[k][1]}
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1:])
