#!/bin/bash
# Frame times of prebuilt library variants (raytracer_amd/lib/variants/librt355_<name>.so) on several workloads (GPU box).
# usage: tools/variant_frames.sh "<workload> ..." <variant> [<variant> ...]
wls=$1; shift; mkdir -p gpurun_out/variants
for wl in $wls; do
  for v in "$@"; do
    RT355_LIB=$PWD/raytracer_amd/lib/variants/librt355_$v.so timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-allpairs --no-scaling --no-others --workload $wl > gpurun_out/variants/frames_${wl}_$v.json 2> gpurun_out/variants/frames_${wl}_$v.err || { echo "FAILED $wl $v"; tail -3 gpurun_out/variants/frames_${wl}_$v.err; exit 1; }
    python3 - "$wl" "$v" gpurun_out/variants/frames_${wl}_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print(f"{sys.argv[1]:32s} {sys.argv[2]:10s} ms/frame {d['ms_per_step']:8.3f}  in HBM {d.get('ms_per_step_device_resident') or 0:8.3f}", flush=True)
PY
  done
done
