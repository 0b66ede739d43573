#!/bin/bash
# Run bench.py once per prebuilt library variant under gpurun_out/variants/ (A/B of kernel parameters).
# usage: tools/bench_variants.sh <workload> <steps> <variant> [<variant> ...]
wl=$1; steps=$2; shift 2; mkdir -p gpurun_out/variants
for v in "$@"; do
  RT355_LIB=$PWD/raytracer_amd/lib/variants/librt355_$v.so timeout -k 10 200 python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-allpairs --workload $wl $BENCH_EXTRA > gpurun_out/variants/bench_${wl}_$v.json 2> gpurun_out/variants/bench_${wl}_$v.err || { echo "FAILED $v"; tail -3 gpurun_out/variants/bench_${wl}_$v.err; exit 1; }
  python - "$v" gpurun_out/variants/bench_${wl}_$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
k=d["kernel_time"]
print(f"{sys.argv[1]:10s} ms/frame {d['ms_per_step']:9.3f}  Mrays/s {d['value']:9.1f}  search {k['search_ms_per_frame']:7.4f}  primary {k['resolve_primary_ms_per_frame']:7.4f}  shade {k['shade_ms_per_frame']:7.4f}")
PY
done
