#!/bin/bash
# GPU box: tools/frame_times.py once per library variant.  usage: tools/variant_times.sh "<workloads>" <variant> ...
wls=$1; shift
for v in "$@"; do
  echo "== $v"
  RT355_LIB=$PWD/raytracer_amd/lib/variants/librt355_$v.so timeout -k 10 300 python3 tools/frame_times.py $wls 2>&1 | grep -v amdgpu.ids || exit 1
done
