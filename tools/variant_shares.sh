#!/bin/bash
# GPU box: tools/share_sweep.py once per library variant.  usage: tools/variant_shares.sh "<workload> <N>" <variant> ...
args=$1; shift
for v in "$@"; do
  echo "== $v"
  RT355_LIB=$PWD/raytracer_amd/lib/variants/librt355_$v.so timeout -k 10 300 python3 tools/share_sweep.py $args "" 2>&1 | grep -v amdgpu.ids || exit 1
done
