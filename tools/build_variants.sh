#!/bin/bash
# Build library variants for A/B runs: tools/build_variants.sh name:"-DFOO=1 -DBAR" ...  -> raytracer_amd/lib/variants/librt355_<name>.so
# (RT355_LIB=<that file> selects one; tools/variant_times.sh times them on the GPU box.)
mkdir -p raytracer_amd/lib/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  [ "$flags" = "$spec" ] && flags=""
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared $flags -o raytracer_amd/lib/variants/librt355_$name.so raytracer_amd/csrc/rt355.hip && echo "built $name" ) &
done
wait
