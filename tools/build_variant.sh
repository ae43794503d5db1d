#!/bin/bash
# Build a variant of the library for a same-box A/B (tools/ab_bench.sh, RAG_AMD_LIB): one or more sources (comma-separated) recompiled
# with extra flags, the rest of the objects as built by `make`.
#   bash tools/build_variant.sh <name> <source.hip[,source2.hip...]> "<extra flags>"   ->  rag_amd/lib/librag_amd_<name>.so
set -e
name=$1; srcs=$2; flags=$3
objs=""; skip=""
for src in ${srcs//,/ }; do
  obj=/tmp/ragmi_variant_${name}_$(basename "${src%.hip}").o
  extra=""
  case "$src" in *disp.hip|*conv3d_c1.hip) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Iinclude -Irag_amd/csrc -Wall -Wno-unused-function $extra $flags -c "$src" -o "$obj"
  objs="$objs $obj"; skip="$skip|$(basename "${src%.hip}").o"
done
others=$(ls rag_amd/csrc/*.o | grep -v -E "/(${skip#|})$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o rag_amd/lib/librag_amd_${name}.so $others $objs
echo "built rag_amd/lib/librag_amd_${name}.so"
