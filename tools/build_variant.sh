#!/bin/bash
# Build a variant of the library for a same-box A/B (tools/ab_bench.sh, RAG_AMD_LIB): one source recompiled with extra flags, the rest
# of the objects as built by `make`.
#   bash tools/build_variant.sh <name> <source.hip> "<extra flags>"   ->  rag_amd/lib/librag_amd_<name>.so
set -e
name=$1; src=$2; flags=$3
obj=/tmp/ragmi_variant_${name}.o
extra=""
case "$src" in *disp.hip) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Iinclude -Irag_amd/csrc -Wall -Wno-unused-function $extra $flags -c "$src" -o "$obj"
others=$(ls rag_amd/csrc/*.o | grep -v "$(basename "${src%.hip}").o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o rag_amd/lib/librag_amd_${name}.so $others "$obj"
echo "built rag_amd/lib/librag_amd_${name}.so"
