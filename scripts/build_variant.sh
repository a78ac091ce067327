#!/bin/bash
# usage: scripts/build_variant.sh <tag> <file.hip> <-D flags...>: poly_amd/libpolyhip_<tag>.so = the library with ONE source
# rebuilt with extra flags (run with POLYHIP_LIB=poly_amd/libpolyhip_<tag>.so)
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift; shift
python -m poly_amd.build > /dev/null
obj=/tmp/variant_${tag}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -pragma-unroll-threshold=100000 -DPH_ABLATION_BUILD -w "$@" -I include -I poly_amd/csrc -c poly_amd/csrc/$src -o $obj
objs=$(ls poly_amd/csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o poly_amd/libpolyhip_${tag}.so $objs $obj -ldl
echo poly_amd/libpolyhip_${tag}.so
