#!/bin/bash
# builds scripts/ubench/k1_v_<tag> = scripts/ubench/k1_ablate.hip with the given -D flags (8 jobs at a time)
#   scripts/ubench/build_k1_variants.sh "base:" "win0:-DPH_BS_WIN=0" ...
cd "$(dirname "$0")/../.."
build() {
    tag=${1%%:*}; flags=${1#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -pragma-unroll-threshold=100000 -DPH_ABLATION_BUILD -w $flags \
        -I include -I poly_amd/csrc scripts/ubench/k1_ablate.hip poly_amd/csrc/runtime.hip poly_amd/csrc/multi_device.hip \
        -o scripts/ubench/k1_v_$tag || echo "BUILD FAILED $tag"
}
for v in "$@"; do
    build "$v" &
    while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 1; done
done
wait
ls scripts/ubench/k1_v_* | wc -l
