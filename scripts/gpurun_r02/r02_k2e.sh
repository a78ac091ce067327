#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" poly_amd/libpolyhip_nc11.so poly_amd/libpolyhip_nc10.so poly_amd/libpolyhip_nc9.so; do POLYHIP_LIB=$v python scripts/quick_k2d.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02_k2_nc.log
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do out=/tmp/p_$c; rm -rf $out; mkdir -p $out
( cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $c --kernel-trace -d $out -o x -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra ) > $out/run.log 2>&1
f=$(find $out -name "*results.db" | head -1); ( cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py $f r02_bench_$c > gpurun_out/r02_bench_${c}.md ); grep "slab_kernel\|general" $GRAFT_REPO_ROOT/gpurun_out/r02_bench_${c}.md | cut -c1-160; done
