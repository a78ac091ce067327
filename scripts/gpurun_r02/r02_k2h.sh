#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|gpu:0).*|Name\s*:\s*\S+" | grep -iE "TCP_|TCC_HIT|TCC_MISS|TCC_REQ|TA_|SQ_INSTS_VMEM|SQ_WAIT|SQ_ACTIVE_INST_VMEM|TCC_EA0_RDREQ|TCC_READ" | sort -u | head -80 > $GRAFT_REPO_ROOT/gpurun_out/r02_counters_list.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/r02_counters_list.txt
ROOT=$GRAFT_REPO_ROOT
run() { tag=$1; shift; out=/tmp/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 "$@" ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); grep "rowjoin_dense" $ROOT/gpurun_out/$tag.md | cut -c1-170; else echo "no db for $tag"; tail -5 $out/run.log; fi; }
run r02_k2_pmc_a --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace -d /tmp/prof_r02_k2_pmc_a -o x -- python scripts/quick_k2c.py
run r02_k2_pmc_b --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d /tmp/prof_r02_k2_pmc_b -o x -- python scripts/quick_k2c.py
run r02_k2_pmc_c --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --kernel-trace -d /tmp/prof_r02_k2_pmc_c -o x -- python scripts/quick_k2c.py
