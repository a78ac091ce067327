#!/bin/bash
R=r03
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # tag, rocprof args..., -- cmd
  tag=$1; shift
  out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 "$@" ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db for $tag"; tail -5 $out/run.log; fi
  rm -rf $out
}
run ${R}_fasta_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_fasta_stats -o x -- python scripts/quick_fasta.py
grep polyhip $ROOT/gpurun_out/${R}_fasta_stats.md | head -14 | cut -c1-150
