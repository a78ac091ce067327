#!/bin/bash
R=r02
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # tag, rocprof args..., -- cmd
  tag=$1; shift
  out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 "$@" ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db for $tag"; tail -5 $out/run.log; fi
}
run ${R}_k2_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k2_stats -o x -- python scripts/quick_k2b.py skipfull
grep -E "k2::|fillBuffer" $ROOT/gpurun_out/${R}_k2_stats.md | head -20
rm -rf $ROOT/gpurun_out/prof_${R}_*
cd $ROOT
timeout 600 python bench.py > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open('gpurun_out/r02b_bench.json'))
print('K1', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['extra'].items():
    if isinstance(v,dict): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('cell_updates_per_s','score_pass_ms','traceback_ms','pairs_per_s_counts','counts_ms','ms','windows_per_s','file_GBs','error')})
print({k:(v if not isinstance(v,dict) else v.get('value')) for k,v in d['cpu_baseline'].items() if k!='sample'})
P
