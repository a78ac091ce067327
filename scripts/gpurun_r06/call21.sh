ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "400000 250 5000" "160000 500 5000" "80000 1000 5000"; do
  tag=$(echo $cfg | tr ' ' '_'); out=$ROOT/gpurun_out/prof_len_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 --kernel-trace --stats -d $out -o x -- python scripts/quick_sw_len.py $cfg ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  echo "== $cfg"; ( cd $ROOT && python scripts/rocpd_summary.py $f len_$tag | grep polyhip | head -6 | cut -c1-130 )
  rm -rf $out
done
