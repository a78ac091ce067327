#!/bin/bash
# usage: prof_summary.sh <tag> -- <command...> : rocprofv3 kernel stats for a command, summary in gpurun_out/<tag>.md
tag=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d $out -o $tag -- "$@" ) > $out/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*results.db" | head -1)
if [ -n "$f" ]; then python scripts/rocpd_summary.py $f > gpurun_out/$tag.md 2>&1; else ls -R $out | head -30; tail -20 $out/run.log; fi
tail -3 $out/run.log
cat gpurun_out/$tag.md 2>/dev/null | head -30
