#!/bin/bash
# GPU box: steady-state per-kernel times of one bench configuration.  usage: tools/kt_quick.sh <name> <bench args...>
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/kt_$name
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/kt -o trace -- python3 bench.py --steps 6 --warmup 3 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 1 "$@" > $out/kt.log 2>&1
python3 tools/steady_profile.py $(find $out/kt -name "*kernel_trace.csv" | head -1) 4 $out/steady.csv > $out/steady.txt
t=$(find $out/kt -name "*kernel_trace.csv" | head -1)
head -1 $t > $out/bn_trace.csv
grep -h "bn_bwd_apply\|bn_bwd_partial\|bn_apply" $t | tail -250 >> $out/bn_trace.csv
rm -rf $out/kt
head -40 $out/steady.txt
