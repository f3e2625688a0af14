#!/bin/bash
# GPU box: rocprofv3 kernel trace of the default bench -> steady-state per-kernel table only (no PMC passes).  usage: tools/kt_steady.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_kt
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/kt -o trace -- python3 bench.py --steps 6 --warmup 3 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 1 > $out/kt.log 2>&1
python3 tools/steady_profile.py $(find $out/kt -name "*kernel_trace.csv" | head -1) 4 $out/${tag}_kernel_steady.csv > $out/${tag}_kernel_steady.txt
rm -rf $out/kt
head -3 $out/${tag}_kernel_steady.txt
