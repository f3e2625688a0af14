#!/bin/bash
# GPU box: per-launch table of one steady step (tools/step_timeline.py).  usage: tools/step_timeline.sh <out dir under gpurun_out>
out=gpurun_out/$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/kt -o trace -- python3 bench.py --steps 3 --warmup 3 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 1 > $out/kt.log 2>&1
python3 tools/step_timeline.py $(find $out/kt -name "*kernel_trace.csv" | head -1) $out/step_timeline.tsv
rm -rf $out/kt
wc -l $out/step_timeline.tsv
