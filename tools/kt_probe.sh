#!/bin/bash
# GPU box: steady per-kernel times of tools/mode_switch_probe.py <arg>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/kt_probe$1
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/kt -o trace -- python3 tools/mode_switch_probe.py $1 > $out/kt.log 2>&1
python3 tools/steady_profile.py $(find $out/kt -name "*kernel_trace.csv" | head -1) 3 $out/steady.csv > $out/steady.txt
rm -rf $out/kt
head -3 $out/steady.txt
