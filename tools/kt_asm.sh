#!/bin/bash
# GPU box: per-kernel times of tools/bench_asm.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/kt_asm; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o trace -- python3 tools/bench_asm.py > $out/log.txt 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/kt_asm/kt/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
rm -rf $out/kt
