#!/bin/bash
# GPU box: per-kernel times of the bench's 4K secondary leg (train leg kept minimal)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/kt_4k; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/kt -o trace -- python3 bench.py --mode infer --rows 2160 --cols 3840 --pad 72 --batch 1 --planes 8 --steps 3 --warmup 2 > $out/log.txt 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/kt_4k/kt/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tail=rows[len(rows)*2//5:]  # skip warm-up
agg=collections.defaultdict(lambda:[0,0.0])
for r in tail:
    a=agg[r["Kernel_Name"][:100]]; a[0]+=1; a[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
tot=sum(v[1] for v in agg.values())
print("kernel ms in the tail:", round(tot,1))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:14]: print("%-100s %5d %8.2f ms"%(k,v[0],v[1]))
PY
tail -1 $out/log.txt | cut -c1-300
rm -rf $out/kt
