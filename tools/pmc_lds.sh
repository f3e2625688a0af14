#!/bin/bash
# usage: tools/pmc_lds.sh <tag> <kernel substring> <one_layer args...>  — LDS counters + duration of one layer
tag=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_pmc_$tag
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/_pmc_$tag -o t -- python3 tools/one_layer.py "$@" > gpurun_out/_pmc_$tag.log 2>&1
python3 - "$tag" "$kern" <<'PY'
import csv,glob,collections,sys
tag,kern=sys.argv[1],sys.argv[2]
f=glob.glob(f"gpurun_out/_pmc_{tag}/**/*counter_collection.csv",recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if kern in r["Kernel_Name"]]
d=collections.defaultdict(list)
for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
t=glob.glob(f"gpurun_out/_pmc_{tag}/**/*kernel_trace.csv",recursive=True)
us=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(t[0])) if kern in r["Kernel_Name"]]
print(tag, {k: round(sum(v[-4:])/len(v[-4:])) for k,v in d.items()}, "us", [round(x,1) for x in us[-3:]])
PY
rm -rf gpurun_out/_pmc_$tag gpurun_out/_pmc_$tag.log
