#!/bin/bash
# usage: tools/pmc_stalls.sh <tag> <kernel substring> <one_layer args...>  — where the waves of one layer's kernel spend their cycles
# SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked at s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stalls) + SQ_ACTIVE_INST_ANY (quad-cycle units, MI355X_MICROARCH.md)
tag=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_pmcs_$tag
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/_pmcs_$tag -o t -- python3 tools/one_layer.py "$@" > gpurun_out/_pmcs_$tag.log 2>&1
python3 - "$tag" "$kern" <<'PY'
import csv,glob,collections,sys,json
tag,kern=sys.argv[1],sys.argv[2]
f=glob.glob(f"gpurun_out/_pmcs_{tag}/**/*counter_collection.csv",recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if kern in r["Kernel_Name"]]
d=collections.defaultdict(list)
for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
t=glob.glob(f"gpurun_out/_pmcs_{tag}/**/*kernel_trace.csv",recursive=True)
us=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(t[0])) if kern in r["Kernel_Name"]]
a={k: sum(v[-4:])/len(v[-4:]) for k,v in d.items()}
w=a.get("SQ_WAVE_CYCLES",1)
out={"tag":tag,"kernel":kern,"us":round(sum(us[-3:])/3,1)}
for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS"): out[k+"/WAVE_CYCLES"]=round(a.get(k,0)/w,3)
out["raw"]={k:round(v) for k,v in a.items()}
print(json.dumps(out))
PY
rm -rf gpurun_out/_pmcs_$tag gpurun_out/_pmcs_$tag.log
