#!/bin/bash
# usage: tools/pmc_layer.sh <tag> <kernel substring> <one_layer args...>   (GPU box; prints mean counter values of the last launches)
tag=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU"; do
  rm -rf gpurun_out/_pmc_$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/_pmc_$tag -o t -- python3 tools/one_layer.py "$@" > gpurun_out/_pmc_$tag.log 2>&1
  python3 - "$tag" "$kern" <<'PY'
import csv,glob,collections,sys
tag,kern=sys.argv[1],sys.argv[2]
f=glob.glob(f"gpurun_out/_pmc_{tag}/**/*counter_collection.csv",recursive=True)
if not f: print("no counters", open(f"gpurun_out/_pmc_{tag}.log").read()[-800:]); sys.exit()
rows=[r for r in csv.DictReader(open(f[0])) if kern in r["Kernel_Name"]]
d=collections.defaultdict(list)
for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(tag, {k: round(sum(v[-5:])/len(v[-5:])) for k,v in d.items()})
t=glob.glob(f"gpurun_out/_pmc_{tag}/**/*kernel_trace.csv",recursive=True)
if t:
    us=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(t[0])) if kern in r["Kernel_Name"]]
    print(tag, "us", [round(x,1) for x in us[-4:]], sorted({r["Kernel_Name"][:70] for r in csv.DictReader(open(t[0])) if kern in r["Kernel_Name"]}))
PY
done
rm -rf gpurun_out/_pmc_$tag gpurun_out/_pmc_$tag.log
