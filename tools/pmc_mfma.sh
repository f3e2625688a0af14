#!/bin/bash
# GPU box: matrix-pipe occupancy of one layer's kernel.  usage: tools/pmc_mfma.sh <tag> <kernel substring> <one_layer args...>
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs (32 per 32x32x16 MFMA: checked against the launch's MFMA count);
# GRBM_GUI_ACTIVE the kernel's cycles summed over the 8 XCDs.
tag=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_pmcm_$tag
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/_pmcm_$tag -o t -- python3 tools/one_layer.py "$@" > gpurun_out/_pmcm_$tag.log 2>&1
python3 - "$tag" "$kern" <<'PY'
import csv,glob,collections,sys,json
tag,kern=sys.argv[1],sys.argv[2]
f=glob.glob(f"gpurun_out/_pmcm_{tag}/**/*counter_collection.csv",recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if kern in r["Kernel_Name"] and "finish" not in r["Kernel_Name"]]  # (not the split-K finish kernel)
d=collections.defaultdict(list)
for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
t=glob.glob(f"gpurun_out/_pmcm_{tag}/**/*kernel_trace.csv",recursive=True)
us=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(t[0])) if kern in r["Kernel_Name"] and "finish" not in r["Kernel_Name"]]
avg={k: sum(v[-4:])/len(v[-4:]) for k,v in d.items()}
out={"tag":tag,"kernel":kern,"args":sys.argv[3:],"counters":{k:round(v) for k,v in avg.items()},"us_under_pmc":[round(x,1) for x in us[-3:]]}
if avg.get("GRBM_GUI_ACTIVE"):
    xcd_cycles = avg["GRBM_GUI_ACTIVE"] / 8            # the counter is summed over the 8 XCDs
    out["shader_clock_ghz"] = round(xcd_cycles / (sum(us[-3:]) / len(us[-3:])) / 1e3, 2)
    out["matrix_pipe_busy"] = round(avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / xcd_cycles, 3)  # per SIMD (1024 of them), of the kernel's cycles
    out["mfma_instructions"] = round(avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 32)
print(json.dumps(out))
PY
rm -rf gpurun_out/_pmcm_$tag gpurun_out/_pmcm_$tag.log
