"""One steady train step out of a rocprofv3 kernel trace, launch by launch (the profiler serialises the streams, so a duration here is the
kernel alone on the chip).  usage: python tools/step_timeline.py <kernel_trace.csv> [out.tsv]   (last step = after the third-last adam launch)"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
tail = rows[adam[-3] + 1:adam[-1] + 1]
t0 = int(tail[0]["Start_Timestamp"])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
out.write("t_us\tdur_us\tgap_us\tqueue\tgrid\twg\tlds\tvgpr\tkernel\n")
prev_end = t0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("lhg::", "")
    grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = r.get("Workgroup_Size_X", "?")
    out.write(f"{(s - t0) / 1e3:.1f}\t{(e - s) / 1e3:.1f}\t{(s - prev_end) / 1e3:.1f}\t{r.get('Queue_Id', '?')}\t{grid}\t{wg}\t{r.get('LDS_Block_Size', '?')}\t{r.get('VGPR_Count', '?')}\t{name[:90]}\n")
    prev_end = max(prev_end, e)
