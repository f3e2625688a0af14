"""Summarise a rocprofv3 kernel trace over its steady-state tail (skips warm-up / autotune launches).
usage: python tools/steady_profile.py <kernel_trace.csv> <steps_in_tail> [out.csv]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady tail = after the last launch of the twiddle kernel / or simply the last fraction delimited by adam kernels
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
per_step = 2  # G and D
start = adam[-(steps * per_step) - 1] + 1 if len(adam) > steps * per_step else 0
tail = rows[start:]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in tail:
    k = r["Kernel_Name"]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3
print(f"steady tail: {len(tail)} dispatches over {steps} steps; kernel time {tot/steps/1e3:.2f} ms/step, wall span {span/steps/1e3:.2f} ms/step")
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
w = csv.writer(open(sys.argv[3], "w")) if len(sys.argv) > 3 else None
if w: w.writerow(["Name", "CallsPerStep", "UsPerStep", "AvgUs", "Percent"])
for k, (n, us) in out[:40]:
    print(f"{k[:96]:96s} {n/steps:7.1f}/step {us/steps:9.1f} us/step avg {us/n:8.1f} us {100*us/tot:5.1f}%")
if w:
    for k, (n, us) in out: w.writerow([k, round(n/steps, 2), round(us/steps, 1), round(us/n, 1), round(100*us/tot, 2)])
