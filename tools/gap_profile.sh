#!/bin/bash
# GPU box: idle time between consecutive kernels of each HIP queue over the steady steps of the bench (rocprofv3 kernel trace).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/gaps; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/kt -o trace -- python3 bench.py --steps 6 --warmup 3 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 0 > $out/kt.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/gaps/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region: the last 6 steps = the tail; take the last 60 % of dispatches
tail = rows[len(rows) * 2 // 5:]
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e6
print("tail: %d dispatches, span %.1f ms" % (len(tail), span))
byq = collections.defaultdict(list)
for r in tail:
    byq[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]))
for q, ks in byq.items():
    busy = sum(e - s for s, e, _ in ks) / 1e6
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 < g < 200000]
    hist = collections.Counter(min(g // 2000, 10) for g in small)
    pairs = collections.defaultdict(lambda: [0, 0])
    for i in range(len(ks) - 1):
        g = ks[i + 1][0] - ks[i][1]
        if 0 < g < 200000:
            a = pairs[(ks[i][2][:34], ks[i + 1][2][:34])]; a[0] += 1; a[1] += g
    for (a, b), (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:10]:
        print("   %-34s -> %-34s %4d gaps %7.2f ms  avg %5.1f us" % (a, b, n, t / 1e6, t / n / 1e3))
    print("queue %s: %d kernels, busy %.1f ms, gaps < 0.2 ms: %d totalling %.2f ms (median %.1f us); by 2-us bins: %s" % (
        q, len(ks), busy, len(small), sum(small) / 1e6, sorted(small)[len(small) // 2] / 1e3 if small else 0, dict(sorted(hist.items()))))
PY
rm -rf $out/kt
