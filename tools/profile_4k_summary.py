"""Summarise tools/profile_4k.sh: per-kernel time of the last `frames` frames of the 4K inference leg and the HBM traffic per frame of its
kernel families (FETCH_SIZE doubled: MI355X_MICROARCH.md, HBM section).  Frames are delimited by lhg::double_phase_kernel (once per frame).
usage: python tools/profile_4k_summary.py <dir> <tag> <git sha> <frames>"""
import collections, csv, glob, json, os, re, sys

d, tag, sha, frames = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])


def family(n):
    if re.search(r"(?<![a-z_])gg[0-9]*[a-z]{0,2}_kernel", n): return "gather-GEMM"
    if re.search(r"cols_|rows_|asm|stockham|bluestein", n, re.I): return "angular-spectrum passes"
    if "thin_" in n: return "thin convolutions"
    if "bn_" in n: return "BatchNorm (eval affine)"
    if "absmax" in n: return "absmax"
    if "maxpool" in n: return "max-pool"
    if "pack_" in n: return "weight packing"
    if "at::native" in n or "rocclr" in n: return "ATen"
    return "other"


def tail(rows):
    rows.sort(key=lambda r: int(r["Start_Timestamp"]) if "Start_Timestamp" in r else int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "double_phase_kernel" in r["Kernel_Name"]]
    start = marks[-frames - 1] + 1 if len(marks) > frames else 0
    return rows[start:marks[-1] + 1] if marks else rows


f = glob.glob(os.path.join(d, "kt", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = tail(list(csv.DictReader(open(f))))
agg, fam = collections.defaultdict(lambda: [0, 0.0]), collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for table, key in ((agg, r["Kernel_Name"]), (fam, family(r["Kernel_Name"]))):
        table[key][0] += 1
        table[key][1] += us
tot = sum(v[1] for v in agg.values())
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
lines = [f"4K frame (3840x2160 bs=1 generator forward + 8-plane propagate), last {frames} frames: {len(rows) / frames:.0f} dispatches, kernel time {tot / frames / 1e3:.2f} ms/frame, wall span {span / frames / 1e3:.2f} ms/frame (git {sha})"]
for k, (n, us) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"  family {k:28s} {n / frames:6.1f}/frame {us / frames / 1e3:8.2f} ms/frame {100 * us / tot:5.1f}%")
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    lines.append(f"{k[:100]:100s} {n / frames:6.1f}/frame {us / frames:9.1f} us/frame avg {us / n:8.1f} us {100 * us / tot:5.1f}%")
open(os.path.join(d, f"{tag}_4k_kernel_steady.txt"), "w").write("\n".join(lines) + "\n")
with open(os.path.join(d, f"{tag}_4k_kernel_steady.csv"), "w") as out:
    w = csv.writer(out)
    w.writerow(["Name", "CallsPerFrame", "UsPerFrame", "AvgUs", "Percent"])
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, round(n / frames, 2), round(us / frames, 1), round(us / n, 1), round(100 * us / tot, 2)])

traffic = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True):
        rs = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
        for r in tail(rs):
            a = per[family(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for k, (n, kib) in per.items():
        traffic.setdefault(k, {})[counter] = {"launches_per_frame": n / frames, "kib_per_frame": kib / frames}
summary = {"git_sha": sha, "frames": frames, "families": {}}
for k, v in traffic.items():
    fe, wr = v.get("FETCH_SIZE", {}), v.get("WRITE_SIZE", {})
    hbm = (2 * fe.get("kib_per_frame", 0.0) + wr.get("kib_per_frame", 0.0)) * 1024
    summary["families"][k] = {"launches_per_frame": fe.get("launches_per_frame"), "fetch_kib_per_frame": fe.get("kib_per_frame"), "write_kib_per_frame": wr.get("kib_per_frame"),
                              "hbm_bytes_per_frame_corrected": hbm,
                              "hbm_bytes_per_launch_corrected": hbm / max(fe.get("launches_per_frame") or 1, 1)}
json.dump(summary, open(os.path.join(d, f"{tag}_4k_pmc_traffic.json"), "w"), indent=1)
