"""HBM bytes per launch of the conv kernels from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs).
usage: python tools/layer_traffic.py <dir with *_counter_collection.csv> [...]   (prints bytes per launch per kernel family)
FETCH_SIZE / WRITE_SIZE are KiB; gfx950 FETCH_SIZE counts half of a wide coalesced read stream (MI355X_MICROARCH.md), hence x2."""
import collections, csv, glob, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            fam = "gg" if "gg" in k.split("(")[0] and "kernel" in k else "wg" if "wg" in k.split("(")[0] and "kernel" in k and "reduce" not in k else None
            if fam is None:
                continue
            a = agg[fam][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
for fam, c in agg.items():
    fetch = c["FETCH_SIZE"][1] / max(c["FETCH_SIZE"][0], 1) * 1024 * 2
    write = c["WRITE_SIZE"][1] / max(c["WRITE_SIZE"][0], 1) * 1024
    print(f"{fam}: launches {c['FETCH_SIZE'][0]} fetch {fetch/1e6:.1f} MB (x2 corrected) write {write/1e6:.1f} MB total {(fetch+write)/1e6:.1f} MB per launch")
