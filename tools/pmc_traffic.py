"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `python3 bench.py ...` into HBM bytes per launch of
the gather-GEMM and weight-gradient kernel families, over the steady-state tail (last `frac` of each family's launches, which
skips the warm-up and autotune launches).
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [frac=0.4] [git_sha]
The output is stamped with the git SHA it was made at (passed in: the GPU box has no .git) and with the hash of the GEMM kernel
sources, which bench.py compares with the build it runs (roofline.traffic_source.matches_this_build).
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM
section), hence the factor 2 on the read side."""
import csv, glob, hashlib, json, os, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha16():
    """= bench.kernel_source_sha16(): every .inc / wg6_*.h of csrc/ plus conv_engine.hip, wgrad6.hip and common.h."""
    d = os.path.join(REPO, "learned_hologram_gan_amd", "csrc")
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(d, "*.inc")) + glob.glob(os.path.join(d, "wg6_*.h")) + [os.path.join(d, "conv_engine.hip"), os.path.join(d, "wgrad6.hip"), os.path.join(d, "common.h")]):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def family(name):
    """gg* kernels (gg, gg2, gg2b, gg3s, gg4s ...) -> "gg", wg* kernels (wg, wg2, wg2s, wg3b, wg4s, wg5p ...) -> "wg"; mangled or demangled names."""
    import re

    head = name.split("(")[0]
    if "wgrad_reduce" in head or "pack_weight" in head or "thin_" in head:
        return None
    m = re.search(r"(?<![a-z_])(gg|wg)[0-9]*[a-z]{0,2}_kernel", head)
    return m.group(1) if m else None


def tail_mean(d, counter, frac):
    rows = {"gg": [], "wg": []}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam and r["Counter_Name"] == counter:
                rows[fam].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = {}
    for fam, v in rows.items():
        v.sort()
        v = v[int(len(v) * (1 - frac)):]
        out[fam] = (len(v), sum(x for _, x in v) / max(len(v), 1))
    return out


fetch_dir, write_dir, out_path = sys.argv[1:4]
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.4
git_sha = sys.argv[5] if len(sys.argv) > 5 else os.environ.get("LHG_GIT_SHA")
fe, wr = tail_mean(fetch_dir, "FETCH_SIZE", frac), tail_mean(write_dir, "WRITE_SIZE", frac)
res = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 3 --cpu-baseline 0",
       "note": "steady-state tail of the run; FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced "
               "stream (MI355X_MICROARCH.md, HBM), hence the factor 2 on the read side",
       "git_sha": git_sha, "kernel_src_sha16": kernel_source_sha16(), "kernels": {}}
for fam in ("gg", "wg"):
    res["kernels"][fam] = {"launches_sampled": fe[fam][0], "write_launches_sampled": wr[fam][0], "fetch_kib_per_launch": round(fe[fam][1], 1),
                           "write_kib_per_launch": round(wr[fam][1], 1),
                           "hbm_bytes_per_launch_corrected": int((2 * fe[fam][1] + wr[fam][1]) * 1024)}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res["kernels"]))
