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
    h = hashlib.sha256()
    for name in ("conv_engine.hip", "gg2_kernel.inc", "gg2b_kernel.inc", "gg3s_kernel.inc", "gg4s_kernel.inc", "wg2_kernel.inc", "wg2b_kernel.inc", "wg2s_kernel.inc", "wg3b_kernel.inc", "wg4s_kernel.inc", "wg5p_kernel.inc",
                 "wg3_kernel.inc"):  # = bench.KERNEL_SOURCES
        with open(os.path.join(REPO, "learned_hologram_gan_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def family(name):
    head = name.split("(")[0]
    if "wgrad_reduce" in head or "pack_weight" in head:
        return None
    if "gg2_kernel" in head or "gg_kernel" in head or "gg3s_kernel" in head or "gg2b_kernel" in head:
        return "gg"
    if "wg3_kernel" in head or "wg2_kernel" in head or "wg_kernel" in head or "wg2s_kernel" in head or "wg2b_kernel" in head:
        return "wg"
    return None


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
    res["kernels"][fam] = {"launches_sampled": fe[fam][0], "fetch_kib_per_launch": round(fe[fam][1], 1),
                           "write_kib_per_launch": round(wr[fam][1], 1),
                           "hbm_bytes_per_launch_corrected": int((2 * fe[fam][1] + wr[fam][1]) * 1024)}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res["kernels"]))
