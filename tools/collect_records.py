"""Move the round's measurement records from gpurun_out/ (what travels back from the GPU box) into profiles/ (tracked), and stamp them:
profiles/<tag>_records_manifest.json lists every collected file with its sha256, the git revision it was collected at and the hash of the
kernel sources (bench.kernel_source_sha16) — the tests and scripts write the records, this script is the only hand in between.

    python tools/collect_records.py [tag, default r05]
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
SOURCES = [  # (path under gpurun_out/, name under profiles/)
    (f"{TAG}_truth_tests.jsonl", f"{TAG}_truth_tests.jsonl"),
    (f"{TAG}_pmc_mfma.jsonl", f"{TAG}_pmc_mfma.jsonl"),
    (f"{TAG}_prof/{TAG}_kernel_steady.txt", f"{TAG}_kernel_steady.txt"),
    (f"{TAG}_prof/{TAG}_kernel_steady.csv", f"{TAG}_kernel_steady.csv"),
    (f"{TAG}_prof/{TAG}_kernel_stats.csv", f"{TAG}_kernel_stats.csv"),
    (f"{TAG}_prof/{TAG}_pmc_traffic.json", f"{TAG}_pmc_traffic.json"),
    (f"{TAG}_prof4k/{TAG}_4k_kernel_steady.txt", f"{TAG}_4k_kernel_steady.txt"),
    (f"{TAG}_prof4k/{TAG}_4k_kernel_steady.csv", f"{TAG}_4k_kernel_steady.csv"),
    (f"{TAG}_prof4k/{TAG}_4k_pmc_traffic.json", f"{TAG}_4k_pmc_traffic.json"),
    (f"{TAG}_bench_line.json", f"{TAG}_bench_line.json"),
    (f"{TAG}_syncbn.jsonl", f"{TAG}_syncbn.jsonl"),
    (f"{TAG}_two_rank_overlap.jsonl", f"{TAG}_two_rank_overlap.jsonl"),
    (f"{TAG}_wg6_sweep.jsonl", f"{TAG}_wg6_sweep.jsonl"),
    (f"{TAG}_pmc_stalls.jsonl", f"{TAG}_pmc_stalls.jsonl"),
]


def main():
    sys.path.insert(0, REPO)
    import bench

    sha = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip()
    manifest = {"tag": TAG, "git_sha_at_collection": sha, "kernel_src_sha16": bench.kernel_source_sha16(), "files": {}}
    for src, dst in SOURCES:
        a, b = os.path.join(REPO, "gpurun_out", src), os.path.join(REPO, "profiles", dst)
        if not os.path.exists(a):
            continue
        if src.endswith(("_truth_tests.jsonl", "syncbn.jsonl", "overlap.jsonl")):
            # records the GPU tests append to: every line carries the hash of the kernel sources it was measured with (tests/conftest.py:
            # record_stamp).  Lines of another build are REFUSED (round 4 stamped whatever file lay around with the new revision).
            fresh = [ln for ln in open(a) if ln.strip() and json.loads(ln).get("kernel_src_sha16") == manifest["kernel_src_sha16"]]
            stale = sum(1 for ln in open(a) if ln.strip()) - len(fresh)
            if not fresh:
                print(f"collect_records: {src}: no line of this build ({stale} of other builds) - skipped")
                continue
            if src.endswith("_truth_tests.jsonl"):  # the LAST record of every (test, mode): the file is appended to by every run
                last = {}
                for ln in fresh:
                    r = json.loads(ln)
                    last[(r["test"], r["mode"])] = ln
                fresh = list(last.values())
            else:  # the newest entries (two ranks) of the rig records
                fresh = fresh[-2:]
            open(b, "w").write("".join(fresh))
        else:
            shutil.copyfile(a, b)
        manifest["files"][dst] = {"from": "gpurun_out/" + src, "sha256": hashlib.sha256(open(b, "rb").read()).hexdigest()[:16]}
    with open(os.path.join(REPO, "profiles", f"{TAG}_records_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
