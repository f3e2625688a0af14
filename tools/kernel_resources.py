"""Register / LDS / occupancy table of the gfx950 kernels, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/kernel_resources.py [file.hip ...] [--match substr]   (default: csrc/conv_engine.hip)"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
if match in args:
    args.remove(match)
files = args or [os.path.join(REPO, "learned_hologram_gan_amd", "csrc", "conv_engine.hip")]
keys = ("VGPRs", "AGPRs", "ScratchSize", "Occupancy", "LDS Size", "VGPRs Spill")
for f in files:
    # the SAME device flags as the shipped build (__graft_entry__.NO_PACKED_FP32): the table must describe the binary that runs
    sys.path.insert(0, REPO)
    from __graft_entry__ import NO_PACKED_FP32

    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-c", f, "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"] + NO_PACKED_FP32, capture_output=True, text=True).stderr
    cur, rows = None, {}
    for ln in out.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", ln)
        if m and cur and m.group(1) in keys:
            rows[cur][m.group(1)] = int(m.group(2))
    for k, v in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void lhg::", "").replace("lhg::", "")
        if match in name:
            print(f"{name[:70]:70s} vgpr {v.get('VGPRs', 0):3d} agpr {v.get('AGPRs', 0):3d} scratch {v.get('ScratchSize', 0):4d} "
                  f"occ {v.get('Occupancy', 0)} lds {v.get('LDS Size', 0):6d} spill {v.get('VGPRs Spill', 0)}")
