"""Disassemble the gfx950 code objects inside liblhg_hip.so and count instructions matching a pattern (default: the packed-fp32 VALU
instructions the build must not contain, see __graft_entry__.NO_PACKED_FP32 and DESIGN.md §5).

    python tools/device_isa_scan.py [regex]          -> {"code_objects": n, "instructions": N, "matches": {mnemonic: count}}
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "learned_hologram_gan_amd", "lib", "liblhg_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def scan(pattern=r"v_pk_(fma|mul|add)_f32", lib=LIB):
    rx = re.compile(pattern)
    counts, total, n_obj = collections.Counter(), 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):  # one bundle per translation unit, concatenated by the linker
            part = os.path.join(tmp, f"bundle{i}.bin")
            open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"dev{i}.co")
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={part}", f"--output={co}"], check=True, capture_output=True)
            if not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            n_obj += 1
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            for ln in dis.splitlines():
                tok = ln.split()
                if len(tok) >= 1 and ln.startswith(("\t", " ")):
                    total += 1
                    m = rx.search(tok[0])
                    if m:
                        counts[tok[0]] += 1
    return {"code_objects": n_obj, "instructions": total, "matches": dict(counts)}


if __name__ == "__main__":
    print(json.dumps(scan(*sys.argv[1:2])))
