#!/usr/bin/env python3
"""VGPRs / spills / scratch / static LDS of the kernels in libzip_hip.so, from the code object's metadata notes
(no recompile).  usage: python3 tools/kernel_resources.py [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = os.environ.get("ZIP_HIP_LIB_PATH") or os.path.join(ROOT, "zinc_amd", "lib", "libzip_hip.so")
    want = sys.argv[1:] or ["raa_commit", "open_columns", "combine_", "sumcheck_round"]
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        lst = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={fat}"], capture_output=True, text=True).stdout.split()
        tgt = [t for t in lst if "gfx950" in t][0]
        co = os.path.join(d, "k.co")
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={tgt}", f"--input={fat}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for b in notes.split("  - .agpr_count")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]
        name = g("name")
        if not any(w in name for w in want):
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void zipk::", "")
        print(f"{dem:60s} vgpr {g('vgpr_count'):>4s}  sgpr_spill {g('sgpr_spill_count'):>4s}  vgpr_spill {g('vgpr_spill_count'):>3s}  scratch {g('private_segment_fixed_size'):>4s}  lds {g('group_segment_fixed_size'):>6s}")


if __name__ == "__main__":
    main()
