#!/usr/bin/env python3
"""Register, spill and scratch figures of every kernel of a built library, from the code
object's metadata:  python tools/kernel_meta.py [path/to/lib.so]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mcmc_amd/lib/libmcmc_amd.so")
    with tempfile.TemporaryDirectory() as d:
        # the device code object sits in the .hip_fatbin section
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin",
                               lib, f"{d}/fat.bin"])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={d}/fat.bin", f"--output={d}/dev.co"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f"{d}/dev.co"],
                                        text=True)
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        name = g("name").group(1)
        name = subprocess.check_output(["c++filt", name], text=True).strip()
        name = name.replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*", "", name)
        print("%-60s vgpr %3s agpr %3s sgpr %3s spill %4s scratch %5s B lds %6s B" % (
            name[-60:], g("vgpr_count").group(1), blk.split()[0], g("sgpr_count").group(1),
            g("vgpr_spill_count").group(1), g("private_segment_fixed_size").group(1),
            g("group_segment_fixed_size").group(1)))


if __name__ == "__main__":
    main()
