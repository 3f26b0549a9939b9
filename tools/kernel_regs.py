#!/usr/bin/env python3
"""VGPR / SGPR / scratch of the kernels of one source file of csrc/, from the compiler's metadata of a fresh device-only compile
(hipcc cross-compiles gfx950 without a GPU): `python tools/kernel_regs.py track_kernels.hip [name-filter]`."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyfeaturetrack_amd", "csrc")


def kernel_regs(source, name_filter=""):
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                        "-fno-fast-math", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", asm,
                        os.path.join(CSRC, source)], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        name, body = m.group(1), m.group(2)
        if name_filter not in name:
            continue

        def field(k):
            r = re.search(r"\.%s:\s+(\d+)" % k, body)
            return int(r.group(1)) if r else None
        out[name] = {"vgpr": field("vgpr_count"), "sgpr": field("sgpr_count"), "scratch": field("private_segment_fixed_size"),
                     "lds_static": field("group_segment_fixed_size")}
    return out


if __name__ == "__main__":
    for name, r in sorted(kernel_regs(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "").items()):
        print("%-90s vgpr %3s  sgpr %3s  scratch %4s  lds %6s" % (name, r["vgpr"], r["sgpr"], r["scratch"], r["lds_static"]))
