#!/usr/bin/env python3
"""Register / spill / LDS figures of every kernel in one source file, from the gfx950 assembly's metadata
(no GPU needed):   tools/kernel_stats.py blur.hip [filter-regex]      (EXTRA="-D..." in the environment is passed on)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "."
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fno-slp-vectorize", "--cuda-device-only", "-S", src, "-o", out] + os.environ.get("EXTRA", "").split()
        subprocess.check_call(cmd, cwd=os.path.join(ROOT, "dps_ttc_amd", "csrc"))
        txt = open(out).read()
    meta = txt[txt.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in txt else ""
    for blk in re.split(r"\n  - ", meta)[1:]:
        def g(key):
            m = re.search(r"\." + key + r":\s*(\S+)", blk)
            return m.group(1) if m else "?"
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void dpsx::", "")
        if re.search(flt, name):
            print("%-58s vgpr %4s spill %4s  sgpr %4s spill %4s  scratch %5s  lds %6s" % (
                name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
                g("private_segment_fixed_size"), g("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
