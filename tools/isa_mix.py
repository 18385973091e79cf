#!/usr/bin/env python3
"""Instruction mix per kernel from a gfx950 assembly listing (hipcc -S ... --cuda-device-only):
    tools/isa_mix.py blur.hip [name-filter]
Counts the opcodes the DESIGN notes argue about (packed / scalar FMAs, LDS read widths, cross-lane ops, scratch)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATS = [("pk_fma", r"v_pk_fma_f32"), ("op_sel", r"v_pk_fma_f32.*op_sel"), ("fma", r"\bv_fma(c|ak|mk)?_f32"),
        ("b64", r"ds_read_b64"), ("b128", r"ds_read_b128"), ("r2b32", r"ds_read2_b32"), ("b32", r"ds_read_b32"),
        ("r2b64", r"ds_read2(st64)?_b64"), ("bperm", r"ds_bpermute"), ("dpp", r"_dpp"), ("scratch", r"scratch_"),
        ("s_load", r"s_load_"), ("valu", r"^\s+v_"), ("total", r"^\s+[a-z]")]


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "."
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fno-slp-vectorize", "--cuda-device-only", "-S", src, "-o", out] + os.environ.get("EXTRA", "").split()
        subprocess.check_call(cmd, cwd=os.path.join(ROOT, "dps_ttc_amd", "csrc"), stderr=subprocess.DEVNULL)
        txt = open(out).read()
    for m in re.finditer(r"^(_ZN4dpsx\w+):.*?\n(.*?)\.Lfunc_end", txt, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void dpsx::", "")
        if not re.search(flt, name):
            continue
        body = m.group(2)
        print("%-44s " % name[:44] + " ".join("%s %d" % (k, len(re.findall(p, body, re.M))) for k, p in PATS))


if __name__ == "__main__":
    main()
