#!/usr/bin/env python
"""Audit of the compiled kernels for the packed-fp32 form that is unsafe next to MFMA waves on gfx950 (build container, no GPU needed).

`v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[..]` - a LOW result half computed from the HIGH dword of a source - returned wrong low halves
while other waves of the CU issued MFMAs (tools/pk_fault_repro.hip: registers only, 0 wrong alone, ~10^6 wrong next to an MFMA kernel; the forms
without modifiers and with op_sel_hi only were always right).  Every translation unit is compiled to assembly with the flags build.py uses and
scanned; exit status 1 if the form occurs anywhere.

    python tools/audit_op_sel.py
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disentangledcolorization_amd import build as B  # noqa: E402

PAT = re.compile(r"v_pk_(fma_f32|mul_f32|add_f32|mov_b32)\b.*\bop_sel:")        # the packed instructions on 64-bit register pairs


def scan(src):
    sp = os.path.join(B.CSRC, src)
    flags = [f for f in B.FLAGS if f != "-fPIC"] + B.EXTRA_FLAGS.get(src, [])
    r = subprocess.run([B.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", "-", sp], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
    kernel, hits, total = None, [], 0
    for line in r.stdout.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
        if re.search(r"v_pk_(fma_f32|mul_f32|add_f32|mov_b32)\b", line):
            total += 1
            if PAT.search(line):
                hits.append((kernel, line.strip()))
    return src, total, hits


def main():
    srcs = [s for s in B.SOURCES if s.endswith(".hip")]
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(scan, srcs))
    bad = 0
    for src, total, hits in res:
        print("%-18s %5d packed fp32 instructions, %d with op_sel" % (src, total, len(hits)))
        for k, l in hits[:5]:
            print("    %s: %s" % (k[:60], l))
        bad += len(hits)
    print("unsafe instructions:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
