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


def audit_built(lib_path=None):
    """The same scan on the BUILT library (what ships): its gfx950 code objects are extracted with `llvm-objdump --offloading` into a
    scratch directory and disassembled - seconds, so __graft_entry__.build() runs it after every build.  Returns (instructions, unsafe)."""
    import glob
    import shutil
    import tempfile
    objdump = os.path.join(os.path.dirname(os.path.dirname(B.HIPCC)), "lib", "llvm", "bin", "llvm-objdump")
    d = tempfile.mkdtemp(prefix="disco_audit_")
    try:
        tmp = os.path.join(d, "lib.so")
        shutil.copy(lib_path or B.LIB, tmp)
        subprocess.run([objdump, "--offloading", tmp], check=True, capture_output=True)
        objs = glob.glob(tmp + ".*gfx950")
        if not objs:
            raise RuntimeError("no gfx950 code object found in %s" % (lib_path or B.LIB))
        total, hits = 0, []
        for co in objs:
            r = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True)
            kernel = None
            for line in r.stdout.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
                if m:
                    kernel = m.group(1)
                if re.search(r"v_pk_(fma_f32|mul_f32|add_f32|mov_b32)\b", line):
                    total += 1
                    if PAT.search(line):
                        hits.append((kernel, line.split("//")[0].strip()))
        return total, hits
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    if "--built" in sys.argv:
        total, hits = audit_built()
        print("%s: %d packed fp32 instructions, %d with op_sel" % (os.path.basename(B.LIB), total, len(hits)))
        for k, l in hits[:10]:
            print("    %s: %s" % ((k or "?")[:60], l))
        return 1 if hits else 0
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
