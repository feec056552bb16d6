#!/usr/bin/env python
"""Per-kernel hash of the gfx950 instruction streams inside a built library (build container, no GPU needed).

    python tools/kernel_isa_hash.py [lib.so] > before.txt;  ...edit / move code...;  python tools/kernel_isa_hash.py > after.txt;  diff

Used when source is MOVED between translation units (round 6: tokens.hip split into tokens / attention / kmeans / anchor_colors): a kernel
whose hash is unchanged is the same machine code, so its results are bit-identical without a GPU run.  Addresses are stripped; only
mnemonics and operands are hashed."""
import glob
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disentangledcolorization_amd import build as B  # noqa: E402


def hashes(lib):
    objdump = os.path.join(os.path.dirname(os.path.dirname(B.HIPCC)), "lib", "llvm", "bin", "llvm-objdump")
    d = tempfile.mkdtemp(prefix="disco_isa_")
    out = {}
    try:
        tmp = os.path.join(d, "lib.so")
        shutil.copy(lib, tmp)
        subprocess.run([objdump, "--offloading", tmp], check=True, capture_output=True)
        for co in glob.glob(tmp + ".*gfx950"):
            r = subprocess.run([objdump, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True)
            name, h, n = None, None, 0
            for line in r.stdout.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    if name:
                        out[name] = (h.hexdigest()[:16], n)
                    name, h, n = m.group(1), hashlib.sha256(), 0
                    continue
                m = re.match(r"^\s+(\S.*?)\s*(//.*)?$", line)
                if name and m and not line.startswith("Disassembly") and m.group(1) != "...":        # ("...": elided padding between kernels)
                    h.update(m.group(1).encode() + b"\n"); n += 1
            if name:
                out[name] = (h.hexdigest()[:16], n)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else B.LIB
    for k, (h, n) in sorted(hashes(lib).items()):
        print("%s %6d %s" % (h, n, k))
