#!/usr/bin/env python
"""Register / scratch table of every conv3x3_mx_kernel instantiation (build container, no GPU needed):
compiles the per-arithmetic translation units with -Rpass-analysis=kernel-resource-usage and prints one row per instantiation.

    python tools/resource_usage.py > profiles/r03_conv_resource_usage.txt
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "disentangledcolorization_amd", "csrc")
# instantiations the default forward launches at batch 64 / 256x256 (profiles/r03_final_kernel_stats.csv)
DEFAULT = set()
stats = os.path.join(ROOT, "profiles", "r03_final_kernel_stats.csv")
if os.path.exists(stats):
    for line in open(stats):
        m = re.search(r"conv3x3_mx_kernel<([^>]*)>", line)
        if m:
            DEFAULT.add(m.group(1).replace(" ", "").replace("false", "0").replace("true", "1"))


def one(ar):
    src = os.path.join(CSRC, "conv_mx_ar%d.hip" % ar)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
           "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src]
    return subprocess.run(cmd, capture_output=True, text=True).stderr


def main():
    with ThreadPoolExecutor(4) as ex:
        outs = list(ex.map(one, range(4)))
    rows = []
    for text in outs:
        cur = None
        for line in text.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                # mangled template arguments: ...conv3x3_mx_kernelILi32ELi16ELi2ELi1ELi8ELi1ELb0ELb0ELi3EE...
                t = re.search(r"conv3x3_mx_kernelI((?:L[ib]\d+E)+)E", m.group(1))
                cur = {"inst": ",".join(re.findall(r"L[ib](\d+)E", t.group(1)))} if t else None
                if cur:
                    rows.append(cur)
                continue
            if cur is None:
                continue
            for key, pat in (("sgpr", r"SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                             ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, line)
                if m and key not in cur:
                    cur[key] = int(m.group(1))
    print("# conv3x3_mx_kernel<TW,TH,NT,STRIDE,WM,WN,MASKED,NSRC2,AR>: hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage (tools/resource_usage.py)")
    print("# AR 0 = f16+fp8x2, 1 = f16x2+fp8 (opt-in x2q), 2 = f16x3 (SpixelNet, ColorProbNet), 3 = f16+fp6x2 (HourGlass2).  * = launched by the default forward at batch 64 / 256x256")
    print("%-36s %5s %5s %8s %7s %7s %5s" % ("instantiation", "SGPR", "VGPR", "scratch", "s-spill", "v-spill", "occ"))
    for r in sorted(rows, key=lambda r: (int(r["inst"].split(",")[-1]), r["inst"])):
        star = "*" if r["inst"] in DEFAULT else " "
        print("%s%-35s %5d %5d %8d %7d %7d %5d" % (star, "<" + r["inst"] + ">", r.get("sgpr", -1), r.get("vgpr", -1) + r.get("agpr", 0), r.get("scratch", -1),
                                                 r.get("sspill", -1), r.get("vspill", -1), r.get("occ", -1)))
    bad = [r["inst"] for r in rows if r["inst"] in DEFAULT and r.get("scratch", 0) != 0]
    print("# default-path instantiations with scratch: %s" % (bad or "none"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
