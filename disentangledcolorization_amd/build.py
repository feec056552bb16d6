"""Build libdisco_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m disentangledcolorization_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so stays next to the sources (git-ignored, but it
travels with the gpurun snapshot).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdisco_hip.so")
SOURCES = ["util.cpp", "api_load.cpp", "api_plan.cpp", "api_calib.cpp", "api_diag.cpp", "api_ops.cpp", "conv_pack.cpp", "conv_mx.hip", "conv_mx_ar0.hip", "conv_mx_ar1.hip", "conv_mx_ar2.hip", "conv_mx_ar3.hip", "conv_direct.hip", "color.hip", "spixel.hip", "pool.hip", "tokens.hip", "attention.hip", "kmeans.hip", "anchor_colors.hip", "attention_mfma.hip", "diag.hip"]
# per-file extra flags
# -fno-slp-vectorize: the SLP vectoriser forms v_pk_*_f32 with op_sel (the low result half takes the HIGH dword of a source), and on this part
# that form returns a wrong low half while other waves of the CU issue MFMAs (tools/pk_fault_repro.hip, profiles/r03_pk_fma_op_sel_fault.txt;
# HISTORY.md section 4).  The conv kernels and tokens.hip contain no such instruction (tools/audit_op_sel.py checks every file).
# -amdgpu-mfma-vgpr-form: MFMA results in VGPRs (attention_mfma.hip: the softmax reads every score; no v_accvgpr_read per element)
EXTRA_FLAGS = {"pool.hip": ["-fno-slp-vectorize"], "spixel.hip": ["-fno-slp-vectorize"], "color.hip": ["-fno-slp-vectorize"],
               "attention_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
HEADERS = ["common.h", "ctx.h", "plan.h", "conv_mx_kernel.h", os.path.join("..", "..", "include", "disco_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        if force or _stale(obj, [sp, os.path.abspath(__file__)] + hdrs):        # (this file holds the flags)
            jobs.append((sp, obj))

    def cc(job):
        sp, obj = job
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(sp), []) + ["-c", sp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (sp, r.stderr))
        return sp

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for done in ex.map(cc, jobs):
                if verbose:
                    print("[build] compiled", os.path.basename(done), file=sys.stderr)
    objs = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print("[build] linked", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
