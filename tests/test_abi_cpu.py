"""CPU-side checks of the C-ABI library and the host mirror: the .so loads, exports every symbol
include/disco_hip.h declares, agrees with layout.py on the checkpoint layout, and the drop-in
class keeps the reference's state_dict contract.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

from disentangledcolorization_amd import _ffi
from disentangledcolorization_amd.layout import state_dict_spec

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from disentangledcolorization_amd.build import build
    build(verbose=False)
    return _ffi.lib()


def test_header_symbols_exported(lib):
    text = open(os.path.join(REPO, "include", "disco_hip.h")).read()
    declared = set(re.findall(r"\b(disco_[a-z0-9_]+)\s*\(", text))
    declared -= {"disco_ctx"}
    assert declared, "no declarations parsed"
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.disco_abi_version() == 11


def test_native_layout_matches_python_spec(lib):
    spec = state_dict_spec()
    assert lib.disco_expected_tensors() == len(spec) == 461
    for i, (key, shape, dt, kind) in enumerate(spec):
        k, shp, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        assert lib.disco_expected_tensor(i, C.byref(k), shp, C.byref(nd)) == 0
        assert k.value.decode() == key and tuple(shp[: nd.value]) == tuple(shape)
    assert lib.disco_expected_tensor(461, C.byref(k), shp, C.byref(nd)) < 0
    assert b"bad index" in lib.disco_last_error()


def test_argument_errors_without_gpu(lib):
    assert lib.disco_workspace_bytes(None, 1, 256, 256, 0, C.byref(C.c_size_t())) < 0
    assert lib.disco_forward(None, None) < 0
    assert lib.disco_op_conv3x3(None, None, None, None, None, None, None, None, None, None) < 0
    nb = C.c_size_t()
    assert lib.disco_op_conv3x3_pack(None, 64, 65, None, C.byref(nb)) == 0
    assert nb.value == 2 * (80 // 16) * 9 * 2 * 1024     # 2 cout blocks x 5 cin chunks x 9 taps x {hi,lo} x 1 KiB


def test_dropin_state_dict_contract(synth_sd):
    from disentangledcolorization_amd.model import AnchorColorProb

    m = AnchorColorProb(inChannel=1, outChannel=313, sp_size=16, d_model=64, use_dense_pos=True, spix_pos=False,
                        learning_pos=False, n_clusters=8, random_hint=False, hint2regress=False, enhanced=True,
                        init_weights=False)
    keys = list(m.state_dict().keys())
    assert keys == [k for k, *_ in state_dict_spec()]
    m.load_state_dict(synth_sd)                       # strict, like utils_train.py:151
    assert torch.equal(m.state_dict()["repnet.conv4_3.2.weight_orig"], synth_sd["repnet.conv4_3.2.weight_orig"])
    bad = dict(synth_sd); bad.pop("mid_word_prj.weight")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    bad = dict(synth_sd); bad["extra.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    m.eval()
    with pytest.raises(NotImplementedError):
        AnchorColorProb(enhanced=False)
    assert AnchorColorProb(enhanced=True, use_mask=True, init_weights=False).use_token_mask is True      # (round 6: supported, model.py:38)
    # --hint2regress checkpoints carry differently shaped head tensors (model.py:63-64); strict both ways
    h = AnchorColorProb(enhanced=True, hint2regress=True, spix_pos=True, init_weights=False)
    assert tuple(h.state_dict()["trg_word_emb.weight"].shape) == (64, 67)
    assert tuple(h.state_dict()["trg_word_prj.weight"].shape) == (2, 64)
    with pytest.raises(RuntimeError):
        h.load_state_dict(synth_sd)
    from disentangledcolorization_amd import synth
    sd_h = synth.synth_state_dict(130, hint2regress=True)
    h.load_state_dict(sd_h)
    same = [k for k in synth_sd if not k.startswith("trg_word")]
    assert all(torch.equal(sd_h[k], synth_sd[k]) for k in same)


def test_checkpoint_file_roundtrip(synth_sd, tmp_path):
    """The reference's checkpoint format (train_colorizer.py:109-113 / utils_train.py:140-151): a torch.save'd dict with
    'state_dict' (+ ignored bookkeeping), loaded with map_location='cpu' and load_state_dict(strict)."""
    from disentangledcolorization_amd.model import AnchorColorProb

    m = AnchorColorProb(enhanced=True, init_weights=False)
    m.load_state_dict(synth_sd)
    path = tmp_path / "checkpoint.pth.rar"
    torch.save({"epoch": 3, "best_loss": 0.5, "state_dict": m.state_dict(), "optimizer": {}}, path)
    data = torch.load(path, map_location=torch.device("cpu"))
    m2 = AnchorColorProb(enhanced=True, init_weights=False)
    m2.load_state_dict(data["state_dict"])
    a, b = m.state_dict(), m2.state_dict()
    assert list(a.keys()) == list(b.keys()) and all(torch.equal(a[k], b[k]) for k in a)
    assert a["segnet.net.conv0a.1.num_batches_tracked"].dtype == torch.int64


def test_forward_refuses_cpu_tensors(synth_sd):
    from disentangledcolorization_amd.model import AnchorColorProb

    m = AnchorColorProb(enhanced=True, init_weights=False)
    with pytest.raises(_ffi.DiscoError):
        m(torch.zeros(1, 1, 32, 32), torch.zeros(1, 2, 32, 32), True, 0)


def test_peek_randint_matches_sequential_draws():
    """model._peek_randint must predict exactly what the reference's torch.randint(len(X),(1,)) calls would draw."""
    from disentangledcolorization_amd.model import AnchorColorProb

    for l in (96, 256, 1536):
        torch.manual_seed(11)
        peek = AnchorColorProb._peek_randint(l, 40)
        seq = [int(torch.randint(l, (1,))) for _ in range(40)]
        assert peek == seq


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no import / include / dlopen of it anywhere in the product package."""
    pkg = os.path.join(REPO, "disentangledcolorization_amd")
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|#\s*include\s*[\"<].*oracle|.*import_module\(.*oracle|.*CDLL\(.*oracle)", re.M)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                assert not pat.search(open(os.path.join(root, f)).read()), f


def test_dataparallel_replicas_share_the_native_context_and_train_is_refused():
    """inference.py:76-82 wraps the model in nn.DataParallel on multi-GPU hosts (spixelseg/inference.py:50-51 likewise) and calls it with
    batch 1: the wrapper must construct, a replica (a __dict__ copy with empty `_parameters`) must keep its origin, share - and never
    destroy - the origin's native context, and refuse only when it is CALLED on another device than the origin's.  train() raises like
    INTEGRATION.md says (eval() stays a no-op)."""
    import gc
    from disentangledcolorization_amd.model import AnchorColorProb, SpixelSeg
    from disentangledcolorization_amd import _ffi

    m = AnchorColorProb(n_clusters=8, enhanced=True, init_weights=False)
    assert m.eval() is m
    with pytest.raises(NotImplementedError, match="inference only"):
        m.train()
    for mod in (m, SpixelSeg()):
        dp = torch.nn.DataParallel(mod)                 # constructs (on a CPU-only host it forwards to .module)
        assert dp.module is mod and dp.eval() is dp
        destroyed = []
        mod._ctx = "sentinel-handle"                    # what a live context would be: the replica must carry it and leave it alone
        real = _ffi.lib
        try:
            _ffi.lib = lambda: type("L", (), {"disco_destroy": staticmethod(lambda h: destroyed.append(h))})
            r = mod._replicate_for_data_parallel()
            r2 = r._replicate_for_data_parallel()       # a replica of a replica still points at the module that owns the context
            assert r._dp_origin is mod and r2._dp_origin is mod and r._ctx == "sentinel-handle" and not r._parameters
            r._drop_ctx()
            del r, r2
            gc.collect()
            assert destroyed == [], "a replica destroyed the shared context"
            mod._drop_ctx()
            assert destroyed == ["sentinel-handle"]
        finally:
            _ffi.lib = real
        # replicated the way DataParallel does (parameters re-attached as plain attributes), then called off the origin's device: refused
        r = mod._replicate_for_data_parallel()
        x = torch.zeros(1, 1, 32, 32, device="meta")
        with pytest.raises(NotImplementedError, match="one process per GPU"):
            r(x) if isinstance(mod, SpixelSeg) else r(x, torch.zeros(1, 2, 32, 32, device="meta"), True, 0)
        with pytest.raises(_ffi.DiscoError, match="no CPU fallback"):      # on the origin's own device the call goes through to the origin
            xc = torch.zeros(1, 1, 32, 32)
            r(xc) if isinstance(mod, SpixelSeg) else r(xc, torch.zeros(1, 2, 32, 32), True, 0)


def test_mx_weight_pack_and_fp8_codec(lib):
    """The host side of the fp8-corrected conv: packed size, and that the packer is callable without a device (bytes only)."""
    nb = C.c_size_t()
    assert lib.disco_op_conv3x3_mx_pack(None, 64, 65, 0, None, None, C.byref(nb)) == 0
    assert nb.value == 2 * (96 // 16) * 9 * 2 * 1024     # 2 cout blocks x 3 groups of 32 channels x {H,Q} chunks x 9 taps x 2 KiB
    # the f16x2+fp8 arithmetic: 5 chunks (H L H L Q) per 64 input channels
    assert lib.disco_op_conv3x3_mx_pack(None, 64, 65, 1, None, None, C.byref(nb)) == 0
    assert nb.value == 2 * (128 // 64 * 5) * 9 * 2 * 1024
    # the f16+fp6x2 arithmetic: the geometry of variant 0 (fp6 slots are 32 bytes, 24 used); callable on the host alone
    assert lib.disco_op_conv3x3_mx_pack(None, 64, 65, 2, None, None, C.byref(nb)) == 0
    assert nb.value == 2 * (96 // 16) * 9 * 2 * 1024
    assert lib.disco_op_conv3x3_mx(None, None, None, None, None, None, None, None, None, None, None, None, None) < 0
    ab = C.c_size_t()
    assert lib.disco_op_act_bytes(2, 64, 8, 8, _ffi.PLANE_LO | _ffi.PLANE_Q, C.byref(ab)) == 0 and ab.value == 2 * 64 * 64 * 6
    assert lib.disco_op_act_bytes(2, 64, 8, 8, _ffi.PLANE_QL, C.byref(ab)) == 0 and ab.value == 2 * 64 * 64 * 3     # hi + al8 only
    assert lib.disco_op_act_bytes(2, 64, 8, 8, _ffi.PLANE_Q6, C.byref(ab)) == 0 and ab.value == 2 * 64 * 64 * 4     # hi + a6 | al6 slots


def test_no_packed_fp32_instruction_with_op_sel():
    """gfx950 erratum found in round 3 (tools/pk_fault_repro.hip): `v_pk_fma_f32 ... op_sel:[0,1,0]` (a LOW result half taken from the HIGH
    dword of a source) returns a wrong low half while other waves of the CU issue MFMAs.  The SLP vectoriser emits such forms, so the
    files whose packed code it generated are built with -fno-slp-vectorize; this compiles every kernel file to assembly and fails if the
    form (on any 64-bit packed instruction) appears anywhere."""
    import shutil
    import subprocess
    import sys
    from disentangledcolorization_amd import build as B
    if not (os.path.exists(B.HIPCC) or shutil.which(B.HIPCC)):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "audit_op_sel.py")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "unsafe instructions: 0" in r.stdout


def test_network_dropins_without_a_gpu(synth_sd):
    """disentangledcolorization_amd/network.py (models/network.py:125,147,260 as modules of their own): the three state_dicts are the
    colorizer's subsets with the prefix removed, strict loading works on the host, unsupported configurations raise, and a CPU tensor
    is refused (no CPU fallback)."""
    import pytest
    import torch
    import torch.nn as nn
    from disentangledcolorization_amd import _ffi, network

    for cls, pre, count in ((network.SpixelNet, "segnet.net.", 94), (network.ColorProbNet, "repnet.", 141), (network.HourGlass2, "enhanceNet.", 79)):
        m = cls()
        want = {k[len(pre):]: v for k, v in synth_sd.items() if k.startswith(pre)}
        assert len(want) == count and list(m.state_dict().keys()) == list(want.keys())
        m.load_state_dict(want)                                          # strict
        assert all(torch.equal(m.state_dict()[k], v) for k, v in want.items())
        with pytest.raises(RuntimeError):
            cls().load_state_dict({k: v for k, v in list(want.items())[1:]})
        with pytest.raises(_ffi.DiscoError):
            m(torch.zeros(1, 65 if cls is network.HourGlass2 else 1, 32, 32))
        with pytest.raises(NotImplementedError):
            m.train()
        assert m.eval() is m
    for bad in (lambda: network.SpixelNet(inChannel=3), lambda: network.ColorProbNet(outChannel=2),
                lambda: network.HourGlass2(inChannel=3, outChannel=1), lambda: network.HourGlass2(normLayer=nn.InstanceNorm2d)):
        with pytest.raises(NotImplementedError):
            bad()


def test_reference_side_change_script(tmp_path):
    """integration/apply_to_reference.py: the one-hunk reference-side change of INTEGRATION.md section 2 on a stand-in checkout (two files
    with the import line the reference has, CRLF line endings like the real checkout): patched, idempotent, revertible byte for byte."""
    import subprocess
    import sys
    script = os.path.join(REPO, "integration", "apply_to_reference.py")
    body = "import os\r\nfrom utils_train import load_checkpoint\r\nimport model, basic\r\nimport util\r\n"
    for sub in ("colorizer", "spixelseg"):
        d = tmp_path / "main" / sub
        d.mkdir(parents=True)
        (d / "inference.py").write_bytes(body.encode())
    run = lambda *a: subprocess.run([sys.executable, script, str(tmp_path)] + list(a), capture_output=True, text=True, check=True).stdout
    assert run().count("patched") == 2 and run().count("already done") == 2
    col = (tmp_path / "main" / "colorizer" / "inference.py").read_bytes()
    assert b"import basic\r\nfrom disentangledcolorization_amd import model\r\nimport util\r\n" in col and b"import model, basic" not in col
    seg = (tmp_path / "main" / "spixelseg" / "inference.py").read_bytes()
    assert b"from disentangledcolorization_amd import model, basic\r\n" in seg
    assert run("--revert").count("reverted") == 2
    assert (tmp_path / "main" / "colorizer" / "inference.py").read_bytes() == body.encode()


def test_ctypes_structs_mirror_the_header():
    """The binding's structs (disentangledcolorization_amd/_ffi.py) against include/disco_hip.h, field by field and in order: a field appended
    to the header (ABI 11: disco_options.use_mask) and forgotten in the binding would shift nothing and silently read as zero - or worse.  Also:
    disco_create refuses superpixel sizes other than 8 / 16 / 32 and the library and the binding agree on the ABI version."""
    from disentangledcolorization_amd import _ffi
    text = open(os.path.join(REPO, "include", "disco_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)

    def header_fields(name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):                       # "int32_t n, h, w" and "float *a, *b" declare several
                out.append(re.sub(r"\[.*?\]", "", part.strip().split()[-1].lstrip("*")))
        return out

    for cname, pyname in (("disco_options", "Options"), ("disco_forward_args", "ForwardArgs"), ("disco_conv_desc", "ConvDesc"), ("disco_conv_mx_desc", "ConvMxDesc")):
        want = header_fields(cname)
        got = [f[0] for f in getattr(_ffi, pyname)._fields_]
        assert got == want, "%s: binding %s, header %s" % (cname, got, want)
    assert re.search(r"#define DISCO_ABI_VERSION (\d+)", text).group(1) == str(_ffi.ABI_VERSION)
    lib = _ffi.lib()
    import ctypes as C
    for sp, ok in ((8, True), (16, True), (32, True), (4, False), (24, False), (64, False)):
        ctx = C.c_void_p()
        rc = lib.disco_create(0, C.byref(_ffi.Options(sp, 8, 0, _ffi.PREC_MX6, 0, 0, 0, 0)), C.byref(ctx))
        # (no GPU here: an accepted size gets as far as the device and fails there with DISCO_EHIP; a refused one never does)
        assert (rc != -3) == ok and (ok or b"sp_size" in lib.disco_last_error()), (sp, rc)
        if rc == 0:
            lib.disco_destroy(ctx)
