"""models/network.py's three conv networks as stand-alone modules on the HIP path (ABI 9: disco_forward_repnet / _enhance next to
disco_forward_segnet; disentangledcolorization_amd/network.py) against the reference's own stand-alone outputs
(tests/golden/networks.npz, oracle/make_golden.py::networks_case) and against the oracle's stages at a second size."""
import ctypes as C
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import disco_ref as R  # noqa: E402  (the checker)


def _err(a, b):
    return float((a.detach().float().cpu() - torch.as_tensor(b).float()).abs().max())


def _sub(sd, pre):
    return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}


@pytest.fixture(scope="module")
def nets(synth_sd):
    from disentangledcolorization_amd import network

    out = {}
    for name, cls, pre in (("spixelnet", network.SpixelNet, "segnet.net."), ("colorprobnet", network.ColorProbNet, "repnet."),
                           ("hourglass2", network.HourGlass2, "enhanceNet.")):
        m = cls()
        m.load_state_dict(_sub(synth_sd, pre))       # strict
        out[name] = m.cuda().eval()
    return out


def test_state_dicts_are_the_reference_subsets(nets, synth_sd):
    for name, pre in (("spixelnet", "segnet.net."), ("colorprobnet", "repnet."), ("hourglass2", "enhanceNet.")):
        want = _sub(synth_sd, pre)
        got = nets[name].state_dict()
        assert list(got.keys()) == list(want.keys())
        assert all(tuple(got[k].shape) == tuple(want[k].shape) for k in want)
    from disentangledcolorization_amd import network
    with pytest.raises(RuntimeError):
        network.ColorProbNet().load_state_dict({"conv1_2.0.bias": torch.zeros(64)})        # strict: missing keys (a fresh module: torch copies what matches before it raises)


def test_against_the_reference_run_stand_alone(nets, golden_dir):
    g = np.load(os.path.join(golden_dir, "networks.npz"))
    gray = torch.from_numpy(g["gray"]).cuda()
    x65 = torch.from_numpy(g["x65"].astype(np.float32)).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        e_s = _err(nets["spixelnet"](gray), g["spixelnet"])
        e_c = _err(nets["colorprobnet"](gray), g["colorprobnet"])
        e_h = _err(nets["hourglass2"](x65), g["hourglass2"])
    torch.cuda.synchronize()
    ref_c, ref_h = float(np.abs(g["colorprobnet"]).max()), float(np.abs(g["hourglass2"]).max())
    print(f"stand-alone networks vs the reference: SpixelNet {e_s:.2e}, ColorProbNet {e_c:.2e} (max |f| {ref_c:.2f}), "
          f"HourGlass2 {e_h:.2e} (max |y| {ref_h:.2f}, on {nets['hourglass2'].enhance_arithmetic()[0]})")
    assert e_s < 1e-4                       # softmax probabilities (f16x3)
    assert e_c < 1e-4 * max(1.0, ref_c)     # features (f16x3: ~1e-5 relative)
    assert e_h < 1e-3                       # pre-tanh ab (the colorizer's tolerance; measured ~1e-4 on fp6 corrections)


@pytest.mark.parametrize("precision", ["mx6", "mx8", "f16x3"])
def test_hourglass2_against_the_oracle_at_another_size(synth_sd, precision):
    from disentangledcolorization_amd import network, synth

    m = network.HourGlass2(precision=precision)
    m.load_state_dict(_sub(synth_sd, "enhanceNet."))
    m = m.cuda().eval()
    gray, _ = synth.synth_inputs(3, 96, 128, seed=31)
    g = torch.Generator().manual_seed(32)
    x = torch.cat([gray, torch.randn(3, 64, 96, 128, generator=g).abs() * 0.25], 1)
    want = R.enhance_forward(synth_sd, x)
    got = m(x.cuda())
    again = m(x.cuda())                     # second forward: no calibration, same result
    torch.cuda.synchronize()
    e = _err(got, want)
    print(f"HourGlass2 stand-alone, {precision}: max|y - y_ref| = {e:.2e} (max |y| {float(want.abs().max()):.2f})")
    assert torch.equal(got, again)
    assert e < (2e-5 if precision == "f16x3" else 1e-3)
    # a batch far outside the calibrated range: the fp8 clamps are noticed, the context re-calibrates and says so
    if precision == "mx8":
        m2 = network.HourGlass2(precision=precision)
        m2.load_state_dict(_sub(synth_sd, "enhanceNet."))
        m2 = m2.cuda().eval()
        m2(x.cuda())
        big = x.clone(); big[:, 1:] *= 400.0
        with pytest.warns(UserWarning, match="clamped"):
            got_big = m2(big.cuda())
        assert _err(got_big, R.enhance_forward(synth_sd, big)) < 1e-3 * max(1.0, float(R.enhance_forward(synth_sd, big).abs().max()))


def test_colorprobnet_against_the_oracle_at_another_size(nets, synth_sd):
    from disentangledcolorization_amd import synth

    gray, _ = synth.synth_inputs(2, 128, 96, seed=33)
    want = R.repnet_forward(synth_sd, gray)
    got = nets["colorprobnet"](gray.cuda())
    torch.cuda.synchronize()
    assert _err(got, want) < 1e-4 * max(1.0, float(want.abs().max()))


def test_unsupported_configurations_and_misuse(nets):
    import torch.nn as nn
    from disentangledcolorization_amd import _ffi, network

    for bad in (lambda: network.SpixelNet(inChannel=3), lambda: network.ColorProbNet(outChannel=2), lambda: network.ColorProbNet(with_SA=True),
                lambda: network.HourGlass2(inChannel=3, outChannel=1), lambda: network.HourGlass2(normLayer=None), lambda: network.HourGlass2(resNum=2)):
        with pytest.raises(NotImplementedError):
            bad()
    with pytest.raises(NotImplementedError):
        nets["colorprobnet"].train()
    with pytest.raises(ValueError):
        nets["hourglass2"](torch.zeros(1, 3, 32, 32, device="cuda"))
    with pytest.raises(ValueError):
        nets["colorprobnet"](torch.zeros(1, 1, 30, 32, device="cuda"))
    with pytest.raises(_ffi.DiscoError):
        nets["spixelnet"](torch.zeros(1, 1, 32, 32))
    # C ABI: an entry point of another network on a stand-alone context, and a HourGlass2 context before its calibration
    L = _ffi.lib()
    x = torch.zeros(1, 1, 32, 32, device="cuda")
    nets["colorprobnet"](x)
    ctx = nets["colorprobnet"]._ctx
    ws = torch.empty(1 << 26, device="cuda", dtype=torch.uint8)
    out = torch.empty(1, 9, 32, 32, device="cuda")
    rc = L.disco_forward_segnet(ctx, 1, 32, 32, x.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -5 and b"another network" in L.disco_last_error()
    need = C.c_size_t()
    assert L.disco_subnet_workspace_bytes(ctx, 2, 1, 32, 32, C.byref(need)) == 0 and 0 < need.value <= ws.numel()
    assert L.disco_subnet_workspace_bytes(ctx, 3, 1, 32, 32, C.byref(need)) == -1


def test_subnet_entries_on_a_colorizer_context(synth_sd):
    """disco_forward_repnet / _enhance also serve a full context (as disco_forward_segnet always did): same kernels, the colorizer's calibration."""
    from disentangledcolorization_amd import _ffi, synth
    from disentangledcolorization_amd.model import AnchorColorProb

    m = AnchorColorProb(n_clusters=8, enhanced=True, init_weights=False)
    m.load_state_dict(synth_sd)
    m = m.cuda().eval()
    gray, ab = synth.synth_inputs(2, 64, 64, seed=34)
    np.random.seed(130); torch.manual_seed(130)
    m(gray.cuda(), ab.cuda(), True, 0)
    L = _ffi.lib()
    need = C.c_size_t()
    assert L.disco_subnet_workspace_bytes(m._ctx, 2, 2, 64, 64, C.byref(need)) == 0
    ws = torch.empty(need.value, device="cuda", dtype=torch.uint8)
    feats = torch.empty(2, 64, 64, 64, device="cuda")
    _ffi.check(L.disco_forward_repnet(m._ctx, 2, 64, 64, gray.cuda().data_ptr(), feats.data_ptr(), ws.data_ptr(), ws.numel(), _ffi.current_stream()))
    torch.cuda.synchronize()
    want = R.repnet_forward(synth_sd, gray)
    assert _err(feats, want) < 1e-4 * max(1.0, float(want.abs().max()))
