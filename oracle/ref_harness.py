"""Import the *real* reference (/root/reference) on a CPU-only box — TEST INFRASTRUCTURE.

Only usable in the build container (the reference tree does not travel to the GPU box and is
never copied into this repo).  Used by oracle/make_golden.py to produce tests/golden/*.npz, which tests/test_oracle_golden.py
replays against the oracle on any box.

Shims: stub modules for imports the forward never uses (torchvision, cv2, skimage,
tensorboardX), identity `.cuda()`, `cuda*` -> cpu in Tensor.to, cwd = main/colorizer because the
gamut .npy paths are cwd-relative (utils/cielab.py:6-7), sys.path like main/_init_paths.py:10-13.
No bytecode is written into the reference tree.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("DISCO_REFERENCE_ROOT", "/root/reference")
_installed = False


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "models", "model.py"))


def install():
    """Make `import model, basic, clusterkit, anchor_gen` resolve to the reference's modules."""
    global _installed
    if _installed:
        return
    import torch

    sys.dont_write_bytecode = True
    for name in ("torchvision", "cv2", "skimage", "skimage.segmentation", "skimage.color", "tensorboardX"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage.segmentation"].mark_boundaries = lambda *a, **k: None
    sys.modules["skimage"].segmentation = sys.modules["skimage.segmentation"]
    sys.modules["skimage"].color = sys.modules["skimage.color"]
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    _to = torch.Tensor.to

    def _cpu_to(self, *args, **kwargs):
        args = list(args)
        for i, a in enumerate(args):
            if isinstance(a, torch.device) and a.type == "cuda":
                args[i] = torch.device("cpu")
            elif isinstance(a, str) and a.startswith("cuda"):
                args[i] = "cpu"
        if "device" in kwargs and str(kwargs["device"]).startswith("cuda"):
            kwargs["device"] = "cpu"
        return _to(self, *args, **kwargs)

    torch.Tensor.to = _cpu_to
    os.chdir(os.path.join(REF_ROOT, "main", "colorizer"))
    for p in (REF_ROOT, os.path.join(REF_ROOT, "models"), os.path.join(REF_ROOT, "utils"),
              os.path.join(REF_ROOT, "main")):
        if p not in sys.path:
            sys.path.append(p)
    _installed = True


def build_reference_model(state_dict, n_clusters=8, random_hint=False, hint2regress=False, spix_pos=False, use_mask=False, sp_size=16):
    """The reference AnchorColorProb exactly as main/colorizer/inference.py:71-74,85,89 builds it
    (--hint2regress / --spix_pos: inference.py:156,158)."""
    install()
    import model  # the reference's models/model.py

    m = model.AnchorColorProb(inChannel=1, outChannel=313, sp_size=sp_size, d_model=64, use_dense_pos=True,
                              spix_pos=spix_pos, learning_pos=False, n_clusters=n_clusters,
                              random_hint=random_hint, hint2regress=hint2regress, enhanced=True, use_mask=use_mask)
    m.load_state_dict(state_dict)  # strict
    m.eval()
    return m
