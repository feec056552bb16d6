"""disentangledcolorization_amd — MI355X-native hot path of DISCO colorization.

Only what `AnchorColorProb.forward(test_mode=True)` needs: hand-written HIP
kernels for gfx950 behind a C ABI (csrc/, include/disco_hip.h) and the Python mirror of
the reference's `model.AnchorColorProb` interface (model.py).  Nothing here imports `oracle/`.
"""
__version__ = "0.1.0"
