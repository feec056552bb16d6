"""-m gpu: the per-rank workloads of the 8-GPU BASELINE configurations, on the one GPU a test box has.

BASELINE config 3 (batch 512 over 8 GPUs, K = 8) and config 5 (batch 256 over 8 GPUs: --diverse K = 16 clustering / random_hint K = 16)
cannot run as a whole here, but what ONE rank does can: rank r of 8 makes the draws of the GLOBAL batch in image order, takes its
contiguous slice at its global offset and runs the HIP forward on it (runner.ShardedColorizer with virtual_rank=(8, r): the product code
of an 8-rank job minus the collective).  The LAST rank is the interesting one - its k-means rows / hint positions sit behind those of
all earlier images in NumPy's / `random`'s stream.  Each share is checked against the CPU oracle fed the same images and the same
draws: anchors exact, ab within the 1e-3 bar.  (The collective itself: tests/test_dist_gloo.py on gloo, tests/test_gpu_dist.py on one GPU.)"""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from disentangledcolorization_amd import synth  # noqa: E402
from disentangledcolorization_amd.model import AnchorColorProb  # noqa: E402
from disentangledcolorization_amd.runner import ShardedColorizer, global_draws, shard_bounds  # noqa: E402
from oracle import disco_ref as R  # noqa: E402

AB_TOL = 1e-3
WORLD = 8


def _err(a, b):
    return (torch.as_tensor(a).detach().cpu().double() - torch.as_tensor(b).detach().cpu().double()).abs().max().item()


def _seed(s=130):
    np.random.seed(s); torch.manual_seed(s); random.seed(s)


def _model(sd, k, random_hint=False):
    m = AnchorColorProb(inChannel=1, outChannel=313, sp_size=16, d_model=64, use_dense_pos=True, n_clusters=k, random_hint=random_hint,
                        enhanced=True, init_weights=False)
    m.load_state_dict(sd)
    return m.cuda().eval()


def _shard_inputs(n_global, rank):
    """bench.py's global batch (synth seed 5) and rank's slice of it."""
    gray, ab = synth.synth_inputs(n_global, 256, 256, seed=5)
    lo, hi = shard_bounds(n_global, WORLD, rank)
    return gray[lo:hi].contiguous(), ab[lo:hi].contiguous(), lo, hi


@pytest.mark.parametrize("rank", [7, 3])
def test_config3_share_of_rank_r_of_8_matches_the_oracle(synth_sd, q_to_ab, rank):
    """Config 3: images 448..511 (rank 7) / 192..255 (rank 3) of the 512-image global batch, K = 8."""
    n_global, k = 512, 8
    gray, ab, lo, hi = _shard_inputs(n_global, rank)
    assert (lo, hi) == (64 * rank, 64 * rank + 64)
    m = _model(synth_sd, k)
    sc = ShardedColorizer.from_model(m, exact_fallback=False, virtual_rank=(WORLD, rank))
    _seed()
    pred, mask = sc.colorize(gray.cuda(), ab.cuda(), n_global, 0, gather=False)
    torch.cuda.synchronize()
    assert pred.shape == (64, 2, 256, 256) and mask.shape == (64, 1, 16, 16)
    # the rows this rank used are the GLOBAL ones at its offset, and no empty-cluster draw happened (then reading the fallback stream from
    # its start, as the unsynchronised mode does, is exact)
    _seed()
    idx, _ = global_draws(n_global, 256, k, False)
    out, events = m.forward_once(gray.cuda(), ab.cuda(), True, 0, idx[lo:hi], None, None, None, True)
    torch.cuda.synchronize()
    assert int(events.sum()) == 0
    assert torch.equal(out[2], pred) and torch.equal(out[5], mask)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    oracle = R.DiscoOracle(synth_sd, q_to_ab, n_clusters=k)
    worst = 0.0
    for i0 in range(0, 64, 8):
        want = oracle.forward(gray[i0:i0 + 8], ab[i0:i0 + 8], init_idx=idx[lo + i0: lo + i0 + 8])
        assert torch.equal(mask[i0:i0 + 8].cpu(), want[5]), "anchors of global images %d..%d differ from the oracle" % (lo + i0, lo + i0 + 7)
        worst = max(worst, _err(pred[i0:i0 + 8], want[2]))
    print("config 3, rank %d of 8: max|ab - oracle| = %.3e" % (rank, worst))
    assert worst <= AB_TOL


def test_config5a_share_of_rank_7_of_8_matches_the_oracle(synth_sd, q_to_ab):
    """Config 5 (a): images 224..255 of the 256-image global batch, --diverse (three colorizations per image), K = 16 clustering.  The
    reference defines diverse sampling for N = 1 (model.py:148-159), so the oracle runs image by image."""
    n_global, k, rank = 256, 16, 7
    gray, ab, lo, hi = _shard_inputs(n_global, rank)
    assert (lo, hi) == (224, 256)
    m = _model(synth_sd, k)
    sc = ShardedColorizer.from_model(m, exact_fallback=False, virtual_rank=(WORLD, rank))
    _seed()
    pred, mask = sc.colorize(gray.cuda(), ab.cuda(), n_global, 1, gather=False)
    torch.cuda.synchronize()
    assert pred.shape == (96, 2, 256, 256) and mask.shape == (96, 1, 16, 16)
    _seed()
    idx, _ = global_draws(n_global, 256, k, False)
    out, events = m.forward_once(gray.cuda(), ab.cuda(), True, 1, idx[lo:hi], None, None, None, True)
    torch.cuda.synchronize()
    assert int(events.sum()) == 0 and torch.equal(out[2], pred)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    oracle = R.DiscoOracle(synth_sd, q_to_ab, n_clusters=k)
    worst = 0.0
    for i in range(hi - lo):
        want = oracle.forward(gray[i:i + 1], ab[i:i + 1], sampled_T=1, init_idx=idx[lo + i: lo + i + 1])
        assert torch.equal(mask[3 * i:3 * i + 3].cpu(), want[5]), "anchors of global image %d differ from the oracle" % (lo + i)
        assert torch.equal(out[4][3 * i:3 * i + 3].cpu(), want[4]), "anchor colours of global image %d differ" % (lo + i)
        worst = max(worst, _err(pred[3 * i:3 * i + 3], want[2]))
    print("config 5a, rank 7 of 8: max|ab - oracle| = %.3e over 96 colorizations" % worst)
    assert worst <= AB_TOL


def test_config5b_share_of_rank_7_of_8_matches_the_oracle(synth_sd, q_to_ab):
    """Config 5 (b): random_hint, K = 16: the hint positions are drawn from Python's `random` for the GLOBAL batch in image order; rank 7
    uses draws 224..255."""
    n_global, k, rank = 256, 16, 7
    gray, ab, lo, hi = _shard_inputs(n_global, rank)
    m = _model(synth_sd, k, random_hint=True)
    sc = ShardedColorizer.from_model(m, exact_fallback=False, virtual_rank=(WORLD, rank))
    _seed()
    pred, mask = sc.colorize(gray.cuda(), ab.cuda(), n_global, 0, gather=False)
    torch.cuda.synchronize()
    _seed()
    _, pos = global_draws(n_global, 256, k, True)
    want_mask = torch.zeros(hi - lo, 256)
    want_mask.scatter_(1, torch.as_tensor(pos[lo:hi], dtype=torch.long), 1.0)
    want_mask = want_mask.reshape(-1, 1, 16, 16)
    assert torch.equal(mask.cpu(), want_mask), "rank 7 did not use the global draws 224..255"
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    oracle = R.DiscoOracle(synth_sd, q_to_ab, n_clusters=k, random_hint=True)
    worst = 0.0
    for i0 in range(0, hi - lo, 8):
        want = oracle.forward(gray[i0:i0 + 8], ab[i0:i0 + 8], hint_mask=want_mask[i0:i0 + 8])
        assert torch.equal(want[5], want_mask[i0:i0 + 8])
        worst = max(worst, _err(pred[i0:i0 + 8], want[2]))
    print("config 5b, rank 7 of 8: max|ab - oracle| = %.3e" % worst)
    assert worst <= AB_TOL


def test_virtual_rank_refuses_collectives(synth_sd):
    m = _model(synth_sd, 8)
    sc = ShardedColorizer.from_model(m, exact_fallback=False, virtual_rank=(8, 7))
    g, a = synth.synth_inputs(1, 256, 256, seed=5)
    with pytest.raises(ValueError):
        sc.colorize(g.cuda(), a.cuda(), 8, 0, gather=True)
    with pytest.raises(ValueError):
        ShardedColorizer.from_model(m, exact_fallback=True, virtual_rank=(8, 7)).colorize(g.cuda(), a.cuda(), 8, 0, gather=False)
    with pytest.raises(ValueError):
        ShardedColorizer.from_model(m, virtual_rank=(8, 8))


def test_config4_full_mixed_batch_matches_the_oracle(synth_sd, q_to_ab):
    """BASELINE config 4 at its full size - eight 512x512 and eight 768x512 L images (bench.py's list: synth seeds 40.. / 60..) through
    runner.colorize_mixed (grouped by shape: two forwards) - against the CPU oracle run the reference's way, one file at a time with the
    draws in file order (inference.py:93-109): anchors exact for all sixteen, ab within the bar.  (bench.py times exactly this list.)"""
    from disentangledcolorization_amd.runner import colorize_mixed
    m = _model(synth_sd, 8)
    grays = [synth.synth_inputs(1, 512, 512, seed=40 + i)[0] for i in range(8)] + [synth.synth_inputs(1, 768, 512, seed=60 + i)[0] for i in range(8)]
    _seed()
    got = colorize_mixed(m, [g.cuda() for g in grays])
    torch.cuda.synchronize()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    oracle = R.DiscoOracle(synth_sd, q_to_ab, n_clusters=8)
    _seed()
    worst = 0.0
    for i, g in enumerate(grays):
        want = oracle.forward(g, torch.zeros(1, 2, g.shape[2], g.shape[3]))          # draws from NumPy's global state, file order
        assert torch.equal(got[i][5].cpu(), want[5]), "anchors of file %d (%dx%d) differ from the oracle" % (i, g.shape[2], g.shape[3])
        worst = max(worst, _err(got[i][2], want[2]))
    print("config 4, 8 x 512x512 + 8 x 768x512: max|ab - oracle| = %.3e" % worst)
    assert worst <= AB_TOL
