"""Pin the CPU oracle (oracle/disco_ref.py) against golden vectors captured from the real
reference by oracle/make_golden.py.  CPU only; a few forwards of the oracle."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import disco_ref as R

TOL = 2e-5  # fp32 CPU vs fp32 CPU, different op order


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _close(a, b, tol=TOL):
    a = torch.as_tensor(np.asarray(a)); b = torch.as_tensor(np.asarray(b))
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item()
    assert err <= tol, f"max abs err {err:.3e} > {tol}"


# ---------------------------------------------------------------- components ------------


def test_gamut_table(golden_dir, q_to_ab):
    g = _load(golden_dir, "components")
    assert np.array_equal(g["q_to_ab"], q_to_ab)


@pytest.mark.parametrize("hw", [(16, 16), (32, 32), (32, 48), (48, 32), (8, 12)])
def test_position_encoding(golden_dir, hw):
    g = _load(golden_dir, "components")
    _close(R.position_encoding(*hw), g["pos_%dx%d" % hw], 1e-6)


def test_pool_unpool_sizes(golden_dir):
    g = _load(golden_dir, "components")
    prob, feat, tok = (torch.from_numpy(g[k]) for k in ("pool_prob", "pool_feat", "up_tok"))
    pooled, conf = R.poolfeat(feat, prob, 16)
    _close(pooled, g["pool_out"], 1e-5)
    _close(conf, g["pool_conf"], 1e-6)
    assert torch.equal(R.spixel_size(prob, 16), torch.from_numpy(g["spix_size"]))  # multiples of 1/256: exact
    _close(R.upfeat(tok, prob, 16), g["up_out"], 1e-6)


@pytest.mark.parametrize("t", [0, 1, 2])
def test_sample_anchor_colors(golden_dir, q_to_ab, t):
    g = _load(golden_dir, "components")
    out = R.sample_anchor_colors(torch.from_numpy(g["samp_prob"]), torch.from_numpy(q_to_ab), t)
    assert torch.equal(out, torch.from_numpy(g["samp_T%d" % t]))  # bin centres / 110: exact


def test_labels_and_decode(golden_dir, q_to_ab):
    g = _load(golden_dir, "components")
    q = torch.from_numpy(q_to_ab)
    ab = torch.from_numpy(g["enc_ab"])
    assert torch.equal(R.color_labels(ab, q), torch.from_numpy(g["enc_label"]))
    _close(R.encode_ab2ind(ab, q)[:, ::7], g["enc_soft_sub"], 1e-6)
    assert torch.equal(R.decode_ind2ab(torch.from_numpy(g["dec_logit"]), q, 0), torch.from_numpy(g["dec_ab_T0"]))


def test_colour_space_and_ranked_decode(golden_dir, q_to_ab):
    """§8f rows 1-2: rgb2lab / lab2rgb (basic.py:395-475) and decode_ind2ab for T = 1..3 (basic.py:196-209)."""
    g = _load(golden_dir, "components")
    _close(R.rgb2lab(torch.from_numpy(g["cs_rgb"])), g["cs_lab"], 2e-6)
    _close(R.lab2rgb(torch.from_numpy(g["cs_lab_in"])), g["cs_rgb_out"], 2e-6)
    q = torch.from_numpy(q_to_ab)
    for t in (1, 2, 3):
        assert torch.equal(R.decode_ind2ab(torch.from_numpy(g["dec_logit"]), q, t), torch.from_numpy(g["dec_ab_T%d" % t]))


def test_mark_color_hints_and_image_io(golden_dir):
    """§8f rows 1-2: mark_color_hints (basic.py:95-117) bit-exact against the reference on anchors at borders,
    corners and next to each other; fetch_data's pad-to-16 quirk (inference.py:26-31) and the uint8 round trip."""
    g = _load(golden_dir, "posthoc")
    gray, target, base, gate = (torch.from_numpy(g[k]) for k in ("gray", "target", "base", "gate"))
    for ks in (3, 5):
        assert torch.equal(R.mark_color_hints(gray, target, gate, ks), torch.from_numpy(g["marked_k%d" % ks]))
        assert torch.equal(R.mark_color_hints(gray, target, gate, ks, base), torch.from_numpy(g["marked_base_k%d" % ks]))
    q = torch.from_numpy(np.asarray(__import__("disentangledcolorization_amd.gamut", fromlist=["gamut_points"]).gamut_points()))
    for key, T in (("ann_ab_T038", 0.38), ("ann_ab_T150", 1.5)):      # annealed-mean decoding (basic.py:210-217)
        _close(R.decode_ind2ab(torch.from_numpy(g["ann_logit"]), q, T), g[key], 2e-6)
    rs = np.random.RandomState(3)
    for (h, w), (hp, wp) in {(37, 50): (48, 64), (32, 50): (48, 64), (37, 48): (48, 64), (32, 48): (32, 48)}.items():
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        gr, ab, rgb, hw = R.fetch_from_rgb8(img, org_size=True)
        assert gr.shape == (1, 1, hp, wp) and ab.shape == (1, 2, hp, wp) and rgb.shape == (1, 3, hp, wp) and hw == (h, w)
        assert torch.equal(gr[..., h:, :], gr[..., h - 1:h, :].expand(-1, -1, hp - h, -1))      # edge replication
        back = R.labs_to_rgb8(torch.cat((gr, ab), 1), h, w)
        assert back.shape == (1, h, w, 3) and np.abs(back[0].astype(int) - img.astype(int)).max() <= 1


def test_cv2_resize_restatement_hand_vectors():
    """inference.py:32-33 (the default input path): cv2.resize(..., (256,256), INTER_LINEAR) on uint8.  cv2 is absent offline,
    so the restatement (oracle cv2_resize_linear_u8, citing opencv 4.6 resize.cpp) is pinned on vectors worked out by hand
    from the published fixed-point algorithm, and cross-checked against float bilinear interpolation (<= 1 LSB)."""
    import torch.nn.functional as F
    # 1x2 -> 1x4 (upscale): f = (d+0.5)*0.5-0.5 = -0.25, 0.25, 0.75, 1.25 -> (s,f) = (0,0) (0,.25) (0,.75) (1,0);
    # coefficients x2048: (2048,0) (1536,512) (512,1536) (2048,0); rows D = 0, 130560, 391680, 522240;
    # one source row: b = (2048,0): ((2048*(D>>4))>>16 + 2)>>2 = (0+2)>>2, (255+2)>>2, (765+2)>>2, (1020+2)>>2
    assert R.cv2_resize_linear_u8(np.array([[[0], [255]]], np.uint8), 1, 4).ravel().tolist() == [0, 64, 191, 255]
    # 1x5 -> 1x2 (downscale, scale 2.5): f = 0.75, 3.25 -> (s,f) = (0,.75) (3,.25); D = 10*512+20*1536 = 35840, 40*1536+50*512 = 87040;
    # ((2048*(35840>>4))>>16 + 2)>>2 = (70+2)>>2 = 18;  ((2048*(87040>>4))>>16 + 2)>>2 = (170+2)>>2 = 43   (float: 17.5 / 42.5, rounded half up)
    assert R.cv2_resize_linear_u8(np.array([[[10], [20], [30], [40], [50]]], np.uint8), 1, 2).ravel().tolist() == [18, 43]
    # 3x1 -> 2x1 (vertical only, scale 1.5): f = 0.25, 1.75 -> (0,.25) (1,.75); rows D = v*2048; b = (1536,512), (512,1536):
    # ((1536*(0>>4))>>16) + ((512*(204800>>4))>>16) + 2 = 0 + 100 + 2 -> 25;  ((512*(204800>>4))>>16) + ((1536*(409600>>4))>>16) + 2 = 100+600+2 -> 175
    assert R.cv2_resize_linear_u8(np.array([[[0]], [[100]], [[200]]], np.uint8), 2, 1).ravel().tolist() == [25, 175]
    # 2x2 -> 4x3: a VERTICAL UPSCALE with a horizontally interpolated column.  x (clamped f): a = (2048,0) (1024,1024) (2048,0); y (f kept,
    # rows clipped): dy=0: s=-1, f=.75 -> rows (0,0), b = (512,1536); dy=3: s=1, f=.25 -> rows (1,1), b = (1536,512).  Middle column:
    # D0 = (10+21)*1024 = 31744, D0>>4 = 1984: top = ((512*1984)>>16) + ((1536*1984)>>16) + 2 = 15 + 46 + 2 -> 63>>2 = 15 (one weight of
    # 2048 on the same row would give (62+2)>>2 = 16); D1 = (110+121)*1024, D1>>4 = 14784: bottom = 346 + 115 + 2 -> 463>>2 = 115 (not 116);
    # dy=1: b = (1536,512) on rows (0,1): 46 + 115 + 2 -> 40
    assert R.cv2_resize_linear_u8(np.array([[[10], [21]], [[110], [121]]], np.uint8), 4, 3)[..., 0].tolist() == \
        [[10, 15, 21], [35, 40, 46], [85, 90, 96], [110, 115, 121]]
    # exact 2x downscale in both directions: cv::resize switches INTER_LINEAR to the area path, (a+b+c+d+2)>>2
    assert R.cv2_resize_linear_u8(np.array([[[1], [2]], [[3], [5]]], np.uint8), 1, 1).ravel().tolist() == [3]
    rs = np.random.RandomState(0)
    for (h, w, ho, wo) in [(37, 53, 256, 256), (480, 640, 256, 256), (612, 612, 256, 256), (512, 512, 256, 256), (100, 300, 64, 48), (256, 256, 256, 256)]:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        got = R.cv2_resize_linear_u8(img, ho, wo)
        t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
        ref = F.avg_pool2d(t, 2) if (h == 2 * ho and w == 2 * wo) else F.interpolate(t, size=(ho, wo), mode="bilinear", align_corners=False)
        assert got.shape == (ho, wo, 3) and got.dtype == np.uint8
        assert np.abs(got.astype(np.float64) - ref[0].permute(1, 2, 0).numpy()).max() <= 1.0
        if (h, w) == (ho, wo):
            assert np.array_equal(got, img)             # identity resize is exact
    gr, ab, rgb, hw = R.fetch_from_rgb8(rs.randint(0, 256, (100, 75, 3)).astype(np.uint8), org_size=False)
    assert gr.shape == (1, 1, 256, 256) and ab.shape == (1, 2, 256, 256) and hw == (100, 75)      # the ORIGINAL size (inference.py:26,42)


def test_spixelseg_standalone(golden_dir, synth_sd):
    """§8f row 4: models.model.SpixelSeg (segnet alone, `net.*` keys) == the oracle's segnet stage."""
    g = _load(golden_dir, "spixelseg")
    from disentangledcolorization_amd import synth

    n, h, w, seed = (int(v) for v in g["recipe"])
    gray, _ = synth.synth_inputs(n, h, w, seed=seed)
    _close(R.segnet_forward(synth_sd, gray), g["prob"], 1e-5)
    assert sorted(k[len("segnet."):] for k in synth_sd if k.startswith("segnet.")) == list(g["keys"])


def test_networks_standalone(golden_dir, synth_sd):
    """VERDICT r3 missing #5: models/network.py's SpixelNet / ColorProbNet / HourGlass2 run on their own in the reference
    (oracle/make_golden.py::networks_case) == the oracle's three conv stages on the same inputs."""
    g = _load(golden_dir, "networks")
    gray = torch.from_numpy(g["gray"])
    x65 = torch.from_numpy(g["x65"].astype(np.float32))
    _close(R.segnet_forward(synth_sd, gray), g["spixelnet"], 1e-5)
    _close(R.repnet_forward(synth_sd, gray), g["colorprobnet"], 2e-5)
    _close(R.enhance_forward(synth_sd, x65), g["hourglass2"], 2e-5)


def test_kmeans_matches_reference(golden_dir):
    g = _load(golden_dir, "components")
    torch.set_rng_state(torch.from_numpy(g["km_torch_rng"]))  # the reference's fallback draws
    for x, init, ids in zip(g["km_x"], g["km_init"], g["km_ids"]):
        a, passes, events = R.kmeans_one(torch.from_numpy(x), init, 8)
        assert np.array_equal(a.numpy(), ids.astype(np.int64))
    assert events > 0  # the last case has duplicated rows: the empty-cluster fallback ran


def test_kmeans_init_stream(golden_dir):
    g = _load(golden_dir, "components")
    np.random.seed(130)
    assert np.array_equal(R.kmeans_init_indices(4, 256, 8), g["km_init"])


# ---------------------------------------------------------------- full forwards ---------


def _run(golden_dir, synth_sd, q_to_ab, name):
    from disentangledcolorization_amd import synth

    g = _load(golden_dir, name)
    n, h, w, k, T, rh, iseed, seed = (int(v) for v in g["recipe"])
    test_mode, h2r, spos = (bool(v) for v in g["flags"]) if "flags" in g.files else (True, False, False)
    if "rgb8" in g.files:          # natural-image fixture: the inputs are the stored uint8 pixels through the reference's input path
        parts = [R.fetch_from_rgb8(im, org_size=True) for im in g["rgb8"]]
        gray, ab = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        assert torch.equal(gray, torch.from_numpy(g["gray"])) and torch.equal(ab[:, :, ::4, ::4], torch.from_numpy(g["ab_sub"]))
    else:
        gray, ab = synth.synth_inputs(n, h, w, seed=iseed, ab_scale=0.5)
    sd = synth.synth_state_dict(seed, hint2regress=True) if h2r else synth_sd
    use_mask = "pad_mask" in g.files       # use_mask fixtures: made on the checkpoint variant that has superpixels below 25 pixels
    if use_mask:
        sd = synth.small_superpixel_variant(sd)
    psize = int(g["psize"]) if "psize" in g.files else 16          # --psize (inference.py:147)
    oracle = R.DiscoOracle(sd, q_to_ab, sp_size=psize, n_clusters=k, random_hint=bool(rh), hint2regress=h2r, spix_pos=spos, use_mask=use_mask)
    np.random.seed(seed); torch.manual_seed(seed); random.seed(seed)
    out, info = oracle.forward(gray, ab, sampled_T=T, return_info=True, test_mode=test_mode)
    return g, out, info


def _check_forward(g, out, info):
    pal, ref, pred, aff, spix, mask = out
    fs, as_ = (int(v) for v in g["strides"])
    sub = int(g["sub"])
    _close(info["feats"][:, :, ::fs, ::fs], g["feats_sub"], 1e-4)   # features are O(10)
    _close(aff[: g["aff_sub"].shape[0], :, ::as_, ::as_], g["aff_sub"], 1e-5)
    _close(info["enc"], g["enc"], 1e-4)
    if "pad_mask" in g.files:          # the float key_padding_mask the reference built (model.py:121-125): must be non-trivial and equal
        pad = R.entry_mask(info["sizes"], int(g["psize"]) if "psize" in g.files else 16)
        assert torch.equal(pad, torch.from_numpy(g["pad_mask"])) and 0 < float(pad.sum()) < pad.numel()
    if "cluster_ids" in g.files:
        assert np.array_equal(info["assign"].numpy(), g["cluster_ids"].astype(np.int64))
    assert torch.equal(mask, torch.from_numpy(g["hint_mask"]))      # anchors bit-exact
    assert torch.equal(spix, torch.from_numpy(g["spix_colors"]))
    _close(info["dec"], g["dec"], 1e-4)
    if sub == 1:
        _close(info["hint"] if "hint" in info else g["hint"], g["hint"], 1e-4)
        _close(pal, g["pal_logit"], 1e-4); _close(ref, g["ref_logit"], 1e-4)
        _close(pred, g["pred_colors"], TOL)
    else:
        _close(pal[:, ::sub], g["pal_logit"], 1e-4); _close(ref[:, ::sub], g["ref_logit"], 1e-4)
        _close(pred[:, :, ::sub, ::sub], g["pred_colors"], TOL)
    assert abs(float(pred.abs().max()) - float(g["pred_absmax"])) < TOL


@pytest.mark.parametrize("name", ["fwd_n2_256_k8", "fwd_diverse_256_k16", "fwd_n1_128x192_k8",
                                  "fwd_randhint_128_k16", "fwd_gt_128_k8", "fwd_n1_512x768_k8",
                                  # validation forward (test_mode=False), --hint2regress, --spix_pos and their mix
                                  "fwd_val_128_k8", "fwd_h2r_128_k8", "fwd_spixpos_128x192_k8",
                                  "fwd_spixpos_h2r_diverse_128_k16",
                                  # two of the photographs the reference ships (data/*.jpg), 256 x 256 (round 4)
                                  "fwd_photo_256_k8",
                                  # use_mask=True (model.py:38,121-125): float key_padding_mask, additive under torch >= 1.9 (round 6)
                                  "fwd_usemask_128x192_k8", "fwd_usemask_512_k8",
                                  # --psize 8 / 32 (inference.py:147), and 8 with use_mask (threshold 25 / psize^2); round 6
                                  "fwd_psize8_128x192_k8", "fwd_psize32_256_k8", "fwd_psize8_usemask_128_k8"])
def test_forward_matches_reference(golden_dir, synth_sd, q_to_ab, name):
    g, out, info = _run(golden_dir, synth_sd, q_to_ab, name)
    _check_forward(g, out, info)


def test_use_mask_changes_the_result_and_is_additive(golden_dir, synth_sd, q_to_ab):
    """The use_mask fixtures are not vacuous: without the mask the same inputs give another encoder output; and the mask is ADDITIVE
    (+1.0 on the small superpixels' scores, torch >= 1.9) - a boolean reading (those keys excluded) gives yet another."""
    from disentangledcolorization_amd import synth
    g = _load(golden_dir, "fwd_usemask_128x192_k8")
    n, h, w, k, T, rh, iseed, seed = (int(v) for v in g["recipe"])
    gray, ab = synth.synth_inputs(n, h, w, seed=iseed, ab_scale=0.5)
    sd = synth.small_superpixel_variant(synth_sd)
    o = R.DiscoOracle(sd, q_to_ab, n_clusters=k)
    aff, feats, src, pos, spix_ab, sizes = o.tokens(gray, ab)
    pad = R.entry_mask(sizes, 16)
    plain = R.encoder_stack(sd, "wildpath", src, pos)
    added = R.encoder_stack(sd, "wildpath", src, pos, key_bias=pad)
    excluded = R.encoder_stack(sd, "wildpath", src, pos, key_bias=pad * -1e30)
    want = torch.from_numpy(g["enc"])
    assert (added - want).abs().max() < 1e-4
    assert (plain - want).abs().max() > 1e-3 and (excluded - want).abs().max() > 1e-3


def test_diverse_requires_single_image(synth_sd, q_to_ab):
    from disentangledcolorization_amd import synth

    gray, ab = synth.synth_inputs(2, 32, 32)
    with pytest.raises(RuntimeError):
        R.DiscoOracle(synth_sd, q_to_ab, n_clusters=2).forward(gray, ab, sampled_T=2)
