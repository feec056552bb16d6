"""CPU oracle for DISCO's colorization hot path  —  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (plain PyTorch fp32 on the CPU) of the
arithmetic of `AnchorColorProb.forward(..., test_mode=True)` in
MenghanXia/DisentangledColorization.  It exists to CHECK the HIP path:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  The product package (`disentangledcolorization_amd/`) never imports
anything from `oracle/` and fails loudly when its HIP library is missing.

Pinning: the reference has no tests of its own, so the oracle is
pinned against outputs of the reference itself, run in the build container by
`oracle/make_golden.py` (which imports /root/reference) and committed as
`tests/golden/*.npz`; `tests/test_oracle_golden.py` replays them.

Every function cites the reference lines it restates (paths relative to the
reference root).  Nothing here is copied: the reference builds nn.Modules and
uses nn.MultiheadAttention / spectral_norm hooks / pad-and-slice shifting; the
oracle works directly on the checkpoint `state_dict` with explicit math.
"""
from __future__ import annotations

import math
import random as _pyrandom
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]
BN_EPS = 1e-5
LN_EPS = 1e-5

# --------------------------------------------------------------------------------------
# weights: spectral norm / batch norm in eval mode
# --------------------------------------------------------------------------------------


def conv_weight(sd: SD, key: str) -> Tensor:
    """Effective conv weight of layer `key`.

    Plain layers carry `.weight`.  Spectral-norm layers (network.py:152-186, :36) carry
    `weight_orig/weight_u/weight_v`; in eval mode torch's hook does no power iteration and
    uses W = weight_orig / (u . (W_mat v)) with W_mat = weight_orig.view(Cout, -1).
    """
    if key + ".weight" in sd:
        return sd[key + ".weight"]
    w = sd[key + ".weight_orig"]
    sigma = torch.dot(sd[key + ".weight_u"], torch.mv(w.reshape(w.shape[0], -1), sd[key + ".weight_v"]))
    return w / sigma


def conv3x3(sd: SD, key: str, x: Tensor, stride: int = 1) -> Tensor:
    """3x3 conv, pad 1 (every Conv2d on the path: network.py:14,19,70,86-87,134,152-201,243,282)."""
    return F.conv2d(x, conv_weight(sd, key), sd.get(key + ".bias"), stride=stride, padding=1)


BNObserver = Callable[[str, Tensor, SD], None]


def batchnorm(sd: SD, key: str, x: Tensor, observer: Optional[BNObserver] = None) -> Tensor:
    """Eval-mode BatchNorm2d: (x-mean)/sqrt(var+1e-5)*weight+bias with running stats.

    `observer(key, x, sd)` (used only by oracle/calibrate_synth.py) sees the pre-BN
    tensor and may rewrite the running stats in `sd` before they are applied.
    """
    if observer is not None:
        observer(key, x, sd)
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                        sd[key + ".weight"], sd[key + ".bias"], False, 0.0, BN_EPS)


def up2(x: Tensor) -> Tensor:
    """Nearest x2 upsample: dst[y,x] = src[y//2, x//2] (network.py:98,188,195,199)."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


# --------------------------------------------------------------------------------------
# a1  SpixelNet  (network.py:260-313, conv()/deconv() :240-258)
# --------------------------------------------------------------------------------------


def segnet_forward(sd: SD, gray: Tensor, observer: Optional[BNObserver] = None) -> Tensor:
    """(N,1,H,W) -> affinity (N,9,H,W), softmax over the 9 neighbour slots."""
    p = "segnet.net."

    def cbl(name: str, x: Tensor, stride: int = 1) -> Tensor:  # conv(no bias) -> BN -> LeakyReLU(0.1)
        y = conv3x3(sd, p + name + ".0", x, stride)
        return F.leaky_relu(batchnorm(sd, p + name + ".1", y, observer), 0.1)

    def dcl(name: str, x: Tensor) -> Tensor:  # ConvTranspose2d 4x4 s2 p1 + bias -> LeakyReLU(0.1)
        y = F.conv_transpose2d(x, sd[p + name + ".0.weight"], sd[p + name + ".0.bias"], stride=2, padding=1)
        return F.leaky_relu(y, 0.1)

    o1 = cbl("conv0b", cbl("conv0a", gray))
    o2 = cbl("conv1b", cbl("conv1a", o1, 2))
    o3 = cbl("conv2b", cbl("conv2a", o2, 2))
    o4 = cbl("conv3b", cbl("conv3a", o3, 2))
    o5 = cbl("conv4b", cbl("conv4a", o4, 2))
    d3 = cbl("conv3_1", torch.cat((o4, dcl("deconv3", o5)), 1))
    d2 = cbl("conv2_1", torch.cat((o3, dcl("deconv2", d3)), 1))
    d1 = cbl("conv1_1", torch.cat((o2, dcl("deconv1", d2)), 1))
    d0 = cbl("conv0_1", torch.cat((o1, dcl("deconv0", d1)), 1))
    return torch.softmax(conv3x3(sd, p + "pred_mask0", d0), dim=1)


# --------------------------------------------------------------------------------------
# a2  ColorProbNet  (network.py:147-236)
# --------------------------------------------------------------------------------------


def repnet_forward(sd: SD, gray: Tensor, observer: Optional[BNObserver] = None,
                   taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """(N,1,H,W) -> features (N,64,H,W), non-negative (ends in ReLU)."""
    p = "repnet."

    def sn_block(name: str, x: Tensor, idxs: Tuple[int, ...], bn_idx: int, first_stride: int) -> Tensor:
        for j, i in enumerate(idxs):  # SN conv + bias -> LeakyReLU(0.2)
            x = F.leaky_relu(conv3x3(sd, f"{p}{name}.{i}", x, first_stride if j == 0 else 1), 0.2)
        x = batchnorm(sd, f"{p}{name}.{bn_idx}", x, observer)  # BN closes the block
        if taps is not None:
            taps[name] = x
        return x

    f1 = sn_block("conv1_2", gray, (0, 2), 4, 1)
    f2 = sn_block("conv2_3", f1, (0, 2, 4), 6, 2)
    f3 = sn_block("conv3_3", f2, (0, 2, 4), 6, 2)
    f4 = sn_block("conv4_3", f3, (0, 2, 4), 6, 2)
    f5 = sn_block("conv5_3", f4, (0, 2, 4), 6, 1)
    f6 = sn_block("conv6_3", f5, (0, 2, 4), 6, 1)
    f7 = sn_block("conv7_3", f6, (0, 2, 4), 6, 1)
    # decoder: up->conv plus a shortcut conv of f3, ReLU, two conv+ReLU, BN   (:187-193, :227-228)
    f8 = F.relu(conv3x3(sd, p + "conv8up.1", up2(f7)) + conv3x3(sd, p + "conv3short8.0", f3))
    f8 = F.relu(conv3x3(sd, p + "conv8_3.1", f8))
    f8 = F.relu(conv3x3(sd, p + "conv8_3.3", f8))
    f8 = batchnorm(sd, p + "conv8_3.5", f8, observer)
    # up->conv (no activation) -> conv -> ReLU -> BN                           (:195-197, :229-230)
    f9 = conv3x3(sd, p + "conv9up.1", up2(f8))
    f9 = batchnorm(sd, p + "conv9_2.2", F.relu(conv3x3(sd, p + "conv9_2.0", f9)), observer)
    # up->conv -> ReLU -> conv -> ReLU                                         (:199-201, :231-232)
    f10 = F.relu(conv3x3(sd, p + "conv10up.1", up2(f9)))
    f10 = F.relu(conv3x3(sd, p + "conv10_2.1", f10))
    if taps is not None:
        taps.update(f8=f8, f9=f9, f10=f10)
    return f10


# --------------------------------------------------------------------------------------
# a13  HourGlass2  (network.py:125-144 with ConvBlock :10-28, DownsampleBlock :66-80,
#                   ResidualBlock :31-47, UpsampleBlock :83-101), tanh in model.py:197
# --------------------------------------------------------------------------------------


def enhance_forward(sd: SD, x: Tensor, observer: Optional[BNObserver] = None) -> Tensor:
    """(N,65,H,W) = cat(gray, upsampled tokens) -> (N,2,H,W) pre-tanh ab."""
    p = "enhanceNet."
    f1 = F.relu(conv3x3(sd, p + "inConv.inConv.0", x))
    f1 = batchnorm(sd, p + "inConv.conv.2", F.relu(conv3x3(sd, p + "inConv.conv.0", f1)), observer)

    def down(name: str, t: Tensor) -> Tensor:
        t = F.relu(conv3x3(sd, f"{p}{name}.conv.0", t, 2))
        return batchnorm(sd, f"{p}{name}.conv.4", F.relu(conv3x3(sd, f"{p}{name}.conv.2", t)), observer)

    f2 = down("down1", f1)
    f3 = down("down2", f2)
    r = f3
    for i in range(3):  # conv -> SN conv -> ReLU -> conv ; + skip ; ReLU  (no norm layers)
        t = conv3x3(sd, f"{p}residual.{i}.conv.0", r)
        t = F.relu(conv3x3(sd, f"{p}residual.{i}.conv.1", t))
        t = conv3x3(sd, f"{p}residual.{i}.conv.3", t)
        r = F.relu(r + t)

    def up(name: str, t: Tensor, skip: Tensor) -> Tensor:
        t = up2(conv3x3(sd, f"{p}{name}.conv1", t))
        t = F.relu(conv3x3(sd, f"{p}{name}.combine", torch.cat((t, skip), 1)))
        t = F.relu(conv3x3(sd, f"{p}{name}.conv2.0", t))
        t = F.relu(conv3x3(sd, f"{p}{name}.conv2.2", t))
        return batchnorm(sd, f"{p}{name}.conv2.4", t, observer)

    r2 = up("up2", r, f2)
    r1 = up("up1", r2, f1)
    return conv3x3(sd, p + "outConv", r1)


# --------------------------------------------------------------------------------------
# a3/a4/a12  superpixel pooling / sizes / unpooling   (basic.py:274-324, :327-335, :338-376)
#   slot c = (dy+1)*3 + (dx+1): a pixel in cell (a,b) with probability P_c belongs to
#   superpixel (a+dy, b+dx).
# --------------------------------------------------------------------------------------


def _slot(c: int) -> Tuple[int, int]:
    return c // 3 - 1, c % 3 - 1


def poolfeat(feat: Tensor, prob: Tensor, sp: int) -> Tuple[Tensor, Tensor]:
    """feat (N,C,H,W), prob (N,9,H,W) -> pooled (N,C,h,w), conf (N,1,h,w).

    num[i,j] = sum_c mean_{p in cell(i-dy, j-dx)} feat(p) P_c(p)  (cells outside the grid give 0),
    den likewise with feat == 1; pooled = num / (den + 1e-8); conf = den.  Slots are
    accumulated in the order 0..8 like the reference.
    """
    n, ch, hh, ww = feat.shape
    h, w = hh // sp, ww // sp
    ext = torch.cat((feat, feat.new_ones(n, 1, hh, ww)), dim=1)
    acc = feat.new_zeros(n, ch + 1, h, w)
    for c in range(9):
        dy, dx = _slot(c)
        cell = F.avg_pool2d(ext * prob[:, c:c + 1], kernel_size=sp, stride=sp)
        i0, i1 = max(dy, 0), h + min(dy, 0)
        j0, j1 = max(dx, 0), w + min(dx, 0)
        acc[:, :, i0:i1, j0:j1] += cell[:, :, i0 - dy:i1 - dy, j0 - dx:j1 - dx]
    den = acc[:, ch:]
    return acc[:, :ch] / (den + 1e-8), den


def spixel_size(prob: Tensor, sp: int) -> Tensor:
    """Hard-assigned pixel count / sp^2 per superpixel; ties count for every maximal slot."""
    hard = (prob == prob.max(dim=1, keepdim=True)[0]).to(prob.dtype)
    return poolfeat(prob.new_ones(prob.shape[0], 1, prob.shape[2], prob.shape[3]), hard, sp)[1]


def upfeat(tok: Tensor, prob: Tensor, sp: int) -> Tensor:
    """tok (N,C,h,w), prob (N,9,H,W) -> (N,C,H,W): out(p) = sum_c P_c(p) tok[cell(p)+(dy,dx)]."""
    n, ch, h, w = tok.shape
    pad = F.pad(tok, (1, 1, 1, 1))
    out = None
    for c in range(9):
        dy, dx = _slot(c)
        nb = pad[:, :, 1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
        term = nb.repeat_interleave(sp, dim=2).repeat_interleave(sp, dim=3) * prob[:, c:c + 1]
        out = term if out is None else out + term
    return out


# --------------------------------------------------------------------------------------
# a5  sine position encoding  (position_encoding.py:26-47, built at model.py:59)
# --------------------------------------------------------------------------------------


def position_encoding(h: int, w: int, n_feats: int = 32, temperature: float = 10000.0) -> Tensor:
    """(64,h,w): channels 0..31 from y, 32..63 from x; even channel sin, odd channel cos."""
    ye = torch.arange(1, h + 1, dtype=torch.float32)
    xe = torch.arange(1, w + 1, dtype=torch.float32)
    ye = ye / (ye[-1:] + 1e-6) * (2 * math.pi)
    xe = xe / (xe[-1:] + 1e-6) * (2 * math.pi)
    k = torch.arange(n_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(k, 2, rounding_mode="floor") / n_feats)

    def enc(e: Tensor) -> Tensor:  # (len,) -> (len, n_feats)
        a = e[:, None] / dim_t
        out = torch.empty_like(a)
        out[:, 0::2] = a[:, 0::2].sin()
        out[:, 1::2] = a[:, 1::2].cos()
        return out

    py = enc(ye).t()[:, :, None].expand(n_feats, h, w)
    px = enc(xe).t()[:, None, :].expand(n_feats, h, w)
    return torch.cat((py, px), dim=0).contiguous()


# --------------------------------------------------------------------------------------
# a6  transformer encoder stack  (transformer2d.py:9-60; nn.MultiheadAttention semantics)
# --------------------------------------------------------------------------------------


def encoder_layer(sd: SD, pre: str, x: Tensor, pos: Tensor, n_head: int = 8, key_bias: Optional[Tensor] = None) -> Tensor:
    """x,pos: (N,L,E).  q=k=x+pos, v=x; post-norm; dropouts are identity in eval.
    key_bias (N,L) float: `use_mask` (model.py:122-124,133,186 -> transformer2d.py:53-54 `key_padding_mask=`).  The reference hands
    nn.MultiheadAttention a FLOAT mask (1.0 at superpixels smaller than 25 pixels, else 0.0); torch >= 1.9 treats a float
    key_padding_mask as ADDITIVE - F.multi_head_attention_forward merges it into attn_mask and computes
    baddbmm(attn_mask, q_scaled, k^T) = mask + q k^T before the softmax - so the small superpixels' keys get +1.0 on every score
    (pinned on the live reference under this container's torch 2.10: tests/golden/fwd_usemask_*.npz).  The reference's own pin,
    torch 1.8 (environment.yaml:59), rejects a float mask in its masked_fill: there the option cannot run at all."""
    n, l, e = x.shape
    hd = e // n_head
    w_in, b_in = sd[pre + "self_attn.in_proj_weight"], sd[pre + "self_attn.in_proj_bias"]
    qk_in = x + pos
    q = F.linear(qk_in, w_in[:e], b_in[:e]) * math.sqrt(1.0 / hd)
    k = F.linear(qk_in, w_in[e:2 * e], b_in[e:2 * e])
    v = F.linear(x, w_in[2 * e:], b_in[2 * e:])
    split = lambda t: t.reshape(n, l, n_head, hd).permute(0, 2, 1, 3)  # (N,heads,L,hd)
    scores = split(q) @ split(k).transpose(-1, -2)
    if key_bias is not None:
        scores = key_bias.to(scores.dtype)[:, None, None, :] + scores
    att = torch.softmax(scores, dim=-1)
    o = (att @ split(v)).permute(0, 2, 1, 3).reshape(n, l, e)
    o = F.linear(o, sd[pre + "self_attn.out_proj.weight"], sd[pre + "self_attn.out_proj.bias"])
    x = F.layer_norm(x + o, (e,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], LN_EPS)
    f = F.linear(F.relu(F.linear(x, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"])),
                 sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])
    return F.layer_norm(x + f, (e,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], LN_EPS)


def encoder_stack(sd: SD, path: str, x: Tensor, pos: Tensor, n_layers: int = 6, key_bias: Optional[Tensor] = None) -> Tensor:
    """Dense-pos variant (transformer2d.py:18-22): pos is re-added to q,k in every layer; key_bias: see encoder_layer."""
    for i in range(n_layers):
        x = encoder_layer(sd, f"{path}.layers.{i}.", x, pos, key_bias=key_bias)
    return x


def entry_mask(sizes: Tensor, sp: int) -> Tensor:
    """model.py:121-124,96-101: 1.0 where a superpixel holds fewer than 25 pixels (spixel_sizes < 25 / sp^2), else 0.0; (N,L)."""
    return (sizes < 25.0 / (sp * sp)).to(sizes.dtype).flatten(1)


# --------------------------------------------------------------------------------------
# a9  k-means  (clusterkit.py:49-58, 31-46, 99-109, 112-208, 253-269)
# --------------------------------------------------------------------------------------


def kmeans_init_indices(n_images: int, n_tokens: int, k: int, rng=None) -> np.ndarray:
    """(n_images,k) int64: np.random.choice(L,K,replace=False) per image, consumed in image order
    from NumPy's legacy global RandomState (clusterkit.py:107; seeded once per run, inference.py:58)."""
    rng = np.random if rng is None else rng
    return np.stack([rng.choice(n_tokens, k, replace=False) for _ in range(n_images)]).astype(np.int64)


def kmeans_one(x: Tensor, init_idx, k: int, iter_limit: int = 20, tol: float = 1e-4,
               fallback_rows: Optional[List[int]] = None) -> Tuple[Tensor, int, int]:
    """Lloyd k-means on x (L,C).  Returns (assignment (L,) int64, passes, empty-cluster events).

    Assignment comes from the last distance pass, i.e. against the centroids *before* the last
    update.  An empty cluster takes one random row of x: torch.randint on the CPU generator in
    the reference (clusterkit.py:181-182); `fallback_rows` supplies those draws explicitly so
    the HIP path and the oracle can be fed the same sequence.
    """
    x = x.float()
    cent = x[torch.as_tensor(np.asarray(init_idx), dtype=torch.long)].clone()
    passes, events = 0, 0
    while True:
        dist = ((x[:, None, :] - cent[None, :, :]) ** 2.0).sum(dim=-1)
        assign = torch.argmin(dist, dim=1)  # first minimum
        prev = cent.clone()
        for j in range(k):
            members = x[assign == j]
            if members.shape[0] == 0:
                row = fallback_rows[events] if fallback_rows is not None else int(torch.randint(len(x), (1,)))
                members = x[row:row + 1]
                events += 1
            cent[j] = members.mean(dim=0)
        shift = torch.sqrt(((cent - prev) ** 2).sum(dim=1)).sum()
        passes += 1
        if shift ** 2 < tol or passes >= iter_limit:
            return assign, passes, events


# --------------------------------------------------------------------------------------
# a8  anchors  (anchor_gen.py:92-107, basic.py:42-47)
# --------------------------------------------------------------------------------------


def anchors_from_clusters(assign: Tensor, sizes: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """assign (N,L) int64, sizes (N,L) fp32 -> anchor token per cluster (N,K) int64, hint_mask (N,L).

    anchor_k = first argmax_t( [assign[t]==k] + sizes[t]*0.01 ); hint_mask[t] = #{k: anchor_k == t}.
    """
    n, l = assign.shape
    onehot = (assign[:, None, :] == torch.arange(k)[None, :, None]).float()  # (N,K,L)
    score = onehot + sizes[:, None, :] * 0.01
    anchor = torch.argmax(score, dim=-1)
    mask = torch.zeros(n, l)
    mask.scatter_add_(1, anchor, torch.ones(n, k))
    return anchor, mask


def random_anchor_mask(n: int, h: int, w: int, k: int, rng=None) -> Tensor:
    """random_hint mode: K distinct tokens per image from Python's `random` module."""
    rng = _pyrandom if rng is None else rng
    m = np.zeros((n, h * w), np.float32)
    for i in range(n):
        m[i, rng.sample(range(0, h * w), rng.randint(k, k))] = 1
    return torch.from_numpy(m.reshape(n, 1, h, w))


# --------------------------------------------------------------------------------------
# a10/a11  anchor colours and labels  (anchor_gen.py:54-90, basic.py:177-194, model.py:166)
# --------------------------------------------------------------------------------------


def sample_anchor_colors(prob: Tensor, q_to_ab: Tensor, t: int) -> Tensor:
    """prob (N,313,h,w) softmax -> ab/110 (N,2,h,w) at every token.

    Stable descending sort; candidates = top-10 bins.  t=0: top-1.  t=1: candidate farthest
    (L2 in ab/110) from top-1.  t=2: candidate maximising dist-to-top1 + dist-to-(t=1 pick).
    Ties resolve to the earlier candidate (stable sort of the distances, descending).
    """
    n, c, h, w = prob.shape
    order = torch.sort(prob, dim=1, descending=True, stable=True)[1][:, :10]      # (N,10,h,w)
    cand = q_to_ab[order.reshape(-1)].reshape(n, 10, h, w, 2) / 110.0
    if t == 0:
        pick = cand[:, 0]
    else:
        d1 = torch.linalg.vector_norm(cand - cand[:, :1], dim=-1)                   # (N,10,h,w)
        j1 = torch.sort(d1, dim=1, descending=True, stable=True)[1][:, :1]
        ab1 = torch.gather(cand, 1, j1[..., None].expand(-1, -1, -1, -1, 2))
        if t == 1:
            pick = ab1[:, 0]
        else:
            d2 = torch.linalg.vector_norm(cand - ab1, dim=-1)
            j2 = torch.sort(d1 + d2, dim=1, descending=True, stable=True)[1][:, t - 2:t - 1]
            pick = torch.gather(cand, 1, j2[..., None].expand(-1, -1, -1, -1, 2))[:, 0]
    return pick.permute(0, 3, 1, 2).contiguous()


def encode_ab2ind(ab: Tensor, q_to_ab: Tensor, neighbours: int = 5, sigma: float = 5.0) -> Tensor:
    """Soft 313-bin encoding: gaussian weights on the 5 nearest bins, normalised  (basic.py:177-194)."""
    n, _, h, w = ab.shape
    pts = (ab * 110.0).permute(1, 0, 2, 3).reshape(2, -1).t()                        # (m,2)
    dist = torch.cdist(q_to_ab, pts)                                                 # (313,m)
    nn_idx = dist.argsort(dim=0)[:neighbours]                                        # (5,m)
    sq = ((q_to_ab[nn_idx] - pts[None]) ** 2).sum(-1)                                # (5,m)
    g = (1.0 / (2 * math.pi * sigma)) * torch.exp(-sq / (2 * sigma ** 2))
    g = g / g.sum(dim=0, keepdim=True)
    q = ab.new_zeros(q_to_ab.shape[0], pts.shape[0])
    q.scatter_(0, nn_idx, g)
    return q.reshape(-1, n, h, w).permute(1, 0, 2, 3)


def color_labels(ab: Tensor, q_to_ab: Tensor) -> Tensor:
    """argmax over the soft encoding = nearest gamut bin (model.py:120,166) -> (N,1,h,w) int64."""
    return torch.max(encode_ab2ind(ab, q_to_ab), dim=1, keepdim=True)[1]


def decode_ind2ab(logit: Tensor, q_to_ab: Tensor, t=0) -> Tensor:
    """Integer t: t-th most probable bin centre / 110 (basic.py:196-209; inference.py:114).
    Non-integer t (the reference's default 0.38): annealed mean over exp(softmax/t) (basic.py:210-217)."""
    prob = torch.softmax(logit, dim=1)
    if t % 1 != 0:
        e = torch.exp(prob / t)
        e = e / e.sum(dim=1, keepdim=True)
        a = torch.tensordot(e, q_to_ab[:, 0], dims=((1,), (0,))).unsqueeze(1)
        b = torch.tensordot(e, q_to_ab[:, 1], dims=((1,), (0,))).unsqueeze(1)
        return torch.cat((a, b), dim=1) / 110.0
    order = torch.sort(prob, dim=1, descending=True, stable=True)[1][:, int(t)]
    return (q_to_ab[order] / 110.0).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------
# §8f row 1  colour space  (basic.py:395-475: rgb2xyz, xyz2lab, lab2xyz, xyz2rgb, rgb2lab, lab2rgb)
# --------------------------------------------------------------------------------------

_WHITE = (0.95047, 1.0, 1.08883)


def rgb2lab(rgb: Tensor) -> Tensor:
    """rgb (N,3,H,W) in [0,1] -> ((L-50)/50, a/110, b/110).  sRGB gamma 2.4, D65 white."""
    lin = torch.where(rgb > 0.04045, ((rgb + 0.055) / 1.055) ** 2.4, rgb / 12.92)
    r, g, b = lin[:, 0], lin[:, 1], lin[:, 2]
    xyz = torch.stack((0.412453 * r + 0.357580 * g + 0.180423 * b,
                       0.212671 * r + 0.715160 * g + 0.072169 * b,
                       0.019334 * r + 0.119193 * g + 0.950227 * b), 1)
    t = xyz / torch.tensor(_WHITE).view(1, 3, 1, 1)
    f = torch.where(t > 0.008856, t ** (1 / 3.0), 7.787 * t + 16.0 / 116.0)
    lab = torch.stack((116.0 * f[:, 1] - 16.0, 500.0 * (f[:, 0] - f[:, 1]), 200.0 * (f[:, 1] - f[:, 2])), 1)
    return torch.cat(((lab[:, :1] - 50.0) / 50.0, lab[:, 1:] / 110.0), 1)


def lab2rgb(lab_rs: Tensor) -> Tensor:
    """inverse of rgb2lab; negative linear RGB clamps to 0 (basic.py:421)."""
    L, a, b = lab_rs[:, 0] * 50.0 + 50.0, lab_rs[:, 1] * 110.0, lab_rs[:, 2] * 110.0
    fy = (L + 16.0) / 116.0
    f = torch.stack((a / 500.0 + fy, fy, torch.clamp(fy - b / 200.0, min=0.0)), 1)
    t = torch.where(f > 0.2068966, f ** 3.0, (f - 16.0 / 116.0) / 7.787) * torch.tensor(_WHITE).view(1, 3, 1, 1)
    x, y, z = t[:, 0], t[:, 1], t[:, 2]
    lin = torch.stack((3.24048134 * x - 1.53715152 * y - 0.49853633 * z,
                       -0.96925495 * x + 1.87599 * y + 0.04155593 * z,
                       0.05564664 * x - 0.20404134 * y + 1.05731107 * z), 1).clamp(min=0.0)
    return torch.where(lin > 0.0031308, 1.055 * lin ** (1.0 / 2.4) - 0.055, 12.92 * lin)


# --------------------------------------------------------------------------------------
# the image I/O either side of the forward and the anchor overlay
# --------------------------------------------------------------------------------------


def dilate_seeds(gate: Tensor, kernel_size: int = 3) -> Tensor:
    """basic.py:119-126: unfold(k, padding=k//2) -> max over the window -> fold(1) = k x k max filter, zero padded."""
    pad = kernel_size // 2
    return F.max_pool2d(F.pad(gate, (pad, pad, pad, pad), value=0.0), kernel_size, stride=1)


def mark_color_hints(gray: Tensor, target_ab: Tensor, gate: Tensor, kernel_size: int = 3, base_ab: Tensor = None) -> Tensor:
    """basic.py:95-117: anchors (gate > 0.7) keep target colours in a k x k centre, get a 1-px white colourless margin."""
    binary = (gate > 0.7).float()
    center = dilate_seeds(binary, kernel_size)
    margin = dilate_seeds(binary, kernel_size + 2) - center
    marked_gray = torch.where(margin > 1e-5, torch.ones_like(gate), gray)
    if base_ab is None:
        marked_ab = torch.where(center < 1e-5, torch.zeros_like(gate), target_ab)
    else:
        marked_ab = torch.where(margin > 1e-5, torch.zeros_like(gate), base_ab)
        marked_ab = torch.where(center > 1e-5, target_ab, marked_ab)
    return torch.cat((marked_gray, marked_ab), dim=1)


def cv2_resize_linear_u8(img: np.ndarray, dst_h: int, dst_w: int) -> np.ndarray:
    """cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR) for uint8 (H,W,C) images - a restatement of the
    published algorithm of the third-party dependency opencv-python==4.6.0.66 (environment.yaml:89; cv2 is not available
    offline), modules/imgproc/src/resize.cpp:
      * cv::resize: an exact 2x downscale in both directions is redirected to the area path (resizeAreaFast_, 8U:
        (a + b + c + d + 2) >> 2);
      * otherwise resizeGeneric_ with HResizeLinear / VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>:
        per axis f = (float)((d + 0.5) * scale - 0.5), s = cvFloor(f), f -= s, coefficients saturate_cast<short>((1-f, f) * 2048)
        (round half to even), scale = 1 / (dst / src) in double.  The two axes treat the image border differently: the x loop of
        cv::resize clamps (s < 0 -> f = 0, s = 0; s >= n-1 -> f = 0, s = n-1); the y loop keeps f and the invoker clips the two ROW
        indices to [0, n-1] instead (clip(sy + k, 0, ssize.height)), so at the top / bottom border both rows are the same row,
        still weighted b0 and b1 - and ((b0 r) >> 16) + ((b1 r) >> 16) can be one less than (2048 r) >> 16 (vertical upscales);
        rows: D = S[s] a0 + S[s+1] a1 (int32); columns: dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2.
    Pinned by hand-computed vectors and a <= 1 LSB cross-check against F.interpolate(bilinear, align_corners=False) in
    tests/test_oracle_golden.py (cv2's own output cannot be generated here: "parity unpinned" against the binary)."""
    H, W = img.shape[:2]
    src = img.astype(np.int64)
    if H == 2 * dst_h and W == 2 * dst_w:
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def axis(n_src, n_dst, clamp_f):
        scale = 1.0 / (float(n_dst) / float(n_src))
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp_f:             # the x loop of cv::resize
            lo = s < 0
            f[lo] = 0; s[lo] = 0
            hi = s >= n_src - 1
            f[hi] = 0; s[hi] = n_src - 1
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1), c0, c1

    x0, x1, a0, a1 = axis(W, dst_w, True)
    y0, y1, b0, b1 = axis(H, dst_h, False)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # (H, dst_w, C) int
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def fetch_from_rgb8(rgb8: np.ndarray, org_size: bool = True, psize: int = 256):
    """main/colorizer/inference.py:23-42 after cv2.imread/cvtColor, for one uint8 RGB image (H,W,3).
    org_size=False (what inference.py's CLI passes unless --no_resize, :32-33, :101): cv2.resize to psize x psize, INTER_LINEAR (cv2_resize_linear_u8).
    org_size=True (--no_resize, :28-31): the pad-to-16 quirk (BOTH dims get `16 - dim % 16` rows/columns, i.e. a full 16
    when only the other one is off).  Then /255 in float64 -> float32, RGB->Lab (the reference's torch rgb2lab stands in
    for cv2's float COLOR_RGB2LAB, see color.hip), gray = (L-50)/50, ab/110, rgb*2-1.
    Returns (gray (1,1,Hp,Wp), ab (1,2,Hp,Wp), rgb (1,3,Hp,Wp), (H,W))."""
    H, W = rgb8.shape[:2]           # the ORIGINAL size in both branches (inference.py:26,42); batch_depadding (:138-139) ignores it when resized
    if not org_size:
        rgb8 = cv2_resize_linear_u8(rgb8, psize, psize)
    elif H % 16 != 0 or W % 16 != 0:
        rgb8 = np.pad(rgb8, ((0, 16 - H % 16), (0, 16 - W % 16), (0, 0)), mode="edge")
    rgb = np.array(rgb8 / 255.0, np.float32)
    rgb_t = torch.from_numpy(rgb.transpose((2, 0, 1)))[None]
    lab = rgb2lab(rgb_t)
    return lab[:, 0:1], lab[:, 1:3], rgb_t * 2.0 - 1.0, (H, W)


def labs_to_rgb8(lab_rs: Tensor, H: int, W: int) -> np.ndarray:
    """utils/util.py:91-106 + batch_depadding (inference.py:44-49): normalised Lab (N,3,Hp,Wp) -> RGB ->
    (rgb*255).astype(uint8) of the top-left H x W crop, (N,H,W,3); values above 1 saturate (cv2 clips there)."""
    rgb = lab2rgb(lab_rs)[:, :, :H, :W]
    return (rgb * 255.0).clamp(max=255.0).permute(0, 2, 3, 1).numpy().astype(np.uint8)


# --------------------------------------------------------------------------------------
# a14  the forward  (model.py:103-199, test_mode=True, enhanced=True, dense pos)
# --------------------------------------------------------------------------------------


class DiscoOracle:
    """Holds a checkpoint `state_dict` and replays AnchorColorProb.forward on the CPU."""

    def __init__(self, state_dict: SD, q_to_ab: np.ndarray, sp_size: int = 16, n_clusters: int = 8,
                 random_hint: bool = False, hint2regress: bool = False, spix_pos: bool = False, use_mask: bool = False):
        self.sd = {k: v.detach().clone() for k, v in state_dict.items()}
        self.q_to_ab = torch.as_tensor(np.asarray(q_to_ab), dtype=torch.float32)
        self.sp = sp_size
        self.k = n_clusters
        self.random_hint = random_hint
        self.hint2regress = hint2regress      # model.py:63-64,177-181,188
        self.spix_pos = spix_pos              # model.py:106-112
        self.use_mask = use_mask              # model.py:38,124

    # -- stages ---------------------------------------------------------------------
    def tokens(self, gray: Tensor, ab: Tensor, observer=None, taps=None):
        """steps 1-6 of SURVEY §3.2: affinity, pooled tokens, colours, sizes, pos."""
        aff = segnet_forward(self.sd, gray, observer)
        feats = repnet_forward(self.sd, gray, observer, taps)
        if self.spix_pos:     # the per-pixel encoding is pooled like a feature: one position sequence per image
            full_pos = position_encoding(gray.shape[2], gray.shape[3])[None].expand(gray.shape[0], -1, -1, -1)
            pooled, _ = poolfeat(torch.cat((feats, ab, full_pos), 1), aff, self.sp)
            tok, spix_ab = pooled[:, :64], pooled[:, 64:66]
            pos = pooled[:, 66:].flatten(2).transpose(1, 2)                            # (N,L,64)
        else:
            pooled, _ = poolfeat(torch.cat((feats, ab), 1), aff, self.sp)
            tok, spix_ab = pooled[:, :64], pooled[:, 64:]
        sizes = spixel_size(aff, self.sp)
        n, _, h, w = tok.shape
        if not self.spix_pos:
            pos = position_encoding(h, w).flatten(1).t()[None].expand(n, -1, -1)      # (N,L,64)
        src = tok.flatten(2).transpose(1, 2)                                         # (N,L,64), t=y*w+x
        return aff, feats, src, pos, spix_ab, sizes

    def anchors(self, enc: Tensor, sizes: Tensor, init_idx=None, fallback_rows=None, hint_mask=None):
        """step 8: k-means on the encoder output + per-cluster anchor, or random hints."""
        n, l, _ = enc.shape
        info = {"passes": [], "events": [], "assign": None, "anchor": None, "init_idx": None}
        if self.random_hint:
            return hint_mask.reshape(n, l).float(), info
        if init_idx is None:
            init_idx = kmeans_init_indices(n, l, self.k)
        assign = []
        for i in range(n):
            fb = None if fallback_rows is None else list(fallback_rows[i])
            a, passes, events = kmeans_one(enc[i], init_idx[i], self.k, fallback_rows=fb)
            assign.append(a); info["passes"].append(passes); info["events"].append(events)
        assign = torch.stack(assign)
        anchor, mask = anchors_from_clusters(assign, sizes.reshape(n, l), self.k)
        info.update(assign=assign, anchor=anchor, init_idx=np.asarray(init_idx))
        return mask, info

    def hint_tokens(self, src: Tensor, labels: Tensor, mask: Tensor, colors: Tensor = None) -> Tensor:
        """trg_word_emb(cat[src, mask*onehot313(label), mask])  (model.py:175,183-185);
        hint2regress: trg_word_emb(cat[src, mask*ab, mask]) with the anchors' ab values (model.py:177-180)."""
        m = mask[..., None]
        if self.hint2regress:
            gt = colors.flatten(2).transpose(1, 2)                                     # (N,L,2)
            return F.linear(torch.cat((src, m * gt, m), dim=-1), self.sd["trg_word_emb.weight"])
        onehot = F.one_hot(labels, 313).float()
        return F.linear(torch.cat((src, m * onehot, m), dim=-1), self.sd["trg_word_emb.weight"])

    # -- full forward ---------------------------------------------------------------
    @torch.no_grad()
    def forward(self, gray: Tensor, ab: Tensor, sampled_T: int = 0, init_idx=None, fallback_rows=None,
                hint_mask=None, observer=None, return_info: bool = False, test_mode: bool = True):
        """Returns the reference's 6-tuple (pal_logit, ref_logit, pred_colors, affinity_map,
        spix_colors, hint_mask); with return_info also a dict of intermediates.
        test_mode=False is the validation forward (train_colorizer.py:206 under model.eval(); model.py:169-171):
        anchors from k-means on the pooled GT colours, GT token labels, sampled_T ignored."""
        sd = self.sd
        n0 = gray.shape[0]
        aff, feats, src, pos, spix_ab, sizes = self.tokens(gray, ab, observer)
        n, l, _ = src.shape
        h, w = gray.shape[2] // self.sp, gray.shape[3] // self.sp
        to_map = lambda t: t.transpose(1, 2).reshape(t.shape[0], -1, h, w)
        pad = entry_mask(sizes, self.sp) if self.use_mask else None       # the same mask for both stacks (model.py:124-125)
        enc = encoder_stack(sd, "wildpath", src, pos, key_bias=pad)
        pal_logit = to_map(F.linear(enc, sd["mid_word_prj.weight"]))
        if self.random_hint and hint_mask is None:
            hint_mask = random_anchor_mask(n, h, w, self.k)
        if not test_mode and self.hint2regress:
            raise NameError("name 'spix_color' is not defined (models/model.py:178)")
        cluster_on = enc if test_mode else spix_ab.flatten(2).transpose(1, 2)        # (N,L,64) | (N,L,2)
        mask, info = self.anchors(cluster_on, sizes, init_idx, fallback_rows, hint_mask)
        prob = torch.softmax(pal_logit, dim=1)
        if not test_mode or sampled_T < 0:   # ground-truth anchor colours (model.py:145-147) / validation forward
            colors = spix_ab
        elif sampled_T > 0:        # diverse: three variants of a single image (model.py:148-159)
            if n != 1:
                raise RuntimeError("diverse sampling is defined for N=1 only (reference expand() fails for N>1)")
            colors = torch.cat([sample_anchor_colors(prob, self.q_to_ab, t) for t in (0, 1, 2)], 0)
            gray, aff, src, pos, mask = (t.expand(3, *t.shape[1:]) for t in (gray, aff, src, pos, mask))
            n = 3       # (the reference does NOT expand src_pad_mask: with use_mask its hint path fails for --diverse; neither is combined here)
            if pad is not None:
                raise RuntimeError("use_mask with diverse sampling: the reference's (1,L) key_padding_mask does not match its batch of 3")
        else:
            colors = sample_anchor_colors(prob, self.q_to_ab, 0)
        labels = color_labels(colors, self.q_to_ab).reshape(n, l)
        hint = self.hint_tokens(src, labels, mask, colors)
        dec = encoder_stack(sd, "hintpath", hint, pos, key_bias=pad)
        ref_logit = to_map(F.linear(dec, sd["trg_word_prj.weight"]))
        full = upfeat(to_map(dec), aff, self.sp)
        pre = enhance_forward(sd, torch.cat((gray, full), 1), observer)
        pred = torch.tanh(pre)
        out = (pal_logit, ref_logit, pred, aff, colors, mask.reshape(n, 1, h, w))
        if return_info:
            info.update(feats=feats, src=src, pos=pos, sizes=sizes, enc=enc, labels=labels, dec=dec,
                        full=full, pre_tanh=pre, spix_ab=spix_ab)
            return out, info
        return out
