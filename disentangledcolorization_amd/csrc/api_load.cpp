// api_load.cpp - context life cycle and checkpoint hand-over: strict layout check (utils_train.py:151), spectral-norm / batch-norm folding,
// hi/lo/q splits and MFMA fragment packing of every conv layer, the HourGlass2 channel levelling and its fp8 fallback (split out of api.cpp, round 6).
#include "ctx.h"

namespace disco_api {


int dev_alloc(disco_ctx* c, size_t bytes, void** out) {
    DISCO_HIP_CHECK(hipMalloc(out, bytes ? bytes : 16));
    c->allocs.push_back(*out);
    return DISCO_OK;
}
int upload(disco_ctx* c, const void* h, size_t bytes, void** out) {
    int rc = dev_alloc(c, bytes, out);
    if (rc) return rc;
    DISCO_HIP_CHECK(hipMemcpy(*out, h, bytes, hipMemcpyHostToDevice));
    return DISCO_OK;
}

const HostTensor& T(disco_ctx* c, const std::string& k) { return c->sd.at(k); }

// asynchronous host->device copy of a small array through the context's pinned staging ring
int staged_h2d(disco_ctx* c, void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
    disco_ctx::Staging& g = c->stg[c->stg_next];
    c->stg_next = (c->stg_next + 1) % 4;
    if (g.used) DISCO_HIP_CHECK(hipEventSynchronize(g.ev));       // its previous copy has been consumed
    if (g.cap < bytes) {
        if (g.h) DISCO_HIP_CHECK(hipHostFree(g.h));
        g.h = nullptr; g.cap = 0;
        DISCO_HIP_CHECK(hipHostMalloc(&g.h, bytes, hipHostMallocDefault));
        g.cap = bytes;
    }
    if (!g.ev) DISCO_HIP_CHECK(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
    memcpy(g.h, h_src, bytes);
    DISCO_HIP_CHECK(hipMemcpyAsync(d_dst, g.h, bytes, hipMemcpyHostToDevice, s));
    DISCO_HIP_CHECK(hipEventRecord(g.ev, s));
    g.used = true;
    return DISCO_OK;
}

// f16x3 layers run on conv3x3_mx_kernel (AR = 2)

// effective conv weight (c_out, c_in, 3, 3): plain `.weight`, or spectral-norm weight_orig / (u . (W v))
std::vector<float> eff_weight(disco_ctx* c, const std::string& key) {
    auto it = c->sd.find(key + ".weight");
    if (it != c->sd.end()) return it->second.data;
    const HostTensor& w = T(c, key + ".weight_orig");
    const std::vector<float>& u = T(c, key + ".weight_u").data;
    const std::vector<float>& v = T(c, key + ".weight_v").data;
    const size_t co = (size_t)w.shape[0], k = w.numel() / co;
    double sigma = 0.0;
    for (size_t o = 0; o < co; ++o) {
        double s = 0.0;
        for (size_t i = 0; i < k; ++i) s += (double)w.data[o * k + i] * (double)v[i];
        sigma += (double)u[o] * s;
    }
    const float sg = (float)sigma;
    std::vector<float> out(w.data.size());
    for (size_t i = 0; i < out.size(); ++i) out[i] = w.data[i] / sg;
    return out;
}

// eval BatchNorm as y = x*scale + shift
void bn_affine(disco_ctx* c, const std::string& key, std::vector<float>& scale, std::vector<float>& shift) {
    const auto& g = T(c, key + ".weight").data; const auto& b = T(c, key + ".bias").data;
    const auto& m = T(c, key + ".running_mean").data; const auto& v = T(c, key + ".running_var").data;
    scale.resize(g.size()); shift.resize(g.size());
    for (size_t i = 0; i < g.size(); ++i) {
        scale[i] = g[i] / std::sqrt(v[i] + 1e-5f);
        shift[i] = b[i] - m[i] * scale[i];
    }
}

// Which layers run on conv3x3_mx_kernel.  DISCO_PREC_MX8: the enhanceNet only - everything downstream of the anchors.  The
// stacks that feed k-means (segnet -> affinity -> pooling / sizes, repnet -> tokens) keep the f16x3 kernel: anchors are a
// discrete decision, and the ~3e-5 perturbation of the fp8-corrected arithmetic at the encoder output flipped them in 1 of
// 108 images against the fp32 oracle (tools/anchor_stability.py, profiles/r02_anchor_stability.txt), f16x3 in none.
// DISCO_PREC_MX8_ALL runs every layer on the mx kernel (measurements only: not anchor-safe).
// DISCO_PREC_X2Q: as MX8, and the ColorProbNet on the kernel's second arithmetic (f16x2 + fp8: both fp16 products of the hi
// plane, only the activation residual through fp8 - conv_mx.hip), 5 pipe units instead of 6.
// DISCO_PREC_MX6 (the default) and DISCO_PREC_X2Q: the enhanceNet on f16 + fp6x2 - the same two correction products with fp6 e2m3 operands,
// which the K = 64 MFMA runs in half the passes - except its first layer, whose sources (upfeat, gray) are written by kernels that
// produce fp8 planes: it reads those and writes fp6 ones.
bool any_mx(const disco_ctx* c) { return c->opt.precision == DISCO_PREC_MX8 || c->opt.precision == DISCO_PREC_MX8_ALL || c->opt.precision == DISCO_PREC_X2Q || c->opt.precision == DISCO_PREC_MX6; }
int arith_of(const disco_ctx* c, const std::string& key) {
    if (c->opt.precision == DISCO_PREC_MX8_ALL) return ARITH_MX8;
    if (c->opt.precision != DISCO_PREC_MX8 && c->opt.precision != DISCO_PREC_X2Q && c->opt.precision != DISCO_PREC_MX6) return ARITH_F16X3;
    if (key.compare(0, 11, "enhanceNet.") == 0) {
        if (c->opt.precision == DISCO_PREC_MX8 || c->enhance_fp8_fallback) return ARITH_MX8;
        return key == "enhanceNet.inConv.inConv.0" ? ARITH_MX8 : ARITH_MX6;
    }
    if (c->opt.precision == DISCO_PREC_X2Q && key.compare(0, 7, "repnet.") == 0) return ARITH_X2Q;
    return ARITH_F16X3;
}
bool use_mx(const disco_ctx* c, const std::string& key) { return arith_of(c, key) != ARITH_F16X3; }
int pad_cout_mx(int co) { return co <= 32 ? 32 : round_up(co, 64); }

// upload bias / BN affine padded to `n` channels (bias 0, scale 1, shift 0 beyond the real ones)
int upload_padded(disco_ctx* c, std::vector<float> v, size_t n, float fill, float** out) {
    v.resize(std::max(v.size(), n), fill);
    return upload_vec(c, v, out);
}

// Pack and upload the weights of an mx layer.  w: (co, ci, 3, 3) effective weights; ci_map / c_in_pad describe the packed
// input channels (multiples of 32 per source); act_out: the layer writes an activation tensor, so its output channels are
// padded to whole blocks with zero weights (fp32 NCHW outputs keep their real channel count).
int finish_mx(disco_ctx* c, ConvLayer& L, const std::vector<float>& w, int co, int ci, const int* ci_map, int c_in_pad, bool act_out, int x2q = 0) {
    L.mx = true; L.x2q = x2q; L.c_in = ci; L.c_in_pad = c_in_pad;
    L.c_out_k = act_out ? pad_cout_mx(co) : co;
    if (x2q == 1 && c_in_pad % 64) { set_error("the f16x2+fp8 arithmetic needs a multiple of 64 input channels (%d)", c_in_pad); return DISCO_ESHAPE; }
    std::vector<char> packed(conv_mx_packed_bytes(L.c_out_k, c_in_pad, x2q));
    std::vector<int32_t> wexp((size_t)round_up(L.c_out_k, 32));
    conv_mx_pack_host(w.data(), co, ci, ci_map, c_in_pad, packed.data(), wexp.data(), x2q);     // rows >= co pack as zeros
    int rc = upload(c, packed.data(), packed.size(), (void**)&L.d_w);
    if (rc) return rc;
    return upload_vec(c, wexp, &L.d_wexp);
}

// Build one MFMA conv layer.  fold_bn: BN directly after the conv (SpixelNet, network.py:240-246) is folded into
// weights+bias; post_bn: BN after the activation (ColorProbNet / HourGlass2 blocks) becomes the epilogue affine.
// ci_map / c_in_pad_override describe the packed input channels when they are not simply the reference's (concat of padded
// sources, permuted inputs); act_out = false for layers whose only output is fp32 NCHW (pred_mask0, outConv).
int make_conv(disco_ctx* c, const std::string& key, const std::string& fold_bn, const std::string& post_bn,
              const std::vector<int>* ci_map = nullptr, int c_in_pad_override = 0, bool stride2 = false, bool act_out = true) {
    std::vector<float> w = eff_weight(c, key);
    const HostTensor& ws = c->sd.count(key + ".weight") ? T(c, key + ".weight") : T(c, key + ".weight_orig");
    const int co = (int)ws.shape[0], ci = (int)ws.shape[1];
    std::vector<float> bias(co, 0.f);
    if (c->sd.count(key + ".bias")) bias = T(c, key + ".bias").data;
    if (!fold_bn.empty()) {
        std::vector<float> sc, sh;
        bn_affine(c, fold_bn, sc, sh);
        for (int o = 0; o < co; ++o) {
            for (int i = 0; i < ci * 9; ++i) w[(size_t)o * ci * 9 + i] *= sc[o];
            bias[o] = bias[o] * sc[o] + sh[o];
        }
    }
    std::vector<float> post_sc, post_sh;
    if (!post_bn.empty()) bn_affine(c, post_bn, post_sc, post_sh);
    {   // channel equalisation (exact: powers of two): input columns, then the output side - through the BN affine when the layer has one
        // behind its activation (x 2^k commutes with ReLU / LeakyReLU), else through the weight rows and the bias
        auto ei = c->eq_in.find(key), eo = c->eq_out.find(key);
        if (ei != c->eq_in.end() && (int)ei->second.size() == ci)
            for (int o = 0; o < co; ++o)
                for (int i = 0; i < ci; ++i)
                    for (int t = 0; t < 9; ++t) w[((size_t)o * ci + i) * 9 + t] *= ei->second[i];
        if (eo != c->eq_out.end() && (int)eo->second.size() == co)
            for (int o = 0; o < co; ++o) {
                if (!post_bn.empty()) { post_sc[o] *= eo->second[o]; post_sh[o] *= eo->second[o]; }
                else { for (int i = 0; i < ci * 9; ++i) w[(size_t)o * ci * 9 + i] *= eo->second[o]; bias[o] *= eo->second[o]; }
            }
    }
    ConvLayer L;
    L.c_in = ci; L.c_out = co; L.c_real = co;
    int rc;
    if (use_mx(c, key)) {
        const int x2q = arith_of(c, key) == ARITH_X2Q ? 1 : (arith_of(c, key) == ARITH_MX6 ? 2 : 0);       // pack variant
        const int cpad = c_in_pad_override ? c_in_pad_override : round_up(ci, x2q == 1 ? 64 : 32);
        if ((rc = finish_mx(c, L, w, co, ci, ci_map ? ci_map->data() : nullptr, cpad, act_out, x2q))) return rc;
        if ((rc = upload_padded(c, bias, (size_t)L.c_out_k, 0.f, &L.d_bias))) return rc;
        if (!post_bn.empty()) {
            if ((rc = upload_padded(c, post_sc, (size_t)L.c_out_k, 1.f, &L.d_bn_scale))) return rc;
            if ((rc = upload_padded(c, post_sh, (size_t)L.c_out_k, 0.f, &L.d_bn_shift))) return rc;
        }
        c->conv[key] = L;
        return DISCO_OK;
    }
    L.c_in_pad = c_in_pad_override ? c_in_pad_override : round_up(ci, 16);
    // (Stride-2 layers keep the plain stride-2 tiles: a space-to-depth packing was measured in round 1 - same LDS footprint and MFMA/LDS
    // ratio as stride 1, but each phase chunk uses half of every 128-byte line it fetches and these layers are bound by L2->LDS line
    // traffic: 0.66 vs 0.65 ms on 64->128@256^2, profiles/r01_conv_s2d_timeline.txt - and removed in round 3.)
    std::vector<char> packed(conv3x3_packed_bytes(co, L.c_in_pad));
    conv3x3_pack_host(w.data(), co, ci, ci_map ? ci_map->data() : nullptr, L.c_in_pad, packed.data());
    rc = upload(c, packed.data(), packed.size(), (void**)&L.d_w);
    if (rc) return rc;
    if ((rc = upload_vec(c, bias, &L.d_bias))) return rc;
    if (!post_bn.empty()) {
        if ((rc = upload_vec(c, post_sc, &L.d_bn_scale))) return rc;
        if ((rc = upload_vec(c, post_sh, &L.d_bn_shift))) return rc;
    }
    c->conv[key] = L;
    return DISCO_OK;
}

int make_c1(disco_ctx* c, const std::string& key, const std::string& fold_bn) {
    std::vector<float> w = eff_weight(c, key);  // (co,1,3,3) == (co,9)
    const HostTensor& ws = c->sd.count(key + ".weight") ? T(c, key + ".weight") : T(c, key + ".weight_orig");
    const int co = (int)ws.shape[0];
    std::vector<float> bias(co, 0.f);
    if (c->sd.count(key + ".bias")) bias = T(c, key + ".bias").data;
    if (!fold_bn.empty()) {
        std::vector<float> sc, sh;
        bn_affine(c, fold_bn, sc, sh);
        for (int o = 0; o < co; ++o) { for (int i = 0; i < 9; ++i) w[o * 9 + i] *= sc[o]; bias[o] = bias[o] * sc[o] + sh[o]; }
    }
    DirectLayer L; L.c_in = 1; L.c_out = co;
    int rc = upload_vec(c, w, &L.d_w); if (rc) return rc;
    if ((rc = upload_vec(c, bias, &L.d_bias))) return rc;
    c->direct[key] = L;
    return DISCO_OK;
}


// 4-phase weights (4*co, ci, 3, 3), phase-major, of a depth-to-space layer -> the mx layer: every phase padded to whole
// 32-channel blocks (zero weights), bias repeated per phase by the kernel (parameters are indexed modulo the phase size)
int finish_phase_mx(disco_ctx* c, ConvLayer& L, const std::vector<float>& w4, int co, int ci, const std::vector<float>& bias, int x2q = 0) {
    const int cop = round_up(co, 32);
    std::vector<float> wp((size_t)4 * cop * ci * 9, 0.f);
    for (int ph = 0; ph < 4; ++ph)
        for (int o = 0; o < co; ++o)
            std::copy(w4.begin() + ((size_t)(ph * co + o)) * ci * 9, w4.begin() + ((size_t)(ph * co + o) + 1) * ci * 9,
                      wp.begin() + ((size_t)(ph * cop + o)) * ci * 9);
    int rc = finish_mx(c, L, wp, 4 * cop, ci, nullptr, round_up(ci, x2q == 1 ? 64 : 32), true, x2q);
    if (rc) return rc;
    L.c_out = 4 * cop; L.c_real = co;
    std::vector<uint32_t> mask(cdiv(L.c_out_k, 32));
    conv3x3_tapmask_host(wp.data(), 4 * cop, ci, mask.data());
    mask.resize(cdiv(L.c_out_k, 32), 0u);
    if ((rc = upload_vec(c, mask, &L.d_tapmask))) return rc;
    return upload_padded(c, bias, (size_t)cop, 0.f, &L.d_bias);
}

// ConvTranspose2d(4,s2,p1) as a 4-phase 3x3 conv on the MFMA kernel with a depth-to-space epilogue
int make_deconv(disco_ctx* c, const std::string& key) {
    const HostTensor& ws = T(c, key + ".weight");
    const int ci = (int)ws.shape[0], co = (int)ws.shape[1];
    std::vector<float> w3((size_t)4 * co * ci * 9);
    deconv_as_conv3x3_host(ws.data.data(), ci, co, w3.data());
    ConvLayer L;
    L.c_in = ci; L.c_out = 4 * co; L.c_real = co; L.c_in_pad = round_up(ci, 16); L.kind = 1;
    int rc;
    if (use_mx(c, key)) {
        if ((rc = finish_phase_mx(c, L, w3, co, ci, T(c, key + ".bias").data))) return rc;
        c->conv[key] = L;
        return DISCO_OK;
    }
    std::vector<char> packed(conv3x3_packed_bytes(L.c_out, L.c_in_pad));
    conv3x3_pack_host(w3.data(), L.c_out, ci, nullptr, L.c_in_pad, packed.data());
    rc = upload(c, packed.data(), packed.size(), (void**)&L.d_w); if (rc) return rc;
    std::vector<uint32_t> mask(cdiv(L.c_out, 32));
    conv3x3_tapmask_host(w3.data(), L.c_out, ci, mask.data());
    if ((rc = upload_vec(c, mask, &L.d_tapmask))) return rc;
    if ((rc = upload_vec(c, T(c, key + ".bias").data, &L.d_bias))) return rc;
    c->conv[key] = L;
    return DISCO_OK;
}

// nn.Upsample(x2, nearest) -> Conv2d 3x3 (network.py:187,195,199) as a 4-phase sub-pixel conv on the low-res
// input: 4 summed taps per phase instead of 9 (2.25x fewer MACs), depth-to-space epilogue
int make_upconv(disco_ctx* c, const std::string& key) {
    const HostTensor& ws = T(c, key + ".weight");
    const int co = (int)ws.shape[0], ci = (int)ws.shape[1];
    std::vector<float> w4((size_t)4 * co * ci * 9);
    upconv_as_conv3x3_host(ws.data.data(), ci, co, w4.data());
    ConvLayer L;
    L.c_in = ci; L.c_out = 4 * co; L.c_real = co; L.c_in_pad = round_up(ci, 16); L.kind = 2;
    int rc;
    if (use_mx(c, key)) {
        if ((rc = finish_phase_mx(c, L, w4, co, ci, T(c, key + ".bias").data, arith_of(c, key) == ARITH_X2Q ? 1 : (arith_of(c, key) == ARITH_MX6 ? 2 : 0)))) return rc;
        c->conv[key] = L;
        return DISCO_OK;
    }
    std::vector<char> packed(conv3x3_packed_bytes(L.c_out, L.c_in_pad));
    conv3x3_pack_host(w4.data(), L.c_out, ci, nullptr, L.c_in_pad, packed.data());
    rc = upload(c, packed.data(), packed.size(), (void**)&L.d_w); if (rc) return rc;
    std::vector<uint32_t> mask(cdiv(L.c_out, 32));
    conv3x3_tapmask_host(w4.data(), L.c_out, ci, mask.data());
    if ((rc = upload_vec(c, mask, &L.d_tapmask))) return rc;
    if ((rc = upload_vec(c, T(c, key + ".bias").data, &L.d_bias))) return rc;
    c->conv[key] = L;
    return DISCO_OK;
}

int make_encoder(disco_ctx* c, const std::string& path, float** out) {
    std::vector<float> w;
    w.reserve(ENC_LAYERS * ENC_LAYER_FLOATS);
    for (int l = 0; l < ENC_LAYERS; ++l) {
        const std::string q = path + ".layers." + std::to_string(l) + ".";
        for (const char* k : {"self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                              "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                              "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"}) {
            const auto& d = T(c, q + k).data;
            w.insert(w.end(), d.begin(), d.end());
        }
    }
    return upload_vec(c, w, out);
}

int get_pos(disco_ctx* c, int h, int w, float** out) {
    auto it = c->pos_cache.find({h, w});
    if (it != c->pos_cache.end()) { *out = it->second; return DISCO_OK; }
    std::vector<float> p((size_t)h * w * 64);
    position_encoding_host(p.data(), h, w);
    float* d = nullptr;
    int rc = upload_vec(c, p, &d);
    if (rc) return rc;
    c->pos_cache[{h, w}] = d;
    *out = d;
    return DISCO_OK;
}

// The HourGlass2's layers (network.py:125-144), packed for the arithmetic arith_of() currently assigns them: called once by disco_finalize and
// once more when the channel-disparity guard moves the stack from fp6 to fp8 corrections (the host weights "enhanceNet.*" stay in c->sd)
int make_enhance(disco_ctx* c) {
    int rc;
    const std::string en = "enhanceNet.";
    {   // input = cat(gray, 64 token features) in the reference; here source 0 = features, source 1 = 16-ch gray plane
        // (80 packed channels in every arithmetic: the f16+fp8x2 kernel takes the gray block as its H-only tail chunk with the gray
        // channel as (g_hi, g_lo, g_hi) against (w_h, w_h, w_l) - launch_gray_tail, conv_mx_pack_host)
        const int cp = 80;
        std::vector<int> map(cp, -1);
        for (int i = 0; i < 64; ++i) map[i] = i + 1;
        map[64] = 0;
        if (use_mx(c, en)) { map[65] = 0; map[66] = CONV_MX_LO_OF(0); }
        if ((rc = make_conv(c, en + "inConv.inConv.0", "", "", &map, cp))) return rc;
    }
    if ((rc = make_conv(c, en + "inConv.conv.0", "", en + "inConv.conv.2"))) return rc;
    for (const char* k : {"down1", "down2"}) {
        if ((rc = make_conv(c, en + k + ".conv.0", "", "", nullptr, 0, true))) return rc;       // down1 / down2: stride 2
        if ((rc = make_conv(c, en + k + ".conv.2", "", en + k + ".conv.4"))) return rc;
    }
    for (int r = 0; r < 3; ++r)
        for (const char* k : {"0", "1", "3"})
            if ((rc = make_conv(c, en + "residual." + std::to_string(r) + ".conv." + k, "", ""))) return rc;
    for (const char* k : {"up2", "up1"}) {
        if ((rc = make_conv(c, en + k + ".conv1", "", ""))) return rc;
        if ((rc = make_conv(c, en + k + ".combine", "", ""))) return rc;
        if ((rc = make_conv(c, en + k + ".conv2.0", "", ""))) return rc;
        if ((rc = make_conv(c, en + k + ".conv2.2", "", en + k + ".conv2.4"))) return rc;
    }
    if ((rc = make_conv(c, en + "outConv", "", "", nullptr, 0, false, false))) return rc;
    return DISCO_OK;
}

// Cross-layer channel equalisation of the HourGlass2 (round 4).  Every tensor between two of its convs has ONE producer - or, along the residual
// chain (y = relu(x + F(x)), network.py:45-47), one class of producers that must share their factors - and known consumers, so each channel c of a
// tensor can be multiplied by s_c = 2^k at its producer(s) and divided at its consumers: exact in fp32 (powers of two), ReLU / LeakyReLU commute
// with positive factors, and the tensors in between come out with level channels - which is what a format that shares one scale over 32
// channels of a pixel needs.  s_c lifts every channel's calibration maximum to within a factor 2 of the tensor's largest (never down; by 2^6 at most).
struct EqTensor { std::vector<std::string> producers; int channels; std::vector<std::pair<std::string, int>> consumers; };
const std::vector<EqTensor>& enhance_tensors() {
    static const std::vector<EqTensor> t = [] {
        const std::string en = "enhanceNet.";
        std::vector<EqTensor> v = {
            {{en + "inConv.inConv.0"}, 64, {{en + "inConv.conv.0", 0}}},
            {{en + "inConv.conv.0"}, 64, {{en + "down1.conv.0", 0}, {en + "up1.combine", 64}}},
            {{en + "down1.conv.0"}, 128, {{en + "down1.conv.2", 0}}},
            {{en + "down1.conv.2"}, 128, {{en + "down2.conv.0", 0}, {en + "up2.combine", 128}}},
            {{en + "down2.conv.0"}, 256, {{en + "down2.conv.2", 0}}},
            {{en + "down2.conv.2", en + "residual.0.conv.3", en + "residual.1.conv.3", en + "residual.2.conv.3"}, 256,
             {{en + "residual.0.conv.0", 0}, {en + "residual.1.conv.0", 0}, {en + "residual.2.conv.0", 0}, {en + "up2.conv1", 0}}},
            {{en + "up2.conv1"}, 128, {{en + "up2.combine", 0}}},
            {{en + "up2.combine"}, 128, {{en + "up2.conv2.0", 0}}},
            {{en + "up2.conv2.0"}, 128, {{en + "up2.conv2.2", 0}}},
            {{en + "up2.conv2.2"}, 128, {{en + "up1.conv1", 0}}},
            {{en + "up1.conv1"}, 64, {{en + "up1.combine", 0}}},
            {{en + "up1.combine"}, 64, {{en + "up1.conv2.0", 0}}},
            {{en + "up1.conv2.0"}, 64, {{en + "up1.conv2.2", 0}}},
            {{en + "up1.conv2.2"}, 64, {{en + "outConv", 0}}},
        };
        for (int r = 0; r < 3; ++r) {
            const std::string k = en + "residual." + std::to_string(r) + ".conv.";
            v.push_back({{k + "0"}, 256, {{k + "1", 0}}});
            v.push_back({{k + "1"}, 256, {{k + "3", 0}}});
        }
        return v;
    }();
    return t;
}
// fills c->eq_out / c->eq_in from the per-channel maxima of the last calibration pass; false when a tensor has not been measured
bool plan_equalisation(disco_ctx* c) {
    std::map<std::string, std::vector<float>> eo, ei;
    auto widths = [&](const std::string& key) -> int { auto it = c->conv.find(key); return it == c->conv.end() ? 0 : it->second.c_in; };
    for (const EqTensor& t : enhance_tensors()) {
        std::vector<float> a(t.channels, 0.f);
        for (const std::string& p : t.producers) {
            auto it = c->chan_amax.find(p);
            if (it == c->chan_amax.end() || (int)it->second.size() < t.channels) return false;
            for (int i = 0; i < t.channels; ++i) a[i] = std::max(a[i], it->second[i]);
        }
        const float top = *std::max_element(a.begin(), a.end());
        if (!(top > 0.f)) continue;
        std::vector<float> sc(t.channels, 1.f);
        for (int i = 0; i < t.channels; ++i)
            if (a[i] > 0.f) {
                int k = (int)std::floor(std::log2(top / a[i]));
                // (at most 2^6: a channel that is almost silent on the calibration images may be as loud as the others on real ones, and
                // a factor 2^6 then still leaves 2^5 of the tensor's 2^11 fp16 headroom)
                sc[i] = std::ldexp(1.f, std::min(std::max(k, 0), 6));
            }
        for (const std::string& p : t.producers) eo[p] = sc;
        for (const auto& cons : t.consumers) {
            const int ci = widths(cons.first);
            if (ci <= 0 || cons.second + t.channels > ci) return false;
            std::vector<float>& v = ei[cons.first];
            if (v.empty()) v.assign(ci, 1.f);
            for (int i = 0; i < t.channels; ++i) v[cons.second + i] = 1.f / sc[i];
        }
    }
    c->eq_out = eo; c->eq_in = ei;
    return true;
}

// MX fp6 planes tolerate this much spread between the per-channel maxima of a 32-channel block before the
// HourGlass2 is moved to fp8 corrections (largest over lower quartile of the live channels): tools/precision_gpu.py --gamma (profiles/r04_channel_disparity.txt) measures max|ab| 1.6e-4 at one
// decade of spread, 2.6e-4 at 1.5, 6.8e-4 at 2 and 1.0e-3 at 3, against 1.1e-4 for fp8 at any of them; with this measure the synthetic checkpoint reads 9, its Student-t variants 10-20, the four spreads 37 / 78 / 159 / 1 153
constexpr float MX6_DISPARITY_LIMIT = 64.f;
// after a calibration pass: rebuild the HourGlass2 on fp8 corrections and calibrate again when the measured disparity asks for it
// Channels levelled first (keeps fp6) when a block's spread exceeds this; the plain synthetic checkpoint (9) is left as it is
constexpr float MX6_EQUALISE_ABOVE = 16.f;
int enhance_disparity_guard(disco_ctx* c, const float* d_user_gray, int un, int uh, int uw) {
    if (c->opt.network == SUBNET_SEG || c->opt.network == SUBNET_REP || c->enhance_fp8_fallback || arith_of(c, "enhanceNet.outConv") != ARITH_MX6) return DISCO_OK;
    if (!c->sd.count("enhanceNet.outConv.weight")) return DISCO_OK;      // (host weights gone: cannot happen after disco_finalize)
    // All or nothing: everything a rebuild touches - the layers, the exponents and maxima of the HourGlass2's tensors, the levelling
    // factors and the flags - is saved first and put back if the rebuild or its calibration fails, so that "a failing disco_calibrate
    // leaves the previous calibration in place" (include/disco_hip.h) also holds on this path
    struct Saved {
        decltype(c->conv) conv; decltype(c->sexp) sexp; decltype(c->sexp_nat) sexp_nat; decltype(c->amax) amax; decltype(c->chan_amax) chan_amax;
        decltype(c->eq_in) eq_in; decltype(c->eq_out) eq_out; bool equalised, fp8; float disp, disp_before;
    };
    auto save = [&]() { return Saved{c->conv, c->sexp, c->sexp_nat, c->amax, c->chan_amax, c->eq_in, c->eq_out, c->equalised, c->enhance_fp8_fallback, c->mx6_disparity, c->mx6_disparity_before_eq}; };
    auto restore = [&](Saved& v) {
        c->conv.swap(v.conv); c->sexp.swap(v.sexp); c->sexp_nat.swap(v.sexp_nat); c->amax.swap(v.amax); c->chan_amax.swap(v.chan_amax);
        c->eq_in.swap(v.eq_in); c->eq_out.swap(v.eq_out); c->equalised = v.equalised; c->enhance_fp8_fallback = v.fp8;
        c->mx6_disparity = v.disp; c->mx6_disparity_before_eq = v.disp_before;
        c->seg_ws_bytes.clear();
    };
    auto rebuild = [&]() -> int {
        // the old layers' device buffers stay in c->allocs until disco_destroy (a few tens of MB); the tensors' exponents and maxima are measured again
        int rc = make_enhance(c);
        if (rc) return rc;
        for (auto* m : {&c->sexp, &c->sexp_nat})
            for (auto it = m->begin(); it != m->end();) it = it->first.compare(0, 11, "enhanceNet.") == 0 ? m->erase(it) : std::next(it);
        for (auto it = c->amax.begin(); it != c->amax.end();) it = it->first.compare(0, 11, "enhanceNet.") == 0 ? c->amax.erase(it) : std::next(it);
        c->chan_amax.clear();
        c->mx6_disparity = 0.f;
        c->seg_ws_bytes.clear();
        return calibrate_ctx(c, d_user_gray, un, uh, uw);
    };
    static const bool no_eq = std::getenv("DISCO_NO_EQUALISE") != nullptr;       // (tests of the fp8 fallback)
    if (!c->equalised && !no_eq && c->mx6_disparity > MX6_EQUALISE_ABOVE) {
        Saved before = save();
        if (plan_equalisation(c)) {
            c->equalised = true;
            c->mx6_disparity_before_eq = c->mx6_disparity;
            if (int rc = rebuild()) { restore(before); return rc; }
        }
    }
    if (!(c->mx6_disparity > MX6_DISPARITY_LIMIT)) return DISCO_OK;
    Saved before = save();
    c->enhance_fp8_fallback = true;
    const float measured = c->mx6_disparity;          // (no fp6 tensor is left to measure after the rebuild: keep what decided it)
    const int rc = rebuild();
    if (rc) { restore(before); return rc; }
    c->mx6_disparity = measured;
    return rc;
}

}  // namespace disco_api

extern "C" {


int disco_expected_tensors(void) { return (int)layout().t.size(); }

static int expected_tensor(const Layout& l, int i, const char** key, int64_t shape[4], int* ndim) {
    if (i < 0 || i >= (int)l.t.size() || !key || !shape || !ndim) { set_error("bad index"); return DISCO_EINVAL; }
    const ExpectedTensor& e = l.t[i];
    *key = e.key.c_str();
    *ndim = (int)e.shape.size();
    for (int d = 0; d < *ndim; ++d) shape[d] = e.shape[d];
    return DISCO_OK;
}

int disco_expected_tensor(int i, const char** key, int64_t shape[4], int* ndim) { return expected_tensor(layout(), i, key, shape, ndim); }

int disco_expected_tensor_ctx(disco_ctx* c, int i, const char** key, int64_t shape[4], int* ndim) {
    if (!c) { set_error("null context"); return DISCO_EINVAL; }
    return expected_tensor(layout(c->opt.hint2regress != 0), i, key, shape, ndim);
}

int disco_create(int device, const disco_options* opt, disco_ctx** out) {
    if (!opt || !out) { set_error("null argument"); return DISCO_EINVAL; }
    // --psize (inference.py:147): the superpixel CELL of the pooling / un-pooling / size count (model.py:109-121,191); SpixelNet's affinity is
    // the same nine-neighbour map whatever the cell.  16 has the dedicated kernels; 8 and 32 run the general ones (ABI 11)
    if (opt->sp_size != 8 && opt->sp_size != 16 && opt->sp_size != 32) { set_error("sp_size %d unsupported (8, 16 or 32; inference.py:147)", opt->sp_size); return DISCO_EUNSUPPORTED; }
    if (opt->n_clusters < 1 || opt->n_clusters > 32) { set_error("n_clusters %d outside [1,32]", opt->n_clusters); return DISCO_EUNSUPPORTED; }
    if (opt->precision != DISCO_PREC_F16X3 && opt->precision != DISCO_PREC_MX8 && opt->precision != DISCO_PREC_MX8_ALL && opt->precision != DISCO_PREC_X2Q && opt->precision != DISCO_PREC_MX6) { set_error("precision %d", opt->precision); return DISCO_EINVAL; }
    if ((opt->hint2regress | opt->spix_pos) & ~1) { set_error("hint2regress / spix_pos must be 0 or 1"); return DISCO_EINVAL; }
    if (opt->network < 0 || opt->network > 3) { set_error("network %d: 0 (colorizer), 1 SpixelNet, 2 ColorProbNet, 3 HourGlass2", opt->network); return DISCO_EINVAL; }
    if (opt->network && (opt->hint2regress || opt->spix_pos)) { set_error("a stand-alone network context takes no colorizer flags"); return DISCO_EINVAL; }
    int ndev = 0;
    DISCO_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { set_error("device %d of %d", device, ndev); return DISCO_EINVAL; }
    DISCO_HIP_CHECK(hipSetDevice(device));
    disco_ctx* c = new (std::nothrow) disco_ctx();
    if (!c) return DISCO_ENOMEM;
    c->device = device; c->opt = *opt;
    *out = c;
    return DISCO_OK;
}

int disco_destroy(disco_ctx* c) {
    if (!c) return DISCO_OK;
    hipSetDevice(c->device);
    for (auto& e : c->prof) hipEventDestroy(e.ev);
    for (auto& e : c->conv_prof) { hipEventDestroy(e.e0); hipEventDestroy(e.e1); }
    for (auto& g : c->stg) { if (g.ev) hipEventDestroy(g.ev); if (g.h) hipHostFree(g.h); }
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    for (void* p : c->allocs) hipFree(p);
    delete c;
    return DISCO_OK;
}

int disco_load_tensor(disco_ctx* c, const char* key, const float* h_data, const int64_t* shape, int ndim) {
    if (!c || !key || ndim < 0 || ndim > 4 || (ndim && !shape)) { set_error("bad argument"); return DISCO_EINVAL; }
    if (c->finalized) { set_error("context already finalized"); return DISCO_ESTATE; }
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    if (h_data) t.data.assign(h_data, h_data + t.numel());
    c->sd[key] = std::move(t);
    return DISCO_OK;
}

int disco_finalize(disco_ctx* c) {
    if (!c) { set_error("null context"); return DISCO_EINVAL; }
    if (c->finalized) return DISCO_OK;
    const int sub = c->opt.network;
    const std::string only = subnet_prefix(sub);            // stand-alone network contexts hold that network's tensors and nothing else
    auto mine = [&](const std::string& key) { return only.empty() || key.compare(0, only.size(), only) == 0; };
    // strict: same key set and shapes as the reference's load_state_dict(strict=True) (utils_train.py:151)
    size_t n_expected = 0;
    const Layout& lay = layout(c->opt.hint2regress != 0);
    for (const ExpectedTensor& e : lay.t) {
        if (!mine(e.key)) continue;
        ++n_expected;
        auto it = c->sd.find(e.key);
        if (it == c->sd.end()) { set_error("missing key in state_dict: %s", e.key.c_str()); return DISCO_ESTATE; }
        if (it->second.shape != e.shape) { set_error("size mismatch for %s", e.key.c_str()); return DISCO_ESHAPE; }
        if (!e.is_count && it->second.data.size() != it->second.numel()) { set_error("no data for %s", e.key.c_str()); return DISCO_EINVAL; }
    }
    if (c->sd.size() != n_expected) {
        for (auto& kv : c->sd) {
            bool found = false;
            for (const ExpectedTensor& e : lay.t)
                if (e.key == kv.first && mine(e.key)) { found = true; break; }
            if (!found) { set_error("unexpected key in state_dict: %s", kv.first.c_str()); return DISCO_ESTATE; }
        }
    }
    DISCO_HIP_CHECK(hipSetDevice(c->device));
    int rc;
    if ((rc = dev_alloc(c, 256, (void**)&c->d_sat))) return rc;
    DISCO_HIP_CHECK(hipMemset(c->d_sat, 0, 256));
    const std::string sg = "segnet.net.";
    if (sub == SUBNET_FULL || sub == SUBNET_SEG) {
    if ((rc = make_c1(c, sg + "conv0a.0", sg + "conv0a.1"))) return rc;
    for (const char* k : {"conv0b", "conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "conv3_1",
                          "conv2_1", "conv1_1"})
        if ((rc = make_conv(c, sg + k + ".0", sg + k + ".1", "", nullptr, 0, k[5] == 'a' && k[6] == '\0' && k[4] != '0'))) return rc;   // conv1a..conv4a: stride 2
    if (use_mx(c, sg)) {     // cat(o1, deconv0): both sources carry 16 real channels in a 32-channel block
        std::vector<int> map(64, -1);
        for (int i = 0; i < 16; ++i) { map[i] = i; map[32 + i] = 16 + i; }
        if ((rc = make_conv(c, sg + "conv0_1.0", sg + "conv0_1.1", "", &map, 64))) return rc;
    } else if ((rc = make_conv(c, sg + "conv0_1.0", sg + "conv0_1.1", "", nullptr, 0, false))) return rc;
    for (const char* k : {"deconv3", "deconv2", "deconv1", "deconv0"}) if ((rc = make_deconv(c, sg + k + ".0"))) return rc;
    if ((rc = make_conv(c, sg + "pred_mask0", "", "", nullptr, 0, false, false))) return rc;
    }
    if (sub == SUBNET_SEG) { c->sd.clear(); c->finalized = true; return calibrate_ctx(c); }
    const std::string rp = "repnet.";
    if (sub == SUBNET_FULL || sub == SUBNET_REP) {
    if ((rc = make_c1(c, rp + "conv1_2.0", ""))) return rc;
    if ((rc = make_conv(c, rp + "conv1_2.2", "", rp + "conv1_2.4"))) return rc;
    for (const char* b : {"conv2_3", "conv3_3", "conv4_3", "conv5_3", "conv6_3", "conv7_3"}) {
        if ((rc = make_conv(c, rp + b + ".0", "", "", nullptr, 0, b[4] >= '2' && b[4] <= '4' /* conv2_3.0, conv3_3.0, conv4_3.0: stride 2 */))) return rc;
        if ((rc = make_conv(c, rp + b + ".2", "", ""))) return rc;
        if ((rc = make_conv(c, rp + b + ".4", "", rp + b + ".6"))) return rc;
    }
    if ((rc = make_upconv(c, rp + "conv8up.1"))) return rc;
    if ((rc = make_conv(c, rp + "conv3short8.0", "", ""))) return rc;
    if ((rc = make_conv(c, rp + "conv8_3.1", "", ""))) return rc;
    if ((rc = make_conv(c, rp + "conv8_3.3", "", rp + "conv8_3.5"))) return rc;
    if ((rc = make_upconv(c, rp + "conv9up.1"))) return rc;
    if ((rc = make_conv(c, rp + "conv9_2.0", "", rp + "conv9_2.2"))) return rc;
    if ((rc = make_upconv(c, rp + "conv10up.1"))) return rc;
    if ((rc = make_conv(c, rp + "conv10_2.1", "", ""))) return rc;
    }
    if (sub == SUBNET_REP) { c->sd.clear(); c->finalized = true; return calibrate_ctx(c); }
    if ((rc = make_enhance(c))) return rc;
    // a stand-alone HourGlass2 has no input of its own to measure ranges on: it is calibrated by disco_calibrate on its caller's first batch
    if (sub == SUBNET_ENH) { c->finalized = true; return DISCO_OK; }
    if ((rc = make_encoder(c, "wildpath", &c->d_enc[0]))) return rc;
    if ((rc = make_encoder(c, "hintpath", &c->d_enc[1]))) return rc;
    for (int i = 0; i < 2; ++i) {
        if ((rc = dev_alloc(c, encoder_packed_floats() * sizeof(float), (void**)&c->d_enc_pk[i]))) return rc;
        if ((rc = launch_encoder_pack(c->d_enc[i], c->d_enc_pk[i], nullptr))) return rc;
    }
    DISCO_HIP_CHECK(hipStreamSynchronize(nullptr));
    if ((rc = upload_vec(c, T(c, "mid_word_prj.weight").data, &c->d_mid_w))) return rc;
    if ((rc = upload_vec(c, T(c, "trg_word_emb.weight").data, &c->d_emb_w))) return rc;
    if ((rc = upload_vec(c, T(c, "trg_word_prj.weight").data, &c->d_trg_w))) return rc;
    std::vector<float> q;
    for (auto& r : GAMUT_RUNS) for (int b = r[1]; b <= r[2]; b += 10) { q.push_back((float)r[0]); q.push_back((float)b); }
    if (q.size() != 2 * N_VOCAB) { set_error("gamut table size"); return DISCO_ESTATE; }
    if ((rc = upload_vec(c, q, &c->d_q_to_ab))) return rc;
    // host copies are no longer needed - except the HourGlass2's, which the channel-disparity guard may have to pack again (28 MB)
    for (auto it = c->sd.begin(); it != c->sd.end();) it = it->first.compare(0, 11, "enhanceNet.") == 0 ? std::next(it) : c->sd.erase(it);
    c->finalized = true;
    if ((rc = calibrate_ctx(c))) return rc;
    return enhance_disparity_guard(c);
}

}  // extern "C"
