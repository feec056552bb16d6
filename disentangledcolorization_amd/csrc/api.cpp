// api.cpp — the C ABI of libdisco_hip.so (include/disco_hip.h): context, strict checkpoint loading,
// spectral-norm / batch-norm folding, weight packing, and the forward plan of
// AnchorColorProb.forward(test_mode=True) (models/model.py:103-199) as a sequence of HIP launches.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <cstring>
#include <array>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

using namespace disco;

namespace {

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    size_t numel() const { size_t n = 1; for (auto d : shape) n *= (size_t)d; return n; }
};

struct ExpectedTensor {
    std::string key;
    std::vector<int64_t> shape;
    bool is_count;  // BatchNorm num_batches_tracked (int64 scalar, unused)
};

// ---- expected checkpoint layout (mirrors disentangledcolorization_amd/layout.py) ----------
struct Layout {
    std::vector<ExpectedTensor> t;
    void add(const std::string& k, std::vector<int64_t> s, bool cnt = false) { t.push_back({k, std::move(s), cnt}); }
    void conv(const std::string& k, int cin, int cout, bool bias = true) {
        add(k + ".weight", {cout, cin, 3, 3});
        if (bias) add(k + ".bias", {cout});
    }
    void sn(const std::string& k, int cin, int cout) {
        add(k + ".bias", {cout});
        add(k + ".weight_orig", {cout, cin, 3, 3});
        add(k + ".weight_u", {cout});
        add(k + ".weight_v", {9 * cin});
    }
    void bn(const std::string& k, int c) {
        add(k + ".weight", {c}); add(k + ".bias", {c}); add(k + ".running_mean", {c}); add(k + ".running_var", {c});
        add(k + ".num_batches_tracked", {}, true);
    }
    explicit Layout(bool hint2regress) {
        const std::string s = "segnet.net.";
        const char* seg[10] = {"conv0a", "conv0b", "conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b"};
        const int seg_ci[10] = {1, 16, 16, 32, 32, 64, 64, 128, 128, 256}, seg_co[10] = {16, 16, 32, 32, 64, 64, 128, 128, 256, 256};
        for (int i = 0; i < 10; ++i) { conv(s + seg[i] + ".0", seg_ci[i], seg_co[i], false); bn(s + seg[i] + ".1", seg_co[i]); }
        const char* dec[4] = {"deconv3", "deconv2", "deconv1", "deconv0"};
        const char* decc[4] = {"conv3_1", "conv2_1", "conv1_1", "conv0_1"};
        const int dci[4] = {256, 128, 64, 32}, dco[4] = {128, 64, 32, 16};
        for (int i = 0; i < 4; ++i) {
            add(s + dec[i] + ".0.weight", {dci[i], dco[i], 4, 4}); add(s + dec[i] + ".0.bias", {dco[i]});
            conv(s + decc[i] + ".0", 2 * dco[i], dco[i], false); bn(s + decc[i] + ".1", dco[i]);
        }
        conv(s + "pred_mask0", 16, 9);
        const std::string r = "repnet.";
        sn(r + "conv1_2.0", 1, 64); sn(r + "conv1_2.2", 64, 64); bn(r + "conv1_2.4", 64);
        const char* blk[6] = {"conv2_3", "conv3_3", "conv4_3", "conv5_3", "conv6_3", "conv7_3"};
        const int bci[6] = {64, 128, 256, 512, 512, 512}, bco[6] = {128, 256, 512, 512, 512, 512};
        for (int i = 0; i < 6; ++i) {
            sn(r + blk[i] + ".0", bci[i], bco[i]); sn(r + blk[i] + ".2", bco[i], bco[i]); sn(r + blk[i] + ".4", bco[i], bco[i]);
            bn(r + blk[i] + ".6", bco[i]);
        }
        conv(r + "conv8up.1", 512, 256); conv(r + "conv3short8.0", 256, 256); conv(r + "conv8_3.1", 256, 256);
        conv(r + "conv8_3.3", 256, 256); bn(r + "conv8_3.5", 256);
        conv(r + "conv9up.1", 256, 128); conv(r + "conv9_2.0", 128, 128); bn(r + "conv9_2.2", 128);
        conv(r + "conv10up.1", 128, 64); conv(r + "conv10_2.1", 64, 64);
        const std::string e = "enhanceNet.";
        conv(e + "inConv.inConv.0", 65, 64); conv(e + "inConv.conv.0", 64, 64); bn(e + "inConv.conv.2", 64);
        conv(e + "down1.conv.0", 64, 128); conv(e + "down1.conv.2", 128, 128); bn(e + "down1.conv.4", 128);
        conv(e + "down2.conv.0", 128, 256); conv(e + "down2.conv.2", 256, 256); bn(e + "down2.conv.4", 256);
        for (int i = 0; i < 3; ++i) {
            const std::string p = e + "residual." + std::to_string(i) + ".conv.";
            conv(p + "0", 256, 256); sn(p + "1", 256, 256); conv(p + "3", 256, 256);
        }
        const char* up[2] = {"up2", "up1"}; const int uci[2] = {256, 128}, uco[2] = {128, 64};
        for (int i = 0; i < 2; ++i) {
            const std::string p = e + up[i];
            conv(p + ".conv1", uci[i], uco[i]); conv(p + ".combine", 2 * uco[i], uco[i]); conv(p + ".conv2.0", uco[i], uco[i]);
            conv(p + ".conv2.2", uco[i], uco[i]); bn(p + ".conv2.4", uco[i]);
        }
        conv(e + "outConv", 64, 2);
        for (const char* path : {"wildpath", "hintpath"})
            for (int l = 0; l < ENC_LAYERS; ++l) {
                const std::string q = std::string(path) + ".layers." + std::to_string(l) + ".";
                add(q + "self_attn.in_proj_weight", {192, 64}); add(q + "self_attn.in_proj_bias", {192});
                add(q + "self_attn.out_proj.weight", {64, 64}); add(q + "self_attn.out_proj.bias", {64});
                add(q + "linear1.weight", {256, 64}); add(q + "linear1.bias", {256});
                add(q + "linear2.weight", {64, 256}); add(q + "linear2.bias", {64});
                add(q + "norm1.weight", {64}); add(q + "norm1.bias", {64}); add(q + "norm2.weight", {64}); add(q + "norm2.bias", {64});
            }
        add("mid_word_prj.weight", {313, 64});
        if (hint2regress) { add("trg_word_emb.weight", {64, 67}); add("trg_word_prj.weight", {2, 64}); }   // model.py:63-64
        else { add("trg_word_emb.weight", {64, 378}); add("trg_word_prj.weight", {313, 64}); }          // model.py:66-67
    }
};
const Layout& layout(bool hint2regress = false) {
    static Layout plain(false), h2r(true);
    return hint2regress ? h2r : plain;
}

// the 313 in-gamut ab bins as (a, b_min, b_max) runs (utils/gamut_pts.npy; same table as gamut.py)
const int GAMUT_RUNS[20][3] = {{-90, 50, 90}, {-80, 20, 90}, {-70, 0, 90}, {-60, -20, 90}, {-50, -30, 100}, {-40, -40, 100},
                               {-30, -50, 100}, {-20, -50, 100}, {-10, -60, 100}, {0, -70, 100}, {10, -80, 90}, {20, -80, 90},
                               {30, -90, 90}, {40, -100, 90}, {50, -100, 80}, {60, -110, 80}, {70, -110, 80}, {80, -110, 70},
                               {90, -110, 70}, {100, -90, 0}};

struct ConvLayer {
    int c_in = 0, c_in_pad = 0, c_out = 0;
    int kind = 0;                 // 0 plain 3x3, 1 ConvTranspose 4x4 s2 as 4-phase conv, 2 upsample+3x3 as 4-phase conv
    bool mx = false;              // packed for conv3x3_mx_kernel (fp16 main product + fp8 corrections)
    int x2q = 0;                  // mx: weight-pack variant / arithmetic of the kernel: 0 = f16 + fp8x2, 1 = f16x2 + fp8 (sources with al8-only q
                                  // planes), 2 = f16 + fp6x2 (sources with fp6 q planes)
    int c_out_k = 0;              // mx: output channels the kernel computes (c_out padded with zero weights so that act
                                  // outputs carry whole 32-channel blocks; per phase for the depth-to-space kinds)
    int c_real = 0;               // real (reference) output channels, per phase for kinds 1 and 2: FLOP accounting
    int32_t* d_wexp = nullptr;    // mx: per-output-channel scale exponents of the fp8 weight planes
    f16* d_w = nullptr;
    uint32_t* d_tapmask = nullptr;
    float* d_bias = nullptr;
    float* d_bn_scale = nullptr;
    float* d_bn_shift = nullptr;
};
struct DirectLayer {  // fp32 VALU convs / deconvs
    int c_in = 0, c_out = 0;
    float* d_w = nullptr;
    float* d_bias = nullptr;
    float* d_bn_scale = nullptr;
    float* d_bn_shift = nullptr;
};

struct ProfEntry { std::string name; hipEvent_t ev; double flops; };

// first-fit arena over the caller's workspace
struct Arena {
    struct Blk { size_t off, size; bool used; };
    std::vector<Blk> blks;
    size_t cap = 0, peak = 0;
    explicit Arena(size_t c) : cap(c) { blks.push_back({0, c, false}); }
    size_t alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        for (size_t i = 0; i < blks.size(); ++i)
            if (!blks[i].used && blks[i].size >= bytes) {
                const size_t off = blks[i].off;
                if (blks[i].size > bytes) { Blk rest{off + bytes, blks[i].size - bytes, false}; blks[i].size = bytes; blks.insert(blks.begin() + i + 1, rest); }
                blks[i].used = true;
                peak = std::max(peak, off + bytes);
                return off;
            }
        return (size_t)-1;
    }
    void release(size_t off) {
        for (size_t i = 0; i < blks.size(); ++i)
            if (blks[i].off == off && blks[i].used) {
                blks[i].used = false;
                if (i + 1 < blks.size() && !blks[i + 1].used) { blks[i].size += blks[i + 1].size; blks.erase(blks.begin() + i + 1); }
                if (i > 0 && !blks[i - 1].used) { blks[i - 1].size += blks[i].size; blks.erase(blks.begin() + i); }
                return;
            }
    }
};

}  // namespace

struct disco_ctx {
    int device = 0;
    disco_options opt{};
    bool finalized = false;
    std::map<std::string, HostTensor> sd;
    std::vector<void*> allocs;
    std::map<std::string, ConvLayer> conv;
    std::map<std::string, DirectLayer> direct;
    std::map<std::string, int> sexp;     // scale exponent of every activation tensor, by producer (set by calibration)
    std::map<std::string, int> sexp_nat; // calibration: the exponent each tensor would take on its own (max |x| 2^e in [16, 32))
    std::map<std::string, std::string> tie;   // tensor -> the earlier tensor it is concatenated with on read (they share one exponent)
    std::map<std::string, float> amax;   // calibration: max |x| of every conv output (fp16 range guard, diagnostics)
    unsigned int* d_sat = nullptr;       // mx: q-plane elements that had to be clamped since the last read
    bool calibrated = false;
    // Channel disparity of the tensors that carry MX fp6 planes (one E8M0 scale per pixel and 32 CHANNELS): per tensor and 32-channel block
    // the ratio of the largest per-channel max |x| to the lower quartile of the block's live channels, measured in the calibration pass; the largest ratio over all
    // blocks is `mx6_disparity`.  Beyond MX6_DISPARITY_LIMIT the HourGlass2 is rebuilt on fp8 corrections (e4m3: 4 exponent bits), see disco_finalize.
    float mx6_disparity = 0.f;
    std::string mx6_disparity_key;
    bool enhance_fp8_fallback = false;
    // Cross-layer channel equalisation of the HourGlass2 (plan_equalisation): per conv layer the power-of-two factor every OUTPUT channel is
    // multiplied by (weight rows + bias, or the BN affine behind the activation) and every INPUT channel's weights are multiplied by (the
    // inverse of its producer's factor) - the network function is unchanged in exact arithmetic, the tensors in between get level channels
    std::map<std::string, std::vector<float>> eq_out, eq_in, chan_amax;
    bool equalised = false;
    float mx6_disparity_before_eq = 0.f;
    // One host thread at a time inside a context: the forward entry points, calibration and the setters below lock this.  The GPU work
    // of successive calls still overlaps across the streams they were given; what is serialised is the host-side issue (staging ring,
    // one-shot progress event, profiling vectors, calibration tables are plain members).
    std::mutex mu;
    float* d_enc[2] = {nullptr, nullptr};
    float* d_enc_pk[2] = {nullptr, nullptr};     // their B-fragment images for encoder_tail_kernel (launch_encoder_pack)
    float* d_mid_w = nullptr; float* d_emb_w = nullptr; float* d_trg_w = nullptr; float* d_q_to_ab = nullptr;
    std::map<std::pair<int, int>, float*> pos_cache;
    // pinned staging ring for the small host->device index arrays of disco_forward: a pageable hipMemcpyAsync
    // blocks the host until the stream reaches the copy, which would stop the host from running ahead
    struct Staging { void* h = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
    Staging stg[4];
    int stg_next = 0;
    int profiling = 0;
    // small batches (run_plan): SpixelNet runs on this stream next to ColorProbNet on the caller's - neither fills the GPU on its own
    bool side_failed = false;                    // creating it failed once: small forwards stay on the caller's stream
    std::map<std::array<int, 3>, size_t> seg_ws_bytes;      // (n, H, W) -> workspace block of a forked SpixelNet
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // disco_set_progress_event: recorded on the next forward's stream behind its `progress_after`-th MFMA conv launch (one shot)
    hipEvent_t progress_ev = nullptr;
    int progress_after = 0, progress_seen = 0;
    // disco_set_debug_checksums: every forward adds a checksum of each stage's output to row (sequence number % rows) of this table
    unsigned long long* d_dbg = nullptr;
    char* d_dump = nullptr; size_t dump_stride = 0;     // disco_set_debug_dump: per-row copies of the first token GEMM's input and output
    int dbg_rows = 0, dbg_cols = 0;
    long dbg_seq = 0;
    std::vector<ProfEntry> prof;
    struct ConvProf { hipEvent_t e0, e1; double flops; std::string key; double bytes; };
    std::vector<ConvProf> conv_prof;
    std::vector<std::pair<std::string, float>> prof_ms;
    std::vector<double> prof_flops;
};

namespace {

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
template <class T>
int upload_vec(disco_ctx* c, const std::vector<T>& v, T** out) { return upload(c, v.data(), v.size() * sizeof(T), (void**)out); }

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
int run_conv(const ConvArgs& ca, hipStream_t s) { return launch_conv3x3_x3(ca, s); }

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
enum { ARITH_F16X3 = 0, ARITH_MX8 = 1, ARITH_X2Q = 2, ARITH_MX6 = 3 };
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

// ------------------------------------------------------------------------------------------------------------------
// forward plan
// ------------------------------------------------------------------------------------------------------------------
struct Plan {
    // set around ONE conv() call: that f16x3 layer computes its input in LDS from the gray image through the Cin = 1 conv `layer`
    // (conv_mx_kernel.h, GENC1); the in0 handed to conv() then only describes the virtual tensor (p == nullptr)
    struct FusedC1 { const float* gray; const DirectLayer* layer; int act; float slope; };
    const FusedC1* fuse = nullptr;
    disco_ctx* c;
    const disco_forward_args* a;
    Arena arena;
    bool dry;            // size pass: no launches
    bool calib = false;  // calibration pass of disco_finalize: measures activation ranges, fixes the q-plane scales
    hipStream_t s;
    char* base;
    int rc = DISCO_OK;

    Plan(disco_ctx* c_, const disco_forward_args* a_, size_t cap, bool dry_)
        : c(c_), a(a_), arena(cap), dry(dry_), s(dry_ ? nullptr : (hipStream_t)a_->stream),
          base(dry_ ? nullptr : (char*)a_->d_workspace) {}

    bool ok() const { return rc == DISCO_OK; }
    void* raw(size_t bytes) {
        const size_t off = arena.alloc(bytes);
        if (off == (size_t)-1) { if (ok()) { set_error("workspace too small (need > %zu bytes)", arena.cap); rc = DISCO_ENOMEM; } return nullptr; }
        return dry ? (void*)(uintptr_t)(off + 256) : (void*)(base + off);   // dry: fake non-null token
    }
    void drop(void* p) { if (p) arena.release(dry ? (size_t)(uintptr_t)p - 256 : (size_t)((char*)p - base)); }
    // planes of an activation tensor: F_LO = fp16 lo plane, F_Q = fp8 q planes a8|al8 (scale exponent of producer `key`),
    // F_QL = al8-only q planes (the operand of the f16x2+fp8 arithmetic), F_Q6 = fp6 q planes (f16+fp6x2)
    enum { F_LO = 1, F_Q = 2, F_QL = 4, F_Q6 = 8 };
    int stage_arith = ARITH_F16X3;     // arithmetic of the stack being planned (set per network by the plan)
    bool mx() const { return stage_arith != ARITH_F16X3; }
    int cpad(int ch) const { return round_up(ch, stage_arith == ARITH_X2Q ? 64 : (mx() ? 32 : 16)); }
    int dfmt() const { return stage_arith == ARITH_X2Q ? (int)F_QL : (stage_arith == ARITH_MX6 ? (int)F_Q6 : (mx() ? (int)F_Q : (int)F_LO)); }          // what a conv -> conv tensor carries
    Act act(int n, int h, int w, int ch, int fmt) {
        Act t; t.n = n; t.h = h; t.w = w; t.c = ch;
        const size_t el = t.elems();
        t.plane = (fmt & F_LO) ? el : 0;
        t.q_off = (fmt & (F_Q | F_QL | F_Q6)) ? el * 2 * ((fmt & F_LO) ? 2 : 1) : 0;
        t.q_kind = (fmt & F_QL) ? 1 : ((fmt & F_Q6) ? 2 : 0);
        t.p = (f16*)raw(t.bytes());
        return t;
    }
    void drop(Act& t) { drop((void*)t.p); t.p = nullptr; }
    long dbg_row = -1;
    int dbg_col = 0;
    // debugging aid: checksum of a stage's output into the context's table (tools/stagger_probe.py finds the first stage whose
    // result depends on what else runs on the GPU)
    void dbg(const void* p, size_t bytes) {
        if (dry || calib || dbg_row < 0 || !ok() || !p) return;
        if (dbg_col < c->dbg_cols) rc = launch_checksum(p, bytes, c->d_dbg + dbg_row * c->dbg_cols + dbg_col, s);
        ++dbg_col;
    }
    void mark(const char* name, double flops = 0.0) {
        // DISCO_HOST_TIMING=1 (diagnostic): host time between the stage marks of every forward, printed at the "enhance" mark
        static const bool host_timing = std::getenv("DISCO_HOST_TIMING") != nullptr;
        if (host_timing && !dry && !calib) {
            static thread_local std::vector<std::pair<const char*, std::chrono::steady_clock::time_point>> ht;
            ht.emplace_back(name, std::chrono::steady_clock::now());
            if (std::string(name) == "enhance") {
                std::string line = "[host us]";
                for (size_t i = 1; i < ht.size(); ++i)
                    line += " " + std::string(ht[i].first) + " " + std::to_string(std::chrono::duration_cast<std::chrono::microseconds>(ht[i].second - ht[i - 1].second).count());
                std::fprintf(stderr, "%s\n", line.c_str());
                ht.clear();
            }
        }
        if (dry || !c->profiling || !ok()) return;
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return;
        hipEventRecord(ev, s);
        c->prof.push_back({name, ev, flops});
    }
    // scale exponent of the tensor produced by `key` (fixed by the calibration pass of disco_finalize)
    bool scale_of(const std::string& key, int* sexp) {
        auto it = c->sexp.find(key);
        if (it == c->sexp.end()) {
            if (calib) { *sexp = 0; return true; }
            set_error("no calibrated scale for the output of %s", key.c_str()); rc = DISCO_ESTATE; return false;
        }
        *sexp = it->second;
        return true;
    }
    // Calibration (disco_finalize / disco_calibrate): `produce` has just written tensor `t` with a provisional exponent (the previous
    // calibration's, or 0).  Measure max |xs| of its hi plane; if the provisional scale overflowed fp16 or buried the tensor in its
    // subnormals, move it by 2^10 and produce again; then fix the exponent so that the maximum lands in [16, 32) - 2^11 of fp16
    // headroom (and 14x of fp8's +-448) for other inputs, values down to 2^-7 of the maximum keep a normal fp16 lo word - and
    // produce once more with the final exponent.  `tie`: a tensor that is concatenated on read with an earlier one (skip
    // connections; the conv accumulates both sources in ONE domain) runs this pass on the earlier tensor's current exponent; after
    // the pass the pair takes the SMALLER of the two natural exponents (calibrate_ctx), so that neither leaves the [16, 32) target
    // upwards (the fp8 planes clamp at 448).  A pair whose ranges differ by more than 2^10 cannot share a scale: the error names it.
    template <class F>
    void calibrate(const std::string& key, Act& t, F&& produce, const std::string& tie = "") {
        if (!calib || dry || !ok()) return;
        float amax_s = 0.f;      // stored maximum
        for (int attempt = 0; attempt < 12; ++attempt) {
            float* d_amax = (float*)raw(256);
            if (!ok()) return;
            if (hipMemsetAsync(d_amax, 0, 4, s) != hipSuccess) { rc = DISCO_EHIP; return; }
            rc = launch_act_amax(t, d_amax, s);
            if (ok() && (hipMemcpyAsync(&amax_s, d_amax, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = hip_fail(hipGetLastError(), "calibration amax");
            drop(d_amax);
            if (!ok()) return;
            const bool too_big = !(amax_s <= 16384.f);                      // also inf / NaN
            const bool too_small = amax_s > 0.f && amax_s < 1.f / 1024.f;
            if (!too_big && !too_small) break;
            if (attempt == 11 || !std::isfinite(std::ldexp(1.f, t.sexp))) {
                set_error("activation range of %s cannot be brought into fp16 range (stored max |x| = %g at scale 2^%d): not a finite network output", key.c_str(), (double)amax_s, t.sexp);
                rc = DISCO_EUNSUPPORTED; return;
            }
            t.sexp += too_big ? -10 : 10;
            produce();
            if (!ok()) return;
        }
        float amax = std::ldexp(amax_s, -t.sexp);                           // true maximum
        {   // calibrations accumulate: a later disco_calibrate on other images can only widen a tensor's range
            auto prev = c->amax.find(key);
            if (prev != c->amax.end() && prev->second > amax) amax = prev->second;
        }
        c->amax[key] = amax;
        int e = 0;
        if (amax > 0.f) { std::frexp(amax, &e); e = 5 - e; }                // amax 2^e in [16, 32)
        c->sexp_nat[key] = e;
        if (!tie.empty()) {
            auto it = c->sexp.find(tie);
            if (it == c->sexp.end()) { set_error("calibration order: %s is tied to %s, which has no exponent yet", key.c_str(), tie.c_str()); rc = DISCO_ESTATE; return; }
            auto nt = c->sexp_nat.find(tie);
            const int e_tie = nt == c->sexp_nat.end() ? it->second : nt->second;
            // (2^10: the tied tensor is produced once at its partner's exponent during this pass - a maximum of [16, 32) 2^10 still fits
            // fp16; round 3 allowed 2^12, where that intermediate overflowed and the error named a downstream layer instead of the pair)
            if (amax > 0.f && c->amax[tie] > 0.f && std::abs(e - e_tie) > 10) {
                set_error("%s and %s are concatenated on read and must share one scale, but their ranges differ too much (max |x| %g vs %g): "
                          "this checkpoint cannot run in fp16 hi/lo arithmetic", key.c_str(), tie.c_str(), (double)amax, (double)c->amax[tie]);
                rc = DISCO_EUNSUPPORTED; return;
            }
            c->tie[key] = tie;
            e = it->second;           // this pass: the partner's current exponent (the concat conv needs equal ones)
        }
        c->sexp[key] = e;
        if (t.sexp != e) { t.sexp = e; produce(); }
        if (t.q_off && t.q_kind == 2 && t.c % 32 == 0 && t.c <= 1024) channel_disparity(key, t);
    }
    // MX fp6 planes share one scale per pixel and 32 channels: a channel whose values sit far below its block's largest loses its correction
    // operands (e2m3: below 1/8 of the block maximum subnormal, below 1/60 zero).  Harmless while the consumer's weights do not make up for the
    // difference - trained BatchNorm affines can (tools/precision_gpu.py --gamma: 2 decades of per-channel spread cost 6.8e-4, 3 decades the
    // 1e-3 bar).  Measured here per tensor: per block the largest per-channel max |x| over the live channels' lower quartile; disco_finalize acts on it.
    void channel_disparity(const std::string& key, const Act& t) {
        float* d = (float*)raw((size_t)t.c * 4);
        if (!ok()) return;
        std::vector<float> h(t.c);
        if (hipMemsetAsync(d, 0, (size_t)t.c * 4, s) != hipSuccess) { rc = DISCO_EHIP; return; }
        rc = launch_act_channel_amax(t, d, s);
        if (ok() && (hipMemcpyAsync(h.data(), d, (size_t)t.c * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = hip_fail(hipGetLastError(), "channel amax");
        drop(d);
        if (!ok()) return;
        {
            std::vector<float>& acc = c->chan_amax[key];
            if (acc.size() != h.size()) acc.assign(h.size(), 0.f);
            for (size_t i = 0; i < h.size(); ++i) acc[i] = std::max(acc[i], std::ldexp(h[i], -t.sexp));       // true values
        }
        for (int b = 0; b + 32 <= t.c; b += 32) {
            // channels that never fire on the calibration images (ReLU-dead: max 0) carry nothing and are left out; of the live ones the
            // largest against the lower quartile: a quarter of a block's channels below 1/64 of its maximum is where fp6 starts to cost
            std::vector<float> v;
            for (int i = 0; i < 32; ++i) if (h[b + i] > 0.f) v.push_back(h[b + i]);
            if (v.size() < 16) continue;
            std::sort(v.begin(), v.end());
            const float ratio = v.back() / v[v.size() / 4];
            if (ratio > c->mx6_disparity) { c->mx6_disparity = ratio; c->mx6_disparity_key = key; }
        }
    }

    // MFMA conv: out = bn(act(conv(cat(in0[,in1])) + bias [+ res]));  ofmt: planes of the output tensor (-1: the default)
    Act conv(const std::string& key, const Act& in0, const Act* in1, int up0, int up1, int stride, int actc, float slope,
             const Act* res = nullptr, float* out_f32 = nullptr, bool d2s = false, bool softmax = false, int ofmt = -1, const std::string& tie = "") {
        const ConvLayer& L = c->conv.at(key);
        const int hin = in0.h << up0, win = in0.w << up0;
        const int ho = (hin - 1) / stride + 1, wo = (win - 1) / stride + 1;
        if (ofmt < 0) ofmt = dfmt();
        const int co_t = L.mx ? L.c_out_k : L.c_out;                 // channels the kernel computes
        Act out{};
        if (d2s) out = act(in0.n, 2 * ho, 2 * wo, co_t / 4, ofmt);
        else if (!out_f32) out = act(in0.n, ho, wo, co_t, ofmt);
        if (dry || !ok()) return out;
        if (in0.c + (in1 ? in1->c : 0) != L.c_in_pad) { set_error("conv %s: input channels %d != %d", key.c_str(), in0.c + (in1 ? in1->c : 0), L.c_in_pad); rc = DISCO_ESHAPE; return out; }
        if (!out_f32 && !scale_of(key, &out.sexp)) return out;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool timed = c->profiling >= 2 && !calib && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
        if (timed) hipEventRecord(e0, s);
        if (L.mx) {
            auto launch = [&]() {
                ConvMxArgs ca{};
                const Act* src[2] = {&in0, in1};
                const int ups[2] = {up0, up1};
                ca.nsrc = in1 ? 2 : 1;
                for (int i = 0; i < ca.nsrc; ++i) {
                    const bool tail_src = i == 1 && src[i]->c == 16 && !src[i]->q_off && L.x2q == 0;     // the H-only tail chunk (launch_conv3x3_mx checks the rest)
                    if (!tail_src && (!src[i]->q_off || src[i]->q_off >= ((size_t)1 << 32) || src[i]->q_kind != L.x2q)) { set_error("conv %s: source %d has no (addressable) q planes of kind %d", key.c_str(), i, L.x2q); rc = DISCO_ESHAPE; return; }
                    ca.src[i] = {src[i]->p, (uint32_t)src[i]->q_off, src[i]->c, src[i]->h, src[i]->w, ups[i], src[i]->sexp};
                }
                ca.n = in0.n; ca.h_in = hin; ca.w_in = win; ca.c_in = L.c_in_pad;
                ca.h_out = ho; ca.w_out = wo; ca.stride = stride;
                ca.w = L.d_w; ca.wexp = L.d_wexp; ca.tapmask = L.d_tapmask; ca.c_out = co_t; ca.c_out_pad = co_t;
                ca.bias = L.d_bias; ca.bn_scale = L.d_bn_scale; ca.bn_shift = L.d_bn_shift;
                ca.res = res ? res->p : nullptr; ca.res_plane = res ? (long)res->plane : 0; ca.res_sexp = res ? res->sexp : 0;
                ca.out = out.p; ca.out_plane = (long)out.plane; ca.out_q_off = out.q_off; ca.out_sexp = out.sexp; ca.out_q_kind = out.q_kind;
                ca.out_f32 = out_f32; ca.d2s_c = d2s ? co_t / 4 : 0; ca.softmax = softmax ? 1 : 0;
                ca.act = actc; ca.slope = slope; ca.sat = calib ? nullptr : c->d_sat; ca.x2q = L.x2q == 1; ca.q6 = L.x2q == 2;
                rc = launch_conv3x3_mx(ca, s);
            };
            launch();
            if (!out_f32) calibrate(key, out, launch, tie);
        } else {
            auto launch = [&]() {
            ConvArgs ca{};
            if ((!in0.plane && !fuse) || (in1 && !in1->plane) || (res && !res->plane) || (!out_f32 && !out.plane)) { set_error("conv %s: the f16x3 kernel needs lo planes", key.c_str()); rc = DISCO_ESHAPE; return; }
            ca.src[0] = {in0.p, (long)in0.plane, in0.c, in0.h, in0.w, up0, in0.sexp};
            ca.nsrc = 1;
            if (fuse) { ca.c1_gray = fuse->gray; ca.c1_w = fuse->layer->d_w; ca.c1_bias = fuse->layer->d_bias; ca.c1_act = fuse->act; ca.c1_slope = fuse->slope; }
            if (in1) { ca.src[1] = {in1->p, (long)in1->plane, in1->c, in1->h, in1->w, up1, in1->sexp}; ca.nsrc = 2; }
            ca.n = in0.n; ca.h_in = hin; ca.w_in = win; ca.c_in = L.c_in_pad;
            ca.h_out = ho; ca.w_out = wo; ca.stride = stride;
            ca.w = L.d_w; ca.tapmask = L.d_tapmask; ca.c_out = L.c_out; ca.c_out_pad = L.c_out;
            ca.bias = L.d_bias; ca.bn_scale = L.d_bn_scale; ca.bn_shift = L.d_bn_shift;
            ca.res = res ? res->p : nullptr; ca.res_plane = res ? (long)res->plane : 0; ca.res_sexp = res ? res->sexp : 0;
            ca.out = out.p; ca.out_plane = (long)out.plane; ca.out_sexp = out.sexp;
            ca.out_f32 = out_f32; ca.d2s_c = d2s ? L.c_out / 4 : 0; ca.softmax = softmax ? 1 : 0;
            ca.act = actc; ca.slope = slope; ca.precision = DISCO_PREC_F16X3;
            rc = run_conv(ca, s);
            };
            launch();
            if (!out_f32) calibrate(key, out, launch, tie);
        }
        if (out_f32) dbg(out_f32, (size_t)in0.n * co_t * ho * wo * 4); else dbg(out.p, out.bytes());
        if (!calib && c->progress_ev && ++c->progress_seen >= c->progress_after) {
            if (hipEventRecord(c->progress_ev, s) != hipSuccess && ok()) rc = DISCO_EHIP;
            c->progress_ev = nullptr;
        }
        if (timed) {
            hipEventRecord(e1, s);
            // algorithmic FLOPs (the reference's dense count on its real channels): 16 taps for a ConvTranspose 4x4 s2 and 9 taps
            // on the UPSAMPLED grid for up-convs, per input pixel of this launch; 9 taps per output pixel otherwise
            const double taps = L.kind == 1 ? 16.0 * L.c_real : (L.kind == 2 ? 36.0 * L.c_real : 9.0 * L.c_real);
            // compulsory HBM bytes: every source plane the kernel reads once (4 B per element: hi + lo, or hi + two fp8 planes), every
            // output plane written once, the residual read once, the packed weights once
            const double bpe_out = out_f32 ? 4.0 : 2.0 * (1 + ((ofmt & F_LO) ? 1 : 0) + ((ofmt & F_Q) ? 1 : 0)) + ((ofmt & F_QL) ? 1.0 : 0.0) + ((ofmt & F_Q6) ? 1.5 : 0.0);
            double bytes = (L.x2q == 1 ? 3.0 : (L.x2q == 2 ? 3.5 : 4.0)) * in0.n * ((double)in0.c * in0.h * in0.w + (in1 ? (double)in1->c * in1->h * in1->w : 0.0));
            if (fuse) bytes = 4.0 * in0.n * (double)in0.h * in0.w;                 // the gray image is all this layer reads
            bytes += bpe_out * in0.n * (double)(out_f32 ? L.c_real : co_t) * ho * wo;
            if (res) bytes += 4.0 * in0.n * (double)co_t * ho * wo;
            bytes += L.mx ? (double)conv_mx_packed_bytes(co_t, L.c_in_pad, L.x2q) : (double)conv3x3_packed_bytes(L.c_out, L.c_in_pad);
            c->conv_prof.push_back({e0, e1, 2.0 * taps * L.c_in * (double)ho * wo * in0.n, key, bytes});
        }
        return out;
    }
    Act deconv(const std::string& key, const Act& in, float slope, const std::string& tie) {
        return conv(key, in, nullptr, 0, 0, 1, DISCO_ACT_LRELU, slope, nullptr, nullptr, true, false, -1, tie);
    }
    Act c1(const std::string& key, const float* gray, int n, int h, int w, int actc, float slope) {
        const DirectLayer& L = c->direct.at(key);
        Act out = act(n, h, w, cpad(L.c_out), dfmt());
        if (dry || !ok()) return out;
        if (!scale_of(key, &out.sexp)) return out;
        auto launch = [&]() { rc = launch_conv_c1(gray, L.d_w, L.d_bias, nullptr, nullptr, out, L.c_out, actc, slope, calib ? nullptr : c->d_sat, s); };
        launch();
        calibrate(key, out, launch);
        dbg(out.p, out.bytes());
        return out;
    }
};

constexpr int RELU = DISCO_ACT_RELU, LRELU = DISCO_ACT_LRELU, NOACT = DISCO_ACT_NONE;

// ---- a1 SpixelNet (network.py:293-313): gray -> affinity (n,9,H,W), softmax over the 9 neighbour slots -------------
void segnet_stage(Plan& P, disco_ctx* c, const float* d_gray, int n, int H, int W, float* d_affinity) {
    const bool dry = P.dry;
    P.stage_arith = arith_of(c, "segnet.");
    hipStream_t s = P.s;
    const std::string sg = "segnet.net.";
    Act s0a = P.c1(sg + "conv0a.0", d_gray, n, H, W, LRELU, 0.1f);
    Act o1 = P.conv(sg + "conv0b.0", s0a, nullptr, 0, 0, 1, LRELU, 0.1f); P.drop(s0a);
    Act t = P.conv(sg + "conv1a.0", o1, nullptr, 0, 0, 2, LRELU, 0.1f);
    Act o2 = P.conv(sg + "conv1b.0", t, nullptr, 0, 0, 1, LRELU, 0.1f); P.drop(t);
    t = P.conv(sg + "conv2a.0", o2, nullptr, 0, 0, 2, LRELU, 0.1f);
    Act o3 = P.conv(sg + "conv2b.0", t, nullptr, 0, 0, 1, LRELU, 0.1f); P.drop(t);
    t = P.conv(sg + "conv3a.0", o3, nullptr, 0, 0, 2, LRELU, 0.1f);
    Act o4 = P.conv(sg + "conv3b.0", t, nullptr, 0, 0, 1, LRELU, 0.1f); P.drop(t);
    t = P.conv(sg + "conv4a.0", o4, nullptr, 0, 0, 2, LRELU, 0.1f);
    Act o5 = P.conv(sg + "conv4b.0", t, nullptr, 0, 0, 1, LRELU, 0.1f); P.drop(t);
    // the transposed convs' outputs are concatenated on read with the encoder tensors of the same level: one exponent per pair
    Act d = P.deconv(sg + "deconv3.0", o5, 0.1f, sg + "conv3b.0"); P.drop(o5);
    Act cc = P.conv(sg + "conv3_1.0", o4, &d, 0, 0, 1, LRELU, 0.1f); P.drop(d); P.drop(o4);
    d = P.deconv(sg + "deconv2.0", cc, 0.1f, sg + "conv2b.0"); P.drop(cc);
    cc = P.conv(sg + "conv2_1.0", o3, &d, 0, 0, 1, LRELU, 0.1f); P.drop(d); P.drop(o3);
    d = P.deconv(sg + "deconv1.0", cc, 0.1f, sg + "conv1b.0"); P.drop(cc);
    cc = P.conv(sg + "conv1_1.0", o2, &d, 0, 0, 1, LRELU, 0.1f); P.drop(d); P.drop(o2);
    d = P.deconv(sg + "deconv0.0", cc, 0.1f, sg + "conv0b.0"); P.drop(cc);
    cc = P.conv(sg + "conv0_1.0", o1, &d, 0, 0, 1, LRELU, 0.1f); P.drop(d); P.drop(o1);
    // pred_mask0 (16 -> 9, bias) + softmax over the 9 slots, fp32 NCHW out (network.py:311-312)
    P.conv(sg + "pred_mask0", cc, nullptr, 0, 0, 1, NOACT, 0.f, nullptr, dry ? (float*)16 : d_affinity, false, true);
    P.drop(cc);
}

// a2 ColorProbNet (network.py:220-236): gray (n,1,H,W) fp32 -> the 64-channel full-resolution feature tensor (fp16 hi + lo planes)
Act repnet_stage(Plan& P, disco_ctx* c, const float* d_gray, int n, int H, int W) {
    const std::string rp = "repnet.";
    P.stage_arith = arith_of(c, rp);
    // conv1_2.0 (Cin = 1) is not a launch of its own when its consumer runs on the 32 x 16 x 64 tile with enough tiles to fill the GPU: conv1_2.2
    // then computes its input tiles in LDS from the gray image, with the stand-alone kernel's arithmetic (bit-identical either way), and the
    // 64-channel full-resolution tensor in between (4 B per element: 1.07 GB at 64 x 256^2) is never written or read.  The calibration pass
    // and the workspace sizing take the two-launch form (the tensor's exponent is measured on the stand-alone kernel).
    static const bool fuse_env = [] { const char* e = std::getenv("DISCO_FUSE_C1"); return !e || std::atoi(e) != 0; }();
    const bool fuse_c1 = fuse_env && !P.dry && !P.calib && P.stage_arith == ARITH_F16X3 && W > 16 && H > 8 &&
                         (long)((W + 31) / 32) * ((H + 15) / 16) * n >= (long)num_cus_current() * 3 / 4;
    Act t{}, f{};
    if (fuse_c1) {
        Act v{};
        v.n = n; v.h = H; v.w = W; v.c = 64;
        if (P.scale_of(rp + "conv1_2.0", &v.sexp)) {
            const Plan::FusedC1 fc{d_gray, &c->direct.at(rp + "conv1_2.0"), LRELU, 0.2f};
            P.fuse = &fc;
            f = P.conv(rp + "conv1_2.2", v, nullptr, 0, 0, 1, LRELU, 0.2f);
            P.fuse = nullptr;
        }
    } else {
        t = P.c1(rp + "conv1_2.0", d_gray, n, H, W, LRELU, 0.2f);
        f = P.conv(rp + "conv1_2.2", t, nullptr, 0, 0, 1, LRELU, 0.2f); P.drop(t);
    }
    Act f3{};
    const char* blk[6] = {"conv2_3", "conv3_3", "conv4_3", "conv5_3", "conv6_3", "conv7_3"};
    for (int b = 0; b < 6; ++b) {
        const std::string k = rp + blk[b];
        Act x1 = P.conv(k + ".0", f, nullptr, 0, 0, b < 3 ? 2 : 1, LRELU, 0.2f);
        if (b != 2) P.drop(f);   // b == 2: f is f3_3, kept alive for the conv3short8 shortcut
        Act x2 = P.conv(k + ".2", x1, nullptr, 0, 0, 1, LRELU, 0.2f); P.drop(x1);
        f = P.conv(k + ".4", x2, nullptr, 0, 0, 1, LRELU, 0.2f); P.drop(x2);
        if (b == 1) f3 = f;
    }
    Act sh = P.conv(rp + "conv3short8.0", f3, nullptr, 0, 0, 1, NOACT, 0.f, nullptr, nullptr, false, false, Plan::F_LO);   // residual only
    P.drop(f3);
    Act f8 = P.conv(rp + "conv8up.1", f, nullptr, 0, 0, 1, RELU, 0.f, &sh, nullptr, true); P.drop(sh); P.drop(f);
    t = P.conv(rp + "conv8_3.1", f8, nullptr, 0, 0, 1, RELU, 0.f); P.drop(f8);
    f8 = P.conv(rp + "conv8_3.3", t, nullptr, 0, 0, 1, RELU, 0.f); P.drop(t);
    t = P.conv(rp + "conv9up.1", f8, nullptr, 0, 0, 1, NOACT, 0.f, nullptr, nullptr, true); P.drop(f8);
    Act f9 = P.conv(rp + "conv9_2.0", t, nullptr, 0, 0, 1, RELU, 0.f); P.drop(t);
    t = P.conv(rp + "conv10up.1", f9, nullptr, 0, 0, 1, RELU, 0.f, nullptr, nullptr, true); P.drop(f9);
    Act feats = P.conv(rp + "conv10_2.1", t, nullptr, 0, 0, 1, RELU, 0.f, nullptr, nullptr, false, false, Plan::F_LO); P.drop(t);   // pooled, not convolved
    return feats;
}

// a13 HourGlass2 (network.py:125-144) from its two input tensors - the 64 feature channels and the 16-channel gray block - to (n,2,H,W) fp32 NCHW;
// out_act: DISCO_ACT_TANH inside the colorizer (model.py:197), NOACT for the stand-alone network
void enhance_stage(Plan& P, disco_ctx* c, Act full, Act g16, int out_act, float* d_out) {
    (void)c;
    const std::string en = "enhanceNet.";
    Act t = P.conv(en + "inConv.inConv.0", full, &g16, 0, 0, 1, RELU, 0.f); P.drop(full); P.drop(g16);
    Act e1 = P.conv(en + "inConv.conv.0", t, nullptr, 0, 0, 1, RELU, 0.f); P.drop(t);
    t = P.conv(en + "down1.conv.0", e1, nullptr, 0, 0, 2, RELU, 0.f);
    Act e2 = P.conv(en + "down1.conv.2", t, nullptr, 0, 0, 1, RELU, 0.f); P.drop(t);
    t = P.conv(en + "down2.conv.0", e2, nullptr, 0, 0, 2, RELU, 0.f);
    const int rfmt = P.mx() ? (Plan::F_LO | P.dfmt()) : Plan::F_LO;      // residual-chain tensors: convolved AND added back
    Act x = P.conv(en + "down2.conv.2", t, nullptr, 0, 0, 1, RELU, 0.f, nullptr, nullptr, false, false, rfmt); P.drop(t);
    for (int r = 0; r < 3; ++r) {
        const std::string k = en + "residual." + std::to_string(r) + ".conv.";
        Act t1 = P.conv(k + "0", x, nullptr, 0, 0, 1, NOACT, 0.f);
        Act t2 = P.conv(k + "1", t1, nullptr, 0, 0, 1, RELU, 0.f); P.drop(t1);
        Act y = P.conv(k + "3", t2, nullptr, 0, 0, 1, RELU, 0.f, &x, nullptr, false, false, rfmt); P.drop(t2); P.drop(x);
        x = y;
    }
    t = P.conv(en + "up2.conv1", x, nullptr, 0, 0, 1, NOACT, 0.f, nullptr, nullptr, false, false, -1, en + "down1.conv.2"); P.drop(x);      // concatenated with e2
    Act u = P.conv(en + "up2.combine", t, &e2, 1, 0, 1, RELU, 0.f); P.drop(t); P.drop(e2);
    t = P.conv(en + "up2.conv2.0", u, nullptr, 0, 0, 1, RELU, 0.f); P.drop(u);
    u = P.conv(en + "up2.conv2.2", t, nullptr, 0, 0, 1, RELU, 0.f); P.drop(t);
    t = P.conv(en + "up1.conv1", u, nullptr, 0, 0, 1, NOACT, 0.f, nullptr, nullptr, false, false, -1, en + "inConv.conv.0"); P.drop(u);     // concatenated with e1
    u = P.conv(en + "up1.combine", t, &e1, 1, 0, 1, RELU, 0.f); P.drop(t); P.drop(e1);
    t = P.conv(en + "up1.conv2.0", u, nullptr, 0, 0, 1, RELU, 0.f); P.drop(u);
    u = P.conv(en + "up1.conv2.2", t, nullptr, 0, 0, 1, RELU, 0.f); P.drop(t);
    P.conv(en + "outConv", u, nullptr, 0, 0, 1, out_act, 0.f, nullptr, d_out);
    P.drop(u);
}

// Stand-alone networks (ABI 9; models/network.py:125,147,260 as modules of their own): a context created with disco_options.network = 1
// (SpixelNet), 2 (ColorProbNet) or 3 (HourGlass2) holds that network's tensors only and serves one entry point.
//   1: d_in gray (n,1,H,W)           -> d_out (n,9,H,W)  softmax over the 9 slots (network.py:312)
//   2: d_in gray (n,1,H,W)           -> d_out (n,64,H,W) features (network.py:234)
//   3: d_in (n,65,H,W) = cat(gray, 64 features) (model.py:196) -> d_out (n,2,H,W) BEFORE the tanh of model.py:197
enum { SUBNET_FULL = 0, SUBNET_SEG = 1, SUBNET_REP = 2, SUBNET_ENH = 3 };
inline const char* subnet_prefix(int which) { return which == SUBNET_SEG ? "segnet.net." : which == SUBNET_REP ? "repnet." : which == SUBNET_ENH ? "enhanceNet." : ""; }
inline size_t subnet_out_channels(int which) { return which == SUBNET_SEG ? 9 : which == SUBNET_REP ? 64 : 2; }
void subnet_stage(Plan& P, disco_ctx* c, int which, const float* d_in, int n, int H, int W, float* d_out) {
    const bool dry = P.dry;
    hipStream_t s = P.s;
    switch (which) {
    case SUBNET_SEG:
        segnet_stage(P, c, d_in, n, H, W, dry ? nullptr : d_out);
        return;
    case SUBNET_REP: {
        Act feats = repnet_stage(P, c, d_in, n, H, W);
        if (!dry && P.ok()) P.rc = launch_act_to_nchw(feats.p, (long)feats.plane, d_out, n, 64, H, W, feats.c, s, feats.sexp);
        P.drop(feats);
        return;
    }
    case SUBNET_ENH: {
        P.stage_arith = arith_of(c, "enhanceNet.");
        const long hw = (long)H * W;
        float* gray = (float*)P.raw((size_t)n * hw * 4);          // channel 0 of every image, contiguous: what the gray-block kernels read
        if (!dry && P.ok() && hipMemcpy2DAsync(gray, hw * 4, d_in, 65 * hw * 4, hw * 4, n, hipMemcpyDeviceToDevice, s) != hipSuccess)
            P.rc = hip_fail(hipGetLastError(), "gray channel copy");
        // the same two input tensors, formats and calibration keys as in the colorizer (run_plan): "upfeat" = the 64 feature channels
        const int infmt = P.stage_arith == ARITH_MX6 ? (int)Plan::F_Q : P.dfmt();
        Act full = P.act(n, H, W, 64, infmt);
        const bool gtail = P.mx();
        Act g16 = gtail ? P.act(n, H, W, 16, 0) : P.act(n, H, W, P.cpad(16), infmt);
        if (!dry && P.ok() && P.scale_of("upfeat", &full.sexp) && P.scale_of("gray16", &g16.sexp)) {}
        unsigned int* sat = P.calib ? nullptr : c->d_sat;
        auto up = [&]() { P.rc = launch_nchw_to_act_mx(d_in + hw, full, 64, s, 65 * hw, sat); };
        auto gr = [&]() { P.rc = gtail ? launch_gray_tail(gray, 1, g16, s) : launch_gray16(gray, 1, g16, sat, s); };
        if (!dry && P.ok()) { up(); P.calibrate("upfeat", full, up); }
        if (!dry && P.ok()) { gr(); P.calibrate("gray16", g16, gr, "upfeat"); }
        P.drop(gray);
        enhance_stage(P, c, full, g16, NOACT, dry ? (float*)16 : d_out);
        return;
    }
    default:
        set_error("not a stand-alone network context"); P.rc = DISCO_ESTATE;
    }
}

int run_plan(disco_ctx* c, const disco_forward_args* a, size_t cap, bool dry, size_t* peak, bool calib = false) {
    Plan P(c, a, cap, dry);
    P.calib = calib;
    if (!dry && !calib && !c->calibrated) { set_error("context used before its calibration pass"); return DISCO_ESTATE; }
    const int n = a->n, H = a->h, W = a->w, sp = c->opt.sp_size, K = c->opt.n_clusters;
    const int hs = H / sp, ws = W / sp, L = hs * ws;
    const bool test = a->test_mode != 0, h2r = c->opt.hint2regress != 0, spos = c->opt.spix_pos != 0;
    const int rep = (test && a->sampled_T > 0) ? 3 : 1, n2 = n * rep;
    const double px = (double)n * H * W;
    hipStream_t s = P.s;
    if (!dry) {
        for (auto& e : c->prof) hipEventDestroy(e.ev);
        c->prof.clear();
        for (auto& e : c->conv_prof) { hipEventDestroy(e.e0); hipEventDestroy(e.e1); }
        c->conv_prof.clear();
    }
    if (!dry && !calib && c->d_dbg && c->dbg_rows > 0) {
        P.dbg_row = c->dbg_seq++ % c->dbg_rows;
        if (hipMemsetAsync(c->d_dbg + P.dbg_row * c->dbg_cols, 0, (size_t)c->dbg_cols * 8, s) != hipSuccess) P.rc = DISCO_EHIP;
    }
    P.mark("start");

    // Small batches (up to 8 x 256^2 worth of pixels): neither conv stack fills 256 CUs on its own (one 256^2 image is 128 full-resolution
    // tiles, the 512-channel layers run 64 workgroups), and SpixelNet and ColorProbNet depend on nothing but the gray image.  SpixelNet then
    // runs on a side stream of the context, in a block of the workspace reserved for it (sized by shape alone, so the sizing pass and the
    // forward agree), and the caller's stream waits for it in front of the pooling kernel.  Same kernels, same results.  Not while a
    // progress event is armed (it counts conv launches in issue order), not under profiling (stage times), not in the calibration pass.
    static const bool fork_env = [] { const char* e = std::getenv("DISCO_FORK_SEGNET"); return !e || std::atoi(e) != 0; }();
    const bool fork_shape = fork_env && (long)n * H * W <= 8L * 256 * 256;
    void* seg_ws = nullptr;
    size_t seg_bytes = 0;
    if (fork_shape) {
        // (cached per shape: every real forward would otherwise run a dry plan of SpixelNet just to learn the size again - host time on
        // the latency path; the table is cleared whenever the layers are rebuilt)
        const std::array<int, 3> key{n, H, W};
        auto it = c->seg_ws_bytes.find(key);
        if (it == c->seg_ws_bytes.end()) {
            Plan S(c, a, (size_t)1 << 46, true);
            segnet_stage(S, c, nullptr, n, H, W, nullptr);
            if (S.rc) P.rc = S.rc;
            else it = c->seg_ws_bytes.emplace(key, S.arena.peak + 4096).first;
        }
        if (P.ok()) { seg_bytes = it->second; seg_ws = P.raw(seg_bytes); }
    }
    bool forked = false;
    if (fork_shape && !dry && !calib && P.ok() && !c->progress_ev && !c->profiling && P.dbg_row < 0) {
        if (!c->side && !c->side_failed) {
            if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
                // whatever was created goes back, and the fork stays off for this context: no retry (and no leak) on every later small forward
                if (c->ev_join) (void)hipEventDestroy(c->ev_join);
                if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
                if (c->side) (void)hipStreamDestroy(c->side);
                c->ev_join = c->ev_fork = nullptr; c->side = nullptr; c->side_failed = true;
                (void)hipGetLastError();
            }
        }
        if (c->side && hipEventRecord(c->ev_fork, s) == hipSuccess && hipStreamWaitEvent(c->side, c->ev_fork, 0) == hipSuccess) {
            disco_forward_args a2 = *a;
            a2.d_workspace = seg_ws; a2.workspace_bytes = seg_bytes; a2.stream = c->side;
            Plan S(c, &a2, seg_bytes, false);
            segnet_stage(S, c, a->d_gray, n, H, W, a->d_affinity);
            if (S.rc) P.rc = S.rc;
            if (hipEventRecord(c->ev_join, c->side) != hipSuccess && P.ok()) P.rc = DISCO_EHIP;
            forked = true;
        }
    }
    if (!forked) segnet_stage(P, c, a->d_gray, n, H, W, dry ? nullptr : a->d_affinity);
    P.mark("segnet", 2.0 * 2.8962e9 * px / 65536.0);

    // ---- a2 ColorProbNet (network.py:220-236) ----------------------------------------------------------------
    Act feats = repnet_stage(P, c, a->d_gray, n, H, W);
    if (forked && P.ok() && hipStreamWaitEvent(s, c->ev_join, 0) != hipSuccess) P.rc = DISCO_EHIP;
    if (forked && !P.ok()) (void)hipStreamSynchronize(c->side);      // an error return must not leave the side stream writing into the caller's buffers
    if (seg_ws) P.drop(seg_ws);
    P.mark("repnet", 2.0 * 68.8914e9 * px / 65536.0);

    // ---- a3-a5 tokens, colours, sizes (model.py:114-121) ------------------------------------------------------
    float* src = (float*)P.raw((size_t)n * L * 64 * 4);
    float* spix_ab = (float*)P.raw((size_t)n * 2 * L * 4);
    float* sizes = (float*)P.raw((size_t)n * L * 4);
    // --spix_pos (model.py:106-112): the sine encoding of every PIXEL is pooled with the features (64 more channels,
    // the same (H*W,64) table for every image), so each image gets its own position sequence (n,L,64)
    const int cpool = spos ? 130 : 66;
    float* pos_img = spos ? (float*)P.raw((size_t)n * L * 64 * 4) : nullptr;
    void* pool_ws = P.raw(poolfeat_ws_bytes(n, cpool, H, W, sp));
    float* pos = nullptr;
    if (!dry && P.ok()) P.rc = spos ? get_pos(c, H, W, &pos) : get_pos(c, hs, ws, &pos);
    if (!dry && P.ok()) {
        PoolArgs pa{};
        pa.feat_act = feats.p; pa.feat_plane = (long)feats.plane; pa.c_act = 64; pa.feat_mul = std::ldexp(1.f, -feats.sexp);
        pa.feat_nchw = a->d_ab; pa.c_nchw = 2; pa.prob = a->d_affinity;
        if (spos) { pa.feat_bc = pos; pa.c_bc = 64; pa.bc_out = pos_img; }
        pa.partial = (float*)pool_ws; pa.cnt = (float*)pool_ws + (size_t)n * L * 9 * (cpool + 1);
        pa.tok_out = src; pa.c_tok = 64; pa.nchw_out = spix_ab; pa.c_from = 64;
        pa.conf = nullptr; pa.sizes = sizes; pa.n = n; pa.H = H; pa.W = W; pa.sp = sp;
        P.rc = launch_poolfeat(pa, s);
    }
    if (!dry && P.ok() && P.dbg_row >= 0 && c->d_dump) {
        // debugging aid: [tokens after pool][tokens at the first GEMM][q|k|v][affinity][feats hi+lo] per row
        char* dst = c->d_dump + (size_t)P.dbg_row * c->dump_stride;
        const size_t sb = (size_t)n * L * 64 * 4, ab_ = (size_t)n * 9 * H * W * 4, fb = feats.bytes();
        if (5 * sb + ab_ + fb <= c->dump_stride) {
            hipMemcpyAsync(dst, src, sb, hipMemcpyDeviceToDevice, s);
            hipMemcpyAsync(dst + 5 * sb, a->d_affinity, ab_, hipMemcpyDeviceToDevice, s);
            hipMemcpyAsync(dst + 5 * sb + ab_, feats.p, fb, hipMemcpyDeviceToDevice, s);
        }
    }
    P.drop(pool_ws); P.drop(feats);
    if (spos) pos = pos_img;
    const int pos_rep = spos ? 1 : 0;       // wild path: one position sequence per image; hint path: per virtual image / rep
    P.dbg(src, (size_t)n * L * 64 * 4);
    P.mark("poolfeat");

    // ---- a6/a7 wild path + palette logits (model.py:133-135) -------------------------------------------------
    float* enc = (float*)P.raw((size_t)n * L * 64 * 4);
    void* enc_ws = P.raw(encoder_ws_bytes(n2, L));
    int enc_dbg_calls = 0;
    const std::function<void(const void*, size_t)> enc_dbg = [&](const void* p, size_t b) {
        P.dbg(p, b);
        // disco_set_debug_dump: the first token GEMM's input as it is at that moment and its result, next to the copies taken right
        // behind the pooling kernels (tools/stagger_probe.py prints which of them differ from a serialised pass)
        if (enc_dbg_calls++ == 0 && c->d_dump) {
            char* dst = c->d_dump + (size_t)P.dbg_row * c->dump_stride;
            const size_t sb = (size_t)n * L * 64 * 4;
            if (sb + b <= c->dump_stride) {
                hipMemcpyAsync(dst + sb, src, sb, hipMemcpyDeviceToDevice, s);
                hipMemcpyAsync(dst + 2 * sb, p, b, hipMemcpyDeviceToDevice, s);
            }
        }
    };
    // use_mask (model.py:121-125): both stacks bias the keys of superpixels below 25 pixels; the mask IS a function of `sizes`, read in the kernels
    const float* key_sizes = c->opt.use_mask ? sizes : nullptr;
    if (!dry && P.ok()) P.rc = launch_encoder_stack(src, pos, pos_rep, c->d_enc[0], enc, n, L, enc_ws, s, P.dbg_row >= 0 ? &enc_dbg : nullptr, c->d_enc_pk[0], key_sizes, 1);
    P.dbg(enc, (size_t)n * L * 64 * 4);
    if (!dry && P.ok()) P.rc = launch_logits(enc, c->d_mid_w, a->d_pal_logit, n, L, s);
    P.dbg(a->d_pal_logit, (size_t)n * N_VOCAB * L * 4);
    P.mark("wildpath", 2.0 * 0.134e9 * n);

    // ---- a8/a9 anchors (model.py:141) ---------------------------------------------------------------------------
    int32_t* d_idx = (int32_t*)P.raw((size_t)n * K * 4);
    const int mf = a->max_fallback > 0 && a->h_fallback_rows ? a->max_fallback : 0;
    int32_t* d_fb = (int32_t*)P.raw((size_t)n * K * 20 * 4);   // fixed upper bound: (K-1)*20 draws per image at most
    int32_t* d_assign = (int32_t*)P.raw((size_t)n * L * 4);
    int32_t* d_anchor = (int32_t*)P.raw((size_t)n * K * 4);
    int32_t* d_info = (int32_t*)P.raw((size_t)n * 2 * 4);
    // scratch of the several-workgroups-per-image k-means (images of more than 512 tokens: running member sums, centres, flags)
    const size_t km_bytes = (test && !c->opt.random_hint) ? kmeans_ws_bytes(n, L) : 0;
    void* km_ws = km_bytes ? P.raw(km_bytes) : nullptr;
    if (!dry && P.ok()) {
        if (c->opt.random_hint) {
            if (!a->h_hint_pos) { set_error("random_hint context needs h_hint_pos"); P.rc = DISCO_EINVAL; }
            else {
                P.rc = staged_h2d(c, d_idx, a->h_hint_pos, (size_t)n * K * 4, s);
                if (P.ok()) P.rc = launch_hint_mask_from_pos(d_idx, a->d_hint_mask, n, L, K, s);
                if (P.ok() && hipMemsetAsync(d_info, 0, (size_t)n * 8, s) != hipSuccess) P.rc = DISCO_EHIP;
            }
        } else {
            if (!a->h_init_idx) { set_error("clustering context needs h_init_idx"); P.rc = DISCO_EINVAL; }
            else {
                P.rc = staged_h2d(c, d_idx, a->h_init_idx, (size_t)n * K * 4, s);
                if (P.ok() && mf) P.rc = staged_h2d(c, d_fb, a->h_fallback_rows, (size_t)n * mf * 4, s);
                // inference clusters the wild-path tokens (model.py:140-141); the validation forward clusters the pooled
                // GT colours (N,2,h,w) (model.py:169-171)
                if (P.ok()) P.rc = test ? launch_kmeans_anchors(enc, sizes, d_idx, mf ? d_fb : nullptr, mf, d_assign, d_anchor, a->d_hint_mask, d_info, n, L, K, s, 64, 0, km_ws, km_bytes, c->d_sat + 1)
                                        : launch_kmeans_anchors(spix_ab, sizes, d_idx, mf ? d_fb : nullptr, mf, d_assign, d_anchor, a->d_hint_mask, d_info, n, L, K, s, 2, 1);
            }
        }
    }
    if (km_ws) P.drop(km_ws);
    P.mark("anchors");

    // ---- a10/a11 anchor colours + labels (model.py:142-168) ----------------------------------------------------
    int32_t* labels = (int32_t*)P.raw((size_t)n2 * L * 4);
    if (!dry && P.ok()) {
        if (!test || a->sampled_T < 0) {
            if (hipMemcpyAsync(a->d_spix_colors, spix_ab, (size_t)n * 2 * L * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) P.rc = DISCO_EHIP;
            if (P.ok()) P.rc = launch_nearest_bin(spix_ab, c->d_q_to_ab, labels, n, L, s);
        } else {
            P.rc = launch_select_colors(a->d_pal_logit, c->d_q_to_ab, a->d_spix_colors, labels, n, L, 0, rep, s);
        }
    }
    // ---- hint tokens + hint path + refined logits (model.py:175-189) -------------------------------------------
    float* hint = (float*)P.raw((size_t)n2 * L * 64 * 4);
    float* dec = (float*)P.raw((size_t)n2 * L * 64 * 4);
    if (!dry && P.ok()) P.rc = launch_hint_embed(src, rep, h2r ? nullptr : labels, h2r ? a->d_spix_colors : nullptr, a->d_hint_mask, rep, c->d_emb_w, hint, n2, L, s);
    if (!dry && P.ok()) P.rc = launch_encoder_stack(hint, pos, pos_rep ? rep : 0, c->d_enc[1], dec, n2, L, enc_ws, s, nullptr, c->d_enc_pk[1], key_sizes, rep);
    if (!dry && P.ok()) P.rc = launch_logits(dec, c->d_trg_w, a->d_ref_logit, n2, L, s, h2r ? 2 : N_VOCAB);
    P.drop(enc_ws); P.drop(hint); P.drop(labels); P.drop(d_idx); P.drop(d_fb); P.drop(d_assign); P.drop(d_anchor);
    P.drop(enc); P.drop(src); P.drop(spix_ab); P.drop(sizes); if (pos_img) P.drop(pos_img);
    P.mark("hintpath", 2.0 * 0.134e9 * n2);

    // ---- a12 upfeat + a13 HourGlass2 + tanh (model.py:194-197) --------------------------------------------------
    P.stage_arith = arith_of(c, "enhanceNet.");
    // (the upfeat kernel writes fp8 q planes; under the fp6 arithmetic inConv.inConv.0 reads those and writes fp6 ones; the gray channel
    // is that layer's fp16 tail chunk)
    const int infmt = P.stage_arith == ARITH_MX6 ? (int)Plan::F_Q : P.dfmt();
    Act full = P.act(n2, H, W, 64, infmt);
    const bool gtail = P.mx();                        // f16+fp8x2: a 16-channel fp16 tail source without q planes
    Act g16 = gtail ? P.act(n2, H, W, 16, 0) : P.act(n2, H, W, P.cpad(16), infmt);
    if (!dry && P.ok() && P.scale_of("upfeat", &full.sexp) && P.scale_of("gray16", &g16.sexp)) {}
    {
        unsigned int* sat = calib ? nullptr : c->d_sat;
        auto up = [&]() { P.rc = launch_upfeat(dec, 1, a->d_affinity, rep, &full, nullptr, n2, 64, hs, ws, sp, sat, s); };
        auto gr = [&]() { P.rc = gtail ? launch_gray_tail(a->d_gray, rep, g16, s) : launch_gray16(a->d_gray, rep, g16, sat, s); };
        if (!dry && P.ok()) { up(); P.calibrate("upfeat", full, up); }
        if (!dry && P.ok()) { gr(); P.calibrate("gray16", g16, gr, "upfeat"); }        // concatenated on read with the up-sampled features
    }
    P.drop(dec);
    P.mark("upfeat");
    enhance_stage(P, c, full, g16, DISCO_ACT_TANH, dry ? (float*)16 : a->d_pred_colors);
    P.mark("enhance", 2.0 * 55.6794e9 * (double)n2 * H * W / 65536.0);

    // k-means bookkeeping for the caller (the one documented host synchronisation)
    if (!dry && P.ok() && a->h_kmeans_events) {
        std::vector<int32_t> info((size_t)n * 2);
        if (hipMemcpyAsync(info.data(), d_info, info.size() * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { set_error("reading k-means info failed"); P.rc = DISCO_EHIP; }
        else for (int i = 0; i < n; ++i) a->h_kmeans_events[i] = info[2 * i + 1];
    }
    P.drop(d_info);
    if (peak) *peak = P.arena.peak;
    return P.rc;
}

void segnet_stage(Plan& P, disco_ctx* c, const float* d_gray, int n, int H, int W, float* d_affinity);

// The calibration pass of a context (end of disco_finalize): one forward over two synthetic 256x256 images - uniform noise
// and a smooth low-frequency pattern - in which every producer of an activation tensor runs (at least) twice: once to measure the
// tensor's max |x|, once more with the power-of-two scale that measurement fixes (Plan::calibrate).  The scales are
// properties of the checkpoint from then on (deterministic: the inputs are generated here); q-plane clamping at run time is
// counted (disco_saturation_count) so that inputs far outside the calibrated range are noticed.
int calibrate_ctx_impl(disco_ctx* c, const float* d_user_gray, int un, int uh, int uw);
int calibrate_ctx(disco_ctx* c, const float* d_user_gray = nullptr, int un = 0, int uh = 0, int uw = 0) {
    // a pass that fails midway must not leave half of the tensors on new exponents (with `calibrated` still set from an earlier pass,
    // forwards would then run on a mix of two calibrations): all or nothing
    const auto sexp0 = c->sexp, nat0 = c->sexp_nat; const auto amax0 = c->amax; const auto tie0 = c->tie;
    const int rc = calibrate_ctx_impl(c, d_user_gray, un, uh, uw);
    if (rc) { c->sexp = sexp0; c->sexp_nat = nat0; c->amax = amax0; c->tie = tie0; }
    return rc;
}
int calibrate_ctx_impl(disco_ctx* c, const float* d_user_gray, int un, int uh, int uw) {
    const int n = d_user_gray ? un : 2, H = d_user_gray ? uh : 256, W = d_user_gray ? uw : 256, K = c->opt.n_clusters, L = (H / 16) * (W / 16);
    std::vector<float> g(d_user_gray ? 0 : (size_t)n * H * W);
    if (!d_user_gray) {
    unsigned st = 20240607u;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            st = st * 1664525u + 1013904223u;
            const float u = (st >> 8) * (1.f / 16777216.f);
            g[(size_t)y * W + x] = 2.f * u - 1.f;
            g[(size_t)H * W + (size_t)y * W + x] = 0.8f * std::sin(x * (1.f / 9.f)) * std::cos(y * (1.f / 13.f)) + 0.1f * (2.f * u - 1.f);
        }
    }
    std::vector<int32_t> idx((size_t)n * K);
    for (int i = 0; i < n; ++i) for (int k = 0; k < K; ++k) idx[(size_t)i * K + k] = (k * 7 + i) % L;
    disco_forward_args a{};
    a.n = n; a.h = H; a.w = W; a.sampled_T = 0; a.test_mode = 1;
    a.h_init_idx = idx.data(); a.h_hint_pos = idx.data();
    const int sub = c->opt.network;
    const bool seg = sub != SUBNET_FULL;          // a stand-alone network: input -> bufs[0], its one output -> bufs[5]
    if (sub == SUBNET_ENH && !d_user_gray) { set_error("a stand-alone HourGlass2 context is calibrated on its caller's input (disco_calibrate)"); return DISCO_ESTATE; }
    size_t peak = 0;
    int rc;
    if (seg) { Plan P(c, &a, (size_t)1 << 46, true); subnet_stage(P, c, sub, nullptr, n, H, W, nullptr); peak = P.arena.peak; rc = P.rc; }
    else rc = run_plan(c, &a, (size_t)1 << 46, true, &peak);
    if (rc) return rc;
    peak += (size_t)1 << 20;
    const size_t px = (size_t)n * H * W, lt = (size_t)n * L, in_ch = sub == SUBNET_ENH ? 65 : 1;
    const size_t outs[7] = {px * in_ch * 4, px * 2 * 4, lt * 313 * 4, lt * 313 * 4, px * 2 * 4, px * (seg ? subnet_out_channels(sub) : 9) * 4, lt * 2 * 4 + lt * 4};
    void* bufs[8] = {};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 7 && e == hipSuccess; ++i) e = hipMalloc(&bufs[i], outs[i]);
    if (e == hipSuccess) e = hipMalloc(&bufs[7], peak);
    if (e == hipSuccess) e = d_user_gray ? hipMemcpy(bufs[0], d_user_gray, px * in_ch * 4, hipMemcpyDeviceToDevice) : hipMemcpy(bufs[0], g.data(), px * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(bufs[1], 0, px * 2 * 4);
    if (e != hipSuccess) rc = hip_fail(e, "calibration buffers");
    if (!rc) {
        a.d_gray = (const float*)bufs[0]; a.d_ab = (const float*)bufs[1];
        a.d_pal_logit = (float*)bufs[2]; a.d_ref_logit = (float*)bufs[3]; a.d_pred_colors = (float*)bufs[4];
        a.d_affinity = (float*)bufs[5]; a.d_spix_colors = (float*)bufs[6]; a.d_hint_mask = (float*)bufs[6] + lt * 2;
        a.d_workspace = bufs[7]; a.workspace_bytes = peak; a.stream = nullptr;
        if (seg) { Plan P(c, &a, peak, false); P.calib = true; subnet_stage(P, c, sub, a.d_gray, n, H, W, a.d_affinity); rc = P.rc; }
        else rc = run_plan(c, &a, peak, false, nullptr, true);
        if (hipStreamSynchronize(nullptr) != hipSuccess && !rc) rc = hip_fail(hipGetLastError(), "calibration forward");
    }
    for (void* b : bufs) if (b) hipFree(b);
    if (!rc) {
        // tensors that are concatenated on read: the pair shares the smaller natural exponent (an all-zero member does not count)
        for (const auto& kv : c->tie) {
            const bool z0 = c->amax[kv.first] == 0.f, z1 = c->amax[kv.second] == 0.f;
            const int e0 = c->sexp_nat[kv.first], e1 = c->sexp_nat[kv.second];
            const int g = z0 ? e1 : (z1 ? e0 : std::min(e0, e1));
            c->sexp[kv.first] = g; c->sexp[kv.second] = g;
        }
        c->calibrated = true;
    }
    return rc;
}

// disco_set_progress_event arms ONE forward.  Whatever way the next forward entry point is left - argument error, a segnet-only
// forward (18 conv launches: fewer than most `after` counts), a HIP failure - the event is recorded on the call's stream when there
// is one and the handle is dropped: it must never fire in an unrelated later forward (by then the caller may have destroyed it).
struct ProgressDisarm {
    disco_ctx* c; hipStream_t s;
    ~ProgressDisarm() {
        if (c && c->progress_ev) { hipEventRecord(c->progress_ev, s); c->progress_ev = nullptr; }
    }
};

int check_forward_args(disco_ctx* c, const disco_forward_args* a) {
    if (!c || !a) { set_error("null argument"); return DISCO_EINVAL; }
    if (!c->finalized) { set_error("disco_forward before disco_finalize"); return DISCO_ESTATE; }
    const int sp = c->opt.sp_size;
    if (a->n < 1 || a->h < sp || a->w < sp || a->h % sp || a->w % sp) { set_error("bad input size %dx%dx%d (multiples of %d)", a->n, a->h, a->w, sp); return DISCO_ESHAPE; }
    if (!c->opt.network && (a->h / sp) * (a->w / sp) < c->opt.n_clusters) { set_error("fewer tokens than clusters"); return DISCO_ESHAPE; }
    if (a->max_fallback < 0) { set_error("max_fallback %d", a->max_fallback); return DISCO_EINVAL; }
    if (a->max_fallback > c->opt.n_clusters * 20) { set_error("max_fallback %d > K*20", a->max_fallback); return DISCO_EINVAL; }
    if (a->test_mode & ~1) { set_error("test_mode must be 0 or 1"); return DISCO_EINVAL; }
    // model.py:178 reads the undefined name `spix_color` when hint2regress meets test_mode=False: the reference raises
    if (!a->test_mode && c->opt.hint2regress) { set_error("hint2regress has no validation forward (models/model.py:178 raises NameError)"); return DISCO_EUNSUPPORTED; }
    // model.py:154-159 expands everything to the batch of 3 EXCEPT src_pad_mask: nn.MultiheadAttention then rejects the (1,L) mask
    if (c->opt.use_mask && a->test_mode && a->sampled_T > 0) { set_error("use_mask has no diverse forward (the reference's key_padding_mask keeps batch 1: models/model.py:154-159,186)"); return DISCO_EUNSUPPORTED; }
    return DISCO_OK;
}

}  // namespace

// ======================================================================================================================
// C ABI
// ======================================================================================================================
// every size an op entry point takes must be positive (a zero superpixel size would divide by zero on the host, an
// empty dimension would launch an empty grid)
static bool positive(const char* op, std::initializer_list<long> dims) {
    for (long d : dims)
        if (d <= 0) { set_error("%s: non-positive size %ld", op, d); return false; }
    return true;
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
int enhance_disparity_guard(disco_ctx* c, const float* d_user_gray = nullptr, int un = 0, int uh = 0, int uw = 0) {
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
    if (opt->sp_size != 16) { set_error("sp_size %d unsupported (16 only, inference.py:146)", opt->sp_size); return DISCO_EUNSUPPORTED; }
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

int disco_calibrate(disco_ctx* c, const float* d_gray, int n, int h, int w) {
    if (!c || !c->finalized || !d_gray) { set_error("disco_calibrate: bad argument / context not finalized"); return DISCO_EINVAL; }
    if (n < 1 || n > 64 || h < 16 || w < 16 || h % 16 || w % 16 || (!c->opt.network && (h / 16) * (w / 16) < c->opt.n_clusters)) { set_error("disco_calibrate: bad size %dx%dx%d", n, h, w); return DISCO_ESHAPE; }
    DISCO_HIP_CHECK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mu);
    ProgressDisarm disarm{c, nullptr};
    DISCO_HIP_CHECK(hipDeviceSynchronize());      // no forward of this context may be in flight: the scales are about to change
    if (int rc = calibrate_ctx(c, d_gray, n, h, w)) return rc;
    return enhance_disparity_guard(c, d_gray, n, h, w);
}

int disco_saturation_count(disco_ctx* c, void* stream, uint64_t* count) {
    if (!c || !count || !c->finalized) { set_error("disco_saturation_count: bad argument"); return DISCO_EINVAL; }
    std::lock_guard<std::mutex> lk(c->mu);
    unsigned int v = 0;
    DISCO_HIP_CHECK(hipSetDevice(c->device));
    DISCO_HIP_CHECK(hipMemcpyAsync(&v, c->d_sat, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    DISCO_HIP_CHECK(hipMemsetAsync(c->d_sat, 0, 4, (hipStream_t)stream));
    DISCO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    *count = v;
    return DISCO_OK;
}

int disco_kmeans_fallback_count(disco_ctx* c, void* stream, uint64_t* count) {
    if (!c || !count || !c->finalized) { set_error("disco_kmeans_fallback_count: bad argument"); return DISCO_EINVAL; }
    std::lock_guard<std::mutex> lk(c->mu);
    unsigned int v = 0;
    DISCO_HIP_CHECK(hipSetDevice(c->device));
    DISCO_HIP_CHECK(hipMemcpyAsync(&v, c->d_sat + 1, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    DISCO_HIP_CHECK(hipMemsetAsync(c->d_sat + 1, 0, 4, (hipStream_t)stream));
    DISCO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    *count = v;
    return DISCO_OK;
}

int disco_calibration_count(disco_ctx* c) {
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->mu);          // (a calibration on another host thread rewrites these tables)
    return (int)c->amax.size();
}

int disco_enhance_arithmetic(disco_ctx* c, int* precision, float* channel_disparity, float* disparity_before_equalisation) {
    if (!c || !precision || !channel_disparity || !disparity_before_equalisation) { set_error("null argument"); return DISCO_EINVAL; }
    std::lock_guard<std::mutex> lk(c->mu);
    *disparity_before_equalisation = c->equalised ? c->mx6_disparity_before_eq : 0.f;
    const int ar = (c->opt.network == SUBNET_SEG || c->opt.network == SUBNET_REP) ? ARITH_F16X3 : arith_of(c, "enhanceNet.outConv");
    *precision = ar == ARITH_MX6 ? DISCO_PREC_MX6 : (ar == ARITH_F16X3 ? DISCO_PREC_F16X3 : DISCO_PREC_MX8);
    *channel_disparity = c->mx6_disparity;
    return DISCO_OK;
}

int disco_calibration_entry(disco_ctx* c, int i, const char** key, float* amax, int* sexp) {
    if (!c || !key || !amax || !sexp) { set_error("bad calibration index"); return DISCO_EINVAL; }
    std::lock_guard<std::mutex> lk(c->mu);
    if (i < 0 || i >= (int)c->amax.size()) { set_error("bad calibration index"); return DISCO_EINVAL; }
    auto it = c->amax.begin();
    std::advance(it, i);
    // the key is COPIED into storage of the calling thread: a pointer into c->amax would dangle as soon as the lock is released and another
    // thread's disco_calibrate (or the HourGlass2 rebuild) reinserts the entries (advisor, round 5).  Valid until this thread's next call.
    static thread_local std::string key_copy;
    key_copy = it->first;
    *key = key_copy.c_str(); *amax = it->second;
    auto sx = c->sexp.find(it->first);
    *sexp = sx == c->sexp.end() ? 0 : sx->second;
    return DISCO_OK;
}

int disco_workspace_bytes(disco_ctx* c, int n, int h, int w, int sampled_T, size_t* bytes) {
    if (!bytes) { set_error("null argument"); return DISCO_EINVAL; }
    disco_forward_args a{};
    a.n = n; a.h = h; a.w = w; a.sampled_T = sampled_T; a.test_mode = 1;   // inference needs at least what validation does
    int rc = check_forward_args(c, &a);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);          // the dry plan reads the layer and exponent tables a concurrent disco_calibrate may rebuild
    size_t peak = 0;
    if (c->opt.network) {
        Plan P(c, &a, (size_t)1 << 46, true);
        subnet_stage(P, c, c->opt.network, nullptr, n, h, w, nullptr);
        peak = P.arena.peak; rc = P.rc;
    } else rc = run_plan(c, &a, (size_t)1 << 46, true, &peak);
    *bytes = peak + 4096;
    return rc;
}

// one network of the colorizer on its own: on a full context or on the stand-alone context of that network
static int forward_subnet(disco_ctx* c, int which, const char* entry, int n, int h, int w, const float* d_in, float* d_out, void* d_ws, size_t ws_bytes,
                          void* stream) {
    std::unique_lock<std::mutex> lk;
    if (c) lk = std::unique_lock<std::mutex>(c->mu);
    ProgressDisarm disarm{c, (hipStream_t)stream};
    disco_forward_args a{};
    a.n = n; a.h = h; a.w = w; a.d_workspace = d_ws; a.workspace_bytes = ws_bytes; a.stream = stream;
    if (!c || !c->finalized) { set_error("%s before disco_finalize", entry); return DISCO_ESTATE; }
    if (c->opt.network && c->opt.network != which) { set_error("%s on the stand-alone context of another network", entry); return DISCO_ESTATE; }
    if (n < 1 || h < 16 || w < 16 || h % 16 || w % 16) { set_error("bad input size %dx%dx%d (multiples of 16)", n, h, w); return DISCO_ESHAPE; }
    if (!d_in || !d_out || !d_ws) { set_error("null tensor pointer"); return DISCO_EINVAL; }
    DISCO_HIP_CHECK(hipSetDevice(c->device));
    if (!c->calibrated) { set_error(c->opt.network == SUBNET_ENH ? "stand-alone HourGlass2 context: disco_calibrate on a first batch of its input comes first" : "context used before its calibration pass"); return DISCO_ESTATE; }
    Plan P(c, &a, ws_bytes, false);
    subnet_stage(P, c, which, d_in, n, h, w, d_out);
    return P.rc;
}

int disco_forward_segnet(disco_ctx* c, int n, int h, int w, const float* d_gray, float* d_affinity, void* d_ws, size_t ws_bytes, void* stream) {
    return forward_subnet(c, SUBNET_SEG, "disco_forward_segnet", n, h, w, d_gray, d_affinity, d_ws, ws_bytes, stream);
}
int disco_forward_repnet(disco_ctx* c, int n, int h, int w, const float* d_gray, float* d_feats, void* d_ws, size_t ws_bytes, void* stream) {
    return forward_subnet(c, SUBNET_REP, "disco_forward_repnet", n, h, w, d_gray, d_feats, d_ws, ws_bytes, stream);
}
int disco_forward_enhance(disco_ctx* c, int n, int h, int w, const float* d_input, float* d_out, void* d_ws, size_t ws_bytes, void* stream) {
    return forward_subnet(c, SUBNET_ENH, "disco_forward_enhance", n, h, w, d_input, d_out, d_ws, ws_bytes, stream);
}
int disco_subnet_workspace_bytes(disco_ctx* c, int which, int n, int h, int w, size_t* bytes) {
    if (!c || !bytes || !c->finalized) { set_error("disco_subnet_workspace_bytes: bad argument / context not finalized"); return DISCO_EINVAL; }
    if (which < SUBNET_SEG || which > SUBNET_ENH || (c->opt.network && c->opt.network != which)) { set_error("network %d is not in this context", which); return DISCO_EINVAL; }
    if (n < 1 || h < 16 || w < 16 || h % 16 || w % 16) { set_error("bad input size %dx%dx%d (multiples of 16)", n, h, w); return DISCO_ESHAPE; }
    std::lock_guard<std::mutex> lk(c->mu);
    disco_forward_args a{};
    a.n = n; a.h = h; a.w = w;
    Plan P(c, &a, (size_t)1 << 46, true);
    subnet_stage(P, c, which, nullptr, n, h, w, nullptr);
    *bytes = P.arena.peak + 4096;
    return P.rc;
}

int disco_forward(disco_ctx* c, const disco_forward_args* a) {
    std::unique_lock<std::mutex> lk;
    if (c) lk = std::unique_lock<std::mutex>(c->mu);
    ProgressDisarm disarm{c, a ? (hipStream_t)a->stream : nullptr};
    if (c && c->opt.network) { set_error("stand-alone network context: use disco_forward_segnet / _repnet / _enhance"); return DISCO_ESTATE; }
    int rc = check_forward_args(c, a);
    if (rc) return rc;
    if (!a->d_gray || !a->d_ab || !a->d_pal_logit || !a->d_ref_logit || !a->d_pred_colors || !a->d_affinity ||
        !a->d_spix_colors || !a->d_hint_mask || !a->d_workspace) { set_error("null tensor pointer"); return DISCO_EINVAL; }
    {   // host index arrays address token rows on the device: range-check them here, a bad row would fault the GPU
        const int L = (a->h / c->opt.sp_size) * (a->w / c->opt.sp_size), K = c->opt.n_clusters;
        auto in_range = [&](const int32_t* p, size_t cnt, const char* what) {
            for (size_t i = 0; p && i < cnt; ++i)
                if (p[i] < 0 || p[i] >= L) { set_error("%s[%zu] = %d outside [0, %d)", what, i, p[i], L); return false; }
            return true;
        };
        if (!in_range(c->opt.random_hint ? a->h_hint_pos : a->h_init_idx, (size_t)a->n * K, c->opt.random_hint ? "h_hint_pos" : "h_init_idx") ||
            !in_range(c->opt.random_hint ? nullptr : a->h_fallback_rows, (size_t)a->n * (a->h_fallback_rows ? a->max_fallback : 0), "h_fallback_rows"))
            return DISCO_EINVAL;
    }
    DISCO_HIP_CHECK(hipSetDevice(c->device));
    return run_plan(c, a, a->workspace_bytes, false, nullptr);      // (ProgressDisarm: fewer conv launches than asked for, or an error)
}

int disco_set_progress_event(disco_ctx* c, void* event, int after_conv_launches) {
    if (!c || after_conv_launches < 0) { set_error("disco_set_progress_event: bad argument"); return DISCO_EINVAL; }
    std::lock_guard<std::mutex> lk(c->mu);
    c->progress_ev = (hipEvent_t)event;
    c->progress_after = after_conv_launches;
    c->progress_seen = 0;
    return DISCO_OK;
}

int disco_set_debug_checksums(disco_ctx* c, void* d_table, int rows, int cols) {
    if (!c || rows < 0 || cols < 0) { set_error("disco_set_debug_checksums: bad argument"); return DISCO_EINVAL; }
    std::lock_guard<std::mutex> lk(c->mu);
    c->d_dbg = (unsigned long long*)d_table; c->dbg_rows = d_table ? rows : 0; c->dbg_cols = cols; c->dbg_seq = 0;
    return DISCO_OK;
}

int disco_set_debug_dump(disco_ctx* c, void* d_buf, size_t bytes_per_row) {
    if (!c) return DISCO_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    c->d_dump = (char*)d_buf; c->dump_stride = d_buf ? bytes_per_row : 0;
    return DISCO_OK;
}

int disco_sync(void* stream) {
    DISCO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return DISCO_OK;
}

int disco_set_profiling(disco_ctx* c, int level) {
    if (!c) return DISCO_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    c->profiling = level;
    return DISCO_OK;
}

int disco_profile_conv(disco_ctx* c, int* launches, float* total_ms, double* total_flops) {
    if (!c || !launches || !total_ms || !total_flops) { set_error("null argument"); return DISCO_EINVAL; }
    *launches = 0; *total_ms = 0.f; *total_flops = 0.0;
    for (auto& e : c->conv_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.e0, e.e1) != hipSuccess) continue;
        *launches += 1; *total_ms += ms; *total_flops += e.flops;
    }
    return DISCO_OK;
}

int disco_profile_conv_bytes(disco_ctx* c, double* total_bytes) {
    if (!c || !total_bytes) { set_error("null argument"); return DISCO_EINVAL; }
    *total_bytes = 0.0;
    for (auto& e : c->conv_prof) *total_bytes += e.bytes;
    return DISCO_OK;
}

int disco_profile_conv_entry(disco_ctx* c, int i, const char** key, float* ms, double* flops) {
    if (!c || i < 0 || i >= (int)c->conv_prof.size() || !key || !ms || !flops) { set_error("bad conv profile index"); return DISCO_EINVAL; }
    auto& e = c->conv_prof[i];
    *key = e.key.c_str(); *flops = e.flops; *ms = -1.f;
    hipEventElapsedTime(ms, e.e0, e.e1);
    return DISCO_OK;
}

int disco_profile_count(disco_ctx* c) {
    if (!c || c->prof.size() < 2) return 0;
    c->prof_ms.clear(); c->prof_flops.clear();
    for (size_t i = 1; i < c->prof.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->prof[i - 1].ev, c->prof[i].ev) != hipSuccess) ms = -1.f;
        c->prof_ms.push_back({c->prof[i].name, ms});
        c->prof_flops.push_back(c->prof[i].flops);
    }
    return (int)c->prof_ms.size();
}

int disco_profile_entry(disco_ctx* c, int i, const char** name, float* ms, double* flops) {
    if (!c || i < 0 || i >= (int)c->prof_ms.size()) { set_error("bad profile index"); return DISCO_EINVAL; }
    *name = c->prof_ms[i].first.c_str(); *ms = c->prof_ms[i].second; *flops = c->prof_flops[i];
    return DISCO_OK;
}

// ---- operator-level entry points ---------------------------------------------------------------------------------

int disco_op_nchw_to_act(const float* d_src, void* d_dst, int n, int ch, int h, int w, int c_pad, void* stream) {
    if (!positive("nchw_to_act", {n, ch, h, w, c_pad})) return DISCO_ESHAPE;
    if (!d_src || !d_dst || c_pad < ch || c_pad % 16) { set_error("bad argument (c_pad must be a multiple of 16 >= c)"); return DISCO_EINVAL; }
    return launch_nchw_to_act(d_src, (f16*)d_dst, (long)n * h * w * c_pad, n, ch, h, w, c_pad, (hipStream_t)stream);
}
int disco_op_act_to_nchw(const void* d_src, float* d_dst, int n, int ch, int h, int w, int c_pad, void* stream) {
    if (!positive("act_to_nchw", {n, ch, h, w, c_pad})) return DISCO_ESHAPE;
    if (!d_src || !d_dst || c_pad < ch || c_pad % 16) { set_error("bad argument (c_pad must be a multiple of 16 >= c)"); return DISCO_EINVAL; }
    return launch_act_to_nchw((const f16*)d_src, (long)n * h * w * c_pad, d_dst, n, ch, h, w, c_pad, (hipStream_t)stream);
}

int disco_op_conv3x3_pack(const float* h_w, int c_out, int c_in, void* d_packed, size_t* bytes) {
    if (!bytes) { set_error("null bytes"); return DISCO_EINVAL; }
    const int cpad = round_up(c_in, 16);
    *bytes = conv3x3_packed_bytes(c_out, cpad);
    if (!d_packed) return DISCO_OK;
    if (!h_w) { set_error("null weight"); return DISCO_EINVAL; }
    std::vector<char> packed(*bytes);
    conv3x3_pack_host(h_w, c_out, c_in, nullptr, cpad, packed.data());
    DISCO_HIP_CHECK(hipMemcpy(d_packed, packed.data(), packed.size(), hipMemcpyHostToDevice));
    return DISCO_OK;
}

int disco_op_conv3x3(const disco_conv_desc* d, const void* d_src0, const void* d_src1, const void* d_packed_w,
                     const float* d_bias, const float* d_bn_scale, const float* d_bn_shift, const void* d_res, void* d_out,
                     void* stream) {
    if (!d || !d_src0 || !d_packed_w || !d_out) { set_error("null argument"); return DISCO_EINVAL; }
    if (!positive("conv3x3", {d->n, d->h_in, d->w_in, d->c_in0, d->c_out}) || d->c_in1 < 0) { if (d->c_in1 < 0) set_error("conv3x3: c_in1 %d", d->c_in1); return DISCO_ESHAPE; }
    if (d->c_in0 % 16 || d->c_in1 % 16) { set_error("conv3x3 op: source channels must be multiples of 16"); return DISCO_ESHAPE; }
    ConvArgs ca{};
    const int h0 = d->up0 ? d->h_in / 2 : d->h_in, w0 = d->up0 ? d->w_in / 2 : d->w_in;
    ca.src[0] = {(const f16*)d_src0, (long)d->n * h0 * w0 * d->c_in0, d->c_in0, h0, w0, d->up0, d->sexp_in};
    ca.nsrc = 1;
    if (d->c_in1) {
        if (!d_src1) { set_error("null second source"); return DISCO_EINVAL; }
        const int h1 = d->up1 ? d->h_in / 2 : d->h_in, w1 = d->up1 ? d->w_in / 2 : d->w_in;
        ca.src[1] = {(const f16*)d_src1, (long)d->n * h1 * w1 * d->c_in1, d->c_in1, h1, w1, d->up1, d->sexp_in};
        ca.nsrc = 2;
    }
    ca.n = d->n; ca.h_in = d->h_in; ca.w_in = d->w_in; ca.c_in = d->c_in0 + d->c_in1;
    ca.stride = d->stride; ca.h_out = (d->h_in - 1) / d->stride + 1; ca.w_out = (d->w_in - 1) / d->stride + 1;
    ca.w = (const f16*)d_packed_w; ca.c_out = d->c_out; ca.c_out_pad = d->c_out;
    ca.bias = d_bias; ca.bn_scale = d_bn_scale; ca.bn_shift = d_bn_shift;
    ca.out = (f16*)d_out; ca.out_plane = (long)d->n * ca.h_out * ca.w_out * d->c_out;
    ca.res = (const f16*)d_res; ca.res_plane = ca.out_plane; ca.res_sexp = d->sexp_res; ca.out_sexp = d->sexp_out;
    ca.act = d->act; ca.slope = d->slope; ca.precision = d->precision;
    if (d->precision != DISCO_PREC_F16X3) { set_error("conv3x3 op: precision %d (the f16x3 arithmetic only; the fp16+fp8 ones go through disco_op_conv3x3_mx)", d->precision); return DISCO_EINVAL; }
    return run_conv(ca, (hipStream_t)stream);
}

static Act flat_act(const void* p, int n, int c_pad, int h, int w, int planes, int sexp) {
    Act t; t.p = (f16*)p; t.n = n; t.h = h; t.w = w; t.c = c_pad; t.sexp = sexp;
    const size_t el = t.elems();
    t.plane = (planes & DISCO_PLANE_LO) ? el : 0;
    t.q_off = (planes & (DISCO_PLANE_Q | DISCO_PLANE_QL | DISCO_PLANE_Q6)) ? el * 2 * ((planes & DISCO_PLANE_LO) ? 2 : 1) : 0;
    t.q_kind = (planes & DISCO_PLANE_QL) ? 1 : ((planes & DISCO_PLANE_Q6) ? 2 : 0);
    return t;
}

int disco_op_act_bytes(int n, int c_pad, int h, int w, int planes, size_t* bytes) {
    if (!bytes || !positive("act_bytes", {n, c_pad, h, w})) return DISCO_EINVAL;
    *bytes = flat_act(nullptr, n, c_pad, h, w, planes, 0).bytes();
    return DISCO_OK;
}

int disco_op_nchw_to_act_mx(const float* d_src, void* d_dst, int n, int ch, int h, int w, int c_pad, int planes, int sexp, void* stream) {
    if (!positive("nchw_to_act_mx", {n, ch, h, w, c_pad})) return DISCO_ESHAPE;
    if (d_dst && (planes & DISCO_PLANE_Q6)) {       // fp6 fields are OR-ed into their slots
        const Act t = flat_act(d_dst, n, c_pad, h, w, planes, sexp);
        DISCO_HIP_CHECK(hipMemsetAsync((char*)d_dst + t.q_off, 0, t.q_bytes(), (hipStream_t)stream));
    }
    if (!d_src || !d_dst || c_pad < ch || c_pad % ((planes & (DISCO_PLANE_Q | DISCO_PLANE_QL | DISCO_PLANE_Q6)) ? 32 : 16) || ((planes & DISCO_PLANE_Q ? 1 : 0) + (planes & DISCO_PLANE_QL ? 1 : 0) + (planes & DISCO_PLANE_Q6 ? 1 : 0) > 1)) { set_error("bad argument (c_pad must be a multiple of 16, 32 with q planes, >= c)"); return DISCO_EINVAL; }
    return launch_nchw_to_act_mx(d_src, flat_act(d_dst, n, c_pad, h, w, planes, sexp), ch, (hipStream_t)stream);
}

int disco_op_act_mx_to_nchw(const void* d_src, float* d_dst, int n, int ch, int h, int w, int c_pad, int planes, int sexp, int which, void* stream) {
    if (!positive("act_mx_to_nchw", {n, ch, h, w, c_pad})) return DISCO_ESHAPE;
    if (!d_src || !d_dst || c_pad < ch) { set_error("bad argument"); return DISCO_EINVAL; }
    const Act t = flat_act(d_src, n, c_pad, h, w, planes, sexp);
    if (which == 0) {
        if (!t.plane) { set_error("act_mx_to_nchw: which = 0 needs the lo plane"); return DISCO_EINVAL; }
        return launch_act_to_nchw(t.p, (long)t.plane, d_dst, n, ch, h, w, c_pad, (hipStream_t)stream, sexp);
    }
    if (!t.q_off || which < 1 || which > 2) { set_error("act_mx_to_nchw: which %d / planes %d", which, planes); return DISCO_EINVAL; }
    return launch_act_q_to_nchw(t, d_dst, ch, which - 1, (hipStream_t)stream);
}

int disco_op_conv3x3_mx_pack(const float* h_w, int c_out, int c_in, int x2q, void* d_packed, int32_t* d_wexp, size_t* bytes) {
    if (!bytes) { set_error("null bytes"); return DISCO_EINVAL; }
    if (x2q < 0 || x2q > 3) { set_error("conv3x3_mx_pack: variant %d", x2q); return DISCO_EINVAL; }
    // variant 3: the f16+fp8x2 arithmetic with the LAST input channel in the kernel's 16-channel fp16 tail chunk as (x_hi, x_lo, x_hi)
    // against (w_h, w_h, w_l) (disco_op_gray_tail builds that source; the forward's HourGlass2 input layer)
    const bool tail = x2q == 3;
    if (tail && (c_in < 33 || (c_in - 1) % 32)) { set_error("conv3x3_mx_pack: the tail variant takes 32 k + 1 input channels (got %d)", c_in); return DISCO_ESHAPE; }
    const int cpad = tail ? c_in - 1 + 16 : round_up(c_in, x2q == 1 ? 64 : 32);
    const int variant = tail ? 0 : x2q;
    *bytes = conv_mx_packed_bytes(c_out, cpad, variant);
    if (!d_packed) return DISCO_OK;
    if (!h_w || !d_wexp) { set_error("null weight"); return DISCO_EINVAL; }
    std::vector<char> packed(*bytes);
    std::vector<int32_t> wexp((size_t)round_up(c_out, 32));
    std::vector<int> map;
    if (tail) {
        map.assign(cpad, -1);
        for (int i = 0; i < c_in; ++i) map[i] = i;
        map[c_in] = c_in - 1; map[c_in + 1] = CONV_MX_LO_OF(c_in - 1);
    }
    conv_mx_pack_host(h_w, c_out, c_in, tail ? map.data() : nullptr, cpad, packed.data(), wexp.data(), variant);
    DISCO_HIP_CHECK(hipMemcpy(d_packed, packed.data(), packed.size(), hipMemcpyHostToDevice));
    DISCO_HIP_CHECK(hipMemcpy(d_wexp, wexp.data(), wexp.size() * 4, hipMemcpyHostToDevice));
    return DISCO_OK;
}

int disco_op_gray_tail(const float* d_gray, void* d_out, int n, int h, int w, int sexp, void* stream) {
    if (!d_gray || !d_out) { set_error("null argument"); return DISCO_EINVAL; }
    if (!positive("gray_tail", {n, h, w})) return DISCO_ESHAPE;
    Act o = flat_act(d_out, n, 16, h, w, 0, sexp);
    return launch_gray_tail(d_gray, 1, o, (hipStream_t)stream);
}

int disco_op_conv3x3_mx(const disco_conv_mx_desc* d, const void* d_src0, const void* d_src1, const void* d_packed_w, const int32_t* d_wexp,
                        const float* d_bias, const float* d_bn_scale, const float* d_bn_shift, const void* d_res, void* d_out,
                        uint32_t* d_sat, const uint32_t* d_tapmask, void* stream) {
    if (!d || !d_src0 || !d_packed_w || !d_wexp || !d_out) { set_error("null argument"); return DISCO_EINVAL; }
    if (!positive("conv3x3_mx", {d->n, d->h_in, d->w_in, d->c_in0, d->c_out}) || d->c_in1 < 0) { if (d->c_in1 < 0) set_error("conv3x3_mx: c_in1 %d", d->c_in1); return DISCO_ESHAPE; }
    ConvMxArgs ca{};
    const int h0 = d->up0 ? d->h_in / 2 : d->h_in, w0 = d->up0 ? d->w_in / 2 : d->w_in;
    if (d->x2q && d->q6) { set_error("conv3x3_mx op: x2q and q6 are different arithmetics"); return DISCO_EINVAL; }
    const int src_planes = d->x2q ? DISCO_PLANE_QL : (d->q6 ? DISCO_PLANE_Q6 : DISCO_PLANE_Q);
    const Act s0 = flat_act(d_src0, d->n, d->c_in0, h0, w0, src_planes, d->sexp0);
    if (s0.q_off >= ((size_t)1 << 32)) { set_error("conv3x3_mx: source too large"); return DISCO_ESHAPE; }
    ca.src[0] = {s0.p, (uint32_t)s0.q_off, d->c_in0, h0, w0, d->up0, d->sexp0};
    ca.nsrc = 1;
    if (d->c_in1) {
        if (!d_src1) { set_error("null second source"); return DISCO_EINVAL; }
        const int h1 = d->up1 ? d->h_in / 2 : d->h_in, w1 = d->up1 ? d->w_in / 2 : d->w_in;
        // 16 channels: the fp16 tail source of a two-source f16+fp8x2 layer (hi plane only)
        const Act s1 = flat_act(d_src1, d->n, d->c_in1, h1, w1, (d->c_in1 == 16 && !d->x2q && !d->q6) ? 0 : src_planes, d->sexp1);
        if (s1.q_off >= ((size_t)1 << 32)) { set_error("conv3x3_mx: source too large"); return DISCO_ESHAPE; }
        ca.src[1] = {s1.p, (uint32_t)s1.q_off, d->c_in1, h1, w1, d->up1, d->sexp1};
        ca.nsrc = 2;
    }
    ca.n = d->n; ca.h_in = d->h_in; ca.w_in = d->w_in; ca.c_in = d->c_in0 + d->c_in1;
    ca.stride = d->stride; ca.h_out = (d->h_in - 1) / d->stride + 1; ca.w_out = (d->w_in - 1) / d->stride + 1;
    ca.w = d_packed_w; ca.wexp = d_wexp; ca.c_out = d->c_out; ca.c_out_pad = d->c_out;
    ca.bias = d_bias; ca.bn_scale = d_bn_scale; ca.bn_shift = d_bn_shift;
    if (d->d2s && (d->out_f32 || d->stride != 1 || d->c_out % 128)) { set_error("conv3x3_mx op: depth-to-space needs an activation output, stride 1, c_out = 4 C with C a multiple of 32"); return DISCO_ESHAPE; }
    if (d->out_f32) ca.out_f32 = (float*)d_out;
    else {
        const Act o = d->d2s ? flat_act(d_out, d->n, d->c_out / 4, 2 * ca.h_out, 2 * ca.w_out, d->out_planes, d->out_sexp)
                             : flat_act(d_out, d->n, d->c_out, ca.h_out, ca.w_out, d->out_planes, d->out_sexp);
        if (d->d2s) ca.d2s_c = d->c_out / 4;
        ca.out = o.p; ca.out_plane = (long)o.plane; ca.out_q_off = o.q_off; ca.out_sexp = d->out_sexp; ca.out_q_kind = o.q_kind;
    }
    ca.tapmask = d_tapmask;
    if (d_res) {
        const Act rr = d->d2s ? flat_act(d_res, d->n, d->c_out / 4, 2 * ca.h_out, 2 * ca.w_out, d->res_planes, 0)
                              : flat_act(d_res, d->n, d->c_out, ca.h_out, ca.w_out, d->res_planes, 0);
        ca.res = rr.p; ca.res_plane = (long)rr.plane; ca.res_sexp = d->res_sexp;
    }
    ca.act = d->act; ca.slope = d->slope; ca.sat = d_sat; ca.x2q = d->x2q ? 1 : 0; ca.q6 = d->q6 ? 1 : 0;
    return launch_conv3x3_mx(ca, (hipStream_t)stream);
}


int disco_op_conv3x3_tapmask(const float* h_w, int c_out, int c_in, uint32_t* d_mask) {
    if (!h_w || !d_mask || c_out <= 0 || c_in <= 0) { set_error("conv3x3_tapmask: bad argument"); return DISCO_EINVAL; }
    std::vector<uint32_t> mask(cdiv(c_out, 32));
    conv3x3_tapmask_host(h_w, c_out, c_in, mask.data());
    DISCO_HIP_CHECK(hipMemcpy(d_mask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
    return DISCO_OK;
}

int disco_diag_mfma_rate(int mode, int iters, double* tflops) { return diag_mfma_rate(mode, iters, tflops); }

int disco_op_deconv4x4_pack(const float* h_w, int c_in, int c_out, void* d_packed, size_t* bytes) {
    if (!bytes) { set_error("null bytes"); return DISCO_EINVAL; }
    const int cpad = round_up(c_in, 16);
    *bytes = conv3x3_packed_bytes(4 * c_out, cpad);
    if (!d_packed) return DISCO_OK;
    if (!h_w) { set_error("null weight"); return DISCO_EINVAL; }
    std::vector<float> w3((size_t)4 * c_out * c_in * 9);
    deconv_as_conv3x3_host(h_w, c_in, c_out, w3.data());
    std::vector<char> packed(*bytes);
    conv3x3_pack_host(w3.data(), 4 * c_out, c_in, nullptr, cpad, packed.data());
    DISCO_HIP_CHECK(hipMemcpy(d_packed, packed.data(), packed.size(), hipMemcpyHostToDevice));
    return DISCO_OK;
}

int disco_op_deconv4x4(const void* d_src, const void* d_packed_w, const float* d_bias, void* d_out, int n, int h_in, int w_in,
                       int c_in, int c_out, float slope, int precision, void* stream) {
    if (!positive("deconv4x4", {n, h_in, w_in, c_in, c_out})) return DISCO_ESHAPE;
    if (!d_src || !d_packed_w || !d_bias || !d_out) { set_error("null argument"); return DISCO_EINVAL; }
    if (c_in % 16 || (4 * c_out) % 64) { set_error("deconv4x4: c_in %% 16 and c_out %% 16 required"); return DISCO_ESHAPE; }
    ConvArgs ca{};
    ca.src[0] = {(const f16*)d_src, (long)n * h_in * w_in * c_in, c_in, h_in, w_in, 0, 0};
    ca.nsrc = 1; ca.n = n; ca.h_in = h_in; ca.w_in = w_in; ca.c_in = c_in; ca.stride = 1; ca.h_out = h_in; ca.w_out = w_in;
    ca.w = (const f16*)d_packed_w; ca.c_out = 4 * c_out; ca.c_out_pad = 4 * c_out; ca.bias = d_bias;
    ca.out = (f16*)d_out; ca.out_plane = (long)n * 4 * h_in * w_in * c_out; ca.d2s_c = c_out;
    ca.act = DISCO_ACT_LRELU; ca.slope = slope; ca.precision = precision;
    if (precision != DISCO_PREC_F16X3) { set_error("deconv4x4 op: precision %d", precision); return DISCO_EINVAL; }
    return run_conv(ca, (hipStream_t)stream);      // the forward's own path: conv3x3_mx_kernel AR = 2 with the depth-to-space epilogue
}

int disco_op_poolfeat(const float* d_feat, const float* d_prob, float* d_pooled, float* d_conf, float* d_sizes, int n, int ch,
                      int h, int w, int sp, void* d_ws, size_t ws_bytes, void* stream) {
    if (!positive("poolfeat", {n, ch, h, w, sp})) return DISCO_ESHAPE;
    if (!d_feat || !d_prob || !d_ws) { set_error("null argument"); return DISCO_EINVAL; }
    if (ws_bytes < poolfeat_ws_bytes(n, ch, h, w, sp)) { set_error("poolfeat workspace too small"); return DISCO_ENOMEM; }
    PoolArgs pa{};
    pa.feat_act = nullptr; pa.c_act = 0; pa.feat_nchw = d_feat; pa.c_nchw = ch; pa.prob = d_prob;
    const size_t cells = (size_t)n * (h / sp) * (w / sp);
    pa.partial = (float*)d_ws; pa.cnt = (float*)d_ws + cells * 9 * (ch + 1);
    pa.tok_out = nullptr; pa.c_tok = 0; pa.nchw_out = d_pooled; pa.c_from = 0;
    pa.conf = d_conf; pa.sizes = d_sizes; pa.n = n; pa.H = h; pa.W = w; pa.sp = sp;
    return launch_poolfeat(pa, (hipStream_t)stream);
}

int disco_op_upfeat(const float* d_tok, const float* d_prob, float* d_out, int n, int ch, int h, int w, int sp, void* stream) {
    if (!positive("upfeat", {n, ch, h, w, sp})) return DISCO_ESHAPE;
    if (!d_tok || !d_prob || !d_out) { set_error("null argument"); return DISCO_EINVAL; }
    return launch_upfeat(d_tok, 0, d_prob, 1, nullptr, d_out, n, ch, h, w, sp, nullptr, (hipStream_t)stream);
}

size_t disco_op_encoder_weight_floats(void) { return ENC_LAYERS * ENC_LAYER_FLOATS; }

int disco_op_encoder_stack(const float* d_x, const float* d_pos, const float* d_weights, float* d_out, int n, int l, void* d_ws,
                           size_t ws_bytes, void* stream) {
    return disco_op_encoder_stack_masked(d_x, d_pos, d_weights, nullptr, d_out, n, l, d_ws, ws_bytes, stream);
}

int disco_op_encoder_stack_masked(const float* d_x, const float* d_pos, const float* d_weights, const float* d_key_sizes, float* d_out, int n,
                                  int l, void* d_ws, size_t ws_bytes, void* stream) {
    if (!positive("encoder_stack", {n, l})) return DISCO_ESHAPE;
    if (!d_x || !d_pos || !d_weights || !d_out || !d_ws) { set_error("null argument"); return DISCO_EINVAL; }
    if (ws_bytes < encoder_ws_bytes(n, l)) { set_error("encoder workspace too small (%zu < %zu)", ws_bytes, encoder_ws_bytes(n, l)); return DISCO_ENOMEM; }
    // a workspace with room for the weights' B-fragment image behind the stack's own buffers takes the 16-row tail kernel (what the
    // forward does for small token counts); a smaller one the 64-row tiles.  Same results (the tests run both and compare).
    const size_t base = (encoder_ws_bytes(n, l) + 255) & ~(size_t)255, pk = encoder_packed_floats() * sizeof(float);
    if (ws_bytes >= base + pk) {
        float* d_pk = reinterpret_cast<float*>(static_cast<char*>(d_ws) + base);
        if (int rc = launch_encoder_pack(d_weights, d_pk, (hipStream_t)stream)) return rc;
        return launch_encoder_stack(d_x, d_pos, 0, d_weights, d_out, n, l, d_ws, (hipStream_t)stream, nullptr, d_pk, d_key_sizes, 1);
    }
    return launch_encoder_stack(d_x, d_pos, 0, d_weights, d_out, n, l, d_ws, (hipStream_t)stream, nullptr, nullptr, d_key_sizes, 1);
}

int disco_op_kmeans_anchors(const float* d_x, const float* d_sizes, const int32_t* d_init_idx, const int32_t* d_fallback_rows,
                            int max_fallback, int32_t* d_assign, int32_t* d_anchor, float* d_hint_mask, int32_t* d_info, int n,
                            int l, int k, int d, int channel_major, void* stream) {
    return disco_op_kmeans_anchors_ws(d_x, d_sizes, d_init_idx, d_fallback_rows, max_fallback, d_assign, d_anchor, d_hint_mask, d_info, n, l, k, d,
                                      channel_major, nullptr, 0, stream);
}

size_t disco_op_kmeans_workspace_bytes(int n, int l) { return (n > 0 && l > 0) ? kmeans_ws_bytes(n, l) : 0; }

int disco_op_kmeans_anchors_ws(const float* d_x, const float* d_sizes, const int32_t* d_init_idx, const int32_t* d_fallback_rows,
                               int max_fallback, int32_t* d_assign, int32_t* d_anchor, float* d_hint_mask, int32_t* d_info, int n,
                               int l, int k, int d, int channel_major, void* d_ws, size_t ws_bytes, void* stream) {
    if (!positive("kmeans_anchors", {n, l, k, d})) return DISCO_ESHAPE;
    if (!d_x || !d_sizes || !d_init_idx || !d_assign || !d_anchor || !d_hint_mask) { set_error("null argument"); return DISCO_EINVAL; }
    return launch_kmeans_anchors(d_x, d_sizes, d_init_idx, d_fallback_rows, max_fallback, d_assign, d_anchor, d_hint_mask, d_info,
                                 n, l, k, (hipStream_t)stream, d, channel_major, d_ws, ws_bytes);
}

int disco_op_kmeans_fallbacks(const void* d_ws, int n, int l, void* stream, int* count) {
    if (!d_ws || !count || n <= 0 || l <= 0) { set_error("disco_op_kmeans_fallbacks: bad argument"); return DISCO_EINVAL; }
    *count = 0;
    if (kmeans_ws_bytes(n, l) == 0) return DISCO_OK;
    std::vector<int> st((size_t)n);
    DISCO_HIP_CHECK(hipMemcpy2DAsync(st.data(), sizeof(int), static_cast<const unsigned char*>(d_ws) + kmeans_state_offset(), kmeans_image_stride(),
                                     sizeof(int), (size_t)n, hipMemcpyDeviceToHost, (hipStream_t)stream));
    DISCO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    for (int v : st) *count += (v >> 30) & 1;
    return DISCO_OK;
}

static int gamut_device(float** out) {
    static float* table[DISCO_MAX_DEVICES] = {};     // the 313-bin table of the op-level entry points, one per device
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    float*& d = table[current_device()];
    if (!d) {
        std::vector<float> q;
        for (auto& r : GAMUT_RUNS) for (int b = r[1]; b <= r[2]; b += 10) { q.push_back((float)r[0]); q.push_back((float)b); }
        DISCO_HIP_CHECK(hipMalloc((void**)&d, q.size() * 4));
        DISCO_HIP_CHECK(hipMemcpy(d, q.data(), q.size() * 4, hipMemcpyHostToDevice));
    }
    *out = d;
    return DISCO_OK;
}

int disco_op_select_colors(const float* d_logit, float* d_colors, int32_t* d_labels, int n, int hw, int t, void* stream) {
    if (!positive("select_colors", {n, hw})) return DISCO_ESHAPE;
    if (!d_logit || !d_colors || t < 0 || t > 2) { set_error("bad argument"); return DISCO_EINVAL; }
    float* q = nullptr;
    int rc = gamut_device(&q);
    if (rc) return rc;
    return launch_select_colors(d_logit, q, d_colors, d_labels, n, hw, t, 1, (hipStream_t)stream);
}

int disco_op_nearest_bin(const float* d_ab, int32_t* d_labels, int n, int hw, void* stream) {
    if (!positive("nearest_bin", {n, hw})) return DISCO_ESHAPE;
    if (!d_ab || !d_labels) { set_error("null argument"); return DISCO_EINVAL; }
    float* q = nullptr;
    int rc = gamut_device(&q);
    if (rc) return rc;
    return launch_nearest_bin(d_ab, q, d_labels, n, hw, (hipStream_t)stream);
}

int disco_op_decode_ind2ab(const float* d_logit, float* d_ab, int n, int hw, int T, void* stream) {
    if (!positive("decode_ind2ab", {n, hw})) return DISCO_ESHAPE;
    if (!d_logit || !d_ab) { set_error("null argument"); return DISCO_EINVAL; }
    if (T < 0 || T > 9) { set_error("decode_ind2ab: integer T in [0,9] supported, got %d", T); return DISCO_EUNSUPPORTED; }
    float* q = nullptr;
    int rc = gamut_device(&q);
    if (rc) return rc;
    return launch_select_colors(d_logit, q, d_ab, nullptr, n, hw, 0, 1, (hipStream_t)stream, T);
}

int disco_op_decode_annealed(const float* d_logit, float* d_ab, int n, int hw, float T, void* stream) {
    if (!d_logit || !d_ab || n < 1 || hw < 1) { set_error("bad argument"); return DISCO_EINVAL; }
    float* q = nullptr;
    int rc = gamut_device(&q);
    if (rc) return rc;
    return launch_decode_annealed(d_logit, q, d_ab, n, hw, T, (hipStream_t)stream);
}

int disco_op_rgb2lab(const float* d_rgb, float* d_lab, int n, int h, int w, void* stream) {
    if (!d_rgb || !d_lab || n < 1 || h < 1 || w < 1) { set_error("bad argument"); return DISCO_EINVAL; }
    return launch_rgb2lab(d_rgb, d_lab, (long)n * h * w, (long)h * w, (hipStream_t)stream);
}

int disco_op_lab2rgb(const float* d_lab, float* d_rgb, int n, int h, int w, void* stream) {
    if (!d_lab || !d_rgb || n < 1 || h < 1 || w < 1) { set_error("bad argument"); return DISCO_EINVAL; }
    return launch_lab2rgb(d_lab, d_rgb, (long)n * h * w, (long)h * w, (hipStream_t)stream);
}

int disco_op_rgb8_to_lab(const uint8_t* d_rgb8, float* d_gray, float* d_ab, float* d_rgb, int n, int h, int w, int hp, int wp,
                         void* stream) {
    if (!d_rgb8 || !d_gray || !d_ab) { set_error("null argument"); return DISCO_EINVAL; }
    return launch_rgb8_to_lab(d_rgb8, d_gray, d_ab, d_rgb, n, h, w, hp, wp, (hipStream_t)stream);
}

int disco_op_rgb8_resize_to_lab(const uint8_t* d_rgb8, uint8_t* d_resized, float* d_gray, float* d_ab, float* d_rgb, int n, int h, int w,
                                int ho, int wo, void* stream) {
    if (!d_rgb8 || !d_gray || !d_ab) { set_error("null argument"); return DISCO_EINVAL; }
    return launch_rgb8_resize_to_lab(d_rgb8, d_resized, d_gray, d_ab, d_rgb, n, h, w, ho, wo, (hipStream_t)stream);
}

int disco_op_lab_to_rgb8(const float* d_lab, uint8_t* d_rgb8, int n, int hp, int wp, int h, int w, void* stream) {
    if (!d_lab || !d_rgb8) { set_error("null argument"); return DISCO_EINVAL; }
    return launch_lab_to_rgb8(d_lab, d_rgb8, n, hp, wp, h, w, (hipStream_t)stream);
}

int disco_op_mark_color_hints(const float* d_gray, const float* d_target_ab, const float* d_gate, const float* d_base_ab,
                              float* d_out, int n, int h, int w, int kernel_size, void* stream) {
    if (!d_gray || !d_target_ab || !d_gate || !d_out || n < 1 || h < 1 || w < 1) { set_error("bad argument"); return DISCO_EINVAL; }
    return launch_mark_hints(d_gray, d_target_ab, d_gate, d_base_ab, d_out, n, h, w, kernel_size, (hipStream_t)stream);
}

int disco_op_position_encoding(float* d_pos, int h, int w, void* stream) {
    if (!positive("position_encoding", {h, w})) return DISCO_ESHAPE;
    if (!d_pos || h < 1 || w < 1) { set_error("bad argument"); return DISCO_EINVAL; }
    std::vector<float> p((size_t)h * w * 64);
    position_encoding_host(p.data(), h, w);
    DISCO_HIP_CHECK(hipMemcpyAsync(d_pos, p.data(), p.size() * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
    DISCO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return DISCO_OK;
}

}  // extern "C"
