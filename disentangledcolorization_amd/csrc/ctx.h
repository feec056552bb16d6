// ctx.h - what the translation units behind the C ABI (include/disco_hip.h) share: the context, the checkpoint layout, the packed-layer
// records and the internal entry points of each part.  Round 6 split api.cpp (2 200 lines) by concern:
//   api_load.cpp    disco_create / load_tensor / finalize: strict layout check, spectral-norm / batch-norm folding, weight packing, channel levelling
//   api_plan.cpp    the forward plan (Plan, the three network stages, run_plan) and the forward entry points
//   api_calib.cpp   the calibration pass (activation exponents), range / fallback counters, calibration record
//   api_diag.cpp    profiling and debugging hooks
//   api_ops.cpp     the operator-level entry points (one op per call: what the -m gpu parity tests drive)
#pragma once

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

namespace disco_api {


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
inline const Layout& layout(bool hint2regress = false) {
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

}  // namespace disco_api
using namespace disco_api;

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

namespace disco_api {

// which network a context holds (disco_options.network)
enum { SUBNET_FULL = 0, SUBNET_SEG = 1, SUBNET_REP = 2, SUBNET_ENH = 3 };
inline const char* subnet_prefix(int which) { return which == SUBNET_SEG ? "segnet.net." : which == SUBNET_REP ? "repnet." : which == SUBNET_ENH ? "enhanceNet." : ""; }
inline size_t subnet_out_channels(int which) { return which == SUBNET_SEG ? 9 : which == SUBNET_REP ? 64 : 2; }
// the conv arithmetic of a layer (arith_of): which instantiation family of conv3x3_mx_kernel serves it
enum { ARITH_F16X3 = 0, ARITH_MX8 = 1, ARITH_X2Q = 2, ARITH_MX6 = 3 };
inline int run_conv(const ConvArgs& ca, hipStream_t s) { return launch_conv3x3_x3(ca, s); }

// ---- api_load.cpp
int dev_alloc(disco_ctx* c, size_t bytes, void** out);
int upload(disco_ctx* c, const void* h, size_t bytes, void** out);
template <class T>
int upload_vec(disco_ctx* c, const std::vector<T>& v, T** out) { return upload(c, v.data(), v.size() * sizeof(T), (void**)out); }
int staged_h2d(disco_ctx* c, void* d_dst, const void* h_src, size_t bytes, hipStream_t s);
bool any_mx(const disco_ctx* c);
int arith_of(const disco_ctx* c, const std::string& key);
bool use_mx(const disco_ctx* c, const std::string& key);
int pad_cout_mx(int co);
int get_pos(disco_ctx* c, int h, int w, float** out);
int make_enhance(disco_ctx* c);
bool plan_equalisation(disco_ctx* c);
int enhance_disparity_guard(disco_ctx* c, const float* d_user_gray = nullptr, int un = 0, int uh = 0, int uw = 0);
// ---- api_plan.cpp
int run_plan(disco_ctx* c, const disco_forward_args* a, size_t cap, bool dry, size_t* peak, bool calib = false);
int check_forward_args(disco_ctx* c, const disco_forward_args* a);
// ---- api_calib.cpp
int calibrate_ctx(disco_ctx* c, const float* d_user_gray = nullptr, int un = 0, int uh = 0, int uw = 0);
// ---- api_ops.cpp
bool positive(const char* op, std::initializer_list<long> dims);

}  // namespace disco_api
