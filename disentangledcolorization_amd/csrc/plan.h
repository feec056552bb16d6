// plan.h - the forward plan shared by api_plan.cpp (the forward entry points) and api_calib.cpp (the calibration pass runs the same plan in
// measuring mode): workspace arena bookkeeping, conv launch helpers with the per-tensor exponents, profiling marks.
#pragma once
#include "ctx.h"

namespace disco_api {

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


// disco_set_progress_event arms ONE forward.  Whatever way the next forward entry point is left - argument error, a segnet-only
// forward (18 conv launches: fewer than most `after` counts), a HIP failure - the event is recorded on the call's stream when there
// is one and the handle is dropped: it must never fire in an unrelated later forward (by then the caller may have destroyed it).
struct ProgressDisarm {
    disco_ctx* c; hipStream_t s;
    ~ProgressDisarm() {
        if (c && c->progress_ev) { hipEventRecord(c->progress_ev, s); c->progress_ev = nullptr; }
    }
};

void segnet_stage(Plan& P, disco_ctx* c, const float* d_gray, int n, int H, int W, float* d_affinity);
void enhance_stage(Plan& P, disco_ctx* c, Act full, Act g16, int out_act, float* d_out);
void subnet_stage(Plan& P, disco_ctx* c, int which, const float* d_in, int n, int H, int W, float* d_out);

}  // namespace disco_api
