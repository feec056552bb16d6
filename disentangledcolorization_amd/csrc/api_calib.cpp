// api_calib.cpp - the calibration pass that fixes one power-of-two exponent per activation tensor, and what reports on it at run time: clamp
// counter, k-means fallback counter, calibration record (split out of api.cpp, round 6).
#include "plan.h"

namespace disco_api {


// The calibration pass of a context (end of disco_finalize): one forward over two synthetic 256x256 images - uniform noise
// and a smooth low-frequency pattern - in which every producer of an activation tensor runs (at least) twice: once to measure the
// tensor's max |x|, once more with the power-of-two scale that measurement fixes (Plan::calibrate).  The scales are
// properties of the checkpoint from then on (deterministic: the inputs are generated here); q-plane clamping at run time is
// counted (disco_saturation_count) so that inputs far outside the calibrated range are noticed.
int calibrate_ctx_impl(disco_ctx* c, const float* d_user_gray, int un, int uh, int uw);
int calibrate_ctx(disco_ctx* c, const float* d_user_gray, int un, int uh, int uw) {
    // a pass that fails midway must not leave half of the tensors on new exponents (with `calibrated` still set from an earlier pass,
    // forwards would then run on a mix of two calibrations): all or nothing
    const auto sexp0 = c->sexp, nat0 = c->sexp_nat; const auto amax0 = c->amax; const auto tie0 = c->tie;
    const int rc = calibrate_ctx_impl(c, d_user_gray, un, uh, uw);
    if (rc) { c->sexp = sexp0; c->sexp_nat = nat0; c->amax = amax0; c->tie = tie0; }
    return rc;
}
int calibrate_ctx_impl(disco_ctx* c, const float* d_user_gray, int un, int uh, int uw) {
    const int n = d_user_gray ? un : 2, H = d_user_gray ? uh : 256, W = d_user_gray ? uw : 256, K = c->opt.n_clusters;
    const int sp = c->opt.sp_size > 0 ? c->opt.sp_size : 16;          // (--psize: the token grid, hence the sizes of the token outputs below)
    const int L = (H / sp) * (W / sp);
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

}  // namespace disco_api

extern "C" {

int disco_calibrate(disco_ctx* c, const float* d_gray, int n, int h, int w) {
    if (!c || !c->finalized || !d_gray) { set_error("disco_calibrate: bad argument / context not finalized"); return DISCO_EINVAL; }
    const int sp = c->opt.sp_size > 16 ? c->opt.sp_size : 16, cell = c->opt.sp_size > 0 ? c->opt.sp_size : 16;
    if (n < 1 || n > 64 || h < sp || w < sp || h % sp || w % sp || (!c->opt.network && (h / cell) * (w / cell) < c->opt.n_clusters)) { set_error("disco_calibrate: bad size %dx%dx%d", n, h, w); return DISCO_ESHAPE; }
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

}  // extern "C"
