// api_ops.cpp - the operator-level entry points of the C ABI: one kernel family per call on caller-owned buffers - what the -m gpu parity
// tests drive, op by op, against the oracle (split out of api.cpp, round 6).
#include "ctx.h"

namespace disco_api {

// every size an op entry point takes must be positive (a zero superpixel size would divide by zero on the host, an
// empty dimension would launch an empty grid)
bool positive(const char* op, std::initializer_list<long> dims) {
    for (long d : dims)
        if (d <= 0) { set_error("%s: non-positive size %ld", op, d); return false; }
    return true;
}

}  // namespace disco_api

extern "C" {

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
