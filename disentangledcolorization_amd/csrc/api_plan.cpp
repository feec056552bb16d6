// api_plan.cpp - AnchorColorProb.forward (models/model.py:103-199) as a sequence of HIP launches: the workspace plan, the three network
// stages, the token path, and the forward entry points of the C ABI (split out of api.cpp, round 6).
#include "plan.h"

namespace disco_api {



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

int run_plan(disco_ctx* c, const disco_forward_args* a, size_t cap, bool dry, size_t* peak, bool calib) {
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
    const float key_thr = 25.f / (float)(sp * sp);         // "fewer than 25 pixels" as a share of the cell (model.py:122)
    if (!dry && P.ok()) P.rc = launch_encoder_stack(src, pos, pos_rep, c->d_enc[0], enc, n, L, enc_ws, s, P.dbg_row >= 0 ? &enc_dbg : nullptr, c->d_enc_pk[0], key_sizes, 1, key_thr);
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
    if (!dry && P.ok()) P.rc = launch_encoder_stack(hint, pos, pos_rep ? rep : 0, c->d_enc[1], dec, n2, L, enc_ws, s, nullptr, c->d_enc_pk[1], key_sizes, rep, key_thr);
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


int check_forward_args(disco_ctx* c, const disco_forward_args* a) {
    if (!c || !a) { set_error("null argument"); return DISCO_EINVAL; }
    if (!c->finalized) { set_error("disco_forward before disco_finalize"); return DISCO_ESTATE; }
    const int sp = c->opt.sp_size;
    const int mult = sp > 16 ? sp : 16;              // whole superpixel cells AND the conv stacks' four stride-2 stages
    if (a->n < 1 || a->h < mult || a->w < mult || a->h % mult || a->w % mult) { set_error("bad input size %dx%dx%d (multiples of %d)", a->n, a->h, a->w, mult); return DISCO_ESHAPE; }
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

}  // namespace disco_api

extern "C" {

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

int disco_sync(void* stream) {
    DISCO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return DISCO_OK;
}

}  // extern "C"
