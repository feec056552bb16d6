/* disco_hip.h — C ABI of the MI355X-native DISCO colorization hot path (libdisco_hip.so).
 *
 * This is the boundary a native replacement of the reference's per-image forward exports
 * (SURVEY §8b).  Entry points and the reference interface each one replaces (paths relative
 * to the reference root):
 *
 *   disco_create / disco_destroy      model.AnchorColorProb(...) ctor        models/model.py:33-76
 *                                     (main/colorizer/inference.py:71-74, .cuda() :81)
 *   disco_load_tensor / disco_finalize  load_checkpoint -> load_state_dict(strict)
 *                                     main/utils_train.py:140-151, inference.py:85
 *   disco_forward                     AnchorColorProb.forward(gray, ab, True, sampled_T)
 *                                     models/model.py:103-199, inference.py:108-109
 *   disco_workspace_bytes             (allocation the reference leaves to torch's caching allocator)
 *   disco_op_*                        the individual stock-PyTorch operators of SURVEY §2b
 *                                     (K1..K16), exposed so each kernel can be parity-tested
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  `d_` arguments are DEVICE pointers on the
 *     context's device, `h_` arguments are HOST pointers.
 *   - ownership: the caller owns every input, output and workspace buffer.  The context owns
 *     only the packed weights it allocated in disco_finalize.  Nothing is returned by
 *     pointer-to-new-memory.
 *   - errors: 0 on success, negative DISCO_E* otherwise; never exit()/throw across the ABI.
 *     disco_last_error() gives a thread-local message for the last failing call.
 *   - threading: launchers are asynchronous on the given hipStream_t (passed as void*); the GPU work of successive calls on
 *     different streams overlaps.  A context holds mutable host state (the pinned staging ring of the index arrays, the one-shot
 *     progress event, profiling records, calibration tables), so its entry points - disco_forward, disco_forward_segnet / _repnet / _enhance,
 *     disco_calibrate, disco_saturation_count and the disco_set_* calls - serialise on a mutex inside the context: several host
 *     threads may share a context (their host-side issue takes turns, their streams still overlap on the GPU), but
 *     disco_set_progress_event + the forward it arms are two calls: arm and launch from ONE thread.  The op-level entry points
 *     (disco_op_*) touch no context.  The only process-global state is one-time per-device setup (function attributes, the op-level
 *     gamut table: std::call_once / atomics).  One context per device.
 *     Small batches (n h w <= 8 x 256 x 256): disco_forward issues SpixelNet on a side stream owned by the context, forked from and joined
 *     back into the caller's stream with events - every result is complete on the caller's stream as always, inputs may be reused once
 *     that stream has passed the forward (DISCO_FORK_SEGNET=0 in the environment keeps everything on the caller's stream).
 *     No hidden host synchronisation except in disco_forward's k-means fallback bookkeeping
 *     (documented there) and disco_sync.
 *   - host-side randomness (k-means initial rows, empty-cluster fallback rows, random hints) is
 *     passed in as int32 index arrays so the native side is deterministic.
 */
#ifndef DISCO_HIP_H
#define DISCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DISCO_ABI_VERSION 11

#define DISCO_OK 0
#define DISCO_EINVAL (-1)       /* bad argument / null pointer */
#define DISCO_ESHAPE (-2)       /* unsupported or inconsistent shape */
#define DISCO_EUNSUPPORTED (-3) /* flag combination outside the hot path */
#define DISCO_EHIP (-4)         /* HIP runtime error (message has hipGetErrorString) */
#define DISCO_ESTATE (-5)       /* call order (e.g. forward before finalize, missing tensor) */
#define DISCO_ENOMEM (-6)

typedef struct disco_ctx disco_ctx;

/* activation codes of the fused conv epilogue */
#define DISCO_ACT_NONE 0
#define DISCO_ACT_RELU 1
#define DISCO_ACT_LRELU 2 /* LeakyReLU(slope) */
#define DISCO_ACT_TANH 3

/* conv precision modes */
#define DISCO_PREC_F16X3 0 /* fp16 hi/lo split operands, 3 MFMA products, fp32 accumulate */
/* (1 was DISCO_PREC_F16X1, "fp16 hi operands only", a measurement mode of rounds 1-2 that ran on round 1's conv kernel; removed
 *  with that kernel in ABI version 6: disco_create rejects it) */
#define DISCO_PREC_MX8 2   /* the default of rounds 2-3a, still available (DISCO_PREC_MX6 is the default now): the enhanceNet (HourGlass2,
                              everything downstream of the anchors) computes
                              w a ~= w_h a_h + fp8(w - w_h) fp8(a) + fp8(w) fp8(a - a_h): an fp16 main product plus two fp8
                              (e4m3) correction products in one K=64 MFMA (csrc/conv_mx.hip).  Its activations carry an fp16
                              plane and two fp8 planes with one power-of-two scale per tensor, fixed by a calibration forward
                              at the end of disco_finalize; clamping against that scale at run time is counted
                              (disco_saturation_count).  SpixelNet and ColorProbNet - the stacks that decide the anchors -
                              stay on DISCO_PREC_F16X3.  Measured: max |ab - reference| ~1e-4, anchors identical. */
#define DISCO_PREC_MX8_ALL 3 /* every conv stack on the fp8-corrected kernel: ~6% faster again, but the ~3e-5 perturbation it
                              leaves at the encoder output flips k-means anchors in ~1% of images (measurements only) */
#define DISCO_PREC_MX6 5     /* the default since round 3: as DISCO_PREC_MX8 with the two correction products of the HourGlass2 in MX fp6 (OCP e2m3,
                              one E8M0 block scale per pixel and 32 channels on the activation side: DISCO_PLANE_Q6 - and, since round 4, one per
                              output channel, 32-channel block and tap on the weight side, so heavy-tailed rows keep their corrections) instead of fp8: the K = 64
                              MFMA runs fp6 operands in half the passes.  The first HourGlass2 layer still reads fp8 planes (its producers
                              write those) and writes fp6 ones.  Measured: max |ab - reference| 1.2-1.3e-4 (MX8: 1.0-1.1e-4 on the same
                              inputs), anchors identical (they are decided upstream); 2-3 % faster end to end. */
#define DISCO_PREC_X2Q 4     /* as DISCO_PREC_MX6, and the ColorProbNet on the same kernel's second arithmetic: w_h a_h + w_l a_h
                              in fp16 and only the activation residual in fp8 (5 matrix-pipe units per 32 channels and tap
                              instead of 6).  ~1.4e-5 at the encoder output where F16X3 leaves ~5e-6; anchors differ from the fp32 reference
                              in 0.66 % of images (F16X3: 0.10 %): opt-in. */

int disco_abi_version(void);
const char *disco_last_error(void);

/* ---- context ------------------------------------------------------------------------- */

typedef struct disco_options {
    int32_t sp_size;     /* superpixel size: --psize (inference.py:147), the cell of poolfeat / get_spixel_size / upfeat (model.py:109-121,
                            191).  16 (the default, dedicated kernels), 8 or 32 (ABI 11: the general pooling kernels); H and W must be
                            multiples of max(16, sp_size) */
    int32_t n_clusters;  /* K anchors (inference.py:156) */
    int32_t random_hint; /* 1: anchors come from h_hint_pos instead of k-means (model.py:69) */
    int32_t precision;   /* DISCO_PREC_* for the conv stacks */
    int32_t network;     /* which network the context holds (ABI 10: the field was `segnet_only` through ABI 9; 0 = the colorizer):
                            1: only the SpixelSeg weights ("segnet.net.*", 94 tensors), serves disco_forward_segnet
                               (models/model.py:12-29, main/spixelseg/inference.py:89);
                            2 (ABI 9): only ColorProbNet's ("repnet.*", models/network.py:147-236), serves disco_forward_repnet;
                            3 (ABI 9): only HourGlass2's ("enhanceNet.*", models/network.py:125-144), serves disco_forward_enhance -
                               disco_finalize does not calibrate such a context (it has no input of its own to measure ranges on):
                               disco_calibrate with a first batch of the caller's (n,65,h,w) input comes before the first forward */
    int32_t hint2regress; /* 1: --hint2regress (inference.py:158): the hint embedding takes the anchors' ab values instead
                             of their one-hot bins, trg_word_emb is (64,67), trg_word_prj (2,64) and ref_logit has 2
                             channels (model.py:63-64,177-181,188) */
    int32_t spix_pos;     /* 1: --spix_pos (inference.py:156): the sine position encoding is evaluated per PIXEL and
                             pooled into the superpixels together with the features, so every image has its own
                             position sequence (model.py:106-112) */
    int32_t use_mask;     /* (ABI 11) 1: use_mask=True (model.py:38,121-125,133,186): both encoder stacks receive the reference's
                             FLOAT key_padding_mask - 1.0 at superpixels of fewer than 25 pixels (get_spixel_size < 25/256) - which
                             nn.MultiheadAttention ADDS to the scores under torch >= 1.9 (the pinned torch 1.8 rejects a float mask;
                             oracle/disco_ref.py encoder_layer).  Not combinable with sampled_T > 0 (the reference fails there) */
} disco_options;

int disco_create(int device, const disco_options *opt, disco_ctx **out);
int disco_destroy(disco_ctx *ctx);

/* Hand one checkpoint tensor (state_dict entry `key`) to the context.
 * h_data: host fp32 (int64 `num_batches_tracked` entries may be passed as NULL, they are unused). */
int disco_load_tensor(disco_ctx *ctx, const char *key, const float *h_data, const int64_t *shape, int ndim);

/* Strict check (every expected key present with the expected shape, no extras), then fold
 * spectral-norm / batch-norm, split to fp16 hi/lo, pack into MFMA fragment order and upload. */
int disco_finalize(disco_ctx *ctx);

/* Number of state_dict entries a default context expects and the i-th expected key/shape; the _ctx variant answers
 * for the options of a given context (hint2regress changes the shapes of trg_word_emb / trg_word_prj). */
int disco_expected_tensors(void);
int disco_expected_tensor(int i, const char **key, int64_t shape[4], int *ndim);
int disco_expected_tensor_ctx(disco_ctx *ctx, int i, const char **key, int64_t shape[4], int *ndim);

/* ---- forward ------------------------------------------------------------------------- */

typedef struct disco_forward_args {
    int32_t n, h, w;          /* input batch; h, w multiples of sp_size */
    int32_t sampled_T;        /* 0: top-1 anchors; >0: diverse (3 outputs per image); <0: GT anchor colours */
    int32_t test_mode;        /* 1: inference (model.py:138-168).  0: the validation forward of train_colorizer.py:206
                                 (model.eval(), test_mode=False): anchors from k-means on the pooled GT colours
                                 (model.py:169-171), hint labels = GT token labels, spix_colors = pooled GT colours,
                                 sampled_T ignored */
    const float *d_gray;      /* (n,1,h,w) fp32 NCHW, L in [-1,1] */
    const float *d_ab;        /* (n,2,h,w) fp32 NCHW, ab/110 */
    const int32_t *h_init_idx;      /* (n,K) k-means initial rows (np.random.choice per image, clusterkit.py:107) */
    const int32_t *h_fallback_rows; /* (n,max_fallback) rows for empty clusters (clusterkit.py:181-182); may be NULL */
    int32_t max_fallback;
    const int32_t *h_hint_pos;      /* (n,K) anchor tokens when random_hint (basic.py:42-47); else NULL */
    /* outputs, fp32 NCHW, n' = n (or 3n when sampled_T>0, image-major [n][t]) */
    float *d_pal_logit;   /* (n ,313,h/sp,w/sp) */
    float *d_ref_logit;   /* (n',313,h/sp,w/sp); (n',2,h/sp,w/sp) when hint2regress */
    float *d_pred_colors; /* (n',2,h,w) */
    float *d_affinity;    /* (n ,9,h,w)  (the reference returns the expanded n' copies; rows repeat) */
    float *d_spix_colors; /* (n',2,h/sp,w/sp) */
    float *d_hint_mask;   /* (n ,1,h/sp,w/sp) */
    int32_t *h_kmeans_events; /* (n) out: empty-cluster events consumed per image; may be NULL */
    void *d_workspace;
    size_t workspace_bytes;
    void *stream; /* hipStream_t */
} disco_forward_args;

int disco_workspace_bytes(disco_ctx *ctx, int n, int h, int w, int sampled_T, size_t *bytes);
/* DISCO_PREC_MX8: number of fp8 activation elements that had to be clamped (|x 2^scale| > 448) since the previous call; the
 * counter is read on `stream` (one synchronisation) and reset.  Non-zero means some activation exceeded ~14x the range the
 * calibration pass saw: the affected correction products lose accuracy (results degrade towards plain-fp16 operands). */
int disco_saturation_count(disco_ctx *ctx, void *stream, uint64_t *count);
/* Images (since the previous call) whose k-means left the several-workgroup kernel for the one-workgroup kernel because their
 * workgroups could not be resident together - the GPU was shared with other work in a way the launch heuristic did not foresee
 * (ABI 11; replaces models/clusterkit.py:49-58, a per-image loop that cannot hang: neither can this).  Results are identical either
 * way; a non-zero count only says that those images took a few hundred microseconds longer.  Synchronises `stream`. */
int disco_kmeans_fallback_count(disco_ctx *ctx, void *stream, uint64_t *count);
/* Widen the fp8 scales of an mx8 context with the activation ranges of the caller's own images: d_gray device fp32 (n,1,h,w),
 * n <= 64, h and w multiples of 16.  Blocking (synchronises the device first: no forward of this context may be in flight).
 * Calibrations accumulate (a tensor's recorded max |x| only grows), so results of later forwards change at the 1e-5 level only
 * when a scale actually moves.  disco_finalize has already calibrated on two synthetic images; call this when
 * disco_saturation_count reports clamping on your data.  On a stand-alone HourGlass2 context (network = 3) the input is that
 * network's own: d_gray = (n,65,h,w), and the first call is what makes the context usable. */
int disco_calibrate(disco_ctx *ctx, const float *d_gray, int n, int h, int w);
/* The calibration pass's per-tensor record (diagnostics): producer key, max |x| over the calibration images, chosen scale exponent
 * (every plane of the tensor stores x 2^sexp, the maximum landing in [16, 32)).  disco_finalize fails with DISCO_EUNSUPPORTED only
 * for a tensor that is not finite, or for two tensors that are concatenated on read whose ranges differ by more than 2^10 (they
 * must share one exponent); a failing disco_calibrate leaves the previous calibration in place. */
int disco_calibration_count(disco_ctx *ctx);
/* ABI 8.  The arithmetic the HourGlass2 of this context actually runs on (DISCO_PREC_MX6, DISCO_PREC_MX8 or DISCO_PREC_F16X3) and the channel
 * disparity the calibration passes measured on its MX-fp6 tensors: per 32-channel block the largest per-channel max |x| over the lower quartile of
 * the block's live channels, maximised over blocks and tensors.  MX fp6 planes share one scale per pixel and 32 channels, so a channel far below
 * its block's largest loses its correction operands - harmful when the consumer's weights make up for the difference, as trained BatchNorm affines
 * can.  disco_finalize / disco_calibrate therefore (1) above a disparity of 16 level the channels of every tensor inside the HourGlass2 with
 * power-of-two factors folded into the producers' output channels and the consumers' input channels (exact in fp32: the network function does
 * not change; *disparity_before_equalisation then holds the first measurement, else 0) and measure again, and (2) if the disparity still exceeds 64,
 * rebuild the HourGlass2 on fp8 corrections (e4m3: 4 exponent bits; same accuracy as DISCO_PREC_MX8, 2-3 % slower): *precision then reports
 * DISCO_PREC_MX8 although the context was created with DISCO_PREC_MX6.  The synthetic checkpoint reads 9 and is left alone. */
int disco_enhance_arithmetic(disco_ctx *ctx, int *precision, float *channel_disparity, float *disparity_before_equalisation);
/* i-th entry of the calibration record; *key points at a copy owned by the calling thread, valid until that thread's next call */
int disco_calibration_entry(disco_ctx *ctx, int i, const char **key, float *amax, int *sexp);
/* One network of the colorizer on its own (the reference's models/network.py classes as modules of their own), on a full context or on
 * the stand-alone context of that network (disco_options.network = 1 / 2 / 3); workspace as reported by
 * disco_subnet_workspace_bytes(ctx, which = 1 / 2 / 3, ...) (on a stand-alone context disco_workspace_bytes reports the same).
 *   disco_forward_segnet:  SpixelNet.forward(gray (n,1,h,w)) -> affinity (n,9,h,w), softmax over the 9 neighbour slots (network.py:293-313)
 *   disco_forward_repnet:  ColorProbNet.forward(gray (n,1,h,w)) -> features (n,64,h,w) (network.py:220-236)                       [ABI 9]
 *   disco_forward_enhance: HourGlass2.forward(x (n,65,h,w) = cat(gray, 64 features), model.py:196) -> (n,2,h,w) BEFORE the tanh of
 *                          model.py:197 (network.py:134-143).  Arithmetic and activation scales as in the colorizer; on a full context the
 *                          input ranges are the ones the colorizer's own features were calibrated on.                              [ABI 9] */
int disco_forward_segnet(disco_ctx *ctx, int n, int h, int w, const float *d_gray, float *d_affinity, void *d_workspace,
                         size_t workspace_bytes, void *stream);
int disco_forward_repnet(disco_ctx *ctx, int n, int h, int w, const float *d_gray, float *d_feats, void *d_workspace,
                         size_t workspace_bytes, void *stream);
int disco_forward_enhance(disco_ctx *ctx, int n, int h, int w, const float *d_input, float *d_out, void *d_workspace,
                          size_t workspace_bytes, void *stream);
int disco_subnet_workspace_bytes(disco_ctx *ctx, int which, int n, int h, int w, size_t *bytes);
int disco_forward(disco_ctx *ctx, const disco_forward_args *a);
int disco_sync(void *stream);

/* Per-stage timing hooks used by bench.py: after a forward with profiling enabled the context
 * holds hipEvent timings of named stages on the forward's stream. */
int disco_set_profiling(disco_ctx *ctx, int level); /* 0 off, 1 stages, 2 stages + every MFMA conv launch */
/* Pipelining hook (ABI 7): the NEXT disco_forward on this context records `hip_event` (a hipEvent_t the caller owns) on its stream right
 * behind its `after_conv_launches`-th MFMA conv launch (at its end if it has fewer); one shot, NULL cancels.  "Next forward" is whatever
 * forward entry point runs next on the context - disco_forward, disco_forward_segnet, disco_calibrate - and however it ends: on an
 * argument error the event is recorded on the call's stream (if it has one) and the handle dropped, so it can never fire later.  A caller that runs two
 * micro-batches on two streams lets the second wait for this event of the first: the two forwards then run half a network apart, and the
 * latency-bound token path / k-means of either (a handful of CUs busy) executes under the other's convolutions instead of both idling
 * the GPU at the same time (runner.py; the reference has no counterpart: it runs one batch on one stream). */
int disco_set_progress_event(disco_ctx *ctx, void *hip_event, int after_conv_launches);
/* Debugging aid: d_table = device uint64 [rows][cols] (NULL: off).  Forward number k (counted from this call) zeroes row k % rows and
 * adds into column j a position-weighted word sum of the output of its j-th stage (every conv launch in order, the tokens, the encoder
 * output, pal_logit) - tools/stagger_probe.py compares rows of forwards that must agree. */
int disco_set_debug_checksums(disco_ctx *ctx, void *d_table, int rows, int cols);
/* ... and per row (same row index) a copy of the first token GEMM's input (n,L,64) and output q|k|v (3,n L,64), bytes_per_row apart */
int disco_set_debug_dump(disco_ctx *ctx, void *d_buf, size_t bytes_per_row);
int disco_profile_count(disco_ctx *ctx);
int disco_profile_entry(disco_ctx *ctx, int i, const char **name, float *ms, double *flops);
/* level 2: number of conv3x3_mx_kernel launches of the last forward, their summed duration (hipEvent pairs around
 * each launch on the forward's stream) and their summed algorithmic FLOPs (2*9*Cin*Cout*Hout*Wout*N). */
int disco_profile_conv(disco_ctx *ctx, int *launches, float *total_ms, double *total_flops);
/* level 2: summed compulsory HBM bytes of those launches (sources and residual read once, output written once,
 * 4 B per element = fp16 hi + lo planes, packed weights once) - the denominator for roofline.traffic */
int disco_profile_conv_bytes(disco_ctx *ctx, double *total_bytes);
/* level 2: the i-th MFMA conv launch of the last forward: checkpoint key of the layer, duration, algorithmic FLOPs */
int disco_profile_conv_entry(disco_ctx *ctx, int i, const char **key, float *ms, double *flops);

/* ---- operator-level entry points (parity tests, micro-benchmarks) ------------------- */

/* Activation tensors are channel-blocked fp16, [N][C/16][H][W][16], in two planes: hi at d_x, lo at
 * d_x + plane_elems (x ~= hi + lo, |lo| <= ulp(hi)/2; C padded to a multiple of 16).  Convert from / to the
 * reference's fp32 NCHW: */
int disco_op_nchw_to_act(const float *d_src, void *d_dst, int n, int c, int h, int w, int c_pad, void *stream);
int disco_op_act_to_nchw(const void *d_src, float *d_dst, int n, int c, int h, int w, int c_pad, void *stream);

typedef struct disco_conv_desc {
    int32_t n, h_in, w_in;     /* logical input size (after the optional x2 upsample-on-read) */
    int32_t c_in0, c_in1;      /* channels of source 0 / source 1 (concat-on-read; c_in1 = 0 if unused) */
    int32_t up0, up1;          /* 1: source is stored at half resolution, nearest-upsampled on read */
    int32_t c_out, stride;     /* 3x3, pad 1, stride 1 or 2 */
    int32_t act;               /* DISCO_ACT_* applied after bias (+residual) */
    float slope;
    int32_t precision;         /* DISCO_PREC_F16X3 (this entry point runs the f16x3 arithmetic; the others: disco_op_conv3x3_mx) */
    int32_t sexp_in, sexp_out, sexp_res;   /* scale exponents of the sources (both the same), the output and the residual: every plane
                                  of an activation buffer stores x 2^sexp (0 = the plain values disco_op_nchw_to_act writes) */
} disco_conv_desc;

/* Pack an effective fp32 OIHW 3x3 weight (host) for the MFMA kernel; returns bytes needed when
 * d_packed == NULL.  c_in = c_in0 + c_in1 (each a multiple of 16, or c_in0 arbitrary when c_in1 = 0). */
int disco_op_conv3x3_pack(const float *h_w_oihw, int c_out, int c_in, void *d_packed, size_t *bytes);
/* out = bn(act(conv3x3(cat(src0,src1)) + bias [+ res]));  bias/bn_scale/bn_shift: device fp32 (c_out) or NULL */
int disco_op_conv3x3(const disco_conv_desc *d, const void *d_src0, const void *d_src1, const void *d_packed_w,
                     const float *d_bias, const float *d_bn_scale, const float *d_bn_shift, const void *d_res,
                     void *d_out, void *stream);

/* ---- conv3x3 with an fp16 main product and fp8 (e4m3, K = 64 MFMA) correction products (csrc/conv_mx.hip) -----------
 * Activation buffers of this path carry ONE power-of-two scale per tensor, xs = x 2^sexp, in every plane: the fp16 hi plane
 * [N][C/16][H][W][16] = fp16(xs), then (planes bit 0) the fp16 lo plane fp16(xs - hi), then (planes bit 1) the fp8 q planes
 * [N][C/32][2][H][W][32]: a8 = fp8(xs) and al8 = fp8((xs - hi) 2^11).  C is padded to a multiple of 16 (32 with q planes).
 * (sexp puts the tensor's largest |xs| into [16, 32): fp16 keeps 2^11 of headroom, fp8's +-448 a factor 14.) */
#define DISCO_PLANE_LO 1
#define DISCO_PLANE_Q 2
#define DISCO_PLANE_QL 4   /* instead of DISCO_PLANE_Q: only the al8 planes, [N][C/32][H][W][32] (operands of the x2q arithmetic) */
#define DISCO_PLANE_Q6 8   /* instead of DISCO_PLANE_Q: MX fp6 (OCP e2m3) planes in the same geometry, block-scaled per pixel and 32 channels:
                              a pixel's 32-byte slot holds the 32 six-bit fields of its 32-channel block in bytes 0-23 (little-endian bit
                              stream; field 4g+i = channel 8g+i, field 16+4g+i = channel 8g+4+i) and its E8M0 scale byte in byte 24;
                              with E = the fp16 exponent of the block's largest |hi|: a6 = fp6(hi / 2^(E-2)), scale byte 127 + E - 2;
                              al6 = fp6((xs - hi) / 2^(E-14)), scale byte 127 + E - 3 - operands of the f16 + fp6x2 arithmetic */
int disco_op_act_bytes(int n, int c_pad, int h, int w, int planes, size_t *bytes);
/* (n,1,h,w) fp32 -> the 16-channel fp16 hi-plane tensor (x_hi, x_lo, x_hi, 0 ...) x 2^sexp: the second source (c_in1 = 16) of a two-source
 * f16 + fp8x2 layer whose weights were packed with variant 3 - the channel then enters as an exact three-product fp16 split in one
 * K = 16 MFMA per tap (models/model.py:194-196: the gray channel of the HourGlass2's input) */
int disco_op_gray_tail(const float *d_gray, void *d_out, int n, int h, int w, int sexp, void *stream);
int disco_op_nchw_to_act_mx(const float *d_src, void *d_dst, int n, int c, int h, int w, int c_pad, int planes, int sexp,
                            void *stream);
/* which = 0: hi (+ lo when present); 1: the a8 plane dequantised; 2: hi + the al8 plane dequantised */
int disco_op_act_mx_to_nchw(const void *d_src, float *d_dst, int n, int c, int h, int w, int c_pad, int planes, int sexp,
                            int which, void *stream);
typedef struct disco_conv_mx_desc {
    int32_t n, h_in, w_in;
    int32_t c_in0, c_in1;      /* multiples of 32; both sources carry hi + q planes (planes = DISCO_PLANE_Q) */
    int32_t up0, up1;
    int32_t sexp0, sexp1;      /* scale exponents of the sources' q planes */
    int32_t c_out, stride;
    int32_t act;
    float slope;
    int32_t out_planes;        /* DISCO_PLANE_* bits of the output buffer; ignored with out_f32 */
    int32_t out_sexp;
    int32_t out_f32;           /* 1: d_out is fp32 NCHW */
    int32_t res_planes;        /* DISCO_PLANE_* bits of the residual buffer (its lo plane is used when present) */
    int32_t res_sexp;          /* scale exponent of the residual buffer */
    int32_t x2q;               /* 1: the f16x2 + fp8 arithmetic: one source with DISCO_PLANE_QL planes, c_in0 a multiple of 64,
                                  weights packed with x2q = 1 */
    int32_t q6;                /* 1: the f16 + fp6x2 arithmetic (the default of the HourGlass2): sources with DISCO_PLANE_Q6 planes, weights packed
                                  with variant 2; q planes of the output: DISCO_PLANE_Q6 */
    int32_t d2s;               /* 1: depth-to-space epilogue (the sub-pixel up-convs / transposed convs of the forward): c_out = 4 C
                                  phase-major output channels, channel ph*C + c of input pixel (y, x) goes to channel c of pixel
                                  (2y + ph/2, 2x + ph%2) of an (n, C, 2h, 2w) activation buffer; C a multiple of 32; stride 1 */
} disco_conv_mx_desc;
/* d_packed == NULL: only *bytes.  d_wexp: device int32 [round_up(c_out, 32)], the per-output-channel weight scale exponents.
 * x2q: the pack variant = arithmetic: 0 f16 + fp8x2, 1 f16x2 + fp8 (x2q), 2 f16 + fp6x2, 3 = 0 with the last of 32 k + 1 input
 * channels in the 16-channel fp16 tail (disco_op_gray_tail) */
int disco_op_conv3x3_mx_pack(const float *h_w_oihw, int c_out, int c_in, int x2q, void *d_packed, int32_t *d_wexp,
                             size_t *bytes);
int disco_op_conv3x3_mx(const disco_conv_mx_desc *d, const void *d_src0, const void *d_src1, const void *d_packed_w,
                        const int32_t *d_wexp, const float *d_bias, const float *d_bn_scale, const float *d_bn_shift,
                        const void *d_res, void *d_out, uint32_t *d_sat, const uint32_t *d_tapmask, void *stream);
/* d_tapmask (optional, what the forward uses for its sub-pixel up-convs / transposed convs): per 32-output-channel block a 9-bit mask
 * of the taps that carry any non-zero weight; masked-out taps are neither fetched nor multiplied.  Device uint32 [ceil(c_out / 32)]: */
int disco_op_conv3x3_tapmask(const float *h_w_oihw, int c_out, int c_in, uint32_t *d_mask);

/* Diagnostic: sustained v_mfma_f32_32x32x16_f16 rate of the current device on a registers-only loop (no LDS, no
 * memory), 2 waves per SIMD on every CU, operands: mode 0 = zeros, 1 = N(0,1) fp16, 2 = the f16x3 product mix
 * (hi*lo, lo*hi, hi*hi) the conv kernel issues, 3 = that mix with post-ReLU-like pixel operands (non-negative, half
 * of the elements exact zeros).  The MI355X clock is power-managed and MFMA power depends on operand
 * toggling, so this - not the 2.5 PFLOP/s datasheet peak - is what a kernel with this data can reach.  Blocking;
 * *tflops = executed TFLOP/s of the second (warm) run of `iters` iterations x 12 MFMAs per wave. */
int disco_diag_mfma_rate(int mode, int iters, double *tflops);

/* ConvTranspose2d 4x4 s2 p1 + bias + LeakyReLU(slope): h_w_iohw is the (c_in,c_out,4,4) fp32 weight */
int disco_op_deconv4x4_pack(const float *h_w_iohw, int c_in, int c_out, void *d_packed, size_t *bytes);
int disco_op_deconv4x4(const void *d_src, const void *d_packed_w, const float *d_bias, void *d_out, int n,
                       int h_in, int w_in, int c_in, int c_out, float slope, int precision, void *stream);

/* superpixel ops on fp32 NCHW tensors (basic.py:274-376) */
int disco_op_poolfeat(const float *d_feat, const float *d_prob, float *d_pooled, float *d_conf, float *d_sizes,
                      int n, int c, int h, int w, int sp, void *d_ws, size_t ws_bytes, void *stream);
/* the forward's own pooling launch: d_act = 64 channels in the activation layout with hi + lo planes (disco_op_nchw_to_act), d_nchw2 =
 * (n,2,h,w) fp32; -> tokens (n, h/16 w/16, 64) and the two pooled NCHW channels (n,2,h/16,w/16).  Workspace: as disco_op_poolfeat with 66 channels. */
int disco_op_poolfeat_act(const void *d_act, const float *d_nchw2, const float *d_prob, float *d_tokens, float *d_pooled2, int n, int h, int w,
                          void *d_ws, size_t ws_bytes, void *stream);
int disco_op_upfeat(const float *d_tok, const float *d_prob, float *d_out, int n, int c, int h, int w, int sp,
                    void *stream);

/* 6-layer encoder stack on (n,L,64) fp32 tokens; weights: the 12 tensors per layer concatenated
 * in state_dict order (h_weights host fp32), pos (L,64) device */
size_t disco_op_encoder_weight_floats(void);
int disco_op_encoder_stack(const float *d_x, const float *d_pos, const float *d_weights, float *d_out, int n,
                           int l, void *d_ws, size_t ws_bytes, void *stream);
/* ... with use_mask's key bias (ABI 11): d_key_sizes (n,l) = get_spixel_size of every token; keys below 25/256 get +1.0 on
 * every attention score of every layer (transformer2d.py:53-54 with the float key_padding_mask of model.py:121-125) */
int disco_op_encoder_stack_masked(const float *d_x, const float *d_pos, const float *d_weights, const float *d_key_sizes,
                                  float *d_out, int n, int l, void *d_ws, size_t ws_bytes, void *stream);

/* k-means (clusterkit.py:112-208) + anchors (anchor_gen.py:96-101) on n point sets of l points with d <= 64
 * features: d_x is (n,l,d) row-major, or (n,d,l) when channel_major (NCHW maps such as the pooled colours that the
 * validation forward clusters, model.py:169-171) */
int disco_op_kmeans_anchors(const float *d_x, const float *d_sizes, const int32_t *d_init_idx,
                            const int32_t *d_fallback_rows, int max_fallback, int32_t *d_assign,
                            int32_t *d_anchor, float *d_hint_mask, int32_t *d_info, int n, int l, int k, int d,
                            int channel_major, void *stream);
/* ... with caller-owned scratch (round 5): point sets of more than 512 points of 64 row-major features then run on SEVERAL workgroups
 * per image (ceil(l / 512), all resident: taken while n * ceil(l / 512) fits a quarter of the CUs) - the member sums travel down the
 * workgroups as a pipeline in ascending point order, so assignments, pass counts and events are those of the one-workgroup form.
 * Residency is PROVEN per image before anything is exchanged (an admission count in the scratch); an image whose workgroups do not
 * all arrive within a few milliseconds - or that runs into the 1e9-cycle backstop later - is given up there and computed by the
 * one-workgroup kernel launched right behind, bit-identical by construction (ABI 11: no trap, no error, no host involvement).
 * disco_op_kmeans_fallbacks counts the images of the LAST call on `d_ws` that went that way (synchronises `stream`).
 * Sizes: disco_op_kmeans_workspace_bytes (0 when the shape does not use scratch). */
size_t disco_op_kmeans_workspace_bytes(int n, int l);
int disco_op_kmeans_fallbacks(const void *d_ws, int n, int l, void *stream, int *count);
int disco_op_kmeans_anchors_ws(const float *d_x, const float *d_sizes, const int32_t *d_init_idx,
                               const int32_t *d_fallback_rows, int max_fallback, int32_t *d_assign,
                               int32_t *d_anchor, float *d_hint_mask, int32_t *d_info, int n, int l, int k, int d,
                               int channel_major, void *d_ws, size_t ws_bytes, void *stream);

/* softmax(313) -> stable top-10 -> colour pick (anchor_gen.py:54-90) and nearest-bin label (basic.py:177-194) */
int disco_op_select_colors(const float *d_logit_nchw, float *d_colors, int32_t *d_labels, int n, int hw, int t,
                           void *stream);
int disco_op_nearest_bin(const float *d_ab_nchw, int32_t *d_labels, int n, int hw, void *stream);
int disco_op_position_encoding(float *d_pos, int h, int w, void *stream);
/* ColorLabel.decode_ind2ab(logit, T) for integer T in [0,9]: ab/110 of the T-th most probable bin (basic.py:196-209) */
int disco_op_decode_ind2ab(const float *d_logit_nchw, float *d_ab_nchw, int n, int hw, int T, void *stream);
/* the same for non-integer T (the function's default T = 0.38): annealed mean, ab = sum_q exp(softmax(logit)_q / T) ab_q /
 * sum_q exp(softmax(logit)_q / T) / 110 (basic.py:210-217) */
int disco_op_decode_annealed(const float *d_logit_nchw, float *d_ab_nchw, int n, int hw, float T, void *stream);
/* basic.rgb2lab / basic.lab2rgb (models/basic.py:395-475): rgb in [0,1] <-> ((L-50)/50, a/110, b/110), (n,3,h,w) fp32 */
int disco_op_rgb2lab(const float *d_rgb, float *d_lab, int n, int h, int w, void *stream);
int disco_op_lab2rgb(const float *d_lab, float *d_rgb, int n, int h, int w, void *stream);
/* fetch_data after the image decode (main/colorizer/inference.py:23-42): uint8 RGB (n,h,w,3) -> bottom/right edge
 * padding to (hp,wp) -> /255 -> RGB2Lab -> gray (n,1,hp,wp) = (L-50)/50, ab (n,2,hp,wp) = ab/110 and, if d_rgb is
 * not NULL, rgb*2-1 (n,3,hp,wp).  The colour transform is the reference's torch rgb2lab (basic.py:395-437). */
int disco_op_rgb8_to_lab(const uint8_t *d_rgb8, float *d_gray, float *d_ab, float *d_rgb, int n, int h, int w, int hp,
                         int wp, void *stream);
/* save_normLabs_from_batch before the image encode (utils/util.py:91-106) with batch_depadding folded in:
 * normalised Lab (n,3,hp,wp) -> RGB -> uint8 (n,h,w,3) of the top-left h x w crop, truncating, saturating at 255 */
/* The default input path of main/colorizer/inference.py:32-40: cv2.resize(rgb8, (wo,ho), INTER_LINEAR) (opencv-python 4.6's
 * published 8-bit algorithm: 11-bit fixed-point coefficients, the 2x2 area path for exact 2x downscales) fused with /255 ->
 * RGB->Lab -> gray / ab / rgb*2-1 split.  d_rgb8: uint8 (n,h,w,3); d_resized: optional uint8 (n,ho,wo,3) copy of the resized
 * image; d_rgb may be NULL. */
int disco_op_rgb8_resize_to_lab(const uint8_t *d_rgb8, uint8_t *d_resized, float *d_gray, float *d_ab, float *d_rgb, int n, int h,
                                int w, int ho, int wo, void *stream);
int disco_op_lab_to_rgb8(const float *d_lab, uint8_t *d_rgb8, int n, int hp, int wp, int h, int w, void *stream);
/* basic.mark_color_hints(input_grays, target_ABs, gate_maps, kernel_size, base_ABs) (models/basic.py:95-117):
 * gray (n,1,h,w), target/base ab (n,2,h,w), gate (n,1,h,w) -> marked Lab (n,3,h,w); d_base_ab may be NULL */
int disco_op_mark_color_hints(const float *d_gray, const float *d_target_ab, const float *d_gate, const float *d_base_ab,
                              float *d_out, int n, int h, int w, int kernel_size, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DISCO_HIP_H */
