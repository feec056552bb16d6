// common.h — internal declarations shared by the HIP translation units of libdisco_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <atomic>
#include <functional>
#include <stddef.h>
#include "../../include/disco_hip.h"

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace disco {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define DISCO_HIP_CHECK(expr)                                  \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) return ::disco::hip_fail(_e, #expr); \
    } while (0)

#define DISCO_LAUNCH_CHECK(what)                                  \
    do {                                                          \
        hipError_t _e = hipGetLastError();                        \
        if (_e != hipSuccess) return ::disco::hip_fail(_e, what); \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---- activation tensor ----------------------------------------------------------------------------
// ONE power-of-two exponent per tensor, `sexp`, fixed at calibration time: every plane stores xs = x * 2^sexp (the largest |xs| of
// the calibration images lands in [16, 32): 2^11 of fp16 headroom above, and values down to 2^-7 of the maximum keep a NORMAL fp16
// lo word).  Rounds 1-2 stored x itself in the fp16 planes: a checkpoint whose activations are not O(1) overflowed fp16 (a hard
// failure above 16 384) or pushed the lo plane into fp16 subnormals.  A consumer undoes the scale with exact power-of-two
// multiplications folded into its epilogue parameters (conv: launch_conv3x3_mx) or applied to the fp32 sum (pooling).
// hi plane: channel-blocked [N][C/16][H][W][16] fp16 = fp16(xs), always present.
// lo plane (optional, `plane` != 0): same layout, fp16(xs - hi), at p + plane   (the f16x3 kernel's second operand plane,
//   residual inputs, the pooling kernel).
// q planes (optional, `q_off` != 0): fp8 e4m3 [N][C/32][2][H][W][32] at (char*)p + q_off: for every 32-channel block the
//   plane a8 = fp8(xs) followed by the plane al8 = fp8((xs - hi) * 2^11)   (the correction operands of conv3x3_mx_kernel).
//   q_kind 1: ONLY the al8 planes, [N][C/32][H][W][32] (1 byte per element): the operand of the f16x2+fp8 arithmetic, which
//   keeps both fp16 products of the hi plane and sends just the activation residual through fp8.
//   q_kind 2: MX fp6 (OCP e2m3) planes in the geometry of q_kind 0: a6 and al6, block-scaled per pixel and 32 channels (mx6_block_scale()
//   below).  A pixel's 32-byte slot of a plane holds the 32 six-bit fields of its 32-channel block as a little-endian bit stream in
//   bytes 0-23, its E8M0 scale byte in byte 24 (bytes 25-27 zero, 28-31 never written): field 4g + i = channel 8g + i, field
//   16 + 4g + i = channel 8g + 4 + i (g, i = 0..3) - the order in which the two lanes that own a pixel in the conv epilogue hold
//   their channels, so each writes 12 contiguous bytes (mx6_field_channel()).
//   The K = 64 MFMA runs fp6 operands in half the passes of fp8 (tools/fp6_probe.hip, profiles/r03_mfma_mix.txt).
struct Act {
    f16* p = nullptr;  // hi plane; lo plane at p + plane
    int n = 0, h = 0, w = 0, c = 0;
    size_t plane = 0;  // elements between the hi and the lo plane (= n*h*w*c), 0: no lo plane
    size_t q_off = 0;  // bytes between p and the q planes, 0: none
    int sexp = 0;
    int q_kind = 0;    // 0: a8 | al8 per 32-channel block, 1: al8 only, 2: a6 | al6 (fp6 slots)
    size_t elems() const { return (size_t)n * h * w * c; }
    size_t q_bytes() const { return q_off ? elems() * (q_kind == 1 ? 1 : 2) : 0; }
    size_t bytes() const { return elems() * sizeof(f16) * (1 + (plane ? 1 : 0)) + q_bytes(); }
};
constexpr int MX_LO_SHIFT = 11;     // al8 carries 2^11 more scale than a8 (|x - fp16(x)| <= 2^-11 |x|)
// fp6 slots carry an E8M0 block scale per pixel and 32 channels (the MX format proper; dword 6 of the slot, which the MFMA takes
// as its per-lane scale operand straight from the fragment registers): with E = the fp16 exponent of the slot's largest |hi word|,
// a6 = hi / 2^(E - 2) in (-8, 8) (saturating at 7.5: within the top binade's rounding step) under the byte 127 + E - 2, and
// al6 = lo / 2^(E - 14) (|lo| <= 2^(E - 11)) under the byte one below (the weights' side already carries the 2^-11 of MX_LO_SHIFT).
__host__ __device__ inline int mx6_block_scale(f16 amax_hi) { return (int)((__builtin_bit_cast(unsigned short, amax_hi) >> 10) & 31u) + 110; }
// channel (0..31 within its block) of six-bit field j of an fp6 slot, and the inverse
__host__ __device__ inline int mx6_field_channel(int j) { return 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3); }
__host__ __device__ inline int mx6_channel_field(int c) { return 16 * ((c >> 2) & 1) + 4 * (c >> 3) + (c & 3); }

// v + the values of lanes ^8, ^16, ^32 (the butterfly  v += shfl_xor(v, 8); v += shfl_xor(v, 16); v += shfl_xor(v, 32)  with the same
// operand order in the lanes of the first row, hence bit-identical there) without the LDS crossbar: DPP row rotate and the gfx950
// permlane swaps are plain VALU instructions, where __shfl_xor compiles to ds_bpermute_b32 (pool_partial_kernel had 216 per thread).
// (Written while hunting the fault described in pool.hip - the shuffles were not its cause - and kept: 3 VALU instead of 3 LDS ops.)
__device__ __forceinline__ float butterfly_add_8_16_32(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    const unsigned b16 = __builtin_bit_cast(unsigned, v);
    const auto s16 = __builtin_amdgcn_permlane16_swap(b16, b16, false, false);      // {rows 0,0,2,2 | rows 1,1,3,3}
    v = __builtin_bit_cast(float, (unsigned)s16[0]) + __builtin_bit_cast(float, (unsigned)s16[1]);
    const unsigned b32 = __builtin_bit_cast(unsigned, v);
    const auto s32 = __builtin_amdgcn_permlane32_swap(b32, b32, false, false);      // {lower half twice | upper half twice}
    v = __builtin_bit_cast(float, (unsigned)s32[0]) + __builtin_bit_cast(float, (unsigned)s32[1]);
#endif
    return v;
}

// four floats -> four fp8 e4m3 bytes (round to nearest even), clamped to the finite range; *sat counts clamped values
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d, unsigned* sat = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)      // the conversion builtins exist in the device pass only
    const float x0 = __builtin_amdgcn_fmed3f(a, -448.f, 448.f), x1 = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    const float x2 = __builtin_amdgcn_fmed3f(c, -448.f, 448.f), x3 = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    if (sat) *sat += (x0 != a) + (x1 != b) + (x2 != c) + (x3 != d);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, w, true);
    return (unsigned)w;
#else
    return 0u;
#endif
}
// Products and sums that must round like the reference's separate torch ops (anchor scores, k-means distances, upfeat slot sums,
// colour distances).  NOT __fmul_rn/__fadd_rn: the HIP headers define those as plain x * y / x + y, which hipcc's default
// -ffp-contract=fast fuses into v_fma / v_fmac (seen in the ISA of round 1's upfeat kernel); the pragma survives inlining.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
// ... and their packed forms (v_pk_mul_f32 / v_pk_add_f32 on register pairs).  Operands must be REAL pairs: build a broadcast with
// pair_of() - the opaque asm keeps the compiler from folding it into an op_sel source modifier, the form that is unsafe next to MFMA
// waves on this part (tools/pk_fault_repro.hip; tools/audit_op_sel.py scans the compiled kernels for it).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t pair_of(float v) {
    f32x2_t p = {v, v};
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(p));
#endif
    return p;
}
__device__ __forceinline__ f32x2_t mul_rn2(f32x2_t a, f32x2_t b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ f32x2_t add_rn2(f32x2_t a, f32x2_t b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

// 8 consecutive channels (the half `half` of a 16-channel block `blk`) of pixel `pix` of image `img` -> the planes of an act;
// v: TRUE values, stored scaled by 2^sexp
__device__ __forceinline__ void store_act8(f16* hi_p, long plane, long q_off, int sexp, long img, int blk, int half, long pix, long hw,
                                            int nblk, const float* v_true, unsigned* sat = nullptr, int q_kind = 0) {
    f16x8 h, l;
    float v[8], lo[8];
    const float sc = ldexpf(1.f, sexp);
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = v_true[j] * sc; h[j] = (f16)v[j]; lo[j] = v[j] - (float)h[j]; l[j] = (f16)lo[j]; }
    f16* o = hi_p + (((long)img * nblk + blk) * hw + pix) * 16 + half * 8;
    *reinterpret_cast<f16x8*>(o) = h;
    if (plane) *reinterpret_cast<f16x8*>(o + plane) = l;
    if (q_off) {
        const float qs = 1.f, qls = ldexpf(1.f, MX_LO_SHIFT);
        unsigned char* q = reinterpret_cast<unsigned char*>(hi_p) + q_off + (((long)img * (nblk >> 1) + (blk >> 1)) * (q_kind ? 1 : 2)) * hw * 32 + pix * 32 + (blk & 1) * 16 + half * 8;
        uint2 a, b;
        b.x = pack_fp8x4(lo[0] * qls, lo[1] * qls, lo[2] * qls, lo[3] * qls, sat); b.y = pack_fp8x4(lo[4] * qls, lo[5] * qls, lo[6] * qls, lo[7] * qls, sat);
        if (q_kind) *reinterpret_cast<uint2*>(q) = b;
        else {
            a.x = pack_fp8x4(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs, sat); a.y = pack_fp8x4(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs, sat);
            *reinterpret_cast<uint2*>(q) = a;
            *reinterpret_cast<uint2*>(q + hw * 32) = b;
        }
    }
}

// ---- conv3x3 (MFMA implicit GEMM) ---------------------------------------------------------------
constexpr int CONV_CK = 16;  // input-channel chunk (one MFMA k-block)

struct ConvSrc {
    const f16* p;    // hi plane
    long plane;      // elements between hi and lo plane
    int c;           // channels of this source (multiple of 16)
    int h, w;        // stored size (half of logical when up)
    int up;          // nearest x2 upsample on read
    int sexp;        // scale exponent of the tensor (both sources of a layer carry the same one)
};

struct ConvArgs {
    ConvSrc src[2];
    int nsrc;
    int n, h_in, w_in;  // logical input size
    int c_in;           // padded total input channels (multiple of 16)
    int h_out, w_out, stride;
    const f16* w;       // packed weights
    const uint32_t* tapmask;  // per 32-cout block: bit t set = tap t has a non-zero weight (null: all 9 taps)
    int c_out;          // real output channels
    int c_out_pad;      // channel stride of the output tensor
    const float* bias;
    const float* bn_scale;
    const float* bn_shift;
    const f16* res;     // residual (same layout as out) or null
    long res_plane;
    int res_sexp;       // scale exponents of the residual and of the output tensor (0 for fp32 NCHW outputs)
    int out_sexp;
    f16* out;
    long out_plane;
    float* out_f32;     // if set: fp32 NCHW output (n, c_out, h_out, w_out) instead of act planes
    int d2s_c;          // > 0: depth-to-space epilogue (ConvTranspose 4x4 s2 as a 4-phase conv): out channel
                        // co' = phase*d2s_c + co goes to pixel (2y + phase/2, 2x + phase%2) of a (2h,2w,d2s_c) act
    int act;
    float slope;
    int precision;
    uint32_t src_bytes[2];  // filled by the launcher: bytes addressable through each source's buffer descriptor
    uint32_t w_bytes;       // ... and through the packed-weight descriptor
    int softmax;        // fp32 NCHW output only: softmax over the c_out (<= 32) channels after the epilogue
    // fused Cin = 1 producer (ConvMxArgs::c1_gray): src[0] then only DESCRIBES the virtual input tensor (c, h, w, sexp; p may be null)
    const float* c1_gray; const float* c1_w; const float* c1_bias; int c1_act; float c1_slope;
};

// ---- conv3x3, fp16 main product + two fp8 (e4m3, K = 64) correction products (conv_mx.hip) ----------------------------
struct MxSrc {
    const f16* p;        // hi plane
    uint32_t q_off;      // bytes from p to the q planes
    int c, h, w, up;     // channels (multiple of 32), stored size, nearest x2 upsample on read
    int sexp;            // scale exponent of the tensor
};
struct ConvMxArgs {
    MxSrc src[2];
    int nsrc;
    int n, h_in, w_in;
    int c_in;                 // total input channels (multiple of 32)
    int h_out, w_out, stride;
    const void* w;            // packed weights (conv_mx_pack_host)
    const int32_t* wexp;      // per output channel (padded to 32): scale exponent of its fp8 weight planes
    const uint32_t* tapmask;
    int c_out, c_out_pad;
    const float* bias;
    const float* bn_scale;
    const float* bn_shift;
    const f16* res;           // residual: hi plane, lo plane at res + res_plane (res_plane 0: hi only)
    long res_plane;
    int res_sexp;             // scale exponent of the residual tensor
    // filled by the launcher from the exponents (exact powers of two): the epilogue computes
    //     x = acc * acc_mul + bias * bias_mul [+ res * res_mul];  x = act(x);  x = x * (bn_scale * bns_mul) + bn_shift * bnh_mul
    // in the domain 2^e_pre (e_pre = the sources' exponent with a BN, the output's without: activations are positively homogeneous;
    // tanh / fp32 outputs: 0) and leaves x * 2^out_sexp; force_bn: run the affine with the defaults (1, 0) although the layer has no BN
    float acc_mul, bias_mul, res_mul, bns_mul, bnh_mul;
    int force_bn;
    f16* out;                 // hi plane
    long out_plane;           // != 0: also write the lo plane at out + out_plane
    size_t out_q_off;         // != 0: also write the q planes at (char*)out + out_q_off
    int out_sexp;             // scale exponent of the output tensor (all planes; 0 for fp32 NCHW outputs)
    float* out_f32;
    int d2s_c;
    int act;
    float slope;
    int softmax;
    unsigned int* sat;        // optional device counter: q-plane elements that had to be clamped to +-448
    uint32_t src_bytes[2], w_bytes, out_bytes, res_bytes;   // filled by the launcher: buffer-descriptor ranges
    // arithmetic: 0 = f16 + fp8x2 (sources carry a8|al8 planes), 1 = f16x2 + fp8 ("x2q": w_h a_h + w_l a_h in fp16, fp8(w) fp8(a_l);
    // one source with al8-only planes, c_in a multiple of 64, weights packed with x2q = 1)
    int x2q;
    int q6;                   // 1: the f16 + fp6x2 arithmetic (AR 3): as 0 with fp6 e2m3 correction operands (sources with q_kind 2 planes,
                              // weights packed with variant 2): half the passes of the fp8 K = 64 MFMA
    int out_q_kind;           // layout of the output's q planes (Act::q_kind)
    // fused Cin = 1 producer (f16x3, stride 1, one source of <= 64 channels that is NOT read: src[0].p may be null): the layer's input is
    // computed in LDS as act(conv3x3(gray, c1_w) + c1_bias) 2^c1_sexp, conv_c1_kernel's arithmetic (conv_mx_kernel.h, GENC1)
    const float* c1_gray;     // (n,1,h,w) fp32, or null: an ordinary layer
    const float* c1_w;        // (c_in, 9)
    const float* c1_bias;     // (c_in) or null
    int c1_act; float c1_slope; int c1_sexp;
    int x3;                   // 1: the f16x3 arithmetic on this kernel (launch_conv3x3_x3): sources = hi + lo planes, MxSrc::q_off = byte
                              // distance between them, weights = conv3x3_pack_host's image, no q planes anywhere
};
// variant: 0 = f16 + fp8x2, 1 = f16x2 + fp8 (x2q), 2 = f16 + fp6x2
size_t conv_mx_packed_bytes(int c_out, int c_in_pad, int variant = 0);
// h_w: effective fp32 weight (c_out, c_in, 3, 3); ci_map as in conv3x3_pack_host; c_in_pad multiple of 32 (x2q: 64); h_wexp: cdiv(c_out,32)*32 ints
#define CONV_MX_LO_OF(ci) (-2 - (ci))      // ci_map code: the fp16 residual w - fp16(w) of real channel ci (conv_mx_pack_host)
void conv_mx_pack_host(const float* h_w, int c_out, int c_in, const int* ci_map, int c_in_pad, void* h_packed, int32_t* h_wexp, int variant = 0);
unsigned char fp6_e2m3_from_float(float x);      // round to nearest even, saturating to +-7.5
float fp6_e2m3_to_float(unsigned char code);
int launch_conv3x3_mx(const ConvMxArgs& a, hipStream_t s);
// the f16x3 layer described by a ConvArgs on conv3x3_mx_kernel (AR = 2)
int launch_conv3x3_x3(const ConvArgs& a, hipStream_t s);
unsigned char fp8_e4m3_from_float(float x);      // round to nearest even, saturating to +-448
float fp8_e4m3_to_float(unsigned char v);
// fp32 NCHW -> act with optional lo / q planes (q: scale exponent sexp); and amax |x| over an act's hi plane
// img_stride: elements between two images of src (0: c * h * w - a channel slice of a wider NCHW tensor passes its own); sat: clamp counter or null
int launch_nchw_to_act_mx(const float* src, const Act& dst, int c, hipStream_t s, long img_stride = 0, unsigned int* sat = nullptr);
int launch_act_amax(const Act& a, float* d_amax, hipStream_t s);
int launch_act_channel_amax(const Act& a, float* d_out /* a.c floats, zero-initialised */, hipStream_t s);
int launch_act_q_to_nchw(const Act& a, float* dst, int c, int which, hipStream_t s);   // tests: dequantised q planes

size_t conv3x3_packed_bytes(int c_out, int c_in_pad);
// h_w: effective fp32 weight (c_out, c_in, 3, 3); ci_map[i] = source channel index for packed channel i or -1 (zero)
void conv3x3_pack_host(const float* h_w, int c_out, int c_in, const int* ci_map, int c_in_pad, void* h_packed);
constexpr int DISCO_MAX_DEVICES = 64;
// ordinal of the calling thread's current HIP device, clamped into [0, DISCO_MAX_DEVICES) (per-device one-time setup tables)
inline int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0; return d % DISCO_MAX_DEVICES; }
// One-time per-device setup of a kernel's dynamic-LDS limit.  `done` is a zero-initialised table of DISCO_MAX_DEVICES flags (one table
// per kernel / instantiation); an entry is set only after hipFuncSetAttribute SUCCEEDED, so a failing first call is reported by every
// launch instead of being swallowed by a consumed once-flag (two threads racing here both set the same value: harmless).
inline hipError_t set_dyn_lds_once(std::atomic<int>* done, const void* kern, int bytes) {
    std::atomic<int>& f = done[current_device()];
    if (f.load(std::memory_order_acquire)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) f.store(1, std::memory_order_release);
    return e;
}
// compute units of the calling thread's current device (cached per device ordinal)
inline int num_cus_current() {
    static std::atomic<int> table[DISCO_MAX_DEVICES];
    std::atomic<int>& t = table[current_device()];
    int v = t.load(std::memory_order_relaxed);
    if (v > 0) return v;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    t.store(v, std::memory_order_relaxed);
    return v;
}
int diag_mfma_rate(int mode, int iters, double* tflops);   // diag.hip
int launch_checksum(const void* p, size_t bytes, unsigned long long* out, hipStream_t s);   // diag.hip
// ConvTranspose2d(4,s2,p1) weight (c_in,c_out,4,4) -> equivalent 3x3 conv weight (4*c_out, c_in, 3, 3), phase-major
void deconv_as_conv3x3_host(const float* h_w_iohw, int c_in, int c_out, float* h_w_oihw);
// nearest-x2-upsample followed by a 3x3 conv (c_out,c_in,3,3) -> 4-phase 3x3 conv on the low-res input
// (4*c_out, c_in, 3, 3), phase-major, taps pre-summed (sub-pixel decomposition; 4 non-zero taps per phase)
void upconv_as_conv3x3_host(const float* h_w_oihw, int c_in, int c_out, float* h_w4_oihw);
// bit t of mask[nb] set iff any weight of tap t is non-zero in 32-cout block nb
void conv3x3_tapmask_host(const float* h_w, int c_out, int c_in, uint32_t* mask /* cdiv(c_out,32) */);

// ---- direct (VALU) convs ------------------------------------------------------------------------
// first layers: Cin = 1, fp32 NCHW gray input -> act output
// `out` may carry lo and/or q planes and more (zero) channels than c_out; sat: optional clamp counter of the q planes
int launch_conv_c1(const float* d_gray, const float* d_w /*(cout,9)*/, const float* d_bias, const float* d_bn_scale,
                   const float* d_bn_shift, const Act& out, int c_out, int act, float slope, unsigned int* sat, hipStream_t s);

// ---- layout conversion --------------------------------------------------------------------------
int launch_nchw_to_act(const float* src, f16* dst, long plane, int n, int c, int h, int w, int c_pad, hipStream_t s, int sexp = 0);
int launch_act_to_nchw(const f16* src, long plane, float* dst, int n, int c, int h, int w, int c_pad, hipStream_t s, int sexp = 0);

// ---- superpixel ops -----------------------------------------------------------------------------
// feature source for pooling: either act planes (c_act channels) and/or extra fp32 NCHW channels
struct PoolArgs {
    const f16* feat_act; long feat_plane; int c_act;   // NHWC act features, channels [0,c_act)   (may be null)
    float feat_mul;                                     // 2^-sexp of feat_act: hi + lo is multiplied by it (exact)
    const float* feat_nchw; int c_nchw;                 // fp32 NCHW features, channels [c_act, c_act+c_nchw)  (may be null)
    const float* feat_bc; int c_bc;                     // fp32 (H*W, c_bc) pixel-major features shared by every image of
                                                        // the batch (the per-pixel position encoding of --spix_pos); last
    const float* prob;                                  // (n,9,H,W) fp32
    float* partial;                                     // workspace (cells,9,C+1)
    float* cnt;                                         // workspace (cells,9)
    float* tok_out; int c_tok;                          // channels [0,c_tok) as tokens (n,L,c_tok)  (may be null)
    float* nchw_out; int c_from;                        // channels [c_from,c_act+c_nchw) as NCHW (n,.,h,w) (may be null)
    float* bc_out;                                      // the pooled feat_bc channels as tokens (n,L,c_bc) (may be null)
    float* conf;                                        // (n,1,h,w) or null
    float* sizes;                                       // (n,h*w) or null
    int n, H, W, sp;
};
size_t poolfeat_ws_bytes(int n, int c, int H, int W, int sp);
int launch_poolfeat(const PoolArgs& a, hipStream_t s);
// upfeat: tokens (n,L,C) [tok_layout] or NCHW -> act planes (c_pad) and/or fp32 NCHW
int launch_upfeat(const float* tok, int tok_layout, const float* prob, int prob_rep, const Act* out_act,
                  float* out_nchw, int n, int c, int h, int w, int sp, unsigned int* sat, hipStream_t s);
// gray (n,1,H,W) -> act of out.c (16 or 32) channels with gray in channel 0, zeros elsewhere (rep: output image i reads gray image i/rep)
int launch_gray16(const float* gray, int rep, const Act& out, unsigned int* sat, hipStream_t s);
// the gray image as the 16-channel fp16 tail source of a two-source f16+fp8x2 layer: channels (g_hi, g_lo, g_hi, 0 ...), hi plane only
int launch_gray_tail(const float* gray, int rep, const Act& out, hipStream_t s);

// ---- token path ---------------------------------------------------------------------------------
constexpr int D_MODEL = 64;
constexpr int D_FF = 256;
constexpr int N_HEAD = 8;
constexpr int N_VOCAB = 313;
constexpr int ENC_LAYERS = 6;
// per layer, state_dict order: in_proj_w(192x64) in_proj_b(192) out_w(64x64) out_b(64) l1_w(256x64) l1_b(256)
//                              l2_w(64x256) l2_b(64) n1_w n1_b n2_w n2_b (64 each)
constexpr size_t ENC_LAYER_FLOATS = 192 * 64 + 192 + 64 * 64 + 64 + 256 * 64 + 256 + 64 * 256 + 64 + 4 * 64;
size_t encoder_ws_bytes(int n, int l);
// pos: (l,64) shared by all images (pos_rep = 0) or (n/pos_rep, l, 64), virtual image i using image i/pos_rep
// packed: the weights' B-fragment image for the 16-row tail kernel (launch_encoder_pack of the same `weights`), or null: 64-row tiles at every size
// attention_mfma.hip: softmax(q k^T) v on the fp32 matrix cores (long token sequences; launch_encoder_stack picks it by the token count alone)
// key_sizes (n / key_rep, l) or null: `use_mask` - keys of superpixels below 25 pixels (size < 25/256) get +1.0 on every score (model.py:121-125)
int launch_attention_mfma(const float* q, const float* k, const float* v, float* out, int n, int l, hipStream_t s, const float* key_sizes = nullptr, int key_rep = 1, float key_thr = 25.f / 256.f);
int launch_attention_valu(const float* q, const float* k, const float* v, float* out, int n, int l, hipStream_t s, const float* key_sizes = nullptr, int key_rep = 1, float key_thr = 25.f / 256.f);      // attention.hip
int launch_encoder_stack(const float* x, const float* pos, int pos_rep, const float* weights, float* out, int n, int l,
                         void* ws, hipStream_t s, const std::function<void(const void*, size_t)>* dbg = nullptr, const float* packed = nullptr,
                         const float* key_sizes = nullptr, int key_rep = 1, float key_thr = 25.f / 256.f);      // use_mask: the superpixel sizes of image i / key_rep bias image i's keys (those below key_thr)
size_t encoder_packed_floats();
int launch_encoder_pack(const float* raw, float* packed, hipStream_t s);
void position_encoding_host(float* h_pos /*(h*w,64)*/, int h, int w);
// logits: (n,L,64) x (n_out,64)^T -> NCHW (n,n_out,L)
int launch_logits(const float* x, const float* w, float* out_nchw, int n, int l, hipStream_t s, int n_out = N_VOCAB);
int launch_select_colors(const float* logit_nchw, const float* q_to_ab, float* colors, int32_t* labels, int n, int l,
                         int t_first, int t_count, hipStream_t s, int plain_rank = -1);
// colour space (models/basic.py:395-475): rgb in [0,1] <-> normalised Lab ((L-50)/50, a/110, b/110), fp32 NCHW
int launch_rgb2lab(const float* rgb, float* lab, long npix_total, long hw, hipStream_t s);
int launch_lab2rgb(const float* lab, float* rgb, long npix_total, long hw, hipStream_t s);
// image I/O either side of the forward (inference.py:23-42, util.py:91-106) and the anchor overlay (basic.py:95-117)
int launch_rgb8_to_lab(const unsigned char* src, float* gray, float* ab, float* rgbn, int n, int H, int W, int Hp, int Wp,
                       hipStream_t s);
int launch_lab_to_rgb8(const float* lab, unsigned char* dst, int n, int Hp, int Wp, int H, int W, hipStream_t s);
// cv2.resize(INTER_LINEAR) of uint8 RGB (n,H,W,3) to (Ho,Wo) fused with /255 -> Lab -> split (inference.py:32-40); resized: optional uint8 out
int launch_rgb8_resize_to_lab(const unsigned char* src, unsigned char* resized, float* gray, float* ab, float* rgbn, int n, int H, int W,
                              int Ho, int Wo, hipStream_t s);
int launch_mark_hints(const float* gray, const float* target, const float* gate, const float* base, float* out, int n, int H,
                      int W, int ks, hipStream_t s);
// annealed-mean decoding (ColorLabel.decode_ind2ab with non-integer T, basic.py:210-217)
int launch_decode_annealed(const float* logit_nchw, const float* q_to_ab, float* ab, int n, int l, float T, hipStream_t s);
int launch_nearest_bin(const float* ab_nchw, const float* q_to_ab, int32_t* labels, int n, int l, hipStream_t s);
int launch_kmeans_anchors(const float* x, const float* sizes, const int32_t* init_idx, const int32_t* fallback_rows,
                          int max_fallback, int32_t* assign, int32_t* anchor, float* hint_mask, int32_t* info, int n,
                          int l, int k, hipStream_t s, int d = 64, int channel_major = 0,   // x: (n,l,d) or (n,d,l)
                          void* ws = nullptr, size_t ws_bytes = 0,   // ws: kmeans_ws_bytes(n, l) of scratch lets images of more than 512 tokens run on several workgroups
                          unsigned int* fallback_counter = nullptr);   // device counter: += 1 per image that the several-workgroup kernel gave up and the one-workgroup kernel computed
size_t kmeans_ws_bytes(int n, int l);
size_t kmeans_state_offset();      // of an image's admission word inside its kmeans_image_stride() bytes of scratch (bit 30 set: the image fell back)
size_t kmeans_image_stride();
int launch_hint_mask_from_pos(const int32_t* pos, float* hint_mask, int n, int l, int k, hipStream_t s);
// hint[t] = W[:, :64] src + m W[:, 64+label] + m W[:, 377]           (labels, W (64,378))
//         = W[:, :64] src + m a W[:, 64] + m b W[:, 65] + m W[:, 66]  (hint2regress: colors (n,2,l), W (64,67))
int launch_hint_embed(const float* src, int src_rep, const int32_t* labels, const float* colors, const float* mask,
                      int mask_rep, const float* w_emb, float* out, int n, int l, hipStream_t s);

}  // namespace disco
