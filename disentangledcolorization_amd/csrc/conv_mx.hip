// conv_mx.hip — host side of the 3x3 implicit-GEMM conv (kernel: conv_mx_kernel.h): argument checks, the dispatch over the three
// arithmetics, weight packing, layout conversion and calibration helpers.
#include <cstring>
#include "conv_mx_kernel.h"

namespace disco {

// instantiated in conv_mx_ar0/1/2.hip (one translation unit per arithmetic: they compile in parallel)
extern template int dispatch_mx_ar<0>(const ConvMxArgs&, hipStream_t);
extern template int dispatch_mx_ar<1>(const ConvMxArgs&, hipStream_t);
extern template int dispatch_mx_ar<2>(const ConvMxArgs&, hipStream_t);
extern template int dispatch_mx_ar<3>(const ConvMxArgs&, hipStream_t);

int dispatch_mx(const ConvMxArgs& a, hipStream_t s) {
    if (a.x2q) return dispatch_mx_ar<1>(a, s);
    if (a.q6) return dispatch_mx_ar<3>(a, s);
    return a.x3 ? dispatch_mx_ar<2>(a, s) : dispatch_mx_ar<0>(a, s);
}

namespace {

// ---- layout conversion / calibration helpers ------------------------------------------------------------------------------
// fp6 e2m3 code of x: round to nearest even, saturating at +-7.5 (what v_cvt_scalef32_pk32_fp6_f16 does: tools/fp6_probe.hip)
__device__ __host__ inline unsigned fp6_code(float x) {
    const unsigned s = std::signbit(x) ? 32u : 0u;
    float ax = fabsf(x);
    if (!(ax == ax) || ax >= 7.5f) return s | 31u;
    int e = 0;
    if (ax >= 4.f) e = 2; else if (ax >= 2.f) e = 1;
    const float step = ldexpf(1.f, e - 3);
    const float v = nearbyintf(ax / step) * step;                 // nearest even multiple of the step (may reach the next binade)
    if (v >= 7.5f) return s | 31u;
    if (v < 1.f) return s | (unsigned)(v * 8.f);
    int ee = 0;
    if (v >= 4.f) ee = 2; else if (v >= 2.f) ee = 1;
    return s | ((unsigned)(ee + 1) << 3) | (unsigned)((v / ldexpf(1.f, ee) - 1.f) * 8.f);
}
__device__ inline unsigned fp6_code_dev(float x) { return fp6_code(x); }
__device__ __host__ inline float fp6_value(unsigned c) {
    const unsigned e = (c >> 3) & 3u, m = c & 7u;
    const float f = e == 0 ? m * 0.125f : ldexpf(1.f + m / 8.f, (int)e - 1);
    return (c & 32u) ? -f : f;
}

__global__ void nchw_to_act_mx_kernel(const float* __restrict__ src, f16* __restrict__ dst, long plane, long q_off, int sexp,
                                      int n, int c, int h, int w, int c_pad, int q_kind, long img_stride, unsigned int* __restrict__ sat) {
    // one thread per (image, 16-channel block, pixel): reads 16 strided fp32, writes 32 B hi (+ lo) (+ 16 B of each q plane)
    const long hw = (long)h * w, nblk = c_pad / 16;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * nblk * hw) return;
    const long pix = idx % hw; const long t = idx / hw; const int blk = (int)(t % nblk); const int img = (int)(t / nblk);
    const float sc = ldexpf(1.f, sexp), qls = ldexpf(1.f, MX_LO_SHIFT);      // every plane stores x 2^sexp
    f16 hi[16], lo[16];
    unsigned char a8[16], l8[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ch = blk * 16 + j;
        const float v = (ch < c ? src[(long)img * img_stride + ch * hw + pix] : 0.f) * sc;
        hi[j] = (f16)v;
        const float l = v - (float)hi[j];
        lo[j] = (f16)l;
        const float x = __builtin_amdgcn_fmed3f(v, -448.f, 448.f), y = __builtin_amdgcn_fmed3f(l * qls, -448.f, 448.f);
        if (sat && q_off && q_kind != 2 && (x != v || y != l * qls)) atomicAdd(sat, 1u);      // an fp8 operand clamped (counted like the conv epilogue's)
        a8[j] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(x, 0.f, 0, false) & 0xff);
        l8[j] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(y, 0.f, 0, false) & 0xff);
    }
    f16* o = dst + ((long)img * nblk + blk) * hw * 16 + pix * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = hi[j];
    if (plane) {
#pragma unroll
        for (int j = 0; j < 16; ++j) o[plane + j] = lo[j];
    }
    if (q_off && q_kind == 2) {
        // fp6 slots (test helper: the two threads of a 32-channel block write disjoint fields of the same bytes, hence the atomics;
        // the buffer is zero-filled by the caller).  Block exponent: of the largest |fp16 hi word| of the pixel's 32 channels.
        unsigned int* q32 = reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(dst) + q_off + (((long)img * (c_pad / 32) + (blk >> 1)) * 2) * hw * 32 + pix * 32);
        float bmax = 0.f;
        for (int j = 0; j < 32; ++j) {
            const int ch = (blk >> 1) * 32 + j;
            bmax = fmaxf(bmax, fabsf((float)(f16)((ch < c ? src[(long)img * img_stride + ch * hw + pix] : 0.f) * sc)));
        }
        const int sa = mx6_block_scale((f16)bmax);
        const float inv = ldexpf(1.f, 127 - sa);                       // a6 = hi / 2^(sa - 127), al6 = lo 2^11 / 2^(sa - 1 - 127)
        for (int j = 0; j < 16; ++j) {
            const int ch = (blk & 1) * 16 + j, f = mx6_channel_field(ch);
            const float v = (float)hi[j], l = (float)lo[j];          // what the conv epilogue converts: the fp16 words
            for (int pl = 0; pl < 2; ++pl) {
                const unsigned code = fp6_code_dev(pl ? l * qls * 2.f * inv : v * inv);
                const int bit = 6 * f, d = bit >> 5, o = bit & 31;
                const unsigned long long sh = (unsigned long long)code << o;
                unsigned int* w0 = q32 + pl * hw * 8 + d;
                atomicOr(w0, (unsigned)sh);
                if (sh >> 32) atomicOr(w0 + 1, (unsigned)(sh >> 32));
            }
        }
        if (!(blk & 1)) { q32[6] = (unsigned)sa; q32[hw * 8 + 6] = (unsigned)(sa - 1); }
    } else if (q_off) {
        unsigned char* q = reinterpret_cast<unsigned char*>(dst) + q_off + (((long)img * (c_pad / 32) + (blk >> 1)) * (q_kind == 1 ? 1 : 2)) * hw * 32 + pix * 32 + (blk & 1) * 16;
        if (q_kind) {
#pragma unroll
            for (int j = 0; j < 16; ++j) q[j] = l8[j];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) { q[j] = a8[j]; q[hw * 32 + j] = l8[j]; }
        }
    }
}

__global__ void act_amax_kernel(const f16* __restrict__ p, long elems, float* __restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf((float)p[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));   // non-negative floats order like uints
}

// test helper: dequantised views of an act's q planes as fp32 NCHW (true values): which = 0: a8 2^-sexp; 1: (hi + al8 2^-11) 2^-sexp
__global__ void act_q_to_nchw_kernel(const f16* __restrict__ src, long q_off, int sexp, float* __restrict__ dst, int n, int c, int h, int w,
                                     int c_pad, int which, int q_kind) {
    const long hw = (long)h * w;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * c * hw) return;
    const long pix = idx % hw; const long t = idx / hw; const int ch = (int)(t % c); const int img = (int)(t / c);
    if (q_kind == 2) {
        const unsigned int* q32 = reinterpret_cast<const unsigned int*>(reinterpret_cast<const unsigned char*>(src) + q_off + (((long)img * (c_pad / 32) + (ch >> 5)) * 2) * hw * 32 + pix * 32);
        const int bit = 6 * mx6_channel_field(ch & 31), d = bit >> 5, o = bit & 31;
        auto fld = [&](const unsigned int* w) { unsigned long long v = w[d]; if (d + 1 < 6) v |= (unsigned long long)w[d + 1] << 32; return (unsigned)((v >> o) & 63u); };
        // value = field 2^(scale byte - 127); the al6 plane carries 2^11 more (MX_LO_SHIFT), as the fp8 one does
        const int sa = (int)(q32[6] & 255u) - 127, sl = (int)(q32[hw * 8 + 6] & 255u) - 127;
        if (which == 0) dst[idx] = ldexpf(fp6_value(fld(q32)), sa - sexp);
        else dst[idx] = ldexpf((float)src[((long)img * (c_pad / 16) + (ch >> 4)) * hw * 16 + pix * 16 + (ch & 15)] + ldexpf(fp6_value(fld(q32 + hw * 8)), sl - MX_LO_SHIFT), -sexp);
        return;
    }
    const unsigned char* q = reinterpret_cast<const unsigned char*>(src) + q_off + (((long)img * (c_pad / 32) + (ch >> 5)) * (q_kind == 1 ? 1 : 2)) * hw * 32 + pix * 32 + (ch & 31);
    auto dq = [](unsigned char v) -> float {
        const int sg = v >> 7, e = (v >> 3) & 15, m = v & 7;
        const float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
        return sg ? -f : f;
    };
    if (which == 0) dst[idx] = q_kind ? 0.f : ldexpf(dq(q[0]), -sexp);        // al8-only planes have no a8 view
    else dst[idx] = ldexpf((float)src[((long)img * (c_pad / 16) + (ch >> 4)) * hw * 16 + pix * 16 + (ch & 15)] + ldexpf(dq(q[q_kind == 1 ? 0 : hw * 32]), -MX_LO_SHIFT), -sexp);
}

}  // namespace

int launch_act_q_to_nchw(const Act& a, float* dst, int c, int which, hipStream_t s) {
    const long total = (long)a.n * c * a.h * a.w;
    hipLaunchKernelGGL(act_q_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.p, (long)a.q_off, a.sexp, dst, a.n, c, a.h, a.w, a.c, which, a.q_kind);
    DISCO_LAUNCH_CHECK("act_q_to_nchw_kernel");
    return DISCO_OK;
}

int launch_nchw_to_act_mx(const float* src, const Act& dst, int c, hipStream_t s, long img_stride, unsigned int* sat) {
    if (dst.c % (dst.q_off ? 32 : 16)) { set_error("nchw_to_act_mx: padded channels %d", dst.c); return DISCO_ESHAPE; }
    const long total = (long)dst.n * (dst.c / 16) * dst.h * dst.w;
    hipLaunchKernelGGL(nchw_to_act_mx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst.p, (long)dst.plane, (long)dst.q_off, dst.sexp,
                       dst.n, c, dst.h, dst.w, dst.c, dst.q_kind, img_stride ? img_stride : (long)c * dst.h * dst.w, sat);
    DISCO_LAUNCH_CHECK("nchw_to_act_mx_kernel");
    return DISCO_OK;
}

namespace {
// max |hi| per CHANNEL of an act (calibration: the channel-disparity measure of the MX fp6 planes); out: c floats, zero-initialised
__global__ void act_channel_amax_kernel(const f16* __restrict__ p, int n, int c, long hw, float* __restrict__ out) {
    // hi plane [n][c/16][hw][16]: thread t of a workgroup keeps channel (t & 15) of its 16-channel block
    const int nblk = c >> 4;
    const int blk = blockIdx.y % nblk;
    const long img = blockIdx.y / nblk;
    const f16* base = p + ((img * nblk + blk) * hw) * 16;
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw * 16; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf((float)base[i]));
    // lanes with equal (lane & 15) hold the same channel: fold 64 -> 16
    m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    if ((threadIdx.x & 63) < 16) atomicMax(reinterpret_cast<unsigned int*>(out + blk * 16 + (threadIdx.x & 15)), __float_as_uint(m));
}
}  // namespace

int launch_act_channel_amax(const Act& a, float* d_out /* a.c floats, zeroed */, hipStream_t s) {
    const long hw = (long)a.h * a.w;
    dim3 grid((unsigned)std::min<long>((hw * 16 + 255) / 256, 32), (unsigned)(a.n * (a.c / 16)));
    hipLaunchKernelGGL(act_channel_amax_kernel, grid, dim3(256), 0, s, a.p, a.n, a.c, hw, d_out);
    DISCO_LAUNCH_CHECK("act_channel_amax_kernel");
    return DISCO_OK;
}

int launch_act_amax(const Act& a, float* d_amax, hipStream_t s) {
    hipLaunchKernelGGL(act_amax_kernel, dim3(1024), dim3(256), 0, s, a.p, (long)a.elems(), d_amax);
    DISCO_LAUNCH_CHECK("act_amax_kernel");
    return DISCO_OK;
}

// ---- fp8 e4m3 (OCP e4m3fn) on the host ----------------------------------------------------------------------------------------
float fp8_e4m3_to_float(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    const float f = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.f + m / 8.f, e - 7);
    return s ? -f : f;
}

unsigned char fp8_e4m3_from_float(float x) {
    const unsigned char sign = std::signbit(x) ? 0x80 : 0;
    float ax = std::fabs(x);
    if (!(ax == ax)) return 0x7f;
    if (ax >= 448.f) return sign | 0x7e;                      // saturate at the largest finite value
    if (ax < std::ldexp(1.f, -10)) return sign;              // below half of the smallest subnormal (2^-9): 0 (the tie rounds to even = 0)
    int e; std::frexp(ax, &e);                               // ax = f 2^e, f in [0.5, 1)  ->  exponent of the leading bit = e - 1
    int ex = e - 1;
    if (ex < -6) ex = -6;                                    // subnormal range shares the exponent of the smallest normal
    const float q = std::ldexp(ax, 3 - ex);                  // in units of the last mantissa bit: [8, 16) for normals, [0, 8) for subnormals
    float rq = std::nearbyint(q);                            // round to nearest even (default rounding mode)
    if (rq >= 16.f) { rq = 8.f; ++ex; }
    if (rq < 8.f) return sign | (unsigned char)rq;           // subnormal (or zero)
    return sign | (unsigned char)(((ex + 7) << 3) | ((int)rq - 8));
}

unsigned char fp6_e2m3_from_float(float x) { return (unsigned char)fp6_code(x); }
float fp6_e2m3_to_float(unsigned char c) { return fp6_value(c); }

size_t conv_mx_packed_bytes(int c_out, int c_in_pad, int variant) {
    return (size_t)cdiv(c_out, 32) * (variant == 1 ? (c_in_pad / 64) * 5 : c_in_pad / 16) * W_NB;
}

void conv_mx_pack_host(const float* h_w, int c_out, int c_in, const int* ci_map, int c_in_pad, void* h_packed, int32_t* h_wexp, int variant) {
    const int x2q = variant == 1;
    const bool q6 = variant == 2;
    unsigned char* dst = reinterpret_cast<unsigned char*>(h_packed);
    const int nb_n = cdiv(c_out, 32), ngrp = c_in_pad / 32;
    // per output channel: 2^wexp maps the largest |w| into [128, 256) (fp8 e4m3 tops out at 448); fp6: into [4, 8) (values in
    // (7.25, 8) saturate at 7.5: at most the one largest weight of a row, by < 7 %, in a correction term)
    for (int co = 0; co < nb_n * 32; ++co) {
        float mx = 0.f;
        if (co < c_out)
            for (size_t i = 0; i < (size_t)c_in * 9; ++i) mx = std::max(mx, std::fabs(h_w[(size_t)co * c_in * 9 + i]));
        int e = 0;
        if (mx > 0.f) { std::frexp(mx, &e); e = (q6 ? 3 : 8) - e; }     // mx 2^e in [128, 256) / [4, 8)
        if (q6 && mx > 0.f && std::ldexp(mx, e) > 7.5f) e -= 1;          // ... fp6: rather lose a bit than clamp: [3.75, 7.5]
        h_wexp[co] = std::min(std::max(e, -100), 100);
    }
    // ci_map[cip]: the real input channel of packed channel cip, -1 = none (zero weights), or CONV_MX_LO_OF(ci) = the fp16 RESIDUAL
    // w - fp16(w) of channel ci's weights (the H-only tail chunk carries a channel as (x_hi, x_lo, x_hi) against (w_h, w_h, w_l))
    auto wat = [&](int co, int cip, int tap) -> float {
        const int ci = ci_map ? ci_map[cip] : (cip < c_in ? cip : -1);
        if (co >= c_out || ci == -1) return 0.f;
        if (ci <= -2) { const float w = h_w[((size_t)co * c_in + (-2 - ci)) * 9 + tap]; return w - (float)(f16)w; }
        return h_w[((size_t)co * c_in + ci) * 9 + tap];
    };
    if (x2q) {
        // chunks of 64-channel group g64: 5 g64 + {0: w_h of channels 0-31, 1: w_l of the same, 2: w_h of 32-63, 3: w_l, 4: w8 of all 64}
        const int nck = (c_in_pad / 64) * 5;
        for (int nb = 0; nb < nb_n; ++nb)
            for (int g64 = 0; g64 < c_in_pad / 64; ++g64)
                for (int tap = 0; tap < 9; ++tap)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = nb * 32 + (lane & 31), kh = lane >> 5;
                        const float ws = std::ldexp(1.f, h_wexp[co]);
                        for (int half = 0; half < 2; ++half) {          // the two 32-channel blocks of the group
                            unsigned char* hb = dst + (((size_t)nb * nck + 5 * g64 + 2 * half) * 9 + tap) * 2 * WBLK;
                            unsigned char* lb = dst + (((size_t)nb * nck + 5 * g64 + 2 * half + 1) * 9 + tap) * 2 * WBLK;
                            for (int j = 0; j < 2; ++j) {               // piece j = fp16 weights of channels 16 j + 8 kh + 0..7 of the block
                                f16* hp = reinterpret_cast<f16*>(hb + j * WBLK + lane * 16);
                                f16* lp = reinterpret_cast<f16*>(lb + j * WBLK + lane * 16);
                                for (int i = 0; i < 8; ++i) {
                                    const float w = wat(co, g64 * 64 + half * 32 + 16 * j + 8 * kh + i, tap);
                                    hp[i] = (f16)w;
                                    lp[i] = (f16)(w - (float)hp[i]);
                                }
                            }
                        }
                        // Q chunk: the lane's 32 bytes = w8 of the 32 channels of block kh; piece j = bytes 16 j .. 16 j + 15
                        unsigned char* qb = dst + (((size_t)nb * nck + 5 * g64 + 4) * 9 + tap) * 2 * WBLK;
                        for (int j = 0; j < 2; ++j) {
                            unsigned char* qp = qb + j * WBLK + lane * 16;
                            for (int i = 0; i < 16; ++i) qp[i] = fp8_e4m3_from_float(wat(co, g64 * 64 + kh * 32 + 16 * j + i, tap) * ws);
                        }
                    }
        return;
    }
    const int nck = c_in_pad / 16;              // chunks per output block: (H, Q) per 32 channels [+ an H-only tail of 16]
    if (c_in_pad % 32 == 16) {
        // the tail chunk (ck = 2 ngrp): piece 0 = fp16 weights of packed channels 32 ngrp + 8 kh + 0..7, piece 1 unused (zero)
        for (int nb = 0; nb < nb_n; ++nb)
            for (int tap = 0; tap < 9; ++tap)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = nb * 32 + (lane & 31), kh = lane >> 5;
                    unsigned char* hb = dst + (((size_t)nb * nck + 2 * ngrp) * 9 + tap) * 2 * WBLK;
                    f16* hp = reinterpret_cast<f16*>(hb + lane * 16);
                    for (int i = 0; i < 8; ++i) hp[i] = (f16)wat(co, ngrp * 32 + 8 * kh + i, tap);
                    memset(hb + WBLK + lane * 16, 0, 16);
                }
    }
    for (int nb = 0; nb < nb_n; ++nb)
        for (int g = 0; g < ngrp; ++g)
            for (int tap = 0; tap < 9; ++tap)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = nb * 32 + (lane & 31), kh = lane >> 5;
                    // H chunk (ck = 2g): piece j (j = 0, 1) = fp16 weights of channels 16 j + 8 kh + 0..7
                    unsigned char* hb = dst + (((size_t)nb * nck + 2 * g) * 9 + tap) * 2 * WBLK;
                    // Q chunk (ck = 2g+1): lane's 32 bytes = channels 0..31 of wl8 (kh = 0) or w8 (kh = 1); piece j = bytes 16 j .. 16 j + 15
                    unsigned char* qb = dst + (((size_t)nb * nck + 2 * g + 1) * 9 + tap) * 2 * WBLK;
                    const float ws = std::ldexp(1.f, h_wexp[co]), wls = std::ldexp(1.f, h_wexp[co] + MX_LO_SHIFT);
                    if (q6) {
                        // the lane's 32 bytes = the fp6 slot of its row: field f = channel mx6_field_channel(f) of wl6 (kh = 0) / w6 (kh = 1),
                        // a little-endian stream of 32 six-bit fields in bytes 0-23, and - the MX format proper, as on the activation side -
                        // ONE E8M0 SCALE PER (output channel, 32-input-channel block, tap) in byte 24, which the MFMA takes as its weight-side
                        // scale operand straight out of dword 6 of the fragment.  Round 3 scaled a whole row (9 Cin weights) with one exponent:
                        // e2m3 has two exponent bits, so a weight below 1/8 of its ROW's maximum fell into subnormals and below 1/60 to zero,
                        // and with it its w a_lo / w_lo a corrections - harmless on Gaussian rows, half of the 1e-3 budget on heavy-tailed
                        // ones (Student-t(3): 5.0e-4 against 2.0e-4 block-scaled).  The block's largest |value| goes to (3.75, 7.5].
                        float v32[32], mx = 0.f;
                        for (int f = 0; f < 32; ++f) {
                            const float w = wat(co, g * 32 + mx6_field_channel(f), tap);
                            v32[f] = kh == 0 ? w - (float)(f16)w : w;
                            mx = std::max(mx, std::fabs(v32[f]));
                        }
                        int e = 0;
                        if (mx > 0.f) { std::frexp(mx, &e); e = 3 - e; if (std::ldexp(mx, e) > 7.5f) e -= 1; }
                        e = std::min(std::max(e, -100), 100);
                        // A/B switch for the parity experiments only (tools/precision_gpu.py --weights t3): round 3's scaling, one exponent per row
                        static const bool row_scale = std::getenv("DISCO_MX6_ROW_SCALE") != nullptr;
                        if (row_scale) e = h_wexp[co] + (kh == 0 ? MX_LO_SHIFT : 0);
                        unsigned char slot[32] = {0};
                        for (int f = 0; f < 32; ++f) {
                            const unsigned code = fp6_code(std::ldexp(v32[f], e));
                            const int bit = 6 * f;
                            const unsigned v = code << (bit & 7);
                            slot[bit >> 3] |= (unsigned char)v;
                            slot[(bit >> 3) + 1] |= (unsigned char)(v >> 8);
                        }
                        // field 2^(byte - 127) = the value; the al6 planes of the pixel side carry lo 2^11 (MX_LO_SHIFT), which the w6 side takes back
                        slot[24] = (unsigned char)(127 - e - (kh == 0 ? 0 : MX_LO_SHIFT));
                        for (int j = 0; j < 2; ++j) memcpy(qb + j * WBLK + lane * 16, slot + 16 * j, 16);
                    }
                    for (int j = 0; j < 2; ++j) {
                        f16* hp = reinterpret_cast<f16*>(hb + j * WBLK + lane * 16);
                        for (int i = 0; i < 8; ++i) hp[i] = (f16)wat(co, g * 32 + 16 * j + 8 * kh + i, tap);
                        unsigned char* qp = qb + j * WBLK + lane * 16;
                        if (q6) continue;
                        for (int i = 0; i < 16; ++i) {
                            const float w = wat(co, g * 32 + 16 * j + i, tap);
                            const float wl = w - (float)(f16)w;
                            qp[i] = kh == 0 ? fp8_e4m3_from_float(wl * wls) : fp8_e4m3_from_float(w * ws);
                        }
                    }
                }
}

// The exact power-of-two factors that carry the accumulators (domain 2^sexp_in of the sources), the parameters and the residual
// (2^res_sexp) into the domain the epilogue works in, and its result into the output's (2^out_sexp); see ConvMxArgs.
static int fold_scales(ConvMxArgs& a, int sexp_in) {
    const bool f32 = a.out_f32 != nullptr;
    if (f32) a.out_sexp = 0;
    for (int e : {sexp_in, a.out_sexp, a.res ? a.res_sexp : 0})
        if (e < -60 || e > 60) { set_error("conv3x3: scale exponent %d outside [-60, 60]", e); return DISCO_EINVAL; }
    const bool bn = a.bn_scale != nullptr;
    const bool true_dom = f32 || a.act == DISCO_ACT_TANH;            // tanh and the fused softmax work on true values
    a.force_bn = (!bn && true_dom && a.out_sexp != 0) ? 1 : 0;       // tanh into a scaled tensor: the affine (1, 0) carries the scale
    const int e_pre = true_dom ? 0 : (bn ? sexp_in : a.out_sexp);    // ReLU / LeakyReLU / none are positively homogeneous
    a.acc_mul = std::ldexp(1.f, e_pre - sexp_in);
    a.bias_mul = std::ldexp(1.f, e_pre);
    a.res_mul = std::ldexp(1.f, e_pre - (a.res ? a.res_sexp : 0));
    a.bns_mul = std::ldexp(1.f, a.out_sexp - e_pre);
    a.bnh_mul = std::ldexp(1.f, a.out_sexp);
    return DISCO_OK;
}

int launch_conv3x3_mx(const ConvMxArgs& a_in, hipStream_t s) {
    ConvMxArgs a = a_in;
    if (a.nsrc < 1 || a.nsrc > 2) { set_error("conv3x3_mx: %d sources", a.nsrc); return DISCO_EINVAL; }
    int csum = 0;
    for (int i = 0; i < a.nsrc; ++i) {
        const MxSrc& sp = a.src[i];
        if (a.x2q && a.q6) { set_error("conv3x3_mx: one arithmetic at a time"); return DISCO_EINVAL; }
        // the second source of a two-source f16+fp8x2 layer may be a 16-channel fp16 tensor WITHOUT q planes: the kernel's H-only tail
        // chunk (conv_mx_kernel.h, KIND 3; conv_mx_pack_host's `lo` channel codes make it an exact split)
        const bool tail = i == 1 && a.nsrc == 2 && !a.x2q && !a.q6 && sp.c == 16 && !sp.q_off && a.src[0].c % 32 == 0;
        if (!tail && (sp.c % (a.x2q ? 64 : 32) || !sp.q_off)) { set_error("conv3x3_mx: source %d needs q planes and a multiple of %d channels (got %d)", i, a.x2q ? 64 : 32, sp.c); return DISCO_ESHAPE; }
        const size_t per = (size_t)a.n * sp.c * sp.h * sp.w * 2;          // bytes of the hi plane = bytes of the a8|al8 planes (al8 only: half)
        const size_t bytes = tail ? per : (size_t)sp.q_off + (a.x2q ? per / 2 : per);
        if (bytes >= ((size_t)1 << 32) || (!tail && sp.q_off < per)) { set_error("conv3x3_mx: activation tensor of %zu bytes exceeds 32-bit buffer addressing; split the batch", bytes); return DISCO_ESHAPE; }
        a.src_bytes[i] = (uint32_t)bytes;
        if ((size_t)sp.h * sp.w * 32 >= (1u << 30)) { set_error("conv3x3_mx: image too large for 30-bit in-image offsets"); return DISCO_ESHAPE; }
        if (sp.sexp != a.src[0].sexp) { set_error("conv3x3_mx: the sources carry different scale exponents (%d, %d): tensors that are concatenated on read must share one", a.src[0].sexp, sp.sexp); return DISCO_ESTATE; }
        csum += sp.c;
    }
    if (csum != a.c_in) { set_error("conv3x3_mx: sources carry %d channels, layer takes %d", csum, a.c_in); return DISCO_ESHAPE; }
    if (a.x2q && a.nsrc != 1) { set_error("conv3x3_mx: the f16x2+fp8 arithmetic takes one source"); return DISCO_ESHAPE; }
    if (!a.out_f32 && a.out_q_off) {
        // which q-plane formats an instantiation can write (conv_mx_kernel.h, CAN_Q6 / CAN_Q8)
        if (a.q6 && a.out_q_kind != 2) { set_error("conv3x3_mx: the f16+fp6x2 arithmetic writes fp6 q planes only (out_q_kind %d)", a.out_q_kind); return DISCO_ESHAPE; }
        if (!a.q6 && a.out_q_kind == 2 && !(a.nsrc == 2 && !a.x2q)) { set_error("conv3x3_mx: fp6 q planes are written by the f16+fp6x2 arithmetic and by two-source f16+fp8x2 layers only"); return DISCO_ESHAPE; }
    }
    if (a.nsrc > 1 && (a.out_f32 || a.d2s_c > 0)) { set_error("conv3x3_mx: a two-source layer writes a plain activation tensor (no fp32 NCHW / depth-to-space output)"); return DISCO_ESHAPE; }
    {
        const size_t wb = conv_mx_packed_bytes(a.c_out, a.c_in, a.x2q ? 1 : 0);
        if (wb >= ((size_t)1 << 32)) { set_error("conv3x3_mx: packed weights too large"); return DISCO_ESHAPE; }
        a.w_bytes = (uint32_t)wb;
    }
    {
        const size_t oelems = (size_t)a.n * (a.d2s_c > 0 ? (size_t)a.d2s_c * 4 : (size_t)a.c_out_pad) * a.h_out * a.w_out;
        size_t ob = a.out_f32 ? (size_t)a.n * a.c_out * a.h_out * a.w_out * 4 : oelems * 2;
        if (!a.out_f32) ob = std::max(ob, std::max((size_t)a.out_plane * 2 + (a.out_plane ? oelems * 2 : 0), a.out_q_off + (a.out_q_off ? oelems * (a.out_q_kind == 1 ? 1 : 2) : 0)));
        const size_t rb = a.res ? (size_t)a.res_plane * 2 + oelems * 2 : 16;
        if (ob >= ((size_t)1 << 32) || rb >= ((size_t)1 << 32)) { set_error("conv3x3_mx: output tensor of %zu bytes exceeds 32-bit buffer addressing; split the batch", ob); return DISCO_ESHAPE; }
        if (!a.out_f32 && !a.out) { set_error("conv3x3_mx: null output"); return DISCO_EINVAL; }
        a.out_bytes = (uint32_t)ob; a.res_bytes = (uint32_t)rb;
    }
    if (!a.wexp) { set_error("conv3x3_mx: missing weight scale exponents"); return DISCO_EINVAL; }
    if (a.stride != 1 && a.stride != 2) { set_error("conv3x3_mx: stride %d", a.stride); return DISCO_ESHAPE; }
    if (a.softmax && (!a.out_f32 || a.c_out > 32)) { set_error("conv3x3_mx: the fused softmax needs the fp32 NCHW output and c_out <= 32"); return DISCO_ESHAPE; }
    if (a.d2s_c > 0 && (a.d2s_c % (a.out_q_off ? 32 : 16) || a.out_f32)) { set_error("conv3x3_mx: depth-to-space channels %d", a.d2s_c); return DISCO_ESHAPE; }
    if (a.out_q_off && !a.out_f32 && (a.d2s_c > 0 ? a.d2s_c : a.c_out_pad) % 32) { set_error("conv3x3_mx: q planes need a multiple of 32 output channels"); return DISCO_ESHAPE; }
    if (a.act == DISCO_ACT_LRELU && !(a.slope >= 0.f && a.slope <= 1.f)) { set_error("conv3x3_mx: LeakyReLU slope %g outside [0, 1]", (double)a.slope); return DISCO_ESHAPE; }
    if (a.c_out > 32 && a.c_out % 64) { set_error("conv3x3_mx: c_out %d (>32) must be a multiple of 64", a.c_out); return DISCO_ESHAPE; }
    if (int rc = fold_scales(a, a.src[0].sexp)) return rc;
    return dispatch_mx(a, s);
}

int launch_conv3x3_x3(const ConvArgs& c, hipStream_t s) {
    ConvMxArgs a{};
    if (c.nsrc < 1 || c.nsrc > 2) { set_error("conv3x3_x3: %d sources", c.nsrc); return DISCO_EINVAL; }
    int csum = 0;
    for (int i = 0; i < c.nsrc; ++i) {
        const ConvSrc& sp = c.src[i];
        if (sp.c % 16 || sp.c <= 0 || (sp.plane <= 0 && !c.c1_gray)) { set_error("conv3x3_x3: source %d needs hi + lo planes and a multiple of 16 channels (got %d)", i, sp.c); return DISCO_ESHAPE; }
        const size_t per = (size_t)c.n * sp.c * sp.h * sp.w * 2;
        const size_t bytes = (size_t)sp.plane * 2 + per;
        if (bytes >= ((size_t)1 << 32)) { set_error("conv3x3: activation tensor of %zu bytes exceeds 32-bit buffer addressing; split the batch", bytes); return DISCO_ESHAPE; }
        if ((size_t)sp.h * sp.w * 32 >= (1u << 30)) { set_error("conv3x3: image too large for 30-bit in-image offsets"); return DISCO_ESHAPE; }
        if (sp.sexp != c.src[0].sexp) { set_error("conv3x3: the sources carry different scale exponents (%d, %d): tensors that are concatenated on read must share one", c.src[0].sexp, sp.sexp); return DISCO_ESTATE; }
        a.src[i] = {sp.p, (uint32_t)((size_t)sp.plane * 2), sp.c, sp.h, sp.w, sp.up, sp.sexp};
        a.src_bytes[i] = (uint32_t)bytes;
        csum += sp.c;
    }
    if (csum != c.c_in) { set_error("conv3x3: sources carry %d channels, layer takes %d", csum, c.c_in); return DISCO_ESHAPE; }
    if (c.nsrc > 1 && (c.out_f32 || c.d2s_c > 0)) { set_error("conv3x3: a two-source layer writes a plain activation tensor (no fp32 NCHW / depth-to-space output)"); return DISCO_ESHAPE; }
    a.nsrc = c.nsrc; a.n = c.n; a.h_in = c.h_in; a.w_in = c.w_in; a.c_in = c.c_in;
    a.h_out = c.h_out; a.w_out = c.w_out; a.stride = c.stride;
    a.w = c.w; a.wexp = nullptr; a.tapmask = c.tapmask; a.c_out = c.c_out; a.c_out_pad = c.c_out_pad;
    a.bias = c.bias; a.bn_scale = c.bn_scale; a.bn_shift = c.bn_shift;
    a.res = c.res; a.res_plane = c.res_plane; a.res_sexp = c.res_sexp; a.out = c.out; a.out_plane = c.out_plane; a.out_f32 = c.out_f32; a.out_sexp = c.out_sexp;
    a.d2s_c = c.d2s_c; a.act = c.act; a.slope = c.slope; a.softmax = c.softmax; a.x3 = 1;
    if (c.c1_gray) {
        // the fused Cin = 1 producer runs on the 32 x 16 x 64 tile only: stride 1, one source of 32..64 channels (>= 2 chunks: the next image's
        // gray tile is fetched during a tile's first chunk and used in its last), >= 64 output channels, no tap mask, an image at least a tile wide
        if (c.nsrc != 1 || c.stride != 1 || c.src[0].up || c.c_in < 32 || c.c_in > 64 || c.c_out < 64 || c.tapmask || c.w_out <= 16 || c.h_out <= 8 || !c.c1_w ||
            c.src[0].h != c.h_in || c.src[0].w != c.w_in || (size_t)c.n * c.h_in * c.w_in * 4 >= ((size_t)1 << 32)) {
            set_error("conv3x3_x3: this layer shape cannot take the fused Cin = 1 producer"); return DISCO_ESHAPE;
        }
        a.c1_gray = c.c1_gray; a.c1_w = c.c1_w; a.c1_bias = c.c1_bias; a.c1_act = c.c1_act; a.c1_slope = c.c1_slope; a.c1_sexp = c.src[0].sexp;
    }
    {
        const size_t wb = (size_t)cdiv(c.c_out, 32) * (c.c_in / 16) * W_NB;          // = conv3x3_packed_bytes
        if (wb >= ((size_t)1 << 32)) { set_error("conv3x3: packed weights too large"); return DISCO_ESHAPE; }
        a.w_bytes = (uint32_t)wb;
    }
    {
        const size_t oelems = (size_t)c.n * (c.d2s_c > 0 ? (size_t)c.d2s_c * 4 : (size_t)c.c_out_pad) * c.h_out * c.w_out;
        const size_t ob = c.out_f32 ? (size_t)c.n * c.c_out * c.h_out * c.w_out * 4 : (size_t)c.out_plane * 2 + oelems * 2;
        const size_t rb = c.res ? (size_t)c.res_plane * 2 + oelems * 2 : 16;
        if (ob >= ((size_t)1 << 32) || rb >= ((size_t)1 << 32)) { set_error("conv3x3: output tensor of %zu bytes exceeds 32-bit buffer addressing; split the batch", ob); return DISCO_ESHAPE; }
        if (!c.out_f32 && (!c.out || c.out_plane <= 0)) { set_error("conv3x3_x3: the output needs hi + lo planes"); return DISCO_EINVAL; }
        if (c.res && c.res_plane <= 0) { set_error("conv3x3_x3: the residual needs hi + lo planes"); return DISCO_EINVAL; }
        a.out_bytes = (uint32_t)ob; a.res_bytes = (uint32_t)rb;
    }
    if (c.stride != 1 && c.stride != 2) { set_error("conv3x3: stride %d", c.stride); return DISCO_ESHAPE; }
    if (c.softmax && (!c.out_f32 || c.c_out > 32)) { set_error("conv3x3: the fused softmax needs the fp32 NCHW output and c_out <= 32"); return DISCO_ESHAPE; }
    if (c.d2s_c > 0 && (c.d2s_c % 16 || c.out_f32)) { set_error("conv3x3: depth-to-space needs a multiple of 16 channels (got %d) and an activation output", c.d2s_c); return DISCO_ESHAPE; }
    if (c.act == DISCO_ACT_LRELU && !(c.slope >= 0.f && c.slope <= 1.f)) { set_error("conv3x3: LeakyReLU slope %g outside [0, 1]", (double)c.slope); return DISCO_ESHAPE; }
    if (c.c_out > 32 && c.c_out % 64) { set_error("conv3x3: c_out %d (>32) must be a multiple of 64", c.c_out); return DISCO_ESHAPE; }
    if (int rc = fold_scales(a, c.src[0].sexp)) return rc;
    return dispatch_mx(a, s);
}

}  // namespace disco
