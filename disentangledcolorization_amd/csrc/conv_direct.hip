// conv_direct.hip — the few convolutions that do not fit the MFMA implicit GEMM, as exact-fp32 VALU
// kernels, plus the converters between fp32 NCHW and the internal activation layout
// (channel-blocked [N][C/16][H][W][16] fp16 hi plane, optional lo plane and fp8 q planes: struct Act in common.h).
//
//   conv_c1          Cin = 1 first layers  (segnet conv0a, network.py:263; repnet conv1_2.0, :152)
// (pred_mask0 + softmax9, enhanceNet.outConv and the ConvTranspose2d layers run on the MFMA kernel: conv_mfma2.hip epilogues.)
#include <algorithm>
#include "common.h"

namespace disco {

namespace {

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == DISCO_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DISCO_ACT_LRELU) return v >= 0.f ? v : v * slope;
    if (act == DISCO_ACT_TANH) return tanhf(v);
    return v;
}

// ---- Cin = 1 ---------------------------------------------------------------------------------------
// thread = (image, 16-channel block, pixel, channel half), half fastest: lanes 2i and 2i+1 compute channels 0-7 and
// 8-15 of the same pixel, so every 16-byte store instruction of a wave covers 1 KiB of contiguous memory per plane
// (one thread per pixel issued 16-byte stores at a 32-byte lane stride: 1.8 TB/s).  blockIdx.y = (image, block):
// the block's 16x9 weights and its bias / BN affine sit in LDS.
__global__ __launch_bounds__(256) void conv_c1_kernel(const float* __restrict__ gray, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ bsc,
                                                      const float* __restrict__ bsh, f16* out, long out_plane, long q_off, int sexp,
                                                      unsigned int* sat_out, int n, int h, int wd, int c_out, int c_pad, int act,
                                                      float slope, int q_kind) {
    __shared__ float sw[16 * 9 + 48];
    const int nblk = c_pad >> 4;
    const int blk = blockIdx.y % nblk;
    const long img = blockIdx.y / nblk;
    const long hw = (long)h * wd;
    const bool real = blk * 16 < c_out;                 // channel blocks beyond c_out are zero padding
    if (threadIdx.x < 144) sw[threadIdx.x] = real ? w[blk * 144 + threadIdx.x] : 0.f;
    else if (threadIdx.x < 160) sw[threadIdx.x] = real && bias ? bias[blk * 16 + threadIdx.x - 144] : 0.f;
    else if (threadIdx.x < 176) sw[threadIdx.x] = real && bsc ? bsc[blk * 16 + threadIdx.x - 160] : 1.f;
    else if (threadIdx.x < 192) sw[threadIdx.x] = real && bsh ? bsh[blk * 16 + threadIdx.x - 176] : 0.f;
    __syncthreads();
    const float* gi = gray + img * hw;
    unsigned sat = 0;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < 2 * hw; u += (long)gridDim.x * blockDim.x) {
        const long p = u >> 1;
        const int half = (int)(u & 1);
        const int x = (int)(p % wd), y = (int)(p / wd);
        float in[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xx = x + kx - 1;
                in[ky * 3 + kx] = (yy >= 0 && yy < h && xx >= 0 && xx < wd) ? gi[(long)yy * wd + xx] : 0.f;
            }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = half * 8 + j;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) s = fmaf(in[k], sw[c * 9 + k], s);
            s += sw[144 + c];
            s = apply_act(s, act, slope);
            v[j] = real ? s * sw[160 + c] + sw[176 + c] : 0.f;
        }
        store_act8(out, out_plane, q_off, sexp, img, blk, half, p, hw, nblk, v, &sat, q_kind);
    }
    if (sat_out && sat) atomicAdd(sat_out, sat);
}

// ---- layout converters -----------------------------------------------------------------------------
__global__ void nchw_to_act_kernel(const float* __restrict__ src, f16* dst, long plane, int n, int c, int h, int w,
                                   int c_pad, float sc) {
    const long total = (long)n * h * w * c_pad;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        // t enumerates the channel-blocked destination: ((img*C/16 + blk)*hw + p)*16 + lane16
        const long hw = (long)h * w;
        const int l16 = (int)(t & 15);
        const long q = t >> 4;
        const long p = q % hw;
        const int blk = (int)((q / hw) % (c_pad >> 4));
        const long img = q / (hw * (c_pad >> 4));
        const int ch = blk * 16 + l16;
        const float v = (ch < c ? src[(img * c + ch) * hw + p] : 0.f) * sc;
        const f16 hi = (f16)v;
        dst[t] = hi;
        dst[t + plane] = (f16)(v - (float)hi);
    }
}

__global__ void act_to_nchw_kernel(const f16* __restrict__ src, long plane, float* dst, int n, int c, int h, int w,
                                   int c_pad, float inv_sc) {
    const long total = (long)n * c * h * w;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long hw = (long)h * w;
        const long p = t % hw;
        const int ch = (int)((t / hw) % c);
        const long img = t / (hw * c);
        const long s = ((img * (c_pad >> 4) + (ch >> 4)) * hw + p) * 16 + (ch & 15);
        dst[t] = ((float)src[s] + (float)src[s + plane]) * inv_sc;
    }
}

inline int grid_for(long total, int block = 256) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

int launch_conv_c1(const float* d_gray, const float* d_w, const float* d_bias, const float* d_bn_scale,
                   const float* d_bn_shift, const Act& out, int c_out, int act, float slope, unsigned int* sat, hipStream_t s) {
    if (c_out % 16 || out.c % (out.q_off ? 32 : 16) || out.c < c_out) { set_error("conv_c1: c_out %d into a %d-channel act", c_out, out.c); return DISCO_ESHAPE; }
    const long hw = (long)out.h * out.w;
    // few fat workgroups per (image, block): the 192-float parameter staging + barrier is paid once per workgroup
    // ... unless that leaves CUs idle (one image): then more, thinner workgroups, up to one pixel pair per thread - a thread's loop iterations
    // are dependent load -> compute -> store round trips, 8 of them in a row at one 256 x 256 image (21 us a launch)
    const long per_block = std::max<long>(64, cdiv(8 * num_cus_current(), out.n * (out.c / 16)));
    dim3 grid((unsigned)std::min<long>((2 * hw + 255) / 256, per_block), (unsigned)(out.n * (out.c / 16)));
    hipLaunchKernelGGL(conv_c1_kernel, grid, dim3(256), 0, s, d_gray, d_w, d_bias, d_bn_scale, d_bn_shift, out.p, (long)out.plane,
                       (long)out.q_off, out.sexp, sat, out.n, out.h, out.w, c_out, out.c, act, slope, out.q_kind);
    DISCO_LAUNCH_CHECK("conv_c1_kernel");
    return DISCO_OK;
}

int launch_nchw_to_act(const float* src, f16* dst, long plane, int n, int c, int h, int w, int c_pad, hipStream_t s, int sexp) {
    const long total = (long)n * h * w * c_pad;
    hipLaunchKernelGGL(nchw_to_act_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, dst, plane, n, c, h, w, c_pad, ldexpf(1.f, sexp));
    DISCO_LAUNCH_CHECK("nchw_to_act_kernel");
    return DISCO_OK;
}

int launch_act_to_nchw(const f16* src, long plane, float* dst, int n, int c, int h, int w, int c_pad, hipStream_t s, int sexp) {
    const long total = (long)n * c * h * w;
    hipLaunchKernelGGL(act_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, plane, dst, n, c, h, w, c_pad, ldexpf(1.f, -sexp));
    DISCO_LAUNCH_CHECK("act_to_nchw_kernel");
    return DISCO_OK;
}

}  // namespace disco
