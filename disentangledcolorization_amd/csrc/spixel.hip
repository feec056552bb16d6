// spixel.hip — un-pooling (K15 of SURVEY §2b) and the gray / upfeat producers of the HourGlass2's input; the pooling kernels live in pool.hip.
//
// Reference: models/basic.py:338-376 (upfeat); model.py:194-196 (the HourGlass2's input).
// Slot c = (dy+1)*3 + (dx+1): a pixel of cell (a,b) with probability P_c belongs to superpixel (a+dy, b+dx).
#include "common.h"

namespace disco {

namespace {

// upfeat: out(p) = sum_c P_c(p) tok[cell(p) + (dy,dx)].  thread = (image, pixel), looping over the 16-channel blocks:
// the 9 probabilities of a pixel are read once, the neighbour tokens are L1/L2-resident broadcasts (a 16x16 cell
// shares them), and a wave writes 64 consecutive 32-byte pixels per block and plane.
__global__ __launch_bounds__(256) void upfeat_kernel(const float* __restrict__ tok, int tok_layout,
                                                     const float* __restrict__ prob, int prob_rep, f16* out_act,
                                                     long out_plane, long q_off, int sexp, unsigned int* sat_out, float* out_nchw,
                                                     int n, int c, int hs, int ws, int sp) {
    const int H = hs * sp, W = ws * sp, L = hs * ws;
    const long HW = (long)H * W;
    const int nblk = c >> 4;
    const long total = (long)n * HW;
    unsigned sat = 0;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long p = t % HW;
        const int img = (int)(t / HW);
        const int y = (int)(p / W), x = (int)(p % W);
        const int cy = y / sp, cx = x / sp;
        const float* pr = prob + (long)(img / prob_rep) * 9 * HW + p;
        float pw[9];
        int tokidx[9];
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int ty = cy + s / 3 - 1, tx = cx + s % 3 - 1;
            pw[s] = pr[s * HW];
            tokidx[s] = (ty < 0 || ty >= hs || tx < 0 || tx >= ws) ? -1 : ty * ws + tx;
        }
        for (int blk = 0; blk < nblk; ++blk) {
            float acc[16];
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                float tv[16];
                if (tokidx[s] < 0) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) tv[j] = 0.f;
                } else if (tok_layout) {
                    const float4* tp = reinterpret_cast<const float4*>(tok + ((long)img * L + tokidx[s]) * c + blk * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float4 a4 = tp[q]; tv[4 * q] = a4.x; tv[4 * q + 1] = a4.y; tv[4 * q + 2] = a4.z; tv[4 * q + 3] = a4.w; }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) tv[j] = tok[((long)img * c + blk * 16 + j) * L + tokidx[s]];
                }
                // the reference multiplies then accumulates in slot order (no fused multiply-add)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = s == 0 ? mul_rn(tv[j], pw[0]) : add_rn(acc[j], mul_rn(tv[j], pw[s]));
            }
            if (out_act) {
                store_act8(out_act, out_plane, q_off, sexp, img, blk, 0, p, HW, nblk, acc, &sat);
                store_act8(out_act, out_plane, q_off, sexp, img, blk, 1, p, HW, nblk, acc + 8, &sat);
            }
            if (out_nchw) {
#pragma unroll
                for (int j = 0; j < 16; ++j) out_nchw[((long)img * c + blk * 16 + j) * HW + p] = acc[j];
            }
        }
    }
    if (sat_out && sat) atomicAdd(sat_out, sat);
}

// sp == 16, token-major input, act output: ONE WORKGROUP PER CELL (thread = pixel of the cell).  The nine neighbour tokens are then
// uniform over the workgroup: their addresses depend on blockIdx only, so they arrive through the scalar cache into SGPRs
// (s_load_dwordx16) instead of 36 sixteen-byte vector loads per pixel and 16-channel block - the pixel-per-thread kernel above is
// bound by exactly those (a row of 64 pixels spans 4 cells).  Same products and the same slot order as above.
// TOK_LDS (round 5, grids that do not fill the GPU): the nine token rows are fetched ONCE, with one vector load per thread, into LDS and read
// from there as broadcasts.  The scalar loads of the other variant are issued trip by trip (576 floats do not fit the 102 SGPRs), so a
// workgroup alone on its CU sits out eight scalar-cache round trips one after the other: 38 us for the 256 cells of one 256 x 256 image.
// (accesses of different types to one byte array: may_alias types, so that type-based alias analysis has no say in their order either)
typedef f16x8 __attribute__((may_alias)) f16x8_a;
typedef uint4 __attribute__((may_alias)) uint4_a;
template <bool TOK_LDS>
__global__ __launch_bounds__(256) void upfeat_cell_kernel(const float* __restrict__ tok, const float* __restrict__ prob, int prob_rep,
                                                          f16* out_act, long out_plane, long q_off, int sexp, unsigned int* sat_out,
                                                          int c, int hs, int ws) {
    const int W = ws * 16, L = hs * ws;
    const long HW = (long)(hs * 16) * W;
    const int nblk = c >> 4;
    const int cell = blockIdx.x;
    const int cx = cell % ws, cy = (cell / ws) % hs, img = cell / L;
    const long p = (long)(cy * 16 + (threadIdx.x >> 4)) * W + cx * 16 + (threadIdx.x & 15);
    const float* pr = prob + (long)(img / prob_rep) * 9 * HW + p;
    // a neighbour outside the grid is a zero token in the reference; here its WEIGHT is zeroed and the token row is read from a
    // clamped (valid) address, so that the scalar loads are unconditional: term = tok * 0 = 0 either way
    float pw[9];
    const float* trow[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int ty = cy + s / 3 - 1, tx = cx + s % 3 - 1;
        const bool inside = ty >= 0 && ty < hs && tx >= 0 && tx < ws;               // uniform over the workgroup
        pw[s] = inside ? pr[s * HW] : 0.f;
        trow[s] = tok + ((long)img * L + (inside ? ty * ws + tx : cy * ws + cx)) * c;
    }
    __shared__ __attribute__((aligned(16))) float s_tok[TOK_LDS ? 9 * 64 : 4];
    if (TOK_LDS) {
        for (int u = threadIdx.x; u < 9 * (c >> 2); u += 256) {
            const int sidx = u / (c >> 2), c4 = (u - sidx * (c >> 2)) * 4;
            *reinterpret_cast<float4*>(s_tok + sidx * 64 + c4) = *reinterpret_cast<const float4*>(trow[sidx] + c4);
        }
        __syncthreads();
    }
    unsigned sat = 0;
    // packed arithmetic written out by hand (this file is built without the SLP vectoriser, see build.py): the nine weights as real
    // register pairs, token pairs straight from SGPR pairs; per element the same mul, add sequence in slot order as the scalar code
    f32x2_t pw2[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) pw2[s] = pair_of(pw[s]);
    // Stores (round 4): a thread owns one pixel, and a pixel's 16 channels of a plane are 32 contiguous bytes, so a thread-per-pixel store of
    // 8 channels is 16 bytes at a 32-byte lane stride - every cache line half written per instruction, the q planes a quarter (round 3:
    // 3.1 TB/s).  Now a wave collects a whole 16-channel block (32-channel block for the q planes), passes it through 2 KiB of LDS - whose
    // image IS the memory image of the wave's four 16-pixel row segments - and every store instruction writes whole lines.  Same values.
    __shared__ __attribute__((aligned(16))) unsigned char s_tr[4][2048];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned char* tr = s_tr[wave];
    const float sc = ldexpf(1.f, sexp), qls = ldexpf(1.f, MX_LO_SHIFT);
    // dense store of the 2 KiB in `tr` (4 rows x 16 pixels x 32 bytes) to plane base `dst` (byte address of pixel 0 of the image's block)
    // The lanes of a wave exchange data through `tr` without a barrier instruction (LDS operations of a wave execute in order) - which the
    // COMPILER has to be told: for one thread the writes (tr + 32 lane + ...) and the reads (tr + 1024 h2 + 16 lane) provably do not overlap, so it
    // may interleave them.  It did, the moment qa / ql left scratch memory (a ds_read_b128 of the row image between the two ds_write_b128 that
    // complete it: pred_colors wrong by 0.97, caught by the output hashes of tools/small_batch_latency.py); before that the order held by luck.
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto store_rows = [&](unsigned char* dst) {
        wave_sync();                       // the row image is complete
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const uint4 v = *reinterpret_cast<const uint4_a*>(tr + h2 * 1024 + lane * 16);
            const int row = cy * 16 + wave * 4 + h2 * 2 + (lane >> 5);
            *reinterpret_cast<uint4*>(dst + ((long)row * W + cx * 16) * 32 + (lane & 31) * 16) = v;
        }
        wave_sync();                       // ... and read before it is rewritten
    };
    unsigned char* const base = reinterpret_cast<unsigned char*>(out_act);
    uint2 qa[4], ql[4];                                                     // a8 / al8 of the current 32-channel block (8 channels per trip)
    f16x8 lpark;                                                            // lo words of the block's first half
    // gridDim.y > 1 (small grids): the workgroups of a cell split its 32-channel blocks (4 trips each) - the same arithmetic per element, a
    // shorter serial walk per workgroup (one image: 256 cells x 8 trips -> 512 workgroups x 4 trips, 38 -> ~22 us)
    const int trips = 2 * nblk / (int)gridDim.y;
    const int hb0 = (int)blockIdx.y * trips;
    for (int hb = hb0; hb < hb0 + trips; ++hb) {       // 8 channels per trip: 9 x 8 token values in SGPRs
        f32x2_t acc2[4];
#pragma unroll
        for (int s = 0; s < 9; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2_t tv = TOK_LDS ? f32x2_t{s_tok[s * 64 + hb * 8 + 2 * j], s_tok[s * 64 + hb * 8 + 2 * j + 1]}
                                           : f32x2_t{trow[s][hb * 8 + 2 * j], trow[s][hb * 8 + 2 * j + 1]};
                acc2[j] = s == 0 ? mul_rn2(tv, pw2[0]) : add_rn2(acc2[j], mul_rn2(tv, pw2[s]));
            }
        }
        const float acc[8] = {acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y, acc2[2].x, acc2[2].y, acc2[3].x, acc2[3].y};
        // the split of store_act8(): xs = x 2^sexp, hi = fp16(xs), lo = fp16(xs - hi), a8 = fp8(xs), al8 = fp8((xs - hi) 2^11)
        f16x8 h, l;
        float v[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = acc[j] * sc; h[j] = (f16)v[j]; lo[j] = v[j] - (float)h[j]; l[j] = (f16)lo[j]; }
        const int blk = hb >> 1, half = hb & 1;
        // (LDS operations of a wave execute in order: a row image is complete when store_rows reads it, and read before it is rewritten)
        *reinterpret_cast<f16x8_a*>(tr + lane * 32 + half * 16) = h;
        if (half == 0) lpark = l;
        else {
            unsigned char* hb_base = base + (((long)img * nblk + blk) * HW) * 32;
            store_rows(hb_base);
            if (out_plane) {
                *reinterpret_cast<f16x8_a*>(tr + lane * 32) = lpark;
                *reinterpret_cast<f16x8_a*>(tr + lane * 32 + 16) = l;
                store_rows(hb_base + out_plane * 2);
            }
        }
        if (q_off) {
            uint2 a, b;
            b.x = pack_fp8x4(lo[0] * qls, lo[1] * qls, lo[2] * qls, lo[3] * qls, &sat); b.y = pack_fp8x4(lo[4] * qls, lo[5] * qls, lo[6] * qls, lo[7] * qls, &sat);
            a.x = pack_fp8x4(v[0], v[1], v[2], v[3], &sat); a.y = pack_fp8x4(v[4], v[5], v[6], v[7], &sat);
            // (statically indexed: `qa[hb & 3] = a` put both arrays into scratch memory - 48 bytes per lane written and read back per pixel next to
            // the 256 bytes of output)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((hb & 3) == k) { qa[k] = a; ql[k] = b; }
            if ((hb & 3) == 3) {
                unsigned char* qb = base + q_off + (((long)img * (nblk >> 1) + (blk >> 1)) * 2) * HW * 32;
                *reinterpret_cast<uint4_a*>(tr + lane * 32) = uint4{qa[0].x, qa[0].y, qa[1].x, qa[1].y};
                *reinterpret_cast<uint4_a*>(tr + lane * 32 + 16) = uint4{qa[2].x, qa[2].y, qa[3].x, qa[3].y};
                store_rows(qb);
                *reinterpret_cast<uint4_a*>(tr + lane * 32) = uint4{ql[0].x, ql[0].y, ql[1].x, ql[1].y};
                *reinterpret_cast<uint4_a*>(tr + lane * 32 + 16) = uint4{ql[2].x, ql[2].y, ql[3].x, ql[3].y};
                store_rows(qb + HW * 32);
            }
        }
    }
    if (sat_out && sat) atomicAdd(sat_out, sat);
}

// generic (c not a multiple of 8) NCHW-only variant: thread = (pixel, channel)
__global__ void upfeat_scalar_kernel(const float* __restrict__ tok, const float* __restrict__ prob, float* out_nchw,
                                     int n, int c, int hs, int ws, int sp) {
    const int H = hs * sp, W = ws * sp, L = hs * ws;
    const long HW = (long)H * W;
    const long total = (long)n * c * HW;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long p = t % HW;
        const int ch = (int)((t / HW) % c);
        const int img = (int)(t / (HW * c));
        const int y = (int)(p / W), x = (int)(p % W);
        const int cy = y / sp, cx = x / sp;
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int ty = cy + s / 3 - 1, tx = cx + s % 3 - 1;
            const float tv = (ty < 0 || ty >= hs || tx < 0 || tx >= ws) ? 0.f : tok[((long)img * c + ch) * L + ty * ws + tx];
            const float term = mul_rn(tv, prob[((long)img * 9 + s) * HW + p]);
            acc = s == 0 ? term : add_rn(acc, term);
        }
        out_nchw[t] = acc;
    }
}

__global__ void gray16_kernel(const float* __restrict__ gray, int rep, f16* out, long out_plane, long q_off, int sexp, int nblk,
                              unsigned int* sat_out, long npix_total, long HW) {
    unsigned sat = 0;
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix_total; pix += (long)gridDim.x * blockDim.x) {
        const long img = pix / HW, p = pix % HW;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        for (int blk = 0; blk < nblk; ++blk)
            for (int half = 0; half < 2; ++half) {
                v[0] = (blk == 0 && half == 0) ? gray[(img / rep) * HW + p] : 0.f;
                store_act8(out, out_plane, q_off, sexp, img, blk, half, p, HW, nblk, v, &sat);
            }
    }
    if (sat_out && sat) atomicAdd(sat_out, sat);
}

// gray as (g_hi, g_lo, g_hi, 0 x 13) per pixel, 32 bytes: against the weights (w_h, w_h, w_l) of conv_mx_pack_host's tail chunk ONE K = 16
// MFMA per tap forms w_h g_hi + w_h g_lo + w_l g_hi - the f16x3 split of the product, exact to fp32 rounding (model.py:194: the gray
// channel of the HourGlass2's input)
__global__ void gray_tail_kernel(const float* __restrict__ gray, int rep, f16* __restrict__ out, int sexp, long npix_total, long HW) {
    const float sc = ldexpf(1.f, sexp);
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix_total; pix += (long)gridDim.x * blockDim.x) {
        const long img = pix / HW, p = pix % HW;
        const float v = gray[(img / rep) * HW + p] * sc;
        const f16 hi = (f16)v, lo = (f16)(v - (float)hi);
        f16x8 a = {hi, lo, hi, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f}, z = {};
        f16x8* o = reinterpret_cast<f16x8*>(out + pix * 16);
        o[0] = a; o[1] = z;
    }
}

inline int grid_for(long total, int block = 256) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

int launch_upfeat(const float* tok, int tok_layout, const float* prob, int prob_rep, const Act* out_act, float* out_nchw, int n,
                  int c, int h, int w, int sp, unsigned int* sat, hipStream_t s) {
    if (c % 16 == 0) {
        if (out_act && (out_act->c != c || (out_act->q_off && c % 32))) { set_error("upfeat: act of %d channels for c=%d", out_act->c, c); return DISCO_ESHAPE; }
        const long total = (long)n * h * sp * w * sp;
        if (sp == 16 && tok_layout && out_act && !out_nchw) {
            // up to four workgroups per CU the launch is a latency matter: token rows through LDS (same arithmetic, same results)
            if (c <= 64 && (long)n * h * w <= 4L * num_cus_current()) {
                // ... and below two workgroups per CU the cells' 32-channel blocks go to workgroups of their own
                const int ysplit = (c % 64 == 0 && (long)n * h * w <= 2L * num_cus_current()) ? 2 : 1;
                hipLaunchKernelGGL(upfeat_cell_kernel<true>, dim3(n * h * w, ysplit), dim3(256), 0, s, tok, prob, prob_rep, out_act->p, (long)out_act->plane,
                                   (long)out_act->q_off, out_act->sexp, sat, c, h, w);
            }
            else
                hipLaunchKernelGGL(upfeat_cell_kernel<false>, dim3(n * h * w), dim3(256), 0, s, tok, prob, prob_rep, out_act->p, (long)out_act->plane,
                                   (long)out_act->q_off, out_act->sexp, sat, c, h, w);
            DISCO_LAUNCH_CHECK("upfeat_cell_kernel");
            return DISCO_OK;
        }
        hipLaunchKernelGGL(upfeat_kernel, dim3(grid_for(total)), dim3(256), 0, s, tok, tok_layout, prob, prob_rep,
                           out_act ? out_act->p : nullptr, out_act ? (long)out_act->plane : 0L, out_act ? (long)out_act->q_off : 0L,
                           out_act ? out_act->sexp : 0, sat, out_nchw, n, c, h, w, sp);
        DISCO_LAUNCH_CHECK("upfeat_kernel");
        return DISCO_OK;
    }
    if (tok_layout || out_act || prob_rep != 1) { set_error("upfeat: c=%d needs the NCHW path", c); return DISCO_ESHAPE; }
    const long total = (long)n * c * h * sp * w * sp;
    hipLaunchKernelGGL(upfeat_scalar_kernel, dim3(grid_for(total)), dim3(256), 0, s, tok, prob, out_nchw, n, c, h, w, sp);
    DISCO_LAUNCH_CHECK("upfeat_scalar_kernel");
    return DISCO_OK;
}

int launch_gray_tail(const float* gray, int rep, const Act& out, hipStream_t s) {
    if (out.c != 16 || out.plane || out.q_off) { set_error("gray_tail: a 16-channel hi-only tensor, got %d channels", out.c); return DISCO_ESHAPE; }
    const long HW = (long)out.h * out.w, total = (long)out.n * HW;
    hipLaunchKernelGGL(gray_tail_kernel, dim3(grid_for(total)), dim3(256), 0, s, gray, rep, out.p, out.sexp, total, HW);
    DISCO_LAUNCH_CHECK("gray_tail_kernel");
    return DISCO_OK;
}

int launch_gray16(const float* gray, int rep, const Act& out, unsigned int* sat, hipStream_t s) {
    if (out.c % (out.q_off ? 32 : 16)) { set_error("gray16: %d channels", out.c); return DISCO_ESHAPE; }
    const long HW = (long)out.h * out.w, total = (long)out.n * HW;
    hipLaunchKernelGGL(gray16_kernel, dim3(grid_for(total)), dim3(256), 0, s, gray, rep, out.p, (long)out.plane, (long)out.q_off, out.sexp,
                       out.c / 16, sat, total, HW);
    DISCO_LAUNCH_CHECK("gray16_kernel");
    return DISCO_OK;
}

}  // namespace disco
