// pool.hip — superpixel pooling / sizes (K4, K5 of SURVEY §2b).  A translation unit of its own because it is compiled with
// -fno-slp-vectorize (build.py): see the note at pool_partial_kernel's 16-byte path.
//
// Reference: models/basic.py:274-324 (poolfeat), :327-335 (get_spixel_size), :338-376 (upfeat).
// Slot c = (dy+1)*3 + (dx+1): a pixel of cell (a,b) with probability P_c belongs to superpixel
// (a+dy, b+dx).  The reference evaluates nine avg_pool2d + pad/shift/accumulate passes over the
// full-resolution tensor; here every cell is read ONCE:
//   pass 1 (one workgroup per cell): partial[cell][c][ch] = mean_{p in cell} feat(p,ch) P_c(p),
//           ch == C is the all-ones channel (-> probability mass); cnt[cell][c] = #{p: P_c(p) == max_c' P_c'(p)}
//   pass 2 (one thread per (superpixel, ch)): num = sum_{c=0..8} partial[cell(i-dy,j-dx)][c][ch]
//           accumulated in slot order 0..8 like the reference; pooled = num / (den + 1e-8);
//           size = (sum_c cnt) / sp^2  (exact: multiples of 1/sp^2)
#include "common.h"


namespace disco {

namespace {

// blockDim = 256; grid = n*h*w cells
__global__ __launch_bounds__(256) void pool_partial_kernel(PoolArgs a) {
    const int C = a.c_act + a.c_nchw + a.c_bc;  // feature channels (ones channel is index C)
    const int hs = a.H / a.sp, ws = a.W / a.sp;
    const int cell = blockIdx.x;
    const int cx = cell % ws, cy = (cell / ws) % hs, n = cell / (ws * hs);
    const int npix = a.sp * a.sp;
    const long HW = (long)a.H * a.W;
    const float* prob = a.prob + (long)n * 9 * HW;
    extern __shared__ float sm[];      // [npix][9] probabilities, then reduction scratch
    float* sp_prob = sm;
    float* red = sm + npix * 9;       // [4 pixel groups][9 slots][64 channels] partial sums
    __shared__ int s_cnt[9];
    if (threadIdx.x < 9) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    // load probabilities of the cell and count hard assignments (ties count for every maximal slot)
    for (int p = threadIdx.x; p < npix; p += blockDim.x) {
        const int py = p / a.sp, px = p % a.sp;
        const long off = (long)(cy * a.sp + py) * a.W + cx * a.sp + px;
        float v[9], m = -1.f;
#pragma unroll
        for (int c = 0; c < 9; ++c) { v[c] = prob[c * HW + off]; m = fmaxf(m, v[c]); sp_prob[p * 9 + c] = v[c]; }
#pragma unroll
        for (int c = 0; c < 9; ++c) if (v[c] == m) atomicAdd(&s_cnt[c], 1);
    }
    __syncthreads();
    // thread = (channel ch = tid & 63 (+64 second round), pixel group g = tid >> 6)
    const int g = threadIdx.x >> 6, lanech = threadIdx.x & 63;
    const float inv = 1.f / (float)npix;
    int ch_first = 0;
    if (a.sp == 16 && a.c_act == 64) {
        // NOTE (round 3): built WITHOUT the SLP vectoriser (build.py).  Its packed code for the 72 multiply-adds below broadcasts the
        // LDS-loaded probabilities with op_sel, and `v_pk_fma_f32 ... op_sel:[0,1,0]` (low result half from the HIGH dword of a source)
        // returns a wrong low half while other waves of the CU issue MFMAs - tools/pk_fault_repro.hip shows it with registers only;
        // here it meant a missing contribution of ~0.1-1 % in a handful of tokens about once per 12 concurrent small forwards
        // (HISTORY.md section 4, profiles/r03_stagger_probe.txt, r03_pk_fma_op_sel_fault.txt; tools/audit_op_sel.py guards every file).
        // The 64 act channels of a 16x16 cell, 16 bytes per load: thread = (8-channel group q = tid & 7, pixel subset r = tid >> 3),
        // pixel p = 32 i + r (i = 0..7): a wave reads 8 consecutive pixels x 64 channels = 4 planes x 256 contiguous bytes per
        // load instruction (the scalar path below moves 2 bytes per lane: 8x the instructions, address-unit bound).  Each thread
        // keeps 8 channels x 9 slots; the 32 pixel subsets are combined in a fixed order: xor butterfly over the 8 subsets of a
        // wave, then the 4 waves through LDS.
        const int q = threadIdx.x & 7, r = threadIdx.x >> 3;
        const long cell0 = (long)(cy * 16) * a.W + cx * 16;
        const f16* s16 = a.feat_act + (((long)n * 4 + (q >> 1)) * HW + cell0) * 16 + (q & 1) * 8;
        float acc[9][8];
#pragma unroll
        for (int c = 0; c < 9; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
#pragma unroll 2
        for (int i = 0; i < 8; ++i) {
            const int p = 32 * i + r;
            const int off = ((p >> 4) * a.W + (p & 15)) * 16;
            const f16x8 h = *reinterpret_cast<const f16x8*>(s16 + off);
            const f16x8 l = *reinterpret_cast<const f16x8*>(s16 + off + a.feat_plane);
            float f[8], pr[9];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = ((float)h[j] + (float)l[j]) * a.feat_mul;
#pragma unroll
            for (int c = 0; c < 9; ++c) pr[c] = sp_prob[p * 9 + c];
#pragma unroll
            for (int c = 0; c < 9; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[c][j] = fmaf(f[j], pr[c], acc[c][j]);
        }
#pragma unroll
        for (int c = 0; c < 9; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[c][j] = butterfly_add_8_16_32(acc[c][j]);
            }
        if ((threadIdx.x & 63) < 8) {          // lanes 0..7 of each wave hold the wave's sums of channel group q
#pragma unroll
            for (int c = 0; c < 9; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[(g * 9 + c) * 64 + q * 8 + j] = acc[c][j];
        }
        __syncthreads();
        for (int o = threadIdx.x; o < 9 * 64; o += 256) {
            const int c = o >> 6, ch = o & 63;
            const float s = (red[(0 * 9 + c) * 64 + ch] + red[(1 * 9 + c) * 64 + ch]) + (red[(2 * 9 + c) * 64 + ch] + red[(3 * 9 + c) * 64 + ch]);
            a.partial[((long)cell * 9 + c) * (C + 1) + ch] = s * inv;
        }
        __syncthreads();
        ch_first = 64;
    }
    for (int ch0 = ch_first; ch0 <= C; ch0 += 64) {
        // A round with few channels left (the ab + ones tail: 3 of 64 lanes would work while the round costs as much as
        // a full one) splits the lanes as (channel slot, pixel subgroup): nslot = 2^k >= channels left, 64/nslot pixel
        // subgroups per wave, combined below by a fixed-order xor butterfly.
        const int left = C + 1 - ch0;
        int nslot = 64;
        while (nslot > 1 && (nslot >> 1) >= left) nslot >>= 1;
        const int nsub = 64 / nslot, chl = lanech & (nslot - 1), sub = lanech / nslot;
        const int ch = ch0 + chl;
        const int pstep = 4 * nsub;                       // pixels p = (g + 4 sub) + pstep * i
        float acc[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) acc[c] = 0.f;
        if (ch <= C) {
            // Per-thread source of its channel, resolved once: pointer to pixel (0,0) of the cell and the element stride
            // between pixels.  (The pass is VALU-issue bound - PMC: 4200 VALU instructions per wave, 65% of them index
            // arithmetic when the 64-bit offsets and the division by sp sat in the inner loop.)
            const long cell0 = (long)(cy * a.sp) * a.W + cx * a.sp;
            const f16* s16 = nullptr; const float* s32 = nullptr; int pstride = 0;
            if (ch == C) {}
            else if (ch < a.c_act) { s16 = a.feat_act + (((long)n * (a.c_act >> 4) + (ch >> 4)) * HW + cell0) * 16 + (ch & 15); pstride = 16; }
            else if (ch < a.c_act + a.c_nchw) { s32 = a.feat_nchw + ((long)n * a.c_nchw + (ch - a.c_act)) * HW + cell0; pstride = 1; }
            else { s32 = a.feat_bc + cell0 * a.c_bc + (ch - a.c_act - a.c_nchw); pstride = a.c_bc; }
            const int rowstride = a.W * pstride;               // elements between cell rows
            const long plane = a.feat_plane;
            // 4 pixels per trip with all loads issued before their use (puts 4x more bytes in flight per wave);
            // full round: pixel p = p0 + 4u, p0 = g, g+16, ...: for sp = 16 that is row p0>>4, columns g, g+4, g+8, g+12
            for (int p0 = g + 4 * sub; p0 < npix; p0 += 4 * pstep) {
                float f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + pstep * u;
                    f[u] = 0.f;
                    if (p < npix) {
                        int py, px;
                        if (a.sp == 16) { py = p >> 4; px = p & 15; } else { py = p / a.sp; px = p - py * a.sp; }
                        const int off = py * rowstride + px * pstride;
                        if (ch == C) f[u] = 1.f;
                        else if (s16) f[u] = ((float)s16[off] + (float)s16[off + plane]) * a.feat_mul;
                        else f[u] = s32[off];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + pstep * u;
                    if (p < npix) {
#pragma unroll
                        for (int c = 0; c < 9; ++c) acc[c] = fmaf(f[u], sp_prob[p * 9 + c], acc[c]);
                    }
                }
            }
        }
        if (nsub > 1) {                                   // wave-uniform: combine the pixel subgroups of each channel slot
#pragma unroll
            for (int c = 0; c < 9; ++c)
                for (int sft = nslot; sft < 64; sft <<= 1) acc[c] += __shfl_xor(acc[c], sft);
        }
        // reduce the 4 pixel groups
        if (sub == 0) {
#pragma unroll
            for (int c = 0; c < 9; ++c) red[(g * 9 + c) * 64 + chl] = acc[c];
        }
        __syncthreads();
        if (g == 0 && sub == 0 && ch <= C) {
            float* dst = a.partial + ((long)cell * 9) * (C + 1) + ch;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const float s = (red[(0 * 9 + c) * 64 + chl] + red[(1 * 9 + c) * 64 + chl]) +
                                (red[(2 * 9 + c) * 64 + chl] + red[(3 * 9 + c) * 64 + chl]);
                dst[(long)c * (C + 1)] = s * inv;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 9) a.cnt[(long)cell * 9 + threadIdx.x] = (float)s_cnt[threadIdx.x];
}

__global__ void pool_gather_kernel(PoolArgs a) {
    const int C = a.c_act + a.c_nchw + a.c_bc, C2 = a.c_act + a.c_nchw;
    const int hs = a.H / a.sp, ws = a.W / a.sp, L = hs * ws;
    const long total = (long)a.n * L * (C + 1);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(t % (C + 1));
        const long sp_i = t / (C + 1);
        const int j = (int)(sp_i % ws), i = (int)((sp_i / ws) % hs);
        const int n = (int)(sp_i / L);
        float num = 0.f, den = 0.f, cn = 0.f;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const int dy = c / 3 - 1, dx = c % 3 - 1;
            const int si = i - dy, sj = j - dx;
            if (si < 0 || si >= hs || sj < 0 || sj >= ws) continue;
            const long cell = ((long)n * hs + si) * ws + sj;
            const float* pp = a.partial + (cell * 9 + c) * (C + 1);
            num = add_rn(num, pp[ch]);
            den = add_rn(den, pp[C]);
            cn += a.cnt[cell * 9 + c];
        }
        const int tok = i * ws + j;
        if (ch == C) {
            if (a.conf) a.conf[(long)n * L + tok] = den;
            if (a.sizes) a.sizes[(long)n * L + tok] = cn / (float)(a.sp * a.sp);
        } else {
            const float v = num / (den + 1e-8f);
            if (a.tok_out && ch < a.c_tok) a.tok_out[((long)n * L + tok) * a.c_tok + ch] = v;
            if (a.nchw_out && ch >= a.c_from && ch < C2) a.nchw_out[((long)n * (C2 - a.c_from) + (ch - a.c_from)) * L + tok] = v;
            if (a.bc_out && ch >= C2) a.bc_out[((long)n * L + tok) * a.c_bc + (ch - C2)] = v;
        }
    }
}
inline int grid_for(long total, int block = 256) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

size_t poolfeat_ws_bytes(int n, int c, int H, int W, int sp) {
    const size_t cells = (size_t)n * (H / sp) * (W / sp);
    return cells * 9 * (c + 1) * sizeof(float) + cells * 9 * sizeof(float);
}

int launch_poolfeat(const PoolArgs& a, hipStream_t s) {
    const int C = a.c_act + a.c_nchw + a.c_bc;
    if (a.H % a.sp || a.W % a.sp) { set_error("poolfeat: %dx%d not a multiple of sp=%d", a.H, a.W, a.sp); return DISCO_ESHAPE; }
    const int cells = a.n * (a.H / a.sp) * (a.W / a.sp);
    const size_t smem = ((size_t)a.sp * a.sp * 9 + 4 * 9 * 64) * sizeof(float);
    hipLaunchKernelGGL(pool_partial_kernel, dim3(cells), dim3(256), smem, s, a);
    DISCO_LAUNCH_CHECK("pool_partial_kernel");
    const long total = (long)cells * (C + 1);
    hipLaunchKernelGGL(pool_gather_kernel, dim3(grid_for(total)), dim3(256), 0, s, a);
    DISCO_LAUNCH_CHECK("pool_gather_kernel");
    return DISCO_OK;
}

}  // namespace disco

// Op-level entry point for the forward's pooling launch (the 16-byte path: 64 act channels with hi + lo planes, plus two fp32 NCHW
// channels): tokens (n, L, 64) and the pooled NCHW channels (n, 2, h, w).  tools/concurrency_probe.py runs it next to other work.
extern "C" int disco_op_poolfeat_act(const void* d_act, const float* d_nchw2, const float* d_prob, float* d_tokens, float* d_pooled2, int n,
                                     int h, int w, void* d_ws, size_t ws_bytes, void* stream) {
    using namespace disco;
    if (!d_act || !d_nchw2 || !d_prob || !d_tokens || !d_pooled2 || !d_ws || n <= 0 || h <= 0 || w <= 0 || h % 16 || w % 16) { set_error("poolfeat_act: bad argument"); return DISCO_EINVAL; }
    if (ws_bytes < poolfeat_ws_bytes(n, 66, h, w, 16)) { set_error("poolfeat_act: workspace too small"); return DISCO_ENOMEM; }
    PoolArgs pa{};
    pa.feat_act = reinterpret_cast<const f16*>(d_act); pa.feat_plane = (long)n * 64 * h * w; pa.c_act = 64; pa.feat_mul = 1.f;
    pa.feat_nchw = d_nchw2; pa.c_nchw = 2; pa.prob = d_prob;
    const size_t cells = (size_t)n * (h / 16) * (w / 16);
    pa.partial = (float*)d_ws; pa.cnt = (float*)d_ws + cells * 9 * 67;
    pa.tok_out = d_tokens; pa.c_tok = 64; pa.nchw_out = d_pooled2; pa.c_from = 64;
    pa.n = n; pa.H = h; pa.W = w; pa.sp = 16;
    return launch_poolfeat(pa, (hipStream_t)stream);
}

