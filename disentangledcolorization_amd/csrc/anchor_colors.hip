// anchor_colors.hip - the anchors' colours: softmax(313) -> stable top-10 -> T-th distinct colour (anchor_gen.py:54-90), nearest gamut bin
// (basic.py:177-194, model.py:166), annealed-mean decode (basic.py:196-218).  Split out of tokens.hip in round 6.
#include <cmath>
#include <cstdlib>
#include <vector>
#include <mutex>
#include "common.h"

namespace disco {

namespace {

// ---- colour selection: one wave per token ---------------------------------------------------------------------
// probabilities exactly as softmax: exp(x-max)/sum; order = (p desc, bin asc) = stable descending sort.
__global__ __launch_bounds__(256) void select_colors_kernel(const float* __restrict__ logit, const float* __restrict__ q_to_ab,
                                                            float* colors, int32_t* labels, int n, int L, int t_first,
                                                            int t_count, int plain_rank) {
    const int lane = threadIdx.x & 63;
    const int tokg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tokg >= n * L) return;
    const int img = tokg / L, t = tokg - img * L;
    const float* lp = logit + (size_t)img * N_VOCAB * L + t;
    float p[5];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int b = lane + 64 * i;
        p[i] = b < N_VOCAB ? lp[(size_t)b * L] : -INFINITY;
        mx = fmaxf(mx, p[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) { p[i] = (lane + 64 * i) < N_VOCAB ? expf(p[i] - mx) : 0.f; s += p[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
#pragma unroll
    for (int i = 0; i < 5; ++i) p[i] = (lane + 64 * i) < N_VOCAB ? p[i] / s : -1.f;
    // top-10 by repeated wave arg-max (value desc, bin asc) - as many rounds as the caller's picks can reach: the most probable bin alone
    // (sampled_T = 0, the default inference: one round instead of ten, 17 -> 7 us for one image), the plain_rank-th, or all ten (T = 1, 2)
    const int rounds = plain_rank >= 0 ? min(plain_rank + 1, 10) : (t_first + t_count > 1 ? 10 : 1);        // (uniform)
    int top[10];
    int last = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r >= rounds) break;
        float bv = -2.f; int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 5; ++i) if (p[i] > bv) { bv = p[i]; bi = lane + 64 * i; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        top[r] = bi;
        last = bi;
#pragma unroll
        for (int i = 0; i < 5; ++i) if (lane + 64 * i == bi) p[i] = -3.f;
    }
    if (lane != 0) return;
    if (rounds < 10) {
        // one pick, the last bin found: the same values the general path below writes for it
        const float a1 = q_to_ab[last * 2] / 110.0f, b1c = q_to_ab[last * 2 + 1] / 110.0f;
        for (int tt = 0; tt < t_count; ++tt) {
            const size_t oi = (size_t)img * t_count + tt;
            colors[(oi * 2 + 0) * L + t] = a1;
            colors[(oi * 2 + 1) * L + t] = b1c;
            if (labels) labels[oi * L + t] = last;
        }
        return;
    }
    float ca[10], cb[10];
#pragma unroll
    for (int r = 0; r < 10; ++r) { ca[r] = q_to_ab[top[r] * 2] / 110.0f; cb[r] = q_to_ab[top[r] * 2 + 1] / 110.0f; }
    // T=1: first candidate farthest from top-1; T=2: first candidate maximising d1 + dist to the T=1 pick
    float d1[10]; int j1 = 0; float b1 = -1.f;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const float da = sub_rn(ca[r], ca[0]), db = sub_rn(cb[r], cb[0]);
        d1[r] = sqrtf(add_rn(mul_rn(da, da), mul_rn(db, db)));
        if (d1[r] > b1) { b1 = d1[r]; j1 = r; }
    }
    int j2 = 0; float b2 = -1.f;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const float da = sub_rn(ca[r], ca[j1]), db = sub_rn(cb[r], cb[j1]);
        const float d2 = add_rn(d1[r], sqrtf(add_rn(mul_rn(da, da), mul_rn(db, db))));
        if (d2 > b2) { b2 = d2; j2 = r; }
    }
    const int pick[3] = {0, j1, j2};
    for (int tt = 0; tt < t_count; ++tt) {
        // plain_rank >= 0: the plain_rank-th most probable bin (ColorLabel.decode_ind2ab, basic.py:196-209)
        const int r = plain_rank >= 0 ? plain_rank : pick[t_first + tt];
        // output image index: image-major [img][tt]
        const size_t oi = (size_t)img * t_count + tt;
        colors[(oi * 2 + 0) * L + t] = ca[r];
        colors[(oi * 2 + 1) * L + t] = cb[r];
        if (labels) labels[oi * L + t] = top[r];   // bin centres are their own nearest bin
    }
}

__global__ void nearest_bin_kernel(const float* __restrict__ ab, const float* __restrict__ q_to_ab, int32_t* labels,
                                   int n, int L) {
    const long total = (long)n * L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long img = i / L, t = i % L;
        const float a = mul_rn(ab[(img * 2 + 0) * L + t], 110.f), b = mul_rn(ab[(img * 2 + 1) * L + t], 110.f);
        float best = INFINITY; int bi = 0;
        for (int q = 0; q < N_VOCAB; ++q) {
            const float da = sub_rn(q_to_ab[q * 2], a), db = sub_rn(q_to_ab[q * 2 + 1], b);
            const float d = add_rn(mul_rn(da, da), mul_rn(db, db));
            if (d < best) { best = d; bi = q; }
        }
        labels[i] = bi;
    }
}

// ColorLabel.decode_ind2ab for non-integer T (basic.py:210-217): p = softmax(logit); e = exp(p / T); ab = sum_q e_q ab_q
// / sum_q e_q / 110.  One wave per token, lanes stride over the 313 bins, fixed-order butterfly reductions.
__global__ __launch_bounds__(256) void decode_annealed_kernel(const float* __restrict__ logit, const float* __restrict__ q_to_ab,
                                                              float* __restrict__ ab, int n, int L, float T) {
    const int lane = threadIdx.x & 63;
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= (long)n * L) return;
    const long img = tok / L, t = tok - img * L;
    const float* lg = logit + img * N_VOCAB * L + t;
    float v[5];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 5; ++i) { const int q = lane + 64 * i; v[i] = q < N_VOCAB ? lg[(long)q * L] : -INFINITY; mx = fmaxf(mx, v[i]); }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) { v[i] = lane + 64 * i < N_VOCAB ? expf(v[i] - mx) : 0.f; sm += v[i]; }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) sm += __shfl_xor(sm, s);
    float se = 0.f, sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int q = lane + 64 * i;
        if (q < N_VOCAB) {
            const float e = expf(v[i] / sm / T);
            se += e; sa = fmaf(e, q_to_ab[2 * q], sa); sb = fmaf(e, q_to_ab[2 * q + 1], sb);
        }
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { se += __shfl_xor(se, s); sa += __shfl_xor(sa, s); sb += __shfl_xor(sb, s); }
    if (lane == 0) {
        ab[(img * 2) * L + t] = sa / se / 110.f;
        ab[(img * 2 + 1) * L + t] = sb / se / 110.f;
    }
}

}  // namespace

int launch_decode_annealed(const float* logit_nchw, const float* q_to_ab, float* ab, int n, int l, float T, hipStream_t s) {
    if (!(T > 0.f)) { set_error("decode_ind2ab: temperature %g", (double)T); return DISCO_EINVAL; }
    hipLaunchKernelGGL(decode_annealed_kernel, dim3(cdiv(n * l, 4)), dim3(256), 0, s, logit_nchw, q_to_ab, ab, n, l, T);
    DISCO_LAUNCH_CHECK("decode_annealed_kernel");
    return DISCO_OK;
}

int launch_select_colors(const float* logit_nchw, const float* q_to_ab, float* colors, int32_t* labels, int n, int l,
                         int t_first, int t_count, hipStream_t s, int plain_rank) {
    if (t_first < 0 || t_first + t_count > 3 || plain_rank > 9) { set_error("select_colors: T range"); return DISCO_EINVAL; }
    hipLaunchKernelGGL(select_colors_kernel, dim3(cdiv(n * l, 4)), dim3(256), 0, s, logit_nchw, q_to_ab, colors, labels,
                       n, l, t_first, t_count, plain_rank);
    DISCO_LAUNCH_CHECK("select_colors_kernel");
    return DISCO_OK;
}

int launch_nearest_bin(const float* ab_nchw, const float* q_to_ab, int32_t* labels, int n, int l, hipStream_t s) {
    hipLaunchKernelGGL(nearest_bin_kernel, dim3(cdiv(n * l, 256)), dim3(256), 0, s, ab_nchw, q_to_ab, labels, n, l);
    DISCO_LAUNCH_CHECK("nearest_bin_kernel");
    return DISCO_OK;
}

}  // namespace disco
