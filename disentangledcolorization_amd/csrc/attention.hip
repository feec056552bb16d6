// attention.hip - softmax(Q K^T) V of the encoder stacks on packed fp32 FMAs (the VALU form: fewer than 1 024 tokens;
// attention_mfma.hip takes over from there).  transformer2d.py:53-54 (nn.MultiheadAttention, 8 heads, d_head = 8).
// Split out of tokens.hip in round 6 (same kernels, same machine code: tools/kernel_isa_hash.py).
#include <cmath>
#include <cstdlib>
#include <vector>
#include <mutex>
#include "common.h"

namespace disco {

namespace {

// ---- attention: softmax(Q K^T) V per (image, head), d_head = 8 --------------------------------------------------
// Block = 64 queries of one (image, head): 16 query groups x 16 key partitions.  A thread owns QT = 4 queries and
// every 16th key, so each K/V fragment it reads from LDS serves 4 queries (one query per thread made the kernel
// LDS-issue bound: 75 us per call; this layout reads 16x less per FLOP and fills the chip with 8192 waves).
// Keys/values are streamed through LDS in chunks of KCH; a thread keeps the 16 x 4 scores of its keys in registers
// (computed once), chunks combine by online softmax, the 16 partitions of a query merge with shuffles.  The dot
// products and the P V accumulation run as packed fp32 FMAs (v_pk_fma_f32); exponentials are v_exp_f32 (__expf:
// relative error ~1e-6 on arguments <= 0, far inside the 2e-5 encoder tolerance).
constexpr int KCH = 256;
constexpr int KP = 16;     // key partitions (lanes) per query group
// QT: queries per thread (a block covers (256 / KP) QT of them).  4 for throughput; 1 when the grid would not fill the GPU (one image of
// 256 tokens: 32 workgroups at QT = 4) - a query's arithmetic does not depend on how many neighbours share its thread: same results
// MASK: `use_mask` (model.py:121-125 -> transformer2d.py:53-54): key j of image i gets +1.0 on every score when superpixel j holds fewer than
// 25 pixels (key_sizes[i / key_rep][j] < key_thr = 25 / cell area) - the reference's FLOAT key_padding_mask, additive under torch >= 1.9: score = mask + q k^T
// (one rounding, like baddbmm).  Without MASK the kernel is the machine code it was.
template <int QT, bool MASK>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* out, int L,
                                                        const float* __restrict__ key_sizes, int key_rep, float key_thr) {
    // halves of a key / value in separate arrays: the 16 partitions of a wave read 16 consecutive float4 (256 contiguous
    // bytes, no bank conflict; interleaved [key][2] rows put partitions p and p+8 on the same banks)
    __shared__ float4 sk[2][KCH];
    __shared__ float4 sv[2][KCH];
    __shared__ float sbias[MASK ? KCH : 1];
    const int qb = blockIdx.x, head = blockIdx.y, img = blockIdx.z;
    constexpr int QPB = (256 / KP) * QT;
    const int part = threadIdx.x & (KP - 1);
    const int q0i = qb * QPB + (threadIdx.x / KP) * QT;
    const size_t base = (size_t)img * L * 64 + head * 8;
    f32x2 qv[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = min(q0i + t, L - 1);        // clamp: the extra lanes compute a duplicate that is not stored
        const float4 a = *reinterpret_cast<const float4*>(q + base + (size_t)qi * 64);
        const float4 b = *reinterpret_cast<const float4*>(q + base + (size_t)qi * 64 + 4);
        qv[t][0] = f32x2{a.x, a.y}; qv[t][1] = f32x2{a.z, a.w}; qv[t][2] = f32x2{b.x, b.y}; qv[t][3] = f32x2{b.z, b.w};
    }
    float m[QT], l[QT];
    f32x2 o[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -INFINITY; l[t] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[t][j] = f32x2{0.f, 0.f};
    }
    for (int c0 = 0; c0 < L; c0 += KCH) {
        const int nk = min(KCH, L - c0);
        __syncthreads();
        for (int u = threadIdx.x; u < nk * 2; u += 256) {
            const int key = u >> 1, half = u & 1;
            sk[half][key] = *reinterpret_cast<const float4*>(k + base + (size_t)(c0 + key) * 64 + half * 4);
            sv[half][key] = *reinterpret_cast<const float4*>(v + base + (size_t)(c0 + key) * 64 + half * 4);
            if constexpr (MASK) { if (half == 0) sbias[key] = key_sizes[(size_t)(img / key_rep) * L + c0 + key] < key_thr ? 1.f : 0.f; }
        }
        __syncthreads();
        float sc[KCH / KP][QT];
        float cm[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) cm[t] = -INFINITY;
#pragma unroll
        for (int i = 0; i < KCH / KP; ++i) {
            const int j = part + KP * i;
            if (j < nk) {
                const float4 a = sk[0][j], b = sk[1][j];
                const f32x2 k0{a.x, a.y}, k1{a.z, a.w}, k2{b.x, b.y}, k3{b.z, b.w};
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    f32x2 d = qv[t][0] * k0;
                    d = __builtin_elementwise_fma(qv[t][1], k1, d);
                    d = __builtin_elementwise_fma(qv[t][2], k2, d);
                    d = __builtin_elementwise_fma(qv[t][3], k3, d);
                    sc[i][t] = add_rn(d.x, d.y);
                    if constexpr (MASK) sc[i][t] = add_rn(sbias[j], sc[i][t]);
                    cm[t] = fmaxf(cm[t], sc[i][t]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < QT; ++t) sc[i][t] = -INFINITY;
            }
        }
        if (part >= nk) continue;                   // fewer than KP keys in the chunk: nothing for this lane
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const float mn = fmaxf(m[t], cm[t]);
            const float alpha = __expf(m[t] - mn);    // 0 on the lane's first chunk (m = -inf)
            l[t] = mul_rn(l[t], alpha);            // (every rounding step spelled out: QT = 1 and QT = 4 must be the same arithmetic)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[t][j] = mul_rn2(o[t][j], pair_of(alpha));
            m[t] = mn;
        }
#pragma unroll
        for (int i = 0; i < KCH / KP; ++i) {
            const int j = part + KP * i;
            if (j < nk) {
                const float4 c = sv[0][j], d = sv[1][j];
                const f32x2 v0{c.x, c.y}, v1{c.z, c.w}, v2{d.x, d.y}, v3{d.z, d.w};
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const float p = __expf(sc[i][t] - m[t]);
                    l[t] = add_rn(l[t], p);
                    const f32x2 pp{p, p};
                    o[t][0] = __builtin_elementwise_fma(pp, v0, o[t][0]);
                    o[t][1] = __builtin_elementwise_fma(pp, v1, o[t][1]);
                    o[t][2] = __builtin_elementwise_fma(pp, v2, o[t][2]);
                    o[t][3] = __builtin_elementwise_fma(pp, v3, o[t][3]);
                }
            }
        }
    }
    // merge the KP key partitions of each query (lanes KP*g .. KP*g + KP-1), then lane `part` < 4 stores pair `part`
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float mm = m[t];
#pragma unroll
        for (int sft = 1; sft < KP; sft <<= 1) mm = fmaxf(mm, __shfl_xor(mm, sft));
        const float scl = m[t] == -INFINITY ? 0.f : __expf(m[t] - mm);
        float ls = mul_rn(l[t], scl);
#pragma unroll
        for (int sft = 1; sft < KP; sft <<= 1) ls = add_rn(ls, __shfl_xor(ls, sft));
        const float inv = 1.f / ls;
        f32x2 mine{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = mul_rn(o[t][j].x, scl), y = mul_rn(o[t][j].y, scl);
#pragma unroll
            for (int sft = 1; sft < KP; sft <<= 1) { x = add_rn(x, __shfl_xor(x, sft)); y = add_rn(y, __shfl_xor(y, sft)); }
            if (part == j) mine = f32x2{mul_rn(x, inv), mul_rn(y, inv)};
        }
        const int qi = q0i + t;
        if (qi < L && part < 4) *reinterpret_cast<f32x2*>(out + base + (size_t)qi * 64 + 2 * part) = mine;
    }
}


}  // namespace

// The form is chosen by the grid the launch would have: QT = 1 when QT = 4 could not give every CU a workgroup.  A query's arithmetic does not
// depend on how many neighbours share its thread (every rounding step of the softmax is spelled out: mul_rn / add_rn / fmaf), so an
// image's result does not depend on the batch it is part of (tests/test_gpu_ops.py::test_encoder_stack_result_does_not_depend_on_the_batch).
int launch_attention_valu(const float* q, const float* k, const float* v, float* out, int n, int l, hipStream_t s, const float* key_sizes, int key_rep, float key_thr) {
    // (beyond one workgroup per CU the two forms run the same: n = 2 ... 16 images measured with the threshold at 1x, 2x, 5x, 9x the CU count)
    const bool small = (long)cdiv(l, 64) * N_HEAD * n < num_cus_current();
    const dim3 grid(cdiv(l, small ? 16 : 64), N_HEAD, n);
    if (key_sizes) {
        if (key_rep < 1) { set_error("attention: key_rep %d", key_rep); return DISCO_EINVAL; }
        if (small) hipLaunchKernelGGL((attention_kernel<1, true>), grid, dim3(256), 0, s, q, k, v, out, l, key_sizes, key_rep, key_thr);
        else hipLaunchKernelGGL((attention_kernel<4, true>), grid, dim3(256), 0, s, q, k, v, out, l, key_sizes, key_rep, key_thr);
    } else if (small)
        hipLaunchKernelGGL((attention_kernel<1, false>), grid, dim3(256), 0, s, q, k, v, out, l, key_sizes, 1, key_thr);
    else
        hipLaunchKernelGGL((attention_kernel<4, false>), grid, dim3(256), 0, s, q, k, v, out, l, key_sizes, 1, key_thr);
    DISCO_LAUNCH_CHECK("attention_kernel");
    return DISCO_OK;
}

}  // namespace disco
