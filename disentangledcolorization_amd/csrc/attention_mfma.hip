// attention_mfma.hip - softmax(Q K^T) V of long token sequences on the fp32 matrix cores (transformer2d.py:52-60 through nn.MultiheadAttention,
// d_head = 8).  Its own translation unit: compiled with -mllvm -amdgpu-mfma-vgpr-form (build.py), which keeps the MFMA results in VGPRs - the
// softmax reads every score, and out of AGPRs each read is a v_accvgpr_read_b32 (one more VALU issue slot per score: measured 1 065 -> 943 us
// per layer at 16 384 tokens on the first version of the kernel, profiles/r05_attn_mfma_ab.txt).
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace disco {

namespace {

// ---- attention on the matrix cores: long token sequences (--no_resize sizes; transformer2d.py:52-60 through nn.MultiheadAttention) ----
// attention_kernel (tokens.hip: fewer than 1 024 tokens) spends ~17 VALU issue slots per score (8 packed FMAs for the two contractions, max, exp, sums) and is VALU-bound;
// at 16 384 tokens (2048 x 2048 input) the two stacks are a third of the forward.  Here both contractions run on fp32 MFMAs and the VALU keeps
// max / subtract / exp / row sum (~7 slots per score):
//   * S^T = K Q^T per 32-key x 32-query tile as four v_mfma_f32_32x32x2_f32 (d_head = 8 = 4 x K 2): rows = keys, columns = queries, so a lane
//     (column = lane % 32, half = lane / 32) holds ONE query's scores against the 16 keys 8 (r / 4) + 4 half + r % 4, r = its register index.
//   * P V as v_mfma_f32_4x4x1_16B_f32: 16 independent 4 x 4 x 1 products per instruction, block b = lanes 4 b .. 4 b + 3.  With A = P (a lane's
//     own probability of key r: the S^T accumulator register IN PLACE, no movement between the accumulator and the operand layout) and B = V of
//     that key (dims 4 h + lane % 4), block b accumulates O[queries 4 (b % 8) .. + 3][dims 4 h .. + 3] over the keys of half b / 8: every one of the
//     256 multiply-adds of an instruction is a useful one, where a 16 x 16 x 4 or 32 x 32 x 2 tile would carry d_head = 8 in 16 / 32 output columns
//     (tools/mfma_4x4x1_probe.hip pins the layout and this data flow; profiles/r05_mfma_4x4x1_probe.txt).  32 instructions of 8 cycles per tile
//     next to the 4 x 64 of S^T: 0.375 matrix-pipe cycles per score and SIMD.
//   * K chunks sit in LDS as [half][key] float4 (dims {half, 2 + half, 4 + half, 6 + half}: one conflict-free ds_read_b128 per tile), V chunks
//     TRANSPOSED, [dim][key] with rows 8 floats apart in bank phase: a lane's V operands of 4 consecutive keys are one ds_read_b128 (8 per tile,
//     shared by the wave's QW query tiles).  The next chunk's global loads are in flight under the current chunk's tiles.
//   * online softmax per (query, key half): running maximum in the log2 domain (q carries log2 e, p = v_exp_f32(s - m)); the accumulators are
//     rescaled only when some lane's maximum moved (wave-uniform branch; the factors reach the 4 x 4 blocks through DPP quad broadcasts); the two
//     key halves of a query merge at the end with one cross-half shuffle.
// Same mathematics as attention_kernel in another summation order: results agree to fp32 rounding (~1e-7 relative), not bit for bit, so the
// choice between the two depends on the token count ALONE (never on the batch size): an image's result does not depend on its batch.
#ifndef AM_KPT
#define AM_KPT 2            // keys per thread and staged chunk (a chunk = 128 NW keys)
#endif
#ifndef AM_ABL
#define AM_ABL 0            // diagnostic builds (timing only, results wrong): bit 0 = no P V MFMAs, bit 1 = no softmax arithmetic, bit 2 = no K Q^T MFMAs
#endif

template <int SEL>
__device__ __forceinline__ float quad_bcast(float x) {     // lane (l & ~3) + SEL's value, in every lane of the quad
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), SEL * 0x55, 0xf, 0xf, false));
}

// QW: 32-query tiles per wave (the launcher takes 1: the QW = 2 form shares the K / V fragments of a key tile between 64 queries but needs 216
//     VGPRs = two waves per SIMD, and measured 5-25 % slower than four waves of one tile each)
// NW: waves per workgroup = how many share a staged chunk of 128 NW keys (the launcher takes 4)
//
// Schedule.  MFMA and VALU instructions of a SIMD share one issue port, and waves running the same code ask for the same pipe at the same
// time: the first version (tile after tile) ran at nearly the SUM of its three parts (profiles/r05_attn_mfma_ab.txt, the AM_ABL builds).
// The instruction stream is software-pipelined - step t issues the four K Q^T MFMAs of tile t + 1 (64 cycles each, one issue slot) with tile
// t's subtract / exp / add between them, then tile t's 32 P V MFMAs (8 cycles each) and the row maxima of tile t + 1; two named score sets
// alternate (static indexing); a tile's mask (the sequence's last, partial tile) and the running-maximum update with its rare rescale run
// between the steps - which by itself measured +-0 against the plain order (939 vs 943 us per layer at 16 384 tokens); what pays is
// occupancy: 126 VGPRs, four waves per SIMD (851 us), which this form keeps because a tile's scores die into the P V operands in place.
// KS: the waves of a workgroup split the KEYS of 32 QW queries (wave u takes the key tiles u, u + NW, ... of every chunk) and merge their partial
// (maximum, sum, O) through LDS at the end: NW times the waves per query where the plain form cannot fill the GPU (one image)
template <int QW, int NW, bool KS, bool MASK>
__global__ __launch_bounds__(64 * NW) void attention_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, float* out, int L,
                                                             const float* __restrict__ key_sizes, int key_rep, float key_thr) {
    constexpr int NTHR = 64 * NW, KPT = AM_KPT, KCH = NTHR * KPT;  // keys per thread and chunk; keys per staged chunk
    constexpr int VTS = KCH + 8;                              // floats per row of the transposed V chunk (rows 32 bytes apart in bank phase)
    __shared__ float4 sK[2][KCH];
    __shared__ __attribute__((aligned(16))) float sVT[8][VTS];
    // MASK (`use_mask`, see attention.hip): the keys' additive biases of the staged chunk, log2 domain - they START the S^T accumulator of a
    // tile (row = key), so the matrix pipe adds them for free; same mathematics as attention_kernel's mask + q k^T in another rounding order
    __shared__ float sBias[MASK ? KCH : 1];
    const int head = blockIdx.y, img = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    // (a scalar: the tile addresses of the key-split form stay in SGPRs - 166 -> 126 VGPRs, four waves per SIMD there as well: eight images of
    // 1 536 tokens 111 -> 105 us per layer, sixteen of 1 024: 105 -> 97)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qn = lane & 31, hk = lane >> 5, j4 = lane & 3;
    const size_t base = (size_t)img * L * 64 + head * 8;
    const int q0 = (KS ? blockIdx.x : blockIdx.x * NW + wave) * (32 * QW);
    constexpr int TS = KS ? 32 * NW : 32;                     // distance of a wave's successive key tiles
    const int tfirst = KS ? 32 * wave : 0;
    const bool active = q0 < L;                               // (wave-uniform; an idle wave still stages its share of every chunk)
    constexpr float LOG2E = 1.4426950408889634f;
    float qb[QW][4];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        const int qi = min(q0 + 32 * w + qn, L - 1);          // clamp: the extra lanes compute a duplicate that is not stored
        const float4 a = *reinterpret_cast<const float4*>(q + base + (size_t)qi * 64);
        const float4 b = *reinterpret_cast<const float4*>(q + base + (size_t)qi * 64 + 4);
        qb[w][0] = (hk ? a.y : a.x) * LOG2E; qb[w][1] = (hk ? a.w : a.z) * LOG2E;
        qb[w][2] = (hk ? b.y : b.x) * LOG2E; qb[w][3] = (hk ? b.w : b.z) * LOG2E;
    }
    float m[QW], mu[QW], l[QW];                               // running maximum (log2 domain), the finite value the exponentials subtract, row sum
    f32x4 o[QW][2];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        m[w] = -INFINITY; mu[w] = 0.f; l[w] = 0.f;
        o[w][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[w][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the chunk in flight: thread t carries keys c0 + t + i NTHR (zeros beyond L: a masked key's p = 0 must not meet a NaN)
    float4 pk0[KPT], pk1[KPT], pv0[KPT], pv1[KPT];
    float pbias[KPT];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int key = c0 + tid + i * NTHR;
            const float4 z = {0.f, 0.f, 0.f, 0.f};
            pk0[i] = pk1[i] = pv0[i] = pv1[i] = z;
            pbias[i] = 0.f;
            if (key < L) {
                if constexpr (MASK) pbias[i] = key_sizes[(size_t)(img / key_rep) * L + key] < key_thr ? LOG2E : 0.f;
                pk0[i] = *reinterpret_cast<const float4*>(k + base + (size_t)key * 64);
                pk1[i] = *reinterpret_cast<const float4*>(k + base + (size_t)key * 64 + 4);
                pv0[i] = *reinterpret_cast<const float4*>(v + base + (size_t)key * 64);
                pv1[i] = *reinterpret_cast<const float4*>(v + base + (size_t)key * 64 + 4);
            }
        }
    };
    int nk = 0;                                               // keys of the current chunk
    // K Q^T of the tile at t0 for query tile w (alone: a chunk's first tile)
    auto qk_tile = [&](f32x16& s, int w, int t0) __attribute__((always_inline)) {
        const float4 kf = sK[hk][t0 + qn];
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = MASK ? sBias[t0 + 8 * (e >> 2) + 4 * hk + (e & 3)] : 0.f;
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qb[w][0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qb[w][1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qb[w][2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qb[w][3], s, 0, 0, 0);
    };
    auto tile_max = [&](const f32x16& s) __attribute__((always_inline)) -> float {
        float tm = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) tm = fmaxf(fmaxf(tm, s[r]), s[r + 1]);
        return fmaxf(tm, s[15]);
    };
    // between the steps: mask of the sequence's last, partial tile; running maximum; the rare rescale (alpha = 1 in the lanes whose maximum
    // stayed; a half that has seen no key yet keeps m = -inf, mu = 0, and sums of 0)
    auto fixup = [&](f32x16& s, float tm, int w, int t0) __attribute__((always_inline)) {
        if (t0 + 32 > nk) {                                   // (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t0 + 8 * (r >> 2) + 4 * hk + (r & 3) >= nk) s[r] = -INFINITY;
            tm = tile_max(s);
        }
        const float mn = fmaxf(m[w], tm);
        if (__ballot(mn > m[w]) != 0ull) {
            const float alpha = __builtin_amdgcn_exp2f(m[w] - (mn == -INFINITY ? 0.f : mn));
            l[w] *= alpha;
            const float a0 = quad_bcast<0>(alpha), a1 = quad_bcast<1>(alpha), a2 = quad_bcast<2>(alpha), a3 = quad_bcast<3>(alpha);
#pragma unroll
            for (int h = 0; h < 2; ++h) { o[w][h][0] *= a0; o[w][h][1] *= a1; o[w][h][2] *= a2; o[w][h][3] *= a3; }
            m[w] = mn;
            mu[w] = mn == -INFINITY ? 0.f : mn;
        }
    };
    // one step: tile t0 (scores in `cur`, fixed up) through softmax and P V, with tile t0 + 32's K Q^T into `nxt` underneath
    auto step = [&](auto next_tag, f32x16 (&cur)[QW], f32x16 (&nxt)[QW], int t0) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(next_tag)::value;
        float4 kfn = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_NEXT) kfn = sK[hk][t0 + TS + qn];
        float4 vf[4][2];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int h = 0; h < 2; ++h) vf[q4][h] = *reinterpret_cast<const float4*>(&sVT[4 * h + j4][t0 + 8 * q4 + 4 * hk]);
        float tmn[QW];
#pragma unroll
        for (int w = 0; w < QW; ++w) {
            f32x16& s = cur[w];
            float ls = 0.f;
            const float kq[4] = {kfn.x, kfn.y, kfn.z, kfn.w};
            if constexpr (HAS_NEXT) {
#pragma unroll
                for (int e = 0; e < 16; ++e) nxt[w][e] = MASK ? sBias[t0 + TS + 8 * (e >> 2) + 4 * hk + (e & 3)] : 0.f;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (HAS_NEXT) nxt[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(kq[g], qb[w][g], nxt[w], 0, 0, 0);
#if !(AM_ABL & 2)
#pragma unroll
                for (int r = 4 * g; r < 4 * g + 4; ++r) {
                    s[r] = __builtin_amdgcn_exp2f(s[r] - mu[w]);
                    ls += s[r];
                }
#endif
            }
            l[w] += ls;
        }
#pragma unroll
        for (int w = 0; w < QW; ++w) {
            f32x16& s = cur[w];
#if !(AM_ABL & 1)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = e == 0 ? vf[q4][0].x : (e == 1 ? vf[q4][0].y : (e == 2 ? vf[q4][0].z : vf[q4][0].w));
                    const float v1 = e == 0 ? vf[q4][1].x : (e == 1 ? vf[q4][1].y : (e == 2 ? vf[q4][1].z : vf[q4][1].w));
                    o[w][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(s[4 * q4 + e], v0, o[w][0], 0, 0, 0);
                    o[w][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(s[4 * q4 + e], v1, o[w][1], 0, 0, 0);
                }
            }
#else
            o[w][0][0] += s[3] + s[7] + vf[0][0].x; o[w][1][1] += s[5] + s[12] + vf[3][1].w;
#endif
            if constexpr (HAS_NEXT) tmn[w] = tile_max(nxt[w]);
        }
        if constexpr (HAS_NEXT) {
#pragma unroll
            for (int w = 0; w < QW; ++w) fixup(nxt[w], tmn[w], w, t0 + TS);
        }
    };
    using Yes = std::true_type; using No = std::false_type;
    f32x16 sa[QW], sb[QW];
    fetch(0);
    for (int c0 = 0; c0 < L; c0 += KCH) {
        __syncthreads();                                      // every wave has read the previous chunk
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int kk = tid + i * NTHR;
            sK[0][kk] = float4{pk0[i].x, pk0[i].z, pk1[i].x, pk1[i].z};
            sK[1][kk] = float4{pk0[i].y, pk0[i].w, pk1[i].y, pk1[i].w};
            sVT[0][kk] = pv0[i].x; sVT[1][kk] = pv0[i].y; sVT[2][kk] = pv0[i].z; sVT[3][kk] = pv0[i].w;
            sVT[4][kk] = pv1[i].x; sVT[5][kk] = pv1[i].y; sVT[6][kk] = pv1[i].z; sVT[7][kk] = pv1[i].w;
            if constexpr (MASK) sBias[kk] = pbias[i];
        }
        __syncthreads();
        if (c0 + KCH < L) fetch(c0 + KCH);
        if (!active) continue;
        nk = min(KCH, L - c0);
        if (nk <= tfirst) continue;                           // (KS: none of the chunk's tiles is this wave's)
        const int ntile = (nk - tfirst + TS - 1) / TS;        // this wave's tiles of the chunk: at tfirst + i TS
        // the first of them on its own, then the steps in pairs (A -> B, B -> A)
#pragma unroll
        for (int w = 0; w < QW; ++w) { qk_tile(sa[w], w, tfirst); fixup(sa[w], tile_max(sa[w]), w, tfirst); }
        int t = 0;
        for (; t + 2 < ntile; t += 2) { step(Yes{}, sa, sb, tfirst + TS * t); step(Yes{}, sb, sa, tfirst + TS * (t + 1)); }
        if (t + 2 == ntile) { step(Yes{}, sa, sb, tfirst + TS * t); step(No{}, sb, sa, tfirst + TS * (t + 1)); }
        else step(No{}, sa, sb, tfirst + TS * t);
    }
    if (!active) return;
    if constexpr (KS) {
        // the waves' partial results (same lane layout in every wave) through LDS; wave 0 merges them in wave order and finishes
        float* xch = reinterpret_cast<float*>(&sK[0][0]);      // (the chunk buffers are free: barrier first)
        static_assert(sizeof(float4) * 2 * KCH >= sizeof(float) * (NW - 1) * QW * 10 * 64, "exchange area");
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int w = 0; w < QW; ++w) {
                float* d = xch + ((size_t)((wave - 1) * QW + w) * 10) * 64 + lane;
                d[0] = m[w]; d[64] = l[w];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[(2 + 4 * h + i) * 64] = o[w][h][i];
            }
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < QW; ++w)
#pragma unroll 1
            for (int u = 1; u < NW; ++u) {
                const float* d = xch + ((size_t)((u - 1) * QW + w) * 10) * 64 + lane;
                const float mo = d[0], lo = d[64];
                const float mt = fmaxf(m[w], mo);
                const float mz = mt == -INFINITY ? 0.f : mt;
                const float a = __builtin_amdgcn_exp2f(m[w] - mz), b = __builtin_amdgcn_exp2f(mo - mz);      // (-inf - finite: 0; -inf - 0: 0)
                l[w] = l[w] * a + lo * b;
                m[w] = mt;
                const float ai[4] = {quad_bcast<0>(a), quad_bcast<1>(a), quad_bcast<2>(a), quad_bcast<3>(a)};
                const float bi[4] = {quad_bcast<0>(b), quad_bcast<1>(b), quad_bcast<2>(b), quad_bcast<3>(b)};
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[w][h][i] = o[w][h][i] * ai[i] + d[(2 + 4 * h + i) * 64] * bi[i];
            }
    }
    // merge the two key halves of every query (lanes l and l ^ 32), normalise, store: lane (quad g = (lane / 4) % 8, j4, half) holds
    // O[q0 + 32 w + 4 g + i][4 h + j4], i = register, and writes the dims of h = its half
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        const float mt = fmaxf(m[w], __shfl_xor(m[w], 32));
        const float a = m[w] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m[w] - mt);
        float lt = l[w] * a;
        lt += __shfl_xor(lt, 32);
        const float f = a / lt;
        const float f0 = quad_bcast<0>(f), f1 = quad_bcast<1>(f), f2 = quad_bcast<2>(f), f3 = quad_bcast<3>(f);
        const float fi[4] = {f0, f1, f2, f3};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x0 = o[w][0][i] * fi[i], x1 = o[w][1][i] * fi[i];
            x0 += __shfl_xor(x0, 32); x1 += __shfl_xor(x1, 32);
            const int qi = q0 + 32 * w + 4 * ((lane >> 2) & 7) + i;
            if (qi < L) out[base + (size_t)qi * 64 + 4 * hk + j4] = hk ? x1 : x0;
        }
    }
}

}  // namespace

// softmax(q k^T) v per (image, head); q, k, v, out: (n, l, 64) fp32, head h in columns 8 h .. 8 h + 7; q pre-scaled by 1 / sqrt(8)
int launch_attention_mfma(const float* q, const float* k, const float* v, float* out, int n, int l, hipStream_t s, const float* key_sizes, int key_rep, float key_thr) {
    // One 32-query tile per wave (126 VGPRs: four waves per SIMD; the QW = 2 form - 216 VGPRs, two waves per SIMD - measured 5-25 % slower at every
    // size, profiles/r05_attn_mfma_ab.txt "forms").  Two forms: (2) workgroups of four query tiles sharing the staged chunks; (3) the four waves of
    // a workgroup split the KEYS of one query tile (four times the waves per query: what one image of up to 2 048 tokens needs to fill the GPU).
    // The forms differ in their summation order (fp32 rounding), so the choice depends on the TOKEN COUNT ALONE - never on the batch size: an
    // image's result does not depend on the batch it is part of (test_batch_of_64_at_512_crosses_the_addressing_limit caught a grid-size rule).
    // Up to 2 048 tokens form 3 (one image: 24 / 33 us per layer at 1 024 / 1 536 tokens against 33 / 42 in form 2; eight images at 1 536: 113
    // against 97 - still under attention_kernel's 138), beyond it form 2.  DISCO_ATTN_FORM = 2 / 3 forces a form (measurements only).
    static const int forced = [] { const char* e = std::getenv("DISCO_ATTN_FORM"); return e ? atoi(e) : 0; }();
    const int form = forced ? forced : (l > 2048 ? 2 : 3);
    if (key_sizes && key_rep < 1) { set_error("attention: key_rep %d", key_rep); return DISCO_EINVAL; }
    if (form == 2) {
        if (key_sizes) hipLaunchKernelGGL((attention_mfma_kernel<1, 4, false, true>), dim3(cdiv(l, 128), N_HEAD, n), dim3(256), 0, s, q, k, v, out, l, key_sizes, key_rep, key_thr);
        else hipLaunchKernelGGL((attention_mfma_kernel<1, 4, false, false>), dim3(cdiv(l, 128), N_HEAD, n), dim3(256), 0, s, q, k, v, out, l, key_sizes, 1, key_thr);
    } else {
        if (key_sizes) hipLaunchKernelGGL((attention_mfma_kernel<1, 4, true, true>), dim3(cdiv(l, 32), N_HEAD, n), dim3(256), 0, s, q, k, v, out, l, key_sizes, key_rep, key_thr);
        else hipLaunchKernelGGL((attention_mfma_kernel<1, 4, true, false>), dim3(cdiv(l, 32), N_HEAD, n), dim3(256), 0, s, q, k, v, out, l, key_sizes, 1, key_thr);
    }
    DISCO_LAUNCH_CHECK("attention_mfma_kernel");
    return DISCO_OK;
}

}  // namespace disco
