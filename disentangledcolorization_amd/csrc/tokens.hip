// tokens.hip — the superpixel-token path in exact fp32 (K6-K14 of SURVEY §2b): the linears and the encoder stacks.
//
//   token_gemm        every nn.Linear on the path as C[T,O] = A[T,K] W[O,K]^T on the fp32 matrix pipe
//                     (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, exact f32) with fused epilogues:
//                       QKV   packed in-projection of nn.MultiheadAttention, q=k=src+pos, v=src, q scaled
//                             (transformer2d.py:52-54)
//                       LOGIT mid_word_prj / trg_word_prj -> NCHW logits (model.py:134-135,187-189)
//                       HINT  trg_word_emb on [src ; m*onehot313(label) ; m] (model.py:183-185)
//   post_attention_kernel  out_proj + residual + LayerNorm + linear1 + ReLU + linear2 + residual + LayerNorm (:55-59), one launch
//   encoder_tail_kernel    the same on 16-row tiles, plus the NEXT layer's q/k/v (the default)
//   launch_encoder_stack   six layers; the attention itself: attention.hip (VALU, < 1 024 tokens) / attention_mfma.hip
// Round 6 moved the k-means kernels to kmeans.hip, the VALU attention to attention.hip and the colour selection to anchor_colors.hip
// (unchanged machine code: tools/kernel_isa_hash.py).
#include <cmath>
#include <cstdlib>
#include <vector>
#include <mutex>
#include "common.h"

namespace disco {

namespace {

enum { EPI_QKV = 0, EPI_LOGIT = 3, EPI_HINT = 4 };

struct GemmArgs {
    const float* A;      // (rows_a, K)
    int a_rep;           // virtual image i reads A image i / a_rep
    const float* pos;    // (L,64) added to A for EPI_QKV q,k tiles; pos_rep > 0: (n/pos_rep, L, 64), one per image
    int pos_rep;
    const float* W;      // (O, ldw) row-major; the first K columns are contracted
    int ldw;
    const float* bias;   // (O) or null
    int T, L, K, O;      // T = virtual rows = n_virtual * L
    float* out;          // QKV: q|k|v each (T,64); LOGIT: (n,O,L); HINT: (T,64)
    float q_scale;
    const int32_t* labels;  // HINT: (T); null = hint2regress, the anchors' ab values are embedded instead
    const float* colors;    // HINT (hint2regress): (n,2,L) NCHW ab/110 of every virtual image
    const float* mask;      // HINT: (T / mask_rep ...) indexed like A with mask_rep
    int mask_rep;
};

constexpr int GP = 65;  // padded LDS row (floats)

template <int EPI>
__global__ __launch_bounds__(256) void token_gemm_kernel(const GemmArgs g) {
    __shared__ float sA[64 * GP];
    __shared__ float sB[64 * GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.x * 64, col0 = blockIdx.y * 64;
    const bool add_pos = EPI == EPI_QKV && blockIdx.y < 2;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    for (int k0 = 0; k0 < g.K; k0 += 64) {
        if (k0) __syncthreads();
        // stage A rows [row0,row0+64) x [k0,k0+64) and W rows [col0,col0+64) x [k0,k0+64)
        for (int u = tid; u < 64 * 16; u += 256) {
            const int r = u >> 4, c4 = (u & 15) * 4;
            const int row = row0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < g.T) {
                const int img = row / g.L, t = row - img * g.L;
                const float* ap = g.A + ((size_t)(img / g.a_rep) * g.L + t) * g.K + k0 + c4;
                v = *reinterpret_cast<const float4*>(ap);
                if (add_pos) {
                    const size_t pimg = g.pos_rep > 0 ? (size_t)(img / g.pos_rep) * g.L : 0;
                    const float4 p = *reinterpret_cast<const float4*>(g.pos + (pimg + t) * 64 + c4);
                    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
                }
            }
            float* d = sA + r * GP + c4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            const int col = col0 + r;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < g.O) w = *reinterpret_cast<const float4*>(g.W + (size_t)col * g.ldw + k0 + c4);
            float* e = sB + r * GP + c4;
            e[0] = w.x; e[1] = w.y; e[2] = w.z; e[3] = w.w;
        }
        __syncthreads();
        const float* pa = sA + (wm * 32 + (lane & 31)) * GP + (lane >> 5);
        const float* pb = sB + (wn * 32 + (lane & 31)) * GP + (lane >> 5);
#pragma unroll 8
        for (int k = 0; k < 64; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
    }
    __syncthreads();
    // C tile -> LDS (reuse sA): row = (e&3) + 8*(e>>2) + 4*(lane>>5), col = lane & 31
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        sA[r * GP + wn * 32 + (lane & 31)] = acc[e];
    }
    __syncthreads();
    // thread = (row = tid/4, 16 columns)
    const int r = tid >> 2, cq = (tid & 3) * 16;
    const int row = row0 + r;
    const bool rok = row < g.T;
    const int img = rok ? row / g.L : 0, t = rok ? row - img * g.L : 0;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int col = col0 + cq + j;
        v[j] = sA[r * GP + cq + j] + ((g.bias && col < g.O) ? g.bias[col] : 0.f);
    }
    if (EPI == EPI_QKV) {
        if (rok) {
            float* o = g.out + (size_t)blockIdx.y * g.T * 64 + (size_t)row * 64 + cq;
#pragma unroll
            for (int j = 0; j < 16; ++j) o[j] = blockIdx.y == 0 ? v[j] * g.q_scale : v[j];
        }
    } else if (EPI == EPI_LOGIT) {
        if (rok) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int col = col0 + cq + j;
                if (col < g.O) g.out[((size_t)img * g.O + col) * g.L + t] = v[j];
            }
        }
    } else if (EPI == EPI_HINT) {
        if (rok) {
            const float m = g.mask[(size_t)(img / g.mask_rep) * g.L + t];
            float* o = g.out + (size_t)row * 64 + cq;
            if (g.labels) {
                const int lab = g.labels[row];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float* wr = g.W + (size_t)(cq + j) * g.ldw;
                    o[j] = v[j] + m * wr[64 + lab] + m * wr[64 + N_VOCAB];
                }
            } else {   // hint2regress (model.py:177-181): [src ; m*a ; m*b ; m] x trg_word_emb (64,67)
                const float ca = m * g.colors[((size_t)img * 2) * g.L + t], cb = m * g.colors[((size_t)img * 2 + 1) * g.L + t];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float* wr = g.W + (size_t)(cq + j) * g.ldw;
                    o[j] = v[j] + ca * wr[64] + cb * wr[65] + m * wr[66];
                }
            }
        }
    }
}

template <int EPI>
int launch_gemm(const GemmArgs& g, hipStream_t s) {
    dim3 grid(cdiv(g.T, 64), cdiv(g.O, 64));
    hipLaunchKernelGGL(token_gemm_kernel<EPI>, grid, dim3(256), 0, s, g);
    DISCO_LAUNCH_CHECK("token_gemm_kernel");
    return DISCO_OK;
}

// ---- fused post-attention half of an encoder layer (transformer2d.py:55-59) ---------------------------------------
//   x1 = LN1(x + att Wo^T + bo);  out = LN2(x1 + relu(x1 W1^T + b1) W2^T + b2)
// Every step is local to a token row, so one workgroup carries a 64-row tile through all three GEMMs with x1 and the
// 64 x 256 hidden tile in LDS: 3 launches and 2 HBM round trips per layer become one launch.  Same MFMA order per
// output element (k ascending, fp32 v_mfma_f32_32x32x2) and the same LayerNorm arithmetic as the separate
// token_gemm_kernel<RELU / RES_LN> launches, so results are bit-identical to them.
//
// The kernel is a serial chain of nine 64x64x64 GEMM steps per workgroup and its duration is the same for 4 and for 256
// workgroups (one per CU), so what counts is the length of that chain:
//   * weight tiles are DOUBLE-BUFFERED: tile i+1 travels global -> registers while step i's MFMAs run and is written to
//     the other LDS buffer afterwards (before: load, barrier, compute, barrier - nine exposed global-load latencies);
//   * operand tiles live in LDS as [k parity][row][k >> 1]: lane (row, k parity) of v_mfma_f32_32x32x2 reads its 32
//     k-values of a step with eight ds_read_b128 up front instead of one ds_read_b32 in front of every MFMA.
// LayerNorm over the 64 columns of a token row held by four neighbouring lanes (16 consecutive columns each): biased variance, eps 1e-5.
// EVERY operation is spelled out (mul_rn / add_rn / sub_rn never contract, fmaf always does): left to -ffp-contract=fast and the SLP
// vectoriser, the sum of squares came out as an irregular mix of fused and unfused steps that depended on the kernel around it - the two
// kernels that run a layer's second half (64-row and 16-row tiles) differed in a few rows per thousand by one ulp of the variance.
__device__ __forceinline__ void layer_norm16(float* v, const float* __restrict__ w, const float* __restrict__ b, int cq) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sum = add_rn(sum, v[j]);
    sum = add_rn(sum, __shfl_xor(sum, 1)); sum = add_rn(sum, __shfl_xor(sum, 2));
    const float mean = mul_rn(sum, 1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float d = sub_rn(v[j], mean); q = __builtin_fmaf(d, d, q); }
    q = add_rn(q, __shfl_xor(q, 1)); q = add_rn(q, __shfl_xor(q, 2));
    const float rstd = 1.f / sqrtf(__builtin_fmaf(q, 1.f / 64.f, 1e-5f));
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(mul_rn(sub_rn(v[j], mean), rstd), w[cq + j], b[cq + j]);
}

constexpr int RS = 36;              // operand row: 32 k-values of one parity + 4 pad floats (16-byte aligned rows, conflict-free b128)
constexpr int OT = 2 * 64 * RS;     // floats of one 64 x 64 operand tile
struct PostAttnArgs {
    const float* att;    // (T,64) attention output
    const float* x;      // (T,64) layer input (residual)
    const float *wo, *bo, *w1, *b1, *w2, *b2, *n1w, *n1b, *n2w, *n2b;
    float* out;          // (T,64)
    int T;
};
constexpr size_t POST_ATTN_SMEM = (size_t)8 * OT * sizeof(float);
static_assert(64 * GP <= OT, "the C staging reuses an operand tile");

__device__ __forceinline__ int op_idx(int row, int k) { return ((k & 1) * 64 + row) * RS + (k >> 1); }

__global__ __launch_bounds__(256) void post_attention_kernel(const PostAttnArgs g) {
    extern __shared__ float smem_pa[];
    float* sW = smem_pa;                 // 2 x OT  weight tiles (double buffer)
    float* sA = sW + 2 * OT;             // OT      attention tile (operand layout); later the row-major [64][GP] C staging
    float* sX = sA + OT;                 // OT      x1 (operand layout)
    float* sH = sX + OT;                 // 4 x OT  relu(x1 W1^T + b1), one operand tile per 64 hidden units
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.x * 64;
    const int r = tid >> 2, cq = (tid & 3) * 16;             // epilogue mapping: thread = (row, 16 columns)
    const int row = row0 + r;
    const bool rok = row < g.T;
    const int rows_valid = min(64, g.T - row0);

    // weight tile i of the chain: 0 = Wo, 1..4 = W1 rows 64(i-1).., 5..8 = W2 columns 64(i-5)..
    float4 wreg[4];
    auto wload = [&](int i) {
        const float* src = i == 0 ? g.wo : (i <= 4 ? g.w1 + (size_t)(i - 1) * 64 * 64 : g.w2 + (size_t)(i - 5) * 64);
        const int ld = i <= 4 ? 64 : 256;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int u = tid + 256 * t;
            wreg[t] = *reinterpret_cast<const float4*>(src + (size_t)(u >> 4) * ld + (u & 15) * 4);
        }
    };
    auto put4 = [&](float* tile, int rr, int c4, const float4& v) {         // columns c4..c4+3 of row rr into the operand layout
        *reinterpret_cast<float2*>(tile + rr * RS + (c4 >> 1)) = make_float2(v.x, v.z);
        *reinterpret_cast<float2*>(tile + (64 + rr) * RS + (c4 >> 1)) = make_float2(v.y, v.w);
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int u = tid + 256 * t; put4(sW + buf * OT, u >> 4, (u & 15) * 4, wreg[t]); }
    };
    auto mma = [&](f32x16& acc, const float* a_tile, const float* b_tile) {    // acc += a_tile[wm rows] * b_tile[wn rows]^T, k ascending
        const float4* pa = reinterpret_cast<const float4*>(a_tile + ((lane >> 5) * 64 + wm * 32 + (lane & 31)) * RS);
        const float4* pb = reinterpret_cast<const float4*>(b_tile + ((lane >> 5) * 64 + wn * 32 + (lane & 31)) * RS);
        float4 av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { av[q] = pa[q]; bv[q] = pb[q]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bv[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bv[q].w, acc, 0, 0, 0);
        }
    };
    auto to_staging = [&](const f32x16& acc) {   // C tile: row = (e&3) + 8*(e>>2) + 4*(lane>>5), row-major [64][GP] in sA
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rr = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            sA[rr * GP + wn * 32 + (lane & 31)] = acc[e];
        }
    };
    auto layer_norm = [&](float* v, const float* w, const float* b) { layer_norm16(v, w, b, cq); };
    f32x16 acc;

    // ---- x1 = LN1(x + att Wo^T + bo) ----
    wload(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {                               // attention tile, rows beyond T zero
        const int u = tid + 256 * t, rr = u >> 4, c4 = (u & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < rows_valid) v = *reinterpret_cast<const float4*>(g.att + (size_t)(row0 + rr) * 64 + c4);
        put4(sA, rr, c4, v);
    }
    wstore(0);
    __syncthreads();
    wload(1);
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    mma(acc, sA, sW);
    wstore(1);
    __syncthreads();                                             // every wave is done with the attention tile
    to_staging(acc);
    __syncthreads();
    {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = sA[r * GP + cq + j] + g.bo[cq + j];
        if (rok) {
            const float* rp = g.x + (size_t)row * 64 + cq;
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] += rp[j];
        }
        layer_norm(v, g.n1w, g.n1b);
#pragma unroll
        for (int j = 0; j < 16; ++j) sX[op_idx(r, cq + j)] = rok ? v[j] : 0.f;
    }
    // ---- h = relu(x1 W1^T + b1): four 64-column blocks (weight tiles 1..4) ----
    for (int cb = 0; cb < 4; ++cb) {
        const int i = 1 + cb;
        __syncthreads();                 // sX complete / tile i complete in buffer i&1 / buffer (i+1)&1 no longer read
        wload(i + 1);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        mma(acc, sX, sW + (i & 1) * OT);
        // + bias, relu, straight from the accumulator layout into the hidden tile
        const int col = cb * 64 + wn * 32 + (lane & 31);
        const float bias = g.b1[col];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rr = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            sH[cb * OT + op_idx(rr, col & 63)] = fmaxf(acc[e] + bias, 0.f);
        }
        wstore((i + 1) & 1);
    }
    // ---- out = LN2(x1 + h W2^T + b2): K = 256 in four chunks (weight tiles 5..8) ----
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int kc = 0; kc < 4; ++kc) {
        const int i = 5 + kc;
        __syncthreads();                 // sH complete / tile i complete / the other buffer no longer read
        if (kc < 3) wload(i + 1);
        mma(acc, sH + kc * OT, sW + (i & 1) * OT);
        if (kc < 3) wstore((i + 1) & 1);
    }
    to_staging(acc);                     // sA has not been read since LN1
    __syncthreads();
    {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = sA[r * GP + cq + j] + g.b2[cq + j] + sX[op_idx(r, cq + j)];
        layer_norm(v, g.n2w, g.n2b);
        if (rok) {
            float* o = g.out + (size_t)row * 64 + cq;
#pragma unroll
            for (int j = 0; j < 16; ++j) o[j] = v[j];
        }
    }
}

// ---- the encoder layer's tail on 16-row tiles (round 5: the latency path of small token counts) ----------------------------------------
//   x1 = LN1(x + att Wo^T + bo);  out = LN2(x1 + relu(x1 W1^T + b1) W2^T + b2);  and, fused behind it, the NEXT layer's packed
//   in-projection: q = ((out + pos) Wq^T + bq) q_scale, k = (out + pos) Wk^T + bk, v = out Wv^T + bv
// One image of a 256 x 256 input has 256 tokens: post_attention_kernel then runs 4 workgroups, each a serial chain of nine 64 x 64 x 64
// steps of 32 dependent v_mfma_f32_32x32x2_f32 (64 cycles each): 18 400 cycles of matrix pipe in a 20 us launch, followed by a 7 us
// launch of its own for the next layer's q, k, v.  Here a workgroup owns 16 token rows and runs on v_mfma_f32_16x16x4_f32 - the same
// arithmetic (both shapes are an fmaf chain over ascending k, bit for bit: tools/mfma_f32_shapes_probe.hip), 16 instructions of 40
// cycles of dependent latency per 64 k instead of 32 of 64 - so the chain is 640 (Wo) + 2 048 (W1: four independent blocks per wave) +
// 2 560 (W2, K = 256) + 1 536 (in-projection) cycles, on 16 workgroups.  Weights come as pre-packed B fragments (encoder_pack_kernel:
// one coalesced 16-byte load per lane and four k-steps) straight from L2 into registers: no weight tile in LDS, no barrier for it.
// Every output element sees the same operations in the same order as in token_gemm_kernel<QKV> + post_attention_kernel (k ascending,
// bias, residual, the LayerNorm's 16-column partial sums and xor-1 / xor-2 merges), so the two paths are bit-identical
// (tests/test_gpu_ops.py::test_encoder_tail_path_equals_the_tiled_one) and may be mixed freely over batch sizes.
constexpr int TL_ROWS = 16;
constexpr int TL_RSA = 64 / 4 + 4;       // A-operand row (floats) of a K = 64 tile: [k & 3][row][k >> 2] + pad, 16-byte aligned
constexpr int TL_RSH = 256 / 4 + 4;      // ... of the K = 256 hidden tile
constexpr int TL_GP = 65;
constexpr size_t PK_WO = 0, PK_W1 = 64 * 64, PK_W2 = PK_W1 + 256 * 64, PK_WIN = PK_W2 + 64 * 256;
constexpr size_t ENC_PACKED_LAYER_FLOATS = PK_WIN + 192 * 64;
// packed[((nb (K/16) + g) 64 + lane) 4 + i] = W[16 nb + (lane & 15)][16 g + 4 i + (lane >> 4)]   for W (O, K) row-major
__global__ void encoder_pack_kernel(const float* __restrict__ raw, float* __restrict__ packed) {
    const int layer = blockIdx.y;
    const float* w = raw + (size_t)layer * ENC_LAYER_FLOATS;
    float* o = packed + (size_t)layer * ENC_PACKED_LAYER_FLOATS;
    const float* in_w = w;  const float* out_w = in_w + 192 * 64 + 192;  const float* l1_w = out_w + 64 * 64 + 64;  const float* l2_w = l1_w + 256 * 64 + 256;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < ENC_PACKED_LAYER_FLOATS; u += (size_t)gridDim.x * blockDim.x) {
        const float* src; int K; size_t v;
        if (u < PK_W1) { src = out_w; K = 64; v = u; }
        else if (u < PK_W2) { src = l1_w; K = 64; v = u - PK_W1; }
        else if (u < PK_WIN) { src = l2_w; K = 256; v = u - PK_W2; }
        else { src = in_w; K = 64; v = u - PK_WIN; }
        const int i = (int)(v & 3), lane = (int)((v >> 2) & 63);
        const size_t blk = v >> 8;                                  // nb (K/16) + g
        const int g = (int)(blk % (K / 16)), nb = (int)(blk / (K / 16));
        o[u] = src[(size_t)(16 * nb + (lane & 15)) * K + 16 * g + 4 * i + (lane >> 4)];
    }
}

struct TailArgs {
    const float* att;    // (T,64) attention output
    const float* x;      // (T,64) layer input (residual)
    const float* pk;     // this layer's packed Wo | W1 | W2 (| its own in-projection, unused here)
    const float *bo, *b1, *b2, *n1w, *n1b, *n2w, *n2b;
    float* out;          // (T,64)
    // the next layer's in-projection (null: the stack's last layer)
    const float* pk_in;  // packed (192,64)
    const float* b_in;   // (192)
    const float* pos; int pos_rep;
    float* qkv;          // q | k | v, each (T,64)
    float q_scale;
    int T, L;
};

__device__ __forceinline__ int a_idx(int row, int k, int rs) { return ((k & 3) * TL_ROWS + row) * rs + (k >> 2); }

__global__ __launch_bounds__(256) void encoder_tail_kernel(const TailArgs g) {
    __shared__ __attribute__((aligned(16))) float sAtt[4 * TL_ROWS * TL_RSA];      // attention tile; later (out + pos)
    __shared__ __attribute__((aligned(16))) float sX[4 * TL_ROWS * TL_RSA];        // x1; operand layout
    __shared__ __attribute__((aligned(16))) float sV[4 * TL_ROWS * TL_RSA];        // out (the v projection's operand)
    __shared__ __attribute__((aligned(16))) float sH[4 * TL_ROWS * TL_RSH];        // relu(x1 W1^T + b1)
    __shared__ float sC[TL_ROWS * TL_GP];                                          // C staging, row-major
    __shared__ float sX1[TL_ROWS * TL_GP];                                         // x1, row-major (LN2's residual)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int am = lane & 15, akq = lane >> 4;                 // this lane's row / k residue of an A fragment (and column / k residue of a B fragment)
    const int row0 = blockIdx.x * TL_ROWS;
    const int rows_valid = min(TL_ROWS, g.T - row0);
    const float4* pk4 = reinterpret_cast<const float4*>(g.pk);
    auto bfrag = [&](size_t mat_off, int kgroups, int nb, int grp) -> float4 { return pk4[(mat_off >> 2) + ((size_t)(nb * kgroups + grp) * 64 + lane)]; };
    auto mma4 = [&](f32x4& acc, const float4& a, const float4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    };
    auto layer_norm = [&](float* v, const float* w, const float* b, int cq) { layer_norm16(v, w, b, cq); };
    // ---- the attention tile (rows beyond T zero) and Wo's fragments ----
    float4 bwo[4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) bwo[gi] = bfrag(PK_WO, 4, wave, gi);
    {
        const int rr = tid >> 4, c4 = (tid & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < rows_valid) v = *reinterpret_cast<const float4*>(g.att + (size_t)(row0 + rr) * 64 + c4);
        sAtt[a_idx(rr, c4, TL_RSA)] = v.x; sAtt[a_idx(rr, c4 + 1, TL_RSA)] = v.y; sAtt[a_idx(rr, c4 + 2, TL_RSA)] = v.z; sAtt[a_idx(rr, c4 + 3, TL_RSA)] = v.w;
    }
    // W1's fragments: this wave's four hidden blocks 4 wave .. 4 wave + 3 (in flight across the first step and LN1)
    float4 bw1[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) bw1[b][gi] = bfrag(PK_W1, 4, 4 * wave + b, gi);
    __syncthreads();
    // ---- x1 = LN1(x + att Wo^T + bo): this wave's 16 output columns ----
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float4* pa = reinterpret_cast<const float4*>(sAtt + (akq * TL_ROWS + am) * TL_RSA);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) mma4(acc, pa[gi], bwo[gi]);
#pragma unroll
        for (int e = 0; e < 4; ++e) sC[(4 * akq + e) * TL_GP + wave * 16 + am] = acc[e];
    }
    __syncthreads();
    if (tid < 64) {
        const int r = tid >> 2, cq = (tid & 3) * 16, row = row0 + r;
        const bool rok = row < g.T;
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = sC[r * TL_GP + cq + j] + g.bo[cq + j];
        if (rok) {
            const float* rp = g.x + (size_t)row * 64 + cq;
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] += rp[j];
        }
        layer_norm(v, g.n1w, g.n1b, cq);
#pragma unroll
        for (int j = 0; j < 16; ++j) { const float o = rok ? v[j] : 0.f; sX[a_idx(r, cq + j, TL_RSA)] = o; sX1[r * TL_GP + cq + j] = o; }
    }
    // W2's fragments (K = 256: 16 groups) for this wave's 16 output columns
    float4 bw2[16];
#pragma unroll
    for (int gi = 0; gi < 16; ++gi) bw2[gi] = bfrag(PK_W2, 16, wave, gi);
    __syncthreads();
    // ---- h = relu(x1 W1^T + b1): four independent 16-column blocks per wave ----
    {
        f32x4 acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4* pa = reinterpret_cast<const float4*>(sX + (akq * TL_ROWS + am) * TL_RSA);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const float4 a = pa[gi];
#pragma unroll
            for (int b = 0; b < 4; ++b) mma4(acc[b], a, bw1[b][gi]);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int col = (4 * wave + b) * 16 + am;
            const float bias = g.b1[col];
#pragma unroll
            for (int e = 0; e < 4; ++e) sH[a_idx(4 * akq + e, col, TL_RSH)] = fmaxf(acc[b][e] + bias, 0.f);
        }
    }
    // the next layer's in-projection fragments: blocks 3 wave .. 3 wave + 2 of its 12
    float4 bin[3][4];
    if (g.pk_in) {
        const float4* pi4 = reinterpret_cast<const float4*>(g.pk_in);
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) bin[b][gi] = pi4[(size_t)((3 * wave + b) * 4 + gi) * 64 + lane];
    }
    __syncthreads();
    // ---- out = LN2(x1 + h W2^T + b2) ----
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float4* pa = reinterpret_cast<const float4*>(sH + (akq * TL_ROWS + am) * TL_RSH);
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) mma4(acc, pa[gi], bw2[gi]);
#pragma unroll
        for (int e = 0; e < 4; ++e) sC[(4 * akq + e) * TL_GP + wave * 16 + am] = acc[e];
    }
    __syncthreads();
    if (tid < 64) {
        const int r = tid >> 2, cq = (tid & 3) * 16, row = row0 + r;
        const bool rok = row < g.T;
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = sC[r * TL_GP + cq + j] + g.b2[cq + j] + sX1[r * TL_GP + cq + j];
        layer_norm(v, g.n2w, g.n2b, cq);
        if (rok) {
            float* o = g.out + (size_t)row * 64 + cq;
#pragma unroll
            for (int j = 0; j < 16; ++j) o[j] = v[j];
        }
        if (g.pk_in) {
            // the operands of the next layer's in-projection: out + pos for q and k (token_gemm_kernel<QKV>'s staging: v += p), out for v
            const int img = rok ? row / g.L : 0, t = rok ? row - img * g.L : 0;
            const size_t pimg = g.pos_rep > 0 ? (size_t)(img / g.pos_rep) * g.L : 0;
            const float* pp = g.pos + (pimg + t) * 64 + cq;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float o = rok ? v[j] : 0.f;
                float qk = o;
                if (rok) qk += pp[j];
                sV[a_idx(r, cq + j, TL_RSA)] = o;
                sAtt[a_idx(r, cq + j, TL_RSA)] = qk;
            }
        }
    }
    if (!g.pk_in) return;
    __syncthreads();
    // ---- the next layer's q, k, v: 12 blocks of 16 columns, three per wave ----
    {
        f32x4 acc[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4* pq = reinterpret_cast<const float4*>(sAtt + (akq * TL_ROWS + am) * TL_RSA);
        const float4* pv = reinterpret_cast<const float4*>(sV + (akq * TL_ROWS + am) * TL_RSA);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const float4 aq = pq[gi], av = pv[gi];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int nb = 3 * wave + b;                  // (wave-uniform) 0-3: q, 4-7: k, 8-11: v
                mma4(acc[b], nb < 8 ? aq : av, bin[b][gi]);
            }
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int nb = 3 * wave + b, mat = nb >> 2, col = (nb & 3) * 16 + am;
            const float bias = g.b_in[nb * 16 + am];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = row0 + 4 * akq + e;
                float val = acc[b][e] + bias;
                if (mat == 0) val = val * g.q_scale;
                if (row < g.T) g.qkv[(size_t)mat * g.T * 64 + (size_t)row * 64 + col] = val;
            }
        }
    }
}

}  // namespace

// workspace of one encoder stack: q,k,v (3 T 64), attention output (T 64), two ping-pong layer outputs (2 T 64)
size_t encoder_ws_bytes(int n, int l) { return (size_t)n * l * (3 * 64 + 3 * 64) * sizeof(float); }

size_t encoder_packed_floats() { return (size_t)ENC_LAYERS * ENC_PACKED_LAYER_FLOATS; }

int launch_encoder_pack(const float* raw, float* packed, hipStream_t s) {
    hipLaunchKernelGGL(encoder_pack_kernel, dim3(48, ENC_LAYERS), dim3(256), 0, s, raw, packed);
    DISCO_LAUNCH_CHECK("encoder_pack_kernel");
    return DISCO_OK;
}

// up to this many token rows the stack runs its layers' tails on 16-row tiles (encoder_tail_kernel); beyond, on 64-row tiles
// (post_attention_kernel + token_gemm_kernel<QKV>).  Same results either way.  Measured at 256 ... 16 384 rows the 16-row kernel is the
// faster one everywhere (one image 36 -> 16 us per layer, 64 images 0.58 -> 0.54 ms per stack: profiles/r05_encoder_tail_ab.txt), so the
// default is "always"; DISCO_ENCODER_TAIL_ROWS overrides (0: never - the A/B switch)
static int encoder_tail_max_rows() {
    static const int v = [] { const char* e = std::getenv("DISCO_ENCODER_TAIL_ROWS"); return e ? atoi(e) : 0x7fffffff; }();
    return v;
}

// token count from which the stack's attention runs on the matrix cores (attention_mfma_kernel); DISCO_ATTN_MFMA overrides
// (0: never - the A/B switch; 1: always)
static int attention_mfma_min_tokens() {
    static const int v = [] { const char* e = std::getenv("DISCO_ATTN_MFMA"); const int t = e ? atoi(e) : 1024; return t <= 0 ? 0x7fffffff : t; }();
    return v;
}

int launch_encoder_stack(const float* x, const float* pos, int pos_rep, const float* weights, float* out, int n, int l,
                         void* ws, hipStream_t s, const std::function<void(const void*, size_t)>* dbg, const float* packed,
                         const float* key_sizes, int key_rep, float key_thr) {
    const int T = n * l;
    const bool tail = packed != nullptr && T <= encoder_tail_max_rows();
    float* qkv = reinterpret_cast<float*>(ws);
    float* att = qkv + (size_t)3 * T * 64;
    float* pp[2] = {att + (size_t)T * 64, att + (size_t)2 * T * 64};
    const float* cur = x;
    // bisecting aid: run only the first k layers.  Clamped to [1, ENC_LAYERS] (0 never wrote `out`, more indexed past the weights: advisor, round 5)
    static const int dbg_layers = [] {
        const char* e = std::getenv("DISCO_ENC_DEBUG_LAYERS");
        const int v = e ? std::min(std::max(atoi(e), 1), (int)ENC_LAYERS) : ENC_LAYERS;
        if (v != ENC_LAYERS) fprintf(stderr, "[disco] DISCO_ENC_DEBUG_LAYERS=%d: the encoder stacks run %d of %d layers - results are NOT the model's\n", v, v, (int)ENC_LAYERS);
        return v;
    }();
    for (int layer = 0; layer < dbg_layers; ++layer) {
        const float* w = weights + (size_t)layer * ENC_LAYER_FLOATS;
        const float* in_w = w;                 const float* in_b = in_w + 192 * 64;
        const float* out_w = in_b + 192;       const float* out_b = out_w + 64 * 64;
        const float* l1_w = out_b + 64;        const float* l1_b = l1_w + 256 * 64;
        const float* l2_w = l1_b + 256;        const float* l2_b = l2_w + 64 * 256;
        const float* n1_w = l2_b + 64;         const float* n1_b = n1_w + 64;
        const float* n2_w = n1_b + 64;         const float* n2_b = n2_w + 64;
        GemmArgs g{};
        g.a_rep = 1; g.T = T; g.L = l; g.mask_rep = 1;
        // q,k,v
        g.A = cur; g.pos = pos; g.pos_rep = pos_rep; g.W = in_w; g.ldw = 64; g.bias = in_b; g.K = 64; g.O = 192; g.out = qkv;
        g.q_scale = (float)std::sqrt(1.0 / 8.0);
        static const bool dbg_noqkv = [] {      // bisecting aid: the tail kernel without its fused in-projection (same results, one more launch per layer)
            const bool on = std::getenv("DISCO_TAIL_NOQKV") != nullptr;
            if (on) fprintf(stderr, "[disco] DISCO_TAIL_NOQKV is set: every encoder layer runs its own q/k/v GEMM\n");
            return on;
        }();
        if (!tail || layer == 0 || dbg_noqkv) {           // (the tail path: layers 1.. get their q, k, v from the previous layer's tail kernel)
            int rc = launch_gemm<EPI_QKV>(g, s);
            if (rc) return rc;
        }
        if (dbg) (*dbg)(qkv, (size_t)3 * T * 64 * 4);
        // from attention_mfma_min_tokens() tokens on: both contractions on the matrix cores.  The choice depends on the token count alone.
        if (l >= attention_mfma_min_tokens()) {
            const int rc = launch_attention_mfma(qkv, qkv + (size_t)T * 64, qkv + (size_t)2 * T * 64, att, n, l, s, key_sizes, key_rep, key_thr);
            if (rc) return rc;
        } else {
            const int rc = launch_attention_valu(qkv, qkv + (size_t)T * 64, qkv + (size_t)2 * T * 64, att, n, l, s, key_sizes, key_rep, key_thr);
            if (rc) return rc;
        }
        if (dbg) (*dbg)(att, (size_t)T * 64 * 4);
        // x1 = LN1(x + att Wo^T + bo); out = LN2(x1 + relu(x1 W1^T + b1) W2^T + b2): one fused launch
        float* dst = layer == dbg_layers - 1 ? out : pp[layer & 1];
        if (tail) {
            const bool more = layer + 1 < ENC_LAYERS && !dbg_noqkv;
            const float* wn = weights + (size_t)(layer + 1) * ENC_LAYER_FLOATS;      // the next layer's in_proj_weight | in_proj_bias
            TailArgs ta{att, cur, packed + (size_t)layer * ENC_PACKED_LAYER_FLOATS, out_b, l1_b, l2_b, n1_w, n1_b, n2_w, n2_b, dst,
                        more ? packed + (size_t)(layer + 1) * ENC_PACKED_LAYER_FLOATS + PK_WIN : nullptr, more ? wn + 192 * 64 : nullptr,
                        pos, pos_rep, qkv, g.q_scale, T, l};
            hipLaunchKernelGGL(encoder_tail_kernel, dim3(cdiv(T, TL_ROWS)), dim3(256), 0, s, ta);
            DISCO_LAUNCH_CHECK("encoder_tail_kernel");
        } else {
            PostAttnArgs pa{att, cur, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b, dst, T};
            constexpr size_t smem = POST_ATTN_SMEM;
            static std::atomic<int> attr_done[DISCO_MAX_DEVICES];      // per device; two host threads may get here together
            DISCO_HIP_CHECK(set_dyn_lds_once(attr_done, reinterpret_cast<const void*>(post_attention_kernel), (int)smem));
            hipLaunchKernelGGL(post_attention_kernel, dim3(cdiv(T, 64)), dim3(256), smem, s, pa);
            DISCO_LAUNCH_CHECK("post_attention_kernel");
        }
        if (dbg) (*dbg)(dst, (size_t)T * 64 * 4);
        cur = dst;
    }
    return DISCO_OK;
}

void position_encoding_host(float* h_pos, int h, int w) {
    // position_encoding.py:26-47 with num_pos_feats=32, normalize=True, scale=2*pi, temperature 1e4 (fp32 ops)
    const float scale = (float)(2.0 * M_PI);
    float dim_t[32];
    for (int i = 0; i < 32; ++i) dim_t[i] = powf(10000.f, (2.f * (float)(i / 2)) / 32.f);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float* p = h_pos + ((size_t)y * w + x) * 64;
            const float ye = (float)(y + 1) / ((float)h + 1e-6f) * scale;
            const float xe = (float)(x + 1) / ((float)w + 1e-6f) * scale;
            for (int i = 0; i < 32; ++i) {
                const float ay = ye / dim_t[i], ax = xe / dim_t[i];
                p[i] = (i & 1) ? cosf(ay) : sinf(ay);
                p[32 + i] = (i & 1) ? cosf(ax) : sinf(ax);
            }
        }
}

int launch_logits(const float* x, const float* w, float* out_nchw, int n, int l, hipStream_t s, int n_out) {
    GemmArgs g{};
    g.A = x; g.a_rep = 1; g.W = w; g.ldw = 64; g.bias = nullptr; g.T = n * l; g.L = l; g.K = 64; g.O = n_out;
    g.out = out_nchw; g.mask_rep = 1;
    return launch_gemm<EPI_LOGIT>(g, s);
}

int launch_hint_embed(const float* src, int src_rep, const int32_t* labels, const float* colors, const float* mask,
                      int mask_rep, const float* w_emb, float* out, int n, int l, hipStream_t s) {
    if (!labels == !colors) { set_error("hint_embed: exactly one of labels / colors"); return DISCO_EINVAL; }
    GemmArgs g{};
    g.A = src; g.a_rep = src_rep; g.W = w_emb; g.ldw = labels ? 64 + N_VOCAB + 1 : 64 + 2 + 1; g.bias = nullptr;
    g.T = n * l; g.L = l; g.K = 64; g.O = 64; g.out = out; g.labels = labels; g.colors = colors; g.mask = mask;
    g.mask_rep = mask_rep;
    return launch_gemm<EPI_HINT>(g, s);
}

}  // namespace disco
