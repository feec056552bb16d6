// tokens.hip — the superpixel-token path in exact fp32 (K6-K14 of SURVEY §2b).
//
//   token_gemm        every nn.Linear on the path as C[T,O] = A[T,K] W[O,K]^T on the fp32 matrix pipe
//                     (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, exact f32) with fused epilogues:
//                       QKV   packed in-projection of nn.MultiheadAttention, q=k=src+pos, v=src, q scaled
//                             (transformer2d.py:52-54)
//                       LOGIT mid_word_prj / trg_word_prj -> NCHW logits (model.py:134-135,187-189)
//                       HINT  trg_word_emb on [src ; m*onehot313(label) ; m] (model.py:183-185)
//   post_attention_kernel  out_proj + residual + LayerNorm + linear1 + ReLU + linear2 + residual + LayerNorm (:55-59), one launch
//   attention_kernel  softmax(Q K^T) V per (image, head), d_head = 8, 4 queries x every 16th key per thread
//   kmeans_anchor_kernel  Lloyd k-means (clusterkit.py:112-208) + per-cluster anchor argmax (anchor_gen.py:96-101)
//   select_colors_kernel  softmax(313) -> stable top-10 -> T-th distinct colour (anchor_gen.py:54-90) + label
//   nearest_bin_kernel    argmax of encode_ab2ind = nearest gamut bin (basic.py:177-194, model.py:166)
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
template <int QT>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* out, int L) {
    // halves of a key / value in separate arrays: the 16 partitions of a wave read 16 consecutive float4 (256 contiguous
    // bytes, no bank conflict; interleaved [key][2] rows put partitions p and p+8 on the same banks)
    __shared__ float4 sk[2][KCH];
    __shared__ float4 sv[2][KCH];
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
                    sc[i][t] = d.x + d.y;
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
            l[t] *= alpha;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[t][j] *= alpha;
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
                    l[t] += p;
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
        float ls = l[t] * scl;
#pragma unroll
        for (int sft = 1; sft < KP; sft <<= 1) ls += __shfl_xor(ls, sft);
        const float inv = 1.f / ls;
        f32x2 mine{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = o[t][j].x * scl, y = o[t][j].y * scl;
#pragma unroll
            for (int sft = 1; sft < KP; sft <<= 1) { x += __shfl_xor(x, sft); y += __shfl_xor(y, sft); }
            if (part == j) mine = f32x2{x * inv, y * inv};
        }
        const int qi = q0i + t;
        if (qi < L && part < 4) *reinterpret_cast<f32x2*>(out + base + (size_t)qi * 64 + 2 * part) = mine;
    }
}

// ---- k-means + anchors: one workgroup per image ------------------------------------------------------------------
// Lloyd iterations exactly as clusterkit.py:112-208 (first-minimum assignment, empty clusters take fallback rows in
// cluster order, stop on (sum of centre shifts)^2 < 1e-4 or 20 passes, assignment of the last distance pass) followed
// by the per-cluster anchor argmax (anchor_gen.py:96-101).  Per pass:
//   assign   1024 threads: 4 threads share a point, each over a quarter of the centres, merged in centre order (first
//            minimum preserved).  The point set lives in LDS when it fits (L <= KM_LDS_TOKENS); larger sets stream
//            through LDS in 256-point tiles (coalesced loads)
//   group    stable counting sort of the points by cluster (wave ballots + a scan over 64-point segments) into a
//            member list, so that
//   update   thread (cluster, feature) sums ONLY its members, in ascending point order - O(L D) work per pass where the
//            scan over all points per cluster was O(K L D): 7.9 ms -> per call on 8 x 1536 points, K = 8
// Summation orders are fixed, so results are run-to-run deterministic and independent of the block size.
constexpr int KMAX = 32;
constexpr int KM_LDS_TOKENS = 384;       // the point set itself lives in LDS up to this many points
constexpr int KM_PITCH = 65;
constexpr int KM_LIST_TOKENS = 4096;     // member list + assignments in LDS up to this many points (1024 x 1024 images)
template <bool XLDS, bool GLIST>
__global__ __launch_bounds__(1024) void kmeans_anchor_kernel(const float* __restrict__ x, int D, long img_stride, int t_stride,
                                                             int c_stride, const float* __restrict__ sizes,
                                                             const int32_t* __restrict__ init_idx,
                                                             const int32_t* __restrict__ fallback, int max_fallback,
                                                             int32_t* assign_out, int32_t* anchor_out, float* hint_mask,
                                                             int32_t* info, int L, int K) {
    // point t, feature c of image img: x[img*img_stride + t*t_stride + c*c_stride]; D <= 64 features.
    constexpr int NTHR = 1024, NW = NTHR / 64;
    extern __shared__ float dyn[];          // [tile rows][D+1] points (XLDS: all L, else 256), asg[L], list[L], seg[nseg][K], best
    __shared__ float cen[KMAX * 64];
    __shared__ float cnew[KMAX * 64];
    __shared__ int cnt[KMAX];               // members per cluster; < 0: empty, take fallback row -(cnt+1)
    __shared__ int start[KMAX];             // first member of the cluster in list[]
    __shared__ float shift_part[KMAX];
    __shared__ int s_events, s_stop;
    __shared__ float red_v[NTHR];
    __shared__ int red_i[NTHR];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pitch = D + 1;                // odd for D = 64 and D = 2: conflict-free row-per-thread reads
    const int nseg = (L + 63) >> 6;
    const float* X = x + (size_t)img * img_stride;
    float* xs = dyn;
    // assignments and member list: LDS up to KM_LIST_TOKENS points; beyond (GLIST) they live in global memory - the
    // assignment output itself and, until the anchors are written at the very end, the image's hint_mask row
    int* lds_ints = reinterpret_cast<int*>(dyn + (size_t)(XLDS ? L : 256) * pitch);
    int* asg = GLIST ? assign_out + (size_t)img * L : lds_ints;
    int* list = GLIST ? reinterpret_cast<int*>(hint_mask + (size_t)img * L) : lds_ints + L;
    int* seg = GLIST ? lds_ints : lds_ints + 2 * L;   // [nseg][K]: members of cluster j in segment s -> exclusive offsets
    float* best_d = reinterpret_cast<float*>(seg + nseg * K);     // [4][256] partial minima of the centre quarters
    int* best_j = reinterpret_cast<int*>(best_d + 4 * 256);
    int32_t* assign = assign_out + (size_t)img * L;
    auto xg = [&](int t, int c) -> float { return X[(size_t)t * t_stride + (size_t)c * c_stride]; };
    if (XLDS)
        for (int u = tid; u < L * D; u += NTHR) { const int t = c_stride == 1 ? u / D : u % L, c = c_stride == 1 ? u % D : u / L; xs[t * pitch + c] = xg(t, c); }
    for (int u = tid; u < K * D; u += NTHR) cen[(u / D) * 64 + (u % D)] = xg(init_idx[img * K + (u / D)], u % D);
    if (tid == 0) { s_events = 0; s_stop = 0; }
    __syncthreads();
    // squared distance of the point whose features sit at row pointer `row` to centre j
    auto dist = [&](const float* row, int j) -> float {
        float d = 0.f;
        if (D == 64) {
#pragma unroll 16
            for (int c = 0; c < 64; ++c) { const float df = row[c] - cen[j * 64 + c]; d = fmaf(df, df, d); }
        } else {    // few features: plain mul + add like the reference's ((A-B)**2).sum(-1) (clusterkit.py:253-269)
            for (int c = 0; c < D; ++c) { const float df = row[c] - cen[j * 64 + c]; d = add_rn(d, mul_rn(df, df)); }
        }
        return d;
    };
    int passes = 0;
    while (true) {
        // ---- assign: first minimum of sum_c (x - c)^2.  4 threads share a point, each over a quarter of the centres ----
        {
            const int KQ = (K + 3) >> 2, grp = tid >> 8, row = tid & 255;
            for (int base = 0; base < L; base += 256) {
                const int rows = min(256, L - base);
                if (!XLDS) {                             // stream the 256-point tile through LDS (coalesced loads)
                    __syncthreads();                     // the previous tile is consumed
                    for (int u = tid; u < rows * D; u += NTHR) {
                        const int r = c_stride == 1 ? u / D : u % rows, c = c_stride == 1 ? u % D : u / rows;
                        xs[r * pitch + c] = xg(base + r, c);
                    }
                }
                __syncthreads();                         // tile ready / best_d of the previous tile consumed
                const float* rowp = xs + (size_t)(XLDS ? base + row : row) * pitch;
                float best = INFINITY; int bi = 0x7fffffff;
                if (row < rows)
                    for (int j = grp * KQ; j < min(K, (grp + 1) * KQ); ++j) { const float d = dist(rowp, j); if (d < best) { best = d; bi = j; } }
                best_d[grp * 256 + row] = best; best_j[grp * 256 + row] = bi;
                __syncthreads();
                if (grp == 0 && row < rows) {            // merge in centre order: a later quarter wins only when strictly smaller
                    float bd = best_d[row]; int bj = best_j[row];
#pragma unroll
                    for (int q = 1; q < 4; ++q) if (best_d[q * 256 + row] < bd) { bd = best_d[q * 256 + row]; bj = best_j[q * 256 + row]; }
                    asg[base + row] = bj;
                }
            }
        }
        __syncthreads();
        // ---- group: stable counting sort by cluster.  Segment s = points 64s .. 64s+63, handled by one wave ----
        for (int sgm = wave; sgm < nseg; sgm += NW) {
            const int t = sgm * 64 + lane;
            const int mine = t < L ? asg[t] : -1;
            for (int j = 0; j < K; ++j) {
                const unsigned long long ball = __ballot(mine == j);
                if (lane == 0) seg[sgm * K + j] = __popcll(ball);
            }
        }
        __syncthreads();
        if (tid < K) {                                  // exclusive scan over the segments of cluster tid
            int run = 0;
            for (int sgm = 0; sgm < nseg; ++sgm) { const int v = seg[sgm * K + tid]; seg[sgm * K + tid] = run; run += v; }
            cnt[tid] = run;
        }
        __syncthreads();
        // empty clusters take a fallback row, in cluster order (sequential bookkeeping by one thread)
        if (tid == 0) {
            int run = 0;
            for (int j = 0; j < K; ++j) {
                start[j] = run; run += cnt[j];
                if (cnt[j] == 0) {
                    const int e = s_events++;
                    const int row = (fallback && e < max_fallback) ? fallback[(size_t)img * max_fallback + e] : 0;
                    cnt[j] = -(row + 1);   // marker: negative = use row
                }
            }
        }
        __syncthreads();
        for (int sgm = wave; sgm < nseg; sgm += NW) {
            const int t = sgm * 64 + lane;
            const int mine = t < L ? asg[t] : -1;
            const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
            for (int j = 0; j < K; ++j) {
                const unsigned long long ball = __ballot(mine == j);
                if (mine == j) list[start[j] + seg[sgm * K + j] + __popcll(ball & below)] = t;
            }
        }
        __syncthreads();
        // ---- update: thread = (cluster, feature), members in ascending point order ----
        for (int u = tid; u < K * D; u += NTHR) {
            const int j = u / D, c = u % D;
            float sum;
            if (cnt[j] < 0) sum = xg(-cnt[j] - 1, c);
            else {
                sum = 0.f;
                const int* mem = list + start[j];
                const int m = cnt[j];
                if (XLDS) { for (int i = 0; i < m; ++i) sum += xs[mem[i] * pitch + c]; }
                else {
                    int i = 0;
                    for (; i + 4 <= m; i += 4) {         // 4 loads in flight, added in order
                        const float v0 = xg(mem[i], c), v1 = xg(mem[i + 1], c), v2 = xg(mem[i + 2], c), v3 = xg(mem[i + 3], c);
                        sum += v0; sum += v1; sum += v2; sum += v3;
                    }
                    for (; i < m; ++i) sum += xg(mem[i], c);
                }
                sum = sum / (float)m;
            }
            cnew[j * 64 + c] = sum;
        }
        __syncthreads();
        // centre shift = sum_j sqrt(sum_c (new-old)^2)
        if (tid < K) {
            float q = 0.f;
            for (int c = 0; c < D; ++c) { const float d = cnew[tid * 64 + c] - cen[tid * 64 + c]; q = __builtin_fmaf(d, d, q); }
            shift_part[tid] = sqrtf(q);
        }
        __syncthreads();
        ++passes;
        if (tid == 0) {
            float sh = 0.f;
            for (int j = 0; j < K; ++j) sh += shift_part[j];
            s_stop = (sh * sh < 1e-4f) || passes >= 20;
        }
        for (int u = tid; u < K * D; u += NTHR) cen[(u / D) * 64 + (u % D)] = cnew[(u / D) * 64 + (u % D)];
        __syncthreads();
        if (s_stop) break;
    }
    if (!GLIST) for (int t = tid; t < L; t += NTHR) assign[t] = asg[t];
    __syncthreads();                        // GLIST: every thread is done with the member list before hint_mask is rewritten
    // anchors: per cluster the first argmax of [assign==j] + sizes*0.01 (exact fp32 ops, no fma)
    const float* sz = sizes + (size_t)img * L;
    float* hm = hint_mask + (size_t)img * L;
    for (int t = tid; t < L; t += NTHR) hm[t] = 0.f;
    __syncthreads();
    // (value, index) maxima under "greater value, then lower index" - an order, so any reduction tree gives the first maximum:
    // shuffles inside each wave, one LDS round across the 16 waves, all K clusters behind the same two barriers
    for (int j = 0; j < K; ++j) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int t = tid; t < L; t += NTHR) {
            const float sc = add_rn(asg[t] == j ? 1.f : 0.f, mul_rn(sz[t], 0.01f));
            if (sc > bv) { bv = sc; bi = t; }   // ascending t: keeps the first maximum
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const float ov = __shfl_xor(bv, sft); const int oi = __shfl_xor(bi, sft);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { red_v[j * NW + wave] = bv; red_i[j * NW + wave] = bi; }
    }
    __syncthreads();
    if (tid < K) {
        float bv = red_v[tid * NW]; int bi = red_i[tid * NW];
        for (int w = 1; w < NW; ++w) {
            const float ov = red_v[tid * NW + w]; const int oi = red_i[tid * NW + w];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        anchor_out[img * K + tid] = bi;
        red_i[tid * NW] = bi;
    }
    __syncthreads();
    if (tid == 0)
        for (int j = 0; j < K; ++j) hm[red_i[j * NW]] += 1.f;       // sequential: two clusters may share an anchor
    if (tid == 0 && info) { info[img * 2] = passes; info[img * 2 + 1] = s_events; }
}

// ---- k-means + anchors for up to 256 points of 64 features: the latency path (round 5) -----------------------------------------------
// A 256 x 256 image has 16 x 16 tokens, and its k-means is a chain of ~5 Lloyd passes on ONE workgroup: what counts is the length of
// a pass, not its work.  kmeans_anchor_kernel spends ~30 us per pass at this size (ten barriers of 1024 threads, a counting sort, member
// sums as chains of dependent LDS reads); this kernel runs a pass behind two barriers:
//   assign   4 threads per point (a quarter of the centres each), the point's 64 features in REGISTERS for the whole kernel, the centres
//            as ds_read_b128 broadcasts; the quarters of a point are neighbouring lanes and merge by shuffles under the order
//            (distance, centre index) = the first minimum over all centres
//   update   one WAVE per cluster, lane = feature: the members from ballot masks of the assignments into a wave-local list, their rows
//            streamed from it sixteen deep and added in ascending point order; the centre's shift as a chain over v_readlane'd lanes - no member list, no
//            sort, no cross-wave reduction; new centres go to the other of two centre buffers (no copy pass)
//   stop     every thread sums the K shifts itself (same order), so the decision needs no third barrier
// The arithmetic is kmeans_anchor_kernel's, expression by expression (fmaf distance chain over ascending features, first minimum, member
// sums in ascending point order divided by the count, shift = sum_j sqrt(sum_c d^2) in ascending order), so assignments, pass counts and
// empty-cluster events are bit-identical to it (tests/test_gpu_ops.py::test_kmeans_small_kernel_equals_the_general_one).
constexpr int KS_PITCH = 65;       // point rows in LDS (floats): odd, conflict-free both by row and by column
constexpr int KS_CP = 68;          // centre rows: 16-byte aligned, consecutive rows on different banks
constexpr int KS_MAXL = 256;
__global__ __launch_bounds__(1024) void kmeans_small_kernel(const float* __restrict__ x, const float* __restrict__ sizes,
                                                            const int32_t* __restrict__ init_idx, const int32_t* __restrict__ fallback,
                                                            int max_fallback, int32_t* assign_out, int32_t* anchor_out, float* hint_mask,
                                                            int32_t* info, int L, int K) {
    extern __shared__ float dyn[];          // [L][KS_PITCH] points
    __shared__ __attribute__((aligned(16))) float cen[2][KMAX * KS_CP];
    __shared__ int asg[KS_MAXL];
    __shared__ float hm_l[KS_MAXL];
    __shared__ int cnt[KMAX];
    __shared__ float shift_part[KMAX];
    __shared__ int s_anchor[KMAX];
    __shared__ int s_events, s_any_empty;
    __shared__ unsigned short mlist[16][KS_MAXL];       // per wave: the member list of the cluster it is summing
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* X = x + (size_t)img * L * 64;
    float* xs = dyn;
    for (int u = tid; u < L * 64; u += 1024) xs[(u >> 6) * KS_PITCH + (u & 63)] = X[u];
    for (int u = tid; u < K * 64; u += 1024) cen[0][(u >> 6) * KS_CP + (u & 63)] = X[(size_t)init_idx[img * K + (u >> 6)] * 64 + (u & 63)];
    if (tid == 0) { s_events = 0; s_any_empty = 0; }
    __syncthreads();
    const int t = tid >> 2, q = tid & 3;          // point, quarter of the centres
    float row[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) row[c] = t < L ? xs[t * KS_PITCH + c] : 0.f;
    const int KQ = (K + 3) >> 2;
    int cur = 0, passes = 0;
    while (true) {
        // ---- assign ----
        {
            float best = INFINITY; int bi = 0x7fffffff;
            for (int j = q * KQ; j < min(K, (q + 1) * KQ); ++j) {
                const float4* cp = reinterpret_cast<const float4*>(cen[cur] + j * KS_CP);
                float d = 0.f;
#pragma unroll
                for (int c4 = 0; c4 < 16; ++c4) {
                    const float4 cv = cp[c4];
                    float df = row[4 * c4] - cv.x; d = fmaf(df, df, d);
                    df = row[4 * c4 + 1] - cv.y; d = fmaf(df, df, d);
                    df = row[4 * c4 + 2] - cv.z; d = fmaf(df, df, d);
                    df = row[4 * c4 + 3] - cv.w; d = fmaf(df, df, d);
                }
                if (d < best) { best = d; bi = j; }
            }
#pragma unroll
            for (int sft = 1; sft < 4; sft <<= 1) {
                const float od = __shfl_xor(best, sft); const int oj = __shfl_xor(bi, sft);
                if (od < best || (od == best && oj < bi)) { best = od; bi = oj; }
            }
            if (q == 0 && t < L) asg[t] = bi;
        }
        __syncthreads();
        // ---- update: wave = cluster, lane = feature ----
        const int nxt = cur ^ 1;
        // the centre's shift contribution sqrt(sum_c (new - old)^2), the sum as the chain q += d d over ascending features
        auto shift_of = [&](float dlane) -> float {
            float qv = 0.f;
#pragma unroll
            for (int c = 0; c < 64; ++c) { const float dc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dlane), c)); qv = __builtin_fmaf(dc, dc, qv); }
            return sqrtf(qv);
        };
        for (int j = wave; j < K; j += 16) {
            // the members of cluster j, ascending, as the byte offsets of their rows: every member lane writes its own entry at its rank
            // (ballot + popcount below the lane) into this wave's list - wave-local, no barrier - and the rows then stream from the list
            // sixteen deep (walking the masks block by block, eight rows at a time, cost ~105 cycles per member: r05_kmeans_coop_phases.txt)
            unsigned short* lst = mlist[wave];
            int m = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int tt = b * 64 + lane;
                const int av = tt < L ? asg[tt] : -1;
                const unsigned long long mk = __ballot(av == j);
                if (av == j) lst[m + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)tt;
                m += __popcll(mk);
            }
            m = __builtin_amdgcn_readfirstlane(m);
            if (m > 0) {
                float sum = 0.f;
                const char* xb = reinterpret_cast<const char*>(xs) + lane * 4;
#pragma unroll 1
                for (int i0 = 0; i0 < m; i0 += 64) {
                    const int cc = min(64, m - i0);                                              // (scalar)
                    const int ov = lane < cc ? (int)lst[i0 + lane] * (KS_PITCH * 4) : 0;         // ONE read: the next 64 members' row offsets
                    float v[2][16];
                    auto ld = [&](int buf, int base) __attribute__((always_inline)) {
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[buf][u] = *reinterpret_cast<const float*>(xb + __builtin_amdgcn_readlane(ov, base + u));
                    };
                    ld(0, 0);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        if (gq * 16 >= cc) break;
                        if (gq < 3 && (gq + 1) * 16 < cc) ld((gq + 1) & 1, (gq + 1) * 16);
                        if ((gq + 1) * 16 <= cc) {
#pragma unroll
                            for (int u = 0; u < 16; ++u) sum += v[gq & 1][u];
                        } else {
#pragma unroll
                            for (int u = 0; u < 16; ++u) if (gq * 16 + u < cc) sum += v[gq & 1][u];
                        }
                    }
                }
                sum = sum / (float)m;
                cen[nxt][j * KS_CP + lane] = sum;
                const float sh = shift_of(sum - cen[cur][j * KS_CP + lane]);
                if (lane == 0) { shift_part[j] = sh; cnt[j] = m; }
            } else if (lane == 0) { cnt[j] = 0; s_any_empty = 1; }
        }
        __syncthreads();
        if (s_any_empty) {
            // empty clusters take a fallback row, in cluster order (sequential bookkeeping by one thread); rare
            if (tid == 0) {
                for (int j = 0; j < K; ++j)
                    if (cnt[j] == 0) {
                        const int e = s_events++;
                        const int r = (fallback && e < max_fallback) ? fallback[(size_t)img * max_fallback + e] : 0;
                        cnt[j] = -(r + 1);
                    }
            }
            __syncthreads();
            for (int j = wave; j < K; j += 16)
                if (cnt[j] < 0) {
                    const float sum = X[(size_t)(-cnt[j] - 1) * 64 + lane];
                    cen[nxt][j * KS_CP + lane] = sum;
                    const float sh = shift_of(sum - cen[cur][j * KS_CP + lane]);
                    if (lane == 0) shift_part[j] = sh;
                }
            __syncthreads();
            if (tid == 0) s_any_empty = 0;
        }
        ++passes;
        float sh = 0.f;
        for (int j = 0; j < K; ++j) sh += shift_part[j];
        cur = nxt;
        if ((sh * sh < 1e-4f) || passes >= 20) break;
    }
    // anchors: per cluster the first argmax of [assign==j] + sizes*0.01 (exact fp32 ops, no fma); one wave per cluster
    const float* sz = sizes + (size_t)img * L;
    for (int j = wave; j < K; j += 16) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int tt = lane; tt < L; tt += 64) {
            const float sc = add_rn(asg[tt] == j ? 1.f : 0.f, mul_rn(sz[tt], 0.01f));
            if (sc > bv) { bv = sc; bi = tt; }
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const float ov = __shfl_xor(bv, sft); const int oi = __shfl_xor(bi, sft);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { anchor_out[img * K + j] = bi; s_anchor[j] = bi; }
    }
    for (int tt = tid; tt < L; tt += 1024) { assign_out[(size_t)img * L + tt] = asg[tt]; hm_l[tt] = 0.f; }
    __syncthreads();
    if (tid == 0) {
        for (int j = 0; j < K; ++j) hm_l[s_anchor[j]] += 1.f;       // sequential: two clusters may share an anchor
        if (info) { info[img * 2] = passes; info[img * 2 + 1] = s_events; }
    }
    __syncthreads();
    for (int tt = tid; tt < L; tt += 1024) hint_mask[(size_t)img * L + tt] = hm_l[tt];
}

// ---- k-means + anchors for MORE than 256 points of 64 features: kmeans_small_kernel's pass, tile by tile (round 5) ------------------------
// The --no_resize path clusters 1 024 ... 16 384 tokens per image, still on one workgroup (the member sums are one sequential chain per
// cluster and feature, in ascending point order: that is what makes the result independent of everything but the data).
// kmeans_anchor_kernel spends ~55 ns per point and pass there (a counting sort per pass, member rows fetched from L2 four at a time:
// 82 us per pass at 1 536 points, 1.2 ms at 16 384).  Here the points stream through LDS in tiles of 256, ONCE per pass, and a tile is
// assigned AND added to the running member sums while it is there: tiles come in ascending point order, members inside a tile in
// ascending order, so the chain of additions per (cluster, feature) is the same as before - bit-identical centres, shifts, pass counts
// and assignments (tests: the k-means cases of tests/test_gpu_ops.py run both kernels) - at ~17 ns per point and pass.
__global__ __launch_bounds__(1024) void kmeans_tiled_kernel(const float* __restrict__ x, const float* __restrict__ sizes,
                                                            const int32_t* __restrict__ init_idx, const int32_t* __restrict__ fallback,
                                                            int max_fallback, int32_t* assign_out, int32_t* anchor_out, float* hint_mask,
                                                            int32_t* info, int L, int K) {
    extern __shared__ float dyn[];          // 2 x [256][KS_PITCH]: the tile being worked on and the one being written
    __shared__ __attribute__((aligned(16))) float cen[2][KMAX * KS_CP];
    __shared__ int asg[2][KS_MAXL];         // the tiles' assignments (double-buffered like the tiles)
    __shared__ int cnt[KMAX];
    __shared__ float shift_part[KMAX];
    __shared__ int s_anchor[KMAX];
    __shared__ int s_events, s_any_empty;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* X = x + (size_t)img * L * 64;
    int32_t* assign = assign_out + (size_t)img * L;
    for (int u = tid; u < K * 64; u += 1024) cen[0][(u >> 6) * KS_CP + (u & 63)] = X[(size_t)init_idx[img * K + (u >> 6)] * 64 + (u & 63)];
    if (tid == 0) { s_events = 0; s_any_empty = 0; }
    const int t = tid >> 2, q = tid & 3;          // point of the tile, quarter of the centres
    const int KQ = (K + 3) >> 2;
    const int ntiles = (L + 255) >> 8;
    int cur = 0, passes = 0;
    auto shift_of = [&](float dlane) -> float {
        float qv = 0.f;
#pragma unroll
        for (int c = 0; c < 64; ++c) { const float dc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dlane), c)); qv = __builtin_fmaf(dc, dc, qv); }
        return sqrtf(qv);
    };
    // a tile travels L2 -> registers (four coalesced 16-byte loads per thread, in flight while the previous tile is worked on) -> LDS
    float4 pre[4];
    auto fetch = [&](int b) {
        const int base = b << 8, rows = min(256, L - base);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u4 = tid + 1024 * i;                     // float4 index inside the tile: row u4 >> 4, columns 4 (u4 & 15) ..
            pre[i] = (u4 >> 4) < rows ? *reinterpret_cast<const float4*>(X + (size_t)base * 64 + (size_t)u4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto deposit = [&](int buf) {
        float* xs = dyn + buf * (256 * KS_PITCH);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u4 = tid + 1024 * i;
            float* d = xs + (u4 >> 4) * KS_PITCH + (u4 & 15) * 4;
            d[0] = pre[i].x; d[1] = pre[i].y; d[2] = pre[i].z; d[3] = pre[i].w;
        }
    };
    while (true) {
        const int nxt = cur ^ 1;
        // this wave's clusters: wave and wave + 16 (K <= 32); running member sum (lane = feature) and member count of each
        float sum0 = 0.f, sum1 = 0.f; int m0 = 0, m1 = 0;
        fetch(0);
        __syncthreads();                                      // the previous pass is done with both tile buffers (and the centres are written)
        deposit(0);
        for (int b = 0; b < ntiles; ++b) {
            const int base = b << 8, rows = min(256, L - base), buf = b & 1;
            const float* xs = dyn + buf * (256 * KS_PITCH);
            if (b + 1 < ntiles) fetch(b + 1);
            __syncthreads();                                  // tile b is in LDS
            // ---- assign ----
            {
                // this thread's centres j0 .. j0 + nq - 1 (at most 8: K <= 32), one distance chain each; the point's features are read four
                // at a time as the chains advance (held all 64 at once next to the prefetched tile they spilled to scratch memory)
                const int j0 = q * KQ, nq = min(K, (q + 1) * KQ) - j0;
                float d[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) d[jj] = 0.f;
                const float* rp = xs + (t < rows ? t : 0) * KS_PITCH;
#pragma unroll 4
                for (int c4 = 0; c4 < 16; ++c4) {
                    const float r0 = rp[4 * c4], r1 = rp[4 * c4 + 1], r2 = rp[4 * c4 + 2], r3 = rp[4 * c4 + 3];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        if (jj >= nq) break;
                        const float4 cv = *reinterpret_cast<const float4*>(cen[cur] + (j0 + jj) * KS_CP + 4 * c4);
                        float df = r0 - cv.x; d[jj] = fmaf(df, df, d[jj]);
                        df = r1 - cv.y; d[jj] = fmaf(df, df, d[jj]);
                        df = r2 - cv.z; d[jj] = fmaf(df, df, d[jj]);
                        df = r3 - cv.w; d[jj] = fmaf(df, df, d[jj]);
                    }
                }
                float best = INFINITY; int bi = 0x7fffffff;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    if (jj < nq && d[jj] < best) { best = d[jj]; bi = j0 + jj; }
#pragma unroll
                for (int sft = 1; sft < 4; sft <<= 1) {
                    const float od = __shfl_xor(best, sft); const int oj = __shfl_xor(bi, sft);
                    if (od < best || (od == best && oj < bi)) { best = od; bi = oj; }
                }
                if (q == 0 && t < rows) { asg[buf][t] = bi; assign[base + t] = bi; }
            }
            __syncthreads();                                  // the tile's assignments are written; everybody is done with tile b - 1
            if (b + 1 < ntiles) deposit(buf ^ 1);             // (its buffer takes tile b + 1 while this one is added up)
            // ---- add the tile's members to the running sums: wave = cluster, lane = feature, ascending point order ----
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = wave + 16 * h;
                if (j >= K) break;
                unsigned long long mk[4]; int m = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int tt = bb * 64 + lane;
                    const int av = tt < rows ? asg[buf][tt] : -1;
                    mk[bb] = __ballot(av == j);
                    m += __popcll(mk[bb]);
                }
                float sum = h ? sum1 : sum0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    unsigned long long mask = mk[bb];
                    while (mask) {
                        float v[8]; bool ok[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            ok[u] = mask != 0ull;
                            const int tt = bb * 64 + (ok[u] ? __builtin_ctzll(mask) : 0);
                            mask &= mask - 1ull;
                            v[u] = xs[tt * KS_PITCH + lane];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (ok[u]) sum += v[u];
                    }
                }
                if (h) { sum1 = sum; m1 += m; } else { sum0 = sum; m0 += m; }
            }
        }
        // ---- the new centres and their shifts ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = wave + 16 * h;
            if (j >= K) break;
            const int m = h ? m1 : m0;
            if (m > 0) {
                const float sum = (h ? sum1 : sum0) / (float)m;
                cen[nxt][j * KS_CP + lane] = sum;
                const float sh = shift_of(sum - cen[cur][j * KS_CP + lane]);
                if (lane == 0) { shift_part[j] = sh; cnt[j] = m; }
            } else if (lane == 0) { cnt[j] = 0; s_any_empty = 1; }
        }
        __syncthreads();
        if (s_any_empty) {
            if (tid == 0) {
                for (int j = 0; j < K; ++j)
                    if (cnt[j] == 0) {
                        const int e = s_events++;
                        const int r = (fallback && e < max_fallback) ? fallback[(size_t)img * max_fallback + e] : 0;
                        cnt[j] = -(r + 1);
                    }
            }
            __syncthreads();
            for (int j = wave; j < K; j += 16)
                if (cnt[j] < 0) {
                    const float sum = X[(size_t)(-cnt[j] - 1) * 64 + lane];
                    cen[nxt][j * KS_CP + lane] = sum;
                    const float sh = shift_of(sum - cen[cur][j * KS_CP + lane]);
                    if (lane == 0) shift_part[j] = sh;
                }
            __syncthreads();
            if (tid == 0) s_any_empty = 0;
        }
        ++passes;
        float sh = 0.f;
        for (int j = 0; j < K; ++j) sh += shift_part[j];
        cur = nxt;
        if ((sh * sh < 1e-4f) || passes >= 20) break;
    }
    __syncthreads();                        // every assignment of the last pass is in assign_out (this workgroup's own writes)
    // anchors: per cluster the first argmax of [assign==j] + sizes*0.01 (exact fp32 ops, no fma); one wave per cluster
    const float* sz = sizes + (size_t)img * L;
    float* hm = hint_mask + (size_t)img * L;
    for (int j = wave; j < K; j += 16) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int tt = lane; tt < L; tt += 64) {
            const float sc = add_rn(assign[tt] == j ? 1.f : 0.f, mul_rn(sz[tt], 0.01f));
            if (sc > bv) { bv = sc; bi = tt; }
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const float ov = __shfl_xor(bv, sft); const int oi = __shfl_xor(bi, sft);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { anchor_out[img * K + j] = bi; s_anchor[j] = bi; }
    }
    for (int tt = tid; tt < L; tt += 1024) hm[tt] = 0.f;
    __syncthreads();
    if (tid == 0) {
        for (int j = 0; j < K; ++j) hm[s_anchor[j]] += 1.f;       // sequential: two clusters may share an anchor
        if (info) { info[img * 2] = passes; info[img * 2 + 1] = s_events; }
    }
}

// ---- k-means + anchors for more than 512 points on SEVERAL workgroups per image (round 5) ----------------------------------------------
// kmeans_tiled_kernel walks an image's tiles one after the other on one workgroup: 40 us per Lloyd pass at 1 536 tokens, 470 at 16 384 -
// 13 % of a --no_resize forward.  Here workgroup g of G keeps tiles 2g and 2g + 1 (512 points) RESIDENT in LDS for the whole kernel, all
// workgroups assign their points at once (the distance pass is VALU-bound: 12 300 cycles per 512 points on one CU) and sort them by
// (cluster, point) - a counting sort from ballot masks - and the member sums - one sequential chain per (cluster, feature) in ascending
// point order: the property that makes the result independent of everything but the data - travel down the workgroups as a pipeline:
// wave j of workgroup g waits for g - 1's running sum of cluster j, adds its own members from its list (rows streamed from LDS sixteen
// deep), hands on; the last workgroup divides, measures the shift, decides, and publishes the new centres, which everybody picks up.
// Same additions in the same order as the one-workgroup kernels: bit-identical assignments, pass counts and events
// (tests/test_gpu_ops.py::test_kmeans_small_kernel_equals_the_general_one runs it against the general kernel).
// Exchange: every word is an aligned 8 bytes {value, tag}, the tag naming the pass (and for a sum the writer and the member count), so
// the reader polls the DATA: one round trip per hop (a flag behind the data cost a store acknowledge, a flag round trip and a data round
// trip).  s_memtime stamps (profiles/r05_kmeans_coop_phases.txt): a hop is ~2 700 cycles whichever way the words travel; what the first
// version of this kernel spent per workgroup was its own sums - ~6 700 cycles walking ballot masks (105 per member) - now ~4 000 from
// the lists.  1 536 tokens 40 -> 23 us per pass, 16 384 tokens 472 -> 198.  A REDUCER form (workers publish their lists, one workgroup
// streams every member's row from L2 in order) was built too: bit-identical, but at 16 rows in flight per wave an L2 row costs ~270
// cycles per member against ~70 from LDS - 264 us per pass at 16 384 tokens; it would need a ring of ~64 rows per cluster in flight
// (LDS-DMA + a second wave per cluster for the list polls) to win.
// The workgroups of an image spin on each other's words, so they must all be resident: the launcher takes this kernel only while n x G
// fits a quarter of the CUs, and every spin is bounded by a deadline (5e9 shader cycles from kernel entry); a wave that runs into it
// TRAPS - the launch fails loudly (hipErrorLaunchFailure), never returns a wrong clustering.
constexpr int KC_MAXG = 64;
struct KmCoopCtl { int done[KC_MAXG]; int pad[4]; };
// per image: the running member sums and the pass's new centres, [KMAX][64] 8-byte words {value, tag} each, + the flags
constexpr size_t KC_IMG_BYTES = ((size_t)(2 * KMAX * 64) * 8 + sizeof(KmCoopCtl) + 255) & ~(size_t)255;
constexpr int KC_MAX_POINTS = 1 << 18;
__global__ __launch_bounds__(1024) void kmeans_coop_kernel(const float* __restrict__ x, const float* __restrict__ sizes,
                                                           const int32_t* __restrict__ init_idx, const int32_t* __restrict__ fallback,
                                                           int max_fallback, int32_t* assign_out, int32_t* anchor_out, float* hint_mask,
                                                           int32_t* info, int L, int K, int G, unsigned char* scratch) {
    extern __shared__ float dyn[];          // 2 x [256][KS_PITCH]: this workgroup's two tiles, resident
    __shared__ __attribute__((aligned(16))) float cen[2][KMAX * KS_CP];
    __shared__ int asg[2][KS_MAXL];
    __shared__ int cnt[KMAX];
    __shared__ float shift_part[KMAX];
    __shared__ int s_anchor[KMAX];
    __shared__ int s_events, s_any_empty, s_stop;
    __shared__ int cntblk[KMAX][8];             // members of cluster j in 64-point block b of this workgroup's 512 points; then their start in order[]
    __shared__ int seg[KMAX][2];                // cluster j's segment of order[]: start (as a byte offset into order), count
    __shared__ int order[512];                  // this workgroup's points sorted by (cluster, point), as the byte offsets of their rows in dyn
    const int img = blockIdx.x / G, g = blockIdx.x - img * G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool last_wg = g == G - 1;
    const float* X = x + (size_t)img * L * 64;
    int32_t* assign = assign_out + (size_t)img * L;
    unsigned char* sc = scratch + (size_t)img * KC_IMG_BYTES;
    typedef unsigned long long u64;
    u64* gS = reinterpret_cast<u64*>(sc);                        // running member sums [K][64]: {sum, pass << 24 | writer << 18 | members so far}
    u64* gC = gS + KMAX * 64;                                    // the pass's new centres [K][64]: {centre, pass << 24 | stop}
    KmCoopCtl* ctl = reinterpret_cast<KmCoopCtl*>(gC + KMAX * 64);
    // Everything the workgroups exchange goes through SYSTEM-scope relaxed accesses (stores written through, loads past the non-coherent
    // caches: the XCDs' L2s do not see each other's lines) - no cache-wide write-back / invalidate per hop, which an agent-scope
    // release / acquire pair costs.  (A variant that placed an image's workgroups on ONE XCD - workgroup id mod 8, verified through
    // HW_REG_XCC_ID - and exchanged through that XCD's L2 - plain stores, polls as atomic ORs executed in the L2; sc0 loads hit the CU's own
    // L1 forever, sc1 accesses go to memory like system-scope ones - was built and measured equal to 0.1 us at every size: the exchange,
    // ~1 400 cycles per hop, is not where a pass spends its time.  Removed.)
    auto st_u = [](u64* ptr, unsigned v, unsigned tag) {
        __hip_atomic_store(ptr, ((u64)tag << 32) | (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto st_i = [](int* ptr, int v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    auto ld_i = [](const int* ptr) -> int { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    // (s_memtime - the shader clock - not s_memrealtime: this runs inside every poll iteration)
    const unsigned long long deadline = __builtin_amdgcn_s_memtime() + 5000000000ull;     // 2-4 s
    // this wave polls one word per lane until every taking-part lane's tag equals `want` under `mask`; a lane with !mine takes no part.
    // A wave that runs into the deadline TRAPS.
    auto poll = [&](const u64* ptr, bool mine, unsigned want, unsigned mask) -> u64 {
        u64 w = 0;
        for (;;) {
            if (mine) w = __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const bool ok = !mine || (((unsigned)(w >> 32)) & mask) == want;
            if (__ballot(!ok) == 0ull) break;
            if (__builtin_amdgcn_s_memtime() > deadline) __builtin_trap();
            __builtin_amdgcn_s_sleep(1);
        }
        return w;
    };
    constexpr unsigned MASK_S = 0xfffc0000u, MASK_C = 0xff000000u;
    const int tile0 = 2 * g, ntl = min(2, ((L + 255) >> 8) - tile0);
    for (int u = tid; u < ntl * 256 * 16; u += 1024) {          // float4 index: tile, row, 4 columns
        const int tl = u >> 12, r = (u >> 4) & 255, c4 = (u & 15) * 4;
        const int pt = (tile0 + tl) * 256 + r;
        const float4 v = pt < L ? *reinterpret_cast<const float4*>(X + (size_t)pt * 64 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = dyn + tl * (256 * KS_PITCH) + r * KS_PITCH + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int u = tid; u < K * 64; u += 1024) cen[0][(u >> 6) * KS_CP + (u & 63)] = X[(size_t)init_idx[img * K + (u >> 6)] * 64 + (u & 63)];
    if (tid == 0) { s_events = 0; s_any_empty = 0; s_stop = 0; }
    __syncthreads();
    const int t = tid >> 2, q = tid & 3;
    const int KQ = (K + 3) >> 2;
    int cur = 0, passes = 0;
    auto shift_of = [&](float dlane) -> float {
        float qv = 0.f;
#pragma unroll
        for (int c = 0; c < 64; ++c) { const float dc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dlane), c)); qv = __builtin_fmaf(dc, dc, qv); }
        return sqrtf(qv);
    };
    while (true) {
        const int nxt = cur ^ 1, p = passes + 1;
        const unsigned ptag = (unsigned)p << 24;
        // ---- assign this workgroup's tiles ----
        for (int tl = 0; tl < ntl; ++tl) {
            const int base = (tile0 + tl) << 8, rows = min(256, L - base);
            const float* xs = dyn + tl * (256 * KS_PITCH);
            const int j0 = q * KQ, nq = min(K, (q + 1) * KQ) - j0;
            float d[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) d[jj] = 0.f;
            const float* rp = xs + (t < rows ? t : 0) * KS_PITCH;
#pragma unroll 4
            for (int c4 = 0; c4 < 16; ++c4) {
                const float r0 = rp[4 * c4], r1 = rp[4 * c4 + 1], r2 = rp[4 * c4 + 2], r3 = rp[4 * c4 + 3];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    if (jj >= nq) break;
                    const float4 cv = *reinterpret_cast<const float4*>(cen[cur] + (j0 + jj) * KS_CP + 4 * c4);
                    float df = r0 - cv.x; d[jj] = fmaf(df, df, d[jj]);
                    df = r1 - cv.y; d[jj] = fmaf(df, df, d[jj]);
                    df = r2 - cv.z; d[jj] = fmaf(df, df, d[jj]);
                    df = r3 - cv.w; d[jj] = fmaf(df, df, d[jj]);
                }
            }
            float best = INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                if (jj < nq && d[jj] < best) { best = d[jj]; bi = j0 + jj; }
#pragma unroll
            for (int sft = 1; sft < 4; sft <<= 1) {
                const float od = __shfl_xor(best, sft); const int oj = __shfl_xor(bi, sft);
                if (od < best || (od == best && oj < bi)) { best = od; bi = oj; }
            }
            if (q == 0 && t < rows) { asg[tl][t] = bi; st_i(assign + base + t, bi); }
        }
        __syncthreads();
        // ---- the member lists: a counting sort of this workgroup's points by (cluster, point) ----
        int my_rank = 0, my_a = -1;
        const int blk_tl = wave >> 2, blk_tt = ((wave & 3) << 6) + lane;      // waves 0..7: one 64-point block each, lane = point
        if (wave < 8) {
            const int rows = blk_tl < ntl ? min(256, L - ((tile0 + blk_tl) << 8)) : 0;
            my_a = blk_tt < rows ? asg[blk_tl][blk_tt] : -1;
            for (int j = 0; j < K; ++j) {
                const unsigned long long mk = __ballot(my_a == j);
                if (lane == 0) cntblk[j][wave] = __popcll(mk);
                if (my_a == j) my_rank = __popcll(mk & ((1ull << lane) - 1ull));
            }
        }
        __syncthreads();
        if (wave == 0) {
            // lane j: its cluster's counts per block -> starts per block; the clusters' segments by a prefix sum over the lanes
            int c[8], tot = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) { c[b] = lane < K ? cntblk[lane][b] : 0; tot += c[b]; }
            int incl = tot;
#pragma unroll
            for (int sft = 1; sft < 32; sft <<= 1) { const int o = __shfl_up(incl, sft); if (lane >= sft) incl += o; }
            int run = incl - tot;
            if (lane < K) {
                seg[lane][0] = run; seg[lane][1] = tot;
#pragma unroll
                for (int b = 0; b < 8; ++b) { cntblk[lane][b] = run; run += c[b]; }
            }
        }
        __syncthreads();
        if (wave < 8 && my_a >= 0) order[cntblk[my_a][wave] + my_rank] = (blk_tl * 256 + blk_tt) * (KS_PITCH * 4);
        __syncthreads();
        // ---- the member sums: wave j takes cluster j over from workgroup g - 1 (polling its tagged words), adds this workgroup's members in
        // ascending order, hands on ----
        const char* dynb = reinterpret_cast<const char*>(dyn) + lane * 4;
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int j = wave + 16 * h;
            if (j >= K) break;
            float sum = 0.f; int m = 0;
            if (g > 0) {
                const u64 w = poll(gS + j * 64 + lane, true, ptag | ((unsigned)(g - 1) << 18), MASK_S);
                sum = __uint_as_float((unsigned)w);
                m = (int)((unsigned)(w >> 32) & 0x3ffffu);
            }
            const int st = __builtin_amdgcn_readfirstlane(seg[j][0]), mine = __builtin_amdgcn_readfirstlane(seg[j][1]);
            m += mine;
#pragma unroll 1
            for (int i0 = 0; i0 < mine; i0 += 64) {
                const int cc = min(64, mine - i0);                                  // (scalar)
                const int ov = lane < cc ? order[st + i0 + lane] : 0;               // ONE read: the next 64 members' row offsets, a lane each
                float v[2][16];
                // 16 rows in flight while the previous 16 are added: the offset comes out of lane `base + u` into a scalar register
                auto ld = [&](int buf, int base) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[buf][u] = *reinterpret_cast<const float*>(dynb + __builtin_amdgcn_readlane(ov, base + u));
                };
                ld(0, 0);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    if (gq * 16 >= cc) break;
                    if (gq < 3 && (gq + 1) * 16 < cc) ld((gq + 1) & 1, (gq + 1) * 16);
                    if ((gq + 1) * 16 <= cc) {
#pragma unroll
                        for (int u = 0; u < 16; ++u) sum += v[gq & 1][u];
                    } else {
#pragma unroll
                        for (int u = 0; u < 16; ++u) if (gq * 16 + u < cc) sum += v[gq & 1][u];
                    }
                }
            }
            if (!last_wg) st_u(gS + j * 64 + lane, __float_as_uint(sum), ptag | ((unsigned)g << 18) | (unsigned)m);
            else if (m > 0) {
                const float c = sum / (float)m;
                cen[nxt][j * KS_CP + lane] = c;
                const float sh = shift_of(c - cen[cur][j * KS_CP + lane]);
                if (lane == 0) { shift_part[j] = sh; cnt[j] = m; }
            } else if (lane == 0) { cnt[j] = 0; s_any_empty = 1; }
        }
        int stop = 0;
        if (!last_wg) {
            // the pass's centres, word by word as they arrive (thread 0's word decides for everybody: every word carries the stop bit)
            for (int u0 = 0; u0 < K * 64; u0 += 1024) {
                const int u = u0 + tid;
                if ((u0 + (wave << 6)) >= K * 64) break;            // (wave-uniform: K x 64 is a multiple of 64)
                const u64 w = poll(gC + u, true, ptag, MASK_C);
                cen[nxt][(u >> 6) * KS_CP + (u & 63)] = __uint_as_float((unsigned)w);
                if (u == 0) s_stop = (int)((unsigned)(w >> 32) & 1u);
            }
            __syncthreads();
            stop = s_stop;
        } else {
            __syncthreads();
            if (s_any_empty) {
                if (tid == 0) {
                    for (int j = 0; j < K; ++j)
                        if (cnt[j] == 0) {
                            const int e = s_events++;
                            const int r = (fallback && e < max_fallback) ? fallback[(size_t)img * max_fallback + e] : 0;
                            cnt[j] = -(r + 1);
                        }
                }
                __syncthreads();
                for (int j = wave; j < K; j += 16)
                    if (cnt[j] < 0) {
                        const float c = X[(size_t)(-cnt[j] - 1) * 64 + lane];
                        cen[nxt][j * KS_CP + lane] = c;
                        const float sh = shift_of(c - cen[cur][j * KS_CP + lane]);
                        if (lane == 0) shift_part[j] = sh;
                    }
                __syncthreads();
                if (tid == 0) s_any_empty = 0;
            }
            float sh = 0.f;
            for (int j = 0; j < K; ++j) sh += shift_part[j];
            stop = (sh * sh < 1e-4f) || p >= 20;
            for (int u = tid; u < K * 64; u += 1024) st_u(gC + u, __float_as_uint(cen[nxt][(u >> 6) * KS_CP + (u & 63)]), ptag | (unsigned)stop);
            __syncthreads();                    // (shift_part / cnt / s_any_empty are rewritten in the next pass)
        }
        ++passes;
        cur = nxt;
        if (stop) break;
    }
    // ---- the end of the run: every workgroup's last assignments have reached memory before the last one reads them ----
    if (!last_wg) {
        __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): a store counts until it is acknowledged
        __syncthreads();
        if (tid == 0) st_i(&ctl->done[g], 1);
        return;
    }
    if (tid < G - 1) {
        while (ld_i(&ctl->done[tid]) == 0) {
            if (__builtin_amdgcn_s_memtime() > deadline) __builtin_trap();
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    // ---- anchors and the hint mask of the image ----
    const float* sz = sizes + (size_t)img * L;
    float* hm = hint_mask + (size_t)img * L;
    for (int j = wave; j < K; j += 16) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int tt = lane; tt < L; tt += 64) {
            const float scv = add_rn(ld_i(assign + tt) == j ? 1.f : 0.f, mul_rn(sz[tt], 0.01f));
            if (scv > bv) { bv = scv; bi = tt; }
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const float ov = __shfl_xor(bv, sft); const int oi = __shfl_xor(bi, sft);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { anchor_out[img * K + j] = bi; s_anchor[j] = bi; }
    }
    for (int tt = tid; tt < L; tt += 1024) hm[tt] = 0.f;
    __syncthreads();
    if (tid == 0) {
        for (int j = 0; j < K; ++j) hm[s_anchor[j]] += 1.f;
        if (info) { info[img * 2] = passes; info[img * 2 + 1] = s_events; }
    }
}

// ---- k-means + anchors, fallback for more than KM_LIST_TOKENS points: one workgroup (256 threads) per image --------
// The token matrix (L x 64 fp32) is staged once in LDS (row pitch 65 floats: conflict-free row-per-thread reads)
// when it fits (L <= KM_LDS_TOKENS); larger images (no_resize path) read it from L2 with unconditional,
// pipelined loads.  Summation orders are fixed (ascending token index), so results are run-to-run deterministic.
template <bool XLDS>
__global__ __launch_bounds__(256) void kmeans_anchor_scan_kernel(const float* __restrict__ x, int D, long img_stride, int t_stride,
                                                            int c_stride, const float* __restrict__ sizes,
                                                            const int32_t* __restrict__ init_idx,
                                                            const int32_t* __restrict__ fallback, int max_fallback,
                                                            int32_t* assign_out, int32_t* anchor_out, float* hint_mask,
                                                            int32_t* info, int L, int K) {
    // point t, feature c of image img: x[img*img_stride + t*t_stride + c*c_stride]; D <= 64 features.
    // (L,64) token rows: t_stride 64, c_stride 1;  NCHW (2,L) colours (validation forward): t_stride 1, c_stride L
    extern __shared__ float dyn[];          // XLDS: [L][D+1] points, then [L] assignments (as int)
    __shared__ float cen[KMAX * 64];
    __shared__ float cnew[KMAX * 64];
    __shared__ int cnt[KMAX];
    __shared__ float shift_part[KMAX];
    __shared__ int s_events, s_stop;
    __shared__ float red_v[256];
    __shared__ int red_i[256];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int pitch = D + 1;                // odd for D = 64 and D = 2: conflict-free row-per-thread reads
    const float* X = x + (size_t)img * img_stride;
    float* xs = dyn;
    int* asg = reinterpret_cast<int*>(dyn + (XLDS ? L * pitch : 0));
    int32_t* assign = assign_out + (size_t)img * L;
    auto xg = [&](int t, int c) -> float { return X[(size_t)t * t_stride + (size_t)c * c_stride]; };
    if (XLDS)
        for (int u = tid; u < L * D; u += 256) { const int t = c_stride == 1 ? u / D : u % L, c = c_stride == 1 ? u % D : u / L; xs[t * pitch + c] = xg(t, c); }
    for (int u = tid; u < K * D; u += 256) cen[(u / D) * 64 + (u % D)] = xg(init_idx[img * K + (u / D)], u % D);
    if (tid == 0) { s_events = 0; s_stop = 0; }
    __syncthreads();
    auto xat = [&](int t, int c) -> float { return XLDS ? xs[t * pitch + c] : xg(t, c); };
    int passes = 0;
    while (true) {
        // assignment: first minimum of sum_c (x - c)^2
        for (int t = tid; t < L; t += 256) {
            float best = INFINITY; int bi = 0;
            for (int j = 0; j < K; ++j) {
                float d = 0.f;
                if (D == 64) {
#pragma unroll 16
                    for (int c = 0; c < 64; ++c) { const float df = xat(t, c) - cen[j * 64 + c]; d = fmaf(df, df, d); }
                } else {    // few features: plain mul + add like the reference's ((A-B)**2).sum(-1) (clusterkit.py:253-269)
                    for (int c = 0; c < D; ++c) { const float df = xat(t, c) - cen[j * 64 + c]; d = add_rn(d, mul_rn(df, df)); }
                }
                if (d < best) { best = d; bi = j; }
            }
            asg[t] = bi;
        }
        if (tid < K) cnt[tid] = 0;
        __syncthreads();
        for (int t = tid; t < L; t += 256) atomicAdd(&cnt[asg[t]], 1);
        __syncthreads();
        // empty clusters take a fallback row, in cluster order (sequential bookkeeping by one thread)
        if (tid == 0) {
            for (int j = 0; j < K; ++j)
                if (cnt[j] == 0) {
                    const int e = s_events++;
                    const int row = (fallback && e < max_fallback) ? fallback[(size_t)img * max_fallback + e] : 0;
                    cnt[j] = -(row + 1);   // marker: negative = use row
                }
        }
        __syncthreads();
        // update: thread = (cluster, channel); unconditional loads so they pipeline, ascending-token sum order
        for (int u = tid; u < K * D; u += 256) {
            const int j = u / D, c = u % D;
            float s;
            if (cnt[j] < 0) s = xg(-cnt[j] - 1, c);
            else {
                s = 0.f;
#pragma unroll 8
                for (int t = 0; t < L; ++t) { const float v = xat(t, c); s += (asg[t] == j) ? v : 0.f; }
                s = s / (float)cnt[j];
            }
            cnew[j * 64 + c] = s;
        }
        __syncthreads();
        // centre shift = sum_j sqrt(sum_c (new-old)^2)
        if (tid < K) {
            float q = 0.f;
            for (int c = 0; c < D; ++c) { const float d = cnew[tid * 64 + c] - cen[tid * 64 + c]; q = __builtin_fmaf(d, d, q); }
            shift_part[tid] = sqrtf(q);
        }
        __syncthreads();
        ++passes;
        if (tid == 0) {
            float sh = 0.f;
            for (int j = 0; j < K; ++j) sh += shift_part[j];
            s_stop = (sh * sh < 1e-4f) || passes >= 20;
        }
        for (int u = tid; u < K * D; u += 256) cen[(u / D) * 64 + (u % D)] = cnew[(u / D) * 64 + (u % D)];
        __syncthreads();
        if (s_stop) break;
    }
    for (int t = tid; t < L; t += 256) assign[t] = asg[t];
    // anchors: per cluster the first argmax of [assign==j] + sizes*0.01 (exact fp32 ops, no fma)
    const float* sz = sizes + (size_t)img * L;
    float* hm = hint_mask + (size_t)img * L;
    for (int t = tid; t < L; t += 256) hm[t] = 0.f;
    __syncthreads();
    for (int j = 0; j < K; ++j) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int t = tid; t < L; t += 256) {
            const float sc = add_rn(asg[t] == j ? 1.f : 0.f, mul_rn(sz[t], 0.01f));
            if (sc > bv) { bv = sc; bi = t; }   // ascending t: keeps the first maximum
        }
        red_v[tid] = bv; red_i[tid] = bi;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                const float ov = red_v[tid + s]; const int oi = red_i[tid + s];
                if (ov > red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])) { red_v[tid] = ov; red_i[tid] = oi; }
            }
            __syncthreads();
        }
        if (tid == 0) { anchor_out[img * K + j] = red_i[0]; hm[red_i[0]] += 1.f; }
        __syncthreads();
    }
    if (tid == 0 && info) { info[img * 2] = passes; info[img * 2 + 1] = s_events; }
}

__global__ void hint_mask_from_pos_kernel(const int32_t* pos, float* hint_mask, int n, int L, int K) {
    const int img = blockIdx.x;
    for (int t = threadIdx.x; t < L; t += blockDim.x) hint_mask[(size_t)img * L + t] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int j = 0; j < K; ++j) hint_mask[(size_t)img * L + pos[img * K + j]] = 1.f;
}

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
                         void* ws, hipStream_t s, const std::function<void(const void*, size_t)>* dbg, const float* packed) {
    const int T = n * l;
    const bool tail = packed != nullptr && T <= encoder_tail_max_rows();
    float* qkv = reinterpret_cast<float*>(ws);
    float* att = qkv + (size_t)3 * T * 64;
    float* pp[2] = {att + (size_t)T * 64, att + (size_t)2 * T * 64};
    const float* cur = x;
    static const int dbg_layers = [] { const char* e = std::getenv("DISCO_ENC_DEBUG_LAYERS"); return e ? atoi(e) : ENC_LAYERS; }();
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
        static const bool dbg_noqkv = std::getenv("DISCO_TAIL_NOQKV") != nullptr;      // bisecting aid: the tail kernel without its fused in-projection
        if (!tail || layer == 0 || dbg_noqkv) {           // (the tail path: layers 1.. get their q, k, v from the previous layer's tail kernel)
            int rc = launch_gemm<EPI_QKV>(g, s);
            if (rc) return rc;
        }
        if (dbg) (*dbg)(qkv, (size_t)3 * T * 64 * 4);
        // from attention_mfma_min_tokens() tokens on: both contractions on the matrix cores.  The choice depends on the token count alone.
        if (l >= attention_mfma_min_tokens()) {
            const int rc = launch_attention_mfma(qkv, qkv + (size_t)T * 64, qkv + (size_t)2 * T * 64, att, n, l, s);
            if (rc) return rc;
        } else
        // (beyond one workgroup per CU the two forms run the same: n = 2 ... 16 images measured with the threshold at 1x, 2x, 5x, 9x the CU count)
        if ((long)cdiv(l, 64) * N_HEAD * n < num_cus_current())
            hipLaunchKernelGGL(attention_kernel<1>, dim3(cdiv(l, 16), N_HEAD, n), dim3(256), 0, s, qkv, qkv + (size_t)T * 64,
                               qkv + (size_t)2 * T * 64, att, l);
        else
            hipLaunchKernelGGL(attention_kernel<4>, dim3(cdiv(l, 64), N_HEAD, n), dim3(256), 0, s, qkv, qkv + (size_t)T * 64,
                               qkv + (size_t)2 * T * 64, att, l);
        DISCO_LAUNCH_CHECK("attention_kernel");
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

size_t kmeans_ws_bytes(int n, int l) { return l > 512 ? (size_t)n * KC_IMG_BYTES : 0; }

int launch_kmeans_anchors(const float* x, const float* sizes, const int32_t* init_idx, const int32_t* fallback_rows,
                          int max_fallback, int32_t* assign, int32_t* anchor, float* hint_mask, int32_t* info, int n,
                          int l, int k, hipStream_t s, int d, int channel_major, void* ws, size_t ws_bytes) {
    if (k < 1 || k > KMAX) { set_error("kmeans: K=%d outside [1,%d]", k, KMAX); return DISCO_ESHAPE; }
    if (k > l) { set_error("kmeans: K=%d larger than %d tokens", k, l); return DISCO_ESHAPE; }
    if (d < 1 || d > 64) { set_error("kmeans: %d features outside [1,64]", d); return DISCO_ESHAPE; }
    const long img_stride = (long)l * d;
    const int t_stride = channel_major ? 1 : d, c_stride = channel_major ? l : 1;
    const int nseg = (l + 63) / 64;
    const size_t lists = ((size_t)2 * l + (size_t)nseg * k) * sizeof(int);
    const size_t best = (size_t)4 * 256 * (sizeof(float) + sizeof(int));
    constexpr int MAX_SMEM = 128 * 1024;      // dynamic part; the kernels hold up to 27 KB of static LDS besides
    // the dynamic-LDS limit is a per-device attribute of each kernel: the three instantiations share one function-pointer
    // type (so one lambda body), hence the table is keyed by variant index, not by a static inside the lambda
    auto launch = [&](auto kern, int variant, size_t smem) -> int {
        static std::atomic<int> attr_done[3][DISCO_MAX_DEVICES];
        DISCO_HIP_CHECK(set_dyn_lds_once(attr_done[variant], reinterpret_cast<const void*>(kern), MAX_SMEM));
        hipLaunchKernelGGL(kern, dim3(n), dim3(1024), smem, s, x, d, img_stride, t_stride, c_stride, sizes, init_idx,
                           fallback_rows, max_fallback, assign, anchor, hint_mask, info, l, k);
        return DISCO_OK;
    };
    const size_t tile = (size_t)256 * (d + 1) * sizeof(float);
    const size_t glist_smem = tile + (size_t)nseg * k * sizeof(int) + best;
    int rc = DISCO_OK;
    // DISCO_KMEANS_V1=1: the general kernel at every size (A/B runs; results are bit-identical)
    static const bool small_ok = [] { const char* e = std::getenv("DISCO_KMEANS_V1"); return !(e && e[0] == '1'); }();
    static const bool coop_ok = [] { const char* e = std::getenv("DISCO_KMEANS_COOP"); return !(e && e[0] == '0'); }();      // 0: one workgroup per image at every size
    if (small_ok && l <= KS_MAXL && d == 64 && !channel_major) {
        const size_t smem = (size_t)l * KS_PITCH * sizeof(float);
        static std::atomic<int> small_done[DISCO_MAX_DEVICES];
        DISCO_HIP_CHECK(set_dyn_lds_once(small_done, reinterpret_cast<const void*>(kmeans_small_kernel), MAX_SMEM));
        hipLaunchKernelGGL(kmeans_small_kernel, dim3(n), dim3(1024), smem, s, x, sizes, init_idx, fallback_rows, max_fallback, assign, anchor,
                           hint_mask, info, l, k);
    } else if (small_ok && d == 64 && !channel_major && l > 512 && ws && ws_bytes >= kmeans_ws_bytes(n, l) && cdiv(l, 512) <= KC_MAXG && l < KC_MAX_POINTS &&
               (long)n * cdiv(l, 512) <= num_cus_current() / 4 && coop_ok) {
        // several workgroups per image, all of them resident (they wait for each other): a quarter of the CUs at most, so that the launches
        // of up to four concurrent forwards (runner.py pipelines two) always fit side by side; the exchange area starts at zero
        const int G = cdiv(l, 512);
        DISCO_HIP_CHECK(hipMemsetAsync(ws, 0, kmeans_ws_bytes(n, l), s));
        const size_t smem = (size_t)2 * 256 * KS_PITCH * sizeof(float);
        static std::atomic<int> coop_done[DISCO_MAX_DEVICES];
        DISCO_HIP_CHECK(set_dyn_lds_once(coop_done, reinterpret_cast<const void*>(kmeans_coop_kernel), MAX_SMEM));
        hipLaunchKernelGGL(kmeans_coop_kernel, dim3(n * G), dim3(1024), smem, s, x, sizes, init_idx, fallback_rows, max_fallback, assign, anchor,
                           hint_mask, info, l, k, G, static_cast<unsigned char*>(ws));
    } else if (small_ok && d == 64 && !channel_major) {
        const size_t smem = (size_t)2 * 256 * KS_PITCH * sizeof(float);
        static std::atomic<int> tiled_done[DISCO_MAX_DEVICES];
        DISCO_HIP_CHECK(set_dyn_lds_once(tiled_done, reinterpret_cast<const void*>(kmeans_tiled_kernel), MAX_SMEM));
        hipLaunchKernelGGL(kmeans_tiled_kernel, dim3(n), dim3(1024), smem, s, x, sizes, init_idx, fallback_rows, max_fallback, assign, anchor,
                           hint_mask, info, l, k);
    } else
    if (l <= KM_LDS_TOKENS) rc = launch(kmeans_anchor_kernel<true, false>, 0, (size_t)l * (d + 1) * sizeof(float) + lists + best);
    else if (l <= KM_LIST_TOKENS) rc = launch(kmeans_anchor_kernel<false, false>, 1, tile + lists + best);
    else if (glist_smem <= (size_t)MAX_SMEM) rc = launch(kmeans_anchor_kernel<false, true>, 2, glist_smem);
    else {
        // scan fallback: one int of LDS per token on top of the kernel's static arrays
        const size_t scan_smem = (size_t)l * sizeof(int);
        if (scan_smem > (size_t)MAX_SMEM) { set_error("kmeans: %d tokens exceed what one workgroup can index in LDS (%d)", l, MAX_SMEM / 4); return DISCO_ESHAPE; }
        static std::atomic<int> scan_done[DISCO_MAX_DEVICES];
        DISCO_HIP_CHECK(set_dyn_lds_once(scan_done, reinterpret_cast<const void*>(kmeans_anchor_scan_kernel<false>), MAX_SMEM));
        hipLaunchKernelGGL(kmeans_anchor_scan_kernel<false>, dim3(n), dim3(256), scan_smem, s, x, d, img_stride,
                           t_stride, c_stride, sizes, init_idx, fallback_rows, max_fallback, assign, anchor, hint_mask, info, l, k);
    }
    if (rc) return rc;
    DISCO_LAUNCH_CHECK("kmeans_anchor_kernel");
    return DISCO_OK;
}

int launch_hint_mask_from_pos(const int32_t* pos, float* hint_mask, int n, int l, int k, hipStream_t s) {
    hipLaunchKernelGGL(hint_mask_from_pos_kernel, dim3(n), dim3(256), 0, s, pos, hint_mask, n, l, k);
    DISCO_LAUNCH_CHECK("hint_mask_from_pos_kernel");
    return DISCO_OK;
}

}  // namespace disco
