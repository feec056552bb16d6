// kmeans.hip - clusterkit's Lloyd k-means + the per-cluster anchors (clusterkit.py:31-58,99-208,253-269; anchor_gen.py:92-107) and the
// random-hint mask (basic.py:42-47).  Five kernels, ONE summation order (ascending points per (cluster, feature) chain), so every path
// gives bit-identical assignments, pass counts and events: kmeans_small_kernel (<= 256 points), kmeans_tiled_kernel (one workgroup per
// image, tiles streamed), kmeans_coop_kernel (> 512 points on several workgroups per image), kmeans_anchor_kernel / _scan_kernel (general:
// other feature counts, channel-major input).  Split out of tokens.hip in round 6.
#include <cmath>
#include <cstdlib>
#include <vector>
#include <mutex>
#include "common.h"

namespace disco {

namespace {

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
// kmeans_coop_kernel's admission word (one per image, in its scratch): workgroups arrived so far | KC_ABORT once the image has been given
// up over there (kmeans_tiled_kernel, launched behind it, then computes the image)
constexpr int KC_ABORT = 1 << 30;
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
                                                            int32_t* info, int L, int K, const int* coop_state, long coop_stride,
                                                            unsigned int* fallback_counter) {
    extern __shared__ float dyn[];          // 2 x [256][KS_PITCH]: the tile being worked on and the one being written
    __shared__ __attribute__((aligned(16))) float cen[2][KMAX * KS_CP];
    __shared__ int asg[2][KS_MAXL];         // the tiles' assignments (double-buffered like the tiles)
    __shared__ int cnt[KMAX];
    __shared__ float shift_part[KMAX];
    __shared__ int s_anchor[KMAX];
    __shared__ int s_events, s_any_empty;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // launched BEHIND kmeans_coop_kernel as its safety net (coop_state != null): this image runs here only if its workgroups over there
    // could not be admitted together (or gave up): the same additions in the same order, so the result is the one they would have produced
    if (coop_state) {
        if (!(coop_state[(size_t)img * coop_stride] & KC_ABORT)) return;
        if (tid == 0 && fallback_counter) atomicAdd(fallback_counter, 1u);
    }
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
// The workgroups of an image spin on each other's words, so they must all be RESIDENT TOGETHER.  The launcher takes this kernel only while
// n x G fits a quarter of the CUs - but that is a heuristic, not a guarantee (other streams' persistent conv workgroups, other contexts or
// processes on the GPU, CU masks), so residency is PROVEN before anything is exchanged (round 6; the first form trusted the heuristic and
// trapped after 5e9 cycles - a sticky hipErrorLaunchFailure that ends a serving process):
//   admission   every workgroup adds itself to the image's `state` word and waits until all G have arrived = all G hold a CU at the same
//               time, and workgroups are never descheduled: from then on every spin below ends.  A workgroup that waits longer than
//               KC_ADMIT_CYCLES gives the IMAGE up: it CASes KC_ABORT into the word (possible only while the count is below G: an image is
//               either committed or aborted, never both); late arrivals see the bit in the value their add returns and leave at once - CUs
//               held by half-arrived images are released, which is what unblocks a set of launches that got in each other's way.
//   safety net  kmeans_tiled_kernel is launched right behind, gated per image on KC_ABORT: an image that was not admitted is computed
//               there, on one workgroup - bit-identical by construction, no host involvement, no error, a few hundred microseconds lost.
//   backstop    the spins of an ADMITTED image still carry a deadline (1e9 cycles; an image takes < 10 ms): running into it can only mean a
//               bug or a hardware fault, and it degrades the same way - the wave sets KC_ABORT, stops waiting (every later poll returns at
//               once, every loop is bounded by the 20-pass limit) and the safety net recomputes the image.
// DISCO_KMEANS_COOP_INJECT (tests/test_gpu_ops.py): 1 = workgroup 0 of every image never arrives (the others time out in admission);
// 2 = workgroup 0 leaves right after admission (the others run into the backstop).
constexpr int KC_MAXG = 64;
constexpr unsigned long long KC_ADMIT_CYCLES = 6000000ull;        // 2.5-5 ms of shader clock
constexpr unsigned long long KC_BACKSTOP_CYCLES = 1000000000ull;  // 0.4-0.8 s
struct KmCoopCtl { int done[KC_MAXG]; int state; int pad[3]; };
// per image: the running member sums and the pass's new centres, [KMAX][64] 8-byte words {value, tag} each, + the flags
constexpr size_t KC_IMG_BYTES = ((size_t)(2 * KMAX * 64) * 8 + sizeof(KmCoopCtl) + 255) & ~(size_t)255;
constexpr int KC_MAX_POINTS = 1 << 18;
__global__ __launch_bounds__(1024) void kmeans_coop_kernel(const float* __restrict__ x, const float* __restrict__ sizes,
                                                           const int32_t* __restrict__ init_idx, const int32_t* __restrict__ fallback,
                                                           int max_fallback, int32_t* assign_out, int32_t* anchor_out, float* hint_mask,
                                                           int32_t* info, int L, int K, int G, unsigned char* scratch, int inject) {
    extern __shared__ float dyn[];          // 2 x [256][KS_PITCH]: this workgroup's two tiles, resident
    __shared__ __attribute__((aligned(16))) float cen[2][KMAX * KS_CP];
    __shared__ int asg[2][KS_MAXL];
    __shared__ int cnt[KMAX];
    __shared__ float shift_part[KMAX];
    __shared__ int s_anchor[KMAX];
    __shared__ int s_events, s_any_empty, s_stop, s_admit;
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
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
    const unsigned long long deadline = t_entry + KC_BACKSTOP_CYCLES;
    // ---- admission: thread 0 registers this workgroup and waits for the image's other workgroups (the others load the tiles meanwhile) ----
    if (tid == 0) {
        int ok = 0;
        if (!(inject == 1 && g == 0)) {
            const int before = __hip_atomic_fetch_add(&ctl->state, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (!(before & KC_ABORT)) {
                for (;;) {
                    int v = ld_i(&ctl->state);
                    if (v & KC_ABORT) break;
                    if (v == G) { ok = 1; break; }                  // committed: all G are here, and nobody can abort from G
                    if (__builtin_amdgcn_s_memtime() - t_entry > KC_ADMIT_CYCLES) {
                        // give the image up - unless it commits at this very moment (then the exchange fails and we look again)
                        if (__hip_atomic_compare_exchange_strong(&ctl->state, &v, v | KC_ABORT, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
                        continue;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        }
        s_admit = ok;
    }
    // this wave polls one word per lane until every taking-part lane's tag equals `want` under `mask`; a lane with !mine takes no part.
    // A wave that runs into the backstop deadline gives the image up (KC_ABORT: kmeans_tiled_kernel recomputes it) and stops waiting.
    bool gave_up = false;
    auto give_up = [&]() {
        if (!gave_up && lane == 0) __hip_atomic_fetch_or(&ctl->state, KC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        gave_up = true;
    };
    auto poll = [&](const u64* ptr, bool mine, unsigned want, unsigned mask) -> u64 {
        u64 w = 0;
        for (;;) {
            if (mine) w = __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const bool ok = !mine || (((unsigned)(w >> 32)) & mask) == want;
            if (__ballot(!ok) == 0ull) break;
            if (gave_up || __builtin_amdgcn_s_memtime() > deadline) { give_up(); break; }
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
    if (!s_admit) return;                       // the image was given up (here or by a sibling): nothing has been exchanged yet
    if (inject == 2 && g == 0) return;          // (fault injection: an admitted workgroup that vanishes)
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
            stop = s_stop | (p >= 20);          // (the last workgroup's stop bit says the same at pass 20; spelled out so that a wave that gave up ends too)
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
            if (__builtin_amdgcn_s_memtime() > deadline) {      // backstop: give the image up, kmeans_tiled_kernel recomputes it
                __hip_atomic_fetch_or(&ctl->state, KC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
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

}  // namespace

size_t kmeans_ws_bytes(int n, int l) { return l > 512 ? (size_t)n * KC_IMG_BYTES : 0; }

size_t kmeans_state_offset() { return (size_t)(2 * KMAX * 64) * 8 + offsetof(KmCoopCtl, state); }
size_t kmeans_image_stride() { return KC_IMG_BYTES; }

int launch_kmeans_anchors(const float* x, const float* sizes, const int32_t* init_idx, const int32_t* fallback_rows,
                          int max_fallback, int32_t* assign, int32_t* anchor, float* hint_mask, int32_t* info, int n,
                          int l, int k, hipStream_t s, int d, int channel_major, void* ws, size_t ws_bytes, unsigned int* fallback_counter) {
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
        const char* inj = std::getenv("DISCO_KMEANS_COOP_INJECT");          // fault injection (tests); read per launch, this path only
        hipLaunchKernelGGL(kmeans_coop_kernel, dim3(n * G), dim3(1024), smem, s, x, sizes, init_idx, fallback_rows, max_fallback, assign, anchor,
                           hint_mask, info, l, k, G, static_cast<unsigned char*>(ws), inj ? atoi(inj) : 0);
        DISCO_LAUNCH_CHECK("kmeans_coop_kernel");
        // the safety net, in stream order: an image whose workgroups were not admitted together (KC_ABORT in its state word) is computed
        // here on one workgroup; the others return at once (one workgroup per image reading one word: ~2 us of a >= 4 ms forward)
        static std::atomic<int> net_done[DISCO_MAX_DEVICES];
        DISCO_HIP_CHECK(set_dyn_lds_once(net_done, reinterpret_cast<const void*>(kmeans_tiled_kernel), MAX_SMEM));
        hipLaunchKernelGGL(kmeans_tiled_kernel, dim3(n), dim3(1024), smem, s, x, sizes, init_idx, fallback_rows, max_fallback, assign, anchor,
                           hint_mask, info, l, k, reinterpret_cast<const int*>(static_cast<unsigned char*>(ws) + kmeans_state_offset()),
                           (long)(KC_IMG_BYTES / sizeof(int)), fallback_counter);
    } else if (small_ok && d == 64 && !channel_major) {
        const size_t smem = (size_t)2 * 256 * KS_PITCH * sizeof(float);
        static std::atomic<int> tiled_done[DISCO_MAX_DEVICES];
        DISCO_HIP_CHECK(set_dyn_lds_once(tiled_done, reinterpret_cast<const void*>(kmeans_tiled_kernel), MAX_SMEM));
        hipLaunchKernelGGL(kmeans_tiled_kernel, dim3(n), dim3(1024), smem, s, x, sizes, init_idx, fallback_rows, max_fallback, assign, anchor,
                           hint_mask, info, l, k, (const int*)nullptr, 0L, (unsigned int*)nullptr);
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
