// mfma_4x4x1_probe.hip - the operand layout and the issue rate of v_mfma_f32_4x4x1_16B_f32 on gfx950, and the rate of v_exp_f32:
// what attention_mfma_kernel (csrc/tokens.hip) is built on.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_4x4x1_probe.hip -o tools/build/mfma_4x4x1_probe && tools/build/mfma_4x4x1_probe
// Layout claim (16 blocks b = lane / 4 of 4 x 4 x 1):  A_b[i] = a of lane 4 b + i,  B_b[j] = b of lane 4 b + j,
//   D_b[i][j] = register i of lane 4 b + j.   Checked with random operands against the host product, bit for bit (one product + one add).
// Rates: cycles per instruction of back-to-back MFMAs on 1 / 2 / 4 accumulators (s_memtime around 4096 instructions, one wave),
//   and of v_exp_f32 (4 dependent chains: a latency figure, an upper bound of the issue cost).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void layout_kernel(const float* a, const float* b, const float* c, float* d) {
    const int lane = threadIdx.x;
    f32x4 acc = {c[lane * 4], c[lane * 4 + 1], c[lane * 4 + 2], c[lane * 4 + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[lane * 4 + i] = acc[i];
}

// S^T = K Q^T on 32x32x2 (rows = keys, columns = queries), P = S^T in place as the A operand of 16 x 2 4x4x1 MFMAs against V: O = P V
// for 32 queries x 32 keys x 8 dims - the data flow of one tile of attention_mfma_kernel, checked against the host
__global__ void tile_kernel(const float* Q, const float* K, const float* V, float* O) {     // Q, K, V: [32][8]; O: [32][8]
    const int lane = threadIdx.x, qn = lane & 31, hk = lane >> 5, j4 = lane & 3;
    f32x16 s;
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(K[qn * 8 + 2 * j + hk], Q[qn * 8 + 2 * j + hk], s, 0, 0, 0);
    f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int r = 0; r < 16; ++r) {
        const int key = 8 * (r >> 2) + 4 * hk + (r & 3);
        for (int h = 0; h < 2; ++h) o[h] = __builtin_amdgcn_mfma_f32_4x4x1f32(s[r], V[key * 8 + 4 * h + j4], o[h], 0, 0, 0);
    }
    // lane (b = lane / 4: g = b % 8, half = b / 8; j4) holds O[query 4 g + i][4 h + j4] summed over the keys of its half: add the halves
    for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 4; ++i) {
            const float t = o[h][i] + __shfl_xor(o[h][i], 32);
            if (hk == 0) O[(4 * ((lane >> 2) & 7) + i) * 8 + 4 * h + j4] = t;
        }
}

template <int CHAINS>
__global__ void rate_kernel(float* out, unsigned long long* cyc, float x) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{x, x, x, x};
    const float a = x * 0.5f, b = x * 0.25f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 4096 / (CHAINS * 8); ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
    }
    float sink = 0.f;
    for (int c = 0; c < CHAINS; ++c) sink += acc[c][0] + acc[c][3];
    asm volatile("" : "+v"(sink));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = sink;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void exp_rate_kernel(float* out, unsigned long long* cyc, float x) {
    float v[4] = {x, x * 0.5f, x * 0.25f, x * 0.125f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_exp2f(v[c]);
    }
    float sink = v[0] + v[1] + v[2] + v[3];
    asm volatile("" : "+v"(sink));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = sink;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    std::vector<float> a(64), b(64), c(256), d(256);
    float *da, *db, *dc, *dd;
    unsigned long long* dcyc;
    hipMalloc(&da, 4096); hipMalloc(&db, 4096); hipMalloc(&dc, 4096); hipMalloc(&dd, 4096); hipMalloc(&dcyc, 64);
    srand(3);
    long bad = 0, total = 0;
    for (int rep = 0; rep < 100; ++rep) {
        for (auto& x : a) x = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& x : b) x = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& x : c) x = (float)rand() / RAND_MAX * 2.f - 1.f;
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        for (int blk = 0; blk < 16; ++blk)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const float want = fmaf(a[4 * blk + i], b[4 * blk + j], c[(4 * blk + j) * 4 + i]);
                    bad += memcmp(&want, &d[(4 * blk + j) * 4 + i], 4) != 0; ++total;
                }
    }
    printf("4x4x1 layout: %ld of %ld elements differ from fmaf(A_b[i], B_b[j], C) under the claimed layout\n", bad, total);

    std::vector<float> Q(256), K(256), V(256), O(256);
    double maxerr = 0;
    for (int rep = 0; rep < 20; ++rep) {
        for (auto* v : {&Q, &K, &V}) for (auto& x : *v) x = (float)rand() / RAND_MAX * 2.f - 1.f;
        hipMemcpy(da, Q.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, K.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dc, V.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(tile_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(O.data(), dd, 1024, hipMemcpyDeviceToHost);
        for (int qi = 0; qi < 32; ++qi)
            for (int dd_ = 0; dd_ < 8; ++dd_) {
                double want = 0;
                for (int key = 0; key < 32; ++key) {
                    double s = 0;
                    for (int e = 0; e < 8; ++e) s += (double)Q[qi * 8 + e] * K[key * 8 + e];
                    want += s * V[key * 8 + dd_];
                }
                maxerr = std::max(maxerr, std::fabs(want - O[qi * 8 + dd_]));
            }
    }
    printf("tile data flow (S^T accumulators as the A operand of the 4x4x1 P V product): max |error| %.3e (fp32 rounding: ~1e-6)\n", maxerr);

    float* dout; hipMalloc(&dout, 4096);
    unsigned long long cyc = 0;
    auto report = [&](const char* name, double n) {
        hipDeviceSynchronize();
        hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
        printf("%s: %.2f cycles per instruction (s_memtime ticks = shader cycles on gfx950)\n", name, (double)cyc / n);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rate_kernel<1>, dim3(1), dim3(64), 0, 0, dout, dcyc, 1e-3f); report("4x4x1, 1 accumulator ", 4096);
        hipLaunchKernelGGL(rate_kernel<2>, dim3(1), dim3(64), 0, 0, dout, dcyc, 1e-3f); report("4x4x1, 2 accumulators", 4096);
        hipLaunchKernelGGL(rate_kernel<4>, dim3(1), dim3(64), 0, 0, dout, dcyc, 1e-3f); report("4x4x1, 4 accumulators", 4096);
        hipLaunchKernelGGL(exp_rate_kernel, dim3(1), dim3(64), 0, 0, dout, dcyc, 0.5f); report("v_exp_f32, 4 chains   ", 4096);
    }
    return bad ? 1 : 0;
}
