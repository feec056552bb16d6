// Does the ORDER in which a wave visits its accumulators change the sustained (power-limited) MFMA rate?  Registers-only loops,
// random operands, 4 accumulators per wave, 2 waves per SIMD:  chain length L = consecutive MFMAs on one accumulator before
// moving to the next.  hipcc --offload-arch=gfx950 -O3 tools/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// KIND 0: f16 only (48 MFMAs per iteration).  KIND 1: per accumulator visit 2 f16 + ... the mx kernel's mix: H visits (L f16) and Q visits (L/2 fp8)
template <int L, int KIND>
__global__ __launch_bounds__(512) void k(const f16x8* __restrict__ ops, const i32x8* __restrict__ ops8, float* out, unsigned long long* clk, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = ops[i * 64 + lane]; b[i] = ops[(4 + i) * 64 + lane]; }
    i32x8 a8[2], b8[2];
    for (int i = 0; i < 2; ++i) { a8[i] = ops8[i * 64 + lane]; b8[i] = ops8[(2 + i) * 64 + lane]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rnd = 0; rnd < 48 / (4 * L); ++rnd)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < L; ++j) {
                    if (KIND == 0 || ((rnd & 1) == 0)) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[(i + j) & 3], a[(j + rnd) & 3], acc[i], 0, 0, 0);
                    else if (j % 2 == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j & 1], a8[(i + rnd) & 1], acc[i], 0, 0, 0, 0, 0, 0);
                }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[0] = s;
}
static float gauss() { float u = 0.f; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.f; }
template <int L, int KIND> void run(const f16x8* d, const i32x8* d8, float* o, unsigned long long* c, int cus) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 6000, blocks = cus * 4;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<L, KIND>), dim3(blocks), dim3(512), 0, 0, d, d8, o, c, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long clk[2]; hipMemcpy(clk, c, 16, hipMemcpyDeviceToHost);
        // pipe units per iteration per wave: KIND 0: 48; KIND 1: half the rounds are f16 (4*L per round), half fp8 (4*L/2 MFMAs of 2 units) -> 48 as well
        const double units = (double)blocks * 8 * iters * 48;
        if (rep) printf("%s chain %2d: %7.3f ms  %6.1f G units/s  clk %.3f GHz\n", KIND ? "f16+fp8 (alternating H/Q rounds)" : "f16 only", L, ms, units / ms * 1e-6, (double)clk[0] / clk[1] * 0.1);
    }
}
int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<_Float16> h(8 * 64 * 8); std::vector<unsigned char> h8(4 * 64 * 32);
    for (auto& v : h) v = (_Float16)gauss();
    for (auto& v : h8) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v &= 0xfe; }
    f16x8* d; i32x8* d8; float* o; unsigned long long* c;
    hipMalloc(&d, h.size() * 2); hipMalloc(&d8, h8.size()); hipMalloc(&o, 4); hipMalloc(&c, 16);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(d8, h8.data(), h8.size(), hipMemcpyHostToDevice);
    run<1, 0>(d, d8, o, c, cus); run<3, 0>(d, d8, o, c, cus); run<6, 0>(d, d8, o, c, cus); run<12, 0>(d, d8, o, c, cus);
    run<2, 1>(d, d8, o, c, cus); run<6, 1>(d, d8, o, c, cus); run<12, 1>(d, d8, o, c, cus);
    return 0;
}
