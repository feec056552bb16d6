// mfma_f32_shapes_probe.hip - are v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 the SAME arithmetic (an fmaf chain over ascending k)?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f32_shapes_probe.hip -o tools/build/mfma_f32_shapes_probe && tools/build/mfma_f32_shapes_probe
// C[m][n] = sum_k A[m][k] B[n][k], K = 64, computed (1) on the host as fmaf(a, b, acc) over k = 0..63, (2) with 32 x 32x32x2 MFMAs,
// (3) with 16 x 16x16x4 MFMAs; compared bit for bit on random data incl. wide dynamic range.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 64;
__global__ void k32(const float* A, const float* B, float* C) {       // one wave: C 32 x 32
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(lane & 31) * K + k + (lane >> 5)], B[(lane & 31) * K + k + (lane >> 5)], acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[e];
}
__global__ void k16(const float* A, const float* B, float* C) {       // one wave: the four 16 x 16 blocks of C 32 x 32
    const int lane = threadIdx.x;
    for (int bm = 0; bm < 2; ++bm)
        for (int bn = 0; bn < 2; ++bn) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < K; k += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(bm * 16 + (lane & 15)) * K + k + (lane >> 4)], B[(bn * 16 + (lane & 15)) * K + k + (lane >> 4)], acc, 0, 0, 0);
            for (int e = 0; e < 4; ++e) C[(bm * 16 + 4 * (lane >> 4) + e) * 32 + bn * 16 + (lane & 15)] = acc[e];
        }
}
int main() {
    std::vector<float> A(32 * K), B(32 * K), C0(1024), C1(1024), C2(1024);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    long bad32 = 0, bad16 = 0, total = 0;
    srand(1);
    for (int rep = 0; rep < 200; ++rep) {
        for (auto* v : {&A, &B})
            for (auto& x : *v) { const float u = (float)rand() / RAND_MAX * 2.f - 1.f; x = rep % 2 ? u * std::ldexp(1.f, rand() % 24 - 12) : u; }
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) { float acc = 0.f; for (int k = 0; k < K; ++k) acc = fmaf(A[m * K + k], B[n * K + k], acc); C0[m * 32 + n] = acc; }
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC); hipMemcpy(C1.data(), dC, 4096, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC); hipMemcpy(C2.data(), dC, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 1024; ++i) { bad32 += memcmp(&C0[i], &C1[i], 4) != 0; bad16 += memcmp(&C0[i], &C2[i], 4) != 0; ++total; }
    }
    printf("%ld results each: 32x32x2 differs from the host fmaf chain in %ld, 16x16x4 in %ld\n", total, bad32, bad16);
    return (bad32 || bad16) ? 1 : 0;
}
