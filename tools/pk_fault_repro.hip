// Reduced reproducer attempt for the concurrency fault of DESIGN.md section 4 (pool_partial_kernel's packed-fp32 path).
//   hipcc --offload-arch=gfx950 -O3 tools/pk_fault_repro.hip -o /tmp/pk_repro            (SLP-packed v_pk_fma_f32)
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/pk_fault_repro.hip -o /tmp/pk_repro_scalar
//   /tmp/pk_repro [bg_streams=7] [rounds=40] [bg_kind=0|1]
// One stream runs a kernel shaped like the 16-byte path of pool_partial_kernel (per 16x16 cell: 256 pixels x 64 fp16 hi/lo channels
// weighted by 9 probabilities from LDS, butterfly over the lanes, 4 waves combined through LDS) again and again while `bg_streams` other
// streams keep every CU busy with persistent 512-thread workgroups holding 150 KB of LDS (bg_kind 0: MFMA + ds_read/ds_write;
// 1: the same with buffer_load ... lds traffic, like the conv kernel's operand staging).  Every result is compared bit for bit with the
// run made alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int W = 256, H = 256, HW = W * H;

__global__ __launch_bounds__(256) void pool_like(const f16* __restrict__ feat, long plane, const float* __restrict__ prob, float mul, float* __restrict__ out, int ws) {
    extern __shared__ float sm[];
    float* sp_prob = sm;
    float* red = sm + 256 * 9;
    const int cell = blockIdx.x, cx = cell % ws, cy = (cell / ws) % ws, n = cell / (ws * ws);
    const float* pr_img = prob + (long)n * 9 * HW;
    for (int p = threadIdx.x; p < 256; p += 256) {
        const long off = (long)(cy * 16 + (p >> 4)) * W + cx * 16 + (p & 15);
        for (int c = 0; c < 9; ++c) sp_prob[p * 9 + c] = pr_img[c * HW + off];
    }
    __syncthreads();
    const int g = threadIdx.x >> 6, q = threadIdx.x & 7, r = threadIdx.x >> 3;
    const long cell0 = (long)(cy * 16) * W + cx * 16;
    const f16* s16 = feat + (((long)n * 4 + (q >> 1)) * HW + cell0) * 16 + (q & 1) * 8;
    float acc[9][8];
    for (int c = 0; c < 9; ++c) for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
        const int p = 32 * i + r;
        const int off = ((p >> 4) * W + (p & 15)) * 16;
        const f16x8 h = *reinterpret_cast<const f16x8*>(s16 + off);
        const f16x8 l = *reinterpret_cast<const f16x8*>(s16 + off + plane);
        float f[8], pr[9];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = ((float)h[j] + (float)l[j]) * mul;
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
            float v = acc[c][j];
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            acc[c][j] = v;
        }
    if ((threadIdx.x & 63) < 8) {
#pragma unroll
        for (int c = 0; c < 9; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(g * 9 + c) * 64 + q * 8 + j] = acc[c][j];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 9 * 64; o += 256) {
        const int c = o >> 6, ch = o & 63;
        out[((long)cell * 9 + c) * 64 + ch] = (red[(0 * 9 + c) * 64 + ch] + red[(1 * 9 + c) * 64 + ch]) + (red[(2 * 9 + c) * 64 + ch] + red[(3 * 9 + c) * 64 + ch]);
    }
}

template <int KIND>
__global__ __launch_bounds__(512, 2) void bg_kernel(const f16x8* __restrict__ ops, const char* __restrict__ stream_src, unsigned src_bytes, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem_bg[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a = ops[lane], b = ops[64 + lane];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f16x8* lds = reinterpret_cast<f16x8*>(smem_bg);
    for (int i = threadIdx.x; i < 150 * 1024 / 16; i += 512) lds[i] = a;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)stream_src, 0, src_bytes, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        if (KIND == 1) {
            // stream 8 KiB per wave per iteration into this wave's LDS slice through the LDS-DMA path
            typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
            for (int pce = 0; pce < 8; ++pce) {
                const unsigned off = ((unsigned)(blockIdx.x * 8 + wave) * 8192u + (unsigned)pce * 1024u + (unsigned)it * 65536u) % (src_bytes - 8192u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem_bg + wave * 16384 + pce * 1024), 16, lane * 16, off & ~15u, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f16x8 x = lds[(wave * 1024 + t * 64 + lane) & (150 * 64 - 1)];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(i & 1 ? b : x, i & 2 ? a : x, acc[i], 0, 0, 0);
        }
        if (KIND == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int nbg = argc > 1 ? atoi(argv[1]) : 7, rounds = argc > 2 ? atoi(argv[2]) : 40, kind = argc > 3 ? atoi(argv[3]) : 0;
    const int n = 8, ws = 16, cells = n * ws * ws;
    const size_t fe = (size_t)n * 64 * HW;
    std::vector<f16> hf(2 * fe);
    std::vector<float> hp((size_t)n * 9 * HW);
    unsigned seed = 1;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) / 16777216.f; };
    for (size_t i = 0; i < fe; ++i) { const float v = rnd() * 20.f; hf[i] = (f16)v; hf[fe + i] = (f16)(v - (float)hf[i]); }
    for (auto& v : hp) v = rnd() * 0.2f;
    f16* dfeat; float *dprob, *dsink; f16x8* dops; char* dsrc;
    CK(hipMalloc(&dfeat, hf.size() * 2)); CK(hipMemcpy(dfeat, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dprob, hp.size() * 4)); CK(hipMemcpy(dprob, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dsink, 4096)); CK(hipMalloc(&dops, 128 * 16)); CK(hipMemset(dops, 0x3c, 128 * 16));
    const unsigned src_bytes = 256u << 20;
    CK(hipMalloc(&dsrc, src_bytes)); CK(hipMemset(dsrc, 0x3c, src_bytes));
    const size_t ob = (size_t)cells * 9 * 64 * 4;
    const int reps = 30;
    std::vector<float*> douts(reps);
    for (auto& p : douts) CK(hipMalloc(&p, ob));
    float* dref; CK(hipMalloc(&dref, ob));
    const size_t smem = (256 * 9 + 4 * 9 * 64) * 4;
    hipStream_t ts; CK(hipStreamCreate(&ts));
    std::vector<hipStream_t> bs(nbg);
    for (auto& s : bs) CK(hipStreamCreate(&s));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bg_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bg_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipLaunchKernelGGL(pool_like, dim3(cells), dim3(256), smem, ts, dfeat, (long)fe, dprob, 0.25f, dref, ws);
    CK(hipDeviceSynchronize());
    std::vector<float> href((size_t)cells * 9 * 64), hout(href.size());
    CK(hipMemcpy(href.data(), dref, ob, hipMemcpyDeviceToHost));
    long bad = 0, total = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        for (int k = 0; k < 12; ++k)
            for (auto& s : bs) {
                if (kind) hipLaunchKernelGGL(bg_kernel<1>, dim3(256), dim3(512), 150 * 1024, s, dops, dsrc, src_bytes, dsink, 300);
                else hipLaunchKernelGGL(bg_kernel<0>, dim3(256), dim3(512), 150 * 1024, s, dops, dsrc, src_bytes, dsink, 300);
            }
        for (int rp = 0; rp < reps; ++rp) hipLaunchKernelGGL(pool_like, dim3(cells), dim3(256), smem, ts, dfeat, (long)fe, dprob, 0.25f, douts[rp], ws);
        CK(hipDeviceSynchronize());
        for (int rp = 0; rp < reps; ++rp) {
            CK(hipMemcpy(hout.data(), douts[rp], ob, hipMemcpyDeviceToHost));
            ++total;
            if (memcmp(hout.data(), href.data(), ob)) {
                ++bad;
                if (bad <= 5) {
                    for (size_t i = 0; i < hout.size(); ++i)
                        if (memcmp(&hout[i], &href[i], 4)) { printf("  round %d rep %d: first difference at cell %zu slot %zu channel %zu: got %.7g want %.7g\n", rd, rp, i / 576, (i / 64) % 9, i % 64, hout[i], href[i]); break; }
                }
            }
        }
    }
    printf("bg streams %d (kind %d): %ld of %ld pool_like runs differ from the run made alone\n", nbg, kind, bad, total);
    return 0;
}
