// What can the L2 -> LDS path deliver?  Workgroups that do nothing but LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction) of
// L2-resident data, in the conv kernel's launch shape (8 waves per workgroup, one workgroup per CU, ~74 KB per "chunk", wait + barrier
// per chunk).  The stride-2 conv tiles move 74 KB of LDS-DMA per 27 (f16x3) / 18 (f16+fp8x2) MFMA-units of work per wave pair; if this
// loop cannot go faster than they do, they are at the limit of the path.   Output: profiles/r03_lds_dma_rate.txt
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_rate.hip -o /tmp/lds_dma_rate && /tmp/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;

template <int PIECES>     // 1 KiB pieces per chunk and workgroup
__global__ __launch_bounds__(512, 2) void dma_kernel(const char* __restrict__ src, unsigned src_bytes, int chunks, unsigned stride, float* out, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned base = (blockIdx.x * 7919u * 1024u) % (src_bytes - PIECES * 1024u - 64u * stride);
    base &= ~1023u;
    for (int c = 0; c < chunks; ++c) {
        char* d = smem + (c & 1) * PIECES * 1024;
#pragma unroll
        for (int i = 0; i < (PIECES + 7) / 8; ++i) {
            const int piece = i * 8 + wave;
            if (piece < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(d + piece * 1024), 16, lane * 16, base + piece * 1024 + (c & 15) * stride, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    if (smem[threadIdx.x] == 123 && chunks < 0) out[0] = 1.f;
}

template <int PIECES>
static void run(const char* name, const char* d_src, unsigned bytes, int cus, int wgs_per_cu, int chunks, unsigned stride, float* d_out, unsigned long long* d_clk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int smem = 2 * PIECES * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(dma_kernel<PIECES>, dim3(cus * wgs_per_cu), dim3(512), smem, 0, d_src, bytes, chunks, stride, d_out, d_clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long clk[2]; hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
        const double total = (double)cus * wgs_per_cu * chunks * PIECES * 1024.0;
        const double ghz = clk[1] ? (double)clk[0] / clk[1] * 0.1 : 0.0;
        printf("%-52s rep%d %8.3f ms  %7.2f TB/s aggregate  %6.1f B/clk/CU at %.2f GHz\n", name, rep, ms, total / ms * 1e-9, total / (ms * 1e-3) / cus / (ghz * 1e9), ghz);
    }
}

int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const unsigned bytes = 24u << 20;       // 24 MiB: inside the 32 MiB of aggregate L2 / well inside the 256 MiB Infinity Cache
    char* d_src; float* d_out; unsigned long long* d_clk;
    hipMalloc(&d_src, bytes); hipMemset(d_src, 1, bytes); hipMalloc(&d_out, 4); hipMalloc(&d_clk, 16);
    run<74>("74 KiB chunks (stride-2 tile: 38 KiB halo + 36 KiB w)", d_src, bytes, cus, 4, 400, 0, d_out, d_clk);
    run<74>("74 KiB chunks, source window moving 64 KiB per chunk", d_src, bytes, cus, 4, 400, 65536, d_out, d_clk);
    run<75>("75 KiB chunks (stride-1 tile: 39 KiB halo + 36 KiB w)", d_src, bytes, cus, 4, 400, 0, d_out, d_clk);
    run<36>("36 KiB chunks (weights only)", d_src, bytes, cus, 4, 800, 0, d_out, d_clk);
    return 0;
}
