// Does a buffer store read its data registers at issue, or later?  (Background: HISTORY.md section 4, "a store-data hazard".)
// Each wave queues NLOAD LDS-DMA loads (cache-missing addresses), then issues ONE buffer_store_dwordx4 and overwrites the store's
// data registers with a poison value WAIT instructions later.  The output is then scanned for poison.
//   hipcc --offload-arch=gfx950 -O3 tools/store_hazard_repro.hip -o /tmp/shr && /tmp/shr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int NLOAD, int WAIT>
__global__ __launch_bounds__(256) void k(const char* src, unsigned src_bytes, int* out, unsigned out_bytes, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, out_bytes, 0x00020000);
    const unsigned wg = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            const unsigned soff = (unsigned)(((wg * 131u + it * 17u + j * 7919u) % (src_bytes / 4096u)) * 4096u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + (wave * NLOAD + j) * 1024), 16, lane * 16, soff, 0, 0);
        }
        const int tag = (wg << 12) | (it & 0xfff);
        const unsigned soff_o = __builtin_amdgcn_readfirstlane((wg * (unsigned)iters + it) * 1024u);
        const unsigned voff_o = lane * 16;
        // the store and the overwrite of its data registers in ONE asm block with fixed registers: nothing can be scheduled,
        // copied or padded in between
#define STORE_THEN_POISON(NOPS)                                                                                              \
        asm volatile("v_mov_b32 v100, %0\n\tv_add_u32 v101, 1, %0\n\tv_add_u32 v102, 2, %0\n\tv_add_u32 v103, 3, %0\n\t"         \
                     "s_nop 4\n\tbuffer_store_dwordx4 v[100:103], %1, %2, %3 offen\n\t" NOPS                                      \
                     "v_mov_b32 v100, -1\n\tv_mov_b32 v101, -1\n\tv_mov_b32 v102, -1\n\tv_mov_b32 v103, -1"                      \
                     :: "v"(tag), "v"(voff_o), "s"(ro), "s"(soff_o) : "v100", "v101", "v102", "v103", "v104", "v105", "memory")
        if (WAIT == 200 || WAIT == 201) {       // the same with global_store_dwordx4 (no wait state / one)
            int* gp = out + (size_t)soff_o / 4 + lane * 4;
#define GSTORE_THEN_POISON(NOPS)                                                                                             \
            asm volatile("v_mov_b32 v100, %0\n\tv_add_u32 v101, 1, %0\n\tv_add_u32 v102, 2, %0\n\tv_add_u32 v103, 3, %0\n\t"     \
                         "s_nop 4\n\tglobal_store_dwordx4 %1, v[100:103], off\n\t" NOPS                                          \
                         "v_mov_b32 v100, -1\n\tv_mov_b32 v101, -1\n\tv_mov_b32 v102, -1\n\tv_mov_b32 v103, -1"                  \
                         :: "v"(tag), "v"(gp) : "v100", "v101", "v102", "v103", "memory")
            if (WAIT == 200) GSTORE_THEN_POISON(""); else GSTORE_THEN_POISON("s_nop 0\n\t");
        } else
        if (WAIT == 0) STORE_THEN_POISON("");
        else if (WAIT == 1) STORE_THEN_POISON("s_nop 0\n\t");
        else if (WAIT == 2) STORE_THEN_POISON("s_nop 1\n\t");
        else if (WAIT == 3) STORE_THEN_POISON("s_nop 2\n\t");
        else if (WAIT == 4) STORE_THEN_POISON("s_nop 3\n\t");
        else if (WAIT == 8) STORE_THEN_POISON("s_nop 7\n\t");
        else if (WAIT == 101) STORE_THEN_POISON("v_mov_b32 v104, 0\n\t");                 // one unrelated VALU instruction in between
        else if (WAIT == 102) STORE_THEN_POISON("v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\t");
        else if (WAIT == 104) STORE_THEN_POISON("v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v104, 1\n\tv_mov_b32 v105, 1\n\t");
        else STORE_THEN_POISON("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <int NLOAD, int WAIT>
static void run(const char* name, const char* d_src, unsigned src_bytes, int* d_out, unsigned out_bytes, int blocks, int iters) {
    hipMemset(d_out, 0, out_bytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NLOAD, WAIT>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16 * 1024);
    hipLaunchKernelGGL((k<NLOAD, WAIT>), dim3(blocks), dim3(256), 4 * 16 * 1024, 0, d_src, src_bytes, d_out, out_bytes, iters);
    hipDeviceSynchronize();
    std::vector<int> h(out_bytes / 4);
    hipMemcpy(h.data(), d_out, out_bytes, hipMemcpyDeviceToHost);
    long poison = 0, wrong = 0, lanes[64] = {0};
    for (size_t i = 0; i < h.size(); ++i) {
        const size_t rec = i / 256;                              // one 1 KiB record per (wave, iteration)
        const int wgi = (int)(rec / iters), it = (int)(rec % iters), lane = (int)((i % 256) / 4), e = (int)(i % 4);
        const int want = ((wgi << 12) | (it & 0xfff)) + e;
        if (h[i] == -1) { ++poison; ++lanes[lane]; }
        else if (h[i] != want) ++wrong;
    }
    printf("%-66s poisoned dwords %8ld of %zu   other mismatches %ld   ", name, poison, h.size(), wrong);
    if (poison) { printf("lanes:"); for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d", l); }
    printf("\n");
}

int main() {
    const unsigned src_bytes = 1u << 30, blocks = 512, iters = 64;
    const unsigned out_bytes = blocks * 4 * iters * 1024;
    char* d_src; int* d_out;
    hipMalloc(&d_src, src_bytes); hipMalloc(&d_out, out_bytes);
    hipMemset(d_src, 1, src_bytes);
    run<0, 0>("no DMA queued, data registers overwritten by the NEXT instruction", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<0, 1>("no DMA queued, 1 wait state (s_nop 0)", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<0, 2>("no DMA queued, 2 wait states", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<0, 3>("no DMA queued, 3 wait states", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<0, 4>("no DMA queued, 4 wait states", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<0, 8>("no DMA queued, 8 wait states", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<0, 101>("no DMA queued, 1 other VALU instruction in between", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<0, 102>("no DMA queued, 2 other VALU instructions", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<0, 104>("no DMA queued, 4 other VALU instructions", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<0, 200>("global_store_dwordx4, data registers overwritten by the NEXT instruction", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<0, 201>("global_store_dwordx4, 1 wait state", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<16, 0>("16 LDS-DMA loads queued, next instruction", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<16, 2>("16 LDS-DMA loads queued, 2 wait states", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<16, 4>("16 LDS-DMA loads queued, 4 wait states", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<16, 8>("16 LDS-DMA loads queued, 8 wait states", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    run<16, 64>("16 LDS-DMA loads queued, 64 wait states", d_src, src_bytes, d_out, out_bytes, blocks, iters);
    return 0;
}
