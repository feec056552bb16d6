// Stand-alone reproducer of the packed-fp32 fault of HISTORY.md section 4 (found through pool_partial_kernel, round 3).
//   hipcc --offload-arch=gfx950 -O3 tools/pk_fault_repro.hip -o /tmp/pk_repro
//   /tmp/pk_repro [bg_streams=7] [rounds=40] [bg_kind=0|1] [bg_lds_kib=100]
// Part 1 (registers only): a kernel runs the same multiply-add recurrences as v_pk_fma_f32 - one operand-selection form per group of
// accumulators - and as v_fmac_f32 on the same numbers, and counts the results that differ: alone, and while `bg_streams` other
// streams run a 512-thread MFMA kernel on every CU.  On the MI355X boxes of this project: 0 wrong alone; next to the MFMA kernel
// `op_sel:[0,1,0]` (low result half from the HIGH dword of src1) gives ~10^6 wrong LOW halves, every other form 0
// (profiles/r03_pk_fma_op_sel_fault.txt).
// Part 2: a kernel shaped like pool_partial_kernel's 16-byte path (the SLP vectoriser packs its 72 multiply-adds per pixel pair with
// op_sel broadcasts) run again and again next to the same background and compared bit for bit with the run made alone; it fails
// when a background workgroup leaves room for it on the CU (bg_lds_kib = 100) and never when it does not (150), and never when
// built with -fno-slp-vectorize.  bg_kind 1 adds buffer_load ... lds traffic to the background (not needed for the fault).
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

// Registers-only self-check: every lane runs the same multiply-add recurrences as packed fp32 (v_pk_fma_f32, one operand-selection form per
// group of accumulators) and as scalar v_fmac_f32 on the same numbers, and counts the results that differ, per form.  No memory, no LDS.
//   form 0: plain pairs                      v_pk_fma_f32 d, a, b, d
//   form 1: src1 low dword to both halves    ... op_sel_hi:[1,0,1]
//   form 2: src1 high dword to both halves   ... op_sel:[0,1,0]
//   form 3: an SGPR pair as src1             v_pk_fma_f32 d, a, s[n:n+1], d
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NF = 9;
__global__ __launch_bounds__(256) void pk_selfcheck(unsigned* __restrict__ bad, float* __restrict__ sink, int iters, float sgpr_val) {
    const float l = (float)(threadIdx.x & 63) * 0.001f + 1.0f;
    f32x2 acc[NF][2];
    float sa[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[f][i] = f32x2{0.f, 0.f}; sa[f][2 * i] = 0.f; sa[f][2 * i + 1] = 0.f; }
    float x = l, y = 0.5f + l * 0.25f;
    const float sv = __builtin_amdgcn_readfirstlane(sgpr_val);
    for (int it = 0; it < iters; ++it) {
        x = x * 0.999f + 0.001f; y = y * 1.0005f - 0.0004f;
        f32x2 pr = {y, y * 0.5f};
        asm volatile("" : "+v"(pr));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x2 xv = {x + (float)i, x - (float)i};
            asm volatile("" : "+v"(xv));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[0][i]) : "v"(xv), "v"(pr));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[1][i]) : "v"(xv), "v"(pr));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[2][i]) : "v"(xv), "v"(pr));
            acc[3][i] = __builtin_elementwise_fma(xv, f32x2{sv, sv}, acc[3][i]);
            float a0 = xv.x, a1 = xv.y, p0 = pr.x, p1 = pr.y, s0 = sv;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(p0), "+v"(p1), "+v"(s0));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[0][2 * i]) : "v"(a0), "v"(p0));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[0][2 * i + 1]) : "v"(a1), "v"(p1));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[1][2 * i]) : "v"(a0), "v"(p0));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[1][2 * i + 1]) : "v"(a1), "v"(p0));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[2][2 * i]) : "v"(a0), "v"(p1));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[2][2 * i + 1]) : "v"(a1), "v"(p1));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[3][2 * i]) : "v"(a0), "v"(s0));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[3][2 * i + 1]) : "v"(a1), "v"(s0));
            // more forms: 4 op_sel:[1,0,0] (src0 high -> low half), 5 op_sel_hi:[0,1,1] (src0 low -> high half), 6 v_pk_mul_f32 op_sel:[0,1], 7 v_pk_add_f32 op_sel:[0,1],
            // 8 op_sel_hi:[0,0,1] (both sources: low dwords to the high half)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc[4][i]) : "v"(xv), "v"(pr));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[5][i]) : "v"(xv), "v"(pr));
            { f32x2 t; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(xv), "v"(pr)); acc[6][i] = t; }
            { f32x2 t; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(xv), "v"(pr)); acc[7][i] = t; }
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(acc[8][i]) : "v"(xv), "v"(pr));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[4][2 * i]) : "v"(a1), "v"(p0));          // low half: src0 HIGH x src1 low
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[4][2 * i + 1]) : "v"(a1), "v"(p1));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[5][2 * i]) : "v"(a0), "v"(p0));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[5][2 * i + 1]) : "v"(a0), "v"(p1));      // high half: src0 LOW x src1 high
            { float t0, t1; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(a0), "v"(p1)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(a1), "v"(p1)); sa[6][2 * i] = t0; sa[6][2 * i + 1] = t1; }
            { float t0, t1; asm volatile("v_add_f32 %0, %1, %2" : "=v"(t0) : "v"(a0), "v"(p1)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(t1) : "v"(a1), "v"(p1)); sa[7][2 * i] = t0; sa[7][2 * i + 1] = t1; }
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[8][2 * i]) : "v"(a0), "v"(p0));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sa[8][2 * i + 1]) : "v"(a0), "v"(p0));      // high half: src0 low x src1 low
        }
    }
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        unsigned nb = 0, nlo = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned dl = __float_as_uint(acc[f][i].x) != __float_as_uint(sa[f][2 * i]), dh = __float_as_uint(acc[f][i].y) != __float_as_uint(sa[f][2 * i + 1]);
            nb += dl + dh; nlo += dl;
            s += acc[f][i].x + acc[f][i].y;
        }
        if (nb) { atomicAdd(bad + 2 * f, nb); atomicAdd(bad + 2 * f + 1, nlo); }
    }
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int KIND>
__global__ __launch_bounds__(512, 2) void bg_kernel(const f16x8* __restrict__ ops, const char* __restrict__ stream_src, unsigned src_bytes, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem_bg[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a = ops[lane], b = ops[64 + lane];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f16x8* lds = reinterpret_cast<f16x8*>(smem_bg);
    for (int i = threadIdx.x; i < 96 * 1024 / 16; i += 512) lds[i] = a;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)stream_src, 0, src_bytes, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        if (KIND == 1) {
            // stream 8 KiB per wave per iteration into this wave's LDS slice through the LDS-DMA path
            typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
            for (int pce = 0; pce < 8; ++pce) {
                const unsigned off = ((unsigned)(blockIdx.x * 8 + wave) * 8192u + (unsigned)pce * 1024u + (unsigned)it * 65536u) % (src_bytes - 8192u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem_bg + wave * 8192 + pce * 1024), 16, lane * 16, off & ~15u, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f16x8 x = lds[(wave * 512 + t * 64 + lane) % (96 * 64)];
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
    const int lds_kib = argc > 4 ? atoi(argv[4]) : 100;
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
                if (kind) hipLaunchKernelGGL(bg_kernel<1>, dim3(256), dim3(512), lds_kib * 1024, s, dops, dsrc, src_bytes, dsink, 300);
                else hipLaunchKernelGGL(bg_kernel<0>, dim3(256), dim3(512), lds_kib * 1024, s, dops, dsrc, src_bytes, dsink, 300);
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
    {   // registers-only self-check next to the same background
        unsigned* dbad; CK(hipMalloc(&dbad, 8 * NF)); CK(hipMemset(dbad, 0, 8 * NF));
        hipLaunchKernelGGL(pk_selfcheck, dim3(4096), dim3(256), 0, ts, dbad, dsink, 2000, 0.75f);
        CK(hipDeviceSynchronize());
        unsigned alone[2 * NF], busy[2 * NF]; CK(hipMemcpy(alone, dbad, 8 * NF, hipMemcpyDeviceToHost));
        CK(hipMemset(dbad, 0, 8 * NF));
        for (int rd = 0; rd < rounds; ++rd) {
            for (int k = 0; k < 12; ++k)
                for (auto& s2 : bs) {
                    if (kind) hipLaunchKernelGGL(bg_kernel<1>, dim3(256), dim3(512), lds_kib * 1024, s2, dops, dsrc, src_bytes, dsink, 300);
                    else hipLaunchKernelGGL(bg_kernel<0>, dim3(256), dim3(512), lds_kib * 1024, s2, dops, dsrc, src_bytes, dsink, 300);
                }
            for (int rp = 0; rp < 10; ++rp) hipLaunchKernelGGL(pk_selfcheck, dim3(4096), dim3(256), 0, ts, dbad, dsink, 2000, 0.75f);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(busy, dbad, 8 * NF, hipMemcpyDeviceToHost));
        const char* names[NF] = {"v_pk_fma_f32, plain pairs", "v_pk_fma_f32 op_sel_hi:[1,0,1] (src1 low -> both)", "v_pk_fma_f32 op_sel:[0,1,0] (src1 HIGH -> both)", "v_pk_fma_f32, broadcast pair in VGPRs",
                                 "v_pk_fma_f32 op_sel:[1,0,0] (src0 HIGH -> low half)", "v_pk_fma_f32 op_sel_hi:[0,1,1] (src0 low -> high half)", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[0,1]",
                                 "v_pk_fma_f32 op_sel_hi:[0,0,1] (src0, src1 low -> high half)"};
        printf("registers-only self-check, v_pk_fma_f32 against v_fmac_f32 on the same numbers (%d launches x 4096 x 256 lanes x 4 results per form):\n", rounds * 10);
        for (int f = 0; f < NF; ++f)
            printf("  %-60s alone: %u wrong   next to the background: %u wrong (%u of them in the LOW half)\n", names[f], alone[2 * f], busy[2 * f], busy[2 * f + 1]);
    }
    printf("bg streams %d (kind %d, %d KiB LDS): %ld of %ld pool_like runs differ from the run made alone\n", nbg, kind, lds_kib, bad, total);
    return 0;
}
