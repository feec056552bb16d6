// Sustained matrix-pipe rates of one MI355X for the instruction mixes the conv kernel can be built from, with the effective
// shader clock of every run (s_memtime ticks / s_memrealtime at 100 MHz), plus a numerical layout check of the
// block-scaled fp8 K=64 instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_mix.hip -o gpurun_out/mfma_mix && gpurun_out/mfma_mix
// Registers only: no LDS or memory traffic in the timed loops.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <sys/time.h>
#include <unistd.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// MIX 0: f16 x1 (12 MFMAs / iter on 4 accumulators, distinct operands)
// MIX 1: f16x3 (hi*lo, lo*hi, hi*hi)
// MIX 2: per accumulator and K=64: 4 f16 MFMAs + 2 fp8 K=64 MFMAs  (= f16 main product + two fp8 correction products)
// MIX 3: fp8 K=64 only
// MIX 4: per accumulator and K=64: 4 f16 + 1 fp8 K=64
// MIX 5: per accumulator and K=64: 8 f16 + 1 fp8 K=64   (w_h a_h + w_l a_h in fp16, only the activation residual in fp8: "x2q")
// MIX 6: f16x3 CHAINED: the 12 MFMAs of four 16-channel blocks (w_lo a_hi, w_hi a_lo, w_hi a_hi each) back to back on ONE accumulator,
//        then the next accumulator - the issue pattern of MIX 2-5 (MIX 1 rotates over the 4 accumulators: issue-limited even on zeros)
// MIX 7: plain f16, chained the same way (12 per accumulator)
// MIX 8 / 9: as MIX 2 with the K=64 instruction's operands declared fp6 (e2m3) / fp4 (e2m1) - half the passes of fp8 if the pipe runs them
//            at the 2x datasheet rate: would the correction products get cheaper?  (MIX 10 / 11: fp6 / fp4 K=64 only)
template <int MIX, int WPS>
__global__ __launch_bounds__(WPS * 256) void loop_kernel(const f16x8* __restrict__ ops, const i32x8* __restrict__ ops8,
                                                          float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = ops[i * 64 + lane]; b[i] = ops[(4 + i) * 64 + lane]; }
    const f16x8 al = ops[8 * 64 + lane], bl = ops[9 * 64 + lane];
    i32x8 a8[2], b8[2];
    for (int i = 0; i < 2; ++i) { a8[i] = ops8[i * 64 + lane]; b8[i] = ops8[(2 + i) * 64 + lane]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
        if (MIX == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[(i + 1) & 3], a[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[i], a[(i + 2) & 3], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[i], a[i], acc[i], 0, 0, 0);
        } else if (MIX == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, a[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[i], al, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[i], a[i], acc[i], 0, 0, 0);
        } else if (MIX == 6 || MIX == 7) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(MIX == 6 ? bl : b[(i + k + 1) & 3], a[k], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[(i + k) & 3], MIX == 6 ? al : a[(k + 1) & 3], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[(i + k) & 3], a[k], acc[i], 0, 0, 0);
                }
        } else if (MIX == 8 || MIX == 9) {
            constexpr int F = MIX == 8 ? 2 : 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[(i + k) & 3], a[k], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[i & 1], a8[0], acc[i], F, F, 0, 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[0], a8[1], acc[i], F, F, 0, 0, 0, 0);
            }
        } else if (MIX == 10 || MIX == 11) {
            constexpr int F = MIX == 10 ? 2 : 4;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[(i + r) & 1], a8[r], acc[i], F, F, 0, 0, 0, 0);
        } else if (MIX == 2 || MIX == 4 || MIX == 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[(i + k) & 3], a[k], acc[i], 0, 0, 0);
                if (MIX == 5) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k & 1 ? bl : b[(i + k + 1) & 3], a[k], acc[i], 0, 0, 0);
                }
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[i & 1], a8[0], acc[i], 0, 0, 0, 0, 0, 0);
                if (MIX == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[0], a8[1], acc[i], 0, 0, 0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[(i + r) & 1], a8[r], acc[i], 0, 0, 0, 0, 0, 0);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[0] = s;
}

// ---- layout check of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 both operands) ----
// hypothesis: lane l holds row (l & 31) of A [32 x 64] (resp. column of B [64 x 32]), k = 32 (l >> 5) + j, j = byte index 0..31
// of its 8 dwords; the scale byte applies to those 32 elements (E8M0: 2^(s - 127)); D[row][col]: col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
__global__ void check_kernel(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* d, int use_scale) {
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (use_scale) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[lane], b[lane], acc, 0, 0, 0, sa[lane], 0, sb[lane]);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[lane], b[lane], acc, 0, 0, 0, 0, 0, 0);
    for (int e = 0; e < 16; ++e) d[lane * 16 + e] = acc[e];
}

static float e4m3_to_float(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -f : f;
}
static float gauss() { float u = 0.f; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.f; }

template <int MIX, int WPS>
static void run(const char* name, const f16x8* d_ops, const i32x8* d_ops8, float* d_out, unsigned long long* d_clk, int cus, int iters, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cus * 4;     // several workgroups per CU over the run
    const double mf16 = MIX == 0 || MIX == 1 ? 12 : (MIX == 6 || MIX == 7 ? 48 : (MIX == 3 || MIX >= 10 ? 0 : (MIX == 5 ? 32 : 16))), m8 = MIX == 2 || MIX == 8 || MIX == 9 ? 8 : (MIX == 3 || MIX >= 10 ? 8 : (MIX == 4 || MIX == 5 ? 4 : 0));
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<MIX, WPS>), dim3(blocks), dim3(WPS * 256), 0, 0, d_ops, d_ops8, d_out, d_clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long clk[2]; hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
        const double waves = (double)blocks * WPS * 4;
        const double flop = waves * iters * (mf16 * 2.0 * 32 * 32 * 16 + m8 * 2.0 * 32 * 32 * 64);
        // pipe time in units of one f16 32x32x16 (fp8 K=64 = 2 units)
        const double units = waves * iters * (mf16 + 2 * m8);
        printf("%-44s wps=%d iters=%-7d rep%d %8.3f ms  %8.1f TFLOP/s  pipe-units %6.1f G/s (%.0f%% of 2.4GHz peak)  clk %.3f GHz\n", name, WPS, iters, rep, ms,
               flop / ms * 1e-9, units / ms * 1e-6, units * 32 / (ms * 1e-3) / (cus * 4 * 2.4e9) * 100, clk[1] ? (double)clk[0] / clk[1] * 0.1 : 0.0);
    }
}

// --sustain S: each mix for S seconds of back-to-back launches on random operands, with wall-clock stamps for tools/power_per_kernel.py
// (which samples the socket power meanwhile): "SEG name t0 t1 pipe-units-G/s executed-TFLOP/s clk-GHz"
static double now_s() { timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + tv.tv_usec * 1e-6; }
template <int MIX, int WPS>
static void sustain(const char* name, const f16x8* d_ops, const i32x8* d_ops8, float* d_out, unsigned long long* d_clk, int cus, int iters, double seconds) {
    const int blocks = cus * 4;
    const double mf16 = MIX == 0 || MIX == 1 ? 12 : (MIX == 6 || MIX == 7 ? 48 : (MIX == 3 || MIX >= 10 ? 0 : (MIX == 5 ? 32 : 16))), m8 = MIX == 2 || MIX == 8 || MIX == 9 ? 8 : (MIX == 3 || MIX >= 10 ? 8 : (MIX == 4 || MIX == 5 ? 4 : 0));
    // fp6 / fp4 K = 64 instructions take half the passes of fp8 (1 unit instead of 2)
    const double u8 = (MIX >= 8) ? 1.0 : 2.0;
    hipDeviceSynchronize();
    usleep(1500000);                  // idle gap between segments (the power reading settles)
    const double t0 = now_s();
    long launches = 0;
    double clk_sum = 0;
    while (now_s() - t0 < seconds) {
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((loop_kernel<MIX, WPS>), dim3(blocks), dim3(WPS * 256), 0, 0, d_ops, d_ops8, d_out, d_clk, iters);
        hipDeviceSynchronize();
        unsigned long long clk[2]; hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
        clk_sum += clk[1] ? (double)clk[0] / clk[1] * 0.1 : 0.0;
        launches += 4;
    }
    const double t1 = now_s();
    const double waves = (double)blocks * WPS * 4;
    const double flop = waves * iters * launches * (mf16 * 2.0 * 32 * 32 * 16 + m8 * 2.0 * 32 * 32 * 64);
    const double units = waves * iters * launches * (mf16 + u8 * m8);
    printf("SEG %s|%.4f|%.4f|%.1f|%.1f|%.3f\n", name, t0, t1, units / (t1 - t0) * 1e-9, flop / (t1 - t0) * 1e-12, clk_sum / (launches / 4));
    fflush(stdout);
}

int main(int argc, char** argv) {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    // ---------------- layout check ----------------
    {
        std::vector<unsigned char> ha(64 * 32), hb(64 * 32);
        std::vector<int> hsa(64), hsb(64);
        for (auto& v : ha) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v &= 0xfe; if (((v >> 3) & 15) > 9) v &= 0xbf; }
        for (auto& v : hb) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v &= 0xfe; if (((v >> 3) & 15) > 9) v &= 0xbf; }
        for (int l = 0; l < 64; ++l) { hsa[l] = 120 + rand() % 12; hsb[l] = 122 + rand() % 8; }
        i32x8 *da, *db; int *dsa, *dsb; float* dd;
        hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 4096);
        hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
        for (int use_scale = 0; use_scale < 2; ++use_scale) {
            hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, use_scale);
            std::vector<float> hd(1024); hipMemcpy(hd.data(), dd, 4096, hipMemcpyDeviceToHost);
            double maxerr = 0, maxref = 0;
            for (int lane = 0; lane < 64; ++lane)
                for (int reg = 0; reg < 16; ++reg) {
                    const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    double ref = 0;
                    for (int k = 0; k < 64; ++k) {
                        const int la = row + 32 * (k >> 5), lb = col + 32 * (k >> 5), j = k & 31;
                        double av = e4m3_to_float(ha[la * 32 + j]), bv = e4m3_to_float(hb[lb * 32 + j]);
                        if (use_scale) { av *= ldexp(1.0, hsa[la] - 127); bv *= ldexp(1.0, hsb[lb] - 127); }
                        ref += av * bv;
                    }
                    maxerr = fmax(maxerr, fabs(ref - hd[lane * 16 + reg])); maxref = fmax(maxref, fabs(ref));
                }
            printf("layout check (%s): max |D - ref| = %.3e  (max |ref| %.3e)\n", use_scale ? "per-lane E8M0 scales" : "scale operands 0 = unscaled", maxerr, maxref);
        }
    }
    // ---------------- rates ----------------
    std::vector<_Float16> h(10 * 64 * 8);
    std::vector<unsigned char> h8(4 * 64 * 32);
    f16x8* d_ops; i32x8* d_ops8; float* d_out; unsigned long long* d_clk;
    hipMalloc(&d_ops, h.size() * 2); hipMalloc(&d_ops8, h8.size()); hipMalloc(&d_out, 4); hipMalloc(&d_clk, 16);
    if (argc > 2 && !strcmp(argv[1], "--sustain")) {
        const double secs = atof(argv[2]);
        for (size_t i = 0; i < h.size(); ++i) { float v = gauss(); if (i >= 8 * 64 * 8) v *= 4.8e-4f; h[i] = (_Float16)v; }
        for (auto& v : h8) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v &= 0xfe; }
        hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(d_ops8, h8.data(), h8.size(), hipMemcpyHostToDevice);
        sustain<8, 2>("mix: 4 f16 + 2 fp6-K64 (f16+fp6x2), 2 waves/SIMD", d_ops, d_ops8, d_out, d_clk, cus, 4000, secs);
        sustain<8, 1>("mix: 4 f16 + 2 fp6-K64 (f16+fp6x2), 1 wave/SIMD", d_ops, d_ops8, d_out, d_clk, cus, 4000, secs);
        sustain<2, 2>("mix: 4 f16 + 2 fp8-K64 (f16+fp8x2), 2 waves/SIMD", d_ops, d_ops8, d_out, d_clk, cus, 4000, secs);
        sustain<6, 2>("mix: f16x3 chained, 2 waves/SIMD", d_ops, d_ops8, d_out, d_clk, cus, 1500, secs);
        sustain<7, 2>("mix: plain f16 chained, 2 waves/SIMD", d_ops, d_ops8, d_out, d_clk, cus, 1500, secs);
        sustain<10, 2>("mix: fp6-K64 only, 2 waves/SIMD", d_ops, d_ops8, d_out, d_clk, cus, 8000, secs);
        // the same mixes on zeros: what the issue structure alone sustains
        hipMemset((void*)d_ops, 0, h.size() * 2); hipMemset((void*)d_ops8, 0, h8.size());
        sustain<8, 2>("mix: 4 f16 + 2 fp6-K64, ZERO operands", d_ops, d_ops8, d_out, d_clk, cus, 4000, secs);
        sustain<6, 2>("mix: f16x3 chained, ZERO operands", d_ops, d_ops8, d_out, d_clk, cus, 1500, secs);
        return 0;
    }
    const int long_iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int ndata = argc > 2 ? atoi(argv[2]) : 2;      // > 2: also the toggle-rate experiments (lo operands with their low mantissa bits cleared)
    for (int data = 0; data < ndata; ++data) {
        const int lo_clear = data == 2 ? 5 : (data == 3 ? 8 : 0);
        printf("==== operand data: %s%s ====\n", data == 0 ? "zeros" : "N(0,1) fp16 hi, 2^-11-scaled lo, random e4m3 bytes",
               data == 2 ? "; lo operands: low 5 mantissa bits cleared" : (data == 3 ? "; lo operands: low 8 mantissa bits cleared; hi operands ReLU-like (half zeros)" : ""));
        for (size_t i = 0; i < h.size(); ++i) {
            float v = data == 0 ? 0.f : gauss();
            if (i >= 8 * 64 * 8) v *= 4.8e-4f;
            else if (data == 3 && i < 4 * 64 * 8 && v < 0.f) v = 0.f;       // a[] = pixel operands
            h[i] = (_Float16)v;
            if (lo_clear && i >= 8 * 64 * 8) { unsigned short b; memcpy(&b, &h[i], 2); b &= (unsigned short)~((1u << lo_clear) - 1u); memcpy(&h[i], &b, 2); }
        }
        for (auto& v : h8) { v = data == 0 ? 0 : (rand() & 0xff); if ((v & 0x7f) == 0x7f) v &= 0xfe; }
        hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(d_ops8, h8.data(), h8.size(), hipMemcpyHostToDevice);
        run<0, 2>("f16 32x32x16 (distinct hi operands)", d_ops, d_ops8, d_out, d_clk, cus, 400, 2);          // short burst
        run<0, 2>("f16 32x32x16 (distinct hi operands)", d_ops, d_ops8, d_out, d_clk, cus, long_iters, 3);
        run<0, 1>("f16 32x32x16 (distinct hi operands)", d_ops, d_ops8, d_out, d_clk, cus, long_iters, 2);
        run<1, 2>("f16x3 mix (hi*lo, lo*hi, hi*hi)", d_ops, d_ops8, d_out, d_clk, cus, long_iters, 3);
        run<1, 1>("f16x3 mix (hi*lo, lo*hi, hi*hi)", d_ops, d_ops8, d_out, d_clk, cus, long_iters, 2);
        run<6, 2>("f16x3 mix, CHAINED per accumulator", d_ops, d_ops8, d_out, d_clk, cus, long_iters / 4, 3);
        run<6, 1>("f16x3 mix, CHAINED per accumulator", d_ops, d_ops8, d_out, d_clk, cus, long_iters / 4, 2);
        run<7, 2>("f16 32x32x16, CHAINED per accumulator", d_ops, d_ops8, d_out, d_clk, cus, long_iters / 4, 3);
        run<3, 2>("fp8 e4m3 32x32x64 only", d_ops, d_ops8, d_out, d_clk, cus, long_iters, 3);
        run<2, 2>("4 f16 + 2 fp8-K64 per K=64 (f16 + 2 fp8 corr.)", d_ops, d_ops8, d_out, d_clk, cus, long_iters * 3 / 4, 3);
        run<8, 2>("4 f16 + 2 fp6-K64 per K=64 (units as if fp8)", d_ops, d_ops8, d_out, d_clk, cus, long_iters * 3 / 4, 3);
        run<9, 2>("4 f16 + 2 fp4-K64 per K=64 (units as if fp8)", d_ops, d_ops8, d_out, d_clk, cus, long_iters * 3 / 4, 3);
        run<10, 2>("fp6 e2m3 32x32x64 only (units as if fp8)", d_ops, d_ops8, d_out, d_clk, cus, long_iters, 2);
        run<11, 2>("fp4 e2m1 32x32x64 only (units as if fp8)", d_ops, d_ops8, d_out, d_clk, cus, long_iters, 2);
        run<2, 1>("4 f16 + 2 fp8-K64 per K=64 (f16 + 2 fp8 corr.)", d_ops, d_ops8, d_out, d_clk, cus, long_iters * 3 / 4, 2);
        run<4, 2>("4 f16 + 1 fp8-K64 per K=64", d_ops, d_ops8, d_out, d_clk, cus, long_iters * 3 / 4, 2);
        run<5, 2>("8 f16 + 1 fp8-K64 per K=64 (x2q)", d_ops, d_ops8, d_out, d_clk, cus, long_iters / 2, 3);
    }
    return 0;
}
