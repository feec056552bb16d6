// Probes for an fp6 (e2m3) variant of the correction products (the K = 64 MFMA runs fp6 operands in half the passes of fp8):
//  1. v_cvt_scalef32_pk32_fp6_f16 / v_cvt_scalef32_2xpk16_fp6_f32: element order of the 192-bit result, direction of the scale, rounding,
//     saturation;
//  2. v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = blgp = 2: is a lane's operand the little-endian stream of 32 six-bit fields in v[0:5]
//     (lanes 0-31: K 0-31, lanes 32-63: K 32-63, as for fp8), and do the E8M0 scales apply as for fp8?
//   hipcc --offload-arch=gfx950 -O3 tools/fp6_probe.hip -o /tmp/fp6_probe && /tmp/fp6_probe       -> profiles/r03_fp6_probe.txt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void cvt_kernel(const _Float16* x, const float* y, int* out, float scale) {
    f16x32 v; for (int j = 0; j < 32; ++j) v[j] = x[threadIdx.x * 32 + j];
    f32x16 a, b; for (int j = 0; j < 16; ++j) { a[j] = y[threadIdx.x * 32 + j]; b[j] = y[threadIdx.x * 32 + 16 + j]; }
    const i32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
    const i32x6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int d = 0; d < 6; ++d) { out[threadIdx.x * 6 + d] = r[d]; out[(64 + threadIdx.x) * 6 + d] = q[d]; }
}
__global__ void mfma_kernel(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* d) {
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[lane], b[lane], acc, 2, 2, 0, sa[lane], 0, sb[lane]);
    for (int e = 0; e < 16; ++e) d[lane * 16 + e] = acc[e];
}

static float fp6_value(int c) {            // OCP e2m3: sign, 2 exponent bits (bias 1), 3 mantissa bits
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float f = e == 0 ? m * 0.125f : ldexpf(1.f + m / 8.f, e - 1);
    return s ? -f : f;
}
static int fp6_encode(float x) {           // round to nearest even, saturating at 7.5
    const int s = std::signbit(x) ? 32 : 0;
    float ax = fabsf(x);
    if (!(ax == ax)) return s | 31;
    if (ax >= 7.5f) return s | 31;
    int e = 0; if (ax >= 1.f) { int ex; frexpf(ax, &ex); e = ex - 1; }      // binade 0, 1, 2
    const float step = ldexpf(1.f, e - 3);
    float q = nearbyintf(ax / step);                                       // in units of the step
    float v = q * step;
    if (v >= 7.5f) return s | 31;
    // re-derive the code from the rounded value
    if (v < 1.f) return s | (int)(v * 8.f);
    int ex; frexpf(v, &ex); const int ee = ex - 1;
    return s | ((ee + 1) << 3) | (int)((v / ldexpf(1.f, ee) - 1.f) * 8.f);
}
static int field(const int* w, int j) {    // six-bit field j of a little-endian bit stream
    const int bit = 6 * j, d = bit >> 5, o = bit & 31;
    unsigned long long v = (unsigned)w[d];
    if (d + 1 < 6) v |= (unsigned long long)(unsigned)w[d + 1] << 32;
    return (int)((v >> o) & 63);
}
static void put(int* w, int j, int code) {
    const int bit = 6 * j, d = bit >> 5, o = bit & 31;
    unsigned long long v = (unsigned long long)(code & 63) << o;
    w[d] |= (int)(unsigned)v;
    if (d + 1 < 8) w[d + 1] |= (int)(unsigned)(v >> 32);
}

int main() {
    // ---------------- 1. conversions ----------------
    std::vector<_Float16> hx(64 * 32); std::vector<float> hy(64 * 32);
    srand(7);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            float v;
            if (l == 0) v = 0.125f * j * (j & 1 ? -1.f : 1.f);                           // subnormal / first binade grid
            else if (l == 1) v = 0.9375f + 0.0625f * j;                                  // ties between codes (multiples of half a step)
            else if (l == 2) v = 6.f + 0.125f * j;                                       // around the top: 7.5 saturation
            else v = ldexpf((rand() / (float)RAND_MAX * 2.f - 1.f), (rand() % 6) - 2);   // random, |v| < 8
            hx[l * 32 + j] = (_Float16)v; hy[l * 32 + j] = (float)(_Float16)v;
        }
    _Float16* dx; float* dy; int* dout;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dy, hy.size() * 4); hipMalloc(&dout, 128 * 24);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dy, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
    for (float scale : {1.f, 0.25f, 4.f}) {
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, dx, dy, dout, scale);
        std::vector<int> ho(128 * 6); hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
        int bad_seq16 = 0, bad_seqf = 0, bad_il = 0, bad_mul = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 32; ++j) {
                const float v = (float)hx[l * 32 + j];
                const int want_div = fp6_encode(v / scale), want_mul = fp6_encode(v * scale);
                const int got16 = field(&ho[l * 6], j);
                bad_seq16 += got16 != want_div; bad_mul += got16 != want_mul;
                // f32 form: src0 = elements 0-15, src1 = elements 16-31 of the same row
                const int seq = field(&ho[(64 + l) * 6], j);
                const int il = field(&ho[(64 + l) * 6], j < 16 ? 2 * j : 2 * (j - 16) + 1);
                bad_seqf += seq != want_div; bad_il += il != want_div;
            }
        printf("scale %-5g pk32_fp6_f16: %d of 2048 fields differ from fp6(x / scale) in sequential order (vs x * scale: %d);  2xpk16_fp6_f32: sequential [src0 | src1] %d, interleaved (src0[i], src1[i]) %d\n",
               scale, bad_seq16, bad_mul, bad_seqf, bad_il);
        if (scale == 1.f) {
            printf("  lane 2 (values 6 + j/8): ");
            for (int j = 0; j < 32; ++j) printf("%g->%g ", (double)(float)hx[2 * 32 + j], (double)fp6_value(field(&ho[2 * 6], j)));
            printf("\n");
        }
    }
    // ---------------- 2. MFMA operand layout ----------------
    std::vector<int> ha(64 * 8, 0), hb(64 * 8, 0), hsa(64), hsb(64);
    std::vector<int> ca(64 * 32), cb(64 * 32);
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 32; ++j) { ca[l * 32 + j] = rand() & 63; cb[l * 32 + j] = rand() & 63; put(&ha[l * 8], j, ca[l * 32 + j]); put(&hb[l * 8], j, cb[l * 32 + j]); }
        ha[l * 8 + 6] = rand(); ha[l * 8 + 7] = rand(); hb[l * 8 + 6] = rand(); hb[l * 8 + 7] = rand();      // the two unused dwords: garbage
        hsa[l] = 124 + rand() % 6; hsb[l] = 125 + rand() % 5;
    }
    i32x8 *da, *db; int *dsa, *dsb; float* dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 4096);
    hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    std::vector<float> hd(1024); hipMemcpy(hd.data(), dd, 4096, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int reg = 0; reg < 16; ++reg) {
            const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            double ref = 0;
            for (int k = 0; k < 64; ++k) {
                const int la = row + 32 * (k >> 5), lb = col + 32 * (k >> 5), j = k & 31;
                ref += (double)fp6_value(ca[la * 32 + j]) * ldexp(1.0, hsa[la] - 127) * fp6_value(cb[lb * 32 + j]) * ldexp(1.0, hsb[lb] - 127);
            }
            maxerr = fmax(maxerr, fabs(ref - hd[lane * 16 + reg])); maxref = fmax(maxref, fabs(ref));
        }
    printf("fp6 MFMA layout check (little-endian 6-bit fields in v[0:5], lanes 32-63 = K 32-63, per-lane E8M0 scales, garbage in v[6:7]): max |D - ref| = %.3e (max |ref| %.3e)\n", maxerr, maxref);
    return 0;
}
