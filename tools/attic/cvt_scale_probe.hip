// Probe of gfx950's v_cvt_scalef32_pk_fp8_f32 (and of the unscaled v_cvt_pk_fp8_f32 above the finite range): does
//     cvt_scalef32(x, scale = 2^-k)  ==  cvt_pk_fp8( clamp(x 2^k, +-448) )      (the conv epilogue's mul + v_med3 + cvt)
// bit for bit - direction of the scale, rounding (nearest even), subnormals, saturation, NaN/Inf - so that the epilogue can
// drop the multiply and the clamp.  Output: profiles/r03_cvt_scale_probe.txt
//   hipcc --offload-arch=gfx950 -O3 tools/cvt_scale_probe.hip -o /tmp/cvt_scale_probe && /tmp/cvt_scale_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef short s16x2 __attribute__((ext_vector_type(2)));

__global__ void k_scaled(const float* x, unsigned* out, float scale, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 >= n) return;
    s16x2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[4 * i], x[4 * i + 1], scale, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[4 * i + 2], x[4 * i + 3], scale, true);
    out[i] = __builtin_bit_cast(unsigned, r);
}
__global__ void k_plain(const float* x, unsigned* out, float mul, int clamp, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 >= n) return;
    float v[4];
    for (int j = 0; j < 4; ++j) { v[j] = x[4 * i + j] * mul; if (clamp) v[j] = __builtin_amdgcn_fmed3f(v[j], -448.f, 448.f); }
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
    out[i] = (unsigned)w;
}

static unsigned char host_fp8(float x) {      // OCP e4m3fn, round to nearest even, saturating (conv_mx.hip's fp8_e4m3_from_float)
    const unsigned char sign = std::signbit(x) ? 0x80 : 0;
    float ax = std::fabs(x);
    if (!(ax == ax)) return 0x7f;
    if (ax >= 448.f) return sign | 0x7e;
    if (ax < std::ldexp(1.f, -10)) return sign;
    int e; std::frexp(ax, &e);
    int ex = e - 1;
    if (ex < -6) ex = -6;
    const float q = std::ldexp(ax, 3 - ex);
    float rq = std::nearbyint(q);
    if (rq >= 16.f) { rq = 8.f; ++ex; }
    if (rq < 8.f) return sign | (unsigned char)rq;
    return sign | (unsigned char)(((ex + 7) << 3) | ((int)rq - 8));
}

int main() {
    std::vector<float> x;
    // every fp8 code's value, the midpoints between neighbouring codes (ties), and values just off the ties
    for (int c = 0; c < 127; ++c) {
        const int e = (c >> 3) & 15, m = c & 7;
        const float v = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.f + m / 8.f, e - 7);
        const int e2 = ((c + 1) >> 3) & 15, m2 = (c + 1) & 7;
        const float v2 = e2 == 0 ? std::ldexp((float)m2, -9) : std::ldexp(1.f + m2 / 8.f, e2 - 7);
        const float mid = 0.5f * (v + v2);
        for (float t : {v, mid, std::nextafterf(mid, 0.f), std::nextafterf(mid, 1e9f)}) { x.push_back(t); x.push_back(-t); }
    }
    for (float t : {448.f, 449.f, 464.f, 465.f, 480.f, 512.f, 1000.f, 1e6f, 3e38f, 1e-3f, 1e-4f, 9.765625e-4f, 4.8828125e-4f, 1e-30f, 0.f}) { x.push_back(t); x.push_back(-t); }
    x.push_back(INFINITY); x.push_back(-INFINITY); x.push_back(NAN); x.push_back(-0.f);
    unsigned st = 12345u;
    for (int i = 0; i < 100000; ++i) {          // random magnitudes over 2^-14 .. 2^10
        st = st * 1664525u + 1013904223u; const float u = (st >> 8) * (1.f / 16777216.f);
        st = st * 1664525u + 1013904223u; const float g = (st >> 8) * (1.f / 16777216.f);
        x.push_back((g < 0.5f ? -1.f : 1.f) * std::ldexp(1.f + u, (int)(g * 48.f) % 24 - 14));
    }
    while (x.size() % 4) x.push_back(0.f);
    const int n = (int)x.size();
    float* dx; unsigned *d0, *d1;
    hipMalloc(&dx, n * 4); hipMalloc(&d0, n); hipMalloc(&d1, n);
    std::vector<unsigned char> r0(n), r1(n);
    auto fp8v = [](unsigned char c) { const int s = c >> 7, e = (c >> 3) & 15, m = c & 7; const float f = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.f + m / 8.f, e - 7); return s ? -f : f; };
    for (int kexp : {0, 3, -3, 7, 11, -5}) {
        // inputs pre-divided by 2^kexp so that x 2^kexp sweeps the interesting range
        std::vector<float> xs(n);
        for (int i = 0; i < n; ++i) xs[i] = std::ldexp(x[i], -kexp);
        hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
        for (int dir = 0; dir < 2; ++dir) {
            const float scale = std::ldexp(1.f, dir ? kexp : -kexp);
            hipLaunchKernelGGL(k_scaled, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, dx, d0, scale, n);
            hipLaunchKernelGGL(k_plain, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, dx, d1, std::ldexp(1.f, kexp), 1, n);
            hipMemcpy(r0.data(), d0, n, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n, hipMemcpyDeviceToHost);
            int diff = 0, diff_host = 0, shown = 0;
            for (int i = 0; i < n; ++i) {
                const bool nan_in = !(xs[i] == xs[i]);
                if (r0[i] != r1[i]) { ++diff; if (shown < 6 && kexp && dir == 0) { printf("    x 2^k = %.9g: scaled cvt 0x%02x (%g), mul+med3+cvt 0x%02x (%g)\n", (double)std::ldexp(xs[i], kexp), r0[i], (double)fp8v(r0[i]), r1[i], (double)fp8v(r1[i])); ++shown; } }
                if (!nan_in && r0[i] != host_fp8(std::ldexp(xs[i], kexp))) ++diff_host;
            }
            printf("k = %3d, scale operand = 2^%d: scaled cvt vs mul+med3+cvt: %d of %d differ; vs host RNE-saturating: %d differ\n", kexp, dir ? kexp : -kexp, diff, n, diff_host);
            if (kexp == 0) break;
        }
    }
    // the unscaled conversion above the finite range, without the clamp
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_plain, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, dx, d0, 1.f, 0, n);
    hipLaunchKernelGGL(k_scaled, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, dx, d1, 1.f, n);
    hipMemcpy(r0.data(), d0, n, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        if (std::fabs(x[i]) >= 448.f || !(x[i] == x[i]))
            if (i < 1100) printf("  x = %-12g  v_cvt_pk_fp8_f32 (no clamp) -> 0x%02x   v_cvt_scalef32_pk_fp8_f32 (scale 1) -> 0x%02x\n", (double)x[i], r0[i], r1[i]);
    return 0;
}
