// color.hip — sRGB <-> CIELab on the device (the image I/O either side of the hot path).
//
// Restates the reference's torch implementation models/basic.py:395-475 (rgb2xyz, xyz2lab, lab2xyz, xyz2rgb,
// rgb2lab, lab2rgb; D65 white 0.95047/1/1.08883, sRGB gamma 2.4) as one fused elementwise kernel per direction:
// the reference runs ~25 ATen ops and materialises 10 intermediates per call.  main/colorizer/inference.py itself
// goes through cv2 (COLOR_RGB2LAB on float32), which is not available offline; the torch functions are the
// importable definition of the same transform and are what the golden vectors pin.
#include "common.h"

namespace disco {

namespace {

// rgb in [0,1] -> normalised Lab ((L-50)/50, a/110, b/110)   (basic.py:395-437)
__device__ inline void rgb2lab_px(const float r[3], float lab[3]) {
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = r[k] > 0.04045f ? powf((r[k] + 0.055f) / 1.055f, 2.4f) : r[k] / 12.92f;
    float x = 0.412453f * c[0] + 0.357580f * c[1] + 0.180423f * c[2];
    float y = 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2];
    float z = 0.019334f * c[0] + 0.119193f * c[1] + 0.950227f * c[2];
    x /= 0.95047f; z /= 1.08883f;
    auto f = [](float u) { return u > 0.008856f ? powf(u, 1.f / 3.f) : 7.787f * u + 16.f / 116.f; };
    const float fx = f(x), fy = f(y), fz = f(z);
    lab[0] = ((116.f * fy - 16.f) - 50.f) / 50.f;
    lab[1] = 500.f * (fx - fy) / 110.f;
    lab[2] = 200.f * (fy - fz) / 110.f;
}

// normalised Lab -> rgb (>= 0, not clipped above)   (basic.py:439-475)
__device__ inline void lab2rgb_px(const float lab[3], float rgb[3]) {
    const float L = lab[0] * 50.f + 50.f, a = lab[1] * 110.f, b = lab[2] * 110.f;
    const float fy = (L + 16.f) / 116.f;
    const float fx = a / 500.f + fy;
    const float fz = fmaxf(0.f, fy - b / 200.f);
    auto g = [](float u) { return u > 0.2068966f ? u * u * u : (u - 16.f / 116.f) / 7.787f; };
    const float x = g(fx) * 0.95047f, y = g(fy), z = g(fz) * 1.08883f;
    float c[3];
    c[0] = 3.24048134f * x - 1.53715152f * y - 0.49853633f * z;
    c[1] = -0.96925495f * x + 1.87599f * y + 0.04155593f * z;
    c[2] = 0.05564664f * x - 0.20404134f * y + 1.05731107f * z;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = fmaxf(c[k], 0.f);
        rgb[k] = v > 0.0031308f ? 1.055f * powf(v, 1.f / 2.4f) - 0.055f : 12.92f * v;
    }
}

__global__ void rgb2lab_kernel(const float* __restrict__ rgb, float* __restrict__ lab, long total, long hw) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long img = t / hw, p = t - img * hw;
        const float* s = rgb + img * 3 * hw + p;
        const float r[3] = {s[0], s[hw], s[2 * hw]};
        float l[3];
        rgb2lab_px(r, l);
        float* o = lab + img * 3 * hw + p;
        o[0] = l[0]; o[hw] = l[1]; o[2 * hw] = l[2];
    }
}

__global__ void lab2rgb_kernel(const float* __restrict__ lab, float* __restrict__ rgb, long total, long hw) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long img = t / hw, p = t - img * hw;
        const float* s = lab + img * 3 * hw + p;
        const float l[3] = {s[0], s[hw], s[2 * hw]};
        float r[3];
        lab2rgb_px(l, r);
        float* o = rgb + img * 3 * hw + p;
        o[0] = r[0]; o[hw] = r[1]; o[2 * hw] = r[2];
    }
}

// fetch_data (main/colorizer/inference.py:23-42) after the decode: uint8 RGB (n,H,W,3) -> edge-padded to (Hp,Wp),
// /255, RGB->Lab, split into gray = (L-50)/50 (n,1,Hp,Wp), ab/110 (n,2,Hp,Wp) and rgb*2-1 (n,3,Hp,Wp).
__global__ void rgb8_to_lab_kernel(const unsigned char* __restrict__ src, float* __restrict__ gray, float* __restrict__ ab,
                                   float* __restrict__ rgbn, int n, int H, int W, int Hp, int Wp) {
    const long hw = (long)Hp * Wp, total = (long)n * hw;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long img = t / hw, p = t - img * hw;
        const int y = (int)(p / Wp), x = (int)(p - (long)y * Wp);
        const int sy = y < H ? y : H - 1, sx = x < W ? x : W - 1;          // np.pad(mode='edge') on the bottom / right
        const unsigned char* q = src + ((img * H + sy) * W + sx) * 3;
        float r[3], l[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) r[k] = (float)((double)q[k] / 255.0);   // np.array(rgb / 255., np.float32)
        rgb2lab_px(r, l);
        gray[img * hw + p] = l[0];
        ab[img * 2 * hw + p] = l[1]; ab[img * 2 * hw + hw + p] = l[2];
        if (rgbn) {
#pragma unroll
            for (int k = 0; k < 3; ++k) rgbn[(img * 3 + k) * hw + p] = r[k] * 2.f - 1.f;
        }
    }
}

// The DEFAULT input path of main/colorizer/inference.py:32-36: cv2.resize(rgb, (256,256), interpolation=INTER_LINEAR) on the
// uint8 image, then /255 and RGB->Lab.  cv2 is a third-party dependency (opencv-python==4.6.0.66, environment.yaml:89) that is
// not available offline; this restates its published algorithm for 8-bit images (modules/imgproc/src/resize.cpp):
//   * exact 2x downscale in both directions: INTER_LINEAR is replaced by the area path, dst = (a + b + c + d + 2) >> 2;
//   * otherwise, per axis: f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s; x: clamped to the image (f = 0 at the borders);
//     y: f is kept and the two ROW indices are clipped instead (the invoker's clip(sy + k, 0, ssize.height)): at the top / bottom border
//     both rows are the same row, still weighted b0 and b1 - one LSB less than a single weight of 2048 in places (vertical upscales);
//     coefficients (1-f, f) are rounded to 11-bit fixed point (x 2048, round half to even); the horizontal pass keeps
//     S[s] a0 + S[s+1] a1 as a 32-bit integer; the vertical pass is
//     dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2        (VResizeLinear<uchar,...>, bit-exact integer math)
// with scale = 1 / (dst / src) in double precision.  One thread per destination pixel; the Lab conversion is fused in.
struct ResizeAxis { int s0, s1; int c0, c1; };
template <bool CLAMP_F>
__device__ inline ResizeAxis resize_axis(int d, int n_src, int n_dst) {
    const double scale = 1.0 / ((double)n_dst / (double)n_src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (CLAMP_F) {          // the x loop of cv::resize
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
    }
    ResizeAxis a;
    a.s0 = min(max(s, 0), n_src - 1); a.s1 = min(max(s + 1, 0), n_src - 1);
    a.c0 = __float2int_rn((1.f - f) * 2048.f);
    a.c1 = __float2int_rn(f * 2048.f);
    return a;
}

__global__ void rgb8_resize_to_lab_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ resized,
                                          float* __restrict__ gray, float* __restrict__ ab, float* __restrict__ rgbn, int n,
                                          int H, int W, int Ho, int Wo) {
    const long hw = (long)Ho * Wo, total = (long)n * hw;
    const bool area2 = H == 2 * Ho && W == 2 * Wo;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long img = t / hw, p = t - img * hw;
        const int y = (int)(p / Wo), x = (int)(p - (long)y * Wo);
        const unsigned char* im = src + img * (long)H * W * 3;
        int v[3];
        if (area2) {
            const unsigned char* q = im + ((long)(2 * y) * W + 2 * x) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = (q[k] + q[3 + k] + q[(long)W * 3 + k] + q[(long)W * 3 + 3 + k] + 2) >> 2;
        } else {
            const ResizeAxis ax = resize_axis<true>(x, W, Wo), ay = resize_axis<false>(y, H, Ho);
            const unsigned char* r0 = im + (long)ay.s0 * W * 3;
            const unsigned char* r1 = im + (long)ay.s1 * W * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int h0 = r0[ax.s0 * 3 + k] * ax.c0 + r0[ax.s1 * 3 + k] * ax.c1;
                const int h1 = r1[ax.s0 * 3 + k] * ax.c0 + r1[ax.s1 * 3 + k] * ax.c1;
                v[k] = (((ay.c0 * (h0 >> 4)) >> 16) + ((ay.c1 * (h1 >> 4)) >> 16) + 2) >> 2;
            }
        }
        float r[3], l[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (resized) resized[t * 3 + k] = (unsigned char)v[k];
            r[k] = (float)((double)v[k] / 255.0);
        }
        rgb2lab_px(r, l);
        gray[img * hw + p] = l[0];
        ab[img * 2 * hw + p] = l[1]; ab[img * 2 * hw + hw + p] = l[2];
        if (rgbn) {
#pragma unroll
            for (int k = 0; k < 3; ++k) rgbn[(img * 3 + k) * hw + p] = r[k] * 2.f - 1.f;
        }
    }
}

// save_normLabs_from_batch (utils/util.py:91-106) before the encode, with batch_depadding folded in: normalised Lab
// (n,3,Hp,Wp) -> Lab->RGB -> (rgb*255).astype(uint8) -> (n,H,W,3), top-left crop.  Values above 1 saturate at 255
// (numpy's float->uint8 cast of out-of-range values is undefined; cv2's LAB2RGB clips to [0,1] before it).
__global__ void lab_to_rgb8_kernel(const float* __restrict__ lab, unsigned char* __restrict__ dst, int n, int Hp, int Wp,
                                   int H, int W) {
    const long hw = (long)Hp * Wp, total = (long)n * H * W;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long img = t / ((long)H * W), p = t - img * (long)H * W;
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
        const float* s = lab + img * 3 * hw + (long)y * Wp + x;
        const float l[3] = {s[0], s[hw], s[2 * hw]};
        float r[3];
        lab2rgb_px(l, r);
        unsigned char* o = dst + t * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = (unsigned char)fminf(r[k] * 255.f, 255.f);   // truncation like astype(np.uint8)
    }
}

// basic.mark_color_hints (models/basic.py:95-117) with dilate_seeds (unfold -> max -> fold = a k x k max filter with
// zero padding): anchors (gate > 0.7) keep the target colours in a k x k centre and get a 1-pixel white, colourless
// margin; elsewhere base_ABs (or zero colour / the input gray when base_ABs is None).  Comparisons only: bit-exact.
__global__ void mark_hints_kernel(const float* __restrict__ gray, const float* __restrict__ target, const float* __restrict__ gate,
                                  const float* __restrict__ base, float* __restrict__ out, int n, int H, int W, int ks) {
    const long hw = (long)H * W, total = (long)n * hw;
    const int rc = ks / 2, rm = (ks + 2) / 2;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long img = t / hw, p = t - img * hw;
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
        const float* g = gate + img * hw;
        float center = 0.f, wide = 0.f;          // zero padding of F.unfold takes part in the max
        for (int dy = -rm; dy <= rm; ++dy)
            for (int dx = -rm; dx <= rm; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const float b = g[(long)yy * W + xx] > 0.7f ? 1.f : 0.f;
                wide = fmaxf(wide, b);
                if (dy >= -rc && dy <= rc && dx >= -rc && dx <= rc) center = fmaxf(center, b);
            }
        const float margin = wide - center;
        float* o = out + img * 3 * hw + p;
        o[0] = margin > 1e-5f ? 1.f : gray[img * hw + p];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float tv = target[(img * 2 + k) * hw + p];
            float v;
            if (!base) v = center < 1e-5f ? 0.f : tv;
            else { v = margin > 1e-5f ? 0.f : base[(img * 2 + k) * hw + p]; if (center > 1e-5f) v = tv; }
            o[(k + 1) * hw] = v;
        }
    }
}

inline int grid_for(long total) { long g = (total + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

}  // namespace

int launch_rgb2lab(const float* rgb, float* lab, long npix_total, long hw, hipStream_t s) {
    hipLaunchKernelGGL(rgb2lab_kernel, dim3(grid_for(npix_total)), dim3(256), 0, s, rgb, lab, npix_total, hw);
    DISCO_LAUNCH_CHECK("rgb2lab_kernel");
    return DISCO_OK;
}

int launch_lab2rgb(const float* lab, float* rgb, long npix_total, long hw, hipStream_t s) {
    hipLaunchKernelGGL(lab2rgb_kernel, dim3(grid_for(npix_total)), dim3(256), 0, s, lab, rgb, npix_total, hw);
    DISCO_LAUNCH_CHECK("lab2rgb_kernel");
    return DISCO_OK;
}

int launch_rgb8_to_lab(const unsigned char* src, float* gray, float* ab, float* rgbn, int n, int H, int W, int Hp, int Wp,
                       hipStream_t s) {
    if (n < 1 || H < 1 || W < 1 || Hp < H || Wp < W) { set_error("rgb8_to_lab: bad sizes %dx%dx%d -> %dx%d", n, H, W, Hp, Wp); return DISCO_ESHAPE; }
    hipLaunchKernelGGL(rgb8_to_lab_kernel, dim3(grid_for((long)n * Hp * Wp)), dim3(256), 0, s, src, gray, ab, rgbn, n, H, W, Hp, Wp);
    DISCO_LAUNCH_CHECK("rgb8_to_lab_kernel");
    return DISCO_OK;
}

int launch_rgb8_resize_to_lab(const unsigned char* src, unsigned char* resized, float* gray, float* ab, float* rgbn, int n, int H, int W,
                              int Ho, int Wo, hipStream_t s) {
    if (n < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1 || H > 32768 || W > 32768) { set_error("rgb8_resize_to_lab: bad sizes %dx%dx%d -> %dx%d", n, H, W, Ho, Wo); return DISCO_ESHAPE; }
    hipLaunchKernelGGL(rgb8_resize_to_lab_kernel, dim3(grid_for((long)n * Ho * Wo)), dim3(256), 0, s, src, resized, gray, ab, rgbn, n, H, W, Ho, Wo);
    DISCO_LAUNCH_CHECK("rgb8_resize_to_lab_kernel");
    return DISCO_OK;
}

int launch_lab_to_rgb8(const float* lab, unsigned char* dst, int n, int Hp, int Wp, int H, int W, hipStream_t s) {
    if (n < 1 || H < 1 || W < 1 || Hp < H || Wp < W) { set_error("lab_to_rgb8: bad sizes %dx%dx%d -> %dx%d", n, Hp, Wp, H, W); return DISCO_ESHAPE; }
    hipLaunchKernelGGL(lab_to_rgb8_kernel, dim3(grid_for((long)n * H * W)), dim3(256), 0, s, lab, dst, n, Hp, Wp, H, W);
    DISCO_LAUNCH_CHECK("lab_to_rgb8_kernel");
    return DISCO_OK;
}

int launch_mark_hints(const float* gray, const float* target, const float* gate, const float* base, float* out, int n, int H,
                      int W, int ks, hipStream_t s) {
    if (ks < 1 || ks > 15 || !(ks & 1)) { set_error("mark_color_hints: kernel_size %d (odd, 1..15)", ks); return DISCO_ESHAPE; }
    hipLaunchKernelGGL(mark_hints_kernel, dim3(grid_for((long)n * H * W)), dim3(256), 0, s, gray, target, gate, base, out, n, H, W, ks);
    DISCO_LAUNCH_CHECK("mark_hints_kernel");
    return DISCO_OK;
}

}  // namespace disco
