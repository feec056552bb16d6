// conv_pack.cpp — host-side weight preparation of the f16x3 conv layers (kernel: conv_mx_kernel.h, AR = 2): the packed fragment image,
// ConvTranspose2d 4x4 s2 / upsample+3x3 as 4-phase 3x3 convs, tap masks.  (The fp16 + fp8 arithmetics pack in conv_mx.hip.)
// Until round 3 these lived next to round 1's conv kernel (conv_mfma2.hip), which is gone: every MFMA conv runs on
// conv3x3_mx_kernel.
#include <algorithm>
#include <vector>
#include "common.h"

namespace disco {

namespace {
constexpr int W_NB = 9 * 2 * 1024;          // bytes of one 16-channel chunk of one 32-cout block: 9 taps x (w_hi 1 KiB + w_lo 1 KiB)
}

size_t conv3x3_packed_bytes(int c_out, int c_in_pad) {
    return (size_t)cdiv(c_out, 32) * (c_in_pad / 16) * W_NB;
}

void conv3x3_pack_host(const float* h_w, int c_out, int c_in, const int* ci_map, int c_in_pad, void* h_packed) {
    f16* dst = reinterpret_cast<f16*>(h_packed);
    const int nb_n = cdiv(c_out, 32), nck = c_in_pad / 16;
    for (int nb = 0; nb < nb_n; ++nb)
        for (int ck = 0; ck < nck; ++ck)
            for (int tap = 0; tap < 9; ++tap)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = nb * 32 + (lane & 31);
                        const int cip = ck * 16 + (lane >> 5) * 8 + j;
                        const int ci = ci_map ? ci_map[cip] : (cip < c_in ? cip : -1);
                        float w = 0.f;
                        if (co < c_out && ci >= 0) w = h_w[((size_t)co * c_in + ci) * 9 + tap];
                        const f16 hi = (f16)w;
                        const f16 lo = (f16)(w - (float)hi);
                        const size_t base = (((size_t)nb * nck + ck) * 9 + tap) * 2 * 512 + lane * 8 + j;
                        dst[base] = hi;
                        dst[base + 512] = lo;
                    }
}
void deconv_as_conv3x3_host(const float* w, int c_in, int c_out, float* out) {
    // out[2a+py, 2b+px] = sum_{ky,kx} in[i,j] W[ci,co,ky,kx] with 2i-1+ky = 2a+py  =>  i = a + dy where
    //   py = 0: ky = 1 -> dy = 0,  ky = 3 -> dy = -1;      py = 1: ky = 0 -> dy = +1,  ky = 2 -> dy = 0   (same in x)
    const size_t total = (size_t)4 * c_out * c_in * 9;
    for (size_t i = 0; i < total; ++i) out[i] = 0.f;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px)
            for (int ky = 0; ky < 4; ++ky) {
                if (((ky + 1) & 1) != py) continue;          // ky parity must match: ky = py+1 (mod 2)
                const int dy = (py + 1 - ky) / 2;            // exact: numerator even
                for (int kx = 0; kx < 4; ++kx) {
                    if (((kx + 1) & 1) != px) continue;
                    const int dx = (px + 1 - kx) / 2;
                    for (int co = 0; co < c_out; ++co)
                        for (int ci = 0; ci < c_in; ++ci)
                            out[(((size_t)((py * 2 + px) * c_out + co) * c_in + ci) * 3 + (dy + 1)) * 3 + (dx + 1)] =
                                w[(((size_t)ci * c_out + co) * 4 + ky) * 4 + kx];
                }
            }
}

void upconv_as_conv3x3_host(const float* w, int c_in, int c_out, float* out) {
    // hi-res output (2a+py, 2b+px) reads upsampled rows 2a+py+ky-1, i.e. low-res rows a+dy with
    //   py = 0: ky=0 -> dy=-1, ky=1,2 -> dy=0;      py = 1: ky=0,1 -> dy=0, ky=2 -> dy=+1      (same in x)
    const size_t total = (size_t)4 * c_out * c_in * 9;
    std::vector<double> accd(total, 0.0);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px)
            for (int ky = 0; ky < 3; ++ky) {
                const int dy = (py + ky - 1) >> 1;                 // floor((py+ky-1)/2) in {-1,0,1}
                for (int kx = 0; kx < 3; ++kx) {
                    const int dx = (px + kx - 1) >> 1;
                    for (int co = 0; co < c_out; ++co)
                        for (int ci = 0; ci < c_in; ++ci)
                            accd[(((size_t)((py * 2 + px) * c_out + co) * c_in + ci) * 3 + (dy + 1)) * 3 + (dx + 1)] +=
                                (double)w[(((size_t)co * c_in + ci) * 3 + ky) * 3 + kx];
                }
            }
    for (size_t i = 0; i < total; ++i) out[i] = (float)accd[i];
}

void conv3x3_tapmask_host(const float* w, int c_out, int c_in, uint32_t* mask) {
    const int nb_n = cdiv(c_out, 32);
    for (int nb = 0; nb < nb_n; ++nb) {
        uint32_t m = 0;
        for (int co = nb * 32; co < std::min(c_out, nb * 32 + 32); ++co)
            for (int ci = 0; ci < c_in; ++ci)
                for (int t = 0; t < 9; ++t)
                    if (w[((size_t)co * c_in + ci) * 9 + t] != 0.f) m |= 1u << t;
        mask[nb] = m;
    }
}

}  // namespace disco
