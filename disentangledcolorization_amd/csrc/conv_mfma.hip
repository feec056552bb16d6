// conv_mfma.hip — 3x3 convolution as an implicit GEMM on the CDNA4 matrix cores (K1 of SURVEY §2b).
//
// Replaces every nn.Conv2d(k=3,pad=1,stride=1|2) on the hot path (models/network.py:14,19,70,86-87,
// 134,152-201,243,282) together with what the reference runs as separate ATen ops around it:
// bias, ReLU/LeakyReLU, eval-mode BatchNorm (after the activation), residual add, nearest x2
// upsample of the input (nn.Upsample / F.interpolate) and channel concat (torch.cat) — the last two
// are folded into the input staging ("on read"), the others into the epilogue.
//
// Numerics: activations and weights are fp16 hi/lo pairs (x = hi + lo).  DISCO_PREC_F16X3 issues
// three MFMAs per k-block (hi*hi + hi*lo + lo*hi, fp32 accumulate), which carries ~22 mantissa bits
// per operand — fp32-class results at 1/3 of the fp16 MFMA rate (5x the fp32 MFMA rate).
// DISCO_PREC_F16X1 uses the hi parts only.
//
// Tiling (one workgroup = 4 waves = 256 threads):
//   output tile  TH x TW pixels (BM = 256 or 128 GEMM rows) x BN = 32*NT output channels
//   K loop       input channels in chunks of 16 (one MFMA k-block); per chunk the (TH*s+2)x(TW*s+2)
//                input halo tile and the 9 taps' weights of the chunk are staged in LDS once and
//                the 9 taps are contracted from LDS (the halo tile is re-used 9x from LDS, not HBM)
//   wave w       owns M-blocks [w*MT, (w+1)*MT) x all NT N-blocks; v_mfma_f32_32x32x16_f16
//   LDS          A: [plane][pixel][16 ch + 16 B pad] (48 B pitch: conflict-free ds_read_b128 rows)
//                W: [nt][tap][plane][lane][8]  (already in fragment order in HBM: linear copy)
//   2 workgroups per CU (<= 80 KB LDS, <= 128 VGPR+AGPR) overlap one group's staging with the
//   other's MFMA phase.
#include "common.h"

namespace disco {

namespace {

constexpr int PITCH = 48;              // bytes per pixel per plane in LDS
constexpr int WBLK = 1024;             // one B fragment block: 64 lanes x 16 B
constexpr int W_NB = 9 * 2 * WBLK;     // bytes per (32-cout block, 16-cin chunk): 9 taps x {hi,lo}

template <int TW, int TH, int STRIDE>
struct Geo {
    static constexpr int MB = TW * TH / 32;
    static constexpr int MT = MB / 4;
    static constexpr int TWI = (TW - 1) * STRIDE + 3;
    static constexpr int THI = (TH - 1) * STRIDE + 3;
    static constexpr int NPIX = TWI * THI;
    static constexpr int A_PLANE = ((NPIX * PITCH + 15) / 16) * 16;
    static constexpr int ROWS_PER_MB = 32 / TW;
    static_assert(MB % 4 == 0, "4 waves split the M blocks");
    static_assert(32 % TW == 0, "an M block covers whole tile rows");
};

template <int TW, int TH, int NT, int STRIDE, bool X3>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const ConvArgs a) {
    using G = Geo<TW, TH, STRIDE>;
    constexpr int MT = G::MT;
    constexpr int NPLANE = X3 ? 2 : 1;
    constexpr int A_BYTES = NPLANE * G::A_PLANE;
    constexpr int NUNITS = NPLANE * G::NPIX * 2;         // 16-byte units of the halo tile
    constexpr int UPT = (NUNITS + 255) / 256;            // units per thread

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sW = smem + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (a.w_out + TW - 1) / TW, tiles_y = (a.h_out + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int ox0 = tx * TW, oy0 = ty * TH;
    const int ix0 = ox0 * STRIDE - 1, iy0 = oy0 * STRIDE - 1;

    // ---- staging descriptors: LDS byte offset and in-image element offset (or -1) per unit -------
    int loff[UPT], goff[UPT];
    int cur_src = -1;
    const f16* src_img = nullptr;   // image base of the current source (hi plane)
    long src_plane = 0;
    int src_c0 = 0;                 // first channel of the current source in the concatenated input
    auto setup_source = [&](int si) {
        const ConvSrc& sp = a.src[si];
        cur_src = si;
        src_img = sp.p + (size_t)n * sp.h * sp.w * sp.c;
        src_plane = sp.plane;
        src_c0 = si == 0 ? 0 : a.src[0].c;
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int u = tid + i * 256;
            const int part = u & 1;
            const int pp = u >> 1;
            const int plane = pp / G::NPIX, pix = pp - plane * G::NPIX;
            const int py = pix / G::TWI, px = pix - py * G::TWI;
            const int gy = iy0 + py, gx = ix0 + px;
            loff[i] = u < NUNITS ? plane * G::A_PLANE + pix * PITCH + part * 16 : -1;
            const bool in = u < NUNITS && gy >= 0 && gy < a.h_in && gx >= 0 && gx < a.w_in;
            goff[i] = in ? (((gy >> sp.up) * sp.w + (gx >> sp.up)) * sp.c + part * 8) | (plane << 30) : -1;
        }
    };

    // ---- per-lane fragment addressing ---------------------------------------------------------------
    const int r = lane & 31, kh = lane >> 5;
    const int lox = r % TW, loy = r / TW;
    const int a_off = ((wave * MT * G::ROWS_PER_MB + loy) * STRIDE * G::TWI + lox * STRIDE) * PITCH + kh * 16;
    const int w_off = lane * 16;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nchunks = a.c_in >> 4;
    const char* wbase = reinterpret_cast<const char*>(a.w) + (size_t)(blockIdx.y * NT) * nchunks * W_NB;

    for (int ck = 0; ck < nchunks; ++ck) {
        int c0 = ck << 4;
        const int si = (a.nsrc > 1 && c0 >= a.src[0].c) ? 1 : 0;
        if (si != cur_src) setup_source(si);
        c0 -= src_c0;
        // global loads first (latency overlaps the previous chunk's tail on the partner workgroup)
        uint4 av[UPT];
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            av[i] = make_uint4(0, 0, 0, 0);
            if (goff[i] >= 0) {
                const int plane = goff[i] >> 30, off = goff[i] & 0x3fffffff;
                av[i] = *reinterpret_cast<const uint4*>(src_img + plane * src_plane + off + c0);
            }
        }
        constexpr int WU = NT * W_NB / 16;                // 16-byte units of the weight tile
        constexpr int WPT = (WU + 255) / 256;
        uint4 wv[WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int u = tid + i * 256;
            if (u < WU) {
                const int nt = u / (W_NB / 16), q = u - nt * (W_NB / 16);
                wv[i] = reinterpret_cast<const uint4*>(wbase + ((size_t)nt * nchunks + ck) * W_NB)[q];
            }
        }
        __syncthreads();   // everyone finished reading the previous chunk's tiles
#pragma unroll
        for (int i = 0; i < UPT; ++i)
            if (loff[i] >= 0) *reinterpret_cast<uint4*>(sA + loff[i]) = av[i];
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int u = tid + i * 256;
            if (u < WU) reinterpret_cast<uint4*>(sW)[u] = wv[i];
        }
        __syncthreads();

#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int off = a_off + ((mt * G::ROWS_PER_MB * STRIDE + ky) * G::TWI + kx) * PITCH;
                ah[mt] = *reinterpret_cast<const f16x8*>(sA + off);
                if (X3) al[mt] = *reinterpret_cast<const f16x8*>(sA + G::A_PLANE + off);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int off = w_off + nt * W_NB + tap * 2 * WBLK;
                bh[nt] = *reinterpret_cast<const f16x8*>(sW + off);
                if (X3) bl[nt] = *reinterpret_cast<const f16x8*>(sW + off + WBLK);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (X3) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    }
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                }
        }
    }

    // ---- epilogue: bias (+res) -> activation -> BN affine -> hi/lo split -> NHWC store -----------
    const int cpad = a.c_out_pad;
    f16* out_img = a.out + (size_t)n * a.h_out * a.w_out * cpad;
    const f16* res_img = a.res ? a.res + (size_t)n * a.h_out * a.w_out * cpad : nullptr;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (blockIdx.y * NT + nt) * 32 + r;
        const bool cok = co < a.c_out;
        const float bias = (cok && a.bias) ? a.bias[co] : 0.f;
        const float bsc = (cok && a.bn_scale) ? a.bn_scale[co] : 1.f;
        const float bsh = (cok && a.bn_shift) ? a.bn_shift[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (e & 3) + 8 * (e >> 2) + 4 * kh;
                const int px = m % TW, py = (wave * MT + mt) * G::ROWS_PER_MB + m / TW;
                const int oy = oy0 + py, ox = ox0 + px;
                if (cok && oy < a.h_out && ox < a.w_out) {
                    const size_t idx = ((size_t)oy * a.w_out + ox) * cpad + co;
                    float v = acc[mt][nt][e] + bias;
                    if (res_img) v += (float)res_img[idx] + (float)res_img[idx + a.res_plane];
                    if (a.act == DISCO_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (a.act == DISCO_ACT_LRELU) v = v >= 0.f ? v : v * a.slope;
                    else if (a.act == DISCO_ACT_TANH) v = tanhf(v);
                    v = v * bsc + bsh;
                    const f16 hi = (f16)v;
                    out_img[idx] = hi;
                    out_img[idx + a.out_plane] = (f16)(v - (float)hi);
                }
            }
        }
    }
}

template <int TW, int TH, int NT, int STRIDE, bool X3>
int launch_cfg(const ConvArgs& a, hipStream_t s) {
    using G = Geo<TW, TH, STRIDE>;
    constexpr int smem = (X3 ? 2 : 1) * G::A_PLANE + NT * W_NB;
    auto kern = conv3x3_mfma_kernel<TW, TH, NT, STRIDE, X3>;
    static bool attr_set = false;   // idempotent; a race only repeats the call
    if (!attr_set) {
        DISCO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int tiles = cdiv(a.w_out, TW) * cdiv(a.h_out, TH) * a.n;
    dim3 grid(tiles, cdiv(a.c_out, 32 * NT));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a);
    DISCO_LAUNCH_CHECK("conv3x3_mfma_kernel");
    return DISCO_OK;
}

template <bool X3>
int dispatch(const ConvArgs& a, hipStream_t s) {
    const bool wide = a.w_out > 16;
    const bool nt2 = a.c_out > 32;
    if (a.stride == 1) {
        if (wide) return nt2 ? launch_cfg<32, 8, 2, 1, X3>(a, s) : launch_cfg<32, 8, 1, 1, X3>(a, s);
        return nt2 ? launch_cfg<16, 16, 2, 1, X3>(a, s) : launch_cfg<16, 16, 1, 1, X3>(a, s);
    }
    if (wide) return nt2 ? launch_cfg<32, 4, 2, 2, X3>(a, s) : launch_cfg<32, 4, 1, 2, X3>(a, s);
    return nt2 ? launch_cfg<16, 8, 2, 2, X3>(a, s) : launch_cfg<16, 8, 1, 2, X3>(a, s);
}

}  // namespace

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

int launch_conv3x3(const ConvArgs& a, hipStream_t s) {
    if (a.stride != 1 && a.stride != 2) { set_error("conv3x3: stride %d", a.stride); return DISCO_ESHAPE; }
    if (a.c_in % 16 || a.src[0].c % 16 || (a.nsrc > 1 && a.src[1].c % 16)) {
        set_error("conv3x3: input channels must be multiples of 16 (got %d)", a.c_in);
        return DISCO_ESHAPE;
    }
    if (a.c_out > 32 && a.c_out % 64) { set_error("conv3x3: c_out %d (>32) must be a multiple of 64", a.c_out); return DISCO_ESHAPE; }
    for (int i = 0; i < a.nsrc; ++i) {
        if ((size_t)a.src[i].h * a.src[i].w * a.src[i].c >= (1u << 30)) {
            set_error("conv3x3: image too large for 30-bit in-image offsets");
            return DISCO_ESHAPE;
        }
    }
    return a.precision == DISCO_PREC_F16X1 ? dispatch<false>(a, s) : dispatch<true>(a, s);
}

}  // namespace disco
