// conv_mfma2.hip — 3x3 implicit-GEMM conv for gfx950 on fp16 hi/lo split operands ("f16x3": hi*lo + lo*hi + hi*hi, three
// v_mfma_f32_32x32x16_f16 per product, fp32 accumulate), built around the CDNA4 async copy engine:
//
//   * both operands of a 16-channel chunk (input halo tile, 9 taps of weights) travel HBM/L2 -> LDS by LDS-DMA through raw
//     buffer descriptors (raw_ptr_buffer_load_lds, 16 bytes per lane: no VGPR round trip, no ds_write pass); a lane whose
//     offset is out of range reads 0 - the hardware range check IS the conv's zero padding;
//   * LDS is double buffered: the DMA of chunk k+1 is issued in ninths between the 9 taps of chunk k and lands while the
//     waves run its MFMAs; each wave waits only for its own DMA pieces (s_waitcnt vmcnt(0)) right before the single
//     barrier of a chunk;
//   * the halo tile is stored 32 B per pixel (16 channels) with a 16-byte XOR swizzle on pixel bit 3 — applied on the DMA
//     *source* address (the LDS image of a DMA piece is lane-linear) and on the fragment read — so the 64-lane
//     ds_read_b128 of a fragment is bank-conflict free; stride-2 tiles are additionally de-interleaved by column parity
//     so consecutive output pixels read consecutive LDS pixels;
//   * activations are channel-blocked, [N][C/16][H][W][16] fp16 (hi plane, lo plane), so a tile row of one 16-channel
//     chunk is one contiguous run of 32-byte pixels: the halo DMA and the epilogue's 16-byte stores touch whole lines;
//   * 8 waves (2 per SIMD) on a 16x32-pixel x 64-channel tile where the image is large enough;
//   * persistent workgroups: a workgroup owns one tile position (and N tile) and walks over the images of the batch;
//     descriptors are computed once, and the DMA prefetch runs across image boundaries, so neither the launch of a
//     workgroup nor the first DMA round trip of a tile is exposed (they cost ~30% on the 4-chunk 64-channel layers).
//
// Round 1's conv kernel.  Since the second half of round 2 the forward's f16x3 layers run on conv_mx.hip's skeleton (its AR = 2
// mode: the same arithmetic and accumulation order, bit-identical results, none of this kernel's 95 spilled SGPRs); this file
// keeps what only it has: the hi-only mode (DISCO_PREC_F16X1), the space-to-depth packing of stride-2 layers, the per-chunk
// timing probe (tools/conv_timeline.py), the weight packers (conv3x3_pack_host & co, shared with conv_mx.hip) - and it is the
// A/B reference: DISCO_X3_OLD=1 routes every f16x3 layer through it again.
// Epilogue variants: channel-blocked fp16 hi/lo planes (default), fp32 NCHW (network outputs), depth-to-space
// (ConvTranspose2d 4x4 s2 p1 expressed as a 4-phase 3x3 conv, network.py:254-258).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>
#include "common.h"

namespace disco {

namespace {

constexpr int WBLK = 1024;
constexpr int W_NB = 9 * 2 * WBLK;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int TW, int TH, int STRIDE>
struct Geo2 {
    static constexpr int MB = TW * TH / 32;
    static constexpr int TWI = (TW - 1) * STRIDE + 3;
    static constexpr int THI = (TH - 1) * STRIDE + 3;
    static constexpr int HALF = (TWI + 1) / 2;                       // stride 2: pixels per column-parity half row
    static constexpr int PITCH = STRIDE == 1 ? TWI : 2 * HALF;       // LDS pixels per tile row
    static constexpr int NPIX = THI * PITCH;
    static constexpr int ROWS_PER_MB = 32 / TW;
    static_assert(32 % TW == 0, "an M block covers whole tile rows");
};

// MASKED: the layer has all-zero taps (sub-pixel up-conv / deconv phases, space-to-depth chunks) that are skipped at run
// time.  Plain layers (MASKED = false) get a straight-line 9-tap body - no per-tap branches - so the compiler can schedule
// fragment reads of one tap under the MFMAs of the previous one.
template <int TW, int TH, int NT, int STRIDE, bool X3, int WM, int WN, bool MASKED>
__global__ __launch_bounds__(WM * WN * 64) void conv3x3_mfma2_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type and LDS-DMA builtins exist in the device pass only
    using G = Geo2<TW, TH, STRIDE>;
    constexpr int NWAVE = WM * WN;
    constexpr int MT = G::MB / WM;             // M blocks per wave
    constexpr int NTW = NT / WN;               // N blocks per wave
    static_assert(G::MB % WM == 0 && NT % WN == 0, "tile split");
    constexpr int NPLANE = X3 ? 2 : 1;
    constexpr int PLANE_B = G::NPIX * 32;                          // bytes per plane in LDS
    constexpr int A_UNITS = NPLANE * G::NPIX * 2;
    constexpr int A_PIECES = (A_UNITS + 63) / 64;
    constexpr int A_BYTES = A_PIECES * 1024;
    constexpr int W_PIECES = NT * 18;
    constexpr int BUF_BYTES = A_BYTES + W_PIECES * 1024;
    constexpr int APW = (A_PIECES + NWAVE - 1) / NWAVE;            // DMA pieces per wave and chunk
    constexpr int WPW = (W_PIECES + NWAVE - 1) / NWAVE;
    constexpr int APT = (APW + 8) / 9, WPT = (WPW + 8) / 9;        // ... issued per tap
    constexpr int PAR_OFF = 2 * BUF_BYTES;                         // epilogue parameters: [3][32*NT] floats

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int tiles_x = (a.w_out + TW - 1) / TW, tiles_y = (a.h_out + TH - 1) / TH;
    // a.s2d (STRIDE == 1 instantiations): a stride-2 conv run as a stride-1 conv over the space-to-depth VIEW of its
    // input - chunk ck = 16 channels (block ck>>2) of ONE sub-pixel phase (ck&3) of the real tensor, so the LDS halo
    // tile has the stride-1 footprint (a stride-2 tile needs 4x the LDS and ran 3 MFMAs per 4 fragment reads).
    // Phase (py,px) meets the 3x3 window in 1/2/2/4 taps: the same 9 taps per channel block, no extra MFMAs.
    const bool s2d = STRIDE == 1 && a.s2d;
    const int nchunks = a.c_in >> 4;               // s2d: the launcher passes c_in = 4 x the real channel count

    // Persistent workgroup: blockIdx.x fixes the tile position (tx, ty) and the N tile (by); the workgroup then
    // walks over the images n = blockIdx.y, blockIdx.y + gridDim.y, ... of the batch.  Everything that depends
    // on the tile position (DMA offsets, weight slice, tap mask) is computed once; per chunk only a scalar
    // byte offset advances.
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int by = bid / tiles_y;
    const int ox0 = tx * TW, oy0 = ty * TH;
    const int img_step = gridDim.y;
    int n = blockIdx.y;
    if (n >= a.n) return;

    unsigned tmask = 0x1ffu;          // taps with non-zero weights in this workgroup's N blocks (wave-uniform)
    // s2d: taps (row r, col c of the 3x3 window over the phase image) that exist for phase (py,px): r in {1} (py=0)
    // or {0,1} (py=1), same for c; bit r*3+c
    auto chunk_mask = [&](int ck) -> unsigned {
        if (!MASKED) return 0x1ffu;
        if (!s2d) return tmask;
        const unsigned ph = (unsigned)ck & 3u;
        return ph == 0 ? 0x010u : (ph == 1 ? 0x018u : (ph == 2 ? 0x012u : 0x01bu));
    };
    if (MASKED && a.tapmask) {
        tmask = 0;
#pragma unroll
        for (int j = 0; j < NT; ++j) tmask |= a.tapmask[by * NT + j];
        tmask = __builtin_amdgcn_readfirstlane(tmask);
    }

    // per-channel epilogue parameters of this workgroup's N tile -> LDS, once (the workgroup is persistent), so the
    // epilogue issues no global loads it would have to wait for
    float* s_par = reinterpret_cast<float*>(smem + PAR_OFF);
    for (int i = tid; i < 3 * 32 * NT; i += NWAVE * 64) {
        const int which = i / (32 * NT), c = i - which * (32 * NT);
        const int co = by * NT * 32 + c;
        const int cpar = a.d2s_c > 0 ? co % a.d2s_c : co;         // parameters are shared by the 4 deconv phases
        const float* src = which == 0 ? a.bias : (which == 1 ? a.bn_scale : a.bn_shift);
        s_par[i] = (src && co < a.c_out) ? src[cpar] : (which == 1 ? 1.f : 0.f);
    }
    // (visible to all waves after the first chunk's barrier)

    // ---- LDS-DMA through raw buffer descriptors: address = base + soffset(chunk, image) + voffset(lane); a lane
    // whose voffset is out of range reads 0 (hardware range check) = the conv's zero padding ----------------------
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[0].p, 0, a.src_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[a.nsrc > 1 ? 1 : 0].p, 0, a.src_bytes[a.nsrc > 1 ? 1 : 0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    unsigned voff[2][APW];            // per source: byte offset (plane included) of this lane's 16 bytes, or OOB
    {
        const int ix0 = ox0 * STRIDE - 1, iy0 = oy0 * STRIDE - 1;
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            const ConvSrc& sp = a.src[si < a.nsrc ? si : 0];
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const int u = (i * NWAVE + wave) * 64 + lane;
                const int plane = u / (G::NPIX * 2);
                const int rem = u - plane * (G::NPIX * 2);
                const int p = rem >> 1, j = rem & 1;
                const int kh = j ^ ((p >> 3) & 1);
                const int py = p / G::PITCH, q = p - py * G::PITCH;
                int px;
                bool slot_ok = u < A_UNITS;
                if (STRIDE == 1) px = q;
                else { const int par = q / G::HALF; px = (q - par * G::HALF) * 2 + par; slot_ok = slot_ok && px < G::TWI; }
                int gy = iy0 + py, gx = ix0 + px;
                bool in = slot_ok && gy >= 0 && gy < a.h_in && gx >= 0 && gx < a.w_in;
                if (s2d) {      // (gy,gx) is a pixel of the phase image: real pixel (2gy+py', 2gx+px'), phase added per chunk
                    in = slot_ok && gy >= 0 && gy < (a.h_in >> 1) && gx >= 0 && gx < (a.w_in >> 1);
                    gy *= 2; gx *= 2;
                }
                voff[si][i] = in ? (unsigned)(plane * sp.plane + ((gy >> sp.up) * sp.w + (gx >> sp.up)) * 16 + kh * 8) * 2u : OOB;
            }
        }
    }
    const int c_src0 = a.src[0].c;
    const unsigned img_b0 = (unsigned)(a.src[0].c * a.src[0].h * a.src[0].w) * 2u, blk_b0 = (unsigned)(a.src[0].h * a.src[0].w) * 32u;
    const unsigned img_b1 = (unsigned)(a.src[1].c * a.src[1].h * a.src[1].w) * 2u, blk_b1 = (unsigned)(a.src[1].h * a.src[1].w) * 32u;
    const unsigned w_tile_b = (unsigned)(by * NT) * (unsigned)nchunks * W_NB, w_nt_b = (unsigned)nchunks * W_NB;

    // one ninth (part 0..8) of the DMA of chunk `ck` of image `img` into LDS buffer `buf`; part < 0: all of it
    auto issue = [&](int img, int ck, int buf, int part) {
        const int c0 = (s2d ? ck >> 2 : ck) << 4;
        const bool s1 = a.nsrc > 1 && c0 >= c_src0;
        unsigned soff = s1 ? (unsigned)img * img_b1 + (unsigned)((c0 - c_src0) >> 4) * blk_b1
                           : (unsigned)img * img_b0 + (unsigned)(c0 >> 4) * blk_b0;
        if (s2d) soff += (unsigned)(((ck >> 1) & 1) * a.src[0].w + (ck & 1)) * 32u;     // phase (py,px) = (ck>>1 & 1, ck & 1)
        const unsigned wmask = chunk_mask(ck);
        char* dA = smem + buf * BUF_BYTES;
        char* dW = dA + A_BYTES;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            if (part >= 0 && i / APT != part) continue;
            const int piece = i * NWAVE + wave;
            if (piece < A_PIECES) {
                if (s1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_void*)(dA + piece * 1024), 16, voff[1][i], soff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_void*)(dA + piece * 1024), 16, voff[0][i], soff, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            if (part >= 0 && i / WPT != part) continue;
            const int piece = i * NWAVE + wave;
            const int nt = piece / 18, q = piece - nt * 18;
            if (piece < W_PIECES && ((wmask >> (q >> 1)) & 1u))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)(dW + piece * 1024), 16, lane * 16,
                                                         w_tile_b + (unsigned)nt * w_nt_b + (unsigned)ck * W_NB + q * 1024, 0, 0);
        }
    };

    // ---- per-lane fragment addressing ---------------------------------------------------------------------------
    const int r = lane & 31, kh = lane >> 5;
    const int lox = r % TW, loy = r / TW;
    const int p_lane = ((wm * MT * G::ROWS_PER_MB + loy) * STRIDE) * G::PITCH + lox;
    const int w_off = lane * 16 + wn * NTW * W_NB;

    issue(n, 0, 0, -1);
    int buf = 0;
    bool dma_waited = false;      // the first chunk's DMA wait of the next image is taken before the epilogue
    int dbg_n = 0;                // timing probe: chunks recorded so far

    for (;;) {
    f32x16 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int next_n = n + img_step;
    for (int ck = 0; ck < nchunks; ++ck) {
        // my DMA pieces of this chunk have landed (vmcnt also counts stores: for the first chunk of an image the
        // wait was already taken BEFORE the previous image's epilogue stores were issued, so those stores drain
        // under this chunk's MFMAs instead of stalling here)
        const bool probe = a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && dbg_n < 64;
        unsigned long long t_a = 0, t_b = 0, t_c = 0;
        if (probe) t_a = __builtin_amdgcn_s_memtime();
        if (!(ck == 0 && dma_waited)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (probe) t_b = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();                        // ... everyone's have; the previous chunk's reads are done
        if (probe) t_c = __builtin_amdgcn_s_memtime();
        // next chunk (or the first chunk of the next image: prefetch across the image boundary)
        // next chunk, or the first chunk of the next image (prefetch across the image boundary); after the very last chunk
        // of the workgroup the slot re-fetches chunk 0 of the current image into the idle buffer - harmless, and it keeps
        // the tap body free of a branch
        const bool more = ck + 1 < nchunks;
        const int dma_img = more ? n : (next_n < a.n ? next_n : n), dma_ck = more ? ck + 1 : 0;
        const char* sA = smem + buf * BUF_BYTES;
        const char* sW = sA + A_BYTES;
        buf ^= 1;
        const unsigned cmask = chunk_mask(ck);
        // Tap order: column-major (kx outer, ky inner) when an M block is one tile row and the wave owns two of them
        // (the 16x32-pixel configurations, 88% of the conv time): the pixel fragment of (mt = 1, ky) is the fragment of
        // (mt = 0, ky + 1) - row y + ky + 1, same column shift - so it stays in registers and every step loads ONE new
        // row instead of two (24 instead of 36 pixel-fragment reads per chunk, -17% LDS read traffic).
        // the tap ORDER is a property of the tile width alone (never of how the tile is split over waves): every instantiation
        // that can serve a given layer shape accumulates in the same order, so a result does not depend on the batch size
        constexpr bool COLMAJOR = STRIDE == 1 && G::ROWS_PER_MB == 1;
        constexpr bool ROWREUSE = COLMAJOR && MT == 2;
        f16x8 ah[MT], al[MT];
#pragma unroll
        for (int slot = 0; slot < 9; ++slot) {
            const int ky = COLMAJOR ? slot % 3 : slot / 3, kx = COLMAJOR ? slot / 3 : slot % 3;
            const int tap = ky * 3 + kx;
            issue(dma_img, dma_ck, buf, slot);           // one ninth of the next chunk's DMA per step (buf already flipped)
            const bool live = !MASKED || ((cmask >> tap) & 1u);   // wave-uniform: all-zero taps (sub-pixel up-conv / deconv / s2d phases) skip the MFMAs
            if (!ROWREUSE && !live) continue;
            const int tapoff = ky * G::PITCH + (STRIDE == 1 ? kx : (kx & 1) * G::HALF + (kx >> 1));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (ROWREUSE && ky > 0 && mt == 0) { ah[0] = ah[1]; al[0] = al[1]; continue; }     // held from the previous step
                const int p = p_lane + mt * G::ROWS_PER_MB * STRIDE * G::PITCH + tapoff;
                const int off = (p << 5) + ((((p >> 3) ^ kh) & 1) << 4);
                ah[mt] = *reinterpret_cast<const f16x8*>(sA + off);
                if (X3) al[mt] = *reinterpret_cast<const f16x8*>(sA + PLANE_B + off);
            }
            if (ROWREUSE && !live) continue;             // the row loads above feed the next step as well
            f16x8 bh[NTW], bl[NTW];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int off = w_off + nt * W_NB + tap * 2 * WBLK;
                bh[nt] = *reinterpret_cast<const f16x8*>(sW + off);
                if (X3) bl[nt] = *reinterpret_cast<const f16x8*>(sW + off + WBLK);
            }
            // weights as the row operand, pixels as the column operand: the accumulator tile is
            // [32 output channels][32 pixels], so a lane owns ONE pixel and 16 channels (vector stores)
            // The two waves that share a SIMD alternate MFMA priority per step (ping-pong: one issues its MFMA cluster
            // while the other fetches fragments).  With equal priorities the older half of the workgroup wins
            // arbitration, finishes ~2000 cycles early and idles at the barrier while the younger half runs a
            // single-wave tail (s_memtime probe, profiles/r01_conv_timeline.txt); a static priority only flips that.
            if ((slot + (wave >= NWAVE / 2 ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    if (X3) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[nt], ah[mt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], al[mt], acc[mt][nt], 0, 0, 0);
                    }
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], ah[mt], acc[mt][nt], 0, 0, 0);
                }
        }
        if (probe) {   // [wave][chunk][4]: before the DMA wait, after it, after the barrier, after the last MFMA was issued
            unsigned long long* d = a.dbg + ((size_t)wave * 64 + dbg_n) * 4;
            d[0] = t_a; d[1] = t_b; d[2] = t_c; d[3] = __builtin_amdgcn_s_memtime();
            ++dbg_n;
        }
    }

    // ---- epilogue: bias (+res) -> activation -> BN affine -> store ---------------------------------------------
    // Three phases: (1) all the math, in registers (results replace the accumulators); (2) wait for the next
    // image's first chunk, whose DMA was issued during the last chunk - its latency hides behind phase 1;
    // (3) the stores.  vmcnt also counts stores, so with the wait taken here the stores of this image drain
    // under the next image's first chunk instead of stalling its first barrier.
    // The code is specialised at compile time by output mode (0: channel-blocked act, 1: depth-to-space act,
    // 2: fp32 NCHW) so the common path carries no integer divisions or dead branches.
    // The epilogue's inputs are invariant across the image loop; without laundering them the compiler hoists
    // ~100 VGPRs of per-channel parameters and lane masks out of the loop and spills.
    int by_e = by, oy0_e = oy0, ox0_e = ox0;
    const float* par_e = s_par;
    asm volatile("" : "+s"(by_e), "+s"(oy0_e), "+s"(ox0_e), "+v"(par_e));
    const int act = a.act;
    const float slope = a.slope;
    const bool has_bn = a.bn_scale != nullptr;
    auto epilogue = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        // activations are channel-blocked: element (n, ch, y, x) at ((n*C/16 + ch/16)*H*W + y*W + x)*16 + ch%16
        const int oc = MODE == 1 ? a.d2s_c : a.c_out_pad;          // channels of the output tensor
        const int oh = MODE == 1 ? 2 * a.h_out : a.h_out, ow = MODE == 1 ? 2 * a.w_out : a.w_out;
        const size_t oblk = (size_t)oh * ow * 16;                  // elements per 16-channel block of one image
        const size_t oimg = (size_t)n * oc * oh * ow;
        // per 16-channel group (nt, q): first channel inside the output tensor and, for depth-to-space, the
        // sub-pixel phase: virtual channel cv goes to hi-res pixel (2oy + ph/2, 2ox + ph%2), channel cv % d2s_c,
        // ph = cv / d2s_c (d2s_c is a multiple of 16, so a group never straddles two phases)
        int cbase[NTW][2], phy[NTW][2], phx[NTW][2];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int cv = (by_e * NT + wn * NTW + nt) * 32 + 16 * q;
                if (MODE == 1) {
                    const int ph = cv / a.d2s_c;
                    cbase[nt][q] = cv - ph * a.d2s_c; phy[nt][q] = ph >> 1; phx[nt][q] = ph & 1;
                } else { cbase[nt][q] = cv; phy[nt][q] = 0; phx[nt][q] = 0; }
            }
        // element offset of this lane's pixel for M block mt and channel group (nt, q), channel offset excluded
        auto pix_off = [&](int mt, int nt, int q, bool& pok) -> size_t {
            const int px = r % TW, py = (wm * MT + mt) * G::ROWS_PER_MB + r / TW;
            const int oy = oy0_e + py, ox = ox0_e + px;
            pok = oy < a.h_out && ox < a.w_out;
            if (MODE == 1) return oimg + ((size_t)(2 * oy + phy[nt][q]) * ow + 2 * ox + phx[nt][q]) * 16;
            return oimg + ((size_t)oy * ow + ox) * 16;
        };
        // ---- phase 1: math ----
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            float bias[16], bsc[16], bsh[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                    // channels c(e) = (e&3) + 8*(e>>2) + 4*kh: 4 groups of 4
                const float4 b4 = *reinterpret_cast<const float4*>(par_e + (wn * NTW + nt) * 32 + 8 * g4 + 4 * kh);
                bias[g4 * 4] = b4.x; bias[g4 * 4 + 1] = b4.y; bias[g4 * 4 + 2] = b4.z; bias[g4 * 4 + 3] = b4.w;
            }
            if (has_bn) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {                // channels c(e) = (e&3) + 8*(e>>2) + 4*kh: 4 groups of 4
                    const int cl = (wn * NTW + nt) * 32 + 8 * g4 + 4 * kh;
                    const float4 s4 = *reinterpret_cast<const float4*>(par_e + 32 * NT + cl);
                    const float4 h4 = *reinterpret_cast<const float4*>(par_e + 64 * NT + cl);
                    bsc[g4 * 4] = s4.x; bsc[g4 * 4 + 1] = s4.y; bsc[g4 * 4 + 2] = s4.z; bsc[g4 * 4 + 3] = s4.w;
                    bsh[g4 * 4] = h4.x; bsh[g4 * 4 + 1] = h4.y; bsh[g4 * 4 + 2] = h4.z; bsh[g4 * 4 + 3] = h4.w;
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[mt][nt][e] + bias[e];
                if (MODE != 2 && a.res) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        bool pok;
                        const size_t po = pix_off(mt, nt, g4 >> 1, pok);
                        const int cc = cbase[nt][g4 >> 1] + 8 * (g4 & 1) + 4 * kh;
                        if (pok && (by_e * NT + wn * NTW + nt) * 32 + 8 * g4 + 4 * kh < a.c_out) {
                            const f16* rp = a.res + po + (size_t)(cc >> 4) * oblk + (cc & 15);
                            const f16x4 rh = *reinterpret_cast<const f16x4*>(rp);
                            const f16x4 rl = *reinterpret_cast<const f16x4*>(rp + a.res_plane);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[g4 * 4 + j] += (float)rh[j] + (float)rl[j];
                        }
                    }
                }
                if (act == DISCO_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (act == DISCO_ACT_LRELU) {            // 0 <= slope <= 1 (checked by the launcher)
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], v[e] * slope);
                } else if (act == DISCO_ACT_TANH) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = tanhf(v[e]);
                }
                if (has_bn) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] * bsc[e] + bsh[e];
                }
                if (MODE == 2 && a.softmax) {
                    // softmax over the c_out (<= 32) channels of the pixel (pred_mask0 + F.softmax(dim=1),
                    // network.py:311-312): the lane pair (l, l^32) holds the pixel's 32 channels
                    const int cob = (by_e * NT + wn * NTW + nt) * 32;
                    float mx = -INFINITY;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (cob + (e & 3) + 8 * (e >> 2) + 4 * kh < a.c_out) mx = fmaxf(mx, v[e]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    float sm = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        v[e] = cob + (e & 3) + 8 * (e >> 2) + 4 * kh < a.c_out ? expf(v[e] - mx) : 0.f;
                        sm += v[e];
                    }
                    sm += __shfl_xor(sm, 32);
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] / sm;
                }
                if (MODE != 2) {
                    // split into fp16 hi + lo, packed in pairs (v_cvt_pk_f16_f32): dword d holds channels c(2d), c(2d+1)
                    unsigned hd[8], ld[8];
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        const f32x2 v2 = {v[2 * d], v[2 * d + 1]};
                        const f16x2 h2 = __builtin_convertvector(v2, f16x2);
                        const f16x2 l2 = __builtin_convertvector(v2 - __builtin_convertvector(h2, f32x2), f16x2);
                        hd[d] = __builtin_bit_cast(unsigned, h2);
                        ld[d] = __builtin_bit_cast(unsigned, l2);
                    }
                    // v_permlane32_swap: lower.(group 1) <-> upper.(group 0), lower.(group 3) <-> upper.(group 2), so
                    // the lower half-wave holds channels 0-7 and 16-23 of its pixel, the upper half 8-15 and 24-31
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            auto sh = __builtin_amdgcn_permlane32_swap(hd[4 * q + d], hd[4 * q + 2 + d], false, false);
                            hd[4 * q + d] = sh[0]; hd[4 * q + 2 + d] = sh[1];
                            auto sl = __builtin_amdgcn_permlane32_swap(ld[4 * q + d], ld[4 * q + 2 + d], false, false);
                            ld[4 * q + d] = sl[0]; ld[4 * q + 2 + d] = sl[1];
                        }
#pragma unroll
                    for (int d = 0; d < 8; ++d) {     // park the packed words in the accumulator registers
                        acc[mt][nt][d] = __builtin_bit_cast(float, hd[d]);
                        acc[mt][nt][8 + d] = __builtin_bit_cast(float, ld[d]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mt][nt][e] = v[e];
                }
            }
        }
        // ---- phase 2: the next image's first chunk has landed ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- phase 3: stores ----
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int cob = (by_e * NT + wn * NTW + nt) * 32;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (MODE != 2) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        bool pok;
                        const size_t po = pix_off(mt, nt, q, pok);
                        if (pok && cob + 16 * q + 8 * kh < a.c_out) {       // 8 consecutive channels
                            const int cc = cbase[nt][q] + 8 * kh;
                            f16* o = a.out + po + (size_t)(cc >> 4) * oblk + (cc & 15);
                            const f32x16& t = acc[mt][nt];
                            *reinterpret_cast<float4*>(o) = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
                            *reinterpret_cast<float4*>(o + a.out_plane) = make_float4(t[8 + 4 * q], t[8 + 4 * q + 1], t[8 + 4 * q + 2], t[8 + 4 * q + 3]);
                        }
                    }
                } else {
                    const int px = r % TW, py = (wm * MT + mt) * G::ROWS_PER_MB + r / TW;
                    const int oy = oy0_e + py, ox = ox0_e + px;
                    if (oy >= a.h_out || ox >= a.w_out) continue;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int co = cob + (e & 3) + 8 * (e >> 2) + 4 * kh;
                        if (co < a.c_out) a.out_f32[(((size_t)n * a.c_out + co) * a.h_out + oy) * a.w_out + ox] = acc[mt][nt][e];
                    }
                }
            }
        }
    };
    if (a.out_f32) epilogue(std::integral_constant<int, 2>{});
    else if (a.d2s_c > 0) epilogue(std::integral_constant<int, 1>{});
    else epilogue(std::integral_constant<int, 0>{});
    dma_waited = true;

    n = next_n;
    if (n >= a.n) break;
    }   // persistent loop over the images of the batch
#endif
}

inline int num_cus() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <int TW, int TH, int NT, int STRIDE, bool X3, int WM, int WN, bool MASKED>
int launch_cfg3(const ConvArgs& a, hipStream_t s) {
    using G = Geo2<TW, TH, STRIDE>;
    constexpr int A_BYTES = (((X3 ? 2 : 1) * G::NPIX * 2 + 63) / 64) * 1024;
    constexpr int smem = 2 * (A_BYTES + NT * 18 * 1024) + 3 * 32 * NT * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_mfma2_kernel<TW, TH, NT, STRIDE, X3, WM, WN, MASKED>;
    // function attributes are per device and per kernel instantiation (this static lives in the instantiation); call_once:
    // two host threads may launch the same layer shape on one device
    static std::once_flag attr_once[DISCO_MAX_DEVICES];
    hipError_t attr_err = hipSuccess;
    std::call_once(attr_once[current_device()], [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    });
    DISCO_HIP_CHECK(attr_err);
    // grid.x = tile positions x N tiles; grid.y = image groups: enough groups to give every CU one persistent
    // workgroup (LDS-limited residency), each walking n = g, g + groups, ... over the batch
    const int combos = cdiv(a.w_out, TW) * cdiv(a.h_out, TH) * cdiv(a.c_out, 32 * NT);
    static const int persist = [] { const char* e = getenv("DISCO_PERSIST"); return e ? atoi(e) : 1; }();
    // image groups g: every workgroup walks ceil(n / g) images and the launch takes ceil(combos g / CUs) rounds of
    // workgroups, so pick the g that minimises rounds x images (ties: the smaller g, longer persistent walks).  E.g. 96
    // combos x 8 images on 256 CUs: g = 3 would run 2 rounds of 3 images, g = 8 runs 3 rounds of 1.
    int groups = a.n;
    if (persist > 0) {
        const long cus = (long)persist * num_cus();
        long best_cost = -1;
        for (int g = 1; g <= a.n; ++g) {
            const long cost = (long)cdiv((long)combos * g, cus) * cdiv(a.n, g);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; groups = g; }
        }
    }
    dim3 grid(combos, groups);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, s, a);
    DISCO_LAUNCH_CHECK("conv3x3_mfma2_kernel");
    return DISCO_OK;
}

template <int TW, int TH, int NT, int STRIDE, bool X3, int WM, int WN>
int launch_cfg2(const ConvArgs& a, hipStream_t s) {
    return (a.tapmask || a.s2d) ? launch_cfg3<TW, TH, NT, STRIDE, X3, WM, WN, true>(a, s)
                                : launch_cfg3<TW, TH, NT, STRIDE, X3, WM, WN, false>(a, s);
}

template <bool X3>
int dispatch2(const ConvArgs& a, hipStream_t s) {
    const bool wide = a.w_out > 16;
    const bool nt2 = a.c_out > 32;
    if (a.stride == 1 || a.s2d) {
        // Tile choice: the largest tile (best MFMA : LDS : DMA ratios) as long as the launch still fills the chip; small
        // batches of small feature maps (e.g. one image, 512 channels at 32 x 32: 16 workgroups with the 16x32x64 tile)
        // fall back to smaller tiles / one N block per workgroup, trading per-workgroup efficiency for parallelism.
        struct Cand { int tw, th, nt; };
        static const Cand order[6] = {{32, 16, 2}, {32, 16, 1}, {32, 8, 2}, {16, 16, 2}, {32, 8, 1}, {16, 16, 1}};
        const long fill = (long)num_cus() * 3 / 4;
        int pick = -1; long best = -1;
        for (int i = 0; i < 6; ++i) {
            const Cand& c = order[i];
            if ((c.nt == 2 && !nt2) || (c.tw == 32 && !wide) || (c.tw == 32 && c.th == 16 && a.h_out <= 8)) continue;
            const long w = (long)cdiv(a.w_out, c.tw) * cdiv(a.h_out, c.th) * cdiv(a.c_out, 32 * c.nt) * a.n;
            if (w >= fill) { pick = i; break; }
            if (w > best) { best = w; pick = i; }
        }
        switch (pick) {
            case 0: return launch_cfg2<32, 16, 2, 1, X3, 8, 1>(a, s);
            case 1: return launch_cfg2<32, 16, 1, 1, X3, 8, 1>(a, s);
            case 2: return launch_cfg2<32, 8, 2, 1, X3, 4, 2>(a, s);
            case 3: return launch_cfg2<16, 16, 2, 1, X3, 4, 2>(a, s);
            case 4: return launch_cfg2<32, 8, 1, 1, X3, 8, 1>(a, s);
            default: return launch_cfg2<16, 16, 1, 1, X3, 8, 1>(a, s);
        }
    }
    if (wide) return nt2 ? launch_cfg2<32, 4, 2, 2, X3, 4, 2>(a, s) : launch_cfg2<32, 4, 1, 2, X3, 4, 1>(a, s);
    return nt2 ? launch_cfg2<16, 8, 2, 2, X3, 4, 2>(a, s) : launch_cfg2<16, 8, 1, 2, X3, 4, 1>(a, s);
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

void conv3x3_s2d_weights_host(const float* w, int c_out, int c_in, int c_in_pad, float* out) {
    // out pixel (y,x) reads real input (2y+ky-1, 2x+kx-1) = pixel (y+dy, x+dx) of phase (py,px) with 2dy+py = ky-1:
    //   py = 0: ky = 1 (dy = 0);   py = 1: ky = 0 (dy = -1), ky = 2 (dy = 0);   window row r = dy + 1   (same in x)
    const int c4 = 4 * c_in_pad;
    const size_t total = (size_t)c_out * c4 * 9;
    for (size_t i = 0; i < total; ++i) out[i] = 0.f;
    for (int co = 0; co < c_out; ++co)
        for (int ci = 0; ci < c_in; ++ci)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int py = (ky + 1) & 1, px = (kx + 1) & 1;
                    const int dy = (ky - 1 - py) / 2, dx = (kx - 1 - px) / 2;     // exact: numerators even
                    const int cc = ((ci >> 4) * 4 + py * 2 + px) * 16 + (ci & 15);
                    out[(((size_t)co * c4 + cc) * 3 + (dy + 1)) * 3 + (dx + 1)] = w[(((size_t)co * c_in + ci) * 3 + ky) * 3 + kx];
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

int launch_conv3x3_v2(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    // raw buffer descriptors address at most 4 GiB: tensor bytes (hi + lo plane) per source
    for (int i = 0; i < 2; ++i) {
        const ConvSrc& sp = a.src[i < a.nsrc ? i : 0];
        const size_t bytes = ((size_t)sp.plane + (size_t)a.n * sp.c * sp.h * sp.w) * sizeof(f16);
        if (bytes >= ((size_t)1 << 32)) { set_error("conv3x3: activation tensor of %zu bytes exceeds 32-bit buffer addressing; split the batch", bytes); return DISCO_ESHAPE; }
        a.src_bytes[i] = (uint32_t)bytes;
        if (i >= a.nsrc) a.src[i] = a.src[0];
    }
    {
        const size_t wb = conv3x3_packed_bytes(a.c_out, a.c_in);
        if (wb >= ((size_t)1 << 32)) { set_error("conv3x3: packed weights too large"); return DISCO_ESHAPE; }
        a.w_bytes = (uint32_t)wb;
    }
    if (a.stride != 1 && a.stride != 2) { set_error("conv3x3: stride %d", a.stride); return DISCO_ESHAPE; }
    if (a.softmax && (!a.out_f32 || a.c_out > 32)) { set_error("conv3x3: the fused softmax needs the fp32 NCHW output and c_out <= 32"); return DISCO_ESHAPE; }
    if (a.s2d) {
        // weights packed for the space-to-depth view: c_in counts the 4 phases; needs a stride-2, single-source, even-sized input
        if (a.stride != 2 || a.nsrc != 1 || a.src[0].up || (a.h_in & 1) || (a.w_in & 1) || a.c_in != 4 * a.src[0].c || a.tapmask) {
            set_error("conv3x3: space-to-depth weights need stride 2, one plain source and even input sizes (%dx%d)", a.h_in, a.w_in);
            return DISCO_ESHAPE;
        }
    }
    if (a.c_in % 16 || a.src[0].c % 16 || (a.nsrc > 1 && a.src[1].c % 16)) {
        set_error("conv3x3: input channels must be multiples of 16 (got %d)", a.c_in);
        return DISCO_ESHAPE;
    }
    if (a.d2s_c > 0 && (a.d2s_c % 16 || a.out_f32)) { set_error("conv3x3: depth-to-space needs a multiple of 16 channels (got %d) and an activation output", a.d2s_c); return DISCO_ESHAPE; }
    if (a.act == DISCO_ACT_LRELU && !(a.slope >= 0.f && a.slope <= 1.f)) { set_error("conv3x3: LeakyReLU slope %g outside [0, 1]", (double)a.slope); return DISCO_ESHAPE; }
    if (a.c_out > 32 && a.c_out % 64) { set_error("conv3x3: c_out %d (>32) must be a multiple of 64", a.c_out); return DISCO_ESHAPE; }
    for (int i = 0; i < a.nsrc; ++i)
        if ((size_t)a.src[i].h * a.src[i].w * 16 >= (1u << 30)) {
            set_error("conv3x3: image too large for 30-bit in-image offsets");
            return DISCO_ESHAPE;
        }
    return a.precision == DISCO_PREC_F16X1 ? dispatch2<false>(a, s) : dispatch2<true>(a, s);
}

}  // namespace disco
