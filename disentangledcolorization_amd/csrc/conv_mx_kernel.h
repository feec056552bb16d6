// conv_mx_kernel.h — the kernel and its launchers (conv_mx.hip: host side; conv_mx_ar{0,1,2}.hip: one instantiation set per arithmetic) — 3x3 implicit-GEMM conv for gfx950 whose products are an fp16 main term plus fp8 corrections:
//
//     w a  ~=  w_h a_h                                  v_mfma_f32_32x32x16_f16      (w_h = fp16(w), a_h = fp16(a))
//            + q8(w - w_h) q8(a) + q8(w) q8(a - a_h)    v_mfma_scale_f32_32x32x64_f8f6f4, fp8 e4m3, both terms in ONE K = 64
//                                                        instruction (K half 0: wl8 x a8, K half 1: w8 x al8)
//
// The two correction terms are ~2^-11 of the main term, so their 3 mantissa bits leave a relative error of ~2^-16 per
// product (measured end to end: profiles/r02_precision_sim.txt), and the fp8 instruction retires 4x the K of the fp16 one
// per pass: per 32 input channels and tap a 32x32 block costs 2 + 2 = 4 matrix-pipe units instead of the 6 of three fp16
// products (conv_mfma2.hip), and the fp8 passes draw less power, so the power-managed clock stays higher
// (profiles/r02_mfma_mix.txt: 54 vs 40 G units/s on random operands).
//
// Data formats (common.h, struct Act): ONE power-of-two scale 2^sexp per tensor (fixed at calibration), carried by every plane:
// as = a 2^sexp; fp16 hi plane [N][C/16][H][W][16] = a_h = fp16(as); q planes [N][C/32][2][H][W][32] fp8 e4m3 holding a8 = fp8(as)
// and al8 = fp8((as - a_h) 2^11); weights per output channel co: w8 = fp8(w 2^wexp[co]), wl8 = fp8((w - w_h) 2^(wexp[co]+11)).  The
// hardware applies 2^-(wexp[co] + 11) to every fp8 product through the instruction's E8M0 scale operand of the weight side (per lane
// = per output channel; the pixel side's is 1), so all products of a tile accumulate in the sources' domain 2^sexp; the epilogue
// moves to the output's with exact power-of-two factors folded into its parameters (ConvMxArgs::acc_mul ...).
//
// XQ = true selects a second arithmetic on the same skeleton, for the layers whose output decides discrete results downstream:
//
//     w a  ~=  w_h a_h + w_l a_h                        four K = 16 fp16 MFMAs per 32 channels (w_l = fp16(w - w_h): an "L" chunk that
//                                                        re-reads the hi plane against the residual weights)
//            + q8(w) q8(a - a_h)                         ONE K = 64 fp8 MFMA per 64 channels (its K halves are two 32-channel blocks)
//
// i.e. only the ACTIVATION residual goes through fp8.  The weight residual's rounding error is the same at every pixel and
// multiplies non-negative activations, so it survives spatial pooling (profiles/r02_precision_sim.txt: it is what moves the
// anchors under the arithmetic above); the activation residual's is zero-mean per pixel.  5 pipe units per 32 channels and tap
// (f16x3: 6).  Its sources carry the hi plane and al8-only q planes (Act::q_kind 1); chunk order per 64 channels: H L H L Q.
//
// Pipeline (same skeleton as conv_mfma2.hip): a "chunk" is 64 bytes per halo pixel and 64 bytes per (tap, output channel):
// either the fp16 hi values of 32 channels (H chunk: two 16-channel planes) or their two fp8 planes (Q chunk).  Chunks
// travel HBM/L2 -> LDS by LDS-DMA through raw buffer descriptors (out-of-range lanes read 0 = zero padding), double
// buffered, one s_barrier per chunk, the next chunk's DMA issued in ninths between the taps; 16-byte XOR swizzle on pixel
// bit 3 (bank-conflict-free ds_read_b128 fragments); persistent workgroups walking over the images of the batch;
// v_permlane32_swap epilogue with 16-byte stores.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>
#include "common.h"

#ifndef MX_PIX_AUX
#define MX_PIX_AUX 0     // cache-policy bits (aux) of the pixel / weight LDS-DMA: experiments only (2 = nt)
#endif
#ifndef MX_W_AUX
#define MX_W_AUX 0
#endif
#ifndef MX_QFMT
#define MX_QFMT 0       // operand format of the K = 64 correction MFMA: 0 = fp8 e4m3.  2 (fp6 e2m3) / 4 (fp4): SPEED EXPERIMENTS ONLY - the data stay fp8 bytes
#endif
#ifndef MX_ABL
#define MX_ABL 0        // diagnostic builds (tools/build_ablations.sh; bit 4 = no output stores, bit 5 = no epilogue, bit 6 = the latency loop on half the chunks): bit 0 = no LDS-DMA after the first chunk, bit 1 = fragments read once per chunk,
                        // bit 2 = no PIXEL pieces after the first chunk, bit 3 = no WEIGHT pieces after the first chunk
#endif

#ifndef MX_DEFER
#define MX_DEFER 0      // 1: deferred epilogue on the main tile (see DEFER in the kernel): A/B builds, tools/build_conv_variants.sh
#endif
#ifndef MX_DEFER_SLOT0
#define MX_DEFER_SLOT0 1
#endif
#ifndef MX_DEFER_STEP
#define MX_DEFER_STEP 2
#endif

#ifndef MX_EPI_INLINE
#define MX_EPI_INLINE 1 // 1: an output block's stores are issued right behind its epilogue math (round 4); 0: all math, then all stores (round 3)
#endif
#ifndef MX_TIMELINE
#define MX_TIMELINE 0   // diagnostic builds (tools/conv_timeline.py): workgroup (0, 0) stamps s_memtime at every phase boundary of waves 0 and NWAVE-1
#endif

namespace disco {

namespace {

#if MX_TIMELINE
constexpr int MX_TL_EVENTS = 8192;
__device__ unsigned long long g_mx_tl[2][MX_TL_EVENTS];      // [first / last wave][event]: (s_memtime << 4) | tag; entry 0 = number of events
#define MX_TL(tag)                                                                                                      \
    do {                                                                                                                \
        if (tl_on) {                                                                                                    \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                 \
            if (lane == 0 && tl_n < MX_TL_EVENTS) g_mx_tl[tl_w][tl_n] = (t_ << 4) | (unsigned long long)(tag);          \
            ++tl_n;                                                                                                     \
        }                                                                                                               \
    } while (0)
// read back / clear the phase stamps of workgroup (0, 0) of this translation unit's instantiations
#define MX_TIMELINE_EXPORT(NAME)                                                                                                     \
    extern "C" int NAME(unsigned long long* h_dst /* [2][8192] */, int clear) {                                                      \
        if (clear) {                                                                                                                 \
            static unsigned long long zeros[2][disco::MX_TL_EVENTS];                                                                 \
            return hipMemcpyToSymbol(HIP_SYMBOL(disco::g_mx_tl), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;                        \
        }                                                                                                                            \
        return hipMemcpyFromSymbol(h_dst, HIP_SYMBOL(disco::g_mx_tl), sizeof(unsigned long long) * 2 * disco::MX_TL_EVENTS) == hipSuccess ? 0 : -1; \
    }
#else
#define MX_TL(tag) do { } while (0)
#endif

constexpr int WBLK = 1024;
constexpr int W_NB = 9 * 2 * WBLK;          // bytes of one chunk of one 32-cout block: 9 taps x 2 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// buffer_store_dwordx4 with two wait states glued behind it.  gfx950 hazard (tools/store_hazard_repro.hip, profiles/r02_store_hazard.txt):
// when the instruction right behind a dwordx4 store is a VALU write to one of its data registers, lanes 12-15 of every row of 16
// store the NEW value (measured: buffer stores need 1 wait state, global stores 2).  The compiler's hazard recogniser covers global /
// flat stores, and buffer stores only when they have NO register soffset ("this hazard only exists if the instruction is not using
// a register in the soffset field") - on this part it exists with one, and these stores all have one.  An asm block is the only way
// to keep the scheduler from moving a VALU instruction into the gap.
__device__ __forceinline__ void buffer_store_b128(i32x4 d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    // ... and five wait states in FRONT of it: a VMEM instruction that reads an SGPR (descriptor, soffset) written by a VALU instruction
    // needs 5 wait states, the compiler's hazard recogniser does not look into an asm block, and in the instantiations that spill SGPRs
    // the descriptor / offset are reloaded with v_readlane_b32 right in front of the store.  Found in round 3 (tools/conv_determinism_probe.py):
    // the masked depth-to-space instantiation of the f16x2+fp8 arithmetic (116 spilled SGPRs) wrote its hi plane through stale offsets,
    // differently from run to run.
    // (readfirstlane: the offset is wave-uniform by construction, but in a few instantiations the compiler keeps it in a VGPR, which an
    // "s" constraint turns into a compile error; where it already lives in an SGPR this folds away)
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(d), "v"(voff), "s"(r), "s"(soff) : "memory");
}

// four floats -> four fp8 e4m3 bytes of (x / scale), scale a power of two: ONE v_cvt_scalef32_pk_fp8_f32 per pair does the scaling
// and the round-to-nearest-even (tools/cvt_scale_probe.hip, profiles/r03_cvt_scale_probe.txt: the instruction DIVIDES by its scale
// operand and is bit-identical to v_mul + v_cvt_pk_fp8_f32 on every code boundary, tie and 100 000 random values).  It does NOT
// saturate: |x / scale| > 464 yields the NaN code 0x7f - callers check the range themselves.
typedef short s16x2 __attribute__((ext_vector_type(2)));
// one v_max_f32.  fmaxf() (and v_med3 written as a builtin: it is folded back) makes the compiler quiet possible signalling NaNs with
// a v_max(x, x) in front of every maximum whose operand comes through a phi - twice the instructions in the conv epilogue
__device__ __forceinline__ float vmax_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned fp8x4_scaled(f32x2 a, f32x2 b, float scale) {
    s16x2 r = __builtin_bit_cast(s16x2, a[0]);        // any defined register will do (the untouched half is overwritten by the second
                                                      // conversion); a source that dies here saves the v_mov of a fresh zero
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, a[0], a[1], scale, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, b[0], b[1], scale, true);
    return __builtin_bit_cast(unsigned, r);
}

// buffer_store_dwordx3 with the same wait states as buffer_store_b128() (a 96-bit store has the same data hazard)
typedef int i32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void buffer_store_b96(i32x3 d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_nop 4\n\tbuffer_store_dwordx3 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(d), "v"(voff), "s"(r), "s"(soff) : "memory");
}
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

template <int TW, int TH, int STRIDE>
struct GeoMx {
    static constexpr int MB = TW * TH / 32;
    static constexpr int TWI = (TW - 1) * STRIDE + 3;
    static constexpr int THI = (TH - 1) * STRIDE + 3;
    static constexpr int HALF = (TWI + 1) / 2;
    static constexpr int PITCH = STRIDE == 1 ? TWI : 2 * HALF;
    static constexpr int NPIX = THI * PITCH;
    static constexpr int ROWS_PER_MB = 32 / TW;
    static_assert(32 % TW == 0, "an M block covers whole tile rows");
};

// NSRC2: the layer concatenates two sources on read (the descriptor and offsets of the second source exist only then).
// AR: 0 = f16 + fp8x2, 1 = f16x2 + fp8 (x2q), 2 = f16x3 (below), 3 = f16 + fp6x2: AR 0 with the two correction operands in fp6 e2m3
// (Act::q_kind 2 sources, conv_mx_pack_host variant 2): the K = 64 MFMA takes half the passes (tools/fp6_probe.hip pins the operand
// layout and the conversion; profiles/r03_mfma_mix.txt the rate)
// GENC1 (round 4, f16x3 only): the layer's 16-channel chunks are not read from a tensor - they are COMPUTED in LDS from the gray image, as
// the outputs of the Cin = 1 conv that precedes it in the network (repnet.conv1_2.0 -> conv1_2.2, network.py:152-153): the same fmaf chain,
// bias, LeakyReLU, scale and hi/lo split as conv_c1_kernel + store_act8 (bit-identical), so that layer's 1.07 GB tensor is never written or read
// NB: LDS buffers.  2 = the throughput loop (a chunk's DMA issued in ninths between its predecessor's taps).  3 = the LATENCY loop (round 5),
// for launches that cannot fill the GPU (one image, the deep layers of small batches): a chunk's whole DMA goes out in one burst TWO chunks
// ahead, right behind the barrier that frees its buffer; fragment reads run one tap ahead of the MFMAs; the chunk barrier sits inside the
// last tap of the preceding chunk.  Same chunks, same taps, same MFMAs in the same order per accumulator: results are bit-identical to NB = 2.
// NJ: images per WEIGHT chunk (throughput loop).  1: a workgroup walks image after image through all chunks, fetching every weight chunk
// once per image.  4 (round 5, the stride-2 tile): it walks FOUR of its images chunk by chunk - the stages (chunk c, image j), j = 0..3, run
// back to back on four accumulator sets, the pixel buffers alternate per stage, the weight buffers per chunk - so a weight chunk comes in
// once per four images.  The stride-2 tile moves 38 KB of pixels + 36 KB of weights per 27 MFMAs of a wave and is bound by that L2->LDS
// volume (r01_conv_s2d_timeline.txt, r03_lds_dma_rate.txt); with NJ = 4 it moves 38 + 9.  Every accumulator sees the same chunks and taps
// in the same order: bit-identical to NJ = 1.
template <int TW, int TH, int NT, int STRIDE, int WM, int WN, bool MASKED, bool NSRC2, int AR, bool GENC1 = false, int NB = 2, int NJ = 1>
__global__ __launch_bounds__((WM * WN + (NB == 3 ? 4 : 0)) * 64, 2) void conv3x3_mx_kernel(const ConvMxArgs a) {
    constexpr bool XQ = AR == 1, X3 = AR == 2, Q6 = AR == 3;
    constexpr int QFMT = Q6 ? 2 : MX_QFMT;                // operand format code of the K = 64 MFMA: 0 = fp8 e4m3, 2 = fp6 e2m3
    static_assert(!(XQ && NSRC2), "the f16x2+fp8 arithmetic takes one source");
#if defined(__HIP_DEVICE_COMPILE__)
    using G = GeoMx<TW, TH, STRIDE>;
    constexpr int NWAVE = WM * WN;
    // The latency loop is wave-specialised: NWAVE CONSUMER waves (fragment reads, MFMAs, epilogue) and NPROD = 4 PRODUCER waves, one per SIMD,
    // which do nothing but issue the LDS-DMA and wait for it.  A consumer alone on its SIMD pays for every instruction between two MFMAs of
    // its accumulator chain, and an LDS-DMA piece costs 60-180 issue cycles there (measured: three pieces per tap group were ~270 of a
    // group's ~560 cycles); in a wave of its own the issue runs next to the consumer's MFMAs.
    constexpr int NPROD = NB == 3 ? 4 : 0;
    // DEFER (MX_DEFER builds): a tile's epilogue runs inside the NEXT tile's first chunk, block by block between its taps, from a second
    // accumulator set - the main 8-wave stride-1 tile of the f16x3 / f16+fp6x2 arithmetics only (64 + 64 accumulator registers)
    constexpr bool DEFER = MX_DEFER && NB == 2 && NJ == 1 && !NSRC2 && !GENC1 && !MASKED && STRIDE == 1 && TW == 32 && TH == 16 && NT == 2 && (AR == 2 || AR == 3);
    constexpr int DEFER_SLOT0 = MX_DEFER_SLOT0, DEFER_STEP = MX_DEFER_STEP;
    constexpr int NDW = NPROD ? NPROD : NWAVE;                // waves that issue DMA
    constexpr int NTHR = (NWAVE + NPROD) * 64;
    constexpr int MT = G::MB / WM;
    constexpr int NTW = NT / WN;
    static_assert(G::MB % WM == 0 && NT % WN == 0, "tile split");
    constexpr int PLANE_B = G::NPIX * 32;
    constexpr int A_UNITS = 2 * G::NPIX * 2;
    constexpr int A_PIECES = (A_UNITS + 63) / 64;
    constexpr int A_BYTES = A_PIECES * 1024;
    constexpr int W_PIECES = NT * 18;
    constexpr int BUF_BYTES = A_BYTES + W_PIECES * 1024;
    constexpr int APW = (A_PIECES + NDW - 1) / NDW;
    constexpr int WPW = (W_PIECES + NDW - 1) / NDW;
    constexpr int APT = (APW + 8) / 9, WPT = (WPW + 8) / 9;
    constexpr int PAR_OFF = NB * BUF_BYTES;
    static_assert(NB == 2 || NB == 3, "two or three LDS buffers");
    static_assert(NJ == 1 || (NB == 2 && !GENC1 && AR != 1 && !NSRC2 && NJ * MT * NTW <= 4), "several images per weight chunk: plain one-source chunk sequences, at most 64 accumulator registers");
    constexpr int DUMP_OFF = PAR_OFF + 3 * 32 * NT * 4;       // NB = 3: 1 KiB that absorbs the out-of-range DMA pieces (every wave issues the same number)
    constexpr int DMA_PER_CHUNK = APW + WPW;                  // NB = 3: LDS-DMA instructions per wave and chunk, exactly (the s_waitcnt immediate)
    static_assert(DMA_PER_CHUNK <= 60, "vmcnt is a 6-bit counter");
    static_assert(NB == 2 || (MT == 1 && NTW == 1 && !GENC1 && !(NSRC2 && AR == 0) && AR != 1), "the latency loop serves one 32 x 32 block per wave, plain chunk sequences");

    // GENC1: behind the parameters: two gray tiles (the tile's input footprint of the Cin = 1 conv: one pixel more on every side than the
    // layer's own halo tile) and that conv's weights + biases, 10 floats per channel
    constexpr int GTW = G::TWI + 2, GTH = G::THI + 2, GT_FLOATS = GTW * GTH;
    constexpr int GT_PIECES = (GT_FLOATS + 63) / 64;                     // 256-byte LDS-DMA pieces (4 bytes per lane)
    constexpr int GT_BYTES = GT_PIECES * 256;
    constexpr int GT_OFF = PAR_OFF + 3 * 32 * NT * 4, C1W_OFF = GT_OFF + 2 * GT_BYTES;
    constexpr int GPW = (GT_PIECES + NWAVE - 1) / NWAVE;
    static_assert(!GENC1 || (AR == 2 && STRIDE == 1 && !NSRC2 && !MASKED), "the fused Cin = 1 producer exists for plain f16x3 layers");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    {
        // Touch every 64-byte line of the kernel-argument segment NOW, in one burst of scalar loads: the compiler reads the arguments where it
        // needs them, in five or six dependent batches, and each batch is a cold miss of its own (the segment is written per launch: nothing of
        // it is cached) - ~3 000 cycles before the first DMA piece could leave (s_memtime stamps of a producer wave, profiles/r05_latency_loop.txt).
        // After this, they hit the scalar cache.
        typedef const __attribute__((address_space(4))) unsigned KWord;
        KWord* ka = (KWord*)__builtin_amdgcn_kernarg_segment_ptr();
        constexpr int LINES = (int)((sizeof(ConvMxArgs) + 63) / 64);
        unsigned t[LINES];
#pragma unroll
        for (int i = 0; i < LINES; ++i) t[i] = ka[16 * i];
#pragma unroll
        for (int i = 0; i < LINES; ++i) asm volatile("" :: "s"(t[i]));
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = NPROD > 0 && wave >= NWAVE;             // (wave-uniform)
    const int dwv = NPROD ? wave - NWAVE : wave;                  // this wave's index among the DMA-issuing waves (consumers of the latency loop: unused)
    const int wm = producer ? 0 : wave % WM, wn = producer ? 0 : wave / WM;
#if MX_TIMELINE
    // (the latency loop: the second trace is the first PRODUCER wave's)
    const bool tl_on = blockIdx.x == 0 && blockIdx.y == 0 && (wave == 0 || wave == (NB == 3 ? NWAVE : NWAVE - 1));
    const int tl_w = wave == 0 ? 0 : 1;
    int tl_n = 1;
    MX_TL(12);                           // kernel entry
#endif
    const int tiles_x = (a.w_out + TW - 1) / TW, tiles_y = (a.h_out + TH - 1) / TH;
#if MX_ABL & 64
    // ablation (timing only): the latency loop walks HALF of a long accumulation chain - what a launch would take if two workgroups per tile each
    // summed half of the chunks ("Next" (i)); results wrong by design
    const int nchunks_full = XQ ? (a.c_in >> 6) * 5 : a.c_in >> 4;
    const int nchunks = (NB == 3 && nchunks_full >= 16) ? nchunks_full >> 1 : nchunks_full;
#else
    const int nchunks = XQ ? (a.c_in >> 6) * 5 : a.c_in >> 4;    // two chunks (H, Q) per 32 input channels; XQ: H L H L Q per 64; X3: one per 16
#endif

    int bid = blockIdx.x;
    int tx, ty, by;
    if constexpr (NB == 3) {
        // output-channel block fastest: workgroup ids go round the 8 XCDs, so an XCD (its own 4 MB L2) serves few channel blocks for ALL pixel
        // tiles - each weight tile comes out of HBM / Infinity Cache once per XCD and is an L2 hit for the other pixel tiles (512 -> 512 @32^2 on
        // 32 x 4 tiles: 2 blocks = 1.2 MB of weights per XCD instead of all 9.4 MB streaming through every L2)
        const int nby = (a.c_out + 32 * NT - 1) / (32 * NT);
        by = bid % nby; bid /= nby;
        tx = bid % tiles_x; ty = bid / tiles_x;
    } else {
        tx = bid % tiles_x; bid /= tiles_x;
        ty = bid % tiles_y;
        by = bid / tiles_y;
    }
    const int ox0 = tx * TW, oy0 = ty * TH;
    const int img_step = gridDim.y;
    int n = blockIdx.y;
    if (n >= a.n) return;

    unsigned tmask = 0x1ffu;
    if (MASKED && a.tapmask) {
        tmask = 0;
#pragma unroll
        for (int j = 0; j < NT; ++j) tmask |= a.tapmask[by * NT + j];
        tmask = __builtin_amdgcn_readfirstlane(tmask);
    }

    float* s_par = reinterpret_cast<float*>(smem + PAR_OFF);
    auto stage_params = [&]() {
        for (int i = tid; i < 3 * 32 * NT; i += NWAVE * 64) {          // (the latency loop's consumers: tid < NWAVE * 64)
            const int which = i / (32 * NT), c = i - which * (32 * NT);
            const int co = by * NT * 32 + c;
            const int cpar = a.d2s_c > 0 ? co % a.d2s_c : co;
            const float* src = which == 0 ? a.bias : (which == 1 ? a.bn_scale : a.bn_shift);
            // exact power-of-two factors carry the parameters into the domains the epilogue works in (ConvMxArgs::acc_mul ...)
            const float pm = which == 0 ? a.bias_mul : (which == 1 ? a.bns_mul : a.bnh_mul);
            s_par[i] = ((src && co < a.c_out) ? src[cpar] : (which == 1 ? 1.f : 0.f)) * pm;
        }
    };
    if constexpr (NB == 2) stage_params();       // (the latency loop stages them behind its first DMA bursts: their load latency runs under the DMA's)

    if (GENC1) {
        // the producing conv's weights and biases: channel c at floats [10 c, 10 c + 9), bias at 10 c + 9
        float* s_c1 = reinterpret_cast<float*>(smem + C1W_OFF);
        for (int i = tid; i < a.c_in * 10; i += NWAVE * 64) {
            const int c = i / 10, k = i - c * 10;
            s_c1[i] = k < 9 ? a.c1_w[c * 9 + k] : (a.c1_bias ? a.c1_bias[c] : 0.f);
        }
    }
    // E8M0 scale operands of the fp8 products: weight side per lane (= per output channel row), pixel side uniform per source
    int wsc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) wsc[j] = (X3 || Q6 || producer) ? 0 : 127 - MX_LO_SHIFT - a.wexp[(by * NT + wn * NTW + j) * 32 + (lane & 31)];       // (wexp: fp8 scaling per output channel, conv_mx_pack_host; fp6 weight slots carry a block scale each: below)
    // pixel-side E8M0 scale: 1 (the tensor's scale stays in the accumulators)
    constexpr int asc0 = 127, asc1 = asc0;                // fp8 activation planes: no block scale.  fp6 slots: dword 6 of the fragment (below)

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[0].p, 0, a.src_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[NSRC2 ? 1 : 0].p, 0, a.src_bytes[NSRC2 ? 1 : 0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    constexpr int NS = NSRC2 ? 2 : 1;
    // GENC1: the gray image through its own descriptor; per lane the byte offset of its element(s) of the (TH + 4) x (TW + 4) tile, or OOB
    // (an out-of-range lane lands as 0 in LDS = the producing conv's zero padding)
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)(GENC1 ? a.c1_gray : nullptr), 0, GENC1 ? (unsigned)a.n * (unsigned)(a.h_in * a.w_in) * 4u : 0u, 0x00020000);
    unsigned gvoff[GPW];
    if (GENC1) {
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            const int e = (i * NWAVE + wave) * 64 + lane;
            const int gy = oy0 - 2 + e / GTW, gx = ox0 - 2 + e % GTW;
            gvoff[i] = (e < GT_FLOATS && gy >= 0 && gy < a.h_in && gx >= 0 && gx < a.w_in) ? (unsigned)(gy * a.w_in + gx) * 4u : OOB;
        }
    }
    // LDS-DMA of image `img`'s gray tile into gray buffer gb
    auto issue_gray = [&](int img, int gb) {
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            const int piece = i * NWAVE + wave;
            if (piece < GT_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (lds_void*)(smem + GT_OFF + gb * GT_BYTES + piece * 256), 4, gvoff[i], (unsigned)img * (unsigned)(a.h_in * a.w_in) * 4u, 0, 0);
        }
    };
    // One wave-iteration of the fused producer: 64 halo pixels x 8 channels (the wave-uniform half j of chunk ck's 16 channels) computed from
    // gray buffer gb and written as one hi and one lo 16-byte unit per pixel into pixel buffer `abuf` - conv_c1_kernel's arithmetic, to the bit:
    // s = 0; s = fmaf(in[k], w[k], s) for k = 0..8; s += bias; LeakyReLU; s * 1 + 0 (its BN affine without a BN); then store_act8's split
    auto gen_c1 = [&](int it, int ck, int gb, int abuf) {
        const int pg = it >> 1, j = it & 1;
        const int p = pg * 64 + lane;
        if (pg * 64 >= G::NPIX) return;                    // (wave-uniform)
        const int py = p / G::PITCH, q = p - py * G::PITCH;
        const int gy = oy0 - 1 + py, gx = ox0 - 1 + q;
        const bool inside = p < G::NPIX && gy >= 0 && gy < a.h_in && gx >= 0 && gx < a.w_in;
        const float* gt = reinterpret_cast<const float*>(smem + GT_OFF + gb * GT_BYTES);
        float in[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) in[ky * 3 + kx] = p < G::NPIX ? gt[(py + ky) * GTW + q + kx] : 0.f;
        const float* wc = reinterpret_cast<const float*>(smem + C1W_OFF) + (ck * 16 + j * 8) * 10;       // (LDS broadcasts; scalar loads of the
        // same weights from global memory were slower: profiles/r04_fused_first_layer_ab.txt)
        const float sc = __builtin_ldexpf(1.f, a.c1_sexp);
        float one = 1.f, zero = 0.f;
        asm volatile("" : "+v"(one), "+v"(zero));          // the affine (1, 0) as run-time values, as the stand-alone kernel has them
        f16x8 h, l;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float sacc = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) sacc = fmaf(in[k], wc[c * 10 + k], sacc);
            sacc += wc[c * 10 + 9];
            if (a.c1_act == DISCO_ACT_RELU) sacc = fmaxf(sacc, 0.f);
            else if (a.c1_act == DISCO_ACT_LRELU) sacc = sacc >= 0.f ? sacc : sacc * a.c1_slope;
            float v = sacc * one + zero;
            if (!inside) v = 0.f;                          // this layer's own zero padding
            const float vs = v * sc;
            h[c] = (f16)vs;
            l[c] = (f16)(vs - (float)h[c]);
        }
        if (p < G::NPIX) {
            char* d = smem + abuf * BUF_BYTES + p * 32 + ((j ^ ((q >> 3) & 1)) << 4);
            *reinterpret_cast<f16x8*>(d) = h;
            *reinterpret_cast<f16x8*>(d + PLANE_B) = l;
        }
    };
    constexpr int GEN_ITERS = ((G::NPIX + 63) / 64) * 2;                 // wave-iterations per chunk
    constexpr int GEN_PER_WAVE = (GEN_ITERS + NWAVE - 1) / NWAVE;
    unsigned voff[NS][APW];           // per source: byte offset (plane included) of this lane's 16 bytes inside a chunk, or OOB
    auto compute_voff = [&]() {
        if (GENC1) return;
        const int ix0 = ox0 * STRIDE - 1, iy0 = oy0 * STRIDE - 1;
        // (the geometry in scalar registers up front, the range test without short-circuit branches: the compiler otherwise re-reads the
        // argument segment inside a branch per entry and test - a dozen dependent scalar-load round trips on the way to the first pixel piece)
        int h_in = a.h_in, w_in = a.w_in;
        asm volatile("" : "+s"(h_in), "+s"(w_in));
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            int sp_h = a.src[si].h, sp_w = a.src[si].w, sp_up = a.src[si].up;
            unsigned sp_qoff = a.src[si].q_off;
            asm volatile("" : "+s"(sp_h), "+s"(sp_w), "+s"(sp_up), "+s"(sp_qoff));
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const int u = (i * NDW + dwv) * 64 + lane;
                const int plane = u / (G::NPIX * 2);
                const int rem = u - plane * (G::NPIX * 2);
                const int p = rem >> 1, j = rem & 1;
                const int py = p / G::PITCH, q = p - py * G::PITCH;
                const int kh = j ^ ((q >> 3) & 1);          // 16-byte XOR swizzle on bit 3 of the COLUMN position (see the fragment reads)
                int px;
                bool slot_ok = u < A_UNITS;
                if (STRIDE == 1) px = q;
                else { const int par = q / G::HALF; px = (q - par * G::HALF) * 2 + par; slot_ok = slot_ok & (px < G::TWI); }
                const int gy = iy0 + py, gx = ix0 + px;
                const bool in = slot_ok & (gy >= 0) & (gy < h_in) & (gx >= 0) & (gx < w_in);
                // both chunk kinds keep their second plane h*w*32 bytes after the first (the next 16-channel block of the hi
                // plane / the al8 plane of the q block), so one offset serves both
                // (X3: the second plane is the tensor's lo plane, q_off bytes after the hi plane)
                const unsigned o3 = (unsigned)plane * sp_qoff + (unsigned)((gy >> sp_up) * sp_w + (gx >> sp_up)) * 32u + kh * 16u;
                const unsigned o0 = (unsigned)((plane * sp_h + (gy >> sp_up)) * sp_w + (gx >> sp_up)) * 32u + kh * 16u;
                voff[si][i] = in ? (X3 ? o3 : o0) : OOB;
            }
        }
    };
    if constexpr (NB == 2) compute_voff();       // (the latency loop's producers compute them behind their first weight pieces)
    const int c_src0 = a.src[0].c;
    const unsigned img_b0 = (unsigned)(a.src[0].c * a.src[0].h * a.src[0].w) * 2u, blk_b0 = (unsigned)(a.src[0].h * a.src[0].w) * 32u;
    const unsigned img_b1 = (unsigned)(a.src[NS - 1].c * a.src[NS - 1].h * a.src[NS - 1].w) * 2u, blk_b1 = (unsigned)(a.src[NS - 1].h * a.src[NS - 1].w) * 32u;
    const unsigned qo0 = a.src[0].q_off, qo1 = a.src[NS - 1].q_off;
    const unsigned w_tile_b = (unsigned)(by * NT) * (unsigned)nchunks * W_NB, w_nt_b = (unsigned)nchunks * W_NB;

    // one ninth (part 0..8) of the DMA of chunk `ck` of image `img` into LDS buffer `buf`; part < 0: all of it
    // (NJ > 1: the weight pieces go to weight buffer `wb` and only if `with_w`; NJ = 1: wb = buf, always)
    auto issue = [&](int img, int ck, int buf, int part, int wb, bool with_w) {
        int c0 = (ck >> 1) << 5;                       // first channel of the chunk's 32-channel group
        bool isq = ck & 1;
        if (X3) { c0 = ck << 4; isq = false; }         // one chunk per 16 channels: planes = hi / lo
        if (XQ) {                                      // chunk 5 g64 + i: i = 0, 1: H, L of channels 64 g64 ..; 2, 3: of 64 g64 + 32 ..; 4: Q of all 64
            const int g64 = ck / 5, i = ck - 5 * g64;
            isq = i == 4;
            c0 = (g64 << 6) + (isq ? 0 : (i >> 1) << 5);
        }
        const bool s1 = NSRC2 && c0 >= c_src0;
        // hi plane: image stride C*H*W*2 bytes, 16-channel block stride H*W*32; q planes: the same image stride (2 x 1 byte
        // per element) and 2 x H*W*32 per 32-channel block - the SAME offsets, shifted by q_off.  XQ: al8-only planes, one byte
        // per element: half the image stride, H*W*32 per 32-channel block.
        unsigned soff = s1 ? (unsigned)img * img_b1 + (unsigned)((c0 - c_src0) >> 4) * blk_b1 + (isq ? qo1 : 0u)
                           : (unsigned)img * img_b0 + (unsigned)(c0 >> 4) * blk_b0 + (isq ? qo0 : 0u);
        if (XQ && isq) soff = (unsigned)img * (img_b0 >> 1) + (unsigned)(c0 >> 5) * blk_b0 + qo0;
        char* dA = smem + buf * BUF_BYTES;
        char* dW = smem + (NJ == 1 ? buf : wb) * BUF_BYTES + A_BYTES;
        const bool tail = NSRC2 && AR == 0 && (nchunks & 1) && ck == nchunks - 1;       // 16-channel H-only chunk: plane 0 alone
        if constexpr (NB == 3) {
            // the latency loop: every producer wave issues exactly DMA_PER_CHUNK instructions per chunk (its s_waitcnt counts them); a piece
            // beyond the tile reads out of range (zeros, no memory traffic) into the dump area; masked taps' weights are fetched like the others
            // (part -1: the whole chunk; -2: its weight pieces only, -3: its pixel pieces only - the first chunk of the stream goes out in that order,
            // with the per-lane pixel offsets computed in between)
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                if (part == -2) break;
                const int piece = i * NDW + dwv;
                const bool ok = (i + 1) * NDW <= A_PIECES || piece < A_PIECES;
                char* dst = ok ? dA + piece * 1024 : smem + DUMP_OFF;
                unsigned v0 = voff[0][i], v1 = voff[NS - 1][i];
                if (NSRC2) asm("" : "+v"(v0), "+v"(v1));
                const unsigned vsel = s1 ? v1 : v0;
                if (s1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_void*)dst, 16, vsel, soff, 0, MX_PIX_AUX);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_void*)dst, 16, vsel, soff, 0, MX_PIX_AUX);
            }
#pragma unroll
            for (int i = 0; i < WPW; ++i) {
                if (part == -3) break;
                const int piece = i * NDW + dwv;
                const bool ok = (i + 1) * NDW <= W_PIECES || piece < W_PIECES;
                const int nt = piece / 18, q = piece - nt * 18;
                char* dst = ok ? dW + piece * 1024 : smem + DUMP_OFF;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)dst, 16, ok ? (unsigned)lane * 16u : OOB,
                                                         ok ? w_tile_b + (unsigned)nt * w_nt_b + (unsigned)ck * W_NB + q * 1024 : 0u, 0, MX_W_AUX);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            if (GENC1) break;                              // the pixel tile is computed (gen_c1), not loaded
            if (part >= 0 && i / APT != part) continue;
#if MX_ABL & 4
            if (part >= 0) continue;
#endif
            const int piece = i * NWAVE + wave;
            if (NSRC2 && AR == 0 && tail && piece * 64 >= G::NPIX * 2) continue;       // a piece that lies entirely in plane 1
            if ((i + 1) * NWAVE <= A_PIECES || piece < A_PIECES) {      // only the last round of pieces needs the run-time test
                // (the offset is selected by VALUE: with the two calls behind an if / else the optimiser merged them into one call that
                // loads its offset through a phi of POINTERS into voff - which put voff into scratch memory, read back inside the tap loop)
                unsigned v0 = voff[0][i], v1 = voff[NS - 1][i];
                if (NSRC2) asm("" : "+v"(v0), "+v"(v1));          // opaque values: no select-of-loads folding
                const unsigned vsel = s1 ? v1 : v0;
                if (s1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_void*)(dA + piece * 1024), 16, vsel, soff, 0, MX_PIX_AUX);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_void*)(dA + piece * 1024), 16, vsel, soff, 0, MX_PIX_AUX);
            }
        }
        if (NJ > 1 && !with_w) return;
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            if (part >= 0 && i / WPT != part) continue;
#if MX_ABL & 8
            if (part >= 0) continue;
#endif
            const int piece = i * NWAVE + wave;
            const int nt = piece / 18, q = piece - nt * 18;
            if (NSRC2 && AR == 0 && tail && (q & 1)) continue;                          // the second plane's weights
            if (((i + 1) * NWAVE <= W_PIECES || piece < W_PIECES) && (!MASKED || ((tmask >> (q >> 1)) & 1u)))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)(dW + piece * 1024), 16, lane * 16,
                                                         w_tile_b + (unsigned)nt * w_nt_b + (unsigned)ck * W_NB + q * 1024, 0, MX_W_AUX);
        }
    };

    // ---- per-lane fragment addressing ----------------------------------------------------------------------------------------
    // LDS pixel (row, column q) of a plane lives at (row PITCH + q) 32 + (logical half ^ bit 3 of q) 16: the swizzle depends on
    // the column only, so a tap (ky, kx) adds a CONSTANT row offset (an instruction immediate) to one of three per-lane column
    // addresses - no address arithmetic inside the tap loop.  (A 16-lane group of a ds_read_b128 covers 16 consecutive
    // columns of one row on the 32-wide tiles: bank-conflict free.)
    const int r = lane & 31, kh = lane >> 5;
    const int lox = r % TW, loy = r / TW;
    const int w_off = lane * 16 + wn * NTW * W_NB;
    int colH[3], colQ[3];             // byte address (within an LDS buffer) of this lane's 16 bytes for kx = 0, 1, 2; row 0 of its M block 0
    {
        const int rowb = ((wm * MT * G::ROWS_PER_MB + loy) * STRIDE) * G::PITCH * 32;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int q = lox + (STRIDE == 1 ? kx : (kx & 1) * G::HALF + (kx >> 1));
            const int sw = (q >> 3) & 1;
            colH[kx] = rowb + q * 32 + ((sw ^ kh) << 4);                   // H chunk: logical half kh of plane 0 (plane 1: + PLANE_B)
            colQ[kx] = rowb + q * 32 + (sw << 4) + kh * PLANE_B;          // Q chunk: logical half 0 of plane kh (half 1: ^ 16)
        }
    }

    int gbuf = 0;                        // GENC1: gray buffer of the image being convolved (the other one receives the next image's tile)
    if (GENC1) {
        // first tile: its gray tile, then its first chunk computed up front (every later chunk is computed under its predecessor's taps)
        issue_gray(n, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the gray tile has landed, the weight staging above has been written
        __builtin_amdgcn_s_barrier();                      // ... and both are visible to every wave
#pragma unroll
        for (int k = 0; k < GEN_PER_WAVE; ++k) gen_c1(k * NWAVE + wave, 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    MX_TL(13);                           // parameters staged, offsets computed: the first chunk's DMA goes out
    int buf = 0;
    int wbuf = 0;                        // NJ > 1: the weight buffer of the chunk being computed (NJ = 1: the pixel buffer's index serves both)
    bool dma_waited = false;
    int l_cbuf = 0;                      // NB = 3: the LDS buffer the consumers compute from
    if constexpr (NB == 3) {
        // ---- the latency loop's PRODUCER waves -------------------------------------------------------------------------------------
        // The stream of chunks this workgroup walks - (image, chunk) pairs across image boundaries - is issued two ahead of the one being
        // computed, each chunk in one burst into the buffer that barrier B(s) has just freed:
        //     for every chunk s of the stream:  wait until chunk s has landed (chunk s + 1, issued behind it, may stay in flight: LDS-DMA
        //     pieces land in issue order and every wave issues exactly DMA_PER_CHUNK of them per chunk); s_barrier B(s) - the consumers
        //     arrive there with their reads of chunk s - 1 returned; issue chunk s + 2 into chunk s - 1's buffer.
        // One barrier per chunk, shared with the consumers (below); while the consumers run a tile's epilogue the producers sit in the next
        // tile's first barrier with its first two chunks in flight.
        if (producer) {
            int l_ibuf = 0, l_img = n, l_ck = 0;
            auto issue_next = [&]() -> bool {
                if (l_img >= a.n) return false;
                issue(l_img, l_ck, l_ibuf, -1, l_ibuf, true);
                l_ibuf = l_ibuf == NB - 1 ? 0 : l_ibuf + 1;
                if (++l_ck == nchunks) { l_ck = 0; l_img += img_step; }
                return true;
            };
            MX_TL(13);
            issue(n, 0, 0, -2, 0, true);                                   // the first chunk's weights: nothing to compute for them
            MX_TL(14);
            compute_voff();
            MX_TL(15);
            issue(n, 0, 0, -3, 0, true);
            l_ibuf = 1;
            if (++l_ck == nchunks) { l_ck = 0; l_img += img_step; }
            MX_TL(14);
            bool ahead = issue_next();
            MX_TL(15);
            for (int img = n; img < a.n; img += img_step)
                for (int ck = 0; ck < nchunks; ++ck) {
                    if (ahead) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_PER_CHUNK) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    MX_TL(3);
                    __builtin_amdgcn_s_barrier();
                    MX_TL(4);
                    ahead = issue_next();
                    MX_TL(2);
                }
#if MX_TIMELINE
            if (tl_on && lane == 0) g_mx_tl[tl_w][0] = (unsigned long long)tl_n;
#endif
            return;
        }
        stage_params();                                           // (the consumers: the producers have nothing in flight but DMA pieces, which is what their s_waitcnt counts)
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the staged parameters are written
        __builtin_amdgcn_s_setprio(2);                            // the consumers' MFMAs before the producers' address arithmetic
    } else issue(n, 0, 0, -1, 0, true);

    // DEFER: the accumulators of the tile before the current one, waiting for their epilogue (n_prev: its image; -1: none)
    f32x16 accP[MT][NTW];
    int n_prev = -1;
    const bool defer_ok = DEFER && !a.out_f32 && a.d2s_c == 0 && !a.res;       // plain activation-tensor outputs without a residual read
    for (;;) {
    MX_TL(1);                            // tile start
    f32x16 accs[NJ][MT][NTW];            // one accumulator set per image of the group (NJ = 1: per image)
#pragma unroll
    for (int g = 0; g < NJ; ++g)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) accs[g][i][j][e] = 0.f;
    f32x16 (&acc)[MT][NTW] = accs[0];    // (the latency loop's name for its one set)

    const int next_n = n + NJ * img_step;       // the first image of the next group
    // ---- epilogue: bias (+res) -> activation -> BN affine -> split into the output planes -> store ------------------------
    // All stores go through ONE buffer descriptor over the output tensor (hi [+ lo] [+ q] planes are one allocation): the
    // per-lane part of an address is a 32-bit pixel offset per M block, everything that depends on the channel group, the
    // image and the plane is a scalar offset, and out-of-tile pixels get an out-of-range offset (the hardware drops the
    // store) - no 64-bit address arithmetic, no predication.
    // (NJ > 1: once per image of the group, from its accumulator set)
    // BLK < 0: all blocks of accumulator set J of the tile just finished (image n + J img_step).  BLK >= 0 (DEFER): output block BLK (= nt MT + mt)
    // of the PARKED accumulators of the previous tile (image n_prev), called from inside the current tile's first chunk - activation-tensor mode,
    // no wait for the DMA in flight
    auto epilogue_of = [&](auto j_tag, auto blk_tag) {
    constexpr int J = decltype(j_tag)::value;
    constexpr int BLK = decltype(blk_tag)::value;
    f32x16 (&acc)[MT][NTW] = *[&]() -> f32x16 (*)[MT][NTW] { if constexpr (BLK >= 0) return &accP; else return &accs[J]; }();
#if MX_ABL & 32
    {
        // ablation: NO epilogue (no math, no stores; the DMA wait stays) - an upper bound of what ANY scheme that hides the epilogue under the
        // next tile's taps could win.  One add per accumulator register and a store that never happens keep the MFMAs alive.
        if (J == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float keep = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) keep += acc[mt][nt][e];
        if (keep == 1.2345e33f) a.out[0] = (f16)keep;
        return;
    }
#endif
    const int n_out = BLK >= 0 ? n_prev : n + J * img_step;
    int by_e = by, oy0_e = oy0, ox0_e = ox0;
    typedef const __attribute__((address_space(3))) float lds_cfloat;
    lds_cfloat* par_e = (lds_cfloat*)s_par;            // LDS address space: ds_read, not flat loads
    // the epilogue reads its parameters from the kernel-argument segment when it runs (a laundered pointer: the loads
    // cannot be hoisted), so they do not occupy SGPRs - or spill into VGPR lanes - during the tap loop
    // (constant address space: the loads are scalar and everything derived from them stays wave-uniform)
    typedef const __attribute__((address_space(4))) ConvMxArgs KArgs;
    KArgs* ep = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    // (two statements: an asm with any VGPR output makes ALL its outputs divergent for the compiler - the scalar offsets derived from
    // by_e then lived in VGPRs, and the residual loads that take them as soffset ran in waterfall loops)
    asm volatile("" : "+s"(by_e), "+s"(oy0_e), "+s"(ox0_e), "+s"(ep));
    asm volatile("" : "+v"(par_e));
    KArgs& a = *ep;                   // shadows the by-value argument inside the epilogue
    const int act = a.act;
    const float slope = a.slope;
    const bool has_bn = a.bn_scale != nullptr || a.force_bn != 0;
    const float acc_mul = a.acc_mul, res_mul = a.res_mul;
    auto epilogue = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;      // 0: channel-blocked act, 1: depth-to-space act, 2: fp32 NCHW
        const int oc = MODE == 1 ? a.d2s_c : a.c_out_pad;
        const int oh = MODE == 1 ? 2 * a.h_out : a.h_out, ow = MODE == 1 ? 2 * a.w_out : a.w_out;
        const unsigned ohw = (unsigned)(oh * ow);
        const bool wr_lo = MODE != 2 && a.out_plane != 0, wr_q = MODE != 2 && a.out_q_off != 0;
        const bool ql_only = a.out_q_kind == 1;              // al8-only q planes: one byte per element, no a8 plane
        // fp6 slots (the consumer runs the f16 + fp6x2 arithmetic).  Compiled in where the forward needs it: AR 3 writes ONLY fp6 q planes, AR 0
        // only fp8 ones - except its two-source instantiations (inConv.inConv.0 reads the fp8 planes of the upfeat kernel - and the gray tail chunk - and feeds
        // an fp6 layer), which carry one epilogue mode and have the registers for both (the launchers check the combination)
        constexpr bool CAN_Q6 = Q6 || (AR == 0 && NSRC2), CAN_Q8 = !Q6;
        const bool q6_out = CAN_Q6 && (!CAN_Q8 || a.out_q_kind == 2);
        constexpr float qs = 1.f;             // the epilogue leaves x 2^out_sexp: the fp8 planes take it as it is
        const __amdgpu_buffer_rsrc_t ro = MODE == 2 ? __builtin_amdgcn_make_buffer_rsrc((void*)a.out_f32, 0, a.out_bytes, 0x00020000)
                                                    : __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
        // scalar byte offsets per channel group (nt, q): hi plane (the lo plane is out_plane*2 further), a8 plane (al8 is
        // ohw*32 further); depth-to-space: virtual channel cv -> phase ph = cv / d2s_c, channel cv % d2s_c, pixel (2y+ph/2, 2x+ph%2)
        unsigned so_hi[NTW][2], so_q[NTW][2];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int cv = (by_e * NT + wn * NTW + nt) * 32 + 16 * q;
                int cb = cv; unsigned php = 0;
                if (MODE == 1) { const int ph = cv / a.d2s_c; cb = cv - ph * a.d2s_c; php = (unsigned)((ph >> 1) * ow + (ph & 1)) * 32u; }
                so_hi[nt][q] = ((unsigned)(n_out * oc) + (unsigned)cb) * ohw * 2u + php;
                so_q[nt][q] = (unsigned)a.out_q_off + ((unsigned)(n_out * oc) + (unsigned)(cb & ~31)) * ohw * (ql_only ? 1u : 2u) + (unsigned)(cb & 16) + php;
            }
        // per-lane pixel offset (bytes, 32 per pixel) of M block mt, or OOB
        unsigned vo[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int px = r % TW, py = (wm * MT + mt) * G::ROWS_PER_MB + r / TW;
            const int oy = oy0_e + py, ox = ox0_e + px;
            const bool pok = oy < a.h_out && ox < a.w_out;
            const unsigned pi = MODE == 1 ? (unsigned)(2 * oy * ow + 2 * ox) : (unsigned)(oy * ow + ox);
            vo[mt] = pok ? (MODE == 2 ? pi * 4u : pi * 32u) : OOB;
        }
        // the stores of one 32 x 32 output block (its words were parked in the accumulator registers by the math below)
        auto store_block = [&](int mt, int nt) __attribute__((always_inline)) {
#if MX_ABL & 16
            return;                     // (ablation: the epilogue's math without its stores)
#endif
            const int cob = (by_e * NT + wn * NTW + nt) * 32;
            {
                // (elements are read by value: __builtin_bit_cast applied to a vector-element lvalue reads element 0)
                const f32x16& t = acc[mt][nt];
                if (MODE != 2) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const bool cok = cob + 16 * q + 8 * kh < a.c_out;
                        const unsigned vb = (vo[mt] == OOB || !cok) ? OOB : vo[mt];
                        const i32x4 d4 = {__float_as_int(t[4 * q]), __float_as_int(t[4 * q + 1]), __float_as_int(t[4 * q + 2]), __float_as_int(t[4 * q + 3])};
                        buffer_store_b128(d4, ro, vb == OOB ? OOB : vb + 16u * kh, so_hi[nt][q]);
                        if (X3) {       // (the other arithmetics stored their lo words in phase 1)
                            const i32x4 l4 = {__float_as_int(t[8 + 4 * q]), __float_as_int(t[9 + 4 * q]), __float_as_int(t[10 + 4 * q]), __float_as_int(t[11 + 4 * q])};
                            buffer_store_b128(l4, ro, vb == OOB ? OOB : vb + 16u * kh, so_hi[nt][q] + (unsigned)a.out_plane * 2u);
                        }
                        if (wr_q && q6_out) {
                            // fp6 slots: this lane owns bytes 12 kh .. 12 kh + 11 of its pixel's slot in either plane (once per 32-channel block)
                            if (q == 0) {
                                const unsigned v6 = (vo[mt] == OOB || cob >= a.c_out) ? OOB : vo[mt] + 12u * kh;
                                buffer_store_b96(i32x3{__float_as_int(t[8]), __float_as_int(t[9]), __float_as_int(t[10])}, ro, v6, so_q[nt][0]);
                                buffer_store_b96(i32x3{__float_as_int(t[12]), __float_as_int(t[13]), __float_as_int(t[14])}, ro, v6, so_q[nt][0] + ohw * 32u);
                                // ... and the lower half-wave the two block scales (dword 6 of the slots)
                                const unsigned v7 = (v6 == OOB || kh) ? OOB : vo[mt] + 24u;
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(t[11]), ro, v7, so_q[nt][0], 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(t[15]), ro, v7, so_q[nt][0] + ohw * 32u, 0);
                            }
                        } else if (wr_q) {
                            // q planes: this lane owns bytes 8 kh .. 8 kh + 7 of its pixel's 16-byte half (cb & 16)
                            const i32x2 q0 = {__float_as_int(t[8 + 2 * q]), __float_as_int(t[9 + 2 * q])};
                            const i32x2 q1 = {__float_as_int(t[12 + 2 * q]), __float_as_int(t[13 + 2 * q])};
                            if (!ql_only) __builtin_amdgcn_raw_buffer_store_b64(q0, ro, vb == OOB ? OOB : vb + 8u * kh, so_q[nt][q], 0);
                            __builtin_amdgcn_raw_buffer_store_b64(q1, ro, vb == OOB ? OOB : vb + 8u * kh, so_q[nt][q] + (ql_only ? 0u : ohw * 32u), 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if (cob + (e & 3) + 8 * (e >> 2) >= a.c_out) continue;       // no lane has this channel (wave-uniform)
                        const int co = cob + (e & 3) + 8 * (e >> 2) + 4 * kh;        // differs between the half-waves: per-lane offset
                        const unsigned vof = (vo[mt] == OOB || co >= a.c_out) ? OOB : vo[mt] + (unsigned)(n_out * a.c_out + co) * ohw * 4u;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(t[e]), ro, vof, 0, 0);
                    }
                }
            }
        };
#if MX_EPI_INLINE
        // Round 4: the DMA wait comes FIRST (the next tile's first chunk was issued during the last chunk's first taps: long landed), and
        // every block's stores go out right behind its math, so that the address unit works through them (1 KiB per instruction at
        // 64 B/clk: ~2 000 cycles per 512 x 64 tile, profiles/r04_conv_timeline_before.txt "stores") while the VALU does the next block -
        // round 3 parked all four blocks and stored them in a phase of its own, with the waves idle behind the store queue.
        // (NJ > 1: a group's further images have nothing new in flight but the previous image's stores)
        if (J == 0 && BLK < 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        // ---- phase 1: math ----
        // Everything the hot (activation-tensor) modes do per element is branch-free and packed where the ISA has a packed form:
        // v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32 for bias, slope, BN and the hi/lo split, v_cvt_scalef32_pk_fp8_f32 (scale + RNE +
        // saturation in one instruction, two elements each) for the fp8 planes; run-time layer properties (residual, BN, lo / q planes)
        // are tested once per 32x32 block, not per element group.  Round 2's epilogue spent ~15 VALU instructions and 5 scalar
        // branches per element / 4-element group here, with the matrix pipe idle behind the chunk barrier
        // (profiles/r02_conv_pmc.txt: 4.1 VALU per MFMA on the f16+fp8x2 instantiation against 0.23 inside its tap loop).
        // Results are bit-identical to that code (tools/cvt_scale_probe.hip pins the scaled conversion against mul + v_med3 + cvt on
        // everything that rounds into the finite range; beyond it the instruction yields the NaN code, hence the slow path below).
        // activation as x = max(x, x slope + 0): ReLU = slope 0 (the "+ 0" turns -0 into +0, as fmaxf(x, 0) does), none = slope 1
        const float slope_e = act == DISCO_ACT_RELU ? 0.f : (act == DISCO_ACT_LRELU ? slope : 1.f);
        const f32x2 slope2 = {slope_e, slope_e}, zero2 = {0.f, 0.f};
        // the scaled conversions divide by their scale operand: 1 and 2^-11
        const float qinv = 1.f, qlinv = __builtin_ldexpf(1.f, -MX_LO_SHIFT);
        const f32x2 acc_mul2 = {acc_mul, acc_mul}, res_mul2 = {res_mul, res_mul};
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (BLK >= 0 && nt * MT + mt != BLK) continue;
                if constexpr (MODE != 2) {
                    f32x2 x[8];                              // pair p = elements 2p, 2p+1; 4-channel group g4 = p >> 1 (c = 8 g4 + 4 kh + 0..3)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int cl = (wn * NTW + nt) * 32 + 8 * g4 + 4 * kh;
                        const float4 b4 = *(const __attribute__((address_space(3))) float4*)(par_e + cl);
                        x[2 * g4] = __builtin_elementwise_fma(f32x2{acc[mt][nt][4 * g4], acc[mt][nt][4 * g4 + 1]}, acc_mul2, f32x2{b4.x, b4.y});
                        x[2 * g4 + 1] = __builtin_elementwise_fma(f32x2{acc[mt][nt][4 * g4 + 2], acc[mt][nt][4 * g4 + 3]}, acc_mul2, f32x2{b4.z, b4.w});
                    }
                    if (a.res) {
                        // residual (same shape and hi-plane layout as the output, its own allocation), plus its lo plane if any
                        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res_bytes, 0x00020000);
                        const bool res_lo = X3 || a.res_plane != 0;
                        f16x4 rh[4], rl[4];
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const unsigned so = so_hi[nt][g4 >> 1];
                            const unsigned vof = vo[mt] == OOB ? OOB : vo[mt] + 16u * (g4 & 1) + 8u * kh;
                            rh[g4] = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rr, vof, so, 0));
                            rl[g4] = res_lo ? __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rr, vof, so + (unsigned)a.res_plane * 2u, 0)) : f16x4{0, 0, 0, 0};
                        }
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                            for (int d = 0; d < 2; ++d) {
                                const f32x2 h2 = {(float)rh[g4][2 * d], (float)rh[g4][2 * d + 1]}, l2 = {(float)rl[g4][2 * d], (float)rl[g4][2 * d + 1]};
                                if (X3) x[2 * g4 + d] = __builtin_elementwise_fma(h2 + l2, res_mul2, x[2 * g4 + d]);      // x + (hi + lo) 2^k
                                else {                                                                                 // (no lo plane: l2 = 0)
                                    x[2 * g4 + d] = __builtin_elementwise_fma(h2, res_mul2, x[2 * g4 + d]);
                                    x[2 * g4 + d] = __builtin_elementwise_fma(l2, res_mul2, x[2 * g4 + d]);
                                }
                            }
                    }
#ifndef MX_DEV_MODE
                    if (act == DISCO_ACT_TANH) {
#pragma unroll
                        for (int p = 0; p < 8; ++p) x[p] = f32x2{tanhf(x[p][0]), tanhf(x[p][1])};
                    } else
#endif
                    if (slope_e == 0.f) {
                        // ReLU (most layers): x slope + 0 is +0 for every finite x, so the maximum takes the constant directly - the same
                        // v_max_f32 on the same operands (bit-identical), without the packed multiply-add and the wait state behind it
                        float zero = 0.f;
                        asm volatile("" : "+v"(zero));
#pragma unroll
                        for (int p = 0; p < 8; ++p) x[p] = f32x2{vmax_f32(x[p][0], zero), vmax_f32(x[p][1], zero)};
                    } else {
#pragma unroll
                        for (int p = 0; p < 8; ++p) {
                            const f32x2 t = __builtin_elementwise_fma(x[p], slope2, zero2);
                            x[p] = f32x2{vmax_f32(x[p][0], t[0]), vmax_f32(x[p][1], t[1])};
                        }
                    }
                    if (has_bn) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int cl = (wn * NTW + nt) * 32 + 8 * g4 + 4 * kh;
                            const float4 s4 = *(const __attribute__((address_space(3))) float4*)(par_e + 32 * NT + cl);
                            const float4 h4 = *(const __attribute__((address_space(3))) float4*)(par_e + 64 * NT + cl);
                            x[2 * g4] = __builtin_elementwise_fma(x[2 * g4], f32x2{s4.x, s4.y}, f32x2{h4.x, h4.y});
                            x[2 * g4 + 1] = __builtin_elementwise_fma(x[2 * g4 + 1], f32x2{s4.z, s4.w}, f32x2{h4.z, h4.w});
                        }
                    }
                    // fp16 hi (+ lo) in pairs; fp8 a8 / al8 in quads: dword g of the fp8 planes = channels 8 g + 4 kh + 0..3
                    unsigned hd[8], ld[8], a8[4], l8[4];
                    f32x2 lf[8];
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        const f16x2 h2 = __builtin_convertvector(x[p], f16x2);
                        // x - float(h) as fma(float(h), -1, x): one v_fma_mix_f32 per element (the f16 -> f32 conversion is an operand
                        // modifier) where the packed subtraction needed two conversions first; the same single rounding
                        // (asm: the compiler canonicalises the fma back into two conversions and a packed subtraction)
                        const unsigned hb = __builtin_bit_cast(unsigned, h2);
                        float l0, l1;
                        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(x[p][0]));
                        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(x[p][1]));
                        lf[p] = f32x2{l0, l1};
                        hd[p] = __builtin_bit_cast(unsigned, h2);
                    }
                    if (X3 || wr_lo) {
#pragma unroll
                        for (int p = 0; p < 8; ++p) ld[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(lf[p], f16x2));
                    }
                    if (!X3 && wr_q) {
                        float bmax = 0.f;                    // max |x| of this lane's 16 values
#pragma unroll
                        for (int p = 0; p < 8; p += 2) {
                            bmax = fmaxf(fmaxf(bmax, fabsf(x[p][0])), fabsf(x[p][1]));
                            bmax = fmaxf(fmaxf(bmax, fabsf(x[p + 1][0])), fabsf(x[p + 1][1]));
                        }
                        if (q6_out) {
                            // fp6 slots: ONE v_cvt_scalef32_pk32_fp6_f16 turns this lane's 16 hi words and its 16 lo residuals (x 2^12, as
                            // fp16) into the 96 + 96 bits of its half of the pixel's a6 and al6 slots: fields in register order = channels
                            // 8 g + 4 kh + i (Act::q_kind 2), divided by the block scale 2^(E - 2), rounded to nearest even, saturated at
                            // +-7.5 (tools/fp6_probe.hip) - no permlanes for the data, no clamp path, nothing to count: the block scale
                            // (mx6_block_scale() of the largest |hi| among the pixel's 32 channels = this lane's 16 and its partner's in the
                            // other half-wave) follows the data wherever they go
                            const unsigned bmb = __float_as_uint(bmax);
                            const auto bsw = __builtin_amdgcn_permlane32_swap(bmb, bmb, false, false);
                            const float pmax = fmaxf(__uint_as_float(bsw[0]), __uint_as_float(bsw[1]));
                            const unsigned sa = (unsigned)mx6_block_scale((f16)pmax);
                            const float bscale = __uint_as_float(sa << 23);
                            f16x32 src;
#pragma unroll
                            for (int p = 0; p < 8; ++p) {
                                const f16x2 h2 = __builtin_bit_cast(f16x2, hd[p]);
                                const f16x2 l2 = __builtin_convertvector(lf[p] * 4096.f, f16x2);
                                src[2 * p] = h2[0]; src[2 * p + 1] = h2[1]; src[16 + 2 * p] = l2[0]; src[16 + 2 * p + 1] = l2[1];
                            }
                            const i32x6 r6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(src, bscale);
                            a8[0] = (unsigned)r6[0]; a8[1] = (unsigned)r6[1]; a8[2] = (unsigned)r6[2]; a8[3] = sa;
                            l8[0] = (unsigned)r6[3]; l8[1] = (unsigned)r6[4]; l8[2] = (unsigned)r6[5]; l8[3] = sa - 1u;
                        } else {
                        // The conversion does not saturate (a value that rounds above 448 becomes the NaN code 0x7f), and calibration
                        // leaves 14x headroom: only when a value of this block is beyond the range (|al8| <= |a8| by construction, so
                        // max |x| decides; wave-uniform test) its operands are clamped first - in place, and counted
                        if (__builtin_expect(__ballot(!(bmax * qs <= 448.f)) != 0ull, 0)) {
                            const float lim = 448.f, liml = 448.f * qlinv;
                            unsigned cnt = 0;
#pragma unroll
                            for (int p = 0; p < 8; ++p)
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    const float xc = __builtin_amdgcn_fmed3f(x[p][e], -lim, lim), lc = __builtin_amdgcn_fmed3f(lf[p][e], -liml, liml);
                                    cnt += (ql_only ? 0 : (xc != x[p][e])) + (lc != lf[p][e]);      // al8-only tensors have no a8 plane to clamp
                                    x[p][e] = xc; lf[p][e] = lc;
                                }
                            if (a.sat && cnt) atomicAdd(a.sat, cnt);
                        }
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            a8[g] = fp8x4_scaled(x[2 * g], x[2 * g + 1], qinv);
                            l8[g] = fp8x4_scaled(lf[2 * g], lf[2 * g + 1], qlinv);
                        }
                        }
                    }
                    // v_permlane32_swap: lower.(group 1) <-> upper.(group 0), lower.(group 3) <-> upper.(group 2): the lower
                    // half-wave then holds channels 0-7 and 16-23 of its pixel, the upper half 8-15 and 24-31
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            auto sh = __builtin_amdgcn_permlane32_swap(hd[4 * q + d], hd[4 * q + 2 + d], false, false);
                            hd[4 * q + d] = sh[0]; hd[4 * q + 2 + d] = sh[1];
                        }
                    if (X3 || wr_lo) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int d = 0; d < 2; ++d) {
                                auto sl = __builtin_amdgcn_permlane32_swap(ld[4 * q + d], ld[4 * q + 2 + d], false, false);
                                ld[4 * q + d] = sl[0]; ld[4 * q + 2 + d] = sl[1];
                            }
                    }
                    if (!X3 && wr_q && !q6_out) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            auto sa = __builtin_amdgcn_permlane32_swap(a8[2 * q], a8[2 * q + 1], false, false);
                            a8[2 * q] = sa[0]; a8[2 * q + 1] = sa[1];
                            auto sl = __builtin_amdgcn_permlane32_swap(l8[2 * q], l8[2 * q + 1], false, false);
                            l8[2 * q] = sl[0]; l8[2 * q + 1] = sl[1];
                        }
                    }
                    // All output words are parked in the accumulator registers and stored in phase 3: hi in 0-7; f16x3: lo in 8-15;
                    // otherwise a8 in 8-11, al8 in 12-15, and the lo words of the few tensors that carry lo AND fp8 planes (the HourGlass2's
                    // residual chain) are stored right here through buffer_store_b128(), whose wait states cover the store-data hazard that
                    // round 2 ran into at this very place (a plain store, the next iteration's first VALU write in a data register one
                    // instruction behind it: lanes 12-15 / 28-31 of one dword arrived stale, differently from run to run).  Parking them
                    // in 32 more registers pushed the two-source instantiation to 256 VGPRs and its SGPR spills into scratch memory.
#pragma unroll
                    for (int d = 0; d < 8; ++d) acc[mt][nt][d] = __builtin_bit_cast(float, hd[d]);
                    if constexpr (X3) {
#pragma unroll
                        for (int d = 0; d < 8; ++d) acc[mt][nt][8 + d] = __builtin_bit_cast(float, ld[d]);
                    } else {
                        if (wr_lo) {
                            const int cob = (by_e * NT + wn * NTW + nt) * 32;
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const bool cok = cob + 16 * q + 8 * kh < a.c_out;
                                const unsigned vb = (vo[mt] == OOB || !cok) ? OOB : vo[mt] + 16u * kh;
                                buffer_store_b128(i32x4{(int)ld[4 * q], (int)ld[4 * q + 1], (int)ld[4 * q + 2], (int)ld[4 * q + 3]}, ro, vb, so_hi[nt][q] + (unsigned)a.out_plane * 2u);
                            }
                        }
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            acc[mt][nt][8 + g] = __builtin_bit_cast(float, a8[g]);
                            acc[mt][nt][12 + g] = __builtin_bit_cast(float, l8[g]);
                        }
                    }
                } else {
                float v[16];
                // fp32 NCHW outputs (pred_mask0 + softmax, outConv + tanh: two launches per forward): per group of 4 channels
                // (c = 8 g4 + 4 kh + 0..3) parameters from LDS, bias -> activation -> BN affine
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int cl = (wn * NTW + nt) * 32 + 8 * g4 + 4 * kh;
                    // (outConv has 2 output channels, pred_mask0 9: a group of 8 channels beyond c_out is skipped - wave-uniform - with
                    // its 16 tanhf / expf per lane)
                    if ((by_e * NT + wn * NTW + nt) * 32 + 8 * g4 >= a.c_out) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[4 * g4 + j] = 0.f;
                        continue;
                    }
                    const float4 b4 = *(const __attribute__((address_space(3))) float4*)(par_e + cl);
                    float x[4] = {fmaf(acc[mt][nt][4 * g4], acc_mul, b4.x), fmaf(acc[mt][nt][4 * g4 + 1], acc_mul, b4.y), fmaf(acc[mt][nt][4 * g4 + 2], acc_mul, b4.z),
                                  fmaf(acc[mt][nt][4 * g4 + 3], acc_mul, b4.w)};
                    if (act == DISCO_ACT_RELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) x[j] = fmaxf(x[j], 0.f);
                    } else if (act == DISCO_ACT_LRELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) x[j] = fmaxf(x[j], x[j] * slope);
                    } else if (act == DISCO_ACT_TANH) {
                        // tanh x = 1 - 2 / (exp(2x) + 1) on v_exp_f32 / v_rcp_f32: absolute error <= 3e-7 (saturates correctly: exp -> inf gives 1,
                        // exp -> 0 gives -1).  libm's tanhf cost outConv's epilogue 5 000 of the tile's 17 400 cycles (30 %), two thirds of them for
                        // channels that do not exist: channels j and 4 + j of the group beyond c_out are skipped (wave-uniform)
                        const int cg = (by_e * NT + wn * NTW + nt) * 32 + 8 * g4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (cg + j >= a.c_out) { x[j] = 0.f; continue; }
                            const float e2 = __builtin_amdgcn_exp2f(x[j] * 2.885390081777927f);      // exp(2x) = 2^(2x log2 e)
                            x[j] = 1.f - 2.f * __builtin_amdgcn_rcpf(e2 + 1.f);
                        }
                    }
                    if (has_bn) {
                        const float4 s4 = *(const __attribute__((address_space(3))) float4*)(par_e + 32 * NT + cl);
                        const float4 h4 = *(const __attribute__((address_space(3))) float4*)(par_e + 64 * NT + cl);
                        x[0] = x[0] * s4.x + h4.x; x[1] = x[1] * s4.y + h4.y; x[2] = x[2] * s4.z + h4.z; x[3] = x[3] * s4.w + h4.w;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[4 * g4 + j] = x[j];
                }
                if (a.softmax) {
                    const int cob = (by_e * NT + wn * NTW + nt) * 32;
                    float mx = -INFINITY;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (cob + (e & 3) + 8 * (e >> 2) + 4 * kh < a.c_out) mx = fmaxf(mx, v[e]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    float sm = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if (cob + 8 * (e >> 2) >= a.c_out) { v[e] = 0.f; continue; }          // wave-uniform
                        v[e] = cob + (e & 3) + 8 * (e >> 2) + 4 * kh < a.c_out ? expf(v[e] - mx) : 0.f;
                        sm += v[e];
                    }
                    sm += __shfl_xor(sm, 32);
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] / sm;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = v[e];
                }
#if MX_EPI_INLINE
                store_block(mt, nt);
#endif
            }
        }
        // ---- phase 2: the next image's first chunk has landed ----
        MX_TL(7);                        // epilogue math done
#if !MX_EPI_INLINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        MX_TL(8);                        // ... and the residual loads / next tile's first DMA have landed
        // ---- phase 3: stores (round 3's order: MX_EPI_INLINE 0) ----
#if !MX_EPI_INLINE
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) store_block(mt, nt);
#endif
    };
#ifdef MX_DEV_MODE       // ISA inspection builds: one epilogue mode only
    epilogue(std::integral_constant<int, MX_DEV_MODE>{});
#else
    // two-source (concat-on-read) layers only ever write plain activation tensors (the launchers refuse anything else): their
    // instantiations carry ONE epilogue, which keeps them under the register budget without scratch memory (round 2: 59-87 SGPR
    // spills through 32 B/lane of scratch in the instantiation that serves inConv.inConv.0 and both `combine` layers)
    if constexpr (NSRC2 || BLK >= 0) epilogue(std::integral_constant<int, 0>{});
    else if (a.out_f32) epilogue(std::integral_constant<int, 2>{});
    else if (a.d2s_c > 0) epilogue(std::integral_constant<int, 1>{});
    else epilogue(std::integral_constant<int, 0>{});
#endif
    };
    // one chunk: wait for its DMA, barrier, then 9 taps with the next chunk's DMA issued in ninths between them.
    // ISQ = false: H chunk, planes = channels 0-15 / 16-31 (fp16); a lane feeds k = 8 kh .. 8 kh + 7 of both planes to two
    //              K = 16 MFMAs.  ISQ = true: Q chunk, planes = a8 / al8; lane half kh reads all 32 bytes of plane kh
    //              (K half kh of one K = 64 MFMA: a8 meets wl8, al8 meets w8).
    // The H and Q chunks of a 32-channel group run back to back in one loop iteration (no branch between the two bodies:
    // a branch made the register allocator keep the accumulators in two places).
    // KIND 3 (TAIL): an H chunk of which only plane 0 (16 channels) exists - the last, odd chunk of a two-source f16+fp8x2 layer whose
    //              second source is a 16-channel fp16 tensor without q planes (HourGlass2 input: the gray image as the channels
    //              (g_hi, g_lo, g_hi) against the weights (w_h, w_h, w_l): an exact three-product split in ONE K = 16 MFMA per tap
    //              where a padded 32-channel chunk pair spent three).  Its second-plane DMA pieces and weight pieces are not issued.
    // KIND 2 (X3): the f16x3 arithmetic of conv_mfma2.hip on this skeleton - planes = hi / lo of 16 channels, weight tile = w_hi / w_lo
    //              fragments (conv3x3_pack_host's image), three K = 16 MFMAs per tap in conv_mfma2's order (w_lo a_hi, w_hi a_lo,
    //              w_hi a_hi), so results are bit-identical to that kernel's
    // NJ > 1: a STAGE = (chunk ck, image J of the group); the next stage is the same chunk of image J + 1 (pixels only) or, behind the
    // group's last image, the next chunk of image 0 with its weights (into the other weight buffer, which the previous chunk left with its
    // last stage)
    auto chunk = [&](auto kind_tag, int ck, auto j_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr int J = decltype(j_tag)::value;
        constexpr bool ISQ = KIND == 1, TAIL = KIND == 3;
        f32x16 (&acc)[MT][NTW] = accs[J];
        MX_TL(2);                        // chunk start
        if (!(ck == 0 && J == 0 && dma_waited)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (GENC1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's share of the computed pixel tile has been written
        MX_TL(3);                        // this chunk's DMA has landed (own pieces)
        __builtin_amdgcn_s_barrier();
        MX_TL(ISQ ? 5 : 4);              // barrier passed: taps of an H (4) / Q (5) chunk begin
        const bool more = ck + 1 < nchunks;
        constexpr bool new_chunk = J == NJ - 1;             // the next stage opens a chunk
        const int dma_img = !new_chunk ? n + (J + 1) * img_step : (more ? n : (next_n < a.n ? next_n : n));
        const int dma_ck = !new_chunk ? ck : (more ? ck + 1 : 0);
        const char* sA = smem + buf * BUF_BYTES;
        const char* sW = smem + (NJ == 1 ? buf : wbuf) * BUF_BYTES + A_BYTES;
        buf ^= 1;
        const int wnext = wbuf ^ 1;
        if (NJ > 1 && new_chunk) wbuf = wnext;
        // the tap ORDER is a property of the tile width alone (never of how the tile is split over waves): every instantiation
        // that can serve a given layer shape accumulates in the same order, so a result does not depend on the batch size
        constexpr bool COLMAJOR = STRIDE == 1 && G::ROWS_PER_MB == 1;
        constexpr bool ROWREUSE = COLMAJOR && MT >= 2;        // M block mt at ky reads tile row mt + ky = what block mt + 1 read at ky - 1
        const int asc = (NSRC2 && !X3 && ((ck >> 1) << 5) >= c_src0) ? asc1 : asc0;
        // this chunk's three column addresses inside the current buffer (the only per-chunk address arithmetic)
        const int bufoff = (int)(sA - smem);
        int ca0[3], ca1[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            ca0[kx] = (ISQ ? colQ[kx] : colH[kx]) + bufoff;
            ca1[kx] = ISQ ? (ca0[kx] ^ 16) : ca0[kx] + PLANE_B;           // all other address terms are multiples of 32
        }
        int wo = w_off;
        asm volatile("" : "+v"(wo));
        i32x4 ra[MT][2];
        i32x4 rb[NTW][2];
#pragma unroll
        for (int slot = 0; slot < 9; ++slot) {
            const int ky = COLMAJOR ? slot % 3 : slot / 3, kx = COLMAJOR ? slot / 3 : slot % 3;
            const int tap = ky * 3 + kx;
#if !(MX_ABL & 1)
            issue(dma_img, dma_ck, buf, slot, wnext, new_chunk);
#endif
            if (GENC1) {
                // the next chunk's pixel tile, a third per wave in taps 0, 3 and 6; in the tile's first chunk also the NEXT image's gray tile
                // (into the other gray buffer: complete and visible from the next chunk's barrier on, needed in the tile's last chunk)
                if (slot == 0 && ck == 0 && next_n < a.n) issue_gray(next_n, gbuf ^ 1);
                if (slot % 3 == 0 && slot / 3 < GEN_PER_WAVE && (more || next_n < a.n))
                    gen_c1((slot / 3) * NWAVE + wave, dma_ck, more ? gbuf : (gbuf ^ 1), buf);
            }
            const bool live = !MASKED || ((tmask >> tap) & 1u);
            if (!ROWREUSE && !live) continue;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (ROWREUSE && ky > 0 && mt + 1 < MT) { ra[mt][0] = ra[mt + 1][0]; ra[mt][1] = ra[mt + 1][1]; continue; }
#if MX_ABL & 2
                if (slot > 0) continue;
#endif
                constexpr int RB = G::PITCH * 32;                           // bytes per tile row
                const int rowc = (mt * G::ROWS_PER_MB * STRIDE + ky) * RB;   // compile-time constant: the ds_read's immediate offset
                ra[mt][0] = *reinterpret_cast<const i32x4*>(smem + ca0[kx] + rowc);
                if (!TAIL) ra[mt][1] = *reinterpret_cast<const i32x4*>(smem + ca1[kx] + rowc);
            }
            if (ROWREUSE && !live) continue;
#if MX_ABL & 2
            static_assert(true, "");
#endif
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
#if MX_ABL & 2
                if (slot > 0) continue;
#endif
                const int off = wo + nt * W_NB + tap * 2 * WBLK;
                rb[nt][0] = *reinterpret_cast<const i32x4*>(sW + off);
                if (!TAIL) rb[nt][1] = *reinterpret_cast<const i32x4*>(sW + off + WBLK);
            }
            if ((slot + (wave >= NWAVE / 2 ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    if (ISQ) {
                        const i32x8 bw = {rb[nt][0][0], rb[nt][0][1], rb[nt][0][2], rb[nt][0][3], rb[nt][1][0], rb[nt][1][1], rb[nt][1][2], rb[nt][1][3]};
                        const i32x8 ap = {ra[mt][0][0], ra[mt][0][1], ra[mt][0][2], ra[mt][0][3], ra[mt][1][0], ra[mt][1][1], ra[mt][1][2], ra[mt][1][3]};
                        // fp6 slots carry their own E8M0 block scale (per pixel - or per output channel and tap - and 32 channels) in byte 24 =
                        // dword 6 of the fragment, which the MFMA ignores as operand data: both scale operands of a lane come straight out of
                        // its fragment registers
                        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bw, ap, acc[mt][nt], QFMT, QFMT, 0, Q6 ? rb[nt][1][2] : wsc[nt], 0, Q6 ? ra[mt][1][2] : asc);
                    } else if (KIND == 2) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, rb[nt][1]), __builtin_bit_cast(f16x8, ra[mt][0]), acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, rb[nt][0]), __builtin_bit_cast(f16x8, ra[mt][1]), acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, rb[nt][0]), __builtin_bit_cast(f16x8, ra[mt][0]), acc[mt][nt], 0, 0, 0);
                    } else {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, rb[nt][0]), __builtin_bit_cast(f16x8, ra[mt][0]), acc[mt][nt], 0, 0, 0);
                        if (!TAIL) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, rb[nt][1]), __builtin_bit_cast(f16x8, ra[mt][1]), acc[mt][nt], 0, 0, 0);
                    }
                }
            // pin this tap's MFMAs here: without a use of the accumulators the optimiser sinks the whole (pure) MFMA chain of a
            // chunk below its last tap, which hoists all 9 taps of fragment reads above it (~200 VGPRs, spills)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    float pin = acc[mt][nt][0];
                    asm volatile("" : "+v"(pin));
                    acc[mt][nt][0] = pin;
                }
            if constexpr (DEFER && J == 0) {
                // the previous tile's epilogue, one output block behind each of the first taps of this tile's first chunk: its VALU work and
                // stores run next to the MFMAs of this wave's SIMD partner and of its own next taps
                if (ck == 0 && n_prev >= 0) {
                    if (slot == DEFER_SLOT0) epilogue_of(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                    if (slot == DEFER_SLOT0 + DEFER_STEP && MT * NTW > 1) epilogue_of(std::integral_constant<int, 0>{}, std::integral_constant<int, (MT * NTW > 1 ? 1 : 0)>{});
                    if (slot == DEFER_SLOT0 + 2 * DEFER_STEP && MT * NTW > 2) epilogue_of(std::integral_constant<int, 0>{}, std::integral_constant<int, (MT * NTW > 2 ? 2 : 0)>{});
                    if (slot == DEFER_SLOT0 + 3 * DEFER_STEP && MT * NTW > 3) epilogue_of(std::integral_constant<int, 0>{}, std::integral_constant<int, (MT * NTW > 3 ? 3 : 0)>{});
                }
            }
        }
        if constexpr (DEFER && J == 0) { if (ck == 0) n_prev = -1; }
    };
    using KH = std::integral_constant<int, 0>; using KQ = std::integral_constant<int, 1>; using K3 = std::integral_constant<int, 2>; using KT = std::integral_constant<int, 3>;
    if constexpr (NB == 3) {
        // ---- the latency loop (see the template parameter NB) ---------------------------------------------------------------------
        //     chunk s, tap t < 8:  read frags(t + 1) | MFMAs(t)
        //     chunk s, tap 8:      wait for chunk s + 1 (own pieces; chunk s + 2's may stay in flight), own reads of chunk s returned,
        //                          s_barrier; issue chunk s + 3 into chunk s's buffer; read frags(chunk s + 1, tap 0) | MFMAs(8)
        // Buffer protocol: the barrier B(s+1) inside tap 8 of chunk s separates every wave's last read of chunk s's buffer from the first DMA
        // write into it (chunk s + 3's, issued right behind the barrier) and every wave's DMA pieces of chunk s + 1 from the first read of
        // them.  A tile's first chunk has its barrier at the top (the previous tile's epilogue lies in between and has waited for everything
        // in flight).  The barrier skew and a chunk's first LDS latency run under the previous tap's MFMAs.
        struct Frag { i32x4 a0, a1, b0, b1; };
        constexpr bool COLMAJOR = STRIDE == 1 && G::ROWS_PER_MB == 1;      // (the tap order of the tile width: as in the throughput loop)
        auto slot_tap = [](int slot) { const int ky = COLMAJOR ? slot % 3 : slot / 3, kx = COLMAJOR ? slot / 3 : slot % 3; return ky * 3 + kx; };
        auto slot_live = [&](int slot) -> bool { return !MASKED || ((tmask >> slot_tap(slot)) & 1u); };
        auto load_frags = [&](auto kind_tag, int cb, int slot, Frag& f) __attribute__((always_inline)) {
            constexpr bool ISQ = decltype(kind_tag)::value == 1;
            const int ky = COLMAJOR ? slot % 3 : slot / 3, kx = COLMAJOR ? slot / 3 : slot % 3;
            const int tap = ky * 3 + kx;
            int bufoff = cb * BUF_BYTES;
            asm("" : "+s"(bufoff));                                  // an opaque scalar, added per read (no table of 9 addresses in registers)
            const int c0 = (ISQ ? colQ[kx] : colH[kx]) + bufoff;
            const int c1 = ISQ ? (c0 ^ 16) : c0 + PLANE_B;
            constexpr int RB = G::PITCH * 32;
            const int rowc = ky * RB;                                  // (MT = 1: the wave's one M block; compile-time: the ds_read's immediate)
            f.a0 = *reinterpret_cast<const i32x4*>(smem + c0 + rowc);
            f.a1 = *reinterpret_cast<const i32x4*>(smem + c1 + rowc);
            const char* sW = smem + bufoff + A_BYTES;
            const int off = w_off + tap * 2 * WBLK;
            f.b0 = *reinterpret_cast<const i32x4*>(sW + off);
            f.b1 = *reinterpret_cast<const i32x4*>(sW + off + WBLK);
        };
        // Taps go in GROUPS of three: the fragments of a group are read (12 ds_read_b128) while the previous group's MFMAs - 9 / 6 / 3 of
        // them, back to back on the wave's one accumulator - run.  A wave alone on its SIMD pays for every instruction that stands between two
        // MFMAs of one accumulator chain (MI355X_MICROARCH.md: +43 cycles for the first issue slot in such a gap, ~6 for each further one), so
        // the chain is broken three times per chunk, not nine.
        struct Grp { Frag t[3]; };
        Grp pf;
#pragma unroll
        for (int i = 0; i < 3; ++i) pf.t[i] = Frag{i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}};
        auto load_group = [&](auto kind_tag, int cb, int g, Grp& f) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (slot_live(3 * g + i)) load_frags(kind_tag, cb, 3 * g + i, f.t[i]);
        };
        // FIRST / LAST: the tile's first / last chunk, as compile-time flags - with run-time branches the compiler's wait-count pass merges
        // the paths and makes a group's MFMA run wait for LDS reads it does not use
        auto chunk_lat = [&](auto kind_tag, auto next_tag, auto first_tag, auto last_tag) {
            constexpr int KIND = decltype(kind_tag)::value;
            constexpr bool ISQ = KIND == 1;
            constexpr bool pf_valid = !decltype(first_tag)::value, last = decltype(last_tag)::value;
            MX_TL(2);
            Grp cur = pf;
            if constexpr (!pf_valid) {
                MX_TL(3);
                __builtin_amdgcn_s_barrier();                     // B(s) of the tile's first chunk: the producers have seen it land
                load_group(kind_tag, l_cbuf, 0, cur);
            }
            MX_TL(ISQ ? 5 : 4);
            const int nbuf = l_cbuf == NB - 1 ? 0 : l_cbuf + 1;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                Grp nxt = cur;
                if (g < 2) load_group(kind_tag, l_cbuf, g + 1, nxt);
                else if constexpr (!last) {
                    // B(s + 1): every fragment read of this chunk has returned (its buffer is the producers' from here on); behind it the next
                    // chunk is complete in LDS
                    MX_TL(10);
                    // (the builtin, not an asm: the compiler's wait-count pass then KNOWS that this chunk's reads have returned; behind an opaque
                    // asm it believed them outstanding next to the 12 new ones - more than the 4-bit counter can express - and made the
                    // MFMA run below wait for the new reads)
                    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0)
                    __builtin_amdgcn_s_barrier();
                    MX_TL(11);
                    load_group(next_tag, nbuf, 0, pf);
                }
                // this group's fragments have returned (the next group's 12 reads may stay in flight): ONE wait in front of the run, none inside
                // it - an s_waitcnt between two MFMAs of the chain is an issue slot like any other
                if (g < 2) __builtin_amdgcn_s_waitcnt(0xcc7f);        // lgkmcnt(12)
                else if constexpr (last) __builtin_amdgcn_s_waitcnt(0xc07f);
                // (the scheduler otherwise sinks the reads into the MFMA run)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (!slot_live(3 * g + i)) continue;
                    const Frag& f = cur.t[i];
                    if (ISQ) {
                        const i32x8 bw = {f.b0[0], f.b0[1], f.b0[2], f.b0[3], f.b1[0], f.b1[1], f.b1[2], f.b1[3]};
                        const i32x8 ap = {f.a0[0], f.a0[1], f.a0[2], f.a0[3], f.a1[0], f.a1[1], f.a1[2], f.a1[3]};
                        acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bw, ap, acc[0][0], QFMT, QFMT, 0, Q6 ? f.b1[2] : wsc[0], 0, Q6 ? f.a1[2] : asc0);
                    } else if (KIND == 2) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.b1), __builtin_bit_cast(f16x8, f.a0), acc[0][0], 0, 0, 0);
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.b0), __builtin_bit_cast(f16x8, f.a1), acc[0][0], 0, 0, 0);
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.b0), __builtin_bit_cast(f16x8, f.a0), acc[0][0], 0, 0, 0);
                    } else {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.b0), __builtin_bit_cast(f16x8, f.a0), acc[0][0], 0, 0, 0);
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.b1), __builtin_bit_cast(f16x8, f.a1), acc[0][0], 0, 0, 0);
                    }
                }
                {
                    float pin = acc[0][0][0];                  // (pins this group's MFMAs here: see the throughput loop)
                    asm volatile("" : "+v"(pin));
                    acc[0][0][0] = pin;
                }
                cur = nxt;
            }
            l_cbuf = nbuf;
        };
        using Yes = std::true_type; using No = std::false_type;
        if constexpr (X3) {
            if (nchunks == 1) chunk_lat(K3{}, K3{}, Yes{}, Yes{});
            else {
                chunk_lat(K3{}, K3{}, Yes{}, No{});
                for (int ck = 1; ck + 1 < nchunks; ++ck) chunk_lat(K3{}, K3{}, No{}, No{});
                chunk_lat(K3{}, K3{}, No{}, Yes{});
            }
        } else {
            // (H, Q) pairs: nchunks is even
            chunk_lat(KH{}, KQ{}, Yes{}, No{});
            for (int ck = 2; ck < nchunks; ck += 2) {
                chunk_lat(KQ{}, KH{}, No{}, No{});
                chunk_lat(KH{}, KQ{}, No{}, No{});
            }
            chunk_lat(KQ{}, KH{}, No{}, Yes{});
        }
    } else {
        using J0 = std::integral_constant<int, 0>;
        // every image of the group through one chunk
        auto stages = [&](auto kind_tag, int ck) __attribute__((always_inline)) {
            chunk(kind_tag, ck, J0{});
            if constexpr (NJ > 1) chunk(kind_tag, ck, std::integral_constant<int, 1>{});
            if constexpr (NJ > 2) chunk(kind_tag, ck, std::integral_constant<int, 2>{});
            if constexpr (NJ > 3) chunk(kind_tag, ck, std::integral_constant<int, 3>{});
        };
        if constexpr (X3) {
            for (int ck = 0; ck < nchunks; ++ck) stages(K3{}, ck);
        } else if constexpr (XQ) {
            for (int ck = 0; ck < nchunks; ck += 5) {
#pragma unroll 1
                for (int h = 0; h < 4; ++h) chunk(KH{}, ck + h, J0{});     // H, L, H, L: the same code, other weights
                chunk(KQ{}, ck + 4, J0{});
            }
        } else {
            for (int ck = 0; ck + 1 < nchunks; ck += 2) {
                stages(KH{}, ck);
                stages(KQ{}, ck + 1);
            }
            if constexpr (NSRC2 && AR == 0) {
                if (nchunks & 1) chunk(KT{}, nchunks - 1, J0{});
            }
        }
    }

    MX_TL(6);                            // taps done
    using AllBlocks = std::integral_constant<int, -1>;
    bool parked = false;
    if constexpr (DEFER) {
        // not the workgroup's last tile: park the sums; their epilogue runs inside the next tile's first chunk
        if (defer_ok && next_n < a.n) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) accP[i][j] = accs[0][i][j];
            n_prev = n;
            parked = true;
        }
    }
    if (!parked) {
        epilogue_of(std::integral_constant<int, 0>{}, AllBlocks{});
        if constexpr (NJ > 1) epilogue_of(std::integral_constant<int, 1>{}, AllBlocks{});
        if constexpr (NJ > 2) epilogue_of(std::integral_constant<int, 2>{}, AllBlocks{});
        if constexpr (NJ > 3) epilogue_of(std::integral_constant<int, 3>{}, AllBlocks{});
    }
    dma_waited = !parked;
    MX_TL(9);                            // stores issued
    gbuf ^= 1;
    n = next_n;
    if (n >= a.n) break;
    }
#if MX_TIMELINE
    if (tl_on && lane == 0) g_mx_tl[tl_w][0] = (unsigned long long)tl_n;
#endif
#endif
}

inline int num_cus_mx() { return num_cus_current(); }

// NJ > 1: images per workgroup must come in whole groups of NJ: mx_groups_nj() says whether a launch can (and how many image slots to use)
inline int mx_groups(long combos, int n, int nj) {
    const long cus = num_cus_mx();
    long best_cost = -1; int groups = 0;
    for (int g = 1; g <= n; ++g) {
        if (nj > 1 && n % (g * nj)) continue;                   // every workgroup walks n / g images: a multiple of nj
        const long cost = (long)cdiv((long)combos * g, cus) * cdiv(n, g);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; groups = g; }
    }
    return groups;                                              // 0: no such split
}

template <int TW, int TH, int NT, int STRIDE, int WM, int WN, bool MASKED, bool NSRC2, int AR = 0, bool GENC1 = false, int NB = 2, int NJ = 1>
int launch_mx4(const ConvMxArgs& a, hipStream_t s) {
    using G = GeoMx<TW, TH, STRIDE>;
    constexpr int A_BYTES = ((2 * G::NPIX * 2 + 63) / 64) * 1024;
    // GENC1: + two gray tiles (256-byte pieces) and 10 floats per input channel of the fused producer (64 channels at most)
    constexpr int GT_BYTES = (((G::TWI + 2) * (G::THI + 2) + 63) / 64) * 256;
    // NB = 3: + the 1 KiB dump area of the out-of-range DMA pieces
    constexpr int smem = NB * (A_BYTES + NT * 18 * 1024) + 3 * 32 * NT * 4 + (GENC1 ? 2 * GT_BYTES + 64 * 10 * 4 : 0) + (NB == 3 ? 1024 : 0);
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_mx_kernel<TW, TH, NT, STRIDE, WM, WN, MASKED, NSRC2, AR, GENC1, NB, NJ>;
    // function attributes are per device and per kernel instantiation (this static lives in the instantiation)
    static std::atomic<int> attr_done[DISCO_MAX_DEVICES];
    DISCO_HIP_CHECK(set_dyn_lds_once(attr_done, reinterpret_cast<const void*>(kern), smem));
    const int combos = cdiv(a.w_out, TW) * cdiv(a.h_out, TH) * cdiv(a.c_out, 32 * NT);
    const int groups = mx_groups(combos, a.n, NJ);
    if (groups <= 0) return DISCO_ESHAPE;                       // (NJ > 1: the dispatcher has asked mx_groups() before coming here)
    dim3 grid(combos, groups);
    hipLaunchKernelGGL(kern, grid, dim3((WM * WN + (NB == 3 ? 4 : 0)) * 64), smem, s, a);
    DISCO_LAUNCH_CHECK("conv3x3_mx_kernel");
    return DISCO_OK;
}
template <int TW, int TH, int NT, int STRIDE, int WM, int WN, int AR, int NB = 2>
int launch_mx2(const ConvMxArgs& a, hipStream_t s) {
    if constexpr (AR == 1) return a.tapmask ? launch_mx4<TW, TH, NT, STRIDE, WM, WN, true, false, 1>(a, s) : launch_mx4<TW, TH, NT, STRIDE, WM, WN, false, false, 1>(a, s);
    else {
        if constexpr (!(AR == 0 && NB == 3)) {       // (the f16+fp8x2 two-source layer ends in a tail chunk: throughput loop only; the dispatcher does not come here with it)
            if (a.nsrc > 1) return a.tapmask ? launch_mx4<TW, TH, NT, STRIDE, WM, WN, true, true, AR, false, NB>(a, s) : launch_mx4<TW, TH, NT, STRIDE, WM, WN, false, true, AR, false, NB>(a, s);
        }
        return a.tapmask ? launch_mx4<TW, TH, NT, STRIDE, WM, WN, true, false, AR, false, NB>(a, s) : launch_mx4<TW, TH, NT, STRIDE, WM, WN, false, false, AR, false, NB>(a, s);
    }
}

// DISCO_CONV_LAT=0: the latency loop off (A/B runs: small grids then take round 4's tiles and loop; results are bit-identical either way)
inline bool conv_lat_enabled() {
    static const bool on = [] { const char* e = std::getenv("DISCO_CONV_LAT"); return !(e && e[0] == '0'); }();
    return on;
}

// DISCO_CONV_NJ=0: the stride-2 tile fetches its weight chunks per image, as in rounds 1-4 (A/B runs; bit-identical results)
inline bool conv_nj_enabled() {
    static const bool on = [] { const char* e = std::getenv("DISCO_CONV_NJ"); return !(e && e[0] == '0'); }();
    return on;
}

}  // namespace

// One translation unit per arithmetic instantiates this (conv_mx_ar0/1/2.hip): the three compile in parallel.
template <int AR>
int dispatch_mx_ar(const ConvMxArgs& a, hipStream_t s) {
    const bool wide = a.w_out > 16;
    const bool nt2 = a.c_out > 32;
    if constexpr (AR == 2) {
        if (a.c1_gray) return launch_mx4<32, 16, 2, 1, 8, 1, false, false, 2, true>(a, s);       // (launch_conv3x3_x3 has checked the shape)
    }
    if (a.stride == 1) {
        // Candidates in order of efficiency at full load; the first that fills 3/4 of the CUs is taken, else the one with most workgroups.
        // The last one runs the LATENCY loop (NB = 3; round 5): it is only reached when the launch cannot fill the GPU with the bigger tiles
        // (the latency loop on the 8-consumer-wave tiles was measured and lost to the throughput loop: profiles/r05_latency_loop_ab.txt).
        // Every candidate a layer can take accumulates in the same order (the tap order is tied to the tile WIDTH; a 16-wide tile is only
        // offered to images at most 16 wide), so an image's result does not depend on the batch it is part of.
        struct Cand { int tw, th, nt; };
        static const Cand order[7] = {{32, 16, 2}, {32, 16, 1}, {32, 8, 2}, {16, 16, 2}, {32, 8, 1}, {16, 16, 1}, {32, 4, 1}};
        const bool lat = AR != 1 && !(AR == 0 && a.nsrc > 1) && conv_lat_enabled();
        const long fill = (long)num_cus_mx() * 3 / 4;
        int pick = -1; long best = -1;
        for (int i = 0; i < (lat ? 7 : 6); ++i) {
            const Cand& c = order[i];
            if ((c.nt == 2 && !nt2) || (c.tw == 32 && !wide) || (c.tw == 32 && c.th == 16 && a.h_out <= 8)) continue;
            const long w = (long)cdiv(a.w_out, c.tw) * cdiv(a.h_out, c.th) * cdiv(a.c_out, 32 * c.nt) * a.n;
            if (w >= fill) { pick = i; break; }
            if (w > best) { best = w; pick = i; }
        }
        switch (pick) {
            case 0: return launch_mx2<32, 16, 2, 1, 8, 1, AR>(a, s);
            case 1: return launch_mx2<32, 16, 1, 1, 8, 1, AR>(a, s);
            case 2: return launch_mx2<32, 8, 2, 1, 4, 2, AR>(a, s);
            case 3: return launch_mx2<16, 16, 2, 1, 4, 2, AR>(a, s);
            case 4: return launch_mx2<32, 8, 1, 1, 8, 1, AR>(a, s);
            case 6:
                if constexpr (AR != 1) return launch_mx2<32, 4, 1, 1, 4, 1, AR, 3>(a, s);
                [[fallthrough]];
            default: return launch_mx2<16, 16, 1, 1, 8, 1, AR>(a, s);
        }
    }
    if constexpr (AR == 2 || AR == 3) {
        // the stride-2 tile with four images per weight chunk (NJ = 4; bit-identical): when the images split into whole groups of four without
        // a worse balance over the CUs than the plain launch finds
        if (wide && nt2 && !a.tapmask && a.nsrc == 1 && conv_nj_enabled()) {
            const long combos = (long)cdiv(a.w_out, 32) * cdiv(a.h_out, 4) * cdiv(a.c_out, 64), cus = num_cus_mx();
            const int g1 = mx_groups(combos, a.n, 1), g4 = mx_groups(combos, a.n, 4);
            if (g4 > 0 && cdiv(combos * g4, cus) * cdiv(a.n, g4) <= cdiv(combos * g1, cus) * cdiv(a.n, g1))
                return launch_mx4<32, 4, 2, 2, 4, 2, false, false, AR, false, 2, 4>(a, s);
        }
    }
    if (wide) return nt2 ? launch_mx2<32, 4, 2, 2, 4, 2, AR>(a, s) : launch_mx2<32, 4, 1, 2, 4, 1, AR>(a, s);
    return nt2 ? launch_mx2<16, 8, 2, 2, 4, 2, AR>(a, s) : launch_mx2<16, 8, 1, 2, 4, 1, AR>(a, s);
}

}  // namespace disco
