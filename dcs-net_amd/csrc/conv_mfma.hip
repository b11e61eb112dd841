// conv_mfma.hip — complex convolution as an fp32 MFMA implicit GEMM (gfx950).
//
// The reference runs four real cuDNN convolutions per complex conv (c_network.py:107-112,
// :135-147).  Here ONE real GEMM does the whole complex product through the 2x2 real embedding
//     D[p][(co,re|im)] = sum_{tap,ci} [x_r x_i] . [[w_r  w_i], [-w_i  w_r]]
//   M = output pixels, N = 2*Cout, K = taps * 2*Cin  — exactly 8 real flops per complex MAC.
// Interleaved (re,im) activations make the K axis contiguous: one ds_read_b128 per lane feeds
// four v_mfma_f32_32x32x2_f32 (k order permuted identically in the pre-packed B panel).
//
//   A (activations)  a 128-pixel output tile's haloed input patch, 8 complex channels at a time, is
//                    gathered into LDS once (cat / nearest-upsample / zero-insertion resolved in
//                    the gather; pixel pitch 20 floats keeps ds_read_b128 conflict-free) and reused
//                    by every tap and every output channel
//   B (weights)      pre-packed per (tap, k-group, 32-column tile) as 1 KiB wave-fragments; read
//                    straight from L2 with one coalesced 16-B-per-lane load per four MFMAs,
//                    prefetched one step ahead
//   C                4 waves x (WM x WN) 32x32 fp32 accumulators; columns = channels on the lanes, so
//                    the epilogue adds the bias per lane and stores 128-B row segments
// fp32-input MFMA is exact fp32 (bit-for-bit an fmaf chain), 157 TFLOP/s peak: the same numerics
// as the direct kernel at ~6x its VALU rate.
#include "conv_common.h"
#include "pack_jobs.h"
#include "conv_mfma_args.h"
#include "conv_ring.h"
#include <cstdio>
#include <type_traits>
#include <cstdlib>

#ifndef DCS_X6_GU16
#define DCS_X6_GU16 4
#endif
#ifndef DCS_EXP_LOOP
#define DCS_EXP_LOOP 0      // timing probes of the tap loop (wrong results): 1 no A reads, 2 no B loads, 4 no gather after chunk 0
#endif
#ifndef DCS_MFMA_GATHER_PREFETCH
#define DCS_MFMA_GATHER_PREFETCH 0      // measured: gather phases of a workgroup halve, its MFMA phases stretch by as much (r04 phase stamps)
#endif
#ifndef DCS_MFMA_PROGRESS_PRIO
#define DCS_MFMA_PROGRESS_PRIO 0      // measured: the two workgroups of a CU finish 7 % apart instead of 25 %, the kernel lasts exactly as long (work-conserving)
#endif
#ifndef DCS_MFMA_STAGGER
#define DCS_MFMA_STAGGER 0
#endif
#ifndef DCS_MFMA_COPY_AT_TOP
#define DCS_MFMA_COPY_AT_TOP 1
#endif
#ifndef DCS_MFMA_LDS_BARRIER
#define DCS_MFMA_LDS_BARRIER 1
#endif
#ifndef DCS_MFMA_XCD_REMAP
#define DCS_MFMA_XCD_REMAP 1
#endif
#ifndef DCS_MFMA_SCALAR_B
#define DCS_MFMA_SCALAR_B 1
#endif
#ifndef DCS_MFMA_PINGPONG
#define DCS_MFMA_PINGPONG 1
#endif
#ifndef DCS_MFMA_EARLY_OPERANDS
#define DCS_MFMA_EARLY_OPERANDS 3
#endif
#ifndef DCS_EXP_M16
#define DCS_EXP_M16 0       // timing probes of the 16-column kernel (wrong results): 1 no stores, 2 no MFMA loop, 4 no gather, 8 return behind the table
#endif
#ifndef DCS_X6_GU32
#define DCS_X6_GU32 8       // gather loads in flight per thread at 32-channel chunks of the emulated kernel
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef DCS_FWD_DIAG
#define FDIAG_NOW() ((long long)__builtin_amdgcn_s_memtime())
long long* g_fdbg = nullptr;
#else
#define FDIAG_NOW() 0LL
#endif



// wave grid: WAVES_N waves along N, 4/WAVES_N along M; each wave owns WM x WN tiles of 32x32.
// CH complex input channels are staged per LDS pass (U = CH/4 k-groups per tap): the gather is latency- not
// bandwidth-bound, so a deeper chunk amortises its two barriers and one memory round trip over U/2 x the MFMA work.
// Software pipeline per wave, all register indices static (the tap loop body is unrolled over the U k-groups):
//   B fragments  ring of R = min(U,4) slots, each refilled right after use with the fragment R iterations ahead
//                (across tap and chunk boundaries; a short-tile iteration is ~256 cycles < one L2 round trip)
//   A fragments  two register sets alternating by k-group parity, loaded one iteration ahead
// BF = true: bf16 operands, fp32 accumulate (dcs_set_conv_precision(1); BASELINE configs[4]).  The gather rounds the
// activations to bf16 on their way into LDS (half the patch), the B panel holds bf16 fragments, and one
// v_mfma_f32_32x32x16_bf16 covers what eight fp32 MFMAs cover (8 complex channels of one tap).  Measured in float
// units the fragment addresses are the same as in the fp32 form: a 16-byte read per lane either way.
// PR = 2: fp32 emulated on the bf16 MFMA ("bf16x6").  Every fp32 operand is split EXACTLY into three bf16 terms
// x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1: 8 + 8 + 8 significand bits); of the nine cross
// products of a*b the six with i + j <= 2 are accumulated in fp32 (each bf16 x bf16 product is exact in fp32), smallest
// terms first.  The dropped terms are below 3 * 2^-24 |a||b|, the size of one fp32 rounding of the product; six
// v_mfma_f32_32x32x16_bf16 cover what eight v_mfma_f32_32x32x2_f32 cover at 192 instead of 512 MFMA cycles.  The patch holds
// three bf16 planes per pixel (CH words each), the B panel three planes of bf16 fragments (packjob::MFMA, flag 3).
// TPI > 1 (shallow layers, CH = 8: only U = 2 k-groups per tap): the tap loop advances a whole kernel ROW of TPI = kw taps
// per iteration — 8 MFMAs per iteration cannot carry the loop's scalar address arithmetic, its waitcnt drain and the
// B-set copy (enc1 forward: 7 x 2 = 14 k-groups = 56 MFMAs per iteration instead of 8).
// WK > 1 (layers with few output pixels and a long K — enc5 / enc6 / dec0, the latent fc and their data gradients at B = 32):
// WK waves share ONE output tile and split the K axis between them — wave wk takes the k-groups [wk U / WK, (wk + 1) U / WK)
// of every chunk and tap — and add their accumulators through LDS in the epilogue (fixed order).  The global split-K these
// layers ran before wrote fp32 slabs to HBM and needed a reduce launch each; here the tile is 32 pixels, so there are
// enough workgroups without slicing, the partial sums never leave the CU and the launch goes.
template <int WAVES_N, int WM, int WN, int CH, int PR, int TPI = 1, bool STAT = false, int WK = 1>
__global__ __launch_bounds__(256) void cconv_mfma_kernel(MArgs m) {
    DCS_PRIO_CRITICAL();
    constexpr bool BF = PR != 0;
    constexpr int NP = PR == 2 ? 3 : 1;                                // bf16 planes per operand
    extern __shared__ __attribute__((aligned(16))) float patch[];      // [rows*cols][PIX]
    constexpr int UALL = BF ? CH / 8 : CH / 4;                         // k-groups per tap and chunk
    // ... of ONE wave.  TSP (a chunk has fewer k-groups than WK: enc1's 8 channels = ONE bf16 k-group): the waves split the
    // TAPS instead — wave wk takes taps wk, wk + WK, ... with all k-groups
    constexpr bool TSP = WK > 1 && (UALL % WK) != 0;
    constexpr int U = TSP ? UALL : UALL / WK;
    constexpr int TSTEP = TPI * (TSP ? WK : 1);                         // tap-loop stride
    static_assert(WK == 1 || TPI == 1, "K split over waves: tap-at-a-time loop");
    constexpr int VU = U * TPI, PIX = BF ? NP * CH + 4 : 2 * CH + 4, Q = CH / 2;
    const conv::Args& a = m.c;
    // XCD-aware tile order (Round 4).  The hardware deals workgroups round-robin over the 8 XCDs (workgroup n -> XCD n % 8, each
    // with its own 4 MB L2) and the two slots of a CU get n and n + 256: in plain (x, y, z) order every XCD saw every
    // (column tile, class) pair — the whole three-plane B panel (dec1: 9.4 MB) streamed through each 4 MB L2 — and the two
    // workgroups of a CU read DIFFERENT panels.  Remapped, XCD j owns a contiguous range of the logical order (x fastest): all the
    // pixel tiles of one or two (column tile, class) pairs, i.e. ~1 MB of B that stays in its L2 and is shared by a CU's
    // workgroups in L1.  Speed only: any placement gives the same result.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#if DCS_MFMA_XCD_REMAP
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const unsigned n = (bz * gy + by) * gx + bx, xcd = n & 7u, idx = n >> 3, q = total >> 3, r = total & 7u;
        const unsigned L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        // (the divisions run through the VALU's reciprocal: hand the quotients back to the scalar unit explicitly)
        bx = __builtin_amdgcn_readfirstlane(L % gx);
        by = __builtin_amdgcn_readfirstlane((L / gx) % gy);
        bz = __builtin_amdgcn_readfirstlane(L / (gx * gy));
#if DCS_MFMA_STAGGER
        // experiment: the workgroup in a CU's second slot (hardware ids 256 .. 511, 768 .. ) starts ~one gather late, so that the
        // two co-resident workgroups alternate gather and MFMA phases instead of running them in lockstep
        if ((n >> 8) & 1u) __builtin_amdgcn_s_sleep(DCS_MFMA_STAGGER);
#endif
    }
#endif
    const conv::Cls& k = m.cls[bz];
    // pixels per workgroup = (4 / (WAVES_N WK)) * WM * 32 = TH * TW
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave % WK, wq = wave / WK;
    const int wm = wq / WAVES_N, wn = wq % WAVES_N;
    const int kk = lane >> 5, li = lane & 31;

    const long long d_start = FDIAG_NOW();
    long long d_gather = 0, d_mfma = 0;
    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int b = bx / tiles_per_img, tile_id = bx % tiles_per_img;
    const int oy0 = (tile_id / a.tiles_w) * m.TH, ox0 = (tile_id % a.tiles_w) * m.TW;
    const int ny = m.NT / (WAVES_N * WN);
    const int kslice = by / ny;
    const int nt0 = ((by % ny) * WAVES_N + wn) * WN;           // first 32-column tile of this wave
    // STAT: CBN statistics of the raw output (a.stat, training; unsliced launches only): a row of partial sums per workgroup
    if (oy0 >= k.Hc || ox0 >= k.Wc) {                                  // tile outside this (smaller) class
        if (STAT) {
            float* const stat_row = a.stat + ((long)bz * gridDim.x + bx);
            const int stat_nt0 = (by % ny) * WAVES_N * WN;
            for (int o = t; o < WAVES_N * WN * 80; o += 256) {
                const int e = o % 80, c = ((stat_nt0 + o / 80) * 32 + 4 * (e & 7)) / 2 + e / 40;
                if (c < a.Cout) stat_row[(long)(c * 5 + ((e % 40) >> 3)) * a.stat_stride] = 0.f;
            }
        }
        return;
    }
    const int vy0 = oy0 * a.sf - k.pad_f, vx0 = ox0 * a.st - k.pad_t;
    const int Cin = a.C1 + a.C2;
    const int ntaps = k.kh * k.kw;
    const int cols = (m.TW - 1) * a.st + k.kw, rows = (m.TH - 1) * a.sf + k.kh;

    // LDS float offset of this lane's pixel for each of its m-tiles (tap (0,0), k-group 0)
    int pixoff[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int pi = (wm * WM + i) * 32 + li;
        pixoff[i] = ((((pi >> m.twshift)) * a.sf) * cols + ((pi & (m.TW - 1))) * a.st) * PIX + kk * 4 + (TSP ? 0 : wk * U * 8);   // (+ this wave's k-groups)
    }
    int t0 = TSP ? __builtin_amdgcn_readfirstlane(wk) : 0;             // this wave's first tap (wave-uniform: kept scalar)
    if (TSP) asm volatile("" : "+s"(t0));
    const int kgo = TSP ? 0 : wk * U;                                  // ... first k-group
    int kgo_s = __builtin_amdgcn_readfirstlane(kgo);                   // (wave-uniform: the scalar copy for the B addresses;
    asm volatile("" : "+s"(kgo_s));                                    //  opaque, or the compiler sinks the readfirstlane below the multiplies)
    const float* bbase = m.bm + k.bm_off + ((long)nt0 * 64 + lane) * 4;
    const long b_tap_stride = (long)(BF ? m.KG / 2 : m.KG) * m.NT * 256, b_kg_stride = (long)m.NT * 256;
    const long b_plane_stride = (long)(k.kh * k.kw) * b_tap_stride;    // (PR = 2: planes of this class's panel)

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // (Measured, no effect: delaying the workgroup in the odd wave slot by 1-3 us so that co-resident workgroups stop gathering
    // and multiplying in step — dec1 / dec2 / enc3 within +-1 %: the MFMA phase is bound by operand delivery, not by overlap.)
    const int c_begin = kslice * m.cps;
    const int n_chunks = (c_begin + m.cps) * CH < Cin ? c_begin + m.cps : Cin / CH;     // end of this slice's chunks
    const int nslots = rows * cols * Q;                                // float4 (2 complex) slots of one patch chunk
    const int tapoff0 = ((t0 / k.kw) * cols + (t0 % k.kw)) * PIX;      // LDS offset of this wave's first tap

    // fragment of (chunk c, tap tp, k-group g).  Always a load (past the end it re-reads the last chunk's fragment,
    // which nobody consumes): the tap loop body stays free of branches, so s_waitcnt counts stay exact instead of
    // draining to zero at every loop header.
    // vg: k-group index inside an iteration = (tap offset vg / U, k-group vg % U)
    // (INPLACE, below: uniform panel base in SGPRs + one 32-bit lane offset, so that the 147 loads of the unrolled rows
    // share one address register instead of a 64-bit VGPR pair each — 252 VGPRs before, one wave per SIMD)
    const char* ubase = reinterpret_cast<const char*>(m.bm + k.bm_off) +
                        (size_t)__builtin_amdgcn_readfirstlane(nt0) * 1024;
    const unsigned lane_off = (unsigned)lane * 16u;
    bool bprobe_first = true;
    auto bload = [&](float4 (*dst)[WN], int c, int tp, int vg) {
        if ((DCS_EXP_LOOP & 2) && !bprobe_first) return;
        tp += vg / U;
        const int g = vg % U;
        if (tp >= ntaps) { tp = TSP ? t0 : tp - ntaps; ++c; }
        c = c < n_chunks ? c : n_chunks - 1;
        if (PR == 2 && TPI > 1) {
            unsigned so = (unsigned)((tp * b_tap_stride + (long)(c * UALL + kgo + g) * b_kg_stride) * 4);
            asm volatile("" : "+s"(so));                               // keep the offset where it is used (no hoisting of 49 rows)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    dst[pl][j] = *reinterpret_cast<const float4*>(ubase + so + (unsigned)((pl * b_plane_stride + j * 256) * 4) + lane_off);
            return;
        }
#if DCS_MFMA_SCALAR_B
        // Round 4: the fragment address is wave-uniform up to the lane's 16 bytes, so it is kept on the SCALAR unit (uniform panel
        // base + scalar byte offset + one constant lane-offset VGPR).  As per-lane 64-bit pointer arithmetic every tap cost two
        // v_mad_u64_u32, two v_mul_lo_u32 and ~8 v_lshl_add_u64 (quarter-rate / 64-bit VALU: ~100 issue cycles per tap per wave,
        // in order with the wave's own MFMAs): the probe that removed the B loads gained 16 % of dec1's forward, most of it this.
        {
            const long so = (tp * b_tap_stride + (long)(c * UALL + kgo_s + g) * b_kg_stride) * 4;     // scalar unit throughout
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    typedef __attribute__((address_space(1))) const char gchar_t;       // (the asm would otherwise leave a flat pointer)
                    typedef float f32x4n __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(1))) const f32x4n gfloat4_t;
                    gchar_t* sb = (gchar_t*)(ubase + (so + (pl * b_plane_stride + j * 256) * 4));
                    unsigned lo = lane_off;
                    asm volatile("" : "+s"(sb), "+v"(lo));             // SGPR pair + 32-bit lane offset in THIS block: global_load v, v_lo, s[base]
                    dst[pl][j] = __builtin_bit_cast(float4, *(gfloat4_t*)(sb + lo));
                }
            return;
        }
#endif
        const float* bp = bbase + tp * b_tap_stride + (long)(c * UALL + kgo + g) * b_kg_stride;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < WN; ++j) dst[pl][j] = *reinterpret_cast<const float4*>(bp + pl * b_plane_stride + j * 256);
    };
    // B fragments of one whole tap (U k-groups) live in registers; the NEXT tap's are fetched during the first half of
    // this tap's k-groups (two per group) into a second set and moved over at the end of the tap.  The compiler drains
    // vmcnt to zero at every loop header (it cannot carry partial counts across the back edge): with the loads at the
    // front of the body that drain finds them >= U/2 groups (>= 512 MFMA cycles) old instead of just issued.
    constexpr int LPG = VU >= 2 ? 2 : 1;
    // INPLACE (the emulated 7x7 layer, enc1: PR = 2, a kernel ROW of 7 taps per iteration, one k-group per tap): three
    // planes x 7 taps of B fragments are 84 VGPRs, so there is ONE set; a slot is refilled right after its MFMAs with the
    // same tap of the next kernel row (7 k-groups = 1344 MFMA cycles ahead), and the row loop is fully unrolled (49 taps:
    // the launcher guarantees kh = kw = 7) — straight-line code keeps exact vmcnt counts, a loop header would drain the
    // refill issued just before it.
    constexpr bool INPLACE = PR == 2 && TPI > 1;
    constexpr int NTAPS_C = INPLACE ? TPI * TPI : 0;
    float4 bcur[VU][NP][WN], bnxt[INPLACE ? 1 : VU][NP][WN];
    constexpr bool PP = DCS_MFMA_PINGPONG && !INPLACE && DCS_MFMA_COPY_AT_TOP && DCS_MFMA_EARLY_OPERANDS == 3 && DCS_EXP_LOOP == 0 &&
                        !DCS_MFMA_GATHER_PREFETCH && WM * WN < 4;         // (2 x 2 tiles per wave: the two bodies spill)
    float4 bs2[PP ? 2 : 1][PP ? VU : 1][NP][WN];                      // PP: the two B-fragment sets (tap_body, below)
#pragma unroll
    for (int g = 0; g < VU; ++g) bload(PP ? bs2[0][PP ? g : 0] : (DCS_MFMA_COPY_AT_TOP && !INPLACE ? bnxt[INPLACE ? 0 : g] : bcur[g]), c_begin, t0, g);
    if (DCS_EXP_LOOP & 2) {
#pragma unroll
        for (int g = 0; g < (INPLACE || DCS_MFMA_COPY_AT_TOP ? 0 : VU); ++g)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int j = 0; j < WN; ++j) bnxt[g][pl][j] = bcur[g][pl][j];
        bprobe_first = false;
    }

    // source pixel of every patch pixel, as the element offset of its channel 0 in x1 (spx) and in x2 (spx2) — or -1 for a
    // zero (padding / inserted zero): the same for all channel chunks.  (Round 4: premultiplied by the channel counts here,
    // once per workgroup — as a pixel index every gather slot paid a 64-bit multiply-add, a quarter-rate instruction.)
    const int npatch = rows * cols;
    int* spx = reinterpret_cast<int*>(patch + npatch * PIX);
    int* spx2 = spx + npatch;
    const unsigned cols_magic = 0xFFFFFFFFu / (unsigned)cols + 1u;    // ceil(2^32 / cols): exact quotients for p < 2^16
    for (int p = t; p < npatch; p += 256) {
        const int py = (int)__umulhi((unsigned)p, cols_magic), px = p - py * cols;        // p / cols, p % cols
        long sp;
        const bool in = conv::src_pixel(a, b, vy0 + py, vx0 + px, &sp);
        spx[p] = in ? (int)sp * a.C1 * (int)sizeof(act2_t) : -1;       // BYTE offsets (32-bit: the launcher checks the extents)
        spx2[p] = in ? (int)sp * a.C2 * (int)sizeof(act2_t) : -1;
    }

    const long long d_loop = FDIAG_NOW();
    // gather in rounds of GU independent loads per thread (all in flight together), then the LDS stores.  The source
    // pixel of every patch pixel comes from the table built once above: per slot a shift, an LDS read and one
    // 64-bit multiply-add (the index arithmetic it replaces — four runtime divisions per slot and chunk — kept the
    // VALU busy for ~30 % of a workgroup's life while its MFMA pipe idled).
    constexpr int GU = (PR == 2 && CH == 32) ? DCS_X6_GU32 : (PR == 2 ? DCS_X6_GU16 : 4);
#if DCS_ACT_IS_BF16
    typedef uint2 raw_t;                                               // two complex values = 4 bf16, as stored
#else
    typedef float4 raw_t;
#endif
    // Round 4: the gather was the kernel's main VALU consumer — ~60 vector instructions per 16-byte slot, four of them
    // quarter-rate 64-bit multiply-adds, every load inside its own exec-masked branch, every bf16 conversion a single-operand
    // v_cvt_pk — about 45 % of the MFMA time of a chunk per wave, and on this part VALU issue and MFMA time of the waves of a
    // SIMD ADD.  Now: a thread owns a fixed channel pair (256 % Q == 0), so its source base and table
    // are chosen once per chunk; a slot is a table read, one 64-bit add, an UNCONDITIONAL load (offset clamped, value selected
    // afterwards), the split as packed conversions (dcs_split_pair) and the LDS stores — ~35 full-rate instructions, no branch.
    constexpr int PPR = 256 / Q;                                       // patch pixels covered by one pass of the 256 threads
    const int tq = t % Q, tp0 = t / Q;
    // a chunk lies in ONE source tensor (make_plan: CH divides C1 when there is a second one), so base and table are
    // wave-uniform: global_load v, v_offset, s[base] — no 64-bit vector arithmetic at all
    typedef __attribute__((address_space(1))) const char gsrc_t;
    // the loads of one round (GU slots per thread: patch pixels pb, pb + PPR, ...) of chunk chx
    auto gissue = [&](int chx, int pb, int* o, int* pp, raw_t* v) {
        const bool first = chx * CH < a.C1;
        gsrc_t* xb = first ? (gsrc_t*)a.x1 : (gsrc_t*)a.x2;
        asm volatile("" : "+s"(xb));                                   // (an SGPR pair, or the loads get 64-bit vector addresses)
        const int* tb = first ? spx : spx2;
        const unsigned cb = (unsigned)((first ? chx * CH : chx * CH - a.C1) + 2 * tq) * (unsigned)sizeof(act2_t);   // this thread's channel pair
#pragma unroll
        for (int u = 0; u < GU; ++u) {                                 // (past the end: the last pixel again — same data, same place)
            pp[u] = pb + u * PPR < npatch ? pb + u * PPR : npatch - 1;
            o[u] = tb[pp[u]];
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            unsigned vo = (unsigned)(o[u] < 0 ? 0 : o[u]) + cb;
            asm volatile("" : "+v"(vo));
#if DCS_ACT_IS_BF16
            typedef unsigned u32x2n __attribute__((ext_vector_type(2)));
            v[u] = __builtin_bit_cast(uint2, *(__attribute__((address_space(1))) const u32x2n*)(xb + vo));
#else
            typedef float f32x4g __attribute__((ext_vector_type(4)));
            v[u] = __builtin_bit_cast(float4, *(__attribute__((address_space(1))) const f32x4g*)(xb + vo));
#endif
        }
    };
    // ... and their conversion + LDS stores
    auto gfinish = [&](const int* o, const int* pp, const raw_t* v) {
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            // (an LDS-address-space pointer with a 32-bit offset: as a generic float* this was a 64-bit multiply-add per slot)
            typedef __attribute__((address_space(3))) float lds_f;
            typedef unsigned nu2 __attribute__((ext_vector_type(2)));            // (native vectors: HIP's structs do not assign into LDS)
            typedef float nf2 __attribute__((ext_vector_type(2)));
            typedef float nf4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) nu2 lds_u2;
            typedef __attribute__((address_space(3))) nf2 lds_f2;
            typedef __attribute__((address_space(3))) nf4 lds_f4;
            lds_f* dst = (lds_f*)patch + (__umul24((unsigned)pp[u], (unsigned)PIX) + (unsigned)(tq * (BF ? 2 : 4)));   // (24-bit multiply: full rate)
            const unsigned keep = o[u] < 0 ? 0u : 0xffffffffu;             // (a mask, not a branch around four moves)
#if DCS_ACT_IS_BF16
            // bf16 activations (precision mode 1 only): the stored bits ARE the MFMA operand — no conversion
            static_assert(PR == 1, "bf16 activations run the bf16-operand kernel");
            *(lds_u2*)dst = nu2{v[u].x & keep, v[u].y & keep};
#else
            float4 r = make_float4(__uint_as_float(__float_as_uint(v[u].x) & keep), __uint_as_float(__float_as_uint(v[u].y) & keep),
                                   __uint_as_float(__float_as_uint(v[u].z) & keep), __uint_as_float(__float_as_uint(v[u].w) & keep));
#if defined(DCS_EXP_GATHER) && DCS_EXP_GATHER == 1
            if (PR == 2) {     // timing probe (wrong results): the loaded bits stored as they are — the gather without the split
                *(lds_f2*)dst = nf2{r.x, r.y};
                *(lds_f2*)(dst + CH) = nf2{r.z, r.w};
                *(lds_f2*)(dst + 2 * CH) = nf2{r.x, r.w};
                continue;
            }
#endif
            if (PR == 2) {                                             // 2 complex -> 3 planes of 4 bf16 (exact split)
                nu2 h0, h1, h2;
                h0.x = dcs_split_pair(r.x, r.y); h0.y = dcs_split_pair(r.z, r.w);
                h1.x = dcs_split_pair(r.x, r.y); h1.y = dcs_split_pair(r.z, r.w);
                h2.x = dcs_pack_bf16x2(r.x, r.y); h2.y = dcs_pack_bf16x2(r.z, r.w);
                *(lds_u2*)dst = h0;
                *(lds_u2*)(dst + CH) = h1;
                *(lds_u2*)(dst + 2 * CH) = h2;
            } else if (BF) {                                           // 2 complex -> 4 bf16 (round to nearest even)
                *(lds_u2*)dst = nu2{dcs_pack_bf16x2(r.x, r.y), dcs_pack_bf16x2(r.z, r.w)};
            } else {
                *(lds_f4*)dst = nf4{r.x, r.y, r.z, r.w};
            }
#endif
        }
    };
    // Round 4: the FIRST round of the next chunk's gather is requested during the LAST tap of this chunk's MFMA loop (behind
    // that tap's B-fragment requests: vmcnt retires in order, and nothing in the rest of the tap waits on vector memory), so
    // its memory round trip — the gather is latency-bound: a table read, a load the previous kernel's write-back has pushed
    // out of L2, then the stores — runs under ~24 MFMAs and the chunk barrier instead of after them.  One round = GU slots per
    // thread = the whole patch for the tiles of the train shapes.
    // (32-channel chunks only: their launches hold two workgroups per CU whatever the register count; the GU x 4 held registers
    // would cost the 16-channel instances their third wave per SIMD)
    constexpr bool PREFETCH = DCS_MFMA_GATHER_PREFETCH != 0 && CH == 32 && PR == 2 && !INPLACE;
    int o_pre[GU], pp_pre[GU];
    raw_t v_pre[GU];
    bool have_pre = false;
    // (Measured and dropped: the first round of the NEXT chunk loaded into registers before this chunk's MFMA loop and stored
    // after it — +37 VGPRs take the 32-channel instances from three waves per SIMD to two and the first B-fragment wait of
    // the loop then also waits for the older patch loads: train step 3.950 -> 3.996 ms, inference 3.22 -> 3.37 ms.)
    // The two barriers of a chunk order LDS traffic only, so they wait on the LDS counter only (Round 4).  __syncthreads() also
    // drains vmcnt: the B fragments of the next chunk's first tap, requested during the last tap of this chunk, were waited for
    // right there — one exposed L2 round trip per chunk (dec1: 8 chunks; the probe without B loads ran 16 % faster, and neither
    // earlier requests, scalar addressing nor an XCD-local tile order recovered any of it).  Now they fly under the gather.
#if DCS_MFMA_LDS_BARRIER
#define DCS_CHUNK_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define DCS_CHUNK_BARRIER() __syncthreads()
#endif
    // Progress-based wave priority (Round 4).  Phase stamps per workgroup (tools/fwd_diag.py, DCS_FDIAG_DUMP) showed the two
    // co-resident workgroups of a CU finishing 25 % apart — dec1: 108 k cycles for the one dispatched first, 135 k for the
    // second: the SIMD arbitrates issue by priority, then AGE, so the older workgroup's waves run nearly unimpeded, the younger
    // one gets the leftover slots and then runs its last quarter alone on a pipe one wave per SIMD cannot fill; the kernel
    // lasts as long as the slower one.  A workgroup now lowers its priority as it progresses through its (chunk, tap)
    // iterations (3 -> 0 in quarters): whichever is ahead yields, both finish together.  (The weight-gradient kernel has done
    // this per pixel tile since round 2.)
#if DCS_MFMA_PROGRESS_PRIO
    const int prio_total = (n_chunks - c_begin) * ((ntaps - t0 + TSTEP - 1) / TSTEP);
    const int prio_q1 = prio_total / 4, prio_q2 = prio_total / 2, prio_q3 = prio_total - prio_total / 4;
    int prio_it = 0;
    __builtin_amdgcn_s_setprio(3);
#endif
    for (int ch = c_begin; ch < n_chunks; ++ch) {
        const long long g0 = FDIAG_NOW();
        DCS_CHUNK_BARRIER();                                           // previous chunk fully consumed
#if (defined(DCS_EXP_GATHER) && DCS_EXP_GATHER == 2) || (DCS_EXP_LOOP & 4)
        // timing probe (wrong results): no gather at all after the first chunk — what a perfectly hidden gather would leave
        if (ch == c_begin)
#endif
        {
            int pb = tp0;
            if (PREFETCH && have_pre) {
                gfinish(o_pre, pp_pre, v_pre);
                pb += PPR * GU;
            }
            for (; pb < npatch; pb += PPR * GU) {
                int o[GU], pp[GU];
                raw_t v[GU];
                gissue(ch, pb, o, pp, v);
                gfinish(o, pp, v);
            }
        }
        DCS_CHUNK_BARRIER();
        const long long g1 = FDIAG_NOW();
        d_gather += g1 - g0;
        // (A fragments as NATIVE 8 x bf16 vectors for the bf16 forms: loaded as HIP's float4 struct and bit-cast for the MFMA, the
        // tap body's by-reference capture left a third of the fragment loads as `load <8 x bfloat> ... align 4` — pairs of
        // ds_read2_b32, two instructions and a two-way bank conflict per fragment, instead of one ds_read_b128)
        float4 af[2][BF ? 1 : NP][BF ? 1 : WM];
        bf16x8 afb[2][BF ? NP : 1][BF ? WM : 1];
        typedef __attribute__((address_space(3))) const bf16x8 lds_bf8_t;
        typedef __attribute__((address_space(3))) const float lds_cf_t;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                if constexpr (BF) afb[0][pl][i] = *(lds_bf8_t*)((lds_cf_t*)patch + (pixoff[i] + pl * CH + tapoff0));
                else af[0][pl][i] = *reinterpret_cast<const float4*>(patch + pixoff[i] + pl * CH + tapoff0);
            }
        // the current tap as a patch-PIXEL offset; its LDS float offset is formed (x PIX) inside the tap body: carried as the float
        // offset, the loop's phi hid that it is a multiple of 4 floats and the fragment loads of the first body of a pair were
        // emitted with `align 4` (see afb above)
        int tapq = tapoff0 / PIX;
        // The tap loop's body (Round 4, ping-pong).  P = which of the two B-fragment sets this tap multiplies from; the other one
        // receives the next tap's fragments, k-group by k-group, and the roles swap with the tap — no hand-over copy
        // (VU x NP x WN x 4 moves per tap and wave), and a k-group's fragments are waited for where THAT k-group starts, a
        // whole tap after their request; the copy at the top of a tap waited for all of them at once, i.e. also for the ones
        // requested one k-group earlier (the removal probe DCS_EXP_LOOP & 8: dec1 72 -> 64 us).
        auto tap_body = [&](auto parc, int tap) {
            constexpr int P = decltype(parc)::value;
            float4 (*const cur_)[NP][WN] = PP ? bs2[P] : bcur;
            float4 (*const nxt_)[NP][WN] = PP ? bs2[P ^ 1] : bnxt;
#if DCS_MFMA_PROGRESS_PRIO
            if (prio_it == prio_q1) __builtin_amdgcn_s_setprio(2);
            else if (prio_it == prio_q2) __builtin_amdgcn_s_setprio(1);
            else if (prio_it == prio_q3) __builtin_amdgcn_s_setprio(0);
            ++prio_it;
#endif
#if DCS_MFMA_COPY_AT_TOP
            // Round 4: the prefetched set moves over at the TOP of a tap, not at its end.  At the end, the last tap of a chunk
            // waited (vmcnt) for the next chunk's first fragments — requested half a tap earlier — BEFORE the gather: one exposed
            // L2 round trip per chunk.  Now that request flies under the gather's two barriers and its own memory round trip.
            // (DCS_EXP_LOOP & 8: timing probe, wrong results — the hand-over is skipped behind an opaque test, the loads stay)
            bool do_copy = true;
            if (DCS_EXP_LOOP & 8) { int z = m.KG; asm volatile("" : "+s"(z)); do_copy = z < 0; }
            if (!INPLACE && !PP && do_copy) {
#pragma unroll
                for (int g = 0; g < VU; ++g)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                        for (int j = 0; j < WN; ++j) cur_[g][pl][j] = nxt_[INPLACE ? 0 : g][pl][j];
            }
#endif
            const int tap2 = tap + TSTEP < ntaps ? tap + TSTEP : tap;  // clamped: the last prefetch re-reads this tap
            const int tapq2 = (tap2 / k.kw) * cols + (tap2 % k.kw);
            const int tapoff = tapq * PIX, tapoff2 = tapq2 * PIX;
#pragma unroll
            for (int g = 0; g < VU; ++g) {
                // A fragments of the next iteration into the other register set (taps of a row are adjacent patch columns)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int i = 0; i < WM; ++i) {
                        const int aoff = pixoff[i] + pl * CH + (g + 1 < VU ? tapoff + ((g + 1) / U) * PIX + ((g + 1) % U) * 8 : tapoff2);
                        if constexpr (BF) {
                            if (DCS_EXP_LOOP & 1) { afb[(g + 1) & 1][pl][i] = afb[g & 1][pl][i]; continue; }
                            afb[(g + 1) & 1][pl][i] = *(lds_bf8_t*)((lds_cf_t*)patch + aoff);
                        } else {
                            if (DCS_EXP_LOOP & 1) { af[(g + 1) & 1][pl][i] = af[g & 1][pl][i]; continue; }
                            af[(g + 1) & 1][pl][i] = *reinterpret_cast<const float4*>(patch + aoff);
                        }
                    }
#if DCS_MFMA_EARLY_OPERANDS
                // Round 4: both operand streams are issued AHEAD of this k-group's MFMAs.  Left to the scheduler the LDS reads of
                // the next k-group sank to just before this group's last MFMA (one MFMA = 32 cycles of cover for a ~100-cycle LDS
                // round trip, every k-group), and the next tap's B fragments were requested after the first k-group's MFMAs, i.e.
                // U - 1 k-groups before the copy that waits for them (one k-group, ~400 cycles, on the two-wave K-split tiles).
                if (DCS_MFMA_EARLY_OPERANDS == 1 && !INPLACE && g * LPG < VU) {
#pragma unroll
                    for (int q = 0; q < LPG; ++q)
                        if (g * LPG + q < VU) bload(nxt_[INPLACE ? 0 : g * LPG + q], ch, tap + TSTEP, g * LPG + q);
                }
                if (DCS_MFMA_EARLY_OPERANDS == 3 && !INPLACE) bload(nxt_[INPLACE ? 0 : g], ch, tap + TSTEP, g);   // this group's slot of the next tap
                if (DCS_MFMA_EARLY_OPERANDS != 3) __builtin_amdgcn_sched_barrier(0);
#endif
                // MFMAs straight from the ring slot ...
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        const float4 av = af[g & 1][0][BF ? 0 : i], bv = cur_[g][0][j];
                        if (PR == 2) {                                  // a0 b2, a1 b1, a2 b0, a0 b1, a1 b0, a0 b0
                            constexpr int pa[6] = {0, 1, 2, 0, 1, 0}, pb[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                            for (int e = 0; e < 6; ++e)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                    afb[g & 1][pa[e] < NP ? pa[e] : 0][BF ? i : 0],
                                    __builtin_bit_cast(bf16x8, cur_[g][pb[e] < NP ? pb[e] : 0][j]), acc[i][j], 0, 0, 0);
                        } else if (BF) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afb[g & 1][0][BF ? i : 0],
                                                                                __builtin_bit_cast(bf16x8, bv), acc[i][j], 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i][j], 0, 0, 0);
                        }
                    }
#if DCS_MFMA_EARLY_OPERANDS == 3
                // Mode 3: the next k-group's A reads and this group's share of the next tap's B loads are written AHEAD of the MFMAs
                // in the source (so each is requested a whole k-group / a whole tap before its use) and INTERLEAVED with them by the
                // scheduler — one load behind each MFMA — instead of in one burst in front of them (mode 1: 6 LDS reads + 6 global
                // loads = ~120 issue cycles before the first MFMA of every k-group: a lone wave ran at 54 % of the pipe, 61 % as
                // the compiler orders it by itself: profiles/r04_lone_wave.txt).
                if (!INPLACE) {
                    constexpr int NM = WM * WN * (PR == 2 ? 6 : (BF ? 1 : 4)), NA = NP * WM, NB = NP * WN;
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // one MFMA
                        if (i < NA) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // ... one LDS read
                        else if (i < NA + NB) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); // ... or one global load
                    }
                    if (NM < NA) __builtin_amdgcn_sched_group_barrier(0x100, NA - NM, 0);
                    if (NM < NA + NB) __builtin_amdgcn_sched_group_barrier(0x020, NA + NB - (NM > NA ? NM : NA), 0);
                }
#endif
                // next tap's fragments (scheduling barriers pin the loads here, ahead of the remaining MFMA groups)
                __builtin_amdgcn_sched_barrier(0);
                if (PREFETCH && g == 0 && tap + TSTEP >= ntaps && ch + 1 < n_chunks) {      // last tap of the chunk, behind its first MFMA group
                    gissue(ch + 1, tp0, o_pre, pp_pre, v_pre);
                    have_pre = true;
                }
                if (INPLACE) {
                    bload(cur_[g], ch, tap + TSTEP, g);
                } else if (DCS_MFMA_EARLY_OPERANDS != 1 && DCS_MFMA_EARLY_OPERANDS != 3 && g * LPG < VU) {
#pragma unroll
                    for (int q = 0; q < LPG; ++q)
                        if (g * LPG + q < VU) bload(nxt_[INPLACE ? 0 : g * LPG + q], ch, tap + TSTEP, g * LPG + q);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!INPLACE && !DCS_MFMA_COPY_AT_TOP) {
#pragma unroll
                for (int g = 0; g < VU; ++g)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                        for (int j = 0; j < WN; ++j) cur_[g][pl][j] = nxt_[INPLACE ? 0 : g][pl][j];
            }
            if (VU & 1) {                                               // odd U (bf16, CH = 8): the prefetch landed in set 1
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int i = 0; i < WM; ++i) {
                        if constexpr (BF) afb[0][pl][i] = afb[1][pl][i];
                        else af[0][pl][i] = af[1][pl][i];
                    }
            }
            tapq = tapq2;
        };
        if constexpr (PP) {
            // two taps per trip, straight-line (the roles of the two sets are compile-time): the loop header — where the compiler
            // drains vmcnt, it cannot carry counts across the back edge — comes once per PAIR; an odd tap count ends on a single
            // tap whose prefetch (the next chunk's first tap) landed in set 1 and is handed to set 0, once per chunk
            int tap = t0;
#pragma unroll 1
            for (; tap + TSTEP < ntaps; tap += 2 * TSTEP) {
                tap_body(std::integral_constant<int, 0>{}, tap);
                tap_body(std::integral_constant<int, 1>{}, tap + TSTEP);
            }
            if (tap < ntaps) {
                tap_body(std::integral_constant<int, 0>{}, tap);
#pragma unroll
                for (int g = 0; g < VU; ++g)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                        for (int j = 0; j < WN; ++j) bs2[0][PP ? g : 0][pl][j] = bs2[PP ? 1 : 0][PP ? g : 0][pl][j];
            }
        } else {
#pragma unroll(INPLACE ? TPI : 1)
            for (int tap = t0; tap < (INPLACE ? NTAPS_C : ntaps); tap += TSTEP) tap_body(std::integral_constant<int, 0>{}, tap);
        }
        d_mfma += FDIAG_NOW() - g1;
    }
#if DCS_MFMA_PROGRESS_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    const long long d_epi = FDIAG_NOW();
#ifdef DCS_FWD_DIAG
    auto diag_out = [&]() {
        if (m.dbg && lane == 0 && wave == 0) {
            const long wg = ((long)bz * gridDim.y + by) * gridDim.x + bx;
            if (wg < 16384) {
                long long* d = m.dbg + wg * 8;
                const long long e = FDIAG_NOW();
                d[0] = d_gather; d[1] = d_mfma; d[2] = e - d_epi; d[3] = e - d_start; d[4] = d_loop - d_start; d[5] = d_start;
                d[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
                d[7] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            }
        }
    };
#endif

    // epilogue.  C/D map of a 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) — a lane holds 16 rows of ONE
    // column, so storing from the accumulators is 16 dword stores per tile that each touch two 128-byte rows (the texture
    // addresser of a CU took ~4-5 us to work off its eight waves' stores, with nothing left to overlap it).  Each wave
    // transposes its tile through its own 4.5 KB of the (now idle) patch instead: 16 ds_write_b32, 4 ds_read_b128, and
    // a lane owns 4 consecutive columns (two complex channels) of 4 rows: 4 float4 stores per tile, 8 whole rows each.
    __syncthreads();                                                   // every wave is done reading the patch
    constexpr int TP = 36;                                             // row pitch (floats): 16-byte aligned rows
    float* tsm = patch + wave * 32 * TP;
    const int c4 = lane & 7, r8 = lane >> 3;
    const float* biasf = reinterpret_cast<const float*>(a.bias);
    float sst[STAT ? WN : 1][10];                                      // {S_r, S_i, S_rr, S_ii, S_ri} of this lane's two channels
    if (STAT) {
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 10; ++e) sst[j][e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n0 = (nt0 + j) * 32 + 4 * c4;                        // this lane's first column after the transpose
        // columns >= nsplit belong to the second tensor of a concatenation (nsplit = 2 * C1 is a multiple of 4)
        const bool second = m.ksplit <= 1 && m.y2 != nullptr && n0 >= m.nsplit;      // (a K slice stores raw [pixel][N] tiles)
        const int width = m.ksplit > 1 || m.y2 == nullptr ? m.N : (second ? m.N - m.nsplit : m.nsplit);
        const int col = second ? n0 - m.nsplit : n0;
        float* pb = nullptr;                                           // a K slice: raw fp32 partial tile
        act_t* yb = nullptr;
        if (m.ksplit > 1) pb = m.part + (long)kslice * m.slab_floats + (long)b * a.Hout * a.Wout * m.N;
        else yb = (second ? m.y2 : reinterpret_cast<act_t*>(a.y)) + (long)b * a.Hout * a.Wout * width;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        float q[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) q[e] = 0.f;
        if (m.ksplit <= 1 && n0 < m.N) {
            if (biasf) bv = *reinterpret_cast<const float4*>(biasf + n0);
            // folded eval-mode CBN: out = fma(c_re, re, fma(c_im, im, c_add)) per component, the association order of
            // cbn_apply_kernel, so the folded and the two-kernel forms agree bit for bit
            if (a.coef) {
#pragma unroll
                for (int e = 0; e < 12; ++e) q[e] = a.coef[6 * (n0 >> 1) + e];
            }
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tsm[((r & 3) + 8 * (r >> 2) + 4 * kk) * TP + li] = acc[i][j][r];
            // (a wave's LDS operations complete in order: no barrier between its own writes and reads)
            if (WK > 1) __syncthreads();                               // K split: the tile's other K shares are in their waves' tiles
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                if (WK > 1 && qq / (4 / WK) != wk) continue;          // each of the WK waves finishes 32 / WK rows of the tile
                const int row = r8 + 8 * qq;
                float4 v;
                if (WK > 1) {
                    const float* t0 = patch + (wq * WK) * 32 * TP + row * TP + 4 * c4;
                    v = *reinterpret_cast<const float4*>(t0);
#pragma unroll
                    for (int kq = 1; kq < WK; ++kq) {
                        const float4 u2 = *reinterpret_cast<const float4*>(t0 + kq * 32 * TP);
                        v.x += u2.x; v.y += u2.y; v.z += u2.z; v.w += u2.w;
                    }
                } else {
                    v = *reinterpret_cast<const float4*>(tsm + row * TP + 4 * c4);
                }
                const int pi = (wm * WM + i) * 32 + row;
                const int oy = oy0 + (pi >> m.twshift), ox = ox0 + (pi & (m.TW - 1));
                if (STAT && n0 < m.N && oy < k.Hc && ox < k.Wc) {      // moments of the UN-biased value (pivot = bias)
                    float* q_ = sst[STAT ? j : 0];
                    q_[0] += v.x; q_[1] += v.y;
                    q_[2] = fmaf(v.x, v.x, q_[2]); q_[3] = fmaf(v.y, v.y, q_[3]); q_[4] = fmaf(v.x, v.y, q_[4]);
                    q_[5] += v.z; q_[6] += v.w;
                    q_[7] = fmaf(v.z, v.z, q_[7]); q_[8] = fmaf(v.w, v.w, q_[8]); q_[9] = fmaf(v.z, v.w, q_[9]);
                }
                if (m.ksplit <= 1) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (a.coef) {
                        const float4 u = v;
                        // Scalar FMAs, kept apart by empty asm statements: paired by the compiler into v_pk_fma_f32 with op_sel
                        // (low half = q1 * u.y + q4 taking the HIGH half of the {u.x, u.y} pair) the low half sporadically
                        // came out as q4 alone — 16 lanes (48..63) of one row at a time, ~300 of 4 M outputs, different ones
                        // every run — in the instances built on the bf16 MFMA (PR = 2, 64x64 tile, CH = 16) with other
                        // workgroups' MFMAs in flight on the CU; never seen beside the fp32 MFMA, gone with scalar FMAs.
#if defined(DCS_EXP_EPI) && DCS_EXP_EPI >= 3
                        // 3 = the failing instruction pair replicated by hand: v_pk_add_f32 (bias) directly followed by the
                        // v_pk_fma_f32 whose LOW lane reads the HIGH half of the sum (op_sel:[0,1,0]); 4 = the same with two wait
                        // states between them; 5 = the same pk_add followed by four scalar v_fma_f32 (control).  One asm block
                        // per pair, so the spacing is exactly what is written (the hazard recognizer does not look inside).
                        typedef float f2v __attribute__((ext_vector_type(2)));
                        v = *reinterpret_cast<const float4*>(tsm + row * TP + 4 * c4);      // un-biased again
#define DCS_EPI_PAIR(VX, VY, BX, BY, QA0, QA1, QB0, QB1, QC0, QC1)                                                        \
                        {                                                                                               \
                            f2v uv = {VX, VY}, bb = {BX, BY}, qa = {QA0, QA1}, qb = {QB0, QB1}, qc = {QC0, QC1}, rr;     \
                            if (DCS_EXP_EPI == 3)                                                                       \
                                asm volatile("v_pk_add_f32 %1, %2, %1\n\tv_pk_fma_f32 %0, %3, %1, %5 op_sel:[0,1,0]\n\ts_nop 0\n\t" \
                                             "v_pk_fma_f32 %0, %4, %1, %0 op_sel_hi:[1,0,1]\n\ts_nop 0"                  \
                                             : "=&v"(rr), "+v"(uv) : "v"(bb), "v"(qa), "v"(qb), "v"(qc));               \
                            else if (DCS_EXP_EPI == 4)                                                                  \
                                asm volatile("v_pk_add_f32 %1, %2, %1\n\ts_nop 1\n\tv_pk_fma_f32 %0, %3, %1, %5 op_sel:[0,1,0]\n\ts_nop 0\n\t" \
                                             "v_pk_fma_f32 %0, %4, %1, %0 op_sel_hi:[1,0,1]\n\ts_nop 0"                  \
                                             : "=&v"(rr), "+v"(uv) : "v"(bb), "v"(qa), "v"(qb), "v"(qc));               \
                            else if (DCS_EXP_EPI == 6) {      /* no cross-half selection: operands broadcast by v_mov first */ \
                                f2v ui, ur;                                                                             \
                                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(uv) : "v"(bb));                           \
                                ui[0] = uv[1]; ui[1] = uv[1]; ur[0] = uv[0]; ur[1] = uv[0];                             \
                                asm volatile("v_pk_fma_f32 %0, %1, %2, %3\n\ts_nop 0\n\tv_pk_fma_f32 %0, %4, %5, %0\n\ts_nop 0" \
                                             : "=&v"(rr) : "v"(qa), "v"(ui), "v"(qc), "v"(qb), "v"(ur));                 \
                            } else if (DCS_EXP_EPI == 7) {    /* first pair scalar, second = op_sel_hi:[1,0,1] (HIGH lane reads the LOW half) */ \
                                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(uv) : "v"(bb));                           \
                                rr[0] = fmaf(qa[0], uv[1], qc[0]);                                                       \
                                asm volatile("" : "+v"(rr[0]));                                                          \
                                rr[1] = fmaf(qa[1], uv[1], qc[1]);                                                       \
                                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\ts_nop 0" : "+v"(rr) : "v"(qb), "v"(uv)); \
                            } else if (DCS_EXP_EPI == 8) {    /* only the cross-half form, fed by registers written long before */ \
                                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(uv) : "v"(bb));                           \
                                asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]\n\ts_nop 7" \
                                             : "=&v"(rr) : "v"(qa), "v"(uv), "v"(qc));                                   \
                                rr[0] = fmaf(qb[0], uv[0], rr[0]);                                                       \
                                asm volatile("" : "+v"(rr[0]));                                                          \
                                rr[1] = fmaf(qb[1], uv[0], rr[1]);                                                       \
                                asm volatile("" : "+v"(rr[1]));                                                          \
                            } else {                                                                                    \
                                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(uv) : "v"(bb));                           \
                                rr[0] = fmaf(qb[0], uv[0], fmaf(qa[0], uv[1], qc[0]));                                   \
                                asm volatile("" : "+v"(rr[0]));                                                          \
                                rr[1] = fmaf(qb[1], uv[0], fmaf(qa[1], uv[1], qc[1]));                                   \
                                asm volatile("" : "+v"(rr[1]));                                                          \
                            }                                                                                           \
                            VX = rr[0]; VY = rr[1];                                                                     \
                        }
                        DCS_EPI_PAIR(v.x, v.y, bv.x, bv.y, q[1], q[3], q[0], q[2], q[4], q[5])
                        DCS_EPI_PAIR(v.z, v.w, bv.z, bv.w, q[7], q[9], q[6], q[8], q[10], q[11])
#undef DCS_EPI_PAIR
#elif defined(DCS_EXP_EPI) && DCS_EXP_EPI >= 1
                        // tools/pk_hazard_probe.py builds: 1 = the failing form (the compiler pairs the FMAs into v_pk_fma_f32
                        // with op_sel directly behind the v_pk_add_f32 of the bias), 2 = the same pairing with wait states
                        // between the bias add and the FMAs
                        float4 w = u;
#if DCS_EXP_EPI == 2
                        asm volatile("s_nop 3" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w));
#endif
                        v.x = fmaf(q[0], w.x, fmaf(q[1], w.y, q[4]));
                        v.y = fmaf(q[2], w.x, fmaf(q[3], w.y, q[5]));
                        v.z = fmaf(q[6], w.z, fmaf(q[7], w.w, q[10]));
                        v.w = fmaf(q[8], w.z, fmaf(q[9], w.w, q[11]));
#else
                        v.x = fmaf(q[0], u.x, fmaf(q[1], u.y, q[4]));
                        asm volatile("" : "+v"(v.x));
                        v.y = fmaf(q[2], u.x, fmaf(q[3], u.y, q[5]));
                        asm volatile("" : "+v"(v.y));
                        v.z = fmaf(q[6], u.z, fmaf(q[7], u.w, q[10]));
                        asm volatile("" : "+v"(v.z));
                        v.w = fmaf(q[8], u.z, fmaf(q[9], u.w, q[11]));
#endif
                    }
                    v.x = dcs_act(v.x, a.act); v.y = dcs_act(v.y, a.act); v.z = dcs_act(v.z, a.act); v.w = dcs_act(v.w, a.act);
                }
                if (n0 < m.N && oy < k.Hc && ox < k.Wc) {    // 32-bit offsets inside one image (launcher checks the extent)
                    const int off = ((oy * m.os_f + k.oo_f) * a.Wout + ox * m.os_t + k.oo_t) * width + col;
                    if (m.ksplit > 1) *reinterpret_cast<float4*>(pb + off) = v;
                    else dcs_st4(yb + off, v);
                }
            }
            if (WK > 1) __syncthreads();                               // (the tiles are rewritten by the next (i, j))
        }
    }
    if (STAT) {
        // lanes c4 + 8 r8 hold the same two channels: sum over r8 through the wave's own transpose tile (in-order LDS), then
        // over the waves that share a column tile (different pixel rows wm) through `comb`, one row of partial sums per workgroup
        float* comb = patch + 4 * 32 * TP;                             // [wave][j][80]
        float* const stat_row = a.stat + ((long)bz * gridDim.x + bx);
        const int stat_nt0 = (by % ny) * WAVES_N * WN;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
#pragma unroll
            for (int e = 0; e < 10; ++e) tsm[e * 64 + lane] = sst[STAT ? j : 0][e];
            float r0 = 0.f, r1 = 0.f;
            const int k5 = lane >> 3;
            if (lane < 40) {
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) {
                    r0 += tsm[k5 * 64 + c4 + 8 * q8];
                    r1 += tsm[(5 + k5) * 64 + c4 + 8 * q8];
                }
                comb[(wave * WN + j) * 80 + lane] = r0;
                comb[(wave * WN + j) * 80 + 40 + lane] = r1;
            }
        }
        __syncthreads();
        for (int o = t; o < WAVES_N * WN * 80; o += 256) {
            const int ct = o / 80, e = o % 80, wn_ = ct / WN, j_ = ct % WN;
            float sum = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < 4; ++w_)                              // every wave whose column tiles are these (any wm, any wk)
                if ((w_ / WK) % WAVES_N == wn_) sum += comb[(w_ * WN + j_) * 80 + e];
            const int c = ((stat_nt0 + ct) * 32 + 4 * (e & 7)) / 2 + e / 40;
            if (c < a.Cout) stat_row[(long)(c * 5 + ((e % 40) >> 3)) * a.stat_stride] = sum;
        }
    }
#ifdef DCS_FWD_DIAG
    diag_out();
#endif
}

// ---- N = 16 (Cout = 8) variant ---------------------------------------------------------------------------
// A 16-column GEMM leaves half of every 32x32 MFMA tile empty (dec5 forward, the data gradient of enc1).  Here the
// instruction is v_mfma_f32_16x16x4_f32: a wave owns 32 pixels x 16 columns as two 16x16 tiles, and one ds_read_b128
// per lane (4 consecutive reals of its pixel, lane group g = lane/16 picks which 4 of the 16 reals of an 8-channel
// block) feeds four MFMAs per tile; the B panel is packed to match (packjob::MFMA with N = 16: [tap][kg8][64 lanes][4]).
// Same LDS patch, gather, classes and B ring as the 32-wide kernel; no split-K (these layers have plenty of pixels).
typedef float f32x4v __attribute__((ext_vector_type(4)));

// PR = 2 (fp32 emulated on the bf16 MFMA, as in the 32-column kernel): v_mfma_f32_16x16x32_bf16, a k-group = 16 complex
// channels; the lane's 16-byte LDS read is its 8 consecutive bf16 reals of one of the three planes, the B panel is
// [tap][kg16][plane][64 lanes][8 bf16] (packjob::MFMA, flag 17).  Six MFMAs of 16 cycles per tile and k-group instead
// of sixteen of 32.
// PR = 1: bf16 operands (dcs_set_conv_precision(1); activations stored in bf16 need no conversion at all): the same kernel with
// ONE plane — one MFMA per tile and k-group, B panel [tap][kg16][64 lanes][8 bf16] (packjob::MFMA, flag 18).
// FUSE (Round 5): ONE workgroup computes ALL output classes of its pixel tile from one gathered patch.  The parity classes of a
// conv over an upsampled input (dec5: four classes of 2 x 2 taps) and the residue classes of a strided conv's data gradient
// (enc1: 4 x 4 .. 3 x 3 taps) read almost the same source pixels — as separate workgroups (blockIdx.z) each built its own
// source table, gathered and split its own patch and paid its own prologue / epilogue: dec5's forward was 4096 workgroups of
// 0.8 us of MFMAs each, 16 per CU.  The patch is the union of the classes' windows (origin = the largest padding), a class's
// taps are offset into it by its own padding, and each class keeps its own two accumulator tiles; statistics rows, output
// addressing and results are exactly those of the per-class launches.
template <int CH, int PR = 0, bool FUSE = false>
__global__ __launch_bounds__(256) void cconv_mfma16_kernel(MArgs m) {
    DCS_PRIO_CRITICAL();
    extern __shared__ __attribute__((aligned(16))) float patch[];      // [rows*cols][PIX]
    constexpr int NP = PR == 2 ? 3 : 1;
    // pixel pitch = payload + 8 words: pitch / 4 = 2 (mod 4) keeps the sixteen 16-byte reads of every ds_read_b128 lane group — pixels
    // li of one tile row, 16-byte piece lane >> 4 — on sixteen different bank quads (the + 4 of the 32-column kernel's pitch suits ITS
    // lane map; here it was a 56 % conflict rate: profiles/r05_mfma16_probes.txt)
    constexpr int U8 = PR != 0 ? CH / 16 : CH / 8, PIX = (PR != 0 ? NP * CH : 2 * CH) + 8, Q = CH / 2;
    constexpr int NC = FUSE ? 4 : 1;                                   // accumulator sets (classes per workgroup)
    static_assert(PR == 0 || CH % 16 == 0, "bf16 forms: 16-channel k-groups");
    static_assert(!DCS_ACT_IS_BF16 || PR != 2, "bf16 activations: native or bf16-operand form");
    const conv::Args& a = m.c;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int g4 = lane >> 4, li = lane & 15;
    const int c_first = FUSE ? 0 : (int)blockIdx.z, n_cls = FUSE ? m.ncls : 1;

    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int b = blockIdx.x / tiles_per_img, tile_id = blockIdx.x % tiles_per_img;
    const int oy0 = (tile_id / a.tiles_w) * m.TH, ox0 = (tile_id % a.tiles_w) * m.TW;
    // the window of the workgroup's classes: origin = the largest padding, extent = the farthest tap of any class
    int pmf = 0, pmt = 0, ext_f = 0, ext_t = 0, Hc_max = 0, Wc_max = 0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        if (cc >= n_cls) break;
        const conv::Cls& kc = m.cls[c_first + cc];
        pmf = kc.pad_f > pmf ? kc.pad_f : pmf; pmt = kc.pad_t > pmt ? kc.pad_t : pmt;
        ext_f = kc.kh - kc.pad_f > ext_f ? kc.kh - kc.pad_f : ext_f; ext_t = kc.kw - kc.pad_t > ext_t ? kc.kw - kc.pad_t : ext_t;
        Hc_max = kc.Hc > Hc_max ? kc.Hc : Hc_max; Wc_max = kc.Wc > Wc_max ? kc.Wc : Wc_max;
    }
    if (oy0 >= Hc_max || ox0 >= Wc_max) {                              // tile outside every class
        if (a.stat && t < 40) {
#pragma unroll
            for (int cc = 0; cc < NC; ++cc)
                if (cc < n_cls) a.stat[((long)(c_first + cc) * gridDim.x + blockIdx.x) + (long)t * a.stat_stride] = 0.f;
        }
        return;
    }
    const int vy0 = oy0 * a.sf - pmf, vx0 = ox0 * a.st - pmt;
    const int Cin = a.C1 + a.C2;
    const int cols = (m.TW - 1) * a.st + pmt + ext_t, rows = (m.TH - 1) * a.sf + pmf + ext_f;

    int pixoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pi = wave * 32 + i * 16 + li;
        pixoff[i] = ((((pi >> m.twshift)) * a.sf) * cols + ((pi & (m.TW - 1))) * a.st) * PIX + g4 * 4;
    }
    const long b_tap_stride = PR != 0 ? (long)(Cin / 16) * NP * 256 : (long)(Cin / 8) * 256, b_kg_stride = PR != 0 ? NP * 256 : 256;
    // per class (static indices: the class bodies below are instantiated per class): tap count, kernel width, window offset
    int ntaps_c[NC], kw_c[NC], woff_c[NC];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        const conv::Cls& kc = m.cls[cc < n_cls ? c_first + cc : c_first];
        ntaps_c[cc] = kc.kh * kc.kw; kw_c[cc] = kc.kw;
        woff_c[cc] = (pmf - kc.pad_f) * cols + (pmt - kc.pad_t);          // (in patch pixels)
    }
    // The B fragment addresses are wave-uniform up to the lane's 16 bytes: RUNNING scalar pointers (one per ring slot) + one constant
    // lane-offset VGPR.  Round 5: computed from scratch per load — class / chunk wrap, clamps, 64-bit multiplies, per-lane 64-bit
    // pointers — every tap carried ~35 scalar and ~20 vector instructions of bookkeeping: 3.8 SALU + 3.6 VALU per 16-cycle MFMA
    // over the whole kernel, issue-bound at twice its MFMA time (profiles/r05_mfma16_probes.txt).
    typedef __attribute__((address_space(1))) const char gchar_t;
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const f32x4n gfloat4_t;
    auto uniform_ptr = [](const float* q) -> gchar_t* {               // (wave-uniform by construction: say so to the compiler)
        const unsigned long long u = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return (gchar_t*)(((unsigned long long)hi << 32) | lo);
    };
    gchar_t* const pb0 = uniform_ptr(m.bm + m.cls[c_first].bm_off);
    gchar_t* const pb1 = uniform_ptr(m.bm + m.cls[NC > 1 && n_cls > 1 ? c_first + 1 : c_first].bm_off);
    gchar_t* const pb2 = uniform_ptr(m.bm + m.cls[NC > 2 && n_cls > 2 ? c_first + 2 : c_first].bm_off);
    gchar_t* const pb3 = uniform_ptr(m.bm + m.cls[NC > 3 && n_cls > 3 ? c_first + 3 : c_first].bm_off);
    auto bb = [&](auto iv) -> gchar_t* {
        constexpr int i = decltype(iv)::value;
        if constexpr (i == 0) return pb0;
        else if constexpr (i == 1) return pb1;
        else if constexpr (i == 2) return pb2;
        else return pb3;
    };
    const unsigned lane_off = (unsigned)lane * 16u;
    const unsigned tap_stride_b = (unsigned)b_tap_stride * 4u, kg_stride_b = (unsigned)b_kg_stride * 4u;

    f32x4v acc[NC][2];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) { acc[cc][0] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc[cc][1] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    const int n_chunks = Cin / CH;

    // the fragments of one tap (all k-groups, all planes) from a wave-uniform address
    auto bfetch = [&](float4 (*dst)[NP], gchar_t* src) {
        unsigned lo = lane_off;
        asm volatile("" : "+s"(src), "+v"(lo));                        // SGPR pair + 32-bit lane offset: global_load v, v_lo, s[src] offset:...
#pragma unroll
        for (int g = 0; g < U8; ++g)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                dst[g][pl] = __builtin_bit_cast(float4, *(gfloat4_t*)(src + lo + (g * NP + pl) * 1024));
    };
    // B fragments: a ring of RT taps.  Slot j holds tap j of a group of RT taps of the current class and is refilled, right after
    // its MFMAs, with tap j of the NEXT group — of the next class behind a class's last group, of the next chunk behind the last
    // class — RT - 1 taps of MFMAs ahead of its use; all slot indices static.  (With one k-group per tap the single-slot ring of
    // rounds 2-4 requested a tap's fragments 192 cycles before their use.)
    constexpr int RT = U8 == 1 ? 4 : 2;
    float4 bring[RT][U8][NP];
    // address of (class base, chunk c, tap tp): tp clamped to the class's taps (a slot past them is never consumed)
    auto baddr = [&](gchar_t* base, int nt, int c, int tp) -> gchar_t* {
        tp = tp < nt ? tp : nt - 1;
        c = c < n_chunks ? c : n_chunks - 1;
        return base + ((unsigned)tp * tap_stride_b + (unsigned)(c * U8) * kg_stride_b);
    };
#pragma unroll
    for (int j = 0; j < RT; ++j) bfetch(bring[j], baddr(pb0, ntaps_c[0], 0, j));

    // source pixel of every patch pixel as the BYTE offset of its channel 0 in x1 (spx) / x2 (spx2), -1: a zero — as in the
    // 32-column kernel (Round 5: this kernel still had the round-2 gather — a pixel-index table, a 64-bit multiply-add and a
    // division per slot, every load inside its own branch)
    const int npatch = rows * cols;
    int* spx = reinterpret_cast<int*>(patch + npatch * PIX);
    int* spx2 = spx + npatch;
    const unsigned cols_magic = 0xFFFFFFFFu / (unsigned)cols + 1u;    // ceil(2^32 / cols): exact quotients for p < 2^16
    for (int p = t; p < npatch; p += 256) {
        const int py = (int)__umulhi((unsigned)p, cols_magic), px = p - py * cols;        // p / cols, p % cols
        long sp;
        const bool in = conv::src_pixel(a, b, vy0 + py, vx0 + px, &sp);
        spx[p] = in ? (int)sp * a.C1 * (int)sizeof(act2_t) : -1;
        spx2[p] = in ? (int)sp * a.C2 * (int)sizeof(act2_t) : -1;
    }
    // a thread owns a fixed channel pair (256 % Q == 0) and walks the patch pixels tp0, tp0 + PPR, ...: its source tensor, table and
    // channel offset are fixed per chunk (a 32-channel chunk of dec5 spans BOTH sources of the concatenation: per thread, not
    // per wave); GU unconditional loads in flight per thread (offset clamped, value masked afterwards)
    constexpr int PPR = 256 / Q, GU = PR == 2 ? 8 : 4;
    static_assert(256 % Q == 0, "a thread owns a fixed channel pair");
    const int tq = t % Q, tp0 = t / Q;

    if (DCS_EXP_M16 & 8) { __syncthreads(); return; }
    for (int ch = 0; ch < n_chunks; ++ch) {
        __syncthreads();
        if (!(DCS_EXP_M16 & 4)) {
            const int c = ch * CH + 2 * tq;
            const bool first = c < a.C1;
            const char* const xb = first ? reinterpret_cast<const char*>(a.x1) : reinterpret_cast<const char*>(a.x2);
            const int* const tb = first ? spx : spx2;
            const unsigned cb = (unsigned)(first ? c : c - a.C1) * (unsigned)sizeof(act2_t);
            for (int pb = tp0; pb < npatch; pb += GU * PPR) {
                int o[GU], pp[GU];
                float4 v[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {                          // (past the end: the last pixel again — same data, same place)
                    pp[u] = pb + u * PPR < npatch ? pb + u * PPR : npatch - 1;
                    o[u] = tb[pp[u]];
                }
#pragma unroll
                for (int u = 0; u < GU; ++u)
                    v[u] = dcs_ld4(reinterpret_cast<const act_t*>(xb + ((unsigned)(o[u] < 0 ? 0 : o[u]) + cb)));   // (bf16 activations: widened)
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const unsigned keep = o[u] < 0 ? 0u : 0xffffffffu;
                    float4 r = make_float4(__uint_as_float(__float_as_uint(v[u].x) & keep), __uint_as_float(__float_as_uint(v[u].y) & keep),
                                           __uint_as_float(__float_as_uint(v[u].z) & keep), __uint_as_float(__float_as_uint(v[u].w) & keep));
                    float* const dst = patch + pp[u] * PIX + tq * (PR != 0 ? 2 : 4);
                    if (PR == 1) {                                     // 2 complex -> 4 bf16 (round to nearest even; exact on bf16-stored values)
                        *reinterpret_cast<uint2*>(dst) = make_uint2(dcs_pack_bf16x2(r.x, r.y), dcs_pack_bf16x2(r.z, r.w));
                    } else if (PR == 2) {                              // 2 complex -> 3 planes of 4 bf16 (exact split)
                        uint2 h0, h1, h2;
                        h0.x = dcs_split_pair(r.x, r.y); h0.y = dcs_split_pair(r.z, r.w);
                        h1.x = dcs_split_pair(r.x, r.y); h1.y = dcs_split_pair(r.z, r.w);
                        h2.x = dcs_pack_bf16x2(r.x, r.y); h2.y = dcs_pack_bf16x2(r.z, r.w);
                        *reinterpret_cast<uint2*>(dst) = h0;
                        *reinterpret_cast<uint2*>(dst + CH) = h1;
                        *reinterpret_cast<uint2*>(dst + 2 * CH) = h2;
                    } else {
                        *reinterpret_cast<float4*>(dst) = r;
                    }
                }
            }
        }
        __syncthreads();
        // (one call per class with a COMPILE-TIME class index: as a loop with an early exit the compiler kept the class index in
        // a register and put the per-class arrays in scratch)
        auto run_class = [&](auto ccv) {
            constexpr int cc = decltype(ccv)::value, cn = cc + 1 < NC ? cc + 1 : 0;
            const int ntaps = ntaps_c[cc], kw = kw_c[cc];
            // where the slots reload from: this class's tap j + RT, + RT per group ...
            const bool more = cc + 1 < NC && cc + 1 < n_cls;
            gchar_t* bp[RT];
            gchar_t* nb[RT];                                            // ... and, behind the class's last group, tap j of the next class (chunk)
#pragma unroll
            for (int j = 0; j < RT; ++j) {
                bp[j] = baddr(bb(ccv), ntaps, ch, j + RT);
                nb[j] = more ? baddr(bb(std::integral_constant<int, cn>{}), ntaps_c[cn], ch, j) : baddr(pb0, ntaps_c[0], ch + 1, j);
            }
            // the A fragments of tap t + 1 are requested ahead of tap t's MFMAs (two register sets alternating by tap: RT is even), the
            // tap's patch offset advances incrementally.  (carried as a patch-PIXEL offset, multiplied by the constexpr pitch at the
            // use: carried in floats the loop phi hid that it is a multiple of four and every fragment read became two ds_read2_b32 —
            // the trap of the 32-column kernel's Round 4)
            float4 af[2][U8][2][NP];
            int tpix = woff_c[cc], tx = 0;                              // patch pixel / kernel column of the tap whose fragments are requested next
            auto a_read = [&](int set) {
#pragma unroll
                for (int g = 0; g < U8; ++g)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int pl = 0; pl < (PR != 0 ? NP : 1); ++pl)
                            af[set][g][i][pl] = *reinterpret_cast<const float4*>(patch + pixoff[i] + tpix * PIX + g * 16 + pl * CH);
                const bool eol = tx + 1 == kw;                          // (past the last tap: a valid, unused address)
                tpix += eol ? cols - (kw - 1) : 1;
                tx = eol ? 0 : tx + 1;
            };
            a_read(0);
            for (int tap0 = 0; tap0 < ntaps; tap0 += RT) {
#pragma unroll
                for (int j = 0; j < RT; ++j) {
                    const int tap = tap0 + j;
                    if (tap + 1 < ntaps) a_read((j + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);                  // (the reads stay AHEAD of this tap's MFMAs: left free they sink to their use)
                    if (tap < ntaps) {
#pragma unroll
                        for (int g = 0; g < U8; ++g) {
                            if (PR != 0) {
                                constexpr int pa[6] = {0, 1, 2, 0, 1, 0}, pb[6] = {2, 1, 0, 1, 0, 0};     // smallest terms first
#pragma unroll
                                for (int e = (PR == 2 ? 0 : 5); e < 6; ++e)                                // (PR = 1: the a0 b0 term alone)
#pragma unroll
                                    for (int i = 0; i < 2; ++i)
                                        acc[cc][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                            __builtin_bit_cast(bf16x8, af[j & 1][g][i][pa[e] < NP ? pa[e] : 0]),
                                            __builtin_bit_cast(bf16x8, bring[j][g][pb[e] < NP ? pb[e] : 0]), acc[cc][i], 0, 0, 0);
                            } else {
                                const float4 a0 = af[j & 1][g][0][0], a1 = af[j & 1][g][1][0];
                                const float4 bv = bring[j][g][0];
                                acc[cc][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bv.x, acc[cc][0], 0, 0, 0);
                                acc[cc][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bv.x, acc[cc][1], 0, 0, 0);
                                acc[cc][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bv.y, acc[cc][0], 0, 0, 0);
                                acc[cc][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bv.y, acc[cc][1], 0, 0, 0);
                                acc[cc][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bv.z, acc[cc][0], 0, 0, 0);
                                acc[cc][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bv.z, acc[cc][1], 0, 0, 0);
                                acc[cc][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bv.w, acc[cc][0], 0, 0, 0);
                                acc[cc][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bv.w, acc[cc][1], 0, 0, 0);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // slot j's next occupant: tap + RT of this class, or — behind this class's last group — tap j of the next one
                    bfetch(bring[j], tap + RT < ntaps ? bp[j] : nb[j]);
                    bp[j] += RT * tap_stride_b;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (!(DCS_EXP_M16 & 2)) run_class(std::integral_constant<int, 0>{});
        if constexpr (NC > 1 && !(DCS_EXP_M16 & 2)) {
            if (n_cls > 1) run_class(std::integral_constant<int, 1>{});
            if (n_cls > 2) run_class(std::integral_constant<int, 2>{});
            if (n_cls > 3) run_class(std::integral_constant<int, 3>{});
        }
    }

    // C/D map of 16x16x4: col = lane & 15, rows (lane >> 4) * 4 + r
    const float* biasf = reinterpret_cast<const float*>(a.bias);
    const int n = li;
    const float bv = biasf ? biasf[n] : 0.f;
    const bool second = m.y2 != nullptr && n >= m.nsplit;
    act_t* yf = second ? m.y2 : reinterpret_cast<act_t*>(a.y);
    const int width = m.y2 == nullptr ? 16 : (second ? 16 - m.nsplit : m.nsplit);
    const int col = second ? n - m.nsplit : n;
    act_t* yb = yf + (long)b * a.Hout * a.Wout * width;
    float c_re = 1.f, c_im = 0.f, c_add = 0.f;                        // folded eval-mode CBN (see the 32-column kernel)
    if (a.coef) {
        const float* q = a.coef + 6 * (n >> 1);
        if (n & 1) { c_re = q[2]; c_im = q[3]; c_add = q[5]; } else { c_re = q[0]; c_im = q[1]; c_add = q[4]; }
    }
    // (Measured and dropped, Round 5: the four classes' outputs assembled in an LDS image of the output tile and stored as 16-byte
    // pieces of whole 2-KB rows instead of 8 dword stores per lane and class that each touch four 64-byte half lines — same box:
    // dec5 forward 46.5 -> 49.9 us, enc1 data gradient 54.5 -> 57.5 us: the extra barrier pair, LDS round trip and address
    // arithmetic cost more than the half-line stores, which the L2 merges: profiles/r05_mfma16_probes.txt.)
    auto store_class = [&](auto ccv) {
        constexpr int cc = decltype(ccv)::value;
        const conv::Cls& k = m.cls[c_first + cc];
        float* const stat_row = a.stat ? a.stat + ((long)(c_first + cc) * gridDim.x + blockIdx.x) : nullptr;   // (Cout = 8: 40 sums)
        float s1 = 0.f, s2 = 0.f, sri = 0.f;                           // CBN statistics of the raw output (a.stat)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pi = wave * 32 + i * 16 + g4 * 4 + r;
                const int oy = oy0 + (pi >> m.twshift), ox = ox0 + (pi & (m.TW - 1));
                if (stat_row) {                                        // (uniform) moments of the UN-biased value
                    const float raw = (oy < k.Hc && ox < k.Wc) ? acc[cc][i][r] : 0.f;
                    s1 += raw; s2 = fmaf(raw, raw, s2); sri = fmaf(raw, dcs_dpp_term<0xB1, 0xf>(raw), sri);
                }
                float v = acc[cc][i][r] + bv;
                if (a.coef) {
                    const float pv = dcs_dpp_term<0xB1, 0xf>(v);
                    v = (n & 1) ? fmaf(c_re, pv, fmaf(c_im, v, c_add)) : fmaf(c_re, v, fmaf(c_im, pv, c_add));
                }
#if DCS_EXP_M16 & 1
                if (oy < k.Hc && ox < k.Wc && v == 123456.f)          // (timing probe: no output stores)
#else
                if (oy < k.Hc && ox < k.Wc)
#endif
                    dcs_st1(yb + ((oy * m.os_f + k.oo_f) * a.Wout + ox * m.os_t + k.oo_t) * width + col, dcs_act(v, a.act));
            }
        if (stat_row) {
            // column li of 4 row groups (lanes li + 16 g4) x 4 waves: shuffles over g4, then LDS over the waves
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64); sri += __shfl_xor(sri, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64); sri += __shfl_xor(sri, 32, 64);
            __syncthreads();                                           // every wave is done with the patch (and with the previous class's sums)
            if (lane < 16) { patch[(wave * 16 + li) * 3] = s1; patch[(wave * 16 + li) * 3 + 1] = s2; patch[(wave * 16 + li) * 3 + 2] = sri; }
            __syncthreads();
            if (t < 40) {                                              // channel c: {S_r, S_i, S_rr, S_ii, S_ri}
                const int c = t / 5, e = t % 5;
                const int colx = 2 * c + (e == 1 || e == 3), which = e < 2 ? 0 : (e < 4 ? 1 : 2);
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) sum += patch[(w * 16 + colx) * 3 + which];
                stat_row[(long)t * a.stat_stride] = sum;
            }
        }
    };
    store_class(std::integral_constant<int, 0>{});
    if constexpr (NC > 1) {
        if (n_cls > 1) store_class(std::integral_constant<int, 1>{});
        if (n_cls > 2) store_class(std::integral_constant<int, 2>{});
        if (n_cls > 3) store_class(std::integral_constant<int, 3>{});
    }
}

// y[p][n] = act(sum_s part[s][p][n] + bias[n]); columns >= nsplit of a cat split go to y2.  One float4 per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, long slab_floats,
                                                            const float* __restrict__ bias, act_t* __restrict__ y,
                                                            act_t* __restrict__ y2, int nsplit, int N, int act,
                                                            const float* __restrict__ coef) {
    DCS_PRIO_CRITICAL();
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    if (i4 * 4 >= slab_floats) return;
    float4 v = *reinterpret_cast<const float4*>(part + i4 * 4);
#pragma unroll 4
    for (int s_ = 1; s_ < S; ++s_) {
        const float4 u = *reinterpret_cast<const float4*>(part + (long)s_ * slab_floats + i4 * 4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const long p = (i4 * 4) / N;
    const int n = (int)((i4 * 4) % N);
    if (bias) { v.x += bias[n]; v.y += bias[n + 1]; v.z += bias[n + 2]; v.w += bias[n + 3]; }
    if (coef) {                                             // folded eval-mode CBN: two complex channels per thread
        const float* q = coef + 6 * (n >> 1);
        const float4 u = v;
        v.x = fmaf(q[0], u.x, fmaf(q[1], u.y, q[4])); v.y = fmaf(q[2], u.x, fmaf(q[3], u.y, q[5]));
        v.z = fmaf(q[6], u.z, fmaf(q[7], u.w, q[10])); v.w = fmaf(q[8], u.z, fmaf(q[9], u.w, q[11]));
    }
    v.x = dcs_act(v.x, act); v.y = dcs_act(v.y, act); v.z = dcs_act(v.z, act); v.w = dcs_act(v.w, act);
    if (y2 == nullptr) dcs_st4(y + i4 * 4, v);
    else if (n < nsplit) dcs_st4(y + p * nsplit + n, v);
    else dcs_st4(y2 + p * (N - nsplit) + (n - nsplit), v);
}

// The same slice sum for a layer followed by a training-mode ComplexBatchNorm2d (plain output, no activation): a thread
// keeps ONE column group (two complex channels) and walks pixel rows, so the CBN statistics of the raw output come out of
// the same pass — partial {S_r, S_i, S_rr, S_ii, S_ri} of (y - bias) per workgroup, column blockIdx.x of float[C][5][stride]
// (conv_common.h Args::stat), fixed order, no atomics.  G = N / 4 column groups, rpi = 256 / G pixel rows per pass.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ part, int S, long slab_floats,
                                                                  const float* __restrict__ bias, act_t* __restrict__ y,
                                                                  float* __restrict__ stat, int stat_stride, long P, int N,
                                                                  int G, int rpi) {
    DCS_PRIO_CRITICAL();
    __shared__ float red[256 * 10];
    const int t = threadIdx.x, g = t % G, r0 = t / G;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + 4 * g);
    float s[10];
#pragma unroll
    for (int e = 0; e < 10; ++e) s[e] = 0.f;
    for (long r = (long)blockIdx.x * rpi + r0; r < P; r += (long)gridDim.x * rpi) {
        const long o = r * N + 4 * g;
        float4 v = *reinterpret_cast<const float4*>(part + o);
#pragma unroll 4
        for (int s_ = 1; s_ < S; ++s_) {
            const float4 u = *reinterpret_cast<const float4*>(part + (long)s_ * slab_floats + o);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        s[0] += v.x; s[1] += v.y; s[2] = fmaf(v.x, v.x, s[2]); s[3] = fmaf(v.y, v.y, s[3]); s[4] = fmaf(v.x, v.y, s[4]);
        s[5] += v.z; s[6] += v.w; s[7] = fmaf(v.z, v.z, s[7]); s[8] = fmaf(v.w, v.w, s[8]); s[9] = fmaf(v.z, v.w, s[9]);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        dcs_st4(y + o, v);
    }
#pragma unroll
    for (int e = 0; e < 10; ++e) red[e * 256 + t] = s[e];
    __syncthreads();
    for (int o = t; o < G * 10; o += 256) {
        const int gg = o % G, e = o / G;
        float a = 0.f;
        for (int r = 0; r < rpi; ++r) a += red[e * 256 + r * G + gg];
        stat[(long)((2 * gg + (e >= 5)) * 5 + (e % 5)) * stat_stride + blockIdx.x] = a;
    }
}

// (the B-panel re-layout kernel lives in pack_jobs.hip: packjob::MFMA)

struct Plan { int cand, TH, TW, CH, S, cps; long blocks; int wk; bool ring_on; RingPlan ring; };
thread_local bool g_force_wide_panel = false;

template <int WAVES_N, int WM, int WN, int CH, int PR, int TPI, bool STAT = false, int WK = 1>
int launch_tpi(MArgs& m, long npix, hipStream_t stream) {
    const conv::Args& a = m.c;
    size_t lds = (size_t)npix * ((PR == 2 ? 3 * CH + 4 : PR == 1 ? CH + 4 : 2 * CH + 4) + 2) * sizeof(float);   // patch + the two source-offset tables
    if (lds < 4 * 32 * 36 * sizeof(float)) lds = 4 * 32 * 36 * sizeof(float);        // the epilogue's four transpose tiles
    if (STAT && lds < (4 * 32 * 36 + 4 * WN * 80) * sizeof(float)) lds = (4 * 32 * 36 + 4 * WN * 80) * sizeof(float);   // + the statistics' combine area
#ifdef DCS_FWD_ONE_PER_CU
    if (lds < 84 * 1024) lds = 84 * 1024;                                            // experiment: one workgroup per CU
#endif
    auto fn = cconv_mfma_kernel<WAVES_N, WM, WN, CH, PR, TPI, STAT, WK>;
    if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
    dim3 grid(a.tiles_w * a.tiles_h * a.B, (m.NT / (WAVES_N * WN)) * m.ksplit, m.ncls);
    if (grid.y > 65535) return DCS_ERR_BADARG;
    DCS_LAUNCH(fn, grid, dim3(256), lds, stream, m);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

template <int WAVES_N, int WM, int WN, int CH, int PR, int WK = 1>
int launch_bf(MArgs& m, long npix, hipStream_t stream) {
    constexpr bool BF = PR != 0;
    if constexpr (WK > 1) {                                            // K split over waves: tap-at-a-time loop only
        const bool stat_ = m.c.stat != nullptr && m.ksplit <= 1;
        return stat_ ? launch_tpi<WAVES_N, WM, WN, CH, PR, 1, true, WK>(m, npix, stream)
                     : launch_tpi<WAVES_N, WM, WN, CH, PR, 1, false, WK>(m, npix, stream);
    }
    // whole kernel rows per tap-loop iteration for the shallow layers (one class, 7-wide kernel, 128 x 32 tile): native fp32,
    // and the emulated form of the 7 x 7 layer (fully unrolled over its 7 rows)
    constexpr bool ROWS = (PR == 0 || PR == 2) && CH == 8 && WM == 1 && WN == 1;
    const bool stat = m.c.stat != nullptr && m.ksplit <= 1;           // (a sliced launch leaves the statistics to its reduce kernel)
    // ... and of the 4 x 4 one (dec5's data gradient: the stride-2 fold of the 3 x 3 kernel over 8 channels), 128- or 64-pixel tiles
    constexpr bool ROWS4 = PR == 2 && CH == 8 && WN == 1;
    if (ROWS4 && m.ncls == 1 && m.cls[0].kw == 4 && m.cls[0].kh == 4 && !stat)
        return launch_tpi<WAVES_N, WM, WN, CH, PR, ROWS4 ? 4 : 1>(m, npix, stream);
    if (ROWS && m.ncls == 1 && m.cls[0].kw == 7 && (PR == 2 ? m.cls[0].kh == 7 : (m.cls[0].kh * m.cls[0].kw) % 7 == 0))
        return stat ? launch_tpi<WAVES_N, WM, WN, CH, PR, ROWS ? 7 : 1, true>(m, npix, stream)
                    : launch_tpi<WAVES_N, WM, WN, CH, PR, ROWS ? 7 : 1>(m, npix, stream);
    return stat ? launch_tpi<WAVES_N, WM, WN, CH, PR, 1, true>(m, npix, stream) : launch_tpi<WAVES_N, WM, WN, CH, PR, 1>(m, npix, stream);
}

template <int WAVES_N, int WM, int WN, int CH, int WK = 1>
int launch_ch(MArgs& m, long npix, hipStream_t stream) {
    // (a caller-packed wide panel — the real-valued convs of DR-Net — is always in the fp32 fragment order)
    const int pr = g_force_wide_panel ? 0 : conv::mfma_precision(m.c.C1 + m.c.C2, m.ncls == 1 ? m.cls[0].kh * m.cls[0].kw : 0);
#if DCS_ACT_IS_BF16
    if (pr != 1) return DCS_ERR_BADARG;
    return launch_bf<WAVES_N, WM, WN, CH, 1, WK>(m, npix, stream);
#else
    if (pr == 2) return launch_bf<WAVES_N, WM, WN, CH, 2, WK>(m, npix, stream);
    if (pr == 1) return launch_bf<WAVES_N, WM, WN, CH, 1, WK>(m, npix, stream);
    return launch_bf<WAVES_N, WM, WN, CH, 0, WK>(m, npix, stream);
#endif
}

template <int WAVES_N, int WM, int WN, int WK = 1>
int launch(MArgs& m, const Plan& p, long npix, hipStream_t stream) {
    if constexpr (WK > 1) {                                            // (K split over waves: 16- or 32-channel chunks; 8: over the taps)
        if (p.CH == 32) return launch_ch<WAVES_N, WM, WN, 32, WK>(m, npix, stream);
        if constexpr (WK == 2) {
            if (p.CH == 16) return launch_ch<WAVES_N, WM, WN, 16, WK>(m, npix, stream);
            if (p.CH == 8) return launch_ch<WAVES_N, WM, WN, 8, WK>(m, npix, stream);
        }
        if constexpr (WK == 4 && WM == 2 && WN == 1) {                 // (enc1: the taps over four waves)
            if (p.CH == 8) return launch_ch<WAVES_N, WM, WN, 8, WK>(m, npix, stream);
        }
        return DCS_ERR_BADARG;
    } else {
        switch (p.CH) {
            case 32: return launch_ch<WAVES_N, WM, WN, 32>(m, npix, stream);
            case 16: return launch_ch<WAVES_N, WM, WN, 16>(m, npix, stream);
            default: return launch_ch<WAVES_N, WM, WN, 8>(m, npix, stream);
        }
    }
}

template <int CH, int PR, bool FUSE>
int launch16_f(MArgs& m, long npix, hipStream_t stream) {
    const conv::Args& a = m.c;
    const size_t lds = (size_t)npix * ((PR == 2 ? 3 * CH : PR == 1 ? CH : 2 * CH) + 8 + 2) * sizeof(float);   // patch + the two source-offset tables
    auto fn = cconv_mfma16_kernel<CH, PR, FUSE>;
    if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
    dim3 grid(a.tiles_w * a.tiles_h * a.B, 1, FUSE ? 1 : m.ncls);
    DCS_LAUNCH(fn, grid, dim3(256), lds, stream, m);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// all classes of a tile in one workgroup (cconv_mfma16_kernel, FUSE) where the union window's patch keeps two workgroups on a CU
// (DCS_CLASS_FUSE=0: one workgroup per class, the form of rounds 2-4)
template <int CH, int PR = 0>
int launch16_ch(MArgs& m, long npix, hipStream_t stream) {
    static const int fuse_on = [] { const char* e = getenv("DCS_CLASS_FUSE"); return e ? atoi(e) : 1; }();
    if (fuse_on && m.ncls > 1) {
        const conv::Args& a = m.c;
        int pmf = 0, pmt = 0, ext_f = 0, ext_t = 0;
        for (int c = 0; c < m.ncls; ++c) {
            const conv::Cls& k = m.cls[c];
            pmf = k.pad_f > pmf ? k.pad_f : pmf; pmt = k.pad_t > pmt ? k.pad_t : pmt;
            ext_f = k.kh - k.pad_f > ext_f ? k.kh - k.pad_f : ext_f; ext_t = k.kw - k.pad_t > ext_t ? k.kw - k.pad_t : ext_t;
        }
        const long npix_u = (long)((m.TH - 1) * a.sf + pmf + ext_f) * ((m.TW - 1) * a.st + pmt + ext_t);
        const long bytes = npix_u * ((PR == 2 ? 3 * CH : PR == 1 ? CH : 2 * CH) + 8 + 2) * (long)sizeof(float);
        if (bytes <= 80L * 1024) return launch16_f<CH, PR, true>(m, npix_u, stream);
    }
    return launch16_f<CH, PR, false>(m, npix, stream);
}

struct Cand { int bm, bn, wk; };
// (pixels, columns, waves along K): the last two are the K-split tiles of the few-pixel layers (cconv_mfma_kernel, WK)
// and {64, 32, 2} / {64, 32, 4}: the LDS-bound 7x7 / 8-channel layer (enc1) — half the tile, so three workgroups share a CU instead of one;
// {64, 64, 2}: the un-sliced 64 x 64 launches with each B fragment feeding two MFMA sets
// {64, 64, 4} (Round 4): ONE 64 x 64 tile per workgroup, the k-groups of a 32-channel chunk over its four waves — every A and
// every B fragment feeds two MFMA sets (0.5 KB of operand traffic per MFMA instead of 0.75 KB in the {64, 64, 2} layout)
constexpr Cand kCands[] = {{128, 128, 1}, {128, 64, 1}, {64, 64, 1}, {128, 32, 1}, {32, 64, 2}, {32, 32, 4}, {64, 32, 2}, {64, 64, 2}, {64, 32, 4},
                           {64, 64, 4}};

// tile, chunk depth and K slices for geometry `a` (FULL output extent in Hout/Wout) and its classes
bool make_plan(const conv::Args& a, int ncls, const conv::Cls* cls, Plan* p, long* npix_out) {
    const int Cin = a.C1 + a.C2;
    if (!conv::mfma_ok(Cin, a.Cout) || (a.C1 & 1) || a.Hout <= 0 || a.Wout <= 0 || ncls < 1 || ncls > 4) return false;
    int Hc = 0, Wc = 0, kh = 0, kw = 0;
    for (int c = 0; c < ncls; ++c) {
        Hc = cls[c].Hc > Hc ? cls[c].Hc : Hc; Wc = cls[c].Wc > Wc ? cls[c].Wc : Wc;
        kh = cls[c].kh > kh ? cls[c].kh : kh; kw = cls[c].kw > kw ? cls[c].kw : kw;
    }
    if (Hc <= 0 || Wc <= 0) return false;
    // Round 5: the producer / consumer kernel (conv_ring.hip) where its 128 x 64 tile fits the layer
    p->ring_on = false;
    if (!g_force_wide_panel &&
        DCS_SYM(dcs_conv_ring_plan)(a, ncls, cls, conv::mfma_precision(Cin, ncls == 1 ? cls[0].kh * cls[0].kw : 0), &p->ring)) {
        p->ring_on = true;
        p->cand = 10; p->TH = p->ring.TH; p->TW = p->ring.TW; p->CH = p->ring.CH; p->S = 1; p->cps = Cin / p->ring.CH;
        p->blocks = p->ring.wgs; p->wk = 1;
        *npix_out = p->ring.npix;
        return true;
    }
    const int NT = (2 * a.Cout + 31) / 32;
    // tile shape th x tw = bmp pixels: least padding past the class extent first, then the smallest haloed patch
    auto shape = [&](int bmp, int* th, int* tw) {
        long best_cost = -1;
        for (int h = 2; h <= 16 && h <= bmp / 8; h *= 2) {
            const int w = bmp / h;
            const long padded = (long)((Hc + h - 1) / h) * h * ((Wc + w - 1) / w) * w;
            const long patch = (long)((h - 1) * a.sf + kh) * ((w - 1) * a.st + kw);
            const long cost = padded * 4096 + patch;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; *th = h; *tw = w; }
        }
    };
    auto blocks_of = [&](int i, double* eff) {
        int th, tw;
        shape(kCands[i].bm, &th, &tw);
        const long ty = (Hc + th - 1) / th, tx = (Wc + tw - 1) / tw;
        *eff = (double)Hc * Wc / ((double)ty * th * tx * tw);             // share of the tiles' pixels that exist
        return ty * tx * a.B * (NT / (kCands[i].bn / 32)) * ncls;
    };
    // candidate workgroup tiles (pixels x columns): the largest whose USEFUL workgroups (workgroups x the share of their
    // pixels inside the map) number >= min_blocks — several per CU, so one workgroup's MFMA phase hides another's
    // patch gather ...
    static const long min_blocks = dcs_knob("DCS_MFMA_MIN_BLOCKS", 768L);   // (3 per CU; at 1024 the
    // inference shapes' enc6 / enc3 took the smaller tile: 118 -> 94 us, 224 -> 217 us; the train shapes do not move)
    static const long split_below = dcs_knob("DCS_MFMA_SPLIT_BELOW", 512L);
    int best = -1; long best_blocks = -1; double best_useful = -1;
    for (int i = 0; i < 4; ++i) {
        if (NT % (kCands[i].bn / 32) != 0) continue;
        double eff;
        const long blocks = blocks_of(i, &eff);
        if (blocks * eff >= min_blocks) { best = i; best_blocks = blocks; best_useful = blocks * eff; break; }
        if (blocks * eff > best_useful) { best_useful = blocks * eff; best_blocks = blocks; best = i; }
    }
    if (best < 0) return false;
    // ... and when even the best tile leaves CUs idle (deep layers at small batch: few pixels, long K), slice K
    // instead: the tile with the best (pixel fit x operand reuse), times enough slices for ~4 workgroups per CU
    int want_s = 1;
    p->wk = 1;
    // K split over the waves of a 32-pixel tile first (no slabs, no reduce launch); global slices only where that cannot run
    static const int wk_on = (int)dcs_knob("DCS_MFMA_WK", 1);
    static const long wk_min = dcs_knob("DCS_MFMA_WK_MIN", 384L);
    static const long wk_below = dcs_knob("DCS_MFMA_WK_BELOW", 512L);
    if (best_useful < wk_below && wk_on && !g_force_wide_panel && 2 * a.Cout >= 32) {
        for (int i = 4; i < 6 && p->wk == 1; ++i) {
            if (NT % (kCands[i].bn / 32) != 0) continue;
            if (Cin % (kCands[i].wk == 4 ? 32 : 16) != 0) continue;         // whole k-groups per wave: 32- (16-) channel chunks
            // measured per layer at B = 32 (profiles/r03_*_wk_vs_splitk.txt): the 32 x 64 tile wins where one class fills
            // >= 1.5 workgroups per CU (enc5: 47 -> 38 us incl. the reduce it replaces), the 32 x 32 tile only for 1x1 kernels
            // (the latent fc and its data gradient: 23 -> 12 us) — with a halo, eight column tiles re-gathering one patch lose
            // to the sliced 64 x 64 tile (enc6: 30 -> 32 us)
            if (i == 4 && ncls != 1) continue;
            if (i == 5 && kh * kw != 1) continue;
            double eff;
            const long blocks = blocks_of(i, &eff);
            if (blocks * eff >= wk_min || i == 5) { best = i; best_blocks = blocks; best_useful = blocks * eff; p->wk = kCands[i].wk; }
        }
    }
    // enc1 (8 -> 16 channels, 7x7, stride 2, emulated): a 128-pixel tile's three-plane patch is 87 KB — ONE workgroup per CU,
    // nothing to overlap its gather with; 64 pixels x 32 columns with the taps split over two waves: 49 KB, three per CU
    static const int wk_enc1 = (int)dcs_knob("DCS_MFMA_WK_ENC1", 1);
    if (wk_on && wk_enc1 && p->wk == 1 && !g_force_wide_panel && NT == 1 && Cin == 8 && kh == 7 && kw == 7 && ncls == 1 &&
        conv::mfma_precision(Cin, 49) == 2) {
        double eff;
        const long blocks = blocks_of(6, &eff);
        if (blocks * eff >= min_blocks) { best = 6; best_blocks = blocks; best_useful = blocks * eff; p->wk = 2; }
        // ... or, better, over FOUR waves that each cover all 64 pixels (WM = 2): a B fragment then feeds two MFMA sets — half the
        // L1 traffic of the two-wave form, whose waves fetch every tap's fragment for one 32-pixel half each (61.7 -> 55.6 us at
        // the train shapes, 201.7 -> 173.7 us at the inference shapes)
        static const int enc1_wk4 = (int)dcs_knob("DCS_MFMA_ENC1_WK4", 1);
        if (best == 6 && enc1_wk4) { best = 8; p->wk = 4; }
    }
    if (p->wk == 1 && best_useful < split_below) {
        const double reuse[4] = {1.0, 0.9, 0.8, 0.7};
        double best_score = -1;
        for (int i = 0; i < 4; ++i) {
            if (NT % (kCands[i].bn / 32) != 0) continue;
            double eff;
            const long blocks = blocks_of(i, &eff);
            if (blocks * 8 < split_below / 2) continue;                     // even 8 slices would not fill the chip
            if (eff * reuse[i] > best_score) { best_score = eff * reuse[i]; best = i; best_blocks = blocks; }
        }
        want_s = (int)((min_blocks + best_blocks - 1) / best_blocks);
    }
    // 64 x 64 with the k-groups split over two waves (each wave 64 pixels x 32 columns: every B fragment feeds two MFMA sets,
    // half the L1 traffic of the 2 x 2 wave layout whose four waves each fetch their own)
    // (dec1 / dec2 forward and data gradient 78 -> 74 us, dec3 43 -> 41, enc3 / enc4 data gradients -2; not with the statistics
    // epilogue, whose combine then runs over twice the waves: enc2 / enc3 forward +2 us)
    static const int wk64 = (int)dcs_knob("DCS_MFMA_WK64", 1);
    if (wk64 && best == 2 && p->wk == 1 && want_s == 1 && !a.stat && !g_force_wide_panel && Cin % 16 == 0 &&
        conv::mfma_precision(Cin, ncls == 1 ? kh * kw : 0) != 0) {
        best = 7; p->wk = 2;
        if (wk64 == 2 && Cin % 32 == 0) { best = 9; p->wk = 4; }
    }
    p->cand = best; p->blocks = best_blocks;
    shape(kCands[best].bm, &p->TH, &p->TW);
    const long npix = (long)((p->TH - 1) * a.sf + kh) * ((p->TW - 1) * a.st + kw);
    // chunk depth: 16 channels whenever the patch stays within ~1/3 of a CU's LDS (2-3 workgroups per CU overlap
    // each other's gathers), 32 only for small patches (occupancy matters more than barrier count there)
    static const long cap16 = dcs_knob("DCS_MFMA_LDS_CAP", 56L * 1024);
    const int pr_for_cap = g_force_wide_panel ? 0 : conv::mfma_precision(Cin, ncls == 1 ? cls[0].kh * cls[0].kw : 0);
    // 32-channel chunks (half the gather rounds, twice the patch): for the emulated kernel also up to 56 KB where the launch
    // puts at most ~3 workgroups on a CU anyway, so the larger patch costs no residency (train shapes: step -1.5 %; with
    // every layer allowed, the inference shapes lose 0.8 %)
    static const long cap32e = dcs_knob("DCS_MFMA_LDS_CAP32", 32L * 1024);
    const long wg_est = best_blocks * (want_s > 1 ? want_s : 1);
    // ... and up to 72 KB where at most two do (enc4: two gather rounds instead of four, 40.3 -> 38.2 us)
    static const long cap32w2 = dcs_knob("DCS_MFMA_LDS_CAP32W2", 72L * 1024);
    const long cap32c = (pr_for_cap == 2 && wg_est <= 512 && cap32e < cap32w2) ? cap32w2
                      : (pr_for_cap == 2 && wg_est <= 768 && cap32e < 56L * 1024) ? 56L * 1024 : cap32e;
    // likewise a 16-channel chunk up to 80 KB where at most two workgroups land on a CU (enc2, 16 channels, 5x5 / stride 2:
    // one 75 KB chunk instead of two gather rounds of 40 KB: 33.7 -> 28.0 us; the same tile split over the taps of two
    // waves as for enc1: 38.8 us)
    static const long cap16w = dcs_knob("DCS_MFMA_LDS_CAP16W", 80L * 1024);
    const long cap16c = (wg_est <= 512 && cap16 < cap16w) ? cap16w : cap16;
    // patch words per pixel at chunk depth ch: fp32 2 ch + 4; bf16 ch + 4; three bf16 planes 3 ch + 4
    const int pr = g_force_wide_panel ? 0 : conv::mfma_precision(Cin, ncls == 1 ? cls[0].kh * cls[0].kw : 0);
    auto pixw = [&](int ch) { return (pr == 2 ? 3 * ch : pr == 1 ? ch : 2 * ch) + 4; };
    if (best == 9) {
        if (npix * pixw(32) * 4 > 150L * 1024) return false;
        p->CH = 32;
    } else if (p->wk > 1 && best != 7) {                                       // K split over waves: the deepest chunk that fits (<= 56 KB)
        if (best == 6 || best == 8) p->CH = 8;
        else if (Cin % 32 == 0 && npix * pixw(32) * 4 <= 56L * 1024) p->CH = 32;
        else if (p->wk == 2 && Cin % 16 == 0 && npix * pixw(16) * 4 <= 56L * 1024) p->CH = 16;
        else if (Cin % 32 == 0 && npix * pixw(32) * 4 <= 150L * 1024) p->CH = 32;
        else return false;
    }
    else if (2 * a.Cout == 16 && conv::mfma_precision16(Cin) != 0 && !(Cin % 32 == 0 && npix * pixw(32) * 4 <= cap32c)) {
        if (npix * pixw(16) * 4 > 150 * 1024) return false;            // the bf16 forms of the 16-column kernel have no 8-channel form
        p->CH = 16;
    }
    else if (Cin % 32 == 0 && npix * pixw(32) * 4 <= cap32c) p->CH = 32;
    else if (Cin % 16 == 0 && npix * pixw(16) * 4 <= cap16c) p->CH = 16;
    else if (npix * pixw(8) * 4 <= 150 * 1024) p->CH = 8;
    else return false;
    // a chunk never straddles the two sources of a concatenation (the gather picks ONE source per chunk; the 16-column kernel
    // selects per slot): a shallower chunk where the first source's channel count asks for it (DR-Net's 8 + 8 complex channels)
    if (!(2 * a.Cout == 16 && p->cand == 3 && !g_force_wide_panel)) {
        while (a.C2 > 0 && a.C1 % p->CH != 0 && p->wk == 1 && p->CH > 8) p->CH /= 2;
        if (a.C2 > 0 && a.C1 % p->CH != 0) return false;
    }
    const int n_chunks = Cin / p->CH;
    if (2 * a.Cout == 16 || p->wk > 1) want_s = 1;           // the 16-column kernel does not slice K; nor do the K-split tiles
    int S = want_s < n_chunks ? want_s : n_chunks;
    if (S > 8) S = 8;
    if (S < 1) S = 1;
    p->cps = (n_chunks + S - 1) / S;
    p->S = (n_chunks + p->cps - 1) / p->cps;                 // no empty slice
    *npix_out = npix;
    return true;
}

}  // namespace

int dcs_conv_mfma_pack(const float* wp_direct, float* bm, int Cout, int Cin, int taps, hipStream_t stream) {
    if (!conv::mfma_ok(Cin, Cout)) return DCS_ERR_BADARG;
    packjob::Job j{};
    j.kind = packjob::MFMA;
    j.Cout = Cout; j.Cin = Cin; j.kh = taps;
    if (2 * Cout == 16 && conv::mfma_precision16(Cin) == 2) {            // 16 columns, three bf16 planes: [tap][kg16][plane][64 lanes][8 bf16]
        j.flag = 17;
        j.total = (long)taps * (Cin / 16) * 64;                          // (a thread writes its element of all three planes)
    } else if (2 * Cout == 16 && conv::mfma_precision16(Cin) == 1) {     // 16 columns, bf16 operands: one plane
        j.flag = 18;
        j.total = (long)taps * (Cin / 16) * 64;
    } else if (2 * Cout == 16) {                                         // 16-column layout of cconv_mfma16_kernel (half the region)
        j.flag = 16;
        j.total = (long)taps * (Cin / 8) * 64;
    } else if (conv::mfma_precision(Cin, taps) == 2) {                   // three planes of bf16 fragments (exact split)
        j.flag = 3;
        j.total = (long)taps * (Cin / 8) * ((2 * Cout + 31) / 32) * 64;   // (a thread writes its element of all three planes)
    } else if (conv::mfma_precision(Cin) == 1) {                         // bf16 fragments: [tap][kg8][nt][64 lanes][8 bf16]
        j.flag = 2;
        j.total = (long)taps * (Cin / 8) * ((2 * Cout + 31) / 32) * 64;
    } else {
        j.total = (long)taps * (Cin / 4) * ((2 * Cout + 31) / 32) * 64;  // float4 elements
    }
    j.dst_bytes = j.total * (long)sizeof(float4) * (j.flag == 3 || j.flag == 17 ? 3 : 1);
    j.src0 = wp_direct; j.dst0 = bm;
    return packjob::emit(j, stream);
}

// split-K statistics: workgroups of splitk_reduce_stats_kernel for P pixels of N columns (0: N does not tile 256 threads)
static int splitk_stat_blocks(long P, int N) {
    const int G = N / 4;
    if ((N & 3) || G < 1 || G > 256 || 256 % G) return 0;
    const long passes = (P + 256 / G - 1) / (256 / G);
    long nb = (passes + 3) / 4;
    return (int)(nb < 1 ? 1 : (nb > 512 ? 512 : nb));
}

int dcs_conv_mfma_stat_rows(const conv::Args& a, int ncls, const conv::Cls* cls, bool have_ws) {
    Plan p;
    long npix;
    if (!make_plan(a, ncls, cls, &p, &npix)) return 0;
    const int N = 2 * a.Cout;
    if (p.S > 1 && have_ws && !(N & 3)) return splitk_stat_blocks((long)a.B * a.Hout * a.Wout, N);
    int Hc = 0, Wc = 0;
    for (int c = 0; c < ncls; ++c) { Hc = cls[c].Hc > Hc ? cls[c].Hc : Hc; Wc = cls[c].Wc > Wc ? cls[c].Wc : Wc; }
    return ((Wc + p.TW - 1) / p.TW) * ((Hc + p.TH - 1) / p.TH) * a.B * ncls;
}

// bytes of split-K scratch the launch of (a, classes) would use (0: the layer is not sliced)
long dcs_conv_mfma_workspace_bytes(const conv::Args& a, int ncls, const conv::Cls* cls) {
    Plan p;
    long npix;
    if (!make_plan(a, ncls, cls, &p, &npix) || p.S <= 1) return 0;
    return (long)p.S * a.B * a.Hout * a.Wout * 2 * a.Cout * (long)sizeof(float);
}

// a: geometry with the FULL output extent in Hout/Wout; cls[0..ncls): output-parity classes (class-space
// extent Hc x Wc, sub-kernel size, padding, panel offset); y2/nsplit: optional column split of the output;
// ws / ws_bytes: optional split-K scratch (dcs_conv_mfma_workspace_bytes); too small or NULL: the layer runs unsliced
int dcs_conv_mfma_launch_classes(conv::Args& a, const float* bm, int ncls, const conv::Cls* cls, int os_f, int os_t,
                                 act_t* y2, int nsplit, void* ws, long ws_bytes, hipStream_t stream) {
    Plan p;
    long npix;
    if (!make_plan(a, ncls, cls, &p, &npix)) return DCS_ERR_BADARG;
#if DCS_ACT_IS_BF16
    if (dcs_conv_precision() != 1) return DCS_ERR_BADARG;     // bf16 activations: bf16 weight panels (dcs_set_conv_precision(1))
#endif
    // The source-offset tables hold 32-bit BYTE offsets from the tensor's base: a source of 2 GiB or more (16 channels in fp32:
    // B T >= 131072 at F = 128 — e.g. B = 64 at T = 2048) runs as several launches over sub-batches, each with its own base
    // pointers (samples are independent; the statistics rows of sub-launch s follow those of s - 1).  ADVICE r4: the first form
    // of the byte tables returned DCS_ERR_BADARG here — a capability the pixel-index tables had had.
    {
        const long per_sample = (long)a.Hin * a.Win * (a.C1 > a.C2 ? a.C1 : a.C2) * (long)sizeof(act2_t);
        if ((long)a.B * per_sample >= (1L << 31)) {
            if (per_sample >= (1L << 31) || p.S > 1) return DCS_ERR_BADARG;     // (one sample alone; a sliced plan never has this many pixels)
            const int nb_max = (int)(((1L << 31) - 1) / per_sample);
            int Hc_ = 0, Wc_ = 0;
            for (int c = 0; c < ncls; ++c) { Hc_ = cls[c].Hc > Hc_ ? cls[c].Hc : Hc_; Wc_ = cls[c].Wc > Wc_ ? cls[c].Wc : Wc_; }
            const long rows_per_sample = (long)((Wc_ + p.TW - 1) / p.TW) * ((Hc_ + p.TH - 1) / p.TH) * ncls;
            const long ypix = (long)a.Hout * a.Wout;
            const int w1 = y2 != nullptr ? nsplit : 2 * a.Cout, w2 = 2 * a.Cout - nsplit;
            for (int b0 = 0; b0 < a.B; b0 += nb_max) {
                conv::Args s_ = a;
                s_.B = a.B - b0 < nb_max ? a.B - b0 : nb_max;
                s_.x1 = a.x1 + (long)b0 * a.Hin * a.Win * a.C1;
                if (a.x2) s_.x2 = a.x2 + (long)b0 * a.Hin * a.Win * a.C2;
                s_.y = reinterpret_cast<act2_t*>(reinterpret_cast<act_t*>(a.y) + (long)b0 * ypix * w1);
                act_t* const y2s = y2 != nullptr ? y2 + (long)b0 * ypix * w2 : nullptr;
                if (a.stat) {
                    Plan ps;
                    long nps;
                    if (!make_plan(s_, ncls, cls, &ps, &nps) || ps.TH != p.TH || ps.TW != p.TW || ps.S > 1) return DCS_ERR_BADARG;
                    s_.stat = a.stat + rows_per_sample * b0;
                }
                const int rc = dcs_conv_mfma_launch_classes(s_, bm, ncls, cls, os_f, os_t, y2s, nsplit, nullptr, 0, stream);
                if (rc != DCS_OK) return rc;
            }
            return DCS_OK;
        }
    }
    if ((long)a.Hout * a.Wout * 2 * a.Cout >= (1L << 31)) return DCS_ERR_BADARG;  // 32-bit store offsets inside an image
    const int Cin = a.C1 + a.C2;
    MArgs m;
    m.c = a;
    m.bm = bm;
    m.y2 = y2; m.nsplit = nsplit;
    m.ncls = ncls; m.os_f = os_f; m.os_t = os_t;
    int Hc = 0, Wc = 0;
    for (int c = 0; c < ncls; ++c) {
        m.cls[c] = cls[c];
        Hc = cls[c].Hc > Hc ? cls[c].Hc : Hc;
        Wc = cls[c].Wc > Wc ? cls[c].Wc : Wc;
    }
    m.N = 2 * a.Cout; m.KG = Cin / 4; m.NT = (m.N + 31) / 32;
    m.TH = p.TH; m.TW = p.TW;
    m.twshift = __builtin_ctz((unsigned)p.TW);
    if ((1 << m.twshift) != p.TW) return DCS_ERR_BADARG;
    m.c.tiles_w = (Wc + m.TW - 1) / m.TW;
    m.c.tiles_h = (Hc + m.TH - 1) / m.TH;
    m.slab_floats = (long)a.B * a.Hout * a.Wout * m.N;
#ifdef DCS_FWD_DIAG
    m.dbg = g_fdbg;
#else
    m.dbg = nullptr;
#endif
    if (a.stat && (a.coef || a.act != DCS_ACT_NONE || y2 != nullptr)) return DCS_ERR_BADARG;   // raw plain outputs only
    m.ksplit = p.S; m.cps = p.cps; m.part = (float*)ws;
    if (p.S > 1 && (!ws || ws_bytes < (long)p.S * m.slab_floats * (long)sizeof(float) || (m.N & 3) ||
                    (y2 != nullptr && (nsplit & 3)))) {
        m.ksplit = 1; m.cps = Cin / p.CH; m.part = nullptr;
    }
    static const bool trace = getenv("DCS_MFMA_TRACE") != nullptr;        // diagnostic: the plan of every launch
    if (trace)
        fprintf(stderr, "[mfma] B %d in %dx%d C %d+%d -> %dx%d Cout %d k %dx%d ncls %d | cand %d tile %dx%d CH %d S %d/%d cps %d prec %d coef %d\n",
                a.B, a.Hin, a.Win, a.C1, a.C2, a.Hout, a.Wout, a.Cout, a.kh, a.kw, ncls, p.cand, p.TH, p.TW, p.CH, m.ksplit, p.S,
                m.cps, g_force_wide_panel ? 0 : conv::mfma_precision(Cin, ncls == 1 ? cls[0].kh * cls[0].kw : 0), a.coef != nullptr);
    if (p.ring_on) return DCS_SYM(dcs_conv_ring_launch)(m, p.ring, stream);
    if (m.N == 16 && p.cand == 3 && !g_force_wide_panel) {    // 128 pixels x 16 columns, v_mfma_f32_16x16x4_f32
        const int pr16 = conv::mfma_precision16(Cin);
#if !DCS_ACT_IS_BF16
        if (pr16 == 2) {                                       // (make_plan chose a 16- or 32-channel chunk)
            if (p.CH == 32) return launch16_ch<32, 2>(m, npix, stream);
            if (p.CH == 16) return launch16_ch<16, 2>(m, npix, stream);
            return DCS_ERR_BADARG;
        }
#endif
        if (pr16 == 1) {
            if (p.CH == 32) return launch16_ch<32, 1>(m, npix, stream);
            if (p.CH == 16) return launch16_ch<16, 1>(m, npix, stream);
            return DCS_ERR_BADARG;
        }
        switch (p.CH) {
            case 32: return launch16_ch<32>(m, npix, stream);
            case 16: return launch16_ch<16>(m, npix, stream);
            default: return launch16_ch<8>(m, npix, stream);
        }
    }
    int rc;
    switch (p.cand) {
        case 0: rc = launch<2, 2, 2>(m, p, npix, stream); break;      // 128 x 128
        case 1: rc = launch<2, 2, 1>(m, p, npix, stream); break;      // 128 x 64
        case 2: rc = launch<2, 1, 1>(m, p, npix, stream); break;      //  64 x 64
        case 4: rc = launch<2, 1, 1, 2>(m, p, npix, stream); break;   //  32 x 64, two waves along K
        case 5: rc = launch<1, 1, 1, 4>(m, p, npix, stream); break;   //  32 x 32, four waves along K
        case 6: rc = launch<1, 1, 1, 2>(m, p, npix, stream); break;   //  64 x 32, two waves along K (taps)
        case 7: rc = launch<2, 2, 1, 2>(m, p, npix, stream); break;   //  64 x 64, two waves along K
        case 8: rc = launch<1, 2, 1, 4>(m, p, npix, stream); break;   //  64 x 32, four waves along K (taps), 64 pixels each
        case 9: rc = launch<1, 2, 2, 4>(m, p, npix, stream); break;   //  64 x 64, four waves along K, one tile
        default: rc = launch<1, 1, 1>(m, p, npix, stream); break;     // 128 x 32
    }
    if (rc != DCS_OK || m.ksplit <= 1) return rc;
    if (a.stat) {                                                      // slice sum + CBN statistics of the raw output
        const long P = (long)a.B * a.Hout * a.Wout;
        const int nb = splitk_stat_blocks(P, m.N);
        if (nb < 1 || y2 != nullptr || a.coef || a.act != DCS_ACT_NONE) return DCS_ERR_BADARG;
        DCS_LAUNCH(splitk_reduce_stats_kernel, dim3(nb), dim3(256), 0, stream, (const float*)m.part, m.ksplit, m.slab_floats,
                   (const float*)a.bias, (act_t*)a.y, a.stat, a.stat_stride, P, m.N, m.N / 4, 256 / (m.N / 4));
        DCS_CHECK_LAUNCH();
        return DCS_OK;
    }
    const long n4 = m.slab_floats / 4;
    DCS_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const float*)m.part,
                       m.ksplit, m.slab_floats, (const float*)a.bias, (act_t*)a.y, y2, nsplit, m.N, a.act, a.coef);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// single class: the plain convolution described by `a` (Hout/Wout set)
static conv::Cls plain_class(const conv::Args& a) {
    conv::Cls c;
    c.kh = a.kh; c.kw = a.kw; c.pad_f = a.pad_f; c.pad_t = a.pad_t; c.oo_f = 0; c.oo_t = 0;
    c.Hc = a.Hout; c.Wc = a.Wout; c.bm_off = 0;
    return c;
}

long dcs_conv_mfma_workspace_bytes_plain(const conv::Args& a) {
    const conv::Cls c = plain_class(a);
    return dcs_conv_mfma_workspace_bytes(a, 1, &c);
}

// the caller's B panel is in the 32-column fragment order even where N = 16 (real-weight panels: dcs_rconv2d_fwd)
int dcs_conv_mfma_launch_wide(conv::Args& a, const float* bm, void* ws, long ws_bytes, hipStream_t stream) {
    const conv::Cls c = plain_class(a);
    g_force_wide_panel = true;
    const int rc = dcs_conv_mfma_launch_classes(a, bm, 1, &c, 1, 1, nullptr, 0, ws, ws_bytes, stream);
    g_force_wide_panel = false;
    return rc;
}

// columns >= nsplit of the output go to y2 (g_x1 | g_x2 of a concatenation)
int dcs_conv_mfma_launch_split(conv::Args& a, const float* bm, act_t* y2, int nsplit, void* ws, long ws_bytes,
                               hipStream_t stream) {
    const conv::Cls c = plain_class(a);
    return dcs_conv_mfma_launch_classes(a, bm, 1, &c, 1, 1, y2, nsplit, ws, ws_bytes, stream);
}

int dcs_conv_mfma_launch(conv::Args& a, const float* bm, void* ws, long ws_bytes, hipStream_t stream) {
    const conv::Cls c = plain_class(a);
    return dcs_conv_mfma_launch_classes(a, bm, 1, &c, 1, 1, nullptr, 0, ws, ws_bytes, stream);
}

#if defined(DCS_FWD_DIAG) && !defined(DCS_ACT_BF16)
extern "C" int dcs_debug_set_fwd_buffer(void* p) { g_fdbg = (long long*)p; return 0; }
#endif
