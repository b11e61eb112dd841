// conv_wgrad_mfma.hip — weight gradient of the complex convolution as an fp32 MFMA GEMM whose K axis is
// the PIXEL axis:
//     D_tap[(co,re|im)][(ci,re|im)] = sum_p  gY[p][(co,.)] * X[p*s - pad + tap][(ci,.)]      (real, 8 flop / complex MAC)
//     g_W[tap][ci][co] = (D_rr + D_ii) + j (D_ir - D_ri)                                   ( = sum_p gY conj(X) )
// v_mfma_f32_16x16x4_f32: 16 (co,re|im) rows x 16 (ci,re|im) columns x 4 pixels per instruction.
//   * the haloed input patch of a 128-pixel tile (8 complex channels = one 16-column tile) is gathered
//     into LDS once — cat / nearest-upsample resolved there, as in the forward kernel — and re-read by
//     every tap with a shifted pixel address (ds_read_b32, conflict-free 80-B pixel pitch);
//   * each wave owns MT row tiles (8 output channels each) and ALL taps: taps*MT*4 accumulator VGPRs
//     (k=3: 144, k=5: 200, k=7: 196) that persist across the workgroup's pixel tiles, so a tile costs no
//     epilogue; gY fragments come straight from L2 (each element is used by one wave only);
//   * workgroups stride over pixel tiles; one partial slab per workgroup row, reduced without atomics by
//     cconv_wgrad_reduce_kernel (conv_direct.hip), which also writes the reference's parameter layout.
#include "conv_common.h"
#include "wgrad_reduce.h"
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CHUNK = 8;       // complex input channels per workgroup column tile (16 real columns)
constexpr int PIX = 20;        // LDS floats per patch pixel
constexpr int BMP = 128;       // pixels per tile

struct WArgs {
    conv::Args c;
    const act_t* gy; float2* slab_w; float* slab_b;            // (activation type: dcs_common.h)
    int n_slabs, total_tiles, co_blocks, TH, TW, twshift;
    unsigned cols_magic;       // ceil(2^32 / c.cols): patch pixel index / cols by one multiply-high (patches are < 2^16 pixels)
    // output-parity classes of an upsample-folded conv (blockIdx.z): class pixel (oy, ox) is g_Y pixel
    // (oy*os_f + oo_f, ox*os_t + oo_t) and reads the SOURCE-resolution input with its own padding; one class = the
    // plain convolution (os = 1, oo = 0, pad = c.pad).  c.Hout / c.Wout: class-space extent (tiling);
    // Hy / Wy: full g_Y extent (addressing).
    int ncls, os_f, os_t, Hy, Wy;
    int pad_f[4], pad_t[4], oo_f[4], oo_t[4];
    long long* dbg;            // diagnostic builds only (-DDCS_WGRAD_DIAG): per-wave phase times (s_memtime ticks)
    // Round 4: g_Y pre-split into its three bf16 planes in MFMA A-fragment order by gy_planes_kernel (below), or nullptr:
    // uint4[cls][tile][k-step][plane][2 Cout / 16][64 lanes] — a lane's 16 bytes are its 8 pixels of one (co, re|im) row
    const uint4* gy_planes;
};

#ifndef DCS_WG_SETPRIO
#define DCS_WG_SETPRIO 0      // progress-based wave priority (3 -> 0 over a workgroup's tiles): off since the kernel runs as the
                             // BACKGROUND of the backward pass (dcs_common.h, DCS_PRIO_CRITICAL); standalone it measured neutral
#endif
#ifndef DCS_WG_RING_BIG
#define DCS_WG_RING_BIG 2
#endif
#ifndef DCS_WG_RING_SMALL
#define DCS_WG_RING_SMALL 4
#endif
#ifdef DCS_WGRAD_DIAG
#define WDIAG_NOW() ((long long)__builtin_amdgcn_s_memtime())
long long* g_wdbg = nullptr;
#else
#define WDIAG_NOW() 0LL
#endif

// MT row tiles (8 output channels each) per wave.  Plain form (TS = false): WS waves split the tile's PIXELS
// (k-steps) and 4/WS waves split the output channels, so layers with few output channels still fill all four
// SIMDs: the pixel partials are combined through LDS once, after the last tile.  Tap-split form (TS = true: the
// 7x7 layer with 16 output channels): the four waves split the TAPS (13 each) and every wave covers all MT*8 output
// channels of the block and all pixels — 104 accumulator registers per lane instead of 196, so two workgroups fit a CU
// and overlap each other's gathers.
// Two workgroups per CU (<= 256 VGPR + AGPR per lane) whenever the accumulator set allows it.
//
// The k-step loop of a tile is FULLY unrolled: the g_Y fragments of the next D k-steps are in flight in a register ring
// whose slots are compile-time indices, every load is unconditional (clamped address, value masked when consumed), so
// the compiler's s_waitcnt vmcnt(N) counts are exact.  As a rolled loop with predicated loads it drained vmcnt(0)
// at every back edge — each trip waited out the L2 round trip of the loads it had just issued.
template <int KH, int KW, int MT, int WS, bool TS>
__global__ __launch_bounds__(256, (MT * (TS ? (KH * KW + 3) / 4 : KH * KW) * 4 <= 160 && !(KH == 5 && WS == 4) ? 2 : 1))
void cconv_wgrad_mfma_kernel(WArgs w) {
    constexpr int TAPS = KH * KW;
    constexpr int TPW = TS ? (TAPS + 3) / 4 : TAPS;     // taps per wave
    constexpr int PS = TS ? 1 : WS;                     // waves sharing a tile's pixels
    constexpr int WCO = TS ? 1 : 4 / WS;                // waves sharing the block's output channels
    constexpr bool GL = TS || (MT == 1 && WS >= 2);     // g_Y tile through LDS (<= 16 output channels per workgroup)
    constexpr int GLW = TS ? MT : WCO;                  // 8-channel groups per pixel of that block
    constexpr int GQ = GLW * 4;                         // float4 per pixel of the g_Y block
    constexpr int GP = GLW * 16 + (GLW == 1 ? 0 : 16);  // float pitch per pixel: pixel stride == 16 banks (mod 32)
    constexpr int KSW = BMP / 4 / PS;                   // k-steps per wave of a full tile
    constexpr int KCYC = MT * TPW * 32;                 // MFMA cycles of one k-step
    constexpr int D0 = (3200 + KCYC - 1) / KCYC;        // ring depth: >= ~1.3 us of MFMAs between a load and its use
    constexpr int D = GL ? 2 : (D0 < 2 ? 2 : (D0 > 8 ? 8 : D0));
    extern __shared__ __attribute__((aligned(16))) float patch[];      // [rows*cols][PIX]
    const conv::Args& a = w.c;
    const int cls = blockIdx.z;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int ci0 = (blockIdx.y / w.co_blocks) * CHUNK;
    const int part = TS ? 0 : wave % WS;                                // which share of the pixels
    const int cb0 = (blockIdx.y % w.co_blocks) * (GLW * 8);             // first output channel of the block (GL)
    const int co0 = TS ? cb0 : (blockIdx.y % w.co_blocks) * (WCO * MT * 8) + (wave / WS) * (MT * 8);   // first output channel
    const int cooff = TS ? 0 : (wave / WS) * 16;                        // this wave's float offset inside a g_Y block pixel
    const int tap0 = TS ? wave * TPW : 0;
    const int N1 = 2 * a.Cout;                                          // floats per gY pixel
    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int npix = a.rows * a.cols;
    const int npx = w.TH * w.TW;                                        // 128; 64 on maps too small to fill 128
    const int pad_f = w.pad_f[cls], pad_t = w.pad_t[cls], oo_f = w.oo_f[cls], oo_t = w.oo_t[cls];
    const int rowoff = a.cols * PIX;

    f32x4 acc[MT][TPW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) acc[i][tp] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) bsum[i] = 0.f;

    // column (row of D) this lane feeds for each of its row tiles; masked beyond Cout
    bool colok[MT];
    int gcol[MT], gcl[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        gcol[i] = 2 * (co0 + i * 8) + li;
        colok[i] = (co0 + i * 8 + (li >> 1)) < a.Cout;
        gcl[i] = colok[i] ? gcol[i] : 0;
    }
    // LDS float offset of each of this wave's taps (tap-split: taps past the last one repeat it and are dropped at the end)
    int toff[TPW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
        const int tap = tap0 + tp < TAPS ? tap0 + tp : TAPS - 1;
        toff[tp] = TS ? (tap / KW) * rowoff + (tap % KW) * PIX : (tp / KW) * rowoff + (tp % KW) * PIX;
    }

    long long d_gather = 0, d_mfma = 0, d_tiles = 0;
    const long long d_start = WDIAG_NOW();
    int tiles_done = 0;
    const int my_tiles = (w.total_tiles - (int)blockIdx.x + w.n_slabs - 1) / w.n_slabs;      // >= 1: n_slabs <= total_tiles
    for (int tl = blockIdx.x; tl < w.total_tiles; tl += w.n_slabs, ++tiles_done) {
        const long long s0 = WDIAG_NOW();
        // Co-resident workgroups do equal work; the SIMD's oldest-first arbitration lets the first one run ahead and the
        // last one finish alone on a pipe one wave cannot fill (dec2, four per CU: lifetimes 76 / 95 / 111 / 124 us).  A
        // workgroup that is ahead yields: its priority falls with the share of its tiles it has finished (100 - 118 us).
        {
            const int q = tiles_done * 4 / my_tiles;
#if DCS_WG_SETPRIO
            if (q == 0) __builtin_amdgcn_s_setprio(3);
            else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#else
            (void)q;
#endif
        }
        const int b = tl / tiles_per_img, tile_id = tl % tiles_per_img;
        const int oy0 = (tile_id / a.tiles_w) * w.TH, ox0 = (tile_id % a.tiles_w) * w.TW;
        const int vy0 = oy0 * a.sf - pad_f, vx0 = ox0 * a.st - pad_t;
        __syncthreads();
        // one thread per patch pixel: its 8 complex channels are 64 contiguous bytes (4 x 16-byte loads), so the index
        // arithmetic (two runtime divisions + the upsample / zero-insertion mapping) runs once per pixel, not per load
        for (int px = t; px < npix; px += 256) {
            const int ix = px % a.cols, iy = px / a.cols;
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            long sp;
            if (conv::src_pixel(a, b, vy0 + iy, vx0 + ix, &sp)) {
                const act2_t* src = (ci0 < a.C1) ? a.x1 + sp * a.C1 + ci0 : a.x2 + sp * a.C2 + (ci0 - a.C1);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = dcs_ld4(reinterpret_cast<const act_t*>(src) + 4 * q);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(patch + px * PIX + q * 4) = v[q];
        }
        float* gt = patch + npix * PIX;
        if (GL) {
            // few output channels (<= 16 per workgroup): a k-step is only TAPS*32 cycles of MFMA, far less than the L2 round trip
            // of its g_Y fragment — stage the tile's g_Y block [pixels][co_per_block*2] in LDS with coalesced
            // 16-byte loads instead (zeros outside the map / beyond Cout), and feed the A fragments from there
            for (int idx = t; idx < npx * GQ; idx += 256) {
                const int q = idx % GQ, p = idx / GQ;
                const int oy = oy0 + (p >> w.twshift), ox = ox0 + (p & (w.TW - 1));
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (oy < a.Hout && ox < a.Wout && 2 * cb0 + 4 * q < N1)
                    v = dcs_ld4(w.gy + (((long)b * w.Hy + oy * w.os_f + oo_f) * w.Wy + ox * w.os_t + oo_t) * N1 + 2 * cb0 + 4 * q);
                *reinterpret_cast<float4*>(gt + p * GP + q * 4) = v;
            }
        }
        __syncthreads();
        const long long s1 = WDIAG_NOW();
        d_gather += s1 - s0;
        // A k-step's four pixels are ks*4 + lk, lk = lane / 16 < 4 <= TW: row and first column of a k-step are
        // wave-uniform (scalar registers), the lane adds lk.  tws / twm are re-read opaquely per tile so that the 32
        // unrolled k-steps' scalars are computed where they are used instead of being hoisted out of the tile loop (as
        // loop invariants they held ~100 registers).
        int tws = w.twshift, twm = w.TW - 1;
        asm volatile("" : "+s"(tws), "+s"(twm));
        const act_t* gyb = w.gy + (long)b * w.Hy * w.Wy * N1;
        const int xlane = lk * a.st * PIX + li;                        // lane part of a patch address (floats)
        const int glane = lk * w.os_t * N1;                            // lane part of a g_Y pixel offset (floats)
        float afr[D][MT];
        auto load_g = [&](int ks, float* dst) {
            const int spy = (ks * 4) >> tws, spx = (ks * 4) & twm;     // scalar
            if (GL) {
                const float* g = gt + (ks * 4 + lk) * GP + cooff + li;
#pragma unroll
                for (int i = 0; i < MT; ++i) dst[i] = g[i * 16];
                return;
            }
            const int oy = oy0 + spy, ox = ox0 + spx;                  // scalar; this lane's column is ox + lk
            const bool inb = oy < a.Hout && ox + lk < a.Wout;
            const int soff = ((oy * w.os_f + oo_f) * w.Wy + ox * w.os_t + oo_t) * N1;       // < 2^31 floats inside one image
            const act_t* gp = gyb + (inb ? soff + glane : 0);
#pragma unroll
            for (int i = 0; i < MT; ++i) dst[i] = dcs_ld1(gp + gcl[i]);          // always a load; masked when consumed
        };
        auto xp_of = [&](int ks) -> const float* {
            const int spy = (ks * 4) >> tws, spx = (ks * 4) & twm;     // scalar
            return patch + ((spy * a.sf) * a.cols + spx * a.st) * PIX + xlane;
        };
        const int ks_last = npx / 4 - 1;
#pragma unroll
        for (int r = 0; r < D; ++r) load_g(part + r * PS, afr[r]);
        // the first PFT taps' values are read one k-step ahead (an LDS round trip is 3-4 MFMAs long), the rest at the top
        // of their own k-step; a scheduling barrier per k-step keeps the unrolled code from hoisting later k-steps' reads
        constexpr int PFT = TPW < 4 ? TPW : 4;
        float bq[PFT];
        {
            const float* xp = xp_of(part);
#pragma unroll
            for (int q = 0; q < PFT; ++q) bq[q] = xp[toff[q]];
        }
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const int ks = part + j * PS;
            if (ks * 4 >= npx) break;
            const int spy = (ks * 4) >> tws, spx = (ks * 4) & twm;
            const bool inb = GL || (oy0 + spy < a.Hout && ox0 + spx + lk < a.Wout);
            float af[MT], bc[PFT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                af[i] = (GL || (inb && colok[i])) ? afr[j % D][i] : 0.f;
                bsum[i] += af[i];
            }
#pragma unroll
            for (int q = 0; q < PFT; ++q) bc[q] = bq[q];
            // (look-ahead k-steps are clamped to the tile's last one: a 64-pixel tile ends before the unrolled loop does)
            if (j + D < KSW) load_g(min(part + (j + D) * PS, ks_last), afr[j % D]);
            const float* xp = xp_of(ks);
            if (j + 1 < KSW) {
                const float* xn = xp_of(min(ks + PS, ks_last));
#pragma unroll
                for (int q = 0; q < PFT; ++q) bq[q] = xn[toff[q]];
            }
#pragma unroll
            for (int tp = 0; tp < TPW; ++tp) {
                const float bf = tp < PFT ? bc[tp] : xp[toff[tp]];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[i][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf, acc[i][tp], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        d_mfma += WDIAG_NOW() - s1;
        ++d_tiles;
    }
    const long long d_epi = WDIAG_NOW();

    if (!TS && WS > 1) {               // combine the pixel shares: waves with part > 0 hand over through LDS
        __syncthreads();
        float* red = patch;            // reused: [(WS-1) * WCO][MT*TAPS*4 + MT][64]
        constexpr int PER = MT * TAPS * 4 + MT;
        if (part > 0) {
            float* dst = red + ((long)((part - 1) * WCO + wave / WS) * PER) * 64 + lane;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[((i * TAPS + tp) * 4 + r) * 64] = acc[i][tp][r];
                dst[(MT * TAPS * 4 + i) * 64] = bsum[i];
            }
        }
        __syncthreads();
        if (part > 0) return;
#pragma unroll
        for (int q = 1; q < WS; ++q) {
            const float* src = red + ((long)((q - 1) * WCO + wave / WS) * PER) * 64 + lane;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][tp][r] += src[((i * TAPS + tp) * 4 + r) * 64];
                bsum[i] += src[(MT * TAPS * 4 + i) * 64];
            }
        }
    }

    // C/D map: column j = lane&15 -> (ci = j>>1, re|im = j&1); row = (lane>>4)*4 + r -> (co = row>>1, re|im = r&1)
    const int Cin = a.C1 + a.C2;
    const long wsz = (long)TAPS * Cin * a.Cout;
    float2* slab = w.slab_w + ((long)blockIdx.x * w.ncls + cls) * wsz;
    const int ci = ci0 + (li >> 1);
    // Lane pair (2m, 2m+1) holds columns (ci, re) and (ci, im) of rows r = 0..3 = (co_a, re), (co_a, im), (co_b, re),
    // (co_b, im).  The even lane finishes co_a = (e0 + o1, e1 - o0), the odd lane co_b = (e2 + o3, e3 - o2): each sends its
    // partner the two values that one needs (one DPP quad swap each) and every lane stores one float2 — the eight lanes
    // of a ci write 64 contiguous bytes.
    const bool odd = li & 1;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int co = co0 + i * 8 + lk * 2 + (odd ? 1 : 0);
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) {
            const int tap = tap0 + tp;
            const f32x4 v = acc[i][tp];
            const float t0 = dcs_dpp_term<0xB1, 0xf>(odd ? v[0] : v[2]);     // quad_perm [1,0,3,2]: the partner's value
            const float t1 = dcs_dpp_term<0xB1, 0xf>(odd ? v[1] : v[3]);
            const float2 g = odd ? make_float2(t0 + v[3], t1 - v[2]) : make_float2(v[0] + t1, v[1] - t0);
            if (co < a.Cout && tap < TAPS) slab[((long)tap * Cin + ci) * a.Cout + co] = g;
        }
    }
    if (ci0 == 0 && (!TS || wave == 0)) {                              // bias: column sums of gY
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float s = bsum[i];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lk == 0 && colok[i]) w.slab_b[((long)blockIdx.x * w.ncls + cls) * N1 + gcol[i]] = s;
        }
    }
#ifdef DCS_WGRAD_DIAG
    if (w.dbg && lane == 0) {
        const long wg = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (wg < 8192) {
            long long* d = w.dbg + (wg * 4 + wave) * 8;
            const long long e = WDIAG_NOW();
            d[0] = d_gather; d[1] = d_mfma; d[2] = e - d_epi; d[3] = e - d_start; d[4] = d_tiles; d[5] = d_start;
            d[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
            d[7] = __builtin_amdgcn_s_getreg((3 << 11) | 20);      // HW_REG_XCC_ID [3:0]
        }
    }
#endif
}

// ---- fp32 emulated on the bf16 MFMA (the forward kernel's "bf16x6", conv_mfma.hip) --------------------------------
// Same decomposition (pixels = the K axis, all taps x MT row tiles of accumulators per wave, slabs), with
// v_mfma_f32_16x16x32_bf16: a k-step is 32 pixels, a lane holds 8 consecutive pixels of one channel for each operand.
//   * X: the patch keeps three bf16 planes per pixel ([pixel][plane][16 reals], 112-byte pitch); the B operand — 8
//     pixels x one input-channel column per lane — is read with ds_read_b64_tr_b16 (a 16-lane group fetches 4 pixel rows
//     x 16 columns and gets them column-major: two reads per plane and tap), so the image stays pixel-major as gathered;
//   * g_Y: straight from L2 as before (a lane's 8 pixels are 8 dword loads, the 16 lanes of a group cover 64 contiguous
//     bytes each) and split into its three planes in registers;
//   * six MFMAs of 16 cycles per (tap, row tile, 32 pixels) instead of eight of 32 cycles per 4 x 8 pixels.
// WS waves split a tile's four k-steps and 4 / WS waves the output channels (few output channels: the pixel partials are
// combined through LDS after the last tile, as in the native kernel); MT <= 2: the raw g_Y ring is 8 x MT registers per slot.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
#ifndef DCS_X6_FASTMASK
#define DCS_X6_FASTMASK 1
#endif
#ifndef DCS_X6_CFMAP
#define DCS_X6_CFMAP 1
#endif
#ifndef DCS_X6_FASTLOAD
#define DCS_X6_FASTLOAD 1
#endif
#ifndef DCS_X6_BPIPE
#define DCS_X6_BPIPE 0       // reads of the next tap group ahead of the current MFMAs: +50 VGPRs, step +0.4 % (measured)
#endif
#ifndef DCS_X6_TAP_GROUP
#define DCS_X6_TAP_GROUP 2
#endif
constexpr int PIXE = 28;       // LDS floats per patch pixel: 3 planes x 8 floats (16 bf16) + 4 floats of pad

// TS (the 7x7 layer with 16 output channels, as in the native kernel): the four waves split the TAPS (13 each) and every
// wave covers all MT*8 output channels and all pixels.
// PA (Round 4; plain form only: WS = 1, no tap split): operand A comes PRE-SPLIT from w.gy_planes.  Every workgroup of the
// same output-channel block — one per 8-channel input chunk: 32 of them at 256 input channels — used to load its g_Y values
// with 8 masked dword loads per row tile and k-step (64-bit per-lane addresses) and split them into the three planes itself:
// ~200 of the ~310 vector instructions a wave issued beside the 72 MFMAs of a k-step (valu / MFMA 5.1, MFMA busy 0.35:
// profiles/r04_a_pmc_wait_states.txt).  Now one pass (gy_planes_kernel) splits g_Y once per layer, edge masks included, and a
// k-step's operand A is MT x 3 unconditional 16-byte loads at a wave-uniform base.
template <int KH, int KW, int MT, int WS = 1, bool TS = false, bool PA = false>
__global__ __launch_bounds__(256, 2) void cconv_wgrad_x6_kernel(WArgs w) {
    static_assert(!PA || (WS == 1 && !TS && !DCS_ACT_IS_BF16), "pre-split g_Y: plain fp32 form only");
    constexpr int TAPS = TS ? (KH * KW + 3) / 4 : KH * KW;          // taps per wave
    constexpr int ALLTAPS = KH * KW;
    constexpr int WCO = TS ? 1 : 4 / WS;                // waves sharing the block's output channels
    extern __shared__ __attribute__((aligned(16))) float patch[];      // [rows*cols][PIXE]
    const conv::Args& a = w.c;
    const int cls = blockIdx.z;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int ci0 = (blockIdx.y / w.co_blocks) * CHUNK;
    const int part = TS ? 0 : wave % WS;                               // which k-steps of a tile
    const int tap0 = TS ? wave * TAPS : 0;
    const int co0 = (blockIdx.y % w.co_blocks) * (WCO * MT * 8) + (TS ? 0 : (wave / WS) * (MT * 8));
    const int N1 = 2 * a.Cout;
    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int npix = a.rows * a.cols;
    const int npx = w.TH * w.TW;
    const int pad_f = w.pad_f[cls], pad_t = w.pad_t[cls], oo_f = w.oo_f[cls], oo_t = w.oo_t[cls];
    // Patch pitch (floats per pixel) and the pixel <-> k-index map are chosen so that the transposed reads are free of bank
    // conflicts: a 32-lane half reads 8 pixel rows x 32 bytes = all 64 banks once if the rows are 8 CONSECUTIVE class pixels
    // at a pitch of 96 bytes (stride 1: 96 n mod 256 runs through the multiples of 32) or 112 bytes (stride 2: 224 n).
    // So lane group lk = 2h + o holds, as k-indices 8 lk .. 8 lk + 7, the pixels 16h + 4o + {0..3} and 16h + 8 + 4o + {0..3}
    // of the k-step — for BOTH operands (with 8 consecutive pixels per group every read was 2-way conflicted at any pitch).
    constexpr bool CF = DCS_X6_CFMAP;
    const int PIXR = CF ? (a.st == 2 ? PIXE : 24) : PIXE;
    // NP = 3: fp32 operands split into three bf16 planes, six MFMAs per product (the emulation).  NP = 1 (activations STORED in
    // bf16, dcs_common.h): the stored bits are the operands — plane 0 only, one MFMA per product, no split anywhere.
    constexpr int NP = DCS_ACT_IS_BF16 ? 1 : 3;

    f32x4 acc[MT][TAPS];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) acc[i][tp] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[MT];
    bool colok[MT];
    int gcol[MT], gcl[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        bsum[i] = 0.f;
        gcol[i] = 2 * (co0 + i * 8) + li;
        colok[i] = (co0 + i * 8 + (li >> 1)) < a.Cout;
        gcl[i] = colok[i] ? gcol[i] : 0;
    }
    int toff[TAPS];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
        const int tap = tap0 + tp < ALLTAPS ? tap0 + tp : ALLTAPS - 1;    // (tap-split: taps past the last repeat it and are dropped)
        toff[tp] = ((tap / KW) * a.cols + (tap % KW)) * PIXR;
    }

    // byte offsets of this lane's 8 pixels of a k-step's first row(s) from the tile origin, row tile 0 (row tile i: + 64 i bytes)
    unsigned goff[8];
    {
        const int p80 = DCS_X6_CFMAP ? 16 * (lk >> 1) + 4 * (lk & 1) : 8 * lk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pp = p80 + (DCS_X6_CFMAP ? (e < 4 ? e : e + 4) : e);
            goff[e] = (unsigned)((((pp >> w.twshift) * w.os_f * w.Wy + (pp & (w.TW - 1)) * w.os_t) * N1 + gcol[0]) * 4);
        }
    }
    long long d_gather = 0, d_mfma = 0, d_tiles = 0;
    const long long d_start = WDIAG_NOW();
    int tiles_done = 0;
    const int my_tiles = (w.total_tiles - (int)blockIdx.x + w.n_slabs - 1) / w.n_slabs;
    for (int tl = blockIdx.x; tl < w.total_tiles; tl += w.n_slabs, ++tiles_done) {
        const long long s0 = WDIAG_NOW();
        {
            const int q = tiles_done * 4 / my_tiles;
#if DCS_WG_SETPRIO
            if (q == 0) __builtin_amdgcn_s_setprio(3);
            else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#else
            (void)q;
#endif
        }
        const int b = tl / tiles_per_img, tile_id = tl % tiles_per_img;
        const int oy0 = (tile_id / a.tiles_w) * w.TH, ox0 = (tile_id % a.tiles_w) * w.TW;
        const int vy0 = oy0 * a.sf - pad_f, vx0 = ox0 * a.st - pad_t;
        const bool interior = oy0 + w.TH <= a.Hout && ox0 + w.TW <= a.Wout && co0 + MT * 8 <= a.Cout;   // (wave-uniform)
        __syncthreads();
        for (int px = t; px < npix; px += 256) {
            const int iy = (int)__umulhi((unsigned)px, w.cols_magic), ix = px - iy * a.cols;
#if DCS_ACT_IS_BF16
            uint4 hv[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};      // 8 complex channels = 16 bf16, as stored
            long sp;
            if (conv::src_pixel(a, b, vy0 + iy, vx0 + ix, &sp)) {
                const act2_t* src = (ci0 < a.C1) ? a.x1 + sp * a.C1 + ci0 : a.x2 + sp * a.C2 + (ci0 - a.C1);
                hv[0] = reinterpret_cast<const uint4*>(src)[0];
                hv[1] = reinterpret_cast<const uint4*>(src)[1];
            }
            *reinterpret_cast<uint4*>(patch + px * PIXR) = hv[0];
            *reinterpret_cast<uint4*>(patch + px * PIXR + 4) = hv[1];
            continue;
#else
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            long sp;
            if (conv::src_pixel(a, b, vy0 + iy, vx0 + ix, &sp)) {
                const float2* src = (ci0 < a.C1) ? a.x1 + sp * a.C1 + ci0 : a.x2 + sp * a.C2 + (ci0 - a.C1);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const float4*>(src)[q];
            }
            float r[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) { r[4 * q] = v[q].x; r[4 * q + 1] = v[q].y; r[4 * q + 2] = v[q].z; r[4 * q + 3] = v[q].w; }
            // the exact three-way split by PAIRS (dcs_split_pair: one packed conversion per pair; element by element the compiler
            // emitted a single-operand v_cvt_pk per value plus the repacking — Round 4)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                uint4 h0, h1;
                if (pl < 2) {
                    h0 = make_uint4(dcs_split_pair(r[0], r[1]), dcs_split_pair(r[2], r[3]), dcs_split_pair(r[4], r[5]), dcs_split_pair(r[6], r[7]));
                    h1 = make_uint4(dcs_split_pair(r[8], r[9]), dcs_split_pair(r[10], r[11]), dcs_split_pair(r[12], r[13]), dcs_split_pair(r[14], r[15]));
                } else {
                    h0 = make_uint4(dcs_pack_bf16x2(r[0], r[1]), dcs_pack_bf16x2(r[2], r[3]), dcs_pack_bf16x2(r[4], r[5]), dcs_pack_bf16x2(r[6], r[7]));
                    h1 = make_uint4(dcs_pack_bf16x2(r[8], r[9]), dcs_pack_bf16x2(r[10], r[11]), dcs_pack_bf16x2(r[12], r[13]), dcs_pack_bf16x2(r[14], r[15]));
                }
                *reinterpret_cast<uint4*>(patch + px * PIXR + pl * 8) = h0;
                *reinterpret_cast<uint4*>(patch + px * PIXR + pl * 8 + 4) = h1;
            }
#endif
        }
        __syncthreads();
        const long long s1 = WDIAG_NOW();
        d_gather += s1 - s0;
        int tws = w.twshift, twm = w.TW - 1;
        asm volatile("" : "+s"(tws), "+s"(twm));
        const act_t* gyb = w.gy + (long)b * w.Hy * w.Wy * N1;
        const int estr = w.os_t * N1;                                  // floats between horizontally adjacent g_Y pixels of the class
        const int nks = npx >> 5;
        // this lane's 8 pixels of k-step ks: tile pixel 32 ks + 8 lk + e, e = 0..7 (one tile row: TW >= 16)
        // PA: this tile's pre-split g_Y (wave-uniform base; a lane's fragment of (k-step, plane, row block) is one uint4)
        constexpr int NPA = PA ? NP : 1;
        uint4 araw[2][MT][NPA];
        const int nblk = N1 >> 4;
        const uint4* pa_tile = PA ? w.gy_planes + ((long)cls * w.total_tiles + tl) * nks * (long)(NP * nblk * 64) + lane : nullptr;
        auto load_a = [&](int ks, uint4 (*dst)[NPA]) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                int mb = __builtin_amdgcn_readfirstlane((co0 + i * 8) >> 3);
                mb = mb < nblk ? mb : 0;                               // (a block past Cout: its rows are dropped by the epilogue)
#pragma unroll
                for (int pl = 0; pl < NPA; ++pl) dst[i][pl] = pa_tile[((long)(ks * NP + pl) * nblk + mb) * 64];
            }
        };
        float raw[2][MT][8];
        // interior tiles (wave-uniform; most of them): the k-step's 8 x MT loads take a wave-uniform base (tile origin + the
        // k-step's whole rows / half rows) and the per-lane BYTE offsets goff[e] computed once per kernel — no masks, no 64-bit
        // per-lane addresses (a third of the ~310 vector instructions of a k-step in the in-kernel-split instances)
        typedef __attribute__((address_space(1))) const char gch_t;
        typedef __attribute__((address_space(1))) const float gfl_t;
        gch_t* gy_tile = (gch_t*)(gyb + ((long)(oy0 * w.os_f + oo_f) * w.Wy + ox0 * w.os_t + oo_t) * N1);
        auto load_g = [&](int ks, float (*dst)[8]) {
            if (DCS_X6_FASTLOAD && !PA && !DCS_ACT_IS_BF16 && interior) {
                int ksu = __builtin_amdgcn_readfirstlane(ks);
                gch_t* sb = gy_tile + (long)((((ksu * 32) >> tws) * w.os_f * w.Wy + ((ksu * 32) & twm) * w.os_t) * N1) * 4;
                asm volatile("" : "+s"(sb));
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    unsigned vo = goff[e];
                    asm volatile("" : "+v"(vo));
#pragma unroll
                    for (int i = 0; i < MT; ++i) dst[i][e] = *(gfl_t*)(sb + vo + 64 * i);
                }
                return;
            }
            const int p8 = CF ? ks * 32 + 16 * (lk >> 1) + 4 * (lk & 1) : ks * 32 + 8 * lk;     // e < 4: pixels p8 + e; e >= 4: p8 + 8 + (e - 4)
            const int oy = oy0 + (p8 >> tws), oxb = ox0 + (p8 & twm);
            const bool rowok = oy < a.Hout;
            const int soff = ((oy * w.os_f + oo_f) * w.Wy + oxb * w.os_t + oo_t) * N1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ee = CF ? (e < 4 ? e : e + 4) : e;
                const int off = (rowok && oxb + ee < a.Wout) ? soff + ee * estr : 0;
#pragma unroll
                for (int i = 0; i < MT; ++i) dst[i][e] = dcs_ld1(gyb + off + gcl[i]);
            }
        };
        if constexpr (PA) load_a(part < nks ? part : nks - 1, araw[0]);
        else load_g(part < nks ? part : nks - 1, raw[0]);
#pragma unroll
        for (int j = 0; j < 8 / WS; ++j) {                             // (tiles of up to 256 pixels: tile_shape_for)
            const int ks = part + j * WS;
            if (ks >= nks) break;
            const int p8 = CF ? ks * 32 + 16 * (lk >> 1) + 4 * (lk & 1) : ks * 32 + 8 * lk;
            const int py = p8 >> tws, px0 = p8 & twm;
            const bool rowok = oy0 + py < a.Hout;
            // operand A: the three planes of this lane's 8 g_Y values per row tile
            bf16x8w ap[MT][NP];
            if constexpr (PA) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) ap[i][pl] = __builtin_bit_cast(bf16x8w, araw[j & 1][i][pl < NPA ? pl : 0]);
                    if (ci0 == 0) {                                    // bias gradient: the exact value back from its three terms
                        const uint4 p0 = araw[j & 1][i][0], p1 = araw[j & 1][i][NPA > 1 ? 1 : 0], p2 = araw[j & 1][i][NPA > 2 ? 2 : 0];
                        const unsigned w0[4] = {p0.x, p0.y, p0.z, p0.w}, w1[4] = {p1.x, p1.y, p1.z, p1.w}, w2[4] = {p2.x, p2.y, p2.z, p2.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bsum[i] += (__uint_as_float(w0[q] << 16) + __uint_as_float(w1[q] << 16)) + __uint_as_float(w2[q] << 16);
                            bsum[i] += (__uint_as_float(w0[q] & 0xffff0000u) + __uint_as_float(w1[q] & 0xffff0000u)) +
                                       __uint_as_float(w2[q] & 0xffff0000u);
                        }
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float r[8];
                // (interior tiles — wave-uniform test — skip the per-value masks; only the workgroups of the first input
                // chunk need the bias sums: ~30 % of the k-step's VALU work beside its 72 MFMAs)
                if (DCS_X6_FASTMASK && interior) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = raw[j & 1][i][e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        r[e] = (rowok && ox0 + px0 + (CF ? (e < 4 ? e : e + 4) : e) < a.Wout && colok[i]) ? raw[j & 1][i][e] : 0.f;
                }
                if (!DCS_X6_FASTMASK || ci0 == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[i] += r[e];
                }
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {                       // (by pairs: dcs_split_pair / one packed conversion for the last plane)
                    uint4 hp;
                    if (pl + 1 < NP) hp = make_uint4(dcs_split_pair(r[0], r[1]), dcs_split_pair(r[2], r[3]), dcs_split_pair(r[4], r[5]), dcs_split_pair(r[6], r[7]));
                    else hp = make_uint4(dcs_pack_bf16x2(r[0], r[1]), dcs_pack_bf16x2(r[2], r[3]), dcs_pack_bf16x2(r[4], r[5]), dcs_pack_bf16x2(r[6], r[7]));
                    ap[i][pl] = __builtin_bit_cast(bf16x8w, hp);
                }
            }
            if (j + 1 < 8 / WS) {
                if constexpr (PA) load_a(ks + WS < nks ? ks + WS : nks - 1, araw[(j + 1) & 1]);
                else load_g(ks + WS < nks ? ks + WS : nks - 1, raw[(j + 1) & 1]);
            }
            // operand B: lane (row q = li / 4, column quad li % 4) of its 16-lane group addresses pixel 8 lk + q (+ 4)
            const int q = li >> 2, p4 = li & 3;
            const int xrow0 = ((py * a.sf) * a.cols + (px0 + q) * a.st) * PIXR + p4 * 2;
            const int xrow1 = xrow0 + (CF ? 8 : 4) * a.st * PIXR;
            // TG taps at a time, terms outermost: consecutive MFMAs go to TG x MT different accumulators.  The transposed reads
            // of tap group n + 1 are issued BEFORE the MFMAs of group n (two operand sets): issued right in front of their own
            // MFMAs, every group waited out an LDS round trip (s_waitcnt lgkmcnt(0) after 12 reads, three times per k-step).
            constexpr int TG = DCS_X6_TAP_GROUP;
            constexpr int NG = (TAPS + TG - 1) / TG;
            bf16x8w bp[2][TG][NP];
            auto read_group = [&](int gidx, bf16x8w (*dst)[NP]) {
#pragma unroll
                for (int u = 0; u < TG; ++u) {
                    const int tp = gidx * TG + u < TAPS ? gidx * TG + u : TAPS - 1;
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(patch + xrow0 + toff[tp] + pl * 8));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(patch + xrow1 + toff[tp] + pl * 8));
                        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        dst[u][pl] = __builtin_bit_cast(bf16x8w, both);
                    }
                }
            };
            if (DCS_X6_BPIPE) read_group(0, bp[0]);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (!DCS_X6_BPIPE) read_group(gi, bp[gi & 1]);
                else if (gi + 1 < NG) read_group(gi + 1, bp[(gi + 1) & 1]);
                if (DCS_X6_BPIPE) __builtin_amdgcn_sched_barrier(0);
                constexpr int pa[6] = {0, 1, 2, 0, 1, 0}, pb[6] = {2, 1, 0, 1, 0, 0};         // smallest terms first
#pragma unroll
                for (int e = (NP == 3 ? 0 : 5); e < 6; ++e)                                   // (NP = 1: the a0 b0 term alone)
#pragma unroll
                    for (int u = 0; u < TG; ++u)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
                            if (gi * TG + u < TAPS) {
                                f32x4& c = acc[i][gi * TG + u < TAPS ? gi * TG + u : 0];
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[i][pa[e] < NP ? pa[e] : 0], bp[gi & 1][u][pb[e] < NP ? pb[e] : 0], c, 0, 0, 0);
                            }
                if (DCS_X6_BPIPE) __builtin_amdgcn_sched_barrier(0);
            }
        }
        d_mfma += WDIAG_NOW() - s1;
        ++d_tiles;
    }
    const long long d_epi = WDIAG_NOW();

    if (!TS && WS > 1) {               // combine the pixel shares: waves with part > 0 hand over through LDS
        __syncthreads();
        float* red = patch;            // reused: [(WS-1) * WCO][MT*TAPS*4 + MT][64]
        constexpr int PER = MT * TAPS * 4 + MT;
        if (part > 0) {
            float* dst = red + ((long)((part - 1) * WCO + wave / WS) * PER) * 64 + lane;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[((i * TAPS + tp) * 4 + r) * 64] = acc[i][tp][r];
                dst[(MT * TAPS * 4 + i) * 64] = bsum[i];
            }
        }
        __syncthreads();
        if (part > 0) return;
#pragma unroll
        for (int q = 1; q < WS; ++q) {
            const float* src = red + ((long)((q - 1) * WCO + wave / WS) * PER) * 64 + lane;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][tp][r] += src[((i * TAPS + tp) * 4 + r) * 64];
                bsum[i] += src[(MT * TAPS * 4 + i) * 64];
            }
        }
    }

    // epilogue: as cconv_wgrad_mfma_kernel (same C/D map)
    const int Cin = a.C1 + a.C2;
    const long wsz = (long)ALLTAPS * Cin * a.Cout;
    float2* slab = w.slab_w + ((long)blockIdx.x * w.ncls + cls) * wsz;
    const int ci = ci0 + (li >> 1);
    const bool odd = li & 1;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int co = co0 + i * 8 + lk * 2 + (odd ? 1 : 0);
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
            const int tap = tap0 + tp;
            const f32x4 v = acc[i][tp];
            const float t0 = dcs_dpp_term<0xB1, 0xf>(odd ? v[0] : v[2]);
            const float t1 = dcs_dpp_term<0xB1, 0xf>(odd ? v[1] : v[3]);
            const float2 g = odd ? make_float2(t0 + v[3], t1 - v[2]) : make_float2(v[0] + t1, v[1] - t0);
            if (co < a.Cout && tap < ALLTAPS) slab[((long)tap * Cin + ci) * a.Cout + co] = g;
        }
    }
    if (ci0 == 0 && (!TS || wave == 0)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float s = bsum[i];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lk == 0 && colok[i]) w.slab_b[((long)blockIdx.x * w.ncls + cls) * N1 + gcol[i]] = s;
        }
    }
#ifdef DCS_WGRAD_DIAG
    if (w.dbg && lane == 0) {                                           // (tools/wgrad_diag.py: same record as the native kernel)
        const long wg = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (wg < 8192) {
            long long* d = w.dbg + (wg * 4 + wave) * 8;
            const long long e = WDIAG_NOW();
            d[0] = d_gather; d[1] = d_mfma; d[2] = e - d_epi; d[3] = e - d_start; d[4] = d_tiles; d[5] = d_start;
            d[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            d[7] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        }
    }
#endif
}

#if !DCS_ACT_IS_BF16
// g_Y -> three bf16 planes in the A-fragment order of cconv_wgrad_x6_kernel<..., PA = true> (WArgs::gy_planes): one thread per
// (class, tile, k-step, 16-row block, lane) reads the lane's 8 pixels of its (co, re|im) row — the access pattern the kernel
// itself had: the 16 lanes of a group cover 64 contiguous bytes per pixel — masks what lies outside the class extent, splits
// exactly as the kernel did (dcs_split_pair) and stores three 16-byte fragments.  Same tile / pixel / k-index maps as the kernel.
__global__ __launch_bounds__(256) void gy_planes_kernel(WArgs w, uint4* __restrict__ out) {
    const conv::Args& a = w.c;
    const int cls = blockIdx.z, mb = blockIdx.y, lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    const int npx = w.TH * w.TW, nks = npx >> 5;
    const long unit = (long)blockIdx.x * 4 + threadIdx.y;
    if (unit >= (long)w.total_tiles * nks) return;
    const int tl = (int)(unit / nks), ks = (int)(unit % nks);
    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int b = tl / tiles_per_img, tile_id = tl % tiles_per_img;
    const int oy0 = (tile_id / a.tiles_w) * w.TH, ox0 = (tile_id % a.tiles_w) * w.TW;
    const int N1 = 2 * a.Cout, nblk = N1 >> 4;
    constexpr bool CF = DCS_X6_CFMAP;
    const int p8 = CF ? ks * 32 + 16 * (lk >> 1) + 4 * (lk & 1) : ks * 32 + 8 * lk;
    const int oy = oy0 + (p8 >> w.twshift), oxb = ox0 + (p8 & (w.TW - 1));
    const float* gyb = w.gy + (long)b * w.Hy * w.Wy * N1 + (mb * 16 + li);
    const long soff = ((long)(oy * w.os_f + w.oo_f[cls]) * w.Wy + oxb * w.os_t + w.oo_t[cls]) * N1;
    const int estr = w.os_t * N1;
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ee = CF ? (e < 4 ? e : e + 4) : e;
        r[e] = (oy < a.Hout && oxb + ee < a.Wout) ? gyb[soff + (long)ee * estr] : 0.f;
    }
    uint4* dst = out + ((((long)cls * w.total_tiles + tl) * nks + ks) * 3 * nblk + mb) * 64 + lane;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        uint4 hp;
        if (pl < 2) hp = make_uint4(dcs_split_pair(r[0], r[1]), dcs_split_pair(r[2], r[3]), dcs_split_pair(r[4], r[5]), dcs_split_pair(r[6], r[7]));
        else hp = make_uint4(dcs_pack_bf16x2(r[0], r[1]), dcs_pack_bf16x2(r[2], r[3]), dcs_pack_bf16x2(r[4], r[5]), dcs_pack_bf16x2(r[6], r[7]));
        dst[(long)pl * nblk * 64] = hp;
    }
}
#endif

template <int KH_, int KW_, int MT_, int WS_, bool TS_ = false> struct Variant {
    static constexpr int KH = KH_, KW = KW_, MT = MT_, WS = WS_;
    static constexpr bool TS = TS_;
    static constexpr int CPB = TS_ ? MT_ * 8 : (4 / WS_) * MT_ * 8;                 // output channels per workgroup
    static constexpr bool GL = TS_ || (MT_ == 1 && WS_ >= 2);
    static constexpr int GLW = TS_ ? MT_ : 4 / WS_;
};

#ifndef DCS_X6_MIN_TAPS
#define DCS_X6_MIN_TAPS 0
#endif
// which emulated instance (cconv_wgrad_x6_kernel) stands in for a native variant, if any
template <class V> struct X6 {
    static constexpr bool ok = V::TS || ((V::WS == 1 || V::MT == 1) && V::KH * V::KW > DCS_X6_MIN_TAPS && V::KH < 7 && !(V::KH == 5 && V::WS == 4));
    static constexpr int MT = V::TS ? V::MT : (V::MT > 2 ? 2 : V::MT);
    static constexpr int WS = V::TS ? 1 : V::WS;
    static constexpr int CPB = V::TS ? V::MT * 8 : (4 / WS) * MT * 8;
};

inline bool wgrad_x6_enabled() {
    static const int e = [] { const char* v = getenv("DCS_WGRAD_X6"); return v ? atoi(v) : 1; }();   // (0: native kernels in every mode)
    if (DCS_ACT_IS_BF16) return true;        // bf16 activations: the same kernels with ONE plane (operands are the stored bits)
    return e != 0 && dcs_conv_precision() == 2;
}

template <int KH, int KW, int MT, int WS, bool TS, bool PA = false>
int resident_x6(size_t lds) {
    static int by_kb[161] = {0};
    int& cached = by_kb[lds / 1024 > 160 ? 160 : lds / 1024];
    if (cached == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cconv_wgrad_x6_kernel<KH, KW, MT, WS, TS, PA>, 256, lds) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 1;
        }
        // (DCS_WGRAD_MAX_PER_CU: experiments with a kernel that leaves room on every CU for the kernels it runs beside)
        static const int cap = (int)dcs_knob("DCS_WGRAD_MAX_PER_CU", 4);
        cached = n > cap ? cap : n;
    }
    return cached;
}

// (MT, WS) so that the 4 waves cover min(Cout, most-per-kernel-size) output channels without idle lanes
template <class F>
int dispatch(int kh, int kw, int co, F&& f) {
    switch (kh * 10 + kw) {
        case 11:
            if (co >= 128) return f(Variant<1, 1, 4, 1>{});
            if (co >= 64) return f(Variant<1, 1, 2, 1>{});
            if (co >= 32) return f(Variant<1, 1, 1, 1>{});
            if (co >= 16) return f(Variant<1, 1, 1, 2>{});
            return f(Variant<1, 1, 1, 4>{});
        case 22:                                   // 3x3 folded over a (2,2) upsample
            if (co >= 128) return f(Variant<2, 2, 4, 1>{});
            if (co >= 64) return f(Variant<2, 2, 2, 1>{});
            if (co >= 32) return f(Variant<2, 2, 1, 1>{});
            if (co >= 16) return f(Variant<2, 2, 1, 2>{});
            return f(Variant<2, 2, 1, 4>{});
        case 23:                                   // 3x3 folded over a (2,1) upsample
            if (co >= 128) return f(Variant<2, 3, 4, 1>{});
            if (co >= 64) return f(Variant<2, 3, 2, 1>{});
            if (co >= 32) return f(Variant<2, 3, 1, 1>{});
            if (co >= 16) return f(Variant<2, 3, 1, 2>{});
            return f(Variant<2, 3, 1, 4>{});
        case 33:
            if (co >= 128) return f(Variant<3, 3, 4, 1>{});
            if (co >= 64) return f(Variant<3, 3, 2, 1>{});
            if (co >= 32) return f(Variant<3, 3, 1, 1>{});
            if (co >= 16) return f(Variant<3, 3, 1, 2>{});
            return f(Variant<3, 3, 1, 4>{});
        case 55:
            // (MT = 2 for >= 64 output channels holds 200 accumulator registers: one workgroup per CU, every stall exposed;
            //  two 32-channel blocks gather the patch twice and run at three workgroups per CU)
            if (co >= 32) return f(Variant<5, 5, 1, 1>{});
            if (co >= 16) return f(Variant<5, 5, 1, 2>{});
            return f(Variant<5, 5, 1, 4>{});
        case 77:
            if (co >= 32) return f(Variant<7, 7, 1, 1>{});
            if (co >= 16) return f(Variant<7, 7, 2, 1, true>{});
            return f(Variant<7, 7, 1, 4>{});
        default: return DCS_ERR_BADARG;
    }
}

template <class V>
size_t lds_bytes(int rows, int cols, int npx) {
    size_t lds = (size_t)rows * cols * PIX * sizeof(float);
    if (V::GL) lds += (size_t)npx * (V::GLW * 16 + (V::GLW == 1 ? 0 : 16)) * sizeof(float);   // g_Y tile
    const size_t red = V::TS ? 0 : (size_t)(V::WS - 1) * (4 / V::WS) * (V::MT * V::KH * V::KW * 4 + V::MT) * 64 * sizeof(float);
    return red > lds ? red : lds;
}

// workgroups of this variant one CU holds at once (registers / LDS), queried once per variant
template <class V>
int resident_per_cu(size_t lds) {
    static int by_kb[161] = {0};                        // per LDS size (KB): the same variant serves several tile geometries
    int& cached = by_kb[lds / 1024 > 160 ? 160 : lds / 1024];
    if (cached == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cconv_wgrad_mfma_kernel<V::KH, V::KW, V::MT, V::WS, V::TS>, 256, lds) !=
                hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 1;
        }
        cached = n > 4 ? 4 : n;
    }
    return cached;
}

// the emulated kernel for a plain (non-GL, WS = 1) variant: MT capped at 2 (more, smaller output-channel blocks)
template <int KH, int KW, int MT, int WS, bool TS = false>
int launch_x6(WArgs& w, int Cin, hipStream_t stream) {
    const conv::Args& a = w.c;
    size_t lds = (size_t)a.rows * a.cols * PIXE * sizeof(float);
    const size_t red = TS ? 0 : (size_t)(WS - 1) * (4 / WS) * (MT * KH * KW * 4 + MT) * 64 * sizeof(float);
    if (red > lds) lds = red;
    if (lds > 150 * 1024) return DCS_ERR_BADARG;
    constexpr int CPBX = TS ? MT * 8 : (4 / WS) * MT * 8;
    w.co_blocks = (a.Cout + CPBX - 1) / CPBX;
    dim3 grid(w.n_slabs, (Cin / CHUNK) * w.co_blocks, w.ncls);
    if (grid.y > 65535) return DCS_ERR_BADARG;
#if !DCS_ACT_IS_BF16
    if constexpr (WS == 1 && !TS) {
        if (w.gy_planes) {                                             // g_Y split once for all input-channel chunks (PA)
            const long units = (long)w.total_tiles * ((w.TH * w.TW) >> 5);
            dim3 pgrid((unsigned)((units + 3) / 4), (unsigned)(2 * a.Cout / 16), (unsigned)w.ncls);
            DCS_LAUNCH(gy_planes_kernel, pgrid, dim3(64, 4), 0, stream, w, const_cast<uint4*>(w.gy_planes));
            DCS_CHECK_LAUNCH();
            auto fnp = cconv_wgrad_x6_kernel<KH, KW, MT, WS, TS, true>;
            if (dcs_ensure_dynamic_lds((const void*)fnp, lds) != hipSuccess) return DCS_ERR_LAUNCH;
            DCS_LAUNCH(fnp, grid, dim3(256), lds, stream, w);
            DCS_CHECK_LAUNCH();
            return DCS_OK;
        }
    }
#endif
    auto fn = cconv_wgrad_x6_kernel<KH, KW, MT, WS, TS>;
    if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
    DCS_LAUNCH(fn, grid, dim3(256), lds, stream, w);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

template <class V>
int launch(WArgs& w, int Cin, hipStream_t stream) {
    const conv::Args& a = w.c;
    if constexpr (X6<V>::ok) {
        if (wgrad_x6_enabled() && w.TW >= 16 && (w.TH * w.TW) % 32 == 0)
            return launch_x6<V::KH, V::KW, X6<V>::MT, X6<V>::WS, V::TS>(w, Cin, stream);
    }
    size_t lds = lds_bytes<V>(a.rows, a.cols, w.TH * w.TW);
    if (lds > 150 * 1024) return DCS_ERR_BADARG;
    auto fn = cconv_wgrad_mfma_kernel<V::KH, V::KW, V::MT, V::WS, V::TS>;
    const int co_per_block = V::CPB;
    w.co_blocks = (a.Cout + co_per_block - 1) / co_per_block;
    dim3 grid(w.n_slabs, (Cin / CHUNK) * w.co_blocks, w.ncls);
    if (grid.y > 65535) return DCS_ERR_BADARG;
    // All workgroups are resident at once and do equal work, so the kernel ends with the CU that was handed the most of
    // them — and the dispatcher fills a CU as far as its registers and LDS allow (dec2: 1024 workgroups planned as four per
    // CU landed as 3 / 4 / 5, lifetimes 76 / 100 / 124 us).  Ask for just enough LDS that one more than planned cannot fit.
    {
        const long wgs = (long)grid.x * grid.y * grid.z;
        const long per = (wgs + 255) / 256;
        if (per <= resident_per_cu<V>(lds)) {
            const size_t want = (size_t)160 * 1024 / (per + 1) + 2048;
            if (want > lds && want <= 150 * 1024) lds = want;
        }
    }
    if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
    DCS_LAUNCH(fn, grid, dim3(256), lds, stream, w);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// ---- upsample-folded weight gradient ---------------------------------------------------------------
// A 3x3 stride-1 conv over a nearest-upsampled input (every decoder stage, c_network.py:214-216) is, per output
// parity class, a 2-tap (upsampled axis) x 3-tap conv over the SOURCE-resolution input with tap-summed weights
// (conv_pack.hip).  Its weight gradient is taken in the same form — per class, over the un-upsampled input, 6/9
// or 4/9 of the MACs — and mapped back: g_W[dy][dx] = sum over classes of g_Wfold_c[jy_c(dy)][jx_c(dx)].
struct Fold {
    int ncls, kh, kw, os_f, os_t;
    int pad_f[4], pad_t[4], oo_f[4], oo_t[4];
};

Fold fold_of(const conv::Args& a) {
    Fold f{};
    f.kh = a.up_f == 2 ? 2 : 3; f.kw = a.up_t == 2 ? 2 : 3;
    f.os_f = a.up_f; f.os_t = a.up_t; f.ncls = a.up_f * a.up_t;
    for (int ry = 0; ry < a.up_f; ++ry)
        for (int rx = 0; rx < a.up_t; ++rx) {
            const int c = ry * a.up_t + rx;
            f.pad_f[c] = a.up_f == 2 ? (ry == 0 ? 1 : 0) : 1;
            f.pad_t[c] = a.up_t == 2 ? (rx == 0 ? 1 : 0) : 1;
            f.oo_f[c] = ry; f.oo_t[c] = rx;
        }
    return f;
}

// class-space problem: source-resolution input, class extent = source extent, folded kernel size
conv::Args class_args(const conv::Args& a, const Fold& f) {
    conv::Args c = a;
    c.up_f = 1; c.up_t = 1; c.zero_ins = 0; c.Hv = a.Hin; c.Wv = a.Win;
    c.kh = f.kh; c.kw = f.kw; c.sf = 1; c.st = 1;
    c.Hout = a.Hin; c.Wout = a.Win;
    return c;
}

long planes_bytes_for(const conv::Args& c, int ncls, int TH, int TW);

// slabs per class for a class-space geometry `c` (tiling filled in), ncls classes
int slabs_for(const conv::Args& c, int ncls, int TH, int TW) {
    const long tiles = (long)((c.Wout + TW - 1) / TW) * ((c.Hout + TH - 1) / TH) * c.B;
    const long wsz = (long)c.kh * c.kw * (c.C1 + c.C2) * c.Cout;
    // Bytes of partial slabs a layer may write.  Every slab is a full weight gradient written by the kernel and read back by the
    // reduction — at [32,256,256] 194 + 200 MB per step under the 96 MB budget of rounds 2-4, more than the kernels' operands — and
    // the kernels run on the side stream beside the data-gradient chain, which pays for that traffic: with 16 MB per layer the
    // weight-gradient kernels themselves are 2 % slower (fewer, longer workgroups on enc4 .. dec3) and the STEP 1.3 % faster
    // (3.267 -> 3.224 ms, same box; 8 MB: 3.34).  The optimum moves with the pixel count: 16 MB up to B = 32 (B = 16: -33 us,
    // bf16 storage B = 32: -16 us), 32 MB at B = 64 (fp32 -14 us, bf16 -8 us; 16 MB loses there): profiles/r05_wgrad_slab_budget.txt.
    static const long slab_mb_knob = dcs_knob("DCS_WGRAD_SLAB_MB", 0);        // (plan sweeps: a fixed budget instead)
    const long slab_mb = slab_mb_knob > 0 ? slab_mb_knob : 16L * (c.B > 32 ? c.B : 32) / 32;
    long cap = (slab_mb << 20) / (wsz * ncls * (long)sizeof(float2));
    if (cap < 1) cap = 1;
    if (cap > 1024) cap = 1024;
    // A workgroup's epilogue writes its whole accumulator set (up to 147 KB), so give each one several pixel
    // tiles rather than one — but keep every CU holding as many workgroups as the variant's registers allow
    // (they overlap each other's gathers): target = 256 CUs x resident workgroups per CU.
    const int rows = (TH - 1) * c.sf + c.kh, cols = (TW - 1) * c.st + c.kw;
    int per_cu = 1, cpb = 8;
    dispatch(c.kh, c.kw, c.Cout, [&](auto v) {
        using V = decltype(v);
        per_cu = resident_per_cu<V>(lds_bytes<V>(rows, cols, TH * TW));
        cpb = V::CPB;
        if constexpr (X6<V>::ok) {
            if (wgrad_x6_enabled() && TW >= 16 && (TH * TW) % 32 == 0) {     // the emulated instance: its own block size / residency
                cpb = X6<V>::CPB;
                per_cu = resident_x6<V::KH, V::KW, X6<V>::MT, X6<V>::WS, V::TS>((size_t)rows * cols * PIXE * sizeof(float));
#if !DCS_ACT_IS_BF16
                // The pre-split form holds ~40 registers less (four workgroups per CU instead of three).  Planned that way
                // (DCS_WGRAD_PA_RESIDENCY=1) the kernels alone gain another 7 % (dec1 82 -> 76 us) but the train step LOSES 0.035 ms:
                // the kernels run on the side stream beside the data gradients, which then find less room (profiles/r04_wgrad_pa.txt).
                if constexpr (!V::TS && X6<V>::WS == 1) {
                    static const int pa_res = (int)dcs_knob("DCS_WGRAD_PA_RESIDENCY", 0);
                    if (pa_res && planes_bytes_for(c, ncls, TH, TW) > 0)
                        per_cu = resident_x6<V::KH, V::KW, X6<V>::MT, 1, false, true>((size_t)rows * cols * PIXE * sizeof(float));
                }
#endif
            }
        }
        return 0;
    });
    const long grid_y = (long)((c.C1 + c.C2) / CHUNK) * ((c.Cout + cpb - 1) / cpb) * ncls;
    if (tiles <= 1) return 1;
    // A CU's MFMA pipe serialises the tiles of all workgroups it holds, and the kernel ends with its fullest CU: pick
    // the slab count n whose busiest CU has the least work = workgroups per CU x (tiles per workgroup + the workgroup's
    // fixed epilogue, ~0.2 tile) / how well that many co-resident workgroups hide each other's gathers and LDS round
    // trips (measured: dec2 / dec3 / dec4 at four per CU beat two per CU with twice the tiles by 10-25 %; 704 workgroups
    // = 2.75 per CU with three tiles each lose to 512 with four tiles each); more than fit at once run in rounds.
    long best = 1;
    double best_cost = 1e30;
    const long nmax = tiles < cap ? tiles : cap;
    for (long n = 1; n <= nmax; ++n) {
        const long wgs = n * grid_y, per = (wgs + 255) / 256, tpw = (tiles + n - 1) / n;
        static const double eff[5] = {1.0, 0.70, 0.88, 0.95, 1.0};
        double cost = (double)per * ((double)tpw + 0.2) / eff[per < per_cu ? per : per_cu];
        if (per > per_cu) cost *= 1.15;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = n; }
    }
    return (int)best;
}

void tile_shape(int Hc, int Wc, int kh, int sf, int* TH, int* TW) {
    // a 7x7 stride-2 patch of an 8 x 16 tile is 21 x 37 pixels = 62 KB: one workgroup per CU by LDS; 4 x 16 tiles (38 KB)
    if (kh >= 7 && sf >= 2 && Hc >= 4) { *TH = 4; *TW = 16; }
    else if (Hc >= 8) { *TH = 8; *TW = 16; }
    else if (Hc >= 4) { *TH = 4; *TW = 32; }
    else if (Wc > 32) { *TH = 2; *TW = 64; }
    else { *TH = 2; *TW = 32; }          // a map too small to fill 128 pixels (enc6 at T = 256: 2 x 32): 64-pixel tiles
}

// The emulated kernel's tile is four k-steps of 32 pixels between two barrier pairs; where the map has the rows and the
// three-plane patch stays small it takes 256-pixel tiles (twice the rows): half as many gathers, barriers and first-fragment
// round trips per pixel.  g: the geometry the launch dispatches on (kh, kw, strides, Cout); Hc x Wc: the (class) extent.
void tile_shape_for(const conv::Args& g, int Hc, int Wc, int* TH, int* TW) {
    tile_shape(Hc, Wc, g.kh, g.sf, TH, TW);
    static const int big = (int)dcs_knob("DCS_WGRAD_X6_BIG", 1);
    if (!big || !wgrad_x6_enabled() || *TW < 16 || ((*TH) * (*TW)) % 32 != 0 || (*TH) * (*TW) != 128) return;
    bool ok = false;
    dispatch(g.kh, g.kw, g.Cout, [&](auto v) { ok = X6<decltype(v)>::ok; return 0; });
    if (!ok || Hc < 2 * (*TH)) return;
    const long patch = (long)((2 * (*TH) - 1) * g.sf + g.kh) * ((*TW - 1) * g.st + g.kw) * PIXE * (long)sizeof(float);
    if (patch <= 48 * 1024) *TH *= 2;
}

// bytes of the pre-split g_Y (WArgs::gy_planes) for a class-space geometry, or 0 where the launch would not use it: the plain
// emulated form (>= 32 output channels, kernel below 7x7) with at least DCS_WGRAD_PA_MIN_CHUNKS input-channel chunks re-reading
// the same g_Y (the split pass costs a read and 1.5 writes of g_Y; below ~8 chunks the kernels' own split is cheaper)
long planes_bytes_for(const conv::Args& c, int ncls, int TH, int TW) {
    if (DCS_ACT_IS_BF16 || !wgrad_x6_enabled() || TW < 16 || (TH * TW) % 32 != 0 || (c.Cout % 8) != 0) return 0;
    static const int min_chunks = [] { const char* e = getenv("DCS_WGRAD_PA_MIN_CHUNKS"); return e ? atoi(e) : 8; }();
    if ((c.C1 + c.C2) / CHUNK < min_chunks) return 0;
    bool ok = false;
    dispatch(c.kh, c.kw, c.Cout, [&](auto v) {
        using V = decltype(v);
        ok = X6<V>::ok && !V::TS && X6<V>::WS == 1;
        return 0;
    });
    if (!ok) return 0;
    const long tiles = (long)((c.Wout + TW - 1) / TW) * ((c.Hout + TH - 1) / TH) * c.B;
    return (long)ncls * tiles * ((TH * TW) >> 5) * 3 * (2 * c.Cout / 16) * 64 * (long)sizeof(uint4);
}

// c: class-space geometry; f: classes; Hy x Wy: full g_Y extent; planes: room for planes_bytes_for(...) bytes, or nullptr
int launch_classes(const conv::Args& c, const Fold& f, int Hy, int Wy, const act_t* gy, float2* slab_w, float* slab_b,
                   int n_slabs, int TH, int TW, hipStream_t stream, void* planes = nullptr) {
    WArgs w;
    w.c = c;
    w.gy_planes = (planes && planes_bytes_for(c, f.ncls, TH, TW) > 0) ? (const uint4*)planes : nullptr;
#ifdef DCS_WGRAD_DIAG
    w.dbg = g_wdbg;
#else
    w.dbg = nullptr;
#endif
    w.gy = gy; w.slab_w = slab_w; w.slab_b = slab_b; w.n_slabs = n_slabs;
    w.TH = TH; w.TW = TW;
    w.twshift = TW == 16 ? 4 : (TW == 32 ? 5 : 6);
    w.c.tiles_w = (c.Wout + TW - 1) / TW;
    w.c.tiles_h = (c.Hout + TH - 1) / TH;
    w.c.rows = (TH - 1) * c.sf + c.kh;
    w.c.cols = (TW - 1) * c.st + c.kw;
    w.cols_magic = (unsigned)(((1ULL << 32) + w.c.cols - 1) / w.c.cols);
    w.total_tiles = w.c.tiles_w * w.c.tiles_h * c.B;
    w.ncls = f.ncls; w.os_f = f.os_f; w.os_t = f.os_t; w.Hy = Hy; w.Wy = Wy;
    for (int i = 0; i < 4; ++i) { w.pad_f[i] = f.pad_f[i]; w.pad_t[i] = f.pad_t[i]; w.oo_f[i] = f.oo_f[i]; w.oo_t[i] = f.oo_t[i]; }
    const int Cin = c.C1 + c.C2;
    return dispatch(c.kh, c.kw, c.Cout, [&](auto v) { return launch<decltype(v)>(w, Cin, stream); });
}

}  // namespace

#if defined(DCS_WGRAD_DIAG) && !defined(DCS_ACT_BF16)
extern "C" int dcs_debug_set_wgrad_buffer(void* p) { g_wdbg = (long long*)p; return 0; }
#endif

bool dcs_conv_wgrad_mfma_ok(int Cin, int Cout, int kh, int kw, int C1) {
    return (Cin % 8) == 0 && (Cout % 8) == 0 && kh == kw && (kh == 1 || kh == 3 || kh == 5 || kh == 7) && !(C1 & 1);
}

// number of partial slabs and the tile shape for a forward geometry (a.Hout / a.Wout set)
int dcs_conv_wgrad_mfma_slabs(const conv::Args& a, int* TH, int* TW) {
    tile_shape_for(a, a.Hout, a.Wout, TH, TW);
    return slabs_for(a, 1, *TH, *TW);
}

// slab_w: float2[n_slabs][taps][Cin][Cout]; slab_b: float[n_slabs][2*Cout]
int dcs_conv_wgrad_mfma_launch(conv::Args& a, const act_t* gy, float2* slab_w, float* slab_b, int n_slabs,
                               hipStream_t stream, void* planes) {
    int TH, TW;
    tile_shape_for(a, a.Hout, a.Wout, &TH, &TW);
    Fold f{};
    f.ncls = 1; f.kh = a.kh; f.kw = a.kw; f.os_f = 1; f.os_t = 1;
    f.pad_f[0] = a.pad_f; f.pad_t[0] = a.pad_t;
    return launch_classes(a, f, a.Hout, a.Wout, gy, slab_w, slab_b, n_slabs, TH, TW, stream, planes);
}

// bytes of the pre-split g_Y the plain (unfolded) launch of geometry `a` can use (0: none) — behind the slabs in the workspace
long dcs_conv_wgrad_mfma_planes_bytes(const conv::Args& a) {
    int TH, TW;
    tile_shape_for(a, a.Hout, a.Wout, &TH, &TW);
    return planes_bytes_for(a, 1, TH, TW);
}

// upsample-folded path: forward geometry `a` (a.Hout / a.Wout set) of a 3x3 stride-1 pad-1 conv over an upsampled input
bool dcs_conv_wgrad_fold_ok(const conv::Args& a) {
    const int Cin = a.C1 + a.C2;
    if (!(dcs_conv_wgrad_mfma_ok(Cin, a.Cout, a.kh, a.kw, a.C1) &&
          conv::fold_ok(Cin, a.Cout, a.kh, a.kw, a.sf, a.st, a.pad_f, a.pad_t, a.up_f, a.up_t)))
        return false;
    // 128-pixel tiles x taps actually issued: a class of a very small source map (dec0: 2 x 32) can fill its tiles
    // worse than the plain form does
    const Fold f = fold_of(a);
    int th, tw;
    tile_shape(a.Hin, a.Win, 3, 1, &th, &tw);
    const long folded = (long)((a.Win + tw - 1) / tw) * ((a.Hin + th - 1) / th) * th * tw * f.ncls * f.kh * f.kw;
    tile_shape(a.Hout, a.Wout, a.kh, a.sf, &th, &tw);
    const long plain = (long)((a.Wout + tw - 1) / tw) * ((a.Hout + th - 1) / th) * th * tw * a.kh * a.kw;
    return folded < plain;
}

long dcs_conv_wgrad_fold_workspace_bytes(const conv::Args& a) {
    const Fold f = fold_of(a);
    const conv::Args c = class_args(a, f);
    int TH, TW;
    tile_shape_for(c, c.Hout, c.Wout, &TH, &TW);
    const long ns = slabs_for(c, f.ncls, TH, TW);
    const long slabs = ns * f.ncls * ((long)f.kh * f.kw * (a.C1 + a.C2) * a.Cout + a.Cout) * (long)sizeof(float2);
    return ((slabs + 255) & ~255L) + planes_bytes_for(c, f.ncls, TH, TW);       // (+ the pre-split g_Y behind the slabs)
}

int dcs_conv_wgrad_fold_run(const conv::Args& a, const act_t* gy, void* workspace, long workspace_bytes, float* gw_r,
                            float* gw_i, float* gb_r, float* gb_i, int transposed, hipStream_t stream) {
    if (!dcs_conv_wgrad_fold_ok(a)) return DCS_ERR_BADARG;
    const Fold f = fold_of(a);
    const conv::Args c = class_args(a, f);
    int TH, TW;
    tile_shape_for(c, c.Hout, c.Wout, &TH, &TW);
    const int ns = slabs_for(c, f.ncls, TH, TW);
    const int Cin = a.C1 + a.C2;
    const long wsz_c = (long)f.kh * f.kw * Cin * a.Cout;
    if (workspace_bytes < (long)ns * f.ncls * (wsz_c + a.Cout) * (long)sizeof(float2)) return DCS_ERR_WORKSPACE;
    float2* slab_w = (float2*)workspace;
    float2* slab_b = slab_w + (long)ns * f.ncls * wsz_c;
    const long slabs = (long)ns * f.ncls * (wsz_c + a.Cout) * (long)sizeof(float2), pb = planes_bytes_for(c, f.ncls, TH, TW);
    void* planes = (pb > 0 && workspace_bytes >= ((slabs + 255) & ~255L) + pb) ? (char*)workspace + ((slabs + 255) & ~255L) : nullptr;
    const int rc = launch_classes(c, f, a.Hout, a.Wout, gy, slab_w, (float*)slab_b, ns, TH, TW, stream, planes);
    if (rc != DCS_OK) return rc;
    wreduce::Job j{};                                     // class gradients -> 3x3 parameter (wgrad_reduce.hip)
    j.slab_w = slab_w; j.slab_b = slab_b; j.gw_r = gw_r; j.gw_i = gw_i; j.gb_r = gb_r; j.gb_i = gb_i;
    j.n_slabs = ns; j.Cout = a.Cout; j.Cin = Cin; j.kh = 3; j.kw = 3; j.transposed = transposed;
    j.up_f = a.up_f; j.up_t = a.up_t;
    return wreduce::emit(j, stream);
    return DCS_OK;
}
