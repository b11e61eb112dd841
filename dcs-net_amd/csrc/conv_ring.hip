// conv_ring.hip — the complex conv implicit GEMM with a producer / consumer split of a workgroup's waves (gfx950, Round 5).
//
// Same arithmetic as cconv_mfma_kernel (conv_mfma.hip: the reference's four real convolutions per complex one,
// c_network.py:107-112, :135-147, as ONE real GEMM through the 2x2 embedding; fp32 products exactly, as six bf16 MFMAs on
// three-way operand splits, or one bf16 MFMA on bf16-stored activations), same B panels, same patch layout, same
// accumulation order per output element — a different schedule.  Round 4 measured what bounds the classic kernel: its four
// waves gather, wait, multiply and store in turn, a lone wave reaches 61-73 % of its MFMA pipe with both operand streams in its
// instruction stream and 93 % without them, and no re-arrangement inside a four-wave, single-buffer workgroup moves the
// per-CU time.  Here a workgroup is EIGHT waves on one 128-pixel x 64-column tile:
//
//   waves 0-3  consumers: 64 pixels x 32 columns each (2 x 2), one per SIMD.  Their instruction stream is ds_read_b128 +
//              MFMA only — no vector-memory instruction, no vmcnt wait, no conversion: A fragments from the current patch
//              buffer, B fragments from the ring slot of the current step, both one item (k-group) ahead in registers.
//   waves 4-7  producers, one per SIMD beside a consumer.  (i) the B panel of the step D steps ahead, by LDS-DMA
//              (global_load_lds_dwordx4: 1 KB fragments land in a ring slot in fragment order, no registers, no VALU);
//              (ii) the NEXT chunk's haloed input patch, a slice per step: table look-up, 16-byte loads, the exact
//              three-way bf16 split (fp32 storage) or nothing (bf16 storage), LDS stores into the OTHER patch buffer.
//
// One s_barrier per step (TPS taps x U k-groups = IPS items of 12 MFMAs per consumer wave).  Invariant at the barrier in
// front of step s: B stages <= s + 1 have landed and the patch of the chunk of step s + 1 is complete — one step early, so
// that a consumer's register prefetch of the next step's first item needs no wait behind the barrier.  A ring slot / patch
// buffer is rewritten only behind the barrier that follows the last step reading it.
//
// Tile shapes other than 128 x 64, K slices and the 16-column layers stay on conv_mfma.hip; dcs_conv_ring_plan() says
// whether a geometry runs here.
#include "conv_common.h"
#include "conv_mfma_args.h"
#include "conv_ring.h"
#include <cstdio>
#include <cstdlib>

#ifndef DCS_RING_PRIO_CONS
#define DCS_RING_PRIO_CONS 3
#endif
#ifndef DCS_RING_PRIO_PROD
#define DCS_RING_PRIO_PROD 3
#endif
#ifndef DCS_RING_ALT_ACC
#define DCS_RING_ALT_ACC 0
#endif
#ifndef DCS_RING_EXP
#define DCS_RING_EXP 0      // timing probes (wrong results): 1 no patch slices after the prologue, 2 no B requests after the prologue,
#endif                      // 4 no fragment reads in the consumer loop, 8 no barriers in the loop, 16 no main loop at all

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char lds_c;
typedef __attribute__((address_space(3))) const bf16x8 lds_bf8_t;

#if DCS_ACT_IS_BF16
typedef uint4 raw_t;                                               // a 16-byte source slot: four complex bf16 values
#else
typedef float4 raw_t;                                              // ... two complex fp32 values
#endif

#define RING_BARRIER() asm volatile("s_barrier" ::: "memory")

template <int PR, int CH, int TPS, int NA, bool STAT>
__global__ __launch_bounds__(512) void cconv_ring_kernel(MArgs m, RingP rp) {
    constexpr int NP = PR == 2 ? 3 : 1;                                // bf16 planes per operand
    constexpr int U = CH / 8, IPS = TPS * U;                           // k-groups per tap and chunk; items per step
    constexpr int PIXW = NP * CH + 4;                                  // words per patch pixel (as conv_mfma.hip)
    constexpr int STAGE = IPS * NP * 2 * 1024;                         // bytes of one B stage: [item][plane][2 column tiles][1 KB]
    constexpr int WM = 2;                                              // 32-pixel m-tiles per consumer wave
    static_assert((IPS & 1) == 0, "fragment sets alternate by item: an even number of items per step");
    static_assert(DCS_ACT_IS_BF16 ? PR == 1 : PR == 2, "fp32 storage: the exact emulation; bf16 storage: bf16 operands");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lds_c* const L = (lds_c*)lds;
    const conv::Args& a = m.c;

    // XCD-aware tile order (as cconv_mfma_kernel): XCD j owns a contiguous range of the logical order (x fastest) — all the
    // pixel tiles of one or two (column pair, class) — so the ~1 MB of B its workgroups stream stays in its L2
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const unsigned n = (bz * gy + by) * gx + bx, xcd = n & 7u, idx = n >> 3, q = total >> 3, r = total & 7u;
        const unsigned Lg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        bx = __builtin_amdgcn_readfirstlane(Lg % gx);
        by = __builtin_amdgcn_readfirstlane((Lg / gx) % gy);
        bz = __builtin_amdgcn_readfirstlane(Lg / (gx * gy));
    }
    const conv::Cls& k = m.cls[bz];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int b = bx / tiles_per_img, tile_id = bx % tiles_per_img;
    const int oy0 = (tile_id / a.tiles_w) * m.TH, ox0 = (tile_id % a.tiles_w) * m.TW;
    float* const stat_row = STAT ? a.stat + ((long)bz * gridDim.x + bx) : nullptr;
    if (oy0 >= k.Hc || ox0 >= k.Wc) {                                  // tile outside this (smaller) class
        if (STAT) {
            for (int o = t; o < 2 * 80; o += 512) {
                const int e = o % 80, c = (((int)by * 2 + o / 80) * 32 + 4 * (e & 7)) / 2 + e / 40;
                if (c < a.Cout) stat_row[(long)(c * 5 + ((e % 40) >> 3)) * a.stat_stride] = 0.f;
            }
        }
        return;
    }
    if (DCS_RING_EXP & 32) return;                                     // (probe 32: the launch alone)
    const int vy0 = oy0 * a.sf - k.pad_f, vx0 = ox0 * a.st - k.pad_t;
    const int Cin = a.C1 + a.C2;
    const int kw = k.kw, ntaps = k.kh * k.kw;
    const int cols = (m.TW - 1) * a.st + k.kw, rows = (m.TH - 1) * a.sf + k.kh;
    const int npix = rows * cols;
    const int n_chunks = Cin / CH;
    const int SPC = ntaps / TPS;                                       // steps per chunk (>= 2: dcs_conv_ring_plan)
    const int nsteps = (DCS_RING_EXP & 16) ? 0 : n_chunks * SPC;      // (probe 16: prologue + epilogue only)
    const int abytes = rp.abytes;

    // source pixel of every patch pixel as the BYTE offset of its channel 0 in x1 / x2, -1: a zero (conv_mfma.hip)
    int* const spx = reinterpret_cast<int*>((char*)lds + 2 * abytes + rp.R * STAGE);
    int* const spx2 = spx + npix;
    {
        const unsigned cols_magic = 0xFFFFFFFFu / (unsigned)cols + 1u;
        for (int p = t; p < npix; p += 512) {
            const int py = (int)__umulhi((unsigned)p, cols_magic), px = p - py * cols;
            long sp;
            const bool in = conv::src_pixel(a, b, vy0 + py, vx0 + px, &sp);
            spx[p] = in ? (int)sp * a.C1 * (int)sizeof(act2_t) : -1;
            spx2[p] = in ? (int)sp * a.C2 * (int)sizeof(act2_t) : -1;
        }
    }
    __syncthreads();
    if (DCS_RING_EXP & 128) return;                                    // (probe 128: launch + table)

    // ---- the patch gather, shared by the prologue (all eight waves, chunk 0) and the producers' slices -----------------
    constexpr int SB = 16 / (int)sizeof(act2_t);                       // complex values per 16-byte source slot: 2 (fp32) / 4 (bf16)
    constexpr int Q = CH / SB;                                         // slots per pixel and chunk
    static_assert(256 % Q == 0, "a lane owns a fixed channel slot");
    typedef __attribute__((address_space(1))) const char gsrc_t;
    typedef float f32x4g __attribute__((ext_vector_type(4)));
    gsrc_t* xs1 = (gsrc_t*)a.x1;                                       // (SGPR pairs, read from the argument block once)
    gsrc_t* xs2 = (gsrc_t*)a.x2;
    asm volatile("" : "+s"(xs1), "+s"(xs2));
    const int C1 = a.C1;
    // conversion + LDS store of one slot: raw 16 bytes r of patch pixel pp (table entry o), channel slot tq, into patch buffer buf
    auto a_store = [&](int buf, int pp, int tq, int o, f32x4g r) {
        const unsigned keep = o < 0 ? 0u : 0xffffffffu;                // (a mask, not a branch)
        lds_c* dst = L + (unsigned)(buf * abytes) + (__umul24((unsigned)pp, (unsigned)PIXW) + (unsigned)(tq * (DCS_ACT_IS_BF16 ? 4 : 2))) * 4u;
#if DCS_ACT_IS_BF16
        typedef unsigned nu4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) nu4 lds_u4;         // the stored bits ARE the MFMA operand
        const nu4 u = __builtin_bit_cast(nu4, r);
        *(lds_u4*)dst = nu4{u.x & keep, u.y & keep, u.z & keep, u.w & keep};
#else
        typedef unsigned nu2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) nu2 lds_u2;
        float x = __uint_as_float(__float_as_uint(r.x) & keep), y = __uint_as_float(__float_as_uint(r.y) & keep);
        float z = __uint_as_float(__float_as_uint(r.z) & keep), w = __uint_as_float(__float_as_uint(r.w) & keep);
        nu2 h0, h1, h2;                                                // 2 complex -> 3 planes of 4 bf16 (exact split)
        if (DCS_RING_EXP & 256) {                                      // (probe 256: the raw bits stored, no split)
            *(lds_u2*)dst = nu2{__float_as_uint(x), __float_as_uint(y)};
            *(lds_u2*)(dst + CH * 4) = nu2{__float_as_uint(z), __float_as_uint(w)};
            *(lds_u2*)(dst + 2 * CH * 4) = nu2{__float_as_uint(x), __float_as_uint(w)};
            return;
        }
        h0.x = dcs_split_pair(x, y); h0.y = dcs_split_pair(z, w);
        h1.x = dcs_split_pair(x, y); h1.y = dcs_split_pair(z, w);
        h2.x = dcs_pack_bf16x2(x, y); h2.y = dcs_pack_bf16x2(z, w);
        *(lds_u2*)dst = h0;
        *(lds_u2*)(dst + CH * 4) = h1;
        *(lds_u2*)(dst + 2 * CH * 4) = h2;
#endif
    };
    // chunk 0, by all 512 lanes: every load of a round in flight together (one memory round trip for the tiles of the train shapes)
    {
        constexpr int PPR8 = 512 / Q, GU0 = 4;
        const int tq = t % Q, tp0 = t / Q;
        const unsigned cb = (unsigned)(SB * tq) * (unsigned)sizeof(act2_t);
        for (int pb = tp0; pb < npix; pb += GU0 * PPR8) {
            int o[GU0], pp[GU0];
            f32x4g v[GU0];
#pragma unroll
            for (int u = 0; u < GU0; ++u) {
                pp[u] = pb + u * PPR8 < npix ? pb + u * PPR8 : npix - 1;   // (past the end: the last pixel again — same data, same place)
                o[u] = spx[pp[u]];
            }
#pragma unroll
            for (int u = 0; u < GU0; ++u) {
                unsigned vo = (unsigned)(o[u] < 0 ? 0 : o[u]) + cb;
                asm volatile("" : "+v"(vo));
                v[u] = *(__attribute__((address_space(1))) const f32x4g*)(xs1 + vo);
            }
#pragma unroll
            for (int u = 0; u < GU0; ++u) a_store(0, pp[u], tq, o[u], v[u]);
        }
    }

    if (DCS_RING_EXP & 64) return;                                     // (probe 64: launch + table + the first chunk's gather)
    f32x16 acc[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    if (wave >= 4) {
        // ============================================ producers ============================================
        // Every vector-memory instruction of a producer is hand-counted inline asm: hipcc waits vmcnt(0) for an ordinary load
        // beside an LDS-DMA in flight, which would drain the B requests of the steps ahead at every patch slice.
        __builtin_amdgcn_s_setprio(DCS_RING_PRIO_PROD);
        const int p = wave - 4, pt = p * 64 + lane;
        constexpr int PPR = 256 / Q;                                   // patch pixels per pass of the 256 producer lanes
        const int tq = pt % Q, tp0 = pt / Q;
        constexpr int slice = NA * PPR;                                // patch pixels per step

        // ---- B: the panel fragments of one step, by LDS-DMA.  Wave p: column tile jb = p & 1, the items of parity p >> 1, all planes.
        typedef __attribute__((address_space(1))) const char gpanel_t;
        gpanel_t* panel = (gpanel_t*)(m.bm + k.bm_off) + ((int)by * 2 + (p & 1)) * 1024;
        asm volatile("" : "+s"(panel));
        const unsigned tap_stride_b = (unsigned)(Cin / 8) * m.NT * 1024u;      // (32-bit: the launcher checks the panel's extent)
        const unsigned plane_stride_b = (unsigned)ntaps * tap_stride_b;
        const unsigned kg_stride_b = (unsigned)m.NT * 1024u;
        const unsigned lane16 = (unsigned)lane * 16u;
        const unsigned ring0 = (unsigned)(2 * abytes) + (unsigned)(p & 1) * 1024u;
        constexpr int NBW = (IPS / 2) * NP;                            // DMA instructions per wave and step
        auto issue_b = [&](int bc, int bjs, int slot) {
#pragma unroll
            for (int i2 = 0; i2 < IPS / 2; ++i2) {
                const int it = 2 * i2 + (p >> 1);                      // (scalar)
                const unsigned so = (unsigned)(bjs * TPS + it / U) * tap_stride_b + (unsigned)(bc * U + it % U) * kg_stride_b;
                const unsigned ldst = ring0 + (unsigned)slot * STAGE + (unsigned)it * (NP * 2048);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    const unsigned vo = lane16 + (so + pl * plane_stride_b);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 :: "v"(vo), "s"(panel), "s"(ldst + pl * 2048u) : "memory");
                }
            }
        };
        // ---- A: one slice = NA passes of the 256 producer lanes (patch pixels pb, pb + PPR, ...) of chunk chx
        f32x4g av[NA];
        int ak[NA];
        auto a_issue = [&](int chx, int pb) {
            const bool first = chx * CH < C1;
            gsrc_t* xb = first ? xs1 : xs2;
            const int* tb = first ? spx : spx2;
            const unsigned cb = (unsigned)((first ? chx * CH : chx * CH - C1) + SB * tq) * (unsigned)sizeof(act2_t);
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                int pp = pb + u * PPR;
                pp = pp < npix ? pp : npix - 1;
                ak[u] = tb[pp];
            }
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const unsigned vo = (unsigned)(ak[u] < 0 ? 0 : ak[u]) + cb;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(av[u]) : "v"(vo), "s"(xb) : "memory");
            }
        };
        auto a_finish = [&](int buf, int pb) {
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                int pp = pb + u * PPR;
                pp = pp < npix ? pp : npix - 1;
                a_store(buf, pp, tq, ak[u], av[u]);
            }
        };
        // the wait that retires a slice's loads (and every older request): all but the NBW DMA requests issued behind them.
        // ONE statement on every path (the tail trips re-request the last stage into its own slot — identical bytes — so that the
        // count never changes): two variants under a branch made the destination registers a phi, and hipcc placed the phi's copies
        // in front of the wait in one of the branches.
        auto a_wait = [&]() {
            constexpr int CNT = (DCS_RING_EXP & 2) ? 0 : NBW;
            if constexpr (NA == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(av[0]), "+v"(av[1]) : "n"(CNT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]) : "n"(CNT) : "memory");
        };

        // prologue: B stages 0 .. D - 1, the first slice of chunk 1 (written during step 0)
        const int D = rp.R - 1;
        int bc = 0, bjs = 0, bslot = 0, bst = 0;                       // the next B stage to request: (chunk, step in chunk), slot, index
        auto b_next = [&]() {
            issue_b(bc, bjs, bslot);
            if (bst + 1 < nsteps) {                                    // (the last stage is requested again and again: see a_wait)
                ++bst;
                if (++bjs == SPC) { bjs = 0; ++bc; }
                if (++bslot == rp.R) bslot = 0;
            }
        };
        for (int st = 0; st < D && st < nsteps; ++st) b_next();
        // slice j of chunk c + 1 is written at the start of step c SPC + j (j <= SPC - 2) and requested one step earlier.
        // (the loads are issued UNCONDITIONALLY — where no slice is due, of a valid address nobody uses — so that their
        // destination registers are defined in the trip that waits for them: a conditional definition makes them loop-carried
        // and hipcc copies loop-carried registers wherever it likes, also in front of the hand-written wait)
        bool have = n_chunks > 1 && !(DCS_RING_EXP & 1);
        a_issue(n_chunks > 1 ? 1 : 0, tp0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // chunk 0 stored; stages 0 .. D - 1 landed
        RING_BARRIER();                                                // B_0
        a_wait();
        if (have) a_finish(1, tp0);
        int cs = 0, js = 0;                                            // position of step s
        for (int s = 0; s < nsteps; ++s) {
            // the slice step s + 1 writes: position (ncs, njs), chunk ncs + 1
            int ncs = cs, njs = js + 1;
            if (njs == SPC) { njs = 0; ++ncs; }
            have = njs <= SPC - 2 && ncs + 1 < n_chunks && s + 1 < nsteps && !(DCS_RING_EXP & 1);
            a_issue(ncs + 1 < n_chunks ? ncs + 1 : n_chunks - 1, have ? tp0 + njs * slice : tp0);
            if (!(DCS_RING_EXP & 2)) b_next();                         // stage s + D (or the last one again)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the patch stores of the slice finished in the previous trip
            if (!(DCS_RING_EXP & 8)) RING_BARRIER();                   // B_{s+1}
            // behind the barrier: retire this trip's slice (its loads are older than the NBW requests behind them) — the wait
            // also retires every B stage requested in EARLIER trips (<= s - 1 + D >= s + 3), in front of the next barrier
            a_wait();
            if (have) a_finish((ncs + 1) & 1, tp0 + njs * slice);
            cs = ncs; js = njs;
        }
    } else {
        // ============================================ consumers ============================================
        __builtin_amdgcn_s_setprio(DCS_RING_PRIO_CONS);
        const int wm = wave >> 1, wn = wave & 1;
        const int kk = lane >> 5, li = lane & 31;
        unsigned apix[WM];                                             // byte offset of this lane's pixel (tap (0,0), k-group 0) in a patch buffer
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int pi = (wm * WM + i) * 32 + li;
            apix[i] = (unsigned)(((((pi >> m.twshift)) * a.sf) * cols + ((pi & (m.TW - 1))) * a.st) * PIXW + kk * 4) * 4u;
        }
        const unsigned bpix = (unsigned)(2 * abytes + wn * 1024 + lane * 16);
        bf16x8 fa[2][NP][WM], fb[2][NP];
        // fragments of item `it` of a step: patch buffer ab (byte offset), its TPS taps' word offsets tw[], ring slot byte offset bs
        auto rd = [&](int set, unsigned ab, const int* tw, unsigned bs, int it) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    fa[set][pl][i] = *(lds_bf8_t*)(L + (apix[i] + ab + (unsigned)tw[it / U] + (unsigned)((pl * CH + (it % U) * 8) * 4)));
                fb[set][pl] = *(lds_bf8_t*)(L + (bpix + bs + (unsigned)((it * NP + pl) * 2048)));
            }
        };
        // step state: current and next
        unsigned c_ab = 0, c_bs = 0, n_ab, n_bs;
        int c_tw[TPS], n_tw[TPS];
        int tqx = 0, ttx = 0;                                          // running tap: patch pixel offset, column
        auto taps_of = [&](int* tw) {
#pragma unroll
            for (int e = 0; e < TPS; ++e) {
                tw[e] = tqx * PIXW * 4;
                if (++ttx == kw) { ttx = 0; tqx += cols - (kw - 1); } else ++tqx;
            }
        };
        taps_of(c_tw);
        int js = 0, slot = 0;
        RING_BARRIER();                                                // B_0: prologue data is in place
        rd(0, c_ab, c_tw, c_bs, 0);
        for (int s = 0; s < nsteps; ++s) {
#pragma unroll
            for (int it = 0; it < IPS; ++it) {
                if (DCS_RING_EXP & 4) {
                } else if (it + 1 < IPS) rd((it + 1) & 1, c_ab, c_tw, c_bs, it + 1);
                else rd((it + 1) & 1, n_ab, n_tw, n_bs, 0);
                if (PR == 2) {                                          // a0 b2, a1 b1, a2 b0, a0 b1, a1 b0, a0 b0 (conv_mfma.hip's order)
                    constexpr int pa[6] = {0, 1, 2, 0, 1, 0}, pb[6] = {2, 1, 0, 1, 0, 0};
#if DCS_RING_ALT_ACC
#pragma unroll
                    for (int e = 0; e < 6; ++e)                         // the two accumulators alternate (each keeps its own term order)
#pragma unroll
                        for (int i = 0; i < WM; ++i)
#else
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int e = 0; e < 6; ++e)
#endif
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[it & 1][pa[e] < NP ? pa[e] : 0][i],
                                                                             fb[it & 1][pb[e] < NP ? pb[e] : 0], acc[i], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < WM; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[it & 1][0][i], fb[it & 1][0], acc[i], 0, 0, 0);
                }
                if (it == 0) {
                    // the state of step s + 1, computed HERE — scalar instructions that issue in the shadow of item 0's MFMAs
                    // (in front of the step they were ~40 scalar instructions during which the matrix pipe idled: the MFMA-only
                    // probe ran at 81 % of the pipe) — and branch-free, so that they stay inside this scheduling region.  Behind the
                    // last step it describes a valid, unused address (the other patch buffer, tap 0, the next slot).
                    const bool wrap = js + 1 == SPC;
                    js = wrap ? 0 : js + 1;
                    tqx = wrap ? 0 : tqx; ttx = wrap ? 0 : ttx;
                    n_ab = wrap ? c_ab ^ (unsigned)abytes : c_ab;
                    slot = slot + 1 == rp.R ? 0 : slot + 1;
                    n_bs = (unsigned)slot * STAGE;
#pragma unroll
                    for (int e = 0; e < TPS; ++e) {
                        n_tw[e] = tqx * PIXW * 4;
                        const bool eol = ttx + 1 == kw;
                        tqx = eol ? tqx + cols - (kw - 1) : tqx + 1;
                        ttx = eol ? 0 : ttx + 1;
                    }
                }
                {   // one LDS read behind each MFMA (the next item's fragments), instead of a burst in front of them
                    constexpr int NM = WM * (PR == 2 ? 6 : 1), NR = NP * WM + NP;
#pragma unroll
                    for (int e = 0; e < NM; ++e) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (e < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    if (NM < NR) __builtin_amdgcn_sched_group_barrier(0x100, NR - NM, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            c_ab = n_ab; c_bs = n_bs;
#pragma unroll
            for (int e = 0; e < TPS; ++e) c_tw[e] = n_tw[e];
            if (!(DCS_RING_EXP & 8)) RING_BARRIER();                   // B_{s+1}
        }
    }
    __builtin_amdgcn_s_setprio(0);

    // ============================================ epilogue ============================================
    // (behind the last barrier every wave is done with the patches and the ring)  As cconv_mfma_kernel's: each consumer wave
    // transposes its 32x32 tiles through its own 4.5 KB of LDS and stores float4 rows; bias, folded eval-mode CBN, activation,
    // the cat split of a data gradient, the CBN statistics of the raw output.
    constexpr int TP = 36;
    const bool cons = wave < 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int kk = lane >> 5, li = lane & 31;
    float* const tsm = lds + (wave & 3) * 32 * TP;
    const int c4 = lane & 7, r8 = lane >> 3;
    const float* biasf = reinterpret_cast<const float*>(a.bias);
    float sst[10];
#pragma unroll
    for (int e = 0; e < 10; ++e) sst[e] = 0.f;
    if (cons) {
        const int nt0 = (int)by * 2 + wn;
        const int n0 = nt0 * 32 + 4 * c4;                              // this lane's first column after the transpose
        const bool second = m.y2 != nullptr && n0 >= m.nsplit;         // columns >= nsplit: the second tensor of a concatenation
        const int width = m.y2 == nullptr ? m.N : (second ? m.N - m.nsplit : m.nsplit);
        const int col = second ? n0 - m.nsplit : n0;
        act_t* const yb = (second ? m.y2 : reinterpret_cast<act_t*>(a.y)) + (long)b * a.Hout * a.Wout * width;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        float q[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) q[e] = 0.f;
        if (n0 < m.N) {
            if (biasf) bv = *reinterpret_cast<const float4*>(biasf + n0);
            if (a.coef) {
#pragma unroll
                for (int e = 0; e < 12; ++e) q[e] = a.coef[6 * (n0 >> 1) + e];
            }
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tsm[((r & 3) + 8 * (r >> 2) + 4 * kk) * TP + li] = acc[i][r];
            // (a wave's LDS operations complete in order: no barrier between its own writes and reads)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int row = r8 + 8 * qq;
                float4 v = *reinterpret_cast<const float4*>(tsm + row * TP + 4 * c4);
                const int pi = (wm * WM + i) * 32 + row;
                const int oy = oy0 + (pi >> m.twshift), ox = ox0 + (pi & (m.TW - 1));
                if (STAT && n0 < m.N && oy < k.Hc && ox < k.Wc) {      // moments of the UN-biased value (pivot = bias)
                    sst[0] += v.x; sst[1] += v.y;
                    sst[2] = fmaf(v.x, v.x, sst[2]); sst[3] = fmaf(v.y, v.y, sst[3]); sst[4] = fmaf(v.x, v.y, sst[4]);
                    sst[5] += v.z; sst[6] += v.w;
                    sst[7] = fmaf(v.z, v.z, sst[7]); sst[8] = fmaf(v.w, v.w, sst[8]); sst[9] = fmaf(v.z, v.w, sst[9]);
                }
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (a.coef) {
                    // scalar FMAs kept apart (the packed-FMA op_sel erratum beside bf16 MFMAs: profiles/r03_pk_fma_op_sel_hazard.txt);
                    // association order of cbn_apply_kernel: the folded and the two-kernel forms agree bit for bit
                    const float4 u = v;
                    v.x = fmaf(q[0], u.x, fmaf(q[1], u.y, q[4]));
                    asm volatile("" : "+v"(v.x));
                    v.y = fmaf(q[2], u.x, fmaf(q[3], u.y, q[5]));
                    asm volatile("" : "+v"(v.y));
                    v.z = fmaf(q[6], u.z, fmaf(q[7], u.w, q[10]));
                    asm volatile("" : "+v"(v.z));
                    v.w = fmaf(q[8], u.z, fmaf(q[9], u.w, q[11]));
                }
                v.x = dcs_act(v.x, a.act); v.y = dcs_act(v.y, a.act); v.z = dcs_act(v.z, a.act); v.w = dcs_act(v.w, a.act);
                if (n0 < m.N && oy < k.Hc && ox < k.Wc) {              // 32-bit offsets inside one image (the launcher checks the extent)
                    const int off = ((oy * m.os_f + k.oo_f) * a.Wout + ox * m.os_t + k.oo_t) * width + col;
                    dcs_st4(yb + off, v);
                }
            }
        }
    }
    if (STAT) {
        // lanes c4 + 8 r8 hold the same two channels: sum over r8 through the wave's own transpose tile (in-order LDS), then over
        // the two waves that share a column tile (wm = 0, 1) through `comb`: one row of partial sums per workgroup
        float* const comb = lds + 4 * 32 * TP;                         // [consumer wave][80]
        if (cons) {
#pragma unroll
            for (int e = 0; e < 10; ++e) tsm[e * 64 + lane] = sst[e];
            float r0 = 0.f, r1 = 0.f;
            const int k5 = lane >> 3;
            if (lane < 40) {
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) {
                    r0 += tsm[k5 * 64 + c4 + 8 * q8];
                    r1 += tsm[(5 + k5) * 64 + c4 + 8 * q8];
                }
                comb[wave * 80 + lane] = r0;
                comb[wave * 80 + 40 + lane] = r1;
            }
        }
        __syncthreads();
        for (int o = t; o < 2 * 80; o += 512) {
            const int ct = o / 80, e = o % 80;                         // ct = wn
            const float sum = comb[ct * 80 + e] + comb[(2 + ct) * 80 + e];       // wm = 0, then wm = 1 (cconv_mfma_kernel's order)
            const int c = (((int)by * 2 + ct) * 32 + 4 * (e & 7)) / 2 + e / 40;
            if (c < a.Cout) stat_row[(long)(c * 5 + ((e % 40) >> 3)) * a.stat_stride] = sum;
        }
    }
}

template <int PR, int CH, int TPS, int NA>
int launch_ring_na(MArgs& m, const RingPlan& rp_, hipStream_t stream) {
    const conv::Args& a = m.c;
    RingP rp;
    rp.R = rp_.R; rp.NA = rp_.NA; rp.abytes = rp_.abytes;
    const size_t lds = (size_t)rp_.lds_bytes;
    dim3 grid(a.tiles_w * a.tiles_h * a.B, m.NT / 2, m.ncls);
    if (grid.y > 65535) return DCS_ERR_BADARG;
    if (m.c.stat != nullptr) {
        auto fn = cconv_ring_kernel<PR, CH, TPS, NA, true>;
        if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
        DCS_LAUNCH(fn, grid, dim3(512), lds, stream, m, rp);
    } else {
        auto fn = cconv_ring_kernel<PR, CH, TPS, NA, false>;
        if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
        DCS_LAUNCH(fn, grid, dim3(512), lds, stream, m, rp);
    }
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

template <int PR, int CH, int TPS>
int launch_ring(MArgs& m, const RingPlan& rp, hipStream_t stream) {
    return rp.NA <= 2 ? launch_ring_na<PR, CH, TPS, 2>(m, rp, stream) : launch_ring_na<PR, CH, TPS, 4>(m, rp, stream);
}

}  // namespace

// Does the geometry (a, classes) run on the ring kernel, and how: tile 128 pixels x 64 columns, chunk depth CH, taps per step
// TPS, ring depth R, LDS bytes.  pr: the precision mode the panel was packed for (conv::mfma_precision).
bool DCS_SYM(dcs_conv_ring_plan)(const conv::Args& a, int ncls, const conv::Cls* cls, int pr, RingPlan* rp) {
    // OPT-IN (DCS_CONV_RING=1): measured at parity with or behind cconv_mfma_kernel on every layer of both networks at the
    // train, inference and bf16-storage shapes (profiles/r05_ring_kernel.txt) — it stays as the tested record
    // of that experiment.  (read per call, not cached: tools/ring_check.py and the tests switch the path inside one process)
    const char* const e_on = getenv("DCS_CONV_RING");
    const char* const e_min = getenv("DCS_RING_MIN_WG");
    if (!e_on || atoi(e_on) == 0) return false;
    const long min_wg = e_min ? atol(e_min) : 192L;
    const int Cin = a.C1 + a.C2, N = 2 * a.Cout;
#if DCS_ACT_IS_BF16
    if (pr != 1) return false;
#else
    if (pr != 2) return false;
#endif
    if (!conv::mfma_ok(Cin, a.Cout) || (a.C1 & 1) || (N % 64) != 0 || ncls < 1 || ncls > 4) return false;
    int Hc = 0, Wc = 0, kh = 0, kw = 0;
    for (int c = 0; c < ncls; ++c) {
        Hc = cls[c].Hc > Hc ? cls[c].Hc : Hc; Wc = cls[c].Wc > Wc ? cls[c].Wc : Wc;
        kh = cls[c].kh > kh ? cls[c].kh : kh; kw = cls[c].kw > kw ? cls[c].kw : kw;
    }
    if (Hc <= 0 || Wc <= 0) return false;
    // tile shape: 128 pixels, least padding past the class extent, then the smallest patch; rows of 32 keep the A reads of a
    // 32-lane half on consecutive patch pixels (pixel pitch = 13 or 7 x 16 bytes: conflict-free)
    int TH = 0, TW = 0;
    {
        long best = -1;
        for (int h = 2; h <= 16; h *= 2) {
            const int w = 128 / h;
            const long padded = (long)((Hc + h - 1) / h) * h * ((Wc + w - 1) / w) * w;
            const long patch = (long)((h - 1) * a.sf + kh) * ((w - 1) * a.st + kw);
            const long cost = padded * 4096 + patch;
            if (best < 0 || cost < best) { best = cost; TH = h; TW = w; }
        }
    }
    const long ty = (Hc + TH - 1) / TH, tx = (Wc + TW - 1) / TW;
    const double eff = (double)Hc * Wc / ((double)ty * TH * tx * TW);
    const long wgs = ty * tx * a.B * (N / 64) * ncls;
    if (wgs * eff < min_wg || eff < 0.7) return false;
    const long npix = (long)((TH - 1) * a.sf + kh) * ((TW - 1) * a.st + kw);
    const int np = pr == 2 ? 3 : 1;
    // (chunk depth, taps per step) instances compiled below, deepest first
#if DCS_ACT_IS_BF16
    const int cand[][2] = {{32, 1}, {32, 2}, {16, 1}, {16, 2}};
#else
    const int cand[][2] = {{16, 1}, {8, 2}};
#endif
    for (const auto& cd : cand) {
        const int CH = cd[0], TPS = cd[1];
        if (Cin % CH != 0 || (a.C2 > 0 && a.C1 % CH != 0)) continue;
        bool ok = true;
        int spc_min = 1 << 30;
        for (int c = 0; c < ncls; ++c) {
            const int nt = cls[c].kh * cls[c].kw;
            if (nt % TPS != 0 || nt / TPS < 2) ok = false;
            spc_min = nt / TPS < spc_min ? nt / TPS : spc_min;
        }
        if (!ok) continue;
        const long abytes = ((npix * (np * CH + 4) * 4 + 15) / 16) * 16;
        const long stage = (long)TPS * (CH / 8) * np * 2 * 1024;
        const long tables = npix * 8 + 64;
        long R = (160L * 1024 - 2 * abytes - tables) / stage;
        if (R > 8) R = 8;
        if (R < 5) continue;                                   // D = R - 1 >= 4 stages ahead (the producers' counted wait)
        const int sb = 16 / (int)sizeof(act2_t), Q = CH / sb, PPR = 256 / Q;
        const long NA = (npix + (long)PPR * (spc_min - 1) - 1) / ((long)PPR * (spc_min - 1));
        if (NA > 4) continue;
        if ((long)ncls * kh * kw * (Cin / 8) * (N / 32) * 1024 * np >= (1L << 31)) continue;     // 32-bit panel offsets
        long lds = 2 * abytes + R * stage + tables;
        const long epi = (4 * 32 * 36 + 4 * 80) * 4;
        if (lds < epi) lds = epi;
        rp->TH = TH; rp->TW = TW; rp->CH = CH; rp->TPS = TPS; rp->R = (int)R; rp->NA = NA <= 2 ? 2 : 4;
        rp->abytes = (int)abytes; rp->lds_bytes = lds; rp->npix = npix; rp->wgs = wgs;
        return true;
    }
    return false;
}

int DCS_SYM(dcs_conv_ring_launch)(MArgs& m, const RingPlan& rp, hipStream_t stream) {
#if DCS_ACT_IS_BF16
    if (rp.CH == 32 && rp.TPS == 1) return launch_ring<1, 32, 1>(m, rp, stream);
    if (rp.CH == 32 && rp.TPS == 2) return launch_ring<1, 32, 2>(m, rp, stream);
    if (rp.CH == 16 && rp.TPS == 1) return launch_ring<1, 16, 1>(m, rp, stream);
    if (rp.CH == 16 && rp.TPS == 2) return launch_ring<1, 16, 2>(m, rp, stream);
#else
    if (rp.CH == 16 && rp.TPS == 1) return launch_ring<2, 16, 1>(m, rp, stream);
    if (rp.CH == 8 && rp.TPS == 2) return launch_ring<2, 8, 2>(m, rp, stream);
#endif
    return DCS_ERR_BADARG;
}
