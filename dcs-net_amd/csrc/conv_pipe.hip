// conv_pipe.hip — the fp32 MFMA implicit-GEMM complex convolution of conv_mfma.hip as a PERSISTENT, software-pipelined
// kernel: the same GEMM, the same B panels, the same accumulation order (so results are bit-identical), a different
// schedule.
//
// conv_mfma.hip runs one output tile per workgroup: barrier -> every thread gathers its share of the input patch chunk
// (global_load -> VGPR -> ds_write) -> barrier -> MFMA tap loop, and relies on 2-3 co-resident workgroups per CU to cover
// each other's gather phases.  Counters (round 1, SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES = 1.65-2.58 of 4) show the
// matrix pipe idle 35-60 % of the time: the co-resident workgroups fall into lockstep and, with two waves per SIMD, each
// wave's MFMA phase runs at half rate while its partner's does.
//
// Here a workgroup (4 waves, as there) lives for the whole launch and walks a sequence of work items = (output tile,
// K slice, channel chunk).  The patch is double-buffered in LDS and filled by LDS-DMA (global_load_lds_dwordx4: no VGPR
// staging, no ds_write pass): right after the barrier that opens item i every wave issues its quarter of item i+1's DMA
// pieces into the OTHER buffer and goes on with item i's MFMA tap loop; the pieces are in flight under the whole loop.
// One `s_waitcnt vmcnt(0)` + one `s_barrier` per item is the only synchronisation (the guide's minimal two-phase loop):
// behind it item i+1's patch has landed for every wave and everybody is done reading item i's.  Workgroups are persistent
// (grid = resident workgroups, units dealt round-robin), so the pipeline also runs across tile boundaries: the next tile's
// first chunk streams in under the current tile's last chunk.
// (First form tried: a fifth, dedicated loader wave.  Beside an MFMA wave on its SIMD it issued one DMA piece per
// 140-170 cycles — 40 alone, tools/micro/dma_rate.hip — and its address arithmetic one instruction per MFMA; at
// s_setprio 3 it reached parity with conv_mfma.hip, no more.)
//
// LDS image.  An LDS-DMA instruction writes wave-uniform base + lane * 16 B: 1 KiB contiguous, so the pixel pitch
// cannot be padded (conv_mfma.hip's conflict-free pitch of 2*CH + 4 floats).  Instead
//   * pixels are CH*8 B = SP 16-byte slots; slot s of pixel slot q is stored at slot s ^ f(q),
//     f(q) = (q / (16 / SP)) % SP — the swizzle is applied to the per-lane SOURCE address (the DMA destination
//     stays linear) and again on the fragment read: a 16-lane ds_read_b128 group then spreads over all 16 slots
//     of the 256-B bank row for consecutive q (2-way at worst across tile rows; LDS is ~12 % utilised here);
//   * for a stride-2 convolution the patch columns are stored as two parity planes per row (even columns, then
//     odd columns): the 32 pixels a wave reads for one tap are then consecutive pixel slots again.
// Padding / out-of-image pixels read a 64-byte zero page.
#include "conv_mfma_args.h"
#include <cstdlib>
#include <cstdio>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ float4 g_zero_page[4];          // zero-initialised: the DMA source of every padding pixel

struct PArgs {
    MArgs m;
    int n_units;               // tiles x batch x column groups x K slices x classes
    int buf_floats;            // one patch buffer (whole 1-KiB DMA pieces)
    int npw_max;               // DMA pieces per wave and item, at most
    long long* dbg;            // diagnostic builds only (-DDCS_PIPE_DIAG): per-wave cycle sums
};

#ifdef DCS_PIPE_DIAG
#define DIAG_NOW() ((long long)__builtin_amdgcn_s_memtime())
#else
#define DIAG_NOW() 0LL
#endif

struct Unit { int cls, b, oy0, ox0, yy, ks, c0, c1; };

constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }

#define DCS_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define DCS_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int WAVES_N, int WM, int WN, int CH, int TPI>
__global__ __launch_bounds__(256) void cconv_pipe_kernel(PArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int U = CH / 4, VU = U * TPI, SP = CH / 2, LOG_SP = ilog2c(SP), PRSH = ilog2c(16 / SP);
    const MArgs& m = pa.m;
    const conv::Args& a = m.c;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    float* const buf0 = lds;
    float* const buf1 = lds + pa.buf_floats;
    // wave-private: source pixel (or -1) of the 64 slots of each DMA piece this wave issues (pieces wave, wave + 4, ...)
    int* const mytab = reinterpret_cast<int*>(lds + 2 * pa.buf_floats) + wave * pa.npw_max * 64 + lane;

    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int nx = tiles_per_img * a.B;
    const int ny = m.NT / (WAVES_N * WN), nyk = ny * m.ksplit;
    const int Cin = a.C1 + a.C2, nch_all = Cin / CH;
    const int n_units = pa.n_units;
    const bool st2 = a.st == 2;

    // unit u -> (class, image, tile, column group, K slice, chunk range); false: the tile lies outside this (smaller) class
    auto decode = [&](int u, Unit* o) -> bool {
        const int x = u % nx, r = u / nx;
        const int y = r % nyk;
        o->cls = r / nyk;
        o->b = x / tiles_per_img;
        const int tid = x % tiles_per_img;
        o->oy0 = (tid / a.tiles_w) * m.TH;
        o->ox0 = (tid % a.tiles_w) * m.TW;
        const conv::Cls& k = m.cls[o->cls];
        if (o->oy0 >= k.Hc || o->ox0 >= k.Wc) return false;
        o->ks = y / ny;
        o->yy = y % ny;
        o->c0 = o->ks * m.cps;
        o->c1 = o->c0 + m.cps < nch_all ? o->c0 + m.cps : nch_all;
        return o->c0 < o->c1;
    };

    // ---- loading half: every wave issues its quarter of an item's DMA pieces ----------------------------------
    // table of unit q: for each of this wave's pieces, the source pixel of this lane's 16-byte slot
    auto build_table = [&](const Unit& q) {
        const conv::Cls& k = m.cls[q.cls];
        const int P = (m.TW - 1) * a.st + k.kw, rows = (m.TH - 1) * a.sf + k.kh, PW0 = (P + 1) >> 1;
        const int NQ = rows * P, NP = (NQ * SP + 63) >> 6;
        const int vy0 = q.oy0 * a.sf - k.pad_f, vx0 = q.ox0 * a.st - k.pad_t;
        constexpr int DQ = 256 >> LOG_SP;                                // pixel slots between two pieces of one wave
        const int drow = DQ / P, dr = DQ - drow * P;
        int ql = (wave * 64 + lane) >> LOG_SP;
        int row = (int)(((float)ql + 0.5f) * (1.0f / (float)P));         // exact: ql, P < 2^12
        int r = ql - row * P;
        int* tp = mytab;
        for (int p = wave; p < NP; p += 4) {
            const int col = st2 ? (r >= PW0 ? 2 * (r - PW0) + 1 : 2 * r) : r;
            long sp;
            *tp = (ql < NQ && conv::src_pixel(a, q.b, vy0 + row, vx0 + col, &sp)) ? (int)sp : -1;
            tp += 64;
            ql += DQ; r += dr; row += drow;
            if (r >= P) { r -= P; ++row; }
        }
    };
    auto issue_item = [&](const Unit& q, int ch, float* dst) {
        const conv::Cls& k = m.cls[q.cls];
        const int P = (m.TW - 1) * a.st + k.kw, rows = (m.TH - 1) * a.sf + k.kh;
        const int NP = (rows * P * SP + 63) >> 6;
        const int c = ch * CH;                                           // first complex channel of the chunk
        const bool second = c >= a.C1;
        const char* base = second ? reinterpret_cast<const char*>(a.x2 + (c - a.C1)) : reinterpret_cast<const char*>(a.x1 + c);
        const long Cb = (long)(second ? a.C2 : a.C1) * 8;                // bytes per source pixel
        const int* tp = mytab;
        // pieces in pairs: both table reads first (one LDS round trip), then both DMAs
        for (int p = wave; p < NP; p += 8) {
            const int sp0 = tp[0];
            const int sp1 = p + 4 < NP ? tp[64] : -1;
            tp += 128;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int pe = p + 4 * e;
                if (pe >= NP) break;
                const int spv = e ? sp1 : sp0;
                const int L = pe * 64 + lane, ql = L >> LOG_SP;
                const int sl = (L & (SP - 1)) ^ ((ql >> PRSH) & (SP - 1));
                const char* g = spv >= 0 ? base + spv * Cb + sl * 16 : reinterpret_cast<const char*>(g_zero_page);
                __builtin_amdgcn_global_load_lds(DCS_GPTR(g), DCS_LPTR(dst + pe * 256), 16, 0, 0);
            }
        }
    };

    // ---- MFMA half ------------------------------------------------------------------------------------------------
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int kk = lane >> 5, li = lane & 31;
    const long b_tap_stride = (long)m.KG * m.NT * 256, b_kg_stride = (long)m.NT * 256;

    auto unit_b = [&](const Unit& q) {      // this lane's B fragments of (tap 0, first chunk of the slice)
        const int nt0 = (q.yy * WAVES_N + wn) * WN;
        return m.bm + m.cls[q.cls].bm_off + ((long)nt0 * 64 + lane) * 4 + (long)(q.c0 * U) * b_kg_stride;
    };
    auto bload = [&](float4* dst, const float* tap_ptr, int vg) {
        const float* bp = tap_ptr + (vg / U) * b_tap_stride + (vg % U) * b_kg_stride;
#pragma unroll
        for (int j = 0; j < WN; ++j) dst[j] = *reinterpret_cast<const float4*>(bp + j * 256);
    };

    Unit cu;
    int u = blockIdx.x;
    while (u < n_units && !decode(u, &cu)) u += gridDim.x;
    if (u >= n_units) return;                    // every wave of the workgroup takes the same decision

    f32x16 acc[WM][WN];
    constexpr int LPG = VU >= 2 ? 2 : 1;
    float4 bcur[VU][WN], bnxt[VU][WN];
    const float* bp_item = unit_b(cu);
#pragma unroll
    for (int g = 0; g < VU; ++g) bload(bcur[g], bp_item, g);
    // prologue: the first item's patch
    build_table(cu);
    issue_item(cu, cu.c0, buf0);
    int cnt = 0;
    long long d_bar = 0, d_comp = 0, d_epi = 0, d_iss = 0, d_start = DIAG_NOW();

    for (;;) {
        Unit nu;
        int u2 = u + gridDim.x;
        while (u2 < n_units && !decode(u2, &nu)) u2 += gridDim.x;
        const bool has_next = u2 < n_units;
        const float* bp_next_unit = has_next ? unit_b(nu) : bp_item;

        const conv::Cls& k = m.cls[cu.cls];
        const int ntaps = k.kh * k.kw;
        const int P = (m.TW - 1) * a.st + k.kw, PW0 = (P + 1) >> 1;
        int qb[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int pi = (wm * WM + i) * 32 + li;
            qb[i] = (pi >> m.twshift) * a.sf * P + (pi & (m.TW - 1));
        }
        // float offset of this lane's 16 bytes: pixel slot q, 16-B slot (2g + kk) ^ f(q)  ->  base + ((hh ^ 2g) << 2)
        auto lane_addr = [&](int i, int dy, int dx, int* base, int* hh) {
            const int q = qb[i] + dy * P + (st2 ? (dx & 1) * PW0 + (dx >> 1) : dx);
            *base = q * (SP * 4);
            *hh = ((q >> PRSH) & (SP - 1)) ^ kk;
        };
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int ch = cu.c0; ch < cu.c1; ++ch) {
            const bool last_chunk = ch + 1 >= cu.c1;
            const float* bp_next_item = last_chunk ? bp_next_unit : bp_item + (long)U * b_kg_stride;
            const float* patch = (cnt & 1) ? buf1 : buf0;
            float* other = (cnt & 1) ? buf0 : buf1;
            // this wave's DMAs of the item have landed; after the barrier so have everyone's, and every wave is done
            // reading the other buffer (the previous item's)
            const long long s0 = DIAG_NOW();
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            const long long s1 = DIAG_NOW();
            d_bar += s1 - s0;
            // the NEXT item's patch into the other buffer: in flight under this item's whole tap loop
            if (!last_chunk) issue_item(cu, ch + 1, other);
            else if (has_next) { build_table(nu); issue_item(nu, nu.c0, other); }
            d_iss += DIAG_NOW() - s1;
            int abase[TPI][WM], ah[TPI][WM], nbase[WM], nh[WM];
            int dy = 0, dx = 0;
#pragma unroll
            for (int d = 0; d < TPI; ++d)
#pragma unroll
                for (int i = 0; i < WM; ++i) lane_addr(i, 0, d, &abase[d][i], &ah[d][i]);
            float4 af[2][WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[0][i] = *reinterpret_cast<const float4*>(patch + abase[0][i] + (ah[0][i] << 2));
            const float* bp_tap = bp_item;
            for (int tap = 0; tap < ntaps; tap += TPI) {
                const bool last = tap + TPI >= ntaps;
                int ndy = dy, ndx = dx;
                if (!last) {
                    if (TPI == 1) { ndx = dx + 1; if (ndx == k.kw) { ndx = 0; ndy = dy + 1; } }
                    else ndy = dy + 1;
                }
#pragma unroll
                for (int i = 0; i < WM; ++i) lane_addr(i, ndy, ndx, &nbase[i], &nh[i]);
                const float* bp_ntap = last ? bp_next_item : bp_tap + TPI * b_tap_stride;
#pragma unroll
                for (int g = 0; g < VU; ++g) {
                    // A fragments of the next k-group into the other register set
#pragma unroll
                    for (int i = 0; i < WM; ++i) {
                        const int off = g + 1 < VU ? abase[(g + 1) / U][i] + ((ah[(g + 1) / U][i] ^ (2 * ((g + 1) % U))) << 2)
                                                   : nbase[i] + (nh[i] << 2);
                        af[(g + 1) & 1][i] = *reinterpret_cast<const float4*>(patch + off);
                    }
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) {
                            const float4 av = af[g & 1][i], bv = bcur[g][j];
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i][j], 0, 0, 0);
                        }
                    // the next iteration's B fragments, issued ahead of the remaining MFMA groups
                    __builtin_amdgcn_sched_barrier(0);
                    if (g * LPG < VU) {
#pragma unroll
                        for (int q = 0; q < LPG; ++q)
                            if (g * LPG + q < VU) bload(bnxt[g * LPG + q], bp_ntap, g * LPG + q);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int g = 0; g < VU; ++g)
#pragma unroll
                    for (int j = 0; j < WN; ++j) bcur[g][j] = bnxt[g][j];
                dy = ndy; dx = ndx; bp_tap = bp_ntap;
                if (TPI == 1) {
#pragma unroll
                    for (int i = 0; i < WM; ++i) { abase[0][i] = nbase[i]; ah[0][i] = nh[i]; }
                } else {
#pragma unroll
                    for (int d = 0; d < TPI; ++d)
#pragma unroll
                        for (int i = 0; i < WM; ++i) lane_addr(i, dy, d, &abase[d][i], &ah[d][i]);
                }
            }
            bp_item = bp_next_item;
            ++cnt;
            d_comp += DIAG_NOW() - s1;
        }
        const long long s5 = DIAG_NOW();

        // epilogue (conv_mfma.hip's): C/D map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const int nt0 = (cu.yy * WAVES_N + wn) * WN;
        const int b = cu.b, oy0 = cu.oy0, ox0 = cu.ox0;
        if (m.ksplit > 1) {
            float* pfb = m.part + (long)cu.ks * m.slab_floats + (long)b * a.Hout * a.Wout * m.N;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = (nt0 + j) * 32 + li;
                if (n >= m.N) continue;
#pragma unroll
                for (int i = 0; i < WM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
                        const int pi = (wm * WM + i) * 32 + row;
                        const int oy = oy0 + (pi >> m.twshift), ox = ox0 + (pi & (m.TW - 1));
                        if (oy < k.Hc && ox < k.Wc)
                            pfb[((oy * m.os_f + k.oo_f) * a.Wout + ox * m.os_t + k.oo_t) * m.N + n] = acc[i][j][r];
                    }
                }
            }
        } else {
            const float* biasf = reinterpret_cast<const float*>(a.bias);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = (nt0 + j) * 32 + li;
                if (n >= m.N) continue;
                const float bv = biasf ? biasf[n] : 0.f;
                float c_re = 1.f, c_im = 0.f, c_add = 0.f;
                if (a.coef) {
                    const float* q = a.coef + 6 * (n >> 1);
                    if (n & 1) { c_re = q[2]; c_im = q[3]; c_add = q[5]; } else { c_re = q[0]; c_im = q[1]; c_add = q[4]; }
                }
                const bool second = m.y2 != nullptr && n >= m.nsplit;
                float* yf = second ? m.y2 : reinterpret_cast<float*>(a.y);
                const int width = m.y2 == nullptr ? m.N : (second ? m.N - m.nsplit : m.nsplit);
                const int col = second ? n - m.nsplit : n;
                float* yb = yf + (long)b * a.Hout * a.Wout * width;
#pragma unroll
                for (int i = 0; i < WM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
                        const int pi = (wm * WM + i) * 32 + row;
                        const int oy = oy0 + (pi >> m.twshift), ox = ox0 + (pi & (m.TW - 1));
                        float v = acc[i][j][r] + bv;
                        if (a.coef) {
                            const float pv = dcs_dpp_term<0xB1, 0xf>(v);
                            v = (n & 1) ? fmaf(c_re, pv, fmaf(c_im, v, c_add)) : fmaf(c_re, v, fmaf(c_im, pv, c_add));
                        }
                        if (oy < k.Hc && ox < k.Wc)
                            yb[((oy * m.os_f + k.oo_f) * a.Wout + ox * m.os_t + k.oo_t) * width + col] = dcs_act(v, a.act);
                    }
                }
            }
        }
        d_epi += DIAG_NOW() - s5;
        if (!has_next) break;
        cu = nu;
        u = u2;
    }
#ifdef DCS_PIPE_DIAG
    if (pa.dbg && lane == 0) {
        long long* d = pa.dbg + ((long)blockIdx.x * 5 + wave) * 8;
        d[0] = d_bar; d[1] = d_comp; d[2] = d_epi; d[3] = d_iss; d[4] = DIAG_NOW() - d_start; d[5] = cnt;
    }
#endif
}

struct Geo { int npmax, nqmax; };

// largest patch over the classes, in pixel slots and whole DMA pieces
Geo pipe_geo(const conv::Args& a, int ncls, const conv::Cls* cls, int TH, int TW, int CH) {
    Geo g{0, 0};
    for (int c = 0; c < ncls; ++c) {
        const int cols = (TW - 1) * a.st + cls[c].kw, rows = (TH - 1) * a.sf + cls[c].kh;
        const int nq = rows * cols, np = (nq * (CH / 2) + 63) / 64;
        g.nqmax = nq > g.nqmax ? nq : g.nqmax;
        g.npmax = np > g.npmax ? np : g.npmax;
    }
    return g;
}

int num_cus() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        return v;
    }();
    return n;
}

template <int WAVES_N, int WM, int WN, int CH, int TPI>
int launch_one(PArgs& pa, size_t lds, hipStream_t stream) {
    auto fn = cconv_pipe_kernel<WAVES_N, WM, WN, CH, TPI>;
    if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
    // resident workgroups per CU for this instantiation at this LDS size (queried once per (kernel, LDS KiB))
    struct Occ { size_t lds; int per_cu; };
    static Occ cache[8];
    static int ncache = 0;
    int per_cu = 0;
    for (int i = 0; i < ncache; ++i)
        if (cache[i].lds == lds) per_cu = cache[i].per_cu;
    if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        static const int cap = [] { const char* e = getenv("DCS_PIPE_WG_PER_CU"); return e ? atoi(e) : 3; }();
        if (per_cu > cap) per_cu = cap;
        if (ncache < 8) cache[ncache++] = Occ{lds, per_cu};
    }
    long grid = (long)per_cu * num_cus();
    if (grid > pa.n_units) grid = pa.n_units;
    DCS_LAUNCH(fn, dim3((unsigned)grid), dim3(256), lds, stream, pa);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

template <int WAVES_N, int WM, int WN>
int launch_ch(PArgs& pa, int CH, size_t lds, hipStream_t stream) {
    const MArgs& m = pa.m;
    if (CH == 8 && WM == 1 && WN == 1 && m.ncls == 1 && m.cls[0].kw == 7)
        return launch_one<WAVES_N, WM, WN, 8, (WM == 1 && WN == 1) ? 7 : 1>(pa, lds, stream);
    switch (CH) {
        case 32: return launch_one<WAVES_N, WM, WN, 32, 1>(pa, lds, stream);
        case 16: return launch_one<WAVES_N, WM, WN, 16, 1>(pa, lds, stream);
        default: return launch_one<WAVES_N, WM, WN, 8, 1>(pa, lds, stream);
    }
}

}  // namespace

namespace {
long long* g_dbg = nullptr;
}
#ifdef DCS_PIPE_DIAG
extern "C" int dcs_debug_set_buffer(void* p) { g_dbg = (long long*)p; return 0; }
#endif
namespace {
// default 0: measured at B=32, T=256 (tools/conv_ab.py, interleaved rounds in one process) the pipelined schedule is at
// parity with conv_mfma.hip on the mid layers and behind it on the shallow / few-tile ones
int g_schedule = [] { const char* e = getenv("DCS_CONV_PIPE"); return (e && atoi(e) != 0) ? 1 : 0; }();
}

bool dcs_conv_pipe_enabled() { return g_schedule == 1; }

extern "C" int dcs_set_conv_schedule(int mode) {
    if (mode != 0 && mode != 1) return DCS_ERR_BADARG;
    g_schedule = mode;
    return DCS_OK;
}

extern "C" int dcs_get_conv_schedule(void) { return g_schedule; }

bool dcs_conv_pipe_eligible(const conv::Args& a, int ncls, const conv::Cls* cls, int cand, int TH, int TW, int CH) {
    if (!dcs_conv_pipe_enabled() || dcs_conv_precision() != 0) return false;
    if (cand < 0 || cand > 3 || a.sf < 1 || a.sf > 2 || a.st < 1 || a.st > 2) return false;
    if (2 * a.Cout < 32 || (CH != 8 && CH != 16 && CH != 32)) return false;
    const int Cin = a.C1 + a.C2;
    if (Cin % CH != 0 || (a.C2 > 0 && a.C1 % CH != 0)) return false;      // a chunk lies in ONE tensor of a concatenation
    const Geo g = pipe_geo(a, ncls, cls, TH, TW, CH);
    const long lds = 2L * g.npmax * 1024 + (long)((g.npmax + 3) / 4) * 4 * 256;
    return lds <= 156 * 1024;
}

int dcs_conv_pipe_launch(MArgs& m, int cand, int CH, hipStream_t stream) {
    const conv::Args& a = m.c;
    const Geo g = pipe_geo(a, m.ncls, m.cls, m.TH, m.TW, CH);
    PArgs pa;
    pa.m = m;
    pa.buf_floats = g.npmax * 256;
    pa.npw_max = (g.npmax + 3) / 4;
    pa.dbg = g_dbg;
    const size_t lds = 2 * (size_t)g.npmax * 1024 + (size_t)pa.npw_max * 4 * 256;
    const int bn = cand == 0 ? 4 : (cand == 1 ? 2 : (cand == 2 ? 2 : 1));        // 32-column tiles per workgroup
    const long units = (long)a.tiles_w * a.tiles_h * a.B * (m.NT / bn) * m.ksplit * m.ncls;
    if (units <= 0 || units >= (1L << 30)) return DCS_ERR_BADARG;
    pa.n_units = (int)units;
#ifdef DCS_PIPE_DIAG
    if (g_dbg) fprintf(stderr, "[pipe] cand %d TH %d TW %d CH %d ksplit %d cps %d ncls %d units %ld lds %zu\n", cand, m.TH, m.TW, CH,
                       m.ksplit, m.cps, m.ncls, units, lds);
#endif
    switch (cand) {
        case 0: return launch_ch<2, 2, 2>(pa, CH, lds, stream);       // 128 x 128
        case 1: return launch_ch<2, 2, 1>(pa, CH, lds, stream);       // 128 x 64
        case 2: return launch_ch<2, 1, 1>(pa, CH, lds, stream);       //  64 x 64
        default: return launch_ch<1, 1, 1>(pa, CH, lds, stream);      // 128 x 32
    }
}
