// conv_enc0.hip — the first encoder convolution (ComplexConv2d 1 -> 8 channels, 7x7, stride 2, pad 3: c_network.py:107-112,
// config.py:83-91) forward on the MFMA units.
//
// With ONE input channel the implicit GEMM of conv_mfma.hip has nothing to contract over per tap (K = 2), so the layer ran
// on the VALU kernel of conv_direct.hip (29-47 TFLOP/s).  Here the 49 taps ARE the K axis:
//     D[p][(co, re|im)] = sum_{tap, part} A[p][(tap, part)] B[(tap, part)][(co, re|im)],   K = 49 x 2 = 98 (padded to 100), N = 16
//   A[p][(tap, re|im)] = x_{re|im}[2 p + tap - pad]          read straight from the LDS input patch: the four k of a step are
//                                                            four consecutive taps, a lane reads its tap's (re, im) with ONE
//                                                            ds_read_b64 and feeds two MFMAs (re step, im step)
//   B                  = the 2x2 real embedding of the 49 x 8 complex weights: 26 fragments of one float per lane, held in
//                        REGISTERS for the whole workgroup (built from the direct panel at kernel start)
// v_mfma_f32_16x16x4_f32: 16 pixels x 16 columns x 4 k per instruction, 26 per 16-pixel tile (taps padded to 52): 94 % of
// the issued MACs are useful.  Two M-tiles run interleaved (independent accumulators, LDS latency).  A tile = 8 x 32 output pixels (16 M-tiles, four per wave) over a 21 x 69 input patch (11.6 KB); workgroups are
// persistent and load the next tile's patch under the current tile's MFMAs.
// Epilogue as everywhere: bias, optional folded eval-mode CBN (conv_common.h), activation; 64-byte rows per pixel.
#include "conv_common.h"

namespace {

#ifdef DCS_ENC0_DIAG
#define EDIAG_NOW() ((long long)__builtin_amdgcn_s_memtime())
long long* g_edbg = nullptr;
#else
#define EDIAG_NOW() 0LL
#endif

constexpr int K7 = 7, TAPS = 49, KGRP = 13, KSTEPS = 2 * KGRP, TR = 8, TC = 32, PR = (TR - 1) * 2 + K7, PC = (TC - 1) * 2 + K7, PCP = PC + 1;
typedef float f32x4v __attribute__((ext_vector_type(4)));


// One haloed input patch element per (thread, k): element i = t + 256 k of patch[r][c] = x[b][2 oy0 - 3 + r][2 ox0 - 3 + c],
// r = i / PC.  The six rows of a thread are tile-invariant: packed 5 bits each in `rows` once per workgroup, so a tile
// costs no divisions; a patch that lies wholly inside the image (the common case) loads without bounds checks.
constexpr int NL = (PR * PC + 255) / 256;
// tile -> (image b, tile row, tile column) with float reciprocals (exact for tile < 2^22: the launch checks), instead of
// two emulated integer divisions per use
struct TileDiv { float inv_per, inv_w; int per, tiles_w; };
__device__ __forceinline__ void tile_split(const TileDiv& d, int tile, int* b, int* ty, int* tx) {
    *b = (int)(((float)tile + 0.5f) * d.inv_per);
    const int tl = tile - *b * d.per;
    *ty = (int)(((float)tl + 0.5f) * d.inv_w);
    *tx = tl - *ty * d.tiles_w;
}
__device__ __forceinline__ int row_of(int rows, int k) { return (rows >> (5 * k)) & 31; }

__device__ __forceinline__ void patch_load(const conv::Args& a, const TileDiv& d, int tile, int t, int rows, float2* pv) {
    int b, ty, tx;
    tile_split(d, tile, &b, &ty, &tx);
    const int y0 = ty * (2 * TR) - 3, x0 = tx * (2 * TC) - 3;
    const act2_t* xb = a.x1 + (long)b * a.Hin * a.Win;
    if (y0 >= 0 && y0 + PR <= a.Hin && x0 >= 0 && x0 + PC <= a.Win) {                 // (uniform)
        const act2_t* org = xb + (long)y0 * a.Win + x0;
        const int skip = a.Win - PC;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = t + 256 * k;
            if (k < NL - 1 || i < PR * PC) pv[k] = conv::ldc(org + i + row_of(rows, k) * skip);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = t + 256 * k, r = row_of(rows, k), c = i - r * PC;
            const int y = y0 + r, x = x0 + c;
            const bool in_ = i < PR * PC && y >= 0 && y < a.Hin && x >= 0 && x < a.Win;
            const float2 v_ = conv::ldc(xb + (in_ ? (long)y * a.Win + x : 0));
            pv[k] = in_ ? v_ : make_float2(0.f, 0.f);
        }
    }
}

// bias, folded eval-mode CBN, activation and store of one lane's 2 x 4 accumulators (M-tiles h = 0, 1; rows kg*4 + r)
template <int ACT, bool CHECK, bool STAT>
__device__ __forceinline__ void store_pair(const conv::Args& a, const f32x4v* acc, act_t* yp, int ox, float bv, float c_re,
                                           float c_im, float c_add, int li, float* st) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (STAT) {                                     // CBN statistics of the raw, UN-biased output
                const float raw = (!CHECK || ox + h * 16 + r < a.Wout) ? acc[h][r] : 0.f;
                st[0] += raw; st[1] = fmaf(raw, raw, st[1]); st[2] = fmaf(raw, dcs_dpp_term<0xB1, 0xf>(raw), st[2]);
            }
            float v = acc[h][r] + bv;
            if (a.coef) {
                const float pv = dcs_dpp_term<0xB1, 0xf>(v);
                v = (li & 1) ? fmaf(c_re, pv, fmaf(c_im, v, c_add)) : fmaf(c_re, v, fmaf(c_im, pv, c_add));
            }
            v = ACT < 0 ? dcs_act(v, a.act) : dcs_act(v, ACT);
            if (!CHECK || ox + h * 16 + r < a.Wout) dcs_st1(yp + (h * 16 + r) * 16, v);
        }
}

// one row of partial CBN sums per (persistent) workgroup: over the 4 row groups of a column by shuffles, over the waves in LDS
__device__ __forceinline__ void stat_rows_out(const conv::Args& a, float* st, float* red, int t, int lane, int wave, int li) {
#pragma unroll
    for (int e = 0; e < 3; ++e) { st[e] += __shfl_xor(st[e], 16, 64); st[e] += __shfl_xor(st[e], 32, 64); }
    __syncthreads();                                        // every wave is done with the patches
    if (lane < 16) { red[(wave * 16 + li) * 3] = st[0]; red[(wave * 16 + li) * 3 + 1] = st[1]; red[(wave * 16 + li) * 3 + 2] = st[2]; }
    __syncthreads();
    if (t < 40) {                                           // channel c: {S_r, S_i, S_rr, S_ii, S_ri}
        const int c = t / 5, e = t % 5;
        const int colx = 2 * c + (e == 1 || e == 3), which = e < 2 ? 0 : (e < 4 ? 1 : 2);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += red[(w * 16 + colx) * 3 + which];
        a.stat[(long)t * a.stat_stride + blockIdx.x] = sum;
    }
}

// Persistent workgroups: tile = blockIdx.x, += gridDim.x.  The NEXT tile's patch is loaded into registers while this
// tile's MFMAs run (two LDS buffers, one barrier per tile), the B fragments are built once per workgroup.
// ACT: the activation at compile time, or -1 for a.act at run time.
// STAT: also leave the training-mode CBN statistics of the raw output (conv_common.h Args::stat), one row per workgroup.
template <int ACT, bool STAT = false>
__global__ __launch_bounds__(256, 4) void cconv_enc0_kernel(conv::Args a, TileDiv d, int ntile, long long* dbg) {
    __shared__ float2 patch[2][PR * PCP];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int tile = blockIdx.x;
    if (tile >= ntile) return;
    int rows = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) rows |= ((t + 256 * k) / PC) << (5 * k);
    float2 pv_[NL];
    patch_load(a, d, tile, t, rows, pv_);

    // K order: step 2j = (tap 4j + kg, re), step 2j + 1 = (tap 4j + kg, im); taps 49..51 are zero padding.
    // B fragments of column n = li (co = li >> 1, re | im = li & 1), and this lane's tap offsets into a pixel window (floats)
    float bf[KSTEPS];
    int toff[KGRP];
    {
        const int co = li >> 1, im = li & 1;
#pragma unroll
        for (int j = 0; j < KGRP; ++j) {
            const int tap = 4 * j + kg;
            const int tc = tap < TAPS ? tap : TAPS - 1;                    // padding taps: any in-range address (B is 0 there)
            // always a load, zeroed afterwards: under `if (tap < TAPS)` each of the 13 loads sat in its own branch and was
            // waited for with vmcnt(0) before the next one was issued — 13 serial round trips at the head of every workgroup
            float2 w = a.wp[tc * a.Cout + co];                             // direct panel complex[tap][ci = 0][co]
            if (tap >= TAPS) w = make_float2(0.f, 0.f);
            bf[2 * j] = im ? w.y : w.x;                                    // x_re: (w_r -> re, w_i -> im)
            bf[2 * j + 1] = im ? w.x : -w.y;                               // x_im: (-w_i -> re, w_r -> im)
            toff[j] = ((tc / K7) * PCP + (tc % K7)) * 2;
        }
    }
    const float* biasf = reinterpret_cast<const float*>(a.bias);
    const float bv = biasf ? biasf[li] : 0.f;
    float c_re = 1.f, c_im = 0.f, c_add = 0.f;                            // folded eval-mode CBN (conv_mfma.hip)
    if (a.coef) {
        const float* q = a.coef + 6 * (li >> 1);
        if (li & 1) { c_re = q[2]; c_im = q[3]; c_add = q[5]; } else { c_re = q[0]; c_im = q[1]; c_add = q[4]; }
    }
    int buf = 0;
    float st[3] = {0.f, 0.f, 0.f};                          // this lane's column: sum, sum of squares, sum of re * im
    long long d_fill = 0, d_comp = 0, d_n = 0;
    const long long d_start = EDIAG_NOW();
#pragma unroll 1
    for (; tile < ntile; tile += gridDim.x, buf ^= 1) {
        const long long e0 = EDIAG_NOW();
        // (buffer `buf` was last read two tiles ago, and every wave passed the barrier of the tile in between since)
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = t + 256 * k;
            if (k < NL - 1 || i < PR * PC) patch[buf][i + row_of(rows, k)] = pv_[k];         // r PCP + c = i + r
        }
        __syncthreads();
        const long long e1 = EDIAG_NOW();
        d_fill += e1 - e0; ++d_n;
        if (tile + (int)gridDim.x < ntile) patch_load(a, d, tile + gridDim.x, t, rows, pv_);
        int b, ty, tx;
        tile_split(d, tile, &b, &ty, &tx);
        const int oy0 = ty * TR, ox0 = tx * TC;
        const float* pf = reinterpret_cast<const float*>(patch[buf]);
#pragma unroll 1
        for (int i = 0; i < 4; i += 2) {
            const int mt = wave * 4 + i;                                   // 16 M-tiles: row mt / 2, 16 columns each;
            const int py = mt >> 1;                                        // mt (columns 0-15) and mt + 1 (16-31) share a row
            const float* base = pf + ((py * 2) * PCP + li * 2) * 2;        // this lane's pixel window in M-tile mt
            f32x4v acc[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
            // A reads run PF tap groups ahead of the MFMAs that consume them (ring of registers, constant indices after
            // unrolling): left to itself the compiler re-uses one register quad and waits out every LDS round trip
            constexpr int PF = 3;
            float2 va[PF][2];
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                va[j][0] = *reinterpret_cast<const float2*>(base + toff[j]);
                va[j][1] = *reinterpret_cast<const float2*>(base + toff[j] + 16 * 2 * 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < KGRP; ++j) {
                const float2 v0 = va[j % PF][0], v1 = va[j % PF][1];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.x, bf[2 * j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.x, bf[2 * j], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.y, bf[2 * j + 1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.y, bf[2 * j + 1], acc[1], 0, 0, 0);
                if (j + PF < KGRP) {
                    va[j % PF][0] = *reinterpret_cast<const float2*>(base + toff[j + PF]);
                    va[j % PF][1] = *reinterpret_cast<const float2*>(base + toff[j + PF] + 16 * 2 * 2);
                }
                __builtin_amdgcn_sched_barrier(0);          // (keeps the read where it is written: the scheduler sinks it to its use)
            }
            // C/D: col = li, rows kg*4 + r -> pixel (oy0 + py, ox0 + h*16 + kg*4 + r); one row pointer, constant offsets
            const int oy = oy0 + py;
            if (oy < a.Hout) {
                act_t* yp = reinterpret_cast<act_t*>(a.y) + (((long)b * a.Hout + oy) * a.Wout + ox0 + kg * 4) * 16 + li;
                if (ox0 + TC <= a.Wout) store_pair<ACT, false, STAT>(a, acc, yp, 0, bv, c_re, c_im, c_add, li, st);
                else store_pair<ACT, true, STAT>(a, acc, yp, ox0 + kg * 4, bv, c_re, c_im, c_add, li, st);
            }
        }
        d_comp += EDIAG_NOW() - e1;
    }
    if (STAT) stat_rows_out(a, st, reinterpret_cast<float*>(patch[0]), t, lane, wave, li);
#ifdef DCS_ENC0_DIAG
    if (dbg && t == 0 && blockIdx.x < 4096) {
        long long* q = dbg + blockIdx.x * 8;
        q[0] = d_fill; q[1] = d_comp; q[2] = EDIAG_NOW() - d_start; q[3] = d_n; q[4] = d_start;
        q[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4); q[6] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    }
#endif
}


// ---- the same layer with bf16 MFMA operands: v_mfma_f32_16x16x32_bf16 ---------------------------------------------------
// The fp32 MFMA above is 2 x slower per MAC than six bf16 MFMAs on exact three-way splits (conv_common.h precision 2), and
// the layer is bound by the MFMA pipe (26 x 32 cycles per 16 pixels against 8 us of HBM traffic).  Here K is ordered
// (kernel row dy, dx, re|im) with dx padded to 8 and dy to 8: 128 = 4 steps of 32, step s = rows 2s, 2s + 1; a lane's eight
// k of a step = (row 2s + (kg >> 1), dx = 4 (kg & 1) .. + 3, re|im) = FOUR ADJACENT PATCH PIXELS of one plane, 16 contiguous
// bytes of the LDS patch, which is kept as NPL planes of (re, im) bf16 pairs (the exact split is done once per patch element
// when it is written).  24 (NPL = 3) or 4 (NPL = 1: bf16 operands) MFMAs of 16 cycles per 16 pixels instead of 26 of 32.
// Patch row 21 and column 69 (the dy = 7 / dx = 7 padding of the last pixels) are zeros written once.
// (Measured and dropped: the transposed product D[column][pixel], one 16-byte store per lane and M-tile instead of four 4-byte
// stores — a lane quad then writes four different 64-byte segments, 26.2 -> 30.5 us.)
constexpr int PRX = PR + 1, KS = 4;

typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

template <int ACT, bool STAT, int NPL>
__global__ __launch_bounds__(256, 4) void cconv_enc0b_kernel(conv::Args a, TileDiv d, int ntile, long long* dbg) {
    const long long d_start = EDIAG_NOW();
#ifdef DCS_ENC0_DIAG
    const long long d_rt0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz, one counter for the device (x 24 = 2.4 GHz units)
#endif
    __shared__ __attribute__((aligned(16))) unsigned patch[2][NPL][PRX * PCP];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int tile = blockIdx.x;
    if (tile >= ntile) return;
    int rows = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) rows |= ((t + 256 * k) / PC) << (5 * k);
    float2 pv_[NL];
    // B fragments: column n = li (co = li >> 1, re | im = li & 1); step s, element e: row 2s + (kg >> 1), dx = 4 (kg & 1) + (e >> 1),
    // part e & 1.  Every wave holds the same 4 x NPL fragments: wave s builds those of step s (the split of all four in
    // every wave was 400 VALU instructions x 16 waves per CU: 6 of the kernel's 19 us) and the waves trade them through LDS.
    uint4 bq[KS][NPL];
    {
        const int co = li >> 1, im = li & 1, s = wave;
        float wv[8];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const int dy = 2 * s + (kg >> 1), dx = 4 * (kg & 1) + e2;
            const bool ok = dy < K7 && dx < K7;
            float2 w = a.wp[(ok ? dy * K7 + dx : 0) * a.Cout + co];        // always a load (see the fp32 kernel above)
            if (!ok) w = make_float2(0.f, 0.f);
            wv[2 * e2] = im ? w.y : w.x;                                   // x_re: (w_r -> re, w_i -> im)
            wv[2 * e2 + 1] = im ? w.x : -w.y;                              // x_im: (-w_i -> re, w_r -> im)
        }
        patch_load(a, d, tile, t, rows, pv_);                              // (after the weight loads: their wait does not include it)
        uint4* bx = reinterpret_cast<uint4*>(&patch[1][0][0]);             // (KS * NPL * 64 * 16 B = 12 KB of the 18 KB buffer)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            unsigned h[4];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                h[e2] = dcs_pack_bf16x2(wv[2 * e2], wv[2 * e2 + 1]);
                wv[2 * e2] -= __uint_as_float(h[e2] << 16);
                wv[2 * e2 + 1] -= __uint_as_float(h[e2] & 0xffff0000u);
            }
            bx[(s * NPL + pl) * 64 + lane] = make_uint4(h[0], h[1], h[2], h[3]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KS; ++q)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) bq[q][pl] = bx[(q * NPL + pl) * 64 + lane];
        __syncthreads();
    }
    // the padding cells (disjoint from what a fill writes; the first barrier below orders them before any read)
    for (int i = t; i < 2 * NPL * (PCP + PR); i += 256) {
        const int c = i % (PCP + PR), q = i / (PCP + PR);
        (&patch[0][0][0])[q * (PRX * PCP) + (c < PCP ? PR * PCP + c : (c - PCP) * PCP + PC)] = 0u;
    }
    const float* biasf = reinterpret_cast<const float*>(a.bias);
    const float bv = biasf ? biasf[li] : 0.f;
    float c_re = 1.f, c_im = 0.f, c_add = 0.f;                            // folded eval-mode CBN (conv_mfma.hip)
    if (a.coef) {
        const float* q = a.coef + 6 * (li >> 1);
        if (li & 1) { c_re = q[2]; c_im = q[3]; c_add = q[5]; } else { c_re = q[0]; c_im = q[1]; c_add = q[4]; }
    }
    int buf = 0;
    float st[3] = {0.f, 0.f, 0.f};
    long long d_fill = 0, d_comp = 0, d_n = 0;
    const long long d_pro = EDIAG_NOW();
    // this lane's window origin inside a patch row pair: pixel column 2 li + 4 (kg & 1), row kg >> 1
    const int lane_off = (kg >> 1) * PCP + 2 * li + 4 * (kg & 1);
#pragma unroll 1
    for (; tile < ntile; tile += gridDim.x, buf ^= 1) {
        const long long e0 = EDIAG_NOW();
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = t + 256 * k;
            if (k < NL - 1 || i < PR * PC) {
                float2 r = pv_[k];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const unsigned h = dcs_pack_bf16x2(r.x, r.y);
                    patch[buf][pl][i + row_of(rows, k)] = h;
                    r.x -= __uint_as_float(h << 16);
                    r.y -= __uint_as_float(h & 0xffff0000u);
                }
            }
        }
        __syncthreads();
        const long long e1 = EDIAG_NOW();
        d_fill += e1 - e0; ++d_n;
        // (opaque: the per-element address terms derived from `rows` are otherwise hoisted out of the tile loop and spilled)
        asm volatile("" : "+v"(rows));
        if (tile + (int)gridDim.x < ntile) patch_load(a, d, tile + gridDim.x, t, rows, pv_);
        int b, ty, tx;
        tile_split(d, tile, &b, &ty, &tx);
        const int oy0 = ty * TR, ox0 = tx * TC;
#pragma unroll 1
        for (int i = 0; i < 4; i += 2) {
            const int mt = wave * 4 + i;                                   // M-tiles mt (columns 0-15) and mt + 1 (16-31): row mt / 2
            const int py = mt >> 1;
            const unsigned* base = &patch[buf][0][0] + (2 * py) * PCP + lane_off;
            f32x4v acc[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
            // A fragments [M-tile][plane]: a tile's registers are re-loaded for the next step as soon as its MFMAs are issued,
            // under the other tile's MFMAs (a two-deep ring of both tiles spilled at 128 VGPRs)
            uint4 av[2][NPL];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const uint2* q = reinterpret_cast<const uint2*>(base + pl * (PRX * PCP) + h * 32);
                    const uint2 lo = q[0], hi = q[1];
                    av[h][pl] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                constexpr int pa[6] = {0, 1, 2, 0, 1, 0}, pb[6] = {2, 1, 0, 1, 0, 0};      // a0 b2, a1 b1, a2 b0, a0 b1, a1 b0, a0 b0
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int e = (NPL == 3 ? 0 : 5); e < 6; ++e)
                        acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, av[h][pa[e]]),
                                                                         __builtin_bit_cast(bf16x8v, bq[s][pb[e]]), acc[h], 0, 0, 0);
                    if (s + 1 < KS) {
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) {
                            const uint2* q = reinterpret_cast<const uint2*>(base + pl * (PRX * PCP) + h * 32 + (s + 1) * 2 * PCP);
                            const uint2 lo = q[0], hi = q[1];
                            av[h][pl] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int oy = oy0 + py;
            if (oy < a.Hout) {
                act_t* yp = reinterpret_cast<act_t*>(a.y) + (((long)b * a.Hout + oy) * a.Wout + ox0 + kg * 4) * 16 + li;
                if (ox0 + TC <= a.Wout) store_pair<ACT, false, STAT>(a, acc, yp, 0, bv, c_re, c_im, c_add, li, st);
                else store_pair<ACT, true, STAT>(a, acc, yp, ox0 + kg * 4, bv, c_re, c_im, c_add, li, st);
            }
        }
        d_comp += EDIAG_NOW() - e1;
    }
    if (STAT) stat_rows_out(a, st, reinterpret_cast<float*>(&patch[0][0][0]), t, lane, wave, li);
#ifdef DCS_ENC0_DIAG
    if (dbg && t == 0 && blockIdx.x < 4096) {
        long long* q = dbg + blockIdx.x * 8;
        q[0] = d_fill; q[1] = d_comp; q[2] = EDIAG_NOW() - d_start; q[3] = d_n; q[4] = d_rt0 * 24; q[2] = ((long long)__builtin_amdgcn_s_memrealtime() - d_rt0) * 24;
        q[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4); q[6] = __builtin_amdgcn_s_getreg((3 << 11) | 20); q[7] = d_pro - d_start;
    }
#endif
}


// ---- weight gradient of the same layer on the MFMA units --------------------------------------------------------------
// g_W[tap][co] = sum_p g_Y[p][co] conj(X[2p - 3 + tap]) as D[(tap, re|im)][(co, re|im)] = sum_p A[(tap, part)][p] B[p][(co, q)]:
//   M = 98 rows (+ row 98: A = 1, the bias gradient; rows 99..111 zero) in 7 M-tiles, N = 16, K = pixels, 4 per instruction;
//   A from the same LDS input patch as the forward pass (ds_read_b32, one per MFMA, read one pixel quad ahead),
//   B = g_Y straight from HBM (a wave's quad = 256 contiguous bytes), 16 quads per wave per tile loaded up front.
// Persistent workgroups keep the 7 x 4 accumulators across their tiles and write ONE partial slab each (the slab layout
// and reduce kernel of the other weight-gradient paths):  G.re = D[(t,re)][(c,re)] + D[(t,im)][(c,im)],
// G.im = D[(t,re)][(c,im)] - D[(t,im)][(c,re)].
constexpr int MT = 7, QPW = 16;                              // M-tiles; pixel quads per wave per tile (2 rows x 8)
constexpr int RED_FLOATS = 3 * MT * 4 * 64, DL_FLOATS = MT * 16 * 16;
constexpr int WG_SMEM = (RED_FLOATS + DL_FLOATS) * 4 > 2 * PR * PCP * 8 ? (RED_FLOATS + DL_FLOATS) * 4 : 2 * PR * PCP * 8;

__global__ __launch_bounds__(256) void cconv_enc0_wgrad_kernel(conv::Args a, TileDiv d, int ntile, const act_t* __restrict__ gy,
                                                               float2* __restrict__ slab_w, float2* __restrict__ slab_b) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[WG_SMEM];
    float2 (*patch)[PR * PCP] = reinterpret_cast<float2 (*)[PR * PCP]>(smem);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int rows = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) rows |= ((t + 256 * k) / PC) << (5 * k);
    // this lane's A row in M-tile m: R = 16 m + li -> (tap = R >> 1, part = R & 1); float offset into a pixel window
    int toff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int tap = 8 * m + (li >> 1), tc = tap < TAPS ? tap : TAPS - 1;
        toff[m] = ((tc / K7) * PCP + (tc % K7)) * 2 + (li & 1);
    }
    const float a6 = li == 2 ? 1.f : 0.f;                    // M-tile 6: rows 96, 97 = tap 48; 98 = ones (bias); above: zero
    f32x4v acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4v{0.f, 0.f, 0.f, 0.f};

    int tile = blockIdx.x, buf = 0;
    float2 pv_[NL];
    if (tile < ntile) patch_load(a, d, tile, t, rows, pv_);
#pragma unroll 1
    for (; tile < ntile; tile += gridDim.x, buf ^= 1) {
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = t + 256 * k;
            if (k < NL - 1 || i < PR * PC) patch[buf][i + row_of(rows, k)] = pv_[k];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntile) patch_load(a, d, tile + gridDim.x, t, rows, pv_);
        int b, ty, tx;
        tile_split(d, tile, &b, &ty, &tx);
        const int oy0 = ty * TR + 2 * wave, ox0 = tx * TC + kg;            // this wave's two rows; this lane's pixel in a quad
        // g_Y of the wave's 16 quads: quad q = row (q >> 3), columns (q & 7) * 4 + kg
        float gv[QPW];
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            const int oy = oy0 + (q >> 3), ox = ox0 + (q & 7) * 4;
            // always a load (clamped pixel), zeroed afterwards: as sixteen predicated loads each one sat in its own branch
            // and was waited for with vmcnt(0) before the next was issued
            const bool inb = oy < a.Hout && ox < a.Wout;
            const float v = dcs_ld1(gy + (((long)b * a.Hout + (inb ? oy : 0)) * a.Wout + (inb ? ox : 0)) * 16 + li);
            gv[q] = inb ? v : 0.f;
        }
        // lane's window origin for quad 0: pixel (2 wave, kg); quad q adds a compile-time offset
        const float* base = reinterpret_cast<const float*>(patch[buf]) + ((4 * wave) * PCP + 2 * kg) * 2;
        const float* am[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) am[m] = base + toff[m];
        float av[2][MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[0][m] = am[m][0];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            if (q + 1 < QPW) {
                const int qo = (((q + 1) >> 3) * 2 * PCP + ((q + 1) & 7) * 8) * 2;
#pragma unroll
                for (int m = 0; m < MT; ++m) av[(q + 1) & 1][m] = am[m][qo];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float x = av[q & 1][m];
                if (m == MT - 1) x = li < 2 ? x : a6;
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, gv[q], acc[m], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);               // (keeps the read-ahead where it is written)
        }
    }
    // workgroup total: waves 1..3 through LDS into wave 0, D rows to LDS, then the complex combination
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    float* Dl = red + RED_FLOATS;
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave - 1) * MT * 4 + m * 4 + r) * 64 + lane] = acc[m][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[m][r];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += red[(w * MT * 4 + m * 4 + r) * 64 + lane];
                Dl[(16 * m + kg * 4 + r) * 16 + li] = v;                   // C/D: col = li, row = kg*4 + r
            }
    }
    __syncthreads();
    float2* slab = slab_w + (long)blockIdx.x * (TAPS * 8);
    for (int i = t; i < TAPS * 8; i += 256) {
        const int tap = i >> 3, co = i & 7;
        const float* d0 = Dl + (2 * tap) * 16 + 2 * co;                    // row (tap, re): cols (co, re), (co, im)
        const float* d1 = d0 + 16;                                         // row (tap, im)
        slab[i] = make_float2(d0[0] + d1[1], d0[1] - d1[0]);
    }
    if (t < 8) slab_b[(long)blockIdx.x * 8 + t] = make_float2(Dl[98 * 16 + 2 * t], Dl[98 * 16 + 2 * t + 1]);
}

}  // namespace

static bool enc0_geom(const conv::Args& a);
bool dcs_conv_enc0_ok(const conv::Args& a) { return enc0_geom(a) && a.x1 && a.wp && a.y; }
static bool enc0_geom(const conv::Args& a) {
    return a.C1 == 1 && a.C2 == 0 && a.Cout == 8 && a.kh == K7 && a.kw == K7 && a.sf == 2 && a.st == 2 && a.pad_f == 3 &&
           a.pad_t == 3 && a.up_f == 1 && a.up_t == 1 && !a.zero_ins &&
           (long)a.B * ((a.Hin / 2 + TR) / TR) * ((a.Win / 2 + TC) / TC) < (1L << 22);
}

int dcs_conv_enc0_launch(conv::Args a, hipStream_t stream) {
    if (!dcs_conv_enc0_ok(a)) return DCS_ERR_BADARG;
    a.Hout = (a.Hin + 2 * a.pad_f - K7) / 2 + 1;
    a.Wout = (a.Win + 2 * a.pad_t - K7) / 2 + 1;
    a.tiles_w = (a.Wout + TC - 1) / TC;
    a.tiles_h = (a.Hout + TR - 1) / TR;
    const long ntile = (long)a.tiles_w * a.tiles_h * a.B;
    if (ntile >= (1L << 22)) return DCS_ERR_BADARG;          // (tile_split's float reciprocals)
    TileDiv d;
    d.per = a.tiles_w * a.tiles_h; d.tiles_w = a.tiles_w; d.inv_per = 1.f / (float)d.per; d.inv_w = 1.f / (float)d.tiles_w;
    static int resident = 0;                                // workgroups the device holds at once (4 / CU: 120 VGPRs)
    if (!resident) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return DCS_ERR_LAUNCH;
        resident = cus * 4;
    }
    const int grid = ntile < resident ? (int)ntile : resident;
#ifdef DCS_ENC0_DIAG
    long long* dbgp = g_edbg;
#else
    long long* dbgp = nullptr;
#endif
    if (a.stat && (a.act != DCS_ACT_NONE || a.coef)) return DCS_ERR_BADARG;
    static const int fp32_mfma = [] { const char* e = getenv("DCS_ENC0_F32"); return e ? atoi(e) : 0; }();          // 1: the fp32-MFMA kernel also under precision 1 / 2
    const int prec = DCS_ACT_IS_BF16 ? 1 : dcs_conv_precision();
    if (prec != 0 && !fp32_mfma) {
#define ENC0B(ACT_, STAT_)                                                                                                 \
        do {                                                                                                               \
            if (prec == 2) DCS_LAUNCH((cconv_enc0b_kernel<ACT_, STAT_, 3>), dim3(grid), dim3(256), 0, stream, a, d, (int)ntile, dbgp);  \
            else DCS_LAUNCH((cconv_enc0b_kernel<ACT_, STAT_, 1>), dim3(grid), dim3(256), 0, stream, a, d, (int)ntile, dbgp);     \
        } while (0)
        if (a.stat) ENC0B(DCS_ACT_NONE, true);
        else if (a.act == DCS_ACT_NONE) ENC0B(DCS_ACT_NONE, false);
        else if (a.act == DCS_ACT_RELU) ENC0B(DCS_ACT_RELU, false);
        else ENC0B(-1, false);
#undef ENC0B
    }
    else if (a.stat) DCS_LAUNCH((cconv_enc0_kernel<DCS_ACT_NONE, true>), dim3(grid), dim3(256), 0, stream, a, d, (int)ntile, dbgp);
    else if (a.act == DCS_ACT_NONE) DCS_LAUNCH(cconv_enc0_kernel<DCS_ACT_NONE>, dim3(grid), dim3(256), 0, stream, a, d, (int)ntile, dbgp);
    else if (a.act == DCS_ACT_RELU) DCS_LAUNCH(cconv_enc0_kernel<DCS_ACT_RELU>, dim3(grid), dim3(256), 0, stream, a, d, (int)ntile, dbgp);
    else DCS_LAUNCH(cconv_enc0_kernel<-1>, dim3(grid), dim3(256), 0, stream, a, d, (int)ntile, dbgp);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// Weight gradient (forward geometry `a`, x1 set): at most max_slabs partial slabs float2[49][8] (+ float2[8] bias) are
// written, *n_used says how many; the caller reduces them (launch_wgrad_reduce of conv_direct.hip).
int dcs_conv_enc0_stat_rows(const conv::Args& a0) {
    conv::Args a = a0;
    if (!enc0_geom(a)) return 0;
    a.Hout = (a.Hin + 2 * a.pad_f - K7) / 2 + 1;
    a.Wout = (a.Win + 2 * a.pad_t - K7) / 2 + 1;
    const long ntile = (long)((a.Wout + TC - 1) / TC) * ((a.Hout + TR - 1) / TR) * a.B;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return 0;
    return (int)(ntile < cus * 4 ? ntile : cus * 4);
}

bool dcs_conv_enc0_wgrad_ok(const conv::Args& a) { return enc0_geom(a) && a.x1; }
int dcs_conv_enc0_wgrad_launch(conv::Args a, const act_t* gy, float2* slab_w, float2* slab_b, int max_slabs, int* n_used,
                               hipStream_t stream) {
    if (!dcs_conv_enc0_wgrad_ok(a) || !gy || !slab_w || !slab_b || max_slabs < 1) return DCS_ERR_BADARG;
    a.Hout = (a.Hin + 2 * a.pad_f - K7) / 2 + 1;
    a.Wout = (a.Win + 2 * a.pad_t - K7) / 2 + 1;
    a.tiles_w = (a.Wout + TC - 1) / TC;
    a.tiles_h = (a.Hout + TR - 1) / TR;
    const long ntile = (long)a.tiles_w * a.tiles_h * a.B;
    if (ntile >= (1L << 22)) return DCS_ERR_BADARG;
    TileDiv d;
    d.per = a.tiles_w * a.tiles_h; d.tiles_w = a.tiles_w; d.inv_per = 1.f / (float)d.per; d.inv_w = 1.f / (float)d.tiles_w;
    int grid = max_slabs < 1024 ? max_slabs : 1024;
    if (ntile < grid) grid = (int)ntile;
    // equal rounds: with R = ceil(ntile / grid) tiles per workgroup, ceil(ntile / R) workgroups do the same work in the same
    // time and leave fewer slabs to reduce
    const int rounds = (int)((ntile + grid - 1) / grid);
    grid = (int)((ntile + rounds - 1) / rounds);
    DCS_LAUNCH(cconv_enc0_wgrad_kernel, dim3(grid), dim3(256), 0, stream, a, d, (int)ntile, gy, slab_w, slab_b);
    DCS_CHECK_LAUNCH();
    *n_used = grid;
    return DCS_OK;
}

#if defined(DCS_ENC0_DIAG) && !defined(DCS_ACT_BF16)
extern "C" int dcs_debug_set_enc0_buffer(void* p) { g_edbg = (long long*)p; return 0; }
#endif
