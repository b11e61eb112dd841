// conv_up1.hip — the last decoder stage's forward pass in one kernel: a 3x3, stride-1 complex conv with ONE output channel
// over the 2x2 nearest upsample of cat(x1, x2) (ComplexConvTranspose2d 16 -> 1 behind cat + upsample: c_network.py:135-141,
// :214-217; config.py:84-106).
//
// The training path factors this stage as a 1x1 MFMA conv into 9 "tap channels" at source resolution plus a tap-sum gather
// (elementwise.hip) because its backward wants the tap channels anyway; going forward that costs a 67 MB intermediate
// (write + read) for a 17 MB result.  Here a workgroup stages a haloed 10 x 34 source tile (16 channels, channel-major
// planes) in LDS once and produces the 16 x 64 outputs above it directly:
//   * upsample folded: an output pixel of parity (py, px) reads a 2 x 2 source neighbourhood with tap-summed weights
//     (4 x 16 complex MACs instead of 9 x 16); the 4 x 2 x 2 x 16 folded weights are formed in LDS from the tap-rows
//     panel at the start of the workgroup;
//   * a wave owns ONE parity class (its weights are broadcast reads), a thread 4 consecutive pixels of one class row:
//     per (channel, source row) it loads 5 inputs for 8 MACs;
//   * complex MAC = two v_pk_fma_f32.
// HBM traffic = the two sources once + the result.
#include "conv_common.h"

namespace {

constexpr int CIN = 16, SRT = 8, SCT = 32, HR = SRT + 2, HC = SCT + 2, HCP = HC + 3;   // source tile, halo; row pitch 37 complex: the 8 x 8 lanes of a half-wave hit distinct banks
typedef float v2f __attribute__((ext_vector_type(2)));

struct Up1Args {
    const void* x1; const void* x2; const float2* wt;           // wt: tap-rows panel complex[CIN][ct], tap = dy*3 + dx
    const float* b_r; const float* b_i; float2* y;
    int Hs, Ws, C1, C2, ct, tiles_w, tiles;
};

// IT: element type of the two sources — float, or bf16 (unsigned short) where the activations live in bf16
// (dcs_cconv_up2_single_fwd_h); the result (the network's fp32 mask) and the arithmetic are fp32 either way.
//
// Round 5, second form.  The first one (a wave = one parity class, a thread four outputs of a class row) read five patch elements
// and two weight elements from LDS per eight complex MACs: its counters said LDS busy 23 us of 41 (42 % conflict cycles on the
// merged 8-byte reads) and 1260 vector instructions per wave for 512 FMAs.  Now a thread owns a 2 x 2 block of SOURCE pixels,
// i.e. the 4 x 4 outputs above it (all four parity classes), and a wave four of the sixteen channels: per channel it reads the
// block's 4 x 4 haloed source window once (eight aligned 16-byte reads) and the sixteen folded weights ({w.x, w.x, w.y, w.y}:
// broadcast reads whose halves are register pairs as they arrive) for 64 complex MACs — 24 LDS instructions per 128 packed FMAs
// instead of 56; the four waves' partial sums meet in LDS (over the spent patch) in a fixed order and leave as 32-byte runs.
template <typename IT>
__global__ __launch_bounds__(256) void cconv_up1_kernel(Up1Args p) {
    constexpr int TP = HC + 2;                               // row pitch 36 float2: 16-byte aligned rows for the window reads
    constexpr int PLANE = HR * TP + 12;                      // 372 float2 = 4 (mod 16): the eight channel pairs of a pixel store to distinct bank groups
    __shared__ __attribute__((aligned(16))) float2 tile[CIN * PLANE];            // [ci][row][col]; reused for the waves' partial sums
    __shared__ __attribute__((aligned(16))) float4 wf[CIN][4][2][2];             // [ci][parity class][a][b]: {w.x, w.x, w.y, w.y}
    static_assert(CIN * PLANE >= 4 * 16 * 64 && (PLANE % 2) == 0, "the partial sums of four waves fit over the patch");
    const int t = threadIdx.x, b = blockIdx.y;
    const int m0 = ((int)blockIdx.x / p.tiles_w) * SRT, n0 = ((int)blockIdx.x % p.tiles_w) * SCT;

    {   // folded weights: rows/cols of the 3x3 kernel that land on the same source pixel are summed
        const int ci = t & 15, bb = (t >> 4) & 1, aa = (t >> 5) & 1, cls = t >> 6;
        const int py = cls >> 1, px = cls & 1;
        // parity 0: a = 0 <- {0}, a = 1 <- {1, 2};  parity 1: a = 0 <- {0, 1}, a = 1 <- {2}
        const int dy_lo = py == 0 ? (aa == 0 ? 0 : 1) : (aa == 0 ? 0 : 2), dy_hi = py == 0 ? (aa == 0 ? 0 : 2) : (aa == 0 ? 1 : 2);
        const int dx_lo = px == 0 ? (bb == 0 ? 0 : 1) : (bb == 0 ? 0 : 2), dx_hi = px == 0 ? (bb == 0 ? 0 : 2) : (bb == 0 ? 1 : 2);
        float2 sw = make_float2(0.f, 0.f);
        for (int dy = dy_lo; dy <= dy_hi; ++dy)
            for (int dx = dx_lo; dx <= dx_hi; ++dx) {
                const float2 w = p.wt[ci * p.ct + dy * 3 + dx];
                sw.x += w.x; sw.y += w.y;
            }
        wf[ci][cls][aa][bb] = make_float4(sw.x, sw.x, sw.y, sw.y);
    }
    // haloed source tile, channel-major planes; one float4 (2 channels) per load.  All of a thread's loads are issued
    // before the first LDS write (a load -> wait -> write loop pays one memory round trip per element: 11 per tile).
    // A thread keeps ONE channel pair (256 % 8 == 0) and walks the patch pixels t / 8, t / 8 + 32, ...: source tensor and channel
    // offset are chosen once, (row, column) advance by carries (per slot: two divisions by constants, the tensor select and a
    // 64-bit multiply-add before — 440 of the kernel's 1260 vector instructions per wave were this loop's).
    constexpr int NPIX = HR * HC, NL = (NPIX + 31) / 32;
    const int q = t & 7, c = 2 * q;
    const bool first = c < p.C1;
    const IT* const sbase = first ? (const IT*)p.x1 + 2 * c : (const IT*)p.x2 + 2 * (c - p.C1);
    const int cs2 = 2 * (first ? p.C1 : p.C2);
    const long img = (long)b * p.Hs * p.Ws;
    float4 sv[NL];
    {
        int hr = (t >> 3) / HC, hc = (t >> 3) % HC;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int sy = m0 - 1 + hr, sx = n0 - 1 + hc;
            const bool in = hr < HR && sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws;
            const int syc = sy < 0 ? 0 : (sy >= p.Hs ? p.Hs - 1 : sy), sxc = sx < 0 ? 0 : (sx >= p.Ws ? p.Ws - 1 : sx);
            const float4 v = dcs_ld4(sbase + (img + (long)syc * p.Ws + sxc) * cs2);
            sv[k] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            hc += 32;
            if (hc >= HC) { hc -= HC; hr += 1; }
        }
    }
    {
        int hr = (t >> 3) / HC, hc = (t >> 3) % HC;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            if (hr < HR) {
                tile[(2 * q) * PLANE + hr * TP + hc] = make_float2(sv[k].x, sv[k].y);
                tile[(2 * q + 1) * PLANE + hr * TP + hc] = make_float2(sv[k].z, sv[k].w);
            }
            hc += 32;
            if (hc >= HC) { hc -= HC; hr += 1; }
        }
    }
    __syncthreads();

    const int lane = t & 63, wave = t >> 6, bx = lane & 15, by = lane >> 4;      // block (by, bx): source rows 2 by, 2 by + 1 of the tile
    // two accumulators per output, P += w.x * x and Q += w.y * x, combined at the end (no cross-half operand selection: dcs_common.h)
    v2f accp[16], accq[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) { accp[o] = v2f{0.f, 0.f}; accq[o] = v2f{0.f, 0.f}; }
#pragma unroll 1
    for (int cc = 0; cc < CIN / 4; ++cc) {
        const int ci = wave * (CIN / 4) + cc;
        v2f xw[4][4];                                        // the block's window: tile rows 2 by .. 2 by + 3, columns 2 bx .. 2 bx + 3
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4* row = reinterpret_cast<const float4*>(tile + ci * PLANE + (2 * by + r) * TP + 2 * bx);
            const float4 lo = row[0], hi = row[1];
            xw[r][0] = v2f{lo.x, lo.y}; xw[r][1] = v2f{lo.z, lo.w}; xw[r][2] = v2f{hi.x, hi.y}; xw[r][3] = v2f{hi.z, hi.w};
        }
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const float4 w = wf[ci][cls][a][bb];
                    const v2f wx = v2f{w.x, w.y}, wy = v2f{w.z, w.w};
#pragma unroll
                    for (int sy = 0; sy < 2; ++sy)
#pragma unroll
                        for (int sx = 0; sx < 2; ++sx) {     // output (2 sy + py, 2 sx + px) of the 4 x 4
                            const int o = (2 * sy + py) * 4 + 2 * sx + px;
                            const v2f x = xw[sy + a + py][sx + bb + px];
                            accp[o] = __builtin_elementwise_fma(wx, x, accp[o]);
                            accq[o] = __builtin_elementwise_fma(wy, x, accq[o]);
                        }
                }
        }
    }
    __syncthreads();                                         // every wave is done with the patch
    float2* red = tile;                                      // red[wave][o][lane]
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        float px_ = accp[o].x, py_ = accp[o].y, qx_ = accq[o].x, qy_ = accq[o].y;
        asm volatile("" : "+v"(px_), "+v"(py_), "+v"(qx_), "+v"(qy_));          // scalar combine (a vector one becomes a cross-half v_pk_add)
        red[(wave * 16 + o) * 64 + lane] = make_float2(px_ - qy_, py_ + qx_);
    }
    __syncthreads();
    // thread (wave, lane) finishes output row `wave` of block `lane`: four adjacent outputs, the waves' partial sums in order 0..3
    const float br = p.b_r ? p.b_r[0] : 0.f, bi = p.b_i ? p.b_i[0] : 0.f;
    float2 outv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int o = wave * 4 + c;
        float2 a0 = red[(0 * 16 + o) * 64 + lane];
        const float2 a1 = red[(1 * 16 + o) * 64 + lane], a2 = red[(2 * 16 + o) * 64 + lane], a3 = red[(3 * 16 + o) * 64 + lane];
        a0.x = ((a0.x + a1.x) + a2.x) + a3.x; a0.y = ((a0.y + a1.y) + a2.y) + a3.y;
        outv[c] = make_float2(a0.x + (br - bi), a0.y + (br + bi));
    }
    const int sy0 = m0 + 2 * by, sx0 = n0 + 2 * bx;          // the block's first source pixel
    const int oy = 2 * sy0 + wave, ox = 2 * sx0;
    if (oy < 2 * p.Hs && sx0 < p.Ws) {
        float2* yrow = p.y + ((long)b * 2 * p.Hs + oy) * (2 * p.Ws) + ox;
        if (sx0 + 1 < p.Ws) {
            *reinterpret_cast<float4*>(yrow) = make_float4(outv[0].x, outv[0].y, outv[1].x, outv[1].y);
            *reinterpret_cast<float4*>(yrow + 2) = make_float4(outv[2].x, outv[2].y, outv[3].x, outv[3].y);
        } else {                                             // (an odd source width: the block's second column is outside)
            *reinterpret_cast<float4*>(yrow) = make_float4(outv[0].x, outv[0].y, outv[1].x, outv[1].y);
        }
    }
}

}  // namespace

namespace {
// ---- backward: data gradient and weight gradient straight from the cotangent (Round 5) ---------------------------------------------
// The factored backward wrote the nine tap sums of g_y per source pixel as a 16-channel tensor (67 MB at [32,256,256], 7 of the 16
// channels padding) and ran a 1x1 MFMA conv and its weight gradient over it: 352 MB of traffic, six launches, 100 us.  Here both
// kernels stage a haloed 18 x 66 tile of g_y in LDS and form the tap sums of a source pixel s in registers — separable over its 4 x 4
// window of outputs (rows / columns 2s - 1 .. 2s + 2; tap d covers window indices {2 - d, 3 - d}): 12 + 9 complex adds — then
//   data:    g_x[s][ci]   = sum_tap conj(w[ci][tap]) T[s][tap]              written once, 16 bytes per lane
//   weight:  g_w[ci][tap] = sum_s conj(x[s][ci]) T[s][tap],  g_b from sum g_y   per-workgroup partial rows, fixed-order reduce
// so the traffic is g_y + the gradient (84 MB) and g_y + the sources (84 MB).  A thread owns (source pixel, channel quarter): the
// four lanes of a pixel cover its 128 bytes, 16 consecutive pixels a wave.  Complex MACs run as packed FMAs over channel PAIRS
// ({c0, c1} in the two halves) with the tap sum as the broadcast operand: the LDS tile holds every g_y element as {r, r, i, i}, so no
// packed instruction needs a cross-half operand selection (dcs_common.h).
constexpr int GC = 2 * SCT + 2;                              // cotangent tile: 2 R + 2 rows (R source rows) x 66 columns
constexpr int WRT = 2;                                       // source rows per tile of the weight-gradient kernel (one item per thread)
constexpr int kPartRow = 9 * CIN * 2 + 2;                    // [(tap * 16 + ci) * 2 + part], then sum g_r, sum g_i

struct Up1BwdArgs {
    const float2* gy; const float2* wt; void* gx1; void* gx2;          // data gradient
    const void* x1; const void* x2; float* part;                       // weight gradient: part[workgroup][kPartRow]
    int Hs, Ws, C1, C2, ct, tiles_w, tiles, B;
};

// g_y tile rows 2 m0 - 1 .. 2 m0 + 16, columns 2 n0 - 1 .. 2 n0 + 64 (zeros outside the image) as {r, r, i, i}; SUM: the
// thread's share of the sum over the tile's INTERIOR (the outputs this tile owns)
template <int R> struct GTile { static constexpr int GR = 2 * R + 2, NE = GR * GC, NL = (NE + 255) / 256; };
// the thread's elements of a tile, loaded (clamped address, value selected afterwards) ...
template <int R>
__device__ __forceinline__ void up1_gtile_load(const Up1BwdArgs& p, int b, int m0, int n0, int t, float2 (&v)[GTile<R>::NL]) {
    const int Ho = 2 * p.Hs, Wo = 2 * p.Ws;
    const float2* img = p.gy + (long)b * Ho * Wo;
#pragma unroll
    for (int k = 0; k < GTile<R>::NL; ++k) {
        const int i = t + 256 * k, r = i / GC, c = i - r * GC;
        const int oy = 2 * m0 - 1 + r, ox = 2 * n0 - 1 + c;
        const bool in = i < GTile<R>::NE && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
        const int oyc = oy < 0 ? 0 : (oy >= Ho ? Ho - 1 : oy), oxc = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
        const float2 g = img[(long)oyc * Wo + oxc];
        v[k] = in ? g : make_float2(0.f, 0.f);
    }
}
// ... and written to LDS
template <int R, bool SUM>
__device__ __forceinline__ void up1_gtile_store(float4* __restrict__ gt, int t, const float2 (&v)[GTile<R>::NL], float& sr, float& si) {
#pragma unroll
    for (int k = 0; k < GTile<R>::NL; ++k) {
        const int i = t + 256 * k, r = i / GC, c = i - r * GC;
        if (i < GTile<R>::NE) gt[i] = make_float4(v[k].x, v[k].x, v[k].y, v[k].y);
        if (SUM && i < GTile<R>::NE && r >= 1 && r <= 2 * R && c >= 1 && c <= 2 * SCT) { sr += v[k].x; si += v[k].y; }
    }
}

// the nine tap sums of the source pixel whose window starts at tile row 2 m, column 2 n: Tr[dy * 3 + dx] = {Re, Re}, Ti = {Im, Im}
__device__ __forceinline__ void up1_tapsums(const float4* __restrict__ gt, int m, int n, v2f (&Tr)[9], v2f (&Ti)[9]) {
    v2f hr[4][3], hi[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float4* row = gt + (2 * m + a) * GC + 2 * n;
        const float4 g0 = row[0], g1 = row[1], g2 = row[2], g3 = row[3];
        const v2f r0 = v2f{g0.x, g0.y}, r1 = v2f{g1.x, g1.y}, r2 = v2f{g2.x, g2.y}, r3 = v2f{g3.x, g3.y};
        const v2f i0 = v2f{g0.z, g0.w}, i1 = v2f{g1.z, g1.w}, i2 = v2f{g2.z, g2.w}, i3 = v2f{g3.z, g3.w};
        hr[a][0] = r2 + r3; hr[a][1] = r1 + r2; hr[a][2] = r0 + r1;
        hi[a][0] = i2 + i3; hi[a][1] = i1 + i2; hi[a][2] = i0 + i1;
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            Tr[dy * 3 + dx] = hr[2 - dy][dx] + hr[3 - dy][dx];
            Ti[dy * 3 + dx] = hi[2 - dy][dx] + hi[3 - dy][dx];
        }
}

// OT: element type of the gradient — float, or bf16 where the activations live in bf16
template <typename OT>
__global__ __launch_bounds__(256) void cconv_up1_bwd_data_kernel(Up1BwdArgs p) {
    __shared__ __attribute__((aligned(16))) float4 gt[GTile<SRT>::NE];
    __shared__ __attribute__((aligned(16))) float4 wl[9][CIN / 2];       // [tap][channel pair]: {w_r(c0), w_r(c1), w_i(c0), w_i(c1)}
    const int t = threadIdx.x, b = blockIdx.y;
    const int m0 = ((int)blockIdx.x / p.tiles_w) * SRT, n0 = ((int)blockIdx.x % p.tiles_w) * SCT;
    const int q = t & 3, pl = t >> 2;
    float2 gv[GTile<SRT>::NL];
    up1_gtile_load<SRT>(p, b, m0, n0, t, gv);
    if (t < 9 * (CIN / 2)) {
        const int tap = t / (CIN / 2), pr = t % (CIN / 2);
        const float2 w0 = p.wt[(2 * pr) * p.ct + tap], w1 = p.wt[(2 * pr + 1) * p.ct + tap];
        wl[tap][pr] = make_float4(w0.x, w1.x, w0.y, w1.y);
    }
    float sr = 0.f, si = 0.f;
    up1_gtile_store<SRT, false>(gt, t, gv, sr, si);
    __syncthreads();
    v2f wr[9][2], wi[9][2];                                  // this quarter's four channels as two pairs
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const float4 w = wl[tap][2 * q + cp];
            wr[tap][cp] = v2f{w.x, w.y}; wi[tap][cp] = v2f{w.z, w.w};
        }
    const int c0 = 4 * q;
    OT* const base = c0 < p.C1 ? (OT*)p.gx1 + 2 * c0 : (OT*)p.gx2 + 2 * (c0 - p.C1);
    const int cs = c0 < p.C1 ? p.C1 : p.C2;
#pragma unroll 1
    for (int k = 0; k < SRT * SCT / 64; ++k) {
        const int pi = k * 64 + pl, m = pi / SCT, n = pi % SCT;
        v2f Tr[9], Ti[9];
        up1_tapsums(gt, m, n, Tr, Ti);
        v2f ar[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}}, ai[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {                 // conj(w) T = (w_r T_r + w_i T_i) + j (w_r T_i - w_i T_r)
                ar[cp] = __builtin_elementwise_fma(wr[tap][cp], Tr[tap], ar[cp]);
                ar[cp] = __builtin_elementwise_fma(wi[tap][cp], Ti[tap], ar[cp]);
                ai[cp] = __builtin_elementwise_fma(wr[tap][cp], Ti[tap], ai[cp]);
                ai[cp] = __builtin_elementwise_fma(-wi[tap][cp], Tr[tap], ai[cp]);
            }
        const int sy = m0 + m, sx = n0 + n;
        if (sy < p.Hs && sx < p.Ws) {
            OT* dst = base + (((long)b * p.Hs + sy) * p.Ws + sx) * cs * 2;
            dcs_st4(dst, make_float4(ar[0].x, ai[0].x, ar[0].y, ai[0].y));
            dcs_st4(dst + 4, make_float4(ar[1].x, ai[1].x, ar[1].y, ai[1].y));
        }
    }
}

// IT: element type of the two sources.  Persistent: workgroup w walks tiles w, w + gridDim.x, ... of the B x tiles grid; a tile is
// WRT = 2 source rows x 32 columns — ONE item per thread, so that the next tile's operands (two g_y elements, two 16-byte source loads)
// are a dozen registers held across the current tile's arithmetic (with the 8 x 32 tile of the data-gradient kernel the fully unrolled
// four items took 262 VGPRs: one wave per SIMD, 85 us).
template <typename IT>
__global__ __launch_bounds__(256) void cconv_up1_bwd_weight_kernel(Up1BwdArgs p) {
    __shared__ __attribute__((aligned(16))) float4 gt[GTile<WRT>::NE];
    __shared__ float red[4][4][74];
    const int t = threadIdx.x, q = t & 3, pl = t >> 2, lane = t & 63, wave = t >> 6;
    const int c0 = 4 * q;
    const IT* const base = c0 < p.C1 ? (const IT*)p.x1 + 2 * c0 : (const IT*)p.x2 + 2 * (c0 - p.C1);
    const int cs = c0 < p.C1 ? p.C1 : p.C2;
    v2f wR[9][2], wI[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) { wR[tap][cp] = v2f{0.f, 0.f}; wI[tap][cp] = v2f{0.f, 0.f}; }
    float sr = 0.f, si = 0.f;
    const int tiles_h = (p.Hs + WRT - 1) / WRT, tiles = tiles_h * p.tiles_w, total = tiles * p.B;
    const int m = pl / SCT, n = pl % SCT;
    float2 gv[GTile<WRT>::NL];
    float4 xa, xb;
    auto fetch = [&](int tile) {
        const int b = tile / tiles, tl = tile - b * tiles;
        const int m0 = (tl / p.tiles_w) * WRT, n0 = (tl % p.tiles_w) * SCT;
        up1_gtile_load<WRT>(p, b, m0, n0, t, gv);
        const int sy = m0 + m, sx = n0 + n;
        const bool in = sy < p.Hs && sx < p.Ws;
        const int syc = sy < p.Hs ? sy : p.Hs - 1, sxc = sx < p.Ws ? sx : p.Ws - 1;
        const IT* src = base + (((long)b * p.Hs + syc) * p.Ws + sxc) * cs * 2;
        const float4 a0 = dcs_ld4(src), a1 = dcs_ld4(src + 4);
        xa = in ? a0 : make_float4(0.f, 0.f, 0.f, 0.f);
        xb = in ? a1 : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    int tile = blockIdx.x;
    if (tile < total) fetch(tile);
#pragma unroll 1
    for (; tile < total; tile += gridDim.x) {
        __syncthreads();                                     // the previous tile's readers are done
        up1_gtile_store<WRT, true>(gt, t, gv, sr, si);
        const float4 x0 = xa, x1 = xb;
        __syncthreads();
        if (tile + (int)gridDim.x < total) fetch(tile + gridDim.x);      // in flight during this tile's arithmetic
        v2f Tr[9], Ti[9];
        up1_tapsums(gt, m, n, Tr, Ti);
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {                     // conj(x) T = (x_r T_r + x_i T_i) + j (x_r T_i - x_i T_r)
            const float4 xv = cp ? x1 : x0;
            const v2f xr = v2f{xv.x, xv.z}, xi = v2f{xv.y, xv.w};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                wR[tap][cp] = __builtin_elementwise_fma(xr, Tr[tap], wR[tap][cp]);
                wR[tap][cp] = __builtin_elementwise_fma(xi, Ti[tap], wR[tap][cp]);
                wI[tap][cp] = __builtin_elementwise_fma(xr, Ti[tap], wI[tap][cp]);
                wI[tap][cp] = __builtin_elementwise_fma(-xi, Tr[tap], wI[tap][cp]);
            }
        }
    }
    // lanes of one quarter (lane % 4) summed over the wave's 16 pixels, then the four waves in a fixed order
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            float v4[4] = {wR[tap][cp].x, wI[tap][cp].x, wR[tap][cp].y, wI[tap][cp].y};     // (c0: re, im), (c1: re, im)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = v4[j];
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
                if (lane < 4) red[wave][q][(tap * 2 + cp) * 4 + j] = v;
            }
        }
    {
        float a = sr, c = si;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
        if (lane == 0) { red[wave][0][72] = a; red[wave][0][73] = c; }
    }
    __syncthreads();
    float* row = p.part + (long)blockIdx.x * kPartRow;
    for (int o = t; o < kPartRow; o += 256) {
        float s;
        if (o < 9 * CIN * 2) {
            const int tap = o / (CIN * 2), ci = (o % (CIN * 2)) >> 1, part = o & 1;
            const int qq = ci >> 2, cp = (ci & 3) >> 1, half = ci & 1, idx = (tap * 2 + cp) * 4 + half * 2 + part;
            s = red[0][qq][idx] + red[1][qq][idx] + red[2][qq][idx] + red[3][qq][idx];
        } else {
            const int idx = 72 + (o - 9 * CIN * 2);
            s = red[0][0][idx] + red[1][0][idx] + red[2][0][idx] + red[3][0][idx];
        }
        row[o] = s;
    }
}

// partial rows -> the parameter gradients: g_w[ci][0][ky][kx] (+)= row sums of tap 8 - (ky * 3 + kx) (the correlation kernel is the
// flipped transposed-conv kernel: dcs_pack_tap_rows), g_b_r = S_r + S_i, g_b_i = S_i - S_r (written).  16 outputs per workgroup,
// 16 row slices each (a slice's loads are independent: unrolled), combined in a fixed order in fp64.
constexpr int kFinOut = 16, kFinSl = 256 / kFinOut;
__global__ __launch_bounds__(256) void cconv_up1_wgrad_final_kernel(const float* __restrict__ part, int nrows, float* __restrict__ gw_r,
                                                                    float* __restrict__ gw_i, float* __restrict__ gb_r,
                                                                    float* __restrict__ gb_i, int accumulate) {
    __shared__ double red[kFinSl][kFinOut];
    const int t = threadIdx.x, ol = t % kFinOut, sl = t / kFinOut, o = blockIdx.x * kFinOut + ol;
    double s = 0;
    if (o < kPartRow) {
#pragma unroll 8
        for (int r = sl; r < nrows; r += kFinSl) s += (double)part[(long)r * kPartRow + o];
    }
    red[sl][ol] = s;
    __syncthreads();
    if (sl == 0) {
        double a = red[0][ol];
#pragma unroll
        for (int k = 1; k < kFinSl; ++k) a += red[k][ol];
        red[0][ol] = a;
    }
    __syncthreads();
    if (sl != 0 || o >= kPartRow) return;
    if (o < 9 * CIN * 2) {
        const int tap = o / (CIN * 2), ci = (o % (CIN * 2)) >> 1, part_ = o & 1;
        float* dst = (part_ ? gw_i : gw_r) + ci * 9 + (8 - tap);
        const float v = (float)red[0][ol];
        if (accumulate) *dst += v; else *dst = v;
    } else if (o == 9 * CIN * 2 && gb_r) {                   // (both sums live in this workgroup: 288 and 289 share a block of 16)
        const double Sr = red[0][ol], Si = red[0][ol + 1];
        gb_r[0] = (float)(Sr + Si); gb_i[0] = (float)(Si - Sr);
    }
}

}  // namespace

// x1 complex[B][Hs][Ws][C1], x2 complex[B][Hs][Ws][C2] (C1 + C2 = 16, both even); wt: the tap-rows panel of
// dcs_pack_tap_rows (complex[16][ct], ct >= 9, column tap = dy*3 + dx of the CORRELATION kernel); b_r / b_i: the layer's
// two real bias scalars (both or neither); y complex[B][2 Hs][2 Ws].
static int up2_single_impl(const void* x1, const void* x2, bool bf16_in, const float* wt, const float* b_r, const float* b_i,
                           float* y, int B, int Hs, int Ws, int C1, int C2, int ct, dcs_stream_t stream) {
    if (!x1 || !wt || !y || B <= 0 || B > 65535 || Hs <= 0 || Ws <= 0 || C1 <= 0 || C2 < 0 || C1 + C2 != CIN || (C1 & 1) ||
        (C2 & 1) || ct < 9 || ((C2 > 0) != (x2 != nullptr)) || ((b_r == nullptr) != (b_i == nullptr)))
        return DCS_ERR_BADARG;
    Up1Args p;
    p.x1 = x1; p.x2 = x2; p.wt = (const float2*)wt; p.b_r = b_r; p.b_i = b_i; p.y = (float2*)y;
    p.Hs = Hs; p.Ws = Ws; p.C1 = C1; p.C2 = C2; p.ct = ct;
    p.tiles_w = (Ws + SCT - 1) / SCT;
    p.tiles = p.tiles_w * ((Hs + SRT - 1) / SRT);
    if (bf16_in) DCS_LAUNCH(cconv_up1_kernel<unsigned short>, dim3(p.tiles, B), dim3(256), 0, dcs_stream(stream), p);
    else DCS_LAUNCH(cconv_up1_kernel<float>, dim3(p.tiles, B), dim3(256), 0, dcs_stream(stream), p);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_cconv_up2_single_fwd(const float* x1, const float* x2, const float* wt, const float* b_r, const float* b_i,
                                        float* y, int B, int Hs, int Ws, int C1, int C2, int ct, dcs_stream_t stream) {
    return up2_single_impl(x1, x2, false, wt, b_r, b_i, y, B, Hs, Ws, C1, C2, ct, stream);
}

// the two sources in bf16 (activations stored in bf16: BASELINE configs[4]); weights, bias and the result fp32
extern "C" int dcs_cconv_up2_single_fwd_h(const unsigned short* x1, const unsigned short* x2, const float* wt, const float* b_r,
                                          const float* b_i, float* y, int B, int Hs, int Ws, int C1, int C2, int ct,
                                          dcs_stream_t stream) {
    return up2_single_impl(x1, x2, true, wt, b_r, b_i, y, B, Hs, Ws, C1, C2, ct, stream);
}

// ---- backward entry points --------------------------------------------------------------------------------------------------------
static bool up2_bwd_geom_ok(int B, int Hs, int Ws, int C1, int C2) {
    return B > 0 && B <= 65535 && Hs > 0 && Ws > 0 && C1 > 0 && C2 >= 0 && C1 + C2 == CIN && (C1 % 4) == 0 && (C2 % 4) == 0 &&
           (long)B * Hs * Ws * CIN * 2 < (1L << 40);
}
static void up2_bwd_tiles(Up1BwdArgs& p, int B, int Hs, int Ws, int C1, int C2) {
    p.Hs = Hs; p.Ws = Ws; p.C1 = C1; p.C2 = C2; p.B = B;
    p.tiles_w = (Ws + SCT - 1) / SCT;
    p.tiles = p.tiles_w * ((Hs + SRT - 1) / SRT);
}

static int up2_single_bwd_data_impl(const float* gy, const float* wt, void* gx1, void* gx2, bool bf16_out, int B, int Hs, int Ws,
                                    int C1, int C2, int ct, dcs_stream_t stream) {
    if (!gy || !wt || !gx1 || !up2_bwd_geom_ok(B, Hs, Ws, C1, C2) || ct < 9 || ((C2 > 0) != (gx2 != nullptr))) return DCS_ERR_BADARG;
    Up1BwdArgs p{};
    p.gy = (const float2*)gy; p.wt = (const float2*)wt; p.gx1 = gx1; p.gx2 = gx2; p.ct = ct;
    up2_bwd_tiles(p, B, Hs, Ws, C1, C2);
    if (bf16_out) DCS_LAUNCH(cconv_up1_bwd_data_kernel<unsigned short>, dim3(p.tiles, B), dim3(256), 0, dcs_stream(stream), p);
    else DCS_LAUNCH(cconv_up1_bwd_data_kernel<float>, dim3(p.tiles, B), dim3(256), 0, dcs_stream(stream), p);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_cconv_up2_single_bwd_data(const float* gy, const float* wt, float* gx1, float* gx2, int B, int Hs, int Ws, int C1,
                                             int C2, int ct, dcs_stream_t stream) {
    return up2_single_bwd_data_impl(gy, wt, gx1, gx2, false, B, Hs, Ws, C1, C2, ct, stream);
}
extern "C" int dcs_cconv_up2_single_bwd_data_h(const float* gy, const float* wt, unsigned short* gx1, unsigned short* gx2, int B, int Hs,
                                               int Ws, int C1, int C2, int ct, dcs_stream_t stream) {
    return up2_single_bwd_data_impl(gy, wt, gx1, gx2, true, B, Hs, Ws, C1, C2, ct, stream);
}

constexpr int kUp1WgradMaxGroups = 768;                      // three per CU (the kernel's registers allow three waves per SIMD): all resident
extern "C" long dcs_cconv_up2_single_bwd_weight_workspace_bytes(void) { return (long)kUp1WgradMaxGroups * kPartRow * (long)sizeof(float); }

static int up2_single_bwd_weight_impl(const float* gy, const void* x1, const void* x2, bool bf16_in, float* gw_r, float* gw_i,
                                      float* gb_r, float* gb_i, int accumulate, void* workspace, long workspace_bytes, int B, int Hs,
                                      int Ws, int C1, int C2, dcs_stream_t stream) {
    if (!gy || !x1 || !gw_r || !gw_i || !up2_bwd_geom_ok(B, Hs, Ws, C1, C2) || ((C2 > 0) != (x2 != nullptr)) ||
        ((gb_r == nullptr) != (gb_i == nullptr)))
        return DCS_ERR_BADARG;
    if (!workspace || workspace_bytes < dcs_cconv_up2_single_bwd_weight_workspace_bytes()) return DCS_ERR_WORKSPACE;
    Up1BwdArgs p{};
    p.gy = (const float2*)gy; p.x1 = x1; p.x2 = x2; p.part = (float*)workspace;
    up2_bwd_tiles(p, B, Hs, Ws, C1, C2);
    const long total = (long)p.tiles_w * ((Hs + WRT - 1) / WRT) * B;
    if (total > 0x7fffffffL) return DCS_ERR_BADARG;
    const int groups = (int)(total < kUp1WgradMaxGroups ? total : kUp1WgradMaxGroups);
    if (bf16_in) DCS_LAUNCH(cconv_up1_bwd_weight_kernel<unsigned short>, dim3(groups), dim3(256), 0, dcs_stream(stream), p);
    else DCS_LAUNCH(cconv_up1_bwd_weight_kernel<float>, dim3(groups), dim3(256), 0, dcs_stream(stream), p);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(cconv_up1_wgrad_final_kernel, dim3((kPartRow + kFinOut - 1) / kFinOut), dim3(256), 0, dcs_stream(stream), (const float*)workspace,
               groups, gw_r, gw_i, gb_r, gb_i, accumulate);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_cconv_up2_single_bwd_weight(const float* gy, const float* x1, const float* x2, float* gw_r, float* gw_i, float* gb_r,
                                               float* gb_i, int accumulate, void* workspace, long workspace_bytes, int B, int Hs,
                                               int Ws, int C1, int C2, dcs_stream_t stream) {
    return up2_single_bwd_weight_impl(gy, x1, x2, false, gw_r, gw_i, gb_r, gb_i, accumulate, workspace, workspace_bytes, B, Hs, Ws, C1,
                                      C2, stream);
}
extern "C" int dcs_cconv_up2_single_bwd_weight_h(const float* gy, const unsigned short* x1, const unsigned short* x2, float* gw_r,
                                                 float* gw_i, float* gb_r, float* gb_i, int accumulate, void* workspace,
                                                 long workspace_bytes, int B, int Hs, int Ws, int C1, int C2, dcs_stream_t stream) {
    return up2_single_bwd_weight_impl(gy, x1, x2, true, gw_r, gw_i, gb_r, gb_i, accumulate, workspace, workspace_bytes, B, Hs, Ws, C1,
                                      C2, stream);
}
