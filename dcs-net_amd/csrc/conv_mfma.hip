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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CHUNK = 8;       // complex input channels staged per LDS pass (= 2 k-groups of 4)
constexpr int PIX = 20;        // LDS floats per patch pixel: 16 + 4 pad

struct MArgs {
    conv::Args c;              // c.Hout / c.Wout: FULL output extent (addressing); c.sf / c.st: stride in class space
    const float* bm;
    float* y2;                 // optional second output: columns >= nsplit go here (g_x1 | g_x2 of a cat)
    int nsplit;
    int TH, TW, N, KG, NT;     // tile shape (TH*TW = pixels per WG), N = 2*Cout, KG = Cin/4, NT = ceil(N/32)
    int ncls, os_f, os_t;      // output-parity classes (blockIdx.z): pixel (oy, ox) of class c is stored at
    conv::Cls cls[4];          //   (oy*os_f + oo_f, ox*os_t + oo_t) and has its own sub-kernel / padding / panel
};

// wave grid: WAVES_N waves along N, 4/WAVES_N along M; each wave owns WM x WN tiles of 32x32
template <int WAVES_N, int WM, int WN>
__global__ __launch_bounds__(256) void cconv_mfma_kernel(MArgs m) {
    extern __shared__ __attribute__((aligned(16))) float patch[];      // [rows*cols][PIX]
    const conv::Args& a = m.c;
    const conv::Cls& k = m.cls[blockIdx.z];
    // pixels per workgroup = (4 / WAVES_N) * WM * 32 = TH * TW
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int kk = lane >> 5, li = lane & 31;

    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int b = blockIdx.x / tiles_per_img, tile_id = blockIdx.x % tiles_per_img;
    const int oy0 = (tile_id / a.tiles_w) * m.TH, ox0 = (tile_id % a.tiles_w) * m.TW;
    if (oy0 >= k.Hc || ox0 >= k.Wc) return;                           // tile outside this (smaller) class
    const int vy0 = oy0 * a.sf - k.pad_f, vx0 = ox0 * a.st - k.pad_t;
    const int nt0 = (blockIdx.y * WAVES_N + wn) * WN;                  // first 32-column tile of this wave
    const int Cin = a.C1 + a.C2;
    const int ntaps = k.kh * k.kw;
    const int cols = (m.TW - 1) * a.st + k.kw, rows = (m.TH - 1) * a.sf + k.kh;

    // LDS float offset of this lane's pixel for each of its m-tiles (tap (0,0), k-group 0)
    int pixoff[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int pi = (wm * WM + i) * 32 + li;
        pixoff[i] = (((pi / m.TW) * a.sf) * cols + (pi % m.TW) * a.st) * PIX + kk * 4;
    }
    const float* bbase = m.bm + k.bm_off + ((long)nt0 * 64 + lane) * 4;
    const long b_tap_stride = (long)m.KG * m.NT * 256, b_kg_stride = (long)m.NT * 256;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int n_chunks = Cin / CHUNK;
    const int npix = rows * cols;
    for (int ch = 0; ch < n_chunks; ++ch) {
        __syncthreads();                                               // previous chunk fully consumed
        for (int idx = t; idx < npix * 4; idx += 256) {                // 4 float4 (2 complex each) per pixel
            const int q = idx & 3, px = idx >> 2;
            const int ix = px % cols, iy = px / cols;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            long sp;
            if (conv::src_pixel(a, b, vy0 + iy, vx0 + ix, &sp)) {
                const int c = ch * CHUNK + 2 * q;
                const float2* src = (c < a.C1) ? a.x1 + sp * a.C1 + c : a.x2 + sp * a.C2 + (c - a.C1);
                v = *reinterpret_cast<const float4*>(src);
            }
            *reinterpret_cast<float4*>(patch + px * PIX + q * 4) = v;
        }
        __syncthreads();
        const int iters = ntaps * 2;                                   // (tap, k-group within chunk)
        float4 bn[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j)
            bn[j] = *reinterpret_cast<const float4*>(bbase + (long)(ch * 2) * b_kg_stride + j * 256);
        for (int it = 0; it < iters; ++it) {
            const int tap = it >> 1, g = it & 1;
            float4 bf[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = bn[j];
            if (it + 1 < iters) {                                      // prefetch the next B fragments
                const int tap2 = (it + 1) >> 1, g2 = (it + 1) & 1;
                const float* bp = bbase + tap2 * b_tap_stride + (long)(ch * 2 + g2) * b_kg_stride;
#pragma unroll
                for (int j = 0; j < WN; ++j) bn[j] = *reinterpret_cast<const float4*>(bp + j * 256);
            }
            const int tapoff = ((tap / k.kw) * cols + (tap % k.kw)) * PIX + g * 8;
            float4 af[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const float4*>(patch + pixoff[i] + tapoff);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }

    // epilogue: C/D map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const float* biasf = reinterpret_cast<const float*>(a.bias);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = (nt0 + j) * 32 + li;
        if (n >= m.N) continue;                                        // zero-padded columns of a 16-wide N
        const float bv = biasf ? biasf[n] : 0.f;
        // columns >= nsplit belong to the second tensor of a concatenation
        const bool second = m.y2 != nullptr && n >= m.nsplit;
        float* yf = second ? m.y2 : reinterpret_cast<float*>(a.y);
        const int width = m.y2 == nullptr ? m.N : (second ? m.N - m.nsplit : m.nsplit);
        const int col = second ? n - m.nsplit : n;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
                const int pi = (wm * WM + i) * 32 + row;
                const int oy = oy0 + pi / m.TW, ox = ox0 + pi % m.TW;
                if (oy < k.Hc && ox < k.Wc)
                    yf[(((long)b * a.Hout + oy * m.os_f + k.oo_f) * a.Wout + ox * m.os_t + k.oo_t) * width + col] =
                        dcs_act(acc[i][j][r] + bv, a.act);
            }
        }
    }
}

// bm[tap][kg][nt][kk][j][e]: e = 0..3 -> (ci = 4kg+2kk, re), (.., im), (ci+1, re), (ci+1, im); column n = nt*32+j
__global__ void pack_mfma_kernel(const float2* __restrict__ wp, float4* __restrict__ bm, int Cout, int Cin, int taps) {
    const int KG = Cin / 4, NT = (2 * Cout + 31) / 32;
    const long total = (long)taps * KG * NT * 64;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 31), kk = (int)((i >> 5) & 1);
    long r = i >> 6;
    const int nt = (int)(r % NT); r /= NT;
    const int kg = (int)(r % KG);
    const int tap = (int)(r / KG);
    const int n = nt * 32 + j, co = n >> 1, im = n & 1;
    if (co >= Cout) { bm[i] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const int ci = 4 * kg + 2 * kk;
    const float2 w0 = wp[((long)tap * Cin + ci) * Cout + co], w1 = wp[((long)tap * Cin + ci + 1) * Cout + co];
    // column (co, re): [ w_r, -w_i ] ; column (co, im): [ w_i, w_r ]
    bm[i] = im ? make_float4(w0.y, w0.x, w1.y, w1.x) : make_float4(w0.x, -w0.y, w1.x, -w1.y);
}

template <int WAVES_N, int WM, int WN>
int launch(MArgs& m, hipStream_t stream) {
    const conv::Args& a = m.c;
    int kh = 0, kw = 0;
    for (int c = 0; c < m.ncls; ++c) { kh = m.cls[c].kh > kh ? m.cls[c].kh : kh; kw = m.cls[c].kw > kw ? m.cls[c].kw : kw; }
    const size_t lds = (size_t)((m.TH - 1) * a.sf + kh) * ((m.TW - 1) * a.st + kw) * PIX * sizeof(float);
    if (lds > 150 * 1024) return DCS_ERR_BADARG;
    auto fn = cconv_mfma_kernel<WAVES_N, WM, WN>;
    if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
    dim3 grid(a.tiles_w * a.tiles_h * a.B, m.NT / (WAVES_N * WN), m.ncls);
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, stream, m);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

}  // namespace

int dcs_conv_mfma_pack(const float* wp_direct, float* bm, int Cout, int Cin, int taps, hipStream_t stream) {
    if (!conv::mfma_ok(Cin, Cout)) return DCS_ERR_BADARG;
    const long total = (long)taps * (Cin / 4) * ((2 * Cout + 31) / 32) * 64;
    hipLaunchKernelGGL(pack_mfma_kernel, dim3(dcs_cdiv(total, 256)), dim3(256), 0, stream, (const float2*)wp_direct,
                       (float4*)bm, Cout, Cin, taps);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// a: geometry with the FULL output extent in Hout/Wout; cls[0..ncls): output-parity classes (class-space
// extent Hc x Wc, sub-kernel size, padding, panel offset); y2/nsplit: optional column split of the output
int dcs_conv_mfma_launch_classes(conv::Args& a, const float* bm, int ncls, const conv::Cls* cls, int os_f, int os_t,
                                 float* y2, int nsplit, hipStream_t stream) {
    const int Cin = a.C1 + a.C2;
    if (!conv::mfma_ok(Cin, a.Cout) || (a.C1 & 1) || a.Hout <= 0 || a.Wout <= 0 || ncls < 1 || ncls > 4)
        return DCS_ERR_BADARG;
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
    if (Hc <= 0 || Wc <= 0) return DCS_ERR_BADARG;
    m.N = 2 * a.Cout; m.KG = Cin / 4; m.NT = (m.N + 31) / 32;
    // candidate workgroup tiles (pixels x columns); take the largest that still yields >= 256 workgroups
    // (one per CU), else the one with the most workgroups: deep layers at small batch have few pixels
    struct Cand { int bm, bn; };
    const Cand cands[] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}};
    auto shape = [&](int bmp, int* th, int* tw) {
        if (bmp == 128) { if (Hc >= 8) { *th = 8; *tw = 16; } else if (Hc >= 4) { *th = 4; *tw = 32; } else { *th = 2; *tw = 64; } }
        else { if (Hc >= 4) { *th = 4; *tw = 16; } else { *th = 2; *tw = 32; } }
    };
    int best = -1; long best_blocks = -1;
    for (int i = 0; i < 4; ++i) {
        if (m.NT % (cands[i].bn / 32) != 0) continue;
        int th, tw;
        shape(cands[i].bm, &th, &tw);
        const long blocks = (long)((Wc + tw - 1) / tw) * ((Hc + th - 1) / th) * a.B * (m.NT / (cands[i].bn / 32)) * ncls;
        if (blocks >= 256) { best = i; break; }
        if (blocks > best_blocks) { best_blocks = blocks; best = i; }
    }
    if (best < 0) return DCS_ERR_BADARG;
    shape(cands[best].bm, &m.TH, &m.TW);
    m.c.tiles_w = (Wc + m.TW - 1) / m.TW;
    m.c.tiles_h = (Hc + m.TH - 1) / m.TH;
    switch (best) {
        case 0: return launch<2, 2, 2>(m, stream);      // 128 x 128
        case 1: return launch<2, 2, 1>(m, stream);      // 128 x 64
        case 2: return launch<2, 1, 1>(m, stream);      //  64 x 64
        default: return launch<1, 1, 1>(m, stream);     // 128 x 32
    }
}

// single class: the plain convolution described by `a` (Hout/Wout set)
int dcs_conv_mfma_launch(conv::Args& a, const float* bm, hipStream_t stream) {
    conv::Cls c;
    c.kh = a.kh; c.kw = a.kw; c.pad_f = a.pad_f; c.pad_t = a.pad_t; c.oo_f = 0; c.oo_t = 0;
    c.Hc = a.Hout; c.Wc = a.Wout; c.bm_off = 0;
    return dcs_conv_mfma_launch_classes(a, bm, 1, &c, 1, 1, nullptr, 0, stream);
}
