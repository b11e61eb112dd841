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
    const float* gy; float2* slab_w; float* slab_b;
    int n_slabs, total_tiles, co_blocks, TH, TW, twshift;
    unsigned cols_magic;       // ceil(2^32 / c.cols): patch pixel index / cols by one multiply-high (patches are < 2^16 pixels)
    // output-parity classes of an upsample-folded conv (blockIdx.z): class pixel (oy, ox) is g_Y pixel
    // (oy*os_f + oo_f, ox*os_t + oo_t) and reads the SOURCE-resolution input with its own padding; one class = the
    // plain convolution (os = 1, oo = 0, pad = c.pad).  c.Hout / c.Wout: class-space extent (tiling);
    // Hy / Wy: full g_Y extent (addressing).
    int ncls, os_f, os_t, Hy, Wy;
    int pad_f[4], pad_t[4], oo_f[4], oo_t[4];
};

// MT row tiles (8 output channels each) per wave; WS waves split the tile's PIXELS (k-steps) and
// 4/WS waves split the output channels, so layers with few output channels still fill all four
// SIMDs: the pixel partials are combined through LDS once, after the last tile.
// Two workgroups per CU (<= 256 VGPR + AGPR per lane) whenever the accumulator set allows it: one gathers while
// the other runs its MFMAs.
template <int KH, int KW, int MT, int WS>
__global__ __launch_bounds__(256, (MT * KH * KW * 4 <= 160 && !(KH == 5 && WS == 4) ? 2 : 1)) void cconv_wgrad_mfma_kernel(WArgs w) {
    constexpr int TAPS = KH * KW;
    const int cls = blockIdx.z;
    constexpr int WCO = 4 / WS;
    constexpr bool GL = MT == 1 && WS >= 2;             // g_Y tile through LDS (<= 16 output channels per workgroup)
    constexpr int GQ = WCO * 4;                         // float4 per pixel of the g_Y block (WCO*8 channels x 2 floats)
    constexpr int GP = WCO * 16 + (WCO == 1 ? 0 : 16);  // float pitch per pixel: pixel stride == 16 banks (mod 32)
    extern __shared__ __attribute__((aligned(16))) float patch[];      // [rows*cols][PIX]
    const conv::Args& a = w.c;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int ci0 = (blockIdx.y / w.co_blocks) * CHUNK;
    const int part = wave % WS;                                       // which share of the pixels
    const int co0 = (blockIdx.y % w.co_blocks) * (WCO * MT * 8) + (wave / WS) * (MT * 8);   // first output channel
    const int N1 = 2 * a.Cout;                                        // floats per gY pixel
    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int npix = a.rows * a.cols;

    f32x4 acc[MT][TAPS];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) acc[i][tp] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) bsum[i] = 0.f;

    // column (row of D) this lane feeds for each of its row tiles; masked beyond Cout
    bool colok[MT];
    int gcol[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        gcol[i] = 2 * (co0 + i * 8) + li;
        colok[i] = (co0 + i * 8 + (li >> 1)) < a.Cout;
    }

    for (int tl = blockIdx.x; tl < w.total_tiles; tl += w.n_slabs) {
        const int b = tl / tiles_per_img, tile_id = tl % tiles_per_img;
        const int oy0 = (tile_id / a.tiles_w) * w.TH, ox0 = (tile_id % a.tiles_w) * w.TW;
        const int vy0 = oy0 * a.sf - w.pad_f[cls], vx0 = ox0 * a.st - w.pad_t[cls];
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
                const float2* src = (ci0 < a.C1) ? a.x1 + sp * a.C1 + ci0 : a.x2 + sp * a.C2 + (ci0 - a.C1);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const float4*>(src)[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(patch + px * PIX + q * 4) = v[q];
        }
        if (GL) {
            // few output channels (<= 16 per workgroup): a k-step is only TAPS*32 cycles of MFMA, far less than the L2 round trip
            // of its g_Y fragment — stage the tile's g_Y block [128 pixels][co_per_block*2] in LDS with coalesced
            // 16-byte loads instead (zeros outside the map / beyond Cout), and feed the A fragments from there
            float* gt = patch + npix * PIX;
            const int cb0 = (blockIdx.y % w.co_blocks) * (WCO * 8);           // first output channel of the block
            for (int idx = t; idx < BMP * GQ; idx += 256) {
                const int q = idx % GQ, p = idx / GQ;
                const int oy = oy0 + (p >> w.twshift), ox = ox0 + (p & (w.TW - 1));
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (oy < a.Hout && ox < a.Wout && 2 * cb0 + 4 * q < N1)
                    v = *reinterpret_cast<const float4*>(w.gy + (((long)b * w.Hy + oy * w.os_f + w.oo_f[cls]) * w.Wy +
                                                                 ox * w.os_t + w.oo_t[cls]) * N1 + 2 * cb0 + 4 * q);
                *reinterpret_cast<float4*>(gt + p * GP + q * 4) = v;
            }
        }
        __syncthreads();
        // gY fragments come from L2 with ~1-2 us latency and a k-step is only MT*TAPS*32 cycles of MFMA: keep the
        // next RING k-steps' loads in flight (static register ring; the k-step loop is unrolled over it)
        constexpr int RING = MT >= 4 ? 2 : 4;                          // BMP/4/WS is 32, 16 or 8
        float afr[RING][MT];
        auto load_g = [&](int ks, float* dst) {
            if (GL) {
                dst[0] = patch[npix * PIX + (ks * 4 + lk) * GP + (wave / WS) * 16 + li];
                return;
            }
            const int p = ks * 4 + lk;
            const int oy = oy0 + (p >> w.twshift), ox = ox0 + (p & (w.TW - 1));
            const bool inb = oy < a.Hout && ox < a.Wout;
            const float* gp = w.gy + (((long)b * w.Hy + (inb ? oy * w.os_f + w.oo_f[cls] : 0)) * w.Wy +
                                      (inb ? ox * w.os_t + w.oo_t[cls] : 0)) * N1;
#pragma unroll
            for (int i = 0; i < MT; ++i) dst[i] = (inb && colok[i]) ? gp[gcol[i]] : 0.f;
        };
#pragma unroll
        for (int r = 0; r < RING; ++r) load_g(part + r * WS, afr[r]);
        for (int ks0 = part; ks0 < BMP / 4; ks0 += WS * RING) {
#pragma unroll
            for (int r = 0; r < RING; ++r) {
                const int ks = ks0 + r * WS;
                const int p = ks * 4 + lk;                             // this lane's pixel of the k-step
                const int py = p >> w.twshift, pxx = p & (w.TW - 1);
                float af[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    af[i] = afr[r][i];
                    bsum[i] += af[i];
                }
                if (ks + WS * RING < BMP / 4) load_g(ks + WS * RING, afr[r]);
                const float* xp = patch + ((py * a.sf) * a.cols + pxx * a.st) * PIX + li;
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp) {
                    const float bf = xp[((tp / KW) * a.cols + (tp % KW)) * PIX];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[i][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf, acc[i][tp], 0, 0, 0);
                }
            }
        }
    }

    if (WS > 1) {                      // combine the pixel shares: waves with part > 0 hand over through LDS
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
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
            const f32x4 v = acc[i][tp];
            // partner lane holds the (ci, im) column of the same rows
            const float o0 = __shfl_xor(v[0], 1, 64), o1 = __shfl_xor(v[1], 1, 64);
            const float o2 = __shfl_xor(v[2], 1, 64), o3 = __shfl_xor(v[3], 1, 64);
            if ((li & 1) == 0) {
                const int co_a = co0 + i * 8 + lk * 2, co_b = co_a + 1;
                // rows r=0,1 -> (co_a, re), (co_a, im); r=2,3 -> (co_b, re), (co_b, im)
                if (co_a < a.Cout) slab[((long)tp * Cin + ci) * a.Cout + co_a] = make_float2(v[0] + o1, v[1] - o0);
                if (co_b < a.Cout) slab[((long)tp * Cin + ci) * a.Cout + co_b] = make_float2(v[2] + o3, v[3] - o2);
            }
        }
    }
    if (ci0 == 0) {                                                    // bias: column sums of gY
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float s = bsum[i];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lk == 0 && colok[i]) w.slab_b[((long)blockIdx.x * w.ncls + cls) * N1 + gcol[i]] = s;
        }
    }
}

template <int KH_, int KW_, int MT_, int WS_> struct Variant { static constexpr int KH = KH_, KW = KW_, MT = MT_, WS = WS_; };

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
            if (co >= 64) return f(Variant<5, 5, 2, 1>{});
            if (co >= 32) return f(Variant<5, 5, 1, 1>{});
            if (co >= 16) return f(Variant<5, 5, 1, 2>{});
            return f(Variant<5, 5, 1, 4>{});
        case 77:
            if (co >= 32) return f(Variant<7, 7, 1, 1>{});
            if (co >= 16) return f(Variant<7, 7, 1, 2>{});
            return f(Variant<7, 7, 1, 4>{});
        default: return DCS_ERR_BADARG;
    }
}

template <class V>
size_t lds_bytes(int rows, int cols) {
    size_t lds = (size_t)rows * cols * PIX * sizeof(float);
    if (V::MT == 1 && V::WS >= 2) lds += (size_t)BMP * ((4 / V::WS) * 16 + (V::WS == 4 ? 0 : 16)) * sizeof(float);   // g_Y tile
    const size_t red = (size_t)(V::WS - 1) * (4 / V::WS) * (V::MT * V::KH * V::KW * 4 + V::MT) * 64 * sizeof(float);
    return red > lds ? red : lds;
}

// workgroups of this variant one CU holds at once (registers / LDS), queried once per variant
template <class V>
int resident_per_cu(size_t lds) {
    static int cached = 0;
    if (cached == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cconv_wgrad_mfma_kernel<V::KH, V::KW, V::MT, V::WS>, 256, lds) !=
                hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 1;
        }
        cached = n > 4 ? 4 : n;
    }
    return cached;
}

template <class V>
int launch(WArgs& w, int Cin, hipStream_t stream) {
    const conv::Args& a = w.c;
    const size_t lds = lds_bytes<V>(a.rows, a.cols);
    if (lds > 150 * 1024) return DCS_ERR_BADARG;
    auto fn = cconv_wgrad_mfma_kernel<V::KH, V::KW, V::MT, V::WS>;
    if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
    const int co_per_block = (4 / V::WS) * V::MT * 8;
    w.co_blocks = (a.Cout + co_per_block - 1) / co_per_block;
    dim3 grid(w.n_slabs, (Cin / CHUNK) * w.co_blocks, w.ncls);
    if (grid.y > 65535) return DCS_ERR_BADARG;
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

// slabs per class for a class-space geometry `c` (tiling filled in), ncls classes
int slabs_for(const conv::Args& c, int ncls, int TH, int TW) {
    const long tiles = (long)((c.Wout + TW - 1) / TW) * ((c.Hout + TH - 1) / TH) * c.B;
    const long wsz = (long)c.kh * c.kw * (c.C1 + c.C2) * c.Cout;
    long cap = (96L << 20) / (wsz * ncls * (long)sizeof(float2));
    if (cap < 1) cap = 1;
    if (cap > 1024) cap = 1024;
    // A workgroup's epilogue writes its whole accumulator set (up to 147 KB), so give each one several pixel
    // tiles rather than one — but keep every CU holding as many workgroups as the variant's registers allow
    // (they overlap each other's gathers): target = 256 CUs x resident workgroups per CU.
    const int rows = (TH - 1) * c.sf + c.kh, cols = (TW - 1) * c.st + c.kw;
    int per_cu = 1, cpb = 8;
    dispatch(c.kh, c.kw, c.Cout, [&](auto v) {
        using V = decltype(v);
        per_cu = resident_per_cu<V>(lds_bytes<V>(rows, cols));
        cpb = (4 / V::WS) * V::MT * 8;
        return 0;
    });
    const long grid_y = (long)((c.C1 + c.C2) / CHUNK) * ((c.Cout + cpb - 1) / cpb) * ncls;
    long want = (256L * per_cu + grid_y - 1) / grid_y;
    if (want < 1) want = 1;
    if (want < cap) cap = want;
    return (int)(tiles < cap ? tiles : cap);
}

void tile_shape(int Hc, int* TH, int* TW) {
    if (Hc >= 8) { *TH = 8; *TW = 16; }
    else if (Hc >= 4) { *TH = 4; *TW = 32; }
    else { *TH = 2; *TW = 64; }
}

// c: class-space geometry; f: classes; Hy x Wy: full g_Y extent
int launch_classes(const conv::Args& c, const Fold& f, int Hy, int Wy, const float* gy, float2* slab_w, float* slab_b,
                   int n_slabs, int TH, int TW, hipStream_t stream) {
    WArgs w;
    w.c = c;
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

bool dcs_conv_wgrad_mfma_ok(int Cin, int Cout, int kh, int kw, int C1) {
    return (Cin % 8) == 0 && (Cout % 8) == 0 && kh == kw && (kh == 1 || kh == 3 || kh == 5 || kh == 7) && !(C1 & 1);
}

// number of partial slabs and the tile shape for a forward geometry (a.Hout / a.Wout set)
int dcs_conv_wgrad_mfma_slabs(const conv::Args& a, int* TH, int* TW) {
    tile_shape(a.Hout, TH, TW);
    return slabs_for(a, 1, *TH, *TW);
}

// slab_w: float2[n_slabs][taps][Cin][Cout]; slab_b: float[n_slabs][2*Cout]
int dcs_conv_wgrad_mfma_launch(conv::Args& a, const float* gy, float2* slab_w, float* slab_b, int n_slabs,
                               hipStream_t stream) {
    int TH, TW;
    tile_shape(a.Hout, &TH, &TW);
    Fold f{};
    f.ncls = 1; f.kh = a.kh; f.kw = a.kw; f.os_f = 1; f.os_t = 1;
    f.pad_f[0] = a.pad_f; f.pad_t[0] = a.pad_t;
    return launch_classes(a, f, a.Hout, a.Wout, gy, slab_w, slab_b, n_slabs, TH, TW, stream);
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
    tile_shape(a.Hin, &th, &tw);
    const long folded = (long)((a.Win + tw - 1) / tw) * ((a.Hin + th - 1) / th) * f.ncls * f.kh * f.kw;
    tile_shape(a.Hout, &th, &tw);
    const long plain = (long)((a.Wout + tw - 1) / tw) * ((a.Hout + th - 1) / th) * a.kh * a.kw;
    return folded < plain;
}

long dcs_conv_wgrad_fold_workspace_bytes(const conv::Args& a) {
    const Fold f = fold_of(a);
    const conv::Args c = class_args(a, f);
    int TH, TW;
    tile_shape(c.Hout, &TH, &TW);
    const long ns = slabs_for(c, f.ncls, TH, TW);
    return ns * f.ncls * ((long)f.kh * f.kw * (a.C1 + a.C2) * a.Cout + a.Cout) * (long)sizeof(float2);
}

int dcs_conv_wgrad_fold_run(const conv::Args& a, const float* gy, void* workspace, long workspace_bytes, float* gw_r,
                            float* gw_i, float* gb_r, float* gb_i, int transposed, hipStream_t stream) {
    if (!dcs_conv_wgrad_fold_ok(a)) return DCS_ERR_BADARG;
    const Fold f = fold_of(a);
    const conv::Args c = class_args(a, f);
    int TH, TW;
    tile_shape(c.Hout, &TH, &TW);
    const int ns = slabs_for(c, f.ncls, TH, TW);
    const int Cin = a.C1 + a.C2;
    const long wsz_c = (long)f.kh * f.kw * Cin * a.Cout;
    if (workspace_bytes < (long)ns * f.ncls * (wsz_c + a.Cout) * (long)sizeof(float2)) return DCS_ERR_WORKSPACE;
    float2* slab_w = (float2*)workspace;
    float2* slab_b = slab_w + (long)ns * f.ncls * wsz_c;
    const int rc = launch_classes(c, f, a.Hout, a.Wout, gy, slab_w, (float*)slab_b, ns, TH, TW, stream);
    if (rc != DCS_OK) return rc;
    wreduce::Job j{};                                     // class gradients -> 3x3 parameter (wgrad_reduce.hip)
    j.slab_w = slab_w; j.slab_b = slab_b; j.gw_r = gw_r; j.gw_i = gw_i; j.gb_r = gb_r; j.gb_i = gb_i;
    j.n_slabs = ns; j.Cout = a.Cout; j.Cin = Cin; j.kh = 3; j.kw = 3; j.transposed = transposed;
    j.up_f = a.up_f; j.up_t = a.up_t;
    return wreduce::emit(j, stream);
    return DCS_OK;
}
