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
};

// MT row tiles (8 output channels each) per wave; WS waves split the tile's PIXELS (k-steps) and
// 4/WS waves split the output channels, so layers with few output channels still fill all four
// SIMDs: the pixel partials are combined through LDS once, after the last tile.
// Two workgroups per CU (<= 256 VGPR + AGPR per lane) whenever the accumulator set allows it: one gathers while
// the other runs its MFMAs.
template <int KS, int MT, int WS>
__global__ __launch_bounds__(256, (MT * KS * KS * 4 <= 160 && !(KS == 5 && WS == 4) ? 2 : 1)) void cconv_wgrad_mfma_kernel(WArgs w) {
    constexpr int TAPS = KS * KS;
    constexpr int WCO = 4 / WS;
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
        const int vy0 = oy0 * a.sf - a.pad_f, vx0 = ox0 * a.st - a.pad_t;
        __syncthreads();
        for (int idx = t; idx < npix * 4; idx += 256) {
            const int q = idx & 3, px = idx >> 2;
            const int ix = px % a.cols, iy = px / a.cols;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            long sp;
            if (conv::src_pixel(a, b, vy0 + iy, vx0 + ix, &sp)) {
                const int c = ci0 + 2 * q;
                const float2* src = (c < a.C1) ? a.x1 + sp * a.C1 + c : a.x2 + sp * a.C2 + (c - a.C1);
                v = *reinterpret_cast<const float4*>(src);
            }
            *reinterpret_cast<float4*>(patch + px * PIX + q * 4) = v;
        }
        __syncthreads();
        // gY fragments come from L2 with ~1-2 us latency and a k-step is only MT*TAPS*32 cycles of MFMA: keep the
        // next RING k-steps' loads in flight (static register ring; the k-step loop is unrolled over it)
        constexpr int RING = MT >= 4 ? 2 : 4;                          // BMP/4/WS is 32, 16 or 8
        float afr[RING][MT];
        auto load_g = [&](int ks, float* dst) {
            const int p = ks * 4 + lk;
            const int oy = oy0 + (p >> w.twshift), ox = ox0 + (p & (w.TW - 1));
            const bool inb = oy < a.Hout && ox < a.Wout;
            const float* gp = w.gy + (((long)b * a.Hout + (inb ? oy : 0)) * a.Wout + (inb ? ox : 0)) * N1;
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
                    const float bf = xp[((tp / KS) * a.cols + (tp % KS)) * PIX];
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
    float2* slab = w.slab_w + (long)blockIdx.x * wsz;
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
            if (lk == 0 && colok[i]) w.slab_b[(long)blockIdx.x * N1 + gcol[i]] = s;
        }
    }
}

template <int KS_, int MT_, int WS_> struct Variant { static constexpr int KS = KS_, MT = MT_, WS = WS_; };

// (MT, WS) so that the 4 waves cover min(Cout, most-per-kernel-size) output channels without idle lanes
template <class F>
int dispatch(int k, int co, F&& f) {
    switch (k) {
        case 1:
            if (co >= 128) return f(Variant<1, 4, 1>{});
            if (co >= 64) return f(Variant<1, 2, 1>{});
            if (co >= 32) return f(Variant<1, 1, 1>{});
            if (co >= 16) return f(Variant<1, 1, 2>{});
            return f(Variant<1, 1, 4>{});
        case 3:
            if (co >= 128) return f(Variant<3, 4, 1>{});
            if (co >= 64) return f(Variant<3, 2, 1>{});
            if (co >= 32) return f(Variant<3, 1, 1>{});
            if (co >= 16) return f(Variant<3, 1, 2>{});
            return f(Variant<3, 1, 4>{});
        case 5:
            if (co >= 64) return f(Variant<5, 2, 1>{});
            if (co >= 32) return f(Variant<5, 1, 1>{});
            if (co >= 16) return f(Variant<5, 1, 2>{});
            return f(Variant<5, 1, 4>{});
        case 7:
            if (co >= 32) return f(Variant<7, 1, 1>{});
            if (co >= 16) return f(Variant<7, 1, 2>{});
            return f(Variant<7, 1, 4>{});
        default: return DCS_ERR_BADARG;
    }
}

template <class V>
size_t lds_bytes(int rows, int cols) {
    size_t lds = (size_t)rows * cols * PIX * sizeof(float);
    const size_t red = (size_t)(V::WS - 1) * (4 / V::WS) * (V::MT * V::KS * V::KS * 4 + V::MT) * 64 * sizeof(float);
    return red > lds ? red : lds;
}

// workgroups of this variant one CU holds at once (registers / LDS), queried once per variant
template <class V>
int resident_per_cu(size_t lds) {
    static int cached = 0;
    if (cached == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cconv_wgrad_mfma_kernel<V::KS, V::MT, V::WS>, 256, lds) !=
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
    auto fn = cconv_wgrad_mfma_kernel<V::KS, V::MT, V::WS>;
    if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
    const int co_per_block = (4 / V::WS) * V::MT * 8;
    w.co_blocks = (a.Cout + co_per_block - 1) / co_per_block;
    dim3 grid(w.n_slabs, (Cin / CHUNK) * w.co_blocks);
    if (grid.y > 65535) return DCS_ERR_BADARG;
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, stream, w);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

}  // namespace

bool dcs_conv_wgrad_mfma_ok(int Cin, int Cout, int kh, int kw, int C1) {
    return (Cin % 8) == 0 && (Cout % 8) == 0 && kh == kw && (kh == 1 || kh == 3 || kh == 5 || kh == 7) && !(C1 & 1);
}

// number of partial slabs and the tile shape for a forward geometry (a.Hout / a.Wout set)
int dcs_conv_wgrad_mfma_slabs(const conv::Args& a, int* TH, int* TW) {
    if (a.Hout >= 8) { *TH = 8; *TW = 16; }
    else if (a.Hout >= 4) { *TH = 4; *TW = 32; }
    else { *TH = 2; *TW = 64; }
    const long tiles = (long)((a.Wout + *TW - 1) / *TW) * ((a.Hout + *TH - 1) / *TH) * a.B;
    const long wsz = (long)a.kh * a.kw * (a.C1 + a.C2) * a.Cout;
    long cap = (96L << 20) / (wsz * (long)sizeof(float2));
    if (cap < 1) cap = 1;
    if (cap > 1024) cap = 1024;
    // A workgroup's epilogue writes its whole accumulator set (up to 147 KB), so give each one several pixel
    // tiles rather than one — but keep every CU holding as many workgroups as the variant's registers allow
    // (they overlap each other's gathers): target = 256 CUs x resident workgroups per CU.
    const int rows = (*TH - 1) * a.sf + a.kh, cols = (*TW - 1) * a.st + a.kw;
    int per_cu = 1, cpb = 8;
    dispatch(a.kh, a.Cout, [&](auto v) {
        using V = decltype(v);
        per_cu = resident_per_cu<V>(lds_bytes<V>(rows, cols));
        cpb = (4 / V::WS) * V::MT * 8;
        return 0;
    });
    static const long scale = [] { const char* e = getenv("DCS_WGRAD_WGS_PER_SLOT"); return e ? atol(e) : 1L; }();
    const long grid_y = (long)((a.C1 + a.C2) / CHUNK) * ((a.Cout + cpb - 1) / cpb);
    long want = (256L * per_cu * scale + grid_y - 1) / grid_y;
    if (want < 1) want = 1;
    if (want < cap) cap = want;
    return (int)(tiles < cap ? tiles : cap);
}

// slab_w: float2[n_slabs][taps][Cin][Cout]; slab_b: float[n_slabs][2*Cout]
int dcs_conv_wgrad_mfma_launch(conv::Args& a, const float* gy, float2* slab_w, float* slab_b, int n_slabs,
                               hipStream_t stream) {
    WArgs w;
    w.c = a;
    w.gy = gy; w.slab_w = slab_w; w.slab_b = slab_b; w.n_slabs = n_slabs;
    int TH, TW;
    dcs_conv_wgrad_mfma_slabs(a, &TH, &TW);
    w.TH = TH; w.TW = TW;
    w.twshift = TW == 16 ? 4 : (TW == 32 ? 5 : 6);
    w.c.tiles_w = (a.Wout + TW - 1) / TW;
    w.c.tiles_h = (a.Hout + TH - 1) / TH;
    w.c.rows = (TH - 1) * a.sf + a.kh;
    w.c.cols = (TW - 1) * a.st + a.kw;
    w.total_tiles = w.c.tiles_w * w.c.tiles_h * a.B;
    const int Cin = a.C1 + a.C2;
    return dispatch(a.kh, a.Cout, [&](auto v) { return launch<decltype(v)>(w, Cin, stream); });
}
