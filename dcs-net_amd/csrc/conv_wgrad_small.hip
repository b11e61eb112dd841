// conv_wgrad_small.hip — weight gradient of the two 7x7 small-channel convolutions of the network, pixel-stationary.
//
//   MODE 0  spatial-attention conv: 2 -> 1 channels, stride 1 (c_network.py:74-84): 98 complex outputs
//   MODE 1  first encoder conv:     1 -> 8 channels, stride 2 (c_network.py:107-112, config.py:83-91): 392 outputs
//
// g_W[tap][ci][co] = sum_p g_Y[p][co] conj(X[p*s - pad + tap][ci]) has far fewer outputs than pixels, so the
// generic output-stationary kernel (conv_direct.hip) re-reads both operands from LDS for every FMA quad.  Here a
// thread owns PIXELS and keeps all 49 taps x 2 of the outputs in registers (196 VGPRs): g_Y comes from HBM once
// per pixel, the haloed input tile is staged in LDS once per 16x16 tile, and each LDS read feeds 4-8 FMAs.
//   MODE 0: the 256 threads take the 256 pixels of a tile; acc[tap][ci].  (The one-pixel-per-thread form of the template
//           below is no longer instantiated: the 2 -> 1 problems run wgrad_sa_body — four pixels per thread on 16 x 64
//           tiles — further down; the template's MODE 0 branches document what it replaced.)
//   MODE 1: wave w owns output channels (2w, 2w+1) and walks all 256 pixels of the tile in 4 passes; acc[tap][co&1].
// The lanes are summed once per workgroup (DPP row shifts / broadcasts; cross-wave through LDS for MODE 0) into
// one partial slab per workgroup — same slab layout and reduce kernel as the other weight-gradient paths, no atomics.
#include "conv_common.h"
#include "wgrad_reduce.h"

#include <mutex>
#include <vector>

namespace {

constexpr int TH = 16, TW = 16, KS = 7, TAPS = KS * KS;
typedef float v2f __attribute__((ext_vector_type(2)));

struct SArgs {
    conv::Args c;
    const float2* gy; float2* slab_w; float2* slab_b;
    int n_slabs, total_tiles;
};

template <int MODE, int S>
struct SmallLds {
    static constexpr int NC = MODE == 0 ? 2 : 1;
    static constexpr int ROWS = (TH - 1) * S + KS, COLS = (TW - 1) * S + KS, COLSP = COLS | 1, PLANE = ROWS * COLSP + 1;
    float2 tile[NC * PLANE];
    float2 red[4][2 * TAPS + 2];
};

// `bid`: this workgroup's slab index inside problem `w` (blockIdx.x of a single launch)
template <int MODE, int S>
__device__ __forceinline__ void wgrad_small_body(const SArgs& w, int bid, SmallLds<MODE, S>& lds) {
    constexpr int NC = SmallLds<MODE, S>::NC;
    constexpr int ROWS = SmallLds<MODE, S>::ROWS, COLS = SmallLds<MODE, S>::COLS, COLSP = SmallLds<MODE, S>::COLSP,
                  PLANE = SmallLds<MODE, S>::PLANE;
    float2* tile = lds.tile;
    float2 (*red)[2 * TAPS + 2] = lds.red;
    const conv::Args& a = w.c;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tiles_per_img = a.tiles_w * a.tiles_h;

    // (re, im) accumulators; g conj(x) = x.re (g.re, g.im) + x.im (g.im, -g.re): two packed FMAs (v_pk_fma_f32) per MAC
    v2f acc[TAPS][2];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) { acc[tp][0] = v2f{0.f, 0.f}; acc[tp][1] = v2f{0.f, 0.f}; }
    float2 b0 = make_float2(0.f, 0.f), b1 = make_float2(0.f, 0.f);

    for (int tl = bid; tl < w.total_tiles; tl += w.n_slabs) {
        const int b = tl / tiles_per_img, tile_id = tl % tiles_per_img;
        const int oy0 = (tile_id / a.tiles_w) * TH, ox0 = (tile_id % a.tiles_w) * TW;
        const int vy0 = oy0 * S - a.pad_f, vx0 = ox0 * S - a.pad_t;
        __syncthreads();
        for (int idx = t; idx < ROWS * COLS * NC; idx += 256) {
            const int ci = idx % NC, px = idx / NC;
            const int ix = px % COLS, iy = px / COLS;
            tile[ci * PLANE + iy * COLSP + ix] = conv::gather(a, b, vy0 + iy, vx0 + ix, ci);
        }
        __syncthreads();
#pragma unroll 1
        for (int sub = 0; sub < (MODE == 0 ? 1 : 4); ++sub) {
            const int p = MODE == 0 ? t : sub * 64 + lane;
            const int py = p / TW, pxx = p % TW;
            const int oy = oy0 + py, ox = ox0 + pxx;
            float2 g0 = make_float2(0.f, 0.f), g1 = make_float2(0.f, 0.f);
            if (oy < a.Hout && ox < a.Wout) {
                const long o = ((long)b * a.Hout + oy) * a.Wout + ox;
                if (MODE == 0) {
                    g0 = w.gy[o];
                } else {
                    const float4 v = *reinterpret_cast<const float4*>(w.gy + o * 8 + 2 * wave);
                    g0 = make_float2(v.x, v.y); g1 = make_float2(v.z, v.w);
                }
            }
            b0.x += g0.x; b0.y += g0.y; b1.x += g1.x; b1.y += g1.y;
            const float2* base = tile + (py * S) * COLSP + pxx * S;
            const v2f g0v = v2f{g0.x, g0.y}, g0r = v2f{g0.y, -g0.x}, g1v = v2f{g1.x, g1.y}, g1r = v2f{g1.y, -g1.x};
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    const int tp = dy * KS + dx;
                    const float2 x0 = base[dy * COLSP + dx];
                    const v2f xx = v2f{x0.x, x0.x}, xy = dcs_bcast2(x0.y);        // (high half broadcast by moves, not by op_sel)
                    acc[tp][0] = __builtin_elementwise_fma(xx, g0v, __builtin_elementwise_fma(xy, g0r, acc[tp][0]));
                    if (MODE == 0) {
                        const float2 x1 = base[PLANE + dy * COLSP + dx];
                        acc[tp][1] = __builtin_elementwise_fma(v2f{x1.x, x1.x}, g0v,
                                                               __builtin_elementwise_fma(dcs_bcast2(x1.y), g0r, acc[tp][1]));
                    } else {
                        acc[tp][1] = __builtin_elementwise_fma(xx, g1v, __builtin_elementwise_fma(xy, g1r, acc[tp][1]));
                    }
                }
        }
    }

    // lanes -> one value per output.  Output j = tap*2 + k: MODE 0 (tap, ci = k); MODE 1 (tap, co = 2*wave + k)
    const long wsz = (long)TAPS * (MODE == 0 ? 2 : 8);
    float2* slab = w.slab_w + (long)bid * wsz;
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float sr = dcs_wave_sum_lane63(acc[tp][k].x), si = dcs_wave_sum_lane63(acc[tp][k].y);
            if (lane == 63) {
                if (MODE == 0) red[wave][tp * 2 + k] = make_float2(sr, si);
                else slab[tp * 8 + 2 * wave + k] = make_float2(sr, si);
            }
        }
    const float2 s0 = make_float2(dcs_wave_sum_lane63(b0.x), dcs_wave_sum_lane63(b0.y));
    const float2 s1 = make_float2(dcs_wave_sum_lane63(b1.x), dcs_wave_sum_lane63(b1.y));
    if (MODE == 1) {
        if (lane == 63) {
            w.slab_b[(long)bid * 8 + 2 * wave] = s0;
            w.slab_b[(long)bid * 8 + 2 * wave + 1] = s1;
        }
        return;
    }
    if (lane == 63) red[wave][2 * TAPS] = s0;
    __syncthreads();
    if (t <= 2 * TAPS) {
        float2 v = red[0][t];
#pragma unroll
        for (int q = 1; q < 4; ++q) { v.x += red[q][t].x; v.y += red[q][t].y; }
        if (t < 2 * TAPS) slab[t] = v;                     // [tap][ci][co = 0]
        else w.slab_b[bid] = v;
    }
}

// ---- MODE 0 re-blocked: a thread owns PBS = 4 horizontally adjacent pixels of a 16 x 64 tile ------------------------------
// One pixel per thread reads one LDS value per complex MAC: 98 ds_read_b64 per pixel, and the 13 attention problems of a
// step are LDS-bandwidth-bound (~65 us of LDS cycles for 16 us of FMAs).  With four pixels per thread the 10 input values
// a kernel row's four windows span are read once (5 x ds_read_b128) for 28 MACs.  Same accumulators (all 49 taps x 2
// channels per thread), same slab layout and cross-lane reduction as the one-pixel form.
constexpr int STH = 16, STW = 64, PBS = 4, SROWS = STH + KS - 1, SCOLS = STW + KS - 1, SCOLSP = SCOLS + 2;
struct SaLds {
    __attribute__((aligned(16))) float2 tile[2][SROWS * SCOLSP];
    float2 red[4][2 * TAPS + 2];
};

__device__ __forceinline__ void wgrad_sa_body(const SArgs& w, int bid, SaLds& lds) {
    const conv::Args& a = w.c;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ty = t / (STW / PBS), tx = (t % (STW / PBS)) * PBS;
    const int H = a.Hout, W = a.Wout;                                   // stride 1, pad 3: input extent = output extent
    const int tiles_w = (W + STW - 1) / STW, tiles_per_img = tiles_w * ((H + STH - 1) / STH);

    float ar[TAPS][2], ai[TAPS][2];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) { ar[tp][0] = ar[tp][1] = ai[tp][0] = ai[tp][1] = 0.f; }
    float2 b0 = make_float2(0.f, 0.f);

    for (int tl = bid; tl < w.total_tiles; tl += w.n_slabs) {
        const int b = tl / tiles_per_img, tile_id = tl % tiles_per_img;
        const int oy0 = (tile_id / tiles_w) * STH, ox0 = (tile_id % tiles_w) * STW;
        const float4* xb = reinterpret_cast<const float4*>(a.x1) + (long)b * H * W;     // 2 complex channels per pixel
        __syncthreads();
        for (int i = t; i < SROWS * SCOLS; i += 256) {
            const int iy = i / SCOLS, ix = i % SCOLS;
            const int y = oy0 - a.pad_f + iy, x = ox0 - a.pad_t + ix;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < H && x >= 0 && x < W) v = xb[(long)y * W + x];
            lds.tile[0][iy * SCOLSP + ix] = make_float2(v.x, v.y);
            lds.tile[1][iy * SCOLSP + ix] = make_float2(v.z, v.w);
        }
        __syncthreads();
        float2 g[PBS];
        const int oy = oy0 + ty;
#pragma unroll
        for (int q = 0; q < PBS; ++q) {
            const int ox = ox0 + tx + q;
            g[q] = (oy < H && ox < W) ? w.gy[((long)b * H + oy) * W + ox] : make_float2(0.f, 0.f);
            b0.x += g[q].x; b0.y += g[q].y;
        }
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int dy = 0; dy < KS; ++dy) {
                float2 xv[PBS + KS - 1];
                const float4* row = reinterpret_cast<const float4*>(&lds.tile[ci][(ty + dy) * SCOLSP + tx]);
#pragma unroll
                for (int j = 0; j < (PBS + KS - 1) / 2; ++j) {
                    const float4 v4 = row[j];
                    xv[2 * j] = make_float2(v4.x, v4.y); xv[2 * j + 1] = make_float2(v4.z, v4.w);
                }
#pragma unroll
                for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                    for (int q = 0; q < PBS; ++q) {                     // g * conj(x)
                        const float2 x0 = xv[q + dx];
                        ar[dy * KS + dx][ci] = fmaf(g[q].x, x0.x, fmaf(g[q].y, x0.y, ar[dy * KS + dx][ci]));
                        ai[dy * KS + dx][ci] = fmaf(g[q].y, x0.x, fmaf(-g[q].x, x0.y, ai[dy * KS + dx][ci]));
                    }
            }
    }

    // lanes -> one value per output j = tap*2 + ci (same tail as the one-pixel form)
    float2 (*red)[2 * TAPS + 2] = lds.red;
    float2* slab = w.slab_w + (long)bid * (TAPS * 2);
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float sr = dcs_wave_sum_lane63(ar[tp][k]), si = dcs_wave_sum_lane63(ai[tp][k]);
            if (lane == 63) red[wave][tp * 2 + k] = make_float2(sr, si);
        }
    const float2 s0 = make_float2(dcs_wave_sum_lane63(b0.x), dcs_wave_sum_lane63(b0.y));
    if (lane == 63) red[wave][2 * TAPS] = s0;
    __syncthreads();
    if (t <= 2 * TAPS) {
        float2 v = red[0][t];
#pragma unroll
        for (int q = 1; q < 4; ++q) { v.x += red[q][t].x; v.y += red[q][t].y; }
        if (t < 2 * TAPS) slab[t] = v;                     // [tap][ci][co = 0]
        else w.slab_b[bid] = v;
    }
}

__global__ __launch_bounds__(256, 2) void cconv_wgrad_sa_kernel(SArgs w) {
    __shared__ SaLds lds;
    wgrad_sa_body(w, blockIdx.x, lds);
}

template <int MODE, int S>
__global__ __launch_bounds__(256, 2) void cconv_wgrad_small_kernel(SArgs w) {
    __shared__ SmallLds<MODE, S> lds;
    wgrad_small_body<MODE, S>(w, blockIdx.x, lds);
}

// several independent 2 -> 1 problems (the 13 spatial-attention convs of a train step) in ONE launch: inside a
// deferred-reduce scope (wgrad_reduce.h) nothing needs these gradients before the optimizer, so their launches are
// recorded and the flush runs them together — 13 kernels of ~10-20 us, each mostly a fixed DPP-reduction tail, overlap
constexpr int kSmallBatch = 16;
struct SmallTable { int n; int blk0[kSmallBatch + 1]; SArgs p[kSmallBatch]; };

__global__ __launch_bounds__(256, 2) void cconv_wgrad_small_multi_kernel(SmallTable t) {
    __shared__ SaLds lds;
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.blk0[k + 1]) ++k;
    wgrad_sa_body(t.p[k], blockIdx.x - t.blk0[k], lds);
}

std::vector<SArgs>* g_small_deferred = nullptr;         // guarded by g_small_mutex: autograd records on its own thread
std::mutex g_small_mutex;

}  // namespace

// forward geometry `a` (fwd_args + conv_geometry of conv_direct.hip: 16x16 tiles)
bool dcs_conv_wgrad_small_ok(const conv::Args& a) {
    if (a.kh != KS || a.kw != KS || a.up_f != 1 || a.up_t != 1 || a.C2 != 0 || a.sf != a.st) return false;
    if (a.C1 == 2 && a.Cout == 1 && a.sf == 1) return true;
    return a.C1 == 1 && a.Cout == 8 && a.sf == 2;
}

// slab_w: float2[n_slabs][49][Cin][Cout]; slab_b: float2[n_slabs][Cout]
int dcs_conv_wgrad_small_launch(const conv::Args& a, const float* gy, float2* slab_w, float2* slab_b, int n_slabs,
                                hipStream_t stream) {
    if (!dcs_conv_wgrad_small_ok(a) || n_slabs < 1) return DCS_ERR_BADARG;
    SArgs w;
    w.c = a;
    w.gy = (const float2*)gy; w.slab_w = slab_w; w.slab_b = slab_b;
    w.n_slabs = n_slabs;
    w.total_tiles = a.tiles_w * a.tiles_h * a.B;
    if (a.C1 == 2)                                           // 16 x 64 tiles (wgrad_sa_body); idle workgroups write zero slabs
        w.total_tiles = ((a.Wout + STW - 1) / STW) * ((a.Hout + STH - 1) / STH) * a.B;
    if (a.C1 == 2 && wreduce::deferring()) {                 // recorded; dcs_conv_wgrad_small_flush launches the batch
        std::lock_guard<std::mutex> lock(g_small_mutex);
        if (!g_small_deferred) g_small_deferred = new std::vector<SArgs>();
        g_small_deferred->push_back(w);
        return DCS_OK;
    }
    if (a.C1 == 2) DCS_LAUNCH(cconv_wgrad_sa_kernel, dim3(n_slabs), dim3(256), 0, stream, w);
    else DCS_LAUNCH((cconv_wgrad_small_kernel<1, 2>), dim3(n_slabs), dim3(256), 0, stream, w);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// launch every recorded 2 -> 1 problem (called by dcs_wgrad_defer_flush before the batched reduces)
int dcs_conv_wgrad_small_flush(hipStream_t stream) {
    std::vector<SArgs>* jobs;
    {
        std::lock_guard<std::mutex> lock(g_small_mutex);
        jobs = g_small_deferred;
        g_small_deferred = nullptr;
    }
    if (!jobs) return DCS_OK;
    int rc = DCS_OK;
    for (size_t i0 = 0; i0 < jobs->size() && rc == DCS_OK; i0 += kSmallBatch) {
        SmallTable t;
        t.n = 0;
        int nb = 0;
        for (size_t i = i0; i < jobs->size() && t.n < kSmallBatch; ++i) {
            t.blk0[t.n] = nb;
            t.p[t.n++] = (*jobs)[i];
            nb += (*jobs)[i].n_slabs;
        }
        t.blk0[t.n] = nb;
        DCS_LAUNCH(cconv_wgrad_small_multi_kernel, dim3(nb), dim3(256), 0, stream, t);
        if (hipGetLastError() != hipSuccess) rc = DCS_ERR_LAUNCH;
    }
    delete jobs;
    return rc;
}
