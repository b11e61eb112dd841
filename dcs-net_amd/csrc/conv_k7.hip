// conv_k7.hip — the 7x7, stride-1, pad-3 complex convolutions with one or two channels on either side: the spatial-
// attention conv (2 -> 1, c_network.py:74-84; 13 per forward pass) and its data gradient (1 -> 2), register-blocked.
//
// The generic direct kernel (conv_direct.hip) does one LDS read of x and one scalar-cache read of w per complex MAC:
// at Cout = 1 that is 2 memory operations per 4 FMAs with nothing to amortise them (measured ~8 TFLOP/s).  Here a thread
// owns PB = 4 horizontally adjacent output pixels: per (input channel, kernel row) it loads the 10 input values the four
// windows span once and re-uses each of them for up to 4 outputs x 7 taps; the 49 x CI x CO weights sit in LDS
// (broadcast reads).  ~6.6 FMAs per LDS access instead of 2.
//   tile  16 rows x 64 columns of output per 256-thread workgroup; haloed input 22 x 70 x CI in LDS
// Problems may be batched (table by value) like the other attention kernels; the grid is compacted: problem k owns
// blockIdx.x in [x0[k], x0[k+1]), so no workgroup is dispatched only to find it has no tile.
#include "conv_common.h"

namespace {

constexpr int K = 7, PAD = 3, TR = 16, TC = 64, PB = 4, ROWS = TR + K - 1, COLS = TC + K - 1, COLSP = COLS + 2;   // 72: rows stay 16-byte aligned
constexpr int kMaxBatch = 8;
typedef float v2f __attribute__((ext_vector_type(2)));

struct K7P {
    const float2* x; const float2* w; const float2* bias; float2* y;
    int H, W, tiles_w, tiles, act;
};
struct K7Table { K7P p[kMaxBatch]; int x0[kMaxBatch + 1]; };

template <int CI, int CO>
__global__ __launch_bounds__(256) void cconv_k7_kernel(K7Table tb) {
    DCS_PRIO_CRITICAL();
    __shared__ __attribute__((aligned(16))) float2 tile[CI][ROWS * COLSP];
    __shared__ __attribute__((aligned(16))) float4 wl[K * K * CI * CO];     // {w.x, w.y, w.y, w.x}: both broadcasts read a LOW half
    int z = 0;
#pragma unroll
    for (int k = 1; k < kMaxBatch; ++k) z += (int)blockIdx.x >= tb.x0[k] ? 1 : 0;
    const K7P& p = tb.p[z];
    const int tl = (int)blockIdx.x - tb.x0[z];
    const int t = threadIdx.x, b = blockIdx.y;
    const int oy0 = (tl / p.tiles_w) * TR, ox0 = (tl % p.tiles_w) * TC;
    static_assert(K * K * CI * CO <= 256, "one weight per thread");
    const float2 wv = p.w[t < K * K * CI * CO ? t : 0];                       // [tap][ci][co]; unconditional: goes out with the tile loads
    const float2* xb = p.x + (long)b * p.H * p.W * CI;
    // The haloed tile in ONE batch of unconditional loads (clamped coordinates, zeroed afterwards).  As a loop of predicated
    // loads every trip was its own memory round trip: a branch, the load, s_waitcnt vmcnt(0), the LDS store — six in a row.
    constexpr int NIT = (ROWS * COLS + 255) / 256;
    float2 tv[NIT][CI];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = t + 256 * k < ROWS * COLS ? t + 256 * k : ROWS * COLS - 1;
        const int y = oy0 - PAD + i / COLS, x = ox0 - PAD + i % COLS;
        const int yc = y < 0 ? 0 : (y >= p.H ? p.H - 1 : y), xc = x < 0 ? 0 : (x >= p.W ? p.W - 1 : x);
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) tv[k][ci] = xb[((long)yc * p.W + xc) * CI + ci];
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = t + 256 * k;
        if (i < ROWS * COLS) {
            const int iy = i / COLS, ix = i % COLS;
            const int y = oy0 - PAD + iy, x = ox0 - PAD + ix;
            const bool in = y >= 0 && y < p.H && x >= 0 && x < p.W;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) tile[ci][iy * COLSP + ix] = in ? tv[k][ci] : make_float2(0.f, 0.f);
        }
    }
    if (t < K * K * CI * CO) wl[t] = make_float4(wv.x, wv.y, wv.y, wv.x);
    __syncthreads();
    const int ty = t / (TC / PB), tx = (t % (TC / PB)) * PB;
    // complex MAC as two packed FMAs (v_pk_fma_f32) on TWO accumulators: P += w.x * (x.re, x.im), Q += w.y * (x.re, x.im);
    // acc = (P.re - Q.im, P.im + Q.re) at the end.  Every packed operand reads its own halves or a broadcast LOW half — the form
    // with the rotated value (-x.im, x.re) had the compiler fold the rotation into the FMA as a cross-half operand selection
    // (op_sel), which gfx950 can get wrong beside bf16-MFMA waves (dcs_common.h); it also frees the ten rotated values' registers.
    v2f accp[PB][CO], accq[PB][CO];
#pragma unroll
    for (int q = 0; q < PB; ++q)
#pragma unroll
        for (int co = 0; co < CO; ++co) { accp[q][co] = v2f{0.f, 0.f}; accq[q][co] = v2f{0.f, 0.f}; }
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
#pragma unroll 1
        for (int dy = 0; dy < K; ++dy) {
            v2f xv[PB + K - 1];                                        // 10 values = 5 aligned 16-byte reads
            const float4* row = reinterpret_cast<const float4*>(&tile[ci][(ty + dy) * COLSP + tx]);
#pragma unroll
            for (int j = 0; j < (PB + K - 1) / 2; ++j) {
                const float4 v4 = row[j];
                xv[2 * j] = v2f{v4.x, v4.y}; xv[2 * j + 1] = v2f{v4.z, v4.w};
            }
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float4 w = wl[((dy * K + dx) * CI + ci) * CO + co];
                    const v2f wx = v2f{w.x, w.x}, wy = v2f{w.z, w.z};      // both broadcasts read the LOW half of a pair
#pragma unroll
                    for (int q = 0; q < PB; ++q) {
                        accp[q][co] = __builtin_elementwise_fma(wx, xv[q + dx], accp[q][co]);
                        accq[q][co] = __builtin_elementwise_fma(wy, xv[q + dx], accq[q][co]);
                    }
                }
            }
        }
    }
    v2f acc[PB][CO];
#pragma unroll
    for (int q = 0; q < PB; ++q)
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[q][co] = v2f{accp[q][co].x - accq[q][co].y, accp[q][co].y + accq[q][co].x};
    const int oy = oy0 + ty;
    if (oy >= p.H) return;
    float2* yb = p.y + ((long)b * p.H + oy) * p.W * CO;
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int ox = ox0 + tx + q;
        if (ox >= p.W) continue;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float2 bv = p.bias ? p.bias[co] : make_float2(0.f, 0.f);
            yb[(long)ox * CO + co] = make_float2(dcs_act(acc[q][co].x + bv.x, p.act), dcs_act(acc[q][co].y + bv.y, p.act));
        }
    }
}

bool k7_geom(const conv::Args& a) {
    return a.kh == K && a.kw == K && a.sf == 1 && a.st == 1 && a.pad_f == PAD && a.pad_t == PAD && a.up_f == 1 &&
           a.up_t == 1 && a.C2 == 0 && a.x1 && a.wp && a.y && !a.coef &&          // (zero insertion with up = 1 inserts nothing)
           ((a.C1 == 2 && a.Cout == 1) || (a.C1 == 1 && a.Cout == 2) || (a.C1 == 1 && a.Cout == 1));
}

}  // namespace

bool dcs_conv_k7_ok(const conv::Args* a, int n) {
    if (n < 1 || n > kMaxBatch) return false;
    for (int i = 0; i < n; ++i)
        if (!k7_geom(a[i]) || a[i].C1 != a[0].C1 || a[i].Cout != a[0].Cout || a[i].B != a[0].B) return false;
    return a[0].B <= 65535;
}

int dcs_conv_k7_launch(const conv::Args* a, int n, hipStream_t stream) {
    if (!dcs_conv_k7_ok(a, n)) return DCS_ERR_BADARG;
    K7Table tb;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        K7P& p = tb.p[i];
        p.x = a[i].x1; p.w = a[i].wp; p.bias = a[i].bias; p.y = a[i].y;
        p.H = a[i].Hin; p.W = a[i].Win; p.act = a[i].act;
        p.tiles_w = (p.W + TC - 1) / TC;
        p.tiles = p.tiles_w * ((p.H + TR - 1) / TR);
        tb.x0[i] = tiles;
        tiles += p.tiles;
    }
    for (int i = n; i <= kMaxBatch; ++i) tb.x0[i] = i == n ? tiles : 0x7fffffff;
    dim3 grid(tiles, a[0].B, 1);
    if (a[0].C1 == 2) DCS_LAUNCH((cconv_k7_kernel<2, 1>), grid, dim3(256), 0, stream, tb);
    else if (a[0].Cout == 2) DCS_LAUNCH((cconv_k7_kernel<1, 2>), grid, dim3(256), 0, stream, tb);
    else DCS_LAUNCH((cconv_k7_kernel<1, 1>), grid, dim3(256), 0, stream, tb);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
