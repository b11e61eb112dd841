// elementwise.hip — stand-alone complex activation and nearest upsample, channels-last.
//
// Only the layer-by-layer drop-in surface (dcsnet/complexLayers.py, complexFunctions.py) uses
// these: in C_NETWORK.forward the activation is fused into dcs_cbn_fwd and the upsample into the
// input gather of dcs_cconv2d_fwd, so neither tensor pass exists there.
#include "dcs_common.h"

namespace {
constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void act_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                        int act) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads)
        y[i] = dcs_act(x[i], act);
}

// y[b][oy][ox][c] = x[b][oy/uf][ox/ut][c]; one float2 (complex) per thread-iteration
__global__ __launch_bounds__(kThreads) void upsample_kernel(const float2* __restrict__ x, float2* __restrict__ y,
                                                             int B, int H, int W, int C, int uf, int ut) {
    const long n = (long)B * H * uf * W * ut * C;
    const int Wo = W * ut, Ho = H * uf;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const int c = (int)(i % C);
        long p = i / C;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        y[i] = x[(((long)b * H + oy / uf) * W + ox / ut) * C + c];
    }
}

inline int ew_grid(long n) {
    long nb = (n + kThreads * 4 - 1) / (kThreads * 4);
    return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
}
}  // namespace

extern "C" int dcs_complex_act_fwd(const float* x, float* y, long n_floats, int act, dcs_stream_t stream) {
    if (!x || !y || n_floats <= 0 || act < DCS_ACT_NONE || act > DCS_ACT_SIGMOID) return DCS_ERR_BADARG;
    DCS_LAUNCH(act_kernel, dim3(ew_grid(n_floats)), dim3(kThreads), 0, dcs_stream(stream), x, y, n_floats, act);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_complex_upsample_fwd(const float* x, float* y, int B, int H, int W, int C, int up_f, int up_t,
                                        dcs_stream_t stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || up_f < 1 || up_t < 1) return DCS_ERR_BADARG;
    const long n = (long)B * H * up_f * W * up_t * C;
    DCS_LAUNCH(upsample_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream), (const float2*)x,
                       (float2*)y, B, H, W, C, up_f, up_t);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// ---- tap-sum: the spatial half of a Cout = 1 convolution ---------------------------------------------
// For a conv with ONE output channel (dec6: ComplexConvTranspose2d 16 -> 1, c_network.py:135-141) the channel
// contraction and the spatial gather commute:  y[o] = sum_tap ( sum_ci W[tap][ci] x[src(o,tap)][ci] ).
// The inner sum is a 1x1 conv 16 -> 9 "tap channels" on the SOURCE-resolution tensor — an MFMA GEMM with full
// lanes (K = 32, N = 18) instead of an N = 2 one — and this kernel is the outer sum: 9 shifted loads per
// output pixel (nearest upsample resolved in the index).  Both passes are HBM-bound.
namespace {
// y[b][oy][ox] = sum_{dy,dx} z[b][(oy-pad_f+dy)/up_f][(ox-pad_t+dx)/up_t][dy*kw+dx]
__global__ __launch_bounds__(kThreads) void tapsum_fwd_kernel(const float2* __restrict__ z, float2* __restrict__ y,
                                                               int B, int Hs, int Ws, int CT, int kh, int kw, int up_f,
                                                               int up_t, int pad_f, int pad_t, const float* __restrict__ b_r,
                                                               const float* __restrict__ b_i) {
    const float br_ = b_r ? b_r[0] : 0.f, bi_ = b_i ? b_i[0] : 0.f;
    const float2 bias = make_float2(br_ - bi_, br_ + bi_);        // (b_r - b_i) + j (b_r + b_i): the complex layer's bias
    const int Ho = Hs * up_f, Wo = Ws * up_t;
    const long n = (long)B * Ho * Wo;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const int ox = (int)(i % Wo);
        long r = i / Wo;
        const int oy = (int)(r % Ho), b = (int)(r / Ho);
        float sr = bias.x, si = bias.y;
        for (int dy = 0; dy < kh; ++dy) {
            const int vy = oy - pad_f + dy;
            if (vy < 0 || vy >= Ho) continue;
            for (int dx = 0; dx < kw; ++dx) {
                const int vx = ox - pad_t + dx;
                if (vx < 0 || vx >= Wo) continue;
                const float2 v = z[(((long)b * Hs + vy / up_f) * Ws + vx / up_t) * CT + dy * kw + dx];
                sr += v.x; si += v.y;
            }
        }
        y[i] = make_float2(sr, si);
    }
}

// Tiled form: a workgroup owns a TOY x TOX output tile, stages the source pixels it touches (all taps of a pixel are
// contiguous: coalesced rows) in LDS once and sums from there.  The streaming form above re-fetches every source row for
// each of the up_f + kh - 1 output rows that read it (PMC: 3x the tensor's bytes from HBM at dec6).
constexpr int TOY = 16, TOX = 64, kTapLds = 10 * 34 * 9;               // float2 elements (24.5 KB): up (2,2), k = 3
// FIX = true: kh = kw = 3, up = (2, 2) as compile-time constants (dec6), so the index arithmetic is shifts
template <bool FIX>
__global__ __launch_bounds__(kThreads) void tapsum_fwd_tiled_kernel(const float2* __restrict__ z, float2* __restrict__ y,
                                                                     int B, int Hs, int Ws, int CT, int kh_, int kw_, int up_f_,
                                                                     int up_t_, int pad_f, int pad_t, int SR_, int SC_,
                                                                     const float* __restrict__ b_r, const float* __restrict__ b_i) {
    const float br_ = b_r ? b_r[0] : 0.f, bi_ = b_i ? b_i[0] : 0.f;
    const float2 bias = make_float2(br_ - bi_, br_ + bi_);
    const int kh = FIX ? 3 : kh_, kw = FIX ? 3 : kw_, up_f = FIX ? 2 : up_f_, up_t = FIX ? 2 : up_t_;
    const int SR = FIX ? 10 : SR_, SC = FIX ? 34 : SC_;
    __shared__ float2 tile[kTapLds];
    const int Ho = Hs * up_f, Wo = Ws * up_t, taps = kh * kw;
    const int tiles_x = (Wo + TOX - 1) / TOX, tiles_y = (Ho + TOY - 1) / TOY;
    const int t = threadIdx.x;
    for (long tl = blockIdx.x; tl < (long)B * tiles_y * tiles_x; tl += gridDim.x) {
        const int tx = (int)(tl % tiles_x), ty = (int)((tl / tiles_x) % tiles_y), b = (int)(tl / ((long)tiles_x * tiles_y));
        const int oy0 = ty * TOY, ox0 = tx * TOX;
        const int vy0 = oy0 - pad_f, vx0 = ox0 - pad_t;
        const int sy0 = (vy0 < 0 ? 0 : vy0) / up_f, sx0 = (vx0 < 0 ? 0 : vx0) / up_t;
        __syncthreads();
        for (int i = t; i < SR * SC * taps; i += kThreads) {
            const int tap = i % taps, px = i / taps;
            const int sx = sx0 + px % SC, sy = sy0 + px / SC;
            float2 v = make_float2(0.f, 0.f);
            if (sy < Hs && sx < Ws) v = z[(((long)b * Hs + sy) * Ws + sx) * CT + tap];
            tile[i] = v;
        }
        __syncthreads();
        for (int p = t; p < TOY * TOX; p += kThreads) {
            const int oy = oy0 + p / TOX, ox = ox0 + p % TOX;
            if (oy >= Ho || ox >= Wo) continue;
            float sr = bias.x, si = bias.y;
            for (int dy = 0; dy < kh; ++dy) {
                const int vy = oy - pad_f + dy;
                if (vy < 0 || vy >= Ho) continue;
                for (int dx = 0; dx < kw; ++dx) {
                    const int vx = ox - pad_t + dx;
                    if (vx < 0 || vx >= Wo) continue;
                    const float2 v = tile[((vy / up_f - sy0) * SC + (vx / up_t - sx0)) * taps + dy * kw + dx];
                    sr += v.x; si += v.y;
                }
            }
            y[((long)b * Ho + oy) * Wo + ox] = make_float2(sr, si);
        }
    }
}

// gz[b][my][mx][tap] = sum over the output pixels that read z[b][my][mx][tap] in the forward pass
__global__ __launch_bounds__(kThreads) void tapsum_bwd_kernel(const float2* __restrict__ gy, float2* __restrict__ gz,
                                                               int B, int Hs, int Ws, int CT, int kh, int kw, int up_f,
                                                               int up_t, int pad_f, int pad_t) {
    const int Ho = Hs * up_f, Wo = Ws * up_t;
    const long n = (long)B * Hs * Ws * CT;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const int tap = (int)(i % CT);
        long r = i / CT;
        const int mx = (int)(r % Ws); r /= Ws;
        const int my = (int)(r % Hs), b = (int)(r / Hs);
        float sr = 0.f, si = 0.f;
        if (tap < kh * kw) {
            const int dy = tap / kw, dx = tap % kw;
            for (int jy = 0; jy < up_f; ++jy) {
                const int oy = up_f * my + pad_f - dy + jy;
                if (oy < 0 || oy >= Ho) continue;
                for (int jx = 0; jx < up_t; ++jx) {
                    const int ox = up_t * mx + pad_t - dx + jx;
                    if (ox < 0 || ox >= Wo) continue;
                    const float2 v = gy[((long)b * Ho + oy) * Wo + ox];
                    sr += v.x; si += v.y;
                }
            }
        }
        gz[i] = make_float2(sr, si);
    }
}

// The network's case (3x3 taps, 2x2 upsample, pad 1: the last decoder stage) with a thread per SOURCE pixel: the 4 x 4 window of
// g_y that its nine taps read is loaded once (16 unconditional loads of clamped coordinates, masked), every tap is a sum of
// four of them, the CT tap channels go out as one contiguous run.  The generic kernel above spends ~8 runtime integer
// divisions and up to four predicated loads per ELEMENT (tap channel): 43 us for 4.7 M elements.
// OT: element type of gz — float, or bf16 (unsigned short) where the activations live in bf16 (dcs_tapsum_bwd_h: g_y, the
// cotangent of the fp32 mask, stays fp32).
template <typename OT>
__global__ __launch_bounds__(kThreads) void tapsum_bwd_3x3_up2_kernel(const float2* __restrict__ gy, OT* __restrict__ gz,
                                                                       int B, int Hs, int Ws, int CT, double* __restrict__ part) {
    const long npx = (long)B * Hs * Ws;
    const long i0 = (long)blockIdx.x * kThreads + threadIdx.x;
    const long i = i0 < npx ? i0 : npx - 1;                           // (threads past the end shadow the last pixel: barrier below)
    const int mx = (int)(i % Ws);
    const long r_ = i / Ws;
    const int my = (int)(r_ % Hs), b = (int)(r_ / Hs);
    const int Ho = 2 * Hs, Wo = 2 * Ws;
    const float2* gb = gy + (long)b * Ho * Wo;
    float2 g[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oy = 2 * my - 1 + r, oyc = oy < 0 ? 0 : (oy >= Ho ? Ho - 1 : oy);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ox = 2 * mx - 1 + c, oxc = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
            const float2 v = gb[(long)oyc * Wo + oxc];
            const bool in = oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
            g[r][c] = in ? v : make_float2(0.f, 0.f);
        }
    }
    if (part != nullptr) {
        // the bias gradient of the layer is the complex sum of g_y: every output pixel is one of the centre four of exactly one
        // thread's window, so the sum falls out of the values already loaded (csum_partial_kernel re-read all of g_y for it)
        __shared__ double red[kThreads / 64][2];
        const bool own = i0 < npx;
        double sr = own ? (double)((g[1][1].x + g[1][2].x) + (g[2][1].x + g[2][2].x)) : 0.0;
        double si = own ? (double)((g[1][1].y + g[1][2].y) + (g[2][1].y + g[2][2].y)) : 0.0;
        sr = dcs_wave_sum_d(sr); si = dcs_wave_sum_d(si);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sr; red[threadIdx.x >> 6][1] = si; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kThreads / 64; ++w) { sr += red[w][0]; si += red[w][1]; }
            part[2 * blockIdx.x] = sr; part[2 * blockIdx.x + 1] = si;
        }
    }
    float2 acc[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            // rows 2 my + 1 - dy + {0, 1} = window rows (2 - dy), (3 - dy); columns likewise; same summation order as the
            // generic kernel (jy outer, jx inner)
            float sr = 0.f, si = 0.f;
#pragma unroll
            for (int jy = 0; jy < 2; ++jy)
#pragma unroll
                for (int jx = 0; jx < 2; ++jx) { sr += g[2 - dy + jy][2 - dx + jx].x; si += g[2 - dy + jy][2 - dx + jx].y; }
            acc[dy * 3 + dx] = make_float2(sr, si);
        }
    if (CT == 16) {
        // a pixel's 16 tap channels are 128 contiguous bytes, the next pixel's the next 128: through LDS the workgroup's
        // 32 KB go out as whole float4 rows (from the registers each store instruction wrote 8 bytes to 64 different lines)
        __shared__ float2 sm[kThreads][17];
        const int t = threadIdx.x;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) sm[t][tp] = acc[tp];
#pragma unroll
        for (int tp = 9; tp < 16; ++tp) sm[t][tp] = make_float2(0.f, 0.f);
        __syncthreads();
        OT* ob = gz + (long)blockIdx.x * kThreads * 16 * 2;
        const long lim = (npx - (long)blockIdx.x * kThreads) * 16;          // elements of this workgroup that exist
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = t + kThreads * k;
            if (e < lim) dcs_st2(ob + 2 * e, sm[e >> 4][e & 15]);
        }
        return;
    }
    if (i0 >= npx) return;
    OT* o = gz + i * CT * 2;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) dcs_st2(o + 2 * tp, acc[tp]);
    for (int tp = 9; tp < CT; ++tp) dcs_st2(o + 2 * tp, make_float2(0.f, 0.f));
}
}  // namespace

// complex sum of n elements: partials (double2 per workgroup), then one workgroup -> the bias gradients of a complex layer
// whose bias enters as (b_r - b_i) + j (b_r + b_i):  g_b_r = S.re + S.im,  g_b_i = S.im - S.re
namespace {
constexpr int kSumBlocks = 256;
__global__ __launch_bounds__(kThreads) void csum_partial_kernel(const float2* __restrict__ g, long n, double* __restrict__ part) {
    __shared__ double red[kThreads / 64][2];
    double sr = 0, si = 0;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const float2 v = g[i];
        sr += v.x; si += v.y;
    }
    sr = dcs_wave_sum_d(sr); si = dcs_wave_sum_d(si);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sr; red[threadIdx.x >> 6][1] = si; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w) { sr += red[w][0]; si += red[w][1]; }
        part[2 * blockIdx.x] = sr; part[2 * blockIdx.x + 1] = si;
    }
}
__global__ __launch_bounds__(kThreads) void csum_bias_final_kernel(const double* __restrict__ part, int nparts, float* __restrict__ gb_r,
                                                                   float* __restrict__ gb_i) {
    __shared__ double red[kThreads / 64][2];
    double sr = 0, si = 0;
    for (int i = threadIdx.x; i < nparts; i += kThreads) { sr += part[2 * i]; si += part[2 * i + 1]; }
    sr = dcs_wave_sum_d(sr); si = dcs_wave_sum_d(si);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sr; red[threadIdx.x >> 6][1] = si; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w) { sr += red[w][0]; si += red[w][1]; }
        gb_r[0] = (float)(sr + si); gb_i[0] = (float)(si - sr);
    }
}
}  // namespace

extern "C" int dcs_tapsum_fwd(const float* z, float* y, const float* b_r, const float* b_i, int B, int Hs, int Ws, int CT,
                              int kh, int kw, int up_f, int up_t, int pad_f, int pad_t, dcs_stream_t stream) {
    if (!z || !y || B <= 0 || Hs <= 0 || Ws <= 0 || kh < 1 || kw < 1 || CT < kh * kw || up_f < 1 || up_t < 1 ||
        pad_f < 0 || pad_t < 0 || ((b_r == nullptr) != (b_i == nullptr)))
        return DCS_ERR_BADARG;
    const long n = (long)B * Hs * up_f * Ws * up_t;
    // source rows / columns a TOY x TOX output tile can touch
    const int SR = (TOY + kh - 2) / up_f + 2, SC = (TOX + kw - 2) / up_t + 2;
    if ((long)SR * SC * kh * kw <= kTapLds) {
        const long tiles = (long)B * ((Hs * up_f + TOY - 1) / TOY) * ((Ws * up_t + TOX - 1) / TOX);
        const dim3 grid((unsigned)(tiles < 8192 ? tiles : 8192));
        if (kh == 3 && kw == 3 && up_f == 2 && up_t == 2 && SR == 10 && SC == 34)
            DCS_LAUNCH(tapsum_fwd_tiled_kernel<true>, grid, dim3(kThreads), 0, dcs_stream(stream), (const float2*)z,
                               (float2*)y, B, Hs, Ws, CT, kh, kw, up_f, up_t, pad_f, pad_t, SR, SC, b_r, b_i);
        else
            DCS_LAUNCH(tapsum_fwd_tiled_kernel<false>, grid, dim3(kThreads), 0, dcs_stream(stream), (const float2*)z,
                               (float2*)y, B, Hs, Ws, CT, kh, kw, up_f, up_t, pad_f, pad_t, SR, SC, b_r, b_i);
        DCS_CHECK_LAUNCH();
        return DCS_OK;
    }
    DCS_LAUNCH(tapsum_fwd_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream), (const float2*)z,
                       (float2*)y, B, Hs, Ws, CT, kh, kw, up_f, up_t, pad_f, pad_t, b_r, b_i);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

constexpr long kTapSumMaxBlocks = 65536;      // partial sums the fused form may leave (one per workgroup of tapsum_bwd_3x3_up2_kernel)
extern "C" long dcs_tapsum_bwd_workspace_bytes(void) { return kTapSumMaxBlocks * 2 * (long)sizeof(double); }

static int tapsum_bwd_impl(const float* gy, void* gz, bool gz_bf16, float* gb_r, float* gb_i, void* workspace,
                           long workspace_bytes, int B, int Hs, int Ws, int CT, int kh, int kw, int up_f, int up_t, int pad_f,
                           int pad_t, dcs_stream_t stream) {
    if (!gy || !gz || B <= 0 || Hs <= 0 || Ws <= 0 || kh < 1 || kw < 1 || CT < kh * kw || up_f < 1 || up_t < 1 ||
        pad_f < 0 || pad_t < 0 || ((gb_r == nullptr) != (gb_i == nullptr)))
        return DCS_ERR_BADARG;
    if (gb_r && (!workspace || workspace_bytes < dcs_tapsum_bwd_workspace_bytes())) return DCS_ERR_WORKSPACE;
    const bool fast = kh == 3 && kw == 3 && up_f == 2 && up_t == 2 && pad_f == 1 && pad_t == 1 && (long)B * Hs * Ws < (1L << 31) * kThreads;
    const long fast_blocks = ((long)B * Hs * Ws + kThreads - 1) / kThreads;
    // bias gradients of the Cout = 1 layer = the complex sum of gy: partial sums out of the 3x3 / 2x2 kernel's own loads where
    // that kernel runs (one pass over g_y less), from csum_partial_kernel otherwise; one small launch adds them up
    const bool fused_sum = gb_r && fast && fast_blocks <= kTapSumMaxBlocks;
    if (gb_r && !fused_sum) {
        const long ny = (long)B * Hs * up_f * Ws * up_t;
        DCS_LAUNCH(csum_partial_kernel, dim3(kSumBlocks), dim3(kThreads), 0, dcs_stream(stream), (const float2*)gy, ny,
                           (double*)workspace);
        DCS_LAUNCH(csum_bias_final_kernel, dim3(1), dim3(kThreads), 0, dcs_stream(stream), (const double*)workspace,
                           kSumBlocks, gb_r, gb_i);
        DCS_CHECK_LAUNCH();
    }
    const long n = (long)B * Hs * Ws * CT;
    if (fast) {
        double* part = fused_sum ? (double*)workspace : nullptr;
        if (gz_bf16)
            DCS_LAUNCH(tapsum_bwd_3x3_up2_kernel<unsigned short>, dim3((unsigned)fast_blocks), dim3(kThreads), 0,
                       dcs_stream(stream), (const float2*)gy, (unsigned short*)gz, B, Hs, Ws, CT, part);
        else
            DCS_LAUNCH(tapsum_bwd_3x3_up2_kernel<float>, dim3((unsigned)fast_blocks), dim3(kThreads), 0,
                       dcs_stream(stream), (const float2*)gy, (float*)gz, B, Hs, Ws, CT, part);
        if (fused_sum)
            DCS_LAUNCH(csum_bias_final_kernel, dim3(1), dim3(kThreads), 0, dcs_stream(stream), (const double*)workspace,
                       (int)fast_blocks, gb_r, gb_i);
        DCS_CHECK_LAUNCH();
        return DCS_OK;
    }
    if (gz_bf16) return DCS_ERR_BADARG;                                 // (bf16 tap channels: the 3x3 / 2x2-upsample stage only)
    DCS_LAUNCH(tapsum_bwd_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream), (const float2*)gy,
                       (float2*)gz, B, Hs, Ws, CT, kh, kw, up_f, up_t, pad_f, pad_t);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_tapsum_bwd(const float* gy, float* gz, float* gb_r, float* gb_i, void* workspace, long workspace_bytes,
                              int B, int Hs, int Ws, int CT, int kh, int kw, int up_f, int up_t, int pad_f, int pad_t,
                              dcs_stream_t stream) {
    return tapsum_bwd_impl(gy, gz, false, gb_r, gb_i, workspace, workspace_bytes, B, Hs, Ws, CT, kh, kw, up_f, up_t, pad_f, pad_t,
                           stream);
}

// gz in bf16 (activations stored in bf16: BASELINE configs[4]); g_y and the bias gradients fp32
extern "C" int dcs_tapsum_bwd_h(const float* gy, unsigned short* gz, float* gb_r, float* gb_i, void* workspace,
                                long workspace_bytes, int B, int Hs, int Ws, int CT, int kh, int kw, int up_f, int up_t,
                                int pad_f, int pad_t, dcs_stream_t stream) {
    return tapsum_bwd_impl(gy, gz, true, gb_r, gb_i, workspace, workspace_bytes, B, Hs, Ws, CT, kh, kw, up_f, up_t, pad_f, pad_t,
                           stream);
}
