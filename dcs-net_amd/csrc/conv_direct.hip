// conv_direct.hip — direct (non-GEMM) complex correlation, its data / weight gradients, and the
// weight packers.
//
// Forward use: enc0 (Cin=1, K=98: SURVEY.md §7 "small-channel ends"), dec6 (Cout=1), the 7x7 2->1
// spatial-attention conv (c_network.py:74), and the generic fallback for any geometry the MFMA
// implicit-GEMM kernel (conv_mfma.hip) does not take.  These stages are HBM-bound (SURVEY.md
// §8d): one haloed input tile per workgroup is staged in LDS (each input element is read from
// HBM once per tile), COB output channels per thread live in VGPRs, weights come through the
// scalar cache (wave-uniform addresses).
//
// The virtual input (nearest upsample of cat(x1, x2): c_network.py:214-216) is resolved in the
// LDS gather, so neither the concatenated nor the upsampled tensor ever exists in HBM.
//
// Backward (what autograd does for the reference through 4 real convs per complex conv):
//   data   g_X = conj(W) (*)^T g_Y : the SAME kernel run on g_Y with zero-insertion (stride) in the
//          gather and the flipped / conjugated / in-out-swapped weight (dcs_pack_conv_weight_bwd)
//   weight g_W[tap,ci,co] = sum_p g_Y[p,co] conj(X[p*s-pad+tap, ci]) : per-tile partial slabs
//          (no atomics: bitwise reproducible) + a reduce that writes the reference's parameter
//          layout (conv_r / conv_i or conv_tran_r / conv_tran_i, and the two biases).
#include "conv_common.h"
#include "pack_jobs.h"
#include "wgrad_reduce.h"

namespace {

constexpr int TH = 16, TW = 16;      // output tile (pixels) per workgroup
constexpr int CHUNK = 8;             // input channels staged per LDS pass

using ConvArgs = conv::Args;
using conv::gather;

template <int COB>
__device__ __forceinline__ void cconv_direct_body(const ConvArgs& a, int tile_id, int co0, int b) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];   // [CHUNK][rows][colsp]
    const int t = threadIdx.x;
    const int tx = t % TW, ty = t / TW;
    const int oy0 = (tile_id / a.tiles_w) * TH, ox0 = (tile_id % a.tiles_w) * TW;
    const int Cin = a.C1 + a.C2;
    const int vy0 = oy0 * a.sf - a.pad_f, vx0 = ox0 * a.st - a.pad_t;

    float accr[COB], acci[COB];
#pragma unroll
    for (int i = 0; i < COB; ++i) { accr[i] = 0.f; acci[i] = 0.f; }

    for (int c0 = 0; c0 < Cin; c0 += CHUNK) {
        const int nc = min(CHUNK, Cin - c0);
        __syncthreads();                                   // previous pass finished reading
        const int total = a.rows * a.cols * nc;
        for (int idx = t; idx < total; idx += TH * TW) {
            const int ci = idx % nc;
            const int px = idx / nc;
            const int ix = px % a.cols, iy = px / a.cols;
            tile[ci * a.plane + iy * a.colsp + ix] = gather(a, b, vy0 + iy, vx0 + ix, c0 + ci);
        }
        __syncthreads();
        for (int ci = 0; ci < nc; ++ci) {
            const float2* pl = tile + ci * a.plane + (ty * a.sf) * a.colsp + tx * a.st;
            for (int dy = 0; dy < a.kh; ++dy) {
                for (int dx = 0; dx < a.kw; ++dx) {
                    const float2 xv = pl[dy * a.colsp + dx];
                    const float2* w = a.wp + ((long)(dy * a.kw + dx) * Cin + (c0 + ci)) * a.Cout + co0;
#pragma unroll
                    for (int i = 0; i < COB; ++i) {
                        const float2 wv = w[i];
                        accr[i] = fmaf(wv.x, xv.x, accr[i]);
                        accr[i] = fmaf(-wv.y, xv.y, accr[i]);
                        acci[i] = fmaf(wv.x, xv.y, acci[i]);
                        acci[i] = fmaf(wv.y, xv.x, acci[i]);
                    }
                }
            }
        }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < a.Hout && ox < a.Wout) {
        act2_t* out = a.y + (((long)b * a.Hout + oy) * a.Wout + ox) * a.Cout + co0;
#pragma unroll
        for (int i = 0; i < COB; ++i) {
            const float2 bv = a.bias ? a.bias[co0 + i] : make_float2(0.f, 0.f);
            float vr = accr[i] + bv.x, vi = acci[i] + bv.y;
            if (a.coef) {                                  // folded eval-mode CBN (conv_common.h)
                const float* q = a.coef + 6 * (co0 + i);
                const float ur = vr, ui = vi;
                vr = fmaf(q[0], ur, fmaf(q[1], ui, q[4])); vi = fmaf(q[2], ur, fmaf(q[3], ui, q[5]));
            }
            conv::stc(out + i, make_float2(dcs_act(vr, a.act), dcs_act(vi, a.act)));
        }
    }
}

template <int COB>
__global__ __launch_bounds__(TH * TW) void cconv_direct_kernel(ConvArgs a) {
    cconv_direct_body<COB>(a, blockIdx.x, blockIdx.y * COB, blockIdx.z);
}

// several problems whose Cout == COB (one output-channel block each) in one launch: problem = blockIdx.y
constexpr int kDirectBatch = 8;
struct DirectTable { ConvArgs p[kDirectBatch]; };

template <int COB>
__global__ __launch_bounds__(TH * TW) void cconv_direct_multi_kernel(DirectTable t) {
    const ConvArgs& a = t.p[blockIdx.y];
    if ((int)blockIdx.x >= a.tiles_w * a.tiles_h) return;
    cconv_direct_body<COB>(a, blockIdx.x, 0, blockIdx.z);
}

// ---- weight gradient ---------------------------------------------------------------------------
constexpr int WG_CO = 8;             // output channels per workgroup (g_Y tile columns)
constexpr int WG_TAPS = 13;          // taps per thread: ceil(49 / 4) covers k = 7

struct WgradArgs {
    const act2_t* x1; const act2_t* x2; const act2_t* gy; float2* slab_w; float2* slab_b;
    int n_slabs, total_tiles, n_co_chunks;
    ConvArgs c;                      // forward geometry (y/wp/bias unused)
};

// grid: (n_slabs, ci_chunks * co_chunks).  The (tap, ci, co) outputs of the workgroup's chunk pair are
// flattened over the 256 threads (<= WG_TAPS each), so the small-channel layers this kernel serves
// (enc0: 49x1x8, dec6: 9x16x1, the 7x7 2->1 attention convs: 49x2x1) keep most lanes busy.
template <int SLOTS>
__global__ __launch_bounds__(TH * TW) void cconv_wgrad_kernel(WgradArgs w) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const ConvArgs& a = w.c;
    float2* tile = lds;                                   // [CHUNK][plane]
    float2* gt = lds + CHUNK * a.plane;                   // [TH*TW][WG_CO]
    const int t = threadIdx.x;
    const int ci0 = (blockIdx.y / w.n_co_chunks) * CHUNK, co0 = (blockIdx.y % w.n_co_chunks) * WG_CO;
    const int Cin = a.C1 + a.C2;
    const int ntaps = a.kh * a.kw;
    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int nc = min(CHUNK, Cin - ci0), nco = min(WG_CO, a.Cout - co0);
    const int per_tap = nc * nco, n_out = ntaps * per_tap;

    // few outputs (Cout = 1 layers: 72..98): Q thread groups take every Q-th pixel of a tile each and are
    // combined through LDS once, after the last tile, so all 256 lanes work
    int Q = 1, q = 0, o0 = t;
    if (SLOTS == 1 && 2 * n_out <= TH * TW) {
        Q = (TH * TW) / n_out;
        q = t / n_out;
        o0 = q < Q ? t % n_out : n_out;                   // lanes beyond Q * n_out stay idle
    }

    float accr[SLOTS], acci[SLOTS];
    int xoff[SLOTS], goff[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        accr[i] = 0.f; acci[i] = 0.f;
        const int o = o0 + TH * TW * i;
        const int tap = o < n_out ? o / per_tap : 0, r = o < n_out ? o % per_tap : 0;
        xoff[i] = (r / nco) * a.plane + (tap / a.kw) * a.colsp + (tap % a.kw);
        goff[i] = r % nco;
    }
    float br = 0.f, bi = 0.f;

    for (int tl = blockIdx.x; tl < w.total_tiles; tl += w.n_slabs) {
        const int b = tl / tiles_per_img, tile_id = tl % tiles_per_img;
        const int oy0 = (tile_id / a.tiles_w) * TH, ox0 = (tile_id % a.tiles_w) * TW;
        const int vy0 = oy0 * a.sf - a.pad_f, vx0 = ox0 * a.st - a.pad_t;
        __syncthreads();
        const int total = a.rows * a.cols * nc;
        for (int idx = t; idx < total; idx += TH * TW) {
            const int ci = idx % nc, px = idx / nc;
            const int ix = px % a.cols, iy = px / a.cols;
            tile[ci * a.plane + iy * a.colsp + ix] = gather(a, b, vy0 + iy, vx0 + ix, ci0 + ci);
        }
        for (int idx = t; idx < TH * TW * WG_CO; idx += TH * TW) {
            const int co = idx % WG_CO, p = idx / WG_CO;
            const int oy = oy0 + p / TW, ox = ox0 + p % TW;
            float2 v = make_float2(0.f, 0.f);
            if (oy < a.Hout && ox < a.Wout && co0 + co < a.Cout)
                v = conv::ldc(w.gy + (((long)b * a.Hout + oy) * a.Wout + ox) * a.Cout + co0 + co);
            gt[p * WG_CO + co] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int p = q; p < TH * TW; p += Q) {
            const int base = ((p / TW) * a.sf) * a.colsp + (p % TW) * a.st;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                if (o0 + TH * TW * i < n_out) {
                    const float2 g = gt[p * WG_CO + goff[i]];
                    const float2 xv = tile[xoff[i] + base];
                    // g * conj(x)
                    accr[i] = fmaf(g.x, xv.x, accr[i]);
                    accr[i] = fmaf(g.y, xv.y, accr[i]);
                    acci[i] = fmaf(g.y, xv.x, acci[i]);
                    acci[i] = fmaf(-g.x, xv.y, acci[i]);
                }
            }
        }
        if (ci0 == 0 && t < WG_CO) {                      // bias: column sums of g_Y, once per co chunk
            for (int p = 0; p < TH * TW; ++p) {
                const float2 g = gt[p * WG_CO + t];
                br += g.x; bi += g.y;
            }
        }
    }
    const long wsz = (long)ntaps * Cin * a.Cout;
    if (Q > 1) {                                          // combine the pixel shares (SLOTS == 1 here)
        __syncthreads();
        if (o0 < n_out) lds[q * n_out + o0] = make_float2(accr[0], acci[0]);
        __syncthreads();
        if (q == 0) {
            for (int k = 1; k < Q; ++k) { const float2 v = lds[k * n_out + o0]; accr[0] += v.x; acci[0] += v.y; }
        } else {
            o0 = n_out;                                   // only group 0 writes the slab
        }
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int o = o0 + TH * TW * i;
        if (o < n_out) {
            const int tap = o / per_tap, r = o % per_tap;
            w.slab_w[(long)blockIdx.x * wsz + ((long)tap * Cin + ci0 + r / nco) * a.Cout + co0 + r % nco] =
                make_float2(accr[i], acci[i]);
        }
    }
    if (ci0 == 0 && t < WG_CO && co0 + t < a.Cout) w.slab_b[(long)blockIdx.x * a.Cout + co0 + t] = make_float2(br, bi);
}

// sum the slabs and scatter into the reference's parameter layout (wgrad_reduce.hip: immediate, or deferred + batched)
int launch_wgrad_reduce(const float2* slab_w, const float2* slab_b, int n_slabs, float* gw_r, float* gw_i, float* gb_r,
                        float* gb_i, int Cout, int Cin, int kh, int kw, int transposed, hipStream_t s) {
    wreduce::Job j{};
    j.slab_w = slab_w; j.slab_b = slab_b; j.gw_r = gw_r; j.gw_i = gw_i; j.gb_r = gb_r; j.gb_i = gb_i;
    j.n_slabs = n_slabs; j.Cout = Cout; j.Cin = Cin; j.kh = kh; j.kw = kw; j.transposed = transposed;
    return wreduce::emit(j, s);
}

// ---- packers ------------------------------------------------------------------------------------
// (the re-layout kernels themselves live in pack_jobs.hip: packjob::DIRECT / packjob::BWD)

// g_x1[b][y][x][c] = sum over the up_f x up_t block of g_Xv[b][..][..][c]; channels >= C1 go to g_x2
__global__ void upsample_cat_bwd_kernel(const act2_t* __restrict__ gxv, act2_t* __restrict__ gx1,
                                        act2_t* __restrict__ gx2, int B, int Hin, int Win, int C1, int C2, int up_f,
                                        int up_t) {
    const int C = C1 + C2;
    const long n = (long)B * Hin * Win * C;
    const int Wv = Win * up_t, Hv = Hin * up_f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long p = i / C;
        const int x = (int)(p % Win); p /= Win;
        const int y = (int)(p % Hin);
        const int b = (int)(p / Hin);
        float sr = 0.f, si = 0.f;
        for (int dy = 0; dy < up_f; ++dy)
            for (int dx = 0; dx < up_t; ++dx) {
                const float2 v = conv::ldc(gxv + (((long)b * Hv + y * up_f + dy) * Wv + x * up_t + dx) * C + c);
                sr += v.x; si += v.y;
            }
        const long sp = ((long)b * Hin + y) * Win + x;
        if (c < C1) conv::stc(gx1 + sp * C1 + c, make_float2(sr, si));
        else        conv::stc(gx2 + sp * C2 + (c - C1), make_float2(sr, si));
    }
}

// direct + MFMA panels of a (CoutRole, CinRole) weight
long base_floats(int CoutRole, int CinRole, int taps) {
    return conv::direct_floats(CoutRole, CinRole, taps) + conv::mfma_floats(CoutRole, CinRole, taps);
}

bool conv_geometry(ConvArgs& a) {
    if (a.Hout <= 0 || a.Wout <= 0) return false;
    a.tiles_w = (a.Wout + TW - 1) / TW;
    a.tiles_h = (a.Hout + TH - 1) / TH;
    a.rows = (TH - 1) * a.sf + a.kh;
    a.cols = (TW - 1) * a.st + a.kw;
    a.colsp = a.cols | 1;                       // odd row pitch (in float2) spreads LDS banks
    a.plane = a.rows * a.colsp + 1;
    return true;
}

// n plain correlations with the same batch size and Cout in {1, 2} (the spatial-attention convs and their data
// gradients) as one launch; `a`: geometry with Hout / Wout set (tiling is filled in here)
int launch_direct_multi(ConvArgs* a, int n, hipStream_t stream) {
    if (n < 1 || n > kDirectBatch) return DCS_ERR_BADARG;
    if (dcs_conv_k7_ok(a, n)) return dcs_conv_k7_launch(a, n, stream);          // the attention convs: conv_k7.hip
    DirectTable t;
    size_t lds = 0;
    int tiles = 0;
    const int cob = a[0].Cout;
    for (int i = 0; i < n; ++i) {
        if (!conv_geometry(a[i]) || a[i].Cout != cob || a[i].B != a[0].B || (cob != 1 && cob != 2)) return DCS_ERR_BADARG;
        const int Cin = a[i].C1 + a[i].C2;
        const size_t l = (size_t)(Cin < CHUNK ? Cin : CHUNK) * a[i].plane * sizeof(float2);
        lds = l > lds ? l : lds;
        tiles = a[i].tiles_w * a[i].tiles_h > tiles ? a[i].tiles_w * a[i].tiles_h : tiles;
        t.p[i] = a[i];
    }
    if (lds > 64 * 1024 || a[0].B > 65535) return DCS_ERR_BADARG;
    dim3 grid(tiles, n, a[0].B);
    if (cob == 1) DCS_LAUNCH(cconv_direct_multi_kernel<1>, grid, dim3(TH * TW), lds, stream, t);
    else DCS_LAUNCH(cconv_direct_multi_kernel<2>, grid, dim3(TH * TW), lds, stream, t);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

int launch_direct(ConvArgs& a, hipStream_t stream) {
    if (!DCS_ACT_IS_BF16 && dcs_conv_k7_ok(&a, 1)) return dcs_conv_k7_launch(&a, 1, stream);        // the attention convs: conv_k7.hip (fp32 maps)
    if (dcs_conv_enc0_ok(a)) return dcs_conv_enc0_launch(a, stream);            // the first encoder conv: conv_enc0.hip
    if (!conv_geometry(a)) return DCS_ERR_BADARG;
    const int Cin = a.C1 + a.C2;
    const size_t lds = (size_t)(Cin < CHUNK ? Cin : CHUNK) * a.plane * sizeof(float2);
    if (lds > 150 * 1024) return DCS_ERR_BADARG;
    const int Cout = a.Cout;
    int cob = (Cout % 8 == 0) ? 8 : (Cout % 4 == 0) ? 4 : (Cout % 2 == 0) ? 2 : 1;
    if (lds > 64 * 1024) {   // above the default dynamic-LDS limit: raise it for this instantiation
        const void* fn = cob == 8 ? (const void*)cconv_direct_kernel<8> : cob == 4 ? (const void*)cconv_direct_kernel<4>
                       : cob == 2 ? (const void*)cconv_direct_kernel<2> : (const void*)cconv_direct_kernel<1>;
        if (dcs_ensure_dynamic_lds(fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
    }
    dim3 grid(a.tiles_w * a.tiles_h, Cout / cob, a.B);
    if (grid.y > 65535 || grid.z > 65535) return DCS_ERR_BADARG;
    switch (cob) {
        case 8: DCS_LAUNCH(cconv_direct_kernel<8>, grid, dim3(TH * TW), lds, stream, a); break;
        case 4: DCS_LAUNCH(cconv_direct_kernel<4>, grid, dim3(TH * TW), lds, stream, a); break;
        case 2: DCS_LAUNCH(cconv_direct_kernel<2>, grid, dim3(TH * TW), lds, stream, a); break;
        default: DCS_LAUNCH(cconv_direct_kernel<1>, grid, dim3(TH * TW), lds, stream, a); break;
    }
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

ConvArgs fwd_args(const void* x1, const void* x2, int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                  int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t) {
    ConvArgs a{};
    a.x1 = (const act2_t*)x1; a.x2 = (const act2_t*)x2;
    a.B = B; a.Hin = Hin; a.Win = Win; a.C1 = C1; a.C2 = C2; a.up_f = up_f; a.up_t = up_t; a.zero_ins = 0;
    a.Cout = Cout; a.kh = kh; a.kw = kw; a.sf = sf; a.st = st; a.pad_f = pad_f; a.pad_t = pad_t; a.act = DCS_ACT_NONE;
    a.Hv = Hin * up_f; a.Wv = Win * up_t;
    a.Hout = (a.Hv + 2 * pad_f - kh) / sf + 1;
    a.Wout = (a.Wv + 2 * pad_t - kw) / st + 1;
    return a;
}

bool fwd_geom_ok(const void* x1, const void* x2, int B, int Hin, int Win, int C1, int C2, int up_f, int up_t, int Cout,
                 int kh, int kw, int sf, int st, int pad_f, int pad_t) {
    if (!x1 || B <= 0 || Hin <= 0 || Win <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0) return false;
    if ((C2 > 0) != (x2 != nullptr)) return false;
    if (up_f < 1 || up_t < 1 || kh < 1 || kw < 1 || sf < 1 || st < 1 || pad_f < 0 || pad_t < 0) return false;
    return true;
}

int wgrad_slabs(const ConvArgs& a, long* wsz_out) {
    const long wsz = (long)a.kh * a.kw * (a.C1 + a.C2) * a.Cout;
    const long total_tiles = (long)a.tiles_w * a.tiles_h * a.B;
    long cap = (32L << 20) / (wsz * (long)sizeof(float2));
    if (cap < 1) cap = 1;
    if (cap > 1024) cap = 1024;
    *wsz_out = wsz;
    return (int)(total_tiles < cap ? total_tiles : cap);
}

}  // namespace

#ifndef DCS_ACT_BF16
int dcs_conv_direct_multi(conv::Args* a, int n, hipStream_t stream) { return launch_direct_multi(a, n, stream); }
#endif

#ifndef DCS_ACT_BF16
extern "C" int dcs_pack_conv_weight(const float* w_r, const float* w_i, const float* b_r, const float* b_i, float* wp,
                                    float* bias_out, int Cout, int Cin, int kh, int kw, int transposed, int up_f,
                                    int up_t, dcs_stream_t stream) {
    if (up_f < 1 || up_t < 1) return DCS_ERR_BADARG;
    if (!w_r || !w_i || !wp || !bias_out || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0) return DCS_ERR_BADARG;
    if ((b_r == nullptr) != (b_i == nullptr)) return DCS_ERR_BADARG;
    long n = (long)kh * kw * Cin * Cout;
    if (n < Cout) n = Cout;
    packjob::Job j{};
    j.kind = packjob::DIRECT;
    j.Cout = Cout; j.Cin = Cin; j.kh = kh; j.kw = kw; j.flag = transposed;
    j.total = n;
    j.dst_bytes = (long)kh * kw * Cin * Cout * (long)sizeof(float2);
    j.src0 = w_r; j.src1 = w_i; j.src2 = b_r; j.src3 = b_i;
    j.dst0 = wp; j.dst1 = bias_out;
    {
        const int rc = packjob::emit(j, dcs_stream(stream));
        if (rc != DCS_OK) return rc;
    }
    if (conv::mfma_ok(Cin, Cout)) {    // second panel: MFMA fragment order (conv_mfma.hip)
        const int rc = dcs_conv_mfma_pack(wp, wp + conv::direct_floats(Cout, Cin, kh * kw), Cout, Cin, kh * kw,
                                          dcs_stream(stream));
        if (rc != DCS_OK) return rc;
    }
    if (conv::fold_ok(Cin, Cout, kh, kw, 1, 1, 1, 1, up_f, up_t))   // third: per-parity folded sub-kernels (conv_pack.hip)
        return conv::pack_fold(wp, wp + base_floats(Cout, Cin, kh * kw), Cout, Cin, up_f, up_t, dcs_stream(stream));
    return DCS_OK;
}
#endif

#ifndef DCS_ACT_BF16
extern "C" long dcs_packed_weight_floats(int Cout, int Cin, int kh, int kw, int up_f, int up_t) {
    if (Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || up_f < 1 || up_t < 1) return -1;
    long n = base_floats(Cout, Cin, kh * kw);
    if (conv::fold_ok(Cin, Cout, kh, kw, 1, 1, 1, 1, up_f, up_t)) n += conv::fold_floats(Cout, Cin, up_f, up_t);
    return n;
}
#endif

#ifndef DCS_ACT_BF16
extern "C" long dcs_packed_weight_bwd_floats(int Cout, int Cin, int kh, int kw, int sf, int st, int pad_f, int pad_t,
                                             int up_f, int up_t) {
    if (Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || sf < 1 || st < 1 || up_f < 1 || up_t < 1) return -1;
    long n = base_floats(Cin, Cout, kh * kw);
    if (conv::fold_ok(Cin, Cout, kh, kw, sf, st, pad_f, pad_t, up_f, up_t)) n += conv::upfold_bwd_floats(Cout, Cin, up_f, up_t);
    else if (up_f * up_t == 1 && conv::stride_ok(Cin, Cout, kh, kw, sf, st, pad_f, pad_t))
        n += conv::stride_bwd_floats(Cout, Cin, kh, kw, sf, st, pad_f, pad_t);
    return n;
}
#endif

#ifndef DCS_ACT_BF16
extern "C" int dcs_pack_conv_weight_bwd(const float* wp, float* wp_bwd, int Cout, int Cin, int kh, int kw, int sf,
                                        int st, int pad_f, int pad_t, int up_f, int up_t, dcs_stream_t stream) {
    if (!wp || !wp_bwd || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || sf < 1 || st < 1 || up_f < 1 || up_t < 1)
        return DCS_ERR_BADARG;
    const long n = (long)kh * kw * Cin * Cout;
    hipStream_t s = dcs_stream(stream);
    packjob::Job j{};
    j.kind = packjob::BWD;
    j.Cout = Cout; j.Cin = Cin; j.kh = kh * kw;
    j.total = n;
    j.dst_bytes = n * (long)sizeof(float2);
    j.src0 = wp; j.dst0 = wp_bwd;
    {
        const int rc = packjob::emit(j, s);
        if (rc != DCS_OK) return rc;
    }
    // in the data-gradient GEMM the roles swap: K runs over the forward Cout, N over the forward Cin
    if (conv::mfma_ok(Cout, Cin)) {
        const int rc = dcs_conv_mfma_pack(wp_bwd, wp_bwd + conv::direct_floats(Cin, Cout, kh * kw), Cin, Cout, kh * kw, s);
        if (rc != DCS_OK) return rc;
    }
    float* extra = wp_bwd + base_floats(Cin, Cout, kh * kw);
    if (conv::fold_ok(Cin, Cout, kh, kw, sf, st, pad_f, pad_t, up_f, up_t))
        return conv::pack_upfold_bwd(wp, extra, Cout, Cin, up_f, up_t, s);
    if (up_f * up_t == 1 && conv::stride_ok(Cin, Cout, kh, kw, sf, st, pad_f, pad_t))
        return conv::pack_stride_bwd(wp_bwd, extra, Cout, Cin, kh, kw, sf, st, pad_f, pad_t, s);
    return DCS_OK;
}
#endif

// forward launch description shared by the workspace query and the launch itself
struct FwdPlan { ConvArgs a; int path, ncls, os_f, os_t; conv::Cls cls[4]; };   // path 0: folded classes, 1: plain MFMA, 2: direct

static FwdPlan fwd_plan(const void* x1, const void* x2, int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                        int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t) {
    FwdPlan p{};
    if (!(C1 & 1) && conv::fold_ok(C1 + C2, Cout, kh, kw, sf, st, pad_f, pad_t, up_f, up_t)) {
        // nearest upsample folded into per-parity sub-kernels on the SOURCE tensors (conv_pack.hip)
        p.a = fwd_args(x1, x2, B, Hin, Win, C1, C2, 1, 1, Cout, kh, kw, 1, 1, pad_f, pad_t);
        p.a.Hout = Hin * up_f; p.a.Wout = Win * up_t;
        p.path = 0; p.ncls = up_f * up_t; p.os_f = up_f; p.os_t = up_t;
        conv::fold_classes(Cout, C1 + C2, up_f, up_t, Hin, Win, p.cls);
        return p;
    }
    p.a = fwd_args(x1, x2, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t);
    p.path = (conv::mfma_ok(C1 + C2, Cout) && !(C1 & 1)) ? 1 : 2;
    return p;
}

#ifndef DCS_ACT_BF16
extern "C" long dcs_cconv2d_fwd_workspace_bytes(int B, int Hin, int Win, int C1, int C2, int up_f, int up_t, int Cout,
                                                int kh, int kw, int sf, int st, int pad_f, int pad_t) {
    if (B <= 0 || Hin <= 0 || Win <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || up_f < 1 || up_t < 1 || kh < 1 || kw < 1 ||
        sf < 1 || st < 1 || pad_f < 0 || pad_t < 0)
        return -1;
    const FwdPlan p = fwd_plan(nullptr, nullptr, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t);
    if (p.a.Hout <= 0 || p.a.Wout <= 0) return -1;
    if (p.path == 0) return dcs_conv_mfma_workspace_bytes(p.a, p.ncls, p.cls);
    if (p.path == 1) return dcs_conv_mfma_workspace_bytes_plain(p.a);
    return 0;
}
#endif

extern "C" int DCS_SYM(dcs_cconv2d_fwd_affine)(const act_t* x1, const act_t* x2, const float* wp, const float* bias, const float* coef,
                                      act_t* y, void* workspace, long workspace_bytes, int B, int Hin, int Win, int C1, int C2,
                                      int up_f, int up_t, int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t,
                                      int act, dcs_stream_t stream);
extern "C" int DCS_SYM(dcs_cconv2d_fwd)(const act_t* x1, const act_t* x2, const float* wp, const float* bias, act_t* y,
                               void* workspace, long workspace_bytes, int B, int Hin, int Win, int C1, int C2, int up_f,
                               int up_t, int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int act,
                               dcs_stream_t stream) {
    return DCS_SYM(dcs_cconv2d_fwd_affine)(x1, x2, wp, bias, nullptr, y, workspace, workspace_bytes, B, Hin, Win, C1, C2, up_f, up_t, Cout,
                                  kh, kw, sf, st, pad_f, pad_t, act, stream);
}

extern "C" int DCS_SYM(dcs_cconv2d_fwd_affine)(const act_t* x1, const act_t* x2, const float* wp, const float* bias, const float* coef,
                                      act_t* y, void* workspace, long workspace_bytes, int B, int Hin, int Win, int C1, int C2,
                                      int up_f, int up_t, int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t,
                                      int act, dcs_stream_t stream) {
    if (!wp || !bias || !y) return DCS_ERR_BADARG;
    if (!fwd_geom_ok(x1, x2, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t)) return DCS_ERR_BADARG;
    if (act < DCS_ACT_NONE || act > DCS_ACT_SIGMOID) return DCS_ERR_BADARG;
    FwdPlan p = fwd_plan(x1, x2, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t);
    p.a.wp = (const float2*)wp; p.a.bias = (const float2*)bias; p.a.y = (act2_t*)y; p.a.act = act; p.a.coef = coef;
    if (p.path == 0)
        return dcs_conv_mfma_launch_classes(p.a, wp + base_floats(Cout, C1 + C2, kh * kw), p.ncls, p.cls, p.os_f, p.os_t,
                                            nullptr, 0, workspace, workspace_bytes, dcs_stream(stream));
    if (p.path == 1)
        return dcs_conv_mfma_launch(p.a, wp + conv::direct_floats(Cout, C1 + C2, kh * kw), workspace, workspace_bytes,
                                    dcs_stream(stream));
    return launch_direct(p.a, dcs_stream(stream));
}

// Rows of CBN partial sums (float[rows][Cout][5]) dcs_cconv2d_fwd_stats writes for this geometry when handed the split-K
// scratch dcs_cconv2d_fwd_workspace_bytes asks for; 0: the geometry has no statistics epilogue (run dcs_cbn_fwd instead).
static int fwd_stat_rows(const FwdPlan& p, bool have_ws) {
    if (p.path == 0) return dcs_conv_mfma_stat_rows(p.a, p.ncls, p.cls, have_ws);
    if (p.path == 1) {
        conv::Cls c;
        c.kh = p.a.kh; c.kw = p.a.kw; c.pad_f = p.a.pad_f; c.pad_t = p.a.pad_t; c.oo_f = 0; c.oo_t = 0;
        c.Hc = p.a.Hout; c.Wc = p.a.Wout; c.bm_off = 0;
        return dcs_conv_mfma_stat_rows(p.a, 1, &c, have_ws);
    }
    if (!dcs_conv_k7_ok(&p.a, 1) && p.a.C2 == 0) return dcs_conv_enc0_stat_rows(p.a);
    return 0;
}

#ifndef DCS_ACT_BF16
extern "C" int dcs_cconv2d_fwd_stats_rows(int B, int Hin, int Win, int C1, int C2, int up_f, int up_t, int Cout, int kh,
                                          int kw, int sf, int st, int pad_f, int pad_t) {
    if (B <= 0 || Hin <= 0 || Win <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || up_f < 1 || up_t < 1 || kh < 1 || kw < 1 ||
        sf < 1 || st < 1 || pad_f < 0 || pad_t < 0)
        return -1;
    const FwdPlan p = fwd_plan(nullptr, nullptr, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t);
    if (p.a.Hout <= 0 || p.a.Wout <= 0) return -1;
    return fwd_stat_rows(p, true);
}
#endif

// dcs_cconv2d_fwd (no activation) that ALSO leaves the training-mode CBN statistics of its raw output: stat =
// float[Cout][5][stat_rows], column r = one workgroup's partial {S_r, S_i, S_rr, S_ii, S_ri} of (y - bias), fixed order;
// *rows_used (host int) = the rows actually written (<= stat_rows = dcs_cconv2d_fwd_stats_rows).  Feeds dcs_cbn_fwd_slabs
// with pivot = bias: the CBN's statistics pass over y (c_network.py:107-114: conv, CBN back to back) disappears.
extern "C" int DCS_SYM(dcs_cconv2d_fwd_stats)(const act_t* x1, const act_t* x2, const float* wp, const float* bias, act_t* y,
                                     float* stat, int stat_rows, int* rows_used, void* workspace, long workspace_bytes, int B,
                                     int Hin, int Win, int C1, int C2, int up_f, int up_t, int Cout, int kh, int kw, int sf,
                                     int st, int pad_f, int pad_t, dcs_stream_t stream) {
    if (!wp || !bias || !y || !stat || !rows_used || stat_rows < 1) return DCS_ERR_BADARG;
    if (!fwd_geom_ok(x1, x2, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t)) return DCS_ERR_BADARG;
    FwdPlan p = fwd_plan(x1, x2, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t);
    const long need = p.path == 0 ? dcs_conv_mfma_workspace_bytes(p.a, p.ncls, p.cls)
                    : p.path == 1 ? dcs_conv_mfma_workspace_bytes_plain(p.a) : 0;
    const bool have_ws = need > 0 && workspace != nullptr && workspace_bytes >= need;
    const int rows = fwd_stat_rows(p, have_ws);
    if (rows < 1 || rows > stat_rows) return DCS_ERR_BADARG;
    *rows_used = rows;
    p.a.wp = (const float2*)wp; p.a.bias = (const float2*)bias; p.a.y = (act2_t*)y; p.a.act = DCS_ACT_NONE; p.a.coef = nullptr;
    p.a.stat = stat; p.a.stat_stride = stat_rows;
    if (p.path == 0)
        return dcs_conv_mfma_launch_classes(p.a, wp + base_floats(Cout, C1 + C2, kh * kw), p.ncls, p.cls, p.os_f, p.os_t,
                                            nullptr, 0, workspace, workspace_bytes, dcs_stream(stream));
    if (p.path == 1)
        return dcs_conv_mfma_launch(p.a, wp + conv::direct_floats(Cout, C1 + C2, kh * kw), workspace, workspace_bytes,
                                    dcs_stream(stream));
    return dcs_conv_enc0_launch(p.a, dcs_stream(stream));
}

// ---- real-valued convolution on the same MFMA kernel (DR-Net: r_network.py:60-66, :90-102) -------------------------
// A real NHWC activation with an even channel count IS an interleaved "complex" one with half as many channels, and the
// complex kernel's GEMM is a plain real GEMM over K = taps x real input channels, N = real output channels whose B
// panel happens to have a 2x2 block structure.  A real conv only needs a B panel WITHOUT that structure: the caller
// packs B[tap][k][n] = w[n][k][tap] into the 32-column fragment order (dcsnet/r_network.py) and this entry runs it.
static bool rconv_ok(int C1r, int C2r, int Coutr) {
    if ((C1r & 1) || (C2r & 1) || (Coutr & 1)) return false;
    return conv::mfma_ok((C1r + C2r) / 2, Coutr / 2) && !((C1r / 2) & 1);
}

#ifndef DCS_ACT_BF16
extern "C" long dcs_rconv2d_fwd_workspace_bytes(int B, int Hin, int Win, int C1r, int C2r, int up_f, int up_t, int Coutr,
                                                int kh, int kw, int sf, int st, int pad_f, int pad_t) {
    if (!rconv_ok(C1r, C2r, Coutr) || B <= 0 || Hin <= 0 || Win <= 0) return -1;
    const ConvArgs a = fwd_args(nullptr, nullptr, B, Hin, Win, C1r / 2, C2r / 2, up_f, up_t, Coutr / 2, kh, kw, sf, st, pad_f,
                                pad_t);
    return dcs_conv_mfma_workspace_bytes_plain(a);
}
#endif

#ifndef DCS_ACT_BF16
extern "C" int dcs_rconv2d_fwd(const float* x1, const float* x2, const float* bm, const float* bias, float* y,
                               void* workspace, long workspace_bytes, int B, int Hin, int Win, int C1r, int C2r, int up_f,
                               int up_t, int Coutr, int kh, int kw, int sf, int st, int pad_f, int pad_t, int act,
                               dcs_stream_t stream) {
    if (!bm || !y || !rconv_ok(C1r, C2r, Coutr)) return DCS_ERR_BADARG;
    if (!fwd_geom_ok(x1, x2, B, Hin, Win, C1r / 2, C2r / 2, up_f, up_t, Coutr / 2, kh, kw, sf, st, pad_f, pad_t))
        return DCS_ERR_BADARG;
    if (act < DCS_ACT_NONE || act > DCS_ACT_SIGMOID) return DCS_ERR_BADARG;
    ConvArgs a = fwd_args(x1, x2, B, Hin, Win, C1r / 2, C2r / 2, up_f, up_t, Coutr / 2, kh, kw, sf, st, pad_f, pad_t);
    a.wp = nullptr; a.bias = (const float2*)bias; a.y = (float2*)y; a.act = act;
    return dcs_conv_mfma_launch_wide(a, bm, workspace, workspace_bytes, dcs_stream(stream));
}
#endif

// Gradient of dcs_rconv2d_fwd's VIRTUAL input (the upsampled concatenation), float[B][Hv][Wv][Cinr]: a stride-1
// correlation of the zero-inserted g_Y with the caller-packed panel of the flipped, in/out-swapped kernel
// (B[tap][k = real output channel][n = real input channel]), padding k-1-p.  The block sum over an upsample and the
// channel split of a concatenation are dcs_upsample_cat_bwd's (complex channel counts Cr/2).
#ifndef DCS_ACT_BF16
static conv::Args rconv_dgrad_args(const float* gy, float* gxv, int B, int Hv, int Wv, int Cinr, int Coutr, int kh, int kw,
                                   int sf, int st, int pad_f, int pad_t) {
    ConvArgs a{};
    const int Hout = (Hv + 2 * pad_f - kh) / sf + 1, Wout = (Wv + 2 * pad_t - kw) / st + 1;
    a.x1 = (const float2*)gy; a.x2 = nullptr; a.bias = nullptr; a.wp = nullptr; a.y = (float2*)gxv;
    a.B = B; a.Hin = Hout; a.Win = Wout; a.C1 = Coutr / 2; a.C2 = 0; a.Cout = Cinr / 2; a.act = DCS_ACT_NONE;
    a.up_f = sf; a.up_t = st; a.zero_ins = 1;
    a.kh = kh; a.kw = kw; a.sf = 1; a.st = 1; a.pad_f = kh - 1 - pad_f; a.pad_t = kw - 1 - pad_t;
    a.Hv = (Hout - 1) * sf + 1; a.Wv = (Wout - 1) * st + 1;
    a.Hout = Hv; a.Wout = Wv;
    return a;
}
#endif

#ifndef DCS_ACT_BF16
extern "C" long dcs_rconv2d_bwd_data_workspace_bytes(int B, int Hv, int Wv, int Cinr, int Coutr, int kh, int kw, int sf,
                                                     int st, int pad_f, int pad_t) {
    if (B <= 0 || Hv <= 0 || Wv <= 0 || !rconv_ok(Coutr, 0, Cinr) || kh < 1 || kw < 1 || sf < 1 || st < 1 || pad_f < 0 ||
        pad_t < 0 || pad_f > kh - 1 || pad_t > kw - 1)
        return -1;
    const ConvArgs a = rconv_dgrad_args(nullptr, nullptr, B, Hv, Wv, Cinr, Coutr, kh, kw, sf, st, pad_f, pad_t);
    if (a.Hin <= 0 || a.Win <= 0) return -1;
    return dcs_conv_mfma_workspace_bytes_plain(a);
}
#endif

#ifndef DCS_ACT_BF16
extern "C" int dcs_rconv2d_bwd_data(const float* gy, const float* bm_bwd, float* gxv, void* workspace, long workspace_bytes,
                                    int B, int Hv, int Wv, int Cinr, int Coutr, int kh, int kw, int sf, int st, int pad_f,
                                    int pad_t, dcs_stream_t stream) {
    if (!gy || !bm_bwd || !gxv || B <= 0 || Hv <= 0 || Wv <= 0 || !rconv_ok(Coutr, 0, Cinr)) return DCS_ERR_BADARG;
    if (kh < 1 || kw < 1 || sf < 1 || st < 1 || pad_f < 0 || pad_t < 0 || pad_f > kh - 1 || pad_t > kw - 1)
        return DCS_ERR_BADARG;
    ConvArgs a = rconv_dgrad_args(gy, gxv, B, Hv, Wv, Cinr, Coutr, kh, kw, sf, st, pad_f, pad_t);
    if (a.Hin <= 0 || a.Win <= 0) return DCS_ERR_BADARG;
    return dcs_conv_mfma_launch_wide(a, bm_bwd, workspace, workspace_bytes, dcs_stream(stream));
}
#endif

// data-gradient launch description shared by the workspace query and the launch itself
struct DgradPlan {
    ConvArgs a{};
    int path;                  // 0: gradient of the upsample-folded conv, 1: enc0 class kernel, 2: stride classes (MFMA),
                               // 3: zero-insertion MFMA, 4: zero-insertion direct
    int ncls, os_f, os_t;
    conv::Cls cls[4];
    long gxv_bytes;            // gradient of the virtual (upsampled / concatenated) input, 0 when written straight to g_x1
    bool split;                // path 3 writes g_x1 | g_x2 from its epilogue
    int Hv, Wv, Hout, Wout;
};

static DgradPlan dgrad_plan(const void* gy, int B, int Hin, int Win, int C1, int C2, int up_f, int up_t, int Cout, int kh,
                            int kw, int sf, int st, int pad_f, int pad_t) {
    DgradPlan p{};
    const int Cin = C1 + C2;
    p.Hv = Hin * up_f; p.Wv = Win * up_t;
    p.Hout = (p.Hv + 2 * pad_f - kh) / sf + 1; p.Wout = (p.Wv + 2 * pad_t - kw) / st + 1;
    ConvArgs& a = p.a;
    a.x1 = (const act2_t*)gy; a.x2 = nullptr; a.bias = nullptr;
    a.B = B; a.Hin = p.Hout; a.Win = p.Wout; a.C1 = Cout; a.C2 = 0; a.Cout = Cin; a.act = DCS_ACT_NONE;
    p.ncls = 1; p.os_f = 1; p.os_t = 1;
    if (!(C1 & 1) && conv::fold_ok(Cin, Cout, kh, kw, sf, st, pad_f, pad_t, up_f, up_t)) {
        // gradient w.r.t. the SOURCE of the upsampled conv: stride-up correlation over g_Y with the effective
        // 4-tap kernel; columns split into g_x1 | g_x2 in the epilogue (no g_Xv, no block-sum pass)
        a.up_f = 1; a.up_t = 1; a.zero_ins = 0; a.Hv = p.Hout; a.Wv = p.Wout;
        a.kh = up_f == 2 ? 4 : 3; a.kw = up_t == 2 ? 4 : 3; a.sf = up_f; a.st = up_t; a.pad_f = 1; a.pad_t = 1;
        a.Hout = Hin; a.Wout = Win;
        conv::Cls& c = p.cls[0];
        c.kh = a.kh; c.kw = a.kw; c.pad_f = 1; c.pad_t = 1; c.oo_f = 0; c.oo_t = 0; c.Hc = Hin; c.Wc = Win;
        c.bm_off = conv::direct_floats(Cin, Cout, a.kh * a.kw);
        p.path = 0;
        return p;
    }
    // stride-1 conv over a plain concatenation (no upsample): the MFMA epilogue splits its columns into g_x1 | g_x2, so the
    // gradient of the virtual input is never materialised (dec6's 1x1 tap conv: a 67 MB round trip + a split kernel)
    p.split = up_f * up_t == 1 && sf == 1 && st == 1 && C2 > 0 && conv::mfma_ok(Cout, Cin) && !(C1 & 1);
    if ((up_f * up_t > 1 || C2 > 0) && !p.split) p.gxv_bytes = (long)B * p.Hv * p.Wv * Cin * (long)sizeof(float2);
    if (C2 == 0 && dcs_conv_small_dgrad_ok(Cin, Cout, kh, kw, sf, st, pad_f, pad_t, up_f, up_t)) { p.path = 1; return p; }
    if (conv::stride_ok(Cin, Cout, kh, kw, sf, st, pad_f, pad_t)) {
        // one compact sub-kernel per residue class of the input pixel instead of zero insertion
        a.up_f = 1; a.up_t = 1; a.zero_ins = 0; a.Hv = p.Hout; a.Wv = p.Wout;
        a.kh = kh; a.kw = kw; a.sf = 1; a.st = 1; a.pad_f = 0; a.pad_t = 0;
        a.Hout = p.Hv; a.Wout = p.Wv;
        conv::stride_classes(Cout, Cin, kh, kw, sf, st, pad_f, pad_t, p.Hv, p.Wv, p.cls);
        p.ncls = sf * st; p.os_f = sf; p.os_t = st;
        p.path = 2;
        return p;
    }
    a.up_f = sf; a.up_t = st; a.zero_ins = 1;
    a.kh = kh; a.kw = kw; a.sf = 1; a.st = 1; a.pad_f = kh - 1 - pad_f; a.pad_t = kw - 1 - pad_t;
    a.Hv = (p.Hout - 1) * sf + 1; a.Wv = (p.Wout - 1) * st + 1;
    a.Hout = p.Hv; a.Wout = p.Wv;                          // explicit: rows past the last tap get zeros
    p.path = conv::mfma_ok(Cout, Cin) ? 3 : 4;
    return p;
}

static long dgrad_split_bytes(const DgradPlan& p) {
    if (p.path == 0 || p.path == 2) return dcs_conv_mfma_workspace_bytes(p.a, p.ncls, p.cls);
    if (p.path == 3) return dcs_conv_mfma_workspace_bytes_plain(p.a);
    return 0;
}

static long align256(long n) { return (n + 255) / 256 * 256; }

#ifndef DCS_ACT_BF16
extern "C" long dcs_cconv2d_bwd_data_workspace_bytes(int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                                                     int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t) {
    if (B <= 0 || Hin <= 0 || Win <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || up_f < 1 || up_t < 1 || kh < 1 || kw < 1 ||
        sf < 1 || st < 1 || pad_f < 0 || pad_t < 0 || pad_f > kh - 1 || pad_t > kw - 1)
        return -1;
    const DgradPlan p = dgrad_plan(nullptr, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t);
    if (p.Hout <= 0 || p.Wout <= 0) return -1;
    return align256(p.gxv_bytes) + dgrad_split_bytes(p);     // [g_Xv | split-K slices]
}
#endif

extern "C" int DCS_SYM(dcs_cconv2d_bwd_data)(const act_t* gy, const float* wp_bwd, act_t* gx1, act_t* gx2, void* workspace,
                                    long workspace_bytes, int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                                    int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t,
                                    dcs_stream_t stream) {
    if (!gy || !wp_bwd || !gx1 || B <= 0 || Hin <= 0 || Win <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0) return DCS_ERR_BADARG;
    if ((C2 > 0) != (gx2 != nullptr)) return DCS_ERR_BADARG;
    if (kh < 1 || kw < 1 || sf < 1 || st < 1 || up_f < 1 || up_t < 1 || pad_f < 0 || pad_t < 0 || pad_f > kh - 1 ||
        pad_t > kw - 1)
        return DCS_ERR_BADARG;
    const int Cin = C1 + C2, taps = kh * kw;
    DgradPlan p = dgrad_plan(gy, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t);
    if (p.Hout <= 0 || p.Wout <= 0) return DCS_ERR_BADARG;
    hipStream_t s = dcs_stream(stream);
    const float* extra = wp_bwd + base_floats(Cin, Cout, taps);
    act_t* gxv = gx1;
    if (p.gxv_bytes > 0) {                                  // generic: gradient of the virtual input, then fold it
        if (!workspace || workspace_bytes < p.gxv_bytes) return DCS_ERR_WORKSPACE;
        gxv = (act_t*)workspace;
    }
    // whatever follows g_Xv in the workspace is split-K scratch (optional: too small just means no slicing)
    const long used = align256(p.gxv_bytes);
    void* ws2 = (workspace && workspace_bytes > used) ? (void*)((char*)workspace + used) : nullptr;
    const long ws2_bytes = ws2 ? workspace_bytes - used : 0;
    p.a.y = (act2_t*)gxv;
    int rc;
    switch (p.path) {
        case 0:
            return dcs_conv_mfma_launch_classes(p.a, extra, 1, p.cls, 1, 1, C2 ? gx2 : nullptr, 2 * C1, ws2, ws2_bytes, s);
        case 1:
            return dcs_conv_small_dgrad_launch(gy, wp_bwd, gx1, B, p.Hv, p.Wv, p.Hout, p.Wout, pad_f, pad_t, s);   // enc0
        case 2:
            rc = dcs_conv_mfma_launch_classes(p.a, extra, p.ncls, p.cls, p.os_f, p.os_t, nullptr, 0, ws2, ws2_bytes, s);
            break;
        case 3:
            p.a.wp = (const float2*)wp_bwd;
            if (p.split)
                return dcs_conv_mfma_launch_split(p.a, wp_bwd + conv::direct_floats(Cin, Cout, taps), gx2, 2 * C1, ws2, ws2_bytes, s);
            rc = dcs_conv_mfma_launch(p.a, wp_bwd + conv::direct_floats(Cin, Cout, taps), ws2, ws2_bytes, s);
            break;
        default:
            p.a.wp = (const float2*)wp_bwd;
            rc = launch_direct(p.a, s);
            break;
    }
    if (rc != DCS_OK || gxv == gx1) return rc;
    const long n = (long)B * Hin * Win * Cin;
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    DCS_LAUNCH(upsample_cat_bwd_kernel, dim3((int)nb), dim3(256), 0, s, (const act2_t*)gxv, (act2_t*)gx1,
                       (act2_t*)gx2, B, Hin, Win, C1, C2, up_f, up_t);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

#ifndef DCS_ACT_BF16
extern "C" long dcs_cconv2d_bwd_weight_workspace_bytes(int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                                                       int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t) {
    if (B <= 0 || Hin <= 0 || Win <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || kh < 1 || kw < 1 || sf < 1 || st < 1 ||
        up_f < 1 || up_t < 1)
        return -1;
    ConvArgs a = fwd_args(nullptr, nullptr, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t);
    if (!conv_geometry(a)) return -1;
    long wsz;
    int ns = wgrad_slabs(a, &wsz);
    if (dcs_conv_wgrad_mfma_ok(C1 + C2, Cout, kh, kw, C1)) {
        int th, tw;
        const int nm = dcs_conv_wgrad_mfma_slabs(a, &th, &tw);
        if (nm > ns) ns = nm;
    }
    long bytes = (long)ns * (wsz + Cout) * (long)sizeof(float2);
    if (dcs_conv_wgrad_mfma_ok(C1 + C2, Cout, kh, kw, C1) && !dcs_conv_wgrad_fold_ok(a))
        bytes = ((bytes + 255) & ~255L) + dcs_conv_wgrad_mfma_planes_bytes(a);      // (the pre-split g_Y behind the slabs)
    if (dcs_conv_wgrad_fold_ok(a)) {
        const long fb = dcs_conv_wgrad_fold_workspace_bytes(a);
        if (fb > bytes) bytes = fb;
    }
    return bytes;
}
#endif

extern "C" int DCS_SYM(dcs_cconv2d_bwd_weight)(const act_t* x1, const act_t* x2, const act_t* gy, float* gw_r, float* gw_i,
                                      float* gb_r, float* gb_i, void* workspace, long workspace_bytes, int B, int Hin,
                                      int Win, int C1, int C2, int up_f, int up_t, int Cout, int kh, int kw, int sf,
                                      int st, int pad_f, int pad_t, int transposed, dcs_stream_t stream) {
    if (!gy || !gw_r || !gw_i || !workspace) return DCS_ERR_BADARG;
    if ((gb_r == nullptr) != (gb_i == nullptr)) return DCS_ERR_BADARG;
    if (!fwd_geom_ok(x1, x2, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t)) return DCS_ERR_BADARG;
    if (kh * kw > 4 * WG_TAPS) return DCS_ERR_BADARG;
    if (dcs_conv_wgrad_mfma_ok(C1 + C2, Cout, kh, kw, C1)) {          // MFMA GEMM over the pixel axis
        ConvArgs a = fwd_args(x1, x2, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t);
        if (a.Hout <= 0 || a.Wout <= 0) return DCS_ERR_BADARG;
        if (dcs_conv_wgrad_fold_ok(a))                                 // decoder stages: per parity class, source resolution
            return dcs_conv_wgrad_fold_run(a, gy, workspace, workspace_bytes, gw_r, gw_i, gb_r, gb_i, transposed,
                                           dcs_stream(stream));
        int th, tw;
        const int ns = dcs_conv_wgrad_mfma_slabs(a, &th, &tw);
        const long wsz = (long)kh * kw * (C1 + C2) * Cout;
        if (workspace_bytes < (long)ns * (wsz + Cout) * (long)sizeof(float2)) return DCS_ERR_WORKSPACE;
        float2* slab_w = (float2*)workspace;
        float2* slab_b = slab_w + (long)ns * wsz;
        hipStream_t s = dcs_stream(stream);
        const long slabs = ((long)ns * (wsz + Cout) * (long)sizeof(float2) + 255) & ~255L, pb = dcs_conv_wgrad_mfma_planes_bytes(a);
        void* planes = (pb > 0 && workspace_bytes >= slabs + pb) ? (char*)workspace + slabs : nullptr;
        const int rc = dcs_conv_wgrad_mfma_launch(a, gy, slab_w, (float*)slab_b, ns, s, planes);
        if (rc != DCS_OK) return rc;
        return launch_wgrad_reduce(slab_w, slab_b, ns, gw_r, gw_i, gb_r, gb_i, Cout, C1 + C2, kh, kw, transposed, s);
    }
    WgradArgs w{};
    w.c = fwd_args(x1, x2, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t);
    if (!conv_geometry(w.c)) return DCS_ERR_BADARG;
    long wsz;
    w.n_slabs = wgrad_slabs(w.c, &wsz);
    if (workspace_bytes < (long)w.n_slabs * (wsz + Cout) * (long)sizeof(float2)) return DCS_ERR_WORKSPACE;
    w.x1 = (const act2_t*)x1; w.x2 = (const act2_t*)x2; w.gy = (const act2_t*)gy;
    w.slab_w = (float2*)workspace;
    w.slab_b = w.slab_w + (long)w.n_slabs * wsz;
    w.total_tiles = w.c.tiles_w * w.c.tiles_h * B;
    const int Cin = C1 + C2;
    w.c.x1 = (const act2_t*)x1;
    if (dcs_conv_enc0_wgrad_ok(w.c)) {           // 7x7 1->8 stride 2: taps x pixels on the MFMA units (conv_enc0.hip)
        hipStream_t s = dcs_stream(stream);
        int ns = 0;
        const int rc = dcs_conv_enc0_wgrad_launch(w.c, gy, w.slab_w, w.slab_b, w.n_slabs, &ns, s);
        if (rc != DCS_OK) return rc;
        return launch_wgrad_reduce(w.slab_w, w.slab_b, ns, gw_r, gw_i, gb_r, gb_i, Cout, Cin, kh, kw, transposed, s);
    }
#ifndef DCS_ACT_BF16                             // (fp32 maps only: the attention convs' operands stay fp32 in either mode)
    if (dcs_conv_wgrad_small_ok(w.c)) {          // 7x7 2->1 / 1->8: pixel-stationary kernel (conv_wgrad_small.hip)
        hipStream_t s = dcs_stream(stream);
        // two resident workgroups per CU.  The 2 -> 1 attention convs run batched (13 problems in one launch inside a
        // train step): ~8 tiles per workgroup there, so that the fixed cross-lane reduction of the 196 accumulators at the
        // end of a workgroup (~2.5 us) is paid per 8 tiles (~6 us of FMAs) rather than per 2
        int ns = w.n_slabs < 512 ? w.n_slabs : 512;
        if (Cin == 2 && Cout == 1) { ns = w.total_tiles / 8; ns = ns < 1 ? 1 : (ns > 512 ? 512 : ns); }   // = ~2 of its 16 x 64 tiles
        const int rc = dcs_conv_wgrad_small_launch(w.c, gy, w.slab_w, w.slab_b, ns, s);
        if (rc != DCS_OK) return rc;
        return launch_wgrad_reduce(w.slab_w, w.slab_b, ns, gw_r, gw_i, gb_r, gb_i, Cout, Cin, kh, kw, transposed, s);
    }
#endif
    const int n_ci = (Cin + CHUNK - 1) / CHUNK;
    w.n_co_chunks = (Cout + WG_CO - 1) / WG_CO;
    const size_t lds = ((size_t)CHUNK * w.c.plane + (size_t)TH * TW * WG_CO) * sizeof(float2);
    if (lds > 150 * 1024) return DCS_ERR_BADARG;
    hipStream_t s = dcs_stream(stream);
    // outputs per thread: taps * min(Cin,8) * min(Cout,8) spread over 256 threads
    const int n_out = kh * kw * (Cin < CHUNK ? Cin : CHUNK) * (Cout < WG_CO ? Cout : WG_CO);
    const int slots = (n_out + TH * TW - 1) / (TH * TW);
    auto fn = slots <= 1 ? cconv_wgrad_kernel<1> : slots <= 2 ? cconv_wgrad_kernel<2>
            : slots <= 4 ? cconv_wgrad_kernel<4> : cconv_wgrad_kernel<WG_TAPS>;
    if (dcs_ensure_dynamic_lds((const void*)fn, lds) != hipSuccess) return DCS_ERR_LAUNCH;
    dim3 grid(w.n_slabs, n_ci * w.n_co_chunks);
    if (grid.y > 65535) return DCS_ERR_BADARG;
    DCS_LAUNCH(fn, grid, dim3(TH * TW), lds, s, w);
    DCS_CHECK_LAUNCH();
    return launch_wgrad_reduce(w.slab_w, w.slab_b, w.n_slabs, gw_r, gw_i, gb_r, gb_i, Cout, Cin, kh, kw, transposed, s);
}

#ifndef DCS_ACT_BF16
extern "C" int dcs_upsample_cat_bwd(const float* gxv, float* gx1, float* gx2, int B, int Hin, int Win, int C1, int C2,
                                    int up_f, int up_t, dcs_stream_t stream) {
    if (!gxv || !gx1 || B <= 0 || Hin <= 0 || Win <= 0 || C1 <= 0 || C2 < 0 || up_f < 1 || up_t < 1) return DCS_ERR_BADARG;
    if ((C2 > 0) != (gx2 != nullptr)) return DCS_ERR_BADARG;
    const long n = (long)B * Hin * Win * (C1 + C2);
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    DCS_LAUNCH(upsample_cat_bwd_kernel, dim3((int)nb), dim3(256), 0, dcs_stream(stream), (const float2*)gxv,
                       (float2*)gx1, (float2*)gx2, B, Hin, Win, C1, C2, up_f, up_t);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
#endif
