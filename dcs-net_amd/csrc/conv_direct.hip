// conv_direct.hip — direct (non-GEMM) complex correlation for the small-channel ends of the
// network, and the weight packer.
//
// Used for enc0 (Cin=1, K=98: SURVEY.md §7 "small-channel ends"), dec6 (Cout=1), the 7x7 2->1
// spatial-attention conv (c_network.py:74) and as the generic fallback for any geometry the
// MFMA implicit-GEMM kernel (conv_mfma.hip) does not take.  These stages are HBM-bound
// (SURVEY.md §8d): the kernel stages one haloed input tile per workgroup in LDS (each input
// element is read from HBM once per tile), keeps COB output channels per thread in VGPRs and
// takes its weights through the scalar cache (wave-uniform addresses).
//
// The virtual input (nearest upsample of cat(x1, x2): c_network.py:214-216) is resolved in the
// LDS gather, so neither the concatenated nor the upsampled tensor ever exists in HBM.
#include "dcs_common.h"

namespace {

constexpr int TH = 16, TW = 16;      // output tile (pixels) per workgroup
constexpr int CHUNK = 8;             // input channels staged per LDS pass

struct ConvArgs {
    const float2* x1; const float2* x2; const float2* wp; const float2* bias; float2* y;
    int B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t, act;
    int Hout, Wout, tiles_w, rows, cols, colsp, plane;
};

template <int COB>
__global__ __launch_bounds__(TH * TW) void cconv_direct_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];   // [CHUNK][rows][colsp]
    const int t = threadIdx.x;
    const int tx = t % TW, ty = t / TW;
    const int tile_id = blockIdx.x;
    const int oy0 = (tile_id / a.tiles_w) * TH, ox0 = (tile_id % a.tiles_w) * TW;
    const int co0 = blockIdx.y * COB;
    const int b = blockIdx.z;
    const int Cin = a.C1 + a.C2;
    const int Hv = a.Hin * a.up_f, Wv = a.Win * a.up_t;       // virtual (upsampled) input extent
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
            const int vy = vy0 + iy, vx = vx0 + ix;
            float2 v = make_float2(0.f, 0.f);
            if (vy >= 0 && vy < Hv && vx >= 0 && vx < Wv) {
                const long sp = ((long)b * a.Hin + vy / a.up_f) * a.Win + vx / a.up_t;
                const int c = c0 + ci;
                v = (c < a.C1) ? a.x1[sp * a.C1 + c] : a.x2[sp * a.C2 + (c - a.C1)];
            }
            tile[ci * a.plane + iy * a.colsp + ix] = v;
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
        float2* out = a.y + (((long)b * a.Hout + oy) * a.Wout + ox) * a.Cout + co0;
#pragma unroll
        for (int i = 0; i < COB; ++i) {
            const float2 bv = a.bias[co0 + i];
            out[i] = make_float2(dcs_act(accr[i] + bv.x, a.act), dcs_act(acci[i] + bv.y, a.act));
        }
    }
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ w_r, const float* __restrict__ w_i,
                                        const float* __restrict__ b_r, const float* __restrict__ b_i,
                                        float2* __restrict__ wp, float2* __restrict__ bias_out, int Cout, int Cin,
                                        int kh, int kw, int transposed) {
    const long n = (long)kh * kw * Cin * Cout;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Cout) {
        const float br = b_r ? b_r[i] : 0.f, bi = b_i ? b_i[i] : 0.f;
        bias_out[i] = make_float2(br - bi, br + bi);
    }
    if (i >= n) return;
    const int co = (int)(i % Cout);
    const int ci = (int)((i / Cout) % Cin);
    const int tap = (int)(i / ((long)Cout * Cin));
    const int dy = tap / kw, dx = tap % kw;
    long src;
    if (transposed)   // ConvTranspose2d weight [Cin][Cout][kh][kw], flipped
        src = (((long)ci * Cout + co) * kh + (kh - 1 - dy)) * kw + (kw - 1 - dx);
    else              // Conv2d weight [Cout][Cin][kh][kw]
        src = (((long)co * Cin + ci) * kh + dy) * kw + dx;
    wp[i] = make_float2(w_r[src], w_i[src]);
}

}  // namespace

extern "C" int dcs_pack_conv_weight(const float* w_r, const float* w_i, const float* b_r, const float* b_i, float* wp,
                                    float* bias_out, int Cout, int Cin, int kh, int kw, int transposed,
                                    dcs_stream_t stream) {
    if (!w_r || !w_i || !wp || !bias_out || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0) return DCS_ERR_BADARG;
    if ((b_r == nullptr) != (b_i == nullptr)) return DCS_ERR_BADARG;
    long n = (long)kh * kw * Cin * Cout;
    if (n < Cout) n = Cout;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(dcs_cdiv(n, 256)), dim3(256), 0, dcs_stream(stream), w_r, w_i,
                       b_r, b_i, (float2*)wp, (float2*)bias_out, Cout, Cin, kh, kw, transposed);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// shared with conv_mfma.hip's dispatcher
int dcs_cconv2d_direct(const float* x1, const float* x2, const float* wp, const float* bias, float* y, int B, int Hin,
                       int Win, int C1, int C2, int up_f, int up_t, int Cout, int kh, int kw, int sf, int st, int pad_f,
                       int pad_t, int act, hipStream_t stream) {
    ConvArgs a;
    a.x1 = (const float2*)x1; a.x2 = (const float2*)x2; a.wp = (const float2*)wp; a.bias = (const float2*)bias;
    a.y = (float2*)y;
    a.B = B; a.Hin = Hin; a.Win = Win; a.C1 = C1; a.C2 = C2; a.up_f = up_f; a.up_t = up_t; a.Cout = Cout;
    a.kh = kh; a.kw = kw; a.sf = sf; a.st = st; a.pad_f = pad_f; a.pad_t = pad_t; a.act = act;
    a.Hout = (Hin * up_f + 2 * pad_f - kh) / sf + 1;
    a.Wout = (Win * up_t + 2 * pad_t - kw) / st + 1;
    if (a.Hout <= 0 || a.Wout <= 0) return DCS_ERR_BADARG;
    a.tiles_w = (a.Wout + TW - 1) / TW;
    const int tiles_h = (a.Hout + TH - 1) / TH;
    a.rows = (TH - 1) * sf + kh;
    a.cols = (TW - 1) * st + kw;
    a.colsp = a.cols | 1;                       // odd row pitch (in float2) spreads LDS banks
    a.plane = a.rows * a.colsp + 1;
    const int Cin = C1 + C2;
    const size_t lds = (size_t)(Cin < CHUNK ? Cin : CHUNK) * a.plane * sizeof(float2);
    if (lds > 150 * 1024) return DCS_ERR_BADARG;
    int cob = (Cout % 8 == 0) ? 8 : (Cout % 4 == 0) ? 4 : (Cout % 2 == 0) ? 2 : 1;
    if (lds > 64 * 1024) {   // above the default dynamic-LDS limit: raise it for this instantiation
        const void* fn = cob == 8 ? (const void*)cconv_direct_kernel<8> : cob == 4 ? (const void*)cconv_direct_kernel<4>
                       : cob == 2 ? (const void*)cconv_direct_kernel<2> : (const void*)cconv_direct_kernel<1>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return DCS_ERR_LAUNCH;
    }
    dim3 grid(a.tiles_w * tiles_h, Cout / cob, B);
    if (grid.y > 65535 || grid.z > 65535) return DCS_ERR_BADARG;
    switch (cob) {
        case 8: hipLaunchKernelGGL(cconv_direct_kernel<8>, grid, dim3(TH * TW), lds, stream, a); break;
        case 4: hipLaunchKernelGGL(cconv_direct_kernel<4>, grid, dim3(TH * TW), lds, stream, a); break;
        case 2: hipLaunchKernelGGL(cconv_direct_kernel<2>, grid, dim3(TH * TW), lds, stream, a); break;
        default: hipLaunchKernelGGL(cconv_direct_kernel<1>, grid, dim3(TH * TW), lds, stream, a); break;
    }
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_cconv2d_fwd(const float* x1, const float* x2, const float* wp, const float* bias, float* y, int B,
                               int Hin, int Win, int C1, int C2, int up_f, int up_t, int Cout, int kh, int kw, int sf,
                               int st, int pad_f, int pad_t, int act, dcs_stream_t stream) {
    if (!x1 || !wp || !bias || !y) return DCS_ERR_BADARG;
    if (B <= 0 || Hin <= 0 || Win <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0) return DCS_ERR_BADARG;
    if ((C2 > 0) != (x2 != nullptr)) return DCS_ERR_BADARG;
    if (up_f < 1 || up_t < 1 || kh < 1 || kw < 1 || sf < 1 || st < 1 || pad_f < 0 || pad_t < 0) return DCS_ERR_BADARG;
    if (act < DCS_ACT_NONE || act > DCS_ACT_SIGMOID) return DCS_ERR_BADARG;
    return dcs_cconv2d_direct(x1, x2, wp, bias, y, B, Hin, Win, C1, C2, up_f, up_t, Cout, kh, kw, sf, st, pad_f, pad_t,
                              act, dcs_stream(stream));
}
