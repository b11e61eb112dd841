// conv_small.hip — data gradient of the first encoder conv (1 -> 8 channels, 7x7, stride (2,2): c_network.py:107-112,
// config.py:83-91) as 8 -> 1 over g_Y with one compact sub-kernel per output-parity class.
//
// The generic path (conv_direct.hip) runs the forward kernel over zero-inserted g_Y: 3/4 of its 49 x 8 complex MACs
// per pixel multiply inserted zeros, and its run-time tap loop issues one scalar weight load per tap.  Here every
// input pixel belongs to one of four parity classes whose sub-kernel (4x4, 4x3, 3x4 or 3x3 taps of the flipped
// kernel, picked by index: no extra weight panel) is fully unrolled: 294 -> 43 us at batch 32 x 256 x 256.
// (Measured and dropped: the same compile-time treatment of the 2->1 / 1->2 / 1->8 FORWARD kernels — the rolled
// generic loop with its small register footprint was as fast or faster.)
#include "conv_common.h"

namespace {

constexpr int TH = 16, TW = 16;

// ---- data gradient of the 1 -> 8, 7x7, stride-(2,2) convolution ----------------------------------------------
constexpr int DC = 8;                 // forward output channels = channels of g_Y
constexpr int DPIX = 10;              // LDS float2 per patch pixel: 8 channels + 2 pad (80-B pitch keeps b128 reads spread)

struct DgArgs {
    const act2_t* gy; const float2* wpb; act2_t* gx;      // wpb: [49][8][1] flipped / conjugated kernel (act2_t: dcs_common.h)
    int B, Hg, Wg, Hx, Wx, tiles_w, tiles_h;               // g_Y extent, g_X extent, class-space tiling (of the largest class)
    int kh[4], kw[4], fy[4], fx[4], py[4], px[4], Hc[4], Wc[4];   // per class (ry*2 + rx): sub-kernel, first full tap, padding, extent
};

// wsm: the class's sub-kernel, [tap jy * KW + jx][channel pair] as {w0.x, w0.y, w1.x, w1.y}, staged in LDS by the caller.  Read
// straight from d.wpb (a pointer inside a by-value argument struct: no noalias / readonly information) the wave-uniform weights
// were VECTOR loads — 196 global_load_dwordx4 per thread in the 4x4 class, 64 lanes x 16 bytes through the L1 for 16 useful bytes:
// more L1 cycles than the kernel has FMA cycles.  From LDS they are broadcast reads.
template <int KH, int KW>
__device__ __forceinline__ void dgrad_class(const DgArgs& d, const float2* patch, const float4* wsm, int cls, int b, int cy0, int cx0) {
    constexpr int COLS = TW + KW - 1;
    const int t = threadIdx.x, tx = t % TW, ty = t / TW;
    float ar = 0.f, ai = 0.f;
    const float2* base = patch + (ty * COLS + tx) * DPIX;
#pragma unroll
    for (int jy = 0; jy < KH; ++jy)
#pragma unroll
        for (int jx = 0; jx < KW; ++jx) {
            const float4* xp = reinterpret_cast<const float4*>(base + (jy * COLS + jx) * DPIX);
#pragma unroll
            for (int q = 0; q < DC / 2; ++q) {
                const float4 xv = xp[q];
                const float4 wq = wsm[(jy * KW + jx) * (DC / 2) + q];
                const float2 w0 = make_float2(wq.x, wq.y), w1 = make_float2(wq.z, wq.w);
                ar = fmaf(w0.x, xv.x, fmaf(-w0.y, xv.y, ar));
                ai = fmaf(w0.x, xv.y, fmaf(w0.y, xv.x, ai));
                ar = fmaf(w1.x, xv.z, fmaf(-w1.y, xv.w, ar));
                ai = fmaf(w1.x, xv.w, fmaf(w1.y, xv.z, ai));
            }
        }
    const int cy = cy0 + ty, cx = cx0 + tx;
    if (cy < d.Hc[cls] && cx < d.Wc[cls])
        conv::stc(d.gx + ((long)b * d.Hx + 2 * cy + (cls >> 1)) * d.Wx + 2 * cx + (cls & 1), make_float2(ar, ai));
}

__global__ __launch_bounds__(TH * TW) void cconv_small_dgrad_s2_kernel(DgArgs d) {
    DCS_PRIO_CRITICAL();
    __shared__ __attribute__((aligned(16))) float2 patch[(TH + 3) * (TW + 3) * DPIX];
    __shared__ __attribute__((aligned(16))) float4 wsm[16 * (DC / 2)];
    const int cls = blockIdx.y, b = blockIdx.z;
    const int cy0 = (blockIdx.x / d.tiles_w) * TH, cx0 = (blockIdx.x % d.tiles_w) * TW;
    if (cy0 >= d.Hc[cls] || cx0 >= d.Wc[cls]) return;
    const int kh = d.kh[cls], kw = d.kw[cls];
    const int rows = TH + kh - 1, cols = TW + kw - 1;
    const int gy0 = cy0 - d.py[cls], gx0 = cx0 - d.px[cls];
    for (int idx = threadIdx.x; idx < rows * cols * (DC / 2); idx += TH * TW) {
        const int q = idx % (DC / 2), p = idx / (DC / 2);
        const int ix = p % cols, iy = p / cols;
        const int y = gy0 + iy, x = gx0 + ix;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < d.Hg && x >= 0 && x < d.Wg)
            v = dcs_ld4(reinterpret_cast<const act_t*>(d.gy + (((long)b * d.Hg + y) * d.Wg + x) * DC + 2 * q));
        *reinterpret_cast<float4*>(patch + p * DPIX + 2 * q) = v;
    }
    if (threadIdx.x < kh * kw * (DC / 2)) {                            // this class's taps of the flipped 7x7 kernel
        const int tap = threadIdx.x / (DC / 2), q = threadIdx.x % (DC / 2), jy = tap / kw, jx = tap % kw;
        wsm[threadIdx.x] = *reinterpret_cast<const float4*>(d.wpb + ((d.fy[cls] + 2 * jy) * 7 + d.fx[cls] + 2 * jx) * DC + 2 * q);
    }
    __syncthreads();
    if (kh == 4 && kw == 4) dgrad_class<4, 4>(d, patch, wsm, cls, b, cy0, cx0);
    else if (kh == 4) dgrad_class<4, 3>(d, patch, wsm, cls, b, cy0, cx0);
    else if (kw == 4) dgrad_class<3, 4>(d, patch, wsm, cls, b, cy0, cx0);
    else dgrad_class<3, 3>(d, patch, wsm, cls, b, cy0, cx0);
}

}  // namespace

// forward geometry: 1 -> 8 channels, 7x7, stride (2,2), padding (pad_f, pad_t), input Hx x Wx, output Hg x Wg
bool dcs_conv_small_dgrad_ok(int Cin, int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int up_f, int up_t) {
    if (!(Cin == 1 && Cout == DC && kh == 7 && kw == 7 && sf == 2 && st == 2 && up_f == 1 && up_t == 1)) return false;
    for (int r = 0; r < 2; ++r) {
        const conv::Axis ay = conv::stride_axis(7, 2, pad_f, r, 1 << 20), ax = conv::stride_axis(7, 2, pad_t, r, 1 << 20);
        if (ay.count < 3 || ay.count > 4 || ax.count < 3 || ax.count > 4) return false;
    }
    return true;
}

int dcs_conv_small_dgrad_launch(const act_t* gy, const float* wp_bwd, act_t* gx, int B, int Hx, int Wx, int Hg, int Wg,
                                int pad_f, int pad_t, hipStream_t stream) {
    DgArgs d;
    d.gy = (const act2_t*)gy; d.wpb = (const float2*)wp_bwd; d.gx = (act2_t*)gx;
    d.B = B; d.Hg = Hg; d.Wg = Wg; d.Hx = Hx; d.Wx = Wx;
    int Hc = 0, Wc = 0;
    for (int ry = 0; ry < 2; ++ry)
        for (int rx = 0; rx < 2; ++rx) {
            const conv::Axis ay = conv::stride_axis(7, 2, pad_f, ry, Hx), ax = conv::stride_axis(7, 2, pad_t, rx, Wx);
            const int c = ry * 2 + rx;
            d.kh[c] = ay.count; d.kw[c] = ax.count; d.fy[c] = ay.first; d.fx[c] = ax.first;
            d.py[c] = ay.pad; d.px[c] = ax.pad; d.Hc[c] = ay.n; d.Wc[c] = ax.n;
            Hc = ay.n > Hc ? ay.n : Hc; Wc = ax.n > Wc ? ax.n : Wc;
        }
    if (Hc <= 0 || Wc <= 0) return DCS_ERR_BADARG;
    d.tiles_w = (Wc + TW - 1) / TW; d.tiles_h = (Hc + TH - 1) / TH;
    dim3 grid(d.tiles_w * d.tiles_h, 4, B);
    if (grid.z > 65535) return DCS_ERR_BADARG;
    DCS_LAUNCH(cconv_small_dgrad_s2_kernel, grid, dim3(TH * TW), 0, stream, d);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
