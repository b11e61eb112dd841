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
    const act2_t* gy; act2_t* gx;
    int B, Hg, Wg, Hx, Wx, tiles_w, tiles_h;               // g_Y extent, g_X extent, class-space tiling (of the largest class)
    int kh[4], kw[4], fy[4], fx[4], py[4], px[4], Hc[4], Wc[4];   // per class (ry*2 + rx): sub-kernel, first full tap, padding, extent
};

// Round 5: a thread owns TWO vertically adjacent class pixels (a workgroup of 256 threads a 16 x 32 tile).  One thread per pixel read
// a 16-byte patch element AND a 16-byte weight element from LDS per two complex MACs — 3.3 GB of LDS traffic per launch at
// [32,256,256] (52.5 us).  Now
//   * the patch lies in LDS as four channel-pair planes [q][row][col] of float4; a thread reads the DP + KH - 1 rows its
//     windows span once per (tap column, plane), lanes on adjacent columns;
//   * a complex MAC is two packed FMAs (P += w.x x, Q += w.y x; the weights lie in LDS as {w, w} pairs);
//   * all of a thread's patch loads are issued before the first LDS write.
// 52.5 -> 46 us.  Measured on the way (profiles/r05_enc0_dgrad.txt): four pixels per thread at 128 threads (42 KB of LDS per
// 128 threads: 1.5 waves per SIMD) 63-68 us; the weights as SGPR operands through the scalar cache 68-87 us (scalar loads and LDS
// reads share lgkmcnt: every step drained both).  Counters of this form: LDS busy 22 us, VALU busy 24 us of the 46 — 886 vector
// instructions per wave of which 370 are FMAs (the rest: the patch's index arithmetic and stores).
constexpr int DTH = 16, DTW = 32, DP = 2, DROWS = DTH + 3, DCOLS = DTW + 3;
template <int KH, int KW>
__device__ __forceinline__ void dgrad_class(const DgArgs& d, const float4* __restrict__ wsm, const float4* __restrict__ planes, int cls,
                                            int b, int cy0, int cx0) {
    const int t = threadIdx.x, tx = t % DTW, tg = t / DTW;
    // two accumulators per pixel, P += w.x * x and Q += w.y * x (packed FMAs on {re, im}; the weights lie in LDS as {w.x, w.x} /
    // {w.y, w.y} pairs, so no cross-half operand selection: dcs_common.h), combined at the end
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f accp[DP], accq[DP];
#pragma unroll
    for (int p = 0; p < DP; ++p) { accp[p] = v2f{0.f, 0.f}; accq[p] = v2f{0.f, 0.f}; }
#pragma unroll 1
    for (int q = 0; q < DC / 2; ++q)                          // (rolled: fully unrolled the scheduler hoisted all 112 patch reads — 444 VGPRs)
#pragma unroll
        for (int jx = 0; jx < KW; ++jx) {
            float4 xv[DP + KH - 1];
#pragma unroll
            for (int r = 0; r < DP + KH - 1; ++r) xv[r] = planes[(q * DROWS + DP * tg + r) * DCOLS + tx + jx];
#pragma unroll
            for (int jy = 0; jy < KH; ++jy) {
                const float4 wa = wsm[((jy * KW + jx) * (DC / 2) + q) * 2], wb = wsm[((jy * KW + jx) * (DC / 2) + q) * 2 + 1];
                const v2f w0x = v2f{wa.x, wa.y}, w0y = v2f{wa.z, wa.w}, w1x = v2f{wb.x, wb.y}, w1y = v2f{wb.z, wb.w};
#pragma unroll
                for (int p = 0; p < DP; ++p) {
                    const float4 x = xv[p + jy];
                    const v2f x0 = v2f{x.x, x.y}, x1 = v2f{x.z, x.w};
                    accp[p] = __builtin_elementwise_fma(w0x, x0, accp[p]);
                    accq[p] = __builtin_elementwise_fma(w0y, x0, accq[p]);
                    accp[p] = __builtin_elementwise_fma(w1x, x1, accp[p]);
                    accq[p] = __builtin_elementwise_fma(w1y, x1, accq[p]);
                }
            }
        }
    float ar[DP], ai[DP];
#pragma unroll
    for (int p = 0; p < DP; ++p) {
        float px = accp[p].x, py = accp[p].y, qx = accq[p].x, qy = accq[p].y;
        asm volatile("" : "+v"(px), "+v"(py), "+v"(qx), "+v"(qy));        // scalar combine: as a vector expression it became a v_pk_add_f32
        ar[p] = px - qy; ai[p] = py + qx;                                  // with a cross-half operand selection (dcs_common.h)
    }
    const int cx = cx0 + tx;
#pragma unroll
    for (int p = 0; p < DP; ++p) {
        const int cy = cy0 + DP * tg + p;
        if (cy < d.Hc[cls] && cx < d.Wc[cls])
            conv::stc(d.gx + ((long)b * d.Hx + 2 * cy + (cls >> 1)) * d.Wx + 2 * cx + (cls & 1), make_float2(ar[p], ai[p]));
    }
}

// wpb: [49][8] flipped / conjugated kernel (complex)
__global__ __launch_bounds__(DTH * DTW / DP) void cconv_small_dgrad_s2_kernel(DgArgs d, const float2* __restrict__ wpb) {
    DCS_PRIO_CRITICAL();
    __shared__ __attribute__((aligned(16))) float4 planes[(DC / 2) * DROWS * DCOLS];
    __shared__ __attribute__((aligned(16))) float4 wsm[16 * (DC / 2) * 2];
    const int cls = blockIdx.y, b = blockIdx.z;
    const int cy0 = (blockIdx.x / d.tiles_w) * DTH, cx0 = (blockIdx.x % d.tiles_w) * DTW;
    if (cy0 >= d.Hc[cls] || cx0 >= d.Wc[cls]) return;
    const int kh = d.kh[cls], kw = d.kw[cls];
    const int rows = DTH + kh - 1, cols = DTW + kw - 1;
    const int gy0 = cy0 - d.py[cls], gx0 = cx0 - d.px[cls];
    // every load of the thread first (clamped address, value selected afterwards), then the LDS writes: as a load -> write loop the
    // 21 elements of a thread went out one memory round trip after the other (~20 us per workgroup)
    constexpr int NT = DTH * DTW / DP, NE = DROWS * DCOLS * (DC / 2), NL = (NE + NT - 1) / NT;
    float4 pv[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int idx = threadIdx.x + k * NT;
        const int q = idx % (DC / 2), p = idx / (DC / 2);
        const int ix = p % DCOLS, iy = p / DCOLS;
        const int y = gy0 + iy, x = gx0 + ix;
        const bool in = iy < rows && ix < cols && y >= 0 && y < d.Hg && x >= 0 && x < d.Wg;
        const int yc = y < 0 ? 0 : (y >= d.Hg ? d.Hg - 1 : y), xc = x < 0 ? 0 : (x >= d.Wg ? d.Wg - 1 : x);
        const float4 v = dcs_ld4(reinterpret_cast<const act_t*>(d.gy + (((long)b * d.Hg + yc) * d.Wg + xc) * DC + 2 * q));
        pv[k] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int idx = threadIdx.x + k * NT;
        const int q = idx % (DC / 2), p = idx / (DC / 2);
        if (idx < NE) planes[(q * DROWS + p / DCOLS) * DCOLS + p % DCOLS] = pv[k];
    }
    if (threadIdx.x < kh * kw * (DC / 2)) {                            // this class's taps of the flipped 7x7 kernel (<= 64 x 16 bytes)
        const int tap = threadIdx.x / (DC / 2), q = threadIdx.x % (DC / 2), jy = tap / kw, jx = tap % kw;
        const float4 w = *reinterpret_cast<const float4*>(wpb + ((d.fy[cls] + 2 * jy) * 7 + d.fx[cls] + 2 * jx) * DC + 2 * q);
        wsm[2 * threadIdx.x] = make_float4(w.x, w.x, w.y, w.y);
        wsm[2 * threadIdx.x + 1] = make_float4(w.z, w.z, w.w, w.w);
    }
    __syncthreads();
    if (kh == 4 && kw == 4) dgrad_class<4, 4>(d, wsm, planes, cls, b, cy0, cx0);
    else if (kh == 4) dgrad_class<4, 3>(d, wsm, planes, cls, b, cy0, cx0);
    else if (kw == 4) dgrad_class<3, 4>(d, wsm, planes, cls, b, cy0, cx0);
    else dgrad_class<3, 3>(d, wsm, planes, cls, b, cy0, cx0);
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
    d.gy = (const act2_t*)gy; d.gx = (act2_t*)gx;
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
    d.tiles_w = (Wc + DTW - 1) / DTW; d.tiles_h = (Hc + DTH - 1) / DTH;
    dim3 grid(d.tiles_w * d.tiles_h, 4, B);
    if (grid.z > 65535) return DCS_ERR_BADARG;
    DCS_LAUNCH(cconv_small_dgrad_s2_kernel, grid, dim3(DTH * DTW / DP), 0, stream, d, (const float2*)wp_bwd);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
