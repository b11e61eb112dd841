// conv_up1.hip — the last decoder stage's forward pass in one kernel: a 3x3, stride-1 complex conv with ONE output channel
// over the 2x2 nearest upsample of cat(x1, x2) (ComplexConvTranspose2d 16 -> 1 behind cat + upsample: c_network.py:135-141,
// :214-217; config.py:84-106).
//
// The training path factors this stage as a 1x1 MFMA conv into 9 "tap channels" at source resolution plus a tap-sum gather
// (elementwise.hip) because its backward wants the tap channels anyway; going forward that costs a 67 MB intermediate
// (write + read) for a 17 MB result.  Here a workgroup stages a haloed 10 x 34 source tile (16 channels, channel-major
// planes) in LDS once and produces the 16 x 64 outputs above it directly:
//   * upsample folded: an output pixel of parity (py, px) reads a 2 x 2 source neighbourhood with tap-summed weights
//     (4 x 16 complex MACs instead of 9 x 16); the 4 x 2 x 2 x 16 folded weights are formed in LDS from the tap-rows
//     panel at the start of the workgroup;
//   * a wave owns ONE parity class (its weights are broadcast reads), a thread 4 consecutive pixels of one class row:
//     per (channel, source row) it loads 5 inputs for 8 MACs;
//   * complex MAC = two v_pk_fma_f32.
// HBM traffic = the two sources once + the result.
#include "conv_common.h"

namespace {

constexpr int CIN = 16, SRT = 8, SCT = 32, HR = SRT + 2, HC = SCT + 2, HCP = HC + 3;   // source tile, halo; row pitch 37 complex: the 8 x 8 lanes of a half-wave hit distinct banks
typedef float v2f __attribute__((ext_vector_type(2)));

struct Up1Args {
    const void* x1; const void* x2; const float2* wt;           // wt: tap-rows panel complex[CIN][ct], tap = dy*3 + dx
    const float* b_r; const float* b_i; float2* y;
    int Hs, Ws, C1, C2, ct, tiles_w, tiles;
};

// IT: element type of the two sources — float, or bf16 (unsigned short) where the activations live in bf16
// (dcs_cconv_up2_single_fwd_h); the result (the network's fp32 mask) and the arithmetic are fp32 either way.
template <typename IT>
__global__ __launch_bounds__(256) void cconv_up1_kernel(Up1Args p) {
    __shared__ __attribute__((aligned(16))) float2 tile[CIN][HR * HCP];
    __shared__ __attribute__((aligned(16))) float4 wf[4][2][2][CIN];     // [parity class][a][b][ci], {w.x, w.y, w.y, w.x}: both broadcasts read a LOW half
    const int t = threadIdx.x, b = blockIdx.y;
    const int m0 = ((int)blockIdx.x / p.tiles_w) * SRT, n0 = ((int)blockIdx.x % p.tiles_w) * SCT;

    {   // folded weights: rows/cols of the 3x3 kernel that land on the same source pixel are summed
        const int ci = t & 15, bb = (t >> 4) & 1, aa = (t >> 5) & 1, cls = t >> 6;
        const int py = cls >> 1, px = cls & 1;
        // parity 0: a = 0 <- {0}, a = 1 <- {1, 2};  parity 1: a = 0 <- {0, 1}, a = 1 <- {2}
        const int dy_lo = py == 0 ? (aa == 0 ? 0 : 1) : (aa == 0 ? 0 : 2), dy_hi = py == 0 ? (aa == 0 ? 0 : 2) : (aa == 0 ? 1 : 2);
        const int dx_lo = px == 0 ? (bb == 0 ? 0 : 1) : (bb == 0 ? 0 : 2), dx_hi = px == 0 ? (bb == 0 ? 0 : 2) : (bb == 0 ? 1 : 2);
        float2 s = make_float2(0.f, 0.f);
        for (int dy = dy_lo; dy <= dy_hi; ++dy)
            for (int dx = dx_lo; dx <= dx_hi; ++dx) {
                const float2 w = p.wt[ci * p.ct + dy * 3 + dx];
                s.x += w.x; s.y += w.y;
            }
        wf[cls][aa][bb][ci] = make_float4(s.x, s.y, s.y, s.x);
    }
    // haloed source tile, channel-major planes; one float4 (2 channels) per load.  All of a thread's loads are issued
    // before the first LDS write (a load -> wait -> write loop pays one memory round trip per element: 11 per tile)
    const long img = (long)b * p.Hs * p.Ws;
    constexpr int NSLOT = HR * HC * (CIN / 2), NL = (NSLOT + 255) / 256;
    float4 sv[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = t + 256 * k;
        const int q = i % (CIN / 2), px_ = i / (CIN / 2);
        const int hc = px_ % HC, hr = px_ / HC;
        const int sy = m0 - 1 + hr, sx = n0 - 1 + hc;
        // always a load (clamped pixel), zeroed afterwards: predicated, each load sat in its own branch and the eleven of a
        // thread went out one memory round trip after the other
        const bool in = i < NSLOT && sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws;
        const int syc = sy < 0 ? 0 : (sy >= p.Hs ? p.Hs - 1 : sy), sxc = sx < 0 ? 0 : (sx >= p.Ws ? p.Ws - 1 : sx);
        const long sp = img + (long)syc * p.Ws + sxc;
        const int c = 2 * q;
        const IT* src = c < p.C1 ? (const IT*)p.x1 + (sp * p.C1 + c) * 2 : (const IT*)p.x2 + (sp * p.C2 + (c - p.C1)) * 2;
        const float4 v = dcs_ld4(src);
        sv[k] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = t + 256 * k;
        if (i >= NSLOT) continue;
        const int q = i % (CIN / 2), px_ = i / (CIN / 2);
        const int hc = px_ % HC, hr = px_ / HC;
        tile[2 * q][hr * HCP + hc] = make_float2(sv[k].x, sv[k].y);
        tile[2 * q + 1][hr * HCP + hc] = make_float2(sv[k].z, sv[k].w);
    }
    __syncthreads();

    const int cls = t >> 6, py = cls >> 1, px = cls & 1;
    const int u = t & 63, m = u >> 3, nq = (u & 7) * 4;
    // two accumulators per output, P += w.x * x and Q += w.y * x, combined at the end (conv_k7.hip: no rotated operand, hence
    // no cross-half operand selection in the packed FMAs)
    v2f accp[4], accq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { accp[q] = v2f{0.f, 0.f}; accq[q] = v2f{0.f, 0.f}; }
#pragma unroll 4
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float2* row = &tile[ci][(m + a + py) * HCP + nq + px];
            v2f xv[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) { const float2 v = row[j]; xv[j] = v2f{v.x, v.y}; }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const float4 w = wf[cls][a][bb][ci];
                const v2f wx = v2f{w.x, w.x}, wy = v2f{w.z, w.z};          // (no cross-half operand selection: dcs_common.h)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    accp[q] = __builtin_elementwise_fma(wx, xv[q + bb], accp[q]);
                    accq[q] = __builtin_elementwise_fma(wy, xv[q + bb], accq[q]);
                }
            }
        }
    }
    v2f acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = v2f{accp[q].x - accq[q].y, accp[q].y + accq[q].x};
    const float br = p.b_r ? p.b_r[0] : 0.f, bi = p.b_i ? p.b_i[0] : 0.f;
    const int sy = m0 + m;
    if (sy >= p.Hs) return;
    const int Wo = 2 * p.Ws;
    float2* yrow = p.y + ((long)b * 2 * p.Hs + 2 * sy + py) * Wo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sx = n0 + nq + q;
        if (sx < p.Ws) yrow[2 * sx + px] = make_float2(acc[q].x + (br - bi), acc[q].y + (br + bi));
    }
}

}  // namespace

// x1 complex[B][Hs][Ws][C1], x2 complex[B][Hs][Ws][C2] (C1 + C2 = 16, both even); wt: the tap-rows panel of
// dcs_pack_tap_rows (complex[16][ct], ct >= 9, column tap = dy*3 + dx of the CORRELATION kernel); b_r / b_i: the layer's
// two real bias scalars (both or neither); y complex[B][2 Hs][2 Ws].
static int up2_single_impl(const void* x1, const void* x2, bool bf16_in, const float* wt, const float* b_r, const float* b_i,
                           float* y, int B, int Hs, int Ws, int C1, int C2, int ct, dcs_stream_t stream) {
    if (!x1 || !wt || !y || B <= 0 || B > 65535 || Hs <= 0 || Ws <= 0 || C1 <= 0 || C2 < 0 || C1 + C2 != CIN || (C1 & 1) ||
        (C2 & 1) || ct < 9 || ((C2 > 0) != (x2 != nullptr)) || ((b_r == nullptr) != (b_i == nullptr)))
        return DCS_ERR_BADARG;
    Up1Args p;
    p.x1 = x1; p.x2 = x2; p.wt = (const float2*)wt; p.b_r = b_r; p.b_i = b_i; p.y = (float2*)y;
    p.Hs = Hs; p.Ws = Ws; p.C1 = C1; p.C2 = C2; p.ct = ct;
    p.tiles_w = (Ws + SCT - 1) / SCT;
    p.tiles = p.tiles_w * ((Hs + SRT - 1) / SRT);
    if (bf16_in) DCS_LAUNCH(cconv_up1_kernel<unsigned short>, dim3(p.tiles, B), dim3(256), 0, dcs_stream(stream), p);
    else DCS_LAUNCH(cconv_up1_kernel<float>, dim3(p.tiles, B), dim3(256), 0, dcs_stream(stream), p);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_cconv_up2_single_fwd(const float* x1, const float* x2, const float* wt, const float* b_r, const float* b_i,
                                        float* y, int B, int Hs, int Ws, int C1, int C2, int ct, dcs_stream_t stream) {
    return up2_single_impl(x1, x2, false, wt, b_r, b_i, y, B, Hs, Ws, C1, C2, ct, stream);
}

// the two sources in bf16 (activations stored in bf16: BASELINE configs[4]); weights, bias and the result fp32
extern "C" int dcs_cconv_up2_single_fwd_h(const unsigned short* x1, const unsigned short* x2, const float* wt, const float* b_r,
                                          const float* b_i, float* y, int B, int Hs, int Ws, int C1, int C2, int ct,
                                          dcs_stream_t stream) {
    return up2_single_impl(x1, x2, true, wt, b_r, b_i, y, B, Hs, Ws, C1, C2, ct, stream);
}
