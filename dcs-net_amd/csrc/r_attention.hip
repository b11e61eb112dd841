// r_attention.hip — the real-valued CBAM attention pair of DR-Net (r_network.py:8-42, applied at :155-158 and
// :166-167) on channels-last float[B][HW][C]:
//     ca = sigmoid(fc(max_pool(x)))                 (the avg branch is computed and overwritten: r_network.py:23-24)
//     z  = ca (.) x ;  sa = sigmoid(conv7x7(cat(mean_c z, max_c z))) ;  y = sa (.) z
// Same structure as attention.hip: one streaming pass per pool, the tiny FC per sample, z never stored.
//   r_ca_maxpool_kernel   1 read of x  -> per-sample channel maxima (chunk slabs)
//   r_ca_fc_kernel        slab max, 1x1 conv, ReLU, 1x1 conv, sigmoid          (weights in the nn.Conv2d layout)
//   r_spatial_pool_kernel 1 read of x  -> (mean_c, max_c) of ca*x per pixel, written as ONE complex channel: the 2 -> 1
//                         real 7x7 conv is then a 1 -> 1 complex conv with weights (w_mean - j w_max) whose real part is the
//                         answer (conv_direct.hip, sigmoid epilogue)
//   r_apply_kernel        1 read + 1 write -> x * ca[c] * sa[p]
// A lane group of G = C/4 lanes owns one pixel (one float4 = 4 channels per lane).
#include "conv_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunks = 64;

inline bool rgeom(int C, int* G) {
    if (C < 4 || (C & 3)) return false;
    const int g = C / 4;
    if (g > 64 || (g & (g - 1)) != 0) return false;       // lane group inside one wave
    *G = g;
    return true;
}
inline int rchunks(long HW, int G) {
    const int rpi = kThreads / G;
    const long it = (HW + rpi - 1) / rpi;
    const long nb = (it + 7) / 8;
    return (int)(nb < 1 ? 1 : (nb > kMaxChunks ? kMaxChunks : nb));
}
inline int rstream_grid(long HW, int G, int B) {
    const int rpi = kThreads / G;
    const long it = (HW + rpi - 1) / rpi;
    long nb = (it + 3) / 4;
    const long cap = (2048 + B - 1) / B;
    if (nb > cap) nb = cap;
    return (int)(nb < 1 ? 1 : nb);
}

__device__ __forceinline__ float gsum(float v, int G) {
    for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float gmax(float v, int G) {
    for (int o = G >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// part[b][chunk][C]: max over this chunk's pixels
__global__ __launch_bounds__(kThreads) void r_ca_maxpool_kernel(const float* __restrict__ x, float* __restrict__ part, long HW,
                                                                 int C, int G) {
    __shared__ float4 red[kThreads];
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G, b = blockIdx.y;
    const float4* x4 = reinterpret_cast<const float4*>(x) + (long)b * HW * G;
    const float ninf = -__builtin_huge_valf();
    float4 m = make_float4(ninf, ninf, ninf, ninf);
    for (long r = (long)blockIdx.x * rpi + r0; r < HW; r += (long)gridDim.x * rpi) {
        const float4 v = x4[r * G + g];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
    red[t] = m;
    __syncthreads();
    if (t < G) {
        for (int r = 1; r < rpi; ++r) {
            const float4 v = red[r * G + t];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        reinterpret_cast<float4*>(part)[((long)b * gridDim.x + blockIdx.x) * G + t] = m;
    }
}

// one workgroup per sample; w1 float[Ch][C] (fc.0.weight), w2 float[C][Ch] (fc.2.weight)
__global__ __launch_bounds__(kThreads) void r_ca_fc_kernel(const float* __restrict__ part, int nchunks,
                                                            const float* __restrict__ w1, const float* __restrict__ w2,
                                                            float* __restrict__ ca, int C, int Ch) {
    __shared__ float mx[256];
    __shared__ float hid[64];
    const int b = blockIdx.x, t = threadIdx.x;
    for (int c = t; c < C; c += kThreads) {
        float m = -__builtin_huge_valf();
        for (int k = 0; k < nchunks; ++k) m = fmaxf(m, part[((long)b * nchunks + k) * C + c]);
        mx[c] = m;
    }
    __syncthreads();
    for (int h = t >> 5; h < Ch; h += kThreads / 32) {                 // 32 lanes per hidden unit
        float a = 0.f;
        for (int c = t & 31; c < C; c += 32) a = fmaf(w1[h * C + c], mx[c], a);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((t & 31) == 0) hid[h] = a > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int c = t; c < C; c += kThreads) {
        float a = 0.f;
        for (int h = 0; h < Ch; ++h) a = fmaf(w2[c * Ch + h], hid[h], a);
        ca[(long)b * C + c] = 1.f / (1.f + expf(-a));
    }
}

// pooled[b][p] = (mean_c z, max_c z), z = ca[b][c] * x[b][p][c]
__global__ __launch_bounds__(kThreads) void r_spatial_pool_kernel(const float* __restrict__ x, const float* __restrict__ ca,
                                                                   float2* __restrict__ pooled, long HW, int C, int G) {
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G, b = blockIdx.y;
    const float4* x4 = reinterpret_cast<const float4*>(x) + (long)b * HW * G;
    const float4 a = reinterpret_cast<const float4*>(ca)[(long)b * G + g];
    const float invC = 1.f / (float)C;
    const long iters = (HW + (long)gridDim.x * rpi - 1) / ((long)gridDim.x * rpi);   // same trip count in a group
    for (long k = 0; k < iters; ++k) {
        const long r = (k * gridDim.x + blockIdx.x) * rpi + r0;
        const bool ok = r < HW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = x4[r * G + g];
        const float z0 = a.x * v.x, z1 = a.y * v.y, z2 = a.z * v.z, z3 = a.w * v.w;
        const float s = gsum((z0 + z1) + (z2 + z3), G);
        const float m = gmax(fmaxf(fmaxf(z0, z1), fmaxf(z2, z3)), G);
        if (ok && g == 0) pooled[(long)b * HW + r] = make_float2(s * invC, m);
    }
}

// y = x * ca[c] * Re(sa[p])
__global__ __launch_bounds__(kThreads) void r_apply_kernel(const float* __restrict__ x, const float* __restrict__ ca,
                                                            const float2* __restrict__ sa, float* __restrict__ y, long HW,
                                                            int G) {
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G, b = blockIdx.y;
    const long base = (long)b * HW * G;
    const float4* x4 = reinterpret_cast<const float4*>(x) + base;
    float4* y4 = reinterpret_cast<float4*>(y) + base;
    const float4 a = reinterpret_cast<const float4*>(ca)[(long)b * G + g];
    for (long r = (long)blockIdx.x * rpi + r0; r < HW; r += (long)gridDim.x * rpi) {
        const float4 v = x4[r * G + g];
        const float s = sa[(long)b * HW + r].x;
        y4[r * G + g] = make_float4(v.x * a.x * s, v.y * a.y * s, v.z * a.z * s, v.w * a.w * s);
    }
}

}  // namespace

extern "C" long dcs_rattention_workspace_bytes(int B, long HW, int C) {
    int G;
    if (B <= 0 || HW <= 0 || !rgeom(C, &G)) return -1;
    // chunk maxima | pooled (mean, max) as one complex channel | sa (complex, real part used)
    return (long)B * rchunks(HW, G) * C * (long)sizeof(float) + 2L * B * HW * (long)sizeof(float2) + 512;
}

// x, y float[B][H][W][C]; w1 / w2: fc.0.weight [Ch][C] and fc.2.weight [C][Ch] (nn.Conv2d 1x1, no bias); wsa / sa_bias:
// dcs_pack_conv_weight of the 7x7 conv as a 1 -> 1 complex conv (w_mean, -w_max) with zero bias; ca_out float[B][C]
extern "C" int dcs_rattention_fwd(const float* x, const float* w1, const float* w2, const float* wsa, const float* sa_bias,
                                  float* ca_out, float* y, void* workspace, long workspace_bytes, int B, int H, int W, int C,
                                  int Ch, int ksize, dcs_stream_t stream) {
    int G;
    const long HW = (long)H * W;
    if (!x || !w1 || !w2 || !wsa || !sa_bias || !ca_out || !y || !workspace || B <= 0 || B > 65535 || H <= 0 || W <= 0 ||
        C > 256 || Ch <= 0 || Ch > 64 || ksize < 1 || !(ksize & 1) || !rgeom(C, &G))
        return DCS_ERR_BADARG;
    if (workspace_bytes < dcs_rattention_workspace_bytes(B, HW, C)) return DCS_ERR_WORKSPACE;
    const int nch = rchunks(HW, G), nxs = rstream_grid(HW, G, B);
    char* ws = (char*)workspace;
    float* part = (float*)ws;
    ws += ((long)B * nch * C * (long)sizeof(float) + 255) / 256 * 256;
    float2* pooled = (float2*)ws;
    float2* sa = pooled + (long)B * HW;
    hipStream_t s = dcs_stream(stream);
    DCS_LAUNCH(r_ca_maxpool_kernel, dim3(nch, B), dim3(kThreads), 0, s, x, part, HW, C, G);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(r_ca_fc_kernel, dim3(B), dim3(kThreads), 0, s, (const float*)part, nch, w1, w2, ca_out, C, Ch);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(r_spatial_pool_kernel, dim3(nxs, B), dim3(kThreads), 0, s, x, (const float*)ca_out, pooled, HW, C, G);
    DCS_CHECK_LAUNCH();
    conv::Args a{};
    a.x1 = pooled; a.x2 = nullptr; a.wp = (const float2*)wsa; a.bias = (const float2*)sa_bias; a.y = sa;
    a.B = B; a.Hin = H; a.Win = W; a.C1 = 1; a.C2 = 0; a.up_f = 1; a.up_t = 1; a.zero_ins = 0; a.Cout = 1;
    a.kh = ksize; a.kw = ksize; a.sf = 1; a.st = 1; a.pad_f = ksize / 2; a.pad_t = ksize / 2; a.act = DCS_ACT_SIGMOID;
    a.Hv = H; a.Wv = W; a.Hout = H; a.Wout = W;
    const int rc = dcs_conv_direct_multi(&a, 1, s);
    if (rc != DCS_OK) return rc;
    DCS_LAUNCH(r_apply_kernel, dim3(nxs, B), dim3(kThreads), 0, s, x, (const float*)ca_out, (const float2*)sa, y, HW, G);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
