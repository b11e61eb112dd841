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

// per-pixel reductions over the G lanes of a pixel: DPP inside a row of 16 lanes (dcs_common.h)
__device__ __forceinline__ float gsum(float v, int G) { return dcs_group_sum(v, G); }
__device__ __forceinline__ float gmax(float v, int G) { return dcs_group_max(v, G); }

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
                                                            float* __restrict__ ca, int C, int Ch,
                                                            float* __restrict__ mx_out, float* __restrict__ hid_out) {
    __shared__ float mx[256];
    __shared__ float hid[64];
    const int b = blockIdx.x, t = threadIdx.x;
    for (int c = t; c < C; c += kThreads) {
        float m = -__builtin_huge_valf();
        for (int k = 0; k < nchunks; ++k) m = fmaxf(m, part[((long)b * nchunks + k) * C + c]);
        mx[c] = m;
        if (mx_out) mx_out[(long)b * C + c] = m;                           // (training: saved for the backward pass)
    }
    __syncthreads();
    for (int h = t >> 5; h < Ch; h += kThreads / 32) {                 // 32 lanes per hidden unit
        float a = 0.f;
        for (int c = t & 31; c < C; c += 32) a = fmaf(w1[h * C + c], mx[c], a);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((t & 31) == 0) {
            hid[h] = a > 0.f ? a : 0.f;
            if (hid_out) hid_out[(long)b * Ch + h] = hid[h];
        }
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


// ---- training: the same block as three differentiable pieces (pool | 7x7 conv | apply) ------------------------------------
// The 7x7 conv keeps the complex path's node (dcsnet/functional.py _CConv2dFn); the pools, the FC and the broadcast
// products get hand-written gradients here instead of ATen autograd (r_network.py:8-42 under loss.backward()).
// torch semantics kept: AdaptiveMaxPool2d and torch.max(dim=1) route the gradient to ONE element — the first maximum in
// scan order (lowest pixel index / lowest channel index).

// g_x = g_y ca sa ;  g_sa[p] = sum_c g_y x ca  (written as the real part of a complex value) ;
// part_ca[b][blk][c] = sum over this workgroup's pixels of g_y x sa
__global__ __launch_bounds__(kThreads) void r_apply_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                const float* __restrict__ ca, const float2* __restrict__ sa,
                                                                float* __restrict__ gx, float2* __restrict__ gsa,
                                                                float* __restrict__ part_ca, long HW, int G) {
    __shared__ float4 red[kThreads];
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G, b = blockIdx.y;
    const long base = (long)b * HW * G;
    const float4* x4 = reinterpret_cast<const float4*>(x) + base;
    const float4* g4 = reinterpret_cast<const float4*>(gy) + base;
    float4* o4 = reinterpret_cast<float4*>(gx) + base;
    const float4 a = reinterpret_cast<const float4*>(ca)[(long)b * G + g];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long iters = (HW + (long)gridDim.x * rpi - 1) / ((long)gridDim.x * rpi);   // same trip count in a group
    for (long k = 0; k < iters; ++k) {
        const long r = (k * gridDim.x + blockIdx.x) * rpi + r0;
        const bool ok = r < HW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), q = v;
        float s = 0.f;
        if (ok) { v = x4[r * G + g]; q = g4[r * G + g]; s = sa[(long)b * HW + r].x; }
        const float4 qx = make_float4(q.x * v.x, q.y * v.y, q.z * v.z, q.w * v.w);
        const float gs = gsum((qx.x * a.x + qx.y * a.y) + (qx.z * a.z + qx.w * a.w), G);
        acc.x = fmaf(qx.x, s, acc.x); acc.y = fmaf(qx.y, s, acc.y); acc.z = fmaf(qx.z, s, acc.z); acc.w = fmaf(qx.w, s, acc.w);
        if (ok) {
            o4[r * G + g] = make_float4(q.x * a.x * s, q.y * a.y * s, q.z * a.z * s, q.w * a.w * s);
            if (g == 0) gsa[(long)b * HW + r] = make_float2(gs, 0.f);
        }
    }
    red[t] = acc;
    __syncthreads();
    if (t < G) {
        for (int r = 1; r < rpi; ++r) {
            const float4 v = red[r * G + t];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(part_ca)[((long)b * gridDim.x + blockIdx.x) * G + t] = acc;
    }
}

// Through the spatial pool: z = ca x, g_z[p][c] = g_mean[p] / C + g_max[p] [c == first argmax_c z[p]];
// g_x += g_z ca ;  part_ca[b][blk][c] = sum_p g_z x ;  part_idx[b][blk][c] = first pixel of this workgroup's share with
// x == mx[b][c] (HW if none): where the channel max pool's gradient lands
template <bool ACC>
__global__ __launch_bounds__(kThreads) void r_pool_bwd_kernel(const float2* __restrict__ gpooled, const float* __restrict__ x,
                                                               const float* __restrict__ ca, const float* __restrict__ mx,
                                                               float* __restrict__ gx, float* __restrict__ part_ca,
                                                               int* __restrict__ part_idx, long HW, int C, int G) {
    __shared__ float4 red[kThreads];
    __shared__ int4 redi[kThreads];
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G, b = blockIdx.y;
    const long base = (long)b * HW * G;
    const float4* x4 = reinterpret_cast<const float4*>(x) + base;
    float4* o4 = reinterpret_cast<float4*>(gx) + base;
    const float4 a = reinterpret_cast<const float4*>(ca)[(long)b * G + g];
    const float4 m = reinterpret_cast<const float4*>(mx)[(long)b * G + g];
    const float invC = 1.f / (float)C;
    const int none = (int)(HW < 0x7fffffffL ? HW : 0x7fffffffL);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 first = make_int4(none, none, none, none);
    const long iters = (HW + (long)gridDim.x * rpi - 1) / ((long)gridDim.x * rpi);
    for (long k = 0; k < iters; ++k) {
        const long r = (k * gridDim.x + blockIdx.x) * rpi + r0;
        const bool ok = r < HW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 gp = make_float2(0.f, 0.f);
        if (ok) { v = x4[r * G + g]; gp = gpooled[(long)b * HW + r]; }
        const float z[4] = {a.x * v.x, a.y * v.y, a.z * v.z, a.w * v.w};
        float best = z[0];
        int bi = 4 * g;
#pragma unroll
        for (int e = 1; e < 4; ++e) if (z[e] > best) { best = z[e]; bi = 4 * g + e; }
        for (int o = G >> 1; o > 0; o >>= 1) {                             // (value, lowest index) over the pixel's lane group
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        const float gm = gp.x * invC;
        const float gz[4] = {gm + (bi == 4 * g ? gp.y : 0.f), gm + (bi == 4 * g + 1 ? gp.y : 0.f),
                             gm + (bi == 4 * g + 2 ? gp.y : 0.f), gm + (bi == 4 * g + 3 ? gp.y : 0.f)};
        if (ok) {
            float4 o = ACC ? o4[r * G + g] : make_float4(0.f, 0.f, 0.f, 0.f);
            o.x = fmaf(gz[0], a.x, o.x); o.y = fmaf(gz[1], a.y, o.y); o.z = fmaf(gz[2], a.z, o.z); o.w = fmaf(gz[3], a.w, o.w);
            o4[r * G + g] = o;
            acc.x = fmaf(gz[0], v.x, acc.x); acc.y = fmaf(gz[1], v.y, acc.y); acc.z = fmaf(gz[2], v.z, acc.z); acc.w = fmaf(gz[3], v.w, acc.w);
            const int ri = (int)r;                                         // (rows ascend with k: the first hit is the lowest)
            if (v.x == m.x && ri < first.x) first.x = ri;
            if (v.y == m.y && ri < first.y) first.y = ri;
            if (v.z == m.z && ri < first.z) first.z = ri;
            if (v.w == m.w && ri < first.w) first.w = ri;
        }
    }
    red[t] = acc; redi[t] = first;
    __syncthreads();
    if (t < G) {
        for (int r = 1; r < rpi; ++r) {
            const float4 v = red[r * G + t];
            const int4 f = redi[r * G + t];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            first.x = f.x < first.x ? f.x : first.x; first.y = f.y < first.y ? f.y : first.y;
            first.z = f.z < first.z ? f.z : first.z; first.w = f.w < first.w ? f.w : first.w;
        }
        reinterpret_cast<float4*>(part_ca)[((long)b * gridDim.x + blockIdx.x) * G + t] = acc;
        reinterpret_cast<int4*>(part_idx)[((long)b * gridDim.x + blockIdx.x) * G + t] = first;
    }
}

// One workgroup per sample: g_ca = g_ca_in + sum of the pool partials; through sigmoid, the second 1x1 conv, ReLU and the
// first; per-sample weight-gradient partials; the channel max pool's gradient g_mx added at its pixel of g_x.
__global__ __launch_bounds__(kThreads) void r_ca_fc_bwd_kernel(const float* __restrict__ g_ca_in, const float* __restrict__ part_ca,
                                                                const int* __restrict__ part_idx, int nblk,
                                                                const float* __restrict__ ca, const float* __restrict__ mx,
                                                                const float* __restrict__ hid, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, float* __restrict__ gx,
                                                                float* __restrict__ gw1_part, float* __restrict__ gw2_part,
                                                                long HW, int C, int Ch) {
    __shared__ float ga[256];
    __shared__ float gh[64];
    const int b = blockIdx.x, t = threadIdx.x;
    for (int c = t; c < C; c += kThreads) {
        float s = g_ca_in ? g_ca_in[(long)b * C + c] : 0.f;
        for (int k = 0; k < nblk; ++k) s += part_ca[((long)b * nblk + k) * C + c];
        const float a = ca[(long)b * C + c];
        ga[c] = s * a * (1.f - a);
    }
    __syncthreads();
    for (int h = t >> 5; h < Ch; h += kThreads / 32) {
        float a = 0.f;
        for (int c = t & 31; c < C; c += 32) a = fmaf(w2[c * Ch + h], ga[c], a);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((t & 31) == 0) gh[h] = hid[(long)b * Ch + h] > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int i = t; i < C * Ch; i += kThreads) {
        const int c2 = i / Ch, h2 = i % Ch;                                // fc.2.weight [C][Ch]
        gw2_part[(long)b * C * Ch + i] = ga[c2] * hid[(long)b * Ch + h2];
        const int h1 = i / C, c1 = i % C;                                  // fc.0.weight [Ch][C]
        gw1_part[(long)b * C * Ch + i] = gh[h1] * mx[(long)b * C + c1];
    }
    for (int c = t; c < C; c += kThreads) {
        float gm = 0.f;
        for (int h = 0; h < Ch; ++h) gm = fmaf(w1[h * C + c], gh[h], gm);
        int first = 0x7fffffff;
        for (int k = 0; k < nblk; ++k) {
            const int f = part_idx[((long)b * nblk + k) * C + c];
            first = f < first ? f : first;
        }
        if (first < HW) gx[((long)b * HW + first) * C + c] += gm;
    }
}

// out[b][c] = sum_k part[b][k][c], fixed order
__global__ __launch_bounds__(kThreads) void r_part_sum_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out, int C) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += kThreads) {
        float s = 0.f;
        for (int k = 0; k < nblk; ++k) s += part[((long)b * nblk + k) * C + c];
        out[(long)b * C + c] = s;
    }
}

// g_w[i] = sum_b part[b][i], fixed order
__global__ __launch_bounds__(kThreads) void r_fc_wgrad_reduce_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                                      float* __restrict__ g1, float* __restrict__ g2, int n, int B) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < B; ++b) { s1 += p1[(long)b * n + i]; s2 += p2[(long)b * n + i]; }
    g1[i] = s1; g2[i] = s2;
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
    DCS_LAUNCH(r_ca_fc_kernel, dim3(B), dim3(kThreads), 0, s, (const float*)part, nch, w1, w2, ca_out, C, Ch, (float*)nullptr, (float*)nullptr);
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


// ---- training entries ----------------------------------------------------------------------------------------------------
extern "C" long dcs_rattention_train_workspace_bytes(int B, long HW, int C, int Ch) {
    int G;
    if (B <= 0 || HW <= 0 || Ch <= 0 || !rgeom(C, &G)) return -1;
    const long nblk = rstream_grid(HW, G, B) > rchunks(HW, G) ? rstream_grid(HW, G, B) : rchunks(HW, G);
    // chunk partials (float + int per channel) | per-sample weight-gradient partials (both FC layers)
    return (long)B * nblk * C * 8 + 2L * B * C * Ch * (long)sizeof(float) + 1024;   // (+ alignment of the second part)
}

// pool piece: ca = sigmoid(fc(max_pool x)) with the pooled maxima and hidden activations kept (mx float[B][C], hid
// float[B][Ch]); pooled = (mean_c, max_c) of ca x as one complex channel, float2[B][H][W]
extern "C" int dcs_rattention_pool_fwd(const float* x, const float* w1, const float* w2, float* ca_out, float* mx_out,
                                       float* hid_out, float* pooled, void* workspace, long workspace_bytes, int B, int H, int W,
                                       int C, int Ch, dcs_stream_t stream) {
    int G;
    const long HW = (long)H * W;
    if (!x || !w1 || !w2 || !ca_out || !mx_out || !hid_out || !pooled || !workspace || B <= 0 || B > 65535 || H <= 0 || W <= 0 ||
        C > 256 || Ch <= 0 || Ch > 64 || !rgeom(C, &G))
        return DCS_ERR_BADARG;
    if (workspace_bytes < dcs_rattention_train_workspace_bytes(B, HW, C, Ch)) return DCS_ERR_WORKSPACE;
    const int nch = rchunks(HW, G), nxs = rstream_grid(HW, G, B);
    float* part = (float*)workspace;
    hipStream_t s = dcs_stream(stream);
    DCS_LAUNCH(r_ca_maxpool_kernel, dim3(nch, B), dim3(kThreads), 0, s, x, part, HW, C, G);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(r_ca_fc_kernel, dim3(B), dim3(kThreads), 0, s, (const float*)part, nch, w1, w2, ca_out, C, Ch, mx_out, hid_out);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(r_spatial_pool_kernel, dim3(nxs, B), dim3(kThreads), 0, s, x, (const float*)ca_out, (float2*)pooled, HW, C, G);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// y = x ca[c] Re(sa[p]);  sa float2[B][H][W] (the 7x7 conv's complex output)
extern "C" int dcs_rattention_apply_fwd(const float* x, const float* ca, const float* sa, float* y, int B, int H, int W, int C,
                                        dcs_stream_t stream) {
    int G;
    const long HW = (long)H * W;
    if (!x || !ca || !sa || !y || B <= 0 || B > 65535 || H <= 0 || W <= 0 || !rgeom(C, &G)) return DCS_ERR_BADARG;
    DCS_LAUNCH(r_apply_kernel, dim3(rstream_grid(HW, G, B), B), dim3(kThreads), 0, dcs_stream(stream), x, ca, (const float2*)sa, y, HW, G);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// gradients of the apply piece: g_x float[B][H][W][C], g_sa float2[B][H][W] (imaginary part 0), g_ca float[B][C]
extern "C" int dcs_rattention_apply_bwd(const float* gy, const float* x, const float* ca, const float* sa, float* gx, float* g_sa,
                                        float* g_ca, void* workspace, long workspace_bytes, int B, int H, int W, int C,
                                        dcs_stream_t stream) {
    int G;
    const long HW = (long)H * W;
    if (!gy || !x || !ca || !sa || !gx || !g_sa || !g_ca || !workspace || B <= 0 || B > 65535 || H <= 0 || W <= 0 || !rgeom(C, &G))
        return DCS_ERR_BADARG;
    if (workspace_bytes < dcs_rattention_train_workspace_bytes(B, HW, C, 1)) return DCS_ERR_WORKSPACE;
    const int nxs = rstream_grid(HW, G, B);
    float* part = (float*)workspace;
    hipStream_t s = dcs_stream(stream);
    DCS_LAUNCH(r_apply_bwd_kernel, dim3(nxs, B), dim3(kThreads), 0, s, gy, x, ca, (const float2*)sa, gx, (float2*)g_sa, part, HW, G);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(r_part_sum_kernel, dim3(B), dim3(kThreads), 0, s, (const float*)part, nxs, g_ca, C);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// gradients of the pool piece.  In: g_pooled float2[B][H][W] (cotangent of (mean_c, max_c)), g_ca float[B][C] (cotangent of
// ca from the apply piece; may be null), the saved ca / mx / hid.  gx float[B][H][W][C]: OVERWRITTEN when accumulate == 0,
// added to otherwise; gw1 float[Ch][C], gw2 float[C][Ch].
extern "C" int dcs_rattention_pool_bwd(const float* g_pooled, const float* g_ca, const float* x, const float* ca, const float* mx,
                                       const float* hid, const float* w1, const float* w2, float* gx, float* gw1, float* gw2,
                                       int accumulate, void* workspace, long workspace_bytes, int B, int H, int W, int C, int Ch,
                                       dcs_stream_t stream) {
    int G;
    const long HW = (long)H * W;
    if (!g_pooled || !x || !ca || !mx || !hid || !w1 || !w2 || !gx || !gw1 || !gw2 || !workspace || B <= 0 || B > 65535 ||
        H <= 0 || W <= 0 || HW >= 0x7fffffffL || C > 256 || Ch <= 0 || Ch > 64 || !rgeom(C, &G))
        return DCS_ERR_BADARG;
    if (workspace_bytes < dcs_rattention_train_workspace_bytes(B, HW, C, Ch)) return DCS_ERR_WORKSPACE;
    const int nxs = rstream_grid(HW, G, B);
    char* ws = (char*)workspace;
    float* part_ca = (float*)ws;
    int* part_idx = (int*)(ws + (long)B * nxs * C * 4);
    float* gw1_part = (float*)(ws + ((long)B * nxs * C * 8 + 255) / 256 * 256);
    float* gw2_part = gw1_part + (long)B * C * Ch;
    hipStream_t s = dcs_stream(stream);
    if (accumulate)
        DCS_LAUNCH(r_pool_bwd_kernel<true>, dim3(nxs, B), dim3(kThreads), 0, s, (const float2*)g_pooled, x, ca, mx, gx, part_ca, part_idx, HW, C, G);
    else
        DCS_LAUNCH(r_pool_bwd_kernel<false>, dim3(nxs, B), dim3(kThreads), 0, s, (const float2*)g_pooled, x, ca, mx, gx, part_ca, part_idx, HW, C, G);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(r_ca_fc_bwd_kernel, dim3(B), dim3(kThreads), 0, s, g_ca, (const float*)part_ca, (const int*)part_idx, nxs, ca, mx, hid,
               w1, w2, gx, gw1_part, gw2_part, HW, C, Ch);
    DCS_CHECK_LAUNCH();
    const int n = C * Ch;
    DCS_LAUNCH(r_fc_wgrad_reduce_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, s, (const float*)gw1_part,
               (const float*)gw2_part, gw1, gw2, n, B);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
