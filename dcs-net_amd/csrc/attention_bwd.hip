// attention_bwd.hip — backward of the fused attention block
//     out = dropout( sa (.) ca (.) x ),   ca = sigmoid_c(2 fc2(relu_c(fc1(mean_p x)))),
//     sa = sigmoid_c(conv7x7([mean_c z, max_c Re z + j max_c Im z])),   z = ca (.) x
// (c_network.py:53-84 applied at :208-211 / :219-222).  Autograd in the reference walks ~25
// element-wise / reduction nodes per attention pair, each streaming the activation.  Here the
// activation-sized work is three streaming passes:
//   att_bwd_sa_kernel   reads x, g_out            -> g_pre[p] = sigmoid'(sa) * sum_c g_o conj(z)
//   (7x7 conv data / weight gradients: conv_direct.hip)
//   att_bwd_x_kernel    reads x, g_out, writes g_x -> g_z = conj(sa) g_o + pool gradients (mean, arg-max
//                       recomputed with wavefront shuffles), g_x = conj(ca) g_z, and the per-sample
//                       partial sums of g_ca = sum_p g_z conj(x)   (fp64 slabs, no atomics)
//   ca_bwd_kernel       one workgroup: sigmoid', both 1x1 convs, CReLU mask, weight gradients in the
//                       reference's layout, and g_pooled
//   att_bwd_pool_kernel g_x += g_pooled[b][c] / HW
// z, the masks and the arg-max indices are recomputed, never stored.
#include "conv_common.h"

namespace {

constexpr int kThreads = 256;
#ifndef DCS_ATT_RED_IT
#define DCS_ATT_RED_IT 2          // row passes per reduction workgroup: each pass is loads + cross-lane reductions IN SERIES (~1 us),
                                  // so eight of them made every decoder block's kernels 16-20 us whatever the tensor size
#endif
#ifndef DCS_ATT_APP_IT
#define DCS_ATT_APP_IT 2          // row passes per streaming workgroup
#endif
#ifndef DCS_ATT_SMALL_IT
#define DCS_ATT_SMALL_IT 256      // row passes of a sample below which the short-chain grids are used
#endif
#ifndef DCS_ATT_MAX_CHUNKS
#define DCS_ATT_MAX_CHUNKS 256
#endif
#ifndef DCS_ATT_GRID_CAP
#define DCS_ATT_GRID_CAP 8192
#endif
constexpr int kMaxChunks = DCS_ATT_MAX_CHUNKS;

inline bool att_geom(int C, int* G) {
    if (C < 2 || (C & 1)) return false;
    int g = C / 2;
    if (g > 64 || (g & (g - 1)) != 0) return false;
    *G = g;
    return true;
}
inline int chunks_for(long HW, int G) {
    const int rpi = kThreads / G;
    long it = (HW + rpi - 1) / rpi;
    // few passes only where there are few rows to begin with (the train shapes' decoder blocks); large maps keep 8 passes and
    // 64 chunks (more chunks there cost more in slab traffic than the shorter chains return: inference 4.31 -> 4.43 ms)
    long nb = it <= DCS_ATT_SMALL_IT ? (it + DCS_ATT_RED_IT - 1) / DCS_ATT_RED_IT : (it + 7) / 8;
    if (it > DCS_ATT_SMALL_IT && nb > 64) nb = 64;
    return (int)(nb < 1 ? 1 : (nb > kMaxChunks ? kMaxChunks : nb));
}

// per-pixel reductions over the G lanes of a pixel: DPP inside a row of 16 lanes (dcs_common.h)
__device__ __forceinline__ float gsum(float v, int G) { return dcs_group_sum(v, G); }
__device__ __forceinline__ float gmax(float v, int G) { return dcs_group_max(v, G); }
__device__ __forceinline__ int gmin_i(int v, int G) { return dcs_group_min_i(v, G); }

template <bool DROP>
__device__ __forceinline__ float4 masked(float4 g, uint64_t seed, uint64_t e, float p, float inv_keep) {
    if (DROP) {
        g.x *= dcs_keep_scale(seed, e, p, inv_keep);
        g.y *= dcs_keep_scale(seed, e + 1, p, inv_keep);
        g.z *= dcs_keep_scale(seed, e + 2, p, inv_keep);
        g.w *= dcs_keep_scale(seed, e + 3, p, inv_keep);
    }
    return g;
}

// g_pre[b][p] = sigmoid'(sa[p]) (.) sum_c g_o[p][c] conj(ca[c] x[p][c])
template <bool DROP>
__device__ __forceinline__ void att_bwd_sa_kernel_body(const act_t* __restrict__ x, const act_t* __restrict__ go,
                                                               const float* __restrict__ ca,
                                                               const float2* __restrict__ sa, float2* __restrict__ gpre,
                                                               long HW, int G, float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev, int vbx, int vby, int vgx) {
    if (seed_dev) seed += seed_dev[0];   // per-step device-side offset (graph replay safe)
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G;
    const int b = vby;
    const long base = (long)b * HW * G;
    const ActIn4<act_t> x4 = act_in4(x) + base, g4 = act_in4(go) + base;
    const float4 a = reinterpret_cast<const float4*>(ca)[(long)b * G + g];
    const float inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
    const long iters = (HW + (long)vgx * rpi - 1) / ((long)vgx * rpi);
    for (long k = 0; k < iters; ++k) {
        const long r = (k * vgx + vbx) * rpi + r0;
        const bool ok = r < HW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), gg = v;
        if (ok) {
            v = x4[r * G + g];
            gg = masked<DROP>(g4[r * G + g], seed, (uint64_t)(base + r * G + g) * 4, drop_p, inv_keep);
        }
        const float z0r = a.x * v.x - a.y * v.y, z0i = a.x * v.y + a.y * v.x;
        const float z1r = a.z * v.z - a.w * v.w, z1i = a.z * v.w + a.w * v.z;
        // g conj(z)
        float sr = gg.x * z0r + gg.y * z0i + gg.z * z1r + gg.w * z1i;
        float si = gg.y * z0r - gg.x * z0i + gg.w * z1r - gg.z * z1i;
        sr = gsum(sr, G); si = gsum(si, G);
        if (ok && g == 0) {
            const float2 s = sa[(long)b * HW + r];
            gpre[(long)b * HW + r] = make_float2(sr * s.x * (1.f - s.x), si * s.y * (1.f - s.y));
        }
    }
}

// g_x = conj(ca) g_z ; part[b][chunk][C][2] = sum_p g_z conj(x)
template <bool DROP>
__device__ __forceinline__ void att_bwd_x_kernel_body(const act_t* __restrict__ x, const act_t* __restrict__ go,
                                                              const float* __restrict__ ca, const float2* __restrict__ sa,
                                                              const float4* __restrict__ gsp, act_t* __restrict__ gx,
                                                              double* __restrict__ part, long HW, int C, int G,
                                                              float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev, int vbx, int vby, int vgx) {
    if (seed_dev) seed += seed_dev[0];   // per-step device-side offset (graph replay safe)
    __shared__ double red[kThreads * 4];
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G;
    const int b = vby;
    const long base = (long)b * HW * G;
    const ActIn4<act_t> x4 = act_in4(x) + base, g4 = act_in4(go) + base;
    const ActOut4<act_t> o4 = act_out4(gx) + base;
    const float4 a = reinterpret_cast<const float4*>(ca)[(long)b * G + g];
    const float inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
    const float invC = 1.f / (float)C;
    float c0r = 0.f, c0i = 0.f, c1r = 0.f, c1i = 0.f;
    const long iters = (HW + (long)vgx * rpi - 1) / ((long)vgx * rpi);
    for (long k = 0; k < iters; ++k) {
        const long r = (k * vgx + vbx) * rpi + r0;
        const bool ok = r < HW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), gg = v, gp = v;
        float2 s = make_float2(0.f, 0.f);
        if (ok) {
            v = x4[r * G + g];
            gg = masked<DROP>(g4[r * G + g], seed, (uint64_t)(base + r * G + g) * 4, drop_p, inv_keep);
            s = sa[(long)b * HW + r];
            gp = gsp[(long)b * HW + r];          // (g_mean.re, g_mean.im, g_max.re, g_max.im)
        }
        const float z0r = a.x * v.x - a.y * v.y, z0i = a.x * v.y + a.y * v.x;
        const float z1r = a.z * v.z - a.w * v.w, z1i = a.z * v.w + a.w * v.z;
        // arg-max over channels, first index on ties (torch.max on the CPU oracle)
        const float mr = gmax(fmaxf(z0r, z1r), G), mi = gmax(fmaxf(z0i, z1i), G);
        const int big = 1 << 30;
        const int ir = gmin_i(z0r == mr ? 2 * g : (z1r == mr ? 2 * g + 1 : big), G);
        const int ii = gmin_i(z0i == mi ? 2 * g : (z1i == mi ? 2 * g + 1 : big), G);
        // g_z = conj(sa) g_o + g_mean / C + one-hot(arg-max) g_max
        float t0r = s.x * gg.x + s.y * gg.y + gp.x * invC + (ir == 2 * g ? gp.z : 0.f);
        float t0i = s.x * gg.y - s.y * gg.x + gp.y * invC + (ii == 2 * g ? gp.w : 0.f);
        float t1r = s.x * gg.z + s.y * gg.w + gp.x * invC + (ir == 2 * g + 1 ? gp.z : 0.f);
        float t1i = s.x * gg.w - s.y * gg.z + gp.y * invC + (ii == 2 * g + 1 ? gp.w : 0.f);
        if (ok) {
            // g_x = conj(ca) g_z
            o4[r * G + g] = make_float4(a.x * t0r + a.y * t0i, a.x * t0i - a.y * t0r,
                                        a.z * t1r + a.w * t1i, a.z * t1i - a.w * t1r);
            // g_ca += g_z conj(x)
            c0r += t0r * v.x + t0i * v.y; c0i += t0i * v.x - t0r * v.y;
            c1r += t1r * v.z + t1i * v.w; c1i += t1i * v.z - t1r * v.w;
        }
    }
    red[t * 4 + 0] = c0r; red[t * 4 + 1] = c0i; red[t * 4 + 2] = c1r; red[t * 4 + 3] = c1i;
    __syncthreads();
    for (int o = t; o < G * 4; o += kThreads) {
        const int gg = o / 4, i = o % 4;
        double acc = 0;
        for (int r = 0; r < rpi; ++r) acc += red[(r * G + gg) * 4 + i];
        part[(((long)b * vgx + vbx) * C + 2 * gg) * 2 + i] = acc;
    }
}

// Per-sample half (grid = B): slab sum -> g_o = 2 sigmoid'(.) g_ca ; g_h = relu'(h) (.) W2^H g_o ;
// g_pooled = W1^H g_h.  scratch: float2 go[B][C], gh[B][Ch], gpooled[B][C]
__device__ __forceinline__ void ca_bwd_sample_kernel_body(const double* __restrict__ part, int nchunks,
                                                                  const float2* __restrict__ ca,
                                                                  const float2* __restrict__ hidden,
                                                                  const float2* __restrict__ w1,
                                                                  const float2* __restrict__ w2, float2* __restrict__ go,
                                                                  float2* __restrict__ gh, float2* __restrict__ gpooled,
                                                                  int C, int Ch, int vbx, int vby, int vgx) {
    __shared__ float2 go_s[128];
    __shared__ float2 gh_s[64];
    const int b = vbx, t = threadIdx.x;
    for (int c = t; c < C; c += kThreads) {
        double sr = 0, si = 0;
#pragma unroll 8
        for (int k = 0; k < nchunks; ++k) {                            // (unrolled: the chunk loads go out together)
            const double* p = part + (((long)b * nchunks + k) * C + c) * 2;
            sr += p[0]; si += p[1];
        }
        const float2 s = ca[(long)b * C + c];
        const float2 g = make_float2(2.f * (float)sr * s.x * (1.f - s.x), 2.f * (float)si * s.y * (1.f - s.y));
        go_s[c] = g;
        go[(long)b * C + c] = g;
    }
    __syncthreads();
    // g_h = relu'(h) (.) W2^H g_o: LP lanes per hidden unit share the sum over the C channels (shuffle reduction).  One thread
    // per hidden unit — Ch <= 16 of the 256 — walked C dependent weight loads on its own: ~3 us of this launch-bound kernel.
    {
        int chp = 1;
        while (chp < Ch) chp <<= 1;
        const int LP = kThreads / chp > 64 ? 64 : kThreads / chp;       // (a power of two <= 64: the lanes of a unit share a wave)
        for (int h0 = 0; h0 < Ch; h0 += kThreads / LP) {
            const int h = h0 + t / LP, sub = t % LP;
            float ar = 0.f, ai = 0.f;
            if (h < Ch)
                for (int c = sub; c < C; c += LP) {
                    const float2 w = w2[h * C + c], g = go_s[c];
                    ar += w.x * g.x + w.y * g.y; ai += w.x * g.y - w.y * g.x;
                }
            for (int o = LP / 2; o > 0; o >>= 1) { ar += __shfl_xor(ar, o, 64); ai += __shfl_xor(ai, o, 64); }
            if (h < Ch && sub == 0) {
                const float2 hv = hidden[(long)b * Ch + h];
                const float2 g = make_float2(hv.x > 0.f ? ar : 0.f, hv.y > 0.f ? ai : 0.f);
                gh_s[h] = g;
                gh[(long)b * Ch + h] = g;
            }
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += kThreads) {
        float ar = 0.f, ai = 0.f;
        for (int h = 0; h < Ch; ++h) {
            const float2 w = w1[c * Ch + h], g = gh_s[h];
            ar += w.x * g.x + w.y * g.y; ai += w.x * g.y - w.y * g.x;
        }
        gpooled[(long)b * C + c] = make_float2(ar, ai);
    }
}

// Weight half (kWLanes lanes per weight element, each summing a strided share of the batch; shuffle reduction):
//   g_w2[h][c] = sum_b g_o conj(relu(h)) ;  g_w1[c][h] = sum_b g_h conj(pooled)
// One thread per element is a chain of ~B dependent memory round trips (15-20 us on its own once nothing hides it).
struct CaWeightArgs {
    const float2* go; const float2* gh; const float2* pooled; const float2* hidden;
    float* g_fc0_r; float* g_fc0_i; float* g_fc2_r; float* g_fc2_i;
    int B, C, Ch;
};

constexpr int kWLanes = 32;

// gi: global thread index inside the weight half; all 64 lanes of a wave enter (two elements per wave)
__device__ __forceinline__ void ca_bwd_weight_element(const CaWeightArgs& w, int gi) {
    const int i = gi / kWLanes, sub = gi % kWLanes;
    const bool live = i < w.C * w.Ch;
    const int c = live ? i / w.Ch : 0, h = live ? i % w.Ch : 0;
    float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f;
    for (int b = sub; live && b < w.B; b += kWLanes) {
        const float2 g = w.go[(long)b * w.C + c];
        float2 hv = w.hidden[(long)b * w.Ch + h];
        hv.x = hv.x > 0.f ? hv.x : 0.f; hv.y = hv.y > 0.f ? hv.y : 0.f;
        ar += g.x * hv.x + g.y * hv.y; ai += g.y * hv.x - g.x * hv.y;
        const float2 q = w.gh[(long)b * w.Ch + h], p = w.pooled[(long)b * w.C + c];
        br += q.x * p.x + q.y * p.y; bi += q.y * p.x - q.x * p.y;
    }
#pragma unroll
    for (int o = kWLanes / 2; o > 0; o >>= 1) {
        ar += __shfl_xor(ar, o, 64); ai += __shfl_xor(ai, o, 64);
        br += __shfl_xor(br, o, 64); bi += __shfl_xor(bi, o, 64);
    }
    if (!live || sub != 0) return;
    w.g_fc2_r[c * w.Ch + h] = ar;                             // fc.2 weight [C][Ch][1][1]
    w.g_fc2_i[c * w.Ch + h] = ai;
    w.g_fc0_r[h * w.C + c] = br;                              // fc.0 weight [Ch][C][1][1]
    w.g_fc0_i[h * w.C + c] = bi;
}

// g_x += g_pooled / HW (broadcast over the sample's pixels).  Workgroups with blockIdx.x >= nx_pool of batch row 0 are
// the FC weight-gradient half (nothing downstream waits for it, so it rides along instead of taking its own launch).
__device__ __forceinline__ void att_bwd_pool_kernel_body(act_t* __restrict__ gx, const float* __restrict__ gpooled,
                                                                 long HW, int G, float inv_hw, int nx_pool, CaWeightArgs w, int vbx, int vby, int vgx) {
    if ((int)vbx >= nx_pool) {
        if (vby == 0) ca_bwd_weight_element(w, ((int)vbx - nx_pool) * kThreads + threadIdx.x);
        return;
    }
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G;
    const int b = vby;
    const ActOut4<act_t> o4 = act_out4(gx) + (long)b * HW * G;
    const ActIn4<act_t> i4 = act_in4((const act_t*)gx) + (long)b * HW * G;
    float4 p = reinterpret_cast<const float4*>(gpooled)[(long)b * G + g];
    p.x *= inv_hw; p.y *= inv_hw; p.z *= inv_hw; p.w *= inv_hw;
    for (long r = (long)vbx * rpi + r0; r < HW; r += (long)nx_pool * rpi) {
        float4 v = i4[r * G + g];
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        o4[r * G + g] = v;
    }
}

// ---- launch forms: one problem per launch, or several problems (blockIdx.z) sharing one launch (see attention.hip) ----
constexpr int kMaxBatch = 8;
// x0: first blockIdx.x of each problem in the compacted grid (attention.hip)
template <class P> struct Tbl { P p[kMaxBatch]; int x0[kMaxBatch + 1]; };
template <class P> __device__ __forceinline__ int tbl_find(const Tbl<P>& t, int bx) {
    int z = 0;
#pragma unroll
    for (int k = 1; k < kMaxBatch; ++k) z += bx >= t.x0[k] ? 1 : 0;
    return z;
}
template <class P> void tbl_compact(Tbl<P>& t, int n, int* total) {
    int a = 0;
    for (int k = 0; k <= kMaxBatch; ++k) {
        t.x0[k] = k < n ? a : 0x7fffffff;
        if (k < n) a += t.p[k].nx;
    }
    t.x0[n] = a;
    *total = a;
}

struct BwdSaP { const act_t* x; const act_t* go; const float* ca; const float2* sa; float2* gpre; long HW; int G, nx; };
struct BwdXP { const act_t* x; const act_t* go; const float* ca; const float2* sa; const float4* gsp; act_t* gx; double* part;
               long HW; int C, G, nx; };
struct CaBwdP { const double* part; int nchunks; const float2* ca; const float2* hidden; const float2* w1; const float2* w2;
                float2* go; float2* gh; float2* gpooled; int C, Ch; };
struct PoolP { act_t* gx; const float* gpooled; long HW; int G; float inv_hw; int nx_pool, nx; CaWeightArgs w; };

template <bool DROP>
__global__ __launch_bounds__(kThreads) void att_bwd_sa_kernel(BwdSaP p, float drop_p, uint64_t seed, const uint64_t* seed_dev) {
    DCS_PRIO_CRITICAL();
    att_bwd_sa_kernel_body<DROP>(p.x, p.go, p.ca, p.sa, p.gpre, p.HW, p.G, drop_p, seed, seed_dev, blockIdx.x, blockIdx.y, gridDim.x);
}
__global__ __launch_bounds__(kThreads) void att_bwd_sa_multi_kernel(Tbl<BwdSaP> t) {
    DCS_PRIO_CRITICAL();
    const int z = tbl_find(t, blockIdx.x);
    const BwdSaP& p = t.p[z];
    att_bwd_sa_kernel_body<false>(p.x, p.go, p.ca, p.sa, p.gpre, p.HW, p.G, 0.f, 0, nullptr, blockIdx.x - t.x0[z], blockIdx.y, p.nx);
}
template <bool DROP>
__global__ __launch_bounds__(kThreads) void att_bwd_x_kernel(BwdXP p, float drop_p, uint64_t seed, const uint64_t* seed_dev) {
    DCS_PRIO_CRITICAL();
    att_bwd_x_kernel_body<DROP>(p.x, p.go, p.ca, p.sa, p.gsp, p.gx, p.part, p.HW, p.C, p.G, drop_p, seed, seed_dev, blockIdx.x,
                                blockIdx.y, gridDim.x);
}
__global__ __launch_bounds__(kThreads) void att_bwd_x_multi_kernel(Tbl<BwdXP> t) {
    DCS_PRIO_CRITICAL();
    const int z = tbl_find(t, blockIdx.x);
    const BwdXP& p = t.p[z];
    att_bwd_x_kernel_body<false>(p.x, p.go, p.ca, p.sa, p.gsp, p.gx, p.part, p.HW, p.C, p.G, 0.f, 0, nullptr, blockIdx.x - t.x0[z],
                                 blockIdx.y, p.nx);
}
__global__ __launch_bounds__(kThreads) void ca_bwd_sample_kernel(CaBwdP p) {
    DCS_PRIO_CRITICAL();
    ca_bwd_sample_kernel_body(p.part, p.nchunks, p.ca, p.hidden, p.w1, p.w2, p.go, p.gh, p.gpooled, p.C, p.Ch, blockIdx.x, 0, 0);
}
__global__ __launch_bounds__(kThreads) void ca_bwd_sample_multi_kernel(Tbl<CaBwdP> t) {
    DCS_PRIO_CRITICAL();
    const CaBwdP& p = t.p[blockIdx.z];
    ca_bwd_sample_kernel_body(p.part, p.nchunks, p.ca, p.hidden, p.w1, p.w2, p.go, p.gh, p.gpooled, p.C, p.Ch, blockIdx.x, 0, 0);
}
__global__ __launch_bounds__(kThreads) void att_bwd_pool_kernel(PoolP p) {
    DCS_PRIO_CRITICAL();
    att_bwd_pool_kernel_body(p.gx, p.gpooled, p.HW, p.G, p.inv_hw, p.nx_pool, p.w, blockIdx.x, blockIdx.y, gridDim.x);
}
__global__ __launch_bounds__(kThreads) void att_bwd_pool_multi_kernel(Tbl<PoolP> t) {
    DCS_PRIO_CRITICAL();
    const int z = tbl_find(t, blockIdx.x);
    const PoolP& p = t.p[z];
    att_bwd_pool_kernel_body(p.gx, p.gpooled, p.HW, p.G, p.inv_hw, p.nx_pool, p.w, blockIdx.x - t.x0[z], blockIdx.y, p.nx);
}

inline int stream_grid(long HW, int G, int B) {
    const int rpi = kThreads / G;
    long it = (HW + rpi - 1) / rpi;
    long nb = it <= DCS_ATT_SMALL_IT ? (it + DCS_ATT_APP_IT - 1) / DCS_ATT_APP_IT : (it + 3) / 4;
    long cap = (it <= DCS_ATT_SMALL_IT ? DCS_ATT_GRID_CAP : 2048) / (B > 0 ? B : 1);
    if (cap < 1) cap = 1;
    return (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}

}  // namespace

extern "C" int DCS_SYM(dcs_attention_bwd_sa)(const act_t* x, const act_t* g_out, const float* ca, const float* sa, float* g_pre,
                                    int B, long HW, int C, float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream) {
    int G;
    if (!x || !g_out || !ca || !sa || !g_pre || B <= 0 || B > 65535 || HW <= 0 || !att_geom(C, &G)) return DCS_ERR_BADARG;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return DCS_ERR_BADARG;
    const int nx = stream_grid(HW, G, B);
    dim3 grid(nx, B);
    const BwdSaP sp{x, g_out, ca, (const float2*)sa, (float2*)g_pre, HW, G, nx};
    if (drop_p > 0.f)
        DCS_LAUNCH(att_bwd_sa_kernel<true>, grid, dim3(kThreads), 0, dcs_stream(stream), sp, drop_p, (uint64_t)seed,
                           (const uint64_t*)seed_dev);
    else
        DCS_LAUNCH(att_bwd_sa_kernel<false>, grid, dim3(kThreads), 0, dcs_stream(stream), sp, drop_p, (uint64_t)seed,
                           (const uint64_t*)seed_dev);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

#ifndef DCS_ACT_BF16
extern "C" long dcs_attention_bwd_workspace_bytes(int B, long HW, int C, int Ch) {
    int G;
    if (B <= 0 || HW <= 0 || Ch <= 0 || !att_geom(C, &G)) return -1;
    return (long)B * chunks_for(HW, G) * C * 2 * (long)sizeof(double) + (2L * B * C + (long)B * Ch) * 8 + 64;
}
#endif

extern "C" int DCS_SYM(dcs_attention_bwd_x)(const act_t* x, const act_t* g_out, const float* ca, const float* sa,
                                   const float* g_sp, const float* pooled, const float* hidden, const float* w1,
                                   const float* w2, act_t* g_x, float* g_fc0_r, float* g_fc0_i, float* g_fc2_r,
                                   float* g_fc2_i, float* g_pooled, void* workspace, long workspace_bytes, int B, long HW,
                                   int C, int Ch, float drop_p, unsigned long long seed,
                                   const unsigned long long* seed_dev, dcs_stream_t stream) {
    int G;
    // all four FC destinations null (only with g_pooled): the FC weight gradient is left to dcs_attention_bwd_fc_weights, which
    // reads the per-sample cotangents this call leaves in `workspace`
    const bool defer_fc = g_pooled && !g_fc0_r && !g_fc0_i && !g_fc2_r && !g_fc2_i;
    if (!x || !g_out || !ca || !sa || !g_sp || !pooled || !hidden || !w1 || !w2 || !g_x || !workspace) return DCS_ERR_BADARG;
    if (!defer_fc && (!g_fc0_r || !g_fc0_i || !g_fc2_r || !g_fc2_i)) return DCS_ERR_BADARG;
    if (B <= 0 || B > 65535 || HW <= 0 || Ch <= 0 || Ch > 64 || C > 128 || !att_geom(C, &G)) return DCS_ERR_BADARG;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return DCS_ERR_BADARG;
    const int nch = chunks_for(HW, G);
    const long part_bytes = (long)B * nch * C * 2 * (long)sizeof(double);
    if (workspace_bytes < part_bytes + (2L * B * C + (long)B * Ch) * 8 + 64) return DCS_ERR_WORKSPACE;
    double* part = (double*)workspace;
    float2* go = (float2*)((char*)workspace + part_bytes);
    float2* gh = go + (long)B * C;
    // g_pooled given: the caller's consumer adds g_pooled / HW to g_x (dcs_cbn_bwd_add), saving the read-modify-write pass
    float2* gpooled = g_pooled ? (float2*)g_pooled : gh + (long)B * Ch;
    hipStream_t s = dcs_stream(stream);
    dim3 grid(nch, B);
    const BwdXP xp{x, g_out, ca, (const float2*)sa, (const float4*)g_sp, g_x, part, HW, C, G, nch};
    if (drop_p > 0.f)
        DCS_LAUNCH(att_bwd_x_kernel<true>, grid, dim3(kThreads), 0, s, xp, drop_p, (uint64_t)seed, (const uint64_t*)seed_dev);
    else
        DCS_LAUNCH(att_bwd_x_kernel<false>, grid, dim3(kThreads), 0, s, xp, drop_p, (uint64_t)seed, (const uint64_t*)seed_dev);
    DCS_CHECK_LAUNCH();
    const CaBwdP cp{(const double*)part, nch, (const float2*)ca, (const float2*)hidden, (const float2*)w1, (const float2*)w2, go, gh,
                    gpooled, C, Ch};
    DCS_LAUNCH(ca_bwd_sample_kernel, dim3(B), dim3(kThreads), 0, s, cp);
    DCS_CHECK_LAUNCH();
    if (defer_fc) return DCS_OK;
    CaWeightArgs cw;
    cw.go = go; cw.gh = gh; cw.pooled = (const float2*)pooled; cw.hidden = (const float2*)hidden;
    cw.g_fc0_r = g_fc0_r; cw.g_fc0_i = g_fc0_i; cw.g_fc2_r = g_fc2_r; cw.g_fc2_i = g_fc2_i;
    cw.B = B; cw.C = C; cw.Ch = Ch;
    const int nx_pool = g_pooled ? 0 : stream_grid(HW, G, B);
    const int nxw = nx_pool + (C * Ch * kWLanes + kThreads - 1) / kThreads;
    const PoolP pp{g_x, (const float*)gpooled, HW, G, 1.f / (float)HW, nx_pool, nxw, cw};
    DCS_LAUNCH(att_bwd_pool_kernel, dim3(nxw, g_pooled ? 1 : B), dim3(kThreads), 0, s, pp);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// The FC weight gradients of a block whose dcs_attention_bwd_x call deferred them (null destinations): the launch nothing downstream
// waits for, so a two-stream step can queue it beside the data-gradient chain.  `workspace`: the SAME buffer that call was given.
#ifndef DCS_ACT_BF16
extern "C" int dcs_attention_bwd_fc_weights(const void* workspace, long workspace_bytes, const float* pooled, const float* hidden,
                                            float* g_fc0_r, float* g_fc0_i, float* g_fc2_r, float* g_fc2_i, int B, long HW, int C,
                                            int Ch, dcs_stream_t stream) {
    int G;
    if (!workspace || !pooled || !hidden || !g_fc0_r || !g_fc0_i || !g_fc2_r || !g_fc2_i) return DCS_ERR_BADARG;
    if (B <= 0 || B > 65535 || HW <= 0 || Ch <= 0 || Ch > 64 || C > 128 || !att_geom(C, &G)) return DCS_ERR_BADARG;
    const long part_bytes = (long)B * chunks_for(HW, G) * C * 2 * (long)sizeof(double);
    if (workspace_bytes < part_bytes + (2L * B * C + (long)B * Ch) * 8 + 64) return DCS_ERR_WORKSPACE;
    const float2* go = (const float2*)((const char*)workspace + part_bytes);
    CaWeightArgs cw;
    cw.go = go; cw.gh = go + (long)B * C; cw.pooled = (const float2*)pooled; cw.hidden = (const float2*)hidden;
    cw.g_fc0_r = g_fc0_r; cw.g_fc0_i = g_fc0_i; cw.g_fc2_r = g_fc2_r; cw.g_fc2_i = g_fc2_i;
    cw.B = B; cw.C = C; cw.Ch = Ch;
    const int nxw = (C * Ch * kWLanes + kThreads - 1) / kThreads;
    const PoolP pp{nullptr, nullptr, HW, G, 1.f / (float)HW, 0, nxw, cw};
    DCS_LAUNCH(att_bwd_pool_kernel, dim3(nxw, 1), dim3(kThreads), 0, dcs_stream(stream), pp);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
#endif

// ---- several attention blocks in one set of launches (backward; see dcs_attention_fwd_batched) ----------------
static long bwd_item_bytes(const dcs_attention_item& it, int B, int G) {
    const long HW = (long)it.H * it.W;
    const long b = (long)B * chunks_for(HW, G) * it.C * 2 * (long)sizeof(double) + (2L * B * it.C + (long)B * it.Ch) * 8 + 64;
    return (b + 255) / 256 * 256;
}

static long bwd_batched_workspace_bytes(int n, const dcs_attention_item* items, int B) {
    if (n < 1 || n > kMaxBatch || !items || B <= 0) return -1;
    long total = 0;
    for (int i = 0; i < n; ++i) {
        int G;
        if (!att_geom(items[i].C, &G) || items[i].Ch <= 0) return -1;
        total += bwd_item_bytes(items[i], B, G);
    }
    return total;
}

#ifndef DCS_ACT_BF16
extern "C" long dcs_attention_bwd_batched_workspace_bytes(int n, const dcs_attention_item* items, int B) {
    return bwd_batched_workspace_bytes(n, items, B);
}
#endif

// (the _h form: x, g_out and g_x of every item are bf16 tensors, all the small maps fp32)
extern "C" int DCS_SYM(dcs_attention_bwd_batched)(int n, const dcs_attention_item* items, void* workspace, long workspace_bytes, int B,
                                         dcs_stream_t stream) {
    if (n < 1 || n > kMaxBatch || !items || !workspace || B <= 0 || B > 65535) return DCS_ERR_BADARG;
    if (workspace_bytes < bwd_batched_workspace_bytes(n, items, B)) return DCS_ERR_WORKSPACE;
    Tbl<BwdSaP> tsa; Tbl<BwdXP> tx; Tbl<CaBwdP> tc; Tbl<PoolP> tp;
    conv::Args dg[kMaxBatch];
    int nx_sa = 1, nx_x = 1, nx_p = 1;
    char* ws = (char*)workspace;
    for (int i = 0; i < n; ++i) {
        const dcs_attention_item& it = items[i];
        int G;
        if (!it.x || !it.w1 || !it.w2 || !it.ca || !it.pooled || !it.hidden || !it.sa || !it.g_out || !it.wsa_bwd || !it.g_pre ||
            !it.g_sp || !it.g_x || !it.g_fc0_r || !it.g_fc0_i || !it.g_fc2_r || !it.g_fc2_i || it.H <= 0 || it.W <= 0 ||
            it.Ch <= 0 || it.Ch > 64 || it.C > 128 || !att_geom(it.C, &G))
            return DCS_ERR_BADARG;
        const long HW = (long)it.H * it.W;
        const int nch = chunks_for(HW, G), nxs = stream_grid(HW, G, B);
        double* part = (double*)ws;
        float2* go = (float2*)(ws + (long)B * nch * it.C * 2 * (long)sizeof(double));
        float2* gh = go + (long)B * it.C;
        // g_pooled given: the consumer of g_x adds the pool's broadcast term (dcs_cbn_bwd_add), the read-modify-write pass is skipped
        float2* gpooled = it.g_pooled ? (float2*)it.g_pooled : gh + (long)B * it.Ch;
        const int nxp = it.g_pooled ? 0 : nxs;
        ws += bwd_item_bytes(it, B, G);
        tsa.p[i] = BwdSaP{(const act_t*)it.x, (const act_t*)it.g_out, it.ca, (const float2*)it.sa, (float2*)it.g_pre, HW, G, nxs};
        tx.p[i] = BwdXP{(const act_t*)it.x, (const act_t*)it.g_out, it.ca, (const float2*)it.sa, (const float4*)it.g_sp, (act_t*)it.g_x, part, HW, it.C, G, nch};
        tc.p[i] = CaBwdP{(const double*)part, nch, (const float2*)it.ca, (const float2*)it.hidden, (const float2*)it.w1,
                         (const float2*)it.w2, go, gh, gpooled, it.C, it.Ch};
        CaWeightArgs cw;
        cw.go = go; cw.gh = gh; cw.pooled = (const float2*)it.pooled; cw.hidden = (const float2*)it.hidden;
        cw.g_fc0_r = it.g_fc0_r; cw.g_fc0_i = it.g_fc0_i; cw.g_fc2_r = it.g_fc2_r; cw.g_fc2_i = it.g_fc2_i;
        cw.B = B; cw.C = it.C; cw.Ch = it.Ch;
        const int nxw = nxp + (it.C * it.Ch * kWLanes + kThreads - 1) / kThreads;
        tp.p[i] = PoolP{(act_t*)it.g_x, (const float*)gpooled, HW, G, 1.f / (float)HW, nxp, nxw, cw};
        nx_sa = nxs > nx_sa ? nxs : nx_sa;
        nx_x = nch > nx_x ? nch : nx_x;
        nx_p = nxw > nx_p ? nxw : nx_p;
        conv::Args& a = dg[i];                            // g_sp = data gradient of the 7x7 2->1 conv: 1 -> 2 over g_pre
        a = conv::Args{};
        a.x1 = (const act2_t*)it.g_pre;    /* (fp32 maps in either build) */ a.x2 = nullptr; a.wp = (const float2*)it.wsa_bwd; a.bias = nullptr;
        a.y = (act2_t*)it.g_sp;
        a.B = B; a.Hin = it.H; a.Win = it.W; a.C1 = 1; a.C2 = 0; a.up_f = 1; a.up_t = 1; a.zero_ins = 0; a.Cout = 2;
        a.kh = 7; a.kw = 7; a.sf = 1; a.st = 1; a.pad_f = 3; a.pad_t = 3; a.act = DCS_ACT_NONE;
        a.Hv = it.H; a.Wv = it.W; a.Hout = it.H; a.Wout = it.W;
    }
    hipStream_t s = dcs_stream(stream);
    tbl_compact(tsa, n, &nx_sa); tbl_compact(tx, n, &nx_x); tbl_compact(tp, n, &nx_p);
    DCS_LAUNCH(att_bwd_sa_multi_kernel, dim3(nx_sa, B, 1), dim3(kThreads), 0, s, tsa);
    DCS_CHECK_LAUNCH();
    const int rc = dcs_conv_direct_multi(dg, n, s);
    if (rc != DCS_OK) return rc;
    DCS_LAUNCH(att_bwd_x_multi_kernel, dim3(nx_x, B, 1), dim3(kThreads), 0, s, tx);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(ca_bwd_sample_multi_kernel, dim3(B, 1, n), dim3(kThreads), 0, s, tc);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(att_bwd_pool_multi_kernel, dim3(nx_p, B, 1), dim3(kThreads), 0, s, tp);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
