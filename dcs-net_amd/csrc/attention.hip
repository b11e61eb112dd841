// attention.hip — the CBAM-style complex attentions of the skip and decoder paths
// (c_network.py:53-84, applied at c_network.py:208-211 and :219-220), channels-last complex64.
//
// All HBM-bound.  The reference streams each activation ~8 times per attention pair (2 pools,
// 2 broadcasts-multiplies, mean, 2 maxes, cat, final multiply).  Here:
//   ca_pool_kernel        1 read  of x      -> per-sample channel sums (fp64 slabs, no atomics)
//   ca_fc_kernel          tiny: slab sum, 1x1 conv, CReLU, 1x1 conv, x2 ("max" == avg quirk), sigmoid
//   spatial_pool_kernel   1 read  of x      -> [mean_c, max_c] of ca*x per pixel (ca*x never stored)
//   (7x7 2->1 conv + sigmoid: conv_direct.hip)
//   attention_apply_kernel 1 read + 1 write -> sa * (ca * x), dropout fused
// A lane group of G = C/2 lanes owns one pixel (one float4 = 2 complex channels per lane), so
// per-pixel channel reductions are wavefront shuffles and every global access is a contiguous
// 16-byte-per-lane stream.
#include "conv_common.h"
#include "cbn_geom.h"

namespace {

constexpr int kThreads = 256;
#ifndef DCS_ATT_APP_IT
#define DCS_ATT_APP_IT 2          // row passes per streaming workgroup
#endif
#ifndef DCS_ATT_GRID_CAP
#define DCS_ATT_GRID_CAP 8192
#endif
constexpr int kMaxChunks = DCS_ATT_MAX_CHUNKS;

inline bool att_geom(int C, int* G) { return att::geom(C, G); }
inline int ca_chunks(long HW, int G) { return att::ca_chunks(HW, G); }

// part[b][chunk][C][2] (double): sum over this chunk's pixels of x[b][p][c]
__device__ __forceinline__ void ca_pool_kernel_body(const act_t* __restrict__ x, double* __restrict__ part,
                                                            long HW, int C, int G, int bx, int by, int gx) {
    __shared__ double red[kThreads * 4];
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G;
    const int b = by;
    const ActIn4<act_t> x4 = act_in4(x) + (long)b * HW * G;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (long r = (long)bx * rpi + r0; r < HW; r += (long)gx * rpi) {
        const float4 v = x4[r * G + g];
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
    red[t * 4 + 0] = s0; red[t * 4 + 1] = s1; red[t * 4 + 2] = s2; red[t * 4 + 3] = s3;
    __syncthreads();
    for (int o = t; o < G * 4; o += kThreads) {
        const int gg = o / 4, i = o % 4;
        double a = 0;
        for (int r = 0; r < rpi; ++r) a += red[(r * G + gg) * 4 + i];
        part[(((long)b * gx + bx) * C + 2 * gg) * 2 + i] = a;   // (c=2gg+(i>>1), ri=i&1)
    }
}

// one workgroup per sample
__device__ __forceinline__ void ca_fc_kernel_body(const double* __restrict__ part, int nchunks,
                                                          const float2* __restrict__ w1, const float2* __restrict__ w2,
                                                          float2* __restrict__ ca_out, float2* __restrict__ pooled_out,
                                                          float2* __restrict__ hidden_out, long HW, int C, int Ch, int bx, int by, int gx) {
    __shared__ float2 pooled[128];
    __shared__ float2 hid[64];
    __shared__ double red[kThreads * 2];
    const int b = bx, t = threadIdx.x;
    {   // chunk slabs -> channel sums: kThreads / C threads share a channel's (up to 64) chunks, fixed combine order
        const int nsl = kThreads / C, c = t % C, sl = t / C;             // C <= 128 is a power of two (att_geom)
        double sr = 0, si = 0;
        if (sl < nsl)
#pragma unroll 8
            for (int k = sl; k < nchunks; k += nsl) {                    // (unrolled: the slab loads go out together)
                const double* p = part + (((long)b * nchunks + k) * C + c) * 2;
                sr += p[0]; si += p[1];
            }
        red[t * 2] = sr; red[t * 2 + 1] = si;
        __syncthreads();
        if (t < C) {
            for (int q = 1; q < nsl; ++q) { sr += red[(q * C + t) * 2]; si += red[(q * C + t) * 2 + 1]; }
            const float2 m = make_float2((float)(sr / (double)HW), (float)(si / (double)HW));
            pooled[t] = m;
            pooled_out[(long)b * C + t] = m;
        }
    }
    __syncthreads();
    // hidden[h] = sum_c w1[c][h] pooled[c]: 32 lanes per hidden unit, each a strided share of the channels (a single
    // thread per unit walks C dependent-latency loads: ~10 us at C = 128)
    for (int h = t >> 5; h < Ch; h += kThreads / 32) {
        float ar = 0.f, ai = 0.f;
#pragma unroll 4
        for (int c = t & 31; c < C; c += 32) {
            const float2 w = w1[c * Ch + h], p = pooled[c];
            ar = fmaf(w.x, p.x, ar); ar = fmaf(-w.y, p.y, ar);
            ai = fmaf(w.x, p.y, ai); ai = fmaf(w.y, p.x, ai);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { ar += __shfl_xor(ar, o, 64); ai += __shfl_xor(ai, o, 64); }
        if ((t & 31) == 0) {
            hidden_out[(long)b * Ch + h] = make_float2(ar, ai);
            hid[h] = make_float2(ar > 0.f ? ar : 0.f, ai > 0.f ? ai : 0.f);
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += kThreads) {
        float ar = 0.f, ai = 0.f;
#pragma unroll 8
        for (int h = 0; h < Ch; ++h) {
            const float2 w = w2[h * C + c], p = hid[h];
            ar = fmaf(w.x, p.x, ar); ar = fmaf(-w.y, p.y, ar);
            ai = fmaf(w.x, p.y, ai); ai = fmaf(w.y, p.x, ai);
        }
        // avg branch + "max" branch (an average, network_functions.py:135-138) = o + o
        ar += ar; ai += ai;
        ca_out[(long)b * C + c] = make_float2(1.f / (1.f + expf(-ar)), 1.f / (1.f + expf(-ai)));
    }
}

// per-pixel reductions over the G lanes of a pixel: DPP inside a row of 16 lanes (dcs_common.h)
__device__ __forceinline__ float group_sum(float v, int G) { return dcs_group_sum(v, G); }
__device__ __forceinline__ float group_max(float v, int G) { return dcs_group_max(v, G); }

// pooled[b][p] = { mean_c z , max_c Re z + j max_c Im z },  z = ca[b][c] * x[b][p][c]
__device__ __forceinline__ void spatial_pool_kernel_body(const act_t* __restrict__ x,
                                                                 const float* __restrict__ ca,
                                                                 float4* __restrict__ pooled, long HW, int C, int G, int bx, int by, int gx) {
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G;
    const int b = by;
    const ActIn4<act_t> x4 = act_in4(x) + (long)b * HW * G;
    float4 a = make_float4(1.f, 0.f, 1.f, 0.f);
    if (ca) a = reinterpret_cast<const float4*>(ca)[(long)b * G + g];
    const float invC = 1.f / (float)C;
    // every lane of a group runs the same trip count, so the shuffles are convergent
    const long iters = (HW + (long)gx * rpi - 1) / ((long)gx * rpi);
    for (long k = 0; k < iters; ++k) {
        const long r = (k * gx + bx) * rpi + r0;
        const bool ok = r < HW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = x4[r * G + g];
        const float z0r = a.x * v.x - a.y * v.y, z0i = a.x * v.y + a.y * v.x;
        const float z1r = a.z * v.z - a.w * v.w, z1i = a.z * v.w + a.w * v.z;
        const float sr = group_sum(z0r + z1r, G), si = group_sum(z0i + z1i, G);
        const float mr = group_max(fmaxf(z0r, z1r), G), mi = group_max(fmaxf(z0i, z1i), G);
        if (ok && g == 0) pooled[(long)b * HW + r] = make_float4(sr * invC, si * invC, mr, mi);
    }
}

template <bool DROP>
__device__ __forceinline__ void attention_apply_kernel_body(const act_t* __restrict__ x,
                                                                    const float* __restrict__ ca,
                                                                    const float2* __restrict__ sa, act_t* __restrict__ y,
                                                                    long HW, int C, int G, float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev, int bx, int by, int gx) {
    if (seed_dev) seed += seed_dev[0];   // per-step device-side offset (graph replay safe)
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G;
    const int b = by;
    const long base = (long)b * HW * G;
    const ActIn4<act_t> x4 = act_in4(x) + base;
    const ActOut4<act_t> y4 = act_out4(y) + base;
    float4 a = make_float4(1.f, 0.f, 1.f, 0.f);
    if (ca) a = reinterpret_cast<const float4*>(ca)[(long)b * G + g];
    const float inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
    for (long r = (long)bx * rpi + r0; r < HW; r += (long)gx * rpi) {
        const float4 v = x4[r * G + g];
        float2 s = make_float2(1.f, 0.f);
        if (sa) s = sa[(long)b * HW + r];
        // (explicit fused forms: left to -ffp-contract the fp32 and the bf16-storage builds of this kernel may pick different
        // associations, and the bf16 build's output must equal the fp32 build's rounded once — tests/test_hip_bf16.py)
        const float z0r = fmaf(a.x, v.x, -(a.y * v.y)), z0i = fmaf(a.x, v.y, a.y * v.x);
        const float z1r = fmaf(a.z, v.z, -(a.w * v.w)), z1i = fmaf(a.z, v.w, a.w * v.z);
        float4 o;
        o.x = fmaf(s.x, z0r, -(s.y * z0i)); o.y = fmaf(s.x, z0i, s.y * z0r);
        o.z = fmaf(s.x, z1r, -(s.y * z1i)); o.w = fmaf(s.x, z1i, s.y * z1r);
        if (DROP) {
            const uint64_t e = (uint64_t)(base + r * G + g) * 4;
            o.x *= dcs_keep_scale(seed, e, drop_p, inv_keep);
            o.y *= dcs_keep_scale(seed, e + 1, drop_p, inv_keep);
            o.z *= dcs_keep_scale(seed, e + 2, drop_p, inv_keep);
            o.w *= dcs_keep_scale(seed, e + 3, drop_p, inv_keep);
        }
        y4[r * G + g] = o;
    }
}

// ---- launch forms: one problem per launch, or several problems (blockIdx.z) sharing one launch ---------------------
// The seven skip attentions of the network depend only on encoder outputs, and their backward pass only on the
// decoder's: run as 7 x 5 (x 6 backward) dependent launches of 5-10 us they cost more in launch boundaries than in
// work.  Every kernel therefore also exists in a table form: problem = blockIdx.z, its own grid width p.nx (blocks
// beyond it exit), same body.
constexpr int kMaxBatch = 8;
// x0[k]: first blockIdx.x of problem k in the COMPACTED grid (prefix sums of the problems' own grid widths; INT_MAX past
// the last one).  A grid as wide as the widest problem for every problem launched thousands of workgroups that only
// returned — at [32,256,256] the seven skip attentions' 7x7 conv took 40 us for 12.6 + 13.0 us of work.
template <class P> struct Tbl { P p[kMaxBatch]; int x0[kMaxBatch + 1]; };
template <class P> __device__ __forceinline__ int tbl_find(const Tbl<P>& t, int bx) {
    int z = 0;
#pragma unroll
    for (int k = 1; k < kMaxBatch; ++k) z += bx >= t.x0[k] ? 1 : 0;
    return z;
}

struct CaPoolP { const act_t* x; double* part; long HW; int C, G, nx; };
struct CaFcP { const double* part; int nchunks; const float2* w1; const float2* w2; float2* ca; float2* pooled; float2* hidden;
               long HW; int C, Ch; };
struct SpPoolP { const act_t* x; const float* ca; float4* pooled; long HW; int C, G, nx; };
struct ApplyP { const act_t* x; const float* ca; const float2* sa; act_t* y; long HW; int C, G, nx; };

__global__ __launch_bounds__(kThreads) void ca_pool_kernel(CaPoolP p) {
    ca_pool_kernel_body(p.x, p.part, p.HW, p.C, p.G, blockIdx.x, blockIdx.y, gridDim.x);
}
__global__ __launch_bounds__(kThreads) void ca_pool_multi_kernel(Tbl<CaPoolP> t) {
    const int z = tbl_find(t, blockIdx.x);
    const CaPoolP& p = t.p[z];
    ca_pool_kernel_body(p.x, p.part, p.HW, p.C, p.G, blockIdx.x - t.x0[z], blockIdx.y, p.nx);
}
__global__ __launch_bounds__(kThreads) void ca_fc_kernel(CaFcP p) {
    ca_fc_kernel_body(p.part, p.nchunks, p.w1, p.w2, p.ca, p.pooled, p.hidden, p.HW, p.C, p.Ch, blockIdx.x, 0, 0);
}
__global__ __launch_bounds__(kThreads) void ca_fc_multi_kernel(Tbl<CaFcP> t) {
    const CaFcP& p = t.p[blockIdx.z];
    ca_fc_kernel_body(p.part, p.nchunks, p.w1, p.w2, p.ca, p.pooled, p.hidden, p.HW, p.C, p.Ch, blockIdx.x, 0, 0);
}
__global__ __launch_bounds__(kThreads) void spatial_pool_kernel(SpPoolP p) {
    spatial_pool_kernel_body(p.x, p.ca, p.pooled, p.HW, p.C, p.G, blockIdx.x, blockIdx.y, gridDim.x);
}
__global__ __launch_bounds__(kThreads) void spatial_pool_multi_kernel(Tbl<SpPoolP> t) {
    const int z = tbl_find(t, blockIdx.x);
    const SpPoolP& p = t.p[z];
    spatial_pool_kernel_body(p.x, p.ca, p.pooled, p.HW, p.C, p.G, blockIdx.x - t.x0[z], blockIdx.y, p.nx);
}
template <bool DROP>
__global__ __launch_bounds__(kThreads) void attention_apply_kernel(ApplyP p, float drop_p, uint64_t seed, const uint64_t* seed_dev) {
    attention_apply_kernel_body<DROP>(p.x, p.ca, p.sa, p.y, p.HW, p.C, p.G, drop_p, seed, seed_dev, blockIdx.x, blockIdx.y, gridDim.x);
}
__global__ __launch_bounds__(kThreads) void attention_apply_multi_kernel(Tbl<ApplyP> t) {      // no dropout on the skip path
    const int z = tbl_find(t, blockIdx.x);
    const ApplyP& p = t.p[z];
    attention_apply_kernel_body<false>(p.x, p.ca, p.sa, p.y, p.HW, p.C, p.G, 0.f, 0, nullptr, blockIdx.x - t.x0[z], blockIdx.y, p.nx);
}

#ifndef DCS_ACT_BF16
__global__ __launch_bounds__(kThreads) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                            float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    if (seed_dev) seed += seed_dev[0];   // per-step device-side offset (graph replay safe)
    const float inv_keep = 1.f / (1.f - drop_p);
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads)
        y[i] = x[i] * (drop_p > 0.f ? dcs_keep_scale(seed, (uint64_t)i, drop_p, inv_keep) : 1.f);
}
#endif

inline int stream_grid(long HW, int G, int B) {
    const int rpi = kThreads / G;
    long it = (HW + rpi - 1) / rpi;
    long nb = it <= DCS_ATT_SMALL_IT ? (it + DCS_ATT_APP_IT - 1) / DCS_ATT_APP_IT : (it + 3) / 4;
    long cap = (it <= DCS_ATT_SMALL_IT ? DCS_ATT_GRID_CAP : 2048) / (B > 0 ? B : 1);
    if (cap < 1) cap = 1;
    return (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}

}  // namespace

#ifndef DCS_ACT_BF16
extern "C" long dcs_ca_workspace_bytes(int B, long HW, int C) {
    int G;
    if (B <= 0 || HW <= 0 || !att_geom(C, &G)) return -1;
    return (long)B * ca_chunks(HW, G) * C * 2 * (long)sizeof(double);
}
#endif

extern "C" int DCS_SYM(dcs_channel_attention_fwd)(const act_t* x, const float* w1, const float* w2, float* ca_out,
                                         float* pooled_out, float* hidden_out, void* workspace, long workspace_bytes,
                                         int B, long HW, int C, int Ch, dcs_stream_t stream) {
    int G;
    if (!x || !w1 || !w2 || !ca_out || !pooled_out || !hidden_out || !workspace) return DCS_ERR_BADARG;
    if (B <= 0 || B > 65535 || HW <= 0 || !att_geom(C, &G) || Ch <= 0 || Ch > 64) return DCS_ERR_BADARG;
    const int nch = ca_chunks(HW, G);
    if (workspace_bytes < (long)B * nch * C * 2 * (long)sizeof(double)) return DCS_ERR_WORKSPACE;
    hipStream_t s = dcs_stream(stream);
    const CaPoolP pp{x, (double*)workspace, HW, C, G, nch};
    DCS_LAUNCH(ca_pool_kernel, dim3(nch, B), dim3(kThreads), 0, s, pp);
    DCS_CHECK_LAUNCH();
    const CaFcP fp{(const double*)workspace, nch, (const float2*)w1, (const float2*)w2, (float2*)ca_out, (float2*)pooled_out,
                   (float2*)hidden_out, HW, C, Ch};
    DCS_LAUNCH(ca_fc_kernel, dim3(B), dim3(kThreads), 0, s, fp);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

#ifndef DCS_ACT_BF16
// The FC half of dcs_channel_attention_fwd alone, for callers whose producer already left the pooling slabs
// (dcs_cbn_fwd_slabs_pool: the CBN apply kernel pools its own output): part = double[B][chunks][C][2] with
// chunks = dcs_ca_pool_chunks(HW, C).  Activation-type independent (slabs and maps are fp64 / fp32).
extern "C" int dcs_ca_pool_chunks(long HW, int C) {
    int G;
    if (HW <= 0 || !att_geom(C, &G)) return -1;
    return ca_chunks(HW, G);
}

extern "C" int dcs_channel_attention_fc_fwd(const void* part, const float* w1, const float* w2, float* ca_out,
                                            float* pooled_out, float* hidden_out, int B, long HW, int C, int Ch,
                                            dcs_stream_t stream) {
    int G;
    if (!part || !w1 || !w2 || !ca_out || !pooled_out || !hidden_out) return DCS_ERR_BADARG;
    if (B <= 0 || B > 65535 || HW <= 0 || !att_geom(C, &G) || Ch <= 0 || Ch > 64) return DCS_ERR_BADARG;
    const CaFcP fp{(const double*)part, ca_chunks(HW, G), (const float2*)w1, (const float2*)w2, (float2*)ca_out,
                   (float2*)pooled_out, (float2*)hidden_out, HW, C, Ch};
    DCS_LAUNCH(ca_fc_kernel, dim3(B), dim3(kThreads), 0, dcs_stream(stream), fp);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
#endif

extern "C" int DCS_SYM(dcs_spatial_pool_fwd)(const act_t* x, const float* ca, float* pooled, int B, long HW, int C,
                                    dcs_stream_t stream) {
    int G;
    if (!x || !pooled || B <= 0 || B > 65535 || HW <= 0 || !att_geom(C, &G)) return DCS_ERR_BADARG;
    const int nx = stream_grid(HW, G, B);
    const SpPoolP sp{x, ca, (float4*)pooled, HW, C, G, nx};
    DCS_LAUNCH(spatial_pool_kernel, dim3(nx, B), dim3(kThreads), 0, dcs_stream(stream), sp);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int DCS_SYM(dcs_attention_apply_fwd)(const act_t* x, const float* ca, const float* sa, act_t* y, int B, long HW,
                                       int C, float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream) {
    int G;
    if (!x || !y || B <= 0 || B > 65535 || HW <= 0 || !att_geom(C, &G)) return DCS_ERR_BADARG;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return DCS_ERR_BADARG;
    const int nx = stream_grid(HW, G, B);
    dim3 grid(nx, B);
    const ApplyP ap{x, ca, (const float2*)sa, y, HW, C, G, nx};
    if (drop_p > 0.f)
        DCS_LAUNCH(attention_apply_kernel<true>, grid, dim3(kThreads), 0, dcs_stream(stream), ap, drop_p,
                           (uint64_t)seed, (const uint64_t*)seed_dev);
    else
        DCS_LAUNCH(attention_apply_kernel<false>, grid, dim3(kThreads), 0, dcs_stream(stream), ap, drop_p,
                           (uint64_t)seed, (const uint64_t*)seed_dev);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// ---- several attention blocks in one set of launches (forward) ---------------------------------------------
// items[i]: one block (the reference's skip_attention[2i], [2i+1] on encoder output i: c_network.py:208-211), no dropout.
// Five launches for all n blocks: channel pooling, channel FCs, spatial pooling, the 7x7 2->1 conv + sigmoid
// (conv_direct.hip), apply.  workspace: sum of the blocks' dcs_ca_workspace_bytes, each rounded up to 256 bytes.
static long fwd_batched_workspace_bytes(int n, const dcs_attention_item* items, int B) {
    if (n < 1 || n > kMaxBatch || !items || B <= 0) return -1;
    long total = 0;
    for (int i = 0; i < n; ++i) {
        int G;
        if (!att_geom(items[i].C, &G)) return -1;
        const long b = (long)B * ca_chunks((long)items[i].H * items[i].W, G) * items[i].C * 2 * (long)sizeof(double);
        total += (b + 255) / 256 * 256;
    }
    return total;
}

#ifndef DCS_ACT_BF16
extern "C" long dcs_attention_fwd_batched_workspace_bytes(int n, const dcs_attention_item* items, int B) {
    return fwd_batched_workspace_bytes(n, items, B);
}
#endif

// (the _h form: x and y of every item are bf16 tensors, all the small maps fp32)
extern "C" int DCS_SYM(dcs_attention_fwd_batched)(int n, const dcs_attention_item* items, void* workspace, long workspace_bytes, int B,
                                         dcs_stream_t stream) {
    if (n < 1 || n > kMaxBatch || !items || !workspace || B <= 0 || B > 65535) return DCS_ERR_BADARG;
    if (workspace_bytes < fwd_batched_workspace_bytes(n, items, B)) return DCS_ERR_WORKSPACE;
    Tbl<CaPoolP> tp; Tbl<CaFcP> tf; Tbl<SpPoolP> ts; Tbl<ApplyP> ta;
    conv::Args ca_[kMaxBatch];
    int nx_pool = 1, nx_stream = 1;
    char* ws = (char*)workspace;
    for (int i = 0; i < n; ++i) {
        const dcs_attention_item& it = items[i];
        int G;
        if (!it.x || !it.w1 || !it.w2 || !it.wsa || !it.sa_bias || !it.ca || !it.pooled || !it.hidden || !it.sp || !it.sa ||
            !it.y || it.H <= 0 || it.W <= 0 || it.Ch <= 0 || it.Ch > 64 || !att_geom(it.C, &G))
            return DCS_ERR_BADARG;
        const long HW = (long)it.H * it.W;
        const int nch = ca_chunks(HW, G), nxs = stream_grid(HW, G, B);
        tp.p[i] = CaPoolP{(const act_t*)it.x, (double*)ws, HW, it.C, G, nch};
        tf.p[i] = CaFcP{(const double*)ws, nch, (const float2*)it.w1, (const float2*)it.w2, (float2*)it.ca, (float2*)it.pooled,
                        (float2*)it.hidden, HW, it.C, it.Ch};
        ts.p[i] = SpPoolP{(const act_t*)it.x, it.ca, (float4*)it.sp, HW, it.C, G, nxs};
        ta.p[i] = ApplyP{(const act_t*)it.x, it.ca, (const float2*)it.sa, (act_t*)it.y, HW, it.C, G, nxs};
        ws += ((long)B * nch * it.C * 2 * (long)sizeof(double) + 255) / 256 * 256;
        nx_pool = nch > nx_pool ? nch : nx_pool;
        nx_stream = nxs > nx_stream ? nxs : nx_stream;
        conv::Args& a = ca_[i];                           // sa = sigmoid(conv7x7(sp)): 2 -> 1 channels, pad 3
        a = conv::Args{};
        a.x1 = (const act2_t*)it.sp;    /* (fp32 maps in either build: conv_k7.hip reads them as float2) */ a.x2 = nullptr; a.wp = (const float2*)it.wsa; a.bias = (const float2*)it.sa_bias;
        a.y = (act2_t*)it.sa;
        a.B = B; a.Hin = it.H; a.Win = it.W; a.C1 = 2; a.C2 = 0; a.up_f = 1; a.up_t = 1; a.zero_ins = 0; a.Cout = 1;
        a.kh = 7; a.kw = 7; a.sf = 1; a.st = 1; a.pad_f = 3; a.pad_t = 3; a.act = DCS_ACT_SIGMOID;
        a.Hv = it.H; a.Wv = it.W; a.Hout = it.H; a.Wout = it.W;
    }
    hipStream_t s = dcs_stream(stream);
    {   // compacted grids: problem k owns blockIdx.x in [x0[k], x0[k+1])
        int ap = 0, as_ = 0;
        for (int k = 0; k <= kMaxBatch; ++k) {
            tp.x0[k] = k < n ? ap : 0x7fffffff; ts.x0[k] = ta.x0[k] = k < n ? as_ : 0x7fffffff;
            if (k < n) { ap += tp.p[k].nx; as_ += ts.p[k].nx; }
        }
        tp.x0[n] = ap; ts.x0[n] = ta.x0[n] = as_;
        nx_pool = ap; nx_stream = as_;
    }
    DCS_LAUNCH(ca_pool_multi_kernel, dim3(nx_pool, B, 1), dim3(kThreads), 0, s, tp);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(ca_fc_multi_kernel, dim3(B, 1, n), dim3(kThreads), 0, s, tf);
    DCS_CHECK_LAUNCH();
    DCS_LAUNCH(spatial_pool_multi_kernel, dim3(nx_stream, B, 1), dim3(kThreads), 0, s, ts);
    DCS_CHECK_LAUNCH();
    const int rc = dcs_conv_direct_multi(ca_, n, s);
    if (rc != DCS_OK) return rc;
    DCS_LAUNCH(attention_apply_multi_kernel, dim3(nx_stream, B, 1), dim3(kThreads), 0, s, ta);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

#ifndef DCS_ACT_BF16
extern "C" int dcs_dropout_fwd(const float* x, float* y, long n, float drop_p, unsigned long long seed,
                               const unsigned long long* seed_dev, dcs_stream_t stream) {
    if (!x || !y || n <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return DCS_ERR_BADARG;
    long nb = (n + kThreads * 4 - 1) / (kThreads * 4);
    const int grid = (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
    DCS_LAUNCH(dropout_kernel, dim3(grid), dim3(kThreads), 0, dcs_stream(stream), x, y, n, drop_p,
                       (uint64_t)seed, (const uint64_t*)seed_dev);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
#endif
