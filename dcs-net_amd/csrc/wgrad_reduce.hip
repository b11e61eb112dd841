// wgrad_reduce.hip — slab reductions of the weight-gradient kernels: single launch, or deferred and batched.
#include "wgrad_reduce.h"

#include <mutex>
#include <vector>

namespace wreduce {
namespace {

// plain slabs: 256 threads = 32 elements x 8 slab groups (coalesced 256-B rows per slab, 8 slabs in flight), LDS combine
__device__ __forceinline__ void store_plain(const Job& q, long j, float sr, float si) {
    const int co = (int)(j % q.Cout);
    const int ci = (int)((j / q.Cout) % q.Cin);
    const int tap = (int)(j / ((long)q.Cout * q.Cin));
    const int dy = tap / q.kw, dx = tap % q.kw;
    long dst;
    if (q.transposed) dst = (((long)ci * q.Cout + co) * q.kh + (q.kh - 1 - dy)) * q.kw + (q.kw - 1 - dx);
    else              dst = (((long)co * q.Cin + ci) * q.kh + dy) * q.kw + dx;
    q.gw_r[dst] = sr;
    q.gw_i[dst] = si;
}

// weights with an even element count: 16-byte loads (two complex elements per thread, a wave reads 1 KB contiguous per
// slab, 8 loads in flight), fixed slab order.  A problem with few elements and many slabs (the 7x7 small-channel convs:
// 196 element pairs x 1024 slabs) is latency-bound on one thread's serial slab loop (measured 50 us for 3 MB), so G
// threads split the slabs of an element pair (s = g, g + G, ..) and are combined through LDS in a fixed order.
// The 32-element x 8-slab-group form below (kept for the bias rows and odd counts) moves 256-byte segments.
__host__ __device__ __forceinline__ bool plain_vec(const Job& q) { return (((long)q.kh * q.kw * q.Cin * q.Cout) & 1) == 0; }
__host__ __device__ __forceinline__ int plain_groups(const Job& q) {
    const long pairs = (long)q.kh * q.kw * q.Cin * q.Cout / 2;
    int G = 1;
    while (G < 32 && q.n_slabs >= 16 * G && pairs * G * 2 <= 32768) G *= 2;
    return G;
}
__host__ __device__ __forceinline__ long plain_wblocks(const Job& q) {
    const long n = (long)q.kh * q.kw * q.Cin * q.Cout;
    const int E = 256 / plain_groups(q);
    return plain_vec(q) ? (n / 2 + E - 1) / E : (n + 31) / 32;
}

__device__ __forceinline__ void reduce_plain(const Job& q, int bid, float4* red4) {
    float2* red = reinterpret_cast<float2*>(red4);
    const long n = (long)q.kh * q.kw * q.Cin * q.Cout;
    const long nb = plain_wblocks(q);                       // blocks [0, nb): weights; [nb, ..): bias
    const bool is_bias = bid >= nb;
    if (!is_bias && plain_vec(q)) {
        const int G = plain_groups(q), E = 256 / G;          // (powers of two; uniform over the problem)
        const int e = threadIdx.x & (E - 1), g = threadIdx.x / E;
        const long j4 = (long)bid * E + e;
        const bool live = j4 * 2 < n;
        const long stride4 = n / 2;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            const float4* src = reinterpret_cast<const float4*>(q.slab_w) + j4;
#pragma unroll 8
            for (int s_ = g; s_ < q.n_slabs; s_ += G) {
                const float4 v = src[(long)s_ * stride4];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
        if (G > 1) {
            red4[threadIdx.x] = a;
            __syncthreads();
            if (g != 0) return;
            for (int k = 1; k < G; ++k) { const float4 v = red4[k * E + e]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        }
        if (!live) return;
        store_plain(q, 2 * j4, a.x, a.y);
        store_plain(q, 2 * j4 + 1, a.z, a.w);
        return;
    }
    const long lim = is_bias ? q.Cout : n;
    const int E = lim > 8 ? 32 : 8, G = 256 / E;             // 8 or 32 slab groups (a bias row of <= 8 channels: 32)
    const int e = threadIdx.x & (E - 1), sg = threadIdx.x / E;
    const long j = is_bias ? (long)(bid - nb) * E + e : (long)bid * E + e;
    const float2* src = is_bias ? q.slab_b : q.slab_w;
    float sr = 0.f, si = 0.f;
    if (j < lim) {
#pragma unroll 8
        for (int s = sg; s < q.n_slabs; s += G) { const float2 v = src[(long)s * lim + j]; sr += v.x; si += v.y; }
    }
    red[threadIdx.x] = make_float2(sr, si);
    __syncthreads();
    if (sg != 0 || j >= lim) return;
    for (int g = 1; g < G; ++g) { const float2 v = red[g * E + e]; sr += v.x; si += v.y; }
    if (is_bias) {
        q.gb_r[j] = sr + si;          // bias = (b_r - b_i) + j (b_r + b_i)
        q.gb_i[j] = si - sr;
        return;
    }
    store_plain(q, j, sr, si);
}

// destination tap index on one axis: which folded tap of residue class r contains original tap d
__device__ __forceinline__ int fold_index(int up, int r, int d) {
    if (up == 1) return d;
    return r == 0 ? (d == 0 ? 0 : 1) : (d == 2 ? 1 : 0);
}

// folded slabs -> 3x3 gradient: g_W[dy][dx] = sum over classes of g_Wfold_c[jy_c(dy)][jx_c(dx)]; one thread per element
// (slab groups per element as in reduce_plain: the 3x3 decoder convs with few channels have few elements and many slabs)
__host__ __device__ __forceinline__ int folded_groups(const Job& q) {
    const long n = 9L * q.Cin * q.Cout;
    int G = 1;
    while (G < 32 && q.n_slabs >= 16 * G && n * G * 2 <= 65536) G *= 2;
    return G;
}

__device__ __forceinline__ void reduce_folded(const Job& q, int bid, float2* red) {
    const int up_f = q.up_f, up_t = q.up_t;
    const int kh_c = up_f == 2 ? 2 : 3, kw_c = up_t == 2 ? 2 : 3, ncls = up_f * up_t;
    const long per = (long)q.Cin * q.Cout, n = 9 * per, wsz_c = (long)kh_c * kw_c * per;
    const int G = folded_groups(q), E = 256 / G;             // (powers of two; uniform over the problem)
    const int el = threadIdx.x & (E - 1), g = threadIdx.x / E;
    const long j = (long)bid * E + el;
    const bool is_w = j < n, is_b = !is_w && q.gb_r != nullptr && j < n + q.Cout;
    float sr = 0.f, si = 0.f;
    int dy = 0, dx = 0;
    long e = 0;
    if (is_w) {
        const int tap = (int)(j / per);
        e = j % per;
        dy = tap / 3; dx = tap % 3;
        for (int ry = 0; ry < up_f; ++ry)
            for (int rx = 0; rx < up_t; ++rx) {
                const int c = ry * up_t + rx;
                const long off = (long)c * wsz_c + (long)(fold_index(up_f, ry, dy) * kw_c + fold_index(up_t, rx, dx)) * per + e;
#pragma unroll 8
                for (int s = g; s < q.n_slabs; s += G) {                 // unrolled: the loads go out together
                    const float2 v = q.slab_w[(long)s * ncls * wsz_c + off];
                    sr += v.x; si += v.y;
                }
            }
    } else if (is_b) {
        const int co = (int)(j - n);
#pragma unroll 8
        for (int s = g; s < q.n_slabs * ncls; s += G) { const float2 v = q.slab_b[(long)s * q.Cout + co]; sr += v.x; si += v.y; }
    }
    if (G > 1) {
        red[threadIdx.x] = make_float2(sr, si);
        __syncthreads();
        if (g != 0) return;
        for (int k = 1; k < G; ++k) { const float2 v = red[k * E + el]; sr += v.x; si += v.y; }
    }
    if (is_w) {
        const int co = (int)(e % q.Cout), ci = (int)(e / q.Cout);
        long dst;
        if (q.transposed) dst = (((long)ci * q.Cout + co) * 3 + (2 - dy)) * 3 + (2 - dx);
        else              dst = (((long)co * q.Cin + ci) * 3 + dy) * 3 + dx;
        q.gw_r[dst] = sr;
        q.gw_i[dst] = si;
    } else if (is_b) {
        const int co = (int)(j - n);
        q.gb_r[co] = sr + si;
        q.gb_i[co] = si - sr;
    }
}

constexpr int kBatch = 24;                             // jobs per batched launch (kernel-argument table, ~2.3 KB)
struct Table { int n; Job jobs[kBatch]; };

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(Job q) {
    __shared__ float4 red[256];
    if (q.up_f > 0) reduce_folded(q, blockIdx.x, reinterpret_cast<float2*>(red)); else reduce_plain(q, blockIdx.x, red);
}

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(Table t) {
    __shared__ float4 red[256];
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.jobs[k + 1].blk0) ++k;
    const Job& q = t.jobs[k];
    const int bid = blockIdx.x - q.blk0;
    if (bid >= q.nblk) return;
    if (q.up_f > 0) reduce_folded(q, bid, reinterpret_cast<float2*>(red)); else reduce_plain(q, bid, red);
}

int blocks_of(const Job& q) {
    if (q.up_f > 0) {
        const int E = 256 / folded_groups(q);
        return (int)((9L * q.Cin * q.Cout + (q.gb_r ? q.Cout : 0) + E - 1) / E);
    }
    const long nbw = plain_wblocks(q);
    return (int)(nbw + (q.gb_r ? (q.Cout + 31) / 32 : 0));
}

std::vector<Job>* g_deferred = nullptr;                // process-wide: autograd runs backward on its own thread
std::mutex g_mutex;
thread_local bool g_suspended = false;                 // this thread's next reductions run immediately (dcs_wgrad_defer_suspend)

}  // namespace

bool deferring() {
    std::lock_guard<std::mutex> lock(g_mutex);
    return g_deferred != nullptr && !g_suspended;
}

int emit(Job j, hipStream_t s) {
    j.blk0 = 0; j.nblk = blocks_of(j);
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (g_deferred && !g_suspended) { g_deferred->push_back(j); return DCS_OK; }
    }
    DCS_LAUNCH(wgrad_reduce_kernel, dim3(j.nblk), dim3(256), 0, s, j);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

}  // namespace wreduce

extern "C" int dcs_wgrad_defer_begin(void) {
    std::lock_guard<std::mutex> lock(wreduce::g_mutex);
    if (wreduce::g_deferred) return DCS_ERR_BADARG;
    wreduce::g_deferred = new std::vector<wreduce::Job>();
    return DCS_OK;
}

extern "C" int dcs_wgrad_defer_suspend(int suspended) {
    wreduce::g_suspended = suspended != 0;
    return DCS_OK;
}

extern "C" int dcs_wgrad_defer_flush(dcs_stream_t stream) {
    std::vector<wreduce::Job>* jobs;
    {
        std::lock_guard<std::mutex> lock(wreduce::g_mutex);
        jobs = wreduce::g_deferred;
        wreduce::g_deferred = nullptr;
    }
    if (!jobs) return DCS_ERR_BADARG;
    hipStream_t s = dcs_stream(stream);
    int rc = dcs_conv_wgrad_small_flush(s);               // recorded kernels first: the reduces read their slabs
    for (size_t i0 = 0; i0 < jobs->size() && rc == DCS_OK; i0 += wreduce::kBatch) {
        wreduce::Table t;
        t.n = 0;
        int nb = 0;
        for (size_t i = i0; i < jobs->size() && t.n < wreduce::kBatch; ++i) {
            wreduce::Job q = (*jobs)[i];
            q.blk0 = nb;
            nb += q.nblk;
            t.jobs[t.n++] = q;
        }
        DCS_LAUNCH(wreduce::wgrad_reduce_multi_kernel, dim3(nb), dim3(256), 0, s, t);
        if (hipGetLastError() != hipSuccess) rc = DCS_ERR_LAUNCH;
    }
    delete jobs;
    return rc;
}
