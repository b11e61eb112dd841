// adam.hip — one fused launch for the optimizer step of the reference's training recipe:
// gradient all-reduce averaging, clip-by-global-norm (Trainer gradient_clip_val=100, "norm":
// config.py:48-49, train.py:145-146) and Adam with L2 weight decay and AMSGrad
// (c_network.py:229-234; lr 1e-4, eps 1e-6, wd 1e-4: config.py:31,44-47), over ONE flat fp32
// parameter bucket (2.9 M floats).  torch.optim.Adam semantics:
//     g = scale * g ;  g *= min(1, max_norm / (||g|| + 1e-6)) ;  g += wd * p
//     m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  vmax = max(vmax, v)
//     p -= lr / (1-b1^t) * m / (sqrt(vmax) / sqrt(1-b2^t) + eps)
// HBM-bound: 5 streams read, 4 written = 36 B per parameter; the global norm comes from the caller
// as a DEVICE scalar (no host synchronisation in the step).
#include "dcs_common.h"

namespace {
constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void adam_amsgrad_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                                 float4* __restrict__ m, float4* __restrict__ v,
                                                                 float4* __restrict__ vmax,
                                                                 const float* __restrict__ grad_norm, float max_norm,
                                                                 float grad_scale, long n4, long n, float lr, float b1,
                                                                 float b2, float eps, float wd, float bc1, float bc2s,
                                                                 const int* __restrict__ step_dev,
                                                                 const float* __restrict__ skip,
                                                                 const double* __restrict__ sumsq_parts, int n_parts) {
    if (skip && skip[0] != 0.f) return;   // NaN-loss guard (c_network.py:257-261): the whole grid takes the same branch
    // ||g||^2 handed over as the partial sums of grad_sumsq_parts_kernel: every workgroup adds them up itself, in one fixed
    // order (n_parts <= 1024: four loads per thread, a wave butterfly, four LDS slots) — no second reduction launch
    __shared__ double wsum[kThreads / 64];
    float norm_parts = 0.f;
    if (n_parts > 0) {
        double a = 0.0;
        for (int i = threadIdx.x; i < n_parts; i += kThreads) a += sumsq_parts[i];
        a = dcs_wave_sum_d(a);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = a;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) tot += wsum[w];
        norm_parts = (float)sqrt(tot);
    }
    if (step_dev) {                       // update count kept on the device (hipGraph replay safe)
        const float t = (float)step_dev[0];
        bc1 = 1.f - powf(b1, t);
        bc2s = sqrtf(1.f - powf(b2, t));
    }
    float clip = grad_scale;
    if ((grad_norm != nullptr || n_parts > 0) && max_norm > 0.f) {
        const float c = max_norm / (grad_scale * (n_parts > 0 ? norm_parts : grad_norm[0]) + 1e-6f);
        clip *= c < 1.f ? c : 1.f;
    }
    const float step = lr / bc1;
    auto upd = [&](float& pp, float gg, float& mm, float& vv, float& vm) {
        gg = fmaf(wd, pp, gg * clip);
        mm = fmaf(b1, mm, (1.f - b1) * gg);
        vv = fmaf(b2, vv, (1.f - b2) * gg * gg);
        vm = fmaxf(vm, vv);
        pp -= step * mm / (sqrtf(vm) / bc2s + eps);
    };
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kThreads) {
        float4 pp = p[i], mm = m[i], vv = v[i], vm = vmax[i];
        const float4 gg = g[i];
        upd(pp.x, gg.x, mm.x, vv.x, vm.x);
        upd(pp.y, gg.y, mm.y, vv.y, vm.y);
        upd(pp.z, gg.z, mm.z, vv.z, vm.z);
        upd(pp.w, gg.w, mm.w, vv.w, vm.w);
        p[i] = pp; m[i] = mm; v[i] = vv; vmax[i] = vm;
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0) {
        float* ps = reinterpret_cast<float*>(p);
        const float* gs = reinterpret_cast<const float*>(g);
        float* ms = reinterpret_cast<float*>(m);
        float* vs = reinterpret_cast<float*>(v);
        float* vx = reinterpret_cast<float*>(vmax);
        for (long i = n4 * 4 + threadIdx.x; i < n; i += kThreads) upd(ps[i], gs[i], ms[i], vs[i], vx[i]);
    }
}
}  // namespace

static int adam_launch(float* p, const float* g, float* m, float* v, float* vmax, const float* grad_norm, const double* sumsq_parts,
                       int n_parts, float max_norm, float grad_scale, long n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, int step, const int* step_dev, const float* skip, dcs_stream_t stream) {
    if (!p || !g || !m || !v || !vmax || n <= 0 || (step < 1 && !step_dev)) return DCS_ERR_BADARG;
    if (n_parts < 0 || n_parts > 1024 || (n_parts > 0 && !sumsq_parts)) return DCS_ERR_BADARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vmax) & 15) return DCS_ERR_BADARG;
    const float bc1 = 1.f - powf(beta1, (float)(step < 1 ? 1 : step));
    const float bc2s = sqrtf(1.f - powf(beta2, (float)(step < 1 ? 1 : step)));
    const long n4 = n / 4;
    long nb = (n4 + kThreads * 2 - 1) / (kThreads * 2);
    const int grid = (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
    DCS_LAUNCH(adam_amsgrad_kernel, dim3(grid), dim3(kThreads), 0, dcs_stream(stream), (float4*)p,
                       (const float4*)g, (float4*)m, (float4*)v, (float4*)vmax, grad_norm, max_norm, grad_scale, n4, n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2s, step_dev, skip, sumsq_parts, n_parts);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, const float* grad_norm,
                                     float max_norm, float grad_scale, long n, float lr, float beta1, float beta2,
                                     float eps, float weight_decay, int step, const int* step_dev, const float* skip,
                                     dcs_stream_t stream) {
    return adam_launch(p, g, m, v, vmax, grad_norm, nullptr, 0, max_norm, grad_scale, n, lr, beta1, beta2, eps, weight_decay, step,
                       step_dev, skip, stream);
}

extern "C" int dcs_adam_amsgrad_step_sumsq(float* p, const float* g, float* m, float* v, float* vmax, const double* sumsq_parts,
                                           int n_parts, float max_norm, float grad_scale, long n, float lr, float beta1,
                                           float beta2, float eps, float weight_decay, int step, const int* step_dev,
                                           const float* skip, dcs_stream_t stream) {
    if (n_parts < 1) return DCS_ERR_BADARG;
    return adam_launch(p, g, m, v, vmax, nullptr, sumsq_parts, n_parts, max_norm, grad_scale, n, lr, beta1, beta2, eps, weight_decay,
                       step, step_dev, skip, stream);
}

namespace {
__global__ void step_guard_kernel(const float* __restrict__ loss, float* __restrict__ skip) {
    skip[0] = (loss[0] != loss[0]) ? 1.f : 0.f;
}
__global__ void step_advance_kernel(const float* __restrict__ skip, int* __restrict__ step_dev,
                                    long long* __restrict__ seed_dev, long long* __restrict__ counters, int n_counters) {
    if (step_dev && !(skip && skip[0] != 0.f)) step_dev[0] += 1;
    if (seed_dev) seed_dev[0] += 1;
    for (int i = 0; i < n_counters; ++i) counters[i] += 1;      // every forward counts, whatever the guard says (complexLayers: num_batches_tracked)
}
}  // namespace

namespace {
// parts[b] = sum of g^2 over workgroup b's grid-stride share (fp32 per thread over <= ~6 float4, fp64 from there on, fixed
// order); workgroup 0 also advances the step's device counters (step_advance_kernel's work: one launch instead of three —
// ATen's memset + reduction for the norm, and the counter launch)
__global__ __launch_bounds__(kThreads) void grad_sumsq_parts_kernel(const float4* __restrict__ g, long n4, long n,
                                                                     double* __restrict__ parts, const float* __restrict__ skip,
                                                                     int* __restrict__ step_dev, long long* __restrict__ seed_dev,
                                                                     long long* __restrict__ counters, int n_counters) {
    __shared__ double wsum[kThreads / 64];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (step_dev && !(skip && skip[0] != 0.f)) step_dev[0] += 1;
        if (seed_dev) seed_dev[0] += 1;
        for (int i = 0; i < n_counters; ++i) counters[i] += 1;
    }
    double a = 0.0;
    for (long i0 = (long)blockIdx.x * kThreads + threadIdx.x; i0 < n4; i0 += (long)gridDim.x * kThreads * 4) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                   // four loads in flight per trip (clamped index, masked value)
            const long i = i0 + (long)u * gridDim.x * kThreads;
            x[u] = g[i < n4 ? i : n4 - 1];
            if (i >= n4) x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) s = fmaf(x[u].x, x[u].x, fmaf(x[u].y, x[u].y, fmaf(x[u].z, x[u].z, fmaf(x[u].w, x[u].w, s))));
        a += (double)s;
    }
    if (blockIdx.x == 0) {
        const float* gs = reinterpret_cast<const float*>(g);
        for (long i = n4 * 4 + threadIdx.x; i < n; i += kThreads) a += (double)gs[i] * (double)gs[i];
    }
    a = dcs_wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) tot += wsum[w];
        parts[blockIdx.x] = tot;
    }
}
}  // namespace

extern "C" int dcs_grad_sumsq_parts(const float* g, long n, double* parts, int n_parts, const float* skip, int* step_dev,
                                    long long* seed_dev, long long* counters, int n_counters, dcs_stream_t stream) {
    if (!g || !parts || n < 4 || n_parts < 1 || n_parts > 1024 || ((uintptr_t)g & 15)) return DCS_ERR_BADARG;
    if (n_counters < 0 || n_counters > 4096 || (n_counters > 0 && !counters)) return DCS_ERR_BADARG;
    DCS_LAUNCH(grad_sumsq_parts_kernel, dim3(n_parts), dim3(kThreads), 0, dcs_stream(stream), (const float4*)g, n / 4, n, parts, skip,
               step_dev, seed_dev, counters, n_counters);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_step_guard(const float* loss, float* skip, dcs_stream_t stream) {
    if (!loss || !skip) return DCS_ERR_BADARG;
    DCS_LAUNCH(step_guard_kernel, dim3(1), dim3(1), 0, dcs_stream(stream), loss, skip);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_step_advance_counters(const float* skip, int* step_dev, long long* seed_dev, long long* counters, int n_counters,
                                         dcs_stream_t stream) {
    if ((!step_dev && !seed_dev && !counters) || n_counters < 0 || n_counters > 4096 || (n_counters > 0 && !counters)) return DCS_ERR_BADARG;
    DCS_LAUNCH(step_advance_kernel, dim3(1), dim3(1), 0, dcs_stream(stream), skip, step_dev, seed_dev, counters, n_counters);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_step_advance(const float* skip, int* step_dev, long long* seed_dev, dcs_stream_t stream) {
    if (!step_dev && !seed_dev) return DCS_ERR_BADARG;
    return dcs_step_advance_counters(skip, step_dev, seed_dev, nullptr, 0, stream);
}
