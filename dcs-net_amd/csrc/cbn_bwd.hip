// cbn_bwd.hip — backward of dcs_cbn_fwd (ComplexBatchNorm2d + activation + dropout).
//
// The reference gets this from autograd through ~12 element-wise / reduction ops per CBN, each
// a full HBM round trip in backward as well.  Closed form, per channel (d = x - mu, y = A d + b',
// A = W R, g_y = g_out * act'(y) * dropout_mask, n = pixels):
//     sums    s = sum g_y (2),   N = sum g_y d^T (2x2)                               -- pass 1
//     g_bias = s ;  g_W from R N ;  g_R = sym(W N) ;  g_C = J_R(C)^T g_R  (closed-form Jacobian of
//     the inverse matrix square root) ;  D = (1/n) [[2 gCrr, gCri], [gCri, 2 gCii]]
//     g_x = A^T g_y + D d - A^T s / n                                                -- pass 2
// Two streaming passes (x and g_out read twice, g_x written once): 40 B per pixel-channel, and
// neither y nor the activation mask nor the dropout mask is ever stored — all three are
// recomputed bit-identically from x, the saved coefficients and (seed, index).
#include "dcs_common.h"
#include "cbn_geom.h"

namespace {

using cbn::kThreads;

template <int ACT>
__device__ __forceinline__ float dact(float y, float g) {
    if (ACT == DCS_ACT_RELU) return y > 0.f ? g : 0.f;
    if (ACT == DCS_ACT_LRELU) return y > 0.f ? g : 0.01f * g;
    return g;
}

struct Chan {   // forward coefficients + mean of one channel
    float a0, a1, a2, a3, c0, c1, mr, mi;
};

__device__ __forceinline__ Chan load_chan(const float* coef, const float* stats, int c) {
    Chan k;
    const float* co = coef + 6 * c;
    k.a0 = co[0]; k.a1 = co[1]; k.a2 = co[2]; k.a3 = co[3]; k.c0 = co[4]; k.c1 = co[5];
    k.mr = stats[8 * c]; k.mi = stats[8 * c + 1];
    return k;
}

// g_y of one complex element (recomputes y, applies act' and the dropout keep-scale)
template <int ACT, bool DROP>
__device__ __forceinline__ float2 grad_y(const Chan& k, float xr, float xi, float gr, float gi, uint64_t seed,
                                         uint64_t e, float p, float inv_keep) {
    const float yr = fmaf(k.a0, xr, fmaf(k.a1, xi, k.c0));
    const float yi = fmaf(k.a2, xr, fmaf(k.a3, xi, k.c1));
    if (DROP) {
        gr *= dcs_keep_scale(seed, e, p, inv_keep);
        gi *= dcs_keep_scale(seed, e + 1, p, inv_keep);
    }
    return make_float2(dact<ACT>(yr, gr), dact<ACT>(yi, gi));
}

// g_out + scale * (per-sample, per-channel constant): the broadcast half of an average pool's backward, folded into
// the consumer instead of a read-modify-write pass over g_out (dcs_cbn_bwd_add)
// sample index r / HW of pixel row r (the per-sample additive term): a float estimate corrected by one either way (the 64-bit
// division it replaces was ~100 instructions per row in two bandwidth-bound kernels).  Precondition: r < 2^31 and at most 2^20
// samples — the estimate carries ~2^-23 relative error, i.e. less than ONE in absolute terms only while r / HW < 2^22, which is
// what a single correction step can repair; the host entry rejects larger sample counts (P / HW > 2^20) when a per-sample
// term is present, rows beyond 2^31 take the exact division.
__device__ __forceinline__ long sample_of(long r, long HW, float inv_hw) {
    if (r >= (1L << 31)) return r / HW;
    const unsigned ru = (unsigned)r, hw = (unsigned)HW;
    unsigned b = (unsigned)((float)ru * inv_hw);
    if ((unsigned long long)b * hw > ru) --b;
    if ((unsigned long long)(b + 1) * hw <= ru) ++b;
    return (long)b;
}

__device__ __forceinline__ void add_sample(float4& g, const float4 a, float scale) {
    g.x = fmaf(a.x, scale, g.x); g.y = fmaf(a.y, scale, g.y); g.z = fmaf(a.z, scale, g.z); g.w = fmaf(a.w, scale, g.w);
}

__device__ __forceinline__ void acc6(float* s, float2 gy, float u, float v) {
    s[0] += gy.x; s[1] += gy.y;
    s[2] = fmaf(gy.x, u, s[2]); s[3] = fmaf(gy.x, v, s[3]);
    s[4] = fmaf(gy.y, u, s[4]); s[5] = fmaf(gy.y, v, s[5]);
}

// part[nblocks][C][6] (double): {sum gyr, sum gyi, N00, N01, N10, N11}
template <int ACT, bool DROP>
__global__ __launch_bounds__(kThreads) void cbn_bwd_reduce_kernel(const act_t* __restrict__ x,
                                                                   const act_t* __restrict__ go,
                                                                   const float* __restrict__ coef,
                                                                   const float* __restrict__ stats,
                                                                   double* __restrict__ part, long P, int C, int G,
                                                                   int rows_per_iter, float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                                   const float4* __restrict__ g_add, float add_scale, long HW,
                                                                   const act_t* __restrict__ g2_) {
    DCS_PRIO_CRITICAL();
    if (seed_dev) seed += seed_dev[0];   // per-step device-side offset (graph replay safe)
    __shared__ double red[kThreads * 12];
    const int t = threadIdx.x;
    const float inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
    const ActIn4<act_t> x4 = act_in4(x), g4 = act_in4(go), g2 = act_in4(g2_);
    float s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = 0.f;

    if (C == 1) {
        const Chan k = load_chan(coef, stats, 0);
        const long nvec = P / 2;
        for (long i = (long)blockIdx.x * kThreads + t; i < nvec; i += (long)gridDim.x * kThreads) {
            const float4 v = x4[i];
            float4 g = g4[i];
            if (g2_) { const float4 h = g2[i]; g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
            const uint64_t e = (uint64_t)i * 4;
            acc6(s, grad_y<ACT, DROP>(k, v.x, v.y, g.x, g.y, seed, e, drop_p, inv_keep), v.x - k.mr, v.y - k.mi);
            acc6(s, grad_y<ACT, DROP>(k, v.z, v.w, g.z, g.w, seed, e + 2, drop_p, inv_keep), v.z - k.mr, v.w - k.mi);
        }
        if ((P & 1) && blockIdx.x == 0 && t == 0) {
            const long q = 2 * (P - 1);
            const float xr = dcs_ld1(x + q), xi = dcs_ld1(x + q + 1);
            acc6(s, grad_y<ACT, DROP>(k, xr, xi, dcs_ld1(go + q), dcs_ld1(go + q + 1), seed, (uint64_t)q, drop_p, inv_keep),
                 xr - k.mr, xi - k.mi);
        }
        double d[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = dcs_wave_sum_d((double)s[i]);
        const int wave = t >> 6, lane = t & 63;
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) red[wave * 6 + i] = d[i];
        }
        __syncthreads();
        if (t < 6) {
            double a = 0;
            for (int w = 0; w < kThreads / 64; ++w) a += red[w * 6 + t];
            part[(long)blockIdx.x * 6 + t] = a;
        }
        return;
    }

    const int g = t % G, r0 = t / G;
    const float inv_hw = 1.f / (float)(HW > 0 ? HW : 1);
    const Chan k0 = load_chan(coef, stats, 2 * g), k1 = load_chan(coef, stats, 2 * g + 1);
    // Four row passes per trip, every load unconditional (clamped row; the optional second cotangent and per-sample term
    // read through a valid stand-in pointer and are dropped by a select): as a rolled loop with `if (g2)` / `if (g_add)`
    // around their loads every pass was its own memory round trip — up to eight in a row, 8-11 us for a 5-us kernel.
    constexpr int UN = 4;
    const long stride = (long)gridDim.x * rows_per_iter;
    const ActIn4<act_t> g2p = g2_ ? g2 : g4;
    const float4* gap = g_add ? g_add : reinterpret_cast<const float4*>(stats);      // (stand-in: element 0 of any fp32 array)
    for (long rb = (long)blockIdx.x * rows_per_iter + r0; rb < P; rb += stride * UN) {
        float4 v[UN], gg[UN], h[UN], ga[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long r = rb + u * stride, rc = r < P ? r : P - 1;
            v[u] = x4[rc * G + g];
            gg[u] = g4[rc * G + g];
            h[u] = g2p[rc * G + g];
            ga[u] = gap[g_add ? sample_of(rc, HW, inv_hw) * G + g : 0];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long r = rb + u * stride;
            if (r >= P) continue;
            if (g2_) { gg[u].x += h[u].x; gg[u].y += h[u].y; gg[u].z += h[u].z; gg[u].w += h[u].w; }
            if (g_add) add_sample(gg[u], ga[u], add_scale);
            const uint64_t e = (uint64_t)(r * G + g) * 4;
            acc6(s, grad_y<ACT, DROP>(k0, v[u].x, v[u].y, gg[u].x, gg[u].y, seed, e, drop_p, inv_keep), v[u].x - k0.mr, v[u].y - k0.mi);
            acc6(s + 6, grad_y<ACT, DROP>(k1, v[u].z, v[u].w, gg[u].z, gg[u].w, seed, e + 2, drop_p, inv_keep), v[u].z - k1.mr,
                 v[u].w - k1.mi);
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) red[t * 12 + i] = (double)s[i];
    __syncthreads();
    for (int o = t; o < G * 12; o += kThreads) {
        const int gg = o / 12, i = o % 12;
        double a = 0;
        for (int r = 0; r < rows_per_iter; ++r) a += red[(r * G + gg) * 12 + i];
        const int c = 2 * gg + (i >= 6);
        part[((long)blockIdx.x * C + c) * 6 + (i % 6)] = a;
    }
}

// bcoef[C][10]: g_x = [at0 at1; at2 at3] g_y + [d0 d1; d2 d3] x + (f0, f1)
__global__ void cbn_bwd_finalize_kernel(const double* __restrict__ part, int nblocks,
                                        const float* __restrict__ weight, const float* __restrict__ stats,
                                        const float* __restrict__ coef, float* __restrict__ g_weight,
                                        float* __restrict__ g_bias, float* __restrict__ bcoef, long P, int C,
                                        int use_batch_stats) {
    DCS_PRIO_CRITICAL();
    // one wavefront per channel: lanes stride over the partial slabs, fp64 butterfly, lane 0 finishes
    const int c = blockIdx.x, lane = threadIdx.x;
    // the channel's statistics, coefficients and weights are requested FIRST (every lane, uniform addresses): behind the slab
    // sums and the `lane != 0` exit they were a second dependent round trip of a kernel that is nothing but round trips
    const float* st = stats + 8 * c;
    const float st_[8] = {st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7]};
    const float* co = coef + 6 * c;
    const float co_[4] = {co[0], co[1], co[2], co[3]};
    float w_[3] = {1.f, 1.f, 0.f};
    if (weight) { w_[0] = weight[3 * c]; w_[1] = weight[3 * c + 1]; w_[2] = weight[3 * c + 2]; }
    // (all of a lane's slab loads in flight at once: see cbn_finalize_kernel)
    double S[6] = {0, 0, 0, 0, 0, 0};
    for (int b0 = lane; b0 < nblocks; b0 += 64 * 8) {
        double v[8][6];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + 64 * k < nblocks ? b0 + 64 * k : nblocks - 1;
#pragma unroll
            for (int i = 0; i < 6; ++i) v[k][i] = part[((long)b * C + c) * 6 + i];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 6; ++i) S[i] += b0 + 64 * k < nblocks ? v[k][i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) S[i] = dcs_wave_sum_d(S[i]);
    if (lane != 0) return;
    const double sgr = S[0], sgi = S[1], N00 = S[2], N01 = S[3], N10 = S[4], N11 = S[5];
    const double mr = st_[0], mi = st_[1], Rrr = st_[2], Rii = st_[3], Rri = st_[4], Crr = st_[5], Cii = st_[6], Cri = st_[7];
    const double W0 = w_[0], W1 = w_[1], W2 = w_[2];
    if (g_weight) {
        g_weight[3 * c + 0] = (float)(Rrr * N00 + Rri * N01);
        g_weight[3 * c + 1] = (float)(Rri * N10 + Rii * N11);
        g_weight[3 * c + 2] = (float)(Rri * N00 + Rii * N01 + Rrr * N10 + Rri * N11);
        g_bias[2 * c + 0] = (float)sgr;
        g_bias[2 * c + 1] = (float)sgi;
    }
    const double a0 = co_[0], a1 = co_[1], a2 = co_[2], a3 = co_[3];
    double d0 = 0, d1 = 0, d3 = 0, f0 = 0, f1 = 0;
    if (use_batch_stats) {
        const double n = (double)P;
        // g_R from M = W N
        const double gRrr = W0 * N00 + W2 * N10;
        const double gRii = W2 * N01 + W1 * N11;
        const double gRri = (W0 * N01 + W2 * N11) + (W2 * N00 + W1 * N10);
        // Jacobian of (Rrr, Rii, Rri) w.r.t. (Crr, Cii, Cri)
        const double s = sqrt(Crr * Cii - Cri * Cri);
        const double t = sqrt(Crr + Cii + 2.0 * s);
        const double q = 1.0 / (s * t);
        const double ds[3] = {Cii / (2.0 * s), Crr / (2.0 * s), -Cri / s};
        const double dC[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};          // d(Crr,Cii,Cri)/dX
        double gC[3];
#pragma unroll
        for (int X = 0; X < 3; ++X) {
            const double dt = (dC[X][0] + dC[X][1] + 2.0 * ds[X]) / (2.0 * t);
            const double dq = -q * (ds[X] / s + dt / t);
            const double dRrr = (dC[X][1] + ds[X]) * q + (Cii + s) * dq;
            const double dRii = (dC[X][0] + ds[X]) * q + (Crr + s) * dq;
            const double dRri = -dC[X][2] * q - Cri * dq;
            gC[X] = gRrr * dRrr + gRii * dRii + gRri * dRri;
        }
        d0 = 2.0 * gC[0] / n; d3 = 2.0 * gC[1] / n; d1 = gC[2] / n;
        const double er = -(a0 * sgr + a2 * sgi) / n, ei = -(a1 * sgr + a3 * sgi) / n;
        f0 = er - (d0 * mr + d1 * mi);
        f1 = ei - (d1 * mr + d3 * mi);
    }
    float* bc = bcoef + 10 * c;
    bc[0] = (float)a0; bc[1] = (float)a2; bc[2] = (float)a1; bc[3] = (float)a3;     // A^T
    bc[4] = (float)d0; bc[5] = (float)d1; bc[6] = (float)d1; bc[7] = (float)d3;
    bc[8] = (float)f0; bc[9] = (float)f1;
}

// Backward finalize of the REAL BatchNorm2d (dcs_rbn_fwd): per real channel, a = gamma / sigma,
//     g_x = a g_y - a s / n - a d N / (n sigma^2),   g_gamma = N / sigma,   g_beta = s,    s = sum g_y, N = sum g_y d
// i.e. the diagonal case of the coefficients above; merge (Cr == 1): both halves are one channel (sums pooled, 2n values).
__global__ void rbn_bwd_finalize_kernel(const double* __restrict__ part, int nblocks, const float* __restrict__ stats,
                                        const float* __restrict__ coef, float* __restrict__ g_weight,
                                        float* __restrict__ g_bias, float* __restrict__ bcoef, long P, int C, int merge,
                                        int use_batch_stats) {
    const int c = blockIdx.x, lane = threadIdx.x;
    // (all of a lane's slab loads in flight at once: see cbn_finalize_kernel)
    double S[6] = {0, 0, 0, 0, 0, 0};
    for (int b0 = lane; b0 < nblocks; b0 += 64 * 8) {
        double v[8][6];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + 64 * k < nblocks ? b0 + 64 * k : nblocks - 1;
#pragma unroll
            for (int i = 0; i < 6; ++i) v[k][i] = part[((long)b * C + c) * 6 + i];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 6; ++i) S[i] += b0 + 64 * k < nblocks ? v[k][i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) S[i] = dcs_wave_sum_d(S[i]);
    if (lane != 0) return;
    double sr = S[0], si = S[1], Nr = S[2], Ni = S[5], n = (double)P;
    const float* st = stats + 8 * c;
    const double mr = st[0], mi = st[1], isr = st[2], isi = st[3];
    if (merge) { sr = si = sr + si; Nr = Ni = Nr + Ni; n = 2.0 * n; }
    if (g_weight) {
        if (merge) { g_weight[0] = (float)(Nr * isr); g_bias[0] = (float)sr; }
        else {
            g_weight[2 * c] = (float)(Nr * isr); g_weight[2 * c + 1] = (float)(Ni * isi);
            g_bias[2 * c] = (float)sr; g_bias[2 * c + 1] = (float)si;
        }
    }
    const double a0 = coef[6 * c], a3 = coef[6 * c + 3];
    double d0 = 0, d3 = 0, f0 = 0, f1 = 0;
    if (use_batch_stats) {
        d0 = -a0 * Nr * isr * isr / n; d3 = -a3 * Ni * isi * isi / n;
        f0 = -a0 * sr / n - d0 * mr;  f1 = -a3 * si / n - d3 * mi;
    }
    float* bc = bcoef + 10 * c;
    bc[0] = (float)a0; bc[1] = 0.f; bc[2] = 0.f; bc[3] = (float)a3;
    bc[4] = (float)d0; bc[5] = 0.f; bc[6] = 0.f; bc[7] = (float)d3;
    bc[8] = (float)f0; bc[9] = (float)f1;
}

struct BChan { float t0, t1, t2, t3, d0, d1, d2, d3, f0, f1; };
__device__ __forceinline__ BChan load_bchan(const float* bcoef, int c) {
    const float* b = bcoef + 10 * c;
    BChan k;
    k.t0 = b[0]; k.t1 = b[1]; k.t2 = b[2]; k.t3 = b[3]; k.d0 = b[4]; k.d1 = b[5]; k.d2 = b[6]; k.d3 = b[7];
    k.f0 = b[8]; k.f1 = b[9];
    return k;
}
__device__ __forceinline__ float2 grad_x(const BChan& k, float2 gy, float xr, float xi) {
    return make_float2(fmaf(k.t0, gy.x, fmaf(k.t1, gy.y, fmaf(k.d0, xr, fmaf(k.d1, xi, k.f0)))),
                       fmaf(k.t2, gy.x, fmaf(k.t3, gy.y, fmaf(k.d2, xr, fmaf(k.d3, xi, k.f1)))));
}

template <int ACT, bool DROP>
__global__ __launch_bounds__(kThreads) void cbn_bwd_apply_kernel(const act_t* __restrict__ x,
                                                                  const act_t* __restrict__ go, act_t* __restrict__ gx,
                                                                  const float* __restrict__ coef,
                                                                  const float* __restrict__ stats,
                                                                  const float* __restrict__ bcoef, long P, int C, int G,
                                                                  int rows_per_iter, float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                                  const float4* __restrict__ g_add, float add_scale, long HW,
                                                                  const act_t* __restrict__ g2_) {
    DCS_PRIO_CRITICAL();
    if (seed_dev) seed += seed_dev[0];   // per-step device-side offset (graph replay safe)
    const int t = threadIdx.x;
    const float inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
    const ActIn4<act_t> x4 = act_in4(x), g4 = act_in4(go), g2 = act_in4(g2_);
    const ActOut4<act_t> o4 = act_out4(gx);
    if (C == 1) {
        const Chan k = load_chan(coef, stats, 0);
        const BChan bk = load_bchan(bcoef, 0);
        const long nvec = P / 2;
        for (long i = (long)blockIdx.x * kThreads + t; i < nvec; i += (long)gridDim.x * kThreads) {
            const float4 v = x4[i];
            float4 g = g4[i];
            if (g2_) { const float4 h = g2[i]; g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
            const uint64_t e = (uint64_t)i * 4;
            const float2 a = grad_x(bk, grad_y<ACT, DROP>(k, v.x, v.y, g.x, g.y, seed, e, drop_p, inv_keep), v.x, v.y);
            const float2 b = grad_x(bk, grad_y<ACT, DROP>(k, v.z, v.w, g.z, g.w, seed, e + 2, drop_p, inv_keep), v.z, v.w);
            o4[i] = make_float4(a.x, a.y, b.x, b.y);
        }
        if ((P & 1) && blockIdx.x == 0 && t == 0) {
            const long q = 2 * (P - 1);
            const float xr = dcs_ld1(x + q), xi = dcs_ld1(x + q + 1);
            const float2 a = grad_x(bk, grad_y<ACT, DROP>(k, xr, xi, dcs_ld1(go + q), dcs_ld1(go + q + 1), seed, (uint64_t)q, drop_p,
                                                          inv_keep), xr, xi);
            dcs_st1(gx + q, a.x); dcs_st1(gx + q + 1, a.y);
        }
        return;
    }
    const int g = t % G, r0 = t / G;
    const float inv_hw = 1.f / (float)(HW > 0 ? HW : 1);
    const Chan k0 = load_chan(coef, stats, 2 * g), k1 = load_chan(coef, stats, 2 * g + 1);
    const BChan b0 = load_bchan(bcoef, 2 * g), b1 = load_bchan(bcoef, 2 * g + 1);
    for (long r = (long)blockIdx.x * rows_per_iter + r0; r < P; r += (long)gridDim.x * rows_per_iter) {
        const float4 v = x4[r * G + g];
        float4 gg = g4[r * G + g];
        if (g2_) { const float4 h = g2[r * G + g]; gg.x += h.x; gg.y += h.y; gg.z += h.z; gg.w += h.w; }
        if (g_add) add_sample(gg, g_add[sample_of(r, HW, inv_hw) * G + g], add_scale);
        const uint64_t e = (uint64_t)(r * G + g) * 4;
        const float2 a = grad_x(b0, grad_y<ACT, DROP>(k0, v.x, v.y, gg.x, gg.y, seed, e, drop_p, inv_keep), v.x, v.y);
        const float2 b = grad_x(b1, grad_y<ACT, DROP>(k1, v.z, v.w, gg.z, gg.w, seed, e + 2, drop_p, inv_keep), v.z, v.w);
        o4[r * G + g] = make_float4(a.x, a.y, b.x, b.y);
    }
}

}  // namespace

#ifndef DCS_ACT_BF16
extern "C" long dcs_cbn_bwd_workspace_bytes(long P, int C) {
    cbn::Geom g;
    if (!cbn::geom(P, C, &g)) return -1;
    return (long)g.nblocks * C * 6 * (long)sizeof(double) + (long)C * 10 * (long)sizeof(float);
}
#endif

extern "C" int DCS_SYM(dcs_cbn_bwd_add)(const act_t* x, const act_t* g_out, act_t* g_x, const float* weight, const float* stats,
                               const float* coef, float* g_weight, float* g_bias, void* workspace, long workspace_bytes,
                               long P, int C, int use_batch_stats, int act, float drop_p, unsigned long long seed,
                               const unsigned long long* seed_dev, const float* g_add, float add_scale, long HW,
                               const act_t* g_out2, dcs_stream_t stream) {
    cbn::Geom g;
    if (!x || !g_out || !stats || !coef || !workspace || !cbn::geom(P, C, &g)) return DCS_ERR_BADARG;   // (g_x NULL: parameter gradients only)
    if (g_add && (C < 2 || HW <= 0 || P % HW != 0)) return DCS_ERR_BADARG;
    if (g_add && P / HW > (1L << 20)) return DCS_ERR_BADARG;              // sample_of's precondition (above)
    if (g_out2 && C == 1 && (P & 1)) return DCS_ERR_BADARG;      // the scalar tail of the one-channel layout reads g_out only
    const act_t* gb = g_out2;
    const float4* ga = reinterpret_cast<const float4*>(g_add);
    if ((g_weight == nullptr) != (g_bias == nullptr)) return DCS_ERR_BADARG;
    if (act != DCS_ACT_NONE && act != DCS_ACT_RELU && act != DCS_ACT_LRELU) return DCS_ERR_BADARG;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return DCS_ERR_BADARG;
    const long part_bytes = (long)g.nblocks * C * 6 * (long)sizeof(double);
    if (workspace_bytes < part_bytes + (long)C * 10 * (long)sizeof(float)) return DCS_ERR_WORKSPACE;
    double* part = (double*)workspace;
    float* bcoef = (float*)((char*)workspace + part_bytes);
    hipStream_t s = dcs_stream(stream);
    const bool drop = drop_p > 0.f;
    const int grid2 = cbn::stream_grid(P, C, g);
#define DCS_CBN_BWD(A, D)                                                                                          \
    do {                                                                                                           \
        DCS_LAUNCH((cbn_bwd_reduce_kernel<A, D>), dim3(g.nblocks), dim3(kThreads), 0, s, x, g_out, coef,    \
                           stats, part, P, C, g.vec_per_row, g.rows_per_iter, drop_p, (uint64_t)seed, (const uint64_t*)seed_dev,   \
                           ga, add_scale, HW, gb);                                                                 \
        DCS_LAUNCH(cbn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, (const double*)part,      \
                           g.nblocks, weight, stats, coef, g_weight, g_bias, bcoef, P, C, use_batch_stats);        \
        if (g_x)                                                                                                   \
            DCS_LAUNCH((cbn_bwd_apply_kernel<A, D>), dim3(grid2), dim3(kThreads), 0, s, x, g_out, g_x, coef,    \
                               stats, (const float*)bcoef, P, C, g.vec_per_row, g.rows_per_iter, drop_p,           \
                               (uint64_t)seed, (const uint64_t*)seed_dev, ga, add_scale, HW, gb);                  \
    } while (0)
    if (act == DCS_ACT_RELU) { if (drop) DCS_CBN_BWD(DCS_ACT_RELU, true); else DCS_CBN_BWD(DCS_ACT_RELU, false); }
    else if (act == DCS_ACT_LRELU) { if (drop) DCS_CBN_BWD(DCS_ACT_LRELU, true); else DCS_CBN_BWD(DCS_ACT_LRELU, false); }
    else { if (drop) DCS_CBN_BWD(DCS_ACT_NONE, true); else DCS_CBN_BWD(DCS_ACT_NONE, false); }
#undef DCS_CBN_BWD
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int DCS_SYM(dcs_cbn_bwd)(const act_t* x, const act_t* g_out, act_t* g_x, const float* weight, const float* stats,
                           const float* coef, float* g_weight, float* g_bias, void* workspace, long workspace_bytes,
                           long P, int C, int use_batch_stats, int act, float drop_p, unsigned long long seed,
                           const unsigned long long* seed_dev, dcs_stream_t stream) {
    return DCS_SYM(dcs_cbn_bwd_add)(x, g_out, g_x, weight, stats, coef, g_weight, g_bias, workspace, workspace_bytes, P, C,
                           use_batch_stats, act, drop_p, seed, seed_dev, nullptr, 0.f, 0, nullptr, stream);
}

#ifndef DCS_ACT_BF16
// Backward of dcs_rbn_fwd: g_x, g_weight[Cr], g_bias[Cr] (both NULL for affine=False); same Cr / P conventions.
extern "C" int dcs_rbn_bwd(const float* x, const float* g_out, float* g_x, const float* stats, const float* coef,
                           float* g_weight, float* g_bias, void* workspace, long workspace_bytes, long P, int Cr,
                           int use_batch_stats, int act, dcs_stream_t stream) {
    const int merge = Cr == 1;
    if (!x || !g_out || !g_x || !stats || !coef || !workspace || Cr < 1 || (!merge && (Cr & 1)) || (merge && (P & 3)))
        return DCS_ERR_BADARG;                       // merge: P/2 complex pixels, float4 pairs -> P % 4 == 0
    const int C = merge ? 1 : Cr / 2;
    const long Pc = merge ? P / 2 : P;
    cbn::Geom g;
    if (!cbn::geom(Pc, C, &g)) return DCS_ERR_BADARG;
    if ((g_weight == nullptr) != (g_bias == nullptr)) return DCS_ERR_BADARG;
    if (act != DCS_ACT_NONE && act != DCS_ACT_RELU && act != DCS_ACT_LRELU) return DCS_ERR_BADARG;
    const long part_bytes = (long)g.nblocks * C * 6 * (long)sizeof(double);
    if (workspace_bytes < part_bytes + (long)C * 10 * (long)sizeof(float)) return DCS_ERR_WORKSPACE;
    double* part = (double*)workspace;
    float* bcoef = (float*)((char*)workspace + part_bytes);
    hipStream_t s = dcs_stream(stream);
    const int grid2 = cbn::stream_grid(Pc, C, g);
#define DCS_RBN_BWD(A)                                                                                             \
    do {                                                                                                           \
        DCS_LAUNCH((cbn_bwd_reduce_kernel<A, false>), dim3(g.nblocks), dim3(kThreads), 0, s, x, g_out, coef, stats, part, \
                   Pc, C, g.vec_per_row, g.rows_per_iter, 0.f, (uint64_t)0, (const uint64_t*)nullptr,              \
                   (const float4*)nullptr, 0.f, (long)0, (const act_t*)nullptr);                                   \
        DCS_LAUNCH(rbn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, (const double*)part, g.nblocks, stats, coef,   \
                   g_weight, g_bias, bcoef, Pc, C, merge, use_batch_stats);                                        \
        DCS_LAUNCH((cbn_bwd_apply_kernel<A, false>), dim3(grid2), dim3(kThreads), 0, s, x, g_out, g_x, coef, stats, \
                   (const float*)bcoef, Pc, C, g.vec_per_row, g.rows_per_iter, 0.f, (uint64_t)0,                   \
                   (const uint64_t*)nullptr, (const float4*)nullptr, 0.f, (long)0, (const act_t*)nullptr);         \
    } while (0)
    if (act == DCS_ACT_RELU) DCS_RBN_BWD(DCS_ACT_RELU);
    else if (act == DCS_ACT_LRELU) DCS_RBN_BWD(DCS_ACT_LRELU);
    else DCS_RBN_BWD(DCS_ACT_NONE);
#undef DCS_RBN_BWD
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
#endif
