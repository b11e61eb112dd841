// cbn.hip — ComplexBatchNorm2d (complexPyTorch 0.3 semantics) for channels-last complex64.
//
// Replaces the ~12 reduction / element-wise ATen launches per CBN of the reference
// (c_network.py:101,113,148) with:
//   cbn_stats_kernel     one coalesced streaming read, per-thread fp32 partials around a
//                        per-channel pivot, fp64 combine through LDS, one [C][5] fp64 slab
//                        per workgroup (no atomics: bitwise reproducible)
//   cbn_finalize_kernel  C threads: slab sum, covariance, closed-form inverse square root,
//                        running-stat update, 2x2 affine folded to 6 coefficients / channel
//   cbn_apply_kernel     y = act(A x + c) (+ dropout), float4 streaming, coefficients in VGPRs
// HBM-bound: algorithmic bytes per pixel-channel = 8 (stats read) + 8 + 8 (apply read + write).
#include "dcs_common.h"
#include "cbn_geom.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 512;

struct CbnGeom {
    int vec_per_row;   // float4 groups per pixel row (C/2), or 0 for the C == 1 layout
    int rows_per_iter; // pixel rows covered by one workgroup iteration
    int nblocks;
};

inline bool cbn_geom(long P, int C, CbnGeom* g) {
    if (P <= 0 || C <= 0) return false;
    if (C == 1) {
        long nvec = P / 2;
        long it = (nvec + kThreads - 1) / kThreads;
        long nb = (it + DCS_CBN_RED_IT - 1) / DCS_CBN_RED_IT;
        g->vec_per_row = 0;
        g->rows_per_iter = 0;
        g->nblocks = (int)(nb < 1 ? 1 : (nb > kMaxBlocks ? kMaxBlocks : nb));
        return true;
    }
    if (C & 1) return false;
    int G = C / 2;
    if (G > kThreads || (kThreads % G) != 0) return false;
    g->vec_per_row = G;
    g->rows_per_iter = kThreads / G;
    long it = (P + g->rows_per_iter - 1) / g->rows_per_iter;
    long nb = (it + DCS_CBN_RED_IT - 1) / DCS_CBN_RED_IT;
    g->nblocks = (int)(nb < 1 ? 1 : (nb > kMaxBlocks ? kMaxBlocks : nb));
    return true;
}

// partial slab layout: double part[nblocks][C][5] = {S_r, S_i, S_rr, S_ii, S_ri} of (x - pivot)
__global__ __launch_bounds__(kThreads) void cbn_stats_kernel(const act_t* __restrict__ x, double* __restrict__ part,
                                                              long P, int C, int G, int rows_per_iter) {
    __shared__ double red[kThreads * 10];
    const int t = threadIdx.x;
    float s[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = 0.f;

    if (C == 1) {
        const float kr = dcs_ld1(x), ki = dcs_ld1(x + 1);
        const long nvec = P / 2;
        const ActIn4<act_t> x4 = act_in4(x);
        for (long i = (long)blockIdx.x * kThreads + t; i < nvec; i += (long)gridDim.x * kThreads) {
            float4 v = x4[i];
            float ar = v.x - kr, ai = v.y - ki, br = v.z - kr, bi = v.w - ki;
            s[0] += ar + br;
            s[1] += ai + bi;
            s[2] = fmaf(ar, ar, fmaf(br, br, s[2]));
            s[3] = fmaf(ai, ai, fmaf(bi, bi, s[3]));
            s[4] = fmaf(ar, ai, fmaf(br, bi, s[4]));
        }
        if ((P & 1) && blockIdx.x == 0 && t == 0) {
            float ar = dcs_ld1(x + 2 * (P - 1)) - kr, ai = dcs_ld1(x + 2 * (P - 1) + 1) - ki;
            s[0] += ar; s[1] += ai; s[2] += ar * ar; s[3] += ai * ai; s[4] += ar * ai;
        }
        // all threads hold channel 0
        double d[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) d[i] = dcs_wave_sum_d((double)s[i]);
        const int wave = t >> 6, lane = t & 63;
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) red[wave * 5 + i] = d[i];
        }
        __syncthreads();
        if (t < 5) {
            double a = 0;
            for (int w = 0; w < kThreads / 64; ++w) a += red[w * 5 + t];
            part[(long)blockIdx.x * 5 + t] = a;
        }
        return;
    }

    const int g = t % G;      // float4 group = complex channels 2g, 2g+1
    const int r0 = t / G;
    const ActIn4<act_t> x4 = act_in4(x);
    const float4 piv = x4[g];   // row 0 of this channel pair
    for (long r = (long)blockIdx.x * rows_per_iter + r0; r < P; r += (long)gridDim.x * rows_per_iter) {
        float4 v = x4[r * G + g];
        float ar = v.x - piv.x, ai = v.y - piv.y, br = v.z - piv.z, bi = v.w - piv.w;
        s[0] += ar; s[1] += ai;
        s[2] = fmaf(ar, ar, s[2]); s[3] = fmaf(ai, ai, s[3]); s[4] = fmaf(ar, ai, s[4]);
        s[5] += br; s[6] += bi;
        s[7] = fmaf(br, br, s[7]); s[8] = fmaf(bi, bi, s[8]); s[9] = fmaf(br, bi, s[9]);
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) red[t * 10 + i] = (double)s[i];
    __syncthreads();
    // threads 0..G*10-1 each own one (g, i) column and sum over the rows_per_iter copies
    for (int o = t; o < G * 10; o += kThreads) {
        const int gg = o / 10, i = o % 10;
        double a = 0;
        for (int r = 0; r < rows_per_iter; ++r) a += red[(r * G + gg) * 10 + i];
        const int c = 2 * gg + (i >= 5);
        part[((long)blockIdx.x * C + c) * 5 + (i % 5)] = a;
    }
}

// PT: element type of the partial slabs — double (cbn_stats_kernel) or float (the conv epilogues' per-workgroup sums,
// conv_common.h Args::stat).  pivot: float[C][2], the value the partial sums are relative to (pixel 0 of x, or the conv's
// bias).  One workgroup of 256 threads per channel: a thread's slab loads are all in flight at once (8 per round), then a
// wave butterfly and an LDS combine in a fixed order.
template <typename PT, typename PV>
__global__ __launch_bounds__(256) void cbn_finalize_kernel(const PV* __restrict__ pivot, const PT* __restrict__ part, int nblocks, int stride,
                                    const float* __restrict__ weight, const float* __restrict__ bias,
                                    float* __restrict__ running_mean, float* __restrict__ running_covar,
                                    float* __restrict__ stats_out, float* __restrict__ coef_out,
                                    long P, int C, float eps, float momentum, int use_batch_stats) {
    __shared__ double wsum[4][5];
    const int c = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // the channel's pivot, running statistics and affine parameters are requested FIRST (uniform addresses): behind the slab sums,
    // the barrier and the `t != 0` exit they were two more dependent round trips of a kernel that is nothing but round trips
    const bool run = momentum >= 0.f && running_mean != nullptr;
    float pv_[2] = {0.f, 0.f}, rm_[2] = {0.f, 0.f}, rc_[3] = {0.f, 0.f, 0.f}, w_[3] = {1.f, 1.f, 0.f}, b_[2] = {0.f, 0.f};
    if (use_batch_stats) { pv_[0] = dcs_ld1(pivot + 2 * c); pv_[1] = dcs_ld1(pivot + 2 * c + 1); }
    if (run || !use_batch_stats) {
        rm_[0] = running_mean[2 * c]; rm_[1] = running_mean[2 * c + 1];
        rc_[0] = running_covar[3 * c]; rc_[1] = running_covar[3 * c + 1]; rc_[2] = running_covar[3 * c + 2];
    }
    if (weight != nullptr) { w_[0] = weight[3 * c]; w_[1] = weight[3 * c + 1]; w_[2] = weight[3 * c + 2]; b_[0] = bias[2 * c]; b_[1] = bias[2 * c + 1]; }
    float mr, mi, Crr, Cii, Cri;
    if (use_batch_stats) {
        double S[5] = {0, 0, 0, 0, 0};
        // double slabs: [block][C][5] (cbn_stats_kernel); float slabs: [C][5][stride] (conv epilogues: coalesced here)
        constexpr bool ROWS = sizeof(PT) == sizeof(float);
        for (int b0 = t; b0 < nblocks; b0 += 256 * 8) {
            PT v[8][5];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int b = b0 + 256 * k < nblocks ? b0 + 256 * k : nblocks - 1;
#pragma unroll
                for (int i = 0; i < 5; ++i) v[k][i] = ROWS ? part[(long)(c * 5 + i) * stride + b] : part[((long)b * C + c) * 5 + i];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 5; ++i) S[i] += b0 + 256 * k < nblocks ? (double)v[k][i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) S[i] = dcs_wave_sum_d(S[i]);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) wsum[wave][i] = S[i];
        }
        __syncthreads();
        if (t != 0) return;
#pragma unroll
        for (int i = 0; i < 5; ++i) S[i] = wsum[0][i] + wsum[1][i] + wsum[2][i] + wsum[3][i];
        const double n = (double)P;
        const double kr = (double)pv_[0], ki = (double)pv_[1];
        const double dr = S[0] / n, di = S[1] / n;
        mr = (float)(kr + dr);
        mi = (float)(ki + di);
        Crr = (float)(S[2] / n - dr * dr) + eps;
        Cii = (float)(S[3] / n - di * di) + eps;
        Cri = (float)(S[4] / n - dr * di);
        if (run) {
            const float f = momentum;
            const float unb = P > 1 ? (float)(n / (n - 1.0)) : 1.f;
            running_mean[2 * c] = f * mr + (1.f - f) * rm_[0];
            running_mean[2 * c + 1] = f * mi + (1.f - f) * rm_[1];
            running_covar[3 * c + 0] = f * Crr * unb + (1.f - f) * rc_[0];
            running_covar[3 * c + 1] = f * Cii * unb + (1.f - f) * rc_[1];
            running_covar[3 * c + 2] = f * Cri * unb + (1.f - f) * rc_[2];
        }
    } else {
        if (t != 0) return;
        mr = rm_[0];
        mi = rm_[1];
        Crr = rc_[0] + eps;
        Cii = rc_[1] + eps;
        Cri = rc_[2];
    }
    const float det = Crr * Cii - Cri * Cri;
    const float s = sqrtf(det);
    const float tt = sqrtf(Cii + Crr + 2.f * s);
    const float ist = 1.0f / (s * tt);
    const float Rrr = (Cii + s) * ist, Rii = (Crr + s) * ist, Rri = -Cri * ist;
    const float W0 = w_[0], W1 = w_[1], W2 = w_[2], b0 = b_[0], b1 = b_[1];
    const float a0 = W0 * Rrr + W2 * Rri, a1 = W0 * Rri + W2 * Rii;
    const float a2 = W2 * Rrr + W1 * Rri, a3 = W2 * Rri + W1 * Rii;
    float* st = stats_out + 8 * c;
    st[0] = mr; st[1] = mi; st[2] = Rrr; st[3] = Rii; st[4] = Rri; st[5] = Crr; st[6] = Cii; st[7] = Cri;
    float* co = coef_out + 6 * c;
    co[0] = a0; co[1] = a1; co[2] = a2; co[3] = a3;
    co[4] = b0 - a0 * mr - a1 * mi;
    co[5] = b1 - a2 * mr - a3 * mi;
}

#ifndef DCS_ACT_BF16
// torch.nn.BatchNorm2d on a REAL channels-last tensor float[P][Cr] (DR-Net, r_network.py:56,66,106): with even Cr the
// tensor is an interleaved complex one with Cr/2 channels, the statistics kernel above already yields every real
// channel's first and second moment (S_ri is ignored), and the affine map is the DIAGONAL 2x2 block
// (a0, 0, 0, a3 | c0, c1), a = gamma / sqrt(var + eps) — same apply kernel.  Cr = 1 (the initial BatchNorm over the
// [B,F,T] magnitude): the P values are read as P/2 complex pixels whose two halves are the same channel (merge).
// stats_out[c] = {mean_r, mean_i, 1/sigma_r, 1/sigma_i, 0, var_r + eps, var_i + eps, 0}.
__global__ void rbn_finalize_kernel(const float* __restrict__ x, const double* __restrict__ part, int nblocks,
                                    const float* __restrict__ weight, const float* __restrict__ bias,
                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                    float* __restrict__ stats_out, float* __restrict__ coef_out, long P, int C, int merge,
                                    float eps, float momentum, int use_batch_stats) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float mr, mi, vr, vi;
    if (use_batch_stats) {
        // all of a lane's slab loads in flight at once (<= 512 slabs = 8 per lane; clamped index, masked value): as a rolled
        // loop of `b < nblocks` trips every trip was its own L2 round trip — 8 x ~0.7 us of a 5.7 us kernel
        double S[5] = {0, 0, 0, 0, 0};
        for (int b0 = lane; b0 < nblocks; b0 += 64 * 8) {
            double v[8][5];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int b = b0 + 64 * k < nblocks ? b0 + 64 * k : nblocks - 1;
#pragma unroll
                for (int i = 0; i < 5; ++i) v[k][i] = part[((long)b * C + c) * 5 + i];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 5; ++i) S[i] += b0 + 64 * k < nblocks ? v[k][i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) S[i] = dcs_wave_sum_d(S[i]);
        if (lane != 0) return;
        const double n = (double)P;                                       // complex pixels
        const double kr = (double)x[2 * c], ki = (double)x[2 * c + 1];   // pivot = pixel 0
        double m_r = kr + S[0] / n, m_i = ki + S[1] / n;
        double v_r = S[2] / n - (S[0] / n) * (S[0] / n), v_i = S[3] / n - (S[1] / n) * (S[1] / n);
        double cnt = n;
        if (merge) {                      // one real channel seen as (re, im) halves: pool the two halves' moments
            const double m = 0.5 * (m_r + m_i);
            const double v = 0.5 * (v_r + (m_r - m) * (m_r - m) + v_i + (m_i - m) * (m_i - m));
            m_r = m_i = m; v_r = v_i = v; cnt = 2 * n;
        }
        mr = (float)m_r; mi = (float)m_i; vr = (float)v_r; vi = (float)v_i;
        if (momentum >= 0.f && running_mean != nullptr) {
            const float f = momentum, unb = cnt > 1 ? (float)(cnt / (cnt - 1.0)) : 1.f;
            if (merge) {
                running_mean[0] = f * mr + (1.f - f) * running_mean[0];
                running_var[0] = f * vr * unb + (1.f - f) * running_var[0];
            } else {
                running_mean[2 * c] = f * mr + (1.f - f) * running_mean[2 * c];
                running_mean[2 * c + 1] = f * mi + (1.f - f) * running_mean[2 * c + 1];
                running_var[2 * c] = f * vr * unb + (1.f - f) * running_var[2 * c];
                running_var[2 * c + 1] = f * vi * unb + (1.f - f) * running_var[2 * c + 1];
            }
        }
    } else {
        if (lane != 0) return;
        mr = running_mean[merge ? 0 : 2 * c]; mi = running_mean[merge ? 0 : 2 * c + 1];
        vr = running_var[merge ? 0 : 2 * c]; vi = running_var[merge ? 0 : 2 * c + 1];
    }
    const float isr = 1.0f / sqrtf(vr + eps), isi = 1.0f / sqrtf(vi + eps);
    float w0 = 1.f, w1 = 1.f, b0 = 0.f, b1 = 0.f;
    if (weight != nullptr) {
        w0 = weight[merge ? 0 : 2 * c]; w1 = weight[merge ? 0 : 2 * c + 1];
        b0 = bias[merge ? 0 : 2 * c]; b1 = bias[merge ? 0 : 2 * c + 1];
    }
    const float a0 = w0 * isr, a3 = w1 * isi;
    float* st = stats_out + 8 * c;
    st[0] = mr; st[1] = mi; st[2] = isr; st[3] = isi; st[4] = 0.f; st[5] = vr + eps; st[6] = vi + eps; st[7] = 0.f;
    float* co = coef_out + 6 * c;
    co[0] = a0; co[1] = 0.f; co[2] = 0.f; co[3] = a3;
    co[4] = b0 - a0 * mr;
    co[5] = b1 - a3 * mi;
}

#endif

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
    if (ACT == DCS_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == DCS_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
    return v;
}

// y = act(A x + c); thread keeps the coefficients of its fixed channel pair in registers.
template <int ACT, bool DROP>
__global__ __launch_bounds__(kThreads) void cbn_apply_kernel(const act_t* __restrict__ x, act_t* __restrict__ y,
                                                              const float* __restrict__ coef, long P, int C, int G,
                                                              int rows_per_iter, float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    if (seed_dev) seed += seed_dev[0];   // per-step device-side offset (graph replay safe)
    const float inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
    const int t = threadIdx.x;
    const ActIn4<act_t> x4 = act_in4(x);
    const ActOut4<act_t> y4 = act_out4(y);
    if (C == 1) {
        const float a0 = coef[0], a1 = coef[1], a2 = coef[2], a3 = coef[3], c0 = coef[4], c1 = coef[5];
        const long nvec = P / 2;
        for (long i = (long)blockIdx.x * kThreads + t; i < nvec; i += (long)gridDim.x * kThreads) {
            float4 v = x4[i], o;
            o.x = act_fn<ACT>(fmaf(a0, v.x, fmaf(a1, v.y, c0)));
            o.y = act_fn<ACT>(fmaf(a2, v.x, fmaf(a3, v.y, c1)));
            o.z = act_fn<ACT>(fmaf(a0, v.z, fmaf(a1, v.w, c0)));
            o.w = act_fn<ACT>(fmaf(a2, v.z, fmaf(a3, v.w, c1)));
            if (DROP) {
                const uint64_t e = (uint64_t)i * 4;
                o.x *= dcs_keep_scale(seed, e, drop_p, inv_keep);
                o.y *= dcs_keep_scale(seed, e + 1, drop_p, inv_keep);
                o.z *= dcs_keep_scale(seed, e + 2, drop_p, inv_keep);
                o.w *= dcs_keep_scale(seed, e + 3, drop_p, inv_keep);
            }
            y4[i] = o;
        }
        if ((P & 1) && blockIdx.x == 0 && t == 0) {
            const float xr = dcs_ld1(x + 2 * (P - 1)), xi = dcs_ld1(x + 2 * (P - 1) + 1);
            float yr = act_fn<ACT>(fmaf(a0, xr, fmaf(a1, xi, c0)));
            float yi = act_fn<ACT>(fmaf(a2, xr, fmaf(a3, xi, c1)));
            if (DROP) {
                yr *= dcs_keep_scale(seed, (uint64_t)2 * (P - 1), drop_p, inv_keep);
                yi *= dcs_keep_scale(seed, (uint64_t)2 * (P - 1) + 1, drop_p, inv_keep);
            }
            dcs_st1(y + 2 * (P - 1), yr);
            dcs_st1(y + 2 * (P - 1) + 1, yi);
        }
        return;
    }
    const int g = t % G, r0 = t / G;
    const float* ca = coef + 12 * g;   // channels 2g and 2g+1, 6 floats each
    const float a0 = ca[0], a1 = ca[1], a2 = ca[2], a3 = ca[3], c0 = ca[4], c1 = ca[5];
    const float e0 = ca[6], e1 = ca[7], e2 = ca[8], e3 = ca[9], f0 = ca[10], f1 = ca[11];
    for (long r = (long)blockIdx.x * rows_per_iter + r0; r < P; r += (long)gridDim.x * rows_per_iter) {
        float4 v = x4[r * G + g], o;
        o.x = act_fn<ACT>(fmaf(a0, v.x, fmaf(a1, v.y, c0)));
        o.y = act_fn<ACT>(fmaf(a2, v.x, fmaf(a3, v.y, c1)));
        o.z = act_fn<ACT>(fmaf(e0, v.z, fmaf(e1, v.w, f0)));
        o.w = act_fn<ACT>(fmaf(e2, v.z, fmaf(e3, v.w, f1)));
        if (DROP) {
            const uint64_t e = (uint64_t)(r * G + g) * 4;
            o.x *= dcs_keep_scale(seed, e, drop_p, inv_keep);
            o.y *= dcs_keep_scale(seed, e + 1, drop_p, inv_keep);
            o.z *= dcs_keep_scale(seed, e + 2, drop_p, inv_keep);
            o.w *= dcs_keep_scale(seed, e + 3, drop_p, inv_keep);
        }
        y4[r * G + g] = o;
    }
}

// The decoder stages' CBN + CLReLU is followed at once by a channel attention whose first step is a per-sample average pool
// of this kernel's OUTPUT (c_network.py:148-150, :219; network_functions.py:135-138): apply and pool in one pass — grid
// (chunks, B), a workgroup covers rows of ONE sample and leaves the slab double[C][2] of its output sums that ca_fc_kernel
// (attention.hip) reads (same layout and chunk count as ca_pool_kernel: att::ca_chunks), so that kernel's read of the
// activation and its launch go.  The sums are taken over the values as STORED (bf16 storage: after rounding).
__device__ __forceinline__ float stored(float v) { return DCS_ACT_IS_BF16 ? dcs_bf16_to_f32(dcs_f32_to_bf16(v)) : v; }

template <int ACT>
__global__ __launch_bounds__(kThreads) void cbn_apply_pool_kernel(const act_t* __restrict__ x, act_t* __restrict__ y,
                                                                   const float* __restrict__ coef, double* __restrict__ part,
                                                                   long HW, int C, int G) {
    __shared__ double red[kThreads * 4];
    const int t = threadIdx.x, g = t % G, r0 = t / G, rpi = kThreads / G;
    const int b = blockIdx.y, bx = blockIdx.x, gx = gridDim.x;
    const ActIn4<act_t> x4 = act_in4(x) + (long)b * HW * G;
    const ActOut4<act_t> y4 = act_out4(y) + (long)b * HW * G;
    const float* ca = coef + 12 * g;
    const float a0 = ca[0], a1 = ca[1], a2 = ca[2], a3 = ca[3], c0 = ca[4], c1 = ca[5];
    const float e0 = ca[6], e1 = ca[7], e2 = ca[8], e3 = ca[9], f0 = ca[10], f1 = ca[11];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (long r = (long)bx * rpi + r0; r < HW; r += (long)gx * rpi) {
        const float4 v = x4[r * G + g];
        float4 o;
        o.x = act_fn<ACT>(fmaf(a0, v.x, fmaf(a1, v.y, c0)));
        o.y = act_fn<ACT>(fmaf(a2, v.x, fmaf(a3, v.y, c1)));
        o.z = act_fn<ACT>(fmaf(e0, v.z, fmaf(e1, v.w, f0)));
        o.w = act_fn<ACT>(fmaf(e2, v.z, fmaf(e3, v.w, f1)));
        y4[r * G + g] = o;
        s0 += stored(o.x); s1 += stored(o.y); s2 += stored(o.z); s3 += stored(o.w);
    }
    red[t * 4 + 0] = s0; red[t * 4 + 1] = s1; red[t * 4 + 2] = s2; red[t * 4 + 3] = s3;
    __syncthreads();
    for (int o = t; o < G * 4; o += kThreads) {
        const int gg = o / 4, i = o % 4;
        double a = 0;
        for (int r = 0; r < rpi; ++r) a += red[(r * G + gg) * 4 + i];
        part[(((long)b * gx + bx) * C + 2 * gg) * 2 + i] = a;
    }
}

}  // namespace

#ifndef DCS_ACT_BF16
extern "C" long dcs_cbn_workspace_bytes(long P, int C) {
    CbnGeom g;
    if (!cbn_geom(P, C, &g)) return -1;
    return (long)g.nblocks * C * 5 * (long)sizeof(double);
}
#endif

static int cbn_fwd_impl(const act_t* x, act_t* y, const float* weight, const float* bias, float* running_mean,
                        float* running_covar, float* stats_out, float* coef_out, void* workspace,
                        long workspace_bytes, long P, int C, float eps, float momentum, int use_batch_stats,
                        int act, float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream,
                        const float* slab_part, int slab_rows, int slab_stride, const float* slab_pivot) {
    CbnGeom g;
    if (!x || !y || !stats_out || !coef_out || !cbn_geom(P, C, &g)) return DCS_ERR_BADARG;
    if ((weight == nullptr) != (bias == nullptr)) return DCS_ERR_BADARG;
    if (use_batch_stats < 0 || use_batch_stats > 3) return DCS_ERR_BADARG;
    if (use_batch_stats == 3 && (!slab_part || !slab_pivot || slab_rows < 1 || slab_stride < slab_rows)) return DCS_ERR_BADARG;
    if (use_batch_stats == 0 && (!running_mean || !running_covar)) return DCS_ERR_BADARG;
    if (act != DCS_ACT_NONE && act != DCS_ACT_RELU && act != DCS_ACT_LRELU) return DCS_ERR_BADARG;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return DCS_ERR_BADARG;
    hipStream_t s = dcs_stream(stream);
    if (use_batch_stats == 1) {
        if (!workspace || workspace_bytes < (long)g.nblocks * C * 5 * (long)sizeof(double)) return DCS_ERR_WORKSPACE;
        DCS_LAUNCH(cbn_stats_kernel, dim3(g.nblocks), dim3(kThreads), 0, s, x, (double*)workspace, P, C,
                           g.vec_per_row, g.rows_per_iter);
        DCS_CHECK_LAUNCH();
    }
    if (use_batch_stats == 3) {            // 3: the producing conv's epilogue left the partial sums (dcs_cbn_fwd_slabs)
        DCS_LAUNCH((cbn_finalize_kernel<float, float>), dim3(C), dim3(256), 0, s, slab_pivot, slab_part, slab_rows, slab_stride, weight, bias,
                   running_mean, running_covar, stats_out, coef_out, P, C, eps, momentum, 1);
        DCS_CHECK_LAUNCH();
    } else if (use_batch_stats != 2) {     // 2: coef_out already holds the coefficients of an earlier eval-mode call
        DCS_LAUNCH((cbn_finalize_kernel<double, act_t>), dim3(C), dim3(256), 0, s, x, (const double*)workspace,
                           g.nblocks, 0, weight, bias, running_mean, running_covar, stats_out, coef_out, P, C, eps, momentum,
                           use_batch_stats);
        DCS_CHECK_LAUNCH();
    }
    // apply: ~4 float4 per thread per workgroup pass, capped at 2048 workgroups
    long iters = (C == 1) ? (P / 2 + kThreads - 1) / kThreads : (P + g.rows_per_iter - 1) / g.rows_per_iter;
    long nb = (iters + DCS_CBN_APP_IT - 1) / DCS_CBN_APP_IT;
    int grid = (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
#define DCS_CBN_APPLY(A, D)                                                                                   \
    DCS_LAUNCH((cbn_apply_kernel<A, D>), dim3(grid), dim3(kThreads), 0, s, x, y, coef_out, P, C,         \
                       g.vec_per_row, g.rows_per_iter, drop_p, (uint64_t)seed, (const uint64_t*)seed_dev)
    const bool drop = drop_p > 0.f;
    if (act == DCS_ACT_RELU) { if (drop) DCS_CBN_APPLY(DCS_ACT_RELU, true); else DCS_CBN_APPLY(DCS_ACT_RELU, false); }
    else if (act == DCS_ACT_LRELU) { if (drop) DCS_CBN_APPLY(DCS_ACT_LRELU, true); else DCS_CBN_APPLY(DCS_ACT_LRELU, false); }
    else { if (drop) DCS_CBN_APPLY(DCS_ACT_NONE, true); else DCS_CBN_APPLY(DCS_ACT_NONE, false); }
#undef DCS_CBN_APPLY
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int DCS_SYM(dcs_cbn_fwd)(const act_t* x, act_t* y, const float* weight, const float* bias, float* running_mean,
                           float* running_covar, float* stats_out, float* coef_out, void* workspace,
                           long workspace_bytes, long P, int C, float eps, float momentum, int use_batch_stats,
                           int act, float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream) {
    if (use_batch_stats == 3) return DCS_ERR_BADARG;
    return cbn_fwd_impl(x, y, weight, bias, running_mean, running_covar, stats_out, coef_out, workspace, workspace_bytes, P, C,
                        eps, momentum, use_batch_stats, act, drop_p, seed, seed_dev, stream, nullptr, 0, 0, nullptr);
}

// Training-mode dcs_cbn_fwd whose batch statistics come from the conv that produced x (dcs_cconv2d_fwd_stats): `part` =
// float[C][5][stride] partial sums of (x - pivot), columns 0..rows-1 valid, `pivot` = float[C][2] (the conv's packed bias).
// No statistics pass over x.
extern "C" int DCS_SYM(dcs_cbn_fwd_slabs)(const act_t* x, act_t* y, const float* weight, const float* bias, float* running_mean,
                                 float* running_covar, float* stats_out, float* coef_out, const float* part, int rows,
                                 int stride, const float* pivot, long P, int C, float eps, float momentum, int act, float drop_p,
                                 unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream) {
    return cbn_fwd_impl(x, y, weight, bias, running_mean, running_covar, stats_out, coef_out, nullptr, 0, P, C, eps, momentum,
                        3, act, drop_p, seed, seed_dev, stream, part, rows, stride, pivot);
}

// dcs_cbn_fwd_slabs (batch statistics from the producing conv) whose apply pass also leaves the per-sample pooling slabs of
// its output for the channel attention that follows: pool_part = double[B][chunks][C][2], chunks = dcs_ca_pool_chunks(HW, C)
// (consumer: dcs_channel_attention_fc_fwd).  x = [B][HW][C] complex; no dropout (the decoder stage's dropout follows the
// attention).  Three launches — finalize, apply + pool, FC — where the unfused chain has four and one more read of y.
extern "C" int DCS_SYM(dcs_cbn_fwd_slabs_pool)(const act_t* x, act_t* y, const float* weight, const float* bias,
                                               float* running_mean, float* running_covar, float* stats_out, float* coef_out,
                                               const float* part, int rows, int stride, const float* pivot, void* pool_part,
                                               long pool_bytes, int B, long HW, int C, float eps, float momentum, int act,
                                               dcs_stream_t stream) {
    int G;
    if (!x || !y || !stats_out || !coef_out || !part || !pivot || !pool_part || B <= 0 || B > 65535 || HW <= 0 ||
        !att::geom(C, &G) || rows < 1 || stride < rows)
        return DCS_ERR_BADARG;
    if ((weight == nullptr) != (bias == nullptr)) return DCS_ERR_BADARG;
    if (act != DCS_ACT_NONE && act != DCS_ACT_RELU && act != DCS_ACT_LRELU) return DCS_ERR_BADARG;
    const int nch = att::ca_chunks(HW, G);
    if (pool_bytes < (long)B * nch * C * 2 * (long)sizeof(double)) return DCS_ERR_WORKSPACE;
    hipStream_t s = dcs_stream(stream);
    const long P = (long)B * HW;
    DCS_LAUNCH((cbn_finalize_kernel<float, float>), dim3(C), dim3(256), 0, s, pivot, part, rows, stride, weight, bias,
               running_mean, running_covar, stats_out, coef_out, P, C, eps, momentum, 1);
    DCS_CHECK_LAUNCH();
#define DCS_CBN_AP(A) DCS_LAUNCH((cbn_apply_pool_kernel<A>), dim3(nch, B), dim3(kThreads), 0, s, x, y, (const float*)coef_out, \
                                 (double*)pool_part, HW, C, G)
    if (act == DCS_ACT_RELU) DCS_CBN_AP(DCS_ACT_RELU);
    else if (act == DCS_ACT_LRELU) DCS_CBN_AP(DCS_ACT_LRELU);
    else DCS_CBN_AP(DCS_ACT_NONE);
#undef DCS_CBN_AP
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

#ifndef DCS_ACT_BF16
// Real BatchNorm2d (+ ReLU / LeakyReLU) of float[P][Cr]; Cr even, or Cr == 1 with P even (see rbn_finalize_kernel).
extern "C" int dcs_rbn_fwd(const float* x, float* y, const float* weight, const float* bias, float* running_mean,
                           float* running_var, float* stats_out, float* coef_out, void* workspace, long workspace_bytes,
                           long P, int Cr, float eps, float momentum, int use_batch_stats, int act, dcs_stream_t stream) {
    const int merge = Cr == 1;
    if (!x || !y || !stats_out || !coef_out || Cr < 1 || (!merge && (Cr & 1)) || (merge && (P & 1))) return DCS_ERR_BADARG;
    const int C = merge ? 1 : Cr / 2;
    const long Pc = merge ? P / 2 : P;
    CbnGeom g;
    if (!cbn_geom(Pc, C, &g)) return DCS_ERR_BADARG;
    if ((weight == nullptr) != (bias == nullptr) || use_batch_stats < 0 || use_batch_stats > 1) return DCS_ERR_BADARG;
    if (use_batch_stats == 0 && (!running_mean || !running_var)) return DCS_ERR_BADARG;
    if (act != DCS_ACT_NONE && act != DCS_ACT_RELU && act != DCS_ACT_LRELU) return DCS_ERR_BADARG;
    hipStream_t s = dcs_stream(stream);
    if (use_batch_stats) {
        if (!workspace || workspace_bytes < (long)g.nblocks * C * 5 * (long)sizeof(double)) return DCS_ERR_WORKSPACE;
        DCS_LAUNCH(cbn_stats_kernel, dim3(g.nblocks), dim3(kThreads), 0, s, x, (double*)workspace, Pc, C, g.vec_per_row,
                   g.rows_per_iter);
        DCS_CHECK_LAUNCH();
    }
    DCS_LAUNCH(rbn_finalize_kernel, dim3(C), dim3(64), 0, s, x, (const double*)workspace, g.nblocks, weight, bias,
               running_mean, running_var, stats_out, coef_out, Pc, C, merge, eps, momentum, use_batch_stats);
    DCS_CHECK_LAUNCH();
    long iters = (C == 1) ? (Pc / 2 + kThreads - 1) / kThreads : (Pc + g.rows_per_iter - 1) / g.rows_per_iter;
    long nb = (iters + DCS_CBN_APP_IT - 1) / DCS_CBN_APP_IT;
    const int grid = (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
#define DCS_RBN_APPLY(A)                                                                                        \
    DCS_LAUNCH((cbn_apply_kernel<A, false>), dim3(grid), dim3(kThreads), 0, s, x, y, coef_out, Pc, C, g.vec_per_row, \
               g.rows_per_iter, 0.f, (uint64_t)0, (const uint64_t*)nullptr)
    if (act == DCS_ACT_RELU) DCS_RBN_APPLY(DCS_ACT_RELU);
    else if (act == DCS_ACT_LRELU) DCS_RBN_APPLY(DCS_ACT_LRELU);
    else DCS_RBN_APPLY(DCS_ACT_NONE);
#undef DCS_RBN_APPLY
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
#endif
