// synth.hip — the waveform-synthesis ends of the step functions (network_functions.py:140-150, :213-221, :244-247),
// i.e. what surrounds the inverse FFT of mag_phase_2_wave / torch.istft:
//
//   dcs_polar_frames_fwd/_bwd   |z| (cos, sin)(atan2(z_i, z_r + eps)), zero-padded by the dropped bin and written
//                               FRAME-MAJOR [B][T][Fp] (each STFT frame's bins contiguous), so the inverse real FFT
//                               is one contiguous batched c2r transform instead of a strided one plus a transposed
//                               copy.  The transpose goes through a 32x33 LDS tile: both sides stay coalesced.
//   dcs_istft_envelope          1 / sum_f w^2 of torch.istft's window-envelope normalisation (depends on T only)
//   dcs_istft_ola_fwd/_bwd      synthesis window, overlap-add, envelope division and the n_fft/2 trim of
//                               torch.istft(center=True) in one pass: each output sample gathers its n_fft/hop frames
//                               (backward: each frame element reads the one sample it fed).
// All HBM-bound; they replace ~10 ATen launches per synthesised signal (complex mul, copies, window mul,
// fill + arange + unfold_backward, slice, div) and their autograd counterparts.
#include "dcs_common.h"

namespace {
constexpr int kThreads = 256;

__device__ __forceinline__ float2 unit_dir(float x, float y) {
    const float h = hypotf(x, y);
    if (h == 0.f) return make_float2(1.f, 0.f);       // atan2(0, 0) = 0
    const float ih = __builtin_amdgcn_rcpf(h);        // (v_rcp_f32 + multiplies instead of IEEE divisions: mask.hip)
    return make_float2(x * ih, y * ih);
}
// d unit_dir(v) / dv applied to a cotangent: (g - u (u.g)) / |v|   (0 at the singular point)
__device__ __forceinline__ float2 unit_dir_bwd(float x, float y, float2 g) {
    const float h = hypotf(x, y);
    if (h == 0.f) return make_float2(0.f, 0.f);
    const float ih = __builtin_amdgcn_rcpf(h);
    const float ux = x * ih, uy = y * ih;
    const float d = ux * g.x + uy * g.y;
    return make_float2((g.x - ux * d) * ih, (g.y - uy * d) * ih);
}

// grid (ceil(T/32), ceil(Fp/32), B); block 32 x 8
__global__ __launch_bounds__(kThreads) void polar_frames_fwd_kernel(const float2* __restrict__ z, float2* __restrict__ out,
                                                                     int F, int Fp, int T, float eps) {
    __shared__ float2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
    const long b = blockIdx.z;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = f0 + ty + 8 * r, t = t0 + tx;
        float2 o = make_float2(0.f, 0.f);
        if (f < F && t < T) {
            const float2 v = z[(b * F + f) * T + t];
            const float m = hypotf(v.x, v.y);
            const float2 d = unit_dir(v.x + eps, v.y);
            o = make_float2(m * d.x, m * d.y);
        }
        tile[ty + 8 * r][tx] = o;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r, f = f0 + tx;
        if (t < T && f < Fp) out[(b * T + t) * Fp + f] = tile[tx][ty + 8 * r];
    }
}

// g_z = (z/|z|) (u.g) + |z| (g - u (u.g)) / |v|,  u = unit(v), v = (z_r + eps, z_i);  g is frame-major
__global__ __launch_bounds__(kThreads) void polar_frames_bwd_kernel(const float2* __restrict__ z, const float2* __restrict__ g,
                                                                     float2* __restrict__ gz, int F, int Fp, int T,
                                                                     float eps, int herm) {
    __shared__ float2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
    const long b = blockIdx.z;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r, f = f0 + tx;
        tile[tx][ty + 8 * r] = (t < T && f < F) ? g[(b * T + t) * Fp + f] : make_float2(0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = f0 + ty + 8 * r, t = t0 + tx;
        if (f < F && t < T) {
            const long i = (b * F + f) * T + t;
            const float2 v = z[i];
            float2 go = tile[ty + 8 * r][tx];
            // herm: g is the plain rfft of the cotangent of an UNNORMALISED inverse real FFT's output; the cotangent of its
            // one-sided input counts every bin but DC and Nyquist twice (Hermitian symmetry)
            if (herm && f > 0 && f < Fp - 1) { go.x *= 2.f; go.y *= 2.f; }
            const float m = hypotf(v.x, v.y);
            const float2 d = unit_dir(v.x + eps, v.y);
            const float dot = d.x * go.x + d.y * go.y;
            float2 o = unit_dir_bwd(v.x + eps, v.y, make_float2(m * go.x, m * go.y));
            if (m > 0.f) { const float im = __builtin_amdgcn_rcpf(m); o.x += dot * v.x * im; o.y += dot * v.y * im; }
            gz[i] = o;
        }
    }
}

// frames covering OLA position p: f in [max(0, ceil((p - n_fft + 1) / hop)), min(T - 1, p / hop)]
__device__ __forceinline__ void frame_range(int p, int n_fft, int hop, int T, int* f_lo, int* f_hi) {
    const int lo = p - n_fft + 1;
    *f_lo = lo > 0 ? (lo + hop - 1) / hop : 0;
    const int hi = p / hop;
    *f_hi = hi < T - 1 ? hi : T - 1;
}

__global__ __launch_bounds__(kThreads) void istft_envelope_kernel(const float* __restrict__ w, float* __restrict__ inv_env,
                                                                   int T, int n_fft, int hop, int Lout) {
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= Lout) return;
    const int p = n + n_fft / 2;
    int f_lo, f_hi;
    frame_range(p, n_fft, hop, T, &f_lo, &f_hi);
    float s = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) { const float v = w[p - f * hop]; s = fmaf(v, v, s); }
    inv_env[n] = 1.f / s;
}

// y[b][n] = scale * inv_env[n] * sum_f w[p - f hop] frames[b][f][p - f hop],  p = n + n_fft/2
__global__ __launch_bounds__(kThreads) void istft_ola_fwd_kernel(const float* __restrict__ frames, const float* __restrict__ w,
                                                                  const float* __restrict__ inv_env, float* __restrict__ y,
                                                                  int T, int n_fft, int hop, int Lout, float scale) {
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= Lout) return;
    const long b = blockIdx.y;
    const int p = n + n_fft / 2;
    int f_lo, f_hi;
    frame_range(p, n_fft, hop, T, &f_lo, &f_hi);
    const float* fr = frames + b * T * n_fft;
    float s = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) {
        const int k = p - f * hop;
        s = fmaf(w[k], fr[(long)f * n_fft + k], s);
    }
    y[b * Lout + n] = scale * inv_env[n] * s;
}

// g_frames[b][f][k] = scale * w[k] * inv_env[n] * g_y[b][n],  n = f hop + k - n_fft/2 (0 outside the trimmed signal)
__global__ __launch_bounds__(kThreads) void istft_ola_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                  const float* __restrict__ inv_env, float* __restrict__ gf,
                                                                  int T, int n_fft, int hop, int Lout, float scale) {
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;        // (f, k) of one batch item
    if (i >= (long)T * n_fft) return;
    const long b = blockIdx.y;
    const int k = (int)(i % n_fft), f = (int)(i / n_fft);
    const int n = f * hop + k - n_fft / 2;
    float v = 0.f;
    if (n >= 0 && n < Lout) v = scale * w[k] * inv_env[n] * gy[b * Lout + n];
    gf[b * T * n_fft + i] = v;
}

// The same, four consecutive k per thread (hop, n_fft / 2 and Lout multiples of 4: a thread's four samples n .. n + 3 are aligned
// and lie together inside or outside the trimmed signal): one float4 load of g_y, w and the envelope, one float4 store, one
// index division per four elements (the scalar form: 8.4 M threads with a 64-bit division and a dword store each, 17.5 us for
// 33 MB — Round 4).
__global__ __launch_bounds__(kThreads) void istft_ola_bwd4_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                   const float* __restrict__ inv_env, float* __restrict__ gf,
                                                                   int T, int n_fft, int hop, int Lout, float scale) {
    const int q4 = n_fft >> 2;                                        // float4 per frame
    const int i = (int)blockIdx.x * kThreads + threadIdx.x;            // (f, k / 4) of one batch item
    if (i >= T * q4) return;
    const long b = blockIdx.y;
    const int f = i / q4, k = (i - f * q4) * 4;
    const int n = f * hop + k - n_fft / 2;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n >= 0 && n < Lout) {
        const float4 wv = *reinterpret_cast<const float4*>(w + k), ev = *reinterpret_cast<const float4*>(inv_env + n);
        const float4 g = *reinterpret_cast<const float4*>(gy + b * Lout + n);
        v = make_float4(scale * wv.x * ev.x * g.x, scale * wv.y * ev.y * g.y, scale * wv.z * ev.z * g.z, scale * wv.w * ev.w * g.w);
    }
    reinterpret_cast<float4*>(gf + b * (long)T * n_fft)[i] = v;
}
}  // namespace

extern "C" int dcs_polar_frames_fwd(const float* z, float* out, int B, int F, int Fp, int T, float eps, dcs_stream_t stream) {
    if (!z || !out || B <= 0 || B > 65535 || F <= 0 || Fp < F || T <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH(polar_frames_fwd_kernel, dim3((T + 31) / 32, (Fp + 31) / 32, B), dim3(kThreads), 0, dcs_stream(stream),
                       (const float2*)z, (float2*)out, F, Fp, T, eps);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_polar_frames_bwd(const float* z, const float* g_out, float* g_z, int B, int F, int Fp, int T, float eps,
                                    int hermitian, dcs_stream_t stream) {
    if (!z || !g_out || !g_z || B <= 0 || B > 65535 || F <= 0 || Fp < F || T <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH(polar_frames_bwd_kernel, dim3((T + 31) / 32, (F + 31) / 32, B), dim3(kThreads), 0, dcs_stream(stream),
                       (const float2*)z, (const float2*)g_out, (float2*)g_z, F, Fp, T, eps, hermitian);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

static bool ola_ok(int B, int T, int n_fft, int hop) {
    return B > 0 && B <= 65535 && T > 0 && n_fft > 1 && !(n_fft & 1) && hop > 0 && hop <= n_fft &&
           (long)hop * (T - 1) > 0;
}

extern "C" int dcs_istft_envelope(const float* window, float* inv_env, int T, int n_fft, int hop, dcs_stream_t stream) {
    if (!window || !inv_env || !ola_ok(1, T, n_fft, hop)) return DCS_ERR_BADARG;
    const int Lout = hop * (T - 1);
    DCS_LAUNCH(istft_envelope_kernel, dim3((Lout + kThreads - 1) / kThreads), dim3(kThreads), 0, dcs_stream(stream),
                       window, inv_env, T, n_fft, hop, Lout);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_istft_ola_fwd(const float* frames, const float* window, const float* inv_env, float* y, int B, int T,
                                 int n_fft, int hop, float scale, dcs_stream_t stream) {
    if (!frames || !window || !inv_env || !y || !ola_ok(B, T, n_fft, hop)) return DCS_ERR_BADARG;
    const int Lout = hop * (T - 1);
    DCS_LAUNCH(istft_ola_fwd_kernel, dim3((Lout + kThreads - 1) / kThreads, B), dim3(kThreads), 0,
                       dcs_stream(stream), frames, window, inv_env, y, T, n_fft, hop, Lout, scale);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_istft_ola_bwd(const float* g_y, const float* window, const float* inv_env, float* g_frames, int B, int T,
                                 int n_fft, int hop, float scale, dcs_stream_t stream) {
    if (!g_y || !window || !inv_env || !g_frames || !ola_ok(B, T, n_fft, hop)) return DCS_ERR_BADARG;
    const int Lout = hop * (T - 1);
    const long per = (long)T * n_fft;
    if (!(hop & 3) && !((n_fft / 2) & 3) && !(Lout & 3) && per / 4 < (1L << 30) &&
        !(((uintptr_t)g_y | (uintptr_t)window | (uintptr_t)inv_env | (uintptr_t)g_frames) & 15))
        DCS_LAUNCH(istft_ola_bwd4_kernel, dim3((unsigned)((per / 4 + kThreads - 1) / kThreads), B), dim3(kThreads), 0,
                           dcs_stream(stream), g_y, window, inv_env, g_frames, T, n_fft, hop, Lout, scale);
    else
        DCS_LAUNCH(istft_ola_bwd_kernel, dim3((unsigned)((per + kThreads - 1) / kThreads), B), dim3(kThreads), 0,
                           dcs_stream(stream), g_y, window, inv_env, g_frames, T, n_fft, hop, Lout, scale);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// ---- SiSNR loss (network_functions.py:30-42) -------------------------------------------------------
// snr_b = 10 log10( |a c|^2 / (|e - a c|^2 + eps) + eps ),  a = <e, c> / (|c|^2 + eps), over the samples of utterance b
// (c = clean, e = estimate).  The reference spells it with ~18 ATen launches forward and ~25 backward on a [32, 8160]
// waveform; here one workgroup per utterance makes two passes over its 2 x 32 KB (L2-resident), and the gradient with
// respect to the estimate is  g_e = k1 c + k2 e  with two scalars per utterance computed in the forward pass.
namespace {
__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = dcs_wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(kThreads) void sisnr_fwd_kernel(const float* __restrict__ clean, const float* __restrict__ est,
                                                              int L, float eps, float* __restrict__ snr,
                                                              float* __restrict__ coef) {
    __shared__ double sh[4];
    const long b = blockIdx.x;
    const float* c = clean + b * L;
    const float* e = est + b * L;
    float dot = 0.f, en = 0.f;
    for (int i = threadIdx.x; i < L; i += kThreads) { const float cv = c[i]; dot = fmaf(e[i], cv, dot); en = fmaf(cv, cv, en); }
    const float D = (float)block_sum((double)dot, sh), E = (float)block_sum((double)en, sh);
    const float a = D / (E + eps);
    float t = 0.f, r = 0.f, rc = 0.f;
    for (int i = threadIdx.x; i < L; i += kThreads) {
        const float cv = c[i], tv = a * cv, rv = e[i] - tv;
        t = fmaf(tv, tv, t); r = fmaf(rv, rv, r); rc = fmaf(rv, cv, rc);
    }
    const float T = (float)block_sum((double)t, sh), R = (float)block_sum((double)r, sh), RC = (float)block_sum((double)rc, sh);
    if (threadIdx.x == 0) {
        const float q = T / (R + eps);
        snr[b] = 10.f * log10f(q + eps);
        // d snr / d e_n = K [ dT_n/(R+eps) - T dR_n/(R+eps)^2 ],  dT_n = 2 a E c_n/(E+eps),  dR_n = 2 r_n - 2 RC c_n/(E+eps)
        const float K = 10.f / (2.302585093f * (q + eps));
        const float ie = 1.f / (E + eps), ir = 1.f / (R + eps);
        const float on_c = K * (2.f * a * E * ie * ir + 2.f * T * RC * ie * ir * ir);   // coefficient of c_n
        const float on_r = -K * 2.f * T * ir * ir;                                      // coefficient of r_n = e_n - a c_n
        coef[2 * b] = on_c - a * on_r;                                                  // g_e = k1 c + k2 e
        coef[2 * b + 1] = on_r;
    }
}

// The same for signals of up to 4 * kThreads * NV samples (L % 4 == 0, 16-byte aligned rows: the train step's 8160-sample
// waveforms): both signals are read ONCE, as float4, all loads in flight together, and kept in registers for the second pass.
// The two-pass form above walks each signal twice with one dword load per thread and trip — 64 dependent memory round trips
// in a workgroup that owns a whole signal: 20.8 us for 4 MB (Round 4).
template <int NV>
__global__ __launch_bounds__(kThreads) void sisnr_fwd_reg_kernel(const float* __restrict__ clean, const float* __restrict__ est,
                                                                  int L, float eps, float* __restrict__ snr,
                                                                  float* __restrict__ coef) {
    __shared__ double sh[4];
    const long b = blockIdx.x;
    const float4* c4 = reinterpret_cast<const float4*>(clean + b * L);
    const float4* e4 = reinterpret_cast<const float4*>(est + b * L);
    const int n4 = L >> 2;
    float4 cv[NV], ev[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = threadIdx.x + kThreads * k;
        const int ic = i < n4 ? i : n4 - 1;                            // (unconditional loads; the tail is zeroed below)
        cv[k] = c4[ic];
        ev[k] = e4[ic];
    }
    float dot = 0.f, en = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (threadIdx.x + kThreads * k >= n4) { cv[k] = make_float4(0.f, 0.f, 0.f, 0.f); ev[k] = cv[k]; }
        dot = fmaf(ev[k].x, cv[k].x, dot); dot = fmaf(ev[k].y, cv[k].y, dot); dot = fmaf(ev[k].z, cv[k].z, dot); dot = fmaf(ev[k].w, cv[k].w, dot);
        en = fmaf(cv[k].x, cv[k].x, en); en = fmaf(cv[k].y, cv[k].y, en); en = fmaf(cv[k].z, cv[k].z, en); en = fmaf(cv[k].w, cv[k].w, en);
    }
    const float D = (float)block_sum((double)dot, sh), E = (float)block_sum((double)en, sh);
    const float a = D / (E + eps);
    float t = 0.f, r = 0.f, rc = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float cs[4] = {cv[k].x, cv[k].y, cv[k].z, cv[k].w}, es[4] = {ev[k].x, ev[k].y, ev[k].z, ev[k].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float tv = a * cs[q], rv = es[q] - tv;
            t = fmaf(tv, tv, t); r = fmaf(rv, rv, r); rc = fmaf(rv, cs[q], rc);
        }
    }
    const float T = (float)block_sum((double)t, sh), R = (float)block_sum((double)r, sh), RC = (float)block_sum((double)rc, sh);
    if (threadIdx.x == 0) {
        const float q = T / (R + eps);
        snr[b] = 10.f * log10f(q + eps);
        const float K = 10.f / (2.302585093f * (q + eps));
        const float ie = 1.f / (E + eps), ir = 1.f / (R + eps);
        const float on_c = K * (2.f * a * E * ie * ir + 2.f * T * RC * ie * ir * ir);
        const float on_r = -K * 2.f * T * ir * ir;
        coef[2 * b] = on_c - a * on_r;
        coef[2 * b + 1] = on_r;
    }
}

// g_est[b][n] = (*g) * scale * (k1_b clean + k2_b est)
__global__ __launch_bounds__(kThreads) void sisnr_bwd_kernel(const float* __restrict__ clean, const float* __restrict__ est,
                                                              const float* __restrict__ coef, const float* __restrict__ g,
                                                              float scale, float* __restrict__ g_est, int L) {
    const long b = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= L) return;
    const float s = g[0] * scale;
    g_est[b * L + i] = s * (coef[2 * b] * clean[b * L + i] + coef[2 * b + 1] * est[b * L + i]);
}

// the configured loss pair on STACKED signals: rows [0, B) = noise, rows [B, 2B) = speech.
//   noise_loss = 1 + alpha mean(snr_n), speech_loss = -alpha mean(snr_s), total = their sum; g_* = upstream gradients of
//   the three outputs (device scalars, any may be null): g_est[b] = (g_noise + g_total) * alpha / B * d snr_b  (b < B),
//                                                                  -(g_speech + g_total) * alpha / B * d snr_b (b >= B)
__global__ __launch_bounds__(kThreads) void sisnr_pair_bwd_kernel(const float* __restrict__ tgt, const float* __restrict__ est,
                                                                   const float* __restrict__ coef, const float* __restrict__ g_noise,
                                                                   const float* __restrict__ g_speech, const float* __restrict__ g_total,
                                                                   float alpha_over_B, float* __restrict__ g_est, int B, int L) {
    const long b = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= L) return;
    const float gt = g_total ? g_total[0] : 0.f;
    const float s = b < B ? ((g_noise ? g_noise[0] : 0.f) + gt) * alpha_over_B : -((g_speech ? g_speech[0] : 0.f) + gt) * alpha_over_B;
    g_est[b * L + i] = s * (coef[2 * b] * tgt[b * L + i] + coef[2 * b + 1] * est[b * L + i]);
}
}  // namespace

namespace {
// the reference's loss assembly for noise_loss_type 6 / speech_loss_type 0 (network_functions.py:168-208) on the two
// per-utterance SiSNR vectors: out = {noise_loss = 1 - alpha * (-mean snr_n), speech_loss = alpha * (-mean snr_s), sum}
__global__ __launch_bounds__(64) void sisnr_losses_kernel(const float* __restrict__ snr_s, const float* __restrict__ snr_n,
                                                           float* __restrict__ out, int B, float alpha, float* __restrict__ skip) {
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < B; i += 64) { a += snr_s[i]; b += snr_n[i]; }
    a = dcs_wave_sum(a) / (float)B;
    b = dcs_wave_sum(b) / (float)B;
    if (threadIdx.x == 0) {
        const float noise_loss = 1.f - alpha * (-b), speech_loss = alpha * (-a);
        out[0] = noise_loss; out[1] = speech_loss; out[2] = noise_loss + speech_loss;
        if (skip) skip[0] = out[2] != out[2] ? 1.f : 0.f;        // the step guard of dcs_step_guard, without its launch
    }
}
}  // namespace

extern "C" int dcs_sisnr_losses_fwd(const float* snr_speech, const float* snr_noise, float* out3, int B, float alpha,
                                    dcs_stream_t stream) {
    if (!snr_speech || !snr_noise || !out3 || B <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH(sisnr_losses_kernel, dim3(1), dim3(64), 0, dcs_stream(stream), snr_speech, snr_noise, out3, B, alpha,
               (float*)nullptr);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_sisnr_losses_guard_fwd(const float* snr_speech, const float* snr_noise, float* out3, int B, float alpha,
                                          float* skip, dcs_stream_t stream) {
    if (!snr_speech || !snr_noise || !out3 || B <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH(sisnr_losses_kernel, dim3(1), dim3(64), 0, dcs_stream(stream), snr_speech, snr_noise, out3, B, alpha, skip);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_sisnr_pair_bwd(const float* target, const float* est, const float* coef, const float* g_noise,
                                  const float* g_speech, const float* g_total, float alpha, float* g_est, int B, int L,
                                  dcs_stream_t stream) {
    if (!target || !est || !coef || !g_est || B <= 0 || 2 * B > 65535 || L <= 0) return DCS_ERR_BADARG;
    if (!g_noise && !g_speech && !g_total) return DCS_ERR_BADARG;
    DCS_LAUNCH(sisnr_pair_bwd_kernel, dim3((L + kThreads - 1) / kThreads, 2 * B), dim3(kThreads), 0, dcs_stream(stream), target,
               est, coef, g_noise, g_speech, g_total, alpha / (float)B, g_est, B, L);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_sisnr_fwd(const float* clean, const float* est, float* snr, float* coef, int B, int L, float eps,
                             dcs_stream_t stream) {
    if (!clean || !est || !snr || !coef || B <= 0 || L <= 0) return DCS_ERR_BADARG;
    if ((L & 3) == 0 && L <= 4 * kThreads * 8 && (((uintptr_t)clean | (uintptr_t)est) & 15) == 0)
        DCS_LAUNCH(sisnr_fwd_reg_kernel<8>, dim3(B), dim3(kThreads), 0, dcs_stream(stream), clean, est, L, eps, snr, coef);
    else
        DCS_LAUNCH(sisnr_fwd_kernel, dim3(B), dim3(kThreads), 0, dcs_stream(stream), clean, est, L, eps, snr, coef);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_sisnr_bwd(const float* clean, const float* est, const float* coef, const float* g, float scale,
                             float* g_est, int B, int L, dcs_stream_t stream) {
    if (!clean || !est || !coef || !g || !g_est || B <= 0 || B > 65535 || L <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH(sisnr_bwd_kernel, dim3((L + kThreads - 1) / kThreads, B), dim3(kThreads), 0, dcs_stream(stream), clean,
                       est, coef, g, scale, g_est, L);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// ---- on-device STFT front end (data.py:104-134) ------------------------------------------------------------
// The reference's DataLoader workers run torch.stft three times per item on the CPU (clean, noise = noisy - clean,
// noisy).  Here a batch of cropped waveforms already on the device becomes the three [B][256][T] complex inputs of the
// train step in three launches around one batched real FFT:
//   stft_frames_kernel   frames[s][b][t][k] = w[k] * x_s[b][reflect(t*hop + k - n_fft/2)]  (torch.stft center=True,
//                        pad_mode='reflect'); s = 0 clean, 1 noise = noisy - clean (subtracted in the time domain, as
//                        data.py:104 does), 2 noisy
//   (rocFFT r2c over the contiguous frames)
//   stft_bins_kernel     out[s][b][f][t] = scale * spec[s][b][t][f + 1], f = 0..n_fft/2-1: drops the DC bin
//                        (data.py:118: [1 : n_fft/2 + 1]), applies normalized = 1/sqrt(n_fft) and transposes to the
//                        network's [B][F][T] layout through a 32x33 LDS tile.
namespace {
__global__ __launch_bounds__(kThreads) void stft_frames_kernel(const float* __restrict__ clean, const float* __restrict__ noisy,
                                                                const float* __restrict__ w, float* __restrict__ frames,
                                                                int B, int L, int T, int n_fft, int hop) {
    const long per = (long)B * T * n_fft;
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= per) return;
    const int k = (int)(i % n_fft);
    const long r = i / n_fft;
    const int t = (int)(r % T), b = (int)(r / T);
    int n = t * hop + k - n_fft / 2;
    if (n < 0) n = -n;                                   // reflect without repeating the edge sample
    if (n >= L) n = 2 * (L - 1) - n;
    const float c = clean[(long)b * L + n], y = noisy[(long)b * L + n], wk = w[k];
    frames[i] = wk * c;
    frames[per + i] = wk * (y - c);
    frames[2 * per + i] = wk * y;
}

// grid (ceil(T/32), ceil(F/32), n_signals * B); spec: complex[SB][T][F + 1]; out: complex[SB][F][T]
__global__ __launch_bounds__(kThreads) void stft_bins_kernel(const float2* __restrict__ spec, float2* __restrict__ out, int F,
                                                              int T, float scale) {
    __shared__ float2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
    const long sb = blockIdx.z;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r, f = f0 + tx;
        float2 v = make_float2(0.f, 0.f);
        if (t < T && f < F) v = spec[(sb * T + t) * (F + 1) + f + 1];
        tile[tx][ty + 8 * r] = make_float2(v.x * scale, v.y * scale);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = f0 + ty + 8 * r, t = t0 + tx;
        if (f < F && t < T) out[(sb * F + f) * T + t] = tile[ty + 8 * r][tx];
    }
}
}  // namespace

extern "C" int dcs_stft_frames_fwd(const float* clean, const float* noisy, const float* window, float* frames, int B, int L,
                                   int T, int n_fft, int hop, dcs_stream_t stream) {
    if (!clean || !noisy || !window || !frames || B <= 0 || L <= n_fft / 2 || T <= 0 || n_fft < 2 || (n_fft & 1) || hop <= 0 ||
        (long)(T - 1) * hop > L)
        return DCS_ERR_BADARG;
    const long per = (long)B * T * n_fft;
    DCS_LAUNCH(stft_frames_kernel, dim3((unsigned)((per + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       dcs_stream(stream), clean, noisy, window, frames, B, L, T, n_fft, hop);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_stft_bins_fwd(const float* spec, float* out, int SB, int T, int F, float scale, dcs_stream_t stream) {
    if (!spec || !out || SB <= 0 || SB > 65535 || T <= 0 || F <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH(stft_bins_kernel, dim3((T + 31) / 32, (F + 31) / 32, SB), dim3(kThreads), 0, dcs_stream(stream),
                       (const float2*)spec, (float2*)out, F, T, scale);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
