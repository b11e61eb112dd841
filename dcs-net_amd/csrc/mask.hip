// mask.hip — bounded complex ratio mask and its subtractive application
// (network_functions.py:62-96 and :240-243), element-wise over complex[n].
//
// bound_cRM in the reference is ~12 transcendental element-wise launches (abs, tanh, 2x atan2,
// 4x sin/cos).  cos(atan2(y,x)) = x/hypot(x,y) and sin(atan2(y,x)) = y/hypot(x,y), with
// atan2(0,0) = 0, so each phase round trip is one hypot and two divides: no trigonometry, and
// closer to the exact value than the reference's own fp32 trig chain (tests bound the
// difference at 2e-6 absolute on a mask whose modulus is < 1).
// HBM-bound: bound = 8 B read + 8 B written per element; bound+apply = 16 B read + 24 B written
// (10.2 KB per 256-bin frame).
#include "dcs_common.h"

namespace {

constexpr int kThreads = 256;

// 1 / h by v_rcp_f32 (1 ulp) and multiplies: an IEEE fp32 division is ~10 instructions, and the double bound's forward +
// backward per element holds 36 of them — the backward kernel ran at 3x its memory time on the VALU
__device__ __forceinline__ float rcp1(float h) { return __builtin_amdgcn_rcpf(h); }

__device__ __forceinline__ float2 unit_dir(float x, float y) {
    const float h = hypotf(x, y);
    if (h == 0.f) return make_float2(1.f, 0.f);       // atan2(0, 0) = 0
    const float ih = rcp1(h);
    return make_float2(x * ih, y * ih);
}

__device__ __forceinline__ float2 bound_one(float mr, float mi, float eps) {
    const float m = tanhf(hypotf(mr, mi));
    const float2 d1 = unit_dir(mr + eps, mi);
    const float2 d2 = unit_dir(m * d1.x + eps, m * d1.y);
    return make_float2(m * d2.x, m * d2.y);
}

__global__ __launch_bounds__(kThreads) void bound_crm_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                              long n, float eps) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const float2 v = in[i];
        out[i] = bound_one(v.x, v.y, eps);
    }
}

// PRE: Min is the network's RAW last-stage output: the first bound_cRM (c_network.py:225) runs here too, so the predicted
// mask is bounded twice in this one pass (network_functions.py:240) and the once-bounded mask never makes a round trip
// through HBM (M1out: optional, for callers that want the network's nominal output as well).
// DROP (with PRE): Min is the last conv's output BEFORE the network's final dropout (c_network.py:221-222): the mask —
// dcs_dropout_fwd's, hash(seed + *seed_dev, float index) — is applied on the way in, so that pass over the tensor goes too.
template <bool PRE, bool DROP>
__global__ __launch_bounds__(kThreads) void bound_mask_apply_kernel(const float2* __restrict__ Y,
                                                                     const float2* __restrict__ Min,
                                                                     float2* __restrict__ M1out,
                                                                     float2* __restrict__ Mout, float2* __restrict__ Nh,
                                                                     float2* __restrict__ Sh, long n, float eps,
                                                                     float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    if (DROP && seed_dev) seed += seed_dev[0];
    const float inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const float2 y = Y[i];
        float2 v = Min[i];
        if (DROP) {
            v.x *= dcs_keep_scale(seed, (uint64_t)(2 * i), drop_p, inv_keep);
            v.y *= dcs_keep_scale(seed, (uint64_t)(2 * i + 1), drop_p, inv_keep);
        }
        if (PRE) {
            v = bound_one(v.x, v.y, eps);
            if (M1out) M1out[i] = v;
        }
        const float2 m = bound_one(v.x, v.y, eps);
        const float nr = y.x * m.x - y.y * m.y;
        const float ni = y.x * m.y + y.y * m.x;
        Mout[i] = m;
        Nh[i] = make_float2(nr, ni);
        Sh[i] = make_float2(y.x - nr, y.y - ni);
    }
}

__global__ __launch_bounds__(kThreads) void crm_kernel(const float2* __restrict__ S, const float2* __restrict__ Y,
                                                        float2* __restrict__ M, long n, float eps) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const float2 s = S[i], y = Y[i];
        const float den = y.x * y.x + y.y * y.y + eps;
        M[i] = make_float2((y.x * s.x + y.y * s.y) / den, (y.x * s.y - y.y * s.x) / den);
    }
}

// d unit_dir(v) / dv applied to a cotangent: (g - u (u.g)) / |v|   (0 at the singular point)
__device__ __forceinline__ float2 unit_dir_bwd(float x, float y, float2 g) {
    const float h = hypotf(x, y);
    if (h == 0.f) return make_float2(0.f, 0.f);
    const float ih = rcp1(h);
    const float ux = x * ih, uy = y * ih;
    const float d = ux * g.x + uy * g.y;
    return make_float2((g.x - ux * d) * ih, (g.y - uy * d) * ih);
}

// cotangent of bound_one's input given the cotangent g of its output
__device__ __forceinline__ float2 bound_one_bwd(float mr, float mi, float eps, float2 g) {
    const float r = hypotf(mr, mi);
    const float m = tanhf(r);
    const float2 d1 = unit_dir(mr + eps, mi);
    const float v2x = m * d1.x + eps, v2y = m * d1.y;
    const float2 d2 = unit_dir(v2x, v2y);
    float gm = g.x * d2.x + g.y * d2.y;                                  // out = m d2
    const float2 gv2 = unit_dir_bwd(v2x, v2y, make_float2(m * g.x, m * g.y));
    gm += gv2.x * d1.x + gv2.y * d1.y;                                   // v2 = m d1 + (eps, 0)
    const float2 gv1 = unit_dir_bwd(mr + eps, mi, make_float2(m * gv2.x, m * gv2.y));
    const float gr = gm * (1.f - m * m);                                 // m = tanh r
    float2 out = gv1;
    if (r > 0.f) { const float ir = rcp1(r); out.x += gr * mr * ir; out.y += gr * mi * ir; }         // r = |M|
    return out;
}

// g_Min for M = bound(M_in), N = Y M, S = Y - N with optional cotangents gM, gN, gS (Y may be NULL
// when only gM is given: plain bound_cRM backward)
// PRE (see the forward kernel): Min is the raw network output; gM1 = optional cotangent of the once-bounded mask.
template <bool PRE, bool DROP>
__global__ __launch_bounds__(kThreads) void bound_mask_apply_bwd_kernel(const float2* __restrict__ Y,
                                                                         const float2* __restrict__ Min,
                                                                         const float2* __restrict__ gM1,
                                                                         const float2* __restrict__ gM,
                                                                         const float2* __restrict__ gN,
                                                                         const float2* __restrict__ gS,
                                                                         float2* __restrict__ gMin, long n, float eps,
                                                                         float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    if (DROP && seed_dev) seed += seed_dev[0];
    const float inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        float2 g = gM ? gM[i] : make_float2(0.f, 0.f);
        if (Y && (gN || gS)) {
            float2 t = gN ? gN[i] : make_float2(0.f, 0.f);
            if (gS) { const float2 s = gS[i]; t.x -= s.x; t.y -= s.y; }
            const float2 y = Y[i];                                       // conj(Y) t
            g.x += y.x * t.x + y.y * t.y;
            g.y += y.x * t.y - y.y * t.x;
        }
        float2 v = Min[i];
        float kx = 1.f, ky = 1.f;
        if (DROP) {
            kx = dcs_keep_scale(seed, (uint64_t)(2 * i), drop_p, inv_keep);
            ky = dcs_keep_scale(seed, (uint64_t)(2 * i + 1), drop_p, inv_keep);
            v.x *= kx; v.y *= ky;
        }
        if (PRE) {
            const float2 m1 = bound_one(v.x, v.y, eps);
            float2 g1 = bound_one_bwd(m1.x, m1.y, eps, g);
            if (gM1) { const float2 e = gM1[i]; g1.x += e.x; g1.y += e.y; }
            const float2 gd = bound_one_bwd(v.x, v.y, eps, g1);
            gMin[i] = make_float2(gd.x * kx, gd.y * ky);
        } else {
            gMin[i] = bound_one_bwd(v.x, v.y, eps, g);
        }
    }
}

// ---- Round 5: mask application fused with the synthesis' polar round trip (train step, configured geometry) --------------------------
// The step function hands both estimates to mag_phase_2_wave (network_functions.py:244-247): |z| e^{j atan2(z_i, z_r + eps)} with a zero
// bin appended, frame-major for the inverse FFT (synth.hip, polar_frames).  Run as two kernels the estimates N_hat, S_hat made a
// round trip through HBM each way (33.5 MB written + read forward; cotangent written + read and the values read again backward, at
// [32,256,256]).  Here one kernel goes from (Y, raw network output D) to the two frame-major spectra [b] = turn(N_hat),
// [B + b] = turn(S_hat), and one kernel from their cotangents back to g_D (everything in between recomputed); a 32 x 32 tile
// is transposed through LDS so that both sides stay coalesced.
__device__ __forceinline__ float2 polar_turn(float2 z, float eps) {
    const float m = hypotf(z.x, z.y);
    const float2 d = unit_dir(z.x + eps, z.y);
    return make_float2(m * d.x, m * d.y);
}
// cotangent of polar_turn's input: (z/|z|) (u.g) + |z| (g - u (u.g)) / |v|,  u = unit(v), v = (z_r + eps, z_i)
__device__ __forceinline__ float2 polar_turn_bwd(float2 z, float eps, float2 g) {
    const float m = hypotf(z.x, z.y);
    const float2 d = unit_dir(z.x + eps, z.y);
    const float dot = d.x * g.x + d.y * g.y;
    float2 o = unit_dir_bwd(z.x + eps, z.y, make_float2(m * g.x, m * g.y));
    if (m > 0.f) { const float im = rcp1(m); o.x += dot * z.x * im; o.y += dot * z.y * im; }
    return o;
}

// (Y, D) element -> the twice-bounded mask m and the estimates n = Y m, s = Y - n; DROP: D = dropout(D_raw)
template <bool DROP>
__device__ __forceinline__ void apply_one(float2 y, float2 v, long i, float eps, float drop_p, float inv_keep, uint64_t seed,
                                          float2& vd, float& kx, float& ky, float2& m1, float2& m, float2& n, float2& sh) {
    kx = 1.f; ky = 1.f;
    if (DROP) {
        kx = dcs_keep_scale(seed, (uint64_t)(2 * i), drop_p, inv_keep);
        ky = dcs_keep_scale(seed, (uint64_t)(2 * i + 1), drop_p, inv_keep);
    }
    vd = make_float2(v.x * kx, v.y * ky);
    m1 = bound_one(vd.x, vd.y, eps);
    m = bound_one(m1.x, m1.y, eps);
    n = make_float2(y.x * m.x - y.y * m.y, y.x * m.y + y.y * m.x);
    sh = make_float2(y.x - n.x, y.y - n.y);
}

// grid (ceil(T/32), ceil(Fp/32), B); block 32 x 8.  Y, D: complex[B][F][T]; out: complex[2B][T][Fp] (bins >= F are zero);
// Mout: optional complex[B][F][T] (the twice-bounded mask)
template <bool DROP>
__global__ __launch_bounds__(kThreads) void bound2_apply_polar_frames_kernel(const float2* __restrict__ Y, const float2* __restrict__ D,
                                                                              float2* __restrict__ Mout, float2* __restrict__ out,
                                                                              int B, int F, int Fp, int T, float eps, float drop_p,
                                                                              uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    __shared__ float2 tn[32][33], ts[32][33];
    if (DROP && seed_dev) seed += seed_dev[0];
    const float inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
    const long b = blockIdx.z;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = f0 + ty + 8 * r, t = t0 + tx;
        float2 on = make_float2(0.f, 0.f), os = on;
        if (f < F && t < T) {
            const long i = (b * F + f) * T + t;
            float2 vd, m1, m, n, sh; float kx, ky;
            apply_one<DROP>(Y[i], D[i], i, eps, drop_p, inv_keep, seed, vd, kx, ky, m1, m, n, sh);
            if (Mout) Mout[i] = m;
            on = polar_turn(n, eps); os = polar_turn(sh, eps);
        }
        tn[ty + 8 * r][tx] = on; ts[ty + 8 * r][tx] = os;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r, f = f0 + tx;
        if (t < T && f < Fp) {
            out[(b * T + t) * Fp + f] = tn[tx][ty + 8 * r];
            out[((b + B) * T + t) * Fp + f] = ts[tx][ty + 8 * r];
        }
    }
}

// g: complex[2B][T][Fp] cotangent of the two spectra (herm: as the plain rfft of the cotangent of an unnormalised inverse real FFT's
// output — every bin but DC and Nyquist counts twice); gM: optional cotangent of the mask; gD: complex[B][F][T]
template <bool DROP>
__global__ __launch_bounds__(kThreads) void bound2_apply_polar_frames_bwd_kernel(const float2* __restrict__ Y, const float2* __restrict__ D,
                                                                                  const float2* __restrict__ g, const float2* __restrict__ gM,
                                                                                  float2* __restrict__ gD, int B, int F, int Fp, int T,
                                                                                  float eps, int herm, float drop_p, uint64_t seed,
                                                                                  const uint64_t* __restrict__ seed_dev) {
    __shared__ float2 tn[32][33], ts[32][33];
    if (DROP && seed_dev) seed += seed_dev[0];
    const float inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
    const long b = blockIdx.z;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r, f = f0 + tx;
        const bool in = t < T && f < F;
        tn[tx][ty + 8 * r] = in ? g[(b * T + t) * Fp + f] : make_float2(0.f, 0.f);
        ts[tx][ty + 8 * r] = in ? g[((b + B) * T + t) * Fp + f] : make_float2(0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = f0 + ty + 8 * r, t = t0 + tx;
        if (f < F && t < T) {
            const long i = (b * F + f) * T + t;
            const float2 y = Y[i];
            float2 vd, m1, m, n, sh; float kx, ky;
            apply_one<DROP>(y, D[i], i, eps, drop_p, inv_keep, seed, vd, kx, ky, m1, m, n, sh);
            float2 gon = tn[ty + 8 * r][tx], gos = ts[ty + 8 * r][tx];
            if (herm && f > 0 && f < Fp - 1) { gon.x *= 2.f; gon.y *= 2.f; gos.x *= 2.f; gos.y *= 2.f; }
            const float2 gn = polar_turn_bwd(n, eps, gon), gs = polar_turn_bwd(sh, eps, gos);
            const float2 tt = make_float2(gn.x - gs.x, gn.y - gs.y);           // n = Y m, s = Y - n
            float2 gm = make_float2(y.x * tt.x + y.y * tt.y, y.x * tt.y - y.y * tt.x);       // conj(Y) t
            if (gM) { const float2 e = gM[i]; gm.x += e.x; gm.y += e.y; }
            const float2 g1 = bound_one_bwd(m1.x, m1.y, eps, gm);
            const float2 gd = bound_one_bwd(vd.x, vd.y, eps, g1);
            gD[i] = make_float2(gd.x * kx, gd.y * ky);
        }
    }
}

inline int ew_grid(long n) {
    long nb = (n + kThreads * 4 - 1) / (kThreads * 4);
    return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
}

}  // namespace

extern "C" int dcs_bound_crm_fwd(const float* M_raw, float* M_out, long n, float eps, dcs_stream_t stream) {
    if (!M_raw || !M_out || n <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH(bound_crm_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream),
                       (const float2*)M_raw, (float2*)M_out, n, eps);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_bound_mask_apply_fwd(const float* Y, const float* M_in, float* M_out, float* N_hat, float* S_hat,
                                        long n, float eps, dcs_stream_t stream) {
    if (!Y || !M_in || !M_out || !N_hat || !S_hat || n <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH((bound_mask_apply_kernel<false, false>), dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream),
                       (const float2*)Y, (const float2*)M_in, (float2*)nullptr, (float2*)M_out, (float2*)N_hat, (float2*)S_hat, n, eps,
                       0.f, (uint64_t)0, (const uint64_t*)nullptr);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// The network's last bound_cRM (c_network.py:225) and the step function's second one + multiply + subtract
// (network_functions.py:240-243) in ONE pass over the raw last-stage output D: M1 = bound(D) (optional output, NULL: not
// stored), M = bound(M1), N_hat = Y M, S_hat = Y - N_hat.
extern "C" int dcs_bound2_mask_apply_fwd(const float* Y, const float* D_raw, float* M1_out, float* M_out, float* N_hat,
                                         float* S_hat, long n, float eps, float drop_p, unsigned long long seed,
                                         const unsigned long long* seed_dev, dcs_stream_t stream) {
    if (!Y || !D_raw || !M_out || !N_hat || !S_hat || n <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return DCS_ERR_BADARG;
    if (drop_p > 0.f)
        DCS_LAUNCH((bound_mask_apply_kernel<true, true>), dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream),
                   (const float2*)Y, (const float2*)D_raw, (float2*)M1_out, (float2*)M_out, (float2*)N_hat, (float2*)S_hat, n, eps,
                   drop_p, (uint64_t)seed, (const uint64_t*)seed_dev);
    else
        DCS_LAUNCH((bound_mask_apply_kernel<true, false>), dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream),
                   (const float2*)Y, (const float2*)D_raw, (float2*)M1_out, (float2*)M_out, (float2*)N_hat, (float2*)S_hat, n, eps,
                   0.f, (uint64_t)0, (const uint64_t*)nullptr);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// Cotangent of D_raw given any of g_M1 (once-bounded mask), g_M, g_N, g_S.
extern "C" int dcs_bound2_mask_apply_bwd(const float* Y, const float* D_raw, const float* g_M1, const float* g_M,
                                         const float* g_N, const float* g_S, float* g_D, long n, float eps, float drop_p,
                                         unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream) {
    if (!D_raw || !g_D || n <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return DCS_ERR_BADARG;
    if ((g_N || g_S) && !Y) return DCS_ERR_BADARG;
    if (drop_p > 0.f)
        DCS_LAUNCH((bound_mask_apply_bwd_kernel<true, true>), dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream),
                   (const float2*)Y, (const float2*)D_raw, (const float2*)g_M1, (const float2*)g_M, (const float2*)g_N,
                   (const float2*)g_S, (float2*)g_D, n, eps, drop_p, (uint64_t)seed, (const uint64_t*)seed_dev);
    else
        DCS_LAUNCH((bound_mask_apply_bwd_kernel<true, false>), dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream),
                   (const float2*)Y, (const float2*)D_raw, (const float2*)g_M1, (const float2*)g_M, (const float2*)g_N,
                   (const float2*)g_S, (float2*)g_D, n, eps, 0.f, (uint64_t)0, (const uint64_t*)nullptr);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_bound_mask_apply_bwd(const float* Y, const float* M_in, const float* g_M, const float* g_N,
                                        const float* g_S, float* g_Min, long n, float eps, dcs_stream_t stream) {
    if (!M_in || !g_Min || n <= 0) return DCS_ERR_BADARG;
    if ((g_N || g_S) && !Y) return DCS_ERR_BADARG;
    DCS_LAUNCH((bound_mask_apply_bwd_kernel<false, false>), dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream),
                       (const float2*)Y, (const float2*)M_in, (const float2*)nullptr, (const float2*)g_M, (const float2*)g_N,
                       (const float2*)g_S, (float2*)g_Min, n, eps, 0.f, (uint64_t)0, (const uint64_t*)nullptr);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_crm_fwd(const float* S, const float* Y, float* M, long n, float eps, dcs_stream_t stream) {
    if (!S || !Y || !M || n <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH(crm_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream), (const float2*)S,
                       (const float2*)Y, (float2*)M, n, eps);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// (Y, raw last-stage output D) -> the frame-major, polar-turned spectra of both estimates for the synthesis (Round 5; see the kernels).
// out complex[2B][T][Fp] (rows [0, B): Y (.) M, rows [B, 2B): Y - Y (.) M; Fp >= F, bins >= F zero); M_out: optional.
extern "C" int dcs_bound2_apply_polar_frames_fwd(const float* Y, const float* D_raw, float* M_out, float* out, int B, int F, int Fp,
                                                 int T, float eps, float drop_p, unsigned long long seed,
                                                 const unsigned long long* seed_dev, dcs_stream_t stream) {
    if (!Y || !D_raw || !out || B <= 0 || B > 65535 || F <= 0 || Fp < F || T <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return DCS_ERR_BADARG;
    const dim3 grid((T + 31) / 32, (Fp + 31) / 32, B);
    if (drop_p > 0.f)
        DCS_LAUNCH(bound2_apply_polar_frames_kernel<true>, grid, dim3(kThreads), 0, dcs_stream(stream), (const float2*)Y,
                   (const float2*)D_raw, (float2*)M_out, (float2*)out, B, F, Fp, T, eps, drop_p, (uint64_t)seed, (const uint64_t*)seed_dev);
    else
        DCS_LAUNCH(bound2_apply_polar_frames_kernel<false>, grid, dim3(kThreads), 0, dcs_stream(stream), (const float2*)Y,
                   (const float2*)D_raw, (float2*)M_out, (float2*)out, B, F, Fp, T, eps, 0.f, (uint64_t)0, (const uint64_t*)nullptr);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// g_D from the cotangent g_out of those spectra (complex[2B][T][Fp]; hermitian as in dcs_polar_frames_bwd) and, optionally, g_M
extern "C" int dcs_bound2_apply_polar_frames_bwd(const float* Y, const float* D_raw, const float* g_out, const float* g_M, float* g_D,
                                                 int B, int F, int Fp, int T, float eps, int hermitian, float drop_p,
                                                 unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream) {
    if (!Y || !D_raw || !g_out || !g_D || B <= 0 || B > 65535 || F <= 0 || Fp < F || T <= 0 || !(drop_p >= 0.f && drop_p < 1.f))
        return DCS_ERR_BADARG;
    const dim3 grid((T + 31) / 32, (F + 31) / 32, B);
    if (drop_p > 0.f)
        DCS_LAUNCH(bound2_apply_polar_frames_bwd_kernel<true>, grid, dim3(kThreads), 0, dcs_stream(stream), (const float2*)Y,
                   (const float2*)D_raw, (const float2*)g_out, (const float2*)g_M, (float2*)g_D, B, F, Fp, T, eps, hermitian, drop_p,
                   (uint64_t)seed, (const uint64_t*)seed_dev);
    else
        DCS_LAUNCH(bound2_apply_polar_frames_bwd_kernel<false>, grid, dim3(kThreads), 0, dcs_stream(stream), (const float2*)Y,
                   (const float2*)D_raw, (const float2*)g_out, (const float2*)g_M, (float2*)g_D, B, F, Fp, T, eps, hermitian, 0.f,
                   (uint64_t)0, (const uint64_t*)nullptr);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
