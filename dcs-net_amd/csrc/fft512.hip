// fft512.hip — the 512-point real FFT pair of the waveform synthesis (n_fft = 512: config.py:57), one wavefront per frame.
//
// torch.istft / mag_phase_2_wave (network_functions.py:140-150) needs, per synthesised signal, an inverse real FFT of
// B*T frames of 257 bins, and its backward a forward real FFT of B*T frames of 512 samples.  Through torch.fft that is
// rocFFT plus two device copies per transform (the c2r clone of its input, staging) and a separate real-to-complex post
// pass: 24 + 25 us per signal for 16.8 MB.  Here:
//   * a real transform of length N = 512 is a complex one of length M = 256 plus an O(M) twist: forward
//       z[m] = g[2m] + j g[2m+1],  Zf = DFT_M(z),  G[k] = 1/2 [(Zf[k] + conj Zf[M-k]) - j W^k (Zf[k] - conj Zf[M-k])],  W = e^{-2 pi j / N}
//     and inverse (unnormalised; the imaginary parts of the DC and Nyquist bins are ignored as every c2r transform does)
//       Z[k] = (X[k] + conj X[M-k]) + j W^{-k} (X[k] - conj X[M-k]),  z = IDFT_M(Z) (no 1/M),  y[2m] = Re z[m], y[2m+1] = Im z[m];
//   * the 256-point complex transform is four radix-4 Stockham passes over two LDS buffers: 64 lanes x one butterfly per
//     pass, twiddles from a 256-entry table built once per workgroup with sincospif;
//   * four frames per 256-thread workgroup, frames contiguous in memory on both sides (no staging copies).
#include "dcs_common.h"

namespace {

constexpr int N = 512, M = 256, kFramesPerWg = 4;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 mulj(float2 a) { return make_float2(-a.y, a.x); }      // j * a

// tw[i] = e^{+2 pi j i / 256}; a forward transform conjugates on use
__device__ __forceinline__ void build_twiddles(float2* tw, float2* tw512) {
    const int t = threadIdx.x;
    float s, c;
    sincospif(2.f * (float)t / (float)M, &s, &c);
    tw[t] = make_float2(c, s);
    sincospif(2.f * (float)t / (float)N, &s, &c);
    tw512[t] = make_float2(c, s);                            // e^{+2 pi j t / 512}, t < 256
}

// In-place (result back in a) 256-point complex DFT over LDS buffers a, b of one wavefront; INV: e^{+...}, unnormalised.
// Stockham radix 4: pass Ns = 1, 4, 16, 64; lane j: inputs a[j + 64 r] * tw^(r (j % Ns) 64 / Ns), outputs b[expand(j) + r Ns].
template <bool INV>
__device__ __forceinline__ void fft256(float2* a, float2* b, const float2* tw, int lane) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int Ns = 1 << (2 * pass);
        const int k = lane & (Ns - 1);
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float2 w = tw[(r * k * (64 / Ns)) & (M - 1)];
            if (!INV) w.y = -w.y;
            v[r] = cmul(a[lane + 64 * r], w);
        }
        const float2 s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]), s13 = cadd(v[1], v[3]), d13 = csub(v[1], v[3]);
        const float2 jd = INV ? mulj(d13) : make_float2(d13.y, -d13.x);            // +j d13 (inverse) or -j d13 (forward)
        const int j0 = ((lane >> (2 * pass)) << (2 * pass + 2)) + k;               // (lane / Ns) * 4 Ns + k
        // (b was last READ one pass ago, before that pass's barrier: no barrier needed before overwriting it)
        b[j0] = cadd(s02, s13);
        b[j0 + Ns] = cadd(d02, jd);
        b[j0 + 2 * Ns] = csub(s02, s13);
        b[j0 + 3 * Ns] = csub(d02, jd);
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
    }
}

// X complex[frames][257] -> y float[frames][512]
__global__ __launch_bounds__(256) void irfft512_kernel(const float2* __restrict__ X, float2* __restrict__ y, long frames) {
    __shared__ float2 tw[M], tw512[M];
    __shared__ float2 buf[kFramesPerWg][2][M];
    build_twiddles(tw, tw512);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long f = (long)blockIdx.x * kFramesPerWg + wave;
    const bool live = f < frames;
    float2* a = buf[wave][0];
    float2* b = buf[wave][1];
    __syncthreads();
    if (live) {
        const float2* xf = X + f * (M + 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            float2 xk = xf[k], xm = xf[M - k];
            if (k == 0) { xk.y = 0.f; xm.y = 0.f; }          // DC and Nyquist: imaginary parts ignored
            const float2 A = cadd(xk, cconj(xm));
            const float2 Bv = cmul(csub(xk, cconj(xm)), tw512[k]);
            a[k] = cadd(A, mulj(Bv));
        }
    }
    __syncthreads();
    fft256<true>(a, b, tw, lane);                            // 4 passes: result back in buf[wave][0]
    if (live) {
        float2* yf = y + f * M;                              // float2 = (y[2m], y[2m+1])
#pragma unroll
        for (int r = 0; r < 4; ++r) yf[lane + 64 * r] = a[lane + 64 * r];
    }
}

// g float[frames][512] -> G complex[frames][257]
__global__ __launch_bounds__(256) void rfft512_kernel(const float2* __restrict__ g, float2* __restrict__ G, long frames) {
    __shared__ float2 tw[M], tw512[M];
    __shared__ float2 buf[kFramesPerWg][2][M];
    build_twiddles(tw, tw512);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long f = (long)blockIdx.x * kFramesPerWg + wave;
    const bool live = f < frames;
    float2* a = buf[wave][0];
    float2* b = buf[wave][1];
    if (live) {
        const float2* gf = g + f * M;
#pragma unroll
        for (int r = 0; r < 4; ++r) a[lane + 64 * r] = gf[lane + 64 * r];
    }
    __syncthreads();
    fft256<false>(a, b, tw, lane);
    if (live) {
        float2* Gf = G + f * (M + 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            const float2 zk = a[k], zm = cconj(a[(M - k) & (M - 1)]);
            const float2 e = cadd(zk, zm), o = cmul(csub(zk, zm), cconj(tw512[k]));       // W^k = conj(tw512[k])
            // G = 1/2 (e - j o)
            Gf[k] = make_float2(0.5f * (e.x + o.y), 0.5f * (e.y - o.x));
        }
        if (lane == 0) {                                     // k = M: W^M = -1, Zf[M] = Zf[0]
            const float2 z0 = a[0];
            Gf[M] = make_float2(z0.x - z0.y, 0.f);
        }
    }
}

// The same transform of the ADJOINT of torch.istft's overlap-add (synth.hip, istft_ola_bwd4_kernel), read on the fly (Round 5):
//   g[b][f][k] = scale * w[k] * inv_env[n] * gy[b][n],  n = f hop + k - 256  (0 outside the trimmed signal [0, Lout))
// — the windowed cotangent frames (33.5 MB at [64, 256 frames]) are neither written nor read back; same products in the same
// order, so G is bit for bit what rfft512_kernel makes of the stored frames.  hop even, Lout = hop (T - 1).
__global__ __launch_bounds__(256) void rfft512_ola_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                           const float* __restrict__ inv_env, float2* __restrict__ G, long frames,
                                                           int T, int hop, int Lout, float scale) {
    __shared__ float2 tw[M], tw512[M];
    __shared__ float2 buf[kFramesPerWg][2][M];
    build_twiddles(tw, tw512);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long f = (long)blockIdx.x * kFramesPerWg + wave;
    const bool live = f < frames;
    float2* a = buf[wave][0];
    float2* b = buf[wave][1];
    if (live) {
        const long bi = f / T;
        const int ft = (int)(f - bi * T);
        const float* gs = gy + bi * Lout;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = lane + 64 * r, k = 2 * m, n = ft * hop + k - N / 2;       // n even: (n, n + 1) inside or outside together
            float2 v = make_float2(0.f, 0.f);
            if (n >= 0 && n < Lout) {
                const float2 wv = *reinterpret_cast<const float2*>(w + k), ev = *reinterpret_cast<const float2*>(inv_env + n);
                const float2 g = *reinterpret_cast<const float2*>(gs + n);
                v = make_float2(scale * wv.x * ev.x * g.x, scale * wv.y * ev.y * g.y);
            }
            a[m] = v;
        }
    }
    __syncthreads();
    fft256<false>(a, b, tw, lane);
    if (live) {
        float2* Gf = G + f * (M + 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            const float2 zk = a[k], zm = cconj(a[(M - k) & (M - 1)]);
            const float2 e = cadd(zk, zm), o = cmul(csub(zk, zm), cconj(tw512[k]));
            Gf[k] = make_float2(0.5f * (e.x + o.y), 0.5f * (e.y - o.x));
        }
        if (lane == 0) {
            const float2 z0 = a[0];
            Gf[M] = make_float2(z0.x - z0.y, 0.f);
        }
    }
}

// The inverse transform with torch.istft's synthesis window, overlap-add, envelope division and centre trim behind it in the same
// kernel (Round 5; synth.hip's istft_ola_fwd_kernel restated on frames that never leave LDS):
//   y[b][n] = scale * inv_env[n] * sum_f w[p - f hop] frame[b][f][p - f hop],  p = n + 256, f ascending — the same sum in the same order.
// A workgroup owns kOlaFrames = 16 consecutive frames of one signal (four per wave, one after the other) and writes the
// NC = 16 - R + 1 output chunks of `hop` samples all of whose R = 512 / hop frames it holds (13 of 16 at hop = 128): 1.23 x the
// transform work and spectrum reads instead of a 33.5 MB round trip of the frames through HBM per synthesised batch.
constexpr int kOlaFR = 4, kOlaFrames = 4 * kOlaFR;
__global__ __launch_bounds__(256) void irfft512_ola_kernel(const float2* __restrict__ X, const float* __restrict__ w,
                                                            const float* __restrict__ inv_env, float* __restrict__ y, int T, int hop,
                                                            int Lout, float scale) {
    __shared__ float2 tw[M], tw512[M];
    __shared__ float2 fa[kOlaFrames][M];                     // frame f_first + slot: transformed in place, (y[2m], y[2m+1])
    __shared__ float2 fb[4][M];                              // a wave's second Stockham buffer
    build_twiddles(tw, tw512);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int R = N / hop, NC = kOlaFrames - R + 1;
    const long bi = blockIdx.y;
    const int c0 = (N / 2) / hop + (int)blockIdx.x * NC;     // first output chunk (p-space: p = n + 256)
    const int f_first = c0 - R + 1;
#pragma unroll 1
    for (int i = 0; i < kOlaFR; ++i) {
        const int slot = wave * kOlaFR + i, f = f_first + slot;
        float2* a = fa[slot];
        if (f >= 0 && f < T) {
            const float2* xf = X + (bi * T + f) * (M + 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = lane + 64 * r;
                float2 xk = xf[k], xm = xf[M - k];
                if (k == 0) { xk.y = 0.f; xm.y = 0.f; }      // DC and Nyquist: imaginary parts ignored
                const float2 A = cadd(xk, cconj(xm));
                const float2 Bv = cmul(csub(xk, cconj(xm)), tw512[k]);
                a[k] = cadd(A, mulj(Bv));
            }
        }
        __syncthreads();                                     // (first round: the twiddle tables as well)
        fft256<true>(a, fb[wave], tw, lane);                 // frames outside [0, T) transform stale LDS: never read below
    }
    __syncthreads();
    for (int o = threadIdx.x; o < NC * hop; o += 256) {
        const int p = c0 * hop + o, n = p - N / 2;
        if (n >= Lout) break;
        const int lo = p - N + 1;
        const int f_lo = lo > 0 ? (lo + hop - 1) / hop : 0;
        const int hi = p / hop, f_hi = hi < T - 1 ? hi : T - 1;
        float sacc = 0.f;
        for (int f = f_lo; f <= f_hi; ++f) {
            const int k = p - f * hop;
            sacc = fmaf(w[k], reinterpret_cast<const float*>(fa[f - f_first])[k], sacc);
        }
        y[bi * Lout + n] = scale * inv_env[n] * sacc;
    }
}

}  // namespace

// X complex[frames][257] (one-sided spectra) -> y float[frames][512], unnormalised inverse real FFT
extern "C" int dcs_irfft512_frames(const float* X, float* y, long frames, dcs_stream_t stream) {
    if (!X || !y || frames <= 0 || frames > (1L << 31)) return DCS_ERR_BADARG;
    DCS_LAUNCH(irfft512_kernel, dim3((unsigned)((frames + kFramesPerWg - 1) / kFramesPerWg)), dim3(256), 0,
                       dcs_stream(stream), (const float2*)X, (float2*)y, frames);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// g float[frames][512] -> G complex[frames][257], forward real FFT (no scaling)
extern "C" int dcs_rfft512_frames(const float* g, float* G, long frames, dcs_stream_t stream) {
    if (!g || !G || frames <= 0 || frames > (1L << 31)) return DCS_ERR_BADARG;
    DCS_LAUNCH(rfft512_kernel, dim3((unsigned)((frames + kFramesPerWg - 1) / kFramesPerWg)), dim3(256), 0,
                       dcs_stream(stream), (const float2*)g, (float2*)G, frames);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// gy float[B][hop (T - 1)] (cotangent of dcs_istft_ola_fwd's output) -> G complex[B * T][257] = rfft of dcs_istft_ola_bwd's frames,
// without storing them (n_fft = 512; hop even, a divisor-free requirement otherwise: any 0 < hop <= 512)
extern "C" int dcs_rfft512_ola_frames(const float* gy, const float* window, const float* inv_env, float* G, int B, int T, int hop,
                                      float scale, dcs_stream_t stream) {
    if (!gy || !window || !inv_env || !G || B <= 0 || T < 2 || hop <= 0 || hop > N || (hop & 1)) return DCS_ERR_BADARG;
    const long frames = (long)B * T;
    if (frames > (1L << 31)) return DCS_ERR_BADARG;
    DCS_LAUNCH(rfft512_ola_kernel, dim3((unsigned)((frames + kFramesPerWg - 1) / kFramesPerWg)), dim3(256), 0, dcs_stream(stream),
               gy, window, inv_env, (float2*)G, frames, T, hop, hop * (T - 1), scale);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// X complex[B * T][257] -> y float[B][hop (T - 1)] = dcs_istft_ola_fwd(dcs_irfft512_frames(X)) without storing the frames
// (n_fft = 512; hop a divisor of the 256-sample centre trim with at most 8 overlapping frames: 64, 128, 256)
extern "C" int dcs_irfft512_ola_frames(const float* X, const float* window, const float* inv_env, float* y, int B, int T, int hop,
                                       float scale, dcs_stream_t stream) {
    if (!X || !window || !inv_env || !y || B <= 0 || B > 65535 || T < 2 || hop < 64 || hop > N / 2 || ((N / 2) % hop) != 0) return DCS_ERR_BADARG;
    const int R = N / hop, NC = kOlaFrames - R + 1;
    const int chunks = T - 1;                                // output chunks of `hop` samples
    DCS_LAUNCH(irfft512_ola_kernel, dim3((unsigned)((chunks + NC - 1) / NC), B), dim3(256), 0, dcs_stream(stream), (const float2*)X,
               window, inv_env, y, T, hop, hop * (T - 1), scale);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
