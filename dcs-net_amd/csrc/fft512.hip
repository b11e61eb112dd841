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
