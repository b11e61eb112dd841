// conv_enc0.hip — the first encoder convolution (ComplexConv2d 1 -> 8 channels, 7x7, stride 2, pad 3: c_network.py:107-112,
// config.py:83-91) forward on the MFMA units.
//
// With ONE input channel the implicit GEMM of conv_mfma.hip has nothing to contract over per tap (K = 2), so the layer ran
// on the VALU kernel of conv_direct.hip (29-47 TFLOP/s).  Here the 49 taps ARE the K axis:
//     D[p][(co, re|im)] = sum_{tap, part} A[p][(tap, part)] B[(tap, part)][(co, re|im)],   K = 49 x 2 = 98 (padded to 100), N = 16
//   A[p][(tap, re|im)] = x_{re|im}[2 p + tap - pad]          read straight from the LDS input patch (ds_read_b32; the four
//                                                            k of a step are two adjacent taps x (re, im): 64 distinct banks)
//   B                  = the 2x2 real embedding of the 49 x 8 complex weights: 25 fragments of one float per lane, held in
//                        REGISTERS for the whole workgroup (built from the direct panel at kernel start)
// v_mfma_f32_16x16x4_f32: 16 pixels x 16 columns x 4 k per instruction, 25 per 16-pixel tile: 98 % of the issued MACs
// are useful.  A workgroup = 8 x 32 output pixels (16 M-tiles, four per wave) over a 21 x 69 input patch (11.6 KB).
// Epilogue as everywhere: bias, optional folded eval-mode CBN (conv_common.h), activation; 64-byte rows per pixel.
#include "conv_common.h"

namespace {

constexpr int K7 = 7, TAPS = 49, KSTEPS = 25, TR = 8, TC = 32, PR = (TR - 1) * 2 + K7, PC = (TC - 1) * 2 + K7, PCP = PC + 1;
typedef float f32x4v __attribute__((ext_vector_type(4)));

// patch offset (in complex elements) of tap t relative to a pixel's window origin
__host__ __device__ constexpr int tap_off(int t) { return (t / K7) * PCP + (t % K7); }

__global__ __launch_bounds__(256) void cconv_enc0_kernel(conv::Args a) {
    __shared__ float2 patch[PR * PCP];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int b = blockIdx.y;
    const int oy0 = ((int)blockIdx.x / a.tiles_w) * TR, ox0 = ((int)blockIdx.x % a.tiles_w) * TC;

    // B fragments: lane (col n = li, k = 4 s + kg) of step s; k -> (tap = k >> 1, part = k & 1)
    float bf[KSTEPS];
    {
        const int co = li >> 1, im = li & 1, part = kg & 1;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int tap = 2 * s + (kg >> 1);
            float v = 0.f;
            if (tap < TAPS) {
                const float2 w = a.wp[tap * a.Cout + co];                  // direct panel complex[tap][ci = 0][co]
                v = part == 0 ? (im ? w.y : w.x) : (im ? w.x : -w.y);     // (re: w_r, -w_i ; im: w_i, w_r)
            }
            bf[s] = v;
        }
    }
    // input patch (single channel): patch[r][c] = x[b][2 oy0 - 3 + r][2 ox0 - 3 + c]
    const float2* xb = a.x1 + (long)b * a.Hin * a.Win;
    // (all of a thread's loads are issued before its first LDS write: a load -> wait -> write loop pays one memory
    //  round trip per element)
    constexpr int NL = (PR * PC + 255) / 256;
    float2 pv_[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = t + 256 * k, r = i / PC, c = i % PC;
        const int y = 2 * oy0 - a.pad_f + r, x = 2 * ox0 - a.pad_t + c;
        pv_[k] = (i < PR * PC && y >= 0 && y < a.Hin && x >= 0 && x < a.Win) ? xb[(long)y * a.Win + x] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = t + 256 * k, r = i / PC, c = i % PC;
        if (i < PR * PC) patch[r * PCP + c] = pv_[k];
    }
    __syncthreads();

    const float* pf = reinterpret_cast<const float*>(patch);
    const int part = kg & 1, odd = kg >> 1;
    const float* biasf = reinterpret_cast<const float*>(a.bias);
    const float bv = biasf ? biasf[li] : 0.f;
    float c_re = 1.f, c_im = 0.f, c_add = 0.f;                            // folded eval-mode CBN (conv_mfma.hip)
    if (a.coef) {
        const float* q = a.coef + 6 * (li >> 1);
        if (li & 1) { c_re = q[2]; c_im = q[3]; c_add = q[5]; } else { c_re = q[0]; c_im = q[1]; c_add = q[4]; }
    }
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        const int mt = wave * 4 + i;                                       // 16 M-tiles: row mt / 2, 16 columns each
        const int py = mt >> 1, px = (mt & 1) * 16 + li;
        const float* base = pf + ((py * 2) * PCP + px * 2) * 2 + part;     // this lane's pixel window, its re or im plane
        f32x4v acc = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            // taps 2s (kg 0,1) and 2s+1 (kg 2,3); tap 49 (step 24, odd) is the zero pad: any in-range address, B is 0 there
            const int o0 = tap_off(2 * s), o1 = tap_off(2 * s + 1 < TAPS ? 2 * s + 1 : 2 * s);
            const float av = base[(odd ? o1 : o0) * 2];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bf[s], acc, 0, 0, 0);
        }
        // C/D: col = li, rows kg*4 + r -> pixel (py, (mt & 1) * 16 + kg*4 + r)
        const int oy = oy0 + py;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ox = ox0 + (mt & 1) * 16 + kg * 4 + r;
            float v = acc[r] + bv;
            if (a.coef) {
                const float pv = dcs_dpp_term<0xB1, 0xf>(v);
                v = (li & 1) ? fmaf(c_re, pv, fmaf(c_im, v, c_add)) : fmaf(c_re, v, fmaf(c_im, pv, c_add));
            }
            if (oy < a.Hout && ox < a.Wout)
                reinterpret_cast<float*>(a.y)[(((long)b * a.Hout + oy) * a.Wout + ox) * 16 + li] = dcs_act(v, a.act);
        }
    }
}

}  // namespace

bool dcs_conv_enc0_ok(const conv::Args& a) {
    return a.C1 == 1 && a.C2 == 0 && a.Cout == 8 && a.kh == K7 && a.kw == K7 && a.sf == 2 && a.st == 2 && a.pad_f == 3 &&
           a.pad_t == 3 && a.up_f == 1 && a.up_t == 1 && !a.zero_ins && a.x1 && a.wp && a.y && a.B <= 65535;
}

int dcs_conv_enc0_launch(conv::Args a, hipStream_t stream) {
    if (!dcs_conv_enc0_ok(a)) return DCS_ERR_BADARG;
    a.Hout = (a.Hin + 2 * a.pad_f - K7) / 2 + 1;
    a.Wout = (a.Win + 2 * a.pad_t - K7) / 2 + 1;
    a.tiles_w = (a.Wout + TC - 1) / TC;
    a.tiles_h = (a.Hout + TR - 1) / TR;
    hipLaunchKernelGGL(cconv_enc0_kernel, dim3(a.tiles_w * a.tiles_h, a.B), dim3(256), 0, stream, a);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
