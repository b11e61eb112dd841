// elementwise.hip — stand-alone complex activation and nearest upsample, channels-last.
//
// Only the layer-by-layer drop-in surface (dcsnet/complexLayers.py, complexFunctions.py) uses
// these: in C_NETWORK.forward the activation is fused into dcs_cbn_fwd and the upsample into the
// input gather of dcs_cconv2d_fwd, so neither tensor pass exists there.
#include "dcs_common.h"

namespace {
constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void act_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                        int act) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads)
        y[i] = dcs_act(x[i], act);
}

// y[b][oy][ox][c] = x[b][oy/uf][ox/ut][c]; one float2 (complex) per thread-iteration
__global__ __launch_bounds__(kThreads) void upsample_kernel(const float2* __restrict__ x, float2* __restrict__ y,
                                                             int B, int H, int W, int C, int uf, int ut) {
    const long n = (long)B * H * uf * W * ut * C;
    const int Wo = W * ut, Ho = H * uf;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const int c = (int)(i % C);
        long p = i / C;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        y[i] = x[(((long)b * H + oy / uf) * W + ox / ut) * C + c];
    }
}

inline int ew_grid(long n) {
    long nb = (n + kThreads * 4 - 1) / (kThreads * 4);
    return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
}
}  // namespace

extern "C" int dcs_complex_act_fwd(const float* x, float* y, long n_floats, int act, dcs_stream_t stream) {
    if (!x || !y || n_floats <= 0 || act < DCS_ACT_NONE || act > DCS_ACT_SIGMOID) return DCS_ERR_BADARG;
    hipLaunchKernelGGL(act_kernel, dim3(ew_grid(n_floats)), dim3(kThreads), 0, dcs_stream(stream), x, y, n_floats, act);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_complex_upsample_fwd(const float* x, float* y, int B, int H, int W, int C, int up_f, int up_t,
                                        dcs_stream_t stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || up_f < 1 || up_t < 1) return DCS_ERR_BADARG;
    const long n = (long)B * H * up_f * W * up_t * C;
    hipLaunchKernelGGL(upsample_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, dcs_stream(stream), (const float2*)x,
                       (float2*)y, B, H, W, C, up_f, up_t);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
