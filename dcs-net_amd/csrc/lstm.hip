// lstm.hip — the recurrent half of ComplexLSTM (c_network.py:12-51) as ONE persistent launch per
// layer.
//
// The reference runs four cuDNN/MIOpen LSTM passes (two weight sets x {re, im} inputs); on
// MI355X that is 2 small launches per time step per layer per direction per pass — 16,000
// launches per forward at T=2000 (seq 500), 77 % of the step (profiles/r01_a_*).  The work is
// latency-bound, not FLOP-bound (2.8 % of forward FLOPs), so:
//   * the input projection x_t W_ih^T + b_ih + b_hh of ALL time steps is one plain GEMM done by the
//     caller (rocBLAS through PyTorch) -> `gx`;
//   * this kernel walks the sequence with one 256-thread workgroup per (sequence, direction):
//     thread j owns gate column j (PyTorch order i,f,g,o, H = 64 each) with its W_hh row in 64
//     VGPRs, h_{t-1} lives in LDS (broadcast reads), c in a register; two barriers per step;
//     next step's gx is prefetched under the FMAs.
// All four passes of a layer (both weight sets, re and im inputs, both directions) run in the
// same launch: 8*B independent workgroups.
#include "dcs_common.h"

namespace {

constexpr int H = 64;
constexpr int G4 = 4 * H;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

template <bool SAVE>
__global__ __launch_bounds__(G4) void lstm_rec_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                           float* __restrict__ out, float* __restrict__ gates_save,
                                                           float* __restrict__ c_save, float* __restrict__ hprev_save,
                                                           int S, int seqs_per_set, long stride_set, long stride_n,
                                                           long stride_t) {
    __shared__ __attribute__((aligned(16))) float h_s[H];
    __shared__ float g_s[G4];
    const int j = threadIdx.x;
    const int n = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int set = n / seqs_per_set, ns = n % seqs_per_set;

    float w[H];
    {
        const float4* wr = reinterpret_cast<const float4*>(whh + ((long)(set * 2 + dir) * G4 + j) * H);
#pragma unroll
        for (int k = 0; k < H / 4; ++k) {
            const float4 v = wr[k];
            w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        }
    }
    const float* gxp = gx + set * stride_set + ns * stride_n + dir * G4 + j;
    const int t0 = dir ? S - 1 : 0, dt = dir ? -1 : 1;
    float c = 0.f;
    if (j < H) h_s[j] = 0.f;
    float pre = gxp[(long)t0 * stride_t];
    __syncthreads();
    const bool is_g = (j >> 6) == 2;
    for (int s = 0; s < S; ++s) {
        const int t = t0 + s * dt;
        float nxt = 0.f;
        if (s + 1 < S) nxt = gxp[(long)(t + dt) * stride_t];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float4* h4 = reinterpret_cast<const float4*>(h_s);
#pragma unroll
        for (int k = 0; k < H / 4; ++k) {
            const float4 hv = h4[k];
            a0 = fmaf(w[4 * k], hv.x, a0);
            a1 = fmaf(w[4 * k + 1], hv.y, a1);
            a2 = fmaf(w[4 * k + 2], hv.z, a2);
            a3 = fmaf(w[4 * k + 3], hv.w, a3);
        }
        const float a = pre + ((a0 + a1) + (a2 + a3));
        const float act = is_g ? tanhf(a) : sigmoidf_(a);
        g_s[j] = act;
        if (SAVE) gates_save[(((long)n * S + t) * 2 + dir) * G4 + j] = act;
        __syncthreads();
        if (j < H) {
            const float ig = g_s[j], fg = g_s[H + j], gg = g_s[2 * H + j], og = g_s[3 * H + j];
            c = fmaf(fg, c, ig * gg);
            const float h = og * tanhf(c);
            if (SAVE && hprev_save) hprev_save[(((long)n * S + t) * 2 + dir) * H + j] = h_s[j];   // state BEFORE this step
            h_s[j] = h;
            out[((long)n * S + t) * (2 * H) + dir * H + j] = h;
            if (SAVE) c_save[(((long)n * S + t) * 2 + dir) * H + j] = c;
        }
        __syncthreads();
        pre = nxt;
    }
}

// Backward through time for one (sequence, direction) per workgroup.  Threads u < H turn the
// cotangent of h_t (from the layer above + from step t+1) into the four pre-activation
// cotangents; then all 256 threads (k = t/4, quarter = t%4) form g_h_{t-1} = W_hh^T g_pre with the
// W_hh column quarter held in 64 VGPRs and a 4-lane shuffle reduction.  g_pre is written out for
// the caller's weight / input GEMMs.
__global__ __launch_bounds__(G4) void lstm_rec_bwd_kernel(const float* __restrict__ g_out,
                                                           const float* __restrict__ gates, const float* __restrict__ cs,
                                                           const float* __restrict__ whh, float* __restrict__ g_pre,
                                                           float* __restrict__ g_bias_part, int S, int seqs_per_set) {
    __shared__ float gp_s[G4];
    __shared__ float gh_s[H];
    const int j = threadIdx.x;
    const int n = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int set = n / seqs_per_set;
    const int k = j >> 2, part = j & 3;

    float w[H];      // W_hh[part*64 + jj][k]
    {
        const float* wb = whh + ((long)(set * 2 + dir) * G4 + part * H) * H + k;
#pragma unroll
        for (int jj = 0; jj < H; ++jj) w[jj] = wb[(long)jj * H];
    }
    const int t0 = dir ? S - 1 : 0, dt = dir ? -1 : 1;
    float gc_rec = 0.f;
    if (j < H) gh_s[j] = 0.f;
    // operands of the step about to run; the next step's are fetched while this one computes (none of them depends on
    // the recurrence, and left in the loop body their ~1 us of dependent global-load latency is paid S times)
    float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, cp = 0.f, go = 0.f;
    float sb0 = 0.f, sb1 = 0.f, sb2 = 0.f, sb3 = 0.f;                  // bias gradient of this (sequence, direction)
    auto fetch = [&](int s_, float& i_, float& f_, float& g_, float& o_, float& cp_, float& go_) {
        const int t_ = t0 + s_ * dt;
        const long gb_ = (((long)n * S + t_) * 2 + dir) * G4;
        i_ = gates[gb_ + j]; f_ = gates[gb_ + H + j]; g_ = gates[gb_ + 2 * H + j]; o_ = gates[gb_ + 3 * H + j];
        cp_ = s_ > 0 ? cs[(((long)n * S + (t_ - dt)) * 2 + dir) * H + j] : 0.f;
        go_ = g_out[((long)n * S + t_) * (2 * H) + dir * H + j];
    };
    if (j < H) {
        fetch(S - 1, ig, fg, gg, og, cp, go);
        c = cs[(((long)n * S + (t0 + (S - 1) * dt)) * 2 + dir) * H + j];
    }
    __syncthreads();
    for (int s = S - 1; s >= 0; --s) {
        const int t = t0 + s * dt;
        float n_ig = 0.f, n_fg = 0.f, n_gg = 0.f, n_og = 0.f, n_cp = 0.f, n_go = 0.f;
        if (j < H) {
            if (s > 0) fetch(s - 1, n_ig, n_fg, n_gg, n_og, n_cp, n_go);
            const long gb = (((long)n * S + t) * 2 + dir) * G4;
            const float tc = tanhf(c);
            const float gh = go + gh_s[j];
            const float gc = gh * og * (1.f - tc * tc) + gc_rec;
            const float pi = gc * gg * ig * (1.f - ig);
            const float pf = gc * cp * fg * (1.f - fg);
            const float pg = gc * ig * (1.f - gg * gg);
            const float po = gh * tc * og * (1.f - og);
            gc_rec = gc * fg;
            sb0 += pi; sb1 += pf; sb2 += pg; sb3 += po;
            gp_s[j] = pi; gp_s[H + j] = pf; gp_s[2 * H + j] = pg; gp_s[3 * H + j] = po;
            g_pre[gb + j] = pi; g_pre[gb + H + j] = pf; g_pre[gb + 2 * H + j] = pg; g_pre[gb + 3 * H + j] = po;
            c = cp;                                                    // c_{t-1} of this step is c_t of the next one
            ig = n_ig; fg = n_fg; gg = n_gg; og = n_og; cp = n_cp; go = n_go;
        }
        __syncthreads();
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float4* g4 = reinterpret_cast<const float4*>(gp_s + part * H);
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            const float4 gv = g4[q];
            a0 = fmaf(w[4 * q], gv.x, a0);
            a1 = fmaf(w[4 * q + 1], gv.y, a1);
            a2 = fmaf(w[4 * q + 2], gv.z, a2);
            a3 = fmaf(w[4 * q + 3], gv.w, a3);
        }
        float a = (a0 + a1) + (a2 + a3);
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        // gh_s of this step was consumed before the barrier above, so it may be overwritten now
        if (part == 0) gh_s[k] = a;
        __syncthreads();
    }
    if (g_bias_part && j < H) {
        float* bp = g_bias_part + (long)blockIdx.x * G4;               // [n][dir][4H]
        bp[j] = sb0; bp[H + j] = sb1; bp[2 * H + j] = sb2; bp[3 * H + j] = sb3;
    }
}

}  // namespace

extern "C" int dcs_lstm_layer_bwd(const float* g_out, const float* gates, const float* c_save, const float* w_hh,
                                  float* g_pre, float* g_bias_part, int n_sets, int seqs_per_set, int S, int Hdim,
                                  dcs_stream_t stream) {
    if (!g_out || !gates || !c_save || !w_hh || !g_pre || n_sets <= 0 || seqs_per_set <= 0 || S <= 0 || Hdim != H)
        return DCS_ERR_BADARG;
    hipLaunchKernelGGL(lstm_rec_bwd_kernel, dim3(n_sets * seqs_per_set * 2), dim3(G4), 0, dcs_stream(stream), g_out,
                       gates, c_save, w_hh, g_pre, g_bias_part, S, seqs_per_set);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_lstm_layer_fwd(const float* gx, const float* w_hh, float* out, float* gates_save, float* c_save,
                                  float* hprev_save, int n_sets, int seqs_per_set, int S, int Hdim, long stride_set, long stride_n,
                                  long stride_t, dcs_stream_t stream) {
    if (!gx || !w_hh || !out || n_sets <= 0 || seqs_per_set <= 0 || S <= 0 || Hdim != H) return DCS_ERR_BADARG;
    if ((gates_save == nullptr) != (c_save == nullptr)) return DCS_ERR_BADARG;
    const int NS = n_sets * seqs_per_set;
    dim3 grid(NS * 2);
    if (gates_save)
        hipLaunchKernelGGL(lstm_rec_fwd_kernel<true>, grid, dim3(G4), 0, dcs_stream(stream), gx, w_hh, out, gates_save,
                           c_save, hprev_save, S, seqs_per_set, stride_set, stride_n, stride_t);
    else
        hipLaunchKernelGGL(lstm_rec_fwd_kernel<false>, grid, dim3(G4), 0, dcs_stream(stream), gx, w_hh, out, gates_save,
                           c_save, hprev_save, S, seqs_per_set, stride_set, stride_n, stride_t);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
