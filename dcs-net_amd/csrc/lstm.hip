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
//     each thread owns one gate row (PyTorch order i,f,g,o, H = 64 each) with its W_hh row in 64
//     VGPRs; the four gates of a hidden unit share a quad (DPP exchange), h_{t-1} lives in a
//     double-buffered LDS vector (broadcast reads), c in a register; ONE barrier per step;
//     next step's gx is prefetched under the FMAs.
// All four passes of a layer (both weight sets, re and im inputs, both directions) run in the
// same launch: 8*B independent workgroups.
#include "dcs_common.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

// tanh through the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each): 1 - 2 / (2^(2x log2 e) + 1), absolute
// error ~1e-7 (saturates correctly at +-1).  The library tanhf / expf pair costs ~100 VALU instructions per step on the
// recurrence's critical path — more than the 64-FMA dot product; the sigmoid gates reuse the same evaluation,
// sigmoid(x) = 0.5 tanh(x / 2) + 0.5, so the four gates of a quad do not diverge.
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.885390081777927f) + 1.f);
}

// quad_perm DPP: every lane of a quad reads lane K of its quad (broadcast) or its xor-partner
// The per-step barrier of the recurrences: LDS traffic drained, then s_barrier.  __syncthreads() would also wait for
// every outstanding GLOBAL access (vmcnt(0)): the prefetched operands of the coming steps and the step's own stores —
// an L2 / HBM round trip on the critical path of every time step.  Only the h / g_pre vector in LDS crosses the barrier.
__device__ __forceinline__ void step_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) { return dcs_dpp_term<CTRL, 0xf>(v); }
__device__ __forceinline__ float quad_bcast0(float v) { return quad_dpp<0x00>(v); }
__device__ __forceinline__ float quad_bcast1(float v) { return quad_dpp<0x55>(v); }
__device__ __forceinline__ float quad_bcast2(float v) { return quad_dpp<0xAA>(v); }
__device__ __forceinline__ float quad_bcast3(float v) { return quad_dpp<0xFF>(v); }
__device__ __forceinline__ float quad_sum(float v) {
    v += quad_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    v += quad_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    return v;
}

// Thread t = 4*u + gate: the four gates (i, f, g, o) of hidden unit u sit in one quad, so the cell update needs no LDS
// round trip — the activations are exchanged with quad_perm DPP and all four lanes update (c, h) redundantly.  h lives
// in a double-buffered LDS vector (step s reads buffer s&1 while lane gate 0 writes the other): ONE barrier per step.
template <int H, bool SAVE>
__global__ __launch_bounds__(4 * H) void lstm_rec_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                           float* __restrict__ out, float* __restrict__ gates_save,
                                                           float* __restrict__ c_save, float* __restrict__ hprev_save,
                                                           int S, int seqs_per_set, long stride_set, long stride_n,
                                                           long stride_t, const float* __restrict__ bias_a,
                                                           const float* __restrict__ bias_b) {
    constexpr int G4 = 4 * H;
    __shared__ __attribute__((aligned(16))) float h_s[2][H];
    const int t = threadIdx.x, u = t >> 2, gate = t & 3, j = gate * H + u;      // j: PyTorch gate row (i, f, g, o blocks)
    const int n = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int set = n / seqs_per_set, ns = n % seqs_per_set;

    // The quad of hidden unit u splits the dot products over h: lane `gate` holds, for ALL four gate rows of the unit, the
    // H/4 weights that multiply h[gate*H/4 .. (gate+1)*H/4) — the same 64 FMAs per lane as one whole row, but as four
    // independent chains over H/16 LDS reads instead of one chain over H/4 reads (the step was bound by that chain's
    // read -> FMA latency, ~1100 cycles), and the four partial sums of a row meet in a quad DPP sum.
    constexpr int Q = H / 4;
    float w[4][Q];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4* wr = reinterpret_cast<const float4*>(whh + ((long)(set * 2 + dir) * G4 + g * H + u) * H + gate * Q);
#pragma unroll
        for (int k = 0; k < Q / 4; ++k) {
            const float4 v = wr[k];
            w[g][4 * k] = v.x; w[g][4 * k + 1] = v.y; w[g][4 * k + 2] = v.z; w[g][4 * k + 3] = v.w;
        }
    }
    const float* gxp = gx + set * stride_set + ns * stride_n + dir * G4 + j;
    // optional gate biases [n_sets][2 dirs][4H] (b_ih, b_hh) added here instead of by a bias-broadcast pass over gx
    float bias = 0.f;
    if (bias_a) bias = bias_a[(set * 2 + dir) * G4 + j];
    if (bias_b) bias += bias_b[(set * 2 + dir) * G4 + j];
    const int t0 = dir ? S - 1 : 0, dt = dir ? -1 : 1;
    float c = 0.f, hprev = 0.f;
    if (t < H) h_s[0][t] = 0.f;
    float pre = gxp[(long)t0 * stride_t] + bias;                      // (consumes the bias loads before the loop: the loop's
                                                                       //  vmcnt counts then only ever see the gx prefetches)
    float pre2 = S > 1 ? gxp[(long)(t0 + dt) * stride_t] : 0.f;        // two steps of input projections in flight
    __syncthreads();
    const bool is_g = gate == 2;
    for (int s = 0; s < S; ++s) {
        const int tt = t0 + s * dt, cur = s & 1;
        const float nxt = pre2;
        // always a load (the last two steps re-read the final time index): a branch around it makes the compiler wait with
        // vmcnt(0) — i.e. for this very load — instead of counting past it
        pre2 = gxp[(long)(s + 2 < S ? tt + 2 * dt : t0 + (S - 1) * dt) * stride_t];
        // packed FMAs (v_pk_fma_f32: half the VALU issue slots of the scalar form), one accumulator pair per gate row
        v2f acc[4] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}};
        const float4* h4 = reinterpret_cast<const float4*>(h_s[cur] + gate * Q);
#pragma unroll
        for (int k = 0; k < Q / 4; ++k) {
            const float4 hv = h4[k];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g] = __builtin_elementwise_fma(v2f{w[g][4 * k], w[g][4 * k + 1]}, v2f{hv.x, hv.y}, acc[g]);
                acc[g] = __builtin_elementwise_fma(v2f{w[g][4 * k + 2], w[g][4 * k + 3]}, v2f{hv.z, hv.w}, acc[g]);
            }
        }
        const float s0 = quad_sum(acc[0].x + acc[0].y), s1 = quad_sum(acc[1].x + acc[1].y);
        const float s2 = quad_sum(acc[2].x + acc[2].y), s3 = quad_sum(acc[3].x + acc[3].y);
        const float a = pre + (gate == 0 ? s0 : (gate == 1 ? s1 : (gate == 2 ? s2 : s3)));
        const float th = fast_tanh(is_g ? a : 0.5f * a);
        const float act = is_g ? th : fmaf(0.5f, th, 0.5f);
        if (SAVE) gates_save[(((long)n * S + tt) * 2 + dir) * G4 + j] = act;
        const float ig = quad_bcast0(act), fg = quad_bcast1(act), gg = quad_bcast2(act), og = quad_bcast3(act);
        c = fmaf(fg, c, ig * gg);
        const float h = og * fast_tanh(c);
        // All four lanes of a quad hold the same (c, h): they all store them (same value, same address — one write after
        // coalescing).  Under `if (gate == 0)` the global stores sat in a divergent branch, the compiler could not count
        // them, and the wait for the next step's input projection became vmcnt(0): every step then also waited for the
        // previous step's stores (the no-save form ran 270 us at S = 500 against 243 us for the form that stores more).
        if (gate == 0) h_s[cur ^ 1][u] = h;
        out[((long)n * S + tt) * (2 * H) + dir * H + u] = h;
        if (SAVE) {
            c_save[(((long)n * S + tt) * 2 + dir) * H + u] = c;
            if (hprev_save) hprev_save[(((long)n * S + tt) * 2 + dir) * H + u] = hprev;   // state BEFORE this step (uniform test)
        }
        hprev = h;
        step_barrier();
        pre = nxt + bias;
    }
}

// Backward through time for one (sequence, direction) per workgroup, same quad mapping (t = 4*u + gate): every lane of
// a quad forms the four pre-activation cotangents of its unit from the saved gates (its own gate loaded, the others by
// DPP broadcast) and keeps its own; g_h_{t-1} = W_hh^T g_pre is a 64-long dot product per lane over its gate's block
// (W_hh column of unit u, rows gate*H..) plus a quad sum — so the recurrent cotangent of unit u never leaves the quad.
// g_pre of the step goes through a double-buffered LDS vector: ONE barrier per step.  The next step's operands are
// fetched while the current one computes (none depends on the recurrence).
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_rec_bwd_kernel(const float* __restrict__ g_out,
                                                           const float* __restrict__ gates, const float* __restrict__ cs,
                                                           const float* __restrict__ whh, float* __restrict__ g_pre,
                                                           float* __restrict__ g_bias_part, int S, int seqs_per_set) {
    DCS_PRIO_CRITICAL();
    constexpr int G4 = 4 * H;
    // The four gate blocks of the step's g_pre vector sit H + 4 floats apart (Round 4).  The lanes of a wave carry all four
    // gates (gate = t & 3), so every 16-byte read of the dot product below has four distinct addresses, one per gate block: at
    // a pitch of H floats (256 / 512 bytes) they fell on the SAME banks — a four-way conflict on all 16 (32) reads of a step,
    // SQ_LDS_BANK_CONFLICT at 0.75 of the LDS-active cycles (profiles/r04_a_pmc_wait_states.txt); at H + 4 they are 4 banks apart.
    constexpr int GS = H + 4;
    __shared__ __attribute__((aligned(16))) float gp_s[2][4 * GS];
    const int t = threadIdx.x, u = t >> 2, gate = t & 3, j = gate * H + u;
    const int n = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int set = n / seqs_per_set;

    float w[H];      // W_hh[gate*64 + jj][u]
    {
        const float* wb = whh + ((long)(set * 2 + dir) * G4 + gate * H) * H + u;
#pragma unroll
        for (int jj = 0; jj < H; ++jj) w[jj] = wb[(long)jj * H];
    }
    const int t0 = dir ? S - 1 : 0, dt = dir ? -1 : 1;
    float gc_rec = 0.f, gh_rec = 0.f, sb = 0.f;                         // sb: bias gradient of gate row j
    float act = 0.f, c = 0.f, cp = 0.f, go = 0.f;
    // every fetch is three UNCONDITIONAL loads (clamped indices, values selected afterwards): a branch around a load makes
    // the compiler wait with vmcnt(0) for the look-ahead loads it has just issued (lstm_rec_fwd_kernel: 270 -> 168 us)
    auto fetch = [&](int s_, float& a_, float& cp_, float& go_) {
        const int t_ = t0 + s_ * dt;
        a_ = gates[(((long)n * S + t_) * 2 + dir) * G4 + j];
        const float cv = cs[(((long)n * S + (s_ > 0 ? t_ - dt : t_)) * 2 + dir) * H + u];
        cp_ = s_ > 0 ? cv : 0.f;
        go_ = g_out[((long)n * S + t_) * (2 * H) + dir * H + u];
    };
    fetch(S - 1, act, cp, go);
    c = cs[(((long)n * S + (t0 + (S - 1) * dt)) * 2 + dir) * H + u];
    // operands of the next TWO steps in flight: one step of compute (~0.5 us) is shorter than an L2 / HBM round trip
    float q_act = 0.f, q_cp = 0.f, q_go = 0.f;                         // step s - 1
    fetch(S > 1 ? S - 2 : 0, q_act, q_cp, q_go);
    for (int s = S - 1; s >= 0; --s) {
        const int tt = t0 + s * dt, cur = s & 1;
        float n_act = q_act, n_cp = q_cp, n_go = q_go;
        fetch(s > 1 ? s - 2 : 0, q_act, q_cp, q_go);                   // (the last two steps re-read step 0: unused)
        const float ig = quad_bcast0(act), fg = quad_bcast1(act), gg = quad_bcast2(act), og = quad_bcast3(act);
        const float tc = fast_tanh(c);
        const float gh = go + gh_rec;
        const float gc = gh * og * (1.f - tc * tc) + gc_rec;
        const float pi = gc * gg * ig * (1.f - ig);
        const float pf = gc * cp * fg * (1.f - fg);
        const float pg = gc * ig * (1.f - gg * gg);
        const float po = gh * tc * og * (1.f - og);
        gc_rec = gc * fg;
        const float mine = gate == 0 ? pi : (gate == 1 ? pf : (gate == 2 ? pg : po));
        sb += mine;
        gp_s[cur][gate * GS + u] = mine;
        g_pre[(((long)n * S + tt) * 2 + dir) * G4 + j] = mine;
        c = cp;                                                        // c_{t-1} of this step is c_t of the next one
        act = n_act; cp = n_cp; go = n_go;
        step_barrier();
        v2f a01 = v2f{0.f, 0.f}, a23 = v2f{0.f, 0.f};
        const float4* g4 = reinterpret_cast<const float4*>(gp_s[cur] + gate * GS);
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            const float4 gv = g4[q];
            a01 = __builtin_elementwise_fma(v2f{w[4 * q], w[4 * q + 1]}, v2f{gv.x, gv.y}, a01);
            a23 = __builtin_elementwise_fma(v2f{w[4 * q + 2], w[4 * q + 3]}, v2f{gv.z, gv.w}, a23);
        }
        gh_rec = quad_sum((a01.x + a01.y) + (a23.x + a23.y));
    }
    if (g_bias_part) g_bias_part[(long)blockIdx.x * G4 + j] = sb;       // [n][dir][4H]
}

}  // namespace

extern "C" int dcs_lstm_layer_bwd(const float* g_out, const float* gates, const float* c_save, const float* w_hh,
                                  float* g_pre, float* g_bias_part, int n_sets, int seqs_per_set, int S, int Hdim,
                                  dcs_stream_t stream) {
    if (!g_out || !gates || !c_save || !w_hh || !g_pre || n_sets <= 0 || seqs_per_set <= 0 || S <= 0 ||
        (Hdim != 64 && Hdim != 128))
        return DCS_ERR_BADARG;
    if (Hdim == 128)                                         // DR-Net's LSTM (r_network.py:75-79)
        DCS_LAUNCH(lstm_rec_bwd_kernel<128>, dim3(n_sets * seqs_per_set * 2), dim3(512), 0, dcs_stream(stream), g_out,
                   gates, c_save, w_hh, g_pre, g_bias_part, S, seqs_per_set);
    else
        DCS_LAUNCH(lstm_rec_bwd_kernel<64>, dim3(n_sets * seqs_per_set * 2), dim3(256), 0, dcs_stream(stream), g_out,
                   gates, c_save, w_hh, g_pre, g_bias_part, S, seqs_per_set);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_lstm_layer_fwd_bias(const float* gx, const float* w_hh, const float* bias_a, const float* bias_b, float* out,
                                       float* gates_save, float* c_save, float* hprev_save, int n_sets, int seqs_per_set, int S,
                                       int Hdim, long stride_set, long stride_n, long stride_t, dcs_stream_t stream) {
    if (!gx || !w_hh || !out || n_sets <= 0 || seqs_per_set <= 0 || S <= 0 || (Hdim != 64 && Hdim != 128)) return DCS_ERR_BADARG;
    if ((gates_save == nullptr) != (c_save == nullptr)) return DCS_ERR_BADARG;
    if (bias_b && !bias_a) return DCS_ERR_BADARG;
    const int NS = n_sets * seqs_per_set;
    dim3 grid(NS * 2);
    if (Hdim == 128 && gates_save)
        DCS_LAUNCH((lstm_rec_fwd_kernel<128, true>), grid, dim3(512), 0, dcs_stream(stream), gx, w_hh, out, gates_save,
                           c_save, hprev_save, S, seqs_per_set, stride_set, stride_n, stride_t, bias_a, bias_b);
    else if (Hdim == 128)
        DCS_LAUNCH((lstm_rec_fwd_kernel<128, false>), grid, dim3(512), 0, dcs_stream(stream), gx, w_hh, out, gates_save,
                           c_save, hprev_save, S, seqs_per_set, stride_set, stride_n, stride_t, bias_a, bias_b);
    else if (gates_save)
        DCS_LAUNCH((lstm_rec_fwd_kernel<64, true>), grid, dim3(256), 0, dcs_stream(stream), gx, w_hh, out, gates_save,
                           c_save, hprev_save, S, seqs_per_set, stride_set, stride_n, stride_t, bias_a, bias_b);
    else
        DCS_LAUNCH((lstm_rec_fwd_kernel<64, false>), grid, dim3(256), 0, dcs_stream(stream), gx, w_hh, out, gates_save,
                           c_save, hprev_save, S, seqs_per_set, stride_set, stride_n, stride_t, bias_a, bias_b);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_lstm_layer_fwd(const float* gx, const float* w_hh, float* out, float* gates_save, float* c_save,
                                  float* hprev_save, int n_sets, int seqs_per_set, int S, int Hdim, long stride_set, long stride_n,
                                  long stride_t, dcs_stream_t stream) {
    return dcs_lstm_layer_fwd_bias(gx, w_hh, nullptr, nullptr, out, gates_save, c_save, hprev_save, n_sets, seqs_per_set, S, Hdim,
                                   stride_set, stride_n, stride_t, stream);
}


// ---- glue of the complex LSTM that autograd ran as ~10 ATen launches per layer and step -------------------------------
namespace {
// ComplexLSTM's recombination (c_network.py:43-46) from the stacked recurrence outputs o[set][{re rows | im rows}][S][W]:
// out[b] = (L_r(x_r) - L_i(x_i)) + j (L_r(x_i) + L_i(x_r)), interleaved complex.  n = B*S*W complex elements.
__global__ __launch_bounds__(256) void lstm_combine_fwd_kernel(const float* __restrict__ o, float2* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float rr = o[i], ir = o[n + i], ri = o[2 * n + i], ii = o[3 * n + i];
    out[i] = make_float2(rr - ii, ir + ri);
}
// its cotangent: g_o[0] = (g.re | g.im), g_o[1] = (g.im | -g.re)
__global__ __launch_bounds__(256) void lstm_combine_bwd_kernel(const float2* __restrict__ g, float* __restrict__ g_o, long n) {
    DCS_PRIO_CRITICAL();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 v = g[i];
    g_o[i] = v.x; g_o[n + i] = v.y; g_o[2 * n + i] = v.y; g_o[3 * n + i] = -v.x;
}
// parameter gradients of one layer from the partial products the backward left:
//   g_whh[set][dir][j][k] += sum_c part[dir][set*CK + c][j][k]      (the K-chunked bmm outputs, fixed order)
//   g_bih[set][q] += s, g_bhh[set][q] += s,  s = sum_n b_part[set][n][q]   (per-sequence bias sums of the BPTT kernel)
//   g_wih[set][i] += sum_c part_ih[set * CK_ih + c][i], i < MN_ih  (the chunked A^T B outputs of dcs_atb_chunks; optional: ih_blocks > 0)
__global__ __launch_bounds__(256) void lstm_param_grads_kernel(const float* __restrict__ part, const float* __restrict__ b_part,
                                                                float* __restrict__ g_whh, float* __restrict__ g_bih,
                                                                float* __restrict__ g_bhh, int CK, int seqs, int H, int w_blocks,
                                                                const float* __restrict__ part_ih, float* __restrict__ g_wih,
                                                                int CK_ih, long MN_ih, int ih_blocks, int ih_first) {
    if ((int)blockIdx.x >= ih_first) {                                  // W_ih chunk sums (chunk_sum_acc_kernel's work, same order)
        const int bx = blockIdx.x - ih_first, s_ = bx / ih_blocks;
        const long i = (long)(bx % ih_blocks) * 256 + threadIdx.x;
        if (i >= MN_ih) return;
        const float* q = part_ih + (long)s_ * CK_ih * MN_ih + i;
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < CK_ih; ++c) a += q[(long)c * MN_ih];
        g_wih[s_ * MN_ih + i] += a;
        return;
    }
    const int WH = 4 * H * H, nW = 4 * WH;                             // [2 sets][2 dirs][4H][H]
    if ((int)blockIdx.x < w_blocks) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= nW) return;
        const int e = i % WH, d = (i / WH) % 2, s_ = i / (2 * WH);
        const float* p = part + ((long)d * 2 * CK + (long)s_ * CK) * WH + e;
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < CK; ++c) a += p[(long)c * WH];            // (unrolled: the chunk loads go out together)
        g_whh[i] += a;
        return;
    }
    // bias blocks: 16 columns x 16 sequence chunks per block (a thread per column alone walked all the sequences in
    // one serial chain of loads: 17 us), combined through LDS in a fixed order
    __shared__ float red[16][17];
    const int q = (blockIdx.x - w_blocks) * 16 + (threadIdx.x & 15), ch = threadIdx.x >> 4;     // column of [2 sets][8H]
    const int s_ = q / (8 * H), qq = q % (8 * H);
    const int per = (seqs + 15) / 16;
    float a = 0.f;
    if (q < 16 * H) {
        const float* p = b_part + (long)s_ * seqs * 8 * H + qq;
        for (int n = ch * per; n < seqs && n < (ch + 1) * per; ++n) a += p[(long)n * 8 * H];
    }
    red[ch][threadIdx.x & 15] = a;
    __syncthreads();
    if (threadIdx.x < 16 && q < 16 * H) {
        float t = 0.f;
        for (int c = 0; c < 16; ++c) t += red[c][threadIdx.x];
        g_bih[q] += t;
        g_bhh[q] += t;
    }
}
}  // namespace

namespace {
typedef float f32x16w __attribute__((ext_vector_type(16)));
// Chunked A^T B on the MFMA pipe (fp32, exact) — the LSTM's parameter-gradient products, which rocBLAS ran as strided batched
// GEMMs with 32x32x128 tiles at ~5-20 TFLOP/s (25 us each):
//   part[(b * CK + c)][m][n] = sum over the R rows r of chunk c of  A_b[r][m] * B_b[r][n],   b = hi * nlo + lo,
//   A_b = A + lo * a_lo + hi * a_hi (row pitch lda), B_b likewise (ldb).
// A workgroup = one 32-row x 64-column output tile of one (batch, chunk); its four waves split the chunk's rows and are
// summed through LDS in a fixed order.  Both operands are read straight from global memory in fragment order (a lane's A
// element is A[r + lane/32][m0 + lane%32]: 128-byte runs), eight row pairs in flight per wave.
struct AtB {
    const float* A; const float* B; float* part;
    long a_lo, a_hi, b_lo, b_hi;
    int nlo, lda, ldb, M, N, R, CK;
    int b_es;                                                           // element stride of B along n (2: one part of a complex-interleaved row)
};
__global__ __launch_bounds__(256) void atb_chunks_kernel(AtB p) {
    __shared__ float red[3][32][64];                                    // waves 1..3 -> wave 0: 2 tiles x 16 registers x 64 lanes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kk = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 64;
    const int bc = blockIdx.z, bb = bc / p.CK, c = bc % p.CK, lo = bb % p.nlo, hi = bb / p.nlo;
    const int RW = p.R / 4;                                            // rows per wave (even)
    const long r0 = (long)c * p.R + (long)wave * RW + kk;
    const float* ga = p.A + lo * p.a_lo + hi * p.a_hi + r0 * p.lda + m0 + li;
    const float* hb = p.B + lo * p.b_lo + hi * p.b_hi + r0 * p.ldb + (n0 + li) * p.b_es;
    const int b32 = 32 * p.b_es;
    const long ga_step = 2L * p.lda, hb_step = 2L * p.ldb;
    f32x16w acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    constexpr int UN = 16;                                             // row pairs in flight (all of a wave's at the LSTM's shapes)
    for (int i0 = 0; i0 < RW / 2; i0 += UN) {
        float av[UN], b0[UN], b1[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {                                  // always loads (clamped row pair), A zeroed past the end
            const bool ok = i0 + u < RW / 2;
            const int ii = ok ? i0 + u : RW / 2 - 1;
            const float a_ = ga[ii * ga_step];
            av[u] = ok ? a_ : 0.f;
            b0[u] = hb[ii * hb_step];
            b1[u] = hb[ii * hb_step + b32];
        }
        __builtin_amdgcn_sched_barrier(0);                             // every load issued before the first MFMA waits
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b1[u], acc1, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { red[wave - 1][r][lane] = acc0[r]; red[wave - 1][16 + r][lane] = acc1[r]; }
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] += red[w][r][lane]; acc1[r] += red[w][16 + r][lane]; }
    // C/D map: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* o = p.part + ((long)bc * p.M + m0) * p.N + n0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
        o[(long)row * p.N] = acc0[r];
        o[(long)row * p.N + 32] = acc1[r];
    }
}
// out_b[i] += sum_c part[(b * CK + c)][i]  (i < MN), out_b = out + lo * o_lo + hi * o_hi; fixed order
__global__ __launch_bounds__(256) void chunk_sum_acc_kernel(const float* __restrict__ part, float* __restrict__ out, long o_lo,
                                                             long o_hi, int nlo, int CK, long MN) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    const int bb = blockIdx.y, lo = bb % nlo, hi = bb / nlo;
    const float* q = part + (long)bb * CK * MN + i;
    float a = 0.f;
#pragma unroll 8
    for (int c = 0; c < CK; ++c) a += q[(long)c * MN];
    out[lo * o_lo + hi * o_hi + i] += a;
}
}  // namespace

extern "C" int dcs_atb_chunks_strided(const float* A, const float* B, float* part, long a_lo, long a_hi, long b_lo, long b_hi,
                                      int nlo, int nhi, int lda, int ldb, int b_es, int M, int N, int R, int CK,
                                      dcs_stream_t stream) {
    if (!A || !B || !part || nlo < 1 || nhi < 1 || M < 32 || (M & 31) || N < 64 || (N & 63) || R < 8 || (R & 7) || CK < 1 || b_es < 1)
        return DCS_ERR_BADARG;
    if ((long)nlo * nhi * CK > 65535 || N / 64 > 65535) return DCS_ERR_BADARG;
    AtB p{A, B, part, a_lo, a_hi, b_lo, b_hi, nlo, lda, ldb, M, N, R, CK, b_es};
    DCS_LAUNCH(atb_chunks_kernel, dim3(M / 32, N / 64, nlo * nhi * CK), dim3(256), 0, dcs_stream(stream), p);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_atb_chunks(const float* A, const float* B, float* part, long a_lo, long a_hi, long b_lo, long b_hi, int nlo,
                              int nhi, int lda, int ldb, int M, int N, int R, int CK, dcs_stream_t stream) {
    return dcs_atb_chunks_strided(A, B, part, a_lo, a_hi, b_lo, b_hi, nlo, nhi, lda, ldb, 1, M, N, R, CK, stream);
}

extern "C" int dcs_chunk_sum_acc(const float* part, float* out, long o_lo, long o_hi, int nlo, int nhi, int CK, long MN,
                                 dcs_stream_t stream) {
    if (!part || !out || nlo < 1 || nhi < 1 || CK < 1 || MN < 1 || (long)nlo * nhi > 65535) return DCS_ERR_BADARG;
    DCS_LAUNCH(chunk_sum_acc_kernel, dim3((unsigned)((MN + 255) / 256), nlo * nhi), dim3(256), 0, dcs_stream(stream), part, out,
               o_lo, o_hi, nlo, CK, MN);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// the recurrent-weight products of one layer: batches (d, s) = (hi, lo), part[d][s*CK + c][4H][H]
extern "C" int dcs_lstm_whh_grad(const float* g_pre, const float* h_prev, float* part, int NT, int CK, int H,
                                 dcs_stream_t stream) {
    if (NT < 8 || CK < 1 || NT % CK != 0 || ((NT / CK) & 7) || H < 64 || (H & 63)) return DCS_ERR_BADARG;
    return dcs_atb_chunks(g_pre, h_prev, part, (long)NT * 8 * H, 4L * H, (long)NT * 2 * H, (long)H, 2, 2, 8 * H, 2 * H, 4 * H, H,
                          NT / CK, CK, stream);
}

extern "C" int dcs_lstm_combine_fwd(const float* o, float* out, long n, dcs_stream_t stream) {
    if (!o || !out || n <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH(lstm_combine_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, dcs_stream(stream), o, (float2*)out, n);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_lstm_combine_bwd(const float* g, float* g_o, long n, dcs_stream_t stream) {
    if (!g || !g_o || n <= 0) return DCS_ERR_BADARG;
    DCS_LAUNCH(lstm_combine_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, dcs_stream(stream), (const float2*)g, g_o, n);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_lstm_param_grads_ih(const float* part, const float* b_part, float* g_whh, float* g_bih, float* g_bhh, int CK,
                                       int seqs_per_set, int H, const float* part_ih, float* g_wih, int CK_ih, long MN_ih, int nsets,
                                       dcs_stream_t stream) {
    if (!part || !b_part || !g_whh || !g_bih || !g_bhh || CK < 1 || seqs_per_set < 1 || H < 1) return DCS_ERR_BADARG;
    if (part_ih && (!g_wih || CK_ih < 1 || MN_ih < 1 || nsets < 1 || nsets > 64)) return DCS_ERR_BADARG;
    const int w_blocks = (16 * H * H + 255) / 256, b_blocks = H;         // 16 H bias columns, 16 per block
    const int ih_blocks = part_ih ? (int)((MN_ih + 255) / 256) : 0;
    DCS_LAUNCH(lstm_param_grads_kernel, dim3(w_blocks + b_blocks + ih_blocks * (part_ih ? nsets : 0)), dim3(256), 0, dcs_stream(stream),
               part, b_part, g_whh, g_bih, g_bhh, CK, seqs_per_set, H, w_blocks, part_ih, g_wih, CK_ih, MN_ih, ih_blocks > 0 ? ih_blocks : 1,
               part_ih ? w_blocks + b_blocks : 0x7fffffff);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_lstm_param_grads(const float* part, const float* b_part, float* g_whh, float* g_bih, float* g_bhh, int CK,
                                    int seqs_per_set, int H, dcs_stream_t stream) {
    return dcs_lstm_param_grads_ih(part, b_part, g_whh, g_bih, g_bhh, CK, seqs_per_set, H, nullptr, nullptr, 0, 0, 0, stream);
}
