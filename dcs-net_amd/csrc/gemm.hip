// gemm.hip — the input projections of ComplexLSTM (c_network.py:33-47: nn.LSTM's x_t W_ih^T for all time steps at once) and
// their data gradients, on the fp32 MFMA pipe (exact fp32 products: v_mfma_f32_32x32x2_f32).
//
// The reference runs these inside nn.LSTM (cuDNN / MIOpen); rounds 1-3 ran them as rocBLAS GEMMs through torch.mm / bmm —
// the last library kernels of the captured step.  They are small (0.5 GFLOP at the train shape, [2048 x 128] x [128 x 1024])
// and the library's 128 x 128 tiles left them at 15 us apiece on a quarter of the CUs.
//
//   C_b[m][n] = sum over segments s < nseg, k < K of  A_{b,s}[m][k] * op(B_{b,s})[k][n]
//   A_{b,s} = A + b * a_batch + s * a_seg  (row pitch lda, K contiguous);  B likewise: BT: B[n][k] (nn.LSTM's weight_ih
//   layout, pitch ldb), else B[k][n];  C_b = C + b * c_batch, row pitch ldc.
//
// A wave owns a 32-row x (32 NT)-column tile of C.  Two workgroup forms, chosen by the host so that the launch has about one
// workgroup per CU or more: SPLIT — the four waves share one tile and split K (a quarter of every segment each), summed through
// LDS in a fixed order (the data gradients: N = 128, K = 1024); spatial — the waves are a 2 x 2 arrangement of tiles and each walks
// all of K (the projections: N = 1024, K = 128).  The form is a function of the shape alone: a given shape always sums in the same
// order (bit-reproducible).  Fragments are read straight from global memory in the order the MFMA wants them: a lane holds row
// (lane % 32) and, of every block of eight k, the four at 4 * (lane / 32): one 16-byte load for A (and for B when it is
// K-contiguous) feeds four MFMAs; the contraction order inside a block is (kk, j) -> k = 8 blk + 4 kk + j for both operands, which
// is all the MFMA's sum needs.  Two operand register sets: the next group's loads fly during this group's MFMAs.
//
// Sized for the train shapes (2048-4096 rows: ~10 us per launch hot, rocBLAS 9-15 us; profiles/r04_lstm_gemm.txt).  At the
// inference shape (16128 rows) a workgroup's life is a third prologue and epilogue and the matrix pipe is busy 0.5 of the time
// (67 TFLOP/s against the library's 96 with its 256 x 256 tiles); a weight-stationary variant for K = 128 (B in 128 registers,
// the workgroup walking row tiles) measured no better — its stores share the in-order vmcnt queue with the next tile's loads.
// The host side therefore sends launches above DCS_LSTM_GEMM_MAX_GFLOP (default 1.5) to the library (functional._project).
#include "dcs_common.h"
#include <cstdlib>

namespace {
typedef float f32x16g __attribute__((ext_vector_type(16)));

struct GemmP {
    const float* A; const float* B; float* C;
    long a_seg, b_seg, a_batch, b_batch, c_batch;
    int M, N, K, lda, ldb, ldc, nseg;
    int c_planes;      // > 0: row m of C is the (m / c_planes) part — 0 real, 1 imaginary — of row m % c_planes of a complex-interleaved
                       // [c_planes][N] output: C[((m % c_planes) * N + n) * 2 + m / c_planes]  (ldc unused)
    int a_planes;      // > 0: A is read the same way from a complex-interleaved [a_planes][K] input (lda unused)
};

// One group of U k-blocks of this wave: t = block index over (segment, block of eight k).  No clamps, no run-time forms: every
// load of a group is issued back to back (the first version clamped ragged groups and chose the A form at run time — both became
// branches with a full wait each, and the kernel ran at four serial round trips per group).
template <bool BT, bool AP, int U, int NT>
__device__ __forceinline__ void load_group(const GemmP& p, const float* ga, const float* gb, int t0, int nblk, int part,
                                           float (&av)[U][4], float (&bv)[NT][U][4]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int t = t0 + u, s = t / nblk, blk = t - s * nblk;
        const float* a = ga + s * p.a_seg + (long)blk * (AP ? 16 : 8);
        if (AP) {                                                       // four complex values, this row's part (a is already offset by the part)
            const float* al = a - (part);
            const float4 x = *reinterpret_cast<const float4*>(al), y = *reinterpret_cast<const float4*>(al + 4);
            av[u][0] = part ? x.y : x.x; av[u][1] = part ? x.w : x.z; av[u][2] = part ? y.y : y.x; av[u][3] = part ? y.w : y.z;
        } else {
            const float4 a4 = *reinterpret_cast<const float4*>(a);
            av[u][0] = a4.x; av[u][1] = a4.y; av[u][2] = a4.z; av[u][3] = a4.w;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (BT) {
                const float4 x = *reinterpret_cast<const float4*>(gb + s * p.b_seg + blk * 8 + (long)(32 * nt) * p.ldb);
                bv[nt][u][0] = x.x; bv[nt][u][1] = x.y; bv[nt][u][2] = x.z; bv[nt][u][3] = x.w;
            } else {
                const float* q = gb + s * p.b_seg + (long)(blk * 8) * p.ldb + 32 * nt;
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[nt][u][j] = q[(long)j * p.ldb];
            }
        }
    }
}
template <int U, int NT>
__device__ __forceinline__ void mma_group(f32x16g (&acc)[NT], const float (&av)[U][4], const float (&bv)[NT][U][4]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][j], bv[nt][u][j], acc[nt], 0, 0, 0);
}

// U: k-blocks per group; the host picks the largest of 4 / 2 / 1 that divides the wave's block count.
// ONE: the wave's whole k range is a single group (the projections in the SPLIT form: K / 4 = 32) — no second operand set, a
// third of the registers, every load of the wave in one round trip.
template <bool BT, bool AP, int U, bool SPLIT, int NT, bool ONE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p) {
    DCS_PRIO_CRITICAL();
    __shared__ float red[SPLIT ? 4 : 1][SPLIT ? 16 * NT : 1][64];     // the waves' partial tiles: NT tiles x 16 registers x 64 lanes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kk = lane >> 5, li = lane & 31;
    const int m0 = SPLIT ? blockIdx.x * 32 : blockIdx.x * 64 + (wave & 1) * 32;
    const int n0 = SPLIT ? blockIdx.y * 32 * NT : (blockIdx.y * 2 + (wave >> 1)) * 32 * NT;
    const int b = blockIdx.z;
    const int KW = SPLIT ? p.K / 4 : p.K, nblk = KW / 8;               // this wave's k range of every segment, in blocks of eight
    const int mr = m0 + li < p.M ? m0 + li : p.M - 1;                  // clamped row (stores are masked)
    const int k0 = (SPLIT ? wave * KW : 0) + 4 * kk;
    const float* ga = p.A + b * p.a_batch + (AP ? ((long)(mr % p.a_planes) * p.K + k0) * 2 + mr / p.a_planes : (long)mr * p.lda + k0);
    const float* gb = p.B + b * p.b_batch + (BT ? (long)(n0 + li) * p.ldb + k0 : (long)k0 * p.ldb + n0 + li);
    f32x16g acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const int T = p.nseg * nblk;                                       // a multiple of U
    const int part = AP ? mr / p.a_planes : 0;
    if (ONE) {
        float a0[U][4], b0[NT][U][4];
        load_group<BT, AP, U, NT>(p, ga, gb, 0, nblk, part, a0, b0);
        mma_group<U, NT>(acc, a0, b0);
    } else {
        float a0[U][4], b0[NT][U][4], a1[U][4], b1[NT][U][4];
        load_group<BT, AP, U, NT>(p, ga, gb, 0, nblk, part, a0, b0);
        int t = U;
        for (; t + U < T; t += 2 * U) {
            load_group<BT, AP, U, NT>(p, ga, gb, t, nblk, part, a1, b1);
            mma_group<U, NT>(acc, a0, b0);
            load_group<BT, AP, U, NT>(p, ga, gb, t + U, nblk, part, a0, b0);
            mma_group<U, NT>(acc, a1, b1);
        }
        if (t < T) {
            load_group<BT, AP, U, NT>(p, ga, gb, t, nblk, part, a1, b1);
            mma_group<U, NT>(acc, a0, b0);
            mma_group<U, NT>(acc, a1, b1);
        } else {
            mma_group<U, NT>(acc, a0, b0);
        }
    }
    float* cb = p.C + b * p.c_batch;
    const int cs = p.c_planes > 0 ? 2 : 1;
    // C/D map: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (SPLIT) {
        // every wave leaves its partial tile in LDS; wave w then sums registers 4w .. 4w+3 of all four (waves 0, 1, 2, 3 in that
        // order) = rows 8w .. 8w+7 of the tile, and stores them: the combine and the stores run on all four waves
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave][nt * 16 + r][lane] = acc[nt][r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * wave + q, m = m0 + q + 8 * wave + 4 * kk;
            float v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                v[nt] = ((red[0][nt * 16 + r][lane] + red[1][nt * 16 + r][lane]) + red[2][nt * 16 + r][lane]) + red[3][nt * 16 + r][lane];
            if (m >= p.M) continue;
            float* o = p.c_planes > 0 ? cb + ((long)(m % p.c_planes) * p.N + n0 + li) * 2 + m / p.c_planes : cb + (long)m * p.ldc + n0 + li;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) o[32 * nt * cs] = v[nt];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (m >= p.M) continue;
        float* o = p.c_planes > 0 ? cb + ((long)(m % p.c_planes) * p.N + n0 + li) * 2 + m / p.c_planes : cb + (long)m * p.ldc + n0 + li;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[32 * nt * cs] = acc[nt][r];
    }
}

}  // namespace

extern "C" int dcs_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                            int b_transposed, int nseg, long a_seg, long b_seg, int nbatch, long a_batch, long b_batch,
                            long c_batch, int a_planes, int c_planes, dcs_stream_t stream) {
    if (!A || !B || !C || M < 1 || N < 64 || (N & 63) || K < 32 || (K & 31) || nseg < 1 || nbatch < 1 || nbatch > 65535)
        return DCS_ERR_BADARG;
    if (a_planes < 0 || c_planes < 0 || (a_planes > 0 && M > 2 * a_planes) || (c_planes > 0 && M > 2 * c_planes)) return DCS_ERR_BADARG;
    if ((a_planes == 0 && (lda < K || (lda & 3))) || (c_planes == 0 && ldc < N) || (b_transposed ? (ldb < K || (ldb & 3)) : ldb < N))
        return DCS_ERR_BADARG;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || (a_seg & 3) || (b_seg & 3) || (a_batch & 3) || (b_batch & 3)) return DCS_ERR_BADARG;
    if (N / 64 > 65535) return DCS_ERR_BADARG;
    GemmP p{A, B, C, a_seg, b_seg, a_batch, b_batch, c_batch, M, N, K, lda, ldb, ldc, nseg, c_planes, a_planes};
    // the form with the largest tiles that still gives about a workgroup per CU (a function of the shape alone); a wave whose
    // whole k range is one group of four blocks (K = 128 split four ways: the projections) takes the SPLIT form whatever the
    // tile count: all its loads are one round trip, and in the step these launches start cold
    const long wg_spatial = (N % 128 == 0) ? (long)((M + 63) / 64) * (N / 128) * nbatch : 0;
    const long wg_split2 = (long)((M + 31) / 32) * (N / 64) * nbatch;
    static const int force = (int)dcs_knob("DCS_GEMM_FORM", -1);     // diagnostic: 0 spatial, 1 / 2 split
    int form = (nseg * (K / 32) == 4 && wg_split2 >= 200) ? 2 : wg_spatial >= 200 ? 0 : (wg_split2 >= 200 ? 2 : 1);
    if (force >= 0 && force <= 2 && !(force == 0 && N % 128)) form = force;
    const dim3 grid(form == 0 ? (M + 63) / 64 : (M + 31) / 32, form == 0 ? N / 128 : (form == 2 ? N / 64 : N / 32), nbatch);
    const int T = nseg * (form == 0 ? K / 8 : K / 32);
    const hipStream_t st = dcs_stream(stream);
#define DCS_GEMM_L(BT_, AP_, U_, SP_, NT_, ONE_) DCS_LAUNCH((gemm_f32_kernel<BT_, AP_, U_, SP_, NT_, ONE_>), grid, dim3(256), 0, st, p)
#define DCS_GEMM_S(BT_, AP_, NT_)                                                                              \
    do {                                                                                                       \
        if (T == 4) DCS_GEMM_L(BT_, AP_, 4, true, NT_, true);                                                  \
        else if (T % 4 == 0) DCS_GEMM_L(BT_, AP_, 4, true, NT_, false);                                        \
        else if (T % 2 == 0) DCS_GEMM_L(BT_, AP_, 2, true, NT_, false);                                        \
        else DCS_GEMM_L(BT_, AP_, 1, true, NT_, false);                                                        \
    } while (0)
#define DCS_GEMM_F(BT_, AP_)                                                                                   \
    do {                                                                                                       \
        if (form == 0) DCS_GEMM_L(BT_, AP_, 4, false, 2, false);                                               \
        else if (form == 2) DCS_GEMM_S(BT_, AP_, 2);                                                           \
        else DCS_GEMM_S(BT_, AP_, 1);                                                                          \
    } while (0)
    if (b_transposed) { if (a_planes > 0) DCS_GEMM_F(true, true); else DCS_GEMM_F(true, false); }
    else { if (a_planes > 0) DCS_GEMM_F(false, true); else DCS_GEMM_F(false, false); }
#undef DCS_GEMM_S
#undef DCS_GEMM_F
#undef DCS_GEMM_L
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}
