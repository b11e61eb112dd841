// conv_common.h — geometry and the virtual-input gather shared by the direct (conv_direct.hip) and
// MFMA implicit-GEMM (conv_mfma.hip) complex convolution kernels.
#pragma once
#include "dcs_common.h"

// The conv family is compiled twice (dcs_common.h: act_t): every cross-file function that touches activations gets its own
// symbol in the bf16 build.  NOT renamed — taken from the once-compiled fp32 objects, whose operands are fp32 in either
// mode: the 7x7 attention convs (conv_k7.hip, conv_wgrad_small.hip, dcs_conv_direct_multi), the weight packers
// (conv_pack.hip, pack_jobs.hip), the slab reduces (wgrad_reduce.hip).
#ifdef DCS_ACT_BF16
#define dcs_conv_mfma_stat_rows dcs_conv_mfma_stat_rows_h
#define dcs_conv_mfma_pack dcs_conv_mfma_pack_h
#define dcs_conv_mfma_workspace_bytes dcs_conv_mfma_workspace_bytes_h
#define dcs_conv_mfma_workspace_bytes_plain dcs_conv_mfma_workspace_bytes_plain_h
#define dcs_conv_mfma_launch dcs_conv_mfma_launch_h
#define dcs_conv_mfma_launch_split dcs_conv_mfma_launch_split_h
#define dcs_conv_mfma_launch_wide dcs_conv_mfma_launch_wide_h
#define dcs_conv_mfma_launch_classes dcs_conv_mfma_launch_classes_h
#define dcs_conv_enc0_ok dcs_conv_enc0_ok_h
#define dcs_conv_enc0_launch dcs_conv_enc0_launch_h
#define dcs_conv_enc0_stat_rows dcs_conv_enc0_stat_rows_h
#define dcs_conv_enc0_wgrad_ok dcs_conv_enc0_wgrad_ok_h
#define dcs_conv_enc0_wgrad_launch dcs_conv_enc0_wgrad_launch_h
#define dcs_conv_wgrad_mfma_ok dcs_conv_wgrad_mfma_ok_h
#define dcs_conv_wgrad_mfma_slabs dcs_conv_wgrad_mfma_slabs_h
#define dcs_conv_wgrad_mfma_launch dcs_conv_wgrad_mfma_launch_h
#define dcs_conv_wgrad_mfma_planes_bytes dcs_conv_wgrad_mfma_planes_bytes_h
#define dcs_conv_wgrad_fold_ok dcs_conv_wgrad_fold_ok_h
#define dcs_conv_wgrad_fold_workspace_bytes dcs_conv_wgrad_fold_workspace_bytes_h
#define dcs_conv_wgrad_fold_run dcs_conv_wgrad_fold_run_h
#define dcs_conv_small_dgrad_ok dcs_conv_small_dgrad_ok_h
#define dcs_conv_small_dgrad_launch dcs_conv_small_dgrad_launch_h
#endif

namespace conv {

struct Args {
    const act2_t* x1; const act2_t* x2; const float2* wp; const float2* bias; act2_t* y;      // (activations: act_t pairs)
    int B, Hin, Win, C1, C2, up_f, up_t, zero_ins, Cout, kh, kw, sf, st, pad_f, pad_t, act;
    int Hv, Wv, Hout, Wout, tiles_w, tiles_h, rows, cols, colsp, plane;
    // optional per-output-channel real 2x2 affine applied between bias and activation (forward epilogues only):
    // (re, im) <- (a0 re + a1 im + c0, a2 re + a3 im + c1), float[Cout][6] = {a0, a1, a2, a3, c0, c1} — an eval-mode
    // ComplexBatchNorm2d folded into the conv that feeds it (dcs_cconv2d_fwd_affine)
    const float* coef;
    // optional CBN statistics of the raw output (training: the ComplexBatchNorm2d that follows reads them instead of a
    // pass over y): float[Cout][5][stat_stride], column `row` = one workgroup's partial {S_r, S_i, S_rr, S_ii, S_ri} of
    // (y - bias) over its valid output pixels (rows contiguous: the finalize kernel's reads coalesce), fixed order, no
    // atomics.  Forward launches with act = NONE and no coef only.
    float* stat;
    int stat_stride;
};

// One output-parity class of a decomposed convolution (conv_mfma.hip): its own sub-kernel, padding,
// extent and weight panel; its pixel (oy, ox) lands at (oy*os + oo) of the full output.
struct Cls {
    int kh, kw, pad_f, pad_t, oo_f, oo_t, Hc, Wc;
    long bm_off;              // float offset of this class's MFMA panel from the `bm` passed to the launch
};

// Element (b, vy, vx, c) of the virtual input: nearest upsample of cat(x1, x2) (c_network.py:214-216),
// or — for data gradients — g_Y with (up_f-1, up_t-1) zeros inserted between samples.  Zero outside.
// (the factors are 1 or 2 on every path of the network: shifts / masks there, a real division only otherwise — this
// runs per patch pixel per tile in the weight-gradient kernels)
__device__ __forceinline__ int div_up(int v, int up) { return up == 1 ? v : (up == 2 ? v >> 1 : v / up); }
__device__ __forceinline__ int mod_up(int v, int up) { return up == 1 ? 0 : (up == 2 ? (v & 1) : v % up); }
__device__ __forceinline__ bool src_pixel(const Args& a, int b, int vy, int vx, long* sp) {
    if (vy < 0 || vy >= a.Hv || vx < 0 || vx >= a.Wv) return false;
    if (a.zero_ins && (mod_up(vy, a.up_f) != 0 || mod_up(vx, a.up_t) != 0)) return false;
    *sp = ((long)b * a.Hin + div_up(vy, a.up_f)) * a.Win + div_up(vx, a.up_t);
    return true;
}

// one complex activation value
__device__ __forceinline__ float2 ldc(const float2* p) { return *p; }
__device__ __forceinline__ float2 ldc(const unsigned* p) { return dcs_ld2(reinterpret_cast<const unsigned short*>(p)); }
__device__ __forceinline__ void stc(float2* p, float2 v) { *p = v; }
__device__ __forceinline__ void stc(unsigned* p, float2 v) { *p = dcs_pack_bf16x2(v.x, v.y); }

__device__ __forceinline__ float2 gather(const Args& a, int b, int vy, int vx, int c) {
    long sp;
    if (!src_pixel(a, b, vy, vx, &sp)) return make_float2(0.f, 0.f);
    return (c < a.C1) ? ldc(a.x1 + sp * a.C1 + c) : ldc(a.x2 + sp * a.C2 + (c - a.C1));
}

// MFMA eligibility of a (Cin, Cout) pair: K groups of 4 complex input channels staged 8 at a time,
// N = 2*Cout real columns in tiles of 32 (a 16-wide remainder is zero-padded).
inline bool mfma_ok(int Cin, int Cout) { return (Cin % 8) == 0 && (Cout % 8) == 0; }
// precision mode of the MFMA forward / data-gradient GEMM with K = 2*Cin: the bf16x6 emulation (mode 2) needs 16-channel
// chunks to amortise its per-iteration cost — 8-channel layers (enc1: the fp32 kernel runs a whole kernel row per
// iteration there) stay on the fp32 MFMA.  Decides the panel layout (pack) and the kernel (launch) alike.
inline int mfma_precision(int Cin, int taps = 0) {
    const int p = dcs_conv_precision();
    return (p == 2 && (Cin % 16) != 0 && !(Cin == 8 && (taps == 49 || taps == 16))) ? 0 : p;      // (8 channels: only the row forms, 7x7 and 4x4)
}
// ... of the 16-column kernel (Cout = 8): its bf16 forms (1: bf16 operands, 2: the emulation) need 16-channel k-groups
inline int mfma_precision16(int Cin) {
    const int p = dcs_conv_precision();
    return (Cin % 16) == 0 ? p : 0;
}
inline long direct_floats(int Cout, int Cin, int taps) { return (long)taps * Cin * Cout * 2; }
inline long mfma_floats(int Cout, int Cin, int taps) {      // N = 2*Cout padded to whole 32-column tiles
    // (precision mode 2 keeps three bf16 planes, each half an fp32 panel: 1.5 x)
    const long n = mfma_ok(Cin, Cout) ? (long)taps * (Cin / 4) * ((2 * Cout + 31) / 32) * 256 : 0;
    return dcs_conv_precision() == 2 ? n + n / 2 : n;
}

// conv_pack.hip: derived panels (upsample fold, its data gradient, strided data gradient)
struct Axis { int first, count, pad, n; };
bool fold_ok(int Cin, int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int up_f, int up_t);
long fold_floats(int Cout, int Cin, int up_f, int up_t);
int pack_fold(const float* wp, float* region, int Cout, int Cin, int up_f, int up_t, hipStream_t s);
void fold_classes(int Cout, int Cin, int up_f, int up_t, int Hin, int Win, Cls* cls);
long upfold_bwd_floats(int Cout, int Cin, int up_f, int up_t);
int pack_upfold_bwd(const float* wp, float* region, int Cout, int Cin, int up_f, int up_t, hipStream_t s);
Axis stride_axis(int k, int s, int pad, int r, int full);
bool stride_ok(int Cin, int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t);
long stride_bwd_floats(int Cout, int Cin, int kh, int kw, int sf, int st, int pad_f, int pad_t);
int pack_stride_bwd(const float* wp_bwd, float* region, int Cout, int Cin, int kh, int kw, int sf, int st, int pad_f,
                    int pad_t, hipStream_t s);
void stride_classes(int Cout, int Cin, int kh, int kw, int sf, int st, int pad_f, int pad_t, int Hv, int Wv, Cls* cls);

}  // namespace conv

// conv_wgrad_mfma.hip
bool dcs_conv_wgrad_mfma_ok(int Cin, int Cout, int kh, int kw, int C1);
int dcs_conv_wgrad_mfma_slabs(const conv::Args& a, int* TH, int* TW);
int dcs_conv_wgrad_mfma_launch(conv::Args& a, const act_t* gy, float2* slab_w, float* slab_b, int n_slabs,
                               hipStream_t stream, void* planes = nullptr);
// bytes of the pre-split g_Y (three bf16 planes in A-fragment order) the emulated kernel of this geometry can use, 0 if none
long dcs_conv_wgrad_mfma_planes_bytes(const conv::Args& a);

// conv_wgrad_mfma.hip: weight gradient of a 3x3 conv over an upsampled input in its folded (per-class, source-
// resolution) form; kernel + reduce into the parameter layout
bool dcs_conv_wgrad_fold_ok(const conv::Args& a);
long dcs_conv_wgrad_fold_workspace_bytes(const conv::Args& a);
int dcs_conv_wgrad_fold_run(const conv::Args& a, const act_t* gy, void* workspace, long workspace_bytes, float* gw_r,
                            float* gw_i, float* gb_r, float* gb_i, int transposed, hipStream_t stream);

// conv_wgrad_small.hip (a: forward geometry with conv_direct.hip's 16x16 tiling filled in)
bool dcs_conv_wgrad_small_ok(const conv::Args& a);
int dcs_conv_wgrad_small_launch(const conv::Args& a, const float* gy, float2* slab_w, float2* slab_b, int n_slabs,
                                hipStream_t stream);

// conv_small.hip: class-decomposed data gradient of the 1->8 stride-2 7x7 conv
bool dcs_conv_small_dgrad_ok(int Cin, int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int up_f, int up_t);
int dcs_conv_small_dgrad_launch(const act_t* gy, const float* wp_bwd, act_t* gx, int B, int Hx, int Wx, int Hg, int Wg,
                                int pad_f, int pad_t, hipStream_t stream);

// conv_direct.hip: n small direct correlations (same batch, Cout 1 or 2) in one launch
int dcs_conv_direct_multi(conv::Args* a, int n, hipStream_t stream);
// conv_k7.hip: register-blocked 7x7 / stride 1 / pad 3 complex convs with (Cin, Cout) in {(2,1), (1,2), (1,1)}
bool dcs_conv_k7_ok(const conv::Args* a, int n);
int dcs_conv_k7_launch(const conv::Args* a, int n, hipStream_t stream);

// conv_enc0.hip: the 1 -> 8 channel, 7x7, stride-2, pad-3 forward conv with the taps as the MFMA K axis
bool dcs_conv_enc0_ok(const conv::Args& a);
int dcs_conv_enc0_launch(conv::Args a, hipStream_t stream);
int dcs_conv_enc0_stat_rows(const conv::Args& a);          // rows of a.stat the launch writes
bool dcs_conv_enc0_wgrad_ok(const conv::Args& a);
int dcs_conv_enc0_wgrad_launch(conv::Args a, const act_t* gy, float2* slab_w, float2* slab_b, int max_slabs, int* n_used,
                               hipStream_t stream);

// conv_mfma.hip
// rows of a.stat the launch of (a, classes) writes (given the split-K scratch it would be handed), 0: no statistics path
int dcs_conv_mfma_stat_rows(const conv::Args& a, int ncls, const conv::Cls* cls, bool have_ws);
int dcs_conv_mfma_pack(const float* wp_direct, float* bm, int Cout, int Cin, int taps, hipStream_t stream);
// split-K scratch (bytes) the launch of this geometry would use; ws / ws_bytes below: that scratch (optional)
long dcs_conv_mfma_workspace_bytes(const conv::Args& a, int ncls, const conv::Cls* cls);
long dcs_conv_mfma_workspace_bytes_plain(const conv::Args& a);
int dcs_conv_mfma_launch(conv::Args& a, const float* bm, void* ws, long ws_bytes, hipStream_t stream);
int dcs_conv_mfma_launch_split(conv::Args& a, const float* bm, act_t* y2, int nsplit, void* ws, long ws_bytes,
                               hipStream_t stream);
int dcs_conv_mfma_launch_wide(conv::Args& a, const float* bm, void* ws, long ws_bytes, hipStream_t stream);
int dcs_conv_mfma_launch_classes(conv::Args& a, const float* bm, int ncls, const conv::Cls* cls, int os_f, int os_t,
                                 act_t* y2, int nsplit, void* ws, long ws_bytes, hipStream_t stream);
