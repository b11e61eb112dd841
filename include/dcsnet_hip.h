/*
 * dcsnet_hip.h — C ABI of libdcsnet_hip.so: the MI355X (gfx950) implementation of the
 * DCS-Net complex encoder/decoder hot path.
 *
 * The reference (jackhwalters/DCS-Net) is pure Python; it has no FFI of its own.  Every
 * entry point below replaces a group of implicit ATen/cuDNN dispatches that the
 * reference issues from Python, and cites the reference call site it stands in for.
 * The reference-side binding (a ctypes stub a maintainer would add) is in INTEGRATION.md;
 * this repo's own binding is dcs-net_amd/dcsnet/_lib.py.
 *
 * Conventions
 *   - extern "C", plain pointers and ints.  No torch / HIP types in any signature:
 *     `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - Every pointer is a DEVICE pointer BORROWED from the caller.  The library never
 *     allocates, frees or retains caller memory; workspaces are passed in.
 *   - Kernels are enqueued on `stream`; no call synchronises.
 *   - ONE CONTEXT PER PROCESS.  The library is built for the one-process-per-GPU model and keeps four pieces of
 *     process-global mutable state (each behind its own entry points, each guarded by a mutex, none per stream):
 *       (1) the conv operand precision                         dcs_set_conv_precision / dcs_get_conv_precision
 *       (2) the deferred weight-gradient reduce scope           dcs_wgrad_defer_begin / _suspend / _flush
 *       (3) the pack-plan recorder                              dcs_pack_plan_begin / _end
 *       (4) the kernel timer's armed slot                       dcs_kernel_timer_begin / _end
 *     Two training drivers in one process (or two host threads that open scopes concurrently) would see each
 *     other's precision, recorded reduces and recorded packs.  Everything else is stateless: kernels read only
 *     their arguments, so forward / backward calls on different streams of one process are safe as long as they
 *     share the precision mode and at most one of them has a defer scope or a plan recording open.
 *   - Return value: 0 = enqueued; <0 = error (DCS_ERR_*), nothing was enqueued.
 *   - Dropout: mask = hash(seed + *seed_dev, element index); `seed_dev` (device uint64, may be NULL) lets
 *     a captured hipGraph be replayed with a fresh mask every step.  Backward calls take the same pair.
 *   - Complex tensors are interleaved (re, im) fp32 pairs — the memory of a torch
 *     complex64 tensor.  Activations are CHANNELS-LAST: x[b][f][t][c] complex,
 *     i.e. float[B][F][T][C][2]  (a torch complex64 [B,C,F,T] tensor in
 *     torch.channels_last memory format).  "pixel" = one (b,f,t) position.
 *   - Parameters keep the reference's checkpoint layout (complexPyTorch 0.3):
 *     conv_r/conv_i weight float[Cout][Cin][kh][kw]; conv_tran_r/_i weight
 *     float[Cin][Cout][kh][kw]; CBN weight float[C][3], bias float[C][2],
 *     running_mean complex[C], running_covar float[C][3].
 */
#ifndef DCSNET_HIP_H
#define DCSNET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dcs_stream_t;

#define DCS_OK              0
#define DCS_ERR_BADARG     (-1)   /* null pointer, non-positive dim, unsupported geometry */
#define DCS_ERR_LAUNCH     (-2)   /* hipLaunch reported an error */
#define DCS_ERR_WORKSPACE  (-3)   /* workspace too small */

/* activation codes shared by several entry points */
#define DCS_ACT_NONE   0
#define DCS_ACT_RELU   1   /* complexPyTorch complex_relu: relu on re and on im (config.py:103) */
#define DCS_ACT_LRELU  2   /* network_functions.py:103-105, slope 0.01 (config.py:104)          */
#define DCS_ACT_SIGMOID 3  /* network_functions.py:111-112                                       */

int         dcs_abi_version(void);            /* bumps when a signature changes */
const char* dcs_error_string(int code);

/* ------------------------------------------------------------------------------------
 * Weight packing.  Replaces nothing in the reference (it keeps two real layers per
 * complex layer, c_network.py:107-112 / :135-147); this turns the pair (w_r, w_i) into
 * ONE correlation weight the conv kernels consume:
 *     wp[tap][ci][co] = (w_r + j w_i)           complex, tap = dy*kw + dx
 * transposed != 0: the inputs are ConvTranspose2d weights [Cin][Cout][kh][kw]; for the
 * stride-1 transposed convs of the decoder (config.py:84,92-100) the equivalent
 * correlation kernel is the spatially flipped, in/out-swapped one, which this writes.
 * bias_out[co] = (b_r - b_i) + j (b_r + b_i)   — both real layers carry a bias
 * (SURVEY.md §8a a2); b_r/b_i may be NULL (bias=False: attention convs c_network.py:58-60,74).
 * wp: complex[kh*kw][Cin][Cout]; bias_out: complex[Cout] (always written, zeros if no bias).
 * When Cin % 8 == 0 and Cout % 8 == 0 a second panel follows in the same buffer: the 2x2
 * real-embedded weight in MFMA fragment order for the implicit-GEMM kernel.  Allocate
 * dcs_packed_weight_floats(Cout, Cin, kh, kw) floats for wp (same for wp_bwd with Cout/Cin swapped);
 * the conv entry points find the second panel themselves.
 * (up_f, up_t): the nearest-upsample factors of the dcs_cconv2d_fwd call this weight will be used with
 * (1,1 = none).  For a 3x3 / stride-1 / pad-1 conv behind a x2 upsample a third panel is appended: the
 * upsample FOLDED into per-output-parity sub-kernels with pre-summed taps (y[2m] = W0 x[m-1] + (W1+W2) x[m],
 * y[2m+1] = (W0+W1) x[m] + W2 x[m+1]), which dcs_cconv2d_fwd then runs on the source tensors.  A weight
 * must be used with the same (up_f, up_t) it was packed for.
 */
long dcs_packed_weight_floats(int Cout, int Cin, int kh, int kw, int up_f, int up_t);
int dcs_pack_conv_weight(const float* w_r, const float* w_i, const float* b_r, const float* b_i,
                         float* wp, float* bias_out,
                         int Cout, int Cin, int kh, int kw, int transposed, int up_f, int up_t,
                         dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * ComplexConv2d / ComplexConvTranspose2d forward (apply_complex of complexPyTorch 0.3).
 * Replaces: c_network.py:107-112 via encoder[i][0] (c_network.py:194), the decoder's
 * conv_tran at c_network.py:135-147 via decoder[i] (c_network.py:217) TOGETHER WITH the
 * torch.cat at c_network.py:214 and complex_upsample at c_network.py:215-216, and the
 * attention convs at c_network.py:58-60,74.
 *
 * The logical input is X[b][iy][ix][ci], ci in [0, C1+C2):
 *     ci <  C1 : x1[b][iy/up_f][ix/up_t][ci]          (decoder path "d")
 *     ci >= C1 : x2[b][iy/up_f][ix/up_t][ci-C1]       (skip path; x2 may be NULL iff C2==0)
 * with x1: complex[B][Hin][Win][C1], x2: complex[B][Hin][Win][C2], nearest-neighbour
 * upsampling by (up_f, up_t) >= 1.  Output y: complex[B][Hout][Wout][Cout],
 *     Hout = (Hin*up_f + 2*pad_f - kh)/sf + 1,  Wout likewise,
 *     y[b,oy,ox,co] = bias[co] + sum_{dy,dx,ci} wp[dy*kw+dx][ci][co] * X[b, oy*sf-pad_f+dy, ox*st-pad_t+dx, ci]
 * (complex product; X is zero outside its bounds).  act: DCS_ACT_* applied to re and im.
 * wp/bias as written by dcs_pack_conv_weight.
 * workspace: optional split-K scratch of dcs_cconv2d_fwd_workspace_bytes() bytes (0 for most geometries).  Deep
 * layers at small batch have too few output tiles to fill 256 CUs; with the scratch their input-channel range is
 * sliced over extra workgroups and a reduce pass adds the slices (+ bias, activation) in a fixed order.  NULL or too
 * small: the layer simply runs unsliced.
 */
long dcs_cconv2d_fwd_workspace_bytes(int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                                     int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t);
int dcs_cconv2d_fwd(const float* x1, const float* x2, const float* wp, const float* bias, float* y,
                    void* workspace, long workspace_bytes,
                    int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                    int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int act,
                    dcs_stream_t stream);

/* Training-mode conv -> ComplexBatchNorm2d pairs (reference: c_network.py:107-114, :135-150 — ComplexConv2d /
 * ComplexConvTranspose2d followed at once by ComplexBatchNorm2d): dcs_cconv2d_fwd (act = NONE) that also leaves the batch
 * statistics of its raw output, taken from the fp32 accumulators in the epilogue, so the CBN needs no pass over y:
 *   stat        float[Cout][5][stat_rows]: column r = one workgroup's partial sums {S_r, S_i, S_rr, S_ii, S_ri} of (y - bias)
 *               over its valid output pixels — fixed summation order, no atomics (bitwise reproducible)
 *   stat_rows   capacity of `stat` in rows, >= dcs_cconv2d_fwd_stats_rows(geometry) (0 there: this geometry has no
 *               statistics epilogue — use dcs_cconv2d_fwd + dcs_cbn_fwd)
 *   rows_used   HOST int, out: rows written (depends on whether the split-K scratch was handed over)
 * Consumer: dcs_cbn_fwd_slabs(part = stat, rows = *rows_used, stride = stat_rows, pivot = bias). */
int dcs_cconv2d_fwd_stats_rows(int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                               int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t);
int dcs_cconv2d_fwd_stats(const float* x1, const float* x2, const float* wp, const float* bias, float* y,
                          float* stat, int stat_rows, int* rows_used, void* workspace, long workspace_bytes,
                          int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                          int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, dcs_stream_t stream);

/* dcs_cconv2d_fwd with an eval-mode ComplexBatchNorm2d folded into the epilogue: between bias and activation every output
 * channel goes through the real 2x2 affine map  (re, im) <- (a0 re + a1 im + c0, a2 re + a3 im + c1),
 * coef float[Cout][6] = {a0, a1, a2, a3, c0, c1} — exactly the `coef_out` an eval-mode dcs_cbn_fwd writes (running statistics
 * are constants at inference, so conv + CBN + activation is one kernel and one pass over the activation).  coef == NULL:
 * dcs_cconv2d_fwd. */
int dcs_cconv2d_fwd_affine(const float* x1, const float* x2, const float* wp, const float* bias, const float* coef, float* y,
                           void* workspace, long workspace_bytes,
                           int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                           int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int act,
                           dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Real-valued convolution (DR-Net / DRS-Net: torch.nn.Conv2d / ConvTranspose2d of r_network.py:60-66, :90-102) on the
 * same fp32 MFMA implicit-GEMM kernel.  A real channels-last activation float[B][H][W][Cr] with even Cr is bit-for-bit
 * an interleaved complex one with Cr/2 channels; the caller supplies the GEMM's B panel directly instead of a packed
 * complex weight:
 *   bm    float[taps][K/8 .. see below]: B[tap][k][n] = w[n][k][dy][dx] (k = real input channel of cat(x1, x2),
 *         n = real output channel) in 32-column fragment order: element (tap, kg, nt, lane, e) with lane = 32*kk + j
 *         holds B[tap][8*kg + 4*kk + e][32*nt + j], kg < (C1r+C2r)/8, nt < ceil(Coutr/32), zero beyond Coutr
 *   bias  float[Coutr] or NULL
 * Geometry as dcs_cconv2d_fwd with real channel counts: (C1r + C2r) % 16 == 0, Coutr % 16 == 0, C1r % 4 == 0.
 * ConvTranspose2d (stride 1): pass the flipped, in/out-swapped kernel and padding k-1-p.
 * dcs_rconv2d_bwd_data: gradient of the VIRTUAL input (the upsampled concatenation, Hv = Hin*up_f, Wv = Win*up_t) of the
 *   forward call with real geometry (Cinr = C1r + C2r, Coutr, k, stride, pad): gxv float[B][Hv][Wv][Cinr] = stride-1
 *   correlation of the zero-inserted g_Y with bm_bwd = the panel (same fragment order) of the flipped, in/out-swapped
 *   kernel, B[tap][k = real output channel][n = real input channel].  Block sum over the upsample and channel split of
 *   the concatenation: dcs_upsample_cat_bwd with complex channel counts C1r/2, C2r/2.  (The weight gradient of a real
 *   conv comes from two dcs_cconv2d_bwd_weight calls, on x and on conj(x): with D_qr = sum g_q x_r the complex
 *   gradients are (D_rr + D_ii) + j (D_ir - D_ri) and (D_rr - D_ii) + j (D_ir + D_ri) — dcsnet/r_network.py.) */
long dcs_rconv2d_fwd_workspace_bytes(int B, int Hin, int Win, int C1r, int C2r, int up_f, int up_t, int Coutr,
                                     int kh, int kw, int sf, int st, int pad_f, int pad_t);
int dcs_rconv2d_fwd(const float* x1, const float* x2, const float* bm, const float* bias, float* y,
                    void* workspace, long workspace_bytes,
                    int B, int Hin, int Win, int C1r, int C2r, int up_f, int up_t,
                    int Coutr, int kh, int kw, int sf, int st, int pad_f, int pad_t, int act,
                    dcs_stream_t stream);
long dcs_rconv2d_bwd_data_workspace_bytes(int B, int Hv, int Wv, int Cinr, int Coutr, int kh, int kw, int sf, int st,
                                          int pad_f, int pad_t);
int dcs_rconv2d_bwd_data(const float* gy, const float* bm_bwd, float* gxv, void* workspace, long workspace_bytes,
                         int B, int Hv, int Wv, int Cinr, int Coutr, int kh, int kw, int sf, int st, int pad_f, int pad_t,
                         dcs_stream_t stream);

/* torch.nn.BatchNorm2d (+ ReLU / LeakyReLU) of a REAL channels-last tensor float[P][Cr] (DR-Net: r_network.py:56,66,106)
 * on the CBN kernels: with even Cr the tensor is an interleaved complex one with Cr/2 channels, dcs_cbn_fwd's statistics
 * pass yields every real channel's moments, and the affine map is the diagonal 2x2 block a = gamma / sqrt(var + eps).
 * Cr == 1 (the initial BatchNorm over the magnitude [B,F,T]; P % 4 == 0): the P values are read as P/2 complex pixels
 * whose halves are the same channel.  weight / bias / running_mean / running_var: float[Cr] (the module's own
 * buffers; running_var updated with the unbiased n/(n-1) estimate, as torch does).  stats_out float[ceil(Cr/2)][8],
 * coef_out float[ceil(Cr/2)][6]: saved for dcs_rbn_bwd.  workspace: dcs_cbn_workspace_bytes(P or P/2, Cr/2 or 1).
 * dcs_rbn_bwd: g_x (may alias g_out), g_weight / g_bias float[Cr] (NULL, NULL for affine=False);
 * workspace: dcs_cbn_bwd_workspace_bytes of the same (pixels, complex channels). */
int dcs_rbn_fwd(const float* x, float* y, const float* weight, const float* bias, float* running_mean, float* running_var,
                float* stats_out, float* coef_out, void* workspace, long workspace_bytes, long P, int Cr, float eps,
                float momentum, int use_batch_stats, int act, dcs_stream_t stream);
int dcs_rbn_bwd(const float* x, const float* g_out, float* g_x, const float* stats, const float* coef, float* g_weight,
                float* g_bias, void* workspace, long workspace_bytes, long P, int Cr, int use_batch_stats, int act,
                dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Gradients of dcs_cconv2d_fwd (what torch.autograd derives for the reference through the four
 * real convolutions of apply_complex; gradients of complex tensors are dL/dRe + j dL/dIm).
 *
 * dcs_pack_conv_weight_bwd: wp_bwd[tap'][co][ci] = conj(wp[ntaps-1-tap'][ci][co]), its MFMA panel, and — given
 *     the forward geometry — the derived panels that skip structurally zero work: one compact sub-kernel per
 *     input-pixel residue class for a strided conv (instead of zero insertion), or the effective 4-tap
 *     stride-2 kernel conj[W2, W1+W2, W0+W1, W0] for the conv behind a x2 nearest upsample.
 *     Allocate dcs_packed_weight_bwd_floats(...) floats.
 * dcs_cconv2d_bwd_data:  g_x1 (complex[B][Hin][Win][C1]) and g_x2 (complex[B][Hin][Win][C2], NULL iff C2 == 0):
 *     gradients of the two inputs of the forward call with the same geometry arguments,
 *       g_Xv[b,vy,vx,ci] = sum_{p,tap,co} conj(wp[tap][ci][co]) g_Y[p,co]   over p*s - pad + tap = (vy,vx),
 *     block-summed over the upsample factors and split at channel C1.  workspace: the first part (required when
 *     the virtual-input gradient has to exist: cat or upsample on a non-folded geometry) holds g_Xv, the rest is the
 *     optional split-K scratch of the deep few-pixel layers (as in dcs_cconv2d_fwd);
 *     dcs_cconv2d_bwd_data_workspace_bytes returns the sum (0 for most geometries).
 * dcs_upsample_cat_bwd:  the stand-alone block sum / channel split of a virtual-input gradient.
 * dcs_cconv2d_bwd_weight: gradients of the reference's parameters, in THEIR layout:
 *     gw_r, gw_i: float[Cout][Cin][kh][kw] (transposed=0) or float[Cin][Cout][kh][kw] (transposed=1)
 *     gb_r, gb_i: float[Cout] (NULL for bias-free layers).  x1/x2/geometry as in the forward call.
 *     workspace >= dcs_cconv2d_bwd_weight_workspace_bytes(...): the partial slabs (no atomics) and, behind them, the cotangent
 *     split once into its three bf16 planes for layers of >= 64 input channels in the emulated-fp32 mode (a workspace that
 *     holds the slabs only is accepted: the kernels then split in place).  The call reads x1 / x2 / gy and the workspace
 *     until its stream has run it — on a side stream (as dcsnet/dp.py issues it) keep them alive until that stream is joined. */
long dcs_packed_weight_bwd_floats(int Cout, int Cin, int kh, int kw, int sf, int st, int pad_f, int pad_t,
                                  int up_f, int up_t);
int  dcs_pack_conv_weight_bwd(const float* wp, float* wp_bwd, int Cout, int Cin, int kh, int kw,
                              int sf, int st, int pad_f, int pad_t, int up_f, int up_t, dcs_stream_t stream);
long dcs_cconv2d_bwd_data_workspace_bytes(int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                                          int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t);
int  dcs_cconv2d_bwd_data(const float* gy, const float* wp_bwd, float* gx1, float* gx2,
                          void* workspace, long workspace_bytes,
                          int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                          int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, dcs_stream_t stream);
int  dcs_upsample_cat_bwd(const float* gxv, float* gx1, float* gx2, int B, int Hin, int Win, int C1, int C2,
                          int up_f, int up_t, dcs_stream_t stream);
long dcs_cconv2d_bwd_weight_workspace_bytes(int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                                            int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t);
int  dcs_cconv2d_bwd_weight(const float* x1, const float* x2, const float* gy,
                            float* gw_r, float* gw_i, float* gb_r, float* gb_i,
                            void* workspace, long workspace_bytes,
                            int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                            int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int transposed,
                            dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * ComplexBatchNorm2d forward (complexPyTorch 0.3), fused with the activation and the
 * dropout that follow it in the reference.
 * Replaces: c_network.py:101 (initial_batchnorm, :190), :113-114 (encoder CBN + CReLU),
 * :148-150 (decoder CBN + CLReLU), and dropout_conv on the real view (:195-196).
 *
 * Training (use_batch_stats != 0): per-channel complex mean and biased 2x2 covariance over
 * all B*F*T pixels, eps added to Crr and Cii; running_mean / running_covar updated with
 * `momentum` and the n/(n-1) factor (skipped when momentum < 0).  Eval: running stats.
 * Then whiten with the closed-form inverse matrix square root and apply the symmetric
 * affine weight [C][3] and bias [C][2]; y = act(.) (DCS_ACT_NONE/RELU/LRELU), then inverted
 * dropout (keep 1-drop_p, independent on re and im, mask = hash(seed, float index)) if drop_p > 0.
 *   x, y        complex[P][C], P = B*F*T pixels   (y may alias x)
 *   stats_out   float[C][8]: mean_r, mean_i, Rrr, Rii, Rri, Crr, Cii, Cri (saved for backward)
 *   coef_out    float[C][6]: y_r = a0 x_r + a1 x_i + c0 ; y_i = a2 x_r + a3 x_i + c1, then act
 *   workspace   >= dcs_cbn_workspace_bytes(P, C) bytes, 16-byte aligned
 * use_batch_stats == 2: apply only — coef_out already holds the coefficients an earlier eval-mode call
 * (use_batch_stats == 0) wrote for the same parameters and running statistics (inference: they are constants).
 */
long dcs_cbn_workspace_bytes(long P, int C);
int  dcs_cbn_fwd(const float* x, float* y, const float* weight, const float* bias,
                 float* running_mean, float* running_covar,
                 float* stats_out, float* coef_out, void* workspace, long workspace_bytes,
                 long P, int C, float eps, float momentum, int use_batch_stats, int act,
                 float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream);

/* Training-mode dcs_cbn_fwd whose batch statistics were left by the conv that produced x (dcs_cconv2d_fwd_stats):
 * part float[C][5][stride] partial sums of (x - pivot), columns 0..rows-1 valid, pivot float[C][2] (that conv's packed bias).
 * Finalize + apply only. */
int  dcs_cbn_fwd_slabs(const float* x, float* y, const float* weight, const float* bias,
                       float* running_mean, float* running_covar, float* stats_out, float* coef_out,
                       const float* part, int rows, int stride, const float* pivot,
                       long P, int C, float eps, float momentum, int act,
                       float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream);

/* A decoder stage's tail (c_network.py:148-150, :219: ComplexBatchNorm2d + CLReLU, then a channel attention that starts with a
 * per-sample average pool of that output): dcs_cbn_fwd_slabs whose apply pass also leaves the pooling slabs of its OUTPUT,
 * pool_part = double[B][chunks][C][2] with chunks = dcs_ca_pool_chunks(HW, C), and the attention's FC half on those slabs
 * (dcs_channel_attention_fwd = pool + FC): three launches for CBN + channel attention instead of four, one read of the
 * activation less.  x = complex[B][HW][C]; no dropout. */
int dcs_ca_pool_chunks(long HW, int C);
int dcs_cbn_fwd_slabs_pool(const float* x, float* y, const float* weight, const float* bias,
                           float* running_mean, float* running_covar, float* stats_out, float* coef_out,
                           const float* part, int rows, int stride, const float* pivot, void* pool_part, long pool_bytes,
                           int B, long HW, int C, float eps, float momentum, int act, dcs_stream_t stream);
int dcs_channel_attention_fc_fwd(const void* pool_part, const float* w1, const float* w2, float* ca_out, float* pooled_out,
                                 float* hidden_out, int B, long HW, int C, int Ch, dcs_stream_t stream);

/* Backward of dcs_cbn_fwd (closed form of what autograd derives through complexPyTorch's CBN,
 * the activation and the dropout).  g_out: gradient w.r.t. y; g_x: gradient w.r.t. x (may alias
 * g_out; NULL = parameter gradients only: the CBN of the network input, no pass over x for g_x); stats/coef: as written by the forward call; g_weight float[C][3], g_bias float[C][2]
 * (both NULL for affine=False); same act / drop_p / seed as the forward call.
 * use_batch_stats = 0 differentiates the eval-mode (running statistics) normalisation. */
long dcs_cbn_bwd_workspace_bytes(long P, int C);
int  dcs_cbn_bwd(const float* x, const float* g_out, float* g_x, const float* weight,
                 const float* stats, const float* coef, float* g_weight, float* g_bias,
                 void* workspace, long workspace_bytes, long P, int C, int use_batch_stats, int act,
                 float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream);
/* dcs_cbn_bwd of g_out[b][p][c] + g_out2[b][p][c] + add_scale * g_add[b][c] (g_add complex[B][C], HW pixels per
 * sample, P = B*HW; g_out2 shaped like g_out; either may be NULL):
 *   g_add   the broadcast half of an average pool's backward folded into this consumer (see dcs_attention_bwd_x);
 *   g_out2  the cotangent of a SECOND consumer of y — an encoder stage's output feeds the next conv and a skip
 *           attention (c_network.py:193-197, :208-211): autograd would first add the two with an element-wise kernel. */
int  dcs_cbn_bwd_add(const float* x, const float* g_out, float* g_x, const float* weight, const float* stats,
                     const float* coef, float* g_weight, float* g_bias, void* workspace, long workspace_bytes,
                     long P, int C, int use_batch_stats, int act, float drop_p, unsigned long long seed,
                     const unsigned long long* seed_dev, const float* g_add, float add_scale, long HW,
                     const float* g_out2, dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * ComplexChannelAttention (c_network.py:53-69): per sample, mean over (F,T) of every
 * channel, the two bias-free 1x1 complex convs with CReLU between, and the complex sigmoid.
 * QUIRK kept: the "max" branch is an average pool (network_functions.py:135-138), so the
 * result is sigmoid_c(fc(avg) + fc(avg)).
 *   x        complex[B][HW][C]
 *   w1       complex[C][Ch]   packed 1x1 weight of fc.0 (dcs_pack_conv_weight, tap dim = 1)
 *   w2       complex[Ch][C]   packed 1x1 weight of fc.2
 *   ca_out   complex[B][C]
 *   pooled_out complex[B][C]  (saved for backward), hidden_out complex[B][Ch] (pre-ReLU, saved)
 *   workspace >= dcs_ca_workspace_bytes(B, HW, C)
 */
long dcs_ca_workspace_bytes(int B, long HW, int C);
int  dcs_channel_attention_fwd(const float* x, const float* w1, const float* w2,
                               float* ca_out, float* pooled_out, float* hidden_out,
                               void* workspace, long workspace_bytes,
                               int B, long HW, int C, int Ch, dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * ComplexSpatialAttention, first half (c_network.py:77-82): with z = ca[b,c] * x[p,c]
 * (complex product; ca may be NULL = 1), per pixel
 *     pooled[p][0] = mean_c z ;  pooled[p][1] = max_c Re z + j max_c Im z
 * The 7x7 2->1 conv + sigmoid of c_network.py:83-84 is dcs_cconv2d_fwd(act=SIGMOID).
 *   x complex[B][HW][C]; ca complex[B][C]; pooled complex[B][HW][2]
 */
int dcs_spatial_pool_fwd(const float* x, const float* ca, float* pooled,
                         int B, long HW, int C, dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Attention application (c_network.py:209,211 and :219-220) fused with dropout_conv
 * (:221-222):  y[p][c] = sa[p] * (ca[b][c] * x[p][c])   (two true complex products),
 * then inverted dropout with keep probability 1-p independently on re and im when p > 0
 * (mask = hash(seed, element index); regenerated, not stored).  y may alias x.
 */
int dcs_attention_apply_fwd(const float* x, const float* ca, const float* sa, float* y,
                            int B, long HW, int C, float drop_p, unsigned long long seed,
                            const unsigned long long* seed_dev, dcs_stream_t stream);

/* Several attention blocks in one set of launches.  The seven skip attentions of the network (c_network.py:208-211)
 * depend only on the encoder outputs — and their backward pass only on the decoder's — so the forward runs all of
 * them in five launches right after the encoder and the backward in five right before the encoder's backward,
 * instead of 5 (6) dependent launches of 5-10 us per block.  One item per block; no dropout on this path; 7x7
 * spatial kernel; n <= 8.  Forward reads x, w1, w2, wsa (+ its zero bias) and writes ca, pooled, hidden, sp, sa, y
 * (shapes as in the per-block entry points).  Backward additionally reads g_out, wsa_bwd and writes g_pre, g_sp, g_x and
 * the four FC weight gradients; the 7x7 conv's own weight gradient stays with dcs_cconv2d_bwd_weight(sp, g_pre). */
typedef struct {
    const float* x; const float* w1; const float* w2; const float* wsa; const float* sa_bias;
    float* ca; float* pooled; float* hidden; float* sp; float* sa; float* y;
    const float* g_out; const float* wsa_bwd; float* g_pre; float* g_sp; float* g_x;
    float* g_fc0_r; float* g_fc0_i; float* g_fc2_r; float* g_fc2_i;
    int H, W, C, Ch;
    float* g_pooled;       /* backward, optional (ABI 17): complex[B][C] — when given, the average pool's broadcast term is written
                              there and NOT added into g_x; the consumer of g_x adds g_pooled[b][c] / (H W) (dcs_cbn_bwd_add's g_add) */
} dcs_attention_item;
long dcs_attention_fwd_batched_workspace_bytes(int n, const dcs_attention_item* items, int B);
int dcs_attention_fwd_batched(int n, const dcs_attention_item* items, void* workspace, long workspace_bytes, int B,
                              dcs_stream_t stream);
long dcs_attention_bwd_batched_workspace_bytes(int n, const dcs_attention_item* items, int B);
int dcs_attention_bwd_batched(int n, const dcs_attention_item* items, void* workspace, long workspace_bytes, int B,
                              dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Real-valued attention pair of DR-Net (r_network.py:8-42; applied :155-158, :166-167) on float[B][H][W][C]:
 *   ca = sigmoid(fc(max_pool(x)))  (the reference's avg branch is overwritten, r_network.py:23-24);
 *   y = sa (.) ca (.) x,  sa = sigmoid(conv_kxk(cat(mean_c(ca x), max_c(ca x)))).
 *   w1, w2    fc.0.weight float[Ch][C], fc.2.weight float[C][Ch] (1x1 nn.Conv2d, no bias, torch layout)
 *   wsa, sa_bias  dcs_pack_conv_weight of the k x k conv taken as a 1 -> 1 COMPLEX conv with weights
 *             (w[:,0] , -w[:,1]) and no bias: its real part over the (mean, max) pair is the real 2 -> 1 conv
 *   ca_out    float[B][C];  y float[B][H][W][C];  C % 4 == 0, C/4 a power of two <= 64;  forward only. */
long dcs_rattention_workspace_bytes(int B, long HW, int C);
int  dcs_rattention_fwd(const float* x, const float* w1, const float* w2, const float* wsa, const float* sa_bias,
                        float* ca_out, float* y, void* workspace, long workspace_bytes,
                        int B, int H, int W, int C, int Ch, int ksize, dcs_stream_t stream);

/* The same block for training (r_network.py:8-42 under loss.backward(): the reference differentiates it with autograd),
 * as three pieces around the k x k conv, which keeps the complex path's entries (dcs_cconv2d_fwd / _bwd_data / _bwd_weight):
 *   pool_fwd   ca, and what its backward needs: mx float[B][C] (pooled maxima), hid float[B][Ch] (post-ReLU hidden units);
 *              pooled float2[B][H][W] = (mean_c, max_c) of ca x as ONE complex channel
 *   apply_fwd  y = x ca[c] Re(sa[p])                 sa float2[B][H][W]: the conv's complex output
 *   apply_bwd  g_x = g_y ca sa (OVERWRITES gx);  g_sa float2[B][H][W] = (sum_c g_y x ca, 0);  g_ca float[B][C] = sum_p g_y x sa
 *   pool_bwd   g_pooled float2[B][H][W] (cotangent of the (mean, max) pair), g_ca float[B][C] or NULL;
 *              gx: added to (accumulate != 0) or overwritten; gw1 float[Ch][C], gw2 float[C][Ch].
 * torch.max / AdaptiveMaxPool2d semantics: each maximum's gradient goes to its FIRST position in scan order.
 * Workspace: dcs_rattention_train_workspace_bytes for every entry that takes one. */
long dcs_rattention_train_workspace_bytes(int B, long HW, int C, int Ch);
int  dcs_rattention_pool_fwd(const float* x, const float* w1, const float* w2, float* ca_out, float* mx_out, float* hid_out,
                             float* pooled, void* workspace, long workspace_bytes, int B, int H, int W, int C, int Ch,
                             dcs_stream_t stream);
int  dcs_rattention_apply_fwd(const float* x, const float* ca, const float* sa, float* y, int B, int H, int W, int C,
                              dcs_stream_t stream);
int  dcs_rattention_apply_bwd(const float* gy, const float* x, const float* ca, const float* sa, float* gx, float* g_sa,
                              float* g_ca, void* workspace, long workspace_bytes, int B, int H, int W, int C,
                              dcs_stream_t stream);
int  dcs_rattention_pool_bwd(const float* g_pooled, const float* g_ca, const float* x, const float* ca, const float* mx,
                             const float* hid, const float* w1, const float* w2, float* gx, float* gw1, float* gw2,
                             int accumulate, void* workspace, long workspace_bytes, int B, int H, int W, int C, int Ch,
                             dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Backward of the fused attention block  out = dropout(sa (.) ca (.) x)  built from the four
 * forward entry points above (channel attention -> spatial pool -> 7x7 conv + sigmoid -> apply).
 * Step 1  dcs_attention_bwd_sa: g_pre[b][p] = sigmoid'(sa) (.) sum_c g_o conj(ca x)   (complex[B][HW]);
 *         the caller then runs the 7x7 conv's data / weight gradients on g_pre
 *         (dcs_cconv2d_bwd_data / _bwd_weight) to get g_sp complex[B][HW][2].
 * Step 2  dcs_attention_bwd_x: g_x (complex[B][HW][C]) including the paths through the spatial pool
 *         (mean and first-index arg-max over channels), through ca's global average pool and both
 *         1x1 convs; writes the gradients of fc.0 / fc.2 conv_r / conv_i in the reference's layout
 *         ([Ch][C][1][1] and [C][Ch][1][1]).  pooled / hidden / ca as saved by
 *         dcs_channel_attention_fwd; w1 / w2 the packed 1x1 weights.  g_out is the gradient of the
 *         block's output; same drop_p / seed as dcs_attention_apply_fwd.
 *         g_pooled (optional, complex[B][C]): when given, the gradient of ca's average pool is written there and
 *         NOT broadcast into g_x — the caller's next kernel adds g_pooled[b][c] / HW to every pixel of g_x
 *         (dcs_cbn_bwd_add does), which saves a read-modify-write pass over g_x. */
int  dcs_attention_bwd_sa(const float* x, const float* g_out, const float* ca, const float* sa, float* g_pre,
                          int B, long HW, int C, float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream);
long dcs_attention_bwd_workspace_bytes(int B, long HW, int C, int Ch);
int  dcs_attention_bwd_x(const float* x, const float* g_out, const float* ca, const float* sa, const float* g_sp,
                         const float* pooled, const float* hidden, const float* w1, const float* w2,
                         float* g_x, float* g_fc0_r, float* g_fc0_i, float* g_fc2_r, float* g_fc2_i, float* g_pooled,
                         void* workspace, long workspace_bytes, int B, long HW, int C, int Ch,
                         float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream);
/* dcs_attention_bwd_x with all four g_fc* null and g_pooled given leaves the FC weight gradients out (two launches instead of
 * three); dcs_attention_bwd_fc_weights computes them later from the per-sample cotangents that call left in ITS `workspace` (pass
 * the same buffer, untouched in between) — nothing downstream waits for them, so a two-stream step queues this launch beside
 * the data-gradient chain. */
int dcs_attention_bwd_fc_weights(const void* workspace, long workspace_bytes, const float* pooled, const float* hidden,
                                 float* g_fc0_r, float* g_fc0_i, float* g_fc2_r, float* g_fc2_i, int B, long HW, int C, int Ch,
                                 dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * ComplexLSTM, recurrent half (c_network.py:12-51; built :118-123, called :201).
 * One launch walks every sequence of one LSTM layer: both weight sets (real_lstm, imag_lstm),
 * both inputs (re, im) and both directions.  PyTorch LSTM semantics: gate order i,f,g,o,
 * c_t = f c_{t-1} + i g, h_t = o tanh(c_t), zero initial state, reverse direction walks t = S-1..0.
 *   gx     pre-activations of the INPUT projection, x_t W_ih^T + b_ih + b_hh, computed by the caller
 *          (one plain GEMM): element (set, n, t, dir, j) at gx[set*stride_set + n*stride_n + t*stride_t + dir*4H + j]
 *   w_hh   float[n_sets][2 dirs][4H][H]   (weight_hh_l{k}, weight_hh_l{k}_reverse)
 *   out    float[n_sets*seqs_per_set][S][2H]   ([.., dir*H + u]: forward half then reverse half)
 *   gates_save / c_save  NULL for inference; else float[NS][S][2][4H] (post-activation gates) and
 *          float[NS][S][2][H] (cell state) saved for dcs_lstm_layer_bwd.
 *   hprev_save  optional (NULL, or with gates_save): float[NS][S][2][H], the hidden state each step STARTED from
 *          (h_{t-1} forward, h_{t+1} reverse, zero at the sequence ends) = the right-hand operand of the W_hh gradient.
 * H = 64 (C_NETWORK: hparams channels[4]//2, config.py:35) or, for inference only (gates_save == NULL), 128
 * (R_NETWORK's real LSTM, r_network.py:71-75, with n_sets = 1). */
int dcs_lstm_layer_fwd(const float* gx, const float* w_hh, float* out, float* gates_save, float* c_save,
                       float* hprev_save, int n_sets, int seqs_per_set, int S, int H, long stride_set, long stride_n, long stride_t,
                       dcs_stream_t stream);

/* Backward through time of dcs_lstm_layer_fwd: from g_out float[NS][S][2H] (cotangent of `out`) and
 * the saved gates / cell states, writes the pre-activation cotangents g_pre float[NS][S][2][4H].
 * The caller's plain GEMMs turn g_pre into the gradients of x, W_ih and W_hh.  g_bias_part (optional):
 * float[NS][2][4H], g_pre summed over the time steps of each (sequence, direction); summing it over a set's
 * sequences gives that set's bias gradient. */
/* dcs_lstm_layer_fwd with the gate biases added inside the recurrence: bias_a / bias_b float[n_sets][2 dirs][4H] (b_ih and
 * b_hh, or their sum and NULL, or both NULL) — gx then holds the bare input projection (no bias-broadcast pass). */
int dcs_lstm_layer_fwd_bias(const float* gx, const float* w_hh, const float* bias_a, const float* bias_b, float* out,
                            float* gates_save, float* c_save, float* hprev_save, int n_sets, int seqs_per_set, int S,
                            int Hdim, long stride_set, long stride_n, long stride_t, dcs_stream_t stream);
int dcs_lstm_layer_bwd(const float* g_out, const float* gates, const float* c_save, const float* w_hh,
                       float* g_pre, float* g_bias_part, int n_sets, int seqs_per_set, int S, int H, dcs_stream_t stream);

/* Glue of the complex LSTM (c_network.py:33-47) that autograd otherwise runs as ~10 element-wise / reduction launches per
 * layer and step.
 * dcs_lstm_combine_fwd: out complex[n] = (o[0][i] - o[3][i]) + j (o[1][i] + o[2][i]) from the stacked recurrence outputs
 *   o float[4][n] = { L_r(x_r), L_r(x_i), L_i(x_r), L_i(x_i) }, n = B*S*2H;  _bwd: g_o = { g.re, g.im, g.im, -g.re }.
 * dcs_lstm_param_grads: g_whh float[2 sets][2 dirs][4H][H] += sum over the CK K-chunks of part float[2 dirs][2*CK][4H][H]
 *   (chunk c of set s at index s*CK + c); g_bih, g_bhh float[2][8H] += sum_n b_part float[2][seqs_per_set][8H]
 *   (dcs_lstm_layer_bwd's g_bias_part).  Fixed summation order. */
/* dcs_lstm_whh_grad: the K-chunked partial products dcs_lstm_param_grads sums, on the MFMA pipe (fp32, exact):
 *   part[d][s*CK + c][j][k] = sum over the NT/CK rows r of chunk c of g_pre[s][r][d][j] * h_prev[s][r][d][k],
 *   g_pre float[2 sets][NT][2 dirs][4H] (dcs_lstm_layer_bwd), h_prev float[2][NT][2][H] (dcs_lstm_layer_fwd), H = 64,
 *   NT/CK a multiple of 8, H a multiple of 64. */
int dcs_lstm_whh_grad(const float* g_pre, const float* h_prev, float* part, int NT, int CK, int H, dcs_stream_t stream);
/* The general form: part[(b * CK + c)][m][n] = sum over the R rows r of chunk c of A_b[r][m] * B_b[r][n], batches
 * b = hi * nlo + lo, A_b = A + lo * a_lo + hi * a_hi (floats; row pitch lda), B_b likewise; M % 32 == 0, N % 64 == 0,
 * R % 8 == 0.  dcs_chunk_sum_acc: out_b[i] += sum_c part[(b * CK + c)][i], i < MN, out_b = out + lo * o_lo + hi * o_hi
 * (the input-projection weight gradients of the LSTM: A = g_pre [NT][8H], B = the layer input [NT][in]). */
int dcs_atb_chunks(const float* A, const float* B, float* part, long a_lo, long a_hi, long b_lo, long b_hi, int nlo, int nhi,
                   int lda, int ldb, int M, int N, int R, int CK, dcs_stream_t stream);
int dcs_chunk_sum_acc(const float* part, float* out, long o_lo, long o_hi, int nlo, int nhi, int CK, long MN,
                      dcs_stream_t stream);
/* dcs_atb_chunks with an element stride b_es along the columns of B (B_b[r][n] = B_b[r * ldb + n * b_es]): b_es = 2, ldb = 2 N,
 * b_lo = 1 reads the real (lo = 0) and imaginary (lo = 1) parts of a complex-interleaved float[rows][N][2] in place — the
 * {re rows | im rows} stacking of ComplexLSTM's input (c_network.py:39-46) without its copy. */
int dcs_atb_chunks_strided(const float* A, const float* B, float* part, long a_lo, long a_hi, long b_lo, long b_hi, int nlo,
                           int nhi, int lda, int ldb, int b_es, int M, int N, int R, int CK, dcs_stream_t stream);
/* dcs_gemm_f32: the LSTM's input projections x_t W_ih^T for all time steps at once, and their data gradients (inside
 * torch.nn.LSTM in the reference: c_network.py:24-31,43-46) on the fp32 MFMA pipe (exact fp32 products, fixed summation
 * order: bit-reproducible):
 *   C_b[m][n] = sum over s < nseg, k < K of A_{b,s}[m][k] * op(B_{b,s})[k][n],  b < nbatch,
 *   A_{b,s} = A + b * a_batch + s * a_seg (floats; row pitch lda, k contiguous), B_{b,s} likewise with b_transposed != 0:
 *   B[n][k] (nn.LSTM's weight_ih layout, pitch ldb >= K) or 0: B[k][n] (pitch ldb >= N); C_b = C + b * c_batch, pitch ldc.
 *   a_planes > 0: A is instead a complex-interleaved float[a_planes][K][2] and row m is part m / a_planes (0 real, 1
 *   imaginary) of its row m % a_planes — the {re rows | im rows} stacking of ComplexLSTM's input (c_network.py:39-46) read
 *   in place; c_planes > 0: C is written the same way into float[c_planes][N][2] (the gradient of that stacking).
 *   N % 64 == 0, K % 32 == 0, any M >= 1 (M <= 2 * planes where planes are used); A, B 16-byte aligned, pitches and
 *   strides multiples of 4 floats.  DCS_ERR_BADARG otherwise. */
int dcs_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int b_transposed,
                 int nseg, long a_seg, long b_seg, int nbatch, long a_batch, long b_batch, long c_batch, int a_planes,
                 int c_planes, dcs_stream_t stream);
int dcs_lstm_combine_fwd(const float* o, float* out, long n, dcs_stream_t stream);
int dcs_lstm_combine_bwd(const float* g, float* g_o, long n, dcs_stream_t stream);
int dcs_lstm_param_grads(const float* part, const float* b_part, float* g_whh, float* g_bih, float* g_bhh, int CK,
                         int seqs_per_set, int H, dcs_stream_t stream);
/* dcs_lstm_param_grads that also sums the chunked input-projection weight gradient (dcs_atb_chunks / _strided outputs):
 * g_wih float[nsets][MN_ih] += sum over c < CK_ih of part_ih[(set * CK_ih + c)][MN_ih] — dcs_chunk_sum_acc's work in the same
 * launch (part_ih null: exactly dcs_lstm_param_grads). */
int dcs_lstm_param_grads_ih(const float* part, const float* b_part, float* g_whh, float* g_bih, float* g_bhh, int CK,
                            int seqs_per_set, int H, const float* part_ih, float* g_wih, int CK_ih, long MN_ih, int nsets,
                            dcs_stream_t stream);

/* Stand-alone inverted dropout on a real view (c_network.py:203-204 dropout_fc after the
 * ComplexLinear; c_network.py:221-222 on the last decoder stage, which has no attention to
 * fuse it into).  n floats; same mask rule as above; y may alias x; drop_p == 0 copies. */
int dcs_dropout_fwd(const float* x, float* y, long n, float drop_p, unsigned long long seed,
                    const unsigned long long* seed_dev, dcs_stream_t stream);

/* Stand-alone complexPyTorch surface used only by the layer-by-layer drop-in modules
 * (ComplexReLU config.py:103 / complex_relu, ComplexLReLU, ComplexSigmoid; complex_upsample
 * c_network.py:215).  C_NETWORK.forward fuses both into their producers / consumers instead.
 *   dcs_complex_act_fwd:      y = act(x) on n_floats floats (re and im alike); y may alias x.
 *   dcs_complex_upsample_fwd: x complex[B][H][W][C] -> y complex[B][H*up_f][W*up_t][C], nearest. */
int dcs_complex_act_fwd(const float* x, float* y, long n_floats, int act, dcs_stream_t stream);
int dcs_complex_upsample_fwd(const float* x, float* y, int B, int H, int W, int C, int up_f, int up_t,
                             dcs_stream_t stream);

/* Tap-sum: the spatial half of a convolution with ONE output channel (the last decoder stage,
 * ComplexConvTranspose2d 16 -> 1 behind cat + x2 upsample: c_network.py:135-141, :214-217).  With Cout = 1 the
 * channel contraction and the spatial gather commute, so the stage runs as a 1x1 complex conv 16 -> kh*kw "tap
 * channels" on the SOURCE-resolution tensors (dcs_cconv2d_fwd: an MFMA GEMM with full lanes) followed by
 *     y[b][oy][ox] = sum_{dy,dx} z[b][(oy - pad_f + dy)/up_f][(ox - pad_t + dx)/up_t][dy*kw + dx]
 * z: complex[B][Hs][Ws][CT] (CT >= kh*kw tap channels, extra ones ignored); y: complex[B][Hs*up_f][Ws*up_t].
 * b_r, b_i (both or neither): the layer's two real bias scalars (conv_tran_r.bias, conv_tran_i.bias, Cout = 1); the
 * complex bias (b_r - b_i) + j (b_r + b_i) is added to every output.
 * dcs_tapsum_bwd is its adjoint: gz from gy (unused tap channels get zeros); gb_r / gb_i (both or neither, one float
 * each) receive the bias gradients (S.re + S.im, S.im - S.re with S = sum of gy) — written, not accumulated; they
 * need `workspace` of dcs_tapsum_bwd_workspace_bytes(). */
int dcs_tapsum_fwd(const float* z, float* y, const float* b_r, const float* b_i, int B, int Hs, int Ws, int CT,
                   int kh, int kw, int up_f, int up_t, int pad_f, int pad_t, dcs_stream_t stream);
long dcs_tapsum_bwd_workspace_bytes(void);
int dcs_tapsum_bwd(const float* gy, float* gz, float* gb_r, float* gb_i, void* workspace, long workspace_bytes,
                   int B, int Hs, int Ws, int CT, int kh, int kw, int up_f, int up_t, int pad_f, int pad_t,
                   dcs_stream_t stream);

/* The same stage (3x3, stride 1, ONE output channel, 2x2 nearest upsample of cat(x1, x2), C1 + C2 = 16) forward in one
 * kernel: no tap-channel intermediate in HBM (conv_up1.hip).  wt: the tap-rows panel of dcs_pack_tap_rows
 * (complex[16][ct], ct >= 9, column tap = dy*3 + dx of the correlation kernel); b_r / b_i as for dcs_tapsum_fwd;
 * y complex[B][2 Hs][2 Ws].  Its backward is the factored one: dcs_tapsum_bwd, then the data / weight gradient of the
 * 1x1 tap conv. */
int dcs_cconv_up2_single_fwd(const float* x1, const float* x2, const float* wt, const float* b_r, const float* b_i,
                             float* y, int B, int Hs, int Ws, int C1, int C2, int ct, dcs_stream_t stream);

/* Its backward without the tap-channel tensor (conv_up1.hip, Round 5; C1 and C2 multiples of 4): both kernels form the nine tap sums of
 * gy (complex[B][2 Hs][2 Ws]) per source pixel in registers from an LDS tile.
 *   _bwd_data:   gx1 complex[B][Hs][Ws][C1], gx2 complex[B][Hs][Ws][C2] (written) = sum_tap conj(w[ci][tap]) T[tap]
 *   _bwd_weight: gw_r / gw_i [16][1][3][3] in the PARAMETER layout of conv_tran_r / _i.weight (added to when accumulate != 0, else
 *                written), gb_r / gb_i the two real bias gradients (written; both or neither); partial rows per workgroup in
 *                `workspace` (>= dcs_cconv_up2_single_bwd_weight_workspace_bytes()), summed in a fixed order in fp64 by a second
 *                launch — bit-reproducible, no atomics.
 * Replaces dcs_tapsum_bwd + dcs_cconv2d_bwd_data + dcs_cconv2d_bwd_weight + dcs_tap_rows_wgrad_scatter for this stage
 * (reference: the autograd backward of ComplexConvTranspose2d behind cat + upsample, c_network.py:135-141, :214-217). */
int dcs_cconv_up2_single_bwd_data(const float* gy, const float* wt, float* gx1, float* gx2, int B, int Hs, int Ws, int C1, int C2,
                                  int ct, dcs_stream_t stream);
long dcs_cconv_up2_single_bwd_weight_workspace_bytes(void);
int dcs_cconv_up2_single_bwd_weight(const float* gy, const float* x1, const float* x2, float* gw_r, float* gw_i, float* gb_r,
                                    float* gb_i, int accumulate, void* workspace, long workspace_bytes, int B, int Hs, int Ws,
                                    int C1, int C2, dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * bound_cRM (network_functions.py:77-88), as called at c_network.py:225:
 *     m = tanh|M| ; phi1 = atan2(Mi, Mr+eps) ; phi2 = atan2(m sin phi1, m cos phi1 + eps)
 *     out = m cos phi2 + j m sin phi2
 * n complex elements; out may alias in.
 */
int dcs_bound_crm_fwd(const float* M_raw, float* M_out, long n, float eps, dcs_stream_t stream);

/* The "complex subtractive mask application" of the step functions
 * (network_functions.py:240-243 train, :313-316 val, :394-397 test):
 *     M = bound_cRM(M_in) ; N_hat = Y (.) M (complex_mat_mult, element-wise) ; S_hat = Y - N_hat
 * Y, M_in: complex[n]; M_out, N_hat, S_hat: complex[n] (M_out may alias M_in).
 */
int dcs_bound_mask_apply_fwd(const float* Y, const float* M_in, float* M_out, float* N_hat, float* S_hat,
                             long n, float eps, dcs_stream_t stream);

/* Backward of dcs_bound_crm_fwd / dcs_bound_mask_apply_fwd w.r.t. the unbounded mask M_in:
 * g_M, g_N, g_S are the cotangents of M, N_hat, S_hat (each may be NULL = zero); Y may be NULL when
 * only g_M is given (plain bound_cRM backward).  g_Min may alias any cotangent.  The noisy input Y
 * is data and receives no gradient. */
int dcs_bound_mask_apply_bwd(const float* Y, const float* M_in, const float* g_M, const float* g_N,
                             const float* g_S, float* g_Min, long n, float eps, dcs_stream_t stream);

/* The network's own bound_cRM (c_network.py:225) AND the step function's second bound + multiply + subtract
 * (network_functions.py:240-243: the predicted mask is bounded twice) in one pass over the RAW last-stage output D:
 * M1 = bound(D) (stored only if M1_out != NULL), M = bound(M1), N_hat = Y (.) M, S_hat = Y - N_hat.  Backward: cotangent
 * of D from any of g_M1, g_M, g_N, g_S (NULL = absent).  drop_p > 0: D_raw is the last conv's output BEFORE the network's
 * final dropout (c_network.py:221-222); dcs_dropout_fwd's mask (same seed convention, float index 2i / 2i + 1) is applied
 * on the way in and to the cotangent on the way out, so the dropout launches of the last stage disappear as well. */
int dcs_bound2_mask_apply_fwd(const float* Y, const float* D_raw, float* M1_out, float* M_out, float* N_hat, float* S_hat,
                              long n, float eps, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                              dcs_stream_t stream);
int dcs_bound2_mask_apply_bwd(const float* Y, const float* D_raw, const float* g_M1, const float* g_M, const float* g_N,
                              const float* g_S, float* g_D, long n, float eps, float drop_p, unsigned long long seed,
                              const unsigned long long* seed_dev, dcs_stream_t stream);

/* The same pass fused with the synthesis' polar round trip (mask.hip, Round 5): straight from (Y, D_raw) complex[B][F][T] to the two
 * frame-major spectra dcs_polar_frames_fwd would make of the estimates — out complex[2B][T][Fp], rows [0, B) from Y (.) M, rows
 * [B, 2B) from Y - Y (.) M, bins F..Fp-1 zero — so that N_hat / S_hat never exist in HBM (network_functions.py:240-247).
 * M_out: optional (NULL: the mask is not stored).  _bwd: g_D from the cotangent g_out of `out` (hermitian as in
 * dcs_polar_frames_bwd) and, optionally, the cotangent g_M of the mask; everything in between is recomputed. */
int dcs_bound2_apply_polar_frames_fwd(const float* Y, const float* D_raw, float* M_out, float* out, int B, int F, int Fp, int T,
                                      float eps, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                                      dcs_stream_t stream);
int dcs_bound2_apply_polar_frames_bwd(const float* Y, const float* D_raw, const float* g_out, const float* g_M, float* g_D,
                                      int B, int F, int Fp, int T, float eps, int hermitian, float drop_p,
                                      unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream);

/* Waveform synthesis around the inverse FFT of mag_phase_2_wave / torch.istft (network_functions.py:140-150 via
 * :213-221 and :244-247).
 * dcs_polar_frames_fwd: out = |z| (cos phi + j sin phi), phi = atan2(z_i, z_r + eps), for bins f < F and zeros for
 *   the padded bins F..Fp-1, written FRAME-MAJOR: z complex[B][F][T] -> out complex[B][T][Fp], so the inverse real
 *   FFT runs over contiguous frames.  _bwd: cotangent of z from the (frame-major) cotangent of out; hermitian != 0:
 *   g_out is the plain forward real FFT of the cotangent of an unnormalised inverse real FFT's output, and the kernel
 *   applies the one-sided weighting itself (x2 for every bin but 0 and Fp-1).
 * dcs_istft_envelope: inv_env[n] = 1 / sum_f window^2 of torch.istft(center=True), n in [0, hop (T-1)).
 * dcs_istft_ola_fwd: y[b][n] = scale * inv_env[n] * sum_f window[k] frames[b][f][k], k = n + n_fft/2 - f hop:
 *   synthesis window, overlap-add, envelope division and the n_fft/2 trim in one pass.  frames float[B][T][n_fft]
 *   (irfft output), y float[B][hop (T-1)]; scale = sqrt(n_fft) for normalized=True.  No NOLA check (it is a host
 *   read-back in torch.istft): the caller guarantees a non-vanishing envelope.
 * dcs_istft_ola_bwd: g_frames from g_y (the adjoint gather). */
/* The 512-point real FFT pair of that synthesis (n_fft = 512, config.py:57), one wavefront per frame (fft512.hip):
 * dcs_irfft512_frames: X complex[frames][257] -> y float[frames][512], UNNORMALISED inverse (the caller folds 1/512 into
 *   its next scale; imaginary parts of bins 0 and 256 ignored, as by every c2r transform);
 * dcs_rfft512_frames: g float[frames][512] -> G complex[frames][257], forward, no scaling (= the adjoint of the above up to
 *   the one-sided x2 weighting that dcs_polar_frames_bwd(hermitian = 1) applies). */
int dcs_irfft512_frames(const float* X, float* y, long frames, dcs_stream_t stream);
int dcs_rfft512_frames(const float* g, float* G, long frames, dcs_stream_t stream);
/* dcs_rfft512_frames of the frames dcs_istft_ola_bwd would write, read on the fly from g_y float[B][hop (T - 1)] (Round 5: the
 * windowed cotangent frames are not stored; bit-identical G complex[B * T][257]).  hop even, n_fft = 512. */
int dcs_rfft512_ola_frames(const float* g_y, const float* window, const float* inv_env, float* G, int B, int T, int hop,
                           float scale, dcs_stream_t stream);
/* dcs_istft_ola_fwd(dcs_irfft512_frames(X)) in one kernel — the frames stay in LDS (Round 5; the same sums in the same order, bit-identical
 * y float[B][hop (T - 1)]).  X complex[B * T][257]; hop in {64, 128, 256}. */
int dcs_irfft512_ola_frames(const float* X, const float* window, const float* inv_env, float* y, int B, int T, int hop,
                            float scale, dcs_stream_t stream);
int dcs_polar_frames_fwd(const float* z, float* out, int B, int F, int Fp, int T, float eps, dcs_stream_t stream);
int dcs_polar_frames_bwd(const float* z, const float* g_out, float* g_z, int B, int F, int Fp, int T, float eps,
                         int hermitian, dcs_stream_t stream);
int dcs_istft_envelope(const float* window, float* inv_env, int T, int n_fft, int hop, dcs_stream_t stream);
int dcs_istft_ola_fwd(const float* frames, const float* window, const float* inv_env, float* y, int B, int T,
                      int n_fft, int hop, float scale, dcs_stream_t stream);
int dcs_istft_ola_bwd(const float* g_y, const float* window, const float* inv_env, float* g_frames, int B, int T,
                      int n_fft, int hop, float scale, dcs_stream_t stream);

/* SiSNR (network_functions.py:30-42) of B utterances of L samples: snr[b] = 10 log10(|a c|^2 / (|e - a c|^2 + eps) + eps),
 * a = <e,c> / (|c|^2 + eps), c = clean, e = estimate (float[B][L] each).  coef: float[B][2] scratch the backward reads.
 * dcs_sisnr_bwd: g_est[b][n] = (*g) * scale * d snr[b] / d est[b][n]; g = DEVICE scalar (the upstream gradient of the
 * batch mean), scale = 1/B for the reference's torch.mean.  The clean signal is data and receives no gradient. */
int dcs_sisnr_fwd(const float* clean, const float* est, float* snr, float* coef, int B, int L, float eps,
                  dcs_stream_t stream);
int dcs_sisnr_bwd(const float* clean, const float* est, const float* coef, const float* g, float scale, float* g_est,
                  int B, int L, dcs_stream_t stream);

/* Loss assembly of the reference's configuration (noise_loss_type 6, speech_loss_type 0: network_functions.py:168-208,
 * config.py:38-39) from the two per-utterance SiSNR vectors of dcs_sisnr_fwd:
 * out3 = { 1 - alpha * (-mean snr_noise), alpha * (-mean snr_speech), their sum } (the "1 -" quirk of :196 kept). */
int dcs_sisnr_losses_fwd(const float* snr_speech, const float* snr_noise, float* out3, int B, float alpha,
                         dcs_stream_t stream);
/* The same with the train step's NaN guard folded in (c_network.py:257-261; dcs_step_guard without its launch):
 * skip[0] = isnan(total) ? 1 : 0 when skip is not null. */
int dcs_sisnr_losses_guard_fwd(const float* snr_speech, const float* snr_noise, float* out3, int B, float alpha,
                               float* skip, dcs_stream_t stream);
/* Backward of that loss pair on STACKED signals (one launch for both estimates): target / est float[2B][L] with rows
 * [0, B) = noise and [B, 2B) = speech, coef float[2B][2] from ONE dcs_sisnr_fwd over the 2B rows; g_noise / g_speech /
 * g_total = upstream gradients of out3 (device scalars, any may be null, not all):
 *   g_est[b] = +(g_noise + g_total) alpha / B d snr_b / d est_b (b < B), -(g_speech + g_total) alpha / B ... (b >= B). */
int dcs_sisnr_pair_bwd(const float* target, const float* est, const float* coef, const float* g_noise,
                       const float* g_speech, const float* g_total, float alpha, float* g_est, int B, int L,
                       dcs_stream_t stream);

/* On-device STFT front end (data.py:104-134: noise = noisy - clean, then torch.stft(n_fft, hop, hann, center=True,
 * normalized)[1 : n_fft/2 + 1] of clean, noise, noisy on the DataLoader's CPU workers).
 * dcs_stft_frames_fwd: frames float[3][B][T][n_fft] = window[k] * x_s[b][reflect(t hop + k - n_fft/2)], s = clean,
 *   noise (= noisy - clean), noisy; clean / noisy float[B][L] cropped waveforms with (T-1) hop <= L.  The caller runs one
 *   contiguous batched real FFT over the last axis (rocFFT), giving spec complex[3][B][T][n_fft/2 + 1].
 * dcs_stft_bins_fwd: out complex[SB][F][T] = scale * spec[SB][T][f + 1], f < F = n_fft/2 (DC bin dropped, transposed to
 *   the network's layout); scale = 1/sqrt(n_fft) for normalized=True. */
int dcs_stft_frames_fwd(const float* clean, const float* noisy, const float* window, float* frames, int B, int L, int T,
                        int n_fft, int hop, dcs_stream_t stream);
int dcs_stft_bins_fwd(const float* spec, float* out, int SB, int T, int F, float scale, dcs_stream_t stream);

/* cRM target mask (network_functions.py:62-75): M = S conj(Y) / (|Y|^2 + 1e-8). */
int dcs_crm_fwd(const float* S, const float* Y, float* M, long n, float eps, dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Optimizer step of the reference's recipe, fused over one flat fp32 bucket of n parameters:
 * data-parallel averaging (grad_scale = 1/world), Trainer clip-by-global-norm
 * (gradient_clip_val 100, train.py:145-146; `grad_norm` = DEVICE scalar holding the 2-norm of the
 * UNSCALED bucket, NULL or max_norm <= 0 disables clipping) and torch.optim.Adam with L2 weight
 * decay and amsgrad (c_network.py:229-234).  step = 1-based update count, or — when step_dev != NULL —
 * read from that device int (so a captured hipGraph can be replayed while the count advances).  All
 * buffers 16-byte aligned, float[n].
 * skip (device float, may be NULL): when *skip != 0 the launch changes nothing — the device-side form of the
 * reference's NaN-loss guard (training_step returns None and the trainer skips the update, c_network.py:257-261),
 * which a captured step cannot take on the host. */
int dcs_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax,
                          const float* grad_norm, float max_norm, float grad_scale, long n,
                          float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                          const int* step_dev, const float* skip, dcs_stream_t stream);
/* The global gradient norm without ATen's two launches: dcs_grad_sumsq_parts leaves n_parts (<= 1024) fp64 partial sums of g^2
 * (fixed order: bit-reproducible) and, in the same launch, advances the step's device counters exactly as
 * dcs_step_advance_counters does (pass null / 0 to leave them alone); dcs_adam_amsgrad_step_sumsq is dcs_adam_amsgrad_step whose
 * every workgroup adds those partials up itself (||g|| = sqrt(sum)) instead of reading a norm scalar.  g 16-byte aligned, n >= 4. */
int dcs_grad_sumsq_parts(const float* g, long n, double* parts, int n_parts, const float* skip, int* step_dev, long long* seed_dev,
                         long long* counters, int n_counters, dcs_stream_t stream);
int dcs_adam_amsgrad_step_sumsq(float* p, const float* g, float* m, float* v, float* vmax, const double* sumsq_parts, int n_parts,
                                float max_norm, float grad_scale, long n, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, const int* step_dev, const float* skip, dcs_stream_t stream);

/* The NaN-loss guard of c_network.py:257-261 without a host round trip.
 * dcs_step_guard:   *skip = isnan(*loss) ? 1 : 0.  `skip` is meant to be one extra element of the flat gradient
 *                   bucket, so that a data-parallel sum-all-reduce turns it into "some rank saw a NaN" and every
 *                   rank takes the same decision.
 * dcs_step_advance: per-step device counters in one launch: *step_dev += (*skip == 0) (Adam's update count advances
 *                   only when the update runs), *seed_dev += 1 (the dropout stream advances every step, as the
 *                   reference's RNG does whether or not the update is skipped).  skip / step_dev / seed_dev may each
 *                   be NULL. */
int dcs_step_guard(const float* loss, float* skip, dcs_stream_t stream);
/* Measurement aid: park `stream` behind a one-wave kernel that polls *host_flag (a word of PINNED, device-visible host
 * memory) and ends when it is non-zero, or after timeout_ms (<= 10000) whatever happens.  A harness enqueues a step's
 * launches and event records behind it and then sets the word, so the kernels execute back to back as under hipGraph
 * replay rather than at the pace of the host's launch calls (bench.py's roofline pass). */
int dcs_stream_hold(const int* host_flag, int timeout_ms, dcs_stream_t stream);
/* Kernel timer: exact durations of the library's kernels without a profiler.  Between dcs_kernel_timer_begin(slot) and
 * dcs_kernel_timer_end() every kernel the library launches is dispatched with a start / stop event pair stamped by the
 * command processor around the dispatch itself (hipExtLaunchKernelGGL) — the quantity rocprofv3's kernel trace reports;
 * events recorded on the stream around a launch also time 6-10 us of marker and dispatch latency.  A slot spans all
 * kernels launched while it was armed (first start to last end).  dcs_kernel_timer_read: milliseconds, after the stream
 * has been synchronised.  One slot armed at a time, process-wide (state (5) of the list above); not for use under
 * stream capture.  _end returns 1 when nothing was launched. */
int dcs_kernel_timer_begin(int slot);
int dcs_kernel_timer_end(void);
int dcs_kernel_timer_read(int slot, float* ms);
int dcs_step_advance(const float* skip, int* step_dev, long long* seed_dev, dcs_stream_t stream);
/* dcs_step_advance that also adds 1 to each of n_counters int64 counters (<= 4096), unconditionally: the num_batches_tracked
 * buffers of the network's ComplexBatchNorm2d layers (complexPyTorch 0.3 counts every training forward) ride the step's own
 * counter launch instead of one of their own. */
int dcs_step_advance_counters(const float* skip, int* step_dev, long long* seed_dev, long long* counters, int n_counters,
                              dcs_stream_t stream);

/* Tap-sum factorisation of a ONE-output-channel stride-1 ComplexConvTranspose2d (the last decoder stage,
 * c_network.py:135-141): y = tapsum(conv1x1(x: Cin -> ct "tap channels")) (dcs_tapsum_fwd).
 * dcs_pack_tap_rows: the 1x1 weight from the reference's conv_tran_r / conv_tran_i.weight (float[Cin][1][kh][kw]):
 *   wp = complex[1][Cin][ct] with column `tap` = the flipped kernel at that tap and zero columns kh*kw..ct-1 (+ its MFMA
 *   panel behind it when Cin, ct are multiples of 8), bias_out = complex[ct] zeros.  Allocate
 *   dcs_packed_weight_floats(ct, Cin, 1, 1, 1, 1) floats.
 * dcs_tap_rows_wgrad_scatter: the adjoint on the weight gradient — gt_r/gt_i float[ct][Cin] (what
 *   dcs_cconv2d_bwd_weight writes for the 1x1 conv) -> gw_r/gw_i float[Cin][1][kh][kw] (added to when accumulate). */
int dcs_pack_tap_rows(const float* w_r, const float* w_i, float* wp, float* bias_out, int Cin, int kh, int kw, int ct,
                      dcs_stream_t stream);
int dcs_tap_rows_wgrad_scatter(const float* gt_r, const float* gt_i, float* gw_r, float* gw_i, int Cin, int kh, int kw,
                               int accumulate, dcs_stream_t stream);

/* Arithmetic of the MFMA convolution GEMMs (forward and data gradient).  fp32 storage and fp32 accumulation in every mode.
 *   2 (default)  fp32 EMULATED on the bf16 MFMA: each fp32 operand is split exactly into three bf16 terms
 *                (x = x0 + x1 + x2, 8 + 8 + 8 significand bits; activations on their way into LDS, weights at pack time)
 *                and the six cross products x_i w_j with i + j <= 2 are accumulated (each exact in fp32); the dropped
 *                terms are below 2^-24 |x w|.  Measured against an fp64 reference the result is closer than the native
 *                fp32 MFMA's (tests/test_hip_parity.py::test_f32_emulation_..., tools/conv_precision_check.py), at
 *                6 x 32 instead of 8 x 64 MFMA cycles per 16 k-values.  Layers with 8-channel chunks stay on mode 0.
 *   0            native v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain).
 *   1            bf16 operands (BASELINE configs[4] "bf16 mixed precision"): activations rounded to nearest-even on
 *                their way into LDS, weights rounded at pack time — a genuine bf16 computation (2^-8 operand error).
 * In mode 2 the weight gradients (conv_wgrad_mfma.hip) and the 16-column kernel are emulated the same way; the small-
 * channel kernels (enc0, the 7x7 attention convs, dec6) are native fp32 in every mode; modes 0 and 1 keep every weight
 * gradient native.  Process-wide (the
 * environment variable DCS_CONV_PRECISION presets it); the packed panel layout and size depend on the mode, so weights
 * packed under one mode are only valid under that mode (the caller re-packs after switching). */
int dcs_set_conv_precision(int mode);
int dcs_get_conv_precision(void);

/* Deferred weight-gradient reductions.  dcs_cconv2d_bwd_weight = a partial-slab kernel + a small reduce that writes the
 * parameter layout; nothing reads a weight gradient before the optimizer, so between dcs_wgrad_defer_begin() and
 * dcs_wgrad_defer_flush() the reduces are recorded instead of launched (process-wide: autograd's backward thread makes
 * the calls) and the flush runs all of them as one batched launch per 24.  While a scope is open the caller gives
 * every dcs_cconv2d_bwd_weight call its OWN workspace and keeps it, and the gradient destinations, alive until the
 * flush.  dcs_wgrad_defer_suspend(1) makes the calling thread's next reductions immediate again (for a call whose
 * result is consumed right away); (0) resumes deferral. */
int dcs_wgrad_defer_begin(void);
int dcs_wgrad_defer_suspend(int suspended);
int dcs_wgrad_defer_flush(dcs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Pack plan: every weight re-layout of a training step in one launch per dependency level.
 * No reference counterpart (cuDNN reads the nn.Parameter layout, c_network.py:107-147); the packed panels
 * dcs_pack_conv_weight / dcs_pack_conv_weight_bwd produce must be re-derived after each optimizer update
 * (~200 launches of 3-5 us).  Between dcs_pack_plan_begin() and dcs_pack_plan_end() every pack call made
 * by the process (any thread: torch.autograd runs backward on its own) still executes AND is recorded
 * with its source / destination pointers; dcs_pack_plan_run() later repeats all of them (one launch per
 * dependency level, <= 4; capturable into a hipGraph).  The caller keeps every recorded source and
 * destination buffer alive and in place for the life of the plan.  dcs_pack_plan_end / _destroy
 * allocate / free device memory: not inside a stream capture. */
int dcs_pack_plan_begin(void);
int dcs_pack_plan_end(void** plan_out);
int dcs_pack_plan_jobs(const void* plan, int* n_jobs, int* n_launches);
int dcs_pack_plan_run(const void* plan, dcs_stream_t stream);
int dcs_pack_plan_destroy(void* plan);

/* ---------------------------------------------------------------------------------------------------------------------
 * bf16 activation storage (BASELINE.json configs[4]: bf16 activations in HBM, fp32 accumulation / statistics / parameters /
 * optimizer; the reference itself trains at precision 32: config.py:70, train.py:144 — this is the build's stated
 * mixed-precision extension).  Every entry point that touches ACTIVATIONS — the float[B][F][T][C][2] tensors between layers
 * and their cotangents — exists a second time with the suffix _h and those tensors as bf16 (dcs_bf16_t = the upper 16 bits
 * of an fp32, round-to-nearest-even on store), same argument order and meaning as the fp32 form it mirrors; every other
 * operand (packed weights, biases, CBN parameters / statistics / coefficients, the per-sample and per-pixel attention maps
 * ca / sa / pooled / hidden / sp / g_pre / g_sp, split-K and weight-gradient slabs, parameter gradients) stays fp32.
 * Kernels read bf16, compute in fp32 (MFMA: bf16 operands, fp32 accumulate) and round once on store.  Requires
 * dcs_set_conv_precision(1) (bf16 weight panels).  Workspace sizes are those of the fp32 queries.
 * In dcs_attention_item the fields x, y, g_out, g_x are bf16 tensors for the _h batched calls.  dcs_cconv_up2_single_fwd_h
 * reads bf16 sources and writes the network's fp32 mask; dcs_tapsum_bwd_h reads the fp32 mask cotangent and writes bf16. */
typedef unsigned short dcs_bf16_t;
int dcs_cconv2d_fwd_h(const dcs_bf16_t* x1, const dcs_bf16_t* x2, const float* wp, const float* bias, dcs_bf16_t* y,
                      void* workspace, long workspace_bytes,
                      int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                      int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int act, dcs_stream_t stream);
int dcs_cconv2d_fwd_affine_h(const dcs_bf16_t* x1, const dcs_bf16_t* x2, const float* wp, const float* bias, const float* coef,
                             dcs_bf16_t* y, void* workspace, long workspace_bytes,
                             int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                             int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int act, dcs_stream_t stream);
int dcs_cconv2d_fwd_stats_h(const dcs_bf16_t* x1, const dcs_bf16_t* x2, const float* wp, const float* bias, dcs_bf16_t* y,
                            float* stat, int stat_rows, int* rows_used, void* workspace, long workspace_bytes,
                            int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                            int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, dcs_stream_t stream);
int dcs_cconv2d_bwd_data_h(const dcs_bf16_t* gy, const float* wp_bwd, dcs_bf16_t* gx1, dcs_bf16_t* gx2,
                           void* workspace, long workspace_bytes,
                           int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                           int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, dcs_stream_t stream);
int dcs_cconv2d_bwd_weight_h(const dcs_bf16_t* x1, const dcs_bf16_t* x2, const dcs_bf16_t* gy,
                             float* gw_r, float* gw_i, float* gb_r, float* gb_i, void* workspace, long workspace_bytes,
                             int B, int Hin, int Win, int C1, int C2, int up_f, int up_t,
                             int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int transposed,
                             dcs_stream_t stream);
int dcs_cconv_up2_single_fwd_h(const dcs_bf16_t* x1, const dcs_bf16_t* x2, const float* wt, const float* b_r, const float* b_i,
                               float* y, int B, int Hs, int Ws, int C1, int C2, int ct, dcs_stream_t stream);
int dcs_cconv_up2_single_bwd_data_h(const float* gy, const float* wt, dcs_bf16_t* gx1, dcs_bf16_t* gx2, int B, int Hs, int Ws, int C1,
                                    int C2, int ct, dcs_stream_t stream);
int dcs_cconv_up2_single_bwd_weight_h(const float* gy, const dcs_bf16_t* x1, const dcs_bf16_t* x2, float* gw_r, float* gw_i,
                                      float* gb_r, float* gb_i, int accumulate, void* workspace, long workspace_bytes, int B, int Hs,
                                      int Ws, int C1, int C2, dcs_stream_t stream);
int dcs_tapsum_bwd_h(const float* gy, dcs_bf16_t* gz, float* gb_r, float* gb_i, void* workspace, long workspace_bytes,
                     int B, int Hs, int Ws, int CT, int kh, int kw, int up_f, int up_t, int pad_f, int pad_t,
                     dcs_stream_t stream);
int dcs_cbn_fwd_h(const dcs_bf16_t* x, dcs_bf16_t* y, const float* weight, const float* bias,
                  float* running_mean, float* running_covar, float* stats_out, float* coef_out,
                  void* workspace, long workspace_bytes, long P, int C, float eps, float momentum, int use_batch_stats,
                  int act, float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream);
int dcs_cbn_fwd_slabs_h(const dcs_bf16_t* x, dcs_bf16_t* y, const float* weight, const float* bias,
                        float* running_mean, float* running_covar, float* stats_out, float* coef_out,
                        const float* part, int rows, int stride, const float* pivot,
                        long P, int C, float eps, float momentum, int act,
                        float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream);
int dcs_cbn_fwd_slabs_pool_h(const dcs_bf16_t* x, dcs_bf16_t* y, const float* weight, const float* bias,
                             float* running_mean, float* running_covar, float* stats_out, float* coef_out,
                             const float* part, int rows, int stride, const float* pivot, void* pool_part, long pool_bytes,
                             int B, long HW, int C, float eps, float momentum, int act, dcs_stream_t stream);
int dcs_cbn_bwd_h(const dcs_bf16_t* x, const dcs_bf16_t* g_out, dcs_bf16_t* g_x, const float* weight,
                  const float* stats, const float* coef, float* g_weight, float* g_bias,
                  void* workspace, long workspace_bytes, long P, int C, int use_batch_stats, int act,
                  float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream);
int dcs_cbn_bwd_add_h(const dcs_bf16_t* x, const dcs_bf16_t* g_out, dcs_bf16_t* g_x, const float* weight, const float* stats,
                      const float* coef, float* g_weight, float* g_bias, void* workspace, long workspace_bytes,
                      long P, int C, int use_batch_stats, int act, float drop_p, unsigned long long seed,
                      const unsigned long long* seed_dev, const float* g_add, float add_scale, long HW,
                      const dcs_bf16_t* g_out2, dcs_stream_t stream);
int dcs_channel_attention_fwd_h(const dcs_bf16_t* x, const float* w1, const float* w2,
                                float* ca_out, float* pooled_out, float* hidden_out,
                                void* workspace, long workspace_bytes, int B, long HW, int C, int Ch, dcs_stream_t stream);
int dcs_spatial_pool_fwd_h(const dcs_bf16_t* x, const float* ca, float* pooled, int B, long HW, int C, dcs_stream_t stream);
int dcs_attention_apply_fwd_h(const dcs_bf16_t* x, const float* ca, const float* sa, dcs_bf16_t* y,
                              int B, long HW, int C, float drop_p, unsigned long long seed,
                              const unsigned long long* seed_dev, dcs_stream_t stream);
int dcs_attention_fwd_batched_h(int n, const dcs_attention_item* items, void* workspace, long workspace_bytes, int B,
                                dcs_stream_t stream);
int dcs_attention_bwd_sa_h(const dcs_bf16_t* x, const dcs_bf16_t* g_out, const float* ca, const float* sa, float* g_pre,
                           int B, long HW, int C, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                           dcs_stream_t stream);
int dcs_attention_bwd_x_h(const dcs_bf16_t* x, const dcs_bf16_t* g_out, const float* ca, const float* sa, const float* g_sp,
                          const float* pooled, const float* hidden, const float* w1, const float* w2,
                          dcs_bf16_t* g_x, float* g_fc0_r, float* g_fc0_i, float* g_fc2_r, float* g_fc2_i, float* g_pooled,
                          void* workspace, long workspace_bytes, int B, long HW, int C, int Ch,
                          float drop_p, unsigned long long seed, const unsigned long long* seed_dev, dcs_stream_t stream);
int dcs_attention_bwd_batched_h(int n, const dcs_attention_item* items, void* workspace, long workspace_bytes, int B,
                                dcs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DCSNET_HIP_H */
