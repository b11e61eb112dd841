// conv_pack.hip — derived weight panels that remove structurally wasted MFMA work.
//
// (1) Nearest upsample folded into the kernel (decoder: c_network.py:214-217 upsample x2, then a 3x3
//     stride-1 transposed conv).  Output row 2m+r only ever sees source rows {m-1, m} (r = 0) or
//     {m, m+1} (r = 1):   y[2m]   = W0 x[m-1] + (W1+W2) x[m]
//                         y[2m+1] = (W0+W1) x[m] + W2 x[m+1]
//     so each output-parity class is a 2-tap (per upsampled axis) correlation on the SOURCE tensor with
//     pre-summed weights: 4 (x2,x2) or 6 (x2,x1) taps instead of 9, and the upsampled tensor is never
//     formed, not even in LDS.
// (2) Its data gradient, directly w.r.t. the source: g_x[m] = sum_{j=0..3} K'[j] g_y[2m-1+j] with
//     K' = conj[W2, W1+W2, W0+W1, W0] — a stride-2, 4-tap correlation over g_y (16 or 12 taps per source
//     pixel instead of 4 or 2 x 9 on the upsampled grid, no block-sum pass).
// (3) Data gradient of a stride-s conv: input pixel s*m+r only receives the taps dy' = first_r + s*i of
//     the flipped kernel, reading g_y[m - pad_r + i].  One compact sub-kernel per residue class replaces
//     the zero-inserted correlation whose MFMAs are 1 - 1/(s_f s_t) zeros.
// All three are consumed by conv_mfma.hip's class launches (dcs_conv_mfma_launch_classes).
#include "conv_common.h"
#include "pack_jobs.h"

namespace {

struct AxisMap { int n; int lo[8]; int hi[8]; };      // destination tap j sums source taps lo[j]..hi[j]

// dst[(jy*nx + jx)][e'] = (conj?) sum_{dy in Y[jy]} sum_{dx in X[jx]} src[(dy*skw + dx)][e]   (packjob::FOLD)
// elements: src [A][B] complex per tap; swap -> dst [B][A] (in/out channel swap)
int fold(const float* src, float* dst, int A, int Bc, int skw, const AxisMap& Y, const AxisMap& X, int swap_conj,
         hipStream_t s) {
    packjob::Job j{};
    j.kind = packjob::FOLD;
    j.Cout = A; j.Cin = Bc; j.kw = skw; j.flag = swap_conj;
    j.yn = Y.n; j.xn = X.n;
    for (int q = 0; q < 8; ++q) {
        j.ylo[q] = (signed char)Y.lo[q]; j.yhi[q] = (signed char)Y.hi[q];
        j.xlo[q] = (signed char)X.lo[q]; j.xhi[q] = (signed char)X.hi[q];
    }
    j.total = (long)Y.n * X.n * A * Bc;
    j.dst_bytes = j.total * (long)sizeof(float2);
    j.src0 = src; j.dst0 = dst;
    return packjob::emit(j, s);
}

AxisMap identity_axis(int k) {
    AxisMap m{};
    m.n = k;
    for (int j = 0; j < k && j < 8; ++j) { m.lo[j] = j; m.hi[j] = j; }
    return m;
}

}  // namespace

namespace conv {

// ---- (1) forward fold ---------------------------------------------------------------------------
bool fold_ok(int Cin, int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t, int up_f, int up_t) {
    return mfma_ok(Cin, Cout) && kh == 3 && kw == 3 && sf == 1 && st == 1 && pad_f == 1 && pad_t == 1 &&
           (up_f == 1 || up_f == 2) && (up_t == 1 || up_t == 2) && up_f * up_t > 1;
}
static AxisMap fold_axis(int up, int r) {
    if (up == 1) return identity_axis(3);
    AxisMap m{};
    m.n = 2;
    if (r == 0) { m.lo[0] = 0; m.hi[0] = 0; m.lo[1] = 1; m.hi[1] = 2; }
    else        { m.lo[0] = 0; m.hi[0] = 1; m.lo[1] = 2; m.hi[1] = 2; }
    return m;
}
long fold_floats(int Cout, int Cin, int up_f, int up_t) {
    const int taps = (up_f == 2 ? 2 : 3) * (up_t == 2 ? 2 : 3);
    return (long)up_f * up_t * (direct_floats(Cout, Cin, taps) + mfma_floats(Cout, Cin, taps));
}
// region: for class c = ry*up_t + rx: [direct_c | mfma_c]
int pack_fold(const float* wp, float* region, int Cout, int Cin, int up_f, int up_t, hipStream_t s) {
    const int taps = (up_f == 2 ? 2 : 3) * (up_t == 2 ? 2 : 3);
    const long per = direct_floats(Cout, Cin, taps) + mfma_floats(Cout, Cin, taps);
    for (int ry = 0; ry < up_f; ++ry)
        for (int rx = 0; rx < up_t; ++rx) {
            float* d = region + (long)(ry * up_t + rx) * per;
            int rc = fold(wp, d, Cin, Cout, 3, fold_axis(up_f, ry), fold_axis(up_t, rx), 0, s);
            if (rc != DCS_OK) return rc;
            rc = dcs_conv_mfma_pack(d, d + direct_floats(Cout, Cin, taps), Cout, Cin, taps, s);
            if (rc != DCS_OK) return rc;
        }
    return DCS_OK;
}
// classes for the forward launch; bm_off relative to `region`
void fold_classes(int Cout, int Cin, int up_f, int up_t, int Hin, int Win, Cls* cls) {
    const int kh = up_f == 2 ? 2 : 3, kw = up_t == 2 ? 2 : 3, taps = kh * kw;
    const long per = direct_floats(Cout, Cin, taps) + mfma_floats(Cout, Cin, taps);
    for (int ry = 0; ry < up_f; ++ry)
        for (int rx = 0; rx < up_t; ++rx) {
            Cls& c = cls[ry * up_t + rx];
            c.kh = kh; c.kw = kw;
            c.pad_f = up_f == 2 ? (ry == 0 ? 1 : 0) : 1;
            c.pad_t = up_t == 2 ? (rx == 0 ? 1 : 0) : 1;
            c.oo_f = ry; c.oo_t = rx; c.Hc = Hin; c.Wc = Win;
            c.bm_off = (long)(ry * up_t + rx) * per + direct_floats(Cout, Cin, taps);
        }
}

// ---- (2) data gradient of the folded conv ---------------------------------------------------------
static AxisMap upfold_bwd_axis(int up) {
    AxisMap m{};
    if (up == 1) { m.n = 3; for (int j = 0; j < 3; ++j) { m.lo[j] = 2 - j; m.hi[j] = 2 - j; } return m; }
    m.n = 4;
    m.lo[0] = 2; m.hi[0] = 2; m.lo[1] = 1; m.hi[1] = 2; m.lo[2] = 0; m.hi[2] = 1; m.lo[3] = 0; m.hi[3] = 0;
    return m;
}
// in the gradient GEMM K runs over the forward Cout and N over the forward Cin
long upfold_bwd_floats(int Cout, int Cin, int up_f, int up_t) {
    const int taps = (up_f == 2 ? 4 : 3) * (up_t == 2 ? 4 : 3);
    return direct_floats(Cin, Cout, taps) + mfma_floats(Cin, Cout, taps);
}
int pack_upfold_bwd(const float* wp, float* region, int Cout, int Cin, int up_f, int up_t, hipStream_t s) {
    const int taps = (up_f == 2 ? 4 : 3) * (up_t == 2 ? 4 : 3);
    int rc = fold(wp, region, Cin, Cout, 3, upfold_bwd_axis(up_f), upfold_bwd_axis(up_t), 1, s);
    if (rc != DCS_OK) return rc;
    return dcs_conv_mfma_pack(region, region + direct_floats(Cin, Cout, taps), Cin, Cout, taps, s);
}

// ---- (3) data gradient of a strided conv -----------------------------------------------------------
Axis stride_axis(int k, int s, int pad, int r, int full) {
    const int padp = k - 1 - pad;
    Axis a;
    a.first = ((padp - r) % s + s) % s;
    a.count = a.first < k ? (k - a.first + s - 1) / s : 0;
    a.pad = -((r - padp + a.first) / s);          // exact: (r - padp + first) is a multiple of s
    a.n = (full - r + s - 1) / s;
    return a;
}
bool stride_ok(int Cin, int Cout, int kh, int kw, int sf, int st, int pad_f, int pad_t) {
    if (!mfma_ok(Cout, Cin) || sf * st <= 1 || sf > 2 || st > 2 || pad_f > kh - 1 || pad_t > kw - 1) return false;
    for (int r = 0; r < sf; ++r) if (stride_axis(kh, sf, pad_f, r, 1 << 20).count < 1) return false;
    for (int r = 0; r < st; ++r) if (stride_axis(kw, st, pad_t, r, 1 << 20).count < 1) return false;
    return true;
}
long stride_bwd_floats(int Cout, int Cin, int kh, int kw, int sf, int st, int pad_f, int pad_t) {
    long n = 0;
    for (int ry = 0; ry < sf; ++ry)
        for (int rx = 0; rx < st; ++rx) {
            const int taps = stride_axis(kh, sf, pad_f, ry, 1 << 20).count * stride_axis(kw, st, pad_t, rx, 1 << 20).count;
            n += direct_floats(Cin, Cout, taps) + mfma_floats(Cin, Cout, taps);
        }
    return n;
}
// wp_bwd: full flipped/conjugated/swapped kernel [kh*kw][Cout][Cin]; region: per class [direct_c | mfma_c]
int pack_stride_bwd(const float* wp_bwd, float* region, int Cout, int Cin, int kh, int kw, int sf, int st, int pad_f,
                    int pad_t, hipStream_t s) {
    float* d = region;
    for (int ry = 0; ry < sf; ++ry)
        for (int rx = 0; rx < st; ++rx) {
            const Axis ay = stride_axis(kh, sf, pad_f, ry, 1 << 20), ax = stride_axis(kw, st, pad_t, rx, 1 << 20);
            AxisMap Y{}, X{};
            Y.n = ay.count; X.n = ax.count;
            for (int j = 0; j < ay.count; ++j) Y.lo[j] = Y.hi[j] = ay.first + sf * j;
            for (int j = 0; j < ax.count; ++j) X.lo[j] = X.hi[j] = ax.first + st * j;
            const int taps = ay.count * ax.count;
            int rc = fold(wp_bwd, d, Cout, Cin, kw, Y, X, 0, s);
            if (rc != DCS_OK) return rc;
            rc = dcs_conv_mfma_pack(d, d + direct_floats(Cin, Cout, taps), Cin, Cout, taps, s);
            if (rc != DCS_OK) return rc;
            d += direct_floats(Cin, Cout, taps) + mfma_floats(Cin, Cout, taps);
        }
    return DCS_OK;
}
void stride_classes(int Cout, int Cin, int kh, int kw, int sf, int st, int pad_f, int pad_t, int Hv, int Wv, Cls* cls) {
    long off = 0;
    for (int ry = 0; ry < sf; ++ry)
        for (int rx = 0; rx < st; ++rx) {
            const Axis ay = stride_axis(kh, sf, pad_f, ry, Hv), ax = stride_axis(kw, st, pad_t, rx, Wv);
            const int taps = ay.count * ax.count;
            Cls& c = cls[ry * st + rx];
            c.kh = ay.count; c.kw = ax.count; c.pad_f = ay.pad; c.pad_t = ax.pad;
            c.oo_f = ry; c.oo_t = rx; c.Hc = ay.n; c.Wc = ax.n;
            c.bm_off = off + direct_floats(Cin, Cout, taps);
            off += direct_floats(Cin, Cout, taps) + mfma_floats(Cin, Cout, taps);
        }
}

}  // namespace conv
