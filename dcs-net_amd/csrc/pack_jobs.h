// pack_jobs.h — every weight re-layout ("pack") kernel of the library as a JOB description.
//
// A training step re-packs every convolution weight after the optimizer update: ~200 launches of 3-5 us
// kernels, each a pure index permutation (+ tap sums) of a few KB.  Describing them as data lets the
// library either launch one immediately (inference, first use) or RECORD them into a pack plan
// (dcs_pack_plan_*) that later re-runs all of them in one launch per dependency level (4 launches).
#pragma once
#include "dcs_common.h"

namespace packjob {

enum Kind : int {
    DIRECT = 0,   // reference parameter layout (w_r, w_i [, b_r, b_i]) -> complex[tap][Cin][Cout] (+ folded bias)
    BWD = 1,      // forward direct panel -> data-gradient panel complex[tap'][Cout][Cin] (flip, conj, swap)
    FOLD = 2,     // tap sums / tap subsets of a direct panel (upsample fold, stride classes), optional swap+conj
    MFMA = 3,     // direct panel -> v_mfma_f32_32x32x2_f32 B-fragment order
    TAPROWS = 4,  // ConvTranspose2d weight [Cin][1][kh][kw] -> 1x1 panel complex[1][Cin][ct]: column `tap` = the flipped kernel
                  // at that tap (the tap-sum factorisation of a Cout = 1 stage), zero columns beyond kh*kw, zero bias
};

struct Job {
    int kind;
    int Cout, Cin, kh, kw, flag;          // DIRECT: flag = transposed; FOLD: Cout = A, Cin = Bc, kw = skw, flag = swap_conj
                                          // BWD / MFMA: kh = taps
    int yn, xn;                           // FOLD: destination taps per axis
    signed char ylo[8], yhi[8], xlo[8], xhi[8];
    long total;                           // threads (elements) of the job
    long dst_bytes;                       // extent of dst0, for dependency tracking
    const void* src0; const void* src1; const void* src2; const void* src3;
    void* dst0; void* dst1;
};

// launch now on `s`; additionally recorded when a plan is being recorded on this thread
int emit(const Job& j, hipStream_t s);

}  // namespace packjob
