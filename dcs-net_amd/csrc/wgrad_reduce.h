// wgrad_reduce.h — the slab reductions that finish every weight-gradient kernel, as JOBS.
//
// Each weight-gradient kernel (conv_direct.hip, conv_wgrad_mfma.hip, conv_wgrad_small.hip) leaves per-workgroup partial
// slabs; a reduce adds them in a fixed order (no atomics: bitwise reproducible) and writes the reference's parameter
// layout.  Nothing downstream of the backward pass reads a weight gradient before the optimizer, so a training step may
// DEFER these ~28 small launches (dcs_wgrad_defer_begin) and run them as one batched launch at the end
// (dcs_wgrad_defer_flush); the slabs then have to stay alive until the flush (the caller gives each call its own
// workspace).  Outside a defer scope a job launches immediately, as before.
#pragma once
#include "dcs_common.h"

namespace wreduce {

struct Job {
    const float2* slab_w; const float2* slab_b;       // plain: [n_slabs][kh*kw][Cin][Cout], [n_slabs][Cout]
                                                      // folded: [n_slabs][ncls][kh_c*kw_c][Cin][Cout], [n_slabs*ncls][Cout]
    float* gw_r; float* gw_i; float* gb_r; float* gb_i;
    int n_slabs, Cout, Cin, kh, kw, transposed;
    int up_f, up_t;                                   // > 0: upsample-folded slabs of a 3x3 conv (conv_wgrad_mfma.hip)
    int blk0, nblk;                                   // block range inside a batched launch
};

// launch now, or record when a defer scope is open
int emit(Job j, hipStream_t s);

// true while the calling thread's weight-gradient work may be postponed to the flush
bool deferring();

}  // namespace wreduce

// conv_wgrad_small.hip: the recorded 2 -> 1 weight-gradient kernels, launched as one batch before the reduces
int dcs_conv_wgrad_small_flush(hipStream_t stream);
