// conv_mfma_args.h — launch descriptor of the implicit-GEMM complex conv kernels (conv_mfma.hip).
#pragma once
#include "conv_common.h"

struct MArgs {
    conv::Args c;              // c.Hout / c.Wout: FULL output extent (addressing); c.sf / c.st: stride in class space
    const float* bm;
    act_t* y2;                 // optional second output: columns >= nsplit go here (g_x1 | g_x2 of a cat)
    int nsplit;
    int twshift;               // log2(TW): tile widths are powers of two
    int TH, TW, N, KG, NT;     // tile shape (TH*TW = pixels per WG), N = 2*Cout, KG = Cin/4, NT = ceil(N/32)
    int ncls, os_f, os_t;      // output-parity classes (blockIdx.z): pixel (oy, ox) of class c is stored at
    conv::Cls cls[4];          //   (oy*os_f + oo_f, ox*os_t + oo_t) and has its own sub-kernel / padding / panel
    // split-K (layers whose pixel x column tiles alone leave most CUs idle): blockIdx.y also indexes ksplit slices of
    // the input-channel chunks; each slice stores its raw partial tile into part[slice][pixel][N] and
    // splitk_reduce_kernel adds the slices (+ bias, activation, cat split) in a fixed order
    int ksplit, cps;           // slices, chunks per slice
    float* part;
    long slab_floats;          // B * Hout * Wout * N
    long long* dbg;            // diagnostic builds only (-DDCS_FWD_DIAG): per-workgroup phase times (core clocks)
};
