// conv_mfma_args.h — launch descriptor shared by the two implicit-GEMM complex conv kernels
// (conv_mfma.hip: one patch per workgroup; conv_pipe.hip: persistent workgroups, LDS-DMA double buffering).
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

// conv_pipe.hip.  eligible: whether a geometry planned with (cand, CH) can run on the pipelined kernel at all
// (fp32 operands, 32-column tiles, the chunk does not straddle the two tensors of a concatenation, LDS fits).
bool dcs_conv_pipe_enabled();
bool dcs_conv_pipe_eligible(const conv::Args& a, int ncls, const conv::Cls* cls, int cand, int TH, int TW, int CH);
int dcs_conv_pipe_launch(MArgs& m, int cand, int CH, hipStream_t stream);
