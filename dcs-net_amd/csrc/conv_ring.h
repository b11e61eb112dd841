// conv_ring.h — plan and launch of the producer / consumer complex conv kernel (conv_ring.hip), called from conv_mfma.hip.
#pragma once
#include "conv_mfma_args.h"

struct RingPlan {
    int TH, TW;            // tile shape (TH * TW = 128 output pixels; 64 columns)
    int CH, TPS;           // complex channels per patch chunk; taps per step
    int R;                 // B ring slots (a stage = one step's panel fragments); R - 1 stages are requested ahead
    int NA;                // patch passes per producer lane and step
    int abytes;            // bytes of one patch buffer
    long lds_bytes, npix, wgs;
};

// kernel-side ring parameters
struct RingP {
    int R, NA, abytes;
};

bool DCS_SYM(dcs_conv_ring_plan)(const conv::Args& a, int ncls, const conv::Cls* cls, int pr, RingPlan* rp);
int DCS_SYM(dcs_conv_ring_launch)(MArgs& m, const RingPlan& rp, hipStream_t stream);
