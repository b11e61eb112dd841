// cbn_geom.h — thread/row mapping shared by the CBN forward (cbn.hip) and backward (cbn_bwd.hip)
// kernels: a 256-thread workgroup covers rows_per_iter pixels x (C/2) float4 column groups per
// pass, so a thread keeps ONE channel pair for the whole kernel (statistics and coefficients
// stay in registers) and every global access is a contiguous 16-byte-per-lane stream.
#pragma once
#ifndef DCS_CBN_RED_IT
#define DCS_CBN_RED_IT 8      // row passes per reduction workgroup
#endif
#ifndef DCS_CBN_APP_IT
#define DCS_CBN_APP_IT 4      // row passes per streaming workgroup
#endif

namespace cbn {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 512;

struct Geom {
    int vec_per_row;   // float4 groups per pixel row (C/2), or 0 for the C == 1 layout
    int rows_per_iter; // pixel rows covered by one workgroup iteration
    int nblocks;       // reduction workgroups (one partial slab each)
};

inline bool geom(long P, int C, Geom* g) {
    if (P <= 0 || C <= 0) return false;
    long it;
    if (C == 1) {
        g->vec_per_row = 0;
        g->rows_per_iter = 0;
        it = (P / 2 + kThreads - 1) / kThreads;
    } else {
        if (C & 1) return false;
        const int G = C / 2;
        if (G > kThreads || (kThreads % G) != 0) return false;
        g->vec_per_row = G;
        g->rows_per_iter = kThreads / G;
        it = (P + g->rows_per_iter - 1) / g->rows_per_iter;
    }
    const long nb = (it + DCS_CBN_RED_IT - 1) / DCS_CBN_RED_IT;
    g->nblocks = (int)(nb < 1 ? 1 : (nb > kMaxBlocks ? kMaxBlocks : nb));
    return true;
}

inline int stream_grid(long P, int C, const Geom& g) {
    const long iters = (C == 1) ? (P / 2 + kThreads - 1) / kThreads : (P + g.rows_per_iter - 1) / g.rows_per_iter;
    const long nb = (iters + DCS_CBN_APP_IT - 1) / DCS_CBN_APP_IT;
    return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
}

}  // namespace cbn

// Channel-attention pooling geometry (attention.hip: ca_pool_kernel, ca_fc_kernel; cbn.hip: the CBN apply kernel that pools
// its own output): per sample, `chunks` workgroups of 256 threads each sum their pixels' channel values into one slab
// double[C][2]; G = C / 2 lanes own a pixel.
#ifndef DCS_ATT_RED_IT
#define DCS_ATT_RED_IT 2          // row passes per reduction workgroup: each pass is loads + cross-lane reductions IN SERIES (~1 us),
                                  // so eight of them made every decoder block's kernels 16-20 us whatever the tensor size
#endif
#ifndef DCS_ATT_SMALL_IT
#define DCS_ATT_SMALL_IT 256      // row passes of a sample below which the short-chain grids are used
#endif
#ifndef DCS_ATT_MAX_CHUNKS
#define DCS_ATT_MAX_CHUNKS 256
#endif
namespace att {
inline bool geom(int C, int* G) {
    if (C < 2 || (C & 1)) return false;
    const int g = C / 2;
    if (g > 64 || (g & (g - 1)) != 0) return false;   // lane group must sit inside one wave
    *G = g;
    return true;
}
inline int ca_chunks(long HW, int G) {
    const int rows_per_iter = 256 / G;
    const long it = (HW + rows_per_iter - 1) / rows_per_iter;
    // few passes only where there are few rows to begin with (the train shapes' decoder blocks); large maps keep 8 passes and
    // 64 chunks (more chunks there cost more in slab traffic than the shorter chains return: inference 4.31 -> 4.43 ms)
    long nb = it <= DCS_ATT_SMALL_IT ? (it + DCS_ATT_RED_IT - 1) / DCS_ATT_RED_IT : (it + 7) / 8;
    if (it > DCS_ATT_SMALL_IT && nb > 64) nb = 64;
    return (int)(nb < 1 ? 1 : (nb > DCS_ATT_MAX_CHUNKS ? DCS_ATT_MAX_CHUNKS : nb));
}
}  // namespace att
