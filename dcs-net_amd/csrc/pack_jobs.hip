// pack_jobs.hip — the weight re-layout kernels (one element per thread, 32-bit index arithmetic: a 64-bit division
// chain per element was most of these kernels) behind a job table, and the pack plan.
//
// The reference has no counterpart: cuDNN consumes the nn.Parameter layout directly (c_network.py:107-147).
// Here the hot kernels read pre-packed panels (conv_direct.hip / conv_mfma.hip / conv_pack.hip), so a training
// step has to re-derive them from the updated parameters; the plan turns that from ~200 serial 4-us launches
// into one launch per dependency level.
#include "pack_jobs.h"
#include "conv_common.h"

#include <mutex>
#include <vector>

namespace packjob {
namespace {

__device__ __forceinline__ void run_direct(const Job& j, int i) {
    const float* w_r = (const float*)j.src0; const float* w_i = (const float*)j.src1;
    const float* b_r = (const float*)j.src2; const float* b_i = (const float*)j.src3;
    float2* wp = (float2*)j.dst0; float2* bias_out = (float2*)j.dst1;
    const int Cout = j.Cout, Cin = j.Cin, kh = j.kh, kw = j.kw;
    const int n = kh * kw * Cin * Cout;
    if (i < Cout) {
        const float br = b_r ? b_r[i] : 0.f, bi = b_i ? b_i[i] : 0.f;
        bias_out[i] = make_float2(br - bi, br + bi);
    }
    if (i >= n) return;
    const int co = i % Cout;
    const int ci = (i / Cout) % Cin;
    const int tap = i / (Cout * Cin);
    const int dy = tap / kw, dx = tap % kw;
    int src;
    if (j.flag)       // ConvTranspose2d weight [Cin][Cout][kh][kw], flipped
        src = ((ci * Cout + co) * kh + (kh - 1 - dy)) * kw + (kw - 1 - dx);
    else              // Conv2d weight [Cout][Cin][kh][kw]
        src = ((co * Cin + ci) * kh + dy) * kw + dx;
    wp[i] = make_float2(w_r[src], w_i[src]);
}

// wp_bwd[tap'][co][ci] = conj(wp[ntaps-1-tap'][ci][co])
__device__ __forceinline__ void run_bwd(const Job& j, int i) {
    const float2* wp = (const float2*)j.src0; float2* wpb = (float2*)j.dst0;
    const int Cout = j.Cout, Cin = j.Cin, ntaps = j.kh;
    const int ci = i % Cin;
    const int co = (i / Cin) % Cout;
    const int tp = i / (Cout * Cin);
    const float2 v = wp[((ntaps - 1 - tp) * Cin + ci) * Cout + co];
    wpb[i] = make_float2(v.x, -v.y);
}

// dst[(jy*nx + jx)][e'] = (conj?) sum_{dy in Y[jy]} sum_{dx in X[jx]} src[(dy*skw + dx)][e]
// elements: src [A][B] complex per tap; swap -> dst [B][A] (in/out channel swap)
__device__ __forceinline__ void run_fold(const Job& j, int i) {
    const float2* src = (const float2*)j.src0; float2* dst = (float2*)j.dst0;
    const int A = j.Cout, Bc = j.Cin, skw = j.kw;
    const int per = A * Bc;
    const int e = i % per;
    const int tap = i / per;
    const int jy = tap / j.xn, jx = tap % j.xn;
    int se = e;
    if (j.flag) {                          // dst element (b, a) <- src element (a, b)
        const int b = e / A, a = e % A;
        se = a * Bc + b;
    }
    float sr = 0.f, si = 0.f;
    for (int dy = j.ylo[jy]; dy <= j.yhi[jy]; ++dy)
        for (int dx = j.xlo[jx]; dx <= j.xhi[jx]; ++dx) {
            const float2 v = src[(dy * skw + dx) * per + se];
            sr += v.x; si += v.y;
        }
    dst[i] = make_float2(sr, j.flag ? -si : si);
}

// bm[tap][kg][nt][kk][j][e]: e = 0..3 -> (ci = 4kg+2kk, re), (.., im), (ci+1, re), (ci+1, im); column n = nt*32+j
// N = 16 (flag == 16): bm[tap][kg8][lane][q]: lane = g*16 + col, real k index r = 4g + q of the 8-channel block kg8
// (ci = 8 kg8 + r/2, re|im = r&1), column col = (co = col>>1, re|im = col&1)
__device__ __forceinline__ void run_mfma16(const Job& jb, int i) {
    const float2* wp = (const float2*)jb.src0; float4* bm = (float4*)jb.dst0;
    const int Cout = jb.Cout, Cin = jb.Cin;
    const int col = i & 15, g = (i >> 4) & 3;
    const int r = i >> 6;
    const int kg8 = r % (Cin / 8), tap = r / (Cin / 8);
    const int co = col >> 1, im = col & 1;
    const int ci = 8 * kg8 + 2 * g;
    const float2 w0 = wp[(tap * Cin + ci) * Cout + co], w1 = wp[(tap * Cin + ci + 1) * Cout + co];
    // k = (ci, re): [w_r | w_i];  k = (ci, im): [-w_i | w_r]  for columns (co, re) | (co, im)
    bm[i] = im ? make_float4(w0.y, w0.x, w1.y, w1.x) : make_float4(w0.x, -w0.y, w1.x, -w1.y);
}

// N = 16, fp32 emulated on the bf16 MFMA (flag == 17): bm[tap][kg16][plane][lane][8 bf16] for v_mfma_f32_16x16x32_bf16:
// lane = 16 kb + col holds the real k indices 8 kb .. 8 kb + 7 of the 16-channel block (ci = 16 kg16 + 4 kb + e/2,
// re|im = e&1) for column col = (co = col>>1, re|im = col&1); plane pl = the pl-th term of the exact bf16 split
__device__ __forceinline__ void run_mfma16_split(const Job& jb, int i) {
    // (one thread per (tap, kg16, lane): it reads its four weights once and writes all three planes)
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const float2* wp = (const float2*)jb.src0; bf16x8* bm = (bf16x8*)jb.dst0;
    const int Cout = jb.Cout, Cin = jb.Cin;
    const int col = i & 15, kb = (i >> 4) & 3;
    const int r = i >> 6;
    const int kg16 = r % (Cin / 16), tap = r / (Cin / 16);
    const int co = col >> 1, im = col & 1;
    float e[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float2 w = wp[(tap * Cin + 16 * kg16 + 4 * kb + q) * Cout + co];
        e[2 * q] = im ? w.y : w.x; e[2 * q + 1] = im ? w.x : -w.y;
    }
    const int np = jb.flag == 18 ? 1 : 3;          // flag 18: ONE plane (bf16 operands, dcs_set_conv_precision(1)): bm[tap][kg16][lane][8]
    for (int pl = 0; pl < np; ++pl) {
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) { o[k] = (__bf16)e[k]; e[k] -= (float)o[k]; }
        bm[(r * np + pl) * 64 + (i & 63)] = o;
    }
}

// bf16 (flag == 2): bm[tap][kg8][nt][lane][8 bf16]: lane = 32h + j, element e = real k index 8h + e of the 8-channel block
// (ci = 8 kg8 + 4h + e/2, re|im = e&1), column n = nt*32 + j — the A/B lane map of v_mfma_f32_32x32x16_bf16
__device__ __forceinline__ void run_mfma_bf16(const Job& jb, int i) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const float2* wp = (const float2*)jb.src0; bf16x8* bm = (bf16x8*)jb.dst0;
    const int Cout = jb.Cout, Cin = jb.Cin;
    const int KG8 = Cin / 8, NT = (2 * Cout + 31) / 32;
    // flag == 3: three such panels in a row, plane pl holding the pl-th term of the exact split w = w0 + w1 + w2
    // (w0 = bf16(w), w1 = bf16(w - w0), w2 = w - w0 - w1; ntaps = jb.kh); one thread reads its four weights once and
    // writes its element of all three planes
    const int per_plane = jb.kh * KG8 * NT * 64;
    const int j = i & 31, h = (i >> 5) & 1;
    int r = i >> 6;
    const int nt = r % NT; r /= NT;
    const int kg8 = r % KG8;
    const int tap = r / KG8;
    const int n = nt * 32 + j, co = n >> 1, im = n & 1;
    float e[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float2 w = make_float2(0.f, 0.f);
        if (co < Cout) w = wp[(tap * Cin + 8 * kg8 + 4 * h + q) * Cout + co];
        // column (co, re): [ w_r, -w_i ] ; column (co, im): [ w_i, w_r ]   for k = (ci, re), (ci, im)
        e[2 * q] = im ? w.y : w.x; e[2 * q + 1] = im ? w.x : -w.y;
    }
    const int np = jb.flag == 3 ? 3 : 1;
    for (int pl = 0; pl < np; ++pl) {
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) { o[k] = (__bf16)e[k]; e[k] -= (float)o[k]; }
        bm[pl * per_plane + i] = o;
    }
}

__device__ __forceinline__ void run_mfma(const Job& jb, int i) {
    if (jb.flag == 16) { run_mfma16(jb, i); return; }
    if (jb.flag == 17 || jb.flag == 18) { run_mfma16_split(jb, i); return; }
    if (jb.flag == 2 || jb.flag == 3) { run_mfma_bf16(jb, i); return; }
    const float2* wp = (const float2*)jb.src0; float4* bm = (float4*)jb.dst0;
    const int Cout = jb.Cout, Cin = jb.Cin;
    const int KG = Cin / 4, NT = (2 * Cout + 31) / 32;
    const int j = i & 31, kk = (i >> 5) & 1;
    int r = i >> 6;
    const int nt = r % NT; r /= NT;
    const int kg = r % KG;
    const int tap = r / KG;
    const int n = nt * 32 + j, co = n >> 1, im = n & 1;
    if (co >= Cout) { bm[i] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const int ci = 4 * kg + 2 * kk;
    const float2 w0 = wp[(tap * Cin + ci) * Cout + co], w1 = wp[(tap * Cin + ci + 1) * Cout + co];
    // column (co, re): [ w_r, -w_i ] ; column (co, im): [ w_i, w_r ]
    bm[i] = im ? make_float4(w0.y, w0.x, w1.y, w1.x) : make_float4(w0.x, -w0.y, w1.x, -w1.y);
}

// wp[0][ci][tap] = w[ci][0][kh-1-dy][kw-1-dx] (tap = dy*kw + dx), 0 for tap >= kh*kw; bias = 0.  Cout = ct columns.
__device__ __forceinline__ void run_taprows(const Job& j, int i) {
    const float* w_r = (const float*)j.src0; const float* w_i = (const float*)j.src1;
    float2* wp = (float2*)j.dst0; float2* bias_out = (float2*)j.dst1;
    const int ct = j.Cout, taps = j.kh * j.kw;
    if (i < ct) bias_out[i] = make_float2(0.f, 0.f);
    const int tap = i % ct, ci = i / ct;
    float2 v = make_float2(0.f, 0.f);
    if (tap < taps) { const int src = ci * taps + (taps - 1 - tap); v = make_float2(w_r[src], w_i[src]); }
    wp[i] = v;
}

__device__ __forceinline__ void run(const Job& j, int i) {
    if (i >= j.total) return;
    switch (j.kind) {
        case DIRECT: run_direct(j, i); break;
        case BWD: run_bwd(j, i); break;
        case FOLD: run_fold(j, i); break;
        case TAPROWS: run_taprows(j, i); break;
        default: run_mfma(j, i); break;
    }
}

__global__ __launch_bounds__(256) void pack_one_kernel(Job j) { run(j, (int)(blockIdx.x * 256 + threadIdx.x)); }

// blk0[k] = first block of job k (ascending; blk0[nj] = grid size)
__global__ __launch_bounds__(256) void pack_multi_kernel(const Job* __restrict__ jobs, const int* __restrict__ blk0, int nj) {
    // which job owns this block: thread k tests job k (one coalesced read of the table; a binary search over it in global
    // memory was six dependent round trips at the head of every block), jobs beyond 256 by the serial search
    __shared__ int s_lo;
    const int b = blockIdx.x, t = threadIdx.x;
    if (nj <= 256) {
        if (t < nj && blk0[t] <= b && b < blk0[t + 1]) s_lo = t;
        __syncthreads();
    } else {
        if (t == 0) {
            int lo = 0, hi = nj - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (blk0[mid] <= b) lo = mid; else hi = mid - 1;
            }
            s_lo = lo;
        }
        __syncthreads();
    }
    const int lo = s_lo;
    run(jobs[lo], (b - blk0[lo]) * 256 + (int)threadIdx.x);
}

// (Measured and dropped, Round 5: the whole plan in ONE launch — every job carried the chain of producers behind its source
// (DIRECT <- BWD <- FOLD <- MFMA, at most four deep) and re-derived each panel element it reads from the parameters themselves, so
// that no job read what another job of the launch wrote.  Bit-identical panels, 4 launches -> 1 — and the step 65 us SLOWER
// (3.587 vs 3.521 ms, same box, three runs each): an MFMA element of a folded data-gradient panel costs 16 parameter reads and
// four levels of runtime index divisions instead of 4 panel reads; the level form's launches are cheaper than the recomputation.)
struct Level { Job* d_jobs = nullptr; int* d_blk0 = nullptr; int nj = 0, nblocks = 0; };
struct Plan { std::vector<Level> levels; int njobs = 0; };

struct Recorder { std::vector<Job> jobs; std::vector<int> level; };
// process-wide, not thread-local: torch.autograd runs backward (and its packs) on its own worker thread
Recorder* g_rec = nullptr;
std::mutex g_rec_mutex;

bool inside(const void* p, const Job& q) {
    const char* c = (const char*)p;
    return p != nullptr && c >= (const char*)q.dst0 && c < (const char*)q.dst0 + q.dst_bytes;
}

// dependency level = 1 + the deepest recorded job whose output this one reads
void record(const Job& j) {
    int lvl = 0;
    for (size_t k = 0; k < g_rec->jobs.size(); ++k) {
        const Job& q = g_rec->jobs[k];
        if (inside(j.src0, q) || inside(j.src1, q) || inside(j.src2, q) || inside(j.src3, q))
            lvl = g_rec->level[k] + 1 > lvl ? g_rec->level[k] + 1 : lvl;
    }
    g_rec->jobs.push_back(j);
    g_rec->level.push_back(lvl);
}

}  // namespace

int emit(const Job& j, hipStream_t s) {
    if (j.total <= 0 || j.total > 0x7fffff00L || j.dst_bytes > 0x7fffffffL * 4) return DCS_ERR_BADARG;   // 32-bit element indices
    {
        std::lock_guard<std::mutex> lock(g_rec_mutex);
        if (g_rec) record(j);
    }
    DCS_LAUNCH(pack_one_kernel, dim3((unsigned)dcs_cdiv(j.total, 256)), dim3(256), 0, s, j);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

}  // namespace packjob

using packjob::Plan;
using packjob::Level;

namespace {
// g_w[ci][0][dy][dx] (+)= g_rows[kh*kw - 1 - (dy*kw + dx)][ci]: the adjoint of packjob::TAPROWS on the 1x1 weight gradient
__global__ __launch_bounds__(256) void tap_rows_scatter_kernel(const float* __restrict__ gt_r, const float* __restrict__ gt_i,
                                                                float* __restrict__ gw_r, float* __restrict__ gw_i, int Cin,
                                                                int taps, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cin * taps) return;
    const int ci = i / taps, t = i % taps;
    const long src = (long)(taps - 1 - t) * Cin + ci;
    if (accumulate) { gw_r[i] += gt_r[src]; gw_i[i] += gt_i[src]; }
    else { gw_r[i] = gt_r[src]; gw_i[i] = gt_i[src]; }
}
}  // namespace

extern "C" int dcs_pack_tap_rows(const float* w_r, const float* w_i, float* wp, float* bias_out, int Cin, int kh, int kw,
                                 int ct, dcs_stream_t stream) {
    if (!w_r || !w_i || !wp || !bias_out || Cin <= 0 || kh < 1 || kw < 1 || ct < kh * kw) return DCS_ERR_BADARG;
    packjob::Job j{};
    j.kind = packjob::TAPROWS;
    j.Cout = ct; j.Cin = Cin; j.kh = kh; j.kw = kw;
    j.total = (long)Cin * ct;
    if (j.total < ct) j.total = ct;
    j.dst_bytes = (long)Cin * ct * (long)sizeof(float2);
    j.src0 = w_r; j.src1 = w_i; j.dst0 = wp; j.dst1 = bias_out;
    const int rc = packjob::emit(j, dcs_stream(stream));
    if (rc != DCS_OK) return rc;
    if ((Cin % 8) == 0 && (ct % 8) == 0)                     // MFMA fragment panel behind the direct one (conv_mfma.hip)
        return dcs_conv_mfma_pack(wp, wp + (long)Cin * ct * 2, ct, Cin, 1, dcs_stream(stream));
    return DCS_OK;
}

extern "C" int dcs_tap_rows_wgrad_scatter(const float* gt_r, const float* gt_i, float* gw_r, float* gw_i, int Cin, int kh,
                                          int kw, int accumulate, dcs_stream_t stream) {
    if (!gt_r || !gt_i || !gw_r || !gw_i || Cin <= 0 || kh < 1 || kw < 1) return DCS_ERR_BADARG;
    const int n = Cin * kh * kw;
    DCS_LAUNCH(tap_rows_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, dcs_stream(stream), gt_r, gt_i, gw_r, gw_i,
                       Cin, kh * kw, accumulate);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

extern "C" int dcs_pack_plan_begin(void) {
    std::lock_guard<std::mutex> lock(packjob::g_rec_mutex);
    if (packjob::g_rec) return DCS_ERR_BADARG;
    packjob::g_rec = new packjob::Recorder();
    return DCS_OK;
}

extern "C" int dcs_pack_plan_end(void** plan_out) {
    packjob::Recorder* rec;
    {
        std::lock_guard<std::mutex> lock(packjob::g_rec_mutex);
        rec = packjob::g_rec;
        packjob::g_rec = nullptr;
    }
    if (!rec) return DCS_ERR_BADARG;
    if (!plan_out) { delete rec; return DCS_ERR_BADARG; }
    int nlev = 0;
    for (int l : rec->level) nlev = l + 1 > nlev ? l + 1 : nlev;
    Plan* plan = new Plan();
    plan->njobs = (int)rec->jobs.size();
    int rc = DCS_OK;
    for (int l = 0; l < nlev && rc == DCS_OK; ++l) {
        std::vector<packjob::Job> jobs;
        std::vector<int> blk0;
        long nb = 0;
        for (size_t k = 0; k < rec->jobs.size(); ++k)
            if (rec->level[k] == l) {
                jobs.push_back(rec->jobs[k]);
                blk0.push_back((int)nb);
                nb += dcs_cdiv(rec->jobs[k].total, 256);
            }
        blk0.push_back((int)nb);
        if (jobs.empty() || nb > 0x7fffffffL) { rc = DCS_ERR_BADARG; break; }
        Level lv;
        lv.nj = (int)jobs.size(); lv.nblocks = (int)nb;
        if (hipMalloc((void**)&lv.d_jobs, jobs.size() * sizeof(packjob::Job)) != hipSuccess ||
            hipMalloc((void**)&lv.d_blk0, blk0.size() * sizeof(int)) != hipSuccess ||
            hipMemcpy(lv.d_jobs, jobs.data(), jobs.size() * sizeof(packjob::Job), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(lv.d_blk0, blk0.data(), blk0.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            rc = DCS_ERR_LAUNCH;
        }
        plan->levels.push_back(lv);
    }
    delete rec;
    if (rc != DCS_OK) {
        for (Level& lv : plan->levels) { (void)hipFree(lv.d_jobs); (void)hipFree(lv.d_blk0); }
        delete plan;
        return rc;
    }
    *plan_out = plan;
    return DCS_OK;
}

extern "C" int dcs_pack_plan_jobs(const void* plan, int* n_jobs, int* n_launches) {
    if (!plan) return DCS_ERR_BADARG;
    const Plan* p = (const Plan*)plan;
    if (n_jobs) *n_jobs = p->njobs;
    if (n_launches) *n_launches = (int)p->levels.size();
    return DCS_OK;
}

extern "C" int dcs_pack_plan_run(const void* plan, dcs_stream_t stream) {
    if (!plan) return DCS_ERR_BADARG;
    const Plan* p = (const Plan*)plan;
    hipStream_t s = dcs_stream(stream);
    for (const Level& lv : p->levels) {
        DCS_LAUNCH(packjob::pack_multi_kernel, dim3((unsigned)lv.nblocks), dim3(256), 0, s, lv.d_jobs, lv.d_blk0,
                           lv.nj);
        DCS_CHECK_LAUNCH();
    }
    return DCS_OK;
}

extern "C" int dcs_pack_plan_destroy(void* plan) {
    if (!plan) return DCS_ERR_BADARG;
    Plan* p = (Plan*)plan;
    for (Level& lv : p->levels) { (void)hipFree(lv.d_jobs); (void)hipFree(lv.d_blk0); }
    delete p;
    return DCS_OK;
}
