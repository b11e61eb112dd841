// dcs_common.h — shared helpers for the gfx950 kernels behind include/dcsnet_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/dcsnet_hip.h"

#define DCS_WAVE 64

// ---- activation storage type -----------------------------------------------------------------------------------------
// Every source that touches ACTIVATIONS (the [B,H,W,C,2] tensors between layers and their cotangents) is compiled twice
// (build.py): once with act_t = float — the reference's precision (config.py:70: precision 32), entry points as declared —
// and once with -DDCS_ACT_BF16, act_t = bf16 in HBM (BASELINE.json configs[4]: bf16 storage, fp32 accumulate / statistics /
// parameters / optimizer), entry points suffixed _h (DCS_SYM).  Kernels do their arithmetic in fp32 either way: they read
// through dcs_ld* / ActIn4 and write through dcs_st* / ActOut4, a 16-byte (fp32) or 8-byte (bf16) access per two complex
// channels.  Small per-sample / per-pixel maps (ca, sa, pooled, statistics, slabs) and every parameter stay fp32.
#ifdef DCS_ACT_BF16
typedef unsigned short act_t;                 // bf16 bits
typedef unsigned int act2_t;                  // one complex value: (re, im) bf16 pair
#define DCS_SYM(name) name##_h
#define DCS_ACT_IS_BF16 1
#else
typedef float act_t;
typedef float2 act2_t;
#define DCS_SYM(name) name
#define DCS_ACT_IS_BF16 0
#endif

#ifdef __HIPCC__
__device__ __forceinline__ float dcs_bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short dcs_f32_to_bf16(float v) {       // round to nearest even; NaN stays NaN (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(unsigned short, (__bf16)v);
}
__device__ __forceinline__ unsigned dcs_pack_bf16x2(float lo, float hi) {          // ONE v_cvt_pk_bf16_f32 (nearest even)
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{lo, hi}, bf16x2_));
}
// One step of the exact three-way bf16 split of fp32 values (the "bf16x6" emulation): returns
// {bf16(a), bf16(b)} in one dword and leaves the residuals a - bf16(a), b - bf16(b) — exact in fp32 — in a and b.  One packed
// conversion, two unpacks (shift / mask) and two subtractions per pair; element by element the compiler emitted a
// single-operand v_cvt_pk per value plus the repacking.
__device__ __forceinline__ unsigned dcs_split_pair(float& a, float& b) {
    const unsigned u = dcs_pack_bf16x2(a, b);
    a -= __uint_as_float(u << 16);
    b -= __uint_as_float(u & 0xffff0000u);
    return u;
}
// one element / one complex value / two complex values (4 consecutive elements; 16-byte resp. 8-byte aligned)
__device__ __forceinline__ float dcs_ld1(const float* p) { return *p; }
__device__ __forceinline__ float dcs_ld1(const unsigned short* p) { return dcs_bf16_to_f32(*p); }
__device__ __forceinline__ void dcs_st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void dcs_st1(unsigned short* p, float v) { *p = dcs_f32_to_bf16(v); }
__device__ __forceinline__ float2 dcs_ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 dcs_ld2(const unsigned short* p) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
    return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
}
__device__ __forceinline__ void dcs_st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void dcs_st2(unsigned short* p, float2 v) { *reinterpret_cast<unsigned*>(p) = dcs_pack_bf16x2(v.x, v.y); }
__device__ __forceinline__ float4 dcs_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 dcs_ld4(const unsigned short* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void dcs_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void dcs_st4(unsigned short* p, float4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(dcs_pack_bf16x2(v.x, v.y), dcs_pack_bf16x2(v.z, v.w));
}
// views that index in units of FOUR elements (two complex channels), so kernels written against float4 pointers keep
// their index arithmetic: `const ActIn4<T> x4 = act_in4(x) + base;  v = x4[i];  y4[i] = o;`
template <typename T> struct ActIn4 {
    const T* p;
    __device__ __forceinline__ float4 operator[](long i) const { return dcs_ld4(p + 4 * i); }
    __device__ __forceinline__ ActIn4 operator+(long off) const { return ActIn4{p + 4 * off}; }
};
template <typename T> struct ActOut4 {
    T* p;
    struct Ref {
        T* q;
        __device__ __forceinline__ void operator=(float4 v) const { dcs_st4(q, v); }
    };
    __device__ __forceinline__ Ref operator[](long i) const { return Ref{p + 4 * i}; }
    __device__ __forceinline__ ActOut4 operator+(long off) const { return ActOut4{p + 4 * off}; }
};
template <typename T> __device__ __forceinline__ ActIn4<T> act_in4(const T* p) { return ActIn4<T>{p}; }
template <typename T> __device__ __forceinline__ ActOut4<T> act_out4(T* p) { return ActOut4<T>{p}; }
#endif

// ---- kernel timer (dcs_kernel_timer_*, api.hip) ----------------------------------------------------------------------
// While a slot is armed every launch of the library goes through hipExtLaunchKernelGGL with a start / stop event pair:
// the command processor stamps the dispatch itself (what rocprofv3's kernel trace reports), not the stream around it —
// an event pair recorded around a launch also times ~6-10 us of marker and dispatch latency.  The first launch after
// arming donates the start stamp, every launch overwrites the stop stamp: a slot spans all the kernels of one C-ABI call.
struct DcsKernelTimer { hipEvent_t start, stop; bool armed, first; };
extern DcsKernelTimer g_dcs_ktimer;
#define DCS_LAUNCH(kern, grid, block, lds, stream, ...)                                                               \
    do {                                                                                                              \
        if (g_dcs_ktimer.armed) {                                                                                     \
            hipEvent_t s__ = g_dcs_ktimer.first ? g_dcs_ktimer.start : nullptr;                                      \
            g_dcs_ktimer.first = false;                                                                               \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, s__, g_dcs_ktimer.stop, 0, __VA_ARGS__);            \
        } else {                                                                                                      \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                          \
        }                                                                                                             \
    } while (0)

#define DCS_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return DCS_ERR_LAUNCH;        \
    } while (0)

// hipGetLastError() is per-thread and sticky across unrelated HIP calls made by the host
// framework; clear it before our launches so DCS_CHECK_LAUNCH reports only our own failures.
static inline hipStream_t dcs_stream(dcs_stream_t s) {
    (void)hipGetLastError();
    return reinterpret_cast<hipStream_t>(s);
}

// {v, v} as a register pair of its own.  A packed fp32 operation whose LOW lane reads the HIGH dword of a source pair
// (op_sel) can lose that lane's product beside co-resident bf16-MFMA waves on gfx950 (profiles/r03_pk_fma_op_sel_hazard.txt):
// hand-written packed complex MACs broadcast a high half through this instead (two v_mov), or keep the value in the LOW half
// of a pair (layouts {x, y, y, x}); tests/test_host_cpu.py scans EVERY kernel of the library for the selecting form.
typedef float dcs_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dcs_v2f dcs_bcast2(float v) {
    dcs_v2f r = {v, v};
    asm volatile("" : "+v"(r));
    return r;
}

__device__ __forceinline__ float dcs_act(float v, int act) {
    if (act == DCS_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DCS_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
    if (act == DCS_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

// wave64 butterfly sum (every lane ends with the total)
__device__ __forceinline__ float dcs_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// wave64 sum through DPP (no LDS crossbar traffic): 4 row shifts + 2 row broadcasts; the total is valid in
// LANE 63 ONLY.  ~6 VALU instructions per value instead of 6 ds_bpermute round trips.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dcs_dpp_term(float v) {
    // lanes without a valid source (shifted in from outside the row, or masked rows) contribute 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// All-reduce over the G lanes that share a pixel (G a power of two <= 64, groups aligned to G): the steps inside a row of 16
// lanes are DPP moves — quad_perm [1,0,3,2] and [2,3,0,1] (xor 1, xor 2), row_half_mirror (lane i <-> 7 - i: the other quad of an
// 8-lane group, which already holds its own sum), row_mirror (i <-> 15 - i) — one VALU instruction each; only the 32- and 64-lane
// steps go through the LDS crossbar.  As __shfl_xor (ds_bpermute + s_waitcnt) every step was an LDS operation: the per-pixel
// channel reductions of the attention kernels ran at half their streaming rate on the 8- and 16-channel maps.
template <int CTRL>
__device__ __forceinline__ float dcs_dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dcs_dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
#define DCS_GROUP_REDUCE(NAME, T, MOV, OP)                                                   \
    __device__ __forceinline__ T NAME(T v, int G) {                                         \
        if (G >= 2) { const T o = MOV<0xB1>(v); v = OP(v, o); }                             \
        if (G >= 4) { const T o = MOV<0x4E>(v); v = OP(v, o); }                             \
        if (G >= 8) { const T o = MOV<0x141>(v); v = OP(v, o); }                            \
        if (G >= 16) { const T o = MOV<0x140>(v); v = OP(v, o); }                           \
        if (G >= 32) { const T o = __shfl_xor(v, 16, 64); v = OP(v, o); }                   \
        if (G >= 64) { const T o = __shfl_xor(v, 32, 64); v = OP(v, o); }                   \
        return v;                                                                           \
    }
__device__ __forceinline__ float dcs_op_add(float a, float b) { return a + b; }
__device__ __forceinline__ float dcs_op_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ int dcs_op_min_i(int a, int b) { return a < b ? a : b; }
DCS_GROUP_REDUCE(dcs_group_sum, float, dcs_dpp_mov, dcs_op_add)
DCS_GROUP_REDUCE(dcs_group_max, float, dcs_dpp_mov, dcs_op_max)
DCS_GROUP_REDUCE(dcs_group_min_i, int, dcs_dpp_mov_i, dcs_op_min_i)
#undef DCS_GROUP_REDUCE

__device__ __forceinline__ float dcs_wave_sum_lane63(float v) {
    v += dcs_dpp_term<0x111, 0xf>(v);        // row_shr:1
    v += dcs_dpp_term<0x112, 0xf>(v);        // row_shr:2
    v += dcs_dpp_term<0x114, 0xf>(v);        // row_shr:4
    v += dcs_dpp_term<0x118, 0xf>(v);        // row_shr:8   -> lane 15 of each row holds the row sum
    v += dcs_dpp_term<0x142, 0xa>(v);        // row_bcast:15 into rows 1 and 3
    v += dcs_dpp_term<0x143, 0xc>(v);        // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ double dcs_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float dcs_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based dropout RNG: one 32-bit hash per real element (re and im are
// dropped independently, c_network.py:195-196).  Regenerated in backward from
// (seed, element index); never stored.
__device__ __forceinline__ uint32_t dcs_hash32(uint64_t seed, uint64_t idx) {
    uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float dcs_keep_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
    // uniform in [0,1): drop when u < p
    float u = (float)(dcs_hash32(seed, idx) >> 8) * (1.0f / 16777216.0f);
    return u < p ? 0.f : inv_keep;
}

static inline int dcs_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Plan / tuning thresholds (tile-plan cut-offs, LDS caps, slab planning, forced kernel forms): compile-time constants in the shipped
// library — a configuration nobody tests does not ride an environment variable.  A diagnostic build
// (tools/exp_build.py <name> <source> -DDCS_PLAN_KNOBS, run with DCS_LIB_PATH) reads them from the environment again: that is
// what the sweep tools (tools/plan_sweep.sh, tools/bf16_sweep.sh, tools/grad_noise_variants.py) use.  The switches that stay
// environment variables in the shipped library are the documented ones of INTEGRATION.md §6, each covered by
// tests/test_switches.py.
#include <cstdlib>
static inline long dcs_knob(const char* name, long dflt) {
#ifdef DCS_PLAN_KNOBS
    const char* e = getenv(name);
    return e ? atol(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// Raise a kernel's dynamic-LDS limit (> 64 KiB) once per (kernel, size high-water mark).  The table makes the
// call idempotent, so steady-state launches issue no attribute call at all — hipFuncSetAttribute is not
// permitted while a stream is being captured into a hipGraph, and the eager warm-up steps have already made it.
hipError_t dcs_ensure_dynamic_lds(const void* fn, size_t bytes);

// 0: native fp32 MFMA, 1: bf16 operands, 2 (default): fp32 emulated by six bf16 MFMAs on exact three-way operand splits, in the
// MFMA conv forward / data-gradient GEMMs (dcs_set_conv_precision)
int dcs_conv_precision();

// Wave priority of kernels on the train step's CRITICAL chain (Round 4).  The weight-gradient kernels run on a side stream
// beside the data-gradient chain of the backward pass (dcsnet/functional.py, _CConv2dFn.backward); they feed nothing before the
// optimizer, so they stay at the default priority 0 while every main-stream kernel that can be co-resident with them raises
// its own: the instruction arbiter of a SIMD then serves the chain's waves first and the weight gradient fills the issue
// slots and MFMA cycles the chain leaves idle, instead of sharing them evenly (age-based arbitration favours the older —
// i.e. the weight-gradient — waves, and slowed the launch-bound kernels of the chain 2-3x: profiles/r04_side_stream_ab.txt).
#ifndef DCS_CRITICAL_PRIO_LEVEL
#define DCS_CRITICAL_PRIO_LEVEL 3
#endif
#if DCS_CRITICAL_PRIO_LEVEL > 0
#define DCS_PRIO_CRITICAL() __builtin_amdgcn_s_setprio(DCS_CRITICAL_PRIO_LEVEL)
#else
#define DCS_PRIO_CRITICAL() ((void)0)
#endif
