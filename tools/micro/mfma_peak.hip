// Sustained fp32 MFMA rate of the device under seconds of load: the clock-limited ceiling the conv kernels are priced
// against in DESIGN.md (the datasheet's 157.3 TFLOP/s assumes 2.4 GHz; MI355X_MICROARCH.md, "DVFS give-back").
//   mode 0  bare v_mfma_f32_32x32x2_f32 chains, operands in registers (random data), one wave per SIMD, 4 accumulators
//   mode 1  the same with one ds_read_b128 per 4 MFMAs (the conv kernels' operand traffic)
//   mode 2  ONE accumulator per wave: every MFMA depends on the one before it (the <2,1,1,CH> conv kernel's chain)
//   mode 3  mode 2 with one ds_read_b128 per 4 MFMAs
// prints TFLOP/s and the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz).
// build + run: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* sink, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[(blockIdx.x * 4096 + i) & 65535];
    __syncthreads();
    float a0 = src[threadIdx.x], a1 = src[threadIdx.x + 256], b0 = src[threadIdx.x + 512], b1 = src[threadIdx.x + 768];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            const float4 v = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + it * 64) & 4092));
            a0 = v.x; a1 = v.y; b0 = v.z; b1 = v.w;
        }
        if (MODE == 3) {
            const float4 v = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + it * 64) & 4092));
            a0 = v.x; a1 = v.y; b0 = v.z; b1 = v.w;
        }
        if (MODE >= 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[0], 0, 0, 0);
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}


// ---- the bf16 pipe the emulated ("bf16x6") conv kernels run on: v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 -------------
// mode 4: 32x32x16, four accumulators per wave; mode 5: 16x16x32, four accumulators; mode 6: 32x32x16 with TWO waves per SIMD
// (the conv kernels' usual residency).  Datasheet: 2500 TFLOP/s dense at 2.4 GHz; fp32 emulated by six bf16 MFMAs: 416.7.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void kb(const float* __restrict__ src, float* sink, unsigned long long* clk, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)src[threadIdx.x + 64 * e]; b[e] = (__bf16)src[threadIdx.x + 64 * e + 512]; }
    f32x16 acc[4];
    f32x4 acc4[4];
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acc4[i][r]; }
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int grid = 256, iters = 20000;          // one 4-wave workgroup per CU: one wave per SIMD
    float *src, *sink; unsigned long long* clk;
    std::vector<float> h(65536 + 1024);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMalloc(&src, h.size() * 4); hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&sink, 4096); hipMalloc(&clk, grid * 16);
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double best = 0, last = 0; float ms = 0;
        const int launches = 20;                   // ~1 s of back-to-back launches
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < launches; ++l) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, src, sink, clk, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, src, sink, clk, iters);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, src, sink, clk, iters);
                else hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, src, sink, clk, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            last = (double)launches * grid * 4 * iters * 16 * (32.0 * 32 * 2 * 2) / (ms * 1e-3) / 1e12;
            best = std::max(best, last);
        }
        std::vector<unsigned long long> c(2 * grid);
        hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost);
        std::vector<double> ghz;
        for (int i = 0; i < grid; ++i) ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        printf("mode %d (%s): %.1f TFLOP/s sustained (last of 3 x %.1f s), best %.1f; in-kernel clock median %.3f GHz\n", mode,
               mode == 0 ? "operands in registers" : mode == 1 ? "ds_read_b128 per 4 MFMAs" : mode == 2 ? "single dependent chain" : "single chain + ds_read_b128 per 4 MFMAs", last, ms * 1e-3, best, ghz[grid / 2]);
    }
    for (int mode = 4; mode < 7; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double best = 0, last = 0; float ms = 0;
        const int launches = 20, g2 = mode == 6 ? 2 * grid : grid;
        const double flop_per_mfma = mode == 5 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < launches; ++l) {
                if (mode == 5) hipLaunchKernelGGL(kb<5>, dim3(g2), dim3(256), 0, 0, src, sink, clk, iters);
                else hipLaunchKernelGGL(kb<4>, dim3(g2), dim3(256), 0, 0, src, sink, clk, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            last = (double)launches * g2 * 4 * iters * 16 * flop_per_mfma / (ms * 1e-3) / 1e12;
            best = std::max(best, last);
        }
        std::vector<unsigned long long> c(2 * grid);
        hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost);
        std::vector<double> ghz;
        for (int i = 0; i < grid; ++i) ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        printf("mode %d (%s): %.1f TFLOP/s bf16 sustained (last of 3 x %.1f s), best %.1f = %.1f TFLOP/s of fp32 emulated by six MFMAs; in-kernel clock median %.3f GHz\n",
               mode, mode == 4 ? "v_mfma_f32_32x32x16_bf16, one wave per SIMD" : mode == 5 ? "v_mfma_f32_16x16x32_bf16, one wave per SIMD"
                                                                                          : "v_mfma_f32_32x32x16_bf16, two waves per SIMD",
               last, ms * 1e-3, best, last / 6.0, ghz[grid / 2]);
    }
    return 0;
}
