// Microbenchmark: cycles per LDS-DMA piece (global_load_lds_dwordx4, 1 KiB per wave instruction) issued by one loader wave,
// alone or beside four waves that run an fp32 MFMA + ds_read_b128 loop (the conv_pipe.hip situation).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/dma_rate.hip -o /tmp/dma_rate ; run: /tmp/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE>   // 0: loader alone; 1: loader + 4 MFMA waves; 2: as 1 with loader at s_setprio 3
__global__ __launch_bounds__(320) void k(const float* __restrict__ src, long long* out, float* sink, int npieces, int reps, int stride_f) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 4) {
        if (MODE == 2) __builtin_amdgcn_s_setprio(3);
        const float* base = src + (long)blockIdx.x * npieces * 256 + (long)lane * stride_f;
        long long t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; ++r) {
            for (int p = 0; p < npieces; ++p)
                __builtin_amdgcn_global_load_lds(GPTR(base + (long)p * 256), LPTR(lds + (p & 31) * 256), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) out[blockIdx.x] = t1 - t0;
        return;
    }
    if (MODE == 0) return;
    f32x16 acc = {0};
    const float4* a = reinterpret_cast<const float4*>(lds + 8192) + threadIdx.x;
    float4 bv = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int it = 0; it < reps * npieces; ++it) {
        float4 av = a[(it & 7) * 256];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    if (acc[0] == 12345.f) sink[threadIdx.x] = acc[1];
}

int main() {
    const int grid = 256, npieces = 24, reps = 50;
    float* src; long long* out; float* sink;
    hipMalloc(&src, (size_t)grid * npieces * 1024 + (1 << 20));
    hipMemset(src, 0, (size_t)grid * npieces * 1024 + (1 << 20));
    hipMalloc(&out, grid * 8); hipMalloc(&sink, 4096);
    std::vector<long long> h(grid);
    for (int stride : {4, 36}) for (int mode = 0; mode < 3; ++mode) {
        for (int w = 0; w < 2; ++w) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(320), 96 * 1024, 0, src, out, sink, npieces, reps, stride);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(320), 96 * 1024, 0, src, out, sink, npieces, reps, stride);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(320), 96 * 1024, 0, src, out, sink, npieces, reps, stride);
        }
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        printf("lane stride %3d floats, mode %d: %.0f cycles per piece (avg over %d WGs)\n", stride, mode, s / grid / (npieces * reps), grid);
    }
    return 0;
}
