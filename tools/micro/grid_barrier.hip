// Cost of a grid-wide barrier inside one kernel (device-scope atomic ticket + spin) against the ~4.8 us a kernel boundary costs
// in a replayed hipGraph: is a multi-phase cooperative kernel (statistics -> finalize -> apply ...) cheaper than its launches?
// 256 workgroups x 256 threads (one per CU, always co-resident); every spin has an iteration cap, so the kernel ends even if a
// workgroup never arrives.  Also the variant where only the data of the phase boundary is released (threadfence + relaxed atomics).
// build + run: hipcc -O3 --offload-arch=gfx950 tools/micro/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(unsigned* cnt, float* data, int phases, unsigned* timeouts) {
    const unsigned n = gridDim.x;
    float v = (float)threadIdx.x;
    for (int p = 0; p < phases; ++p) {
        data[(size_t)blockIdx.x * 256 + threadIdx.x] = v;               // the phase's output ...
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();                                            // ... released to the device
            atomicAdd(cnt, 1u);
            const unsigned target = (unsigned)(p + 1) * n;
            int spins = 0;
            while (__atomic_load_n(cnt, __ATOMIC_RELAXED) < target && ++spins < 2000000) __builtin_amdgcn_s_sleep(1);
            if (spins >= 2000000) atomicAdd(timeouts, 1u);
            __threadfence();
        }
        __syncthreads();
        v += data[(size_t)((blockIdx.x + 1) % n) * 256 + threadIdx.x];   // ... and read by another workgroup
    }
    if (v == 123.456f) data[0] = v;
}

int main() {
    unsigned *cnt, *to; float* data;
    hipMalloc(&cnt, 4); hipMalloc(&to, 4); hipMalloc(&data, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {64, 256, 512, 1024}) {
        for (int phases : {1, 9, 33}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(cnt, 0, 4); hipMemset(to, 0, 4);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, cnt, data, phases, to);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            unsigned h = 0; hipMemcpy(&h, to, 4, hipMemcpyDeviceToHost);
            printf("grid %4d phases %2d: %.1f us (timeouts %u)\n", grid, phases, best * 1e3f, h);
        }
    }
    return 0;
}
