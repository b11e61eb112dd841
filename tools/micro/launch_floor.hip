// What does a dependent kernel boundary cost on this box?  rocprof of the replayed train step shows ~4.8 us per trivial kernel
// (profiles/r03_f_step_sequence_train.txt: 135 kernels at that floor); MI355X_MICROARCH.md prices a boundary at ~1.45 us.
// Chains of N dependent launches on one stream, eager and as a replayed hipGraph (stream capture), for kernels of different
// weight: 1 thread; 256 workgroups; 64 KB of dynamic LDS; 1 MB written per kernel and read by the next; a 320-byte by-value
// argument block (as MArgs).  Prints wall time per kernel (events around the whole chain, best of 5).
// build + run: hipcc -O3 --offload-arch=gfx950 tools/micro/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big { float* p; int pad[78]; };

__global__ void k_one(float* p) { if (p && threadIdx.x == 1000) p[0] = 1.f; }
__global__ __launch_bounds__(256) void k_grid(float* p) { if (threadIdx.x == 0) p[blockIdx.x] = 1.f; }
__global__ __launch_bounds__(256) void k_lds(float* p) {
    extern __shared__ float s[];
    s[threadIdx.x] = p[blockIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) p[blockIdx.x] = s[255 - threadIdx.x] + 1.f;
}
__global__ __launch_bounds__(256) void k_mb(const float4* __restrict__ a, float4* __restrict__ b) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 v = a[i]; v.x += 1.f; b[i] = v;
}
__global__ void k_big(Big a) { if (threadIdx.x == 1000) a.p[0] = (float)a.pad[77]; }

template <class F>
static void run(const char* name, int n, F launch) {
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float eager = 1e9f, graph = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < n; ++i) launch(s, i);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < eager) eager = ms;
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(s, i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 6; ++rep) {
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < graph) graph = ms;
    }
    // two replays back to back: the per-replay fixed cost drops out of the difference
    float two = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s); hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < two) two = ms;
    }
    printf("%-44s n %4d: eager %6.2f us/kernel   graph %6.2f us/kernel   (2nd replay %6.2f us/kernel)\n", name, n,
           eager * 1e3f / n, graph * 1e3f / n, (two - graph) * 1e3f / n);
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(s);
}

int main() {
    float *p, *a, *b;
    hipMalloc(&p, 1 << 20); hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20);
    hipMemset(p, 0, 1 << 20); hipMemset(a, 0, 1 << 20);
    hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int n : {50, 240}) {
        run("1 workgroup, 64 threads", n, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, s, p); });
        run("256 workgroups x 256", n, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_grid, dim3(256), dim3(256), 0, s, p); });
        run("2048 workgroups x 256", n, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_grid, dim3(2048), dim3(256), 0, s, p); });
        run("256 workgroups, 64 KB dynamic LDS", n, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 65536, s, p); });
        run("1 MB read + 1 MB written, ping-pong", n, [&](hipStream_t s, int i) {
            hipLaunchKernelGGL(k_mb, dim3(256), dim3(256), 0, s, (const float4*)(i & 1 ? b : a), (float4*)(i & 1 ? a : b)); });
        run("1 workgroup, 320-byte argument block", n, [&](hipStream_t s, int) { Big g{}; g.p = p; hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, g); });
    }
    return 0;
}
