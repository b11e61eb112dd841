// graph_fork_probe.hip — when does a forked side-branch kernel of a captured hipGraph start?  (Round 4)
// A main chain of N short kernels on stream A; after every F-th one a "bulk" kernel is forked onto stream B (B's kernels also
// form a chain), one join at the end.  Every kernel stamps wall_clock64 at start and end; replayed as a graph (and launched
// eagerly for comparison) the program prints, per side kernel, how long after its fork parent's end it started.
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/graph_fork_probe.hip -o /tmp/graph_fork_probe ; run: /tmp/graph_fork_probe [order]
//   order 0: parent, fork, main-next..., side kernel captured AFTER the next main kernel;  1: side kernel captured BEFORE it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin_kernel(long long* stamps, int slot, long long ticks) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot] = t0;
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}

int main(int argc, char** argv) {
    const int order = argc > 1 ? atoi(argv[1]) : 0;
    const int N = 200, F = 10, NS = N / F;
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const double us = 1e3 / rate_khz;                       // microseconds per tick
    const long long short_t = (long long)(3.0 / us), bulk_t = (long long)(40.0 / us);
    long long* d; CK(hipMalloc(&d, sizeof(long long) * 2 * (N + NS)));
    std::vector<long long> h(2 * (N + NS));
    hipStream_t A, B; CK(hipStreamCreate(&A)); CK(hipStreamCreate(&B));
    hipEvent_t ev[NS], join; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    auto issue = [&]() {
        for (int i = 0; i < N; ++i) {
            const bool fork = (i % F) == F - 1;
            if (order == 1 && i > 0 && ((i - 1) % F) == F - 1) {}   // (side kernel already issued below)
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, A, d, i, short_t);
            if (fork) {
                CK(hipEventRecord(ev[i / F], A));
                CK(hipStreamWaitEvent(B, ev[i / F], 0));
                if (order == 1) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, B, d, N + i / F, bulk_t);
            }
            if (order == 0 && i > 0 && ((i - 1) % F) == F - 1) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, B, d, N + (i - 1) / F, bulk_t);
        }
        CK(hipEventRecord(join, B));
        CK(hipStreamWaitEvent(A, join, 0));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, A, d, N - 1, short_t);   // (re-stamps the last slot: the join's consumer)
    };
    auto report = [&](const char* what) {
        CK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        const long long t0 = h[0];
        printf("%s: main chain %.1f us; side kernels (ready -> start lag us):", what, (h[2 * (N - 2) + 1] - t0) * us);
        for (int s = 0; s < NS - 1; ++s) {
            const long long ready = h[2 * (s * F + F - 1) + 1], prev = s ? h[2 * (N + s - 1) + 1] : 0;
            const long long r = ready > prev ? ready : prev;
            printf(" %.0f", (h[2 * (N + s)] - r) * us);
        }
        printf(" | total %.1f us\n", (h[2 * (N - 1) + 1] - t0) * us);
    };
    issue(); CK(hipDeviceSynchronize()); issue(); CK(hipDeviceSynchronize());
    report("eager");
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
    issue();
    CK(hipStreamEndCapture(A, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) { CK(hipGraphLaunch(ge, A)); CK(hipStreamSynchronize(A)); }
    report("graph");
    CK(hipGraphLaunch(ge, A)); CK(hipGraphLaunch(ge, A)); CK(hipStreamSynchronize(A));
    report("graph (2nd of two back-to-back)");
    return 0;
}
