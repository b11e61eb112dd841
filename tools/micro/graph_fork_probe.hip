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

// Two LINEAR graphs instead of one forked graph: the main chain signals through a device flag (a one-thread kernel behind the fork
// parent), the side chain waits on it in a one-wave kernel with a time-out.  Both graphs count their own replays (epoch), so the
// flag values only ever grow and nothing has to be reset.
__global__ void epoch_kernel(int* epoch) { epoch[0] += 1; }
__global__ void signal_kernel(int* flag, const int* epoch, int s) {
    __hip_atomic_store(flag, epoch[0] * 1024 + s + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void wait_kernel(const int* flag, const int* epoch, int s, long long max_ticks, int* timeouts) {
    const int want = epoch[0] * 1024 + s + 1;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (wall_clock64() - t0 > max_ticks) { if (threadIdx.x == 0) atomicAdd(timeouts, 1); break; }
        __builtin_amdgcn_s_sleep(8);
    }
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
    // ---- the main chain alone as a linear graph (what a node of an unforked graph costs) ----
    {
        hipGraph_t g1; hipGraphExec_t ge1;
        CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, A, d, i, short_t);
        CK(hipStreamEndCapture(A, &g1));
        CK(hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) { CK(hipGraphLaunch(ge1, A)); CK(hipStreamSynchronize(A)); }
        CK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        printf("linear graph, main chain only: %.1f us (%.2f us per node beyond its 3 us)\n", (h[2 * (N - 1) + 1] - h[0]) * us,
               ((h[2 * (N - 1) + 1] - h[0]) * us - 3.0 * N) / N);
    }
    // ---- two linear graphs, flag-synchronised ----
    {
        int* sync; CK(hipMalloc(&sync, 4 * sizeof(int))); CK(hipMemset(sync, 0, 4 * sizeof(int)));   // flag, epoch A, epoch B, time-outs
        const long long max_t = (long long)(20000.0 / us);                                          // 20 ms
        hipGraph_t gA, gB; hipGraphExec_t geA, geB;
        CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(epoch_kernel, dim3(1), dim3(1), 0, A, sync + 1);
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, A, d, i, short_t);
            if ((i % F) == F - 1) hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, A, sync, sync + 1, i / F);
        }
        CK(hipStreamEndCapture(A, &gA));
        CK(hipGraphInstantiate(&geA, gA, nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(B, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(epoch_kernel, dim3(1), dim3(1), 0, B, sync + 2);
        for (int s_ = 0; s_ < NS; ++s_) {
            hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, B, sync, sync + 2, s_, max_t, sync + 3);
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, B, d, N + s_, bulk_t);
        }
        CK(hipStreamEndCapture(B, &gB));
        CK(hipGraphInstantiate(&geB, gB, nullptr, nullptr, 0));
        for (int r = 0; r < 4; ++r) {
            CK(hipGraphLaunch(geA, A)); CK(hipGraphLaunch(geB, B));
            CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        }
        CK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        int hs[4]; CK(hipMemcpy(hs, sync, sizeof(hs), hipMemcpyDeviceToHost));
        const long long t0 = h[0];
        long long last = h[2 * (N - 1) + 1];
        for (int s_ = 0; s_ < NS; ++s_) last = h[2 * (N + s_) + 1] > last ? h[2 * (N + s_) + 1] : last;
        printf("two linear graphs + flags: main chain %.1f us (%.2f us per node beyond its 3 us, signals included); side kernels (ready -> start lag us):",
               (h[2 * (N - 1) + 1] - t0) * us, ((h[2 * (N - 1) + 1] - t0) * us - 3.0 * N) / N);
        for (int s_ = 0; s_ < NS; ++s_) {
            const long long ready = h[2 * (s_ * F + F - 1) + 1], prev = s_ ? h[2 * (N + s_ - 1) + 1] : 0;
            printf(" %.0f", (h[2 * (N + s_)] - (ready > prev ? ready : prev)) * us);
        }
        printf(" | total %.1f us, time-outs %d\n", (last - t0) * us, hs[3]);
    }
    return 0;
}
