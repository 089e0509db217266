// graph_probe -- is a captured hipGraph a cheaper way to issue the one-epoch call's launch pattern than the launches themselves?
// The pattern of gal_synth_execute for a short batch (INTEGRATION.md option B): two walker streams fork off, (A -> B) and (C -> D),
// join the caller's stream, then E -> F -> G there; kernels that run `us` microseconds each and touch nothing.  Timed per iteration,
// host clock, enqueue + wait for completion: (1) the launches with events as the library issues them, (2) the same thing captured
// once (hipStreamBeginCapture, cross-stream events) and replayed with hipGraphLaunch.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_probe tools/graph_probe.hip && /tmp/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_busy(int ticks, int *sink)  // ~ticks x 10 ns
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    if (ticks < 0) *sink = 1;
}

struct Ctx {
    hipStream_t st, ws, as;
    hipEvent_t e_prep, e_walk, e_aux;
    int *sink;
};

static void enqueue(const Ctx &c, const int *us)
{
    CK(hipEventRecord(c.e_prep, c.st));
    CK(hipStreamWaitEvent(c.ws, c.e_prep, 0));
    CK(hipStreamWaitEvent(c.as, c.e_prep, 0));
    hipLaunchKernelGGL(k_busy, dim3(16), dim3(64), 0, c.ws, us[0] * 100, c.sink);   // k_walk_carr
    hipLaunchKernelGGL(k_busy, dim3(16), dim3(256), 0, c.ws, us[1] * 100, c.sink);  // k_scanm
    CK(hipEventRecord(c.e_walk, c.ws));
    hipLaunchKernelGGL(k_busy, dim3(16), dim3(64), 0, c.as, us[2] * 100, c.sink);   // k_walk_code
    hipLaunchKernelGGL(k_busy, dim3(16), dim3(64), 0, c.as, us[3] * 100, c.sink);   // k_pages
    CK(hipEventRecord(c.e_aux, c.as));
    CK(hipStreamWaitEvent(c.st, c.e_walk, 0));
    CK(hipStreamWaitEvent(c.st, c.e_aux, 0));
    hipLaunchKernelGGL(k_busy, dim3(64), dim3(512), 0, c.st, us[4] * 100, c.sink);  // k_synth_g
    hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, c.st, us[5] * 100, c.sink);  // k_repair_g
    hipLaunchKernelGGL(k_busy, dim3(1), dim3(256), 0, c.st, us[6] * 100, c.sink);   // k_publish
}

int main()
{
    Ctx c;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&c.ws, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&c.as, hipStreamNonBlocking, hi));
    CK(hipEventCreateWithFlags(&c.e_prep, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&c.e_walk, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&c.e_aux, hipEventDisableTiming));
    CK(hipMalloc(&c.sink, 4));
    const int us_epoch[7] = {45, 17, 31, 5, 23, 4, 4};  // the one-epoch call's kernels (profiles/r05z_trace_epoch.log)
    const int us_none[7] = {1, 1, 1, 1, 1, 1, 1};       // launch cost alone
    for (const int *us : {us_none, us_epoch}) {
        const int critical = us[0] + us[1] + us[4] + us[5] + us[6];
        // (1) direct
        for (int i = 0; i < 50; ++i) { enqueue(c, us); CK(hipStreamSynchronize(c.st)); }
        const int n = 1000;
        auto t0 = std::chrono::steady_clock::now();
        double enq = 0.0;
        for (int i = 0; i < n; ++i) {
            auto a = std::chrono::steady_clock::now();
            enqueue(c, us);
            enq += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
            CK(hipStreamSynchronize(c.st));
        }
        const double direct = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        // (2) captured
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(c.st, hipStreamCaptureModeGlobal));
        enqueue(c, us);
        CK(hipStreamEndCapture(c.st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 50; ++i) { CK(hipGraphLaunch(ge, c.st)); CK(hipStreamSynchronize(c.st)); }
        t0 = std::chrono::steady_clock::now();
        double genq = 0.0;
        for (int i = 0; i < n; ++i) {
            auto a = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, c.st));
            genq += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
            CK(hipStreamSynchronize(c.st));
        }
        const double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("kernels of %2d us (critical path %3d us): direct %7.1f us per call (enqueue %5.1f) | graph %7.1f us per call (launch %5.1f)\n",
               us[0], critical, direct, enq / n, graph, genq / n);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
