// Does a HIP graph shorten a chain of DEPENDENT small kernels on this stack?  The frame of sgs_render is such a chain: memset, 9 kernels
// of 5-130 us, one copy.  build: hipcc --offload-arch=gfx950 -O2 scripts/graph_floor.hip -o build/graph_floor ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { int v[960]; };                     // a 3.8-KiB by-value argument, like FrameGroup
__global__ void k_small(Big b, int* p, int spin) {
    int x = b.v[threadIdx.x & 511];
    for (int i = 0; i < spin; ++i) x = x * 1664525 + 1013904223;
    if (x == 42) p[0] = x;
}
int main() {
    int* d; CK(hipMalloc(&d, 4096));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Big b = {};
    const int N = 9, REP = 200;
    for (int spin : {200, 2000, 20000}) {
        auto chain = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, b, d, spin); };
        for (int i = 0; i < 20; ++i) chain();
        CK(hipStreamSynchronize(s));
        // (a) one chain at a time, host-timed (what a synchronous frame sees)
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r) { chain(); CK(hipStreamSynchronize(s)); }
        double a_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
        // (b) the same chain as an instantiated graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        double b_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
        // (c) graph + per-launch parameter update of every node (a new camera per frame)
        std::vector<hipGraphNode_t> nodes(N); size_t nn = N; CK(hipGraphGetNodes(g, nodes.data(), &nn));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r) {
            b.v[0] = r;
            for (size_t i = 0; i < nn; ++i) {
                hipKernelNodeParams kp; CK(hipGraphKernelNodeGetParams(nodes[i], &kp));
                void* args[3] = {&b, &d, &spin}; kp.kernelParams = args;
                CK(hipGraphExecKernelNodeSetParams(ge, nodes[i], &kp));
            }
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        }
        double c_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
        // (d) kernel time alone: one launch of N x the work
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s)); for (int r = 0; r < REP; ++r) chain(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("spin %6d: chain of %d dependent launches, one chain at a time: stream %.1f us, graph %.1f us, graph + SetParams x%d %.1f us; back-to-back chains (no host wait) %.1f us per chain\n",
               spin, N, a_us, b_us, N, c_us, 1e3 * ms / REP);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
