// How many small dependent kernels per second does one MI355X retire, with S streams of chains in flight?
// (the floor of a light frame: five launches that each do a few microseconds of work)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_small(unsigned* p, int work) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < work; ++i) v = v * 1664525u + 1013904223u;
    if (v == 0xdeadbeefu) p[0] = v;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096);
    for (int blocks : {64, 512}) for (int work : {16, 2000}) for (int S : {1, 2, 3, 4, 6, 8}) {
        std::vector<hipStream_t> st(S);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        const int chains = 300, per = 5;
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int c = 0; c < chains; ++c) for (int k = 0; k < per; ++k) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, st[c % S], d, work);
            hipDeviceSynchronize();
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("blocks %4d work %5d streams %d: %.2f us per kernel, %.1f us per 5-kernel chain\n", blocks, work, S, us / (chains * per), us / chains);
        }
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
