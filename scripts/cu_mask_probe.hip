// cu_mask_probe.hip — which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  (development aid)
//   hipcc --offload-arch=gfx950 -O2 scripts/cu_mask_probe.hip -o scripts/cu_mask_probe.bin && scripts/cu_mask_probe.bin
// Launches a kernel of many short workgroups on streams with different masks; every workgroup records (XCC_ID, HW_ID); the host prints, per
// mask, how many distinct CUs ran workgroups and how they spread over the XCCs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <set>
__global__ void k(unsigned* out, int spin) {
    unsigned long long t0 = clock64();
    while (clock64() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
        out[2 * blockIdx.x + 1] = (unsigned)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    }
}
int main() {
    const int nb = 4096;
    unsigned* d; hipMalloc(&d, nb * 8);
    std::vector<unsigned> h(nb * 2);
    struct Case { const char* name; std::vector<uint32_t> mask; };
    std::vector<Case> cases;
    cases.push_back({"all 256 bits", std::vector<uint32_t>(8, 0xffffffffu)});
    cases.push_back({"bits 0-63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}});
    cases.push_back({"bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}});
    cases.push_back({"bits 64-255", {0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}});
    cases.push_back({"every 4th bit", std::vector<uint32_t>(8, 0x11111111u)});
    cases.push_back({"bits 0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0}});
    for (auto& c : cases) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)c.mask.size(), c.mask.data());
        if (e != hipSuccess) { printf("%-16s hipExtStreamCreateWithCUMask: %s\n", c.name, hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0xff, nb * 8, s);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, s, d, 20000);
        hipStreamSynchronize(s);
        hipEventRecord(a, s);
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, s, d, 20000);
        hipEventRecord(b, s); hipStreamSynchronize(s);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (int i = 0; i < nb; ++i) per_xcc[h[2 * i] & 15u].insert((h[2 * i + 1] >> 8) & 0xffu);      // cu 11:8, sh 12, se 15:13
        size_t cus = 0; for (auto& kv : per_xcc) cus += kv.second.size();
        printf("%-16s %6.3f ms  %3zu CUs:", c.name, ms, cus);
        for (auto& kv : per_xcc) printf(" xcc%u:%zu", kv.first, kv.second.size());
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
