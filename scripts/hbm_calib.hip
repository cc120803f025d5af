// HBM counter calibration for gfx950 (development aid): kernels that move a KNOWN number of bytes in the access
// patterns this library uses, to be run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes).
//   hipcc --offload-arch=gfx950 -O3 scripts/hbm_calib.hip -o build/hbm_calib
// Buffers are 2 GiB (8x the 256 MiB Infinity Cache) so nothing is absorbed on-die.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read16(const float4* __restrict__ p, size_t n, float* out) {       // 16 B / lane, coalesced rows
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i]; acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *out = acc;
}
__global__ void read4(const float* __restrict__ p, size_t n, float* out) {          // 4 B / lane, coalesced
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678f) *out = acc;
}
__global__ void gather48(const float4* __restrict__ p, size_t n_rec, size_t n_gather, float* out) {   // random 48-B records
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_gather; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = (i * 2654435761ull + 12345ull) % n_rec;
        const float4 a = p[3 * r], b = p[3 * r + 1], c = p[3 * r + 2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 12345.678f) *out = acc;
}
__global__ void write16(float4* __restrict__ p, size_t n) {                          // 16 B / lane streaming store
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void scatter8(unsigned long long* __restrict__ p, size_t n_slots, size_t n_writes) {   // random 8-B stores
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_writes; i += (size_t)gridDim.x * blockDim.x)
        p[(i * 2654435761ull + 777ull) % n_slots] = i;
}

int main() {
    const size_t bytes = 2ull << 30;
    void* buf; float* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const dim3 g(256 * 16), b(256);
    hipLaunchKernelGGL(read16, g, b, 0, 0, (const float4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(read4, g, b, 0, 0, (const float*)buf, bytes / 4, out);
    hipLaunchKernelGGL(gather48, g, b, 0, 0, (const float4*)buf, bytes / 48, (size_t)(16u << 20), out);
    hipLaunchKernelGGL(write16, g, b, 0, 0, (float4*)buf, bytes / 16);
    hipLaunchKernelGGL(scatter8, g, b, 0, 0, (unsigned long long*)buf, bytes / 8, (size_t)(16u << 20));
    hipDeviceSynchronize();
    printf("known bytes: read16 %zu  read4 %zu  gather48 %zu (48 B x 16 Mi records, 64-B sectors: %zu, 128-B lines: up to %zu)  write16 %zu  scatter8 %zu (8 B x 16 Mi; 64-B sectors: %zu)\n",
           bytes, bytes, (size_t)48 * (16u << 20), (size_t)64 * (16u << 20) * 3 / 2, (size_t)128 * (16u << 20), bytes, (size_t)8 * (16u << 20), (size_t)64 * (16u << 20));
    return 0;
}
