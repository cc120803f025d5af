// tests/hipemu/hip/hip_runtime.h — a wave64 SIMT *emulator* for unit-testing kernel logic on a host
// without a GPU.  TEST INFRASTRUCTURE ONLY: it is not a backend, it is never built into or loaded by
// the product (libsage_gs.so is compiled by hipcc against the real <hip/hip_runtime.h>), and
// nothing under sage-3d_official_amd/ refers to it.  tests/ compile the product's own
// csrc/sgs_api.hip + csrc/sgs_kernels.h against this header with g++ so that the CPU test-suite can
// exercise the exact kernel source (ballot/shuffle compaction, LDS radix sort, composite) against
// the oracle before a GPU box is involved.
//
// Execution model: every workgroup runs its threads as ucontext fibers on one OS thread; a fiber
// runs until it reaches a workgroup barrier or a wave collective and then yields.  Collectives
// rendezvous over the lanes of a wave that are still alive (as the hardware's exec mask does for
// wave-uniform control flow — the only way the kernels use them).  Workgroups run in parallel on
// OpenMP threads, so global atomics are real atomics.
#pragma once
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <ucontext.h>

#include <functional>
#include <vector>


// ---- language surface ---------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

namespace hipemu {

constexpr size_t kStackBytes = 256 * 1024;

// Fiber switches: on x86-64 a dozen instructions of our own (callee-saved registers and the stack pointer) — glibc's swapcontext / getcontext
// make an rt_sigprocmask system call per switch, and a lane switches at every collective and every barrier: the system calls were a third of
// the CPU suite's time.  Elsewhere: ucontext.
#if defined(__x86_64__)
#define HIPEMU_ASM_SWITCH 1
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .weak hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
)");
#else
#define HIPEMU_ASM_SWITCH 0
#endif

struct Fiber {
#if HIPEMU_ASM_SWITCH
    void* sp = nullptr;
#else
    ucontext_t ctx;
#endif
    char* stack = nullptr;
    bool finished = true;
    dim3 tid;
};

struct WaveState {
    unsigned long long vals[2][64];
    unsigned long long part[2];
    unsigned long long cur_mask = 0;
    int arrived = 0, alive = 0;
    unsigned gen = 0;
};

struct BlockExec {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
#if HIPEMU_ASM_SWITCH
    void* sched = nullptr;
#else
    ucontext_t sched;
#endif
    int cur = 0, nthreads = 0, alive = 0, bar_arrived = 0;
    unsigned bar_gen = 0;
    dim3 bid, bdim, gdim;
    const std::function<void()>* body = nullptr;
};

inline BlockExec*& exec() { static thread_local BlockExec* e = nullptr; return e; }

#if HIPEMU_ASM_SWITCH
inline void yield() { BlockExec* e = exec(); hipemu_switch(&e->fibers[e->cur].sp, e->sched); }
#else
inline void yield() { BlockExec* e = exec(); swapcontext(&e->fibers[e->cur].ctx, &e->sched); }
#endif

inline void wave_release(WaveState& w) {
    const int p = w.gen & 1;
    w.part[p] = w.cur_mask; w.cur_mask = 0; w.arrived = 0; w.gen++;
}

// Rendezvous of the wave's live lanes; returns the parity slot holding everybody's value.
inline int collective(unsigned long long v) {
    BlockExec* e = exec();
    const unsigned t = e->fibers[e->cur].tid.x;
    WaveState& w = e->waves[t >> 6];
    const int lane = t & 63;
    const unsigned g = w.gen;
    const int p = g & 1;
    w.vals[p][lane] = v;
    w.cur_mask |= 1ull << lane;
    if (++w.arrived == w.alive) wave_release(w);
    else while (w.gen == g) yield();
    return p;
}

inline void fiber_exit_bookkeeping(BlockExec* e) {
    Fiber& f = e->fibers[e->cur];
    f.finished = true;
    e->alive--;
    WaveState& w = e->waves[f.tid.x >> 6];
    w.alive--;
    if (w.alive > 0 && w.arrived == w.alive) wave_release(w);
    if (e->alive > 0 && e->bar_arrived == e->alive) { e->bar_arrived = 0; e->bar_gen++; }
}

inline void fiber_entry() {
    BlockExec* e = exec();
    (*e->body)();
    fiber_exit_bookkeeping(e);
    yield();                               // (for good: a finished fiber is never switched to again)
    abort();
}

inline void run_block(BlockExec* e, dim3 bid, dim3 bdim, dim3 gdim, const std::function<void()>& body) {
    const int n = (int)bdim.x;
    if ((int)e->fibers.size() < n) {
        const size_t old = e->fibers.size();
        e->fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) {
            e->fibers[i].stack = (char*)mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE,
                                             MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (e->fibers[i].stack == (char*)MAP_FAILED) { perror("hipemu: mmap"); abort(); }
        }
    }
    e->waves.assign((n + 63) / 64, WaveState());
    e->nthreads = n; e->alive = n; e->bar_arrived = 0; e->bar_gen = 0;
    e->bid = bid; e->bdim = bdim; e->gdim = gdim; e->body = &body;
    for (int i = 0; i < n; ++i) {
        Fiber& f = e->fibers[i];
        f.finished = false; f.tid = dim3((unsigned)i);
        e->waves[i >> 6].alive++;
#if HIPEMU_ASM_SWITCH
        // the first switch to the fiber pops six registers and "returns" into fiber_entry with the stack as a call leaves it (rsp = 8 mod 16)
        void** top = reinterpret_cast<void**>((reinterpret_cast<uintptr_t>(f.stack) + kStackBytes) & ~uintptr_t(15)) - 2;
        top[0] = reinterpret_cast<void*>(&fiber_entry);
        for (int k = 1; k <= 6; ++k) top[-k] = nullptr;
        f.sp = top - 6;
#else
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStackBytes; f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
    }
    long spins = 0;
    while (e->alive > 0) {
        for (int i = 0; i < n; ++i) {
            if (e->fibers[i].finished) continue;
            e->cur = i;
#if HIPEMU_ASM_SWITCH
            hipemu_switch(&e->sched, e->fibers[i].sp);
#else
            swapcontext(&e->sched, &e->fibers[i].ctx);
#endif
        }
        if (++spins > 50000000L) { fprintf(stderr, "hipemu: deadlock (divergent collective/barrier?)\n"); abort(); }
    }
}

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const long nblocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel
    {
        static thread_local BlockExec* mine = nullptr;
        if (!mine) mine = new BlockExec;
        exec() = mine;
#pragma omp for schedule(dynamic, 1)
        for (long b = 0; b < nblocks; ++b)
            run_block(mine, dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y))),
                      block, grid, body);
    }
}

inline Fiber& me() { BlockExec* e = exec(); return e->fibers[e->cur]; }
template <class T> inline unsigned long long bits(T v) { unsigned long long b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T unbits(unsigned long long b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
template <class T> inline T shfl_from(T v, int src) {
    BlockExec* e = exec();
    const unsigned t = e->fibers[e->cur].tid.x;
    const int p = collective(bits(v));
    WaveState& w = e->waves[t >> 6];
    if (src < 0 || src > 63 || !((w.part[p] >> src) & 1ull)) return v;
    return unbits<T>(w.vals[p][src]);
}

}  // namespace hipemu

#define threadIdx (hipemu::me().tid)
#define blockIdx (hipemu::exec()->bid)
#define blockDim (hipemu::exec()->bdim)
#define gridDim (hipemu::exec()->gdim)

// ---- device intrinsics used by the kernels ------------------------------------------------------
static inline void __syncthreads() {
    hipemu::BlockExec* e = hipemu::exec();
    const unsigned g = e->bar_gen;
    if (++e->bar_arrived == e->alive) { e->bar_arrived = 0; e->bar_gen++; }
    else while (e->bar_gen == g) hipemu::yield();
}
static inline unsigned long long __ballot(bool pred) {
    hipemu::BlockExec* e = hipemu::exec();
    const unsigned t = e->fibers[e->cur].tid.x;
    const int p = hipemu::collective(pred ? 1ull : 0ull);
    hipemu::WaveState& w = e->waves[t >> 6];
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (((w.part[p] >> l) & 1ull) && w.vals[p][l]) m |= 1ull << l;
    return m;
}
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) { return __ballot(pred); }
template <class T> static inline T __shfl(T v, int src) { return hipemu::shfl_from(v, src & 63); }
template <class T> static inline T __shfl_up(T v, int d) { return hipemu::shfl_from(v, (int)(hipemu::me().tid.x & 63) - d); }
template <class T> static inline T __shfl_xor(T v, int d) { return hipemu::shfl_from(v, (int)((hipemu::me().tid.x & 63) ^ (unsigned)d)); }

static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicSub(unsigned* p, unsigned v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMin(unsigned* p, unsigned v) {
    unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}

static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
#define __log2f(x) log2f(x)
#define __fdividef(a, b) ((a) / (b))
static inline unsigned long long clock64() { return 0; }
#define __builtin_amdgcn_s_setprio(p_) ((void)0)
#define __builtin_amdgcn_s_getreg(reg_) (blockIdx.x & 7u)      /* emulated XCC id */
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 4
template <class T> static inline T __hip_atomic_load(const T* p, int, int) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
// device- / workgroup-scope fences: workgroups are OS threads here
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
template <class T> static inline T __hip_atomic_fetch_add(T* p, T v, int, int) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { return v; }   // callers pass wave-uniform values
// IEEE double multiply / subtract, each rounded on its own (never contracted into an fma)
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double __dsub_rn(double a, double b) { volatile double r = a - b; return r; }
// v_exp_f32 / v_rcp_f32 / v_sqrt_f32 (1 ulp on the hardware; correctly rounded here — the kernels pad every use)
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static inline double __builtin_amdgcn_rsq(double x) { return 1.0 / sqrt(x); }
// Wave-scope fence / barrier: a rendezvous of the wave's fibers orders their LDS accesses.
#define __builtin_amdgcn_fence(order_, scope_) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { (void)__ballot(true); }
// v_mbcnt_lo/hi_u32_b32: base + the set bits of the mask half below this lane.
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base) {
    const unsigned lane = hipemu::me().tid.x & 63u;
    return base + (unsigned)__builtin_popcount(lane >= 32u ? mask : mask & ((1u << lane) - 1u));
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base) {
    const unsigned lane = hipemu::me().tid.x & 63u;
    return base + (lane > 32u ? (unsigned)__builtin_popcount(mask & ((1u << (lane - 32u)) - 1u)) : 0u);
}
// v_readlane_b32: the value of lane `src` (callers read an active lane).
static inline int __builtin_amdgcn_readlane(int v, int src) { return hipemu::shfl_from(v, src & 63); }
// v_mov_b32 with a DPP modifier (the controls the kernels use): every lane of the wave takes part; a lane whose source
// lane does not exist, or whose row is masked off, gets `old` (bound_ctrl = false: the destination is not written).
//   0x00..0xff quad_perm   0x101..0x10f row_shl:n   0x111..0x11f row_shr:n   0x142 row_bcast:15   0x143 row_bcast:31
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    hipemu::BlockExec* e = hipemu::exec();
    const unsigned t = e->fibers[e->cur].tid.x;
    const int lane = (int)(t & 63), row = lane >> 4, in_row = lane & 15;
    const int p = hipemu::collective(hipemu::bits(src));
    hipemu::WaveState& w = e->waves[t >> 6];
    int from = -1;
    if (ctrl >= 0 && ctrl <= 0xff) from = (lane & ~3) + ((ctrl >> (2 * (lane & 3))) & 3);     // quad_perm:[a,b,c,d]
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; if (in_row >= n) from = lane - n; }
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl - 0x100; if (in_row + n <= 15) from = lane + n; }
    else if (ctrl == 0x142) { if (row >= 1) from = 16 * row - 1; }                  // last lane of the previous row
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }                            // last lane of row 1
    else { fprintf(stderr, "hipemu: DPP control 0x%x is not emulated\n", ctrl); abort(); }
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
    if (from < 0 || !((w.part[p] >> from) & 1ull)) return bound_ctrl ? 0 : old;
    return hipemu::unbits<int>(w.vals[p][from]);
}
static inline int min(int a, int b) { return a < b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }

// ---- host runtime surface used by sgs_api.hip ---------------------------------------------------
typedef enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 } hipError_t;
typedef enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
typedef void* hipStream_t;
typedef struct hipemu_event { double t; }* hipEvent_t;

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) {
    *p = aligned_alloc(256, ((n + 255) / 256) * 256);
    if (*p) memset(*p, 0xCD, n);            // poison: uninitialised reads show up
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event{0.0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); e->t = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; return hipSuccess;
}
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
// streams are synchronous in the emulator: every "stream" executes at enqueue time
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)new int(0); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete (int*)s; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }

// dynamic LDS: a buffer per OS thread (= per workgroup in flight), sized by the launch that is running
namespace hipemu {
inline size_t& dyn_bytes() { static size_t b = 0; return b; }
inline void* dyn_shared() {
    static thread_local std::vector<char> buf;
    if (buf.size() < dyn_bytes()) buf.resize(dyn_bytes());
    return buf.data();
}
}  // namespace hipemu
#define SGS_DYNAMIC_LDS(T, name) T* const name = static_cast<T*>(hipemu::dyn_shared())
#define SGS_PIN_VGPR(x) ((void)(x))      // (register-class hint of the GPU build)
// v_fmac_f32 with a DPP quad_perm:[I,I,I,I] source (the GPU build writes it as asm)
#define SGS_FMAC_QUAD(C, v, I, w)                                                                      \
    C = __builtin_fmaf(__uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), (I) * 0x55, 0xf, 0xf, true)), w, C);
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    (hipemu::dyn_bytes() = (size_t)(shmem), hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); }))
