// Instruction-rate micro-benchmark for gfx950 (development aid; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
// Every kernel runs ITER trips of 16 independent instructions of one kind per wave; blocks of 256 threads,
// WPS waves per SIMD resident on every CU.  Reported: issue cycles per wave-instruction per SIMD, taking the
// plain v_fma_f32 loop as 4.0 cycles' worth of clock (and the clock that implies).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096

#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, const float* in) {
    __shared__ float4 lds[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 256) lds[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    float a[16];
    v2f p[16];
    unsigned long long q[16];
    for (int i = 0; i < 16; ++i) { a[i] = in[i] + tid; p[i] = v2f{in[i], in[i + 1]}; q[i] = (unsigned long long)(in[i] * 1000.f) + tid; }
    const float b = in[17], c = in[18];
    const v2f pb = {in[19], in[20]}, pc = {in[21], in[22]};
    unsigned cnt = 0;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc));
        } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(q[i]), "v"(q[(i + 1) & 15]) : "vcc");
        } else if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
        } else if (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
        } else if (KIND == 6) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
        } else if (KIND == 7) {   // broadcast ds_read_b128
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)((it & 63) * 256)), "n"(i * 16));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc.x += v.x;
            }
        } else if (KIND == 8) {   // v_mul_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 9) {   // v_add_co / addc pair stand-in: v_add_u32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(cnt) : "v"((unsigned)tid));
        } else if (KIND == 10) {  // v_med3_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 11) {  // v_pk_add_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
        } else if (KIND == 12) {  // scalar: s_and_b64 chain (shared scalar unit?)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("s_and_b64 s[20:21], s[20:21], exec" ::: "s20", "s21", "scc");
        } else if (KIND == 13) {  // per-lane ds_read_b128 (stride 16 B: conflict-free)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(((tid + i * 64) & 1023) * 16)));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc.x += v.x;
            }
        } else if (KIND == 14) {  // v_fma_f32 with a dependent chain of 4 (latency view): 4 chains x 4
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 3]) : "v"(b), "v"(c));
        } else if (KIND == 15) {  // v_fmac (VOP2) form
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 16) {  // v_cmp to SGPR pair + v_cndmask from it (VOP3)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
                asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
            }
        } else if (KIND == 17) {  // ballot-like: v_cmp + s_cbranch-free scalar read
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                asm volatile("s_and_b64 s[20:21], vcc, exec" ::: "s20", "s21", "scc");
            }
        } else if (KIND == 18) {  // v_cndmask e64 with an SGPR pair mask, no compare in the loop
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 19) {  // v_cmp (vcc) + v_cndmask e32 pairs
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            }
        } else if (KIND == 20) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 21) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 22) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
        } else if (KIND == 23) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 24) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 25) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 26) {  // v_cmp e64 -> SGPR pair only
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
        } else if (KIND == 27) {  // v_cmp_class / cmpx? -> v_cmpx writes exec: skip; v_mad_u32_u24
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 28) {  // ds_read_b64 broadcast
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                v2f v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)((it & 63) * 256)), "n"(i * 8));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc.x += v.x;
            }
        } else if (KIND == 29) {  // ds_read_b32 broadcast
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)((it & 63) * 256)), "n"(i * 4));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc.x += v;
            }
        } else if (KIND == 30) {  // v_readfirstlane
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a[i]) : "s20");
        } else if (KIND == 31) {  // v_fma with an SGPR operand
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, s20, %0" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 32) {  // v_sub_f32 with SGPR source (VOP2 e32)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_sub_f32 %0, s20, %0" : "+v"(a[i]));
        } else if (KIND == 33) {  // v_fma_f32 with two SGPRs? not allowed on gfx9 (one constant bus read): use literal
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0" : "+v"(a[i]));
        } else if (KIND == 34) {  // s_ff1 / s_andn2 chain (bit scan walk)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("s_ff1_i32_b64 s22, s[20:21]" ::: "s22");
                asm volatile("s_bitset0_b64 s[20:21], s22" ::: "s20", "s21");
            }
        } else if (KIND == 35) {  // s_load_dwordx4 from a constant address (scalar cache hit)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("s_load_dwordx4 s[24:27], %0, 0x0" : : "s"(in) : "s24", "s25", "s26", "s27");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            }
        } else if (KIND == 36) {  // v_exp_f32 interleaved with independent fma (does the transcendental pipe overlap?)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i + 8]) : "v"(b), "v"(c));
            }
        } else if (KIND == 37) {  // v_cmp_lt_f32 + v_cndmask via sgpr, interleaved with 2 fma each
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i + 4]) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i + 8]) : "v"(b), "v"(c));
                asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
            }
        } else if (KIND == 38) {  // DPP mov (cross-lane)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 39) {  // v_fma_f64
            double* d = reinterpret_cast<double*>(q);
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
        } else if (KIND == 40) {  // v_fma_f32 with a neg modifier
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 41) {  // v_mul_f32 e64 with a neg modifier
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32_e64 %0, %1, -%0" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 42) {  // v_med3_f32 with an SGPR and an inline constant
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_med3_f32 %0, %0, s20, 0" : "+v"(a[i]) : : );
        } else if (KIND == 43) {  // v_cmp_le_f32 vcc, sgpr, vgpr
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cmp_le_f32 vcc, s20, %0" : : "v"(a[i]) : "vcc");
        } else if (KIND == 44) {  // v_exp_f32 e64 with neg
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32_e64 %0, -%0" : "+v"(a[i]));
        } else if (KIND == 45) {  // the composite's alpha sequence (12 instructions per splat), four splats interleaved
            for (int rep = 0; rep < 1; ++rep) {
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*0+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*1+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*2+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*3+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*0+1]) : "v"(c), "v"(b));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*1+1]) : "v"(c), "v"(b));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*2+1]) : "v"(c), "v"(b));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*3+1]) : "v"(c), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*0+2]) : "v"(a[4*0+0]), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*1+2]) : "v"(a[4*1+0]), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*2+2]) : "v"(a[4*2+0]), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*3+2]) : "v"(a[4*3+0]), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*0+2]) : "v"(c), "v"(a[4*0+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*1+2]) : "v"(c), "v"(a[4*1+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*2+2]) : "v"(c), "v"(a[4*2+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*3+2]) : "v"(c), "v"(a[4*3+1]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*0+3]) : "v"(a[4*0+1]), "v"(c));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*1+3]) : "v"(a[4*1+1]), "v"(c));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*2+3]) : "v"(a[4*2+1]), "v"(c));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*3+3]) : "v"(a[4*3+1]), "v"(c));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*0+3]) : "v"(a[4*0+1]));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*1+3]) : "v"(a[4*1+1]));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*2+3]) : "v"(a[4*2+1]));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*3+3]) : "v"(a[4*3+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*0+3]) : "v"(a[4*0+0]), "v"(a[4*0+2]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*1+3]) : "v"(a[4*1+0]), "v"(a[4*1+2]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*2+3]) : "v"(a[4*2+0]), "v"(a[4*2+2]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*3+3]) : "v"(a[4*3+0]), "v"(a[4*3+2]));
                asm volatile("v_exp_f32_e64 %0, -%1" : "=v"(a[4*0+0]) : "v"(a[4*0+3]));
                asm volatile("v_exp_f32_e64 %0, -%1" : "=v"(a[4*1+0]) : "v"(a[4*1+3]));
                asm volatile("v_exp_f32_e64 %0, -%1" : "=v"(a[4*2+0]) : "v"(a[4*2+3]));
                asm volatile("v_exp_f32_e64 %0, -%1" : "=v"(a[4*3+0]) : "v"(a[4*3+3]));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*0+0]) : "v"(b));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*1+0]) : "v"(b));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*2+0]) : "v"(b));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*3+0]) : "v"(b));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[4*0+0]) : "v"(c));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[4*1+0]) : "v"(c));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[4*2+0]) : "v"(c));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[4*3+0]) : "v"(c));
                asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b), "v"(a[4*0+3]) : "vcc");
                asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b), "v"(a[4*1+3]) : "vcc");
                asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b), "v"(a[4*2+3]) : "vcc");
                asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b), "v"(a[4*3+3]) : "vcc");
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*0+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*1+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*2+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*3+0]) : : );
            }
        } else if (KIND == 46) {  // the exponent-domain alpha sequence (11 per splat), four splats interleaved
            for (int rep = 0; rep < 1; ++rep) {
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*0+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*1+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*2+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*3+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*0+1]) : "v"(c), "v"(b));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*1+1]) : "v"(c), "v"(b));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*2+1]) : "v"(c), "v"(b));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*3+1]) : "v"(c), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*0+2]) : "v"(a[4*0+0]), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*1+2]) : "v"(a[4*1+0]), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*2+2]) : "v"(a[4*2+0]), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*3+2]) : "v"(a[4*3+0]), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*0+2]) : "v"(c), "v"(a[4*0+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*1+2]) : "v"(c), "v"(a[4*1+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*2+2]) : "v"(c), "v"(a[4*2+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*3+2]) : "v"(c), "v"(a[4*3+1]));
                asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(a[4*0+3]) : "v"(c), "v"(a[4*0+1]));
                asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(a[4*1+3]) : "v"(c), "v"(a[4*1+1]));
                asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(a[4*2+3]) : "v"(c), "v"(a[4*2+1]));
                asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(a[4*3+3]) : "v"(c), "v"(a[4*3+1]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4*0+3]) : "v"(a[4*0+1]), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4*1+3]) : "v"(a[4*1+1]), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4*2+3]) : "v"(a[4*2+1]), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4*3+3]) : "v"(a[4*3+1]), "v"(b));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[4*0+3]) : "v"(a[4*0+0]), "v"(a[4*0+2]));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[4*1+3]) : "v"(a[4*1+0]), "v"(a[4*1+2]));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[4*2+3]) : "v"(a[4*2+0]), "v"(a[4*2+2]));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[4*3+3]) : "v"(a[4*3+0]), "v"(a[4*3+2]));
                asm volatile("v_cmp_le_f32 vcc, s20, %0" : : "v"(a[4*0+3]) : "vcc");
                asm volatile("v_cmp_le_f32 vcc, s20, %0" : : "v"(a[4*1+3]) : "vcc");
                asm volatile("v_cmp_le_f32 vcc, s20, %0" : : "v"(a[4*2+3]) : "vcc");
                asm volatile("v_cmp_le_f32 vcc, s20, %0" : : "v"(a[4*3+3]) : "vcc");
                asm volatile("v_exp_f32 %0, %1" : "=v"(a[4*0+0]) : "v"(a[4*0+3]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(a[4*1+0]) : "v"(a[4*1+3]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(a[4*2+0]) : "v"(a[4*2+3]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(a[4*3+0]) : "v"(a[4*3+3]));
                asm volatile("v_med3_f32 %0, %0, s21, 0" : "+v"(a[4*0+0]) : : );
                asm volatile("v_med3_f32 %0, %0, s21, 0" : "+v"(a[4*1+0]) : : );
                asm volatile("v_med3_f32 %0, %0, s21, 0" : "+v"(a[4*2+0]) : : );
                asm volatile("v_med3_f32 %0, %0, s21, 0" : "+v"(a[4*3+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*0+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*1+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*2+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*3+0]) : : );
            }
        } else if (KIND == 47) {  // alpha sequence, splat after splat
            for (int rep = 0; rep < 1; ++rep) {
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*0+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*0+1]) : "v"(c), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*0+2]) : "v"(a[4*0+0]), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*0+2]) : "v"(c), "v"(a[4*0+1]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*0+3]) : "v"(a[4*0+1]), "v"(c));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*0+3]) : "v"(a[4*0+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*0+3]) : "v"(a[4*0+0]), "v"(a[4*0+2]));
                asm volatile("v_exp_f32_e64 %0, -%1" : "=v"(a[4*0+0]) : "v"(a[4*0+3]));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*0+0]) : "v"(b));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[4*0+0]) : "v"(c));
                asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b), "v"(a[4*0+3]) : "vcc");
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*0+0]) : : );
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*1+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*1+1]) : "v"(c), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*1+2]) : "v"(a[4*1+0]), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*1+2]) : "v"(c), "v"(a[4*1+1]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*1+3]) : "v"(a[4*1+1]), "v"(c));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*1+3]) : "v"(a[4*1+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*1+3]) : "v"(a[4*1+0]), "v"(a[4*1+2]));
                asm volatile("v_exp_f32_e64 %0, -%1" : "=v"(a[4*1+0]) : "v"(a[4*1+3]));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*1+0]) : "v"(b));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[4*1+0]) : "v"(c));
                asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b), "v"(a[4*1+3]) : "vcc");
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*1+0]) : : );
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*2+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*2+1]) : "v"(c), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*2+2]) : "v"(a[4*2+0]), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*2+2]) : "v"(c), "v"(a[4*2+1]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*2+3]) : "v"(a[4*2+1]), "v"(c));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*2+3]) : "v"(a[4*2+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*2+3]) : "v"(a[4*2+0]), "v"(a[4*2+2]));
                asm volatile("v_exp_f32_e64 %0, -%1" : "=v"(a[4*2+0]) : "v"(a[4*2+3]));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*2+0]) : "v"(b));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[4*2+0]) : "v"(c));
                asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b), "v"(a[4*2+3]) : "vcc");
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*2+0]) : : );
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*3+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*3+1]) : "v"(c), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*3+2]) : "v"(a[4*3+0]), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*3+2]) : "v"(c), "v"(a[4*3+1]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*3+3]) : "v"(a[4*3+1]), "v"(c));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*3+3]) : "v"(a[4*3+1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*3+3]) : "v"(a[4*3+0]), "v"(a[4*3+2]));
                asm volatile("v_exp_f32_e64 %0, -%1" : "=v"(a[4*3+0]) : "v"(a[4*3+3]));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[4*3+0]) : "v"(b));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[4*3+0]) : "v"(c));
                asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b), "v"(a[4*3+3]) : "vcc");
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*3+0]) : : );
            }
        } else if (KIND == 48) {  // exponent-domain sequence, splat after splat
            for (int rep = 0; rep < 1; ++rep) {
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*0+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*0+1]) : "v"(c), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*0+2]) : "v"(a[4*0+0]), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*0+2]) : "v"(c), "v"(a[4*0+1]));
                asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(a[4*0+3]) : "v"(c), "v"(a[4*0+1]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4*0+3]) : "v"(a[4*0+1]), "v"(b));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[4*0+3]) : "v"(a[4*0+0]), "v"(a[4*0+2]));
                asm volatile("v_cmp_le_f32 vcc, s20, %0" : : "v"(a[4*0+3]) : "vcc");
                asm volatile("v_exp_f32 %0, %1" : "=v"(a[4*0+0]) : "v"(a[4*0+3]));
                asm volatile("v_med3_f32 %0, %0, s21, 0" : "+v"(a[4*0+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*0+0]) : : );
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*1+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*1+1]) : "v"(c), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*1+2]) : "v"(a[4*1+0]), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*1+2]) : "v"(c), "v"(a[4*1+1]));
                asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(a[4*1+3]) : "v"(c), "v"(a[4*1+1]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4*1+3]) : "v"(a[4*1+1]), "v"(b));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[4*1+3]) : "v"(a[4*1+0]), "v"(a[4*1+2]));
                asm volatile("v_cmp_le_f32 vcc, s20, %0" : : "v"(a[4*1+3]) : "vcc");
                asm volatile("v_exp_f32 %0, %1" : "=v"(a[4*1+0]) : "v"(a[4*1+3]));
                asm volatile("v_med3_f32 %0, %0, s21, 0" : "+v"(a[4*1+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*1+0]) : : );
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*2+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*2+1]) : "v"(c), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*2+2]) : "v"(a[4*2+0]), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*2+2]) : "v"(c), "v"(a[4*2+1]));
                asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(a[4*2+3]) : "v"(c), "v"(a[4*2+1]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4*2+3]) : "v"(a[4*2+1]), "v"(b));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[4*2+3]) : "v"(a[4*2+0]), "v"(a[4*2+2]));
                asm volatile("v_cmp_le_f32 vcc, s20, %0" : : "v"(a[4*2+3]) : "vcc");
                asm volatile("v_exp_f32 %0, %1" : "=v"(a[4*2+0]) : "v"(a[4*2+3]));
                asm volatile("v_med3_f32 %0, %0, s21, 0" : "+v"(a[4*2+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*2+0]) : : );
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*3+0]) : "v"(b), "v"(c));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[4*3+1]) : "v"(c), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[4*3+2]) : "v"(a[4*3+0]), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4*3+2]) : "v"(c), "v"(a[4*3+1]));
                asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(a[4*3+3]) : "v"(c), "v"(a[4*3+1]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4*3+3]) : "v"(a[4*3+1]), "v"(b));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[4*3+3]) : "v"(a[4*3+0]), "v"(a[4*3+2]));
                asm volatile("v_cmp_le_f32 vcc, s20, %0" : : "v"(a[4*3+3]) : "vcc");
                asm volatile("v_exp_f32 %0, %1" : "=v"(a[4*3+0]) : "v"(a[4*3+3]));
                asm volatile("v_med3_f32 %0, %0, s21, 0" : "+v"(a[4*3+0]) : : );
                asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[4*3+0]) : : );
            }
        }
    }
    float s = acc.x + cnt;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)q[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int KIND>
float run(int wps, float* out, const float* in) {
    const int blocks = 256 * wps;     // 256 CUs x wps blocks of 4 waves -> wps waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, in);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, in);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

typedef float (*runfn)(int, float*, const float*);
int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    float *out, *in;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    std::vector<float> h(64, 0.5f);
    hipMalloc(&in, 64 * sizeof(float));
    hipMemcpy(in, h.data(), 64 * sizeof(float), hipMemcpyHostToDevice);
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_cmp_lt_u64", "v_pk_mul_f32", "v_cmp_lt_f32(vcc)",
                           "v_cndmask_b32", "ds_read_b128 bcast", "v_mul_f32", "v_add_u32 (dep)", "v_med3_f32", "v_pk_add_f32",
                           "s_and_b64 (dep)", "ds_read_b128 lane", "v_fma dep4", "v_fmac_f32", "cmp->sgpr+cndmask (x2)", "cmp vcc + s_and (x2)",
                           "v_cndmask e64 sgpr", "cmp vcc+cndmask e32 (x2)", "v_max_f32", "v_add_f32", "v_mov_b32", "v_and_b32", "v_log_f32", "v_rcp_f32",
                           "v_cmp e64 sgpr", "v_mad_u32_u24", "ds_read_b64 bcast", "ds_read_b32 bcast", "v_readfirstlane", "v_fma sgpr opnd", "v_sub_f32 sgpr e32",
                           "v_mul literal", "s_ff1+s_bitset0 (x2)", "s_load_dwordx4", "exp|fma interleaved", "cmp,fma,fma,cndmask", "v_mov_dpp", "v_fma_f64",
                           "v_fma neg", "v_mul e64 neg", "v_med3 sgpr,0", "v_cmp_le vcc,sgpr", "v_exp e64 neg", "alpha x4 interleaved (48)", "alpha-exp x4 interleaved (44)", "alpha x4 sequential (48)", "alpha-exp x4 sequential (44)"};
    runfn fns[] = {run<0>, run<1>, run<2>, run<3>, run<4>, run<5>, run<6>, run<7>, run<8>, run<9>, run<10>, run<11>, run<12>,
                   run<13>, run<14>, run<15>, run<16>, run<17>, run<18>, run<19>, run<20>, run<21>, run<22>, run<23>, run<24>, run<25>,
                   run<26>, run<27>, run<28>, run<29>, run<30>, run<31>, run<32>, run<33>, run<34>, run<35>, run<36>, run<37>, run<38>, run<39>,
                   run<40>, run<41>, run<42>, run<43>, run<44>, run<45>, run<46>, run<47>, run<48>};
    printf("%-26s", names[kind]);
    for (int wps : {1, 6}) {
        const float base = run<0>(wps, out, in);
        const float ms = fns[kind](wps, out, in);
        printf("  w%d: %7.3f ms %6.2f cyc (fma clock %.2f GHz)", wps, ms, 4.0 * ms / base, (double)ITER * 16 * wps * 4 / (base * 1e-3) / 1e9);
        fflush(stdout);
    }
    printf("\n");
    return 0;
}
