// ubench2.hip — issue-rate table for gfx950 with IN-KERNEL clocks (development aid; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I sage-3d_official_amd/csrc scripts/ubench2.hip -o scripts/ubench2.bin
//   scripts/ubench2.bin            (on the GPU box)
// Round 4's table (scripts/ubench.hip) DEFINED v_fma_f32 as 4 cycles and derived the clock from it.  This one measures:
// every wave reads s_memtime (shader clock, clock64) and s_memrealtime (constant 100 MHz, wall_clock64) around its loop, so
//   shader clock        = sum(d clock64) / sum(d wall_clock64) x wall-clock rate
//   wave-instr / cycle / SIMD = waves per SIMD x instructions per wave / median(d clock64)
// Waves per SIMD is not assumed: every wave records HW_ID / XCC_ID and the host counts the waves each (XCC, SE, CU, SIMD) held.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <algorithm>
#include "sgs_kernels.h"

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096
#define NI 16            // instructions per trip of the plain kinds

struct WaveRec { unsigned long long c0, c1, w0, w1; unsigned hwid, xcc; };

enum Kind { K_FMA, K_FMA_DEP, K_PK_FMA, K_MUL, K_EXP, K_MED3, K_CMP, K_CNDMASK, K_MAX, K_FMA_SGPR, K_EXP_FMA31, K_RCP,
            K_DS64_BCAST, K_DS64_LANE, K_DS128_BCAST, K_DS32_BCAST, K_DS64_BCAST_FMA4, K_DS64_BCAST_FMA8,
            K_TRIP, K_TRIP_NOLDS, K_TRIP_LDSONLY, K_TRIP_P, K_TRIP_PF, K_TRIP_PPF, K_TRIP_PK, K_TRIP_MFMA, K_TRIP_MFMA_NOLDS, K_MFMA16, K_COUNT };
static const char* kind_name[] = {"v_fma_f32", "v_fma_f32 dependent chain", "v_pk_fma_f32", "v_mul_f32", "v_exp_f32", "v_med3_f32", "v_cmp_lt_f32 vcc",
                                  "v_cndmask_b32 vcc", "v_max_f32", "v_fma_f32 sgpr operand", "3 v_fma + 1 v_exp", "v_rcp_f32",
                                  "ds_read_b64 broadcast", "ds_read_b64 per lane", "ds_read_b128 broadcast", "ds_read_b32 broadcast",
                                  "ds_read_b64 bcast + 4 v_fma", "ds_read_b64 bcast + 8 v_fma",
                                  "blend trip (4 splats: 66 VALU + 21 LDS)", "blend trip, splats in VGPRs", "blend trip, LDS reads only",
                                  "blend trip, prefix-product apply", "blend trip, list entry read a trip ahead", "blend trip, prefix-product + read ahead", "blend trip, (t0,V) and (C0,C1) as v_pk_fma_f32",
                                  "blend trip, U and V on the matrix pipe", "... its arithmetic alone (no LDS)", "v_mfma_f32_16x16x1_4b_f32"};
// instructions (of the kind the row is about) per trip
static const int kind_ni[] = {NI, NI, NI, NI, NI, NI, NI, NI, NI, NI, NI, NI, NI, NI, NI, NI, NI, NI, 1, 1, 1, 1, 1, 1, 1, 1, 1, NI};

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, const float* in, WaveRec* rec, int iters) {
    __shared__ __attribute__((aligned(16))) float2 s_arena[5 * (SGS_BATCH + 1)];
    __shared__ __attribute__((aligned(16))) unsigned s_sorted[4 * (SGS_BATCH + 4)];
    float2* const s_p0 = s_arena; float2* const s_p1 = s_arena + (SGS_BATCH + 1); float2* const s_p2 = s_arena + 2 * (SGS_BATCH + 1);
    float2* const s_p3 = s_arena + 3 * (SGS_BATCH + 1); float2* const s_p4 = s_arena + 4 * (SGS_BATCH + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a staged batch of plausible splats: centres spread over the tile, a ~ 0.3 / px, alphas small (the pixel never saturates)
    for (int j = tid; j <= SGS_BATCH; j += 256) {
        const float rx = (float)((j * 7) & 15), ry = (float)((j * 11) & 15), a = 0.25f + 0.001f * j, ak = 0.05f, c = 0.3f;
        s_p0[j] = make_float2(a * rx + ak * ry, a); s_p1[j] = make_float2(ak, c * ry); s_p2[j] = make_float2(c, j == SGS_BATCH ? 1.0e30f : 6.0f);
        s_p3[j] = make_float2(0.5f, 0.25f); s_p4[j] = make_float2(0.125f, 3.0f);
    }
    if (KIND == K_TRIP_MFMA) {       // the candidate's staged layout over the same arena: uv[j] = (m, -a, -a k | c ry, 0, -c), cn[j] = (r, g, b, nlo)
        __syncthreads();
        float* const uv = reinterpret_cast<float*>(s_arena);
        float4* const cn = reinterpret_cast<float4*>(uv + 6 * (SGS_BATCH + 2));
        for (int j = tid; j <= SGS_BATCH; j += 256) {
            const float rx = (float)((j * 7) & 15), ry = (float)((j * 11) & 15), a = 0.25f + 0.001f * j, ak = 0.05f, c = 0.3f;
            uv[6 * j] = a * rx + ak * ry; uv[6 * j + 1] = -a; uv[6 * j + 2] = -ak; uv[6 * j + 3] = c * ry; uv[6 * j + 4] = 0.f; uv[6 * j + 5] = -c;
            cn[j] = make_float4(0.5f, 0.25f, 0.125f, j == SGS_BATCH ? 1.0e30f : 6.0f);
        }
    }
    unsigned* const lst = s_sorted + (unsigned)wave * (SGS_BATCH + 4);
    for (int j = lane; j < SGS_BATCH + 4; j += 64) lst[j] = (unsigned)(j < 64 ? ((j * 4 + wave) & 255) : SGS_BATCH) << 3;
    __syncthreads();
    float a[NI]; v2f p[NI];
    for (int i = 0; i < NI; ++i) { a[i] = in[i] + tid * 1e-3f; p[i] = v2f{in[i], in[i + 1]}; }
    float b = in[17], c = in[18];
    const v2f pb = {in[19], in[20]}, pc = {in[21], in[22]};
    float sb = in[23];                                   // stays in an SGPR (uniform load)
    float4 acc = make_float4(0, 0, 0, 0);
    // the blend's state, named as the macros of sgs_kernels.h expect
    constexpr bool AUX = false, STATS = false, TF = false;
    const float lx = (float)((wave & 1) * 8 + (lane & 7)), ly = (float)((wave >> 1) * 8 + (lane >> 3));
    float amax = 0.99f, big = SGS_BIG; SGS_PIN_VGPR(amax); SGS_PIN_VGPR(big);
    const float cq = 7.98f; float cq_big = cq * SGS_BIG, nt_big = -(0.99f * 1e-4f) * SGS_BIG; SGS_PIN_VGPR(cq_big); SGS_PIN_VGPR(nt_big);
    float Tm = amax, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dz = 0.f, Wsum = 0.f; unsigned used = 0; const unsigned base = 0, m = 64; (void)m; (void)cq;
    const unsigned cntq = 64; bool wave_done = false; (void)wave_done;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_FMA) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == K_FMA_DEP) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(b), "v"(c));
        } else if (KIND == K_PK_FMA) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc));
        } else if (KIND == K_MUL) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        } else if (KIND == K_EXP) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == K_MED3) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == K_CMP) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
        } else if (KIND == K_CNDMASK) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
        } else if (KIND == K_MAX) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        } else if (KIND == K_FMA_SGPR) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sb), "v"(c));
        } else if (KIND == K_EXP_FMA31) {
#pragma unroll
            for (int i = 0; i < NI; i += 4) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i + 1]) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i + 2]) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i + 3]) : "v"(b), "v"(c));
            }
        } else if (KIND == K_RCP) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == K_DS64_BCAST || KIND == K_DS64_BCAST_FMA4 || KIND == K_DS64_BCAST_FMA8) {
            const unsigned addr = (unsigned)((it & 15) * 128);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                v2f v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(i * 8));
                if (KIND != K_DS64_BCAST) {
#pragma unroll
                    for (int f = 0; f < (KIND == K_DS64_BCAST_FMA4 ? 4 : 8); ++f) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(i + f) & (NI - 1)]) : "v"(b), "v"(c));
                }
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                asm volatile("" :: "v"(v));
            }
        } else if (KIND == K_DS64_LANE) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                v2f v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(lane * 8)), "n"(i * 512));
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                asm volatile("" :: "v"(v));
            }
        } else if (KIND == K_DS128_BCAST) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)((it & 15) * 256)), "n"(i * 16));
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            }
        } else if (KIND == K_DS32_BCAST) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)((it & 15) * 64)), "n"(i * 4));
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                asm volatile("" :: "v"(v));
            }
        } else if (KIND == K_TRIP) {
            // the product's list loop over 64 splats: 16 trips of 4
            Tm = amax;
            SGS_LIST_LOOP(SGS_ALPHA_F)
        } else if (KIND == K_TRIP_P || KIND == K_TRIP_PF || KIND == K_TRIP_PPF) {
            // candidates: (P) the four splats' stop rule and weights from PREFIX PRODUCTS of (1 - alpha) — the serial part of a trip is
            // 3 dependent multiplies instead of 16 dependent operations; (PF) the next trip's list entry is read before this trip's arithmetic
            Tm = amax;
#define UB_APPLY4(A, B, C, D)                                                                          \
            {                                                                                          \
                const float e0 = __builtin_fmaf(-al0, amax, 1.0f), e1 = __builtin_fmaf(-al1, amax, 1.0f);     \
                const float e2 = __builtin_fmaf(-al2, amax, 1.0f), e3 = __builtin_fmaf(-al3, amax, 1.0f);     \
                const float P2 = e0 * e1, P34 = e2 * e3, P3 = P2 * e2, P4 = P2 * P34;                  \
                const float T1 = Tm * e0, T2 = Tm * P2, T3 = Tm * P3, T4 = Tm * P4;                    \
                const float l1 = SGS_SAT(__builtin_fmaf(T1, big, nt_big)), l2 = SGS_SAT(__builtin_fmaf(T2, big, nt_big)); \
                const float l3 = SGS_SAT(__builtin_fmaf(T3, big, nt_big)), l4 = SGS_SAT(__builtin_fmaf(T4, big, nt_big)); \
                const float w0 = al0 * Tm * l1, w1 = al1 * T1 * l2, w2 = al2 * T2 * l3, w3 = al3 * T3 * l4; \
                Tm = T4 * l4;                                                                          \
                C0 = __builtin_fmaf(w0, (A##3).x, C0); C1 = __builtin_fmaf(w0, (A##3).y, C1); C2 = __builtin_fmaf(w0, (A##4).x, C2); \
                C0 = __builtin_fmaf(w1, (B##3).x, C0); C1 = __builtin_fmaf(w1, (B##3).y, C1); C2 = __builtin_fmaf(w1, (B##4).x, C2); \
                C0 = __builtin_fmaf(w2, (C##3).x, C0); C1 = __builtin_fmaf(w2, (C##3).y, C1); C2 = __builtin_fmaf(w2, (C##4).x, C2); \
                C0 = __builtin_fmaf(w3, (D##3).x, C0); C1 = __builtin_fmaf(w3, (D##3).y, C1); C2 = __builtin_fmaf(w3, (D##4).x, C2); \
            }
            uint4 nxt = *reinterpret_cast<const uint4*>(lst);
            for (unsigned kq = 0; kq < cntq; kq += 4) {
                uint4 pk;
                if (KIND == K_TRIP_P) pk = *reinterpret_cast<const uint4*>(lst + kq);
                else { pk = nxt; nxt = *reinterpret_cast<const uint4*>(lst + kq + 4); }
                const unsigned o0 = pk.x, o1 = pk.y, o2 = pk.z, o3 = pk.w;
                SGS_LOAD(o0, sa) SGS_LOAD(o1, sb_) SGS_LOAD(o2, sc) SGS_LOAD(o3, sd)
                SGS_ALPHA_F(o0, sa, al0) SGS_ALPHA_F(o1, sb_, al1) SGS_ALPHA_F(o2, sc, al2) SGS_ALPHA_F(o3, sd, al3)
                if (KIND == K_TRIP_PF) { SGS_APPLY(sa, al0) SGS_APPLY(sb_, al1) SGS_APPLY(sc, al2) SGS_APPLY(sd, al3) }
                else UB_APPLY4(sa, sb_, sc, sd)
                if (__ballot(Tm > 0.0f) == 0ull) break;
            }
        } else if (KIND == K_TRIP_PK) {
            // candidate: the staged pairs re-laid as (m, c ry) (a k, c) (a, nlo) (r, g) (b, .) so that ONE packed fma gives t0 = m - a k ly and
            // V = c ry - c ly, and one more the red / green sums: 14.5 instead of 16.5 VALU per splat, bit-identical (an fma per component)
            Tm = amax;
            v2f C01 = {C0, C1};
            const v2f lyy = {ly, ly};
#define UB_ALPHA_PK(N, AL)                                                                              \
            float AL;                                                                                   \
            {                                                                                           \
                const v2f pa = {(N##0).x, (N##0).y}, pb = {(N##1).x, (N##1).y};                         \
                const v2f tv = __builtin_elementwise_fma(-pb, lyy, pa);                                 \
                const float U = __builtin_fmaf(-(N##2).x, lx, tv.x);                                    \
                const float q = __builtin_fmaf(U, U, __builtin_fmaf(tv.y, tv.y, (N##2).y));             \
                AL = SGS_SAT(SGS_EXP2(-q)) * SGS_SAT(__builtin_fmaf(-q, big, cq_big));                  \
            }
#define UB_APPLY_PK(N, AL)                                                                              \
            {                                                                                           \
                float wgt = (AL) * Tm;                                                                  \
                const float tt = __builtin_fmaf(-wgt, amax, Tm);                                        \
                const float lv = SGS_SAT(__builtin_fmaf(tt, big, nt_big));                              \
                wgt *= lv; Tm = tt * lv;                                                                \
                const v2f ww = {wgt, wgt}, rg = {(N##3).x, (N##3).y};                                   \
                C01 = __builtin_elementwise_fma(ww, rg, C01); C2 = __builtin_fmaf(wgt, (N##4).x, C2);  \
            }
            for (unsigned kq = 0; kq < cntq; kq += 4) {
                const uint4 pk = *reinterpret_cast<const uint4*>(lst + kq);
                SGS_LOAD(pk.x, sa) SGS_LOAD(pk.y, sb_) SGS_LOAD(pk.z, sc) SGS_LOAD(pk.w, sd)
                UB_ALPHA_PK(sa, al0) UB_ALPHA_PK(sb_, al1) UB_ALPHA_PK(sc, al2) UB_ALPHA_PK(sd, al3)
                UB_APPLY_PK(sa, al0) UB_APPLY_PK(sb_, al1) UB_APPLY_PK(sc, al2) UB_APPLY_PK(sd, al3)
                if (__ballot(Tm > 0.0f) == 0ull) break;
            }
            C0 = C01.x; C1 = C01.y;
        } else if (KIND == K_TRIP_MFMA || KIND == K_TRIP_MFMA_NOLDS) {
            // candidate (round 6): the two LINEAR forms of a splat — U = m - a lx - a k ly and V = c ry - c ly — on the matrix pipe.  One
            // v_mfma_f32_16x16x1_4b_f32 is four 16 x 16 outer products; lane l holds D[block b][row 4 (l / 16) + r][column l % 16] in VGPR 4 b + r,
            // i.e. 16 items (b, r) for ONE pixel when pixel = (group g = l / 16, column j = l % 16) = lane l of the 8 x 8 quadrant (rows 2 g, 2 g + 1).
            // Items = 8 splats x {U, V}; three k-steps (B = 1, column, row-in-group) accumulate  const_g + coef_x col + coef_y h  with the group's
            // row offset folded into the constant by the lane that supplies A (lane 16 b + 4 g' + r: item (b, r) for group g').  Same three
            // products per form as the fma chain of SGS_ALPHA_F, so the same error class; q = U^2 + (V^2 + nlo) stays on the VALU.
            // Staged layout of the candidate: uv[j] = (m, -a, -a k | c ry, 0, -c), cn[j] = (r, g, b, nlo).
            float* const uv = reinterpret_cast<float*>(s_arena);                       // 257 x 6 floats
            float4* const cn = reinterpret_cast<float4*>(uv + 6 * (SGS_BATCH + 2));    // 257 x 4 floats (16-byte aligned: 6 * 258 * 4 = 6192)
            Tm = amax;
            const unsigned bq = (unsigned)lane >> 4, rq = (unsigned)lane & 3u, gq = ((unsigned)lane >> 2) & 3u;
            const unsigned s_mine = 2u * bq + (rq >> 1), kind = rq & 1u;
            const float QX = (float)((wave & 1) * 8), gy = (float)((wave >> 1) * 8) + 2.0f * (float)gq;
            float one = 1.0f, colf = (float)(lane & 7), hf = (float)((lane >> 3) & 1);
            SGS_PIN_VGPR(one); SGS_PIN_VGPR(colf); SGS_PIN_VGPR(hf);
            typedef float v16f __attribute__((ext_vector_type(16)));
            for (unsigned kq = 0; kq < cntq; kq += 8) {
                float r0, r1, r2;
                if (KIND == K_TRIP_MFMA) {
                    const unsigned myo = lst[kq + s_mine] >> 3;                        // my item's splat (per-lane read: 8 distinct addresses)
                    const float* rec = uv + myo * 6u + kind * 3u;
                    r0 = rec[0]; r1 = rec[1]; r2 = rec[2];
                } else { r0 = a[0] + (float)kq; r1 = a[1]; r2 = a[2]; }
                const float c0 = __builtin_fmaf(r2, gy, __builtin_fmaf(r1, QX, r0));
                v16f acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(c0, one, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(r1, colf, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(r2, hf, acc, 0, 0, 0);
                uint4 pa, pb4;
                if (KIND == K_TRIP_MFMA) { pa = *reinterpret_cast<const uint4*>(lst + kq); pb4 = *reinterpret_cast<const uint4*>(lst + kq + 4); }
                else { pa = uint4{0u, 0u, 0u, 0u}; pb4 = pa; }
#define UB_SPLAT_M(OFF, IU)                                                                             \
                {                                                                                       \
                    float4 c4;                                                                          \
                    if (KIND == K_TRIP_MFMA) c4 = SGS_AT(cn, float4, (OFF) << 1);                       \
                    else c4 = make_float4(a[6], a[7], a[8], a[5] + 6.0f);                               \
                    const float U = acc[IU], V = acc[(IU) + 1];                                         \
                    const float q = __builtin_fmaf(U, U, __builtin_fmaf(V, V, c4.w));                   \
                    const float al = SGS_SAT(SGS_EXP2(-q)) * SGS_SAT(__builtin_fmaf(-q, big, cq_big));  \
                    float wgt = al * Tm;                                                                \
                    const float tt = __builtin_fmaf(-wgt, amax, Tm);                                    \
                    const float lv = SGS_SAT(__builtin_fmaf(tt, big, nt_big));                          \
                    wgt *= lv; Tm = tt * lv;                                                            \
                    C0 = __builtin_fmaf(wgt, c4.x, C0); C1 = __builtin_fmaf(wgt, c4.y, C1); C2 = __builtin_fmaf(wgt, c4.z, C2); \
                }
                UB_SPLAT_M(pa.x, 0) UB_SPLAT_M(pa.y, 2) UB_SPLAT_M(pa.z, 4) UB_SPLAT_M(pa.w, 6)
                if (__ballot(Tm > 0.0f) == 0ull) break;
                UB_SPLAT_M(pb4.x, 8) UB_SPLAT_M(pb4.y, 10) UB_SPLAT_M(pb4.z, 12) UB_SPLAT_M(pb4.w, 14)
                if (__ballot(Tm > 0.0f) == 0ull) break;
            }
        } else if (KIND == K_MFMA16) {
            typedef float v16f __attribute__((ext_vector_type(16)));
            v16f acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NI; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[i], b, acc, 0, 0, 0);
            a[0] += acc[0] + acc[5] + acc[10] + acc[15];
        } else if (KIND == K_TRIP_NOLDS) {
            Tm = amax;
            for (unsigned kq = 0; kq < cntq; kq += 4) {
                // the same arithmetic with the splats' values taken from registers (perturbed per trip so nothing folds)
#define UB_FAKE(N, s) const float2 N##0 = make_float2(a[0] + s, a[1]), N##1 = make_float2(a[2], a[3] + s), N##2 = make_float2(a[4], a[5] + 6.0f), N##3 = make_float2(a[6], a[7]), N##4 = make_float2(a[8], a[9]);
                const float Tb = Tm; (void)Tb;
                const float s0 = (float)kq;
                UB_FAKE(sa, s0) UB_FAKE(sb_, s0 + 1.f) UB_FAKE(sc, s0 + 2.f) UB_FAKE(sd, s0 + 3.f)
                SGS_ALPHA_F(0, sa, al0) SGS_ALPHA_F(0, sb_, al1) SGS_ALPHA_F(0, sc, al2) SGS_ALPHA_F(0, sd, al3)
                SGS_APPLY(sa, al0) SGS_APPLY(sb_, al1) SGS_APPLY(sc, al2) SGS_APPLY(sd, al3)
                if (__ballot(Tm > 0.0f) == 0ull) break;
            }
        } else if (KIND == K_TRIP_LDSONLY) {
            for (unsigned kq = 0; kq < cntq; kq += 4) {
                const uint4 pk = *reinterpret_cast<const uint4*>(lst + kq);
                SGS_LOAD(pk.x, sa) SGS_LOAD(pk.y, sb_) SGS_LOAD(pk.z, sc) SGS_LOAD(pk.w, sd)
                asm volatile("" :: "v"(sa0), "v"(sa1), "v"(sa2), "v"(sa3), "v"(sa4), "v"(sb_0), "v"(sb_1), "v"(sb_2), "v"(sb_3), "v"(sb_4));
                asm volatile("" :: "v"(sc0), "v"(sc1), "v"(sc2), "v"(sc3), "v"(sc4), "v"(sd0), "v"(sd1), "v"(sd2), "v"(sd3), "v"(sd4));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (lane == 0) {
        WaveRec r; r.c0 = c0; r.c1 = c1; r.w0 = w0; r.w1 = w1;
        r.hwid = (unsigned)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
        r.xcc = (unsigned)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
        rec[blockIdx.x * 4 + wave] = r;
    }
    float s = acc.x + Tm + C0 + C1 + C2 + Dz + Wsum + (float)used;
    for (int i = 0; i < NI; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + tid] = s;
}

struct Result { double cyc_med, ghz, ipc, ms, cpi_span; int wmin, wmax, simds; };

template <int KIND>
Result run(int wps, int wgs_per_launch_cu, float* out, const float* in, WaveRec* rec_d, double wall_hz) {
    const int blocks = 256 * wps;
    int iters = ITER;
    if (KIND >= K_TRIP && KIND != K_MFMA16) iters = ITER / 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, in, rec_d, iters);       // warm (clock ramp, code fetch)
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, in, rec_d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, in, rec_d, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<WaveRec> r(blocks * 4);
    hipMemcpy(r.data(), rec_d, r.size() * sizeof(WaveRec), hipMemcpyDeviceToHost);
    std::vector<unsigned long long> cyc; double sc = 0, sw = 0;
    struct Simd { unsigned long long c0 = ~0ull, c1 = 0; int n = 0; };
    std::map<unsigned, Simd> per_simd;
    for (auto& x : r) {
        cyc.push_back(x.c1 - x.c0); sc += (double)(x.c1 - x.c0); sw += (double)(x.w1 - x.w0);
        // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
        Simd& S = per_simd[(x.xcc & 15u) << 16 | ((x.hwid >> 4) & 3u) | ((x.hwid >> 8) & 0xffu) << 2];
        S.c0 = std::min(S.c0, x.c0); S.c1 = std::max(S.c1, x.c1); S.n++;
    }
    std::sort(cyc.begin(), cyc.end());
    Result R; R.cyc_med = (double)cyc[cyc.size() / 2]; R.ghz = sc / sw * wall_hz * 1e-9; R.ms = ms;
    R.wmin = 1 << 30; R.wmax = 0; R.simds = (int)per_simd.size();
    const double per_wave = (double)iters * (KIND >= K_TRIP && KIND != K_MFMA16 ? 16.0 : (double)kind_ni[KIND]);
    std::vector<double> cpi;           // per SIMD: cycles from its first wave's start to its last wave's end / instructions it retired in between
    for (auto& kv : per_simd) {
        R.wmin = std::min(R.wmin, kv.second.n); R.wmax = std::max(R.wmax, kv.second.n);
        cpi.push_back((double)(kv.second.c1 - kv.second.c0) / (per_wave * kv.second.n));
    }
    std::sort(cpi.begin(), cpi.end());
    R.cpi_span = cpi[cpi.size() / 2];
    // ... and from the median wave's own duration, as if all `wps` waves of its SIMD ran beside it throughout (an under-estimate when starts are staggered)
    R.ipc = (double)wps * per_wave / R.cyc_med;
    return R;
}

typedef Result (*runfn)(int, int, float*, const float*, WaveRec*, double);
int main(int argc, char** argv) {
    float *out, *in; WaveRec* rec;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&rec, 256 * 8 * 4 * sizeof(WaveRec));
    std::vector<float> h(64, 0.5f);
    hipMalloc(&in, 64 * sizeof(float));
    hipMemcpy(in, h.data(), 64 * sizeof(float), hipMemcpyHostToDevice);
    int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("# %s  CUs %d  wall clock %d kHz  max shader clock %d kHz\n", prop.name, prop.multiProcessorCount, wall_khz, clk_khz);
    printf("# per row: waves/SIMD -> shader cycles (s_memtime) per wave-instruction per SIMD: median over SIMDs of (last wave's end - first wave's start) / instructions retired;\n");
    printf("#          in brackets the same from the median wave's own duration; effective shader clock; [waves per SIMD seen, min-max over the 1024 SIMDs]\n");
    printf("# blend-trip rows: cycles per TRIP (4 splats) per SIMD\n");
    const double wall_hz = (double)wall_khz * 1e3;
    runfn fns[] = {run<0>, run<1>, run<2>, run<3>, run<4>, run<5>, run<6>, run<7>, run<8>, run<9>, run<10>, run<11>, run<12>, run<13>, run<14>, run<15>,
                   run<16>, run<17>, run<18>, run<19>, run<20>, run<21>, run<22>, run<23>, run<24>, run<25>, run<26>, run<27>};
    static_assert(sizeof(fns) / sizeof(fns[0]) == K_COUNT, "kinds");
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    for (int kind = 0; kind < K_COUNT; ++kind) {
        if (only >= 0 && kind != only) continue;
        printf("%-42s", kind_name[kind]);
        for (int wps : {1, 2, 3, 4, 5, 8}) {
            const Result R = fns[kind](wps, 0, out, in, rec, wall_hz);
            printf(" | w%d %7.2f (%.2f) %.2f GHz [%d-%d]", wps, R.cpi_span, 1.0 / R.ipc, R.ghz, R.wmin, R.wmax);
            fflush(stdout);
        }
        printf("\n");
    }
    return 0;
}
