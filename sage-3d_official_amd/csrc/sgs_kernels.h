// sgs_kernels.h — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the 3DGS forward path.
//
// Stage map (BASELINE.json north_star):
//   k_scene_layout   upload-time re-layout of the scene into wave-chunked float4 rows (A5)
//   k_preprocess     S1 SH colour, S2 EWA projection (fp64 geometry), S3 AABB -> tile rect (+ the tighter rect that is
//                    binned); slot == Gaussian index, one 64-bit ballot per chunk is the visibility mask
//   k_bin_count      S4: per-workgroup LDS tile histograms -> one global atomic per touched tile
//   k_tile_scan      S4: exclusive scan of the per-tile counts, longest-queue-first render order
//   k_bin_emit       S4: duplication of each splat into the queues of the tiles it can reach
//   k_tile_render    S5+S6 fused: per-tile MSD bucket partition, lazy rank sort of ~192-record groups (ties -> index),
//                    front-to-back alpha composite of LDS-staged batches
//   k_pack_rgba8     fp32 RGB -> uint8 RGBA (get_rgba()-shaped surface)
//
// None of these has a counterpart in the reference (it has no rasterizer); the
// arithmetic follows SURVEY.md §8(a) rows S1-S6 and is checked against oracle/ by tests/.
#pragma once
#include "sgs_common.h"

namespace sgs {

#define SGS_LOG2E 1.44269504088896341f
#define SGS_SAT0(x) __builtin_amdgcn_fmed3f((x), 0.0f, 1.0f)

// ------------------------------------------------------------------------------------------------
// wave64 helpers
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }
// how many lanes BELOW this one are set in a wave mask: v_mbcnt_lo + v_mbcnt_hi (two instructions; the mask-and-popcount form is six and keeps a
// 64-bit lane mask alive, which the compiler then hoists to the top of every loop nest that uses it)
__device__ __forceinline__ unsigned lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// The XCD this workgroup runs on (HW_REG_XCC_ID, bits 3:0).  Each tile queue has one sub-queue per XCD:
// the records an XCD's workgroups emit into a queue are contiguous, so that XCD's L2 can write-combine
// them (measured: emit write amplification 6x -> 2.7x).  A placement hint only — any value 0..7 is correct.
// (An earlier version also bumped the sub-counters with XCD-local, workgroup-scope L2 atomics; that bought
// nothing measurable and leans on undocumented cross-kernel L2 behaviour, so the counters use ordinary
// device-scope atomics.)
__device__ __forceinline__ unsigned xcc_id() {
    return (unsigned)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & (SGS_XCDS - 1);
}

// Wave-wide scan / reductions: DPP sequences (row shifts inside the 16-lane rows, then the two row broadcasts): pure
// VALU, where __shfl_* would be ds_bpermute — six DEPENDENT round-trips through the LDS pipe per call, and these sit in
// the partition of every tile and in the binning walk of every chunk.  (The CPU test harness emulates update_dpp /
// readlane lane for lane, so this is the one and only code path.)
// one step: combine x with the value `ctrl` selects; lanes whose source does not exist (or whose row is masked
// off) combine with the identity
#define SGS_DPP_STEP(OP, ID, CTRL, ROWMASK)                                                            \
    x = OP(x, (unsigned)__builtin_amdgcn_update_dpp((int)(ID), (int)x, CTRL, ROWMASK, 0xf, false));
#define SGS_DPP_SCAN(OP, ID)                                                                           \
    SGS_DPP_STEP(OP, ID, 0x111, 0xf) SGS_DPP_STEP(OP, ID, 0x112, 0xf)   /* row_shr:1, row_shr:2 */      \
    SGS_DPP_STEP(OP, ID, 0x114, 0xf) SGS_DPP_STEP(OP, ID, 0x118, 0xf)   /* row_shr:4, row_shr:8 */      \
    SGS_DPP_STEP(OP, ID, 0x142, 0xa)                                    /* row_bcast:15 -> rows 1, 3 */ \
    SGS_DPP_STEP(OP, ID, 0x143, 0xc)                                    /* row_bcast:31 -> rows 2, 3 */
__device__ __forceinline__ unsigned sgs_op_add(unsigned a, unsigned b) { return a + b; }
__device__ __forceinline__ unsigned sgs_op_max(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned sgs_op_min(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned wave_incl_scan(unsigned x, int) { SGS_DPP_SCAN(sgs_op_add, 0u) return x; }
__device__ __forceinline__ unsigned wave_max(unsigned x) {
    SGS_DPP_SCAN(sgs_op_max, 0u)
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ unsigned wave_min(unsigned x) {
    SGS_DPP_SCAN(sgs_op_min, 0xffffffffu)
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ unsigned wave_sum(unsigned x) {
    SGS_DPP_SCAN(sgs_op_add, 0u)
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// orders a wave's own LDS writes before its later LDS reads by OTHER lanes of the same wave
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// Upload.  A scene arrives as the caller's AoS fp32 tensors (sgs_scene_upload) or as the PlayCanvas "compressed.ply" payload
// InteriorGS ships (sgs_scene_upload_compressed: 16 bytes per Gaussian + a table per 256-Gaussian chunk + 8-bit SH; README.md:197-243
// of the reference names the format, its bit layout is restated in sage_gs/ply.py) and is laid out on the device:
//   k_mean_bounds -> k_morton_keys -> 8 x (k_radix_count, k_radix_scan, k_radix_scatter) -> k_scene_layout -> k_chunk_bounds
// i.e. the Z-order (Morton) permutation of the means is made HERE, by an LSD radix sort of (63-bit key, index) pairs, and the layout
// kernel gathers through it — dequantising on the way when the source is compressed.  (Rounds 1-3 copied the means to the host,
// sorted 3 M pairs with one thread of std::sort and copied the permutation back: ~0.5 s of a scene load.)

// A compressed scene as the kernels see it.  chunk[c] = 18 floats: min xyz, max xyz, min scale xyz, max scale xyz (log scales), min rgb,
// max rgb (0 / 1 when the file carries no colour range); packed[i] = (position 11-10-11, rotation 2+10-10-10, scale 11-10-11, colour 8-8-8-8);
// sh[i * 3 k_rest ..] = the file's f_rest bytes, channel-major.
struct PackedScene {
    const float* chunk; const uint4* packed; const unsigned char* sh;
    int k_rest;                       // SH coefficients per channel beyond the DC term: 0, 3, 8 or 15
};
struct UnpackedG { float m[3], s[3], q[4], o, dc[3]; };
__device__ __forceinline__ float sgs_unorm(unsigned v, int bits) { return (float)(v & ((1u << bits) - 1u)) / (float)((1u << bits) - 1u); }
__device__ __forceinline__ float sgs_lerp(float u, float lo, float hi) { return lo + u * (hi - lo); }
// the position of Gaussian i alone (Morton keys)
__device__ __forceinline__ void unpack_position(const PackedScene& Z, long long i, float* m) {
    const float* c = Z.chunk + (i >> 8) * 18;
    const unsigned p = Z.packed[i].x;
    m[0] = sgs_lerp(sgs_unorm(p >> 21, 11), c[0], c[3]); m[1] = sgs_lerp(sgs_unorm(p >> 11, 10), c[1], c[4]); m[2] = sgs_lerp(sgs_unorm(p, 11), c[2], c[5]);
}
// everything but the SH rest: the arithmetic of sage_gs/ply.py load_compressed_ply, in fp32
__device__ __forceinline__ void unpack_gaussian(const PackedScene& Z, long long i, UnpackedG& g) {
    const float* c = Z.chunk + (i >> 8) * 18;
    const uint4 p = Z.packed[i];
    g.m[0] = sgs_lerp(sgs_unorm(p.x >> 21, 11), c[0], c[3]); g.m[1] = sgs_lerp(sgs_unorm(p.x >> 11, 10), c[1], c[4]); g.m[2] = sgs_lerp(sgs_unorm(p.x, 11), c[2], c[5]);
    g.s[0] = expf(sgs_lerp(sgs_unorm(p.z >> 21, 11), c[6], c[9])); g.s[1] = expf(sgs_lerp(sgs_unorm(p.z >> 11, 10), c[7], c[10]));
    g.s[2] = expf(sgs_lerp(sgs_unorm(p.z, 11), c[8], c[11]));
    // rotation: the three smallest components at 10 bits each in [-1/sqrt2, 1/sqrt2], the index of the dropped (largest, positive) one on top
    const float r2 = 1.41421356237309505f;
    const float a = (sgs_unorm(p.y >> 20, 10) - 0.5f) * r2, b = (sgs_unorm(p.y >> 10, 10) - 0.5f) * r2, cc = (sgs_unorm(p.y, 10) - 0.5f) * r2;
    const float mx = sqrtf(fmaxf(0.0f, 1.0f - (a * a + b * b + cc * cc)));
    const unsigned which = p.y >> 30;
    g.q[0] = which == 0u ? mx : a;                              // (w, x, y, z)
    g.q[1] = which == 0u ? a : which == 1u ? mx : b;
    g.q[2] = which <= 1u ? b : which == 2u ? mx : cc;
    g.q[3] = which == 3u ? mx : cc;
    const float cr = sgs_unorm(p.w >> 24, 8), cg = sgs_unorm(p.w >> 16, 8), cb = sgs_unorm(p.w >> 8, 8);
    g.o = sgs_unorm(p.w, 8);
    const float k0 = 1.0f / 0.28209479177387814f;               // colour = 0.5 + C0 dc
    g.dc[0] = (sgs_lerp(cr, c[12], c[15]) - 0.5f) * k0; g.dc[1] = (sgs_lerp(cg, c[13], c[16]) - 0.5f) * k0; g.dc[2] = (sgs_lerp(cb, c[14], c[17]) - 0.5f) * k0;
}
// An 8-bit SH coefficient of the compressed payload: the centre of its truncation bin, (v / 256 - 0.5) * 8 + 4 / 256 = v / 32 - 4 + 1 / 64
// (sage_gs/ply.py load_compressed_ply) — every value a multiple of 1 / 64 below 4 in magnitude, so the ONE fma is exact in fp32 and equal,
// bit for bit, to the four-step form: the layout kernel (scenes inflated to fp32 rows) and k_preprocess (scenes that keep the bytes in
// HBM, below) decode to the same floats.
__device__ __forceinline__ float sgs_sh_byte(unsigned v) { return __builtin_fmaf((float)v, 1.0f / 32.0f, -4.0f + 1.0f / 64.0f); }
// The three readings of such a byte (sage_gs.h SGS_SH_DECODE_*), selected per scene (wave-uniform `mode`).  LINEAR255 is evaluated as a
// converter written in JavaScript evaluates it — in DOUBLE, v * (8 / 255) - 4, product and difference rounded separately — and then
// rounded to fp32 (255 -> exactly 4); BIN_CENTRE_ENDS moves the two end codes out to -4 / +4 with saturated fmas (no compare, no select).
__device__ __forceinline__ float sgs_sh_byte_mode(unsigned v, unsigned mode) {
    if (mode == 1u) return (float)__dsub_rn(__dmul_rn((double)v, 8.0 / 255.0), 4.0);
    const float fv = (float)v;
    float x = __builtin_fmaf(fv, 1.0f / 32.0f, -4.0f + 1.0f / 64.0f);
    if (mode == 2u) {
        const float lo = SGS_SAT0(1.0f - fv), hi = SGS_SAT0(fv - 254.0f);          // 1 for v = 0 / v = 255, else 0
        x += (hi - lo) * (1.0f / 64.0f);
    }
    return x;
}
// SH coefficient k (0 .. 3 (k_rest + 1) - 1, Gaussian-major [coefficient][channel] as the renderer takes them) of Gaussian i
__device__ __forceinline__ float unpack_sh(const PackedScene& Z, long long i, const UnpackedG& g, int k) {
    const int coef = k / 3, ch = k - 3 * coef;
    if (coef == 0) return g.dc[ch];
    return sgs_sh_byte((unsigned)Z.sh[i * (3 * Z.k_rest) + ch * Z.k_rest + (coef - 1)]);
}

// floats <-> unsigned keys that order the same way (atomicMin / atomicMax on the bits)
__device__ __forceinline__ unsigned sgs_ordered(float f) { const unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : b | 0x80000000u; }
__device__ __forceinline__ float sgs_unordered(unsigned k) { return __uint_as_float((k & 0x80000000u) ? k & 0x7fffffffu : ~k); }
// bounds of the finite means: out[0..2] = min (ordered keys), out[3..5] = max; initialised to ~0 / 0 by the host
template <bool PACKED>
__global__ __launch_bounds__(256) void k_mean_bounds(long long n, const float* __restrict__ means, const PackedScene Z, unsigned* __restrict__ out) {
    unsigned lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0u, 0u, 0u};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float m[3];
        if (PACKED) unpack_position(Z, i, m); else { m[0] = means[3 * i]; m[1] = means[3 * i + 1]; m[2] = means[3 * i + 2]; }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (fabsf(m[c]) < 3.0e38f) { const unsigned k = sgs_ordered(m[c]); lo[c] = k < lo[c] ? k : lo[c]; hi[c] = k > hi[c] ? k : hi[c]; }   // (NaN / inf: never visible)
    }
    // one set of six device-scope atomics per WORKGROUP: they land on one cache line and the fabric serialises them (~12 ns each;
    // per wave, 4096 waves: 0.29 ms of a scene load)
    __shared__ unsigned s_lo[4][3], s_hi[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned l = wave_min(lo[c]), h = wave_max(hi[c]);
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][c] = l; s_hi[threadIdx.x >> 6][c] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        atomicMin(&out[c], min(min(s_lo[0][c], s_lo[1][c]), min(s_lo[2][c], s_lo[3][c])));
        atomicMax(&out[3 + c], max(max(s_hi[0][c], s_hi[1][c]), max(s_hi[2][c], s_hi[3][c])));
    }
}
__device__ __forceinline__ unsigned long long sgs_spread21(unsigned long long v) {          // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full; v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
// (63-bit Morton code of the mean, index) per Gaussian: 21 bits per axis over the scene's bounds
template <bool PACKED>
__global__ __launch_bounds__(256) void k_morton_keys(long long n, const float* __restrict__ means, const PackedScene Z, const unsigned* __restrict__ bounds,
                                                      unsigned long long* __restrict__ keys, unsigned* __restrict__ idx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float m[3];
    if (PACKED) unpack_position(Z, i, m); else { m[0] = means[3 * i]; m[1] = means[3 * i + 1]; m[2] = means[3 * i + 2]; }
    unsigned long long q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float lo = sgs_unordered(bounds[c]), hi = sgs_unordered(bounds[3 + c]);
        const float inv = hi > lo ? 2097151.0f / (hi - lo) : 0.0f;
        const float u = fabsf(m[c]) < 3.0e38f ? (m[c] - lo) * inv : 0.0f;
        q[c] = (unsigned long long)fminf(2097151.0f, fmaxf(0.0f, u));
    }
    keys[i] = sgs_spread21(q[0]) | sgs_spread21(q[1]) << 1 | sgs_spread21(q[2]) << 2;
    idx[i] = (unsigned)i;
}

// LSD radix sort of (key, index) pairs, 8 bits per pass, stable.  Workgroup b owns the keys [b T, (b + 1) T), T = SGS_RSORT_TILE:
//   k_radix_count    per-workgroup digit histogram  -> hist[digit * B + b]
//   k_radix_scan     exclusive scan of hist (digit major, workgroup minor): where workgroup b's keys of each digit go
//   k_radix_scatter  every wave walks its contiguous quarter of the tile a row (64 keys) at a time; the lanes of a row that share a
//                    digit find each other with 8 ballots, the first of them takes the run's base from the wave's cursor
#define SGS_RSORT_TILE 2048
__global__ __launch_bounds__(256) void k_radix_count(long long n, const unsigned long long* __restrict__ keys, int shift, unsigned nblocks,
                                                      unsigned* __restrict__ hist) {
    __shared__ unsigned s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * SGS_RSORT_TILE;
    for (int r = 0; r < SGS_RSORT_TILE / 256; ++r) {
        const long long i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(unsigned)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = s_h[threadIdx.x];
}
// Workgroup g scans the SGS_RSCAN_SPAN counters [g S, (g + 1) S) in place (four consecutive counters per thread: 16-byte coalesced loads) and
// leaves their sum in bsum[g]; k_radix_scatter adds the sums of the workgroups in front (at most a few hundred values, read by every lane of a
// wave at once).  (One workgroup walking all 256 * B counters, 366 strided ones per thread at 3 M Gaussians, took 0.62 ms per pass: 5 of the 9.4 ms
// of a scene load, profiles/r04ks.)
#define SGS_RSCAN_SPAN 4096
__global__ __launch_bounds__(1024) void k_radix_scan(unsigned total, unsigned* __restrict__ hist, unsigned* __restrict__ bsum) {
    __shared__ unsigned s_w[16];
    const unsigned tid = threadIdx.x, i0 = blockIdx.x * SGS_RSCAN_SPAN + tid * 4u;
    uint4 c = {0u, 0u, 0u, 0u};
    if (i0 + 3u < total) c = *reinterpret_cast<const uint4*>(hist + i0);
    else { if (i0 < total) c.x = hist[i0]; if (i0 + 1u < total) c.y = hist[i0 + 1u]; if (i0 + 2u < total) c.z = hist[i0 + 2u]; }
    const unsigned sum = c.x + c.y + c.z + c.w;
    const unsigned incl = wave_incl_scan(sum, (int)(tid & 63));
    if ((tid & 63) == 63) s_w[tid >> 6] = incl;
    __syncthreads();
    unsigned run = incl - sum, all = 0;
#pragma unroll
    for (unsigned w = 0; w < 16; ++w) { run += w < (tid >> 6) ? s_w[w] : 0u; all += s_w[w]; }
    const uint4 o = {run, run + c.x, run + c.x + c.y, run + c.x + c.y + c.z};
    if (i0 + 3u < total) *reinterpret_cast<uint4*>(hist + i0) = o;
    else { if (i0 < total) hist[i0] = o.x; if (i0 + 1u < total) hist[i0 + 1u] = o.y; if (i0 + 2u < total) hist[i0 + 2u] = o.z; }
    if (tid == 0) bsum[blockIdx.x] = all;
}
__global__ __launch_bounds__(256) void k_radix_scatter(long long n, const unsigned long long* __restrict__ keys_in, const unsigned* __restrict__ idx_in,
                                                        unsigned long long* __restrict__ keys_out, unsigned* __restrict__ idx_out, int shift,
                                                        unsigned nblocks, const unsigned* __restrict__ hist, const unsigned* __restrict__ bsum) {
    __shared__ unsigned s_cur[4][256];               // per wave and digit: where the wave's next key of that digit goes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * SGS_RSORT_TILE + (long long)wave * (SGS_RSORT_TILE / 4);
#pragma unroll
    for (int w = 0; w < 4; ++w) s_cur[w][tid] = 0;
    __syncthreads();
    for (int r = 0; r < SGS_RSORT_TILE / 256; ++r) {                // the wave's own digit counts
        const long long i = base + r * 64 + lane;
        if (i < n) atomicAdd(&s_cur[wave][(unsigned)(keys_in[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {   // digit tid: the workgroup's base (scan) + the counts of the waves in front
        const unsigned at = (unsigned)tid * nblocks + blockIdx.x;
        unsigned run = hist[at];
        for (unsigned g = 0; g < at / SGS_RSCAN_SPAN; ++g) run += bsum[g];       // (k_radix_scan: the spans in front)
#pragma unroll
        for (int w = 0; w < 4; ++w) { const unsigned c = s_cur[w][tid]; s_cur[w][tid] = run; run += c; }
    }
    __syncthreads();
    for (int r = 0; r < SGS_RSORT_TILE / 256; ++r) {
        const long long i = base + r * 64 + lane;
        const bool valid = i < n;
        const unsigned long long k = valid ? keys_in[i] : 0ull;
        const unsigned v = valid ? idx_in[i] : 0u;
        const unsigned d = (unsigned)(k >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(valid && bit);
            peers &= bit ? bal : ~bal;
        }
        const unsigned rank = lanes_below(peers);
        const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
        unsigned dst = 0;
        if (valid && rank == 0) dst = atomicAdd(&s_cur[wave][d], (unsigned)__popcll(peers));
        dst = __shfl(dst, leader);
        if (valid) { keys_out[dst + rank] = k; idx_out[dst + rank] = v; }
    }
}

// Layout: position p of the laid-out scene holds Gaussian perm[p] (identity when perm == nullptr); geom rows (mx,my,mz,opacity)
// (sx,sy,sz,qw) (qx,qy,qz,original index) and the SH floats, 4 per row, zero padded — every per-frame load is then a 1-KiB coalesced row
// (64 lanes x 16 B).  PACKED: the source is the compressed payload, dequantised here (unpack_gaussian) — the scene is never
// materialised as fp32 arrays.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_scene_layout(long long n, int n_sh_floats, int sh_rows,
                                                      const unsigned* __restrict__ perm,
                                                      const float* __restrict__ means,
                                                      const float* __restrict__ scales,
                                                      const float* __restrict__ quats,
                                                      const float* __restrict__ opac,
                                                      const float* __restrict__ sh,
                                                      const PackedScene Z,
                                                      float4* __restrict__ geom,
                                                      float4* __restrict__ shq) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // position in the laid-out scene
    const long long n_pad = ((n + SGS_WAVE - 1) / SGS_WAVE) * SGS_WAVE;
    if (p >= n_pad) return;
    const long long chunk = p >> 6;
    const int lane = (int)(p & 63);
    // the original index travels in the last word of the geometry rows: splats, records and depth ties use it
    const long long i = p < n ? (perm ? (long long)perm[p] : p) : -1;
    float4 g0 = make_float4(0.f, 0.f, -1.0e30f, 0.f), g1 = make_float4(1.f, 1.f, 1.f, 1.f),
           g2 = make_float4(0.f, 0.f, 0.f, 0.f);
    UnpackedG u;
    if (i >= 0) {
        if (PACKED) {
            unpack_gaussian(Z, i, u);
            g0 = make_float4(u.m[0], u.m[1], u.m[2], u.o);
            g1 = make_float4(u.s[0], u.s[1], u.s[2], u.q[0]);
            g2 = make_float4(u.q[1], u.q[2], u.q[3], __uint_as_float((unsigned)i));
        } else {
            g0 = make_float4(means[3 * i], means[3 * i + 1], means[3 * i + 2], opac[i]);
            g1 = make_float4(scales[3 * i], scales[3 * i + 1], scales[3 * i + 2], quats[4 * i]);
            g2 = make_float4(quats[4 * i + 1], quats[4 * i + 2], quats[4 * i + 3], __uint_as_float((unsigned)i));
        }
    }
    geom[(chunk * SGS_GEOM_ROWS + 0) * SGS_WAVE + lane] = g0;
    geom[(chunk * SGS_GEOM_ROWS + 1) * SGS_WAVE + lane] = g1;
    geom[(chunk * SGS_GEOM_ROWS + 2) * SGS_WAVE + lane] = g2;
    if (PACKED) {
        // A compressed scene KEEPS its 8-bit SH in HBM: per Gaussian the byte string [dc r, g, b as fp32 (12 B)] [the 3 k_rest coefficient
        // bytes in the renderer's order, byte 3 (coef - 1) + channel] padded to 16-byte rows — 64 B per Gaussian at degree 3 (4 rows) where
        // the fp32 layout streams 192 B (12 rows) every frame.  k_preprocess dequantises (eval_sh<DEG, true>): the same floats, bit for bit.
        const int nb = 3 * Z.k_rest;
        for (int r = 0; r < sh_rows; ++r) {
            unsigned w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int wi = 4 * r + c;                               // word of the byte string
                if (wi < 3) w[c] = i >= 0 ? __float_as_uint(u.dc[wi]) : 0u;
                else {
                    unsigned x = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int j = 4 * (wi - 3) + b;                 // coefficient byte j = 3 (coef - 1) + channel
                        if (i >= 0 && j < nb) x |= (unsigned)Z.sh[i * nb + (j % 3) * Z.k_rest + j / 3] << (8 * b);
                    }
                    w[c] = x;
                }
            }
            shq[(chunk * sh_rows + r) * SGS_WAVE + lane] = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]));
        }
        return;
    }
    for (int r = 0; r < sh_rows; ++r) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 4 * r + c;
            v[c] = (i >= 0 && k < n_sh_floats) ? sh[i * n_sh_floats + k] : 0.f;
        }
        shq[(chunk * sh_rows + r) * SGS_WAVE + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Upload: a bounding sphere of the means of every 64-Gaussian chunk plus the chunk's largest scale, two float4 per chunk:
// (cx, cy, cz, R) (s_max, 0, 0, 0).  k_preprocess tests it against the frame's band of pixel rows before it loads
// anything else of the chunk (chunk_outside below): with the scene in Z-order a chunk is a patch of a few decimetres, so
// a rank that owns a band of tile rows touches only the chunks that can reach it, and a single GPU skips what lies
// behind the camera or outside the image.  One wave per chunk; runs once per scene.
__global__ __launch_bounds__(256) void k_chunk_bounds(long long n, long long n_chunks, const float4* __restrict__ geom,
                                                      float4* __restrict__ cbound) {
    const int lane = threadIdx.x & 63;
    const long long chunk = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (chunk >= n_chunks) return;
    const long long pos = chunk * SGS_WAVE + lane;
    const bool real = pos < n;                                   // the last chunk is padded
    const float4 g0 = geom[(chunk * SGS_GEOM_ROWS + 0) * SGS_WAVE + lane];
    const float4 g1 = geom[(chunk * SGS_GEOM_ROWS + 1) * SGS_WAVE + lane];
    float lo[3] = {real ? g0.x : 3.0e38f, real ? g0.y : 3.0e38f, real ? g0.z : 3.0e38f};
    float hi[3] = {real ? g0.x : -3.0e38f, real ? g0.y : -3.0e38f, real ? g0.z : -3.0e38f};
    float sm = real ? fmaxf(g1.x, fmaxf(g1.y, g1.z)) : 0.f;
    bool bad = real && !(g0.x == g0.x && g0.y == g0.y && g0.z == g0.z && sm == sm);       // NaN anywhere: never cull this chunk
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], d)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], d)); }
        sm = fmaxf(sm, __shfl_xor(sm, d));
    }
    const float cx = 0.5f * (lo[0] + hi[0]), cy = 0.5f * (lo[1] + hi[1]), cz = 0.5f * (lo[2] + hi[2]);
    const float dx = g0.x - cx, dy = g0.y - cy, dz = g0.z - cz;
    float r = real ? sqrtf(dx * dx + dy * dy + dz * dz) : 0.f;
    for (int d = 32; d >= 1; d >>= 1) r = fmaxf(r, __shfl_xor(r, d));
    r = r * 1.0001f + 1.0e-6f * (fabsf(cx) + fabsf(cy) + fabsf(cz)) + 1.0e-30f;          // padded for its own rounding
    if (__ballot(bad) != 0ull || !(r < 3.0e37f) || !(sm < 3.0e37f)) r = __uint_as_float(0x7f800000u);   // +inf: keep always
    if (lane == 0) {
        cbound[2 * chunk] = make_float4(cx, cy, cz, r);
        cbound[2 * chunk + 1] = make_float4(sm, 0.f, 0.f, 0.f);
    }
}

// The scene's PROBE (sgs_api.hip fine_shift_of): the geometry rows of M Gaussians at even strides through the layout, for the host.
__global__ __launch_bounds__(256) void k_probe_gather(long long n, int M, const float4* __restrict__ geom, float4* __restrict__ out) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= M) return;
    long long p = (long long)(((double)k + 0.5) * (double)n / (double)M);
    p = p < n ? p : n - 1;
    const long long chunk = p >> 6, lane = p & 63;
#pragma unroll
    for (int r = 0; r < SGS_GEOM_ROWS; ++r) out[(size_t)k * SGS_GEOM_ROWS + r] = geom[(chunk * SGS_GEOM_ROWS + r) * SGS_WAVE + lane];
}

// Can ANY Gaussian of a chunk (means inside the sphere (c, R), scales <= s_max) be visible in this frame / band?
// Wave-uniform, fp64, conservative with respect to the per-Gaussian exclusion in preprocess_chunk: that one keeps a
// Gaussian only if  tz > near,  px + rb + ex >= 1,  px - rb - ex < 16 gx,  py + rb + ey >= cull_y0 + 1,  py - rb - ey < cull_y1
// with  px = fx tx / tz + cx - 0.5,  rb = (3 sqrt(2 lmax + 0.3163) + 1) 1.001 + 0.5,  lmax = (fmax / tz)^2 k smax^2 + dilation,
// k = 2 + lx^2 + ly^2.  sqrt(a + b) <= sqrt(a) + sqrt(b) gives  rb <= A smax / tz + c0  with A = 1.001 * 3 sqrt(2 k) fmax and
// c0 = 1.001 (3 sqrt(2 dilation + 0.3163) + 1) + 0.5, so each condition, multiplied by tz > 0, is implied false for the whole
// sphere when a LINEAR form  n . t + A s_max  is negative at the sphere's most favourable point  n . c + |n| R  (t = view-space
// position).  The comparison carries a relative slack of 1e-4 of the magnitudes involved (ex, ey <= 1e-5 |p| + 0.01 and the
// rounding of the evaluation itself) and one extra pixel.  A non-finite radius (NaN means, see k_chunk_bounds) keeps the chunk.
__device__ __forceinline__ bool chunk_outside(const FrameParams& P, const float4 b0, const float4 b1) {
    const double R = (double)b0.w, sm = (double)b1.x;
    if (!(R < 1.0e37)) return false;
    const double mx = b0.x, my = b0.y, mz = b0.z;
    const double tx = (double)P.view[0] * mx + (double)P.view[1] * my + (double)P.view[2] * mz + (double)P.view[3];
    const double ty = (double)P.view[4] * mx + (double)P.view[5] * my + (double)P.view[6] * mz + (double)P.view[7];
    const double tz = (double)P.view[8] * mx + (double)P.view[9] * my + (double)P.view[10] * mz + (double)P.view[11];
    const double Rp = R * 1.0001 + 1.0e-6 * (fabs(tx) + fabs(ty) + fabs(tz));      // the view transform of the centre rounds too
    if (tz + Rp < (double)P.near_z || tz - Rp > (double)P.far_z) return true;
    // the four planes: form(u) = f u + off tz + nrm Rp + A s_max (f = +-fx for left/right with u = tx, +-fy for top/bottom
    // with u = ty; off, nrm, A: constants of the frame, fill_params); negative => the condition fails for every member
    const double d = P.cull_A * sm, zmag = fabs(tz) + Rp;
    const double fx = (double)P.fx, fy = (double)P.fy;
#define SGS_PLANE_NEG(U, F, K)                                                                           \
    (((F) * (U) + P.cull_off[K] * tz + P.cull_nrm[K] * Rp + d) <                                         \
     -1.0e-4 * (fabs((F) * (U)) + fabs(P.cull_off[K]) * zmag + P.cull_nrm[K] * Rp + d))
    const bool out = SGS_PLANE_NEG(tx, fx, 0) || SGS_PLANE_NEG(tx, -fx, 1) || SGS_PLANE_NEG(ty, fy, 2) || SGS_PLANE_NEG(ty, -fy, 3);
#undef SGS_PLANE_NEG
    return out;
}

// The frame's LIVE LIST: one lane per 64-Gaussian chunk tests the chunk's bounds against the frame / the band of tile
// rows (chunk_outside) and the survivors' indices are compacted into live_list (ballot + one atomic per workgroup; order
// inside a workgroup preserved).  k_preprocess and both binning kernels then walk that list — 47 k chunks at 3 M
// Gaussians, of which a rank's band of a sharded frame keeps a few per cent — instead of sweeping the scene.  Skipped
// chunks get an empty visibility mask here (bigmask all-ones marks "skipped by its bounds" for the tests).
#define SGS_CULL_THREADS 256
// (behind the live list of a group's first frame: the group's WORK LIST, k_chunk_cull_group)
__device__ __forceinline__ unsigned* sgs_work_list(unsigned* live_list, long long n_chunks) { return live_list + ((n_chunks + 1) & ~1ll); }
__global__ __launch_bounds__(SGS_CULL_THREADS) void k_chunk_cull(const FrameGroup G) {
    const FrameSlot& S = G.s[blockIdx.y];
    const FrameParams& P = S.P;
    __shared__ unsigned s_wcnt[SGS_CULL_THREADS / SGS_WAVE];
    __shared__ unsigned s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long chunk = (long long)blockIdx.x * SGS_CULL_THREADS + tid;
    bool live = false;
    if (chunk < P.n_chunks) {
        live = (P.flags & 64u) != 0u || !chunk_outside(P, G.cbound[2 * chunk], G.cbound[2 * chunk + 1]);
        if (!live) { S.vismask[chunk] = 0ull; S.bigmask[chunk] = ~0ull; }
    }
    const unsigned long long m = __ballot(live);
    if (lane == 0) s_wcnt[wave] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SGS_CULL_THREADS / SGS_WAVE; ++w) { before += w < wave ? s_wcnt[w] : 0u; total += s_wcnt[w]; }
    if (tid == 0) s_base = total ? atomicAdd(&S.st->n_live, total) : 0u;
    __syncthreads();
    if (live) S.live_list[s_base + before + lanes_below(m)] = (unsigned)chunk;
}

// The same for the nf full frames of a GROUP, whose projection shares its reads of the scene (k_preprocess_shared): one lane per chunk tests it
// against every frame of the group — every frame's live list as above — and the survivors of ANY frame go, chunk-major, into the group's WORK
// LIST of (chunk << 3 | frame) pairs behind the first frame's live list (FrameStatus.n_work of that frame counts them): the frames that want a
// chunk are neighbours in it.
__global__ __launch_bounds__(SGS_CULL_THREADS) void k_chunk_cull_group(const FrameGroup G, const unsigned nf) {
    __shared__ unsigned s_wcnt[SGS_MAX_GROUP + 1][SGS_CULL_THREADS / SGS_WAVE];
    __shared__ unsigned s_base[SGS_MAX_GROUP + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long chunk = (long long)blockIdx.x * SGS_CULL_THREADS + tid;
    const long long n_chunks = G.s[0].P.n_chunks;
    unsigned lm = 0u;                                   // the frames this chunk is live in
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (chunk < n_chunks) { b0 = G.cbound[2 * chunk]; b1 = G.cbound[2 * chunk + 1]; }
    for (unsigned f = 0; f < nf; ++f) {
        const FrameSlot& S = G.s[f];
        const FrameParams& P = S.P;
        bool live = false;
        if (chunk < n_chunks) {
            live = (P.flags & 64u) != 0u || !chunk_outside(P, b0, b1);
            if (!live) { S.vismask[chunk] = 0ull; S.bigmask[chunk] = ~0ull; }
        }
        lm |= live ? 1u << f : 0u;
        const unsigned long long m = __ballot(live);
        if (lane == 0) s_wcnt[f][wave] = (unsigned)__popcll(m);
    }
    const unsigned mine = (unsigned)__builtin_popcount(lm);
    const unsigned incl = wave_incl_scan(mine, lane);
    if (lane == 63) s_wcnt[nf][wave] = incl;
    __syncthreads();
    if (tid <= (int)nf) {          // one atomic per frame's list and one for the work list
        unsigned total = 0;
#pragma unroll
        for (int w = 0; w < SGS_CULL_THREADS / SGS_WAVE; ++w) total += s_wcnt[tid][w];
        s_base[tid] = total ? atomicAdd(tid < (int)nf ? &G.s[tid].st->n_live : &G.s[0].st->n_work, total) : 0u;
    }
    __syncthreads();
    for (unsigned f = 0; f < nf; ++f) {
        const bool live = (lm >> f) & 1u;
        const unsigned long long m = __ballot(live);
        unsigned before = 0;
#pragma unroll
        for (int w = 0; w < SGS_CULL_THREADS / SGS_WAVE; ++w) before += w < wave ? s_wcnt[f][w] : 0u;
        if (live) G.s[f].live_list[s_base[f] + before + lanes_below(m)] = (unsigned)chunk;
    }
    {
        unsigned before = incl - mine;
#pragma unroll
        for (int w = 0; w < SGS_CULL_THREADS / SGS_WAVE; ++w) before += w < wave ? s_wcnt[nf][w] : 0u;
        unsigned* work = sgs_work_list(G.s[0].live_list, n_chunks) + s_base[nf] + before;
        for (unsigned f = 0, k = 0; f < nf; ++f)
            if ((lm >> f) & 1u) work[k++] = ((unsigned)chunk << 3) | f;
    }
}

// ------------------------------------------------------------------------------------------------
// S1: SH colour.  `row0` points at this lane's float4 in row 0 of its chunk; rows are 64 float4 apart.
// PACKED: the rows hold a compressed scene's byte string (k_scene_layout<true>): 12 B of fp32 DC, then one byte per coefficient.
template <int DEG, bool PACKED>
__device__ __forceinline__ void eval_sh(const float4* __restrict__ row0, float x, float y, float z,
                                        float& out_r, float& out_g, float& out_b, unsigned mode = 0u) {
    constexpr int NF = 3 * (DEG + 1) * (DEG + 1);
    constexpr int ROWS = PACKED ? (12 + (NF - 3) + 15) / 16 : (NF + 3) / 4;
    float c[PACKED ? NF + 4 : ROWS * 4];
    if (PACKED) {
        unsigned w[ROWS * 4];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const float4 v = row0[r * SGS_WAVE];
            w[4 * r] = __float_as_uint(v.x); w[4 * r + 1] = __float_as_uint(v.y); w[4 * r + 2] = __float_as_uint(v.z); w[4 * r + 3] = __float_as_uint(v.w);
        }
        c[0] = __uint_as_float(w[0]); c[1] = __uint_as_float(w[1]); c[2] = __uint_as_float(w[2]);
        if (mode == 0u) {
#pragma unroll
            for (int j = 0; j < NF - 3; ++j) c[3 + j] = sgs_sh_byte((w[3 + (j >> 2)] >> (8 * (j & 3))) & 0xffu);     // (v_cvt_f32_ubyteN + one fma)
        } else {
#pragma unroll
            for (int j = 0; j < NF - 3; ++j) c[3 + j] = sgs_sh_byte_mode((w[3 + (j >> 2)] >> (8 * (j & 3))) & 0xffu, mode);
        }
    } else {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const float4 v = row0[r * SGS_WAVE];
            c[4 * r] = v.x; c[4 * r + 1] = v.y; c[4 * r + 2] = v.z; c[4 * r + 3] = v.w;
        }
    }
    float res[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
#define SGS_S(k) c[3 * (k) + ch]
        float r = 0.28209479177387814f * SGS_S(0);
        if (DEG >= 1) {
            r = r - 0.4886025119029199f * y * SGS_S(1) + 0.4886025119029199f * z * SGS_S(2)
                  - 0.4886025119029199f * x * SGS_S(3);
        }
        if (DEG >= 2) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            r = r + 1.0925484305920792f * xy * SGS_S(4) - 1.0925484305920792f * yz * SGS_S(5)
                  + 0.31539156525252005f * (2.0f * zz - xx - yy) * SGS_S(6)
                  - 1.0925484305920792f * xz * SGS_S(7) + 0.5462742152960396f * (xx - yy) * SGS_S(8);
            if (DEG >= 3) {
                r = r - 0.5900435899266435f * y * (3.0f * xx - yy) * SGS_S(9)
                      + 2.890611442640554f * xy * z * SGS_S(10)
                      - 0.4570457994644658f * y * (4.0f * zz - xx - yy) * SGS_S(11)
                      + 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SGS_S(12)
                      - 0.4570457994644658f * x * (4.0f * zz - xx - yy) * SGS_S(13)
                      + 1.445305721320277f * z * (xx - yy) * SGS_S(14)
                      - 0.5900435899266435f * x * (xx - 3.0f * yy) * SGS_S(15);
            }
        }
#undef SGS_S
        r += 0.5f;
        res[ch] = r < 0.f ? 0.f : r;
    }
    out_r = res[0]; out_g = res[1]; out_b = res[2];
}

// 1 / sqrt(x) in fp64 from v_rsq_f64 and two Newton steps (~1 ulp; 9 instructions where sqrt followed by a division expands to ~45).
// Used where the argument is a squared length of well-scaled values (a quaternion's norm, a view vector): 0, inf and NaN give
// NaN products downstream exactly as 1.0 / sqrt(x) did (0 * inf there, 0 * NaN here).
__device__ __forceinline__ double rsqrt64(double x) {
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = fma(-(x * y), y, 1.0);      // 1 - x y^2
        y = fma(0.5 * y, e, y);
    }
    return y;
}

// floor(v) clamped to [lo, hi]; NaN -> lo.  Mirrors clampi() of oracle/sgs_oracle.c.
__device__ __forceinline__ int tile_clamp(double v, int lo, int hi) {
    const double f = floor(v);
    if (!(f > (double)lo)) return lo;
    if (f > (double)hi) return hi;
    return (int)f;
}

// Fine tiles (sgs_common.h): the shift z of the frame; every row / tile index of FrameParams counts cells of (16 >> z)^2 pixels.
__device__ __forceinline__ int fine_shift(const FrameParams& P) { return (int)((P.flags >> SGS_PFLAG_FINE_SHIFT) & 3u); }
// The first row this call owns that is not above frame tile row f (clamped to [row_begin, row_end]): frame rows [a, b)
// are the owned rows [owned_row(a), owned_row(b)).  Owned 16-pixel row k is frame 16-pixel row k * row_stride + row_phase; with fine
// tiles a 16-pixel row is 2^z rows of cells, owned or not as a whole: owned row (k << z) + s is frame row ((k stride + phase) << z) + s.
__device__ __forceinline__ int owned_row(const FrameParams& P, int f) {
    int k = f;
    if (P.row_stride > 1) {
        const int z = fine_shift(P), F = f >> z;
        const int k16 = F > P.row_phase ? (F - P.row_phase + P.row_stride - 1) / P.row_stride : 0;
        k = (k16 << z) + (k16 * P.row_stride + P.row_phase == F ? f - (F << z) : 0);
    }
    return min(max(k, P.row_begin), P.row_end);
}
// ... and back: the frame's row of cells that owned row r of this call is
__device__ __forceinline__ unsigned frame_row_of(const FrameParams& P, unsigned r) {
    if (P.row_stride <= 1) return r;
    const unsigned z = (unsigned)fine_shift(P), k16 = r >> z;
    return ((k16 * (unsigned)P.row_stride + (unsigned)P.row_phase) << z) + (r - (k16 << z));
}

// One wave's worth of S1-S3: 64 Gaussians of chunk `chunk`.  FINE: the frame may be rendered through fine tiles (sgs_common.h; z from the
// frame's flags) — an instantiation of its own, so that the ordinary frame's kernel carries none of it (the shifts and scalings cost the
// HBM-bound 1080p projection 2 us of 64 when they were unconditional, r06t).
template <bool FINE>
__device__ __forceinline__ void preprocess_chunk(const FrameParams& P, const float4* __restrict__ geom,
                                                 const float4* __restrict__ shq, Splat* __restrict__ splats,
                                                 unsigned long long* __restrict__ vismask,
                                                 unsigned long long* __restrict__ bigmask, unsigned* __restrict__ big_list,
                                                 uint4* __restrict__ binrec,
                                                 FrameStatus* __restrict__ st, long long chunk, int lane) {
    const long long pos = chunk * SGS_WAVE + lane;      // position in the (Z-ordered) scene layout
    const int zf = FINE ? fine_shift(P) : 0;            // (uniform; the compiler folds every use away when !FINE)

    // all three geometry rows of the chunk are requested at once (1-KiB rows; a live chunk nearly always has lanes that need
    // the second and third): loading each one only behind the test that needs it made three dependent trips to HBM
    const float4 g0 = geom[(chunk * SGS_GEOM_ROWS + 0) * SGS_WAVE + lane];
    const float4 g1 = geom[(chunk * SGS_GEOM_ROWS + 1) * SGS_WAVE + lane];
    const float4 g2 = geom[(chunk * SGS_GEOM_ROWS + 2) * SGS_WAVE + lane];
    const double mx = g0.x, my = g0.y, mz = g0.z;
    const double tx = (double)P.view[0] * mx + (double)P.view[1] * my + (double)P.view[2] * mz + (double)P.view[3];
    const double ty = (double)P.view[4] * mx + (double)P.view[5] * my + (double)P.view[6] * mz + (double)P.view[7];
    const double tz = (double)P.view[8] * mx + (double)P.view[9] * my + (double)P.view[10] * mz + (double)P.view[11];
    const bool front = pos < P.n && tz > (double)P.near_z && tz <= (double)P.far_z;   // pos >= n: padding of the last chunk

    bool vis = false, big = false;
    unsigned rect01 = 0, rect23 = 0, slot = 0;         // slot = the Gaussian's ORIGINAL index
    unsigned brect01 = 0, brect23 = 0;                 // the rect the binning kernels walk (see below)
    int bnt = 0;
    float sx = 0.f, sy = 0.f, ca = 0.f, cb = 0.f, cc = 0.f;
    float ext_x = 3.0e38f, ext_y = 3.0e38f, qmax = -1.0f;   // extents of {alpha >= alpha_min} for the composite; log2(o / alpha_min)
    // Cheap exclusion before the fp64 work (17-64 % of the chunks in front of the near plane end with no visible
    // lane, and visibility is dense inside a chunk, so whole waves skip): an fp32 UPPER bound of the 3-sigma radius,
    //     lambda_max(J W Sigma W^T J^T) <= |J|_F^2 s_max^2,  |J|_F^2 <= (f_max / tz)^2 (2 + limx^2 + limy^2),
    //     lam = mid + sqrt(max(0.1, mid^2 - det)) <= 2 (lambda_max + dilation) + 0.3163,  radius <= 3 sqrt(lam) + 1,
    // padded for fp32 rounding.  A Gaussian whose centre +- that bound misses the image band has an empty tile rect
    // in the exact computation too, so dropping it here changes nothing (the view matrix is rigid by contract).
    bool maybe = front;
    if (front) {
        const float smax = fmaxf(g1.x, fmaxf(g1.y, g1.z));
        const float inv = __builtin_amdgcn_rcpf((float)tz);                      // (v_rcp_f32, 1 ulp: inside the 1.001 and the ex / ey below)
        const float lx = (float)P.limx, ly = (float)P.limy;
        const float jf = fmaxf(P.fx, P.fy) * inv;
        const float lmax = jf * jf * (2.0f + lx * lx + ly * ly) * (smax * smax) + P.dilation;
        const float rb = (3.0f * sqrtf(2.0f * lmax + 0.3163f) + 1.0f) * 1.001f + 0.5f;
        const float pxf = P.fx * (float)tx * inv + P.cx - 0.5f, pyf = P.fy * (float)ty * inv + P.cy - 0.5f;
        const float ex = 1.0e-5f * fabsf(pxf) + 0.01f, ey = 1.0e-5f * fabsf(pyf) + 0.01f;
        const bool out_x = pxf + rb + ex < 1.0f || pxf - rb - ex >= (float)(SGS_TILE_PX * ((P.gx + (1 << zf) - 1) >> zf));   // (16 x the frame's 16-pixel tiles)
        const bool out_y = pyf + rb + ey < (float)P.cull_y0 + 1.0f || pyf - rb - ey >= (float)P.cull_y1;
        maybe = !(out_x || out_y);                   // (NaN anywhere keeps the Gaussian)
    }
    if (maybe) {
        slot = __float_as_uint(g2.w);
        // S2: Sigma = R S S^T R^T
        const double qw0 = g1.w, qx0 = g2.x, qy0 = g2.y, qz0 = g2.z;
        // (an fp64 division expands to ~35 instructions: one reciprocal per denominator, then multiplies)
        const double iqn = rsqrt64(qw0 * qw0 + qx0 * qx0 + qy0 * qy0 + qz0 * qz0);
        const double w = qw0 * iqn, x = qx0 * iqn, y = qy0 * iqn, z = qz0 * iqn;
        const double s0 = g1.x, s1 = g1.y, s2 = g1.z;
        const double M00 = (1 - 2 * (y * y + z * z)) * s0, M01 = (2 * (x * y - w * z)) * s1, M02 = (2 * (x * z + w * y)) * s2;
        const double M10 = (2 * (x * y + w * z)) * s0, M11 = (1 - 2 * (x * x + z * z)) * s1, M12 = (2 * (y * z - w * x)) * s2;
        const double M20 = (2 * (x * z - w * y)) * s0, M21 = (2 * (y * z + w * x)) * s1, M22 = (1 - 2 * (x * x + y * y)) * s2;
        const double S00 = M00 * M00 + M01 * M01 + M02 * M02;
        const double S01 = M00 * M10 + M01 * M11 + M02 * M12;
        const double S02 = M00 * M20 + M01 * M21 + M02 * M22;
        const double S11 = M10 * M10 + M11 * M11 + M12 * M12;
        const double S12 = M10 * M20 + M11 * M21 + M12 * M22;
        const double S22 = M20 * M20 + M21 * M21 + M22 * M22;
        // S2: EWA projection, cov' = J W Sigma W^T J^T + dilation I
        const double fx = P.fx, fy = P.fy;
        const double limx = P.limx, limy = P.limy;
        const double itz = 1.0 / tz;
        const double xz = tx * itz, yz = ty * itz;
        const double txc = fmin(limx, fmax(-limx, xz)) * tz;
        const double tyc = fmin(limy, fmax(-limy, yz)) * tz;
        const double j00 = fx * itz, j02 = -fx * txc * (itz * itz);
        const double j11 = fy * itz, j12 = -fy * tyc * (itz * itz);
        const double T00 = j00 * (double)P.view[0] + j02 * (double)P.view[8];
        const double T01 = j00 * (double)P.view[1] + j02 * (double)P.view[9];
        const double T02 = j00 * (double)P.view[2] + j02 * (double)P.view[10];
        const double T10 = j11 * (double)P.view[4] + j12 * (double)P.view[8];
        const double T11 = j11 * (double)P.view[5] + j12 * (double)P.view[9];
        const double T12 = j11 * (double)P.view[6] + j12 * (double)P.view[10];
        const double U00 = T00 * S00 + T01 * S01 + T02 * S02;
        const double U01 = T00 * S01 + T01 * S11 + T02 * S12;
        const double U02 = T00 * S02 + T01 * S12 + T02 * S22;
        const double U10 = T10 * S00 + T11 * S01 + T12 * S02;
        const double U11 = T10 * S01 + T11 * S11 + T12 * S12;
        const double U12 = T10 * S02 + T11 * S12 + T12 * S22;
        const double a = U00 * T00 + U01 * T01 + U02 * T02 + (double)P.dilation;
        const double b = U00 * T10 + U01 * T11 + U02 * T12;
        const double c = U10 * T10 + U11 * T11 + U12 * T12 + (double)P.dilation;
        const double det = a * c - b * b;
        if (det > 0.0) {
            // S3: 3-sigma radius, pixel AABB -> tile rect (clipped to this call's tile rows)
            const double mid = 0.5 * (a + c);
            const double lam = mid + sqrt(fmax(0.1, mid * mid - det));
            const double radius = ceil(3.0 * sqrt(lam));
            const double px = fx * xz + (double)P.cx - 0.5;
            const double py = fy * yz + (double)P.cy - 0.5;
            // (S3's rect is a rect of 16x16-PIXEL tiles whatever the frame's cells are: with fine tiles — z > 0, sgs_common.h — it is
            //  computed on the frame's 16-pixel grid and then mapped to the cells it covers, 2^z per tile and axis, clipped to the grid of cells)
            const int zr = (1 << zf) - 1;
            const int gx16 = (P.gx + zr) >> zf, gy16 = (P.gy + zr) >> zf;
            const int x0 = min(tile_clamp((px - radius) / SGS_TILE_PX, 0, gx16) << zf, P.gx);
            const int x1 = min(tile_clamp((px + radius + (SGS_TILE_PX - 1)) / SGS_TILE_PX, 0, gx16) << zf, P.gx);
            // frame rows [fy0, fy1) -> the rows this call owns: the contiguous band, or every row_stride-th row from
            // row_phase (then a rect's rows are again a contiguous range of OWNED rows, so binning never knows)
            const int fy0 = min(tile_clamp((py - radius) / SGS_TILE_PX, 0, gy16) << zf, P.gy);
            const int fy1 = min(tile_clamp((py + radius + (SGS_TILE_PX - 1)) / SGS_TILE_PX, 0, gy16) << zf, P.gy);
            const int y0 = owned_row(P, fy0), y1 = owned_row(P, fy1);
            const int nt = (x1 - x0) * (y1 - y0);
            if (nt > 0) {
                vis = true;
                rect01 = (unsigned)x0 | ((unsigned)y0 << 16);
                rect23 = (unsigned)x1 | ((unsigned)y1 << 16);
                // What gets BINNED is tighter than S3's square of side 2 ceil(3 sqrt(lambda_max)): a pixel can only
                // pass alpha >= alpha_min inside the ellipse d^T Sigma'^-1 d <= K = 2 ln(o / alpha_min), whose
                // axis-aligned extent is sqrt(K a) x sqrt(K c) — 30 % fewer records on the indoor scenes (anisotropic
                // and low-opacity splats), none of which any pixel could have used.  Tile t holds the pixel centres
                // 16t .. 16t+15.  fp32 with outward padding; the reference rect stays in the splat record and is what
                // SGS_FLAG_LOOSE_CULL (tests) bins.
                brect01 = rect01; brect23 = rect23;
                // The conic AS THE COMPOSITE SEES IT: rounded to fp32 (the format of the path, and what the oracle blends with).
                // The extents below — of the ellipse alpha >= alpha_min — are derived from THAT form, not from the fp64
                // covariance it was inverted from: for a needle (sigma_minor ~ 0.55 px from the dilation, sigma_major thousands
                // of pixels) the rounding of three nearly dependent entries moves the determinant, i.e. the length of the long
                // axis, by per cent — the composite blended pixels (alpha 1.01 alpha_min, 2 500 px from the centre) that the
                // covariance's ellipse, and with it the bin rect, ended 5 px short of (gpu_fuzz seed 3002, r03).
                //   q2 = A dx^2 + B dx dy + C dy^2 <= qmax:  |dx| <= sqrt(qmax C / det),  |dy| <= sqrt(qmax A / det),  det = A C - B^2 / 4
                // (log2 units, as in the record; det <= 0: a degenerate form, every pixel may pass: no tightening).
                const double idet = 1.0 / det;
                ca = (float)(c * idet); cb = (float)(-b * idet); cc = (float)(a * idet);
                const double A64e = (0.5 * 1.4426950408889634) * (double)ca, B64e = 1.4426950408889634 * (double)cb,
                             C64e = (0.5 * 1.4426950408889634) * (double)cc;
                const double det_e = A64e * C64e - 0.25 * B64e * B64e;
                const double idet_e = 1.0 / det_e;                                               // (one reciprocal: the extents are padded below)
                const float wx = det_e > 0.0 ? (float)(C64e * idet_e) : 3.0e38f, wy = det_e > 0.0 ? (float)(A64e * idet_e) : 3.0e38f;   // extent^2 per unit of q2
                // what the composite needs to decide which 8x8 quadrants of a tile the splat can reach (k_tile_render):
                // alpha >= alpha_min  <=>  q2 <= qmax = log2(o / alpha_min); half extents padded generously (the exact quadrant test follows)
                qmax = __log2f(g0.w) - __log2f(P.alpha_min);
                if (qmax > 0.0f) {
                    const float ex_ = sqrtf(qmax * wx) * 1.01f + 0.5f, ey_ = sqrtf(qmax * wy) * 1.01f + 0.5f;
                    ext_x = ex_ < 3.0e38f ? ex_ : 3.0e38f; ext_y = ey_ < 3.0e38f ? ey_ : 3.0e38f;      // (inf / NaN -> everywhere)
                }
                if (!(P.flags & 32u)) {
                    const float Kc = qmax * 1.0001f + 1.0e-4f;
                    if (Kc > 0.0f) {
                        const float hx_ = sqrtf(Kc * wx) * 1.0001f + 0.02f, hy_ = sqrtf(Kc * wy) * 1.0001f + 0.02f;
                        const float hx = hx_ < 1.0e9f ? hx_ : 1.0e9f, hy = hy_ < 1.0e9f ? hy_ : 1.0e9f;          // (inf / NaN -> the whole rect)
                        const float fpx = (float)px, fpy = (float)py;
                        const float epx = 1.0e-6f * fabsf(fpx), epy = 1.0e-6f * fabsf(fpy);
                        // (a cell holds the pixel centres cp t .. cp t + cp - 1, cp = 16 >> z: a power of two, the products are exact)
                        const float cpl = (float)((SGS_TILE_PX >> zf) - 1), icp = (float)(1 << zf) * (1.0f / SGS_TILE_PX);
                        const float lo_x = ceilf((fpx - hx - epx - cpl) * icp), hi_x = floorf((fpx + hx + epx) * icp) + 1.0f;
                        const float lo_y = ceilf((fpy - hy - epy - cpl) * icp), hi_y = floorf((fpy + hy + epy) * icp) + 1.0f;
                        const int bx0 = max(x0, (int)fmaxf(lo_x, -1.0e6f)), bx1 = min(x1, (int)fminf(hi_x, 1.0e6f));
                        const int by0 = owned_row(P, max(fy0, (int)fmaxf(lo_y, -1.0e6f))), by1 = owned_row(P, min(fy1, (int)fminf(hi_y, 1.0e6f)));
                        if (bx1 > bx0 && by1 > by0) {
                            brect01 = (unsigned)bx0 | ((unsigned)by0 << 16); brect23 = (unsigned)bx1 | ((unsigned)by1 << 16);
                            bnt = (bx1 - bx0) * (by1 - by0);
                        } else { brect01 = 0u; brect23 = 0u; bnt = 0; }     // reaches no pixel: nothing to bin
                    } else { brect01 = 0u; brect23 = 0u; bnt = 0; }         // opacity below alpha_min: never blended
                } else bnt = nt;
                if (bnt > 0) {   // what the binning kernels walk is the rect in SUPER-TILES (level 1 of the binning)
                    const unsigned sx0 = (brect01 & 0xffffu) >> SGS_ST_SHIFT, sx1 = ((brect23 & 0xffffu) + SGS_ST - 1) >> SGS_ST_SHIFT;
                    const unsigned sy0 = (brect01 >> 16) >> SGS_ST_SHIFT, sy1 = ((brect23 >> 16) + SGS_ST - 1) >> SGS_ST_SHIFT;
                    big = (sx1 - sx0) * (sy1 - sy0) > (unsigned)SGS_BIG_RECT;
                }
                sx = (float)px; sy = (float)py;
            }
        }
    }

    // No compaction: a splat lives at its Gaussian's ORIGINAL index (slot), and the wave's ballot is the
    // chunk's visibility mask.  Ties on depth can then break on the slot number itself.
    const unsigned long long vmask = __ballot(vis);
    // A splat whose rect covers hundreds of tiles (a near-camera Gaussian) would keep ONE wave of the
    // binning kernels busy for its whole expansion; those few go to a global list instead and are
    // expanded by entire workgroups.  bigmask tells the per-chunk walk to skip them.
    {   // one atomic per wave (the big splats of a near-camera surface come in runs: up to 64 per chunk)
        const unsigned long long wants = __ballot(big);
        if (wants != 0ull) {
            const int leader = __ffsll((long long)wants) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&st->n_big, (unsigned)__popcll(wants));
            base = __shfl(base, leader);
            if (big) {
                const unsigned k = base + lanes_below(wants);
                if (k < SGS_BIG_CAP) big_list[k] = (unsigned)pos; else big = false;
            }
        }
    }
    const unsigned long long bmask = __ballot(big);
    if (lane == 0) { vismask[chunk] = vmask; bigmask[chunk] = bmask; }

    if (vis) {
        // S1: view direction in model space (fp64 difference, fp32 polynomial)
        const double dx = mx - P.campos[0], dy = my - P.campos[1], dz = mz - P.campos[2];
        const double idn = rsqrt64(dx * dx + dy * dy + dz * dz);
        const float ux = (float)(dx * idn), uy = (float)(dy * idn), uz = (float)(dz * idn);
        const float4* row0 = shq + (chunk * P.sh_rows) * SGS_WAVE + lane;
        float r, g, b;
        if (P.flags & SGS_PFLAG_SH_PACKED) {         // a compressed scene: 8-bit coefficients, dequantised here (wave-uniform branch)
            const unsigned shm = (P.flags >> SGS_PFLAG_SH_MODE_SHIFT) & 3u;
            switch (P.sh_degree) {
                case 0: eval_sh<0, true>(row0, ux, uy, uz, r, g, b, shm); break;
                case 1: eval_sh<1, true>(row0, ux, uy, uz, r, g, b, shm); break;
                case 2: eval_sh<2, true>(row0, ux, uy, uz, r, g, b, shm); break;
                default: eval_sh<3, true>(row0, ux, uy, uz, r, g, b, shm); break;
            }
        } else switch (P.sh_degree) {
            case 0: eval_sh<0, false>(row0, ux, uy, uz, r, g, b); break;
            case 1: eval_sh<1, false>(row0, ux, uy, uz, r, g, b); break;
            case 2: eval_sh<2, false>(row0, ux, uy, uz, r, g, b); break;
            default: eval_sh<3, false>(row0, ux, uy, uz, r, g, b); break;
        }
        const float depth = (float)tz;
        // the record of the composite (sgs_common.h): constants folded per splat, not per (splat, tile) or per pixel
        float4* sp = reinterpret_cast<float4*>(splats + slot);
        // q2 = -power * log2(e) = A dx^2 + B dx dy + C dy^2 with the fp32 conic (ca, cb, cc) — the form S6 defines — is stored as
        // the completed square  A (dx + k dy)^2 + C' dy^2,  k = B / 2A,  C' = C - B^2 / 4A = det / A,  derived in fp64 FROM the
        // fp32 conic: two non-negative terms instead of three that cancel.  A needle (sigma_minor ~ 0.55 px from the dilation,
        // sigma_major hundreds of pixels) evaluated far along its axis has A dx^2 ~ 1e6 cancelling to q2 ~ 1; the three-term
        // form in fp32 then carries an error of ~0.1 in the exponent (found by the wild fuzz cases), this one ~1e-4.
        const double A64 = (0.5 * 1.4426950408889634) * (double)ca, B64 = 1.4426950408889634 * (double)cb,
                     C64 = (0.5 * 1.4426950408889634) * (double)cc;
        const double iA = A64 > 0.0 ? 1.0 / A64 : 0.0;
        const double k64 = 0.5 * B64 * iA, Cp64 = C64 - 0.25 * B64 * B64 * iA;
        // ... and the record holds the ROOTS a = sqrt(A), a k, c = sqrt(|C'|) (with the sign of C': negative = the fp32 conic
        // rounded to an indefinite form), so that the composite evaluates  q2 = U^2 + V^2,  U = a (dx + k dy),  V = c dy  with
        // two fmas for U, one for V and two for the sum (k_tile_render)
        const double a64 = sqrt(A64 > 0.0 ? A64 : 0.0), c64 = copysign(sqrt(fabs(Cp64)), Cp64);
        // the opacity enters the composite's exponent: alpha / alpha_max = min(1, 2^-(q2 + nlo)),  nlo = log2(alpha_max / o)
        const float nlo = __log2f(P.alpha_max) - __log2f(g0.w);
        // Fine tiles (sgs_common.h): the composite works in CELL pixels, 2^z per pixel — positions and extents times 2^z, the roots divided
        // by it (exact), so that U, V and q come out as they would on the frame's own pixel grid, bit for bit up to the cell's origin
        const float up = (float)(1 << zf), dn = 1.0f / up;
        sp[0] = make_float4(sx * up, sy * up, (float)a64 * dn, (float)(a64 * k64) * dn);
        sp[1] = make_float4((float)c64 * dn, nlo, r, g);
        sp[2] = make_float4(b, depth, fminf(ext_x * up, 3.0e38f), fminf(ext_y * up, 3.0e38f));
        sp[3] = make_float4(qmax, g0.w, __uint_as_float(rect01), __uint_as_float(rect23));
        // what the binning kernels need, densely: one coalesced 1-KiB row per chunk instead of a 48-B-stride gather
        binrec[pos] = uint4{__float_as_uint(depth), brect01, brect23, slot};
    }

}

// S1-S3.  One lane = one Gaussian, one wave = one 64-Gaussian chunk of the scene (all loads are full
// 1-KiB rows).  Geometry runs in fp64 (MI355X fp64 vector rate is 1/2 of fp32 and this kernel is
// HBM-bound), which makes every integer decision (cull, radius, rect, tile counts) agree with the
// fp64 oracle.  A pure map kernel: no LDS, no atomics; vismask[chunk] (the wave's ballot) tells the
// consumers which of the chunk's 64 splat slots are live this frame.
// LOOP = false: wave k of the launch takes entry k of the frame's live list (k_chunk_cull) — a wave per chunk of the SCENE is
//               launched, the chunks that cannot reach this frame are never touched (their waves end at once);
// LOOP = true : the waves of a smaller grid take the entries in turn.  For the frames of a group that are narrow bands of tile
//               rows (a few per cent of 47 k chunks live: the launch otherwise starts 45 k waves per frame only to end them).
//               Not for full frames: the loop makes the compiler keep the frame's constants (the view matrix as doubles, ...)
//               in 43 more VGPRs — 135 instead of 92, three waves per SIMD instead of five, and a wave that no longer fits
//               beside the composite's workgroups of the frames in flight (a sweep was 2 % slower, r03z).
template <bool LOOP, bool FINE>
__global__ __launch_bounds__(256) void k_preprocess(const FrameGroup G) {
    const FrameSlot& S = G.s[blockIdx.y];                  // this workgroup's frame of the group
    const FrameParams& P = S.P;
    const float4* __restrict__ geom = G.geom; const float4* __restrict__ shq = G.shq;
    Splat* __restrict__ splats = S.splats;
    unsigned long long* __restrict__ vismask = S.vismask; unsigned long long* __restrict__ bigmask = S.bigmask;
    unsigned* __restrict__ big_list = S.big_list; uint4* __restrict__ binrec = S.binrec;
    FrameStatus* __restrict__ st = S.st;
    const int lane = threadIdx.x & 63;
    const unsigned n_live = st->n_live, k0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (!LOOP) {
        if (k0 >= n_live) return;                          // wave-uniform
        preprocess_chunk<FINE>(P, geom, shq, splats, vismask, bigmask, big_list, binrec, st, (long long)S.live_list[k0], lane);
    } else {
        const unsigned nw = gridDim.x * (blockDim.x >> 6);
        for (unsigned k = k0; k < n_live; k += nw)         // wave-uniform
            preprocess_chunk<FINE>(P, geom, shq, splats, vismask, bigmask, big_list, binrec, st, (long long)S.live_list[k], lane);
    }
}

// The projection of a frame GROUP (full frames): the frames of a group read the SAME scene — a chunk that is live in
// several of them (a trajectory's consecutive frames: almost all) is 15 KiB of geometry and SH rows per frame, from HBM every time when each
// frame walks its own live list in its own part of the grid.  Here the waves take the entries of the group's WORK LIST (k_chunk_cull_group:
// (chunk, frame) pairs, chunk-major), so that the frames that want a chunk run side by side — mostly in one workgroup, always on one XCD:
// workgroup b runs on XCD b mod 8 (round-robin dispatch) and takes position (b mod 8) ceil(n / 8) + b / 8 of the n live workgroups, i.e.
// every XCD walks ONE contiguous eighth of the list, in order — and the second to fourth find the rows in the CU's L1 / the XCD's L2 (or merge
// with the request in flight).  Waves beyond the list's end end at once, as in k_preprocess<false>.  Same per-chunk code, same outputs:
// frames bit-identical.
template <bool FINE>
__global__ __launch_bounds__(256) void k_preprocess_shared(const FrameGroup G) {
    const unsigned n_work = G.s[0].st->n_work;
    const unsigned wpb = blockDim.x >> 6, n_wg = (n_work + wpb - 1) / wpb, per_xcd = (n_wg + SGS_XCDS - 1) / SGS_XCDS;
    const unsigned b = blockIdx.x, r = b % SGS_XCDS, t = b / SGS_XCDS;
    if (t >= per_xcd) return;                                               // workgroup-uniform
    const unsigned k = (r * per_xcd + t) * wpb + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (k >= n_work) return;                                                // wave-uniform
    const unsigned e = sgs_work_list(G.s[0].live_list, G.s[0].P.n_chunks)[k];
    const FrameSlot& S = G.s[e & 7u];
    preprocess_chunk<FINE>(S.P, G.geom, G.shq, S.splats, S.vismask, S.bigmask, S.big_list, S.binrec, S.st, (long long)(e >> 3), threadIdx.x & 63);
}

// ------------------------------------------------------------------------------------------------
// S4: exclusive scan of the tile counts (ONE counter per tile: level 2 of the binning needs no per-XCD sub-queues, see
// k_expand), D, the longest queue and the render order; the queue of tile t is [offset[t], offset[t+1]).
//
// One workgroup per SGS_SCAN_THREADS (512) tiles of the band, and NO communication between them: every workgroup reads ALL the band's
// counters (32 KB at 1080p, L2-resident, every load coalesced and in flight together) and derives what it needs about the
// other workgroups' tiles itself — the records queued before its own tiles, and the tiles per length class before them
// and overall — then writes the offsets and render-order entries of its own tiles (one per thread).  A single
// workgroup doing all of it was bound by ONE CU's memory and LDS pipelines (24 us at 1080p, 90 us at 3840x2160 as a
// kernel of its own in r01; 30 us as the tail of k_expand<false> in r03w: profiles/r03w_fused_scans_experiment.txt).
// The counters are cleared by k_expand<true> (the next launch), not here: other workgroups may still be reading them.
// 512 threads per workgroup: 1024 (sixteen waves at 116 VGPRs: nearly a whole idle CU) waited ~95 us for a place beside the
// composite's workgroups of the frames in flight; 512 costs 2.4 us more alone (16 workgroups each read every counter) and a sweep
// is 2 % faster; 256: +8 us alone, +1.3 % (r03z)
#ifndef SGS_SCAN_THREADS
#define SGS_SCAN_THREADS 512
#endif
#define SGS_SCAN_SLABS 8
// Per (wave, class) ONE LDS atomic instead of 64 serialised ones on the same address: neighbouring tiles have queues
// of similar length, so a wave holds two or three classes.  Returns the lane's position (old cursor + rank in its class).
__device__ __forceinline__ unsigned class_take(unsigned* s_cls, unsigned cls, bool valid, int lane) {
    unsigned pos = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned lc = (unsigned)__shfl((int)cls, leader);
        const bool mine = valid && cls == lc;
        const unsigned long long m = __ballot(mine);
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&s_cls[lc], (unsigned)__popcll(m));
        base = (unsigned)__shfl((int)base, leader);
        if (mine) pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    return pos;
}
__device__ __forceinline__ unsigned queue_class(unsigned c) { return c ? 32u - (unsigned)__clz((int)c) : 0u; }

__global__ __launch_bounds__(SGS_SCAN_THREADS) void k_tile_scan(const FrameGroup G) {
    const FrameSlot& S = G.s[blockIdx.y];
    const FrameParams& P = S.P;
    const unsigned* __restrict__ tile_count = S.tile_count; unsigned* __restrict__ tile_offset = S.tile_offset;
    uint4* __restrict__ tile_order = S.tile_order; unsigned long long* __restrict__ row_acc = G.row_acc;
    FrameStatus* __restrict__ st = S.st;
    constexpr int NW = SGS_SCAN_THREADS / SGS_WAVE;
    __shared__ unsigned s_row[SGS_SCAN_THREADS + 2];  // records per tile row among my tiles (-> row_acc, see the end)
    __shared__ unsigned s_all[33], s_before[33];      // tiles per log2(queue length) class: whole band / before my tiles
    __shared__ unsigned s_cur[33];                    // my tiles' cursors: first render position of each class for them
    __shared__ unsigned s_wa[NW], s_wb[NW], s_wm[NW], s_wi[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t_lo = P.row_begin * P.gx, t_hi = P.row_end * P.gx;      // the tiles this call renders: rects are clipped
    const int g = (int)blockIdx.x;                                     // to them in k_preprocess (a rank scans 1/N)
    if (tid < 33) { s_all[tid] = 0; s_before[tid] = 0; }
    for (int i = tid; i < SGS_SCAN_THREADS + 2; i += SGS_SCAN_THREADS) s_row[i] = 0;
    __syncthreads();
    unsigned sum_all = 0, sum_before = 0, mx = 0;
    unsigned mine = 0;                                // my tile's count
    for (int t0 = t_lo, slab0 = 0; t0 < t_hi; t0 += SGS_SCAN_THREADS * SGS_SCAN_SLABS, slab0 += SGS_SCAN_SLABS) {
        unsigned cc[SGS_SCAN_SLABS];
#pragma unroll
        for (int j = 0; j < SGS_SCAN_SLABS; ++j) {
            const int t = t0 + j * SGS_SCAN_THREADS + tid;
            cc[j] = t < t_hi ? tile_count[t] : 0u;
        }
#pragma unroll
        for (int j = 0; j < SGS_SCAN_SLABS; ++j) {
            const int t = t0 + j * SGS_SCAN_THREADS + tid;
            const unsigned c = cc[j];
            const bool before = slab0 + j < g;                         // slab k = the 1024 tiles of workgroup k (uniform)
            sum_all += c; sum_before += before ? c : 0u;
            mx = c > mx ? c : mx;
            // tiles per class, counted per (wave, class)
            unsigned long long todo = __ballot(t < t_hi);
            const unsigned cls = queue_class(c);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const unsigned lc = (unsigned)__shfl((int)cls, leader);
                const unsigned long long m = __ballot(t < t_hi && cls == lc);
                if (lane == leader) { atomicAdd(&s_all[lc], (unsigned)__popcll(m)); if (before) atomicAdd(&s_before[lc], (unsigned)__popcll(m)); }
                todo &= ~m;
            }
            if (slab0 + j == g) mine = c;
        }
    }
    // records before my tiles, in the band; longest queue
    const unsigned wa = wave_sum(sum_all), wb = wave_sum(sum_before), wm = wave_max(mx);
    const int t = t_lo + g * SGS_SCAN_THREADS + tid;                   // my tile
    const unsigned c = mine;
    const unsigned incl = wave_incl_scan(c, lane);
    if (lane == 63) s_wi[wave] = incl;
    if (lane == 0) { s_wa[wave] = wa; s_wb[wave] = wb; s_wm[wave] = wm; }
    __syncthreads();
    unsigned total = 0, base = 0, wbase = 0, tmax = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        total += s_wa[w]; base += s_wb[w]; tmax = s_wm[w] > tmax ? s_wm[w] : tmax;
        wbase += w < wave ? s_wi[w] : 0u;
    }
    // Render order: longest queues first (log2 classes).  k_tile_render's blocks are dispatched in index order and a
    // long tile costs as much as the whole kernel's average share, so it must not start last.  Inside a class: by tile.
    if (tid < 33) {
        unsigned first = 0;
        for (int k = 32; k > tid; --k) first += s_all[k];
        s_cur[tid] = first + s_before[tid];
    }
    __syncthreads();
    if (t < t_hi) tile_offset[t] = base + wbase + incl - c;
    {
        const unsigned pos = class_take(s_cur, queue_class(c), t < t_hi, lane);
        // (tile, first record, queue length): all the composite's workgroup needs before it can fetch its queue
        if (t < t_hi) tile_order[pos] = uint4{(unsigned)t, base + wbase + incl - c, c, 0u};
    }
    if ((unsigned long long)total <= (unsigned long long)P.rec_capacity) {   // (uniform; an overflowed frame is rendered again: counted then)
        // records per FRAME tile row, accumulated over frames (sgs_row_records: what cost-balanced bands are cut from)
        const int first_y = (t_lo + g * SGS_SCAN_THREADS) / P.gx;
        const int my_y = t < t_hi ? t / P.gx - first_y : -1;
        if (P.gx >= SGS_WAVE) {                      // a wave's 64 consecutive tiles lie in two rows at most
            const int ya = (t_lo + g * SGS_SCAN_THREADS + wave * SGS_WAVE) / P.gx - first_y;
            const unsigned sa = wave_sum(my_y == ya ? c : 0u), sb = wave_sum(my_y == ya + 1 ? c : 0u);
            if (lane == 0) { if (sa) atomicAdd(&s_row[ya], sa); if (sb) atomicAdd(&s_row[ya + 1], sb); }
        } else if (my_y >= 0 && c) atomicAdd(&s_row[my_y], c);
        __syncthreads();
        for (int i = tid; i < SGS_SCAN_THREADS + 2; i += SGS_SCAN_THREADS) {
            const unsigned v = s_row[i];
            const int fr = (int)(frame_row_of(P, (unsigned)(first_y + i)) >> fine_shift(P));     // (the frame's 16-pixel row)
            if (v && fr < SGS_MAX_ROWS) atomicAdd(&row_acc[fr], (unsigned long long)v);
        }
    }
    if (g == 0 && tid == 0) {
        tile_offset[t_hi] = total;                         // end of the band's last queue
        st->d_total = total;
        st->max_tile_len = tmax;
        if ((unsigned long long)total > (unsigned long long)P.rec_capacity) st->overflow = 1u;      // (k_stile_scan may have set it already)
    }
}

// ------------------------------------------------------------------------------------------------
// S4 binning, part 1 (count) and part 2 (emit).  Both walk the live splats of the workgroup's ranges
// (b, b+B, b+2B, ...: a uniform sample of the scene, so the static split is balanced) and expand every
// rect into its tiles: one lane per splat for rects of <= SGS_BIG_RECT tiles, the whole workgroup per
// splat for the few larger ones (k_preprocess's big list).
//
// Device-scope atomics on MI355X execute in the fabric and serialise per address (~12 ns each), so
// one atomic per (splat, tile) record is hopeless for a tile that receives 20 k records.  Instead
// each workgroup keeps the counters of a WINDOW of SGS_WT tiles in LDS (32 KB; a 1080p frame is
// one window, 4K is four), counts its records there with LDS atomics, and then issues ONE global
// atomicAdd per tile it touched: the returned value is the workgroup's base inside that tile's queue.
// The (tile, base) pairs go to a per-workgroup list; after the scan, k_bin_emit reloads them as
// absolute cursors into LDS and writes every record with an LDS atomic only.
// The live chunks of this workgroup's ranges (b, b+B, b+2B, ...), found in ONE coalesced sweep over the
// visibility masks and appended to an LDS list — so that no wave ever waits on a mask load just to learn
// that a chunk is culled.  Must be called by all threads; ends with a barrier.
struct LiveChunks {
    unsigned long long mask[SGS_MAX_LIVE];  // live, non-big lanes of the chunk
    unsigned chunk[SGS_MAX_LIVE];           // chunk index
    unsigned n;
    unsigned n_vis;                         // live splats (popcount of the visibility masks)
};
// One sweep covers SGS_BIN_THREADS / 16 of the workgroup's ranges; a pass does at most
// SGS_MAX_LIVE / SGS_BIN_THREADS sweeps, so the list cannot overflow however large the scene is.
// The binning grid is (workgroups per window) x (windows): workgroup (b, w) bins the ranges b, b+B, b+2B, ... into
// window w.  A frame of more than SGS_WT tiles (4K) thus runs its windows side by side instead of as sequential
// passes of every workgroup — these kernels leave the chip mostly idle (~1 wave per SIMD), so the passes overlap.
__device__ __forceinline__ unsigned bin_B(const FrameParams& P) { return gridDim.x / (unsigned)max(1, P.n_windows); }
__device__ __forceinline__ unsigned bin_b(const FrameParams& P) { return blockIdx.x % bin_B(P); }
__device__ __forceinline__ int bin_w(const FrameParams& P) { return (int)(blockIdx.x / bin_B(P)); }
#define SGS_RANGES_PER_SWEEP (SGS_BIN_THREADS / SGS_RANGE_CHUNKS)
#define SGS_SWEEPS_PER_PASS (SGS_MAX_LIVE / SGS_BIN_THREADS)
// The binning kernels deal RANGES of SGS_RANGE_CHUNKS consecutive entries of the frame's live list round-robin to the
// workgroups: range j of workgroup b is range j * B + b of the list.
__device__ __forceinline__ int bin_sweeps(const FrameParams& P, unsigned n_live) {
    const int n_ranges = (int)((n_live + SGS_RANGE_CHUNKS - 1) / SGS_RANGE_CHUNKS);
    const int mine = (n_ranges - (int)bin_b(P) + (int)bin_B(P) - 1) / (int)bin_B(P);
    return mine > 0 ? (mine + SGS_RANGES_PER_SWEEP - 1) / SGS_RANGES_PER_SWEEP : 0;
}
__device__ __forceinline__ void find_live_chunks(const FrameParams& P, const unsigned long long* __restrict__ vismask,
                                                 const unsigned long long* __restrict__ bigmask,
                                                 const unsigned* __restrict__ live_list, unsigned n_live,
                                                 LiveChunks& lc, int sweep0, int sweep1) {
    if (threadIdx.x == 0) { lc.n = 0; lc.n_vis = 0; }
    __syncthreads();
    unsigned vis = 0;
    for (int sw = sweep0; sw < sweep1; ++sw) {
        const int j = sw * SGS_RANGES_PER_SWEEP + (int)(threadIdx.x / SGS_RANGE_CHUNKS);     // j-th range of this workgroup
        const long long r = (long long)j * bin_B(P) + bin_b(P);
        const long long e = r * SGS_RANGE_CHUNKS + (threadIdx.x % SGS_RANGE_CHUNKS);         // entry of the live list
        if (e < (long long)n_live) {
            const long long chunk = (long long)live_list[e];
            const unsigned long long vm = vismask[chunk];
            if (vm != 0ull) {
                vis += (unsigned)__popcll(vm);
                const unsigned long long m = vm & ~bigmask[chunk];
                if (m != 0ull) {
                    const unsigned k = atomicAdd(&lc.n, 1u);
                    lc.chunk[k] = (unsigned)chunk; lc.mask[k] = m;
                }
            }
        }
    }
    if (vis) atomicAdd(&lc.n_vis, vis);
    __syncthreads();
}

// ---- level 1: splats -> SUPER-TILES (SGS_ST x SGS_ST tiles, 64 x 64 px) --------------------------------------------------
// Binning is two-level.  A splat's tile rect covers a handful of tiles but one or two super-tiles, so the scattered,
// LDS-atomic-per-record duplication runs on ~1/3 (small splats) to ~1/12 (large ones) of the records, against a window of
// at most 8192 super-tile counters that is zeroed, swept and flushed in a fraction of the time the per-tile window took
// (a 1080p frame has 510 super-tiles); what it queues per super-tile is the splat's 16-byte binning record itself.
// Level 2 (k_expand) then hands every super-tile queue, in segments of SGS_SEG records, to workgroups that expand the
// records to the 16 tiles of THEIR super-tile with wave ballots — no atomics per record, one global atomic per (segment,
// tile) — and write each tile's records as contiguous runs.  (Per-tile binning in one level wrote 8-byte records one
// by one: on a scene with trained-3DGS statistics, D = 28 M, count + emit took 480 us of a 790-us frame, r03j.)
struct SuperGrid { int gxs, sr0, sr1, ns; };
__device__ __forceinline__ SuperGrid super_grid(const FrameParams& P) {
    SuperGrid g;
    g.gxs = (P.gx + SGS_ST - 1) >> SGS_ST_SHIFT;
    g.sr0 = P.row_begin >> SGS_ST_SHIFT; g.sr1 = (P.row_end + SGS_ST - 1) >> SGS_ST_SHIFT;     // rows of the band, in super-tiles
    g.ns = max(0, g.sr1 - g.sr0) * g.gxs;
    return g;
}
// a binning record's tile rect -> its super-tile rect, rows relative to the band's first super-row (empty rect: on = false)
__device__ __forceinline__ bool super_rect(const SuperGrid& SG, const uint4& br, unsigned& x0, unsigned& x1, unsigned& y0, unsigned& y1) {
    const unsigned tx0 = br.y & 0xffffu, tx1 = br.z & 0xffffu, ty0 = br.y >> 16, ty1 = br.z >> 16;
    if (tx1 <= tx0 || ty1 <= ty0) return false;
    x0 = tx0 >> SGS_ST_SHIFT; x1 = (tx1 + SGS_ST - 1) >> SGS_ST_SHIFT;
    y0 = (ty0 >> SGS_ST_SHIFT) - (unsigned)SG.sr0; y1 = ((ty1 + SGS_ST - 1) >> SGS_ST_SHIFT) - (unsigned)SG.sr0;
    return true;
}

// Bins the records of the listed chunks into the super-tiles of the band.
//   EMIT == false: s_arr = per-super-tile counters (LDS); counts every record.
//   EMIT == true : s_arr = per-super-tile write cursors (LDS); writes the 16-byte binning records to `srec`.
// One wave per chunk.  Neighbouring Gaussians cover the same super-tiles, so when the lanes overlap heavily the wave
// walks the UNION of its lanes' rects cell by cell and ballots "who covers this cell": one LDS atomic per (wave, cell)
// adds popc(ballot) and, in the emit, the covering lanes store one contiguous run.  Otherwise every lane walks its own rect.
#define SGS_WALK_TILES 8
#define SGS_FLUSH_ROUNDS 8
template <bool EMIT>
__device__ __forceinline__ void bin_walk(const SuperGrid& SG, const uint4* __restrict__ binrec, const LiveChunks& lc,
                                         unsigned* s_arr, uint4* __restrict__ srec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const unsigned nlive = lc.n;
    // a wave's chunks are a chain of (list entry -> 16-B record load -> walk): the next chunk's records are requested
    // before the current chunk is walked, so that their latency hides behind the walk
    uint4 br_next = {0u, 0u, 0u, 0u};
    bool live_next = false;
    if ((unsigned)wave < nlive) {
        live_next = (lc.mask[wave] >> lane) & 1ull;
        if (live_next) br_next = binrec[lc.chunk[wave] * SGS_WAVE + (unsigned)lane];
    }
    for (unsigned k = (unsigned)wave; k < nlive; k += (unsigned)nwaves) {
        unsigned x0 = 0xffffu, x1 = 0, y0 = 0xffffu, y1 = 0;         // empty rect
        const uint4 br = br_next;
        const bool live = live_next;
        {
            const unsigned kn = k + (unsigned)nwaves;
            live_next = kn < nlive && ((lc.mask[min(kn, (unsigned)SGS_MAX_LIVE - 1u)] >> lane) & 1ull);
            if (live_next) br_next = binrec[lc.chunk[kn] * SGS_WAVE + (unsigned)lane];
        }
        bool on = false;
        if (live) {
            unsigned a0, a1, b0, b1;
            if (super_rect(SG, br, a0, a1, b0, b1)) { x0 = a0; x1 = a1; y0 = b0; y1 = b1; on = true; }
        }
        const unsigned cnt = on ? (x1 - x0) * (y1 - y0) : 0u;
        const unsigned ux0 = wave_min(x0), uy0 = wave_min(y0), ux1 = wave_max(x1), uy1 = wave_max(y1);
        if (ux1 <= ux0 || uy1 <= uy0) continue;                              // nothing to bin
        const unsigned area = (ux1 - ux0) * (uy1 - uy0), total = wave_sum(cnt);
        if (area * 16u <= total) {
            // cell-major pays ~300 cycles per union cell (a ballot -> scalar -> one-lane-atomic chain): worth it only when the
            // lanes overlap heavily (>= 16 records per cell of the union rect)
            for (unsigned ty = uy0; ty < uy1; ++ty) {
                const bool row_on = on && ty >= y0 && ty < y1;
                const unsigned row = ty * (unsigned)SG.gxs;
                for (unsigned tx = ux0; tx < ux1; ++tx) {
                    const unsigned long long m = __ballot(row_on && tx >= x0 && tx < x1);
                    if (m == 0ull) continue;
                    const int leader = __ffsll((long long)m) - 1;
                    if (!EMIT) {
                        if (lane == leader) atomicAdd(&s_arr[row + tx], (unsigned)__popcll(m));
                    } else {
                        unsigned base = 0;
                        if (lane == leader) base = atomicAdd(&s_arr[row + tx], (unsigned)__popcll(m));
                        base = __shfl(base, leader);
                        if ((m >> lane) & 1ull) srec[base + lanes_below(m)] = br;
                    }
                }
            }
        } else {
            // lane-major for the SMALL rects: every lane walks its own, SGS_WALK_TILES cells per trip; every cell's (row, column) comes
            // from its own index (i + 0.5) / w — exact in fp32 for rects of up to 2^21 cells — and the LDS atomics of a trip are all in
            // flight together.
            // A LARGER rect is walked by the WHOLE WAVE instead (round 6): its owner's record is read into scalar
            // registers (v_readlane) and the 64 lanes take the cells 64 at a time — distinct cells, so the LDS atomics of a trip do not
            // collide, and in the emit the record goes out as one broadcast store per cell.  Left to its owner, a lane with a 100-cell
            // rect kept its wave for 13 trips while 63 lanes waited: on a scene with trained-3DGS statistics (heavy-tailed scales: 3.9
            // super-tiles per splat on average, a long tail up to SGS_BIG_RECT) the walk was 36-51 k of k_bin_count's 67-88 k cycles per
            // workgroup (profiles/r06e_bin_prof.txt).
            // Which rects count as "more than a lane should walk" is decided per chunk (wave-uniform): with the threshold at T cells the
            // lane-major part takes T / 8 trips (all lanes side by side) and every rect above it one or two passes of the whole wave, which
            // cost about two trips each — twenty lanes with 12-cell rects are two lane-major trips, not twenty passes (T = 8 for every
            // chunk made the room scene's binning 4 % slower, profiles/r06g).
            unsigned T = SGS_WALK_TILES, best = ~0u;
#pragma unroll
            for (unsigned cand = SGS_WALK_TILES; cand <= 16u * SGS_WALK_TILES; cand <<= 1) {
                const unsigned cost = cand / SGS_WALK_TILES + 2u * (unsigned)__popcll(__ballot(on && cnt > cand));
                if (cost < best) { best = cost; T = cand; }
            }
            const bool coop = on && cnt > T;
            if (on && !coop) {
                const unsigned w = x1 - x0;
                const float rw = 1.0f / (float)w;
                const unsigned origin = y0 * (unsigned)SG.gxs + x0;
                for (unsigned i = 0; i < cnt; i += SGS_WALK_TILES) {
                    unsigned tl[SGS_WALK_TILES], dst[SGS_WALK_TILES];
#pragma unroll
                    for (int u = 0; u < SGS_WALK_TILES; ++u) {
                        const unsigned iu = i + (unsigned)u;
                        const unsigned ty = (unsigned)(((float)iu + 0.5f) * rw);
                        tl[u] = origin + ty * (unsigned)SG.gxs + (iu - ty * w);
                    }
#pragma unroll
                    for (int u = 0; u < SGS_WALK_TILES; ++u) if (i + (unsigned)u < cnt) dst[u] = atomicAdd(&s_arr[tl[u]], 1u);
                    if (EMIT) {
#pragma unroll
                        for (int u = 0; u < SGS_WALK_TILES; ++u) if (i + (unsigned)u < cnt) srec[dst[u]] = br;
                    }
                }
            }
            unsigned long long cm = __ballot(coop);
            while (cm != 0ull) {                                      // (wave-uniform)
                const int src = __ffsll((long long)cm) - 1;
                cm &= cm - 1ull;
                const unsigned rx0 = (unsigned)__builtin_amdgcn_readlane((int)x0, src), rx1 = (unsigned)__builtin_amdgcn_readlane((int)x1, src);
                const unsigned ry0 = (unsigned)__builtin_amdgcn_readlane((int)y0, src), ry1 = (unsigned)__builtin_amdgcn_readlane((int)y1, src);
                uint4 rbr = {0u, 0u, 0u, 0u};
                if (EMIT) {
                    rbr.x = (unsigned)__builtin_amdgcn_readlane((int)br.x, src); rbr.y = (unsigned)__builtin_amdgcn_readlane((int)br.y, src);
                    rbr.z = (unsigned)__builtin_amdgcn_readlane((int)br.z, src); rbr.w = (unsigned)__builtin_amdgcn_readlane((int)br.w, src);
                }
                const unsigned w = rx1 - rx0, total = w * (ry1 - ry0);
                const float rw = 1.0f / (float)w;
                const unsigned origin = ry0 * (unsigned)SG.gxs + rx0;
                for (unsigned k = (unsigned)lane; k < total; k += 64u) {
                    const unsigned ty = (unsigned)(((float)k + 0.5f) * rw);
                    const unsigned d = atomicAdd(&s_arr[origin + ty * (unsigned)SG.gxs + (k - ty * w)], 1u);
                    if (EMIT) srec[d] = rbr;
                }
            }
        }
    }
}

// The big-rect splats (k_preprocess's list: more than SGS_BIG_RECT super-tiles), dealt round-robin to the workgroups;
// every thread of the workgroup takes a share of each rect.  The workgroup first fetches the records of up to
// SGS_BIN_THREADS of its rects in ONE round of loads (list entry -> record: two dependent global loads, ~3 us) and then
// walks them out of LDS.  Must be called by all threads.  f(cell, record).
template <class F>
__device__ __forceinline__ void bin_walk_big(const FrameParams& P, const SuperGrid& SG, const uint4* __restrict__ binrec,
                                             const unsigned* __restrict__ big_list, unsigned n_big, uint4* s_big, F&& f) {
    n_big = min(n_big, (unsigned)SGS_BIG_CAP);
    const unsigned B = bin_B(P), b = bin_b(P);
    const unsigned mine = n_big > b ? (n_big - b + B - 1u) / B : 0u;      // rects b, b + B, b + 2B, ... (uniform)
    for (unsigned r0 = 0; r0 < mine; r0 += SGS_BIN_THREADS) {
        const unsigned nr = min((unsigned)SGS_BIN_THREADS, mine - r0);
        if (threadIdx.x < nr) s_big[threadIdx.x] = binrec[big_list[b + (r0 + threadIdx.x) * B]];
        __syncthreads();
        for (unsigned j = 0; j < nr; ++j) {
            const uint4 br = s_big[j];
            unsigned x0, x1, y0, y1;
            if (!super_rect(SG, br, x0, x1, y0, y1)) continue;   // workgroup-uniform
            const unsigned w = x1 - x0, total = w * (y1 - y0);
            const float rw = 1.0f / (float)w;
            for (unsigned k = threadIdx.x; k < total; k += blockDim.x) {
                // k / w without an integer divide: exact for k < 2^21 (tests/test_emu_parity.py)
                const unsigned ty = (unsigned)(((float)k + 0.5f) * rw);
                f((y0 + ty) * (unsigned)SG.gxs + x0 + (k - ty * w), br);
            }
        }
        __syncthreads();
    }
}

// (~15 KB of LDS per workgroup at 1080p: what a binning workgroup takes from a CU is what the composite workgroups of
// the frames in flight cannot use)
__global__ __launch_bounds__(SGS_BIN_THREADS) void k_bin_count(const FrameGroup G) {
    const FrameSlot& S = G.s[blockIdx.y];
    const FrameParams& P = S.P;
    const uint4* __restrict__ binrec = S.binrec;
    const unsigned long long* __restrict__ vismask = S.vismask; const unsigned long long* __restrict__ bigmask = S.bigmask;
    const unsigned* __restrict__ big_list = S.big_list; unsigned* __restrict__ stile_count = S.stile_count;
    uint2* __restrict__ blk_list = S.blk_list; unsigned* __restrict__ blk_len = S.blk_len;
    FrameStatus* __restrict__ st = S.st; unsigned long long* prof = S.bin_prof;
    (void)prof;
#ifdef SGS_TILE_PROF
    unsigned long long bt0 = clock64(), bt_find = 0, bt_walk = 0, bt_flush = 0, bt_big = 0, btm = bt0;
#define SGS_BPROF(acc) do { unsigned long long now_ = clock64(); acc += now_ - btm; btm = now_; } while (0)
#else
#define SGS_BPROF(acc) do { } while (0)
#endif
    SGS_DYNAMIC_LDS(unsigned, s_cnt);                // one counter per super-tile of the band (rounded up to whole flush rounds)
    __shared__ unsigned s_nlist;
    __shared__ LiveChunks lc;
    __shared__ uint4 s_big[SGS_BIN_THREADS];          // the records of this workgroup's big rects, a round at a time
    const int tid = threadIdx.x;
    const unsigned xcd = xcc_id();
    const unsigned n_live = st->n_live;
    const unsigned* __restrict__ live_list = S.live_list;
    const int n_sweeps = bin_sweeps(P, n_live);
    const SuperGrid SG = super_grid(P);
    unsigned n_vis = 0;
    const unsigned b = bin_b(P);
    {
        const int used = ((SG.ns + 127) / 128) * 128;
        for (int i = tid; i < used; i += SGS_BIN_THREADS) s_cnt[i] = 0;
        if (tid == 0) s_nlist = 0;
        __syncthreads();
        for (int sw = 0; sw < n_sweeps; sw += SGS_SWEEPS_PER_PASS) {
            find_live_chunks(P, vismask, bigmask, live_list, n_live, lc, sw, min(n_sweeps, sw + SGS_SWEEPS_PER_PASS));
            SGS_BPROF(bt_find);
            n_vis += lc.n_vis;
            bin_walk<false>(SG, binrec, lc, s_cnt, nullptr);
            __syncthreads();
            SGS_BPROF(bt_walk);
        }
        bin_walk_big(P, SG, binrec, big_list, st->n_big, s_big, [&](unsigned cell, const uint4&) { atomicAdd(&s_cnt[cell], 1u); });
        __syncthreads();
        SGS_BPROF(bt_big);
        // flush: one device-scope atomic per touched super-tile — its return value is our base in that super-tile's
        // sub-queue.  The touched counters are compacted with a ballot straight into the workgroup's (cell, base) list.
        uint2* out = blk_list + (size_t)b * SGS_WT;
        {
            const int lane = tid & 63, wave = tid >> 6;
            for (int i0 = wave * (SGS_FLUSH_ROUNDS * 64); i0 < used; i0 += (SGS_BIN_THREADS / 64) * (SGS_FLUSH_ROUNDS * 64)) {
                unsigned tl[SGS_FLUSH_ROUNDS], c[SGS_FLUSH_ROUNDS], pre[SGS_FLUSH_ROUNDS], base[SGS_FLUSH_ROUNDS];
                unsigned tot = 0;
#pragma unroll
                for (int u = 0; u < SGS_FLUSH_ROUNDS; ++u) {
                    tl[u] = (unsigned)(i0 + u * 64 + lane);
                    c[u] = tl[u] < (unsigned)used ? s_cnt[tl[u]] : 0u;
                    const unsigned long long m = __ballot(c[u] != 0u);
                    pre[u] = tot + lanes_below(m);
                    tot += (unsigned)__popcll(m);
                }
                if (tot == 0u) continue;                                // (wave-uniform)
                unsigned k0 = 0;
                if (lane == 0) k0 = atomicAdd(&s_nlist, tot);
                k0 = __shfl(k0, 0);
#pragma unroll
                for (int u = 0; u < SGS_FLUSH_ROUNDS; ++u) {
                    base[u] = 0;
                    if (c[u] != 0u) base[u] = atomicAdd(&stile_count[(size_t)tl[u] * SGS_XCDS + xcd], c[u]);
                }
#pragma unroll
                for (int u = 0; u < SGS_FLUSH_ROUNDS; ++u)
                    if (c[u] != 0u) out[k0 + pre[u]] = make_uint2(tl[u], base[u]);
            }
        }
        __syncthreads();
        const unsigned nl = s_nlist;
        if (tid == 0) {
            blk_len[b] = nl;
            blk_len[SGS_BIN_BLOCKS + b] = xcd;                        // k_bin_emit must use the same sub-queue
            if (n_vis) atomicAdd(&st->n_visible, n_vis);              // one per workgroup
        }
        __syncthreads();
        SGS_BPROF(bt_flush);
    }
#ifdef SGS_TILE_PROF
    if (tid == 0 && prof) {
        unsigned long long* o = prof + (size_t)b * 8;
        o[0] = lc.n; o[1] = bt_find; o[2] = bt_walk; o[3] = bt_flush; o[4] = s_nlist; o[5] = n_vis; o[6] = clock64() - bt0; o[7] = bt_big;
    }
#endif
}

// Exclusive scan of the super-tile sub-counters (one workgroup: at most 8192 super-tiles), the total, and the level-2
// work list: super-tile s with c records becomes ceil(c / SGS_SEG) jobs (super-tile, first record, records).
#define SGS_SSCAN_THREADS 1024
__global__ __launch_bounds__(SGS_SSCAN_THREADS) void k_stile_scan(const FrameGroup G) {
    const FrameSlot& S = G.s[blockIdx.y];
    const FrameParams& P = S.P;
    const unsigned* __restrict__ stile_count = S.stile_count; unsigned* __restrict__ stile_offset = S.stile_offset;
    uint4* __restrict__ jobs = S.jobs; FrameStatus* __restrict__ st = S.st;
    constexpr int NW = SGS_SSCAN_THREADS / SGS_WAVE;
    __shared__ unsigned s_wr[NW], s_wj[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const SuperGrid SG = super_grid(P);
    const int per = (SG.ns + SGS_SSCAN_THREADS - 1) / SGS_SSCAN_THREADS;       // consecutive super-tiles per thread (<= 8)
    const int s0 = tid * per, s1 = min(SG.ns, s0 + per);
    unsigned rsum = 0, jsum = 0;
    for (int s = s0; s < s1; ++s) {
        const uint4* cp = reinterpret_cast<const uint4*>(stile_count + (size_t)s * SGS_XCDS);
        const uint4 a = cp[0], bq = cp[1];
        const unsigned c = a.x + a.y + a.z + a.w + bq.x + bq.y + bq.z + bq.w;
        rsum += c; jsum += (c + SGS_SEG - 1) / SGS_SEG;
    }
    const unsigned ri = wave_incl_scan(rsum, lane), ji = wave_incl_scan(jsum, lane);
    if (lane == 63) { s_wr[wave] = ri; s_wj[wave] = ji; }
    __syncthreads();
    unsigned rbase = ri - rsum, jbase = ji - jsum, rtot = 0, jtot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { rbase += w < wave ? s_wr[w] : 0u; jbase += w < wave ? s_wj[w] : 0u; rtot += s_wr[w]; jtot += s_wj[w]; }
    const bool fits = (unsigned long long)rtot <= (unsigned long long)(P.rec_capacity >> 1) && jtot <= (unsigned)P.job_capacity;
    for (int s = s0; s < s1; ++s) {
        const uint4* cp = reinterpret_cast<const uint4*>(stile_count + (size_t)s * SGS_XCDS);
        const uint4 a = cp[0], bq = cp[1];
        unsigned run = rbase;
        uint4 o0, o1;
        o0.x = run; run += a.x; o0.y = run; run += a.y; o0.z = run; run += a.z; o0.w = run; run += a.w;
        o1.x = run; run += bq.x; o1.y = run; run += bq.y; o1.z = run; run += bq.z; o1.w = run; run += bq.w;
        uint4* op = reinterpret_cast<uint4*>(stile_offset + (size_t)s * SGS_XCDS);
        op[0] = o0; op[1] = o1;
        const unsigned c = run - rbase;
        if (fits)
            for (unsigned q = 0; q < c; q += SGS_SEG) jobs[jbase++] = uint4{(unsigned)s, rbase + q, min((unsigned)SGS_SEG, c - q), 0u};
        rbase = run;
    }
    if (tid == 0) {
        st->ds_total = rtot;
        st->n_jobs = fits ? jtot : 0u;
        if (!fits) st->overflow = 1u;                 // the super-tile queues do not fit: nothing after this kernel runs (the caller grows them)
    }
}

__global__ __launch_bounds__(SGS_BIN_THREADS) void k_bin_emit(const FrameGroup G) {
    const FrameSlot& S = G.s[blockIdx.y];
    const FrameParams& P = S.P;
    const uint4* __restrict__ binrec = S.binrec;
    const unsigned long long* __restrict__ vismask = S.vismask; const unsigned long long* __restrict__ bigmask = S.bigmask;
    const unsigned* __restrict__ big_list = S.big_list; const unsigned* __restrict__ stile_offset = S.stile_offset;
    const uint2* __restrict__ blk_list = S.blk_list; const unsigned* __restrict__ blk_len = S.blk_len;
    uint4* __restrict__ srec = reinterpret_cast<uint4*>(S.alt); unsigned* __restrict__ stile_count = S.stile_count;
    const FrameStatus* __restrict__ st = S.st;
    SGS_DYNAMIC_LDS(unsigned, s_next);               // write cursors, one per super-tile
    __shared__ LiveChunks lc;
    __shared__ uint4 s_big[SGS_BIN_THREADS];
    const int tid = threadIdx.x;
    const SuperGrid SG = super_grid(P);
    {   // k_stile_scan has consumed the counters: zero again for the next frame (no per-frame memset), also when this
        // frame overflowed
        uint4* z = reinterpret_cast<uint4*>(stile_count);
        const unsigned nz = (unsigned)SG.ns * (SGS_XCDS / 4);
        for (unsigned i = blockIdx.x * SGS_BIN_THREADS + tid; i < nz; i += gridDim.x * SGS_BIN_THREADS) z[i] = uint4{0u, 0u, 0u, 0u};
    }
    if (st->overflow) return;
    const unsigned b = bin_b(P);
    const unsigned n_live = st->n_live;
    const unsigned* __restrict__ live_list = S.live_list;
    const int n_sweeps = bin_sweeps(P, n_live);
    {
        const unsigned xcd = blk_len[SGS_BIN_BLOCKS + b];              // the XCD k_bin_count ran workgroup b on
        const unsigned nl = blk_len[b];
        const uint2* in = blk_list + (size_t)b * SGS_WT;
        for (unsigned i = tid; i < nl; i += SGS_BIN_THREADS) {
            const uint2 e = in[i];
            s_next[e.x] = stile_offset[(size_t)e.x * SGS_XCDS + xcd] + e.y;
        }
        __syncthreads();
        for (int sw = 0; sw < n_sweeps; sw += SGS_SWEEPS_PER_PASS) {
            find_live_chunks(P, vismask, bigmask, live_list, n_live, lc, sw, min(n_sweeps, sw + SGS_SWEEPS_PER_PASS));
            bin_walk<true>(SG, binrec, lc, s_next, srec);
            __syncthreads();
        }
        bin_walk_big(P, SG, binrec, big_list, st->n_big, s_big,
                     [&](unsigned cell, const uint4& br) { srec[atomicAdd(&s_next[cell], 1u)] = br; });
        __syncthreads();
    }
}

// ---- level 2: super-tile queues -> tile queues -----------------------------------------------------------------------------
// Job j = SGS_SEG consecutive records of one super-tile's queue, four per thread.  Every record covers some of the
// super-tile's 16 tiles (its tile rect, clipped); for tile t the covering lanes of a wave find each other with ONE ballot:
//   EMIT == false: the ballots' popcounts are summed per tile over the workgroup and added to the tile's counter with one
//                  global atomic per tile — its return value is the job's base in that tile's queue (job_base).  (ONE counter
//                  per tile: unlike level 1, whose records arrive one by one from everywhere and are kept contiguous per XCD
//                  by sub-queues, a job writes each tile's records as one run whichever XCD it is on);
//   EMIT == true : the same ballots give every covering lane its rank: the records of (wave, slot, tile) are stored as one
//                  contiguous run at  tile offset + job base + (what the job's earlier waves and slots put into the tile).
// No per-record atomics, and the 8-byte records reach HBM in runs instead of one by one.
template <bool EMIT>
__global__ __launch_bounds__(SGS_EXP_THREADS) void k_expand(const FrameGroup G) {
    const FrameSlot& S = G.s[blockIdx.y];
    const FrameParams& P = S.P;
    const uint4* __restrict__ srec = reinterpret_cast<const uint4*>(S.alt); const uint4* __restrict__ jobs = S.jobs;
    unsigned* __restrict__ tile_count = S.tile_count; const unsigned* __restrict__ tile_offset = S.tile_offset;
    unsigned* __restrict__ job_base = S.job_base; unsigned long long* __restrict__ rec = S.rec;
    const FrameStatus* __restrict__ st = S.st;
    constexpr int NW = SGS_EXP_THREADS / SGS_WAVE, NT = SGS_ST * SGS_ST, R = SGS_SEG / SGS_EXP_THREADS;
    __shared__ unsigned s_wc[NW][NT];                // records per (wave, tile) of this job; EMIT: where the wave's run of the tile begins
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((unsigned)tid >> 6);          // (scalar: the empty slots of a short job are branches, not exec masks)
    if (EMIT) {   // k_tile_scan has consumed the band's tile counters: zero again for the next frame, also when this frame overflowed
        unsigned* z = tile_count + (size_t)P.row_begin * P.gx;
        const unsigned nz = (unsigned)((P.row_end - P.row_begin) * P.gx);
        for (unsigned i = blockIdx.x * SGS_EXP_THREADS + tid; i < nz; i += gridDim.x * SGS_EXP_THREADS) z[i] = 0u;
    }
    if (st->overflow) return;
    const SuperGrid SG = super_grid(P);
    const unsigned n_jobs = st->n_jobs;
    for (unsigned job = blockIdx.x; job < n_jobs; job += gridDim.x) {
        const uint4 J = jobs[job];
        const unsigned scell = __builtin_amdgcn_readfirstlane(J.x), qb = __builtin_amdgcn_readfirstlane(J.y), cnt = __builtin_amdgcn_readfirstlane(J.z);
        const unsigned tx0 = (scell % (unsigned)SG.gxs) << SGS_ST_SHIFT;
        const unsigned ty0 = (scell / (unsigned)SG.gxs + (unsigned)SG.sr0) << SGS_ST_SHIFT;     // the super-tile's first tile (owned rows)
        // slot r of this wave holds the records r * 256 + wave * 64 + (0..63) of the job: a short job (every super-tile's
        // last one) leaves whole slots empty, and their ballots are skipped (wave-uniform)
        bool slot_on[R];
#pragma unroll
        for (int r = 0; r < R; ++r) slot_on[r] = (unsigned)(r * SGS_EXP_THREADS + wave * SGS_WAVE) < cnt;
        // which of the 16 tiles each of my records covers (bit ly * 4 + lx)
        unsigned cover[R]; unsigned long long rv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned i = (unsigned)tid + (unsigned)(r * SGS_EXP_THREADS);
            cover[r] = 0u; rv[r] = 0ull;
            if (i < cnt) {
                const uint4 br = srec[qb + i];
                rv[r] = ((unsigned long long)br.x << 32) | br.w;
                // the rect [x0, x1) x [y0, y1) clipped to the super-tile's 4 x 4 tiles, as two bit ranges: the columns' mask replicated
                // into every row (x 0x1111) AND the rows' nibbles — a dozen integer operations per record instead of sixteen
                // compares and twelve selects
                const int x0 = (int)(br.y & 0xffffu) - (int)tx0, x1 = (int)(br.z & 0xffffu) - (int)tx0;
                const int y0 = (int)(br.y >> 16) - (int)ty0, y1 = (int)(br.z >> 16) - (int)ty0;
                const unsigned xl = (unsigned)min(max(x0, 0), SGS_ST), xh = (unsigned)min(max(x1, 0), SGS_ST);
                const unsigned yl = (unsigned)min(max(y0, 0), SGS_ST), yh = (unsigned)min(max(y1, 0), SGS_ST);
                static_assert(SGS_ST == 4, "cover masks are written for 4 x 4 tiles per super-tile");
                const unsigned xm = (1u << xh) - (1u << xl);                                   // xh >= xl: x1 > x0
                const unsigned rows = (1u << (yh << SGS_ST_SHIFT)) - (1u << (yl << SGS_ST_SHIFT));
                cover[r] = (xm * 0x1111u) & rows;
            }
        }
        // per (wave, tile) counts; lane t < 16 of every wave keeps tile t's
        unsigned mine = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            unsigned c = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) if (slot_on[r]) c += (unsigned)__popcll(__ballot((cover[r] >> t) & 1u));
            mine = lane == t ? c : mine;
        }
        if (lane < NT) s_wc[wave][lane] = mine;
        __syncthreads();
        if (tid < NT) {
            const unsigned tile = (ty0 + (unsigned)(tid >> SGS_ST_SHIFT)) * (unsigned)P.gx + tx0 + (unsigned)(tid & (SGS_ST - 1));
            unsigned tot = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += s_wc[w][tid];
            if (!EMIT) {
                // (a tile outside the grid / the band has no covering record: tot == 0 and nothing is touched)
                job_base[(size_t)job * NT + tid] = tot ? atomicAdd(&tile_count[tile], tot) : 0u;
            } else {
                unsigned pos = tot ? tile_offset[tile] + job_base[(size_t)job * NT + tid] : 0u;
#pragma unroll
                for (int w = 0; w < NW; ++w) { const unsigned c = s_wc[w][tid]; s_wc[w][tid] = pos; pos += c; }
            }
        }
        if (EMIT) {
            __syncthreads();
            // Lane t of every wave holds where the wave's records of tile t go; the position of a run is wave-uniform and stays
            // in scalar registers (v_readlane, s_bcnt1, s_add), a lane's place in the run is v_mbcnt of the ballot.
            const unsigned wpos = s_wc[wave][lane & (NT - 1)];
#pragma unroll
            for (int r = 0; r < R; ++r) SGS_PIN_VGPR(cover[r]);     // (the bits are tested again here, not kept in 64 registers since the count)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                unsigned long long* run = rec + (unsigned)__builtin_amdgcn_readlane((int)wpos, t);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (!slot_on[r]) continue;
                    const bool hit = (cover[r] & (1u << t)) != 0u;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);      // (the same i1 feeds the ballot and the branch: one compare)
                    if (hit) run[__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = rv[r];
                    run += (unsigned)__popcll(m);
                }
            }
        }
        __syncthreads();                              // s_wc / s_base are rewritten by the next job
    }
}

// ------------------------------------------------------------------------------------------------
// S5: per-tile LSD radix sort.  One workgroup (4 waves) per tile; the queue lives in LDS (classes
// S/M/L) or ping-pongs between two HBM buffers (class X, queues longer than SGS_CAP_L).
//
// One pass = per-wave digit histogram (LDS atomics) -> scan -> stable scatter.  Each wave owns a
// contiguous segment of the queue and walks it a row (64 keys) at a time; the lanes of a row that
// share a digit find each other with 8 wave ballots (one per digit bit), the lowest of them bumps
// the wave's running offset for that digit and broadcasts it.  No cross-wave traffic inside a pass.
struct SortShared {
    unsigned hist[4][SGS_RADIX];
    unsigned wsum[4];
    unsigned kmin, kmax, flag;
};

__device__ __forceinline__ void radix_pass(const unsigned* src_k, const unsigned* src_v,
                                           unsigned* dst_k, unsigned* dst_v, unsigned n,
                                           unsigned sub, unsigned shift, SortShared& sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned rows_total = (n + 63u) >> 6;
    const unsigned rows_per_wave = (rows_total + 3u) >> 2;
    const unsigned seg_beg = min(n, (unsigned)wave * rows_per_wave * 64u);
    const unsigned seg_end = min(n, seg_beg + rows_per_wave * 64u);
    // (1) zero
#pragma unroll
    for (int w = 0; w < 4; ++w) sh.hist[w][tid] = 0;
    __syncthreads();
    // (2) per-wave digit counts
    for (unsigned i = seg_beg + lane; i < seg_end; i += 64)
        atomicAdd(&sh.hist[wave][((src_k[i] - sub) >> shift) & (SGS_RADIX - 1)], 1u);
    __syncthreads();
    // (3) exclusive scan over (digit major, wave minor)
    {
        const unsigned c0 = sh.hist[0][tid], c1 = sh.hist[1][tid], c2 = sh.hist[2][tid], c3 = sh.hist[3][tid];
        const unsigned tot = c0 + c1 + c2 + c3;
        const unsigned incl = wave_incl_scan(tot, lane);
        if (lane == 63) sh.wsum[wave] = incl;
        __syncthreads();
        unsigned ex = incl - tot;
        for (int w = 0; w < wave; ++w) ex += sh.wsum[w];
        sh.hist[0][tid] = ex; sh.hist[1][tid] = ex + c0; sh.hist[2][tid] = ex + c0 + c1;
        sh.hist[3][tid] = ex + c0 + c1 + c2;
    }
    __syncthreads();
    // (4) stable scatter, row by row
    const unsigned rows_mine = (seg_end - seg_beg + 63u) >> 6;
    for (unsigned r = 0; r < rows_mine; ++r) {
        const unsigned i = seg_beg + r * 64u + lane;
        const bool valid = i < seg_end;
        const unsigned k = valid ? src_k[i] : 0u;
        const unsigned v = valid ? src_v[i] : 0u;
        const unsigned d = ((k - sub) >> shift) & (SGS_RADIX - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < SGS_RADIX_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(valid && bit);
            peers &= bit ? bal : ~bal;
        }
        const unsigned rank = lanes_below(peers);
        const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
        unsigned base = 0;
        if (valid && rank == 0) base = atomicAdd(&sh.hist[wave][d], (unsigned)__popcll(peers));
        base = __shfl(base, leader);
        if (valid) { dst_k[base + rank] = k; dst_v[base + rank] = v; }
    }
    __syncthreads();
}

// Sorts (k,v)[0..n) by ((k - sub) bits [0, nbits)), result back in (a_k, a_v).
__device__ __forceinline__ void radix_sort_bits(unsigned* a_k, unsigned* a_v, unsigned* b_k, unsigned* b_v,
                                                unsigned n, unsigned sub, unsigned nbits, SortShared& sh) {
    unsigned *sk = a_k, *sv = a_v, *dk = b_k, *dv = b_v;
    for (unsigned shift = 0; shift < nbits; shift += SGS_RADIX_BITS) {
        radix_pass(sk, sv, dk, dv, n, sub, shift, sh);
        unsigned* t = sk; sk = dk; dk = t;
        t = sv; sv = dv; dv = t;
    }
    if (sk != a_k) {
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x) { a_k[i] = sk[i]; a_v[i] = sv[i]; }
        __syncthreads();
    }
}

// Sorts (k,v)[0..n) by (depth bits, slot); b_k/b_v are ping-pong space of the same size.  A splat's
// slot IS its Gaussian index, so ties on depth break on v itself.  Keys lie in [sub, sub + 2^nbits).
// All threads of the workgroup must call it.  (Used for depth buckets too long for the LDS rank sort.)
__device__ __forceinline__ void sort_segment(unsigned* a_k, unsigned* a_v, unsigned* b_k, unsigned* b_v,
                                             unsigned n, unsigned sub, unsigned nbits, SortShared& sh,
                                             const Splat* __restrict__ splats, long long n_gauss,
                                             FrameStatus* st) {
    const unsigned tid = threadIdx.x;
    if (tid == 0) sh.flag = 0u;
    __syncthreads();
    radix_sort_bits(a_k, a_v, b_k, b_v, n, sub, nbits, sh);
    // Equal depth bits must order by Gaussian index.  Runs of equal keys are
    // contiguous now; short ones are fixed in place by their first lane, a long one flags the segment.
    for (unsigned i = tid; i + 1 < n; i += 256) {
        const unsigned k = a_k[i];
        if (a_k[i + 1] == k && (i == 0 || a_k[i - 1] != k)) {
            unsigned e = i + 2;
            while (e < n && a_k[e] == k) ++e;
            if (e - i > SGS_TIE_RUN_MAX) { atomicOr(&sh.flag, 1u); }
            else {
                for (unsigned p = i + 1; p < e; ++p) {
                    const unsigned v = a_v[p];
                    unsigned q = p;
                    while (q > i && a_v[q - 1] > v) { a_v[q] = a_v[q - 1]; --q; }
                    a_v[q] = v;
                }
            }
        }
    }
    __syncthreads();
    if (sh.flag) {
        // rare: stable two-key sort — first by Gaussian index, then by depth bits
        for (unsigned i = tid; i < n; i += 256) a_k[i] = a_v[i];
        __syncthreads();
        const unsigned idbits = n_gauss > 1 ? 64u - (unsigned)__clzll((long long)(n_gauss - 1)) : 1u;
        radix_sort_bits(a_k, a_v, b_k, b_v, n, 0u, idbits, sh);
        for (unsigned i = tid; i < n; i += 256) a_k[i] = splats[a_v[i]].key;
        __syncthreads();
        radix_sort_bits(a_k, a_v, b_k, b_v, n, sub, nbits, sh);
        if (tid == 0) atomicAdd(&st->n_resort_tiles, 1u);
    }
}

// Rank sort of cnt <= R*256 records held in LDS as kk[i] = depth bits << 32 | slot (all distinct);
// kk is padded with ~0 up to a multiple of 8.  Lane t owns records t, t+256, ...; it walks the whole
// list with broadcast LDS reads (8 per trip, so the read latency is paid once per 8 compares) and counts
// the records that precede each of its own.  Two barriers, no passes, ties resolved by the slot half.
template <int R>
__device__ __forceinline__ void rank_sort(const unsigned long long* kk, unsigned* out_v, unsigned cnt) {
    const unsigned tid = threadIdx.x;
    unsigned long long mine[R];
    unsigned rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned i = tid + 256u * (unsigned)r;
        mine[r] = i < cnt ? kk[i] : ~0ull;
        rank[r] = 0;
    }
    const unsigned cnt8 = (cnt + 7u) & ~7u;
    for (unsigned j = 0; j < cnt8; j += 8) {
        unsigned long long x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = kk[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) rank[r] += x[u] < mine[r] ? 1u : 0u;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (tid + 256u * (unsigned)r < cnt) out_v[rank[r]] = (unsigned)mine[r];
}

// ---- the blend loop of k_tile_render -----------------------------------------------------------------
// Issue costs measured on gfx950 (scripts/ubench2.hip, in-kernel clocks; shader cycles per wave64 instruction and SIMD at 8 waves per SIMD,
// profiles/r05b_ubench2_table.txt): v_fma_f32 2.40 (v_mul 2.29), packed fp32 (v_pk_*) 4.34 — barely ahead of two plain ops —, v_cmp / v_max /
// v_med3 and any SGPR operand 4.2-4.4, v_cndmask on vcc 19 (!), v_exp_f32 8.2, a broadcast ds_read_b32/b64 2.1 on the CU's shared LDS pipe
// (= 8.4 of each of the four SIMDs' time when all four stream reads), ds_read_b128 twice that; source modifiers (neg, abs) and the VOP3 `clamp`
// output modifier (result saturated to [0, 1], NaN -> 0) are free.  One wave ALONE on its SIMD issues a VALU instruction every 7.3 cycles.  The composite is issue-bound, so the per-pixel work is
// written for the smallest issue COST: no compare, select or min is left in the common trip — every predicate is a
// saturated fma, i.e. a factor 0 / 1, and every constant is folded into the splat once (k_preprocess), not per pixel:
//   * q2 = A dx^2 + B dx dy + C dy^2 = -power log2(e) is evaluated as the completed square A (dx + k dy)^2 + C' dy^2
//     (k = B / 2A, C' = C - B^2 / 4A: no cancellation between large terms for needle-shaped splats) from the ROOTS
//     a = sqrt(A), c = sqrt(C'):  q2 = U^2 + V^2,  U = a (dx + k dy),  V = c dy, and the opacity rides in the same fma
//     chain:   q = U^2 + (V^2 + nlo),  nlo = log2(alpha_max / o)   =>   2^-q = alpha / alpha_max;
//   * S6's clamp  alpha = min(alpha_max, o e^power)  is the clamp modifier of the v_exp:   E = sat(2^-q) = alpha / alpha_max;
//   * S6's cut-off  alpha >= alpha_min  <=>  q <= cq = log2(alpha_max / alpha_min)  is  vf = sat((cq - q) 2^100)  — exactly
//     0 or 1 for any fp32 q (the smallest non-zero |cq - q| times 2^100 is far above 1; q = cq counts as outside, a set of
//     measure zero inside the margin the tests check two-sidedly; NaN -> 0);
//     S6's other skip, power > 0, cannot fire while C' >= 0 (then q2 is a sum of squares).  A splat whose fp32 conic rounded
//     to an indefinite form (C' < 0, stored as c < 0: needles hundreds of pixels long) is flagged at staging and its whole
//     batch runs the EXACT variant of the trip (SGS_ALPHA_X: q2 = U^2 - V^2 for such a splat, compares and selects;
//     bit-identical results for every other splat of the batch);
//   * the pixel carries Tm = alpha_max T, so that  w = E vf Tm = alpha T  is the blend weight with no further factor and
//     tt = Tm - alpha_max w = alpha_max T (1 - alpha);  S6's stop rule  T (1 - alpha) < t_min  is  lv = sat((tt - alpha_max
//     t_min) 2^100), again exactly 0 or 1:  w *= lv (the splat that would end the pixel is not blended),  Tm = tt lv.
//     A finished pixel (and one outside the image) has Tm = 0 and stays there: w = 0, tt = 0, lv = 0.  No live mask;
//   * the transmittance a stopped pixel ended with (background term, coverage) is 1 - sum of its weights, accumulated
//     only in the instantiations that need it (TF: a non-black background, or the depth / coverage outputs);
//   * D_f (how far into the queue the tile's pixels read) is not tracked per splat: when a trip leaves a wave with
//     no live pixel — once per wave and tile — the trip is replayed from the saved Tm to find the splat that ended
//     the last pixel.
// Staged batch, five arrays of 8-byte pairs at ONE byte offset O = 8 j (so a trip computes no addresses); the tile-relative
// constants are computed at staging, once per splat and tile:
//   s_p0[j] = (m, a)   s_p1[j] = (a k, c ry)   s_p2[j] = (c, nlo)   s_p3[j] = (r, g)   s_p4[j] = (b, view depth)
//   m = a rx + a k ry,  (rx, ry) = centre - tile origin;  for the pixel (lx, ly) of the tile:
//   U = m - a lx - a k ly = a (dx + k dy),   V = c ry - c ly = c dy,   q = U^2 + (V^2 + nlo)      — five fmas.
// Tile-relative coordinates keep every term of U and V at the magnitude of sqrt(q) wherever a pixel can pass the cut-off
// (a far centre comes with a small a), so nothing cancels.
// A wave walks the splats of the batch that can touch ITS quadrant four per trip: the four alphas are
// independent (ILP hides the LDS and transcendental latency), then the short sequential part (T, colour, stop)
// is applied in depth order.  A short tail reads the inert dummy splat at index SGS_BATCH (nlo = 1e30: E = vf = 0).
#define SGS_EXP2(x) __builtin_amdgcn_exp2f(x)
#define SGS_RCP(x) __builtin_amdgcn_rcpf(x)         // 1 ulp; every use below is padded outward
#define SGS_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define SGS_SAT(x) __builtin_amdgcn_fmed3f((x), 0.0f, 1.0f)   // folds into the producing instruction's clamp modifier
#define SGS_BIG 1.2676506002282294e30f              // 2^100
#ifdef SGS_TILE_PROF   // profiling build: how many (wave, splat) evaluations had no pixel inside the alpha cut-off
#define SGS_PROF_EVAL(valid, J)                                                                        \
    if ((J) != (unsigned)SGS_BATCH) {                                                                  \
        const unsigned long long vb_ = __ballot(valid), lb_ = __ballot((valid) && Tm > 0.0f);          \
        ++pe_eval; pe_empty += vb_ == 0ull; pe_valid += (unsigned)__popcll(vb_); pe_useful += (unsigned)__popcll(lb_); \
        pe_dead += lb_ == 0ull; pe_few += lb_ != 0ull && __popcll(lb_) <= 2;                               \
    }
#define SGS_PROF_STAGED_ALL() atomicAdd(&s_pe[4], 1u);
#define SGS_PROF_STAGED(qb) if ((qb) != 0u) atomicAdd(&s_pe[5], 1u);
// a wave's list of a batch, by the number of its live pixels: splats listed, per class 1-4 / 5-8 / 9-16 / 17-32 / 33-64 (r06c: 89-97 % of them are listed
// by waves with more than 8 live pixels — why the tail blend of round 6 bought nothing); cycles of the trip loops
#define SGS_PROF_LIST(NL, CQ) { const unsigned nl_ = (NL); pw_hist[nl_ <= 4u ? 0 : nl_ <= 8u ? 1 : nl_ <= 16u ? 2 : nl_ <= 32u ? 3 : 4] += (CQ); pw_t0 = clock64(); }
#define SGS_PROF_LIST_END() { pw_list_cyc += clock64() - pw_t0; }
#else
#define SGS_PROF_EVAL(valid, J)
#define SGS_PROF_STAGED_ALL()
#define SGS_PROF_STAGED(qb)
#define SGS_PROF_LIST(NL, CQ)
#define SGS_PROF_LIST_END()
#endif
#define SGS_AT(arr, T_, off) (*reinterpret_cast<const T_*>(reinterpret_cast<const char*>(arr) + (off)))
#define SGS_NEXT(OV)                                                                                   \
    const unsigned OV = (mm != 0ull ? gwb + (unsigned)(__ffsll((long long)mm) - 1) : (unsigned)SGS_BATCH) << 3; \
    mm &= mm - 1ull;
// O = byte offset of the splat's slot in the five staging arrays (8 B per splat in each).  The order of a trip's reads and
// arithmetic is the compiler's: every order forced on it was measured slower (r03c/d/f/i) — all reads ahead of the
// arithmetic through volatile reads and a scheduling barrier +8 %, sched_group_barrier pipelines of "a splat's reads, then
// the previous splat's alpha" +7 ... +15 %, 16-byte reads instead of pairs +3 ... +7 %.
#define SGS_LOAD(O, N)                                                                                 \
    const float2 N##0 = SGS_AT(s_p0, float2, O), N##1 = SGS_AT(s_p1, float2, O), N##2 = SGS_AT(s_p2, float2, O), \
                 N##3 = SGS_AT(s_p3, float2, O), N##4 = SGS_AT(s_p4, float2, O);
// AL = alpha / alpha_max, or 0 outside the cut-off
#define SGS_ALPHA_F(O, N, AL)                                                                          \
    float AL;                                                                                          \
    {                                                                                                  \
        const float U = __builtin_fmaf(-(N##1).x, ly, __builtin_fmaf(-(N##0).y, lx, (N##0).x));   /* a (dx + k dy) */ \
        const float V = __builtin_fmaf(-(N##2).x, ly, (N##1).y);                                  /* c dy */ \
        const float q = __builtin_fmaf(U, U, __builtin_fmaf(V, V, (N##2).y));                     /* U^2 + (V^2 + nlo) */ \
        AL = SGS_SAT(SGS_EXP2(-q)) * SGS_SAT(__builtin_fmaf(-q, big, cq_big));                         \
        SGS_PROF_EVAL(q < cq, (O) >> 3)                                                                \
    }
// the exact variant (a batch holding a splat with an indefinite conic): S6's power > 0 skip as well; the same q, hence the
// same AL bit for bit, for every splat with A, C' >= 0
#define SGS_ALPHA_X(O, N, AL)                                                                          \
    float AL;                                                                                          \
    {                                                                                                  \
        const float U = __builtin_fmaf(-(N##1).x, ly, __builtin_fmaf(-(N##0).y, lx, (N##0).x));        \
        const float V = __builtin_fmaf(-(N##2).x, ly, (N##1).y);                                       \
        const float Vs = (N##2).x < 0.0f ? -V : V;                             /* V^2 carries the sign of C' */ \
        const float q = __builtin_fmaf(U, U, __builtin_fmaf(Vs, V, (N##2).y));                         \
        const float q2 = __builtin_fmaf(U, U, Vs * V);                         /* the sign S6 tests */ \
        const bool valid = q < cq && q2 >= 0.0f;                                                       \
        const float e = SGS_SAT(SGS_EXP2(-q));                                                         \
        AL = valid ? e : 0.0f;                                                                         \
        SGS_PROF_EVAL(valid, (O) >> 3)                                                                 \
    }
#define SGS_APPLY(N, AL)                                                                               \
    {                                                                                                  \
        float wgt = (AL) * Tm;                                                  /* alpha T */          \
        const float tt = __builtin_fmaf(-wgt, amax, Tm);                        /* alpha_max T (1 - alpha) */ \
        const float lv = SGS_SAT(__builtin_fmaf(tt, big, nt_big));              /* 0: ends here, or ended before */ \
        wgt *= lv;                                 /* the splat that would end the pixel is not blended */ \
        Tm = tt * lv;                                                                                  \
        C0 = __builtin_fmaf(wgt, (N##3).x, C0); C1 = __builtin_fmaf(wgt, (N##3).y, C1); C2 = __builtin_fmaf(wgt, (N##4).x, C2); \
        if (AUX) Dz = __builtin_fmaf(wgt, (N##4).y, Dz);      /* expected depth (template instantiation only) */ \
        if (TF) Wsum += wgt;                                                                           \
    }
#define SGS_REPLAY(O, AL)                                                                              \
    {                                                                                                  \
        if (__ballot(Ts > 0.0f) != 0ull) last = ((O) >> 3) + 1u;                                       \
        const float tt = __builtin_fmaf(-((AL) * Ts), amax, Ts);                                       \
        Ts = tt * SGS_SAT(__builtin_fmaf(tt, big, nt_big));                                        \
    }
#define SGS_TRIP(ALPHA, o0, o1, o2, o3)                                                                \
    const float Tb = Tm;                                                                               \
    SGS_LOAD(o0, sa) SGS_LOAD(o1, sb) SGS_LOAD(o2, sc) SGS_LOAD(o3, sd)                                \
    ALPHA(o0, sa, al0) ALPHA(o1, sb, al1) ALPHA(o2, sc, al2) ALPHA(o3, sd, al3)                        \
    SGS_APPLY(sa, al0) SGS_APPLY(sb, al1) SGS_APPLY(sc, al2) SGS_APPLY(sd, al3)                        \
    if (__ballot(Tm > 0.0f) == 0ull) {                                                                 \
        /* the wave's last pixel ended in this trip: replay it to find the splat that did it */        \
        if (STATS) {                                                                                   \
            float Ts = Tb; unsigned last = 0u;                                                         \
            SGS_REPLAY(o0, al0) SGS_REPLAY(o1, al1) SGS_REPLAY(o2, al2) SGS_REPLAY(o3, al3)            \
            used = base + last;                                                                        \
        }                                                                                              \
        wave_done = true; break;                                                                       \
    }
// multi-batch groups: walk the quadrant's 64-bit masks with scalar bit scans
#define SGS_SCAN_LOOP(ALPHA)                                                                           \
        for (int gw = 0; gw < 4 && !wave_done; ++gw) {                                                 \
            unsigned long long mm = uniform_u64(s_ball[par][wave][gw]);   /* splats with a footprint in this quadrant */ \
            const unsigned gwb = (unsigned)gw * 64u;                                                   \
            while (mm != 0ull) {                                                                       \
                SGS_NEXT(o0) SGS_NEXT(o1) SGS_NEXT(o2) SGS_NEXT(o3)                                    \
                SGS_TRIP(ALPHA, o0, o1, o2, o3)                                                        \
            }                                                                                          \
        }
#define SGS_BLEND_WAVE_SCAN()                                                                          \
    if (__ballot(Tm > 0.0f) != 0ull) {                                                                 \
        bool wave_done = false;                                                                        \
        if (!hyper) { SGS_SCAN_LOOP(SGS_ALPHA_F) } else { SGS_SCAN_LOOP(SGS_ALPHA_X) }                 \
        if (STATS) used = Tm > 0.0f ? base + m : used;   /* still live: the whole batch counts as examined */ \
    }
// single-batch groups (the common case): the wave first compacts ITS quadrant's splats into a private list of
// staging offsets (in the idle s_sorted storage) — then the loop has no bit scans on the CU-shared scalar unit,
// no address moves, and one ragged tail per batch instead of one per 64-splat mask word
#define SGS_LIST_LOOP(ALPHA)                                                                           \
        for (unsigned k = 0; k < cntq; k += 4) {                                                       \
            const uint4 pk = *reinterpret_cast<const uint4*>(lst + k);                                 \
            const unsigned o0 = pk.x, o1 = pk.y, o2 = pk.z, o3 = pk.w;                                 \
            SGS_TRIP(ALPHA, o0, o1, o2, o3)                                                            \
        }
#define SGS_BLEND_WAVE_LIST()                                                                          \
    if (__ballot(Tm > 0.0f) != 0ull) {                                                                 \
        unsigned* const lst = s_sorted + (unsigned)wave * (SGS_BATCH + 4);                             \
        unsigned cntq = 0;                                                                             \
        for (int gw = 0; gw < 4; ++gw) {                                                               \
            const unsigned long long mq = uniform_u64(s_ball[par][wave][gw]);                          \
            if ((mq >> lane) & 1ull)                                                                   \
                lst[cntq + lanes_below(mq)] = ((unsigned)gw * 64u + (unsigned)lane) << 3;        \
            cntq += (unsigned)__popcll(mq);                                                            \
        }                                                                                              \
        if (lane < 4) lst[cntq + (unsigned)lane] = (unsigned)(SGS_BATCH << 3);         /* inert tail */ \
        wave_lds_sync();                                                                               \
        bool wave_done = false;                                                                        \
        SGS_PROF_LIST((unsigned)__popcll(__ballot(Tm > 0.0f)), cntq)                                   \
        if (!hyper) { SGS_LIST_LOOP(SGS_ALPHA_F) } else { SGS_LIST_LOOP(SGS_ALPHA_X) }                 \
        SGS_PROF_LIST_END()                                                                            \
        (void)wave_done;                                                                               \
        if (STATS) used = Tm > 0.0f ? base + m : used;                                                 \
    }
// staging: write splat J from its record (A_ = x,y,a,ak  B_ = c,nlo,r,g  C_ = b,depth,hx,hy) with the tile-relative constants
// m = a rx + a k ry and c ry.  A splat whose conic is not positive semi-definite (c < 0; NaN) flags the batch.
#define SGS_STAGE(J, A_, B_, C_, RX, RY)                                                               \
    {                                                                                                  \
        s_p0[J] = make_float2(__builtin_fmaf(A_.w, RY, A_.z * (RX)), A_.z); s_p1[J] = make_float2(A_.w, B_.x * (RY)); \
        s_p2[J] = make_float2(B_.x, B_.y); s_p3[J] = make_float2(B_.z, B_.w);                          \
        s_p4[J] = make_float2(C_.x, C_.y);                                                             \
        if (!(A_.z >= 0.0f) || !(B_.x >= 0.0f)) s_hyper[par] = 1u;                                     \
    }
#define SGS_STAGE_DUMMY()                                                                              \
    {                                                                                                  \
        s_p0[SGS_BATCH] = make_float2(0.f, 0.f); s_p1[SGS_BATCH] = make_float2(0.f, 0.f);              \
        s_p2[SGS_BATCH] = make_float2(0.f, 1.0e30f); s_p3[SGS_BATCH] = make_float2(0.f, 0.f);          \
        s_p4[SGS_BATCH] = make_float2(0.f, 0.f);                                                       \
    }
// ---- which 8x8 quadrants of the tile can a splat reach? -------------------------------------------------
// The axis-aligned extent of {alpha >= alpha_min} is a loose test for elongated splats (measured: 27 % of the
// (wave, splat) evaluations it let through had no pixel inside the cut-off).  This is the exact one: the minimum of
//     q2(d) = A dx^2 + B dx dy + C dy^2        (B = 2 A k, C = A k^2 + C' of the staged record; q2 <= qmax  <=>  alpha >= alpha_min)
// over the rectangle of a quadrant's pixel centres is 0 if the centre lies inside, else it is attained on one of
// the (at most two) edges facing the centre, where q2 is a 1-D parabola (minimiser -B e / 2C, clamped to the edge).  The comparison carries a
// bound on the rounding error of both this evaluation and the per-pixel one (a few ulps of the largest possible sum
// of terms inside the quadrant), so a quadrant holding a pixel the blend would accept is never rejected (a NaN
// anywhere accepts).
__device__ __forceinline__ float sgs_edge_min(float e, float d0, float d1, float P_, float B_, float R_, float k) {
    const float t = __builtin_amdgcn_fmed3f(k * e, d0, d1);
    return (P_ * e + B_ * t) * e + (R_ * t) * t;
}
__device__ __forceinline__ unsigned sgs_quadrant_hits(float rx, float ry, float A, float k, float Cp, float qmax, const float4* lrect) {
    // rx, ry: splat centre relative to the tile's first pixel; lrect[q] = (x first, x last, y first, y last): the rectangle of
    // quadrant q's pixel centres that are still LIVE, in tile pixels — the whole quadrant ([0,7] / [8,15]) at first.
    // The record holds the completed square A (dx + k dy)^2 + C' dy^2: B = 2 A k, C = A k^2 + C'.
    const float Ak = A * k, B = 2.0f * Ak, C = __builtin_fmaf(Ak, k, Cp);
    const float kv = -Ak * SGS_RCP(C), kh = -k;        // the edge minimisers -B e / 2C and -B e / 2A (an ulp off changes the minimum by O(ulp^2))
    const float aB = fabsf(B);
    unsigned bits = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 lr = lrect[q];
        const float xa = lr.x - rx, xb = lr.y - rx, ya = lr.z - ry, yb = lr.w - ry;
        // (coordinates are relative to the centre: it lies between two edges when they straddle 0)
        const bool in_x = xa <= 0.0f && xb >= 0.0f, in_y = ya <= 0.0f && yb >= 0.0f;
        // q2 is convex with its minimum at the centre, so over a rectangle that does not hold the centre it is least on an
        // edge that FACES the centre (at the minimiser -grad q2 is an outward normal, and q2 decreases towards the centre):
        // the nearer vertical edge unless the centre is within the x range, the nearer horizontal one unless within the y range
        const float ex = xa > 0.0f ? xa : xb, ey = ya > 0.0f ? ya : yb;
        const float mv = in_x ? 3.0e38f : sgs_edge_min(ex, ya, yb, A, B, C, kv);
        const float mh = in_y ? 3.0e38f : sgs_edge_min(ey, xa, xb, C, B, A, kh);
        const float m = fminf(mv, mh);
        // every term of any evaluation inside this quadrant (edge minima here, per-pixel q2 in the blend) is bounded
        // by S: a few ulps of S cover the rounding of both sides of the comparison
        const float fx = fmaxf(fabsf(xa), fabsf(xb)), fy = fmaxf(fabsf(ya), fabsf(yb));      // farthest pixel of the quadrant
        const float S = A * fx * fx + aB * fx * fy + C * fy * fy;
        const float thr = qmax + 8.0e-6f * S + 1.0e-5f;
        if ((in_x && in_y) || !(m > thr)) bits |= 1u << q;
    }
    return bits;
}
// (the record holds the roots a = sqrt(A), a k, c = +-sqrt(|C'|); the test's own slack of ~130 ulps covers these conversions)
__device__ __forceinline__ unsigned sgs_quadrant_hits_roots(float rx, float ry, float a, float ak, float c, float qmax, const float4* lrect) {
    const float k = a > 0.0f ? ak * SGS_RCP(a) : 0.0f;
    return sgs_quadrant_hits(rx, ry, a * a, k, c * fabsf(c), qmax, lrect);
}
// the axis-aligned extent (hx, hy around the centre) against the four live rectangles: the cheap test in front of the exact one
__device__ __forceinline__ unsigned sgs_quadrant_extent(float rx, float ry, float hx, float hy, const float4* lrect) {
    unsigned bits = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 lr = lrect[q];
        bits |= (rx - hx <= lr.y && rx + hx >= lr.x && ry - hy <= lr.w && ry + hy >= lr.z) ? 1u << q : 0u;     // (an empty rect: first > last)
    }
    return bits;
}
// the rectangle of a quadrant's live pixels from the wave's ballot (lane = 8 row + column): (first col, last col, first row, last row)
__device__ __forceinline__ float4 sgs_live_rect(unsigned long long m, float x0, float y0) {
    if (m == 0ull) return make_float4(1.0e30f, -1.0e30f, 1.0e30f, -1.0e30f);        // empty: nothing can reach it
    unsigned long long c = m | (m >> 32); c |= c >> 16; c |= c >> 8;                // OR of the 8 rows -> live columns
    const unsigned cols = (unsigned)c & 0xffu;
    const int ca = __ffsll((long long)cols) - 1, cb = 31 - __clz((int)cols);
    const int ra = (__ffsll((long long)m) - 1) >> 3, rb = (63 - __clzll((long long)m)) >> 3;
    return make_float4(x0 + (float)ca, x0 + (float)cb, y0 + (float)ra, y0 + (float)rb);
}

#define SGS_BATCH 256                 // splats staged per batch (k_tile_render)
// ------------------------------------------------------------------------------------------------
// S5 + S6 fused: per-tile LAZY depth sort feeding the front-to-back composite.  One workgroup per 16x16 tile,
// one lane per pixel, each of the four waves owns an 8x8 quadrant.
//
// Measured on the 3 M-Gaussian scene: a tile's queue holds ~500 records on average (up to 15 k), but its pixels
// saturate after ~140 — most of a fully sorted queue would never be read.  So the workgroup of a tile
//   1. partitions its queue into SGS_NB depth buckets with one MSD pass on the fp32 depth bits
//      (bucket = (bits >> 18) - (bits(near) >> 18): 32 buckets per binade of view depth, so buckets are fine near
//      the camera where it matters); counters in LDS.  Queues of <= SGS_QCAP records are read from HBM once and
//      live in LDS from then on, bucket-contiguous; longer ones stay in HBM and a WINDOW of whole buckets is
//      filled by one coalesced re-scan and serves several groups (queues of <= SGS_GROUP records skip the partition);
//   2. takes the buckets front to back in groups of about SGS_GROUP records.  A group of <= SGS_BATCH records (the
//      common case) is handled in one go: every lane owns a record, issues the gather of its 48-B splat, ranks the
//      record inside its own bucket while the loads are in flight, and stores the splat at staging[rank] together
//      with the quadrants it can reach (axis-aligned extent, then the exact ellipse/rectangle test); each wave
//      compacts its quadrant's splats into a private list and blends them four per trip.  Larger groups (one
//      oversized bucket) are rank-sorted with 2-4 records per lane and streamed through the staging area in
//      batches; a bucket beyond SGS_QCAP (thousands of splats within 2 % of one depth) is radix-sorted through HBM;
//   3. stops as soon as every pixel of the tile has terminated.
// Workgroup b renders position b of k_tile_scan's longest-queue-first order.
#define SGS_NB 256
#ifndef SGS_PART_MIN
#define SGS_PART_MIN 64               // queues up to this long are one group ranked all pairs; longer ones are partitioned into depth buckets
#endif
#ifndef SGS_TAIL_AFTER
#define SGS_TAIL_AFTER 3u
#endif
#ifndef SGS_GROUP
#define SGS_GROUP 192                 // soft cap of a group: buckets are added while the total stays below (192 vs 256: -1.3 % per frame, r02y)
#endif
#ifndef SGS_QCAP
#define SGS_QCAP 1024                 // queues up to this long live entirely in LDS; also the rank sort's hard cap
#endif
#ifndef SGS_PASS_R
#define SGS_PASS_R 4                  // records per lane and trip of a pass over a long queue (loads in flight: the passes are latency-bound)
#endif
#define SGS_RANK_BUCKET_MAX 64        // bucket-local ranking walks at most this many records per lane
#ifndef SGS_REFINE_SPAN
#define SGS_REFINE_SPAN 1024          // records of a long queue's ordered copy that one refinement re-partitions (at least the bucket that asked)
#endif
#ifndef SGS_DEEP_LIVE
#define SGS_DEEP_LIVE 48u             // ... and only while at most this many of its 256 pixels are live (more: too many records survive the cull)
#endif
#ifndef SGS_DEEP_AFTER
#define SGS_DEEP_AFTER 3u             // a tile past this many batches culls whole resident windows against its live pixels before it ranks or stages anything
#endif
#ifndef SGS_LAZY_WINDOWS
#define SGS_LAZY_WINDOWS 2u           // windows a long queue fills by scanning itself before its bucket-ordered copy is made (k_tile_render)
#endif

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    // the value is wave-uniform by construction; tell the compiler so the bit scan stays on the SALU
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// AUX: expected depth + coverage outputs.  STATS: D_f bookkeeping.  TF: the final transmittance of stopped pixels is needed
// (AUX, or a background that is not black) — one more add per (pixel, splat).
template <bool AUX, bool STATS, bool TF>
// workgroups per CU: five (96 VGPRs, no spills, 32 KB of LDS each) render as fast as six (80 VGPRs, 10-19 spilled) alone
// and 2-3 % faster with frames in flight (r03d/g/h)
#ifndef SGS_RENDER_WGS
#define SGS_RENDER_WGS 5
#endif
__device__ __forceinline__ void render_tile(const FrameSlot& S, const unsigned bx) {
    static_assert(TF || !AUX, "the coverage output needs the final transmittance");
    const FrameParams& P = S.P;
    const uint4* __restrict__ tile_order = S.tile_order;
    const unsigned long long* __restrict__ rec = S.rec; unsigned long long* rec_w = S.rec;    // (a long queue lends its own storage as scratch once it has been copied in bucket order)
    unsigned long long* alt = S.alt; unsigned long long* part = S.part; unsigned* sorted_out = S.sorted_out;
    const Splat* __restrict__ splats = S.splats;
    float* __restrict__ out_rgb = S.out_rgb; float* __restrict__ out_aux = S.out_aux;
    FrameStatus* st = S.st; unsigned long long* prof = S.tile_prof;
    (void)prof; (void)out_aux;
    // LDS: the (partitioned) queue or the current group, the staged batch, bucket tables.
    __shared__ unsigned long long s_q[SGS_QCAP + 8 + SGS_RANK_BUCKET_MAX];  // records: depth bits << 32 | slot (+8 sentinels, + slack
                                                                          // for the masked reads past a short bucket)
    __shared__ __attribute__((aligned(16))) float2 s_arena[5 * (SGS_BATCH + 1)];
    static_assert(sizeof(s_arena) >= sizeof(SortShared), "arena");
    float2* const s_p0 = s_arena;                                     // blend phase: the staged batch (+1 inert dummy),
    float2* const s_p1 = s_arena + (SGS_BATCH + 1);                   // five arrays of pairs at one common offset
    float2* const s_p2 = s_arena + 2 * (SGS_BATCH + 1);
    float2* const s_p3 = s_arena + 3 * (SGS_BATCH + 1);
    float2* const s_p4 = s_arena + 4 * (SGS_BATCH + 1);
    SortShared& sh = *reinterpret_cast<SortShared*>(s_arena);   // HBM radix path only (never while blending)
    __shared__ __attribute__((aligned(16))) unsigned s_sorted[(SGS_QCAP + 16) > 4 * (SGS_BATCH + 4) ? (SGS_QCAP + 16) : 4 * (SGS_BATCH + 4)];   // the group's slots in (depth, index) order; single-batch
                                                                              // groups: the four waves' splat lists
    __shared__ unsigned s_bcnt[SGS_NB];               // bucket counts, then scatter cursors
    __shared__ unsigned s_ne_end[SGS_NB];             // non-empty buckets, in order: end offset in the queue
    __shared__ unsigned short s_ne_bkt[SGS_NB];       //                              bucket index
    __shared__ unsigned s_wsum[4], s_wne[4], s_kmn[4], s_kmx[4];
    __shared__ unsigned s_c_bcnt[SGS_NB], s_c_end[SGS_NB];    // the queue's own partition while a refinement is the current one
    __shared__ unsigned short s_c_bkt[SGS_NB];
    __shared__ unsigned s_cs[12];                              // ... and its scalars / the refinement's state (CS_*)
    __shared__ unsigned long long s_ball[2][4][4];    // [batch parity][quadrant][gathering wave]
    __shared__ unsigned s_any[2];                     // pixels still unfinished after batch (by parity)
    __shared__ unsigned s_hyper[2];                   // the batch holds a splat with an indefinite conic: exact trips (by parity)
    // the rectangle of every quadrant's pixels that are still live (tile pixels; written by the quadrant's wave after each
    // batch): a tile that keeps consuming batches for a few pixels that never saturate — a gap in the scene, a window —
    // then stages and walks only the splats that can reach THOSE pixels, not the whole 8x8 quadrant
    __shared__ float4 s_lrect[4];
    __shared__ unsigned s_used[4];
    __shared__ unsigned long long s_skey[SGS_BATCH + 8];      // deep tiles: the records of a window that can reach a live pixel (+ sentinels)
    __shared__ unsigned s_nsurv;
#ifdef SGS_TILE_PROF
    // profiling build only (lib/libsage_gs_prof.so): per-tile shader-clock cycles of each phase
    unsigned long long pt0 = clock64(), pt_part = 0, pt_sort = 0, pt_blend = 0, pn_groups = 0, pn_batches = 0, ptm;
    unsigned long long pt_rank = 0, pt_bar1 = 0, pt_stage = 0, pt_job = 0, pt_rec = 0;     // sub-phases of the single-batch path (pt_sort = the rest: barrier 2)
    unsigned pe_eval = 0, pe_empty = 0, pe_valid = 0, pe_useful = 0, pe_dead = 0, pe_few = 0;    // (dead: no LIVE pixel inside the cut-off; few: one or two)
    unsigned long long pn_deep_try = 0, pn_deep_ok = 0, pn_deep_surv = 0;                          // deep-tile culls attempted / taken, survivors of the taken ones
    const unsigned long long prt0 = wall_clock64();      // 100 MHz, common to all XCDs
    __shared__ unsigned s_pe[8];
    __shared__ unsigned long long s_pw[8];
    unsigned pw_hist[5] = {0u, 0u, 0u, 0u, 0u}; unsigned long long pw_t0 = 0, pw_list_cyc = 0;
    if (threadIdx.x < 8) { s_pe[threadIdx.x] = 0; s_pw[threadIdx.x] = 0ull; }
#define SGS_PROF_MARK(acc) do { unsigned long long now_ = clock64(); acc += now_ - ptm; ptm = now_; } while (0)
#else
#define SGS_PROF_MARK(acc) do { } while (0)
#endif
    if (st->overflow) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tid_entry = tid;
    // blocks take the tiles longest queue first (k_tile_scan's order)
    const unsigned ntiles = (unsigned)((P.row_end - P.row_begin) * P.gx);
    if (bx >= ntiles) return;            // workgroup-uniform
    const uint4 job = tile_order[bx];                // (tile, first record, queue length) from k_tile_scan
#ifdef SGS_TILE_PROF
    __builtin_amdgcn_s_waitcnt(0); asm volatile("" ::: "memory");
    pt_job = clock64() - pt0;      // kernel entry -> job (and the status word) arrived
#endif
    const unsigned tile = job.x;
    const unsigned tile_x = tile % (unsigned)P.gx, tile_y = tile / (unsigned)P.gx;
    // (px, py: CELL pixels — the frame's own pixels unless the frame is rendered through fine tiles, sgs_common.h: then pixel (i, j) of the
    //  frame is cell pixel (i << z, j << z) and the other lanes have no pixel)
    const unsigned zf = (unsigned)fine_shift(P), zmask = (1u << zf) - 1u;
    const unsigned px = tile_x * 16u + (unsigned)(wave & 1) * 8u + (unsigned)(lane & 7);
    // tile_y counts the rows this call owns; the pixels it covers are those of frame row tile_y * stride + phase,
    // and it is stored at row tile_y of the (compact, when stride > 1) output image
    const unsigned frame_y = frame_row_of(P, tile_y);
    const unsigned in_y = (unsigned)(wave >> 1) * 8u + (unsigned)(lane >> 3);
    const unsigned py = frame_y * 16u + in_y;
    const bool inside = ((px | py) & zmask) == 0u && (px >> zf) < (unsigned)P.width && (py >> zf) < (unsigned)P.height;      // (again at the end, for the store)
    const float tile_fx = (float)(tile_x * 16u), tile_fy = (float)(frame_y * 16u);
    // per-frame constants of the trip (the header of the blend macros): all in VGPRs, an SGPR operand costs 1.6 issues
    float amax = P.alpha_max, big = SGS_BIG;
    SGS_PIN_VGPR(amax); SGS_PIN_VGPR(big);
    const float cq = __log2f(P.alpha_max) - __log2f(P.alpha_min);      // alpha >= alpha_min  <=>  q <= cq
    float cq_big = cq * SGS_BIG, nt_big = -(P.alpha_max * P.t_min) * SGS_BIG;
    SGS_PIN_VGPR(cq_big); SGS_PIN_VGPR(nt_big);
    const bool full_sort = (P.flags & 8u) != 0u;       // SGS_FLAG_FULL_SORT (tests): order the whole queue
    const bool loose_cull = (P.flags & 32u) != 0u;     // SGS_FLAG_LOOSE_CULL (tests): extent-only quadrant test
    const bool deep_ok = (P.flags & 256u) == 0u;      // SGS_FLAG_NO_DEEP (tests, A/B): never cull a window against the live pixels

    const unsigned beg = job.y, n = job.z;            // the tile's 8 per-XCD sub-queues are adjacent: one queue
    float Tm = inside ? amax : 0.0f;     // alpha_max x transmittance; 0 = finished (or outside the image): takes nothing more
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dz = 0.f, Wsum = 0.f;
    unsigned used = 0;                   // queue position up to which this pixel examined records (D_f bookkeeping)

    // ---- 1. MSD partition into depth buckets (queues longer than one group only) -----------------
    // n <= SGS_QCAP: the queue is read from HBM exactly once (4 records per lane, held in registers
    // between the histogram and the scatter) and lives in LDS from then on.  Longer queues are
    // partitioned through HBM into the alt buffers and groups are loaded into s_q one at a time.
    const bool parted = n > SGS_PART_MIN;
    const bool in_lds = n <= SGS_QCAP;
    // The SGS_NB buckets span the depth range of THIS tile's queue (keys are the bits of positive floats: monotone), not a
    // fixed grid of the frustum: a tile mostly sees one or two surfaces, its keys sit in a sliver of the depth range, and
    // buckets cut from that sliver hold a handful of records each — the bucket-local rank below is then one or two trips
    // instead of ten, and a bucket too long for one batch (the slow paths) all but disappears.  Any monotone map is correct:
    // bucket(key) = 0 at or below klo, (key - klo) >> ksh above it, capped at SGS_NB - 1.
    // A LONG queue (n > SGS_QCAP) is copied ONCE into `part` in bucket order (partition_range); a bucket of it that is still longer than one
    // batch — a 640x480 frame's tile sees nine times the scene area of a 1080p tile: queues of 10-60 k records, hundreds per bucket — is
    // partitioned again over ITS key range and ITS slice of the copy (refinement, below; the queue's own partition waits meanwhile).
    unsigned klo = 0u, ksh = 0u, pbase = 0u;                    // pbase: queue position of the current partition's first bucket
    // A long queue's bucket that is refined: the partition of ITS slice [.., fine_hi) is the current one, the queue's own partition waits in
    // the s_c_* tables until the slice has been consumed.  This state (and the saved klo, ksh, pbase, n_ne, e_next) lives in LDS, s_cs[]: as
    // registers its handful of values stayed live across the blend — the kernel's register peak — and cost 19 more spilled SGPRs and 3 VGPRs
    // (r04n: +3 us on the 1080p composite, where one tile in six has a long queue and one in a thousand is ever refined).
    // The ordered copy is made LAZILY (s_cs[7]): most long queues saturate their pixels inside the first window, which one scan of the queue
    // fills.  A queue that asks for a second window, a refinement or the HBM sort is a tile that reads deep: it pays two more passes once
    // (partition_range again: same buckets, same positions) and nothing per window after that.
    enum { CS_KLO = 0, CS_KSH, CS_PBASE, CS_NNE, CS_ENEXT, CS_FINE_HI, CS_FINE, CS_ORDERED, CS_B1, CS_COUNT };
#define SGS_CS(k) __builtin_amdgcn_readfirstlane(s_cs[k])
#define SGS_BUCKET_OF(key) ((key) <= klo ? 0u : min((unsigned)(SGS_NB - 1), ((key) - klo) >> ksh))
    if (tid < 2) { s_any[tid] = 0; s_hyper[tid] = 0; }
    if (tid < 12) s_cs[tid] = 0u;
    {
        const unsigned long long im = __ballot(inside);
        if (lane == 0) s_lrect[wave] = sgs_live_rect(im, (float)((wave & 1) * 8), (float)((wave >> 1) * 8));
    }
    if (tid < 64) reinterpret_cast<unsigned*>(&s_ball[0][0][0])[tid] = 0u;   // both parities: 2 x 4 quadrants x 4 x 64 bits
    unsigned n_ne = 0;                                   // non-empty buckets (uniform)
    // exclusive scan of the SGS_NB bucket counts in s_bcnt (-> cursors, positions from pbase) + ordered compaction of the
    // non-empty buckets; all threads, ends with a barrier
    auto scan_buckets = [&]() {
        int tid_l = tid_entry; SGS_PIN_VGPR(tid_l);          // (a fresh copy of the thread index: see the group loop)
        const int tid = tid_l, lane = tid & 63, wave = tid >> 6;
        const unsigned c = s_bcnt[tid];
        const unsigned incl = wave_incl_scan(c, lane);
        const unsigned long long nem = __ballot(c != 0u);
        if (lane == 63) s_wsum[wave] = incl;
        if (lane == 0) s_wne[wave] = (unsigned)__popcll(nem);
        __syncthreads();
        unsigned ex = pbase + incl - c, nb = 0;
        n_ne = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) { ex += s_wsum[w]; nb += s_wne[w]; } n_ne += s_wne[w]; }
        s_bcnt[tid] = ex;                             // scatter cursor
        if (c != 0u) {
            const unsigned p = nb + lanes_below(nem);
            s_ne_end[p] = ex + c; s_ne_bkt[p] = (unsigned short)tid;
        }
        __syncthreads();
    };
    // Long queues: the partition of the records src[beg + a .. beg + b) under the current (klo, ksh) — histogram, scan (cursors from pbase = a),
    // and the BUCKET-ORDERED COPY of those records into dst[beg + a ..): the cursors place every record, as the in-LDS scatter does for a
    // short queue.  A window of the queue is then a contiguous slice of that copy — one coalesced read of <= 1024 records — and a bucket too
    // long for one batch is refined by partitioning ITS slice again (below), not the queue.  Rounds 1-3 re-scanned the WHOLE queue per window
    // and three times per refinement: a 640x480 tile sees nine times the scene area of a 1080p one, a 320x240 tile thirty-six times; their
    // queues hold 10-100 k records, their pixels take 30-200 batches to saturate, and the re-scans were quadratic (r04h / r04l: 70 windows
    // and 78 refinements = 12.7 M of the slowest 320x240 tile's 14.9 M cycles; 0.8 M of 2.0 M at 640x480).  (Scattered 8-byte stores
    // amplify — 32 B each — but this is one pass per partition.)  All threads.
    // scatter_only: the histogram and the cursors of this very partition are still in place (nothing, or only the buckets up to `skip`, has been
    // placed with them: the queue's first window) — only the copy is made, of the buckets after `skip`.
    auto partition_range = [&](const unsigned long long* src, unsigned a, unsigned b, unsigned long long* dst, bool scatter_only, int skip) {
        int tid_l = tid_entry; SGS_PIN_VGPR(tid_l);
        const int tid = tid_l;
        if (!scatter_only) {
        s_bcnt[tid] = 0;
        __syncthreads();
        for (unsigned i0 = a; i0 < b; i0 += 256u * SGS_PASS_R) {                 // SGS_PASS_R loads in flight per lane
            unsigned long long x[SGS_PASS_R];
#pragma unroll
            for (int r = 0; r < SGS_PASS_R; ++r) {
                const unsigned i = i0 + (unsigned)tid + 256u * (unsigned)r;
                x[r] = i < b ? src[beg + i] : ~0ull;
            }
#pragma unroll
            for (int r = 0; r < SGS_PASS_R; ++r)
                if (i0 + (unsigned)tid + 256u * (unsigned)r < b) atomicAdd(&s_bcnt[SGS_BUCKET_OF((unsigned)(x[r] >> 32))], 1u);
        }
        __syncthreads();
        scan_buckets();
        }
        if (dst == nullptr) return;      // (histogram + scan only: the queue's FIRST window is filled by a scan of the queue, see below)
        for (unsigned i0 = a; i0 < b; i0 += 256u * SGS_PASS_R) {
            unsigned long long x[SGS_PASS_R];
#pragma unroll
            for (int r = 0; r < SGS_PASS_R; ++r) {
                const unsigned i = i0 + (unsigned)tid + 256u * (unsigned)r;
                x[r] = i < b ? src[beg + i] : ~0ull;
            }
#pragma unroll
            for (int r = 0; r < SGS_PASS_R; ++r) {
                const unsigned bk = SGS_BUCKET_OF((unsigned)(x[r] >> 32));
                if (i0 + (unsigned)tid + 256u * (unsigned)r < b && (int)bk > skip) dst[beg + atomicAdd(&s_bcnt[bk], 1u)] = x[r];
            }
        }
        __syncthreads();                 // (cursors are bucket ENDS now, as after the in-LDS scatter: the ranking reads them)
    };
    {
        unsigned long long rq[4];                      // the queue (n <= SGS_QCAP), or its first SGS_QCAP records: a sample of its depths
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned i = (unsigned)tid + 256u * (unsigned)r;
            rq[r] = i < n ? rec[beg + i] : ~0ull;
        }
#ifdef SGS_TILE_PROF
        __builtin_amdgcn_s_waitcnt(0); asm volatile("" ::: "memory");
        pt_rec = clock64() - pt0;  // ... -> the queue's records arrived (queues of <= SGS_QCAP)
#endif
        if (!parted) {
            s_q[tid] = rq[0];                                              // n <= 256: one group, padded with ~0
            if (tid < 8) s_q[SGS_GROUP + tid] = ~0ull;
        } else {
            s_bcnt[tid] = 0;
            {   // the depth range of the records held (all of them, or the sample of a long queue: widened by half on both sides)
                unsigned kmn = 0xffffffffu, kmx = 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((unsigned)tid + 256u * (unsigned)r < n) {
                        const unsigned key = (unsigned)(rq[r] >> 32);
                        kmn = key < kmn ? key : kmn; kmx = key > kmx ? key : kmx;
                    }
                kmn = wave_min(kmn); kmx = wave_max(kmx);
                if (lane == 0) { s_kmn[wave] = kmn; s_kmx[wave] = kmx; }
            }
            __syncthreads();
            {
                unsigned kmn = s_kmn[0], kmx = s_kmx[0];
#pragma unroll
                for (int w = 1; w < 4; ++w) { kmn = s_kmn[w] < kmn ? s_kmn[w] : kmn; kmx = s_kmx[w] > kmx ? s_kmx[w] : kmx; }
                if (!in_lds) {
                    const unsigned half = (kmx - kmn) >> 1;
                    kmn = kmn > half ? kmn - half : 0u; kmx = kmx < 0xffffffffu - half ? kmx + half : 0xffffffffu;
                }
                const unsigned span = kmx - kmn;                                  // (span >> ksh) < SGS_NB
                klo = kmn; ksh = span < (unsigned)SGS_NB ? 0u : 24u - (unsigned)__clz((int)span);
            }
            if (in_lds) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((unsigned)tid + 256u * (unsigned)r < n)
                        atomicAdd(&s_bcnt[SGS_BUCKET_OF((unsigned)(rq[r] >> 32))], 1u);
                __syncthreads();
                scan_buckets();
            } else partition_range(rec, 0u, n, nullptr, false, -1);
            if (in_lds) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((unsigned)tid + 256u * (unsigned)r < n) {
                        const unsigned bk = SGS_BUCKET_OF((unsigned)(rq[r] >> 32));
                        s_q[atomicAdd(&s_bcnt[bk], 1u)] = rq[r];
                    }
                if (tid < 8) s_q[n + tid] = ~0ull;
            }
            // longer queues stay where they are: s_q holds a WINDOW of whole buckets (<= SGS_QCAP records) that one
            // coalesced re-scan of the queue fills, placing records with the bucket cursors; several groups are
            // served from a window (a scatter of 8-byte records into HBM measured 8x write amplification)
        }
    }
    __syncthreads();
#ifdef SGS_TILE_PROF
    ptm = clock64(); pt_part = ptm - pt0;
#endif

    // ---- 2. groups of buckets, front to back -------------------------------------------------------
    unsigned it = 0;                     // batch counter (parity of the LDS flags)
    bool tile_done = false;
    unsigned e_next = 0, lo = 0;         // next non-empty bucket (index into s_ne_*) / its queue position
    unsigned win_lo = 0, win_hi = in_lds ? n : 0u;   // queue range resident in s_q (the whole queue when it fits)
    unsigned n_refine = 0u, no_refine_at = 0xffffffffu;
    unsigned n_live_px = 256u;           // pixels of the tile still live after the last batch (uniform)
    unsigned deep_hold = 0u;             // groups for which the deep-tile cull is not attempted (uniform): its last attempt left more than a batch
    unsigned n_fill = 0u;                // windows of a long queue filled so far (uniform; reported by the profiling build)
    (void)n_refine; (void)n_fill;
    while (lo < n && (!tile_done || full_sort)) {
        // Everything a group derives from the thread index is derived HERE, per group, from a copy the compiler cannot see
        // through: hoisted out of the loop, the dozen index expressions, LDS addresses and pixel coordinates below stayed live
        // across the blend (the kernel's register peak) and were spilled in the prologue — 44 bytes per lane, 92 MB of scratch
        // writes per 1080p frame, as much as the kernel's whole algorithmic traffic (r03c).
        int tid_group = tid_entry; SGS_PIN_VGPR(tid_group);
        const int tid = tid_group, lane = tid & 63, wave = tid >> 6;
        if (!in_lds && SGS_CS(CS_FINE) != 0u && lo >= SGS_CS(CS_FINE_HI)) {     // the refined slice has been consumed: the queue's own
            s_bcnt[tid] = s_c_bcnt[tid]; s_ne_end[tid] = s_c_end[tid]; s_ne_bkt[tid] = s_c_bkt[tid];   // partition again, at the bucket after it
            klo = SGS_CS(CS_KLO); ksh = SGS_CS(CS_KSH); pbase = SGS_CS(CS_PBASE); n_ne = SGS_CS(CS_NNE); e_next = SGS_CS(CS_ENEXT);
            win_lo = lo; win_hi = lo;
            __syncthreads();             // (every thread has read the state)
            if (tid == 0) s_cs[CS_FINE] = 0u;
            __syncthreads();
        }
        const float lx = (float)((unsigned)(wave & 1) * 8u + (unsigned)(lane & 7)), ly = (float)((unsigned)(wave >> 1) * 8u + (unsigned)(lane >> 3));
        unsigned hi, e0 = e_next, e1 = e_next;
        if (!parted) hi = n;
        else {
            // The group = as many whole buckets from e0 on as stay within SGS_GROUP records (a group that starts inside
            // the resident window also ends inside it: buckets are placed once); at least one bucket.  s_ne_end is
            // increasing, so the buckets that fit are a prefix: every wave counts them with independent reads and
            // ballots — two LDS round-trips instead of one per bucket.
            // (a tile past its third batch is going to read most of its queue — pixels that never saturate — and what it pays
            //  per batch is mostly fixed: full batches from then on)
            const unsigned gcap = it >= SGS_TAIL_AFTER ? (unsigned)SGS_BATCH : (unsigned)SGS_GROUP;
            const unsigned limit = lo < win_hi ? min(lo + gcap, win_hi) : lo + gcap;
            unsigned fit = 0;
#pragma unroll
            for (int r = 0; r < SGS_NB / 64; ++r) {
                const unsigned e = e0 + (unsigned)(r * 64 + lane);
                fit += (unsigned)__popcll(__ballot(e < n_ne && s_ne_end[e] <= limit));
            }
            e1 = e0 + (fit ? fit - 1u : 0u);
            hi = s_ne_end[e1];
            e_next = e1 + 1;
        }
        if (parted && !in_lds) {
            // Two things a long queue may need before this group can be taken, both ONE partition_range (one instance of its code: the
            // kernel's instruction footprint is what a 1080p frame pays for it):
            //   (1) the ordered copy itself — when the group is not resident and the first window has been used, when its bucket wants
            //       refining, or when it is longer than the LDS holds (HBM sort): partition the queue into `part`, then select the group again;
            //   (2) REFINEMENT — the group is ONE bucket that holds more than a batch: its records are the slice [lo, hi) of the copy; one
            //       pass over the SLICE finds their key range, partition_range cuts 256 buckets from it and orders the slice into `alt`
            //       (instead of sorting the whole bucket: the rank-sort / HBM-radix paths below, kept for the bucket whose keys are all equal).
            const bool want_refine = e1 == e0 && hi - lo > (unsigned)SGS_BATCH && lo != no_refine_at && SGS_CS(CS_FINE) == 0u;
            const bool have_copy = SGS_CS(CS_ORDERED) != 0u;
            const unsigned long long* p_src = rec; unsigned long long* p_dst = part;
            unsigned p_a = 0u, p_b = n;
            bool go = !have_copy && (want_refine || hi - lo > (unsigned)SGS_QCAP || (hi > win_hi && n_fill >= SGS_LAZY_WINDOWS));
            if (!go && want_refine) {
                // the slice that is refined: this bucket and as many of the buckets behind it as keep it within SGS_REFINE_SPAN records — a
                // depth range dense enough to overfill one bucket usually overfills its neighbours too (a wall, a shelf), and one
                // refinement per bucket would pay the fixed cost of a partition (its barriers, the scan, the tables) each time; the span
                // bounds what a refinement costs on a queue of 100 k records
                unsigned rfit = 0;
#pragma unroll
                for (int r = 0; r < SGS_NB / 64; ++r) {
                    const unsigned e = e0 + 1u + (unsigned)(r * 64 + lane);
                    rfit += (unsigned)__popcll(__ballot(e < n_ne && s_ne_end[e] - lo <= (unsigned)SGS_REFINE_SPAN));
                }
                const unsigned ek = e0 + rfit;
                const unsigned hi_r = s_ne_end[ek];
                unsigned kmn = 0xffffffffu, kmx = 0u;
                for (unsigned i0 = lo; i0 < hi_r; i0 += 256u * SGS_PASS_R) {
                    unsigned long long x[SGS_PASS_R];
#pragma unroll
                    for (int r = 0; r < SGS_PASS_R; ++r) {
                        const unsigned i = i0 + (unsigned)tid + 256u * (unsigned)r;
                        x[r] = i < hi_r ? part[beg + i] : ~0ull;
                    }
#pragma unroll
                    for (int r = 0; r < SGS_PASS_R; ++r) {
                        const unsigned key = (unsigned)(x[r] >> 32);
                        if (i0 + (unsigned)tid + 256u * (unsigned)r < hi_r) { kmn = key < kmn ? key : kmn; kmx = key > kmx ? key : kmx; }
                    }
                }
                kmn = wave_min(kmn); kmx = wave_max(kmx);
                __syncthreads();                          // (s_kmn / s_kmx: everyone is past their last use)
                if (lane == 0) { s_kmn[wave] = kmn; s_kmx[wave] = kmx; }
                __syncthreads();
                kmn = min(min(s_kmn[0], s_kmn[1]), min(s_kmn[2], s_kmn[3])); kmx = max(max(s_kmx[0], s_kmx[1]), max(s_kmx[2], s_kmx[3]));
                if (kmx > kmn) {
                    s_c_bcnt[tid] = s_bcnt[tid]; s_c_end[tid] = s_ne_end[tid]; s_c_bkt[tid] = s_ne_bkt[tid];      // the queue's own partition waits
                    if (tid == 0) {
                        s_cs[CS_KLO] = klo; s_cs[CS_KSH] = ksh; s_cs[CS_PBASE] = pbase; s_cs[CS_NNE] = n_ne; s_cs[CS_ENEXT] = ek + 1u;
                        s_cs[CS_FINE_HI] = hi_r; s_cs[CS_FINE] = 1u;
                    }
                    const unsigned span = kmx - kmn;
                    klo = kmn; ksh = span < (unsigned)SGS_NB ? 0u : 24u - (unsigned)__clz((int)span);
                    pbase = lo;
                    p_src = part; p_dst = alt; p_a = lo; p_b = hi_r;
                    go = true;
                    ++n_refine;
                } else no_refine_at = lo;                 // all keys equal: the sorting paths order them by index
            }
            if (go) {
                // (1) re-uses the queue's histogram and cursors where they still stand: before any window (nothing placed), or for a group
                // behind the first window (the buckets up to its last one, s_cs[CS_B1], have been placed and consumed)
                const bool copy_only = p_dst == part && (n_fill == 0u || lo >= win_hi);
                partition_range(p_src, p_a, p_b, p_dst, copy_only, copy_only && n_fill != 0u ? (int)SGS_CS(CS_B1) : -1);
                if (p_dst == part) {                      // (1): the same group again, now from the copy
                    if (tid == 0) s_cs[CS_ORDERED] = 1u;
                    e_next = e0;
                    __syncthreads();                      // (the flag is read by every thread right after the group has been selected)
                }
                else e_next = 0;                                                             // (2): the slice's first bucket
                win_lo = lo; win_hi = lo;                 // nothing of the current partition is resident
                continue;
            }
        }
        const unsigned cnt = hi - lo;
        const unsigned* gv = nullptr;    // the group's slots in (depth, index) order
        if (cnt <= SGS_QCAP && hi > win_hi) {
            // long queue, group not resident: slide the window to start at this group and take as many whole
            // buckets as fit
            unsigned wfit = 0;                                        // same prefix count, for the window's capacity
#pragma unroll
            for (int r = 0; r < SGS_NB / 64; ++r) {
                const unsigned e = e1 + 1u + (unsigned)(r * 64 + lane);
                wfit += (unsigned)__popcll(__ballot(e < n_ne && s_ne_end[e] - lo <= (unsigned)SGS_QCAP));
            }
            const unsigned ew = e1 + wfit;
            win_lo = lo; win_hi = s_ne_end[ew];
            ++n_fill;
            if (SGS_CS(CS_ORDERED) == 0u) {               // (n_fill <= SGS_LAZY_WINDOWS: the copy is made before a later window)
                // one of the queue's first windows: one coalesced scan of the queue places the records of its buckets with the bucket cursors
                const unsigned b0 = s_ne_bkt[e0], b1 = s_ne_bkt[ew];
                if (tid == 0) s_cs[CS_B1] = b1;
                for (unsigned i0 = 0; i0 < n; i0 += 256u * SGS_PASS_R) {
                    unsigned long long x[SGS_PASS_R];
#pragma unroll
                    for (int r = 0; r < SGS_PASS_R; ++r) {
                        const unsigned i = i0 + (unsigned)tid + 256u * (unsigned)r;
                        x[r] = i < n ? rec[beg + i] : ~0ull;
                    }
#pragma unroll
                    for (int r = 0; r < SGS_PASS_R; ++r) {
                        const unsigned bk = SGS_BUCKET_OF((unsigned)(x[r] >> 32));
                        if (i0 + (unsigned)tid + 256u * (unsigned)r < n && bk >= b0 && bk <= b1)
                            s_q[atomicAdd(&s_bcnt[bk], 1u) - win_lo] = x[r];
                    }
                }
            } else {   // the window = a slice of the bucket-ordered copy (partition_range)
                const unsigned long long* src_w = SGS_CS(CS_FINE) != 0u ? alt : part;
                unsigned long long x[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned i = (unsigned)tid + 256u * (unsigned)r;
                    x[r] = i < win_hi - win_lo ? src_w[beg + win_lo + i] : ~0ull;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned i = (unsigned)tid + 256u * (unsigned)r;
                    if (i < win_hi - win_lo) s_q[i] = x[r];
                }
            }
            if (tid < 8) s_q[win_hi - win_lo + tid] = ~0ull;
            __syncthreads();
        }
        // Common case — the group is a single batch: every lane owns one record, issues the gather of ITS
        // splat first, ranks its record against the group while the loads are in flight, and stores the
        // splat at staging[rank].  No sorted index list, no exposed gather latency.
        //
        // DEEP TILES.  A tile that is still consuming batches after SGS_DEEP_AFTER of them keeps going for a few pixels that never
        // saturate — a gap between surfaces, a doorway — and from then on reads most of its queue: 14 batches at 1080p, 190 at
        // 320x240, where ONE such tile is the kernel's critical path (r04u: 4.0 M of a frame's 4.0 M cycles).  What it pays per batch is
        // fixed — a gather's round trip to HBM, the ranking of 256 records, their staging and quadrant tests, two barriers — for the
        // two or three dozen splats that can still reach a live pixel.  So such a tile first CULLS the whole resident window (up to
        // SGS_QCAP records: the rest of an in-LDS queue, or a long queue's window of whole buckets) against the live rectangles — four
        // records per lane, only the 32 bytes of the splat the test reads — and the survivors' keys are compacted into s_skey.  They are a
        // single batch (else: the ordinary path): ranked by their full keys (depth bits, index — so bucket boundaries and oversized buckets
        // no longer matter), staged and blended by the code below, unchanged.  A culled splat has no pixel it could change (the quadrant
        // test is conservative, and the live set only shrinks), so the frame is the same bit for bit.  Not with STATS: D_f is a queue
        // position, which survivors do not carry (the tests hold the frames of the two instantiations against each other).
        const unsigned long long* kk_d = s_q + (lo - win_lo);
        unsigned cnt_d = cnt, hi_d = hi, e_next_d = e_next;
        bool parted_d = parted;
        if (!STATS && deep_ok && !full_sort && !tile_done && parted && it >= SGS_DEEP_AFTER && n_live_px <= SGS_DEEP_LIVE && deep_hold == 0u && hi <= win_hi && win_hi - lo > cnt) {
            const unsigned wcnt = win_hi - lo;            // (<= SGS_QCAP)
#ifdef SGS_TILE_PROF
            ++pn_deep_try;
#endif
            if (tid == 0) s_nsurv = 0u;
            unsigned wfit = 0;                            // the window's last bucket: the buckets from e0 on that end inside it
#pragma unroll
            for (int r = 0; r < SGS_NB / 64; ++r) {
                const unsigned e = e0 + (unsigned)(r * 64 + lane);
                wfit += (unsigned)__popcll(__ballot(e < n_ne && s_ne_end[e] <= win_hi));
            }
            __syncthreads();
            for (unsigned r0 = 0; r0 < wcnt; r0 += 512u) {                       // two records per lane in flight
                unsigned long long kq[2]; float4 qA[2]; float qc[2], qm[2]; float2 qh[2]; bool hv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned i = r0 + (unsigned)tid + 256u * (unsigned)u;
                    hv[u] = i < wcnt;
                    kq[u] = hv[u] ? s_q[lo - win_lo + i] : ~0ull;
                    const float4* const sp = reinterpret_cast<const float4*>(splats + (hv[u] ? (unsigned)kq[u] : 0u));
                    qA[u] = sp[0]; qc[u] = *reinterpret_cast<const float*>(sp + 1);
                    qh[u] = reinterpret_cast<const float2*>(sp + 2)[1]; qm[u] = *reinterpret_cast<const float*>(sp + 3);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    unsigned qb4 = 0u;
                    if (hv[u] && qm[u] > 0.0f) {
                        const float rx = qA[u].x - tile_fx, ry = qA[u].y - tile_fy;
                        qb4 = sgs_quadrant_extent(rx, ry, qh[u].x, qh[u].y, s_lrect);
                        if (qb4 && !loose_cull) qb4 &= sgs_quadrant_hits_roots(rx, ry, qA[u].z, qA[u].w, qc[u], qm[u], s_lrect);
                    }
                    const unsigned long long sm = __ballot(qb4 != 0u);
                    if (sm != 0ull) {                     // one LDS atomic per wave and slot
                        unsigned sb = 0u;
                        if (lane == __ffsll((long long)sm) - 1) sb = atomicAdd(&s_nsurv, (unsigned)__popcll(sm));
                        sb = (unsigned)__builtin_amdgcn_readfirstlane((int)__shfl((int)sb, __ffsll((long long)sm) - 1));
                        const unsigned pos = sb + lanes_below(sm);
                        if (qb4 != 0u && pos < (unsigned)SGS_BATCH) s_skey[pos] = kq[u];
                    }
                }
            }
            __syncthreads();
            const unsigned ns = s_nsurv;                  // (uniform)
            SGS_PROF_MARK(pt_bar1);                       // (profiling build: the cycles of the cull)
#ifdef SGS_TILE_PROF
            if (ns <= (unsigned)SGS_BATCH) { ++pn_deep_ok; pn_deep_surv += ns; }
#endif
            if (ns <= (unsigned)SGS_BATCH) {
                if (tid == 0) atomicAdd(&st->n_deep, 1u);
                if (tid < 8) s_skey[ns + (unsigned)tid] = ~0ull;          // the sentinels of the 8-wide ranking walk
                kk_d = s_skey; cnt_d = ns; parted_d = false; hi_d = win_hi; e_next_d = e0 + wfit;
                __syncthreads();
                if (ns == 0u) { lo = hi_d; e_next = e_next_d; continue; }          // nothing in this window can reach a live pixel
            } else deep_hold = 2u;                        // too many pixels still live: the ordinary path for this group and the next
        } else if (deep_hold != 0u) --deep_hold;
        const bool direct = cnt_d <= SGS_BATCH && !tile_done && !full_sort;
        if (direct) {
            const unsigned long long* kk = kk_d;
            const unsigned cnt = cnt_d, hi = hi_d;
            const bool parted = parted_d;
            e_next = e_next_d;
            const unsigned par = it & 1u;
            const bool have = (unsigned)tid < cnt;
            const unsigned long long mine = have ? kk[tid] : ~0ull;
            // UNCONDITIONAL loads (lanes without a record read slot 0 and never use it): inside `if (have)` the
            // compiler has to merge the loaded registers with their defaults at the end of the branch, i.e. wait for
            // the gather right here — instead of behind the ranking below, which is what hides its latency
            const float4* const sp = reinterpret_cast<const float4*>(splats + (have ? (unsigned)mine : 0u));
            const float4 nA = sp[0], nB = sp[1], nC = sp[2];
            const float nD = *reinterpret_cast<const float*>(sp + 3);           // qmax
            // Rank of my record inside the group.  The resident queue is bucket-contiguous (MSD partition) and the
            // buckets are disjoint depth ranges, so rank = (records of shallower buckets) + (rank inside MY bucket):
            // a lane walks only its own bucket's slice — typically a few dozen records instead of the group's ~256.
            // Decided per wave (no barrier inside): a wave with one long bucket falls back to the broadcast walk.
            unsigned rank = 0;
            unsigned bbeg = win_lo, blen = 0;
            if (parted && have) {
                const unsigned bk = SGS_BUCKET_OF((unsigned)(mine >> 32));
                bbeg = bk ? s_bcnt[bk - 1] : pbase;         // cursors after the scatter = bucket ends = next bucket's start
                blen = s_bcnt[bk] - bbeg;
            }
            if (parted && __ballot(blen * 2u > cnt || blen > SGS_RANK_BUCKET_MAX) == 0ull) {
                const unsigned long long* bq = s_q + (bbeg - win_lo);
                unsigned r = 0;
                for (unsigned t = 0; __ballot(t < blen) != 0ull; t += 4) {       // four independent reads per trip
                    const unsigned long long x0 = bq[t], x1 = bq[t + 1], x2 = bq[t + 2], x3 = bq[t + 3];   // (reads past a
                    r += (t < blen && x0 < mine) ? 1u : 0u;                                               //  short bucket
                    r += (t + 1u < blen && x1 < mine) ? 1u : 0u;                                          //  are masked out)
                    r += (t + 2u < blen && x2 < mine) ? 1u : 0u;
                    r += (t + 3u < blen && x3 < mine) ? 1u : 0u;
                }
                rank = (bbeg - lo) + r;
            } else if ((unsigned)wave * 64u < cnt) {
                const unsigned cnt8 = (cnt + 7u) & ~7u;
                for (unsigned j = 0; j < cnt8; j += 8) {
                    unsigned long long x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = kk[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) rank += x[u] < mine ? 1u : 0u;
                }
            }
            SGS_PROF_MARK(pt_rank);
            // (no barrier here: ranking reads s_q, staging writes the arena, and this parity's quadrant masks were
            //  cleared when the batch before last was consumed)
            if (tid == 0) SGS_STAGE_DUMMY()
            if (have) {
                const float qmax = nD, hx = nC.z, hy = nC.w;        // log2(o / alpha_min); half extents of {alpha >= alpha_min}
                SGS_PROF_STAGED_ALL()
                const float rx = nA.x - tile_fx, ry = nA.y - tile_fy;      // centre relative to the tile
                SGS_STAGE(rank, nA, nB, nC, rx, ry)
                if (qmax > 0.0f) {
                    unsigned qb4 = sgs_quadrant_extent(rx, ry, hx, hy, s_lrect);       // the live pixels of each quadrant
                    if (qb4 && !loose_cull) qb4 &= sgs_quadrant_hits_roots(rx, ry, nA.z, nA.w, nB.x, qmax, s_lrect);
                    SGS_PROF_STAGED(qb4)
                    unsigned* bw = reinterpret_cast<unsigned*>(&s_ball[par][0][0]);      // [q][rank/64] as 2 x 32-bit
                    const unsigned word = rank >> 5, bit = 1u << (rank & 31u);
                    if (qb4 & 1u) atomicOr(&bw[0 * 8 + word], bit);
                    if (qb4 & 2u) atomicOr(&bw[1 * 8 + word], bit);
                    if (qb4 & 4u) atomicOr(&bw[2 * 8 + word], bit);
                    if (qb4 & 8u) atomicOr(&bw[3 * 8 + word], bit);
                }
            }
            SGS_PROF_MARK(pt_stage);
            __syncthreads();             // batch staged in depth order
            SGS_PROF_MARK(pt_sort);
#ifdef SGS_TILE_PROF
            ++pn_groups; ++pn_batches;
#endif
            if (tid == 0) { s_any[par ^ 1u] = 0; s_hyper[par ^ 1u] = 0; }
            const unsigned base = lo, m = cnt;
            const bool hyper = s_hyper[par] != 0u;           // (uniform)
            SGS_BLEND_WAVE_LIST()
            const unsigned long long live_m = __ballot(Tm > 0.0f);
            const bool still_live = live_m != 0ull;
            if (lane == 0) {
                if (still_live) atomicAdd(&s_any[par], (unsigned)__popcll(live_m));
                s_lrect[wave] = sgs_live_rect(live_m, (float)((wave & 1) * 8), (float)((wave >> 1) * 8));   // (read after the barrier)
            }
            __syncthreads();
            n_live_px = s_any[par]; tile_done = n_live_px == 0u;
            if (tid < 32) reinterpret_cast<unsigned*>(&s_ball[par][0][0])[tid] = 0u;    // consumed: ready for the batch after next
            ++it;
            // a tile that keeps consuming batches is on the kernel's critical path: let its waves win the issue
            // arbitration against the short-lived tiles they share the SIMDs with
            if (it == 1) __builtin_amdgcn_s_setprio(1); else if (it == 2) __builtin_amdgcn_s_setprio(2); else if (it == 4) __builtin_amdgcn_s_setprio(3);
            SGS_PROF_MARK(pt_blend);
            lo = hi;
            continue;
        }
        if (cnt <= SGS_QCAP) {
            const unsigned long long* kk = s_q + (lo - win_lo);   // resident: sort the slice in place; the records
                                                                  // behind it are deeper (or ~0), the sentinels the 8-wide walk needs
            if (cnt <= 256) rank_sort<1>(kk, s_sorted, cnt);
            else if (cnt <= 512) rank_sort<2>(kk, s_sorted, cnt);
            else rank_sort<4>(kk, s_sorted, cnt);
            __syncthreads();
            gv = s_sorted;
            if (full_sort) {
                for (unsigned i = tid; i < cnt; i += 256) sorted_out[beg + lo + i] = s_sorted[i];
                __syncthreads();         // s_sorted is rewritten by the next group when the blend is skipped
            }
        } else {
            // One oversized bucket of a long queue (keys all equal, or more than the LDS holds): radix sort through HBM.
            const unsigned g0 = s_ne_bkt[e0], g1 = s_ne_bkt[e1];
            // buckets g0..g1 hold the keys [klo + (g0 << ksh), klo + ((g1 + 1) << ksh)); the first and the last bucket are open-ended
            unsigned sub = klo + (g0 << ksh);
            unsigned nbits = ksh + (32u - (unsigned)__clz((int)(g1 - g0 + 1u)));
            if (g0 == 0u || g1 == SGS_NB - 1 || nbits >= 32u) { sub = 0u; nbits = 32u; }
            // The group's records are the slice [lo, hi) of the current ordered copy (`part`, or `alt` inside a refined bucket).  They are
            // split into a key half and a slot half in the OTHER of the two buffers (that slice of it holds a stale order of the same
            // records), and the queue itself — every record of it lives in the ordered copy by now — lends the ping-pong space.
            const bool fine = SGS_CS(CS_FINE) != 0u;
            const unsigned long long* src_g = fine ? alt : part;
            unsigned* b_k = reinterpret_cast<unsigned*>((fine ? part : alt) + beg + lo); unsigned* b_v = b_k + cnt;     // keys[cnt] | slots[cnt]
            unsigned* a_k = reinterpret_cast<unsigned*>(rec_w + beg + lo); unsigned* a_v = a_k + cnt;                   // ping-pong space
            for (unsigned i = (unsigned)tid; i < cnt; i += 256u) {
                const unsigned long long x = src_g[beg + lo + i];
                b_k[i] = (unsigned)(x >> 32); b_v[i] = (unsigned)x;
            }
            __syncthreads();
            sort_segment(b_k, b_v, a_k, a_v, cnt, sub, nbits, sh, splats, P.n, st);
            gv = b_v;
            if (tid == 0) atomicAdd(&st->class_count[3], 1u);
            if (full_sort) {
                __syncthreads();
                for (unsigned i = tid; i < cnt; i += 256) sorted_out[beg + lo + i] = gv[i];
            }
            __syncthreads();
        }
        SGS_PROF_MARK(pt_sort);
#ifdef SGS_TILE_PROF
        ++pn_groups;
#endif

        // ---- 3. blend the group in batches of 256 ------------------------------------------------
        if (!tile_done) {
            // (a group of more than one batch is one oversized depth bucket: rare — no splat prefetch is carried across the
            //  blend here, its registers are worth more to the common path)
            for (unsigned gb = 0; gb < cnt && !tile_done; gb += SGS_BATCH, ++it) {
                const unsigned par = it & 1u;
                const unsigned m = min((unsigned)SGS_BATCH, cnt - gb);
                const unsigned base = lo + gb;               // queue position of this batch
                const bool have = (unsigned)tid < m;
                unsigned qbits = 0;
                if (tid == 0) SGS_STAGE_DUMMY()       // (the arena is shared with the sort scratch: rewritten per batch)
                if (have) {
                    const float4* sp = reinterpret_cast<const float4*>(splats + gv[gb + (unsigned)tid]);
                    const float4 nA = sp[0], nB = sp[1], nC = sp[2];
                    const float qmax = *reinterpret_cast<const float*>(sp + 3), hx = nC.z, hy = nC.w;
                    const float rx = nA.x - tile_fx, ry = nA.y - tile_fy;          // centre relative to the tile
                    SGS_STAGE((unsigned)tid, nA, nB, nC, rx, ry)
                    if (qmax > 0.0f) {
                        qbits = sgs_quadrant_extent(rx, ry, hx, hy, s_lrect);
                        if (qbits && !loose_cull) qbits &= sgs_quadrant_hits_roots(rx, ry, nA.z, nA.w, nB.x, qmax, s_lrect);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned long long bal = __ballot((qbits >> q) & 1u);
                    if (lane == 0) s_ball[par][q][wave] = bal;
                }
                __syncthreads();                 // batch staged
                if (tid == 0) { s_any[par ^ 1u] = 0; s_hyper[par ^ 1u] = 0; }   // the other parity's flags: all their readers are past
                const bool hyper = s_hyper[par] != 0u;       // (uniform)
                SGS_BLEND_WAVE_SCAN()
                const unsigned long long live_m = __ballot(Tm > 0.0f);
                const bool still_live = live_m != 0ull;
                if (lane == 0) {
                    if (still_live) atomicAdd(&s_any[par], (unsigned)__popcll(live_m));
                    s_lrect[wave] = sgs_live_rect(live_m, (float)((wave & 1) * 8), (float)((wave >> 1) * 8));
                }
                __syncthreads();                 // batch consumed by every wave, liveness posted
                n_live_px = s_any[par]; tile_done = n_live_px == 0u;    // uniform
                if (tid < 32) reinterpret_cast<unsigned*>(&s_ball[par][0][0])[tid] = 0u;   // (the single-batch path ORs into it)
                if (it == 0) __builtin_amdgcn_s_setprio(1); else if (it == 1) __builtin_amdgcn_s_setprio(2); else if (it == 3) __builtin_amdgcn_s_setprio(3);
#ifdef SGS_TILE_PROF
                ++pn_batches;
#endif
            }
        }
        SGS_PROF_MARK(pt_blend);
        lo = hi;
    }
#ifdef SGS_TILE_PROF
    if (lane == 0) {
        for (int k_ = 0; k_ < 5; ++k_) atomicAdd(&s_pw[k_], (unsigned long long)pw_hist[k_]);
        atomicAdd(&s_pw[7], pw_list_cyc);
    }
    if (lane == 0) { atomicAdd(&s_pe[0], pe_eval); atomicAdd(&s_pe[1], pe_empty); atomicAdd(&s_pe[2], pe_valid); atomicAdd(&s_pe[3], pe_useful); atomicAdd(&s_pe[6], pe_dead); atomicAdd(&s_pe[7], pe_few); }
    __syncthreads();
    if (tid == 0 && prof) {
        unsigned long long* o = prof + (size_t)tile * SGS_PROF_WORDS;
        o[0] = n; o[1] = pt_part; o[2] = pt_sort; o[3] = pt_blend; o[4] = pn_groups; o[5] = pn_batches;
        o[6] = clock64() - pt0; o[7] = pt0;
        o[8] = s_pe[0]; o[9] = s_pe[1]; o[10] = s_pe[2]; o[11] = s_pe[3]; o[12] = s_pe[4]; o[13] = s_pe[5]; o[14] = prt0; o[15] = wall_clock64(); o[16] = pt_rank; o[17] = pt_bar1; o[18] = pt_stage; o[19] = pt_job;
        for (int k_ = 0; k_ < 8; ++k_) o[24 + k_] = s_pw[k_];
        o[20] = pt_rec; o[21] = n_refine | (pn_deep_try << 16) | (pn_deep_ok << 32); o[22] = n_fill | (pn_deep_surv << 16); o[23] = (unsigned long long)s_pe[6] | ((unsigned long long)s_pe[7] << 32);
    }
#endif
    {   // (the pixel's coordinates again, from a fresh copy of the thread index: see the group loop)
        int tid_out = tid_entry; SGS_PIN_VGPR(tid_out);
        const unsigned lane_o = (unsigned)tid_out & 63u, wave_o = (unsigned)tid_out >> 6;
        const unsigned cpx = tile_x * 16u + (wave_o & 1u) * 8u + (lane_o & 7u), in_y = (wave_o >> 1) * 8u + (lane_o >> 3);
        const unsigned cpy = frame_y * 16u + in_y;
        const bool inside = ((cpx | cpy) & zmask) == 0u && (cpx >> zf) < (unsigned)P.width && (cpy >> zf) < (unsigned)P.height;
        const unsigned px = cpx >> zf, out_py = (tile_y * 16u + in_y) >> zf;          // cell pixels -> the frame's
    if (inside) {
        float* o = out_rgb + ((size_t)out_py * P.width + px) * 3;
        if (TF) {
            // a pixel that never stopped still holds alpha_max T; one that stopped ended with 1 - (the sum of its weights)
            const float Tf = Tm > 0.0f ? Tm / amax : fmaxf(1.0f - Wsum, 0.0f);
            o[0] = C0 + Tf * P.bg[0]; o[1] = C1 + Tf * P.bg[1]; o[2] = C2 + Tf * P.bg[2];
            if (AUX) {                    // expected depth sum(T alpha z) and coverage 1 - T_final
                float* a = out_aux + ((size_t)out_py * P.width + px) * 2;
                a[0] = Dz; a[1] = 1.0f - Tf;
            }
        } else { o[0] = C0; o[1] = C1; o[2] = C2; }  // black background
    }
    }
    if (STATS) {                         // SGS_FLAG_STATS: D_f = furthest queue position any pixel examined
        const unsigned wu = wave_max(inside ? used : 0u);
        __syncthreads();
        if (lane == 0) s_used[wave] = wu;
        __syncthreads();
        if (tid == 0) {
            unsigned u = s_used[0];
            for (int w = 1; w < 4; ++w) u = s_used[w] > u ? s_used[w] : u;
            atomicAdd(&st->d_fetched, (unsigned long long)u);
        }
    }
}

template <bool AUX, bool STATS, bool TF>
__global__ __launch_bounds__(256, AUX ? 4 : SGS_RENDER_WGS) void k_tile_render(const FrameGroup G) {
    render_tile<AUX, STATS, TF>(G.s[blockIdx.y], blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// fp32 RGB -> uint8 RGBA, alpha 255; values clamped to [0,1], round to nearest.
__global__ __launch_bounds__(256) void k_pack_rgba8(const float* __restrict__ rgb,
                                                    unsigned* __restrict__ rgba, long long n_px) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_px) return;
    const float r = fminf(fmaxf(rgb[3 * i], 0.f), 1.f), g = fminf(fmaxf(rgb[3 * i + 1], 0.f), 1.f),
                b = fminf(fmaxf(rgb[3 * i + 2], 0.f), 1.f);
    rgba[i] = (unsigned)(r * 255.0f + 0.5f) | ((unsigned)(g * 255.0f + 0.5f) << 8) |
              ((unsigned)(b * 255.0f + 0.5f) << 16) | 0xff000000u;
}

}  // namespace sgs
