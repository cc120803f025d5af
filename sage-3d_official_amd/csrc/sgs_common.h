// sgs_common.h — types shared by the host side (sgs_api.hip) and the gfx950 kernels (sgs_kernels.h).
// Data layout in HBM is described in DESIGN.md §3.
#pragma once
#include <stdint.h>

#define SGS_WAVE 64                 // CDNA wavefront width; every wave idiom below assumes it
#define SGS_TILE_PX 16
#define SGS_GEOM_ROWS 3             // float4 rows per Gaussian in the scene geometry block
#define SGS_MAX_SH_ROWS 12          // ceil(48 floats / 4) at SH degree 3

// Compaction / binning (S4)
#ifndef SGS_RANGE
#define SGS_RANGE 256               // Gaussians per binning range: a workgroup owns ranges b, b+B, b+2B, ...  Consecutive
                                    // Gaussians share a surface (hence rect size), so coarse ranges of 1024 left the slowest
                                    // workgroup at 2-3x the mean; 256 measured best (64: more sweeps than it saves)
#endif
#define SGS_RANGE_CHUNKS (SGS_RANGE / SGS_WAVE)
// the launch-sized part of a workgroup's LDS (a CPU test harness may supply its own definition)
#ifndef SGS_DYNAMIC_LDS
#define SGS_DYNAMIC_LDS(T, name) extern __shared__ T name[]
#endif
// a wave-uniform value the hot loop multiplies with: pinned to a vector register (an SGPR or literal operand makes a VALU
// instruction cost 1.75 issue slots on gfx950: 4.2 against 2.4 cycles, scripts/ubench2.hip).  Value-preserving; a CPU test harness defines it away.
#ifndef SGS_PIN_VGPR
#define SGS_PIN_VGPR(x) asm volatile("" : "+v"(x))
#endif
#define SGS_WT 8192                 // super-tiles per binning window: per-workgroup counters live in LDS (32 KB)
#define SGS_BIN_THREADS 512
#define SGS_BIN_BLOCKS 512          // binning workgroups (2 per CU); each owns ranges b, b+B, b+2B, ...
#define SGS_MAX_WINDOWS 16          // ceil(tiles / SGS_WT) the queues support: 131072 tiles (8192x4096 px)
#define SGS_XCDS 8                  // sub-queues per tile: one per XCD the binning workgroups run on
#define SGS_MAX_ROWS 4096           // tile rows of a frame (65536 px): length of the per-row record counters
#define SGS_ST 4                    // a super-tile is SGS_ST x SGS_ST tiles (64 x 64 px): level 1 of the binning
#define SGS_ST_SHIFT 2
#define SGS_SEG 1024                // records of a super-tile queue per level-2 job (four per thread)
#define SGS_EXP_THREADS 256
#define SGS_EXP_GRID 2048           // level-2 workgroups per launch (each loops over jobs b, b + grid, ...)
#define SGS_BIG_RECT 128            // splats touching more SUPER-TILES than this are expanded by a whole workgroup (a wave walks its
                                    // chunk as long as its largest rect)
#define SGS_BIG_CAP 65536           // entries of the per-frame big-splat list (overflow falls back to the wave path)
#define SGS_MAX_LIVE 512             // live-chunk list per pass (one sweep: 512 chunks = 32 K Gaussians per workgroup and pass); LDS is what
                                     // decides how many composite workgroups fit beside a binning workgroup on a CU

// Radix sort (S5)
#define SGS_RADIX_BITS 8
#define SGS_RADIX (1 << SGS_RADIX_BITS)
#define SGS_SORT_CLASSES 4
#define SGS_PROF_WORDS 32            // profiling build: words per tile in the tile_prof buffer
#define SGS_TIE_RUN_MAX 32          // equal-depth runs longer than this take the (index,depth) resort

#define SGS_PFLAG_SH_PACKED 0x80000000u   // FrameParams.flags, set by the library (never by a caller's sgs_config): the scene's SH rows are packed bytes,
#define SGS_PFLAG_SH_MODE_SHIFT 29        // ... decoded with mode (flags >> 29) & 3 (sage_gs.h SGS_SH_DECODE_*)
#define SGS_PFLAG_FINE_SHIFT 27           // ... bits 27-28, set by the library: z, the frame's FINE-TILE shift.  Its tiles are (16 >> z)^2 pixels
                                          //     (FrameParams.gx / gy / row_begin / row_end count THOSE tiles), see "Fine tiles" below
#define SGS_PFLAG_INTERNAL 0xF8000000u
// Fine tiles.  The composite puts a workgroup on a tile and a lane on a pixel; a frame of 320x240 pixels is 300 tiles of 16x16 — a chip of
// 256 CUs is a quarter full, and the frame takes as long as its slowest tile, whose four waves walk lists of thousands of splats one after
// the other (profiles/r06c: 3.9 M of a 320x240 frame's 3.9 M cycles are ONE tile).  Such frames are rendered through tiles of
// (16 >> z)^2 pixels instead, WITHOUT a second set of kernels: every kernel keeps working on a grid of 16x16 "cells", and k_preprocess hands
// it the splats in coordinates scaled by 2^z — positions and extents times 2^z, the roots a, a k, c of the conic divided by 2^z (powers of
// two: exact) — so that pixel (i, j) of the frame is cell-pixel (i << z, j << z); the composite's lanes on the other cell-pixels are
// "outside the image" from the start (Tm = 0), the same mechanism that handles a tile at the frame's edge.  A 16x16 cell is then
// (16 >> z)^2 real pixels, a wave's quadrant (8 >> z)^2, the slowest tile's queue is split over 4^z workgroups and its waves' lists over
// 4^z times as many waves.  What S3 defines — a splat reaches the pixels of the 16x16-pixel tiles its rect covers — is kept: the rect is
// computed on the 16-pixel grid and mapped to the cells it covers (k_preprocess).  z is chosen per call by the library (sgs_api.hip).
// Per-frame parameters, passed BY VALUE to every kernel (kernarg segment, scalar-loaded).
struct FrameParams {
    float view[12];                 // rows 0..2 of the model->camera matrix (row-major 3x4)
    double campos[3];               // camera centre in model space
    float fx, fy, cx, cy;
    float near_z, far_z, dilation, clamp;
    float alpha_min, alpha_max, t_min;
    float bg[3];
    int32_t width, height, gx, gy;
    int32_t row_begin, row_end;     // tile rows rendered by this call
    int32_t sh_degree;              // degree evaluated
    int32_t sh_rows;                // 16-byte rows per Gaussian stored in the scene (by scene degree): fp32 coefficients, or — SGS_PFLAG_SH_PACKED,
                                    // a scene uploaded from the compressed payload — fp32 DC + one byte per higher coefficient
    int64_t n;                      // Gaussians
    int64_t n_chunks;               // ceil(n / 64)
    int32_t n_ranges;               // ceil(n / SGS_RANGE)
    int32_t n_windows;              // binning windows (one: the super-tiles of a band fit one window)
    int64_t rec_capacity;           // records the tile queues can hold (the super-tile queues: half as many 16-byte records)
    int32_t job_capacity;           // level-2 jobs the job table can hold
    uint32_t flags;
    int32_t row_stride, row_phase;  // interleaved tile rows: local row k of this call is frame row k * row_stride + row_phase
    int32_t cull_y0, cull_y1;       // pixel rows outside [cull_y0, cull_y1) cannot matter to this call (conservative)
    // S2's clamp of t.xy / t.z, constants of the frame (a division each: once on the host, not once per Gaussian):
    double limx, limy;              // clamp * (0.5 * width / fx), clamp * (0.5 * height / fy) — fp64 from the fp32 parameters, as the oracle forms them
    // k_chunk_cull's four planes (left, right, top, bottom), constants of the frame (filled on the host, fill_params):
    // a chunk is outside when  f u + off tz + nrm R + A s_max < 0  for one of them (chunk_outside, sgs_kernels.h)
    double cull_A;                  // 1.001 * 3 sqrt(2 (2 + lx^2 + ly^2)) max(fx, fy): radius bound = A s_max / tz + c0
    double cull_off[4];             // the plane's tz coefficient (image edge, principal point, c0)
    double cull_nrm[4];             // sqrt(f^2 + off^2)
};
static_assert(sizeof(FrameParams) == 304, "FrameGroup (8 slots, by value) must stay inside the 4-KiB kernarg segment");

// Device-resident per-frame status; zeroed by a memset node at frame start, copied to pinned host
// memory at frame end.
struct alignas(128) FrameStatus {
    // line 0: written once by one kernel, read at the start of every workgroup of the kernels after it
    uint32_t n_live;                // chunks that passed the per-chunk bounds (k_chunk_cull): length of the frame's live list
    uint32_t d_total;               // D
    uint32_t overflow;              // D > rec_capacity: emit/sort/composite did nothing
    uint32_t max_tile_len;
    uint32_t ds_total;              // records in the super-tile queues (level 1 of the binning)
    uint32_t n_jobs;                // level-2 jobs (k_stile_scan)
    uint32_t n_work;                // (the FIRST frame of a group only) entries of the group's work list: (chunk, frame) pairs (k_chunk_cull_group)
    uint32_t pad0_[9];
    // line 1: counters that workgroups ADD to while others of the same kernel read line 0 (a device-scope atomic occupies
    // its line in the fabric for ~12 ns: on one line the readers queued behind the adders)
    uint32_t n_visible;             // N_v
    uint32_t n_big;                 // splats in the big-rect list (may exceed SGS_BIG_CAP; consumers clamp)
    unsigned long long d_fetched;   // D_f (SGS_FLAG_STATS)
    uint32_t class_count[SGS_SORT_CLASSES];   // [3]: oversized depth buckets sorted through HBM; others unused
    uint32_t n_resort_tiles;        // tiles that needed the (index, depth) resort for long tie runs
    uint32_t n_deep;                // windows culled against the tile's live pixels before ranking (k_tile_render, deep tiles)
    uint32_t pad1_[6];
};
static_assert(sizeof(FrameStatus) == 128, "two cache lines");

struct Splat;

// Frame groups.  A launch of any per-frame kernel covers up to SGS_MAX_GROUP frames of one scene: blockIdx.y selects the
// frame, and everything that differs between the frames of a group — parameters, intermediates, output — is one FrameSlot
// of the FrameGroup passed BY VALUE (kernarg segment, scalar-loaded; no device copy to keep alive).  Why: a light frame
// (a rank's band of tile rows) is five launches of ~25 us each, and an MI355X retires small kernels from several streams
// barely faster than from one (round 2's launch-floor probe: 1.6-2x at best) — so a sweep's frames ride the same five
// launches instead of five launches each.
#define SGS_MAX_GROUP 8
struct FrameSlot {
    FrameParams P;
    // the frame's intermediates (one "lane" of the context)
    Splat* splats; unsigned long long* vismask; unsigned long long* bigmask; unsigned* big_list; uint4* binrec;
    unsigned* live_list;            // chunks (layout order) that passed k_chunk_cull; FrameStatus.n_live of them
    unsigned* tile_count; unsigned* tile_offset; uint4* tile_order; uint2* blk_list; unsigned* blk_len;
    unsigned* stile_count; unsigned* stile_offset; uint4* jobs; unsigned* job_base;      // two-level binning: super-tile sub-counters / offsets,
                                                                                        // level-2 jobs, their per-tile bases (+ XCD)
    unsigned long long* rec; unsigned long long* alt; unsigned long long* part; unsigned* sorted_out;
    unsigned long long* tile_prof; unsigned long long* bin_prof;
    // the frame's output and status word
    float* out_rgb; float* out_aux; FrameStatus* st;
};
struct FrameGroup {
    const float4* geom; const float4* shq; const float4* cbound;      // the scene (shared by the group's frames)
    unsigned long long* row_acc;                                       // per-row record counters of the context
    FrameSlot s[SGS_MAX_GROUP];
};

// One projected Gaussian ("splat"): 64 B, 64-B aligned — exactly one HBM access sector, written once per frame by
// k_preprocess (at the Gaussian's ORIGINAL index: with the scene in Z-order that is a scatter, and a 48-B record
// straddling sectors made every such write a read-modify-write) and gathered by k_tile_render.  Everything the composite
// needs per (splat, tile) that does not depend on the tile is computed here once, per splat:
//   word 0..3   x, y, a, a k          q2(d) = A dx^2 + B dx dy + C dy^2 = -power * log2(e)  (A = conic_a log2(e)/2, B = conic_b log2(e), ...)
//                                     as the completed square A (dx + k dy)^2 + C' dy^2 (k = B / 2A, C' = C - B^2 / 4A), stored by its
//                                     roots a = sqrt(A) (slot `A`), a k (slot `B`):  q2 = U^2 + V^2,  U = a (dx + k dy),  V = c dy
//   word 4..7   c, nlo, r, g          c = sqrt(|C'|) with the sign of C' (slot `C`; negative: an indefinite fp32 conic);  nlo = log2(alpha_max /
//                                     opacity): the composite evaluates q = q2 + nlo, alpha / alpha_max = min(1, 2^-q), and
//                                     alpha >= alpha_min <=> q <= log2(alpha_max / alpha_min)
//   word 8..11  b, depth bits, hx, hy hx, hy = half extents of the ellipse {alpha >= alpha_min}, padded outward
//   word 12..15 qmax, opacity, rect x0|y0<<16, rect x1|y1<<16     qmax = log2(opacity / alpha_min); rect = S3's reference rect (tests)
struct alignas(64) Splat {
    float x, y, A, B;
    float C, nlo, r, g;
    float b; uint32_t key; float hx, hy;
    float qmax, o; uint32_t rect01; uint32_t rect23;
};
