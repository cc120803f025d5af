// sgs_api.hip — host side of libsage_gs.so: the C ABI of include/sage_gs.h over the gfx950 kernels.
//
// A frame is five stream-ordered launches (k_preprocess, k_bin_count, k_tile_scan, k_bin_emit, k_tile_render) with
// no host synchronisation in between (grids that depend on device-side counts are fixed-size and grid-stride); the
// frame's FrameStatus is copied to pinned host memory at the end and inspected when the caller synchronises.
// Ordinary frames run on the caller's stream with lane 0's intermediates; pipelined frames (SGS_FLAG_PIPELINED,
// sgs_render_batch) rotate over a few lanes, each with its own stream and intermediates, so that independent
// frames overlap on the chip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sage_gs.h"
#include "sgs_kernels.h"

namespace {

constexpr int kStatusRing = 256;
thread_local std::string g_create_error;

struct Scratch {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

// One Gaussian of a scene's PROBE (fine_shift_of): mean, 3-D covariance (+ its trace), ln(opacity), sampling weight — 512 of them, drawn at
// upload (layout_scene) and kept on the host.  Half of the draws are uniform over the scene, half proportional to the Gaussian's squared size
// (trace of its covariance): the record count of a scene with trained-3DGS statistics is carried by a few per cent of large splats, which a
// uniform sample of 512 holds a handful of — the estimate of the growth under a split was off by 20 % on such scenes and decided wrongly for
// one pose in six (r06w); with the mixture (each draw weighted by the inverse of its probability) the estimate follows the frame's real ratio.
struct ProbeSample { float m[3]; float S[6]; float ln_o; float trace; float wgt; };

struct sgs_scene {
    std::vector<ProbeSample> probe;     // (empty: fewer Gaussians than a probe is worth)
    int64_t n = 0, n_chunks = 0;
    int sh_degree = 0, sh_rows = 0;     // sh_rows: 16-byte rows of SH per Gaussian (12 at degree 3; 4 when sh_packed)
    bool sh_packed = false;             // uploaded from the compressed payload: the 8-bit coefficients stay bytes in HBM (k_scene_layout<true>)
    int sh_decode = 0;                  // ... and are read as sage_gs.h SGS_SH_DECODE_* says
    float4* geom = nullptr;
    float4* shq = nullptr;
    float4* cbound = nullptr;           // per chunk: bounding sphere of the means + largest scale (k_chunk_bounds)
    unsigned* perm_host = nullptr;      // Z-order: layout position -> original index (nullptr = identity)
};

// The intermediates of ONE frame in flight.  Lane 0 serves ordinary frames on the caller's stream; pipelined
// frames (SGS_FLAG_PIPELINED, sgs_render_batch) rotate over all lanes, each on its own stream, so that a
// few frames overlap: binning is latency- and imbalance-bound, the composite issue-bound, and together they
// fill the chip better than back to back (measured +35 % frames/s with three lanes).
struct Lane {
    // per-Gaussian scratch
    int64_t splat_cap = 0;
    Splat* splats = nullptr;                 // one slot per Gaussian (slot == index)
    unsigned long long* vismask = nullptr;   // per 64-Gaussian chunk: which slots are live this frame
    unsigned long long* bigmask = nullptr;   //   ... and which of those went to the big-rect list
    unsigned* big_list = nullptr;
    uint4* binrec = nullptr;                 // per slot: depth bits, rect01, rect23 (dense copy for the binning kernels)
    unsigned* live_list = nullptr;           // chunks that passed the per-chunk bounds this frame (k_chunk_cull)
    // per-tile scratch
    int tile_cap = 0;
    unsigned *tile_count = nullptr, *tile_offset = nullptr;
    uint4* tile_order = nullptr;                     // render order: (tile, first record, queue length) per position
    unsigned long long* tile_prof = nullptr;         // profiling build: 8 words per tile
    unsigned long long* bin_prof = nullptr;          // profiling build: 8 words per binning workgroup
    // binning scratch: per-workgroup (tile, base) lists
    uint2* blk_list = nullptr;                       // per level-1 binning workgroup: (super-tile, base) of the super-tiles it touched
    unsigned* blk_len = nullptr;
    // two-level binning: super-tile sub-counters / offsets (SGS_WT super-tiles at most), the level-2 job table and the jobs' bases
    unsigned *stile_count = nullptr, *stile_offset = nullptr;
    uint4* jobs = nullptr;
    unsigned* job_base = nullptr;
    int64_t job_cap = 0;
    // per-record scratch
    int64_t rec_cap = 0;
    unsigned long long *rec = nullptr;                   // tile queues of (depth bits << 32 | slot) records
    unsigned long long *alt = nullptr, *part = nullptr;  // scratch of the HBM radix path (oversized depth buckets only)
    unsigned* sorted_out = nullptr;                      // SGS_FLAG_FULL_SORT (tests): fully ordered queues
    int64_t sorted_cap = 0;
    // pipelined frames
    hipStream_t stream = nullptr;            // internal, non-blocking
    hipEvent_t fork = nullptr, done = nullptr;
    bool busy = false;                       // has pipelined work that no synchronisation has collected yet
};

constexpr int kMaxLanes = 16;

struct sgs_ctx {
    int device = 0;
    std::string err;
    Lane lanes[kMaxLanes];
    int n_lanes = 3, next_lane = 0;          // sgs_tuning.lanes: lanes that SGS_FLAG_PIPELINED single frames rotate over
    int group = 8, group_lanes = 2;          // sgs_tuning.group x .group_lanes <= kMaxLanes: sgs_render_batch* issues `group` frames per
                                             // set of launches (blockIdx.y = frame), groups rotating over `group_lanes` streams
    int last_lane = 0;
    int exp_grid = SGS_EXP_GRID;             // level-2 binning workgroups per launch (settled by A/B: r03, r04)
    int bin_grid = SGS_BIN_BLOCKS;           // binning workgroups per launch (<= SGS_BIN_BLOCKS)
    int pre_grid = 8192;                     // k_preprocess workgroups per launch of a frame GROUP: its waves loop over the live list (r03y)
    int64_t fine_tile_pixels = 640 * 480;    // sgs_tuning.fine_tile_pixels: frames of at most this many pixels may be rendered through 8x8-pixel
                                             // tiles (fine_shift_of; sgs_common.h "Fine tiles"); 0 = never
    double fine_tile_growth = 2.2;           // sgs_tuning.fine_tile_growth: ... while a split multiplies the frame's records by no more than this
    bool morton = true;                      // Z-order the scene at upload (sgs_tuning.morton = 0 keeps the caller's order): a chunk of 64
                                             // Gaussians is then a compact patch, which is what makes the per-chunk bounds
                                             // (k_chunk_bounds / chunk_outside) worth testing — trained scenes come in no spatial order
    unsigned long long* row_acc = nullptr;   // records queued per frame tile row, summed over the frames since the last
                                             // sgs_row_records(reset) — what cost-balanced tile-row bands are cut from
    const sgs_scene* last_scene = nullptr;
    int64_t rec_cap_wanted = 16ll << 20;
    // status ring
    FrameStatus* d_status = nullptr;
    FrameStatus* h_status = nullptr;
    int next_slot = 0;
    // bookkeeping of the most recent frame
    int last_slot = -1;
    bool last_timed = false;
    int64_t last_n = 0, last_pixels = 0;
    int last_tiles = 0, last_sh_rows = 0, last_T = 0;
    int last_t_lo = 0, last_t_hi = 0;         // the band of tiles the last frame rendered (k_tile_scan fills only those)
    int last_retries = 0;
    hipStream_t last_stream = nullptr;
    // frames issued since the last synchronisation (ring slots pending_begin .. +pending_count)
    int pending_begin = 0, pending_count = 0;
    // per-slot event sets for SGS_FLAG_TIMING, created on first use
    hipEvent_t (*ev)[SGS_NUM_STAGES + 1] = nullptr;
    bool slot_timed[256] = {};
};

#define SGS_FAIL(ctx, code, ...)                                  \
    do {                                                          \
        char buf_[512];                                           \
        snprintf(buf_, sizeof buf_, __VA_ARGS__);                 \
        (ctx)->err = buf_;                                        \
        return (code);                                            \
    } while (0)

#define SGS_HIP(ctx, call)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            SGS_FAIL(ctx, e_ == hipErrorOutOfMemory ? SGS_ERR_OOM : SGS_ERR_HIP, "%s: %s", #call, \
                     hipGetErrorString(e_));                                                   \
    } while (0)

namespace {

template <class T>
int grow(sgs_ctx* ctx, T*& p, size_t count) {
    if (p) { (void)hipFree(p); p = nullptr; }
    if (count == 0) count = 1;
    SGS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
    return SGS_OK;
}

int ensure_splats(sgs_ctx* ctx, Lane& L, int64_t n) {
    if (n <= L.splat_cap && L.blk_len) return SGS_OK;
    const int64_t chunks = std::max<int64_t>(1, (n + 63) / 64);
    const int64_t cap = chunks * 64;
    int rc;
    if ((rc = grow(ctx, L.splats, (size_t)cap)) != SGS_OK) return rc;
    if ((rc = grow(ctx, L.vismask, (size_t)chunks)) != SGS_OK) return rc;
    if ((rc = grow(ctx, L.bigmask, (size_t)chunks)) != SGS_OK) return rc;
    if ((rc = grow(ctx, L.binrec, (size_t)cap)) != SGS_OK) return rc;
    if ((rc = grow(ctx, L.live_list, (size_t)chunks * (1 + SGS_MAX_GROUP) + 2)) != SGS_OK) return rc;     // the list | a group's work list (sgs_work_list)
    if (!L.big_list && (rc = grow(ctx, L.big_list, (size_t)SGS_BIG_CAP)) != SGS_OK) return rc;
    if ((rc = grow(ctx, L.blk_len, (size_t)SGS_BIN_BLOCKS * 2)) != SGS_OK) return rc;                      // list lengths | XCD ids
    if (!L.blk_list && (rc = grow(ctx, L.blk_list, (size_t)SGS_BIN_BLOCKS * SGS_WT)) != SGS_OK) return rc;
    if (!L.stile_count) {
        if ((rc = grow(ctx, L.stile_count, (size_t)SGS_WT * SGS_XCDS)) != SGS_OK) return rc;
        if ((rc = grow(ctx, L.stile_offset, (size_t)SGS_WT * SGS_XCDS + 1)) != SGS_OK) return rc;
        // k_bin_emit zeroes every count k_stile_scan has consumed, so one memset at allocation suffices (see ensure_tiles)
        SGS_HIP(ctx, hipMemset(L.stile_count, 0, (size_t)SGS_WT * SGS_XCDS * sizeof(unsigned)));
        SGS_HIP(ctx, hipStreamSynchronize(nullptr));
    }
    if ((rc = grow(ctx, L.bin_prof, (size_t)SGS_BIN_BLOCKS * 8)) != SGS_OK) return rc;
    L.splat_cap = cap;
    return SGS_OK;
}

int ensure_tiles(sgs_ctx* ctx, Lane& L, int tiles) {
    if (tiles <= L.tile_cap) return SGS_OK;
    int rc;
    if ((rc = grow(ctx, L.tile_count, (size_t)tiles + 1)) != SGS_OK) return rc;          // level 2: one counter per tile
    if ((rc = grow(ctx, L.tile_offset, (size_t)tiles + 1)) != SGS_OK) return rc;
    if ((rc = grow(ctx, L.tile_prof, (size_t)tiles * SGS_PROF_WORDS)) != SGS_OK) return rc;
    if ((rc = grow(ctx, L.tile_order, (size_t)tiles)) != SGS_OK) return rc;
    // k_expand<true> zeroes every count k_tile_scan has consumed, so one memset at allocation suffices
    SGS_HIP(ctx, hipMemset(L.tile_count, 0, ((size_t)tiles + 1) * sizeof(unsigned)));
    // (hipMemset of device memory runs on the NULL stream and may return before it has run; the frames use non-blocking
    //  streams that do not order with it — an unfinished clear would land in the middle of a frame's counting)
    SGS_HIP(ctx, hipStreamSynchronize(nullptr));
    L.tile_cap = tiles;
    return SGS_OK;
}

int ensure_records(sgs_ctx* ctx, Lane& L) {
    if (L.rec_cap >= ctx->rec_cap_wanted && L.rec) return SGS_OK;
    const int64_t cap = std::max<int64_t>(ctx->rec_cap_wanted, 1024);
    if (cap > 0xfffffff0ll) SGS_FAIL(ctx, SGS_ERR_INVALID, "record capacity %lld exceeds 2^32", (long long)cap);
    int rc;
    if ((rc = grow(ctx, L.rec, (size_t)cap)) != SGS_OK) return rc;
    if ((rc = grow(ctx, L.alt, (size_t)cap)) != SGS_OK) return rc;
    if ((rc = grow(ctx, L.part, (size_t)cap)) != SGS_OK) return rc;
    // level-2 jobs: the super-tile queues (cap / 2 records of 16 bytes, in `alt`) in segments of SGS_SEG, plus one ragged
    // segment per super-tile
    const int64_t jcap = cap / 2 / SGS_SEG + SGS_WT + 1;
    if ((rc = grow(ctx, L.jobs, (size_t)jcap)) != SGS_OK) return rc;
    if ((rc = grow(ctx, L.job_base, (size_t)jcap * (SGS_ST * SGS_ST))) != SGS_OK) return rc;
    L.job_cap = jcap;
    L.rec_cap = cap;
    return SGS_OK;
}

int ensure_sorted_out(sgs_ctx* ctx, Lane& L) {
    if (L.sorted_cap >= L.rec_cap && L.sorted_out) return SGS_OK;
    int rc;
    if ((rc = grow(ctx, L.sorted_out, (size_t)L.rec_cap)) != SGS_OK) return rc;
    L.sorted_cap = L.rec_cap;
    return SGS_OK;
}

// A lane's stream and events exist from its first pipelined frame on.
int ensure_lane_stream(sgs_ctx* ctx, Lane& L) {
    if (L.stream) return SGS_OK;
    SGS_HIP(ctx, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    SGS_HIP(ctx, hipEventCreateWithFlags(&L.fork, hipEventDisableTiming));
    SGS_HIP(ctx, hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    return SGS_OK;
}

// Host-wait for every pipelined frame in flight.
int drain_lanes(sgs_ctx* ctx) {
    for (int l = 0; l < kMaxLanes; ++l) {
        Lane& L = ctx->lanes[l];
        if (L.busy) { SGS_HIP(ctx, hipEventSynchronize(L.done)); L.busy = false; }
    }
    return SGS_OK;
}

// Fine tiles (sgs_common.h): the shift z of a call — its frame is rendered through tiles of (16 >> z)^2 pixels.  A function of the scene, the
// camera and the configuration ALONE, so that a frame is the same frame however it is issued (alone, pipelined, in a batch, as a band).
//   * Never under the test hooks whose point is the REFERENCE's integer structures — queues and offsets of 16x16-pixel tiles.
//   * Small frames only: z <= the number of times the frame's pixel count, quadrupled, stays within sgs_tuning.fine_tile_pixels (640x480
//     -> 1, 320x240 -> 2) — a large frame fills the chip with 16x16 tiles.
//   * And only while halving the tiles does not multiply the RECORDS by more than sgs_tuning.fine_tile_growth (2.2).  Every record is binned, partitioned, ranked and
//     quadrant-tested once per tile it lands in, so what a split costs is the growth of D — small where splats are smaller than the tiles
//     (x1.5-2.0 on the indoor scenes: the split wins 20-50 % of a frame), towards x4 where they are larger (x2.5-3.5 on scenes with
//     trained-3DGS statistics, whose pixels saturate inside the first batch anyway: there the split LOSES 30 %).  Measured per pose, both
//     scene kinds, three resolutions (profiles/r06w_fine_tiles_growth_rule.txt): the frame's time is shorter with the split below a growth of
//     2.1-2.4 and longer above; the rule lands within 1 % of choosing the better tiling per pose.  D at each tile size is ESTIMATED here, on
//     the host, from the scene's probe — a thousand Gaussians projected with S2's arithmetic (fp32) and binned over the extent of
//     {alpha >= alpha_min} as k_preprocess bins it: ~10 us per frame, within a few per cent of the frame's real D ratio.
// floor / ceil of a value well inside the int range, without the libm call a portable build makes of std::floor
inline int ifloor(float v) { const int i = (int)v; return i - (v < (float)i ? 1 : 0); }
inline int iceil(float v) { const int i = (int)v; return i + (v > (float)i ? 1 : 0); }
int fine_shift_of(const sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cam, const sgs_config& cfg) {
    if (cfg.flags & (SGS_FLAG_NO_FINE_TILES | SGS_FLAG_FULL_SORT | SGS_FLAG_LOOSE_CULL)) return 0;
    int zcap = 0;
    while (zcap < 2 && (((int64_t)cam->width * cam->height) << (2 * zcap)) <= ctx->fine_tile_pixels) ++zcap;
    if (zcap == 0 || !scene || scene->probe.empty() || ctx->fine_tile_growth >= 16.0) return zcap;
    const float* V = cam->view;
    const float fx = cam->fx, fy = cam->fy, W = (float)cam->width, H = (float)cam->height;
    const float limx = cfg.clamp * (0.5f * W / fx), limy = cfg.clamp * (0.5f * H / fy);
    const float ln_amin = std::log(cfg.alpha_min);
    const float jb = std::max(fx, fy), jk = 2.0f + limx * limx + limy * limy;
    double D[3] = {0.0, 0.0, 0.0};
    for (const ProbeSample& g : scene->probe) {
        const float tz = V[8] * g.m[0] + V[9] * g.m[1] + V[10] * g.m[2] + V[11];
        if (!(tz > cfg.near_z) || !(tz <= cfg.far_z)) continue;
        const float K = 2.0f * (g.ln_o - ln_amin);                    // alpha >= alpha_min  <=>  d^T Sigma'^-1 d <= K
        if (!(K > 0.0f)) continue;                                    // (never blended: nothing is binned)
        const float tx = V[0] * g.m[0] + V[1] * g.m[1] + V[2] * g.m[2] + V[3], ty = V[4] * g.m[0] + V[5] * g.m[1] + V[6] * g.m[2] + V[7];
        const float itz = 1.0f / tz, xz = tx * itz, yz = ty * itz;
        const float px = fx * xz + cam->cx - 0.5f, py = fy * yz + cam->cy - 0.5f;
        {   // off screen by more than any footprint it can have (k_preprocess's own pre-check: lambda_max <= |J|_F^2 trace(Sigma) + dilation,
            // radius <= 3 sqrt(2 lambda_max + 0.3163) + 1, squared and rounded up): most of a probe ends here
            const float jf = jb * itz, lmax = jf * jf * jk * g.trace + cfg.dilation;
            const float rb2 = 9.5f * (2.0f * lmax + 0.3163f) + 4.0f;
            const float ox = px < 0.0f ? -px : px > W ? px - W : 0.0f, oy = py < 0.0f ? -py : py > H ? py - H : 0.0f;
            if (ox * ox > rb2 || oy * oy > rb2) continue;
        }
        const float txc = std::min(limx, std::max(-limx, xz)) * tz, tyc = std::min(limy, std::max(-limy, yz)) * tz;
        const float j00 = fx * itz, j02 = -fx * txc * itz * itz, j11 = fy * itz, j12 = -fy * tyc * itz * itz;
        const float T0[3] = {j00 * V[0] + j02 * V[8], j00 * V[1] + j02 * V[9], j00 * V[2] + j02 * V[10]};
        const float T1[3] = {j11 * V[4] + j12 * V[8], j11 * V[5] + j12 * V[9], j11 * V[6] + j12 * V[10]};
        const float* S = g.S;                                         // (00, 01, 02, 11, 12, 22)
        const float u0 = S[0] * T0[0] + S[1] * T0[1] + S[2] * T0[2], u1 = S[1] * T0[0] + S[3] * T0[1] + S[4] * T0[2], u2 = S[2] * T0[0] + S[4] * T0[1] + S[5] * T0[2];
        const float w0 = S[0] * T1[0] + S[1] * T1[1] + S[2] * T1[2], w1 = S[1] * T1[0] + S[3] * T1[1] + S[4] * T1[2], w2 = S[2] * T1[0] + S[4] * T1[1] + S[5] * T1[2];
        const float a = T0[0] * u0 + T0[1] * u1 + T0[2] * u2 + cfg.dilation, b = T1[0] * u0 + T1[1] * u1 + T1[2] * u2,
                    c = T1[0] * w0 + T1[1] * w1 + T1[2] * w2 + cfg.dilation;
        const float det = a * c - b * b;
        if (!(det > 0.0f)) continue;
        const float mid = 0.5f * (a + c), r3 = 3.0f * __builtin_sqrtf(mid + __builtin_sqrtf(std::max(0.1f, mid * mid - det)));     // S3
        if (!(r3 < 1.0e6f) || !(std::fabs(px) < 1.0e6f) || !(std::fabs(py) < 1.0e6f)) continue;       // (NaN / wild values: not a probe worth counting)
        const float radius = (float)iceil(r3);
        const float hx = std::min(__builtin_sqrtf(K * a), radius), hy = std::min(__builtin_sqrtf(K * c), radius);
        if (!(hx >= 0.0f) || !(hy >= 0.0f)) continue;
        const float xa = px - hx, xb = px + hx, ya = py - hy, yb = py + hy;
        float icp = 1.0f / SGS_TILE;
        for (int z = 0; z <= zcap; ++z, icp *= 2.0f) {                // cells of cp = 16 >> z pixels: a cell holds the pixel centres cp t .. cp t + cp - 1
            const int cp = SGS_TILE >> z, gxz = (cam->width + cp - 1) / cp, gyz = (cam->height + cp - 1) / cp;
            const int nx = std::min(gxz, ifloor(xb * icp) + 1) - std::max(0, iceil((xa - (float)(cp - 1)) * icp));
            const int ny = std::min(gyz, ifloor(yb * icp) + 1) - std::max(0, iceil((ya - (float)(cp - 1)) * icp));
            if (nx > 0 && ny > 0) D[z] += (double)g.wgt * (double)(nx * ny);
        }
    }
    int z = 0;
    while (z < zcap && D[z + 1] <= ctx->fine_tile_growth * D[z]) ++z;     // (nothing of the probe in view: 0 <= 0, the pixel rule alone)
    return z;
}

int validate(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cam, const sgs_config* cfg,
             int& row_begin, int& row_end, const float* out_rgb, int* fine_shift = nullptr) {
    if (!scene || !cam || !out_rgb) SGS_FAIL(ctx, SGS_ERR_INVALID, "null scene / camera / output");
    if (cam->width <= 0 || cam->height <= 0 || cam->width > 65535 * SGS_TILE || cam->height > 65535 * SGS_TILE)
        SGS_FAIL(ctx, SGS_ERR_INVALID, "bad resolution %dx%d", cam->width, cam->height);
    if (!(cam->fx > 0.f) || !(cam->fy > 0.f)) SGS_FAIL(ctx, SGS_ERR_INVALID, "focal lengths must be positive");
    const int stride = cfg && cfg->tile_row_stride > 1 ? cfg->tile_row_stride : 1, phase = cfg && stride > 1 ? cfg->tile_row_phase : 0;
    if (phase < 0 || phase >= stride) SGS_FAIL(ctx, SGS_ERR_INVALID, "tile_row_phase %d outside [0, stride %d)", phase, stride);
    const int gy_frame = (cam->height + SGS_TILE - 1) / SGS_TILE;
    const int gy = gy_frame > phase ? (gy_frame - phase + stride - 1) / stride : 0;      // rows this call owns
    if (row_end < 0 || row_end > gy) row_end = gy;
    if (row_begin < 0) row_begin = 0;
    if (row_begin > row_end) SGS_FAIL(ctx, SGS_ERR_INVALID, "tile_row_begin %d > tile_row_end %d", row_begin, row_end);
    if (cfg && cfg->sh_degree > 3) SGS_FAIL(ctx, SGS_ERR_INVALID, "sh_degree %d > 3", cfg->sh_degree);
    if (cfg) {
        // what the kernels lean on: depth keys are the bits of a POSITIVE float >= near_z (unsigned order, the composite's
        // bucket index (key >> 18) - (bits(near_z) >> 18), 1 / tz), and the thresholds are compared as bit patterns
        if (!(cfg->near_z > 0.f) || !(cfg->far_z > cfg->near_z)) SGS_FAIL(ctx, SGS_ERR_INVALID, "need 0 < near_z < far_z (got %g, %g)", cfg->near_z, cfg->far_z);
        if (!(cfg->alpha_min > 0.f) || !(cfg->alpha_min < 1.f) || !(cfg->alpha_max >= cfg->alpha_min) || !(cfg->alpha_max < 1.f))
            SGS_FAIL(ctx, SGS_ERR_INVALID, "need 0 < alpha_min <= alpha_max < 1 (got %g, %g)", cfg->alpha_min, cfg->alpha_max);
        if (!(cfg->t_min > 0.f) || !(cfg->t_min < 1.f)) SGS_FAIL(ctx, SGS_ERR_INVALID, "need 0 < t_min < 1 (got %g)", cfg->t_min);
        if (!(cfg->dilation >= 0.f) || !(cfg->clamp > 0.f)) SGS_FAIL(ctx, SGS_ERR_INVALID, "need dilation >= 0 and clamp > 0");
    }
    {   // the view must be rigid (sage_gs.h): k_preprocess's fp32 screen-bound exclusion prices a Gaussian's footprint with
        // |J|_F^2 s_max^2, which a scaled or sheared view would break silently (a USD xformOp:scale != 1 has to be applied
        // to the Gaussians' means and scales by the caller)
        const float* V = cam->view;
        for (int r = 0; r < 3; ++r)
            for (int c = r; c < 3; ++c) {
                const double d = (double)V[4 * r] * V[4 * c] + (double)V[4 * r + 1] * V[4 * c + 1] + (double)V[4 * r + 2] * V[4 * c + 2];
                // (fp32 pose matrices are orthonormal to ~1e-6; the per-chunk bounds and the footprint bound of k_preprocess
                //  absorb ~1e-4 of non-rigidity, so the contract is an order of magnitude inside that)
                if (!(std::fabs(d - (r == c ? 1.0 : 0.0)) < 1.0e-5))
                    SGS_FAIL(ctx, SGS_ERR_INVALID, "camera view is not rigid: rows %d.%d of its 3x3 give %g", r, c, d);
            }
    }
    if (gy_frame > SGS_MAX_ROWS) SGS_FAIL(ctx, SGS_ERR_INVALID, "height %d exceeds %d tile rows", cam->height, SGS_MAX_ROWS);
    // level 1 of the binning keeps one counter per super-tile of the band in LDS (the band's rows and tiles as the kernels count them:
    // cells of (16 >> z)^2 pixels, sgs_common.h "Fine tiles")
    sgs_config cfg_d;
    if (!cfg) sgs_config_default(&cfg_d);
    const int z = fine_shift_of(ctx, scene, cam, cfg ? *cfg : cfg_d), cp = SGS_TILE >> z;
    if (fine_shift) *fine_shift = z;
    const int gx = (cam->width + cp - 1) / cp, rb = row_begin << z, re = row_end << z;
    const int64_t ns = (int64_t)((gx + SGS_ST - 1) / SGS_ST) * (((re + SGS_ST - 1) / SGS_ST) - rb / SGS_ST);
    if (ns > SGS_WT) SGS_FAIL(ctx, SGS_ERR_INVALID, "band of %d tile rows x %d tiles exceeds %d super-tiles", re - rb, gx, SGS_WT);
    return SGS_OK;
}

void fill_params(FrameParams& P, const sgs_ctx* ctx, const Lane& L, const sgs_scene* scene, const sgs_camera* cam,
                 const sgs_config& cfg, int row_begin, int row_end, int z) {
    memset(&P, 0, sizeof P);
    for (int i = 0; i < 12; ++i) P.view[i] = cam->view[i];
    const float* V = cam->view;
    for (int c = 0; c < 3; ++c)
        P.campos[c] = -((double)V[c] * V[3] + (double)V[4 + c] * V[7] + (double)V[8 + c] * V[11]);
    P.fx = cam->fx; P.fy = cam->fy; P.cx = cam->cx; P.cy = cam->cy;
    P.near_z = cfg.near_z; P.far_z = cfg.far_z; P.dilation = cfg.dilation; P.clamp = cfg.clamp;
    P.alpha_min = cfg.alpha_min; P.alpha_max = cfg.alpha_max; P.t_min = cfg.t_min;
    for (int c = 0; c < 3; ++c) P.bg[c] = cfg.bg[c];
    P.width = cam->width; P.height = cam->height;
    // Fine tiles (sgs_common.h): the grid, the band and every tile index the kernels see count CELLS of (16 >> z)^2 pixels; row_begin / row_end
    // arrive in 16-pixel rows (the C ABI's unit), each of which is 2^z rows of cells — but for the frame's last one when the height leaves it short
    const int cp = SGS_TILE >> z;                 // (z: fine_shift_of, decided once per call by validate)
    const int gx16 = (cam->width + SGS_TILE - 1) / SGS_TILE, gy16 = (cam->height + SGS_TILE - 1) / SGS_TILE;
    P.gx = (cam->width + cp - 1) / cp; P.gy = (cam->height + cp - 1) / cp;
    P.row_stride = cfg.tile_row_stride > 1 ? cfg.tile_row_stride : 1;
    P.row_phase = P.row_stride > 1 ? cfg.tile_row_phase : 0;
    P.row_begin = row_begin << z; P.row_end = row_end << z;
    if (row_end > row_begin && (row_end - 1) * P.row_stride + P.row_phase == gy16 - 1) P.row_end -= (gy16 << z) - P.gy;
    // a contiguous band ignores what projects outside its pixel rows; interleaved rows span the frame
    P.cull_y0 = P.row_stride > 1 ? 0 : SGS_TILE * row_begin;
    P.cull_y1 = P.row_stride > 1 ? SGS_TILE * gy16 : SGS_TILE * row_end;
    P.sh_degree = cfg.sh_degree < 0 ? scene->sh_degree : std::min(cfg.sh_degree, scene->sh_degree);
    P.sh_rows = scene->sh_rows;
    P.n = scene->n; P.n_chunks = scene->n_chunks;
    P.n_ranges = (int32_t)((scene->n + SGS_RANGE - 1) / SGS_RANGE);
    P.n_windows = 1;                                                              // (one window of super-tiles: sgs_kernels.h, level 1)
    P.limx = (double)P.clamp * (0.5 * (double)P.width / (double)P.fx); P.limy = (double)P.clamp * (0.5 * (double)P.height / (double)P.fy);
    P.rec_capacity = L.rec_cap;
    P.job_capacity = (int32_t)std::min<int64_t>(L.job_cap, 0x7fffffff);
    P.flags = (cfg.flags & ~SGS_PFLAG_INTERNAL) | (scene->sh_packed ? SGS_PFLAG_SH_PACKED | ((uint32_t)scene->sh_decode << SGS_PFLAG_SH_MODE_SHIFT) : 0u) |
              ((uint32_t)z << SGS_PFLAG_FINE_SHIFT);
    {   // k_chunk_cull's planes (sgs_kernels.h chunk_outside: the derivation and why each constant is conservative)
        const double lx = P.limx, ly = P.limy;
        P.cull_A = 1.001 * 3.0 * std::sqrt(2.0 * (2.0 + lx * lx + ly * ly)) * std::max((double)P.fx, (double)P.fy) * 1.0001;
        const double c0 = 1.001 * (3.0 * std::sqrt(2.0 * (double)P.dilation + 0.3163) + 1.0) + 0.5 + 1.0;      // + one pixel of slack
        P.cull_off[0] = (double)P.cx - 0.5 + c0;                                     // px + rb >= 0
        P.cull_off[1] = (double)(SGS_TILE * gx16) - (double)P.cx + 0.5 + c0;         // px - rb <  16 gx (gx: 16-pixel tiles)
        P.cull_off[2] = (double)P.cy - 0.5 - (double)P.cull_y0 + c0;                 // py + rb >= cull_y0
        P.cull_off[3] = (double)P.cull_y1 - (double)P.cy + 0.5 + c0;                 // py - rb <  cull_y1
        const double f[4] = {(double)P.fx, (double)P.fx, (double)P.fy, (double)P.fy};
        for (int k = 0; k < 4; ++k) P.cull_nrm[k] = std::sqrt(f[k] * f[k] + P.cull_off[k] * P.cull_off[k]);
    }
}

// ---- the launches of a frame (group), stage by stage.  G.s[0 .. nf) are the group's frames (same resolution and tile rows). ----
// S1-S3: per-chunk bounds -> the frame's live list, then the projection (k_chunk_cull, k_preprocess)
void launch_cull(const FrameGroup& G, int nf, hipStream_t stream, bool share = false) {
    const FrameParams& P = G.s[0].P;
    if (P.n_chunks > 0 && share)       // (the lists of all frames of the group + the group's work list, one lane per chunk)
        hipLaunchKernelGGL(sgs::k_chunk_cull_group, dim3((unsigned)((P.n_chunks + SGS_CULL_THREADS - 1) / SGS_CULL_THREADS)), dim3(SGS_CULL_THREADS), 0, stream,
                           G, (unsigned)nf);
    else if (P.n_chunks > 0)
        hipLaunchKernelGGL(sgs::k_chunk_cull, dim3((unsigned)((P.n_chunks + SGS_CULL_THREADS - 1) / SGS_CULL_THREADS), (unsigned)nf),
                           dim3(SGS_CULL_THREADS), 0, stream, G);
}
void launch_project(const sgs_ctx* ctx, const FrameGroup& G, int nf, hipStream_t stream) {
    const FrameParams& P = G.s[0].P;
    if (P.n_chunks <= 0) return;
    const bool fine = ((P.flags >> SGS_PFLAG_FINE_SHIFT) & 3u) != 0u;        // (fine tiles: the instantiation that scales the splats, sgs_common.h)
    const int64_t all = (P.n_chunks + 3) / 4, cap = std::max(256, ctx->pre_grid / nf);
    const bool narrow = nf > 1 && 2 * (P.row_end - P.row_begin) < P.gy && cap < all;
    // The full frames of a GROUP share what they read of the scene: one cull kernel writes every frame's live list and the group's work
    // list — (chunk, frame) pairs, chunk-major — and the projection's waves take THAT (k_preprocess_shared), so the frames that want a chunk
    // run side by side and its 15 KiB of rows come from HBM once.  Never slower than every frame walking its own list in its own part of
    // the grid, whatever the views share (profiles/r06zb: +-0 at an overlap of 1.3 of 4, -2 ... -5 % per frame at the bench's 1.4, -9 % at
    // 2.3, -7 ... -10 % for a trajectory's consecutive headings); frames bit-identical.  (SGS_NO_SHARE: A/B builds, scripts/build_variant.sh.)
#ifdef SGS_NO_SHARE
    const bool share = false;
#else
    const bool share = nf > 1 && !narrow;
#endif
    launch_cull(G, nf, stream, share);
    // a wave per chunk of the scene (most end at once) — except for a group of narrow bands, whose frames share
    // pre_grid workgroups that loop over the live list (r03y: 0.0454 -> 0.0425 ms per frame of a 3-row band)
    if (narrow) {
        if (fine) hipLaunchKernelGGL((sgs::k_preprocess<true, true>), dim3((unsigned)cap, (unsigned)nf), dim3(256), 0, stream, G);
        else hipLaunchKernelGGL((sgs::k_preprocess<true, false>), dim3((unsigned)cap, (unsigned)nf), dim3(256), 0, stream, G);
    } else if (share) {
        const unsigned grid = (unsigned)((all * nf + SGS_XCDS - 1) / SGS_XCDS + 1) * SGS_XCDS;       // (every XCD's eighth of the list, rounded up)
        if (fine) hipLaunchKernelGGL((sgs::k_preprocess_shared<true>), dim3(grid), dim3(256), 0, stream, G);
        else hipLaunchKernelGGL((sgs::k_preprocess_shared<false>), dim3(grid), dim3(256), 0, stream, G);
    } else {
        if (fine) hipLaunchKernelGGL((sgs::k_preprocess<false, true>), dim3((unsigned)all, (unsigned)nf), dim3(256), 0, stream, G);
        else hipLaunchKernelGGL((sgs::k_preprocess<false, false>), dim3((unsigned)all, (unsigned)nf), dim3(256), 0, stream, G);
    }
}
// S4, two levels (six launches); ev_mid (nullable) is recorded between the levels
int launch_binning(sgs_ctx* ctx, const FrameGroup& G, int nf, hipStream_t stream, hipEvent_t ev_mid) {
    const FrameParams& P = G.s[0].P;
    const unsigned F = (unsigned)nf;
    const int gx = P.gx, row_begin = P.row_begin, row_end = P.row_end;
    // the binning grids are sized per LAUNCH: the frames of a group share them (a group of four band frames with 512 + 2048
    // workgroups EACH spent its time starting workgroups that found a chunk or a job apiece: 0.047 -> 0.041 ms per frame
    // of a 3-row band, 0.057 -> 0.051 of an 18-row one, r03y)
    const unsigned bin_blocks = (unsigned)std::min<int64_t>(std::max(32, ctx->bin_grid / nf), P.n_ranges);
    // level 1: splats -> super-tile queues (count, scan + level-2 job list, emit)
    const int gxs = (gx + SGS_ST - 1) / SGS_ST, ns = gxs * ((row_end + SGS_ST - 1) / SGS_ST - row_begin / SGS_ST);
    const size_t win_bytes = (size_t)((std::max(ns, 1) + 127) / 128) * 128 * sizeof(unsigned);
    const bool bin = P.n_ranges > 0 && ns > 0;
    if (bin) hipLaunchKernelGGL(sgs::k_bin_count, dim3(bin_blocks, F), dim3(SGS_BIN_THREADS), win_bytes, stream, G);
    hipLaunchKernelGGL(sgs::k_stile_scan, dim3(1, F), dim3(SGS_SSCAN_THREADS), 0, stream, G);
    if (bin) hipLaunchKernelGGL(sgs::k_bin_emit, dim3(bin_blocks, F), dim3(SGS_BIN_THREADS), win_bytes, stream, G);
    if (ev_mid) SGS_HIP(ctx, hipEventRecord(ev_mid, stream));
    // level 2: super-tile queues -> tile queues (count, scan of the tile counters + render order, emit)
    const unsigned exp_grid = (unsigned)std::min<int64_t>(std::max(64, ctx->exp_grid / nf), (int64_t)ns + (P.n + SGS_SEG - 1) / SGS_SEG);
    if (bin) hipLaunchKernelGGL((sgs::k_expand<false>), dim3(std::max(1u, exp_grid), F), dim3(SGS_EXP_THREADS), 0, stream, G);
    // one workgroup per SGS_SCAN_THREADS tiles of the band (16 at 1080p, 64 at 3840x2160), independent of each other
    const unsigned scan_groups = std::max(1u, ((unsigned)((row_end - row_begin) * gx) + SGS_SCAN_THREADS - 1) / SGS_SCAN_THREADS);
    hipLaunchKernelGGL(sgs::k_tile_scan, dim3(scan_groups, F), dim3(SGS_SCAN_THREADS), 0, stream, G);
    if (bin) hipLaunchKernelGGL((sgs::k_expand<true>), dim3(std::max(1u, exp_grid), F), dim3(SGS_EXP_THREADS), 0, stream, G);
    return SGS_OK;
}
// the final transmittance of stopped pixels (one more add per pixel and splat) only where something reads it
bool need_tf_of(const sgs_config& cfg, const float* out_aux) { return out_aux || cfg.bg[0] != 0.f || cfg.bg[1] != 0.f || cfg.bg[2] != 0.f; }
// S5 + S6: one workgroup per tile of the band
void launch_composite(const FrameGroup& G, int nf, hipStream_t stream, bool aux, bool count_df, bool need_tf) {
    const FrameParams& P = G.s[0].P;
    const unsigned F = (unsigned)nf, ntiles = (unsigned)((P.row_end - P.row_begin) * P.gx);
    if (ntiles == 0) return;
    const unsigned grid = ((ntiles + 7u) / 8u) * 8u;
    // (D_f is counted only on request: the per-pixel bookkeeping and the end-of-tile reduction cost ~4 % of a sweep)
    if (aux) {
        if (count_df) hipLaunchKernelGGL((sgs::k_tile_render<true, true, true>), dim3(grid, F), dim3(256), 0, stream, G);
        else hipLaunchKernelGGL((sgs::k_tile_render<true, false, true>), dim3(grid, F), dim3(256), 0, stream, G);
    } else if (need_tf) {
        if (count_df) hipLaunchKernelGGL((sgs::k_tile_render<false, true, true>), dim3(grid, F), dim3(256), 0, stream, G);
        else hipLaunchKernelGGL((sgs::k_tile_render<false, false, true>), dim3(grid, F), dim3(256), 0, stream, G);
    } else {
        if (count_df) hipLaunchKernelGGL((sgs::k_tile_render<false, true, false>), dim3(grid, F), dim3(256), 0, stream, G);
        else hipLaunchKernelGGL((sgs::k_tile_render<false, false, false>), dim3(grid, F), dim3(256), 0, stream, G);
    }
}

// The FrameGroup of nf frames of one scene: frame f uses the intermediates of lane set0 + f (grown here if need be) and status slot
// slot0 + f.
int build_group(sgs_ctx* ctx, FrameGroup& G, const sgs_scene* scene, const sgs_camera* cams, int nf, const sgs_config& cfg,
                int row_begin, int row_end, float* const* outs, int slot0, float* out_aux, int set0, int z) {
    int rc;
    const int cp = SGS_TILE >> z;                 // (the frames of a group share a resolution, a configuration and the fine-tile shift)
    const int gx = (cams->width + cp - 1) / cp, gy = (cams->height + cp - 1) / cp;
    memset(&G, 0, sizeof G);
    G.geom = scene->geom; G.shq = scene->shq; G.cbound = scene->cbound; G.row_acc = ctx->row_acc;
    for (int f = 0; f < nf; ++f) {
        Lane& A = ctx->lanes[set0 + f];
        if ((rc = ensure_splats(ctx, A, scene->n)) != SGS_OK) return rc;
        if ((rc = ensure_tiles(ctx, A, gx * gy)) != SGS_OK) return rc;
        if ((rc = ensure_records(ctx, A)) != SGS_OK) return rc;
        FrameSlot& S = G.s[f];
        fill_params(S.P, ctx, A, scene, &cams[f], cfg, row_begin, row_end, z);
        if ((S.P.flags & SGS_FLAG_FULL_SORT) && (rc = ensure_sorted_out(ctx, A)) != SGS_OK) return rc;
        S.splats = A.splats; S.vismask = A.vismask; S.bigmask = A.bigmask; S.big_list = A.big_list; S.binrec = A.binrec;
        S.live_list = A.live_list;
        S.tile_count = A.tile_count; S.tile_offset = A.tile_offset; S.tile_order = A.tile_order;
        S.blk_list = A.blk_list; S.blk_len = A.blk_len;
        S.stile_count = A.stile_count; S.stile_offset = A.stile_offset; S.jobs = A.jobs; S.job_base = A.job_base;
        S.rec = A.rec; S.alt = A.alt; S.part = A.part; S.sorted_out = A.sorted_out;
        S.tile_prof = A.tile_prof; S.bin_prof = A.bin_prof;
        S.out_rgb = outs[f]; S.out_aux = out_aux; S.st = ctx->d_status + slot0 + f;
    }
    return SGS_OK;
}

// bookkeeping of "the most recent frame" (what sgs_frame_sync reports and sgs_debug_read looks at)
void note_last(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cam, const FrameParams& P, int slot, int lane, bool timed,
               hipStream_t caller_stream) {
    const int gx = P.gx, gy = P.gy, row_begin = P.row_begin, row_end = P.row_end;
    ctx->last_slot = slot; ctx->last_timed = timed; ctx->last_stream = caller_stream; ctx->last_lane = lane;
    ctx->last_n = scene->n; ctx->last_tiles = (row_end - row_begin) * gx; ctx->last_sh_rows = scene->sh_rows;
    ctx->last_T = gx * gy;
    ctx->last_t_lo = row_begin * gx; ctx->last_t_hi = row_end * gx;
    ctx->last_scene = scene;
    int64_t pixel_rows = 0;                       // pixel rows of the frame this call wrote
    const int z = (int)((P.flags >> SGS_PFLAG_FINE_SHIFT) & 3u), cp = SGS_TILE >> z;     // (rows of cells: sgs_common.h "Fine tiles")
    for (int k = row_begin; k < row_end; ++k) {
        const int k16 = k >> z;
        const int y0 = (((k16 * P.row_stride + P.row_phase) << z) + (k - (k16 << z))) * cp;
        pixel_rows += std::max(0, std::min(y0 + cp, cam->height) - y0);
    }
    ctx->last_pixels = pixel_rows * cam->width;
}

// Enqueue a GROUP of nf <= SGS_MAX_GROUP frames of one scene — same resolution, same tile rows, one camera each — as
// ONE set of five launches (blockIdx.y = frame; sgs_common.h FrameGroup).  Frame f uses the intermediates of lane
// set0 + f and status slot slot0 + f; the launches go to `stream`.
//   * ordinary frames: nf = 1, set0 = 0, stream = the caller's;
//   * SGS_FLAG_PIPELINED frames: nf = 1, the next lane and its own stream, forked from the caller's stream;
//   * sgs_render_batch*: groups of ctx->group frames rotating over ctx->group_lanes streams (in_batch: the caller forks
//     the streams, zeroes and collects the status slots and waits ONCE per batch — the per-frame runtime calls around the
//     launches were ~40 us, as long as a light band of tile rows takes on the GPU).
int enqueue_group(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cams, int nf, const sgs_config& cfg,
                  int row_begin, int row_end, float* const* outs, int slot0, hipStream_t caller_stream, bool timed,
                  float* out_aux, bool pipelined, bool in_batch, int set0, int z, int stream_lane = -1) {
    int rc;
    // The stream: a pipelined frame's own lane's; a batch's groups rotate over the streams of lanes 0 .. group_lanes-1 — the
    // SAME streams single pipelined frames use.  (r03y: the groups used to run on the streams of lanes 0 and 4; a process that
    // had issued one batch and then pipelined single frames owned four lane streams + the caller's, the runtime maps streams onto
    // four hardware queues, two lanes shared one and the sweep was 11 % slower — 4180 vs 3750 frames/s — for the rest of the process.)
    Lane& L = ctx->lanes[stream_lane >= 0 ? stream_lane : set0];
    hipStream_t stream = caller_stream;
    if (pipelined) {
        if ((rc = ensure_lane_stream(ctx, L)) != SGS_OK) return rc;
        stream = L.stream;
    }
    FrameGroup G;
    if ((rc = build_group(ctx, G, scene, cams, nf, cfg, row_begin, row_end, outs, slot0, out_aux, set0, z)) != SGS_OK) return rc;
    const FrameParams& P = G.s[0].P;
    if (pipelined && !in_batch) {
        // start after whatever the caller already put on its stream (scene upload, consumers of the output buffer)
        SGS_HIP(ctx, hipEventRecord(L.fork, caller_stream));
        SGS_HIP(ctx, hipStreamWaitEvent(L.stream, L.fork, 0));
    } else if (!pipelined) {
        for (int f = 0; f < nf; ++f)               // these lanes' buffers may still be in use by a pipelined frame
            if (ctx->lanes[set0 + f].busy) SGS_HIP(ctx, hipStreamWaitEvent(caller_stream, ctx->lanes[set0 + f].done, 0));
    }
    FrameStatus* st = ctx->d_status + slot0;
    if (!in_batch) SGS_HIP(ctx, hipMemsetAsync(st, 0, sizeof(FrameStatus) * (size_t)nf, stream));
    hipEvent_t* ev = nullptr;
    if (timed) {
        if (!ctx->ev) {
            ctx->ev = new (std::nothrow) hipEvent_t[kStatusRing][SGS_NUM_STAGES + 1];
            if (!ctx->ev) SGS_FAIL(ctx, SGS_ERR_OOM, "out of host memory");
            for (int i = 0; i < kStatusRing; ++i)
                for (int j = 0; j <= SGS_NUM_STAGES; ++j) SGS_HIP(ctx, hipEventCreate(&ctx->ev[i][j]));
        }
        ev = ctx->ev[slot0];
    }
    for (int f = 0; f < nf; ++f) ctx->slot_timed[slot0 + f] = timed && f == 0;
    if (timed) SGS_HIP(ctx, hipEventRecord(ev[0], stream));

    launch_project(ctx, G, nf, stream);
    if (timed) SGS_HIP(ctx, hipEventRecord(ev[1], stream));
    launch_binning(ctx, G, nf, stream, timed ? ev[2] : nullptr);
    if (timed) SGS_HIP(ctx, hipEventRecord(ev[3], stream));
    launch_composite(G, nf, stream, out_aux != nullptr, (cfg.flags & SGS_FLAG_STATS) != 0, need_tf_of(cfg, out_aux));
    if (timed) SGS_HIP(ctx, hipEventRecord(ev[4], stream));
    SGS_HIP(ctx, hipGetLastError());
    if (!in_batch) {
        SGS_HIP(ctx, hipMemcpyAsync(ctx->h_status + slot0, st, sizeof(FrameStatus) * (size_t)nf, hipMemcpyDeviceToHost, stream));
        if (pipelined) {
            SGS_HIP(ctx, hipEventRecord(L.done, L.stream));
            L.busy = true;
        }
    }

    note_last(ctx, scene, &cams[nf - 1], P, slot0 + nf - 1, set0 + nf - 1, timed, caller_stream);    // "the last frame" = the group's last
    return SGS_OK;
}

// The record capacity an overflowed frame is known to need: its D when level 2 of the binning counted it, and twice its D_s (the
// super-tile queues take half the capacity) — an overflow at level 1 leaves d_total at 0.
int64_t records_needed(const FrameStatus& s) { return std::max<int64_t>((int64_t)s.d_total, 2 * (int64_t)s.ds_total); }

// Fill `stats` from the (already synchronised) status of ring slot `slot`.
void collect(sgs_ctx* ctx, int slot, sgs_stats* stats, int64_t n, int ntiles, int64_t pixels, int sh_rows,
             bool timed) {
    if (!stats) return;
    const FrameStatus& s = ctx->h_status[slot];
    memset(stats, 0, sizeof *stats);
    stats->n_gaussians = n;
    stats->n_visible = s.n_visible;
    stats->d_total = s.d_total;
    stats->d_fetched = (int64_t)s.d_fetched;
    stats->n_pixels = pixels;
    stats->n_tiles = ntiles;
    stats->max_tile_len = (int32_t)s.max_tile_len;
    stats->n_spill_tiles = (int32_t)s.class_count[3];
    stats->n_deep_windows = (int64_t)s.n_deep;
    stats->retries = ctx->last_retries;
    // Algorithmic bytes per stage — DESIGN.md §4 (what the stage must move, not what it happens to).
    const int64_t nv = s.n_visible, D = s.d_total, Df = (int64_t)s.d_fetched;
    stats->bytes[SGS_STAGE_PREPROCESS] = 16 * n + (32 + 16 * (int64_t)sh_rows + 64 + 16) * nv;   // rows read; splat + binning record written
    // two-level binning (D_s = records in the super-tile queues): level 1 reads every visible splat's 16-B binning record in
    // both of its passes and writes D_s of them; level 2 reads those in both of its passes and writes the D 8-B tile records
    const int64_t Ds = s.ds_total;
    stats->d_super = Ds;
    stats->bytes[SGS_STAGE_COUNT] = 2 * 16 * nv + 16 * Ds;
    stats->bytes[SGS_STAGE_EMIT] = 2 * 16 * Ds + 8 * D + 8 * ((int64_t)ctx->last_T + 1);
    stats->bytes[SGS_STAGE_RENDER] = 8 * D + 36 * Df + 12 * pixels;               // every record seen once, D_f splats blended
    if (timed && ctx->ev) {
        hipEvent_t* ev = ctx->ev[slot];
        for (int i = 0; i < SGS_NUM_STAGES; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) stats->ms[i] = ms;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[0], ev[SGS_NUM_STAGES]) == hipSuccess) stats->ms_total = ms;
    }
}

}  // namespace

extern "C" {

int sgs_version(void) { return SGS_VERSION; }

void sgs_struct_sizes(int32_t* camera_bytes, int32_t* config_bytes, int32_t* stats_bytes) {
    if (camera_bytes) *camera_bytes = (int32_t)sizeof(sgs_camera);
    if (config_bytes) *config_bytes = (int32_t)sizeof(sgs_config);
    if (stats_bytes) *stats_bytes = (int32_t)sizeof(sgs_stats);
}

void sgs_config_default(sgs_config* cfg) {
    if (!cfg) return;
    cfg->near_z = 0.2f; cfg->far_z = 1.0e30f; cfg->dilation = 0.3f; cfg->clamp = 1.3f;
    cfg->alpha_min = 1.0f / 255.0f; cfg->alpha_max = 0.99f; cfg->t_min = 1.0e-4f;
    cfg->bg[0] = cfg->bg[1] = cfg->bg[2] = 0.f;
    cfg->sh_degree = -1; cfg->flags = 0;
    cfg->tile_row_stride = 1; cfg->tile_row_phase = 0;
}

const char* sgs_last_error(const sgs_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int sgs_create(int device_id, int backend, sgs_ctx** out) {
    if (!out) { g_create_error = "sgs_create: out is NULL"; return SGS_ERR_INVALID; }
    *out = nullptr;
    if (backend != SGS_BACKEND_HIP) {
        g_create_error = "sgs_create: only SGS_BACKEND_HIP exists; the CPU restatement lives in oracle/ (tests only)";
        return SGS_ERR_BACKEND;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("sgs_create: no HIP device (") + hipGetErrorString(e) + ")";
        return SGS_ERR_HIP;
    }
    if (device_id < 0 || device_id >= count) { g_create_error = "sgs_create: device_id out of range"; return SGS_ERR_INVALID; }
    sgs_ctx* ctx = new (std::nothrow) sgs_ctx;
    if (!ctx) { g_create_error = "sgs_create: out of host memory"; return SGS_ERR_OOM; }
    ctx->device = device_id;
    auto fail = [&](const char* what, hipError_t err) {
        g_create_error = std::string("sgs_create: ") + what + ": " + hipGetErrorString(err);
        sgs_destroy(ctx);
        return SGS_ERR_HIP;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return fail("hipSetDevice", e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_status), sizeof(FrameStatus) * kStatusRing)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_status), sizeof(FrameStatus) * kStatusRing, 0)) != hipSuccess) return fail("hipHostMalloc", e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->row_acc), sizeof(unsigned long long) * SGS_MAX_ROWS)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMemset(ctx->row_acc, 0, sizeof(unsigned long long) * SGS_MAX_ROWS)) != hipSuccess) return fail("hipMemset", e);
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return fail("hipStreamSynchronize", e);
    // (nothing is read from the environment: sgs_set_tuning is the library's whole tuning surface)
    *out = ctx;
    return SGS_OK;
}

int sgs_destroy(sgs_ctx* ctx) {
    if (!ctx) return SGS_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (Lane& L : ctx->lanes) {
        void* bufs[] = {L.splats, L.vismask, L.bigmask, L.big_list, L.binrec, L.live_list, L.tile_count, L.tile_offset, L.tile_order,
                        L.tile_prof, L.bin_prof, L.blk_list, L.blk_len, L.rec, L.alt, L.part, L.sorted_out,
                        L.stile_count, L.stile_offset, L.jobs, L.job_base};
        for (void* b : bufs) if (b) (void)hipFree(b);
        if (L.stream) (void)hipStreamDestroy(L.stream);
        if (L.fork) (void)hipEventDestroy(L.fork);
        if (L.done) (void)hipEventDestroy(L.done);
    }
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->row_acc) (void)hipFree(ctx->row_acc);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    if (ctx->ev) {
        for (int i = 0; i < kStatusRing; ++i)
            for (int j = 0; j <= SGS_NUM_STAGES; ++j) (void)hipEventDestroy(ctx->ev[i][j]);
        delete[] ctx->ev;
    }
    delete ctx;
    return SGS_OK;
}

void sgs_tuning_default(sgs_tuning* out) {
    if (!out) return;
    out->lanes = 3; out->group = 8; out->group_lanes = 2; out->morton = 1; out->record_capacity = 16ll << 20;
    out->fine_tile_pixels = 640 * 480; out->fine_tile_growth = 2.2;
}

int sgs_get_tuning(const sgs_ctx* ctx, sgs_tuning* out) {
    if (!ctx || !out) return SGS_ERR_INVALID;
    out->lanes = ctx->n_lanes; out->group = ctx->group; out->group_lanes = ctx->group_lanes; out->morton = ctx->morton ? 1 : 0;
    out->record_capacity = ctx->rec_cap_wanted; out->fine_tile_pixels = ctx->fine_tile_pixels; out->fine_tile_growth = ctx->fine_tile_growth;
    return SGS_OK;
}

int sgs_set_tuning(sgs_ctx* ctx, const sgs_tuning* t) {
    if (!ctx) return SGS_ERR_INVALID;
    if (!t) SGS_FAIL(ctx, SGS_ERR_INVALID, "sgs_set_tuning: tuning is NULL");
    if (t->lanes < 1 || t->lanes > kMaxLanes) SGS_FAIL(ctx, SGS_ERR_INVALID, "sgs_tuning.lanes %d outside [1, %d]", t->lanes, kMaxLanes);
    if (t->group < 1 || t->group > std::min(kMaxLanes, SGS_MAX_GROUP)) SGS_FAIL(ctx, SGS_ERR_INVALID, "sgs_tuning.group %d outside [1, %d]", t->group, std::min(kMaxLanes, SGS_MAX_GROUP));
    if (t->group_lanes < 1 || t->group * t->group_lanes > kMaxLanes)
        SGS_FAIL(ctx, SGS_ERR_INVALID, "sgs_tuning.group x group_lanes = %d x %d exceeds the %d lanes of a context", t->group, t->group_lanes, kMaxLanes);
    if (t->record_capacity <= 0) SGS_FAIL(ctx, SGS_ERR_INVALID, "sgs_tuning.record_capacity must be positive");
    if (t->fine_tile_pixels < 0) SGS_FAIL(ctx, SGS_ERR_INVALID, "sgs_tuning.fine_tile_pixels must not be negative (0 = never)");
    if (!(t->fine_tile_growth >= 1.0)) SGS_FAIL(ctx, SGS_ERR_INVALID, "sgs_tuning.fine_tile_growth must be at least 1 (>= 16: whenever the pixel rule allows)");
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();                          // frames in flight were issued under the old values (their verdicts stay for sgs_frame_sync)
    ctx->n_lanes = t->lanes; ctx->next_lane = 0; ctx->group = t->group; ctx->group_lanes = t->group_lanes; ctx->morton = t->morton != 0;
    ctx->fine_tile_pixels = t->fine_tile_pixels; ctx->fine_tile_growth = t->fine_tile_growth;
    if (t->record_capacity != ctx->rec_cap_wanted) return sgs_set_record_capacity(ctx, t->record_capacity);
    return SGS_OK;
}

int sgs_set_record_capacity(sgs_ctx* ctx, int64_t max_records) {
    if (!ctx) return SGS_ERR_INVALID;
    if (max_records <= 0) SGS_FAIL(ctx, SGS_ERR_INVALID, "max_records must be positive");
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    ctx->rec_cap_wanted = max_records;
    for (Lane& L : ctx->lanes) {      // force reallocation at the requested size (lanes other than 0: on next use)
        L.rec_cap = 0;
        if (&L != &ctx->lanes[0]) {
            for (unsigned long long** q : {&L.rec, &L.alt, &L.part}) if (*q) { (void)hipFree(*q); *q = nullptr; }
        }
    }
    return ensure_records(ctx, ctx->lanes[0]);
}

namespace {

// The device side of a scene load (sgs_kernels.h "Upload"): Z-order permutation of the means by a radix sort on the device, layout
// (dequantising when the source is the compressed payload), per-chunk bounds.  src = the five fp32 arrays on the device, or Z.
int layout_scene(sgs_ctx* ctx, sgs_scene* sc, const float* const* src, const sgs::PackedScene& Z, bool packed) {
    const int64_t n = sc->n;
    const int nf = 3 * (sc->sh_degree + 1) * (sc->sh_degree + 1);
    const size_t npad = (size_t)std::max<int64_t>(sc->n_chunks, 1) * 64;
    hipError_t e;
    unsigned long long* keys[2] = {nullptr, nullptr};
    unsigned* idx[2] = {nullptr, nullptr};
    unsigned *hist = nullptr, *bounds = nullptr;
    const unsigned* d_perm = nullptr;
    int rc = SGS_OK;
    auto fail = [&](const char* what, hipError_t err) {
        ctx->err = std::string("scene layout: ") + what + ": " + hipGetErrorString(err);
        rc = err == hipErrorOutOfMemory ? SGS_ERR_OOM : SGS_ERR_HIP;
    };
    const float* means = packed ? nullptr : src[0];
    if (ctx->morton && n > SGS_WAVE) {
        // Z-order (Morton) permutation of the means, once per scene: 64 consecutive Gaussians then occupy a compact cell, so a chunk is
        // visible or culled as a whole (no half-used SH cache lines in k_preprocess) and its splats overlap on screen (binning)
        const unsigned nblocks = (unsigned)((n + SGS_RSORT_TILE - 1) / SGS_RSORT_TILE);
        for (int k = 0; k < 2 && rc == SGS_OK; ++k) {
            if ((e = hipMalloc(reinterpret_cast<void**>(&keys[k]), (size_t)n * 8)) != hipSuccess) fail("hipMalloc", e);
            else if ((e = hipMalloc(reinterpret_cast<void**>(&idx[k]), (size_t)n * 4)) != hipSuccess) fail("hipMalloc", e);
        }
        const unsigned nscan = (256u * nblocks + SGS_RSCAN_SPAN - 1) / SGS_RSCAN_SPAN;          // (hist, then k_radix_scan's span sums)
        if (rc == SGS_OK && (e = hipMalloc(reinterpret_cast<void**>(&hist), ((size_t)256 * nblocks + nscan) * 4)) != hipSuccess) fail("hipMalloc", e);
        if (rc == SGS_OK && (e = hipMalloc(reinterpret_cast<void**>(&bounds), 6 * 4)) != hipSuccess) fail("hipMalloc", e);
        if (rc == SGS_OK) {
            const unsigned init[6] = {~0u, ~0u, ~0u, 0u, 0u, 0u};
            if ((e = hipMemcpy(bounds, init, sizeof init, hipMemcpyHostToDevice)) != hipSuccess) fail("hipMemcpy", e);
        }
        if (rc == SGS_OK) {
            const unsigned g1 = (unsigned)std::min<int64_t>(512, (n + 255) / 256), gn = (unsigned)((n + 255) / 256);
            if (packed) {
                hipLaunchKernelGGL((sgs::k_mean_bounds<true>), dim3(g1), dim3(256), 0, 0, (long long)n, means, Z, bounds);
                hipLaunchKernelGGL((sgs::k_morton_keys<true>), dim3(gn), dim3(256), 0, 0, (long long)n, means, Z, bounds, keys[0], idx[0]);
            } else {
                hipLaunchKernelGGL((sgs::k_mean_bounds<false>), dim3(g1), dim3(256), 0, 0, (long long)n, means, Z, bounds);
                hipLaunchKernelGGL((sgs::k_morton_keys<false>), dim3(gn), dim3(256), 0, 0, (long long)n, means, Z, bounds, keys[0], idx[0]);
            }
            int cur = 0;
            for (int shift = 0; shift < 64; shift += 8, cur ^= 1) {      // 63 key bits: eight stable passes
                hipLaunchKernelGGL(sgs::k_radix_count, dim3(nblocks), dim3(256), 0, 0, (long long)n, keys[cur], shift, nblocks, hist);
                hipLaunchKernelGGL(sgs::k_radix_scan, dim3(nscan), dim3(1024), 0, 0, 256u * nblocks, hist, hist + (size_t)256 * nblocks);
                hipLaunchKernelGGL(sgs::k_radix_scatter, dim3(nblocks), dim3(256), 0, 0, (long long)n, keys[cur], idx[cur], keys[cur ^ 1], idx[cur ^ 1],
                                   shift, nblocks, hist, hist + (size_t)256 * nblocks);
            }
            d_perm = idx[cur];               // (an even number of passes: back in buffer 0)
            sc->perm_host = (unsigned*)malloc((size_t)n * 4);
            if (!sc->perm_host) { ctx->err = "scene layout: out of host memory"; rc = SGS_ERR_OOM; }
            else if ((e = hipMemcpy(sc->perm_host, d_perm, (size_t)n * 4, hipMemcpyDeviceToHost)) != hipSuccess) fail("reading the permutation", e);
        }
    }
    if (rc == SGS_OK) {
        const unsigned grid = (unsigned)((npad + 255) / 256);
        if (packed)
            hipLaunchKernelGGL((sgs::k_scene_layout<true>), dim3(grid), dim3(256), 0, 0, (long long)n, nf, sc->sh_rows, d_perm,
                               nullptr, nullptr, nullptr, nullptr, nullptr, Z, sc->geom, sc->shq);
        else
            hipLaunchKernelGGL((sgs::k_scene_layout<false>), dim3(grid), dim3(256), 0, 0, (long long)n, nf, sc->sh_rows, d_perm,
                               src[0], src[1], src[2], src[3], src[4], Z, sc->geom, sc->shq);
        hipLaunchKernelGGL(sgs::k_chunk_bounds, dim3((unsigned)((sc->n_chunks + 3) / 4)), dim3(256), 0, 0, (long long)n,
                           (long long)sc->n_chunks, sc->geom, sc->cbound);
        if ((e = hipDeviceSynchronize()) != hipSuccess) fail("k_scene_layout / k_chunk_bounds", e);
    }
    if (rc == SGS_OK && n > 0) {
        // the probe (fine_shift_of): a pre-sample of up to 64 Ki Gaussians at even strides through the layout comes to the host; 512 of them
        // are drawn from it — systematically, with probability 1/2 (1 / M0 + trace_i / sum of traces) each — and their covariances formed once
        const int M0 = (int)std::min<int64_t>(n, 65536), M = std::min(M0, 512);
        float4* d_probe = nullptr;
        std::vector<float4> rows((size_t)M0 * SGS_GEOM_ROWS);
        if ((e = hipMalloc(reinterpret_cast<void**>(&d_probe), rows.size() * sizeof(float4))) != hipSuccess) fail("hipMalloc", e);
        else {
            hipLaunchKernelGGL(sgs::k_probe_gather, dim3((unsigned)((M0 + 255) / 256)), dim3(256), 0, 0, (long long)n, M0, sc->geom, d_probe);
            if ((e = hipMemcpy(rows.data(), d_probe, rows.size() * sizeof(float4), hipMemcpyDeviceToHost)) != hipSuccess) fail("reading the probe", e);
            (void)hipFree(d_probe);
        }
        if (rc == SGS_OK) {
            std::vector<double> tr((size_t)M0);
            double T = 0.0;
            for (int i = 0; i < M0; ++i) {
                const float4 g1 = rows[(size_t)3 * i + 1];
                const double t = (double)g1.x * g1.x + (double)g1.y * g1.y + (double)g1.z * g1.z;       // trace(R S S^T R^T) = |s|^2
                tr[(size_t)i] = t < 1.0e30 ? t : 0.0;                                                 // (NaN / inf scales: never drawn by size)
                T += tr[(size_t)i];
            }
            sc->probe.clear(); sc->probe.reserve((size_t)M);
            double cum = 0.0; int i = 0;
            for (int k = 0; k < M; ++k) {
                double pi_i = 1.0 / M0;
                if (M0 > M) {              // (a scene smaller than a probe: every Gaussian once, weight 1)
                    const double target = ((double)k + 0.5) / M;
                    for (;;) {
                        pi_i = 0.5 * (1.0 / M0 + (T > 0.0 ? tr[(size_t)i] / T : 1.0 / M0));
                        if (cum + pi_i >= target || i == M0 - 1) break;
                        cum += pi_i; ++i;
                    }
                } else i = k;
                const float4 g0 = rows[(size_t)3 * i], g1 = rows[(size_t)3 * i + 1], g2 = rows[(size_t)3 * i + 2];
                ProbeSample q;
                q.m[0] = g0.x; q.m[1] = g0.y; q.m[2] = g0.z; q.ln_o = std::log(g0.w);
                q.wgt = (float)(1.0 / (pi_i * M0));                       // (relative to a uniform draw)
                const double qn = std::sqrt((double)g1.w * g1.w + (double)g2.x * g2.x + (double)g2.y * g2.y + (double)g2.z * g2.z);
                const double w = g1.w / qn, x = g2.x / qn, y = g2.y / qn, z = g2.z / qn;
                const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                                        {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                                        {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
                const double s2[3] = {(double)g1.x * g1.x, (double)g1.y * g1.y, (double)g1.z * g1.z};
                int t = 0;
                for (int a_ = 0; a_ < 3; ++a_)
                    for (int b_ = a_; b_ < 3; ++b_)
                        q.S[t++] = (float)(R[a_][0] * R[b_][0] * s2[0] + R[a_][1] * R[b_][1] * s2[1] + R[a_][2] * R[b_][2] * s2[2]);     // (00, 01, 02, 11, 12, 22)
                q.trace = q.S[0] + q.S[3] + q.S[5];
                sc->probe.push_back(q);
            }
        }
    }
    for (int k = 0; k < 2; ++k) { if (keys[k]) (void)hipFree(keys[k]); if (idx[k]) (void)hipFree(idx[k]); }
    if (hist) (void)hipFree(hist);
    if (bounds) (void)hipFree(bounds);
    return rc;
}

// a new scene object with its device buffers
int new_scene(sgs_ctx* ctx, int64_t n, int sh_degree, bool sh_packed, sgs_scene** out) {
    sgs_scene* sc = new (std::nothrow) sgs_scene;
    if (!sc) SGS_FAIL(ctx, SGS_ERR_OOM, "out of host memory");
    const int nf = 3 * (sh_degree + 1) * (sh_degree + 1);
    sc->n = n; sc->n_chunks = (n + 63) / 64; sc->sh_degree = sh_degree; sc->sh_packed = sh_packed;
    sc->sh_rows = sh_packed ? (12 + (nf - 3) + 15) / 16 : (nf + 3) / 4;      // packed: 12 B of fp32 DC + one byte per higher coefficient
    const size_t npad = (size_t)std::max<int64_t>(sc->n_chunks, 1) * 64;
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&sc->geom), npad * SGS_GEOM_ROWS * sizeof(float4))) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void**>(&sc->shq), npad * sc->sh_rows * sizeof(float4))) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void**>(&sc->cbound), (npad / 64) * 2 * sizeof(float4))) != hipSuccess) {
        ctx->err = std::string("scene upload: hipMalloc: ") + hipGetErrorString(e);
        sgs_scene_free(ctx, sc);
        return e == hipErrorOutOfMemory ? SGS_ERR_OOM : SGS_ERR_HIP;
    }
    *out = sc;
    return SGS_OK;
}

// host arrays -> device copies (on_device: the caller's pointers as they are).  staged[] holds what must be freed.
int stage(sgs_ctx* ctx, int on_device, const void* const* src, const size_t* bytes, int count, const void** dev, void** staged) {
    for (int i = 0; i < count; ++i) {
        staged[i] = nullptr;
        if (on_device || !src[i] || bytes[i] == 0) { dev[i] = src[i]; continue; }
        hipError_t e;
        if ((e = hipMalloc(&staged[i], bytes[i])) != hipSuccess || (e = hipMemcpy(staged[i], src[i], bytes[i], hipMemcpyHostToDevice)) != hipSuccess) {
            ctx->err = std::string("scene upload: staging: ") + hipGetErrorString(e);
            return e == hipErrorOutOfMemory ? SGS_ERR_OOM : SGS_ERR_HIP;
        }
        dev[i] = staged[i];
    }
    return SGS_OK;
}

}  // namespace

int sgs_scene_upload(sgs_ctx* ctx, int64_t n, int sh_degree, const float* means, const float* scales,
                     const float* quats, const float* opacities, const float* sh, int on_device,
                     sgs_scene** out) {
    if (!ctx) return SGS_ERR_INVALID;
    if (!out) SGS_FAIL(ctx, SGS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n < 0 || n > 0x7fffffffll) SGS_FAIL(ctx, SGS_ERR_INVALID, "n = %lld out of range", (long long)n);
    if (sh_degree < 0 || sh_degree > 3) SGS_FAIL(ctx, SGS_ERR_INVALID, "sh_degree %d not in 0..3", sh_degree);
    if (n > 0 && (!means || !scales || !quats || !opacities || !sh)) SGS_FAIL(ctx, SGS_ERR_INVALID, "null input array");
    SGS_HIP(ctx, hipSetDevice(ctx->device));
    sgs_scene* sc = nullptr;
    int rc;
    if ((rc = new_scene(ctx, n, sh_degree, false, &sc)) != SGS_OK) return rc;
    if (n > 0) {
        const int nf = 3 * (sh_degree + 1) * (sh_degree + 1);
        const void* src[5] = {means, scales, quats, opacities, sh};
        const size_t bytes[5] = {(size_t)n * 12, (size_t)n * 12, (size_t)n * 16, (size_t)n * 4, (size_t)n * nf * 4};
        const void* dev[5]; void* staged[5] = {};
        rc = stage(ctx, on_device, src, bytes, 5, dev, staged);
        if (rc == SGS_OK) {
            const float* f[5] = {(const float*)dev[0], (const float*)dev[1], (const float*)dev[2], (const float*)dev[3], (const float*)dev[4]};
            sgs::PackedScene Z = {nullptr, nullptr, nullptr, 0};
            rc = layout_scene(ctx, sc, f, Z, false);
        }
        for (void* q : staged) if (q) (void)hipFree(q);
        if (rc != SGS_OK) { sgs_scene_free(ctx, sc); return rc; }
    }
    *out = sc;
    return SGS_OK;
}

int sgs_scene_upload_compressed(sgs_ctx* ctx, const sgs_compressed_scene* z, int on_device, sgs_scene** out) {
    if (!ctx) return SGS_ERR_INVALID;
    if (!out || !z) SGS_FAIL(ctx, SGS_ERR_INVALID, "null argument");
    *out = nullptr;
    const int64_t n = z->n;
    if (n < 0 || n > 0x7fffffffll) SGS_FAIL(ctx, SGS_ERR_INVALID, "n = %lld out of range", (long long)n);
    if (z->sh_degree < 0 || z->sh_degree > 3) SGS_FAIL(ctx, SGS_ERR_INVALID, "sh_degree %d not in 0..3", z->sh_degree);
    const int k_rest = (z->sh_degree + 1) * (z->sh_degree + 1) - 1;
    if (z->sh_decode < SGS_SH_DECODE_UNSPECIFIED || z->sh_decode > SGS_SH_DECODE_BIN_CENTRE) SGS_FAIL(ctx, SGS_ERR_INVALID, "sh_decode %d is not one of SGS_SH_DECODE_*", z->sh_decode);
    if (k_rest > 0 && z->sh_decode == SGS_SH_DECODE_UNSPECIFIED)
        SGS_FAIL(ctx, SGS_ERR_INVALID, "sh_decode is required for a scene with 8-bit SH coefficients (sh_degree %d): SGS_SH_DECODE_BIN_CENTRE, _LINEAR255 or "
                                       "_BIN_CENTRE_ENDS — there is no default, include/sage_gs.h says why", z->sh_degree);
    if (n > 0 && (!z->chunks || !z->packed || (k_rest > 0 && !z->sh))) SGS_FAIL(ctx, SGS_ERR_INVALID, "null input array");
    if (z->n_chunks != (n + 255) / 256) SGS_FAIL(ctx, SGS_ERR_INVALID, "n_chunks %lld is not ceil(n / 256)", (long long)z->n_chunks);
    SGS_HIP(ctx, hipSetDevice(ctx->device));
    sgs_scene* sc = nullptr;
    int rc;
    if ((rc = new_scene(ctx, n, z->sh_degree, true, &sc)) != SGS_OK) return rc;
    // (the kernels' mode: 0 = bin centre — one exact fma —, 1 = linear255, 2 = bin centre with exact ends)
    sc->sh_decode = (z->sh_decode == SGS_SH_DECODE_BIN_CENTRE || z->sh_decode == SGS_SH_DECODE_UNSPECIFIED) ? 0 : z->sh_decode;
    if (n > 0) {
        const void* src[3] = {z->chunks, z->packed, z->sh};
        const size_t bytes[3] = {(size_t)z->n_chunks * 18 * 4, (size_t)n * 16, (size_t)n * 3 * k_rest};
        const void* dev[3]; void* staged[3] = {};
        rc = stage(ctx, on_device, src, bytes, 3, dev, staged);
        if (rc == SGS_OK) {
            sgs::PackedScene Z = {(const float*)dev[0], (const uint4*)dev[1], (const unsigned char*)dev[2], k_rest};
            rc = layout_scene(ctx, sc, nullptr, Z, true);
        }
        for (void* q : staged) if (q) (void)hipFree(q);
        if (rc != SGS_OK) { sgs_scene_free(ctx, sc); return rc; }
    }
    *out = sc;
    return SGS_OK;
}

int sgs_scene_free(sgs_ctx* ctx, sgs_scene* scene) {
    if (!scene) return SGS_OK;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipDeviceSynchronize(); }
    if (scene->geom) (void)hipFree(scene->geom);
    if (scene->shq) (void)hipFree(scene->shq);
    if (scene->cbound) (void)hipFree(scene->cbound);
    if (scene->perm_host) free(scene->perm_host);
    if (ctx && ctx->last_scene == scene) ctx->last_scene = nullptr;
    delete scene;
    return SGS_OK;
}

int sgs_frame_sync(sgs_ctx* ctx, sgs_stats* stats) {
    if (!ctx) return SGS_ERR_INVALID;
    if (ctx->last_slot < 0) SGS_FAIL(ctx, SGS_ERR_INVALID, "no frame has been issued");
    SGS_HIP(ctx, hipSetDevice(ctx->device));
    SGS_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    { int rc_ = drain_lanes(ctx); if (rc_ != SGS_OK) return rc_; }
    collect(ctx, ctx->last_slot, stats, ctx->last_n, ctx->last_tiles, ctx->last_pixels, ctx->last_sh_rows, ctx->last_timed);
    // every frame issued since the previous synchronisation is checked, not just the last one
    int bad = -1, n_bad = 0;
    double ms_sum[SGS_NUM_STAGES + 1] = {};
    int n_timed = 0;
    for (int k = 0; k < ctx->pending_count; ++k) {
        const int slot = (ctx->pending_begin + k) % kStatusRing;
        if (ctx->h_status[slot].overflow) { bad = slot; ++n_bad; }
        if (ctx->slot_timed[slot] && ctx->ev) {
            float ms = 0.f;
            for (int i = 0; i < SGS_NUM_STAGES; ++i)
                if (hipEventElapsedTime(&ms, ctx->ev[slot][i], ctx->ev[slot][i + 1]) == hipSuccess) ms_sum[i] += ms;
            if (hipEventElapsedTime(&ms, ctx->ev[slot][0], ctx->ev[slot][SGS_NUM_STAGES]) == hipSuccess) ms_sum[SGS_NUM_STAGES] += ms;
            ++n_timed;
        }
    }
    if (stats && n_timed > 1) {          // several pipelined frames: report the per-frame average
        for (int i = 0; i < SGS_NUM_STAGES; ++i) stats->ms[i] = (float)(ms_sum[i] / n_timed);
        stats->ms_total = (float)(ms_sum[SGS_NUM_STAGES] / n_timed);
    }
    ctx->pending_begin = ctx->next_slot; ctx->pending_count = 0;
    if (n_bad)
        SGS_FAIL(ctx, SGS_ERR_OVERFLOW, "%d frame(s) overflowed the record capacity %lld (one needs at least %lld records: D = %u, D_s = %u "
                 "super-tile records of which the capacity holds half as many); call sgs_set_record_capacity", n_bad,
                 (long long)ctx->lanes[0].rec_cap, (long long)records_needed(ctx->h_status[bad]), ctx->h_status[bad].d_total, ctx->h_status[bad].ds_total);
    return SGS_OK;
}

int sgs_render_rgbd(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cam, const sgs_config* cfg_in,
                    int tile_row_begin, int tile_row_end, float* out_rgb, float* out_aux, sgs_stats* stats,
                    void* hip_stream) {
    if (!ctx) return SGS_ERR_INVALID;
    int rc;
    int z = 0;                                       // the frame's fine-tile shift (fine_shift_of)
    if ((rc = validate(ctx, scene, cam, cfg_in, tile_row_begin, tile_row_end, out_rgb, &z)) != SGS_OK) return rc;
    sgs_config cfg;
    if (cfg_in) cfg = *cfg_in; else sgs_config_default(&cfg);
    SGS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const bool timed = (cfg.flags & SGS_FLAG_TIMING) != 0;
    ctx->last_retries = 0;
    for (;;) {
        if (ctx->pending_count == kStatusRing) {        // the status ring is full: drain it first
            if ((rc = sgs_frame_sync(ctx, nullptr)) != SGS_OK) return rc;
        }
        const int slot = ctx->next_slot;
        ctx->next_slot = (ctx->next_slot + 1) % kStatusRing;
        if (ctx->pending_count == 0) ctx->pending_begin = slot;
        ctx->pending_count++;
        const bool pipelined = (cfg.flags & SGS_FLAG_PIPELINED) && (cfg.flags & SGS_FLAG_ASYNC) && ctx->n_lanes > 1 &&
                               !(cfg.flags & SGS_FLAG_FULL_SORT);
        int lane = 0;
        if (pipelined) { lane = ctx->next_lane; ctx->next_lane = (ctx->next_lane + 1) % ctx->n_lanes; }
        float* outs[1] = {out_rgb};
        if ((rc = enqueue_group(ctx, scene, cam, 1, cfg, tile_row_begin, tile_row_end, outs, slot, stream, timed, out_aux,
                                pipelined, false, lane, z)) != SGS_OK)
            return rc;
        if (cfg.flags & SGS_FLAG_ASYNC) return SGS_OK;
        rc = sgs_frame_sync(ctx, stats);
        if (rc != SGS_ERR_OVERFLOW) return rc;
        // synchronous path: grow the queues to fit and render again.  A frame can overflow at level 1 of the binning (the super-tile
        // queues hold rec_cap / 2 records of 16 bytes): level 2 then never ran and d_total is 0 — what is known is D_s
        const int64_t need = records_needed(ctx->h_status[slot]);
        ctx->rec_cap_wanted = std::max<int64_t>(need + need / 4, ctx->lanes[0].rec_cap * 2);
        ctx->lanes[0].rec_cap = 0;
        if ((rc = ensure_records(ctx, ctx->lanes[0])) != SGS_OK) return rc;
        if (++ctx->last_retries > 4) SGS_FAIL(ctx, SGS_ERR_OVERFLOW, "record capacity still too small after 4 retries");
    }
}

int sgs_render(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cam, const sgs_config* cfg,
               int tile_row_begin, int tile_row_end, float* out_rgb, sgs_stats* stats, void* hip_stream) {
    return sgs_render_rgbd(ctx, scene, cam, cfg, tile_row_begin, tile_row_end, out_rgb, nullptr, stats, hip_stream);
}

int sgs_render_batch_strided(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cams, int n_cams,
                             const sgs_config* cfg_in, int tile_row_begin, int tile_row_end, float* out_rgb,
                             int64_t frame_stride, sgs_stats* stats, void* hip_stream) {
    if (!ctx) return SGS_ERR_INVALID;
    if (n_cams < 0 || (n_cams > 0 && !cams)) SGS_FAIL(ctx, SGS_ERR_INVALID, "bad camera array");
    sgs_config cfg;
    if (cfg_in) cfg = *cfg_in; else sgs_config_default(&cfg);
    cfg.flags &= ~(uint32_t)SGS_FLAG_TIMING;       // per-stage events are a single-frame facility
    SGS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    ctx->last_retries = 0;
    int rc;
    if (ctx->pending_count > 0 && (rc = sgs_frame_sync(ctx, nullptr)) != SGS_OK) return rc;
    const bool lanes = !(cfg.flags & SGS_FLAG_FULL_SORT);           // own streams (the FULL_SORT test hook stays on the caller's)
    const int F = ctx->group, GL = lanes ? ctx->group_lanes : 1;
    // every camera (and the stride) is validated BEFORE anything is enqueued: an error return half-way through a batch
    // would leave earlier groups running on the lane streams with nobody waiting for them
    int rb0 = tile_row_begin, re0 = tile_row_end;
    std::vector<signed char> zs((size_t)std::max(n_cams, 1), 0);     // every frame's fine-tile shift (fine_shift_of: a function of its camera)
    for (int i = 0; i < n_cams; ++i) {
        int rb = tile_row_begin, re = tile_row_end, z = 0;
        if ((rc = validate(ctx, scene, &cams[i], &cfg, rb, re, out_rgb, &z)) != SGS_OK) return rc;
        zs[(size_t)i] = (signed char)z;
        if (cams[i].width != cams[0].width || cams[i].height != cams[0].height)
            SGS_FAIL(ctx, SGS_ERR_INVALID, "the cameras of a batch must share a resolution");
        if (i == 0) { rb0 = rb; re0 = re; }
    }
    if (n_cams > 1) {
        const int stride_t = cfg.tile_row_stride > 1 ? cfg.tile_row_stride : 1, phase_t = stride_t > 1 ? cfg.tile_row_phase : 0;
        int64_t rows = 0;                           // pixel rows one frame of the batch writes
        for (int k = rb0; k < re0; ++k) {
            const int y0 = (k * stride_t + phase_t) * SGS_TILE;
            rows += std::max(0, std::min(y0 + SGS_TILE, cams[0].height) - y0);
        }
        if (frame_stride < rows * (int64_t)cams[0].width * 3)
            SGS_FAIL(ctx, SGS_ERR_INVALID, "frame_stride %lld is smaller than the band a frame writes (%lld floats)",
                     (long long)frame_stride, (long long)(rows * (int64_t)cams[0].width * 3));
    }
    for (int c0 = 0; c0 < n_cams; c0 += kStatusRing) {
        const int cn = std::min(kStatusRing, n_cams - c0);
        int64_t px[kStatusRing]; int tl[kStatusRing];
        // once per chunk of frames, not once per frame: zero the status slots, fork the group streams from the caller's
        SGS_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(FrameStatus) * (size_t)cn, stream));
        if (lanes) {
            if ((rc = ensure_lane_stream(ctx, ctx->lanes[0])) != SGS_OK) return rc;
            SGS_HIP(ctx, hipEventRecord(ctx->lanes[0].fork, stream));
            for (int gl = 0; gl < GL; ++gl) {
                if ((rc = ensure_lane_stream(ctx, ctx->lanes[gl])) != SGS_OK) return rc;
                SGS_HIP(ctx, hipStreamWaitEvent(ctx->lanes[gl].stream, ctx->lanes[0].fork, 0));
            }
        }
        // The chunk's frames are dealt to the group streams in EQUAL shares (the stream with an extra group finished it alone, without a
        // neighbour's kernels to overlap with; more than four frames are worth a second stream), each share cut into EQUAL groups of <= F:
        // 20 frames on two streams are 5,5 + 5,5 under F = 8 (not 8,2 + 8,2: the frames of a group share their reads of the scene and their
        // launches, a group of two shares little).
        const int n_streams = lanes ? std::min(GL, (cn + 3) / 4) : 1;
        int left[kMaxLanes], todo[kMaxLanes];           // frames / groups each stream still has to issue
        for (int sidx = 0; sidx < n_streams; ++sidx) {
            left[sidx] = cn / n_streams + (sidx < cn % n_streams ? 1 : 0);
            todo[sidx] = (left[sidx] + F - 1) / F;
        }
        for (int i = 0, g = 0; i < cn; ++g) {
            const int sidx = g % n_streams;
            if (left[sidx] <= 0) continue;
            if (todo[sidx] <= 0) todo[sidx] = (left[sidx] + F - 1) / F;       // (a group that ended early, below, left frames behind)
            int nf = (left[sidx] + todo[sidx] - 1) / todo[sidx];
            --todo[sidx];
            // (the frames of a group share one set of launches, hence one grid of tiles: a group ends where the fine-tile shift changes)
            for (int f = 1; f < nf; ++f) if (zs[(size_t)(c0 + i + f)] != zs[(size_t)(c0 + i)]) { nf = f; break; }
            left[sidx] -= nf;
            float* outs[SGS_MAX_GROUP];
            const int rb = rb0, re = re0;
            for (int f = 0; f < nf; ++f) outs[f] = out_rgb + (size_t)(c0 + i + f) * (size_t)frame_stride;
            if ((rc = enqueue_group(ctx, scene, &cams[c0 + i], nf, cfg, rb, re, outs, i, stream, false, nullptr, lanes, true,
                                    sidx * F, zs[(size_t)(c0 + i)], sidx)) != SGS_OK)
                return rc;
            for (int f = 0; f < nf; ++f) { px[i + f] = ctx->last_pixels; tl[i + f] = ctx->last_tiles; }
            i += nf;
        }
        // ... wait for the lanes and fetch every frame's status in one copy
        if (lanes) for (int gl = 0; gl < GL; ++gl) SGS_HIP(ctx, hipStreamSynchronize(ctx->lanes[gl].stream));
        SGS_HIP(ctx, hipStreamSynchronize(stream));
        if ((rc = drain_lanes(ctx)) != SGS_OK) return rc;           // (frames issued outside this call)
        SGS_HIP(ctx, hipMemcpy(ctx->h_status, ctx->d_status, sizeof(FrameStatus) * (size_t)cn, hipMemcpyDeviceToHost));
        // The redo below goes through sgs_render, which takes ring slots of its own (from next_slot, two or more per
        // grow-and-retry) — i.e. slots of frames of THIS chunk that have not been looked at yet.  So every frame's
        // verdict and statistics are taken out of the ring before anything is re-rendered.
        bool over[kStatusRing];
        for (int i = 0; i < cn; ++i) {
            over[i] = ctx->h_status[i].overflow != 0;
            if (stats) collect(ctx, i, stats + c0 + i, scene->n, tl[i], px[i], scene->sh_rows, false);
        }
        ctx->next_slot = 0; ctx->pending_begin = 0; ctx->pending_count = 0;
        for (int i = 0; i < cn; ++i) {
            if (over[i]) {
                // redo this one frame synchronously (grows the queues), then carry on
                int rb = tile_row_begin, re = tile_row_end;
                float* out = out_rgb + (size_t)(c0 + i) * (size_t)frame_stride;
                sgs_config c1 = cfg; c1.flags &= ~(uint32_t)(SGS_FLAG_ASYNC | SGS_FLAG_PIPELINED);
                if ((rc = sgs_render(ctx, scene, &cams[c0 + i], &c1, rb, re, out, stats ? stats + c0 + i : nullptr, hip_stream)) != SGS_OK)
                    return rc;
            }
        }
        ctx->next_slot = 0; ctx->pending_begin = 0; ctx->pending_count = 0;
    }
    return SGS_OK;
}

int sgs_render_batch(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cams, int n_cams,
                     const sgs_config* cfg_in, int tile_row_begin, int tile_row_end, float* out_rgb,
                     sgs_stats* stats, void* hip_stream) {
    if (!ctx) return SGS_ERR_INVALID;
    if (n_cams < 0 || (n_cams > 0 && !cams)) SGS_FAIL(ctx, SGS_ERR_INVALID, "bad camera array");
    for (int i = 1; i < n_cams; ++i)
        if (cams[i].width != cams[0].width || cams[i].height != cams[0].height)
            SGS_FAIL(ctx, SGS_ERR_INVALID, "the cameras of a batch must share a resolution");
    const int64_t stride = n_cams > 0 ? (int64_t)cams[0].width * cams[0].height * 3 : 0;
    return sgs_render_batch_strided(ctx, scene, cams, n_cams, cfg_in, tile_row_begin, tile_row_end, out_rgb, stride, stats, hip_stream);
}

int sgs_row_records(sgs_ctx* ctx, int64_t* out, int n_rows, int reset) {
    if (!ctx) return SGS_ERR_INVALID;
    if (n_rows < 0 || n_rows > SGS_MAX_ROWS || (n_rows > 0 && !out)) SGS_FAIL(ctx, SGS_ERR_INVALID, "bad row buffer (%d rows)", n_rows);
    SGS_HIP(ctx, hipSetDevice(ctx->device));
    // (no device-wide synchronisation: a sweep's framebuffer exchange may be in flight on another stream and must not
    //  be waited for here; the frames the caller has synchronised are complete, and that is the contract)
    static_assert(sizeof(unsigned long long) == sizeof(int64_t), "row counters");
    if (n_rows > 0) SGS_HIP(ctx, hipMemcpy(out, ctx->row_acc, sizeof(int64_t) * (size_t)n_rows, hipMemcpyDeviceToHost));
    if (reset) {
        SGS_HIP(ctx, hipMemset(ctx->row_acc, 0, sizeof(unsigned long long) * SGS_MAX_ROWS));
        SGS_HIP(ctx, hipStreamSynchronize(nullptr));      // (see ensure_tiles: the clear must not race later frames)
    }
    return SGS_OK;
}

int sgs_pack_rgba8(sgs_ctx* ctx, const float* rgb, uint8_t* rgba, int width, int height, void* hip_stream) {
    if (!ctx) return SGS_ERR_INVALID;
    if (!rgb || !rgba || width <= 0 || height <= 0) SGS_FAIL(ctx, SGS_ERR_INVALID, "bad pack arguments");
    SGS_HIP(ctx, hipSetDevice(ctx->device));
    const long long n = (long long)width * height;
    hipLaunchKernelGGL(sgs::k_pack_rgba8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(hip_stream), rgb, reinterpret_cast<unsigned*>(rgba), n);
    SGS_HIP(ctx, hipGetLastError());
    return SGS_OK;
}

int64_t sgs_debug_read(sgs_ctx* ctx, int what, void* host_dst, int64_t bytes) {
    if (!ctx) return SGS_ERR_INVALID;
    if (ctx->last_slot < 0) SGS_FAIL(ctx, SGS_ERR_INVALID, "no frame has been rendered");
    SGS_HIP(ctx, hipSetDevice(ctx->device));
    SGS_HIP(ctx, hipDeviceSynchronize());
    const FrameStatus& s = ctx->h_status[ctx->last_slot];
    const Lane& L = ctx->lanes[ctx->last_lane];
    const int64_t n_chunks = (ctx->last_n + 63) / 64;
    const int64_t n_slots = n_chunks * 64;
    const void* src = nullptr;
    int64_t have = 0, elem = 0;
    switch (what) {
        case SGS_BUF_TILE_OFFSETS: have = ((int64_t)ctx->last_T + 1) * 4; break;       // every 8th sub-queue offset
        case SGS_BUF_SORTED_SLOTS: src = L.sorted_out; have = (s.overflow || !L.sorted_out) ? 0 : (int64_t)s.d_total * 4; break;
        case SGS_BUF_SLOT_IDS: elem = 4; have = n_slots * elem; break;
        case SGS_BUF_SPLATS: elem = 48; have = n_slots * elem; break;          // the 12-word view documented in sage_gs.h
        case SGS_BUF_CHUNK_SKIPPED: have = n_chunks; break;
        case SGS_BUF_SCENE_GEOM: have = ctx->last_scene ? ctx->last_n * 11 * 4 : 0; break;
        case SGS_BUF_SCENE_SH: have = ctx->last_scene ? ctx->last_n * 3 * (ctx->last_scene->sh_degree + 1) * (ctx->last_scene->sh_degree + 1) * 4 : 0; break;
        case 100: src = L.tile_prof; have = (int64_t)ctx->last_T * 8 * SGS_PROF_WORDS; break;    // profiling build only
        case 101: src = L.bin_prof; have = (int64_t)SGS_BIN_BLOCKS * 64; break;  // profiling build only
        default: SGS_FAIL(ctx, SGS_ERR_INVALID, "unknown buffer id %d", what);
    }
    const int64_t n = std::min(have, bytes);
    if (n <= 0 || !host_dst) return have;
    if (src) SGS_HIP(ctx, hipMemcpy(host_dst, src, (size_t)n, hipMemcpyDeviceToHost));
    if (what == SGS_BUF_TILE_OFFSETS) {
        const size_t cnt = (size_t)ctx->last_T + 1;
        unsigned* tmp = (unsigned*)malloc(cnt * 4);
        if (!tmp) SGS_FAIL(ctx, SGS_ERR_OOM, "out of host memory");
        hipError_t e = hipMemcpy(tmp, L.tile_offset, cnt * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(tmp); SGS_FAIL(ctx, SGS_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e)); }
        // k_tile_scan writes the offsets of the band it rendered (and the band's end); outside it they are constant
        const unsigned total = tmp[(size_t)ctx->last_t_hi];
        for (int64_t i = 0; (i + 1) * 4 <= n; ++i)
            ((unsigned*)host_dst)[i] = i < ctx->last_t_lo ? 0u : i >= ctx->last_t_hi ? total : tmp[(size_t)i];
        free(tmp);
    }
    if (what == SGS_BUF_SCENE_GEOM) {
        std::vector<float4> rows((size_t)n_slots * SGS_GEOM_ROWS);
        hipError_t e = hipMemcpy(rows.data(), ctx->last_scene->geom, rows.size() * sizeof(float4), hipMemcpyDeviceToHost);
        if (e != hipSuccess) SGS_FAIL(ctx, SGS_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e));
        float* dst = (float*)host_dst;
        for (int64_t p = 0; p < ctx->last_n; ++p) {
            const int64_t chunk = p >> 6, lane = p & 63;
            const float4 g0 = rows[(size_t)((chunk * SGS_GEOM_ROWS + 0) * 64 + lane)], g1 = rows[(size_t)((chunk * SGS_GEOM_ROWS + 1) * 64 + lane)],
                         g2 = rows[(size_t)((chunk * SGS_GEOM_ROWS + 2) * 64 + lane)];
            unsigned i; memcpy(&i, &g2.w, 4);
            if ((int64_t)(i + 1) * 11 * 4 > n) continue;
            float* o = dst + (size_t)i * 11;
            o[0] = g0.x; o[1] = g0.y; o[2] = g0.z; o[3] = g0.w; o[4] = g1.x; o[5] = g1.y; o[6] = g1.z; o[7] = g1.w; o[8] = g2.x; o[9] = g2.y; o[10] = g2.z;
        }
    }
    if (what == SGS_BUF_SCENE_SH) {
        const sgs_scene* sc = ctx->last_scene;
        const int nf = 3 * (sc->sh_degree + 1) * (sc->sh_degree + 1), rows_n = sc->sh_rows;
        std::vector<float4> rows((size_t)n_slots * rows_n), g2((size_t)n_slots);
        hipError_t e = hipMemcpy(rows.data(), sc->shq, rows.size() * sizeof(float4), hipMemcpyDeviceToHost);
        if (e != hipSuccess) SGS_FAIL(ctx, SGS_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e));
        std::vector<float4> geo((size_t)n_slots * SGS_GEOM_ROWS);
        if ((e = hipMemcpy(geo.data(), sc->geom, geo.size() * sizeof(float4), hipMemcpyDeviceToHost)) != hipSuccess) SGS_FAIL(ctx, SGS_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e));
        float* dst = (float*)host_dst;
        std::vector<unsigned> w((size_t)rows_n * 4);
        for (int64_t p = 0; p < ctx->last_n; ++p) {
            const int64_t chunk = p >> 6, lane = p & 63;
            unsigned i; memcpy(&i, &geo[(size_t)((chunk * SGS_GEOM_ROWS + 2) * 64 + lane)].w, 4);       // the original index
            if ((int64_t)(i + 1) * nf * 4 > n) continue;
            for (int r = 0; r < rows_n; ++r) memcpy(&w[(size_t)4 * r], &rows[(size_t)((chunk * rows_n + r) * 64 + lane)], 16);
            float* o = dst + (size_t)i * nf;
            if (!sc->sh_packed) memcpy(o, w.data(), (size_t)nf * 4);
            else {      // 12 B of fp32 DC, then a byte per coefficient: v / 32 - 4 + 1 / 64, exact in fp32 (sgs_kernels.h sgs_sh_byte)
                memcpy(o, w.data(), 12);
                for (int j = 0; j < nf - 3; ++j) {      // sgs_kernels.h sgs_sh_byte / sgs_sh_byte_mode, restated: one correctly rounded fma (exact in double, rounded once)
                    const unsigned v = (w[(size_t)3 + (j >> 2)] >> (8 * (j & 3))) & 0xffu;
                    volatile double prod = (double)v * (8.0 / 255.0);         // (two roundings, as the kernel's __dmul_rn / __dsub_rn)
                    float x = sc->sh_decode == 1 ? (float)(prod - 4.0) : (float)((double)v * (1.0 / 32.0) + (-4.0 + 1.0 / 64.0));
                    if (sc->sh_decode == 2) x = v == 0u ? -4.0f : v == 255u ? 4.0f : x;
                    o[3 + j] = x;
                }
            }
        }
    }
    if (what == SGS_BUF_CHUNK_SKIPPED) {
        std::vector<unsigned long long> vm((size_t)std::max<int64_t>(1, n_chunks)), bm(vm.size());
        hipError_t e = hipMemcpy(vm.data(), L.vismask, (size_t)n_chunks * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(bm.data(), L.bigmask, (size_t)n_chunks * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) SGS_FAIL(ctx, SGS_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e));
        for (int64_t i = 0; i < n; ++i) ((unsigned char*)host_dst)[i] = vm[(size_t)i] == 0ull && bm[(size_t)i] == ~0ull;
    }
    if (what == SGS_BUF_SPLATS) {
        // device records (sgs_common.h: 64 B, conic pre-scaled for the composite, opacity also as an exponent offset) -> x,y,conic a,b | c,opacity,r,g | b,depth,rect01,rect23
        const int64_t cnt = n / elem;
        std::vector<Splat> tmp((size_t)std::max<int64_t>(1, cnt));
        hipError_t e = hipMemcpy(tmp.data(), L.splats, (size_t)cnt * sizeof(Splat), hipMemcpyDeviceToHost);
        if (e != hipSuccess) SGS_FAIL(ctx, SGS_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e));
        const double l2e = 1.4426950408889634;
        for (int64_t i = 0; i < cnt; ++i) {
            const Splat& sp = tmp[(size_t)i];
            float* o = (float*)((char*)host_dst + i * elem);
            o[0] = sp.x; o[1] = sp.y;
            // (the record holds the roots of the completed square A (dx + k dy)^2 + C' dy^2 in the slots A, B, C: a = sqrt(A),
            //  a k, c = +-sqrt(|C'|);  B = 2 A k = 2 a (a k),  C = A k^2 + C' = (a k)^2 + C')
            const double a_ = sp.A, ak_ = sp.B, c_ = sp.C;
            const double A_ = a_ * a_, Cp_ = c_ * std::fabs(c_);
            o[2] = (float)(A_ / (0.5 * l2e)); o[3] = (float)(2.0 * a_ * ak_ / l2e); o[4] = (float)((ak_ * ak_ + Cp_) / (0.5 * l2e));
            o[5] = sp.o; o[6] = sp.r; o[7] = sp.g; o[8] = sp.b;
            memcpy(o + 9, &sp.key, 4); memcpy(o + 10, &sp.rect01, 4); memcpy(o + 11, &sp.rect23, 4);
        }
    }
    if (elem) {
        // a splat lives at its Gaussian's index; the per-chunk visibility masks say which are live.
        // Dead slots are blanked (slot ids -> 0xFFFFFFFF, splats -> 0) so stale data cannot pass for live.
        unsigned long long* vm = (unsigned long long*)malloc((size_t)std::max<int64_t>(1, n_chunks) * 8);
        if (!vm) SGS_FAIL(ctx, SGS_ERR_OOM, "out of host memory");
        hipError_t e = hipMemcpy(vm, L.vismask, (size_t)n_chunks * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(vm); SGS_FAIL(ctx, SGS_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e)); }
        char* dst = (char*)host_dst;
        // vismask is indexed by layout position; slots by original index
        const unsigned* perm = ctx->last_scene ? ctx->last_scene->perm_host : nullptr;
        std::vector<unsigned char> live((size_t)n_slots, 0);
        for (int64_t p = 0; p < ctx->last_n; ++p)
            if ((vm[p >> 6] >> (p & 63)) & 1ull) live[perm ? perm[p] : (size_t)p] = 1;
        for (int64_t i = 0; i < n_slots && (i + 1) * elem <= n; ++i) {
            if (what == SGS_BUF_SLOT_IDS) ((unsigned*)dst)[i] = live[(size_t)i] ? (unsigned)i : 0xFFFFFFFFu;
            else if (!live[(size_t)i]) memset(dst + i * elem, 0, (size_t)elem);
        }
        free(vm);
    }
    return have;
}

}  // extern "C"
