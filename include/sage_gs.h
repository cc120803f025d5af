/* sage_gs.h — C ABI of libsage_gs.so, the MI355X-native 3D-Gaussian-splatting scene renderer that
 * stands in for the Isaac Sim render step of Galery23/SAGE-3D_Official.
 *
 * The reference has NO FFI for this path: it drives a closed renderer through the Isaac Sim Camera
 * protocol.  Each entry point below names the reference call site(s) it replaces (paths relative to
 * the reference root).  Signatures are plain C: pointers, sizes, PODs; no C++ or
 * torch types, no exceptions.  Every function returns 0 on success or a negative sgs_status; the
 * message of the last failure on a context is available from sgs_last_error().
 *
 * Ownership: the caller owns every input and output buffer.  The library owns only its per-context
 * scratch (freed by sgs_destroy) and the re-laid-out scene copy (freed by sgs_scene_free).
 * Threading: a context is bound to one HIP device and is not re-entrant; work is stream-ordered on
 * the stream passed in (NULL = the device's default stream).  Distinct contexts may be used from
 * distinct threads/processes (one process per GPU is the intended deployment).
 *
 * There is deliberately no CPU backend behind this ABI: sgs_create(…, SGS_BACKEND_CPU, …) fails
 * with SGS_ERR_BACKEND.  The CPU restatement lives in oracle/ and is test infrastructure only.
 */
#ifndef SAGE_GS_H
#define SAGE_GS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGS_VERSION 114            /* major*100 + minor.  The version changes whenever a struct below changes size or meaning
                                    * (100 -> 101: sgs_stats grew d_super; 110: round-4 entry points; 111: sgs_stats grew n_deep_windows, sgs_compressed_scene.reserved_ became sh_decode; 112: SGS_FLAG_NO_DEEP; sgs_set_tuning,
                                    * SGS_BUF_SCENE_SH; 113: fine tiles — sgs_tuning grew fine_tile_pixels, SGS_FLAG_NO_FINE_TILES;
                                    * 114: sgs_tuning grew fine_tile_growth): a caller compiled against
                                    * another header MUST refuse to run — check sgs_version() == SGS_VERSION and, for bindings
                                    * that restate the structs by hand (ctypes, cgo), sgs_struct_sizes() — before the first call
                                    * that takes a struct.  The library writes whole structs (sgs_stats arrays with ITS stride). */
#define SGS_TILE 16                /* 16x16-pixel tiles (BASELINE.json north_star) */

typedef enum sgs_status {
    SGS_OK = 0,
    SGS_ERR_INVALID = -1,          /* bad argument */
    SGS_ERR_HIP = -2,              /* a HIP runtime call failed */
    SGS_ERR_OOM = -3,              /* device allocation failed */
    SGS_ERR_OVERFLOW = -4,         /* more (Gaussian,tile) records than the record capacity */
    SGS_ERR_BACKEND = -5           /* backend not available (only SGS_BACKEND_HIP exists) */
} sgs_status;

enum { SGS_BACKEND_CPU = 0, SGS_BACKEND_HIP = 1 };

/* sgs_config.flags */
enum {
    SGS_FLAG_ASYNC  = 1u << 0,     /* do not synchronise the stream; collect with sgs_frame_sync() */
    SGS_FLAG_TIMING = 1u << 1,     /* bracket every stage with HIP events (fills sgs_stats.ms[]) */
    SGS_FLAG_STATS  = 1u << 2,     /* also count D_f (records consumed by the composite): OFF by default — the per-pixel
                                      bookkeeping costs a sweep ~6 %; sgs_stats.d_fetched and bytes[RENDER]'s D_f term are 0 without it */
    SGS_FLAG_FULL_SORT = 1u << 3,  /* tests: order every queue completely (the production path sorts
                                      lazily and stops once a tile's pixels have all terminated) */
    SGS_FLAG_LOOSE_CULL = 1u << 5, /* tests: bin every splat over S3's reference rect (the production path bins the part of it
                                    * the alpha >= alpha_min ellipse can reach) and decide the 8x8 quadrants inside a tile from
                                    * the axis-aligned extent only (production: exact ellipse/rectangle test).  D, tile
                                    * offsets and queues then are exactly the reference's; frames must be bit-identical
                                    * either way */
    SGS_FLAG_NO_CHUNK_CULL = 1u << 6, /* tests: project every chunk of the scene (the production path first tests each
                                    * 64-Gaussian chunk's bounding sphere against the frame / the band of tile rows and skips
                                    * the chunks that cannot reach it).  N_v, D, queues and frames must not change */
    SGS_FLAG_NO_DEEP = 1u << 8,    /* tests, A/B: never cull a resident window of a long-lived tile against its live pixels before ranking it
                                    * (DESIGN.md §4.2 item 8).  Frames must not change, bit for bit */
    SGS_FLAG_NO_FINE_TILES = 1u << 9, /* tests, A/B: render this frame through 16x16-pixel tiles whatever its size.  By default a frame of at
                                    * most sgs_tuning.fine_tile_pixels pixels (640x480: the reference's own resolutions, simple_env.py:52,
                                    * run_benchmark.py:1409-1419) is rendered through 8x8-pixel tiles — four times the workgroups, a quarter of the
                                    * queue and of the per-wave splat lists in each: such a frame's time is its slowest tile's — unless its splats are
                                    * so large that the split would more than double the records (sgs_tuning.fine_tile_growth).  WHICH splats reach
                                    * a pixel does not change (S3's rect stays a rect of 16x16-pixel tiles); the tile origin the blend's
                                    * coordinates are relative to does, so the two renderings of a frame agree to fp32 rounding (both within the
                                    * parity tolerance of the oracle), not bit for bit.  SGS_FLAG_FULL_SORT / SGS_FLAG_LOOSE_CULL imply this flag:
                                    * the reference's integer structures are those of 16x16-pixel tiles.  sgs_stats.n_tiles, d_total, d_super,
                                    * max_tile_len and SGS_BUF_TILE_OFFSETS count the tiles actually used.  Version 113 */
    SGS_FLAG_PIPELINED = 1u << 4   /* with SGS_FLAG_ASYNC: the frame may run CONCURRENTLY with other pipelined frames on
                                    * the library's internal streams (a few frames in flight, each with its own
                                    * intermediates: one frame's binning fills the compute units another frame's
                                    * composite leaves idle).  It starts after the work already submitted to `stream`;
                                    * its output is complete — and ordered before later work — only after
                                    * sgs_frame_sync().  sgs_render_batch() always works this way. */
};

/* Pipeline stages, in launch order (index of sgs_stats.ms[] / .bytes[]). */
enum {
    SGS_STAGE_PREPROCESS = 0,      /* S1-S3: SH, EWA projection, AABB, compaction (k_preprocess)          */
    SGS_STAGE_COUNT      = 1,      /* S4a: per-tile counts + exclusive scan (k_bin_count, k_tile_scan)     */
    SGS_STAGE_EMIT       = 2,      /* S4b: duplication into per-tile queues (k_bin_emit)                   */
    SGS_STAGE_RENDER     = 3,      /* S5+S6 fused: lazy per-tile radix depth sort + composite (k_tile_render) */
    SGS_NUM_STAGES       = 4
};

typedef struct sgs_ctx sgs_ctx;        /* opaque */
typedef struct sgs_scene sgs_scene;    /* opaque */

/* Pinhole camera: +Z forward, +X right, +Y down; pixel i covers [i, i+1) so a point on the optical
 * axis lands at pixel coordinate cx - 0.5.  `view` maps MODEL space to camera space (row-major
 * 4x4, RIGID — rows of its 3x3 orthonormal to 1e-5, else SGS_ERR_INVALID at enqueue time: the culling bounds absorb ~1e-4 of
 * non-rigidity and the contract sits an order of magnitude inside that.  A view composed or inverted in fp32 can miss it (a few 1e-6
 * is typical, 1e-5 happens): re-orthonormalise it first, as sage_gs.renderer does for its callers): the asset's model->world transform
 * (Data/template.usda:115-124, rotateXYZ -90,0,0) is folded in by the caller.  Replaces Camera(prim_path, frequency, resolution) + focalLength
 * (simple_env.py:840-844,905; generate_images.py:344-350) and cam.set_world_pose(position,
 * orientation) (simple_env.py:1284; generate_images.py:419-421). */
typedef struct sgs_camera {
    int32_t width, height;
    float fx, fy, cx, cy;
    float view[16];
} sgs_camera;

/* Constants of stages S2-S6; sgs_config_default() fills the canonical values. */
typedef struct sgs_config {
    float near_z;          /* 0.2   cull tz <= near_z                                   */
    float far_z;           /* 1e30  cull tz >  far_z                                    */
    float dilation;        /* 0.3   px^2 added to the 2-D covariance diagonal           */
    float clamp;           /* 1.3   frustum clamp factor on t.xy / t.z                  */
    float alpha_min;       /* 1/255 */
    float alpha_max;       /* 0.99  */
    float t_min;           /* 1e-4  */
    float bg[3];           /* background, linear RGB (Data/template.usda tonemap is NOT applied) */
    int32_t sh_degree;     /* -1 = the scene's degree                                   */
    uint32_t flags;        /* SGS_FLAG_*                                                */
    /* Interleaved tile rows (multi-GPU sharding of one frame): with stride S > 1 the call owns
     * the tile rows phase, phase + S, phase + 2S, ... of the frame — every rank gets the same mix of cheap and
     * expensive rows, whatever the camera looks at.  tile_row_begin/_end then index the OWNED rows (0 .. their
     * count; end < 0 = all of them) and out_rgb is a COMPACT image: owned row k occupies pixel rows
     * [16k, 16k+16) of it.  0 or 1 = the contiguous band [tile_row_begin, tile_row_end) of the frame itself. */
    int32_t tile_row_stride;
    int32_t tile_row_phase;
} sgs_config;

typedef struct sgs_stats {
    int64_t n_gaussians;   /* N                                                          */
    int64_t n_visible;     /* N_v: survivors of culling                                  */
    int64_t d_total;       /* D: records queued (tiles reachable by each splat; S3's rect areas under LOOSE_CULL) */
    int64_t d_fetched;     /* D_f (SGS_FLAG_STATS): records consumed before every pixel of their tile stopped */
    int64_t n_pixels;      /* pixels written by this call                                */
    int32_t n_tiles;       /* tiles in [tile_row_begin, tile_row_end)                    */
    int32_t max_tile_len;  /* longest per-tile queue                                     */
    int32_t n_spill_tiles; /* depth buckets too long for LDS, sorted through HBM         */
    int32_t retries;       /* re-renders after growing the record capacity               */
    float ms[SGS_NUM_STAGES];      /* per-stage GPU time (SGS_FLAG_TIMING), else 0       */
    float ms_total;                /* first launch -> last launch (SGS_FLAG_TIMING)      */
    int64_t bytes[SGS_NUM_STAGES]; /* algorithmic bytes per stage         */
    int64_t d_super;       /* D_s: records in the super-tile queues (level 1 of the binning) */
    int64_t n_deep_windows; /* windows of tile queues that the composite culled against the tile's live pixels BEFORE ranking / staging (tiles that
                            * kept consuming batches; never with SGS_FLAG_STATS) — version 111 */
} sgs_stats;

int sgs_version(void);
/* sizeof(sgs_camera), sizeof(sgs_config), sizeof(sgs_stats) as THIS library was compiled (any pointer may be NULL): what a hand-written
 * binding compares its own struct sizes with at load time (sage_gs/_capi.py does).  The reference has no counterpart (no FFI at all). */
void sgs_struct_sizes(int32_t* camera_bytes, int32_t* config_bytes, int32_t* stats_bytes);
void sgs_config_default(sgs_config* cfg);

/* Replaces SimulationApp({...}) + World() construction (simple_env.py:160-230). */
int sgs_create(int device_id, int backend, sgs_ctx** out);
int sgs_destroy(sgs_ctx* ctx);
const char* sgs_last_error(const sgs_ctx* ctx);    /* ctx may be NULL: last creation error */

/* Record capacity (number of (Gaussian,tile) records the scratch can hold).  Grown automatically
 * by synchronous renders; asynchronous renders fail with SGS_ERR_OVERFLOW instead. */
int sgs_set_record_capacity(sgs_ctx* ctx, int64_t max_records);

/* The library's tuning surface — ALL of it: the library reads nothing from the environment (rounds 2-5 read eight SGS_* variables at
 * sgs_create; four of them — the grids of the binning and projection launches — were settled by A/B runs and are constants now).
 * sgs_tuning_default() fills what the bench runs; sgs_set_tuning() takes effect for the scenes uploaded and the frames issued AFTER it
 * (it completes the frames in flight first).  Frames do not depend on any of these, bit for bit (tests/test_gpu_parity.py).  No
 * counterpart in the reference (one synchronous SimulationApp per process, simple_env.py:163).  Version 112 (113: fine_tile_pixels, 114: fine_tile_growth — the two
 * fields frames DO depend on, to fp32 rounding). */
typedef struct sgs_tuning {
    int32_t lanes;            /* 3  frames in flight for SGS_FLAG_PIPELINED single frames: each lane has its own stream and intermediates (1..16) */
    int32_t group;            /* 8  frames per set of launches in sgs_render_batch* (blockIdx.y selects the frame; 1..8).  The full frames of a
                               *    group are projected by ONE launch that reads what they share of the scene once (r06zb); 8 x 2 against
                               *    4 x 2: -4.5 % per frame on the bench's poses (r06ze) */
    int32_t group_lanes;      /* 2  streams the groups of a batch alternate over (group x group_lanes <= 16) */
    int32_t morton;           /* 1  lay the scene out in Z-order at upload (device radix sort); 0 keeps the caller's order */
    int64_t record_capacity;  /* 16 Mi  (Gaussian, tile) records the queues of a lane hold; an overflowing frame grows them and is rendered again
                               *        (= sgs_set_record_capacity) */
    int64_t fine_tile_pixels; /* 307200 (640x480)  frames of at most this many pixels MAY be rendered through 8x8-pixel tiles, of at most a quarter
                               *        of it through 4x4 (SGS_FLAG_NO_FINE_TILES above); 0 = never.  Version 113 */
    double fine_tile_growth;  /* 2.2  ... and ARE, as long as halving the tiles multiplies the frame's (Gaussian, tile) records by no more than this
                               *        (what a split costs; estimated per frame on the host from 512 Gaussians of the scene, ~6 us).
                               *        Indoor scenes of small splats grow by 1.5-2.0 and gain 20-50 % of a frame; scenes with trained-3DGS
                               *        statistics grow by 2.5-3.5 and would lose 30 %.  >= 16 = whenever the pixel rule allows
                               *        (tests); must be >= 1.  Version 114 */
} sgs_tuning;
void sgs_tuning_default(sgs_tuning* out);
int sgs_set_tuning(sgs_ctx* ctx, const sgs_tuning* tuning);
int sgs_get_tuning(const sgs_ctx* ctx, sgs_tuning* out);

/* Scene load — replaces open_stage(usd_path) resolving /World/gauss (simple_env.py:219;
 * generate_images.py:320-327).  Inputs are fp32, activations already applied:
 *   means[N,3], scales[N,3] (linear), quats[N,4] (w,x,y,z; normalised by the library),
 *   opacities[N] in (0,1), sh[N,(sh_degree+1)^2,3].
 * on_device != 0: the pointers are device pointers on the context's device. */
int sgs_scene_upload(sgs_ctx* ctx, int64_t n, int sh_degree, const float* means, const float* scales,
                     const float* quats, const float* opacities, const float* sh, int on_device,
                     sgs_scene** out);
/* The same, from the PlayCanvas "compressed.ply" payload InteriorGS ships (`3dgs_compressed.ply`, README.md:197-231 of the reference — which
 * converts it back to a standard .ply with @playcanvas/splat-transform and then to USDZ before Isaac Sim can load it): the payload is
 * dequantised ON THE DEVICE while the scene is laid out; the fp32 arrays never exist.  16 bytes per Gaussian instead of 236.
 *   chunks[n_chunks][18]   per 256 Gaussians: min xyz, max xyz, min / max of log(scale) xyz, min rgb, max rgb (0 and 1 when the file has no
 *                          colour range: the chunk element's first 12 or 18 float properties, in file order)
 *   packed[n][4]           the vertex element's uint32 properties packed_position (11-10-11), packed_rotation (2 + 10-10-10), packed_scale
 *                          (11-10-11), packed_color (8-8-8-8), in THIS order
 *   sh[n][3 * ((d+1)^2-1)] the `sh` element's uint8 properties f_rest_*, in file order (channel-major); NULL at degree 0
 * sage_gs/ply.py (read_compressed_payload) produces exactly these from a file.  n_chunks must be ceil(n / 256). */
enum { SGS_SH_DECODE_UNSPECIFIED = 0, SGS_SH_DECODE_LINEAR255 = 1, SGS_SH_DECODE_BIN_CENTRE_ENDS = 2, SGS_SH_DECODE_BIN_CENTRE = 3 };
typedef struct sgs_compressed_scene {
    int64_t n;
    int64_t n_chunks;
    int32_t sh_degree;
    int32_t sh_decode;     /* how an 8-bit SH coefficient v becomes a float — REQUIRED when sh_degree > 0 (version 112: 0 = UNSPECIFIED is refused with
                            * SGS_ERR_INVALID instead of silently meaning "bin centre" as in version 111; ignored at degree 0):
                            *   SGS_SH_DECODE_BIN_CENTRE (3)  v / 32 - 4 + 1 / 64: the centre of the truncation bin trunc((x / 8 + 0.5) * 256) that the
                            *                                 PlayCanvas WRITER (and sage_gs.ply.encode_compressed) puts x into;
                            *   SGS_SH_DECODE_LINEAR255  (1)  v * 8 / 255 - 4: 0 -> -4, 255 -> +4, linear in between — to this builder's recollection what
                            *                                 the PlayCanvas READERS (engine GSplatCompressedData, splat-transform's decompress) apply;
                            *   SGS_SH_DECODE_BIN_CENTRE_ENDS (2)  bin centres, but v = 0 -> -4 and v = 255 -> +4 exactly.
                            * The three differ by at most 1 / 64 per coefficient.  The reference never decodes these bytes itself (README.md:197-231 hands
                            * the file to @playcanvas/splat-transform, un-vendored, unpinned and not installable here), so which of them that tool
                            * applies cannot be pinned from this container — which is why there is no default: say which one your converter uses
                            * (tests/golden/compressed_ply_kat.json lists every byte's value under all three). */
    const float* chunks;
    const uint32_t* packed;
    const uint8_t* sh;
} sgs_compressed_scene;
int sgs_scene_upload_compressed(sgs_ctx* ctx, const sgs_compressed_scene* z, int on_device, sgs_scene** out);
int sgs_scene_free(sgs_ctx* ctx, sgs_scene* scene);

/* One frame — replaces world.step(render=True) x2..5 + cam.get_rgba() (simple_env.py:1368-1380;
 * generate_images.py:425-428).  Renders the tile rows [tile_row_begin, tile_row_end) of the frame
 * (tile_row_end < 0: all rows) into out_rgb, a DEVICE buffer of height*width*3 floats (row-major,
 * RGB interleaved); rows outside the range are left untouched.  stats may be NULL. */
int sgs_render(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cam, const sgs_config* cfg,
               int tile_row_begin, int tile_row_end, float* out_rgb, sgs_stats* stats,
               void* hip_stream);

/* As sgs_render, plus out_aux: a DEVICE buffer of height*width*2 floats — per pixel the expected view
 * depth sum_i T_i alpha_i z_i (metres along the optical axis) and the coverage 1 - T_final.  This is the
 * natural Gaussian-scene counterpart of the reference's depth channel, which renders the depth of the
 * collision mesh instead (simple_env.py:1395-1589: get_depth, clipped to [0.1, 6.5] m). */
int sgs_render_rgbd(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cam, const sgs_config* cfg,
                    int tile_row_begin, int tile_row_end, float* out_rgb, float* out_aux,
                    sgs_stats* stats, void* hip_stream);

/* B frames of one scene, back to back, one synchronisation at the end — the frame loop of
 * generate_images.py:408-436.  out_rgb holds B consecutive frames; stats (nullable) B entries.  (With
 * cfg->tile_row_stride > 1 every frame still has a height*width*3 slot; its compact image starts at the slot.)
 * The frames are issued in groups of up to sgs_tuning.group CONSECUTIVE cameras; the frames of a group are projected by one launch
 * that reads what they both see of the scene once — pass a trajectory's cameras in path order (neighbours share nearly everything:
 * 7-10 % per frame; unrelated views: 2-5 %).  Every frame equals the frame sgs_render() renders, bit for bit. */
int sgs_render_batch(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cams, int n_cams,
                     const sgs_config* cfg, int tile_row_begin, int tile_row_end, float* out_rgb,
                     sgs_stats* stats, void* hip_stream);

/* As sgs_render_batch, with the frames' outputs `frame_stride` FLOATS apart: frame i is stored as if out_rgb + i *
 * frame_stride were the address of its pixel (0,0).  This is what a rank of a tile-row-sharded sweep uses — every
 * frame's band of rows goes to its own slab (sage_gs/dist.py) — and what keeps the host out of the way there: the whole
 * batch is one call, and the library forks its internal streams, clears and collects the frames' status words and waits
 * for completion once per batch instead of once per frame. */
int sgs_render_batch_strided(sgs_ctx* ctx, const sgs_scene* scene, const sgs_camera* cams, int n_cams,
                             const sgs_config* cfg, int tile_row_begin, int tile_row_end, float* out_rgb,
                             int64_t frame_stride, sgs_stats* stats, void* hip_stream);

/* Completes frames issued with SGS_FLAG_ASYNC: waits for the stream of the MOST RECENT frame and for every
 * pipelined frame in flight, checks the status of every frame issued since the previous synchronisation and
 * reports the statistics of the most recent one.  A caller that spreads asynchronous, non-pipelined frames over
 * several streams of its own must synchronise the earlier streams itself (one stream per context is the
 * intended use; pipelined frames are the library's way to overlap frames). */
int sgs_frame_sync(sgs_ctx* ctx, sgs_stats* stats);

/* Records queued per FRAME tile row (out[r], r < n_rows <= 4096), summed over every frame rendered since the last call with
 * reset != 0 — each frame adds the rows it rendered (its band).  This is the per-row cost from which cost-balanced
 * tile-row bands are cut (SURVEY.md §8e "optional cost-balanced ranges from the previous frame's per-row D"); the
 * reference has no counterpart (it shards by scene only, generate_images.py:136-139).  Covers the frames that have
 * been completed (synchronous frames, or asynchronous ones after sgs_frame_sync); frames still in flight may be
 * counted partly. */
int sgs_row_records(sgs_ctx* ctx, int64_t* out, int n_rows, int reset);

/* fp32 RGB -> uint8 RGBA (alpha 255), the shape cam.get_rgba() returns (simple_env.py:1380-1386;
 * generate_images.py:428-431).  `rgb` is a device buffer; `rgba` is a device buffer or PINNED host memory the device can address
 * (hipHostMalloc / a torch pin_memory tensor: the kernel then writes the host buffer itself, over the link — for a single frame a
 * caller is waiting for that is 25-30 us shorter than a device buffer plus a copy, sage_gs.Renderer.render_rgba8_host). */
int sgs_pack_rgba8(sgs_ctx* ctx, const float* rgb, uint8_t* rgba, int width, int height,
                   void* hip_stream);

/* Test hook: copy an intermediate buffer of the LAST synchronous frame to host memory.
 * Returns the number of bytes the buffer holds (copying at most `bytes`), or a negative status. */
enum {
    SGS_BUF_TILE_OFFSETS = 0,      /* uint32[T+1]                                                */
    SGS_BUF_SORTED_SLOTS = 1,      /* uint32[D]    per-tile queues in (depth, index) order — complete only with SGS_FLAG_FULL_SORT */
    SGS_BUF_SLOT_IDS     = 2,      /* uint32[S]    slot i holds Gaussian i: i if live this frame, 0xFFFFFFFF if culled; S = ceil(N/64)*64 */
    SGS_BUF_CHUNK_SKIPPED = 4,     /* uint8[ceil(N/64)]  1 = the 64-Gaussian chunk (in layout order) was skipped by its bounds */
    SGS_BUF_SPLATS       = 3,      /* S x 12 words: x,y,conic a,b | c,opacity,r,g | b,depth bits,rect01,rect23 (dead = 0) */
    SGS_BUF_SCENE_GEOM   = 5,      /* float[N][11] of the LAST RENDERED scene as the device holds it, by original index: mean xyz, opacity, scale xyz,
                                    * quaternion wxyz (as uploaded / as dequantised from a compressed payload) */
    SGS_BUF_SCENE_SH     = 6       /* float[N][3 (d+1)^2] of the LAST RENDERED scene, by original index, [coefficient][channel]: the SH coefficients the
                                    * projection kernel evaluates — the fp32 rows as uploaded, or (a scene uploaded from the compressed payload, which
                                    * keeps its 8-bit coefficients as bytes in HBM) those bytes dequantised exactly as the kernel does */
};
int64_t sgs_debug_read(sgs_ctx* ctx, int what, void* host_dst, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* SAGE_GS_H */
