/* sgs_oracle.c — plain-C restatement of the 3DGS scene-render hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The reference (Galery23/SAGE-3D_Official) has no rasterizer source: frames come
 * from Isaac Sim at Code/benchmark/environment_evaluation/simple_env.py:1368-1380 and
 * Code/data_pipeline/training_data_construction/generate_images.py:425-428 (closed binary, absent),
 * and the repository holds no tests or golden frames.  The six stages below
 * restate BASELINE.json:north_star with the constants of SURVEY.md §8(a) rows S1-S6, written from
 * the public EWA-splatting equations.  Pinned by closed-form known-answer tests and by agreement
 * with the independent NumPy restatement oracle/oracle_np.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It is also the timed "CPU rasterizer" of BASELINE.md (kind "port"): build with -DORC_REAL=float
 * for the fp32 timing build, default double for the checker build.
 *
 * Stage map:  S1 eval_sh()  S2/S3 project_one()  S4 bin (count/scan/fill)
 *                             S5 per-tile sort by (fp32 depth bits, index)  S6 composite_tile().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef ORC_REAL
#define ORC_REAL double
#endif
typedef ORC_REAL real;
#define TILE 16

typedef struct {
    int32_t width, height;
    float fx, fy, cx, cy;
    float view[16];              /* model -> camera, row-major */
} orc_camera;

typedef struct {
    float near_z, far_z, dilation, clamp, alpha_min, alpha_max, t_min;
    float bg[3];
    int32_t sh_degree;           /* -1: scene degree */
} orc_config;

typedef struct {
    int64_t N, n_visible, D, D_f;
    int32_t gx, gy, row_begin, row_end;
    /* per Gaussian */
    uint32_t* depth_bits; int32_t* rect; int32_t* tiles;
    float* xy; float* conic; float* opacity; float* rgb;
    /* per tile */
    int64_t* offsets; int32_t* ids; int64_t* consumed;
    /* per pixel */
    float* image; float* final_T; int32_t* n_contrib;
    int32_t width, height;
    float* depth_img;            /* per pixel: sum T alpha z */
    float* margin;               /* per pixel: min |alpha/alpha_min - 1| over examined pairs (checker build) */
} orc_frame;

static const double C0 = 0.28209479177387814, C1 = 0.4886025119029199;
static const double C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                             -1.0925484305920792, 0.5462742152960396};
static const double C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                             0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                             -0.5900435899266435};

/* S1: colour of one Gaussian seen along unit direction (x,y,z); sh is [K][3]. */
static void eval_sh(const float* sh, int deg, real x, real y, real z, real out[3]) {
    for (int ch = 0; ch < 3; ++ch) {
#define S(k) ((real)sh[(k) * 3 + ch])
        real r = (real)C0 * S(0);
        if (deg >= 1) r = r - (real)C1 * y * S(1) + (real)C1 * z * S(2) - (real)C1 * x * S(3);
        if (deg >= 2) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            r = r + (real)C2[0] * xy * S(4) + (real)C2[1] * yz * S(5)
                  + (real)C2[2] * (2 * zz - xx - yy) * S(6)
                  + (real)C2[3] * xz * S(7) + (real)C2[4] * (xx - yy) * S(8);
            if (deg >= 3)
                r = r + (real)C3[0] * y * (3 * xx - yy) * S(9) + (real)C3[1] * xy * z * S(10)
                      + (real)C3[2] * y * (4 * zz - xx - yy) * S(11)
                      + (real)C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * S(12)
                      + (real)C3[4] * x * (4 * zz - xx - yy) * S(13)
                      + (real)C3[5] * z * (xx - yy) * S(14) + (real)C3[6] * x * (xx - 3 * yy) * S(15);
        }
#undef S
        r += (real)0.5;
        out[ch] = r < 0 ? 0 : r;
    }
}

static inline int clampi(double v, int lo, int hi) {
    double f = floor(v);
    if (!(f > lo)) return lo;       /* also catches NaN */
    if (f > hi) return hi;
    return (int)f;
}

/* S2 + S3 for Gaussian i.  Geometry is always evaluated in double (precision contract). */
static void project_one(orc_frame* f, int64_t i, int K, int deg, const float* means,
                        const float* scales, const float* quats, const float* opac, const float* sh,
                        const orc_camera* cam, const orc_config* cfg, const double campos[3]) {
    const float* V = cam->view;
    double m[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    double t[3];
    for (int r = 0; r < 3; ++r)
        t[r] = (double)V[4 * r] * m[0] + (double)V[4 * r + 1] * m[1] + (double)V[4 * r + 2] * m[2] + (double)V[4 * r + 3];
    float dz = (float)t[2];
    memcpy(&f->depth_bits[i], &dz, 4);
    f->tiles[i] = 0;
    int32_t* rc = &f->rect[4 * i];
    rc[0] = rc[1] = rc[2] = rc[3] = 0;
    if (!(t[2] > (double)cfg->near_z) || t[2] > (double)cfg->far_z) return;

    double q[4] = {quats[4 * i], quats[4 * i + 1], quats[4 * i + 2], quats[4 * i + 3]};
    double qn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double w = q[0] / qn, x = q[1] / qn, y = q[2] / qn, z = q[3] / qn;
    double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                      {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                      {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
    double M[3][3], Sg[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r][c] = R[r][c] * (double)scales[3 * i + c];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
        Sg[r][c] = M[r][0] * M[c][0] + M[r][1] * M[c][1] + M[r][2] * M[c][2];

    double tanx = 0.5 * cam->width / (double)cam->fx, tany = 0.5 * cam->height / (double)cam->fy;
    double limx = (double)cfg->clamp * tanx, limy = (double)cfg->clamp * tany;
    double tz = t[2];
    double txc = fmin(limx, fmax(-limx, t[0] / tz)) * tz;
    double tyc = fmin(limy, fmax(-limy, t[1] / tz)) * tz;
    double J[2][3] = {{(double)cam->fx / tz, 0, -(double)cam->fx * txc / (tz * tz)},
                      {0, (double)cam->fy / tz, -(double)cam->fy * tyc / (tz * tz)}};
    double T[2][3];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c)
        T[r][c] = J[r][0] * V[c] + J[r][1] * V[4 + c] + J[r][2] * V[8 + c];
    double TS[2][3];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c)
        TS[r][c] = T[r][0] * Sg[0][c] + T[r][1] * Sg[1][c] + T[r][2] * Sg[2][c];
    double a = TS[0][0] * T[0][0] + TS[0][1] * T[0][1] + TS[0][2] * T[0][2] + (double)cfg->dilation;
    double b = TS[0][0] * T[1][0] + TS[0][1] * T[1][1] + TS[0][2] * T[1][2];
    double c = TS[1][0] * T[1][0] + TS[1][1] * T[1][1] + TS[1][2] * T[1][2] + (double)cfg->dilation;
    double det = a * c - b * b;
    if (!(det > 0.0)) return;

    double mid = 0.5 * (a + c);
    double lam = mid + sqrt(fmax(0.1, mid * mid - det));
    double radius = ceil(3.0 * sqrt(lam));
    double px = (double)cam->fx * t[0] / tz + (double)cam->cx - 0.5;
    double py = (double)cam->fy * t[1] / tz + (double)cam->cy - 0.5;
    int x0 = clampi((px - radius) / TILE, 0, f->gx), x1 = clampi((px + radius + (TILE - 1)) / TILE, 0, f->gx);
    int y0 = clampi((py - radius) / TILE, f->row_begin, f->row_end);
    int y1 = clampi((py + radius + (TILE - 1)) / TILE, f->row_begin, f->row_end);
    int nt = (x1 - x0) * (y1 - y0);
    if (nt <= 0) return;
    rc[0] = x0; rc[1] = y0; rc[2] = x1; rc[3] = y1;
    f->tiles[i] = nt;
    f->xy[2 * i] = (float)px; f->xy[2 * i + 1] = (float)py;
    f->conic[3 * i] = (float)(c / det); f->conic[3 * i + 1] = (float)(-b / det); f->conic[3 * i + 2] = (float)(a / det);
    f->opacity[i] = opac[i];

    double d[3] = {m[0] - campos[0], m[1] - campos[1], m[2] - campos[2]};
    double dn = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    real col[3];
    eval_sh(sh + (size_t)i * K * 3, deg, (real)(d[0] / dn), (real)(d[1] / dn), (real)(d[2] / dn), col);
    f->rgb[3 * i] = (float)col[0]; f->rgb[3 * i + 1] = (float)col[1]; f->rgb[3 * i + 2] = (float)col[2];
}

typedef struct { uint32_t key; int32_t id; } rec_t;
static int rec_cmp(const void* pa, const void* pb) {
    const rec_t* a = (const rec_t*)pa; const rec_t* b = (const rec_t*)pb;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    return (a->id > b->id) - (a->id < b->id);
}

/* S6 for one tile. */
static void composite_tile(orc_frame* f, int tx, int ty, const orc_config* cfg) {
    int t = ty * f->gx + tx;
    const int32_t* q = f->ids + f->offsets[t];
    int64_t n = f->offsets[t + 1] - f->offsets[t];
    int64_t used_max = 0;
    for (int py = ty * TILE; py < (ty + 1) * TILE && py < f->height; ++py)
        for (int px = tx * TILE; px < (tx + 1) * TILE && px < f->width; ++px) {
            real T = 1, C[3] = {0, 0, 0}, Z = 0;
            int32_t nc = 0; int64_t k;
            double mg = 1e30;
            for (k = 0; k < n; ++k) {
                int32_t g = q[k];
                real dx = (real)f->xy[2 * g] - (real)px, dy = (real)f->xy[2 * g + 1] - (real)py;
                real power = (real)-0.5 * ((real)f->conic[3 * g] * dx * dx + (real)f->conic[3 * g + 2] * dy * dy)
                             - (real)f->conic[3 * g + 1] * dx * dy;
                if (power > 0) continue;
                real e = sizeof(real) == 4 ? (real)expf((float)power) : (real)exp((double)power);
                real alpha = (real)f->opacity[g] * e;
                if (alpha > (real)cfg->alpha_max) alpha = (real)cfg->alpha_max;
#ifdef ORC_MARGIN
                { double m1 = fabs((double)alpha / (double)cfg->alpha_min - 1.0); if (m1 < mg) mg = m1; }
#endif
                if (alpha < (real)cfg->alpha_min) continue;
                real testT = T * (1 - alpha);
#ifdef ORC_MARGIN
                { double m2 = fabs((double)testT / (double)cfg->t_min - 1.0); if (m2 < mg) mg = m2; }
#endif
                if (testT < (real)cfg->t_min) { ++k; break; }
                real wgt = alpha * T;
                C[0] += wgt * (real)f->rgb[3 * g]; C[1] += wgt * (real)f->rgb[3 * g + 1]; C[2] += wgt * (real)f->rgb[3 * g + 2];
                { float zf; memcpy(&zf, &f->depth_bits[g], 4); Z += wgt * (real)zf; }
                T = testT; nc = (int32_t)(k + 1);
            }
            if (k > used_max) used_max = k;
            size_t p = (size_t)py * f->width + px;
            f->image[3 * p] = (float)(C[0] + T * (real)cfg->bg[0]);
            f->image[3 * p + 1] = (float)(C[1] + T * (real)cfg->bg[1]);
            f->image[3 * p + 2] = (float)(C[2] + T * (real)cfg->bg[2]);
            f->final_T[p] = (float)T; f->n_contrib[p] = nc; f->margin[p] = (float)mg; f->depth_img[p] = (float)Z;
        }
    f->consumed[t] = used_max;
}

/* Two-sided evaluation of ONE pixel (the checker's answer to S6's two discontinuities).  A (pixel, Gaussian) pair whose
 * alpha lies within rel_margin (relative) of alpha_min may legitimately be taken or skipped by a differently rounded
 * evaluation, and a pair whose T(1-alpha) lies within rel_margin of t_min may legitimately end the pixel or be blended.
 * This walks the pixel's queue and BRANCHES at every such decision (depth-first, at most `cap` branch points on a
 * path; beyond that the nominal decision is taken and *capped is set), evaluates the colour of every leaf and returns
 * in *best_err the smallest max-over-channels |leaf - got| — i.e. `got` is accepted iff it matches the result of
 * SOME admissible set of decisions within the tolerance the caller applies.  Returns the number of leaves (1 = the
 * pixel has no decision inside the margin).  The frame must still hold its per-Gaussian arrays and queues. */
typedef struct { const orc_frame* f; const orc_config* cfg; const int32_t* q; int64_t n; int px, py; double rm; int cap;
                 const float* got; double best; int64_t leaves; int capped; } orc_var;

static void var_leaf(orc_var* v, real T, const real C[3]) {
    double e = 0;
    for (int c = 0; c < 3; ++c) {
        double d = fabs((double)(float)(C[c] + T * (real)v->cfg->bg[c]) - (double)v->got[c]);
        if (isnan(d)) { v->leaves++; return; }                   /* a NaN channel matches nothing: this leaf cannot be the best */
        if (d > e) e = d;
    }
    if (e < v->best) v->best = e;
    v->leaves++;
}

static void var_walk(orc_var* v, int64_t k, real T, real C0, real C1, real C2, int depth) {
    const orc_frame* f = v->f; const orc_config* cfg = v->cfg;
    real C[3] = {C0, C1, C2};
    for (; k < v->n; ++k) {
        int32_t g = v->q[k];
        real dx = (real)f->xy[2 * g] - (real)v->px, dy = (real)f->xy[2 * g + 1] - (real)v->py;
        real power = (real)-0.5 * ((real)f->conic[3 * g] * dx * dx + (real)f->conic[3 * g + 2] * dy * dy)
                     - (real)f->conic[3 * g + 1] * dx * dy;
        if (power > 0) continue;
        real alpha = (real)f->opacity[g] * (real)exp((double)power);
        if (alpha > (real)cfg->alpha_max) alpha = (real)cfg->alpha_max;
        int hit = alpha >= (real)cfg->alpha_min;
        if (fabs((double)alpha / (double)cfg->alpha_min - 1.0) < v->rm) {
            if (depth < v->cap) {
                ++depth;
                var_walk(v, k + 1, T, C[0], C[1], C[2], depth);  /* this pair skipped ... */
                hit = 1;                                         /* ... or taken (continues below) */
            } else v->capped = 1;
        }
        if (!hit) continue;
        real testT = T * (1 - alpha);
        int stop = testT < (real)cfg->t_min;
        if (fabs((double)testT / (double)cfg->t_min - 1.0) < v->rm) {
            if (depth < v->cap) {
                ++depth;
                var_leaf(v, T, C);                               /* the pixel ends here ... */
                stop = 0;                                        /* ... or this pair is blended */
            } else v->capped = 1;
        }
        if (stop) break;
        real wgt = alpha * T;
        C[0] += wgt * (real)f->rgb[3 * g]; C[1] += wgt * (real)f->rgb[3 * g + 1]; C[2] += wgt * (real)f->rgb[3 * g + 2];
        T = testT;
    }
    var_leaf(v, T, C);
}

int64_t orc_pixel_variants(const orc_frame* f, const orc_config* cfg, int px, int py, double rel_margin, int cap,
                           const float got[3], double* best_err, int* capped) {
    if (px < 0 || py < 0 || px >= f->width || py >= f->height) return -1;
    int t = (py / TILE) * f->gx + px / TILE;
    orc_var v = {f, cfg, f->ids + f->offsets[t], f->offsets[t + 1] - f->offsets[t], px, py, rel_margin, cap, got, 1e300, 0, 0};
    var_walk(&v, 0, 1, 0, 0, 0, 0);
    if (best_err) *best_err = v.best;
    if (capped) *capped = v.capped;
    return v.leaves;
}

void orc_frame_free(orc_frame* f) {
    if (!f) return;
    free(f->depth_bits); free(f->rect); free(f->tiles); free(f->xy); free(f->conic); free(f->opacity);
    free(f->rgb); free(f->offsets); free(f->ids); free(f->consumed); free(f->image); free(f->final_T);
    free(f->n_contrib); free(f->margin); free(f->depth_img); free(f);
}

int orc_real_bytes(void) { return (int)sizeof(real); }
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Runs all six stages.  threads <= 0: all host cores.  Returns NULL on allocation failure. */
orc_frame* orc_render(int64_t N, int sh_degree, const float* means, const float* scales,
                      const float* quats, const float* opac, const float* sh, const orc_camera* cam,
                      const orc_config* cfg, int row_begin, int row_end, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    orc_frame* f = (orc_frame*)calloc(1, sizeof(orc_frame));
    if (!f) return NULL;
    f->N = N; f->width = cam->width; f->height = cam->height;
    f->gx = (cam->width + TILE - 1) / TILE; f->gy = (cam->height + TILE - 1) / TILE;
    if (row_end < 0 || row_end > f->gy) row_end = f->gy;
    if (row_begin < 0) row_begin = 0;
    f->row_begin = row_begin; f->row_end = row_end;
    int64_t ntile = (int64_t)f->gx * f->gy; size_t P = (size_t)cam->width * cam->height;
    size_t n1 = (size_t)(N > 0 ? N : 1);
    f->depth_bits = calloc(n1, 4); f->rect = calloc(n1 * 4, 4); f->tiles = calloc(n1, 4);
    f->xy = calloc(n1 * 2, 4); f->conic = calloc(n1 * 3, 4); f->opacity = calloc(n1, 4); f->rgb = calloc(n1 * 3, 4);
    f->offsets = calloc(ntile + 1, 8); f->consumed = calloc(ntile, 8);
    f->image = calloc(P * 3 + 1, 4); f->final_T = calloc(P + 1, 4); f->n_contrib = calloc(P + 1, 4); f->margin = calloc(P + 1, 4); f->depth_img = calloc(P + 1, 4);
    int K = (sh_degree + 1) * (sh_degree + 1);
    int deg = cfg->sh_degree < 0 ? sh_degree : (cfg->sh_degree < sh_degree ? cfg->sh_degree : sh_degree);
    const float* V = cam->view;
    double campos[3];
    for (int c = 0; c < 3; ++c)
        campos[c] = -((double)V[c] * V[3] + (double)V[4 + c] * V[7] + (double)V[8 + c] * V[11]);

    /* S1-S3 */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i)
        project_one(f, i, K, deg, means, scales, quats, opac, sh, cam, cfg, campos);

    /* S4: count -> scan -> fill (slots handed out atomically; S5 makes the order deterministic) */
    int64_t* cnt = calloc(ntile + 1, 8);
    int64_t nvis = 0;
#pragma omp parallel for schedule(static) reduction(+ : nvis)
    for (int64_t i = 0; i < N; ++i) {
        if (!f->tiles[i]) continue;
        ++nvis;
        const int32_t* rc = &f->rect[4 * i];
        for (int y = rc[1]; y < rc[3]; ++y) for (int x = rc[0]; x < rc[2]; ++x) {
#pragma omp atomic
            cnt[(int64_t)y * f->gx + x] += 1;
        }
    }
    f->n_visible = nvis;
    for (int64_t t = 0; t < ntile; ++t) f->offsets[t + 1] = f->offsets[t] + cnt[t];
    f->D = f->offsets[ntile];
    rec_t* recs = malloc(sizeof(rec_t) * (size_t)(f->D > 0 ? f->D : 1));
    f->ids = malloc(4 * (size_t)(f->D > 0 ? f->D : 1));
    memset(cnt, 0, 8 * (size_t)(ntile + 1));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        if (!f->tiles[i]) continue;
        const int32_t* rc = &f->rect[4 * i];
        for (int y = rc[1]; y < rc[3]; ++y) for (int x = rc[0]; x < rc[2]; ++x) {
            int64_t t = (int64_t)y * f->gx + x, s;
#pragma omp atomic capture
            s = cnt[t]++;
            recs[f->offsets[t] + s].key = f->depth_bits[i];
            recs[f->offsets[t] + s].id = (int32_t)i;
        }
    }
    free(cnt);

    /* S5: per-tile sort by (depth bits, index) */
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t t = 0; t < ntile; ++t) {
        int64_t n = f->offsets[t + 1] - f->offsets[t];
        if (n > 1) qsort(recs + f->offsets[t], (size_t)n, sizeof(rec_t), rec_cmp);
        for (int64_t k = 0; k < n; ++k) f->ids[f->offsets[t] + k] = recs[f->offsets[t] + k].id;
    }
    free(recs);

    /* S6 */
    for (size_t p = 0; p < P; ++p) f->final_T[p] = 1.0f;
    int64_t nt_rows = (int64_t)(row_end - row_begin) * f->gx;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t k = 0; k < nt_rows; ++k)
        composite_tile(f, (int)(k % f->gx), row_begin + (int)(k / f->gx), cfg);
    int64_t df = 0;
    for (int64_t t = 0; t < ntile; ++t) df += f->consumed[t];
    f->D_f = df;
    return f;
}
