/*
 * lfm_oracle.c -- CPU restatement of LightFM's `_lightfm_fast` native engine.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (lightfm_amd/) links,
 * imports or executes this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks every
 * function here bit-for-bit against the reference's own shipped Cython output
 * compiled by oracle/Makefile (`make ref` -> oracle/_ref/strict), and
 * tests/golden/ holds fixtures produced by that reference build
 * (tests/golden/make_golden.py) for machines where /root/reference is absent.
 *
 * Every function cites the reference lines it restates:
 *   PYX   = /root/reference/lightfm/_lightfm_fast.pyx.template
 *   C_OMP = /root/reference/lightfm/_lightfm_fast_openmp.c (type promotions)
 *
 * Numerics recipe (verified against C_OMP): representations and dot products
 * are float32 multiply then float32 add, left to right, no FMA contraction
 * (build with -ffp-contract=off); every optimizer cell update is evaluated in
 * float64 from float32-loaded operands and rounded to float32 at each store.
 *
 * Extensions over the reference (needed to check a parallel GPU kernel, all
 * opt-in through orc_opts):
 *   - rng_mode 1: one PRNG stream per shuffled position (counter-based seed),
 *     so each interaction's draws do not depend on processing order;
 *   - dot_mode 1: the wavefront "butterfly" summation order of the HIP fast
 *     path (documented deviation from the reference's sequential sum);
 *   - per-position logs of (negative item, sampled count) and totals of
 *     draws / updates / in_positives probes, which the reference never exposes.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef float flt; /* PYX:12 */

typedef struct {
    const int32_t *indices; /* PYX:151 */
    const int32_t *indptr;  /* PYX:152 */
    const flt *data;        /* PYX:153 */
    int32_t rows, cols;     /* PYX:155-156 */
    int64_t nnz;
} orc_csr;

typedef struct { /* PYX:185-213, same order as FastLightFM.__init__ */
    flt *item_W, *item_G, *item_M, *item_b, *item_bG, *item_bM;
    flt *user_W, *user_G, *user_M, *user_b, *user_bG, *user_bM;
    int32_t n_item_feat, n_user_feat;
    int32_t d;        /* no_components */
    int32_t adadelta; /* PYX:208 */
    flt lr, rho, eps; /* stored as float32, PYX:209-211 */
    int32_t max_sampled;
    double item_scale, user_scale; /* PYX:214-215 */
} orc_model;

typedef struct {
    int32_t rng_mode;     /* 0 reference per-thread streams, 1 per-position streams */
    int32_t dot_mode;     /* 0 reference sequential sum, 1 wave64 butterfly order */
    int32_t *neg_log;     /* [n] chosen negative per shuffled position or -1 */
    int32_t *sampled_log; /* [n] draws consumed per shuffled position (0 = skipped) */
    int64_t counters[4];  /* positives visited, draws, updates, in_positives probes */
} orc_opts;

/* ---------------------------------------------------------------- PRNG --- */

/* PYX:64-76 */
static uint32_t temper(uint32_t x)
{
    x ^= x >> 11;
    x ^= (x << 7) & 0x9D2C5680u;
    x ^= (x << 15) & 0xEFC60000u;
    x ^= x >> 18;
    return x;
}

/* PYX:79-81; C_OMP:2653-2692: the division is an integer division of the
 * tempered unsigned value, so the result is a non-negative 31-bit int. */
int32_t orc_rand_r(uint32_t *seed)
{
    *seed = *seed * 1103515245u + 12345u;
    return (int32_t)(temper(*seed) / 2u);
}

/* PYX:84-90 */
static int32_t sample_range(int32_t lo, int32_t hi, uint32_t *seed)
{
    return lo + orc_rand_r(seed) % (hi - lo);
}

/* Extension (rng_mode 1): seed of the stream owned by shuffled position i.
 * murmur3 finaliser over a Weyl sequence; the HIP kernels use the same rule. */
uint32_t orc_position_seed(uint32_t base, uint64_t i)
{
    uint32_t h = base + (uint32_t)i * 0x9E3779B9u + (uint32_t)(i >> 32) * 0x85EBCA6Bu;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

/* libgomp static schedule without a chunk clause (C_OMP:7224 `#pragma omp for`):
 * contiguous blocks, the first n%T threads get one extra iteration. */
static void static_chunk(int64_t n, int32_t T, int32_t t, int64_t *lo, int64_t *hi)
{
    int64_t q = n / T, r = n % T;
    if (t < r) { *lo = (q + 1) * t; *hi = *lo + q + 1; }
    else { *lo = q * t + r; *hi = *lo + q; }
}

/* ------------------------------------------------------------- helpers --- */

/* PYX:262-267; C_OMP:4669-4680: exp in double, result rounded to float32 */
static flt sigmoid(flt v) { return (flt)(1.0 / (1.0 + exp(-(double)v))); }

/* PYX:270-284 (libc bsearch over the sorted row; any correct search agrees) */
int32_t orc_in_positives(int32_t item_id, int32_t user_id, const orc_csr *m)
{
    int32_t lo = m->indptr[user_id], hi = m->indptr[user_id + 1];
    while (lo < hi) {
        int32_t mid = lo + (hi - lo) / 2;
        int32_t v = m->indices[mid];
        if (v == item_id) return 1;
        if (v < item_id) lo = mid + 1; else hi = mid;
    }
    return 0;
}

/* PYX:287-317; C_OMP:4896: w = (float)((double)data * scale) */
static void compute_representation(const orc_csr *f, const flt *W, const flt *b, int32_t d,
                                   int32_t row, double scale, flt *rep)
{
    for (int32_t j = 0; j <= d; j++) rep[j] = 0.0f;
    for (int32_t i = f->indptr[row]; i < f->indptr[row + 1]; i++) {
        int32_t feat = f->indices[i];
        flt w = (flt)((double)f->data[i] * scale);
        for (int32_t j = 0; j < d; j++) {
            flt p = w * W[(int64_t)feat * d + j];
            rep[j] = rep[j] + p;
        }
        flt pb = w * b[feat];
        rep[d] = rep[d] + pb;
    }
}

/* PYX:320-334; C_OMP:4954-5020: sequential float32 sum starting from the biases.
 * dot_mode 1 (extension): per-lane products summed by a 64-lane xor butterfly
 * (offsets 32,16,8,4,2,1), components c >= 64 folded lane-wise first (c % 64),
 * then (bias_u + bias_i) + tree.  Mirrors lightfm_amd/csrc fast path. */
static flt compute_prediction(const flt *u, const flt *v, int32_t d, int32_t dot_mode)
{
    if (dot_mode == 0) {
        flt r = u[d] + v[d];
        for (int32_t i = 0; i < d; i++) {
            flt p = u[i] * v[i];
            r = r + p;
        }
        return r;
    }
    flt lane[64];
    for (int32_t l = 0; l < 64; l++) {
        flt acc = 0.0f;
        for (int32_t c = l; c < d; c += 64) {
            flt p = u[c] * v[c];
            acc = acc + p;
        }
        lane[l] = acc;
    }
    for (int32_t off = 32; off >= 1; off >>= 1) {
        flt nxt[64];
        for (int32_t l = 0; l < 64; l++) nxt[l] = lane[l] + lane[l ^ off];
        memcpy(lane, nxt, sizeof(lane));
    }
    flt bias = u[d] + v[d];
    return bias + lane[0];
}

/* One optimizer cell.  PYX:358-395 / 416-455; promotions from C_OMP:5088-5250,
 * 5340-5560.  Returns the local learning rate. */
static double update_cell(flt *W, flt *G, flt *M, double w, double g, const orc_model *m,
                          double alpha)
{
    double lr;
    if (m->adadelta) {
        flt rg = m->rho * *G; /* float32 * float32 product (C_OMP:5382) */
        *G = (flt)((double)rg + (1.0 - (double)m->rho) * ((w * g) * (w * g)));
        flt me = *M + m->eps; /* float32 adds (C_OMP:5396) */
        flt ge = *G + m->eps;
        lr = sqrt((double)me) / sqrt((double)ge);
        double upd = (lr * g) * w;
        flt rm = m->rho * *M;
        *M = (flt)((double)rm + (1.0 - (double)m->rho) * (upd * upd));
        *W = (flt)((double)*W - upd);
    } else {
        lr = (double)m->lr / sqrt((double)*G);
        *W = (flt)((double)*W - (lr * w) * g);
        *G = (flt)((double)*G + (g * w) * (g * w));
    }
    *W = (flt)((double)*W * (1.0 + alpha * lr));
    return lr;
}

/* PYX:337-391 */
static double update_biases(const orc_csr *f, int32_t start, int32_t stop, flt *b, flt *bG,
                            flt *bM, double g, const orc_model *m, double alpha)
{
    double sum = 0.0;
    for (int32_t i = start; i < stop; i++) {
        int32_t feat = f->indices[i];
        sum += update_cell(&b[feat], &bG[feat], &bM[feat], (double)f->data[i], g, m, alpha);
    }
    return sum;
}

/* PYX:394-451 */
static double update_features(const orc_csr *f, flt *W, flt *G, flt *M, int32_t d, int32_t comp,
                              int32_t start, int32_t stop, double g, const orc_model *m,
                              double alpha)
{
    double sum = 0.0;
    for (int32_t i = start; i < stop; i++) {
        int64_t o = (int64_t)f->indices[i] * d + comp;
        sum += update_cell(&W[o], &G[o], &M[o], (double)f->data[i], g, m, alpha);
    }
    return sum;
}

/* PYX:454-534 (logistic) */
static void update(double loss, const orc_csr *itf, const orc_csr *usf, int32_t user, int32_t item,
                   const flt *urep, const flt *irep, orc_model *m, double ia, double ua)
{
    int32_t is = itf->indptr[item], ie = itf->indptr[item + 1];
    int32_t us = usf->indptr[user], ue = usf->indptr[user + 1];
    int32_t d = m->d;
    double avg = 0.0;
    avg += update_biases(itf, is, ie, m->item_b, m->item_bG, m->item_bM, loss, m, ia);
    avg += update_biases(usf, us, ue, m->user_b, m->user_bG, m->user_bM, loss, m, ua);
    for (int32_t c = 0; c < d; c++) {
        flt uc = urep[c], ic = irep[c];
        avg += update_features(itf, m->item_W, m->item_G, m->item_M, d, c, is, ie,
                               loss * (double)uc, m, ia);
        avg += update_features(usf, m->user_W, m->user_G, m->user_M, d, c, us, ue,
                               loss * (double)ic, m, ua);
    }
    avg /= (double)((d + 1) * (ue - us) + (d + 1) * (ie - is));
    m->item_scale *= (1.0 + ia * avg);
    m->user_scale *= (1.0 + ua * avg);
}

/* PYX:537-649 */
static void warp_update(double loss, const orc_csr *itf, const orc_csr *usf, int32_t user,
                        int32_t pos, int32_t neg, const flt *urep, const flt *prep,
                        const flt *nrep, orc_model *m, double ia, double ua)
{
    int32_t ps = itf->indptr[pos], pe = itf->indptr[pos + 1];
    int32_t ns = itf->indptr[neg], ne = itf->indptr[neg + 1];
    int32_t us = usf->indptr[user], ue = usf->indptr[user + 1];
    int32_t d = m->d;
    double avg = 0.0;
    avg += update_biases(itf, ps, pe, m->item_b, m->item_bG, m->item_bM, -loss, m, ia);
    avg += update_biases(itf, ns, ne, m->item_b, m->item_bG, m->item_bM, loss, m, ia);
    avg += update_biases(usf, us, ue, m->user_b, m->user_bG, m->user_bM, loss, m, ua);
    for (int32_t c = 0; c < d; c++) {
        flt uc = urep[c], pc = prep[c], nc = nrep[c];
        flt diff = nc - pc; /* float32 subtraction, PYX:634-635 */
        avg += update_features(itf, m->item_W, m->item_G, m->item_M, d, c, ps, pe,
                               -loss * (double)uc, m, ia);
        avg += update_features(itf, m->item_W, m->item_G, m->item_M, d, c, ns, ne,
                               loss * (double)uc, m, ia);
        avg += update_features(usf, m->user_W, m->user_G, m->user_M, d, c, us, ue,
                               loss * (double)diff, m, ua);
    }
    avg /= (double)((d + 1) * (ue - us) + (d + 1) * (pe - ps) + (d + 1) * (ne - ns));
    m->item_scale *= (1.0 + ia * avg);
    m->user_scale *= (1.0 + ua * avg);
}

/* PYX:652-675 */
void orc_regularize(orc_model *m)
{
    int32_t d = m->d;
    for (int64_t i = 0; i < m->n_item_feat; i++) {
        for (int32_t j = 0; j < d; j++)
            m->item_W[i * d + j] = (flt)((double)m->item_W[i * d + j] / m->item_scale);
        m->item_b[i] = (flt)((double)m->item_b[i] / m->item_scale);
    }
    for (int64_t i = 0; i < m->n_user_feat; i++) {
        for (int32_t j = 0; j < d; j++)
            m->user_W[i * d + j] = (flt)((double)m->user_W[i * d + j] / m->user_scale);
        m->user_b[i] = (flt)((double)m->user_b[i] / m->user_scale);
    }
    m->item_scale = 1.0;
    m->user_scale = 1.0;
}

#define MAX_REG_SCALE 1000000.0 /* PYX:19 */

/* PYX:678-691 (single-threaded here, so no lock) */
static void maybe_regularize(orc_model *m)
{
    if (m->item_scale > MAX_REG_SCALE || m->user_scale > MAX_REG_SCALE) orc_regularize(m);
}

typedef struct { flt *u, *p, *n; } scratch;

static int scratch_alloc(scratch *s, int32_t d)
{
    s->u = malloc(sizeof(flt) * (d + 1));
    s->p = malloc(sizeof(flt) * (d + 1));
    s->n = malloc(sizeof(flt) * (d + 1));
    return s->u && s->p && s->n;
}

static void scratch_free(scratch *s) { free(s->u); free(s->p); free(s->n); }

static void log_pos(orc_opts *o, int64_t i, int32_t neg, int32_t sampled)
{
    if (o && o->neg_log) o->neg_log[i] = neg;
    if (o && o->sampled_log) o->sampled_log[i] = sampled;
}

static const orc_opts default_opts;

/* ------------------------------------------------------------ fit_warp --- */

/* PYX:784-912.  `seeds` = random_state.randint(0, INT32_MAX, size=num_threads)
 * as uint32 (PYX:812-814); thread t walks its static chunk with seeds[t]. */
int orc_fit_warp(const orc_csr *itf, const orc_csr *usf, const orc_csr *positives,
                 const int32_t *user_ids, const int32_t *item_ids, const flt *Y,
                 const flt *weight, const int32_t *shuffle, int64_t n, orc_model *m,
                 double item_alpha, double user_alpha, const uint32_t *seeds, int32_t n_seeds,
                 orc_opts *o)
{
    orc_opts local = default_opts;
    if (!o) o = &local;
    scratch s;
    if (!scratch_alloc(&s, m->d)) return -1;
    const double MAX_LOSS = 10.0;
    int32_t d = m->d;
    for (int32_t t = 0; t < n_seeds; t++) {
        int64_t lo, hi;
        static_chunk(n, n_seeds, t, &lo, &hi);
        uint32_t state = seeds[t];
        for (int64_t i = lo; i < hi; i++) {
            int32_t row = shuffle[i];
            int32_t user = user_ids[row], pos = item_ids[row];
            log_pos(o, i, -1, 0);
            if (!(Y[row] > 0)) continue; /* PYX:831-832, before any RNG use */
            if (o->rng_mode == 1) state = orc_position_seed(seeds[0], (uint64_t)i);
            flt w = weight[row];
            o->counters[0]++;
            compute_representation(usf, m->user_W, m->user_b, d, user, m->user_scale, s.u);
            compute_representation(itf, m->item_W, m->item_b, d, pos, m->item_scale, s.p);
            double pp = (double)compute_prediction(s.u, s.p, d, o->dot_mode);
            int32_t sampled = 0, chosen = -1;
            while (sampled < m->max_sampled) {
                sampled++;
                int32_t neg = orc_rand_r(&state) % itf->rows; /* PYX:860-861 */
                o->counters[1]++;
                compute_representation(itf, m->item_W, m->item_b, d, neg, m->item_scale, s.n);
                double np_ = (double)compute_prediction(s.u, s.n, d, o->dot_mode);
                if (np_ > pp - 1.0) { /* PYX:875, compared as doubles */
                    o->counters[3]++;
                    if (orc_in_positives(neg, user, positives)) continue; /* PYX:878-879 */
                    /* PYX:881; C_OMP:7446: C integer division, then floor of an integer */
                    double fl = floor((double)((long)(itf->rows - 1) / (long)sampled));
                    double loss = (double)w * log(fl > 1.0 ? fl : 1.0);
                    if (loss > MAX_LOSS) loss = MAX_LOSS;
                    warp_update(loss, itf, usf, user, pos, neg, s.u, s.p, s.n, m, item_alpha,
                                user_alpha);
                    o->counters[2]++;
                    chosen = neg;
                    break;
                }
            }
            log_pos(o, i, chosen, sampled);
            maybe_regularize(m); /* PYX:901-904 */
        }
    }
    scratch_free(&s);
    orc_regularize(m); /* PYX:910-912 */
    return 0;
}

/* ------------------------------------------------------------- fit_bpr --- */

/* PYX:1074-1182 */
int orc_fit_bpr(const orc_csr *itf, const orc_csr *usf, const orc_csr *positives,
                const int32_t *user_ids, const int32_t *item_ids, const flt *Y, const flt *weight,
                const int32_t *shuffle, int64_t n, orc_model *m, double item_alpha,
                double user_alpha, const uint32_t *seeds, int32_t n_seeds, orc_opts *o)
{
    orc_opts local = default_opts;
    if (!o) o = &local;
    scratch s;
    if (!scratch_alloc(&s, m->d)) return -1;
    int32_t d = m->d;
    for (int32_t t = 0; t < n_seeds; t++) {
        int64_t lo, hi;
        static_chunk(n, n_seeds, t, &lo, &hi);
        uint32_t state = seeds[t];
        for (int64_t i = lo; i < hi; i++) {
            int32_t row = shuffle[i];
            log_pos(o, i, -1, 0);
            if (!(Y[row] > 0)) continue; /* PYX:1116-1117 */
            if (o->rng_mode == 1) state = orc_position_seed(seeds[0], (uint64_t)i);
            flt w = weight[row];
            int32_t user = user_ids[row], pos = item_ids[row];
            o->counters[0]++;
            int32_t neg = 0, draws = 0;
            for (int64_t j = 0; j < n; j++) { /* PYX:1123-1127 */
                neg = item_ids[orc_rand_r(&state) % (int32_t)n];
                draws++;
                o->counters[1]++;
                o->counters[3]++;
                if (!orc_in_positives(neg, user, positives)) break;
            }
            compute_representation(usf, m->user_W, m->user_b, d, user, m->user_scale, s.u);
            compute_representation(itf, m->item_W, m->item_b, d, pos, m->item_scale, s.p);
            compute_representation(itf, m->item_W, m->item_b, d, neg, m->item_scale, s.n);
            double pp = (double)compute_prediction(s.u, s.p, d, o->dot_mode);
            double np_ = (double)compute_prediction(s.u, s.n, d, o->dot_mode);
            /* PYX:1158; C_OMP:9316: (pp - np) narrowed to float32 for sigmoid */
            double loss = (double)w * (1.0 - (double)sigmoid((flt)(pp - np_)));
            warp_update(loss, itf, usf, user, pos, neg, s.u, s.p, s.n, m, item_alpha, user_alpha);
            o->counters[2]++;
            log_pos(o, i, neg, draws);
            maybe_regularize(m);
        }
    }
    scratch_free(&s);
    orc_regularize(m);
    return 0;
}

/* -------------------------------------------------------- fit_logistic --- */

/* PYX:694-781 (no RNG, no positives lookup; every row is visited) */
int orc_fit_logistic(const orc_csr *itf, const orc_csr *usf, const int32_t *user_ids,
                     const int32_t *item_ids, const flt *Y, const flt *weight,
                     const int32_t *shuffle, int64_t n, orc_model *m, double item_alpha,
                     double user_alpha, orc_opts *o)
{
    orc_opts local = default_opts;
    if (!o) o = &local;
    scratch s;
    if (!scratch_alloc(&s, m->d)) return -1;
    int32_t d = m->d;
    for (int64_t i = 0; i < n; i++) {
        int32_t row = shuffle[i];
        int32_t user = user_ids[row], item = item_ids[row];
        flt w = weight[row];
        compute_representation(usf, m->user_W, m->user_b, d, user, m->user_scale, s.u);
        compute_representation(itf, m->item_W, m->item_b, d, item, m->item_scale, s.p);
        double prediction = (double)sigmoid(compute_prediction(s.u, s.p, d, o->dot_mode));
        int y = (Y[row] <= 0) ? 0 : 1; /* PYX:751-755 */
        if (y) o->counters[0]++;
        double loss = (double)w * (prediction - (double)y); /* C_OMP:6628 */
        update(loss, itf, usf, user, item, s.u, s.p, m, item_alpha, user_alpha);
        o->counters[2]++;
        maybe_regularize(m);
    }
    scratch_free(&s);
    orc_regularize(m);
    return 0;
}

/* -------------------------------------------------------- fit_warp_kos --- */

typedef struct { int32_t idx; flt val; } pair_t; /* PYX:109-111 */

/* PYX:114-122: never returns 0 */
static int reverse_pair_compare(const void *a, const void *b)
{
    flt diff = ((const pair_t *)a)->val - ((const pair_t *)b)->val;
    return diff < 0 ? 1 : -1;
}

/* PYX:915-1071.  Iterates over COO rows (user_ids), ignores Y and weights. */
int orc_fit_warp_kos(const orc_csr *itf, const orc_csr *usf, const orc_csr *data,
                     const int32_t *user_ids, const int32_t *shuffle, int64_t n, orc_model *m,
                     double item_alpha, double user_alpha, int32_t k, int32_t n_pos,
                     const uint32_t *seeds, int32_t n_seeds, orc_opts *o)
{
    orc_opts local = default_opts;
    if (!o) o = &local;
    scratch s;
    if (!scratch_alloc(&s, m->d)) return -1;
    pair_t *pairs = malloc(sizeof(pair_t) * (size_t)(n_pos > 0 ? n_pos : 1));
    if (!pairs) return -1;
    const double MAX_LOSS = 10.0;
    int32_t d = m->d;
    for (int32_t t = 0; t < n_seeds; t++) {
        int64_t lo, hi;
        static_chunk(n, n_seeds, t, &lo, &hi);
        uint32_t state = seeds[t];
        for (int64_t i = lo; i < hi; i++) {
            int32_t row = shuffle[i];
            int32_t user = user_ids[row];
            log_pos(o, i, -1, 0);
            if (o->rng_mode == 1) state = orc_position_seed(seeds[0], (uint64_t)i);
            compute_representation(usf, m->user_W, m->user_b, d, user, m->user_scale, s.u);
            int32_t start = data->indptr[user], stop = data->indptr[user + 1];
            if (stop == start) continue; /* PYX:971-972 */
            o->counters[0]++;
            int32_t no_pos = n_pos < stop - start ? n_pos : stop - start; /* PYX:975 */
            for (int32_t j = 0; j < no_pos; j++) {
                int32_t it = data->indices[sample_range(start, stop, &state)];
                compute_representation(itf, m->item_W, m->item_b, d, it, m->item_scale, s.p);
                pairs[j].idx = it;
                pairs[j].val = compute_prediction(s.u, s.p, d, o->dot_mode);
            }
            qsort(pairs, (size_t)no_pos, sizeof(pair_t), reverse_pair_compare); /* PYX:997 */
            int32_t kk = (k < no_pos ? k : no_pos) - 1;                          /* PYX:1002 */
            int32_t pos = pairs[kk].idx;
            double pp = (double)pairs[kk].val;
            compute_representation(itf, m->item_W, m->item_b, d, pos, m->item_scale, s.p);
            int32_t sampled = 0, chosen = -1;
            while (sampled < m->max_sampled) {
                sampled++;
                int32_t neg = orc_rand_r(&state) % itf->rows;
                o->counters[1]++;
                compute_representation(itf, m->item_W, m->item_b, d, neg, m->item_scale, s.n);
                double np_ = (double)compute_prediction(s.u, s.n, d, o->dot_mode);
                if (np_ > pp - 1.0) {
                    o->counters[3]++;
                    if (orc_in_positives(neg, user, data)) continue;
                    /* PYX:1039; C_OMP:8452: no max(1.0, .) and no sample weight */
                    double loss = log(floor((double)((long)(itf->rows - 1) / (long)sampled)));
                    if (loss > MAX_LOSS) loss = MAX_LOSS;
                    warp_update(loss, itf, usf, user, pos, neg, s.u, s.p, s.n, m, item_alpha,
                                user_alpha);
                    o->counters[2]++;
                    chosen = neg;
                    break;
                }
            }
            log_pos(o, i, chosen, sampled);
            maybe_regularize(m);
        }
    }
    free(pairs);
    scratch_free(&s);
    orc_regularize(m);
    return 0;
}

/* ------------------------------------------------------------- predict --- */

/* PYX:1185-1229 */
int orc_predict(const orc_csr *itf, const orc_csr *usf, const int32_t *user_ids,
                const int32_t *item_ids, flt *out, int64_t n, const orc_model *m,
                int32_t dot_mode)
{
    scratch s;
    if (!scratch_alloc(&s, m->d)) return -1;
    for (int64_t i = 0; i < n; i++) {
        compute_representation(usf, m->user_W, m->user_b, m->d, user_ids[i], m->user_scale, s.u);
        compute_representation(itf, m->item_W, m->item_b, m->d, item_ids[i], m->item_scale, s.p);
        out[i] = compute_prediction(s.u, s.p, m->d, dot_mode);
    }
    scratch_free(&s);
    return 0;
}

/* PYX:1232-1323 */
int orc_predict_ranks(const orc_csr *itf, const orc_csr *usf, const orc_csr *test,
                      const orc_csr *train, flt *ranks, const orc_model *m, int32_t dot_mode)
{
    scratch s;
    if (!scratch_alloc(&s, m->d)) return -1;
    int32_t maxlen = 0;
    for (int32_t u = 0; u < test->rows; u++) {
        int32_t len = test->indptr[u + 1] - test->indptr[u];
        if (len > maxlen) maxlen = len;
    }
    flt *preds = malloc(sizeof(flt) * (size_t)(maxlen + 1));
    int32_t *ids = malloc(sizeof(int32_t) * (size_t)(maxlen + 1));
    if (!preds || !ids) return -1;
    for (int32_t u = 0; u < test->rows; u++) {
        int32_t rs = test->indptr[u], re = test->indptr[u + 1];
        if (re == rs) continue;
        compute_representation(usf, m->user_W, m->user_b, m->d, u, m->user_scale, s.u);
        for (int32_t i = 0; i < re - rs; i++) {
            int32_t it = test->indices[rs + i];
            compute_representation(itf, m->item_W, m->item_b, m->d, it, m->item_scale, s.p);
            ids[i] = it;
            preds[i] = compute_prediction(s.u, s.p, m->d, dot_mode);
        }
        for (int32_t it = 0; it < test->cols; it++) {
            if (orc_in_positives(it, u, train)) continue; /* PYX:1303-1304 */
            compute_representation(itf, m->item_W, m->item_b, m->d, it, m->item_scale, s.p);
            flt pr = compute_prediction(s.u, s.p, m->d, dot_mode);
            for (int32_t i = 0; i < re - rs; i++)
                if (it != ids[i] && pr >= preds[i]) ranks[rs + i] += 1.0f; /* PYX:1317-1319 */
        }
    }
    free(preds);
    free(ids);
    scratch_free(&s);
    return 0;
}

static int flt_compare(const void *a, const void *b) /* PYX:135-142 */
{
    flt d = *(const flt *)a - *(const flt *)b;
    return d > 0 ? 1 : (d < 0 ? -1 : 0);
}

/* PYX:1326-1376.  Sorts rank_data in place, exactly like the reference. */
int orc_auc_from_rank(const orc_csr *ranks, const int32_t *num_train_positives, flt *rank_data,
                      flt *auc)
{
    for (int32_t u = 0; u < ranks->rows; u++) {
        int32_t rs = ranks->indptr[u], re = ranks->indptr[u + 1];
        int32_t npos = re - rs;
        int32_t nneg = ranks->cols - (npos + num_train_positives[u]);
        if (npos == 0 || nneg == ranks->cols) { auc[u] = 0.5f; continue; }
        qsort(&rank_data[rs], (size_t)npos, sizeof(flt), flt_compare);
        for (int32_t i = 0; i < npos; i++) {
            flt rank = ranks->data[rs + i];
            rank = rank - (flt)i;
            if (rank < 0) rank = 0;
            /* C: auc += 1.0 - rank / num_negatives  (double arithmetic, float store) */
            auc[u] = (flt)((double)auc[u] + (1.0 - (double)(rank / (flt)nneg)));
        }
        if (npos != 0) auc[u] = auc[u] / (flt)npos;
    }
    return 0;
}
