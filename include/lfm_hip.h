/*
 * lfm_hip.h -- C ABI of liblfm_hip.so: the MI355X (gfx950) replacement for
 * LightFM's native extension `lightfm._lightfm_fast`.
 *
 * Every entry point names the reference interface it replaces
 *   PYX = /root/reference/lightfm/_lightfm_fast.pyx.template
 *   LFM = /root/reference/lightfm/lightfm.py
 * and is what a ctypes/cffi binding inside the reference's lightfm/_lightfm_fast.py
 * would bind.  Plain pointers and sizes only; all buffers
 * are HOST buffers owned by the caller (numpy arrays in the reference), exactly
 * like the typed memoryviews the Cython functions take.  Weights are updated in
 * place (one-shot calls) or kept device-resident between calls (session API).
 *
 * All functions return 0 on success and a negative LFM_E* code on failure;
 * lfm_last_error() returns a thread-local description.  There is no CPU
 * fallback: without a usable HIP device every compute call fails with
 * LFM_ENODEV.
 */
#ifndef LFM_HIP_H
#define LFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFM_OK 0
#define LFM_EINVAL (-1)   /* bad argument (the reference would raise ValueError) */
#define LFM_ENODEV (-2)   /* no HIP device / HIP runtime failure */
#define LFM_ENOMEM (-3)   /* device or host allocation failed */
#define LFM_ECOMM (-4)    /* RCCL failure */
#define LFM_EUNSUPPORTED (-5)

/* Widest model lfm_session_create accepts (the reference has no bound on no_components, PYX:185-259; here a lane of the
 * one-interaction-per-wavefront kernels keeps up to 16 coordinates of a row): LFM_EUNSUPPORTED beyond. */
#ifndef LFM_MAX_COMPONENTS
#define LFM_MAX_COMPONENTS 1024
#endif
#define LFM_ECORRUPT (-6) /* an epoch kernel read a shuffle entry outside [0, n) (invalid shuffle input or
                              corrupted device memory); LIGHTFM_AMD_VALIDATE=1 (debugging): a read-only device input of the session
                             changed, or the shuffle slot is not a permutation (csrc/session.hip) */

/* CSRMatrix (PYX:145-182): int32 indices/indptr, float32 data, C-contiguous. */
typedef struct lfm_csr {
    const int32_t *indices;
    const int32_t *indptr;
    const float *data;
    int32_t rows, cols;
    int64_t nnz;
} lfm_csr;

/* FastLightFM (PYX:185-259): the 12 weight arrays in constructor order plus the
 * hyper-parameters.  *_W/_G/_M are [n_feat, d] row-major float32, *_b/_bG/_bM
 * are [n_feat].  lr/rho/eps are float32 exactly as the reference stores them. */
typedef struct lfm_model {
    float *item_W, *item_G, *item_M, *item_b, *item_bG, *item_bM;
    float *user_W, *user_G, *user_M, *user_b, *user_bG, *user_bM;
    int32_t n_item_feat, n_user_feat;
    int32_t d;        /* no_components */
    int32_t adadelta; /* 0 = adagrad, 1 = adadelta */
    float lr, rho, eps;
    int32_t max_sampled;
    double item_scale, user_scale; /* always 1.0 between calls (PYX:674-675) */
} lfm_model;

/* Execution modes (not part of the reference's Python API). */
#define LFM_MODE_PARALLEL 0 /* Hogwild over thousands of wavefronts; one PRNG stream per
                               shuffled position (seed = f(seeds[0], position))        */
#define LFM_MODE_SERIAL 1   /* ONE wavefront walks the shuffled list in order with the
                               reference's per-thread rand_r streams: reproduces the
                               reference (num_threads = n_seeds run back to back)
                               bit for bit; for parity tests                           */

typedef struct lfm_opts {
    int32_t mode;               /* LFM_MODE_* */
    int32_t launches_per_epoch; /* parallel mode: kernel launches per epoch (a launch
                                   boundary is a device-wide release/acquire); 0 = auto */
    int32_t first_batch;        /* negatives scored speculatively in the first batch; 0 = auto */
    int32_t max_waves;          /* parallel mode: FIXED cap on interactions in flight (between
                                   reading the weights and publishing the update); 0 = auto =
                                   ramp with the training history up to the whole chip
                                   (`history`, `ramp_k` below; DESIGN.md "Hogwild at GPU width") */
    int32_t *neg_log;           /* host [n] or NULL: chosen negative per shuffled position, -1 = none */
    int32_t *sampled_log;       /* host [n] or NULL: draws consumed per shuffled position */
    int64_t counters[4];        /* out: positives visited, draws, updates, in_positives probes */
    float kernel_ms;            /* out: device time of the epoch's kernels (HIP events)   */
    int32_t update_mode;        /* parallel mode, how a cell update reaches memory:
                                   0 = auto = 3;
                                   1 = plain load / store (the literal Hogwild of the reference's
                                       OpenMP loop; on a GPU, whose L2s are neither coherent
                                       with each other nor write-through, whole updates are
                                       lost -- kept for experiments and the bit-exactness tests);
                                   2 = compute but do not write (profiling ablation);
                                   3 = publish new - old with global_atomic_add_f32: no update
                                       is lost (DESIGN.md "Hogwild at GPU width")              */
    int32_t feat_kernel;        /* parallel mode, models the lane-group tile kernel does not cover
                                   (feature CSRs, BPR, k-OS, logistic): 0 = auto (the pipelined
                                   row-stream kernels, csrc/feat_kernel.hpp: adagrad up to d = 256, adadelta up to
                                   d = 128, any alpha), 1 = force the generic kernels, 2 = row-stream kernels
                                   instrumented with per-phase cycle counters (BPR / k-OS, d > 64) */
    int32_t warp_kernel;        /* parallel-mode WARP with identity features (any alpha; adadelta only
                                   without regularisation): 0 = auto (the lane-group tile kernel,
                                   csrc/warp_tile.hip, when d % 4 == 0 and d <= 128),
                                   1 = do not use the tile kernel (the row-stream or generic kernels run),
                                   2 = tile kernel instrumented with per-phase cycle counters */
    int32_t debug;              /* kernel experiments; bits 0-2: force the tile kernel's
                                   interactions per wavefront pass (1, 2 or 4), 0 = auto; bit 6 (64):
                                   the register-staged variant of the 4-per-pass tile kernel instead of
                                   the LDS-DMA one (global_load_lds_dwordx4); bit 0 with the profiling
                                   builds: drain the memory counters at every phase stamp; bit 5 (32):
                                   no bias snapshots; bit 7 (128): consecutive full-size launches on ONE stream
                                   (default: two streams alternately, so that a launch's draining tail is
                                   filled by the next launch's workgroups); bit 8 (256): tile kernel without the
                                   Bloom pre-filter of in_positives; bit 9 (512): the filter probed for every
                                   candidate with its row instead of for the violators after the scoring pass; bit 14
                                   (16384): a hybrid model's shared feature rows stay on the float atomics instead of being
                                   accumulated in LDS slices between launches (csrc/hot_slices.hip; `plan_flags` bit 5) */
    int64_t phase_cycles[8];    /* out, warp_kernel = 2 / feat_kernel = 2 (profiling builds): shader
                                   cycles summed over wavefronts per phase of a pass -- 0 loop
                                   head, 1 gathers, 2 scoring, 3 in_positives, 4 accumulator
                                   loads, 5 cell arithmetic + atomics, 6 tail */
    int32_t tile_ng;            /* out: interactions per wavefront pass the tile kernel ran with
                                   (4, 2, 1), 0 = a generic kernel ran                       */
    int32_t in_flight;          /* out: interactions in flight of the epoch's last launch   */
    int64_t history;            /* in: interactions this model has already been trained on (all
                                   earlier epochs); concurrency is ramped with it, see max_waves.
                                   0 = a fresh (or unknown) model: the epoch starts the ramp   */
    int32_t ramp_k;             /* in: at most (history + done) / ramp_k interactions in flight;
                                   0 = auto (32), < 0 = no ramp                               */
    int32_t launches;           /* out: kernel launches of the epoch                        */
    int32_t kernel_used;        /* out: 0 generic one-interaction-per-wavefront kernels, 1 lane-group
                                   WARP tile kernel, 2 pipelined row-stream kernels (feat_kernel.hpp) */
    int32_t shared_cap;         /* in: steady-state bound on interactions in flight for models with shared
                                   feature rows; 0 = auto (feature rows / avg nnz per row of the side),
                                   < 0 = none */
    int64_t pos_begin, pos_end; /* in, parallel mode: run only shuffled positions [pos_begin, pos_end)
                                   of the slot (0, 0 = the whole epoch).  The multi-GPU driver runs
                                   an epoch as segments with a merge of the replicated tables
                                   between them (lfm_session_comm_merge); `history` must then count
                                   the positions of earlier segments too                        */
    int32_t streams_used;       /* out: HIP streams the epoch's launches were spread over: 2 when consecutive
                                   full-residency launches of the tile kernel alternated between the session's
                                   two streams (see `debug` bit 7), else 1                                      */
    int32_t user_store;         /* out: 1 when the epoch wrote the USER rows of its updates with plain stores instead of float
                                   atomics: parallel mode, identity user features, W and G of the user side in uncached memory,
                                   adagrad, a model of <= 192 MB and few same-user collisions in the data -- (interactions in
                                   flight at full residency) x sum_u c_u^2 / n^2 <= 0.3 (csrc/session.hip; `debug` bit 11 = 2048
                                   forces it for uncached tables, bit 12 = 4096 switches it off)                    */
    int32_t tile_ahead;         /* out: 1 when the epoch's last launch ran the steady-state variant of the tile kernel
                                   with the next pass's gather issued inside the current pass (csrc/warp_tile_ahead.hpp;
                                   `debug` bit 10 = 1024 keeps the plain tile kernel)                            */
    int32_t plan_flags;         /* out: what the epoch's launch plan looked like (tests assert the branch they mean to cover):
                                   bit 0 / 1 = the item / user side's biases were scored from per-launch cached snapshots (else
                                   the live table: sides above 2 MiB of biases); bit 2 / 3 = the item / user embedding tables
                                   live in uncached memory; bit 4 = the item embedding table is >= 4 GB (64-bit row offsets
                                   in the tile kernel's gathers); bit 5 = the shared tag rows were accumulated in LDS slices
                                   (csrc/hot_slices.hip) instead of by float atomics of every interaction; bit 6 = the narrow-model
                                   tile kernel ran (rows of <= 16 floats: two interactions per lane group, eight per
                                   wavefront pass; csrc/warp_tile_narrow.hpp); bit 7 = ... on rows that carry W, G, b and bG
                                   of a feature in ONE 128-byte line (d <= 12: an update is three line operations); bit 8 = the
                                   logistic lane-group kernel ran on such rows (csrc/logistic_tile.hip: identity features,
                                   d <= 12 -- the reference's default LightFM()); bit 9 = its BPR counterpart ran; bit 10 = fit_bpr ran
                                   on the BPR instantiations of the tile kernel (csrc/warp_tile_bpr.hip: identity features, 12 < d <= 256);
                                   bit 11 = fit_logistic on its logistic instantiations                                    */
} lfm_opts;

#define LFM_LOSS_LOGISTIC 0
#define LFM_LOSS_WARP 1
#define LFM_LOSS_BPR 2
#define LFM_LOSS_WARP_KOS 3

const char *lfm_last_error(void);
/* device time (HIP events, ms) of the kernels of this thread's last lfm_*predict_ranks call */
float lfm_last_kernel_ms(void);
int lfm_device_count(void);
/* name (<=255 chars) and CU count of a device; used by bench.py */
int lfm_device_info(int device, char *name, int32_t *cus, int64_t *hbm_bytes);

/* Device memory of the library comes from a process-wide pool (csrc/pool.hpp): buffers of destroyed
 * sessions are kept and reused instead of going through hipFree / hipMalloc (see DESIGN.md "Root cause of
 * the round-2 process abort").  lfm_device_trim hands the unused part back to the HIP runtime (bytes via
 * *released, may be NULL); lfm_device_pool_stats reports the bytes held / currently unused.  No
 * counterpart in the reference (host memory is numpy's). */
int lfm_device_trim(int64_t *released);
/* Host-side helper of the Python class (no device involved): ONE multi-threaded pass over a float32 array that
 * answers what LFM:383-386 (np.array_equiv(data, 1.0)) and LFM:447-472 / 617-625 (isfinite(sum)) ask in two to
 * three numpy passes.  *all_ones = every value == 1.0f; *finite = no inf / nan and the sum inside float32 range. */
int lfm_host_scan_f32(const float *p, int64_t n, int32_t *all_ones, int32_t *finite);
/* Host-side helper: *out = sum of word[i] * (2 i + 1) modulo 2^64 over n 32-bit words -- the signature with which the
 * Python class re-validates a resident scoring session against the caller's weight arrays (no BLAS, four threads). */
int lfm_host_checksum_u32(const uint32_t *p, int64_t n, uint64_t *out);
/* Host-side helper: out[i] = (float)((u_i - 0.5) / d) for the next n uniform doubles u_i of numpy's legacy MT19937
 * stream (RandomState.rand: LFM:281-312 initialises the embedding tables with it), on the 624-word key and the position
 * of RandomState.get_state(), both updated in place -- bit-identical to numpy, values and stream position. */
int lfm_host_mt19937_table(uint32_t *key, int32_t *pos, float *out, int64_t n, int32_t d);
int lfm_device_pool_stats(int64_t *reserved, int64_t *cached);
/* Device self-test (tests): n pseudo-random adagrad cells (PYX:416-449) through the kernels' exact float64 cell and through the
 * variant without the float64 square root and division that the hot-slice kernel runs (csrc/device.hpp: cell_math_adagrad);
 * *mismatches = cells whose new (W, G) bit patterns differ (the variant is bit-identical by construction: 0), *fallbacks =
 * cells for which the variant took its exact fallback (results too close to a float32 rounding boundary). */
int lfm_selftest_adagrad_cell(int64_t n, uint32_t seed, float learning_rate, int64_t *mismatches, int64_t *fallbacks);
/* Device self-test (tests): `tiles` pseudo-random 32 x 32 tiles of (user, item) pairs with no_components = d (components and
 * biases of random sign over `spread` binades) through the instruction sequence of predict_ranks' default sweep -- the products on
 * the bf16 matrix pipe with two-way split operands (csrc/predict_kernels.hip: ranks_mfma3_kernel<.., true>) -- against the
 * reference's sequential float32 dot (PYX:320-334): *worst_fraction = the largest |difference| / (the rounding band the sweep takes
 * for the pair), *beyond = pairs outside their band (the band's assumption about the pipe's internal rounding holds iff 0). */
int lfm_selftest_ranks_bf16_band(int64_t tiles, uint32_t seed, int32_t d, int32_t spread, float *worst_fraction, int64_t *beyond);

/* ------------------------------------------------------------------------
 * One-shot epoch drivers: upload, run ONE epoch on device 0, download.
 * Drop-in for the Cython functions of the same name; `seeds` is
 * random_state.randint(0, INT32_MAX, size=num_threads).astype(uint32)
 * (PYX:812-814) drawn by the caller so its RandomState advances identically.
 * `opts` may be NULL (parallel mode, defaults).
 * ------------------------------------------------------------------------ */

/* fit_warp, PYX:784-912 (call site LFM:695-711) */
int lfm_fit_warp(const lfm_csr *item_features, const lfm_csr *user_features,
                 const lfm_csr *interactions, const int32_t *user_ids, const int32_t *item_ids,
                 const float *Y, const float *sample_weight, const int32_t *shuffle_indices,
                 int64_t n, lfm_model *model, double item_alpha, double user_alpha,
                 const uint32_t *seeds, int32_t n_seeds, lfm_opts *opts);

/* fit_bpr, PYX:1074-1182 (LFM:713-728) */
int lfm_fit_bpr(const lfm_csr *item_features, const lfm_csr *user_features,
                const lfm_csr *interactions, const int32_t *user_ids, const int32_t *item_ids,
                const float *Y, const float *sample_weight, const int32_t *shuffle_indices,
                int64_t n, lfm_model *model, double item_alpha, double user_alpha,
                const uint32_t *seeds, int32_t n_seeds, lfm_opts *opts);

/* fit_logistic, PYX:694-781 (LFM:746-759) */
int lfm_fit_logistic(const lfm_csr *item_features, const lfm_csr *user_features,
                     const int32_t *user_ids, const int32_t *item_ids, const float *Y,
                     const float *sample_weight, const int32_t *shuffle_indices, int64_t n,
                     lfm_model *model, double item_alpha, double user_alpha, lfm_opts *opts);

/* fit_warp_kos, PYX:915-1071 (LFM:730-744) */
int lfm_fit_warp_kos(const lfm_csr *item_features, const lfm_csr *user_features,
                     const lfm_csr *data, const int32_t *user_ids,
                     const int32_t *shuffle_indices, int64_t n, lfm_model *model,
                     double item_alpha, double user_alpha, int32_t k, int32_t n_positives,
                     const uint32_t *seeds, int32_t n_seeds, lfm_opts *opts);

/* predict_lightfm, PYX:1185-1229 (LFM:862-870): predictions[i] for pairs */
int lfm_predict(const lfm_csr *item_features, const lfm_csr *user_features,
                const int32_t *user_ids, const int32_t *item_ids, float *predictions, int64_t n,
                const lfm_model *model);

/* predict_ranks, PYX:1232-1323 (LFM:979-987): ranks[] += count, in place.  The dense all-items pass runs
 * on the matrix cores (v_mfma_f32_32x32x2_f32 pre-filter + sequential-dot re-check inside the rounding
 * band): the ranks are the reference's integers.  LIGHTFM_AMD_RANKS_MFMA = 0 | 1 | 2 select the scalar / the
 * first / the second MFMA kernel (cross-checks); unset or 3 = the bucket-search sweep. */
int lfm_predict_ranks(const lfm_csr *item_features, const lfm_csr *user_features,
                      const lfm_csr *test_interactions, const lfm_csr *train_interactions,
                      float *ranks, const lfm_model *model);

/* calculate_auc_from_rank, PYX:1326-1376 (evaluation.py:247-249).  rank_data is
 * sorted in place per row exactly like the reference. */
int lfm_auc_from_rank(const lfm_csr *ranks, const int32_t *num_train_positives, float *rank_data,
                      float *auc);

/* __test_in_positives, PYX:1380-1385 (tests/test_fast_functions.py:9-17); 1/0 or <0 */
int lfm_in_positives(int32_t row, int32_t col, const lfm_csr *mat);

/* ------------------------------------------------------------------------
 * Device-resident session: what LightFM.fit_partial's epoch loop (LFM:654-664)
 * uses so weights/CSR/COO are uploaded once per fit_partial, not once per epoch.
 * ------------------------------------------------------------------------ */
typedef struct lfm_session lfm_session;

/* Uploads the model and both feature matrices to `device`. */
int lfm_session_create(lfm_session **out, int device, const lfm_model *model,
                       const lfm_csr *item_features, const lfm_csr *user_features);
/* A SCORING session: only what predict / predict_rank / get_*_representations read is uploaded and kept
 * resident -- embeddings and biases of both sides (the accumulator / momentum pointers of `model` may be
 * NULL).  The host class keeps one per model between calls (call sites LFM:862-870, 979-987, which hand
 * the full FastLightFM to every call) and re-uploads nothing while the weights are unchanged.
 * lfm_session_epoch and the merge calls fail on it with LFM_EINVAL; lfm_session_load_model re-uploads
 * just these four arrays.  The one-shot lfm_predict / lfm_predict_ranks use it internally. */
int lfm_session_create_scoring(lfm_session **out, int device, const lfm_model *model,
                               const lfm_csr *item_features, const lfm_csr *user_features);
/* Replaces the resident feature matrices (predict / predict_rank take them per call, LFM:843-860,
 * 961-977); NULL keeps a side's matrix.  The weight tables stay where they are. */
int lfm_session_set_features(lfm_session *s, const lfm_csr *item_features, const lfm_csr *user_features);
/* Uploads the training COO (+ positives lookup CSR; NULL for logistic).  Y and
 * sample_weight may alias (LFM:412-415).  item_ids/Y/sample_weight NULL for k-OS. */
int lfm_session_set_interactions(lfm_session *s, const lfm_csr *positives,
                                 const int32_t *user_ids, const int32_t *item_ids,
                                 const float *Y, const float *sample_weight, int64_t n);
/* Shuffle slots: device copies of shuffle index arrays (slot 0 is the default). */
int lfm_session_upload_shuffle(lfm_session *s, int32_t slot, const int32_t *shuffle, int64_t n);
/* Fills `slot` ON DEVICE with a keyed pseudo-random permutation of [0, n) (6-round Feistel
 * network with cycle walking): replaces the host-side random_state.shuffle(arange(n)) of
 * LFM:689-690 + the upload when the caller only needs *a* uniform shuffle, not numpy's. */
int lfm_session_device_shuffle(lfm_session *s, int32_t slot, uint32_t key0, uint32_t key1);
/* ... the same, written on a stream of the session's own WHILE the epoch of another slot runs: call it for epoch e + 1's
 * slot before lfm_session_epoch of epoch e; the epoch that uses the slot waits for it (LightFM.fit_partial alternates
 * two slots). */
int lfm_session_device_shuffle_ahead(lfm_session *s, int32_t slot, uint32_t key0, uint32_t key1);
/* The same permutation computed on the host (no device needed); and a slot's content. */
int lfm_shuffle_permutation(int32_t *out, int64_t n, uint32_t key0, uint32_t key1);
int lfm_session_download_shuffle(lfm_session *s, int32_t slot, int32_t *out, int64_t n);
/* One epoch with the shuffle held in `slot`. */
int lfm_session_epoch(lfm_session *s, int32_t loss, int32_t slot, double item_alpha,
                      double user_alpha, int32_t k, int32_t n_positives, const uint32_t *seeds,
                      int32_t n_seeds, lfm_opts *opts);
/* 1 if every item/user embedding and bias is finite (LFM:447-464), else 0. */
int lfm_session_check_finite(lfm_session *s);
/* Pairwise predictions with the resident weights (ids are host arrays). */
int lfm_session_predict(lfm_session *s, const int32_t *user_ids, const int32_t *item_ids,
                        float *predictions, int64_t n);
int lfm_session_predict_ranks(lfm_session *s, const lfm_csr *test, const lfm_csr *train,
                              float *ranks);
/* Copies the 12 arrays back into the caller's buffers. */
int lfm_session_sync_to_host(lfm_session *s, lfm_model *model);
/* The inverse: overwrites the resident tables with the caller's arrays (same shapes). */
int lfm_session_load_model(lfm_session *s, const lfm_model *model);
/* Builds the positives lookup (LFM:365-372: interactions.tocsr() with sorted indices, duplicates
 * summed away) ON DEVICE from the uploaded COO instead of taking it from the host: sort by
 * (user, item) key + unique + row pointers.  Call after lfm_session_set_interactions(positives =
 * NULL, ...); n_users / n_items = shape of the interaction matrix. */
int lfm_session_build_positives(lfm_session *s, int32_t n_users, int32_t n_items);
/* The resident positives lookup back on the host (tests): *nnz always; indptr [rows + 1] and
 * indices [*nnz] when not NULL. */
int lfm_session_download_positives(lfm_session *s, int32_t *indptr, int32_t *indices, int64_t *nnz);
/* get_item_representations / get_user_representations with a feature matrix (LFM:991-1047):
 * biases[r] = sum_f features[r,f] * b[f], embeddings[r,:] = sum_f features[r,f] * W[f,:], float32
 * accumulation in CSR order.  side 0 = item, 1 = user; embeddings is [features.rows, d] row-major. */
int lfm_session_representations(lfm_session *s, int32_t side, const lfm_csr *features, float *biases,
                                float *embeddings);
int lfm_session_destroy(lfm_session *s);

/* ------------------------------------------------------------------------
 * Multi-GPU (no reference counterpart): one process per GPU,
 * interactions sharded by user; the replicated item-side tables are merged with an
 * RCCL all-reduce of their deltas at a cadence the host driver chooses
 * (lightfm_amd/distributed.py: merge_schedule).
 * ------------------------------------------------------------------------ */
#define LFM_UNIQUE_ID_BYTES 128
/* Loads RCCL now -- the librccl next to the HIP runtime this library is linked to, by absolute path (a process that has
 * imported torch holds torch's own bundled RCCL and HIP runtime, which know nothing of this library's device contexts).
 * The Python layer calls it before importing torch in a multi-process job. */
int lfm_comm_preload(void);
int lfm_comm_unique_id(char id[LFM_UNIQUE_ID_BYTES]);
int lfm_session_comm_init(lfm_session *s, const char id[LFM_UNIQUE_ID_BYTES], int32_t rank,
                          int32_t nranks);
/* Merge of the replicated tables (all ranks must call it the same number of times; the state
 * after the call is the start of the next interval).  sides: bit 0 = item tables, bit 1 = user
 * tables (only when user features are shared; with identity user features every rank holds just
 * its own users' rows and they are never communicated).  mode:
 *   LFM_MERGE_SUM       X := X0 + sum_r (X_r - X0) for every table (local SGD with summed steps)
 *   LFM_MERGE_MEAN      embeddings/biases take the MEAN of the ranks' deltas, accumulators the sum
 *   LFM_MERGE_ADAGRAD   accumulators are summed first; every rank's embedding delta is then rescaled
 *                       by sqrt((G0 + dG_r/2) / (G0 + sum dG/2)) -- the step it would have taken had
 *                       it seen the other ranks' squared gradients too -- and the rescaled deltas summed */
#define LFM_MERGE_SUM 0
#define LFM_MERGE_MEAN 1
#define LFM_MERGE_ADAGRAD 2
int lfm_session_comm_merge(lfm_session *s, int32_t sides, int32_t mode);
/* The same merge, proportional to what the interval touched (csrc/session.hip: merge_group_sparse): at
 * merge time one streaming compare of the tables with the interval's snapshot flags the rows that changed
 * in a byte map (the epoch kernels mark nothing: marking cost the tile kernel 6 %); the ranks' maps are
 * OR-ed (an all-reduce over n_feat bytes), and only the rows of the union travel -- packed deltas of W, G,
 * b, bG, all-reduced on the session's communication stream.  overlap != 0: the call returns once the exchange is enqueued; the next
 * segment trains while it runs, and its result is applied at the start of the next merge call or by
 * lfm_session_comm_merge_flush (call it before reading the tables: check_finite, sync_to_host, the end of
 * an epoch).  *bytes (may be NULL) = what this rank handed to RCCL.  Same modes and the same arithmetic as the dense merge
 * (adadelta models carry their momentum tables too; LFM_MERGE_ADAGRAD means LFM_MERGE_MEAN for them, as there).
 * Per side and merge the packed deltas of all kinds are ONE buffer and travel in ONE all-reduce, whatever the mode: one
 * pack launch, one exchange, one apply launch (LFM_MERGE_ADAGRAD sends dW sqrt(G0 + dG / 2) next to dG and divides the
 * summed numerators by sqrt(G0 + sum dG / 2) when applying: the same step as the dense merge's two exchanges up to
 * float rounding).  Once a merge's union has covered >= 90 % of a side's rows (lfm_session_set_merge_dense_fraction) the
 * following merges of that side skip the detection, the all-reduce of the byte maps and the compaction and carry
 * every row: bit-identical (an untouched row's deltas are zeros), and what a small, fully touched table (ML-20M's
 * 26 744 item rows) costs is then three launches instead of sixteen and a host round trip. */
int lfm_session_comm_merge_sparse(lfm_session *s, int32_t sides, int32_t mode, int32_t overlap, int64_t *bytes);
int lfm_session_comm_merge_flush(lfm_session *s);
int lfm_session_set_merge_dense_fraction(lfm_session *s, float fraction);
/* HOT rows: feature rows that many interactions of every rank update (the tag / genre rows of a hybrid model:
 * 16 of the 19 rows an interaction of BASELINE config C3 touches are among its 1 128 tag rows) tolerate far
 * shorter merge intervals than rows one interaction in thousands touches (measured on one GPU with 8 emulated
 * ranks: C3 loses 0.027 precision@10 at the 8 Mi-interaction interval that suits ML-20M's identity rows, nothing
 * at 1 Mi).  lfm_session_set_hot_rows names them once (ascending feature rows of `side`, the same list on every
 * rank); lfm_session_comm_merge_hot then merges exactly these rows -- no detection, no union: one packed
 * all-reduce of n_rows x (2 d + 2) floats -- between the full merges, which include them anyway. */
int lfm_session_set_hot_rows(lfm_session *s, int32_t side, const int32_t *rows, int64_t n_rows);
int lfm_session_comm_merge_hot(lfm_session *s, int32_t sides, int32_t mode, int32_t overlap, int64_t *bytes);
/* Marks the current tables of `sides` as the start of a merge interval (lfm_session_comm_init
 * does it for the replicated sides; sessions merged with lfm_sessions_merge_local call it once
 * before training). */
int lfm_session_merge_begin(lfm_session *s, int32_t sides);
/* max over ranks of a flag (e.g. "my tables are not finite"), so that all ranks raise together */
int lfm_session_comm_any(lfm_session *s, int32_t flag);
int lfm_session_comm_barrier(lfm_session *s);
/* The same merge arithmetic for K sessions living in ONE process on ONE device (no RCCL): how the
 * multi-GPU semantics are measured on a single GPU (tools/multi_gpu_emulation.py) and tested. */
int lfm_sessions_merge_local(lfm_session **sessions, int32_t k, int32_t sides, int32_t mode);
/* ... and of the sparse merge; overlap != 0 defers the application of the merged deltas to the next call
 * (or to lfm_sessions_merge_local_flush): the one-segment delay of the overlapped multi-GPU exchange. */
int lfm_sessions_merge_local_sparse(lfm_session **sessions, int32_t k, int32_t sides, int32_t mode, int32_t overlap);
int lfm_sessions_merge_local_flush(lfm_session **sessions, int32_t k);
int lfm_sessions_merge_local_hot(lfm_session **sessions, int32_t k, int32_t sides, int32_t mode, int32_t overlap);

/* OWNER-SHARDED item tables, one-device form (tests / emulation; no reference counterpart).  For an item side too
 * large to replicate and merge -- BASELINE config C4: 2.6 GB of tables against 57 ms of kernels per epoch, every
 * rank touches nearly every row within one merge interval (DESIGN.md "Multi-GPU") -- the tables are cut into K
 * contiguous row ranges, range j lives with owner j, and every rank's kernels gather from and publish to the
 * OWNER's copy (global_atomic_add_f32 into the owner's memory): no replicas, no merges, plain Hogwild across all
 * ranks.  This call wires K training sessions of ONE device that way (session j owns rows [j * rps, (j + 1) * rps),
 * rps = ceil(n_items / K); the other rows of a session's own tables are then unused); in the multi-GPU form the
 * K base pointers are peer mappings of the owners' memory over xGMI.  Parallel WARP with identity features,
 * adagrad, no regularisation, d <= 64, max_sampled = 10 (the steady-state tile kernel); lfm_session_epoch fails
 * with LFM_EUNSUPPORTED otherwise. */
int lfm_sessions_share_items_local(lfm_session **sessions, int32_t k);

/* ... and the multi-PROCESS form (one process per GPU; several processes may also share one GPU): every process
 * exports the HIP IPC handles of its four item-side allocations (W, G, b, bG), the K exports travel over the job's
 * rendezvous (lightfm_amd/distributed.py hands them round with torch.distributed / gloo, like the RCCL unique id),
 * and lfm_session_share_items_ipc maps the other owners' allocations into this process (hipIpcOpenMemHandle: the
 * same memory on one GPU, peer mappings over xGMI across the GPUs of a node) and wires the session's kernels to
 * them exactly as lfm_sessions_share_items_local does.  all[my_rank] must be this process's own export.  The
 * mappings are closed by lfm_session_destroy; every process keeps its session alive until all have finished
 * training (a barrier of the rendezvous).  A sharded session's lfm_session_check_finite answers for the rows it
 * owns; lfm_session_gather_shared_items copies the other owners' rows into the session's own tables (device to
 * device) so that lfm_session_sync_to_host returns the whole model -- call it on every process between two
 * barriers, after the last epoch.  At least two item rows per owner; same kernel scope as the local form. */
#define LFM_IPC_HANDLE_BYTES 64
typedef struct lfm_item_export {
    char handle[4][LFM_IPC_HANDLE_BYTES]; /* hipIpcMemHandle_t of the allocations holding W, G, b, bG */
    int64_t offset[4];                    /* of the table inside its allocation (0 with the library's pool) */
    int64_t bytes[4];
    int32_t n_items, d, device, reserved;
    int64_t pid;
} lfm_item_export;
int lfm_session_export_items(lfm_session *s, lfm_item_export *out);
int lfm_session_share_items_ipc(lfm_session *s, const lfm_item_export *all, int32_t k, int32_t my_rank);
int lfm_session_gather_shared_items(lfm_session *s);

#ifdef __cplusplus
}
#endif
#endif /* LFM_HIP_H */
