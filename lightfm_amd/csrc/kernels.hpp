// kernels.hpp -- host-callable launchers implemented in the .hip files.
#pragma once
#include <algorithm>
#include "device.hpp"

#include <map>
#include <mutex>
#include <tuple>

namespace lfm {

// Resident workgroups per CU of a kernel at (block, LDS bytes): the runtime's occupancy query, asked once
// per (kernel, block, smem) and remembered (an epoch may consist of thousands of launches).
template <typename K>
inline int occupancy_cached(K kernel, int block, size_t smem)
{
    static std::mutex mu;
    static std::map<std::tuple<const void *, int, size_t>, int> memo;
    const auto key = std::make_tuple((const void *)kernel, block, smem);
    std::lock_guard<std::mutex> lk(mu);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, smem) != hipSuccess) per_cu = 0;
    memo[key] = per_cu;
    return per_cu;
}

struct PredictArgs {
    DCsr itf, usf;
    DModel m;
    const int32_t *uids, *iids;
    float *out;
    int64_t n;
    int32_t tile_rows, tile_stride;
};

struct RanksArgs {
    const float *user_rep;  // [n_test_users_rows, rs] dense representations (bias at [d])
    const float *item_rep;  // [rs, n_items] component-major (coalesced across items)
    int32_t rs, d;
    DCsr test, train;
    float *ranks;           // aligned with test.data
    const int32_t *ulist;   // ranks_mfma_kernel: users that have test interactions
    int32_t n_ulist;
    float *item_eps;        // ranks_mfma2_kernel: [2][n_items] item-side terms of the pre-filter's error bound
    float *test_scores;     // ranks_mfma2_kernel: [test_nnz] exact scores of the test interactions
    int64_t test_nnz;
    const int32_t *work;    // ranks_mfma2_kernel: [n_work][2] (32-user tile of ulist, first test item of the pass);
                            // ranks_mfma3_kernel: [n_work][4] (tile, first test item, first item, end item of the segment)
    int32_t n_work;
    int32_t item_rows;      // ranks_mfma3_kernel: rows of item_rep (it reads the table through a buffer descriptor)
    const float *item_rows_rm;  // optional [n_items][rs] row-major item representations (bias at [d]): test_scores_kernel reads
                                // two contiguous rows per test interaction instead of d + 1 lines of the component-major table
    void *item_bf;              // optional, ranks_mfma3_kernel: room for the table of bf16 pieces (ranks_mfma3_bf_bytes) -- set: the
                                // products run on the bf16 matrix pipe (split operands); nullptr: fp32 products
};

// grid_used (optional): the grid actually launched (after the residency clamp)
hipError_t launch_fit(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st,
                      int cus = 0, int *grid_used = nullptr);
// warp_tile_bpr.hip: the BPR / logistic instantiations of the lane-group tile kernel (identity features, adagrad, no L2 penalty; vec = 4)
hipError_t launch_fit_bpr_wide_tile(const FitArgs &a, int ng, int vec, int grid, size_t smem, hipStream_t st, int cus,
                                    int *grid_used, bool dma4, bool logistic);
// fit_kernels_wide.hip: the same kernels for 512 < d <= LFM_MAX_COMPONENTS
hipError_t launch_fit_wide(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus, int *grid_used);
// warp_tile.hip: lane-group tile kernel (identity features, alpha == 0, parallel mode)
// ng = interactions per wavefront pass (1, 2, 4); vec = floats of a row per lane
// dma4: the LDS-DMA (global_load_lds_dwordx4) variant of ng = 4 with its candidate-major tile
size_t warp_tile_geometry(int d, int max_sampled, int ng, int *rows, int *stride, int *vec, bool dma4 = false);
hipError_t launch_fit_warp_tile(const FitArgs &a, int ng, int vec, int grid, size_t smem, hipStream_t st,
                                int cus, bool timed = false, int *grid_used = nullptr, bool dma4 = false);
// warp_tile_ahead.hip: the steady-state variant of the four-per-pass LDS-DMA tile kernel with the gather of the next
// pass issued inside the current one (warp_tile_ahead.hpp); smem = 0: outside its scope
size_t warp_tile_ahead_smem(int d, int max_sampled, int first_batch);
hipError_t launch_fit_warp_tile_ahead(const FitArgs &a, int grid, hipStream_t st, int cus, int *grid_used = nullptr);
// warp_tile_narrow.hpp: the same for rows of <= 16 floats, two interactions per lane group (eight per wavefront pass);
// per_cu_cap > 0 bounds the workgroups per CU
size_t warp_tile_narrow_smem(int d, int max_sampled, int first_batch, int64_t n_items);
hipError_t launch_fit_warp_tile_narrow(const FitArgs &a, int grid, hipStream_t st, int cus, int per_cu_cap, int *grid_used = nullptr);
// logistic_tile.hip: fit_logistic for narrow identity models (d <= 12: the reference's default LightFM()) on the one-line-per-feature
// rows (FitArgs::rp with the bias cells); 0 = outside its scope, else the LDS bytes of a workgroup
size_t logistic_tile_smem(int d, int64_t n_users, int64_t n_items);
hipError_t launch_fit_logistic_tile(const FitArgs &a, int grid, hipStream_t st, int cus, int *grid_used = nullptr);
// ... and fit_bpr for the same models (needs FitArgs::pos, item_ids, seeds)
size_t bpr_tile_smem(int d, int64_t n_users, int64_t n_items);
int bpr_tile_per_wave();  // interactions of a wavefront pass
hipError_t launch_fit_bpr_tile(const FitArgs &a, int grid, hipStream_t st, int cus, int *grid_used = nullptr);
// feat_kernels.hip: pipelined row-stream kernels (feature CSRs, BPR, k-OS, logistic; feat_kernel.hpp)
struct FeatPlan {
    int rr, ts, sr, cand_base, pair_cap, first_batch;  // tile rows / stride, stage rows, ...
    int waves_per_block;
    int waves_per_cu = 8;  // wavefronts the session keeps resident per CU (session.hip)
    size_t smem;  // LDS bytes per workgroup
};
// rows_hint: rows of a typical update list (f_user + 2 f_item), so that one chunk covers it
// no_shared_rows: identity features on both sides -- no row has many concurrent writers, the atomic unit is not the bound:
// sixteen wavefronts per CU where a sixteenth of the LDS holds the plan
bool feat_plan(int loss, int d, int max_sampled, int n_positives, int first_batch, int rows_hint, FeatPlan *p, size_t budget_cap = 0,
               bool few_atomics = false, bool no_shared_rows = false);
hipError_t launch_fit_feat(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                           int *grid_used = nullptr, bool timed = false);
// feat_kernels_ada.hip: the adadelta instantiations of the row-stream kernels (d <= 128)
hipError_t launch_fit_feat_ada(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                               int *grid_used = nullptr);
// feat_kernels_hot.hip: the HOT instantiations of the row-stream kernels (a model with a hot set: FitArgs::hot_slot)
hipError_t launch_fit_feat_hot(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                               int *grid_used = nullptr);
// hot_slices.hip: the hot rows' records of one launch applied to component slices held in LDS (device.hpp: HotRec)
struct HotArgs {
    const HotRec *rec;   // [n_rec] the launch's records
    const float *x;      // [n_rec][d]
    int64_t n_rec;
    int32_t d, hot_n, n_rep;
    const int32_t *rows;                    // [hot_n] feature row of a slot
    float *W, *G, *b, *bG;                  // the live item-side tables
    float *snapW, *snapG, *snapb, *snapbG;  // [hot_n][d] / [hot_n]: the hot rows as they were when the records' launch ended
    float lr, rho, eps;
};
int hot_slice_components(int hot_n, int d);  // components per LDS slice (8, 4, 2), 0 = the hot set does not fit
hipError_t launch_hot_slices(const HotArgs &a, int cs, int threads, hipStream_t st);
hipError_t launch_selftest_adagrad(int64_t n, uint32_t seed, float lr, unsigned long long *out, hipStream_t st);
// predict_kernels.hip: the bf16-split sweep's scores against the sequential dot, as a fraction of its rounding band (out[0] = largest
// fraction, float bits; out[1] = pairs beyond the band)
hipError_t launch_ranks_bf_band_selftest(int64_t tiles, uint32_t seed, int d, int spread, unsigned *out, hipStream_t st);
hipError_t launch_column_counts(const int32_t *indices, int64_t nnz, int32_t cols, int32_t *counts, hipStream_t st);
// csr_build.hip: the Bloom filter over the positives lookup (device.hpp: Bloom); bloom has Bloom::words(nnz) words
hipError_t build_positives_bloom(const int32_t *indptr, const int32_t *indices, int32_t n_rows, int64_t nnz, uint32_t *bloom,
                                 hipStream_t st);
hipError_t launch_pack_records(const int32_t *user_ids, const int32_t *item_ids, const float *Y,
                               const float *weight, int64_t n, void *out, hipStream_t st);
// lazy L2 regularisation in parallel mode (device.hpp: RegScale): reg_log[2] float64 totals at the last launch
// boundary, reg_live[4] the live state; both nullptr in serial mode (m.scales)
hipError_t launch_reg_log_init(const double *scales, double *reg_log, float *reg_live, hipStream_t st);
hipError_t launch_regularize(const DModel &m, double *reg_log, float *reg_live, int force, hipStream_t st, int64_t positions = 0);
hipError_t launch_nonfinite(const float *x, int64_t n, int *flag, hipStream_t st);

hipError_t launch_predict(const PredictArgs &a, int grid, size_t smem, hipStream_t st);
// dense representation table of every row of f (PYX:287-317), out[row*rs + 0..d]
// transposed = 1: component-major, out[c*f.rows + row]
// bias_out != nullptr: biases go to bias_out[row] instead of out[row*rs + d] (rs may then be d)
hipError_t launch_rep_rows(const DCsr &f, const float *W, const float *b, int d, int rs,
                           float *out, hipStream_t st, int transposed = 0, float *bias_out = nullptr);
// csr_build.hip: positives lookup CSR (sorted, duplicate-free) from the COO, on device
hipError_t build_positives_csr(const int32_t *user_ids, const int32_t *item_ids, int64_t n, int32_t n_users,
                               int32_t n_items, int32_t *indices_out, int32_t *indptr_out, int64_t *nnz_out,
                               hipStream_t st);
// csr_build.hip: ascending positions of the non-zero bytes of flags[0 .. n) (synchronises the stream)
hipError_t compact_flagged_rows(const unsigned char *flags, int64_t n, int32_t *ids_out, int64_t *count, hipStream_t st);
hipError_t launch_ranks(const RanksArgs &a, hipStream_t st);
// MFMA pre-filtered variant (d <= 128): false if the shape is outside what it supports
bool ranks_mfma_supported(int d);
hipError_t launch_ranks_mfma(const RanksArgs &a, hipStream_t st, int cus);
// users as tile columns (a lane owns a user for the whole sweep); ulist ordered by test count, largest first
hipError_t launch_ranks_mfma2(const RanksArgs &a, hipStream_t st, int cus);
int ranks_mfma2_item_rows(int d);  // rows the component-major item table must have for it
// the bucket-search sweep (the default): work items of four ints (tile, first test item of the pass, item segment)
hipError_t launch_ranks_mfma3(const RanksArgs &a, hipStream_t st, int cus);
int ranks_mfma3_waves_per_cu(int d);
int ranks_mfma3_pass_items();
bool ranks_mfma3_supported(int d, int64_t n_items, int item_rows);  // the table must fit one 2 GB buffer
size_t ranks_mfma3_bf_bytes(int d, int64_t n_items);  // RanksArgs::item_bf: bytes of the table of bf16 pieces (0: not available)
hipError_t launch_auc(const DCsr &ranks, const int32_t *num_train_positives, float *rank_data,
                      float *auc, hipStream_t st);

}  // namespace lfm
