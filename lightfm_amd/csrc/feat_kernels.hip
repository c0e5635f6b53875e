// feat_kernels.hip -- instantiations, LDS geometry and launcher of the pipelined row-stream epoch
// kernels (feat_kernel.hpp).
#include <stdlib.h>

#include "feat_kernel.hpp"

namespace lfm {

// LDS of one wavefront: [stage_rows][d] staging (LDS-DMA target) + [rr][d + 4] representations
// (+ 3 * pair_cap k-OS slots).  The budget per wavefront decides how many wavefronts a CU holds
// (160 KiB LDS): many rows in flight per wavefront against many wavefronts.
bool feat_plan(int loss, int d, int max_sampled, int n_positives, int first_batch, int rows_hint, FeatPlan *p, size_t budget_cap,
               bool few_atomics, bool no_shared_rows)
{
    if (d < 4 || d > 256 || (d & 3) != 0 || max_sampled < 0) return false;
    FeatPlan g;
    g.ts = d + 4;
    g.pair_cap = 0;
    int cb = std::max(1, std::min(max_sampled, 16));
    int kos_rows = 0;  // k-OS: tile rows the positives phase needs (the candidates reuse them)
    switch (loss) {
    case LFM_LOSS_LOGISTIC_ID: g.cand_base = 2; cb = 0; break;
    case LFM_LOSS_BPR_ID: g.cand_base = 3; cb = 0; break;
    case LFM_LOSS_WARP_ID: g.cand_base = 2; break;
    case LFM_LOSS_WARP_KOS_ID:
        if (n_positives < 1 || n_positives > WAVE - 1) return false;  // (one job per lane: the user + n sampled positives)
        // row 0 = the user, rows 1 .. n = the sampled positives; once the k-th is chosen (and kept in registers) the
        // candidates take the same rows: 1 + max(n, batch) tile rows instead of 1 + n + batch
        g.cand_base = 1;
        kos_rows = 1 + n_positives;
        g.pair_cap = ((n_positives + 3) / 4) * 4;
        break;
    default: return false;
    }
    static int budget_kb = -1, wpb_env = -1;
    if (budget_kb < 0) {
        const char *e = getenv("LIGHTFM_AMD_FEAT_LDS_KB");
        budget_kb = e ? atoi(e) : 0;
        const char *w = getenv("LIGHTFM_AMD_FEAT_WAVES_PER_BLOCK");
        wpb_env = w ? atoi(w) : 0;
    }
    // default budget per wavefront (8 wavefronts run per CU, session.hip): the representation tile
    // of the loss plus a staging area of >= 16 rows (C3 sweep: 20 staged rows at 8 wavefronts per CU
    // beat 38 rows at 7), at most 19 KiB
    const size_t tile_bytes = (size_t)std::max(g.cand_base + cb, kos_rows) * g.ts * 4 + 3 * (size_t)g.pair_cap * 4;
    const size_t floor_b = d > 64 ? 12 * 1024 : 6 * 1024;
    size_t budget = budget_kb > 0 ? (size_t)budget_kb * 1024
                                  : std::min<size_t>(19 * 1024, std::max(floor_b, tile_bytes + (size_t)16 * d * 4 + 256));
    // WARP / k-OS (not bound by the atomic unit): 12 wavefronts per CU when a twelfth of the LDS still stages >= 8 rows;
    // k-OS (held to 128 VGPRs, feat_kernel.hpp): 16 when a sixteenth still stages 6
    int waves_per_cu = 8, min_sr = 8;
    // (few_atomics: BPR / logistic whose shared rows are accumulated in LDS slices, session.hip: HotSet -- no longer bound by
    // the atomic unit, they take the twelve wavefronts too: C3 80.8 -> 83.5 M/s, profiles/r06_hot_overlap_ab.txt)
    // (no_shared_rows: BPR / logistic / WARP over identity features on both sides -- every row has few concurrent writers, the
    // atomic unit is not what bounds them: sixteen wavefronts too.  ML-20M shape, d = 64: logistic 327 -> 553 M interactions/s,
    // BPR 226 -> 378, profiles/r06_logistic_tile.txt)
    if (budget_kb <= 0 && (loss == LFM_LOSS_WARP_ID || loss == LFM_LOSS_WARP_KOS_ID || few_atomics || no_shared_rows)) {
        const size_t b12 = (size_t)(156 * 1024 / 12) & ~(size_t)255, b16 = (size_t)(160 * 1024 / 16);
        if ((loss == LFM_LOSS_WARP_KOS_ID || no_shared_rows) && b16 >= tile_bytes + 2 * WAVE * 4 + (size_t)6 * d * 4) {
            budget = b16;
            waves_per_cu = 16;
            min_sr = 6;
        } else if (b12 >= tile_bytes + 2 * WAVE * 4 + (size_t)8 * d * 4) {
            budget = b12;
            waves_per_cu = 12;
        }
    }
    // (budget_cap: a caller that needs LDS room beside these wavefronts -- the hot-slice kernel running under the next
    // launch, session.hip -- bounds the budget per wavefront)
    if (budget_cap > 0 && budget > budget_cap) {
        budget = budget_cap;
        min_sr = std::min(min_sr, 6);
    }
    g.waves_per_block = (wpb_env == 1 || wpb_env == 2 || wpb_env == 4) ? wpb_env : 1;
    const int want = std::min(64, std::max(8, 2 * (rows_hint + 1)));  // W and G rows of one update list
    for (;; --cb) {
        g.rr = std::max(g.cand_base + cb, kos_rows);
        const size_t fixed = (size_t)g.rr * g.ts * 4 + 3 * (size_t)g.pair_cap * 4 + 2 * WAVE * 4;
        if (fixed < budget) {
            int sr = (int)((budget - fixed) / ((size_t)d * 4));
            sr = std::min(sr, want) & ~1;
            if (sr >= min_sr || (sr >= 6 && budget_kb > 0) || (sr >= 4 && cb <= 1)) {
                g.sr = sr;
                break;
            }
        }
        if (cb <= 1) return false;
    }
    if (g.rr > WAVE) return false;
    // candidates whose representations are built speculatively in the first batch: 4 (C5 shard, 7.7 draws per interaction
    // on average: 57.2 M/s with all 10 in one batch, 59.4 with 4 + the rest -- an issue-bound kernel pays for every
    // representation it builds in vain; profiles/r04_visit_g.txt)
    g.first_batch = cb == 0 ? 1 : std::max(1, std::min(first_batch > 0 ? first_batch : std::min(max_sampled, 4), cb));
    g.smem = (size_t)g.waves_per_block * ((size_t)g.sr * d + (size_t)g.rr * g.ts + 3 * (size_t)g.pair_cap + 2 * WAVE) * 4;
    // residency (wavefronts per CU the session launches): the atomic-heavy losses publish fastest from 8 (C3: 43 M/s at
    // 2 048 interactions in flight against 35 M/s at 3 072); WARP / k-OS take what the LDS allows, up to 12 (C5 shard:
    // 43.3 -> 49.8 M/s from 8 to 12 once the reduce was cheap, profiles/r04_visit_f.txt)
    g.waves_per_cu = waves_per_cu;
    *p = g;
    return true;
}

hipError_t launch_fit_feat(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                           int *grid_used, bool timed)
{
    if (a.m.d <= 64) return launch_feat_nc<1>(loss, a, grid, block, smem, st, cus, grid_used, false);
    if (a.m.d <= 128) return launch_feat_nc<2>(loss, a, grid, block, smem, st, cus, grid_used, timed);
    if (a.m.d <= 256) return launch_fit_feat_wide(loss, a, grid, block, smem, st, cus, grid_used);  // feat_kernels_wide.hip
    return hipErrorInvalidValue;
}

}  // namespace lfm
