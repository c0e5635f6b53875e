// fit_kernels.hip -- epoch kernels for the four LightFM losses on gfx950.
//
// fit_warp      PYX:784-912      fit_bpr       PYX:1074-1182
// fit_logistic  PYX:694-781      fit_warp_kos  PYX:915-1071
// (PYX = /root/reference/lightfm/_lightfm_fast.pyx.template)
//
// Grid: 256-thread workgroups = 4 independent wavefronts, each looping over
// shuffled positions begin+gw, begin+gw+nwaves, ...  Serial (parity) mode is the
// same code launched as a single wavefront that walks [begin, end) in order.
#include "device.hpp"
#include "kernels.hpp"

namespace lfm {

namespace {

struct WaveCtx {
    int lane;
    float *tile;   // wave-private LDS: tile_rows x tile_stride floats
    int TS;
    unsigned long long c0, c1, c2, c3;
};

__device__ __forceinline__ void log_pos(const FitArgs &a, int64_t i, int neg, int sampled, int lane)
{
    if (lane == 0) {
        if (a.neg_log) a.neg_log[i] = neg;
        if (a.sampled_log) a.sampled_log[i] = sampled;
    }
}

// regularize (PYX:652-675) executed by ONE wavefront (serial mode, mid-epoch).
__device__ void regularize_inline(const FitArgs &a, Scales &sc, int lane)
{
    for (int side = 0; side < 2; ++side) {
        double s = side == 0 ? sc.item : sc.user;
        int64_t nW = (int64_t)a.m.n_feat[side] * a.m.d;
        for (int64_t j = lane; j < nW; j += WAVE)
            a.m.W[side][j] = (float)((double)a.m.W[side][j] / s);
        for (int64_t j = lane; j < a.m.n_feat[side]; j += WAVE)
            a.m.b[side][j] = (float)((double)a.m.b[side][j] / s);
    }
    sc.item = 1.0;
    sc.user = 1.0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
}

__device__ __forceinline__ void after_example(const FitArgs &a, Scales &sc, int lane)
{
    if (a.serial) {
        // make this interaction's stores visible to the next one's loads
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        if (sc.item > MAX_REG_SCALE || sc.user > MAX_REG_SCALE) regularize_inline(a, sc, lane);  // PYX:901-904
    }
}

__device__ __forceinline__ void wave_begin(const FitArgs &a, WaveCtx &w, Scales &sc, float *smem)
{
    w.lane = lane_id();
    int wib = threadIdx.x >> 6;
    w.TS = a.tile_stride;
    w.tile = smem + (size_t)wib * a.tile_rows * a.tile_stride;
    w.c0 = w.c1 = w.c2 = w.c3 = 0;
    sc.item = a.m.scales[0];  // serial mode; parallel mode refreshes per interaction (device.hpp: RegScale)
    sc.user = a.m.scales[1];
    sc.live.begin();  // device.hpp: RegScale
}

__device__ __forceinline__ void wave_end(const FitArgs &a, WaveCtx &w, Scales &sc)
{
    if (reg_active(a)) RegScale::publish(a.reg_live, sc.live.p_i, sc.live.p_u, w.lane, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (w.lane == 0) {
        if (w.c0) atomicAdd(a.counters + 0, w.c0);
        if (w.c1) atomicAdd(a.counters + 1, w.c1);
        if (w.c2) atomicAdd(a.counters + 2, w.c2);
        if (w.c3) atomicAdd(a.counters + 3, w.c3);
        if (a.serial) {
            a.m.scales[0] = sc.item;
            a.m.scales[1] = sc.user;
        }
    }
}

// The WARP negative-sampling loop + update shared by fit_warp and fit_warp_kos
// (PYX:855-899 / 1014-1057).  Tile row 0 = user, row 1 = positive item, rows 2..
// = candidate negatives.  Negatives are drawn and scored `nb` at a time: within one
// interaction the weights do not change between draws, so taking the first
// violator of a batch is identical to the sequential loop; the PRNG is advanced by
// exactly the number of draws the sequential loop would have made.
template <int NC, bool FAST>
__device__ __forceinline__ void warp_negatives(const FitArgs &a, WaveCtx &w, Scales &sc, int64_t i,
                                               int user, int pos, double pp, float weight,
                                               bool kos, uint32_t &state, const Rep<NC> &U,
                                               const Rep<NC> &P, int pos_lo, int pos_hi)
{
    const int lane = w.lane, d = a.m.d, TS = w.TS;
    const int max_sampled = a.m.max_sampled;
    const int n_items = a.itf.rows;
    int sampled = 0, chosen = -1, chosen_slot = -1;
    while (sampled < max_sampled && chosen < 0) {
        int cap = (sampled == 0) ? a.first_batch : (a.tile_rows - 2);
        int nb = min(max_sampled - sampled, cap);
        // lane k holds the state after k+1 steps and the (sampled+k+1)-th draw
        uint32_t s = state;
        for (int j = 0; j < nb; ++j)
            if (j <= lane) s = lcg(s);
        int myneg = (int)(draw(s) % (uint32_t)n_items);  // PYX:860-861
        for (int k = 0; k < nb; ++k) {
            int neg = read_lane(myneg, k);
            Rep<NC> N;
            load_rep<NC, FAST>(a.itf, a.m.W[0], a.m.b[0], d, neg, sc.item, lane, N);
            rep_to_tile<NC>(w.tile + (size_t)(2 + k) * TS, N, d, lane);
        }
        wave_sync();
        float score = 0.0f;
        bool mine = (lane >= 2) && (lane < 2 + nb);
        if (mine) score = tile_dot(w.tile, w.tile + (size_t)lane * TS, d);
        // PYX:875 compares doubles: negative_prediction > positive_prediction - 1
        unsigned long long mask = __ballot(mine && ((double)score > pp - 1.0));
        int used = nb;
        while (mask) {
            int slot = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            int neg = read_lane(myneg, slot - 2);
            w.c3++;
            if (in_positives_range(a.pos, neg, pos_lo, pos_hi, lane)) continue;  // PYX:878-879, draw counted
            chosen = neg;
            chosen_slot = slot;
            used = slot - 1;
            break;
        }
        sampled += used;
        state = (uint32_t)read_lane((int)s, used - 1);
        wave_sync();
    }
    w.c1 += (unsigned long long)sampled;
    if (chosen >= 0) {
        // PYX:881-885 (k-OS: PYX:1039-1043, no weight, no max(1, .)); log table from host libm
        double loss = kos ? a.logtab[sampled] : (double)weight * a.logtab[sampled];
        if (loss > MAX_LOSS) loss = MAX_LOSS;
        Rep<NC> N;
        rep_from_tile<NC>(w.tile + (size_t)chosen_slot * TS, d, lane, N);
        wave_sync();
        if constexpr (FAST)
            warp_update_identity<NC>(loss, a, user, pos, chosen, U, P, N, lane);
        else
            warp_update<NC>(loss, a, user, pos, chosen, U, P, N, sc, lane);
        w.c2++;
    }
    log_pos(a, i, chosen, sampled, lane);
}

}  // namespace

// ------------------------------------------------------------------ WARP ---

// One interaction of fit_warp (PYX:826-904); its COO fields arrive prefetched.
template <int NC, bool FAST>
__device__ __forceinline__ void warp_example(const FitArgs &a, WaveCtx &w, Scales &sc, int64_t i,
                                             int user, int pos, float y, float weight,
                                             uint32_t &state, uint32_t base_seed)
{
    const int lane = w.lane, d = a.m.d, TS = w.TS;
    if (!(y > 0.0f)) {  // PYX:831-832, before any RNG use
        log_pos(a, i, -1, 0, lane);
        return;
    }
    if (FAST || !a.serial) state = position_seed(base_seed, (uint64_t)i);
    if constexpr (!FAST) refresh_scales(a, sc, i);
    w.c0++;
    // issued together with the row gathers; consumed by in_positives after the dots
    int pos_lo = a.pos.indptr[user], pos_hi = a.pos.indptr[user + 1];
    Rep<NC> U, P;
    load_rep<NC, FAST>(a.usf, a.m.W[1], a.m.b[1], d, user, sc.user, lane, U);
    load_rep<NC, FAST>(a.itf, a.m.W[0], a.m.b[0], d, pos, sc.item, lane, P);
    rep_to_tile<NC>(w.tile, U, d, lane);
    rep_to_tile<NC>(w.tile + TS, P, d, lane);
    wave_sync();
    float ps = 0.0f;
    if (lane == 1) ps = tile_dot(w.tile, w.tile + TS, d);
    double pp = (double)read_lanef(ps, 1);
    warp_negatives<NC, FAST>(a, w, sc, i, user, pos, pp, weight, false, state, U, P, uni(pos_lo),
                             uni(pos_hi));
    if constexpr (!FAST) after_example(a, sc, lane);
}

// FAST: parallel mode, identity features on both sides, no regularisation (the
// representations ARE embedding rows; BASELINE configs C2/C4).  Otherwise generic.
template <int NC, bool FAST, int OCC>
__global__ __launch_bounds__(256, OCC) void fit_warp_kernel(FitArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx w;
    Scales sc;
    wave_begin(a, w, sc, smem);
    const int lane = w.lane;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    uint32_t state = (!FAST && a.serial) ? a.seeds[a.seed_idx] : 0u;
    const uint32_t base_seed = a.seeds[0];
    // two-deep software pipeline over the COO: while interaction i is processed, the
    // (user, item, y, weight) of i+nw and the shuffle entry of i+2nw are in flight
    int64_t i = a.begin + gw;
    int row1 = 0, c_user = 0, c_pos = 0;
    float c_y = 0.0f, c_w = 0.0f;
    if (i < a.end) {
        int row0 = guard_row(a, a.shuffle[i]);
        c_user = a.user_ids[row0];
        c_pos = a.item_ids[row0];
        c_y = a.Y[row0];
        c_w = a.weight[row0];
    }
    if (i + nw < a.end) row1 = a.shuffle[i + nw];
    for (; i < a.end; i += nw) {
        int row2 = 0, n_user = 0, n_pos = 0;
        float n_y = 0.0f, n_w = 0.0f;
        if (i + 2 * nw < a.end) row2 = a.shuffle[i + 2 * nw];
        if (i + nw < a.end) {
            row1 = guard_row(a, row1);
            n_user = a.user_ids[row1];
            n_pos = a.item_ids[row1];
            n_y = a.Y[row1];
            n_w = a.weight[row1];
        }
        warp_example<NC, FAST>(a, w, sc, i, uni(c_user), uni(c_pos), unif(c_y), unif(c_w), state,
                               base_seed);
        row1 = row2;
        c_user = n_user;
        c_pos = n_pos;
        c_y = n_y;
        c_w = n_w;
    }
    if (!FAST && a.serial && lane == 0) const_cast<uint32_t *>(a.seeds)[a.seed_idx] = state;
    wave_end(a, w, sc);
}

// ------------------------------------------------------------------- BPR ---

template <int NC>
__global__ __launch_bounds__(256) void fit_bpr_kernel(FitArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx w;
    Scales sc;
    wave_begin(a, w, sc, smem);
    const int lane = w.lane, d = a.m.d, TS = w.TS;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    uint32_t state = a.serial ? a.seeds[a.seed_idx] : 0u;
    const uint32_t base_seed = a.seeds[0];
    const uint32_t n_examples = (uint32_t)a.n;
    for (int64_t i = a.begin + gw; i < a.end; i += nw) {
        int row = guard_row(a, uni(a.shuffle[i]));
        if (!(a.Y[row] > 0.0f)) {  // PYX:1116-1117
            log_pos(a, i, -1, 0, lane);
            continue;
        }
        float weight = unif(a.weight[row]);
        int user = uni(a.user_ids[row]), pos = uni(a.item_ids[row]);
        if (!a.serial) state = position_seed(base_seed, (uint64_t)i);
        refresh_scales(a, sc, i);
        w.c0++;
        int neg = 0, draws = 0;
        for (int64_t j = 0; j < a.n; ++j) {  // PYX:1123-1127
            state = lcg(state);
            neg = uni(a.item_ids[draw(state) % n_examples]);
            draws++;
            w.c3++;
            if (!in_positives(a.pos, neg, user, lane)) break;
        }
        w.c1 += (unsigned long long)draws;
        Rep<NC> U, P, N;
        load_rep<NC>(a.usf, a.m.W[1], a.m.b[1], d, user, sc.user, lane, U);
        load_rep<NC>(a.itf, a.m.W[0], a.m.b[0], d, pos, sc.item, lane, P);
        load_rep<NC>(a.itf, a.m.W[0], a.m.b[0], d, neg, sc.item, lane, N);
        rep_to_tile<NC>(w.tile, U, d, lane);
        rep_to_tile<NC>(w.tile + TS, P, d, lane);
        rep_to_tile<NC>(w.tile + 2 * TS, N, d, lane);
        wave_sync();
        float s = 0.0f;
        if (lane == 1 || lane == 2) s = tile_dot(w.tile, w.tile + (size_t)lane * TS, d);
        double pp = (double)read_lanef(s, 1), np_ = (double)read_lanef(s, 2);
        wave_sync();
        // PYX:1158: weight * (1 - sigmoid(pp - np)); the difference is narrowed to float32
        double loss = (double)weight * (1.0 - (double)sigmoidf_ref((float)(pp - np_)));
        warp_update<NC>(loss, a, user, pos, neg, U, P, N, sc, lane);
        w.c2++;
        log_pos(a, i, neg, draws, lane);
        after_example(a, sc, lane);
    }
    if (a.serial && lane == 0) const_cast<uint32_t *>(a.seeds)[a.seed_idx] = state;
    wave_end(a, w, sc);
}

// -------------------------------------------------------------- logistic ---

template <int NC>
__global__ __launch_bounds__(256) void fit_logistic_kernel(FitArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx w;
    Scales sc;
    wave_begin(a, w, sc, smem);
    const int lane = w.lane, d = a.m.d, TS = w.TS;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t i = a.begin + gw; i < a.end; i += nw) {
        int row = guard_row(a, uni(a.shuffle[i]));
        int user = uni(a.user_ids[row]), item = uni(a.item_ids[row]);
        float weight = unif(a.weight[row]);
        refresh_scales(a, sc, i);
        Rep<NC> U, I;
        load_rep<NC>(a.usf, a.m.W[1], a.m.b[1], d, user, sc.user, lane, U);
        load_rep<NC>(a.itf, a.m.W[0], a.m.b[0], d, item, sc.item, lane, I);
        rep_to_tile<NC>(w.tile, U, d, lane);
        rep_to_tile<NC>(w.tile + TS, I, d, lane);
        wave_sync();
        float s = 0.0f;
        if (lane == 1) s = tile_dot(w.tile, w.tile + TS, d);
        s = read_lanef(s, 1);
        wave_sync();
        double prediction = (double)sigmoidf_ref(s);  // PYX:745-747
        int y = (a.Y[row] <= 0.0f) ? 0 : 1;           // PYX:751-755
        if (y) w.c0++;
        double loss = (double)weight * (prediction - (double)y);
        pair_update<NC>(loss, a, user, item, U, I, sc, lane);
        w.c2++;
        after_example(a, sc, lane);
    }
    wave_end(a, w, sc);
}

// ------------------------------------------------------------------ k-OS ---

template <int NC>
__global__ __launch_bounds__(256) void fit_warp_kos_kernel(FitArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx w;
    Scales sc;
    wave_begin(a, w, sc, smem);
    const int lane = w.lane, d = a.m.d, TS = w.TS;
    const int wib = threadIdx.x >> 6;
    // (idx, val) pairs of the sampled positives, PYX:109-111, after the tiles
    float *pair_base = smem + (size_t)WAVES_PER_BLOCK * a.tile_rows * a.tile_stride +
                       (size_t)wib * 2 * a.pair_cap;
    int *pair_idx = reinterpret_cast<int *>(pair_base);
    float *pair_val = pair_base + a.pair_cap;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    uint32_t state = a.serial ? a.seeds[a.seed_idx] : 0u;
    const uint32_t base_seed = a.seeds[0];
    for (int64_t i = a.begin + gw; i < a.end; i += nw) {
        int row = guard_row(a, uni(a.shuffle[i]));
        int user = uni(a.user_ids[row]);
        if (!a.serial) state = position_seed(base_seed, (uint64_t)i);
        int start = uni(a.pos.indptr[user]), stop = uni(a.pos.indptr[user + 1]);
        if (stop == start) {  // PYX:971-972
            log_pos(a, i, -1, 0, lane);
            continue;
        }
        w.c0++;
        refresh_scales(a, sc, i);
        Rep<NC> U, P;
        load_rep<NC>(a.usf, a.m.W[1], a.m.b[1], d, user, sc.user, lane, U);
        rep_to_tile<NC>(w.tile, U, d, lane);
        int no_pos = min(a.n_pos, stop - start);  // PYX:975
        const int per = a.tile_rows - 1;
        for (int j0 = 0; j0 < no_pos; j0 += per) {
            int nb = min(per, no_pos - j0);
            int myit = 0;
            for (int jj = 0; jj < nb; ++jj) {
                state = lcg(state);  // sample_range, PYX:84-90
                int it = uni(a.pos.indices[start + (int)(draw(state) % (uint32_t)(stop - start))]);
                if (lane == 1 + jj) myit = it;
                Rep<NC> C;
                load_rep<NC>(a.itf, a.m.W[0], a.m.b[0], d, it, sc.item, lane, C);
                rep_to_tile<NC>(w.tile + (size_t)(1 + jj) * TS, C, d, lane);
            }
            wave_sync();
            if (lane >= 1 && lane <= nb) {
                pair_idx[j0 + lane - 1] = myit;
                pair_val[j0 + lane - 1] = tile_dot(w.tile, w.tile + (size_t)lane * TS, d);
            }
            wave_sync();
        }
        // qsort(reverse_pair_compare), PYX:997: stable descending insertion sort
        if (lane == 0) {
            for (int x = 1; x < no_pos; ++x) {
                int ki = pair_idx[x];
                float kv = pair_val[x];
                int y = x - 1;
                while (y >= 0 && (pair_val[y] - kv) < 0.0f) {
                    pair_idx[y + 1] = pair_idx[y];
                    pair_val[y + 1] = pair_val[y];
                    --y;
                }
                pair_idx[y + 1] = ki;
                pair_val[y + 1] = kv;
            }
        }
        wave_sync();
        int kk = min(a.k, no_pos) - 1;  // PYX:1002-1003
        int pos = uni(pair_idx[kk]);
        double pp = (double)unif(pair_val[kk]);
        wave_sync();
        load_rep<NC>(a.itf, a.m.W[0], a.m.b[0], d, pos, sc.item, lane, P);
        rep_to_tile<NC>(w.tile + TS, P, d, lane);
        wave_sync();
        warp_negatives<NC, false>(a, w, sc, i, user, pos, pp, 1.0f, true, state, U, P, start, stop);
        after_example(a, sc, lane);
    }
    if (a.serial && lane == 0) const_cast<uint32_t *>(a.seeds)[a.seed_idx] = state;
    wave_end(a, w, sc);
}

// ---------------------------------------------------- lazy regularisation ---

#ifndef LFM_FIT_WIDE_UNIT
// Parallel mode (device.hpp: RegScale): reg_log[2] = log(item_scale), log(user_scale) at the last launch
// boundary (float64 running totals); reg_live = line 0 {the same as float32, the growth of the logs per
// position measured over the last launch}, lines 1.. {slots collecting the growth of the logs since then}.
// Serial mode and the host see m.scales[2].
// Start of a parallel epoch: reg_log := log(scales), slots := 0 (the measured rates stay).
__global__ void reg_log_init_kernel(const double *scales, double *reg_log, float *reg_live)
{
    const int t = threadIdx.x;
    if (blockIdx.x != 0) return;
    if (t < 2) {
        reg_log[t] = log(scales[t]);
        reg_live[t] = (float)log(scales[t]);
    }
    if (t < RegScale::SLOTS) {
        reg_live[RegScale::LINE * (1 + t) + 0] = 0.0f;
        reg_live[RegScale::LINE * (1 + t) + 1] = 0.0f;
    }
}

// growth of log(scale) of `side` collected by the launch that just ended
__device__ __forceinline__ double reg_slots_sum(const float *reg_live, int side)
{
    double acc = 0.0;
    for (int s = 0; s < RegScale::SLOTS; ++s) acc += (double)reg_live[RegScale::LINE * (1 + s) + side];
    return acc;
}

// What a launch boundary folds (device.hpp: RegScale, FOLDS): `force` (end of the epoch, PYX:910-912) folds
// the whole scale of both sides; otherwise a side that has crossed log(MAX_REG_SCALE) nf times folds
// exp(nf LMAX) and keeps the remainder, and -- locked_regularize regularises BOTH sides when either
// scale has passed the bound (PYX:686-689) -- the other side folds its whole scale with it.
// fold[side] = log of the divisor, rest[side] = log-scale afterwards.
__device__ __forceinline__ void reg_fold_plan(const double l[2], int force, double fold[2], double rest[2])
{
    const double LMAX = log(MAX_REG_SCALE);
    double nf[2];
    for (int k = 0; k < 2; ++k) nf[k] = l[k] > LMAX ? floor(l[k] / LMAX) : 0.0;
    const bool any = nf[0] > 0.0 || nf[1] > 0.0;
    for (int k = 0; k < 2; ++k) {
        if (force || (any && nf[k] == 0.0)) fold[k] = l[k];
        else fold[k] = nf[k] * LMAX;
        rest[k] = l[k] - fold[k];
    }
}

// regularize (PYX:652-675).  reg_log != nullptr: parallel mode between two launches (the plan above),
// else serial mode: m.scales, folded whole when `force` or past the bound (PYX:678-691).
__global__ void regularize_kernel(DModel m, const double *reg_log, const float *reg_live, int force)
{
    double div[2];
    if (reg_log) {
        const double l[2] = {reg_log[0] + reg_slots_sum(reg_live, 0), reg_log[1] + reg_slots_sum(reg_live, 1)};
        double fold[2], rest[2];
        reg_fold_plan(l, force, fold, rest);
        div[0] = exp(fold[0]);
        div[1] = exp(fold[1]);
    } else {
        div[0] = m.scales[0];
        div[1] = m.scales[1];
        if (!force && !(div[0] > MAX_REG_SCALE || div[1] > MAX_REG_SCALE)) return;
    }
    if (div[0] == 1.0 && div[1] == 1.0) return;  // x / 1.0 == x bit for bit
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int side = 0; side < 2; ++side) {
        const double s = div[side];
        if (s == 1.0) continue;
        int64_t nW = (int64_t)m.n_feat[side] * m.d;
        for (int64_t j = t; j < nW; j += stride) m.W[side][j] = (float)((double)m.W[side][j] / s);
        for (int64_t j = t; j < m.n_feat[side]; j += stride)
            m.b[side][j] = (float)((double)m.b[side][j] / s);
    }
}

// After regularize_kernel: the launch's growth moves into the running totals, minus what was folded.
__global__ void reg_boundary_kernel(double *scales, double *reg_log, float *reg_live, int force, int64_t positions)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (!reg_log) {  // serial mode
            if (force || scales[0] > MAX_REG_SCALE || scales[1] > MAX_REG_SCALE) {
                scales[0] = 1.0;
                scales[1] = 1.0;
            }
            return;
        }
        const double d0 = reg_slots_sum(reg_live, 0), d1 = reg_slots_sum(reg_live, 1);
        const double l[2] = {reg_log[0] + d0, reg_log[1] + d1};
        double fold[2], rest[2];
        reg_fold_plan(l, force, fold, rest);
        if (positions >= 256) {  // the rate the next launch's readers extrapolate with (device.hpp: RegScale)
            reg_live[2] = (float)(d0 / (double)positions);
            reg_live[3] = (float)(d1 / (double)positions);
        }
        for (int k = 0; k < 2; ++k) {
            reg_log[k] = rest[k];
            reg_live[k] = (float)rest[k];
            scales[k] = exp(rest[k]);  // 1.0 after the end-of-epoch fold: what the host and serial mode see
        }
        for (int s = 0; s < RegScale::SLOTS; ++s) {
            reg_live[RegScale::LINE * (1 + s) + 0] = 0.0f;
            reg_live[RegScale::LINE * (1 + s) + 1] = 0.0f;
        }
    }
}

// isfinite(sum(x)) of LFM:447-464 as "any non-finite element" per array
__global__ void nonfinite_kernel(const float *x, int64_t n, int *flag)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int bad = 0;
    for (int64_t j = t; j < n; j += stride) bad |= !isfinite(x[j]);
    if (bad) atomicOr(flag, 1);
}

// ---------------------------------------------------------------- launch ---

#endif  // LFM_FIT_WIDE_UNIT

// Launch with the grid capped at what is actually resident (blocks/CU from the
// occupancy query x CUs): every wavefront then runs its grid-stride loop from the
// start of the launch instead of queueing behind a first wave of blocks.
template <typename K>
static hipError_t launch_resident(K kernel, const FitArgs &a, int grid, int block, size_t smem,
                                  hipStream_t st, int cus, int *grid_used)
{
    if (cus > 0 && block == 256) {
        const int per_cu = occupancy_cached(kernel, block, smem);
        if (per_cu > 0) grid = std::min(grid, per_cu * cus);
    }
    if (grid_used) *grid_used = grid;
    if (smem > 64 * 1024) {  // (rows wider than 1 024 floats, long k-OS pair buffers: beyond the default dynamic LDS bound)
        const hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    kernel<<<grid, block, smem, st>>>(a);
    return hipGetLastError();
}

template <int NC>
static hipError_t launch_nc(int loss, const FitArgs &a, int grid, int block, size_t smem,
                            hipStream_t st, int cus, int *grid_used)
{
    switch (loss) {
    case 0: return launch_resident(fit_logistic_kernel<NC>, a, grid, block, smem, st, cus, grid_used);
    case 1:
        if (!a.serial && a.itf.identity && a.usf.identity && a.item_alpha == 0.0 && a.user_alpha == 0.0) {
            return launch_resident(fit_warp_kernel<NC, true, 1>, a, grid, block, smem, st, cus, grid_used);
        }
        return launch_resident(fit_warp_kernel<NC, false, 1>, a, grid, block, smem, st, cus, grid_used);
    case 2: return launch_resident(fit_bpr_kernel<NC>, a, grid, block, smem, st, cus, grid_used);
    case 3: return launch_resident(fit_warp_kos_kernel<NC>, a, grid, block, smem, st, cus, grid_used);
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

#ifdef LFM_FIT_WIDE_UNIT
// fit_kernels_wide.hip: 512 < d <= LFM_MAX_COMPONENTS (the reference has no bound on no_components, PYX:185-259: a lane keeps
// 16 coordinates of a row)
hipError_t launch_fit_wide(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus, int *grid_used)
{
    if (a.m.d <= LFM_MAX_COMPONENTS) return launch_nc<16>(loss, a, grid, block, smem, st, cus, grid_used);
    return hipErrorInvalidValue;
}
#else
hipError_t launch_fit(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st,
                      int cus, int *grid_used)
{
    int d = a.m.d;
    if (d <= 64) return launch_nc<1>(loss, a, grid, block, smem, st, cus, grid_used);
    if (d <= 128) return launch_nc<2>(loss, a, grid, block, smem, st, cus, grid_used);
    if (d <= 256) return launch_nc<4>(loss, a, grid, block, smem, st, cus, grid_used);
    if (d <= 512) return launch_nc<8>(loss, a, grid, block, smem, st, cus, grid_used);
    return launch_fit_wide(loss, a, grid, block, smem, st, cus, grid_used);
}

hipError_t launch_reg_log_init(const double *scales, double *reg_log, float *reg_live, hipStream_t st)
{
    reg_log_init_kernel<<<1, 64, 0, st>>>(scales, reg_log, reg_live);
    return hipGetLastError();
}

hipError_t launch_regularize(const DModel &m, double *reg_log, float *reg_live, int force, hipStream_t st, int64_t positions)
{
    // one pass over W and b of both sides (it returns at once unless a fold is due)
    const int64_t cells = (int64_t)m.d * std::max(m.n_feat[0], m.n_feat[1]);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (cells + 255) / 256));
    regularize_kernel<<<grid, 256, 0, st>>>(m, reg_log, reg_live, force);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    reg_boundary_kernel<<<1, 64, 0, st>>>(m.scales, reg_log, reg_live, force, positions);
    return hipGetLastError();
}

hipError_t launch_nonfinite(const float *x, int64_t n, int *flag, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    int grid = (int)std::min<int64_t>(2048, (n + 255) / 256);
    nonfinite_kernel<<<grid, 256, 0, st>>>(x, n, flag);
    return hipGetLastError();
}

#endif  // LFM_FIT_WIDE_UNIT

}  // namespace lfm
