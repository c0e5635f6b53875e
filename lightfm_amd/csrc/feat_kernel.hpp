// feat_kernel.hpp -- pipelined row-stream epoch kernels for every parallel-mode model the
// lane-group WARP tile kernel (warp_tile_kernel.hpp) does not cover: feature CSRs on either side
// (hybrid models), BPR, k-OS WARP and logistic.  BASELINE configs C3 (BPR, d = 128, item tags)
// and C5 (k-OS, d = 128, 8 nnz item rows).  PYX = /root/reference/lightfm/_lightfm_fast.pyx.template
//
// One interaction per wavefront, like the generic kernels of fit_kernels.hip, but organised as
// the tile kernel is -- every memory phase is ONE round trip for everything it needs:
//
//   jobs      the representations an interaction needs (user; positive item; ALL candidate
//             negatives of a batch / the n sampled positives of k-OS) are "jobs", one per lane.
//             Their CSR row bounds are fetched together (lane j reads indptr of job j), an
//             exclusive scan lays their feature entries out in one flat list, and lane t fetches
//             entry t: (feature, weight) of every job in one round trip.  An identity matrix
//             contributes its implicit entry without touching memory.
//   gather    the embedding rows of the flat list go memory -> LDS by LDS-DMA
//             (global_load_lds_dwordx4: d/4 lanes x 16 B per row, 64/(d/4) rows = 1 KiB per wave
//             instruction, no VGPRs, no ds_write), a whole chunk of rows in flight at once.
//   reduce    lane c owns component c: it walks the staged rows in CSR order and accumulates
//             w * row[c] in float32 exactly as compute_representation does (PYX:287-317); a job
//             boundary flushes the accumulator into the wave's representation tile.
//   score     lane r computes the reference's sequential float32 dot (PYX:320-334) of tile row r
//             with the user row: the positive and every candidate of the batch in one pass.
//   sample    as in the reference, with the position's own rand_r stream: WARP / k-OS take the first
//             violator of the speculatively scored batch (PYX:857-899, 1014-1057), BPR draws its
//             candidate negatives eight at a time and keeps the first non-positive (PYX:1123-1127);
//             streams advance by exactly the draws the sequential loop would have made.
//   update    the rows of the three (two) representations that are updated form one flat list
//             again; their W rows AND G rows are DMA'd together, lane c evaluates the reference's
//             float64 cell arithmetic (PYX:416-449) for its coordinate of every row and publishes
//             new - old with global_atomic_add_f32; the rows' bias cells are handled one per lane.
//
// Scope: parallel mode, adagrad, no_components a multiple of 4 up to 128; REG instantiations carry the
// lazy L2 regularisation (item_alpha / user_alpha != 0, PYX:640-691; device.hpp: RegScale).
// Everything else (serial mode, adadelta, wider models) runs the generic kernels.
#pragma once
#include "device.hpp"
#include "kernels.hpp"

namespace lfm {

namespace {

typedef __attribute__((address_space(3))) float lds_f32_t;

// One round (<= 64 entries) of a flat entry list: lane t holds entry t.
struct Entries {
    int feat;    // feature (embedding row) of the entry
    float w;     // its weight in the representation
    int job;     // which job of the list it belongs to
    int eside;   // 0 item-side tables, 1 user-side tables
    int n;       // entries in the round (wave-uniform)
};

}  // namespace

// LOSS: LFM_LOSS_* (0 logistic, 1 WARP, 2 BPR, 3 k-OS WARP).  NC = ceil(d / 64).
// TIMED (profiling builds, lfm_opts.feat_kernel = 2): every wavefront accumulates shader-clock
// deltas per phase of an interaction into a.counters[4..11] -- 0 record / sampling loads and
// in_positives, 1 CSR extents + entry lists, 2 representation row gathers, 3 reduction into the
// tile, 4 scoring and loss, 5 update row gathers (W and G), 6 cell arithmetic + publication,
// 7 everything else (loop tail, logs).
// Register budget: left alone the k-OS instantiation takes 129-133 VGPRs -- one too many for a fourth wavefront per SIMD.
// Under a launch bound of four it compiles to 106-120 without scratch (tests/test_kernel_resources.py), and 16 wavefronts per
// CU with a 10 KB LDS budget each (feat_kernels.hip: feat_plan) beat the 12 of round 4 by 2.9 % on the C5 shard (59.15 ->
// 60.87 M interactions/s at --scale 0.25; the same kernel at 12 wavefronts loses 1.3 % to the tighter allocation;
// profiles/r05_visit_i.txt): the kernel is instruction-bound, residency buys little.
#define LFM_FEAT_MIN_BLOCKS(LOSS, TIMED, ADA) (((LOSS) == 3 && !(TIMED) && !(ADA)) ? 4 : 2)  // (adadelta's third table: no 128-VGPR diet)
// HOT (round 6): the model has a hot set (device.hpp: HotRec; session.hip: HotSet).  The update leaves the hot item-feature
// rows out -- no W / G row gathers, no cell arithmetic, no atomics for them -- and writes their (slot, weight) entries, the
// jobs' gradient coefficients and the user representation to the position's record; hot_slice_kernel (hot_slices.hip)
// applies the records after the launch.  Scoring, sampling and every other row's update are unchanged.
// ADA (round 6): the adadelta schedule (PYX:416-434) -- the momentum rows M travel with W and G (three staged tables instead
// of two), the cell arithmetic is cell_math's adadelta branch and the moving averages are published by compare-and-swap
// (device.hpp: publish_adadelta: a summed delta would apply the decay once per concurrent writer).
template <int LOSS, int NC, bool TIMED = false, bool REG = false, bool HOT = false, bool ADA = false>
__global__ __launch_bounds__(256, LFM_FEAT_MIN_BLOCKS(LOSS, TIMED, ADA)) void fit_feat_kernel(FitArgs a)
{
    static_assert(!(HOT && (REG || TIMED || ADA)), "the hot-set variants: adagrad, no lazy regularisation, no phase timers");
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int k) {
        if constexpr (TIMED) {
            if (a.debug & 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // drain: pure phases
            unsigned long long t = __builtin_readcyclecounter();
            ph[k] += t - tprev;
            tprev = t;
        }
    };
    if constexpr (TIMED) tprev = __builtin_readcyclecounter();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), wib = uni((int)(threadIdx.x >> 6));
    const int d = a.m.d, TS = a.tile_stride, RR = a.tile_rows, SR = a.stage_rows;
    const int LPR = d >> 2, RPI = WAVE / LPR;  // lanes per row, rows per DMA instruction
    const size_t wave_floats = (size_t)SR * d + (size_t)RR * TS + 3 * (size_t)a.pair_cap + 2 * WAVE;
    float *stage = smem + (size_t)wib * wave_floats;  // [SR][d] packed: what LDS-DMA deposits
    float *reps = stage + (size_t)SR * d;              // [RR][TS] representations, bias in column d
    int *pair_idx = reinterpret_cast<int *>(reps + (size_t)RR * TS);  // k-OS (PYX:109-111)
    float *pair_val = reinterpret_cast<float *>(pair_idx + a.pair_cap);
    int *pair_slot = reinterpret_cast<int *>(pair_val + a.pair_cap);
    // weights and biases of a round's entries, by entry (fast reduce below): LDS broadcasts instead of lane reads
    float *wl = reinterpret_cast<float *>(pair_slot + a.pair_cap), *bl = wl + WAVE;
    // d == 64 NC (d = 64 or 128: every BASELINE configuration): the reduce runs with compile-time LDS offsets
    constexpr int DF = 64 * NC;
    const bool fastd = d == DF;
    const Hyper h{ADA ? 1 : 0, a.m.lr, a.m.rho, a.m.eps};
    const int um = a.update_mode;
    const bool ustore = a.user_store && um == 0;  // user-side rows (identity user features): plain stores, FitArgs::user_store
    const int max_sampled = a.m.max_sampled;
    const int cand_base = a.cand_base, CB = RR - cand_base;  // candidate rows of the tile
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    // REG: (float)(1.0 * scale) of the two sides for the current interaction (PYX:306), wave-uniform;
    // `reg` = this wavefront's view of the live scales (device.hpp: RegScale)
    float wsc_i = 1.0f, wsc_u = 1.0f;
    const double alpha_i = REG ? a.item_alpha : 0.0, alpha_u = REG ? a.user_alpha : 0.0;
    RegScale reg;
    reg.begin();
    auto refresh = [&](int64_t pos) {
        if constexpr (REG) RegScale::scales(a.reg_live, pos - a.begin, wsc_i, wsc_u);
    };
    // PYX:640-649 after an update: lr_sum = this lane's share of the cells' learning rates, T = entries updated
    auto scale_step = [&](double lr_sum, int T) {
        if constexpr (REG) {
            const double avg = wave_sum(lr_sum) / ((double)(a.m.d_real + 1) * (double)max(T, 1));
            if (um != 2) reg.add(RegScale::log1p_f32((float)(alpha_i * avg)), RegScale::log1p_f32((float)(alpha_u * avg)));
        }
    };

    // ---- a list of jobs (lane j = job j): CSR extent of every job and the flat entry layout
    auto job_extent = [&](int row, int side, int J, int &start, int &len, int &off, int &T) {
        start = 0;
        len = 0;
        if (lane < J) {
            const bool ident = side ? a.usf.identity : a.itf.identity;
            if (ident) {
                start = row;
                len = 1;
            } else {
                const int32_t *ip = side ? a.usf.indptr : a.itf.indptr;
                start = ip[row];
                len = ip[row + 1] - start;
            }
        }
        int incl = len;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int v = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += v;
        }
        off = incl - len;
        T = read_lane(incl, WAVE - 1);
    };
    // entry r * 64 + lane of the flat list: which job, which feature row, which weight
    auto round_entries = [&](int r, int J, int start, int len, int off, int side, int T, int &feat, float &w,
                             int &job, int &eside) {
        const int g = r * WAVE + lane;
        job = 0;
        for (int jj = 0; jj + 1 < J; ++jj) {  // jobs that end at or before g
            const int end_jj = read_lane(off, jj) + read_lane(len, jj);
            if (g >= end_jj) job = jj + 1;
        }
        const int js = __shfl(start, job, WAVE), jo = __shfl(off, job, WAVE);
        eside = __shfl(side, job, WAVE);
        const int k = js + (g - jo);
        feat = 0;
        w = 0.0f;
        if (g < T) {
            const bool ident = eside ? a.usf.identity : a.itf.identity;
            if (ident) {
                feat = k;
                w = 1.0f;
            } else {
                feat = (eside ? a.usf.indices : a.itf.indices)[k];
                w = (eside ? a.usf.data : a.itf.data)[k];
            }
        }
    };
    // rows of entries [c0e, c0e + nc) of the current round: memory -> dst[0 .. nc) by LDS-DMA
    auto dma_rows = [&](int feat, int eside, int c0e, int nc, float *dst, int tabk) {  // tabk: 0 = W, 1 = G, 2 = M rows
        const int rsub = lane / LPR, piece = lane - rsub * LPR;
        if (RPI == 2) {
            // d = 128: two rows per instruction.  Every lane forms the address of ITS entry's row once; an instruction's
            // two row addresses then come from lane reads (scalar) and a select by the lane's half -- no ds_bpermute
            // round trips, no per-instruction table select and 64-bit multiply (27 -> ~12 instructions per two rows)
            const float *tab = tabk == 2 ? (eside ? a.m.M[1] : a.m.M[0]) : (tabk == 1 ? (eside ? a.m.G[1] : a.m.G[0]) : (eside ? a.m.W[1] : a.m.W[0]));
            const unsigned long long mine = (unsigned long long)(uintptr_t)(tab + (size_t)feat * d);
            const int alo = (int)(unsigned)mine, ahi = (int)(unsigned)(mine >> 32);
            auto two_rows = [&](int i0, bool guard) {
                const int e0 = (c0e + i0) & (WAVE - 1), e1 = (c0e + i0 + 1) & (WAVE - 1);
                const unsigned lo0 = (unsigned)read_lane(alo, e0), hi0 = (unsigned)read_lane(ahi, e0);
                const unsigned lo1 = (unsigned)read_lane(alo, e1), hi1 = (unsigned)read_lane(ahi, e1);
                const unsigned long long base = rsub ? (((unsigned long long)hi1 << 32) | lo1) : (((unsigned long long)hi0 << 32) | lo0);
                const float *src = reinterpret_cast<const float *>((uintptr_t)base) + piece * 4;
                if (!guard || i0 + rsub < nc) __builtin_amdgcn_global_load_lds(src, (lds_f32_t *)(dst + (size_t)i0 * d), 16, 0, 0);
            };
            int i0 = 0;
            for (; i0 + 4 <= nc; i0 += 4) {  // four rows per step, no per-lane predicate
                two_rows(i0, false);
                two_rows(i0 + 2, false);
            }
            for (; i0 < nc; i0 += 2) two_rows(i0, true);
            return;
        }
        for (int i0 = 0; i0 < nc; i0 += RPI) {
            const int e = c0e + i0 + rsub;
            const bool valid = rsub < RPI && (i0 + rsub) < nc;
            const int fe = __shfl(feat, e & (WAVE - 1), WAVE), se = __shfl(eside, e & (WAVE - 1), WAVE);
            const float *tab = tabk == 2 ? (se ? a.m.M[1] : a.m.M[0]) : (tabk == 1 ? (se ? a.m.G[1] : a.m.G[0]) : (se ? a.m.W[1] : a.m.W[0]));
            const float *src = tab + (size_t)fe * d + piece * 4;
            if (valid) __builtin_amdgcn_global_load_lds(src, (lds_f32_t *)(dst + (size_t)i0 * d), 16, 0, 0);
        }
    };

    // ---- representations of J jobs -> tile rows rrow[j]  (compute_representation, PYX:287-317).
    // When the whole flat list fits one round (<= 64 entries) it is handed back in `keep` so that
    // an update of the same jobs need not fetch it again.
    auto build_reps = [&](int row, int side, int rrow, int J, Entries *keep) {
        int start, len, off, T;
        stamp(0);
        job_extent(row, side, J, start, len, off, T);
        // a job without entries has the zero representation (every other job's row is written when its last entry has
        // been reduced): only those rows are cleared -- rare (an empty feature row), so the common case clears nothing
        for (unsigned long long em = __ballot(lane < J && len == 0); em != 0ull; em &= em - 1ull) {
            float *rp = reps + (size_t)read_lane(rrow, __ffsll((long long)em) - 1) * TS;
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int c = lane + WAVE * q;
                if (c < d) rp[c] = 0.0f;
            }
            if (lane == 0) rp[d] = 0.0f;
        }
        if (keep) keep->n = -1;
        int cur = -1;
        float acc[NC], accb = 0.0f;
        float accb_job = 0.0f;  // fast reduce: lane j = the bias accumulator of job j
        int cc_[NC];  // this lane's components, lanes past d clamped to component 0 (never stored)
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            acc[q] = 0.0f;
            cc_[q] = (lane + WAVE * q) < d ? lane + WAVE * q : 0;
        }
        auto flush = [&]() {
            float *rp = reps + (size_t)read_lane(rrow, cur) * TS;
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int c = lane + WAVE * q;
                if (c < d) rp[c] = acc[q];
            }
            if (!fastd && lane == 0) rp[d] = accb;  // (fast reduce: the bias column is written once, after the last round)
        };
        for (int r = 0; r * WAVE < T; ++r) {
            Entries e;
            round_entries(r, J, start, len, off, side, T, e.feat, e.w, e.job, e.eside);
            e.n = min(WAVE, T - r * WAVE);
            if (keep && T <= WAVE) *keep = e;
            float bx = 0.0f;
            if (lane < e.n) bx = (e.eside ? a.m.b[1] : a.m.b[0])[e.feat];
            if constexpr (TIMED) asm volatile("" : "+v"(e.feat), "+v"(e.w));
            stamp(1);
            // end of every job's entries inside this round (lane j = job j), for the fast reduce
            const int jend_round = off + len - r * WAVE;
            for (int ce = 0; ce < e.n; ce += SR) {
                const int nc = min(SR, e.n - ce);
                dma_rows(e.feat, e.eside, ce, nc, stage, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA'd rows have landed
                wave_sync();
                stamp(2);
                if (fastd) {
                    if (ce == 0) {  // the round's weights (PYX:306: data * scale through float64) and biases, by entry
                        float wv = e.w;
                        if constexpr (REG) wv = (float)((double)wv * (double)(e.eside ? wsc_u : wsc_i));
                        wl[lane] = wv;
                        bl[lane] = bx;
                        wave_sync();
                    }
                    // Fast reduce (d = 64 NC): job by job, four entries per step.  Within a job the float32 accumulation
                    // is one sequential chain in CSR order (PYX:306-313); everything around it is now cheap: the rows'
                    // LDS offsets are compile-time constants of ONE address register, the weights are LDS broadcasts
                    // (two ds_read2 per four entries instead of one lane read per entry), and the bias column is
                    // summed afterwards by one lane per job.
                    for (int t = ce; t < ce + nc;) {
                        const int jt = read_lane(e.job, t);
                        if (jt != cur) {
                            if (cur >= 0) flush();
                            cur = jt;
#pragma unroll
                            for (int q = 0; q < NC; ++q) acc[q] = 0.0f;
                        }
                        const int jend = min(ce + nc, read_lane(jend_round, jt));
                        const float *xb = stage + (size_t)(t - ce) * DF + lane;  // this lane's components of entry t's row
                        const float *wb = wl + t;                               // wave-uniform
                        int left = jend - t;
                        for (; left >= 4; left -= 4, xb += 4 * DF, wb += 4) {
                            float xv[4][NC], wt[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                wt[u] = wb[u];
#pragma unroll
                                for (int q = 0; q < NC; ++q) xv[u][q] = xb[u * DF + WAVE * q];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int q = 0; q < NC; ++q) acc[q] = __fadd_rn(acc[q], __fmul_rn(wt[u], xv[u][q]));
                        }
                        for (; left > 0; --left, xb += DF, ++wb) {
                            const float wt = wb[0];
#pragma unroll
                            for (int q = 0; q < NC; ++q) acc[q] = __fadd_rn(acc[q], __fmul_rn(wt, xb[WAVE * q]));
                        }
                        t = jend;
                    }
                    wave_sync();  // the stage is rewritten by the next chunk
                    stamp(3);
                    continue;
                }
#ifdef LFM_FEAT_REDUCE_BATCH4  // A/B build (profiles/r04_visit_b.txt: C5 -6 %, C3 +4 %: not the bound -- kept for study)
                // Reduction of the staged rows, job by job.  Within a job the float32 accumulation is one
                // sequential chain in CSR order (PYX:306-313); what is NOT sequential is everything around it, so
                // the rows of up to four consecutive entries of the job are read from the stage together (one LDS
                // latency per four entries instead of one per entry) and their weights are fetched by lane reads
                // while those reads are in flight.  No per-lane predicates in the loop: lanes past d read
                // component 0 and their accumulators are never stored (flush() writes c < d only).
                for (int t = ce; t < ce + nc;) {
                    const int jt = read_lane(e.job, t);
                    if (jt != cur) {
                        if (cur >= 0) flush();
                        cur = jt;
#pragma unroll
                        for (int q = 0; q < NC; ++q) acc[q] = 0.0f;
                        accb = 0.0f;
                    }
                    // entries [t, jend) of this chunk belong to job `cur`
                    const int jend = min(ce + nc, read_lane(off, jt) + read_lane(len, jt) - r * WAVE);
                    for (; t < jend; t += 4) {
                        const int m = jend - t;  // >= 1; entries t .. t + min(m, 4) - 1
                        float xv[4][NC], wt[4], bt[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int tu = min(t + u, jend - 1);  // past the job's end: its last row again (unused)
                            const float *sr = stage + (size_t)(tu - ce) * d;
#pragma unroll
                            for (int q = 0; q < NC; ++q) xv[u][q] = sr[cc_[q]];
                            wt[u] = read_lanef(e.w, tu);
                            bt[u] = read_lanef(bx, tu);
                            if constexpr (REG)  // feature_weight = data * scale, PYX:306 (C_OMP:4896: through float64)
                                wt[u] = (float)((double)wt[u] * (double)(read_lane(e.eside, tu) ? wsc_u : wsc_i));
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (u < m) {  // wave-uniform
#pragma unroll
                                for (int q = 0; q < NC; ++q) acc[q] = __fadd_rn(acc[q], __fmul_rn(wt[u], xv[u][q]));
                                accb = __fadd_rn(accb, __fmul_rn(wt[u], bt[u]));
                            }
                        }
                    }
                    t = jend;
                }
#else
                for (int t = ce; t < ce + nc; ++t) {
                    const int jt = read_lane(e.job, t);
                    if (jt != cur) {
                        if (cur >= 0) flush();
                        cur = jt;
#pragma unroll
                        for (int q = 0; q < NC; ++q) acc[q] = 0.0f;
                        accb = 0.0f;
                    }
                    float wt = read_lanef(e.w, t);
                    const float bt = read_lanef(bx, t);
                    if constexpr (REG)  // feature_weight = data * scale, PYX:306 (C_OMP:4896: through float64)
                        wt = (float)((double)wt * (double)(read_lane(e.eside, t) ? wsc_u : wsc_i));
                    const float *sr = stage + (size_t)(t - ce) * d;
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        const int c = lane + WAVE * q;
                        const float xv = c < d ? sr[c] : 0.0f;
                        acc[q] = __fadd_rn(acc[q], __fmul_rn(wt, xv));
                    }
                    accb = __fadd_rn(accb, __fmul_rn(wt, bt));
                }
#endif
                wave_sync();  // the stage is rewritten by the next chunk
                stamp(3);
            }
            if (fastd) {
                // bias column of the round: lane j walks job j's entries of this round in CSR order (PYX:314)
                if (lane < J) {
                    const int b0 = max(off - r * WAVE, 0), b1 = min(jend_round, e.n);
                    for (int k = b0; k < b1; ++k) accb_job = __fadd_rn(accb_job, __fmul_rn(wl[k], bl[k]));
                }
                wave_sync();  // wl / bl are rewritten by the next round
            }
        }
        if (cur >= 0) flush();
        if (fastd && lane < J) reps[(size_t)rrow * TS + d] = accb_job;  // (a job without entries: the zero it already holds)
        wave_sync();
        stamp(3);
    };

    // ---- update of the feature rows of one round of a flat list (update / warp_update,
    // PYX:454-649): job j's coordinate cells get g = gj * x[c] with x = xI for item-side rows and
    // xU for user-side rows; its bias cells get g = gj.
    // The reference updates the rows of an interaction one after the other.  Entries that name
    // DIFFERENT rows commute, so they are all fetched, computed and published together.  A row
    // that occurs twice (a tag shared by the positive and the negative item; a repeated column)
    // must see its earlier update: such entries get a "generation" (how many earlier entries name
    // the same row) and the generations are processed one after the other with a device-scope
    // fence in between -- rare, and exactly the sequential result.
    HotRec *hrec = nullptr;                  // HOT: this position's record
    int hot_n0 = 0, hot_n1 = 0, hot_n2 = 0;  // HOT: entries recorded so far for job 0 / 1 / 2 of the current interaction
    auto update_round = [&](const Entries &e_in, double g0, double g1, double g2, const float (&xI)[NC],
                            const float (&xU)[NC], double &lr_sum, bool whole_list = true) {
        Entries e = e_in;
        if constexpr (HOT) {
            // entries that name a hot item-side row go to the record (in list order: job by job, as the reference walks
            // them); the others are compacted to the front of the list and updated below as ever.  All of an
            // interaction's hot entries or none: a tag shared by the positive and the negative item must see its two
            // updates in the reference's order (early on, G = 1, the two nearly cancel and what remains depends on the
            // order), so an interaction whose list takes several rounds (> 64 entries) or holds more hot entries than a
            // record (HOT_EMAX) publishes everything as before.
            int hs = -1;
            if (whole_list && lane < e.n && e.eside == 0) hs = a.hot_slot[e.feat];
            unsigned long long hm = __ballot(hs >= 0);
            if (__popcll(hm) > HOT_EMAX) {
                hm = 0ull;
                hs = -1;
            }
            if (hm != 0ull) {
                const unsigned long long below = (1ull << lane) - 1ull;
                const int rank = hot_n0 + hot_n1 + hot_n2 + __popcll(hm & below);
                const bool rec = hs >= 0;
                if (rec) {
                    HotRec::Entry en;
                    en.slot = hs;
                    en.w = e.w;
                    hrec->e[rank] = en;
                }
                hot_n0 += __popcll(__ballot(rec && e.job == 0));
                hot_n1 += __popcll(__ballot(rec && e.job == 1));
                hot_n2 += __popcll(__ballot(rec && e.job == 2));
                const unsigned long long km = __ballot(lane < e.n && !rec);
                const int nk = __popcll(km);
                const int dst = ((km >> lane) & 1ull) ? __popcll(km & below) : nk + __popcll(~km & below);  // a permutation
                e.feat = __builtin_amdgcn_ds_permute(dst << 2, e.feat);
                e.w = __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(e.w)));
                const int js = __builtin_amdgcn_ds_permute(dst << 2, e.job | (e.eside << 8));
                e.job = js & 0xff;
                e.eside = js >> 8;
                e.n = nk;
            }
        }
        const int SRh = ADA ? SR / 3 : SR >> 1;
        float *stW = stage, *stG = stage + (size_t)SRh * d, *stM = stage + 2 * (size_t)SRh * d;
        const bool on = lane < e.n;
        int gen = 0;
        stamp(4);
        for (int t = 0; t + 1 < e.n; ++t) {
            const int ft = read_lane(e.feat, t), st_ = read_lane(e.eside, t);
            if (on && lane > t && e.feat == ft && e.eside == st_) ++gen;
        }
        float *bp = (e.eside ? a.m.b[1] : a.m.b[0]) + e.feat, *bgp = (e.eside ? a.m.bG[1] : a.m.bG[0]) + e.feat;
        float *bmp = ADA ? (e.eside ? a.m.bM[1] : a.m.bM[0]) + e.feat : nullptr;
        const double gb = e.job == 0 ? g0 : (e.job == 1 ? g1 : g2);
        for (int g = 0;; ++g) {
            const unsigned long long live = __ballot(on && gen == g);
            if (live == 0ull) break;
            if (g > 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // the earlier generation is visible
            const bool mine = on && gen == g;
            float obW = 0.0f, obG = 1.0f, obM = 0.0f;
            if (mine) {  // bias cells: requested with the first rows, consumed after the last
                obW = *bp;
                obG = *bgp;
                if constexpr (ADA) obM = *bmp;
            }
            for (int ce = 0; ce < e.n; ce += SRh) {
                const int nc = min(SRh, e.n - ce);
                if (((live >> ce) & ((nc >= 64) ? ~0ull : ((1ull << nc) - 1ull))) == 0ull) continue;
                dma_rows(e.feat, e.eside, ce, nc, stW, 0);
                dma_rows(e.feat, e.eside, ce, nc, stG, 1);
                if constexpr (ADA) dma_rows(e.feat, e.eside, ce, nc, stM, 2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_sync();
                stamp(5);
                for (int t = ce; t < ce + nc; ++t) {
                    if (!((live >> t) & 1ull)) continue;
                    const int jt = read_lane(e.job, t), st_ = read_lane(e.eside, t), fe = read_lane(e.feat, t);
                    const double wt = (double)read_lanef(e.w, t);
                    const double gc = jt == 0 ? g0 : (jt == 1 ? g1 : g2);
                    float *Wp = (st_ ? a.m.W[1] : a.m.W[0]) + (size_t)fe * d;
                    float *Gp = (st_ ? a.m.G[1] : a.m.G[0]) + (size_t)fe * d;
                    const float *sw = stW + (size_t)(t - ce) * d, *sg = stG + (size_t)(t - ce) * d, *sm = stM + (size_t)(t - ce) * d;
                    float *Mp = ADA ? (st_ ? a.m.M[1] : a.m.M[0]) + (size_t)fe * d : nullptr;
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        const int c = lane + WAVE * q;
                        if (c < d) {
                            const float oW = sw[c], oG = sg[c], oM = ADA ? sm[c] : 0.0f;
                            const float x = st_ ? xU[q] : xI[q];
                            float nW, nG, nM;
                            double lr;
                            const double gcell = gc * (double)x, al = st_ ? alpha_u : alpha_i;
                            if constexpr (!ADA && !REG) cell_math_adagrad(oW, oG, wt, gcell, h.lr, nW, nG);  // (= cell_math, bit for bit)
                            else cell_math(oW, oG, oM, wt, gcell, h, al, nW, nG, nM, lr);
                            if constexpr (ADA) {
                                // (the moving averages by compare-and-swap: the learning rate that counts is the one of
                                // the value actually replaced)
                                if (um == 0) lr = publish_adadelta(Wp + c, Gp + c, Mp + c, oW, oG, oM, wt, gcell, h, al);
                                else publish_cell(Wp + c, Gp + c, Mp + c, oW, oG, oM, nW, nG, nM, wt, gcell, h, al, um);
                            } else {
                                publish(Wp + c, nW, oW, (st_ && ustore) ? 1 : um);
                                if (!(a.debug & 65536)) publish(Gp + c, nG, oG, (st_ && ustore) ? 1 : um);  // (timing experiment, WRONG results: no accumulator rows)
                            }
                            if constexpr (REG) lr_sum += c < a.m.d_real ? lr : 0.0;
                        }
                    }
                }
                wave_sync();
                stamp(6);
            }
            {
                float nW, nG, nM;
                double lr;
                const double alb = e.eside ? alpha_u : alpha_i;
                if constexpr (!ADA && !REG) cell_math_adagrad(obW, obG, (double)e.w, gb, h.lr, nW, nG);
                else cell_math(obW, obG, obM, (double)e.w, gb, h, alb, nW, nG, nM, lr);
                if (mine) {
                    if constexpr (ADA) {
                        if (um == 0) lr = publish_adadelta(bp, bgp, bmp, obW, obG, obM, (double)e.w, gb, h, alb);
                        else publish_cell(bp, bgp, bmp, obW, obG, obM, nW, nG, nM, (double)e.w, gb, h, alb, um);
                    } else {
                        if (!(a.debug & 32768)) {  // (timing experiment, WRONG results: no bias publication)
                            publish(bp, nW, obW, (e.eside && ustore) ? 1 : um);
                            publish(bgp, nG, obG, (e.eside && ustore) ? 1 : um);
                        }
                    }
                    if constexpr (REG) lr_sum += lr;
                }
            }
            stamp(6);
        }
    };
    // HOT: the record's header and the vector the hot rows' gradients multiply (item-side rows: x = xI, PYX:602-638),
    // written once the interaction's lists have been walked
    auto hot_finish = [&](double g0, double g1, double g2, const float (&xI)[NC]) {
        if constexpr (HOT) {
            const int tot = hot_n0 + hot_n1 + hot_n2;
            if (tot > 0) {
                if (lane == 0) {
                    hrec->g[0] = g0;
                    hrec->g[1] = g1;
                    hrec->g[2] = g2;
                    hrec->cnt[0] = (unsigned char)hot_n0;
                    hrec->cnt[1] = (unsigned char)hot_n1;
                    hrec->cnt[2] = (unsigned char)hot_n2;
                    hrec->n_total = tot;
                }
                float *xp = a.hot_x + (size_t)(hrec - a.hot_rec) * d;
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int c = lane + WAVE * q;
                    if (c < d) xp[c] = xI[q];
                }
            }
            hot_n0 = hot_n1 = hot_n2 = 0;
        }
    };
    // the same for a list of jobs given as rows (fetches the flat list round by round)
    auto update_rows = [&](int row, int side, int J, double g0, double g1, double g2, const float (&xI)[NC],
                           const float (&xU)[NC]) {
        int start, len, off, T;
        job_extent(row, side, J, start, len, off, T);
        double lr_sum = 0.0;
        for (int r = 0; r * WAVE < T; ++r) {
            Entries e;
            round_entries(r, J, start, len, off, side, T, e.feat, e.w, e.job, e.eside);
            e.n = min(WAVE, T - r * WAVE);
            if (r > 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // rounds are sequential
            update_round(e, g0, g1, g2, xI, xU, lr_sum, T <= WAVE);
        }
        scale_step(lr_sum, T);
        hot_finish(g0, g1, g2, xI);
    };
    // a list that fitted one round and was kept from the representation phase
    auto update_kept = [&](const Entries &e, double g0, double g1, double g2, const float (&xI)[NC], const float (&xU)[NC]) {
        double lr_sum = 0.0;
        update_round(e, g0, g1, g2, xI, xU, lr_sum);
        scale_step(lr_sum, e.n);
        hot_finish(g0, g1, g2, xI);
    };
    auto rep_regs = [&](int r, float (&v)[NC]) {
        const float *rp = reps + (size_t)r * TS;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int c = lane + WAVE * q;
            v[q] = c < d ? rp[c] : 0.0f;
        }
    };
    auto log_pos = [&](int64_t i, int neg, int sampled) {
        if (lane == 0) {
            if (a.neg_log) a.neg_log[i] = neg;
            if (a.sampled_log) a.sampled_log[i] = sampled;
        }
    };

    const int wpb = (int)(blockDim.x >> 6);  // 1, 2 or 4 wavefronts per workgroup (feat_plan)
    const int64_t gw = (int64_t)blockIdx.x * wpb + wib;
    const int64_t nw = (int64_t)gridDim.x * wpb;
    const uint32_t base_seed = LOSS != LFM_LOSS_LOGISTIC_ID ? a.seeds[0] : 0u;
    auto fetch = [&](int row) -> int4 {
        row = guard_row(a, row);
        if constexpr (LOSS == LFM_LOSS_WARP_KOS_ID) return make_int4(a.user_ids[row], 0, 0, 0);
        else return a.recs[row];
    };
    // BPR: what the NEXT position's sampling needs first is requested while the current position is worked on -- its first
    // eight candidate negatives (PYX:1124-1125: they depend on the position's PRNG stream alone) at the top of the current
    // iteration, its user's row bounds in the positives lookup once its record has arrived -- two dependent round trips
    // off the chain of an interaction
    int pre_cand = 0, pre_lo = 0, pre_hi = 0;
    bool pre_cand_ok = false, pre_pos_ok = false;
    int64_t i = a.begin + gw;
    int4 cur = make_int4(0, 0, 0, 0);
    int row1 = 0;
    if (i < a.end) cur = fetch(a.shuffle[i]);
    if (i + nw < a.end) row1 = a.shuffle[i + nw];
    for (; i < a.end; i += nw) {
        int4 nxt = make_int4(0, 0, 0, 0);
        int row2 = 0;
        if (i + 2 * nw < a.end) row2 = a.shuffle[i + 2 * nw];
        if (i + nw < a.end) nxt = fetch(row1);
        stamp(7);
        const int user = uni(cur.x), item = uni(cur.y);
        const float y = unif(__int_as_float(cur.z)), wgt = unif(__int_as_float(cur.w));
        cur = nxt;
        row1 = row2;
        refresh(i);
        if constexpr (HOT) hrec = a.hot_rec + (i - a.begin);
        const int cand_first = pre_cand, lo_first = pre_lo, hi_first = pre_hi;
        const bool have_cand = pre_cand_ok, have_bounds = pre_pos_ok;
        pre_cand_ok = pre_pos_ok = false;
        if constexpr (LOSS == LFM_LOSS_BPR_ID) {
            if (i + nw < a.end) {
                uint32_t sn = position_seed(base_seed, (uint64_t)(i + nw));
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j <= lane) sn = lcg(sn);
                if (lane < 8) pre_cand = a.item_ids[draw(sn) % (uint32_t)a.n];
                pre_cand_ok = true;
            }
        }

        if constexpr (LOSS == LFM_LOSS_LOGISTIC_ID) {
            // fit_logistic, PYX:726-775
            Entries el;
            build_reps(lane == 0 ? user : item, lane == 0 ? 1 : 0, lane, 2, &el);
            float s = 0.0f;
            if (lane == 1) s = tile_dot(reps, reps + TS, d);
            s = read_lanef(s, 1);
            float Uv[NC], Iv[NC];
            rep_regs(0, Uv);
            rep_regs(1, Iv);
            wave_sync();
            const double prediction = (double)sigmoidf_ref(s);  // PYX:745-747
            const int yb = (y <= 0.0f) ? 0 : 1;                  // PYX:751-755
            if (yb) c0++;
            const double loss = (double)wgt * (prediction - (double)yb);
            // jobs in the order of the representation list: user row (x = item), item row (x = user)
            if (el.n >= 0) update_kept(el, loss, loss, 0.0, Uv, Iv);
            else update_rows(lane == 0 ? user : item, lane == 0 ? 1 : 0, 2, loss, loss, 0.0, Uv, Iv);
            c2++;
            continue;
        } else {
            if constexpr (LOSS != LFM_LOSS_WARP_KOS_ID) {
                if (!(y > 0.0f)) {  // PYX:831-832 / 1116-1117, before any RNG use
                    log_pos(i, -1, 0);
                    continue;
                }
            }
            uint32_t state = position_seed(base_seed, (uint64_t)i);
            // requested now, made wave-uniform where first needed (so that it travels with later loads)
            int lo_v = lo_first, hi_v = hi_first;
            if (!have_bounds) {
                lo_v = a.pos.indptr[user];
                hi_v = a.pos.indptr[user + 1];
            }
            // (every loss with a positives lookup: the NEXT position's row bounds are requested after this position's first
            // representation pass, by when its record has arrived)
            auto prefetch_next_bounds = [&]() {
                if (i + nw < a.end && !pre_pos_ok) {
                    pre_lo = a.pos.indptr[cur.x];
                    pre_hi = a.pos.indptr[cur.x + 1];
                    pre_pos_ok = true;
                }
            };

            if constexpr (LOSS == LFM_LOSS_BPR_ID) {
                // fit_bpr, PYX:1118-1169
                c0++;
                const uint32_t n_examples = (uint32_t)a.n;
                int neg = -1, draws = 0, lo = 0, hi = 0;
                while (neg < 0) {
                    uint32_t s = state;  // lane j: the stream after min(j + 1, 8) steps
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j <= lane) s = lcg(s);
                    int cand = cand_first;  // (requested during the previous position)
                    if (!(have_cand && draws == 0) && lane < 8) cand = a.item_ids[draw(s) % n_examples];  // PYX:1124-1125
                    lo = uni(lo_v);
                    hi = uni(hi_v);
                    int used = 8;
                    for (int j = 0; j < 8; ++j) {
                        const int c = read_lane(cand, j);
                        c3++;
                        const bool last = (int64_t)draws + j + 1 >= a.n;  // PYX:1123: at most no_examples draws
                        if (last || !in_positives_range(a.pos, c, lo, hi, lane)) {
                            neg = c;
                            used = j + 1;
                            break;
                        }
                    }
                    draws += used;
                    state = (uint32_t)read_lane((int)s, used - 1);
                }
                c1 += (unsigned long long)draws;
                Entries el;
                build_reps(lane == 0 ? user : (lane == 1 ? item : neg), lane == 0 ? 1 : 0, lane, 3, &el);
                prefetch_next_bounds();  // the next position's record has arrived by now (cur holds it since the top of the loop)
                float sc = 0.0f;
                if (lane == 1 || lane == 2) sc = tile_dot(reps, reps + (size_t)lane * TS, d);
                const double pp = (double)read_lanef(sc, 1), np_ = (double)read_lanef(sc, 2);
                float Uv[NC], Pv[NC], Nv[NC], diff[NC];
                rep_regs(0, Uv);
                rep_regs(1, Pv);
                rep_regs(2, Nv);
#pragma unroll
                for (int q = 0; q < NC; ++q) diff[q] = __fsub_rn(Nv[q], Pv[q]);
                wave_sync();
                // PYX:1158: weight * (1 - sigmoid(pp - np)); the difference is narrowed to float32
                const double loss = (double)wgt * (1.0 - (double)sigmoidf_ref((float)(pp - np_)));
                // jobs in the order of the representation list: user (+loss, x = neg - pos), positive
                // (-loss, x = user), negative (+loss, x = user)
                if (el.n >= 0) update_kept(el, loss, -loss, loss, Uv, diff);
                else update_rows(lane == 0 ? user : (lane == 1 ? item : neg), lane == 0 ? 1 : 0, 3, loss, -loss, loss,
                                 Uv, diff);
                c2++;
                log_pos(i, neg, draws);
                continue;
            } else {
                // fit_warp (PYX:826-904) and fit_warp_kos (PYX:958-1061) share the negative sampling
                int pos_item = item, prow = 1;
                float Pkos[NC];  // k-OS: the chosen positive's representation (its tile row is reused by the candidates)
#pragma unroll
                for (int q = 0; q < NC; ++q) Pkos[q] = 0.0f;
                double pp = 0.0;
                bool have_pos = false;  // the positive's score is known (k-OS: before the negatives)
                int lo = 0, hi = 0;
                if constexpr (LOSS == LFM_LOSS_WARP_KOS_ID) {
                    lo = uni(lo_v);
                    hi = uni(hi_v);
                    if (hi == lo) {  // PYX:971-972
                        log_pos(i, -1, 0);
                        continue;
                    }
                    c0++;
                    const int no_pos = min(a.n_pos, hi - lo);  // PYX:975
                    uint32_t s = state;                        // lane l: the stream after min(l, no_pos) steps
                    for (int j = 0; j < no_pos; ++j)
                        if (j < lane) s = lcg(s);
                    int it = 0;
                    if (lane >= 1 && lane <= no_pos)  // sample_range, PYX:84-90
                        it = a.pos.indices[lo + (int)(draw(s) % (uint32_t)(hi - lo))];
                    state = (uint32_t)read_lane((int)s, no_pos);
                    build_reps(lane == 0 ? user : it, lane == 0 ? 1 : 0, lane, 1 + no_pos, nullptr);
                    prefetch_next_bounds();
                    if (lane >= 1 && lane <= no_pos) {
                        pair_idx[lane - 1] = it;
                        pair_val[lane - 1] = tile_dot(reps, reps + (size_t)lane * TS, d);
                        pair_slot[lane - 1] = lane;
                    }
                    wave_sync();
                    if (lane == 0) {  // qsort(reverse_pair_compare), PYX:997: stable descending insertion sort
                        for (int x = 1; x < no_pos; ++x) {
                            const int ki = pair_idx[x], ks = pair_slot[x];
                            const float kv = pair_val[x];
                            int yv = x - 1;
                            while (yv >= 0 && (pair_val[yv] - kv) < 0.0f) {
                                pair_idx[yv + 1] = pair_idx[yv];
                                pair_val[yv + 1] = pair_val[yv];
                                pair_slot[yv + 1] = pair_slot[yv];
                                --yv;
                            }
                            pair_idx[yv + 1] = ki;
                            pair_val[yv + 1] = kv;
                            pair_slot[yv + 1] = ks;
                        }
                    }
                    wave_sync();
                    const int kk = min(a.k, no_pos) - 1;  // PYX:1002-1003
                    pos_item = uni(pair_idx[kk]);
                    pp = (double)unif(pair_val[kk]);
                    prow = uni(pair_slot[kk]);
                    have_pos = true;
                    rep_regs(prow, Pkos);  // the sampled positives' rows are dead from here on: candidates take them (feat_plan)
                    wave_sync();
                } else {
                    c0++;
                }
                int sampled = 0, chosen = -1, chosen_row = -1;
                while (sampled < max_sampled && chosen < 0) {
                    const int nb = min(max_sampled - sampled, min(sampled == 0 ? a.first_batch : CB, CB));
                    uint32_t s = state;  // lane k: the stream after min(k + 1, nb) steps = draw #(sampled + k + 1)
                    for (int j = 0; j < nb; ++j)
                        if (j <= lane) s = lcg(s);
                    const int myneg = (int)(draw(s) % (uint32_t)a.itf.rows);  // PYX:860-861
                    if (!have_pos) {
                        // the first batch gathers user, positive and candidates in one pass
                        const int cand = __shfl(myneg, max(lane - 2, 0), WAVE);
                        build_reps(lane == 0 ? user : (lane == 1 ? pos_item : cand), lane == 0 ? 1 : 0,
                                   lane < 2 ? lane : cand_base + lane - 2, 2 + nb, nullptr);
                        prefetch_next_bounds();
                        lo = uni(lo_v);
                        hi = uni(hi_v);
                    } else {
                        build_reps(myneg, 0, cand_base + lane, nb, nullptr);
                    }
                    float sc = 0.0f;
                    const bool mine = lane >= cand_base && lane < cand_base + nb;
                    if (mine || (!have_pos && lane == prow)) sc = tile_dot(reps, reps + (size_t)lane * TS, d);
                    if (!have_pos) {
                        pp = (double)read_lanef(sc, prow);
                        have_pos = true;
                    }
                    // PYX:875 compares doubles: negative_prediction > positive_prediction - 1
                    unsigned long long mask = __ballot(mine && ((double)sc > pp - 1.0));
                    int used = nb;
                    while (mask) {
                        const int slot = __ffsll((long long)mask) - 1 - cand_base;
                        mask &= mask - 1;
                        const int neg = read_lane(myneg, slot);
                        c3++;
                        if (in_positives_range(a.pos, neg, lo, hi, lane)) continue;  // PYX:878-879, draw counted
                        chosen = neg;
                        chosen_row = cand_base + slot;
                        used = slot + 1;
                        break;
                    }
                    sampled += used;
                    state = (uint32_t)read_lane((int)s, used - 1);
                    wave_sync();
                }
                c1 += (unsigned long long)sampled;
                if (chosen >= 0) {
                    // PYX:881-885 (k-OS: PYX:1039-1043, no weight); log table from the host libm
                    double loss = LOSS == LFM_LOSS_WARP_KOS_ID ? a.logtab[sampled] : (double)wgt * a.logtab[sampled];
                    if (loss > MAX_LOSS) loss = MAX_LOSS;
                    float Uv[NC], Pv[NC], Nv[NC], diff[NC];
                    rep_regs(0, Uv);
                    if constexpr (LOSS == LFM_LOSS_WARP_KOS_ID) {
#pragma unroll
                        for (int q = 0; q < NC; ++q) Pv[q] = Pkos[q];
                    } else {
                        rep_regs(prow, Pv);
                    }
                    rep_regs(chosen_row, Nv);
#pragma unroll
                    for (int q = 0; q < NC; ++q) diff[q] = __fsub_rn(Nv[q], Pv[q]);
                    wave_sync();
                    update_rows(lane == 0 ? pos_item : (lane == 1 ? chosen : user), lane == 2 ? 1 : 0, 3, -loss, loss,
                                loss, Uv, diff);
                    c2++;
                }
                log_pos(i, chosen, sampled);
            }
        }
    }
    if constexpr (REG) RegScale::publish(a.reg_live, reg.p_i, reg.p_u, lane, (unsigned)(blockIdx.x * (blockDim.x >> 6) + wib));
    if constexpr (TIMED) {
        stamp(7);
        if (lane == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(a.counters + 4 + k, ph[k]);
    }
    if (lane == 0) {
        if (c0) atomicAdd(a.counters + 0, c0);
        if (c1) atomicAdd(a.counters + 1, c1);
        if (c2) atomicAdd(a.counters + 2, c2);
        if (c3) atomicAdd(a.counters + 3, c3);
    }
}

// host side: picks the instantiation of a loss (feat_kernels.hip: NC = 1, 2; feat_kernels_wide.hip: NC = 4)
template <int NC>
inline hipError_t launch_feat_nc(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                                 int *grid_used, bool timed)
{
    void (*kernel)(FitArgs) = nullptr;
    if (timed && a.item_alpha == 0.0 && a.user_alpha == 0.0) {  // profiling builds (per-phase shader clocks), the two BASELINE losses only
        if (loss == LFM_LOSS_BPR_ID) kernel = fit_feat_kernel<LFM_LOSS_BPR_ID, NC, true>;
        else if (loss == LFM_LOSS_WARP_KOS_ID) kernel = fit_feat_kernel<LFM_LOSS_WARP_KOS_ID, NC, true>;
    }
    const bool reg = a.item_alpha != 0.0 || a.user_alpha != 0.0;
    if (!kernel && reg) switch (loss) {  // lazy L2 regularisation (device.hpp: RegScale)
    case LFM_LOSS_LOGISTIC_ID: kernel = fit_feat_kernel<LFM_LOSS_LOGISTIC_ID, NC, false, true>; break;
    case LFM_LOSS_WARP_ID: kernel = fit_feat_kernel<LFM_LOSS_WARP_ID, NC, false, true>; break;
    case LFM_LOSS_BPR_ID: kernel = fit_feat_kernel<LFM_LOSS_BPR_ID, NC, false, true>; break;
    case LFM_LOSS_WARP_KOS_ID: kernel = fit_feat_kernel<LFM_LOSS_WARP_KOS_ID, NC, false, true>; break;
    default: return hipErrorInvalidValue;
    }
    if (!kernel) switch (loss) {
    case LFM_LOSS_LOGISTIC_ID: kernel = fit_feat_kernel<LFM_LOSS_LOGISTIC_ID, NC>; break;
    case LFM_LOSS_WARP_ID: kernel = fit_feat_kernel<LFM_LOSS_WARP_ID, NC>; break;
    case LFM_LOSS_BPR_ID: kernel = fit_feat_kernel<LFM_LOSS_BPR_ID, NC>; break;
    case LFM_LOSS_WARP_KOS_ID: kernel = fit_feat_kernel<LFM_LOSS_WARP_KOS_ID, NC>; break;
    default: return hipErrorInvalidValue;
    }
    if (cus > 0) {  // only resident workgroups: every wavefront runs its grid-stride loop from the start
        const int per_cu = occupancy_cached(kernel, block, smem);
        if (per_cu > 0) grid = std::min(grid, per_cu * cus);
    }
    if (grid_used) *grid_used = grid;
    kernel<<<grid, block, smem, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_fit_feat_wide(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                                int *grid_used);

// the ADA instantiations (feat_kernels_ada.hip): the adadelta schedule, with and without lazy regularisation, d <= 128
template <int NC>
inline hipError_t launch_feat_ada_nc(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                                     int *grid_used)
{
    void (*kernel)(FitArgs) = nullptr;
    const bool reg = a.item_alpha != 0.0 || a.user_alpha != 0.0;
    if (reg) switch (loss) {
    case LFM_LOSS_LOGISTIC_ID: kernel = fit_feat_kernel<LFM_LOSS_LOGISTIC_ID, NC, false, true, false, true>; break;
    case LFM_LOSS_WARP_ID: kernel = fit_feat_kernel<LFM_LOSS_WARP_ID, NC, false, true, false, true>; break;
    case LFM_LOSS_BPR_ID: kernel = fit_feat_kernel<LFM_LOSS_BPR_ID, NC, false, true, false, true>; break;
    case LFM_LOSS_WARP_KOS_ID: kernel = fit_feat_kernel<LFM_LOSS_WARP_KOS_ID, NC, false, true, false, true>; break;
    default: return hipErrorInvalidValue;
    }
    else switch (loss) {
    case LFM_LOSS_LOGISTIC_ID: kernel = fit_feat_kernel<LFM_LOSS_LOGISTIC_ID, NC, false, false, false, true>; break;
    case LFM_LOSS_WARP_ID: kernel = fit_feat_kernel<LFM_LOSS_WARP_ID, NC, false, false, false, true>; break;
    case LFM_LOSS_BPR_ID: kernel = fit_feat_kernel<LFM_LOSS_BPR_ID, NC, false, false, false, true>; break;
    case LFM_LOSS_WARP_KOS_ID: kernel = fit_feat_kernel<LFM_LOSS_WARP_KOS_ID, NC, false, false, false, true>; break;
    default: return hipErrorInvalidValue;
    }
    if (cus > 0) {
        const int per_cu = occupancy_cached(kernel, block, smem);
        if (per_cu > 0) grid = std::min(grid, per_cu * cus);
    }
    if (grid_used) *grid_used = grid;
    kernel<<<grid, block, smem, st>>>(a);
    return hipGetLastError();
}

// the HOT instantiations (feat_kernels_hot.hip): adagrad, no regularisation, d <= 128
template <int NC>
inline hipError_t launch_feat_hot_nc(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                                     int *grid_used)
{
    void (*kernel)(FitArgs) = nullptr;
    switch (loss) {
    case LFM_LOSS_LOGISTIC_ID: kernel = fit_feat_kernel<LFM_LOSS_LOGISTIC_ID, NC, false, false, true>; break;
    case LFM_LOSS_WARP_ID: kernel = fit_feat_kernel<LFM_LOSS_WARP_ID, NC, false, false, true>; break;
    case LFM_LOSS_BPR_ID: kernel = fit_feat_kernel<LFM_LOSS_BPR_ID, NC, false, false, true>; break;
    case LFM_LOSS_WARP_KOS_ID: kernel = fit_feat_kernel<LFM_LOSS_WARP_KOS_ID, NC, false, false, true>; break;
    default: return hipErrorInvalidValue;
    }
    if (cus > 0) {
        const int per_cu = occupancy_cached(kernel, block, smem);
        if (per_cu > 0) grid = std::min(grid, per_cu * cus);
    }
    if (grid_used) *grid_used = grid;
    kernel<<<grid, block, smem, st>>>(a);
    return hipGetLastError();
}

}  // namespace lfm
