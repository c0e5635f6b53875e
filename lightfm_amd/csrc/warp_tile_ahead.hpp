// warp_tile_ahead.hpp -- the steady-state variant of the lane-group WARP tile kernel (warp_tile_kernel.hpp,
// DMA4 layout: 16 lanes per row, four interactions per wavefront pass, rows memory -> LDS by
// global_load_lds_dwordx4) with the gather of pass t + 1 issued INSIDE pass t.
// PYX = /root/reference/lightfm/_lightfm_fast.pyx.template (fit_warp, PYX:784-912)
//
// A pass of the tile kernel is a chain of dependent round trips: gather (user, positive and all candidate
// rows) -> scoring -> Bloom probe + accumulator rows -> cell arithmetic -> atomics.  With 12 wavefronts per CU
// (LDS bound) that chain, not a bandwidth, sets the rate.  Here the chain is one round trip shorter.  After the
// scoring pass the tile is needed for one thing only: coordinate `lane` of the user row, the positive row and
// the chosen negative's row, for the update.  The choice is the first violator unless that candidate is one
// of the user's positives (PYX:878), so those three cells of every interaction with a violator are copied to
// registers speculatively -- and the rows of pass t + 1 (its record was prefetched two passes ago, its
// candidates follow from the position's PRNG stream alone) are requested right then, BEFORE pass t probes the
// Bloom filter, fetches its accumulator rows, evaluates the float64 cell arithmetic and publishes.  Requests
// return in order, so pass t's probe / accumulator round trip and pass t + 1's gather are one wait instead of
// two.  When the choice is a later violator (the first one was a positive) its row is re-read from the table
// -- weights "as of now", a legal Hogwild read; identical in every sequential or conflict-free execution.
// Biases travel through LDS too (global_load_lds_dword), so nothing loaded is carried in a VGPR across the
// loop's back edge.
//
// Scope: d <= 64 (a multiple of 4), adagrad, no L2 regularisation, max_sampled == NBF (one batch), parallel
// mode.  Everything else runs fit_warp_tile_kernel.  Semantics are those of fit_warp_tile_kernel: the
// scoring, sampling, in_positives (+ Bloom pre-filter) and update code is the same, in the same order.
#pragma once
#include "warp_tile_kernel.hpp"

namespace lfm {

// Where an item row lives.  SHARDED = false: one table (base + row * d).  SHARDED = true: device.hpp: ItemShards --
// the row's range j = row / rows_per_shard picks the owner's base pointer (a select chain over <= 8 pointers: a
// dynamically indexed kernel-argument array would go through scratch memory), the rest of the id is the row inside.
template <bool SHARDED>
struct ItemAddr {
    const ItemShards &sh;
    float *W0, *G0, *b0, *bG0;
    const float *bread0;
    int d;
    int bstride = 1;  // floats between rows of bread0 (2: the (b, bG) pair table)
    __device__ __forceinline__ static float *pick(float *const (&p)[8], uint32_t j)
    {
        float *a01 = (j & 1u) ? p[1] : p[0], *a23 = (j & 1u) ? p[3] : p[2];
        float *a45 = (j & 1u) ? p[5] : p[4], *a67 = (j & 1u) ? p[7] : p[6];
        float *lo = (j & 2u) ? a23 : a01, *hi = (j & 2u) ? a67 : a45;
        return (j & 4u) ? hi : lo;
    }
    __device__ __forceinline__ void split(int row, uint32_t &j, uint32_t &local) const
    {
        j = __umulhi((uint32_t)row, sh.magic);
        int r = row - (int)(j * sh.rows_per_shard);
        if (r < 0) {  // magic = floor(2^32 / n) + 1 overshoots by at most one
            --j;
            r += (int)sh.rows_per_shard;
        }
        local = (uint32_t)r;
    }
    // row address for the gathers: rows are >= 0 and a table below 4 GB is addressed with ONE 32-bit multiply
    __device__ __forceinline__ const float *Wg(int row, bool small) const
    {
        if constexpr (!SHARDED) {
            if (small) return W0 + (uint32_t)row * (uint32_t)d;
            return W0 + (size_t)(uint32_t)row * (size_t)d;
        } else {
            return W(row);
        }
    }
    __device__ __forceinline__ float *W(int row) const
    {
        if constexpr (!SHARDED) return W0 + (size_t)row * d;
        uint32_t j, l;
        split(row, j, l);
        return pick(sh.W, j) + (size_t)l * d;
    }
    __device__ __forceinline__ float *G(int row) const
    {
        if constexpr (!SHARDED) return G0 + (size_t)row * d;
        uint32_t j, l;
        split(row, j, l);
        return pick(sh.G, j) + (size_t)l * d;
    }
    __device__ __forceinline__ float *b(int row) const
    {
        if constexpr (!SHARDED) return b0 + row;
        uint32_t j, l;
        split(row, j, l);
        return pick(sh.b, j) + l;
    }
    __device__ __forceinline__ float *bG(int row) const
    {
        if constexpr (!SHARDED) return bG0 + row;
        uint32_t j, l;
        split(row, j, l);
        return pick(sh.bG, j) + l;
    }
    // the bias the SCORING reads: the launch's cached snapshot, or -- sharded -- the owner's live cell
    __device__ __forceinline__ const float *bscore(int row) const
    {
        if constexpr (!SHARDED) return bread0 + (size_t)row * bstride;
        return b(row);
    }
};

// USTORE (FitArgs::user_store, decided by the session): the USER row of an update is written with plain stores instead of
// float atomics (uncached tables only: a store is then visible to every XCD).  With identity user features a user's row
// is touched by that user's interactions alone, so what a plain read-modify-write can lose is one of two updates of the
// same user that are in flight at the same time -- the reference's own Hogwild race -- and a third of C2's float
// atomics (2 x 64 of 390 per update) leave the atomic unit: C2 1.25 -> 1.44 G interactions/s in a one-box A/B,
// precision@10 0.1743 against 0.1747 (8 seeds each; profiles/r05_visit_f.txt).
// VEC (round 6): floats of a row per lane.  4 = rows of up to 64 floats (global_load_lds_dwordx4).  1 = rows of up to 16
// floats (global_load_lds_dword): the reference's DEFAULT width (no_components = 10: rows of 12 floats on the device) --
// the tile of an interaction then takes a quarter of the LDS (15 KB instead of 52 KB per workgroup), so that registers,
// not LDS, bound the residency: a kernel whose passes are chains of dependent round trips (0.83 KB per interaction: it
// never waits for bandwidth) runs faster with every wavefront more per CU.
#ifndef LFM_NARROW_BLOCKS
#define LFM_NARROW_BLOCKS 4  // workgroups per CU the VEC = 1 instantiation is compiled for (5: 96 VGPRs with 20-32 B of scratch)
#endif
template <int NBF, bool SHARDED = false, bool USTORE = false, int VEC = 4>
__global__ __launch_bounds__(256, VEC == 1 ? LFM_NARROW_BLOCKS : 3) void fit_warp_tile_ahead_kernel(FitArgs a)
{
    static_assert(VEC == 4 || VEC == 1, "rows of up to 64 or up to 16 floats");
    constexpr int LPR = 16, NG = 4;
    constexpr unsigned long long GM = 0xffffull;
    static_assert(NBF >= 1 && NBF <= LPR - 1, "one batch of candidates, one per lane of a group");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), wib = uni((int)(threadIdx.x >> 6));
    const int g = lane / LPR, p = lane % LPR, gbase = g * LPR;
    const int d = a.m.d;
    constexpr int RG = NBF + 1;                      // tile rows per group: the positive + NBF candidates
    constexpr int KS = NG * LPR * VEC + 4;           // floats between consecutive rows of a group
    constexpr int GS = LPR * VEC;                    // between the groups' first rows
    constexpr int UB = RG * KS;                      // first user row
    constexpr int US = LPR * VEC + 4;                // between user rows
    constexpr int BB = UB + NG * US;                 // bias slots: [BB + l] item bias of lane l's row, [BB + 64 + l] user bias
    constexpr int WAVE_FLOATS = BB + 2 * WAVE;
    float *tile = smem + (size_t)wib * WAVE_FLOATS;
    float *vrows = tile + (size_t)g * GS;
    float *urow = tile + UB + (size_t)g * US;
    const bool pc = VEC * p < d;
    const float *Wu = a.m.W[1];
    const float *bu_tab = a.b_read[1];
    const ItemAddr<SHARDED> item{a.shards, a.m.W[0], a.m.G[0], a.m.b[0], a.m.bG[0], a.b_read[0], d, a.b_read_stride[0]};
    // bias cells as (b, bG) pairs of one line (FitArgs::bb): lanes 16 g + 0 / 1 / 2 hold the W cell of the positive item / the
    // negative item / the user, lanes 16 g + 3 / 4 / 5 the accumulator cell next to it -- one load and ONE publication
    // instruction per pass touch both cells of a row with one line operation (the separate tables: two)
    const bool paired = !SHARDED && a.bb[0] != nullptr;
    const bool small_items = (uint64_t)a.itf.rows * (uint64_t)d < (1ull << 30);  // item table below 4 GB: 32-bit row offsets
    const uint32_t n_items = (uint32_t)a.itf.rows, magic = a.n_items_magic;
    const uint32_t base_seed = a.seeds[0];
    const Hyper h{0, a.m.lr, a.m.rho, a.m.eps};
    const uint32_t *bloom = a.bloom;

    // lane p needs the position's stream after min(p, NBF) draws: (A^k, C (A^(k-1) + ... + 1)) mod 2^32
    uint32_t lcgA = 1u, lcgC = 0u;
    for (int j = 0; j < min(p, NBF); ++j) {
        lcgA *= 1103515245u;
        lcgC = lcgC * 1103515245u + 12345u;
    }

    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6) * NG;
    const int32_t *indptr = a.pos.indptr, *indices = a.pos.indices;
    float *WuW = a.m.W[1], *Gu = a.m.G[1];

    // The whole first-batch gather of one pass: straight-line, every lane takes part (the record of a
    // group past the end of the launch is zero: rows 0 are fetched and never used).  Returns the lane's
    // candidate item and its stream state.
    auto dma_piece = [&](const float *g_, float *lds_base) {  // VEC floats per lane, memory -> LDS
        if constexpr (VEC == 4) dma_lane_x4(g_, lds_base);
        else dma_lane_dword(g_, lds_base);
    };
    auto issue_gather = [&](int user, int pos, int64_t i, int &myitem, uint32_t &s) {
        const uint32_t state = position_seed(base_seed, (uint64_t)i);
        s = lcgA * state + lcgC;
        myitem = (p == 0) ? pos : fast_mod(draw(s), n_items, magic);  // PYX:860-861
#pragma nounroll
        for (int gg = 0; gg < NG; ++gg) {  // (the LDS base of an instruction is M0: wave-uniform)
            float *ub = tile + UB + __builtin_amdgcn_readfirstlane(gg) * (US - LPR * VEC);
            if (g == gg && pc) dma_piece(Wu + (size_t)user * d + VEC * p, ub);
        }
        if (pc) dma_piece(item.Wg(pos, small_items) + VEC * p, tile);  // row 0 of every group
#pragma unroll
        for (int k = 1; k <= NBF; ++k) {
            const int neg = row_bcast(myitem, k);
            if (pc) dma_piece(item.Wg(neg, small_items) + VEC * p, tile + (size_t)k * KS);
        }
        dma_lane_dword(item.bscore(myitem), tile + BB);
        dma_lane_dword(bu_tab + (size_t)user * a.b_read_stride[1], tile + BB + WAVE);
    };

    // record pipeline, three passes deep (as fit_warp_tile_kernel)
    int64_t ib = a.begin + gw * NG;
    int4 cur = make_int4(0, 0, 0, 0), nxt = cur;
    int c_lo = 0, c_hi = 0, row2 = 0;
    if (ib + g < a.end) {
        cur = a.recs[guard_row(a, a.shuffle[ib + g])];
        c_lo = indptr[cur.x];
        c_hi = indptr[cur.x + 1];
    }
    if (ib + stride + g < a.end) nxt = a.recs[guard_row(a, a.shuffle[ib + stride + g])];
    if (ib + 2 * stride + g < a.end) row2 = a.shuffle[ib + 2 * stride + g];

    int myitem = 0;
    uint32_t s = 0;
    if (ib < a.end) issue_gather(cur.x, cur.y, ib + g, myitem, s);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the prologue's gather (the builtin, not inline asm: the compiler then knows)

    for (; ib < a.end; ib += stride) {
        const int64_t i = ib + g;
        const bool in = i < a.end;
        // (the rows of this pass were requested right after the scoring of the previous pass and have been waited for
        // there, together with that pass's accumulator rows: no wait here, in particular not for the acknowledgements
        // of the atomics the previous pass issued last)
        wave_sync();
        const int c_user = cur.x, c_pos = cur.y;
        const float c_y = __int_as_float(cur.z), c_w = __int_as_float(cur.w);
        const bool act = in && (c_y > 0.0f);  // PYX:831-832, before any RNG use
        int sampled = 0, chosen = -1;

        // ---- scoring: lane r of a group computes the sequential dot of tile row r (PYX:320-334)
        unsigned long long vm = 0ull;
        bool viol = false;
        if (__ballot(act) != 0ull) {
            const float bi = tile[BB + lane], bu = tile[BB + WAVE + lane];
            const bool rowlane = act && p <= NBF;
            float score = 0.0f;
            if (rowlane) score = row_dot<false>(urow, vrows + (size_t)p * KS, d, bu, bi, 1.0f, 1.0f);
            const double pp = (double)__shfl(score, gbase, WAVE);
            // PYX:875 compares doubles: negative_prediction > positive_prediction - 1
            viol = act && p >= 1 && p <= NBF && ((double)score > pp - 1.0);
            vm = (__ballot(viol) >> gbase) & GM;
        }
        // The first violator is almost always the choice (PYX:878: unless it is one of the user's positives).  What
        // its update reads from the tile -- coordinate `lane` of the user, positive and that candidate's row -- goes
        // to registers now; the tile is then free for the next pass.
        const int r1 = vm != 0ull ? (__ffsll((long long)vm) - 1) : 0;
        const int spec_cand = __shfl(myitem, gbase + r1, WAVE);
        const unsigned long long specm = __ballot(act && vm != 0ull && p == 0);
        float cU[NG], cP[NG], cN[NG];
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) {
            cU[gg] = cP[gg] = cN[gg] = 0.0f;
            if ((specm >> (gg * LPR)) & 1ull) {
                const int cr = __builtin_amdgcn_readlane(r1, gg * LPR);
                const int cc = lane < d ? lane : 0;
                cU[gg] = (tile + UB + (size_t)gg * US)[cc];
                cP[gg] = (tile + (size_t)gg * GS)[cc];
                cN[gg] = (tile + (size_t)gg * GS + (size_t)cr * KS)[cc];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of this pass's tile has completed
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);

        // ---- the gather of the NEXT pass (a dummy one after the last pass keeps the code straight-line)
        int myitem_n;
        uint32_t s_n;
        issue_gather(nxt.x, nxt.y, i + stride, myitem_n, s_n);
        // ... and the records of the passes after it (requested here, not at the top of the pass: nothing loaded is
        // then pending while the scoring pass runs, and the wait below covers them)
        int n_lo = 0, n_hi = 0, row3 = 0;
        int4 rec2 = make_int4(0, 0, 0, 0);
        if (i + stride < a.end) {
            n_lo = indptr[nxt.x];
            n_hi = indptr[nxt.x + 1];
        }
        if (i + 2 * stride < a.end) rec2 = a.recs[guard_row(a, row2)];
        if (i + 3 * stride < a.end) row3 = a.shuffle[i + 3 * stride];
        __builtin_amdgcn_sched_barrier(0);

        // ---- in_positives (Bloom pre-filter probed by the violators, device.hpp: Bloom) and the accumulator rows of
        // the first violator's update: one round trip, behind the gather's
        float gP[NG], gN[NG], gU[NG];
        auto load_rows = [&](int gg, int user, int pos, int neg, bool only_neg) {
            const size_t bu_ = (size_t)user * d;
            unsigned cc = lane < d ? (unsigned)lane : 0u;
            asm volatile("" : "+v"(cc));  // uniform row base + a lane offset the compiler cannot hoist
            gN[gg] = item.G(neg)[cc];
            if (only_neg) cN[gg] = item.W(neg)[cc];  // the choice is a later violator: its row is re-read (weights as of now)
            if (!only_neg) {
                gP[gg] = item.G(pos)[cc];
                gU[gg] = (Gu + bu_)[cc];
            }
        };
        // The bias cells (PYX:571-599) of ALL interactions of the pass are one lane each: lane 16 g + 0 = positive item of
        // group g, + 1 = its negative, + 2 = its user -- one cell evaluation and two publications per pass instead of
        // one of each per updating interaction.
        const bool has_viol = act && vm != 0ull;
        float obW = 0.0f, obG = 1.0f;
        auto bias_ptrs = [&](int neg, float *&bWp, float *&bGp) {
            if (p == 2) {
                bWp = a.m.b[1] + c_user;
                bGp = a.m.bG[1] + c_user;
            } else {
                const int irow = p == 0 ? c_pos : neg;
                bWp = item.b(irow);
                bGp = item.bG(irow);
            }
        };
        // (b, bG) pair table: the cell of lane p < 6 -- role p % 3 (positive item, negative item, user), part p / 3 (W, G)
        auto pair_ptr = [&](int neg) -> float * {
            const int role = p < 3 ? p : p - 3;
            const int row = role == 2 ? c_user : (role == 0 ? c_pos : neg);
            return a.bb[role == 2 ? 1 : 0] + 2 * (size_t)row + (p < 3 ? 0 : 1);
        };
        if (__ballot(act) != 0ull) {
            const uint32_t bh = Bloom::mix((uint32_t)myitem);
            uint32_t bword = 0xffffffffu;
            if (bloom && viol) bword = bloom[Bloom::word(bh, c_lo, c_hi)];
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
                if ((specm >> (gg * LPR)) & 1ull)
                    load_rows(gg, __builtin_amdgcn_readlane(c_user, gg * LPR), __builtin_amdgcn_readlane(c_pos, gg * LPR),
                              __builtin_amdgcn_readlane(spec_cand, gg * LPR), false);
            }
            const uint32_t bmask = Bloom::mask(bh);
            const int maybe_pos = ((bword & bmask) == bmask) ? 1 : 0;
            if (paired) {
                if (has_viol && p < 6) obW = *pair_ptr(spec_cand);  // (this lane's cell; the pairs meet after the wait)
            } else if (has_viol && p < 3) {  // bias cells of the speculated update
                float *bWp, *bGp;
                bias_ptrs(spec_cand, bWp, bGp);
                obW = *bWp;
                obG = *bGp;
            }
            int used = NBF;
            while (true) {
                const bool part = act && chosen < 0 && vm != 0ull;
                if (__ballot(part) == 0ull) break;
                const int r = part ? (__ffsll((long long)vm) - 1) : 0;
                if (part) vm &= vm - 1ull;
                const int cand = __shfl(myitem, gbase + r, WAVE);
                const bool ask = part && __shfl(maybe_pos, gbase + r, WAVE) != 0;
                bool found = false;
                if (__ballot(ask) != 0ull) found = group_in_positives<LPR>(indices, cand, c_lo, c_hi, ask, gbase, p);
                c3 += (uint32_t)__popcll(__ballot(part && p == 0));  // PYX:878-879: the draw still counts
                if (part && !found) {
                    chosen = cand;
                    used = r;
                }
            }
            if (act) sampled = used;
            c0 += (uint32_t)__popcll(__ballot(act && p == 0));
            c2 += (uint32_t)__popcll(__ballot(act && chosen >= 0 && p == 0));
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) c1 += (uint32_t)__builtin_amdgcn_readlane(sampled, gg * LPR);
        }

        // ONE wait for everything requested so far: the probe and the accumulator rows of this pass (consumed below)
        // and, older than those, the rows of the next pass -- which therefore never waits for the atomics issued below
        // (the builtin, not inline asm: the compiler's own wait-count bookkeeping then knows that nothing loaded is
        // outstanding, and places no conservative vmcnt(0) -- which would wait for the atomics -- at the loop's top)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt unconstrained
        double lossd = 0.0;
        if (act && chosen >= 0) {
            lossd = (double)c_w * a.logtab[sampled];  // PYX:881-885, log from host libm
            if (lossd > MAX_LOSS) lossd = MAX_LOSS;
        }
        const unsigned long long upd = __ballot(act && chosen >= 0 && p == 0);
        if (in && p == 0) {
            if (a.neg_log) a.neg_log[i] = chosen;
            if (a.sampled_log) a.sampled_log[i] = sampled;
        }

        // ---- updates: float64 cell arithmetic (PYX:416-449) and atomic publication, one interaction after the other
        const bool bupd = act && chosen >= 0 && p < 3;
        if (upd != 0ull) {
            {   // the first violator was a positive and a later one is the choice: its bias cell (lane 16 g + 1)
                const bool bre = act && chosen >= 0 && chosen != spec_cand && (paired ? (p == 1 || p == 4) : p == 1);
                if (__ballot(bre) != 0ull) {
                    if (bre) {
                        if (paired) obW = *pair_ptr(chosen);
                        else {
                            float *bWp, *bGp;
                            bias_ptrs(chosen, bWp, bGp);
                            obW = *bWp;
                            obG = *bGp;
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                }
            }
            // ... and its rows
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
                if ((upd >> (gg * LPR)) & 1ull) {
                    const int neg = __builtin_amdgcn_readlane(chosen, gg * LPR);
                    if (neg != __builtin_amdgcn_readlane(spec_cand, gg * LPR)) {
                        load_rows(gg, __builtin_amdgcn_readlane(c_user, gg * LPR), __builtin_amdgcn_readlane(c_pos, gg * LPR), neg, true);
                        // waited for in the block that requested them (rare path): otherwise the compiler has to assume
                        // them pending at the loop's top and guards the scoring pass's registers with a vmcnt(0) -- which
                        // would wait for the previous pass's atomics
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                    }
                }
            }
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
                if ((upd >> (gg * LPR)) & 1ull) {
                    const int user = __builtin_amdgcn_readlane(c_user, gg * LPR);
                    const int pos = __builtin_amdgcn_readlane(c_pos, gg * LPR);
                    const int neg = __builtin_amdgcn_readlane(chosen, gg * LPR);
                    const double loss = read_laned(lossd, gg * LPR);
                    const size_t bu_ = (size_t)user * d;
                    float *const wP = item.W(pos), *const wN = item.W(neg), *const aP = item.G(pos), *const aN = item.G(neg);
                    const float Ur = cU[gg], Pr = cP[gg], Nr = cN[gg];
                    const double u = (double)Ur;
                    const double df = (double)__fsub_rn(Nr, Pr);  // float32 subtraction, PYX:634-635
                    float nWP, nGP, nWN, nGN, nWU, nGU, nM;
                    double lr;
                    // (the float64 cell as it stands: cell_math_adagrad -- the same cell without root and quotient, which the
                    // row-stream and slice kernels run -- measured -1 % here; this kernel waits for memory, not for its ALUs:
                    // profiles/r06_fast_cell_ab.txt)
                    cell_math(Pr, gP[gg], 0.0f, 1.0, -loss * u, h, 0.0, nWP, nGP, nM, lr);
                    cell_math(Nr, gN[gg], 0.0f, 1.0, loss * u, h, 0.0, nWN, nGN, nM, lr);
                    cell_math(Ur, gU[gg], 0.0f, 1.0, loss * df, h, 0.0, nWU, nGU, nM, lr);
                    asm volatile("" : "+v"(nWP), "+v"(nGP), "+v"(nWN), "+v"(nGN), "+v"(nWU), "+v"(nGU));
                    if (lane < d) {
                        // publication: new - old by global_atomic_add_f32, unconditionally (this variant runs update_mode
                        // 0 only; a zero delta is added as such: the per-publication mode switch and zero test cost a
                        // dozen scalar instructions and three branches each, eight times per interaction)
                        unsigned cq = (unsigned)lane;
                        asm volatile("" : "+v"(cq));
                        atomicAdd(wP + cq, __fsub_rn(nWP, Pr));
                        atomicAdd(aP + cq, __fsub_rn(nGP, gP[gg]));
                        atomicAdd(wN + cq, __fsub_rn(nWN, Nr));
                        atomicAdd(aN + cq, __fsub_rn(nGN, gN[gg]));
                        if constexpr (USTORE) {
                            WuW[bu_ + cq] = nWU;
                            Gu[bu_ + cq] = nGU;
                        } else {
                            atomicAdd(WuW + bu_ + cq, __fsub_rn(nWU, Ur));
                            atomicAdd(Gu + bu_ + cq, __fsub_rn(nGU, gU[gg]));
                        }
                    }
                }
            }
            // the pass's bias cells, all interactions at once
            {
                float bnW, bnG, bnM;
                double blr;
                const float cell_own = obW;
                if (paired) obG = __shfl(obW, lane + 3, WAVE);  // lanes p < 3: the accumulator cell from the lane that loaded it
                cell_math(obW, obG, 0.0f, 1.0, p == 0 ? -lossd : lossd, h, 0.0, bnW, bnG, bnM, blr);
                if (paired) {
                    // lanes p < 3 publish the W cell, lanes 3..5 the G cell the lane three below computed: one instruction
                    const float nG_up = __shfl(bnG, lane - 3, WAVE);
                    const bool mine = act && chosen >= 0 && p < 6;
                    const float nv = p < 3 ? bnW : nG_up;
                    if (mine) {
                        float *cp = pair_ptr(chosen);
                        if (USTORE && (p == 2 || p == 5)) *cp = nv;
                        else atomicAdd(cp, __fsub_rn(nv, cell_own));
                    }
                } else
                if (bupd) {
                    float *bWp, *bGp;
                    bias_ptrs(chosen, bWp, bGp);
                    if (USTORE && p == 2) {
                        // the user's bias cells with the user's row: touched by that user's interactions alone.  (A 4-byte
                        // atomic costs the atomic units a LINE operation like a 32-float half row: the six bias cells were 6
                        // of an update's 10 line operations at the reference's default width, profiles/r06_narrow_tile_ab.txt)
                        *bWp = bnW;
                        *bGp = bnG;
                    } else {
                        atomicAdd(bWp, __fsub_rn(bnW, obW));
                        atomicAdd(bGp, __fsub_rn(bnG, obG));
                    }
                }
            }
        }
        cur = nxt;
        c_lo = n_lo;
        c_hi = n_hi;
        nxt = rec2;
        row2 = row3;
        myitem = myitem_n;
        s = s_n;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the dummy gather of the last pass

    if (lane == 0) {
        if (c0) atomicAdd(a.counters + 0, (unsigned long long)c0);
        if (c1) atomicAdd(a.counters + 1, (unsigned long long)c1);
        if (c2) atomicAdd(a.counters + 2, (unsigned long long)c2);
        if (c3) atomicAdd(a.counters + 3, (unsigned long long)c3);
    }
}

// LDS bytes per 256-thread workgroup (four wavefronts)
template <int NBF, int VEC = 4>
constexpr size_t tile_ahead_smem()
{
    return (size_t)WAVES_PER_BLOCK * ((size_t)(NBF + 1) * (4 * 16 * VEC + 4) + 4 * (16 * VEC + 4) + 2 * WAVE) * sizeof(float);
}

}  // namespace lfm
