// warp_tile_narrow.hpp -- the steady-state WARP tile kernel for NARROW models: rows of up to 16 floats on the device, i.e.
// no_components <= 16 -- the reference's DEFAULT is 10 (LFM:191: rows of 12 floats).  PYX = _lightfm_fast.pyx.template.
//
// Why a kernel of its own.  fit_warp_tile_ahead_kernel (warp_tile_ahead.hpp) executes ~350 instructions per interaction
// whatever the width, and its three wavefronts per SIMD issue ~88 % of the time: at d = 10 it runs 1.7 G interactions/s
// on 0.83 KB per interaction (0.18 of the HBM roofline), and neither more interactions in flight (a VEC = 1 layout:
// measured slower) nor fewer atomics move it (profiles/r06_narrow_tile_ab.txt).  What a narrow model leaves idle is the
// LANES: a 12-float row occupies 3 of the 16 lanes of a group in the gathers and 12 of 64 in the update.  Here a 16-lane
// group works on TWO interactions per pass (eight per wavefront pass), and every phase is laid out for 16-float rows:
//
//   gather   a row is 4 lanes x 16 bytes, so ONE global_load_lds_dwordx4 deposits FOUR rows per group: the twelve rows of
//            an interaction (positive, ten candidates, the user) are three instructions -- 8 LDS-DMA instructions per pass
//            of eight interactions (biases included) against 14 per pass of four;
//   scoring  lane p of a group computes the sequential float32 dot (PYX:320-334) of row p of BOTH its interactions;
//   sampling / in_positives / Bloom pre-filter: per interaction, as in the wide kernel (same streams, same order);
//   update   lane (g, c) owns coordinate c of group g's interaction: the four groups update at once (the wide kernel
//            takes them one after the other), twice per pass.
// Semantics are the wide kernel's: same PRNG stream per position, same summation order, the first violator that is not
// one of the user's positives, the reference's float64 cell, publication by float atomics (user rows and user bias cells
// by plain stores under USTORE).  The gather of pass t + 1 is issued inside pass t (after the scoring), as there.
// Scope: parallel mode, identity features, adagrad, no L2 penalty, max_sampled == NBF == 10, 4 <= d <= 16 (a multiple of 4),
// item tables below 4 GB, no owner-sharding.
#pragma once
#include "warp_tile_kernel.hpp"

namespace lfm {

// RP: W and G of a row live in ONE 128-byte line of a pair table [W(16) | G(16)] (FitArgs::rp; packed by the session for the
// launches of this kernel): the gathers read the W half, the update's accumulator cells come from the line the gather
// already fetched, and BOTH halves of a row are published by one instruction -- lanes of an even group carry their own W
// deltas while the lanes of the odd group next to them carry the even group's G deltas (and the other way round in a second
// instruction): two line operations per pair of updating rows instead of four.
// BIN (with RP, d <= 12: the reference's default width): the bias cells live in the row's line too -- [W(d) | b | .. || G(d) | bG
// | ..], slot d of each half -- so the gather brings the bias with the row (no bias requests), lane d of a group runs the bias
// cell through the same cell arithmetic (gradient -loss / loss / loss, PYX:571-599) and its deltas leave with the row's: an
// update is THREE line operations (positive, negative, user) instead of six.
template <int NBF, bool USTORE = false, bool RP = false, bool BIN = false>
__global__ __launch_bounds__(256, 4) void fit_warp_tile_narrow_kernel(FitArgs a)
{
    constexpr int LPR = 16, NG = 4, Q = 2;
    constexpr unsigned long long GM = 0xffffull;
    static_assert(NBF == 10, "twelve rows per interaction: the positive, ten candidates, the user");
    static_assert(RP || !BIN, "the bias cells ride in the row pairs");
    constexpr int KU = NBF + 1;                 // the user's row index (11)
    constexpr int QUAD = NG * 4 * 16;           // floats one LDS-DMA instruction deposits: 4 rows of 16 floats per group
    constexpr int BB = Q * 3 * QUAD;            // bias slots: [BB + q * 64 + lane]
    constexpr int WAVE_FLOATS = BB + Q * WAVE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), wib = uni((int)(threadIdx.x >> 6));
    const int g = lane / LPR, p = lane % LPR, gbase = g * LPR;
    const int d = a.m.d;
    float *tile = smem + (size_t)wib * WAVE_FLOATS;
    // row k of this lane's group, interaction q
    auto row_of = [&](int q, int k) -> float * { return tile + (q * 3 + (k >> 2)) * QUAD + g * 64 + (k & 3) * 16; };
    // row r of a side: RS floats apart; its accumulator cells GO floats after its embedding cells (RP: the same line)
    const int RS = RP ? 32 : d;
    float *WiW = RP ? a.rp[0] : a.m.W[0], *WuW = RP ? a.rp[1] : a.m.W[1];
    float *Gi = RP ? a.rp[0] + 16 : a.m.G[0], *Gu = RP ? a.rp[1] + 16 : a.m.G[1];
    const float *Wi = WiW, *Wu = WuW;
    const float *bi_tab = a.b_read[0], *bu_tab = a.b_read[1];
    const uint32_t n_items = (uint32_t)a.itf.rows, magic = a.n_items_magic;
    const uint32_t base_seed = a.seeds[0];
    const Hyper h{0, a.m.lr, a.m.rho, a.m.eps};
    const uint32_t *bloom = a.bloom;
    const int32_t *indptr = a.pos.indptr, *indices = a.pos.indices;
    const int piece = p & 3, prow = p >> 2;     // this lane's 16-byte piece and row inside a quad
    const bool pc = BIN ? 4 * piece <= d : 4 * piece < d;  // (BIN: the piece that holds slot d as well)

    // lane p needs the position's stream after min(p, NBF) draws: (A^k, C (A^(k-1) + ... + 1)) mod 2^32
    uint32_t lcgA = 1u, lcgC = 0u;
    for (int j = 0; j < min(p, NBF); ++j) {
        lcgA *= 1103515245u;
        lcgC = lcgC * 1103515245u + 12345u;
    }

    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6) * NG * Q;

    // The whole gather of one interaction of every group: straight-line, every lane takes part (the record of an
    // interaction past the end of the launch is zero: rows 0 are fetched and never used).
    auto issue_gather = [&](int q, int user, int pos, int64_t i, int &myitem, uint32_t &s) {
        const uint32_t state = position_seed(base_seed, (uint64_t)i);
        s = lcgA * state + lcgC;
        myitem = (p == 0) ? pos : fast_mod(draw(s), n_items, magic);  // PYX:860-861 (lanes past NBF: unused)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int k = 4 * t + prow;  // this lane's row of the quad
            const int it = __shfl(myitem, gbase + (k <= NBF ? k : 0), WAVE);
            const float *src = (k == KU ? Wu + (size_t)user * RS : Wi + (uint32_t)it * (uint32_t)RS) + 4 * piece;  // (item table < 4 GB: one 32-bit multiply)
            if (pc) dma_lane_x4(src, tile + (q * 3 + t) * QUAD);
        }
        if constexpr (!BIN) {
            const float *bsrc = p == KU ? bu_tab + (size_t)user * a.b_read_stride[1] : bi_tab + (size_t)myitem * a.b_read_stride[0];
            if (p <= KU) dma_lane_dword(bsrc, tile + BB + q * WAVE);
        }
    };

    // record pipeline, three passes deep (as fit_warp_tile_ahead_kernel), two interactions per group
    int64_t ib = a.begin + gw * NG * Q;
    int4 cur[Q], nxt[Q];
    int c_lo[Q], c_hi[Q], row2[Q], myitem[Q];
    uint32_t s[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        cur[q] = nxt[q] = make_int4(0, 0, 0, 0);
        c_lo[q] = c_hi[q] = row2[q] = myitem[q] = 0;
        s[q] = 0u;
        const int64_t i = ib + 2 * g + q;
        if (i < a.end) {
            cur[q] = a.recs[guard_row(a, a.shuffle[i])];
            c_lo[q] = indptr[cur[q].x];
            c_hi[q] = indptr[cur[q].x + 1];
        }
        if (i + stride < a.end) nxt[q] = a.recs[guard_row(a, a.shuffle[i + stride])];
        if (i + 2 * stride < a.end) row2[q] = a.shuffle[i + 2 * stride];
    }
    if (ib < a.end) {
#pragma unroll
        for (int q = 0; q < Q; ++q) issue_gather(q, cur[q].x, cur[q].y, ib + 2 * g + q, myitem[q], s[q]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the prologue's gather

    for (; ib < a.end; ib += stride) {
        wave_sync();
        bool in[Q], act[Q], viol[Q];
        int sampled[Q], chosen[Q], r1[Q], spec_cand[Q];
        unsigned long long vm[Q];
        float cU[Q], cP[Q], cN[Q];
        const bool cell_lane = BIN ? p <= d : p < d;  // (BIN: lane d = the bias cell)
        const bool bias_lane = BIN && p == d;
        const int cc = cell_lane ? p : 0;  // this lane's coordinate in the merged update (lanes past d: idle)
        bool any_act = false;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t i = ib + 2 * g + q;
            in[q] = i < a.end;
            act[q] = in[q] && (__int_as_float(cur[q].z) > 0.0f);  // PYX:831-832, before any RNG use
            sampled[q] = 0;
            chosen[q] = -1;
            vm[q] = 0ull;
            viol[q] = false;
            any_act = any_act || (__ballot(act[q]) != 0ull);
        }
        // ---- scoring: lane r of a group computes the sequential dot of tile row r of both interactions (PYX:320-334)
        if (any_act) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const bool rowlane = act[q] && p <= NBF;
                float bi, bu;
                if constexpr (BIN) {
                    bi = p <= NBF ? row_of(q, p)[d] : 0.0f;
                    bu = row_of(q, KU)[d];
                } else {
                    bi = tile[BB + q * WAVE + lane];
                    bu = tile[BB + q * WAVE + gbase + KU];
                }
                float score = 0.0f;
                if (rowlane) score = row_dot<false>(row_of(q, KU), row_of(q, p), d, bu, bi, 1.0f, 1.0f);
                const double pp = (double)__shfl(score, gbase, WAVE);
                // PYX:875 compares doubles: negative_prediction > positive_prediction - 1
                viol[q] = act[q] && p >= 1 && p <= NBF && ((double)score > pp - 1.0);
                vm[q] = (__ballot(viol[q]) >> gbase) & GM;
            }
        }
        // the first violator is almost always the choice: what its update reads from the tile goes to registers now
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            r1[q] = vm[q] != 0ull ? (__ffsll((long long)vm[q]) - 1) : 0;
            spec_cand[q] = __shfl(myitem[q], gbase + r1[q], WAVE);
            cU[q] = cP[q] = cN[q] = 0.0f;
            if (act[q] && vm[q] != 0ull) {
                cU[q] = row_of(q, KU)[cc];
                cP[q] = row_of(q, 0)[cc];
                cN[q] = row_of(q, r1[q])[cc];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of this pass's tile has completed
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);

        // ---- the gather of the NEXT pass (a dummy one after the last pass keeps the code straight-line)
        int myitem_n[Q], n_lo[Q], n_hi[Q], row3[Q];
        uint32_t s_n[Q];
        int4 rec2[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t i = ib + 2 * g + q;
            issue_gather(q, nxt[q].x, nxt[q].y, i + stride, myitem_n[q], s_n[q]);
            n_lo[q] = n_hi[q] = row3[q] = 0;
            rec2[q] = make_int4(0, 0, 0, 0);
            if (i + stride < a.end) {
                n_lo[q] = indptr[nxt[q].x];
                n_hi[q] = indptr[nxt[q].x + 1];
            }
            if (i + 2 * stride < a.end) rec2[q] = a.recs[guard_row(a, row2[q])];
            if (i + 3 * stride < a.end) row3[q] = a.shuffle[i + 3 * stride];
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- Bloom pre-filter probes of the violators, the accumulator cells of the speculated updates, their bias cells
        float gP[Q], gN[Q], gU[Q], obW[Q], obG[Q];
        int maybe_pos[Q];
        // bias cells as (b, bG) pairs of one line (FitArgs::bb): lanes 16 g + 0 / 1 / 2 hold the W cell of the positive item / the
        // negative item / the user, lanes 16 g + 3 / 4 / 5 the accumulator cell next to it (as in warp_tile_ahead.hpp)
        const bool paired = a.bb[0] != nullptr;
        auto pair_ptr = [&](int q, int neg) -> float * {
            const int role = p < 3 ? p : p - 3;
            const int row = role == 2 ? cur[q].x : (role == 0 ? cur[q].y : neg);
            return a.bb[role == 2 ? 1 : 0] + 2 * (size_t)row + (p < 3 ? 0 : 1);
        };
        auto bias_ptrs = [&](int q, int neg, float *&bWp, float *&bGp) {
            if (p == 2) {
                bWp = a.m.b[1] + cur[q].x;
                bGp = a.m.bG[1] + cur[q].x;
            } else {
                const int irow = p == 0 ? cur[q].y : neg;
                bWp = a.m.b[0] + irow;
                bGp = a.m.bG[0] + irow;
            }
        };
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            gP[q] = gN[q] = gU[q] = 1.0f;
            obW[q] = 0.0f;
            obG[q] = 1.0f;
            maybe_pos[q] = 1;
            const bool has_viol = act[q] && vm[q] != 0ull;
            if (__ballot(act[q]) != 0ull) {
                const uint32_t bh = Bloom::mix((uint32_t)myitem[q]);
                uint32_t bword = 0xffffffffu;
                if (bloom && viol[q]) bword = bloom[Bloom::word(bh, c_lo[q], c_hi[q])];
                if (has_viol && cell_lane) {
                    gP[q] = (Gi + (size_t)cur[q].y * RS)[cc];
                    gN[q] = (Gi + (size_t)spec_cand[q] * RS)[cc];
                    gU[q] = (Gu + (size_t)cur[q].x * RS)[cc];
                }
                if constexpr (BIN) {
                } else if (paired) {
                    if (has_viol && p < 6) obW[q] = *pair_ptr(q, spec_cand[q]);
                } else if (has_viol && p < 3) {
                    float *bWp, *bGp;
                    bias_ptrs(q, spec_cand[q], bWp, bGp);
                    obW[q] = *bWp;
                    obG[q] = *bGp;
                }
                const uint32_t bmask = Bloom::mask(bh);
                maybe_pos[q] = ((bword & bmask) == bmask) ? 1 : 0;
            }
        }
        // ---- in_positives (PYX:878): the first violator that is not one of the user's positives
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (__ballot(act[q]) == 0ull) continue;
            int used = NBF;
            while (true) {
                const bool part = act[q] && chosen[q] < 0 && vm[q] != 0ull;
                if (__ballot(part) == 0ull) break;
                const int r = part ? (__ffsll((long long)vm[q]) - 1) : 0;
                if (part) vm[q] &= vm[q] - 1ull;
                const int cand = __shfl(myitem[q], gbase + r, WAVE);
                const bool ask = part && __shfl(maybe_pos[q], gbase + r, WAVE) != 0;
                bool found = false;
                if (__ballot(ask) != 0ull) found = group_in_positives<LPR>(indices, cand, c_lo[q], c_hi[q], ask, gbase, p);
                c3 += (uint32_t)__popcll(__ballot(part && p == 0));  // PYX:878-879: the draw still counts
                if (part && !found) {
                    chosen[q] = cand;
                    used = r;
                }
            }
            if (act[q]) sampled[q] = used;
            c0 += (uint32_t)__popcll(__ballot(act[q] && p == 0));
            c2 += (uint32_t)__popcll(__ballot(act[q] && chosen[q] >= 0 && p == 0));
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) c1 += (uint32_t)__builtin_amdgcn_readlane(sampled[q], gg * LPR);
        }

        // ONE wait for everything requested so far: the probes and the accumulator cells of this pass and, older than those,
        // the rows of the next pass (which therefore never waits for the atomics issued below)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        double lossd[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            lossd[q] = 0.0;
            if (act[q] && chosen[q] >= 0) {
                lossd[q] = (double)__int_as_float(cur[q].w) * a.logtab[sampled[q]];  // PYX:881-885, log from host libm
                if (lossd[q] > MAX_LOSS) lossd[q] = MAX_LOSS;
            }
            const int64_t i = ib + 2 * g + q;
            if (in[q] && p == 0) {
                if (a.neg_log) a.neg_log[i] = chosen[q];
                if (a.sampled_log) a.sampled_log[i] = sampled[q];
            }
        }

        // ---- updates: float64 cell arithmetic (PYX:416-449), the four groups at once, one interaction after the other
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const bool upd = act[q] && chosen[q] >= 0;
            if (__ballot(upd) == 0ull) continue;
            {   // the first violator was a positive and a later one is the choice: its cells are re-read (weights as of now)
                const bool re = upd && chosen[q] != spec_cand[q];
                if (__ballot(re) != 0ull) {
                    if (re && cell_lane) {
                        gN[q] = (Gi + (size_t)chosen[q] * RS)[cc];
                        cN[q] = (WiW + (size_t)chosen[q] * RS)[cc];
                    }
                    if constexpr (BIN) {
                    } else if (paired) {
                        if (re && (p == 1 || p == 4)) obW[q] = *pair_ptr(q, chosen[q]);
                    } else if (re && p == 1) {
                        obW[q] = a.m.b[0][chosen[q]];
                        obG[q] = a.m.bG[0][chosen[q]];
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                }
            }
            const double loss = lossd[q];
            const float Ur = cU[q], Pr = cP[q], Nr = cN[q];
            // (the bias lane of BIN: gradients -loss / loss / loss, PYX:571-599 -- the factor is exactly 1)
            const double u = bias_lane ? 1.0 : (double)Ur;
            const double df = bias_lane ? 1.0 : (double)__fsub_rn(Nr, Pr);  // float32 subtraction, PYX:634-635
            float nWP, nGP, nWN, nGN, nWU, nGU, nM;
            double lr;
            cell_math(Pr, gP[q], 0.0f, 1.0, -loss * u, h, 0.0, nWP, nGP, nM, lr);
            cell_math(Nr, gN[q], 0.0f, 1.0, loss * u, h, 0.0, nWN, nGN, nM, lr);
            cell_math(Ur, gU[q], 0.0f, 1.0, loss * df, h, 0.0, nWU, nGU, nM, lr);
            asm volatile("" : "+v"(nWP), "+v"(nGP), "+v"(nWN), "+v"(nGN), "+v"(nWU), "+v"(nGU));
            if constexpr (RP) {
                // one instruction per row pair: in `half` 0 the even groups publish their own W deltas and the odd groups the even
                // groups' G deltas (into the same lines); in half 1 the roles swap
                const bool mine = upd && cell_lane;
                auto publish_rows = [&](float *tab, uint32_t off, float dW, float dG) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const bool own = (g & 1) == half;             // this lane publishes its own W delta in this half
                        const int from = own ? lane : (half == 0 ? lane - LPR : lane + LPR);
                        const uint32_t off_o = (uint32_t)__shfl((int)off, from, WAVE);
                        const float dG_o = __shfl(dG, from, WAVE);
                        const bool on_o = __shfl((int)mine, from, WAVE) != 0;
                        if (own ? mine : on_o) atomicAdd(tab + (own ? off : off_o + 16u) + cc, own ? dW : dG_o);
                    }
                };
                publish_rows(WiW, (uint32_t)cur[q].y * 32u, __fsub_rn(nWP, Pr), __fsub_rn(nGP, gP[q]));
                publish_rows(WiW, (uint32_t)chosen[q] * 32u, __fsub_rn(nWN, Nr), __fsub_rn(nGN, gN[q]));
                if constexpr (USTORE) {
                    if (mine) {
                        const size_t oU = (size_t)cur[q].x * 32 + cc;
                        WuW[oU] = nWU;
                        WuW[oU + 16] = nGU;
                    }
                } else {
                    publish_rows(WuW, (uint32_t)cur[q].x * 32u, __fsub_rn(nWU, Ur), __fsub_rn(nGU, gU[q]));
                }
            } else
            if (upd && p < d) {
                const size_t oP = (size_t)cur[q].y * d + cc, oN = (size_t)chosen[q] * d + cc, oU = (size_t)cur[q].x * d + cc;
                atomicAdd(WiW + oP, __fsub_rn(nWP, Pr));
                atomicAdd(Gi + oP, __fsub_rn(nGP, gP[q]));
                atomicAdd(WiW + oN, __fsub_rn(nWN, Nr));
                atomicAdd(Gi + oN, __fsub_rn(nGN, gN[q]));
                if constexpr (USTORE) {
                    WuW[oU] = nWU;
                    Gu[oU] = nGU;
                } else {
                    atomicAdd(WuW + oU, __fsub_rn(nWU, Ur));
                    atomicAdd(Gu + oU, __fsub_rn(nGU, gU[q]));
                }
            }
            // the bias cells (PYX:571-599): lane 16 g + 0 = positive item, + 1 = negative item, + 2 = user
            if constexpr (!BIN) {
                float bnW, bnG, bnM;
                double blr;
                const float cell_own = obW[q];
                if (paired) obG[q] = __shfl(obW[q], lane + 3, WAVE);
                cell_math(obW[q], obG[q], 0.0f, 1.0, p == 0 ? -loss : loss, h, 0.0, bnW, bnG, bnM, blr);
                if (paired) {
                    const float nG_up = __shfl(bnG, lane - 3, WAVE);
                    const float nv = p < 3 ? bnW : nG_up;
                    if (upd && p < 6) {
                        float *cp = pair_ptr(q, chosen[q]);
                        if (USTORE && (p == 2 || p == 5)) *cp = nv;
                        else atomicAdd(cp, __fsub_rn(nv, cell_own));
                    }
                } else
                if (upd && p < 3) {
                    float *bWp, *bGp;
                    bias_ptrs(q, chosen[q], bWp, bGp);
                    if (USTORE && p == 2) {
                        *bWp = bnW;
                        *bGp = bnG;
                    } else {
                        atomicAdd(bWp, __fsub_rn(bnW, obW[q]));
                        atomicAdd(bGp, __fsub_rn(bnG, obG[q]));
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            cur[q] = nxt[q];
            c_lo[q] = n_lo[q];
            c_hi[q] = n_hi[q];
            nxt[q] = rec2[q];
            row2[q] = row3[q];
            myitem[q] = myitem_n[q];
            s[q] = s_n[q];
        }
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
constexpr size_t tile_narrow_smem() { return (size_t)WAVES_PER_BLOCK * (2 * 3 * 256 + 2 * WAVE) * sizeof(float); }

}  // namespace lfm
