// logistic_tile.hip -- fit_logistic (PYX:694-781) for NARROW identity models: the reference's literal default,
// LightFM() = logistic loss, no_components = 10.  PYX = _lightfm_fast.pyx.template.
//
// The row-stream kernels (feat_kernel.hpp) give this model a whole wavefront per interaction: 0.35 G interactions/s at the
// ML-20M shape, an eighth of what WARP runs at on the same width although logistic does less per interaction (two rows, no
// sampling, no positives lookup).  Here the model's tables are the one-line-per-feature rows of the narrow WARP kernel
// (session.hip: row pairs with biases, [W(d) | b | .. || G(d) | bG | ..], 128 bytes, d <= 12), and
//
//   gather   a 16-lane group works on TWO interactions per pass (eight per wavefront pass); lanes 0-7 of a group hold the
//            eight 16-byte pieces of the user's line, lanes 8-15 those of the item's: ONE global_load_dwordx4 per
//            interaction brings W, G, b and bG of both rows; the next pass's loads are issued before this pass computes;
//   score    the products of piece j sit in lanes j and 8 + j after one row_ror:8 exchange; the reference's sequential sum
//            (PYX:320-334: biases first, then component by component) walks lanes 0 .. d/4 - 1 through row_shr:1;
//   update   lane j of a row's W half evaluates the reference's float64 cell (PYX:416-449; cell_math_adagrad: bit-identical)
//            for its four coordinates -- gradient loss * x_c with x the other row's embedding (PYX:454-535), slot d is the
//            bias cell with x = 1 -- and hands the accumulator deltas to the lane that holds the G piece;
//   publish  the sixteen lines of a pass are transposed through 2 KB of LDS so that ONE instruction publishes a whole line
//            (32 lanes x one float, two lines per instruction): two line operations per interaction -- the write-side line
//            operations are what bounds these kernels (DESIGN.md "Tile kernel, round 6").
// Semantics: parallel (Hogwild) mode as every tile kernel -- weights as of the gather, float-atomic publication of
// new - old; one interaction per launch reproduces the sequential oracle (tests/test_hip_logistic_tile.py).
// Scope: identity features on both sides, adagrad, no L2 penalty, 4 <= d <= 12 (device width, a multiple of 4).
#include <stdlib.h>

#include <algorithm>

#include "warp_tile_kernel.hpp"

namespace lfm {

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int ROW_SHL4 = 0x104, ROW_SHR1 = 0x111, ROW_SHR4 = 0x114, ROW_ROR8 = 0x128;

}  // namespace

__global__ __launch_bounds__(256) void fit_logistic_tile_kernel(FitArgs a)
{
    constexpr int Q = 2, LPR = 16;
    constexpr int LINES = 16;  // lines of a pass: 8 interactions x (user, item)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), wib = uni((int)(threadIdx.x >> 6));
    const int g = lane / LPR, t = lane % LPR, gbase = g * LPR;
    const int side = t >> 3, piece = t & 7;  // side 0: the user's line, 1: the item's; pieces 0-3 = W half, 4-7 = G half
    const int d = a.m.d, nq = d >> 2;        // nq W pieces hold embedding cells; piece nq, cell 0 = the bias
    float *tr = smem + (size_t)wib * (LINES * 32 + LINES);  // [line][32] deltas, then the lines' rows
    int *tr_row = (int *)(tr + LINES * 32);
    float *tabU = a.rp[1], *tabI = a.rp[0];
    const float lr = a.m.lr;

    uint32_t c0 = 0, c2 = 0;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6) * 4 * Q;
    int64_t ib = a.begin + gw * 4 * Q;

    auto fetch = [&](const int4 &r, bool ok) -> float4 {
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (ok) v = *(const float4 *)((side ? tabI + (size_t)r.y * 32 : tabU + (size_t)r.x * 32) + 4 * piece);
        return v;
    };

    int4 cur[Q], nxt[Q];
    int row2[Q];
    float4 Lc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t i = ib + 2 * g + q;
        cur[q] = nxt[q] = make_int4(0, 0, 0, 0);
        row2[q] = 0;
        if (i < a.end) cur[q] = a.recs[guard_row(a, a.shuffle[i])];
        if (i + stride < a.end) nxt[q] = a.recs[guard_row(a, a.shuffle[i + stride])];
        if (i + 2 * stride < a.end) row2[q] = a.shuffle[i + 2 * stride];
        Lc[q] = fetch(cur[q], i < a.end);
    }

    for (; ib < a.end; ib += stride) {
        // ---- the next pass's lines and the records after it
        float4 Ln[Q];
        int4 rec2[Q];
        int row3[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t i = ib + 2 * g + q;
            Ln[q] = fetch(nxt[q], i + stride < a.end);
            rec2[q] = make_int4(0, 0, 0, 0);
            row3[q] = 0;
            if (i + 2 * stride < a.end) rec2[q] = a.recs[guard_row(a, row2[q])];
            if (i + 3 * stride < a.end) row3[q] = a.shuffle[i + 3 * stride];
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t i = ib + 2 * g + q;
            const bool in = i < a.end;
            const float4 v = Lc[q];
            // the other row's piece of the same index (lanes j <-> 8 + j)
            const float4 o = make_float4(dpp<ROW_ROR8>(v.x), dpp<ROW_ROR8>(v.y), dpp<ROW_ROR8>(v.z), dpp<ROW_ROR8>(v.w));
            // ---- prediction (PYX:320-334): user_repr[d] + item_repr[d], then += user_repr[c] * item_repr[c], c = 0 .. d - 1
            const float bsum = side ? __fadd_rn(o.x, v.x) : __fadd_rn(v.x, o.x);  // lane nq: the two bias cells
            float s = __shfl(bsum, gbase + nq, WAVE);
            const float p0 = __fmul_rn(v.x, o.x), p1 = __fmul_rn(v.y, o.y), p2 = __fmul_rn(v.z, o.z), p3 = __fmul_rn(v.w, o.w);
            float run = 0.0f;
            for (int j = 0; j < nq; ++j) {  // lane j continues the sum lane j - 1 hands over
                run = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(s, p0), p1), p2), p3);
                s = dpp<ROW_SHR1>(run);
            }
            const float score = __shfl(run, gbase + nq - 1, WAVE);
            const double prediction = (double)sigmoidf_ref(score);                 // PYX:745-747
            const int yb = (__int_as_float(cur[q].z) <= 0.0f) ? 0 : 1;           // PYX:751-755
            const double loss = (double)__int_as_float(cur[q].w) * (prediction - (double)yb);
            c0 += (uint32_t)__popcll(__ballot(in && yb && t == 0));
            c2 += (uint32_t)__popcll(__ballot(in && t == 0));
            // ---- update (PYX:454-535): W-half lanes run the cells of their piece; G comes from the lane four to the right
            const float4 G = make_float4(dpp<ROW_SHL4>(v.x), dpp<ROW_SHL4>(v.y), dpp<ROW_SHL4>(v.z), dpp<ROW_SHL4>(v.w));
            const bool wlane = piece <= nq && piece < 4;   // (d = 12: piece 3 holds the bias cell)
            const bool bias_piece = piece == nq;
            float4 dW = make_float4(0.0f, 0.0f, 0.0f, 0.0f), dG = dW;
            if (in && wlane) {
                float nW, nG;
                // x: the other row's embedding cell; the bias cell's gradient is the loss itself (x = 1, exactly)
                cell_math_adagrad(v.x, G.x, 1.0, loss * (bias_piece ? 1.0 : (double)o.x), lr, nW, nG);
                dW.x = __fsub_rn(nW, v.x);
                dG.x = __fsub_rn(nG, G.x);
                if (!bias_piece) {
                    cell_math_adagrad(v.y, G.y, 1.0, loss * (double)o.y, lr, nW, nG);
                    dW.y = __fsub_rn(nW, v.y);
                    dG.y = __fsub_rn(nG, G.y);
                    cell_math_adagrad(v.z, G.z, 1.0, loss * (double)o.z, lr, nW, nG);
                    dW.z = __fsub_rn(nW, v.z);
                    dG.z = __fsub_rn(nG, G.z);
                    cell_math_adagrad(v.w, G.w, 1.0, loss * (double)o.w, lr, nW, nG);
                    dW.w = __fsub_rn(nW, v.w);
                    dG.w = __fsub_rn(nG, G.w);
                }
            }
            // the G-half lanes take the accumulator deltas from four lanes to the left
            const float4 gG = make_float4(dpp<ROW_SHR4>(dG.x), dpp<ROW_SHR4>(dG.y), dpp<ROW_SHR4>(dG.z), dpp<ROW_SHR4>(dG.w));
            const float4 mine = piece < 4 ? dW : gG;
            const int line = 2 * (2 * g + q) + side;
            *(float4 *)(tr + line * 32 + 4 * piece) = mine;
            if (piece == 0) tr_row[line] = in ? (side ? cur[q].y : cur[q].x) : -1;
        }
        wave_sync();
        // ---- publication: one instruction per pair of lines (lanes 0-31 a user's line, lanes 32-63 an item's)
        {
            float *tab = (lane >> 5) ? tabI : tabU;
            const int f = lane & 31;
#pragma unroll
            for (int k = 0; k < LINES / 2; ++k) {
                const int line = 2 * k + (lane >> 5);
                const float dl = tr[line * 32 + f];
                const int row = tr_row[line];
                if (row >= 0 && dl != 0.0f) atomicAdd(tab + (size_t)row * 32 + f, dl);
            }
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            cur[q] = nxt[q];
            nxt[q] = rec2[q];
            row2[q] = row3[q];
            Lc[q] = Ln[q];
        }
    }

    if (lane == 0) {
        if (c0) atomicAdd(a.counters + 0, (unsigned long long)c0);
        if (c2) atomicAdd(a.counters + 2, (unsigned long long)c2);
    }
}

// fit_bpr (PYX:1074-1182) for the same models: three lines per interaction.  A 16-lane group again works on two interactions per
// pass with TWO loads each: (user | positive item) and (candidate 0 | candidate 1) -- the first two draws of the position's
// stream, negative_item_id = item_ids[rand % no_examples] (PYX:1124-1125: negatives are drawn from the interaction list), both
// requested a pass ahead.  The negative is the first candidate that is not one of the user's positives (PYX:1126-1127;
// in_positives: the 16-ary search of the WARP tile kernels); when both are (rare) the draws continue one at a time and the
// chosen line is fetched then.  Scores: two sequential sums side by side (user . positive, user . negative); loss = weight
// (1 - sigmoid(pp - np)) (PYX:1158); warp_update (PYX:537-649): the user row moves along neg - pos (a float32 difference), the
// positive against and the negative along the user row, the three bias cells by -/+ the loss; 24 lines per pass go through
// the LDS transposition, one instruction per pair of lines: three line operations per interaction.
template <int Q>
__global__ __launch_bounds__(256) void fit_bpr_tile_kernel(FitArgs a)
{
    constexpr int LPR = 16;
    constexpr int LINES = 12 * Q;  // lines of a pass: 4 Q interactions x (user, positive, negative)
    constexpr int ITEM = 1 << 30;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), wib = uni((int)(threadIdx.x >> 6));
    const int g = lane / LPR, t = lane % LPR, gbase = g * LPR;
    const int side = t >> 3, piece = t & 7;
    const int d = a.m.d, nq = d >> 2;
    float *tr = smem + (size_t)wib * (LINES * 32 + 32);
    int *tr_row = (int *)(tr + LINES * 32);
    float *tabU = a.rp[1], *tabI = a.rp[0];
    const float lr = a.m.lr;
    const int32_t *indptr = a.pos.indptr, *indices = a.pos.indices;
    const uint32_t n_examples = (uint32_t)a.n, base_seed = a.seeds[0];

    uint32_t c0 = 0, c1 = 0;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6) * 4 * Q;
    int64_t ib = a.begin + gw * 4 * Q;
    auto pos_of = [&](int64_t b) -> int64_t { return b + Q * g; };  // first position of this lane's group in the pass at b

    auto fetch1 = [&](const int4 &r, bool ok) -> float4 {
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (ok) v = *(const float4 *)((side ? tabI + (size_t)r.y * 32 : tabU + (size_t)r.x * 32) + 4 * piece);
        return v;
    };
    auto fetch2 = [&](int cand, bool ok) -> float4 {
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (ok) v = *(const float4 *)(tabI + (size_t)cand * 32 + 4 * piece);
        return v;
    };
    // this lane's candidate of position i: draw 1 (lanes 0-7 of a group) or draw 2 (lanes 8-15) of the position's stream
    auto cand_of = [&](int64_t i, bool ok) -> int {
        if (!ok) return 0;
        uint32_t s = lcg(position_seed(base_seed, (uint64_t)i));
        if (side) s = lcg(s);
        return a.item_ids[draw(s) % n_examples];  // PYX:1124-1125
    };

    int4 cur[Q], nxt[Q];
    int row2[Q], c_lo[Q], c_hi[Q], candc[Q], candn[Q];
    float4 L1c[Q], L2c[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t i = pos_of(ib) + q;
        cur[q] = nxt[q] = make_int4(0, 0, 0, 0);
        row2[q] = c_lo[q] = c_hi[q] = 0;
        if (i < a.end) {
            cur[q] = a.recs[guard_row(a, a.shuffle[i])];
            c_lo[q] = indptr[cur[q].x];
            c_hi[q] = indptr[cur[q].x + 1];
        }
        if (i + stride < a.end) nxt[q] = a.recs[guard_row(a, a.shuffle[i + stride])];
        if (i + 2 * stride < a.end) row2[q] = a.shuffle[i + 2 * stride];
        candc[q] = cand_of(i, i < a.end);
        candn[q] = cand_of(i + stride, i + stride < a.end);
        L1c[q] = fetch1(cur[q], i < a.end);
        L2c[q] = fetch2(candc[q], i < a.end);
    }

    for (; ib < a.end; ib += stride) {
        float4 L1n[Q], L2n[Q];
        int4 rec2[Q];
        int row3[Q], n_lo[Q], n_hi[Q], cand2[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t i = pos_of(ib) + q;
            const bool nok = i + stride < a.end;
            L1n[q] = fetch1(nxt[q], nok);
            L2n[q] = fetch2(candn[q], nok);
            rec2[q] = make_int4(0, 0, 0, 0);
            row3[q] = n_lo[q] = n_hi[q] = 0;
            if (nok) {
                n_lo[q] = indptr[nxt[q].x];
                n_hi[q] = indptr[nxt[q].x + 1];
            }
            if (i + 2 * stride < a.end) rec2[q] = a.recs[guard_row(a, row2[q])];
            if (i + 3 * stride < a.end) row3[q] = a.shuffle[i + 3 * stride];
            cand2[q] = cand_of(i + 2 * stride, i + 2 * stride < a.end);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t i = pos_of(ib) + q;
            const bool in = i < a.end;
            const bool act = in && (__int_as_float(cur[q].z) > 0.0f);  // PYX:1116-1117, before any RNG use
            const float4 v1 = L1c[q];
            const float4 o1 = make_float4(dpp<ROW_ROR8>(v1.x), dpp<ROW_ROR8>(v1.y), dpp<ROW_ROR8>(v1.z), dpp<ROW_ROR8>(v1.w));
            float4 v2 = L2c[q];
            // ---- the negative: the first candidate that is not one of the user's positives (at most no_examples draws, PYX:1123)
            const int cand0 = __shfl(candc[q], gbase, WAVE), cand1 = __shfl(candc[q], gbase + 8, WAVE);
            const int lo = c_lo[q], hi = c_hi[q];
            int chosen = cand0, draws = act ? 1 : 0;
            bool from_b = false, reload = false;
            if (__ballot(act) != 0ull) {
                const bool found0 = group_in_positives<LPR>(indices, cand0, lo, hi, act, gbase, t);
                const bool need1 = act && found0 && a.n > 1;
                if (__ballot(need1) != 0ull) {
                    const bool found1 = group_in_positives<LPR>(indices, cand1, lo, hi, need1, gbase, t);
                    bool more = need1 && found1 && a.n > 2;
                    if (need1) {
                        chosen = cand1;
                        draws = 2;
                        from_b = true;
                    }
                    if (__ballot(more) != 0ull) {  // both were positives: one draw at a time from here
                        uint32_t s = lcg(lcg(position_seed(base_seed, (uint64_t)i)));
                        while (__ballot(more) != 0ull) {
                            s = lcg(s);
                            const int cnd = more ? a.item_ids[draw(s) % n_examples] : 0;
                            const bool f = group_in_positives<LPR>(indices, cnd, lo, hi, more, gbase, t);
                            if (more) {
                                ++draws;
                                if (!f || (int64_t)draws >= a.n) {
                                    chosen = cnd;
                                    more = false;
                                    from_b = false;
                                    reload = true;
                                }
                            }
                        }
                    }
                }
            }
            // the negative's line into lanes 0-7 of the group
            {
                const float4 sw = make_float4(dpp<ROW_ROR8>(v2.x), dpp<ROW_ROR8>(v2.y), dpp<ROW_ROR8>(v2.z), dpp<ROW_ROR8>(v2.w));
                if (from_b) v2 = sw;
                if (__ballot(reload) != 0ull) {
                    if (reload) v2 = *(const float4 *)(tabI + (size_t)chosen * 32 + 4 * piece);
                }
            }
            if (in && t == 0) {
                if (a.neg_log) a.neg_log[i] = act ? chosen : -1;
                if (a.sampled_log) a.sampled_log[i] = draws;
            }
            c0 += (uint32_t)__popcll(__ballot(act && t == 0));
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) c1 += (uint32_t)__builtin_amdgcn_readlane(draws, gg * LPR);
            // ---- the two predictions (PYX:320-334), side by side: lanes 0 .. nq - 1 of the group
            const float bp = __fadd_rn(v1.x, o1.x), bn = __fadd_rn(v1.x, v2.x);  // lane nq: user bias + item bias
            float sp = __shfl(bp, gbase + nq, WAVE), sn = __shfl(bn, gbase + nq, WAVE);
            const float pp0 = __fmul_rn(v1.x, o1.x), pp1 = __fmul_rn(v1.y, o1.y), pp2 = __fmul_rn(v1.z, o1.z), pp3 = __fmul_rn(v1.w, o1.w);
            const float pn0 = __fmul_rn(v1.x, v2.x), pn1 = __fmul_rn(v1.y, v2.y), pn2 = __fmul_rn(v1.z, v2.z), pn3 = __fmul_rn(v1.w, v2.w);
            float runp = 0.0f, runn = 0.0f;
            for (int j = 0; j < nq; ++j) {
                runp = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(sp, pp0), pp1), pp2), pp3);
                runn = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(sn, pn0), pn1), pn2), pn3);
                sp = dpp<ROW_SHR1>(runp);
                sn = dpp<ROW_SHR1>(runn);
            }
            const double pp = (double)__shfl(runp, gbase + nq - 1, WAVE), np_ = (double)__shfl(runn, gbase + nq - 1, WAVE);
            // PYX:1158: weight * (1 - sigmoid(pp - np)); the difference is narrowed to float32
            const double loss = (double)__int_as_float(cur[q].w) * (1.0 - (double)sigmoidf_ref((float)(pp - np_)));
            // ---- warp_update (PYX:537-649)
            const float4 G1 = make_float4(dpp<ROW_SHL4>(v1.x), dpp<ROW_SHL4>(v1.y), dpp<ROW_SHL4>(v1.z), dpp<ROW_SHL4>(v1.w));
            const float4 G2 = make_float4(dpp<ROW_SHL4>(v2.x), dpp<ROW_SHL4>(v2.y), dpp<ROW_SHL4>(v2.z), dpp<ROW_SHL4>(v2.w));
            const bool wlane = piece <= nq && piece < 4;
            const bool bias_piece = piece == nq;
            float4 dW1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), dG1 = dW1, dW2 = dW1, dG2 = dW1;
            if (act && wlane) {
                float nW, nG;
                // lanes 0-7: the user's row, x = neg - pos, +loss; lanes 8-15: the positive's row, x = user, -loss
                const double l1 = side ? -loss : loss;
                const float x0 = side ? o1.x : __fsub_rn(v2.x, o1.x), x1 = side ? o1.y : __fsub_rn(v2.y, o1.y);
                const float x2 = side ? o1.z : __fsub_rn(v2.z, o1.z), x3 = side ? o1.w : __fsub_rn(v2.w, o1.w);
                cell_math_adagrad(v1.x, G1.x, 1.0, l1 * (bias_piece ? 1.0 : (double)x0), lr, nW, nG);
                dW1.x = __fsub_rn(nW, v1.x);
                dG1.x = __fsub_rn(nG, G1.x);
                if (!bias_piece) {
                    cell_math_adagrad(v1.y, G1.y, 1.0, l1 * (double)x1, lr, nW, nG);
                    dW1.y = __fsub_rn(nW, v1.y);
                    dG1.y = __fsub_rn(nG, G1.y);
                    cell_math_adagrad(v1.z, G1.z, 1.0, l1 * (double)x2, lr, nW, nG);
                    dW1.z = __fsub_rn(nW, v1.z);
                    dG1.z = __fsub_rn(nG, G1.z);
                    cell_math_adagrad(v1.w, G1.w, 1.0, l1 * (double)x3, lr, nW, nG);
                    dW1.w = __fsub_rn(nW, v1.w);
                    dG1.w = __fsub_rn(nG, G1.w);
                }
                if (side == 0) {  // the negative's row: x = user, +loss
                    cell_math_adagrad(v2.x, G2.x, 1.0, loss * (bias_piece ? 1.0 : (double)v1.x), lr, nW, nG);
                    dW2.x = __fsub_rn(nW, v2.x);
                    dG2.x = __fsub_rn(nG, G2.x);
                    if (!bias_piece) {
                        cell_math_adagrad(v2.y, G2.y, 1.0, loss * (double)v1.y, lr, nW, nG);
                        dW2.y = __fsub_rn(nW, v2.y);
                        dG2.y = __fsub_rn(nG, G2.y);
                        cell_math_adagrad(v2.z, G2.z, 1.0, loss * (double)v1.z, lr, nW, nG);
                        dW2.z = __fsub_rn(nW, v2.z);
                        dG2.z = __fsub_rn(nG, G2.z);
                        cell_math_adagrad(v2.w, G2.w, 1.0, loss * (double)v1.w, lr, nW, nG);
                        dW2.w = __fsub_rn(nW, v2.w);
                        dG2.w = __fsub_rn(nG, G2.w);
                    }
                }
            }
            // every draw was one of the user's positives and the last one the positive itself (a user with the whole catalogue):
            // the reference updates that row twice in sequence -- the negative's cells start from what the positive's leave
            const bool same = act && chosen == cur[q].y;
            if (__ballot(same) != 0ull) {
                const float4 pW = make_float4(dpp<ROW_ROR8>(dW1.x), dpp<ROW_ROR8>(dW1.y), dpp<ROW_ROR8>(dW1.z), dpp<ROW_ROR8>(dW1.w));
                const float4 pG = make_float4(dpp<ROW_ROR8>(dG1.x), dpp<ROW_ROR8>(dG1.y), dpp<ROW_ROR8>(dG1.z), dpp<ROW_ROR8>(dG1.w));
                auto again = [&](float w, float gg, float pw, float pg, double grad, float &dw, float &dg) {
                    const float ow = __fadd_rn(w, pw), og = __fadd_rn(gg, pg);
                    float nW, nG;
                    cell_math_adagrad(ow, og, 1.0, grad, lr, nW, nG);
                    dw = __fsub_rn(nW, ow);
                    dg = __fsub_rn(nG, og);
                };
                if (same && wlane && side == 0) {
                    again(v2.x, G2.x, pW.x, pG.x, loss * (bias_piece ? 1.0 : (double)v1.x), dW2.x, dG2.x);
                    if (!bias_piece) {
                        again(v2.y, G2.y, pW.y, pG.y, loss * (double)v1.y, dW2.y, dG2.y);
                        again(v2.z, G2.z, pW.z, pG.z, loss * (double)v1.z, dW2.z, dG2.z);
                        again(v2.w, G2.w, pW.w, pG.w, loss * (double)v1.w, dW2.w, dG2.w);
                    }
                }
            }
            // (component by component: a select between two float4 values is compiled into an indexed array in scratch)
            // (the lane exchanges run with every lane enabled, before the selects)
            const bool wh = piece < 4;
            float h1x = dpp<ROW_SHR4>(dG1.x), h1y = dpp<ROW_SHR4>(dG1.y), h1z = dpp<ROW_SHR4>(dG1.z), h1w = dpp<ROW_SHR4>(dG1.w);
            float h2x = dpp<ROW_SHR4>(dG2.x), h2y = dpp<ROW_SHR4>(dG2.y), h2z = dpp<ROW_SHR4>(dG2.z), h2w = dpp<ROW_SHR4>(dG2.w);
            asm volatile("" : "+v"(h1x), "+v"(h1y), "+v"(h1z), "+v"(h1w), "+v"(h2x), "+v"(h2y), "+v"(h2z), "+v"(h2w));
            const float m1x = wh ? dW1.x : h1x, m1y = wh ? dW1.y : h1y, m1z = wh ? dW1.z : h1z, m1w = wh ? dW1.w : h1w;
            const float m2x = wh ? dW2.x : h2x, m2y = wh ? dW2.y : h2y, m2z = wh ? dW2.z : h2z, m2w = wh ? dW2.w : h2w;
            const int base = 3 * (Q * g + q);
            *(float4 *)(tr + (base + side) * 32 + 4 * piece) = make_float4(m1x, m1y, m1z, m1w);
            if (side == 0) *(float4 *)(tr + (base + 2) * 32 + 4 * piece) = make_float4(m2x, m2y, m2z, m2w);
            if (piece == 0) {
                tr_row[base + side] = act ? (side ? (cur[q].y | ITEM) : cur[q].x) : -1;
                if (side == 0) tr_row[base + 2] = act ? (chosen | ITEM) : -1;
            }
        }
        wave_sync();
        {
            const int f = lane & 31;
#pragma unroll
            for (int k = 0; k < LINES / 2; ++k) {
                const int line = 2 * k + (lane >> 5);
                const float dl = tr[line * 32 + f];
                const int row = tr_row[line];
                if (row >= 0 && dl != 0.0f) atomicAdd(((row & ITEM) ? tabI : tabU) + (size_t)(row & (ITEM - 1)) * 32 + f, dl);
            }
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            cur[q] = nxt[q];
            nxt[q] = rec2[q];
            row2[q] = row3[q];
            c_lo[q] = n_lo[q];
            c_hi[q] = n_hi[q];
            L1c[q] = L1n[q];
            L2c[q] = L2n[q];
            candc[q] = candn[q];
            candn[q] = cand2[q];
        }
    }

    if (lane == 0) {
        if (c0) atomicAdd(a.counters + 0, (unsigned long long)c0);
        if (c1) atomicAdd(a.counters + 1, (unsigned long long)c1);
        if (c0) atomicAdd(a.counters + 2, (unsigned long long)c0);  // (every visited positive is an update)
        if (c1) atomicAdd(a.counters + 3, (unsigned long long)c1);  // (every draw is one in_positives probe)
    }
}

// 0 outside the kernel's scope, else its LDS bytes per 256-thread workgroup (LIGHTFM_AMD_LOGISTIC_TILE=0 keeps the row-stream kernel)
size_t logistic_tile_smem(int d, int64_t n_users, int64_t n_items)
{
    const char *e = getenv("LIGHTFM_AMD_LOGISTIC_TILE");  // (read per call: the tests compare the two kernels in one process)
    const bool on = !e || atoi(e) != 0;
    if (!on || d < 4 || d > 12 || (d & 3) != 0 || std::max(n_users, n_items) * 32 >= (1ll << 30)) return 0;
    return (size_t)WAVES_PER_BLOCK * (16 * 32 + 16) * sizeof(float);
}

// the same for fit_bpr_tile_kernel (LIGHTFM_AMD_BPR_TILE=0 keeps the row-stream kernel)
#ifndef BPR_TILE_Q
#define BPR_TILE_Q 1  // interactions per lane group and pass (2: 189 VGPRs and scratch)
#endif
int bpr_tile_per_wave() { return 4 * BPR_TILE_Q; }
size_t bpr_tile_smem(int d, int64_t n_users, int64_t n_items)
{
    const char *e = getenv("LIGHTFM_AMD_BPR_TILE");
    const bool on = !e || atoi(e) != 0;
    if (!on || d < 4 || d > 12 || (d & 3) != 0 || std::max(n_users, n_items) * 32 >= (1ll << 30)) return 0;
    return (size_t)WAVES_PER_BLOCK * (12 * BPR_TILE_Q * 32 + 32) * sizeof(float);
}

hipError_t launch_fit_bpr_tile(const FitArgs &a, int grid, hipStream_t st, int cus, int *grid_used)
{
    const size_t smem = (size_t)WAVES_PER_BLOCK * (12 * BPR_TILE_Q * 32 + 32) * sizeof(float);
    if (cus > 0) {
        const int per_cu = occupancy_cached(fit_bpr_tile_kernel<BPR_TILE_Q>, 256, smem);
        if (per_cu > 0) grid = std::min(grid, per_cu * cus);
    }
    if (grid_used) *grid_used = grid;
    fit_bpr_tile_kernel<BPR_TILE_Q><<<grid, 256, smem, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_fit_logistic_tile(const FitArgs &a, int grid, hipStream_t st, int cus, int *grid_used)
{
    const size_t smem = (size_t)WAVES_PER_BLOCK * (16 * 32 + 16) * sizeof(float);
    if (cus > 0) {
        const int per_cu = occupancy_cached(fit_logistic_tile_kernel, 256, smem);
        if (per_cu > 0) grid = std::min(grid, per_cu * cus);
    }
    if (grid_used) *grid_used = grid;
    fit_logistic_tile_kernel<<<grid, 256, smem, st>>>(a);
    return hipGetLastError();
}

}  // namespace lfm
