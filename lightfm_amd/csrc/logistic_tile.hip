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

// 0 outside the kernel's scope, else its LDS bytes per 256-thread workgroup (LIGHTFM_AMD_LOGISTIC_TILE=0 keeps the row-stream kernel)
size_t logistic_tile_smem(int d, int64_t n_users, int64_t n_items)
{
    const char *e = getenv("LIGHTFM_AMD_LOGISTIC_TILE");  // (read per call: the tests compare the two kernels in one process)
    const bool on = !e || atoi(e) != 0;
    if (!on || d < 4 || d > 12 || (d & 3) != 0 || std::max(n_users, n_items) * 32 >= (1ll << 30)) return 0;
    return (size_t)WAVES_PER_BLOCK * (16 * 32 + 16) * sizeof(float);
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
