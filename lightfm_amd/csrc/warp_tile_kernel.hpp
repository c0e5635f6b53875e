// warp_tile_kernel.hpp -- the production WARP epoch kernel (fit_warp, PYX:784-912) for the
// layout every BASELINE throughput configuration uses: identity user and item
// features, no L2 regularisation, parallel (Hogwild) mode.
// PYX = /root/reference/lightfm/_lightfm_fast.pyx.template
//
// Work mapping (wave64).  A wavefront is split into NG = 64/LPR lane groups of LPR lanes;
// each group owns ONE interaction per pass and a lane carries VEC floats of a row.
// Instantiated (warp_tile_lpr*.hip) for (LPR, VEC) = (16,4) NG = 4; (32,2|4) NG = 2;
// (64,1|2|4) NG = 1.  NG = 4 needs the fewest instructions per interaction (the scoring pass
// and the PRNG are shared); session.hip launches fewer interactions per wavefront when few
// may be in flight (the concurrency ramp), to have more wavefronts hiding latency.
//
//   gather   an embedding row is LPR lanes x 4*VEC bytes: ONE global_load_dwordx4 per wave
//            (NG = 4) fetches one row for each of the NG interactions (1 KiB per
//            instruction).  User row, positive row and the rows of ALL candidate negatives
//            of the batch (max_sampled of them) are requested back to back before the first
//            is consumed, then staged in a wave-private LDS tile; with NG = 1, VEC = 1 they
//            go memory -> LDS directly (global_load_lds_dword).  Biases are read from
//            a.b_read: the live tables or per-launch cached snapshots (session.hip).
//   score    lane r of a group computes the reference's SEQUENTIAL float32 dot
//            (PYX:320-334: (b_u + b_i) + u0*v0 + u1*v1 ...) of tile row r with the
//            group's user row: the positive (r = 0) and every candidate negative
//            (r = 1..nb) of all NG interactions are scored in one 16/32-step pass
//            of ds_read_b128 -- the summation order, hence every margin test and
//            every sample count, is the reference's.
//   sample   within one interaction the weights do not change between draws
//            (PYX:857-899 updates once, after the loop), so the first violator of
//            the speculatively scored batch IS the sequential loop's choice; the
//            PRNG stream of the position is advanced by exactly `sampled` draws.
//   lookup   in_positives (PYX:270-284) as an LPR-ary search run by the group.
//   update   groups with a violator are handed, one after the other, to the WHOLE
//            wave: lane c owns coordinate c (Adagrad/Adadelta are per-coordinate,
//            PYX:416-449), reads its cell of the three rows from the LDS tile, the
//            accumulators from memory, evaluates the reference's float64 cell
//            arithmetic and publishes new-old with global_atomic_add_f32.
#pragma once
#include "device.hpp"
#include "kernels.hpp"

#include <type_traits>

namespace lfm {

namespace {

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }

// The piece of a row one lane carries during the gather: VEC consecutive floats.
template <int VEC> struct Piece;
template <> struct Piece<4> { using T = float4; };
template <> struct Piece<2> { using T = float2; };
template <> struct Piece<1> { using T = float; };
template <int VEC>
__device__ __forceinline__ typename Piece<VEC>::T ldp(const float *p)
{
    return *reinterpret_cast<const typename Piece<VEC>::T *>(p);
}
template <int VEC>
__device__ __forceinline__ void stp(float *p, const typename Piece<VEC>::T &v)
{
    *reinterpret_cast<typename Piece<VEC>::T *>(p) = v;
}

// LDS-DMA: every active lane fetches 4 bytes from its own global address straight into
// lds_row[lane] (global_load_lds_dword: no VGPR, no ds_write).  `lds_row` must be wave-uniform.
typedef __attribute__((address_space(3))) float lds_float_t;
__device__ __forceinline__ void dma_lane_dword(const float *g, float *lds_row)
{
    __builtin_amdgcn_global_load_lds(g, (lds_float_t *)lds_row, 4, 0, 0);
}
// The same with 16 bytes per lane (global_load_lds_dwordx4): lane l deposits its 16 bytes at
// lds_base + 16 l -- one instruction lays 1 KiB of LDS.
__device__ __forceinline__ void dma_lane_x4(const float *g, float *lds_base)
{
    __builtin_amdgcn_global_load_lds(g, (lds_float_t *)lds_base, 16, 0, 0);
}

// x % n for x < 2^31, 2 <= n < 2^31, with magic = floor(2^32 / n) + 1:
// floor(x*magic / 2^32) is floor(x/n) or one more (x*magic/2^32 lies in (x/n, x/n + 1/2)).
__device__ __forceinline__ int fast_mod(uint32_t x, uint32_t n, uint32_t magic)
{
    uint32_t q = __umulhi(x, magic);
    int r = (int)(x - q * n);
    return r < 0 ? r + (int)n : r;
}

// Lane (k & 15) of every 16-lane DPP row, broadcast to the row (v_mov_b32 row_newbcast).
__device__ __forceinline__ int row_bcast(int v, int k)
{
    switch (k & 15) {
#define LFM_RB(K) case K: return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, false);
        LFM_RB(0) LFM_RB(1) LFM_RB(2) LFM_RB(3) LFM_RB(4) LFM_RB(5) LFM_RB(6) LFM_RB(7)
        LFM_RB(8) LFM_RB(9) LFM_RB(10) LFM_RB(11) LFM_RB(12) LFM_RB(13) LFM_RB(14) LFM_RB(15)
#undef LFM_RB
    }
    return v;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// Sequential float32 dot of PYX:320-334 over two LDS rows, biases passed in registers.
// REG (lazy L2 regularisation active): the tile holds the RAW embedding rows and every element of a
// representation is fl(w * x) with w = (float)(1.0 * scale) of its side (compute_representation over an
// identity row, PYX:306-313) -- formed here, on the fly.
template <bool REG>
__device__ __forceinline__ float row_dot(const float *u, const float *v, int d, float bu, float bi, float wu, float wi)
{
    if constexpr (REG) {
        bu = __fmul_rn(wu, bu);
        bi = __fmul_rn(wi, bi);
    }
    float acc = __fadd_rn(bu, bi);
    int c = 0;
    // 8 ds_read_b128 in flight per 16 coordinates: the LDS latency is paid once per block
    for (; c + 16 <= d; c += 16) {
        float4 a[4], x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = ld4(u + c + 4 * j);
            x[j] = ld4(v + c + 4 * j);
            if constexpr (REG) {
                a[j] = make_float4(__fmul_rn(wu, a[j].x), __fmul_rn(wu, a[j].y), __fmul_rn(wu, a[j].z), __fmul_rn(wu, a[j].w));
                x[j] = make_float4(__fmul_rn(wi, x[j].x), __fmul_rn(wi, x[j].y), __fmul_rn(wi, x[j].z), __fmul_rn(wi, x[j].w));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // the products are independent: two per v_pk_mul_f32 (each still one IEEE f32
            // multiply); the additions stay one sequential chain
            const v2f lo = v2f{a[j].x, a[j].y} * v2f{x[j].x, x[j].y};
            const v2f hi = v2f{a[j].z, a[j].w} * v2f{x[j].z, x[j].w};
            acc = __fadd_rn(acc, lo.x);
            acc = __fadd_rn(acc, lo.y);
            acc = __fadd_rn(acc, hi.x);
            acc = __fadd_rn(acc, hi.y);
        }
    }
    for (; c < d; c += 4) {
        float4 a = ld4(u + c);
        float4 x = ld4(v + c);
        if constexpr (REG) {
            a = make_float4(__fmul_rn(wu, a.x), __fmul_rn(wu, a.y), __fmul_rn(wu, a.z), __fmul_rn(wu, a.w));
            x = make_float4(__fmul_rn(wi, x.x), __fmul_rn(wi, x.y), __fmul_rn(wi, x.z), __fmul_rn(wi, x.w));
        }
        acc = __fadd_rn(acc, __fmul_rn(a.x, x.x));
        acc = __fadd_rn(acc, __fmul_rn(a.y, x.y));
        acc = __fadd_rn(acc, __fmul_rn(a.z, x.z));
        acc = __fadd_rn(acc, __fmul_rn(a.w, x.w));
    }
    return acc;
}

// in_positives (PYX:270-284) for NG lane groups at once: every participating group
// searches its own sorted row [lo, hi) for its own item with an LPR-ary search.
template <int LPR>
__device__ __forceinline__ bool group_in_positives(const int32_t *indices, int item, int lo, int hi,
                                                   bool part, int gbase, int p)
{
    constexpr unsigned long long GM = LPR == 64 ? ~0ull : ((1ull << LPR) - 1ull);
    constexpr int SH = LPR == 64 ? 6 : (LPR == 32 ? 5 : 4);
    bool dead = !part;  // group already knows the answer is "absent"
    while (true) {
        bool wide = !dead && (hi - lo > LPR);
        if (__ballot(wide) == 0ull) break;
        int step = (hi - lo + LPR - 1) >> SH;
        int idx = lo + p * step;
        bool ok = wide && idx < hi;
        int v = ok ? indices[idx] : 0x7fffffff;
        unsigned long long m = __ballot(ok && v <= item);
        int cnt = __popcll((m >> gbase) & GM);
        if (wide) {
            if (cnt == 0) dead = true;  // item below the row's first entry
            else {
                lo = lo + (cnt - 1) * step;
                hi = min(hi, lo + step);
            }
        }
    }
    int idx = lo + p;
    bool hit = !dead && idx < hi && indices[idx] == item;
    unsigned long long m = __ballot(hit);
    return ((m >> gbase) & GM) != 0ull;
}

}  // namespace

// TIMED (profiling builds of the same kernel, lfm_opts.warp_kernel = 2): every wave
// accumulates s_memtime deltas per phase of a pass into a.counters[4..11].
//
// DMA4 (LPR = 16, VEC = 4: four interactions per pass, the steady-state variant): rows go memory ->
// LDS by LDS-DMA, 16 bytes per lane, with no staging registers and no ds_write.  One instruction
// deposits the k-th row of ALL four groups as one contiguous KiB (lane (g, p) -> bytes
// 256 g + 16 p), so the tile is laid out candidate-major: row k of group g at k * (4 * 64 + 4) +
// 64 g floats.  The 4-float skew per candidate keeps the scoring pass free of bank conflicts: the
// sixteen lanes one ds_read_b128 phase serves hold candidates 0-3 of one group and 4-11 of
// another (MI355X_MICROARCH.md, LDS), i.e. twelve different bank quadruples.  The four user rows
// are fetched by one instruction per group whose LDS base is skewed by 16 bytes per group.
// Without the staging registers the kernel fits three workgroups per CU (12 wavefronts).
//
// REG: item_alpha / user_alpha != 0 (lazy L2 regularisation, PYX:640-691).  The tile keeps RAW rows; a pass
// reads the two live scales (device.hpp: RegScale) once, scales representations on the fly in the scoring
// and update phases, multiplies every updated cell by 1 + alpha * lr (cell_math), sums the cells' learning
// rates and adds log1p(alpha * avg_lr) of its interactions to the global log-scales with one float64 atomic
// per side and pass.
//
// LOSS (round 6): LFM_LOSS_WARP_ID, or LFM_LOSS_BPR_ID -- fit_bpr (PYX:1074-1182) of an identity model on the same tile: the
// candidates of a batch are item_ids[rand_r % no_examples] (PYX:1124-1125: negatives are drawn from the interaction list), the
// negative is the FIRST candidate that is not one of the user's positives whatever it scores (PYX:1126-1127; the same
// speculation: the first candidate almost always is), at most no_examples draws and the last one taken when the loop ends on
// its bound; loss = weight (1 - sigmoid(pp - np)) (PYX:1158); the update is the same warp_update (PYX:537-649).  Two
// candidates per batch (session.hip), adagrad, with or without the lazy L2 regularisation (REG).
// LFM_LOSS_LOGISTIC_ID: fit_logistic (PYX:694-781) likewise -- every record is visited (y <= 0 is the label 0, PYX:751-755), no
// candidates: the tile holds the user's and the item's row, lane 0 of a group scores the pair, loss = weight (sigmoid(score) - y)
// (PYX:745-757), and `update` (PYX:454-535) is warp_update without the negative: the item's row moves along the user's and the
// user's along the item's, both bias cells by the loss.
template <int LPR, int VEC, bool TIMED, bool ADADELTA, bool DMA4 = false, bool REG = false, int LOSS = LFM_LOSS_WARP_ID>
__global__ __launch_bounds__(256, (DMA4 && !ADADELTA && !TIMED) ? 3 : 2) void fit_warp_tile_kernel(FitArgs a)
{
    static_assert(!DMA4 || (LPR == 16 && VEC == 4), "the LDS-DMA tile layout is the four-group one");
    static_assert(LOSS == LFM_LOSS_WARP_ID || ((LOSS == LFM_LOSS_BPR_ID || LOSS == LFM_LOSS_LOGISTIC_ID) && !ADADELTA && !TIMED),
                  "BPR / logistic: adagrad, with or without the lazy L2 regularisation");
    static_assert(LOSS != LFM_LOSS_LOGISTIC_ID || VEC == 4, "logistic: the staged / LDS-DMA layouts with four floats per lane");
    constexpr bool BPR = LOSS == LFM_LOSS_BPR_ID, LGT = LOSS == LFM_LOSS_LOGISTIC_ID;
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
    constexpr int NG = WAVE / LPR;        // interactions per wave pass
    constexpr int NC = (LPR * VEC) / WAVE;  // coordinates per lane in the update phase
    constexpr unsigned long long GM = LPR == 64 ? ~0ull : ((1ull << LPR) - 1ull);
    using PieceT = typename Piece<VEC>::T;
    static_assert(LPR * VEC >= WAVE && (LPR * VEC) % WAVE == 0, "a row must cover whole wavefronts of coordinates");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), wib = uni((int)(threadIdx.x >> 6));
    // one interaction per wavefront (LPR == 64): group index and every per-group value are
    // wave-uniform -- say so, and the compiler keeps them in SGPRs / branches instead of masks
    const int g = LPR == 64 ? 0 : lane / LPR, p = LPR == 64 ? lane : lane % LPR, gbase = g * LPR;
    const int d = a.m.d, TS = a.tile_stride, RG = a.tile_rows;
    // wave-private tile: NG*RG item rows (row 0 of a group = the positive) + NG user rows.
    // KS = floats between consecutive rows of a group, GS = between the groups' first rows,
    // UB = first user row, US = between user rows.
    const int KS = DMA4 ? (NG * LPR * VEC + 4) : TS;
    const int GS = DMA4 ? (LPR * VEC) : RG * TS;
    const int UB = DMA4 ? RG * KS : NG * RG * TS;
    const int US = DMA4 ? (LPR * VEC + 4) : TS;
    float *tile = smem + (size_t)wib * (UB + NG * US);
    float *vrows = tile + (size_t)g * GS;
    float *urow = tile + UB + (size_t)g * US;
    const bool pc = VEC * p < d;  // this lane carries a piece (VEC floats) of every gathered row
    const float *Wi = a.m.W[0], *Wu = a.m.W[1];
    const float *bi_tab = a.b_read[0], *bu_tab = a.b_read[1];
    // (BPR: the draw loop is bounded by the number of interactions, PYX:1123)
    const int max_sampled = LGT ? 0 : (BPR ? (int)std::min<int64_t>(a.n, 0x7fffffff) : a.m.max_sampled);
    const uint32_t n_examples = (uint32_t)a.n;
    const uint32_t n_items = (uint32_t)a.itf.rows, magic = a.n_items_magic;
    const uint32_t base_seed = a.seeds[0];
    const Hyper h{ADADELTA ? 1 : 0, a.m.lr, a.m.rho, a.m.eps};
    const int um = a.update_mode;
    const int umU = (a.user_store && um == 0) ? 1 : um;  // the user row of an update: see FitArgs::user_store
    const uint32_t *bloom = a.bloom;              // in_positives pre-filter, nullptr = none
    // when the filter is probed: after the scoring pass, by the violating candidates only (default: one round trip
    // with the speculative accumulator loads instead of the search's two; C2 +5 %, C4 shard +7 % against the
    // plain search, A/B of one box, profiles/r04_visit_b.txt), or -- debug bit 9 (512) -- for every candidate
    // together with its embedding row (no round trip of its own but ten times the requests: +3 % / +5 %)
    const bool bloom_early = (a.debug & 512) != 0;
    // BPR / logistic (an update on EVERY interaction): the live bias cells as (b, bG) pairs of one line when the session packed them
    // (FitArgs::bb, as for the steady-state kernel) -- a row's two cells are then loaded with one 8-byte load and published by ONE
    // instruction (lanes 0-2 the W cells of the positive / negative / user, lanes 3-5 the accumulator cells next to them): three
    // line operations per update instead of six.  Scoring reads b_read with its stride (the pair table, or a launch's snapshot).
    const bool paired = (BPR || LGT) && a.bb[0] != nullptr && um == 0;
    const size_t bstr_i = (BPR || LGT) ? (size_t)a.b_read_stride[0] : 1, bstr_u = (BPR || LGT) ? (size_t)a.b_read_stride[1] : 1;

    // rand_r's LCG is affine, so k steps collapse into one multiply-add.  For the first batch of
    // every pass lane p needs the stream after min(p, nb_first) more draws: it keeps
    // (A^k, C*(A^(k-1) + ... + 1)) mod 2^32 for that k.
    const int nb_first = min(max_sampled, a.first_batch);
    uint32_t lcgA = 1u, lcgC = 0u;
    for (int j = 0; j < min(p, nb_first); ++j) {
        lcgA *= 1103515245u;
        lcgC = lcgC * 1103515245u + 12345u;
    }

    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;  // per-wave event counts: wave-uniform (ballots / lane reads), i.e. SGPRs
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    RegScale reg;  // REG: the regularisation steps this wavefront has collected (device.hpp: RegScale)
    reg.begin();
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6) * NG;
    const int32_t *indptr = a.pos.indptr, *indices = a.pos.indices;
    float *WiW = a.m.W[0], *Gi = a.m.G[0], *Mi = a.m.M[0];
    float *WuW = a.m.W[1], *Gu = a.m.G[1], *Mu = a.m.M[1];

    // Software pipeline over the interaction records, three passes deep: while pass t is
    // processed, the bounds of pass t+1's positives row, the (user, item, y, weight) record
    // of pass t+2 and the shuffle entry of pass t+3 are in flight.
    int64_t ib = a.begin + gw * NG;
    int4 cur = make_int4(0, 0, 0, 0), nxt = cur;
    int c_lo = 0, c_hi = 0, row2 = 0;
    if (ib + g < a.end) {
        cur = a.recs[guard_row(a, a.shuffle[ib + g])];
        if constexpr (!LGT) {  // (logistic has no positives lookup)
            c_lo = indptr[cur.x];
            c_hi = indptr[cur.x + 1];
        }
    }
    if (ib + stride + g < a.end) nxt = a.recs[guard_row(a, a.shuffle[ib + stride + g])];
    if (ib + 2 * stride + g < a.end) row2 = a.shuffle[ib + 2 * stride + g];
    // BPR: this lane's candidate of the first batch (it depends on the position's stream alone) is looked up a pass ahead --
    // item_ids[rand_r % no_examples] is a round trip of its own in front of the gather otherwise
    int pf_item = 0;
    if constexpr (BPR) {
        if (ib + g < a.end) pf_item = a.item_ids[draw(lcgA * position_seed(base_seed, (uint64_t)(ib + g)) + lcgC) % n_examples];
    }

    for (; ib < a.end; ib += stride) {
        const int64_t i = ib + g;
        const bool in = i < a.end;
        int pf_next = 0;
        if constexpr (BPR) {
            if (i + stride < a.end) pf_next = a.item_ids[draw(lcgA * position_seed(base_seed, (uint64_t)(i + stride)) + lcgC) % n_examples];
        }
        int n_lo = 0, n_hi = 0, row3 = 0;
        int4 rec2 = make_int4(0, 0, 0, 0);
        if constexpr (!LGT) {
            if (i + stride < a.end) {
                n_lo = indptr[nxt.x];
                n_hi = indptr[nxt.x + 1];
            }
        }
        if (i + 2 * stride < a.end) rec2 = a.recs[guard_row(a, row2)];
        if (i + 3 * stride < a.end) row3 = a.shuffle[i + 3 * stride];
        // with one interaction per wavefront the ids are wave-uniform: keep them in SGPRs so row
        // addresses are scalar arithmetic (global_load with an SGPR base)
        const int c_user = LPR == 64 ? uni(cur.x) : cur.x, c_pos = LPR == 64 ? uni(cur.y) : cur.y;
        const float c_y = __int_as_float(cur.z), c_w = __int_as_float(cur.w);
        bool act = in && (LGT || c_y > 0.0f);  // PYX:831-832, before any RNG use (logistic: every record, PYX:751-755)
        if constexpr (LPR == 64) act = __ballot(act) != 0ull;
        int sampled = 0, chosen = -1, chosen_r = 0;
        float chosen_score = 0.0f;          // BPR: the negative's prediction
        stamp(0);

        if (__ballot(act) != 0ull) {
            // REG: the live regularisation scales of this pass as (float)(1.0 * scale) (PYX:306), wave-uniform.
            // Their logarithms are requested together with the first batch of candidate rows and turned
            // into scales right after that batch's wait (a short register lifetime, no extra round trip).
            float wu = 1.0f, wi = 1.0f;
            // ---- gather: user row, positive row, user bias
            const bool gl = act && pc;
            // One interaction per wavefront with one float per lane: a row is exactly what one
            // global_load_lds_dword deposits (lane p -> row[p]), so rows go memory -> LDS directly.
            constexpr bool DMA = (LPR == 64 && VEC == 1);
            PieceT u4, p4;
            if constexpr (DMA) {
                if (gl) {
                    dma_lane_dword(Wu + (size_t)c_user * d + p, urow);
                    dma_lane_dword(Wi + (size_t)c_pos * d + p, vrows);
                }
            } else if constexpr (DMA4) {
                if (gl) {
                    // lane (gg, p) -> tile[UB + gg * US + 4 p]; a scalar loop: the LDS base of an
                    // instruction is M0, i.e. wave-uniform, and must not be merged into a per-lane select
#pragma nounroll
                    for (int gg = 0; gg < NG; ++gg) {
                        float *ub = tile + UB + __builtin_amdgcn_readfirstlane(gg) * (US - LPR * VEC);
                        if (g == gg) dma_lane_x4(Wu + (size_t)c_user * d + VEC * p, ub);
                    }
                    dma_lane_x4(Wi + (size_t)c_pos * d + VEC * p, tile);  // row 0 of every group
                }
            } else {
                // lanes without a piece read the table's first 16 bytes instead of branching
                u4 = ldp<VEC>(gl ? Wu + (size_t)c_user * d + VEC * p : Wu);
                p4 = ldp<VEC>(gl ? Wi + (size_t)c_pos * d + VEC * p : Wi);
            }
            float bu = 0.0f;
            if (act) bu = bu_tab[(size_t)c_user * bstr_u];
            uint32_t state = position_seed(base_seed, (uint64_t)i);  // stream of this position
            if constexpr (!DMA && !DMA4) {
                if (gl) {
                    stp<VEC>(urow + VEC * p, u4);
                    stp<VEC>(vrows + VEC * p, p4);
                }
            }
            // accumulator rows / bias cells of the groups that will (probably) update
            float gP[NG][NC], gN[NG][NC], gU[NG][NC], mP[NG][NC], mN[NG][NC], mU[NG][NC];
            float obW[NG], obG[NG], obM[NG];
            unsigned long long specm = 0ull;  // groups whose accumulators were requested early
            int spec_cand = -1;               // ... for this candidate negative
            auto load_rows = [&](int gg, int user, int pos, int neg, bool only_neg) {
                const size_t bp = (size_t)pos * d, bn = (size_t)neg * d, bu_ = (size_t)user * d;
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    // plain loads, lanes past d re-read coordinate 0: no exec-masked branches,
                    // so all rows of all groups are in flight together
                    const int c = lane + WAVE * q;
                    // (uniform row base + a 32-bit lane offset the compiler cannot hoist: global_load saddr form
                    // instead of one loop-invariant per-lane 64-bit pointer per table held across the pass)
                    unsigned cc = c < d ? (unsigned)c : 0u;
                    asm volatile("" : "+v"(cc));
                    gN[gg][q] = (Gi + bn)[cc];
                    if (ADADELTA) mN[gg][q] = (Mi + bn)[cc];
                    if (!only_neg) {
                        gP[gg][q] = (Gi + bp)[cc];
                        gU[gg][q] = (Gu + bu_)[cc];
                        if (ADADELTA) {
                            mP[gg][q] = (Mi + bp)[cc];
                            mU[gg][q] = (Mu + bu_)[cc];
                        }
                    }
                }
                // bias cells: lane 0 = positive item, 1 = negative item, 2.. = user (PYX:571-599)
                const int brow = lane == 0 ? pos : (lane == 1 ? neg : user);
                const float *bWp = lane >= 2 ? a.m.b[1] : a.m.b[0];
                const float *bGp = lane >= 2 ? a.m.bG[1] : a.m.bG[0];
                const float *bMp = lane >= 2 ? a.m.bM[1] : a.m.bM[0];
                if ((BPR || LGT) && paired) {
                    if (!only_neg || lane == 1) {
                        const float2 pr = *reinterpret_cast<const float2 *>(a.bb[lane >= 2 ? 1 : 0] + 2 * (size_t)brow);
                        obW[gg] = pr.x;
                        obG[gg] = pr.y;
                    }
                } else if (!only_neg || lane == 1) {
                    obW[gg] = bWp[brow];
                    obG[gg] = bGp[brow];
                    if (ADADELTA) obM[gg] = bMp[brow];
                }
            };
            double pp = 0.0;
            int done = 0;  // draws consumed by every group that is still looking (wave-uniform)
            if constexpr (LGT) {
                // the pair's prediction (PYX:320-334) by lane 0 of the group; the accumulator rows of its update requested at once
                float bi = 0.0f;
                if (act && p == 0) bi = bi_tab[(size_t)c_pos * bstr_i];
                if constexpr (DMA4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two DMA'd rows have landed
                wave_sync();
                if constexpr (REG) RegScale::scales(a.reg_live, ib - a.begin, wi, wu);  // (float)(1.0 * scale), PYX:306; wave-uniform
                float score = 0.0f;
                if (act && p == 0) score = row_dot<REG>(urow, vrows, d, bu, bi, wu, wi);
                pp = (double)(LPR == 64 ? read_lanef(score, 0) : __shfl(score, gbase, WAVE));
                if (act) chosen = c_pos;  // (the "negative" of the shared update code: never read)
                spec_cand = c_pos;
                specm = __ballot(act && p == 0);
#pragma unroll
                for (int gg = 0; gg < NG; ++gg) {
                    if ((specm >> (gg * LPR)) & 1ull)
                        load_rows(gg, __builtin_amdgcn_readlane(c_user, gg * LPR), __builtin_amdgcn_readlane(c_pos, gg * LPR),
                                  __builtin_amdgcn_readlane(c_pos, gg * LPR), false);
                }
            }
            while (done < max_sampled) {
                bool need = act && chosen < 0;
                if constexpr (LPR == 64) need = __ballot(need) != 0ull;
                if (__ballot(need) == 0ull) break;
                const int nb = min(max_sampled - done, done == 0 ? a.first_batch : RG - 1);
                // lane p holds the stream after min(p, nb) further steps: draw #(done + p)
                uint32_t s = lcgA * state + lcgC;
                if (done != 0) {  // later batches (max_sampled above the tile's rows): step by step
                    s = state;
                    for (int j = 0; j < nb; ++j)
                        if (j < p) s = lcg(s);
                }
                int myitem;
                if constexpr (BPR) {  // PYX:1124-1125 (the first batch's lookups were made a pass ago)
                    if (done == 0) myitem = (p == 0 || !need) ? c_pos : pf_item;
                    else myitem = (p == 0 || !need) ? c_pos : a.item_ids[draw(s) % n_examples];
                }
                else myitem = (p == 0) ? c_pos : fast_mod(draw(s), n_items, magic);                        // PYX:860-861
                const bool rowlane = need && p <= nb && (p > 0 || done == 0);
                float bi = 0.0f;
                if (rowlane) bi = bi_tab[(size_t)myitem * bstr_i];
                // in_positives pre-filter (device.hpp: Bloom); early variant: the filter word of every candidate
                // travels with its embedding row (its address needs only the user's row bounds, prefetched a pass ahead)
                const uint32_t bh = Bloom::mix((uint32_t)myitem);
                uint32_t bword = 0xffffffffu;  // no filter: every candidate is "maybe a positive"
                if (bloom && bloom_early && rowlane && p > 0) bword = bloom[Bloom::word(bh, c_lo, c_hi)];
                // up to 10 candidate rows per round, ALL requested before the first is staged
                const bool gln = need && pc;
                if constexpr (DMA) {
                    if (need) {  // wave-uniform
                        for (int k = 1; k <= nb; ++k) {
                            const int neg = read_lane(myitem, k);
                            if (pc) dma_lane_dword(Wi + (size_t)neg * d + p, vrows + (size_t)k * TS);
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA'd rows have landed
                } else if constexpr (DMA4) {
                    // one instruction per candidate: the k-th row of every group that is still looking
                    // (a finished group's lanes stay masked, its chosen row survives)
#pragma unroll
                    for (int k = 1; k < (BPR ? 3 : LPR); ++k) {  // (BPR: batches of two candidates)
                        if (k <= nb) {  // wave-uniform
                            const int neg = row_bcast(myitem, k);
                            if (gln) dma_lane_x4(Wi + (size_t)neg * d + VEC * p, tile + (size_t)k * KS);
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // user, positive and candidate rows have landed
                } else if constexpr (LPR == 16) {
                    // a lane group is a DPP row: row_newbcast:k hands lane k's item to its 16
                    // lanes in one VALU instruction (no LDS round trip per candidate)
                    auto round = [&](auto K0) {
                        constexpr int k0 = decltype(K0)::value;
                        PieceT v[10];
#pragma unroll
                        for (int c5 = 0; c5 < 10; c5 += 5) {
                            if (k0 + c5 <= nb) {  // wave-uniform
#pragma unroll
                                for (int j = 0; j < 5; ++j) {
                                    // lanes past nb hold draw nb again: row nb is simply restaged
                                    const int neg = row_bcast(myitem, k0 + c5 + j);
                                    v[c5 + j] = ldp<VEC>(gln ? Wi + (size_t)neg * d + VEC * p : Wi);
                                }
                            }
                        }
                        // only groups still looking restage: a finished group's chosen row must survive
#pragma unroll
                        for (int c5 = 0; c5 < 10; c5 += 5) {
                            if (k0 + c5 <= nb && gln) {
#pragma unroll
                                for (int j = 0; j < 5; ++j)
                                    stp<VEC>(vrows + (size_t)min(k0 + c5 + j, nb) * TS + VEC * p, v[c5 + j]);
                            }
                        }
                    };
                    round(std::integral_constant<int, 1>());
                    if (nb > 10) round(std::integral_constant<int, 11>());
                } else {
                    for (int k0 = 1; k0 <= nb; k0 += 10) {
                        PieceT v[10];
                        int kk[10];
#pragma unroll
                        for (int c5 = 0; c5 < 10; c5 += 5) {
                            if (k0 + c5 <= nb) {  // wave-uniform
                                int negs[5];
#pragma unroll
                                for (int j = 0; j < 5; ++j) {
                                    kk[c5 + j] = min(k0 + c5 + j, nb);
                                    negs[j] = LPR == 64 ? read_lane(myitem, kk[c5 + j])
                                                        : __shfl(myitem, gbase + kk[c5 + j], WAVE);
                                }
#pragma unroll
                                for (int j = 0; j < 5; ++j)
                                    v[c5 + j] = ldp<VEC>(gln ? Wi + (size_t)negs[j] * d + VEC * p : Wi);
                            }
                        }
#pragma unroll
                        for (int c5 = 0; c5 < 10; c5 += 5) {
                            if (k0 + c5 <= nb && gln) {
#pragma unroll
                                for (int j = 0; j < 5; ++j)
                                    stp<VEC>(vrows + (size_t)kk[c5 + j] * TS + VEC * p, v[c5 + j]);
                            }
                        }
                    }
                }
                wave_sync();
                stamp(1);  // gathers landed and staged
                if constexpr (REG) {
                    if (done == 0) RegScale::scales(a.reg_live, ib - a.begin, wi, wu);  // (float)(1.0 * scale), PYX:306; wave-uniform
                }
                float score = 0.0f;
                if (rowlane) score = row_dot<REG>(urow, vrows + (size_t)p * KS, d, bu, bi, wu, wi);
                // (one interaction per wavefront: lane reads instead of shuffles keep the whole
                // sampling control flow below in SGPRs and scalar branches)
                if (done == 0) pp = (double)(LPR == 64 ? read_lanef(score, 0) : __shfl(score, gbase, WAVE));
                // PYX:875 compares doubles: negative_prediction > positive_prediction - 1
                // (BPR: every candidate is examined in draw order, PYX:1126-1127)
                const bool viol = need && p >= 1 && p <= nb && (BPR || (double)score > pp - 1.0);
                unsigned long long vm = (__ballot(viol) >> gbase) & GM;
                if (bloom && !bloom_early && viol) bword = bloom[Bloom::word(bh, c_lo, c_hi)];  // probed for the violators only
                int used = nb;
                stamp(2);  // scoring pass
                if (done == 0) {
                    // The first violator is almost always the choice (a uniformly drawn item is
                    // rarely one of the user's positives): request the accumulator rows of its
                    // update now, so they travel while in_positives confirms it.
                    const int r1 = vm != 0ull ? (__ffsll((long long)vm) - 1) : 0;
                    spec_cand = LPR == 64 ? read_lane(myitem, r1) : __shfl(myitem, gbase + r1, WAVE);
                    specm = __ballot(need && vm != 0ull && p == 0);
#pragma unroll
                    for (int gg = 0; gg < NG; ++gg) {
                        if ((specm >> (gg * LPR)) & 1ull)
                            load_rows(gg, __builtin_amdgcn_readlane(c_user, gg * LPR),
                                      __builtin_amdgcn_readlane(c_pos, gg * LPR),
                                      __builtin_amdgcn_readlane(spec_cand, gg * LPR), false);
                    }
                }
                // lane r's verdict on ITS candidate: all three filter bits set = maybe a positive
                const uint32_t bmask = Bloom::mask(bh);
                const int maybe_pos = ((bword & bmask) == bmask) ? 1 : 0;
                while (true) {
                    const bool part = need && chosen < 0 && vm != 0ull;
                    if (__ballot(part) == 0ull) break;
                    const int r = part ? (__ffsll((long long)vm) - 1) : 0;
                    if (part) vm &= vm - 1ull;
                    const int cand = LPR == 64 ? read_lane(myitem, r) : __shfl(myitem, gbase + r, WAVE);
                    float cand_score = 0.0f;
                    if constexpr (BPR) cand_score = LPR == 64 ? read_lanef(score, r) : __shfl(score, gbase + r, WAVE);
                    // the exact search (PYX:270-284) only where the filter cannot rule the candidate out
                    const bool ask = part && (LPR == 64 ? read_lane(maybe_pos, r) : __shfl(maybe_pos, gbase + r, WAVE)) != 0;
                    bool found = false;
                    if (__ballot(ask) != 0ull)
                        found = group_in_positives<LPR>(indices, cand, LPR == 64 ? uni(c_lo) : c_lo,
                                                        LPR == 64 ? uni(c_hi) : c_hi, ask, gbase, p);
                    c3 += (uint32_t)__popcll(__ballot(part && p == 0));  // PYX:878-879: the draw still counts
                    if (part) {
                        if (!found) {
                            chosen = cand;
                            chosen_r = r;
                            chosen_score = cand_score;
                            used = r;
                        }
                    }
                }
                const uint32_t ns = (uint32_t)(LPR == 64 ? read_lane((int)s, used) : __shfl((int)s, gbase + used, WAVE));
                if constexpr (BPR) {
                    if (done + nb >= max_sampled) {  // PYX:1123-1127: the loop ends on its bound -- the last draw is the negative
                        const int lc = LPR == 64 ? read_lane(myitem, nb) : __shfl(myitem, gbase + nb, WAVE);
                        const float ls = LPR == 64 ? read_lanef(score, nb) : __shfl(score, gbase + nb, WAVE);
                        if (need && chosen < 0) {
                            chosen = lc;
                            chosen_r = nb;
                            chosen_score = ls;
                        }
                    }
                }
                if (need) {
                    sampled += used;
                    state = ns;
                }
                wave_sync();
                done += nb;
                stamp(3);  // in_positives searches
            }
            c0 += (uint32_t)__popcll(__ballot(act && p == 0 && (!LGT || c_y > 0.0f)));
            c2 += (uint32_t)__popcll(__ballot(act && chosen >= 0 && p == 0));
#pragma unroll
            for (int gg = 0; gg < NG; ++gg)  // inactive groups hold sampled == 0
                c1 += (uint32_t)__builtin_amdgcn_readlane(sampled, gg * LPR);

            // ---- updates: one group at a time, the whole wave on its three rows ----
            // The prefetched records are forced into registers on EVERY path before any atomic
            // is issued: the next pass then never waits for them behind the atomics'
            // acknowledgements.
            asm volatile("" : "+v"(n_lo), "+v"(n_hi), "+v"(row3), "+v"(rec2.x), "+v"(rec2.y), "+v"(rec2.z),
                         "+v"(rec2.w));
            if constexpr (BPR) asm volatile("" : "+v"(pf_next));
            double lossd = 0.0;
            if (act && chosen >= 0) {
                if constexpr (BPR) {
                    lossd = (double)c_w * (1.0 - (double)sigmoidf_ref((float)(pp - (double)chosen_score)));  // PYX:1158
                } else if constexpr (LGT) {
                    lossd = (double)c_w * ((double)sigmoidf_ref((float)pp) - (c_y <= 0.0f ? 0.0 : 1.0));    // PYX:745-757
                } else {
                    lossd = (double)c_w * a.logtab[sampled];  // PYX:881-885, log from host libm
                    if (lossd > MAX_LOSS) lossd = MAX_LOSS;
                }
            }
            const unsigned long long upd = __ballot(act && chosen >= 0 && p == 0);
            if (upd != 0ull) {
                // accumulators not requested early (violator found in a later batch), or requested
                // for a candidate that turned out to be a positive
#pragma unroll
                for (int gg = 0; gg < NG; ++gg) {
                    if ((upd >> (gg * LPR)) & 1ull) {
                        const int neg = __builtin_amdgcn_readlane(chosen, gg * LPR);
                        const bool early = (specm >> (gg * LPR)) & 1ull;
                        if (!early || neg != __builtin_amdgcn_readlane(spec_cand, gg * LPR))
                            load_rows(gg, __builtin_amdgcn_readlane(c_user, gg * LPR),
                                      __builtin_amdgcn_readlane(c_pos, gg * LPR), neg, early);
                    }
                }
                // Every accumulator must be in registers before the first atomic is issued: a wait for
                // one of them placed later would also wait for that atomic's acknowledgement
                // (vmcnt is an in-order counter).
#pragma unroll
                for (int gg = 0; gg < NG; ++gg) {
                    if ((upd >> (gg * LPR)) & 1ull) {
#pragma unroll
                        for (int q = 0; q < NC; ++q) {
                            asm volatile("" : "+v"(gP[gg][q]), "+v"(gN[gg][q]), "+v"(gU[gg][q]));
                            if (ADADELTA) asm volatile("" : "+v"(mP[gg][q]), "+v"(mN[gg][q]), "+v"(mU[gg][q]));
                        }
                        asm volatile("" : "+v"(obW[gg]), "+v"(obG[gg]));
                        if (ADADELTA) asm volatile("" : "+v"(obM[gg]));
                    }
                }
                stamp(4);  // accumulator rows landed
                float avg_lr[NG];  // REG: average learning rate of each updated interaction (wave-uniform)
#pragma unroll
                for (int gg = 0; gg < NG; ++gg) avg_lr[gg] = 0.0f;
                const double ia = REG ? a.item_alpha : 0.0, ua = REG ? a.user_alpha : 0.0;
                // cell arithmetic (PYX:416-449 in float64) and atomic publication
#pragma unroll
                for (int gg = 0; gg < NG; ++gg) {
                    if ((upd >> (gg * LPR)) & 1ull) {
                        const int user = __builtin_amdgcn_readlane(c_user, gg * LPR);
                        const int pos = __builtin_amdgcn_readlane(c_pos, gg * LPR);
                        const int neg = __builtin_amdgcn_readlane(chosen, gg * LPR);
                        const int cr = __builtin_amdgcn_readlane(chosen_r, gg * LPR);
                        const double loss = read_laned(lossd, gg * LPR);
                        const size_t bp = (size_t)pos * d, bn = (size_t)neg * d, bu_ = (size_t)user * d;
                        const float *tu = tile + UB + (size_t)gg * US;
                        const float *tp = tile + (size_t)gg * GS;
                        const float *tn = tp + (size_t)cr * KS;
                        // All cell arithmetic of the group first, as ONE straight-line block (the
                        // 3*NC row cells and the bias cell are independent float64 dependency
                        // chains the scheduler interleaves), then every publication.
                        float oWr[NC][3], nWr[NC][3], nGr[NC][3], nMr[NC][3];
                        double lr, lr_acc = 0.0;
#pragma unroll
                        for (int q = 0; q < NC; ++q) {
                            const int c = lane + WAVE * q;
                            const int cc = c < d ? c : 0;
                            const float Ur = tu[cc], Pr = tp[cc], Nr = tn[cc];  // raw embedding cells
                            // representation cells (identity features): fl(w * x), PYX:306-313
                            const float Uc = REG ? __fmul_rn(wu, Ur) : Ur, Pc = REG ? __fmul_rn(wi, Pr) : Pr,
                                        Nc = REG ? __fmul_rn(wi, Nr) : Nr;
                            const double u = (double)Uc;
                            const double df = (double)__fsub_rn(Nc, Pc);  // float32 subtraction, PYX:634-635
                            oWr[q][0] = Pr;
                            oWr[q][1] = Nr;
                            oWr[q][2] = Ur;
                            cell_math(Pr, gP[gg][q], ADADELTA ? mP[gg][q] : 0.0f, 1.0, (LGT ? loss : -loss) * u, h, ia,
                                      nWr[q][0], nGr[q][0], nMr[q][0], lr);
                            if (REG && c < a.m.d_real) lr_acc += lr;
                            if constexpr (REG) __builtin_amdgcn_sched_barrier(0);  // one cell's float64 chain at a time: registers
                            if constexpr (BPR) {
                                // the negative IS the positive (a user with the whole catalogue: the draw loop ended on its bound):
                                // the reference updates that row twice in sequence -- the second pass starts from what the first leaves
                                if (neg == pos) {
                                    oWr[q][1] = nWr[q][0];
                                    gN[gg][q] = nGr[q][0];
                                }
                            }
                            if constexpr (!LGT)
                                cell_math(oWr[q][1], gN[gg][q], ADADELTA ? mN[gg][q] : 0.0f, 1.0, loss * u, h, ia,
                                          nWr[q][1], nGr[q][1], nMr[q][1], lr);
                            else nWr[q][1] = nGr[q][1] = nMr[q][1] = 0.0f;
                            if (REG && !LGT && c < a.m.d_real) lr_acc += lr;
                            if constexpr (REG) __builtin_amdgcn_sched_barrier(0);
                            // (logistic: the user's row moves along the item's, PYX:519-533)
                            cell_math(Ur, gU[gg][q], ADADELTA ? mU[gg][q] : 0.0f, 1.0, loss * (LGT ? (double)Pc : df), h, ua,
                                      nWr[q][2], nGr[q][2], nMr[q][2], lr);
                            if (REG && c < a.m.d_real) lr_acc += lr;
                            if constexpr (REG) __builtin_amdgcn_sched_barrier(0);
                        }
                        float bnW, bnG, bnM;
                        const float ooM = ADADELTA ? obM[gg] : 0.0f;
                        const double balpha = lane == 2 ? ua : ia;
                        cell_math(obW[gg], obG[gg], ooM, 1.0, (lane == 0 && !LGT) ? -loss : loss, h, balpha, bnW, bnG, bnM, lr);
                        if constexpr (BPR) {
                            if (neg == pos) {  // (the bias cell of that row likewise: lane 1 continues from lane 0's result)
                                const float w0 = read_lanef(bnW, 0), g0 = read_lanef(bnG, 0);
                                if (lane == 1) {
                                    obW[gg] = w0;
                                    obG[gg] = g0;
                                    cell_math(w0, g0, ooM, 1.0, loss, h, balpha, bnW, bnG, bnM, lr);
                                }
                            }
                        }
                        if constexpr (REG) {
                            // avg_learning_rate of PYX:640-646: the 3 (d + 1) cells of three identity rows
                            // (logistic, PYX:527-530: the 2 (d + 1) cells of two identity rows)
                            if (lane < 3 && !(LGT && lane == 1)) lr_acc += lr;
                            avg_lr[gg] = unif((float)(wave_sum(lr_acc) / (double)((LGT ? 2 : 3) * (a.m.d_real + 1))));
                        }
                        // keep the arithmetic above one block: nothing of it may sink below a publication
#pragma unroll
                        for (int q = 0; q < NC; ++q)
                            asm volatile("" : "+v"(nWr[q][0]), "+v"(nWr[q][1]), "+v"(nWr[q][2]), "+v"(nGr[q][0]),
                                         "+v"(nGr[q][1]), "+v"(nGr[q][2]));
                        asm volatile("" : "+v"(bnW), "+v"(bnG));
#pragma unroll
                        for (int q = 0; q < NC; ++q) {
                            const int c = lane + WAVE * q;
                            if (c < d) {
                                // uniform row bases + a 32-bit lane offset the compiler cannot hoist (see load_rows)
                                unsigned cq = (unsigned)c;
                                asm volatile("" : "+v"(cq));
                                float *const wP = WiW + bp, *const wN = WiW + bn, *const wU = WuW + bu_;
                                float *const aP = Gi + bp, *const aN = Gi + bn, *const aU = Gu + bu_;
                                if constexpr (ADADELTA) {
                                    // moving-average accumulators: compare-and-swap (device.hpp: publish_adadelta)
                                    const float Uc = REG ? __fmul_rn(wu, oWr[q][2]) : oWr[q][2];
                                    const float Pc = REG ? __fmul_rn(wi, oWr[q][0]) : oWr[q][0];
                                    const float Nc = REG ? __fmul_rn(wi, oWr[q][1]) : oWr[q][1];
                                    const double u = (double)Uc;
                                    const double df = (double)__fsub_rn(Nc, Pc);
                                    publish_cell(wP + cq, aP + cq, Mi + bp + cq, oWr[q][0], gP[gg][q], mP[gg][q],
                                                 nWr[q][0], nGr[q][0], nMr[q][0], 1.0, -loss * u, h, ia, um);
                                    publish_cell(wN + cq, aN + cq, Mi + bn + cq, oWr[q][1], gN[gg][q], mN[gg][q],
                                                 nWr[q][1], nGr[q][1], nMr[q][1], 1.0, loss * u, h, ia, um);
                                    publish_cell(wU + cq, aU + cq, Mu + bu_ + cq, oWr[q][2], gU[gg][q], mU[gg][q],
                                                 nWr[q][2], nGr[q][2], nMr[q][2], 1.0, loss * df, h, ua, um);
                                } else {
                                    publish(wP + cq, nWr[q][0], oWr[q][0], um);
                                    publish(aP + cq, nGr[q][0], gP[gg][q], um);
                                    if constexpr (!LGT) {
                                        publish(wN + cq, nWr[q][1], oWr[q][1], um);
                                        publish(aN + cq, nGr[q][1], gN[gg][q], um);
                                    }
                                    publish(wU + cq, nWr[q][2], oWr[q][2], umU);  // (FitArgs::user_store: plain stores)
                                    publish(aU + cq, nGr[q][2], gU[gg][q], umU);
                                }
                            }
                        }
                        if ((BPR || LGT) && paired) {
                            // lanes 0-2: the W cell's delta; lanes 3-5: the accumulator cell's, handed over from lanes 0-2
                            const float dWb = __fsub_rn(bnW, obW[gg]), dGb = __fsub_rn(bnG, obG[gg]);
                            const float dGs = __shfl(dGb, lane >= 3 ? lane - 3 : lane, WAVE);
                            const int role = lane < 3 ? lane : lane - 3;
                            if (lane < 6 && !(LGT && role == 1)) {
                                const int brow = role == 0 ? pos : (role == 1 ? neg : user);
                                const float dl = lane < 3 ? dWb : dGs;
                                if (dl != 0.0f) atomicAdd(a.bb[role == 2 ? 1 : 0] + 2 * (size_t)brow + (lane < 3 ? 0 : 1), dl);
                            }
                        } else if (lane < 3 && !(LGT && lane == 1)) {
                            const int brow = lane == 0 ? pos : (lane == 1 ? neg : user);
                            float *bWp = lane == 2 ? a.m.b[1] : a.m.b[0];
                            float *bGp = lane == 2 ? a.m.bG[1] : a.m.bG[0];
                            float *bMp = lane == 2 ? a.m.bM[1] : a.m.bM[0];
                            publish_cell(bWp + brow, bGp + brow, bMp + brow, obW[gg], obG[gg], ooM, bnW, bnG, bnM, 1.0,
                                         (lane == 0 && !LGT) ? -loss : loss, h, balpha, um);
                        }
                    }
                }
                if constexpr (REG) {
                    // PYX:648-649 for the pass's interactions (after the publications: the logarithms'
                    // temporaries must not overlap the cell arithmetic's registers); collected in registers,
                    // published every RegScale::PERIOD passes
                    float add_i = 0.0f, add_u = 0.0f;
#pragma unroll
                    for (int gg = 0; gg < NG; ++gg) {
                        if ((upd >> (gg * LPR)) & 1ull) {
                            add_i += RegScale::log1p_f32((float)ia * avg_lr[gg]);
                            add_u += RegScale::log1p_f32((float)ua * avg_lr[gg]);
                        }
                    }
                    if (um != 2) reg.add(add_i, add_u);
                }
                wave_sync();  // the tile is rewritten by the next pass
                stamp(5);  // cell arithmetic, atomics issued (acknowledged, in the timed build)
            }
        }
        else {
            asm volatile("" : "+v"(n_lo), "+v"(n_hi), "+v"(row3), "+v"(rec2.x), "+v"(rec2.y), "+v"(rec2.z),
                         "+v"(rec2.w));
        }
        stamp(6);
        if (in && p == 0) {
            if (a.neg_log) a.neg_log[i] = chosen;
            if (a.sampled_log) a.sampled_log[i] = sampled;
        }
        cur = nxt;
        c_lo = n_lo;
        c_hi = n_hi;
        nxt = rec2;
        row2 = row3;
        if constexpr (BPR) pf_item = pf_next;
    }
    if constexpr (REG) RegScale::publish(a.reg_live, reg.p_i, reg.p_u, lane, (unsigned)gw);

    // counters: one atomic per wave and counter
    if constexpr (TIMED) {
        if (lane == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(a.counters + 4 + k, ph[k]);
    }
    if (lane == 0) {
        if (c0) atomicAdd(a.counters + 0, (unsigned long long)c0);
        if (c1) atomicAdd(a.counters + 1, (unsigned long long)c1);
        if (c2) atomicAdd(a.counters + 2, (unsigned long long)c2);
        if (c3) atomicAdd(a.counters + 3, (unsigned long long)c3);
    }
}

// Launch helper shared by the per-LPR translation units (warp_tile_lpr*.hip).
template <int LPR, int VEC, bool DMA4 = false>
hipError_t launch_tile_variant(const FitArgs &a, int grid, size_t smem, hipStream_t st, int cus, bool timed,
                               int *grid_used)
{
    void (*kernel)(FitArgs);
    const bool reg = a.item_alpha != 0.0 || a.user_alpha != 0.0;
    if (a.m.adadelta && reg) return hipErrorInvalidValue;  // session.hip routes adadelta + regularisation to the generic kernel
    if (a.m.adadelta) kernel = fit_warp_tile_kernel<LPR, VEC, false, true, DMA4>;
    else if (reg) kernel = fit_warp_tile_kernel<LPR, VEC, false, false, DMA4, true>;
    else if (timed) kernel = fit_warp_tile_kernel<LPR, VEC, true, false, DMA4>;
    else kernel = fit_warp_tile_kernel<LPR, VEC, false, false, DMA4>;
    if (cus > 0) {
        const int per_cu = occupancy_cached(kernel, 256, smem);
        if (per_cu > 0) grid = std::min(grid, per_cu * cus);
    }
    if (grid_used) *grid_used = grid;
    kernel<<<grid, 256, smem, st>>>(a);
    return hipGetLastError();
}

// ... and of the BPR / logistic instantiations (warp_tile_bpr.hip)
template <int LPR, int VEC, bool DMA4 = false>
hipError_t launch_tile_bpr_variant(const FitArgs &a, int grid, size_t smem, hipStream_t st, int cus, int *grid_used, bool logistic)
{
    const bool reg = a.item_alpha != 0.0 || a.user_alpha != 0.0;
    void (*kernel)(FitArgs);
    if (logistic) kernel = reg ? fit_warp_tile_kernel<LPR, VEC, false, false, DMA4, true, LFM_LOSS_LOGISTIC_ID>
                               : fit_warp_tile_kernel<LPR, VEC, false, false, DMA4, false, LFM_LOSS_LOGISTIC_ID>;
    else kernel = reg ? fit_warp_tile_kernel<LPR, VEC, false, false, DMA4, true, LFM_LOSS_BPR_ID>
                      : fit_warp_tile_kernel<LPR, VEC, false, false, DMA4, false, LFM_LOSS_BPR_ID>;
    if (a.m.adadelta) return hipErrorInvalidValue;
    if (cus > 0) {
        const int per_cu = occupancy_cached(kernel, 256, smem);
        if (per_cu > 0) grid = std::min(grid, per_cu * cus);
    }
    if (grid_used) *grid_used = grid;
    kernel<<<grid, 256, smem, st>>>(a);
    return hipGetLastError();
}

}  // namespace lfm
