// predict_kernels.hip -- scoring / evaluation kernels.
//
// predict_lightfm          PYX:1185-1229
// predict_ranks            PYX:1232-1323
// calculate_auc_from_rank  PYX:1326-1376
// All scores use the reference's sequential float32 summation order, so ranks
// (integer counts of score comparisons) are bit-identical.
#include "device.hpp"
#include "kernels.hpp"

namespace lfm {

// One wavefront scores tile_rows/2 (user, item) pairs per pass.
template <int NC>
__global__ __launch_bounds__(256) void predict_kernel(PredictArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), wib = threadIdx.x >> 6, d = a.m.d, TS = a.tile_stride;
    float *tile = smem + (size_t)wib * a.tile_rows * TS;
    const int PP = a.tile_rows / 2;
    const int64_t gw = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wib;
    const int64_t nw = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    for (int64_t i0 = gw * PP; i0 < a.n; i0 += nw * PP) {
        int np_ = (int)min((int64_t)PP, a.n - i0);
        for (int p = 0; p < np_; ++p) {
            int user = uni(a.uids[i0 + p]), item = uni(a.iids[i0 + p]);
            Rep<NC> U, I;
            load_rep<NC>(a.usf, a.m.W[1], a.m.b[1], d, user, 1.0, lane, U);
            load_rep<NC>(a.itf, a.m.W[0], a.m.b[0], d, item, 1.0, lane, I);
            rep_to_tile<NC>(tile + (size_t)(2 * p) * TS, U, d, lane);
            rep_to_tile<NC>(tile + (size_t)(2 * p + 1) * TS, I, d, lane);
        }
        wave_sync();
        if (lane < np_)
            a.out[i0 + lane] = tile_dot(tile + (size_t)(2 * lane) * TS, tile + (size_t)(2 * lane + 1) * TS, d);
        wave_sync();
    }
}

// Dense representations of every row of f.  Row-major (out[row*rs + c]) or, with `transposed`,
// component-major (out[c*rows + row]): the layout predict_ranks reads coalesced across items.
template <int NC>
__global__ __launch_bounds__(256) void rep_rows_kernel(DCsr f, const float *W, const float *b, int d,
                                                       int rs, float *out, int transposed, float *bias_out)
{
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    for (int64_t row = gw; row < f.rows; row += nw) {
        Rep<NC> r;
        load_rep<NC>(f, W, b, d, (int)row, 1.0, lane, r);
        float *o = transposed ? out + row : out + (size_t)row * rs;
        const size_t cs = transposed ? (size_t)f.rows : 1;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            int c = lane + WAVE * q;
            if (c < d) o[c * cs] = r.v[q];
        }
        if (lane == 0) {
            if (bias_out) bias_out[row] = r.bias;
            else o[d * cs] = r.bias;
        }
    }
}

// Sequential float32 dot (PYX:320-334) of the user's representation (LDS) with item `j` of the
// component-major item table: lanes of a wavefront read consecutive items, so every load is
// coalesced.
__device__ __forceinline__ float dense_dot(const float *u, const float *vT, size_t n_items, int j, int d)
{
    float acc = __fadd_rn(u[d], vT[(size_t)d * n_items + j]);
    int c = 0;
    for (; c + 8 <= d; c += 8) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = vT[(size_t)(c + k) * n_items + j];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = __fadd_rn(acc, __fmul_rn(u[c + k], x[k]));
    }
    for (; c < d; ++c) acc = __fadd_rn(acc, __fmul_rn(u[c], vT[(size_t)c * n_items + j]));
    return acc;
}

__device__ __forceinline__ bool bsearch_row(const DCsr &m, int row, int item)
{
    int lo = m.indptr[row], hi = m.indptr[row + 1];
    while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        int v = m.indices[mid];
        if (v == item) return true;
        if (v < item) lo = mid + 1; else hi = mid;
    }
    return false;
}

constexpr int RANK_CHUNK = 256;

// One 256-thread workgroup per user with test interactions (PYX:1264-1319).
__global__ __launch_bounds__(256) void ranks_kernel(RanksArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *urep = smem;                               // [rs]
    float *tscore = smem + a.rs;                      // [RANK_CHUNK]
    int *tid_ = reinterpret_cast<int *>(tscore + RANK_CHUNK);  // [RANK_CHUNK]
    int *tcount = tid_ + RANK_CHUNK;                  // [RANK_CHUNK]
    const int lane = lane_id();
    for (int user = blockIdx.x; user < a.test.rows; user += gridDim.x) {
        int rs0 = a.test.indptr[user], re0 = a.test.indptr[user + 1];
        if (re0 == rs0) continue;
        __syncthreads();
        for (int c = threadIdx.x; c <= a.d; c += blockDim.x) urep[c] = a.user_rep[(size_t)user * a.rs + c];
        for (int c0 = rs0; c0 < re0; c0 += RANK_CHUNK) {
            int m = min(RANK_CHUNK, re0 - c0);
            __syncthreads();
            if ((int)threadIdx.x < m) {
                int it = a.test.indices[c0 + threadIdx.x];
                tid_[threadIdx.x] = it;
                tscore[threadIdx.x] = dense_dot(urep, a.item_rep, (size_t)a.test.cols, it, a.d);
                tcount[threadIdx.x] = 0;
            }
            __syncthreads();
            int n_items = a.test.cols;
            int rounds = (n_items + blockDim.x - 1) / blockDim.x;
            for (int r = 0; r < rounds; ++r) {
                int j = r * blockDim.x + threadIdx.x;
                bool live = j < n_items && !bsearch_row(a.train, user, j);  // PYX:1303-1304
                float sj = 0.0f;
                if (live) sj = dense_dot(urep, a.item_rep, (size_t)a.test.cols, j, a.d);
                for (int t = 0; t < m; ++t) {
                    bool hit = live && (j != tid_[t]) && (sj >= tscore[t]);  // PYX:1317-1319
                    unsigned long long mk = __ballot(hit);
                    if (lane == 0 && mk) atomicAdd(&tcount[t], __popcll(mk));
                }
            }
            __syncthreads();
            if ((int)threadIdx.x < m) a.ranks[c0 + threadIdx.x] += (float)tcount[threadIdx.x];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// predict_ranks as a tiled dense pass (the "predict-all-items" path): one wavefront owns 32 users
// and sweeps the item table in tiles of 32; the 32 x 32 scores of a tile come from 32 (d = 64)
// v_mfma_f32_32x32x2_f32 whose accumulators start at (user bias + item bias) and run over the
// components in the reference's order.  An MFMA step is a FUSED multiply-add, the reference
// (PYX:320-334, -ffp-contract=off) rounds every product: the two chains differ by at most
// eps(u, j) = 4 (d + 2) 2^-24 (|b_u| + |b_j| + |u| |v_j|).  So the MFMA score only PRE-FILTERS: a
// comparison against a test item's (exact, sequential-dot) score that lands inside +-eps is
// re-decided with the exact sequential dot.  Ranks therefore stay integer-identical to the
// reference's.  Train positives are masked by walking each user's sorted train row alongside the
// item sweep (one 32-bit mask per user and tile) instead of a binary search per (user, item).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MT>
__global__ __launch_bounds__(64) void ranks_mfma_kernel(RanksArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
    const int d = a.d, AS = d + 1, I = a.test.cols;
    float *A = smem;                                     // [32][AS] user representations, bias at [d]
    float *thr = A + 32 * AS;                            // [32][MT] exact scores of the test items
    int *tid = reinterpret_cast<int *>(thr + 32 * MT);   // [32][MT] their item ids
    int *cnt = tid + 32 * MT;                            // [32][MT] items ranked at or above them
    float *unorm = reinterpret_cast<float *>(cnt + 32 * MT);  // [32] |u|
    int *ucnt = reinterpret_cast<int *>(unorm + 32);          // [32] test items of the user in this pass
    const float kappa = 4.0f * (float)(d + 2) * 5.9604645e-8f;
    const float *vT = a.item_rep;
    for (int tile = blockIdx.x; tile * 32 < a.n_ulist; tile += gridDim.x) {
        wave_sync();
        for (int idx = lane; idx < 32 * AS; idx += WAVE) {
            const int r = idx / AS, c = idx - r * AS, ui = tile * 32 + r;
            A[idx] = ui < a.n_ulist ? a.user_rep[(size_t)a.ulist[ui] * a.rs + c] : 0.0f;
        }
        wave_sync();
        const int ui = tile * 32 + col;
        const bool uok = ui < a.n_ulist;
        const int user = uok ? a.ulist[ui] : 0;
        int t_lo = 0, t_hi = 0;
        if (uok && half == 0) {
            t_lo = a.test.indptr[user];
            t_hi = a.test.indptr[user + 1];
            float n2 = 0.0f;
            for (int c = 0; c < d; ++c) n2 += A[col * AS + c] * A[col * AS + c];
            unorm[col] = sqrtf(n2) * 1.0000005f;
        }
        int m_max = t_hi - t_lo;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m_max = max(m_max, __shfl_xor(m_max, off, WAVE));
        for (int p0 = 0; p0 < m_max; p0 += MT) {
            // ---- this pass's test items: ids and exact scores (PYX:1278-1293)
            wave_sync();
            if (half == 0) {
                const int m = uok ? max(0, min(MT, (t_hi - t_lo) - p0)) : 0;
                ucnt[col] = m;
                for (int t = 0; t < m; ++t) {
                    const int it = a.test.indices[t_lo + p0 + t];
                    tid[col * MT + t] = it;
                    thr[col * MT + t] = dense_dot(A + col * AS, vT, (size_t)I, it, d);
                    cnt[col * MT + t] = 0;
                }
            }
            wave_sync();
            // train row cursor of the user (lanes 0..31)
            int tp = 0, tend = 0, next = 0x7fffffff;
            if (uok && half == 0) {
                tp = a.train.indptr[user];
                tend = a.train.indptr[user + 1];
                if (tp < tend) next = a.train.indices[tp];
            }
            for (int j0 = 0; j0 < I; j0 += 32) {
                const int j = j0 + col;
                const bool jok = j < I;
                const float bj = jok ? vT[(size_t)d * I + j] : 0.0f;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                    acc[r] = __fadd_rn(A[i * AS + d], bj);
                }
                float n2 = 0.0f;
                for (int k0 = 0; k0 < d; k0 += 2) {
                    const int k = k0 + half;
                    const bool kok = k < d;
                    const float av = kok ? A[col * AS + k] : 0.0f;
                    const float bv = (kok && jok) ? vT[(size_t)k * I + j] : 0.0f;
                    n2 += bv * bv;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
                }
                n2 += __shfl_xor(n2, 32, WAVE);
                const float nj = sqrtf(n2) * 1.0000005f;
                // train positives inside [j0, j0 + 32) (PYX:1303-1304)
                unsigned tmask = 0u;
                while (next < j0 + 32) {
                    tmask |= 1u << (next - j0);
                    ++tp;
                    next = tp < tend ? a.train.indices[tp] : 0x7fffffff;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int iA = (r & 3) + 8 * (r >> 2), iB = iA + 4, i = half ? iB : iA;
                    const unsigned um = half ? (unsigned)read_lane((int)tmask, iB) : (unsigned)read_lane((int)tmask, iA);
                    const int mtA = uni(ucnt[iA]), mtB = uni(ucnt[iB]);
                    const int mt = max(mtA, mtB), mine = half ? mtB : mtA;
                    if (mt == 0) continue;
                    const bool valid = jok && !((um >> col) & 1u);
                    const float eps = kappa * (fabsf(A[i * AS + d]) + fabsf(bj) + unorm[i] * nj);
                    const float sc = acc[r];
                    for (int t = 0; t < mt; ++t) {
                        const float th = thr[i * MT + t];
                        const int id = tid[i * MT + t];
                        const bool live = valid && t < mine && j != id;  // PYX:1317-1319
                        const float df = sc - th;
                        bool hit = live && df > eps;
                        const bool band = live && !(df > eps) && !(df < -eps);
                        if (__ballot(band) != 0ull) {
                            if (band) hit = dense_dot(A + i * AS, vT, (size_t)I, j, d) >= th;
                        }
                        const unsigned long long hm = __ballot(hit);
                        const int c = half ? __popc((unsigned)(hm >> 32)) : __popc((unsigned)hm);
                        if (col == 0 && c) cnt[i * MT + t] += c;
                    }
                }
            }
            wave_sync();
            if (half == 0) {
                const int m = ucnt[col];
                for (int t = 0; t < m; ++t) a.ranks[t_lo + p0 + t] += (float)cnt[col * MT + t];
            }
        }
    }
}

// Per item: the two item-side terms of the pre-filter's error bound, kappa |b_j| and kappa |v_j|
// (out[j], out[n_items + j]), from the component-major table.
__global__ void item_eps_kernel(const float *vT, int n_items, int d, float kappa, float *out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_items) return;
    float n2 = 0.0f;
    for (int k = 0; k < d; ++k) {
        const float v = vT[(size_t)k * n_items + j];
        n2 += v * v;
    }
    out[j] = kappa * fabsf(vT[(size_t)d * n_items + j]);
    out[(size_t)n_items + j] = kappa * (sqrtf(n2) * 1.0000005f);
}

// ---------------------------------------------------------------------------------------------
// predict_ranks, second MFMA formulation: USERS are the columns of the 32 x 32 output tile and
// ITEMS its rows, so a lane owns ONE user for the whole sweep over the item table:
//   * its user's representation (the B operand of every v_mfma_f32_32x32x2_f32 of the sweep), the
//     exact scores of its test items (thresholds) and their counters live in registers;
//   * per 32-item tile: 32 coalesced loads of the component-major item table (the A operand,
//     requested for the NEXT tile while this tile's scores are compared), d/2 MFMAs, and per
//     score four VALU instructions per threshold -- `count += (score - eps > threshold)`
//     (v_cmp + v_addc) and the distance to the nearest threshold (v_sub + v_min);
//   * a score within eps of a threshold is re-decided with the reference's sequential dot
//     (rare: eps ~ 1e-5 of the score scale), so the ranks are the reference's integers;
//   * train positives: the lane walks its user's sorted train row alongside the sweep (one
//     32-bit mask per tile), the next entry always requested one step ahead.
// One wavefront = 32 users; users are ordered by their number of test items (host side) so the
// 32 users of a wavefront need the same number of passes of 16 thresholds.
// Scores within eps of a threshold, re-decided with the reference's sequential dot (PYX:1317-1319).
// Kept out of line: it runs for a few scores per thousand and must not cost the sweep registers.
__device__ __forceinline__ void band_recheck(unsigned bandmask, int lane, int j0, int m, float nu, float eu,
                                          const float *urow, const float *vT, int I, int d, const float *sc_s,
                                          const float *ej_s, const float *nj_s, const float *thr_s,
                                          const int *tid_s, int *xcnt_s)
{
    constexpr int MT = 16;
    const int half = lane >> 5;
    // rare path: keep its address arithmetic here (opaque copies, so that nothing of it is hoisted
    // into the sweep and kept in registers there)
    asm volatile("" : "+s"(I), "+s"(d), "+s"(vT));
    while (bandmask) {
        const int r = __ffs((int)bandmask) - 1;
        bandmask &= bandmask - 1u;
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half, item = j0 + i;
        const float sc = sc_s[r * WAVE + lane];
        const float eps = __fmaf_rn(nu, nj_s[i], eu + ej_s[i]);
        bool need = false;
        for (int t = 0; t < m; ++t)
            need = need || (fabsf(sc - thr_s[lane * MT + t]) <= eps && item != tid_s[lane * MT + t]);
        if (need) {
            // the reference's sequential dot (PYX:320-334), one component at a time: few registers
            float ex = __fadd_rn(urow[d], vT[(size_t)d * I + item]);
#pragma unroll 1
            for (int c = 0; c < d; ++c) ex = __fadd_rn(ex, __fmul_rn(urow[c], vT[(size_t)c * I + item]));
            for (int t = 0; t < m; ++t) {
                const float tht = thr_s[lane * MT + t];
                if (fabsf(sc - tht) <= eps && item != tid_s[lane * MT + t] && ex >= tht) xcnt_s[lane * MT + t] += 1;
            }
        }
    }
}

// Exact (sequential-dot) score of every test interaction: the thresholds of the sweep.
__global__ void test_scores_kernel(RanksArgs a)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.test_nnz) return;
    // the row of entry t: binary search of indptr
    int lo = 0, hi = a.test.rows;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)a.test.indptr[mid] <= t) lo = mid; else hi = mid;
    }
    const float *u = a.user_rep + (size_t)lo * a.rs;
    if (a.item_rows_rm == nullptr) {
        a.test_scores[t] = dense_dot(u, a.item_rep, (size_t)a.test.cols, a.test.indices[t], a.d);
        return;
    }
    // the same operations in the same order (PYX:320-334) from two contiguous rows (rs is a multiple of four floats)
    const float *v = a.item_rows_rm + (size_t)a.test.indices[t] * a.rs;
    const int d = a.d;
    float acc = __fadd_rn(u[d], v[d]);
    int c = 0;
    for (; c + 4 <= d; c += 4) {
        const float4 x = *(const float4 *)(u + c), y = *(const float4 *)(v + c);
        acc = __fadd_rn(acc, __fmul_rn(x.x, y.x));
        acc = __fadd_rn(acc, __fmul_rn(x.y, y.y));
        acc = __fadd_rn(acc, __fmul_rn(x.z, y.z));
        acc = __fadd_rn(acc, __fmul_rn(x.w, y.w));
    }
    for (; c < d; ++c) acc = __fadd_rn(acc, __fmul_rn(u[c], v[c]));
    a.test_scores[t] = acc;
}

template <int KSTEPS>
__global__ __launch_bounds__(64, KSTEPS > 32 ? 1 : 2) void ranks_mfma2_kernel(RanksArgs a)
{
    constexpr int MT = 16;
    __shared__ float thr_s[WAVE * MT];  // thresholds, ids and exact in-band hits, by lane
    __shared__ int tid_s[WAVE * MT];
    __shared__ int xcnt_s[WAVE * MT];
    __shared__ float sc_s[16 * WAVE];   // slow path: the tile's scores [r][lane]
    __shared__ float ej_s[32], nj_s[32];  // slow path: eps terms of the tile's items
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
    const int d = a.d, I = a.test.cols;
    const float *vT = a.item_rep;  // [rs][I] component-major, row d = item bias
    const float kappa = 4.0f * (float)(d + 2) * 5.9604645e-8f;
    const float INF = __int_as_float(0x7f800000);
    // a work item = (32-user tile, pass of 16 test items): a heavy user's many passes run on different
    // wavefronts instead of one after the other
    for (int w = blockIdx.x; w < a.n_work; w += gridDim.x) {
        const int tile = a.work[2 * w], p0 = a.work[2 * w + 1];
        const int ui = tile * 32 + col;
        const bool uok = ui < a.n_ulist;
        const int user = uok ? a.ulist[ui] : 0;
        const float *urow = a.user_rep + (size_t)user * a.rs;
        float ub[KSTEPS];
        float n2 = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int k = 2 * kk + half;
            ub[kk] = (uok && k < d) ? urow[k] : 0.0f;
            n2 += ub[kk] * ub[kk];
        }
        n2 += __shfl_xor(n2, 32, WAVE);
        const float bu = uok ? urow[d] : 0.0f;
        const float nu = sqrtf(n2) * 1.0000005f, eu = kappa * fabsf(bu);
        const int t_lo = uok ? a.test.indptr[user] : 0, t_hi = uok ? a.test.indptr[user + 1] : 0;
        {
            // ---- this pass's test items: ids and exact scores (PYX:1278-1293)
            const int m = max(0, min(MT, (t_hi - t_lo) - p0));
#pragma unroll 1
            for (int t = 0; t < MT; ++t) {
                // exact scores of the test items (PYX:1278-1293) come from test_scores_kernel
                const bool on = t < m;
                thr_s[lane * MT + t] = on ? a.test_scores[t_lo + p0 + t] : INF;
                tid_s[lane * MT + t] = on ? a.test.indices[t_lo + p0 + t] : -1;
                xcnt_s[lane * MT + t] = 0;
            }
            float th[MT];
            int cl[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                th[t] = thr_s[lane * MT + t];
                cl[t] = 0;
            }
            // train row cursor of the user, one entry ahead
            int tp = 0, tend = 0, next = 0x7fffffff, next2 = 0x7fffffff;
            if (uok) {
                tp = a.train.indptr[user];
                tend = a.train.indptr[user + 1];
                if (tp < tend) next = a.train.indices[tp];
                if (tp + 1 < tend) next2 = a.train.indices[tp + 1];
            }
            // A operand of a tile: V[item j0 + col][2 kk + half], walked down the component-major table
            // (the table has >= 2 KSTEPS rows, zero beyond d); items past the end re-read the last one
            float av[KSTEPS], bj = 0.0f, ej = 0.0f, njk = 0.0f;
            auto load_tile = [&](int j0) {
                // scalar row pointer + one 32-bit lane offset for all rows (global_load saddr form)
                const unsigned jc = (unsigned)min(j0 + col, I - 1);
                const unsigned voff = jc + (unsigned)half * (unsigned)I;
                const float *rowp = vT;
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    av[kk] = rowp[voff];
                    rowp += 2 * (size_t)I;
                }
                bj = (vT + (size_t)d * I)[jc];
                ej = a.item_eps[jc];
                njk = (a.item_eps + (size_t)I)[jc];
            };
            int first = 0;
            asm volatile("" : "+s"(first));  // opaque: the first tile's addresses are not worth keeping in registers
            load_tile(first);
            for (int j0 = 0; j0 < I; j0 += 32) {
                const float bjt = bj;
                const float ejt = ej, njt = njk;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int iA = (r & 3) + 8 * (r >> 2);
                    const float b_lo = read_lanef(bjt, iA), b_hi = read_lanef(bjt, iA + 4);
                    acc[r] = __fadd_rn(bu, half ? b_hi : b_lo);
                }
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], ub[kk], acc, 0, 0, 0);
                // the MFMAs have read this tile's operands: request the next tile's now, they
                // travel while the scores are compared
                if (j0 + 32 < I) load_tile(j0 + 32);
                // train positives inside [j0, j0 + 32) (PYX:1303-1304)
                unsigned tmask = 0u;
                while (next < j0 + 32) {
                    tmask |= 1u << (next - j0);
                    ++tp;
                    next = next2;
                    next2 = tp + 1 < tend ? a.train.indices[tp + 1] : 0x7fffffff;
                }
                unsigned bandmask = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int iA = (r & 3) + 8 * (r >> 2), i = iA + 4 * half;
                    // one score at a time: nothing of score r starts before the counts of score r - 1
                    // are done (otherwise all 16 x 16 differences are computed up front and spill)
                    float ejr = ejt, njr = njt, scr = acc[r];
                    asm volatile("" : "+v"(ejr), "+v"(njr), "+v"(scr)
                                 : "v"(cl[0]), "v"(cl[1]), "v"(cl[2]), "v"(cl[3]), "v"(cl[4]), "v"(cl[5]), "v"(cl[6]), "v"(cl[7]),
                                   "v"(cl[8]), "v"(cl[9]), "v"(cl[10]), "v"(cl[11]), "v"(cl[12]), "v"(cl[13]), "v"(cl[14]),
                                   "v"(cl[15]), "v"(bandmask));
                    const float e_lo = read_lanef(ejr, iA), e_hi = read_lanef(ejr, iA + 4);
                    const float n_lo = read_lanef(njr, iA), n_hi = read_lanef(njr, iA + 4);
                    const float eps = __fmaf_rn(nu, half ? n_hi : n_lo, eu + (half ? e_hi : e_lo));
                    const bool valid = (j0 + i < I) && !((tmask >> i) & 1u);
                    // an item that does not count scores -inf: above no threshold, near none
                    const float sc = valid ? scr : -INF;
                    sc_s[r * WAVE + lane] = sc;  // for the rare exact re-check below (same lane reads it)
                    float dmin = INF;
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const float df = sc - th[t];  // ONE rounded difference decides "above" and "near"
                        cl[t] += (df > eps) ? 1 : 0;
                        dmin = fminf(dmin, fabsf(df));
                    }
                    bandmask |= (dmin <= eps) ? (1u << r) : 0u;
                }
                if (__ballot(bandmask != 0u) != 0ull) {
                    // some score lies within eps of a threshold: re-decide those pairs with the
                    // reference's sequential dot (PYX:1317-1319).  Rolled loops over LDS copies.
                    if (half == 0) {
                        ej_s[col] = ejt;
                        nj_s[col] = njt;
                    }
                    wave_sync();
                    band_recheck(bandmask, lane, j0, m, nu, eu, urow, vT, I, d, sc_s, ej_s, nj_s, thr_s, tid_s, xcnt_s);
                    wave_sync();
                }
            }
            // both halves of a user's column counted different items
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                int c = cl[t] + xcnt_s[lane * MT + t];
                c += __shfl_xor(c, 32, WAVE);
                if (half == 0 && t < m) a.ranks[t_lo + p0 + t] += (float)c;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// predict_ranks, third formulation (the default): the lane-per-user MFMA sweep of ranks_mfma2_kernel
// with the per-(score, threshold) compare chain replaced by a SEARCH.  ranks_mfma2_kernel spends ~100
// VALU instructions per score on 16 thresholds (8.6 G VALU instructions against 0.17 G MFMAs per ML-20M
// sweep: it is bound by VALU issue, profiles/r04_visit_m.txt).  Here
//   * a pass holds up to 31 test items of a user, sorted ascending in LDS ([rank][user column], +inf
//     beyond; the two half-wave lanes of a user share the column);
//   * per score: x_lo = s - eps, x_hi = s + eps, then log2(32) = 5 (4, 3 for light passes) search steps give
//     k = #{thresholds < x_lo}: the first two against three thresholds held in registers, the others a
//     ds_read_b32 (the row offset is an instruction immediate) + subtract + shift + v_and_or_b32 each -- the
//     comparison is the sign of (threshold - x_lo).  One more read returns the smallest threshold >= x_lo (in
//     the rounding band iff it is <= x_hi), one ds_add_u32 counts the score in bucket k.  The count of the
//     threshold of rank r is the sum of the buckets above r, taken once per work item;
//   * eps of a score is the bound of ITS octet of items (max over 8 items, x (1 + 1 / (2 (d + 2))) for the roundings
//     of s -/+ eps): one instruction per score instead of eight; the band is re-decided with the reference's
//     sequential dot (PYX:1317-1319), computed by the whole wavefront, so the ranks are the reference's integers;
//   * a work item = (32-user tile, pass, segment of the item table): partial counts are integers, published with
//     float atomics -- exact, in any order, while a rank stays below 2^24 (catalogues of up to 16.7 M items; beyond
//     that the sums round like the reference's own float32 `rank += 1.0`, PYX:1321, which stops counting at 2^24,
//     but their rounding then depends on the order of the atomics); segments (< 2^26 items each) are dispatched
//     segment-major so that the wavefronts resident at one time walk the same ~1 MB of the table.
// LDS 12.3 KB and <= 168 VGPRs per wavefront: three wavefronts per SIMD (d <= 64).
// max over the aligned group of eight lanes, in all eight
__device__ __forceinline__ float octet_max(float v)
{
    int x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false)));  // quad_perm [1,0,3,2]
    x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false)));  // [2,3,0,1]
    return fmaxf(v, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F)));  // lane ^ 4
}

constexpr int R3_ROWS = 32;            // rows of the sorted-threshold / bucket arrays: up to 31 thresholds per pass
constexpr unsigned R3_CNT = 0x3ffffffu;  // hist_s: bucket count in the low 26 bits, the test slot of rank r above

// The 16 scores of a lane: bucket counts, and the mask of scores with a threshold in their rounding band
// (score r -> bit 15 - r).  Items that do not count arrive as -inf (bucket 0, near nothing).  ejq / njq: item-side
// bound terms, octet maxima, by item lane.  cb: the lane's byte offset in a row of srt_s / hist_s (4 col).
// t0, t1a, t1b: the thresholds the first two steps of every search compare with (rows R/2 - 1, R/4 - 1, 3R/4 - 1),
// held in registers.  The scores go through the further steps W at a time.
#ifndef LFM_RANKS3_GROUP
#define LFM_RANKS3_GROUP 8
#endif
// LFM_RANKS3_LOCKSTEP 1: scheduling barriers keep the W scores of a group in step through the search (W reads in flight,
// one wait).  Measured SLOWER than the compiler's own interleaving of shorter chains (visit r4q: 9.0 against 7.5 ms
// without the exact re-checks), so it is off.
#ifndef LFM_RANKS3_LOCKSTEP
#define LFM_RANKS3_LOCKSTEP 0
#endif
// LFM_R3X: timing experiments of tools/visit.sh r4q (WRONG ranks): 1 no bucket atomics, 2 no LDS search steps,
// 3 no matrix products; all without the exact re-checks
#ifndef LFM_R3X
#define LFM_R3X 0
#endif
// LFM_RANKS3_KAPPA_MULT: timing experiment (ranks stay exact: a wider band is still a band) -- what the sweep costs with the
// rounding band the bf16-split products would need
// LFM_RANKS3_DEFER 0: scores inside a band are re-decided on the spot (band_recheck3: the path of rounds 4-5, and the
// cross-check of the deferred queue)
#ifndef LFM_RANKS3_DEFER
#define LFM_RANKS3_DEFER 1
#endif
#ifndef LFM_RANKS3_KAPPA_MULT
#define LFM_RANKS3_KAPPA_MULT 1.0f
#endif
// (a & mask) | c and (m & a) | (~m & b) as the single instructions they are (the compiler's own forms of the
// expressions below are compare + select chains)
__device__ __forceinline__ unsigned and_or(unsigned a, unsigned mask, unsigned c)
{
    unsigned d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(mask), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned bit_select(unsigned m, unsigned a, unsigned b)
{
    unsigned d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b));
    return d;
}

// "x > v" is taken from the SIGN of v - x (equal -> +0, v = +inf or x = -inf -> +inf: "not above", as the compare
// gives; no NaN reaches here): shifts and bit merges instead of compare + select, whose VCC round trip costs a
// wait state per use on this chip.
template <int L>
__device__ __forceinline__ unsigned count_tile(const f32x16 &acc, float ejq, float njq, float nu2, float eu2,
                                               unsigned cb, float t0, float t1a, float t1b, const char *srt_b, char *hist_b)
{
    constexpr int W = LFM_RANKS3_GROUP;
    constexpr int SH0 = L + 6;  // log2 of the byte offset of half the rows (2^L rows of 128 B)
    unsigned nband = 0u;        // complement of the band mask
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += W) {
        float xl[W], xh[W];
        unsigned ib[W];
#pragma unroll
        for (int gq = 0; gq < W / 4; ++gq) {
            const int g = g0 / 4 + gq;  // rows 8 g + 4 half + 0..3 of the tile: half of an octet of items, whose bound it takes
            const float eps = __fmaf_rn(nu2, read_lanef(njq, 8 * g), __fadd_rn(eu2, read_lanef(ejq, 8 * g)));
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int q = 4 * gq + qq, r = g0 + q;
                xl[q] = __fsub_rn(acc[r], eps);
                xh[q] = __fadd_rn(acc[r], eps);
                const unsigned m0 = (unsigned)(__float_as_int(__fsub_rn(t0, xl[q])) >> 31);  // all ones: above t0
                ib[q] = and_or(m0, 1u << SH0, cb);
                const float v1 = __uint_as_float(bit_select(m0, __float_as_uint(t1b), __float_as_uint(t1a)));
                ib[q] = and_or(__float_as_uint(__fsub_rn(v1, xl[q])) >> (31 - (SH0 - 1)), 1u << (SH0 - 1), ib[q]);
            }
        }
#pragma unroll
        for (int lv = 2; lv < L; ++lv) {
            const int sh = SH0 - lv;  // this step adds 2^sh bytes
            float v[W];
            if (LFM_R3X == 2) continue;
            if (LFM_RANKS3_LOCKSTEP) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < W; ++q) v[q] = *(const float *)(srt_b + ib[q] + ((1 << sh) - 128));
            if (LFM_RANKS3_LOCKSTEP) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < W; ++q) ib[q] = and_or(__float_as_uint(__fsub_rn(v[q], xl[q])) >> (31 - sh), 1u << sh, ib[q]);
        }
        float succ[W];
        if (LFM_RANKS3_LOCKSTEP) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < W; ++q) succ[q] = *(const float *)(srt_b + ib[q]);  // rows >= the pass's thresholds hold +inf
#pragma unroll
        for (int q = 0; q < W; ++q)
            if (LFM_R3X != 1) atomicAdd((unsigned *)(hist_b + ib[q]), 1u);
        if (LFM_RANKS3_LOCKSTEP) __builtin_amdgcn_sched_barrier(0);
        // in the band iff succ <= x_hi, i.e. x_hi - succ is not negative
#pragma unroll
        for (int q = 0; q < W; ++q) nband = __builtin_amdgcn_alignbit(nband, __float_as_uint(__fsub_rn(xh[q], succ[q])), 31);
    }
    return ~nband & 0xffffu;
}

// Scores with a threshold inside [x_lo, x_hi], re-decided with the reference's sequential dot (PYX:1317-1319, 320-334).
// Every lane takes its next flagged score and looks whether one of its thresholds really needs the exact score; the
// exact scores are then computed by the WHOLE wavefront, one pending lane at a time: lane c multiplies component c
// (one load round trip for all d products), and the sum runs over them in the reference's order through lane reads --
// the same float32 operations as the reference's loop, without its d dependent memory round trips (one lane walking
// the d components alone cost ~5 000 cycles per score, a third of the sweep's time at ML-20M: visit r4q).
__device__ __forceinline__ void band_recheck3(unsigned band, int lane, int j0, int m, float nu2, float eu2, int user,
                                           const float *user_rep, int rs, const float *vT, const float *v_rows, int I, int d,
                                           const float *sc_s, const float *ej_s, const float *nj_s, const float *srt_s,
                                           const unsigned *hist_s, const int32_t *tids, float *ranks)
{
    const int half = lane >> 5, col = lane & 31;
    asm volatile("" : "+s"(I), "+s"(d), "+s"(vT), "+s"(v_rows));  // rare path: nothing of its address arithmetic is hoisted into the sweep
    while (__ballot(band != 0u) != 0ull) {
        bool pend = false;
        int item = 0, k = 0;
        float xl = 0.0f, xh = 0.0f;
        if (band) {
            const int bit = __ffs((int)band) - 1;
            band &= band - 1u;
            const int r = 15 - bit;
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            item = j0 + i;
            const float sc = sc_s[r * WAVE + lane];
            const float eps = __fmaf_rn(nu2, nj_s[i & 24], __fadd_rn(eu2, ej_s[i & 24]));  // the octet's bound, as the sweep took it
            xl = __fsub_rn(sc, eps);
            xh = __fadd_rn(sc, eps);
            // the thresholds in [x_lo, x_hi] are the ranks k, k + 1, ... (k = thresholds below x_lo; rows >= m hold +inf)
            for (int step = R3_ROWS / 2; step >= 1; step >>= 1)
                if (xl > srt_s[(k + step - 1) * 32 + col]) k += step;
            for (int t = k; t < m && srt_s[t * 32 + col] <= xh; ++t) pend = pend || item != tids[hist_s[t * 32 + col] >> 26];
        }
        unsigned long long todo = __ballot(pend);
        while (todo) {
            const int L = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const int usr = read_lane(user, L), itm = read_lane(item, L);
            const float *ur = user_rep + (size_t)usr * rs;
            float p0 = 0.0f, p1 = 0.0f, ex;
            if (v_rows) {  // the item's row-major representation: ONE coalesced request instead of a line per component
                const float *vr = v_rows + (size_t)itm * rs;
                if (lane < d) p0 = __fmul_rn(ur[lane], vr[lane]);
                if (lane + WAVE < d) p1 = __fmul_rn(ur[lane + WAVE], vr[lane + WAVE]);
                ex = __fadd_rn(ur[d], vr[d]);
            } else {
                if (lane < d) p0 = __fmul_rn(ur[lane], vT[(size_t)lane * I + itm]);
                if (lane + WAVE < d) p1 = __fmul_rn(ur[lane + WAVE], vT[(size_t)(lane + WAVE) * I + itm]);
                ex = __fadd_rn(ur[d], vT[(size_t)d * I + itm]);
            }
            const int d0 = min(d, WAVE);
            for (int c = 0; c < d0; ++c) ex = __fadd_rn(ex, read_lanef(p0, c));
            for (int c = WAVE; c < d; ++c) ex = __fadd_rn(ex, read_lanef(p1, c - WAVE));
            if (lane == L) {
                for (int t = k; t < m && srt_s[t * 32 + col] <= xh; ++t) {
                    const int slot = (int)(hist_s[t * 32 + col] >> 26);
                    if (item != tids[slot] && ex >= srt_s[t * 32 + col]) atomicAdd(&ranks[slot], 1.0f);
                }
            }
        }
    }
}

// Deferred re-checks (round 6).  A score with a threshold inside its band is QUEUED -- (item, user column), x_lo, x_hi, three
// words in the wavefront's sorting scratch -- instead of being re-decided on the spot, and the queue is worked off when it
// is full and at the end of the work item: one entry per LANE, every lane running the reference's sequential dot
// (PYX:320-334: the operations of test_scores_kernel) of ITS pair from the two row-major representations.  The search, the
// own-item test and the exact comparison are band_recheck3's; only the order of the (commuting, integer) additions to the
// ranks changes.  A re-check costs a few instructions of a lane instead of ~300 of the whole wavefront (the band
// measured 1.1 ms per unit of its width before: profiles/r06_ranks_bf16_experiment.txt).
constexpr int R3_QCAP = 320;  // entries of 12 bytes in the 4 KB of sc_s (a tile appends at most 64 per step)

__device__ __forceinline__ void ranks3_flush(int qn, int lane, const unsigned *q_s, const float *srt_s, const unsigned *hist_s,
                                             const RanksArgs &a, int tile, int p0, int d)
{
    for (int e = lane; e < qn; e += WAVE) {
        const unsigned packed = q_s[3 * e];
        const float xl = __uint_as_float(q_s[3 * e + 1]), xh = __uint_as_float(q_s[3 * e + 2]);
        const int item = (int)(packed & 0x3ffffffu), c = (int)(packed >> 26);
        const int user = a.ulist[tile * 32 + c];  // (a queued column always holds a user)
        const int t_lo = a.test.indptr[user], t_hi = a.test.indptr[user + 1];
        const int m = max(0, min(R3_ROWS - 1, (t_hi - t_lo) - p0));
        const int32_t *tids = a.test.indices + t_lo + p0;
        float *ranks = a.ranks + t_lo + p0;
        // the thresholds in [x_lo, x_hi] are the ranks k, k + 1, ... (k = thresholds below x_lo; rows >= m hold +inf)
        int k = 0;
        for (int step = R3_ROWS / 2; step >= 1; step >>= 1)
            if (xl > srt_s[(k + step - 1) * 32 + c]) k += step;
        bool pend = false;
        for (int t = k; t < m && srt_s[t * 32 + c] <= xh; ++t) pend = pend || item != tids[hist_s[t * 32 + c] >> 26];
        if (!pend) continue;
        const float *u = a.user_rep + (size_t)user * a.rs, *v = a.item_rows_rm + (size_t)item * a.rs;
        float ex = __fadd_rn(u[d], v[d]);
        int cc = 0;
#pragma unroll 2
        for (; cc + 4 <= d; cc += 4) {
            const float4 x = *(const float4 *)(u + cc), y = *(const float4 *)(v + cc);
            ex = __fadd_rn(ex, __fmul_rn(x.x, y.x));
            ex = __fadd_rn(ex, __fmul_rn(x.y, y.y));
            ex = __fadd_rn(ex, __fmul_rn(x.z, y.z));
            ex = __fadd_rn(ex, __fmul_rn(x.w, y.w));
        }
        for (; cc < d; ++cc) ex = __fadd_rn(ex, __fmul_rn(u[cc], v[cc]));
        for (int t = k; t < m && srt_s[t * 32 + c] <= xh; ++t) {
            const int slot = (int)(hist_s[t * 32 + c] >> 26);
            if (item != tids[slot] && ex >= srt_s[t * 32 + c]) atomicAdd(&ranks[slot], 1.0f);
        }
    }
}

// BF (round 6, the default): the products run on the bf16 matrix pipe, which -- unlike the fp32 MFMA, which executes on the
// VALU's own lanes (SQ_VALU_MFMA_COEXEC_CYCLES = 0) -- CO-EXECUTES with the search.  Every float32 operand is split into two
// bf16 pieces, x = h + l + r with h = bf16(x), l = bf16(x - h) (round to nearest: |x - h| <= 2^-8 |x|, |r| <= 2^-16 |x|), and a
// step of 16 components is THREE v_mfma_f32_32x32x16_bf16 (h h, h l, l h) on operands that occupy the registers the fp32
// operands did: 3 d / 16 + 1 instructions of 8 passes per tile instead of d / 2 + 1 of 16.  The item pieces come from a
// table built once per call (item_bf_kernel: [step][piece][item][half] x 16 bytes, one buffer_load_dwordx4 per operand), the
// user pieces are split when a work item starts; the two biases stay ONE fp32 step, (b_j, 1) . (1, b_u), as in the fp32 sweep
// (two registers; its 16 passes are 8 % of a tile's matrix time).  What this costs is a wider rounding band (ranks_bf_kappa
// below: ~5 x the fp32 sweep's at d = 64), which the deferred re-checks make cheap (ranks3_flush).
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// round to nearest even (finite values)
__device__ __forceinline__ unsigned bf16_rn(float f)
{
    const unsigned x = __float_as_uint(f);
    return (x + 0x7fffu + ((x >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void bf16_split(float f, unsigned &h, unsigned &l)
{
    h = bf16_rn(f);
    l = bf16_rn(__fsub_rn(f, __uint_as_float(h << 16)));  // (the difference is exact)
}

// Rounding band of the bf16-split sweep, as multiples of T = |b_u| + |b_j| + |u| |v_j| (kT) and of |u| |v_j| alone (kS):
//   reference, PYX:320-334: d + 2 roundings of at most 2^-24 T each;
//   the sweep: 16 (3 NS + 1) accumulations (NS = steps of 16 components) whose internal rounding the ISA does not specify --
//   taken as 2^-23 each (faithful), on partial sums of at most (1 + 2^-8)^2 (1 + 2^-7) T;
//   the split: sum_k |l_u l_v + r_u v + u r_v| <= 3.03 x 2^-16 sum_k |u_k v_k| <= 3.03 x 2^-16 |u| |v_j|; (the bias step is exact fp32 arithmetic:
//   counted among the accumulations);
// each with a quarter on top.  (d = 64: kT = 3.7e-5, kS = 4.7e-5 against the fp32 sweep's 4 (d + 2) 2^-24 = 1.6e-5.)
__host__ __device__ __forceinline__ float ranks_bf_kappa_t(int d, int ns)
{
    return 1.25f * 5.9604645e-8f * ((float)(d + 2) + 1.02f * 2.0f * 16.0f * (float)(3 * ns + 1) + 1.0f);
}
__host__ __device__ __forceinline__ float ranks_bf_kappa_s() { return 1.25f * 3.03f * 1.52587890625e-5f; }
__host__ __device__ __forceinline__ int ranks_bf_steps(int d) { return d <= 32 ? 2 : (d <= 64 ? 4 : 8); }

// [step][piece][item][half] x 16 bytes
__global__ void item_bf_kernel(const float *vT, int n_items, int d, int ns, u32x4 *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_step = (int64_t)n_items * 2;
    if (t >= per_step * ns) return;
    const int st = (int)(t / per_step), j = (int)((t % per_step) >> 1), half = (int)(t & 1);
    unsigned h[8], l[8];
    for (int i = 0; i < 8; ++i) {
        const int k = 16 * st + 8 * half + i;
        bf16_split(k < d ? vT[(size_t)k * n_items + j] : 0.0f, h[i], l[i]);
    }
    u32x4 H = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    u32x4 L = {l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    out[((size_t)(2 * st) * n_items + j) * 2 + half] = H;
    out[((size_t)(2 * st + 1) * n_items + j) * 2 + half] = L;
}

// the item-side terms of the bf16-split sweep's band: kT |b_j| and (kT + kS) |v_j|
__global__ void item_eps_bf_kernel(const float *vT, int n_items, int d, float kt, float ks, float *out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_items) return;
    float n2 = 0.0f;
    for (int k = 0; k < d; ++k) {
        const float v = vT[(size_t)k * n_items + j];
        n2 += v * v;
    }
    out[j] = kt * fabsf(vT[(size_t)d * n_items + j]);
    out[(size_t)n_items + j] = (kt + ks) * (sqrtf(n2) * 1.0000005f);
}

template <int KSTEPS, bool BF = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KSTEPS > 32 ? 2 : 3, KSTEPS > 32 ? 2 : 3)))
void ranks_mfma3_kernel(RanksArgs a)
{
    constexpr int ROWS = R3_ROWS, MT = ROWS - 1;
    constexpr int NS = KSTEPS / 8;  // BF: steps of 16 components
    __shared__ float srt_s[ROWS * 32];      // [rank][user column] thresholds of the pass, ascending, +inf beyond
    __shared__ unsigned hist_s[ROWS * 32];  // [bucket][user column] scores with `bucket` thresholds below | slot of the rank << 26
    __shared__ float sc_s[16 * WAVE];       // sorting scratch [slot][user column]; slow path: the tile's scores [r][lane]
    __shared__ float ej_s[32], nj_s[32];    // slow path: bound terms of the tile's items
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
    const int d = a.d, I = a.test.cols;
    const float *vT = a.item_rep;  // [rs][I] component-major, row d = item bias
    const float kappa = BF ? ranks_bf_kappa_t(d, NS) : LFM_RANKS3_KAPPA_MULT * 4.0f * (float)(d + 2) * 5.9604645e-8f;
    const float INF = __int_as_float(0x7f800000);
    // s - eps and s + eps are rounded: each by at most u |s| (1 + ...) <= eps / (4 (d + 2)), as eps >= kappa |s|; twice that
    // is added to eps so that "x_lo > threshold" still implies "s - threshold > eps"
    const float margin = 1.0f + 1.0f / (float)(2 * (d + 2));
    // the component-major item table and the bound terms as buffers (launch_ranks_mfma3 checks they are < 2 GB)
    const unsigned row2 = 8u * (unsigned)I;  // bytes of two table rows
    const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc((void *)vT, 0, (int)(4u * (unsigned)I * (unsigned)a.item_rows), 0x00020000);
    const __amdgpu_buffer_rsrc_t esrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.item_eps, 0, (int)row2, 0x00020000);
    // BF: the table of bf16 pieces, 32 I bytes per (step, piece)
    const unsigned bstep = 32u * (unsigned)I;
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(BF ? (const void *)a.item_bf : (const void *)vT), 0,
                                                                          BF ? (int)(bstep * (unsigned)(2 * NS)) : 0, 0x00020000);
    const unsigned cb = 4u * (unsigned)col;
    const char *srt_b = (const char *)srt_s;
    char *hist_b = (char *)hist_s;
    for (int w = blockIdx.x; w < a.n_work; w += gridDim.x) {
        const int tile = a.work[4 * w], p0 = a.work[4 * w + 1], jb = a.work[4 * w + 2], je = a.work[4 * w + 3];
        const int ui = tile * 32 + col;
        const bool uok = ui < a.n_ulist;
        const int user = uok ? a.ulist[ui] : 0;
        const float *urow = a.user_rep + (size_t)user * a.rs;
        float ub[KSTEPS];   // fp32: component 2 kk + half; BF: the pieces, ub[8 st .. + 3] = h and ub[8 st + 4 .. + 7] = l of
                            // components 16 st + 8 half + 0..7 (two bf16 per register)
        float n2 = 0.0f;
        if constexpr (BF) {
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                unsigned h[8], l[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = 16 * st + 8 * half + i;
                    const float x = (uok && k < d) ? urow[k] : 0.0f;
                    n2 += x * x;
                    bf16_split(x, h[i], l[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ub[8 * st + i] = __uint_as_float(h[2 * i] | (h[2 * i + 1] << 16));
                    ub[8 * st + 4 + i] = __uint_as_float(l[2 * i] | (l[2 * i + 1] << 16));
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int k = 2 * kk + half;
                ub[kk] = (uok && k < d) ? urow[k] : 0.0f;
                n2 += ub[kk] * ub[kk];
            }
        }
        n2 += __shfl_xor(n2, 32, WAVE);
        const float bu = uok ? urow[d] : 0.0f;
        // the user-side terms of the bound, with the margin for the roundings of s - eps and s + eps
        const float nu2 = sqrtf(n2) * 1.0000005f * margin, eu2 = kappa * fabsf(bu) * margin;
        const int t_lo = uok ? a.test.indptr[user] : 0, t_hi = uok ? a.test.indptr[user + 1] : 0;
        const int m = max(0, min(MT, (t_hi - t_lo) - p0));
        const int32_t *tids = a.test.indices + t_lo + p0;
        float *ranks = a.ranks + t_lo + p0;
        // ---- this pass's thresholds (exact scores of the test items, PYX:1278-1293, from test_scores_kernel),
        // sorted by (value, slot): rank = number of smaller ones, each half-wave lane places every other slot
        wave_sync();
#pragma unroll 1
        for (int t = half; t < ROWS; t += 2) {
            sc_s[t * 32 + col] = t < m ? a.test_scores[t_lo + p0 + t] : INF;
            srt_s[t * 32 + col] = INF;
            hist_s[t * 32 + col] = 0u;
        }
        wave_sync();
#pragma unroll 1
        for (int t = half; t < m; t += 2) {
            const float v = sc_s[t * 32 + col];
            int rk = 0;
#pragma unroll 8
            for (int t2 = 0; t2 < ROWS; ++t2) {  // all rows (+inf beyond the lane's m: never smaller, never equal): eight reads in flight
                const float o = sc_s[t2 * 32 + col];
                rk += (o < v || (o == v && t2 < t)) ? 1 : 0;
            }
            srt_s[rk * 32 + col] = v;
            hist_s[rk * 32 + col] = (unsigned)t << 26;
        }
        wave_sync();
        int qn = 0;  // queued re-checks of this work item (sc_s is free between the sort above and the next work item)
        unsigned *q_s = (unsigned *)sc_s;
        const bool deferred = BF || (a.item_rows_rm != nullptr && LFM_RANKS3_DEFER);  // (BF: the session hands over both tables)
        const int levels = __ballot(m > 15) != 0ull ? 5 : (__ballot(m > 7) != 0ull ? 4 : 3);
        // the thresholds of the first two steps of every search of this pass
        const int rows = 1 << levels;
        const float t0 = srt_s[(rows / 2 - 1) * 32 + col], t1a = srt_s[(rows / 4 - 1) * 32 + col],
                    t1b = srt_s[(3 * rows / 4 - 1) * 32 + col];
        // train row cursor of the user at the segment's first item, one entry ahead
        int tp = 0, tend = 0, next = 0x7fffffff, next2 = 0x7fffffff;
        if (uok) {
            tp = a.train.indptr[user];
            tend = a.train.indptr[user + 1];
            int lo = tp, hi = tend;  // first entry >= jb
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (a.train.indices[mid] < jb) lo = mid + 1; else hi = mid;
            }
            tp = lo;
            if (tp < tend) next = a.train.indices[tp];
            if (tp + 1 < tend) next2 = a.train.indices[tp + 1];
        }
        // A operand of a tile: V[item j0 + col][2 kk + half], walked down the component-major table through ONE
        // buffer descriptor: a 32-bit lane offset, the row as the scalar offset (no per-load address arithmetic)
        float av[KSTEPS], bj = 0.0f, ej = 0.0f, njk = 0.0f;
        auto load_tile = [&](int j0) {
            const unsigned jc = (unsigned)min(j0 + col, I - 1);
            const unsigned voff = 4u * (jc + (unsigned)half * (unsigned)I);
            if constexpr (BF) {
                const unsigned boff = 16u * (2u * jc + (unsigned)half);
#pragma unroll
                for (int q = 0; q < 2 * NS; ++q) {  // (step q / 2, piece q % 2): registers av[4 q .. 4 q + 3]
                    const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(bsrc, boff, (unsigned)q * bstep, 0);
                    av[4 * q] = __uint_as_float(x.x);
                    av[4 * q + 1] = __uint_as_float(x.y);
                    av[4 * q + 2] = __uint_as_float(x.z);
                    av[4 * q + 3] = __uint_as_float(x.w);
                }
                bj = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, 4u * jc, (unsigned)d * (row2 >> 1), 0));
            } else {
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk)
                    av[kk] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, voff, (unsigned)kk * row2, 0));
                bj = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, 4u * jc, (unsigned)d * (row2 >> 1), 0));
            }
            ej = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(esrc, 4u * jc, 0, 0));
            njk = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(esrc, 4u * jc, row2 >> 1, 0));
        };
        load_tile(jb);
        // the biases as one more step of the product: (b_j, 1) . (1, b_u)
        const float ub_x = half ? bu : 1.0f;
        for (int j0 = jb; j0 < je; j0 += 32) {
            const float av_x = half ? 1.0f : bj;
            const float ejq = octet_max(ej) * margin, njq = octet_max(njk);
            // items that do not count: train positives inside [j0, j0 + 32) (PYX:1303-1304), rows past the table
            unsigned tmask = j0 + 32 > I ? ~0u << (I - j0) : 0u;
            while (next < j0 + 32) {
                tmask |= 1u << (next - j0);
                ++tp;
                next = next2;
                next2 = tp + 1 < tend ? a.train.indices[tp + 1] : 0x7fffffff;
            }
            const unsigned tm = tmask >> (4 * half);
            // they start from -inf and stay there: below every threshold, near none
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int iA = (r & 3) + 8 * (r >> 2);
                int off;  // 0 / -1 (written as the instruction: the compiler's own form is and + compare + select)
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(off) : "v"(tm), "n"(iA));
                acc[r] = __int_as_float(off & (int)0xff800000);
            }
            if constexpr (BF) {
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    const u32x4 ah = {__float_as_uint(av[8 * st]), __float_as_uint(av[8 * st + 1]), __float_as_uint(av[8 * st + 2]), __float_as_uint(av[8 * st + 3])};
                    const u32x4 al = {__float_as_uint(av[8 * st + 4]), __float_as_uint(av[8 * st + 5]), __float_as_uint(av[8 * st + 6]), __float_as_uint(av[8 * st + 7])};
                    const u32x4 bh = {__float_as_uint(ub[8 * st]), __float_as_uint(ub[8 * st + 1]), __float_as_uint(ub[8 * st + 2]), __float_as_uint(ub[8 * st + 3])};
                    const u32x4 bl = {__float_as_uint(ub[8 * st + 4]), __float_as_uint(ub[8 * st + 5]), __float_as_uint(ub[8 * st + 6]), __float_as_uint(ub[8 * st + 7])};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av_x, ub_x, acc, 0, 0, 0);  // the biases: one fp32 step, (b_j, 1) . (1, b_u)
            } else {
#if LFM_R3X == 4
            // timing experiment (WRONG ranks): the products on the bf16 matrix pipe -- two-way split operands in the SAME registers
            // (KSTEPS / 8 steps of 16 components x {hi hi, hi lo, lo hi}) + one step for the item bias, the user bias by VALU
            {
                typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
                typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = __fadd_rn(acc[r], bu);
#pragma unroll
                for (int st = 0; st < KSTEPS / 8; ++st) {
                    const f32x4 a1 = {av[8 * st], av[8 * st + 1], av[8 * st + 2], av[8 * st + 3]};
                    const f32x4 a2 = {av[8 * st + 4], av[8 * st + 5], av[8 * st + 6], av[8 * st + 7]};
                    const f32x4 b1 = {ub[8 * st], ub[8 * st + 1], ub[8 * st + 2], ub[8 * st + 3]};
                    const f32x4 b2 = {ub[8 * st + 4], ub[8 * st + 5], ub[8 * st + 6], ub[8 * st + 7]};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b2), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, b1), acc, 0, 0, 0);
                }
                const f32x4 ax = {av_x, 0.0f, 0.0f, 0.0f}, bx = {ub_x, 0.0f, 0.0f, 0.0f};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ax), __builtin_bit_cast(bf16x8, bx), acc, 0, 0, 0);
            }
#else
#pragma unroll
            for (int kk = 0; kk < (LFM_R3X == 3 ? 1 : KSTEPS); ++kk)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], ub[kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av_x, ub_x, acc, 0, 0, 0);
#endif
            }
            // the MFMAs have read this tile's operands: request the next tile's now
            if (j0 + 32 < je) load_tile(j0 + 32);
            unsigned band;
            if (levels == 5) band = count_tile<5>(acc, ejq, njq, nu2, eu2, cb, t0, t1a, t1b, srt_b, hist_b);
            else if (levels == 4) band = count_tile<4>(acc, ejq, njq, nu2, eu2, cb, t0, t1a, t1b, srt_b, hist_b);
            else band = count_tile<3>(acc, ejq, njq, nu2, eu2, cb, t0, t1a, t1b, srt_b, hist_b);
            if (LFM_R3X) band = 0u;
            if (deferred) {
                while (true) {
                    const bool has = band != 0u;
                    const unsigned long long hm = __ballot(has);
                    if (hm == 0ull) break;
                    const int nh = __popcll(hm);
                    if (qn + nh > R3_QCAP) {  // (wave-uniform)
                        wave_sync();
                        ranks3_flush(qn, lane, q_s, srt_s, hist_s, a, tile, p0, d);
                        wave_sync();
                        qn = 0;
                    }
                    const int bit = has ? __ffs((int)band) - 1 : 0;
                    band &= band - 1u;
                    const int r = 15 - bit;
                    float sc = acc[0];
#pragma unroll
                    for (int rr = 1; rr < 16; ++rr) sc = r == rr ? acc[rr] : sc;
                    // the octet's bound, as count_tile took it (items 8 g .. 8 g + 7 of the tile: g = r / 4)
                    const float eps = __fmaf_rn(nu2, __shfl(njq, 8 * (r >> 2), WAVE), __fadd_rn(eu2, __shfl(ejq, 8 * (r >> 2), WAVE)));
                    if (has) {
                        const int idx = qn + __popcll(hm & ((1ull << lane) - 1ull));
                        q_s[3 * idx] = (unsigned)(j0 + (r & 3) + 8 * (r >> 2) + 4 * half) | ((unsigned)col << 26);
                        q_s[3 * idx + 1] = __float_as_uint(__fsub_rn(sc, eps));
                        q_s[3 * idx + 2] = __float_as_uint(__fadd_rn(sc, eps));
                    }
                    qn += nh;
                }
            } else if (__ballot(band != 0u) != 0ull) {
                if (half == 0) {
                    ej_s[col] = ejq;
                    nj_s[col] = njq;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sc_s[r * WAVE + lane] = acc[r];
                wave_sync();
                band_recheck3(band, lane, j0, m, nu2, eu2, user, a.user_rep, a.rs, vT, a.item_rows_rm, I, d, sc_s, ej_s, nj_s, srt_s, hist_s, tids, ranks);
                wave_sync();
            }
        }
        if (qn) {
            wave_sync();
            ranks3_flush(qn, lane, q_s, srt_s, hist_s, a, tile, p0, d);
        }
        // counts: threshold of rank r is exceeded by the scores of every bucket above r
        wave_sync();
        if (half == 0) {
            unsigned run = 0u;
#pragma unroll 1
            for (int k = ROWS - 1; k >= 1; --k) {
                run += hist_s[k * 32 + col] & R3_CNT;
                if (k - 1 < m && run) atomicAdd(&ranks[hist_s[(k - 1) * 32 + col] >> 26], (float)run);
            }
        }
    }
}

bool ranks_mfma_supported(int d) { return d >= 1 && d <= 128; }

hipError_t launch_ranks_mfma(const RanksArgs &a, hipStream_t st, int cus)
{
    if (a.n_ulist <= 0) return hipSuccess;
    constexpr int MT = 16;
    const size_t smem = sizeof(float) * ((size_t)32 * (a.d + 1) + 3 * 32 * MT + 64);
    const int tiles = (a.n_ulist + 31) / 32;
    int per_cu = 0;
    int grid = tiles;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ranks_mfma_kernel<MT>, 64, smem) == hipSuccess && per_cu > 0)
        grid = std::min(tiles, per_cu * std::max(cus, 1));
    ranks_mfma_kernel<MT><<<grid, 64, smem, st>>>(a);
    return hipGetLastError();
}

template <int KSTEPS>
static hipError_t launch_ranks_mfma2_k(const RanksArgs &a, hipStream_t st, int cus)
{
    int per_cu = 0, grid = a.n_work;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ranks_mfma2_kernel<KSTEPS>, 64, 0) == hipSuccess && per_cu > 0)
        grid = std::min(a.n_work, per_cu * std::max(cus, 1));
    ranks_mfma2_kernel<KSTEPS><<<grid, 64, 0, st>>>(a);
    return hipGetLastError();
}

template <int KSTEPS, bool BF>
static hipError_t launch_ranks_mfma3_k(const RanksArgs &a, hipStream_t st, int cus)
{
    int per_cu = 0, grid = a.n_work;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ranks_mfma3_kernel<KSTEPS, BF>, 64, 0) == hipSuccess && per_cu > 0)
        grid = std::min(a.n_work, per_cu * std::max(cus, 1));
    ranks_mfma3_kernel<KSTEPS, BF><<<grid, 64, 0, st>>>(a);
    return hipGetLastError();
}

// wavefronts launch_ranks_mfma3 keeps resident per CU (the host sizes its work items by it)
int ranks_mfma3_waves_per_cu(int d) { return d <= 64 ? 12 : 8; }
int ranks_mfma3_pass_items() { return R3_ROWS - 1; }
bool ranks_mfma3_supported(int d, int64_t n_items, int item_rows)
{
    return ranks_mfma_supported(d) && n_items * item_rows * 4 < ((int64_t)1 << 31);
}

// a.work: [n_work][4] (32-user tile of ulist, first test item of the pass, first item, end item of the segment);
// segments start at multiples of 32 and hold < 2^26 items; a.ulist / a.item_rep / a.item_eps as for launch_ranks_mfma2
hipError_t launch_ranks_mfma3(const RanksArgs &a, hipStream_t st, int cus)
{
    if (a.n_ulist <= 0 || a.n_work <= 0) return hipSuccess;
    const float kappa = LFM_RANKS3_KAPPA_MULT * 4.0f * (float)(a.d + 2) * 5.9604645e-8f;
    test_scores_kernel<<<(int)((a.test_nnz + 255) / 256), 256, 0, st>>>(a);
    if (a.item_bf) {  // the products on the bf16 matrix pipe (split operands; the deferred re-checks need the row-major rows)
        const int ns = ranks_bf_steps(a.d);
        item_eps_bf_kernel<<<(a.test.cols + 255) / 256, 256, 0, st>>>(a.item_rep, a.test.cols, a.d, ranks_bf_kappa_t(a.d, ns), ranks_bf_kappa_s(), a.item_eps);
        const int64_t cells = (int64_t)a.test.cols * 2 * ns;
        item_bf_kernel<<<(int)((cells + 255) / 256), 256, 0, st>>>(a.item_rep, a.test.cols, a.d, ns, (u32x4 *)a.item_bf);
        if (a.d <= 32) return launch_ranks_mfma3_k<16, true>(a, st, cus);
        if (a.d <= 64) return launch_ranks_mfma3_k<32, true>(a, st, cus);
        if (a.d <= 128) return launch_ranks_mfma3_k<64, true>(a, st, cus);
        return hipErrorInvalidValue;
    }
    item_eps_kernel<<<(a.test.cols + 255) / 256, 256, 0, st>>>(a.item_rep, a.test.cols, a.d, kappa, a.item_eps);
    if (a.d <= 32) return launch_ranks_mfma3_k<16, false>(a, st, cus);
    if (a.d <= 64) return launch_ranks_mfma3_k<32, false>(a, st, cus);
    if (a.d <= 128) return launch_ranks_mfma3_k<64, false>(a, st, cus);
    return hipErrorInvalidValue;
}

// Device self-test of the bf16-split sweep's rounding band (lfm_selftest_ranks_bf16_band).  One wavefront per 32 x 32 tile of
// pseudo-random users and items (components and biases of random sign over a range of exponents; `spread` binades): the scores
// through EXACTLY the instruction sequence of ranks_mfma3_kernel<.., true> (bf16_split pieces, l h / h l / h h per step, the bias
// step, the user bias as the initial value) against the reference's sequential float32 dot (PYX:320-334), as a fraction of the
// band the sweep would take for the pair, kT (|b_u| + |b_j| + |u| |v_j|) + kS |u| |v_j| (without the margins).  out[0] = the largest
// fraction seen (float bits), out[1] = pairs beyond 1.0 -- the band's assumption about the matrix pipe's internal rounding
// holds iff that is 0.
template <int NS>
__global__ __launch_bounds__(64) void ranks_bf_band_selftest_kernel(int64_t tiles, uint32_t seed, int d, int spread, unsigned *out)
{
    __shared__ float U[32 * 132], V[32 * 132];  // [row][d + 1], bias at [d]
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
    unsigned worst = 0u, beyond = 0u;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        __syncthreads();
        for (int i = lane; i < 32 * (d + 1); i += WAVE) {
            for (int side = 0; side < 2; ++side) {
                uint32_t x = seed ^ (uint32_t)(t * 2654435761ull) ^ (uint32_t)(i * 40503u + side * 977u);
                x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
                const int e = 127 - 6 + (int)((x >> 23) % (unsigned)max(spread, 1));
                const float v = __uint_as_float((x & 0x80000000u) | ((unsigned)e << 23) | (x & 0x7fffffu));
                (side ? V : U)[(i / (d + 1)) * 132 + (i % (d + 1))] = v;
            }
        }
        __syncthreads();
        // B operand: user `col`; A operand: item `col`
        u32x4 ah[NS], al[NS], bh[NS], bl[NS];
        float nu = 0.0f, nv = 0.0f;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            unsigned h[8], l[8], hv[8], lv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 16 * st + 8 * half + i;
                const float xu = k < d ? U[col * 132 + k] : 0.0f, xv = k < d ? V[col * 132 + k] : 0.0f;
                nu += xu * xu;
                nv += xv * xv;
                bf16_split(xu, h[i], l[i]);
                bf16_split(xv, hv[i], lv[i]);
            }
            bh[st] = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
            bl[st] = u32x4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
            ah[st] = u32x4{hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16), hv[4] | (hv[5] << 16), hv[6] | (hv[7] << 16)};
            al[st] = u32x4{lv[0] | (lv[1] << 16), lv[2] | (lv[3] << 16), lv[4] | (lv[5] << 16), lv[6] | (lv[7] << 16)};
        }
        nu += __shfl_xor(nu, 32, WAVE);
        nv += __shfl_xor(nv, 32, WAVE);
        const float bu = U[col * 132 + d], bj = V[col * 132 + d];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[st]), __builtin_bit_cast(bf16x8, bh[st]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[st]), __builtin_bit_cast(bf16x8, bl[st]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[st]), __builtin_bit_cast(bf16x8, bh[st]), acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? 1.0f : bj, half ? bu : 1.0f, acc, 0, 0, 0);
        const float kt = ranks_bf_kappa_t(d, NS), ks = ranks_bf_kappa_s();
        const float unorm = sqrtf(nu);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int item = (r & 3) + 8 * (r >> 2) + 4 * half;  // row of the tile; column = this lane's user
            const float *ur = U + col * 132, *vr = V + item * 132;
            float ex = __fadd_rn(ur[d], vr[d]), vn = 0.0f;
            for (int c = 0; c < d; ++c) {
                ex = __fadd_rn(ex, __fmul_rn(ur[c], vr[c]));
                vn += vr[c] * vr[c];
            }
            const float uv = unorm * sqrtf(vn);
            const float band = kt * (fabsf(ur[d]) + fabsf(vr[d]) + uv) + ks * uv;
            const float frac = fabsf(__fsub_rn(acc[r], ex)) / band;
            worst = max(worst, __float_as_uint(frac));
            beyond += frac > 1.0f ? 1u : 0u;
        }
    }
    atomicMax(out, worst);
    if (beyond) atomicAdd(out + 1, beyond);
}

hipError_t launch_ranks_bf_band_selftest(int64_t tiles, uint32_t seed, int d, int spread, unsigned *out, hipStream_t st)
{
    const int grid = (int)std::min<int64_t>(tiles, 8192);
    if (d <= 32) ranks_bf_band_selftest_kernel<2><<<grid, 64, 0, st>>>(tiles, seed, d, spread, out);
    else if (d <= 64) ranks_bf_band_selftest_kernel<4><<<grid, 64, 0, st>>>(tiles, seed, d, spread, out);
    else if (d <= 128) ranks_bf_band_selftest_kernel<8><<<grid, 64, 0, st>>>(tiles, seed, d, spread, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// bytes of the table of bf16 pieces launch_ranks_mfma3 fills when RanksArgs::item_bf is set; 0 = outside the variant's scope
size_t ranks_mfma3_bf_bytes(int d, int64_t n_items)
{
    if (!ranks_mfma_supported(d)) return 0;
    const int64_t bytes = (int64_t)(2 * ranks_bf_steps(d)) * 32 * n_items;
    return bytes < ((int64_t)1 << 31) ? (size_t)bytes : 0;
}

int ranks_mfma2_item_rows(int d) { return std::max(d + 1, d <= 32 ? 32 : (d <= 64 ? 64 : 128)); }

// a.ulist must be ordered by the users' number of test interactions, largest first; a.item_rep has
// ranks_mfma2_item_rows(d) rows (zero beyond the bias row d), a.item_eps room for 2 * n_items floats
hipError_t launch_ranks_mfma2(const RanksArgs &a, hipStream_t st, int cus)
{
    if (a.n_ulist <= 0 || a.n_work <= 0) return hipSuccess;
    const float kappa = 4.0f * (float)(a.d + 2) * 5.9604645e-8f;
    item_eps_kernel<<<(a.test.cols + 255) / 256, 256, 0, st>>>(a.item_rep, a.test.cols, a.d, kappa, a.item_eps);
    test_scores_kernel<<<(int)((a.test_nnz + 255) / 256), 256, 0, st>>>(a);
    if (a.d <= 32) return launch_ranks_mfma2_k<16>(a, st, cus);
    if (a.d <= 64) return launch_ranks_mfma2_k<32>(a, st, cus);
    if (a.d <= 128) return launch_ranks_mfma2_k<64>(a, st, cus);
    return hipErrorInvalidValue;
}

__device__ void heap_sift(float *x, int start, int end)
{
    int root = start;
    while (2 * root + 1 <= end) {
        int child = 2 * root + 1, sw = root;
        if (x[sw] < x[child]) sw = child;
        if (child + 1 <= end && x[sw] < x[child + 1]) sw = child + 1;
        if (sw == root) return;
        float t = x[root]; x[root] = x[sw]; x[sw] = t;
        root = sw;
    }
}

// One thread per user (PYX:1336-1376); the row of rank_data is sorted ascending in place.
__global__ void auc_kernel(DCsr ranks, const int32_t *ntp, float *rank_data, float *auc)
{
    int user = blockIdx.x * blockDim.x + threadIdx.x;
    if (user >= ranks.rows) return;
    int rs0 = ranks.indptr[user], re0 = ranks.indptr[user + 1];
    int npos = re0 - rs0;
    int nneg = ranks.cols - (npos + ntp[user]);
    if (npos == 0 || nneg == ranks.cols) { auc[user] = 0.5f; return; }
    float *x = rank_data + rs0;
    for (int s = (npos - 2) / 2; s >= 0; --s) heap_sift(x, s, npos - 1);
    for (int e = npos - 1; e > 0; --e) {
        float t = x[e]; x[e] = x[0]; x[0] = t;
        heap_sift(x, 0, e - 1);
    }
    float acc = auc[user];
    for (int i = 0; i < npos; ++i) {
        float rank = ranks.data[rs0 + i];
        rank = __fsub_rn(rank, (float)i);
        if (rank < 0.0f) rank = 0.0f;
        acc = (float)((double)acc + (1.0 - (double)__fdiv_rn(rank, (float)nneg)));
    }
    if (npos != 0) acc = __fdiv_rn(acc, (float)npos);
    auc[user] = acc;
}

template <int NC>
static hipError_t predict_nc(const PredictArgs &a, int grid, size_t smem, hipStream_t st)
{
    if (smem > 64 * 1024) {  // (two rows of more than 2 000 floats per wavefront)
        const hipError_t e = hipFuncSetAttribute((const void *)predict_kernel<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    predict_kernel<NC><<<grid, 256, smem, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_predict(const PredictArgs &a, int grid, size_t smem, hipStream_t st)
{
    int d = a.m.d;
    if (d <= 64) return predict_nc<1>(a, grid, smem, st);
    if (d <= 128) return predict_nc<2>(a, grid, smem, st);
    if (d <= 256) return predict_nc<4>(a, grid, smem, st);
    if (d <= 512) return predict_nc<8>(a, grid, smem, st);
    if (d <= LFM_MAX_COMPONENTS) return predict_nc<16>(a, grid, smem, st);
    return hipErrorInvalidValue;
}

hipError_t launch_rep_rows(const DCsr &f, const float *W, const float *b, int d, int rs, float *out,
                           hipStream_t st, int transposed, float *bias_out)
{
    if (f.rows <= 0) return hipSuccess;
    int grid = (int)std::min<int64_t>(4096, ((int64_t)f.rows + 3) / 4);
    if (d <= 64) rep_rows_kernel<1><<<grid, 256, 0, st>>>(f, W, b, d, rs, out, transposed, bias_out);
    else if (d <= 128) rep_rows_kernel<2><<<grid, 256, 0, st>>>(f, W, b, d, rs, out, transposed, bias_out);
    else if (d <= 256) rep_rows_kernel<4><<<grid, 256, 0, st>>>(f, W, b, d, rs, out, transposed, bias_out);
    else if (d <= 512) rep_rows_kernel<8><<<grid, 256, 0, st>>>(f, W, b, d, rs, out, transposed, bias_out);
    else if (d <= LFM_MAX_COMPONENTS) rep_rows_kernel<16><<<grid, 256, 0, st>>>(f, W, b, d, rs, out, transposed, bias_out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_ranks(const RanksArgs &a, hipStream_t st)
{
    if (a.test.rows <= 0) return hipSuccess;
    int grid = std::min(a.test.rows, 8192);
    size_t smem = sizeof(float) * (size_t)(a.rs + 3 * RANK_CHUNK);
    ranks_kernel<<<grid, 256, smem, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_auc(const DCsr &ranks, const int32_t *ntp, float *rank_data, float *auc,
                      hipStream_t st)
{
    if (ranks.rows <= 0) return hipSuccess;
    auc_kernel<<<(ranks.rows + 255) / 256, 256, 0, st>>>(ranks, ntp, rank_data, auc);
    return hipGetLastError();
}

}  // namespace lfm
