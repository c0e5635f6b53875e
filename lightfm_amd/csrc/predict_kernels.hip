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
    a.test_scores[t] = dense_dot(a.user_rep + (size_t)lo * a.rs, a.item_rep, (size_t)a.test.cols, a.test.indices[t], a.d);
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
