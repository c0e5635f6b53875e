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
