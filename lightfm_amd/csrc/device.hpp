// device.hpp -- gfx950 device-side building blocks shared by the epoch kernels.
//
// Work mapping (wave64): ONE wavefront owns one interaction at a time.  Lane l
// owns embedding components c = l, l+64, ... (NC = ceil(d/64) registers), so a
// feature row is a coalesced read, and -- because Adagrad/Adadelta are strictly
// per-coordinate (PYX:416-449) -- every lane read-modify-writes only its own
// column: no intra-wave conflicts.  Control flow (sampling loop, in_positives,
// feature loops) is wave-uniform.
//
// The float32 summation ORDER of the reference (PYX:287-334) is kept: the
// representations of the user and of every candidate item are staged in a
// wave-private LDS tile and lane r computes the whole sequential dot product of
// tile row r, so up to 16 candidate scores cost one 64..d-step pass.
//
// PYX = /root/reference/lightfm/_lightfm_fast.pyx.template
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lfm {

constexpr int WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;
#ifndef LFM_MAX_COMPONENTS
#define LFM_MAX_COMPONENTS 1024  // include/lfm_hip.h: widest model (the generic kernels' NC = 16 coordinates per lane)
#endif
constexpr double MAX_REG_SCALE = 1000000.0;  // PYX:19
constexpr double MAX_LOSS = 10.0;            // PYX:817
// loss ids of include/lfm_hip.h (LFM_LOSS_*)
constexpr int LFM_LOSS_LOGISTIC_ID = 0, LFM_LOSS_WARP_ID = 1, LFM_LOSS_BPR_ID = 2, LFM_LOSS_WARP_KOS_ID = 3;

struct DCsr {
    const int32_t *indices;
    const int32_t *indptr;
    const float *data;
    int32_t rows, cols;
    int32_t identity;  // 1: indptr[i]=i, indices[i]=i, data[i]=1 (host-verified) -> no CSR reads
};

// side 0 = item, 1 = user
struct DModel {
    float *W[2], *G[2], *M[2], *b[2], *bG[2], *bM[2];
    int32_t n_feat[2];
    int32_t d;       // floats per embedding row ON THE DEVICE: no_components rounded up to a multiple of 4 (session.hip:
                     // padded components are zeros in W / M and ones in G; their products, gradients and updates are zeros)
    int32_t d_real;  // no_components: what the learning-rate average of the regularisation scales counts (PYX:640-649)
    int32_t adadelta;
    float lr, rho, eps;
    int32_t max_sampled;
    double *scales;  // device [2]: item_scale, user_scale (PYX:214-215)
};

// Owner-sharded item tables (warp_tile_ahead.hpp, SHARDED): the item-side tables of a multi-GPU job whose item side
// is too large to replicate and merge (BASELINE config C4: 2.6 GB of tables against 57 ms of kernels per epoch) are
// cut into n contiguous row ranges of rows_per_shard rows; range j lives in the memory of its owner (a peer GPU
// reached over xGMI, or -- tests / one-GPU emulation -- another session of this device) and every rank gathers from
// and publishes to the owner's copy: no replicas, no merges, plain Hogwild.  W[j] / G[j] / b[j] / bG[j] point to
// the FIRST ROW OF RANGE j (item row j * rows_per_shard).
struct ItemShards {
    float *W[8], *G[8], *b[8], *bG[8];
    int32_t n;                       // 0: not sharded (a.m.W[0] ... are the tables)
    uint32_t rows_per_shard, magic;  // magic = floor(2^32 / rows_per_shard) + 1 (exact division of ids < 2^31)
};

// ---- shared ("hot") item-feature rows accumulated in LDS slices (hot_slices.hip; round 6) ------------------------------
// A hybrid model's tag / genre rows are updated by a large share of ALL interactions (BASELINE config C3: 16 of the 19
// rows an interaction updates are among 1 128 tag rows), and publishing every cell of every interaction through the L2
// float-atomic unit is that configuration's ceiling (320 G dwords/s).  With a hot set the row-stream kernels
// (feat_kernel.hpp, HOT instantiations) leave those rows OUT of their update and write one record per interaction
// instead; between two launches hot_slice_kernel applies the records to component SLICES of the hot rows held in LDS and
// publishes each slice's total change once.  One record = what update_features needs for the hot entries of the
// interaction (PYX:602-638): per job the gradient coefficient g_j (the hot rows sit on the item side, so the other
// factor is the vector x written next to it: the user representation), and per entry the row's slot and its weight.
constexpr int HOT_EMAX = 24;  // hot entries one record holds; an interaction with more publishes the rest as before
struct __attribute__((aligned(32))) HotRec {
    double g[3];              // gradient coefficient of job 0 / 1 / 2 (PYX:537-649: -loss, +loss, ...)
    int32_t n_total;          // entries; <= 0: nothing to apply at this position (the launch memsets the records to -1)
    unsigned char cnt[3];     // entries of job 0 / 1 / 2, in this order in e[]
    unsigned char pad;
    struct Entry {
        int32_t slot;         // slot of the row in the hot set
        float w;              // the feature weight (PYX:613)
    } e[HOT_EMAX];
    int32_t tail[8];
};
static_assert(sizeof(HotRec) == 256, "HotRec is one 256-byte record");

struct FitArgs {
    DCsr itf, usf, pos;
    DModel m;
    const int32_t *user_ids, *item_ids;
    const float *Y, *weight;
    const int32_t *shuffle;
    const int4 *recs;    // warp_tile.hip: (user, item, Y bits, weight bits) per example, AoS
    const float *b_read[2];  // warp_tile_kernel.hpp: where the SCORING reads biases (item, user):
                             // the live tables, or per-launch cached snapshots of them
    int64_t n;           // all examples (BPR modulo, PYX:1124)
    int64_t begin, end;  // shuffled positions of this launch
    double item_alpha, user_alpha;
    const uint32_t *seeds;
    int32_t seed_idx;      // serial mode: which per-thread stream this launch continues
    const double *logtab;  // [max_sampled+1] log term of the WARP loss, host libm
    int32_t serial;
    int32_t update_mode;   // 0 atomic deltas, 1 plain stores, 2 no writes (ablation)
    int32_t tile_rows, tile_stride, first_batch;
    int32_t debug;           // lfm_opts.debug
    uint32_t n_items_magic;  // floor(2^32 / n_items) + 1 (warp_tile.hip: fast_mod)
    int32_t k, n_pos;      // k-OS
    int32_t pair_cap;      // k-OS: LDS pair slots per wave
    int32_t stage_rows;    // feat_kernel.hpp: rows of the wave's LDS-DMA staging area
    int32_t cand_base;     // feat_kernel.hpp: first candidate-negative row of the representation tile
    int32_t *neg_log, *sampled_log;
    unsigned long long *counters;  // [13]: 4 event counters, 8 phase timers, the fault flag (guard_row)
    const uint32_t *bloom;         // Bloom filter over the positives lookup (struct Bloom below), nullptr = none
    ItemShards shards;             // owner-sharded item tables (n = 0: none)
    int32_t user_store;            // parallel mode: the USER row of an update (identity user features: touched by that user's
                                   // interactions alone) is written with plain stores instead of float atomics (session.hip)
    float *bb[2];                  // steady-state tile kernels (warp_tile_ahead.hpp, warp_tile_narrow.hpp): the bias tables of a side
                                   // as ONE table of (b, bG) pairs, [n_feat][2] -- both cells of a row in one line, so that an
                                   // update publishes them with one line operation instead of two (session.hip: bias pairs);
                                   // nullptr = the separate tables m.b / m.bG
    float *rp[2];                  // narrow-model kernel (warp_tile_narrow.hpp, RP instantiations): W and G of a side as ONE table of
                                   // 128-byte rows [W(16) | G(16)] -- both halves of a row published by one instruction, i.e. one
                                   // line operation instead of two (session.hip: row pairs); nullptr = m.W / m.G
    int32_t rp_bias;               // ... with the bias cells in slot d of each half, [W(d) | b .. | G(d) | bG ..] (d <= 12; BIN instantiations)
    int32_t b_read_stride[2];      // floats between consecutive rows of b_read[side] (1: a snapshot or m.b; 2: bb)
    const int32_t *hot_slot;       // [n_item_feat] slot of a hot item-feature row, -1 otherwise; nullptr = no hot set (HotRec above)
    HotRec *hot_rec;               // [end - begin] one record per position of the launch
    float *hot_x;                  // [end - begin][d] the vector the hot rows' gradients multiply (the user representation)
    float *reg_live;               // [RegScale::FLOATS] parallel mode, lazy L2 regularisation (see RegScale): line 0 =
                                   // min(item_scale, MAX), min(user_scale, MAX) at the last launch boundary, lines 1.. =
                                   // the slots collecting the growth of log(scale) since then (float atomics)
};

// counters[12]: set when a shuffle entry outside [0, n) was read.  A shuffle slot is a permutation of
// [0, n), so this never fires on intact inputs; on corrupted device memory the epoch then ends with
// LFM_ECORRUPT instead of a wild COO index faulting the GPU (which aborts the host process).
constexpr int FAULT_SLOT = 12;
__device__ __forceinline__ int guard_row(const FitArgs &a, int r)
{
    if ((uint32_t)r >= (uint32_t)a.n) {
        a.counters[FAULT_SLOT] = 1ull;
        r = 0;
    }
    return r;
}

// Bloom filter over the positives lookup: an EXACT pre-filter of in_positives (PYX:270-284).  The filter of
// user u occupies the words [lo >> 1, lo >> 1 + n_w) of ONE array, (lo, hi) = u's row bounds in the lookup
// and n_w = max(1, ((hi + 1) >> 1) - (lo >> 1)): 16 bits per positive, addressed from the row bounds the
// kernels hold anyway (no table of its own; neighbouring rows may share a boundary word, which only adds
// false positives).  An item sets / tests 3 bits of one word.  "All three set" = maybe a positive (the exact
// search decides); anything else = certainly not a positive -- the common case of a uniformly drawn
// negative -- decided by ONE 4-byte read whose address does not depend on the row's contents, so it can
// travel together with the candidate's embedding row.
struct Bloom {
    __host__ __device__ static inline uint32_t mix(uint32_t item)
    {
        uint32_t h = item * 0x9E3779B1u;
        h ^= h >> 16;
        h *= 0x85EBCA6Bu;
        h ^= h >> 13;
        return h;
    }
    __host__ __device__ static inline int64_t words(int64_t nnz) { return ((nnz + 1) >> 1) + 1; }
    __host__ __device__ static inline uint32_t word(uint32_t h, int lo, int hi)
    {
        const uint32_t lw = (uint32_t)lo >> 1;
        const uint32_t hw = ((uint32_t)hi + 1u) >> 1;
        const uint32_t nw = hw > lw ? hw - lw : 1u;
        return lw + (uint32_t)(((uint64_t)h * (uint64_t)nw) >> 32);
    }
    __host__ __device__ static inline uint32_t mask(uint32_t h)
    {
        const uint32_t g = h * 0xC2B2AE35u;
        return (1u << (g >> 27)) | (1u << ((g >> 22) & 31u)) | (1u << ((g >> 17) & 31u));
    }
};

// ------------------------------------------------------------------ lanes ---

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniu(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ float unif(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ int read_lane(int v, int l)
{
    return __builtin_amdgcn_readlane(v, uni(l));
}
__device__ __forceinline__ float read_lanef(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), uni(l)));
}
__device__ __forceinline__ double unid(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double read_laned(double v, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), uni(l));
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), uni(l));
    return __hiloint2double(hi, lo);
}
// LDS traffic between lanes of ONE wave: LDS ops of a wave execute in order, so a
// compiler-level barrier is all that is needed.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// ------------------------------------------------------------------- PRNG ---

// PYX:64-76
__device__ __forceinline__ uint32_t temper(uint32_t x)
{
    x ^= x >> 11;
    x ^= (x << 7) & 0x9D2C5680u;
    x ^= (x << 15) & 0xEFC60000u;
    x ^= x >> 18;
    return x;
}
// PYX:79-81: seed = seed*1103515245+12345; return temper(seed)/2
__device__ __forceinline__ uint32_t lcg(uint32_t s) { return s * 1103515245u + 12345u; }
__device__ __forceinline__ uint32_t draw(uint32_t s) { return temper(s) >> 1; }

// Parallel mode: stream owned by shuffled position i (same rule as
// oracle/lfm_oracle.c:orc_position_seed).
__device__ __forceinline__ uint32_t position_seed(uint32_t base, uint64_t i)
{
    uint32_t h = base + (uint32_t)i * 0x9E3779B9u + (uint32_t)(i >> 32) * 0x85EBCA6Bu;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// ---------------------------------------------------------- table access ---

// Embedding tables are written concurrently by other wavefronts.  Plain loads: how stale a
// value may be is bounded by the launch (a kernel boundary is a device-wide release/acquire;
// serial mode additionally fences after every interaction), and measured precision@10 does not
// depend on it.  Agent-scope atomic loads (`global_load sc1`) would bypass only the
// L1 and make hipcc drain vmcnt before every one of them, serialising the row gathers.
__device__ __forceinline__ float ldw(const float *p) { return *p; }

// PYX:270-284 as a 64-ary search: each round the wave probes 64 evenly spaced
// entries of the sorted row, so rows up to 4096 long need 2 dependent loads.
// [lo, hi) = indptr[user], indptr[user+1] (callers prefetch them).
__device__ __forceinline__ bool in_positives_range(const DCsr &p, int item, int lo, int hi, int lane)
{
    while (hi - lo > WAVE) {
        int step = (hi - lo + WAVE - 1) >> 6;
        int idx = lo + lane * step;
        bool ok = idx < hi;
        int v = ok ? p.indices[idx] : 0x7fffffff;
        unsigned long long m = __ballot(ok && v <= item);
        int cnt = __popcll(m);
        if (cnt == 0) return false;
        lo = lo + (cnt - 1) * step;
        hi = min(hi, lo + step);
    }
    int idx = lo + lane;
    bool hit = (idx < hi) && (p.indices[idx] == item);
    return __ballot(hit) != 0ull;
}

__device__ __forceinline__ bool in_positives(const DCsr &p, int item, int user, int lane)
{
    return in_positives_range(p, item, uni(p.indptr[user]), uni(p.indptr[user + 1]), lane);
}

template <int NC>
struct Rep {
    float v[NC];  // component lane+64q
    float bias;   // wave-uniform
};

// compute_representation, PYX:287-317 (w = (float)((double)data*scale), C_OMP:4896;
// float32 multiply then add, features in CSR order).
template <int NC, bool IDENT = false>
__device__ __forceinline__ void load_rep(const DCsr &f, const float *W, const float *b, int d,
                                         int row, double scale, int lane, Rep<NC> &r)
{
    if (IDENT || f.identity) {
        float w = IDENT ? 1.0f : (float)(1.0 * scale);
        const float *wr = W + (size_t)row * d;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            int c = lane + WAVE * q;
            float x = (c < d) ? ldw(wr + c) : 0.0f;
            r.v[q] = __fadd_rn(0.0f, __fmul_rn(w, x));
        }
        r.bias = __fadd_rn(0.0f, __fmul_rn(w, ldw(b + row)));
        return;
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) r.v[q] = 0.0f;
    r.bias = 0.0f;
    const int s = uni(f.indptr[row]), e = uni(f.indptr[row + 1]);
    // The row's (feature, weight) entries are fetched 64 at a time, one per lane (one round
    // trip), then the embedding rows 8 at a time (one round trip per 8 features); the float32
    // accumulation still runs in CSR order.
    for (int k0 = s; k0 < e; k0 += WAVE) {
        const int cnt = min(WAVE, e - k0);
        const int myfeat = lane < cnt ? f.indices[k0 + lane] : 0;
        const float myw = lane < cnt ? f.data[k0 + lane] : 0.0f;
        for (int j0 = 0; j0 < cnt; j0 += 8) {
            float x[8][NC], bx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int feat = read_lane(myfeat, min(j0 + j, cnt - 1));
                const float *wr = W + (size_t)feat * d;
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int c = lane + WAVE * q;
                    x[j][q] = ldw(wr + (c < d ? c : 0));
                }
                bx[j] = ldw(b + feat);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j0 + j < cnt) {
                    const float w = (float)((double)read_lanef(myw, j0 + j) * scale);
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        const int c = lane + WAVE * q;
                        r.v[q] = __fadd_rn(r.v[q], __fmul_rn(w, c < d ? x[j][q] : 0.0f));
                    }
                    r.bias = __fadd_rn(r.bias, __fmul_rn(w, bx[j]));
                }
            }
        }
    }
}

template <int NC>
__device__ __forceinline__ void rep_to_tile(float *trow, const Rep<NC> &r, int d, int lane)
{
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        int c = lane + WAVE * q;
        if (c < d) trow[c] = r.v[q];
    }
    if (lane == 0) trow[d] = r.bias;
}

template <int NC>
__device__ __forceinline__ void rep_from_tile(const float *trow, int d, int lane, Rep<NC> &r)
{
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        int c = lane + WAVE * q;
        r.v[q] = (c < d) ? trow[c] : 0.0f;
    }
    r.bias = trow[d];
}

// compute_prediction_from_repr, PYX:320-334: sequential float32 sum starting
// from the two biases.  One lane walks one tile row.
__device__ __forceinline__ float tile_dot(const float *u, const float *v, int d)
{
    float acc = __fadd_rn(u[d], v[d]);
    int c = 0;
    if ((d & 3) == 0) {
        // 16 components per step: eight ds_read_b128 in flight, the products (independent) before the additions (one
        // sequential chain) -- the LDS latency is paid once per 16 components instead of once per 4
        for (; c + 16 <= d; c += 16) {
            float4 a[4], x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = *reinterpret_cast<const float4 *>(u + c + 4 * j);
                x[j] = *reinterpret_cast<const float4 *>(v + c + 4 * j);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p0 = __fmul_rn(a[j].x, x[j].x), p1 = __fmul_rn(a[j].y, x[j].y);
                const float p2 = __fmul_rn(a[j].z, x[j].z), p3 = __fmul_rn(a[j].w, x[j].w);
                acc = __fadd_rn(acc, p0);
                acc = __fadd_rn(acc, p1);
                acc = __fadd_rn(acc, p2);
                acc = __fadd_rn(acc, p3);
            }
        }
        for (; c < d; c += 4) {
            float4 a = *reinterpret_cast<const float4 *>(u + c);
            float4 x = *reinterpret_cast<const float4 *>(v + c);
            acc = __fadd_rn(acc, __fmul_rn(a.x, x.x));
            acc = __fadd_rn(acc, __fmul_rn(a.y, x.y));
            acc = __fadd_rn(acc, __fmul_rn(a.z, x.z));
            acc = __fadd_rn(acc, __fmul_rn(a.w, x.w));
        }
    } else {
        for (; c < d; ++c) acc = __fadd_rn(acc, __fmul_rn(u[c], v[c]));
    }
    return acc;
}

// PYX:262-267
__device__ __forceinline__ float sigmoidf_ref(float v)
{
    return (float)(1.0 / (1.0 + exp(-(double)v)));
}

// ------------------------------------------------------------ optimizer ---

struct Hyper {
    int adadelta;
    float lr, rho, eps;
};

// One optimizer cell: PYX:416-449 with the float64 promotions of C_OMP:5340-5560,
// as a pure function of the old (W, G, M) values.
__device__ __forceinline__ void cell_math(float oW, float oG, float oM, double w, double g,
                                          const Hyper &h, double alpha, float &nW, float &nG,
                                          float &nM, double &lr)
{
    nM = oM;
    if (h.adadelta) {
        float rg = __fmul_rn(h.rho, oG);
        double wg = w * g;
        nG = (float)((double)rg + (1.0 - (double)h.rho) * (wg * wg));
        float me = __fadd_rn(oM, h.eps), ge = __fadd_rn(nG, h.eps);
        lr = sqrt((double)me) / sqrt((double)ge);
        double upd = (lr * g) * w;
        float rm = __fmul_rn(h.rho, oM);
        nM = (float)((double)rm + (1.0 - (double)h.rho) * (upd * upd));
        nW = (float)((double)oW - upd);
    } else {
        lr = (double)h.lr / sqrt((double)oG);
        nW = (float)((double)oW - (lr * w) * g);
        double gw = g * w;
        nG = (float)((double)oG + gw * gw);
    }
    nW = (float)((double)nW * (1.0 + alpha * lr));
}

// The adagrad cell of cell_math (alpha = 0) WITHOUT the float64 square root and division, bit-identical by construction
// (round 6: the float64 cell is what the LDS slice kernel is bound by, profiles/r06_membench_lds.txt).
// lr / sqrt(G) is taken as r = fl(lr y), y = rsq(G) refined by two Newton steps (what is left is the rounding of the last
// step and of the product: |r - fl(lr / fl(sqrt(G)))| <= 2^-50.8 |r|, the reference's own two roundings included).  The same
// float64 operations as the reference then give t2' = fl(fl(r w) g) within 2^-49.9 |t2| of the reference's t2 and d' =
// fl(oW - t2') within 2^-49.9 |t2| + 2^-52 |d'| of its d; D = 2^-49 |t2'| + 2^-51 |d'| bounds that with room to spare.  (float)d' equals (float)d unless a float32 rounding boundary -- the midpoint between two neighbouring
// floats -- lies between the two, i.e. unless d' is within D of such a midpoint: checked, and only then (a few cells in
// 10^8; also NaN, infinities, tiny values) the exact cell runs.  nG needs neither root nor quotient.
__device__ __forceinline__ void cell_math_adagrad(float oW, float oG, double w, double g, float lr_f, float &nW, float &nG)
{
    const double G = (double)oG;
    const double gw = g * w;
    nG = (float)(G + gw * gw);
    double y = __builtin_amdgcn_rsq(G);                        // (v_rsq_f64: ~2^-26; two steps leave rounding only)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-G * y, y, 1.0);        // 1 - G y^2
        y = __builtin_fma(0.5 * y, e, y);                      // y (1 + e / 2)
    }
    const double r = (double)lr_f * y;
    const double t2 = (r * w) * g;
    const double d = (double)oW - t2;
    const float m = (float)d;
    const int mb = __float_as_int(m) & 0x7f800000;             // exponent field of m
    const double half_ulp = (double)__int_as_float(mb - (24 << 23));  // ulp32(m) / 2 for normal m with exponent field >= 25
    const double rho = fabs(d - (double)m);
    const double D = 0x1p-49 * fabs(t2) + 0x1p-51 * fabs(d);
    // (a power of two has its lower neighbour at half the spacing: the rare m = 2^k takes the exact cell as well)
    if (mb >= (25 << 23) && mb < 0x7f800000 && (__float_as_int(m) & 0x007fffff) != 0 && (half_ulp - rho) > D) {
        nW = m;
    } else {
        const double lr = (double)lr_f / sqrt(G);
        nW = (float)((double)oW - (lr * w) * g);
    }
}

// Publish new-old with global_atomic_add_f32: exactly the new value when nobody else
// touched the cell in between (Hogwild otherwise).
__device__ __forceinline__ void publish(float *p, float nv, float ov, int mode = 0)
{
    if (mode == 0) {
        float dlt = __fsub_rn(nv, ov);
        if (dlt != 0.0f) atomicAdd(p, dlt);
    } else if (mode == 1) {
        if (nv != ov) *p = nv;
    }
}

// Adadelta in parallel (Hogwild) mode.  Its accumulators are exponential moving averages
// (PYX:416-434): publishing new - old would apply the decay -(1-rho)*old once per CONCURRENT
// writer -- K wavefronts that read the same G of a popular row leave G*(1 - K(1-rho)) + ...,
// negative once K > 1/(1-rho) = 20, and the next sqrt is a NaN.  So G and M are written with
// compare-and-swap against the value the arithmetic started from and recomputed from the value
// actually in memory on conflict: every writer sees a distinct predecessor, exactly like some
// sequential interleaving.  Without contention this is the plain cell (bit-identical).  W is
// still an additive delta.
__device__ __forceinline__ double publish_adadelta(float *Wp, float *Gp, float *Mp, float oW, float oG,
                                                   float oM, double w, double g, const Hyper &h,
                                                   double alpha)
{
    float nW, nG, nM, cg = oG, cm = oM;
    double lr;
    for (;;) {
        cell_math(oW, cg, cm, w, g, h, alpha, nW, nG, nM, lr);
        const int prev = atomicCAS(reinterpret_cast<int *>(Gp), __float_as_int(cg), __float_as_int(nG));
        if (prev == __float_as_int(cg)) break;
        cg = __int_as_float(prev);
    }
    for (;;) {
        const int prev = atomicCAS(reinterpret_cast<int *>(Mp), __float_as_int(cm), __float_as_int(nM));
        if (prev == __float_as_int(cm)) break;
        cm = __int_as_float(prev);
        cell_math(oW, cg, cm, w, g, h, alpha, nW, nG, nM, lr);
    }
    const float dlt = __fsub_rn(nW, oW);
    if (dlt != 0.0f) atomicAdd(Wp, dlt);
    return lr;
}

// Publication of one cell whose new values were computed by cell_math from (oW, oG, oM).
// mode: 0 atomic (deltas; adadelta accumulators by compare-and-swap), 1 plain stores, 2 nothing.
__device__ __forceinline__ void publish_cell(float *Wp, float *Gp, float *Mp, float oW, float oG,
                                             float oM, float nW, float nG, float nM, double w, double g,
                                             const Hyper &h, double alpha, int mode)
{
    if (mode == 0 && h.adadelta) {
        publish_adadelta(Wp, Gp, Mp, oW, oG, oM, w, g, h, alpha);
        return;
    }
    publish(Wp, nW, oW, mode);
    publish(Gp, nG, oG, mode);
    if (h.adadelta) publish(Mp, nM, oM, mode);
}

// Load-compute-store of one cell.  `atomic`: publish deltas; otherwise plain stores
// (serial mode).
__device__ __forceinline__ double cell_update(float *Wp, float *Gp, float *Mp, double w, double g,
                                              const Hyper &h, double alpha, bool atomic)
{
    float oW = ldw(Wp), oG = ldw(Gp), oM = h.adadelta ? ldw(Mp) : 0.0f;
    float nW, nG, nM;
    double lr;
    cell_math(oW, oG, oM, w, g, h, alpha, nW, nG, nM, lr);
    if (atomic) {
        publish_cell(Wp, Gp, Mp, oW, oG, oM, nW, nG, nM, w, g, h, alpha, 0);
    } else {
        *Wp = nW;
        *Gp = nG;
        if (h.adadelta) *Mp = nM;
    }
    return lr;
}

// Learning rates summed the way the reference sums them (PYX:571-638): one
// partial per (row, coordinate) and per row's biases.
template <int NC>
struct LrSums {
    double bias[3];
    double comp[3][NC];
};

// update_biases + update_features for ONE row of a feature matrix (PYX:337-451).
// x[q] is the per-lane factor of the gradient: g = gcoef * (double)x[q].
template <int NC>
__device__ __forceinline__ void update_row(const DCsr &f, int row, int side, const DModel &m,
                                           const float (&x)[NC], double gcoef, double gbias,
                                           double alpha, bool atomic, int lane, double &lr_bias,
                                           double (&lr_comp)[NC])
{
    const Hyper h{m.adadelta, m.lr, m.rho, m.eps};
    const int d = m.d;
    int s, e;
    if (f.identity) { s = row; e = row + 1; }
    else { s = uni(f.indptr[row]); e = uni(f.indptr[row + 1]); }
    lr_bias = 0.0;
#pragma unroll
    for (int q = 0; q < NC; ++q) lr_comp[q] = 0.0;
    // biases first (PYX:571-599)
    for (int k = s; k < e; ++k) {
        int feat = f.identity ? k : uni(f.indices[k]);
        double w = f.identity ? 1.0 : (double)unif(f.data[k]);
        double lr = 0.0;
        if (lane == 0)
            lr = cell_update(m.b[side] + feat, m.bG[side] + feat, m.bM[side] + feat, w, gbias, h,
                             alpha, atomic);
        lr_bias += read_laned(lr, 0);
    }
    // then every coordinate (PYX:602-638); lane owns its coordinates
    for (int k = s; k < e; ++k) {
        int feat = f.identity ? k : uni(f.indices[k]);
        double w = f.identity ? 1.0 : (double)unif(f.data[k]);
        size_t base = (size_t)feat * d;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            int c = lane + WAVE * q;
            if (c < d) {
                double g = gcoef * (double)x[q];
                const double lr = cell_update(m.W[side] + base + c, m.G[side] + base + c,
                                              m.M[side] + base + c, w, g, h, alpha, atomic);
                if (c < m.d_real) lr_comp[q] += lr;  // (padded components are no cells of the reference's model)
            }
        }
    }
}

// update_row for parallel (Hogwild) mode: the same cells and the same arithmetic, but the row's
// entries are fetched in one round trip (one per lane), every bias cell is updated by its own
// lane at once, and the coordinate cells of 4 features are in flight together.  A feature that
// occurs twice in the row gets both updates computed from the same old value (Hogwild within the
// wavefront); serial mode keeps the strictly sequential update_row above.
template <int NC>
__device__ __forceinline__ void update_row_batched(const DCsr &f, int row, int side, const DModel &m,
                                                   const float (&x)[NC], double gcoef, double gbias,
                                                   double alpha, bool atomic, int lane,
                                                   double &lr_bias, double (&lr_comp)[NC])
{
    const Hyper h{m.adadelta, m.lr, m.rho, m.eps};
    const int d = m.d;
    int s, e;
    if (f.identity) { s = row; e = row + 1; }
    else { s = uni(f.indptr[row]); e = uni(f.indptr[row + 1]); }
    lr_bias = 0.0;
#pragma unroll
    for (int q = 0; q < NC; ++q) lr_comp[q] = 0.0;
    for (int k0 = s; k0 < e; k0 += WAVE) {
        const int cnt = min(WAVE, e - k0);
        int myfeat = k0 + lane;
        float myw = 1.0f;
        if (!f.identity) {
            myfeat = lane < cnt ? f.indices[k0 + lane] : 0;
            myw = lane < cnt ? f.data[k0 + lane] : 0.0f;
        }
        // biases (PYX:571-599): lane j owns entry j
        double lrb = 0.0;
        if (lane < cnt)
            lrb = cell_update(m.b[side] + myfeat, m.bG[side] + myfeat, m.bM[side] + myfeat, (double)myw,
                              gbias, h, alpha, atomic);
        if (alpha != 0.0) lr_bias += wave_sum(lrb);
        // coordinates (PYX:602-638): lane owns its coordinates, 4 entries in flight
        for (int j0 = 0; j0 < cnt; j0 += 4) {
            float oW[4][NC], oG[4][NC], oM[4][NC];
            size_t base[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                base[j] = (size_t)read_lane(myfeat, min(j0 + j, cnt - 1)) * d;
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int c = lane + WAVE * q;
                    const int cc = c < d ? c : 0;
                    oW[j][q] = ldw(m.W[side] + base[j] + cc);
                    oG[j][q] = ldw(m.G[side] + base[j] + cc);
                    oM[j][q] = h.adadelta ? ldw(m.M[side] + base[j] + cc) : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j0 + j < cnt) {
                    const double w = (double)read_lanef(myw, j0 + j);
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        const int c = lane + WAVE * q;
                        if (c < d) {
                            float nW, nG, nM;
                            double lr;
                            cell_math(oW[j][q], oG[j][q], oM[j][q], w, gcoef * (double)x[q], h, alpha, nW,
                                      nG, nM, lr);
                            if (c < m.d_real) lr_comp[q] += lr;
                            publish_cell(m.W[side] + base[j] + c, m.G[side] + base[j] + c, m.M[side] + base[j] + c,
                                         oW[j][q], oG[j][q], oM[j][q], nW, nG, nM, w, gcoef * (double)x[q], h, alpha,
                                         atomic ? 0 : 1);
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Parallel (Hogwild) update of the 2 or 3 feature rows an interaction touches, as ONE pipeline:
//   A  the (feature, weight) entries of every row, one per lane: one round trip for all rows;
//   B  every bias cell by its own lane: all loads, then all arithmetic, then all publications;
//   C  the coordinate cells, entries of all rows flattened and taken CH at a time; the loads of
//      chunk k+1 are issued BEFORE chunk k is published, so no load ever waits behind an
//      atomic's acknowledgement (vmcnt is an in-order counter).
// Same cells and same float64 cell arithmetic as update_row (PYX:337-451); what is relaxed is the
// order between cells, which Hogwild mode does not keep anyway.  Rows longer than a wavefront
// (and NC > 2) take the per-row path above.  Returns the sum of the cells' learning rates.
template <int NC>
struct RowJob {
    DCsr f;  // by value: a pointer into the kernel argument block would force it into scratch
    int row, side;
    float x[NC];          // per-lane gradient factors of the row
    double gcoef, gbias;  // g = gcoef * x[q] for coordinates, gbias for the bias cells
    double alpha;
};

// Three named values picked by a wave-uniform row index.  (Deliberately not an array: LLVM turns
// a select chain over array elements into a dynamically indexed load, i.e. scratch memory.)
template <typename T>
struct Tri {
    T a, b, c;
    __device__ __forceinline__ T pick(int r) const { return r == 0 ? a : (r == 1 ? b : c); }
    __device__ __forceinline__ T &at(int r) { return r == 0 ? a : (r == 1 ? b : c); }
};

template <int NC, int NR>
__device__ __forceinline__ double rows_update_parallel(const DModel &m, const RowJob<NC> (&job)[NR],
                                                       bool atomic, int lane)
{
    static_assert(NR == 2 || NR == 3, "two (logistic) or three (WARP / BPR / k-OS) rows");
    constexpr int CH = 4;
    constexpr int L = NR - 1;  // index of the last real row; unused slots of a Tri mirror it
    const Hyper h{m.adadelta, m.lr, m.rho, m.eps};
    const int d = m.d;
    const Tri<int> side{job[0].side, job[1].side, job[L].side};
    const Tri<double> gcoef{job[0].gcoef, job[1].gcoef, job[L].gcoef};
    const Tri<double> gbias{job[0].gbias, job[1].gbias, job[L].gbias};
    const Tri<double> alpha{job[0].alpha, job[1].alpha, job[L].alpha};
    Tri<float> x[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) x[q] = Tri<float>{job[0].x[q], job[1].x[q], job[L].x[q]};
    Tri<int> n{0, 0, 0}, feat{0, 0, 0};
    Tri<float> w{0.0f, 0.0f, 0.0f};
    bool big = false;
#pragma unroll
    for (int r = 0; r < NR; ++r) {  // A
        const DCsr &f = job[r].f;
        if (f.identity) {
            n.at(r) = 1;
            feat.at(r) = job[r].row;
            w.at(r) = 1.0f;
        } else {
            const int s = uni(f.indptr[job[r].row]), e = uni(f.indptr[job[r].row + 1]);
            n.at(r) = e - s;
            big = big || (e - s) > WAVE;
            feat.at(r) = lane < e - s ? f.indices[s + min(lane, e - s - 1)] : 0;
            w.at(r) = lane < e - s ? f.data[s + min(lane, e - s - 1)] : 0.0f;
        }
    }
    double lr_acc = 0.0;
    // adadelta's accumulators are published by compare-and-swap (publish_adadelta), which needs
    // the old values at publication time: the per-row path keeps them
    if (big || (h.adadelta && atomic)) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            double lb, lc[NC];
            float xv[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) xv[q] = job[r].x[q];
            update_row_batched<NC>(job[r].f, job[r].row, job[r].side, m, xv, job[r].gcoef, job[r].gbias,
                                   job[r].alpha, atomic, lane, lb, lc);
            lr_acc += lane == 0 ? lb : 0.0;
#pragma unroll
            for (int q = 0; q < NC; ++q) lr_acc += lc[q];
        }
        return wave_sum(lr_acc);
    }
    auto put = [&](float *p, float nv, float ov) {
        if (atomic) {
            const float dlt = __fsub_rn(nv, ov);
            if (dlt != 0.0f) atomicAdd(p, dlt);
        } else {
            *p = nv;
        }
    };
    {  // B
        float oW[NR], oG[NR], oM[NR], nW[NR], nG[NR], nM[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bool on = lane < n.pick(r);
            const int s = side.pick(r), ft = feat.pick(r);
            oW[r] = on ? ldw(m.b[s] + ft) : 0.0f;
            oG[r] = on ? ldw(m.bG[s] + ft) : 1.0f;
            oM[r] = (on && h.adadelta) ? ldw(m.bM[s] + ft) : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            double lr;
            cell_math(oW[r], oG[r], oM[r], (double)w.pick(r), gbias.pick(r), h, alpha.pick(r), nW[r], nG[r],
                      nM[r], lr);
            if (lane < n.pick(r)) lr_acc += lr;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (lane < n.pick(r)) {
                const int s = side.pick(r), ft = feat.pick(r);
                put(m.b[s] + ft, nW[r], oW[r]);
                put(m.bG[s] + ft, nG[r], oG[r]);
                if (h.adadelta) put(m.bM[s] + ft, nM[r], oM[r]);
            }
        }
    }
    // C
    const int total = n.a + n.b + (NR == 3 ? n.c : 0);
    auto locate = [&](int e, int &r, int &j) {  // wave-uniform
        r = 0;
        j = e;
        if (j >= n.a) {
            j -= n.a;
            r = 1;
            if (NR == 3 && j >= n.b) {
                j -= n.b;
                r = 2;
            }
        }
    };
    float oW[CH][NC], oG[CH][NC], oM[CH][NC], cw[CH];
    int cfeat[CH], crow[CH];
    auto load_chunk = [&](int e0) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            int r, j;
            locate(min(e0 + k, total - 1), r, j);
            const int sd = side.pick(r);
            crow[k] = r;
            cw[k] = read_lanef(w.pick(r), j);
            cfeat[k] = read_lane(feat.pick(r), j);
            const size_t base = (size_t)cfeat[k] * d;
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int cc = (lane + WAVE * q) < d ? (lane + WAVE * q) : 0;
                oW[k][q] = ldw(m.W[sd] + base + cc);
                oG[k][q] = ldw(m.G[sd] + base + cc);
                oM[k][q] = h.adadelta ? ldw(m.M[sd] + base + cc) : 0.0f;
            }
        }
    };
    load_chunk(0);
    for (int e0 = 0; e0 < total; e0 += CH) {
        // arithmetic of the chunk; v* = what is published (deltas when atomic)
        float vW[CH][NC], vG[CH][NC], vM[CH][NC];
        int vfeat[CH], vrow[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            vfeat[k] = cfeat[k];
            vrow[k] = crow[k];
            const int r = crow[k];
            const double gc = gcoef.pick(r), al = alpha.pick(r);
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                float nW, nG, nM;
                double lr;
                cell_math(oW[k][q], oG[k][q], oM[k][q], (double)cw[k], gc * (double)x[q].pick(r), h, al, nW, nG,
                          nM, lr);
                if (e0 + k < total && lane + WAVE * q < m.d_real) lr_acc += lr;
                vW[k][q] = atomic ? __fsub_rn(nW, oW[k][q]) : nW;
                vG[k][q] = atomic ? __fsub_rn(nG, oG[k][q]) : nG;
                vM[k][q] = atomic ? __fsub_rn(nM, oM[k][q]) : nM;
            }
        }
        // the next chunk's loads go out BEFORE this chunk is published
        if (e0 + CH < total) load_chunk(e0 + CH);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (e0 + k < total) {
                const int sd = side.pick(vrow[k]);
                const size_t base = (size_t)vfeat[k] * d;
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int cidx = lane + WAVE * q;
                    if (cidx < d) {
                        if (atomic) {
                            if (vW[k][q] != 0.0f) atomicAdd(m.W[sd] + base + cidx, vW[k][q]);
                            if (vG[k][q] != 0.0f) atomicAdd(m.G[sd] + base + cidx, vG[k][q]);
                            if (h.adadelta && vM[k][q] != 0.0f) atomicAdd(m.M[sd] + base + cidx, vM[k][q]);
                        } else {
                            m.W[sd][base + cidx] = vW[k][q];
                            m.G[sd][base + cidx] = vG[k][q];
                            if (h.adadelta) m.M[sd][base + cidx] = vM[k][q];
                        }
                    }
                }
            }
        }
    }
    // only the lazy-regularisation scale step (alpha != 0) consumes the learning-rate sum
    return (alpha.a != 0.0 || alpha.b != 0.0 || alpha.c != 0.0) ? wave_sum(lr_acc) : 0.0;
}

// Ordered (reference order) or tree sum of the learning-rate partials -> the
// avg_learning_rate of PYX:640-649 before the division.
template <int NC, int NROWS>
__device__ __forceinline__ double sum_lr(const double (&lr_bias)[3], const double (&lr_comp)[3][NC],
                                         int d, bool ordered, int lane)
{
    if (!ordered) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < NROWS; ++r)
#pragma unroll
            for (int q = 0; q < NC; ++q) t += lr_comp[r][q];
        t = wave_sum(t);
#pragma unroll
        for (int r = 0; r < NROWS; ++r) t += lr_bias[r];
        return t;
    }
    double avg = 0.0;
    for (int r = 0; r < NROWS; ++r) avg += lr_bias[r];
    for (int c = 0; c < d; ++c) {
        int q = c >> 6, l = c & 63;
        for (int r = 0; r < NROWS; ++r) {
            double v = 0.0;
#pragma unroll
            for (int qq = 0; qq < NC; ++qq)
                if (qq == q) v = lr_comp[r][qq];
            avg += read_laned(v, l);
        }
    }
    return avg;
}

__device__ __forceinline__ int row_len(const DCsr &f, int row)
{
    return f.identity ? 1 : (f.indptr[row + 1] - f.indptr[row]);
}

// Lazy L2 regularisation in PARALLEL mode (PYX:640-691).  The reference keeps ONE global scale per side
// that every interaction multiplies by (1 + alpha * avg_lr) and that multiplies every representation
// computed afterwards; all OpenMP threads share it (a plain, racy read-modify-write of a double).  Here
// the scale of a side is
//     S = S0 * exp(D)
// with S0 = the scale at the last launch boundary (float32, constant while a launch runs) and D = the
// growth of its logarithm since then, summed over every interaction of the launch that has updated.
// One address that a billion interactions per second read and add to serialises the chip on ONE L2
// channel (measured in round 3: ~85 M same-line operations/s -- C2 fell from 1.05 G to 0.10 G
// interactions/s), so during a launch nothing shared is read or written:
//   * WRITERS: a wavefront collects the log(1 + alpha * avg_lr) of its own interactions in two registers
//     and publishes them ONCE, when it leaves the kernel, to one of SLOTS accumulator lines (wave id mod
//     SLOTS; float atomics, additions commute).  The boundary kernels (fit_kernels.hip) add the slots to
//     float64 running totals -- the exact product of the launch -- reset them, and fold the scale into
//     the weights (W / scale, scale := 1: regularize, PYX:652-675) once it has passed MAX_REG_SCALE, and
//     at the end of the epoch.
//   * READERS extrapolate: the logarithm of the scale at position q of the launch is
//     T(q) = L0 + rho * (q - begin), L0 = its value at the boundary, rho = the growth per position
//     MEASURED over the previous launch (boundary kernel: D / positions).  Positions are visited in order
//     by the grid-stride loops, so this is first-order exact; the error is the drift of the rate from one
//     launch to the next (the Adagrad rates and the update frequency change by a few per cent per
//     launch at most) times D, and the session bounds a launch to D <= ~0.5 (session.hip) unless alpha is
//     excessive: a relative scale error of ~1e-2 at the very worst, 1e-4 at alpha = 1e-6 -- less than the
//     reference's own threads lose in their racy multiply.  At every boundary the exact total replaces
//     the estimate.
//   * FOLDS.  The reference folds the moment a scale passes MAX_REG_SCALE (locked_regularize, PYX:678-691):
//     W := W / scale, scale := 1 -- after which representations (scale * W, PYX:306-313) are smaller by
//     scale^2.  A launch cannot divide the tables, so a crossing inside a launch is a VIRTUAL fold: with
//     LMAX = log(MAX_REG_SCALE) and nf = floor(T / LMAX) crossings so far, the reference's running scale is
//     exp(T - nf LMAX) and its stored weights are ours / exp(nf LMAX); readers therefore multiply our
//     weights by exp(T - 2 nf LMAX).  The boundary kernel applies the nf folds for real (W / exp(nf LMAX),
//     L0 := T - nf LMAX).  At sane alpha nf is 0 for every launch but one in many epochs; at excessive
//     alpha (the reference's alpha = 1 test: a crossing every ~280 interactions) the multiplier underflows
//     and the model is flattened to zero exactly as the reference's is -- in full-length launches.
//     (Cell updates between a virtual fold and the next boundary are applied at the pre-fold magnitude.)
// a.reg_live = [1 + SLOTS] lines of 128 B (uncached device memory): line 0 = {L0_item, L0_user, rho_item,
// rho_user} (written between launches only), line 1 + s = {D_item, D_user} of slot s.
struct RegScale {
    static constexpr int SLOTS = 16;
    static constexpr int LINE = 32;                    // floats per 128-B line
    static constexpr int FLOATS = LINE * (1 + SLOTS);  // size of the reg_live buffer

    float p_i, p_u;  // wave-uniform: collected by this wavefront, published when it leaves

    // exp(t), t >= 0 small (a launch's growth); the hardware exponential beyond
    __device__ static __forceinline__ float exp_f32(float t)
    {
        if (t < 0.03125f) return 1.0f + t * (1.0f + t * 0.5f * (1.0f + t * (1.0f / 3) * (1.0f + t * 0.25f)));
        return __expf(t);
    }
    // log(1 + x) for the x = alpha * avg_lr >= 0 of one interaction
    __device__ static __forceinline__ float log1p_f32(float x)
    {
        if (x < 0.0625f)
            return x * (1.0f - x * (0.5f - x * ((1.0f / 3) - x * (0.25f - x * (0.2f - x * (1.0f / 6))))));
        return __logf(1.0f + x);
    }
    __device__ static __forceinline__ float first_lane(float x)
    {
        return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
    }
    static constexpr float LMAX = 13.815510557964274f;  // log(MAX_REG_SCALE)
    // multiplier of this table's stored weights at log-scale T (see FOLDS above)
    __device__ static __forceinline__ float multiplier(float T)
    {
        T = fmaxf(T, 0.0f);
        const float nf = floorf(T * (1.0f / LMAX));
        const float e = T - 2.0f * nf * LMAX;
        return (e >= 0.0f && e < 0.03125f) ? exp_f32(e) : __expf(e);
    }
    // (float)(1.0 * scale) of both sides at position `done` (= q - begin) of the launch, as
    // compute_representation uses it (PYX:306); wave-uniform
    __device__ static __forceinline__ void scales(const float *reg_live, int64_t done, float &w_item, float &w_user)
    {
        const float4 h = *reinterpret_cast<const float4 *>(reg_live);  // constant while the launch runs
        const float n = (float)done;
        w_item = first_lane(multiplier(h.x + h.z * n));
        w_user = first_lane(multiplier(h.y + h.w * n));
    }
    __device__ __forceinline__ void begin() { p_i = p_u = 0.0f; }
    // one updated interaction of this wavefront (all lanes call it; lane 0's values count)
    __device__ __forceinline__ void add(float add_item, float add_user)
    {
        p_i = first_lane(p_i + add_item);
        p_u = first_lane(p_u + add_user);
    }
    __device__ static __forceinline__ void publish(float *reg_live, float pi, float pu, int lane, unsigned wave_id)
    {
        if (lane == 0) {
            float *q = reg_live + LINE * (1 + (int)(wave_id % (unsigned)SLOTS));
            if (pi != 0.0f) atomicAdd(q + 0, pi);  // global_atomic_add_f32
            if (pu != 0.0f) atomicAdd(q + 1, pu);
        }
    }
};

struct Scales {
    double item, user;  // scales used when computing representations (PYX:311)
    RegScale live;      // parallel mode with alpha != 0
};

__device__ __forceinline__ bool reg_active(const FitArgs &a)
{
    return !a.serial && (a.item_alpha != 0.0 || a.user_alpha != 0.0);
}

// Start of the interaction at shuffled position i in parallel mode: the estimate of the live scales.
__device__ __forceinline__ void refresh_scales(const FitArgs &a, Scales &sc, int64_t i)
{
    if (!reg_active(a)) return;
    float wi, wu;
    RegScale::scales(a.reg_live, i - a.begin, wi, wu);
    sc.item = (double)wi;
    sc.user = (double)wu;
}

// PYX:648-649 after an interaction's update; avg = its average learning rate.
__device__ __forceinline__ void apply_scale_step(const FitArgs &a, Scales &sc, double avg, int lane)
{
    const double ia = a.item_alpha, ua = a.user_alpha;
    if (ia == 0.0 && ua == 0.0) return;  // exact no-op in the reference (scale *= 1.0)
    if (a.serial) {
        sc.item *= (1.0 + ia * avg);
        sc.user *= (1.0 + ua * avg);
    } else {
        sc.live.add(RegScale::log1p_f32((float)(ia * avg)), RegScale::log1p_f32((float)(ua * avg)));
    }
}

// warp_update, PYX:537-649: positive item row (g = -loss*u), negative item row
// (g = +loss*u), user row (g = loss*(neg - pos), float32 subtraction).
template <int NC>
__device__ __forceinline__ void warp_update(double loss, const FitArgs &a, int user, int pos,
                                            int neg, const Rep<NC> &U, const Rep<NC> &P,
                                            const Rep<NC> &N, Scales &sc, int lane)
{
    const bool atomic = !a.serial && a.update_mode == 0;
    double lrb[3], lrc[3][NC];
    bool pre_summed = false;
    float diff[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) diff[q] = __fsub_rn(N.v[q], P.v[q]);
    // Every address sees its read-modify-writes in the reference's order: a lane keeps
    // pos -> neg -> user for its coordinates, lane 0 does the same for the bias cells.
    if (a.serial) {
        update_row<NC>(a.itf, pos, 0, a.m, U.v, -loss, -loss, a.item_alpha, atomic, lane, lrb[0], lrc[0]);
        update_row<NC>(a.itf, neg, 0, a.m, U.v, loss, loss, a.item_alpha, atomic, lane, lrb[1], lrc[1]);
        update_row<NC>(a.usf, user, 1, a.m, diff, loss, loss, a.user_alpha, atomic, lane, lrb[2], lrc[2]);
    } else if constexpr (NC <= 2) {
        RowJob<NC> jobs[3] = {{a.itf, pos, 0, {}, -loss, -loss, a.item_alpha},
                              {a.itf, neg, 0, {}, loss, loss, a.item_alpha},
                              {a.usf, user, 1, {}, loss, loss, a.user_alpha}};
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            jobs[0].x[q] = U.v[q];
            jobs[1].x[q] = U.v[q];
            jobs[2].x[q] = diff[q];
        }
        const double lr_total = rows_update_parallel<NC, 3>(a.m, jobs, atomic, lane);
        lrb[0] = lr_total;  // already the wave-wide sum: see the tail below
        pre_summed = true;
    } else {
        update_row_batched<NC>(a.itf, pos, 0, a.m, U.v, -loss, -loss, a.item_alpha, atomic, lane, lrb[0], lrc[0]);
        update_row_batched<NC>(a.itf, neg, 0, a.m, U.v, loss, loss, a.item_alpha, atomic, lane, lrb[1], lrc[1]);
        update_row_batched<NC>(a.usf, user, 1, a.m, diff, loss, loss, a.user_alpha, atomic, lane, lrb[2], lrc[2]);
    }
    if (a.item_alpha != 0.0 || a.user_alpha != 0.0) {
        double avg = pre_summed ? lrb[0] : sum_lr<NC, 3>(lrb, lrc, a.m.d_real, a.serial != 0, lane);
        int cells = (a.m.d_real + 1) * (row_len(a.usf, user) + row_len(a.itf, pos) + row_len(a.itf, neg));
        avg /= (double)cells;
        apply_scale_step(a, sc, avg, lane);
    }
}

// update, PYX:454-534 (logistic): item row g = loss*u, user row g = loss*item.
template <int NC>
__device__ __forceinline__ void pair_update(double loss, const FitArgs &a, int user, int item,
                                            const Rep<NC> &U, const Rep<NC> &I, Scales &sc, int lane)
{
    const bool atomic = !a.serial && a.update_mode == 0;
    double lrb[3] = {0.0, 0.0, 0.0}, lrc[3][NC];
    bool pre_summed = false;
    if (a.serial) {
        update_row<NC>(a.itf, item, 0, a.m, U.v, loss, loss, a.item_alpha, atomic, lane, lrb[0], lrc[0]);
        update_row<NC>(a.usf, user, 1, a.m, I.v, loss, loss, a.user_alpha, atomic, lane, lrb[1], lrc[1]);
    } else if constexpr (NC <= 2) {
        RowJob<NC> jobs[2] = {{a.itf, item, 0, {}, loss, loss, a.item_alpha},
                              {a.usf, user, 1, {}, loss, loss, a.user_alpha}};
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            jobs[0].x[q] = U.v[q];
            jobs[1].x[q] = I.v[q];
        }
        const double lr_total = rows_update_parallel<NC, 2>(a.m, jobs, atomic, lane);
        lrb[0] = lr_total;  // already the wave-wide sum: see the tail below
        pre_summed = true;
    } else {
        update_row_batched<NC>(a.itf, item, 0, a.m, U.v, loss, loss, a.item_alpha, atomic, lane, lrb[0], lrc[0]);
        update_row_batched<NC>(a.usf, user, 1, a.m, I.v, loss, loss, a.user_alpha, atomic, lane, lrb[1], lrc[1]);
    }
    if (a.item_alpha != 0.0 || a.user_alpha != 0.0) {
        double avg = pre_summed ? lrb[0] : sum_lr<NC, 2>(lrb, lrc, a.m.d_real, a.serial != 0, lane);
        int cells = (a.m.d_real + 1) * (row_len(a.usf, user) + row_len(a.itf, item));
        avg /= (double)cells;
        apply_scale_step(a, sc, avg, lane);
    }
}

// warp_update for identity features on both sides with alpha == 0 (so every
// representation IS the embedding row), parallel mode: all loads are issued first
// (three G rows + the three bias cells, one per lane 0..2), then the maths, then the
// atomics -- one memory round trip instead of six.  Same arithmetic as warp_update.
template <int NC>
__device__ __forceinline__ void warp_update_identity(double loss, const FitArgs &a, int user,
                                                     int pos, int neg, const Rep<NC> &U,
                                                     const Rep<NC> &P, const Rep<NC> &N, int lane)
{
    const Hyper h{a.m.adadelta, a.m.lr, a.m.rho, a.m.eps};
    const int d = a.m.d, um = a.update_mode;
    const size_t bp = (size_t)pos * d, bn = (size_t)neg * d, bu = (size_t)user * d;
    float *Wi = a.m.W[0], *Gi = a.m.G[0], *Mi = a.m.M[0];
    float *Wu = a.m.W[1], *Gu = a.m.G[1], *Mu = a.m.M[1];
    float gP[NC], gN[NC], gU[NC], mP[NC], mN[NC], mU[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        int c = lane + WAVE * q;
        bool ok = c < d;
        gP[q] = ok ? ldw(Gi + bp + c) : 1.0f;
        gN[q] = ok ? ldw(Gi + bn + c) : 1.0f;
        gU[q] = ok ? ldw(Gu + bu + c) : 1.0f;
        mP[q] = (ok && h.adadelta) ? ldw(Mi + bp + c) : 0.0f;
        mN[q] = (ok && h.adadelta) ? ldw(Mi + bn + c) : 0.0f;
        mU[q] = (ok && h.adadelta) ? ldw(Mu + bu + c) : 0.0f;
    }
    // bias cells: lane 0 = positive item (g = -loss), 1 = negative item, 2 = user (PYX:571-599)
    const int side = lane == 2 ? 1 : 0;
    const int brow = lane == 0 ? pos : (lane == 1 ? neg : user);
    float *bW = a.m.b[side] + brow, *bG = a.m.bG[side] + brow, *bM = a.m.bM[side] + brow;
    float obW = 0.0f, obG = 1.0f, obM = 0.0f;
    if (lane < 3) {
        obW = ldw(bW);
        obG = ldw(bG);
        if (h.adadelta) obM = ldw(bM);
    }
    float nW, nG, nM;
    double lr;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        int c = lane + WAVE * q;
        if (c < d) {
            double u = (double)U.v[q];
            cell_math(P.v[q], gP[q], mP[q], 1.0, -loss * u, h, 0.0, nW, nG, nM, lr);
            publish_cell(Wi + bp + c, Gi + bp + c, Mi + bp + c, P.v[q], gP[q], mP[q], nW, nG, nM, 1.0, -loss * u,
                         h, 0.0, um);
            cell_math(N.v[q], gN[q], mN[q], 1.0, loss * u, h, 0.0, nW, nG, nM, lr);
            publish_cell(Wi + bn + c, Gi + bn + c, Mi + bn + c, N.v[q], gN[q], mN[q], nW, nG, nM, 1.0, loss * u,
                         h, 0.0, um);
            double df = (double)__fsub_rn(N.v[q], P.v[q]);
            cell_math(U.v[q], gU[q], mU[q], 1.0, loss * df, h, 0.0, nW, nG, nM, lr);
            publish_cell(Wu + bu + c, Gu + bu + c, Mu + bu + c, U.v[q], gU[q], mU[q], nW, nG, nM, 1.0, loss * df,
                         h, 0.0, um);
        }
    }
    if (lane < 3) {
        const double gb = lane == 0 ? -loss : loss;
        cell_math(obW, obG, obM, 1.0, gb, h, 0.0, nW, nG, nM, lr);
        publish_cell(bW, bG, bM, obW, obG, obM, nW, nG, nM, 1.0, gb, h, 0.0, um);
    }
}

}  // namespace lfm
