// session.hip -- host side of liblfm_hip.so: the C ABI of include/lfm_hip.h.
//
// Device-resident state for one model on one GPU (weights, feature CSRs, training
// COO, shuffle slots), the epoch driver that replaces the prange loops of
// _lightfm_fast.pyx.template (PYX:719-724, 819-825, 951-957, 1107-1113), and the
// one-shot entry points that mirror the Cython functions one for one.
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/lfm_hip.h"
#include "device.hpp"
#include "kernels.hpp"
#include "pool.hpp"

using namespace lfm;

// ------------------------------------------------------------------ errors ---

static thread_local std::string g_err;

static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? LFM_ENOMEM : LFM_ENODEV,               \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

#define LFM_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != LFM_OK) return rc_; \
    } while (0)

extern "C" const char *lfm_last_error(void) { return g_err.c_str(); }

static thread_local float g_kernel_ms = 0.0f;
extern "C" float lfm_last_kernel_ms(void) { return g_kernel_ms; }

// Host-side input scan of LightFM.fit_partial (LFM:383-386, 447-472 and 617-625 test the interaction values for "all
// ones" and for finiteness with two or three numpy passes over 80 MB at the ML-20M shape -- a quarter of a 10-epoch
// fit once the epochs run on the GPU): ONE multi-threaded pass.  *all_ones = every value == 1.0f; *finite = every
// value finite and their float64 sum inside the float32 range (the reference's test is isfinite(float32 sum)).
extern "C" int lfm_host_scan_f32(const float *p, int64_t n, int32_t *all_ones, int32_t *finite)
{
    if (n < 0 || (n && !p) || !all_ones || !finite) return fail(LFM_EINVAL, "bad scan arguments");
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())),
                                                            n / (1 << 20)));
    std::vector<double> sums((size_t)T, 0.0);
    std::vector<int> ones((size_t)T, 1), fin((size_t)T, 1);
    auto work = [&](int t) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        double acc = 0.0;
        uint32_t not_one = 0u, bad = 0u;
        for (int64_t j = lo; j < hi; ++j) {
            const float v = p[j];
            uint32_t bits;
            memcpy(&bits, &v, 4);
            not_one |= bits ^ 0x3f800000u;
            bad |= ((bits & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;  // inf or nan
            acc += (double)v;
        }
        sums[(size_t)t] = acc;
        ones[(size_t)t] = not_one == 0u;
        fin[(size_t)t] = bad == 0u;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    double total = 0.0;
    int o = 1, f = 1;
    for (int t = 0; t < T; ++t) {
        total += sums[(size_t)t];
        o &= ones[(size_t)t];
        f &= fin[(size_t)t];
    }
    *all_ones = o;
    *finite = (f && fabs(total) <= 3.4028234663852886e38) ? 1 : 0;
    return LFM_OK;
}

// Host-side signature of a weight array (LightFM._scoring_session re-validates its resident tables against the host arrays
// before every predict / predict_rank): sum of word[i] * (2 i + 1) modulo 2^64 over the array's 32-bit words -- exact (any
// single-word edit changes it) and position-sensitive (row / column moves change it) -- by a few host threads.  Round 4 used
// a numpy bit-pattern sum plus a BLAS matrix-vector probe; on a 256-core box the BLAS call woke every core, which under a
// container CPU quota (16 CPUs per 100 ms on the GPU boxes) throttled the process: every other predict_rank call stalled by
// 75 ms (profiles/r05_visit_k.txt).
extern "C" int lfm_host_checksum_u32(const uint32_t *p, int64_t n, uint64_t *out)
{
    if (n < 0 || (n && !p) || !out) return fail(LFM_EINVAL, "bad checksum arguments");
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n >= (8 << 20) ? 8 : 4, n / (1 << 20)));  // (well inside a 16-CPU quota)
    std::vector<uint64_t> part((size_t)T, 0ull);
    auto work = [&](int t) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        uint64_t acc = 0ull;
        for (int64_t j = lo; j < hi; ++j) acc += (uint64_t)p[j] * (2ull * (uint64_t)j + 1ull);
        part[(size_t)t] = acc;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    uint64_t total = 0ull;
    for (int t = 0; t < T; ++t) total += part[(size_t)t];
    *out = total;
    return LFM_OK;
}

// Host-side initialisation of an embedding table (LFM:281-312: ((random_state.rand(rows, d) - 0.5) / d).astype(float32)): the
// Mersenne Twister of numpy's legacy RandomState restated -- block regeneration, tempering and the 53-bit double of two
// outputs ((a >> 5) * 2^26 + (b >> 6)) / 2^53 -- with the subtraction, the division and the float32 cast fused into the
// pass, on the generator state the caller hands in and gets back (RandomState.get_state / set_state): the same values
// and the same stream position as numpy, bit for bit (tests/test_host_logic.py), at a quarter of its time.  The 8.9 M
// draws of the ML-20M user table were 40 of the 210 ms of a 10-epoch LightFM.fit.
extern "C" int lfm_host_mt19937_table(uint32_t *key, int32_t *pos_io, float *out, int64_t n, int32_t d)
{
    if (!key || !pos_io || (n && !out) || n < 0 || d <= 0 || *pos_io < 0 || *pos_io > 624) return fail(LFM_EINVAL, "bad generator arguments");

    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    int pos = *pos_io;
    uint32_t t[624];  // tempered outputs of the current block
    auto regen = [&]() {
        uint32_t *mt = key;
        int kk;
        for (kk = 0; kk < 624 - 397; ++kk) {
            uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & MATRIX_A);
        }
        for (; kk < 623; ++kk) {
            uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & MATRIX_A);
        }
        uint32_t y = (mt[623] & UPPER) | (mt[0] & LOWER);
        mt[623] = mt[396] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & MATRIX_A);
    };
    auto temper_block = [&]() {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = key[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            t[i] = y;
        }
    };
    const double dd = (double)d;
    int64_t i = 0;
    bool have = false;
    while (i < n) {
        if (pos >= 624) { regen(); pos = 0; have = false; }
        if (!have) { temper_block(); have = true; }
        // pairs inside this block
        int avail = 624 - pos;
        if (avail >= 2) {
            int64_t pairs = avail / 2;
            if (pairs > n - i) pairs = n - i;
            const uint32_t *tp = t + pos;
            for (int64_t j = 0; j < pairs; ++j) {
                const double x = ((double)(tp[2 * j] >> 5) * 67108864.0 + (double)(tp[2 * j + 1] >> 6)) / 9007199254740992.0;
                out[i + j] = (float)((x - 0.5) / dd);
            }
            i += pairs;
            pos += 2 * (int)pairs;
        } else {  // one output left in the block: the pair straddles two blocks
            uint32_t a = t[pos] >> 5;
            regen(); pos = 0; temper_block();
            uint32_t b = t[pos++] >> 6;
            const double x = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
            out[i++] = (float)((x - 0.5) / dd);
        }
    }
    *pos_io = pos;
    return LFM_OK;
}

extern "C" int lfm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int lfm_device_info(int device, char *name, int32_t *cus, int64_t *hbm_bytes)
{
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name) {
        strncpy(name, p.name[0] ? p.name : p.gcnArchName, 255);  // some boxes report an empty name
        name[255] = 0;
    }
    if (cus) *cus = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return LFM_OK;
}

// Releases the library's cached (unused) device memory to the HIP runtime (csrc/pool.hpp); returns
// the number of bytes released through *released.  Sessions in use are not affected.
extern "C" int lfm_device_trim(int64_t *released)
{
    (void)hipDeviceSynchronize();
    const size_t b = DevPool::instance().trim();
    if (released) *released = (int64_t)b;
    return LFM_OK;
}

// bytes the pool holds from the runtime / the part of them not handed out at the moment
extern "C" int lfm_device_pool_stats(int64_t *reserved, int64_t *cached)
{
    size_t r = 0, c = 0;
    DevPool::instance().stats(&r, &c);
    if (reserved) *reserved = (int64_t)r;
    if (cached) *cached = (int64_t)c;
    return LFM_OK;
}

// ------------------------------------------------------------ device memory ---

// Allocation flavour of the weight tables: 0 = hipMalloc (coarse-grained: an XCD's L2 may serve
// lines another XCD has rewritten until the launch ends), 1 = hipDeviceMallocFinegrained,
// 3 = hipDeviceMallocUncached (default for models beyond the L2s' capacity): 14 % faster on the
// bench workload (0.84 vs 0.72 G interactions/s), same measured precision@10, and reads are never
// stale.  The gain comes from the small, hot bias tables (mask ablation): their lines
// are what the atomics of other wavefronts keep dropping from every XCD's L2.
static int table_alloc_flags()
{
    static int f = -1;
    if (f < 0) {
        const char *e = getenv("LIGHTFM_AMD_TABLE_ALLOC");
        f = e ? atoi(e) : 3;
    }
    return f;
}

// which of the six table kinds (W, G, M, b, bG, bM = bits 0..5) get that flavour (experiment knob)
static int table_alloc_mask()
{
    static int m = -1;
    if (m < 0) {
        const char *e = getenv("LIGHTFM_AMD_TABLE_ALLOC_MASK");
        m = e ? atoi(e) : 63;
    }
    return m;
}

// LIGHTFM_AMD_TRACE=1 (debugging): every device allocation / release and every epoch launch is
// written to stderr, so that a faulting address reported by the runtime can be attributed.
static bool trace_enabled()
{
    static const bool on = [] {
        const char *e = getenv("LIGHTFM_AMD_TRACE");
        return e && atoi(e) != 0;
    }();
    return on;
}

// LIGHTFM_AMD_VALIDATE (debugging): 1 = checksums of the read-only device inputs around every epoch
// (validate_inputs); 2 = additionally every uploaded buffer keeps a host-side shadow copy and is read
// back and compared with it before every epoch (DBuf::verify): a mismatch names the buffer, the first
// differing word and what the device holds instead.
static int validate_level()
{
    static const int lvl = [] {
        const char *e = getenv("LIGHTFM_AMD_VALIDATE");
        return e ? atoi(e) : 0;
    }();
    return lvl;
}

template <typename T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    int flags = 0;  // hipExtMallocWithFlags flags (0 = plain hipMalloc)
    std::vector<unsigned char> shadow;  // LIGHTFM_AMD_VALIDATE=2: what was uploaded last
    ~DBuf() { release(); }
    void release()
    {
        if (p) {
            if (trace_enabled()) fprintf(stderr, "LFM_FREE  %p %zu\n", (void *)p, n * sizeof(T));
            pool_free(p);  // pool.hpp: no synchronisation -- the owner has waited for the work that used it
        }
        p = nullptr;
        n = 0;
        shadow.clear();
    }
    int alloc(size_t count)
    {
        if (count == n && p) return LFM_OK;
        // a buffer that changes size goes back to the pool: nothing in flight may still use it
        if (p) (void)hipDeviceSynchronize();
        release();
        if (count == 0) return LFM_OK;
        hipError_t e = pool_alloc((void **)&p, count * sizeof(T), flags);
        if (e != hipSuccess) return fail(LFM_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        n = count;
        if (trace_enabled()) fprintf(stderr, "LFM_ALLOC %p %zu flags %d\n", (void *)p, n * sizeof(T), flags);
        return LFM_OK;
    }
    // capacity semantics (buffers whose size changes from call to call): grows, never shrinks
    int reserve(size_t count)
    {
        if (p && n >= count) return LFM_OK;
        return alloc(std::max(count, n + n / 2));
    }
    int upload(const T *src, size_t count)
    {
        LFM_TRY(alloc(count));
        if (count) HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
        if (validate_level() >= 2) {
            shadow.assign((const unsigned char *)src, (const unsigned char *)src + count * sizeof(T));
            LFM_TRY(verify("right after its upload"));
        }
        return LFM_OK;
    }
    // LIGHTFM_AMD_VALIDATE=2: the device contents against the shadow copy of the last upload
    int verify(const char *when) const
    {
        if (shadow.empty() || !p || shadow.size() != n * sizeof(T)) return LFM_OK;
        std::vector<unsigned char> back(shadow.size());
        HIP_TRY(hipMemcpy(back.data(), p, back.size(), hipMemcpyDeviceToHost));
        if (memcmp(back.data(), shadow.data(), back.size()) == 0) return LFM_OK;
        const uint32_t *b = (const uint32_t *)back.data(), *e = (const uint32_t *)shadow.data();
        const size_t words = back.size() / 4;
        size_t first = words, last = 0, bad = 0;
        for (size_t i = 0; i < words; ++i)
            if (b[i] != e[i]) {
                if (first == words) first = i;
                last = i;
                ++bad;
            }
        char msg[512];
        snprintf(msg, sizeof(msg), "LIGHTFM_AMD_VALIDATE: device buffer %p (%zu bytes) differs from what was uploaded, %s: %zu of %zu "
                 "words, first at word %zu (expected %08x %08x %08x %08x, device holds %08x %08x %08x %08x), last at word %zu",
                 (void *)p, back.size(), when, bad, words, first, first < words ? e[first] : 0u, first + 1 < words ? e[first + 1] : 0u,
                 first + 2 < words ? e[first + 2] : 0u, first + 3 < words ? e[first + 3] : 0u, first < words ? b[first] : 0u,
                 first + 1 < words ? b[first + 1] : 0u, first + 2 < words ? b[first + 2] : 0u, first + 3 < words ? b[first + 3] : 0u, last);
        fprintf(stderr, "%s\n", msg);
        return fail(LFM_ECORRUPT, msg);
    }
    int download(T *dst) const
    {
        if (n) HIP_TRY(hipMemcpy(dst, p, n * sizeof(T), hipMemcpyDeviceToHost));
        return LFM_OK;
    }
};

struct DevCsr {
    DBuf<int32_t> indices, indptr;
    DBuf<float> data;
    int32_t rows = 0, cols = 0;
    int64_t nnz = 0;
    bool identity = false;

    static bool is_identity(const lfm_csr *m)
    {
        if (m->nnz != m->rows || m->cols < m->rows) return false;
        for (int32_t i = 0; i < m->rows; ++i)
            if (m->indptr[i] != i || m->indices[i] != i || m->data[i] != 1.0f) return false;
        return m->indptr[m->rows] == m->rows;
    }
    int upload(const lfm_csr *m, bool detect_identity, bool need_data)
    {
        rows = m->rows;
        cols = m->cols;
        nnz = m->nnz;
        identity = detect_identity && is_identity(m);
        if (identity) {  // never dereferenced on device
            indices.release();
            indptr.release();
            data.release();
            return LFM_OK;
        }
        LFM_TRY(indptr.upload(m->indptr, (size_t)m->rows + 1));
        LFM_TRY(indices.upload(m->indices, (size_t)m->nnz));
        if (need_data) LFM_TRY(data.upload(m->data, (size_t)m->nnz));
        return LFM_OK;
    }
    void clear()
    {
        indices.release();
        indptr.release();
        data.release();
        rows = cols = 0;
        nnz = 0;
        identity = false;
    }
    DCsr view() const { return DCsr{indices.p, indptr.p, data.p, rows, cols, identity ? 1 : 0}; }
};

// Per-call device buffers go back to the pool when their DBuf leaves scope -- on every path, error
// returns included -- and the pool does not synchronise: declare one of these AFTER the buffers (it
// is then destroyed BEFORE them) so the stream has drained by the time they are released.
struct DrainOnExit {
    hipStream_t st;
    bool device_wide;
    explicit DrainOnExit(hipStream_t s, bool all = false) : st(s), device_wide(all) {}
    ~DrainOnExit()
    {
        if (device_wide) (void)hipDeviceSynchronize();
        else (void)hipStreamSynchronize(st);
    }
};

static int validate_csr(const lfm_csr *m, const char *what)
{
    if (!m) return fail(LFM_EINVAL, std::string(what) + ": null matrix");
    if (m->rows < 0 || m->cols < 0 || m->nnz < 0) return fail(LFM_EINVAL, std::string(what) + ": negative size");
    if (!m->indptr || (m->nnz && (!m->indices || !m->data)))
        return fail(LFM_EINVAL, std::string(what) + ": null buffer");
    if (m->indptr[0] != 0 || (int64_t)m->indptr[m->rows] != m->nnz)
        return fail(LFM_EINVAL, std::string(what) + ": indptr does not span nnz");
    return LFM_OK;
}

// ------------------------------------------------------------------ RCCL ---

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        // The RCCL that belongs to the HIP runtime THIS library runs on: next to the libamdhip64 that provides our hip* symbols, by
        // absolute path.  A process that has imported torch (bench.py and DistributedFit use torch.distributed's gloo backend for
        // the rendezvous) already holds torch's own bundled librccl.so -- linked to torch's own copy of the HIP runtime, in which no
        // device of ours is initialised: dlopen("librccl.so") returned THAT one and ncclCommInitRank failed with "no ROCm-capable
        // device is detected" (round 5, tools/rccl_with_torch_probe.py; rounds 2-4 had only ever initialised RCCL in processes
        // without torch).  LIGHTFM_AMD_RCCL=<path> overrides.
        std::vector<std::string> names;
        if (const char *e = getenv("LIGHTFM_AMD_RCCL")) names.push_back(e);
        Dl_info info;
        if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash);
                names.push_back(dir + "/librccl.so.1");
                names.push_back(dir + "/librccl.so");
            }
        }
        names.push_back("/opt/rocm/lib/librccl.so.1");
        names.push_back("/opt/rocm/lib/librccl.so");
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        for (const std::string &n : names) {
            r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.lib) {
                if (trace_enabled()) fprintf(stderr, "LFM_RCCL %s\n", n.c_str());
                break;
            }
        }
        if (r.lib) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
            r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
            r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        }
    }
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) return nullptr;
    return &r;
}

#define NCCL_TRY(expr)                                                                     \
    do {                                                                                   \
        ncclResult_t r_ = (expr);                                                          \
        if (r_ != ncclSuccess)                                                             \
            return fail(LFM_ECOMM, std::string(#expr) + ": " +                             \
                                       (rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "rccl error")); \
    } while (0)

// ---------------------------------------------------------------- session ---

struct lfm_session {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int cus = 256;

    // model: [side][kind] with kind 0..5 = W,G,M,b,bG,bM ; side 0 item, 1 user
    DBuf<float> tab[2][6];
    int32_t n_feat[2] = {0, 0};
    int32_t d = 0, adadelta = 0, max_sampled = 0;  // d: floats per embedding row ON THE DEVICE (a multiple of 4, see create_session)
    int32_t d_host = 0;                             // no_components: the row length of the caller's arrays
    float lr = 0, rho = 0, eps = 0;
    bool scoring_only = false;  // lfm_session_create_scoring: only W and b of both sides are resident
    DBuf<double> scales;      // [2]
    DBuf<double> reg_log;     // [2] parallel mode: log of the regularisation scales at the last launch boundary
    bool reg_rate_known = false;  // a regularised launch of this session has measured the scales' growth per position
    DBuf<float> reg_live;     // [RegScale::FLOATS] ... and their live state (device.hpp: RegScale), uncached memory
    DBuf<unsigned long long> counters;
    DBuf<uint32_t> seeds;
    DBuf<double> logtab;
    DBuf<int> flag;

    DevCsr itf, usf, pos;
    // predict_ranks: the train matrix of the last call stays resident (the same 72 MB of ML-20M train indices were uploaded
    // by every precision_at_k / auc_score call: 3 of a call's 12 ms); a call re-validates it with the position-sensitive
    // checksum of its index arrays (lfm_host_checksum_u32) and uploads only what changed
    DevCsr train_keep;
    uint64_t train_sig[2] = {0, 0};
    bool train_keep_valid = false;
    DBuf<int32_t> user_ids, item_ids;
    DBuf<float> Y, weight;
    bool weight_aliases_Y = false;
    DBuf<float> bias_snap[2][2];  // per-launch cached copies of the bias tables (tile kernel scoring): [side][launch parity]
    // Bias PAIRS (steady-state tile kernels): while launches of fit_warp_tile_ahead_kernel / fit_warp_tile_narrow_kernel run, the
    // live bias cells of a side are ONE table of (b, bG) pairs -- both cells of a row in one line, published by one line
    // operation (C2 +4.4 % in the timing experiment, profiles/r06_narrow_kernel.txt) -- packed from tab[side][3 / 4] before
    // the first such launch of an epoch call and unpacked into them before anything else reads the tables
    DBuf<float> bias_pairs[2];
    bool pairs_live = false;
    // Row PAIRS (narrow-model kernel, d <= 16): likewise W and G of a side as one table of 128-byte rows [W(16) | G(16)]
    // (d <= 12: with the bias cells in slot d of each half -- rows_bias; lfm_opts.plan_flags bit 7)
    DBuf<float> row_pairs[2];
    bool rows_live = false, rows_bias = false;
    hipStream_t stream2 = nullptr;  // full-residency launches alternate between `stream` and this one (see lfm_session_epoch)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DBuf<int4> recs;  // AoS copy of (user_ids, item_ids, Y, weight) for warp_tile.hip, built on demand
    ItemShards shards = {};  // owner-sharded item tables (lfm_sessions_share_items_local / _ipc); n = 0: this session's own tables
    // Lifetime of a sharing: the members of a one-process group hold one ShareGroup; a member that is destroyed (or whose
    // tables are re-uploaded) marks it broken and the others' epochs then fail instead of addressing freed memory.
    struct ShareGroup { bool broken = false; };
    std::shared_ptr<ShareGroup> share;
    int share_rank = -1, share_k = 0;       // this session's range of the sharded item rows
    std::vector<void *> ipc_mappings;       // lfm_session_share_items_ipc: the peers' allocations mapped into this process
    DBuf<uint32_t> bloom;  // Bloom filter over the rows of `pos` (device.hpp: Bloom), built with the lookup
    bool bloom_valid = false;
    bool recs_valid = false;
    int64_t n = 0;
    // The hot set of the item side (hot_slices.hip; device.hpp: HotRec): feature rows shared by so many items that an
    // interaction touches each with probability >= 1/512 (a hybrid model's tag / genre rows) are accumulated in LDS slices
    // between launches instead of by float atomics of every interaction.  Derived from the resident feature matrix by the
    // first epoch that qualifies (build_hot_set).
    struct HotSet {
        int state = 0;       // 0 = not looked at yet, 1 = in use, -1 = none (identity features, no shared rows, does not pay)
        int n = 0, cs = 0;   // hot rows; components per LDS slice
        DBuf<int32_t> slot;  // [n_item_feat] slot of a hot row, -1 otherwise
        DBuf<int32_t> rows;  // [n] feature row of a slot, ascending
        DBuf<float> snapW, snapG, snapb, snapbG;
        DBuf<HotRec> rec[2];  // one record per position of a launch; two sets: the slice kernel of launch k runs under launch k + 1
        DBuf<float> x[2];     // [positions][d]
        hipEvent_t ev_p1[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};  // records written / records applied
        bool pending[2] = {false, false};
        double share = 0.0;  // hot entries / all entries of the feature matrix
    } hot;
    double user_pair_share = -1.0;  // sum_u c_u^2 / n^2 of the uploaded COO (c_u = interactions of user u); < 0: not computed yet
    std::vector<DBuf<int32_t> *> shuffles;
    // lfm_session_device_shuffle_ahead: the permutation of the NEXT epoch is written on a stream of its own while the current
    // epoch's kernels run; the epoch that uses the slot waits for the event
    hipStream_t aux_stream = nullptr;
    std::vector<hipEvent_t> shuffle_ready;  // per slot; nullptr = nothing pending
    DBuf<int32_t> neg_log, sampled_log;

    // multi-GPU
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    DBuf<float> snap[2][6];     // the tables at the start of the current merge interval
    int snap_sides = 0;         // bit 0 item, bit 1 user: which sides have a valid snapshot
    DBuf<float> scratch[2];     // merge temporaries (ADAGRAD mode: this rank's dG; local reduce: sums)
    // sparse merge (merge_group_sparse): the rows that differ from the interval's snapshot (a byte map per
    // side, filled at merge time), and the exchange that may still be in flight
    DBuf<unsigned char> dirty[2];
    DBuf<int32_t> hot_ids[2];   // rows merged at the short cadence (lfm_session_set_hot_rows), ascending
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_pack = nullptr, ev_comm = nullptr;
    struct PendingMerge {
        bool active = false, on_comm_stream = false;
        int sides = 0;
        float wscale = 1.0f, ascale = 1.0f;  // what the summed deltas of the weights / of the accumulators are scaled by
        int adagrad_mode = 0;                // LFM_MERGE_ADAGRAD: the weight plane carries numerators (FusedPlan)
        int64_t n_u[2] = {0, 0};
        DBuf<int32_t> ids[2];     // the union of touched rows, ascending (unused when the merge covers every row)
        bool all_rows[2] = {false, false};  // this merge covers every row of the side: no id list
        DBuf<float> sum[2];       // packed deltas of all kinds, one buffer (session.hip: FusedPlan): this rank's, then the sum over ranks
        DBuf<float> loc[2];       // this rank's own packed deltas (unscaled)
    } pend;
    // A merge whose union covered >= merge_dense_frac of a side's rows makes the NEXT merges of that side skip the detection,
    // the OR-all-reduce of the byte maps and the compaction: every row travels (bit-identical result: an untouched row's
    // deltas are zeros).  The union is the same on every rank, so every rank takes the same path.
    bool merge_all_rows[2] = {false, false};
    int merge_all_uses[2] = {0, 0};  // merges of the side that ran on the latch since the rows were last detected
    float merge_dense_frac = 0.9f;

    // LIGHTFM_AMD_VALIDATE=1 (debugging): checksums of the read-only device inputs at the end of the
    // previous epoch, by (address, bytes): see validate_inputs()
    struct GuardSum { const void *p; size_t bytes; unsigned long long sum; };
    std::vector<GuardSum> guard_sums;
    DBuf<unsigned long long> guard_dev;
    int64_t epochs_run = 0;

    ~lfm_session()
    {
        // the buffers go back to the pool without any implicit synchronisation (hipFree had one)
        if (stream) (void)hipStreamSynchronize(stream);
        if (share) share->broken = true;  // the other members must not train against this session's tables any more
        for (void *m : ipc_mappings) (void)hipIpcCloseMemHandle(m);
        for (auto *s : shuffles) delete s;
        if (stream2) {
            (void)hipStreamSynchronize(stream2);
            (void)hipStreamDestroy(stream2);
        }
        if (aux_stream) {
            (void)hipStreamSynchronize(aux_stream);
            (void)hipStreamDestroy(aux_stream);
        }
        for (hipEvent_t e : shuffle_ready)
            if (e) (void)hipEventDestroy(e);
        for (int i = 0; i < 2; ++i) {
            if (hot.ev_p1[i]) (void)hipEventDestroy(hot.ev_p1[i]);
            if (hot.ev_done[i]) (void)hipEventDestroy(hot.ev_done[i]);
        }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (comm_stream) (void)hipStreamSynchronize(comm_stream);
        if (comm && rccl()) rccl()->CommDestroy(comm);
        if (comm_stream) (void)hipStreamDestroy(comm_stream);
        if (ev_pack) (void)hipEventDestroy(ev_pack);
        if (ev_comm) (void)hipEventDestroy(ev_comm);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }

    DModel dmodel()
    {
        DModel m;
        for (int s = 0; s < 2; ++s) {
            m.W[s] = tab[s][0].p;
            m.G[s] = tab[s][1].p;
            m.M[s] = tab[s][2].p;
            m.b[s] = tab[s][3].p;
            m.bG[s] = tab[s][4].p;
            m.bM[s] = tab[s][5].p;
            m.n_feat[s] = n_feat[s];
        }
        m.d = d;
        m.d_real = d_host;
        m.adadelta = adadelta;
        m.lr = lr;
        m.rho = rho;
        m.eps = eps;
        m.max_sampled = max_sampled;
        m.scales = scales.p;
        return m;
    }
};

static float *host_tab(const lfm_model *m, int side, int kind)
{
    float *const t[2][6] = {
        {m->item_W, m->item_G, m->item_M, m->item_b, m->item_bG, m->item_bM},
        {m->user_W, m->user_G, m->user_M, m->user_b, m->user_bG, m->user_bM}};
    return t[side][kind];
}

static size_t tab_count(const lfm_session *s, int side, int kind)
{
    return kind < 3 ? (size_t)s->n_feat[side] * s->d : (size_t)s->n_feat[side];
}

static bool kind_used(const lfm_session *s, int kind)
{
    if (s->scoring_only) return kind == 0 || kind == 3;  // scoring reads embeddings and biases only
    return s->adadelta || (kind != 2 && kind != 5);
}

// Embedding tables between the caller's [n_feat, no_components] arrays and the device's [n_feat, d] rows.  d =
// no_components rounded up to a multiple of 4 floats, so that EVERY width runs the 16-byte-vectorised production
// kernels (the reference's default is no_components = 10, LFM:191).  A padded component is 0 in W (and in adadelta's
// M) and 1 in G: its product in a score is +0 (the sequential float32 sum is unchanged), every gradient of it is
// loss * 0, so W stays 0 and G stays 1 whatever the schedule and the L2 penalty (0 * (1 + alpha lr) = 0); the
// learning-rate average of the regularisation scales counts no_components cells (DModel::d_real).  The arrays that
// come back are the caller's shape again.
static int upload_table(lfm_session *s, int side, int kind, const float *host)
{
    DBuf<float> &t = s->tab[side][kind];
    if (kind >= 3 || s->d == s->d_host) return t.upload(host, tab_count(s, side, kind));
    const size_t rows = (size_t)s->n_feat[side];
    LFM_TRY(t.alloc(rows * s->d));
    if (!rows) return LFM_OK;
    if (kind == 1) HIP_TRY(hipMemsetD32(t.p, 0x3f800000, rows * s->d));  // 1.0f
    else HIP_TRY(hipMemset(t.p, 0, rows * s->d * sizeof(float)));
    HIP_TRY(hipMemcpy2D(t.p, (size_t)s->d * sizeof(float), host, (size_t)s->d_host * sizeof(float),
                        (size_t)s->d_host * sizeof(float), rows, hipMemcpyHostToDevice));
    return LFM_OK;
}

static int download_table(lfm_session *s, int side, int kind, float *host)
{
    DBuf<float> &t = s->tab[side][kind];
    if (kind >= 3 || s->d == s->d_host) return t.download(host);
    const size_t rows = (size_t)s->n_feat[side];
    if (rows)
        HIP_TRY(hipMemcpy2D(host, (size_t)s->d_host * sizeof(float), t.p, (size_t)s->d * sizeof(float),
                            (size_t)s->d_host * sizeof(float), rows, hipMemcpyDeviceToHost));
    return LFM_OK;
}

static int validate_model(const lfm_model *m, bool scoring = false)
{
    if (!m) return fail(LFM_EINVAL, "null model");
    if (m->d <= 0) return fail(LFM_EINVAL, "no_components must be positive");
    if (m->d > LFM_MAX_COMPONENTS) return fail(LFM_EUNSUPPORTED, "no_components > 1024 is not supported by the HIP backend");
    if (m->n_item_feat < 0 || m->n_user_feat < 0) return fail(LFM_EINVAL, "negative feature count");
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k < 6; ++k)
            if (!host_tab(m, s, k) && (s == 0 ? m->n_item_feat : m->n_user_feat) > 0 && (!scoring || k == 0 || k == 3))
                return fail(LFM_EINVAL, "null weight array");
    return LFM_OK;
}

// ids that index device tables are range-checked on the device at upload: an out-of-range id would
// otherwise fault the GPU inside an epoch kernel, and a GPU memory fault aborts the host process
// (the reference reads out of bounds in the same situation, PYX:828-829).
__global__ void id_range_kernel(const int32_t *p, int64_t n, int32_t *lohi)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    int32_t lo = 0x7fffffff, hi = (int32_t)0x80000000;
    for (int64_t j = t; j < n; j += st) {
        const int32_t v = p[j];
        lo = min(lo, v);
        hi = max(hi, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off, WAVE));
        hi = max(hi, __shfl_xor(hi, off, WAVE));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0 && lo <= hi) {
        atomicMin(lohi, lo);
        atomicMax(lohi + 1, hi);
    }
}

static int check_id_range(lfm_session *s, const int32_t *dev, int64_t n, int64_t limit, const char *what)
{
    if (n <= 0 || !dev) return LFM_OK;
    DBuf<int32_t> lohi;
    DrainOnExit drain(s->stream);
    const int32_t init[2] = {0x7fffffff, (int32_t)0x80000000};
    LFM_TRY(lohi.alloc(2));
    HIP_TRY(hipMemcpyAsync(lohi.p, init, sizeof(init), hipMemcpyHostToDevice, s->stream));
    id_range_kernel<<<(int)std::min<int64_t>(2048, (n + 255) / 256), 256, 0, s->stream>>>(dev, n, lohi.p);
    HIP_TRY(hipGetLastError());
    int32_t got[2];
    HIP_TRY(hipMemcpyAsync(got, lohi.p, sizeof(got), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (got[0] < 0 || (int64_t)got[1] >= limit) {
        char msg[256];
        snprintf(msg, sizeof(msg), "%s out of range: values span [%d, %d], valid is [0, %lld)", what, got[0], got[1], (long long)limit);
        return fail(LFM_EINVAL, msg);
    }
    return LFM_OK;
}

static int create_session(lfm_session **out, int device, const lfm_model *model, const lfm_csr *item_features,
                          const lfm_csr *user_features, bool scoring)
{
    if (!out) return fail(LFM_EINVAL, "null out pointer");
    *out = nullptr;
    LFM_TRY(validate_model(model, scoring));
    LFM_TRY(validate_csr(item_features, "item_features"));
    LFM_TRY(validate_csr(user_features, "user_features"));
    if (item_features->cols > model->n_item_feat || user_features->cols > model->n_user_feat)
        return fail(LFM_EINVAL, "feature matrix has more columns than there are embeddings");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(LFM_ENODEV, "no HIP device available (the HIP backend has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(LFM_EINVAL, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    lfm_session *s = new lfm_session();
    s->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->cus = prop.multiProcessorCount;
    int rc = LFM_OK;
    auto guard = [&](int r) { if (r != LFM_OK && rc == LFM_OK) rc = r; };
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) {
        delete s;
        return fail(LFM_ENODEV, "cannot create HIP stream/events");
    }
    s->n_feat[0] = model->n_item_feat;
    s->n_feat[1] = model->n_user_feat;
    s->d_host = model->d;
    s->d = (model->d + 3) / 4 * 4;
    // Training sessions: rows of 17..63 floats are padded on to 32 / 48 / 64, so that no row straddles more 128-byte lines than
    // its length needs (an 80-byte row at a 16-byte-aligned offset touches 1.6 lines on average, a 224-byte row 2.5, a 128- or
    // 192- or 256-byte row at its own stride exactly 1 / 2 / 2): the tile kernels are bound by line operations, not bytes (WARP
    // at the ML-20M shape: d = 20 12.4 -> 11.8 ms per epoch, d = 40 13.5 -> 13.3, d = 56 15.0 -> 13.8).  The row-stream kernels
    // pay for every padded component's cell arithmetic: on the C3 shape (BPR + item tags) d = 50 -> 64 gains 22 %, but
    // d = 20 -> 32 loses 2 % and d = 40 -> 48 8 % -- so rows of 49..63 floats always go to 64, rows of 17..47 only when BOTH
    // feature matrices are identities (the models the tile kernels serve); profiles/r06_row_align.txt.
    // LIGHTFM_AMD_ROW_ALIGN (read per session): 0 = multiples of 4 as before; 2 = the 49..63 rule only -- what
    // lightfm_amd/distributed.py asks for, because the ranks of a job exchange item rows and must agree on their stride,
    // while a rank's slice of the user feature matrix may look like an identity on one rank and not on another.  (Scoring
    // sessions keep the multiple of 4: their cost is the K extent of the matrix sweep.)
    const char *row_align_env = getenv("LIGHTFM_AMD_ROW_ALIGN");
    const int row_align = row_align_env ? atoi(row_align_env) : 1;
    if (!scoring && row_align != 0 && s->d > 16 && s->d < 64) {
        if (s->d > 48) s->d = 64;
        else if (row_align == 1 && DevCsr::is_identity(item_features) && DevCsr::is_identity(user_features)) s->d = s->d <= 32 ? 32 : 48;
    }
    s->adadelta = model->adadelta;
    s->max_sampled = model->max_sampled;
    s->lr = model->lr;
    s->rho = model->rho;
    s->eps = model->eps;
    s->scoring_only = scoring;
    // Tables that fit ONE XCD's 4 MiB L2 stay cached (small models, the parity tests); beyond that
    // they are allocated uncached (see table_alloc_flags): measured on a 1/8 row shard of the
    // ML-20M shape (23 MB of tables) 880 vs 726 M interactions/s.
    size_t table_bytes = 0;
    for (int side = 0; side < 2; ++side)
        for (int k = 0; k < 6; ++k)
            if (kind_used(s, k)) table_bytes += tab_count(s, side, k) * sizeof(float);
    // (a scoring session only reads its tables: ordinary cached memory)
    const bool big_tables = !scoring && (table_bytes > (4u << 20) || getenv("LIGHTFM_AMD_TABLE_ALLOC") != nullptr);
    for (int side = 0; side < 2 && rc == LFM_OK; ++side)
        for (int k = 0; k < 6 && rc == LFM_OK; ++k)
            s->tab[side][k].flags = (big_tables && ((table_alloc_mask() >> k) & 1)) ? table_alloc_flags() : 0;
    for (int side = 0; side < 2 && rc == LFM_OK; ++side)
        for (int k = 0; k < 6 && rc == LFM_OK; ++k)
            if (kind_used(s, k)) guard(upload_table(s, side, k, host_tab(model, side, k)));
    double sc[2] = {model->item_scale, model->user_scale}, zero[2] = {0.0, 0.0};
    if (rc == LFM_OK) guard(s->scales.upload(sc, 2));
    if (rc == LFM_OK) guard(s->reg_log.upload(zero, 2));
    std::vector<float> live0(RegScale::FLOATS, 0.0f);  // device.hpp: RegScale (line 0 = log-scales and rates, then the slots)
    s->reg_live.flags = (int)hipDeviceMallocUncached;  // read and added to by every XCD while a launch runs
    if (rc == LFM_OK) guard(s->reg_live.upload(live0.data(), live0.size()));
    if (rc == LFM_OK) guard(s->counters.alloc(13));
    if (rc == LFM_OK) guard(s->flag.alloc(1));
    if (rc == LFM_OK) guard(s->itf.upload(item_features, true, true));
    if (rc == LFM_OK) guard(s->usf.upload(user_features, true, true));
    // feature ids index the embedding tables
    if (rc == LFM_OK && !s->itf.identity)
        guard(check_id_range(s, s->itf.indices.p, s->itf.nnz, s->n_feat[0], "item_features.indices"));
    if (rc == LFM_OK && !s->usf.identity)
        guard(check_id_range(s, s->usf.indices.p, s->usf.nnz, s->n_feat[1], "user_features.indices"));
    if (rc != LFM_OK) {
        delete s;
        return rc;
    }
    *out = s;
    return LFM_OK;
}

extern "C" int lfm_session_create(lfm_session **out, int device, const lfm_model *model,
                                  const lfm_csr *item_features, const lfm_csr *user_features)
{
    return create_session(out, device, model, item_features, user_features, false);
}

extern "C" int lfm_session_create_scoring(lfm_session **out, int device, const lfm_model *model,
                                          const lfm_csr *item_features, const lfm_csr *user_features)
{
    return create_session(out, device, model, item_features, user_features, true);
}

extern "C" int lfm_selftest_adagrad_cell(int64_t n, uint32_t seed, float learning_rate, int64_t *mismatches, int64_t *fallbacks)
{
    if (n < 0 || !mismatches || !fallbacks) return fail(LFM_EINVAL, "bad self-test arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(LFM_ENODEV, "no HIP device");
    HIP_TRY(hipSetDevice(0));
    DBuf<unsigned long long> out;
    LFM_TRY(out.alloc(2));
    HIP_TRY(hipMemset(out.p, 0, 2 * sizeof(unsigned long long)));
    HIP_TRY(launch_selftest_adagrad(n, seed, learning_rate, out.p, nullptr));
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, out.p, sizeof(h), hipMemcpyDeviceToHost));
    *mismatches = (int64_t)h[0];
    *fallbacks = (int64_t)h[1];
    return LFM_OK;
}

extern "C" int lfm_selftest_ranks_bf16_band(int64_t tiles, uint32_t seed, int32_t d, int32_t spread, float *worst_fraction, int64_t *beyond)
{
    if (tiles < 0 || d < 1 || d > 128 || spread < 1 || spread > 64 || !worst_fraction || !beyond) return fail(LFM_EINVAL, "bad self-test arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(LFM_ENODEV, "no HIP device");
    HIP_TRY(hipSetDevice(0));
    DBuf<unsigned> out;
    LFM_TRY(out.alloc(2));
    HIP_TRY(hipMemset(out.p, 0, 2 * sizeof(unsigned)));
    HIP_TRY(launch_ranks_bf_band_selftest(tiles, seed, d, spread, out.p, nullptr));
    unsigned h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, out.p, sizeof(h), hipMemcpyDeviceToHost));
    memcpy(worst_fraction, &h[0], sizeof(float));
    *beyond = (int64_t)h[1];
    return LFM_OK;
}

extern "C" int lfm_session_set_features(lfm_session *s, const lfm_csr *item_features, const lfm_csr *user_features)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const lfm_csr *f[2] = {item_features, user_features};
    DevCsr *dst[2] = {&s->itf, &s->usf};
    const char *names[2] = {"item_features", "user_features"}, *idx[2] = {"item_features.indices", "user_features.indices"};
    for (int side = 0; side < 2; ++side) {
        if (!f[side]) continue;  // NULL keeps the resident matrix
        LFM_TRY(validate_csr(f[side], names[side]));
        if (f[side]->cols > s->n_feat[side]) return fail(LFM_EINVAL, "feature matrix has more columns than there are embeddings");
        dst[side]->clear();
        if (side == 0) s->hot.state = 0;  // the hot set is derived from the item feature matrix
        LFM_TRY(dst[side]->upload(f[side], true, true));
        if (!dst[side]->identity) LFM_TRY(check_id_range(s, dst[side]->indices.p, dst[side]->nnz, s->n_feat[side], idx[side]));
    }
    s->guard_sums.clear();
    return LFM_OK;
}

extern "C" int lfm_session_destroy(lfm_session *s)
{
    if (!s) return LFM_OK;
    (void)hipSetDevice(s->device);
    delete s;
    return LFM_OK;
}

// The Bloom filter over the resident positives lookup (tile kernel: in_positives pre-filter).
static int build_bloom(lfm_session *s)
{
    s->bloom_valid = false;
    if (!s->pos.indptr.p || s->pos.rows <= 0 || s->pos.nnz <= 0) return LFM_OK;
    LFM_TRY(s->bloom.alloc((size_t)Bloom::words(s->pos.nnz)));
    HIP_TRY(build_positives_bloom(s->pos.indptr.p, s->pos.indices.p, s->pos.rows, s->pos.nnz, s->bloom.p, s->stream));
    s->bloom_valid = true;
    return LFM_OK;
}

extern "C" int lfm_session_set_interactions(lfm_session *s, const lfm_csr *positives,
                                            const int32_t *user_ids, const int32_t *item_ids,
                                            const float *Y, const float *sample_weight, int64_t n)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    if (n < 0 || n > 0x7fffffffLL) return fail(LFM_EINVAL, "interaction count out of int32 range");
    if (n && !user_ids) return fail(LFM_EINVAL, "null user_ids");
    HIP_TRY(hipSetDevice(s->device));
    s->n = n;
    s->recs_valid = false;
    s->user_pair_share = -1.0;
    s->guard_sums.clear();  // new contents: nothing to compare against
    // an argument that is NULL leaves nothing of an earlier upload behind
    if (positives) {
        LFM_TRY(validate_csr(positives, "interactions"));
        LFM_TRY(s->pos.upload(positives, false, false));
    } else {
        s->pos.clear();
    }
    LFM_TRY(s->user_ids.upload(user_ids, (size_t)n));
    if (item_ids) LFM_TRY(s->item_ids.upload(item_ids, (size_t)n)); else s->item_ids.release();
    if (Y) LFM_TRY(s->Y.upload(Y, (size_t)n)); else s->Y.release();
    s->weight_aliases_Y = (Y != nullptr && sample_weight == Y);
    if (sample_weight && !s->weight_aliases_Y) LFM_TRY(s->weight.upload(sample_weight, (size_t)n));
    else s->weight.release();
    LFM_TRY(check_id_range(s, s->user_ids.p, n, s->usf.rows, "user_ids (rows of user_features)"));
    LFM_TRY(check_id_range(s, s->item_ids.p, s->item_ids.p ? n : 0, s->itf.rows, "item_ids (rows of item_features)"));
    if (positives) {
        if (positives->rows < 0 || positives->cols > s->itf.rows)
            return fail(LFM_EINVAL, "interactions matrix has more columns than item_features has rows");
        LFM_TRY(check_id_range(s, s->pos.indices.p, s->pos.nnz, std::max<int64_t>(positives->cols, 1), "interactions.indices"));
    }
    LFM_TRY(build_bloom(s));
    return LFM_OK;
}

extern "C" int lfm_session_upload_shuffle(lfm_session *s, int32_t slot, const int32_t *shuffle, int64_t n)
{
    if (!s || slot < 0 || slot > 4096) return fail(LFM_EINVAL, "bad shuffle slot");
    if (n != s->n) return fail(LFM_EINVAL, "shuffle length differs from the interaction count");
    HIP_TRY(hipSetDevice(s->device));
    while ((int)s->shuffles.size() <= slot) s->shuffles.push_back(new DBuf<int32_t>());
    return s->shuffles[slot]->upload(shuffle, (size_t)n);
}

// Keyed pseudo-random permutation of [0, n): a 6-round Feistel network over 2^(2h) >= n values
// with cycle walking (re-encrypt until the value falls below n).  Every index is computed
// independently, so the shuffle of an epoch is one streaming kernel instead of numpy's
// sequential Fisher-Yates (0.3-0.6 s per 18 M entries on the host) plus an upload.
__host__ __device__ inline uint32_t feistel_round(uint32_t x, uint32_t key)
{
    x ^= key;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

__host__ __device__ inline uint32_t feistel_permute(uint32_t i, uint32_t n, int half_bits, uint32_t k0,
                                                    uint32_t k1)
{
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t v = i;
    do {
        uint32_t l = v >> half_bits, r = v & mask;
        for (uint32_t round = 0; round < 6; ++round) {
            uint32_t t = l ^ (feistel_round(r, (round & 1 ? k1 : k0) + round * 0x9E3779B9u) & mask);
            l = r;
            r = t;
        }
        v = (l << half_bits) | r;
    } while (v >= n);
    return v;
}

__global__ void device_shuffle_kernel(int32_t *out, int64_t n, int half_bits, uint32_t k0, uint32_t k1)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) out[j] = (int32_t)feistel_permute((uint32_t)j, (uint32_t)n, half_bits, k0, k1);
}

static int feistel_half_bits(int64_t n)
{
    int h = 1;
    while ((1ll << (2 * h)) < n) ++h;
    return h;
}

extern "C" int lfm_session_device_shuffle(lfm_session *s, int32_t slot, uint32_t key0, uint32_t key1)
{
    if (!s || slot < 0 || slot > 4096) return fail(LFM_EINVAL, "bad shuffle slot");
    if (s->n >= (1ll << 31)) return fail(LFM_EINVAL, "interaction count out of int32 range");
    HIP_TRY(hipSetDevice(s->device));
    while ((int)s->shuffles.size() <= slot) s->shuffles.push_back(new DBuf<int32_t>());
    LFM_TRY(s->shuffles[slot]->alloc((size_t)s->n));
    if (s->n == 0) return LFM_OK;
    int grid = (int)std::min<int64_t>(4096, (s->n + 255) / 256);
    device_shuffle_kernel<<<grid, 256, 0, s->stream>>>(s->shuffles[slot]->p, s->n, feistel_half_bits(s->n), key0, key1);
    HIP_TRY(hipGetLastError());
    return LFM_OK;
}

// The same permutation written on the session's auxiliary stream: call it for the slot of epoch e + 1 BEFORE running epoch e
// (another slot); lfm_session_epoch on the slot waits for it.  The Feistel rounds are arithmetic, the epoch kernels wait
// for memory: the 0.4 ms the shuffle of 20 M positions takes alone at the head of an epoch disappear under the epoch before.
extern "C" int lfm_session_device_shuffle_ahead(lfm_session *s, int32_t slot, uint32_t key0, uint32_t key1)
{
    if (!s || slot < 0 || slot > 4096) return fail(LFM_EINVAL, "bad shuffle slot");
    if (s->n >= (1ll << 31)) return fail(LFM_EINVAL, "interaction count out of int32 range");
    HIP_TRY(hipSetDevice(s->device));
    while ((int)s->shuffles.size() <= slot) s->shuffles.push_back(new DBuf<int32_t>());
    while ((int)s->shuffle_ready.size() <= slot) s->shuffle_ready.push_back(nullptr);
    if (!s->shuffles[slot]->p || s->shuffles[slot]->n != (size_t)s->n) {
        HIP_TRY(hipStreamSynchronize(s->stream));  // (an allocation that changes size waits for everything: first use of the slot)
        LFM_TRY(s->shuffles[slot]->alloc((size_t)s->n));
    }
    if (s->n == 0) return LFM_OK;
    if (!s->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&s->aux_stream, hipStreamNonBlocking));
    if (!s->shuffle_ready[slot]) HIP_TRY(hipEventCreateWithFlags(&s->shuffle_ready[slot], hipEventDisableTiming));
    int grid = (int)std::min<int64_t>(4096, (s->n + 255) / 256);
    device_shuffle_kernel<<<grid, 256, 0, s->aux_stream>>>(s->shuffles[slot]->p, s->n, feistel_half_bits(s->n), key0, key1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->shuffle_ready[slot], s->aux_stream));
    return LFM_OK;
}

// Host restatement of the same permutation (tests).
extern "C" int lfm_shuffle_permutation(int32_t *out, int64_t n, uint32_t key0, uint32_t key1)
{
    if (n < 0 || n >= (1ll << 31) || (n && !out)) return fail(LFM_EINVAL, "bad permutation request");
    const int h = feistel_half_bits(n);
    for (int64_t j = 0; j < n; ++j) out[j] = (int32_t)feistel_permute((uint32_t)j, (uint32_t)n, h, key0, key1);
    return LFM_OK;
}

// Downloads a shuffle slot (tests / debugging).
extern "C" int lfm_session_download_shuffle(lfm_session *s, int32_t slot, int32_t *out, int64_t n)
{
    if (!s || slot < 0 || slot >= (int)s->shuffles.size() || !out) return fail(LFM_EINVAL, "bad shuffle slot");
    if ((size_t)n != s->shuffles[slot]->n) return fail(LFM_EINVAL, "shuffle length differs");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->aux_stream) HIP_TRY(hipStreamSynchronize(s->aux_stream));
    return s->shuffles[slot]->download(out);
}

static void static_chunk(int64_t n, int32_t T, int32_t t, int64_t *lo, int64_t *hi)
{
    int64_t q = n / T, r = n % T;  // libgomp static schedule, no chunk clause (C_OMP:7224)
    if (t < r) { *lo = (q + 1) * t; *hi = *lo + q + 1; }
    else { *lo = q * t + r; *hi = *lo + q; }
}

// ------------------------------------------------------------- multi-GPU ---

__global__ void copy_kernel(float *dst, const float *src, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) dst[j] = src[j];
}
__global__ void sub_inplace_kernel(float *x, const float *y, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) x[j] -= y[j];
}
// x := (x * scale) + y ; snap := x   (end of a merge: deltas back to values, new interval start)
__global__ void finish_merge_kernel(float *x, float *snap, float scale, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) {
        const float v = x[j] * scale + snap[j];
        x[j] = v;
        snap[j] = v;
    }
}
// LFM_MERGE_ADAGRAD: dW (this rank's embedding delta, [rows, cols]) *= sqrt((G0 + dG_r/2) / (G0 + dG_all/2)),
// element by element: the step this rank would have taken had its accumulators also seen the other
// ranks' squared gradients of the interval (both at the interval's midpoint).
__global__ void adagrad_rescale_kernel(float *dW, const float *g0, const float *dg_rank, const float *dg_all,
                                       int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) {
        const float num = g0[j] + 0.5f * dg_rank[j], den = g0[j] + 0.5f * dg_all[j];
        if (den > 0.0f && num >= 0.0f && den > num) dW[j] *= sqrtf(num / den);
    }
}
struct PtrPack { float *p[16]; };
// every x_k := sum_k x_k  (the all-reduce of K sessions living on one device)
__global__ void local_allreduce_kernel(PtrPack xs, int k, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) {
        float acc = 0.0f;
        for (int r = 0; r < k; ++r) acc += xs.p[r][j];
        for (int r = 0; r < k; ++r) xs.p[r][j] = acc;
    }
}

static int grid_for(int64_t cnt) { return (int)std::max<int64_t>(1, std::min<int64_t>(4096, (cnt + 255) / 256)); }

static int snapshot_side(lfm_session *s, int side)
{
    for (int k = 0; k < 6; ++k) {
        if (!kind_used(s, k)) continue;
        size_t cnt = tab_count(s, side, k);
        LFM_TRY(s->snap[side][k].alloc(cnt));
        if (cnt)
            HIP_TRY(hipMemcpyAsync(s->snap[side][k].p, s->tab[side][k].p, cnt * sizeof(float),
                                   hipMemcpyDeviceToDevice, s->stream));
    }
    s->snap_sides |= 1 << side;
    return LFM_OK;
}

static int complete_pending(lfm_session *s, bool exact = false);

extern "C" int lfm_session_merge_begin(lfm_session *s, int32_t sides)
{
    if (!s || sides < 0 || sides > 3) return fail(LFM_EINVAL, "bad merge_begin arguments");
    if (s->scoring_only) return fail(LFM_EINVAL, "a scoring session has no merge intervals");
    HIP_TRY(hipSetDevice(s->device));
    for (int side = 0; side < 2; ++side)
        if ((sides >> side) & 1) {
            LFM_TRY(snapshot_side(s, side));
            LFM_TRY(s->dirty[side].alloc((size_t)s->n_feat[side]));
        }
    if (!s->comm_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_pack, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_comm, hipEventDisableTiming));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    return LFM_OK;
}

// The merge of SURVEY 8(e) for a group of sessions: ONE session whose peers are other processes
// (reduction = RCCL all-reduce over xGMI), or K sessions of this process on one device
// (reduction = local_allreduce_kernel; same arithmetic, used to measure N-GPU semantics on one GPU).
//   X := X0 + op_r (X_r - X0),  X0 = the snapshot taken at the start of the interval.
static int merge_group(lfm_session **ss, int k, int nranks_total, int sides, int mode)
{
    lfm_session *s0 = ss[0];
    const bool use_rccl = (k == 1 && s0->comm != nullptr);
    Rccl *r = use_rccl ? rccl() : nullptr;
    if (use_rccl && !r) return fail(LFM_ECOMM, "librccl.so not available");
    if (k > 16) return fail(LFM_EINVAL, "at most 16 local sessions per merge group");
    if (mode < 0 || mode > 2) return fail(LFM_EINVAL, "unknown merge mode");
    if (s0->adadelta && mode == LFM_MERGE_ADAGRAD) mode = LFM_MERGE_MEAN;  // moving averages: no squared-gradient sums
    // local groups run on sessions[0]'s stream; the others' streams are drained first
    hipStream_t st = s0->stream;
    for (int i = 1; i < k; ++i) HIP_TRY(hipStreamSynchronize(ss[i]->stream));
    for (int i = 0; i < k; ++i) {  // a sparse exchange still in flight lands first; the dense merge covers every row
        LFM_TRY(complete_pending(ss[i]));
        HIP_TRY(hipStreamSynchronize(ss[i]->stream));
    }
    auto reduce_kinds = [&](int side, std::initializer_list<int> kinds) -> int {
        if (use_rccl) {
            if (r->GroupStart) NCCL_TRY(r->GroupStart());
            for (int kind : kinds) {
                if (!kind_used(s0, kind)) continue;
                size_t cnt = tab_count(s0, side, kind);
                if (!cnt) continue;
                NCCL_TRY(r->AllReduce(s0->tab[side][kind].p, s0->tab[side][kind].p, cnt, ncclFloat, ncclSum, s0->comm, st));
            }
            if (r->GroupEnd) NCCL_TRY(r->GroupEnd());
        } else if (k > 1) {
            for (int kind : kinds) {
                if (!kind_used(s0, kind)) continue;
                int64_t cnt = (int64_t)tab_count(s0, side, kind);
                if (!cnt) continue;
                PtrPack pk;
                for (int i = 0; i < k; ++i) pk.p[i] = ss[i]->tab[side][kind].p;
                local_allreduce_kernel<<<grid_for(cnt), 256, 0, st>>>(pk, k, cnt);
            }
        }
        return LFM_OK;
    };
    for (int side = 0; side < 2; ++side) {
        if (!((sides >> side) & 1)) continue;
        for (int i = 0; i < k; ++i) {
            if (!((ss[i]->snap_sides >> side) & 1)) return fail(LFM_EINVAL, "merge without lfm_session_merge_begin");
            if (ss[i]->n_feat[side] != s0->n_feat[side] || ss[i]->d != s0->d) return fail(LFM_EINVAL, "sessions differ in shape");
        }
        // 1. tables -> deltas
        for (int i = 0; i < k; ++i)
            for (int kind = 0; kind < 6; ++kind) {
                if (!kind_used(s0, kind)) continue;
                int64_t cnt = (int64_t)tab_count(s0, side, kind);
                if (cnt) sub_inplace_kernel<<<grid_for(cnt), 256, 0, st>>>(ss[i]->tab[side][kind].p, ss[i]->snap[side][kind].p, cnt);
            }
        float wscale = 1.0f;
        if (mode == LFM_MERGE_ADAGRAD) {
            // accumulators first (keeping this rank's own dG), then the rescaled embedding deltas
            for (int i = 0; i < k; ++i) {
                lfm_session *s = ss[i];
                const size_t cw = tab_count(s, side, 1), cb = tab_count(s, side, 4);
                LFM_TRY(s->scratch[0].alloc(cw));
                LFM_TRY(s->scratch[1].alloc(cb));
                if (cw) HIP_TRY(hipMemcpyAsync(s->scratch[0].p, s->tab[side][1].p, cw * sizeof(float), hipMemcpyDeviceToDevice, st));
                if (cb) HIP_TRY(hipMemcpyAsync(s->scratch[1].p, s->tab[side][4].p, cb * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
            LFM_TRY(reduce_kinds(side, {1, 4}));
            for (int i = 0; i < k; ++i) {
                lfm_session *s = ss[i];
                const int64_t cw = (int64_t)tab_count(s, side, 1), cb = (int64_t)tab_count(s, side, 4);
                if (cw) adagrad_rescale_kernel<<<grid_for(cw), 256, 0, st>>>(s->tab[side][0].p, s->snap[side][1].p, s->scratch[0].p, s->tab[side][1].p, cw);
                if (cb) adagrad_rescale_kernel<<<grid_for(cb), 256, 0, st>>>(s->tab[side][3].p, s->snap[side][4].p, s->scratch[1].p, s->tab[side][4].p, cb);
            }
            LFM_TRY(reduce_kinds(side, {0, 3}));
        } else {
            LFM_TRY(reduce_kinds(side, {0, 1, 2, 3, 4, 5}));
            if (mode == LFM_MERGE_MEAN) wscale = 1.0f / (float)nranks_total;
        }
        // 2. deltas -> values; the merged state starts the next interval
        for (int i = 0; i < k; ++i)
            for (int kind = 0; kind < 6; ++kind) {
                if (!kind_used(s0, kind)) continue;
                int64_t cnt = (int64_t)tab_count(s0, side, kind);
                if (!cnt) continue;
                const bool is_weight = kind == 0 || kind == 3;
                const float sc = (is_weight || s0->adadelta) ? wscale : 1.0f;
                finish_merge_kernel<<<grid_for(cnt), 256, 0, st>>>(ss[i]->tab[side][kind].p, ss[i]->snap[side][kind].p, sc, cnt);
            }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return LFM_OK;
}

// ------------------------------------------------- sparse (dirty-row) merge ---
// The dense merge above moves every replicated table through the fabric six times and through xGMI once per
// merge, whatever the interval touched: 2.6 GB for the 5 M-row item tables of BASELINE config C4, against
// ~1 ms of training per 2^20 interactions.  This one is proportional to what changed:
//   1. the rows whose W, G, b or bG differ from the interval's snapshot are flagged in a byte map (one streaming
//      read of the tables at merge time.  Marking the rows from inside the epoch kernels was measured first: one
//      predicated byte store per updated row cost the C2 tile kernel 6 %, 0.99 against 1.05 G interactions/s in
//      an A/B of two builds on one box -- on single-GPU runs that never merge);
//   2. the ranks' maps are OR-ed (all-reduce MAX over bytes: n_feat bytes) -- every rank then holds the same
//      union U of touched rows -- and compacted to an ascending id list (csr_build.hip: compact_flagged_rows);
//   3. each rank packs its deltas (table - snapshot) of the rows of U into dense [|U|, d] buffers;
//   4. the packed buffers are all-reduced (RCCL over xGMI; LFM_MERGE_ADAGRAD: accumulators first, then the
//      embedding deltas rescaled exactly as in the dense merge) on the session's COMMUNICATION stream;
//   5. apply, rows of U only: table += sum - local delta, snapshot += sum.
// With `overlap` steps 4-5 are asynchronous: the call returns after enqueueing the exchange, the next segment
// trains on tables that hold only the rank's own updates, and step 5 runs at the start of the NEXT merge call
// (or lfm_session_comm_merge_flush): the exchange of segment j overlaps the kernels of segment j + 1, at the
// price of the other ranks' updates arriving one segment late.  The arithmetic does not depend on it: whatever
// the local table gained between pack and apply stays in (table - snapshot) for the next merge.
// flags[r] = 1 if row r of (W, G) or its (b, bG) cell differs from the snapshot; one wavefront per row
__global__ void detect_dirty_kernel(const float *W, const float *sW, const float *G, const float *sG, const float *b, const float *sb,
                                    const float *bG, const float *sbG, const float *M, const float *sM, const float *bM,
                                    const float *sbM, int64_t rows, int d, unsigned char *flags)
{
    const int lane = threadIdx.x & 63;
    const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = w0; r < rows; r += nw) {
        bool diff = false;
        const size_t base = (size_t)r * d;
        for (int c = lane; c < d; c += 64) {
            diff |= (W[base + c] != sW[base + c]) | (G[base + c] != sG[base + c]);
            if (M) diff |= M[base + c] != sM[base + c];  // adadelta: the momentum tables travel too
        }
        if (lane == 0) {
            diff |= (b[r] != sb[r]) | (bG[r] != sbG[r]);
            if (bM) diff |= bM[r] != sbM[r];
        }
        const unsigned long long any = __ballot(diff);
        if (lane == 0) flags[r] = any ? 1 : 0;
    }
}
// ---- steps 3-5 on ONE buffer per side and ONE all-reduce per merge.  A side's packed deltas live in one allocation of
// K planes of P = n_u (d + 1) floats, a plane = the cells of one table kind over the union's rows: first the n_u x d
// embedding cells (row-major over the union), then the n_u bias cells.  Planes: adagrad [accumulators | weights],
// adadelta [accumulators | momenta | weights].  One thread owns cell r of EVERY plane (the accumulator and the weight of
// a cell are merged together), so a merge is: one pack launch, one all-reduce over K P floats, one apply launch --
// the merge of a small table (ML-20M's 26 744 item rows: 14 MB) is bound by its launches, not its bytes.
// LFM_MERGE_ADAGRAD in one exchange: the merged step sum_r dW_r sqrt((G0 + dG_r / 2) / (G0 + sum dG / 2)) has a
// rank-independent denominator, so a rank sends dW_r sqrt(G0 + dG_r / 2) next to dG_r and divides the summed
// numerators by sqrt(G0 + sum dG / 2) when it applies them (round 4 reduced the accumulators first, rescaled, and
// reduced the weights: two exchanges; the same quantity up to the rounding of sqrt(a) / sqrt(b) against sqrt(a / b)).
struct FusedPlan {
    float *tab[3][2], *snap[3][2];  // [plane: 0 accumulators, 1 weights, 2 momenta (adadelta)][0 embedding table, 1 bias table]
    int planes;                     // 2 (adagrad) or 3 (adadelta)
    int d, adagrad_mode;
    float wscale, ascale;           // what the summed deltas of the weights / accumulators (and momenta) are scaled by
    int64_t n_u, P;                 // rows of the union, cells per plane = n_u (d + 1)
    const int32_t *ids;             // nullptr: every row (u is the row)
    // buffer order: accumulators, [momenta], weights
    __device__ __forceinline__ int64_t off_w() const { return (int64_t)(planes - 1) * P; }
    __device__ __forceinline__ void locate(int64_t r, int &t, size_t &at) const
    {
        const int64_t ne = n_u * d;
        if (r < ne) {
            const int64_t u = r / d;
            t = 0;
            at = (size_t)(ids ? ids[u] : (int32_t)u) * d + (size_t)(r - u * d);
        } else {
            const int64_t u = r - ne;
            t = 1;
            at = (size_t)(ids ? ids[u] : (int32_t)u);
        }
    }
};
__global__ void pack_fused_kernel(FusedPlan p, float *loc, float *sum)
{
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    const int64_t ow = p.off_w();
    for (int64_t r = t0; r < p.P; r += st) {
        int t;
        size_t at;
        p.locate(r, t, at);
        const float g0 = p.snap[0][t][at];
        const float dg = p.tab[0][t][at] - g0, dw = p.tab[1][t][at] - p.snap[1][t][at];
        loc[r] = sum[r] = dg;
        loc[ow + r] = dw;
        sum[ow + r] = p.adagrad_mode ? dw * sqrtf(fmaxf(g0 + 0.5f * dg, 0.0f)) : dw;
        if (p.planes == 3) loc[p.P + r] = sum[p.P + r] = p.tab[2][t][at] - p.snap[2][t][at];
    }
}
// table += scale * sum - local ; snapshot += scale * sum   (rows of U).  exact: nothing trained since the pack
// (synchronous merge): table := snapshot := snapshot + scale * sum, bit-identical on every rank.
__global__ void apply_fused_kernel(FusedPlan p, const float *sum, const float *loc, int exact)
{
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    const int64_t ow = p.off_w();
    for (int64_t r = t0; r < p.P; r += st) {
        int t;
        size_t at;
        p.locate(r, t, at);
        const float g0 = p.snap[0][t][at], sg_raw = sum[r];
        float sw = sum[ow + r];
        if (p.adagrad_mode) {
            const float den = g0 + 0.5f * sg_raw;  // (adagrad accumulators start at 1 and only grow)
            sw = den > 0.0f ? sw / sqrtf(den) : 0.0f;
        } else {
            sw *= p.wscale;
        }
        const float sg = sg_raw * p.ascale;
        const float ng = g0 + sg, nw = p.snap[1][t][at] + sw;
        p.tab[0][t][at] = exact ? ng : p.tab[0][t][at] + (sg - loc[r]);
        p.snap[0][t][at] = ng;
        p.tab[1][t][at] = exact ? nw : p.tab[1][t][at] + (sw - loc[ow + r]);
        p.snap[1][t][at] = nw;
        if (p.planes == 3) {
            const float sm = sum[p.P + r] * p.ascale, nm = p.snap[2][t][at] + sm;
            p.tab[2][t][at] = exact ? nm : p.tab[2][t][at] + (sm - loc[p.P + r]);
            p.snap[2][t][at] = nm;
        }
    }
}
struct BytePack { unsigned char *p[16]; };
__global__ void local_or_kernel(BytePack xs, int k, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) {
        unsigned char v = 0;
        for (int r = 0; r < k; ++r) v |= xs.p[r][j];
        for (int r = 0; r < k; ++r) xs.p[r][j] = v;
    }
}

// the plan of a side's merge (see FusedPlan)
static FusedPlan fused_plan(lfm_session *s, int side, int64_t n_u, const int32_t *ids, float wscale, float ascale, int adagrad_mode)
{
    FusedPlan p;
    memset(&p, 0, sizeof(p));
    p.planes = s->adadelta ? 3 : 2;
    static const int KIND[3][2] = {{1, 4}, {0, 3}, {2, 5}};  // accumulators (G, bG), weights (W, b), momenta (M, bM)
    for (int pl = 0; pl < p.planes; ++pl)
        for (int t = 0; t < 2; ++t) {
            p.tab[pl][t] = s->tab[side][KIND[pl][t]].p;
            p.snap[pl][t] = s->snap[side][KIND[pl][t]].p;
        }
    p.d = s->d;
    p.adagrad_mode = adagrad_mode;
    p.wscale = wscale;
    p.ascale = ascale;
    p.n_u = n_u;
    p.P = n_u * ((int64_t)s->d + 1);
    p.ids = ids;
    return p;
}
static int64_t fused_row_floats(const lfm_session *s) { return (s->adadelta ? 3 : 2) * ((int64_t)s->d + 1); }

// step 5 of a merge whose exchange may still be in flight
static int complete_pending(lfm_session *s, bool exact)
{
    if (!s->pend.active) return LFM_OK;
    if (s->pend.on_comm_stream) HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_comm, 0));
    for (int side = 0; side < 2; ++side) {
        if (!((s->pend.sides >> side) & 1)) continue;
        const int64_t n_u = s->pend.n_u[side];
        if (n_u == 0) continue;
        const FusedPlan p = fused_plan(s, side, n_u, s->pend.all_rows[side] ? nullptr : s->pend.ids[side].p, s->pend.wscale, s->pend.ascale,
                                       s->pend.adagrad_mode);
        apply_fused_kernel<<<grid_for(p.P), 256, 0, s->stream>>>(p, s->pend.sum[side].p, s->pend.loc[side].p, exact ? 1 : 0);
    }
    HIP_TRY(hipGetLastError());
    s->pend.active = false;
    return LFM_OK;
}

static int merge_group_sparse(lfm_session **ss, int k, int nranks_total, int sides, int mode, bool overlap, int64_t *bytes_out,
                              bool hot_only = false)
{
    lfm_session *s0 = ss[0];
    const bool use_rccl = (k == 1 && s0->comm != nullptr);
    Rccl *r = use_rccl ? rccl() : nullptr;
    if (use_rccl && !r) return fail(LFM_ECOMM, "librccl.so not available");
    if (k > 16) return fail(LFM_EINVAL, "at most 16 local sessions per merge group");
    if (mode < 0 || mode > 2) return fail(LFM_EINVAL, "unknown merge mode");
    if (s0->adadelta && mode == LFM_MERGE_ADAGRAD) mode = LFM_MERGE_MEAN;  // moving averages: no squared-gradient sums (as merge_group)
    int64_t bytes = 0;
    // local groups run on sessions[0]'s stream; the others' streams are drained first
    hipStream_t st = s0->stream;
    for (int i = 1; i < k; ++i) HIP_TRY(hipStreamSynchronize(ss[i]->stream));
    // the previous merge's exchange has to land before this one's deltas are formed
    for (int i = 0; i < k; ++i) {
        if (!ss[i]->pend.active) continue;
        if (i > 0 && ss[i]->pend.on_comm_stream) return fail(LFM_EINVAL, "local merge group with a communicator");
        hipStream_t own = ss[i]->stream;
        ss[i]->stream = st;  // (local groups: everything on sessions[0]'s stream)
        const int rc = complete_pending(ss[i]);
        ss[i]->stream = own;
        LFM_TRY(rc);
    }
    const float wscale = mode == LFM_MERGE_MEAN ? 1.0f / (float)nranks_total : 1.0f;
    const float ascale = s0->adadelta ? wscale : 1.0f;  // adadelta's accumulators are moving averages: averaged like the weights
    const int64_t row_floats = fused_row_floats(s0);
    for (int side = 0; side < 2; ++side) {
        if (!((sides >> side) & 1)) continue;
        const int64_t nf = s0->n_feat[side];
        for (int i = 0; i < k; ++i) {
            if (!((ss[i]->snap_sides >> side) & 1) || !ss[i]->dirty[side].p)
                return fail(LFM_EINVAL, "sparse merge without lfm_session_merge_begin");
            if (ss[i]->n_feat[side] != nf || ss[i]->d != s0->d || ss[i]->adadelta != s0->adadelta)
                return fail(LFM_EINVAL, "sessions differ in shape");
        }
        if (nf == 0) continue;
        // ... re-examined every 16 merges (a long early interval of a large table must not make every later, shorter
        // interval ship the whole table: ADVICE r5), and never taken for a side whose fused buffer exceeds 64 MB -- the
        // latch pays where detection, OR and compaction cost more than the zeros they would have saved (ML-20M's item side:
        // 13.9 MB).  Both conditions depend on counts every rank shares, so all ranks take the same path.
        bool all_rows = !hot_only && s0->merge_all_rows[side] && s0->merge_dense_frac > 0.0f;
        if (all_rows && (s0->merge_all_uses[side] >= 16 || (int64_t)nf * row_floats * (int64_t)sizeof(float) > ((int64_t)64 << 20))) {
            all_rows = false;
            for (int i = 0; i < k; ++i) {
                ss[i]->merge_all_rows[side] = false;
                ss[i]->merge_all_uses[side] = 0;
            }
        }
        if (!hot_only && s0->merge_all_rows[side] && s0->merge_dense_frac <= 0.0f) all_rows = true;  // "always": the caller's choice
        if (all_rows)
            for (int i = 0; i < k; ++i) ++ss[i]->merge_all_uses[side];
        if (hot_only) {
            // the given rows (identical on every rank by contract): no detection, no union, no compaction
            for (int i = 0; i < k; ++i) {
                lfm_session *s = ss[i];
                const size_t nh = s->hot_ids[side].n;
                if (nh != s0->hot_ids[side].n) return fail(LFM_EINVAL, "sessions differ in their hot rows");
                s->pend.n_u[side] = (int64_t)nh;
                s->pend.all_rows[side] = false;
                if (nh) {
                    LFM_TRY(s->pend.ids[side].reserve(nh));
                    HIP_TRY(hipMemcpyAsync(s->pend.ids[side].p, s->hot_ids[side].p, nh * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
                }
            }
        } else if (all_rows) {
            // an earlier union covered (nearly) the whole table: every row travels, nothing is detected, OR-ed or compacted
            for (int i = 0; i < k; ++i) {
                ss[i]->pend.n_u[side] = nf;
                ss[i]->pend.all_rows[side] = true;
            }
        } else {
            // 1. rows that differ from the snapshot
            for (int i = 0; i < k; ++i) {
                lfm_session *s = ss[i];
                const int dgrid = (int)std::max<int64_t>(1, std::min<int64_t>(8192, (nf + 3) / 4));
                detect_dirty_kernel<<<dgrid, 256, 0, st>>>(s->tab[side][0].p, s->snap[side][0].p, s->tab[side][1].p, s->snap[side][1].p,
                                                           s->tab[side][3].p, s->snap[side][3].p, s->tab[side][4].p, s->snap[side][4].p,
                                                           s->adadelta ? s->tab[side][2].p : nullptr, s->adadelta ? s->snap[side][2].p : nullptr,
                                                           s->adadelta ? s->tab[side][5].p : nullptr, s->adadelta ? s->snap[side][5].p : nullptr,
                                                           nf, s->d, s->dirty[side].p);
            }
            // 2. union of the maps
            if (use_rccl) {
                NCCL_TRY(r->AllReduce(s0->dirty[side].p, s0->dirty[side].p, (size_t)nf, ncclUint8, ncclMax, s0->comm, st));
                bytes += nf;
            } else if (k > 1) {
                BytePack pk;
                for (int i = 0; i < k; ++i) pk.p[i] = ss[i]->dirty[side].p;
                local_or_kernel<<<grid_for(nf), 256, 0, st>>>(pk, k, nf);
            }
            // ... compacted (identical on every rank: same map, ascending order)
            for (int i = 0; i < k; ++i) {
                lfm_session *s = ss[i];
                LFM_TRY(s->pend.ids[side].reserve((size_t)nf));
                int64_t n_u = 0;
                hipError_t e = compact_flagged_rows(s->dirty[side].p, nf, s->pend.ids[side].p, &n_u, st);
                if (e != hipSuccess) return fail(LFM_ENODEV, std::string("compact_flagged_rows: ") + hipGetErrorString(e));
                s->pend.n_u[side] = n_u;
                s->pend.all_rows[side] = false;
                // (the union is the same on every rank: all take the same path from the next merge on)
                if ((double)n_u >= (double)s->merge_dense_frac * (double)nf && (int64_t)nf * row_floats * (int64_t)sizeof(float) <= ((int64_t)64 << 20)) {
                    s->merge_all_rows[side] = true;
                    s->merge_all_uses[side] = 0;
                }
            }
        }
        const int64_t n_u = s0->pend.n_u[side];
        for (int i = 1; i < k; ++i)
            if (ss[i]->pend.n_u[side] != n_u) return fail(LFM_ECORRUPT, "local merge group: the sessions' unions differ");
        // 3. pack: one launch for all kinds
        for (int i = 0; i < k && n_u; ++i) {
            lfm_session *s = ss[i];
            LFM_TRY(s->pend.sum[side].reserve((size_t)(n_u * row_floats)));
            LFM_TRY(s->pend.loc[side].reserve((size_t)(n_u * row_floats)));
            const FusedPlan plan = fused_plan(s, side, n_u, s->pend.all_rows[side] ? nullptr : s->pend.ids[side].p, wscale, ascale,
                                              mode == LFM_MERGE_ADAGRAD ? 1 : 0);
            pack_fused_kernel<<<grid_for(plan.P), 256, 0, st>>>(plan, s->pend.loc[side].p, s->pend.sum[side].p);
        }
        HIP_TRY(hipGetLastError());
        if (use_rccl) bytes += n_u * row_floats * (int64_t)sizeof(float);
    }
    // 4. the exchange: RCCL on the communication stream (overlaps the next segment), local groups in place
    hipStream_t cs = st;
    if (use_rccl) {
        cs = s0->comm_stream;
        HIP_TRY(hipEventRecord(s0->ev_pack, st));
        HIP_TRY(hipStreamWaitEvent(cs, s0->ev_pack, 0));
    }
    auto reduce = [&](int side, int64_t off, int64_t cnt) -> int {  // floats [off, off + cnt) of the side's fused buffer
        if (cnt <= 0) return LFM_OK;
        if (use_rccl) {
            NCCL_TRY(r->AllReduce(s0->pend.sum[side].p + off, s0->pend.sum[side].p + off, (size_t)cnt, ncclFloat, ncclSum, s0->comm, cs));
        } else if (k > 1) {
            PtrPack pk;
            for (int i = 0; i < k; ++i) pk.p[i] = ss[i]->pend.sum[side].p + off;
            local_allreduce_kernel<<<grid_for(cnt), 256, 0, cs>>>(pk, k, cnt);
        }
        return LFM_OK;
    };
    for (int side = 0; side < 2; ++side) {
        if (!((sides >> side) & 1) || s0->n_feat[side] == 0) continue;
        const int64_t n_u = s0->pend.n_u[side];
        if (n_u == 0) continue;
        LFM_TRY(reduce(side, 0, n_u * row_floats));  // ONE exchange in every mode (FusedPlan)
    }
    HIP_TRY(hipGetLastError());
    if (use_rccl) HIP_TRY(hipEventRecord(s0->ev_comm, cs));
    for (int i = 0; i < k; ++i) {
        ss[i]->pend.active = true;
        ss[i]->pend.sides = sides;
        ss[i]->pend.wscale = wscale;
        ss[i]->pend.ascale = ascale;
        ss[i]->pend.adagrad_mode = mode == LFM_MERGE_ADAGRAD ? 1 : 0;
        ss[i]->pend.on_comm_stream = use_rccl;
    }
    if (!overlap) {
        for (int i = 0; i < k; ++i) {
            hipStream_t own = ss[i]->stream;
            ss[i]->stream = st;
            const int rc = complete_pending(ss[i], true);
            ss[i]->stream = own;
            LFM_TRY(rc);
        }
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (bytes_out) *bytes_out = bytes;
    return LFM_OK;
}

// From which share of a side's rows in a merge's union the following merges of that side carry every row without
// detecting (default 0.9; > 1: never, <= 0: from the first merge on).  The same value on every rank.
extern "C" int lfm_session_set_merge_dense_fraction(lfm_session *s, float fraction)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    s->merge_dense_frac = fraction;
    for (int side = 0; side < 2; ++side) s->merge_all_rows[side] = fraction <= 0.0f;
    return LFM_OK;
}

extern "C" int lfm_session_comm_merge_sparse(lfm_session *s, int32_t sides, int32_t mode, int32_t overlap, int64_t *bytes)
{
    if (bytes) *bytes = 0;
    if (!s || sides < 0 || sides > 3) return fail(LFM_EINVAL, "bad merge arguments");
    if (!s->comm || sides == 0) return LFM_OK;
    HIP_TRY(hipSetDevice(s->device));
    lfm_session *g[1] = {s};
    return merge_group_sparse(g, 1, s->nranks, sides, mode, overlap != 0, bytes);
}

extern "C" int lfm_session_set_hot_rows(lfm_session *s, int32_t side, const int32_t *rows, int64_t n_rows)
{
    if (!s || (side != 0 && side != 1) || n_rows < 0 || (n_rows && !rows)) return fail(LFM_EINVAL, "bad hot-row arguments");
    for (int64_t j = 0; j < n_rows; ++j)
        if (rows[j] < 0 || rows[j] >= s->n_feat[side] || (j && rows[j] <= rows[j - 1]))
            return fail(LFM_EINVAL, "hot rows must be ascending feature rows of the side");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (n_rows == 0) {
        s->hot_ids[side].release();
        return LFM_OK;
    }
    return s->hot_ids[side].upload(rows, (size_t)n_rows);
}

extern "C" int lfm_session_comm_merge_hot(lfm_session *s, int32_t sides, int32_t mode, int32_t overlap, int64_t *bytes)
{
    if (bytes) *bytes = 0;
    if (!s || sides < 0 || sides > 3) return fail(LFM_EINVAL, "bad merge arguments");
    if (!s->comm || sides == 0) return LFM_OK;
    HIP_TRY(hipSetDevice(s->device));
    lfm_session *g[1] = {s};
    return merge_group_sparse(g, 1, s->nranks, sides, mode, overlap != 0, bytes, true);
}

extern "C" int lfm_sessions_merge_local_hot(lfm_session **sessions, int32_t k, int32_t sides, int32_t mode, int32_t overlap)
{
    if (!sessions || k < 1 || sides < 0 || sides > 3) return fail(LFM_EINVAL, "bad merge arguments");
    for (int i = 0; i < k; ++i)
        if (!sessions[i] || sessions[i]->device != sessions[0]->device || sessions[i]->comm)
            return fail(LFM_EINVAL, "local merge needs sessions of one device without a communicator");
    HIP_TRY(hipSetDevice(sessions[0]->device));
    return merge_group_sparse(sessions, k, k, sides, mode, overlap != 0, nullptr, true);
}

extern "C" int lfm_session_comm_merge_flush(lfm_session *s)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    HIP_TRY(hipSetDevice(s->device));
    LFM_TRY(complete_pending(s));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return LFM_OK;
}

extern "C" int lfm_sessions_merge_local_sparse(lfm_session **sessions, int32_t k, int32_t sides, int32_t mode, int32_t overlap)
{
    if (!sessions || k < 1 || sides < 0 || sides > 3) return fail(LFM_EINVAL, "bad merge arguments");
    for (int i = 0; i < k; ++i)
        if (!sessions[i] || sessions[i]->device != sessions[0]->device || sessions[i]->comm)
            return fail(LFM_EINVAL, "local merge needs sessions of one device without a communicator");
    HIP_TRY(hipSetDevice(sessions[0]->device));
    return merge_group_sparse(sessions, k, k, sides, mode, overlap != 0, nullptr);
}

extern "C" int lfm_sessions_merge_local_flush(lfm_session **sessions, int32_t k)
{
    if (!sessions || k < 1) return fail(LFM_EINVAL, "bad arguments");
    HIP_TRY(hipSetDevice(sessions[0]->device));
    for (int i = 0; i < k; ++i) {
        if (!sessions[i]) return fail(LFM_EINVAL, "null session");
        LFM_TRY(complete_pending(sessions[i]));
        HIP_TRY(hipStreamSynchronize(sessions[i]->stream));
    }
    return LFM_OK;
}

// Resolves librccl now (see rccl()): called by the Python layer of a multi-process job BEFORE it imports torch, so that nothing
// torch brings along can take RCCL's place.  0 when RCCL is available.
extern "C" int lfm_comm_preload(void)
{
    return rccl() ? LFM_OK : fail(LFM_ECOMM, "librccl.so not available");
}

extern "C" int lfm_comm_unique_id(char id[LFM_UNIQUE_ID_BYTES])
{
    Rccl *r = rccl();
    if (!r) return fail(LFM_ECOMM, "librccl.so not available");
    static_assert(sizeof(ncclUniqueId) == LFM_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCCL_TRY(r->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return LFM_OK;
}

extern "C" int lfm_session_comm_init(lfm_session *s, const char id[LFM_UNIQUE_ID_BYTES], int32_t rank,
                                     int32_t nranks)
{
    if (!s || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(LFM_EINVAL, "bad comm arguments");
    Rccl *r = rccl();
    if (!r) return fail(LFM_ECOMM, "librccl.so not available");
    HIP_TRY(hipSetDevice(s->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    NCCL_TRY(r->CommInitRank(&s->comm, nranks, u, rank));
    s->rank = rank;
    s->nranks = nranks;
    // replicated = the item side, and the user side when user features are shared between users
    return lfm_session_merge_begin(s, s->usf.identity ? 1 : 3);
}

extern "C" int lfm_session_comm_merge(lfm_session *s, int32_t sides, int32_t mode)
{
    if (!s || sides < 0 || sides > 3) return fail(LFM_EINVAL, "bad merge arguments");
    if (!s->comm || sides == 0) return LFM_OK;
    HIP_TRY(hipSetDevice(s->device));
    lfm_session *g[1] = {s};
    return merge_group(g, 1, s->nranks, sides, mode);
}

extern "C" int lfm_sessions_merge_local(lfm_session **sessions, int32_t k, int32_t sides, int32_t mode)
{
    if (!sessions || k < 1 || sides < 0 || sides > 3) return fail(LFM_EINVAL, "bad merge arguments");
    for (int i = 0; i < k; ++i)
        if (!sessions[i] || sessions[i]->device != sessions[0]->device || sessions[i]->comm)
            return fail(LFM_EINVAL, "local merge needs sessions of one device without a communicator");
    HIP_TRY(hipSetDevice(sessions[0]->device));
    return merge_group(sessions, k, k, sides, mode);
}

// Owner-sharded item tables over K sessions of ONE device (device.hpp: ItemShards): session j owns the item rows
// [j * rps, (j + 1) * rps), rps = ceil(n_items / K), and every session's epoch kernels gather from and publish to
// the owner's tables.  The one-process form of the multi-GPU decomposition for item sides too large to replicate
// and merge (DESIGN.md "Multi-GPU": C4); lfm_session_share_items_ipc below is the multi-process form.
static int shard_geometry(uint32_t n_items, int k, uint32_t *rps, uint32_t *magic)
{
    if (k < 1 || k > 8) return fail(LFM_EINVAL, "1 to 8 owners");
    *rps = (n_items + (uint32_t)k - 1) / (uint32_t)k;
    // (rows_per_shard = 1 would make magic = 2^32 / 1 + 1 wrap to 1: every row would map to owner 0)
    if (*rps < 2) return fail(LFM_EINVAL, "owner-sharded item tables need at least two item rows per owner");
    if ((uint64_t)(*rps) * (uint64_t)(k - 1) >= (uint64_t)n_items)
        return fail(LFM_EINVAL, "owner-sharded item tables: the last owner's row range would be empty (fewer owners, please)");
    *magic = (uint32_t)((1ull << 32) / (uint64_t)(*rps)) + 1u;
    return LFM_OK;
}

static int shareable(const lfm_session *s)
{
    if (!s || s->scoring_only || s->adadelta || !s->itf.identity || s->comm)
        return fail(LFM_EINVAL, "sharing needs a training session with an identity item side, adagrad, no communicator");
    return LFM_OK;
}

extern "C" int lfm_sessions_share_items_local(lfm_session **sessions, int32_t k)
{
    if (!sessions || k < 1 || k > 8) return fail(LFM_EINVAL, "1 to 8 sessions");
    lfm_session *s0 = sessions[0];
    for (int i = 0; i < k; ++i) {
        lfm_session *s = sessions[i];
        LFM_TRY(shareable(s));
        if (s->device != s0->device || s->n_feat[0] != s0->n_feat[0] || s->d != s0->d)
            return fail(LFM_EINVAL, "sharing needs sessions of one device with the same item side");
        if (s->share || !s->ipc_mappings.empty()) return fail(LFM_EINVAL, "a session's item tables can be shared once");
    }
    uint32_t rps = 0, magic = 0;
    LFM_TRY(shard_geometry((uint32_t)s0->n_feat[0], k, &rps, &magic));
    const uint32_t n_items = (uint32_t)s0->n_feat[0];
    auto group = std::make_shared<lfm_session::ShareGroup>();
    for (int i = 0; i < k; ++i) {
        ItemShards &sh = sessions[i]->shards;
        memset(&sh, 0, sizeof(sh));
        sh.n = k;
        sh.rows_per_shard = rps;
        sh.magic = magic;
        for (int j = 0; j < 8; ++j) {
            lfm_session *owner = sessions[std::min(j, k - 1)];
            const size_t first = (size_t)std::min<uint32_t>((uint32_t)j * rps, n_items);
            sh.W[j] = owner->tab[0][0].p + first * (size_t)owner->d;
            sh.G[j] = owner->tab[0][1].p + first * (size_t)owner->d;
            sh.b[j] = owner->tab[0][3].p + first;
            sh.bG[j] = owner->tab[0][4].p + first;
        }
        sessions[i]->share = group;
        sessions[i]->share_rank = i;
        sessions[i]->share_k = k;
    }
    return LFM_OK;
}

// ---- the multi-PROCESS form: every process exports its four item-side allocations as HIP IPC handles, the handles travel
// over the job's rendezvous (like the RCCL unique id), and every process maps the other owners' allocations: on one
// GPU they are the same memory, across the GPUs of a node they are peer mappings over xGMI (hipIpcOpenMemHandle enables
// peer access lazily).  The tables of large models are allocated uncached (table_alloc_flags), so a float atomic from
// any process or device is performed at the owner's memory.  A pool block is one runtime allocation (pool.hpp), so
// a table's pointer is its allocation's base; the offset is exported anyway and checked.
static const int SHARED_KINDS[4] = {0, 1, 3, 4};  // W, G, b, bG

extern "C" int lfm_session_export_items(lfm_session *s, lfm_item_export *out)
{
    if (!s || !out) return fail(LFM_EINVAL, "null argument");
    LFM_TRY(shareable(s));
    static_assert(sizeof(hipIpcMemHandle_t) <= LFM_IPC_HANDLE_BYTES, "handle size");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    memset(out, 0, sizeof(*out));
    for (int q = 0; q < 4; ++q) {
        float *p = s->tab[0][SHARED_KINDS[q]].p;
        if (!p) return fail(LFM_EINVAL, "item tables not resident");
        void *base = nullptr;
        size_t size = 0;
        HIP_TRY(hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)p));
        hipIpcMemHandle_t h;
        HIP_TRY(hipIpcGetMemHandle(&h, base));
        memcpy(out->handle[q], &h, sizeof(h));
        out->offset[q] = (int64_t)((char *)p - (char *)base);
        out->bytes[q] = (int64_t)(tab_count(s, 0, SHARED_KINDS[q]) * sizeof(float));
    }
    out->n_items = s->n_feat[0];
    out->d = s->d;
    out->device = s->device;
    out->pid = (int64_t)getpid();
    return LFM_OK;
}

extern "C" int lfm_session_share_items_ipc(lfm_session *s, const lfm_item_export *all, int32_t k, int32_t my_rank)
{
    if (!s || !all) return fail(LFM_EINVAL, "null argument");
    LFM_TRY(shareable(s));
    if (my_rank < 0 || my_rank >= k) return fail(LFM_EINVAL, "rank outside [0, k)");
    if (s->share || !s->ipc_mappings.empty()) return fail(LFM_EINVAL, "a session's item tables can be shared once");
    uint32_t rps = 0, magic = 0;
    LFM_TRY(shard_geometry((uint32_t)s->n_feat[0], k, &rps, &magic));
    for (int j = 0; j < k; ++j)
        if (all[j].n_items != s->n_feat[0] || all[j].d != s->d)
            return fail(LFM_EINVAL, "an exported item side has another shape than this session's");
    if (all[my_rank].pid != (int64_t)getpid())
        return fail(LFM_EINVAL, "all[my_rank] is not this process's export");
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t n_items = (uint32_t)s->n_feat[0];
    ItemShards sh;
    memset(&sh, 0, sizeof(sh));
    sh.n = k;
    sh.rows_per_shard = rps;
    sh.magic = magic;
    float **dst[4] = {sh.W, sh.G, sh.b, sh.bG};
    std::vector<void *> mapped;
    auto undo = [&] { for (void *m : mapped) (void)hipIpcCloseMemHandle(m); };
    for (int j = 0; j < k; ++j) {
        const size_t first = (size_t)std::min<uint32_t>((uint32_t)j * rps, n_items);
        for (int q = 0; q < 4; ++q) {
            float *base;
            if (j == my_rank) {
                base = s->tab[0][SHARED_KINDS[q]].p;
            } else {
                hipIpcMemHandle_t h;
                memcpy(&h, all[j].handle[q], sizeof(h));
                void *m = nullptr;
                hipError_t e = hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) {
                    undo();
                    return fail(LFM_ENODEV, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
                }
                mapped.push_back(m);
                base = (float *)((char *)m + all[j].offset[q]);
            }
            dst[q][j] = base + first * (q < 2 ? (size_t)s->d : (size_t)1);
        }
    }
    for (int j = k; j < 8; ++j)
        for (int q = 0; q < 4; ++q) dst[q][j] = dst[q][k - 1];
    s->shards = sh;
    s->ipc_mappings = mapped;
    s->share_rank = my_rank;
    s->share_k = k;
    return LFM_OK;
}

// After training: the owners' rows copied into this session's own (otherwise stale) item tables, device to device through
// the mappings, so that check_finite / sync_to_host / predict see the whole model.  All processes call it between two
// barriers of their rendezvous (nobody trains while rows are copied).
extern "C" int lfm_session_gather_shared_items(lfm_session *s)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    if (s->shards.n <= 0 || s->share_rank < 0) return fail(LFM_EINVAL, "the session's item tables are not owner-sharded");
    if (s->share && s->share->broken) return fail(LFM_EINVAL, "a session of the sharing group has been destroyed");
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t n_items = (uint32_t)s->n_feat[0], rps = s->shards.rows_per_shard;
    float *const *src[4] = {s->shards.W, s->shards.G, s->shards.b, s->shards.bG};
    for (int j = 0; j < s->shards.n; ++j) {
        if (j == s->share_rank) continue;
        const size_t first = (size_t)std::min<uint32_t>((uint32_t)j * rps, n_items);
        const size_t rows = (size_t)std::min<uint32_t>(rps, n_items - (uint32_t)first);
        for (int q = 0; q < 4; ++q) {
            const size_t w = q < 2 ? (size_t)s->d : (size_t)1;
            HIP_TRY(hipMemcpyAsync(s->tab[0][SHARED_KINDS[q]].p + first * w, src[q][j], rows * w * sizeof(float),
                                   hipMemcpyDeviceToDevice, s->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    return LFM_OK;
}

extern "C" int lfm_session_comm_any(lfm_session *s, int32_t flag)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    if (!s->comm) return flag ? 1 : 0;
    HIP_TRY(hipSetDevice(s->device));
    int v = flag ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(s->flag.p, &v, sizeof(int), hipMemcpyHostToDevice, s->stream));
    NCCL_TRY(rccl()->AllReduce(s->flag.p, s->flag.p, 1, ncclInt32, ncclMax, s->comm, s->stream));
    HIP_TRY(hipMemcpyAsync(&v, s->flag.p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return v ? 1 : 0;
}

extern "C" int lfm_session_comm_barrier(lfm_session *s)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    if (!s->comm) return LFM_OK;
    HIP_TRY(hipSetDevice(s->device));
    NCCL_TRY(rccl()->AllReduce(s->flag.p, s->flag.p, 1, ncclInt32, ncclSum, s->comm, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return LFM_OK;
}

// ------------------------------------------------ input integrity (debug) ---
// LIGHTFM_AMD_VALIDATE=1: every lfm_session_epoch checksums the READ-ONLY device inputs (COO arrays,
// packed records, positives lookup, feature CSRs, loss table) before its first launch and after its
// last one, and range-checks the shuffle slot.  A buffer whose checksum changed inside the epoch was
// written by one of the epoch's kernels; one that changed since the previous epoch of this session
// was overwritten in between (another session, an upload to a stale address, the allocator).  The
// call then fails with LFM_ECORRUPT naming the buffer instead of running kernels on garbage.
__global__ void checksum_kernel(const uint32_t *p, int64_t n_words, unsigned long long *out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0ull;
    for (int64_t j = t; j < n_words; j += st) acc += (unsigned long long)p[j] * (2ull * (unsigned long long)j + 1ull);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0 && acc) atomicAdd(out, acc);
}
// out[0] = entries outside [0, n), out[1] = sum of the entries (a permutation sums to n (n - 1) / 2)
__global__ void shuffle_check_kernel(const int32_t *p, int64_t n, unsigned long long *out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0ull, sum = 0ull;
    for (int64_t j = t; j < n; j += st) {
        const int32_t v = p[j];
        if (v < 0 || (int64_t)v >= n) ++bad; else sum += (unsigned long long)v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        bad += __shfl_xor(bad, off, WAVE);
        sum += __shfl_xor(sum, off, WAVE);
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        if (bad) atomicAdd(out, bad);
        if (sum) atomicAdd(out + 1, sum);
    }
}

static bool validate_enabled() { return validate_level() >= 1; }

// when: 0 = before the epoch's first launch, 1 = after its last one (stream synchronised by the caller)
static int validate_inputs(lfm_session *s, int slot, int when, bool recs_in_use)
{
    struct Item { const char *name; const void *p; size_t bytes; const std::vector<unsigned char> *shadow; };
    std::vector<Item> items;
    auto add = [&](const char *name, const void *p, size_t bytes, const std::vector<unsigned char> *shadow = nullptr) {
        if (p && bytes >= 4) items.push_back(Item{name, p, bytes & ~(size_t)3, shadow});
    };
    add("user_ids", s->user_ids.p, s->user_ids.n * 4, &s->user_ids.shadow);
    add("item_ids", s->item_ids.p, s->item_ids.n * 4, &s->item_ids.shadow);
    add("Y", s->Y.p, s->Y.n * 4, &s->Y.shadow);
    add("sample_weight", s->weight.p, s->weight.n * 4, &s->weight.shadow);
    if (recs_in_use && s->recs_valid) add("records", s->recs.p, s->recs.n * sizeof(int4));
    add("positives.indptr", s->pos.indptr.p, s->pos.indptr.p ? ((size_t)s->pos.rows + 1) * 4 : 0, &s->pos.indptr.shadow);
    add("positives.indices", s->pos.indices.p, (size_t)s->pos.nnz * 4, &s->pos.indices.shadow);
    add("item_features.indptr", s->itf.indptr.p, s->itf.indptr.n * 4, &s->itf.indptr.shadow);
    add("item_features.indices", s->itf.indices.p, s->itf.indices.n * 4, &s->itf.indices.shadow);
    add("item_features.data", s->itf.data.p, s->itf.data.n * 4, &s->itf.data.shadow);
    add("user_features.indptr", s->usf.indptr.p, s->usf.indptr.n * 4, &s->usf.indptr.shadow);
    add("user_features.indices", s->usf.indices.p, s->usf.indices.n * 4, &s->usf.indices.shadow);
    add("user_features.data", s->usf.data.p, s->usf.data.n * 4, &s->usf.data.shadow);
    if (validate_level() >= 2) {
        const char *w = when ? "after the epoch" : "before the epoch";
        LFM_TRY(s->user_ids.verify(w));
        LFM_TRY(s->item_ids.verify(w));
        LFM_TRY(s->Y.verify(w));
        LFM_TRY(s->weight.verify(w));
        for (const DevCsr *f : {&s->pos, &s->itf, &s->usf}) {
            LFM_TRY(f->indptr.verify(w));
            LFM_TRY(f->indices.verify(w));
            LFM_TRY(f->data.verify(w));
        }
    }
    const size_t k = items.size();
    LFM_TRY(s->guard_dev.alloc(k + 2));
    HIP_TRY(hipMemsetAsync(s->guard_dev.p, 0, (k + 2) * sizeof(unsigned long long), s->stream));
    for (size_t i = 0; i < k; ++i) {
        const int64_t words = (int64_t)(items[i].bytes / 4);
        checksum_kernel<<<grid_for(words), 256, 0, s->stream>>>((const uint32_t *)items[i].p, words, s->guard_dev.p + i);
    }
    if (s->n > 0)
        shuffle_check_kernel<<<grid_for(s->n), 256, 0, s->stream>>>(s->shuffles[slot]->p, s->n, s->guard_dev.p + k);
    HIP_TRY(hipGetLastError());
    std::vector<unsigned long long> sums(k + 2);
    HIP_TRY(hipMemcpyAsync(sums.data(), s->guard_dev.p, (k + 2) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    char msg[256];
    if (s->n > 0) {
        const unsigned long long want = (unsigned long long)s->n * (unsigned long long)(s->n - 1) / 2ull;
        if (sums[k] != 0ull || sums[k + 1] != want) {
            snprintf(msg, sizeof(msg), "LIGHTFM_AMD_VALIDATE: shuffle slot %d is not a permutation of [0, %lld): %llu entries out of "
                     "range, sum %llu instead of %llu (%s epoch %lld)", slot, (long long)s->n, sums[k], sums[k + 1], want,
                     when ? "after" : "before", (long long)s->epochs_run);
            return fail(LFM_ECORRUPT, msg);
        }
    }
    // level 2: what a KERNEL reads (through the caches) against the uploaded bytes (the shadow copy)
    for (size_t i = 0; i < k && validate_level() >= 2; ++i) {
        if (!items[i].shadow || items[i].shadow->size() < items[i].bytes) continue;
        const uint32_t *w = (const uint32_t *)items[i].shadow->data();
        unsigned long long want = 0ull;
        for (size_t j = 0; j < items[i].bytes / 4; ++j) want += (unsigned long long)w[j] * (2ull * (unsigned long long)j + 1ull);
        if (want != sums[i]) {
            snprintf(msg, sizeof(msg), "LIGHTFM_AMD_VALIDATE: a kernel reads device buffer '%s' (%zu bytes at %p) differently from what was "
                     "uploaded (checksum %016llx, uploaded %016llx) %s epoch %lld although a read-back copy matches",
                     items[i].name, items[i].bytes, items[i].p, sums[i], want, when ? "after" : "before", (long long)s->epochs_run);
            fprintf(stderr, "%s\n", msg);
            return fail(LFM_ECORRUPT, msg);
        }
    }
    for (size_t i = 0; i < k; ++i) {
        for (const auto &g : s->guard_sums) {
            if (g.p == items[i].p && g.bytes == items[i].bytes && g.sum != sums[i]) {
                snprintf(msg, sizeof(msg), "LIGHTFM_AMD_VALIDATE: read-only device buffer '%s' (%zu bytes at %p) changed %s: checksum "
                         "%016llx -> %016llx", items[i].name, items[i].bytes, items[i].p,
                         when ? "INSIDE the epoch (written by one of its kernels)" : "BETWEEN two epochs of this session",
                         g.sum, sums[i]);
                return fail(LFM_ECORRUPT, msg);
            }
        }
    }
    s->guard_sums.clear();
    for (size_t i = 0; i < k; ++i) s->guard_sums.push_back(lfm_session::GuardSum{items[i].p, items[i].bytes, sums[i]});
    return LFM_OK;
}

__global__ void bias_pack_kernel(const float *b, const float *bG, float *pairs, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        pairs[2 * i] = b[i];
        pairs[2 * i + 1] = bG[i];
    }
}
__global__ void bias_unpack_kernel(float *b, float *bG, const float *pairs, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        b[i] = pairs[2 * i];
        bG[i] = pairs[2 * i + 1];
    }
}
// (b != nullptr: the bias cells ride in slot d of each half)
__global__ void row_pack_kernel(const float *W, const float *G, const float *b, const float *bG, float *pairs, int64_t rows, int d)
{
    const int64_t cells = rows * 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i >> 4;
        const int c = (int)(i & 15);
        const bool bias = b && c == d;
        pairs[r * 32 + c] = c < d ? W[r * d + c] : (bias ? b[r] : 0.0f);
        pairs[r * 32 + 16 + c] = c < d ? G[r * d + c] : (bias ? bG[r] : 1.0f);
    }
}
__global__ void row_unpack_kernel(float *W, float *G, float *b, float *bG, const float *pairs, int64_t rows, int d)
{
    const int64_t cells = rows * 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i >> 4;
        const int c = (int)(i & 15);
        if (c < d) {
            W[r * d + c] = pairs[r * 32 + c];
            G[r * d + c] = pairs[r * 32 + 16 + c];
        } else if (b && c == d) {
            b[r] = pairs[r * 32 + c];
            bG[r] = pairs[r * 32 + 16 + c];
        }
    }
}
__global__ void copy_strided_kernel(float *dst, const float *src, int64_t n, int stride)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i * stride];
}

// sum_u c_u^2 / n^2 over the uploaded COO, c_u = interactions of user u: the probability that two interactions drawn at
// random belong to the same user.  Times the interactions in flight it is the share of interactions that have ANOTHER
// interaction of their user in flight with them -- what the plain-store user rows (FitArgs::user_store) can lose an update
// to.  One histogram pass with integer atomics (ML-20M: 20 M adds on 138 k counters), once per uploaded COO.
__global__ void user_count_kernel(const int32_t *user_ids, int64_t n, int32_t *counts, int32_t n_users)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t u = user_ids[i];
        if (u >= 0 && u < n_users) atomicAdd(counts + u, 1);
    }
}

__global__ void count_sumsq_kernel(const int32_t *counts, int64_t m, unsigned long long *out)
{
    unsigned long long acc = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long c = (unsigned long long)counts[i];
        acc += c * c;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

static int user_pair_share(lfm_session *s, double *out)
{
    if (s->user_pair_share < 0.0) {
        if (s->n <= 0 || s->n_feat[1] <= 0) {
            s->user_pair_share = 1.0;
        } else {
            DBuf<int32_t> counts;
            DBuf<unsigned long long> acc;
            LFM_TRY(counts.alloc((size_t)s->n_feat[1]));
            LFM_TRY(acc.alloc(1));
            HIP_TRY(hipMemsetAsync(counts.p, 0, (size_t)s->n_feat[1] * sizeof(int32_t), s->stream));
            HIP_TRY(hipMemsetAsync(acc.p, 0, sizeof(unsigned long long), s->stream));
            const int g1 = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (s->n + 255) / 256));
            user_count_kernel<<<g1, 256, 0, s->stream>>>(s->user_ids.p, s->n, counts.p, s->n_feat[1]);
            const int g2 = (int)std::max<int64_t>(1, std::min<int64_t>(1024, ((int64_t)s->n_feat[1] + 255) / 256));
            count_sumsq_kernel<<<g2, 256, 0, s->stream>>>(counts.p, (int64_t)s->n_feat[1], acc.p);
            HIP_TRY(hipGetLastError());
            unsigned long long sq = 0ull;
            HIP_TRY(hipMemcpyAsync(&sq, acc.p, sizeof(sq), hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            s->user_pair_share = (double)sq / ((double)s->n * (double)s->n);
        }
    }
    *out = s->user_pair_share;
    return LFM_OK;
}

// The hot set of the resident item feature matrix: columns an interaction touches with probability >= 1/512 -- 2 x
// occurrences / rows (the positive and the negative item), the rule of lightfm_amd/distributed.py: hot_rows -- and at
// least twice; taken when they carry >= 1/4 of the matrix's entries (C3: 1 128 tag columns, 8 of 9 entries of every row;
// C5's hashed columns: none) and fit the LDS slices (<= 4 864 rows, the most frequent ones beyond that).
static int build_hot_set(lfm_session *s)
{
    lfm_session::HotSet &h = s->hot;
    h.state = -1;
    h.n = 0;
    const int32_t cols = s->n_feat[0];
    if (s->itf.identity || s->itf.nnz <= 0 || s->itf.rows <= 0 || cols <= 0 || (s->d & 1)) return LFM_OK;
    std::vector<int32_t> counts((size_t)cols);
    {
        DBuf<int32_t> dc;
        LFM_TRY(dc.alloc((size_t)cols));
        HIP_TRY(launch_column_counts(s->itf.indices.p, s->itf.nnz, cols, dc.p, s->stream));
        HIP_TRY(hipMemcpyAsync(counts.data(), dc.p, (size_t)cols * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    std::vector<int32_t> hot;
    for (int32_t c = 0; c < cols; ++c)
        if (counts[c] >= 2 && 2.0 * (double)counts[c] * 512.0 >= (double)s->itf.rows) hot.push_back(c);
    const size_t cap = 4864;
    if (hot.size() > cap) {
        std::partial_sort(hot.begin(), hot.begin() + cap, hot.end(), [&](int32_t x, int32_t y) { return counts[x] > counts[y] || (counts[x] == counts[y] && x < y); });
        hot.resize(cap);
        std::sort(hot.begin(), hot.end());
    }
    int64_t covered = 0;
    for (int32_t c : hot) covered += counts[c];
    h.share = (double)covered / (double)s->itf.nnz;
    const int cs = hot.empty() ? 0 : hot_slice_components((int)hot.size(), s->d);
    if (hot.empty() || cs == 0 || 4 * covered < s->itf.nnz) return LFM_OK;
    std::vector<int32_t> slot((size_t)cols, -1);
    for (size_t j = 0; j < hot.size(); ++j) slot[(size_t)hot[j]] = (int32_t)j;
    LFM_TRY(h.slot.upload(slot.data(), slot.size()));
    LFM_TRY(h.rows.upload(hot.data(), hot.size()));
    LFM_TRY(h.snapW.alloc(hot.size() * (size_t)s->d));
    LFM_TRY(h.snapG.alloc(hot.size() * (size_t)s->d));
    LFM_TRY(h.snapb.alloc(hot.size()));
    LFM_TRY(h.snapbG.alloc(hot.size()));
    h.n = (int)hot.size();
    h.cs = cs;
    h.state = 1;
    return LFM_OK;
}

// Bias pairs (lfm_session::bias_pairs): the live bias cells move between tab[side][3 / 4] and the (b, bG) pair tables
static int bias_pairs_pack(lfm_session *s, hipStream_t st)
{
    for (int side = 0; side < 2; ++side) {
        const int64_t n = (int64_t)s->n_feat[side];
        s->bias_pairs[side].flags = s->tab[side][3].flags;
        LFM_TRY(s->bias_pairs[side].alloc((size_t)(2 * n)));
        if (n) bias_pack_kernel<<<(int)std::min<int64_t>(1024, (n + 255) / 256), 256, 0, st>>>(s->tab[side][3].p, s->tab[side][4].p, s->bias_pairs[side].p, n);
    }
    HIP_TRY(hipGetLastError());
    s->pairs_live = true;
    return LFM_OK;
}

static int bias_pairs_unpack(lfm_session *s, hipStream_t st)
{
    for (int side = 0; side < 2; ++side) {
        const int64_t n = (int64_t)s->n_feat[side];
        if (n && s->bias_pairs[side].p)
            bias_unpack_kernel<<<(int)std::min<int64_t>(1024, (n + 255) / 256), 256, 0, st>>>(s->tab[side][3].p, s->tab[side][4].p, s->bias_pairs[side].p, n);
    }
    HIP_TRY(hipGetLastError());
    s->pairs_live = false;
    return LFM_OK;
}

static int row_pairs_pack(lfm_session *s, hipStream_t st, bool with_bias)
{
    for (int side = 0; side < 2; ++side) {
        const int64_t n = (int64_t)s->n_feat[side];
        s->row_pairs[side].flags = s->tab[side][0].flags;
        LFM_TRY(s->row_pairs[side].alloc((size_t)(32 * n)));
        if (n) row_pack_kernel<<<(int)std::min<int64_t>(4096, (16 * n + 255) / 256), 256, 0, st>>>(
            s->tab[side][0].p, s->tab[side][1].p, with_bias ? s->tab[side][3].p : nullptr, with_bias ? s->tab[side][4].p : nullptr,
            s->row_pairs[side].p, n, s->d);
    }
    HIP_TRY(hipGetLastError());
    s->rows_live = true;
    s->rows_bias = with_bias;
    return LFM_OK;
}

static int row_pairs_unpack(lfm_session *s, hipStream_t st)
{
    for (int side = 0; side < 2; ++side) {
        const int64_t n = (int64_t)s->n_feat[side];
        if (n && s->row_pairs[side].p)
            row_unpack_kernel<<<(int)std::min<int64_t>(4096, (16 * n + 255) / 256), 256, 0, st>>>(
                s->tab[side][0].p, s->tab[side][1].p, s->rows_bias ? s->tab[side][3].p : nullptr, s->rows_bias ? s->tab[side][4].p : nullptr,
                s->row_pairs[side].p, n, s->d);
    }
    HIP_TRY(hipGetLastError());
    s->rows_live = s->rows_bias = false;
    return LFM_OK;
}

// ------------------------------------------------------------------ epoch ---

static void tile_geometry(int d, int want_rows, int *rows, int *stride)
{
    int ts = ((d + 1 + 3) / 4) * 4;  // bias in column d, rows 16-byte aligned
    int budget_rows = (6 * 1024) / (ts * 4);
    int r = std::min(want_rows, std::min(16, budget_rows));
    r = std::max(r, 3);
    *rows = r;
    *stride = ts;
}

extern "C" int lfm_session_epoch(lfm_session *s, int32_t loss, int32_t slot, double item_alpha,
                                 double user_alpha, int32_t k, int32_t n_positives,
                                 const uint32_t *seeds, int32_t n_seeds, lfm_opts *opts)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    if (s->scoring_only) return fail(LFM_EINVAL, "a scoring session (lfm_session_create_scoring) cannot train");
    if (s->stream2) (void)hipStreamSynchronize(s->stream2);  // (left running only by an epoch that failed half-way)
    if (s->pairs_live || s->rows_live) {  // (likewise: live cells are still in the pair tables)
        HIP_TRY(hipSetDevice(s->device));
        if (s->pairs_live) LFM_TRY(bias_pairs_unpack(s, s->stream));
        if (s->rows_live) LFM_TRY(row_pairs_unpack(s, s->stream));
    }
    if (s->share && s->share->broken)
        return fail(LFM_EINVAL, "a session this one shares item tables with has been destroyed: its rows are gone");
    if (loss < 0 || loss > 3) return fail(LFM_EINVAL, "unknown loss");
    if (slot < 0 || slot >= (int)s->shuffles.size() || s->shuffles[slot]->n != (size_t)s->n)
        return fail(LFM_EINVAL, "shuffle slot not uploaded");
    const bool needs_rng = loss != LFM_LOSS_LOGISTIC;
    if (needs_rng && (!seeds || n_seeds < 1)) return fail(LFM_EINVAL, "seeds required");
    if (loss != LFM_LOSS_WARP_KOS && s->n && (!s->item_ids.p || !s->Y.p))
        return fail(LFM_EINVAL, "item_ids / Y not uploaded");
    if (loss != LFM_LOSS_WARP_KOS && s->n && !s->weight_aliases_Y && !s->weight.p)
        return fail(LFM_EINVAL, "sample_weight not uploaded");
    if (needs_rng && !s->pos.indptr.p && s->pos.rows == 0 && s->n)
        return fail(LFM_EINVAL, "positives lookup not uploaded");
    if (loss == LFM_LOSS_WARP_KOS && (k < 1 || n_positives < 1)) return fail(LFM_EINVAL, "k and n must be positive");
    if (s->max_sampled < 0) return fail(LFM_EINVAL, "max_sampled must not be negative");
    lfm_opts local;
    memset(&local, 0, sizeof(local));
    if (!opts) opts = &local;
    const bool serial = opts->mode == LFM_MODE_SERIAL;
    HIP_TRY(hipSetDevice(s->device));
    if (slot < (int)s->shuffle_ready.size() && s->shuffle_ready[slot])  // a permutation written ahead on the auxiliary stream
        HIP_TRY(hipStreamWaitEvent(s->stream, s->shuffle_ready[slot], 0));

    FitArgs a;
    memset(&a, 0, sizeof(a));
    a.itf = s->itf.view();
    a.usf = s->usf.view();
    a.pos = s->pos.view();
    a.m = s->dmodel();
    a.user_ids = s->user_ids.p;
    a.item_ids = s->item_ids.p;
    a.Y = s->Y.p;
    a.weight = s->weight_aliases_Y ? s->Y.p : s->weight.p;
    a.shuffle = s->shuffles[slot]->p;
    a.n = s->n;
    a.item_alpha = item_alpha;
    a.user_alpha = user_alpha;
    a.serial = serial ? 1 : 0;
    // kernel-side encoding: 0 atomic deltas, 1 plain stores, 2 no writes
    a.update_mode = serial ? 1 : (opts->update_mode == 1 ? 1 : (opts->update_mode == 2 ? 2 : 0));
    a.debug = opts->debug;
    a.k = k;
    a.n_pos = n_positives;
    a.counters = s->counters.p;
    a.reg_live = s->reg_live.p;
    // in_positives pre-filter of the tile kernel (debug bit 8 = 256: off; bit 9 = 512: probed for every candidate
    // together with its row instead of after the scoring pass for the violators only)
    a.bloom = (s->bloom_valid && !(opts->debug & 256)) ? s->bloom.p : nullptr;
    a.shards = s->shards;
    int base_user_store = 0;
    {
        // The user row of an update by plain stores instead of float atomics (FitArgs::user_store; warp_tile_ahead.hpp): identity
        // user features (a user's row is touched by that user's interactions alone), uncached tables -- W AND G of the user side
        // (a store is then visible to every XCD) --, atomic publication, adagrad.  What a plain read-modify-write loses is one
        // of two updates of the SAME user that are in flight together, so the rule is about the data's own collision rate:
        // share = (interactions in flight at full residency, 48 per CU) x sum_u c_u^2 / n^2 = the fraction of interactions that
        // have another interaction of their user in flight with them (user_pair_share above; uniform activity gives in-flight /
        // users, a heavy tail more).  Measured precision@10 cost: -0.0004 at share 0.21 (C2: log-normal activity, 8 seeds),
        // -0.0032 at share 1.57 (the 1/8-scale gate of tests/test_precision_parity.py, which failed) -- ~0.002 per unit, so the
        // switch is taken up to share 0.3 (round 5's rule counted users only and assumed uniform activity: ADVICE r5).  And
        // only for a model that lives in the Infinity Cache -- on the C4 shard (3.3 GB of tables) the same switch LOST 5-10 %
        // (uncached partial-line stores to HBM; profiles/r05_visit_f.txt).  lfm_opts.debug bit 11 (2048) forces it for uncached
        // tables of any size and collision rate, bit 12 (4096) switches it off.
        // And only for rows of more than 48 floats: the cost of a lost user update grows as the rows shrink, and the
        // gain vanishes -- at the full ML-20M shape, 16 seeds per arm (profiles/r06_narrow_quality20m.txt): d = 32 stores -0.0006
        // against atomics -0.0002 at the SAME kernel time (11.5 ms per epoch); d = 10 stores -0.0041 / -0.0032 / -0.0023 (24 576 /
        // 12 288 / 8 192 in flight) against atomics -0.0004 ... +0.0001; d = 64: -0.0002 for +15 % (profiles/r05_ustore_quality.txt).
        // Kernel time, stores against atomics (profiles/r06_ustore_ab.txt): d = 48: +-1 % for WARP / BPR / logistic alike; d = 64: WARP
        // +10 %, BPR +20 %, logistic +7 %; d = 128: BPR +24 %, logistic +30 % -- so the line is drawn above 48.
        size_t bytes = 0;
        for (int side = 0; side < 2; ++side)
            for (int kk = 0; kk < 6; ++kk) bytes += s->tab[side][kk].n * sizeof(float);
        const bool eligible = !serial && a.update_mode == 0 && s->usf.identity && s->tab[1][0].flags != 0 && s->tab[1][1].flags != 0 &&
                              !s->adadelta && s->shards.n == 0;
        bool rare_collisions = false;
        if (eligible && !(opts->debug & (4096 | 2048)) && bytes <= ((size_t)192 << 20)) {
            double share = 1.0;
            LFM_TRY(user_pair_share(s, &share));
            // (interactions in flight per CU at full residency: 48 for the tile kernels, 128 for the narrow-model kernel --
            // four workgroups of eight per wavefront pass, warp_tile_narrow.hpp)
            const bool narrow = loss == LFM_LOSS_WARP && s->itf.identity && s->max_sampled == 10 && (opts->first_batch <= 0 || opts->first_batch == 10) &&
                                opts->warp_kernel == 0 && !(opts->debug & (1024 | 512 | 64)) && item_alpha == 0.0 && user_alpha == 0.0 &&
                                warp_tile_narrow_smem(s->d, 10, 10, (int64_t)s->itf.rows) != 0;
            static const int narrow_blocks = [] { const char *e = getenv("LIGHTFM_AMD_NARROW_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4; }();
            const double per_cu = narrow ? 32.0 * narrow_blocks : 48.0;
            rare_collisions = s->d_host > 48 && share * per_cu * (double)std::max(1, s->cus) <= 0.3;
        }
        base_user_store = a.user_store = (eligible && !(opts->debug & 4096) && (rare_collisions || (opts->debug & 2048))) ? 1 : 0;
    }

    // WARP loss term per sampled count, evaluated with the HOST libm so the device
    // never calls log(): PYX:881 / C_OMP:7446 (WARP), PYX:1039 / C_OMP:8452 (k-OS).
    {
        std::vector<double> lt((size_t)s->max_sampled + 1, 0.0);
        int rows = s->itf.rows;
        for (int q = 1; q <= s->max_sampled; ++q) {
            double fl = floor((double)((long)(rows - 1) / (long)q));
            lt[q] = (loss == LFM_LOSS_WARP_KOS) ? log(fl) : log(fl > 1.0 ? fl : 1.0);
        }
        LFM_TRY(s->logtab.upload(lt.data(), lt.size()));
        a.logtab = s->logtab.p;
    }
    {
        std::vector<uint32_t> sd(seeds ? seeds : nullptr, seeds ? seeds + n_seeds : nullptr);
        if (sd.empty()) sd.push_back(0u);
        LFM_TRY(s->seeds.upload(sd.data(), sd.size()));
        a.seeds = s->seeds.p;
    }
    int want_rows = (loss == LFM_LOSS_WARP || loss == LFM_LOSS_WARP_KOS) ? s->max_sampled + 2 : 3;
    if (loss == LFM_LOSS_WARP_KOS) want_rows = std::max(want_rows, std::min(n_positives, 15) + 1);
    tile_geometry(s->d, want_rows, &a.tile_rows, &a.tile_stride);
    a.first_batch = opts->first_batch > 0 ? opts->first_batch : s->max_sampled;  // auto: score every allowed draw in one batch
    a.first_batch = std::max(1, std::min(a.first_batch, a.tile_rows - 2));
    a.pair_cap = (loss == LFM_LOSS_WARP_KOS) ? ((n_positives + 3) / 4) * 4 : 0;
    size_t smem = sizeof(float) * ((size_t)WAVES_PER_BLOCK * a.tile_rows * a.tile_stride +
                                   (size_t)WAVES_PER_BLOCK * 2 * a.pair_cap);
    if (smem > 160 * 1024) return fail(LFM_EUNSUPPORTED, "k-OS n too large for the LDS pair buffer");

    // Interactions allowed in flight (between reading the weights and publishing the update).
    // Every one of them is computed against weights the others are changing, and all their
    // updates land.  That is harmless once the model has left its initial state, and ruinous
    // before: with all scores ~0 every interaction violates the margin with the maximum loss, and
    // hundreds of concurrent maximum steps on a popular row (or a shared feature row) overshoot
    // it for good -- its accumulators then freeze the damage.  So concurrency is RAMPED with the
    // training history: at most (interactions already trained on) / ramp_k in flight, up to the
    // whole chip (DESIGN.md "Hogwild at GPU width": measured precision@10 parity).  An explicit
    // lfm_opts.max_waves is a fixed cap instead (experiments, tests).
    const int64_t ramp_k = opts->ramp_k > 0 ? opts->ramp_k : 32;
    const int64_t history0 = opts->history > 0 ? opts->history : 0;
    const bool fixed_cap = opts->max_waves > 0;
    // Shared feature rows (hybrid models) stay sensitive after the ramp: every interaction in
    // flight touches avg-nnz of a side's feature rows.  Steady-state bound for such a side:
    // (feature rows) / (avg nnz per row) interactions in flight (measured at the ML-100k shape
    // with 40 tags x 4 per item; identity sides are not bounded).
    int64_t shared_cap = INT64_MAX / 4;
    if (opts->shared_cap > 0) shared_cap = opts->shared_cap;
    for (const DevCsr *f : {&s->itf, &s->usf}) {
        if (opts->shared_cap != 0 || f->identity || f->rows <= 0 || f->nnz <= 0) continue;
        const double avg = (double)f->nnz / (double)f->rows;
        shared_cap = std::min<int64_t>(shared_cap, std::max<int64_t>(64, (int64_t)((double)f->cols / std::max(1.0, avg))));
    }
    // ... and no more interactions in flight than the smaller side has rows: every interaction in flight updates one user
    // row and two item rows, so beyond that every row has several concurrent writers whatever the history (ML-100k, 943 x
    // 1 682, is a BASELINE shape; round 5 measured -0.0020 / -0.0021 precision@10 at 300 x 120 / 1 000 x 400 without
    // this bound: profiles/r05_visit_f.txt).  No BASELINE bench shape is bound by it (C2: 26 744 rows > 12 288 in flight).
    const int64_t rows_cap = std::max<int64_t>(8, std::min<int64_t>(s->n_feat[0], s->n_feat[1]));
    auto allowed_in_flight = [&](int64_t done_this_epoch) -> int64_t {
        if (fixed_cap) return opts->max_waves;
        if (opts->ramp_k < 0) return INT64_MAX / 4;  // ramp disabled
        return std::min<int64_t>(std::min(shared_cap, rows_cap), std::max<int64_t>(8, (history0 + done_this_epoch) / ramp_k));
    };

    // Parallel WARP over identity features, with or without L2 regularisation (BASELINE configs C2/C4):
    // the lane-group tile kernel (warp_tile_kernel.hpp), NG interactions per wavefront pass.
    // NG = 4 is the most instruction-efficient mapping; when a launch may keep only few
    // interactions in flight, fewer per wavefront buy more wavefronts (latency hiding).
    struct TilePlan { bool ok = false, dma4 = false, ahead = false, narrow = false; size_t smem = 0; int rows = 0, stride = 0, vec = 0, first_batch = 1; };
    TilePlan tile[5];  // indexed by NG (1, 2, 4)
    bool use_tile = false;
    // fit_logistic / fit_bpr of a NARROW identity model (d <= 12: the reference's default LightFM() is logistic at 10): the lane-group
    // kernels on one-line-per-feature rows (logistic_tile.hip) -- adagrad, no L2 penalty, atomic publication
    const size_t ltile_smem = loss == LFM_LOSS_LOGISTIC ? logistic_tile_smem(s->d, (int64_t)s->n_feat[1], (int64_t)s->n_feat[0])
                              : (loss == LFM_LOSS_BPR ? bpr_tile_smem(s->d, (int64_t)s->n_feat[1], (int64_t)s->n_feat[0]) : 0);
    // fit_bpr / fit_logistic of a WIDER identity model: the BPR / logistic instantiations of the tile kernel (warp_tile_bpr.hip) --
    // adagrad, with or without an L2 penalty; BPR: two candidates per batch (LIGHTFM_AMD_BPR_WIDE_TILE=0: the row-stream kernel)
    const bool bpr_wide_env = [] { const char *e = getenv("LIGHTFM_AMD_BPR_WIDE_TILE"); return !e || atoi(e) != 0; }();  // (per epoch call: the tests switch arms inside one process)
    const bool lgt_tile = loss == LFM_LOSS_LOGISTIC;
    // (rows of up to 12 floats without an L2 penalty belong to the narrow lane-group kernels above)
    const bool narrow_scope = s->d <= 12 && item_alpha == 0.0 && user_alpha == 0.0 && a.update_mode == 0;
    const bool bpr_tile = (loss == LFM_LOSS_BPR || lgt_tile) && bpr_wide_env && !narrow_scope && s->d <= 256 && !s->adadelta &&
                          opts->feat_kernel == 0 && s->shards.n == 0 && s->n > 0 &&
                          (lgt_tile || (s->pos.indptr.p != nullptr && s->item_ids.p != nullptr));
    if (!serial && (loss == LFM_LOSS_WARP || bpr_tile) && opts->warp_kernel != 1 && s->itf.identity &&
        s->usf.identity && s->itf.rows >= 2 && !(s->adadelta && (item_alpha != 0.0 || user_alpha != 0.0))) {
        const int forced = opts->debug & 7;  // experiment override: 1, 2 or 4
        for (int ng : {4, 2, 1}) {
            if (forced && ng != forced) continue;
            TilePlan &t = tile[ng];
            // four interactions per pass: rows go memory -> LDS by LDS-DMA (debug bit 6: the
            // register-staged variant instead)
            t.dma4 = ng == 4 && !(opts->debug & 64);
            const int batch = bpr_tile ? (lgt_tile ? 1 : 2) : s->max_sampled;  // candidate rows of a group's tile
            t.smem = warp_tile_geometry(s->d, batch, ng, &t.rows, &t.stride, &t.vec, t.dma4);
            if (t.smem == 0 && t.dma4) {
                t.dma4 = false;
                t.smem = warp_tile_geometry(s->d, batch, ng, &t.rows, &t.stride, &t.vec, false);
            }
            if (t.smem == 0) continue;
            if (bpr_tile && t.vec != 4) continue;  // (instantiated with four floats per lane: d <= 64 / 128 / 256 at 4 / 2 / 1 interactions per pass)
            t.ok = true;
            t.first_batch = opts->first_batch > 0 ? opts->first_batch : batch;
            t.first_batch = std::max(1, std::min(t.first_batch, t.rows - 1));
            // the steady-state variant with the next pass's gather issued inside the current pass (warp_tile_ahead.hpp):
            // adagrad, no regularisation, max_sampled = 10 in one batch; debug bit 10 (1024) keeps the plain kernel
            if (!bpr_tile && t.dma4 && !s->adadelta && item_alpha == 0.0 && user_alpha == 0.0 && opts->warp_kernel != 2 &&
                a.update_mode == 0 && !(opts->debug & (1024 | 512))) {  // (atomic publication only: no per-publication mode switch)
                const size_t ahead = warp_tile_ahead_smem(s->d, s->max_sampled, t.first_batch);
                if (ahead) {
                    t.ahead = true;
                    t.smem = ahead;
                    // narrow models (rows of <= 16 floats: the reference's default width): two interactions per lane group,
                    // eight per wavefront pass (warp_tile_narrow.hpp)
                    const size_t narrow = s->shards.n == 0 ? warp_tile_narrow_smem(s->d, s->max_sampled, t.first_batch, (int64_t)s->itf.rows) : 0;
                    if (narrow) {
                        t.narrow = true;
                        t.smem = narrow;
                    }
                }
            }
            use_tile = true;
        }
        if (s->shards.n > 0 && !(tile[4].ok && tile[4].ahead))
            return fail(LFM_EUNSUPPORTED, "owner-sharded item tables run on the steady-state tile kernel only (parallel WARP, identity "
                                          "features, adagrad, no regularisation, d <= 64, max_sampled = 10)");
        if (use_tile) {
            a.n_items_magic = (uint32_t)((1ull << 32) / (uint64_t)s->itf.rows) + 1u;
            if (!s->recs_valid) {
                LFM_TRY(s->recs.alloc((size_t)s->n));
                HIP_TRY(launch_pack_records(a.user_ids, a.item_ids, a.Y, a.weight, s->n, s->recs.p, s->stream));
                s->recs_valid = true;
            }
            a.recs = s->recs.p;
        }
    }

    if (s->shards.n > 0 && !use_tile)
        return fail(LFM_EUNSUPPORTED, "owner-sharded item tables run on the steady-state tile kernel only (parallel WARP, identity "
                                      "features, adagrad, no regularisation, d <= 64, max_sampled = 10)");
    // Every other parallel-mode adagrad model (with or without L2 regularisation): the pipelined row-stream kernels
    // (feat_kernel.hpp) -- feature CSRs, BPR, k-OS, logistic (BASELINE configs C3 / C5).
    // (the narrow lane-group kernels, logistic_tile.hip: ltile_smem above)
    const bool use_ltile = !serial && !use_tile && s->itf.identity && s->usf.identity && !s->adadelta &&
                           item_alpha == 0.0 && user_alpha == 0.0 && a.update_mode == 0 && opts->feat_kernel == 0 && s->shards.n == 0 &&
                           s->n > 0 && ltile_smem != 0 && (loss == LFM_LOSS_LOGISTIC || (s->pos.indptr.p != nullptr && s->item_ids.p != nullptr));
    if (use_ltile) {
        if (!s->recs_valid) {
            LFM_TRY(s->recs.alloc((size_t)s->n));
            HIP_TRY(launch_pack_records(a.user_ids, a.item_ids, a.Y, a.weight, s->n, s->recs.p, s->stream));
            s->recs_valid = true;
        }
        a.recs = s->recs.p;
    }
    FeatPlan fplan;
    bool use_feat = false;
    // (adadelta: the ADA instantiations, d <= 128 and not the instrumented build; wider adadelta models run the generic kernels)
    if (!serial && !use_tile && !use_ltile && opts->feat_kernel != 1 && !(s->adadelta && (s->d > 128 || opts->feat_kernel == 2)) && s->itf.rows >= 1 && s->n > 0) {
        auto avg_len = [](const DevCsr &f) { return f.identity || f.rows <= 0 ? 1.0 : (double)f.nnz / (double)f.rows; };
        const int rows_hint = (int)(avg_len(s->usf) + 2.0 * avg_len(s->itf) + 0.999);
        const bool no_shared_rows = s->itf.identity && s->usf.identity && !s->adadelta;
        use_feat = feat_plan(loss, s->d, s->max_sampled, n_positives, opts->first_batch, rows_hint, &fplan, 0, false, no_shared_rows);
        if (use_feat && loss != LFM_LOSS_WARP_KOS) {
            if (!s->recs_valid) {
                LFM_TRY(s->recs.alloc((size_t)s->n));
                HIP_TRY(launch_pack_records(a.user_ids, a.item_ids, a.Y, a.weight, s->n, s->recs.p, s->stream));
                s->recs_valid = true;
            }
            a.recs = s->recs.p;
        }
    }

    // Shared item-feature rows in LDS slices (hot_slices.hip): the row-stream kernels of an adagrad model without
    // regularisation, atomic publication.  A launch is then at most `hot_chunk` positions long -- the hot rows a launch reads
    // are as old as the launch -- and ramps with the training history like the interactions in flight do (hot_k).
    // lfm_opts.debug bit 14 (16384) / LIGHTFM_AMD_HOT_SLICES=0 keep the rows on the float atomics.
    // (read per epoch, not once per process: the study scripts vary them between fits)
    const int hot_env = [] { const char *e = getenv("LIGHTFM_AMD_HOT_SLICES"); return e ? atoi(e) : 1; }();
    const int64_t hot_chunk_env = [] { const char *e = getenv("LIGHTFM_AMD_HOT_CHUNK"); const long v = e ? atol(e) : 0; return (int64_t)(v >= 64 ? v : 131072); }();
    const int64_t hot_k_env = [] { const char *e = getenv("LIGHTFM_AMD_HOT_K"); const long v = e ? atol(e) : 0; return (int64_t)(v >= 1 ? v : 128); }();
    const int hot_rep_env = [] { const char *e = getenv("LIGHTFM_AMD_HOT_REPLICAS"); return e ? atoi(e) : 0; }();
    const int64_t hot_floor_env = [] { const char *e = getenv("LIGHTFM_AMD_HOT_FLOOR"); const long v = e ? atol(e) : 0; return (int64_t)(v >= 1 ? v : 8); }();
    const int hot_overlap_env = [] { const char *e = getenv("LIGHTFM_AMD_HOT_OVERLAP"); return e ? atoi(e) : 0; }();
    const int64_t hot_min_env = [] { const char *e = getenv("LIGHTFM_AMD_HOT_MIN"); const long v = e ? atol(e) : 0; return (int64_t)(v >= 1 ? v : 2048); }();
    bool use_hot = false;
    if (use_feat && hot_env && !(opts->debug & 16384) && a.update_mode == 0 && item_alpha == 0.0 && user_alpha == 0.0 &&
        !s->adadelta && s->d <= 128 && opts->feat_kernel != 2 && s->shards.n == 0) {
        if (s->hot.state == 0) LFM_TRY(build_hot_set(s));
        use_hot = s->hot.state == 1;
    }
    // LIGHTFM_AMD_HOT_OVERLAP=1 (off by default): the slice kernel of launch k on the second stream UNDER launch k + 1, the
    // row-stream wavefronts on an LDS budget that leaves one slice workgroup per CU its room.  Built and measured: 77.3
    // against 80.8 M/s on C3 (the row-stream kernel is issue-bound, the slice kernel ALU-bound: side by side they take each
    // other's issue slots), precision@10 unchanged (profiles/r06_hot_overlap_ab.txt).  Only for full-length launches of the
    // default plan: while the record length still ramps, and under a caller's own launch plan (the sequential parity
    // tests), a launch's records are applied before the next launch starts.
    bool hot_overlap = use_hot && hot_overlap_env != 0 && opts->launches_per_epoch <= 0 && !fixed_cap;
    auto avg_len = [](const DevCsr &f) { return f.identity || f.rows <= 0 ? 1.0 : (double)f.nnz / (double)f.rows; };
    if (use_hot && !hot_overlap) {
        // without their shared rows BPR / logistic are not bound by the atomic unit any more: the residency of WARP / k-OS
        FeatPlan wide;
        const int rows_hint = (int)(avg_len(s->usf) + 2.0 * avg_len(s->itf) + 0.999);
        if (feat_plan(loss, s->d, s->max_sampled, n_positives, opts->first_batch, rows_hint, &wide, 0, true)) fplan = wide;
    }
    if (hot_overlap) {
        const size_t slice_bytes = (size_t)s->hot.n * s->hot.cs * 2 * sizeof(float);
        const size_t room = (size_t)156 * 1024 > slice_bytes ? (size_t)156 * 1024 - slice_bytes : 0;
        FeatPlan tight;
        const int rows_hint = (int)(avg_len(s->usf) + 2.0 * avg_len(s->itf) + 0.999);
        const size_t cap = (room / (size_t)std::max(1, fplan.waves_per_cu)) & ~(size_t)15;
        if (cap >= 4096 && feat_plan(loss, s->d, s->max_sampled, n_positives, opts->first_batch, rows_hint, &tight, cap) &&
            tight.waves_per_cu == fplan.waves_per_cu && tight.smem * (size_t)tight.waves_per_cu / (size_t)tight.waves_per_block + slice_bytes <= (size_t)158 * 1024)
            fplan = tight;
        else
            hot_overlap = false;
    }
    // (the steady-state bound on the interactions in flight of a model with shared rows -- shared_cap above -- stays what it
    // is: relaxing it because the hot rows left the atomic path cost the hybrid WARP / k-OS gates 0.002-0.003 precision@10
    // whatever the record length, profiles/r06_hot_gate_sweep.txt)

    if (opts->neg_log) { LFM_TRY(s->neg_log.alloc((size_t)s->n)); a.neg_log = s->neg_log.p; }
    if (opts->sampled_log) { LFM_TRY(s->sampled_log.alloc((size_t)s->n)); a.sampled_log = s->sampled_log.p; }
    if (a.neg_log) HIP_TRY(hipMemsetAsync(a.neg_log, 0xff, (size_t)s->n * 4, s->stream));
    if (a.sampled_log) HIP_TRY(hipMemsetAsync(a.sampled_log, 0, (size_t)s->n * 4, s->stream));
    HIP_TRY(hipMemsetAsync(s->counters.p, 0, 13 * sizeof(unsigned long long), s->stream));

    const bool recs_in_use = a.recs != nullptr;
    if (validate_enabled()) LFM_TRY(validate_inputs(s, slot, 0, recs_in_use));
    const bool reg = item_alpha != 0.0 || user_alpha != 0.0;
    if (reg && !serial) HIP_TRY(launch_reg_log_init(s->scales.p, s->reg_log.p, s->reg_live.p, s->stream));
    // bound of a position's growth of a log-scale: log1p(alpha * lr_max).  Adagrad rates never exceed the
    // learning rate (G >= 1); adadelta's sqrt(M + eps) / sqrt(G + eps) is not bounded by it: 1.0 there
    const double lr_max = s->adadelta ? 1.0 : (double)std::max(s->lr, 1e-6f);
    const double reg_step = log1p(std::max(item_alpha, user_alpha) * lr_max);
    // accuracy: a growth of <= 0.5 per launch; never below 64 Ki positions for that -- but never so long that
    // the scale could grow by more than e^55 (four virtual folds) either: cells an interaction touches are
    // multiplied by (1 + alpha lr) in place (PYX:433, 446), and only a boundary divides them again
    const int64_t reg_len_cap = !reg ? INT64_MAX : (int64_t)std::max(256.0, std::max(
        std::min(1e12, 0.5 / std::max(reg_step, 1e-300)), std::min(65536.0, 55.0 / std::max(reg_step, 1e-300))));
    int in_flight = 1, tile_ng_used = 0, n_launches = 0, plan_flags = 0;
    bool used_second_stream = false;
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    if (serial) {
        int T = needs_rng ? n_seeds : 1;
        for (int t = 0; t < T; ++t) {
            static_chunk(s->n, T, t, &a.begin, &a.end);
            a.seed_idx = t;
            if (a.end > a.begin) HIP_TRY(launch_fit(loss, a, 1, WAVE, smem, s->stream));
        }
    } else {
        // Launch plan: the epoch is cut into launches (a kernel boundary is a device-wide
        // release/acquire) of at most `slice` positions; while the ramp is still below the chip's
        // residency a launch covers 64 passes of its wavefronts, so the ramp costs milliseconds.
        int L = opts->launches_per_epoch;
        // 2 Mi positions per launch: a launch costs ~75 us beyond its passes (its wavefronts start in lockstep
        // and drain unevenly) -- C2 at 20 launches of 1 Mi per epoch ran 1.04 G interactions/s, at 10 launches
        // 1.14 G/s, at 5 launches 1.17 G/s (round 3; precision@10 of 3 seeds unchanged at 10).  What a boundary
        // still gives: fresh bias snapshots for the tile kernel's scoring and the regularisation folds.
        // (LIGHTFM_AMD_LAUNCH_LOG2: experiments with the launch length, default 21 = 2 Mi positions)
        static const int launch_log2 = [] { const char *e = getenv("LIGHTFM_AMD_LAUNCH_LOG2"); const int v = e ? atoi(e) : 21; return v >= 16 && v <= 28 ? v : 21; }();
        if (L <= 0) L = (int)std::max<int64_t>(1, std::min<int64_t>(64, (s->n + (1ll << launch_log2) - 1) >> launch_log2));
        const int64_t slice = std::max<int64_t>(1, (s->n + L - 1) / L);
        const size_t generic_smem = smem;
        const bool snap_biases = use_tile && s->tab[0][3].flags != 0 && !(opts->debug & 32);
        // ... a side's bias table is snapshot only while it fits an L2 with room to spare (2 MiB): the C4 shard's 20 MB of item
        // biases miss the L2s anyway: the per-launch copies and their L2 footprint buy nothing there (C4 shard +2-8 % without
        // snapshots, profiles/r05_visit_f.txt; the memory-side requests are 128-byte lines with or without them)
        bool snap_side[2] = {false, false};
        for (int side = 0; side < 2; ++side) snap_side[side] = snap_biases && tab_count(s, side, 3) * sizeof(float) <= ((size_t)2 << 20);
        for (int side = 0; side < 2; ++side)
            if (snap_side[side])
                for (int par = 0; par < 2; ++par) LFM_TRY(s->bias_snap[side][par].alloc(tab_count(s, side, 3)));
        // Consecutive full-residency launches go to two streams alternately: the wavefronts of a launch finish
        // unevenly (same number of passes each, passes of different length: a tail of ~100 us in which the chip
        // drains), and the next launch's workgroups, already queued on the other stream, take the slots as they
        // free up.  Residency -- hence the interactions in flight -- is what it was.  Not with lazy regularisation
        // (its boundary kernels order the launches) and not below residency (the ramp).  debug bit 7 (128) disables it.
        // (nor when the caller fixed the launch plan or asked for plain stores: the bit-exactness tests rely on
        // one launch seeing the previous one's stores)
        const bool two_streams = !reg && !(opts->debug & 128) && opts->launches_per_epoch <= 0 && a.update_mode == 0;
        bool forked = false;
        int n_full = 0;
        FitArgs base = a;
        // [pos_begin, pos_end): one segment of the epoch (multi-GPU driver), default the whole epoch
        const int64_t seg_begin = std::max<int64_t>(0, std::min<int64_t>(opts->pos_begin, s->n));
        const int64_t seg_end = opts->pos_end > 0 ? std::max(seg_begin, std::min<int64_t>(opts->pos_end, s->n)) : s->n;
        int64_t begin = seg_begin;
        int ng_used = 0;
        while (begin < seg_end) {
            const int64_t allowed = allowed_in_flight(begin - seg_begin);
            // pick the kernel variant for this launch
            int ng = 0;
            if (use_tile && s->shards.n > 0) ng = 4;  // the only kernel that addresses sharded item tables
            else if (use_tile) {
                for (int c : {4, 2, 1})
                    if (!ng && tile[c].ok && (allowed / c >= (int64_t)s->cus * 8 || c == 1)) ng = c;
                if (!ng)
                    for (int c : {1, 2, 4})
                        if (!ng && tile[c].ok) ng = c;
            }
            a = base;
            size_t lsmem = generic_smem;
            int per_wave = 1, wpb = WAVES_PER_BLOCK;
            if (use_feat) {
                lsmem = fplan.smem;
                wpb = fplan.waves_per_block;
                a.tile_rows = fplan.rr;
                a.tile_stride = fplan.ts;
                a.stage_rows = fplan.sr;
                a.cand_base = fplan.cand_base;
                a.pair_cap = fplan.pair_cap;
                a.first_batch = fplan.first_batch;
            }
            if (ng) {
                const TilePlan &t = tile[ng];
                lsmem = t.smem;
                per_wave = t.narrow ? 2 * ng : ng;
                a.tile_rows = t.rows;
                a.tile_stride = t.stride;
                a.first_batch = t.first_batch;
            }
            if (use_ltile) {
                lsmem = ltile_smem;
                per_wave = loss == LFM_LOSS_LOGISTIC ? 8 : bpr_tile_per_wave();
            }
            // row-stream kernels: 8 wavefronts per CU publish fastest (C3: 43 M/s at 2 048 interactions
            // in flight against 35 M/s at 3 072 -- the float atomics queue up in the fabric)
            // (LIGHTFM_AMD_FEAT_WAVES_PER_CU: experiments with the row-stream kernels' residency; their LDS budget per
            // wavefront -- LIGHTFM_AMD_FEAT_LDS_KB, feat_kernels.hip -- must allow it)
            static const int feat_waves_env = [] { const char *e = getenv("LIGHTFM_AMD_FEAT_WAVES_PER_CU"); return e ? atoi(e) : 0; }();
            const int feat_waves = feat_waves_env > 0 ? feat_waves_env : fplan.waves_per_cu;
            size_t cu_blocks = use_feat ? (size_t)std::max(1, feat_waves / wpb) : 8;
            // The tile kernel's LOGISTIC instantiation: two workgroups per CU (8 192 interactions in flight at four per pass) on ONE
            // stream.  Logistic steps on every interaction with a loss that does not shrink as the model converges (noisy labels), so the
            // popular rows take many stale full-size steps at once: at the FULL ML-20M shape (38 M labelled interactions, d = 64)
            // precision@10 against the reference's 0.08008 is 0.0793 / 0.0786 / 0.0777 at 4 096 / 8 192 / 12 288+ in flight (547 / 946 /
            // 1 300 M interactions/s) -- the third workgroup per CU (and the second stream, which doubles what is in flight whenever a
            // grid is below the hardware's residency) buys rate outside the +-0.002 gate (profiles/r06_quality20m_identity_logistic.txt).
            // BPR and WARP, whose losses vanish where the ranking is right, hold at full residency (+0.0002, -0.0002).
            // (LIGHTFM_AMD_LGT_BLOCKS = 1..3: experiments)
            if (ng && bpr_tile && lgt_tile) {
                const int lgt_blocks = [] { const char *e = getenv("LIGHTFM_AMD_LGT_BLOCKS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 3 ? v : 2; }();
                cu_blocks = std::min<size_t>(cu_blocks, (size_t)lgt_blocks);
            }
            const int blocks_per_cu = (int)std::max<size_t>(1, std::min<size_t>(cu_blocks, (160 * 1024) / std::max<size_t>(lsmem, 1)));
            int max_grid = s->cus * blocks_per_cu;
            bool below_residency = allowed / (wpb * per_wave) < max_grid;
            if (((ng && tile[ng].narrow) || use_ltile) && below_residency && !fixed_cap) {
                // the narrow-model kernel keeps 128 interactions per CU in flight: on a catalogue of a few ten thousand items it
                // is the STEADY-STATE bound (never more in flight than the smaller side has rows) that binds, not the history
                // ramp -- such launches are full-length and alternate between the two streams like any launch at residency
                const int64_t by_history = opts->ramp_k < 0 ? INT64_MAX / 4 : std::max<int64_t>(8, (history0 + (begin - seg_begin)) / ramp_k);
                if (by_history / (wpb * per_wave) >= max_grid) below_residency = false;
            }
            max_grid = (int)std::max<int64_t>(1, std::min<int64_t>(max_grid, allowed / (wpb * per_wave)));
            if (max_grid > s->cus) max_grid -= max_grid % s->cus;  // whole workgroups per CU
            const int64_t flight = (int64_t)max_grid * wpb * per_wave;
            int64_t len = std::min<int64_t>(slice, seg_end - begin);
            if (!fixed_cap && below_residency) len = std::min<int64_t>(len, std::max<int64_t>(flight * 64, 1024));
            // Lazy regularisation: the launch's readers extrapolate the scale's growth from the rate of the
            // previous launch (device.hpp: RegScale); a launch is kept to a growth of at most ~0.5 in log
            // scale (at the bound log1p(alpha * lr) per position) so that the extrapolation error stays
            // around a per cent -- but not below 64 Ki positions for accuracy's sake: past that alpha the model is
            // flattened whatever the scale's third digit is ("excessive regularisation"; see reg_len_cap).
            if (reg) len = std::min<int64_t>(len, reg_len_cap);
            bool hot_launch = use_hot;
            if (use_hot) {
                const int64_t hist = history0 + (begin - seg_begin);
                // (floor: what the in-flight ramp itself starts from -- at history 0 every interaction is a maximum-loss step, and
                // a launch of 256 positions against frozen hot rows cost the hybrid WARP gate 0.0035 whatever hot_k was)
                const int64_t hot_len = opts->ramp_k < 0 ? hot_chunk_env : std::min<int64_t>(hot_chunk_env, std::max<int64_t>(hot_floor_env, hist / hot_k_env));
                // While history / hot_k is still below hot_min positions (the first 1-2 % of a first epoch) the launches run the
                // shared rows on the float atomics as ever, at the lengths the in-flight ramp gives them: a pair of launches per 8 ..
                // 2 000 positions cost the first epoch of C3 ~1 400 launch pairs and twice a later epoch's time.  (A caller's own
                // launch plan and a disabled ramp -- the parity tests -- take the hot set from the first position.)
                if (opts->launches_per_epoch <= 0 && opts->ramp_k >= 0 && !fixed_cap && hot_len < hot_min_env) {
                    hot_launch = false;
                    len = std::min<int64_t>(len, std::max<int64_t>(1, hot_min_env * hot_k_env - hist));  // (switch on time)
                } else {
                    len = std::min<int64_t>(len, hot_len);
                }
            }
            // ... and the FIRST regularised launch of a session has no measured rate to extrapolate with (reg_live is
            // created zeroed; LightFM.fit_partial opens a session per call): its readers see the launch-start scale
            // throughout, so it is kept to a growth of <= 0.02 -- the launch after it extrapolates from its rate
            if (reg && !s->reg_rate_known)
                len = std::min<int64_t>(len, (int64_t)std::max(256.0, std::min(1e12, 0.02 / std::max(reg_step, 1e-300))));
            a.begin = begin;
            a.end = begin + len;
            const int64_t waves = (len + per_wave - 1) / per_wave;
            const int grid = (int)std::min<int64_t>(max_grid, (waves + wpb - 1) / wpb);
            if (trace_enabled())
                fprintf(stderr, "LFM_LAUNCH s=%p loss %d [%lld, %lld) grid %d ng %d feat %d shuffle %p ids %p %p Y %p w %p pos %p %p W %p %p\n",
                        (void *)s, loss, (long long)a.begin, (long long)a.end, grid, ng, (int)use_feat, (const void *)a.shuffle,
                        (const void *)a.user_ids, (const void *)a.item_ids, (const void *)a.Y, (const void *)a.weight,
                        (const void *)a.pos.indptr, (const void *)a.pos.indices, (void *)a.m.W[0], (void *)a.m.W[1]);
            int grid_used = grid;
            hipStream_t lst = s->stream;
            int par = 0;
            // (tile kernel only: its grid IS the hardware's residency.  The row-stream and generic kernels are launched
            // with fewer workgroups than would fit -- 8 wavefronts per CU publish fastest -- and two of their launches
            // side by side would double the interactions in flight: C3 fell from 42.5 to 37.6 M interactions/s)
            if (two_streams && (ng || use_ltile) && !(bpr_tile && lgt_tile) && !below_residency && !fixed_cap) {
                par = n_full++ & 1;
                if (par) {
                    if (!s->stream2) {
                        HIP_TRY(hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking));
                        HIP_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
                        HIP_TRY(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
                    }
                    if (!forked) {  // everything queued so far (shuffle, record packing, the ramp) precedes the second stream
                        HIP_TRY(hipEventRecord(s->ev_fork, s->stream));
                        HIP_TRY(hipStreamWaitEvent(s->stream2, s->ev_fork, 0));
                        forked = true;
                    }
                    lst = s->stream2;
                    used_second_stream = true;
                }
            }
            if (ng) {
                // Scoring reads twelve 4-byte biases per interaction.  With uncached tables each
                // is a fabric request; a cached snapshot taken at the launch boundary (the tables
                // are tiny) serves them from L1/L2 instead, and no atomic ever drops its lines.
                // Updates still read and publish the live tables.  (LFM debug bit 5 disables it.)
                // the steady-state kernels keep the live bias cells as (b, bG) pairs (lfm_session::bias_pairs); LIGHTFM_AMD_BIAS_PAIRS=0:
                // the separate tables
                static const bool pairs_env = [] { const char *e = getenv("LIGHTFM_AMD_BIAS_PAIRS"); return !e || atoi(e) != 0; }();
                // ... and the narrow-model kernel W and G as rows of one line (LIGHTFM_AMD_ROW_PAIRS=0: the separate tables), for d <= 12
                // with the bias cells in that line too (LIGHTFM_AMD_ROW_PAIRS=1: without them)
                // (read per epoch call: the tests switch layouts inside one process)
                const int rows_env = [] { const char *e = getenv("LIGHTFM_AMD_ROW_PAIRS"); return e ? atoi(e) : 2; }();
                const bool want_rows = rows_env != 0 && tile[ng].narrow && (int64_t)std::max(s->n_feat[0], s->n_feat[1]) * 32 < (1ll << 30);
                const bool want_rows_bias = want_rows && rows_env >= 2 && s->d <= 12;
                // (... and the tile kernel's BPR / logistic instantiations: an update on every interaction)
                // (not with the lazy regularisation: its folds at the launch boundaries work on the bias tables)
                const bool want_pairs = pairs_env && (tile[ng].ahead || (bpr_tile && a.update_mode == 0 && !reg)) && s->shards.n == 0 && !want_rows_bias;
                if (want_pairs && !s->pairs_live) LFM_TRY(bias_pairs_pack(s, s->stream));  // (before the streams fork: both see it)
                if (!want_pairs && s->pairs_live) {  // (a launch of another kernel after steady-state launches: does not happen in the shipped plan)
                    if (s->stream2) HIP_TRY(hipStreamSynchronize(s->stream2));
                    LFM_TRY(bias_pairs_unpack(s, s->stream));
                    HIP_TRY(hipStreamSynchronize(s->stream));
                }
                if (want_rows && !s->rows_live) LFM_TRY(row_pairs_pack(s, s->stream, want_rows_bias));  // (after the bias pairs went back)
                if (!want_rows && s->rows_live) {
                    if (s->stream2) HIP_TRY(hipStreamSynchronize(s->stream2));
                    LFM_TRY(row_pairs_unpack(s, s->stream));
                    HIP_TRY(hipStreamSynchronize(s->stream));
                }
                a.rp[0] = a.rp[1] = nullptr;
                a.rp_bias = 0;
                if (s->rows_live) {
                    a.rp[0] = s->row_pairs[0].p;
                    a.rp[1] = s->row_pairs[1].p;
                    a.rp_bias = s->rows_bias ? 1 : 0;
                    if (s->rows_bias) plan_flags |= 128;
                }
                a.b_read[0] = a.m.b[0];
                a.b_read[1] = a.m.b[1];
                a.b_read_stride[0] = a.b_read_stride[1] = 1;
                a.bb[0] = a.bb[1] = nullptr;
                if (s->pairs_live) {
                    for (int side = 0; side < 2; ++side) {
                        a.bb[side] = s->bias_pairs[side].p;
                        a.b_read[side] = s->bias_pairs[side].p;
                        a.b_read_stride[side] = 2;
                    }
                }
                {
                    for (int side = 0; side < 2; ++side) {
                        if (!snap_side[side] || s->rows_bias) continue;  // (biases in the row pairs: scoring reads them with the rows)
                        const int64_t cnt = (int64_t)tab_count(s, side, 3);
                        if (cnt) {
                            const int cgrid = (int)std::min<int64_t>(1024, (cnt + 255) / 256);
                            if (s->pairs_live) copy_strided_kernel<<<cgrid, 256, 0, lst>>>(s->bias_snap[side][par].p, s->bias_pairs[side].p, cnt, 2);
                            else copy_kernel<<<cgrid, 256, 0, lst>>>(s->bias_snap[side][par].p, s->tab[side][3].p, cnt);
                        }
                        a.b_read[side] = s->bias_snap[side][par].p;
                        a.b_read_stride[side] = 1;
                        plan_flags |= 1 << side;
                    }
                }
                if (tile[ng].narrow) {
                    static const int narrow_blocks_env = [] { const char *e = getenv("LIGHTFM_AMD_NARROW_BLOCKS"); return e ? atoi(e) : 0; }();
                    HIP_TRY(launch_fit_warp_tile_narrow(a, grid, lst, s->cus, narrow_blocks_env, &grid_used));
                    plan_flags |= 64;
                }
                else if (tile[ng].ahead) HIP_TRY(launch_fit_warp_tile_ahead(a, grid, lst, s->cus, &grid_used));
                else if (bpr_tile) {
                    HIP_TRY(launch_fit_bpr_wide_tile(a, ng, tile[ng].vec, grid, lsmem, lst, s->cus, &grid_used, tile[ng].dma4, lgt_tile));
                    plan_flags |= lgt_tile ? 2048 : 1024;
                }
                else HIP_TRY(launch_fit_warp_tile(a, ng, tile[ng].vec, grid, lsmem, lst, s->cus, opts->warp_kernel == 2,
                                                  &grid_used, tile[ng].dma4));
            }
            else if (use_ltile) {
                if (s->pairs_live) {  // (does not happen: no other kernel of a logistic epoch packs them)
                    if (s->stream2) HIP_TRY(hipStreamSynchronize(s->stream2));
                    LFM_TRY(bias_pairs_unpack(s, s->stream));
                }
                if (!s->rows_live) LFM_TRY(row_pairs_pack(s, s->stream, true));  // (before the streams fork: both see it)
                a.rp[0] = s->row_pairs[0].p;
                a.rp[1] = s->row_pairs[1].p;
                a.rp_bias = 1;
                if (loss == LFM_LOSS_LOGISTIC) HIP_TRY(launch_fit_logistic_tile(a, grid, lst, s->cus, &grid_used));
                else HIP_TRY(launch_fit_bpr_tile(a, grid, lst, s->cus, &grid_used));
                plan_flags |= 128 | (loss == LFM_LOSS_LOGISTIC ? 256 : 512);
            }
            else if (use_feat && hot_launch) {
                lfm_session::HotSet &h = s->hot;
                const int par = n_launches & 1;
                // this launch's records may run under the next launch (hot_overlap) once the record length has ramped up
                const bool under_next = hot_overlap && len >= 8192;
                if (under_next && !s->stream2) {
                    HIP_TRY(hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
                }
                for (int i = 0; i < 2; ++i) {
                    if (!h.ev_p1[i]) HIP_TRY(hipEventCreateWithFlags(&h.ev_p1[i], hipEventDisableTiming));
                    if (!h.ev_done[i]) HIP_TRY(hipEventCreateWithFlags(&h.ev_done[i], hipEventDisableTiming));
                }
                // the record set of this parity is free once the slice kernel that read it last has finished; a launch whose
                // records are applied in line waits for EVERYTHING still running on the second stream
                for (int i = 0; i < 2; ++i) {
                    if (h.pending[i] && (i == par || !under_next)) {
                        HIP_TRY(hipStreamWaitEvent(lst, h.ev_done[i], 0));
                        h.pending[i] = false;
                    }
                }
                LFM_TRY(h.rec[par].reserve((size_t)len));
                LFM_TRY(h.x[par].reserve((size_t)len * (size_t)s->d));
                HIP_TRY(hipMemsetAsync(h.rec[par].p, 0xff, (size_t)len * sizeof(HotRec), lst));  // n_total = -1: nothing to apply
                a.hot_slot = h.slot.p;
                a.hot_rec = h.rec[par].p;
                a.hot_x = h.x[par].p;
                HIP_TRY(launch_fit_feat_hot(loss, a, grid, wpb * WAVE, lsmem, lst, s->cus, &grid_used));
                HotArgs ha;
                ha.rec = h.rec[par].p;
                ha.x = h.x[par].p;
                ha.n_rec = len;
                ha.d = s->d;
                ha.hot_n = h.n;
                // replicas of a slice: the chip's workgroup slots over all slices -- two workgroups of 8 wavefronts per CU
                // when the slice kernel has the chip to itself, one of 16 when it runs under the next launch; a short launch
                // takes fewer (one record per launch: ONE replica, the sequential result)
                const int n_slices = s->d / h.cs + 1;
                const int per_cu = under_next ? 1 : 2;
                int n_rep = hot_rep_env > 0 ? hot_rep_env : std::max(1, (per_cu * s->cus) / n_slices);
                n_rep = (int)std::max<int64_t>(1, std::min<int64_t>(n_rep, len / 64));
                ha.n_rep = n_rep;
                ha.rows = h.rows.p;
                ha.W = s->tab[0][0].p;
                ha.G = s->tab[0][1].p;
                ha.b = s->tab[0][3].p;
                ha.bG = s->tab[0][4].p;
                ha.snapW = h.snapW.p;
                ha.snapG = h.snapG.p;
                ha.snapb = h.snapb.p;
                ha.snapbG = h.snapbG.p;
                ha.lr = s->lr;
                ha.rho = s->rho;
                ha.eps = s->eps;
                static const int hot_threads_env = [] { const char *e = getenv("LIGHTFM_AMD_HOT_THREADS"); const int v = e ? atoi(e) : 0; return (v == 256 || v == 512 || v == 1024) ? v : 0; }();
                // (1 024 threads: 32 wavefronts per CU hide the record loads -- C3 92.0 -> 99.9 M/s against 512, 76 at 256;
                // profiles/r06_c3_tuning.txt)
                const int threads = n_rep > 1 ? (hot_threads_env ? hot_threads_env : 1024) : 64;
                if (under_next) {
                    HIP_TRY(hipEventRecord(h.ev_p1[par], lst));
                    HIP_TRY(hipStreamWaitEvent(s->stream2, h.ev_p1[par], 0));
                    HIP_TRY(launch_hot_slices(ha, h.cs, threads, s->stream2));
                    HIP_TRY(hipEventRecord(h.ev_done[par], s->stream2));
                    h.pending[par] = true;
                    used_second_stream = true;
                } else {
                    HIP_TRY(launch_hot_slices(ha, h.cs, threads, lst));
                }
                plan_flags |= 32;
            }
            else if (use_feat && s->adadelta) HIP_TRY(launch_fit_feat_ada(loss, a, grid, wpb * WAVE, lsmem, lst, s->cus, &grid_used));
            else if (use_feat) HIP_TRY(launch_fit_feat(loss, a, grid, wpb * WAVE, lsmem, lst, s->cus, &grid_used,
                                                       opts->feat_kernel == 2));
            else HIP_TRY(launch_fit(loss, a, grid, 256, lsmem, lst, s->cus, &grid_used));
            // Lazy L2 regularisation (device.hpp: RegScale): the scales live in s->reg_log while the
            // launch runs; between launches they are folded into the weights when one has passed
            // MAX_REG_SCALE (locked_regularize, PYX:678-691) -- decided on the device, no host round trip.
            if (reg) {
                HIP_TRY(launch_regularize(a.m, s->reg_log.p, s->reg_live.p, 0, s->stream, len));
                s->reg_rate_known = true;
            }
            begin += len;
            // of the last (largest) launch, after the launcher's residency clamp
            in_flight = (int)std::min<int64_t>((int64_t)grid_used * wpb * per_wave, INT32_MAX);
            ng_used = ng;
            ++n_launches;
        }
        tile_ng_used = ng_used;
        for (int i = 0; i < 2; ++i) {  // slice kernels still running under the last launch (hot_overlap)
            if (s->hot.pending[i]) {
                HIP_TRY(hipStreamWaitEvent(s->stream, s->hot.ev_done[i], 0));
                s->hot.pending[i] = false;
            }
        }
        if (forked) {  // the second stream joins before anything else of this session runs
            HIP_TRY(hipEventRecord(s->ev_join, s->stream2));
            HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_join, 0));
        }
        if (s->pairs_live) LFM_TRY(bias_pairs_unpack(s, s->stream));  // the bias cells go back to their tables
        if (s->rows_live) LFM_TRY(row_pairs_unpack(s, s->stream));     // ... and the narrow-model kernel's rows
    }
    if (reg) HIP_TRY(launch_regularize(a.m, serial ? nullptr : s->reg_log.p, serial ? nullptr : s->reg_live.p, 1, s->stream));  // PYX:910-912
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    opts->kernel_ms = ms;
    unsigned long long c[13];
    LFM_TRY(s->counters.download(c));
    for (int i = 0; i < 4; ++i) opts->counters[i] = (int64_t)c[i];
    for (int i = 0; i < 8; ++i) opts->phase_cycles[i] = (int64_t)c[4 + i];
    opts->tile_ng = tile_ng_used;
    opts->kernel_used = (tile_ng_used || use_ltile) ? 1 : (use_feat ? 2 : 0);
    opts->in_flight = in_flight;
    opts->launches = n_launches;
    opts->streams_used = used_second_stream ? 2 : 1;
    opts->tile_ahead = (tile_ng_used == 4 && tile[4].ahead) ? 1 : 0;
    opts->user_store = (base_user_store && (tile_ng_used || use_feat)) ? 1 : 0;
    for (int side = 0; side < 2; ++side)
        if (s->tab[side][0].flags != 0) plan_flags |= 4 << side;
    if ((uint64_t)s->itf.rows * (uint64_t)s->d >= (1ull << 30)) plan_flags |= 16;
    opts->plan_flags = plan_flags;
    if (opts->neg_log) LFM_TRY(s->neg_log.download(opts->neg_log));
    if (opts->sampled_log) LFM_TRY(s->sampled_log.download(opts->sampled_log));
    if (validate_enabled()) LFM_TRY(validate_inputs(s, slot, 1, recs_in_use));
    ++s->epochs_run;
    if (c[FAULT_SLOT])  // device.hpp: guard_row
        return fail(LFM_ECORRUPT, "a shuffle entry outside [0, n) was read on the device: the shuffle slot is not a permutation "
                                  "(lfm_session_upload_shuffle input) or device memory is corrupted; the epoch's updates are unreliable");
    return LFM_OK;
}

extern "C" int lfm_session_check_finite(lfm_session *s)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemsetAsync(s->flag.p, 0, sizeof(int), s->stream));
    for (int side = 0; side < 2; ++side) {
        int64_t first = 0, rows = s->n_feat[side];
        if (side == 0 && s->shards.n > 0 && s->share_rank >= 0) {
            // owner-sharded item tables: this session answers for the rows it owns (the others' copies here are stale;
            // their owners check them, and the host driver combines the answers)
            first = std::min<int64_t>((int64_t)s->share_rank * s->shards.rows_per_shard, rows);
            rows = std::min<int64_t>(s->shards.rows_per_shard, rows - first);
        }
        HIP_TRY(launch_nonfinite(s->tab[side][0].p + first * s->d, rows * s->d, s->flag.p, s->stream));
        HIP_TRY(launch_nonfinite(s->tab[side][3].p + first, rows, s->flag.p, s->stream));
    }
    int f = 0;
    HIP_TRY(hipMemcpyAsync(&f, s->flag.p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return f ? 0 : 1;
}

extern "C" int lfm_session_sync_to_host(lfm_session *s, lfm_model *model)
{
    if (!s || !model) return fail(LFM_EINVAL, "null argument");
    if (model->n_item_feat != s->n_feat[0] || model->n_user_feat != s->n_feat[1] || model->d != s->d_host)
        return fail(LFM_EINVAL, "model shape differs from the session's");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (int side = 0; side < 2; ++side)
        for (int k = 0; k < 6; ++k)
            if (kind_used(s, k)) LFM_TRY(download_table(s, side, k, host_tab(model, side, k)));
    double sc[2];
    LFM_TRY(s->scales.download(sc));
    model->item_scale = sc[0];
    model->user_scale = sc[1];
    return LFM_OK;
}

extern "C" int lfm_session_load_model(lfm_session *s, const lfm_model *model)
{
    if (!s || !model) return fail(LFM_EINVAL, "null argument");
    if (model->n_item_feat != s->n_feat[0] || model->n_user_feat != s->n_feat[1] || model->d != s->d_host)
        return fail(LFM_EINVAL, "model shape differs from the session's");
    LFM_TRY(validate_model(model, s->scoring_only));
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (int side = 0; side < 2; ++side)
        for (int k = 0; k < 6; ++k)
            if (kind_used(s, k)) LFM_TRY(upload_table(s, side, k, host_tab(model, side, k)));
    double sc[2] = {model->item_scale, model->user_scale};
    LFM_TRY(s->scales.upload(sc, 2));
    return LFM_OK;
}

extern "C" int lfm_session_build_positives(lfm_session *s, int32_t n_users, int32_t n_items)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    if (n_users < 0 || n_items <= 0) return fail(LFM_EINVAL, "bad interaction matrix shape");
    if (s->n && (!s->user_ids.p || !s->item_ids.p)) return fail(LFM_EINVAL, "user_ids / item_ids not uploaded");
    if ((double)n_users * (double)n_items >= 1.8e19) return fail(LFM_EUNSUPPORTED, "shape beyond 64-bit keys");
    HIP_TRY(hipSetDevice(s->device));
    s->pos.clear();
    s->guard_sums.clear();
    LFM_TRY(s->pos.indptr.alloc((size_t)n_users + 1));
    DBuf<int32_t> idx;
    DrainOnExit drain(s->stream);
    LFM_TRY(idx.alloc((size_t)std::max<int64_t>(s->n, 1)));
    int64_t nnz = 0;
    hipError_t e = build_positives_csr(s->user_ids.p, s->item_ids.p, s->n, n_users, n_items, idx.p,
                                       s->pos.indptr.p, &nnz, s->stream);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? LFM_ENOMEM : LFM_ENODEV, std::string("build_positives_csr: ") + hipGetErrorString(e));
    // keep the (slightly larger) buffer: duplicates are rare, a trimmed copy would cost more than it frees
    std::swap(s->pos.indices.p, idx.p);
    std::swap(s->pos.indices.n, idx.n);
    s->pos.rows = n_users;
    s->pos.cols = n_items;
    s->pos.nnz = nnz;
    s->pos.identity = false;
    LFM_TRY(build_bloom(s));
    return LFM_OK;
}

extern "C" int lfm_session_download_positives(lfm_session *s, int32_t *indptr, int32_t *indices, int64_t *nnz)
{
    if (!s || !nnz) return fail(LFM_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    *nnz = s->pos.nnz;
    if (indptr && s->pos.indptr.p)
        HIP_TRY(hipMemcpy(indptr, s->pos.indptr.p, ((size_t)s->pos.rows + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (indices && s->pos.indices.p && s->pos.nnz)
        HIP_TRY(hipMemcpy(indices, s->pos.indices.p, (size_t)s->pos.nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    return LFM_OK;
}

extern "C" int lfm_session_representations(lfm_session *s, int32_t side, const lfm_csr *features, float *biases,
                                           float *embeddings)
{
    if (!s || !biases || !embeddings) return fail(LFM_EINVAL, "null argument");
    if (side != 0 && side != 1) return fail(LFM_EINVAL, "side must be 0 (item) or 1 (user)");
    LFM_TRY(validate_csr(features, "features"));
    if (features->cols > s->n_feat[side]) return fail(LFM_EINVAL, "feature matrix has more columns than there are embeddings");
    if (features->rows == 0) return LFM_OK;
    HIP_TRY(hipSetDevice(s->device));
    DevCsr f;
    DBuf<float> demb, dbias;
    DrainOnExit drain(s->stream);
    LFM_TRY(f.upload(features, true, true));
    if (!f.identity) LFM_TRY(check_id_range(s, f.indices.p, f.nnz, s->n_feat[side], "features.indices"));
    LFM_TRY(demb.alloc((size_t)features->rows * s->d));
    LFM_TRY(dbias.alloc((size_t)features->rows));
    HIP_TRY(launch_rep_rows(f.view(), s->tab[side][0].p, s->tab[side][3].p, s->d, s->d, demb.p, s->stream, 0, dbias.p));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->d == s->d_host) LFM_TRY(demb.download(embeddings));
    else if (features->rows)  // the caller's rows are no_components wide
        HIP_TRY(hipMemcpy2D(embeddings, (size_t)s->d_host * sizeof(float), demb.p, (size_t)s->d * sizeof(float),
                            (size_t)s->d_host * sizeof(float), (size_t)features->rows, hipMemcpyDeviceToHost));
    return dbias.download(biases);
}

// ---------------------------------------------------------------- predict ---

extern "C" int lfm_session_predict(lfm_session *s, const int32_t *user_ids, const int32_t *item_ids,
                                   float *predictions, int64_t n)
{
    if (!s) return fail(LFM_EINVAL, "null session");
    if (n < 0) return fail(LFM_EINVAL, "negative count");
    if (n == 0) return LFM_OK;
    if (!user_ids || !item_ids || !predictions) return fail(LFM_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(s->device));
    DBuf<int32_t> du, di;
    DBuf<float> dout;
    DrainOnExit drain(s->stream);
    LFM_TRY(du.upload(user_ids, (size_t)n));
    LFM_TRY(di.upload(item_ids, (size_t)n));
    LFM_TRY(dout.alloc((size_t)n));
    PredictArgs a;
    memset(&a, 0, sizeof(a));
    a.itf = s->itf.view();
    a.usf = s->usf.view();
    a.m = s->dmodel();
    a.uids = du.p;
    a.iids = di.p;
    a.out = dout.p;
    a.n = n;
    tile_geometry(s->d, 16, &a.tile_rows, &a.tile_stride);
    a.tile_rows &= ~1;
    if (a.tile_rows < 2) a.tile_rows = 2;
    size_t smem = sizeof(float) * (size_t)WAVES_PER_BLOCK * a.tile_rows * a.tile_stride;
    int64_t waves = (n + a.tile_rows / 2 - 1) / (a.tile_rows / 2);
    int grid = (int)std::min<int64_t>((int64_t)s->cus * 8, (waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    HIP_TRY(launch_predict(a, std::max(grid, 1), smem, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return dout.download(predictions);
}

// flag := 1 if a row of the CSR has a column index below its predecessor
// Are all rows ascending?  *flag += (entries smaller than their predecessor) - (first entries of rows smaller than the
// entry before them): zero iff every descent of the index array sits on a row boundary.  Two coalesced passes (the
// row-per-thread loop over ML-20M's 18 M train entries cost 0.32 ms per predict_ranks call); no atomics on sorted input.
__global__ void descents_kernel(const int32_t *indices, int64_t nnz, int *flag)
{
    // four entries per 16-byte load (the pool's blocks are 256-byte aligned), the entry before them one more request
    const int64_t st = (int64_t)gridDim.x * blockDim.x, quads = nnz / 4;
    const int4 *v4 = (const int4 *)indices;
    int n = 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += st) {
        const int4 v = v4[q];
        const int prev = q > 0 ? indices[4 * q - 1] : v.x;
        n += (v.x < prev ? 1 : 0) + (v.y < v.x ? 1 : 0) + (v.z < v.y ? 1 : 0) + (v.w < v.z ? 1 : 0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t j = std::max<int64_t>(4 * quads, 1); j < nnz; ++j) n += indices[j] < indices[j - 1] ? 1 : 0;
    if (n) atomicAdd(flag, n);
}
__global__ void boundary_descents_kernel(const int32_t *indptr, const int32_t *indices, int32_t rows, int *flag)
{
    const int64_t st = (int64_t)gridDim.x * blockDim.x;
    int n = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += st) {
        const int32_t p = indptr[r];
        if (p > 0 && indptr[r + 1] > p) n += indices[p] < indices[p - 1] ? 1 : 0;
    }
    if (n) atomicSub(flag, n);
}

extern "C" int lfm_session_predict_ranks(lfm_session *s, const lfm_csr *test, const lfm_csr *train, float *ranks)
{
    if (!s || !ranks) return fail(LFM_EINVAL, "null argument");
    LFM_TRY(validate_csr(test, "test_interactions"));
    LFM_TRY(validate_csr(train, "train_interactions"));
    if (test->rows > s->usf.rows || test->cols > s->itf.rows)
        return fail(LFM_EINVAL, "interaction matrix larger than the feature matrices");
    if (train->rows < test->rows) return fail(LFM_EINVAL, "train matrix has fewer rows than test");
    if (test->nnz == 0) return LFM_OK;
    HIP_TRY(hipSetDevice(s->device));
    DevCsr dtest;
    DBuf<float> urep, irep, irows, dranks, ieps, tscores, ibf;
    DBuf<int32_t> ulist, work;
    DrainOnExit drain(s->stream);
    // LIGHTFM_AMD_TIMING=1: host wall time of the call's phases on stderr
    static const bool timing = [] { const char *e = getenv("LIGHTFM_AMD_TIMING"); return e && atoi(e) != 0; }();
    const auto tp0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count(); };
    LFM_TRY(dtest.upload(test, false, false));
    {
        uint64_t sig[2] = {0, 0};
        bool same = false;
        if (train->nnz >= (1 << 16)) {  // (a small matrix is uploaded faster than it is summed)
            LFM_TRY(lfm_host_checksum_u32((const uint32_t *)train->indices, train->nnz, &sig[0]));
            LFM_TRY(lfm_host_checksum_u32((const uint32_t *)train->indptr, (int64_t)train->rows + 1, &sig[1]));
            same = s->train_keep_valid && s->train_keep.rows == train->rows && s->train_keep.cols == train->cols &&
                   s->train_keep.nnz == train->nnz && s->train_sig[0] == sig[0] && s->train_sig[1] == sig[1];
        }
        if (!same) {
            s->train_keep_valid = false;
            LFM_TRY(s->train_keep.upload(train, false, false));
            s->train_sig[0] = sig[0];
            s->train_sig[1] = sig[1];
            s->train_keep_valid = train->nnz >= (1 << 16);
        }
    }
    DevCsr &dtrain = s->train_keep;
    const double ms_upload = since();
    int rs = ((s->d + 1 + 3) / 4) * 4;
    DCsr usf = s->usf.view(), itf = s->itf.view();
    usf.rows = test->rows;  // only users/items of the interaction matrix (PYX:1264, 1301)
    itf.rows = test->cols;
    LFM_TRY(urep.alloc((size_t)usf.rows * rs));
    // component-major item table: the MFMA sweep walks 2 * ceil-pow2(d / 2) rows, zero beyond the bias row
    const int irows_n = std::max(rs, ranks_mfma_supported(s->d) ? ranks_mfma2_item_rows(s->d) : rs);
    LFM_TRY(irep.alloc((size_t)itf.rows * irows_n));
    LFM_TRY(ieps.alloc((size_t)itf.rows * 2));
    LFM_TRY(tscores.alloc((size_t)test->nnz));
    HIP_TRY(hipMemsetAsync(irep.p, 0, (size_t)itf.rows * irows_n * sizeof(float), s->stream));
    LFM_TRY(dranks.upload(ranks, (size_t)test->nnz));
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    HIP_TRY(launch_rep_rows(usf, s->tab[1][0].p, s->tab[1][3].p, s->d, rs, urep.p, s->stream));
    HIP_TRY(launch_rep_rows(itf, s->tab[0][0].p, s->tab[0][3].p, s->d, rs, irep.p, s->stream, 1));
    // ... and row-major, for the exact scores of the test interactions (the thresholds of the MFMA sweeps)
    LFM_TRY(irows.alloc((size_t)itf.rows * rs));
    HIP_TRY(launch_rep_rows(itf, s->tab[0][0].p, s->tab[0][3].p, s->d, rs, irows.p, s->stream));
    RanksArgs a;
    a.user_rep = urep.p;
    a.item_rep = irep.p;
    a.rs = rs;
    a.d = s->d;
    a.test = dtest.view();
    a.train = dtrain.view();
    a.ranks = dranks.p;
    a.ulist = nullptr;
    a.n_ulist = 0;
    a.item_eps = ieps.p;
    a.test_scores = tscores.p;
    a.test_nnz = test->nnz;
    a.item_rows = irows_n;
    a.item_rows_rm = irows.p;
    // LIGHTFM_AMD_RANKS_MFMA: 0 the scalar kernel, 1 the first MFMA formulation (users as tile rows), 2 the
    // second (a lane owns a user, compare chain), 3 the third (a lane owns a user, bucket search, fp32 products), unset / 4
    // the third with the products on the bf16 matrix pipe (split operands); the tests compare all five
    const char *mfma_env = getenv("LIGHTFM_AMD_RANKS_MFMA");
    int mfma_mode = mfma_env == nullptr ? 4 : atoi(mfma_env);
    if (mfma_mode < 0 || mfma_mode > 4) mfma_mode = 4;  // anything but the five documented values: the default kernel
    a.item_bf = nullptr;
    // the MFMA sweeps mask train positives by walking each user's train row alongside the item tiles: they
    // need ascending column indices (tocsr() of a COO gives them; a CSR handed in by the caller may not).
    // Unsorted rows run the scalar kernel, whose lookup is the reference's binary search.  Checked on the
    // device (the host loop over ML-20M's 18 M train entries cost 15 ms per call).
    const bool check_sorted = mfma_mode != 0 && train->nnz > 1;
    if (check_sorted) {
        HIP_TRY(hipMemsetAsync(s->flag.p, 0, sizeof(int), s->stream));
        descents_kernel<<<(int)std::min<int64_t>(8192, ((int64_t)train->nnz / 4 + 256) / 256), 256, 0, s->stream>>>(
            dtrain.indices.p, train->nnz, s->flag.p);
        boundary_descents_kernel<<<(int)std::min<int64_t>(4096, ((int64_t)train->rows + 255) / 256), 256, 0, s->stream>>>(
            dtrain.indptr.p, dtrain.indices.p, train->rows, s->flag.p);
    }
    // (host, while the representation kernels and the check run) users with test interactions, in tiles of 32 per
    // wavefront, most test items first: the users of a wavefront then need the same number of threshold passes and
    // the heavy wavefronts are dispatched first.  A counting sort: the order of a stable sort by count.
    std::vector<int32_t> ul;
    if (mfma_mode != 0 && ranks_mfma_supported(s->d)) {
        int32_t cmax = 0;
        size_t n_with = 0;
        for (int32_t u = 0; u < test->rows; ++u) {
            const int32_t c = test->indptr[u + 1] - test->indptr[u];
            if (c > 0) {
                ++n_with;
                cmax = std::max(cmax, c);
            }
        }
        ul.resize(n_with);
        std::vector<size_t> start((size_t)cmax + 2, 0);  // bucket cmax - c: the heaviest users first
        if (mfma_mode != 1)
            for (int32_t u = 0; u < test->rows; ++u) {
                const int32_t c = test->indptr[u + 1] - test->indptr[u];
                if (c > 0) ++start[(size_t)(cmax - c) + 1];
            }
        for (size_t b = 1; b < start.size(); ++b) start[b] += start[b - 1];
        for (int32_t u = 0; u < test->rows; ++u) {
            const int32_t c = test->indptr[u + 1] - test->indptr[u];
            if (c > 0) ul[start[mfma_mode != 1 ? (size_t)(cmax - c) : 0]++] = u;
        }
    }
    // work items of the lane-per-user sweeps
    std::vector<int32_t> wl;
    const bool bucket_search = mfma_mode != 2 && ranks_mfma3_supported(s->d, itf.rows, irows_n);
    if (mfma_mode >= 2 && ranks_mfma_supported(s->d) && !bucket_search) {
        // ranks_mfma2_kernel: every 32-user tile x every pass of 16 test items its heaviest (first) user needs
        for (size_t t0 = 0; t0 < ul.size(); t0 += 32) {
            const int32_t u0 = ul[t0];
            const int32_t cnt = test->indptr[u0 + 1] - test->indptr[u0];
            for (int32_t p0 = 0; p0 < cnt; p0 += 16) {
                wl.push_back((int32_t)(t0 / 32));
                wl.push_back(p0);
            }
        }
    } else if (mfma_mode >= 2 && ranks_mfma_supported(s->d)) {
        // ranks_mfma3_kernel: (32-user tile, pass of up to 31 test items of its heaviest user, segment of the item
        // table).  Enough segments that every resident wavefront gets ~8 items (an even finish), none shorter than
        // 32 tiles of items; segment-major, heaviest passes first: the wavefronts resident at one time read the
        // same part of the table.
        const int32_t mt = ranks_mfma3_pass_items();
        std::vector<int32_t> passes;
        for (size_t t0 = 0; t0 < ul.size(); t0 += 32) {
            const int32_t u0 = ul[t0];
            const int32_t cnt = test->indptr[u0 + 1] - test->indptr[u0];
            for (int32_t p0 = 0; p0 < cnt; p0 += mt) {
                passes.push_back((int32_t)(t0 / 32));
                passes.push_back(p0);
            }
        }
        const int64_t n_pass = (int64_t)passes.size() / 2;
        const int64_t resident = (int64_t)std::max(s->cus, 1) * ranks_mfma3_waves_per_cu(s->d);
        const int64_t item_tiles = ((int64_t)test->cols + 31) / 32;
        int64_t segs = std::max<int64_t>(1, (8 * resident + n_pass - 1) / std::max<int64_t>(n_pass, 1));
        segs = std::min(segs, std::max<int64_t>(1, item_tiles / 32));
        segs = std::max(segs, (item_tiles * 32 + (1 << 25) - 1) / (1 << 25));  // < 2^26 items per segment
        if (const char *e = getenv("LIGHTFM_AMD_RANKS_SEGMENTS"))
            segs = std::max<int64_t>(1, std::min<int64_t>(atoi(e), item_tiles));
        const int64_t seg_tiles = (item_tiles + segs - 1) / segs;
        if (n_pass * segs > (int64_t)INT32_MAX / 4) return fail(LFM_EINVAL, "too many test interactions for one call");
        wl.reserve((size_t)(n_pass * segs * 4));
        for (int64_t sg = 0; sg < segs; ++sg) {
            const int64_t jb = sg * seg_tiles * 32, je = std::min<int64_t>((sg + 1) * seg_tiles * 32, test->cols);
            if (jb >= je) break;
            for (int64_t p = 0; p < n_pass; ++p) {
                wl.push_back(passes[2 * p]);
                wl.push_back(passes[2 * p + 1]);
                wl.push_back((int32_t)jb);
                wl.push_back((int32_t)je);
            }
        }
    }
    if (check_sorted) {
        int unsorted = 0;
        HIP_TRY(hipMemcpyAsync(&unsorted, s->flag.p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (unsorted) mfma_mode = 0;
    }
    a.work = nullptr;
    a.n_work = 0;
    if (mfma_mode != 0 && ranks_mfma_supported(s->d)) {
        LFM_TRY(ulist.upload(ul.data(), ul.size()));
        a.ulist = ulist.p;
        a.n_ulist = (int32_t)ul.size();
        if (mfma_mode == 1) {
            HIP_TRY(launch_ranks_mfma(a, s->stream, s->cus));
        } else {
            LFM_TRY(work.upload(wl.data(), wl.size()));
            a.work = work.p;
            a.n_work = (int32_t)(wl.size() / (bucket_search ? 4 : 2));
            if (bucket_search && mfma_mode == 4) {
                const size_t bf_bytes = ranks_mfma3_bf_bytes(s->d, itf.rows);
                if (bf_bytes) {
                    LFM_TRY(ibf.alloc((bf_bytes + 3) / 4));
                    a.item_bf = ibf.p;
                }
            }
            if (bucket_search) HIP_TRY(launch_ranks_mfma3(a, s->stream, s->cus));
            else HIP_TRY(launch_ranks_mfma2(a, s->stream, s->cus));
        }
    } else {
        HIP_TRY(launch_ranks(a, s->stream));
    }
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    const double ms_enqueued = since();
    HIP_TRY(hipStreamSynchronize(s->stream));
    const double ms_synced = since();
    HIP_TRY(hipEventElapsedTime(&g_kernel_ms, s->ev0, s->ev1));
    const int rc = dranks.download(ranks);
    if (timing)
        fprintf(stderr, "[lfm predict_ranks] uploads of test + train %.2f ms, everything enqueued at %.2f ms, device done at %.2f ms (kernels %.2f ms), "
                        "ranks downloaded at %.2f ms\n", ms_upload, ms_enqueued, ms_synced, (double)g_kernel_ms, since());
    return rc;
}

// ---------------------------------------------------- one-shot entry points ---

namespace {
struct SessionHolder {
    lfm_session *s = nullptr;
    ~SessionHolder() { lfm_session_destroy(s); }
};
}  // namespace

static int one_shot_fit(int loss, const lfm_csr *itf, const lfm_csr *usf, const lfm_csr *positives,
                        const int32_t *user_ids, const int32_t *item_ids, const float *Y,
                        const float *w, const int32_t *shuffle, int64_t n, lfm_model *model,
                        double ia, double ua, int32_t k, int32_t npos, const uint32_t *seeds,
                        int32_t n_seeds, lfm_opts *opts)
{
    if (n && !shuffle) return fail(LFM_EINVAL, "null shuffle_indices");
    SessionHolder h;
    LFM_TRY(lfm_session_create(&h.s, 0, model, itf, usf));
    LFM_TRY(lfm_session_set_interactions(h.s, positives, user_ids, item_ids, Y, w, n));
    LFM_TRY(lfm_session_upload_shuffle(h.s, 0, shuffle, n));
    LFM_TRY(lfm_session_epoch(h.s, loss, 0, ia, ua, k, npos, seeds, n_seeds, opts));
    return lfm_session_sync_to_host(h.s, model);
}

extern "C" int lfm_fit_warp(const lfm_csr *itf, const lfm_csr *usf, const lfm_csr *interactions,
                            const int32_t *user_ids, const int32_t *item_ids, const float *Y,
                            const float *sample_weight, const int32_t *shuffle, int64_t n,
                            lfm_model *model, double item_alpha, double user_alpha,
                            const uint32_t *seeds, int32_t n_seeds, lfm_opts *opts)
{
    return one_shot_fit(LFM_LOSS_WARP, itf, usf, interactions, user_ids, item_ids, Y, sample_weight,
                        shuffle, n, model, item_alpha, user_alpha, 0, 0, seeds, n_seeds, opts);
}

extern "C" int lfm_fit_bpr(const lfm_csr *itf, const lfm_csr *usf, const lfm_csr *interactions,
                           const int32_t *user_ids, const int32_t *item_ids, const float *Y,
                           const float *sample_weight, const int32_t *shuffle, int64_t n,
                           lfm_model *model, double item_alpha, double user_alpha,
                           const uint32_t *seeds, int32_t n_seeds, lfm_opts *opts)
{
    return one_shot_fit(LFM_LOSS_BPR, itf, usf, interactions, user_ids, item_ids, Y, sample_weight,
                        shuffle, n, model, item_alpha, user_alpha, 0, 0, seeds, n_seeds, opts);
}

extern "C" int lfm_fit_logistic(const lfm_csr *itf, const lfm_csr *usf, const int32_t *user_ids,
                                const int32_t *item_ids, const float *Y, const float *sample_weight,
                                const int32_t *shuffle, int64_t n, lfm_model *model,
                                double item_alpha, double user_alpha, lfm_opts *opts)
{
    return one_shot_fit(LFM_LOSS_LOGISTIC, itf, usf, nullptr, user_ids, item_ids, Y, sample_weight,
                        shuffle, n, model, item_alpha, user_alpha, 0, 0, nullptr, 0, opts);
}

extern "C" int lfm_fit_warp_kos(const lfm_csr *itf, const lfm_csr *usf, const lfm_csr *data,
                                const int32_t *user_ids, const int32_t *shuffle, int64_t n,
                                lfm_model *model, double item_alpha, double user_alpha, int32_t k,
                                int32_t n_positives, const uint32_t *seeds, int32_t n_seeds,
                                lfm_opts *opts)
{
    return one_shot_fit(LFM_LOSS_WARP_KOS, itf, usf, data, user_ids, nullptr, nullptr, nullptr, shuffle,
                        n, model, item_alpha, user_alpha, k, n_positives, seeds, n_seeds, opts);
}

extern "C" int lfm_predict(const lfm_csr *itf, const lfm_csr *usf, const int32_t *user_ids,
                           const int32_t *item_ids, float *predictions, int64_t n, const lfm_model *model)
{
    SessionHolder h;
    LFM_TRY(lfm_session_create_scoring(&h.s, 0, model, itf, usf));
    return lfm_session_predict(h.s, user_ids, item_ids, predictions, n);
}

extern "C" int lfm_predict_ranks(const lfm_csr *itf, const lfm_csr *usf, const lfm_csr *test,
                                 const lfm_csr *train, float *ranks, const lfm_model *model)
{
    SessionHolder h;
    LFM_TRY(lfm_session_create_scoring(&h.s, 0, model, itf, usf));
    return lfm_session_predict_ranks(h.s, test, train, ranks);
}

extern "C" int lfm_auc_from_rank(const lfm_csr *ranks, const int32_t *num_train_positives,
                                 float *rank_data, float *auc)
{
    LFM_TRY(validate_csr(ranks, "ranks"));
    if (!num_train_positives || !auc || (ranks->nnz && !rank_data)) return fail(LFM_EINVAL, "null buffer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(LFM_ENODEV, "no HIP device available (the HIP backend has no CPU fallback)");
    HIP_TRY(hipSetDevice(0));
    DevCsr dr;
    DBuf<int32_t> dntp;
    DBuf<float> drd, dauc;
    DrainOnExit drain(nullptr, true);
    LFM_TRY(dr.upload(ranks, false, true));
    LFM_TRY(dntp.upload(num_train_positives, (size_t)ranks->rows));
    LFM_TRY(dauc.upload(auc, (size_t)ranks->rows));
    DCsr v = dr.view();
    float *rd = dr.data.p;  // the reference passes ranks.data as rank_data (evaluation.py:247-249)
    if (rank_data != ranks->data) {
        LFM_TRY(drd.upload(rank_data, (size_t)ranks->nnz));
        rd = drd.p;
    }
    HIP_TRY(launch_auc(v, dntp.p, rd, dauc.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    if (ranks->nnz) HIP_TRY(hipMemcpy(rank_data, rd, (size_t)ranks->nnz * sizeof(float), hipMemcpyDeviceToHost));
    return dauc.download(auc);
}

__global__ void in_positives_kernel(DCsr m, int row, int col, int *out)
{
    bool r = in_positives(m, col, row, lane_id());
    if (threadIdx.x == 0) *out = r ? 1 : 0;
}

extern "C" int lfm_in_positives(int32_t row, int32_t col, const lfm_csr *mat)
{
    LFM_TRY(validate_csr(mat, "mat"));
    if (row < 0 || row >= mat->rows) return fail(LFM_EINVAL, "row out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(LFM_ENODEV, "no HIP device available (the HIP backend has no CPU fallback)");
    HIP_TRY(hipSetDevice(0));
    DevCsr dm;
    DBuf<int> out;
    DrainOnExit drain(nullptr, true);
    LFM_TRY(dm.upload(mat, false, false));
    LFM_TRY(out.alloc(1));
    in_positives_kernel<<<1, WAVE>>>(dm.view(), row, col, out.p);
    HIP_TRY(hipGetLastError());
    int r = 0;
    HIP_TRY(hipMemcpy(&r, out.p, sizeof(int), hipMemcpyDeviceToHost));
    return r;
}
