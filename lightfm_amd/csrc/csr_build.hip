// csr_build.hip -- the positives lookup matrix built ON DEVICE from the training COO.
//
// Replaces the host-side `interactions.tocsr()` + `sorted_indices()` of
// lightfm/lightfm.py:365-372 (LFM:681-686 rebuilds it every epoch; 0.4-0.8 s per 20 M entries on a
// host core, more than ten device epochs): a radix sort of the 64-bit keys user * n_items + item
// (only the bits the shape needs), removal of duplicate keys (tocsr() sums duplicates into one
// entry, and in_positives, PYX:270-284, never reads the values) and one lower-bound search per row
// for the row pointers.  rocPRIM (via hipCUB) supplies the sort and the unique pass.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.hpp"
#include "pool.hpp"

namespace lfm {

__global__ void make_keys_kernel(const int32_t *user_ids, const int32_t *item_ids, int64_t n, uint64_t n_items,
                                 uint64_t *keys)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) keys[j] = (uint64_t)(uint32_t)user_ids[j] * n_items + (uint64_t)(uint32_t)item_ids[j];
}

__global__ void split_keys_kernel(const uint64_t *keys, int64_t n, uint64_t n_items, int32_t *indices)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) indices[j] = (int32_t)(keys[j] % n_items);
}

// indptr[r] = number of keys below r * n_items (keys sorted ascending, unique)
__global__ void row_pointers_kernel(const uint64_t *keys, int64_t n, uint64_t n_items, int32_t n_rows, int32_t *indptr)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = t; r <= n_rows; r += st) {
        const uint64_t bound = (uint64_t)r * n_items;
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (keys[mid] < bound) lo = mid + 1; else hi = mid;
        }
        indptr[r] = (int32_t)lo;
    }
}

// Bloom filter over the rows of the positives lookup (device.hpp: Bloom): eight lanes walk a row.
__global__ void bloom_build_kernel(const int32_t *indptr, const int32_t *indices, int32_t n_rows, uint32_t *bloom)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    const int sub = (int)(t & 7);
    for (int64_t r = t >> 3; r < n_rows; r += st >> 3) {
        const int lo = indptr[r], hi = indptr[r + 1];
        for (int k = lo + sub; k < hi; k += 8) {
            const uint32_t h = Bloom::mix((uint32_t)indices[k]);
            atomicOr(bloom + Bloom::word(h, lo, hi), Bloom::mask(h));
        }
    }
}

hipError_t build_positives_bloom(const int32_t *indptr, const int32_t *indices, int32_t n_rows, int64_t nnz, uint32_t *bloom,
                                 hipStream_t st)
{
    hipError_t e = hipMemsetAsync(bloom, 0, (size_t)Bloom::words(nnz) * sizeof(uint32_t), st);
    if (e != hipSuccess || n_rows <= 0 || nnz <= 0) return e;
    const int grid = (int)std::min<int64_t>(8192, ((int64_t)n_rows * 8 + 255) / 256);
    bloom_build_kernel<<<grid, 256, 0, st>>>(indptr, indices, n_rows, bloom);
    return hipGetLastError();
}

static int bits_for(uint64_t v)
{
    int b = 1;
    while (b < 64 && (v >> b) != 0) ++b;
    return b;
}

// indices_out: device buffer of n int32 (capacity); indptr_out: device buffer of n_users + 1.
// *nnz_out = number of distinct (user, item) pairs.
hipError_t build_positives_csr(const int32_t *user_ids, const int32_t *item_ids, int64_t n, int32_t n_users,
                               int32_t n_items, int32_t *indices_out, int32_t *indptr_out, int64_t *nnz_out,
                               hipStream_t st)
{
    *nnz_out = 0;
    if (n_users < 0 || n_items <= 0) return hipErrorInvalidValue;
    if (n == 0) return hipMemsetAsync(indptr_out, 0, ((size_t)n_users + 1) * sizeof(int32_t), st);
    if (n > 0x7fffffffLL) return hipErrorInvalidValue;  // the unique pass counts in int
    hipError_t e;
    uint64_t *k0 = nullptr, *k1 = nullptr;
    int *d_count = nullptr;
    void *tmp = nullptr;
    // the temporaries come from the library's pool (pool.hpp) and go back there once the stream has drained
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(st);
        pool_free(k0);
        pool_free(k1);
        pool_free(d_count);
        pool_free(tmp);
    };
#define CSR_TRY(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    CSR_TRY(pool_alloc((void **)&k0, (size_t)n * sizeof(uint64_t)));
    CSR_TRY(pool_alloc((void **)&k1, (size_t)n * sizeof(uint64_t)));
    CSR_TRY(pool_alloc((void **)&d_count, sizeof(int)));
    const int grid = (int)std::min<int64_t>(8192, (n + 255) / 256);
    make_keys_kernel<<<grid, 256, 0, st>>>(user_ids, item_ids, n, (uint64_t)n_items, k0);
    const int end_bit = std::min(64, bits_for((uint64_t)n_users * (uint64_t)n_items));
    hipcub::DoubleBuffer<uint64_t> keys(k0, k1);
    size_t sort_bytes = 0, uniq_bytes = 0;
    CSR_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys, (int)n, 0, end_bit, st));
    CSR_TRY(hipcub::DeviceSelect::Unique(nullptr, uniq_bytes, k0, k1, d_count, (int)n, st));
    CSR_TRY(pool_alloc(&tmp, std::max(sort_bytes, uniq_bytes)));
    CSR_TRY(hipcub::DeviceRadixSort::SortKeys(tmp, sort_bytes, keys, (int)n, 0, end_bit, st));
    uint64_t *sorted = keys.Current(), *uniq = keys.Alternate();
    CSR_TRY(hipcub::DeviceSelect::Unique(tmp, uniq_bytes, sorted, uniq, d_count, (int)n, st));
    int count = 0;
    CSR_TRY(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
    CSR_TRY(hipStreamSynchronize(st));
    split_keys_kernel<<<grid, 256, 0, st>>>(uniq, (int64_t)count, (uint64_t)n_items, indices_out);
    const int rgrid = (int)std::min<int64_t>(4096, ((int64_t)n_users + 1 + 255) / 256);
    row_pointers_kernel<<<rgrid, 256, 0, st>>>(uniq, (int64_t)count, (uint64_t)n_items, n_users, indptr_out);
    CSR_TRY(hipGetLastError());
    CSR_TRY(hipStreamSynchronize(st));
#undef CSR_TRY
    cleanup();
    *nnz_out = count;
    return hipSuccess;
}

// ids_out[0 .. *count) = the positions j < n with flags[j] != 0, ascending (multi-GPU merge: the rows touched
// since the last merge).  ids_out has capacity n.  Synchronises the stream (the count goes to the host).
hipError_t compact_flagged_rows(const unsigned char *flags, int64_t n, int32_t *ids_out, int64_t *count, hipStream_t st)
{
    *count = 0;
    if (n <= 0) return hipSuccess;
    if (n > 0x7fffffffLL) return hipErrorInvalidValue;
    hipError_t e;
    int *d_count = nullptr;
    void *tmp = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(st);
        pool_free(d_count);
        pool_free(tmp);
    };
#define CMP_TRY(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    CMP_TRY(pool_alloc((void **)&d_count, sizeof(int)));
    hipcub::CountingInputIterator<int32_t> rows(0);
    size_t bytes = 0;
    CMP_TRY(hipcub::DeviceSelect::Flagged(nullptr, bytes, rows, flags, ids_out, d_count, (int)n, st));
    CMP_TRY(pool_alloc(&tmp, std::max<size_t>(bytes, 16)));
    CMP_TRY(hipcub::DeviceSelect::Flagged(tmp, bytes, rows, flags, ids_out, d_count, (int)n, st));
    int c = 0;
    CMP_TRY(hipMemcpyAsync(&c, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
    CMP_TRY(hipStreamSynchronize(st));
#undef CMP_TRY
    cleanup();
    *count = c;
    return hipSuccess;
}

}  // namespace lfm
