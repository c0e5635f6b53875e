// coherence_test.hip -- is a buffer written by kernel A visible to kernel B of the SAME stream when B's
// workgroups run on other XCDs than the ones that wrote the lines?  (The hunt of the round-3 process abort:
// DESIGN.md "Root cause of the round-2 abort".)
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/coherence_test tools/coherence_test.hip
//   tools/_bin/coherence_test [rounds]
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void write_kernel(int32_t *out, int64_t n, int32_t tag)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) out[j] = tag + (int32_t)(j & 0xffff);
}
// every wavefront reads ONE element (like the epoch kernels read shuffle[i]) of a slice [begin, end)
__global__ void read_kernel(const int32_t *in, int64_t begin, int64_t end, int32_t tag, unsigned long long *bad, int32_t *first)
{
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t i = begin + gw; i < end; i += nw) {
        const int32_t v = in[i];
        if (v != tag + (int32_t)(i & 0xffff) && (threadIdx.x & 63) == 0) {
            if (atomicAdd(bad, 1ull) == 0ull) { first[0] = (int32_t)i; first[1] = v; first[2] = (int32_t)(__builtin_amdgcn_s_getreg(20 | (3 << 11))); }
        }
    }
}
__global__ void make_keys(uint64_t *k, int64_t n, uint32_t seed)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st) { uint32_t h = (uint32_t)j * 2654435761u + seed; h ^= h >> 15; k[j] = 100000ull + (uint64_t)(h % 1300000u); }
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;  // 0: reuse of freed sort buffers (the product's sequence), 1: one buffer kept, no free
    const int64_t n = 92820;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *bad; int32_t *first;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first, 12));
    unsigned long long total_bad = 0; int bad_rounds = 0;
    int32_t *keep = nullptr;
    if (mode == 1) CK(hipMalloc(&keep, n * 4));
    for (int r = 0; r < rounds; ++r) {
        int32_t *buf = keep;
        if (mode == 0) {
            // the product's sequence: rocPRIM sort temporaries are allocated, used and freed, then the buffer is allocated
            uint64_t *k0, *k1; void *tmp = nullptr; size_t tb = 0;
            CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8));
            make_keys<<<363, 256, 0, st>>>(k0, n, (uint32_t)r);
            hipcub::DoubleBuffer<uint64_t> keys(k0, k1);
            CK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, keys, (int)n, 0, 21, st));
            CK(hipMalloc(&tmp, tb));
            CK(hipcub::DeviceRadixSort::SortKeys(tmp, tb, keys, (int)n, 0, 21, st));
            CK(hipStreamSynchronize(st));
            CK(hipFree(k0)); CK(hipFree(k1)); CK(hipFree(tmp));
            CK(hipMalloc(&buf, n * 4));
        }
        CK(hipMemsetAsync(bad, 0, 8, st));
        for (int e = 0; e < 4; ++e) {
            const int32_t tag = (r * 4 + e) << 16;
            write_kernel<<<363, 256, 0, st>>>(buf, n, tag);
            for (int64_t b = 0; b < n; b += 56) read_kernel<<<14, 256, 0, st>>>(buf, b, b + 56 < n ? b + 56 : n, tag, bad, first);
        }
        unsigned long long hb = 0; int32_t hf[3] = {0, 0, 0};
        CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, st));
        CK(hipMemcpyAsync(hf, first, 12, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        if (hb) {
            if (bad_rounds < 8) printf("round %d: %llu stale reads, first at [%d] = %d (0x%x) on XCC %d, buffer %p\n", r, hb, hf[0], hf[1], hf[1], hf[2], (void *)buf);
            ++bad_rounds; total_bad += hb;
        }
        if (mode == 0) CK(hipFree(buf));
    }
    printf("mode %d: %d of %d rounds saw stale reads (%llu reads)\n", mode, bad_rounds, rounds, total_bad);
    return 0;
}
