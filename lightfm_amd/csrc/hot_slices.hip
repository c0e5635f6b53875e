// hot_slices.hip -- the shared ("hot") item-feature rows of a hybrid model, accumulated in LDS instead of through the L2
// float-atomic unit (round 6; device.hpp: HotRec; the HOT instantiations of feat_kernel.hpp write the records).
// PYX = /root/reference/lightfm/_lightfm_fast.pyx.template
//
// Why.  BASELINE config C3 (BPR, d = 128, item features [identity | 8 tags of 1 128]) updates 19 feature rows per
// interaction, 16 of them among 1 128 tag rows: 4 632 float atomics per interaction, and the chip's L2 atomic units
// perform 320 G of them per second whatever the table (tools/membench.hip) -- C3 sat at 0.66 of that rate and at 0.28 of
// the HBM roofline for four rounds.  Only taking those rows out of the atomic unit removes the charge.
//
// How.  W and G of all hot rows are 1.16 MB: they fit the LDS of a CU only SLICED BY COMPONENT.  A workgroup owns CS
// components of every hot row (CS = 8: 1 128 rows x 8 cells x {W, G} = 72 KB), one more slice owns the bias cells; each
// slice has n_rep replicas that split the launch's records between them.  A workgroup
//   1. loads its slice from the SNAPSHOT of the hot rows taken when the records' launch had finished (hot_snapshot_kernel),
//   2. walks its share of the records in order; per record and job (positive item, negative item: PYX:537-649) a pass of
//      the wavefront applies the job's entries -- lane = (entry, component) -- with the reference's float64 cell
//      arithmetic (device.hpp: cell_math, PYX:416-449) on the LDS cells: plain read-modify-write, no atomics.  (LDS float
//      atomics are no alternative: ds_add_f32 runs at 199 G dwords/s chip-wide, BELOW the L2 units' 320 G; plain LDS
//      read-modify-writes with the float64 cell run at 850-1 040 G dwords/s, profiles/r06_membench_lds.txt.)  The
//      wavefronts of a workgroup share the slice like the reference's OpenMP threads share the tables -- an
//      unsynchronised read-modify-write, Hogwild exactly as PYX does it -- and within a wavefront the passes of a record
//      are sequential, so a tag shared by the positive and the negative item sees its first update (the row-stream
//      kernel's "generations");
//   3. publishes cell - snapshot with one float atomic per cell: n_rep atomics per cell and launch instead of one per
//      cell and interaction.
// Semantics: a hot row's updates are applied to a copy that is as old as the launch is long (the session keeps such
// launches short, session.hip: hot_chunk) and the replicas' changes are summed -- the merge arithmetic of the multi-GPU
// path (LFM_MERGE_SUM) in small, for which DESIGN.md "Multi-GPU" has the measured tolerance of exactly these rows.
// With ONE replica and one record per launch the result is the sequential one (tests/test_hot_slices.py).
#include "device.hpp"
#include "kernels.hpp"

namespace lfm {

namespace {

// snapshot of the hot rows: snapW / snapG [hot_n][d], snapb / snapbG [hot_n]
__global__ void hot_snapshot_kernel(HotArgs a)
{
    const int64_t cells = (int64_t)a.hot_n * a.d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i / a.d), c = (int)(i - (int64_t)slot * a.d);
        const size_t src = (size_t)a.rows[slot] * a.d + c;
        a.snapW[i] = a.W[src];
        a.snapG[i] = a.G[src];
        if (c == 0) {
            a.snapb[slot] = a.b[a.rows[slot]];
            a.snapbG[slot] = a.bG[a.rows[slot]];
        }
    }
}

// One workgroup = one (slice, replica).  CS: components per slice (lanes per entry); the last slice (index d / CS)
// holds the bias cells and runs with one lane per entry.
template <int CS>
__global__ __launch_bounds__(1024) void hot_slice_kernel(HotArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float2 cells[];  // [hot_n][CS] (W, G); bias slice: [hot_n]
    const int n_comp_slices = a.d / CS;
    const int slice = (int)blockIdx.x / a.n_rep, rep = (int)blockIdx.x - slice * a.n_rep;
    const bool bias = slice == n_comp_slices;
    const int width = bias ? 1 : CS;            // cells per hot row in this slice
    const int c0 = slice * CS;                  // first component of the slice
    const int n_cells = a.hot_n * width;
    for (int i = threadIdx.x; i < n_cells; i += blockDim.x) {
        if (bias) cells[i] = make_float2(a.snapb[i], a.snapbG[i]);
        else {
            const int slot = i / CS, c = i - slot * CS;
            cells[i] = make_float2(a.snapW[(size_t)slot * a.d + c0 + c], a.snapG[(size_t)slot * a.d + c0 + c]);
        }
    }
    __syncthreads();

    const int lane = lane_id(), wib = uni((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    const int EPP = WAVE / width;               // entries per pass: 8 for CS = 8, 64 for the bias slice
    const int k = lane / width, c = lane - k * width;
    const Hyper h{0, a.lr, a.rho, a.eps};
    const int64_t stride = (int64_t)a.n_rep * wpb;
    int64_t r = (int64_t)rep * wpb + wib;

    // software pipeline, one record deep: everything a record needs is requested while the previous one is applied (two deep
    // measured slower: 95.8 against 100.0 M/s on C3 -- 62 instead of 53 VGPRs; profiles/r06_c3_tuning.txt)
    int n_total = 0, cnt = 0, eslot = 0;
    float ew = 0.0f, x = 0.0f;
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
    auto request = [&](int64_t rr) {
        n_total = 0;
        if (rr < a.n_rec) {
            const HotRec *rec = a.rec + rr;
            n_total = rec->n_total;
            cnt = *reinterpret_cast<const int *>(rec->cnt);
            g0 = rec->g[0];
            g1 = rec->g[1];
            g2 = rec->g[2];
            const HotRec::Entry en = rec->e[lane < HOT_EMAX ? lane : 0];  // lane t holds entry t
            eslot = en.slot;
            ew = en.w;
            x = bias ? 1.0f : a.x[(size_t)rr * a.d + c0 + c];
        }
    };
    request(r);
    for (; r < a.n_rec; r += stride) {
        const int nt = uni(n_total), cn = uni(cnt);
        const double cg0 = unid(g0), cg1 = unid(g1), cg2 = unid(g2);
        const int cslot = eslot;
        const float cw = ew, cx = x;
        request(r + stride);
        if (nt <= 0) continue;
        int first = 0;
#pragma unroll 1
        for (int j = 0; j < 3; ++j) {
            const int nj = (cn >> (8 * j)) & 0xff;
            const double gj = j == 0 ? cg0 : (j == 1 ? cg1 : cg2);
            // a pass never mixes jobs: the entries of one job name different rows (a CSR row holds a column once), so a
            // pass's cells are disjoint; the next job's pass reads what this one wrote (LDS operations of a wavefront
            // complete in order)
            for (int p0 = 0; p0 < nj; p0 += EPP) {
                const int t = first + p0 + k;                      // the entry this lane works on
                const int slot = __shfl(cslot, t & (WAVE - 1), WAVE);
                const float w = __shfl(cw, t & (WAVE - 1), WAVE);
                if (p0 + k < nj) {
                    float2 *cell = cells + (size_t)slot * width + c;
                    const float2 o = *cell;
                    float nW, nG;
                    // PYX:602-638: gradient = g_job * x[component]; bias cells (PYX:571-599): gradient = g_job.
                    // (device.hpp: cell_math_adagrad = cell_math's adagrad cell, bit for bit, without the float64 root and
                    // quotient wherever the result cannot depend on them)
                    cell_math_adagrad(o.x, o.y, (double)w, bias ? gj : gj * (double)cx, h.lr, nW, nG);
                    *cell = make_float2(nW, nG);
                }
            }
            first += nj;
        }
    }
    __syncthreads();
    // publication: what this replica changed, one float atomic per cell that moved
    for (int i = threadIdx.x; i < n_cells; i += blockDim.x) {
        const float2 v = cells[i];
        if (bias) {
            const float dW = __fsub_rn(v.x, a.snapb[i]), dG = __fsub_rn(v.y, a.snapbG[i]);
            const int row = a.rows[i];
            if (dW != 0.0f) atomicAdd(a.b + row, dW);
            if (dG != 0.0f) atomicAdd(a.bG + row, dG);
        } else {
            const int slot = i / CS, cc = i - slot * CS;
            const size_t s = (size_t)slot * a.d + c0 + cc, dst = (size_t)a.rows[slot] * a.d + c0 + cc;
            const float dW = __fsub_rn(v.x, a.snapW[s]), dG = __fsub_rn(v.y, a.snapG[s]);
            if (dW != 0.0f) atomicAdd(a.W + dst, dW);
            if (dG != 0.0f) atomicAdd(a.G + dst, dG);
        }
    }
}

// Self-test of cell_math_adagrad against cell_math (lfm_selftest_adagrad_cell): pseudo-random cells over the ranges training
// produces and beyond -- W in +-[2^-20, 2^4], G in [2^-10, 2^30], weights 1 and (0.1, 2), gradients +-[2^-30, 2^7] -- counted:
// out[0] = cells whose (nW, nG) bit patterns differ (must be 0), out[1] = cells that took the exact fallback.
__global__ void selftest_adagrad_kernel(int64_t n, uint32_t seed, float lr_f, unsigned long long *out)
{
    unsigned long long bad = 0, slow = 0;
    const Hyper h{0, lr_f, 0.95f, 1e-6f};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t s = position_seed(seed, (uint64_t)i);
        auto next = [&]() { s = lcg(s); return (s >> 8) * (1.0f / 16777216.0f); };  // [0, 1)
        const float oW = (next() < 0.5f ? -1.0f : 1.0f) * exp2f(-20.0f + 24.0f * next()) * (1.0f + next());
        const float oG = exp2f(-10.0f + 40.0f * next() * next()) * (1.0f + next());
        const double w = next() < 0.5f ? 1.0 : (double)(0.1f + 1.9f * next());
        const double g = (double)((next() < 0.5f ? -1.0f : 1.0f) * exp2f(-30.0f + 37.0f * next()) * (1.0f + next())) * (double)(0.5f + next());
        float nW, nG, nM, fW, fG;
        double lr;
        cell_math(oW, oG, 0.0f, w, g, h, 0.0, nW, nG, nM, lr);
        cell_math_adagrad(oW, oG, w, g, lr_f, fW, fG);
        if (__float_as_int(nW) != __float_as_int(fW) || __float_as_int(nG) != __float_as_int(fG)) ++bad;
        // the fallback's condition, restated
        const double G = (double)oG;
        double y = __builtin_amdgcn_rsq(G);
        for (int it = 0; it < 2; ++it) y = __builtin_fma(0.5 * y, __builtin_fma(-G * y, y, 1.0), y);
        const double t2 = (((double)lr_f * y) * w) * g, d = (double)oW - t2;
        const float m = (float)d;
        const int mb = __float_as_int(m) & 0x7f800000;
        const double hu = (double)__int_as_float(mb - (24 << 23));
        if (!(mb >= (25 << 23) && mb < 0x7f800000 && (__float_as_int(m) & 0x007fffff) != 0 && (hu - fabs(d - (double)m)) > 0x1p-49 * fabs(t2) + 0x1p-51 * fabs(d))) ++slow;
    }
    if (bad) atomicAdd(out, bad);
    if (slow) atomicAdd(out + 1, slow);
}

// occurrences of every column of a CSR (the session derives the hot set from them)
__global__ void column_count_kernel(const int32_t *indices, int64_t nnz, int32_t cols, int32_t *counts)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t col = indices[i];
        if (col >= 0 && col < cols) atomicAdd(counts + col, 1);
    }
}

}  // namespace

hipError_t launch_selftest_adagrad(int64_t n, uint32_t seed, float lr, unsigned long long *out, hipStream_t st)
{
    selftest_adagrad_kernel<<<2048, 256, 0, st>>>(n, seed, lr, out);
    return hipGetLastError();
}

int hot_slice_components(int hot_n, int d)
{
    // the largest slice width whose (W, G) cells of all hot rows leave room for two workgroups per CU
    for (int cs : {8, 4, 2})
        if (d % cs == 0 && (size_t)hot_n * cs * sizeof(float2) <= 76 * 1024) return cs;
    return 0;
}

hipError_t launch_column_counts(const int32_t *indices, int64_t nnz, int32_t cols, int32_t *counts, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)cols * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    if (nnz > 0) {
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (nnz + 255) / 256));
        column_count_kernel<<<grid, 256, 0, st>>>(indices, nnz, cols, counts);
    }
    return hipGetLastError();
}

hipError_t launch_hot_slices(const HotArgs &a, int cs, int threads, hipStream_t st)
{
    if (a.hot_n <= 0 || a.n_rec <= 0) return hipSuccess;
    const int64_t cells = (int64_t)a.hot_n * a.d;
    hot_snapshot_kernel<<<(int)std::max<int64_t>(1, std::min<int64_t>(1024, (cells + 255) / 256)), 256, 0, st>>>(a);
    const int grid = (a.d / cs + 1) * a.n_rep;
    const size_t smem = (size_t)a.hot_n * cs * sizeof(float2);
    hipError_t e = hipSuccess;
    auto go = [&](auto kernel) {
        static thread_local const void *raised = nullptr;
        if (smem > 64 * 1024 && raised != (const void *)kernel) {
            e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised = (const void *)kernel;
        }
        if (e == hipSuccess) kernel<<<grid, threads, smem, st>>>(a);
    };
    switch (cs) {
    case 8: go(hot_slice_kernel<8>); break;
    case 4: go(hot_slice_kernel<4>); break;
    case 2: go(hot_slice_kernel<2>); break;
    default: return hipErrorInvalidValue;
    }
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace lfm
