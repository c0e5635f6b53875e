// tools/ipc_probe.hip -- can two PROCESSES on one MI355X share device allocations through HIP IPC, and do
// global_atomic_add_f32 from both land?  (The multi-process form of the owner-sharded item tables needs exactly this;
// on a multi-GPU node the same handles map the owners' memory over xGMI.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/ipc_probe tools/ipc_probe.hip && tools/_bin/ipc_probe
// The parent forks BEFORE touching HIP; handles travel over a pipe.  Tested per allocation flavour: hipMalloc,
// hipExtMallocWithFlags(hipDeviceMallocUncached), hipDeviceMallocFinegrained, and an interior pointer of a hipMalloc block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%s] %s -> %s\n", who, #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void add_kernel(float *p, size_t n, int rounds)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < rounds; ++r)
        for (size_t j = i; j < n; j += st) atomicAdd(p + j, 1.0f);
}

static const size_t N = 1 << 20;
static const int ROUNDS = 50;
struct Msg { hipIpcMemHandle_t h[4]; int ok[4]; };

static int xread(int fd, void *b, size_t n) { char *p = (char *)b; while (n) { ssize_t r = read(fd, p, n); if (r <= 0) return -1; p += r; n -= r; } return 0; }
static int xwrite(int fd, const void *b, size_t n) { const char *p = (const char *)b; while (n) { ssize_t r = write(fd, p, n); if (r <= 0) return -1; p += r; n -= r; } return 0; }

int main()
{
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 2;
    pid_t pid = fork();
    const char *who = pid ? "owner" : "peer";
    if (pid == 0) {
        Msg m;
        if (xread(p2c[0], &m, sizeof(m))) return 3;
        CK(hipSetDevice(0));
        float *q[4] = {nullptr, nullptr, nullptr, nullptr};
        int opened[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; ++k) {
            if (!m.ok[k]) continue;
            hipError_t e = hipIpcOpenMemHandle((void **)&q[k], m.h[k], hipIpcMemLazyEnablePeerAccess);
            opened[k] = e == hipSuccess;
            if (!opened[k]) fprintf(stderr, "[peer] open %d -> %s\n", k, hipGetErrorString(e));
        }
        if (xwrite(c2p[1], opened, sizeof(opened))) return 3;
        char go;
        if (xread(p2c[0], &go, 1)) return 3;
        for (int k = 0; k < 4; ++k)
            if (opened[k]) add_kernel<<<512, 256>>>(q[k], N, ROUNDS);
        CK(hipDeviceSynchronize());
        for (int k = 0; k < 4; ++k)
            if (opened[k]) (void)hipIpcCloseMemHandle(q[k]);
        if (xwrite(c2p[1], &go, 1)) return 3;
        return 0;
    }
    CK(hipSetDevice(0));
    float *b[4] = {nullptr, nullptr, nullptr, nullptr}, *block = nullptr;
    const char *names[4] = {"hipMalloc", "uncached", "finegrained", "interior-of-hipMalloc"};
    Msg m;
    memset(&m, 0, sizeof(m));
    CK(hipMalloc((void **)&b[0], N * 4));
    if (hipExtMallocWithFlags((void **)&b[1], N * 4, hipDeviceMallocUncached) != hipSuccess) b[1] = nullptr;
    if (hipExtMallocWithFlags((void **)&b[2], N * 4, hipDeviceMallocFinegrained) != hipSuccess) b[2] = nullptr;
    CK(hipMalloc((void **)&block, N * 4 + (1 << 20)));
    b[3] = block + (1 << 18);  // 1 MiB into the block
    for (int k = 0; k < 4; ++k) {
        if (!b[k]) { printf("%-22s allocation failed\n", names[k]); continue; }
        CK(hipMemset(b[k], 0, N * 4));
        hipError_t e = hipIpcGetMemHandle(&m.h[k], b[k]);
        m.ok[k] = e == hipSuccess;
        if (!m.ok[k]) printf("%-22s hipIpcGetMemHandle -> %s\n", names[k], hipGetErrorString(e));
    }
    CK(hipDeviceSynchronize());
    if (xwrite(p2c[1], &m, sizeof(m))) return 3;
    int opened[4];
    if (xread(c2p[0], opened, sizeof(opened))) { fprintf(stderr, "peer died\n"); return 4; }
    char go = 1;
    if (xwrite(p2c[1], &go, 1)) return 3;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int k = 0; k < 4; ++k)
        if (b[k]) add_kernel<<<512, 256>>>(b[k], N, ROUNDS);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (xread(c2p[0], &go, 1)) { fprintf(stderr, "peer died\n"); return 4; }
    int status = 0;
    waitpid(pid, &status, 0);
    std::vector<float> h(N);
    int rc = 0;
    for (int k = 0; k < 4; ++k) {
        if (!b[k]) continue;
        CK(hipMemcpy(h.data(), b[k], N * 4, hipMemcpyDeviceToHost));
        const float want = (float)(ROUNDS * (opened[k] ? 2 : 1));
        size_t bad = 0;
        for (size_t i = 0; i < N; ++i) bad += h[i] != want;
        printf("%-22s handle %s, peer open %s, cells == %g: %s (%zu off, first %g)\n", names[k], m.ok[k] ? "ok" : "FAILED",
               opened[k] ? "ok" : "FAILED", want, bad ? "NO" : "yes", bad, h[0]);
        if (k < 2 && (bad || !opened[k])) rc = 1;
    }
    printf("owner's four add kernels: %.2f ms; peer exit %d\n", ms, WEXITSTATUS(status));
    return rc;
}
