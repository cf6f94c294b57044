// stepshape.hip -- a recorded small-shard suite step reduced to its launch shape, bare HIP, no torch: does the stream the step is bound
// to (the NULL stream, or a stream the caller created) change what the step costs?
//   M   (the caller's stream)  one long grid: 840 two-wave workgroups with 32 KB of LDS, ~1.0 ms each      (the LONG job grid)
//   A1  (side stream)          eight short grids of 6 250 four-wave blocks (~20-60 us each)                 (the ROW chain)
//   A2  (side stream)          one grid of 40 two-wave workgroups, ~300 us                                  (the producers)
//   A3  (side stream)          waits for A1's second kernel and for A2, then two grids of 40 x ~300 us       (the links)
//   fork: an event on M that A1..A3 wait for;  join: M waits for the last event of A1..A3;  50 steps back to back.
// Variants: M = NULL stream | created (non-blocking, normal priority) | created (highest priority); side streams at the highest | normal
// priority; events with hipEventDisableTiming alone | + hipEventReleaseToDevice | + hipEventDisableSystemFence.
// Ideal: the step is its long grid (~1.0 ms); everything above that is launch / dependency cost.
// build: hipcc -O3 --offload-arch=gfx950 stepshape.hip -o stepshape
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long ticks, unsigned long long *sink) { // wall_clock64: 100 MHz
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = wall_clock64();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) { *sink = t; lds[0] = 1; }
}
// which compute pipe of the command processor serves a stream's hardware queue: HW_ID (hwreg 4) bits 7:6 = pipe, 26:24 = queue slot, 31:30 = micro-engine
__global__ void probe(unsigned *out) { *out = __builtin_amdgcn_s_getreg((31 << 11) | 4); }
static unsigned hw_id_of(hipStream_t st, unsigned *dev) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, st, dev);
    CK(hipStreamSynchronize(st));
    unsigned h = 0;
    CK(hipMemcpy(&h, dev, 4, hipMemcpyDeviceToHost));
    return h;
}
static unsigned g_pipes[4];
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Cfg { int main_kind; int aux_prio; unsigned ev_flags; const char *name; };

static double run(const Cfg &c, int steps, unsigned long long *sink) {
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t M = nullptr, A[3];
    if (c.main_kind == 1) CK(hipStreamCreateWithPriority(&M, hipStreamNonBlocking, 0));
    if (c.main_kind == 2) CK(hipStreamCreateWithPriority(&M, hipStreamNonBlocking, hi));
    if (c.main_kind == 3) CK(hipStreamCreateWithFlags(&M, hipStreamDefault)); // a blocking stream
    for (auto &a : A) CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, c.aux_prio ? hi : 0));
    hipEvent_t fork, head, done[3];
    CK(hipEventCreateWithFlags(&fork, c.ev_flags));
    CK(hipEventCreateWithFlags(&head, c.ev_flags));
    for (auto &e : done) CK(hipEventCreateWithFlags(&e, c.ev_flags));
    auto step = [&]() {
        CK(hipEventRecord(fork, M));
        hipLaunchKernelGGL(spin, dim3(840), dim3(128), 32768, M, 100000ULL, sink);
        for (auto &a : A) CK(hipStreamWaitEvent(a, fork, 0));
        hipLaunchKernelGGL(spin, dim3(40), dim3(128), 24576, A[1], 30000ULL, sink + 1);
        CK(hipEventRecord(done[1], A[1]));
        for (int k = 0; k < 8; k++) {
            hipLaunchKernelGGL(spin, dim3(6250), dim3(256), 0, A[0], 200ULL, sink + 2);
            if (k == 1) CK(hipEventRecord(head, A[0]));
        }
        CK(hipEventRecord(done[0], A[0]));
        CK(hipStreamWaitEvent(A[2], head, 0));
        CK(hipStreamWaitEvent(A[2], done[1], 0));
        hipLaunchKernelGGL(spin, dim3(40), dim3(128), 24576, A[2], 30000ULL, sink + 3);
        hipLaunchKernelGGL(spin, dim3(40), dim3(128), 24576, A[2], 30000ULL, sink + 3);
        CK(hipEventRecord(done[2], A[2]));
        for (auto &e : done) CK(hipStreamWaitEvent(M, e, 0));
    };
    {
        unsigned *dev = reinterpret_cast<unsigned *>(sink + 6);
        const hipStream_t all[4] = {M, A[0], A[1], A[2]};
        for (int k = 0; k < 4; k++) { const unsigned h = hw_id_of(all[k], dev); g_pipes[k] = ((h >> 30) & 3) << 8 | ((h >> 6) & 3) << 4 | ((h >> 24) & 7); }
    }
    for (int k = 0; k < 5; k++) step();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        const double t0 = now_us();
        for (int k = 0; k < steps; k++) step();
        CK(hipStreamSynchronize(M));
        const double dt = (now_us() - t0) / steps;
        best = dt < best ? dt : best;
    }
    CK(hipDeviceSynchronize());
    if (M) CK(hipStreamDestroy(M));
    for (auto &a : A) CK(hipStreamDestroy(a));
    CK(hipEventDestroy(fork)); CK(hipEventDestroy(head));
    for (auto &e : done) CK(hipEventDestroy(e));
    return best;
}

int main(int argc, char **argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 50;
    unsigned long long *sink;
    CK(hipMalloc(&sink, 64));
    const unsigned DT = hipEventDisableTiming;
    const Cfg cfgs[] = {
        {0, 1, DT, "main NULL stream, side streams highest priority"},
        {0, 0, DT, "main NULL stream, side streams normal priority"},
        {1, 1, DT, "main created (normal), side streams highest priority"},
        {1, 0, DT, "main created (normal), side streams normal priority"},
        {2, 1, DT, "main created (highest), side streams highest priority"},
        {3, 1, DT, "main created BLOCKING (normal), side streams highest priority"},
        {1, 1, DT | hipEventReleaseToDevice, "main created (normal), side highest, events release-to-device"},
        {1, 1, DT | hipEventDisableSystemFence, "main created (normal), side highest, events without system fence"},
        {0, 1, DT | hipEventDisableSystemFence, "main NULL stream, side highest, events without system fence"},
        {0, 1, DT, "main NULL stream, side streams highest priority (again)"},
    };
    printf("{\"long_grid_us_nominal\": 1000, \"steps\": %d, \"runs\": [\n", steps);
    bool first = true;
    for (const Cfg &c : cfgs) {
        const double us = run(c, steps, sink);
        printf("%s  {\"config\": \"%s\", \"us_per_step\": %.1f, \"me.pipe.queue\": {\"main\": \"%x\", \"rows\": \"%x\", \"producers\": \"%x\", \"links\": \"%x\"}}", first ? "" : ",\n", c.name, us,
               g_pipes[0], g_pipes[1], g_pipes[2], g_pipes[3]);
        fflush(stdout);
        first = false;
    }
    printf("\n]}\n");
    return 0;
}
