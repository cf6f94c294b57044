// How many streams run concurrently?  N streams x one 1 ms spin kernel each.  (GPU_MAX_HW_QUEUES defaults to 4.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(unsigned long long ticks, int *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (ticks == 0) *sink = 1;
}
int main() {
    int *sink; CK(hipMalloc(&sink, 4));
    hipStream_t st[16]; hipEvent_t done[16];
    for (int i = 0; i < 16; i++) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); }
    hipStream_t main_st; CK(hipStreamCreate(&main_st));
    hipEvent_t e0, e1, fork; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int rep = 0; rep < 2; rep++)
    for (int n : {1, 2, 3, 4, 5, 6, 8, 12, 16}) {
        CK(hipEventRecord(e0, main_st));
        CK(hipEventRecord(fork, main_st));
        for (int i = 0; i < n; i++) {
            CK(hipStreamWaitEvent(st[i], fork, 0));
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], 100000ULL, sink);
            CK(hipEventRecord(done[i], st[i]));
            CK(hipStreamWaitEvent(main_st, done[i], 0));
        }
        CK(hipEventRecord(e1, main_st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("%2d streams x one 1 ms kernel: %.3f ms\n", n, ms);
    }
    return 0;
}
