// Does hipExtAnyOrderLaunch let consecutive kernels of ONE stream run concurrently on gfx950?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(unsigned long long ticks, int *sink) { // wall_clock64: 100 MHz; bounded
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (ticks == 0) *sink = 1;
}
int main() {
    int *sink; CK(hipMalloc(&sink, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int flags : {0, 1}) {
        for (int n : {1, 4, 8}) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < n; i++)
                hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, flags, 100000ULL, sink); // 1 ms each
            CK(hipGetLastError());
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("flags=%d  %d x 1 ms spin kernels in one stream: %.3f ms\n", flags, n, ms);
        }
    }
    return 0;
}
