// Microbenchmark: how does a CU's throughput scale with resident waves for SALU-heavy / VALU-f64-heavy dependent loops?
// grid = 256*WPC single-wave workgroups; each wave runs ITER iterations of S dependent SALU adds + V dependent f64 FMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int S, int V>
__global__ __launch_bounds__(64) void mix(double *out, int iters) {
    int s = blockIdx.x;
    double x = threadIdx.x, a = 1.0000001, b = 0.5;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < S; k++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");
#pragma unroll
        for (int k = 0; k < V; k++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    }
    if (x == 12345.0 || s == -1) out[0] = x + s;
}
template <int S, int V>
void run(double *out, int wpc) {
    const int iters = 20000;
    CK(hipGetLastError());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix<S, V>), dim3(256 * wpc), dim3(64), 0, 0, out, iters);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix<S, V>), dim3(256 * wpc), dim3(64), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double cyc = ms * 1e-3 * 2.4e9 / iters;
    printf("S=%3d V=%3d waves/CU=%2d  %7.3f ms  %7.1f cycles/iter/wave  -> CU retires %.2f SALU + %.2f VALU per cycle\n", S, V, wpc, ms, cyc,
           S * wpc / cyc, V * wpc / cyc);
}
int main() {
    double *out; CK(hipMalloc(&out, 8));
    for (int w : {1, 2, 4, 8, 16, 32}) run<64, 0>(out, w);
    for (int w : {1, 2, 4, 8, 16, 32}) run<0, 32>(out, w);
    for (int w : {1, 2, 4, 8, 16, 32}) run<48, 32>(out, w);
    for (int w : {4, 8, 16}) run<32, 8>(out, w);
    return 0;
}
