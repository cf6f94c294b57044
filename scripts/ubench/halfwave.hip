// halfwave.hip -- does a wavefront with fewer ACTIVE lanes issue its f64 vector instructions faster on gfx950?
// (If the SIMD skipped the passes of switched-off lanes, a small shard could walk 32 or 16 series per wavefront at twice / four times the
// rate: the lane-per-series bodies are bound by their own instruction stream when a wave has its SIMD to itself.)
// One wave, 8 independent fma chains (issue-bound) and 1 chain (latency-bound), with 64 / 32 / 16 / 1 active lanes -- once by launching a
// smaller block, once by masking lanes of a full wave with a branch.  Prints s_memtime ticks per instruction.
// build: hipcc -O3 --offload-arch=gfx950 halfwave.hip -o halfwave
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int CH>
__device__ void chains(int n, double y, double *out, unsigned long long *t, int slot) {
    double e[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) e[c] = 1.5 + (double)(threadIdx.x + c);
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(e[c]) : "v"(y));
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += e[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) t[slot] = m1 - m0;
}
__global__ void k(int n, double y, int active, double *out, unsigned long long *t) {
    if ((int)threadIdx.x < active) { // (a full wave with lanes switched off, or -- launched with `active` threads -- every lane on)
        chains<8>(n, y, out, t, 0);
        chains<1>(n, y, out, t, 1);
    }
}
int main() {
    double *o; unsigned long long *t, h[2];
    (void)hipMalloc(&o, 8 * 64); (void)hipMalloc(&t, 16);
    const int n = 4000;
    printf("%-28s %14s %14s   (s_memtime ticks per v_fma_f64)\n", "wave", "8 chains", "1 chain");
    for (int masked = 0; masked < 2; masked++)
        for (int active : {64, 32, 16, 1}) {
            for (int rep = 0; rep < 2; rep++) { k<<<1, masked ? 64 : active>>>(n, 1.0000001, active, o, t); (void)hipDeviceSynchronize(); }
            (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            char name[64];
            snprintf(name, sizeof name, "%s, %d lanes active", masked ? "64-lane block" : "small block", active);
            printf("%-28s %14.2f %14.2f\n", name, (double)h[0] / (8.0 * n), (double)h[1] / n);
        }
    return 0;
}
