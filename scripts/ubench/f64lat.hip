// f64lat.hip -- dependent-issue latency / issue rate of f64 VALU ops on gfx950, and the s_memtime tick against wall_clock64 (100 MHz).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off f64lat.hip -o f64lat
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CHAINS, bool FMASUB>
__global__ void k(double *out, unsigned long long *t, int n, double a, double x) {
    double e[CHAINS];
    for (int c = 0; c < CHAINS; c++) e[c] = (double)threadIdx.x + c;
    unsigned long long m0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            double d;
            if (FMASUB) asm("v_fma_f64 %0, %1, -1.0, %2" : "=v"(d) : "v"(e[c]), "v"(x)); // x - e as an fma (same rounding)
            else d = x - e[c];           // the EMA step: sub, then fma
            e[c] = fma(a, d, e[c]);
        }
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += e[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = m1 - m0; t[1] = w1 - w0; }
}
template <int CHAINS, bool FMASUB = false>
void run(int waves_per_block) {
    double *out; unsigned long long *t, h[2];
    (void)hipMalloc(&out, 8 * 1024 * 64); (void)hipMalloc(&t, 16);
    const int n = 20000;
    k<CHAINS, FMASUB><<<1, 64 * waves_per_block>>>(out, t, n, 0.1, 3.0);
    (void)hipDeviceSynchronize();
    k<CHAINS, FMASUB><<<1, 64 * waves_per_block>>>(out, t, n, 0.1, 3.0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%s chains=%d waves/block=%d: %.2f memtime ticks per (sub+fma) pair-step, %.3f wallclock(100MHz) ticks; memtime/wall = %.2f\n", FMASUB ? "fma-sub" : "add-sub", CHAINS, waves_per_block,
           (double)h[0] / n, (double)h[1] / n, (double)h[0] / (double)h[1]);
}
int main() {
    run<1>(1); run<2>(1); run<3>(1); run<4>(1); run<8>(1);
    run<1>(4); run<1>(8); run<2>(8); run<1>(16);
    run<1, true>(1); run<2, true>(1); run<4, true>(1); run<8, true>(1); run<2, true>(8);
    return 0;
}
