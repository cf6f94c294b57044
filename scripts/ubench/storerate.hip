// Per-wave store issue rate: W waves per workgroup, one workgroup per CU-ish (G workgroups), each wave writes `iters` batches of
// 4 store instructions of 64 lanes x 16 B, LPS lanes per series (LPS x 16 contiguous bytes per series, series `pitch` doubles apart),
// without waiting for anything.  Prints cycles (s_memtime) per store instruction and GB/s.   usage: storerate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));
template <int LPS, bool NT>
__global__ void k(double *out, long pitch, long rows_per_series, int iters, unsigned long long *ticks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int SPI = 64 / LPS;                                  // series per instruction
    const long s0 = ((long)blockIdx.x * nw + wave) * 64;           // this wave's 64 series
    double *base = out + (s0 + lane / LPS) * pitch + (lane % LPS) * 2;
    const d2v v = {1.0 + lane, 2.0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const long t = ((long)it * LPS * 2) % (rows_per_series - LPS * 2);
#pragma unroll
        for (int i = 0; i < 64 / SPI; i++) {
            d2v *p = reinterpret_cast<d2v *>(base + (long)i * SPI * pitch + t);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { atomicAdd(&ticks[0], t1 - t0); atomicAdd(&ticks[1], t2 - t1); }
}
template <int LPS, bool NT> void run(double *out, long pitch, long T, int G, int W, unsigned long long *d_ticks) {
    const int iters = 2000;
    CK(hipMemset(d_ticks, 0, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<LPS, NT>), dim3(G), dim3(64 * W), 0, 0, out, pitch, T, 10, d_ticks); // warm
    CK(hipMemset(d_ticks, 0, 16));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<LPS, NT>), dim3(G), dim3(64 * W), 0, 0, out, pitch, T, iters, d_ticks);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, d_ticks, 16, hipMemcpyDeviceToHost));
    const double n_inst = (double)iters * (64 / (64 / LPS)), waves = (double)G * W;
    printf("pieces %4d B  %s  G=%4d W=%d : %7.1f ticks/store issue, drain %8.0f ticks, %7.1f GB/s total, %6.2f GB/s per wave\n", LPS * 16, NT ? "nt   " : "plain", G, W,
           (double)h[0] / waves / n_inst, (double)h[1] / waves, waves * n_inst * 1024.0 / ms / 1e6, n_inst * 1024.0 / ms / 1e6);
}
int main() {
    const long T = 2528, N = 64L * 4096;
    double *out; CK(hipMalloc(&out, N * T * 8));
    unsigned long long *d_ticks; CK(hipMalloc(&d_ticks, 16));
    for (int W : {1, 2, 4}) for (int G : {79, 256, 1024}) {
        if ((long)G * W * 64 > N) continue;
        run<4, true>(out, T, T, G, W, d_ticks);
        run<8, true>(out, T, T, G, W, d_ticks);
        run<64, true>(out, T, T, G, W, d_ticks);
        run<4, false>(out, T, T, G, W, d_ticks);
    }
    return 0;
}
