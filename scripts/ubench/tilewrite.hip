// Microbenchmark: pure writes of [64 rows][RUN bytes] tiles into symbol-major [N][T] f64 matrices (row pitch T*8 B).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int RUN>
__global__ __launch_bounds__(64) void tilewrite(double *out, long N, long T, int rowmode) {
    constexpr int CPL = RUN / 16, SPI = 64 / CPL, NI = 64 / SPI, KR = RUN / 8;
    const int lane = threadIdx.x;
    const long ntile = (N + 63) / 64;
    double *dst = out + (long)blockIdx.y * N * T;
    const int csym = lane / CPL, cchunk = lane % CPL;
    long crow[NI];
    for (int i = 0; i < NI; i++) {
        long r = i * SPI + csym;                                // row within the tile, 0..63
        long cs = rowmode == 0 ? (long)blockIdx.x * 64 + r      // 64 consecutive rows per wave
                               : r * ntile + blockIdx.x;        // rows interleaved across waves
        crow[i] = (cs < N ? cs : N - 1) * T + cchunk * 2;
    }
    const long nt = T / KR;
    double2 v = make_double2((double)lane, 1.0);
    for (long it = 0; it < nt; it++)
#pragma unroll
        for (int i = 0; i < NI; i++) *reinterpret_cast<double2 *>(dst + crow[i] + it * KR) = v;
}
__global__ void plainwrite(double2 *out, long n2) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    double2 v = make_double2(1.0, 2.0);
    for (; i < n2; i += stride) out[i] = v;
}
template <int RUN>
void run(double *out, long N, long T, int c, int rowmode) {
    dim3 grid((N + 63) / 64, c);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((tilewrite<RUN>), grid, dim3(64), 0, 0, out, N, T, rowmode);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((tilewrite<RUN>), grid, dim3(64), 0, 0, out, N, T, rowmode);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("RUN=%4d B rowmode=%d copies=%2d  %7.3f ms  write %6.1f GB/s\n", RUN, rowmode, c, ms, (double)c * N * T * 8 / ms / 1e6);
}
int main(int argc, char **argv) {
    const long N = 5000, T = argc > 1 ? atol(argv[1]) : 2528; const int MAXC = 32;
    double *out; CK(hipMalloc(&out, (size_t)MAXC * N * T * 8));
    {
        long n2 = (long)MAXC * N * T / 2;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(plainwrite, dim3(256 * 16), dim3(256), 0, 0, (double2 *)out, n2);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(plainwrite, dim3(256 * 16), dim3(256), 0, 0, (double2 *)out, n2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("plain streaming write: %7.3f ms  %6.1f GB/s\n", ms, n2 * 16.0 / ms / 1e6);
    }
    for (int rm : {0, 1}) {
        run<64>(out, N, T, 32, rm); run<128>(out, N, T, 32, rm); run<256>(out, N, T, 32, rm); run<512>(out, N, T, 32, rm);
    }
    run<64>(out, N, T, 8, 0); run<128>(out, N, T, 8, 0);
    return 0;
}
