// Microbenchmark: how fast can MI355X copy a symbol-major [N][T] f64 matrix when every wavefront moves
// [64 rows][RUN bytes] tiles (row stride T*8 bytes)?  This is the memory access pattern of the SEQ tile body.
// Variants: RUN bytes per row per tile, tiles in flight per wave (DEPTH), waves per CU via grid size.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int RUN, int DEPTH>
__global__ __launch_bounds__(64) void tilecopy(const double *in, double *out, long N, long T, int ncopies) {
    constexpr int CPL = RUN / 16;          // lanes per row segment
    constexpr int SPI = 64 / CPL;          // rows per wave-wide access
    constexpr int NI = 64 / SPI;           // accesses per tile
    constexpr int KR = RUN / 8;            // rows(time) per tile
    const int lane = threadIdx.x;
    const long tile_s0 = (long)blockIdx.x * 64;
    const long copy = blockIdx.y;
    const double *src = in;                          // every copy job reads the same input (like the suite)
    double *dst = out + copy * N * T;
    const int csym = lane / CPL, cchunk = lane % CPL;
    long crow[NI];
    for (int i = 0; i < NI; i++) { long cs = tile_s0 + i * SPI + csym; crow[i] = (cs < N ? cs : N - 1) * T + cchunk * 2; }
    const long nt = T / KR;
    double2 buf[DEPTH][NI];
    for (int f = 0; f < DEPTH; f++)
        if (f < nt) for (int i = 0; i < NI; i++) buf[f][i] = *reinterpret_cast<const double2 *>(src + crow[i] + (long)f * KR);
    for (long it = 0; it < nt; it += DEPTH) {
#pragma unroll
        for (int f = 0; f < DEPTH; f++) {
            if (it + f < nt) {
                double2 v[NI];
#pragma unroll
                for (int i = 0; i < NI; i++) v[i] = buf[f][i];
                if (it + f + DEPTH < nt)
#pragma unroll
                    for (int i = 0; i < NI; i++) buf[f][i] = *reinterpret_cast<const double2 *>(src + crow[i] + (it + f + DEPTH) * KR);
#pragma unroll
                for (int i = 0; i < NI; i++)
                    if (tile_s0 + i * SPI + csym < N) *reinterpret_cast<double2 *>(dst + crow[i] + (it + f) * KR) = v[i];
            }
        }
    }
}

template <int RUN, int DEPTH>
void run(const double *in, double *out, long N, long T, int ncopies) {
    dim3 grid((N + 63) / 64, ncopies);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((tilecopy<RUN, DEPTH>), grid, dim3(64), 0, 0, in, out, N, T, ncopies);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((tilecopy<RUN, DEPTH>), grid, dim3(64), 0, 0, in, out, N, T, ncopies);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double wbytes = (double)ncopies * N * T * 8;
    printf("RUN=%4d B  depth=%d  copies=%2d  %7.3f ms   write %6.1f GB/s  (read+write %6.1f GB/s)\n", RUN, DEPTH, ncopies, ms,
           wbytes / ms / 1e6, 2 * wbytes / ms / 1e6);
}

__global__ void plaincopy(const double2 *in, double2 *out, long n2, long per_copy) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) out[i] = in[i % per_copy];
}
int main(int argc, char **argv) {
    const long N = 5000, T = argc > 1 ? atol(argv[1]) : 2520; const int MAXC = 32;
    printf("row stride %ld elements = %ld bytes\n", T, T * 8);
    double *in, *out;
    CK(hipMalloc(&in, N * T * 8)); CK(hipMalloc(&out, (size_t)MAXC * N * T * 8));
    CK(hipMemset(in, 0, N * T * 8));
    {
        long per = N * T / 2, n2 = per * 32;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(plaincopy, dim3(256 * 16), dim3(256), 0, 0, (const double2 *)in, (double2 *)out, n2, per);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(plaincopy, dim3(256 * 16), dim3(256), 0, 0, (const double2 *)in, (double2 *)out, n2, per);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("plain streaming copy x32: %7.3f ms  write %6.1f GB/s\n", ms, n2 * 16.0 / ms / 1e6);
    }
    if (argc > 2) { // same tile code on a matrix whose rows are 64 B and contiguous: a purely sequential stream
        long N2 = N * T / 8, T2 = 8;
        run<64, 1>(in, out, N2, T2, 32);
        run<64, 4>(in, out, N2, T2, 32);
        return 0;
    }
    for (int c : {32}) {
        run<64, 1>(in, out, N, T, c);  run<128, 1>(in, out, N, T, c); run<256, 1>(in, out, N, T, c); run<512, 1>(in, out, N, T, c);
        run<64, 4>(in, out, N, T, c);  run<128, 4>(in, out, N, T, c); run<256, 2>(in, out, N, T, c); run<128, 8>(in, out, N, T, c);
    }
    return 0;
}
