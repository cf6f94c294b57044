// Microbenchmark: does it help the write path when the storer wave keeps ACC consecutive tiles in registers and issues
// their stores back to back (ACC x RUN contiguous bytes per series within a few cycles) instead of RUN bytes per tile period?
// 2-wave workgroups as in pq_dev.h (loader / storer), NT stores, plain loads.  usage: tilecopy3 [T]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));
template <int RUN, int ACC>
__global__ __launch_bounds__(128) void tilecopy3(const double *in, double *out, long N, long T) {
    constexpr int CPL = RUN / 16, SPI = 64 / CPL, NI = 64 / SPI, KR = RUN / 8;
    __shared__ d2v lds[2][NI][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tile_s0 = (long)blockIdx.x * 64;
    double *dst = out + (long)blockIdx.y * N * T;
    const int csym = lane / CPL, cchunk = lane % CPL;
    long crow[NI];
    for (int i = 0; i < NI; i++) { long cs = tile_s0 + i * SPI + csym; crow[i] = (cs < N ? cs : N - 1) * T + cchunk * 2; }
    const long nt = T / KR;
    if (wave == 0) {
        d2v buf[NI];
        for (int i = 0; i < NI; i++) buf[i] = *reinterpret_cast<const d2v *>(in + crow[i]);
        for (long it = 0; it < nt; it++) {
            const int slot = (int)(it & 1);
#pragma unroll
            for (int i = 0; i < NI; i++) lds[slot][i][lane] = buf[i];
            if (it + 1 < nt)
#pragma unroll
                for (int i = 0; i < NI; i++) buf[i] = *reinterpret_cast<const d2v *>(in + crow[i] + (it + 1) * KR);
            __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0) only
            __builtin_amdgcn_s_barrier();
        }
    } else {
        d2v v[ACC][NI];
        for (long it = 0; it < nt; it += ACC) {
#pragma unroll
            for (int a = 0; a < ACC; a++) {
                if (it + a < nt) {
                    __builtin_amdgcn_s_barrier();
                    const int slot = (int)((it + a) & 1);
#pragma unroll
                    for (int i = 0; i < NI; i++) v[a][i] = lds[slot][i][lane];
                }
            }
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int a = 0; a < ACC; a++)
                    if (it + a < nt && tile_s0 + i * SPI + csym < N)
                        __builtin_nontemporal_store(v[a][i], reinterpret_cast<d2v *>(dst + crow[i] + (it + a) * KR));
        }
    }
}
template <int RUN, int ACC>
void run(const double *in, double *out, long N, long T, int c) {
    dim3 grid((N + 63) / 64, c);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((tilecopy3<RUN, ACC>), grid, dim3(128), 0, 0, in, out, N, T);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((tilecopy3<RUN, ACC>), grid, dim3(128), 0, 0, in, out, N, T);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double wbytes = (double)c * N * T * 8;
    printf("RUN=%4d B  ACC=%d  copies=%2d  %7.3f ms  write %6.1f GB/s\n", RUN, ACC, c, ms, wbytes / ms / 1e6);
}
int main(int argc, char **argv) {
    const long N = 5000, T = argc > 1 ? atol(argv[1]) : 2520; const int MAXC = 64;
    double *in, *out;
    CK(hipMalloc(&in, N * T * 8)); CK(hipMalloc(&out, (size_t)MAXC * N * T * 8));
    CK(hipMemset(in, 0, N * T * 8));
    for (int c : {32, 64}) {
        run<64, 1>(in, out, N, T, c); run<64, 2>(in, out, N, T, c); run<64, 4>(in, out, N, T, c);
        run<128, 1>(in, out, N, T, c); run<128, 2>(in, out, N, T, c); run<128, 4>(in, out, N, T, c);
        run<256, 1>(in, out, N, T, c); run<256, 2>(in, out, N, T, c);
    }
    return 0;
}
