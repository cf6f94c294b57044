// Microbenchmark: tile copy with the loads and the stores issued by DIFFERENT wavefronts of a 2-wave workgroup.
// vmcnt is one in-order counter for loads but stores retire out of order, so a wave that has stores in flight cannot wait
// for a prefetched load without also waiting for its stores.  Here wave 0 only loads (deep prefetch, exact in-order waits)
// and hands tiles over through LDS; wave 1 only stores and never waits on vmcnt.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double d2v __attribute__((ext_vector_type(2)));
template <int RUN, int DEPTH, int NT = 0>
__global__ __launch_bounds__(128) void tilecopy2(const double *in, double *out, long N, long T) {
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
        d2v buf[DEPTH][NI];
        for (int f = 0; f < DEPTH; f++)
            if (f < nt) for (int i = 0; i < NI; i++) buf[f][i] = (NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const d2v *>(in + crow[i] + (long)f * KR)) : *reinterpret_cast<const d2v *>(in + crow[i] + (long)f * KR);
        for (long it = 0; it < nt; it += DEPTH) {
#pragma unroll
            for (int f = 0; f < DEPTH; f++) {
                if (it + f < nt) {
                    const int slot = (int)((it + f) & 1);
#pragma unroll
                    for (int i = 0; i < NI; i++) lds[slot][i][lane] = buf[f][i];
                    if (it + f + DEPTH < nt)
#pragma unroll
                        for (int i = 0; i < NI; i++) buf[f][i] = (NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const d2v *>(in + crow[i] + (it + f + DEPTH) * KR)) : *reinterpret_cast<const d2v *>(in + crow[i] + (it + f + DEPTH) * KR);
                    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0) only
                    __builtin_amdgcn_s_barrier();       // tile ready
                }
            }
        }
    } else {
        for (long it = 0; it < nt; it++) {
            __builtin_amdgcn_s_barrier();               // wait for tile `it`
            const int slot = (int)(it & 1);
            d2v v[NI];
#pragma unroll
            for (int i = 0; i < NI; i++) v[i] = lds[slot][i][lane];
#pragma unroll
            for (int i = 0; i < NI; i++)
                if (tile_s0 + i * SPI + csym < N) { if (NT & 2) __builtin_nontemporal_store(v[i], reinterpret_cast<d2v *>(dst + crow[i] + it * KR)); else *reinterpret_cast<d2v *>(dst + crow[i] + it * KR) = v[i]; }
        }
    }
}
template <int RUN, int DEPTH, int NT = 0>
void run(const double *in, double *out, long N, long T, int c) {
    dim3 grid((N + 63) / 64, c);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((tilecopy2<RUN, DEPTH, NT>), grid, dim3(128), 0, 0, in, out, N, T);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((tilecopy2<RUN, DEPTH, NT>), grid, dim3(128), 0, 0, in, out, N, T);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double wbytes = (double)c * N * T * 8;
    printf("2-wave NT=%d RUN=%4d B depth=%d copies=%2d  %7.3f ms  write %6.1f GB/s (read+write %6.1f)\n", NT, RUN, DEPTH, c, ms, wbytes / ms / 1e6, 2 * wbytes / ms / 1e6);
}
int main(int argc, char **argv) {
    const long N = 5000, T = argc > 1 ? atol(argv[1]) : 2560; const int MAXC = 104;
    double *in, *out;
    CK(hipMalloc(&in, N * T * 8)); CK(hipMalloc(&out, (size_t)MAXC * N * T * 8));
    CK(hipMemset(in, 0, N * T * 8));
    for (int c : {32}) {
        run<128, 1, 0>(in, out, N, T, c); run<128, 1, 1>(in, out, N, T, c); run<128, 1, 2>(in, out, N, T, c); run<128, 1, 3>(in, out, N, T, c);
        run<64, 1, 0>(in, out, N, T, c); run<64, 1, 1>(in, out, N, T, c); run<64, 1, 2>(in, out, N, T, c); run<64, 1, 3>(in, out, N, T, c);
        run<128, 2, 3>(in, out, N, T, c); run<256, 1, 3>(in, out, N, T, c);
    }
    return 0;
}
