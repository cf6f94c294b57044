// mixbw.hip -- what this chip sustains for the STEP's traffic in its ideal shape: every "row block" reads R input streams and writes W
// output streams (distinct arrays), 16 bytes per lane and access, grid-stride, non-temporal stores.  The suite step moves 4.0 GB of reads
// and 11.1 GB of writes (27 : 73) to ~190 distinct columns; a plain copy (1 : 1, two arrays) moves 4.8-5.0 TB/s.
// hipcc -O3 --offload-arch=gfx950 mixbw.hip -o mixbw && ./mixbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int R, int W>
__global__ __launch_bounds__(256) void k(const d2 *__restrict__ src, d2 *__restrict__ dst, size_t n_col) {
    // column c of src / dst starts at c * n_col
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (; i < n_col; i += step) {
        d2 acc = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < R; r++) acc += src[(size_t)r * n_col + i];
#pragma unroll
        for (int w = 0; w < W; w++) { d2 v = acc; v.x += (double)w; __builtin_nontemporal_store(v, &dst[(size_t)w * n_col + i]); }
    }
}
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int R, int W>
int run(const d2 *a, d2 *b, size_t n_col, int blocks) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<R, W>), dim3(blocks), dim3(256), 0, 0, a, b, n_col);
    CHK(hipEventRecord(e0, 0));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k<R, W>), dim3(blocks), dim3(256), 0, 0, a, b, n_col);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    const double moved = (double)(R + W) * (double)n_col * 16.0;
    printf("%6d blocks  %2d read + %2d written streams of %4.0f MB  %7.3f ms  %7.1f GB/s (reads %2.0f %%)\n", blocks, R, W, n_col * 16.0 / 1e6, ms, moved / ms * 1e-6,
           100.0 * R / (R + W));
    return 0;
}
int main() {
    const size_t n_col = (size_t)100800000 / 16; // one 5000 x 2520 f64 column
    d2 *a, *b;
    CHK(hipMalloc(&a, 8 * n_col * 16)); CHK(hipMalloc(&b, 24 * n_col * 16));
    CHK(hipMemset(a, 1, 8 * n_col * 16)); CHK(hipMemset(b, 0, 24 * n_col * 16));
    for (int blocks : {4096, 16384}) {
        if (run<1, 1>(a, b, n_col, blocks)) return 1;
        if (run<1, 3>(a, b, n_col, blocks)) return 1;
        if (run<2, 6>(a, b, n_col, blocks)) return 1;
        if (run<4, 11>(a, b, n_col, blocks)) return 1;
        if (run<5, 16>(a, b, n_col, blocks)) return 1;
        if (run<8, 24>(a, b, n_col, blocks)) return 1;
        if (run<0, 8>(a, b, n_col, blocks)) return 1;
        if (run<8, 1>(a, b, n_col, blocks)) return 1;
    }
    return 0;
}
