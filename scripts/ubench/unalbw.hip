// unalbw.hip -- do 16-byte global loads / stores work, and at what rate, on addresses that are only 8-byte aligned?  (the KFD sets
// SH_MEM_CONFIG.ALIGNMENT_MODE = unaligned for compute queues; the tiled bodies' UNAL form moves 8 bytes per lane because the
// compiler, told the truth about the alignment, splits the access.)
// hipcc -O3 --offload-arch=gfx950 unalbw.hip -o unalbw && ./unalbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void copy16(const double *src, double *dst, size_t n_pairs) { // src / dst possibly 8 mod 16
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (; i < n_pairs; i += step) {
        d2 v = *reinterpret_cast<const d2 *>(src + 2 * i);
        __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(dst + 2 * i));
    }
}
__global__ __launch_bounds__(256) void copy8(const double *src, double *dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) __builtin_nontemporal_store(src[i], dst + i);
}
// the tiled bodies' shape: a wave-wide access covers 8 series x 128 bytes (8 lanes x 16 B per series), the series 2521 * 8 bytes apart
__global__ __launch_bounds__(64) void tile16(const double *src, double *dst, int64_t pitch, int64_t T, int64_t n_series) {
    const int lane = threadIdx.x, sy = lane >> 3, ch = lane & 7;
    for (int64_t s0 = (int64_t)blockIdx.x * 8; s0 < n_series; s0 += (int64_t)gridDim.x * 8) {
        const int64_t s = s0 + sy;
        if (s >= n_series) continue;
        for (int64_t t0 = 0; t0 + 16 <= T; t0 += 16) {
            const int64_t o = s * pitch + t0 + 2 * ch;
            d2 v = *reinterpret_cast<const d2 *>(src + o);
            __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(dst + o));
        }
    }
}
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t n = (size_t)1 << 27; // 1 GiB of doubles
    double *a, *b;
    CHK(hipMalloc(&a, (n + 16) * 8)); CHK(hipMalloc(&b, (n + 16) * 8));
    CHK(hipMemset(a, 1, (n + 16) * 8)); CHK(hipMemset(b, 0, (n + 16) * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int off = 0; off < 2; off++) {
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, a + off, b + off, n / 2);
        CHK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, a + off, b + off, n / 2);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("16-byte copy, base %s: %.3f ms  %.1f GB/s read + the same written\n", off ? "8 mod 16" : "16-byte aligned", ms, n * 8.0 / ms * 1e-6);
    }
    {
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(copy8, dim3(8192), dim3(256), 0, 0, a + 1, b + 1, n);
        CHK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(copy8, dim3(8192), dim3(256), 0, 0, a + 1, b + 1, n);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("8-byte copy: %.3f ms  %.1f GB/s read + the same written\n", ms, n * 8.0 / ms * 1e-6);
    }
    // correctness of the misaligned 16-byte form + the tile shape at pitch 2521 / 2528
    {
        const int64_t NS = 5000, T = 2520;
        for (int64_t pitch : {2528, 2521}) {
            double *h = (double *)malloc(NS * pitch * 8);
            for (int64_t i = 0; i < NS * pitch; i++) h[i] = (double)i * 0.5;
            CHK(hipMemcpy(a, h, NS * pitch * 8, hipMemcpyHostToDevice)); CHK(hipMemset(b, 0, NS * pitch * 8));
            for (int w = 0; w < 2; w++) hipLaunchKernelGGL(tile16, dim3(2048), dim3(64), 0, 0, a, b, pitch, T, NS);
            CHK(hipEventRecord(e0, 0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(tile16, dim3(2048), dim3(64), 0, 0, a, b, pitch, T, NS);
            CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            double *g = (double *)malloc(NS * pitch * 8);
            CHK(hipMemcpy(g, b, NS * pitch * 8, hipMemcpyDeviceToHost));
            long bad = 0;
            for (int64_t s = 0; s < NS; s++) for (int64_t t = 0; t < 2512; t++) bad += g[s * pitch + t] != h[s * pitch + t];
            printf("tile copy 5000 x 2520, pitch %ld: %.3f ms  %.1f GB/s each way, %ld wrong values\n", (long)pitch, ms, NS * 2512 * 8.0 / ms * 1e-6, bad);
            free(h); free(g);
        }
    }
    return 0;
}
