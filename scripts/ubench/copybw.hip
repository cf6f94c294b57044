// copybw.hip -- what a streaming copy reaches on this chip: grid-stride 16-byte copies, plain / non-temporal, and pure read / pure write.
// hipcc -O3 --offload-arch=gfx950 copybw.hip -o copybw && ./copybw     (bytes counted once per direction; GB/s = 1e9 B/s)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE> // 0 copy plain, 1 copy non-temporal stores, 2 read only (sum), 3 write only (nt)
__global__ __launch_bounds__(256) void k(const d2 *__restrict__ src, d2 *__restrict__ dst, size_t n, double *sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    d2 acc = {0.0, 0.0};
    for (; i < n; i += step) {
        if (MODE == 0) dst[i] = src[i];
        else if (MODE == 1) __builtin_nontemporal_store(src[i], &dst[i]);
        else if (MODE == 2) acc += src[i];
        else { d2 v = {1.0, 2.0}; __builtin_nontemporal_store(v, &dst[i]); }
    }
    if (MODE == 2 && acc.x + acc.y == 123.456) *sink = acc.x;
}
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    d2 *a, *b; double *sink;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMalloc(&sink, 8));
    CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char *names[4] = {"copy (plain stores)", "copy (non-temporal stores)", "read only", "write only (non-temporal)"};
    for (int blocks : {2048, 8192, 32768}) for (int mode = 0; mode < 4; mode++) {
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
            else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
        };
        for (int w = 0; w < 3; w++) launch();
        CHK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; r++) launch();
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        const double moved = (mode < 2 ? 2.0 : 1.0) * (double)bytes;
        printf("%6d blocks  %-28s %7.3f ms  %7.1f GB/s total (%s)\n", blocks, names[mode], ms, moved / ms * 1e-6, mode < 2 ? "read + write" : "one direction");
    }
    return 0;
}
