// pipes.hip -- which compute pipe / hardware queue slot does the runtime give a stream?  Streams are created one after the other and kept
// alive; a one-thread kernel on each reads HW_ID (hwreg 4): bits 7:6 pipe, 26:24 queue slot, 31:30 micro-engine.  Printed as me.pipe.queue.
// build: hipcc -O2 --offload-arch=gfx950 pipes.hip -o pipes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void probe(unsigned *out) { *out = __builtin_amdgcn_s_getreg((31 << 11) | 4); }
static unsigned *dev;
static void show(const char *what, hipStream_t st) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, st, dev);
    CK(hipStreamSynchronize(st));
    unsigned h = 0;
    CK(hipMemcpy(&h, dev, 4, hipMemcpyDeviceToHost));
    printf("  %-28s %u.%u.%u\n", what, (h >> 30) & 3, (h >> 6) & 3, (h >> 24) & 7);
}
int main() {
    CK(hipMalloc(&dev, 4));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("priority range: least %d, greatest %d\n", lo, hi);
    show("null stream", nullptr);
    hipStream_t h[8], n[8], l[4];
    char name[64];
    for (int k = 0; k < 8; k++) { CK(hipStreamCreateWithPriority(&h[k], hipStreamNonBlocking, hi)); snprintf(name, sizeof name, "high %d (first launch)", k); show(name, h[k]); }
    for (int k = 0; k < 8; k++) { snprintf(name, sizeof name, "high %d (again)", k); show(name, h[k]); }
    for (int k = 0; k < 8; k++) { CK(hipStreamCreateWithPriority(&n[k], hipStreamNonBlocking, 0)); snprintf(name, sizeof name, "normal %d (first launch)", k); show(name, n[k]); }
    for (int k = 0; k < 4; k++) { CK(hipStreamCreateWithPriority(&l[k], hipStreamNonBlocking, lo)); snprintf(name, sizeof name, "low %d (first launch)", k); show(name, l[k]); }
    show("null stream (again)", nullptr);
    for (int k = 0; k < 8; k++) CK(hipStreamDestroy(h[k]));
    hipStream_t again[4];
    for (int k = 0; k < 4; k++) { CK(hipStreamCreateWithPriority(&again[k], hipStreamNonBlocking, hi)); snprintf(name, sizeof name, "high, after destroy %d", k); show(name, again[k]); }
    return 0;
}
