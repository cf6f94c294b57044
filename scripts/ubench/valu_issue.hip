// valu_issue.hip -- issue cost (cycles per wave instruction, 8 independent streams: no dependency stalls) and dependent latency (1 stream)
// of the VALU operations the f64 kernels are made of, gfx950.  build: hipcc -O3 --offload-arch=gfx950 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OPS(X) \
    X(add_f64, "v_add_f64 %0, %0, %1", double, "+v", "v") \
    X(mul_f64, "v_mul_f64 %0, %0, %1", double, "+v", "v") \
    X(fma_f64_vop3, "v_fma_f64 %0, %0, %1, %1", double, "+v", "v") \
    X(fmac_f64, "v_fmac_f64_e32 %0, %1, %1", double, "+v", "v") \
    X(max_f64, "v_max_f64 %0, %0, %1", double, "+v", "v") \
    X(min_f64, "v_min_f64 %0, %0, %1", double, "+v", "v") \
    X(floor_f64, "v_floor_f64_e32 %0, %0", double, "+v", "v") \
    X(rcp_f64, "v_rcp_f64_e32 %0, %0", double, "+v", "v") \
    X(sqrt_f64, "v_sqrt_f64_e32 %0, %0", double, "+v", "v") \
    X(mov_b64, "v_mov_b64_e32 %0, %1", double, "+v", "v") \
    X(add_f32, "v_add_f32_e32 %0, %0, %1", float, "+v", "v") \
    X(fma_f32, "v_fma_f32 %0, %0, %1, %1", float, "+v", "v") \
    X(cndmask_b32, "v_cndmask_b32_e32 %0, %0, %1, vcc", float, "+v", "v") \
    X(lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1", double, "+v", "v") \
    X(and_b32, "v_and_b32_e32 %0, %0, %1", float, "+v", "v")
template <int CHAINS, class T, class F>
__device__ void run(F f, T *out, unsigned long long *t, int slot, int n, T seed) {
    T e[CHAINS];
    for (int c = 0; c < CHAINS; c++) e[c] = seed + (T)(threadIdx.x + c);
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) f(e[c]);
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    T s = 0;
    for (int c = 0; c < CHAINS; c++) s += e[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) t[slot] = m1 - m0;
}
__global__ void k(double *o, float *of, unsigned long long *t, int n, double y, float yf) {
    int slot = 0;
#define X(NAME, ASM, TY, C0, C1) \
    run<8, TY>([&](TY &e) { asm volatile(ASM : C0(e) : C1((TY)(sizeof(TY) == 8 ? y : yf))); }, (TY *)(sizeof(TY) == 8 ? (void *)o : (void *)of), t, slot++, n, (TY)1.5); \
    run<1, TY>([&](TY &e) { asm volatile(ASM : C0(e) : C1((TY)(sizeof(TY) == 8 ? y : yf))); }, (TY *)(sizeof(TY) == 8 ? (void *)o : (void *)of), t, slot++, n, (TY)1.5);
    OPS(X)
#undef X
    // compares (write vcc; no register dependency between them: issue cost only)
    run<8, double>([&](double &e) { asm volatile("v_cmp_gt_f64_e32 vcc, %0, %1" : "+v"(e) : "v"(y) : "vcc"); }, o, t, slot++, n, 1.5);
    run<1, double>([&](double &e) { asm volatile("v_cmp_gt_f64_e32 vcc, %0, %1" : "+v"(e) : "v"(y) : "vcc"); }, o, t, slot++, n, 1.5);
    run<8, double>([&](double &e) { asm volatile("v_cmp_eq_u64_e32 vcc, %0, %1" : "+v"(e) : "v"(y) : "vcc"); }, o, t, slot++, n, 1.5);
    run<1, double>([&](double &e) { asm volatile("v_cmp_eq_u64_e32 vcc, %0, %1" : "+v"(e) : "v"(y) : "vcc"); }, o, t, slot++, n, 1.5);
}
int main() {
    double *o; float *of; unsigned long long *t, h[64];
    (void)hipMalloc(&o, 8 * 64); (void)hipMalloc(&of, 4 * 64); (void)hipMalloc(&t, 8 * 64);
    const int n = 4000;
    for (int rep = 0; rep < 2; rep++) { k<<<1, 64>>>(o, of, t, n, 1.0000001, 1.0000001f); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, t, 8 * 64, hipMemcpyDeviceToHost);
    const char *names[] = {
#define X(NAME, ASM, TY, C0, C1) #NAME,
        OPS(X)
#undef X
        "cmp_gt_f64", "cmp_eq_u64"};
    printf("%-22s %12s %12s   (shader clocks; s_memtime = 2.4 GHz)\n", "op", "issue/instr", "dependent");
    for (int i = 0; i < (int)(sizeof names / sizeof *names); i++)
        printf("%-22s %12.2f %12.2f\n", names[i], (double)h[2 * i] / (8.0 * n), (double)h[2 * i + 1] / n);
    return 0;
}
