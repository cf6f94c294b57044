// Input-tile loading for the SEQ bodies, two ways.  A: what run_seq_lds does -- one tile of register prefetch (global_load_dwordx4,
// lane = (series l/4, chunk l%4)), ds_write into rows of 72 bytes, per-lane row reads.  B: direct-to-LDS loads
// (global_load_lds_dwordx4), lane l = (series l%16, chunk l/16) so that the 1 KB an instruction deposits lane-linearly holds chunk c of
// series s at slot s%16 + 16c; DEPTH tiles in flight; per-lane row reads at the swizzled slot.  Both sum every value they read (checked).
// NCOL input columns per workgroup, K = 8 rows per tile.   usage: ldsdirect
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
constexpr int K = 8, ROWB = 72;
template <int NCOL>
__global__ __launch_bounds__(64) void tile_a(const double *in, long N, long T, long pitch, double *sums, unsigned long long *ticks) {
    __shared__ __align__(16) unsigned char lds[NCOL * 64 * ROWB];
    const int lane = threadIdx.x;
    const long s0 = (long)blockIdx.x * 64;
    const int csym = lane / 4, cchunk = lane % 4;
    const long nt = T / K;
    double2 pre[NCOL][4];
    auto prefetch = [&](long t0) {
        for (int k = 0; k < NCOL; k++)
            for (int i = 0; i < 4; i++) pre[k][i] = *reinterpret_cast<const double2 *>(in + (long)k * N * pitch + (s0 + i * 16 + csym) * pitch + t0 + cchunk * 2);
    };
    prefetch(0);
    double acc = 0.0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (long it = 0; it < nt; it++) {
        for (int k = 0; k < NCOL; k++)
            for (int i = 0; i < 4; i++) {
                double *q = reinterpret_cast<double *>(lds + k * 64 * ROWB + (i * 16 + csym) * ROWB + cchunk * 16);
                q[0] = pre[k][i].x; q[1] = pre[k][i].y;
            }
        if (it + 1 < nt) prefetch((it + 1) * K);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int j = 0; j < K; j++)
            for (int k = 0; k < NCOL; k++) acc += *reinterpret_cast<const double *>(lds + k * 64 * ROWB + lane * ROWB + j * 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    sums[s0 + lane] = acc;
    if (lane == 0) atomicAdd(ticks, c1 - c0);
}
template <int NCOL, int DEPTH>
__global__ __launch_bounds__(64) void tile_b(const double *in, long N, long T, long pitch, double *sums, unsigned long long *ticks) {
    __shared__ __align__(16) unsigned char lds[DEPTH][NCOL][4][1024];
    const int lane = threadIdx.x;
    const long s0 = (long)blockIdx.x * 64;
    const int lsym = lane % 16, lchunk = lane / 16;
    const long nt = T / K;
    auto issue = [&](long it) { // NCOL * 4 instructions, each 1 KB lane-linear in LDS
        const int slot = (int)(it % DEPTH);
        for (int k = 0; k < NCOL; k++)
            for (int i = 0; i < 4; i++)
                __builtin_amdgcn_global_load_lds((glb_void *)(in + (long)k * N * pitch + (s0 + i * 16 + lsym) * pitch + it * K + lchunk * 2),
                                                 (lds_void *)&lds[slot][k][i][0], 16, 0, 0);
    };
    for (int f = 0; f < DEPTH - 1 && f < nt; f++) issue(f);
    double acc = 0.0;
    const int blk = lane / 16, rs = lane % 16; // this lane's series: block of 16, slot base
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (long it = 0; it < nt; it++) {
        if (it + DEPTH - 1 < nt) issue(it + DEPTH - 1);
        // wait until tile `it` has landed: at most (DEPTH-1) tiles = (DEPTH-1)*NCOL*4 loads may stay in flight
        if (it + DEPTH - 1 < nt) {
            if (DEPTH == 2) { if (NCOL == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (NCOL == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
            else { if (NCOL == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (NCOL == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int slot = (int)(it % DEPTH);
        for (int j = 0; j < K; j++)
            for (int k = 0; k < NCOL; k++) acc += *reinterpret_cast<const double *>(&lds[slot][k][blk][(rs + 16 * (j / 2)) * 16 + (j % 2) * 8]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    sums[s0 + lane] = acc;
    if (lane == 0) atomicAdd(ticks, c1 - c0);
}
int main() {
    const long N = 64 * 1027, T = 2520, pitch = 2528; // 1027 workgroups, as many as the suite holds at t = 0
    constexpr int NC = 3;
    double *in, *sums; unsigned long long *ticks;
    CK(hipMalloc(&in, (size_t)NC * N * pitch * 8)); CK(hipMalloc(&sums, N * 8)); CK(hipMalloc(&ticks, 8));
    double *h = (double *)malloc((size_t)NC * N * pitch * 8);
    for (long i = 0; i < (long)NC * N * pitch; i++) h[i] = (double)((i * 2654435761u) % 1000) * 0.001;
    CK(hipMemcpy(in, h, (size_t)NC * N * pitch * 8, hipMemcpyHostToDevice));
    double *ref = (double *)calloc(N, 8);
    auto check = [&](int ncol, const char *what) {
        double *g = (double *)malloc(N * 8); CK(hipMemcpy(g, sums, N * 8, hipMemcpyDeviceToHost));
        long bad = 0;
        for (long s = 0; s < N; s += 977) {
            double e = 0.0;
            for (long t = 0; t < T / K * K; t += K) for (int j = 0; j < K; j++) for (int k = 0; k < ncol; k++) e += h[(long)k * N * pitch + s * pitch + t + j];
            if (e != g[s]) bad++;
        }
        printf("%s: %s\n", what, bad ? "WRONG SUMS" : "sums ok"); free(g);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(LABEL, KERNEL, NCOL)                                                                                              \
    for (int rep = 0; rep < 2; rep++) {                                                                                       \
        CK(hipMemset(ticks, 0, 8)); CK(hipEventRecord(e0));                                                                   \
        hipLaunchKernelGGL(KERNEL, dim3((unsigned)(N / 64)), dim3(64), 0, 0, in, N, T, pitch, sums, ticks);                   \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                                  \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); unsigned long long tk; CK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost)); \
        if (rep) { printf("%-34s %7.3f ms  %7.0f ticks/tile  %6.2f TB/s read  ", LABEL, ms, (double)tk / (N / 64) / (T / K), (double)NCOL * N * T * 8 / ms / 1e9); check(NCOL, ""); } \
    }
    RUN("A regs+ds_write, 1 col", (tile_a<1>), 1) RUN("B lds-direct depth2, 1 col", (tile_b<1, 2>), 1) RUN("B lds-direct depth3, 1 col", (tile_b<1, 3>), 1)
    RUN("A regs+ds_write, 3 col", (tile_a<3>), 3) RUN("B lds-direct depth2, 3 col", (tile_b<3, 2>), 3) RUN("B lds-direct depth3, 3 col", (tile_b<3, 3>), 3)
    return 0;
}
