// suite_mj.hip -- multi-job workgroups: MJ_NC compute waves of DIFFERENT jobs on one 64-symbol tile + one storer wave that drains all
// of them (suite_jobs.h says why).  The compute side is run_seq_lds<Op, false, true> (pq_dev.h): the op walks exactly as in the
// two-wave form -- same loads, same LDS tiles, same arithmetic in the same order, so the columns are bit-identical -- and hands a
// finished out tile over by counting it in LDS instead of meeting its storer at a barrier.  The storer is op-agnostic: an out tile is
// NOUT x [64 series][K rows] doubles at a fixed pitch, whatever produced it (the one exception: the Hilbert job's derived columns).
#include "suite_jobs.h"

#if defined(PQ_EXPERIMENTS) && defined(PQ_EXP_MJ)


#ifndef PQ_MJ_NO_DERIVE
#define PQ_MJ_NO_DERIVE 0
#endif
namespace {

// one finished out tile of job `job` -> registers -> (counter: LDS is free again) -> global memory, 16 bytes per lane and access
template <int K, int NOUTMAX, bool HT = false>
__device__ __forceinline__ void mj_store_tile(const SeqJob &job, const unsigned char *lds_job, MjCtl *ctl, unsigned it, const Dims &d, int64_t tile_s0, int lane) {
    constexpr int ROWB = K * 8 + 8, TB = 64 * ROWB, CPL = K / 2, SPI = 64 / CPL, NI = K / 2;
    const int nout = job.nout;
    const int csym = lane / CPL, cchunk = lane % CPL;
    const int64_t tile_left = d.n - 1 - tile_s0;
    const unsigned rel_max = tile_left < 63 ? (tile_left > 0 ? (unsigned)tile_left : 0u) : 63u;
    const unsigned stride_b = (unsigned)d.stride * 8u;
    const unsigned lane_part = (unsigned)csym * stride_b + (unsigned)cchunk * 16u;
    const unsigned char *co_base = lds_job + csym * ROWB + cchunk * 16;
    double2 v[NOUTMAX][NI];
#pragma unroll
    for (int k = 0; k < NOUTMAX; k++)
        if (k < nout) {
#pragma unroll
            for (int i = 0; i < NI; i++) { // two b64 reads: LDS rows are only 8-byte aligned
                const double *q = reinterpret_cast<const double *>(co_base + i * (SPI * ROWB) + k * TB);
                v[k][i] = make_double2(q[0], q[1]);
            }
        }
    lds_fence();
    mj_post(&ctl->taken, it + 1); // the compute wave may overwrite the tile
    const int64_t t0 = (int64_t)it * K;
    const int64_t tile_base = tile_s0 * d.stride + t0;
#pragma unroll
    for (int k = 0; k < NOUTMAX; k++)
        if (k < nout) {
            unsigned char *const col = reinterpret_cast<unsigned char *>(job.out[k] + tile_base); // wave-uniform
#pragma unroll
            for (int i = 0; i < NI; i++)
                if ((unsigned)(csym + i * SPI) <= rel_max)
                    nt_store2(reinterpret_cast<double *>(col + (lane_part + (unsigned)(i * SPI) * stride_b)), v[k][i]);
        }
    if constexpr (HT && !PQ_MJ_NO_DERIVE) {
        static_assert(K == SeqTile<HtAll6Op>::K && NOUTMAX == 3, "the Hilbert job's tile shape");
        { // derived columns (NDer): dcphase / sine / leadsine from the phasor rows this lane holds
            HtAll6Op hop;
            __builtin_memcpy(&hop, job.op, sizeof(HtAll6Op)); // (only `der` is used: the rest of the copy is dead)
            double *const *der = hop.der;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                // one evaluation at a time (the scheduler would otherwise run the eight inlined atan of a tile side by side: ~60 registers
                // spilled under the 192 cap)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    double y[3], z[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) y[k] = h ? v[k][i].y : v[k][i].x;
                    HtAll6Op::derive(y, z);
#pragma unroll
                    for (int k = 0; k < 3; k++) { if (h) v[k][i].y = z[k]; else v[k][i].x = z[k]; }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if ((unsigned)(csym + i * SPI) <= rel_max) {
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        nt_store2(reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(der[k] + tile_base) + (lane_part + (unsigned)(i * SPI) * stride_b)), v[k][i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// NOT inlined: the 64-bit constants of derive()'s atan (~80 registers) would otherwise be hoisted out of the storer's loop and stay live
// across the eight held columns of the general path (measured: 50 registers spilled under the 192 cap)
__device__ __attribute__((noinline)) void mj_store_tile_ht(const SeqJob &job, const unsigned char *lds_job, MjCtl *ctl, unsigned it, const Dims &d,
                                                           int64_t tile_s0, int lane) {
    mj_store_tile<8, 3, true>(job, lds_job, ctl, it, d, tile_s0, lane);
}
} // namespace

__global__ __attribute__((amdgpu_num_vgpr(PQ_NV0))) __launch_bounds__(64 * (MJ_NC + 1), 2)
void seq_mj_kernel(const SeqJob *jobs, const MjGroup *groups, Dims d, unsigned *err, unsigned long long *dbg) {
    extern __shared__ __align__(16) unsigned char mj_lds[];
    const int64_t s0 = (int64_t)blockIdx.x * SEQ_BLOCK;
    if (s0 >= d.n) return; // grid.x is padded to a multiple of 8 (whole workgroup)
    const MjGroup &g = groups[blockIdx.y];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63); // (wave-uniform values in scalar registers)
    MjCtl *ctl = reinterpret_cast<MjCtl *>(mj_lds);
    if (threadIdx.x < MJ_NC * 4) reinterpret_cast<unsigned *>(mj_lds)[threadIdx.x] = 0u;
    __syncthreads(); // the only workgroup barrier: the counters are zero before anybody reads them
    if (wave < MJ_NC) { // ------------------------------------------------------------ compute wave of job `wave`
        if (wave >= g.njobs) return;
        const int jx = g.job[wave];
        const SeqJob &job = jobs[jx];
        if (dbg && lane == 0) atomicMin(&dbg[2 * jx], wall_clock64());
        if (job.prio == 3) __builtin_amdgcn_s_setprio(3);
        else if (job.prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (job.prio == 1) __builtin_amdgcn_s_setprio(1);
        unsigned char *lds_job = mj_lds + __builtin_amdgcn_readfirstlane((int)g.lds_off[wave]);
        switch (job.kind) { // wave-uniform
#define X(OP)                                                                                                        \
    case OP::SEQ_ID: {                                                                                               \
        if constexpr (MjOk<OP>::value) {                                                                             \
            OP op;                                                                                                   \
            __builtin_memcpy(&op, job.op, sizeof(OP));                                                               \
            run_seq_lds<OP, false, true>(op, job.in, job.out, d, s0, lds_job, ctl + wave, err);                      \
        }                                                                                                            \
    } break;
            SEQ_OPS_LIGHT(X)
#undef X
        default: break;
        }
        if (dbg && lane == 0) atomicMax(&dbg[2 * jx + 1], wall_clock64());
        return;
    }
    // ---------------------------------------------------------------------------------- storer: any job's finished tile
    // (`taken` of a job is this wave's own count of the tiles it has stored: no per-job state in registers, one loop body for any job)
    unsigned left = 0;
    for (int w = 0; w < g.njobs; w++) left += (unsigned)(d.len / jobs[g.job[w]].tile_k);
    unsigned idle = 0;
    while (left > 0) {
        bool any = false;
#pragma unroll 1
        for (int w = 0; w < g.njobs; w++) {
            const unsigned taken = mj_peek(&ctl[w].taken);
            if (mj_peek(&ctl[w].ready) <= taken) continue;
            const SeqJob &job = jobs[g.job[w]];
            const unsigned char *lds_job = mj_lds + g.lds_off[w];
            switch (job.tile_k) {
            case 16: mj_store_tile<16, 2>(job, lds_job, ctl + w, taken, d, s0, lane); break;
            case 8: // (the Hilbert job's derived columns -- an inlined atan -- get an instantiation of their own: three held columns, not eight)
                if (job.kind == HtAll6Op::SEQ_ID) mj_store_tile_ht(job, lds_job, ctl + w, taken, d, s0, lane);
                else mj_store_tile<8, 8>(job, lds_job, ctl + w, taken, d, s0, lane);
                break;
            case 4: mj_store_tile<4, 8>(job, lds_job, ctl + w, taken, d, s0, lane); break;
            default: break;
            }
            left--;
            any = true;
        }
        if (any) { idle = 0; continue; }
        __builtin_amdgcn_s_sleep(4);
        if (++idle > MJ_SPIN_LIMIT) { if (lane == 0 && err) atomicExch(err, 2u); return; }
    }
}

bool mj_kind_supported(int kind) {
    switch (kind) {
#define X(OP) case OP::SEQ_ID: return MjOk<OP>::value && (SeqTile<OP>::K == 16 ? OP::NOUT <= 2 : (SeqTile<OP>::K == 8 || SeqTile<OP>::K == 4)) && OP::NOUT <= 8;
        SEQ_OPS_LIGHT(X)
#undef X
    default: return false;
    }
}

pq_status mj_launch(pq_ctx *ctx, hipStream_t st, const SeqJob *d_jobs, const MjGroup *d_groups, int n_groups, unsigned tiles, unsigned lds_bytes,
                    Dims d, unsigned *d_err, unsigned long long *dbg) {
    (void)ctx;
    if (lds_bytes > 64 * 1024) // dynamic LDS above 64 KB is an opt-in per kernel and device
        PQ_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&seq_mj_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(seq_mj_kernel, dim3(tiles, (unsigned)n_groups), dim3(64 * (MJ_NC + 1)), lds_bytes, st, d_jobs, d_groups, d, d_err, dbg);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}

#else
// MEASURED AND NOT TAKEN (EXPERIMENTS.md, round 5: 5.68 against 4.17 ms per step, columns bit-identical): the product build carries neither
// the kernel nor its op instantiations; `scripts/ab_build.sh mj "-DPQ_EXPERIMENTS -DPQ_EXP_MJ" all` builds the variant, PQ_MJ=1 selects it.
bool mj_kind_supported(int) { return false; }
pq_status mj_launch(pq_ctx *, hipStream_t, const SeqJob *, const MjGroup *, int, unsigned, unsigned, Dims, unsigned *, unsigned long long *) {
    pq_set_error("multi-job workgroups are not part of this build (PQ_EXP_MJ)");
    return PQ_ERR_UNSUPPORTED;
}
#endif
