// wave_util.h -- one-wavefront-per-symbol plumbing shared by the backtest (ops_backtest_wave.h) and the indicators (wt_dev.h): cross-lane
// moves by DPP, the wave prefix-scan step list, the LDS geometry of a symbol's rows ([64 chunks][odd pitch]) and its coalesced staging.
#pragma once
#include "pq_dev.h"

constexpr int BTW_MAX_C = 128;           // rows per lane chunk: up to two 64-bit signal masks per lane (BtwBits<2>) => len <= 8192
__device__ __forceinline__ unsigned long long btw_ballot(bool x) { return __builtin_amdgcn_ballot_w64(x); }
__device__ __forceinline__ unsigned long long btw_readlane(unsigned long long v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double btw_readlane(double v, int l) {
    return __longlong_as_double((long long)btw_readlane((unsigned long long)__double_as_longlong(v), l));
}
__device__ __forceinline__ unsigned long long btw_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
// One wave per workgroup: its LDS operations execute in program order, so lanes exchange data through LDS without a barrier;
// the compiler only has to keep the accesses in order (__syncthreads would also wait for every outstanding global store).
__device__ __forceinline__ void btw_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }


// ---- cross-lane moves by DPP (gfx9: row shifts, row broadcasts and the whole-wave shift) instead of ds_bpermute: a prefix scan over
// the 64 lanes is six steps -- row_shr 1 / 2 / 4 / 8 inside the rows of 16, then lane 15 of a row into the next row and lane 31 into
// the upper half -- of ~10 clocks each, against ~250 for a __shfl_up of a double (two bpermutes and a wait per value).  A lane
// without a source keeps `old`, which every caller sets to its operator's identity.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int btw_dpp(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double btw_dpp(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int BTW_ROW_SHR1 = 0x111, BTW_ROW_SHR2 = 0x112, BTW_ROW_SHR4 = 0x114, BTW_ROW_SHR8 = 0x118, BTW_BCAST15 = 0x142,
              BTW_BCAST31 = 0x143, BTW_WAVE_SHR1 = 0x138;
// inclusive prefix scan over the lanes: v <- op(values of the lower lanes (already combined), v); `id` = identity of op
#define BTW_SCAN_STEPS(STEP)                                                                                 \
    STEP(BTW_ROW_SHR1, 0xf) STEP(BTW_ROW_SHR2, 0xf) STEP(BTW_ROW_SHR4, 0xf) STEP(BTW_ROW_SHR8, 0xf) STEP(BTW_BCAST15, 0xa) STEP(BTW_BCAST31, 0xc)
// lane l <- lane l - 1 (lane 0 keeps `old`)
template <class T>
__device__ __forceinline__ T btw_prev_lane(T old, T v) { return btw_dpp<BTW_WAVE_SHR1>(old, v); }


// geometry of one symbol's rows in LDS: row i of the series at i + (i / C) * (P - C) (chunk pitch P odd: lane c reads its chunk's
// rows c * P + b without bank conflicts, and the wave reads 64 consecutive rows without them too)
struct BtwGeom {
    int T, C, P;
    unsigned magic; // ceil(2^20 / C)
    __device__ __forceinline__ int addr(int i) const { return i + (int)(((unsigned)i * magic) >> 20) * (P - C); }
};
// one column of the symbol, coalesced, into LDS (rows >= T: `fill`); returns whether this lane saw a NULL row.  `part` of `nparts`: the
// waves of a workgroup that share one symbol take the access batches in turn
__device__ __forceinline__ bool btw_stage(const BtwGeom &g, int lane, const double *src, double *dst, double fill, int part = 0, int nparts = 1) {
    const int T = g.T, C = g.C;
    bool null_seen = false;
    auto addr = [&](int i) { return g.addr(i); };
    {
        if (((reinterpret_cast<uintptr_t>(src) & 15) == 0)) { // 16 bytes per lane: rows 128 * j + 2 * lane, + 1
            const int npair = (64 * C + 127) / 128; // every LDS row below 64 * C is written (a ragged batch sizes C for its LONGEST group: rows in [T, 64 * C) get `fill`)
            for (int j0 = 0; j0 * nparts < npair; j0 += 8) {
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = (j0 + u) * nparts + part;
                    const int i = 128 * j + 2 * lane;
                    v[u] = make_double2(fill, fill);
                    if (j < npair) {
                        if (i + 1 < T) v[u] = *reinterpret_cast<const double2 *>(src + i);
                        else if (i < T) v[u].x = src[i];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = (j0 + u) * nparts + part;
                    const int i = 128 * j + 2 * lane;
                    if (j < npair && i < 64 * C) {
                        dst[addr(i)] = v[u].x;
                        dst[addr(i + 1)] = v[u].y;
                        null_seen |= (i < T && pq_isnull(v[u].x)) || (i + 1 < T && pq_isnull(v[u].y));
                    }
                }
            }
        } else {
            for (int j0 = 0; j0 * nparts < C; j0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = (j0 + u) * nparts + part;
                    const int i = 64 * j + lane;
                    v[u] = (j < C && i < T) ? src[i] : fill;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = (j0 + u) * nparts + part;
                    const int i = 64 * j + lane;
                    if (j < C) {
                        dst[addr(i)] = v[u];
                        null_seen |= i < T && pq_isnull(v[u]);
                    }
                }
            }
        }
    }
    return null_seen;
}

