// wt_dev.h -- indicators in the ONE-SYMBOL-PER-WAVEFRONT, time-parallel form (round 4): the primitives.
//
// The lane-per-symbol bodies (pq_dev.h) walk 2 520 serial rows per lane; their loads and stores are 64 / 128-byte pieces of 64
// different series and every store passes through one storer wave.  Here a workgroup is ONE wavefront and owns ONE symbol: its
// columns sit in LDS ([64 chunks][P] doubles, chunk pitch P odd: conflict-free both for "lane c walks chunk c" and for "the wave
// touches 64 consecutive rows"), every global access is 1 KB of one series per wave instruction, and lane c owns the rows
// [c*C, (c+1)*C), C = ceil(T / 64).  An indicator is a composition of four primitives over LDS columns:
//
//   wt_stage*  columns of the symbol (or row-wise functions of them: +DM / -DM / TR, up / down moves) -> LDS, coalesced;
//   wt_chain   a scalar CONTRACTIVE recurrence along the column -- e <- fma(alpha, x - e, e) (calc_ema, overlap.rs:660-730) or
//              r <- (r*(p-1) + x) / p (calc_rma, D-1) -- bit for bit the serial walk's values, in (nW + 3) chunk walks instead
//              of 64 (below);
//   wt_map     row-wise arithmetic between columns (exact in any order);
//   wt_store   a column (or a row-wise function of columns) -> global memory, 1 KB contiguous per instruction, non-temporal.
//
// wt_chain, the time-parallel recurrence.  The first output row rs = r0 + p - 1 lies in chunk hc: that lane computes the seed
// exactly as the reference does (p ordered adds, one division) and the rest of its chunk.  Every other lane needs the state in
// front of its own chunk:
//   * in exact arithmetic one row is the affine map e -> q e + g x, a chunk is the composition of its rows' maps, and affine maps
//     compose associatively: one Horner pass per lane and one DPP prefix scan over the lanes give the value at every chunk
//     boundary to ~1e-15 -- not the bits (every step of the true recurrence rounds);
//   * lane c starts nW chunks early from that approximate value and runs the TRUE recurrence into its own chunk: the map is a
//     contraction, the trajectory approaches the serial one geometrically and then merges with it BITWISE (two trajectories that
//     hold the same double are identical from then on);
//   * proof instead of hope: a lane's state at its first own row is compared AS RAW BITS with its predecessor's state after its
//     last row.  The lanes hc+1 .. hc+nW+1 start from the anchor's exact state and are exact by construction, so exactness
//     propagates lane by lane; a chunk that fails the test is re-run from its predecessor's end state and counted.  A flat
//     column (a recurrence that does not contract) costs time, never correctness.
// Cascades (DEMA / TEMA / TRIX, the MACD signal line, ADX over DX) are chains over columns that earlier chains have made exact.
//
// Scope: null-free symbols, len <= 4 096; regular batches from 1 024 rows on, ragged batches (pq_batch.offsets: one wavefront per
// group) whose groups average 1 024 rows.  A symbol with a NULL / NaN input sets its 64-symbol tile's flag in `gate`, and the
// lane-per-symbol kernel of the same function -- launched behind this one, gated by those flags -- redoes that tile (it exits at once
// when no flag is set).  Nothing here is a second definition of an indicator's semantics for nulls.  Which functions use the form by
// default, and why a recorded suite does not: wt.hip, ops_wt.h (wt_try), DESIGN.md section 3c.
#pragma once
#include "pq_cores.h"
#include "wave_util.h"

constexpr int WT_MIN_LEN = 1024; // below: chunks shorter than 16 rows, the serial head dominates -- the lane-per-symbol body is the better shape
constexpr int WT_MAX_LEN = 4096; // 64 chunks of at most 64 rows
enum { WT_EMA = 0, WT_RMA = 1 };

struct WtArgs {
    int32_t C, P;              // rows per lane chunk, chunk pitch in LDS (odd)
    uint32_t magic;            // ceil(2^20 / C): i / C == (i * magic) >> 20 for i < 64 * C
    unsigned *gate;            // [tiles of 64 symbols]: != 0 -> the gated lane-per-symbol launch redoes the tile
    unsigned long long *stats; // nullable: [0] symbols done here, [1] chunks that failed the bit test, [2] chunk re-runs, [3] symbols sent to the gate
    double warm;               // warm-up rows = warm / alpha (a speed knob only: an unmerged chunk is re-run)
};

#ifdef PQ_WT_PROF // scripts/ab_build.sh only: device time per phase summed over the waves, stats[8 + k] in shader clocks
#define WT_T(w, k) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if ((w).lane == 0 && (w).prof) atomicAdd((w).prof + (k), t__ - (w).t_prev); (w).t_prev = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WT_T(w, k)
#endif
struct WtCtx {
    BtwGeom g;
    int lane;
    int64_t base;              // first row of this symbol in the batch's columns
    double *lds;
    double warm;
    unsigned long long nfail, nrerun;
    unsigned long long *prof, t_prev;
    __device__ __forceinline__ double *col(int k) const { return lds + (size_t)k * 64 * g.P; }
    __device__ __forceinline__ int nW(double alpha) const { // warm-up chunks of a chain with gain alpha
        double rows = warm / alpha;
        if (!(rows < 4096.0)) rows = 4096.0;
        int n = ((int)rows + g.C - 1) / g.C;
        return n < 1 ? 1 : (n > 62 ? 62 : n);
    }
};

template <int KIND>
struct WtRec {
    double a;       // EMA: alpha = 2 / (p + 1)
    double pm1, pf; // RMA: p - 1, p
    __device__ __forceinline__ double step(double e, double x) const {
        if constexpr (KIND == WT_EMA) return fma(a, x - e, e); // overlap.rs:698 alpha.mul_add(x - ema, ema)
        else return (e * pm1 + x) / pf;                        // D-1 calc_rma
    }
    __device__ __forceinline__ double q() const { if constexpr (KIND == WT_EMA) return 1.0 - a; else return pm1 / pf; }
    __device__ __forceinline__ double gain() const { if constexpr (KIND == WT_EMA) return a; else return 1.0 / pf; }
};
__device__ __forceinline__ WtRec<WT_EMA> wt_ema(int p) { return WtRec<WT_EMA>{pq_uniform(2.0 / ((double)p + 1.0)), 0.0, 0.0}; }
__device__ __forceinline__ WtRec<WT_RMA> wt_rma(int p) { return WtRec<WT_RMA>{0.0, pq_uniform((double)p - 1.0), pq_uniform((double)p)}; }

struct WtInId { __device__ __forceinline__ double operator()(double v) const { return v; } };
struct WtInZ0 { __device__ __forceinline__ double operator()(double v) const { return pq_isnull(v) ? 0.0 : v; } }; // None -> 0.0 (Q-MACD, Q-TRIX)
struct WtOutId { __device__ __forceinline__ double operator()(int, double e) const { return e; } };

__device__ __forceinline__ bool wt_bad(double v) { return v != v; } // NULL (a NaN bit pattern) or any other NaN: the general path decides

// One raw column of the symbol into LDS (16-byte loads when the series starts aligned); rows in [T, 64 C) become 0.0.
// Returns whether this lane saw a NULL / NaN.
__device__ __forceinline__ bool wt_stage(const WtCtx &w, const double *gcol, double *dst) {
    bool bad = false;
    const int T = w.g.T, C = w.g.C, lane = w.lane;
    const double *src = gcol + w.base;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        // every load of the column in flight before the first LDS write: ONE memory round trip per column (C <= 64: <= 32 pairs)
        const int npair = (64 * C + 127) / 128;
        for (int j0 = 0; j0 < npair; j0 += 16) {
            double2 v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int i = 128 * (j0 + u) + 2 * lane;
                v[u] = make_double2(0.0, 0.0);
                if (i + 1 < T) v[u] = *reinterpret_cast<const double2 *>(src + i);
                else if (i < T) v[u].x = src[i];
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int i = 128 * (j0 + u) + 2 * lane;
                if (i < 64 * C) {
                    dst[w.g.addr(i)] = v[u].x;
                    dst[w.g.addr(i + 1)] = v[u].y;
                    bad |= wt_bad(v[u].x) | wt_bad(v[u].y);
                }
            }
        }
    } else {
        for (int j0 = 0; j0 < C; j0 += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int i = 64 * (j0 + u) + lane;
                v[u] = (j0 + u < C && i < T) ? src[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int i = 64 * (j0 + u) + lane;
                if (j0 + u < C) { dst[w.g.addr(i)] = v[u]; bad |= wt_bad(v[u]); }
            }
        }
    }
    WT_T(const_cast<WtCtx &>(w), 0);
    return bad;
}
// NOUT row-wise functions of global columns into LDS columns: f(i, double (&v)[NOUT]) for 0 <= i < T (it reads global memory: 512
// contiguous bytes per column and instruction; a lag-1 operand comes from L1).  Rows in [T, 64 C) become 0.0.
template <int NOUT, class F>
__device__ __forceinline__ void wt_stage_fn(const WtCtx &w, double *const (&dst)[NOUT], F &&f) {
    const int T = w.g.T, C = w.g.C, lane = w.lane;
    for (int j0 = 0; j0 < C; j0 += 8) {
        double v[8][NOUT];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = 64 * (j0 + u) + lane;
#pragma unroll
            for (int k = 0; k < NOUT; k++) v[u][k] = 0.0;
            if (j0 + u < C && i < T) f(i, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = 64 * (j0 + u) + lane;
            if (j0 + u < C) {
                const int a = w.g.addr(i);
#pragma unroll
                for (int k = 0; k < NOUT; k++) dst[k][a] = v[u][k];
            }
        }
    }
    WT_T(const_cast<WtCtx &>(w), 7);
}
// f(i, a) for every row i < T (a = its LDS address), 64 consecutive rows per step; LDS reads of a step precede its writes
template <class F>
__device__ __forceinline__ void wt_map(const WtCtx &w, F &&f) {
    const int T = w.g.T, C = w.g.C;
    for (int j = 0; j < C; j++) {
        const int i = 64 * j + w.lane;
        if (i < T) f(i, w.g.addr(i));
    }
    WT_T(const_cast<WtCtx &>(w), 9);
}
// gcol[base + i] = f(i, a) for i < T: 1 KB contiguous per instruction (16 bytes per lane) when the series starts 16-byte aligned
template <class F>
__device__ __forceinline__ void wt_store(const WtCtx &w, double *gcol, F &&f) {
    if (!gcol) return;
    double *dst = gcol + w.base;
    const int T = w.g.T, lane = w.lane;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const int nb2 = (T + 127) / 128;
        int j = 0;
        for (; j + 4 <= nb2 && 128 * (j + 4) <= T; j += 4) { // whole steps: the LDS reads of four steps in flight, then four 1 KB stores
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i0 = 128 * (j + u) + 2 * lane; v[u] = make_double2(f(i0, w.g.addr(i0)), f(i0 + 1, w.g.addr(i0 + 1))); }
#pragma unroll
            for (int u = 0; u < 4; u++) nt_store2(dst + 128 * (j + u) + 2 * lane, v[u]);
        }
        for (; j < nb2; j++) {
            const int i0 = 128 * j + 2 * lane;
            if (i0 + 1 < T) nt_store2(dst + i0, make_double2(f(i0, w.g.addr(i0)), f(i0 + 1, w.g.addr(i0 + 1))));
            else if (i0 < T) __builtin_nontemporal_store(f(i0, w.g.addr(i0)), dst + i0);
        }
    } else {
        const int nb = (T + 63) / 64;
        for (int j = 0; j < nb; j++) {
            const int i = 64 * j + lane;
            if (i < T) __builtin_nontemporal_store(f(i, w.g.addr(i)), dst + i);
        }
    }
    WT_T(const_cast<WtCtx &>(w), 8);
}
// two columns from one pass over the rows (f writes both values)
template <class F>
__device__ __forceinline__ void wt_store2(const WtCtx &w, double *g0, double *g1, F &&f) {
    double *d0 = g0 ? g0 + w.base : nullptr, *d1 = g1 ? g1 + w.base : nullptr;
    const int T = w.g.T, lane = w.lane;
    const bool wide = (((d0 ? reinterpret_cast<uintptr_t>(d0) : 0) | (d1 ? reinterpret_cast<uintptr_t>(d1) : 0)) & 15) == 0;
    const int nb2 = (T + 127) / 128;
    int j = 0;
    if (wide)
        for (; j + 2 <= nb2 && 128 * (j + 2) <= T; j += 2) { // whole steps, two at a time
            double2 va[2], vb[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i0 = 128 * (j + u) + 2 * lane;
                f(i0, w.g.addr(i0), va[u].x, vb[u].x);
                f(i0 + 1, w.g.addr(i0 + 1), va[u].y, vb[u].y);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (d0) nt_store2(d0 + 128 * (j + u) + 2 * lane, va[u]);
                if (d1) nt_store2(d1 + 128 * (j + u) + 2 * lane, vb[u]);
            }
        }
    for (; j < nb2; j++) {
        const int i0 = 128 * j + 2 * lane;
        double a0 = 0, b0 = 0, a1 = 0, b1 = 0;
        if (i0 < T) f(i0, w.g.addr(i0), a0, b0);
        if (i0 + 1 < T) f(i0 + 1, w.g.addr(i0 + 1), a1, b1);
        if (wide && i0 + 1 < T) {
            if (d0) nt_store2(d0 + i0, make_double2(a0, a1));
            if (d1) nt_store2(d1 + i0, make_double2(b0, b1));
        } else {
            if (i0 < T) { if (d0) __builtin_nontemporal_store(a0, d0 + i0); if (d1) __builtin_nontemporal_store(b0, d1 + i0); }
            if (i0 + 1 < T) { if (d0) __builtin_nontemporal_store(a1, d0 + i0 + 1); if (d1) __builtin_nontemporal_store(b1, d1 + i0 + 1); }
        }
    }
    WT_T(const_cast<WtCtx &>(w), 8);
}

// K independent chains of one kind in ONE set of walks.  Chain k: dst[k][i] = out(k, i, e_i) for i >= rs = r0 + p - 1, NULL below, where
// e_rs = (in(src[r0]) + ... + in(src[rs])) / p with the adds in row order and e_i = rec.step(e_{i-1}, in(src[i])): the serial walk's
// values, bit for bit (header).  dst may be src (in place).  `in` maps a stored value to the recurrence's operand (identity, or None ->
// 0.0); `out` may read other LDS columns at row i.  The chains' dependency chains interleave (a dependent f64 step costs ~44 clocks on a
// lone wave, three interleaved ones ~56 together: scripts/ubench/f64lat.hip), which is what a wave that has a SIMD to itself needs.
// Aliasing rule: a chain that runs in place over ANOTHER chain's source must come after it in the list (the emit order relies on it).
template <int KIND>
struct WtChainSpec {
    const double *src;
    double *dst;
    WtRec<KIND> rec;
    int p, r0;
};
template <int K, int KIND, class InF, class OutF>
__device__ __forceinline__ void wt_chains(WtCtx &w, const WtChainSpec<KIND> (&ch)[K], InF in, OutF out) {
    const int T = w.g.T, C = w.g.C, P = w.g.P, c = w.lane;
    const int nlive = (T + C - 1) / C;
    int rs[K], hc[K], nW[K], q0[K], nWmax = 0;
    bool dead[K], act[K], spec[K];
    double e[K], e_seed[K], s_start[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        rs[k] = ch[k].r0 + ch[k].p - 1;
        dead[k] = ch[k].p <= 0 || rs[k] >= T;
        hc[k] = dead[k] ? 64 : (int)(((unsigned)rs[k] * w.g.magic) >> 20);
        nW[k] = w.nW(ch[k].rec.gain());
        q0[k] = (c - nW[k] > hc[k] + 1) ? c - nW[k] : hc[k] + 1;
        act[k] = !dead[k] && c > hc[k] && c < nlive;
        spec[k] = act[k] && q0[k] > hc[k] + 1;
        if (!dead[k] && nW[k] > nWmax) nWmax = nW[k];
        e[k] = 0.0; e_seed[k] = 0.0;
    }
    // 1. anchors
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (c == hc[k]) {
            const double *src = ch[k].src;
            double sum = 0.0;
            int i = ch[k].r0;
            for (; i + 8 <= rs[k] + 1; i += 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) x[u] = in(src[w.g.addr(i + u)]);
#pragma unroll
                for (int u = 0; u < 8; u++) sum += x[u];
            }
            for (; i <= rs[k]; i++) sum += in(src[w.g.addr(i)]);
            e_seed[k] = sum / (double)ch[k].p;
            double ee = e_seed[k];
            const double *row = src + hc[k] * P;
            int b = rs[k] - hc[k] * C + 1;
            for (; b + 8 <= C; b += 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) x[u] = in(row[b + u]);
#pragma unroll
                for (int u = 0; u < 8; u++) ee = ch[k].rec.step(ee, x[u]);
            }
            for (; b < C; b++) ee = ch[k].rec.step(ee, in(row[b]));
            e[k] = ee;
        }
    }
    WT_T(w, 1);
    // 2. affine maps of my chunk and their prefix compositions
    double A[K], B[K];
    {
        double q[K], gn[K];
#pragma unroll
        for (int k = 0; k < K; k++) { q[k] = ch[k].rec.q(); gn[k] = ch[k].rec.gain(); A[k] = 1.0; B[k] = 0.0; }
        int b = 0;
        for (; b + 8 <= C; b += 8) {
            double x[K][8];
#pragma unroll
            for (int k = 0; k < K; k++)
#pragma unroll
                for (int u = 0; u < 8; u++) x[k][u] = in(ch[k].src[c * P + b + u]);
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int k = 0; k < K; k++) { B[k] = fma(q[k], B[k], gn[k] * x[k][u]); A[k] *= q[k]; }
        }
        for (; b < C; b++)
#pragma unroll
            for (int k = 0; k < K; k++) { B[k] = fma(q[k], B[k], gn[k] * in(ch[k].src[c * P + b])); A[k] *= q[k]; }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (c < hc[k]) { A[k] = 1.0; B[k] = 0.0; }
        if (c == hc[k]) { A[k] = 0.0; B[k] = e[k]; }
    }
#define WT_AFFINE_K(CTRL, RM)                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < K; k++) {                                                                             \
        const double pA = btw_dpp<CTRL, RM>(1.0, A[k]), pB = btw_dpp<CTRL, RM>(0.0, B[k]);                                      \
        B[k] = fma(A[k], pB, B[k]); A[k] *= pA;                                                                                 \
    }
    BTW_SCAN_STEPS(WT_AFFINE_K)
#undef WT_AFFINE_K
    WT_T(w, 2);
#pragma unroll
    for (int k = 0; k < K; k++) {
        const double seed = __shfl(B[k], q0[k] - 1);
        if (act[k]) e[k] = seed;
    }
    // 3. warm-up: every chain walks its chunks q0 .. c-1 in step
    for (int kk = 0; kk < nWmax; kk++) {
        const int kq = __builtin_amdgcn_readfirstlane(kk);
        bool on[K], any = false;
        const double *row[K];
        double en[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            on[k] = act[k] && q0[k] + kq < c;
            any |= on[k];
            row[k] = ch[k].src + (on[k] ? q0[k] + kq : 0) * P;
            en[k] = e[k];
        }
        if (btw_ballot(any) == 0) break;
        int b = 0;
        for (; b + 8 <= C; b += 8) {
            double x[K][8];
#pragma unroll
            for (int k = 0; k < K; k++)
#pragma unroll
                for (int u = 0; u < 8; u++) x[k][u] = in(row[k][b + u]);
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int k = 0; k < K; k++) en[k] = ch[k].rec.step(en[k], x[k][u]);
        }
        for (; b < C; b++)
#pragma unroll
            for (int k = 0; k < K; k++) en[k] = ch[k].rec.step(en[k], in(row[k][b]));
#pragma unroll
        for (int k = 0; k < K; k++) e[k] = on[k] ? en[k] : e[k];
    }
    WT_T(w, 3);
    // 4. own chunks, state only
#pragma unroll
    for (int k = 0; k < K; k++) s_start[k] = e[k];
    {
        int b = 0;
        for (; b + 8 <= C; b += 8) {
            double x[K][8];
#pragma unroll
            for (int k = 0; k < K; k++)
#pragma unroll
                for (int u = 0; u < 8; u++) x[k][u] = in(ch[k].src[c * P + b + u]);
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int k = 0; k < K; k++) e[k] = ch[k].rec.step(e[k], x[k][u]);
        }
        for (; b < C; b++)
#pragma unroll
            for (int k = 0; k < K; k++) e[k] = ch[k].rec.step(e[k], in(ch[k].src[c * P + b]));
    }
    WT_T(w, 4);
    // 5. bit tests and re-runs, chain by chain (rare)
#pragma unroll
    for (int k = 0; k < K; k++) {
        auto mismatch = [&]() {
            const unsigned long long pe = btw_bits(btw_prev_lane(0.0, e[k]));
            return spec[k] && btw_bits(s_start[k]) != pe;
        };
        unsigned long long mism = btw_ballot(mismatch());
        w.nfail += (unsigned long long)__popcll(mism);
        while (mism) {
            const int cs = __builtin_ctzll(mism);
            mism &= mism - 1;
            const double pe = btw_readlane(e[k], cs - 1);
            if (c == cs) {
                s_start[k] = pe;
                double ee = pe;
                const double *row = ch[k].src + c * P;
                for (int b = 0; b < C; b++) ee = ch[k].rec.step(ee, in(row[b]));
                e[k] = ee;
            }
            w.nrerun++;
            const bool again = mismatch();
            mism |= btw_ballot(again && c == cs + 1);
        }
    }
    WT_T(w, 5);
    // 6. emit, chain by chain (a lane reads the rows of a batch before it writes them, so a chain may run in place -- also over the
    // source of an EARLIER chain of the list, whose emit is complete by then)
#pragma unroll
    for (int k = 0; k < K; k++) {
        const double *row = ch[k].src + c * P;
        double *drow = ch[k].dst + c * P;
        if (dead[k]) {
            if (c < nlive) for (int b = 0; b < C; b++) drow[b] = pq_null();
        } else if (c < hc[k]) {
            for (int b = 0; b < C; b++) drow[b] = pq_null();
        } else if (c == hc[k]) {
            const int bs = rs[k] - hc[k] * C;
            double ee = e_seed[k];
            for (int b0 = 0; b0 < C; b0 += 8) {
                double x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; u++) x[u] = b0 + u < C ? in(row[b0 + u]) : 0.0;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int b = b0 + u;
                    if (b > bs) ee = ch[k].rec.step(ee, x[u]);
                    y[u] = (b < bs) ? pq_null() : (b < C ? out(k, hc[k] * C + b, ee) : 0.0);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) if (b0 + u < C) drow[b0 + u] = y[u];
            }
        } else if (act[k]) {
            double ee = s_start[k];
            int b = 0;
            for (; b + 8 <= C; b += 8) {
                double x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; u++) x[u] = in(row[b + u]);
#pragma unroll
                for (int u = 0; u < 8; u++) { ee = ch[k].rec.step(ee, x[u]); y[u] = out(k, c * C + b + u, ee); }
#pragma unroll
                for (int u = 0; u < 8; u++) drow[b + u] = y[u];
            }
            for (; b < C; b++) { ee = ch[k].rec.step(ee, in(row[b])); drow[b] = out(k, c * C + b, ee); }
        }
    }
    btw_lds_fence();
    WT_T(w, 6);
}

// one chain: dst[i] = out(i, e_i)
template <int KIND, class InF, class OutF>
__device__ __forceinline__ void wt_chain(WtCtx &w, const double *src, double *dst, const WtRec<KIND> rec, int p, int r0, InF in, OutF out) {
    const WtChainSpec<KIND> ch[1] = {{src, dst, rec, p, r0}};
    wt_chains<1>(w, ch, in, [&](int, int i, double e) { return out(i, e); });
}

// The kernel: one wavefront = one symbol; Op::run(w) composes the primitives.  Op contract:
//   static constexpr int NCOL;                    LDS columns
//   __device__ bool run(WtCtx &w) const;          false: an input held a NULL / NaN (nothing was stored; the gate takes the tile)
template <class Op>
__global__ __launch_bounds__(64) void wt_kernel(Op op, WtArgs a, Dims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wt_lds_raw[];
    const int lane = (int)threadIdx.x;
    const int64_t s = blockIdx.x;
    const int T = (int)dims_len(d, s);
    if (T == 0) return;
    WtCtx w{BtwGeom{T, a.C, a.P, a.magic}, lane, dims_base(d, s), reinterpret_cast<double *>(wt_lds_raw), a.warm, 0ULL, 0ULL, a.stats ? a.stats + 8 : nullptr,
            __builtin_amdgcn_s_memtime()};
    const bool ok = op.run(w);
    if (lane == 0) {
        if (!ok) atomicOr(&a.gate[s >> 6], 1u);
        if (a.stats) {
            atomicAdd(a.stats + (ok ? 0 : 3), 1ULL);
            if (w.nfail) atomicAdd(a.stats + 1, w.nfail);
            if (w.nrerun) atomicAdd(a.stats + 2, w.nrerun);
        }
    }
}
