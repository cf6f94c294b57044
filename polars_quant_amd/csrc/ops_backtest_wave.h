// ops_backtest_wave.h -- the per-symbol backtest in its ONE-SYMBOL-PER-WAVEFRONT, row-parallel form (round 3).
//
// Reference: src/backtest/vectorized.rs:124-194 (scan), src/backtest/metrics.rs:7-152 (summary), momentum.rs:250-283 + D-8
// (MACD-cross signals).  The lane-per-symbol form (ops_backtest.h) walks T rows of a strict recurrence serially and costs the
// same whether the chip holds 5 000 symbols or 625; this form breaks the walk into three parts whose serial length is not T:
//
//  (A) SIGNALS by speculative T-chunks.  Lane c owns the rows [c*C, (c+1)*C) (C = ceil(T / 64)).  It starts the MACD state
//      machine nW chunks EARLY, as if the series began there (SMA seeds and all), and runs into its own chunk.  The EMA
//      recurrence fma(alpha, x - e, e) is a contraction: the speculative trajectory approaches the true one geometrically and
//      then MERGES BITWISE with it (once two trajectories hold the same double they are identical forever: the step is a
//      deterministic function of state and input).  Proof per chunk instead of hope: lane c's state at its first own row is
//      compared AS RAW BITS with lane c-1's state after its last row; equal => every value lane c produced is the serial
//      one.  Lane 0 starts at row 0 and is exact by construction, so exactness propagates lane by lane.  A chunk that fails the
//      test is re-run from its predecessor's end state (C rows, counted in `stats`); flat prices (a trajectory that never
//      contracts) therefore cost time, never correctness.
//  (B) STATE WALK over events only.  (pos, cash, entry_cost) change only on rows where a buy finds the pool flat or a sell
//      finds it long (vectorized.rs:146-175): the wave jumps from event to event with ballot / find-first-set over the
//      per-lane signal masks -- ~100 events per symbol instead of 2 520 rows -- and does the event's arithmetic once,
//      wave-uniform, in the reference's operation order.
//  (C) ROW-PARALLEL FILL.  64 consecutive rows per wave instruction: position / cash come from the block's event table,
//      equity = cash + pos * price (the same two roundings), stores are 512 contiguous bytes per column and instruction.
//      No LDS transpose of outputs, no storer wave.  The summary's running max is an exact prefix max, max_drawdown /
//      win_rate / total_trades are exact; the ordered f64 sums of calculate_summary (mean, variance, covariance) are summed
//      per chunk and then across lanes in a fixed order: <= 1e-12 relative (they are tolerance columns of the parity bar).
#pragma once
#include <type_traits>
#include "ops_backtest.h"
#include "wave_util.h"

constexpr int BTW_NG = 8;                // register groups of 64 events each: the dense form holds <= 512 events per symbol

struct BtWaveArgs {
    const double *price;
    const uint8_t *buy, *sell; // nullptr: MACD-cross signals are generated in-kernel
    const double *bench;       // nullable, [n][stride]
    double *position, *cash, *equity, *summary; // each nullable
    pq_bt_params prm;
    int32_t fast, slow, sig;
    int32_t C, P, nW, nW2;     // chunk rows, chunk pitch in LDS (odd: conflict-free per-lane reads), warm-up chunks (arbitrary / scanned seeds)
    int32_t kcap;              // capacity of the event lists in LDS (more events: the block form)
    uint32_t magic;            // ceil(2^20 / C): i / C == (i * magic) >> 20 for i < 64 * C
    unsigned long long *stats; // nullable: [0] symbols, [1] speculative chunks that failed the bit test, [2] chunk re-runs
};

// The MACD-cross state machine of one lane (BtMacdOp::step without the trade part)
struct BtwMacd {
    EmaCore ef, es, eg;
    double prev_m, prev_s;
    __device__ void init(int64_t fast, int64_t slow, int64_t sig, int64_t T) {
        ef.init(fast, T); es.init(slow, T); eg.init(sig, T);
        prev_m = pq_null(); prev_s = pq_null();
    }
    __device__ bool steady() const { return ef.steady() && es.steady() && eg.steady() && !pq_isnull(prev_m) && !pq_isnull(prev_s); }
    // general row (nulls, warm-up): momentum.rs:250-283 + D-8
    __device__ __forceinline__ void step(int i, double px, bool &buy, bool &sell) {
        const double f = ef.step(px), sl = es.step(px);
        const double m = (!pq_isnull(f) && !pq_isnull(sl)) ? f - sl : pq_null();
        const double g = eg.step(z0b(m));
        const bool ok = i > 0 && !pq_isnull(m) && !pq_isnull(g) && !pq_isnull(prev_m) && !pq_isnull(prev_s);
        buy = ok && (prev_m <= prev_s) && (m > g);
        sell = ok && (prev_m >= prev_s) && (m < g);
        prev_m = m; prev_s = g;
    }
    // steady state, non-null input: the same operations without the tests
    __device__ __forceinline__ void fast_nosig(double px) {
        const double f = ef.fast(px), sl = es.fast(px);
        const double m = f - sl;
        prev_s = eg.fast(m);
        prev_m = m;
    }
};
// EmaCore::step on a non-null value when `cnt` (valid rows seen including this one) and the period are wave-uniform
__device__ __forceinline__ double btw_ema_lock(EmaCore &e, double v, int cnt, int p) {
    if (cnt < p) { e.sum += v; return pq_null(); }
    if (cnt == p) { e.sum += v; e.ema = e.sum / (double)p; return e.ema; }
    e.ema = fma(e.alpha, v - e.ema, e.ema);
    return e.ema;
}
__device__ __forceinline__ void btw_bcast_ema(EmaCore &e, int l) { // every lane receives lane l's core
    e.count = (int64_t)btw_readlane((unsigned long long)e.count, l);
    e.ema = btw_readlane(e.ema, l);
    e.sum = btw_readlane(e.sum, l);
}

// calculate_summary (metrics.rs:7-152) of the equity row that sits in LDS (`px`; overwritten with the daily returns): lane c owns
// rows [c*C, (c+1)*C).  The running max is an exact prefix max; max_drawdown / max_profit / win_rate / total_trades are exact; the
// ordered f64 sums (mean, variance, covariance) are summed per chunk and then across the lanes in a fixed order (<= 1e-12).
// `bm`: the benchmark rows or null -- in LDS with the same geometry, or (bm_linear) a plain series in global memory (the leveraged
// engine's ONE benchmark shared by all symbols: it stays in L2, so it is not worth a second LDS row per wave).
__device__ __forceinline__ void btw_summary(const BtwGeom &g, int lane, double *px, const double *bm, double initial_capital, int trades, int wins,
                                            double *sm, bool bm_linear = false) {
    const int T = g.T, C = g.C, P = g.P;
    auto addr = [&](int i) { return g.addr(i); };
    auto baddr = [&](int i) { return bm_linear ? i : g.addr(i); };
    struct { double initial_capital; } prm{initial_capital};
    const bool has_bench = bm != nullptr;
    // ---- summary (metrics.rs:7-152): lane c owns rows [c*C, (c+1)*C) of the equity row now in LDS
    const int c = lane, lo = c * C, hi = lo + C < T ? lo + C : T;
    const int nrow = hi - lo; // <= 0: idle lane
    double *erow = px + c * P;
    const double init = prm.initial_capital;
    const double NEG_INF = __longlong_as_double((long long)0xFFF0000000000000ULL);
    double lm = NEG_INF;
    for (int b0 = 0; b0 < nrow; b0 += 8) { // (eight LDS reads in flight, then the compares)
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; u++) e[u] = b0 + u < nrow ? erow[b0 + u] : NEG_INF;
#pragma unroll
        for (int u = 0; u < 8; u++) if (e[u] > lm) lm = e[u];
    }
    double Mx = lm; // inclusive prefix max over the lanes (exact in any order)
#define BTW_MAX(CTRL, RM) { const double t = btw_dpp<CTRL, RM>(NEG_INF, Mx); if (t > Mx) Mx = t; }
    BTW_SCAN_STEPS(BTW_MAX)
#undef BTW_MAX
    double max_eq = btw_prev_lane(NEG_INF, Mx);
    if (lane == 0 || !(max_eq > init)) max_eq = init; // the running max starts at initial_capital (metrics.rs:21)
    double prev = init;
    if (c > 0 && nrow > 0) prev = px[addr(lo - 1)];
    const double last_eq = px[addr(T - 1)];
    double max_dd = 0.0, rs = 0.0;
    for (int b0 = 0; b0 < nrow; b0 += 4) { // metrics.rs:26-49, four rows at a time: their eight divisions overlap
        double e[4], mx[4], pv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = b0 + u < nrow ? erow[b0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (b0 + u < nrow && e[u] > max_eq) max_eq = e[u];
            mx[u] = max_eq;
            pv[u] = prev;
            if (b0 + u < nrow) prev = e[u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double dd = (mx[u] > 0.0) ? (mx[u] - e[u]) / mx[u] : 0.0;
            const double rr = (pv[u] > 0.0) ? (e[u] - pv[u]) / pv[u] : 0.0;
            if (b0 + u < nrow) {
                if (dd > max_dd) max_dd = dd;
                rs += rr;
                erow[b0 + u] = rr;
            }
        }
    }
    // wave sums in a fixed order (an inclusive DPP scan, the total read from lane 63): the same value on every lane
    auto wave_sum3 = [&](double &v0, double &v1, double &v2) {
#define BTW_SUM3(CTRL, RM) { v0 += btw_dpp<CTRL, RM>(0.0, v0); v1 += btw_dpp<CTRL, RM>(0.0, v1); v2 += btw_dpp<CTRL, RM>(0.0, v2); }
        BTW_SCAN_STEPS(BTW_SUM3)
#undef BTW_SUM3
        v0 = btw_readlane(v0, 63); v1 = btw_readlane(v1, 63); v2 = btw_readlane(v2, 63);
    };
    double bs = 0.0, pb0 = 0.0;
    const double *brow = bm + c * (bm_linear ? C : P);
    if (has_bench) { // metrics.rs:86-140: the benchmark's daily returns
        pb0 = (c == 0 || nrow <= 0) ? bm[0] : bm[baddr(lo - 1)];
        double pb = pb0;
        for (int b0 = 0; b0 < nrow; b0 += 4) {
            double bv[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { bv[u] = b0 + u < nrow ? brow[b0 + u] : 0.0; pv[u] = pb; if (b0 + u < nrow) pb = bv[u]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const double br = (pv[u] > 0.0) ? (bv[u] - pv[u]) / pv[u] : 0.0; if (b0 + u < nrow) bs += br; }
        }
    }
    {
#define BTW_MAXDD(CTRL, RM) { const double t = btw_dpp<CTRL, RM>(0.0, max_dd); if (t > max_dd) max_dd = t; }
        BTW_SCAN_STEPS(BTW_MAXDD)
#undef BTW_MAXDD
        max_dd = btw_readlane(max_dd, 63);
        double zero = 0.0;
        wave_sum3(rs, bs, zero);
    }
    const double ret_sum = rs;
    const double DAYS = 252.0, RF = 0.03;
    const double total_return = (last_eq - init) / init;
    const double mean = ret_sum / (double)T;
    const double dof = fmax((double)T - 1.0, 1.0);
    const double bmean = bs / (double)T;
    double vs = 0.0, bvs = 0.0, cvs = 0.0;
    {
        double pb = pb0;
        for (int b0 = 0; b0 < nrow; b0 += 4) {
            double rr[4], bv[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                rr[u] = b0 + u < nrow ? erow[b0 + u] : mean;
                bv[u] = 0.0; pv[u] = pb;
                if (has_bench) { bv[u] = b0 + u < nrow ? brow[b0 + u] : 0.0; if (b0 + u < nrow) pb = bv[u]; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double dlt = rr[u] - mean;
                if (b0 + u < nrow) vs += dlt * dlt;
                if (has_bench) {
                    const double br = (pv[u] > 0.0) ? (bv[u] - pv[u]) / pv[u] : 0.0;
                    const double db = br - bmean;
                    if (b0 + u < nrow) { bvs += db * db; cvs += dlt * db; }
                }
            }
        }
    }
    wave_sum3(vs, bvs, cvs);
    const double ann = (total_return > -1.0) ? pow(1.0 + total_return, DAYS / (double)T) - 1.0 : -1.0;
    const double var = vs / dof;
    const double vol = sqrt(var) * sqrt(DAYS);
    const double sharpe = (vol > 0.0) ? (ann - RF) / vol : 0.0;
    const double win_rate = (trades > 0) ? (double)wins / (double)trades : 0.0;
    double alpha = 0.0, beta = 0.0;
    if (has_bench) {
        const double bvar = bvs / dof, cov = cvs / dof;
        if (bvar > 0.0) beta = cov / bvar;
        const double b0 = bm[0], b1 = bm[baddr(T - 1)];
        const double btr = (b0 > 0.0) ? (b1 - b0) / b0 : 0.0;
        const double bann = (btr > -1.0) ? pow(1.0 + btr, DAYS / (double)T) - 1.0 : -1.0;
        alpha = ann - (RF + beta * (bann - RF));
    }
    if (lane == 0) {
        sm[0] = ann; sm[1] = max_dd; sm[2] = alpha; sm[3] = beta; sm[4] = sharpe;
        sm[5] = fmax(total_return, 0.0); sm[6] = win_rate; sm[7] = (double)trades;
    }
}

// Per-lane row masks: bit b = row b of the lane's chunk (MACD signals) or of the lane's 64-row block (signals as inputs).  NW = 1
// covers len <= 4096 (C <= 64); NW = 2 -- chunks of up to 128 rows, two blocks per lane (lane w: blocks w and w + 64) -- len <= 8192.
template <int NW>
struct BtwBits {
    unsigned long long w[NW];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < NW; k++) w[k] = 0;
    }
    __device__ __forceinline__ void or_bit(int b, bool v) { // b runtime (the general per-row path)
        if constexpr (NW == 1) w[0] |= (unsigned long long)v << b;
        else {
            const unsigned long long bit = (unsigned long long)v << (b & 63);
            if (b < 64) w[0] |= bit; else w[1] |= bit;
        }
    }
    __device__ __forceinline__ int popc() const {
        int n = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) n += __popcll(w[k]);
        return n;
    }
};
// lane-held table of 64-bit words indexed by a wave-uniform j < 64 * NW: entry j lives in lane j & 63, word j >> 6
template <int NW>
__device__ __forceinline__ unsigned long long btw_word(const BtwBits<NW> &m, int j) {
    if constexpr (NW == 1) return btw_readlane(m.w[0], j);
    else return j < 64 ? btw_readlane(m.w[0], j) : btw_readlane(m.w[1], j - 64);
}

// ---- several wavefronts per symbol (NWV > 1: small shards, where a symbol's wave would otherwise have a SIMD to itself and the
// kernel time is that one wave's instruction count).  Wave 0 runs the kernel below; the HELPER waves take their share of the three
// parts that are parallel over rows -- staging, the dense fill, the summary -- and meet wave 0 at workgroup barriers in between.
// shd[]: doubles in LDS behind the event tables.
enum { BTW_SH_NULL = 0 /* + wave */, BTW_SH_DENSE = 8, BTW_SH_RED = 16 /* 3 rounds x 3 values x waves */, BTW_SH_DOUBLES = 64 };

// the dense fill of the tiles first, first + step, ...: the event words and the state tables come from LDS (any wave can run it)
template <int NW>
__device__ __forceinline__ void btw_fill_tiles(const BtWaveArgs &a, const BtwGeom &g, int64_t base, int lane, int first, int step, double *px,
                                               const unsigned long long *evw, const double *evr, const double *evp) {
    const int T = g.T, C = g.C;
    BtwBits<NW> mw;
    int pre[NW], carry = 0; // pre[k]: events in the blocks up to and including mine (block = lane + 64 k)
#pragma unroll
    for (int k = 0; k < NW; k++) {
        mw.w[k] = evw[lane + 64 * k];
        int cnt = __popcll(mw.w[k]);
#define BTW_ADD(CTRL, RM) cnt += btw_dpp<CTRL, RM>(0, cnt);
        BTW_SCAN_STEPS(BTW_ADD)
#undef BTW_ADD
        pre[k] = carry + cnt;
        carry += __builtin_amdgcn_readlane(cnt, 63);
    }
    auto before = [&](int blk) -> int { // events in the blocks in front of block `blk` (wave-uniform)
        if (blk == 0) return 0;
        const int j = blk - 1;
        if constexpr (NW == 1) return __builtin_amdgcn_readlane(pre[0], j);
        else return j < 64 ? __builtin_amdgcn_readlane(pre[0], j) : __builtin_amdgcn_readlane(pre[NW - 1], j - 64);
    };
    const int nb2 = (T + 127) / 128;
    const bool wide = (((a.position ? reinterpret_cast<uintptr_t>(a.position + base) : 0) | (a.cash ? reinterpret_cast<uintptr_t>(a.cash + base) : 0) |
                        (a.equity ? reinterpret_cast<uintptr_t>(a.equity + base) : 0)) & 15) == 0;
    const double NANV = __longlong_as_double(0x7FF8000000000000LL);
    const int b0 = (2 * lane) & 63;
    const bool upper = lane >= 32;
    const unsigned long long upto = (2ULL << b0) - 1ULL;
    for (int jj0 = first; jj0 < nb2; jj0 += step) {
        const int jj = __builtin_amdgcn_readfirstlane(jj0);
        const unsigned long long w0 = btw_word(mw, 2 * jj), w1 = btw_word(mw, 2 * jj + 1);
        const unsigned long long ws = upper ? w1 : w0;
        const int i0 = 128 * jj + 2 * lane;
        const int idx0 = before(2 * jj) + (upper ? __popcll(w0) : 0) + __popcll(ws & upto);
        const int idx1 = idx0 + (int)((ws >> (b0 + 1)) & 1ULL);
        const int a0 = g.addr(i0), a1 = g.addr(i0 + 1);
        double x0 = px[a0], x1 = px[a1];
        const double p0 = evr[idx0], c_0 = evp[idx0], p1 = evr[idx1], c_1 = evp[idx1];
        if (pq_isnull(x0)) x0 = NANV; // null -> NaN (vectorized.rs:70-78)
        if (pq_isnull(x1)) x1 = NANV;
        const double e0 = c_0 + p0 * x0, e1 = c_1 + p1 * x1;
        if (wide && i0 + 1 < T) {
            if (a.position) nt_store2(a.position + base + i0, make_double2(p0, p1));
            if (a.cash) nt_store2(a.cash + base + i0, make_double2(c_0, c_1));
            if (a.equity) nt_store2(a.equity + base + i0, make_double2(e0, e1));
        } else {
            if (i0 < T) {
                if (a.position) __builtin_nontemporal_store(p0, &a.position[base + i0]);
                if (a.cash) __builtin_nontemporal_store(c_0, &a.cash[base + i0]);
                if (a.equity) __builtin_nontemporal_store(e0, &a.equity[base + i0]);
            }
            if (i0 + 1 < T) {
                if (a.position) __builtin_nontemporal_store(p1, &a.position[base + i0 + 1]);
                if (a.cash) __builtin_nontemporal_store(c_1, &a.cash[base + i0 + 1]);
                if (a.equity) __builtin_nontemporal_store(e1, &a.equity[base + i0 + 1]);
            }
        }
        if (i0 < 64 * C) { px[a0] = e0; px[a1] = e1; }
    }
}

// calculate_summary over NWV wavefronts: the lanes of the workgroup, in row order, own a QUARTER chunk each (global lane 4 c + q: rows
// c * C + q * Cq ...), the running maximum and the sums are combined wave by wave through shd[] in a fixed order.  Same exact columns
// (prefix maximum, max_drawdown, win_rate, total_trades); the ordered sums are summed per quarter chunk, then per wave, then over the
// waves (<= 1e-12, as in the one-wave form).  Every wave passes the three barriers; only wave 0 writes the row.
template <int NWV>
__device__ __forceinline__ void btw_summary_mw(const BtwGeom &g, int lane, int wv, double *px, const double *bm, double initial_capital, int trades,
                                               int wins, double *sm, double *shd) {
    static_assert(NWV == 2 || NWV == 4, "sub-chunks per chunk: a power of two");
    const int T = g.T, C = g.C, P = g.P;
    const int Cq = (C + NWV - 1) / NWV;
    const int gl = wv * 64 + lane, c = gl / NWV, q = gl % NWV;
    const int lo = c * C + q * Cq;
    int hi = lo + Cq;
    if (hi > (c + 1) * C) hi = (c + 1) * C;
    if (hi > T) hi = T;
    const int nrow = hi - lo; // <= 0: idle lane
    double *erow = px + c * P + q * Cq;
    const bool has_bench = bm != nullptr;
    const double *brow = has_bench ? bm + c * P + q * Cq : nullptr;
    const double init = initial_capital;
    const double NEG_INF = __longlong_as_double((long long)0xFFF0000000000000ULL);
    double lm = NEG_INF;
    for (int b0 = 0; b0 < nrow; b0 += 8) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; u++) e[u] = b0 + u < nrow ? erow[b0 + u] : NEG_INF;
#pragma unroll
        for (int u = 0; u < 8; u++) if (e[u] > lm) lm = e[u];
    }
    double Mx = lm; // inclusive prefix max over this wave's lanes
#define BTW_MAX(CTRL, RM) { const double t = btw_dpp<CTRL, RM>(NEG_INF, Mx); if (t > Mx) Mx = t; }
    BTW_SCAN_STEPS(BTW_MAX)
#undef BTW_MAX
    double max_eq = btw_prev_lane(NEG_INF, Mx);
    // everything another wave will overwrite with its returns is read BEFORE the barrier: my predecessor's last row, the last row
    double prev = init;
    if (gl > 0 && nrow > 0) prev = px[g.addr(lo - 1)];
    const double last_eq = px[g.addr(T - 1)];
    if (lane == 0) shd[BTW_SH_RED + wv] = btw_readlane(Mx, 63);
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NWV - 1; w++) { const double t = shd[BTW_SH_RED + w]; if (w < wv && t > max_eq) max_eq = t; }
    if (gl == 0 || !(max_eq > init)) max_eq = init; // the running max starts at initial_capital (metrics.rs:21)
    double max_dd = 0.0, rs = 0.0;
    for (int b0 = 0; b0 < nrow; b0 += 4) { // metrics.rs:26-49
        double e[4], mx[4], pv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = b0 + u < nrow ? erow[b0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (b0 + u < nrow && e[u] > max_eq) max_eq = e[u];
            mx[u] = max_eq;
            pv[u] = prev;
            if (b0 + u < nrow) prev = e[u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double dd = (mx[u] > 0.0) ? (mx[u] - e[u]) / mx[u] : 0.0;
            const double rr = (pv[u] > 0.0) ? (e[u] - pv[u]) / pv[u] : 0.0;
            if (b0 + u < nrow) {
                if (dd > max_dd) max_dd = dd;
                rs += rr;
                erow[b0 + u] = rr;
            }
        }
    }
    auto wave_sum3 = [&](double &v0, double &v1, double &v2) { // the wave's totals, the same value on every lane
#define BTW_SUM3(CTRL, RM) { v0 += btw_dpp<CTRL, RM>(0.0, v0); v1 += btw_dpp<CTRL, RM>(0.0, v1); v2 += btw_dpp<CTRL, RM>(0.0, v2); }
        BTW_SCAN_STEPS(BTW_SUM3)
#undef BTW_SUM3
        v0 = btw_readlane(v0, 63); v1 = btw_readlane(v1, 63); v2 = btw_readlane(v2, 63);
    };
    // the workgroup's totals of three values, waves in order (round = a region of shd[] of its own)
    auto group3 = [&](int round, double &v0, double &v1, double &v2) {
        double *r = shd + BTW_SH_RED + 4 + round * 3 * NWV;
        if (lane == 0) { r[wv] = v0; r[NWV + wv] = v1; r[2 * NWV + wv] = v2; }
        __syncthreads();
        v0 = r[0]; v1 = r[NWV]; v2 = r[2 * NWV];
#pragma unroll
        for (int w = 1; w < NWV; w++) { v0 += r[w]; v1 += r[NWV + w]; v2 += r[2 * NWV + w]; }
    };
    double bs = 0.0, pb0 = 0.0;
    if (has_bench) { // metrics.rs:86-140: the benchmark's daily returns
        pb0 = (gl == 0 || nrow <= 0) ? bm[0] : bm[g.addr(lo - 1)];
        double pb = pb0;
        for (int b0 = 0; b0 < nrow; b0 += 4) {
            double bv[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { bv[u] = b0 + u < nrow ? brow[b0 + u] : 0.0; pv[u] = pb; if (b0 + u < nrow) pb = bv[u]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const double br = (pv[u] > 0.0) ? (bv[u] - pv[u]) / pv[u] : 0.0; if (b0 + u < nrow) bs += br; }
        }
    }
    {
#define BTW_MAXDD(CTRL, RM) { const double t = btw_dpp<CTRL, RM>(0.0, max_dd); if (t > max_dd) max_dd = t; }
        BTW_SCAN_STEPS(BTW_MAXDD)
#undef BTW_MAXDD
        max_dd = btw_readlane(max_dd, 63);
        double zero = 0.0;
        wave_sum3(rs, bs, zero);
        // (the maximum rides in the third slot as a sum would: combined below by comparison instead)
        double *r = shd + BTW_SH_RED + 4;
        if (lane == 0) { r[wv] = rs; r[NWV + wv] = bs; r[2 * NWV + wv] = max_dd; }
        __syncthreads();
        rs = r[0]; bs = r[NWV]; max_dd = r[2 * NWV];
#pragma unroll
        for (int w = 1; w < NWV; w++) { rs += r[w]; bs += r[NWV + w]; if (r[2 * NWV + w] > max_dd) max_dd = r[2 * NWV + w]; }
    }
    const double ret_sum = rs;
    const double DAYS = 252.0, RF = 0.03;
    const double total_return = (last_eq - init) / init;
    const double mean = ret_sum / (double)T;
    const double dof = fmax((double)T - 1.0, 1.0);
    const double bmean = bs / (double)T;
    double vs = 0.0, bvs = 0.0, cvs = 0.0;
    {
        double pb = pb0;
        for (int b0 = 0; b0 < nrow; b0 += 4) {
            double rr[4], bv[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                rr[u] = b0 + u < nrow ? erow[b0 + u] : mean;
                bv[u] = 0.0; pv[u] = pb;
                if (has_bench) { bv[u] = b0 + u < nrow ? brow[b0 + u] : 0.0; if (b0 + u < nrow) pb = bv[u]; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double dlt = rr[u] - mean;
                if (b0 + u < nrow) vs += dlt * dlt;
                if (has_bench) {
                    const double br = (pv[u] > 0.0) ? (bv[u] - pv[u]) / pv[u] : 0.0;
                    const double db = br - bmean;
                    if (b0 + u < nrow) { bvs += db * db; cvs += dlt * db; }
                }
            }
        }
    }
    wave_sum3(vs, bvs, cvs);
    group3(1, vs, bvs, cvs);
    if (wv != 0) return;
    const double ann = (total_return > -1.0) ? pow(1.0 + total_return, DAYS / (double)T) - 1.0 : -1.0;
    const double var = vs / dof;
    const double vol = sqrt(var) * sqrt(DAYS);
    const double sharpe = (vol > 0.0) ? (ann - RF) / vol : 0.0;
    const double win_rate = (trades > 0) ? (double)wins / (double)trades : 0.0;
    double alpha = 0.0, beta = 0.0;
    if (has_bench) {
        const double bvar = bvs / dof, cov = cvs / dof;
        if (bvar > 0.0) beta = cov / bvar;
        const double b0 = bm[0], b1 = bm[g.addr(T - 1)];
        const double btr = (b0 > 0.0) ? (b1 - b0) / b0 : 0.0;
        const double bann = (btr > -1.0) ? pow(1.0 + btr, DAYS / (double)T) - 1.0 : -1.0;
        alpha = ann - (RF + beta * (bann - RF));
    }
    if (lane == 0) {
        sm[0] = ann; sm[1] = max_dd; sm[2] = alpha; sm[3] = beta; sm[4] = sharpe;
        sm[5] = fmax(total_return, 0.0); sm[6] = win_rate; sm[7] = (double)trades;
    }
}

// what a helper wave does (see above); `a.summary` etc. are workgroup-uniform, so every wave passes the same barriers
template <int NW, int NWV>
__device__ __forceinline__ void btw_helper_wave(const BtWaveArgs &a, const BtwGeom &geo, int64_t base, int64_t s, int lane, int wv, double *px, double *bm,
                                                const unsigned long long *evw, const double *evp, const double *evr, double *shd) {
    const bool ns = btw_stage(geo, lane, a.price + base, px, pq_null(), wv, NWV);
    if (a.bench) (void)btw_stage(geo, lane, a.bench + base, bm, 0.0, wv, NWV);
    const bool any = btw_ballot(ns) != 0;
    if (lane == 0) shd[BTW_SH_NULL + wv] = any ? 1.0 : 0.0;
    __syncthreads(); // B1: the rows are staged
    __syncthreads(); // B2: wave 0 has marked the events and run the cash chain
    if (shd[BTW_SH_DENSE] != 0.0) btw_fill_tiles<NW>(a, geo, base, lane, wv, NWV, px, evw, evr, evp);
    __syncthreads(); // B3: the equity row is complete
    if (!a.summary) return;
    btw_summary_mw<NWV>(geo, lane, wv, px, a.bench ? bm : nullptr, a.prm.initial_capital, 0, 0, a.summary + s * PQ_SUMMARY_COLS, shd);
}

#ifdef PQ_BTW_PROF // scripts/ab_build.sh only: per-phase device time summed over the waves, stats[4 + k] in 10 ns ticks
#define BTW_T(k) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if (lane == 0 && a.stats) atomicAdd(a.stats + 4 + (k), t__ - t_prev); t_prev = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BTW_T(k)
#endif
template <bool MACD, int NW = 1, int NWV = 1>
__global__ __launch_bounds__(64 * NWV) void bt_wave_kernel(BtWaveArgs a, Dims d) {
#ifdef PQ_BTW_PROF
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __align__(16) unsigned char btw_lds[];
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6; // wv > 0: a helper wave (NWV > 1, btw_helper_wave)
    const int64_t s = blockIdx.x;
    const int T = (int)dims_len(d, s), C = a.C, P = a.P, PC = P - C; // (ragged: C, P, kcap are sized for the longest series)
    const unsigned magic = a.magic;
    const int64_t base = dims_base(d, s);
    double *px = reinterpret_cast<double *>(btw_lds);   // [64 * P]: row i at i + (i / C) * (P - C); later the equity row, then r
    double *bm = px + 64 * P;                           // [64 * P] when a.bench
    unsigned long long *evw = reinterpret_cast<unsigned long long *>(bm + (a.bench ? 64 * P : 0)); // [64 * NW] event flags per block
    double *evp = reinterpret_cast<double *>(evw + 64 * NW); // [kcap + 1] event prices; then [k] = the cash after k events (k = 0: the initial capital)
    double *evr = evp + a.kcap + 1;                     // [kcap + 1] then [k] = the position after k events
    double *shd = evr + a.kcap + 1;                     // [BTW_SH_DOUBLES] what the waves of a workgroup tell each other (NWV > 1)
    auto addr = [&](int i) { return i + (int)(((unsigned)i * magic) >> 20) * PC; };
    const pq_bt_params prm = a.prm;
    if (T == 0) {
        if (a.summary && wv == 0 && lane < 8) a.summary[s * PQ_SUMMARY_COLS + lane] = 0.0;
        return;
    }
    const BtwGeom geo{T, C, P, magic};
    if constexpr (NWV > 1) {
        if (wv != 0) { btw_helper_wave<NW, NWV>(a, geo, base, s, lane, wv, px, bm, evw, evp, evr, shd); return; }
    }

    // ---- phase 0: the symbol's rows, coalesced, into LDS; signal masks when the signals are inputs
    BtwBits<NW> bmask, smask; // MACD: bit b of lane c = row c*C + b; else: bit b of word k of lane w = row 64 * (w + 64 * k) + b
    bmask.clear(); smask.clear();
    bool null_seen;
    null_seen = btw_stage(geo, lane, a.price + base, px, pq_null(), 0, NWV);
    if (a.bench) (void)btw_stage(geo, lane, a.bench + base, bm, 0.0, 0, NWV);
    if constexpr (NWV > 1) {
        __syncthreads(); // B1: the helpers' share of the rows is in LDS, their null flags in shd[]
#pragma unroll
        for (int w = 1; w < NWV; w++) null_seen |= shd[BTW_SH_NULL + w] != 0.0;
    }
    if (!MACD) {
        btw_lds_fence();
        for (int j0 = 0; j0 < C; j0 += 8) {
            unsigned char bb[8], sb[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = 64 * (j0 + u) + lane;
                const bool in = j0 + u < C && i < T;
                bb[u] = in ? a.buy[base + i] : 0;
                sb[u] = in ? a.sell[base + i] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (j0 + u < C) {
                    const double v = px[addr(64 * (j0 + u) + lane)];
                    const bool valid = !(isnan(v) || v <= 0.0); // vectorized.rs:141: such rows leave the state untouched (a NULL is a NaN)
                    const unsigned long long wb = btw_ballot(bb[u] != 0 && valid), ws = btw_ballot(sb[u] != 0 && valid);
                    if (lane == ((j0 + u) & 63)) {
                        if (NW == 1 || j0 + u < 64) { bmask.w[0] = wb; smask.w[0] = ws; }
                        else { bmask.w[NW - 1] = wb; smask.w[NW - 1] = ws; }
                    }
                }
            }
        }
    }
    const bool any_null = btw_ballot(null_seen) != 0;
    btw_lds_fence();
    BTW_T(0);

    // ---- phase 1 (A): MACD-cross signals by speculative chunks
    if (MACD) {
        const int c = lane;
        BtwMacd st;
        st.init(a.fast, a.slow, a.sig, T);
        if (!(st.ef.dead || st.es.dead || st.eg.dead)) { // a dead average => no signal on any row
            const int nlive = (T + C - 1) / C;
            const bool live = c < nlive;
            const int pf = a.fast, ps = a.slow, pg = a.sig;
            const int R1 = pf > ps ? pf : ps;  // m = fast - slow exists from the R1-th row on
            const int H = R1 > pg ? R1 : pg;   // after H rows every average is seeded and prev_m / prev_s are values
            const bool head = !any_null && H <= C;
            // warm-up chunks: nW from an arbitrary seed; nW2 (much shorter) when the seeds come from the affine prefix scan below
            const int nWe = head ? a.nW2 : a.nW;
            const int nk = nlive < nWe + 1 ? nlive : nWe + 1; // chunk iterations
            // Schedule.  Lanes c > nW ("speculative") start nW chunks early and reach their own chunk in the LAST iteration.  The
            // first nW + 1 chunks have no room for a warm-up: lane 0 walks them serially from row 0 (exact) WHILE the speculative
            // lanes warm up, and hands its state after chunk k to lane k + 1; in the last iteration the lanes 1 .. nW run their
            // own chunk from that exact state.
            // Null-free series (`head`): lane 0 first does chunk 0 alone -- the rows on which the averages are being seeded
            // (uniform-count branches), then the rest of the chunk in steady state -- and the speculative lanes start in steady
            // state from an arbitrary seed (the first price of their warm-up): every iteration of the loop is then straight-line
            // code.  Otherwise (nulls, or a seeding phase longer than a chunk) the speculative lanes start as if the series began
            // at their warm-up row and the loop runs the general per-row state machine while any active lane is not seeded.
            const bool spec = live && c > nWe;
            auto hand_over = [&](int to) { // lane 0's state -> lane `to`: the exact state in front of that lane's chunk
                BtwMacd h = st;
                btw_bcast_ema(h.ef, 0); btw_bcast_ema(h.es, 0); btw_bcast_ema(h.eg, 0);
                h.prev_m = btw_readlane(st.prev_m, 0); h.prev_s = btw_readlane(st.prev_s, 0);
                if (lane == to) st = h;
            };
            // steady rows [b0, C) of chunk `row` with their signals kept: buy = !(m' > g') && (m > g), sell = !(m' < g') && (m < g)
            // (a NaN average is absorbing -- every later m, g is NaN and both predicates stay false --, so "not greater on the
            // previous row" can stand for the reference's "less or equal")
            auto steady_rows_keep = [&](const double *row, int b0) {
#pragma unroll
                for (int k = 0; k < NW; k++) { // rows [64 k, 64 k + 64) of the chunk -> word k
                    const int lo = 64 * k, hi = C < lo + 64 ? C : lo + 64;
                    const int s0 = (b0 > lo ? b0 : lo) - lo; // first bit of this word that is walked
                    unsigned long long gt = 0, lt = 0, vm = 0;
                    const bool pgt = st.prev_m > st.prev_s, plt = st.prev_m < st.prev_s; // the row in front of bit s0
                    for (int b = lo + s0; b < hi; b++) {
                        const double x = row[b];
                        st.fast_nosig(x);
                        gt |= (unsigned long long)(st.prev_m > st.prev_s) << (b - lo);
                        lt |= (unsigned long long)(st.prev_m < st.prev_s) << (b - lo);
                        vm |= (unsigned long long)(x > 0.0) << (b - lo); // valid price (NaN / NULL compare false)
                    }
                    const unsigned s1 = s0 < 64 ? s0 : 63; // (s0 >= 64: the word is not walked, gt = lt = 0)
                    const unsigned long long pvg = (gt << 1) | ((unsigned long long)pgt << s1), pvl = (lt << 1) | ((unsigned long long)plt << s1);
                    bmask.w[k] = gt & ~pvg & vm;
                    smask.w[k] = lt & ~pvl & vm;
                }
            };
            BTW_T(14);
            if (head) {
                if (lane == 0) { // rows 0 .. H-1: no signal is possible yet (prev_m / prev_s are null until row H - 1)
                    // Rows 0 .. R1-2 only feed the two price averages, each a chain of its own (EmaCore::step's operations for the
                    // counts 1 .. n: the sum in row order, the seed, then the recurrence): two plain loops instead of one loop with
                    // three uniform branches per row and average -- this lane is alone on its SIMD here, every instruction counts.
                    auto advance = [&](EmaCore &e, int p, int n) {
                        const int ns = n < p ? n : p;
                        for (int b0 = 0; b0 < ns; b0 += 8) { // (eight rows leave LDS together: the serial part is the sum, not the reads)
                            double xr[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) xr[u] = px[b0 + u < ns ? b0 + u : 0];
#pragma unroll
                            for (int u = 0; u < 8; u++) if (b0 + u < ns) e.sum += xr[u];
                        }
                        if (n < p) return;
                        e.ema = e.sum / (double)p;
                        for (int b0 = p; b0 < n; b0 += 8) {
                            double xr[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) xr[u] = px[b0 + u < n ? b0 + u : 0];
#pragma unroll
                            for (int u = 0; u < 8; u++) if (b0 + u < n) e.ema = fma(e.alpha, xr[u] - e.ema, e.ema);
                        }
                    };
                    advance(st.ef, pf, R1 - 1);
                    advance(st.es, ps, R1 - 1);
                    for (int cnt = R1; cnt <= H; cnt++) { // m exists from here on; before, the signal line saw zeros: its sum and its seed stay +0.0
                        const double x = px[cnt - 1];
                        const double f = btw_ema_lock(st.ef, x, cnt, pf), sl = btw_ema_lock(st.es, x, cnt, ps);
                        st.prev_m = f - sl;
                        st.prev_s = btw_ema_lock(st.eg, st.prev_m, cnt, pg);
                    }
                    st.ef.count = st.es.count = st.eg.count = H;
                    BTW_T(15);
                    steady_rows_keep(px, H);
                }
                BTW_T(10);
                hand_over(1);
                BTW_T(11);
                // SEEDS.  In exact arithmetic an EMA is the affine map e -> (1-a) e + a x per row; a chunk is the composition of its
                // rows' maps, and affine maps compose associatively: one Horner pass per lane over its chunk and one wave prefix scan
                // give every average's value at every chunk boundary to ~1e-15 -- not the bits (each fma of the true recurrence rounds),
                // but a seed a few ulps from the truth, from which the speculative run merges bitwise within ~1/alpha rows instead of
                // ~36/alpha.  Lane 0 enters the scan as the constant map onto its exact state after chunk 0.  The signal line needs
                // the per-row m = fast - slow: a second pass recomputes the rows of the chunk from the approximate boundary values.
                {
                    const double af = st.ef.alpha, as = st.es.alpha, ag = st.eg.alpha;
                    const double qf = 1.0 - af, qs = 1.0 - as, qg = 1.0 - ag;
                    const double *row = px + (live ? c : 0) * P;
                    double Af = 1.0, Bf = 0.0, As = 1.0, Bs = 0.0;
                    int b = 0;
                    for (; b + 8 <= C; b += 8) { // (eight LDS reads in flight per batch)
                        double x[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) x[u] = row[b + u];
#pragma unroll
                        for (int u = 0; u < 8; u++) { Bf = fma(qf, Bf, af * x[u]); Bs = fma(qs, Bs, as * x[u]); Af *= qf; As *= qs; }
                    }
                    for (; b < C; b++) { const double x = row[b]; Bf = fma(qf, Bf, af * x); Bs = fma(qs, Bs, as * x); Af *= qf; As *= qs; }
                    if (lane == 0) { Af = 0.0; Bf = st.ef.ema; As = 0.0; Bs = st.es.ema; }
                    // inclusive scan: (A, B) of lanes 0 .. c composed in row order (identity: A = 1, B = 0)
#define BTW_AFFINE2(CTRL, RM)                                                                                \
    {                                                                                                        \
        const double pAf = btw_dpp<CTRL, RM>(1.0, Af), pBf = btw_dpp<CTRL, RM>(0.0, Bf);                     \
        const double pAs = btw_dpp<CTRL, RM>(1.0, As), pBs = btw_dpp<CTRL, RM>(0.0, Bs);                     \
        Bf = fma(Af, pBf, Bf); Af *= pAf; Bs = fma(As, pBs, Bs); As *= pAs;                                  \
    }
                    BTW_SCAN_STEPS(BTW_AFFINE2)
#undef BTW_AFFINE2
                    BTW_T(12);
                    // Bf / Bs: fast / slow average after chunk c (approximate; exact on lane 0)
                    double f = btw_prev_lane(Bf, Bf), sl = btw_prev_lane(Bs, Bs); // ... in front of my chunk
                    double Ag = 1.0, Bg = 0.0;
                    b = 0;
                    for (; b + 8 <= C; b += 8) {
                        double x[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) x[u] = row[b + u];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            f = fma(af, x[u] - f, f); sl = fma(as, x[u] - sl, sl);
                            Bg = fma(qg, Bg, ag * (f - sl));
                            Ag *= qg;
                        }
                    }
                    for (; b < C; b++) {
                        const double x = row[b];
                        f = fma(af, x - f, f); sl = fma(as, x - sl, sl);
                        Bg = fma(qg, Bg, ag * (f - sl));
                        Ag *= qg;
                    }
                    if (lane == 0) { Ag = 0.0; Bg = st.eg.ema; }
#define BTW_AFFINE1(CTRL, RM)                                                                                \
    {                                                                                                        \
        const double pAg = btw_dpp<CTRL, RM>(1.0, Ag), pBg = btw_dpp<CTRL, RM>(0.0, Bg);                     \
        Bg = fma(Ag, pBg, Bg); Ag *= pAg;                                                                    \
    }
                    BTW_SCAN_STEPS(BTW_AFFINE1)
#undef BTW_AFFINE1
                    BTW_T(13);
                    // a speculative lane starts in front of chunk c - nWe: the values after chunk c - nWe - 1
                    const double sf = __shfl_up(Bf, nWe + 1), ss = __shfl_up(Bs, nWe + 1), sg = __shfl_up(Bg, nWe + 1);
                    if (spec) {
                        st.ef.ema = sf; st.es.ema = ss; st.eg.ema = sg;
                        st.ef.count = st.es.count = st.eg.count = H;
                        st.prev_m = sf - ss; st.prev_s = sg;
                    }
                }
            }
            BTW_T(5);
            double s_f = 0, s_s = 0, s_g = 0, s_pm = 0, s_ps = 0;
            bool s_steady = false;
            for (int kk = 0; kk < nk; kk++) {
                const int k = __builtin_amdgcn_readfirstlane(kk); // (the compiler otherwise keeps the counter in a VGPR and treats
                                                                  // everything derived from it as divergent)
                const bool last = k == nk - 1;
                const int q = last ? c : (spec ? c - nWe + k : (head ? k + 1 : k));
                const bool lane0_active = head ? k + 2 < nk : true; // wave-uniform: lane 0 walks a chunk in this iteration
                const bool lane0 = c == 0 && lane0_active;
                const bool active = live && (last ? (c > 0 || (nk == 1 && !head)) : (spec || lane0));
                const bool rec = active && q == c; // this lane's own rows: their signals are kept
                if (last) { s_f = st.ef.ema; s_s = st.es.ema; s_g = st.eg.ema; s_pm = st.prev_m; s_ps = st.prev_s; s_steady = st.steady(); }
                const double *row = px + (active ? q : 0) * P;
                const bool fastk = !any_null && btw_ballot(active && !st.steady()) == 0; // wave-uniform
                const bool anyrec = btw_ballot(rec) != 0;
                if (fastk && !anyrec) {
                    if (active) {
                        int b = 0;
                        for (; b + 8 <= C; b += 8) {
                            double x[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) x[u] = row[b + u];
#pragma unroll
                            for (int u = 0; u < 8; u++) st.fast_nosig(x[u]);
                        }
                        for (; b < C; b++) st.fast_nosig(row[b]);
                    }
                } else if (fastk) {
                    if (active) {
                        const BtwBits<NW> kb = bmask, ks = smask;
                        steady_rows_keep(row, 0);
                        if (!rec) { bmask = kb; smask = ks; }
                    }
                } else {
                    if (active)
                        for (int b = 0; b < C; b++) {
                            const double x = row[b];
                            bool bu, se;
                            st.step(q * C + b, x, bu, se);
                            if (rec) {
                                const bool valid = !(isnan(x) || x <= 0.0);
                                bmask.or_bit(b, bu && valid);
                                smask.or_bit(b, se && valid);
                            }
                        }
                }
#ifdef PQ_BTW_PROF
                if (k == nk - 2) BTW_T(6); else if (last) BTW_T(7);
#endif
                if (!last && lane0_active) hand_over(head ? k + 2 : k + 1);
            }
            // verification: my state at my first row == my predecessor's state after its last row, as raw bits
            auto mismatch = [&]() {
                // every cross-lane read BEFORE any lane-dependent control flow: a DPP move executed under a short-circuited `&&`
                // finds the source lanes that already failed switched off and returns `old` for them -- lane c then fails its
                // k-th comparison because lane c - 1 failed its (k-1)-th (lane 1 always fails its first: lane 0 walked on), and
                // with five comparisons the lanes up to 5 were flagged (and re-run) for nothing (rounds 3-4: every symbol, whenever
                // fewer than five warm-up chunks were planned)
                const bool p_steady = btw_prev_lane(0, (int)st.steady()) != 0;
                const unsigned long long q_f = btw_bits(btw_prev_lane(0.0, st.ef.ema)), q_s = btw_bits(btw_prev_lane(0.0, st.es.ema)),
                                         q_g = btw_bits(btw_prev_lane(0.0, st.eg.ema)), q_pm = btw_bits(btw_prev_lane(0.0, st.prev_m)),
                                         q_ps = btw_bits(btw_prev_lane(0.0, st.prev_s));
                const bool same = s_steady & p_steady & (btw_bits(s_f) == q_f) & (btw_bits(s_s) == q_s) & (btw_bits(s_g) == q_g) &
                                  (btw_bits(s_pm) == q_pm) & (btw_bits(s_ps) == q_ps);
                return spec && !same;
            };
            unsigned long long mism = btw_ballot(mismatch());
            const int n_failed = __popcll(mism);
            int n_rerun = 0;
            while (mism) {
                const int cs = __builtin_ctzll(mism); // lowest failing chunk: its predecessor is exact
                mism &= mism - 1;
                BtwMacd pr = st;                      // every lane: the END state of lane cs - 1
                btw_bcast_ema(pr.ef, cs - 1); btw_bcast_ema(pr.es, cs - 1); btw_bcast_ema(pr.eg, cs - 1);
                pr.prev_m = btw_readlane(st.prev_m, cs - 1); pr.prev_s = btw_readlane(st.prev_s, cs - 1);
                if (lane == cs) {
                    st = pr;
                    s_f = st.ef.ema; s_s = st.es.ema; s_g = st.eg.ema; s_pm = st.prev_m; s_ps = st.prev_s; s_steady = true;
                    bmask.clear(); smask.clear();
                    const double *row = px + cs * P;
                    if (!any_null && st.steady()) steady_rows_keep(row, 0);
                    else
                        for (int b = 0; b < C; b++) {
                            const double x = row[b];
                            bool bu, se;
                            st.step(cs * C + b, x, bu, se);
                            const bool valid = !(isnan(x) || x <= 0.0);
                            bmask.or_bit(b, bu && valid);
                            smask.or_bit(b, se && valid);
                        }
                }
                n_rerun++;
                // the re-run changed lane cs's end state: its successor is tested again
                const bool again = mismatch() && lane == cs + 1;
                mism |= btw_ballot(again);
            }
            if (a.stats && lane == 0) {
                atomicAdd(a.stats + 1, (unsigned long long)n_failed);
                atomicAdd(a.stats + 2, (unsigned long long)n_rerun);
            }
        }
    }
    if (a.stats && lane == 0) atomicAdd(a.stats + 0, 1ULL);
    BTW_T(4);

    // ---- phase 2 (B): which rows are events?  A row is an event iff its signal finds the pool in the right state: flat for a
    // buy, long for a sell.  Assuming every buy can afford a share, that is a two-state automaton over the rows; automata
    // compose associatively, so the per-lane transfer functions are combined by a wave prefix scan and every lane marks the
    // events among its own rows.  (A pool that cannot afford one share makes a buy signal a non-event: the walks below notice
    // -- qty <= 0 -- and fall back to searching the signal masks row by row from there.)
    const int CQ = MACD ? C : 64;
    BtwBits<NW> myword; // entry j (lane j & 63, word j >> 6): bit l = row 64 * j + l is an event
    BtwBits<NW> evm;    // the same events in the mapping of bmask / smask
    constexpr int NGRP = MACD ? 1 : NW; // scan groups in row order: the lane's whole chunk (MACD) / one group per block word (inputs)
    int K = 0, kexcl[NGRP];             // events of the symbol; events in front of my rows (of each group)
    {
        int s_in = 0; // the pool (0 flat, 1 long) in front of the group's first row
        evm.clear();
#pragma unroll
        for (int gi = 0; gi < NGRP; gi++) {
            BtwBits<NW> ev0, ev1; // events among my rows if the pool arrives flat / long
            ev0.clear(); ev1.clear();
            int st0 = 0, st1 = 1; // state after my rows
#pragma unroll
            for (int k = 0; k < NW; k++) {
                if (!MACD && k != gi) continue;
                unsigned long long m = bmask.w[k] | smask.w[k];
                while (m) {
                    const int b = __builtin_ctzll(m);
                    m &= m - 1;
                    const bool isb = (bmask.w[k] >> b) & 1, iss = (smask.w[k] >> b) & 1;
                    if (st0 ? iss : isb) { st0 ^= 1; ev0.w[k] |= 1ULL << b; }
                    if (st1 ? iss : isb) { st1 ^= 1; ev1.w[k] |= 1ULL << b; }
                }
            }
            int f0 = st0, f1 = st1; // inclusive scan of the composition: state after lanes 0..c given the state in front of lane 0
#define BTW_AUTO(CTRL, RM)                                                                                   \
    {                                                                                                        \
        const int p0 = btw_dpp<CTRL, RM>(0, f0), p1 = btw_dpp<CTRL, RM>(1, f1); /* identity: 0 -> 0, 1 -> 1 */ \
        const int n0 = p0 ? f1 : f0, n1 = p1 ? f1 : f0;                                                      \
        f0 = n0; f1 = n1;                                                                                    \
    }
            BTW_SCAN_STEPS(BTW_AUTO)
#undef BTW_AUTO
            const int q0 = btw_prev_lane(0, f0), q1 = btw_prev_lane(1, f1); // the lower lanes' rows as a map (lane 0: identity)
            const int in = s_in ? q1 : q0;
            int mine = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                if (!MACD && k != gi) continue;
                evm.w[k] = in ? ev1.w[k] : ev0.w[k];
                mine += __popcll(evm.w[k]);
            }
            int cnt = mine;
#define BTW_ADD(CTRL, RM) cnt += btw_dpp<CTRL, RM>(0, cnt);
            BTW_SCAN_STEPS(BTW_ADD)
#undef BTW_ADD
            kexcl[gi] = K + cnt - mine;
            K += __builtin_amdgcn_readlane(cnt, 63);
            if (gi + 1 < NGRP) s_in = s_in ? __builtin_amdgcn_readlane(f1, 63) : __builtin_amdgcn_readlane(f0, 63);
        }
        if (MACD) { // chunk-mapped bits -> block words, through LDS
#pragma unroll
            for (int k = 0; k < NW; k++) evw[lane + 64 * k] = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                unsigned long long e = evm.w[k];
                while (e) {
                    const int row = lane * C + 64 * k + __builtin_ctzll(e);
                    e &= e - 1;
                    atomicOr(&evw[row >> 6], 1ULL << (row & 63));
                }
            }
            btw_lds_fence();
#pragma unroll
            for (int k = 0; k < NW; k++) myword.w[k] = evw[lane + 64 * k];
        } else myword = evm;
    }
    BTW_T(1);

    double pos = 0.0, avail = prm.initial_capital, entry_cost = 0.0;
    int trades = 0, wins = 0;
    bool flat = true;
    // The serial part of a symbol is the cash recurrence through its ~100 round trips, and an f64 operation that waits for its
    // predecessor costs ~22 cycles on this chip (scripts/ubench/f64lat.hip): the IEEE division in qty = floor(cash * size / exec)
    // alone is 13 dependent operations.  exec depends on the price only, so its reciprocal is taken lane-parallel, OFF the chain,
    // and the chain uses q~ = deploy * (1 / exec): two roundings instead of one, |q~ - RN(deploy / exec)| <= 1.5 * 2^-52 * q.
    // floor(q~) can differ from floor(RN(deploy / exec)) only if q~ lies within that distance of an integer; such a row
    // (probability ~1e-11), a non-finite / huge q~ and a buy that cannot afford a share are not decided by the fast forms: they
    // set `bad`, and the walk is then repeated from its saved state with the reference's own operations.  No branch in the chain.
    bool bad = false;
    auto buy_fast = [&](double p, double r) { // vectorized.rs:146-161 on a flat pool (pos == +0.0: cash + pos * price == cash)
        const double exec = p + prm.buy_slippage;
        const double deploy = avail * prm.position_size;
        const double qa = deploy * r;
        const double qty = floor(qa);
        // decided here only if 1 <= q~ < 2^28 and its fraction lies in (2^-22, 1 - 2^-22) (the admissible error is q~ * 2^-51 <=
        // 2^-23); the two range tests are integer compares on the high words, off the f64 dependency chain
        const unsigned hq = (unsigned)(btw_bits(qa) >> 32), hf = (unsigned)(btw_bits(qa - qty) >> 32);
        bad |= (hq - 0x3FF00000u) >= (0x41B00000u - 0x3FF00000u) || (hf - 0x3E900000u) >= (0x3FEFFFFEu - 0x3E900000u);
        const double cost = qty * exec;
        const double fee = fmax(cost * prm.buy_commission_rate, prm.min_commission);
        pos += qty;
        avail -= cost + fee;
        entry_cost = pos * p;
        trades += 1;
    };
    auto sell_fast = [&](double p) { // :162-175
        const double exec = p - prm.sell_slippage;
        const double revenue = pos * exec;
        const double fee = fmax(revenue * prm.sell_commission_rate, prm.min_commission);
        const double net = revenue - fee;
        wins += net > entry_cost ? 1 : 0;
        avail += net;
        pos = 0.0;
    };
    // The reference's own operations (an infinite price makes qty NaN there and 0 here: no trade either way).
    auto trade = [&](double p) -> bool { // false: a buy that cannot afford one share, nothing changes
        if (flat) {
            const double exec = p + prm.buy_slippage;
            const double deploy = avail * prm.position_size;
            const double qty = floor(deploy / exec);
            if (!(qty > 0.0)) return false;
            const double cost = qty * exec;
            const double fee = fmax(cost * prm.buy_commission_rate, prm.min_commission);
            pos += qty;
            avail -= cost + fee;
            entry_cost = pos * p;
            trades += 1;
            flat = false;
        } else {
            sell_fast(p);
            flat = true;
        }
        return true;
    };

    // ---- phases 3 + 4, dense form (the usual case): the events' prices are gathered into a list, the chain runs over the list,
    // the rows are filled from the list of states afterwards -- the chain sees no stores and the stores are 16 bytes per lane.
    bool dense = K <= a.kcap;
    if (dense) {
        { // event k of the symbol -> evp[k] = its price (the lane that owns the row knows k)
            int k = kexcl[0];
#pragma unroll
            for (int wk = 0; wk < NW; wk++) {
                unsigned long long e = evm.w[wk];
                if (!MACD) k = kexcl[wk < NGRP ? wk : 0];
                while (e) {
                    const int b = __builtin_ctzll(e);
                    e &= e - 1;
                    evp[k++] = MACD ? px[lane * P + 64 * wk + b] : px[addr(64 * (lane + 64 * wk) + b)];
                }
            }
        }
        btw_lds_fence();
        // event k lives in lane k & 63 of register group k >> 6: the chain takes its inputs with v_readlane and leaves its results
        // with a lane-select -- no LDS round trip, no wait inside the recurrence
        double ev[BTW_NG], rv[BTW_NG], tpv[BTW_NG], tcv[BTW_NG];
#pragma unroll
        for (int g = 0; g < BTW_NG; g++) {
            ev[g] = 1.0; rv[g] = 1.0; tpv[g] = 0.0; tcv[g] = 0.0;
            if (64 * g < K) {
                if (64 * g + lane < K) ev[g] = evp[64 * g + lane];
                rv[g] = 1.0 / (ev[g] + prm.buy_slippage); // the pool starts flat: even events are buys (odd lanes: unused)
            }
        }
        BTW_T(8);
        // One round trip = one basic block: the sell's arithmetic overlaps the tail of the buy's dependency chain.
#pragma unroll
        for (int g = 0; g < BTW_NG; g++) {
            if (64 * g < K) {
                const int n = K - 64 * g < 64 ? K - 64 * g : 64;
                for (int t0 = 0; t0 + 1 < n; t0 += 2) {
                    const int t = __builtin_amdgcn_readfirstlane(t0);
                    const double pb = btw_readlane(ev[g], t), rb = btw_readlane(rv[g], t), psl = btw_readlane(ev[g], t + 1);
                    buy_fast(pb, rb);
                    if (lane == t) { tpv[g] = pos; tcv[g] = avail; }
                    sell_fast(psl);
                    if (lane == t + 1) tcv[g] = avail; // the position after a sell is 0.0
                }
                if (n & 1) { // last event of the series: the position stays open
                    buy_fast(btw_readlane(ev[g], n - 1), btw_readlane(rv[g], n - 1));
                    if (lane == n - 1) { tpv[g] = pos; tcv[g] = avail; }
                }
            }
        }
        // after k events: evr[k] = position, evp[k] = cash (the fill looks states up by event count; no event yet: flat, the initial capital)
#pragma unroll
        for (int g = 0; g < BTW_NG; g++)
            if (64 * g + lane < K) { evr[64 * g + lane + 1] = tpv[g]; evp[64 * g + lane + 1] = tcv[g]; }
        if (lane == 0) { evr[0] = 0.0; evp[0] = prm.initial_capital; }
        flat = !(K & 1);
        btw_lds_fence();
        BTW_T(9);
        if (bad) { // wave-uniform: start over in the block form below
            dense = false;
            pos = 0.0; avail = prm.initial_capital; entry_cost = 0.0; trades = 0; wins = 0; flat = true;
        }
    }
    if constexpr (NWV > 1) {
        if (!MACD) { // (MACD: the marking left the block words in evw[])
#pragma unroll
            for (int k = 0; k < NW; k++) evw[lane + 64 * k] = myword.w[k];
        }
        if (lane == 0) shd[BTW_SH_DENSE] = dense ? 1.0 : 0.0;
        __syncthreads(); // B2: event words, state tables and `dense` are in LDS
        if (dense) btw_fill_tiles<NW>(a, geo, base, lane, 0, NWV, px, evw, evr, evp);
    }
    if (NWV == 1 && dense) { // fill: tiles of 128 rows, lane l = rows 128 * jj + 2 * l, + 1
        const int nb2 = (T + 127) / 128;
        const bool wide = (((a.position ? reinterpret_cast<uintptr_t>(a.position + base) : 0) | (a.cash ? reinterpret_cast<uintptr_t>(a.cash + base) : 0) |
                            (a.equity ? reinterpret_cast<uintptr_t>(a.equity + base) : 0)) & 15) == 0;
        const double NANV = __longlong_as_double(0x7FF8000000000000LL);
        int kbase = 0;
        const int b0 = (2 * lane) & 63;
        const bool upper = lane >= 32;
        const unsigned long long upto = (2ULL << b0) - 1ULL;
        struct Tile { int i0, a0, a1, idx0, idx1; double x0, x1; };
        auto open = [&](int jj, Tile &t) { // my two rows of tile jj: their prices, and how many events lie at or before each
            const unsigned long long w0 = btw_word(myword, 2 * jj), w1 = btw_word(myword, 2 * jj + 1);
            const unsigned long long ws = upper ? w1 : w0;
            const int c0 = __popcll(w0);
            t.i0 = 128 * jj + 2 * lane;
            t.idx0 = kbase + (upper ? c0 : 0) + __popcll(ws & upto);
            t.idx1 = t.idx0 + (int)((ws >> (b0 + 1)) & 1ULL);
            t.a0 = addr(t.i0); t.a1 = addr(t.i0 + 1);
            t.x0 = px[t.a0]; t.x1 = px[t.a1];
            kbase += c0 + __popcll(w1);
        };
        // A lone wave (a 625-symbol shard leaves most SIMDs with one) issues a DEPENDENT instruction every ~32 clocks and an
        // independent one every ~8 (scripts/ubench/valu_issue.hip), and a tile is one dependency chain -- event count, table look-up,
        // equity, store --: four whole tiles at a time, written as four interleavable streams without a branch, when all three
        // columns are wanted on 16-byte aligned rows (or none: summary only).
        const int nfull = T / 128;
        int jj0 = 0;
        auto groups = [&](auto store) {
            for (; jj0 + 4 <= nfull; jj0 += 4) {
                const int jb = __builtin_amdgcn_readfirstlane(jj0);
                Tile t[4];
#pragma unroll
                for (int u = 0; u < 4; u++) open(jb + u, t[u]);
                double p0[4], p1[4], c_0[4], c_1[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { p0[u] = evr[t[u].idx0]; c_0[u] = evp[t[u].idx0]; p1[u] = evr[t[u].idx1]; c_1[u] = evp[t[u].idx1]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const double x0 = pq_isnull(t[u].x0) ? NANV : t[u].x0, x1 = pq_isnull(t[u].x1) ? NANV : t[u].x1; // null -> NaN (vectorized.rs:70-78)
                    const double e0 = c_0[u] + p0[u] * x0, e1 = c_1[u] + p1[u] * x1;
                    if constexpr (decltype(store)::value) {
                        nt_store2(a.position + base + t[u].i0, make_double2(p0[u], p1[u]));
                        nt_store2(a.cash + base + t[u].i0, make_double2(c_0[u], c_1[u]));
                        nt_store2(a.equity + base + t[u].i0, make_double2(e0, e1));
                    }
                    px[t[u].a0] = e0; px[t[u].a1] = e1;
                }
            }
        };
        if (wide && a.position && a.cash && a.equity) groups(std::true_type{});
        else if (!a.position && !a.cash && !a.equity) groups(std::false_type{});
        for (; jj0 < nb2; jj0++) { // the last, partial tiles; any other combination of outputs
            const int jj = __builtin_amdgcn_readfirstlane(jj0);
            Tile t;
            open(jj, t);
            const int i0 = t.i0;
            const double p0 = evr[t.idx0], c_0 = evp[t.idx0], p1 = evr[t.idx1], c_1 = evp[t.idx1];
            const double x0 = pq_isnull(t.x0) ? NANV : t.x0, x1 = pq_isnull(t.x1) ? NANV : t.x1;
            const double e0 = c_0 + p0 * x0, e1 = c_1 + p1 * x1;
            if (wide && i0 + 1 < T) {
                if (a.position) nt_store2(a.position + base + i0, make_double2(p0, p1));
                if (a.cash) nt_store2(a.cash + base + i0, make_double2(c_0, c_1));
                if (a.equity) nt_store2(a.equity + base + i0, make_double2(e0, e1));
            } else {
                if (i0 < T) {
                    if (a.position) __builtin_nontemporal_store(p0, &a.position[base + i0]);
                    if (a.cash) __builtin_nontemporal_store(c_0, &a.cash[base + i0]);
                    if (a.equity) __builtin_nontemporal_store(e0, &a.equity[base + i0]);
                }
                if (i0 + 1 < T) {
                    if (a.position) __builtin_nontemporal_store(p1, &a.position[base + i0 + 1]);
                    if (a.cash) __builtin_nontemporal_store(c_1, &a.cash[base + i0 + 1]);
                    if (a.equity) __builtin_nontemporal_store(e1, &a.equity[base + i0 + 1]);
                }
            }
            if (i0 < 64 * C) { px[t.a0] = e0; px[t.a1] = e1; }
        }
    }

    // ---- phases 3 + 4, block form (more events than the lists hold, or a row the fast chain could not decide): per 64-row
    // block, the block's events in order with the reference's operations, then its rows
    bool searching = false; // true after a buy failed: the precomputed events no longer hold, search the signal masks instead
    int cq = 0, cb = 0;     // search cursor: chunk / word, bit
    auto find = [&]() -> int { // next row >= cursor whose (valid-price) signal the pool can act on; -1 if none
        if constexpr (MACD) { // cursor: chunk cq (= lane), bit cb of its up to 64 * NW rows
            unsigned long long m[NW], any = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                m[k] = flat ? bmask.w[k] : smask.w[k];
                if (lane < cq) m[k] = 0;
                if (lane == cq) {
                    const int sh = cb - 64 * k;
                    if (sh >= 64) m[k] = 0;
                    else if (sh > 0) m[k] = (m[k] >> sh) << sh;
                }
                any |= m[k];
            }
            const unsigned long long bal = btw_ballot(any != 0);
            if (!bal) return -1;
            const int c1 = __builtin_ctzll(bal);
            int b1 = -1;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const unsigned long long mk = btw_readlane(m[k], c1);
                if (b1 < 0 && mk) b1 = 64 * k + __builtin_ctzll(mk);
            }
            cq = c1; cb = b1 + 1;
            return c1 * CQ + b1;
        } else { // cursor: block cq (word cq >> 6 of lane cq & 63), bit cb
#pragma unroll
            for (int k = 0; k < NW; k++) {
                unsigned long long m = flat ? bmask.w[k] : smask.w[k];
                const int blk = lane + 64 * k;
                if (blk < cq) m = 0;
                if (blk == cq) m = cb >= 64 ? 0 : (m >> cb) << cb;
                const unsigned long long bal = btw_ballot(m != 0);
                if (bal) {
                    const int c1 = __builtin_ctzll(bal);
                    const int b1 = __builtin_ctzll(btw_readlane(m, c1));
                    cq = c1 + 64 * k; cb = b1 + 1;
                    return cq * 64 + b1;
                }
            }
            return -1;
        }
    };
    int r = -1;
    const int nblk = dense ? 0 : (T + 63) / 64;
    for (int j0 = 0; j0 < nblk; j0++) {
        const int j = __builtin_amdgcn_readfirstlane(j0);
        const int i = 64 * j + lane;
        const int ai = addr(i);
        double x = px[ai];
        double tp = pos, tc = avail; // state after the last event at or before my row
        unsigned long long w = searching ? 0ULL : btw_word(myword, j);
        while (w) {
            const int l = __builtin_ctzll(w);
            if (!trade(btw_readlane(x, l))) {
                searching = true;
                const int nxt = 64 * j + l + 1;
                cq = MACD ? (int)(((unsigned)nxt * magic) >> 20) : nxt >> 6;
                cb = nxt - cq * CQ;
                r = find();
                break;
            }
            w &= w - 1;
            if (lane >= l) { tp = pos; tc = avail; }
        }
        if (searching) {
            const int rend = 64 * (j + 1);
            while (r >= 0 && r < rend) {
                const int l = r & 63;
                if (trade(btw_readlane(x, l)) && lane >= l) { tp = pos; tc = avail; }
                r = find();
            }
        }
        if (pq_isnull(x)) x = __longlong_as_double(0x7FF8000000000000LL); // null -> NaN (vectorized.rs:70-78)
        const double eq = tc + tp * x;
        if (i < T) {
            if (a.position) __builtin_nontemporal_store(tp, &a.position[base + i]);
            if (a.cash) __builtin_nontemporal_store(tc, &a.cash[base + i]);
            if (a.equity) __builtin_nontemporal_store(eq, &a.equity[base + i]);
        }
        px[ai] = eq;
    }
    btw_lds_fence();
    BTW_T(2);
    if constexpr (NWV > 1) __syncthreads(); // B3: the equity row is complete (dense: every wave's tiles; else the block form above)
    if (!a.summary) return;

    if constexpr (NWV > 1) btw_summary_mw<NWV>(geo, lane, 0, px, a.bench ? bm : nullptr, prm.initial_capital, trades, wins, a.summary + s * PQ_SUMMARY_COLS, shd);
    else btw_summary(geo, lane, px, a.bench ? bm : nullptr, prm.initial_capital, trades, wins, a.summary + s * PQ_SUMMARY_COLS);
    BTW_T(3);
}

// ---- the leveraged engine (SURVEY 8(f) rank 1 / decision D-10, oracle/backtest.c pqo_backtest_leveraged) in the same form: one
// symbol per wavefront, 64 rows per step.  What is serial here:
//   * the debt while a leveraged position is open compounds every row (debt += debt * rate / 252, a rounded product and a rounded
//     sum): a wave-uniform two-operation chain per row, from which every lane keeps the value of its own row;
//   * the events -- entries on a buy signal while flat, exits on the first row whose margin test fails (evaluated row-parallel
//     against the lanes' own debt values) or that carries a sell signal -- with the reference's own operations (lot search loop
//     included), once per event, wave-uniform.
// Everything else (validity, last valid price, the three daily columns, the summary) is row-parallel.
struct LevWaveArgs {
    LevArgs a;
    int32_t C, P;
    uint32_t magic;
};
__global__ __launch_bounds__(64) void lev_wave_kernel(LevWaveArgs w, Dims d) {
    extern __shared__ __align__(16) unsigned char btw_lds[];
    const LevArgs &a = w.a;
    const int lane = (int)threadIdx.x;
    const int64_t s = blockIdx.x;
    const int T = (int)dims_len(d, s);
    const int64_t base = dims_base(d, s);
    const BtwGeom geo{T, w.C, w.P, w.magic};
    double *px = reinterpret_cast<double *>(btw_lds); // the price row; then the total_value row; then its daily returns
    const pq_lev_params prm = a.prm;
    if (a.trade_count && T == 0 && lane == 0) a.trade_count[s] = 0;
    if (T == 0) {
        if (a.summary && lane < 8) a.summary[s * PQ_SUMMARY_COLS + lane] = 0.0;
        return;
    }
    (void)btw_stage(geo, lane, a.price + base, px, pq_null());
    btw_lds_fence();

    double cash = prm.initial_capital, debt = 0.0, shares = 0.0, last_px = 0.0, e_outlay = 0.0, e_price = 0.0;
    int e_day = 0, trades = 0, wins = 0;
    const double rc = prm.interest_rate / 252.0; // D-10 step 1: the daily rate is formed first
    const int64_t rb = s * (int64_t)a.max_trades;
    const int nblk = (T + 63) / 64;
    for (int j0 = 0; j0 < nblk; j0++) {
        const int j = __builtin_amdgcn_readfirstlane(j0);
        const int i = 64 * j + lane;
        const int ai = geo.addr(i);
        double x = px[ai];
        if (pq_isnull(x)) x = __longlong_as_double(0x7FF8000000000000LL);
        const bool in = i < T;
        const bool valid = in && !(isnan(x) || x <= 0.0);
        bool bb = false, sb = false;
        if (in) { bb = a.buy[base + i] != 0; sb = a.sell[base + i] != 0; }
        const unsigned long long vmask = btw_ballot(valid), bmask = btw_ballot(bb && valid), smask = btw_ballot(sb && valid);
        // the last valid price at or before my row (stock_value is marked to it)
        const unsigned long long below = vmask & ((2ULL << lane) - 1ULL);
        const int src = below ? 63 - __builtin_clzll(below) : lane;
        const double lpv = __shfl(x, src);
        const double lp = below ? lpv : last_px;
        // state after my row
        double tcash = cash, tdebt = debt, tshares = shares;
        int cur = 0;
        const int nrow = T - 64 * j < 64 ? T - 64 * j : 64;
        while (cur < nrow) {
            const unsigned long long from = ~0ULL << cur;
            if (shares == 0.0) { // flat: the next buy signal on a valid price
                const unsigned long long m = bmask & from;
                if (!m) break;
                const int r = __builtin_ctzll(m);
                const double p = btw_readlane(x, r);
                const double exec = p * (1.0 + prm.slippage);
                const double power = cash * prm.position_size * prm.leverage;
                double lots = floor(power / (exec * 100.0));
                double cost = 0.0, fee = 0.0;
                while (lots > 0.0) {
                    cost = lots * 100.0 * exec;
                    fee = fmax(cost * prm.commission_rate, prm.min_commission);
                    if (cost + fee <= cash * prm.leverage) break;
                    lots -= 1.0;
                }
                if (lots > 0.0) {
                    const double outlay = cost + fee;
                    debt = fmax(outlay - cash, 0.0);
                    cash = fmax(cash - outlay, 0.0);
                    shares = lots * 100.0;
                    e_outlay = outlay; e_price = exec; e_day = 64 * j + r;
                    if (lane >= r) { tcash = cash; tdebt = debt; tshares = shares; }
                }
                cur = r + 1;
            } else { // long: the debt compounds row by row; the first failed margin test or sell signal closes the position
                // (the position is closed on the first sell signal at the latest: the chain need not run past that row)
                const unsigned long long sfrom = smask & from;
                const int last = sfrom ? __builtin_ctzll(sfrom) : nrow - 1;
                double dl = tdebt; // my row's debt
                if (debt > 0.0) {
                    double dd = debt;
                    for (int t = cur; t <= last; t++) {
                        dd += dd * rc;
                        if (lane == t) dl = dd;
                    }
                }
                const double eqv = shares * x;
                const bool mc = valid && lane >= cur && lane <= last && dl > 0.0 && cash + eqv - dl < prm.margin_call_threshold * eqv;
                const unsigned long long mcm = btw_ballot(mc);
                const unsigned long long ex = (mcm | smask) & from;
                if (lane >= cur && lane <= last) tdebt = dl;
                if (!ex) { // still long at the end of the block
                    if (debt > 0.0) debt = btw_readlane(dl, nrow - 1);
                    break;
                }
                const int r = __builtin_ctzll(ex);
                const int why = (mcm >> r) & 1ULL ? 2 : 1;
                const double p = btw_readlane(x, r);
                const double dr = btw_readlane(dl, r); // the debt on the exit row (0 if none)
                const double exec = p * (1.0 - prm.slippage);
                const double rev = shares * exec;
                const double fee = fmax(rev * prm.commission_rate, prm.min_commission);
                const double net = rev - fee;
                const double gain = net - e_outlay;
                if (lane == 0 && trades < a.max_trades && a.entry_day) {
                    const int64_t q = rb + trades;
                    a.entry_day[q] = e_day; a.exit_day[q] = 64 * j + r;
                    a.entry_price[q] = e_price; a.exit_price[q] = exec; a.quantity[q] = shares;
                    a.pnl[q] = gain; a.pnl_pct[q] = gain / e_outlay * 100.0; a.reason[q] = why;
                }
                trades += 1;
                if (gain > 0.0) wins += 1;
                cash = cash + net - dr;
                debt = 0.0;
                shares = 0.0;
                if (lane >= r) { tcash = cash; tdebt = 0.0; tshares = 0.0; }
                cur = r + 1;
            }
        }
        if (vmask) last_px = btw_readlane(x, 63 - __builtin_clzll(vmask));
        const double cn = tcash - tdebt, sv = tshares * lp, tv = cn + sv;
        if (in) {
            __builtin_nontemporal_store(cn, &a.cash_net[base + i]);
            __builtin_nontemporal_store(sv, &a.stock_value[base + i]);
            __builtin_nontemporal_store(tv, &a.total_value[base + i]);
        }
        px[ai] = tv;
    }
    btw_lds_fence();
    if (a.trade_count && lane == 0) a.trade_count[s] = trades;
    if (a.summary) btw_summary(geo, lane, px, a.bench, prm.initial_capital, trades, wins, a.summary + s * PQ_SUMMARY_COLS, true);
}

// host side: shape of the wave form for a batch, or false when it does not apply (len > 64 * BTW_MAX_C)
static inline bool btw_plan(const pq_batch *b, int64_t fast, int64_t slow, int64_t sig, bool macd, BtWaveArgs &a, size_t &lds_bytes, bool bench) {
    if (b->len > 64 * BTW_MAX_C) return false;
    const int T = (int)b->len;
    int C = (T + 63) / 64;
    if (C < 1) C = 1;
    a.C = C;
    a.P = C | 1;
    a.magic = (uint32_t)(((1u << 20) + (unsigned)C - 1) / (unsigned)C);
    a.nW = 0;
    if (macd) {
        // rows until a restarted MACD state machine has merged bitwise with the true one (measured on the SURVEY 8d generator,
        // scripts/sim_macd_merge.py: median 32 / alpha, 99.9 % within 38 / alpha of the slowest average, the signal line ~10 /
        // alpha_sig behind).  A chunk that has not merged is re-run, so this is a speed knob only.
        const double af = 2.0 / ((double)(fast > slow ? fast : slow) + 1.0), ag = 2.0 / ((double)sig + 1.0);
        double rows = 36.0 / af + 8.0 / ag;
        if (!(rows < 1e9)) rows = 1e9;
        int64_t nW = ((int64_t)rows + C - 1) / C;
        if (const char *e = getenv("PQ_BT_WARM_CHUNKS")) nW = atoll(e); // tests: force failing chunks (re-run path)
        if (nW < 1) nW = 1;
        a.nW = (int32_t)(nW > 64 ? 64 : nW);
        // with seeds from the affine prefix scan (null-free series) only the last ulps have to merge.  Measured at 5 000 x 2 520,
        // MACD(12, 26, 9), of 315 000 chunks: 80 rows of warm-up leave 4 454 unmerged, 120 rows 195, 160 rows 13, 200 rows none;
        // a chunk that has not merged costs one re-run of its C rows, a warm-up chunk costs C rows on every wave.  Round 6 re-measured
        // the sweep on the four-wave form (PQ_BT_WARM_CHUNKS2 = 1 .. 5, us per call at 625 / 5 000 symbols): 126 / 566, 57.2 / 274, **49.8 /
        // 264.9**, 50.9 / 270.5, 52.0 / 276.7 -- 120 rows (3 chunks at this shape; round 4 planned 160) are the fastest setting at both sizes
        double rows2 = 6.4 / af + 6.4 / ag;
        if (!(rows2 < 1e9)) rows2 = 1e9;
        int64_t nW2 = ((int64_t)rows2 + C - 1) / C;
        if (const char *e = getenv("PQ_BT_WARM_CHUNKS2")) nW2 = atoll(e);
        if (nW2 < 1) nW2 = 1;
        a.nW2 = (int32_t)(nW2 > a.nW ? a.nW : nW2);
    }
    a.kcap = 8 * C < T + 1 ? 8 * C : T + 1; // ~12 % of the rows may be events (MACD(12,26,9): 8 %) before the block form takes over
    if (a.kcap > 64 * BTW_NG) a.kcap = 64 * BTW_NG;
    lds_bytes = (size_t)64 * a.P * 8 * (bench ? 2 : 1) + 64 * 8 * (C > 64 ? 2 : 1) + 2 * ((size_t)a.kcap + 1) * 8 + BTW_SH_DOUBLES * 8;
    return true;
}
