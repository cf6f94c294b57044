// ops_wt.h -- the indicators that run in the one-symbol-per-wavefront form (wt_dev.h), each a composition of its primitives.
// Every op states the lane-per-symbol op it must equal bit for bit (the gated fallback IS that op) and the reference lines.
#pragma once
#include "wt_dev.h"

// EMA / DEMA / TEMA / TRIX of one timeperiod (= EmaAllOp; overlap.rs:660-730, :543-598 (D-2), :1177-1311, momentum.rs:544-569).
// Any output pointer may be null.  Columns: A = x, then scratch; B = e0 = EMA(x).
struct WtEmaAllOp {
    static constexpr const char *NAME = "ema";
    static constexpr int NCOL = 2;
    const double *x;
    double *ema, *dema, *tema, *trix;
    int32_t p;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0), *B = w.col(1);
        const bool bad = wt_stage(w, x, A);
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        const WtRec<WT_EMA> rec = wt_ema(p);
        const BtwGeom g = w.g;
        wt_chain(w, A, B, rec, p, 0, WtInId{}, WtOutId{});                       // e0: first value at row p-1
        wt_store(w, ema, [&](int, int a) { return B[a]; });
        if (trix) { // EMA(z0(EMA(z0(e0)))): the zero-filled rows count (quirk Q-TRIX), so both levels are seeded at row p-1
            wt_chain(w, B, A, rec, p, 0, WtInZ0{}, WtOutId{});
            wt_chain(w, A, A, rec, p, 0, WtInZ0{}, WtOutId{});
            wt_store(w, trix, [&](int i, int a) {
                const double c3 = A[a], pv = i >= 1 ? A[g.addr(i - 1)] : pq_null();
                const double v = (c3 - pv) / pv * 100.0;                        // momentum.rs:563-566
                return (i >= 1 && !pq_isnull(c3) && !pq_isnull(pv) && pv != 0.0) ? v : pq_null();
            });
        }
        if (dema || tema) {
            wt_chain(w, B, A, rec, p, p - 1, WtInId{}, WtOutId{});               // e1 = EMA over the valid rows of e0: first value at row 2p-2
            const int d0 = 2 * p - 1;                                            // DEMA emits from count 2p on (overlap.rs:590-597)
            wt_store(w, dema, [&](int i, int a) { return i >= d0 ? 2.0 * B[a] - A[a] : pq_null(); });
            if (tema) {   // e2 over e1 (first value at row 3p-3), written as 3 e0 - 3 e1 + e2 over e1 itself (overlap.rs:1290-1305)
                wt_chain(w, A, A, rec, p, 2 * p - 2, WtInId{}, [&](int i, double e2) { const int a = g.addr(i); return 3.0 * B[a] - 3.0 * A[a] + e2; });
                wt_store(w, tema, [&](int, int a) { return A[a]; });
            }
        }
        return true;
    }
};

// EMA alone / TRIX alone: ONE LDS column (the chains run in place), i.e. seven resident waves per CU instead of three
struct WtEmaOp {
    static constexpr const char *NAME = "ema";
    static constexpr int NCOL = 1;
    const double *x;
    double *ema;
    int32_t p;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0);
        const bool bad = wt_stage(w, x, A);
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        wt_chain(w, A, A, wt_ema(p), p, 0, WtInId{}, WtOutId{});
        wt_store(w, ema, [&](int, int a) { return A[a]; });
        return true;
    }
};
struct WtTrixOp {
    static constexpr const char *NAME = "ema";
    static constexpr int NCOL = 1;
    const double *x;
    double *trix;
    int32_t p;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0);
        const bool bad = wt_stage(w, x, A);
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        const WtRec<WT_EMA> rec = wt_ema(p);
        const BtwGeom g = w.g;
        wt_chain(w, A, A, rec, p, 0, WtInId{}, WtOutId{});
        wt_chain(w, A, A, rec, p, 0, WtInZ0{}, WtOutId{});
        wt_chain(w, A, A, rec, p, 0, WtInZ0{}, WtOutId{});
        wt_store(w, trix, [&](int i, int a) {
            const double c3 = A[a], pv = i >= 1 ? A[g.addr(i - 1)] : pq_null();
            const double v = (c3 - pv) / pv * 100.0;                            // momentum.rs:563-566
            return (i >= 1 && !pq_isnull(c3) && !pq_isnull(pv) && pv != 0.0) ? v : pq_null();
        });
        return true;
    }
};

// MACD(fast, slow, sig) and a second signal period over the same two averages (= MacdPairOp when MACDFIX's fixed 12 / 26 are
// this call's fast / slow; momentum.rs:250-283, quirk Q-MACD; momentum.py:90-92).  Columns: A = x -> slow EMA -> signal; B = fast
// EMA -> dif.
struct WtMacdOp {
    static constexpr const char *NAME = "macd";
    static constexpr int NCOL = 2;
    const double *x;
    double *macd, *signal, *hist;      // signal period sig
    double *macd2, *signal2, *hist2;   // signal period sig2 (all null: absent)
    int32_t fast, slow, sig, sig2;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0), *B = w.col(1);
        const bool bad = wt_stage(w, x, A);
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        {   // the two averages of x in one set of walks (the slow one in place over x: it comes second)
            const WtChainSpec<WT_EMA> ch[2] = {{A, B, wt_ema(fast), fast, 0}, {A, A, wt_ema(slow), slow, 0}};
            wt_chains<2>(w, ch, WtInId{}, [](int, int, double e) { return e; });
        }
        wt_map(w, [&](int, int a) { const double f = B[a], s = A[a]; B[a] = (!pq_isnull(f) && !pq_isnull(s)) ? f - s : pq_null(); }); // dif
        btw_lds_fence();
        wt_store2(w, macd, macd2, [&](int, int a, double &u, double &v) { u = v = B[a]; });
        auto sig_line = [&](int ps, double *gs, double *gh) { // dea = EMA(dif with None -> 0.0, ps); hist where both are values
            wt_chain(w, B, A, wt_ema(ps), ps, 0, WtInZ0{}, WtOutId{});
            wt_store2(w, gs, gh, [&](int, int a, double &u, double &v) {
                const double dif = B[a], dea = A[a];
                u = dea;
                v = (!pq_isnull(dif) && !pq_isnull(dea)) ? dif - dea : pq_null();
            });
        };
        sig_line(sig, signal, hist);
        if (signal2 || hist2) {
            if (sig2 == sig) wt_store2(w, signal2, hist2, [&](int, int a, double &u, double &v) {
                const double dif = B[a], dea = A[a];
                u = dea;
                v = (!pq_isnull(dif) && !pq_isnull(dea)) ? dif - dea : pq_null();
            });
            else sig_line(sig2, signal2, hist2);
        }
        return true;
    }
};

// up / down moves of a close column (momentum.rs:513-524; row 0 = 0.0)
__device__ __forceinline__ void wt_updown(const double *x, int i, double &u, double &d, bool &bad) {
    const double cur = x[i];
    bad |= wt_bad(cur);
    u = 0.0; d = 0.0;
    if (i >= 1) {
        const double diff = cur - x[i - 1];
        u = (diff > 0.0) ? diff : 0.0;
        d = (diff > 0.0) ? 0.0 : -diff;
    }
}
// RSI (= RsiOp; momentum.rs:507-541, calc_rma D-1).  Columns: A = up -> rma(up), B = down -> rma(down).
struct WtRsiOp {
    static constexpr const char *NAME = "rsi";
    // recorded for a SMALL shard too where RSI heads a dependency chain (STOCHRSI = RSI -> fast-k -> MA, momentum.py:197-205; pq_ctx::chain_head)
    // and the shard is at most 12 tiles: 625 symbols run this form in 0.1 - 0.15 ms against 0.55 - 0.85 ms for the lone-wave job the chain
    // would wait for (1 250 symbols: 0.2 - 0.65 ms, no better than the job)
    static constexpr bool SMALL_SUITE = true;
    static constexpr int NCOL = 2;
    const double *x;
    double *rsi;
    int32_t p;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0), *B = w.col(1);
        bool bad = false;
        const double *xs = x + w.base;
        double *const dst[2] = {A, B};
        wt_stage_fn<2>(w, dst, [&](int i, double (&v)[2]) { wt_updown(xs, i, v[0], v[1], bad); });
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        const WtRec<WT_RMA> rec = wt_rma(p);
        {
            const WtChainSpec<WT_RMA> ch[2] = {{A, A, rec, p, 0}, {B, B, rec, p, 0}};
            wt_chains<2>(w, ch, WtInId{}, [](int, int, double e) { return e; });
        }
        wt_store(w, rsi, [&](int, int a) {
            const double up = A[a], dn = B[a];
            const double rs = up / dn;
            const double v = 100.0 - (100.0 / (1.0 + rs));                       // momentum.rs:536-538
            return (pq_isnull(up) || pq_isnull(dn)) ? pq_null() : ((dn == 0.0) ? 100.0 : v);
        });
        return true;
    }
};

// +DM / -DM / TR of row i (momentum.rs:676-699; row 0 = 0.0)
__device__ __forceinline__ void wt_dm_row(const double *h, const double *l, const double *c, int i, double &pdm, double &mdm, double &tr, bool &bad) {
    const double hi = h[i], lo = l[i];
    bad |= wt_bad(hi) | wt_bad(lo);
    if (c) bad |= wt_bad(c[i]);
    pdm = 0.0; mdm = 0.0; tr = 0.0;
    if (i >= 1) {
        const double up_move = hi - h[i - 1], down_move = l[i - 1] - lo;
        pdm = (up_move > down_move && up_move > 0.0) ? up_move : 0.0;
        mdm = (down_move > up_move && down_move > 0.0) ? down_move : 0.0;
        if (c) { const double pc = c[i - 1]; tr = fmax(fmax(hi - lo, fabs(hi - pc)), fabs(lo - pc)); }
    }
}
// PLUS_DM / MINUS_DM (= DmPairOp; momentum.rs:414-436, :359-381).  Columns: A = +DM -> rma, B = -DM -> rma.
struct WtDmPairOp {
    static constexpr const char *NAME = "dmpair";
    static constexpr int NCOL = 2;
    const double *h, *l;
    double *plus_dm, *minus_dm;
    int32_t p;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0), *B = w.col(1);
        bool bad = false;
        const double *hs = h + w.base, *ls = l + w.base;
        double *const dst[2] = {A, B};
        wt_stage_fn<2>(w, dst, [&](int i, double (&v)[2]) { double tr; wt_dm_row(hs, ls, nullptr, i, v[0], v[1], tr, bad); });
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        const WtRec<WT_RMA> rec = wt_rma(p);
        {
            const WtChainSpec<WT_RMA> ch[2] = {{A, A, rec, p, 0}, {B, B, rec, p, 0}};
            wt_chains<2>(w, ch, WtInId{}, [](int, int, double e) { return e; });
        }
        wt_store2(w, plus_dm, minus_dm, [&](int, int a, double &u, double &v) { u = A[a]; v = B[a]; });
        return true;
    }
};
template <bool PLUS> // PLUS_DM or MINUS_DM alone: one column
struct WtDmRawOp {
    static constexpr const char *NAME = "dmpair";
    static constexpr int NCOL = 1;
    const double *h, *l;
    double *out;
    int32_t p;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0);
        bool bad = false;
        const double *hs = h + w.base, *ls = l + w.base;
        double *const dst[1] = {A};
        wt_stage_fn<1>(w, dst, [&](int i, double (&v)[1]) { double pd, md, tr; wt_dm_row(hs, ls, nullptr, i, pd, md, tr, bad); v[0] = PLUS ? pd : md; });
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        wt_chain(w, A, A, wt_rma(p), p, 0, WtInId{}, WtOutId{});
        wt_store(w, out, [&](int, int a) { return A[a]; });
        return true;
    }
};
// calc_dm and its five users: DX, PLUS_DI (= DX, quirk Q-PDI / D-5), MINUS_DI, ADX, ADXR (= DmAllOp<true>; momentum.rs:668-727,
// :11-61).  Columns: A = +DM -> its rma -> z0(dx) -> adx; B = -DM -> rma; C = TR -> rma.
struct WtDmiOp {
    static constexpr const char *NAME = "dmi";
    static constexpr int NCOL = 3;
    const double *h, *l, *c;
    double *dx, *plus_di, *minus_di, *adx, *adxr;
    int32_t p;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0), *B = w.col(1), *Cc = w.col(2);
        bool bad = false;
        const double *hs = h + w.base, *ls = l + w.base, *cs = c + w.base;
        double *const dst[3] = {A, B, Cc};
        wt_stage_fn<3>(w, dst, [&](int i, double (&v)[3]) { wt_dm_row(hs, ls, cs, i, v[0], v[1], v[2], bad); });
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        const WtRec<WT_RMA> rec = wt_rma(p);
        const BtwGeom g = w.g;
        {   // the three Wilder averages in one set of walks
            const WtChainSpec<WT_RMA> ch[3] = {{A, A, rec, p, 0}, {B, B, rec, p, 0}, {Cc, Cc, rec, p, 0}};
            wt_chains<3>(w, ch, WtInId{}, [](int, int, double e) { return e; });
        }
        // momentum.rs:700-724 row by row: DI from the three averages, DX from the DIs; the ADX average is fed dx with None -> 0.0
        auto di_row = [&](int a, double &dxv, double &mdi) {
            const double sp = A[a], sm = B[a], st = Cc[a];
            const double pdi = 100.0 * sp / st, mdi_ = 100.0 * sm / st;
            const double diff = fabs(pdi - mdi_), sum = pdi + mdi_;
            const double dxr = 100.0 * diff / sum;
            const bool ok = !pq_isnull(sp) && !pq_isnull(sm) && !pq_isnull(st) && st != 0.0;
            mdi = ok ? mdi_ : pq_null();
            dxv = ok ? ((sum == 0.0) ? 0.0 : dxr) : pq_null();
        };
        wt_store2(w, dx, plus_di, [&](int, int a, double &u, double &v) { double m; di_row(a, u, m); v = u; });
        wt_store(w, minus_di, [&](int, int a) { double d, m; di_row(a, d, m); return m; });
        if (adx || adxr) {
            wt_map(w, [&](int, int a) { double d, m; di_row(a, d, m); A[a] = pq_isnull(d) ? 0.0 : d; });
            btw_lds_fence();
            wt_chain(w, A, A, rec, p, 0, WtInId{}, WtOutId{});
            wt_store2(w, adx, adxr, [&](int i, int a, double &u, double &v) {
                u = A[a];
                const double pv = i >= p - 1 ? A[g.addr(i - (p - 1))] : pq_null(); // momentum.rs:50-59
                v = (i >= p - 1 && !pq_isnull(u) && !pq_isnull(pv)) ? (u + pv) * 0.5 : pq_null();
            });
        }
        return true;
    }
};
// ATR / NATR (= AtrAllOp; volatility.rs:18-48: calc_ema(trange, 2p-1), trange null on row 0).  Column: A = TR -> its EMA.
struct WtAtrOp {
    static constexpr const char *NAME = "atr";
    static constexpr int NCOL = 1;
    const double *h, *l, *c;
    double *atr, *natr;
    int32_t p;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0);
        bool bad = false;
        const double *hs = h + w.base, *ls = l + w.base, *cs = c + w.base;
        double *const dst[1] = {A};
        wt_stage_fn<1>(w, dst, [&](int i, double (&v)[1]) {
            const double hi = hs[i], lo = ls[i], cl = cs[i];
            bad |= wt_bad(hi) | wt_bad(lo) | wt_bad(cl);
            v[0] = i >= 1 ? true_range(hi, lo, cs[i - 1]) : pq_null();
        });
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        const int n = 2 * p - 1;
        wt_chain(w, A, A, wt_ema(n), n, 1, WtInId{}, WtOutId{});
        // close comes from global memory again (L2): a second LDS column would cost a third of the resident waves (measured:
        // 0.35 ms with close staged against 0.20 ms at 5 000 x 2 520)
        wt_store2(w, atr, natr, [&](int i, int a, double &u, double &v) {
            u = A[a];
            v = pq_isnull(u) ? pq_null() : u / cs[i] * 100.0;                     // volatility.rs:44
        });
        return true;
    }
};

// MIDPOINT (= MidpointOp; overlap.rs:180-278 with quirk Q-MID: the min deque never expires => cumulative min; values from row 0):
// extrema are exact in any order.  Columns: A = x, B = cumulative min.
struct WtMidpointOp {
    static constexpr const char *NAME = "midpoint";
    static constexpr int NCOL = 2;
    const double *x;
    double *out;
    int32_t p;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0), *B = w.col(1);
        const bool bad = wt_stage(w, x, A);
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        const int C = w.g.C, P = w.g.P, c = w.lane, T = w.g.T;
        const double INF = __longlong_as_double(0x7FF0000000000000LL);
        const int lo = c * C, nrow = lo + C <= T ? C : (T > lo ? T - lo : 0);
        double m = INF;
        for (int b = 0; b < nrow; b++) m = fmin(m, A[c * P + b]);
        double M = m; // inclusive prefix min over the lanes
#define WT_MIN(CTRL, RM) { M = fmin(M, btw_dpp<CTRL, RM>(INF, M)); }
        BTW_SCAN_STEPS(WT_MIN)
#undef WT_MIN
        double run = btw_prev_lane(INF, M);
        for (int b = 0; b < nrow; b++) { run = fmin(run, A[c * P + b]); B[c * P + b] = run; }
        btw_lds_fence();
        const BtwGeom g = w.g;
        const int pp = p;
        wt_store(w, out, [&](int i, int a) {
            double mx = A[a];
            const int k1 = i + 1 < pp ? i + 1 : pp; // the window holds the last min(p, i + 1) values
            for (int k = 1; k < k1; k++) mx = fmax(mx, A[g.addr(i - k)]);
            return (mx + B[a]) / 2.0;
        });
        return true;
    }
};
// MIDPRICE (= MidpriceOp; overlap.rs:281-404, no-bitmap branches).  Columns: A = high, B = low.
struct WtMidpriceOp {
    static constexpr const char *NAME = "midprice";
    static constexpr int NCOL = 2;
    const double *h, *l;
    double *out;
    int32_t p;
    __device__ bool run(WtCtx &w) const {
        double *A = w.col(0), *B = w.col(1);
        const bool bad = wt_stage(w, h, A) | wt_stage(w, l, B);
        if (btw_ballot(bad)) return false;
        btw_lds_fence();
        const BtwGeom g = w.g;
        const int pp = p;
        wt_store(w, out, [&](int i, int a) {
            double mx = A[a], mn = B[a];
            const int k1 = i + 1 < pp ? i + 1 : pp;
            for (int k = 1; k < k1; k++) { const int ak = g.addr(i - k); mx = fmax(mx, A[ak]); mn = fmin(mn, B[ak]); }
            return (mx + mn) / 2.0;
        });
        return true;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// host side: launch a wave-per-symbol op with the lane-per-symbol op of the same function gated behind it
void *rec_alloc_zero(pq_ctx *ctx, size_t bytes); // suite.hip: zeroed device memory owned by the suite being recorded

static inline bool wt_on() { return getenv("PQ_NO_WT") == nullptr; } // A/B runs and tests: PQ_NO_WT=1 keeps the lane-per-symbol bodies
// A/B runs: PQ_WT_OPS=atr,midpoint,... keeps the wave form for the listed ops only (names: WtOp::NAME)
static inline bool wt_op_on(const char *name) {
    const char *e = getenv("PQ_WT_OPS");
    if (!e) return true;
    const size_t n = strlen(name);
    for (const char *q = e; (q = strstr(q, name)) != nullptr; q += n)
        if ((q == e || q[-1] == ',') && (q[n] == 0 || q[n] == ',')) return true;
    return false;
}
static inline double wt_warm() { const char *e = getenv("PQ_WT_WARM"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 10.0; }

template <class WtOp, class SeqOp>
struct WtBlob {
    WtOp wop;
    WtArgs a;
    SeqOp sop;
    InCols<SeqOp::NIN> in;
    OutCols<SeqOp::NOUT> out;
    pq_batch b;
    unsigned lds_wt, lds_seq;
    int gather; // the gated general path is the per-lane gather body (ragged batches)
};
template <class Op, class = void>
struct WtSmallSuite { static constexpr bool value = false; };
template <class Op>
struct WtSmallSuite<Op, decltype((void)Op::SMALL_SUITE)> { static constexpr bool value = Op::SMALL_SUITE; };
template <class WtOp, class SeqOp>
static void wt_launch_blob(const void *blob, hipStream_t stream) {
    const WtBlob<WtOp, SeqOp> &w = *reinterpret_cast<const WtBlob<WtOp, SeqOp> *>(blob);
    if (w.lds_wt > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&wt_kernel<WtOp>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return;
    hipLaunchKernelGGL(wt_kernel<WtOp>, dim3((unsigned)w.b.n_series), dim3(64), w.lds_wt, stream, w.wop, w.a, dims_of(&w.b));
    // the general path, gated: a workgroup whose tile is not flagged returns at once
    const dim3 tiles((unsigned)((w.b.n_series + SEQ_BLOCK - 1) / SEQ_BLOCK));
    if constexpr (!IsLdsOnly<SeqOp>::value) {
        if (w.gather) {
            hipLaunchKernelGGL((seq_kernel<SeqOp, false>), tiles, dim3(SEQ_BLOCK), 0, stream, w.sop, w.in, w.out, dims_of(&w.b), w.a.gate);
            return;
        }
    }
    hipLaunchKernelGGL((seq_kernel<SeqOp, true>), tiles, dim3(SEQ_LDS_BLOCK), w.lds_seq, stream, w.sop, w.in, w.out, dims_of(&w.b), w.a.gate);
}
// true: handled (launched or recorded; *st holds the status).  false: outside the wave form's scope -- the caller takes its usual path.
template <class WtOp, class SeqOp>
static inline bool wt_try(pq_ctx *ctx, const pq_batch *b, const WtOp &wop, const SeqOp &sop, const InCols<SeqOp::NIN> &in,
                          const OutCols<SeqOp::NOUT> &out, pq_status *st) {
    // Inside a recorded suite the lane-per-symbol jobs win: many functions over thousands of symbols fill the chip by themselves, and a
    // wave-per-symbol job holds 21 KB of LDS per column and symbol (measured at 5 000 x 2 520: the step takes 6.6 ms with these
    // kernels recorded in place of their jobs against 4.0 ms).  PQ_WT_SUITE=1 records them anyway (A/B runs).
    if (ctx->rec && !getenv("PQ_WT_SUITE") &&
        !(ctx->rec_small && ctx->chain_head && WtSmallSuite<WtOp>::value && b->n_series <= (getenv("PQ_WT_SMALL_TILES") ? atoll(getenv("PQ_WT_SMALL_TILES")) : 12) * SEQ_BLOCK &&
          !getenv("PQ_NO_WT_SMALL")))
        return false;
    if (!wt_on() || !wt_op_on(WtOp::NAME) || b->len > WT_MAX_LEN || b->n_series <= 0 || b->n_series > 0x7fffffffLL) return false;
    const bool ragged = b->offsets != nullptr;
    if (ragged) {
        // RAGGED batches (the groups of `.over("symbol")`: pq_batch.offsets) would otherwise run the per-lane gather body, the slowest
        // shape of the library; a wavefront per group takes any length and any 8-byte aligned start.  Worth it when the groups are
        // long on average (a wave costs ~250 serial steps per chain whatever its group's length); the gated general path is the gather
        // body, so the function must have one (the fused multi-output forms do not: their entry points take ragged batches apart).
        if (IsLdsOnly<SeqOp>::value || b->stride / b->n_series < WT_MIN_LEN) return false; // (stride = the total row count of a ragged batch)
        for (int k = 0; k < SeqOp::NIN; k++) if (reinterpret_cast<uintptr_t>(in.p[k]) % 8) return false;
        for (int k = 0; k < SeqOp::NOUT; k++) if (reinterpret_cast<uintptr_t>(out.p[k]) % 8) return false;
    } else {
        if (b->len < WT_MIN_LEN) return false;
        if (seq_lds_bytes(sop) > SEQ_LDS_LIMIT || seq_cols_tiling<SeqOp::NIN, SeqOp::NOUT>(b, in.p, out.p) != 0) return false; // the gated general path is the (16-byte) tiled body
    }
    WtBlob<WtOp, SeqOp> w{};
    const int T = (int)b->len;
    w.a.C = (T + 63) / 64;
    w.a.P = w.a.C | 1;
    w.a.magic = (uint32_t)(((1u << 20) + (unsigned)w.a.C - 1) / (unsigned)w.a.C);
    w.a.warm = wt_warm();
    w.a.stats = reinterpret_cast<unsigned long long *>(ctx->d_flag) + 24;
    w.lds_wt = (unsigned)((size_t)WtOp::NCOL * 64 * w.a.P * 8);
    if (w.lds_wt > 160 * 1024) return false;
    w.lds_seq = (unsigned)seq_lds_bytes(sop);
    w.gather = ragged ? 1 : 0;
    w.wop = wop; w.sop = sop; w.in = in; w.out = out; w.b = *b;
    const size_t tiles = (size_t)((b->n_series + SEQ_BLOCK - 1) / SEQ_BLOCK);
    *st = PQ_OK;
    if (ctx->rec) {
        static_assert(sizeof(WtBlob<WtOp, SeqOp>) <= sizeof(RowThunk::blob), "wave-per-symbol blob too large for a recorded launch");
        w.a.gate = reinterpret_cast<unsigned *>(rec_alloc_zero(ctx, tiles * sizeof(unsigned)));
        if (!w.a.gate) { pq_set_error("out of device memory for a gate"); *st = PQ_ERR_NOMEM; return true; }
        RowThunk t{};
        t.launch = &wt_launch_blob<WtOp, SeqOp>;
        t.row_id = 0;
        t.long_launch = 1;
        t.blob_bytes = (int)sizeof w;
        t.dims = dims_of(b);
        memcpy(t.blob, &w, sizeof w);
        for (int k = 0; k < SeqOp::NIN; k++) t.reads[t.n_reads++] = in.p[k];
        for (int k = 0; k < SeqOp::NOUT; k++) if (out.p[k]) t.writes[t.n_writes++] = out.p[k];
        *st = rec_add_row(ctx, t);
        return true;
    }
    if (ctx->wt_gate_tiles < tiles) {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { pq_set_error("hipStreamSynchronize failed"); *st = PQ_ERR_HIP; return true; }
        if (ctx->wt_gate) (void)hipFree(ctx->wt_gate);
        ctx->wt_gate = nullptr; ctx->wt_gate_tiles = 0;
        if (hipMalloc((void **)&ctx->wt_gate, tiles * sizeof(unsigned)) != hipSuccess || hipMemset(ctx->wt_gate, 0, tiles * sizeof(unsigned)) != hipSuccess) {
            pq_set_error("out of device memory for a gate"); *st = PQ_ERR_NOMEM; return true;
        }
        ctx->wt_gate_tiles = tiles;
    }
    w.a.gate = ctx->wt_gate;
    wt_launch_blob<WtOp, SeqOp>(&w, ctx->stream);
    if (hipGetLastError() != hipSuccess) { pq_set_error("wave-per-symbol launch failed"); *st = PQ_ERR_HIP; }
    return true;
}
