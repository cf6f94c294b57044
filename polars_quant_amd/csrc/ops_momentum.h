// momentum.hip -- kernels + C ABI for the momentum indicators (reference: src/talib/momentum.rs and
// the pure-Python composites of python/polars_quant/talib/momentum.py).
// momentum.rs functions are N-B (nulls rejected, momentum.rs:12-13): inputs are assumed null-free
// here; the host layer calls pq_count_nulls first and raises like the reference does.
#pragma once
#include "pq_cores.h"

__device__ __forceinline__ double z0(double x) { return pq_isnull(x) ? 0.0 : x; } // .unwrap_or(0.0)
#define F64_MAX 1.7976931348623157e308

// ---------------------------------------------------------------- ROW ops (exact, order-free)
template <int KIND> // 0 mom, 1 roc, 2 rocp, 3 rocr, 4 rocr100   (momentum.rs:384-397, :439-504)
struct LagOp {
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int ROW_ID = 8 + KIND;
    typedef double OutT;
    int64_t p;
    __device__ void eval(const Row<1> &r, int64_t t, double (&y)[1]) {
        y[0] = pq_null();
        if (p < 0 || t < p) return;
        double c = r.in[0][t], pr = r.in[0][t - p];
        if (KIND == 0) { y[0] = c - pr; return; }
        if (pr == 0.0) return;
        if (KIND == 1) y[0] = (c - pr) / pr * 100.0;
        else if (KIND == 2) y[0] = (c - pr) / pr;
        else if (KIND == 3) y[0] = c / pr;
        else y[0] = (c / pr) * 100.0;
    }
};
// README.md:46-75 returns(df, price_col, period, method) (README-only; decision D-13): METHOD 0 simple = (p[t] - p[t-period]) /
// p[t-period], 1 log = ln(p[t] / p[t-period]); null for t < period and where either price is null; IEEE-754 on a zero price
template <int METHOD>
struct ReturnsOp {
    static constexpr int NIN = 1, NOUT = 1;
    typedef double OutT;
    int64_t p;
    __device__ void eval(const Row<1> &r, int64_t t, double (&y)[1]) {
        y[0] = pq_null();
        if (p <= 0 || t < p) return;
        const double c = r.in[0][t], pr = r.in[0][t - p];
        if (pq_isnull(c) || pq_isnull(pr)) return;
        y[0] = METHOD == 0 ? (c - pr) / pr : log(c / pr);
    }
};
// rolling maximum / minimum over the last p rows (the Polars rolling_max / rolling_min frame of momentum.py:181-183: null until
// the frame holds p non-null rows) -- the channel bounds of the README's breakout (Donchian) strategy
template <bool IS_MAX>
struct RollingExtOp {
    static constexpr int NIN = 1, NOUT = 1;
    typedef double OutT;
    int64_t p;
    __device__ void eval(const Row<1> &r, int64_t t, double (&y)[1]) {
        y[0] = pq_null();
        if (p <= 0 || t < p - 1) return;
        double best = __longlong_as_double(0x7FF8000000000000LL); // decision D-14: NaN values are ignored; a frame of NaNs gives NaN
        for (int64_t j = t + 1 - p; j <= t; j++) {
            const double v = r.in[0][j];
            if (pq_isnull(v)) return;
            if (v != v) continue;
            if (best != best || (IS_MAX ? v > best : v < best)) best = v;
        }
        y[0] = best;
    }
};
struct BopOp { // momentum.rs:113-135
    static constexpr int NIN = 4, NOUT = 1;
    static constexpr int ROW_ID = 13;
    typedef double OutT;
    __device__ void eval(const Row<4> &r, int64_t t, double (&y)[1]) {
        double diff = r.in[1][t] - r.in[2][t];
        y[0] = (diff == 0.0) ? 0.0 : (r.in[3][t] - r.in[0][t]) / diff;
    }
};
template <int MODE> // 0: (up, down)  1: up - down (AROONOSC, decision D-6)  2: (up, down, up - down) in one scan   momentum.rs:70-110
struct AroonOp {
    static constexpr int NIN = 2, NOUT = (MODE == 0 ? 2 : (MODE == 1 ? 1 : 3));
    static constexpr int ROW_ID = 14 + MODE;
    typedef double OutT;
    int64_t p;
    __device__ void eval(const Row<2> &r, int64_t t, double (&y)[NOUT]) {
#pragma unroll
        for (int k = 0; k < NOUT; k++) y[k] = pq_null();
        if (p < 0 || t < p) return;
        int64_t start = t - p, max_idx = 0, min_idx = 0;
        double max_val = -F64_MAX, min_val = F64_MAX;
        for (int64_t j = start; j <= t; j++) {
            double h = r.in[0][j], l = r.in[1][j];
            if (h >= max_val) { max_val = h; max_idx = j - start; }
            if (l <= min_val) { min_val = l; min_idx = j - start; }
        }
        double up = ((double)max_idx / (double)p) * 100.0, dn = ((double)min_idx / (double)p) * 100.0;
        if (MODE == 0) { y[0] = up; y[NOUT - 1] = dn; }
        else if (MODE == 1) y[0] = up - dn;
        else { y[0] = up; y[NOUT > 1 ? 1 : 0] = dn; y[NOUT - 1] = up - dn; }
    }
};
struct WillrOp { // momentum.rs:630-662
    static constexpr int NIN = 3, NOUT = 1;
    static constexpr int ROW_ID = 17;
    typedef double OutT;
    int64_t p;
    __device__ void eval(const Row<3> &r, int64_t t, double (&y)[1]) {
        y[0] = pq_null();
        if (p <= 0 || t < p - 1) return;
        double max_h = -F64_MAX, min_l = F64_MAX;
        for (int64_t j = t + 1 - p; j <= t; j++) { max_h = fmax(max_h, r.in[0][j]); min_l = fmin(min_l, r.in[1][j]); }
        double diff = max_h - min_l;
        y[0] = (diff == 0.0) ? 0.0 : -100.0 * (max_h - r.in[2][t]) / diff;
    }
};
struct CciDevOp { // momentum.rs:161-176: mean-abs-deviation pass given sma(tp)
    static constexpr int NIN = 4, NOUT = 1; // high, low, close, sma_tp
    typedef double OutT;
    int64_t p;
    __device__ void eval(const Row<4> &r, int64_t t, double (&y)[1]) {
        y[0] = pq_null();
        if (p <= 0 || t < p - 1) return;
        double avg = r.in[3][t];
        if (pq_isnull(avg)) return;
        double mean_dev = 0.0;
        for (int64_t j = t + 1 - p; j <= t; j++) {
            double tp = (r.in[0][j] + r.in[1][j] + r.in[2][j]) / 3.0;
            mean_dev += fabs(tp - avg);
        }
        if (mean_dev != 0.0) {
            mean_dev /= (double)p;
            double tp = (r.in[0][t] + r.in[1][t] + r.in[2][t]) / 3.0;
            y[0] = (tp - avg) / (0.015 * mean_dev);
        }
    }
};
struct AdxrOp { // momentum.rs:50-59
    static constexpr int NIN = 1, NOUT = 1;
    typedef double OutT;
    int64_t p;
    __device__ void eval(const Row<1> &r, int64_t t, double (&y)[1]) {
        y[0] = pq_null();
        if (p <= 0 || t < p - 1) return;
        double curr = r.in[0][t], prev = r.in[0][t - (p - 1)];
        if (!pq_isnull(curr) && !pq_isnull(prev)) y[0] = (curr + prev) * 0.5;
    }
};
// Polars rolling_min/max(window) + fastk (momentum.py:181-183): null until the frame holds `k` non-null rows
struct FastkOp {
    static constexpr int NIN = 3, NOUT = 1; // high, low, close
    typedef double OutT;
    int64_t k;
    __device__ void eval(const Row<3> &r, int64_t t, double (&y)[1]) {
        y[0] = pq_null();
        if (k <= 0 || t < k - 1) return;
        double c = r.in[2][t];
        if (pq_isnull(c)) return;
        double hn = __longlong_as_double(0x7FF8000000000000LL), ln = hn; // decision D-14: NaN values are ignored (a frame of NaNs: NaN)
        for (int64_t j = t + 1 - k; j <= t; j++) {
            double h = r.in[0][j], l = r.in[1][j];
            if (pq_isnull(h) || pq_isnull(l)) return;
            if (h == h && (hn != hn || h > hn)) hn = h;
            if (l == l && (ln != ln || l < ln)) ln = l;
        }
        y[0] = (c - ln) * 100.0 / (hn - ln);
    }
};
template <int KIND> // 0: a-b   1: (a-b)/b*100 (null if b == 0)   -- null if either side null
struct BinOp {
    static constexpr int NIN = 2, NOUT = 1;
    typedef double OutT;
    __device__ void eval(const Row<2> &r, int64_t t, double (&y)[1]) {
        double a = r.in[0][t], b = r.in[1][t];
        if (pq_isnull(a) || pq_isnull(b)) { y[0] = pq_null(); return; }
        if (KIND == 0) y[0] = a - b;
        else y[0] = (b == 0.0) ? pq_null() : (a - b) / b * 100.0;
    }
};

// ---------------------------------------------------------------- SEQ ops
struct CmoOp { // momentum.rs:181-223: rolling SUMS of up/down moves; the lagged terms are recomputed
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 20;
    static constexpr int COST_NS = 220;
    static constexpr int NTAP = 2;
    static constexpr int TAP_COL[2] = {0, 0};
    int64_t p;
    double su, sd, prev;
    __device__ void init(const Row<1> &) { su = sd = 0.0; prev = 0.0; }
    __device__ void tap_lags(int64_t (&lag)[2]) const { lag[0] = p > 0 ? p : 0; lag[1] = p > 0 ? p + 1 : 0; }
    __device__ static void updown(double curr, double prv, double &u, double &d) {
        double diff = curr - prv;
        u = 0.0; d = 0.0;
        if (diff > 0.0) u = diff; else d = -diff;
    }
    __device__ void step(const Row<1> &, int64_t i, const double (&x)[1], const double (&tp)[2], double (&y)[1]) {
        y[0] = pq_null();
        if (p <= 0) return;
        double u = 0.0, d = 0.0;
        if (i >= 1) updown(x[0], prev, u, d);
        prev = x[0];
        su += u; sd += d;
        if (i >= p) {
            double ou = 0.0, od = 0.0;
            if (i - p >= 1) updown(tp[0], tp[1], ou, od);
            su -= ou; sd -= od;
        }
        if (i >= p - 1) {
            double total = su + sd;
            y[0] = (total == 0.0) ? 0.0 : 100.0 * (su - sd) / total;
        }
    }
    Ring w; // last p+1 inputs
    __host__ __device__ int64_t ring_slots() const { return p > 0 ? p + 1 : 1; }
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { init(r); w = ra.make(p + 1); }
    __device__ void step_lds(int64_t i, const double (&x)[1], double (&y)[1]) {
        y[0] = pq_null();
        if (p <= 0) return;
        double u = 0.0, d = 0.0;
        if (i >= 1) updown(x[0], prev, u, d);
        prev = x[0];
        su += u; sd += d;
        if (i >= p) {
            double ou = 0.0, od = 0.0;
            if (i - p >= 1) updown(w.get((int)p), w.get((int)p + 1), ou, od);
            su -= ou; sd -= od;
        }
        w.push(x[0]);
        if (i >= p - 1) {
            double total = su + sd;
            y[0] = (total == 0.0) ? 0.0 : 100.0 * (su - sd) / total;
        }
    }
    static constexpr bool FAST_NULL_OK = true; // N-B: no null handling in the row body
    __device__ static void updown_sel(double curr, double prv, double &u, double &d) { // updown() with selects
        const double diff = curr - prv;
        u = (diff > 0.0) ? diff : 0.0;
        d = (diff > 0.0) ? 0.0 : -diff;
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return p > 0 && t0 >= p + 2; } // the lagged pair is rows >= 1
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) {
        double u, d, ou, od;
        updown_sel(x[0], prev, u, d);
        prev = x[0];
        su += u; sd += d;
        updown_sel(w.get((int)p), w.get((int)p + 1), ou, od);
        su -= ou; sd -= od;
        w.push(x[0]);
        const double total = su + sd;
        y[0] = (total == 0.0) ? 0.0 : 100.0 * (su - sd) / total;
    }
};

struct RsiOp { // momentum.rs:507-541
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 21;
    static constexpr int COST_NS = 250;
    int64_t p;
    RmaCore au, ad;
    double prev;
    __device__ void init(const Row<1> &r) { au.init(p, r.len); ad.init(p, r.len); prev = 0.0; }
    __device__ void step(const Row<1> &, int64_t i, const double (&x)[1], double (&y)[1]) {
        double u = 0.0, d = 0.0;
        if (i >= 1) CmoOp::updown(x[0], prev, u, d);
        prev = x[0];
        double a = au.step(i, u), b = ad.step(i, d);
        if (pq_isnull(a) || pq_isnull(b)) { y[0] = pq_null(); return; }
        if (b == 0.0) y[0] = 100.0;
        else { double rs = a / b; y[0] = 100.0 - (100.0 / (1.0 + rs)); }
    }
    static constexpr bool FAST_NULL_OK = true; // N-B
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return t0 >= 1 && au.steady(t0) && ad.steady(t0); }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) {
        double u, d;
        CmoOp::updown_sel(x[0], prev, u, d);
        prev = x[0];
        const double a = au.fast(u), b = ad.fast(d);
        const double rs = a / b;
        const double v = 100.0 - (100.0 / (1.0 + rs));
        y[0] = (b == 0.0) ? 100.0 : v;
    }
};

struct MacdOp { // momentum.rs:250-283 (quirk Q-MACD: signal = EMA(dif with None -> 0.0))
    static constexpr int NIN = 1, NOUT = 3;
    static constexpr int SEQ_ID = 22;
    static constexpr int COST_NS = 261;
    int64_t fast, slow, sig;
    EmaCore ef, es, eg;
    __device__ void init(const Row<1> &r) { ef.init(fast, r.len); es.init(slow, r.len); eg.init(sig, r.len); }
    __device__ void step(const Row<1> &, int64_t, const double (&x)[1], double (&y)[3]) {
        double f = ef.step(x[0]), s = es.step(x[0]);
        double dif = (!pq_isnull(f) && !pq_isnull(s)) ? f - s : pq_null();
        double dea = eg.step(z0(dif));
        y[0] = dif; y[1] = dea;
        y[2] = (!pq_isnull(dif) && !pq_isnull(dea)) ? dif - dea : pq_null();
    }
    static constexpr bool FAST_NULL_OK = true; // N-B
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return ef.steady() && es.steady() && eg.steady(); }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[3]) {
        const double f = ef.fast(x[0]), s = es.fast(x[0]);
        const double dif = f - s;
        const double dea = eg.fast(dif);
        y[0] = dif; y[1] = dea; y[2] = dif - dea;
    }
};

struct TrixOp { // momentum.rs:544-569 (quirk Q-TRIX)
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 23;
    static constexpr int COST_NS = 200;
    int64_t p;
    EmaCore e1, e2, e3;
    double prev3;
    __device__ void init(const Row<1> &r) { e1.init(p, r.len); e2.init(p, r.len); e3.init(p, r.len); prev3 = pq_null(); }
    __device__ void step(const Row<1> &, int64_t i, const double (&x)[1], double (&y)[1]) {
        double a = e1.step(x[0]);
        double b = e2.step(z0(a));
        double c = e3.step(z0(b));
        y[0] = pq_null();
        if (i >= 1 && !pq_isnull(c) && !pq_isnull(prev3) && prev3 != 0.0) y[0] = (c - prev3) / prev3 * 100.0;
        prev3 = c;
    }
    static constexpr bool FAST_NULL_OK = true; // N-B
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return t0 >= 1 && e1.steady() && e2.steady() && e3.steady() && !pq_isnull(prev3); }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) {
        const double a = e1.fast(x[0]);
        const double b = e2.fast(a);
        const double c = e3.fast(b);
        const double v = (c - prev3) / prev3 * 100.0;
        y[0] = (prev3 != 0.0) ? v : pq_null();
        prev3 = c;
    }
};

template <int TK> // rows per tile: 4 on a full chip (LDS: four workgroups per CU); 8 for SMALL shards, where LDS is free and the per-tile costs count
struct UltoscOpK { // momentum.rs:572-627
    static constexpr int NIN = 3, NOUT = 1; // high, low, close
    static constexpr int SEQ_ID = TK == 4 ? 24 : 87;
    static constexpr int COST_NS = 517;
    static constexpr int TILE_K = TK; // 4-row tiles: 36.4 KB instead of 42.5 KB with the default periods, i.e. 4 workgroups per CU
    static constexpr int NTAP = 12; // per window: high, low, close at i-p and close at i-p-1
    static constexpr int TAP_COL[12] = {0, 1, 2, 2, 0, 1, 2, 2, 0, 1, 2, 2};
    int64_t p1, p2, p3;
    double sb[3], st[3], prev_c;
    __device__ void init(const Row<3> &) { for (int k = 0; k < 3; k++) sb[k] = st[k] = 0.0; prev_c = 0.0; }
    __device__ void tap_lags(int64_t (&lag)[12]) const {
        const int64_t ps[3] = {p1, p2, p3};
        bool ok = p1 > 0 && p2 > 0 && p3 > 0;
        for (int k = 0; k < 3; k++) {
            lag[4 * k] = lag[4 * k + 1] = lag[4 * k + 2] = ok ? ps[k] : 0;
            lag[4 * k + 3] = ok ? ps[k] + 1 : 0;
        }
    }
    __device__ static void bptr(double h, double l, double c, double pc, double &bp, double &tr) {
        double min_l_pc = fmin(l, pc), max_h_pc = fmax(h, pc);
        bp = c - min_l_pc; tr = max_h_pc - min_l_pc;
    }
    __device__ void step(const Row<3> &, int64_t i, const double (&x)[3], const double (&tp)[12], double (&y)[1]) {
        y[0] = pq_null();
        if (p1 <= 0 || p2 <= 0 || p3 <= 0) return;
        double bp = 0.0, tr = 0.0;
        if (i >= 1) bptr(x[0], x[1], x[2], prev_c, bp, tr);
        prev_c = x[2];
        const int64_t ps[3] = {p1, p2, p3};
        double a[3];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            sb[k] += bp; st[k] += tr;
            int64_t p = ps[k];
            if (i >= p) {
                double obp = 0.0, otr = 0.0;
                int64_t q = i - p;
                if (q >= 1) bptr(tp[4 * k], tp[4 * k + 1], tp[4 * k + 2], tp[4 * k + 3], obp, otr);
                sb[k] -= obp; st[k] -= otr;
            }
            if (i >= p - 1 && st[k] != 0.0) a[k] = sb[k] / st[k]; else ok = false;
        }
        if (ok) y[0] = 100.0 * (4.0 * a[0] + 2.0 * a[1] + a[2]) / 7.0;
    }
    Ring wb, wt; // bp / tr of the last max(p1,p2,p3) rows
    __host__ __device__ int64_t pmax() const { int64_t a = p1 > p2 ? p1 : p2; return a > p3 ? a : p3; }
    __host__ __device__ int64_t ring_slots() const { return pmax() > 0 ? 2 * pmax() : 2; }
    __device__ void init_lds(const Row<3> &r, RingAlloc &ra) { init(r); wb = ra.make(pmax()); wt = ra.make(pmax()); }
    __device__ void step_lds(int64_t i, const double (&x)[3], double (&y)[1]) {
        y[0] = pq_null();
        if (p1 <= 0 || p2 <= 0 || p3 <= 0) return;
        double bp = 0.0, tr = 0.0;
        if (i >= 1) bptr(x[0], x[1], x[2], prev_c, bp, tr);
        prev_c = x[2];
        const int64_t ps[3] = {p1, p2, p3};
        double a[3];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            sb[k] += bp; st[k] += tr;
            int64_t p = ps[k];
            if (i >= p) { sb[k] -= wb.get((int)p); st[k] -= wt.get((int)p); } // bp/tr of row i-p (row 0 holds 0.0)
            if (i >= p - 1 && st[k] != 0.0) a[k] = sb[k] / st[k]; else ok = false;
        }
        wb.push(bp); wt.push(tr);
        if (ok) y[0] = 100.0 * (4.0 * a[0] + 2.0 * a[1] + a[2]) / 7.0;
    }
    static constexpr bool FAST_NULL_OK = true; // N-B
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return p1 > 0 && p2 > 0 && p3 > 0 && t0 >= 1 && t0 >= pmax(); }
    __device__ void step_fast(int64_t, const double (&x)[3], double (&y)[1]) {
        double bp, tr;
        bptr(x[0], x[1], x[2], prev_c, bp, tr);
        prev_c = x[2];
        const int64_t ps[3] = {p1, p2, p3};
        double a[3];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            sb[k] += bp; st[k] += tr;
            sb[k] -= wb.get((int)ps[k]); st[k] -= wt.get((int)ps[k]);
            a[k] = sb[k] / st[k];
            ok = ok && st[k] != 0.0;
        }
        wb.push(bp); wt.push(tr);
        const double v = 100.0 * (4.0 * a[0] + 2.0 * a[1] + a[2]) / 7.0;
        y[0] = ok ? v : pq_null();
    }
};

typedef UltoscOpK<4> UltoscOp;
typedef UltoscOpK<8> UltoscOp8;

struct MfiOp { // momentum.rs:286-342
    static constexpr int NIN = 4, NOUT = 1; // high, low, close, volume
    static constexpr int SEQ_ID = 25;
    static constexpr int COST_NS = 880; // scheduling weight (x 1.4 for the volume-family job it anchors: weight search, -0.8 % per step)
    static constexpr int NTAP = 7; // high, low, close, volume at i-p; high, low, close at i-p-1
    static constexpr int TAP_COL[7] = {0, 1, 2, 3, 0, 1, 2};
    int64_t p;
    double pos, neg, prev_tp;
    __device__ void init(const Row<4> &) { pos = neg = 0.0; prev_tp = 0.0; }
    __device__ void tap_lags(int64_t (&lag)[7]) const {
        for (int k = 0; k < 4; k++) lag[k] = p > 0 ? p : 0;
        for (int k = 4; k < 7; k++) lag[k] = p > 0 ? p + 1 : 0;
    }
    __device__ void step(const Row<4> &, int64_t i, const double (&x)[4], const double (&lagv)[7], double (&y)[1]) {
        y[0] = pq_null();
        double tp = (x[0] + x[1] + x[2]) / 3.0;
        double mf = tp * x[3];
        if (i >= 1) {
            if (tp > prev_tp) pos += mf;
            else if (tp < prev_tp) neg += mf;
            if (i >= p) {
                int64_t q = i - p;
                if (q > 0) {
                    double tq = (lagv[0] + lagv[1] + lagv[2]) / 3.0;
                    double tq1 = (lagv[4] + lagv[5] + lagv[6]) / 3.0;
                    double mq = tq * lagv[3];
                    if (p == 0) { tq = tp; tq1 = prev_tp; mq = mf; } // the row just added is removed again
                    if (tq > tq1) pos -= mq;
                    else if (tq < tq1) neg -= mq;
                }
                if (neg == 0.0) y[0] = 100.0;
                else { double mr = pos / neg; y[0] = 100.0 - (100.0 / (1.0 + mr)); }
            }
        }
        prev_tp = tp;
    }
    Ring wtp, wmf; // typical price / money flow of the last p+1 rows
    __host__ __device__ int64_t ring_slots() const { return p > 0 ? 2 * (p + 1) : 2; }
    __device__ void init_lds(const Row<4> &r, RingAlloc &ra) { init(r); wtp = ra.make(p + 1); wmf = ra.make(p + 1); }
    __device__ void step_lds(int64_t i, const double (&x)[4], double (&y)[1]) {
        y[0] = pq_null();
        double tp = (x[0] + x[1] + x[2]) / 3.0;
        double mf = tp * x[3];
        if (i >= 1) {
            // selects, not branches: a conditional update of one of two accumulators makes the compiler index them
            // in scratch memory (two scratch round trips per row)
            const double padd = pos + mf, nadd = neg + mf;
            pos = (tp > prev_tp) ? padd : pos;
            neg = (tp < prev_tp) ? nadd : neg;
            if (p >= 0 && i >= p) {
                int64_t q = i - p;
                if (q > 0) {
                    double tq = tp, tq1 = prev_tp, mq = mf; // p == 0: the row just added is removed again
                    if (p > 0) { tq = wtp.get((int)p); tq1 = wtp.get((int)p + 1); mq = wmf.get((int)p); }
                    const double psub = pos - mq, nsub = neg - mq;
                    pos = (tq > tq1) ? psub : pos;
                    neg = (tq < tq1) ? nsub : neg;
                }
                if (neg == 0.0) y[0] = 100.0;
                else { double mr = pos / neg; y[0] = 100.0 - (100.0 / (1.0 + mr)); }
            }
        }
        prev_tp = tp;
        if (p > 0) { wtp.push(tp); wmf.push(mf); }
    }
    static constexpr bool FAST_NULL_OK = true; // N-B
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return p > 0 && t0 >= p + 1; } // the removed row q = i - p is >= 1
    __device__ void step_fast(int64_t, const double (&x)[4], double (&y)[1]) {
        const double tp = (x[0] + x[1] + x[2]) / 3.0;
        const double mf = tp * x[3];
        const double padd = pos + mf, nadd = neg + mf;
        pos = (tp > prev_tp) ? padd : pos;
        neg = (tp < prev_tp) ? nadd : neg;
        const double tq = wtp.get((int)p), tq1 = wtp.get((int)p + 1), mq = wmf.get((int)p);
        const double psub = pos - mq, nsub = neg - mq;
        pos = (tq > tq1) ? psub : pos;
        neg = (tq < tq1) ? nsub : neg;
        const double mr = pos / neg;
        const double v = 100.0 - (100.0 / (1.0 + mr));
        y[0] = (neg == 0.0) ? 100.0 : v;
        prev_tp = tp;
        wtp.push(tp); wmf.push(mf);
    }
};

// momentum.rs:668-727 calc_dm + its users.  MODE: 0 dx (also plus_di, quirk Q-PDI / D-5), 1 minus_di, 2 adx
template <int MODE>
struct DmOp {
    static constexpr int NIN = 3, NOUT = 1; // high, low, close
    static constexpr int SEQ_ID = 26 + MODE;
    static constexpr int COST_NS = 450;
    int64_t p;
    RmaCore rp, rm, rt, radx;
    double ph, pl, pc;
    __device__ void init(const Row<3> &r) {
        rp.init(p, r.len); rm.init(p, r.len); rt.init(p, r.len); radx.init(p, r.len);
        ph = pl = pc = 0.0;
    }
    // one row of calc_dm: returns adx (MODE 2 semantics), dx and minus_di through the references
    __device__ double step_all(int64_t i, const double (&x)[3], double &dx, double &mdi) {
        double p_dm = 0.0, m_dm = 0.0, tr = 0.0;
        if (i >= 1) {
            double up_move = x[0] - ph, down_move = pl - x[1];
            if (up_move > down_move && up_move > 0.0) p_dm = up_move;
            if (down_move > up_move && down_move > 0.0) m_dm = down_move;
            tr = fmax(fmax(x[0] - x[1], fabs(x[0] - pc)), fabs(x[1] - pc));
        }
        ph = x[0]; pl = x[1]; pc = x[2];
        double sp = rp.step(i, p_dm), sm = rm.step(i, m_dm), st = rt.step(i, tr);
        double pdi = pq_null();
        mdi = pq_null(); dx = pq_null();
        if (!pq_isnull(sp) && !pq_isnull(sm) && !pq_isnull(st) && st != 0.0) {
            pdi = 100.0 * sp / st;
            mdi = 100.0 * sm / st;
            double diff = fabs(pdi - mdi), sum = pdi + mdi;
            dx = (sum == 0.0) ? 0.0 : 100.0 * diff / sum;
        }
        return (MODE == 2) ? radx.step(i, z0(dx)) : pq_null(); // momentum.rs:21-27
    }
    __device__ void step(const Row<3> &, int64_t i, const double (&x)[3], double (&y)[1]) {
        double dx, mdi;
        double adx = step_all(i, x, dx, mdi);
        if (MODE == 0) y[0] = dx;
        else if (MODE == 1) y[0] = mdi;
        else y[0] = adx;
    }
    static constexpr bool FAST_NULL_OK = true; // N-B
    // all four Wilder averages seeded (the ADX one is fed z0(dx) on every row, so it is seeded at row p - 1 as well)
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return t0 >= 1 && rp.steady(t0) && (MODE != 2 || radx.steady(t0)); }
    __device__ double fast_all(const double (&x)[3], double &dx, double &mdi) {
        const double up_move = x[0] - ph, down_move = pl - x[1];
        const double p_dm = (up_move > down_move && up_move > 0.0) ? up_move : 0.0;
        const double m_dm = (down_move > up_move && down_move > 0.0) ? down_move : 0.0;
        const double tr = fmax(fmax(x[0] - x[1], fabs(x[0] - pc)), fabs(x[1] - pc));
        ph = x[0]; pl = x[1]; pc = x[2];
        const double sp = rp.fast(p_dm), sm = rm.fast(m_dm), st = rt.fast(tr);
        const double pdi = 100.0 * sp / st, mdi_ = 100.0 * sm / st;
        const double diff = fabs(pdi - mdi_), sum = pdi + mdi_;
        const double dxv = 100.0 * diff / sum;
        const bool ok = st != 0.0;
        mdi = ok ? mdi_ : pq_null();
        dx = ok ? ((sum == 0.0) ? 0.0 : dxv) : pq_null();
        return (MODE == 2) ? radx.fast(ok ? ((sum == 0.0) ? 0.0 : dxv) : 0.0) : pq_null();
    }
    __device__ void step_fast(int64_t, const double (&x)[3], double (&y)[1]) {
        double dx, mdi;
        double adx = fast_all(x, dx, mdi);
        if (MODE == 0) y[0] = dx;
        else if (MODE == 1) y[0] = mdi;
        else y[0] = adx;
    }
};
template <bool PLUS> // momentum.rs:414-436 / :359-381
struct DmRawOp {
    static constexpr int NIN = 2, NOUT = 1; // high, low
    static constexpr int SEQ_ID = 29 + (PLUS ? 0 : 1);
    static constexpr int COST_NS = 90; // scheduling weight (the plus_dm / minus_dm pair a little later in its grid: -1 % per step, weight search)
    int64_t p;
    RmaCore rr;
    double ph, pl;
    __device__ void init(const Row<2> &r) { rr.init(p, r.len); ph = pl = 0.0; }
    __device__ void step(const Row<2> &, int64_t i, const double (&x)[2], double (&y)[1]) {
        double d = 0.0;
        if (i >= 1) {
            double up_move = x[0] - ph, down_move = pl - x[1];
            if (PLUS) { if (up_move > down_move && up_move > 0.0) d = up_move; }
            else { if (down_move > up_move && down_move > 0.0) d = down_move; }
        }
        ph = x[0]; pl = x[1];
        y[0] = rr.step(i, d);
    }
    static constexpr bool FAST_NULL_OK = true; // N-B
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return t0 >= 1 && rr.steady(t0); }
    __device__ void step_fast(int64_t, const double (&x)[2], double (&y)[1]) {
        const double up_move = x[0] - ph, down_move = pl - x[1];
        double d;
        if (PLUS) d = (up_move > down_move && up_move > 0.0) ? up_move : 0.0;
        else d = (down_move > up_move && down_move > 0.0) ? down_move : 0.0;
        ph = x[0]; pl = x[1];
        y[0] = rr.fast(d);
    }
};
struct SmaTpOp { // momentum.rs:148-158: calc_sma(tp) on a null-free slice; the lagged tp is recomputed
    static constexpr int NIN = 3, NOUT = 1;
    static constexpr int SEQ_ID = 31;
    static constexpr int COST_NS = 150;
    static constexpr int NTAP = 3;
    static constexpr int TAP_COL[3] = {0, 1, 2};
    int64_t p;
    double sum, denom;
    bool dead;
    __device__ void init(const Row<3> &r) { dead = (p <= 0 || r.len < p); sum = 0.0; denom = 1.0 / (double)p; }
    __device__ void tap_lags(int64_t (&lag)[3]) const { lag[0] = lag[1] = lag[2] = dead ? 0 : p; }
    __device__ void step(const Row<3> &, int64_t i, const double (&x)[3], const double (&tp)[3], double (&y)[1]) {
        y[0] = pq_null();
        if (dead) return;
        sum += (x[0] + x[1] + x[2]) / 3.0;
        if (i < p - 1) return;
        if (i >= p) sum -= (tp[0] + tp[1] + tp[2]) / 3.0;
        y[0] = sum * denom;
    }
    Ring w;
    __host__ __device__ int64_t ring_slots() const { return p > 0 ? p : 1; }
    __device__ void init_lds(const Row<3> &r, RingAlloc &ra) { init(r); w = ra.make(p); }
    __device__ void step_lds(int64_t i, const double (&x)[3], double (&y)[1]) {
        y[0] = pq_null();
        if (dead) return;
        double tp = (x[0] + x[1] + x[2]) / 3.0;
        sum += tp;
        double old = w.swap(tp);
        if (i < p - 1) return;
        if (i >= p) sum -= old;
        y[0] = sum * denom;
    }
};

