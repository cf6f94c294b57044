// ops_backtest.h -- per-symbol backtest scan + summary (device body shared by backtest.hip and the suite job grid)
#pragma once
#include "pq_cores.h"

__device__ __forceinline__ double z0b(double x) { return pq_isnull(x) ? 0.0 : x; }

struct BtArgs {
    const double *price;
    const uint8_t *buy, *sell; // nullptr when MACD signals are generated in-kernel
    const double *bench;       // nullable
    double *position, *cash;   // nullable
    double *equity;            // always present (user buffer or workspace)
    double *summary;           // [n][8], nullable
    uint8_t *buy_out, *sell_out; // signals-only mode
    pq_bt_params prm;
    int64_t fast, slow, sig;
};

template <bool MACD_SIGNALS, bool SIGNALS_ONLY>
__device__ __forceinline__ void backtest_body(const BtArgs &a, const Dims &d, int64_t s) {
    const int64_t base = dims_base(d, s), T = dims_len(d, s);
    const double *price = a.price + base;
    const pq_bt_params prm = a.prm;

    EmaCore ef, es, eg;
    double prev_m = pq_null(), prev_s = pq_null();
    if (MACD_SIGNALS) { ef.init(a.fast, T); es.init(a.slow, T); eg.init(a.sig, T); }

    double pos = 0.0, avail = prm.initial_capital, peak = prm.initial_capital, entry_cost = 0.0;
    int64_t trades = 0, wins = 0;
    // summary pass 1 state (metrics.rs:21-49)
    double max_dd = 0.0, max_eq = prm.initial_capital, prev_eq = prm.initial_capital, ret_sum = 0.0;
    double last_eq = prm.initial_capital;

    for (int64_t i = 0; i < T; i++) {
        double px = price[i];
        bool buy, sell;
        if (MACD_SIGNALS) { // momentum.rs:250-283 + D-8 cross rule
            double f = ef.step(px), sl = es.step(px);
            double m = (!pq_isnull(f) && !pq_isnull(sl)) ? f - sl : pq_null();
            double g = eg.step(z0b(m));
            bool ok = i > 0 && !pq_isnull(m) && !pq_isnull(g) && !pq_isnull(prev_m) && !pq_isnull(prev_s);
            buy = ok && (prev_m <= prev_s) && (m > g);
            sell = ok && (prev_m >= prev_s) && (m < g);
            prev_m = m; prev_s = g;
            if (SIGNALS_ONLY) { a.buy_out[base + i] = buy; a.sell_out[base + i] = sell; continue; }
        } else {
            buy = a.buy[base + i] != 0;
            sell = a.sell[base + i] != 0;
        }
        double eq;
        if (pq_isnull(px)) px = __longlong_as_double(0x7FF8000000000000LL); // null -> NaN (vectorized.rs:70-78)
        if (isnan(px) || px <= 0.0) { // vectorized.rs:141-144: state untouched
            eq = avail + pos * px;
        } else {
            if (buy && pos == 0.0) { // :146-161
                double exec = px + prm.buy_slippage;
                double cur_eq = avail + pos * px;
                double deploy = cur_eq * prm.position_size;
                double qty = floor(deploy / exec);
                if (qty > 0.0) {
                    double cost = qty * exec;
                    double fee = fmax(cost * prm.buy_commission_rate, prm.min_commission);
                    pos += qty;
                    avail -= cost + fee;
                    entry_cost = pos * px;
                    trades += 1;
                }
            } else if (sell && pos > 0.0) { // :162-175
                double exec = px - prm.sell_slippage;
                double revenue = pos * exec;
                double fee = fmax(revenue * prm.sell_commission_rate, prm.min_commission);
                double net = revenue - fee;
                if (net > entry_cost) wins += 1;
                avail += net;
                pos = 0.0;
            }
            eq = avail + pos * px;
            if (eq > peak) peak = eq;
        }
        if (a.position) a.position[base + i] = pos;
        if (a.cash) a.cash[base + i] = avail;
        a.equity[base + i] = eq;
        // metrics.rs:26-49
        if (eq > max_eq) max_eq = eq;
        double dd = (max_eq > 0.0) ? (max_eq - eq) / max_eq : 0.0;
        if (dd > max_dd) max_dd = dd;
        double r = (prev_eq > 0.0) ? (eq - prev_eq) / prev_eq : 0.0;
        ret_sum += r;
        prev_eq = eq;
        last_eq = eq;
    }
    if (SIGNALS_ONLY || a.summary == nullptr) return;
    double *sm = a.summary + s * PQ_SUMMARY_COLS;
    if (T == 0) { for (int k = 0; k < 8; k++) sm[k] = 0.0; return; }
    const double DAYS = 252.0, RF = 0.03;
    double total_return = (last_eq - prm.initial_capital) / prm.initial_capital;
    double ann = (total_return > -1.0) ? pow(1.0 + total_return, DAYS / (double)T) - 1.0 : -1.0;
    double mean = ret_sum / (double)T;
    double dof = fmax((double)T - 1.0, 1.0);
    // pass 2: variance of daily returns, recomputed from the lane's own equity rows
    const double *eqr = a.equity + base;
    double vs = 0.0, pe = prm.initial_capital;
    for (int64_t i = 0; i < T; i++) {
        double e = eqr[i];
        double r = (pe > 0.0) ? (e - pe) / pe : 0.0;
        double dlt = r - mean;
        vs += dlt * dlt;
        pe = e;
    }
    double var = vs / dof;
    double vol = sqrt(var) * sqrt(DAYS);
    double sharpe = (vol > 0.0) ? (ann - RF) / vol : 0.0;
    double win_rate = (trades > 0) ? (double)wins / (double)trades : 0.0;
    double alpha = 0.0, beta = 0.0;
    if (a.bench) { // metrics.rs:86-140
        const double *bm = a.bench + base;
        double pb = bm[0], bs = 0.0;
        for (int64_t i = 0; i < T; i++) { double bv = bm[i]; bs += (pb > 0.0) ? (bv - pb) / pb : 0.0; pb = bv; }
        double bmean = bs / (double)T;
        double bvar = 0.0, cov = 0.0;
        pb = bm[0];
        for (int64_t i = 0; i < T; i++) {
            double bv = bm[i];
            double br = (pb > 0.0) ? (bv - pb) / pb : 0.0;
            double dlt = br - bmean;
            bvar += dlt * dlt;
            pb = bv;
        }
        bvar /= dof;
        pb = bm[0]; pe = prm.initial_capital;
        for (int64_t i = 0; i < T; i++) {
            double bv = bm[i], e = eqr[i];
            double br = (pb > 0.0) ? (bv - pb) / pb : 0.0;
            double r = (pe > 0.0) ? (e - pe) / pe : 0.0;
            cov += (r - mean) * (br - bmean);
            pb = bv; pe = e;
        }
        cov /= dof;
        if (bvar > 0.0) beta = cov / bvar;
        double b0 = bm[0], b1 = bm[T - 1];
        double btr = (b0 > 0.0) ? (b1 - b0) / b0 : 0.0;
        double bann = (btr > -1.0) ? pow(1.0 + btr, DAYS / (double)T) - 1.0 : -1.0;
        alpha = ann - (RF + beta * (bann - RF));
    }
    sm[0] = ann; sm[1] = max_dd; sm[2] = alpha; sm[3] = beta; sm[4] = sharpe;
    sm[5] = fmax(total_return, 0.0); sm[6] = win_rate; sm[7] = (double)trades;
}


// The MACD-cross backtest as a tiled SEQ op (price in; position, cash, equity out): same arithmetic as backtest_body<true,
// false> with all three state columns requested and no benchmark, but the columns move through the coalesced tile path
// instead of per-lane 8-byte accesses.  finish() is the summary: pass 2 re-reads the lane's own equity row.
struct BtMacdOp {
    static constexpr int NIN = 1, NOUT = 3;
    static constexpr int SEQ_ID = 62;
    static constexpr int COST_NS = 635;
    pq_bt_params prm;
    int64_t fast, slow, sig;
    double *summary; // [n][8], nullable
    EmaCore ef, es, eg;
    double prev_m, prev_s, pos, avail, peak, entry_cost, max_dd, max_eq, prev_eq, ret_sum, last_eq;
    int64_t trades, wins;
    __device__ void init(const Row<1> &r) {
        ef.init(fast, r.len); es.init(slow, r.len); eg.init(sig, r.len);
        prev_m = pq_null(); prev_s = pq_null();
        pos = 0.0; avail = prm.initial_capital; peak = prm.initial_capital; entry_cost = 0.0;
        trades = 0; wins = 0;
        max_dd = 0.0; max_eq = prm.initial_capital; prev_eq = prm.initial_capital; ret_sum = 0.0; last_eq = prm.initial_capital;
    }
    __device__ void step(const Row<1> &, int64_t i, const double (&x)[1], double (&y)[3]) {
        double px = x[0];
        double f = ef.step(px), sl = es.step(px); // momentum.rs:250-283 + D-8 cross rule
        double m = (!pq_isnull(f) && !pq_isnull(sl)) ? f - sl : pq_null();
        double g = eg.step(z0b(m));
        bool ok = i > 0 && !pq_isnull(m) && !pq_isnull(g) && !pq_isnull(prev_m) && !pq_isnull(prev_s);
        bool buy = ok && (prev_m <= prev_s) && (m > g);
        bool sell = ok && (prev_m >= prev_s) && (m < g);
        prev_m = m; prev_s = g;
        trade_row(px, buy, sell, y);
    }
    static constexpr bool HAS_FAST = true;
    // the three EMAs seeded and the previous row's macd / signal known: the signal rule needs no null tests
    __device__ bool steady(int64_t t0) const {
        return t0 > 0 && ef.steady() && es.steady() && eg.steady() && !pq_isnull(prev_m) && !pq_isnull(prev_s);
    }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[3]) {
        const double px = x[0];
        const double f = ef.fast(px), sl = es.fast(px);
        const double m = f - sl;
        const double g = eg.fast(m);
        const bool buy = (prev_m <= prev_s) && (m > g);
        const bool sell = (prev_m >= prev_s) && (m < g);
        prev_m = m; prev_s = g;
        trade_row(px, buy, sell, y);
    }
    // vectorized.rs:130-194 for one row + the running parts of calculate_summary (metrics.rs:26-49)
    __device__ __forceinline__ void trade_row(double px, bool buy, bool sell, double (&y)[3]) {
        double eq;
        if (pq_isnull(px)) px = __longlong_as_double(0x7FF8000000000000LL); // null -> NaN (vectorized.rs:70-78)
        if (isnan(px) || px <= 0.0) { // vectorized.rs:141-144: state untouched
            eq = avail + pos * px;
        } else {
            if (buy && pos == 0.0) { // :146-161
                double exec = px + prm.buy_slippage;
                double cur_eq = avail + pos * px;
                double deploy = cur_eq * prm.position_size;
                double qty = floor(deploy / exec);
                if (qty > 0.0) {
                    double cost = qty * exec;
                    double fee = fmax(cost * prm.buy_commission_rate, prm.min_commission);
                    pos += qty;
                    avail -= cost + fee;
                    entry_cost = pos * px;
                    trades += 1;
                }
            } else if (sell && pos > 0.0) { // :162-175
                double exec = px - prm.sell_slippage;
                double revenue = pos * exec;
                double fee = fmax(revenue * prm.sell_commission_rate, prm.min_commission);
                double net = revenue - fee;
                if (net > entry_cost) wins += 1;
                avail += net;
                pos = 0.0;
            }
            eq = avail + pos * px;
            if (eq > peak) peak = eq;
        }
        y[0] = pos; y[1] = avail; y[2] = eq;
        if (eq > max_eq) max_eq = eq; // metrics.rs:26-49
        double dd = (max_eq > 0.0) ? (max_eq - eq) / max_eq : 0.0;
        if (dd > max_dd) max_dd = dd;
        double r = (prev_eq > 0.0) ? (eq - prev_eq) / prev_eq : 0.0;
        ret_sum += r;
        prev_eq = eq;
        last_eq = eq;
    }
    __host__ __device__ void *finish_writes() const { return summary; } // what finish() writes besides the op's columns
    // called once per live lane after every row of the series has been stored
    __device__ void finish(double *const *outp, const Dims &d, int64_t s) {
        if (summary == nullptr) return;
        const int64_t T = dims_len(d, s);
        double *sm = summary + s * PQ_SUMMARY_COLS;
        if (T == 0) { for (int k = 0; k < 8; k++) sm[k] = 0.0; return; }
        const double DAYS = 252.0, RF = 0.03;
        double total_return = (last_eq - prm.initial_capital) / prm.initial_capital;
        double ann = (total_return > -1.0) ? pow(1.0 + total_return, DAYS / (double)T) - 1.0 : -1.0;
        double mean = ret_sum / (double)T;
        double dof = fmax((double)T - 1.0, 1.0);
        const double *eqr = outp[2] + dims_base(d, s);
        double vs = 0.0, pe = prm.initial_capital;
        for (int64_t i = 0; i < T; i++) {
            double e = eqr[i];
            double r = (pe > 0.0) ? (e - pe) / pe : 0.0;
            double dlt = r - mean;
            vs += dlt * dlt;
            pe = e;
        }
        double var = vs / dof;
        double vol = sqrt(var) * sqrt(DAYS);
        double sharpe = (vol > 0.0) ? (ann - RF) / vol : 0.0;
        double win_rate = (trades > 0) ? (double)wins / (double)trades : 0.0;
        sm[0] = ann; sm[1] = max_dd; sm[2] = 0.0; sm[3] = 0.0; sm[4] = sharpe;
        sm[5] = fmax(total_return, 0.0); sm[6] = win_rate; sm[7] = (double)trades;
    }
};

// calculate_summary (metrics.rs:7-152) from a stored equity row; `bench`: this lane's benchmark rows or nullptr.
__device__ __forceinline__ void bt_summary_from_row(const double *eqr, int64_t T, double initial_capital, int64_t trades,
                                                    int64_t wins, const double *bm, double *sm) {
    if (T == 0) { for (int k = 0; k < 8; k++) sm[k] = 0.0; return; }
    const double DAYS = 252.0, RF = 0.03;
    double max_dd = 0.0, max_eq = initial_capital, pe = initial_capital, ret_sum = 0.0;
    for (int64_t i = 0; i < T; i++) { // metrics.rs:26-49
        double e = eqr[i];
        if (e > max_eq) max_eq = e;
        double dd = (max_eq > 0.0) ? (max_eq - e) / max_eq : 0.0;
        if (dd > max_dd) max_dd = dd;
        ret_sum += (pe > 0.0) ? (e - pe) / pe : 0.0;
        pe = e;
    }
    const double last_eq = eqr[T - 1];
    double total_return = (last_eq - initial_capital) / initial_capital;
    double ann = (total_return > -1.0) ? pow(1.0 + total_return, DAYS / (double)T) - 1.0 : -1.0;
    double mean = ret_sum / (double)T;
    double dof = fmax((double)T - 1.0, 1.0);
    double vs = 0.0;
    pe = initial_capital;
    for (int64_t i = 0; i < T; i++) {
        double e = eqr[i];
        double r = (pe > 0.0) ? (e - pe) / pe : 0.0;
        double dlt = r - mean;
        vs += dlt * dlt;
        pe = e;
    }
    double var = vs / dof;
    double vol = sqrt(var) * sqrt(DAYS);
    double sharpe = (vol > 0.0) ? (ann - RF) / vol : 0.0;
    double win_rate = (trades > 0) ? (double)wins / (double)trades : 0.0;
    double alpha = 0.0, beta = 0.0;
    if (bm) { // metrics.rs:86-140
        double pb = bm[0], bs = 0.0;
        for (int64_t i = 0; i < T; i++) { double bv = bm[i]; bs += (pb > 0.0) ? (bv - pb) / pb : 0.0; pb = bv; }
        double bmean = bs / (double)T;
        double bvar = 0.0, cov = 0.0;
        pb = bm[0];
        for (int64_t i = 0; i < T; i++) {
            double bv = bm[i];
            double br = (pb > 0.0) ? (bv - pb) / pb : 0.0;
            double dlt = br - bmean;
            bvar += dlt * dlt;
            pb = bv;
        }
        bvar /= dof;
        pb = bm[0]; pe = initial_capital;
        for (int64_t i = 0; i < T; i++) {
            double bv = bm[i], e = eqr[i];
            double br = (pb > 0.0) ? (bv - pb) / pb : 0.0;
            double r = (pe > 0.0) ? (e - pe) / pe : 0.0;
            cov += (r - mean) * (br - bmean);
            pb = bv; pe = e;
        }
        cov /= dof;
        if (bvar > 0.0) beta = cov / bvar;
        double b0 = bm[0], b1 = bm[T - 1];
        double btr = (b0 > 0.0) ? (b1 - b0) / b0 : 0.0;
        double bann = (btr > -1.0) ? pow(1.0 + btr, DAYS / (double)T) - 1.0 : -1.0;
        alpha = ann - (RF + beta * (bann - RF));
    }
    sm[0] = ann; sm[1] = max_dd; sm[2] = alpha; sm[3] = beta; sm[4] = sharpe;
    sm[5] = fmax(total_return, 0.0); sm[6] = win_rate; sm[7] = (double)trades;
}

// SURVEY 8(f) rank 1 / decision D-10 (oracle/backtest.c pqo_backtest_leveraged): leveraged per-symbol pool.
struct LevArgs {
    const double *price;
    const uint8_t *buy, *sell;
    const double *bench; // one shared series or nullptr
    double *cash_net, *stock_value, *total_value;
    int32_t max_trades;
    int32_t *trade_count, *entry_day, *exit_day, *reason; // nullable (records) ; trade_count nullable
    double *entry_price, *exit_price, *quantity, *pnl, *pnl_pct;
    double *summary;
    pq_lev_params prm;
};
static __global__ __launch_bounds__(SEQ_BLOCK) void lev_backtest_kernel(LevArgs a, Dims d) {
    const int64_t s = (int64_t)blockIdx.x * SEQ_BLOCK + threadIdx.x;
    if (s >= d.n) return;
    const int64_t base = dims_base(d, s), T = dims_len(d, s);
    const pq_lev_params prm = a.prm;
    double cash = prm.initial_capital, debt = 0.0, shares = 0.0, last_px = 0.0, e_outlay = 0.0, e_price = 0.0;
    int64_t e_day = 0, trades = 0, wins = 0;
    const int64_t rb = s * (int64_t)a.max_trades;
    for (int64_t t = 0; t < T; t++) {
        double p = a.price[base + t];
        if (pq_isnull(p)) p = __longlong_as_double(0x7FF8000000000000LL);
        const bool valid = !(isnan(p) || p <= 0.0);
        if (debt > 0.0) debt += debt * (prm.interest_rate / 252.0); // D-10 step 1: the daily rate is formed first
        if (valid) {
            last_px = p;
            int do_sell = 0;
            if (shares > 0.0) {
                if (debt > 0.0 && cash + shares * p - debt < prm.margin_call_threshold * (shares * p)) do_sell = 2;
                else if (a.sell[base + t]) do_sell = 1;
            }
            if (do_sell) {
                double exec = p * (1.0 - prm.slippage);
                double rev = shares * exec;
                double fee = fmax(rev * prm.commission_rate, prm.min_commission);
                double net = rev - fee;
                double gain = net - e_outlay;
                if (trades < a.max_trades && a.entry_day) {
                    a.entry_day[rb + trades] = (int32_t)e_day; a.exit_day[rb + trades] = (int32_t)t;
                    a.entry_price[rb + trades] = e_price; a.exit_price[rb + trades] = exec; a.quantity[rb + trades] = shares;
                    a.pnl[rb + trades] = gain; a.pnl_pct[rb + trades] = gain / e_outlay * 100.0; a.reason[rb + trades] = do_sell;
                }
                trades += 1;
                if (gain > 0.0) wins += 1;
                cash = cash + net - debt;
                debt = 0.0;
                shares = 0.0;
            } else if (a.buy[base + t] && shares == 0.0) {
                double exec = p * (1.0 + prm.slippage);
                double power = cash * prm.position_size * prm.leverage;
                double lots = floor(power / (exec * 100.0));
                double cost = 0.0, fee = 0.0;
                while (lots > 0.0) {
                    cost = lots * 100.0 * exec;
                    fee = fmax(cost * prm.commission_rate, prm.min_commission);
                    if (cost + fee <= cash * prm.leverage) break;
                    lots -= 1.0;
                }
                if (lots > 0.0) {
                    double outlay = cost + fee;
                    debt = fmax(outlay - cash, 0.0);
                    cash = fmax(cash - outlay, 0.0);
                    shares = lots * 100.0;
                    e_outlay = outlay; e_price = exec; e_day = t;
                }
            }
        }
        double sv = shares * last_px;
        a.cash_net[base + t] = cash - debt;
        a.stock_value[base + t] = sv;
        a.total_value[base + t] = (cash - debt) + sv;
    }
    if (a.trade_count) a.trade_count[s] = (int32_t)trades;
    if (a.summary) bt_summary_from_row(a.total_value + base, T, prm.initial_capital, trades, wins, a.bench, a.summary + s * PQ_SUMMARY_COLS);
}

// The same engine as a tiled SEQ op: the price and the three daily columns move through the coalesced tile path; the
// two uint8 signal columns are read by the lane itself, eight rows (one aligned 8-byte word each) at a time and one tile
// ahead; trade records are rare events and go straight to global memory.  Requires stride % 8 == 0 and 8-byte aligned
// signal columns (the launcher checks; otherwise lev_backtest_kernel above runs).
struct LevOp {
    static constexpr int NIN = 1, NOUT = 3;
    static constexpr int TILE_K = 8;
    static constexpr int SEQ_ID = 63;
    static constexpr int COST_NS = 1500; // branchy per-lane state machine: some lane of the wave trades on nearly every row
    // columns the op reads beside its tile inputs (hazard tracking of a recorded suite)
    __host__ void extra_reads(const void *(&r)[4]) const { r[0] = a.buy; r[1] = a.sell; r[2] = a.bench; r[3] = nullptr; }
    LevArgs a;
    int64_t stride; // elements between series (set by the launcher)
    // per-lane state
    const uint8_t *brow, *srow;
    int64_t rec0, T;
    unsigned long long bcur, scur, bnext, snext;
    double cash, debt, shares, last_px, e_outlay, e_price;
    int64_t e_day, trades, wins;
    __device__ void init(const Row<1> &r) {
        const int64_t off = r.in[0] - a.price; // element offset of this lane's series
        brow = a.buy + off; srow = a.sell + off;
        T = r.len;
        rec0 = (off / stride) * (int64_t)a.max_trades; // this series' slice of the trade-record arrays
        cash = a.prm.initial_capital; debt = 0.0; shares = 0.0; last_px = 0.0; e_outlay = 0.0; e_price = 0.0;
        e_day = 0; trades = 0; wins = 0;
        bcur = scur = 0;
        bnext = T >= 8 ? *reinterpret_cast<const unsigned long long *>(brow) : 0;
        snext = T >= 8 ? *reinterpret_cast<const unsigned long long *>(srow) : 0;
    }
    __device__ void step(const Row<1> &, int64_t t, const double (&x)[1], double (&y)[3]) {
        const int j = (int)(t & 7);
        if (j == 0) { // next eight rows of signals: use the words fetched a tile ago, fetch the following ones
            if (t + 8 <= T) {
                bcur = bnext; scur = snext;
                if (t + 16 <= T) {
                    bnext = *reinterpret_cast<const unsigned long long *>(brow + t + 8);
                    snext = *reinterpret_cast<const unsigned long long *>(srow + t + 8);
                }
            } else { // ragged tail: byte loads
                bcur = scur = 0;
                for (int64_t q = t; q < T; q++) {
                    bcur |= (unsigned long long)(brow[q] != 0) << (8 * (q - t));
                    scur |= (unsigned long long)(srow[q] != 0) << (8 * (q - t));
                }
            }
        }
        const bool buy = ((bcur >> (8 * j)) & 0xff) != 0, sell = ((scur >> (8 * j)) & 0xff) != 0;
        const pq_lev_params &prm = a.prm;
        double p = x[0];
        if (pq_isnull(p)) p = __longlong_as_double(0x7FF8000000000000LL);
        const bool valid = !(isnan(p) || p <= 0.0);
        if (debt > 0.0) debt += debt * (prm.interest_rate / 252.0); // D-10 step 1: the daily rate is formed first
        if (valid) {
            last_px = p;
            int do_sell = 0;
            if (shares > 0.0) {
                if (debt > 0.0 && cash + shares * p - debt < prm.margin_call_threshold * (shares * p)) do_sell = 2;
                else if (sell) do_sell = 1;
            }
            if (do_sell) {
                double exec = p * (1.0 - prm.slippage);
                double rev = shares * exec;
                double fee = fmax(rev * prm.commission_rate, prm.min_commission);
                double net = rev - fee;
                double gain = net - e_outlay;
                if (trades < a.max_trades && a.entry_day) {
                    const int64_t rb = rec0 + trades;
                    a.entry_day[rb] = (int32_t)e_day; a.exit_day[rb] = (int32_t)t;
                    a.entry_price[rb] = e_price; a.exit_price[rb] = exec; a.quantity[rb] = shares;
                    a.pnl[rb] = gain; a.pnl_pct[rb] = gain / e_outlay * 100.0; a.reason[rb] = do_sell;
                }
                trades += 1;
                if (gain > 0.0) wins += 1;
                cash = cash + net - debt;
                debt = 0.0;
                shares = 0.0;
            } else if (buy && shares == 0.0) {
                double exec = p * (1.0 + prm.slippage);
                double power = cash * prm.position_size * prm.leverage;
                double lots = floor(power / (exec * 100.0));
                double cost = 0.0, fee = 0.0;
                while (lots > 0.0) {
                    cost = lots * 100.0 * exec;
                    fee = fmax(cost * prm.commission_rate, prm.min_commission);
                    if (cost + fee <= cash * prm.leverage) break;
                    lots -= 1.0;
                }
                if (lots > 0.0) {
                    double outlay = cost + fee;
                    debt = fmax(outlay - cash, 0.0);
                    cash = fmax(cash - outlay, 0.0);
                    shares = lots * 100.0;
                    e_outlay = outlay; e_price = exec; e_day = t;
                }
            }
        }
        const double sv = shares * last_px;
        y[0] = cash - debt; y[1] = sv; y[2] = (cash - debt) + sv;
    }
    __host__ __device__ void *finish_writes() const { return a.summary; }
    __device__ void finish(double *const *outp, const Dims &d, int64_t s) {
        if (a.trade_count) a.trade_count[s] = (int32_t)trades;
        if (a.summary) bt_summary_from_row(outp[2] + dims_base(d, s), dims_len(d, s), a.prm.initial_capital, trades, wins, a.bench, a.summary + s * PQ_SUMMARY_COLS);
    }
};

// SURVEY 8(f) rank 2 / decision D-11 (oracle/backtest.c): Strategy signal rules, ROW shape, two uint8 columns out.
__device__ __forceinline__ double sig_at(const double *col, int64_t i) { return i >= 0 ? col[i] : pq_null(); }
struct CrossSigOp {
    static constexpr int NIN = 2, NOUT = 2;
    typedef uint8_t OutT;
    __device__ void eval(const Row<2> &r, int64_t i, uint8_t (&y)[2]) {
        const double a1 = r.in[0][i], b1 = r.in[1][i], a0 = sig_at(r.in[0], i - 1), b0 = sig_at(r.in[1], i - 1);
        const bool ok = i > 0 && !pq_isnull(a1) && !pq_isnull(b1) && !pq_isnull(a0) && !pq_isnull(b0);
        y[0] = ok && (a0 <= b0) && (a1 > b1);
        y[1] = ok && (a0 >= b0) && (a1 < b1);
    }
};
struct BandSigOp {
    static constexpr int NIN = 1, NOUT = 2;
    typedef uint8_t OutT;
    double lower, upper;
    __device__ void eval(const Row<1> &r, int64_t i, uint8_t (&y)[2]) {
        const double x1 = r.in[0][i], x0 = sig_at(r.in[0], i - 1);
        const bool ok = i > 0 && !pq_isnull(x1) && !pq_isnull(x0);
        y[0] = ok && (x0 < lower) && (x1 >= lower);
        y[1] = ok && (x0 > upper) && (x1 <= upper);
    }
};
template <int MODE>
struct ChannelSigOp {
    static constexpr int NIN = 3, NOUT = 2; // price, lo, hi
    typedef uint8_t OutT;
    __device__ void eval(const Row<3> &r, int64_t i, uint8_t (&y)[2]) {
        const double p1 = r.in[0][i], p0 = sig_at(r.in[0], i - 1);
        const double lo0 = sig_at(r.in[1], i - 1), hi0 = sig_at(r.in[2], i - 1);
        if (MODE == 0) {
            const double lo1 = r.in[1][i], hi1 = r.in[2][i];
            const bool ok = i > 0 && !pq_isnull(p1) && !pq_isnull(p0) && !pq_isnull(lo1) && !pq_isnull(hi1) && !pq_isnull(lo0) && !pq_isnull(hi0);
            y[0] = ok && (p1 < lo1) && (p0 >= lo0);
            y[1] = ok && (p1 > hi1) && (p0 <= hi0);
        } else {
            const bool ok = i > 0 && !pq_isnull(p1) && !pq_isnull(lo0) && !pq_isnull(hi0);
            y[0] = ok && (p1 > hi0);
            y[1] = ok && (p1 < lo0);
        }
    }
};

// get_performance_metrics: per-day sum over the symbols (blocks of 256, one (day, block) per thread: coalesced across days),
// then the day-to-day columns; beta needs ordered sums over the days and is done by one thread (T is a few thousand).
constexpr int PORTFOLIO_BLOCK = 256; // = oracle PQO_SUM_BLOCK: the per-day sum is defined over blocks of symbols
static __global__ __launch_bounds__(64) void portfolio_partial_kernel(const double *tv, Dims d, double *part) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= d.len) return;
    const int64_t s_lo = (int64_t)blockIdx.y * PORTFOLIO_BLOCK, s_hi = s_lo + PORTFOLIO_BLOCK < d.n ? s_lo + PORTFOLIO_BLOCK : d.n;
    double pv = 0.0;
    int64_t s = s_lo;
    for (; s + 32 <= s_hi; s += 32) { // 32 independent loads in flight, then the adds in ascending symbol order
        double v[32];
#pragma unroll
        for (int k = 0; k < 32; k++) v[k] = tv[(s + k) * d.stride + t];
#pragma unroll
        for (int k = 0; k < 32; k++) pv += v[k];
    }
    for (; s < s_hi; s++) pv += tv[s * d.stride + t];
    part[(int64_t)blockIdx.y * d.len + t] = pv;
}
static __global__ __launch_bounds__(64) void portfolio_combine_kernel(const double *part, int64_t nblk, int64_t len, double *out) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= len) return;
    double pv = 0.0;
    for (int64_t k = 0; k < nblk; k++) pv += part[k * len + t];
    out[t * PQ_PORTFOLIO_COLS] = pv;
}
static __global__ __launch_bounds__(256) void portfolio_metrics_kernel(int64_t T, double initial_total, const double *bm, double *out) {
    for (int64_t t = threadIdx.x; t < T; t += 256) {
        double *o = out + t * PQ_PORTFOLIO_COLS;
        const double pv = o[0], prev = t > 0 ? out[(t - 1) * PQ_PORTFOLIO_COLS] : initial_total;
        o[1] = pv - prev;
        o[2] = (prev > 0.0) ? (pv - prev) / prev * 100.0 : 0.0;
        o[3] = pv - initial_total;
        o[4] = (initial_total > 0.0) ? (pv - initial_total) / initial_total * 100.0 : 0.0;
        o[5] = o[6] = o[7] = o[8] = o[9] = 0.0;
        if (bm) {
            double pb = t > 0 ? bm[t - 1] : bm[0];
            o[5] = (t > 0 && pb > 0.0) ? (bm[t] - pb) / pb * 100.0 : 0.0;
            o[6] = o[2] - o[5];
            o[7] = o[4] - ((bm[0] > 0.0) ? (bm[t] - bm[0]) / bm[0] * 100.0 : 0.0);
        }
    }
    __syncthreads();
    if (bm && T > 0) {
        __shared__ double beta_s;
        if (threadIdx.x == 0) {
            double sr = 0.0, sb = 0.0;
            for (int64_t t = 0; t < T; t++) { sr += out[t * PQ_PORTFOLIO_COLS + 2]; sb += out[t * PQ_PORTFOLIO_COLS + 5]; }
            double mr = sr / (double)T, mb = sb / (double)T, cv = 0.0, bv = 0.0;
            for (int64_t t = 0; t < T; t++) {
                double dr = out[t * PQ_PORTFOLIO_COLS + 2] - mr, db = out[t * PQ_PORTFOLIO_COLS + 5] - mb;
                cv += dr * db; bv += db * db;
            }
            double dof = fmax((double)T - 1.0, 1.0);
            cv /= dof; bv /= dof;
            beta_s = (bv > 0.0) ? cv / bv : 0.0;
        }
        __syncthreads();
        for (int64_t t = threadIdx.x; t < T; t += 256) out[t * PQ_PORTFOLIO_COLS + 8] = beta_s;
    }
}

template <bool MACD_SIGNALS, bool SIGNALS_ONLY>
__global__ __launch_bounds__(SEQ_BLOCK) void backtest_kernel(BtArgs a, Dims d) {
    const int64_t s = (int64_t)blockIdx.x * SEQ_BLOCK + threadIdx.x;
    if (s >= d.n) return;
    backtest_body<MACD_SIGNALS, SIGNALS_ONLY>(a, d, s);
}
constexpr int SEQ_ID_BACKTEST = 60; // + (MACD_SIGNALS ? 1 : 0)
pq_status rec_add_backtest(pq_ctx *ctx, const pq_batch *b, int kind, const BtArgs &a);
