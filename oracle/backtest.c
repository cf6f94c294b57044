/* backtest.c -- CPU ORACLE (test infrastructure) for src/backtest/vectorized.rs and
 * src/backtest/metrics.rs, the D-8 MACD-cross strategy, and the SURVEY 8(d) synthetic generator.
 * Compile with -ffp-contract=off. */
#include "pqo_common.h"

/* metrics.rs:7-152 calculate_summary.  Sums are plain left-to-right (iter().sum()). */
void pqo_summary(const double *equity, const double *bench, int64_t n, int64_t n_bench,
                 double initial_capital, int64_t trades, int64_t wins, double *s) {
    for (int k = 0; k < 8; k++) s[k] = 0.0;
    if (n == 0) return;                                                                   /* :17-19 */
    const double DAYS = 252.0, RF = 0.03;
    double max_dd = 0.0, max_eq = initial_capital, prev = initial_capital;
    double *ret = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < n; i++) {                                                     /* :26-49 */
        double e = equity[i];
        if (e > max_eq) max_eq = e;
        double dd = (max_eq > 0.0) ? (max_eq - e) / max_eq : 0.0;
        if (dd > max_dd) max_dd = dd;
        ret[i] = (prev > 0.0) ? (e - prev) / prev : 0.0;
        prev = e;
    }
    double final_equity = equity[n - 1];
    double total_return = (final_equity - initial_capital) / initial_capital;             /* :52 */
    double ann = (total_return > -1.0) ? pow(1.0 + total_return, DAYS / (double)n) - 1.0 : -1.0; /* :54-58 */
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) sum += ret[i];
    double mean = sum / (double)n;                                                        /* :60 */
    double dof = RMAX((double)n - 1.0, 1.0);                                              /* :61 */
    double vs = 0.0;
    for (int64_t i = 0; i < n; i++) { double d = ret[i] - mean; vs += d * d; }            /* :63-67 powi(2) */
    double var = vs / dof;
    double vol = sqrt(var) * sqrt(DAYS);                                                  /* :69 */
    double sharpe = (vol > 0.0) ? (ann - RF) / vol : 0.0;                                 /* :71-75 */
    double win_rate = (trades > 0) ? (double)wins / (double)trades : 0.0;
    double alpha = 0.0, beta = 0.0;
    if (bench && n_bench == n) {                                                          /* :86 */
        double *br = (double *)malloc(sizeof(double) * (size_t)n);
        double pb = bench[0];
        for (int64_t i = 0; i < n; i++) {
            br[i] = (pb > 0.0) ? (bench[i] - pb) / pb : 0.0;                              /* :91-98 */
            pb = bench[i];
        }
        double bs = 0.0;
        for (int64_t i = 0; i < n; i++) bs += br[i];
        double bmean = bs / (double)n;
        double bv = 0.0, cv = 0.0;
        for (int64_t i = 0; i < n; i++) { double d = br[i] - bmean; bv += d * d; }
        bv /= dof;
        for (int64_t i = 0; i < n; i++) cv += (ret[i] - mean) * (br[i] - bmean);          /* :109-116 */
        cv /= dof;
        if (bv > 0.0) beta = cv / bv;
        double b0 = bench[0], b1 = bench[n - 1];
        double btr = (b0 > 0.0) ? (b1 - b0) / b0 : 0.0;
        double bann = (btr > -1.0) ? pow(1.0 + btr, DAYS / (double)n) - 1.0 : -1.0;
        alpha = ann - (RF + beta * (bann - RF));                                          /* :138-139 */
        free(br);
    }
    s[0] = ann; s[1] = max_dd; s[2] = alpha; s[3] = beta; s[4] = sharpe;
    s[5] = RMAX(total_return, 0.0); s[6] = win_rate; s[7] = (double)trades;               /* :142-149 */
    free(ret);
}

/* vectorized.rs:69-224 VectorizedBacktester::run.  price null -> f64::NAN (:70-78): a null price row arrives here as PQO null and
 * becomes the plain NaN before anything is computed from it, so the row's equity is a NaN VALUE (not a null), as in the reference. */
void pqo_backtest(const double *price, const uint8_t *buy, const uint8_t *sell,
                  const double *benchmark, int64_t n, const pqo_bt_params *prm,
                  double *position, double *cash, double *equity, double *summary) {
    double pos = 0.0, avail = prm->initial_capital, peak = prm->initial_capital, entry_cost = 0.0;
    int64_t trades = 0, wins = 0;
    for (int64_t i = 0; i < n; i++) {
        double px = price[i];
        if (pqo_isnull(px)) px = NAN;                                                     /* :70-78 */
        if (isnan(px) || px <= 0.0) {                                                     /* :141-144 */
            position[i] = pos; cash[i] = avail; equity[i] = avail + pos * px;
            continue;
        }
        if (buy[i] && pos == 0.0) {                                                       /* :146 */
            double exec = px + prm->buy_slippage;
            double cur_eq = avail + pos * px;
            double deploy = cur_eq * prm->position_size;
            double qty = floor(deploy / exec);
            if (qty > 0.0) {
                double cost = qty * exec;
                double fee = RMAX(cost * prm->buy_commission_rate, prm->min_commission);
                pos += qty;
                avail -= cost + fee;                                                      /* :158 */
                entry_cost = pos * px;
                trades += 1;
            }
        } else if (sell[i] && pos > 0.0) {                                                /* :162 */
            double exec = px - prm->sell_slippage;
            double revenue = pos * exec;
            double fee = RMAX(revenue * prm->sell_commission_rate, prm->min_commission);
            double net = revenue - fee;
            if (net > entry_cost) wins += 1;
            avail += net;
            pos = 0.0;
        }
        double eq = avail + pos * px;                                                     /* :177 */
        if (eq > peak) peak = eq;
        position[i] = pos; cash[i] = avail; equity[i] = eq;
    }
    if (summary) pqo_summary(equity, benchmark, n, benchmark ? n : 0, prm->initial_capital, trades, wins, summary);
}

/* D-8 (README.md:912-917 Strategy.macd is spec-only): on the reference MACD (momentum.rs:250-283)
 *   buy[i]  = macd[i-1] <= signal[i-1] && macd[i] > signal[i]
 *   sell[i] = macd[i-1] >= signal[i-1] && macd[i] < signal[i]
 * false wherever any of the four values is null (i.e. rows < slow). */
void pqo_macd_cross_signals(const double *close, int64_t n, int64_t fast, int64_t slow,
                            int64_t sig, uint8_t *buy, uint8_t *sell) {
    size_t m = (size_t)(n > 0 ? n : 1);
    double *md = (double *)malloc(8 * m), *sg = (double *)malloc(8 * m), *hs = (double *)malloc(8 * m);
    pqo_macd(close, n, fast, slow, sig, md, sg, hs);
    for (int64_t i = 0; i < n; i++) {
        buy[i] = sell[i] = 0;
        if (i == 0) continue;
        if (pqo_isnull(md[i]) || pqo_isnull(sg[i]) || pqo_isnull(md[i - 1]) || pqo_isnull(sg[i - 1])) continue;
        buy[i] = (md[i - 1] <= sg[i - 1]) && (md[i] > sg[i]);
        sell[i] = (md[i - 1] >= sg[i - 1]) && (md[i] < sg[i]);
    }
    free(md); free(sg); free(hs);
}

/* ---------------------------------------------------------------------------------------------------------
 * the README's multi-symbol `Backtest` with leverage / margin call / interest and 100-share lots
 * (README.md:346-640).  README-only: there is no source, so decision D-10 below IS the specification.
 *
 * D-10, per symbol with its own capital pool (README.md:17,405), day t with price p:
 *   0. p null / NaN / <= 0: no trading; the position is valued at the last valid price (0 before the first).
 *   1. interest: debt += debt * (interest_rate / 252)                    (financing cost compounds daily)
 *   2. margin call: holding with debt > 0 and  cash + shares*p - debt  <  margin_call_threshold * shares*p
 *      -> forced sale today (reason 2), same mechanics as a signalled sale
 *   3. else sell signal while holding -> sale (reason 1): exec = p*(1-slippage), fee = max(rev*rate, min_commission),
 *      cash += rev - fee - debt, debt = 0; pnl = (rev - fee) - entry_outlay; a win iff pnl > 0
 *   4. else buy signal while flat: power = cash * position_size * leverage, exec = p*(1+slippage),
 *      lots = floor(power / (exec*100)), reduced while lots*100*exec + fee > cash*leverage; outlay = cost + fee;
 *      debt = max(outlay - cash, 0); cash = max(cash - outlay, 0)
 *   daily record: cash_net = cash - debt, stock_value = shares * p, total_value = cash_net + stock_value.
 * Trade records: the first max_trades closed trades per symbol (trade_count keeps counting).  A position still open
 * on the last day is not a trade.  The per-symbol summary is calculate_summary (metrics.rs:7-152) on total_value with
 * an optional single benchmark series shared by all symbols (README.md:366).                                        */
void pqo_backtest_leveraged(const double *price, const uint8_t *buy, const uint8_t *sell, const double *benchmark,
                            int64_t n, const pqo_lev_params *prm, double *cash_net, double *stock_value,
                            double *total_value, int32_t max_trades, int32_t *trade_count, int32_t *entry_day,
                            int32_t *exit_day, double *entry_price, double *exit_price, double *quantity, double *pnl,
                            double *pnl_pct, int32_t *reason, double *summary) {
    double cash = prm->initial_capital, debt = 0.0, shares = 0.0, last_px = 0.0;
    double e_outlay = 0.0, e_price = 0.0;
    int64_t e_day = 0, trades = 0, wins = 0;
    const double daily_rate = prm->interest_rate / 252.0;
    for (int64_t t = 0; t < n; t++) {
        double p = price[t];
        const int valid = !(isnan(p) || p <= 0.0);
        if (debt > 0.0) debt += debt * daily_rate;
        if (valid) {
            last_px = p;
            int do_sell = 0;
            if (shares > 0.0) {
                if (debt > 0.0 && cash + shares * p - debt < prm->margin_call_threshold * (shares * p)) do_sell = 2;
                else if (sell[t]) do_sell = 1;
            }
            if (do_sell) {
                double exec = p * (1.0 - prm->slippage);
                double rev = shares * exec;
                double fee = RMAX(rev * prm->commission_rate, prm->min_commission);
                double net = rev - fee;
                double gain = net - e_outlay;
                if (trades < max_trades && entry_day) {
                    entry_day[trades] = (int32_t)e_day; exit_day[trades] = (int32_t)t;
                    entry_price[trades] = e_price; exit_price[trades] = exec; quantity[trades] = shares;
                    pnl[trades] = gain; pnl_pct[trades] = gain / e_outlay * 100.0; reason[trades] = do_sell;
                }
                trades += 1;
                if (gain > 0.0) wins += 1;
                cash = cash + net - debt;
                debt = 0.0;
                shares = 0.0;
            } else if (buy[t] && shares == 0.0) {
                double exec = p * (1.0 + prm->slippage);
                double power = cash * prm->position_size * prm->leverage;
                double lots = floor(power / (exec * 100.0));
                double cost = 0.0, fee = 0.0;
                while (lots > 0.0) {
                    cost = lots * 100.0 * exec;
                    fee = RMAX(cost * prm->commission_rate, prm->min_commission);
                    if (cost + fee <= cash * prm->leverage) break;
                    lots -= 1.0;
                }
                if (lots > 0.0) {
                    double outlay = cost + fee;
                    debt = RMAX(outlay - cash, 0.0);
                    cash = RMAX(cash - outlay, 0.0);
                    shares = lots * 100.0;
                    e_outlay = outlay; e_price = exec; e_day = t;
                }
            }
        }
        double sv = shares * last_px;
        cash_net[t] = cash - debt;
        stock_value[t] = sv;
        total_value[t] = (cash - debt) + sv;
    }
    if (trade_count) *trade_count = (int32_t)trades;
    if (summary) pqo_summary(total_value, benchmark, n, benchmark ? n : 0, prm->initial_capital, trades, wins, summary);
}

#define PQO_SUM_BLOCK 256 /* cross-sectional sums (portfolio value, IC) are defined over blocks of this many symbols */
/* README.md:455-477 get_performance_metrics over all symbols.  total_value: [n_sym][n] row-major.  out: [n][10] =
 * portfolio_value, daily_pnl, daily_return_pct, cumulative_pnl, cumulative_return_pct, benchmark_return_pct, alpha_pct,
 * relative_return_pct, beta, 0 (columns 5-8 are 0 without a benchmark).  Sums run over blocks of 256 symbols (see below);
 * beta = sample covariance / sample variance (dof max(n-1, 1)) of the daily percentage returns, one value for all rows. */
void pqo_portfolio_metrics(const double *total_value, int64_t n_sym, int64_t n, int64_t stride, double initial_total,
                           const double *benchmark, double *out) {
    double prev = initial_total;
    for (int64_t t = 0; t < n; t++) {
        double pv = 0.0; /* blocks of PQO_SUM_BLOCK symbols: ascending inside a block, block sums added in ascending order */
        for (int64_t s0 = 0; s0 < n_sym; s0 += PQO_SUM_BLOCK) {
            double part = 0.0;
            for (int64_t s = s0; s < n_sym && s < s0 + PQO_SUM_BLOCK; s++) part += total_value[s * stride + t];
            pv += part;
        }
        double *o = out + t * 10;
        o[0] = pv;
        o[1] = pv - prev;
        o[2] = (prev > 0.0) ? (pv - prev) / prev * 100.0 : 0.0;
        o[3] = pv - initial_total;
        o[4] = (initial_total > 0.0) ? (pv - initial_total) / initial_total * 100.0 : 0.0;
        o[5] = o[6] = o[7] = o[8] = o[9] = 0.0;
        if (benchmark) {
            double pb = t > 0 ? benchmark[t - 1] : benchmark[0];
            o[5] = (t > 0 && pb > 0.0) ? (benchmark[t] - pb) / pb * 100.0 : 0.0;
            o[6] = o[2] - o[5];
            o[7] = o[4] - ((benchmark[0] > 0.0) ? (benchmark[t] - benchmark[0]) / benchmark[0] * 100.0 : 0.0);
        }
        prev = pv;
    }
    if (benchmark && n > 0) {
        double sr = 0.0, sb = 0.0;
        for (int64_t t = 0; t < n; t++) { sr += out[t * 10 + 2]; sb += out[t * 10 + 5]; }
        double mr = sr / (double)n, mb = sb / (double)n, cv = 0.0, bv = 0.0;
        for (int64_t t = 0; t < n; t++) {
            double dr = out[t * 10 + 2] - mr, db = out[t * 10 + 5] - mb;
            cv += dr * db; bv += db * db;
        }
        double dof = RMAX((double)n - 1.0, 1.0);
        cv /= dof; bv /= dof;
        double beta = (bv > 0.0) ? cv / bv : 0.0;
        for (int64_t t = 0; t < n; t++) out[t * 10 + 8] = beta;
    }
}

/* ---------------------------------------------------------------------------------------------------------
 * the README's `Strategy` signal generators (README.md:862-994; README-only => decision D-11).
 * Three row-local rules turn indicator columns into the uint8 buy/sell columns of the backtests; a rule is false
 * wherever one of the values it looks at is null (or, for rules using row i-1, on row 0):
 *   cross(a, b):            buy  = a[i-1] <= b[i-1] && a[i] >  b[i]       (golden cross: MA, MACD, STOCH %K/%D, trend)
 *                           sell = a[i-1] >= b[i-1] && a[i] <  b[i]
 *   band(x, lower, upper):  buy  = x[i-1] <  lower  && x[i] >= lower      (leaves the oversold zone: RSI, CCI, STOCH)
 *                           sell = x[i-1] >  upper  && x[i] <= upper      (leaves the overbought zone)
 *   channel(p, lo, hi, mode 0 = reversion): buy = p[i] < lo[i] && p[i-1] >= lo[i-1]; sell = p[i] > hi[i] && p[i-1] <= hi[i-1]
 *                           (mode 1 = breakout): buy = p[i] > hi[i-1];  sell = p[i] < lo[i-1]       (Donchian on prior bars) */
void pqo_cross_signals(const double *a, const double *b, int64_t n, uint8_t *buy, uint8_t *sell) {
    for (int64_t i = 0; i < n; i++) {
        buy[i] = sell[i] = 0;
        if (i == 0 || pqo_isnull(a[i]) || pqo_isnull(b[i]) || pqo_isnull(a[i - 1]) || pqo_isnull(b[i - 1])) continue;
        buy[i] = (a[i - 1] <= b[i - 1]) && (a[i] > b[i]);
        sell[i] = (a[i - 1] >= b[i - 1]) && (a[i] < b[i]);
    }
}
void pqo_band_signals(const double *x, int64_t n, double lower, double upper, uint8_t *buy, uint8_t *sell) {
    for (int64_t i = 0; i < n; i++) {
        buy[i] = sell[i] = 0;
        if (i == 0 || pqo_isnull(x[i]) || pqo_isnull(x[i - 1])) continue;
        buy[i] = (x[i - 1] < lower) && (x[i] >= lower);
        sell[i] = (x[i - 1] > upper) && (x[i] <= upper);
    }
}
void pqo_channel_signals(const double *p, const double *lo, const double *hi, int64_t n, int mode, uint8_t *buy,
                         uint8_t *sell) {
    for (int64_t i = 0; i < n; i++) {
        buy[i] = sell[i] = 0;
        if (i == 0 || pqo_isnull(p[i])) continue;
        if (mode == 0) {
            if (pqo_isnull(p[i - 1]) || pqo_isnull(lo[i]) || pqo_isnull(hi[i]) || pqo_isnull(lo[i - 1]) || pqo_isnull(hi[i - 1])) continue;
            buy[i] = (p[i] < lo[i]) && (p[i - 1] >= lo[i - 1]);
            sell[i] = (p[i] > hi[i]) && (p[i - 1] <= hi[i - 1]);
        } else {
            if (pqo_isnull(lo[i - 1]) || pqo_isnull(hi[i - 1])) continue;
            buy[i] = p[i] > hi[i - 1];
            sell[i] = p[i] < lo[i - 1];
        }
    }
}

/* ---------------------------------------------------------------------------------------------------------
 * cross-sectional factor evaluation -- Factor.ic / rank_ic / rolling_ic (README.md:1429-1430,
 * :1480-1482, :1626-1634; README-only => decision D-12).  factor and fwd_return are symbol-major [n_sym][stride].
 * Per day t the cross-section is the set of symbols where BOTH values are non-null and finite (pairwise deletion), n_t
 * of them; fewer than 2, or a zero variance on either side, gives a null IC.
 *   method 0 (IC, Pearson):  mx = sum(x)/n, my = sum(y)/n; sxy = sum((x-mx)*(y-my)), sxx, syy -- every sum taken over
 *       blocks of 256 symbols (ascending inside a block, block sums added in ascending order);  ic = sxy / (sqrt(sxx) * sqrt(syy))
 *   method 1 (Rank IC, Spearman): x and y are replaced by their average ranks (1-based, ties share the mean rank)
 *       inside the day's cross-section; the sums Sx, Sy, Sxx, Syy, Sxy of ranks are exact in f64 (half-integers, n <= 2^20),
 *       and  ic = (n*Sxy - Sx*Sy) / (sqrt(n*Sxx - Sx*Sx) * sqrt(n*Syy - Sy*Sy))
 * rolling_ic(window w): row t = mean of ic[t-w+1 .. t] if all w values are non-null, else null (sum in ascending order);
 * rolling_ir = that mean / sample standard deviation (dof w-1) of the same w values, null if w < 2 or the deviation is 0. */
typedef struct { double v; int64_t i; } pqo_kv;
static int pqo_kv_cmp(const void *a, const void *b) {
    double x = ((const pqo_kv *)a)->v, y = ((const pqo_kv *)b)->v;
    return (x > y) - (x < y);
}
static void pqo_avg_ranks(const double *v, int64_t n, double *rank, pqo_kv *tmp) {
    for (int64_t k = 0; k < n; k++) { tmp[k].v = v[k]; tmp[k].i = k; }
    qsort(tmp, (size_t)n, sizeof(pqo_kv), pqo_kv_cmp);
    for (int64_t a = 0; a < n;) {
        int64_t b = a + 1;
        while (b < n && tmp[b].v == tmp[a].v) b++;
        double r = ((double)(a + 1) + (double)b) / 2.0; /* mean of the 1-based positions a+1 .. b */
        for (int64_t k = a; k < b; k++) rank[tmp[k].i] = r;
        a = b;
    }
}
void pqo_factor_ic(const double *factor, const double *ret, int64_t n_sym, int64_t n, int64_t stride, int method,
                   double *ic, int32_t *n_valid) {
    size_t m = (size_t)(n_sym > 0 ? n_sym : 1);
    double *x = (double *)malloc(8 * m), *y = (double *)malloc(8 * m), *rx = (double *)malloc(8 * m), *ry = (double *)malloc(8 * m);
    pqo_kv *tmp = (pqo_kv *)malloc(sizeof(pqo_kv) * m);
    for (int64_t t = 0; t < n; t++) {
        int64_t k = 0;
        for (int64_t s = 0; s < n_sym; s++) {
            double a = factor[s * stride + t], b = ret[s * stride + t];
            if (pqo_isnull(a) || pqo_isnull(b) || !isfinite(a) || !isfinite(b)) continue;
            x[k] = a; y[k] = b; k++;
        }
        if (n_valid) n_valid[t] = (int32_t)k;
        ic[t] = pqo_null();
        if (k < 2) continue;
        const double nn = (double)k;
        if (method == 0) { /* blocks of PQO_SUM_BLOCK SYMBOLS (valid or not): partial sums in ascending order, then the partials */
            double sx = 0.0, sy = 0.0;
            for (int64_t s0 = 0; s0 < n_sym; s0 += PQO_SUM_BLOCK) {
                double px = 0.0, py = 0.0;
                for (int64_t s = s0; s < n_sym && s < s0 + PQO_SUM_BLOCK; s++) {
                    double a = factor[s * stride + t], b = ret[s * stride + t];
                    if (pqo_isnull(a) || pqo_isnull(b) || !isfinite(a) || !isfinite(b)) continue;
                    px += a; py += b;
                }
                sx += px; sy += py;
            }
            double mx = sx / nn, my = sy / nn, sxy = 0.0, sxx = 0.0, syy = 0.0;
            for (int64_t s0 = 0; s0 < n_sym; s0 += PQO_SUM_BLOCK) {
                double pxy = 0.0, pxx = 0.0, pyy = 0.0;
                for (int64_t s = s0; s < n_sym && s < s0 + PQO_SUM_BLOCK; s++) {
                    double a = factor[s * stride + t], b = ret[s * stride + t];
                    if (pqo_isnull(a) || pqo_isnull(b) || !isfinite(a) || !isfinite(b)) continue;
                    double dx = a - mx, dy = b - my;
                    pxy += dx * dy; pxx += dx * dx; pyy += dy * dy;
                }
                sxy += pxy; sxx += pxx; syy += pyy;
            }
            if (sxx > 0.0 && syy > 0.0) ic[t] = sxy / (sqrt(sxx) * sqrt(syy));
        } else {
            pqo_avg_ranks(x, k, rx, tmp);
            pqo_avg_ranks(y, k, ry, tmp);
            double Sx = 0.0, Sy = 0.0, Sxx = 0.0, Syy = 0.0, Sxy = 0.0;
            for (int64_t j = 0; j < k; j++) { Sx += rx[j]; Sy += ry[j]; Sxx += rx[j] * rx[j]; Syy += ry[j] * ry[j]; Sxy += rx[j] * ry[j]; }
            double vx = nn * Sxx - Sx * Sx, vy = nn * Syy - Sy * Sy;
            if (vx > 0.0 && vy > 0.0) ic[t] = (nn * Sxy - Sx * Sy) / (sqrt(vx) * sqrt(vy));
        }
    }
    free(x); free(y); free(rx); free(ry); free(tmp);
}
void pqo_rolling_ic(const double *ic, int64_t n, int64_t w, double *rolling_ic, double *rolling_ir) {
    for (int64_t t = 0; t < n; t++) {
        rolling_ic[t] = rolling_ir[t] = pqo_null();
        if (w <= 0 || t + 1 < w) continue;
        int ok = 1;
        double sum = 0.0;
        for (int64_t j = t - w + 1; j <= t; j++) { if (pqo_isnull(ic[j])) { ok = 0; break; } sum += ic[j]; }
        if (!ok) continue;
        double mean = sum / (double)w;
        rolling_ic[t] = mean;
        if (w < 2) continue;
        double vs = 0.0;
        for (int64_t j = t - w + 1; j <= t; j++) { double d = ic[j] - mean; vs += d * d; }
        double sd = sqrt(vs / (double)(w - 1));
        if (sd > 0.0) rolling_ir[t] = mean / sd;
    }
}

/* SURVEY.md 8(d) generator: splitmix64-driven, transcendental-free, bit-reproducible. */
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static double u01(uint64_t seed, uint64_t k) { return (double)(splitmix64(seed + k) >> 11) * (1.0 / 9007199254740992.0); }

void pqo_gen_ohlcv(uint64_t seed, int64_t n_sym, int64_t T, int mode,
                   double *open, double *high, double *low, double *close, double *volume) {
    double ret_rng = mode ? 0.16 : 0.04, gap_rng = mode ? 0.04 : 0.01, sh_rng = mode ? 0.08 : 0.01;
    for (int64_t s = 0; s < n_sym; s++) {
        double prev = 10.0 + (double)(s % 90);
        for (int64_t t = 0; t < T; t++) {
            uint64_t k = 5ULL * (uint64_t)(s * T + t);
            double ret = (u01(seed, k) - 0.5) * ret_rng;
            double c = prev * (1.0 + ret);
            double o = prev * (1.0 + (u01(seed, k + 1) - 0.5) * gap_rng);
            double h = (o > c ? o : c) * (1.0 + u01(seed, k + 2) * sh_rng);
            double l = (o < c ? o : c) * (1.0 - u01(seed, k + 3) * sh_rng);
            double v = floor(1e5 + u01(seed, k + 4) * 9e5);
            int64_t i = s * T + t;
            open[i] = o; high[i] = h; low[i] = l; close[i] = c; volume[i] = v;
            prev = c;
        }
    }
}
