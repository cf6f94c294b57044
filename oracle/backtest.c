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

/* vectorized.rs:69-224 VectorizedBacktester::run.  price null -> NaN (:70-78) is applied by the
 * caller: a null price row arrives here as PQO null (a NaN) and takes the is_nan() branch. */
void pqo_backtest(const double *price, const uint8_t *buy, const uint8_t *sell,
                  const double *benchmark, int64_t n, const pqo_bt_params *prm,
                  double *position, double *cash, double *equity, double *summary) {
    double pos = 0.0, avail = prm->initial_capital, peak = prm->initial_capital, entry_cost = 0.0;
    int64_t trades = 0, wins = 0;
    for (int64_t i = 0; i < n; i++) {
        double px = price[i];
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
