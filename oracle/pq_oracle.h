/*
 * pq_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar C restatement of the reference's (Firstastor/polars-quant v1.0.0)
 * technical-indicator + per-symbol backtest algorithms.  Every function cites
 * the reference file:line it follows (paths relative to /root/reference).
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 * (tests/__init__.py:1-5 is a scratch TA-Lib call) and can be neither built
 * (no Rust toolchain; source has undefined helpers) nor imported (no polars,
 * no prebuilt .so) here.  The oracle is pinned only by hand-derivable KATs in
 * tests/test_oracle_kat.py; decisions where the reference is undefined are
 * D-1..D-9 in DESIGN.md.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this library.  The product path (polars_quant_amd) never links it.
 *
 * Conventions
 *   - one call = one series of n rows (a symbol); f64 in, f64 (or i32) out
 *   - a NULL row is encoded as the NaN bit pattern PQO_NULL_BITS (in and out)
 *   - must be compiled with -ffp-contract=off: Rust never contracts a*b+c;
 *     fma() is used exactly where the reference calls f64::mul_add
 */
#ifndef PQ_ORACLE_H
#define PQ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQO_NULL_BITS 0x7FF80000504E554CULL /* quiet NaN, payload "PNUL" */

/* ---- overlap (src/talib/overlap.rs) ---- */
void pqo_sma(const double *v, int64_t n, int64_t p, double *out);
void pqo_ema(const double *v, int64_t n, int64_t p, double *out);
void pqo_bbands(const double *v, int64_t n, int64_t p, double up, double dn,
                double *upper, double *middle, double *lower);
void pqo_dema(const double *v, int64_t n, int64_t p, double *out);
void pqo_tema(const double *v, int64_t n, int64_t p, double *out);
void pqo_t3(const double *v, int64_t n, int64_t p, double vfactor, double *out);
void pqo_trima(const double *v, int64_t n, int64_t p, double *out);
void pqo_wma(const double *v, int64_t n, int64_t p, double *out);
void pqo_kama(const double *v, int64_t n, int64_t p, double *out);
void pqo_ma(const double *v, int64_t n, int64_t p, int64_t matype, double *out);
void pqo_midpoint(const double *v, int64_t n, int64_t p, double *out);
void pqo_midprice(const double *h, const double *l, int64_t n, int64_t p, double *out);
void pqo_mama(const double *v, int64_t n, double fastlimit, double slowlimit,
              double *mama, double *fama);
void pqo_mavp(const double *v, const double *periods, int64_t n, int64_t minp,
              int64_t maxp, int64_t matype, double *out);
void pqo_sar(const double *h, const double *l, int64_t n, double accel, double maxv, double *out);
void pqo_sarext(const double *h, const double *l, int64_t n, double startvalue,
                double offsetonreverse, double ai_long, double a_long, double am_long,
                double ai_short, double a_short, double am_short, double *out);

/* ---- momentum (src/talib/momentum.rs) ---- */
void pqo_rma(const double *x, int64_t n, int64_t p, double *out); /* D-1 */
void pqo_adx(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out);
void pqo_adxr(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out);
void pqo_dx(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out);
void pqo_plus_di(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out);
void pqo_minus_di(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out);
void pqo_plus_dm(const double *h, const double *l, int64_t n, int64_t p, double *out);
void pqo_minus_dm(const double *h, const double *l, int64_t n, int64_t p, double *out);
void pqo_aroon(const double *h, const double *l, int64_t n, int64_t p, double *up, double *down);
void pqo_aroonosc(const double *h, const double *l, int64_t n, int64_t p, double *out);
void pqo_bop(const double *o, const double *h, const double *l, const double *c, int64_t n, double *out);
void pqo_cci(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out);
void pqo_cmo(const double *v, int64_t n, int64_t p, double *out);
void pqo_macd(const double *v, int64_t n, int64_t fast, int64_t slow, int64_t sig,
              double *macd, double *signal, double *hist);
void pqo_mfi(const double *h, const double *l, const double *c, const double *vol, int64_t n,
             int64_t p, double *out);
void pqo_mom(const double *v, int64_t n, int64_t p, double *out);
void pqo_roc(const double *v, int64_t n, int64_t p, double *out);
void pqo_rocp(const double *v, int64_t n, int64_t p, double *out);
void pqo_rocr(const double *v, int64_t n, int64_t p, double *out);
void pqo_rocr100(const double *v, int64_t n, int64_t p, double *out);
void pqo_returns(const double *v, int64_t n, int64_t period, int64_t method, double *out); /* README.md:46-75, D-13 */
void pqo_rolling_max(const double *v, int64_t n, int64_t w, double *out); /* momentum.py:182 (Polars rolling_max) */
void pqo_rolling_min(const double *v, int64_t n, int64_t w, double *out);
void pqo_rsi(const double *v, int64_t n, int64_t p, double *out);
void pqo_trix(const double *v, int64_t n, int64_t p, double *out);
void pqo_ultosc(const double *h, const double *l, const double *c, int64_t n, int64_t p1,
                int64_t p2, int64_t p3, double *out);
void pqo_willr(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out);
void pqo_apo(const double *v, int64_t n, int64_t fast, int64_t slow, int64_t matype, double *out);
void pqo_ppo(const double *v, int64_t n, int64_t fast, int64_t slow, int64_t matype, double *out);
/* python composites (python/polars_quant/talib/momentum.py) */
void pqo_macdext(const double *v, int64_t n, int64_t fast, int64_t fastmt, int64_t slow,
                 int64_t slowmt, int64_t sig, int64_t sigmt, double *macd, double *signal,
                 double *hist);
void pqo_macdfix(const double *v, int64_t n, int64_t sig, double *macd, double *signal, double *hist);
void pqo_stoch(const double *h, const double *l, const double *c, int64_t n, int64_t fastk,
               int64_t slowk, int64_t slowk_mt, int64_t slowd, int64_t slowd_mt, double *outk,
               double *outd);
void pqo_stochf(const double *h, const double *l, const double *c, int64_t n, int64_t fastk,
                int64_t fastd, int64_t fastd_mt, double *outk, double *outd);
void pqo_stochrsi(const double *v, int64_t n, int64_t p, int64_t fastk, int64_t fastd,
                  int64_t fastd_mt, double *outk, double *outd);

/* ---- volatility / volume / price ---- */
void pqo_trange(const double *h, const double *l, const double *c, int64_t n, double *out);
void pqo_atr(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out);
void pqo_natr(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out);
void pqo_ad(const double *h, const double *l, const double *c, const double *vol, int64_t n, double *out);
void pqo_adosc(const double *h, const double *l, const double *c, const double *vol, int64_t n,
               int64_t fast, int64_t slow, double *out);
void pqo_obv(const double *c, const double *vol, int64_t n, double *out);
void pqo_avgprice(const double *o, const double *h, const double *l, const double *c, int64_t n, double *out);
void pqo_medprice(const double *h, const double *l, int64_t n, double *out);
void pqo_typprice(const double *h, const double *l, const double *c, int64_t n, double *out);
void pqo_wclprice(const double *h, const double *l, const double *c, int64_t n, double *out);

/* ---- cycle (src/talib/cycle.rs) ---- */
void pqo_ht_dcperiod(const double *v, int64_t n, double *out);
void pqo_ht_dcphase(const double *v, int64_t n, double *out);
void pqo_ht_phasor(const double *v, int64_t n, double *inphase, double *quadrature);
void pqo_ht_sine(const double *v, int64_t n, double *sine, double *leadsine);
void pqo_ht_trendline(const double *v, int64_t n, double *out);
/* int32 out; rows < 31 (or all rows if n < 32) are PQO_NULL_I32 */
#define PQO_NULL_I32 ((int32_t)0x80000000)
void pqo_ht_trendmode(const double *v, int64_t n, int32_t *out);

/* ---- patterns (src/talib/pattern.rs): id = index into pqo_pattern_names ---- */
#define PQO_N_PATTERNS 61
extern const char *const pqo_pattern_names[PQO_N_PATTERNS];
void pqo_pattern(int id, const double *o, const double *h, const double *l, const double *c,
                 int64_t n, double penetration, int32_t *out);

/* ---- backtest (src/backtest/vectorized.rs, metrics.rs) ---- */
typedef struct {
    double initial_capital, buy_slippage, sell_slippage, buy_commission_rate,
        sell_commission_rate, min_commission, position_size;
} pqo_bt_params;
/* summary[8] order: annualized_return, max_drawdown, alpha, beta, sharpe_ratio,
 * max_profit, win_rate, total_trades (metrics.rs:142-149) */
void pqo_backtest(const double *price, const uint8_t *buy, const uint8_t *sell,
                  const double *benchmark /* NULL or n rows */, int64_t n,
                  const pqo_bt_params *prm, double *position, double *cash, double *equity,
                  double *summary);
void pqo_summary(const double *equity, const double *benchmark, int64_t n, int64_t n_bench,
                 double initial_capital, int64_t trades, int64_t wins, double *summary);
/* D-8 MACD-cross strategy signals: buy = macd crosses above signal, sell = below */
void pqo_macd_cross_signals(const double *close, int64_t n, int64_t fast, int64_t slow,
                            int64_t sig, uint8_t *buy, uint8_t *sell);

/* ---- README multi-symbol Backtest with leverage (decision D-10, see backtest.c) ---- */
typedef struct {
    double initial_capital, position_size, leverage, margin_call_threshold, interest_rate, commission_rate,
        min_commission, slippage;
} pqo_lev_params;
void pqo_backtest_leveraged(const double *price, const uint8_t *buy, const uint8_t *sell, const double *benchmark,
                            int64_t n, const pqo_lev_params *prm, double *cash_net, double *stock_value,
                            double *total_value, int32_t max_trades, int32_t *trade_count, int32_t *entry_day,
                            int32_t *exit_day, double *entry_price, double *exit_price, double *quantity, double *pnl,
                            double *pnl_pct, int32_t *reason, double *summary);
void pqo_portfolio_metrics(const double *total_value, int64_t n_sym, int64_t n, int64_t stride, double initial_total,
                           const double *benchmark, double *out /* [n][10] */);

/* ---- Strategy signal rules (decision D-11, see backtest.c) ---- */
void pqo_cross_signals(const double *a, const double *b, int64_t n, uint8_t *buy, uint8_t *sell);
void pqo_band_signals(const double *x, int64_t n, double lower, double upper, uint8_t *buy, uint8_t *sell);
void pqo_channel_signals(const double *p, const double *lo, const double *hi, int64_t n, int mode, uint8_t *buy, uint8_t *sell);

/* ---- Factor.ic / rank_ic / rolling_ic (decision D-12, see backtest.c) ---- */
void pqo_factor_ic(const double *factor, const double *fwd_return, int64_t n_sym, int64_t n, int64_t stride,
                   int method /*0 Pearson, 1 Spearman*/, double *ic /* [n] */, int32_t *n_valid /* [n] or NULL */);
void pqo_rolling_ic(const double *ic, int64_t n, int64_t window, double *rolling_ic, double *rolling_ir);

/* ---- synthetic OHLCV generator ---- */
void pqo_gen_ohlcv(uint64_t seed, int64_t n_sym, int64_t T, int mode /*0 plain,1 pattern-rich*/,
                   double *open, double *high, double *low, double *close, double *volume);

#ifdef __cplusplus
}
#endif
#endif
