/* Memory-safety driver for the CPU oracle (test infrastructure): every entry point of pq_oracle.h over series lengths 0 ..
 * 70 (below / at / above every warm-up), periods 0, 1, n-1, n, n+1 and beyond, with and without nulls, every output buffer
 * allocated at EXACTLY n elements.  Built with -fsanitize=address,undefined by `make -C oracle asan` (the oracle uses
 * hand-rolled deques and malloc per call: overlap.c).  Prints a checksum so that the work cannot be optimised away. */
#include "pq_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double acc = 0.0;
static void fold(const double *a, int64_t n) { for (int64_t i = 0; i < n; i++) if (isfinite(a[i])) acc += a[i] * 1e-9; }
static void foldi(const int32_t *a, int64_t n) { for (int64_t i = 0; i < n; i++) acc += a[i] * 1e-9; }
static double *dbl(int64_t n) { return (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 0) + (n > 0 ? 0 : 1)); }

int main(void) {
    const int64_t lens[] = {0, 1, 2, 3, 5, 6, 7, 13, 14, 15, 27, 28, 29, 30, 31, 32, 33, 59, 60, 61, 70};
    for (size_t li = 0; li < sizeof lens / sizeof lens[0]; li++) {
        const int64_t n = lens[li];
        for (int withnull = 0; withnull < 2; withnull++) {
            double *o = dbl(n), *h = dbl(n), *l = dbl(n), *c = dbl(n), *v = dbl(n), *per = dbl(n);
            if (n > 0) pqo_gen_ohlcv(0xA5A50000ULL + (uint64_t)n, 1, n, (int)(li & 1), o, h, l, c, v);
            for (int64_t i = 0; i < n; i++) per[i] = (double)((i * 7) % 45) - 2.0;
            if (withnull) {
                uint64_t nb = PQO_NULL_BITS; double nul; memcpy(&nul, &nb, 8);
                for (int64_t i = 0; i < n; i += 5) { c[i] = nul; if (i % 2) h[i] = nul; if (i % 3 == 0) v[i] = nul; }
                if (n > 2) { o[1] = nul; l[n - 1] = nul; per[0] = nul; }
            }
            double *a = dbl(n), *b = dbl(n), *d = dbl(n);
            int32_t *ai = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
            uint8_t *u1 = (uint8_t *)malloc((size_t)(n > 0 ? n : 1)), *u2 = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
            const int64_t ps[] = {0, 1, 2, 3, 5, 9, 14, 30, n - 1, n, n + 1, 2 * n + 3};
            for (size_t pi = 0; pi < sizeof ps / sizeof ps[0]; pi++) {
                const int64_t p = ps[pi] < 0 ? 0 : ps[pi];
                pqo_sma(c, n, p, a); fold(a, n); pqo_ema(c, n, p, a); fold(a, n);
                pqo_bbands(c, n, p, 2.0, 2.0, a, b, d); fold(a, n); fold(b, n); fold(d, n);
                pqo_dema(c, n, p, a); fold(a, n); pqo_tema(c, n, p, a); fold(a, n); pqo_t3(c, n, p, 0.7, a); fold(a, n);
                pqo_trima(c, n, p, a); fold(a, n); pqo_wma(c, n, p, a); fold(a, n); pqo_kama(c, n, p, a); fold(a, n);
                for (int64_t mt = 0; mt < 10; mt++) { pqo_ma(c, n, p, mt, a); fold(a, n); }
                pqo_midpoint(c, n, p, a); fold(a, n); pqo_midprice(h, l, n, p, a); fold(a, n);
                for (int64_t mt = 0; mt < 9; mt++) { pqo_mavp(c, per, n, p / 2, p, mt, a); fold(a, n); }
                pqo_rma(c, n, p, a); fold(a, n);
                pqo_adx(h, l, c, n, p, a); fold(a, n); pqo_adxr(h, l, c, n, p, a); fold(a, n); pqo_dx(h, l, c, n, p, a); fold(a, n);
                pqo_plus_di(h, l, c, n, p, a); fold(a, n); pqo_minus_di(h, l, c, n, p, a); fold(a, n);
                pqo_plus_dm(h, l, n, p, a); fold(a, n); pqo_minus_dm(h, l, n, p, a); fold(a, n);
                pqo_aroon(h, l, n, p, a, b); fold(a, n); fold(b, n); pqo_aroonosc(h, l, n, p, a); fold(a, n);
                pqo_cci(h, l, c, n, p, a); fold(a, n); pqo_cmo(c, n, p, a); fold(a, n);
                pqo_macd(c, n, p, p + 3, p / 2 + 1, a, b, d); fold(a, n); fold(b, n); fold(d, n);
                pqo_mfi(h, l, c, v, n, p, a); fold(a, n);
                pqo_mom(c, n, p, a); fold(a, n); pqo_roc(c, n, p, a); fold(a, n); pqo_rocp(c, n, p, a); fold(a, n);
                pqo_rocr(c, n, p, a); fold(a, n); pqo_rocr100(c, n, p, a); fold(a, n);
                pqo_returns(c, n, p, 0, a); fold(a, n); pqo_returns(c, n, p, 1, a); fold(a, n);
                pqo_rolling_max(h, n, p, a); fold(a, n); pqo_rolling_min(l, n, p, a); fold(a, n);
                pqo_rsi(c, n, p, a); fold(a, n); pqo_trix(c, n, p, a); fold(a, n);
                pqo_ultosc(h, l, c, n, p, p + 2, 2 * p + 1, a); fold(a, n); pqo_willr(h, l, c, n, p, a); fold(a, n);
                for (int64_t mt = 0; mt < 9; mt += 4) {
                    pqo_apo(c, n, p, p + 4, mt, a); fold(a, n); pqo_ppo(c, n, p, p + 4, mt, a); fold(a, n);
                    pqo_macdext(c, n, p, mt, p + 4, (mt + 1) % 9, p / 2 + 1, (mt + 2) % 9, a, b, d); fold(a, n); fold(b, n); fold(d, n);
                    pqo_stoch(h, l, c, n, p, p / 2 + 1, mt, 3, (mt + 1) % 9, a, b); fold(a, n); fold(b, n);
                    pqo_stochf(h, l, c, n, p, p / 2 + 1, mt, a, b); fold(a, n); fold(b, n);
                    pqo_stochrsi(c, n, p, p / 2 + 1, 3, mt, a, b); fold(a, n); fold(b, n);
                }
                pqo_macdfix(c, n, p, a, b, d); fold(a, n);
                pqo_atr(h, l, c, n, p, a); fold(a, n); pqo_natr(h, l, c, n, p, a); fold(a, n);
                pqo_adosc(h, l, c, v, n, p, p + 7, a); fold(a, n);
                pqo_macd_cross_signals(c, n, p, p + 5, 3, u1, u2);
                pqo_rolling_ic(c, n, p, a, b); fold(a, n); fold(b, n);
            }
            pqo_mama(c, n, 0.5, 0.05, a, b); fold(a, n); fold(b, n);
            pqo_sar(h, l, n, 0.02, 0.2, a); fold(a, n);
            pqo_sarext(h, l, n, -3.0, 0.01, 0.02, 0.02, 0.2, 0.03, 0.03, 0.3, a); fold(a, n);
            pqo_bop(o, h, l, c, n, a); fold(a, n); pqo_trange(h, l, c, n, a); fold(a, n);
            pqo_ad(h, l, c, v, n, a); fold(a, n); pqo_obv(c, v, n, a); fold(a, n);
            pqo_avgprice(o, h, l, c, n, a); fold(a, n); pqo_medprice(h, l, n, a); fold(a, n);
            pqo_typprice(h, l, c, n, a); fold(a, n); pqo_wclprice(h, l, c, n, a); fold(a, n);
            pqo_ht_dcperiod(c, n, a); fold(a, n); pqo_ht_dcphase(c, n, a); fold(a, n);
            pqo_ht_phasor(c, n, a, b); fold(a, n); pqo_ht_sine(c, n, a, b); fold(b, n);
            pqo_ht_trendline(c, n, a); fold(a, n); pqo_ht_trendmode(c, n, ai); foldi(ai, n);
            if (!withnull)
                for (int id = 0; id < PQO_N_PATTERNS; id++) { pqo_pattern(id, o, h, l, c, n, 0.3, ai); foldi(ai, n); }
            /* backtests + summary + signals + factor */
            pqo_macd_cross_signals(c, n, 12, 26, 9, u1, u2);
            pqo_bt_params bp = {100000.0, 0.01, 0.01, 0.0003, 0.0003, 5.0, 0.7};
            double summ[8];
            pqo_backtest(c, u1, u2, withnull ? NULL : o, n, &bp, a, b, d, summ); fold(d, n); fold(summ, 8);
            pqo_summary(d, o, n, n, 100000.0, 3, 1, summ); fold(summ, 8);
            pqo_lev_params lp = {100000.0, 1.0, 2.0, 0.5, 0.08, 0.0005, 5.0, 0.001};
            for (int32_t mtr = 0; mtr < 5; mtr += 4) {
                int32_t cnt = 0, ed[4], xd[4], rs[4]; double ep[4], xp[4], q[4], pn[4], pp[4];
                pqo_backtest_leveraged(c, u1, u2, withnull ? NULL : o, n, &lp, a, b, d, mtr, &cnt, mtr ? ed : NULL, mtr ? xd : NULL, mtr ? ep : NULL,
                                       mtr ? xp : NULL, mtr ? q : NULL, mtr ? pn : NULL, mtr ? pp : NULL, mtr ? rs : NULL, summ);
                fold(d, n); fold(summ, 8);
            }
            pqo_cross_signals(c, o, n, u1, u2); pqo_band_signals(c, n, 30.0, 70.0, u1, u2);
            pqo_channel_signals(c, l, h, n, 0, u1, u2); pqo_channel_signals(c, l, h, n, 1, u1, u2);
            free(o); free(h); free(l); free(c); free(v); free(per); free(a); free(b); free(d); free(ai); free(u1); free(u2);
        }
    }
    /* cross-sectional functions: [n_sym][n] blocks incl. ragged / tiny shapes */
    const int64_t shapes[][2] = {{1, 1}, {2, 3}, {3, 1}, {7, 40}, {300, 9}, {257, 33}};
    for (size_t si = 0; si < sizeof shapes / sizeof shapes[0]; si++) {
        const int64_t ns = shapes[si][0], n = shapes[si][1];
        double *f = dbl(ns * n), *r = dbl(ns * n), *o = dbl(ns * n), *h = dbl(ns * n), *v = dbl(ns * n);
        pqo_gen_ohlcv(77 + si, ns, n, 0, o, h, f, r, v);
        uint64_t nb = PQO_NULL_BITS; double nul; memcpy(&nul, &nb, 8);
        for (int64_t i = 0; i < ns * n; i += 11) f[i] = nul;
        for (int64_t i = 3; i < ns * n; i += 17) r[i] = NAN;
        double *ic = dbl(n), *out = dbl(n * 10);
        int32_t *nv = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
        for (int m = 0; m < 2; m++) { pqo_factor_ic(f, r, ns, n, n, m, ic, nv); fold(ic, n); foldi(nv, n); pqo_factor_ic(f, r, ns, n, n, m, ic, NULL); }
        pqo_portfolio_metrics(o, ns, n, n, 1e5 * (double)ns, h, out); fold(out, n * 10);
        pqo_portfolio_metrics(o, ns, n, n, 1e5 * (double)ns, NULL, out); fold(out, n * 10);
        free(f); free(r); free(o); free(h); free(v); free(ic); free(out); free(nv);
    }
    printf("asan driver OK, checksum %.6f\n", acc);
    return 0;
}
