/* batch.c -- CPU ORACLE (test infrastructure): runs the whole indicator suite + MACD-cross
 * backtest over a symbol-major [N][T] OHLCV block, one symbol at a time (OpenMP over symbols
 * mirrors Polars' rayon-over-groups).  Used by bench.py's cpu_baseline leg only. */
#include "pqo_common.h"
#include <omp.h>

/* Returns a checksum so the work cannot be optimised away.  threads <= 0 -> omp default. */
double pqo_suite_bench(const double *o, const double *h, const double *l, const double *c,
                       const double *v, int64_t n_sym, int64_t T, int threads) {
    double total = 0.0;
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int64_t s = 0; s < n_sym; s++) {
        const double *O = o + s * T, *H = h + s * T, *L = l + s * T, *C = c + s * T, *V = v + s * T;
        size_t m = (size_t)T;
        double *a = (double *)malloc(8 * m), *b = (double *)malloc(8 * m), *d = (double *)malloc(8 * m);
        double *per = (double *)malloc(8 * m);
        int32_t *ip = (int32_t *)malloc(4 * m);
        uint8_t *bu = (uint8_t *)malloc(m), *se = (uint8_t *)malloc(m);
        double acc = 0.0;
#define ACC(x) acc += (pqo_isnull((x)[T - 1]) ? 0.0 : (x)[T - 1])
        /* overlap -- python wrapper defaults (overlap.py) */
        pqo_bbands(C, T, 20, 2.0, 2.0, a, b, d); ACC(a); ACC(b); ACC(d);
        pqo_dema(C, T, 30, a); ACC(a);   pqo_ema(C, T, 30, a); ACC(a);   pqo_kama(C, T, 30, a); ACC(a);
        pqo_ma(C, T, 30, 0, a); ACC(a);  pqo_mama(C, T, 0.0, 0.0, a, b); ACC(a); ACC(b);
        for (int64_t i = 0; i < T; i++) per[i] = (double)(2 + (i % 29));
        pqo_mavp(C, per, T, 2, 30, 0, a); ACC(a);
        pqo_midpoint(C, T, 14, a); ACC(a); pqo_midprice(H, L, T, 14, a); ACC(a);
        pqo_sar(H, L, T, 0.0, 0.0, a); ACC(a);
        pqo_sarext(H, L, T, 0, 0, 0, 0, 0, 0, 0, 0, a); ACC(a);
        pqo_sma(C, T, 30, a); ACC(a);    pqo_t3(C, T, 5, 0.7, a); ACC(a); pqo_tema(C, T, 30, a); ACC(a);
        pqo_trima(C, T, 30, a); ACC(a);  pqo_wma(C, T, 30, a); ACC(a);
        /* momentum */
        pqo_adx(H, L, C, T, 14, a); ACC(a);  pqo_adxr(H, L, C, T, 14, a); ACC(a);
        pqo_apo(C, T, 12, 26, 0, a); ACC(a); pqo_aroon(H, L, T, 14, a, b); ACC(a); ACC(b);
        pqo_aroonosc(H, L, T, 14, a); ACC(a); pqo_bop(O, H, L, C, T, a); ACC(a);
        pqo_cci(H, L, C, T, 14, a); ACC(a);  pqo_cmo(C, T, 14, a); ACC(a); pqo_dx(H, L, C, T, 14, a); ACC(a);
        pqo_macd(C, T, 12, 26, 9, a, b, d); ACC(a); ACC(b); ACC(d);
        pqo_macdext(C, T, 12, 0, 26, 0, 9, 0, a, b, d); ACC(a); ACC(b); ACC(d);
        pqo_macdfix(C, T, 9, a, b, d); ACC(a); ACC(b); ACC(d);
        pqo_mfi(H, L, C, V, T, 14, a); ACC(a);
        pqo_minus_di(H, L, C, T, 14, a); ACC(a); pqo_minus_dm(H, L, T, 14, a); ACC(a);
        pqo_mom(C, T, 10, a); ACC(a); pqo_plus_di(H, L, C, T, 14, a); ACC(a); pqo_plus_dm(H, L, T, 14, a); ACC(a);
        pqo_ppo(C, T, 12, 26, 0, a); ACC(a);
        pqo_roc(C, T, 10, a); ACC(a); pqo_rocp(C, T, 10, a); ACC(a); pqo_rocr(C, T, 10, a); ACC(a);
        pqo_rocr100(C, T, 10, a); ACC(a); pqo_rsi(C, T, 14, a); ACC(a);
        pqo_stoch(H, L, C, T, 5, 3, 0, 3, 0, a, b); ACC(a); ACC(b);
        pqo_stochf(H, L, C, T, 5, 3, 0, a, b); ACC(a); ACC(b);
        pqo_stochrsi(C, T, 14, 5, 3, 0, a, b); ACC(a); ACC(b);
        pqo_trix(C, T, 30, a); ACC(a); pqo_ultosc(H, L, C, T, 7, 14, 28, a); ACC(a);
        pqo_willr(H, L, C, T, 14, a); ACC(a);
        /* volatility / volume / price */
        pqo_atr(H, L, C, T, 14, a); ACC(a); pqo_natr(H, L, C, T, 14, a); ACC(a); pqo_trange(H, L, C, T, a); ACC(a);
        pqo_ad(H, L, C, V, T, a); ACC(a); pqo_adosc(H, L, C, V, T, 3, 10, a); ACC(a); pqo_obv(C, V, T, a); ACC(a);
        pqo_avgprice(O, H, L, C, T, a); ACC(a); pqo_medprice(H, L, T, a); ACC(a);
        pqo_typprice(H, L, C, T, a); ACC(a); pqo_wclprice(H, L, C, T, a); ACC(a);
        /* cycle */
        pqo_ht_dcperiod(C, T, a); ACC(a); pqo_ht_dcphase(C, T, a); ACC(a);
        pqo_ht_phasor(C, T, a, b); ACC(a); ACC(b); pqo_ht_sine(C, T, a, b); ACC(a); ACC(b);
        pqo_ht_trendline(C, T, a); ACC(a); pqo_ht_trendmode(C, T, ip); acc += ip[T - 1] == PQO_NULL_I32 ? 0 : ip[T - 1];
        /* patterns: python defaults (0.3 except darkcloud 0.5, mathold 0.5, piercing 0.5) */
        for (int id = 0; id < PQO_N_PATTERNS; id++) {
            double pen = (id == 14 || id == 41 || id == 45) ? 0.5 : 0.3;
            pqo_pattern(id, O, H, L, C, T, pen, ip);
            acc += ip[T - 1];
        }
        /* MACD-cross backtest */
        pqo_bt_params prm = {100000.0, 0.0, 0.0, 0.0003, 0.0003, 5.0, 1.0};
        double sm[8];
        pqo_macd_cross_signals(C, T, 12, 26, 9, bu, se);
        pqo_backtest(C, bu, se, NULL, T, &prm, a, b, d, sm);
        acc += sm[0] + sm[7];
        total += acc;
        free(a); free(b); free(d); free(per); free(ip); free(bu); free(se);
    }
    return total;
}
