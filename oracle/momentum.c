/* momentum.c -- CPU ORACLE (test infrastructure) for src/talib/momentum.rs and the pure-Python
 * composites of python/polars_quant/talib/momentum.py.  All momentum.rs functions are N-B
 * (rechunk().cont_slice()? -> error on nulls, momentum.rs:12-13): inputs here are null-free;
 * the host layer rejects nulls before calling.  Compile with -ffp-contract=off. */
#include <math.h>
#include "pqo_common.h"

static double *dalloc(int64_t n) { return (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1)); }
static double z(double x) { return pqo_isnull(x) ? 0.0 : x; } /* .unwrap_or(0.0) */

/* ------------------------------------------------------------------------------------------
 * D-1: slice helpers that momentum.rs calls (momentum.rs:2, :21, :155, :257) but the crate
 * never defines.  calc_sma/calc_ema: same arithmetic as overlap.rs:871-937 / :660-730 on a
 * null-free slice, None for i < p-1.  calc_rma: Wilder smoothing -- None for i < p-1,
 * seed = (x[0]+...+x[p-1]) / p at i = p-1 (left-to-right sum), then (prev*(p-1) + x) / p.
 * p == 0 or n < p -> all None.
 * ------------------------------------------------------------------------------------------ */
static void sma_slice(const double *x, int64_t n, int64_t p, double *out) { pqo_sma(x, n, p, out); }
static void ema_slice(const double *x, int64_t n, int64_t p, double *out) { pqo_ema(x, n, p, out); }

void pqo_rma(const double *x, int64_t n, int64_t p, double *out) {
    pqo_fill_null(out, n);
    if (p <= 0 || n < p) return;
    double sum = 0.0, r = 0.0, pm1 = (double)p - 1.0, pf = (double)p;
    for (int64_t i = 0; i < n; i++) {
        if (i < p - 1) sum += x[i];
        else if (i == p - 1) { sum += x[i]; r = sum / pf; out[i] = r; }
        else { r = (r * pm1 + x[i]) / pf; out[i] = r; }
    }
}

/* momentum.rs:668-727 calc_dm -> (dx, minus_di); plus_di kept for the D-5 note */
static void calc_dm(const double *h, const double *l, const double *c, int64_t n, int64_t p,
                    double *dx, double *minus_di, double *plus_di) {
    double *p_dm = (double *)calloc((size_t)(n > 0 ? n : 1), 8), *m_dm = (double *)calloc((size_t)(n > 0 ? n : 1), 8),
           *tr = (double *)calloc((size_t)(n > 0 ? n : 1), 8);
    for (int64_t i = 1; i < n; i++) {
        double up_move = h[i] - h[i - 1], down_move = l[i - 1] - l[i], pc = c[i - 1];
        if (up_move > down_move && up_move > 0.0) p_dm[i] = up_move;                      /* :691 */
        if (down_move > up_move && down_move > 0.0) m_dm[i] = down_move;                  /* :694 */
        tr[i] = RMAX(RMAX(h[i] - l[i], fabs(h[i] - pc)), fabs(l[i] - pc));                /* :697 */
    }
    double *sp = dalloc(n), *sm = dalloc(n), *st = dalloc(n);
    pqo_rma(p_dm, n, p, sp); pqo_rma(m_dm, n, p, sm); pqo_rma(tr, n, p, st);
    for (int64_t i = 0; i < n; i++) {
        double pdi = pqo_null(), mdi = pqo_null();
        if (!pqo_isnull(sp[i]) && !pqo_isnull(sm[i]) && !pqo_isnull(st[i]) && st[i] != 0.0) { /* :708-709 */
            pdi = 100.0 * sp[i] / st[i];
            mdi = 100.0 * sm[i] / st[i];
        }
        if (plus_di) plus_di[i] = pdi;
        if (minus_di) minus_di[i] = mdi;
        if (dx) {
            if (!pqo_isnull(pdi) && !pqo_isnull(mdi)) {
                double diff = fabs(pdi - mdi), sum = pdi + mdi;
                dx[i] = (sum == 0.0) ? 0.0 : 100.0 * diff / sum;                          /* :721-723 */
            } else dx[i] = pqo_null();
        }
    }
    free(p_dm); free(m_dm); free(tr); free(sp); free(sm); free(st);
}

/* momentum.rs:11-29 */
void pqo_adx(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out) {
    double *dx = dalloc(n);
    calc_dm(h, l, c, n, p, dx, NULL, NULL);
    for (int64_t i = 0; i < n; i++) dx[i] = z(dx[i]);                                     /* :22-25 */
    pqo_rma(dx, n, p, out);
    free(dx);
}
/* momentum.rs:32-61 */
void pqo_adxr(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out) {
    double *adx = dalloc(n);
    pqo_adx(h, l, c, n, p, adx);
    pqo_fill_null(out, n);
    if (p <= 0) { free(adx); return; }
    for (int64_t i = p - 1; i < n; i++) {
        double curr = adx[i], prev = adx[i - (p - 1)];                                    /* :55 */
        if (!pqo_isnull(curr) && !pqo_isnull(prev)) out[i] = (curr + prev) * 0.5;
    }
    free(adx);
}
/* momentum.rs:226-237 */
void pqo_dx(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out) {
    calc_dm(h, l, c, n, p, out, NULL, NULL);
}
/* momentum.rs:400-411 -- quirk Q-PDI / decision D-5: takes calc_dm().0, i.e. returns DX. */
void pqo_plus_di(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out) {
    calc_dm(h, l, c, n, p, out, NULL, NULL);
}
/* momentum.rs:345-356 */
void pqo_minus_di(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out) {
    calc_dm(h, l, c, n, p, NULL, out, NULL);
}
/* momentum.rs:414-436 */
void pqo_plus_dm(const double *h, const double *l, int64_t n, int64_t p, double *out) {
    double *d = (double *)calloc((size_t)(n > 0 ? n : 1), 8);
    for (int64_t i = 1; i < n; i++) {
        double up_move = h[i] - h[i - 1], down_move = l[i - 1] - l[i];
        if (up_move > down_move && up_move > 0.0) d[i] = up_move;
    }
    pqo_rma(d, n, p, out);
    free(d);
}
/* momentum.rs:359-381 */
void pqo_minus_dm(const double *h, const double *l, int64_t n, int64_t p, double *out) {
    double *d = (double *)calloc((size_t)(n > 0 ? n : 1), 8);
    for (int64_t i = 1; i < n; i++) {
        double up_move = h[i] - h[i - 1], down_move = l[i - 1] - l[i];
        if (down_move > up_move && down_move > 0.0) d[i] = down_move;
    }
    pqo_rma(d, n, p, out);
    free(d);
}
/* momentum.rs:70-110 */
void pqo_aroon(const double *h, const double *l, int64_t n, int64_t p, double *up, double *down) {
    pqo_fill_null(up, n); pqo_fill_null(down, n);
    if (p < 0) return;
    for (int64_t i = p; i < n; i++) {
        int64_t start = i - p, max_idx = 0, min_idx = 0;
        double max_val = -1.7976931348623157e308, min_val = 1.7976931348623157e308;      /* f64::MIN / MAX */
        for (int64_t j = start; j <= i; j++) {
            if (h[j] >= max_val) { max_val = h[j]; max_idx = j - start; }                 /* :90 */
            if (l[j] <= min_val) { min_val = l[j]; min_idx = j - start; }                 /* :96 */
        }
        up[i] = ((double)max_idx / (double)p) * 100.0;                                    /* :103 */
        down[i] = ((double)min_idx / (double)p) * 100.0;
    }
}
/* D-6: AROONOSC has a Python name (momentum.py:40-45) but no Rust; defined TA-Lib style on the
 * reference's own AROON: aroon_up - aroon_down. */
void pqo_aroonosc(const double *h, const double *l, int64_t n, int64_t p, double *out) {
    double *u = dalloc(n), *d = dalloc(n);
    pqo_aroon(h, l, n, p, u, d);
    for (int64_t i = 0; i < n; i++) out[i] = (pqo_isnull(u[i]) || pqo_isnull(d[i])) ? pqo_null() : u[i] - d[i];
    free(u); free(d);
}
/* momentum.rs:113-135 */
void pqo_bop(const double *o, const double *h, const double *l, const double *c, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) {
        double diff = h[i] - l[i];
        out[i] = (diff == 0.0) ? 0.0 : (c[i] - o[i]) / diff;
    }
}
/* momentum.rs:138-178 */
void pqo_cci(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out) {
    pqo_fill_null(out, n);
    if (p <= 0) return;
    double *tp = dalloc(n), *sma_tp = dalloc(n);
    for (int64_t i = 0; i < n; i++) tp[i] = (h[i] + l[i] + c[i]) / 3.0;                    /* :151 */
    sma_slice(tp, n, p, sma_tp);
    for (int64_t i = p - 1; i < n; i++) {
        if (pqo_isnull(sma_tp[i])) continue;
        double avg = sma_tp[i], mean_dev = 0.0;
        for (int64_t j = i + 1 - p; j <= i; j++) mean_dev += fabs(tp[j] - avg);           /* :165-170 */
        if (mean_dev != 0.0) {
            mean_dev /= (double)p;
            out[i] = (tp[i] - avg) / (0.015 * mean_dev);                                  /* :173 */
        }
    }
    free(tp); free(sma_tp);
}
/* momentum.rs:181-223 */
void pqo_cmo(const double *v, int64_t n, int64_t p, double *out) {
    pqo_fill_null(out, n);
    if (p <= 0) return;
    double *ups = (double *)calloc((size_t)(n > 0 ? n : 1), 8), *downs = (double *)calloc((size_t)(n > 0 ? n : 1), 8);
    for (int64_t i = 1; i < n; i++) {
        double diff = v[i] - v[i - 1];
        if (diff > 0.0) ups[i] = diff; else downs[i] = -diff;                             /* :193-197 */
    }
    double su = 0.0, sd = 0.0;
    for (int64_t i = 0; i < n; i++) {
        su += ups[i]; sd += downs[i];
        if (i >= p) { su -= ups[i - p]; sd -= downs[i - p]; }
        if (i >= p - 1) {
            double total = su + sd;
            out[i] = (total == 0.0) ? 0.0 : 100.0 * (su - sd) / total;                    /* :215-219 */
        }
    }
    free(ups); free(downs);
}
/* momentum.rs:250-283 (quirk Q-MACD: signal = EMA of dif with None -> 0.0) */
void pqo_macd(const double *v, int64_t n, int64_t fast, int64_t slow, int64_t sig,
              double *macd, double *signal, double *hist) {
    double *fe = dalloc(n), *se = dalloc(n), *dz = dalloc(n);
    ema_slice(v, n, fast, fe); ema_slice(v, n, slow, se);
    for (int64_t i = 0; i < n; i++) {
        macd[i] = (!pqo_isnull(fe[i]) && !pqo_isnull(se[i])) ? fe[i] - se[i] : pqo_null(); /* :263-265 */
        dz[i] = z(macd[i]);
    }
    ema_slice(dz, n, sig, signal);                                                        /* :268-271 */
    for (int64_t i = 0; i < n; i++)
        hist[i] = (!pqo_isnull(macd[i]) && !pqo_isnull(signal[i])) ? macd[i] - signal[i] : pqo_null();
    free(fe); free(se); free(dz);
}
/* momentum.rs:286-342 */
void pqo_mfi(const double *h, const double *l, const double *c, const double *vol, int64_t n,
             int64_t p, double *out) {
    pqo_fill_null(out, n);
    double *tp = dalloc(n), *mf = dalloc(n);
    for (int64_t i = 0; i < n; i++) { tp[i] = (h[i] + l[i] + c[i]) / 3.0; mf[i] = tp[i] * vol[i]; }
    double pos = 0.0, neg = 0.0;
    for (int64_t i = 1; i < n; i++) {
        if (tp[i] > tp[i - 1]) pos += mf[i];
        else if (tp[i] < tp[i - 1]) neg += mf[i];
        if (i >= p) {
            int64_t prev_idx = i - p;
            if (prev_idx > 0) {                                                           /* :322 */
                if (tp[prev_idx] > tp[prev_idx - 1]) pos -= mf[prev_idx];
                else if (tp[prev_idx] < tp[prev_idx - 1]) neg -= mf[prev_idx];
            }
            if (neg == 0.0) out[i] = 100.0;
            else { double mr = pos / neg; out[i] = 100.0 - (100.0 / (1.0 + mr)); }         /* :335-336 */
        }
    }
    free(tp); free(mf);
}
/* momentum.rs:384-397, :439-504 */
void pqo_mom(const double *v, int64_t n, int64_t p, double *out) {
    pqo_fill_null(out, n);
    for (int64_t i = (p < 0 ? 0 : p); i < n; i++) out[i] = v[i] - v[i - p];
}
void pqo_roc(const double *v, int64_t n, int64_t p, double *out) {
    pqo_fill_null(out, n);
    for (int64_t i = (p < 0 ? 0 : p); i < n; i++) { double pr = v[i - p]; if (pr != 0.0) out[i] = (v[i] - pr) / pr * 100.0; }
}
void pqo_rocp(const double *v, int64_t n, int64_t p, double *out) {
    pqo_fill_null(out, n);
    for (int64_t i = (p < 0 ? 0 : p); i < n; i++) { double pr = v[i - p]; if (pr != 0.0) out[i] = (v[i] - pr) / pr; }
}
void pqo_rocr(const double *v, int64_t n, int64_t p, double *out) {
    pqo_fill_null(out, n);
    for (int64_t i = (p < 0 ? 0 : p); i < n; i++) { double pr = v[i - p]; if (pr != 0.0) out[i] = v[i] / pr; }
}
void pqo_rocr100(const double *v, int64_t n, int64_t p, double *out) {
    pqo_fill_null(out, n);
    for (int64_t i = (p < 0 ? 0 : p); i < n; i++) { double pr = v[i - p]; if (pr != 0.0) out[i] = (v[i] / pr) * 100.0; }
}
/* README.md:46-75 `returns(df, price_col, period, method)` (README-only, no source; decision D-13): method 0 "simple" =
 * (p[t] - p[t-period]) / p[t-period], method 1 "log" = ln(p[t] / p[t-period]); null for t < period and where either price
 * is null (a Polars shift/arithmetic expression: nulls propagate, a zero denominator follows IEEE-754).  period <= 0 or an
 * unknown method -> all null.  Pinned by the reference-held vector README.md:75. */
void pqo_returns(const double *v, int64_t n, int64_t period, int64_t method, double *out) {
    pqo_fill_null(out, n);
    if (period <= 0 || (method != 0 && method != 1)) return;
    for (int64_t i = period; i < n; i++) {
        double c = v[i], pr = v[i - period];
        if (pqo_isnull(c) || pqo_isnull(pr)) continue;
        out[i] = method == 0 ? (c - pr) / pr : log(c / pr);
    }
}
/* Polars rolling_max / rolling_min(window) (python/polars_quant/talib/momentum.py:181-183; py-polars 1.39.3, uv.lock:213-214 -- a
 * dependency whose source is not in the tree: decision D-14): null until the frame of the last `window` rows holds `window` non-null
 * rows; a NaN VALUE is ignored like in Polars' max() / min() (the rolling kernels' NaN-ignoring policy), the result is NaN only if
 * every value of the frame is NaN. */
static double frame_ext(const double *v, int64_t lo, int64_t hi, int is_max, int *ok) { /* rows lo .. hi */
    double best = NAN;
    *ok = 1;
    for (int64_t j = lo; j <= hi; j++) {
        if (pqo_isnull(v[j])) { *ok = 0; return best; }
        if (v[j] != v[j]) continue;
        if (best != best || (is_max ? v[j] > best : v[j] < best)) best = v[j];
    }
    return best;
}
static void roll_ext(const double *v, int64_t n, int64_t w, int is_max, double *out) {
    pqo_fill_null(out, n);
    if (w <= 0) return;
    for (int64_t i = w - 1; i < n; i++) {
        int ok;
        const double best = frame_ext(v, i + 1 - w, i, is_max, &ok);
        if (ok) out[i] = best;
    }
}
void pqo_rolling_max(const double *v, int64_t n, int64_t w, double *out) { roll_ext(v, n, w, 1, out); }
void pqo_rolling_min(const double *v, int64_t n, int64_t w, double *out) { roll_ext(v, n, w, 0, out); }
/* momentum.rs:507-541 */
void pqo_rsi(const double *v, int64_t n, int64_t p, double *out) {
    double *ups = (double *)calloc((size_t)(n > 0 ? n : 1), 8), *downs = (double *)calloc((size_t)(n > 0 ? n : 1), 8);
    for (int64_t i = 1; i < n; i++) {
        double diff = v[i] - v[i - 1];
        if (diff > 0.0) ups[i] = diff; else downs[i] = -diff;
    }
    double *au = dalloc(n), *ad = dalloc(n);
    pqo_rma(ups, n, p, au); pqo_rma(downs, n, p, ad);
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(au[i]) || pqo_isnull(ad[i])) { out[i] = pqo_null(); continue; }
        if (ad[i] == 0.0) out[i] = 100.0;
        else { double rs = au[i] / ad[i]; out[i] = 100.0 - (100.0 / (1.0 + rs)); }         /* :535-536 */
    }
    free(ups); free(downs); free(au); free(ad);
}
/* momentum.rs:544-569 (quirk Q-TRIX: None -> 0.0 between EMA stages) */
void pqo_trix(const double *v, int64_t n, int64_t p, double *out) {
    double *e1 = dalloc(n), *e2 = dalloc(n), *e3 = dalloc(n);
    ema_slice(v, n, p, e1);
    for (int64_t i = 0; i < n; i++) e1[i] = z(e1[i]);
    ema_slice(e1, n, p, e2);
    for (int64_t i = 0; i < n; i++) e2[i] = z(e2[i]);
    ema_slice(e2, n, p, e3);
    pqo_fill_null(out, n);
    for (int64_t i = 1; i < n; i++) {
        if (pqo_isnull(e3[i]) || pqo_isnull(e3[i - 1])) continue;
        if (e3[i - 1] != 0.0) out[i] = (e3[i] - e3[i - 1]) / e3[i - 1] * 100.0;            /* :564 */
    }
    free(e1); free(e2); free(e3);
}
/* momentum.rs:572-627 */
static void ult_avg(const double *bp, const double *tr, int64_t n, int64_t p, double *res) {
    pqo_fill_null(res, n);
    double s_bp = 0.0, s_tr = 0.0;
    for (int64_t i = 0; i < n; i++) {
        s_bp += bp[i]; s_tr += tr[i];
        if (i >= p) { s_bp -= bp[i - p]; s_tr -= tr[i - p]; }
        if (i >= p - 1 && s_tr != 0.0) res[i] = s_bp / s_tr;                              /* :609-611 */
    }
}
void pqo_ultosc(const double *h, const double *l, const double *c, int64_t n, int64_t p1,
                int64_t p2, int64_t p3, double *out) {
    pqo_fill_null(out, n);
    if (p1 <= 0 || p2 <= 0 || p3 <= 0) return;
    double *bp = (double *)calloc((size_t)(n > 0 ? n : 1), 8), *tr = (double *)calloc((size_t)(n > 0 ? n : 1), 8);
    for (int64_t i = 1; i < n; i++) {
        double pc = c[i - 1];
        double min_l_pc = RMIN(l[i], pc), max_h_pc = RMAX(h[i], pc);
        bp[i] = c[i] - min_l_pc; tr[i] = max_h_pc - min_l_pc;                             /* :593-594 */
    }
    double *a1 = dalloc(n), *a2 = dalloc(n), *a3 = dalloc(n);
    ult_avg(bp, tr, n, p1, a1); ult_avg(bp, tr, n, p2, a2); ult_avg(bp, tr, n, p3, a3);
    for (int64_t i = 0; i < n; i++)
        if (!pqo_isnull(a1[i]) && !pqo_isnull(a2[i]) && !pqo_isnull(a3[i]))
            out[i] = 100.0 * (4.0 * a1[i] + 2.0 * a2[i] + a3[i]) / 7.0;                   /* :623 */
    free(bp); free(tr); free(a1); free(a2); free(a3);
}
/* momentum.rs:630-662 */
void pqo_willr(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out) {
    pqo_fill_null(out, n);
    if (p <= 0) return;
    for (int64_t i = p - 1; i < n; i++) {
        double max_h = -1.7976931348623157e308, min_l = 1.7976931348623157e308;
        for (int64_t j = i + 1 - p; j <= i; j++) { max_h = RMAX(max_h, h[j]); min_l = RMIN(min_l, l[j]); }
        double diff = max_h - min_l;
        out[i] = (diff == 0.0) ? 0.0 : -100.0 * (max_h - c[i]) / diff;                    /* :654-658 */
    }
}

/* D-6: APO / PPO have Python names (momentum.py:25-30, :136-141) but no Rust.  Defined TA-Lib
 * style on the reference's own calc_ma: APO = MA(fast) - MA(slow); PPO = (MA(fast)-MA(slow))/MA(slow)*100
 * (null where either MA is null; PPO null where MA(slow) == 0). */
void pqo_apo(const double *v, int64_t n, int64_t fast, int64_t slow, int64_t matype, double *out) {
    double *f = dalloc(n), *s = dalloc(n);
    pqo_ma(v, n, fast, matype, f); pqo_ma(v, n, slow, matype, s);
    for (int64_t i = 0; i < n; i++) out[i] = (pqo_isnull(f[i]) || pqo_isnull(s[i])) ? pqo_null() : f[i] - s[i];
    free(f); free(s);
}
void pqo_ppo(const double *v, int64_t n, int64_t fast, int64_t slow, int64_t matype, double *out) {
    double *f = dalloc(n), *s = dalloc(n);
    pqo_ma(v, n, fast, matype, f); pqo_ma(v, n, slow, matype, s);
    for (int64_t i = 0; i < n; i++)
        out[i] = (pqo_isnull(f[i]) || pqo_isnull(s[i]) || s[i] == 0.0) ? pqo_null() : (f[i] - s[i]) / s[i] * 100.0;
    free(f); free(s);
}

/* ---------------- python composites (momentum.py:83-92, :178-205) ---------------- */
/* momentum.py:83-88: MA - MA (null if either null); signal = MA(macd_line) with N-A skipping */
void pqo_macdext(const double *v, int64_t n, int64_t fast, int64_t fastmt, int64_t slow,
                 int64_t slowmt, int64_t sig, int64_t sigmt, double *macd, double *signal, double *hist) {
    double *f = dalloc(n), *s = dalloc(n);
    pqo_ma(v, n, fast, fastmt, f); pqo_ma(v, n, slow, slowmt, s);
    for (int64_t i = 0; i < n; i++) macd[i] = (pqo_isnull(f[i]) || pqo_isnull(s[i])) ? pqo_null() : f[i] - s[i];
    pqo_ma(macd, n, sig, sigmt, signal);
    for (int64_t i = 0; i < n; i++)
        hist[i] = (pqo_isnull(macd[i]) || pqo_isnull(signal[i])) ? pqo_null() : macd[i] - signal[i];
    free(f); free(s);
}
/* momentum.py:90-92 */
void pqo_macdfix(const double *v, int64_t n, int64_t sig, double *macd, double *signal, double *hist) {
    pqo_macd(v, n, 12, 26, sig, macd, signal, hist);
}
/* Polars rolling_min/rolling_max(window) (py-polars 1.39.3): null until `window` non-null values
 * are inside the `window`-row frame (min_samples = window), i.e. null if any row in the frame is null. */
static void rolling_ext(const double *x, int64_t n, int64_t w, int is_max, double *out) { roll_ext(x, n, w, is_max, out); }
static void fastk_of(const double *h, const double *l, const double *c, int64_t n, int64_t k, double *fk) {
    double *ln = dalloc(n), *hn = dalloc(n);
    rolling_ext(l, n, k, 0, ln); rolling_ext(h, n, k, 1, hn);
    for (int64_t i = 0; i < n; i++)
        fk[i] = (pqo_isnull(c[i]) || pqo_isnull(ln[i]) || pqo_isnull(hn[i])) ? pqo_null()
                : (c[i] - ln[i]) * 100.0 / (hn[i] - ln[i]);                               /* momentum.py:183 */
    free(ln); free(hn);
}
/* momentum.py:178-186 */
void pqo_stoch(const double *h, const double *l, const double *c, int64_t n, int64_t fastk,
               int64_t slowk, int64_t slowk_mt, int64_t slowd, int64_t slowd_mt, double *outk, double *outd) {
    double *fk = dalloc(n);
    fastk_of(h, l, c, n, fastk, fk);
    pqo_ma(fk, n, slowk, slowk_mt, outk);
    pqo_ma(outk, n, slowd, slowd_mt, outd);
    free(fk);
}
/* momentum.py:188-195 */
void pqo_stochf(const double *h, const double *l, const double *c, int64_t n, int64_t fastk,
                int64_t fastd, int64_t fastd_mt, double *outk, double *outd) {
    fastk_of(h, l, c, n, fastk, outk);
    pqo_ma(outk, n, fastd, fastd_mt, outd);
}
/* momentum.py:197-205 */
void pqo_stochrsi(const double *v, int64_t n, int64_t p, int64_t fastk, int64_t fastd,
                  int64_t fastd_mt, double *outk, double *outd) {
    double *rsi = dalloc(n);
    pqo_rsi(v, n, p, rsi);
    fastk_of(rsi, rsi, rsi, n, fastk, outk);
    pqo_ma(outk, n, fastd, fastd_mt, outd);
    free(rsi);
}
