/* overlap.c -- CPU ORACLE (test infrastructure) for src/talib/overlap.rs.
 * Literal restatement: same operation order, fma() only where Rust calls mul_add,
 * null-transparent streaming (N-A) exactly as the PrimitiveChunkedBuilder loops do.
 * Compile with -ffp-contract=off. */
#include "pqo_common.h"

/* overlap.rs:871-937 calc_sma */
void pqo_sma(const double *v, int64_t n, int64_t p, double *out) {
    if (p <= 0 || n < p) { pqo_fill_null(out, n); return; }  /* :874-876 */
    double denominator = 1.0 / (double)p;                      /* :880 */
    int64_t count = 0;
    double sum = 0.0;
    pqo_deque w; dq_init(&w, n);
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(v[i])) { out[i] = pqo_null(); continue; } /* :892-895 */
        double value = v[i];
        count += 1; sum += value; dq_push_back(&w, value);     /* :897-899 */
        if (count < p) out[i] = pqo_null();
        else {
            if (count > p) { double old = dq_pop_front(&w); sum -= old; count -= 1; } /* :904-909 */
            out[i] = sum * denominator;                         /* :910 */
        }
    }
    dq_free(&w);
}

/* overlap.rs:660-730 calc_ema */
void pqo_ema(const double *v, int64_t n, int64_t p, double *out) {
    if (p <= 0 || n < p) { pqo_fill_null(out, n); return; }  /* :663-665 */
    double alpha = 2.0 / ((double)p + 1.0);                    /* :669 */
    int64_t count = 0;
    double ema = 0.0, sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(v[i])) { out[i] = pqo_null(); continue; }
        double value = v[i];
        count += 1;
        if (count < p) { sum += value; out[i] = pqo_null(); }
        else if (count == p) { sum += value; ema = sum / (double)p; out[i] = ema; }   /* :692-696 */
        else { ema = fma(alpha, value - ema, ema); out[i] = ema; }                   /* :698 */
    }
}

/* overlap.rs:47-116 bbands */
void pqo_bbands(const double *v, int64_t n, int64_t p, double up, double dn,
                double *upper, double *middle, double *lower) {
    if (p <= 0 || n < p) { pqo_fill_null(upper, n); pqo_fill_null(middle, n); pqo_fill_null(lower, n); return; }
    int64_t count = 0;
    double sum = 0.0, sum_sq = 0.0;
    pqo_deque w; dq_init(&w, n);
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(v[i])) { upper[i] = middle[i] = lower[i] = pqo_null(); continue; }
        double value = v[i];
        count += 1; sum += value; sum_sq += value * value; dq_push_back(&w, value); /* :84-87 */
        if (count < p) { upper[i] = middle[i] = lower[i] = pqo_null(); }
        else {
            if (count > p) {
                double old = dq_pop_front(&w);
                sum -= old; sum_sq -= old * old; count -= 1;    /* :95-99 */
            }
            double mean = sum / (double)p;
            double variance = (sum_sq / (double)p) - mean * mean; /* :102 */
            double sd = sqrt(RMAX(variance, 0.0));               /* :103 */
            upper[i] = mean + up * sd;
            middle[i] = mean;
            lower[i] = mean - dn * sd;
        }
    }
    dq_free(&w);
}

/* overlap.rs:543-598 calc_dema, bitmap branch (decision D-2: the no-bitmap branch :600-653
 * is a pasted TEMA that indexes [2] on 2-element ArrayVecs and aborts; the bitmap branch is
 * the only executable definition and is used for every input). */
void pqo_dema(const double *v, int64_t n, int64_t p, double *out) {
    if (p <= 0 || n < 2 * p - 1) { pqo_fill_null(out, n); return; } /* :546-548 */
    double alpha = 2.0 / ((double)p + 1.0);
    int64_t count = 0;
    double e[2] = {0.0, 0.0}, s[2] = {0.0, 0.0};
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(v[i])) { out[i] = pqo_null(); continue; }
        double value = v[i];
        count += 1;
        if (count < p) { s[0] += value; out[i] = pqo_null(); }
        else if (count == p) { s[0] += value; e[0] = s[0] / (double)p; s[1] = e[0]; out[i] = pqo_null(); }
        else if (count < 2 * p - 1) { e[0] = fma(alpha, value - e[0], e[0]); s[1] += e[0]; out[i] = pqo_null(); }
        else if (count == 2 * p - 1) {
            e[0] = fma(alpha, value - e[0], e[0]); s[1] += e[0]; e[1] = s[1] / (double)p; out[i] = pqo_null();
        } else {
            e[0] = fma(alpha, value - e[0], e[0]);
            e[1] = fma(alpha, e[0] - e[1], e[1]);
            out[i] = 2.0 * e[0] - e[1];                          /* :595 */
        }
    }
}

/* overlap.rs:1177-1311 calc_tema */
void pqo_tema(const double *v, int64_t n, int64_t p, double *out) {
    if (p <= 0 || n < 3 * p - 2) { pqo_fill_null(out, n); return; }
    double alpha = 2.0 / ((double)p + 1.0);
    int64_t count = 0;
    double e[3] = {0, 0, 0}, s[3] = {0, 0, 0};
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(v[i])) { out[i] = pqo_null(); continue; }
        double value = v[i];
        count += 1;
        if (count < p) { s[0] += value; out[i] = pqo_null(); }
        else if (count == p) { s[0] += value; e[0] = s[0] / (double)p; s[1] = e[0]; out[i] = pqo_null(); }
        else if (count < 2 * p - 1) { e[0] = fma(alpha, value - e[0], e[0]); s[1] += e[0]; out[i] = pqo_null(); }
        else if (count == 2 * p - 1) {
            e[0] = fma(alpha, value - e[0], e[0]); s[1] += e[0]; e[1] = s[1] / (double)p; s[2] = e[1];
            out[i] = pqo_null();
        } else if (count < 3 * p - 2) {
            e[0] = fma(alpha, value - e[0], e[0]); e[1] = fma(alpha, e[0] - e[1], e[1]); s[2] += e[1];
            out[i] = pqo_null();
        } else if (count == 3 * p - 2) {
            e[0] = fma(alpha, value - e[0], e[0]); e[1] = fma(alpha, e[0] - e[1], e[1]); s[2] += e[1];
            e[2] = s[2] / (double)p;
            out[i] = 3.0 * e[0] - 3.0 * e[1] + e[2];             /* :1238-1240 */
        } else {
            e[0] = fma(alpha, value - e[0], e[0]);
            e[1] = fma(alpha, e[0] - e[1], e[1]);
            e[2] = fma(alpha, e[1] - e[2], e[2]);
            out[i] = 3.0 * e[0] - 3.0 * e[1] + e[2];
        }
    }
}

/* overlap.rs:939-1175 calc_t3.  Decision D-3: the output formula of the no-bitmap branch
 * (:1160-1166, true T3 coefficients via nested mul_add) is used for every input; the quirk that
 * e5 is never seeded (stays 0.0 until its first update at count == 6p-5) is kept. */
void pqo_t3(const double *v, int64_t n, int64_t p, double vf, double *out) {
    if (p <= 0 || n < 6 * p - 5) { pqo_fill_null(out, n); return; }
    double alpha = 2.0 / ((double)p + 1.0);
    double c1 = -(vf * vf * vf);                                 /* :949 -vfactor.powi(3) */
    double c2 = 3.0 * (vf * vf) - 3.0 * c1;                      /* :950 */
    double c3 = -2.0 * c2 - 3.0 * c1 - 3.0 * vf;                 /* :951 */
    double c4 = 1.0 - c1 - c2 - c3;                              /* :952 */
    int64_t count = 0;
    double e[6] = {0, 0, 0, 0, 0, 0}, s[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(v[i])) { out[i] = pqo_null(); continue; }
        double value = v[i];
        count += 1;
        out[i] = pqo_null();
        if (count < p) { s[0] += value; continue; }
        if (count == p) { s[0] += value; e[0] = s[0] / (double)p; s[1] = e[0]; continue; }
        /* stage k (1..5): e[k-1] is live; s[k] accumulates e[k-1] until count == (k+1)p-k seeds e[k]
         * -- except e[5], which the reference never seeds (:1042-1050). */
        e[0] = fma(alpha, value - e[0], e[0]);
        int done = 0;
        for (int k = 1; k <= 5 && !done; k++) {
            int64_t seed_at = (k + 1) * p - k;
            if (count < seed_at) { s[k] += e[k - 1]; done = 1; }
            else if (count == seed_at && k < 5) { s[k] += e[k - 1]; e[k] = s[k] / (double)p; s[k + 1] = e[k]; done = 1; }
            else if (count == seed_at && k == 5) {
                /* count == 6p-5 is the first '_' arm (:1153): full update incl. e5 from 0.0 */
                e[5] = fma(alpha, e[4] - e[5], e[5]);
                out[i] = fma(c1, e[5], fma(c2, e[4], fma(c3, e[3], c4 * e[2])));
                done = 1;
            } else {
                e[k] = fma(alpha, e[k - 1] - e[k], e[k]);
                if (k == 5) { out[i] = fma(c1, e[5], fma(c2, e[4], fma(c3, e[3], c4 * e[2]))); done = 1; }
            }
        }
    }
}

/* overlap.rs:1313-1326 calc_trima */
void pqo_trima(const double *v, int64_t n, int64_t p, double *out) {
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (p % 2 == 1) { int64_t k = p / 2 + 1; pqo_sma(v, n, k, tmp); pqo_sma(tmp, n, k, out); }
    else { int64_t k = p / 2; pqo_sma(v, n, k, tmp); pqo_sma(tmp, n, k + 1, out); }
    free(tmp);
}

/* overlap.rs:1328-1399 calc_wma (quirk Q-WMA reproduced literally) */
void pqo_wma(const double *v, int64_t n, int64_t p, double *out) {
    if (p <= 0 || n < p) { pqo_fill_null(out, n); return; }
    int64_t count = 0;
    double denominator = (double)(p * (p + 1) / 2);              /* :1338 */
    double numerator = 0.0, sum = 0.0;
    pqo_deque w; dq_init(&w, n);
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(v[i])) { out[i] = pqo_null(); continue; }
        double value = v[i];
        count += 1; sum += value;
        numerator += ((double)count) * value;                    /* :1356 */
        dq_push_back(&w, value);
        if (count < p) out[i] = pqo_null();
        else {
            if (count > p) {
                double old = dq_pop_front(&w);
                sum -= old; numerator -= ((double)p) * old; count -= 1; /* :1364-1366 */
            }
            out[i] = numerator / denominator;
        }
    }
    (void)sum;
    dq_free(&w);
}

/* overlap.rs:732-855 calc_kama.  Pass 2 does values.cont_slice().unwrap() (:826) which aborts
 * on nulls; the oracle keeps pass-1's null-skipping and defines pass 2 over the same rows
 * (values[i] at the sc row index), which is the reference behaviour for null-free input. */
void pqo_kama(const double *v, int64_t n, int64_t p, double *out) {
    /* p == 1: window_sum.pop_front().unwrap() (:775) hits an empty deque -> the reference aborts; all-null here */
    if (p <= 1 || n < p) { pqo_fill_null(out, n); return; }
    double *er = (double *)malloc(sizeof(double) * (size_t)n);
    int64_t count = 0;
    double diff_abs = 0.0, sum = 0.0;
    pqo_deque w, ws; dq_init(&w, n); dq_init(&ws, n);
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(v[i])) { er[i] = pqo_null(); continue; }
        double value = v[i];
        if (count == 0) { count += 1; dq_push_back(&w, value); er[i] = pqo_null(); }           /* :760-764 */
        else if (count < p) {
            count += 1;
            diff_abs = fabs(value - dq_front(&w));                                            /* :767 */
            sum += diff_abs; dq_push_back(&w, value); dq_push_back(&ws, diff_abs); er[i] = pqo_null();
        } else {
            diff_abs = fabs(value - dq_pop_front(&w));                                        /* :774 */
            sum += diff_abs - dq_pop_front(&ws);                                              /* :775 */
            dq_push_back(&w, value); dq_push_back(&ws, diff_abs);
            er[i] = diff_abs / sum;                                                           /* :778 */
        }
    }
    dq_free(&w); dq_free(&ws);
    double fast_sc = 2.0 / 3.0, slow_sc = 2.0 / 31.0;
    double k = fast_sc - slow_sc;
    int64_t c2 = 0;
    double kama = 0.0, sum2 = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(er[i])) { out[i] = pqo_null(); continue; }                              /* :830-833 */
        double sc_sqrt = er[i] * k + slow_sc;                                                 /* :818 */
        double sc = sc_sqrt * sc_sqrt;                                                        /* :819 */
        if (c2 < p) { c2 += 1; sum2 += v[i]; out[i] = pqo_null(); }                            /* :836-840 */
        else if (c2 == p) { c2 += 1; kama = sum2 / (double)p; out[i] = kama; }                /* :841-845 */
        else { kama = fma(sc, v[i] - kama, kama); out[i] = kama; }                            /* :847 */
    }
    free(er);
}

/* overlap.rs:857-869 calc_ma */
void pqo_ma(const double *v, int64_t n, int64_t p, int64_t matype, double *out) {
    switch (matype) {
    case 1: pqo_ema(v, n, p, out); break;
    case 2: pqo_wma(v, n, p, out); break;
    case 3: pqo_dema(v, n, p, out); break;
    case 4: pqo_tema(v, n, p, out); break;
    case 5: pqo_trima(v, n, p, out); break;
    case 6: pqo_kama(v, n, p, out); break;
    case 7: pqo_sma(v, n, p, out); break; /* "MAMA TODO" in the reference (:865) */
    case 8: pqo_t3(v, n, p, 0.0, out); break;
    default: pqo_sma(v, n, p, out); break;
    }
}

/* overlap.rs:180-278 midpoint.  Literal, including quirk Q-MID: the min-deque's expiry tests
 * window_max.front() (:227-231, :264-268), and `count - timeperiod` is a wrapping usize
 * subtraction.  No warm-up nulls. */
void pqo_midpoint(const double *v, int64_t n, int64_t p, double *out) {
    /* D-7b: timeperiod <= 0 is degenerate in the reference (for p == 0 the mis-targeted expiry test fires
     * whenever the new value is the running max); defined as all-null here and in the HIP path. */
    if (p <= 0) { pqo_fill_null(out, n); return; }
    uint64_t count = 0;
    double mx = 0.0, mn = 0.0;
    pqo_ideque wmax, wmin; idq_init(&wmax, n); idq_init(&wmin, n);
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(v[i])) { out[i] = pqo_null(); continue; }
        double value = v[i];
        count += 1;
        while (!idq_empty(&wmax) && wmax.val[wmax.tail - 1] <= value) wmax.tail--;            /* :205-211 */
        if (!idq_empty(&wmax) && wmax.idx[wmax.head] == count - (uint64_t)p) wmax.head++;     /* :212-216 */
        wmax.idx[wmax.tail] = count; wmax.val[wmax.tail] = value; wmax.tail++;
        mx = wmax.val[wmax.head];
        while (!idq_empty(&wmin) && wmin.val[wmin.tail - 1] >= value) wmin.tail--;            /* :220-226 */
        if (!idq_empty(&wmax) && wmax.idx[wmax.head] == count - (uint64_t)p) {                /* :227 (sic) */
            if (!idq_empty(&wmin)) wmin.head++;
        }
        wmin.idx[wmin.tail] = count; wmin.val[wmin.tail] = value; wmin.tail++;
        mn = wmin.val[wmin.head];
        out[i] = (mx + mn) / 2.0;                                                             /* :234 */
    }
    idq_free(&wmax); idq_free(&wmin);
}

/* overlap.rs:281-404 midprice, no-bitmap branches (:325-345 high, :378-398 low).  The bitmap
 * branch for `low` (:353-377) appends nulls to the *high* builder and uses max logic; it cannot
 * produce equal-length columns when a null is present, so (decision D-7) a null in either input
 * yields a null output row and does not advance that input's window. */
void pqo_midprice(const double *h, const double *l, int64_t n, int64_t p, double *out) {
    if (p <= 0) { pqo_fill_null(out, n); return; } /* D-7b */
    double *hm = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *lm = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    pqo_ideque w; idq_init(&w, n);
    uint64_t count = 0;
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(h[i])) { hm[i] = pqo_null(); continue; }
        double value = h[i];
        count += 1;
        while (!idq_empty(&w) && w.val[w.tail - 1] <= value) w.tail--;
        if (!idq_empty(&w) && w.idx[w.head] == count - (uint64_t)p) w.head++;
        w.idx[w.tail] = count; w.val[w.tail] = value; w.tail++;
        hm[i] = w.val[w.head];
    }
    w.head = w.tail = 0; count = 0;
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(l[i])) { lm[i] = pqo_null(); continue; }
        double value = l[i];
        count += 1;
        while (!idq_empty(&w) && w.val[w.tail - 1] >= value) w.tail--;
        if (!idq_empty(&w) && w.idx[w.head] == count - (uint64_t)p) w.head++;
        w.idx[w.tail] = count; w.val[w.tail] = value; w.tail++;
        lm[i] = w.val[w.head];
    }
    for (int64_t i = 0; i < n; i++)
        out[i] = (pqo_isnull(hm[i]) || pqo_isnull(lm[i])) ? pqo_null() : (hm[i] + lm[i]) / 2.0; /* :401 */
    idq_free(&w); free(hm); free(lm);
}

/* ------------------------------------------------------------------------------------------
 * Helpers the reference calls but never defines (overlap.rs:173, :432, :452, :478).
 * Decision D-4: defined here from the published TA-Lib algorithms (ta-lib 0.6.8 is the
 * reference's dev dependency, pyproject.toml:26-29), nulls -> 0.0 as the call sites do
 * (overlap.rs:162-171, :416-430, :445-450).  These definitions ARE the spec for the HIP path.
 * ------------------------------------------------------------------------------------------ */

static double n0(double x) { return pqo_isnull(x) ? 0.0 : x; }

/* shared Hilbert pipeline state == cycle.rs:21-24 */
typedef struct {
    double detrend[7], q1[7], i1[7];
    double i2, q2, re, im, period;
} ht_state;
void pqo__ht_step(ht_state *st, const double *smooth, int64_t i); /* cycle.c */

/* MAMA (D-4): reference Hilbert pipeline (cycle.rs:27-63) + TA-Lib's adaptive alpha:
 *   phase  = atan(q1/i1)*180/pi (0 if i1 == 0);  dphase = max(prev_phase - phase, 1)
 *   alpha  = fastlimit/dphase clamped to [slowlimit, fastlimit]
 *   mama   = alpha*x + (1-alpha)*mama ;  fama = 0.5*alpha*mama + (1-0.5*alpha)*fama
 * mama = fama = 0 before row 6; outputs from row 31 (aligned with cycle.rs:66); n < 32 -> null. */
void pqo_mama(const double *v, int64_t n, double fastlimit, double slowlimit, double *mama_o, double *fama_o) {
    pqo_fill_null(mama_o, n); pqo_fill_null(fama_o, n);
    if (n < 32) return;
    double *real = (double *)malloc(sizeof(double) * (size_t)n);
    double *smooth = (double *)calloc((size_t)n, sizeof(double));
    for (int64_t i = 0; i < n; i++) real[i] = n0(v[i]);
    for (int64_t i = 3; i < n; i++)
        smooth[i] = (4.0 * real[i] + 3.0 * real[i - 1] + 2.0 * real[i - 2] + real[i - 3]) * 0.1;
    ht_state st; memset(&st, 0, sizeof st);
    double mama = 0.0, fama = 0.0, prev_phase = 0.0;
    const double PI = 3.14159265358979323846;
    for (int64_t i = 6; i < n; i++) {
        pqo__ht_step(&st, smooth, i);
        double phase = (st.i1[0] != 0.0) ? atan(st.q1[0] / st.i1[0]) * 180.0 / PI : 0.0;
        double dphase = prev_phase - phase;
        if (dphase < 1.0) dphase = 1.0;
        double alpha = fastlimit / dphase;
        if (alpha < slowlimit) alpha = slowlimit;
        if (alpha > fastlimit) alpha = fastlimit;
        mama = alpha * real[i] + (1.0 - alpha) * mama;
        double ha = 0.5 * alpha;
        fama = ha * mama + (1.0 - ha) * fama;
        prev_phase = phase;
        if (i >= 31) { mama_o[i] = mama; fama_o[i] = fama; }
    }
    free(real); free(smooth);
}

/* MAVP (D-4): p_i = clamp(periods[i] as i64, minperiod, maxperiod); out[i] = calc_ma(real, p_i,
 * matype)[i]; rows i < maxperiod-1 are null. */
void pqo_mavp(const double *v, const double *periods, int64_t n, int64_t minp, int64_t maxp,
              int64_t matype, double *out) {
    pqo_fill_null(out, n);
    if (n <= 0 || minp > maxp || minp < 0) return;
    double *real = (double *)malloc(sizeof(double) * (size_t)n);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < n; i++) real[i] = n0(v[i]);
    for (int64_t P = minp; P <= maxp; P++) {
        int used = 0;
        for (int64_t i = 0; i < n && !used; i++) {
            int64_t pi = (int64_t)n0(periods[i]);
            if (pi < minp) pi = minp; if (pi > maxp) pi = maxp;
            if (pi == P) used = 1;
        }
        if (!used) continue;
        pqo_ma(real, n, P, matype, tmp);
        for (int64_t i = (maxp > 0 ? maxp - 1 : 0); i < n; i++) {
            int64_t pi = (int64_t)n0(periods[i]);
            if (pi < minp) pi = minp; if (pi > maxp) pi = maxp;
            if (pi == P) out[i] = tmp[i];
        }
    }
    free(real); free(tmp);
}

/* SAR (D-4): TA-Lib ta_SAR.c algorithm.  Row 0 null; n < 2 -> all null. */
void pqo_sar(const double *hi, const double *lo, int64_t n, double accel, double maxv, double *out) {
    pqo_fill_null(out, n);
    if (n < 2) return;
    double af = accel;
    if (af > maxv) { af = accel = maxv; }
    double h0 = n0(hi[0]), l0 = n0(lo[0]), h1 = n0(hi[1]), l1 = n0(lo[1]);
    double diffP = h1 - h0, diffM = l0 - l1;
    int is_long = !(diffM > 0.0 && diffP < diffM);
    double ep, sar;
    if (is_long) { ep = h1; sar = l0; } else { ep = l1; sar = h0; }
    double new_low = l1, new_high = h1;
    for (int64_t t = 1; t < n; t++) {
        double prev_low = new_low, prev_high = new_high;
        new_low = n0(lo[t]); new_high = n0(hi[t]);
        if (is_long) {
            if (new_low <= sar) {
                is_long = 0; sar = ep;
                if (sar < prev_high) sar = prev_high;
                if (sar < new_high) sar = new_high;
                out[t] = sar;
                af = accel; ep = new_low;
                sar = sar + af * (ep - sar);
                if (sar < prev_high) sar = prev_high;
                if (sar < new_high) sar = new_high;
            } else {
                out[t] = sar;
                if (new_high > ep) { ep = new_high; af += accel; if (af > maxv) af = maxv; }
                sar = sar + af * (ep - sar);
                if (sar > prev_low) sar = prev_low;
                if (sar > new_low) sar = new_low;
            }
        } else {
            if (new_high >= sar) {
                is_long = 1; sar = ep;
                if (sar > prev_low) sar = prev_low;
                if (sar > new_low) sar = new_low;
                out[t] = sar;
                af = accel; ep = new_high;
                sar = sar + af * (ep - sar);
                if (sar > prev_low) sar = prev_low;
                if (sar > new_low) sar = new_low;
            } else {
                out[t] = sar;
                if (new_low < ep) { ep = new_low; af += accel; if (af > maxv) af = maxv; }
                sar = sar + af * (ep - sar);
                if (sar < prev_high) sar = prev_high;
                if (sar < new_high) sar = new_high;
            }
        }
    }
}

/* SAREXT (D-4): TA-Lib ta_SAREXT.c algorithm; values emitted while short are negated. */
void pqo_sarext(const double *hi, const double *lo, int64_t n, double startvalue,
                double offsetonreverse, double ai_long, double a_long, double am_long,
                double ai_short, double a_short, double am_short, double *out) {
    pqo_fill_null(out, n);
    if (n < 2) return;
    double af_long = ai_long, af_short = ai_short;
    if (af_long > am_long) af_long = ai_long = am_long;
    if (a_long > am_long) a_long = am_long;
    if (af_short > am_short) af_short = ai_short = am_short;
    if (a_short > am_short) a_short = am_short;
    double h0 = n0(hi[0]), l0 = n0(lo[0]), h1 = n0(hi[1]), l1 = n0(lo[1]);
    int is_long;
    double ep, sar;
    if (startvalue == 0.0) {
        double diffP = h1 - h0, diffM = l0 - l1;
        is_long = !(diffM > 0.0 && diffP < diffM);
        if (is_long) { ep = h1; sar = l0; } else { ep = l1; sar = h0; }
    } else if (startvalue > 0.0) { is_long = 1; ep = h1; sar = startvalue; }
    else { is_long = 0; ep = l1; sar = fabs(startvalue); }
    double new_low = l1, new_high = h1;
    for (int64_t t = 1; t < n; t++) {
        double prev_low = new_low, prev_high = new_high;
        new_low = n0(lo[t]); new_high = n0(hi[t]);
        if (is_long) {
            if (new_low <= sar) {
                is_long = 0; sar = ep;
                if (sar < prev_high) sar = prev_high;
                if (sar < new_high) sar = new_high;
                if (offsetonreverse != 0.0) sar += sar * offsetonreverse;
                out[t] = -sar;
                af_short = ai_short; ep = new_low;
                sar = sar + af_short * (ep - sar);
                if (sar < prev_high) sar = prev_high;
                if (sar < new_high) sar = new_high;
            } else {
                out[t] = sar;
                if (new_high > ep) { ep = new_high; af_long += a_long; if (af_long > am_long) af_long = am_long; }
                sar = sar + af_long * (ep - sar);
                if (sar > prev_low) sar = prev_low;
                if (sar > new_low) sar = new_low;
            }
        } else {
            if (new_high >= sar) {
                is_long = 1; sar = ep;
                if (sar > prev_low) sar = prev_low;
                if (sar > new_low) sar = new_low;
                if (offsetonreverse != 0.0) sar -= sar * offsetonreverse;
                out[t] = sar;
                af_long = ai_long; ep = new_high;
                sar = sar + af_long * (ep - sar);
                if (sar > prev_low) sar = prev_low;
                if (sar > new_low) sar = new_low;
            } else {
                out[t] = -sar;
                if (new_low < ep) { ep = new_low; af_short += a_short; if (af_short > am_short) af_short = am_short; }
                sar = sar + af_short * (ep - sar);
                if (sar < prev_high) sar = prev_high;
                if (sar < new_high) sar = new_high;
            }
        }
    }
}
