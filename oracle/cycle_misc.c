/* cycle_misc.c -- CPU ORACLE (test infrastructure) for src/talib/{cycle,volatility,volume,price}.rs.
 * Compile with -ffp-contract=off. */
#include "pqo_common.h"

/* ===================== cycle.rs ===================== */
typedef struct {
    double detrend[7], q1[7], i1[7];
    double i2, q2, re, im, period;
} ht_state;

/* cycle.rs:455-460 shift_push7 */
static void shift_push7(double *dq, double val) {
    for (int i = 6; i >= 1; i--) dq[i] = dq[i - 1];
    dq[0] = val;
}
/* cycle.rs:462-470 calc_smooth */
static double *calc_smooth(const double *real, int64_t n) {
    double *smooth = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    for (int64_t i = 3; i < n; i++)
        smooth[i] = (4.0 * real[i] + 3.0 * real[i - 1] + 2.0 * real[i - 2] + real[i - 3]) * 0.1;
    return smooth;
}
static double clampd(double x, double lo, double hi) { /* f64::clamp */
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
/* one iteration of the shared pipeline, cycle.rs:28-63 (identical in all six functions) */
void pqo__ht_step(ht_state *st, const double *smooth, int64_t i) {
    const double TAU = 6.28318530717958647692;
    double prev_period = (i > 6) ? st->period : 6.0;                                     /* :28 */
    double adj = 0.075 * prev_period + 0.54;                                             /* :29 */
    double detrend_curr = (0.0962 * smooth[i] + 0.5769 * smooth[i - 2] - 0.5769 * smooth[i - 4]
                           - 0.0962 * smooth[i - 6]) * adj;                              /* :31-34 */
    shift_push7(st->detrend, detrend_curr);
    double q1_curr = (0.0962 * st->detrend[0] + 0.5769 * st->detrend[2] - 0.5769 * st->detrend[4]
                      - 0.0962 * st->detrend[6]) * adj;                                  /* :37-39 */
    shift_push7(st->q1, q1_curr);
    shift_push7(st->i1, st->detrend[3]);                                                 /* :41 */
    double ji = (0.0962 * st->i1[0] + 0.5769 * st->i1[2] - 0.5769 * st->i1[4] - 0.0962 * st->i1[6]) * adj;
    double jq = (0.0962 * st->q1[0] + 0.5769 * st->q1[2] - 0.5769 * st->q1[4] - 0.0962 * st->q1[6]) * adj;
    double i2_curr = 0.2 * (st->i1[0] - jq) + 0.8 * st->i2;                              /* :46 */
    double q2_curr = 0.2 * (st->q1[0] + ji) + 0.8 * st->q2;                              /* :47 */
    double re_curr = 0.2 * (i2_curr * st->i2 + q2_curr * st->q2) + 0.8 * st->re;         /* :49 */
    double im_curr = 0.2 * (i2_curr * st->q2 - q2_curr * st->i2) + 0.8 * st->im;         /* :50 */
    st->i2 = i2_curr; st->q2 = q2_curr; st->re = re_curr; st->im = im_curr;
    if (st->im != 0.0 && st->re != 0.0) st->period = TAU / atan(st->im / st->re);        /* :57-59 */
    st->period = clampd(clampd(st->period, 0.67 * prev_period, 1.5 * prev_period), 6.0, 50.0); /* :60-62 */
    st->period = 0.2 * st->period + 0.8 * prev_period;                                   /* :63 */
}

static const double PI_ = 3.14159265358979323846;

/* cycle.rs:10-72 */
void pqo_ht_dcperiod(const double *v, int64_t n, double *out) {
    pqo_fill_null(out, n);
    if (n < 32) return;
    double *smooth = calc_smooth(v, n);
    ht_state st; memset(&st, 0, sizeof st);
    double smooth_period = 0.0;
    for (int64_t i = 6; i < n; i++) {
        pqo__ht_step(&st, smooth, i);
        smooth_period = 0.33 * st.period + 0.67 * smooth_period;                         /* :64 */
        if (i >= 31) out[i] = smooth_period;
    }
    free(smooth);
}
/* cycle.rs:75-147 */
void pqo_ht_dcphase(const double *v, int64_t n, double *out) {
    pqo_fill_null(out, n);
    if (n < 32) return;
    double *smooth = calc_smooth(v, n);
    ht_state st; memset(&st, 0, sizeof st);
    for (int64_t i = 6; i < n; i++) {
        pqo__ht_step(&st, smooth, i);
        if (i >= 31) {
            double dc_phase = (st.i1[0] != 0.0) ? atan(st.q1[0] / st.i1[0]) * 180.0 / PI_ : 0.0; /* :130-134 */
            dc_phase += 90.0;
            if (st.i1[0] < 0.0) dc_phase += 180.0;
            if (dc_phase > 315.0) dc_phase -= 360.0;
            out[i] = dc_phase;
        }
    }
    free(smooth);
}
/* cycle.rs:159-227 */
void pqo_ht_phasor(const double *v, int64_t n, double *inphase, double *quadrature) {
    pqo_fill_null(inphase, n); pqo_fill_null(quadrature, n);
    if (n < 32) return;
    double *smooth = calc_smooth(v, n);
    ht_state st; memset(&st, 0, sizeof st);
    for (int64_t i = 6; i < n; i++) {
        pqo__ht_step(&st, smooth, i);
        if (i >= 31) { inphase[i] = st.i1[0]; quadrature[i] = st.q1[0]; }
    }
    free(smooth);
}
/* cycle.rs:236-307 */
void pqo_ht_sine(const double *v, int64_t n, double *sine, double *leadsine) {
    pqo_fill_null(sine, n); pqo_fill_null(leadsine, n);
    if (n < 32) return;
    double *smooth = calc_smooth(v, n);
    ht_state st; memset(&st, 0, sizeof st);
    for (int64_t i = 6; i < n; i++) {
        pqo__ht_step(&st, smooth, i);
        if (i >= 31) {
            double dc_phase = (st.i1[0] != 0.0) ? atan(st.q1[0] / st.i1[0]) * 180.0 / PI_ : 0.0;
            sine[i] = sin(dc_phase * PI_ / 180.0);                                       /* :299 */
            leadsine[i] = sin((dc_phase + 45.0) * PI_ / 180.0);                          /* :300 */
        }
    }
    free(smooth);
}
/* cycle.rs:310-374 (the pipeline is computed but unused) */
void pqo_ht_trendline(const double *v, int64_t n, double *out) {
    pqo_fill_null(out, n);
    if (n < 32) return;
    for (int64_t i = 31; i < n; i++) {
        double trendline = 0.0;
        for (int j = 0; j < 4; j++) trendline += v[i - j];                               /* :365-368 */
        out[i] = trendline * 0.25;
    }
}
/* cycle.rs:377-448 */
void pqo_ht_trendmode(const double *v, int64_t n, int32_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = PQO_NULL_I32;
    if (n < 32) return;
    for (int64_t i = 31; i < n; i++) {
        double trendline = 0.0;
        for (int j = 0; j < 4; j++) trendline += v[i - j];
        trendline *= 0.25;
        out[i] = (fabs(v[i] - trendline) > 0.01 * trendline) ? 1 : 0;                    /* :438-442 */
    }
}

/* ===================== volatility.rs ===================== */
/* volatility.rs:67-84 calc_trange (N-C; pre_close = close.shift(1) -> row 0 null) */
void pqo_trange(const double *h, const double *l, const double *c, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) {
        if (i == 0 || pqo_isnull(h[i]) || pqo_isnull(l[i]) || pqo_isnull(c[i - 1])) { out[i] = pqo_null(); continue; }
        double pc = c[i - 1];
        out[i] = RMAX(RMAX(h[i] - l[i], fabs(h[i] - pc)), fabs(l[i] - pc));              /* :77 */
    }
}
/* volatility.rs:18-31 */
void pqo_atr(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out) {
    double *tr = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    pqo_trange(h, l, c, n, tr);
    pqo_ema(tr, n, 2 * p - 1, out);                                                      /* :30 */
    free(tr);
}
/* volatility.rs:34-48: (&atr / close * 100) */
void pqo_natr(const double *h, const double *l, const double *c, int64_t n, int64_t p, double *out) {
    pqo_atr(h, l, c, n, p, out);
    for (int64_t i = 0; i < n; i++)
        out[i] = (pqo_isnull(out[i]) || pqo_isnull(c[i])) ? pqo_null() : out[i] / c[i] * 100.0;
}

/* ===================== volume.rs ===================== */
/* volume.rs:100-126 calc_ad (quirk Q-AD: emits 0.0, not the running sum, when h == l) */
void pqo_ad(const double *h, const double *l, const double *c, const double *vol, int64_t n, double *out) {
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(h[i]) || pqo_isnull(l[i]) || pqo_isnull(c[i]) || pqo_isnull(vol[i])) { out[i] = pqo_null(); continue; }
        double diff = h[i] - l[i];
        if (diff == 0.0) out[i] = 0.0;                                                   /* :115-116 */
        else { sum += (2.0 * c[i] - l[i] - h[i]) / diff * vol[i]; out[i] = sum; }        /* :118 */
    }
}
/* volume.rs:34-67 (quirk Q-ADOSC: cum-sums the already cumulative AD) */
void pqo_adosc(const double *h, const double *l, const double *c, const double *vol, int64_t n,
               int64_t fast, int64_t slow, double *out) {
    size_t m = (size_t)(n > 0 ? n : 1);
    double *ad = (double *)malloc(sizeof(double) * m), *adl = (double *)malloc(sizeof(double) * m);
    double *ef = (double *)malloc(sizeof(double) * m), *es = (double *)malloc(sizeof(double) * m);
    pqo_ad(h, l, c, vol, n, ad);
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (pqo_isnull(ad[i])) adl[i] = pqo_null();
        else { sum += ad[i]; adl[i] = sum; }                                             /* :50-58 */
    }
    pqo_ema(adl, n, fast, ef); pqo_ema(adl, n, slow, es);
    for (int64_t i = 0; i < n; i++)
        out[i] = (pqo_isnull(ef[i]) || pqo_isnull(es[i])) ? pqo_null() : ef[i] - es[i];   /* :65 */
    free(ad); free(adl); free(ef); free(es);
}
/* volume.rs:70-94 (quirk Q-OBV: close_diff = prev_close - close, so the sign is inverted) */
void pqo_obv(const double *c, const double *vol, int64_t n, double *out) {
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (i == 0 || pqo_isnull(c[i]) || pqo_isnull(c[i - 1]) || pqo_isnull(vol[i])) { out[i] = pqo_null(); continue; }
        double c_diff = c[i - 1] - c[i];                                                 /* :78 */
        if (c_diff > 0.0) sum += vol[i];
        else if (c_diff < 0.0) sum -= vol[i];
        out[i] = sum;
    }
}

/* ===================== price.rs ===================== */
#define ANY4NULL(a, b, c, d) (pqo_isnull(a) || pqo_isnull(b) || pqo_isnull(c) || pqo_isnull(d))
void pqo_avgprice(const double *o, const double *h, const double *l, const double *c, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++)
        out[i] = ANY4NULL(o[i], h[i], l[i], c[i]) ? pqo_null() : (o[i] + h[i] + l[i] + c[i]) * 0.25; /* price.rs:25 */
}
void pqo_medprice(const double *h, const double *l, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++)
        out[i] = (pqo_isnull(h[i]) || pqo_isnull(l[i])) ? pqo_null() : (h[i] + l[i]) * 0.5;          /* price.rs:44 */
}
void pqo_typprice(const double *h, const double *l, const double *c, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++)
        out[i] = ANY4NULL(h[i], l[i], c[i], 0.0) ? pqo_null() : (h[i] + l[i] + c[i]) / 3.0;          /* price.rs:65 */
}
void pqo_wclprice(const double *h, const double *l, const double *c, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++)
        out[i] = ANY4NULL(h[i], l[i], c[i], 0.0) ? pqo_null() : (h[i] + l[i] + 2.0 * c[i]) / 4.0;    /* price.rs:86 */
}
