/* pattern.c -- CPU ORACLE (test infrastructure) for src/talib/pattern.rs: 61 candlestick
 * recognisers, one literal loop each.  O(k)/H(k)/L(k)/C(k) = value k rows back from row i.
 * Int32 out in {-100, 0, +100}; rows below the look-back are 0.  Compile with -ffp-contract=off. */
#include "pqo_common.h"

/* pattern.rs:2067-2143 predicate helpers, products written exactly as in the source */
static inline int bull(double o, double c) { return c > o; }
static inline int bear(double o, double c) { return c < o; }
static inline double body_abs(double o, double c) { return fabs(o - c); }
static inline double oc_min(double o, double c) { return RMIN(o, c); }
static inline double oc_max(double o, double c) { return RMAX(o, c); }
static inline double upper_shadow(double o, double h, double c) { return h - oc_max(o, c); }
static inline double lower_shadow(double o, double l, double c) { return oc_min(o, c) - l; }
static inline int long_body(double o, double c) { return body_abs(o, c) > 0.05 * (o + c) * 0.5; }
static inline int short_body(double o, double c) { return body_abs(o, c) < 0.1 * (o + c) * 0.5; }
static inline int doji(double o, double h, double l, double c) { (void)h; (void)l; return body_abs(o, c) <= 0.005 * (o + c) * 0.5; }
static inline int long_up_shadow(double o, double h, double c) { return upper_shadow(o, h, c) > 2.0 * body_abs(o, c); }
static inline int long_dn_shadow(double o, double l, double c) { return lower_shadow(o, l, c) > 2.0 * body_abs(o, c); }
static inline int short_up_shadow(double o, double h, double l, double c) { (void)l; return upper_shadow(o, h, c) < 0.5 * body_abs(o, c); }
static inline int short_dn_shadow(double o, double h, double l, double c) { (void)h; return lower_shadow(o, l, c) < 0.5 * body_abs(o, c); }
static inline int vshort_up_shadow(double o, double h, double l, double c) { (void)l; return upper_shadow(o, h, c) < 0.1 * body_abs(o, c); }
static inline int vshort_dn_shadow(double o, double h, double l, double c) { (void)h; return lower_shadow(o, l, c) < 0.1 * body_abs(o, c); }
static inline int vlong_dn_shadow(double o, double l, double c) { return lower_shadow(o, l, c) > 3.0 * body_abs(o, c); }
static inline int near_(double v1, double v2, double h, double l) { return fabs(v1 - v2) < 0.01 * (h + l) * 0.5; }
static inline int equal_(double v1, double v2, double h, double l) { return fabs(v1 - v2) < 0.001 * (h + l) * 0.5; }

#define O(k) op[i - (k)]
#define H(k) hi[i - (k)]
#define L(k) lo[i - (k)]
#define C(k) cl[i - (k)]
#define ARGS const double *op, const double *hi, const double *lo, const double *cl, int64_t n, double pen, int32_t *out
#define LOOP(lb) for (int64_t i = (lb); i < n; i++)
#define UNUSED (void)op; (void)hi; (void)lo; (void)cl; (void)pen

/* pattern.rs:10-40 */
static void cdl2crows(ARGS) { UNUSED; LOOP(2) {
    int m = bull(O(2), C(2)) && long_body(O(2), C(2)) && bear(O(1), C(1)) && (O(1) > C(2)) && bear(O(0), C(0))
            && (O(0) > O(1)) && (O(0) < C(1)) && (C(0) > O(2)) && (C(0) < C(2));
    if (m) out[i] = -100; } }
/* :43-73 */
static void cdl3blackcrows(ARGS) { UNUSED; LOOP(2) {
    int m = bear(O(2), C(2)) && long_body(O(2), C(2)) && bear(O(1), C(1)) && long_body(O(1), C(1))
            && bear(O(0), C(0)) && long_body(O(0), C(0)) && (O(1) < O(2)) && (O(1) > C(2))
            && (O(0) < O(1)) && (O(0) > C(1)) && (C(1) < C(2)) && (C(0) < C(1));
    if (m) out[i] = -100; } }
/* :76-111 */
static void cdl3inside(ARGS) { UNUSED; LOOP(2) {
    int b = bear(O(2), C(2)) && long_body(O(2), C(2)) && bull(O(1), C(1)) && (C(1) < O(2)) && (O(1) > C(2))
            && bull(O(0), C(0)) && (C(0) > O(2));
    int s = bull(O(2), C(2)) && long_body(O(2), C(2)) && bear(O(1), C(1)) && (O(1) < C(2)) && (C(1) > O(2))
            && bear(O(0), C(0)) && (C(0) < O(2));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :114-157 (o1..o3 = rows i-3..i-1) */
static void cdl3linestrike(ARGS) { UNUSED; LOOP(3) {
    int b3 = bear(O(3), C(3)) && bear(O(2), C(2)) && bear(O(1), C(1)) && (C(2) < C(3)) && (C(1) < C(2))
             && (O(2) > C(3)) && (O(2) < O(3)) && (O(1) > C(2)) && (O(1) < O(2));
    int bs = bull(O(0), C(0)) && (O(0) < C(1)) && (C(0) > O(3));
    int s3 = bull(O(3), C(3)) && bull(O(2), C(2)) && bull(O(1), C(1)) && (C(2) > C(3)) && (C(1) > C(2))
             && (O(2) < C(3)) && (O(2) > O(3)) && (O(1) < C(2)) && (O(1) > O(2));
    int ss = bear(O(0), C(0)) && (O(0) > C(1)) && (C(0) < O(3));
    if (b3 && bs) out[i] = 100; else if (s3 && ss) out[i] = -100; } }
/* :160-191 */
static void cdl3outside(ARGS) { UNUSED; LOOP(2) {
    int b = bear(O(2), C(2)) && bull(O(1), C(1)) && (O(1) <= C(2)) && (C(1) >= O(2)) && bull(O(0), C(0)) && (C(0) > C(1));
    int s = bull(O(2), C(2)) && bear(O(1), C(1)) && (O(1) >= C(2)) && (C(1) <= O(2)) && bear(O(0), C(0)) && (C(0) < C(1));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :194-231 */
static void cdl3starsinsouth(ARGS) { UNUSED; LOOP(2) {
    int m = bear(O(2), C(2)) && long_body(O(2), C(2)) && long_dn_shadow(O(2), L(2), C(2)) && bear(O(1), C(1))
            && (L(1) > L(2)) && (C(1) > C(2)) && bear(O(0), C(0)) && short_body(O(0), C(0))
            && (H(0) < H(1)) && (L(0) > L(1));
    if (m) out[i] = 100; } }
/* :234-265 */
static void cdl3whitesoldiers(ARGS) { UNUSED; LOOP(2) {
    int m = bull(O(2), C(2)) && long_body(O(2), C(2)) && bull(O(1), C(1)) && long_body(O(1), C(1))
            && bull(O(0), C(0)) && long_body(O(0), C(0)) && (O(1) > O(2)) && (O(1) <= C(2))
            && (O(0) > O(1)) && (O(0) <= C(1)) && (C(1) > C(2)) && (C(0) > C(1));
    if (m) out[i] = 100; } }
/* :268-306 (penetration accepted by the Python wrapper, ignored by the Rust) */
static void cdlabandonedbaby(ARGS) { UNUSED; LOOP(2) {
    int d2 = doji(O(1), H(1), L(1), C(1));
    int b = bear(O(2), C(2)) && long_body(O(2), C(2)) && d2 && (H(1) < L(2)) && bull(O(0), C(0)) && (L(0) > H(1));
    int s = bull(O(2), C(2)) && long_body(O(2), C(2)) && d2 && (L(1) > H(2)) && bear(O(0), C(0)) && (H(0) < L(1));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :309-342 */
static void cdladvanceblock(ARGS) { UNUSED; LOOP(2) {
    int m = bull(O(2), C(2)) && long_body(O(2), C(2)) && bull(O(1), C(1)) && bull(O(0), C(0))
            && (O(1) > O(2)) && (O(1) <= C(2)) && (O(0) > O(1)) && (O(0) <= C(1)) && (C(1) > C(2)) && (C(0) > C(1))
            && (body_abs(O(0), C(0)) < body_abs(O(1), C(1)));
    if (m) out[i] = -100; } }
/* :345-370 */
static void cdlbelthold(ARGS) { UNUSED; LOOP(0) {
    int b = bull(O(0), C(0)) && long_body(O(0), C(0)) && vshort_dn_shadow(O(0), H(0), L(0), C(0));
    int s = bear(O(0), C(0)) && long_body(O(0), C(0)) && vshort_up_shadow(O(0), H(0), L(0), C(0));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :373-411 (o1,o2 = rows i-4,i-3; c3 = row i-2) */
static void cdlbreakaway(ARGS) { UNUSED; LOOP(4) {
    int b = bear(O(4), C(4)) && long_body(O(4), C(4)) && bear(O(3), C(3)) && (O(3) < C(4)) && (C(2) < C(3))
            && bull(O(0), C(0)) && (C(0) > O(3)) && (C(0) < C(4));
    int s = bull(O(4), C(4)) && long_body(O(4), C(4)) && bull(O(3), C(3)) && (O(3) > C(4)) && (C(2) > C(3))
            && bear(O(0), C(0)) && (C(0) < O(3)) && (C(0) > C(4));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :414-439 */
static void cdlclosingmarubozu(ARGS) { UNUSED; LOOP(0) {
    int b = bull(O(0), C(0)) && long_body(O(0), C(0)) && vshort_up_shadow(O(0), H(0), L(0), C(0));
    int s = bear(O(0), C(0)) && long_body(O(0), C(0)) && vshort_dn_shadow(O(0), H(0), L(0), C(0));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :442-484 */
static void cdlconcealbabyswall(ARGS) { UNUSED; LOOP(3) {
    int m = bear(O(3), C(3)) && long_body(O(3), C(3))
            && vshort_up_shadow(O(3), H(3), L(3), C(3)) && vshort_dn_shadow(O(3), H(3), L(3), C(3))
            && bear(O(2), C(2)) && long_body(O(2), C(2))
            && vshort_up_shadow(O(2), H(2), L(2), C(2)) && vshort_dn_shadow(O(2), H(2), L(2), C(2))
            && (C(2) < C(3)) && bear(O(1), C(1)) && (H(1) > C(2)) && bear(O(0), C(0)) && long_body(O(0), C(0))
            && (O(0) > H(1)) && (C(0) < L(2));
    if (m) out[i] = 100; } }
/* :487-516 */
static void cdlcounterattack(ARGS) { UNUSED; LOOP(1) {
    int b = bear(O(1), C(1)) && long_body(O(1), C(1)) && bull(O(0), C(0)) && long_body(O(0), C(0)) && near_(C(0), C(1), H(0), L(0));
    int s = bull(O(1), C(1)) && long_body(O(1), C(1)) && bear(O(0), C(0)) && long_body(O(0), C(0)) && near_(C(0), C(1), H(0), L(0));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :519-550 */
static void cdldarkcloudcover(ARGS) { UNUSED; LOOP(1) {
    int m = bull(O(1), C(1)) && long_body(O(1), C(1)) && bear(O(0), C(0)) && (O(0) > C(1))
            && (C(0) < (C(1) - (body_abs(O(1), C(1)) * pen))) && (C(0) > O(1));
    if (m) out[i] = -100; } }
/* :553-575 */
static void cdldoji(ARGS) { UNUSED; LOOP(0) { if (doji(O(0), H(0), L(0), C(0))) out[i] = 100; } }
/* :578-607 */
static void cdldojistar(ARGS) { UNUSED; LOOP(1) {
    int d = doji(O(0), H(0), L(0), C(0));
    double mid = (O(0) + C(0)) / 2.0;
    int b = bear(O(1), C(1)) && long_body(O(1), C(1)) && d && (mid < C(1));
    int s = bull(O(1), C(1)) && long_body(O(1), C(1)) && d && (mid > C(1));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :610-632 */
static void cdldragonflydoji(ARGS) { UNUSED; LOOP(0) {
    if (doji(O(0), H(0), L(0), C(0)) && long_dn_shadow(O(0), L(0), C(0)) && vshort_up_shadow(O(0), H(0), L(0), C(0))) out[i] = 100; } }
/* :635-662 */
static void cdlengulfing(ARGS) { UNUSED; LOOP(1) {
    int b = bear(O(1), C(1)) && bull(O(0), C(0)) && (O(0) <= C(1)) && (C(0) >= O(1)) && ((O(0) < C(1)) || (C(0) > O(1)));
    int s = bull(O(1), C(1)) && bear(O(0), C(0)) && (O(0) >= C(1)) && (C(0) <= O(1)) && ((O(0) > C(1)) || (C(0) < O(1)));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :665-700 */
static void cdleveningdojistar(ARGS) { UNUSED; LOOP(2) {
    int m = bull(O(2), C(2)) && long_body(O(2), C(2)) && doji(O(1), H(1), L(1), C(1)) && (oc_min(O(1), C(1)) > C(2))
            && bear(O(0), C(0)) && (C(0) < (C(2) - (body_abs(O(2), C(2)) * pen)));
    if (m) out[i] = -100; } }
/* :703-736 */
static void cdleveningstar(ARGS) { UNUSED; LOOP(2) {
    int m = bull(O(2), C(2)) && long_body(O(2), C(2)) && short_body(O(1), C(1)) && (oc_min(O(1), C(1)) > C(2))
            && bear(O(0), C(0)) && (C(0) < (C(2) - (body_abs(O(2), C(2)) * pen)));
    if (m) out[i] = -100; } }
/* :739-774 */
static void cdlgapsidesidewhite(ARGS) { UNUSED; LOOP(2) {
    int b2 = bull(O(1), C(1)), b3 = bull(O(0), C(0));
    int sim = near_(body_abs(O(0), C(0)), body_abs(O(1), C(1)), H(0), L(0));
    int so = near_(O(0), O(1), H(0), L(0));
    int up = bull(O(2), C(2)) && (O(1) > C(2)) && b2 && b3 && sim && so;
    int dn = bear(O(2), C(2)) && (C(1) < C(2)) && b2 && b3 && sim && so;
    if (up) out[i] = 100; else if (dn) out[i] = -100; } }
/* :777-799 */
static void cdlgravestonedoji(ARGS) { UNUSED; LOOP(0) {
    if (doji(O(0), H(0), L(0), C(0)) && long_up_shadow(O(0), H(0), C(0)) && vshort_dn_shadow(O(0), H(0), L(0), C(0))) out[i] = -100; } }
/* :802-829 */
static void cdlhammer(ARGS) { UNUSED; LOOP(1) {
    double ba = body_abs(O(0), C(0)), ls = lower_shadow(O(0), L(0), C(0));
    int m = short_body(O(0), C(0)) && (ls > (2.0 * ba)) && vshort_up_shadow(O(0), H(0), L(0), C(0));
    if (m && bear(O(1), C(1))) out[i] = 100; } }
/* :832-859 */
static void cdlhangingman(ARGS) { UNUSED; LOOP(1) {
    double ba = body_abs(O(0), C(0)), ls = lower_shadow(O(0), L(0), C(0));
    int m = short_body(O(0), C(0)) && (ls > (2.0 * ba)) && vshort_up_shadow(O(0), H(0), L(0), C(0));
    if (m && bull(O(1), C(1))) out[i] = -100; } }
/* :862-893 */
static void cdlharami(ARGS) { UNUSED; LOOP(1) {
    int b = bear(O(1), C(1)) && long_body(O(1), C(1)) && bull(O(0), C(0)) && short_body(O(0), C(0)) && (O(0) > C(1)) && (C(0) < O(1));
    int s = bull(O(1), C(1)) && long_body(O(1), C(1)) && bear(O(0), C(0)) && short_body(O(0), C(0)) && (O(0) < C(1)) && (C(0) > O(1));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :896-926 */
static void cdlharamicross(ARGS) { UNUSED; LOOP(1) {
    int d = doji(O(0), H(0), L(0), C(0));
    int b = bear(O(1), C(1)) && long_body(O(1), C(1)) && d && (oc_max(O(0), C(0)) < O(1)) && (oc_min(O(0), C(0)) > C(1));
    int s = bull(O(1), C(1)) && long_body(O(1), C(1)) && d && (oc_max(O(0), C(0)) < C(1)) && (oc_min(O(0), C(0)) > O(1));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :929-953 */
static void cdlhighwave(ARGS) { UNUSED; LOOP(0) {
    int m = short_body(O(0), C(0)) && long_up_shadow(O(0), H(0), C(0)) && long_dn_shadow(O(0), L(0), C(0));
    if (m && bull(O(0), C(0))) out[i] = 100; else if (m && bear(O(0), C(0))) out[i] = -100; } }
/* :956-984 */
static void cdlhikkake(ARGS) { UNUSED; LOOP(2) {
    int inside = (H(1) < H(2)) && (L(1) > L(2));
    int b = inside && (C(0) > H(2)) && bull(O(0), C(0));
    int s = inside && (C(0) < L(2)) && bear(O(0), C(0));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :987-1018 */
static void cdlhikkakemod(ARGS) { UNUSED; LOOP(3) {
    int inside = (H(2) < H(3)) && (L(2) > L(3));
    int second = (H(1) < H(2)) && (L(1) > L(2));
    int b = inside && second && (C(0) > H(3)) && bull(O(0), C(0));
    int s = inside && second && (C(0) < L(3)) && bear(O(0), C(0));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :1021-1045 */
static void cdlhomingpigeon(ARGS) { UNUSED; LOOP(1) {
    int m = bear(O(1), C(1)) && long_body(O(1), C(1)) && bear(O(0), C(0)) && short_body(O(0), C(0)) && (O(0) < O(1)) && (C(0) > C(1));
    if (m) out[i] = 100; } }
/* :1048-1080 */
static void cdlidentical3crows(ARGS) { UNUSED; LOOP(2) {
    int m = bear(O(2), C(2)) && long_body(O(2), C(2)) && bear(O(1), C(1)) && long_body(O(1), C(1))
            && bear(O(0), C(0)) && long_body(O(0), C(0)) && equal_(O(1), C(2), H(0), L(0)) && equal_(O(0), C(1), H(0), L(0))
            && (C(1) < C(2)) && (C(0) < C(1));
    if (m) out[i] = -100; } }
/* :1083-1108 */
static void cdlinneck(ARGS) { UNUSED; LOOP(1) {
    int m = bear(O(1), C(1)) && long_body(O(1), C(1)) && bull(O(0), C(0)) && (O(0) < C(1)) && near_(C(0), C(1), H(0), L(0));
    if (m) out[i] = -100; } }
/* :1111-1138 */
static void cdlinvertedhammer(ARGS) { UNUSED; LOOP(1) {
    double ba = body_abs(O(0), C(0)), us = upper_shadow(O(0), H(0), C(0));
    int m = short_body(O(0), C(0)) && (us > (2.0 * ba)) && vshort_dn_shadow(O(0), H(0), L(0), C(0));
    if (m && bear(O(1), C(1))) out[i] = 100; } }
static inline int maru(double o, double h, double l, double c) { return long_body(o, c) && vshort_up_shadow(o, h, l, c) && vshort_dn_shadow(o, h, l, c); }
/* :1141-1180 */
static void cdlkicking(ARGS) { UNUSED; LOOP(1) {
    int m1bear = bear(O(1), C(1)) && maru(O(1), H(1), L(1), C(1));
    int m1bull = bull(O(1), C(1)) && maru(O(1), H(1), L(1), C(1));
    int m0bull = bull(O(0), C(0)) && maru(O(0), H(0), L(0), C(0));
    int m0bear = bear(O(0), C(0)) && maru(O(0), H(0), L(0), C(0));
    int b = m1bear && m0bull && (O(0) > O(1));
    int s = m1bull && m0bear && (O(0) < O(1));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :1183-1226 */
static void cdlkickingbylength(ARGS) { UNUSED; LOOP(1) {
    int m1bear = bear(O(1), C(1)) && maru(O(1), H(1), L(1), C(1));
    int m1bull = bull(O(1), C(1)) && maru(O(1), H(1), L(1), C(1));
    int m0bull = bull(O(0), C(0)) && maru(O(0), H(0), L(0), C(0));
    int m0bear = bear(O(0), C(0)) && maru(O(0), H(0), L(0), C(0));
    double ba1 = body_abs(O(1), C(1)), ba0 = body_abs(O(0), C(0));
    int bk = m1bear && m0bull && (O(0) > O(1));
    int sk = m1bull && m0bear && (O(0) < O(1));
    int bl = bk && (ba0 >= ba1), sl = sk && (ba0 >= ba1);
    if (bl || (bk && !sl)) out[i] = 100; else if (sl || (sk && !bl)) out[i] = -100; } }
/* :1229-1264 (o1..o4 = rows i-4..i-1) */
static void cdlladderbottom(ARGS) { UNUSED; LOOP(4) {
    int m = bear(O(4), C(4)) && long_body(O(4), C(4)) && bear(O(3), C(3)) && (C(3) < C(4))
            && bear(O(2), C(2)) && (C(2) < C(3)) && bear(O(1), C(1)) && long_up_shadow(O(1), H(1), C(1))
            && bull(O(0), C(0)) && (O(0) > O(1));
    if (m) out[i] = 100; } }
/* :1267-1289 */
static void cdllongleggeddoji(ARGS) { UNUSED; LOOP(0) {
    if (doji(O(0), H(0), L(0), C(0)) && long_up_shadow(O(0), H(0), C(0)) && long_dn_shadow(O(0), L(0), C(0))) out[i] = 100; } }
/* :1292-1318 */
static void cdllongline(ARGS) { UNUSED; LOOP(0) {
    int m = long_body(O(0), C(0)) && short_up_shadow(O(0), H(0), L(0), C(0)) && short_dn_shadow(O(0), H(0), L(0), C(0));
    if (m && bull(O(0), C(0))) out[i] = 100; else if (m && bear(O(0), C(0))) out[i] = -100; } }
/* :1321-1346 */
static void cdlmarubozu(ARGS) { UNUSED; LOOP(0) {
    int m = maru(O(0), H(0), L(0), C(0));
    if (m && bull(O(0), C(0))) out[i] = 100; else if (m && bear(O(0), C(0))) out[i] = -100; } }
/* :1349-1373 */
static void cdlmatchinglow(ARGS) { UNUSED; LOOP(1) {
    int m = bear(O(1), C(1)) && long_body(O(1), C(1)) && bear(O(0), C(0)) && equal_(C(0), C(1), H(0), L(0));
    if (m) out[i] = 100; } }
/* :1376-1413 (penetration ignored by the Rust) */
static void cdlmathold(ARGS) { UNUSED; LOOP(4) {
    int m = bull(O(4), C(4)) && long_body(O(4), C(4)) && short_body(O(3), C(3)) && (O(3) > C(4))
            && short_body(O(2), C(2)) && short_body(O(1), C(1))
            && (L(3) > O(4)) && (L(2) > O(4)) && (L(1) > O(4)) && bull(O(0), C(0)) && (C(0) > C(4));
    if (m) out[i] = 100; } }
/* :1416-1451 */
static void cdlmorningdojistar(ARGS) { UNUSED; LOOP(2) {
    int m = bear(O(2), C(2)) && long_body(O(2), C(2)) && doji(O(1), H(1), L(1), C(1)) && (oc_max(O(1), C(1)) < C(2))
            && bull(O(0), C(0)) && (C(0) > (C(2) + (body_abs(O(2), C(2)) * pen)));
    if (m) out[i] = 100; } }
/* :1454-1487 */
static void cdlmorningstar(ARGS) { UNUSED; LOOP(2) {
    int m = bear(O(2), C(2)) && long_body(O(2), C(2)) && short_body(O(1), C(1)) && (oc_max(O(1), C(1)) < C(2))
            && bull(O(0), C(0)) && (C(0) > (C(2) + (body_abs(O(2), C(2)) * pen)));
    if (m) out[i] = 100; } }
/* :1490-1516 */
static void cdlonneck(ARGS) { UNUSED; LOOP(1) {
    int m = bear(O(1), C(1)) && long_body(O(1), C(1)) && bull(O(0), C(0)) && (O(0) < C(1)) && near_(C(0), L(1), H(0), L(0));
    if (m) out[i] = -100; } }
/* :1519-1550 */
static void cdlpiercing(ARGS) { UNUSED; LOOP(1) {
    int m = bear(O(1), C(1)) && long_body(O(1), C(1)) && bull(O(0), C(0)) && (O(0) < C(1))
            && (C(0) > (C(1) + (body_abs(O(1), C(1)) * pen))) && (C(0) < O(1));
    if (m) out[i] = 100; } }
/* :1553-1578 */
static void cdlrickshawman(ARGS) { UNUSED; LOOP(0) {
    double us = upper_shadow(O(0), H(0), C(0)), ls = lower_shadow(O(0), L(0), C(0));
    int m = doji(O(0), H(0), L(0), C(0)) && long_up_shadow(O(0), H(0), C(0)) && long_dn_shadow(O(0), L(0), C(0)) && near_(us, ls, H(0), L(0));
    if (m) out[i] = 100; } }
/* :1581-1644 */
static void cdlrisefall3methods(ARGS) { UNUSED; LOOP(4) {
    int mid = short_body(O(3), C(3)) && short_body(O(2), C(2)) && short_body(O(1), C(1));
    int hin = (H(3) < H(4)) && (H(2) < H(4)) && (H(1) < H(4));
    int lin = (L(3) > L(4)) && (L(2) > L(4)) && (L(1) > L(4));
    int r = bull(O(4), C(4)) && long_body(O(4), C(4)) && mid && hin && lin && bull(O(0), C(0)) && long_body(O(0), C(0)) && (C(0) > C(4));
    int f = bear(O(4), C(4)) && long_body(O(4), C(4)) && mid && lin && hin && bear(O(0), C(0)) && long_body(O(0), C(0)) && (C(0) < C(4));
    if (r) out[i] = 100; else if (f) out[i] = -100; } }
/* :1647-1676 */
static void cdlseparatinglines(ARGS) { UNUSED; LOOP(1) {
    int b = bear(O(1), C(1)) && long_body(O(1), C(1)) && bull(O(0), C(0)) && long_body(O(0), C(0)) && equal_(O(0), O(1), H(0), L(0));
    int s = bull(O(1), C(1)) && long_body(O(1), C(1)) && bear(O(0), C(0)) && long_body(O(0), C(0)) && equal_(O(0), O(1), H(0), L(0));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :1679-1706 */
static void cdlshootingstar(ARGS) { UNUSED; LOOP(1) {
    double ba = body_abs(O(0), C(0)), us = upper_shadow(O(0), H(0), C(0));
    int m = short_body(O(0), C(0)) && (us > (2.0 * ba)) && vshort_dn_shadow(O(0), H(0), L(0), C(0));
    if (m && bull(O(1), C(1))) out[i] = -100; } }
/* :1709-1735 */
static void cdlshortline(ARGS) { UNUSED; LOOP(0) {
    int m = short_body(O(0), C(0)) && short_up_shadow(O(0), H(0), L(0), C(0)) && short_dn_shadow(O(0), H(0), L(0), C(0));
    if (m && bull(O(0), C(0))) out[i] = 100; else if (m && bear(O(0), C(0))) out[i] = -100; } }
/* :1738-1763 */
static void cdlspinningtop(ARGS) { UNUSED; LOOP(0) {
    int m = short_body(O(0), C(0)) && (upper_shadow(O(0), H(0), C(0)) > body_abs(O(0), C(0)))
            && (lower_shadow(O(0), L(0), C(0)) > body_abs(O(0), C(0)));
    if (m && bull(O(0), C(0))) out[i] = 100; else if (m && bear(O(0), C(0))) out[i] = -100; } }
/* :1766-1794 */
static void cdlstalledpattern(ARGS) { UNUSED; LOOP(2) {
    int m = bull(O(2), C(2)) && long_body(O(2), C(2)) && bull(O(1), C(1)) && long_body(O(1), C(1)) && (C(1) > C(2))
            && bull(O(0), C(0)) && short_body(O(0), C(0)) && (C(0) > C(1)) && (O(0) > O(1)) && (O(0) <= C(1));
    if (m) out[i] = -100; } }
/* :1797-1828 */
static void cdlsticksandwich(ARGS) { UNUSED; LOOP(2) {
    int m = bear(O(2), C(2)) && long_body(O(2), C(2)) && bull(O(1), C(1)) && long_body(O(1), C(1)) && (O(1) > C(2))
            && bear(O(0), C(0)) && long_body(O(0), C(0)) && equal_(C(0), C(2), H(0), L(0));
    if (m) out[i] = 100; } }
/* :1831-1853 */
static void cdltakuri(ARGS) { UNUSED; LOOP(0) {
    if (doji(O(0), H(0), L(0), C(0)) && vlong_dn_shadow(O(0), L(0), C(0)) && vshort_up_shadow(O(0), H(0), L(0), C(0))) out[i] = 100; } }
/* :1856-1891 */
static void cdltasukigap(ARGS) { UNUSED; LOOP(2) {
    int b = bull(O(2), C(2)) && bull(O(1), C(1)) && (O(1) > C(2)) && bear(O(0), C(0)) && (O(0) > O(1)) && (O(0) < C(1))
            && (C(0) > O(2)) && (C(0) < C(2));
    int s = bear(O(2), C(2)) && bear(O(1), C(1)) && (O(1) < C(2)) && bull(O(0), C(0)) && (O(0) < O(1)) && (O(0) > C(1))
            && (C(0) < O(2)) && (C(0) > C(2));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :1894-1919 (penetration ignored; midpoint = c1 + body*0.5, :1910) */
static void cdlthrusting(ARGS) { UNUSED; LOOP(1) {
    double midpoint = C(1) + (body_abs(O(1), C(1)) * 0.5);
    int m = bear(O(1), C(1)) && long_body(O(1), C(1)) && bull(O(0), C(0)) && (O(0) < C(1)) && (C(0) > C(1)) && (C(0) < midpoint);
    if (m) out[i] = -100; } }
/* :1922-1961 */
static void cdltristar(ARGS) { UNUSED; LOOP(2) {
    int d = doji(O(2), H(2), L(2), C(2)) && doji(O(1), H(1), L(1), C(1)) && doji(O(0), H(0), L(0), C(0));
    double mid1 = (O(2) + C(2)) / 2.0, mid2 = (O(1) + C(1)) / 2.0, mid3 = (O(0) + C(0)) / 2.0;
    int b = d && (mid2 < mid1) && (mid3 > mid2);
    int s = d && (mid2 > mid1) && (mid3 < mid2);
    if (b) out[i] = 100; else if (s) out[i] = -100; } }
/* :1964-1994 */
static void cdlunique3river(ARGS) { UNUSED; LOOP(2) {
    int m = bear(O(2), C(2)) && long_body(O(2), C(2)) && bear(O(1), C(1)) && (L(1) < L(2)) && (C(1) > L(1))
            && (O(1) < O(2)) && (O(1) > C(2)) && bull(O(0), C(0)) && short_body(O(0), C(0)) && (C(0) < C(1));
    if (m) out[i] = 100; } }
/* :1997-2024 */
static void cdlupsidegap2crows(ARGS) { UNUSED; LOOP(2) {
    int m = bull(O(2), C(2)) && long_body(O(2), C(2)) && bear(O(1), C(1)) && (O(1) > C(2)) && (C(1) > C(2))
            && bear(O(0), C(0)) && (O(0) > O(1)) && (C(0) > C(2)) && (C(0) < C(1));
    if (m) out[i] = -100; } }
/* :2027-2062 */
static void cdlxsidegap3methods(ARGS) { UNUSED; LOOP(2) {
    int b = bull(O(2), C(2)) && bull(O(1), C(1)) && (O(1) > C(2)) && bear(O(0), C(0)) && (O(0) < C(1)) && (O(0) > O(1))
            && (C(0) > O(2)) && (C(0) < C(2));
    int s = bear(O(2), C(2)) && bear(O(1), C(1)) && (O(1) < C(2)) && bull(O(0), C(0)) && (O(0) > C(1)) && (O(0) < O(1))
            && (C(0) < O(2)) && (C(0) > C(2));
    if (b) out[i] = 100; else if (s) out[i] = -100; } }

typedef void (*pat_fn)(ARGS);
static const pat_fn pat_fns[PQO_N_PATTERNS] = {
    cdl2crows, cdl3blackcrows, cdl3inside, cdl3linestrike, cdl3outside, cdl3starsinsouth, cdl3whitesoldiers,
    cdlabandonedbaby, cdladvanceblock, cdlbelthold, cdlbreakaway, cdlclosingmarubozu, cdlconcealbabyswall,
    cdlcounterattack, cdldarkcloudcover, cdldoji, cdldojistar, cdldragonflydoji, cdlengulfing, cdleveningdojistar,
    cdleveningstar, cdlgapsidesidewhite, cdlgravestonedoji, cdlhammer, cdlhangingman, cdlharami, cdlharamicross,
    cdlhighwave, cdlhikkake, cdlhikkakemod, cdlhomingpigeon, cdlidentical3crows, cdlinneck, cdlinvertedhammer,
    cdlkicking, cdlkickingbylength, cdlladderbottom, cdllongleggeddoji, cdllongline, cdlmarubozu, cdlmatchinglow,
    cdlmathold, cdlmorningdojistar, cdlmorningstar, cdlonneck, cdlpiercing, cdlrickshawman, cdlrisefall3methods,
    cdlseparatinglines, cdlshootingstar, cdlshortline, cdlspinningtop, cdlstalledpattern, cdlsticksandwich,
    cdltakuri, cdltasukigap, cdlthrusting, cdltristar, cdlunique3river, cdlupsidegap2crows, cdlxsidegap3methods};
const char *const pqo_pattern_names[PQO_N_PATTERNS] = {
    "cdl2crows", "cdl3blackcrows", "cdl3inside", "cdl3linestrike", "cdl3outside", "cdl3starsinsouth",
    "cdl3whitesoldiers", "cdlabandonedbaby", "cdladvanceblock", "cdlbelthold", "cdlbreakaway",
    "cdlclosingmarubozu", "cdlconcealbabyswall", "cdlcounterattack", "cdldarkcloudcover", "cdldoji",
    "cdldojistar", "cdldragonflydoji", "cdlengulfing", "cdleveningdojistar", "cdleveningstar",
    "cdlgapsidesidewhite", "cdlgravestonedoji", "cdlhammer", "cdlhangingman", "cdlharami", "cdlharamicross",
    "cdlhighwave", "cdlhikkake", "cdlhikkakemod", "cdlhomingpigeon", "cdlidentical3crows", "cdlinneck",
    "cdlinvertedhammer", "cdlkicking", "cdlkickingbylength", "cdlladderbottom", "cdllongleggeddoji",
    "cdllongline", "cdlmarubozu", "cdlmatchinglow", "cdlmathold", "cdlmorningdojistar", "cdlmorningstar",
    "cdlonneck", "cdlpiercing", "cdlrickshawman", "cdlrisefall3methods", "cdlseparatinglines",
    "cdlshootingstar", "cdlshortline", "cdlspinningtop", "cdlstalledpattern", "cdlsticksandwich", "cdltakuri",
    "cdltasukigap", "cdlthrusting", "cdltristar", "cdlunique3river", "cdlupsidegap2crows",
    "cdlxsidegap3methods"};

void pqo_pattern(int id, const double *o, const double *h, const double *l, const double *c,
                 int64_t n, double penetration, int32_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = 0;
    if (id < 0 || id >= PQO_N_PATTERNS) return;
    pat_fns[id](o, h, l, c, n, penetration, out);
}
