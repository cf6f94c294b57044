// pattern.hip -- the 61 candlestick recognisers of src/talib/pattern.rs as ONE pass over OHLC.
// ROW shape: one (series, row) per thread; each thread builds the 5-candle window ending at its row,
// derives the per-candle predicates (pattern.rs:2067-2143) once, and evaluates every requested
// recogniser from them.  Int32 in {-100, 0, +100}; rows below a recogniser's look-back are 0.
// Reading OHLC once for all 61 outputs is the fused form SURVEY 8(a) prices at 276 B/row instead
// of 61 x 36 B/row.
#include "pq_dev.h"

struct Cdl { // one candle and its predicates; k = rows back from the current row
    double o, h, l, c;
    double body, us, ls, mx, mn;
    bool bull, bear, lng, sht, doji, lus, lds, sus, sds, vsus, vsds, vlds;
    __device__ void set(double o_, double h_, double l_, double c_) {
        o = o_; h = h_; l = l_; c = c_;
        bull = c > o;                                   // :2069
        bear = c < o;                                   // :2073
        body = fabs(o - c);                             // :2077
        mn = fmin(o, c);                                // :2081
        mx = fmax(o, c);                                // :2085
        us = h - mx;                                    // :2089
        ls = mn - l;                                    // :2093
        lng = body > 0.05 * (o + c) * 0.5;              // :2097
        sht = body < 0.1 * (o + c) * 0.5;               // :2101
        doji = body <= 0.005 * (o + c) * 0.5;           // :2105
        lus = us > 2.0 * body;                          // :2109
        lds = ls > 2.0 * body;                          // :2113
        sus = us < 0.5 * body;                          // :2117
        sds = ls < 0.5 * body;                          // :2121
        vsus = us < 0.1 * body;                         // :2125
        vsds = ls < 0.1 * body;                         // :2129
        vlds = ls > 3.0 * body;                         // :2133
    }
    __device__ bool maru() const { return lng && vsus && vsds; }
};
__device__ __forceinline__ bool nearq(double a, double b, const Cdl &k) { return fabs(a - b) < 0.01 * (k.h + k.l) * 0.5; }   // :2137
__device__ __forceinline__ bool equalq(double a, double b, const Cdl &k) { return fabs(a - b) < 0.001 * (k.h + k.l) * 0.5; } // :2141
__device__ __forceinline__ int pm(bool up, bool dn) { return up ? 100 : (dn ? -100 : 0); }

// a = current row, b .. e = 1 .. 4 rows back.  Returns the recogniser's value; caller masks t < lookback.
// The conditions are combined with the NON-short-circuit & and | on purpose: with `&` every term becomes a branch on the
// execution mask (a wave leaves a condition chain only when all 64 lanes fail it), and half of the kernel's instructions
// were scalar mask / branch instructions; with & / | each comparison is one v_cmp + one scalar AND and the recogniser is
// straight-line code.
__device__ __forceinline__ int cdl_eval(int id, const Cdl &a, const Cdl &b, const Cdl &c, const Cdl &d, const Cdl &e, double pen,
                                        int *lookback) {
    switch (id) {
    case 0: *lookback = 2; // cdl2crows :10-40
        return (c.bull & c.lng & b.bear & b.o > c.c & a.bear & a.o > b.o & a.o < b.c & a.c > c.o & a.c < c.c) ? -100 : 0;
    case 1: *lookback = 2; // cdl3blackcrows :43-73
        return (c.bear & c.lng & b.bear & b.lng & a.bear & a.lng & b.o < c.o & b.o > c.c & a.o < b.o & a.o > b.c &
                b.c < c.c & a.c < b.c) ? -100 : 0;
    case 2: *lookback = 2; // cdl3inside :76-111
        return pm(c.bear & c.lng & b.bull & b.c < c.o & b.o > c.c & a.bull & a.c > c.o,
                  c.bull & c.lng & b.bear & b.o < c.c & b.c > c.o & a.bear & a.c < c.o);
    case 3: *lookback = 3; // cdl3linestrike :114-157
        return pm(d.bear & c.bear & b.bear & c.c < d.c & b.c < c.c & c.o > d.c & c.o < d.o & b.o > c.c & b.o < c.o &
                      a.bull & a.o < b.c & a.c > d.o,
                  d.bull & c.bull & b.bull & c.c > d.c & b.c > c.c & c.o < d.c & c.o > d.o & b.o < c.c & b.o > c.o &
                      a.bear & a.o > b.c & a.c < d.o);
    case 4: *lookback = 2; // cdl3outside :160-191
        return pm(c.bear & b.bull & b.o <= c.c & b.c >= c.o & a.bull & a.c > b.c,
                  c.bull & b.bear & b.o >= c.c & b.c <= c.o & a.bear & a.c < b.c);
    case 5: *lookback = 2; // cdl3starsinsouth :194-231
        return (c.bear & c.lng & c.lds & b.bear & b.l > c.l & b.c > c.c & a.bear & a.sht & a.h < b.h & a.l > b.l) ? 100 : 0;
    case 6: *lookback = 2; // cdl3whitesoldiers :234-265
        return (c.bull & c.lng & b.bull & b.lng & a.bull & a.lng & b.o > c.o & b.o <= c.c & a.o > b.o & a.o <= b.c &
                b.c > c.c & a.c > b.c) ? 100 : 0;
    case 7: *lookback = 2; // cdlabandonedbaby :268-306
        return pm(c.bear & c.lng & b.doji & b.h < c.l & a.bull & a.l > b.h,
                  c.bull & c.lng & b.doji & b.l > c.h & a.bear & a.h < b.l);
    case 8: *lookback = 2; // cdladvanceblock :309-342
        return (c.bull & c.lng & b.bull & a.bull & b.o > c.o & b.o <= c.c & a.o > b.o & a.o <= b.c & b.c > c.c &
                a.c > b.c & a.body < b.body) ? -100 : 0;
    case 9: *lookback = 0; // cdlbelthold :345-370
        return pm(a.bull & a.lng & a.vsds, a.bear & a.lng & a.vsus);
    case 10: *lookback = 4; // cdlbreakaway :373-411
        return pm(e.bear & e.lng & d.bear & d.o < e.c & c.c < d.c & a.bull & a.c > d.o & a.c < e.c,
                  e.bull & e.lng & d.bull & d.o > e.c & c.c > d.c & a.bear & a.c < d.o & a.c > e.c);
    case 11: *lookback = 0; // cdlclosingmarubozu :414-439
        return pm(a.bull & a.lng & a.vsus, a.bear & a.lng & a.vsds);
    case 12: *lookback = 3; // cdlconcealbabyswall :442-484
        return (d.bear & d.lng & d.vsus & d.vsds & c.bear & c.lng & c.vsus & c.vsds & c.c < d.c & b.bear &
                b.h > c.c & a.bear & a.lng & a.o > b.h & a.c < c.l) ? 100 : 0;
    case 13: *lookback = 1; // cdlcounterattack :487-516
        return pm(b.bear & b.lng & a.bull & a.lng & nearq(a.c, b.c, a), b.bull & b.lng & a.bear & a.lng & nearq(a.c, b.c, a));
    case 14: *lookback = 1; // cdldarkcloudcover :519-550
        return (b.bull & b.lng & a.bear & a.o > b.c & a.c < (b.c - (b.body * pen)) & a.c > b.o) ? -100 : 0;
    case 15: *lookback = 0; // cdldoji :553-575
        return a.doji ? 100 : 0;
    case 16: { *lookback = 1; // cdldojistar :578-607
        double mid = (a.o + a.c) / 2.0;
        return pm(b.bear & b.lng & a.doji & mid < b.c, b.bull & b.lng & a.doji & mid > b.c); }
    case 17: *lookback = 0; // cdldragonflydoji :610-632
        return (a.doji & a.lds & a.vsus) ? 100 : 0;
    case 18: *lookback = 1; // cdlengulfing :635-662
        return pm(b.bear & a.bull & a.o <= b.c & a.c >= b.o & (a.o < b.c | a.c > b.o),
                  b.bull & a.bear & a.o >= b.c & a.c <= b.o & (a.o > b.c | a.c < b.o));
    case 19: *lookback = 2; // cdleveningdojistar :665-700
        return (c.bull & c.lng & b.doji & b.mn > c.c & a.bear & a.c < (c.c - (c.body * pen))) ? -100 : 0;
    case 20: *lookback = 2; // cdleveningstar :703-736
        return (c.bull & c.lng & b.sht & b.mn > c.c & a.bear & a.c < (c.c - (c.body * pen))) ? -100 : 0;
    case 21: { *lookback = 2; // cdlgapsidesidewhite :739-774
        bool common = b.bull & a.bull & nearq(a.body, b.body, a) & nearq(a.o, b.o, a);
        return pm(c.bull & b.o > c.c & common, c.bear & b.c < c.c & common); }
    case 22: *lookback = 0; // cdlgravestonedoji :777-799
        return (a.doji & a.lus & a.vsds) ? -100 : 0;
    case 23: *lookback = 1; // cdlhammer :802-829
        return (a.sht & a.ls > (2.0 * a.body) & a.vsus & b.bear) ? 100 : 0;
    case 24: *lookback = 1; // cdlhangingman :832-859
        return (a.sht & a.ls > (2.0 * a.body) & a.vsus & b.bull) ? -100 : 0;
    case 25: *lookback = 1; // cdlharami :862-893
        return pm(b.bear & b.lng & a.bull & a.sht & a.o > b.c & a.c < b.o, b.bull & b.lng & a.bear & a.sht & a.o < b.c & a.c > b.o);
    case 26: *lookback = 1; // cdlharamicross :896-926
        return pm(b.bear & b.lng & a.doji & a.mx < b.o & a.mn > b.c, b.bull & b.lng & a.doji & a.mx < b.c & a.mn > b.o);
    case 27: { *lookback = 0; // cdlhighwave :929-953
        bool m = a.sht & a.lus & a.lds;
        return pm(m & a.bull, m & a.bear); }
    case 28: { *lookback = 2; // cdlhikkake :956-984
        bool inside = b.h < c.h & b.l > c.l;
        return pm(inside & a.c > c.h & a.bull, inside & a.c < c.l & a.bear); }
    case 29: { *lookback = 3; // cdlhikkakemod :987-1018
        bool inside = c.h < d.h & c.l > d.l & b.h < c.h & b.l > c.l;
        return pm(inside & a.c > d.h & a.bull, inside & a.c < d.l & a.bear); }
    case 30: *lookback = 1; // cdlhomingpigeon :1021-1045
        return (b.bear & b.lng & a.bear & a.sht & a.o < b.o & a.c > b.c) ? 100 : 0;
    case 31: *lookback = 2; // cdlidentical3crows :1048-1080
        return (c.bear & c.lng & b.bear & b.lng & a.bear & a.lng & equalq(b.o, c.c, a) & equalq(a.o, b.c, a) &
                b.c < c.c & a.c < b.c) ? -100 : 0;
    case 32: *lookback = 1; // cdlinneck :1083-1108
        return (b.bear & b.lng & a.bull & a.o < b.c & nearq(a.c, b.c, a)) ? -100 : 0;
    case 33: *lookback = 1; // cdlinvertedhammer :1111-1138
        return (a.sht & a.us > (2.0 * a.body) & a.vsds & b.bear) ? 100 : 0;
    case 34: *lookback = 1; // cdlkicking :1141-1180
        return pm(b.bear & b.maru() & a.bull & a.maru() & a.o > b.o, b.bull & b.maru() & a.bear & a.maru() & a.o < b.o);
    case 35: { *lookback = 1; // cdlkickingbylength :1183-1226
        bool bk = b.bear & b.maru() & a.bull & a.maru() & a.o > b.o;
        bool sk = b.bull & b.maru() & a.bear & a.maru() & a.o < b.o;
        bool longer = a.body >= b.body;
        bool bl = bk & longer, sl = sk & longer;
        return pm(bl | (bk & !sl), sl | (sk & !bl)); }
    case 36: *lookback = 4; // cdlladderbottom :1229-1264
        return (e.bear & e.lng & d.bear & d.c < e.c & c.bear & c.c < d.c & b.bear & b.lus & a.bull & a.o > b.o) ? 100 : 0;
    case 37: *lookback = 0; // cdllongleggeddoji :1267-1289
        return (a.doji & a.lus & a.lds) ? 100 : 0;
    case 38: { *lookback = 0; // cdllongline :1292-1318
        bool m = a.lng & a.sus & a.sds;
        return pm(m & a.bull, m & a.bear); }
    case 39: { *lookback = 0; // cdlmarubozu :1321-1346
        bool m = a.maru();
        return pm(m & a.bull, m & a.bear); }
    case 40: *lookback = 1; // cdlmatchinglow :1349-1373
        return (b.bear & b.lng & a.bear & equalq(a.c, b.c, a)) ? 100 : 0;
    case 41: *lookback = 4; // cdlmathold :1376-1413
        return (e.bull & e.lng & d.sht & d.o > e.c & c.sht & b.sht & d.l > e.o & c.l > e.o & b.l > e.o & a.bull &
                a.c > e.c) ? 100 : 0;
    case 42: *lookback = 2; // cdlmorningdojistar :1416-1451
        return (c.bear & c.lng & b.doji & b.mx < c.c & a.bull & a.c > (c.c + (c.body * pen))) ? 100 : 0;
    case 43: *lookback = 2; // cdlmorningstar :1454-1487
        return (c.bear & c.lng & b.sht & b.mx < c.c & a.bull & a.c > (c.c + (c.body * pen))) ? 100 : 0;
    case 44: *lookback = 1; // cdlonneck :1490-1516
        return (b.bear & b.lng & a.bull & a.o < b.c & nearq(a.c, b.l, a)) ? -100 : 0;
    case 45: *lookback = 1; // cdlpiercing :1519-1550
        return (b.bear & b.lng & a.bull & a.o < b.c & a.c > (b.c + (b.body * pen)) & a.c < b.o) ? 100 : 0;
    case 46: *lookback = 0; // cdlrickshawman :1553-1578
        return (a.doji & a.lus & a.lds & nearq(a.us, a.ls, a)) ? 100 : 0;
    case 47: { *lookback = 4; // cdlrisefall3methods :1581-1644
        bool mid = d.sht & c.sht & b.sht & d.h < e.h & c.h < e.h & b.h < e.h & d.l > e.l & c.l > e.l & b.l > e.l;
        return pm(e.bull & e.lng & mid & a.bull & a.lng & a.c > e.c, e.bear & e.lng & mid & a.bear & a.lng & a.c < e.c); }
    case 48: *lookback = 1; // cdlseparatinglines :1647-1676
        return pm(b.bear & b.lng & a.bull & a.lng & equalq(a.o, b.o, a), b.bull & b.lng & a.bear & a.lng & equalq(a.o, b.o, a));
    case 49: *lookback = 1; // cdlshootingstar :1679-1706
        return (a.sht & a.us > (2.0 * a.body) & a.vsds & b.bull) ? -100 : 0;
    case 50: { *lookback = 0; // cdlshortline :1709-1735
        bool m = a.sht & a.sus & a.sds;
        return pm(m & a.bull, m & a.bear); }
    case 51: { *lookback = 0; // cdlspinningtop :1738-1763
        bool m = a.sht & a.us > a.body & a.ls > a.body;
        return pm(m & a.bull, m & a.bear); }
    case 52: *lookback = 2; // cdlstalledpattern :1766-1794
        return (c.bull & c.lng & b.bull & b.lng & b.c > c.c & a.bull & a.sht & a.c > b.c & a.o > b.o & a.o <= b.c) ? -100 : 0;
    case 53: *lookback = 2; // cdlsticksandwich :1797-1828
        return (c.bear & c.lng & b.bull & b.lng & b.o > c.c & a.bear & a.lng & equalq(a.c, c.c, a)) ? 100 : 0;
    case 54: *lookback = 0; // cdltakuri :1831-1853
        return (a.doji & a.vlds & a.vsus) ? 100 : 0;
    case 55: *lookback = 2; // cdltasukigap :1856-1891
        return pm(c.bull & b.bull & b.o > c.c & a.bear & a.o > b.o & a.o < b.c & a.c > c.o & a.c < c.c,
                  c.bear & b.bear & b.o < c.c & a.bull & a.o < b.o & a.o > b.c & a.c < c.o & a.c > c.c);
    case 56: { *lookback = 1; // cdlthrusting :1894-1919
        double midpoint = b.c + (b.body * 0.5);
        return (b.bear & b.lng & a.bull & a.o < b.c & a.c > b.c & a.c < midpoint) ? -100 : 0; }
    case 57: { *lookback = 2; // cdltristar :1922-1961
        bool dj = c.doji & b.doji & a.doji;
        double m1 = (c.o + c.c) / 2.0, m2 = (b.o + b.c) / 2.0, m3 = (a.o + a.c) / 2.0;
        return pm(dj & m2 < m1 & m3 > m2, dj & m2 > m1 & m3 < m2); }
    case 58: *lookback = 2; // cdlunique3river :1964-1994
        return (c.bear & c.lng & b.bear & b.l < c.l & b.c > b.l & b.o < c.o & b.o > c.c & a.bull & a.sht & a.c < b.c) ? 100 : 0;
    case 59: *lookback = 2; // cdlupsidegap2crows :1997-2024
        return (c.bull & c.lng & b.bear & b.o > c.c & b.c > c.c & a.bear & a.o > b.o & a.c > c.c & a.c < b.c) ? -100 : 0;
    case 60: *lookback = 2; // cdlxsidegap3methods :2027-2062
        return pm(c.bull & b.bull & b.o > c.c & a.bear & a.o < b.c & a.o > b.o & a.c > c.o & a.c < c.c,
                  c.bear & b.bear & b.o < c.c & a.bull & a.o > b.c & a.o < b.o & a.c < c.o & a.c > c.c);
    }
    *lookback = 0;
    return 0;
}

struct CdlArgs {
    const double *o, *h, *l, *c;
    int32_t *out[PQ_N_PATTERNS];
    double pen[PQ_N_PATTERNS];
};

__device__ __forceinline__ Cdl load_candle(const CdlArgs &a, int64_t base, int64_t q) {
    Cdl k;
    if (q >= 0) k.set(a.o[base + q], a.h[base + q], a.l[base + q], a.c[base + q]);
    else k.set(0.0, 0.0, 0.0, 0.0);
    return k;
}

// R consecutive rows per thread: the R + 4 candles they look at are loaded and classified once (instead of 5 per row), and
// each recogniser's R results leave as one 4R-byte store.  `vec`: the int32 rows are 4R-byte aligned (stride % R == 0).
// Measured at 5000 x 2520 (solo / suite step, one session): R = 1 0.87 / 5.34 ms, R = 2 1.07 / 5.42, R = 4 2.41 / 6.28 (the
// classified candles of more rows cost more registers than the shared loads save) -- 0.87 ms is 3.75 GB at 4.3 TB/s, the
// non-temporal store ceiling.  Before the recognisers were made branch-free (see cdl_eval) the kernel took 1.4 ms solo.
#ifndef PQ_CDL_R
#define PQ_CDL_R 1
#endif
constexpr int CDL_R = PQ_CDL_R;
// ALL: every recogniser has an output column (the usual call): no per-recogniser test of its pointer -- 61 uniform branches per row,
// each the end of a scheduling region -- and one store form (VEC: the int32 rows are 4R-byte aligned).
template <bool ALL, bool VEC>
__global__ __launch_bounds__(ROW_BLOCK) void cdl_all_kernel(CdlArgs a, Dims d) {
    const int64_t s = blockIdx.y;
    const int64_t t0 = ((int64_t)blockIdx.x * ROW_BLOCK + threadIdx.x) * CDL_R;
    const int64_t slen = dims_len(d, s);
    if (t0 >= slen) return;
    const int64_t base = dims_base(d, s);
    // Issue priority beside the SEQ grids of a suite (whose long jobs raise their own).  Round 2: 1 (the kernel was only issued in the gaps
    // and became the critical path at 0; 3 cost +4 %).  Round 3: 0 -- its waves now sit beside two job waves on every SIMD (192-VGPR job
    // kernel) and advance all the time; at 0 they take what the job waves leave: 4.06 against 4.12 ms per step (2: 4.12, 3: 4.19).
#ifndef PQ_CDL_PRIO
#define PQ_CDL_PRIO 0
#endif
    __builtin_amdgcn_s_setprio(PQ_CDL_PRIO);
    Cdl w[CDL_R + 4]; // w[k] = candle of row t0 + CDL_R - 1 - k
#pragma unroll
    for (int k = 0; k < CDL_R + 4; k++) {
        const int64_t q = t0 + CDL_R - 1 - k;
        w[k] = load_candle(a, base, q < slen ? q : slen - 1); // rows past the end (ragged last thread) are never stored
    }
#pragma unroll
    for (int id = 0; id < PQ_N_PATTERNS; id++) {
        if (!ALL && a.out[id] == nullptr) continue; // wave-uniform
        int v[CDL_R];
#pragma unroll
        for (int r = 0; r < CDL_R; r++) { // row t0 + r: current candle w[CDL_R - 1 - r]
            int lb;
            const int x = cdl_eval(id, w[CDL_R - 1 - r], w[CDL_R - r], w[CDL_R + 1 - r], w[CDL_R + 2 - r], w[CDL_R + 3 - r], a.pen[id], &lb);
            v[r] = (t0 + r >= lb) ? x : 0;
        }
#if PQ_EXP_NOSTORE_ON
        if (v[0] == 123456789) a.out[id][base + t0] = v[0];
#else
        int32_t *dst = &a.out[id][base + t0];
        if (VEC && (CDL_R == 1 || t0 + CDL_R <= slen)) {
            typedef int pq_irv __attribute__((ext_vector_type(CDL_R)));
            pq_irv vv;
#pragma unroll
            for (int r = 0; r < CDL_R; r++) vv[r] = v[r];
            __builtin_nontemporal_store(vv, reinterpret_cast<pq_irv *>(dst));
        } else {
#pragma unroll
            for (int r = 0; r < CDL_R; r++)
                if (t0 + r < slen) __builtin_nontemporal_store(v[r], dst + r);
        }
#endif
    }
}

__global__ __launch_bounds__(ROW_BLOCK) void cdl_one_kernel(CdlArgs a, int id, Dims d) {
    const int64_t s = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * ROW_BLOCK + threadIdx.x;
    if (t >= dims_len(d, s)) return;
    const int64_t base = dims_base(d, s);
    Cdl w[5];
#pragma unroll
    for (int k = 0; k < 5; k++) w[k] = load_candle(a, base, t - k);
    int lb;
    int v = cdl_eval(id, w[0], w[1], w[2], w[3], w[4], a.pen[0], &lb);
    a.out[0][base + t] = (t >= lb) ? v : 0;
}

static const char *const k_pattern_names[PQ_N_PATTERNS] = {
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

#include <string.h>
extern "C" {

const char *pq_pattern_name(int32_t id) { return (id >= 0 && id < PQ_N_PATTERNS) ? k_pattern_names[id] : nullptr; }
int32_t pq_pattern_id(const char *name) {
    if (!name) return -1;
    for (int i = 0; i < PQ_N_PATTERNS; i++)
        if (strcmp(name, k_pattern_names[i]) == 0) return i;
    return -1;
}

struct CdlBlob {
    CdlArgs args;
    int id;
    pq_batch b;
};
static void cdl_launch_blob(const void *blob, hipStream_t stream);
static pq_status cdl_launch(pq_ctx *ctx, const pq_batch *b, const CdlArgs &args, int id) {
    if (b->n_series == 0 || b->len == 0) return PQ_OK;
    CdlBlob cb{args, id, *b};
    if (ctx->rec) {
        static_assert(sizeof(CdlBlob) <= sizeof(RowThunk::blob), "CdlBlob too large");
        RowThunk t;
        t.launch = &cdl_launch_blob;
        t.row_id = 0;
        t.blob_bytes = (int)sizeof cb;
        t.dims = dims_of(b);
        memcpy(t.blob, &cb, sizeof cb);
        t.n_reads = 4;
        t.reads[0] = args.o; t.reads[1] = args.h; t.reads[2] = args.l; t.reads[3] = args.c;
        t.n_writes = 0;
        for (int i = 0; i < PQ_N_PATTERNS; i++) if (args.out[i]) t.writes[t.n_writes++] = args.out[i];
        return rec_add_row(ctx, t);
    }
    cdl_launch_blob(&cb, ctx->stream);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}
static void cdl_launch_blob(const void *blob, hipStream_t stream) {
    const CdlBlob &cb = *reinterpret_cast<const CdlBlob *>(blob);
    const pq_batch *b = &cb.b;
    const CdlArgs &args = cb.args;
    const int id = cb.id;
    for (int64_t s0 = 0; s0 < b->n_series; s0 += 65535) {
        int64_t ns = b->n_series - s0 < 65535 ? b->n_series - s0 : 65535;
        CdlArgs a2 = args;
        int64_t off = b->offsets ? 0 : s0 * b->stride; // (a ragged slice keeps the column pointers: its offsets are absolute rows)
        a2.o += off; a2.h += off; a2.l += off; a2.c += off;
        for (int i = 0; i < PQ_N_PATTERNS; i++) if (a2.out[i]) a2.out[i] += off;
        dim3 grid((unsigned)((b->len + ROW_BLOCK - 1) / ROW_BLOCK), (unsigned)ns);
        Dims d{ns, b->len, b->stride, b->offsets ? b->offsets + s0 : nullptr};
        if (id < 0) {
            int vec = (b->stride % CDL_R) == 0 && !b->offsets;
            for (int i = 0; i < PQ_N_PATTERNS; i++)
                if (a2.out[i] && reinterpret_cast<uintptr_t>(a2.out[i]) % (4 * CDL_R)) vec = 0;
            const int64_t per_block = (int64_t)ROW_BLOCK * CDL_R;
            dim3 grid2((unsigned)((b->len + per_block - 1) / per_block), (unsigned)ns);
            bool all = true;
            for (int i = 0; i < PQ_N_PATTERNS; i++) all &= a2.out[i] != nullptr;
            if (all && vec) hipLaunchKernelGGL((cdl_all_kernel<true, true>), grid2, dim3(ROW_BLOCK), 0, stream, a2, d);
            else if (all) hipLaunchKernelGGL((cdl_all_kernel<true, false>), grid2, dim3(ROW_BLOCK), 0, stream, a2, d);
            else if (vec) hipLaunchKernelGGL((cdl_all_kernel<false, true>), grid2, dim3(ROW_BLOCK), 0, stream, a2, d);
            else hipLaunchKernelGGL((cdl_all_kernel<false, false>), grid2, dim3(ROW_BLOCK), 0, stream, a2, d);
        } else hipLaunchKernelGGL(cdl_one_kernel, grid, dim3(ROW_BLOCK), 0, stream, a2, id, d);
    }
}

pq_status pq_cdl(pq_ctx *ctx, const pq_batch *b, int32_t id, const double *o, const double *h, const double *l,
                 const double *c, double penetration, int32_t *out) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(o && h && l && c && out, "pq_cdl: null pointer");
    PQ_REQUIRE(id >= 0 && id < PQ_N_PATTERNS, "pq_cdl: pattern id out of range");
    CdlArgs a;
    memset(&a, 0, sizeof a);
    a.o = o; a.h = h; a.l = l; a.c = c;
    a.out[0] = out; a.pen[0] = penetration;
    return cdl_launch(ctx, b, a, id);
}

pq_status pq_cdl_all(pq_ctx *ctx, const pq_batch *b, const double *o, const double *h, const double *l,
                     const double *c, const double *penetrations, int32_t *const *outs) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(o && h && l && c && penetrations && outs, "pq_cdl_all: null pointer");
    CdlArgs a;
    memset(&a, 0, sizeof a);
    a.o = o; a.h = h; a.l = l; a.c = c;
    for (int i = 0; i < PQ_N_PATTERNS; i++) { a.out[i] = outs[i]; a.pen[i] = penetrations[i]; }
    return cdl_launch(ctx, b, a, -1);
}

} // extern "C"
