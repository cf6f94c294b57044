// misc.hip -- kernels + C ABI for volatility / volume / price transforms and the Hilbert-transform
// cycle indicators (reference: src/talib/{volatility,volume,price,cycle}.rs) plus MAMA (D-4).
#pragma once
#include "pq_cores.h"

__device__ __forceinline__ double n0m(double x) { return pq_isnull(x) ? 0.0 : x; }

// ---------------------------------------------------------------- price.rs (N-C, ROW)
template <int KIND> // 0 avgprice(o,h,l,c) 1 medprice(h,l) 2 typprice(h,l,c) 3 wclprice(h,l,c)
struct PriceOp {
    static constexpr int NIN = (KIND == 0 ? 4 : (KIND == 1 ? 2 : 3)), NOUT = 1;
    static constexpr int ROW_ID = 1 + KIND;
    typedef double OutT;
    __device__ void eval(const Row<NIN> &r, int64_t t, double (&y)[1]) {
        double a[NIN];
        bool nul = false;
#pragma unroll
        for (int k = 0; k < NIN; k++) { a[k] = r.in[k][t]; nul |= pq_isnull(a[k]); }
        if (nul) { y[0] = pq_null(); return; }
        if constexpr (KIND == 0) y[0] = (a[0] + a[1] + a[2] + a[NIN - 1]) * 0.25; // price.rs:25
        else if constexpr (KIND == 1) y[0] = (a[0] + a[1]) * 0.5;                        // price.rs:44
        else if constexpr (KIND == 2) y[0] = (a[0] + a[1] + a[NIN - 1]) / 3.0; // price.rs:65
        else y[0] = (a[0] + a[1] + 2.0 * a[NIN - 1]) / 4.0;                   // price.rs:86
    }
};

// ---------------------------------------------------------------- volatility.rs
__device__ __forceinline__ double true_range(double h, double l, double pc) { // volatility.rs:77
    return fmax(fmax(h - l, fabs(h - pc)), fabs(l - pc));
}
struct TrangeOp { // volatility.rs:67-84 (ROW; row 0 null because pre_close = close.shift(1))
    static constexpr int NIN = 3, NOUT = 1;
    static constexpr int ROW_ID = 5;
    typedef double OutT;
    __device__ void eval(const Row<3> &r, int64_t t, double (&y)[1]) {
        y[0] = pq_null();
        if (t == 0) return;
        double h = r.in[0][t], l = r.in[1][t], pc = r.in[2][t - 1];
        if (pq_isnull(h) || pq_isnull(l) || pq_isnull(pc)) return;
        y[0] = true_range(h, l, pc);
    }
};
template <bool NATR> // volatility.rs:18-31 / :34-48: calc_ema(trange, 2p-1) [/ close * 100]
struct AtrOp {
    static constexpr int NIN = 3, NOUT = 1;
    static constexpr int SEQ_ID = 40 + (NATR ? 1 : 0);
    static constexpr int COST_NS = 200;
    int64_t p;
    EmaCore e;
    double pc;
    __device__ void init(const Row<3> &r) { e.init(2 * p - 1, r.len); pc = pq_null(); }
    __device__ void step(const Row<3> &, int64_t t, const double (&x)[3], double (&y)[1]) {
        double tr = pq_null();
        if (t > 0 && !pq_isnull(x[0]) && !pq_isnull(x[1]) && !pq_isnull(pc)) tr = true_range(x[0], x[1], pc);
        pc = x[2];
        double a = e.step(tr);
        if (NATR) y[0] = (pq_isnull(a) || pq_isnull(x[2])) ? pq_null() : a / x[2] * 100.0;
        else y[0] = a;
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return t0 > 0 && e.steady() && !pq_isnull(pc); }
    __device__ void step_fast(int64_t, const double (&x)[3], double (&y)[1]) {
        const double tr = true_range(x[0], x[1], pc);
        pc = x[2];
        const double a = e.fast(tr);
        if (NATR) y[0] = a / x[2] * 100.0;
        else y[0] = a;
    }
};

// ---------------------------------------------------------------- volume.rs
template <bool OSC> // volume.rs:100-126 calc_ad (quirk Q-AD); OSC: volume.rs:34-67 (quirk Q-ADOSC)
struct AdOp {
    static constexpr int NIN = 4, NOUT = 1; // high, low, close, volume
    static constexpr int SEQ_ID = 42 + (OSC ? 1 : 0);
    static constexpr int COST_NS = OSC ? 350 : 250;
    int64_t fast, slow;
    double sum, sum2;
    EmaCore ef, es;
    __device__ void init(const Row<4> &r) { sum = 0.0; sum2 = 0.0; if (OSC) { ef.init(fast, r.len); es.init(slow, r.len); } }
    __device__ void step(const Row<4> &, int64_t, const double (&x)[4], double (&y)[1]) {
        double ad;
        if (pq_isnull(x[0]) || pq_isnull(x[1]) || pq_isnull(x[2]) || pq_isnull(x[3])) ad = pq_null();
        else {
            double diff = x[0] - x[1];
            if (diff == 0.0) ad = 0.0;
            else { sum += (2.0 * x[2] - x[1] - x[0]) / diff * x[3]; ad = sum; }
        }
        if (!OSC) { y[0] = ad; return; }
        double adl = pq_null();
        if (!pq_isnull(ad)) { sum2 += ad; adl = sum2; }
        double f = ef.step(adl), s = es.step(adl);
        y[0] = (pq_isnull(f) || pq_isnull(s)) ? pq_null() : f - s;
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return !OSC || (ef.steady() && es.steady()); }
    __device__ void step_fast(int64_t, const double (&x)[4], double (&y)[1]) {
        const double diff = x[0] - x[1];
        const double add = sum + (2.0 * x[2] - x[1] - x[0]) / diff * x[3]; // unused (and possibly inf / nan) on flat bars
        sum = (diff == 0.0) ? sum : add;
        const double ad = (diff == 0.0) ? 0.0 : sum;
        if (!OSC) { y[0] = ad; return; }
        sum2 += ad;
        const double f = ef.fast(sum2), s = es.fast(sum2);
        y[0] = f - s;
    }
};
struct ObvOp { // volume.rs:70-94 (quirk Q-OBV: d = prev_close - close)
    static constexpr bool RG_GATHER = true; // a direct call on a ragged batch keeps the per-lane form: alone on the chip it beats re-housing + the tiled body (profiles/r05_bench_ragged.json)
    static constexpr int64_t DIRECT_LANE_MAX = 8192;
    static constexpr int NIN = 2, NOUT = 1; // close, volume
    static constexpr int SEQ_ID = 44;
    static constexpr int COST_NS = 120;
    double sum, pc;
    __device__ void init(const Row<2> &) { sum = 0.0; pc = pq_null(); }
    __device__ void step(const Row<2> &, int64_t t, const double (&x)[2], double (&y)[1]) {
        double prev = pc;
        pc = x[0];
        if (t == 0 || pq_isnull(x[0]) || pq_isnull(prev) || pq_isnull(x[1])) { y[0] = pq_null(); return; }
        double c_diff = prev - x[0];
        if (c_diff > 0.0) sum += x[1];
        else if (c_diff < 0.0) sum -= x[1];
        y[0] = sum;
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return t0 > 0 && !pq_isnull(pc); }
    __device__ void step_fast(int64_t, const double (&x)[2], double (&y)[1]) {
        const double c_diff = pc - x[0];
        pc = x[0];
        const double up = sum + x[1], dn = sum - x[1];
        sum = (c_diff > 0.0) ? up : ((c_diff < 0.0) ? dn : sum);
        y[0] = sum;
    }
};

// ---------------------------------------------------------------- cycle.rs
// Shared pipeline cycle.rs:27-63; MODE selects the emitted columns:
//   0 ht_dcperiod  1 ht_dcphase  2 ht_phasor(2)  3 ht_sine(2)  4 mama(2, decision D-4)
#define PQ_PI 3.14159265358979323846
#define PQ_TAU 6.28318530717958647692
// degrees per radian as ONE constant: `x * 180.0 / PI` costs a multiplication and a true division per row (no fast-math);
// the outputs that pass through it are transcendental-class (1e-12), one more rounding of ~1e-16 is immaterial
#define PQ_RAD2DEG (180.0 / PQ_PI)
template <int MODE>
struct HtOp {
    static constexpr bool RG_GATHER = true; // a direct call on a ragged batch keeps the per-lane form: alone on the chip it beats re-housing + the tiled body (profiles/r05_bench_ragged.json)
    static constexpr int64_t DIRECT_LANE_MAX = (int64_t)1 << 40; // a DIRECT call always runs the per-lane form (register delay lines: 1.15 against 1.57 ms for ht_dcperiod at any batch size, profiles/r05_direct_lane.json); the tiled form is the suite's (192-register cap)
    static constexpr int NIN = 1, NOUT = (MODE >= 2 ? 2 : 1);
    static constexpr int SEQ_ID = 45 + MODE;
    static constexpr int COST_NS = 880;
    double fastlimit, slowlimit; // MODE 4
    double rl[4];                // real[i], real[i-1], real[i-2], real[i-3]
    // The delay lines of the pipeline.  Gather body: four 7-deep shift registers (56 VGPRs, 24 register moves per row).  Tiled body:
    // three LDS rings ([slot][lane], zero-filled) -- smooth (6 deep), detrender (9 deep) and Q1 (6 deep); the I1 line is the
    // detrender line three rows later (i1[k] = detrend[k + 3]: both are pushed on the same rows, from zeros), so it needs no storage
    // of its own.  A tap `k rows back after this row's push` is read as `pushed k pushes ago` BEFORE the push, i.e. all thirteen taps
    // of a row are old values whose LDS reads are issued at the top of the row, off the dependency chain.  Same values, same order
    // of arithmetic; ~56 VGPRs less, which is what lets the Hilbert jobs run in the 192-VGPR light job kernel.
    double sm[7];                // smooth[i] ... smooth[i-6]
    double detrend[7], q1[7], i1[7];
    Ring r_sm, r_dt, r_q1;
    double i1_0, q1_0;           // this row's in-phase / quadrature components (either storage form)
    double i2, q2, re, im, period, smooth_period;
    double mama, fama, prev_phase;
    bool dead;
    __device__ static void push7(double (&dq)[7], double v) {
#pragma unroll
        for (int k = 6; k >= 1; k--) dq[k] = dq[k - 1];
        dq[0] = v;
    }
    __device__ void init_scalars(int64_t len) {
        dead = len < 32; // cycle.rs:16
#pragma unroll
        for (int k = 0; k < 4; k++) rl[k] = 0.0;
        i2 = q2 = re = im = period = smooth_period = 0.0;
        mama = fama = prev_phase = 0.0;
        i1_0 = q1_0 = 0.0;
    }
    __device__ void init(const Row<1> &r) {
        init_scalars(r.len);
#pragma unroll
        for (int k = 0; k < 7; k++) sm[k] = detrend[k] = q1[k] = i1[k] = 0.0;
    }
    __host__ __device__ int64_t ring_slots() const { return 6 + 9 + 6; }
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) {
        init_scalars(r.len);
        r_sm = ra.make(6); r_dt = ra.make(9); r_q1 = ra.make(6);
#pragma unroll
        for (int k = 0; k < 9; k++) { if (k < 6) { r_sm.base[k * 64] = 0.0; r_q1.base[k * 64] = 0.0; } r_dt.base[k * 64] = 0.0; }
    }
    static constexpr bool FAST_NULL_OK = true; // N-B (MODE 4: N-0, nulls become 0.0 in the row body itself)
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return !dead && t0 >= 32; } // every warm-up test of the row body is past
    __device__ void step_fast(int64_t i, const double (&x)[1], double (&y)[NOUT]) { row<true, true>(i, x, y); }   // tiled body only
    __device__ void step_lds(int64_t i, const double (&x)[1], double (&y)[NOUT]) { row<false, true>(i, x, y); }
    __device__ void step(const Row<1> &, int64_t i, const double (&x)[1], double (&y)[NOUT]) { row<false, false>(i, x, y); }
    template <bool FAST, bool RINGS> // FAST: i >= 32 and !dead are known
    __device__ __forceinline__ void row(int64_t i, const double (&x)[1], double (&y)[NOUT]) {
#pragma unroll
        for (int k = 0; k < NOUT; k++) y[k] = pq_null();
        if (!FAST && dead) return;
        double v = (MODE == 4) ? n0m(x[0]) : x[0];
        rl[3] = rl[2]; rl[2] = rl[1]; rl[1] = rl[0]; rl[0] = v;
        // cycle.rs:462-470 calc_smooth (0 for i < 3)
        double s = (FAST || i >= 3) ? (4.0 * rl[0] + 3.0 * rl[1] + 2.0 * rl[2] + rl[3]) * 0.1 : 0.0;
        double sm2, sm4, sm6;
        if constexpr (RINGS) { sm2 = r_sm.get(2); sm4 = r_sm.get(4); sm6 = r_sm.get(6); r_sm.push(s); }
        else { push7(sm, s); sm2 = sm[2]; sm4 = sm[4]; sm6 = sm[6]; }
        if (!FAST && i < 6) return;
        double d2, d4, d6, i1_2, i1_4, i1_6, q2t, q4t, q6t; // the taps of this row (after its pushes)
        if constexpr (RINGS) {
            d2 = r_dt.get(2); i1_0 = r_dt.get(3); d4 = r_dt.get(4); i1_2 = r_dt.get(5); d6 = r_dt.get(6); i1_4 = r_dt.get(7); i1_6 = r_dt.get(9);
            q2t = r_q1.get(2); q4t = r_q1.get(4); q6t = r_q1.get(6);
        }
        double prev_period = (FAST || i > 6) ? period : 6.0;
        double adj = 0.075 * prev_period + 0.54;
        double detrend_curr = (0.0962 * s + 0.5769 * sm2 - 0.5769 * sm4 - 0.0962 * sm6) * adj;
        if constexpr (RINGS) r_dt.push(detrend_curr);
        else { push7(detrend, detrend_curr); d2 = detrend[2]; d4 = detrend[4]; d6 = detrend[6]; }
        double q1_curr = (0.0962 * detrend_curr + 0.5769 * d2 - 0.5769 * d4 - 0.0962 * d6) * adj;
        if constexpr (RINGS) r_q1.push(q1_curr);
        else {
            push7(q1, q1_curr); q2t = q1[2]; q4t = q1[4]; q6t = q1[6];
            push7(i1, detrend[3]); i1_0 = i1[0]; i1_2 = i1[2]; i1_4 = i1[4]; i1_6 = i1[6];
        }
        q1_0 = q1_curr;
        double ji = (0.0962 * i1_0 + 0.5769 * i1_2 - 0.5769 * i1_4 - 0.0962 * i1_6) * adj;
        double jq = (0.0962 * q1_0 + 0.5769 * q2t - 0.5769 * q4t - 0.0962 * q6t) * adj;
        double i2_curr = 0.2 * (i1_0 - jq) + 0.8 * i2;
        double q2_curr = 0.2 * (q1_0 + ji) + 0.8 * q2;
        double re_curr = 0.2 * (i2_curr * i2 + q2_curr * q2) + 0.8 * re;
        double im_curr = 0.2 * (i2_curr * q2 - q2_curr * i2) + 0.8 * im;
        i2 = i2_curr; q2 = q2_curr; re = re_curr; im = im_curr;
        if (im != 0.0 && re != 0.0) period = PQ_TAU / atan(im / re);
        double lo = 0.67 * prev_period, hi = 1.5 * prev_period; // f64::clamp twice (cycle.rs:60-62)
        if (period < lo) period = lo;
        if (period > hi) period = hi;
        if (period < 6.0) period = 6.0;
        if (period > 50.0) period = 50.0;
        period = 0.2 * period + 0.8 * prev_period;
        if (MODE == 0) {
            smooth_period = 0.33 * period + 0.67 * smooth_period;
            if (FAST || i >= 31) y[0] = smooth_period;
        } else if (MODE == 1) {
            if (FAST || i >= 31) {
                double dc_phase = (i1_0 != 0.0) ? atan(q1_0 / i1_0) * PQ_RAD2DEG : 0.0;
                dc_phase += 90.0;
                if (i1_0 < 0.0) dc_phase += 180.0;
                if (dc_phase > 315.0) dc_phase -= 360.0;
                y[0] = dc_phase;
            }
        } else if (MODE == 2) {
            if (FAST || i >= 31) { y[0] = i1_0; y[NOUT - 1] = q1_0; }
        } else if (MODE == 3) {
            if (FAST || i >= 31) {
                double dc_phase = (i1_0 != 0.0) ? atan(q1_0 / i1_0) * PQ_RAD2DEG : 0.0;
                y[0] = sin(dc_phase * PQ_PI / 180.0);
                y[NOUT - 1] = sin((dc_phase + 45.0) * PQ_PI / 180.0);
            }
        } else {
            double phase = (i1_0 != 0.0) ? atan(q1_0 / i1_0) * PQ_RAD2DEG : 0.0;
            double dphase = prev_phase - phase;
            if (dphase < 1.0) dphase = 1.0;
            double alpha = fastlimit / dphase;
            if (alpha < slowlimit) alpha = slowlimit;
            if (alpha > fastlimit) alpha = fastlimit;
            mama = alpha * v + (1.0 - alpha) * mama;
            double ha = 0.5 * alpha;
            fama = ha * mama + (1.0 - ha) * fama;
            prev_phase = phase;
            if (FAST || i >= 31) { y[0] = mama; y[NOUT - 1] = fama; }
        }
    }
};
// cycle.rs:310-374 / :377-448: pure functions of real[i-3..i] (the pipeline result is unused)
struct TrendlineOp {
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int ROW_ID = 6;
    typedef double OutT;
    __device__ void eval(const Row<1> &r, int64_t i, double (&y)[1]) {
        y[0] = pq_null();
        if (r.len < 32 || i < 31) return;
        double tl = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) tl += r.in[0][i - j];
        y[0] = tl * 0.25;
    }
};
struct TrendmodeOp {
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int ROW_ID = 7;
    typedef int32_t OutT;
    __device__ void eval(const Row<1> &r, int64_t i, int32_t (&y)[1]) {
        y[0] = PQ_NULL_I32;
        if (r.len < 32 || i < 31) return;
        double tl = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) tl += r.in[0][i - j];
        tl *= 0.25;
        y[0] = (fabs(r.in[0][i] - tl) > 0.01 * tl) ? 1 : 0;
    }
};

