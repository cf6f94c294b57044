// overlap.hip -- SEQ kernels + C ABI for the overlap studies (reference: src/talib/overlap.rs).
// One series per lane, reference operation order, null-transparent streaming (N-A).
#pragma once
#include "pq_cores.h"

// ---------------------------------------------------------------- functors
struct SmaOp { // overlap.rs:871-937
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 1;
    static constexpr int COST_NS = 116;
    static constexpr int NTAP = 1;
    static constexpr int TAP_COL[1] = {0};
    int64_t p;
    SmaCore c;
    __device__ void init(const Row<1> &r) { c.init(p, r.len); }
    __device__ void tap_lags(int64_t (&lag)[1]) const { lag[0] = c.dead ? 0 : p; }
    __device__ void step(const Row<1> &r, int64_t t, const double (&x)[1], const double (&tp)[1], double (&y)[1]) {
        y[0] = c.step(r.in[0], t, x[0], tp[0]);
    }
    Ring w;
    __host__ __device__ int64_t ring_slots() const { return p > 0 ? p : 1; }
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { c.init(p, r.len); w = ra.make(p); }
    __device__ void step_lds(int64_t, const double (&x)[1], double (&y)[1]) { y[0] = c.step_ring(w, x[0]); }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return c.steady(); }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) { y[0] = c.fast_ring(w, x[0]); }
    static constexpr bool FAST_BATCH = true;
    static constexpr int FAST_UNROLL = 16;
    template <int N>
    __device__ void steps_fast(int64_t, const double (&x)[N][1], double (&y)[N][1]) {
        double v[N], o[N];
#pragma unroll
        for (int u = 0; u < N; u++) v[u] = x[u][0];
        c.fast_ring_n<N>(w, v, o);
#pragma unroll
        for (int u = 0; u < N; u++) y[u][0] = o[u];
    }
};

struct EmaOp { // overlap.rs:660-730
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 2;
    static constexpr int COST_NS = 100;
    int64_t p;
    EmaCore c;
    __device__ void init(const Row<1> &r) { c.init(p, r.len); }
    __device__ void step(const Row<1> &, int64_t, const double (&x)[1], double (&y)[1]) { y[0] = c.step(x[0]); }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return c.steady(); }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) { y[0] = c.fast(x[0]); }
    static constexpr int FAST_UNROLL = 16;
};

struct BbandsOp { // overlap.rs:47-116
    static constexpr int NIN = 1, NOUT = 3;
    static constexpr int SEQ_ID = 3;
    static constexpr int COST_NS = 263;
    static constexpr int NTAP = 1;
    static constexpr int TAP_COL[1] = {0};
    int64_t p;
    double up, dn;
    int64_t count;
    double sum, sum_sq;
    bool dead;
    Regular reg;
    ValidCursor tail;
    __device__ void init(const Row<1> &r) {
        dead = (p <= 0 || r.len < p);
        count = 0; sum = 0.0; sum_sq = 0.0; reg.init();
    }
    __device__ void tap_lags(int64_t (&lag)[1]) const { lag[0] = dead ? 0 : p; }
    __device__ void step(const Row<1> &r, int64_t t, const double (&x)[1], const double (&tp)[1], double (&y)[3]) {
        y[0] = y[1] = y[2] = pq_null();
        double v = x[0];
        if (dead || pq_isnull(v)) return;
        if (reg.first < 0) tail.start(t);
        bool regular = reg.push(t);
        count += 1; sum += v; sum_sq += v * v;
        if (count < p) return;
        if (count > p) {
            double old;
            if (regular) { old = tp[0]; tail.idx = t - p + 1; } else old = tail.pop(r.in[0]);
            sum -= old; sum_sq -= old * old; count -= 1;
        }
        double mean = sum / (double)p;
        double variance = (sum_sq / (double)p) - mean * mean;
        double sd = sqrt(fmax(variance, 0.0));
        y[0] = mean + up * sd; y[1] = mean; y[2] = mean - dn * sd;
    }
    Ring w;
    __host__ __device__ int64_t ring_slots() const { return p > 0 ? p : 1; }
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { init(r); w = ra.make(p); }
    __device__ void step_lds(int64_t, const double (&x)[1], double (&y)[3]) {
        y[0] = y[1] = y[2] = pq_null();
        double v = x[0];
        if (dead || pq_isnull(v)) return;
        count += 1; sum += v; sum_sq += v * v;
        double old = w.swap(v);
        if (count < p) return;
        if (count > p) { sum -= old; sum_sq -= old * old; count -= 1; }
        double mean = sum / (double)p;
        double variance = (sum_sq / (double)p) - mean * mean;
        double sd = sqrt(fmax(variance, 0.0));
        y[0] = mean + up * sd; y[1] = mean; y[2] = mean - dn * sd;
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return !dead && count >= p; } // full window: count stays at p
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[3]) {
        const double v = x[0];
        sum += v; sum_sq += v * v;
        const double old = w.swap(v);
        sum -= old; sum_sq -= old * old;
        double mean = sum / (double)p;
        double variance = (sum_sq / (double)p) - mean * mean;
        double sd = sqrt(fmax(variance, 0.0));
        y[0] = mean + up * sd; y[1] = mean; y[2] = mean - dn * sd;
    }
    static constexpr bool FAST_BATCH = true;
    static constexpr int FAST_UNROLL = 8;
    template <int N>
    __device__ void steps_fast(int64_t, const double (&x)[N][1], double (&y)[N][3]) {
        double v[N], old[N], s1[N], s2[N];
#pragma unroll
        for (int u = 0; u < N; u++) v[u] = x[u][0];
        w.swap_n<N>(v, old);
#pragma unroll
        for (int u = 0; u < N; u++) { // the running sums: the only loop-carried chain
            sum += v[u]; sum_sq += v[u] * v[u];
            sum -= old[u]; sum_sq -= old[u] * old[u];
            s1[u] = sum; s2[u] = sum_sq;
        }
#pragma unroll
        for (int u = 0; u < N; u++) { // per-row output arithmetic, independent across rows
            double mean = s1[u] / (double)p;
            double variance = (s2[u] / (double)p) - mean * mean;
            double sd = sqrt(fmax(variance, 0.0));
            y[u][0] = mean + up * sd; y[u][1] = mean; y[u][2] = mean - dn * sd;
        }
    }
};

struct DemaOp { // overlap.rs:543-598 (bitmap branch, decision D-2)
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 4;
    static constexpr int COST_NS = 105;
    int64_t p, count;
    double alpha, e0, e1, s0, s1;
    bool dead;
    __device__ void init(const Row<1> &r) {
        dead = (p <= 0 || r.len < 2 * p - 1);
        alpha = 2.0 / ((double)p + 1.0);
        count = 0; e0 = e1 = s0 = s1 = 0.0;
    }
    __device__ void step(const Row<1> &, int64_t, const double (&x)[1], double (&y)[1]) {
        y[0] = pq_null();
        double v = x[0];
        if (dead || pq_isnull(v)) return;
        count += 1;
        if (count < p) { s0 += v; }
        else if (count == p) { s0 += v; e0 = s0 / (double)p; s1 = e0; }
        else if (count < 2 * p - 1) { e0 = fma(alpha, v - e0, e0); s1 += e0; }
        else if (count == 2 * p - 1) { e0 = fma(alpha, v - e0, e0); s1 += e0; e1 = s1 / (double)p; }
        else {
            e0 = fma(alpha, v - e0, e0);
            e1 = fma(alpha, e0 - e1, e1);
            y[0] = 2.0 * e0 - e1;
        }
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return !dead && count >= 2 * p - 1; } // every later valid row is the last branch
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) {
        e0 = fma(alpha, x[0] - e0, e0);
        e1 = fma(alpha, e0 - e1, e1);
        y[0] = 2.0 * e0 - e1;
    }
    static constexpr int FAST_UNROLL = 16;
};

struct TemaOp { // overlap.rs:1177-1311
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 5;
    static constexpr int COST_NS = 110;
    int64_t p, count;
    double alpha, e0, e1, e2, s0, s1, s2;
    bool dead;
    __device__ void init(const Row<1> &r) {
        dead = (p <= 0 || r.len < 3 * p - 2);
        alpha = 2.0 / ((double)p + 1.0);
        count = 0; e0 = e1 = e2 = s0 = s1 = s2 = 0.0;
    }
    __device__ void step(const Row<1> &, int64_t, const double (&x)[1], double (&y)[1]) {
        y[0] = pq_null();
        double v = x[0];
        if (dead || pq_isnull(v)) return;
        count += 1;
        if (count < p) { s0 += v; return; }
        if (count == p) { s0 += v; e0 = s0 / (double)p; s1 = e0; return; }
        e0 = fma(alpha, v - e0, e0);
        if (count < 2 * p - 1) { s1 += e0; return; }
        if (count == 2 * p - 1) { s1 += e0; e1 = s1 / (double)p; s2 = e1; return; }
        e1 = fma(alpha, e0 - e1, e1);
        if (count < 3 * p - 2) { s2 += e1; return; }
        if (count == 3 * p - 2) { s2 += e1; e2 = s2 / (double)p; }
        else e2 = fma(alpha, e1 - e2, e2);
        y[0] = 3.0 * e0 - 3.0 * e1 + e2;
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return !dead && count >= 3 * p - 2; }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) {
        e0 = fma(alpha, x[0] - e0, e0);
        e1 = fma(alpha, e0 - e1, e1);
        e2 = fma(alpha, e1 - e2, e2);
        y[0] = 3.0 * e0 - 3.0 * e1 + e2;
    }
    static constexpr int FAST_UNROLL = 16;
};

struct T3Op { // overlap.rs:939-1175 (output formula :1160-1166, decision D-3; e5 never seeded)
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 6;
    static constexpr int COST_NS = 106;
    int64_t p, count;
    double alpha, c1, c2, c3, c4;
    double e0, e1, e2, e3, e4, e5, s0, s1, s2, s3, s4, s5;
    bool dead;
    __device__ void init(const Row<1> &r) {
        dead = (p <= 0 || r.len < 6 * p - 5);
        alpha = 2.0 / ((double)p + 1.0);
        count = 0;
        e0 = e1 = e2 = e3 = e4 = e5 = s0 = s1 = s2 = s3 = s4 = s5 = 0.0;
    }
    __device__ void step(const Row<1> &, int64_t, const double (&x)[1], double (&y)[1]) {
        y[0] = pq_null();
        double v = x[0];
        if (dead || pq_isnull(v)) return;
        count += 1;
        if (count < p) { s0 += v; return; }
        if (count == p) { s0 += v; e0 = s0 / (double)p; s1 = e0; return; }
        e0 = fma(alpha, v - e0, e0);
        if (count < 2 * p - 1) { s1 += e0; return; }
        if (count == 2 * p - 1) { s1 += e0; e1 = s1 / (double)p; s2 = e1; return; }
        e1 = fma(alpha, e0 - e1, e1);
        if (count < 3 * p - 2) { s2 += e1; return; }
        if (count == 3 * p - 2) { s2 += e1; e2 = s2 / (double)p; s3 = e2; return; }
        e2 = fma(alpha, e1 - e2, e2);
        if (count < 4 * p - 3) { s3 += e2; return; }
        if (count == 4 * p - 3) { s3 += e2; e3 = s3 / (double)p; s4 = e3; return; }
        e3 = fma(alpha, e2 - e3, e3);
        if (count < 5 * p - 4) { s4 += e3; return; }
        if (count == 5 * p - 4) { s4 += e3; e4 = s4 / (double)p; s5 = e4; return; }
        e4 = fma(alpha, e3 - e4, e4);
        if (count < 6 * p - 5) { s5 += e4; return; }
        e5 = fma(alpha, e4 - e5, e5);
        y[0] = fma(c1, e5, fma(c2, e4, fma(c3, e3, c4 * e2)));
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return !dead && count >= 6 * p - 5; }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) {
        e0 = fma(alpha, x[0] - e0, e0);
        e1 = fma(alpha, e0 - e1, e1);
        e2 = fma(alpha, e1 - e2, e2);
        e3 = fma(alpha, e2 - e3, e3);
        e4 = fma(alpha, e3 - e4, e4);
        e5 = fma(alpha, e4 - e5, e5);
        y[0] = fma(c1, e5, fma(c2, e4, fma(c3, e3, c4 * e2)));
    }
    static constexpr int FAST_UNROLL = 16;
};

struct WmaOp { // overlap.rs:1328-1399 (quirk Q-WMA kept)
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 7;
    static constexpr int COST_NS = 146;
    static constexpr int NTAP = 1;
    static constexpr int TAP_COL[1] = {0};
    int64_t p, count;
    double denominator, numerator;
    bool dead;
    Regular reg;
    ValidCursor tail;
    __device__ void init(const Row<1> &r) {
        dead = (p <= 0 || r.len < p);
        denominator = (double)(p * (p + 1) / 2);
        numerator = 0.0; count = 0; reg.init();
    }
    __device__ void tap_lags(int64_t (&lag)[1]) const { lag[0] = dead ? 0 : p; }
    __device__ void step(const Row<1> &r, int64_t t, const double (&x)[1], const double (&tp)[1], double (&y)[1]) {
        y[0] = pq_null();
        double v = x[0];
        if (dead || pq_isnull(v)) return;
        if (reg.first < 0) tail.start(t);
        bool regular = reg.push(t);
        count += 1;
        numerator += ((double)count) * v;
        if (count < p) return;
        if (count > p) {
            double old;
            if (regular) { old = tp[0]; tail.idx = t - p + 1; } else old = tail.pop(r.in[0]);
            numerator -= ((double)p) * old;
            count -= 1;
        }
        y[0] = numerator / denominator;
    }
    Ring w;
    __host__ __device__ int64_t ring_slots() const { return p > 0 ? p : 1; }
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { init(r); w = ra.make(p); }
    __device__ void step_lds(int64_t, const double (&x)[1], double (&y)[1]) {
        y[0] = pq_null();
        double v = x[0];
        if (dead || pq_isnull(v)) return;
        count += 1;
        numerator += ((double)count) * v;
        double old = w.swap(v);
        if (count < p) return;
        if (count > p) { numerator -= ((double)p) * old; count -= 1; }
        y[0] = numerator / denominator;
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return !dead && count >= p; } // full window: count stays at p
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) {
        numerator += ((double)(p + 1)) * x[0]; // the general path multiplies by (double)count after its increment to p + 1
        const double old = w.swap(x[0]);
        numerator -= ((double)p) * old;
        y[0] = numerator / denominator;
    }
    static constexpr bool FAST_BATCH = true;
    static constexpr int FAST_UNROLL = 16;
    template <int N>
    __device__ void steps_fast(int64_t, const double (&x)[N][1], double (&y)[N][1]) {
        double v[N], old[N], nm[N];
#pragma unroll
        for (int u = 0; u < N; u++) v[u] = x[u][0];
        w.swap_n<N>(v, old);
#pragma unroll
        for (int u = 0; u < N; u++) {
            numerator += ((double)(p + 1)) * v[u];
            numerator -= ((double)p) * old[u];
            nm[u] = numerator;
        }
#pragma unroll
        for (int u = 0; u < N; u++) y[u][0] = nm[u] / denominator;
    }
};

struct KamaOp { // overlap.rs:732-855, both passes fused into one walk
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 8;
    static constexpr int COST_NS = 178;
    static constexpr int NTAP = 3;
    static constexpr int TAP_COL[3] = {0, 0, 0};
    int64_t p;
    int64_t count;    // pass-1 count (saturates at p)
    int64_t j;        // number of valid values seen before the current one
    double sum, x0;
    int64_t c2;       // pass-2 count
    double kama, sum2;
    bool dead;
    Regular reg;
    ValidCursor cur_a; // x[j-p]      : window.pop_front()
    ValidCursor cur_b; // x[j-p+1]    : newer end of the popped diff
    ValidCursor cur_c; // x[j-2p+1]   : older end of the popped diff (once that diff is a p-lag diff)
    __device__ void init(const Row<1> &r) {
        dead = (p <= 1 || r.len < p); // p == 1: the reference pops an empty VecDeque and aborts (:775)
        count = 0; j = 0; sum = 0.0; x0 = 0.0; c2 = 0; kama = 0.0; sum2 = 0.0; reg.init();
    }
    __device__ void tap_lags(int64_t (&lag)[3]) const {
        lag[0] = dead ? 0 : p; lag[1] = dead ? 0 : p - 1; lag[2] = dead ? 0 : 2 * p - 1;
    }
    __device__ void step(const Row<1> &r, int64_t t, const double (&x)[1], const double (&tp)[3], double (&y)[1]) {
        y[0] = pq_null();
        double v = x[0];
        if (dead || pq_isnull(v)) return;
        const double *col = r.in[0];
        bool regular = reg.push(t);
        if (count == 0) { // :760-764
            count = 1; x0 = v; j = 1;
            cur_a.start(t); cur_c.start(t);
            return;
        }
        if (j == 1) cur_b.start(t); // second valid value = index 1 of the compacted series
        if (count < p) { // :765-772: diffs against window.front() == first valid value
            count += 1;
            sum += fabs(v - x0);
            j += 1;
            return;
        }
        // :773-779; the popped diff has compacted index k = j-p+1: |x_k - x_0| while k < p, else |x_k - x_{k-p}|
        int64_t k = j - p + 1;
        double xa, xk, xc = 0.0;
        if (regular) {
            xa = tp[0]; cur_a.idx = t - p + 1;
            xk = tp[1]; cur_b.idx = t - p + 2;
            if (k >= p) { xc = tp[2]; cur_c.idx = t - 2 * p + 2; }
        } else {
            xa = cur_a.pop(col);
            xk = cur_b.pop(col);
            if (k >= p) xc = cur_c.pop(col);
        }
        double diff_abs = fabs(v - xa);
        double popped = (k < p) ? fabs(xk - x0) : fabs(xk - xc);
        sum += diff_abs - popped;
        double er = diff_abs / sum;
        j += 1;
        // pass 2 (:815-852)
        double sc_sqrt = er * (2.0 / 3.0 - 2.0 / 31.0) + 2.0 / 31.0;
        double sc = sc_sqrt * sc_sqrt;
        if (c2 < p) { c2 += 1; sum2 += v; return; }
        if (c2 == p) { c2 += 1; kama = sum2 / (double)p; y[0] = kama; return; }
        kama = fma(sc, v - kama, kama);
        y[0] = kama;
    }
    Ring w; // last 2p valid values
    __host__ __device__ int64_t ring_slots() const { return p > 0 ? 2 * p : 1; }
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { init(r); w = ra.make(2 * p); }
    __device__ void step_lds(int64_t, const double (&x)[1], double (&y)[1]) {
        y[0] = pq_null();
        double v = x[0];
        if (dead || pq_isnull(v)) return;
        if (count == 0) { count = 1; x0 = v; j = 1; w.push(v); return; }
        if (count < p) { count += 1; sum += fabs(v - x0); j += 1; w.push(v); return; }
        int64_t k = j - p + 1;
        double xa = w.get((int)p), xk = w.get((int)p - 1);
        double diff_abs = fabs(v - xa);
        double popped = (k < p) ? fabs(xk - x0) : fabs(xk - w.get((int)(2 * p - 1)));
        sum += diff_abs - popped;
        double er = diff_abs / sum;
        j += 1;
        w.push(v);
        double sc_sqrt = er * (2.0 / 3.0 - 2.0 / 31.0) + 2.0 / 31.0;
        double sc = sc_sqrt * sc_sqrt;
        if (c2 < p) { c2 += 1; sum2 += v; return; }
        if (c2 == p) { c2 += 1; kama = sum2 / (double)p; y[0] = kama; return; }
        kama = fma(sc, v - kama, kama);
        y[0] = kama;
    }
    // count saturated at p, the popped diff is a p-lag diff (k = j-p+1 >= p), pass 2 seeded (c2 > p): j and c2 stop mattering
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return !dead && count >= p && j >= 2 * p - 1 && c2 > p; }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) {
        const double v = x[0];
        const double xa = w.get((int)p), xk = w.get((int)p - 1), xc = w.get((int)(2 * p - 1));
        double diff_abs = fabs(v - xa);
        double popped = fabs(xk - xc);
        sum += diff_abs - popped;
        double er = diff_abs / sum;
        w.push(v);
        double sc_sqrt = er * (2.0 / 3.0 - 2.0 / 31.0) + 2.0 / 31.0;
        double sc = sc_sqrt * sc_sqrt;
        kama = fma(sc, v - kama, kama);
        y[0] = kama;
    }
};

struct MidpointOp { // overlap.rs:180-278 incl. quirk Q-MID (min never expires => cumulative min)
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 9;
    static constexpr int COST_NS = 246;
    int64_t p;
    RollExt<true> mx;
    double mn;
    bool any, frozen; // frozen: a NaN value has entered the minimum's deque -- it is never popped (no `>=` holds for it, and that deque
                      // never expires its front), so nothing newer reaches the front again: the minimum stays what it was
    const double *col; // the series (tiled bodies: kept for roll_ext_ref)
    __device__ void init(const Row<1> &r) { mx.init(p); mn = 0.0; any = false; frozen = false; col = r.in[0]; }
    __device__ __forceinline__ void low(double v) {
        if (!any) { mn = v; any = true; }
        else if (!frozen && v <= mn) mn = v;
        frozen |= v != v;
    }
    __device__ void step(const Row<1> &r, int64_t t, const double (&x)[1], double (&y)[1]) {
        double v = x[0];
        if (p <= 0 || pq_isnull(v)) { y[0] = pq_null(); return; } // p <= 0: decision D-7b
        double m = mx.nan_row(r.in[0], t, v, mx.step(r.in[0], t, v));
        low(v);
        y[0] = (m + mn) / 2.0;
    }
    Ring wc, ws;
    __host__ __device__ int64_t ring_slots() const { return p > 0 ? 2 * p : 2; }
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { init(r); mx.init_ring(); wc = ra.make(p); ws = ra.make(p); }
    __device__ void step_lds(int64_t t, const double (&x)[1], double (&y)[1]) {
        double v = x[0];
        if (p <= 0 || pq_isnull(v)) { y[0] = pq_null(); return; }
        double m = mx.nan_row(col, t, v, mx.step_ring2(wc, ws, v));
        low(v);
        y[0] = (m + mn) / 2.0;
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return p > 0 && any; }
    __device__ void step_fast(int64_t t, const double (&x)[1], double (&y)[1]) {
        const double v = x[0];
        const double m = mx.nan_row(col, t, v, mx.step_ring2(wc, ws, v));
        mn = (!frozen && v <= mn) ? v : mn;
        frozen |= v != v;
        y[0] = (m + mn) / 2.0;
    }
};

struct MidpriceOp { // overlap.rs:281-404, no-bitmap branches; null in either input -> null (D-7)
    static constexpr int NIN = 2, NOUT = 1;
    static constexpr int SEQ_ID = 10;
    static constexpr int COST_NS = 416;
    int64_t p;
    RollExt<true> mx;
    RollExt<false> mn;
    const double *ch, *cl; // the two series (tiled bodies: kept for roll_ext_ref)
    __device__ void init(const Row<2> &r) { mx.init(p); mn.init(p); ch = r.in[0]; cl = r.in[1]; }
    __device__ void step(const Row<2> &r, int64_t t, const double (&x)[2], double (&y)[1]) {
        double hm = pq_null(), lm = pq_null();
        if (p <= 0) { y[0] = pq_null(); return; } // decision D-7b
        if (!pq_isnull(x[0])) hm = mx.nan_row(r.in[0], t, x[0], mx.step(r.in[0], t, x[0]));
        if (!pq_isnull(x[1])) lm = mn.nan_row(r.in[1], t, x[1], mn.step(r.in[1], t, x[1]));
        y[0] = (pq_isnull(hm) || pq_isnull(lm)) ? pq_null() : (hm + lm) / 2.0;
    }
    Ring whc, whs, wlc, wls;
    __host__ __device__ int64_t ring_slots() const { return p > 0 ? 4 * p : 4; }
    __device__ void init_lds(const Row<2> &r, RingAlloc &ra) {
        init(r); mx.init_ring(); mn.init_ring();
        whc = ra.make(p); whs = ra.make(p); wlc = ra.make(p); wls = ra.make(p);
    }
    __device__ void step_lds(int64_t t, const double (&x)[2], double (&y)[1]) {
        if (p <= 0) { y[0] = pq_null(); return; }
        double hm = pq_null(), lm = pq_null();
        if (!pq_isnull(x[0])) hm = mx.nan_row(ch, t, x[0], mx.step_ring2(whc, whs, x[0]));
        if (!pq_isnull(x[1])) lm = mn.nan_row(cl, t, x[1], mn.step_ring2(wlc, wls, x[1]));
        y[0] = (pq_isnull(hm) || pq_isnull(lm)) ? pq_null() : (hm + lm) / 2.0;
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return p > 0; }
    __device__ void step_fast(int64_t t, const double (&x)[2], double (&y)[1]) {
        const double hm = mx.nan_row(ch, t, x[0], mx.step_ring2(whc, whs, x[0]));
        const double lm = mn.nan_row(cl, t, x[1], mn.step_ring2(wlc, wls, x[1]));
        y[0] = (hm + lm) / 2.0;
    }
};

// MIDPRICE as a ROW op: its value is a pure function of the last p VALID highs / lows up to the row (the front of the reference's
// monotonic deques, overlap.rs:325-345 / :378-398, holds exactly their extremum; extrema are exact in any order), so it needs no
// serial walk: thread per row, the window from L1 -- one more job of the suite's fused ROW grid instead of a lane-per-symbol job
// with 39 KB of LDS rings (1.04 ms alone at 5 000 x 2 520, the most expensive one-column job of a step).
struct MidpriceRowOp {
    static constexpr int NIN = 2, NOUT = 1;
    static constexpr int ROW_ID = 18;
    typedef double OutT;
    int64_t p;
    __device__ void eval(const Row<2> &r, int64_t t, double (&y)[1]) {
        y[0] = pq_null();
        if (p <= 0) return;                                   // decision D-7b
        const double h0 = r.in[0][t], l0 = r.in[1][t];
        if (pq_isnull(h0) || pq_isnull(l0)) return;          // a null in either input -> null row (D-7)
        y[0] = (roll_ext_ref<true>(r.in[0], t, p) + roll_ext_ref<false>(r.in[1], t, p)) / 2.0; // overlap.rs:401
    }
};

// SAR / SAREXT (decision D-4: TA-Lib algorithm; nulls -> 0.0 as overlap.rs:445-450)
__device__ __forceinline__ double n0(double x) { return pq_isnull(x) ? 0.0 : x; }

struct SarextOp {
    static constexpr bool RG_GATHER = true; // a direct call on a ragged batch keeps the per-lane form: alone on the chip it beats re-housing + the tiled body (profiles/r05_bench_ragged.json)
    static constexpr int64_t DIRECT_LANE_MAX = 16384; // a DIRECT call on a regular batch of up to this many series runs the per-lane form: alone on the chip it is faster (profiles/r05_direct_lane.json)
    static constexpr int NIN = 2, NOUT = 1;
    static constexpr int SEQ_ID = 11;
    static constexpr int COST_NS = 450;
    bool ext; // false: plain SAR (no offset, no negation)
    double startvalue, offset, ai_long, a_long, am_long, ai_short, a_short, am_short;
    double af_long, af_short, ep, sar, new_low, new_high;
    bool is_long;
    __device__ void init(const Row<2> &r) {
        af_long = ai_long; af_short = ai_short;
        if (af_long > am_long) af_long = ai_long = am_long;
        if (a_long > am_long) a_long = am_long;
        if (af_short > am_short) af_short = ai_short = am_short;
        if (a_short > am_short) a_short = am_short;
        if (r.len < 2) return;
        double h0 = n0(r.in[0][0]), l0 = n0(r.in[1][0]), h1 = n0(r.in[0][1]), l1 = n0(r.in[1][1]);
        if (startvalue == 0.0) {
            double diffP = h1 - h0, diffM = l0 - l1;
            is_long = !(diffM > 0.0 && diffP < diffM);
            if (is_long) { ep = h1; sar = l0; } else { ep = l1; sar = h0; }
        } else if (startvalue > 0.0) { is_long = true; ep = h1; sar = startvalue; }
        else { is_long = false; ep = l1; sar = fabs(startvalue); }
        new_low = l1; new_high = h1;
    }
    __device__ void step(const Row<2> &r, int64_t t, const double (&x)[2], double (&y)[1]) {
        y[0] = pq_null();
        if (r.len < 2 || t == 0) return;
        row(x, y);
    }
    static constexpr bool HAS_FAST = true;
    static constexpr bool FAST_NULL_OK = true; // N-0: nulls become 0.0 in the row body
    __device__ bool steady(int64_t t0) const { return t0 >= 1; } // rows >= 1 exist, so the series has at least two
    __device__ void step_fast(int64_t, const double (&x)[2], double (&y)[1]) { row(x, y); }
    __device__ __forceinline__ void row(const double (&x)[2], double (&y)[1]) {
        double prev_low = new_low, prev_high = new_high;
        new_high = n0(x[0]); new_low = n0(x[1]);
        if (is_long) {
            if (new_low <= sar) {
                is_long = false; sar = ep;
                if (sar < prev_high) sar = prev_high;
                if (sar < new_high) sar = new_high;
                if (ext && offset != 0.0) sar += sar * offset;
                y[0] = ext ? -sar : sar;
                af_short = ai_short; ep = new_low;
                sar = sar + af_short * (ep - sar);
                if (sar < prev_high) sar = prev_high;
                if (sar < new_high) sar = new_high;
            } else {
                y[0] = sar;
                if (new_high > ep) { ep = new_high; af_long += a_long; if (af_long > am_long) af_long = am_long; }
                sar = sar + af_long * (ep - sar);
                if (sar > prev_low) sar = prev_low;
                if (sar > new_low) sar = new_low;
            }
        } else {
            if (new_high >= sar) {
                is_long = true; sar = ep;
                if (sar > prev_low) sar = prev_low;
                if (sar > new_low) sar = new_low;
                if (ext && offset != 0.0) sar -= sar * offset;
                y[0] = sar;
                af_long = ai_long; ep = new_high;
                sar = sar + af_long * (ep - sar);
                if (sar > prev_low) sar = prev_low;
                if (sar > new_low) sar = new_low;
            } else {
                y[0] = ext ? -sar : sar;
                if (new_low < ep) { ep = new_low; af_short += a_short; if (af_short > am_short) af_short = am_short; }
                sar = sar + af_short * (ep - sar);
                if (sar < prev_high) sar = prev_high;
                if (sar < new_high) sar = new_high;
            }
        }
    }
};

// MAVP (decision D-4) as independent jobs, one per candidate period P: run the reference MA for period P over
// the whole series and write only the rows whose clamped period is P (every row is written by exactly one job).
template <class Inner>
struct MavpSelOp {
    static constexpr int NIN = 2, NOUT = 1; // real (nulls -> 0.0 here, overlap.rs:416-424), periods
    static constexpr int SEQ_ID = 100 + Inner::SEQ_ID;
    static constexpr bool MASKED = true;
    static constexpr int NTAP = NTap<Inner>::value; // the inner MA's lag taps all read column 0
    static constexpr int TAP_COL[3] = {0, 0, 0};
    int64_t P, minp, maxp;
    Inner inner;
    __device__ void init(const Row<2> &r) {
        Row<1> r1; r1.in[0] = r.in[0]; r1.len = r.len;
        inner.init(r1);
    }
    __device__ void tap_lags(int64_t (&lag)[NTAP > 0 ? NTAP : 1]) const {
        if constexpr (NTAP > 0) inner.tap_lags(lag);
    }
    __device__ double pick(int64_t t, double per, double ma) const {
        int64_t pi = (int64_t)n0(per);
        if (pi < minp) pi = minp;
        if (pi > maxp) pi = maxp;
        return (pi != P) ? pq_skip() : ((t >= maxp - 1) ? ma : pq_null());
    }
    __device__ void step(const Row<2> &r, int64_t t, const double (&x)[2], double (&y)[1]) {
        Row<1> r1; r1.in[0] = r.in[0]; r1.len = r.len;
        double xi[1] = {n0(x[0])}, yi[1];
        if constexpr (NTAP == 0) inner.step(r1, t, xi, yi);
        y[0] = pick(t, x[1], yi[0]);
    }
    __device__ void step(const Row<2> &r, int64_t t, const double (&x)[2], const double (&tp)[NTAP > 0 ? NTAP : 1], double (&y)[1]) {
        Row<1> r1; r1.in[0] = r.in[0]; r1.len = r.len;
        double xi[1] = {n0(x[0])}, yi[1];
        if constexpr (NTAP > 0) {
            double tz[NTAP > 0 ? NTAP : 1];
#pragma unroll
            for (int k = 0; k < NTAP; k++) tz[k] = n0(tp[k]);
            inner.step(r1, t, xi, tz, yi);
        }
        y[0] = pick(t, x[1], yi[0]);
    }
    // LDS body: forward to the inner op's ring variant when it has one
    __host__ __device__ int64_t ring_slots() const {
        if constexpr (HasRings<Inner>::value) return inner.ring_slots(); else return 0;
    }
    __device__ void init_lds(const Row<2> &r, RingAlloc &ra) {
        Row<1> r1; r1.in[0] = r.in[0]; r1.len = r.len;
        if constexpr (HasRings<Inner>::value) inner.init_lds(r1, ra); else inner.init(r1);
    }
    __device__ void step_lds(int64_t t, const double (&x)[2], double (&y)[1]) {
        double xi[1] = {n0(x[0])}, yi[1];
        if constexpr (HasRings<Inner>::value) inner.step_lds(t, xi, yi);
        else { Row<1> r1; r1.in[0] = nullptr; r1.len = 0; inner.step(r1, t, xi, yi); }
        y[0] = pick(t, x[1], yi[0]);
    }
};
// select rows of a materialised MA(P) column (used for matypes that are not a single SEQ op)
struct MavpPickOp {
    static constexpr int NIN = 2, NOUT = 1; // ma(P), periods
    static constexpr int SEQ_ID = 12;
    static constexpr bool MASKED = true;
    int64_t P, minp, maxp;
    __device__ void init(const Row<2> &) {}
    __device__ void step(const Row<2> &, int64_t t, const double (&x)[2], double (&y)[1]) {
        int64_t pi = (int64_t)n0(x[1]);
        if (pi < minp) pi = minp;
        if (pi > maxp) pi = maxp;
        y[0] = (pi != P) ? pq_skip() : ((t >= maxp - 1) ? x[0] : pq_null());
    }
};
static inline void t3_coeffs(T3Op &op, double vf) { // overlap.rs:949-952
    op.c1 = -(vf * vf * vf);
    op.c2 = 3.0 * (vf * vf) - 3.0 * op.c1;
    op.c3 = -2.0 * op.c2 - 3.0 * op.c1 - 3.0 * vf;
    op.c4 = 1.0 - op.c1 - op.c2 - op.c3;
}
struct FillNullOp {
    static constexpr int NIN = 0, NOUT = 1;
    typedef double OutT;
    __device__ void eval(const Row<0> &, int64_t, double (&y)[1]) { y[0] = pq_null(); }
};
struct ReplaceNullOp { // nulls -> 0.0 (N-0 families)
    static constexpr int NIN = 1, NOUT = 1;
    typedef double OutT;
    __device__ void eval(const Row<1> &r, int64_t t, double (&y)[1]) { y[0] = n0(r.in[0][t]); }
};

