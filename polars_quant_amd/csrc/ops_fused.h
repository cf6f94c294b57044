// ops_fused.h -- single-walk versions of the functions the reference builds from several passes.
// Each op keeps every intermediate series in registers / LDS rings, so a composite is ONE sequential job (one
// phase of a suite grid) instead of a chain of launches through scratch columns.  Arithmetic and null rules are
// those of the chained form (bit-identical; tests compare both against the oracle).  These ops exist only for
// the LDS body; the C ABI falls back to the chained launches when the LDS body cannot be used.
#pragma once
#include "ops_misc.h"
#include "ops_momentum.h"
#include "ops_overlap.h"

// calc_ma (overlap.rs:857-869) restricted to the types that are a single core: SMA (0, 7, default) and EMA (1).  One state for both
// kinds (the kind is a run-time parameter, so two separate cores would both stay live in registers): `acc` is the SMA's running sum,
// or the EMA's seed sum and, from the seed row on, the average itself; every operation and its order are SmaCore's / EmaCore's
// (pq_cores.h; overlap.rs:871-937 / :660-730).
struct Ma2 {
    int kind; // 0 SMA, 1 EMA
    int dead;
    int64_t p, count;
    double k;   // SMA: 1/p ; EMA: alpha = 2/(p+1)
    double acc;
    Ring w;
    __host__ __device__ static bool supports(int64_t matype) { return !(matype >= 2 && matype <= 6) && matype != 8; }
    __host__ __device__ static int64_t slots(int64_t matype, int64_t p) { return matype == 1 ? 0 : (p > 0 ? p : 1); }
    __device__ __forceinline__ void init_core(int kind_, int64_t p_, int64_t n) {
        kind = kind_;
        p = p_;
        dead = (p <= 0 || n < p) ? 1 : 0;
        k = pq_uniform(kind ? 2.0 / ((double)p + 1.0) : 1.0 / (double)p);
        count = 0;
        acc = 0.0;
    }
    __device__ __forceinline__ void init(int64_t matype, int64_t p_, int64_t n, RingAlloc &ra) {
        init_core(matype == 1 ? 1 : 0, p_, n);
        if (!kind) w = ra.make(p_);
    }
    __device__ __forceinline__ bool steady() const { return !dead && count >= p; } // SMA: full window (count stays at p); EMA: seeded
    // SMA with the expiring value supplied by the caller (`old` = the valid value pushed p pushes ago)
    __device__ __forceinline__ double step_old(double v, double old) {
        if (wave_all(!dead && count >= p && !pq_isnull(v))) { // full window on the whole wave
            acc += v;
            acc -= old;
            return acc * k;
        }
        if (dead || pq_isnull(v)) return pq_null();
        count += 1;
        acc += v;
        if (count < p) return pq_null();
        if (count > p) {
            acc -= old;
            count -= 1;
        }
        return acc * k;
    }
    __device__ __forceinline__ double fast_old(double v, double old) {
        acc += v;
        acc -= old;
        return acc * k;
    }
    __device__ __forceinline__ double step(double v) {
        if (!kind) { // the ring holds the last p valid values: the popped value is ring.swap(v)
            if (wave_all(!dead && count >= p && !pq_isnull(v))) {
                acc += v;
                acc -= w.swap(v);
                return acc * k;
            }
            if (dead || pq_isnull(v)) return pq_null();
            count += 1;
            acc += v;
            const double old = w.swap(v);
            if (count < p) return pq_null();
            if (count > p) {
                acc -= old;
                count -= 1;
            }
            return acc * k;
        }
        if (wave_all(!dead && count >= p && !pq_isnull(v))) { // seeded on the whole wave
            acc = fma(k, v - acc, acc);
            return acc;
        }
        if (dead || pq_isnull(v)) return pq_null();
        count += 1;
        if (count < p) {
            acc += v;
            return pq_null();
        } else if (count == p) {
            acc += v;
            acc = acc / (double)p;
            return acc;
        }
        acc = fma(k, v - acc, acc);
        return acc;
    }
    __device__ __forceinline__ double fast(double v) {
        if (!kind) {
            acc += v;
            acc -= w.swap(v);
            return acc * k;
        }
        acc = fma(k, v - acc, acc);
        return acc;
    }
};

// Two moving averages of the SAME input: when both are SMAs they share one ring (depth = the longer period; the shorter
// one reads its expiring value with Ring::get) -- same values, same operation order, 6-13 KB less LDS per job.
struct Ma2Pair {
    Ma2 a, b;
    Ring w;
    bool shared;
    int pa, pb;
    __host__ __device__ static int64_t slots(int64_t mta, int64_t pa_, int64_t mtb, int64_t pb_) {
        if (mta != 1 && mtb != 1) { int64_t m = pa_ > pb_ ? pa_ : pb_; return m > 0 ? m : 1; }
        return Ma2::slots(mta, pa_) + Ma2::slots(mtb, pb_);
    }
    __device__ __forceinline__ void init(int64_t mta, int64_t pa_, int64_t mtb, int64_t pb_, int64_t n, RingAlloc &ra) {
        shared = (mta != 1 && mtb != 1);
        if (shared) {
            a.init_core(0, pa_, n); b.init_core(0, pb_, n);
            pa = (int)(pa_ > 0 ? pa_ : 1); pb = (int)(pb_ > 0 ? pb_ : 1);
            w = ra.make(pa_ > pb_ ? pa_ : pb_);
            // keeps this store where it is: merged with the store of b's own ring below (one store to a run-time offset inside the
            // op) it would pin the whole op struct to scratch memory instead of registers
            asm volatile("" ::: "memory");
        } else {
            a.init(mta, pa_, n, ra); b.init(mtb, pb_, n, ra);
        }
    }
    __device__ __forceinline__ void step(double v, double &ya, double &yb) {
        if (!shared) { ya = a.step(v); yb = b.step(v); return; }
        if (pq_isnull(v)) { ya = yb = pq_null(); return; }
        const double oa = w.get(pa), ob = w.get(pb);
        ya = a.step_old(v, oa);
        yb = b.step_old(v, ob);
        w.push(v);
    }
    __device__ __forceinline__ bool steady() const { return a.steady() && b.steady(); }
    __device__ __forceinline__ void fast(double v, double &ya, double &yb) {
        if (!shared) { ya = a.fast(v); yb = b.fast(v); return; }
        const double oa = w.get(pa), ob = w.get(pb);
        ya = a.fast_old(v, oa);
        yb = b.fast_old(v, ob);
        w.push(v);
    }
};

struct TrimaOp {
    static constexpr bool LDS_ONLY = true; // overlap.rs:1313-1326: sma(sma(x, k1), k2)
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 70;
    static constexpr int COST_NS = 150;
    int64_t k1, k2;
    SmaCore a, b;
    Ring wa, wb;
    __host__ __device__ int64_t ring_slots() const { return (k1 > 0 ? k1 : 1) + (k2 > 0 ? k2 : 1); }
    __device__ void init(const Row<1> &) {}
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { a.init(k1, r.len); b.init(k2, r.len); wa = ra.make(k1); wb = ra.make(k2); }
    __device__ void step(const Row<1> &, int64_t, const double (&)[1], double (&y)[1]) { y[0] = pq_null(); }
    __device__ void step_lds(int64_t, const double (&x)[1], double (&y)[1]) { y[0] = b.step_ring(wb, a.step_ring(wa, x[0])); }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return a.steady() && b.steady(); }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) { y[0] = b.fast_ring(wb, a.fast_ring(wa, x[0])); }
};

// SMA(p) and MA(p, matype 0) of one input are the same walk (overlap.rs:857-869: `ma` dispatches to calc_sma, :871-937): one job, the
// value written to both columns -- the common-subexpression form of two calls a query makes with the wrapper defaults
// (python/polars_quant/talib/overlap.py: SMA(timeperiod=30), MA(timeperiod=30, matype=0)).
struct SmaDupOp {
    static constexpr bool LDS_ONLY = true;
    static constexpr int NIN = 1, NOUT = 2;
    static constexpr int SEQ_ID = 88;
    static constexpr int ALG_COLS = 4; // the two calls it replaces
    static constexpr int COST_NS = 130;
    SmaOp a;
    __host__ __device__ int64_t ring_slots() const { return a.ring_slots(); }
    __device__ void init(const Row<1> &) {}
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { a.init_lds(r, ra); }
    __device__ void step(const Row<1> &, int64_t, const double (&)[1], double (&y)[2]) { y[0] = y[1] = pq_null(); }
    __device__ void step_lds(int64_t t, const double (&x)[1], double (&y)[2]) { double v[1]; a.step_lds(t, x, v); y[0] = y[1] = v[0]; }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return a.steady(t0); }
    __device__ void step_fast(int64_t t, const double (&x)[1], double (&y)[2]) { double v[1]; a.step_fast(t, x, v); y[0] = y[1] = v[0]; }
    static constexpr bool FAST_BATCH = true;
    static constexpr int FAST_UNROLL = 8;
    template <int N>
    __device__ void steps_fast(int64_t t, const double (&x)[N][1], double (&y)[N][2]) {
        double v[N][1];
        a.template steps_fast<N>(t, x, v);
#pragma unroll
        for (int u = 0; u < N; u++) y[u][0] = y[u][1] = v[u][0];
    }
};

template <int MODE> // 0 APO, 1 PPO (decision D-6)
struct MaDiffOp {
    static constexpr bool LDS_ONLY = true;
    static constexpr int NIN = 1, NOUT = 1;
    static constexpr int SEQ_ID = 71 + MODE;
    static constexpr int COST_NS = 500; // a scheduling weight rather than a duration (solo 0.37 us per row for the pair): a coordinate search over the
                                        // weights of the benchmark suite's jobs found this one change (x 2: earlier in the LONG grid, priority 3), -1 % per step
    int64_t fast, slow, matype;
    Ma2Pair fs;
    __host__ __device__ int64_t ring_slots() const { return Ma2Pair::slots(matype, fast, matype, slow); }
    __device__ void init(const Row<1> &) {}
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { fs.init(matype, fast, matype, slow, r.len, ra); }
    __device__ void step(const Row<1> &, int64_t, const double (&)[1], double (&y)[1]) { y[0] = pq_null(); }
    __device__ void step_lds(int64_t, const double (&x)[1], double (&y)[1]) {
        double a, b;
        fs.step(x[0], a, b);
        if (pq_isnull(a) || pq_isnull(b)) { y[0] = pq_null(); return; }
        if (MODE == 0) y[0] = a - b;
        else y[0] = (b == 0.0) ? pq_null() : (a - b) / b * 100.0;
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return fs.steady(); }
    __device__ void step_fast(int64_t, const double (&x)[1], double (&y)[1]) {
        double a, b;
        fs.fast(x[0], a, b);
        if (MODE == 0) y[0] = a - b;
        else { const double v = (a - b) / b * 100.0; y[0] = (b == 0.0) ? pq_null() : v; }
    }
};

struct MacdextOp {
    static constexpr bool LDS_ONLY = true; // momentum.py:83-88
    static constexpr int NIN = 1, NOUT = 3;
    static constexpr int SEQ_ID = 73;
    static constexpr int COST_NS = 338;
    int64_t fast, fastmt, slow, slowmt, sig, sigmt;
    Ma2Pair fs;
    Ma2 g;
    __host__ __device__ int64_t ring_slots() const { return Ma2Pair::slots(fastmt, fast, slowmt, slow) + Ma2::slots(sigmt, sig); }
    __device__ __forceinline__ void init(const Row<1> &) {}
    __device__ __forceinline__ void init_lds(const Row<1> &r, RingAlloc &ra) {
        fs.init(fastmt, fast, slowmt, slow, r.len, ra); g.init(sigmt, sig, r.len, ra);
    }
    __device__ __forceinline__ void step(const Row<1> &, int64_t, const double (&)[1], double (&y)[3]) { y[0] = y[1] = y[2] = pq_null(); }
    __device__ __forceinline__ void step_lds(int64_t, const double (&x)[1], double (&y)[3]) {
        double a, b;
        fs.step(x[0], a, b);
        double m = (pq_isnull(a) || pq_isnull(b)) ? pq_null() : a - b;
        double d = g.step(m); // N-A: a null macd row is skipped by the signal MA
        y[0] = m; y[1] = d;
        y[2] = (pq_isnull(m) || pq_isnull(d)) ? pq_null() : m - d;
    }
    static constexpr bool HAS_FAST = true;
    __device__ __forceinline__ bool steady(int64_t) const { return fs.steady() && g.steady(); }
    __device__ __forceinline__ void step_fast(int64_t, const double (&x)[1], double (&y)[3]) {
        double a, b;
        fs.fast(x[0], a, b);
        const double m = a - b;
        const double d = g.fast(m);
        y[0] = m; y[1] = d; y[2] = m - d;
    }
};

// Polars rolling_min/rolling_max(window=k) + fastk (momentum.py:181-183): the frame is the last k ROWS; the result is
// null until the frame holds k non-null rows, i.e. whenever any of its rows is null.  Extrema by block decomposition
// (see RollExt::step_ring2); nulls are counted per frame and entered as neutral elements.
struct FastkCore {
    int64_t k, rows, nulls_h, nulls_l;
    Ring nh, nl;          // null flags of the frame rows (1.0 / 0.0)
    RollExt<true> mx;
    RollExt<false> mn;
    Ring hc, hs, lc, ls;
    __host__ __device__ static int64_t slots(int64_t k) { return 6 * (k > 0 ? k : 1); }
    __device__ void init(int64_t k_, RingAlloc &ra) {
        k = k_; rows = 0; nulls_h = nulls_l = 0;
        nh = ra.make(k); nl = ra.make(k);
        mx.init(k); mn.init(k); mx.init_ring(); mn.init_ring();
        hc = ra.make(k); hs = ra.make(k); lc = ra.make(k); ls = ra.make(k);
    }
    __device__ double step(double h, double l, double c) {
        if (k <= 0) return pq_null();
        const bool hn_ = pq_isnull(h), ln_ = pq_isnull(l);
        double oh = nh.swap(hn_ ? 1.0 : 0.0), ol = nl.swap(ln_ ? 1.0 : 0.0);
        if (rows >= k) { nulls_h -= (oh != 0.0) ? 1 : 0; nulls_l -= (ol != 0.0) ? 1 : 0; }
        nulls_h += hn_ ? 1 : 0; nulls_l += ln_ ? 1 : 0;
        rows += 1;
        // a null enters the window structures as the neutral element (it can only matter in frames that are null anyway)
        double hmax = mx.step_ring2(hc, hs, hn_ ? -1.7976931348623157e308 : h);
        double lmin = mn.step_ring2(lc, ls, ln_ ? 1.7976931348623157e308 : l);
        if (rows < k || nulls_h || nulls_l || pq_isnull(c)) return pq_null();
        return (c - lmin) * 100.0 / (hmax - lmin);
    }
    // full frame without a null row: the flag rings are all zero and stay so (their position is irrelevant then), rows
    // is only compared with k
    __device__ bool steady() const { return k > 0 && rows >= k && nulls_h == 0 && nulls_l == 0; }
    __device__ double fast(double h, double l, double c) {
        const double hmax = mx.step_ring2(hc, hs, h);
        const double lmin = mn.step_ring2(lc, ls, l);
        return (c - lmin) * 100.0 / (hmax - lmin);
    }
};

template <int MODE> // 0 STOCH -> (slowk, slowd); 1 STOCHF -> (fastk, fastd)     momentum.py:178-195
struct StochOp {
    static constexpr bool RG_GATHER = true; // a direct call on a ragged batch keeps the per-lane form: alone on the chip it beats re-housing + the tiled body (profiles/r05_bench_ragged.json)
    static constexpr int64_t DIRECT_LANE_MAX = 16384; // a DIRECT call on a regular batch of up to this many series runs the per-lane form: alone on the chip it is faster (profiles/r05_direct_lane.json)
    static constexpr bool LDS_ONLY = true;
    static constexpr int NIN = 3, NOUT = 2; // high, low, close
    static constexpr int SEQ_ID = 74 + MODE;
    static constexpr int COST_NS = MODE == 0 ? 637 : 579;
    static constexpr bool HEAVY = false; // (STOCH's three MA cores behind the rolling extrema fit the light kernel's 192 VGPRs since Ma2 holds one overlaid state)
    int64_t fastk, p1, mt1, p2, mt2; // STOCH: slowk/slowd MA params; STOCHF: (p1, mt1) = fastd, second MA unused
    FastkCore fk;
    Ma2 m1, m2;
    __host__ __device__ int64_t ring_slots() const {
        return FastkCore::slots(fastk) + Ma2::slots(mt1, p1) + (MODE == 0 ? Ma2::slots(mt2, p2) : 0);
    }
    __device__ void init(const Row<3> &) {}
    __device__ void init_lds(const Row<3> &r, RingAlloc &ra) {
        fk.init(fastk, ra);
        m1.init(mt1, p1, r.len, ra);
        if (MODE == 0) m2.init(mt2, p2, r.len, ra);
    }
    __device__ void step(const Row<3> &, int64_t, const double (&)[3], double (&y)[2]) { y[0] = y[1] = pq_null(); }
    __device__ void step_lds(int64_t, const double (&x)[3], double (&y)[2]) {
        double k = fk.step(x[0], x[1], x[2]);
        double a = m1.step(k);
        if (MODE == 0) { y[0] = a; y[1] = m2.step(a); }
        else { y[0] = k; y[1] = a; }
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return fk.steady() && m1.steady() && (MODE != 0 || m2.steady()); }
    __device__ void step_fast(int64_t, const double (&x)[3], double (&y)[2]) {
        const double k = fk.fast(x[0], x[1], x[2]);
        const double a = m1.fast(k);
        if (MODE == 0) { y[0] = a; y[1] = m2.fast(a); }
        else { y[0] = k; y[1] = a; }
    }
};

// STOCH and STOCHF of the same fastk_period in one walk: the rolling-extrema core (the expensive part) is evaluated once;
// slowk/slowd and fastd keep their own moving averages, so every column is bit-identical to the single function's.
struct StochAllOp {
    static constexpr bool RG_GATHER = true; // a direct call on a ragged batch keeps the per-lane form: alone on the chip it beats re-housing + the tiled body (profiles/r05_bench_ragged.json)
    static constexpr bool LDS_ONLY = true;
    static constexpr int NIN = 3, NOUT = 4; // -> slowk, slowd, fastk, fastd
    static constexpr int ALG_COLS = 5 + 5;  // stoch, stochf
    static constexpr int SEQ_ID = 96;
    static constexpr int COST_NS = 800;
    int64_t fastk, slowk, slowk_mt, slowd, slowd_mt, fastd, fastd_mt;
    FastkCore fk;
    Ma2 mk, md, mf;
    __host__ __device__ int64_t ring_slots() const {
        return FastkCore::slots(fastk) + Ma2::slots(slowk_mt, slowk) + Ma2::slots(slowd_mt, slowd) + Ma2::slots(fastd_mt, fastd);
    }
    __device__ void init(const Row<3> &) {}
    __device__ void init_lds(const Row<3> &r, RingAlloc &ra) {
        fk.init(fastk, ra);
        mk.init(slowk_mt, slowk, r.len, ra); md.init(slowd_mt, slowd, r.len, ra); mf.init(fastd_mt, fastd, r.len, ra);
    }
    __device__ void step(const Row<3> &, int64_t, const double (&)[3], double (&y)[4]) { y[0] = y[1] = y[2] = y[3] = pq_null(); }
    __device__ void step_lds(int64_t, const double (&x)[3], double (&y)[4]) {
        const double k = fk.step(x[0], x[1], x[2]);
        const double a = mk.step(k);
        y[0] = a; y[1] = md.step(a); y[2] = k; y[3] = mf.step(k);
    }
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t) const { return fk.steady() && mk.steady() && md.steady() && mf.steady(); }
    __device__ void step_fast(int64_t, const double (&x)[3], double (&y)[4]) {
        const double k = fk.fast(x[0], x[1], x[2]);
        const double a = mk.fast(k);
        y[0] = a; y[1] = md.fast(a); y[2] = k; y[3] = mf.fast(k);
    }
};

struct StochRsiOp {
    static constexpr bool RG_GATHER = true; // a direct call on a ragged batch keeps the per-lane form: alone on the chip it beats re-housing + the tiled body (profiles/r05_bench_ragged.json)
    static constexpr int64_t DIRECT_LANE_MAX = 16384; // a DIRECT call on a regular batch of up to this many series runs the per-lane form: alone on the chip it is faster (profiles/r05_direct_lane.json)
    static constexpr bool LDS_ONLY = true; // momentum.py:197-205
    static constexpr int NIN = 1, NOUT = 2;
    static constexpr int SEQ_ID = 76;
    static constexpr int COST_NS = 618;
    int64_t p, fastk, fastd, fastd_mt;
    RsiOp rsi;
    FastkCore fk;
    Ma2 m;
    __host__ __device__ int64_t ring_slots() const { return FastkCore::slots(fastk) + Ma2::slots(fastd_mt, fastd); }
    __device__ void init(const Row<1> &) {}
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { rsi.p = p; rsi.init(r); fk.init(fastk, ra); m.init(fastd_mt, fastd, r.len, ra); }
    __device__ void step(const Row<1> &, int64_t, const double (&)[1], double (&y)[2]) { y[0] = y[1] = pq_null(); }
    __device__ void step_lds(int64_t i, const double (&x)[1], double (&y)[2]) {
        double rv[1];
        Row<1> dummy; dummy.in[0] = nullptr; dummy.len = 0;
        rsi.step(dummy, i, x, rv);
        double k = fk.step(rv[0], rv[0], rv[0]);
        y[0] = k;
        y[1] = m.step(k);
    }
    static constexpr bool FAST_NULL_OK = true; // N-B (the RSI core)
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return rsi.steady(t0) && fk.steady() && m.steady(); }
    __device__ void step_fast(int64_t i, const double (&x)[1], double (&y)[2]) {
        double rv[1];
        rsi.step_fast(i, x, rv);
        const double k = fk.fast(rv[0], rv[0], rv[0]);
        y[0] = k;
        y[1] = m.fast(k);
    }
};

struct CciOp {
    static constexpr bool LDS_ONLY = true; // momentum.rs:138-178 in one walk: sma(tp) + mean absolute deviation over the same window (oldest first)
    static constexpr int NIN = 3, NOUT = 1;
    static constexpr int SEQ_ID = 77;
    __host__ int cost_ns() const { return 320 + 40 * (int)(p > 0 ? (p < 1000 ? p : 1000) : 0); } // O(p) mean-deviation loop per row
    int64_t p;
    double sum, denom;
    bool dead;
    Ring w;
    __host__ __device__ int64_t ring_slots() const { return p > 0 ? p : 1; }
    __device__ void init(const Row<3> &) {}
    __device__ void init_lds(const Row<3> &r, RingAlloc &ra) {
        dead = (p <= 0 || r.len < p); sum = 0.0; denom = 1.0 / (double)p; w = ra.make(p);
    }
    __device__ void step(const Row<3> &, int64_t, const double (&)[3], double (&y)[1]) { y[0] = pq_null(); }
    __device__ void step_lds(int64_t i, const double (&x)[3], double (&y)[1]) {
        y[0] = pq_null();
        if (dead) return;
        double tp = (x[0] + x[1] + x[2]) / 3.0;
        sum += tp;
        double old = w.swap(tp);
        if (i < p - 1) return;
        if (i >= p) sum -= old;
        double avg = sum * denom;
        double mean_dev = 0.0;
        for (int b0 = (int)p; b0 >= 1; b0 -= 8) { // oldest (p pushes ago) to newest, in the reference's summation order
            double v8[8];
            w.get8<-1>(b0, v8);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (b0 - u >= 1) mean_dev += fabs(v8[u] - avg);
        }
        if (mean_dev != 0.0) {
            mean_dev /= (double)p;
            y[0] = (tp - avg) / (0.015 * mean_dev);
        }
    }
    static constexpr bool FAST_NULL_OK = true; // N-B
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return !dead && t0 >= p; }
    __device__ void step_fast(int64_t, const double (&x)[3], double (&y)[1]) {
        const double tp = (x[0] + x[1] + x[2]) / 3.0;
        sum += tp;
        sum -= w.swap(tp);
        const double avg = sum * denom;
        double mean_dev = 0.0;
        for (int b0 = (int)p; b0 >= 1; b0 -= 8) { // oldest to newest, the reference's summation order
            double v8[8];
            w.get8<-1>(b0, v8);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (b0 - u >= 1) mean_dev += fabs(v8[u] - avg);
        }
        const double md = mean_dev / (double)p;
        const double v = (tp - avg) / (0.015 * md);
        y[0] = (mean_dev != 0.0) ? v : pq_null();
    }
    // Eight rows at a time.  The windows of eight consecutive rows overlap in all but seven values: W = the seven values the
    // batch pushes out of the ring + the ring after the batch's pushes (p values, oldest first); row r sums W[r .. r+p-1],
    // oldest to newest (the reference's order), so the p + 7 values are read from LDS once per batch instead of p per row and
    // the eight running sums are independent chains.
    static constexpr bool FAST_BATCH = true;
    static constexpr int FAST_UNROLL = 8;
    template <int N>
    __device__ void steps_fast(int64_t, const double (&x)[N][3], double (&y)[N][1]) {
        static_assert(N == 8, "CciOp::steps_fast is written for batches of eight rows");
        double tp[8], old[8], avg[8], md[8];
#pragma unroll
        for (int r = 0; r < 8; r++) tp[r] = (x[r][0] + x[r][1] + x[r][2]) / 3.0;
        w.swap_n<8>(tp, old);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            sum += tp[r];
            sum -= old[r];
            avg[r] = sum * denom;
            md[r] = 0.0;
        }
        double W[15]; // W[b0 .. b0+14] of the concatenated window list
#pragma unroll
        for (int i = 0; i < 7; i++) W[i] = old[i + 1];
        int k = w.pos; // after the pushes: the oldest ring slot
        const int ip = (int)p;
        for (int b0 = 0; b0 < ip; b0 += 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) { // ring entries b0+i (reads beyond the window are masked below)
                W[7 + i] = w.base[k * 64];
                k = (k + 1 == w.depth) ? 0 : k + 1;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) { // window element b0+i of every row: W[r + i]
                if (b0 + i < ip) {
#pragma unroll
                    for (int r = 0; r < 8; r++) md[r] += fabs(W[r + i] - avg[r]);
                }
            }
#pragma unroll
            for (int i = 0; i < 7; i++) W[i] = W[i + 8];
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const double m = md[r] / (double)p;
            const double v = pq_keep((tp[r] - avg[r]) / (0.015 * m));
            y[r][0] = (md[r] != 0.0) ? v : pq_null();
        }
    }
};

// momentum.rs:668-727 calc_dm once for all five of its users: dx, plus_di (= dx, quirk Q-PDI), minus_di, adx, adxr
template <bool ALL> // ALL: the five columns; else adxr alone
struct DmAllOp {
    static constexpr bool LDS_ONLY = true;
    static constexpr int NIN = 3, NOUT = ALL ? 5 : 1;
    static constexpr int ALG_COLS = ALL ? 5 * 4 : 4; // dx, plus_di, minus_di, adx, adxr: five (3 in, 1 out) calls
    static constexpr int SEQ_ID = ALL ? 78 : 80;
    static constexpr int COST_NS = 550;
    int64_t p;
    DmOp<2> core; // carries the three RMAs + the ADX RMA
    Ring wadx;
    __host__ __device__ int64_t ring_slots() const { return p > 1 ? p - 1 : 1; }
    __device__ void init(const Row<3> &) {}
    __device__ void init_lds(const Row<3> &r, RingAlloc &ra) { core.p = p; core.init(r); wadx = ra.make(p - 1); }
    __device__ void step(const Row<3> &, int64_t, const double (&)[3], double (&y)[NOUT]) { for (int k = 0; k < NOUT; k++) y[k] = pq_null(); }
    __device__ void step_lds(int64_t i, const double (&x)[3], double (&y)[NOUT]) {
        double dx, mdi;
        double adx = core.step_all(i, x, dx, mdi);
        if (ALL) { y[0] = dx; y[1 % NOUT] = dx; y[2 % NOUT] = mdi; y[3 % NOUT] = adx; }
        // momentum.rs:50-59: (adx[i] + adx[i-(p-1)]) * 0.5 for i >= p-1
        double adxr = pq_null();
        if (p > 0 && i >= p - 1) {
            double prev = (p == 1) ? adx : wadx.get((int)(p - 1));
            if (!pq_isnull(adx) && !pq_isnull(prev)) adxr = (adx + prev) * 0.5;
        }
        if (p > 1) wadx.push(adx);
        y[NOUT - 1] = adxr;
    }
    static constexpr bool FAST_NULL_OK = true; // N-B
    // the ADX ring holds the last p-1 values: all of them are non-null once 2p rows have passed
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return p > 0 && core.steady(t0) && t0 >= 2 * p; }
    __device__ void step_fast(int64_t, const double (&x)[3], double (&y)[NOUT]) {
        double dx, mdi;
        const double adx = core.fast_all(x, dx, mdi);
        if (ALL) { y[0] = dx; y[1 % NOUT] = dx; y[2 % NOUT] = mdi; y[3 % NOUT] = adx; }
        const double prev = (p == 1) ? adx : wadx.get((int)(p - 1));
        if (p > 1) wadx.push(adx);
        y[NOUT - 1] = (adx + prev) * 0.5;
    }
};

// cycle.rs: the shared Hilbert pipeline once for ht_dcperiod, ht_dcphase, ht_phasor and ht_sine.  The sequential job emits what
// needs the serial walk -- the smoothed period and the phasor components -- and HtPhaseSineOp (a ROW kernel, any order) derives
// dcphase, sine and leadsine from the stored phasor columns: they are pure functions of (inphase, quadrature) (cycle.rs:130-134,
// :294-300).  The six-output form of this job was the last workgroup to finish in a suite step and spent 40 % of its time waiting
// for its storer wave's queue position in the write path (11 600 of 27 300 cycles per 8-row tile, PQ_PROFILE_WAVES); with three
// output columns and one atan less per row it is no longer the step's critical path.
struct HtAllOp {
    static constexpr bool RG_GATHER = true; // a direct call on a ragged batch keeps the per-lane form: alone on the chip it beats re-housing + the tiled body (profiles/r05_bench_ragged.json)
    static constexpr int NIN = 1, NOUT = 3; // dcperiod, inphase, quadrature
    static constexpr int ALG_COLS = 2 + 3;  // ht_dcperiod, ht_phasor (ht_dcphase and ht_sine are credited to HtPhaseSineOp's launch)
    static constexpr int SEQ_ID = 79;
    static constexpr int COST_NS = 800;
    HtOp<0> core;
    __device__ void init(const Row<1> &r) { core.init(r); }
    __host__ __device__ int64_t ring_slots() const { return core.ring_slots(); } // tiled body: the delay lines live in LDS rings (HtOp)
    __device__ void init_lds(const Row<1> &r, RingAlloc &ra) { core.init_lds(r, ra); }
    static constexpr bool FAST_NULL_OK = true;
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return core.steady(t0); }
    __device__ void step_fast(int64_t i, const double (&x)[1], double (&y)[3]) { row<true, true>(i, x, y); }
    __device__ void step_lds(int64_t i, const double (&x)[1], double (&y)[3]) { row<false, true>(i, x, y); }
    __device__ void step(const Row<1> &, int64_t i, const double (&x)[1], double (&y)[3]) { row<false, false>(i, x, y); }
    template <bool FAST, bool RINGS>
    __device__ __forceinline__ void row(int64_t i, const double (&x)[1], double (&y)[3]) {
        double yp[1];
        core.template row<FAST, RINGS>(i, x, yp); // advances the pipeline, emits the smoothed period
        y[0] = yp[0];
        y[1] = y[2] = pq_null();
        if (!FAST && (core.dead || i < 31)) return;
        y[1] = core.i1_0; y[2] = core.q1_0;
    }
};
// dcphase (cycle.rs:130-139), sine and leadsine (cycle.rs:294-300) from the phasor components of the same row.
__device__ __forceinline__ void ht_phase_sine(double i1, double q1, double (&y)[3]) {
    y[0] = y[1] = y[2] = pq_null();
    if (pq_isnull(i1)) return; // a row the pipeline does not emit (series shorter than 32 rows, rows 0..30)
    const double tq = q1 / i1;                                          // the quotient both the phase and the sine take
    double ph = (i1 != 0.0) ? atan(tq) * PQ_RAD2DEG : 0.0;
    double dc_phase = ph + 90.0;
    if (i1 < 0.0) dc_phase += 180.0;
    if (dc_phase > 315.0) dc_phase -= 360.0;
    y[0] = dc_phase;
    // sine / leadsine without a device sin: with phi = atan(t), sin(phi) = t / sqrt(1 + t^2) and cos(phi) = 1 / sqrt(1 + t^2)
    // (phi in (-pi/2, pi/2): cos >= 0), and sin(phi + pi/4) = (sin + cos) / sqrt(2).  The reference takes sin() of phi after a
    // degrees round trip (two roundings, ~2e-16 relative on the argument); this form is within a few ulp of the exact value: far
    // inside the 1e-12 budget of these outputs (the single-output pq_ht_sine keeps the reference's two sin calls).
    double sn = 0.0, cs = 1.0;
    if (i1 != 0.0) {
        const double rr = 1.0 / sqrt(1.0 + tq * tq);
        sn = tq * rr; cs = rr;
        if (fabs(tq) >= 1e150) { sn = copysign(1.0, tq); cs = 0.0; }     // t^2 overflows: phi = +-pi/2 (false for a NaN t)
    }
    y[1] = sn;
    y[2] = (sn + cs) * 0.70710678118654752440;
}
struct HtPhaseSineOp {
    static constexpr int NIN = 2, NOUT = 3; // inphase, quadrature -> dcphase, sine, leadsine
    typedef double OutT;
    __device__ void eval(const Row<2> &r, int64_t t, double (&y)[3]) { ht_phase_sine(r.in[0][t], r.in[1][t], y); }
};
// HtAllOp + the three derived columns in ONE job (tiled body): the storer wave computes dcphase / sine / leadsine from the phasor rows
// it holds (NDer, pq_dev.h) -- the values HtPhaseSineOp would read back from memory, through the same function.  Saves that launch, which
// could only start when the whole grid of its producer had drained (the last 0.1 ms of a suite step), and 0.2 GB of reads.
struct HtAll6Op : HtAllOp {
    static constexpr bool LDS_ONLY = true;
    static constexpr int SEQ_ID = 85;
    static constexpr int ALG_COLS = 2 + 3 + 2 + 3; // ht_dcperiod, ht_phasor, ht_dcphase, ht_sine
    static constexpr int NDER = 3;
    double *der[3]; // dcphase, sine, leadsine
    // Recorded for a SMALL shard the job is split in time (pq_ht_all, fused.hip): the Hilbert pipeline forgets its start -- a walk that
    // begins 640 rows early agrees with the full walk to a few ulp (4e-15 of the value over 2 000 series x 2 starts, the same size as the
    // device atan's own difference from the host's), and its outputs are tolerance-class (<= 1e-12) to begin with
    static constexpr bool TS_OK = true;
    double *chk[3]; // the check columns of a time-split job (dcperiod, inphase, quadrature of its last warm-up tile), else unused
    __device__ void ts_shift(int64_t row0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { der[k] += row0; chk[k] = chk[k] ? chk[k] + row0 : nullptr; }
    }
    __device__ static void derive(const double (&y)[3], double (&z)[3]) { ht_phase_sine(y[1], y[2], z); }
};

// MAVP with the SMA core (matype 0 / 7 / other): a job advances SIXTEEN candidate periods [lo, hi] in one walk.  Every
// candidate's running sum lives in registers and is the exact add-new / subtract-old sequence of overlap.rs:897-910 (so the
// rounding of every candidate is the reference's); all of them are fed from one shared input ring, and the job writes the
// rows whose clamped period falls in its range (masked, like the block op below, which keeps its states in LDS and needs
// twice the jobs).
template <int NG> // NG groups of eight candidate periods per job: 2 (sixteen) on a full chip; 1 (eight) for SMALL shards, where a job's length counts
struct MavpSmaNOp {
    static constexpr int NC = 8 * NG;
    static constexpr bool LDS_ONLY = true;
    static constexpr bool MASKED = true;
    static constexpr int NIN = 2, NOUT = 1; // real (nulls -> 0.0), periods
    static constexpr int SEQ_ID = NG == 2 ? 83 : 86;
    static constexpr int COST_NS = NG == 2 ? 835 : 470; // NC running sums per row
    int lo, hi, minp, maxp, n;
    Ring w;
    const double *tab; // 1/P for P = lo .. lo+NC-1 (shared by the wave)
    double s[NC];
    // the ring keeps lo + NC + 7 values: the batched fast path pushes eight rows first and then reads the NC + 7 values
    // x[t-lo-NC+1 .. t+7-lo] the NC candidates need for those rows (the oldest was pushed lo + NC + 7 pushes ago)
    __host__ __device__ int64_t ring_slots() const { return (lo > 0 ? lo : 1) + NC + 7 + 1; }
    __device__ void init(const Row<2> &) {}
    __device__ void init_lds(const Row<2> &r, RingAlloc &ra) {
        n = (int)(r.len < 0x7fffffff ? r.len : 0x7fffffff);
        w = ra.make((lo > 0 ? lo : 1) + NC + 7);
        double *t = ra.make_shared(NC);
        const int k = threadIdx.x & 63;
        if (k < NC) t[k] = 1.0 / (double)(lo + k);
        tab = t;
#pragma unroll
        for (int u = 0; u < NC; u++) s[u] = 0.0;
        lds_fence();
    }
    __device__ void step(const Row<2> &, int64_t, const double (&)[2], double (&y)[1]) { y[0] = pq_skip(); }
    __device__ void step_lds(int64_t t64, const double (&x)[2], double (&y)[1]) {
        const double v = n0(x[0]);
        const int t = (int)t64;
        const double pd = n0(x[1]);
        const int64_t p64 = (int64_t)pd;
        const int pi = p64 < minp ? minp : (p64 > maxp ? maxp : (int)p64);
        const int c = t + 1; // every row is valid after nulls -> 0.0
        double asel = 0.0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            double old[8];
            w.get8<1>(lo + 8 * g, old); // x[t - P] for the eight periods of this group
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int P = lo + 8 * g + u;
                double a = s[8 * g + u] + v;
                const double b = a - old[u];
                a = (c > P) ? b : a; // wave-uniform condition
                a = (P <= hi && P > 0 && n >= P) ? a : 0.0;
                s[8 * g + u] = a;
                asel = (P == pi) ? a : asel;
            }
        }
        w.push(v);
        const bool mine = pi >= lo && pi <= hi;
        const bool ok = pi > 0 && n >= pi && c >= pi && t >= maxp - 1;
        const double res = asel * tab[mine ? pi - lo : 0];
        y[0] = mine ? (ok ? res : pq_null()) : pq_skip();
    }
    static constexpr bool FAST_NULL_OK = true; // N-0: nulls become 0.0 in the row body
    static constexpr int FAST_UNROLL = 8;
    // every candidate window is full (c = t+1 > lo+15) and the output gate t >= maxp-1 is open; candidates beyond `hi` or
    // longer than the series carry garbage here that is never selected (pi is clamped into [minp, maxp] and `mine` tests
    // [lo, hi]) and that the general path resets to 0.0 on its next row
    static constexpr bool HAS_FAST = true;
    __device__ bool steady(int64_t t0) const { return t0 >= lo + NC && t0 >= maxp - 1 && lo > 0; }
    __device__ void step_fast(int64_t, const double (&x)[2], double (&y)[1]) {
        const double v = n0(x[0]);
        const int64_t p64 = (int64_t)n0(x[1]);
        const int pi = p64 < minp ? minp : (p64 > maxp ? maxp : (int)p64);
        double asel = 0.0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            double old[8];
            w.get8<1>(lo + 8 * g, old);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const double a = (s[8 * g + u] + v) - old[u];
                s[8 * g + u] = a;
                asel = (lo + 8 * g + u == pi) ? a : asel;
            }
        }
        w.push(v);
        const bool mine = pi >= lo && pi <= hi;
        const bool ok = pi > 0 && n >= pi;
        const double res = asel * tab[mine ? pi - lo : 0];
        y[0] = mine ? (ok ? res : pq_null()) : pq_skip();
    }
    // Eight rows at a time: candidate P = lo+u subtracts x[t+r-P] at row t+r, so rows r = 0..7 and u = 0..15 touch only the
    // 23 values X[i] = x[t - lo - 15 + i], i = r - u + 15: one LDS read each per batch (instead of 16 per row), and the
    // sixteen running sums are independent chains of plain register arithmetic.
    static constexpr bool FAST_BATCH = true;
    template <int N>
    __device__ void steps_fast(int64_t, const double (&x)[N][2], double (&y)[N][1]) {
        static_assert(N == 8, "MavpSmaNOp::steps_fast is written for batches of eight rows");
        double v[8];
        int pi[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            v[r] = n0(x[r][0]);
            const int64_t p64 = (int64_t)n0(x[r][1]);
            pi[r] = p64 < minp ? minp : (p64 > maxp ? maxp : (int)p64);
        }
        w.push_n<8>(v);
        double X[NC + 7];
        {   // X[i] was pushed (NC + 7 + lo - i) pushes ago (X[NC + 6 + lo] would be the newest, v[7])
            int k = w.pos - (NC + 7 + lo);
            k += (k < 0) ? w.depth : 0;
#pragma unroll
            for (int i = 0; i < NC + 7; i++) {
                X[i] = w.base[k * 64];
                k = (k + 1 == w.depth) ? 0 : k + 1;
            }
        }
        double asel[8];
#pragma unroll
        for (int r = 0; r < 8; r++) asel[r] = 0.0;
#pragma unroll
        for (int r = 0; r < 8; r++) { // row-major: sixteen independent chains advance together
#pragma unroll
            for (int u = 0; u < NC; u++) {
                s[u] = (s[u] + v[r]) - X[r - u + NC - 1];
                asel[r] = (lo + u == pi[r]) ? s[u] : asel[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const bool mine = pi[r] >= lo && pi[r] <= hi;
            const bool ok = pi[r] > 0 && n >= pi[r];
            const double res = asel[r] * tab[mine ? pi[r] - lo : 0];
            y[r][0] = mine ? (ok ? res : pq_null()) : pq_skip();
        }
    }
};
typedef MavpSmaNOp<2> MavpSma16Op;
typedef MavpSmaNOp<1> MavpSma8Op;

// MAVP with the SMA core, up to THIRTY-TWO candidate periods [lo, hi] = [minperiod, maxperiod] in ONE job (the default 2 .. 30 is
// 29): the job covers every period a row can ask for, so its output is an ordinary tile column -- coalesced 128-byte pieces through
// the storer wave -- instead of the row-masked 8-byte stores two 16-candidate jobs need to share a column, and the inputs are read
// once.  The candidates run as two halves of sixteen register chains per 8-row batch (same arithmetic and order per candidate as
// MavpSma16Op: overlap.rs:897-910); the upper half's running sums are parked in LDS between batches ([16][lane] doubles), the lower
// half's stay in registers.
struct MavpSma32Op {
    static constexpr bool LDS_ONLY = true;
    static constexpr int NIN = 2, NOUT = 1; // real (nulls -> 0.0), periods
    static constexpr int SEQ_ID = 84;
    static constexpr int COST_NS = 600;
    int lo, hi, minp, maxp, n;
    Ring w;
    double *sb;        // parked sums of the candidates lo+16 .. lo+31, lane-offset: sb[u * 64]
    const double *tab; // 1/P for P = lo .. lo+31 (shared by the wave)
    double s[16];
    // ring: the batched fast path pushes four rows first and then reads x[t-lo-31 .. t+3-lo] (the oldest was pushed lo + 35 pushes ago)
    __host__ __device__ int64_t ring_slots() const { return (lo > 0 ? lo : 1) + 35 + 16 + 1; }
    __device__ void init(const Row<2> &) {}
    __device__ void init_lds(const Row<2> &r, RingAlloc &ra) {
        n = (int)(r.len < 0x7fffffff ? r.len : 0x7fffffff);
        w = ra.make((lo > 0 ? lo : 1) + 35);
        Ring park = ra.make(16);
        sb = park.base;
        double *t = ra.make_shared(32);
        const int k = threadIdx.x & 63;
        if (k < 32) t[k] = 1.0 / (double)(lo + k);
        tab = t;
#pragma unroll
        for (int u = 0; u < 16; u++) { s[u] = 0.0; sb[u * 64] = 0.0; }
        lds_fence();
    }
    __device__ void step(const Row<2> &, int64_t, const double (&)[2], double (&y)[1]) { y[0] = pq_null(); }
    __device__ void step_lds(int64_t t64, const double (&x)[2], double (&y)[1]) { // warm-up rows: every gate of the definition
        const double v = n0(x[0]);
        const int t = (int)t64;
        const int64_t p64 = (int64_t)n0(x[1]);
        const int pi = p64 < minp ? minp : (p64 > maxp ? maxp : (int)p64);
        const int c = t + 1; // every row is valid after nulls -> 0.0
        double asel = 0.0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            double old[8];
            w.get8<1>(lo + 8 * g, old); // x[t - P] for the eight periods of this group
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int P = lo + 8 * g + u;
                double a = (g < 2 ? s[(8 * g + u) & 15] : sb[(8 * (g - 2) + u) * 64]) + v;
                const double b = a - old[u];
                a = (c > P) ? b : a; // wave-uniform condition.  (A candidate beyond `hi`, of period 0 or longer than the series carries
                                     // garbage: never selected -- pi is clamped into [lo, hi] and `ok` tests pi > 0 and n >= pi.)
                if (g < 2) s[(8 * g + u) & 15] = a; else sb[(8 * (g - 2) + u) * 64] = a;
                asel = (P == pi) ? a : asel;
            }
        }
        w.push(v);
        const bool ok = pi > 0 && n >= pi && c >= pi && t >= maxp - 1;
        y[0] = ok ? asel * tab[pi - lo] : pq_null();
    }
    static constexpr bool FAST_NULL_OK = true; // N-0: nulls become 0.0 in the row body
    static constexpr int FAST_UNROLL = 4;
    // every candidate window is full (c = t+1 > lo+31) and the output gate t >= maxp-1 is open; candidates beyond `hi` or longer
    // than the series carry garbage here that is never selected (pi is clamped into [minp, maxp] = [lo, hi])
    static constexpr bool HAS_FAST = true;
    static constexpr bool FAST_BATCH = true;
    __device__ bool steady(int64_t t0) const { return t0 >= lo + 32 && t0 >= maxp - 1 && lo > 0; }
    __device__ void step_fast(int64_t t, const double (&x)[2], double (&y)[1]) { step_lds(t, x, y); }
    // Four rows at a time (see MavpSma16Op::steps_fast for the scheme): half H's candidate P = lo+16H+u subtracts x[t+r-P] at row
    // t+r, i.e. rows r = 0..3 touch the 19 values X[i] = x[t - lo - 16H - 15 + i], i = r - u + 15.  (Four rows instead of eight: 19
    // window values, 4 selections and 4 staged input rows live instead of 23, 8 and 8 -- the op fits the 192-VGPR job kernel.)
    template <int H>
    __device__ __forceinline__ void half(const double (&v)[4], const int (&pi)[4], double (&sum)[16], double (&asel)[4]) {
        double X[19];
        int k = w.pos - (19 + lo + 16 * H); // X[i] was pushed (19 + lo + 16H - i) pushes ago (the newest is v[3])
        k += (k < 0) ? w.depth : 0;
#pragma unroll
        for (int i = 0; i < 19; i++) {
            X[i] = w.base[k * 64];
            k = (k + 1 == w.depth) ? 0 : k + 1;
        }
        int d[4]; // the row's candidate index within this half (outside 0..15: the other half has it)
#pragma unroll
        for (int r = 0; r < 4; r++) d[r] = pi[r] - (lo + 16 * H);
#pragma unroll
        for (int r = 0; r < 4; r++) // row-major: sixteen independent chains advance together
#pragma unroll
            for (int u = 0; u < 16; u++) {
                sum[u] = (sum[u] + v[r]) - X[r - u + 15];
                sel_eq(asel[r], u, d[r], sum[u]);
            }
    }
    // dst = (d == U) ? src : dst with the compare in VCC, consumed at once.  Written as `cond ? a : b` the 64 compares of a batch are
    // scheduled ahead of their selects and their lane masks (an SGPR pair each) no longer fit the scalar file: 233 of them were
    // spilled to VGPR lanes and read back one by one.
    __device__ static __forceinline__ void sel_eq(double &dst, int U, int d, double src) {
        int dl = __double2loint(dst), dh = __double2hiint(dst);
        const int sl = __double2loint(src), sh = __double2hiint(src);
        asm("v_cmp_eq_u32 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc"
            : "+v"(dl), "+v"(dh) : "n"(U), "v"(d), "v"(sl), "v"(sh) : "vcc");
        dst = __hiloint2double(dh, dl);
    }
    template <int N>
    __device__ void steps_fast(int64_t, const double (&x)[N][2], double (&y)[N][1]) {
        static_assert(N == 4, "MavpSma32Op::steps_fast is written for batches of four rows");
        double v[4], asel[4];
        int pi[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = n0(x[r][0]);
            const int64_t p64 = (int64_t)n0(x[r][1]);
            pi[r] = p64 < minp ? minp : (p64 > maxp ? maxp : (int)p64);
            asel[r] = 0.0;
        }
        w.push_n<4>(v);
        if (hi >= lo + 16) { // wave-uniform: the upper half exists
            double sB[16];
#pragma unroll
            for (int u = 0; u < 16; u++) sB[u] = sb[u * 64];
            half<1>(v, pi, sB, asel);
#pragma unroll
            for (int u = 0; u < 16; u++) sb[u * 64] = sB[u];
        }
        half<0>(v, pi, s, asel);
#pragma unroll
        for (int r = 0; r < 4; r++) y[r][0] = (n >= pi[r]) ? asel[r] * tab[pi[r] - lo] : pq_null();
    }
};

// MAVP for the single-core MA types (SMA: matype 0/7/other, EMA: matype 1): a job advances up to EIGHT candidate
// periods [lo, hi] together from one shared input ring (their running states live in an LDS array [period][lane]) and
// writes the rows whose clamped period falls in its range.  Same arithmetic per period as MavpSelOp<SmaOp/EmaOp>; the
// inputs are read once per 8 periods instead of once per period.
template <int KIND>
struct MavpBlockOp {
    static constexpr bool LDS_ONLY = true;
    static constexpr bool MASKED = true;
    static constexpr int NIN = 2, NOUT = 1; // real (nulls -> 0.0), periods
    static constexpr int SEQ_ID = 81 + KIND;
    static constexpr int COST_NS = 800;
    int lo, hi, minp, maxp, n;
    Ring w, st;
    const double *tab; // SMA: 1/P ; EMA: 2/(P+1)
    __host__ __device__ int64_t ring_slots() const { return (KIND == 0 ? (hi > 0 ? hi : 1) : 0) + 8 + 1; }
    __device__ void init(const Row<2> &) {}
    __device__ void init_lds(const Row<2> &r, RingAlloc &ra) {
        n = (int)(r.len < 0x7fffffff ? r.len : 0x7fffffff);
        if (KIND == 0) w = ra.make(hi);
        st = ra.make(8);
        double *t = ra.make_shared(8);
        int k = threadIdx.x & 63;
        if (k < 8) {
            double P = (double)(lo + k);
            t[k] = (KIND == 0) ? 1.0 / P : 2.0 / (P + 1.0);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) st.base[u * 64] = 0.0;
        tab = t;
        lds_fence();
    }
    __device__ void step(const Row<2> &, int64_t, const double (&)[2], double (&y)[1]) { y[0] = pq_skip(); }
    __device__ void step_lds(int64_t t64, const double (&x)[2], double (&y)[1]) {
        const double v = n0(x[0]);
        const int t = (int)t64;
        double pd = n0(x[1]);
        int64_t p64 = (int64_t)pd;
        int pi = p64 < minp ? minp : (p64 > maxp ? maxp : (int)p64);
        const int c = t + 1; // every row is valid after nulls -> 0.0
        double s[8], old[8], tb[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { s[u] = st.base[u * 64]; tb[u] = tab[u]; }
        if (KIND == 0) w.get8<1>(lo, old); // x[t - P] for P = lo .. lo+7
        double sel = pq_skip();
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int P = lo + u;
            double res = pq_null();
            double a = s[u];
            if (P <= hi && P > 0 && n >= P) {
                if (KIND == 0) { // overlap.rs:897-910
                    a += v;
                    if (c > P) a -= old[u];
                    if (c >= P) res = a * tb[u];
                } else {         // overlap.rs:687-700
                    if (c < P) a += v;
                    else if (c == P) { a += v; a = a / (double)P; res = a; }
                    else { a = fma(tb[u], v - a, a); res = a; }
                }
            }
            st.base[u * 64] = a;
            if (P == pi && P <= hi) sel = (t >= maxp - 1) ? res : pq_null();
        }
        if (KIND == 0) w.push(v);
        y[0] = sel;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Two SEQ ops over the SAME input columns as one job: one in-tile, one walk, the outputs side by side.  The sub-ops keep
// their own state machines (bit-identical results); what is shared is the tile traffic, the per-tile hand-off and the
// workgroup -- the expensive parts for short jobs.  Nests: Fuse2<ID, Fuse2<0, A, B>, Fuse2<0, C, D>>.
template <int ID, class A, class B>
struct Fuse2 {
    static_assert(A::NIN == B::NIN, "fused ops must read the same input columns");
    static_assert(!IsMasked<A>::value && !IsMasked<B>::value && !HasFinish<A>::value && !HasFinish<B>::value, "not fusable");
    static_assert(NTap<A>::value == 0 || HasRings<A>::value, "tap-only ops cannot be fused");
    static_assert(NTap<B>::value == 0 || HasRings<B>::value, "tap-only ops cannot be fused");
    static constexpr int NIN = A::NIN, NOUT = A::NOUT + B::NOUT;
    static constexpr int ALG_COLS = AlgCols<A>::value + AlgCols<B>::value; // the calls it replaces
    static constexpr int SEQ_ID = ID;
    __host__ int cost_ns() const { return (3 * (OpCost<A>::get(a) + OpCost<B>::get(b))) / 4; } // one walk, shared tile traffic
    static constexpr bool HEAVY = IsHeavy<A>::value || IsHeavy<B>::value;
    static constexpr bool LDS_ONLY = true; // (the gather driver's tap plumbing is per op)
    A a;
    B b;
    __host__ __device__ int64_t ring_slots() const {
        int64_t n = 0;
        if constexpr (HasRings<A>::value) n += a.ring_slots();
        if constexpr (HasRings<B>::value) n += b.ring_slots();
        return n;
    }
    __device__ void init(const Row<NIN> &) {}
    __device__ void init_lds(const Row<NIN> &r, RingAlloc &ra) {
        if constexpr (HasRings<A>::value) a.init_lds(r, ra); else a.init(r);
        if constexpr (HasRings<B>::value) b.init_lds(r, ra); else b.init(r);
        row = r;
    }
    Row<NIN> row; // the ring-free sub-ops' step() takes it
    __device__ void step(const Row<NIN> &, int64_t, const double (&)[NIN], double (&y)[NOUT]) {
#pragma unroll
        for (int k = 0; k < NOUT; k++) y[k] = pq_null();
    }
    __device__ void step_lds(int64_t t, const double (&x)[NIN], double (&y)[NOUT]) {
        double ya[A::NOUT], yb[B::NOUT];
        if constexpr (HasRings<A>::value) a.step_lds(t, x, ya); else a.step(row, t, x, ya);
        if constexpr (HasRings<B>::value) b.step_lds(t, x, yb); else b.step(row, t, x, yb);
#pragma unroll
        for (int k = 0; k < A::NOUT; k++) y[k] = ya[k];
#pragma unroll
        for (int k = 0; k < B::NOUT; k++) y[A::NOUT + k] = yb[k];
    }
    static constexpr bool FAST_NULL_OK = FastNullOk<A>::value && FastNullOk<B>::value;
    static constexpr int FAST_UNROLL = FastUnroll<A>::value < FastUnroll<B>::value ? FastUnroll<A>::value : FastUnroll<B>::value;
    static constexpr bool HAS_FAST = HasFast<A>::value && HasFast<B>::value; // steady / step_fast are instantiated only then
    __device__ bool steady(int64_t t0) const { return a.steady(t0) && b.steady(t0); }
    __device__ void step_fast(int64_t t, const double (&x)[NIN], double (&y)[NOUT]) {
        double ya[A::NOUT], yb[B::NOUT];
        a.step_fast(t, x, ya);
        b.step_fast(t, x, yb);
#pragma unroll
        for (int k = 0; k < A::NOUT; k++) y[k] = ya[k];
#pragma unroll
        for (int k = 0; k < B::NOUT; k++) y[A::NOUT + k] = yb[k];
    }
};

// A ring-free op that reads a SUBSET (I0, I1) of a wider job's input columns
template <int NIN_, class Op, int I0, int I1>
struct Pick2 {
    static_assert(Op::NIN == 2 && !HasRings<Op>::value && NTap<Op>::value == 0, "Pick2 wraps a plain two-input op");
    static constexpr int NIN = NIN_, NOUT = Op::NOUT;
    static constexpr int ALG_COLS = AlgCols<Op>::value;
    static constexpr int COST_NS = OpCostStatic<Op>::value;
    Op op;
    Row<2> rr;
    __device__ void init(const Row<NIN> &r) { rr.in[0] = r.in[I0]; rr.in[1] = r.in[I1]; rr.len = r.len; op.init(rr); }
    __device__ void step(const Row<NIN> &, int64_t t, const double (&x)[NIN], double (&y)[NOUT]) {
        const double xx[2] = {x[I0], x[I1]};
        op.step(rr, t, xx, y);
    }
    static constexpr bool FAST_NULL_OK = FastNullOk<Op>::value;
    static constexpr bool HAS_FAST = HasFast<Op>::value;
    __device__ bool steady(int64_t t0) const { return op.steady(t0); }
    __device__ void step_fast(int64_t t, const double (&x)[NIN], double (&y)[NOUT]) {
        const double xx[2] = {x[I0], x[I1]};
        op.step_fast(t, xx, y);
    }
};

typedef Fuse2<90, Fuse2<0, EmaOp, DemaOp>, Fuse2<0, TemaOp, TrixOp>> EmaAllOp; // ema, dema, tema, trix of one timeperiod
typedef Fuse2<91, AtrOp<false>, AtrOp<true>> AtrAllOp;                          // atr, natr
typedef Fuse2<92, DmRawOp<true>, DmRawOp<false>> DmPairOp;                      // plus_dm, minus_dm
typedef Fuse2<93, AdOp<false>, AdOp<true>> AdAllOp;                             // ad, adosc
typedef Fuse2<94, MacdOp, MacdOp> MacdPairOp;                                   // macd, macdfix
typedef Fuse2<95, MaDiffOp<0>, MaDiffOp<1>> ApoPpoOp;                           // apo, ppo
typedef Fuse2<97, SarextOp, SarextOp> SarPairOp;                                // sar, sarext
typedef Fuse2<99, DmAllOp<true>, AtrAllOp> DmiAtrOp;                             // dx, +di, -di, adx, adxr, atr, natr
typedef Fuse2<89, CmoOp, RsiOp> CmoRsiOp;                                       // cmo, rsi
typedef Fuse2<98, MfiOp, Fuse2<0, AdAllOp, Pick2<4, ObvOp, 2, 3>>> VolumeAllOp; // mfi, ad, adosc, obv: 4 in / 4 out, MFI's LDS
