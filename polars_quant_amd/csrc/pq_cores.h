// pq_cores.h -- per-lane streaming state machines shared by the SEQ kernels.
// Each core advances one series by one row in the reference's exact operation order.
#pragma once
#include "pq_dev.h"

// true when the condition holds on every active lane: lets a state machine take its steady-state path without any
// per-lane branching (the warm-up / null paths below stay the general case)
__device__ __forceinline__ bool wave_all(bool x) { return __builtin_amdgcn_ballot_w64(x) == __builtin_amdgcn_ballot_w64(true); }
// A constant derived from a wave-uniform parameter (2/(p+1), 1/p, (double)p ...) is computed by the vector ALU -- there is no scalar
// f64 unit -- and would occupy a VGPR pair for the whole walk; moved to scalar registers it costs none (same bits in every lane).
__device__ __forceinline__ double pq_uniform(double x) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

// overlap.rs:660-730 calc_ema: null-transparent (N-A), SMA seed at count == p, then
// alpha.mul_add(x - ema, ema).
struct EmaCore {
    int64_t p, count;
    double alpha, ema, sum;
    bool dead;
    __device__ void init(int64_t p_, int64_t n) {
        p = p_;
        dead = (p <= 0 || n < p);
        alpha = pq_uniform(2.0 / ((double)p + 1.0));
        count = 0;
        ema = 0.0;
        sum = 0.0;
    }
    __device__ double step(double v) {
        if (wave_all(!dead && count >= p && !pq_isnull(v))) { // steady state on the whole wave (count only matters relative to p)
            ema = fma(alpha, v - ema, ema);
            return ema;
        }
        if (dead || pq_isnull(v)) return pq_null();
        count += 1;
        if (count < p) {
            sum += v;
            return pq_null();
        } else if (count == p) {
            sum += v;
            ema = sum / (double)p;
            return ema;
        }
        ema = fma(alpha, v - ema, ema);
        return ema;
    }
    // steady state (see HasFast in pq_dev.h): seeded, so every valid row is the recurrence; count is only compared with p
    __device__ bool steady() const { return !dead && count >= p; }
    __device__ double fast(double v) {
        ema = fma(alpha, v - ema, ema);
        return ema;
    }
};

// Walks a series forward over its valid (non-null) rows: used to pop "the oldest value still in
// the window" when nulls are skipped (VecDeque::pop_front in the reference).
struct ValidCursor {
    int64_t idx;
    __device__ void start(int64_t i) { idx = i; }
    __device__ double pop(const double *col) {
        double v = col[idx];
        do { idx++; } while (pq_isnull(col[idx])); // the current (valid) row bounds the walk
        return v;
    }
};

// Tracks whether every row since the first valid one has been valid.  While that holds ("regular"), the
// oldest value of a p-window is simply row t-p and arrives through a prefetched lag tap; after the first
// interior null the window is walked with ValidCursors instead (exact either way).
struct Regular {
    int64_t first, seen;
    __device__ void init() { first = -1; seen = 0; }
    __device__ bool push(int64_t t) { // call once per valid row; returns regular?
        if (first < 0) first = t;
        seen += 1;
        return (t - first + 1) == seen;
    }
};

// overlap.rs:871-937 calc_sma: running sum, +new then (count > p) -old, out = sum * (1/p).
struct SmaCore {
    int64_t p, count;
    double denom, sum;
    bool dead;
    Regular reg;
    ValidCursor tail;
    __device__ void init(int64_t p_, int64_t n) {
        p = p_;
        dead = (p <= 0 || n < p);
        denom = pq_uniform(1.0 / (double)p);
        count = 0;
        sum = 0.0;
        reg.init();
    }
    // tap = col[t - p] (prefetched)
    __device__ double step(const double *col, int64_t t, double v, double tap) {
        if (dead || pq_isnull(v)) return pq_null();
        if (reg.first < 0) tail.start(t);
        bool regular = reg.push(t);
        count += 1;
        sum += v;
        if (count < p) return pq_null();
        if (count > p) {
            double old;
            if (regular) { old = tap; tail.idx = t - p + 1; } else old = tail.pop(col);
            sum -= old;
            count -= 1;
        }
        return sum * denom;
    }
    // shared-ring variant: `old` = the valid value pushed p pushes ago, read by the caller (Ring::get(p) before its push)
    __device__ double step_old(double v, double old) {
        if (wave_all(!dead && count >= p && !pq_isnull(v))) { // full window on the whole wave: count stays at p
            sum += v;
            sum -= old;
            return sum * denom;
        }
        if (dead || pq_isnull(v)) return pq_null();
        count += 1;
        sum += v;
        if (count < p) return pq_null();
        if (count > p) {
            sum -= old;
            count -= 1;
        }
        return sum * denom;
    }
    // LDS-ring variant: the ring holds the last p valid values, so the popped value is ring.swap(v)
    __device__ double step_ring(Ring &w, double v) {
        if (wave_all(!dead && count >= p && !pq_isnull(v))) { // full window on the whole wave: count stays at p
            sum += v;
            sum -= w.swap(v);
            return sum * denom;
        }
        if (dead || pq_isnull(v)) return pq_null();
        count += 1;
        sum += v;
        double old = w.swap(v);
        if (count < p) return pq_null();
        if (count > p) {
            sum -= old;
            count -= 1;
        }
        return sum * denom;
    }
    __device__ bool steady() const { return !dead && count >= p; } // full window: count stays at p
    __device__ double fast_ring(Ring &w, double v) {
        sum += v;
        sum -= w.swap(v);
        return sum * denom;
    }
    template <int N>
    __device__ void fast_ring_n(Ring &w, const double (&v)[N], double (&out)[N]) {
        double old[N];
        w.swap_n<N>(v, old);
#pragma unroll
        for (int u = 0; u < N; u++) {
            sum += v[u];
            sum -= old[u];
            out[u] = sum * denom;
        }
    }
    __device__ double fast_old(double v, double old) {
        sum += v;
        sum -= old;
        return sum * denom;
    }
};

// D-1 calc_rma (Wilder) over a null-free slice: None for i < p-1, seed = mean(x[0..p)) at i = p-1,
// then (prev*(p-1) + x) / p.
struct RmaCore {
    int64_t p;
    double sum, r, pm1, pf;
    bool dead;
    __device__ void init(int64_t p_, int64_t n) {
        p = p_;
        dead = (p <= 0 || n < p);
        sum = 0.0;
        r = 0.0;
        pm1 = pq_uniform((double)p - 1.0);
        pf = pq_uniform((double)p);
    }
    __device__ double step(int64_t i, double x) {
        if (wave_all(!dead && i >= p)) { // steady state on the whole wave
            r = (r * pm1 + x) / pf;
            return r;
        }
        if (dead) return pq_null();
        if (i < p - 1) {
            sum += x;
            return pq_null();
        } else if (i == p - 1) {
            sum += x;
            r = sum / pf;
            return r;
        }
        r = (r * pm1 + x) / pf;
        return r;
    }
    __device__ bool steady(int64_t i) const { return !dead && i >= p; }
    __device__ double fast(double x) {
        r = (r * pm1 + x) / pf;
        return r;
    }
};

// Rolling extremum over the last p valid values with a lazy rescan when the extremum expires.
// Value-equivalent to the reference's monotonic deques (overlap.rs:205-218, :330-343, :383-396);
// `count - p` wraps for count < p, i.e. nothing expires until the window is full.
// The front of the reference's monotonic deque (overlap.rs:203-217 / :325-345 / :378-398) when a NaN VALUE (not a NULL) is among the last
// p valid values.  The deque pops its back while `back <= value` (`>=` for the minimum), which no NaN satisfies in either role: a NaN is
// never popped from the back and nothing older can be popped past it, so the front is the extremum of the values OLDER than the oldest
// NaN of the window -- the NaN itself once nothing older is left -- and everything newer is hidden until that NaN expires.  Without a
// NaN in the window this is the window's extremum (the newest of equal ones, as `<=` / `>=` keep it).  One backward pass, newest first.
template <bool IS_MAX>
__device__ __forceinline__ double roll_ext_ref(const double *col, int64_t t, int64_t p) {
    double run = 0.0;
    bool have = false;
    int64_t cnt = 0;
    for (int64_t i = t; i >= 0 && cnt < p; i--) {
        const double w = col[i];
        if (pq_isnull(w)) continue;
        cnt++;
        if (w != w) have = false;
        else if (!have || (IS_MAX ? w > run : w < run)) { run = w; have = true; }
    }
    return have ? run : __longlong_as_double(0x7FF8000000000000LL);
}
template <bool IS_MAX>
struct RollExt {
    int64_t p, j;      // j = number of valid values seen (1-based index of the newest)
    double best;
    int64_t best_j;    // 1-based valid-index of the current extremum
    Regular reg;
    int64_t nan_left;  // valid rows for which a NaN value is still among the last p (the structures below ignore NaNs: see nan_row)
    __device__ void init(int64_t p_) {
        p = p_;
        j = 0;
        best = 0.0;
        best_j = 0;
        nan_left = 0;
        reg.init();
    }
    // after step() / step_ring2() on the valid value v of row t: the reference's value while a NaN sits in the window (roll_ext_ref)
    __device__ __forceinline__ double nan_row(const double *col, int64_t t, double v, double m) {
        if (v != v) nan_left = p;
        if (nan_left > 0) { nan_left -= 1; m = roll_ext_ref<IS_MAX>(col, t, p); }
        return m;
    }
    __device__ static bool beats(double a, double b) { return IS_MAX ? (a >= b) : (a <= b); }
    __device__ double step(const double *col, int64_t t, double v) {
        bool regular = reg.push(t);
        j += 1;
        if (j == 1 || beats(v, best)) {
            best = v;
            best_j = j;
        } else if (p > 0 && best_j == j - p) { // expired: rescan the last p valid values, newest first
            best = v;
            best_j = j;
            if (regular) { // rows t-1 .. t-p+1, independent loads in batches of 8
                int64_t k = 1;
                for (; k + 8 <= p; k += 8) {
                    double w[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) w[u] = col[t - k - u];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (IS_MAX ? (w[u] > best) : (w[u] < best)) { best = w[u]; best_j = j - k - u; }
                }
                for (; k < p; k++) {
                    double w = col[t - k];
                    if (IS_MAX ? (w > best) : (w < best)) { best = w; best_j = j - k; }
                }
            } else {
                int64_t jj = j, i = t;
                for (int64_t k = 1; k < p; k++) {
                    do { i--; } while (pq_isnull(col[i]));
                    jj--;
                    double w = col[i];
                    if (IS_MAX ? (w > best) : (w < best)) { best = w; best_j = jj; }
                }
            }
        }
        return best;
    }
    // LDS variant: block-decomposed sliding extremum (van Herk / Gil-Werman), O(1) LDS traffic per row and no
    // data-dependent branches.  Valid values are cut into blocks of p; `cur` holds the running block, `suf` the suffix
    // extrema of the previous block.  Window(last p values) = prefix of the current block + suffix of the previous one,
    // so its extremum is ext(prefix, suf[m+1]).  A lazy rescan is hopeless in SIMT (some lane expires on nearly every
    // row and the whole wave pays the rescan); here the only burst -- turning a finished block into its suffix extrema --
    // falls on the same row for every lane unless nulls have shifted a lane.  max/min are exact in any order, so the
    // value is the one at the front of the reference's monotonic deque.
    int64_t m;        // values in the current block
    bool has_prev;
    double prefix;
    __device__ void init_ring() { m = 0; has_prev = false; prefix = 0.0; }
    // cur, suf: rings used as plain arrays of depth >= p
    __device__ double step_ring2(Ring &cur, Ring &suf, double v) {
        if (p <= 0) return v;
        prefix = (m == 0) ? v : (IS_MAX ? fmax(prefix, v) : fmin(prefix, v));
        cur.base[m * 64] = v;
        m += 1;
        double out = prefix;
        if (has_prev && m < p) {
            double s = suf.base[m * 64]; // extremum of previous-block entries m .. p-1
            out = IS_MAX ? fmax(out, s) : fmin(out, s);
        }
        if (m == p) { // block complete: suffix extrema become the next block's `suf`
            double run = cur.base[(p - 1) * 64];
            suf.base[(p - 1) * 64] = run;
            for (int64_t k = p - 2; k >= 0; k--) {
                double x = cur.base[k * 64];
                run = IS_MAX ? fmax(run, x) : fmin(run, x);
                suf.base[k * 64] = run;
            }
            m = 0;
            has_prev = true;
        }
        return out;
    }
};
