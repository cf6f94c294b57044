// ops_backtest_wave.h -- the per-symbol backtest in its ONE-SYMBOL-PER-WAVEFRONT, row-parallel form (round 3).
//
// Reference: src/backtest/vectorized.rs:124-194 (scan), src/backtest/metrics.rs:7-152 (summary), momentum.rs:250-283 + D-8
// (MACD-cross signals).  The lane-per-symbol form (ops_backtest.h) walks T rows of a strict recurrence serially and costs the
// same whether the chip holds 5 000 symbols or 625; this form breaks the walk into three parts whose serial length is not T:
//
//  (A) SIGNALS by speculative T-chunks.  Lane c owns the rows [c*C, (c+1)*C) (C = ceil(T / 64)).  It starts the MACD state
//      machine nW chunks EARLY, as if the series began there (SMA seeds and all), and runs into its own chunk.  The EMA
//      recurrence fma(alpha, x - e, e) is a contraction: the speculative trajectory approaches the true one geometrically and
//      then MERGES BITWISE with it (once two trajectories hold the same double they are identical forever: the step is a
//      deterministic function of state and input).  Proof per chunk instead of hope: lane c's state at its first own row is
//      compared AS RAW BITS with lane c-1's state after its last row; equal => every value lane c produced is the serial
//      one.  Lane 0 starts at row 0 and is exact by construction, so exactness propagates lane by lane.  A chunk that fails the
//      test is re-run from its predecessor's end state (C rows, counted in `stats`); flat prices (a trajectory that never
//      contracts) therefore cost time, never correctness.
//  (B) STATE WALK over events only.  (pos, cash, entry_cost) change only on rows where a buy finds the pool flat or a sell
//      finds it long (vectorized.rs:146-175): the wave jumps from event to event with ballot / find-first-set over the
//      per-lane signal masks -- ~100 events per symbol instead of 2 520 rows -- and does the event's arithmetic once,
//      wave-uniform, in the reference's operation order.
//  (C) ROW-PARALLEL FILL.  64 consecutive rows per wave instruction: position / cash come from the block's event table,
//      equity = cash + pos * price (the same two roundings), stores are 512 contiguous bytes per column and instruction.
//      No LDS transpose of outputs, no storer wave.  The summary's running max is an exact prefix max, max_drawdown /
//      win_rate / total_trades are exact; the ordered f64 sums of calculate_summary (mean, variance, covariance) are summed
//      per chunk and then across lanes in a fixed order: <= 1e-12 relative (they are tolerance columns of the parity bar).
#pragma once
#include "ops_backtest.h"

constexpr int BTW_MAX_C = 64;            // rows per lane chunk: one 64-bit signal mask per lane => len <= 4096
constexpr int BTW_TAB = 66;              // event table of one 64-row block: state before the block + at most 64 events

struct BtWaveArgs {
    const double *price;
    const uint8_t *buy, *sell; // nullptr: MACD-cross signals are generated in-kernel
    const double *bench;       // nullable, [n][stride]
    double *position, *cash, *equity, *summary; // each nullable
    pq_bt_params prm;
    int32_t fast, slow, sig;
    int32_t C, P, nW;          // chunk rows, chunk pitch in LDS (odd: conflict-free per-lane reads), warm-up chunks
    uint32_t magic;            // ceil(2^20 / C): i / C == (i * magic) >> 20 for i < 64 * C
    unsigned long long *stats; // nullable: [0] symbols, [1] speculative chunks that failed the bit test, [2] chunk re-runs
};

__device__ __forceinline__ unsigned long long btw_ballot(bool x) { return __builtin_amdgcn_ballot_w64(x); }
__device__ __forceinline__ unsigned long long btw_readlane(unsigned long long v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double btw_readlane(double v, int l) {
    return __longlong_as_double((long long)btw_readlane((unsigned long long)__double_as_longlong(v), l));
}
__device__ __forceinline__ unsigned long long btw_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ void btw_lds_sync() { __syncthreads(); } // one wave per workgroup: orders LDS traffic for the compiler

// The MACD-cross state machine of one lane (BtMacdOp::step without the trade part)
struct BtwMacd {
    EmaCore ef, es, eg;
    double prev_m, prev_s;
    __device__ void init(int64_t fast, int64_t slow, int64_t sig, int64_t T) {
        ef.init(fast, T); es.init(slow, T); eg.init(sig, T);
        prev_m = pq_null(); prev_s = pq_null();
    }
    __device__ bool steady() const { return ef.steady() && es.steady() && eg.steady() && !pq_isnull(prev_m) && !pq_isnull(prev_s); }
    // general row (nulls, warm-up): momentum.rs:250-283 + D-8
    __device__ __forceinline__ void step(int i, double px, bool &buy, bool &sell) {
        const double f = ef.step(px), sl = es.step(px);
        const double m = (!pq_isnull(f) && !pq_isnull(sl)) ? f - sl : pq_null();
        const double g = eg.step(z0b(m));
        const bool ok = i > 0 && !pq_isnull(m) && !pq_isnull(g) && !pq_isnull(prev_m) && !pq_isnull(prev_s);
        buy = ok && (prev_m <= prev_s) && (m > g);
        sell = ok && (prev_m >= prev_s) && (m < g);
        prev_m = m; prev_s = g;
    }
    // steady state, non-null input: the same operations without the tests
    __device__ __forceinline__ void fast_nosig(double px) {
        const double f = ef.fast(px), sl = es.fast(px);
        const double m = f - sl;
        prev_s = eg.fast(m);
        prev_m = m;
    }
};
__device__ __forceinline__ void btw_bcast_ema(EmaCore &e, int l) { // every lane receives lane l's core
    e.count = (int64_t)btw_readlane((unsigned long long)e.count, l);
    e.ema = btw_readlane(e.ema, l);
    e.sum = btw_readlane(e.sum, l);
}

template <bool MACD>
__global__ __launch_bounds__(64) void bt_wave_kernel(BtWaveArgs a, Dims d) {
    extern __shared__ __align__(16) unsigned char btw_lds[];
    const int lane = (int)threadIdx.x;
    const int64_t s = blockIdx.x;
    const int T = (int)d.len, C = a.C, P = a.P, PC = P - C;
    const unsigned magic = a.magic;
    const int64_t base = s * d.stride;
    double *px = reinterpret_cast<double *>(btw_lds);   // [64 * P]: row i at i + (i / C) * (P - C); later the equity row, then r
    double *bm = px + 64 * P;                           // [64 * P] when a.bench
    double *tab = bm + (a.bench ? 64 * P : 0);          // [2][BTW_TAB]
    auto addr = [&](int i) { return i + (int)(((unsigned)i * magic) >> 20) * PC; };
    const pq_bt_params prm = a.prm;
    if (T == 0) {
        if (a.summary && lane < 8) a.summary[s * PQ_SUMMARY_COLS + lane] = 0.0;
        return;
    }

    // ---- phase 0: the symbol's rows, coalesced, into LDS; signal masks when the signals are inputs
    unsigned long long bmask = 0, smask = 0; // MACD: bit b of lane c = row c*C + b; else: bit b of lane w = row 64*w + b
    bool null_seen = false;
#pragma unroll 4
    for (int j = 0; j < C; j++) {
        const int i = 64 * j + lane;
        double v = pq_null();
        if (i < T) v = a.price[base + i];
        px[addr(i)] = v;
        null_seen |= (i < T) && pq_isnull(v);
        if (a.bench) bm[addr(i)] = i < T ? a.bench[base + i] : 0.0;
        if (!MACD) {
            const bool valid = !(isnan(v) || v <= 0.0); // vectorized.rs:141: such rows leave the state untouched (a NULL is a NaN)
            bool b = false, se = false;
            if (i < T) { b = a.buy[base + i] != 0; se = a.sell[base + i] != 0; }
            const unsigned long long bb = btw_ballot(b && valid), sb = btw_ballot(se && valid);
            if (lane == j) { bmask = bb; smask = sb; }
        }
    }
    const bool any_null = btw_ballot(null_seen) != 0;
    btw_lds_sync();

    // ---- phase 1 (A): MACD-cross signals by speculative chunks
    if (MACD) {
        const int c = lane;
        BtwMacd st;
        st.init(a.fast, a.slow, a.sig, T);
        if (!(st.ef.dead || st.es.dead || st.eg.dead)) { // a dead average => no signal on any row
            const int nlive = (T + C - 1) / C;
            const bool live = c < nlive;
            const int nk = nlive < a.nW + 1 ? nlive : a.nW + 1; // chunk iterations
            // Schedule.  Lanes c > nW ("speculative") start fresh nW chunks early and reach their own chunk in the LAST iteration.
            // The first nW + 1 chunks have no room for a warm-up: lane 0 walks chunks 0 .. nk-2 serially from row 0 (exact) WHILE
            // the speculative lanes warm up, and hands its state after chunk k to lane k + 1; in the last iteration the lanes
            // 1 .. nW run their own chunk from that exact state.  So only iteration 0 sees averages that are not seeded yet
            // (general path); every later one is straight-line code.
            const bool spec = live && c > a.nW;
            double s_f = 0, s_s = 0, s_g = 0, s_pm = 0, s_ps = 0;
            bool s_steady = false;
            for (int k = 0; k < nk; k++) {
                const bool last = k == nk - 1;
                const int q = last ? c : (spec ? c - a.nW + k : k);
                const bool active = live && (last ? (c > 0 || nk == 1) : (spec || c == 0));
                const bool rec = active && q == c; // this lane's own rows: their signals are kept
                if (last) { s_f = st.ef.ema; s_s = st.es.ema; s_g = st.eg.ema; s_pm = st.prev_m; s_ps = st.prev_s; s_steady = st.steady(); }
                const double *row = px + (active ? q : 0) * P;
                const bool fastk = !any_null && btw_ballot(active && !st.steady()) == 0; // wave-uniform
                const bool anyrec = btw_ballot(rec) != 0;
                if (fastk && !anyrec) {
                    if (active) {
                        int b = 0;
                        for (; b + 8 <= C; b += 8) {
                            double x[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) x[u] = row[b + u];
#pragma unroll
                            for (int u = 0; u < 8; u++) st.fast_nosig(x[u]);
                        }
                        for (; b < C; b++) st.fast_nosig(row[b]);
                    }
                } else if (fastk) {
                    if (active) {
                        // buy = !(m' > g') && (m > g), sell = !(m' < g') && (m < g) on non-NaN values: two predicates per row
                        unsigned long long gt = 0, lt = 0, vm = 0;
                        const bool pgt = st.prev_m > st.prev_s, plt = st.prev_m < st.prev_s;
                        for (int b = 0; b < C; b++) {
                            const double x = row[b];
                            st.fast_nosig(x);
                            gt |= (unsigned long long)(st.prev_m > st.prev_s) << b;
                            lt |= (unsigned long long)(st.prev_m < st.prev_s) << b;
                            vm |= (unsigned long long)(x > 0.0) << b; // valid price (NaN / NULL compare false)
                        }
                        // (a NaN average is absorbing -- every later m, g is NaN and both predicates stay false --, so "not greater on
                        // the previous row" can stand for the reference's "less or equal")
                        if (rec) {
                            const unsigned long long pg = (gt << 1) | (unsigned long long)pgt, pl = (lt << 1) | (unsigned long long)plt;
                            bmask = gt & ~pg & vm;
                            smask = lt & ~pl & vm;
                        }
                    }
                } else {
                    if (active)
                        for (int b = 0; b < C; b++) {
                            const double x = row[b];
                            bool bu, se;
                            st.step(q * C + b, x, bu, se);
                            if (rec) {
                                const bool valid = !(isnan(x) || x <= 0.0);
                                bmask |= (unsigned long long)(bu && valid) << b;
                                smask |= (unsigned long long)(se && valid) << b;
                            }
                        }
                }
                if (!last) { // lane 0 -> lane k + 1: the exact state in front of chunk k + 1
                    BtwMacd h = st;
                    btw_bcast_ema(h.ef, 0); btw_bcast_ema(h.es, 0); btw_bcast_ema(h.eg, 0);
                    h.prev_m = btw_readlane(st.prev_m, 0); h.prev_s = btw_readlane(st.prev_s, 0);
                    if (lane == k + 1) st = h;
                }
            }
            // verification: my state at my first row == my predecessor's state after its last row, as raw bits
            auto mismatch = [&]() {
                const bool p_steady = __shfl_up((int)st.steady(), 1) != 0;
                const bool same = s_steady && p_steady && btw_bits(s_f) == btw_bits(__shfl_up(st.ef.ema, 1)) &&
                                  btw_bits(s_s) == btw_bits(__shfl_up(st.es.ema, 1)) && btw_bits(s_g) == btw_bits(__shfl_up(st.eg.ema, 1)) &&
                                  btw_bits(s_pm) == btw_bits(__shfl_up(st.prev_m, 1)) && btw_bits(s_ps) == btw_bits(__shfl_up(st.prev_s, 1));
                return spec && !same;
            };
            unsigned long long mism = btw_ballot(mismatch());
            const int n_failed = __popcll(mism);
            int n_rerun = 0;
            while (mism) {
                const int cs = __builtin_ctzll(mism); // lowest failing chunk: its predecessor is exact
                mism &= mism - 1;
                BtwMacd pr = st;                      // every lane: the END state of lane cs - 1
                btw_bcast_ema(pr.ef, cs - 1); btw_bcast_ema(pr.es, cs - 1); btw_bcast_ema(pr.eg, cs - 1);
                pr.prev_m = btw_readlane(st.prev_m, cs - 1); pr.prev_s = btw_readlane(st.prev_s, cs - 1);
                if (lane == cs) {
                    st = pr;
                    s_f = st.ef.ema; s_s = st.es.ema; s_g = st.eg.ema; s_pm = st.prev_m; s_ps = st.prev_s; s_steady = true;
                    bmask = 0; smask = 0;
                    const double *row = px + cs * P;
                    for (int b = 0; b < C; b++) {
                        const double x = row[b];
                        bool bu, se;
                        st.step(cs * C + b, x, bu, se);
                        const bool valid = !(isnan(x) || x <= 0.0);
                        bmask |= (unsigned long long)(bu && valid) << b;
                        smask |= (unsigned long long)(se && valid) << b;
                    }
                }
                n_rerun++;
                // the re-run changed lane cs's end state: its successor is tested again
                const bool again = mismatch() && lane == cs + 1;
                mism |= btw_ballot(again);
            }
            if (a.stats && lane == 0) {
                atomicAdd(a.stats + 1, (unsigned long long)n_failed);
                atomicAdd(a.stats + 2, (unsigned long long)n_rerun);
            }
        }
    }
    if (a.stats && lane == 0) atomicAdd(a.stats + 0, 1ULL);

    // ---- phases 2 + 3 (B, C): per 64-row block, walk the block's events, then fill its rows
    const int CQ = MACD ? C : 64;
    double pos = 0.0, avail = prm.initial_capital, entry_cost = 0.0;
    int trades = 0, wins = 0;
    bool flat = true;
    int cq = 0, cb = 0; // search cursor: chunk / word, bit
    auto find = [&]() -> int { // next row >= cursor whose (valid-price) signal the pool can act on; -1 if none
        unsigned long long m = flat ? bmask : smask;
        if (lane < cq) m = 0;
        if (lane == cq) m = cb >= 64 ? 0 : (m >> cb) << cb;
        const unsigned long long bal = btw_ballot(m != 0);
        if (!bal) return -1;
        const int c1 = __builtin_ctzll(bal);
        const int b1 = __builtin_ctzll(btw_readlane(m, c1));
        cq = c1; cb = b1 + 1;
        return c1 * CQ + b1;
    };
    int r = find();
    const int nblk = (T + 63) / 64;
    for (int j = 0; j < nblk; j++) {
        int nev = 0;
        if (lane == 0) { tab[0] = pos; tab[BTW_TAB] = avail; }
        bool evb = false;
        const int rend = 64 * (j + 1);
        while (r >= 0 && r < rend) {
            const double p = px[addr(r)]; // wave-uniform address
            bool ev = false;
            if (flat) { // vectorized.rs:146-161
                const double exec = p + prm.buy_slippage;
                const double cur_eq = avail + pos * p;
                const double deploy = cur_eq * prm.position_size;
                const double qty = floor(deploy / exec);
                if (qty > 0.0) {
                    const double cost = qty * exec;
                    const double fee = fmax(cost * prm.buy_commission_rate, prm.min_commission);
                    pos += qty;
                    avail -= cost + fee;
                    entry_cost = pos * p;
                    trades += 1;
                    ev = true;
                    flat = false;
                }
            } else { // :162-175
                const double exec = p - prm.sell_slippage;
                const double revenue = pos * exec;
                const double fee = fmax(revenue * prm.sell_commission_rate, prm.min_commission);
                const double net = revenue - fee;
                if (net > entry_cost) wins += 1;
                avail += net;
                pos = 0.0;
                ev = true;
                flat = true;
            }
            if (ev) {
                nev++;
                if (lane == 0) { tab[nev] = pos; tab[BTW_TAB + nev] = avail; }
                evb |= lane == (r & 63);
            }
            r = find();
        }
        btw_lds_sync();
        const int i = 64 * j + lane;
        const unsigned long long bal = btw_ballot(evb);
        const int idx = __popcll(bal & ((2ULL << lane) - 1ULL)); // events at rows <= mine in this block
        const double pi = tab[idx], ci = tab[BTW_TAB + idx];
        const int ai = addr(i);
        double x = px[ai];
        if (pq_isnull(x)) x = __longlong_as_double(0x7FF8000000000000LL); // null -> NaN (vectorized.rs:70-78)
        const double eq = ci + pi * x;
        if (i < T) {
            if (a.position) __builtin_nontemporal_store(pi, &a.position[base + i]);
            if (a.cash) __builtin_nontemporal_store(ci, &a.cash[base + i]);
            if (a.equity) __builtin_nontemporal_store(eq, &a.equity[base + i]);
        }
        px[ai] = eq;
        btw_lds_sync();
    }
    if (!a.summary) return;

    // ---- summary (metrics.rs:7-152): lane c owns rows [c*C, (c+1)*C) of the equity row now in LDS
    const int c = lane, lo = c * C, hi = lo + C < T ? lo + C : T;
    const int nrow = hi - lo; // <= 0: idle lane
    double *erow = px + c * P;
    const double init = prm.initial_capital;
    const double NEG_INF = __longlong_as_double((long long)0xFFF0000000000000ULL);
    double lm = NEG_INF;
    for (int b = 0; b < nrow; b++) { const double e = erow[b]; if (e > lm) lm = e; }
    double Mx = lm; // inclusive prefix max over the lanes (exact in any order)
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(Mx, off);
        if (lane >= off && t > Mx) Mx = t;
    }
    double max_eq = __shfl_up(Mx, 1);
    if (lane == 0 || !(max_eq > init)) max_eq = init; // the running max starts at initial_capital (metrics.rs:21)
    double prev = init;
    if (c > 0 && nrow > 0) prev = px[addr(lo - 1)];
    const double last_eq = px[addr(T - 1)];
    double max_dd = 0.0, rs = 0.0;
    for (int b = 0; b < nrow; b++) { // metrics.rs:26-49
        const double e = erow[b];
        if (e > max_eq) max_eq = e;
        const double dd = (max_eq > 0.0) ? (max_eq - e) / max_eq : 0.0;
        if (dd > max_dd) max_dd = dd;
        const double rr = (prev > 0.0) ? (e - prev) / prev : 0.0;
        rs += rr;
        erow[b] = rr;
        prev = e;
    }
    auto wave_sum = [&](double v) { // fixed order, the same value on every lane
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        return v;
    };
    for (int off = 32; off > 0; off >>= 1) { const double t = __shfl_xor(max_dd, off); if (t > max_dd) max_dd = t; }
    const double ret_sum = wave_sum(rs);
    const double DAYS = 252.0, RF = 0.03;
    const double total_return = (last_eq - init) / init;
    const double mean = ret_sum / (double)T;
    const double dof = fmax((double)T - 1.0, 1.0);
    double bmean = 0.0;
    const double *brow = bm + c * P;
    double pb0 = 0.0;
    if (a.bench) { // metrics.rs:86-140
        pb0 = (c == 0 || nrow <= 0) ? bm[0] : bm[addr(lo - 1)];
        double pb = pb0, bs = 0.0;
        for (int b = 0; b < nrow; b++) { const double bv = brow[b]; bs += (pb > 0.0) ? (bv - pb) / pb : 0.0; pb = bv; }
        bmean = wave_sum(bs) / (double)T;
    }
    double vs = 0.0, bvs = 0.0, cvs = 0.0;
    {
        double pb = pb0;
        for (int b = 0; b < nrow; b++) {
            const double dlt = erow[b] - mean;
            vs += dlt * dlt;
            if (a.bench) {
                const double bv = brow[b];
                const double br = (pb > 0.0) ? (bv - pb) / pb : 0.0;
                const double db = br - bmean;
                bvs += db * db;
                cvs += dlt * db;
                pb = bv;
            }
        }
    }
    vs = wave_sum(vs);
    const double ann = (total_return > -1.0) ? pow(1.0 + total_return, DAYS / (double)T) - 1.0 : -1.0;
    const double var = vs / dof;
    const double vol = sqrt(var) * sqrt(DAYS);
    const double sharpe = (vol > 0.0) ? (ann - RF) / vol : 0.0;
    const double win_rate = (trades > 0) ? (double)wins / (double)trades : 0.0;
    double alpha = 0.0, beta = 0.0;
    if (a.bench) {
        const double bvar = wave_sum(bvs) / dof, cov = wave_sum(cvs) / dof;
        if (bvar > 0.0) beta = cov / bvar;
        const double b0 = bm[0], b1 = bm[addr(T - 1)];
        const double btr = (b0 > 0.0) ? (b1 - b0) / b0 : 0.0;
        const double bann = (btr > -1.0) ? pow(1.0 + btr, DAYS / (double)T) - 1.0 : -1.0;
        alpha = ann - (RF + beta * (bann - RF));
    }
    if (lane == 0) {
        double *sm = a.summary + s * PQ_SUMMARY_COLS;
        sm[0] = ann; sm[1] = max_dd; sm[2] = alpha; sm[3] = beta; sm[4] = sharpe;
        sm[5] = fmax(total_return, 0.0); sm[6] = win_rate; sm[7] = (double)trades;
    }
}

// host side: shape of the wave form for a batch, or false when it does not apply (len > 64 * BTW_MAX_C)
static inline bool btw_plan(const pq_batch *b, int64_t fast, int64_t slow, int64_t sig, bool macd, BtWaveArgs &a, size_t &lds_bytes, bool bench) {
    if (b->len > 64 * BTW_MAX_C) return false;
    const int T = (int)b->len;
    int C = (T + 63) / 64;
    if (C < 1) C = 1;
    a.C = C;
    a.P = C | 1;
    a.magic = (uint32_t)(((1u << 20) + (unsigned)C - 1) / (unsigned)C);
    a.nW = 0;
    if (macd) {
        // rows until a restarted MACD state machine has merged bitwise with the true one (measured on the SURVEY 8d generator,
        // scripts/sim_macd_merge.py: median 32 / alpha, 99.9 % within 38 / alpha of the slowest average, the signal line ~10 /
        // alpha_sig behind).  A chunk that has not merged is re-run, so this is a speed knob only.
        const double af = 2.0 / ((double)(fast > slow ? fast : slow) + 1.0), ag = 2.0 / ((double)sig + 1.0);
        double rows = 36.0 / af + 8.0 / ag;
        if (!(rows < 1e9)) rows = 1e9;
        int64_t nW = ((int64_t)rows + C - 1) / C;
        if (const char *e = getenv("PQ_BT_WARM_CHUNKS")) nW = atoll(e); // tests: force failing chunks (re-run path)
        if (nW < 1) nW = 1;
        a.nW = (int32_t)(nW > 64 ? 64 : nW);
    }
    lds_bytes = (size_t)64 * a.P * 8 * (bench ? 2 : 1) + 2 * BTW_TAB * 8;
    return true;
}
