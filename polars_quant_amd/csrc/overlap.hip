// overlap.hip -- SEQ kernels + C ABI for the overlap studies (reference: src/talib/overlap.rs).
// One series per lane, reference operation order, null-transparent streaming (N-A).
#include "ops_fused.h"
#include "wt_api.h"

// ---------------------------------------------------------------- C ABI
#define IN1(a) InCols<1>{{a}}
#define IN2(a, b) InCols<2>{{a, b}}
#define OUT1(a) OutCols<1>{{a}}

template <class Inner>
static pq_status mavp_jobs(pq_ctx *ctx, const pq_batch *b, const double *r0, const double *periods, int64_t minp,
                           int64_t maxp, Inner proto, double *out) {
    for (int64_t P = minp; P <= maxp; P++) {
        MavpSelOp<Inner> op{}; op.P = P; op.minp = minp; op.maxp = maxp; op.inner = proto; op.inner.p = P;
        PQ_TRY(launch_seq(ctx, b, op, IN2(r0, periods), OUT1(out)));
    }
    return PQ_OK;
}

extern "C" {

pq_status pq_sma(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && out, "pq_sma: null pointer");
    SmaOp op{}; op.p = p;
    return launch_seq(ctx, b, op, IN1(real), OUT1(out));
}
pq_status pq_ema(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && out, "pq_ema: null pointer");
    { pq_status st; if (wt_ema_all(ctx, b, real, p, out, nullptr, nullptr, nullptr, &st)) return st; }
    EmaOp op{}; op.p = p;
    return launch_seq(ctx, b, op, IN1(real), OUT1(out));
}
pq_status pq_bbands(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double up, double dn,
                    double *u, double *m, double *l) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && u && m && l, "pq_bbands: null pointer");
    BbandsOp op{}; op.p = p; op.up = up; op.dn = dn;
    return launch_seq(ctx, b, op, IN1(real), OutCols<3>{{u, m, l}});
}
pq_status pq_dema(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && out, "pq_dema: null pointer");
    { pq_status st; if (wt_ema_all(ctx, b, real, p, nullptr, out, nullptr, nullptr, &st)) return st; }
    DemaOp op{}; op.p = p;
    return launch_seq(ctx, b, op, IN1(real), OUT1(out));
}
pq_status pq_tema(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && out, "pq_tema: null pointer");
    { pq_status st; if (wt_ema_all(ctx, b, real, p, nullptr, nullptr, out, nullptr, &st)) return st; }
    TemaOp op{}; op.p = p;
    return launch_seq(ctx, b, op, IN1(real), OUT1(out));
}
pq_status pq_t3(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double vf, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && out, "pq_t3: null pointer");
    T3Op op{}; op.p = p;
    t3_coeffs(op, vf);
    return launch_seq(ctx, b, op, IN1(real), OUT1(out));
}
pq_status pq_trima_chain(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && out, "pq_trima: null pointer");
    PQ_TRY(pq_ws_reserve(ctx, sizeof(double) * batch_rows(b) * 8));
    PQ_WS_COL(tmp, ctx, b, 7); // scratch column 7 is reserved for trima (see pq_ma users)
    int64_t k1, k2;                     // overlap.rs:1313-1326
    if (p % 2 == 1) { k1 = p / 2 + 1; k2 = k1; } else { k1 = p / 2; k2 = k1 + 1; }
    PQ_TRY(pq_sma(ctx, b, real, k1, tmp));
    return pq_sma(ctx, b, tmp, k2, out);
}
pq_status pq_wma(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && out, "pq_wma: null pointer");
    WmaOp op{}; op.p = p;
    return launch_seq(ctx, b, op, IN1(real), OUT1(out));
}
pq_status pq_kama(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && out, "pq_kama: null pointer");
    KamaOp op{}; op.p = p;
    return launch_seq(ctx, b, op, IN1(real), OUT1(out));
}
pq_status pq_ma(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, int64_t matype, double *out) {
    switch (matype) { // overlap.rs:857-869
    case 1: return pq_ema(ctx, b, real, p, out);
    case 2: return pq_wma(ctx, b, real, p, out);
    case 3: return pq_dema(ctx, b, real, p, out);
    case 4: return pq_tema(ctx, b, real, p, out);
    case 5: return pq_trima(ctx, b, real, p, out);
    case 6: return pq_kama(ctx, b, real, p, out);
    case 7: return pq_sma(ctx, b, real, p, out);
    case 8: return pq_t3(ctx, b, real, p, 0.0, out);
    default: return pq_sma(ctx, b, real, p, out);
    }
}
pq_status pq_midpoint(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && out, "pq_midpoint: null pointer");
    { pq_status st; if (wt_midpoint(ctx, b, real, p, out, &st)) return st; }
    MidpointOp op{}; op.p = p;
    return launch_seq(ctx, b, op, IN1(real), OUT1(out));
}
pq_status pq_midprice(pq_ctx *ctx, const pq_batch *b, const double *high, const double *low, int64_t p, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(high && low && out, "pq_midprice: null pointer");
    // A pure function of the window: a direct call is a ROW launch (thread per row, the window from L1: 0.1 ms at 5 000 x 2 520 against
    // 1.04 ms for the lane-per-symbol job and 0.35 ms for the wave-per-symbol form).  Inside a recorded suite it stays a sequential
    // job: the fused ROW grid is the tail of a step there and one more job in it costs more than the job it replaces (4.03 against
    // 3.88 ms per step, A/B in one session; PQ_MIDPRICE_ROW=1 records the ROW form anyway).  PQ_MIDPRICE_SEQ=1: never the ROW form.
    // The ROW form rescans the last p valid values per row (walking further back over NULLs): O(p) per row -- taken up to p = 64 (0.1 ms
    // at p = 14; beyond that the O(1)-amortised wave-per-symbol / lane-per-symbol forms below win, and a column of long NULL runs cannot
    // turn the call quadratic)
    // (a recording of a SMALL shard takes the ROW form too: there the sequential job is 1.4 ms of a lone wave and the ROW grid is not the tail)
    if (p <= 64 && (!ctx->rec || ctx->rec_small || getenv("PQ_MIDPRICE_ROW")) && !getenv("PQ_MIDPRICE_SEQ")) { MidpriceRowOp rop{}; rop.p = p; return launch_row(ctx, b, rop, IN2(high, low), OutColsT<MidpriceRowOp, double>{{out}}); }
    { pq_status st; if (wt_midprice(ctx, b, high, low, p, out, &st)) return st; }
    MidpriceOp op{}; op.p = p;
    return launch_seq(ctx, b, op, IN2(high, low), OUT1(out));
}
pq_status pq_sar(pq_ctx *ctx, const pq_batch *b, const double *high, const double *low, double accel, double maxv,
                 double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(high && low && out, "pq_sar: null pointer");
    SarextOp op{}; op.ext = false; op.startvalue = 0.0; op.offset = 0.0;
    op.ai_long = op.a_long = op.ai_short = op.a_short = accel;
    op.am_long = op.am_short = maxv;
    return launch_seq(ctx, b, op, IN2(high, low), OUT1(out));
}
pq_status pq_sarext(pq_ctx *ctx, const pq_batch *b, const double *high, const double *low, double startvalue,
                    double offsetonreverse, double ai_long, double a_long, double am_long, double ai_short,
                    double a_short, double am_short, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(high && low && out, "pq_sarext: null pointer");
    SarextOp op{}; op.ext = true; op.startvalue = startvalue; op.offset = offsetonreverse;
    op.ai_long = ai_long; op.a_long = a_long; op.am_long = am_long;
    op.ai_short = ai_short; op.a_short = a_short; op.am_short = am_short;
    return launch_seq(ctx, b, op, IN2(high, low), OUT1(out));
}
pq_status pq_mavp(pq_ctx *ctx, const pq_batch *b, const double *real, const double *periods, int64_t minp,
                  int64_t maxp, int64_t matype, double *out) {
    PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(real && periods && out, "pq_mavp: null pointer");
    if (minp > maxp || minp < 0)
        return launch_row(ctx, b, FillNullOp{}, InCols<0>{}, OutColsT<FillNullOp, double>{{out}});
    const double *r0 = real; // the select jobs map nulls to 0.0 themselves (overlap.rs:416-424)
    const bool sma_core = matype != 1 && matype != 2 && matype != 3 && matype != 4 && matype != 5 && matype != 6 && matype != 8;
    // (not in a recording of a SMALL shard: there the job's length counts, and blocks of sixteen candidates are two jobs of half the length)
    if (sma_core && maxp - minp < 32 && maxp < (1 << 30) && (PQ_FUSE_OK(ctx) || maxp - minp < 16)) { // every candidate in ONE job: an ordinary (unmasked) output column
        MavpSma32Op op{}; op.lo = (int)minp; op.hi = (int)maxp; op.minp = (int)minp; op.maxp = (int)maxp;
        InCols<2> in{{r0, periods}}; OutCols<1> o1{{out}};
        if (seq_can_lds(ctx, b, op, in, o1)) return launch_seq(ctx, b, op, in, o1); // (one job: a plain launch, or one recorded job)
    }
    // otherwise one masked-select job per candidate period; recorded into a (possibly temporary) suite so that all
    // of them run as ONE grid
    SuiteScope scope(ctx, b);
    PQ_TRY(scope.status);
    rec_set_shared_out(ctx, true); // the jobs write disjoint rows of `out`
    pq_status st = PQ_OK;
    bool blocked = false;
    if (matype != 2 && matype != 3 && matype != 4 && matype != 5 && matype != 6 && matype != 8 && maxp < (1 << 30)) {
        // SMA: sixteen candidate periods per job (states in registers); EMA: eight (states in LDS).  Two passes: decide
        // first whether EVERY block fits the tiled body -- nothing may be recorded before that is known, or the rows of `out`
        // would get a second writer from the per-period fallback below
        const bool sma8 = matype != 1 && !PQ_FUSE_OK(ctx); // a small-shard recording: blocks of eight candidates, jobs of half the length
        const int64_t per_job = (matype == 1 || sma8) ? 8 : 16;
        InCols<2> in{{r0, periods}}; OutCols<1> o1{{out}};
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 0) blocked = true;
            for (int64_t lo = minp; lo <= maxp && st == PQ_OK && blocked; lo += per_job) {
                int64_t hi = lo + per_job - 1 < maxp ? lo + per_job - 1 : maxp;
                if (matype != 1 && sma8) {
                    MavpSma8Op op{}; op.lo = (int)lo; op.hi = (int)hi; op.minp = (int)minp; op.maxp = (int)maxp;
                    if (pass == 0) blocked = seq_can_lds(ctx, b, op, in, o1);
                    else st = launch_seq(ctx, b, op, in, o1);
                } else if (matype != 1) {
                    MavpSma16Op op{}; op.lo = (int)lo; op.hi = (int)hi; op.minp = (int)minp; op.maxp = (int)maxp;
                    if (pass == 0) blocked = seq_can_lds(ctx, b, op, in, o1);
                    else st = launch_seq(ctx, b, op, in, o1);
                } else {
                    MavpBlockOp<1> op{}; op.lo = (int)lo; op.hi = (int)hi; op.minp = (int)minp; op.maxp = (int)maxp;
                    if (pass == 0) blocked = seq_can_lds(ctx, b, op, in, o1);
                    else st = launch_seq(ctx, b, op, in, o1);
                }
            }
            if (!blocked) break;
        }
    }
    if (!blocked && st == PQ_OK) switch (matype) { // overlap.rs:857-869
    case 1: st = mavp_jobs(ctx, b, r0, periods, minp, maxp, EmaOp{}, out); break;
    case 2: st = mavp_jobs(ctx, b, r0, periods, minp, maxp, WmaOp{}, out); break;
    case 3: st = mavp_jobs(ctx, b, r0, periods, minp, maxp, DemaOp{}, out); break;
    case 4: st = mavp_jobs(ctx, b, r0, periods, minp, maxp, TemaOp{}, out); break;
    case 6: st = mavp_jobs(ctx, b, r0, periods, minp, maxp, KamaOp{}, out); break;
    case 8: { T3Op t3{}; t3_coeffs(t3, 0.0);
              st = mavp_jobs(ctx, b, r0, periods, minp, maxp, t3, out); break; }
    case 5: { // TRIMA is two chained SMAs: select from a materialised MA column per period
        PQ_TRY(pq_ws_reserve(ctx, sizeof(double) * batch_rows(b) * 8));
        PQ_WS_COL(rz, ctx, b, 5); PQ_WS_COL(ma, ctx, b, 6);
        rec_set_shared_out(ctx, false);
        PQ_TRY(launch_row(ctx, b, ReplaceNullOp{}, IN1(real), OutColsT<ReplaceNullOp, double>{{rz}}));
        r0 = rz;
        st = PQ_OK;
        for (int64_t P = minp; P <= maxp && st == PQ_OK; P++) {
            rec_set_shared_out(ctx, false);
            st = pq_trima(ctx, b, r0, P, ma);
            if (st != PQ_OK) break;
            MavpPickOp sel{}; sel.P = P; sel.minp = minp; sel.maxp = maxp;
            rec_set_shared_out(ctx, true);
            st = launch_seq(ctx, b, sel, IN2(ma, periods), OUT1(out));
        }
        break; }
    default: st = mavp_jobs(ctx, b, r0, periods, minp, maxp, SmaOp{}, out); break;
    }
    (void)0;
    rec_set_shared_out(ctx, false);
    PQ_TRY(st);
    return scope.finish();
}

} // extern "C"
