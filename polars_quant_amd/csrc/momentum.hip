// momentum.hip -- kernels + C ABI for the momentum indicators (reference: src/talib/momentum.rs and
// the pure-Python composites of python/polars_quant/talib/momentum.py).
// momentum.rs functions are N-B (nulls rejected, momentum.rs:12-13): inputs are assumed null-free
// here; the host layer calls pq_count_nulls first and raises like the reference does.
#include "ops_momentum.h"
#include "wt_api.h"

// ---------------------------------------------------------------- C ABI
#define CHK(name, cond) PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(cond, name ": null pointer")
#define WS(ncols) PQ_TRY(pq_ws_reserve(ctx, sizeof(double) * batch_rows(b) * 8))

extern "C" {

#define LAGFN(name, KIND)                                                                                   \
    pq_status name(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {            \
        CHK(#name, real && out);                                                                            \
        LagOp<KIND> op{}; op.p = p;                                                                           \
        return launch_row(ctx, b, op, InCols<1>{{real}}, OutColsT<LagOp<KIND>, double>{{out}});             \
    }
LAGFN(pq_mom, 0)
LAGFN(pq_roc, 1)
LAGFN(pq_rocp, 2)
LAGFN(pq_rocr, 3)
LAGFN(pq_rocr100, 4)

pq_status pq_returns(pq_ctx *ctx, const pq_batch *b, const double *price, int64_t period, int64_t method, double *out) {
    CHK("pq_returns", price && out);
    if (method == 0) { ReturnsOp<0> op{}; op.p = period; return launch_row(ctx, b, op, InCols<1>{{price}}, OutColsT<ReturnsOp<0>, double>{{out}}); }
    ReturnsOp<1> op{}; op.p = (method == 1) ? period : 0; // an unknown method: all null (as period <= 0)
    return launch_row(ctx, b, op, InCols<1>{{price}}, OutColsT<ReturnsOp<1>, double>{{out}});
}
pq_status pq_rolling_max(pq_ctx *ctx, const pq_batch *b, const double *x, int64_t window, double *out) {
    CHK("pq_rolling_max", x && out);
    RollingExtOp<true> op{}; op.p = window;
    return launch_row(ctx, b, op, InCols<1>{{x}}, OutColsT<RollingExtOp<true>, double>{{out}});
}
pq_status pq_rolling_min(pq_ctx *ctx, const pq_batch *b, const double *x, int64_t window, double *out) {
    CHK("pq_rolling_min", x && out);
    RollingExtOp<false> op{}; op.p = window;
    return launch_row(ctx, b, op, InCols<1>{{x}}, OutColsT<RollingExtOp<false>, double>{{out}});
}
pq_status pq_bop(pq_ctx *ctx, const pq_batch *b, const double *o, const double *h, const double *l, const double *c,
                 double *out) {
    CHK("pq_bop", o && h && l && c && out);
    return launch_row(ctx, b, BopOp{}, InCols<4>{{o, h, l, c}}, OutColsT<BopOp, double>{{out}});
}
pq_status pq_aroon(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *up, double *dn) {
    CHK("pq_aroon", h && l && up && dn);
    AroonOp<0> op{}; op.p = p;
    return launch_row(ctx, b, op, InCols<2>{{h, l}}, OutColsT<AroonOp<0>, double>{{up, dn}});
}
pq_status pq_aroonosc(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *out) {
    CHK("pq_aroonosc", h && l && out);
    AroonOp<1> op{}; op.p = p;
    return launch_row(ctx, b, op, InCols<2>{{h, l}}, OutColsT<AroonOp<1>, double>{{out}});
}
pq_status pq_aroon_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *up, double *dn, double *osc) {
    CHK("pq_aroon_all", h && l && up && dn && osc);
    AroonOp<2> op{}; op.p = p;
    return launch_row(ctx, b, op, InCols<2>{{h, l}}, OutColsT<AroonOp<2>, double>{{up, dn, osc}});
}
pq_status pq_willr(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p,
                   double *out) {
    CHK("pq_willr", h && l && c && out);
    WillrOp op{}; op.p = p;
    return launch_row(ctx, b, op, InCols<3>{{h, l, c}}, OutColsT<WillrOp, double>{{out}});
}
pq_status pq_cci_chain(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p,
                 double *out) {
    CHK("pq_cci", h && l && c && out);
    WS(8);
    PQ_WS_COL(sma_tp, ctx, b, 0);
    SmaTpOp s{}; s.p = p;
    PQ_TRY(launch_seq(ctx, b, s, InCols<3>{{h, l, c}}, OutCols<1>{{sma_tp}}));
    CciDevOp op{}; op.p = p;
    return launch_row(ctx, b, op, InCols<4>{{h, l, c, sma_tp}}, OutColsT<CciDevOp, double>{{out}});
}
pq_status pq_cmo(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    CHK("pq_cmo", real && out);
    CmoOp op{}; op.p = p;
    return launch_seq(ctx, b, op, InCols<1>{{real}}, OutCols<1>{{out}});
}
pq_status pq_rsi(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    CHK("pq_rsi", real && out);
    { pq_status st; if (wt_rsi(ctx, b, real, p, out, &st)) return st; }
    RsiOp op{}; op.p = p;
    return launch_seq(ctx, b, op, InCols<1>{{real}}, OutCols<1>{{out}});
}
pq_status pq_macd(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t sig,
                  double *macd, double *signal, double *hist) {
    CHK("pq_macd", real && macd && signal && hist);
    { pq_status st; if (wt_macd(ctx, b, real, fast, slow, sig, sig, macd, signal, hist, nullptr, nullptr, nullptr, &st)) return st; }
    MacdOp op{}; op.fast = fast; op.slow = slow; op.sig = sig;
    return launch_seq(ctx, b, op, InCols<1>{{real}}, OutCols<3>{{macd, signal, hist}});
}
pq_status pq_macdfix(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t sig, double *macd, double *signal,
                     double *hist) {
    return pq_macd(ctx, b, real, 12, 26, sig, macd, signal, hist); // momentum.py:90-92
}
pq_status pq_trix(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    CHK("pq_trix", real && out);
    { pq_status st; if (wt_ema_all(ctx, b, real, p, nullptr, nullptr, nullptr, out, &st)) return st; }
    TrixOp op{}; op.p = p;
    return launch_seq(ctx, b, op, InCols<1>{{real}}, OutCols<1>{{out}});
}
pq_status pq_ultosc(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p1,
                    int64_t p2, int64_t p3, double *out) {
    CHK("pq_ultosc", h && l && c && out);
    UltoscOp op{}; op.p1 = p1; op.p2 = p2; op.p3 = p3;
    if (ctx->rec && ctx->rec_small) { // a small-shard recording: 8-row tiles (half the per-tile fills and hand-offs of the longest job; measured better at 625 AND 1 250 symbols)
        UltoscOp8 op8{}; op8.p1 = p1; op8.p2 = p2; op8.p3 = p3;
        InCols<3> in{{h, l, c}}; OutCols<1> o{{out}};
        if (seq_can_lds(ctx, b, op8, in, o)) return launch_seq(ctx, b, op8, in, o);
    }
    return launch_seq(ctx, b, op, InCols<3>{{h, l, c}}, OutCols<1>{{out}});
}
pq_status pq_mfi(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, const double *v,
                 int64_t p, double *out) {
    CHK("pq_mfi", h && l && c && v && out);
    MfiOp op{}; op.p = p;
    return launch_seq(ctx, b, op, InCols<4>{{h, l, c, v}}, OutCols<1>{{out}});
}
#define DMFN(name, MODE)                                                                                       \
    pq_status name(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, \
                   double *out) {                                                                              \
        CHK(#name, h && l && c && out);                                                                        \
        { pq_status st; if (wt_dmi(ctx, b, h, l, c, p, MODE == 0 ? out : nullptr, nullptr, MODE == 1 ? out : nullptr, MODE == 2 ? out : nullptr, nullptr, &st)) return st; } \
        DmOp<MODE> op{}; op.p = p;                                                                               \
        return launch_seq(ctx, b, op, InCols<3>{{h, l, c}}, OutCols<1>{{out}});                                \
    }
DMFN(pq_dx, 0)
DMFN(pq_plus_di, 0) /* momentum.rs:409 returns calc_dm().0 == DX (quirk Q-PDI, decision D-5: literal) */
DMFN(pq_minus_di, 1)
DMFN(pq_adx, 2)
pq_status pq_adxr_chain(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p,
                  double *out) {
    CHK("pq_adxr", h && l && c && out);
    WS(8);
    PQ_WS_COL(adx, ctx, b, 0);
    PQ_TRY(pq_adx(ctx, b, h, l, c, p, adx));
    AdxrOp op{}; op.p = p;
    return launch_row(ctx, b, op, InCols<1>{{adx}}, OutColsT<AdxrOp, double>{{out}});
}
pq_status pq_adxr_from_adx(pq_ctx *ctx, const pq_batch *b, const double *adx, int64_t p, double *out) { // the second half of pq_adxr_chain
    AdxrOp op{}; op.p = p;
    return launch_row(ctx, b, op, InCols<1>{{adx}}, OutColsT<AdxrOp, double>{{out}});
}
pq_status pq_plus_dm(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *out) {
    CHK("pq_plus_dm", h && l && out);
    { pq_status st; if (wt_dm_pair(ctx, b, h, l, p, out, nullptr, &st)) return st; }
    DmRawOp<true> op{}; op.p = p;
    return launch_seq(ctx, b, op, InCols<2>{{h, l}}, OutCols<1>{{out}});
}
pq_status pq_minus_dm(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *out) {
    CHK("pq_minus_dm", h && l && out);
    { pq_status st; if (wt_dm_pair(ctx, b, h, l, p, nullptr, out, &st)) return st; }
    DmRawOp<false> op{}; op.p = p;
    return launch_seq(ctx, b, op, InCols<2>{{h, l}}, OutCols<1>{{out}});
}
// D-6: APO = MA(fast) - MA(slow); PPO = (MA(fast)-MA(slow))/MA(slow)*100 on the reference's calc_ma
pq_status pq_apo_chain(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t matype,
                 double *out) {
    CHK("pq_apo", real && out);
    WS(8);
    PQ_WS_COL(f, ctx, b, 0); PQ_WS_COL(s, ctx, b, 1);
    PQ_TRY(pq_ma(ctx, b, real, fast, matype, f));
    PQ_TRY(pq_ma(ctx, b, real, slow, matype, s));
    return launch_row(ctx, b, BinOp<0>{}, InCols<2>{{f, s}}, OutColsT<BinOp<0>, double>{{out}});
}
pq_status pq_ppo_chain(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t matype,
                 double *out) {
    CHK("pq_ppo", real && out);
    WS(8);
    PQ_WS_COL(f, ctx, b, 0); PQ_WS_COL(s, ctx, b, 1);
    PQ_TRY(pq_ma(ctx, b, real, fast, matype, f));
    PQ_TRY(pq_ma(ctx, b, real, slow, matype, s));
    return launch_row(ctx, b, BinOp<1>{}, InCols<2>{{f, s}}, OutColsT<BinOp<1>, double>{{out}});
}
pq_status pq_macdext_chain(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t fastmt, int64_t slow,
                     int64_t slowmt, int64_t sig, int64_t sigmt, double *macd, double *signal, double *hist) {
    CHK("pq_macdext", real && macd && signal && hist); // momentum.py:83-88
    WS(8);
    PQ_WS_COL(f, ctx, b, 0); PQ_WS_COL(s, ctx, b, 1);
    PQ_TRY(pq_ma(ctx, b, real, fast, fastmt, f));
    PQ_TRY(pq_ma(ctx, b, real, slow, slowmt, s));
    PQ_TRY(launch_row(ctx, b, BinOp<0>{}, InCols<2>{{f, s}}, OutColsT<BinOp<0>, double>{{macd}}));
    PQ_TRY(pq_ma(ctx, b, macd, sig, sigmt, signal));
    return launch_row(ctx, b, BinOp<0>{}, InCols<2>{{macd, signal}}, OutColsT<BinOp<0>, double>{{hist}});
}
pq_status pq_stochf_chain(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t fastk,
                    int64_t fastd, int64_t fastd_mt, double *outk, double *outd) {
    CHK("pq_stochf", h && l && c && outk && outd); // momentum.py:188-195
    FastkOp op{}; op.k = fastk;
    PQ_TRY(launch_row(ctx, b, op, InCols<3>{{h, l, c}}, OutColsT<FastkOp, double>{{outk}}));
    return pq_ma(ctx, b, outk, fastd, fastd_mt, outd);
}
pq_status pq_stoch_chain(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t fastk,
                   int64_t slowk, int64_t slowk_mt, int64_t slowd, int64_t slowd_mt, double *outk, double *outd) {
    CHK("pq_stoch", h && l && c && outk && outd); // momentum.py:178-186
    WS(8);
    PQ_WS_COL(fk, ctx, b, 0);
    FastkOp op{}; op.k = fastk;
    PQ_TRY(launch_row(ctx, b, op, InCols<3>{{h, l, c}}, OutColsT<FastkOp, double>{{fk}}));
    PQ_TRY(pq_ma(ctx, b, fk, slowk, slowk_mt, outk));
    return pq_ma(ctx, b, outk, slowd, slowd_mt, outd);
}
pq_status pq_stochrsi_chain(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, int64_t fastk, int64_t fastd,
                      int64_t fastd_mt, double *outk, double *outd) {
    CHK("pq_stochrsi", real && outk && outd); // momentum.py:197-205
    WS(8);
    PQ_WS_COL(rsi, ctx, b, 0);
    ctx->chain_head = true; // (a small-shard recording: this RSI is what the rest of the chain waits for)
    const pq_status st_rsi = pq_rsi(ctx, b, real, p, rsi);
    ctx->chain_head = false;
    PQ_TRY(st_rsi);
    FastkOp op{}; op.k = fastk;
    PQ_TRY(launch_row(ctx, b, op, InCols<3>{{rsi, rsi, rsi}}, OutColsT<FastkOp, double>{{outk}}));
    return pq_ma(ctx, b, outk, fastd, fastd_mt, outd);
}

} // extern "C"
