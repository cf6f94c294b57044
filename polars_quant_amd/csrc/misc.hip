// misc.hip -- kernels + C ABI for volatility / volume / price transforms and the Hilbert-transform
// cycle indicators (reference: src/talib/{volatility,volume,price,cycle}.rs) plus MAMA (D-4).
#include "ops_misc.h"
#include "wt_api.h"

// ---------------------------------------------------------------- C ABI
#define CHK(name, cond) PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(cond, name ": null pointer")
extern "C" {

pq_status pq_avgprice(pq_ctx *ctx, const pq_batch *b, const double *o, const double *h, const double *l,
                      const double *c, double *out) {
    CHK("pq_avgprice", o && h && l && c && out);
    return launch_row(ctx, b, PriceOp<0>{}, InCols<4>{{o, h, l, c}}, OutColsT<PriceOp<0>, double>{{out}});
}
pq_status pq_medprice(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, double *out) {
    CHK("pq_medprice", h && l && out);
    return launch_row(ctx, b, PriceOp<1>{}, InCols<2>{{h, l}}, OutColsT<PriceOp<1>, double>{{out}});
}
pq_status pq_typprice(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, double *out) {
    CHK("pq_typprice", h && l && c && out);
    return launch_row(ctx, b, PriceOp<2>{}, InCols<3>{{h, l, c}}, OutColsT<PriceOp<2>, double>{{out}});
}
pq_status pq_wclprice(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, double *out) {
    CHK("pq_wclprice", h && l && c && out);
    return launch_row(ctx, b, PriceOp<3>{}, InCols<3>{{h, l, c}}, OutColsT<PriceOp<3>, double>{{out}});
}
pq_status pq_trange(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, double *out) {
    CHK("pq_trange", h && l && c && out);
    return launch_row(ctx, b, TrangeOp{}, InCols<3>{{h, l, c}}, OutColsT<TrangeOp, double>{{out}});
}
pq_status pq_atr(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p,
                 double *out) {
    CHK("pq_atr", h && l && c && out);
    { pq_status st; if (wt_atr(ctx, b, h, l, c, p, out, nullptr, &st)) return st; }
    AtrOp<false> op{}; op.p = p;
    return launch_seq(ctx, b, op, InCols<3>{{h, l, c}}, OutCols<1>{{out}});
}
pq_status pq_natr(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p,
                  double *out) {
    CHK("pq_natr", h && l && c && out);
    { pq_status st; if (wt_atr(ctx, b, h, l, c, p, nullptr, out, &st)) return st; }
    AtrOp<true> op{}; op.p = p;
    return launch_seq(ctx, b, op, InCols<3>{{h, l, c}}, OutCols<1>{{out}});
}
pq_status pq_ad(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, const double *v,
                double *out) {
    CHK("pq_ad", h && l && c && v && out);
    AdOp<false> op{}; op.fast = op.slow = 0;
    return launch_seq(ctx, b, op, InCols<4>{{h, l, c, v}}, OutCols<1>{{out}});
}
pq_status pq_adosc(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, const double *v,
                   int64_t fast, int64_t slow, double *out) {
    CHK("pq_adosc", h && l && c && v && out);
    AdOp<true> op{}; op.fast = fast; op.slow = slow;
    return launch_seq(ctx, b, op, InCols<4>{{h, l, c, v}}, OutCols<1>{{out}});
}
pq_status pq_obv(pq_ctx *ctx, const pq_batch *b, const double *c, const double *v, double *out) {
    CHK("pq_obv", c && v && out);
    return launch_seq(ctx, b, ObvOp{}, InCols<2>{{c, v}}, OutCols<1>{{out}});
}
pq_status pq_ht_dcperiod(pq_ctx *ctx, const pq_batch *b, const double *real, double *out) {
    CHK("pq_ht_dcperiod", real && out);
    return launch_seq(ctx, b, HtOp<0>{}, InCols<1>{{real}}, OutCols<1>{{out}});
}
pq_status pq_ht_dcphase(pq_ctx *ctx, const pq_batch *b, const double *real, double *out) {
    CHK("pq_ht_dcphase", real && out);
    return launch_seq(ctx, b, HtOp<1>{}, InCols<1>{{real}}, OutCols<1>{{out}});
}
pq_status pq_ht_phasor(pq_ctx *ctx, const pq_batch *b, const double *real, double *inphase, double *quadrature) {
    CHK("pq_ht_phasor", real && inphase && quadrature);
    return launch_seq(ctx, b, HtOp<2>{}, InCols<1>{{real}}, OutCols<2>{{inphase, quadrature}});
}
pq_status pq_ht_sine(pq_ctx *ctx, const pq_batch *b, const double *real, double *sine, double *leadsine) {
    CHK("pq_ht_sine", real && sine && leadsine);
    return launch_seq(ctx, b, HtOp<3>{}, InCols<1>{{real}}, OutCols<2>{{sine, leadsine}});
}
pq_status pq_mama(pq_ctx *ctx, const pq_batch *b, const double *real, double fastlimit, double slowlimit,
                  double *mama, double *fama) {
    CHK("pq_mama", real && mama && fama);
    HtOp<4> op{}; op.fastlimit = fastlimit; op.slowlimit = slowlimit;
    return launch_seq(ctx, b, op, InCols<1>{{real}}, OutCols<2>{{mama, fama}});
}
pq_status pq_ht_trendline(pq_ctx *ctx, const pq_batch *b, const double *real, double *out) {
    CHK("pq_ht_trendline", real && out);
    return launch_row(ctx, b, TrendlineOp{}, InCols<1>{{real}}, OutColsT<TrendlineOp, double>{{out}});
}
pq_status pq_ht_trendmode(pq_ctx *ctx, const pq_batch *b, const double *real, int32_t *out) {
    CHK("pq_ht_trendmode", real && out);
    return launch_row(ctx, b, TrendmodeOp{}, InCols<1>{{real}}, OutColsT<TrendmodeOp, int32_t>{{out}});
}

} // extern "C"
