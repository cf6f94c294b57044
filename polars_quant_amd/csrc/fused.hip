// fused.hip -- C ABI of the composite functions: one sequential job when the LDS body applies (ops_fused.h),
// the chained launches through scratch columns otherwise (very long windows, unaligned columns, MA types that are
// themselves multi-pass).  Both forms are bit-identical.
#include "ops_fused.h"
#include "wt_api.h"

extern "C" {
pq_status pq_trima_chain(pq_ctx *, const pq_batch *, const double *, int64_t, double *);
pq_status pq_cci_chain(pq_ctx *, const pq_batch *, const double *, const double *, const double *, int64_t, double *);
pq_status pq_adxr_chain(pq_ctx *, const pq_batch *, const double *, const double *, const double *, int64_t, double *);
pq_status pq_apo_chain(pq_ctx *, const pq_batch *, const double *, int64_t, int64_t, int64_t, double *);
pq_status pq_ppo_chain(pq_ctx *, const pq_batch *, const double *, int64_t, int64_t, int64_t, double *);
pq_status pq_macdext_chain(pq_ctx *, const pq_batch *, const double *, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                           double *, double *, double *);
pq_status pq_stoch_chain(pq_ctx *, const pq_batch *, const double *, const double *, const double *, int64_t, int64_t,
                         int64_t, int64_t, int64_t, double *, double *);
pq_status pq_stochf_chain(pq_ctx *, const pq_batch *, const double *, const double *, const double *, int64_t, int64_t,
                          int64_t, double *, double *);
pq_status pq_stochrsi_chain(pq_ctx *, const pq_batch *, const double *, int64_t, int64_t, int64_t, int64_t, double *, double *);
}

#define CHK(name, cond) PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(cond, name ": null pointer")

extern "C" {

pq_status pq_trima(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    CHK("pq_trima", real && out);
    TrimaOp op{};
    if (p % 2 == 1) { op.k1 = p / 2 + 1; op.k2 = op.k1; } else { op.k1 = p / 2; op.k2 = op.k1 + 1; } // overlap.rs:1313-1326
    InCols<1> in{{real}}; OutCols<1> o{{out}};
    if (p > 0 && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_trima_chain(ctx, b, real, p, out);
}
pq_status pq_apo(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t matype, double *out) {
    CHK("pq_apo", real && out);
    MaDiffOp<0> op{}; op.fast = fast; op.slow = slow; op.matype = matype;
    InCols<1> in{{real}}; OutCols<1> o{{out}};
    if (Ma2::supports(matype) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_apo_chain(ctx, b, real, fast, slow, matype, out);
}
pq_status pq_ppo(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t matype, double *out) {
    CHK("pq_ppo", real && out);
    MaDiffOp<1> op{}; op.fast = fast; op.slow = slow; op.matype = matype;
    InCols<1> in{{real}}; OutCols<1> o{{out}};
    if (Ma2::supports(matype) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_ppo_chain(ctx, b, real, fast, slow, matype, out);
}
pq_status pq_macdext(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t fastmt, int64_t slow,
                     int64_t slowmt, int64_t sig, int64_t sigmt, double *macd, double *signal, double *hist) {
    CHK("pq_macdext", real && macd && signal && hist);
    MacdextOp op{}; op.fast = fast; op.fastmt = fastmt; op.slow = slow; op.slowmt = slowmt; op.sig = sig; op.sigmt = sigmt;
    InCols<1> in{{real}}; OutCols<3> o{{macd, signal, hist}};
    if (Ma2::supports(fastmt) && Ma2::supports(slowmt) && Ma2::supports(sigmt) && seq_can_lds(ctx, b, op, in, o))
        return launch_seq(ctx, b, op, in, o);
    return pq_macdext_chain(ctx, b, real, fast, fastmt, slow, slowmt, sig, sigmt, macd, signal, hist);
}
pq_status pq_stoch(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t fastk,
                   int64_t slowk, int64_t slowk_mt, int64_t slowd, int64_t slowd_mt, double *outk, double *outd) {
    CHK("pq_stoch", h && l && c && outk && outd);
    StochOp<0> op{}; op.fastk = fastk; op.p1 = slowk; op.mt1 = slowk_mt; op.p2 = slowd; op.mt2 = slowd_mt;
    InCols<3> in{{h, l, c}}; OutCols<2> o{{outk, outd}};
    if (Ma2::supports(slowk_mt) && Ma2::supports(slowd_mt) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_stoch_chain(ctx, b, h, l, c, fastk, slowk, slowk_mt, slowd, slowd_mt, outk, outd);
}
pq_status pq_stochf(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t fastk,
                    int64_t fastd, int64_t fastd_mt, double *outk, double *outd) {
    CHK("pq_stochf", h && l && c && outk && outd);
    StochOp<1> op{}; op.fastk = fastk; op.p1 = fastd; op.mt1 = fastd_mt; op.p2 = 0; op.mt2 = 0;
    InCols<3> in{{h, l, c}}; OutCols<2> o{{outk, outd}};
    if (Ma2::supports(fastd_mt) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_stochf_chain(ctx, b, h, l, c, fastk, fastd, fastd_mt, outk, outd);
}
pq_status pq_stochrsi(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, int64_t fastk, int64_t fastd,
                      int64_t fastd_mt, double *outk, double *outd) {
    CHK("pq_stochrsi", real && outk && outd);
    StochRsiOp op{}; op.p = p; op.fastk = fastk; op.fastd = fastd; op.fastd_mt = fastd_mt;
    InCols<1> in{{real}}; OutCols<2> o{{outk, outd}};
    if (Ma2::supports(fastd_mt) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_stochrsi_chain(ctx, b, real, p, fastk, fastd, fastd_mt, outk, outd);
}
pq_status pq_cci(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *out) {
    CHK("pq_cci", h && l && c && out);
    CciOp op{}; op.p = p;
    InCols<3> in{{h, l, c}}; OutCols<1> o{{out}};
    if (seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_cci_chain(ctx, b, h, l, c, p, out);
}
pq_status pq_adxr(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *out) {
    CHK("pq_adxr", h && l && c && out);
    { pq_status st; if (wt_dmi(ctx, b, h, l, c, p, nullptr, nullptr, nullptr, nullptr, out, &st)) return st; }
    DmAllOp<false> op{}; op.p = p;
    InCols<3> in{{h, l, c}}; OutCols<1> o{{out}};
    if (seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_adxr_chain(ctx, b, h, l, c, p, out);
}
// calc_dm (momentum.rs:668-727) evaluated once for all five of its users
pq_status pq_dmi_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p,
                     double *dx, double *plus_di, double *minus_di, double *adx, double *adxr) {
    CHK("pq_dmi_all", h && l && c && dx && plus_di && minus_di && adx && adxr);
    { pq_status st; if (wt_dmi(ctx, b, h, l, c, p, dx, plus_di, minus_di, adx, adxr, &st)) return st; }
    DmAllOp<true> op{}; op.p = p;
    InCols<3> in{{h, l, c}}; OutCols<5> o{{dx, plus_di, minus_di, adx, adxr}};
    if (seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_dx(ctx, b, h, l, c, p, dx));
    PQ_TRY(pq_plus_di(ctx, b, h, l, c, p, plus_di));
    PQ_TRY(pq_minus_di(ctx, b, h, l, c, p, minus_di));
    PQ_TRY(pq_adx(ctx, b, h, l, c, p, adx));
    return pq_adxr_chain(ctx, b, h, l, c, p, adxr);
}
// the shared Hilbert pipeline (cycle.rs:27-63) evaluated once for ht_dcperiod / ht_dcphase / ht_phasor / ht_sine
pq_status pq_ht_all(pq_ctx *ctx, const pq_batch *b, const double *real, double *dcperiod, double *dcphase, double *inphase,
                    double *quadrature, double *sine, double *leadsine) {
    CHK("pq_ht_all", real && dcperiod && dcphase && inphase && quadrature && sine && leadsine);
    {   // tiled body: one job, the derived columns leave through its storer wave
        HtAll6Op op6{};
        op6.der[0] = dcphase; op6.der[1] = sine; op6.der[2] = leadsine;
        InCols<1> in{{real}}; OutCols<3> o3{{dcperiod, inphase, quadrature}};
        const double *nodep[1] = {real};
        double *ders[3] = {dcphase, sine, leadsine};
        if (b->len % SeqTile<HtAll6Op>::K == 0 && seq_can_lds(ctx, b, op6, in, o3) && seq_cols_aligned<1, 3>(b, nodep, ders)) return launch_seq(ctx, b, op6, in, o3);
    }
    PQ_TRY(launch_seq(ctx, b, HtAllOp{}, InCols<1>{{real}}, OutCols<3>{{dcperiod, inphase, quadrature}}));
    // (in a recorded suite the ROW launch reads what the job writes, so it lands in the next phase)
    return launch_row(ctx, b, HtPhaseSineOp{}, InCols<2>{{inphase, quadrature}}, OutColsT<HtPhaseSineOp, double>{{dcphase, sine, leadsine}});
}

// ---- multi-output forms: several reference functions over the same inputs as ONE job (bit-identical columns) ----
pq_status pq_ema_all(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *ema, double *dema, double *tema, double *trix) {
    CHK("pq_ema_all", real && ema && dema && tema && trix);
    { pq_status st; if (wt_ema_all(ctx, b, real, p, ema, dema, tema, trix, &st)) return st; } // one symbol per wavefront (ops_wt.h)
    EmaAllOp op{};
    op.a.a.p = p; op.a.b.p = p; op.b.a.p = p; op.b.b.p = p;
    InCols<1> in{{real}}; OutCols<4> o{{ema, dema, tema, trix}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_ema(ctx, b, real, p, ema)); PQ_TRY(pq_dema(ctx, b, real, p, dema)); PQ_TRY(pq_tema(ctx, b, real, p, tema));
    return pq_trix(ctx, b, real, p, trix);
}
pq_status pq_atr_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *atr, double *natr) {
    CHK("pq_atr_all", h && l && c && atr && natr);
    { pq_status st; if (wt_atr(ctx, b, h, l, c, p, atr, natr, &st)) return st; }
    AtrAllOp op{}; op.a.p = p; op.b.p = p;
    InCols<3> in{{h, l, c}}; OutCols<2> o{{atr, natr}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_atr(ctx, b, h, l, c, p, atr));
    return pq_natr(ctx, b, h, l, c, p, natr);
}
pq_status pq_dm_pair(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *plus_dm, double *minus_dm) {
    CHK("pq_dm_pair", h && l && plus_dm && minus_dm);
    { pq_status st; if (wt_dm_pair(ctx, b, h, l, p, plus_dm, minus_dm, &st)) return st; }
    DmPairOp op{}; op.a.p = p; op.b.p = p;
    InCols<2> in{{h, l}}; OutCols<2> o{{plus_dm, minus_dm}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_plus_dm(ctx, b, h, l, p, plus_dm));
    return pq_minus_dm(ctx, b, h, l, p, minus_dm);
}
pq_status pq_ad_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, const double *v, int64_t fast,
                    int64_t slow, double *ad, double *adosc) {
    CHK("pq_ad_all", h && l && c && v && ad && adosc);
    AdAllOp op{}; op.a.fast = op.a.slow = 0; op.b.fast = fast; op.b.slow = slow;
    InCols<4> in{{h, l, c, v}}; OutCols<2> o{{ad, adosc}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_ad(ctx, b, h, l, c, v, ad));
    return pq_adosc(ctx, b, h, l, c, v, fast, slow, adosc);
}
pq_status pq_macd_pair(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t sig, int64_t fix_sig,
                       double *macd, double *signal, double *hist, double *fmacd, double *fsignal, double *fhist) {
    CHK("pq_macd_pair", real && macd && signal && hist && fmacd && fsignal && fhist);
    { pq_status st; if (wt_macd(ctx, b, real, fast, slow, sig, fix_sig, macd, signal, hist, fmacd, fsignal, fhist, &st)) return st; }
    MacdPairOp op{};
    op.a.fast = fast; op.a.slow = slow; op.a.sig = sig; op.b.fast = 12; op.b.slow = 26; op.b.sig = fix_sig; // momentum.py:90-92
    InCols<1> in{{real}}; OutCols<6> o{{macd, signal, hist, fmacd, fsignal, fhist}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_macd(ctx, b, real, fast, slow, sig, macd, signal, hist));
    return pq_macdfix(ctx, b, real, fix_sig, fmacd, fsignal, fhist);
}
pq_status pq_sar_pair(pq_ctx *ctx, const pq_batch *b, const double *high, const double *low, double accel, double maxv, double startvalue,
                      double offsetonreverse, double ai_long, double a_long, double am_long, double ai_short, double a_short,
                      double am_short, double *sar, double *sarext) {
    CHK("pq_sar_pair", high && low && sar && sarext);
    SarPairOp op{};
    op.a.ext = false; op.a.startvalue = 0.0; op.a.offset = 0.0;
    op.a.ai_long = op.a.a_long = op.a.ai_short = op.a.a_short = accel; op.a.am_long = op.a.am_short = maxv;
    op.b.ext = true; op.b.startvalue = startvalue; op.b.offset = offsetonreverse;
    op.b.ai_long = ai_long; op.b.a_long = a_long; op.b.am_long = am_long; op.b.ai_short = ai_short; op.b.a_short = a_short; op.b.am_short = am_short;
    InCols<2> in{{high, low}}; OutCols<2> o{{sar, sarext}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_sar(ctx, b, high, low, accel, maxv, sar));
    return pq_sarext(ctx, b, high, low, startvalue, offsetonreverse, ai_long, a_long, am_long, ai_short, a_short, am_short, sarext);
}
pq_status pq_volume_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, const double *v, int64_t mfi_p,
                        int64_t fast, int64_t slow, double *mfi, double *ad, double *adosc, double *obv) {
    CHK("pq_volume_all", h && l && c && v && mfi && ad && adosc && obv);
    VolumeAllOp op{};
    op.a.p = mfi_p; op.b.a.a.fast = op.b.a.a.slow = 0; op.b.a.b.fast = fast; op.b.a.b.slow = slow;
    InCols<4> in{{h, l, c, v}}; OutCols<4> o{{mfi, ad, adosc, obv}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_mfi(ctx, b, h, l, c, v, mfi_p, mfi)); PQ_TRY(pq_ad(ctx, b, h, l, c, v, ad));
    PQ_TRY(pq_adosc(ctx, b, h, l, c, v, fast, slow, adosc));
    return pq_obv(ctx, b, c, v, obv);
}
pq_status pq_dm_system_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *dx,
                           double *plus_di, double *minus_di, double *adx, double *adxr, double *atr, double *natr) {
    CHK("pq_dm_system_all", h && l && c && dx && plus_di && minus_di && adx && adxr && atr && natr);
    {   // wave-per-symbol form: two jobs (the DI / DX / ADX chains need three LDS columns, ATR / NATR one)
        pq_status st;
        if (wt_dmi(ctx, b, h, l, c, p, dx, plus_di, minus_di, adx, adxr, &st)) {
            PQ_TRY(st);
            return pq_atr_all(ctx, b, h, l, c, p, atr, natr);
        }
    }
    DmiAtrOp op{}; op.a.p = p; op.b.a.p = p; op.b.b.p = p;
    InCols<3> in{{h, l, c}}; OutCols<7> o{{dx, plus_di, minus_di, adx, adxr, atr, natr}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_dmi_all(ctx, b, h, l, c, p, dx, plus_di, minus_di, adx, adxr));
    return pq_atr_all(ctx, b, h, l, c, p, atr, natr);
}
pq_status pq_sma_ma(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *sma, double *ma) {
    CHK("pq_sma_ma", real && sma && ma);
    SmaDupOp op{}; op.a.p = p;
    InCols<1> in{{real}}; OutCols<2> o{{sma, ma}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_sma(ctx, b, real, p, sma));
    return pq_ma(ctx, b, real, p, 0, ma);
}
pq_status pq_cmo_rsi(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *cmo, double *rsi) {
    CHK("pq_cmo_rsi", real && cmo && rsi);
    {   // RSI's Wilder averages contract (wave-per-symbol form); CMO's rolling sums do not: it stays a lane-per-symbol job
        pq_status st;
        if (wt_rsi(ctx, b, real, p, rsi, &st)) { PQ_TRY(st); return pq_cmo(ctx, b, real, p, cmo); }
    }
    CmoRsiOp op{}; op.a.p = p; op.b.p = p;
    InCols<1> in{{real}}; OutCols<2> o{{cmo, rsi}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_cmo(ctx, b, real, p, cmo));
    return pq_rsi(ctx, b, real, p, rsi);
}
pq_status pq_stoch_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t fastk, int64_t slowk,
                       int64_t slowk_mt, int64_t slowd, int64_t slowd_mt, int64_t fastd, int64_t fastd_mt, double *slowk_out,
                       double *slowd_out, double *fastk_out, double *fastd_out) {
    CHK("pq_stoch_all", h && l && c && slowk_out && slowd_out && fastk_out && fastd_out);
    StochAllOp op{};
    op.fastk = fastk; op.slowk = slowk; op.slowk_mt = slowk_mt; op.slowd = slowd; op.slowd_mt = slowd_mt; op.fastd = fastd; op.fastd_mt = fastd_mt;
    InCols<3> in{{h, l, c}}; OutCols<4> o{{slowk_out, slowd_out, fastk_out, fastd_out}};
    if (PQ_FUSE_OK(ctx) && Ma2::supports(slowk_mt) && Ma2::supports(slowd_mt) && Ma2::supports(fastd_mt) && seq_can_lds(ctx, b, op, in, o))
        return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_stoch(ctx, b, h, l, c, fastk, slowk, slowk_mt, slowd, slowd_mt, slowk_out, slowd_out));
    return pq_stochf(ctx, b, h, l, c, fastk, fastd, fastd_mt, fastk_out, fastd_out);
}
pq_status pq_apo_ppo(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t matype, double *apo, double *ppo) {
    CHK("pq_apo_ppo", real && apo && ppo);
    ApoPpoOp op{};
    op.a.fast = fast; op.a.slow = slow; op.a.matype = matype; op.b.fast = fast; op.b.slow = slow; op.b.matype = matype;
    InCols<1> in{{real}}; OutCols<2> o{{apo, ppo}};
    if (PQ_FUSE_OK(ctx) && Ma2::supports(matype) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_apo(ctx, b, real, fast, slow, matype, apo));
    return pq_ppo(ctx, b, real, fast, slow, matype, ppo);
}

} // extern "C"
