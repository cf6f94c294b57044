// wt.hip -- the wave-per-symbol indicator kernels (wt_dev.h, ops_wt.h) and their host entry points: which lane-per-symbol op
// stands gated behind each launch, and when the form applies.
#include "ops_fused.h"
#include "ops_wt.h"
#include "wt_api.h"

static inline bool wt_period_ok(int64_t p) { return p >= 1 && p <= 1024; }
// Which forms are used by default: those that are faster than the lane-per-symbol body on their own at 5 000 x 2 520
// (scripts/bench_wt.py, profiles/r04_bench_wt.json).  The others -- DEMA / TEMA alone, MACD alone, the DI / DX / ADX family, whose three
// LDS columns leave two waves per CU -- are built and tested (PQ_WT_ALL=1 selects them) but lose to the body they would replace.
// On RAGGED batches every form wins: the alternative there is the per-lane gather body (1.1 ... 11 x slower, profiles/r04_bench_wt.json).
static inline bool wt_all(const pq_batch *b) { return b->offsets != nullptr || getenv("PQ_WT_ALL") != nullptr; }

bool wt_ema_all(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *ema, double *dema, double *tema, double *trix, pq_status *st) {
    if (!wt_period_ok(p)) return false;
    WtEmaAllOp w{real, ema, dema, tema, trix, (int32_t)p};
    InCols<1> in{{real}};
    const int n = (ema != nullptr) + (dema != nullptr) + (tema != nullptr) + (trix != nullptr);
    // DEMA / TEMA at timeperiod 1: every seeding branch of the reference's if-chain (overlap.rs:590-597, :1238-1240) falls on the first
    // row and only the first is taken -- the later averages start from 0.0 instead of being seeded, and TEMA's first row is null.  The
    // lane-per-symbol ops restate the chain literally; this form seeds every level properly, so it leaves p = 1 to them.
    if ((dema || tema) && p < 2) return false;
    if (n == 4) {
        EmaAllOp op{};
        op.a.a.p = p; op.a.b.p = p; op.b.a.p = p; op.b.b.p = p;
        return wt_try(ctx, b, w, op, in, OutCols<4>{{ema, dema, tema, trix}}, st);
    }
    if (n != 1) return false;
    if (ema) { EmaOp op{}; op.p = p; return wt_try(ctx, b, WtEmaOp{real, ema, (int32_t)p}, op, in, OutCols<1>{{ema}}, st); }
    if (!wt_all(b) && !trix) return false; // DEMA / TEMA alone: 0.26 ms against 0.16 ms
    if (dema) { DemaOp op{}; op.p = p; return wt_try(ctx, b, w, op, in, OutCols<1>{{dema}}, st); }
    if (tema) { TemaOp op{}; op.p = p; return wt_try(ctx, b, w, op, in, OutCols<1>{{tema}}, st); }
    TrixOp op{}; op.p = p;
    return wt_try(ctx, b, WtTrixOp{real, trix, (int32_t)p}, op, in, OutCols<1>{{trix}}, st);
}

bool wt_macd(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t sig, int64_t sig2, double *macd, double *signal,
             double *hist, double *macd2, double *signal2, double *hist2, pq_status *st) {
    if (!wt_period_ok(fast) || !wt_period_ok(slow) || !wt_period_ok(sig) || !macd || !signal || !hist) return false;
    InCols<1> in{{real}};
    if (macd2 || signal2 || hist2) { // the pair: MACDFIX rides on this call's two averages only when they are its fixed 12 / 26
        if (!(macd2 && signal2 && hist2) || fast != 12 || slow != 26 || !wt_period_ok(sig2)) return false;
        WtMacdOp w{real, macd, signal, hist, macd2, signal2, hist2, (int32_t)fast, (int32_t)slow, (int32_t)sig, (int32_t)sig2};
        MacdPairOp op{};
        op.a.fast = fast; op.a.slow = slow; op.a.sig = sig; op.b.fast = 12; op.b.slow = 26; op.b.sig = sig2;
        return wt_try(ctx, b, w, op, in, OutCols<6>{{macd, signal, hist, macd2, signal2, hist2}}, st);
    }
    if (!wt_all(b)) return false; // MACD alone: 0.29 ms against 0.27 ms (the pair with MACDFIX: 0.31 against 0.40)
    WtMacdOp w{real, macd, signal, hist, nullptr, nullptr, nullptr, (int32_t)fast, (int32_t)slow, (int32_t)sig, (int32_t)sig};
    MacdOp op{}; op.fast = fast; op.slow = slow; op.sig = sig;
    return wt_try(ctx, b, w, op, in, OutCols<3>{{macd, signal, hist}}, st);
}

bool wt_rsi(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *rsi, pq_status *st) {
    if (!wt_period_ok(p)) return false;
    WtRsiOp w{real, rsi, (int32_t)p};
    RsiOp op{}; op.p = p;
    return wt_try(ctx, b, w, op, InCols<1>{{real}}, OutCols<1>{{rsi}}, st);
}

bool wt_dm_pair(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *plus_dm, double *minus_dm, pq_status *st) {
    if (!wt_period_ok(p)) return false;
    WtDmPairOp w{h, l, plus_dm, minus_dm, (int32_t)p};
    InCols<2> in{{h, l}};
    if (plus_dm && minus_dm) { DmPairOp op{}; op.a.p = p; op.b.p = p; return wt_try(ctx, b, w, op, in, OutCols<2>{{plus_dm, minus_dm}}, st); }
    if (plus_dm) { DmRawOp<true> op{}; op.p = p; return wt_try(ctx, b, WtDmRawOp<true>{h, l, plus_dm, (int32_t)p}, op, in, OutCols<1>{{plus_dm}}, st); }
    if (minus_dm) { DmRawOp<false> op{}; op.p = p; return wt_try(ctx, b, WtDmRawOp<false>{h, l, minus_dm, (int32_t)p}, op, in, OutCols<1>{{minus_dm}}, st); }
    return false;
}

bool wt_dmi(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *dx, double *plus_di, double *minus_di,
            double *adx, double *adxr, pq_status *st) {
    if (!wt_period_ok(p) || !wt_all(b)) return false; // 1.35 ms against 1.20 ms for the five columns, 1.28 against 0.89 for ADX alone
    WtDmiOp w{h, l, c, dx, plus_di, minus_di, adx, adxr, (int32_t)p};
    InCols<3> in{{h, l, c}};
    const int n = (dx != nullptr) + (plus_di != nullptr) + (minus_di != nullptr) + (adx != nullptr) + (adxr != nullptr);
    if (n == 5) { DmAllOp<true> op{}; op.p = p; return wt_try(ctx, b, w, op, in, OutCols<5>{{dx, plus_di, minus_di, adx, adxr}}, st); }
    if (n != 1) return false;
    if (dx || plus_di) { DmOp<0> op{}; op.p = p; return wt_try(ctx, b, w, op, in, OutCols<1>{{dx ? dx : plus_di}}, st); } // plus_di = dx (D-5)
    if (minus_di) { DmOp<1> op{}; op.p = p; return wt_try(ctx, b, w, op, in, OutCols<1>{{minus_di}}, st); }
    if (adx) { DmOp<2> op{}; op.p = p; return wt_try(ctx, b, w, op, in, OutCols<1>{{adx}}, st); }
    DmAllOp<false> op{}; op.p = p;
    return wt_try(ctx, b, w, op, in, OutCols<1>{{adxr}}, st);
}

bool wt_atr(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *atr, double *natr, pq_status *st) {
    if (!wt_period_ok(p)) return false;
    WtAtrOp w{h, l, c, atr, natr, (int32_t)p};
    InCols<3> in{{h, l, c}};
    if (atr && natr) { AtrAllOp op{}; op.a.p = p; op.b.p = p; return wt_try(ctx, b, w, op, in, OutCols<2>{{atr, natr}}, st); }
    if (atr) { AtrOp<false> op{}; op.p = p; return wt_try(ctx, b, w, op, in, OutCols<1>{{atr}}, st); }
    if (natr) { AtrOp<true> op{}; op.p = p; return wt_try(ctx, b, w, op, in, OutCols<1>{{natr}}, st); }
    return false;
}

bool wt_midpoint(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out, pq_status *st) {
    if (p < 1 || p > 256) return false;
    WtMidpointOp w{real, out, (int32_t)p};
    MidpointOp op{}; op.p = p;
    return wt_try(ctx, b, w, op, InCols<1>{{real}}, OutCols<1>{{out}}, st);
}
bool wt_midprice(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *out, pq_status *st) {
    if (p < 1 || p > 256) return false;
    WtMidpriceOp w{h, l, out, (int32_t)p};
    MidpriceOp op{}; op.p = p;
    return wt_try(ctx, b, w, op, InCols<2>{{h, l}}, OutCols<1>{{out}}, st);
}

extern "C" {
// [0] symbols computed by the wave-per-symbol kernels since the last reset, [1] speculative chunks that failed the bit test,
// [2] chunk re-runs, [3] symbols handed to the gated lane-per-symbol path (NULL / NaN inputs)
pq_status pq_wt_stats(pq_ctx *ctx, int64_t *out4, int32_t reset) {
    PQ_REQUIRE(ctx && out4, "pq_wt_stats: null pointer");
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    PQ_HIP_TRY(hipMemcpy(out4, ctx->d_flag + 24, 4 * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (reset) PQ_HIP_TRY(hipMemset(ctx->d_flag + 24, 0, 4 * sizeof(int64_t)));
    return PQ_OK;
}
#ifdef PQ_WT_PROF // profiling builds (scripts/prof_wt.py): shader clocks per phase, summed over the waves
pq_status pq_wt_prof(pq_ctx *ctx, int64_t *out16, int32_t reset) {
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    PQ_HIP_TRY(hipMemcpy(out16, ctx->d_flag + 32, 16 * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (reset) PQ_HIP_TRY(hipMemset(ctx->d_flag + 32, 0, 16 * sizeof(int64_t)));
    return PQ_OK;
}
#endif
}
