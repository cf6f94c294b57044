// wt_api.h -- host entry points of the wave-per-symbol forms (wt.hip).  Each returns true when it has handled the call (launched,
// or recorded into the suite being recorded; *st = the status) and false when the batch / parameters are outside the form's scope
// (a regular batch whose rows are not 16-byte aligned or whose len is outside [1 024, 4 096], a ragged batch whose groups average fewer
// than 1 024 rows or whose longest exceeds 4 096, a period out of range, PQ_NO_WT set): the caller then takes its usual path.  RAGGED
// batches are inside the scope -- any 8-byte group start; wt_all() enables every form for them, the per-lane gather body of the same
// function runs gated behind for groups with a NULL / NaN -- see wt_try (wt.hip).  Output pointers may be null (that output is not
// computed).  wt_try grows ctx->wt_gate on demand: like every entry point, these must not run on one pq_ctx from two host threads at once.
#pragma once
#include "pq_dev.h"

bool wt_ema_all(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *ema, double *dema, double *tema, double *trix, pq_status *st);
bool wt_macd(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t sig, int64_t sig2, double *macd, double *signal,
             double *hist, double *macd2, double *signal2, double *hist2, pq_status *st);
bool wt_rsi(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *rsi, pq_status *st);
bool wt_dm_pair(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *plus_dm, double *minus_dm, pq_status *st);
bool wt_dmi(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *dx, double *plus_di, double *minus_di,
            double *adx, double *adxr, pq_status *st);
bool wt_atr(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *atr, double *natr, pq_status *st);
bool wt_midpoint(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out, pq_status *st);
bool wt_midprice(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *out, pq_status *st);
