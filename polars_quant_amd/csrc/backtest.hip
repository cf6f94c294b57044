// backtest.hip -- per-symbol independent-capital-pool backtest (reference: src/backtest/vectorized.rs:69-224)
// fused with its summary reduction (src/backtest/metrics.rs:7-152) and, optionally, with the MACD-cross
// signal generation in front of it (SURVEY 8(f) rank 2 / decision D-8).
// SEQ shape: one symbol per lane; the scan is branchy and strictly serial in time, so the parallel
// axis is the symbol axis only.  The lane writes position/cash/equity as it goes and then re-reads its
// own equity (and benchmark) rows for the two-pass mean/variance exactly as calculate_summary does.
#include "ops_backtest_wave.h"
#include <stdlib.h>

template <bool M, bool S>
static pq_status bt_launch(pq_ctx *ctx, const pq_batch *b, const BtArgs &a) {
    if (b->n_series == 0) return PQ_OK;
    if (ctx->rec) {
        if (S) { pq_set_error("signal generation cannot be recorded into a suite"); return PQ_ERR_UNSUPPORTED; }
        return rec_add_backtest(ctx, b, SEQ_ID_BACKTEST + (M ? 1 : 0), a);
    }
    dim3 grid((unsigned)((b->n_series + SEQ_BLOCK - 1) / SEQ_BLOCK));
    hipLaunchKernelGGL((backtest_kernel<M, S>), grid, dim3(SEQ_BLOCK), 0, ctx->stream, a, dims_of(b));
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}

// ---- wave-per-symbol form (ops_backtest_wave.h): one workgroup of one wavefront per symbol
struct BtWaveBlob {
    BtWaveArgs a;
    pq_batch b;
    unsigned lds;
    int macd;
    int waves; // wavefronts per symbol: 1, or 4 for small batches
};
static void btw_launch_blob(const void *blob, hipStream_t stream) {
    const BtWaveBlob &w = *reinterpret_cast<const BtWaveBlob *>(blob);
    // above 64 KB of dynamic LDS a kernel has to opt in; the attribute belongs to the CURRENT device's copy of the function, so it
    // is set on every such launch (a host-side table write) rather than remembered in a process-wide flag
    using Kern = void (*)(BtWaveArgs, Dims);
    const bool two = w.a.C > 64; // chunks of more than 64 rows (len > 4096): two mask words per lane
    // Four waves per symbol share the staging, the fill and the summary (ops_backtest_wave.h, btw_helper_wave): a small batch (a
    // 625-symbol shard of config 3 on 8 GPUs) leaves most SIMDs with one wave, whose instruction count is then the kernel's time
    const bool multi = !two && w.waves > 1;
    const Kern kern = multi ? (w.macd ? &bt_wave_kernel<true, 1, 4> : &bt_wave_kernel<false, 1, 4>)
                            : w.macd ? (two ? &bt_wave_kernel<true, 2> : &bt_wave_kernel<true, 1>) : (two ? &bt_wave_kernel<false, 2> : &bt_wave_kernel<false, 1>);
    if (w.lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return; // the launch below would fail with a less specific error; hipGetLastError reports this one
    hipLaunchKernelGGL(kern, dim3((unsigned)w.b.n_series), dim3(multi ? 256 : 64), w.lds, stream, w.a, dims_of(&w.b));
}
// true: handled (launched or recorded, *st holds the status); false: the shape is outside the wave form
static bool bt_wave(pq_ctx *ctx, const pq_batch *b, bool macd, const BtArgs &g, pq_status *st) {
    if (getenv("PQ_BT_LANE_FORM")) return false; // A/B and tests: force the lane-per-symbol form
    BtWaveBlob w{};
    size_t lds = 0;
    if (b->n_series > 0x7fffffffLL || !btw_plan(b, g.fast, g.slow, g.sig, macd, w.a, lds, g.bench != nullptr)) return false;
    *st = PQ_OK;
    if (b->n_series == 0) return true;
    w.a.price = g.price; w.a.buy = g.buy; w.a.sell = g.sell; w.a.bench = g.bench;
    w.a.position = g.position; w.a.cash = g.cash; w.a.equity = g.equity; w.a.summary = g.summary;
    w.a.prm = g.prm; w.a.fast = (int32_t)g.fast; w.a.slow = (int32_t)g.slow; w.a.sig = (int32_t)g.sig;
    w.a.stats = reinterpret_cast<unsigned long long *>(ctx->d_flag) + 4;
    w.b = *b; w.lds = (unsigned)lds; w.macd = macd ? 1 : 0;
    {   // Four waves per symbol (ops_backtest_wave.h, btw_helper_wave) are faster at every batch size measured -- 256 symbols 42 against
        // 53 us, 625: 51 / 58, 2 500: 161 / 178, 5 000: 285 / 313 -- except just above what one round of four-wave workgroups holds
        // (four per CU by registers): 1 250 symbols 103 against 96 us, where the one-wave form still fits a single round (six per CU by
        // LDS).  PQ_BT_WAVES=1 / 4: A/B runs and tests.
        const char *e = getenv("PQ_BT_WAVES");
        const int cus = ctx->cus > 0 ? ctx->cus : 256;
        w.waves = e ? atoi(e) : ((b->n_series > (int64_t)cus * 4 && b->n_series <= (int64_t)cus * 5) ? 1 : 4);
        // recorded into a suite the kernel runs in the tail of a step beside the job grids, where the waiting helper waves cost the
        // other kernels wave slots and registers: 4.02 against 3.85 ms per step -- one wave per symbol there
        if (ctx->rec && !e) w.waves = 1;
        if (w.waves != 4) w.waves = 1;
    }
    if (ctx->rec) {
        static_assert(sizeof(BtWaveBlob) <= sizeof(RowThunk::blob), "wave backtest blob too large");
        RowThunk t{};
        t.launch = &btw_launch_blob;
        t.row_id = 0;
        t.blob_bytes = (int)sizeof w;
        t.dims = dims_of(b);
        memcpy(t.blob, &w, sizeof w);
        const void *rd[4] = {g.price, g.buy, g.sell, g.bench};
        for (int k = 0; k < 4; k++) if (rd[k]) t.reads[t.n_reads++] = rd[k];
        void *wr[4] = {g.position, g.cash, g.equity, g.summary};
        for (int k = 0; k < 4; k++) if (wr[k]) t.writes[t.n_writes++] = wr[k];
        *st = rec_add_row(ctx, t);
        return true;
    }
    btw_launch_blob(&w, ctx->stream);
    if (hipGetLastError() != hipSuccess) { pq_set_error("wave backtest launch failed"); *st = PQ_ERR_HIP; }
    return true;
}

struct LevWaveBlob {
    LevWaveArgs w;
    pq_batch b;
    unsigned lds;
};
static void lev_wave_launch_blob(const void *blob, hipStream_t stream) {
    const LevWaveBlob &lb = *reinterpret_cast<const LevWaveBlob *>(blob);
    if (lb.lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&lev_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return;
    hipLaunchKernelGGL(lev_wave_kernel, dim3((unsigned)lb.b.n_series), dim3(64), lb.lds, stream, lb.w, dims_of(&lb.b));
}
// the leveraged engine, one symbol per wavefront (len <= 8192); true: handled
static bool lev_wave(pq_ctx *ctx, const pq_batch *b, const LevArgs &a, pq_status *st) {
    if (getenv("PQ_BT_LANE_FORM") || b->len > 64 * BTW_MAX_C || b->n_series > 0x7fffffffLL) return false;
    LevWaveBlob lb{};
    BtWaveArgs plan{};
    size_t lds = 0;
    btw_plan(b, 0, 0, 0, false, plan, lds, false);
    lb.w.a = a; lb.w.C = plan.C; lb.w.P = plan.P; lb.w.magic = plan.magic;
    lb.b = *b;
    lb.lds = (unsigned)((size_t)64 * plan.P * 8);
    *st = PQ_OK;
    if (ctx->rec) {
        static_assert(sizeof(LevWaveBlob) <= sizeof(RowThunk::blob), "leveraged wave blob too large");
        RowThunk t{};
        t.launch = &lev_wave_launch_blob;
        t.blob_bytes = (int)sizeof lb;
        t.dims = dims_of(b);
        memcpy(t.blob, &lb, sizeof lb);
        const void *rd[4] = {a.price, a.buy, a.sell, a.bench};
        for (int k = 0; k < 4; k++) if (rd[k]) t.reads[t.n_reads++] = rd[k];
        void *wr[] = {a.cash_net, a.stock_value, a.total_value, a.summary, a.trade_count, a.entry_day, a.exit_day, a.reason, a.entry_price,
                      a.exit_price, a.quantity, a.pnl, a.pnl_pct};
        for (void *q : wr) if (q) t.writes[t.n_writes++] = q;
        *st = rec_add_row(ctx, t);
        return true;
    }
    lev_wave_launch_blob(&lb, ctx->stream);
    if (hipGetLastError() != hipSuccess) { pq_set_error("leveraged wave backtest launch failed"); *st = PQ_ERR_HIP; }
    return true;
}

extern "C" {

// [0] symbols run by the wave form since the last reset, [1] speculative chunks that failed the bit test, [2] chunk re-runs
pq_status pq_backtest_wave_stats(pq_ctx *ctx, int64_t *out3, int32_t reset) {
    PQ_REQUIRE(ctx && out3, "pq_backtest_wave_stats: null pointer");
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    PQ_HIP_TRY(hipMemcpy(out3, ctx->d_flag + 4, 3 * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (reset) PQ_HIP_TRY(hipMemset(ctx->d_flag + 4, 0, 3 * sizeof(int64_t)));
    return PQ_OK;
}
#ifdef PQ_BTW_PROF
pq_status pq_backtest_wave_prof(pq_ctx *ctx, int64_t *out5, int32_t reset) {
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    PQ_HIP_TRY(hipMemcpy(out5, ctx->d_flag + 8, 16 * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (reset) PQ_HIP_TRY(hipMemset(ctx->d_flag + 8, 0, 16 * sizeof(int64_t)));
    return PQ_OK;
}
#endif

pq_status pq_backtest_vectorized(pq_ctx *ctx, const pq_batch *b, const double *price, const uint8_t *buy,
                                 const uint8_t *sell, const double *benchmark, const pq_bt_params *params,
                                 double *position, double *cash, double *equity, double *summary) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(price && buy && sell && params, "pq_backtest_vectorized: null pointer");
    BtArgs a{};
    a.price = price; a.buy = buy; a.sell = sell; a.bench = benchmark;
    a.position = position; a.cash = cash; a.summary = summary; a.prm = *params;
    a.equity = equity;
    pq_status wst;
    if (bt_wave(ctx, b, false, a, &wst)) return wst; // one symbol per wavefront (len <= 8192)
    if (!equity) {
        PQ_TRY(pq_ws_reserve(ctx, sizeof(double) * batch_rows(b) * 8));
        a.equity = pq_ws_col(ctx, b, 0);
        if (!a.equity) { pq_set_error("out of device memory for a scratch column"); return PQ_ERR_NOMEM; }
    }
    return bt_launch<false, false>(ctx, b, a);
}

pq_status pq_backtest_macd_cross(pq_ctx *ctx, const pq_batch *b, const double *close, int64_t fast, int64_t slow,
                                 int64_t sig, const pq_bt_params *params, double *position, double *cash,
                                 double *equity, double *summary) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(close && params, "pq_backtest_macd_cross: null pointer");
    BtArgs a{};
    a.price = close; a.position = position; a.cash = cash; a.summary = summary; a.prm = *params;
    a.fast = fast; a.slow = slow; a.sig = sig;
    a.equity = equity;
    pq_status wst;
    if (bt_wave(ctx, b, true, a, &wst)) return wst; // one symbol per wavefront (len <= 8192)
    if (position && cash && equity) { // all three state columns: the tiled SEQ op (coalesced column traffic)
        BtMacdOp op{};
        op.prm = *params; op.fast = fast; op.slow = slow; op.sig = sig; op.summary = summary;
        InCols<1> in{{close}};
        OutCols<3> out{{position, cash, equity}};
        return launch_seq(ctx, b, op, in, out);
    }
    if (equity) a.equity = equity;
    else {
        PQ_TRY(pq_ws_reserve(ctx, sizeof(double) * batch_rows(b) * 8));
        a.equity = pq_ws_col(ctx, b, 0);
        if (!a.equity) { pq_set_error("out of device memory for a scratch column"); return PQ_ERR_NOMEM; }
    }
    return bt_launch<true, false>(ctx, b, a);
}

pq_status pq_macd_cross_signals(pq_ctx *ctx, const pq_batch *b, const double *close, int64_t fast, int64_t slow,
                                int64_t sig, uint8_t *buy, uint8_t *sell) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(close && buy && sell, "pq_macd_cross_signals: null pointer");
    BtArgs a{};
    a.price = close; a.buy_out = buy; a.sell_out = sell; a.fast = fast; a.slow = slow; a.sig = sig;
    return bt_launch<true, true>(ctx, b, a);
}

pq_status pq_cross_signals(pq_ctx *ctx, const pq_batch *b, const double *a, const double *c, uint8_t *buy, uint8_t *sell) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(a && c && buy && sell, "pq_cross_signals: null pointer");
    return launch_row(ctx, b, CrossSigOp{}, InCols<2>{{a, c}}, OutColsT<CrossSigOp, uint8_t>{{buy, sell}});
}
pq_status pq_band_signals(pq_ctx *ctx, const pq_batch *b, const double *x, double lower, double upper, uint8_t *buy, uint8_t *sell) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(x && buy && sell, "pq_band_signals: null pointer");
    BandSigOp op{}; op.lower = lower; op.upper = upper;
    return launch_row(ctx, b, op, InCols<1>{{x}}, OutColsT<BandSigOp, uint8_t>{{buy, sell}});
}
pq_status pq_channel_signals(pq_ctx *ctx, const pq_batch *b, const double *price, const double *lo, const double *hi, int32_t mode,
                             uint8_t *buy, uint8_t *sell) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(price && lo && hi && buy && sell, "pq_channel_signals: null pointer");
    PQ_REQUIRE(mode == 0 || mode == 1, "pq_channel_signals: mode must be 0 (reversion) or 1 (breakout)");
    if (mode == 0) return launch_row(ctx, b, ChannelSigOp<0>{}, InCols<3>{{price, lo, hi}}, OutColsT<ChannelSigOp<0>, uint8_t>{{buy, sell}});
    return launch_row(ctx, b, ChannelSigOp<1>{}, InCols<3>{{price, lo, hi}}, OutColsT<ChannelSigOp<1>, uint8_t>{{buy, sell}});
}

pq_status pq_backtest_leveraged(pq_ctx *ctx, const pq_batch *b, const double *price, const uint8_t *buy, const uint8_t *sell,
                                const double *benchmark, const pq_lev_params *params, double *cash_net,
                                double *stock_value, double *total_value, int32_t max_trades, int32_t *trade_count,
                                int32_t *entry_day, int32_t *exit_day, double *entry_price, double *exit_price,
                                double *quantity, double *pnl, double *pnl_pct, int32_t *reason, double *summary) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(price && buy && sell && params && cash_net && stock_value && total_value, "pq_backtest_leveraged: null pointer");
    const int nrec = (entry_day != nullptr) + (exit_day != nullptr) + (entry_price != nullptr) + (exit_price != nullptr) +
                     (quantity != nullptr) + (pnl != nullptr) + (pnl_pct != nullptr) + (reason != nullptr);
    PQ_REQUIRE(nrec == 0 || nrec == 8, "pq_backtest_leveraged: pass all eight trade-record arrays or none");
    PQ_REQUIRE(max_trades >= 0, "pq_backtest_leveraged: max_trades < 0");
    if (b->n_series == 0) return PQ_OK;
    LevArgs a{};
    a.price = price; a.buy = buy; a.sell = sell; a.bench = benchmark;
    a.cash_net = cash_net; a.stock_value = stock_value; a.total_value = total_value;
    a.max_trades = max_trades; a.trade_count = trade_count; a.entry_day = entry_day; a.exit_day = exit_day; a.reason = reason;
    a.entry_price = entry_price; a.exit_price = exit_price; a.quantity = quantity; a.pnl = pnl; a.pnl_pct = pnl_pct;
    a.summary = summary; a.prm = *params;
    PQ_REQUIRE(!(b->offsets && benchmark), "pq_backtest_leveraged: a shared benchmark series has no meaning for a ragged batch (pass NULL)");
    pq_status wst;
    if (lev_wave(ctx, b, a, &wst)) return wst; // one symbol per wavefront (len <= 8192)
    if (!b->offsets && b->stride % 8 == 0 && reinterpret_cast<uintptr_t>(buy) % 8 == 0 && reinterpret_cast<uintptr_t>(sell) % 8 == 0) {
        LevOp op{};                        // tiled path: coalesced column traffic
        op.a = a; op.stride = b->stride;
        return launch_seq(ctx, b, op, InCols<1>{{price}}, OutCols<3>{{cash_net, stock_value, total_value}});
    }
    if (ctx->rec) { pq_set_error("pq_backtest_leveraged can only be recorded into a suite with stride % 8 == 0 and 8-byte aligned signals"); return PQ_ERR_UNSUPPORTED; }
    dim3 grid((unsigned)((b->n_series + SEQ_BLOCK - 1) / SEQ_BLOCK));
    hipLaunchKernelGGL(lev_backtest_kernel, grid, dim3(SEQ_BLOCK), 0, ctx->stream, a, dims_of(b));
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}

pq_status pq_portfolio_metrics(pq_ctx *ctx, const pq_batch *b, const double *total_value, double initial_total,
                               const double *benchmark, double *out) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(total_value && out, "pq_portfolio_metrics: null pointer");
    PQ_NO_RAGGED(b, "pq_portfolio_metrics");
    if (ctx->rec) { pq_set_error("pq_portfolio_metrics cannot be recorded into a suite"); return PQ_ERR_UNSUPPORTED; }
    if (b->len == 0) return PQ_OK;
    const int64_t nblk = (b->n_series + PORTFOLIO_BLOCK - 1) / PORTFOLIO_BLOCK > 0 ? (b->n_series + PORTFOLIO_BLOCK - 1) / PORTFOLIO_BLOCK : 1;
    PQ_TRY(pq_ws_reserve(ctx, (size_t)nblk * (size_t)b->len * 8));
    hipLaunchKernelGGL(portfolio_partial_kernel, dim3((unsigned)((b->len + 63) / 64), (unsigned)nblk), dim3(64), 0, ctx->stream, total_value,
                       dims_of(b), (double *)ctx->ws);
    hipLaunchKernelGGL(portfolio_combine_kernel, dim3((unsigned)((b->len + 63) / 64)), dim3(64), 0, ctx->stream, (const double *)ctx->ws, nblk,
                       b->len, out);
    hipLaunchKernelGGL(portfolio_metrics_kernel, dim3(1), dim3(256), 0, ctx->stream, b->len, initial_total, benchmark, out);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}

} // extern "C"
