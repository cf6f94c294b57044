/*
 * pq_hip.h -- C ABI of libpolars_quant_hip.so: the MI355X (gfx950) execution path for the
 * polars-quant technical-indicator + per-symbol backtest hot path.
 *
 * This header is the drop-in boundary.  Each entry point replaces one `#[polars_expr]` plugin
 * function / PyO3 method of the reference (file:line cited per function, paths relative to the
 * reference repo) with a BATCHED call over n_series independent series (symbols): one call here
 * replaces the N per-group plugin calls Polars makes under `.over("symbol")`.
 *
 * Conventions
 *   - all data pointers are DEVICE pointers (hipMalloc'd or torch CUDA tensors); plain C types only
 *   - a column is symbol-major: element (s, t) lives at ptr[s * stride + t], 0 <= t < len
 *   - f64 NULL rows are the NaN bit pattern PQ_NULL_BITS (in and out); int32 outputs that can be
 *     null (ht_trendmode) use PQ_NULL_I32.  pq_nulls_from_arrow / pq_validity_to_arrow convert
 *     from / to Arrow validity bitmaps.
 *   - every call is asynchronous on the context's HIP stream; pq_ctx_sync waits for it
 *   - return value: PQ_OK or an error code; pq_last_error() gives a thread-local message
 *   - parameters whose reference handling would panic (timeperiod == 0 underflow, momentum.rs:52)
 *     yield all-null output instead of aborting
 */
#ifndef PQ_HIP_H
#define PQ_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQ_NULL_BITS 0x7FF80000504E554CULL
#define PQ_NULL_I32 ((int32_t)0x80000000)

typedef int32_t pq_status;
enum {
    PQ_OK = 0,
    PQ_ERR_ARG = 1,     /* bad argument (null pointer, negative size, stride < len) */
    PQ_ERR_HIP = 2,     /* a HIP runtime call failed */
    PQ_ERR_NULLS = 3,   /* input has nulls and the reference function rejects them (N-B family) */
    PQ_ERR_NOMEM = 4,
    PQ_ERR_UNSUPPORTED = 5,
    PQ_WARN_SLOW_LAYOUT = 100 /* pq_layout_check only: not an error -- results are identical, the kernels take their 8-byte / per-lane forms */
};

typedef struct pq_ctx pq_ctx; /* device + stream + scratch workspace; one per host thread/stream.  A pq_ctx is NOT re-entrant: calls grow its
                               * workspaces on demand, so two host threads must not use one context concurrently (create one each: contexts
                               * are cheap; the Polars plugin symbols keep one per calling thread) */

/* n_series series of `len` rows; consecutive series start `stride` elements apart (stride >= len).
 * RAGGED batches (offsets != NULL): what the reference sees under `.over("symbol")` -- one plugin call per group of whatever
 * length the group has (python/polars_quant/talib/momentum.py:13-16, is_elementwise=False).  The columns are the LONG columns
 * sorted by symbol; `offsets` is a DEVICE array of n_series + 1 row indices, series s = rows [offsets[s], offsets[s + 1]);
 * `len` = the longest series, `stride` = the total row count offsets[n_series].  Every series is computed as if it were
 * passed alone (a series shorter than a warm-up is all-null, as in the reference); output columns have the same long layout.
 * Accepted by every indicator / pattern entry point, pq_backtest_vectorized, pq_backtest_macd_cross, pq_macd_cross_signals, the
 * signal rules and pq_returns; the cross-sectional calls (pq_factor_ic, pq_portfolio_metrics, pq_backtest_leveraged with a
 * benchmark) and suite recording return PQ_ERR_UNSUPPORTED. */
typedef struct {
    int64_t n_series;
    int64_t len;
    int64_t stride;
    const int64_t *offsets; /* NULL: the regular [n_series][stride] layout */
} pq_batch;

/* ---- layout advice (no reference counterpart: Polars hands over dense Arrow buffers) ----
 * Every kernel moves [64 series][8 or 16 rows] tiles with 16-byte accesses when the rows of a column are 16-byte aligned
 * (stride even AND every column base a multiple of 16 B), and is fastest when the row pitch is a multiple of 128 B: every 64 / 128-byte
 * piece of a tile is then one aligned cache line.  Measured on the full suite at 5 000 x 2 520: pitch 2 528
 * 3.9 ms per step, dense even pitch 2 520 4.1 ms, ODD pitch 2 521 5.8 ms (the 8-byte forms: every piece straddles two cache lines).
 *   pq_recommended_stride(len)  the smallest multiple of 16 elements (128 B) >= len: the pitch to allocate [n_series][stride] columns
 *                               with (pq_memcpy_h2d_pitched places a dense host column there)
 *   pq_layout_check(b, cols, n) PQ_OK if the n column bases and the batch's stride take the fast forms, PQ_WARN_SLOW_LAYOUT if the
 *                               call will run the 8-byte forms (odd stride, or a base 8 bytes off) or the per-lane gather forms (a
 *                               base not even 8-byte aligned); pq_last_error() then says which.  Ragged batches (offsets != NULL)
 *                               always return PQ_OK: their groups start at arbitrary rows by construction.  Advisory only: every
 *                               entry point accepts every layout and returns the same values.
 * The library's own allocations follow the advice: the Polars plugin's device copies of a balanced `_over` panel, loader.DeviceFrame
 * and polars_quant_amd.Suite (which also re-houses inputs handed over at a slow pitch once, at record time). */
int64_t pq_recommended_stride(int64_t len);
pq_status pq_layout_check(const pq_batch *b, const void *const *cols, int32_t n_cols);

/* ---- runtime ---- */
int32_t pq_abi_version(void);
const char *pq_last_error(void);
pq_status pq_device_count(int32_t *count);
/* hip_stream: the hipStream_t every call launches on (e.g. torch's current stream); NULL = the default stream */
pq_status pq_ctx_create(int32_t device, void *hip_stream, pq_ctx **out);
pq_status pq_ctx_destroy(pq_ctx *ctx);
pq_status pq_ctx_set_stream(pq_ctx *ctx, void *hip_stream);
pq_status pq_ctx_sync(pq_ctx *ctx);
pq_status pq_malloc(pq_ctx *ctx, size_t bytes, void **dptr);
pq_status pq_free(pq_ctx *ctx, void *dptr);
/* pin / unpin a host buffer (an Arrow data buffer) so that the two copies below DMA straight from / to it */
pq_status pq_host_register(void *host_ptr, size_t bytes);
pq_status pq_host_unregister(void *host_ptr);
pq_status pq_memcpy_h2d(pq_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
pq_status pq_memcpy_d2h(pq_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* The same for `rows` rows of `width_bytes` each with different row pitches on the two sides: a dense host column
 * ([n_series][len], what Arrow hands over) to / from a device column whose pitch is a multiple of 128 B (pq_batch.stride > len:
 * every 64 / 128-byte tile piece is then one aligned cache line, which the suite replay rewards with ~8 %). */
pq_status pq_memcpy_h2d_pitched(pq_ctx *ctx, void *dst_dev, size_t dst_pitch_bytes, const void *src_host, size_t src_pitch_bytes,
                                size_t width_bytes, size_t rows);
pq_status pq_memcpy_d2h_pitched(pq_ctx *ctx, void *dst_host, size_t dst_pitch_bytes, const void *src_dev, size_t src_pitch_bytes,
                                size_t width_bytes, size_t rows);
/* Arrow validity bitmap (LSB-first, bit i = row i of the long column, starting at bit `bit_offset`)
 * -> overwrite null rows of `col` (n contiguous rows) with PQ_NULL_BITS */
pq_status pq_nulls_from_arrow(pq_ctx *ctx, double *col, const uint8_t *validity_bits, int64_t bit_offset, int64_t n);
/* PQ_NULL_BITS rows of `col` -> Arrow validity bitmap (ceil(n/8) bytes) + null count (device int64) */
pq_status pq_validity_to_arrow(pq_ctx *ctx, const double *col, int64_t n, uint8_t *validity_bits, int64_t *null_count);
pq_status pq_count_nulls(pq_ctx *ctx, const pq_batch *b, const double *col, int64_t *host_count);

/* ---- overlap studies (src/talib/overlap.rs) ---- */
pq_status pq_sma(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);              /* :494 */
pq_status pq_ema(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);              /* :128 */
pq_status pq_bbands(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double nbdevup,
                    double nbdevdn, double *bb_upper, double *bb_middle, double *bb_lower);                     /* :47 */
pq_status pq_dema(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);             /* :119 */
pq_status pq_tema(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);             /* :513 */
pq_status pq_t3(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double vfactor, double *out); /* :503 */
pq_status pq_trima(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);            /* :522 */
pq_status pq_wma(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);              /* :531 */
pq_status pq_kama(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);             /* :137 */
pq_status pq_ma(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, int64_t matype, double *out); /* :146 */
pq_status pq_midpoint(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);         /* :180 */
pq_status pq_midprice(pq_ctx *, const pq_batch *, const double *high, const double *low, int64_t timeperiod,
                      double *out);                                                                             /* :281 */
pq_status pq_mama(pq_ctx *, const pq_batch *, const double *real, double fastlimit, double slowlimit,
                  double *mama, double *fama);                                                                  /* :156 */
pq_status pq_mavp(pq_ctx *, const pq_batch *, const double *real, const double *periods, int64_t minperiod,
                  int64_t maxperiod, int64_t matype, double *out);                                              /* :407 */
pq_status pq_sar(pq_ctx *, const pq_batch *, const double *high, const double *low, double acceleration,
                 double maximum, double *out);                                                                  /* :437 */
pq_status pq_sarext(pq_ctx *, const pq_batch *, const double *high, const double *low, double startvalue,
                    double offsetonreverse, double accelerationinitlong, double accelerationlong,
                    double accelerationmaxlong, double accelerationinitshort, double accelerationshort,
                    double accelerationmaxshort, double *out);                                                  /* :457 */

/* ---- momentum (src/talib/momentum.rs; composites python/polars_quant/talib/momentum.py) ---- */
pq_status pq_adx(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                 int64_t timeperiod, double *out);                                                              /* :11 */
pq_status pq_adxr(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                  int64_t timeperiod, double *out);                                                             /* :32 */
pq_status pq_dx(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                int64_t timeperiod, double *out);                                                               /* :226 */
pq_status pq_plus_di(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                     int64_t timeperiod, double *out);                                                          /* :400 */
pq_status pq_minus_di(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                      int64_t timeperiod, double *out);                                                         /* :345 */
pq_status pq_plus_dm(pq_ctx *, const pq_batch *, const double *high, const double *low, int64_t timeperiod,
                     double *out);                                                                              /* :414 */
pq_status pq_minus_dm(pq_ctx *, const pq_batch *, const double *high, const double *low, int64_t timeperiod,
                      double *out);                                                                             /* :359 */
pq_status pq_aroon(pq_ctx *, const pq_batch *, const double *high, const double *low, int64_t timeperiod,
                   double *aroon_up, double *aroon_down);                                                       /* :70 */
pq_status pq_aroonosc(pq_ctx *, const pq_batch *, const double *high, const double *low, int64_t timeperiod,
                      double *out);                                                             /* momentum.py:40 */
/* Multi-output forms: several reference functions over the same inputs evaluated as ONE job (one in-tile, one walk); every
 * output column is bit-identical to the single function's.  A DataFrame query that asks for several of them
 * (df.with_columns([EMA, DEMA, TEMA, TRIX, ...])) maps onto these like common-subexpression elimination. */
pq_status pq_ema_all(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *ema, double *dema, double *tema,
                     double *trix);
pq_status pq_atr_all(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close, int64_t timeperiod,
                     double *atr, double *natr);
pq_status pq_dm_pair(pq_ctx *, const pq_batch *, const double *high, const double *low, int64_t timeperiod, double *plus_dm,
                     double *minus_dm);
pq_status pq_ad_all(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close, const double *volume,
                    int64_t fastperiod, int64_t slowperiod, double *ad, double *adosc);
pq_status pq_macd_pair(pq_ctx *, const pq_batch *, const double *real, int64_t fastperiod, int64_t slowperiod, int64_t signalperiod,
                       int64_t macdfix_signalperiod, double *macd, double *macdsignal, double *macdhist, double *fix_macd,
                       double *fix_signal, double *fix_hist);
/* Wilder's directional system over (high, low, close): DX, +DI, -DI, ADX, ADXR and ATR, NATR of one timeperiod as one job */
pq_status pq_dm_system_all(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close, int64_t timeperiod,
                           double *dx, double *plus_di, double *minus_di, double *adx, double *adxr, double *atr, double *natr);
/* SMA(timeperiod) and MA(timeperiod, matype 0) are the same walk (overlap.rs:857-869: `ma` dispatches to calc_sma): one job, both columns */
pq_status pq_sma_ma(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *sma, double *ma);
/* the two up/down-move oscillators of one timeperiod as one job */
pq_status pq_cmo_rsi(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *cmo, double *rsi);
/* the volume family over (high, low, close, volume): MFI + AD + ADOSC + OBV as one job (4 in / 4 out: MFI's own LDS need) */
pq_status pq_volume_all(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close, const double *volume,
                        int64_t mfi_timeperiod, int64_t adosc_fastperiod, int64_t adosc_slowperiod, double *mfi, double *ad,
                        double *adosc, double *obv);
pq_status pq_sar_pair(pq_ctx *, const pq_batch *, const double *high, const double *low, double acceleration, double maximum,
                      double startvalue, double offsetonreverse, double accelerationinitlong, double accelerationlong,
                      double accelerationmaxlong, double accelerationinitshort, double accelerationshort,
                      double accelerationmaxshort, double *sar, double *sarext);
/* STOCH + STOCHF of one fastk_period: the rolling extrema are evaluated once */
pq_status pq_stoch_all(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close, int64_t fastk_period,
                       int64_t slowk_period, int64_t slowk_matype, int64_t slowd_period, int64_t slowd_matype, int64_t fastd_period,
                       int64_t fastd_matype, double *slowk, double *slowd, double *fastk, double *fastd);
pq_status pq_apo_ppo(pq_ctx *, const pq_batch *, const double *real, int64_t fastperiod, int64_t slowperiod, int64_t matype,
                     double *apo, double *ppo);
/* AROON and AROONOSC of the same timeperiod from one window scan (multi-output form, like pq_dmi_all) */
pq_status pq_aroon_all(pq_ctx *, const pq_batch *, const double *high, const double *low, int64_t timeperiod,
                       double *aroon_up, double *aroon_down, double *aroonosc);
pq_status pq_bop(pq_ctx *, const pq_batch *, const double *open, const double *high, const double *low,
                 const double *close, double *out);                                                             /* :113 */
pq_status pq_cci(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                 int64_t timeperiod, double *out);                                                              /* :138 */
pq_status pq_cmo(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);              /* :181 */
pq_status pq_macd(pq_ctx *, const pq_batch *, const double *real, int64_t fastperiod, int64_t slowperiod,
                  int64_t signalperiod, double *macd, double *macd_signal, double *macd_hist);                  /* :250 */
pq_status pq_macdext(pq_ctx *, const pq_batch *, const double *real, int64_t fastperiod, int64_t fastmatype,
                     int64_t slowperiod, int64_t slowmatype, int64_t signalperiod, int64_t signalmatype,
                     double *macd_dif, double *macd_dea, double *macd_hist);                    /* momentum.py:83 */
pq_status pq_macdfix(pq_ctx *, const pq_batch *, const double *real, int64_t signalperiod, double *macd,
                     double *macd_signal, double *macd_hist);                                   /* momentum.py:90 */
pq_status pq_mfi(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                 const double *volume, int64_t timeperiod, double *out);                                        /* :286 */
pq_status pq_mom(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);              /* :384 */
pq_status pq_roc(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);              /* :439 */
pq_status pq_rocp(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);             /* :456 */
pq_status pq_rocr(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);             /* :473 */
pq_status pq_rocr100(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);          /* :490 */
pq_status pq_rsi(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);              /* :507 */
pq_status pq_trix(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, double *out);             /* :544 */
pq_status pq_ultosc(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                    int64_t timeperiod1, int64_t timeperiod2, int64_t timeperiod3, double *out);                /* :572 */
pq_status pq_willr(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                   int64_t timeperiod, double *out);                                                            /* :630 */
pq_status pq_apo(pq_ctx *, const pq_batch *, const double *real, int64_t fastperiod, int64_t slowperiod,
                 int64_t matype, double *out);                                                  /* momentum.py:25 */
pq_status pq_ppo(pq_ctx *, const pq_batch *, const double *real, int64_t fastperiod, int64_t slowperiod,
                 int64_t matype, double *out);                                                 /* momentum.py:136 */
pq_status pq_stoch(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                   int64_t fastk_period, int64_t slowk_period, int64_t slowk_matype, int64_t slowd_period,
                   int64_t slowd_matype, double *slowk, double *slowd);                        /* momentum.py:178 */
pq_status pq_stochf(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                    int64_t fastk_period, int64_t fastd_period, int64_t fastd_matype, double *fastk,
                    double *fastd);                                                            /* momentum.py:188 */
pq_status pq_stochrsi(pq_ctx *, const pq_batch *, const double *real, int64_t timeperiod, int64_t fastk_period,
                      int64_t fastd_period, int64_t fastd_matype, double *fastk, double *fastd); /* momentum.py:197 */

/* fused multi-output forms: the shared core is evaluated once (bit-identical to the single-output calls) */
pq_status pq_dmi_all(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                     int64_t timeperiod, double *dx, double *plus_di, double *minus_di, double *adx,
                     double *adxr);                                                    /* momentum.rs:668-727 calc_dm */
pq_status pq_ht_all(pq_ctx *, const pq_batch *, const double *real, double *ht_dcperiod, double *ht_dcphase,
                    double *inphase, double *quadrature, double *sine, double *leadsine);    /* cycle.rs:27-63 */

/* ---- volatility / volume / price (src/talib/{volatility,volume,price}.rs) ---- */
pq_status pq_trange(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                    double *out);                                                              /* volatility.rs:51 */
pq_status pq_atr(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                 int64_t timeperiod, double *out);                                             /* volatility.rs:18 */
pq_status pq_natr(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                  int64_t timeperiod, double *out);                                            /* volatility.rs:34 */
pq_status pq_ad(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                const double *volume, double *out);                                                /* volume.rs:19 */
pq_status pq_adosc(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                   const double *volume, int64_t fastperiod, int64_t slowperiod, double *out);     /* volume.rs:34 */
pq_status pq_obv(pq_ctx *, const pq_batch *, const double *close, const double *volume, double *out); /* volume.rs:70 */
pq_status pq_avgprice(pq_ctx *, const pq_batch *, const double *open, const double *high, const double *low,
                      const double *close, double *out);                                            /* price.rs:10 */
pq_status pq_medprice(pq_ctx *, const pq_batch *, const double *high, const double *low, double *out); /* price.rs:34 */
pq_status pq_typprice(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                      double *out);                                                                 /* price.rs:53 */
pq_status pq_wclprice(pq_ctx *, const pq_batch *, const double *high, const double *low, const double *close,
                      double *out);                                                                 /* price.rs:74 */

/* ---- returns (README.md:46-75 `returns(df, price_col, period, method, return_col)`; README-only, decision D-13) ----
 * method 0 "simple": (p[t] - p[t-period]) / p[t-period];  1 "log": ln(p[t] / p[t-period]).  Null for t < period and where
 * either price is null; period <= 0 or another method: all null.  Pinned by the reference-held vector README.md:75. */
pq_status pq_returns(pq_ctx *, const pq_batch *, const double *price, int64_t period, int64_t method, double *out);

/* rolling maximum / minimum over the last `window` rows, null until the frame holds `window` non-null rows (the Polars
 * rolling_max / rolling_min of momentum.py:181-183): channel bounds for the README's breakout strategy (README.md:947) */
pq_status pq_rolling_max(pq_ctx *, const pq_batch *, const double *x, int64_t window, double *out);
pq_status pq_rolling_min(pq_ctx *, const pq_batch *, const double *x, int64_t window, double *out);

/* ---- cycle (src/talib/cycle.rs) ---- */
pq_status pq_ht_dcperiod(pq_ctx *, const pq_batch *, const double *real, double *out);                          /* :10 */
pq_status pq_ht_dcphase(pq_ctx *, const pq_batch *, const double *real, double *out);                           /* :75 */
pq_status pq_ht_phasor(pq_ctx *, const pq_batch *, const double *real, double *inphase, double *quadrature);    /* :159 */
pq_status pq_ht_sine(pq_ctx *, const pq_batch *, const double *real, double *sine, double *leadsine);           /* :236 */
pq_status pq_ht_trendline(pq_ctx *, const pq_batch *, const double *real, double *out);                         /* :310 */
pq_status pq_ht_trendmode(pq_ctx *, const pq_batch *, const double *real, int32_t *out);                        /* :377 */

/* ---- candlestick patterns (src/talib/pattern.rs:10-2062) ---- */
#define PQ_N_PATTERNS 61
/* names in id order, lower-case plugin names ("cdl2crows", ...) */
const char *pq_pattern_name(int32_t id);
int32_t pq_pattern_id(const char *name);
/* one recogniser; penetration is used by ids 14,19,20,42,43,45 (pattern.rs:529,675,713,1426,1464,1529) */
pq_status pq_cdl(pq_ctx *, const pq_batch *, int32_t pattern_id, const double *open, const double *high,
                 const double *low, const double *close, double penetration, int32_t *out);
/* all 61 in one pass over OHLC: outs[id] may be NULL to skip a pattern; penetrations[id] per pattern */
pq_status pq_cdl_all(pq_ctx *, const pq_batch *, const double *open, const double *high, const double *low,
                     const double *close, const double *penetrations /* host, 61 */, int32_t *const *outs /* host, 61 device ptrs */);

/* ---- backtest (src/backtest/vectorized.rs:69-224, src/backtest/metrics.rs:7-152) ---- */
typedef struct {
    double initial_capital, buy_slippage, sell_slippage, buy_commission_rate, sell_commission_rate,
        min_commission, position_size;
} pq_bt_params; /* vectorized.rs:38 defaults: 1e5, 0, 0, 3e-4, 3e-4, 5, 1 */
#define PQ_SUMMARY_COLS 8 /* annualized_return, max_drawdown, alpha, beta, sharpe_ratio, max_profit, win_rate, total_trades */
/* buy/sell: uint8 0/1 per row (null -> 0, vectorized.rs:80-98); price null -> NaN (:70-78);
 * EVERY column (price, buy, sell, benchmark, position, cash, equity) is [n_series][stride]: element (s, t) at
 * ptr[s * stride + t] -- a benchmark shared by all series must be replicated per series by the caller;
 * benchmark may be NULL; position/cash/equity may be NULL (summary only); summary is [n_series][8] */
pq_status pq_backtest_vectorized(pq_ctx *, const pq_batch *, const double *price, const uint8_t *buy,
                                 const uint8_t *sell, const double *benchmark, const pq_bt_params *params,
                                 double *position, double *cash, double *equity, double *summary);
/* fused strategy + backtest: MACD(fast,slow,signal) cross signals generated on the fly
 * (buy: macd crosses above signal; sell: crosses below), then the same scan + summary */
pq_status pq_backtest_macd_cross(pq_ctx *, const pq_batch *, const double *close, int64_t fastperiod,
                                 int64_t slowperiod, int64_t signalperiod, const pq_bt_params *params,
                                 double *position, double *cash, double *equity, double *summary);
pq_status pq_macd_cross_signals(pq_ctx *, const pq_batch *, const double *close, int64_t fastperiod,
                                int64_t slowperiod, int64_t signalperiod, uint8_t *buy, uint8_t *sell);
/* Both backtests run ONE SYMBOL PER WAVEFRONT for len <= 8192 (csrc/ops_backtest_wave.h): the MACD state machine is run in
 * 64 speculative row chunks per symbol whose hand-over states are compared bit for bit (a chunk that fails is re-run from
 * its predecessor's state, so results are exact either way).  out3 (host): [0] symbols processed that way since the last
 * reset, [1] chunks that failed the bit test, [2] chunk re-runs.  Synchronises the context's stream. */
pq_status pq_backtest_wave_stats(pq_ctx *, int64_t *out3, int32_t reset);
/* The contractive recurrences of the indicator suite -- calc_ema (overlap.rs:660-730) and its cascades DEMA / TEMA / TRIX, MACD
 * (momentum.rs:250-283), Wilder's calc_rma behind RSI (momentum.rs:507-541), +DM / -DM, DX / DI / ADX / ADXR (momentum.rs:668-727),
 * ATR / NATR (volatility.rs:18-48) -- and the extrema of MIDPOINT have a ONE-SYMBOL-PER-WAVEFRONT form (csrc/wt_dev.h): 64 row chunks
 * per symbol, started from prefix-scanned seeds, whose hand-over states are compared bit for bit (a chunk that fails is re-run from
 * its predecessor's state: results are the serial walk's either way).  DIRECT calls (not recorded into a suite) take it on regular
 * batches with 1 024 <= len <= 4 096 where it beats the lane-per-symbol kernel of the function (EMA, TRIX, RSI, +DM / -DM, ATR / NATR,
 * MIDPOINT, pq_ema_all, pq_macd_pair, pq_dm_pair, pq_atr_all), and on RAGGED batches whose groups average >= 1 024 rows (every function
 * named above).  A symbol / group with a NULL or NaN input is left to the lane-per-symbol kernel of the same function, launched gated
 * behind.  out4 (host): [0] symbols computed that way since the last reset, [1] chunks that failed the bit test, [2] chunk re-runs,
 * [3] symbols handed to the gated general path.  Synchronises the context's stream. */
pq_status pq_wt_stats(pq_ctx *, int64_t *out4, int32_t reset);
/* Ragged batches whose groups are of similar length (n_series x pq_recommended_stride(longest) <= 1.5 x total rows, >= 16 groups,
 * >= 16 384 rows) do not run the per-lane gather forms of the sequential kernels: every function here is causal in time, so the groups
 * are re-housed as the rows of a regular padded batch (one streaming copy per input column), the tiled kernel of the same function
 * walks it -- told every group's own length, so short groups see the reference's short-series rules -- and the groups' rows of every
 * output are copied back (csrc/pq_dev.h launch_seq, csrc/runtime.hip rg_pack / rg_unpack).  Bit-identical to the gather forms, 2-10 x
 * faster on `.over("symbol")`-shaped batches (profiles/r05_bench_ragged.json).  *calls: launches that took this path on the context
 * since the last reset.  PQ_NO_RG_PACK=1 in the environment keeps the gather forms (A/B runs and tests). */
pq_status pq_ragged_rehouse_stats(pq_ctx *, int64_t *calls, int32_t reset);

/* ---- multi-GPU: symbols are split statically over the ranks -- rank r of G owns [floor(N r / G), floor(N (r + 1) / G)),
 * a contiguous byte range of every symbol-major column -- and every rank runs the calls above on its own shard with no
 * communication.  The one exchange is the per-symbol summary table (reference: the dict of vectorized.rs:204-223 per symbol):
 *   pq_comm_unique_id   rank 0 makes the 128-byte rendezvous id; the HOST ships it to the other ranks (its own channel)
 *   pq_comm_init        collective over the `world` ranks (one process per GPU); RCCL is bound at run time (dlopen)
 *   pq_gather_summaries local [n_local][8] (this rank's shard, device) -> all [n_symbols][8] (device, symbol order) on every
 *                       rank, on the context's stream: one ncclAllGather when the shards are equal, else one grouped broadcast
 *                       per rank */
#define PQ_COMM_ID_BYTES 128
pq_status pq_comm_unique_id(void *id128);
pq_status pq_comm_init(pq_ctx *, int32_t rank, int32_t world, const void *id128);
pq_status pq_comm_destroy(pq_ctx *);
pq_status pq_shard_range(int64_t n_symbols, int32_t rank, int32_t world, int64_t *lo, int64_t *hi);
pq_status pq_gather_summaries(pq_ctx *, const double *local, int64_t n_symbols, double *all);
/* The same exchange OFF the step's critical path (round 5): at 8 GPUs a 625-symbol backtest step is ~50 us and an all-gather tens of
 * us, so a gather in series behind every step would cost a third of the throughput.
 *   pq_gather_summaries_begin  marks everything enqueued on the context's stream so far (the step that produced `local`) with an
 *                              event; the communicator's OWN stream (created on first use, highest priority) waits for it and takes
 *                              the collective.  Returns at once: the context's stream is free for the next step.  slot = 0 / 1.
 *   pq_gather_summaries_end    the context's stream waits -- on the device, the host does not block -- for that slot's collective:
 *                              work enqueued afterwards may read `all` and overwrite `local`.  A no-op on an idle slot.
 * Double-buffer `local` / `all` by slot and call _end(slot) just before the step that rewrites local[slot]: one exchange is then in
 * flight beside the next step's kernels (polars_quant_amd/distributed.py: OverlappedGather; bench.py --gpus N times it).  No
 * reference counterpart (single process). */
pq_status pq_gather_summaries_begin(pq_ctx *, const double *local, int64_t n_symbols, double *all, int32_t slot);
pq_status pq_gather_summaries_end(pq_ctx *, int32_t slot);
/* host-side wait for the communicator's own stream (e.g. to bound the FIRST exchange from a helper thread: a collective that never
 * returns then blocks that thread, not the context's stream) */
pq_status pq_comm_sync(pq_ctx *);

/* ---- the README's `Strategy` signal rules (README.md:862-994; README-only, decision D-11 in
 * oracle/backtest.c): indicator columns -> uint8 buy / sell columns for the backtests above.  Row-parallel.
 *   cross:   buy = a[i-1] <= b[i-1] && a[i] > b[i];  sell mirrored            (MA / MACD / STOCH / trend strategies)
 *   band:    buy = x[i-1] < lower && x[i] >= lower;  sell = x[i-1] > upper && x[i] <= upper      (RSI / CCI / STOCH)
 *   channel: mode 0 reversion (BBANDS): buy = price crosses below `lo`, sell = crosses above `hi`;
 *            mode 1 breakout (Donchian): buy = price[i] > hi[i-1], sell = price[i] < lo[i-1]
 * a rule is false where any value it reads is null, and on row 0 */
pq_status pq_cross_signals(pq_ctx *, const pq_batch *, const double *a, const double *b, uint8_t *buy, uint8_t *sell);
pq_status pq_band_signals(pq_ctx *, const pq_batch *, const double *x, double lower, double upper, uint8_t *buy, uint8_t *sell);
pq_status pq_channel_signals(pq_ctx *, const pq_batch *, const double *price, const double *lo, const double *hi, int32_t mode,
                             uint8_t *buy, uint8_t *sell);

/* The remaining comparisons of the README's fourteen `Strategy` generators (README.md:942-956; definitions = decision D-11b),
 * row-parallel and recordable into a suite in front of pq_backtest_vectorized.  A NULL compares false everywhere.
 *   pq_gate_signals   (buy_in, sell_in) -> (buy, sell), in place allowed; mode 0 zones: buy &= a < k0, sell &= a > k1 (stoch);
 *                     1 strength: both &= a > k0 (adx); 2: buy &= a > c (ma trend filter); 3: buy &= a[i] > a[i-1] (slope filter);
 *                     4: buy &= (a - c) > |c| * k0 (distance filter)
 *   pq_zscore         z = (price - mid) / (upper - mid), NULL where upper is (reversion: then pq_band_signals on z)
 *   pq_scale_band     lo = base * f_lo, hi = base * f_hi, NULL where base is (grid: then pq_channel_signals mode 0)
 *   pq_volume_surge_signals  surge = volume > multiplier * avg_volume; buy = surge on an up close, sell = surge on a down close
 *   pq_gap_signals    buy = open[i] > high[i-1] * f_up, sell = open[i] < low[i-1] * f_dn
 *   pq_pattern_any_signals   buy = any of <= 16 recogniser columns == +100, sell = any of <= 16 == -100 (host arrays of device pointers)
 *   pq_ma_stack_signals      buy = first bar where mas[0] > mas[1] > ... holds, sell = first bar of the reverse order (2..16 columns) */
pq_status pq_gate_signals(pq_ctx *, const pq_batch *, const double *a, const double *c, int32_t mode, double k0, double k1,
                          const uint8_t *buy_in, const uint8_t *sell_in, uint8_t *buy, uint8_t *sell);
pq_status pq_zscore(pq_ctx *, const pq_batch *, const double *price, const double *upper, const double *mid, double *z);
pq_status pq_scale_band(pq_ctx *, const pq_batch *, const double *base, double f_lo, double f_hi, double *lo, double *hi);
pq_status pq_volume_surge_signals(pq_ctx *, const pq_batch *, const double *volume, const double *avg_volume, const double *close,
                                  double multiplier, uint8_t *buy, uint8_t *sell);
pq_status pq_gap_signals(pq_ctx *, const pq_batch *, const double *open, const double *high, const double *low, double f_up, double f_dn,
                         uint8_t *buy, uint8_t *sell);
pq_status pq_pattern_any_signals(pq_ctx *, const pq_batch *, const int32_t *const *bullish, int32_t n_bullish,
                                 const int32_t *const *bearish, int32_t n_bearish, uint8_t *buy, uint8_t *sell);
pq_status pq_ma_stack_signals(pq_ctx *, const pq_batch *, const double *const *mas, int32_t n, uint8_t *buy, uint8_t *sell);

/* ---- the README's multi-symbol `Backtest` (README.md:346-640; README-only, no source).
 * Semantics = decision D-10 (oracle/backtest.c): independent capital pool per symbol, 100-share lots, leverage
 * with daily compounding interest on the debt, margin call (forced sale), commission with a minimum, proportional
 * slippage.  One symbol per lane; every column is [n_series][stride].
 *   cash_net = cash - debt, stock_value, total_value: daily records (get_daily_records), all required.
 *   trade records (get_position_records): the first max_trades closed trades per symbol in [n_series][max_trades] arrays,
 *   all eight arrays or none (NULL); trade_count [n_series] counts every closed trade.  reason: 1 signal, 2 margin call.
 *   benchmark: ONE series of b->len rows shared by all symbols, or NULL; summary [n_series][8] as pq_backtest_vectorized. */
typedef struct {
    double initial_capital, position_size, leverage, margin_call_threshold, interest_rate, commission_rate,
        min_commission, slippage;
} pq_lev_params; /* README.md:356-365 defaults: 1e5, 1, 1, 0.3, 0.06, 3e-4, 5, 0 */
pq_status pq_backtest_leveraged(pq_ctx *, const pq_batch *, const double *price, const uint8_t *buy, const uint8_t *sell,
                                const double *benchmark, const pq_lev_params *params, double *cash_net,
                                double *stock_value, double *total_value, int32_t max_trades, int32_t *trade_count,
                                int32_t *entry_day, int32_t *exit_day, double *entry_price, double *exit_price,
                                double *quantity, double *pnl, double *pnl_pct, int32_t *reason, double *summary);
#define PQ_PORTFOLIO_COLS 10 /* portfolio_value, daily_pnl, daily_return_pct, cumulative_pnl, cumulative_return_pct,
                                benchmark_return_pct, alpha_pct, relative_return_pct, beta, (reserved 0) */
/* get_performance_metrics (README.md:455-477): per-day sums over the symbols of this batch (blocks of 256 symbols, ascending
 * inside a block, block sums added in ascending order), then the
 * day-to-day metrics; out is [b->len][PQ_PORTFOLIO_COLS]; benchmark: one series of b->len rows or NULL */
pq_status pq_portfolio_metrics(pq_ctx *, const pq_batch *, const double *total_value, double initial_total,
                               const double *benchmark, double *out);

/* ---- cross-sectional factor evaluation, Factor.ic / rank_ic / rolling_ic (README.md:1429-1430,
 * :1480-1482, :1626-1634; README-only, decision D-12 in oracle/backtest.c).  factor / fwd_return: [n_series][stride];
 * per day the cross-section = symbols where both values are non-null and finite.  method 0: Pearson IC (sums over
 * blocks of 256 symbols in ascending order), 1: Spearman Rank-IC (average ranks; n_series <= 100000, n_series*len < 2^32; uses the
 * context workspace, ~52 bytes per cell).  ic: [len] (null where fewer than 2 pairs or zero variance); n_valid: [len] or NULL */
pq_status pq_factor_ic(pq_ctx *, const pq_batch *, const double *factor, const double *fwd_return, int32_t method, double *ic,
                       int32_t *n_valid);
/* rolling mean of ic over `window` rows (null unless all of them are non-null) and mean / sample std (IR) */
pq_status pq_rolling_ic(pq_ctx *, const double *ic, int64_t len, int64_t window, double *rolling_ic, double *rolling_ir);

/* ---- suites: record many calls, replay them as a few chip-filling grids ----
 * One indicator over N symbols is only N/64 wavefronts -- far too few for 256 CUs -- but a DataFrame query asks
 * for many indicators at once (df.with_columns([...]) in the reference; Polars then calls the plugin once per
 * expression and group).  Between pq_suite_begin and pq_suite_end every pq_* compute call on `ctx` is RECORDED
 * instead of launched (same batch shape for all; pointers must stay valid for the suite's lifetime);
 * pq_suite_run replays the whole set: per dependency phase ONE grid runs all sequential jobs
 * (blockIdx.y = job) and the row-parallel launches follow.  Results are bit-identical to the direct calls. */
typedef struct pq_suite pq_suite;
pq_status pq_suite_begin(pq_ctx *ctx, const pq_batch *b);
pq_status pq_suite_end(pq_ctx *ctx, pq_suite **out);
pq_status pq_suite_abort(pq_ctx *ctx);
pq_status pq_suite_run(pq_ctx *ctx, pq_suite *suite);
pq_status pq_suite_destroy(pq_ctx *ctx, pq_suite *suite);
pq_status pq_suite_info(const pq_suite *suite, int32_t *n_phases, int32_t *n_seq_jobs, int32_t *n_row_launches);
/* measurement: HIP events around every sequential-job grid, on the stream it is launched on */
pq_status pq_suite_set_timing(pq_suite *suite, int32_t on);
pq_status pq_suite_grid_stats(pq_suite *suite, int32_t grid, double *avg_ms, double *algorithmic_bytes, int32_t *n_jobs,
                              int32_t *lds_bytes, int32_t *runs);
/* which kernel a grid launches: 0 = seq_jobs_kernel<0> (tiled bodies), 1 = seq_jobs_kernel<1> (register-heavy ops),
 * 2 = seq_jobs_kernel<2> (gather bodies) */
pq_status pq_suite_grid_variant(pq_suite *suite, int32_t grid, int32_t *variant);
/* the launches of one kernel overlap inside a step: mean (latest end - earliest start) over the grids launching kernel
 * `variant`, and the sum of their algorithmic bytes */
pq_status pq_suite_span_stats(pq_suite *suite, int32_t variant, double *avg_span_ms, double *algorithmic_bytes);

#ifdef __cplusplus
}
#endif
#endif
