/*
 * pq_polars_plugin.h -- the Polars expression-plugin symbols of libpolars_quant_hip.so: what
 * `polars.plugins.register_plugin_function(plugin_path=<.so>, function_name="ema")` resolves with dlsym
 * (python/polars_quant/talib/overlap.py:36-43) and what `#[polars_expr]` generates in the reference -- one
 * `_polars_plugin_<f>` / `_polars_plugin_field_<f>` pair for EVERY plugin function the reference's Python registers: the 115 Rust
 * names + aroonosc, apo, ppo (registered by momentum.py:27,42,138, absent from the Rust library; decision D-6) = 118 pairs, and
 * for each of them a BATCHED twin `_polars_plugin_<f>_over` (below).
 *
 * ABI: the C structs of pyo3-polars 0.26 / polars-ffi (version 0.1) as far as their published layout is known --
 * UNVERIFIED against the pinned polars 0.53 (Cargo.lock:950-951): no `polars` wheel exists in this image, so the entry
 * points are exercised by tests/ with Arrow C Data Interface structs built by pyarrow, not by Polars itself.
 *
 * Parameters: both conventions of the reference are accepted: pickled kwargs ({"timeperiod": 20}, the Rust side,
 * overlap.rs:18-22) and a trailing literal input Series of one i64 row (the Python wrapper's `args=[real, timeperiod]`).
 * `_polars_plugin_<f>`: each call is ONE series (Polars calls once per expression and once per group under .over("symbol")):
 * n_series = 1 -- an H2D copy, a launch and a D2H copy per ~20 KB group.
 * `_polars_plugin_<f>_over`: the batched form, the only plugin shape worth shipping on a GPU.  inputs = the function's columns (whole
 * columns of a frame in which equal keys are contiguous, i.e. sorted by symbol), then the KEY column (any integer / float /
 * utf8 / large_utf8 / string-view / dictionary column; null keys form one group), then the literals.  The group offsets are
 * derived from the key and every group runs in ONE ragged launch (pq_batch.offsets, pq_hip.h); the result is the concatenation
 * of the per-group results -- what `.over(key)` assembles from its per-group calls.  Python side:
 *     register_plugin_function(args=[real, pl.col("symbol"), timeperiod], plugin_path=_LIB, function_name="ema_over", is_elementwise=False)
 */
#ifndef PQ_POLARS_PLUGIN_H
#define PQ_POLARS_PLUGIN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Arrow C Data Interface (https://arrow.apache.org/docs/format/CDataInterface.html) */
struct ArrowSchema {
    const char *format, *name, *metadata;
    int64_t flags, n_children;
    struct ArrowSchema **children, *dictionary;
    void (*release)(struct ArrowSchema *);
    void *private_data;
};
struct ArrowArray {
    int64_t length, null_count, offset, n_buffers, n_children;
    const void **buffers;
    struct ArrowArray **children, *dictionary;
    void (*release)(struct ArrowArray *);
    void *private_data;
};
/* polars-ffi version_0::SeriesExport: one Series = a field + `len` chunks */
typedef struct pq_series_export {
    struct ArrowSchema *field;
    struct ArrowArray **arrays;
    size_t len;
    void (*release)(struct pq_series_export *);
    void *private_data;
} pq_series_export;

uint32_t _polars_plugin_get_version(void);                       /* (major << 16) | minor = 0.1 */
const char *_polars_plugin_get_last_error_message(void);        /* thread-local, set when a call leaves return_value empty */
/* One pair of symbols (+ the _over pair) per reference function of the shape (1..4 numeric columns[, timeperiod]) -> Float64.
 *   _polars_plugin_<f>(inputs, n_inputs, kwargs, kwargs_len, return_value, context)
 *       inputs: `n_inputs` exported Series -- the columns in the reference's order (high, low, close, volume ...), optionally
 *       followed by the period as a literal Series (momentum.rs / volatility.rs: `inputs[k].i64()?.get(0)`).  A column may be
 *       of any numeric Arrow type -- int8 .. uint64 (c C s S i I l L), float16 / 32 / 64 (e f g) or Boolean (b): it is cast to
 *       Float64 on the way in, as every reference function does with `inputs[k].cast(&DataType::Float64)?` (overlap.rs:120,129,
 *       momentum.rs:12, volume.rs:19-31, pattern.rs:11-17, cycle.rs:11); only a non-numeric column is an error;
 *       kwargs: pickle bytes of {"timeperiod": n} or NULL/0 (overlap.rs:11-28 MaKwargs); return_value: filled on success
 *       (release != NULL), left zeroed on failure with the message in _polars_plugin_get_last_error_message().  A null in the
 *       input of a function that goes through `rechunk().cont_slice()?` in the reference (momentum / cycle family) is such a failure.
 *   _polars_plugin_field_<f>(fields, n_fields, return_value, kwargs, kwargs_len)
 *       output field: the first input's name, Float64 (`#[polars_expr(output_type=Float64)]`; Int32 for the patterns)
 * reference: overlap.rs:494 sma, :127 ema, :531 wma, :119 dema, :513 tema, :522 trima, :137 kama, :180 midpoint, :281 midprice;
 *   momentum.rs:507 rsi, :181 cmo, :384 mom, :439 roc, :456 rocp, :473 rocr, :490 rocr100, :544 trix, :11 adx, :32 adxr, :226 dx,
 *   :400 plus_di, :345 minus_di, :414 plus_dm, :359 minus_dm, :113 bop, :138 cci, :286 mfi, :630 willr;
 *   volatility.rs:18 atr, :34 natr, :51 trange; volume.rs:19 ad, :70 obv; price.rs:10 avgprice, :33 medprice, :52 typprice,
 *   :73 wclprice;
 *   cycle.rs:10 ht_dcperiod, :75 ht_dcphase, :310 ht_trendline; aroonosc: registered by python/polars_quant/talib/momentum.py:42,
 *   absent from the Rust library (decision D-6: AROON up - down) */
#define PQ_PLUGIN_DECL(NAME)                                                                                                   \
    void _polars_plugin_##NAME(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len,            \
                               pq_series_export *return_value, void *context);                                                \
    void _polars_plugin_field_##NAME(struct ArrowSchema *fields, size_t n_fields, struct ArrowSchema *return_value,            \
                                     const uint8_t *kwargs, size_t kwargs_len);                                               \
    void _polars_plugin_##NAME##_over(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len,     \
                                      pq_series_export *return_value, void *context);                                         \
    void _polars_plugin_field_##NAME##_over(struct ArrowSchema *fields, size_t n_fields, struct ArrowSchema *return_value,     \
                                            const uint8_t *kwargs, size_t kwargs_len);
PQ_PLUGIN_DECL(sma) PQ_PLUGIN_DECL(ema) PQ_PLUGIN_DECL(wma) PQ_PLUGIN_DECL(dema) PQ_PLUGIN_DECL(tema) PQ_PLUGIN_DECL(trima)
PQ_PLUGIN_DECL(kama) PQ_PLUGIN_DECL(midpoint) PQ_PLUGIN_DECL(rsi) PQ_PLUGIN_DECL(cmo) PQ_PLUGIN_DECL(mom) PQ_PLUGIN_DECL(roc)
PQ_PLUGIN_DECL(rocp) PQ_PLUGIN_DECL(rocr) PQ_PLUGIN_DECL(rocr100) PQ_PLUGIN_DECL(trix) PQ_PLUGIN_DECL(ht_dcperiod)
PQ_PLUGIN_DECL(ht_dcphase) PQ_PLUGIN_DECL(ht_trendline) PQ_PLUGIN_DECL(midprice) PQ_PLUGIN_DECL(plus_dm) PQ_PLUGIN_DECL(minus_dm)
PQ_PLUGIN_DECL(aroonosc) PQ_PLUGIN_DECL(medprice) PQ_PLUGIN_DECL(obv) PQ_PLUGIN_DECL(adx) PQ_PLUGIN_DECL(adxr) PQ_PLUGIN_DECL(dx)
PQ_PLUGIN_DECL(plus_di) PQ_PLUGIN_DECL(minus_di) PQ_PLUGIN_DECL(cci) PQ_PLUGIN_DECL(willr) PQ_PLUGIN_DECL(atr) PQ_PLUGIN_DECL(natr)
PQ_PLUGIN_DECL(trange) PQ_PLUGIN_DECL(typprice) PQ_PLUGIN_DECL(wclprice) PQ_PLUGIN_DECL(mfi) PQ_PLUGIN_DECL(bop) PQ_PLUGIN_DECL(ad)
PQ_PLUGIN_DECL(avgprice)

/* functions with other scalar parameters: each from the pickled kwargs by name, else from its trailing literal input (in this order),
 * else the reference's default.  ma(timeperiod 30, matype 0) overlap.rs:146; t3(timeperiod 5, vfactor 0.0) :503;
 * sar(acceleration 0.0, maximum 0.0) :437; sarext(startvalue, offsetonreverse, accelerationinitlong, accelerationlong,
 * accelerationmaxlong, accelerationinitshort, accelerationshort, accelerationmaxshort: all 0.0) :457;
 * ultosc(timeperiod1 7, timeperiod2 14, timeperiod3 28) momentum.rs:572; adosc(fastperiod 3, slowperiod 10) volume.rs:34;
 * ht_trendmode -> Int32 cycle.rs:377 */
PQ_PLUGIN_DECL(ma) PQ_PLUGIN_DECL(t3) PQ_PLUGIN_DECL(ultosc) PQ_PLUGIN_DECL(adosc) PQ_PLUGIN_DECL(sar) PQ_PLUGIN_DECL(sarext)
PQ_PLUGIN_DECL(ht_trendmode)
/* apo / ppo(real; fastperiod 12, slowperiod 26, matype 0): registered by python/polars_quant/talib/momentum.py:25-30, :136-141 */
PQ_PLUGIN_DECL(apo) PQ_PLUGIN_DECL(ppo)
/* mavp(real, periods; minperiod 2, maxperiod 30, matype 0) overlap.rs:407: the period column may be Int64 / Int32 / Float64 */
PQ_PLUGIN_DECL(mavp)

/* Struct-valued functions: return_value is one "+s" array whose children are Float64 columns; struct and field names are the
 * reference's: bbands{bb_upper, bb_middle, bb_lower}(timeperiod 20, nbdevup 2.0, nbdevdn 2.0) overlap.rs:30-47;
 * mama{mama, fama}(fastlimit 0.0, slowlimit 0.0) :40-44,:156; aroon{aroon_up, aroon_down}(timeperiod 14) momentum.rs:63-70;
 * macd_res{macd, macd_signal, macd_hist}(fastperiod 12, slowperiod 26, signalperiod 9) :239-250; ht_phasor{inphase, quadrature}
 * cycle.rs:149-159; ht_sine{sine, leadsine} :229-236.  _polars_plugin_field_<f> returns the same Struct field. */
PQ_PLUGIN_DECL(bbands) PQ_PLUGIN_DECL(mama) PQ_PLUGIN_DECL(aroon) PQ_PLUGIN_DECL(macd) PQ_PLUGIN_DECL(ht_phasor) PQ_PLUGIN_DECL(ht_sine)

/* the 61 candlestick recognisers (pattern.rs:10-2062): inputs open, high, low, close[, penetration as a Float64 literal, default 0.3];
 * Int32 output, never null; a null in an input is an error (cont_slice) */
PQ_PLUGIN_DECL(cdl2crows) PQ_PLUGIN_DECL(cdl3blackcrows) PQ_PLUGIN_DECL(cdl3inside) PQ_PLUGIN_DECL(cdl3linestrike)
PQ_PLUGIN_DECL(cdl3outside) PQ_PLUGIN_DECL(cdl3starsinsouth) PQ_PLUGIN_DECL(cdl3whitesoldiers)
PQ_PLUGIN_DECL(cdlabandonedbaby) PQ_PLUGIN_DECL(cdladvanceblock) PQ_PLUGIN_DECL(cdlbelthold) PQ_PLUGIN_DECL(cdlbreakaway)
PQ_PLUGIN_DECL(cdlclosingmarubozu) PQ_PLUGIN_DECL(cdlconcealbabyswall) PQ_PLUGIN_DECL(cdlcounterattack)
PQ_PLUGIN_DECL(cdldarkcloudcover) PQ_PLUGIN_DECL(cdldoji) PQ_PLUGIN_DECL(cdldojistar) PQ_PLUGIN_DECL(cdldragonflydoji)
PQ_PLUGIN_DECL(cdlengulfing) PQ_PLUGIN_DECL(cdleveningdojistar) PQ_PLUGIN_DECL(cdleveningstar)
PQ_PLUGIN_DECL(cdlgapsidesidewhite) PQ_PLUGIN_DECL(cdlgravestonedoji) PQ_PLUGIN_DECL(cdlhammer)
PQ_PLUGIN_DECL(cdlhangingman) PQ_PLUGIN_DECL(cdlharami) PQ_PLUGIN_DECL(cdlharamicross) PQ_PLUGIN_DECL(cdlhighwave)
PQ_PLUGIN_DECL(cdlhikkake) PQ_PLUGIN_DECL(cdlhikkakemod) PQ_PLUGIN_DECL(cdlhomingpigeon) PQ_PLUGIN_DECL(cdlidentical3crows)
PQ_PLUGIN_DECL(cdlinneck) PQ_PLUGIN_DECL(cdlinvertedhammer) PQ_PLUGIN_DECL(cdlkicking) PQ_PLUGIN_DECL(cdlkickingbylength)
PQ_PLUGIN_DECL(cdlladderbottom) PQ_PLUGIN_DECL(cdllongleggeddoji) PQ_PLUGIN_DECL(cdllongline) PQ_PLUGIN_DECL(cdlmarubozu)
PQ_PLUGIN_DECL(cdlmatchinglow) PQ_PLUGIN_DECL(cdlmathold) PQ_PLUGIN_DECL(cdlmorningdojistar) PQ_PLUGIN_DECL(cdlmorningstar)
PQ_PLUGIN_DECL(cdlonneck) PQ_PLUGIN_DECL(cdlpiercing) PQ_PLUGIN_DECL(cdlrickshawman) PQ_PLUGIN_DECL(cdlrisefall3methods)
PQ_PLUGIN_DECL(cdlseparatinglines) PQ_PLUGIN_DECL(cdlshootingstar) PQ_PLUGIN_DECL(cdlshortline)
PQ_PLUGIN_DECL(cdlspinningtop) PQ_PLUGIN_DECL(cdlstalledpattern) PQ_PLUGIN_DECL(cdlsticksandwich) PQ_PLUGIN_DECL(cdltakuri)
PQ_PLUGIN_DECL(cdltasukigap) PQ_PLUGIN_DECL(cdlthrusting) PQ_PLUGIN_DECL(cdltristar) PQ_PLUGIN_DECL(cdlunique3river)
PQ_PLUGIN_DECL(cdlupsidegap2crows) PQ_PLUGIN_DECL(cdlxsidegap3methods)

/* host-only helper behind the kwargs path (CPU-testable): the int64 value of `key` in a pickled dict of scalars.
 * returns 1 found, 0 absent or None, -1 malformed / unsupported pickle */
int32_t pq_plugin_kwargs_i64(const uint8_t *pickle, size_t len, const char *key, int64_t *out);

/* The input-column cache (csrc/plugin.hip).  A Float64 column that arrives as ONE chunk without nulls is uploaded straight from its Arrow
 * buffer and its device copy is kept under (buffer address, rows, layout of the call) + a 64-bit hash of the WHOLE buffer, recomputed on
 * every call: the sixty expressions of one `with_columns` (python/polars_quant/talib/momentum.py:13-16: one plugin call per expression)
 * upload `close` once, and a freed-and-reused address with other content is a miss.  PQ_PLUGIN_CACHE_MB (default 2048; 0 = no cache) bounds
 * the device memory it holds; least recently used entries that no call holds go first.  No reference counterpart.
 * Host side of the results: an exported column owns uninitialised, page-touched host memory; its release hands blocks of >= 1 MB to a
 * bounded pool (PQ_PLUGIN_HOSTPOOL_MB, default 1024; 0 = off) that later calls of about that size draw from instead of faulting in fresh
 * pages (fresh blocks of >= 4 MB are 2 MB-aligned and ask for transparent huge pages).  PQ_PLUGIN_TIMING=1 prints the host time of every phase of a call on stderr.
 *   pq_plugin_cache_stats   hits / misses / bytes held / entries since load (or the last clear); any pointer may be NULL
 *   pq_plugin_cache_clear   frees every entry no call holds, zeroes the counters */
void pq_plugin_cache_stats(int64_t *hits, int64_t *misses, int64_t *bytes, int64_t *entries);
void pq_plugin_cache_clear(void);

#ifdef __cplusplus
}
#endif
#endif
