/*
 * pq_polars_plugin.h -- the Polars expression-plugin symbols of libpolars_quant_hip.so (a spike for two
 * functions): what `polars.plugins.register_plugin_function(plugin_path=<.so>, function_name="ema")` resolves with dlsym
 * (python/polars_quant/talib/overlap.py:36-43) and what `#[polars_expr]` generates in the reference for
 * src/talib/overlap.rs:127-134 (`ema`) and :494-500 (`sma`).
 *
 * ABI: the C structs of pyo3-polars 0.26 / polars-ffi (version 0.1) as far as their published layout is known --
 * UNVERIFIED against the pinned polars 0.53 (Cargo.lock:950-951): no `polars` wheel exists in this image, so the entry
 * points are exercised by tests/ with Arrow C Data Interface structs built by pyarrow, not by Polars itself.
 *
 * Parameters: both conventions of the reference are accepted: pickled kwargs ({"timeperiod": 20}, the Rust side,
 * overlap.rs:18-22) and a trailing literal input Series of one i64 row (the Python wrapper's `args=[real, timeperiod]`).
 * Each call is ONE series (Polars calls once per expression and once per group under .over("symbol")): n_series = 1.
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
/* inputs: `n_inputs` exported Series; kwargs: pickle bytes or NULL/0; return_value: filled on success (release != NULL) */
void _polars_plugin_ema(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len,
                        pq_series_export *return_value, void *context);                      /* overlap.rs:127 */
void _polars_plugin_sma(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len,
                        pq_series_export *return_value, void *context);                      /* overlap.rs:494 */
/* output field (name of the first input, Float64): `#[polars_expr(output_type=Float64)]` */
void _polars_plugin_field_ema(struct ArrowSchema *fields, size_t n_fields, struct ArrowSchema *return_value,
                              const uint8_t *kwargs, size_t kwargs_len);
void _polars_plugin_field_sma(struct ArrowSchema *fields, size_t n_fields, struct ArrowSchema *return_value,
                              const uint8_t *kwargs, size_t kwargs_len);

/* host-only helper behind the kwargs path (CPU-testable): the int64 value of `key` in a pickled dict of scalars.
 * returns 1 found, 0 absent or None, -1 malformed / unsupported pickle */
int32_t pq_plugin_kwargs_i64(const uint8_t *pickle, size_t len, const char *key, int64_t *out);

#ifdef __cplusplus
}
#endif
#endif
