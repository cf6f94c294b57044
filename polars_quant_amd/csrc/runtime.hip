// runtime.hip -- context (device + stream + scratch workspace), error reporting, memory helpers and the
// Arrow-validity <-> null-sentinel conversions of the C ABI (include/pq_hip.h).
#include "pq_dev.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void pq_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

pq_status pq_check(pq_ctx *ctx, const pq_batch *b) {
    if (!ctx) { pq_set_error("null context"); return PQ_ERR_ARG; }
    if (!b) { pq_set_error("null batch descriptor"); return PQ_ERR_ARG; }
    if (b->offsets && (b->n_series < 0 || b->len < 0 || b->stride < 0 || b->len > b->stride)) {
        pq_set_error("bad ragged batch: n_series=%lld longest=%lld total rows=%lld", (long long)b->n_series, (long long)b->len, (long long)b->stride);
        return PQ_ERR_ARG;
    }
    if (!b->offsets && (b->n_series < 0 || b->len < 0 || b->stride < b->len)) {
        pq_set_error("bad batch: n_series=%lld len=%lld stride=%lld", (long long)b->n_series, (long long)b->len,
                     (long long)b->stride);
        return PQ_ERR_ARG;
    }
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { pq_set_error("hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e)); return PQ_ERR_HIP; }
    return PQ_OK;
}

extern "C" int64_t pq_recommended_stride(int64_t len) { return len <= 0 ? 0 : (len + 15) / 16 * 16; }
extern "C" pq_status pq_layout_check(const pq_batch *b, const void *const *cols, int32_t n_cols) {
    if (!b || (n_cols > 0 && !cols) || n_cols < 0) { pq_set_error("pq_layout_check: bad argument"); return PQ_ERR_ARG; }
    if (b->offsets) return PQ_OK;
    for (int32_t k = 0; k < n_cols; k++)
        if (reinterpret_cast<uintptr_t>(cols[k]) % 8) {
            pq_set_error("slow layout: column %d is not 8-byte aligned: the per-lane gather forms run", (int)k);
            return PQ_WARN_SLOW_LAYOUT;
        }
    if (b->stride % 2) {
        pq_set_error("slow layout: stride %lld is odd (rows only 8-byte aligned): the 8-byte forms of the tiled kernels run, ~1.5x slower; "
                     "allocate columns at pq_recommended_stride(len) = %lld", (long long)b->stride, (long long)pq_recommended_stride(b->len));
        return PQ_WARN_SLOW_LAYOUT;
    }
    for (int32_t k = 0; k < n_cols; k++)
        if (reinterpret_cast<uintptr_t>(cols[k]) % 16) {
            pq_set_error("slow layout: column %d starts 8 bytes off a 16-byte boundary: the 8-byte forms of the tiled kernels run", (int)k);
            return PQ_WARN_SLOW_LAYOUT;
        }
    return PQ_OK;
}

// ---- ragged -> regular re-housing (pq_dev.h: launch_seq) ---------------------------------------------------------------------------
pq_status rg_reserve(pq_ctx *ctx, size_t bytes) {
    if (ctx->rg_ws_bytes >= bytes) return PQ_OK;
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream)); // grow-only; earlier launches may still read the old block
    if (ctx->rg_ws) PQ_HIP_TRY(hipFree(ctx->rg_ws));
    ctx->rg_ws = nullptr;
    ctx->rg_ws_bytes = 0;
    hipError_t e = hipMalloc(&ctx->rg_ws, bytes);
    if (e != hipSuccess) { pq_set_error("ragged workspace hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return PQ_ERR_NOMEM; }
    ctx->rg_ws_bytes = bytes;
    return PQ_OK;
}
struct RgCols { const double *src[8]; double *dst[8]; int n; };
// one thread per (series, row of the padded pitch); consecutive threads = consecutive rows: both sides coalesced
__global__ __launch_bounds__(256) void rg_pack_kernel(RgCols c, const int64_t *offs, int64_t s_base, int64_t pitch, int64_t *lens) {
    const int64_t s = s_base + blockIdx.y, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= pitch) return;
    const int64_t base = offs[s], len = offs[s + 1] - base;
    if (lens && t == 0) lens[s] = len;
    for (int k = 0; k < c.n; k++) c.dst[k][s * pitch + t] = t < len ? c.src[k][base + t] : 0.0;
}
__global__ __launch_bounds__(256) void rg_unpack_kernel(RgCols c, const int64_t *offs, int64_t s_base, int64_t pitch) {
    const int64_t s = s_base + blockIdx.y, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t base = offs[s], len = offs[s + 1] - base;
    if (t >= len) return;
    for (int k = 0; k < c.n; k++) c.dst[k][base + t] = c.src[k][s * pitch + t];
}
pq_status rg_pack(pq_ctx *ctx, const pq_batch *b, int64_t pitch, const double *const *src, double *const *dst, int n_cols, int64_t *lens) {
    if (n_cols > 8) { pq_set_error("internal: rg_pack takes at most 8 columns"); return PQ_ERR_UNSUPPORTED; }
    RgCols c{};
    c.n = n_cols;
    for (int k = 0; k < n_cols; k++) { c.src[k] = src[k]; c.dst[k] = dst[k]; }
    for (int64_t s0 = 0; s0 < b->n_series; s0 += 65535) { // grid.y is limited to 65535
        const int64_t ns = b->n_series - s0 < 65535 ? b->n_series - s0 : 65535;
        hipLaunchKernelGGL(rg_pack_kernel, dim3((unsigned)((pitch + 255) / 256), (unsigned)ns), dim3(256), 0, ctx->stream, c, b->offsets, s0, pitch, lens);
    }
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}
pq_status rg_unpack(pq_ctx *ctx, const pq_batch *b, int64_t pitch, const double *const *src, double *const *dst, int n_cols) {
    if (n_cols > 8) { pq_set_error("internal: rg_unpack takes at most 8 columns"); return PQ_ERR_UNSUPPORTED; }
    RgCols c{};
    c.n = n_cols;
    for (int k = 0; k < n_cols; k++) { c.src[k] = src[k]; c.dst[k] = dst[k]; }
    for (int64_t s0 = 0; s0 < b->n_series; s0 += 65535) {
        const int64_t ns = b->n_series - s0 < 65535 ? b->n_series - s0 : 65535;
        hipLaunchKernelGGL(rg_unpack_kernel, dim3((unsigned)((b->len + 255) / 256), (unsigned)ns), dim3(256), 0, ctx->stream, c, b->offsets, s0, pitch);
    }
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}

extern "C" pq_status pq_ragged_rehouse_stats(pq_ctx *ctx, int64_t *calls, int32_t reset) {
    if (!ctx || !calls) { pq_set_error("pq_ragged_rehouse_stats: null pointer"); return PQ_ERR_ARG; }
    *calls = ctx->rg_calls;
    if (reset) ctx->rg_calls = 0;
    return PQ_OK;
}

// the context's tile flags for DIRECT launches of a fast form with a gated general path behind it (ops_wt.h, misc.hip): all zero
// between launches -- the gated kernel clears what it consumes, in stream order
pq_status ctx_gate(pq_ctx *ctx, size_t tiles, unsigned **gate) {
    if (ctx->wt_gate_tiles < tiles) {
        PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->wt_gate) (void)hipFree(ctx->wt_gate);
        ctx->wt_gate = nullptr; ctx->wt_gate_tiles = 0;
        if (hipMalloc((void **)&ctx->wt_gate, tiles * sizeof(unsigned)) != hipSuccess || hipMemset(ctx->wt_gate, 0, tiles * sizeof(unsigned)) != hipSuccess) {
            pq_set_error("out of device memory for a gate");
            return PQ_ERR_NOMEM;
        }
        ctx->wt_gate_tiles = tiles;
    }
    *gate = ctx->wt_gate;
    return PQ_OK;
}
pq_status pq_ws_reserve(pq_ctx *ctx, size_t bytes) {
    if (ctx->ws_bytes >= bytes) return PQ_OK;
    // grow-only; earlier launches may still read the old block -> drain the stream before freeing it
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->ws) PQ_HIP_TRY(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    hipError_t e = hipMalloc(&ctx->ws, bytes);
    if (e != hipSuccess) { pq_set_error("workspace hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return PQ_ERR_NOMEM; }
    ctx->ws_bytes = bytes;
    return PQ_OK;
}

// ------------------------------------------------------------------ null conversions
__global__ void nulls_from_arrow_kernel(double *col, const uint8_t *bits, int64_t bit_offset, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t bi = bit_offset + i;
    if (!((bits[bi >> 3] >> (bi & 7)) & 1)) col[i] = pq_null();
}
// one thread per output byte (8 rows); null count via per-wave popcount + one atomic per wave
__global__ void validity_to_arrow_kernel(const double *col, int64_t n, uint8_t *bits, unsigned long long *null_count) {
    int64_t byte = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t nbytes = (n + 7) >> 3;
    unsigned nulls = 0;
    if (byte < nbytes) {
        unsigned v = 0;
        for (int k = 0; k < 8; k++) {
            int64_t i = byte * 8 + k;
            if (i < n) {
                if (!pq_isnull(col[i])) v |= 1u << k; else nulls++;
            }
        }
        bits[byte] = (uint8_t)v;
    }
    for (int off = 32; off > 0; off >>= 1) nulls += __shfl_down(nulls, off, 64);
    if ((threadIdx.x & 63) == 0 && nulls && null_count) atomicAdd(null_count, (unsigned long long)nulls);
}
__global__ void count_nulls_kernel(const double *col, Dims d, unsigned long long *count) {
    const int64_t s = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nulls = (t < dims_len(d, s) && pq_isnull(col[dims_base(d, s) + t])) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) nulls += __shfl_down(nulls, off, 64);
    if ((threadIdx.x & 63) == 0 && nulls) atomicAdd(count, (unsigned long long)nulls);
}

extern "C" {

int32_t pq_abi_version(void) { return 1; }
const char *pq_last_error(void) { return g_err; }

pq_status pq_device_count(int32_t *count) {
    if (!count) { pq_set_error("pq_device_count: null pointer"); return PQ_ERR_ARG; }
    int n = 0;
    PQ_HIP_TRY(hipGetDeviceCount(&n));
    *count = n;
    return PQ_OK;
}

pq_status pq_ctx_create(int32_t device, void *hip_stream, pq_ctx **out) {
    if (!out) { pq_set_error("pq_ctx_create: null pointer"); return PQ_ERR_ARG; }
    PQ_HIP_TRY(hipSetDevice(device));
    pq_ctx *c = new pq_ctx();
    c->device = device;
    c->ws = nullptr;
    c->ws_bytes = 0;
    c->d_flag = nullptr;
    c->rec = nullptr;
    c->comm = nullptr;
    c->comm_rank = 0;
    c->comm_world = 0;
    if (hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->cus <= 0) c->cus = 256;
    c->stream = (hipStream_t)hip_stream; // NULL = the device's default (null) stream
    c->own_stream = false;
    hipError_t e = hipMalloc((void **)&c->d_flag, 64 * sizeof(int64_t)); // [0] reduction scalar, [4..6] wave-backtest statistics (+ [8..23] profiling builds), [24..27] wave-per-symbol indicators (+ [32..63] profiling builds)
    if (e == hipSuccess) e = hipMemset(c->d_flag, 0, 64 * sizeof(int64_t));
    if (e != hipSuccess) { if (c->own_stream) (void)hipStreamDestroy(c->stream); delete c; pq_set_error("hipMalloc: %s", hipGetErrorString(e)); return PQ_ERR_NOMEM; }
    *out = c;
    return PQ_OK;
}
pq_status pq_ctx_destroy(pq_ctx *ctx) {
    if (!ctx) return PQ_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)pq_comm_destroy(ctx);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->d_flag) (void)hipFree(ctx->d_flag);
    if (ctx->wt_gate) (void)hipFree(ctx->wt_gate);
    if (ctx->rg_ws) (void)hipFree(ctx->rg_ws);
    for (hipStream_t &st : ctx->suite_aux) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PQ_OK;
}
pq_status pq_ctx_set_stream(pq_ctx *ctx, void *hip_stream) {
    if (!ctx) { pq_set_error("null context"); return PQ_ERR_ARG; }
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) { (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    ctx->stream = (hipStream_t)hip_stream;
    return PQ_OK;
}
pq_status pq_ctx_sync(pq_ctx *ctx) {
    if (!ctx) { pq_set_error("null context"); return PQ_ERR_ARG; }
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PQ_OK;
}
pq_status pq_malloc(pq_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) { pq_set_error("pq_malloc: null pointer"); return PQ_ERR_ARG; }
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) { pq_set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return PQ_ERR_NOMEM; }
    return PQ_OK;
}
pq_status pq_free(pq_ctx *ctx, void *dptr) {
    if (!ctx) { pq_set_error("null context"); return PQ_ERR_ARG; }
    if (dptr) { PQ_HIP_TRY(hipStreamSynchronize(ctx->stream)); PQ_HIP_TRY(hipFree(dptr)); }
    return PQ_OK;
}
// Pin a host buffer (e.g. an Arrow data buffer) so that pq_memcpy_h2d / pq_memcpy_d2h move it by DMA straight from / to
// the caller's memory instead of through the runtime's staging copies: the "zero-copy" hand-over of a host column.
pq_status pq_host_register(void *host_ptr, size_t bytes) {
    if (!host_ptr || !bytes) { pq_set_error("pq_host_register: null pointer or empty range"); return PQ_ERR_ARG; }
    PQ_HIP_TRY(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    return PQ_OK;
}
pq_status pq_host_unregister(void *host_ptr) {
    if (!host_ptr) { pq_set_error("pq_host_unregister: null pointer"); return PQ_ERR_ARG; }
    PQ_HIP_TRY(hipHostUnregister(host_ptr));
    return PQ_OK;
}
pq_status pq_memcpy_h2d(pq_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) { pq_set_error("pq_memcpy_h2d: null pointer"); return PQ_ERR_ARG; }
    PQ_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return PQ_OK;
}
pq_status pq_memcpy_d2h(pq_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) { pq_set_error("pq_memcpy_d2h: null pointer"); return PQ_ERR_ARG; }
    PQ_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PQ_OK;
}
pq_status pq_memcpy_h2d_pitched(pq_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows) {
    if (!ctx || !dst || !src) { pq_set_error("pq_memcpy_h2d_pitched: null pointer"); return PQ_ERR_ARG; }
    if (width > dpitch || width > spitch) { pq_set_error("pq_memcpy_h2d_pitched: a row is wider than its pitch"); return PQ_ERR_ARG; }
    if (!width || !rows) return PQ_OK;
    PQ_HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyHostToDevice, ctx->stream));
    return PQ_OK;
}
pq_status pq_memcpy_d2h_pitched(pq_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows) {
    if (!ctx || !dst || !src) { pq_set_error("pq_memcpy_d2h_pitched: null pointer"); return PQ_ERR_ARG; }
    if (width > dpitch || width > spitch) { pq_set_error("pq_memcpy_d2h_pitched: a row is wider than its pitch"); return PQ_ERR_ARG; }
    if (!width || !rows) return PQ_OK;
    PQ_HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToHost, ctx->stream));
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PQ_OK;
}
pq_status pq_nulls_from_arrow(pq_ctx *ctx, double *col, const uint8_t *bits, int64_t bit_offset, int64_t n) {
    if (!ctx || !col || !bits || n < 0 || bit_offset < 0) { pq_set_error("pq_nulls_from_arrow: bad argument"); return PQ_ERR_ARG; }
    if (n == 0) return PQ_OK;
    hipLaunchKernelGGL(nulls_from_arrow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, col, bits, bit_offset, n);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}
pq_status pq_validity_to_arrow(pq_ctx *ctx, const double *col, int64_t n, uint8_t *bits, int64_t *null_count) {
    if (!ctx || !col || !bits || n < 0) { pq_set_error("pq_validity_to_arrow: bad argument"); return PQ_ERR_ARG; }
    if (null_count) PQ_HIP_TRY(hipMemsetAsync(null_count, 0, sizeof(int64_t), ctx->stream));
    if (n == 0) return PQ_OK;
    int64_t nbytes = (n + 7) >> 3;
    hipLaunchKernelGGL(validity_to_arrow_kernel, dim3((unsigned)((nbytes + 255) / 256)), dim3(256), 0, ctx->stream, col, n, bits,
                       (unsigned long long *)null_count);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}
pq_status pq_count_nulls(pq_ctx *ctx, const pq_batch *b, const double *col, int64_t *host_count) {
    PQ_TRY(pq_check(ctx, b));
    if (!col || !host_count) { pq_set_error("pq_count_nulls: null pointer"); return PQ_ERR_ARG; }
    *host_count = 0;
    if (b->n_series == 0 || b->len == 0) return PQ_OK;
    PQ_HIP_TRY(hipMemsetAsync(ctx->d_flag, 0, sizeof(int64_t), ctx->stream));
    for (int64_t s0 = 0; s0 < b->n_series; s0 += 65535) {
        int64_t ns = b->n_series - s0 < 65535 ? b->n_series - s0 : 65535;
        Dims d{ns, b->len, b->stride, b->offsets ? b->offsets + s0 : nullptr};
        hipLaunchKernelGGL(count_nulls_kernel, dim3((unsigned)((b->len + 255) / 256), (unsigned)ns), dim3(256), 0, ctx->stream,
                           col + (b->offsets ? 0 : s0 * b->stride), d, (unsigned long long *)ctx->d_flag);
        PQ_HIP_TRY(hipGetLastError());
    }
    PQ_HIP_TRY(hipMemcpyAsync(host_count, ctx->d_flag, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PQ_OK;
}

} // extern "C"
