// factor.hip -- SURVEY 8(f) rank 3: cross-sectional factor evaluation, Factor.ic / rank_ic / rolling_ic
// (README.md:1429-1430, :1480-1482, :1626-1634; README-only => decision D-12, oracle/backtest.c pqo_factor_ic).
//
// The columns are symbol-major [n_series][stride]; a day's cross-section is a strided column.
//  * Pearson IC: the sums are order-sensitive, so the oracle's order IS the definition: blocks of 256 symbols, ascending
//    inside a block, block sums added in ascending order.  One (day, block) per thread; consecutive threads read
//    consecutive days -> coalesced; 16 loads per column in flight.
//  * Rank IC: a tiled transpose builds day-major key rows (invalid pairs -> +inf), rocPRIM's segmented radix sort orders
//    every day's row (keys + symbol ids), a per-day workgroup turns sorted positions into average ranks (ties share the
//    mean rank) and accumulates the five rank sums.  Ranks are half-integers, so the sums are exact in f64 in ANY order
//    (n_series <= 100 000) and the closed form below is bit-identical to the oracle.
#include <cstring>
#include "pq_dev.h"
#include <rocprim/rocprim.hpp>

__device__ __forceinline__ bool ic_valid(double a, double b) { return !pq_isnull(a) && !pq_isnull(b) && isfinite(a) && isfinite(b); }

// Pearson IC in four launches: cross-sectional sums are DEFINED over blocks of 256 symbols (ascending inside a block, block
// sums added in ascending order, oracle PQO_SUM_BLOCK), which gives (len/64) x (n/256) workgroups instead of len/64.
constexpr int IC_BLOCK = 256;
template <int PASS> // 0: n, sum x, sum y   1: centred sums given the means
__global__ __launch_bounds__(64) void ic_partial_kernel(const double *x, const double *y, Dims d, const double *means, double *part) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= d.len) return;
    const int64_t s_lo = (int64_t)blockIdx.y * IC_BLOCK, s_hi = s_lo + IC_BLOCK < d.n ? s_lo + IC_BLOCK : d.n;
    double mx = 0.0, my = 0.0;
    if (PASS == 1) { mx = means[t * 3 + 1]; my = means[t * 3 + 2]; }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    constexpr int B = 16;
    for (int64_t s0 = s_lo; s0 < s_hi; s0 += B) {
        double a[B], b[B];
#pragma unroll
        for (int k = 0; k < B; k++) {
            const int64_t s = s0 + k < s_hi ? s0 + k : s_hi - 1;
            a[k] = x[s * d.stride + t]; b[k] = y[s * d.stride + t];
        }
#pragma unroll
        for (int k = 0; k < B; k++)
            if (s0 + k < s_hi && ic_valid(a[k], b[k])) {
                if (PASS == 0) { a0 += 1.0; a1 += a[k]; a2 += b[k]; }
                else { const double dx = a[k] - mx, dy = b[k] - my; a0 += dx * dy; a1 += dx * dx; a2 += dy * dy; }
            }
    }
    double *o = part + ((int64_t)blockIdx.y * d.len + t) * 3;
    o[0] = a0; o[1] = a1; o[2] = a2;
}
template <int PASS>
__global__ __launch_bounds__(64) void ic_combine_kernel(const double *part, int64_t nblk, int64_t len, double *means, double *ic, int32_t *n_valid) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= len) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int64_t k = 0; k < nblk; k++) {
        const double *o = part + (k * len + t) * 3;
        a0 += o[0]; a1 += o[1]; a2 += o[2];
    }
    if (PASS == 0) { // a0 = n (a sum of small integers: exact), means
        means[t * 3] = a0;
        means[t * 3 + 1] = a0 > 0.0 ? a1 / a0 : 0.0;
        means[t * 3 + 2] = a0 > 0.0 ? a2 / a0 : 0.0;
        if (n_valid) n_valid[t] = (int32_t)a0;
    } else {
        const double n = means[t * 3];
        ic[t] = (n >= 2.0 && a1 > 0.0 && a2 > 0.0) ? a0 / (sqrt(a1) * sqrt(a2)) : pq_null();
    }
}

// [n][stride] -> day-major [len][n] keys (+inf where the pair is invalid), symbol ids, per-day valid counts
__global__ __launch_bounds__(256) void rank_prep_kernel(const double *x, const double *y, Dims d, double *kx, double *ky,
                                                        unsigned *ids, int32_t *n_valid) {
    __shared__ double tx[32][33], ty[32][33];
    const int64_t t0 = (int64_t)blockIdx.x * 32, s0 = (int64_t)blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5; // 32 x 8
    const double inf = __longlong_as_double(0x7FF0000000000000LL);
    for (int r = ly; r < 32; r += 8) { // rows = symbols, columns = days (coalesced along t)
        const int64_t s = s0 + r, t = t0 + lx;
        double a = inf, b = inf;
        if (s < d.n && t < d.len) {
            a = x[s * d.stride + t]; b = y[s * d.stride + t];
            if (!ic_valid(a, b)) { a = inf; b = inf; }
        }
        tx[r][lx] = a; ty[r][lx] = b;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) { // rows = days, columns = symbols (coalesced along s)
        const int64_t t = t0 + r, s = s0 + lx;
        const bool in = t < d.len && s < d.n;
        const double a = tx[lx][r], b = ty[lx][r];
        if (in) {
            kx[t * d.n + s] = a; ky[t * d.n + s] = b;
            ids[t * d.n + s] = (unsigned)s;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(in && a != inf);
        if (lx == 0 && t < d.len) { // a wave holds two of the tile's days: lanes 0-31 one, lanes 32-63 the other
            const unsigned half = (unsigned)((m >> ((threadIdx.x & 32) ? 32 : 0)) & 0xffffffffULL);
            if (half) atomicAdd(&n_valid[t], __popc(half));
        }
    }
}
// [a, b) = the run of entries equal to ks[i] in the sorted row ks[0 .. nv): a short linear walk (runs are single elements for
// continuous factors), then binary searches -- a discrete or constant factor (signals, buckets, all zeros before a warm-up)
// has runs of thousands of entries and a purely linear walk would cost O(nv^2) loads per day.
__device__ __forceinline__ void tie_run(const double *ks, int nv, int i, int &a, int &b) {
    const double key = ks[i];
    a = i; b = i + 1;
    int steps = 0;
    while (a > 0 && steps < 4 && ks[a - 1] == key) { a--; steps++; }
    if (a > 0 && ks[a - 1] == key) { // lower bound in [0, a)
        int lo = 0, hi = a - 1;      // ks[hi] == key; find the first index with ks == key
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ks[mid] < key) lo = mid + 1; else hi = mid; }
        a = lo;
    }
    steps = 0;
    while (b < nv && steps < 4 && ks[b] == key) { b++; steps++; }
    if (b < nv && ks[b] == key) {    // upper bound in (b, nv]
        int lo = b, hi = nv;         // ks[lo] == key; find the first index with ks > key
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ks[mid] <= key) lo = mid + 1; else hi = mid; }
        b = lo;
    }
}
// y side: ranks by symbol.  One workgroup per day; ks/is = the day's sorted keys / symbol ids.
__global__ __launch_bounds__(256) void tie_rank_scatter_kernel(const double *ks, const unsigned *is, const int32_t *n_valid, int64_t n,
                                                               double *rank_by_symbol) {
    const int64_t t = blockIdx.x, base = t * n;
    const int nv = n_valid[t];
    for (int i = threadIdx.x; i < nv; i += 256) {
        int a, b;
        tie_run(ks + base, nv, i, a, b);
        rank_by_symbol[base + is[base + i]] = ((double)(a + 1) + (double)b) / 2.0;
    }
}
// x side: ranks on the fly + the five sums + the closed form
__global__ __launch_bounds__(256) void rank_corr_kernel(const double *ks, const unsigned *is, const int32_t *n_valid, int64_t n,
                                                        const double *ry_by_symbol, double *ic) {
    const int64_t t = blockIdx.x, base = t * n;
    const int nv = n_valid[t];
    double Sx = 0.0, Sy = 0.0, Sxx = 0.0, Syy = 0.0, Sxy = 0.0;
    for (int i = threadIdx.x; i < nv; i += 256) {
        int a, b;
        tie_run(ks + base, nv, i, a, b);
        const double rx = ((double)(a + 1) + (double)b) / 2.0, ry = ry_by_symbol[base + is[base + i]];
        Sx += rx; Sy += ry; Sxx += rx * rx; Syy += ry * ry; Sxy += rx * ry;
    }
    __shared__ double red[5][256];
    red[0][threadIdx.x] = Sx; red[1][threadIdx.x] = Sy; red[2][threadIdx.x] = Sxx; red[3][threadIdx.x] = Syy; red[4][threadIdx.x] = Sxy;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { // every partial sum is an exact multiple of 1/4 below 2^51: any order is exact
        if ((int)threadIdx.x < w)
            for (int k = 0; k < 5; k++) red[k][threadIdx.x] += red[k][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double out = pq_null();
        if (nv >= 2) {
            const double nn = (double)nv;
            const double vx = nn * red[2][0] - red[0][0] * red[0][0], vy = nn * red[3][0] - red[1][0] * red[1][0];
            if (vx > 0.0 && vy > 0.0) out = (nn * red[4][0] - red[0][0] * red[1][0]) / (sqrt(vx) * sqrt(vy));
        }
        ic[t] = out;
    }
}
__global__ __launch_bounds__(256) void iota_offsets_kernel(unsigned *off, int64_t segs, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i <= segs) off[i] = (unsigned)(i * n);
}
__global__ __launch_bounds__(256) void rolling_ic_kernel(const double *ic, int64_t n, int64_t w, double *ric, double *rir) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    double m = pq_null(), r = pq_null();
    if (w > 0 && t + 1 >= w) {
        bool ok = true;
        double sum = 0.0;
        for (int64_t j = t - w + 1; j <= t; j++) { const double v = ic[j]; if (pq_isnull(v)) { ok = false; break; } sum += v; }
        if (ok) {
            m = sum / (double)w;
            if (w >= 2) {
                double vs = 0.0;
                for (int64_t j = t - w + 1; j <= t; j++) { const double dlt = ic[j] - m; vs += dlt * dlt; }
                const double sd = sqrt(vs / (double)(w - 1));
                if (sd > 0.0) r = m / sd;
            }
        }
    }
    ric[t] = m; rir[t] = r;
}

extern "C" {

pq_status pq_factor_ic(pq_ctx *ctx, const pq_batch *b, const double *factor, const double *fwd_return, int32_t method, double *ic,
                       int32_t *n_valid) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(factor && fwd_return && ic, "pq_factor_ic: null pointer");
    PQ_REQUIRE(method == 0 || method == 1, "pq_factor_ic: method must be 0 (Pearson IC) or 1 (Spearman Rank-IC)");
    if (ctx->rec) { pq_set_error("pq_factor_ic cannot be recorded into a suite"); return PQ_ERR_UNSUPPORTED; }
    if (b->len == 0) return PQ_OK;
    const Dims d = dims_of(b);
    if (method == 0 || b->n_series == 0) {
        const int64_t nblk = (d.n + IC_BLOCK - 1) / IC_BLOCK > 0 ? (d.n + IC_BLOCK - 1) / IC_BLOCK : 1;
        const size_t part_bytes = (size_t)nblk * (size_t)d.len * 24, means_bytes = (size_t)d.len * 24;
        PQ_TRY(pq_ws_reserve(ctx, part_bytes + means_bytes));
        double *part = (double *)ctx->ws, *means = (double *)((unsigned char *)ctx->ws + part_bytes);
        const dim3 gp((unsigned)((d.len + 63) / 64), (unsigned)nblk), gc((unsigned)((d.len + 63) / 64));
        hipLaunchKernelGGL(ic_partial_kernel<0>, gp, dim3(64), 0, ctx->stream, factor, fwd_return, d, (const double *)nullptr, part);
        hipLaunchKernelGGL(ic_combine_kernel<0>, gc, dim3(64), 0, ctx->stream, part, nblk, d.len, means, ic, n_valid);
        hipLaunchKernelGGL(ic_partial_kernel<1>, gp, dim3(64), 0, ctx->stream, factor, fwd_return, d, (const double *)means, part);
        hipLaunchKernelGGL(ic_combine_kernel<1>, gc, dim3(64), 0, ctx->stream, part, nblk, d.len, means, ic, n_valid);
        PQ_HIP_TRY(hipGetLastError());
        return PQ_OK;
    }
    PQ_REQUIRE(b->n_series <= 100000, "pq_factor_ic: rank IC supports at most 100000 series (exact rank sums)");
    const size_t cells = (size_t)d.len * (size_t)d.n;
    PQ_REQUIRE(cells < (1ull << 32), "pq_factor_ic: rank IC needs n_series * len < 2^32");
    // workspace: kx, ky, sx, sy (f64) | ry (f64) | ids, ix, iy (u32) | offsets (u32) | counts (i32) | rocPRIM temp
    size_t tmp_bytes = 0;
    PQ_HIP_TRY(rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, (double *)nullptr, (double *)nullptr, (unsigned *)nullptr,
                                                   (unsigned *)nullptr, (unsigned)cells, (unsigned)d.len, (unsigned *)nullptr,
                                                   (unsigned *)nullptr, 0, 64, ctx->stream));
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_kx = 0, o_ky = o_kx + al(cells * 8), o_sx = o_ky + al(cells * 8), o_sy = o_sx + al(cells * 8), o_ry = o_sy + al(cells * 8),
                 o_id = o_ry + al(cells * 8), o_ix = o_id + al(cells * 4), o_iy = o_ix + al(cells * 4), o_off = o_iy + al(cells * 4),
                 o_cnt = o_off + al((size_t)(d.len + 1) * 4), o_tmp = o_cnt + al((size_t)d.len * 4), total = o_tmp + al(tmp_bytes);
    PQ_TRY(pq_ws_reserve(ctx, total));
    unsigned char *w = (unsigned char *)ctx->ws;
    double *kx = (double *)(w + o_kx), *ky = (double *)(w + o_ky), *sx = (double *)(w + o_sx), *sy = (double *)(w + o_sy), *ry = (double *)(w + o_ry);
    unsigned *ids = (unsigned *)(w + o_id), *ix = (unsigned *)(w + o_ix), *iy = (unsigned *)(w + o_iy), *off = (unsigned *)(w + o_off);
    int32_t *cnt = (int32_t *)(w + o_cnt);
    PQ_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)d.len * 4, ctx->stream));
    hipLaunchKernelGGL(rank_prep_kernel, dim3((unsigned)((d.len + 31) / 32), (unsigned)((d.n + 31) / 32)), dim3(256), 0, ctx->stream, factor,
                       fwd_return, d, kx, ky, ids, cnt);
    hipLaunchKernelGGL(iota_offsets_kernel, dim3((unsigned)((d.len + 256) / 256)), dim3(256), 0, ctx->stream, off, d.len, d.n);
    PQ_HIP_TRY(rocprim::segmented_radix_sort_pairs(w + o_tmp, tmp_bytes, kx, sx, ids, ix, (unsigned)cells, (unsigned)d.len, off, off + 1, 0, 64,
                                                   ctx->stream));
    PQ_HIP_TRY(rocprim::segmented_radix_sort_pairs(w + o_tmp, tmp_bytes, ky, sy, ids, iy, (unsigned)cells, (unsigned)d.len, off, off + 1, 0, 64,
                                                   ctx->stream));
    hipLaunchKernelGGL(tie_rank_scatter_kernel, dim3((unsigned)d.len), dim3(256), 0, ctx->stream, sy, iy, cnt, d.n, ry);
    hipLaunchKernelGGL(rank_corr_kernel, dim3((unsigned)d.len), dim3(256), 0, ctx->stream, sx, ix, cnt, d.n, ry, ic);
    if (n_valid) PQ_HIP_TRY(hipMemcpyAsync(n_valid, cnt, (size_t)d.len * 4, hipMemcpyDeviceToDevice, ctx->stream));
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}

pq_status pq_rolling_ic(pq_ctx *ctx, const double *ic, int64_t len, int64_t window, double *rolling_ic, double *rolling_ir) {
    PQ_REQUIRE(ctx && ic && rolling_ic && rolling_ir, "pq_rolling_ic: null pointer");
    PQ_REQUIRE(len >= 0, "pq_rolling_ic: negative length");
    if (ctx->rec) { pq_set_error("pq_rolling_ic cannot be recorded into a suite"); return PQ_ERR_UNSUPPORTED; }
    if (len == 0) return PQ_OK;
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(rolling_ic_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, ic, len, window, rolling_ic, rolling_ir);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}

} // extern "C"
