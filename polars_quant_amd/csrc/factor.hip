// factor.hip -- SURVEY 8(f) rank 3: cross-sectional factor evaluation, Factor.ic / rank_ic / rolling_ic
// (README.md:1429-1430, :1480-1482, :1626-1634; README-only => decision D-12, oracle/backtest.c pqo_factor_ic).
//
// The columns are symbol-major [n_series][stride]; a day's cross-section is a strided column.
//  * Pearson IC: the sums are order-sensitive, so the oracle's order IS the definition: blocks of 256 symbols, ascending
//    inside a block, block sums added in ascending order.  One (day, block) per thread; consecutive threads read
//    consecutive days -> coalesced; 16 loads per column in flight.
//  * Rank IC: a tiled transpose builds day-major key rows (invalid pairs -> +inf).  Up to 16 384 symbols one workgroup per day
//    sorts the day's keys in LDS and ranks every symbol by binary search (rank_ic_lds_kernel below).  Wider cross-sections:
//    rocPRIM's segmented radix sort orders every day's row (keys + symbol ids) and a per-day workgroup turns sorted positions
//    into average ranks (ties share the mean rank).  Ranks are half-integers, so the five rank sums are exact in f64 in ANY
//    order (n_series <= 100 000) and the closed form is bit-identical to the oracle.
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
            if (ids) ids[t * d.n + s] = (unsigned)s;
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
// ---- Rank IC, n_series <= RANK_LDS_MAX: one workgroup per day, the day's keys sorted in LDS ------------------------------------
// Keys only, no payload: a bitonic network over P = 2^p >= n keys (invalid pairs and the padding are +inf and end up last), every
// thread holding 16 keys in registers so that up to four compare-exchange stages cost one LDS round trip (29 round trips instead of
// 105 stages at P = 16384).  Every merge is written in its all-ascending form -- the first stage of the merge of size k pairs key i
// with its mirror i ^ (k-1), the later stages are plain half-cleaners -- so a compare-exchange is one v_min_f64 + one v_max_f64 with
// static register pairs and no direction selects.  A symbol's rank is then two binary searches of its own key in the sorted row --
// [lb, ub) is its tie run, 2 x rank = lb + ub + 1 -- so nothing is scattered and the rank sums are integer sums (exact; converted
// once).  The closed form is the oracle's.
constexpr int RANK_LDS_MAX = 16384;
#ifndef RK_NB
#define RK_NB 8
#endif
__device__ __forceinline__ int rk_phys(int i) { return i + (i >> 4); } // one pad slot per 16 keys: stride-16 rows hit distinct banks
__device__ __forceinline__ void rk_cmpex(double &a, double &b) {       // a <= b afterwards (keys are never NaN)
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo; b = hi;
}
// orders this wave's LDS writes before its later LDS reads (one wave's DS operations execute in order)
#define RK_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
template <int B> __device__ __forceinline__ void rk_half_clean(double (&r)[16]) { // register bit B
#pragma unroll
    for (int m = 0; m < 16; m++)
        if (!(m & (1 << B))) rk_cmpex(r[m], r[m | (1 << B)]);
}
template <int W> __device__ __forceinline__ void rk_mirror(double (&r)[16]) {     // first stage of a merge of W registers
#pragma unroll
    for (int m = 0; m < 16; m++)
        if (!(m & (W >> 1))) rk_cmpex(r[m], r[m ^ (W - 1)]);
}
// NST stages of one merge on the 16 keys {base | m << SH} (register bits 3 .. 4-NST).  TOP: the group opens the merge, so register
// bit 3 is the mirror stage -- the upper eight registers then hold the keys whose bits below SH are the complement of the task's.
template <int SH, int NST, bool TOP> __device__ __forceinline__ void rk_group(double *S, int task, int n) {
    const int base = (task & ((1 << SH) - 1)) | ((task >> SH) << (SH + 4));
    if ((base | (1 << SH)) >= n) return; // at most one key below n: see rk_sort
    const int base_hi = TOP ? base ^ ((1 << SH) - 1) : base;
    double r[16];
#pragma unroll
    for (int m = 0; m < 16; m++) r[m] = S[rk_phys((m < 8 ? base : base_hi) | (m << SH))];
    if (TOP) rk_mirror<16>(r); else rk_half_clean<3>(r);
    if (NST >= 2) rk_half_clean<2>(r);
    if (NST >= 3) rk_half_clean<1>(r);
    if (NST >= 4) rk_half_clean<0>(r);
#pragma unroll
    for (int m = 0; m < 16; m++) S[rk_phys((m < 8 ? base : base_hi) | (m << SH))] = r[m];
}
// sorts S[0 .. P) ascending (P = 1 << p >= 16, P / 16 tasks spread over the workgroup); ends with a barrier.  S[n .. P) is +inf
// and stays +inf (every compare-exchange keeps the larger key at the larger index), so a compare-exchange that touches an index
// >= n changes nothing and a task with fewer than two keys below n is skipped: the network costs ~n/P of the full one.
__device__ __forceinline__ void rk_sort(double *S, int p, int n, int tid, int nthr) {
    const int ntask = 1 << (p - 4);
    for (int task = tid; task < ntask; task += nthr) { // merges of size 2, 4, 8, 16 in registers
        if (task * 16 + 1 >= n) break;
        double r[16];
#pragma unroll
        for (int m = 0; m < 16; m++) r[m] = S[rk_phys(task * 16 + m)];
        rk_half_clean<0>(r);
        rk_mirror<4>(r); rk_half_clean<0>(r);
        rk_mirror<8>(r); rk_half_clean<1>(r); rk_half_clean<0>(r);
        rk_mirror<16>(r); rk_half_clean<2>(r); rk_half_clean<1>(r); rk_half_clean<0>(r);
#pragma unroll
        for (int m = 0; m < 16; m++) S[rk_phys(task * 16 + m)] = r[m];
    }
    RK_WAVE_SYNC();
    // Thread t of wave w owns task t, and every group with SH <= 6 only touches keys of the wave's own block of 1024 (bits 0..9):
    // such groups are ordered by the wave's own in-order LDS queue, no workgroup barrier, and the waves drift apart so that one
    // wave's LDS round trip overlaps another's min/max.  Only the stages on bits >= 10 (merges of 2048 keys and more) cross waves.
#define RK_RUN(SH, NST, TOP) { for (int task = tid; task < ntask; task += nthr) rk_group<SH, NST, TOP>(S, task, n); }
#define RK_LOCAL(SH, NST, TOP) { RK_RUN(SH, NST, TOP) RK_WAVE_SYNC(); }
#define RK_CROSS(SH, NST) { __syncthreads(); RK_RUN(SH, NST, true) __syncthreads(); RK_LOCAL(6, 2, false) RK_LOCAL(4, 4, false) }
    for (int q = 5; q <= p; q++) {   // merge size 2^q: stage bits q-1 .. 0, up to four per LDS round trip
        switch (q) {
        case 5: RK_LOCAL(1, 1, true) break;
        case 6: RK_LOCAL(2, 2, true) break;
        case 7: RK_LOCAL(3, 3, true) break;
        case 8: RK_LOCAL(4, 4, true) break;
        case 9: RK_LOCAL(5, 1, true) RK_LOCAL(4, 4, false) break;
        case 10: RK_LOCAL(6, 2, true) RK_LOCAL(4, 4, false) break;
        case 11: RK_CROSS(7, 1) break;
        case 12: RK_CROSS(8, 2) break;
        case 13: RK_CROSS(9, 3) break;
        default: RK_CROSS(10, 4) break;
        }
        RK_LOCAL(0, 4, false)
    }
#undef RK_RUN
#undef RK_LOCAL
#undef RK_CROSS
    __syncthreads();
}
// 2 x the average rank of each of NB keys in the sorted row (0 for an invalid key): NB independent binary searches in step, so that
// their LDS latencies overlap; the second search (the end of the tie run) only runs where a key has an equal right neighbour.
template <int NB> __device__ __forceinline__ void rk_rank2(const double *S, int P, const double (&key)[NB], int (&r2)[NB]) {
    const double inf = __longlong_as_double(0x7FF0000000000000LL);
    int lb[NB], ub[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) lb[j] = 0;
    for (int step = P >> 1; step > 0; step >>= 1)
#pragma unroll
        for (int j = 0; j < NB; j++)
            if (S[rk_phys(lb[j] + step - 1)] < key[j]) lb[j] += step;
    bool tie = false;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        ub[j] = lb[j] + 1;
        tie |= key[j] != inf && ub[j] < P && S[rk_phys(ub[j])] == key[j];
    }
    if (tie) { // ub = the number of keys <= key
#pragma unroll
        for (int j = 0; j < NB; j++) ub[j] = 0;
        for (int step = P >> 1; step > 0; step >>= 1)
#pragma unroll
            for (int j = 0; j < NB; j++)
                if (S[rk_phys(ub[j] + step - 1)] <= key[j]) ub[j] += step;
#pragma unroll
        for (int j = 0; j < NB; j++)
            if (ub[j] == P - 1 && S[rk_phys(P - 1)] <= key[j]) ub[j] = P;
    }
#pragma unroll
    for (int j = 0; j < NB; j++) r2[j] = key[j] != inf ? lb[j] + ub[j] + 1 : 0;
}
__global__ __launch_bounds__(1024) void rank_ic_lds_kernel(const double *kx, const double *ky, const int32_t *n_valid, int64_t n, int p,
                                                           double *ic) {
    extern __shared__ __align__(16) unsigned char rank_lds[];
    double *S = (double *)rank_lds;
    const int64_t base = (int64_t)blockIdx.x * n;
    const int P = 1 << p, tid = threadIdx.x, nthr = blockDim.x;
    const double inf = __longlong_as_double(0x7FF0000000000000LL);
    int rx2[16];
    unsigned long long sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int side = 0; side < 2; side++) {
        const double *row = (side ? ky : kx) + base;
        for (int i0 = (tid >> 6) << 10; i0 < P; i0 += nthr << 4) // each wave fills the 1024-key blocks its tasks sort first
            for (int i = i0 + (tid & 63); i < i0 + 1024 && i < P; i += 64) S[rk_phys(i)] = i < n ? row[i] : inf;
        RK_WAVE_SYNC();
        rk_sort(S, p, (int)n, tid, nthr);
#pragma unroll
        for (int m0 = 0; m0 < 16; m0 += RK_NB) {
            double key[RK_NB];
            int r2[RK_NB];
#pragma unroll
            for (int j = 0; j < RK_NB; j++) {
                const int i = tid + (m0 + j) * nthr;
                key[j] = i < n ? row[i] : inf;
            }
            rk_rank2<RK_NB>(S, P, key, r2);
#pragma unroll
            for (int j = 0; j < RK_NB; j++) {
                if (side == 0) rx2[m0 + j] = r2[j];
                else {
                    const unsigned long long a = (unsigned)rx2[m0 + j], c = (unsigned)r2[j];
                    sx += a; sy += c; sxx += a * a; syy += c * c; sxy += a * c;
                }
            }
        }
        __syncthreads();
    }
    unsigned long long v[5] = {sx, sy, sxx, syy, sxy};
#pragma unroll
    for (int k = 0; k < 5; k++)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    unsigned long long *red = (unsigned long long *)rank_lds; // the key row is dead
    if ((tid & 63) == 0)
        for (int k = 0; k < 5; k++) red[(tid >> 6) * 5 + k] = v[k];
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < (nthr >> 6); w++)
            for (int k = 0; k < 5; k++) v[k] += red[w * 5 + k];
        const int nv = n_valid[blockIdx.x];
        double out = pq_null();
        if (nv >= 2) { // ranks are half-integers: the sums below are exact, as are the oracle's
            const double nn = (double)nv, Sx = (double)v[0] / 2.0, Sy = (double)v[1] / 2.0, Sxx = (double)v[2] / 4.0, Syy = (double)v[3] / 4.0,
                         Sxy = (double)v[4] / 4.0;
            const double vx = nn * Sxx - Sx * Sx, vy = nn * Syy - Sy * Sy;
            if (vx > 0.0 && vy > 0.0) out = (nn * Sxy - Sx * Sy) / (sqrt(vx) * sqrt(vy));
        }
        ic[blockIdx.x] = out;
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
    PQ_NO_RAGGED(b, "pq_factor_ic (a cross-section needs every symbol on every day)");
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
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    if (d.n <= RANK_LDS_MAX) { // workspace: kx, ky (f64, day-major) | counts (i32)
        const size_t o_ky = al(cells * 8), o_cnt = o_ky + al(cells * 8), total = o_cnt + al((size_t)d.len * 4);
        PQ_TRY(pq_ws_reserve(ctx, total));
        unsigned char *w = (unsigned char *)ctx->ws;
        double *kx = (double *)w, *ky = (double *)(w + o_ky);
        int32_t *cnt = (int32_t *)(w + o_cnt);
        int p = 4;
        while ((1 << p) < d.n) p++;
        const int P = 1 << p, nthr = P / 16 > 64 ? P / 16 : 64;
        const size_t lds = (size_t)(P + P / 16) * 8 > 1024 ? (size_t)(P + P / 16) * 8 : 1024;
        PQ_HIP_TRY(hipFuncSetAttribute((const void *)rank_ic_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PQ_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)d.len * 4, ctx->stream));
        hipLaunchKernelGGL(rank_prep_kernel, dim3((unsigned)((d.len + 31) / 32), (unsigned)((d.n + 31) / 32)), dim3(256), 0, ctx->stream, factor,
                           fwd_return, d, kx, ky, (unsigned *)nullptr, cnt);
        hipLaunchKernelGGL(rank_ic_lds_kernel, dim3((unsigned)d.len), dim3(nthr), lds, ctx->stream, (const double *)kx, (const double *)ky,
                           (const int32_t *)cnt, d.n, p, ic);
        if (n_valid) PQ_HIP_TRY(hipMemcpyAsync(n_valid, cnt, (size_t)d.len * 4, hipMemcpyDeviceToDevice, ctx->stream));
        PQ_HIP_TRY(hipGetLastError());
        return PQ_OK;
    }
    // wider cross-sections: segmented radix sort of (key, symbol id) pairs
    PQ_REQUIRE(cells < (1ull << 32), "pq_factor_ic: rank IC needs n_series * len < 2^32");
    // workspace: kx, ky, sx, sy (f64) | ry (f64) | ids, ix, iy (u32) | offsets (u32) | counts (i32) | rocPRIM temp
    size_t tmp_bytes = 0;
    PQ_HIP_TRY(rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, (double *)nullptr, (double *)nullptr, (unsigned *)nullptr,
                                                   (unsigned *)nullptr, (unsigned)cells, (unsigned)d.len, (unsigned *)nullptr,
                                                   (unsigned *)nullptr, 0, 64, ctx->stream));
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
