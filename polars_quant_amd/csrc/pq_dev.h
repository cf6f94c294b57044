// pq_dev.h -- shared device/host plumbing for the gfx950 kernels.
//
// Execution shapes used by every kernel in this library:
//   SEQ  : one series (symbol) per lane, 64 series per wavefront, each lane walks its own series
//          left-to-right in exactly the reference's operation order (bit-exact f64).  Inputs are
//          pulled in register chunks of CH rows (independent loads issued ahead of the dependent
//          recurrence), outputs are pushed the same way.
//   ROW  : one (series, row) per thread, row index fastest => fully coalesced; used for everything
//          whose value is a pure function of a bounded look-back window (exact, order-free).
// A SEQ launch of one function has only n_series/64 wavefronts (79 for 5000 symbols), far too few to
// fill 256 CUs, so the library can also RECORD calls instead of launching them (pq_suite_*, suite.hip):
// recorded SEQ jobs of all functions run as ONE grid (blockIdx.y = job) that does fill the chip.
// All arithmetic is compiled with -ffp-contract=off; fma() appears only where the reference
// calls f64::mul_add.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/pq_hip.h"

struct Recorder; // suite.hip

struct pq_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    void *ws;        // scratch workspace (device)
    size_t ws_bytes;
    int64_t *d_flag; // 1 x int64 device scalar for reductions
    Recorder *rec;   // non-null while a suite is being recorded
};

void pq_set_error(const char *fmt, ...);
pq_status pq_ws_reserve(pq_ctx *ctx, size_t bytes);
// k-th scratch column ([n_series][stride] doubles).  While recording, every request returns a fresh
// column owned by the suite (recorded jobs run concurrently, so scratch cannot be shared).
double *pq_ws_col(pq_ctx *ctx, const pq_batch *b, int k);
pq_status pq_check(pq_ctx *ctx, const pq_batch *b);

#define PQ_HIP_TRY(expr)                                                                         \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            pq_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return PQ_ERR_HIP;                                                                   \
        }                                                                                        \
    } while (0)
#define PQ_TRY(expr)                      \
    do {                                  \
        pq_status s__ = (expr);           \
        if (s__ != PQ_OK) return s__;     \
    } while (0)
#define PQ_REQUIRE(cond, msg)             \
    do {                                  \
        if (!(cond)) {                    \
            pq_set_error("%s", msg);      \
            return PQ_ERR_ARG;            \
        }                                 \
    } while (0)

// ---------------------------------------------------------------- device helpers
#define PQ_SKIP_BITS 0x7FF80000534B4950ULL // "SKIP": a SEQ op with MASKED = true returns this to leave a row unwritten
__device__ __forceinline__ double pq_null() { return __longlong_as_double((long long)PQ_NULL_BITS); }
__device__ __forceinline__ bool pq_isnull(double x) {
    return (unsigned long long)__double_as_longlong(x) == PQ_NULL_BITS;
}
__device__ __forceinline__ double pq_skip() { return __longlong_as_double((long long)PQ_SKIP_BITS); }
__device__ __forceinline__ bool pq_isskip(double x) {
    return (unsigned long long)__double_as_longlong(x) == PQ_SKIP_BITS;
}

struct Dims {
    int64_t n, len, stride;
};
static inline Dims dims_of(const pq_batch *b) { return Dims{b->n_series, b->len, b->stride}; }

template <int N>
struct InCols {
    const double *p[N > 0 ? N : 1];
};
template <int N>
struct OutCols {
    double *p[N > 0 ? N : 1];
};

// Per-lane view of one series: random access to its own rows (used for lagged reads).
template <int NIN>
struct Row {
    const double *in[NIN > 0 ? NIN : 1]; // already offset to this series
    int64_t len;
};

constexpr int SEQ_BLOCK = 64; // one wavefront per workgroup: more workgroups to spread over 256 CUs
template <int NIN>
struct SeqChunk { // rows per register chunk: 64 B per lane and column, fewer for wide ops (VGPR budget)
    static constexpr int value = NIN <= 2 ? 8 : 4;
};
template <class Op, class = void>
struct IsMasked { static constexpr bool value = false; };
template <class Op>
struct IsMasked<Op, decltype((void)Op::MASKED)> { static constexpr bool value = Op::MASKED; };

template <class Op, class = void>
struct NTap { static constexpr int value = 0; };
template <class Op>
struct NTap<Op, decltype((void)Op::NTAP)> { static constexpr int value = Op::NTAP; };

// SEQ body for one lane.  Op contract:
//   static constexpr int NIN, NOUT;   [static constexpr bool MASKED = true;  outputs may be pq_skip()]
//   __device__ void init(const Row<NIN>& r);                       // once per series
//   __device__ void step(const Row<NIN>& r, int64_t t, const double (&x)[NIN], double (&y)[NOUT]);
// Lag taps (optional): an op that needs input column TAP_COL[i] at row t - lag_i declares
//   static constexpr int NTAP;  static constexpr int TAP_COL[NTAP];
//   __device__ void tap_lags(int64_t (&lag)[NTAP]) const;          // after init; lag <= 0 disables a tap
//   __device__ void step(r, t, x, const double (&tap)[NTAP], y);
// and gets the lagged values prefetched with the chunk instead of paying a dependent load per row.
template <class Op>
__device__ __forceinline__ void run_seq(Op &op, const double *const *inp, double *const *outp, const Dims &d, int64_t s) {
    constexpr int NIN = Op::NIN, NOUT = Op::NOUT, CH = SeqChunk<NIN>::value, NT = NTap<Op>::value;
    constexpr int NTA = NT > 0 ? NT : 1;
    constexpr bool MASKED = IsMasked<Op>::value;
    Row<NIN> r;
    r.len = d.len;
#pragma unroll
    for (int k = 0; k < NIN; k++) r.in[k] = inp[k] + s * d.stride;
    double *o[NOUT];
#pragma unroll
    for (int k = 0; k < NOUT; k++) o[k] = outp[k] + s * d.stride;
    op.init(r);
    int64_t lag[NTA];
    if constexpr (NT > 0) op.tap_lags(lag);
    const int64_t T = d.len;
    int64_t t0 = 0;
    double xb[NIN][CH], tb[NTA][CH];
    auto load_chunk = [&](int64_t base) {
#pragma unroll
        for (int k = 0; k < NIN; k++)
#pragma unroll
            for (int j = 0; j < CH; j++) xb[k][j] = r.in[k][base + j];
        if constexpr (NT > 0) {
#pragma unroll
            for (int i = 0; i < NT; i++)
#pragma unroll
                for (int j = 0; j < CH; j++) {
                    int64_t q = base + j - lag[i];
                    tb[i][j] = (lag[i] > 0 && q >= 0) ? r.in[Op::TAP_COL[i]][q] : 0.0;
                }
        }
    };
    if (T >= CH) load_chunk(0); // prologue: first chunk
    for (; t0 + CH <= T; t0 += CH) {
        double xc[NIN][CH], tc[NTA][CH];
#pragma unroll
        for (int k = 0; k < NIN; k++)
#pragma unroll
            for (int j = 0; j < CH; j++) xc[k][j] = xb[k][j];
        if constexpr (NT > 0) {
#pragma unroll
            for (int i = 0; i < NT; i++)
#pragma unroll
                for (int j = 0; j < CH; j++) tc[i][j] = tb[i][j];
        }
        if (t0 + 2 * CH <= T) load_chunk(t0 + CH); // next chunk's loads go out before this chunk's dependent recurrence
        double yb[NOUT][CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            double x[NIN], y[NOUT];
#pragma unroll
            for (int k = 0; k < NIN; k++) x[k] = xc[k][j];
            if constexpr (NT > 0) {
                double tp[NTA];
#pragma unroll
                for (int i = 0; i < NT; i++) tp[i] = tc[i][j];
                op.step(r, t0 + j, x, tp, y);
            } else {
                op.step(r, t0 + j, x, y);
            }
#pragma unroll
            for (int k = 0; k < NOUT; k++) yb[k][j] = y[k];
        }
#pragma unroll
        for (int k = 0; k < NOUT; k++)
#pragma unroll
            for (int j = 0; j < CH; j++)
                if (!MASKED || !pq_isskip(yb[k][j])) o[k][t0 + j] = yb[k][j];
    }
    for (; t0 < T; t0++) { // tail
        double x[NIN], y[NOUT];
#pragma unroll
        for (int k = 0; k < NIN; k++) x[k] = r.in[k][t0];
        if constexpr (NT > 0) {
            double tp[NTA];
#pragma unroll
            for (int i = 0; i < NT; i++) {
                int64_t q = t0 - lag[i];
                tp[i] = (lag[i] > 0 && q >= 0) ? r.in[Op::TAP_COL[i]][q] : 0.0;
            }
            op.step(r, t0, x, tp, y);
        } else {
            op.step(r, t0, x, y);
        }
#pragma unroll
        for (int k = 0; k < NOUT; k++)
            if (!MASKED || !pq_isskip(y[k])) o[k][t0] = y[k];
    }
}

template <class Op>
__global__ __launch_bounds__(SEQ_BLOCK) void seq_kernel(Op op, InCols<Op::NIN> in, OutCols<Op::NOUT> out, Dims d) {
    const int64_t s = (int64_t)blockIdx.x * SEQ_BLOCK + threadIdx.x;
    if (s >= d.n) return;
    run_seq(op, in.p, out.p, d, s);
}

// ---- recording hooks (implemented in suite.hip)
// a recordable SEQ op carries `static constexpr int SEQ_ID` = its switch case in the job grid (suite.hip)
pq_status rec_add_seq(pq_ctx *ctx, const pq_batch *b, int kind, const void *op, size_t op_bytes, const double *const *in,
                      int nin, double *const *out, int nout);
struct RowThunk { // type-erased ROW launch for replay
    void (*launch)(const void *blob, hipStream_t stream);
    unsigned char blob[1200];
    const void *reads[8];
    int n_reads;
    void *writes[64];
    int n_writes;
};
pq_status rec_add_row(pq_ctx *ctx, const RowThunk &t);
void rec_set_shared_out(pq_ctx *ctx, bool on); // jobs recorded while on may write disjoint rows of one column
// Records the enclosed calls into a suite and runs it once at finish(); a no-op inside an outer recording.
struct SuiteScope {
    pq_ctx *ctx;
    bool owner;
    pq_status status;
    SuiteScope(pq_ctx *ctx, const pq_batch *b);
    ~SuiteScope();
    pq_status finish();
};

template <class Op, class = void>
struct HasSeqId { static constexpr bool value = false; };
template <class Op>
struct HasSeqId<Op, decltype((void)Op::SEQ_ID)> { static constexpr bool value = true; };

template <class Op>
static inline pq_status launch_seq(pq_ctx *ctx, const pq_batch *b, const Op &op, const InCols<Op::NIN> &in,
                                   const OutCols<Op::NOUT> &out) {
    if (b->n_series == 0 || b->len == 0) return PQ_OK;
    if (ctx->rec) {
        if constexpr (HasSeqId<Op>::value) {
            static_assert(sizeof(Op) <= 512, "SEQ op too large for a job slot");
            return rec_add_seq(ctx, b, Op::SEQ_ID, &op, sizeof(Op), in.p, Op::NIN, out.p, Op::NOUT);
        } else {
            pq_set_error("this SEQ op cannot be recorded into a suite");
            return PQ_ERR_UNSUPPORTED;
        }
    }
    dim3 grid((unsigned)((b->n_series + SEQ_BLOCK - 1) / SEQ_BLOCK));
    hipLaunchKernelGGL(seq_kernel<Op>, grid, dim3(SEQ_BLOCK), 0, ctx->stream, op, in, out, dims_of(b));
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}

// ROW driver.  Op contract:
//   static constexpr int NIN, NOUT;  typedef OutT (double or int32_t)
//   __device__ void eval(const Row<NIN>& r, int64_t t, OutT (&y)[NOUT]);
constexpr int ROW_BLOCK = 256;
template <class Op, class OutT>
struct OutColsT {
    OutT *p[Op::NOUT];
};
template <class Op>
__global__ __launch_bounds__(ROW_BLOCK) void row_kernel(Op op, InCols<Op::NIN> in, OutColsT<Op, typename Op::OutT> out, Dims d) {
    const int64_t s = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * ROW_BLOCK + threadIdx.x;
    if (t >= d.len) return;
    Row<Op::NIN> r;
    r.len = d.len;
#pragma unroll
    for (int k = 0; k < Op::NIN; k++) r.in[k] = in.p[k] + s * d.stride;
    typename Op::OutT y[Op::NOUT];
    op.eval(r, t, y);
#pragma unroll
    for (int k = 0; k < Op::NOUT; k++) out.p[k][s * d.stride + t] = y[k];
}
template <class Op>
struct RowBlob {
    Op op;
    InCols<Op::NIN> in;
    OutColsT<Op, typename Op::OutT> out;
    pq_batch b;
};
template <class Op>
static void row_launch_blob(const void *blob, hipStream_t stream) {
    const RowBlob<Op> &rb = *reinterpret_cast<const RowBlob<Op> *>(blob);
    const pq_batch *b = &rb.b;
    for (int64_t s0 = 0; s0 < b->n_series; s0 += 65535) { // grid.y is limited to 65535: slice the series axis
        int64_t ns = b->n_series - s0 < 65535 ? b->n_series - s0 : 65535;
        InCols<Op::NIN> in2 = rb.in;
        OutColsT<Op, typename Op::OutT> out2 = rb.out;
        for (int k = 0; k < Op::NIN; k++) in2.p[k] += s0 * b->stride;
        for (int k = 0; k < Op::NOUT; k++) out2.p[k] += s0 * b->stride;
        dim3 grid((unsigned)((b->len + ROW_BLOCK - 1) / ROW_BLOCK), (unsigned)ns);
        Dims d{ns, b->len, b->stride};
        hipLaunchKernelGGL(row_kernel<Op>, grid, dim3(ROW_BLOCK), 0, stream, rb.op, in2, out2, d);
    }
}
template <class Op>
static inline pq_status launch_row(pq_ctx *ctx, const pq_batch *b, const Op &op, const InCols<Op::NIN> &in,
                                   const OutColsT<Op, typename Op::OutT> &out) {
    if (b->n_series == 0 || b->len == 0) return PQ_OK;
    RowBlob<Op> rb{op, in, out, *b};
    if (ctx->rec) {
        static_assert(sizeof(RowBlob<Op>) <= sizeof(RowThunk::blob), "ROW blob too large");
        RowThunk t;
        t.launch = &row_launch_blob<Op>;
        memcpy(t.blob, &rb, sizeof rb);
        t.n_reads = Op::NIN;
        for (int k = 0; k < Op::NIN; k++) t.reads[k] = in.p[k];
        t.n_writes = Op::NOUT;
        for (int k = 0; k < Op::NOUT; k++) t.writes[k] = out.p[k];
        return rec_add_row(ctx, t);
    }
    row_launch_blob<Op>(&rb, ctx->stream);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}
