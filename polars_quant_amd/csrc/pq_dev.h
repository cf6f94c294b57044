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
#include <stdlib.h>
#include <string.h>
#include "../../include/pq_hip.h"
#include "experiments.h"

struct Recorder; // suite.hip

struct pq_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    void *ws;        // scratch workspace (device)
    size_t ws_bytes;
    int64_t *d_flag; // 8 x int64 device scalars: [0] reductions, [4..6] statistics of the wave-per-symbol backtest
    Recorder *rec;   // non-null while a suite is being recorded
    // the suite being recorded covers a SMALL shard (suite.hip small_shard: few 64-series tiles, e.g. 625 symbols = one rank's share of
    // 5 000 on 8 GPUs): every job is then a handful of lone wavefronts bound by their own instruction stream, the chip is mostly idle,
    // and what shortens the step is the LENGTH of the longest job, not bytes -- the multi-output forms record their members as separate
    // jobs, MAVP its candidate periods in blocks of 16, MIDPRICE its row-parallel form, the Hilbert job its time-split form
    bool rec_small = false;
    bool chain_head = false; // set by a composite around the call that heads its dependency chain (pq_stochrsi_chain: its RSI): a small-shard
                             // recording then takes the call's fastest form even where that form does not pay for a call nobody waits for
    void *comm;      // ncclComm_t of pq_comm_init (comm.hip), or null
    int comm_rank, comm_world;
    // the communicator's own stream + one event pair per slot: pq_gather_summaries_begin / _end run the exchange of step k beside the
    // kernels of step k + 1 (comm.hip); the stream is created lazily by the first pq_gather_summaries_begin (comm_stream_make), destroyed by pq_comm_destroy
    hipStream_t comm_stream = nullptr;
    hipEvent_t comm_ev_in[2] = {nullptr, nullptr}, comm_ev_done[2] = {nullptr, nullptr};
    bool comm_pending[2] = {false, false};
    void *rg_ws = nullptr;       // workspace of the ragged -> regular re-housing (rg_reserve): packed input / output columns + the per-series lengths
    size_t rg_ws_bytes = 0;
    int64_t rg_calls = 0;        // launches that took the re-housed path (pq_ragged_rehouse_stats)
    int cus = 0;                 // compute units of the device (read once at pq_ctx_create)
    hipStream_t suite_aux[4] = {nullptr, nullptr, nullptr, nullptr}; // side streams of suite replays (suite.hip: one per chain, shared by every suite of the context)
    unsigned *wt_gate = nullptr; // [wt_gate_tiles] flags of the wave-per-symbol kernels' direct launches (ops_wt.h): tiles the gated general path redoes
    size_t wt_gate_tiles = 0;
};

void pq_set_error(const char *fmt, ...);
pq_status pq_ws_reserve(pq_ctx *ctx, size_t bytes);
// k-th scratch column ([n_series][stride] doubles).  While recording, every request returns a fresh
// column owned by the suite (recorded jobs run concurrently, so scratch cannot be shared).
double *pq_ws_col(pq_ctx *ctx, const pq_batch *b, int k);
// scratch column or PQ_ERR_NOMEM (while recording every request is a fresh hipMalloc that can fail)
#define PQ_WS_COL(var, ctx, b, k)                                                                  \
    double *var = pq_ws_col(ctx, b, k);                                                            \
    if (!var) { pq_set_error("out of device memory for a scratch column"); return PQ_ERR_NOMEM; }
pq_status pq_check(pq_ctx *ctx, const pq_batch *b);
pq_status ctx_gate(pq_ctx *ctx, size_t tiles, unsigned **gate); // runtime.hip: the context's tile flags of gated direct launches
void *rec_alloc_zero(pq_ctx *ctx, size_t bytes);                // suite.hip: zeroed device memory owned by the suite being recorded

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
    const int64_t *offs; // ragged batch: series s = rows [offs[s], offs[s + 1]) of the long column (len = the longest); else null
    const int64_t *lens = nullptr; // a re-housed ragged batch (rg_pack): the series' own row counts (<= len); the tiled body hands them to the op
};
static inline Dims dims_of(const pq_batch *b) { return Dims{b->n_series, b->len, b->stride, b->offsets}; }
__device__ __forceinline__ int64_t dims_base(const Dims &d, int64_t s) { return d.offs ? d.offs[s] : s * d.stride; }
__device__ __forceinline__ int64_t dims_len(const Dims &d, int64_t s) { return d.offs ? d.offs[s + 1] - d.offs[s] : d.len; }
// rows of one column of the batch (scratch sizing)
static inline size_t batch_rows(const pq_batch *b) { return b->offsets ? (size_t)b->stride : (size_t)b->n_series * (size_t)b->stride; }
#define PQ_NO_RAGGED(b, what)                                                                  \
    do {                                                                                       \
        if ((b)->offsets) { pq_set_error(what ": ragged batches are not supported"); return PQ_ERR_UNSUPPORTED; } \
    } while (0)

// Algorithmic column transfers a job is credited with (8 bytes per f64 column and row PER REFERENCE CALL): a plain
// op is one call (NIN + NOUT); multi-output forms declare / sum the calls they replace.
template <class Op, class = void>
struct AlgCols { static constexpr int value = Op::NIN + Op::NOUT; };
template <class Op>
struct AlgCols<Op, decltype((void)Op::ALG_COLS)> { static constexpr int value = Op::ALG_COLS; };
// Derived outputs: NDER further columns that are ELEMENT-WISE functions of the job's own output row (e.g. dcphase / sine / leadsine of the
// Hilbert phasor).  The storer wave -- idle between hand-offs -- computes them from the out tiles it already holds in registers and
// streams them out with the same 128-byte pieces: no extra LDS, no extra pass over memory, nothing added to the compute wave's
// dependency chain.  Op contract: static constexpr int NDER;  double *der[NDER] (device pointers, 16-byte aligned, the batch's pitch);
//   __device__ static void derive(const double (&y)[NOUT], double (&z)[NDER]).   Tiled body, per-tile storer (no pair mode), series of
// whole tiles (len % K == 0: the host entry falls back to a separate launch otherwise).
template <class Op, class = void>
struct NDer { static constexpr int value = 0; };
template <class Op>
struct NDer<Op, decltype((void)Op::NDER)> { static constexpr int value = Op::NDER; };
// Time-split jobs (small shards, fused.hip pq_ht_all): an op with `static constexpr bool TS_OK = true; double *chk[NOUT];` may be recorded
// as several jobs over row ranges of the same series -- each a plain job on columns offset to its first row (so the op sees a series that
// starts there), told to keep its first `skip` tiles to itself: the warm-up of a chunk that starts W rows early.  Of those, only the LAST
// tile leaves the workgroup, into the op's `chk` columns (same layout as the outputs): the hand-over check compares it with what the
// previous chunk wrote for the same rows.
template <class Op, class = void>
struct TsOk { static constexpr bool value = false; };
template <class Op>
struct TsOk<Op, decltype((void)Op::TS_OK)> { static constexpr bool value = Op::TS_OK; };
template <class Op, class = void>
struct HasFinish { static constexpr bool value = false; }; // void finish(double *const *outp, const Dims &, int64_t s): per-series epilogue
template <class Op>
struct HasFinish<Op, decltype((void)&Op::finish)> { static constexpr bool value = true; };

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

// Steady-state fast path (optional).  An op that declares
//   static constexpr bool HAS_FAST = true;
//   __device__ bool steady(int64_t t0) const;        // per lane: from row t0 on every row takes the op's steady-state path
//   __device__ void step_fast(int64_t t, const double (&x)[NIN], double (&y)[NOUT]);
// gets whole tiles walked by step_fast -- straight-line code, unrolled, no per-row warm-up / null branching -- whenever
// steady() holds on every lane of the wave and (unless FAST_NULL_OK) no input value of the tile is NULL.  step_fast must
// leave the op's state exactly as the general step would (same arithmetic in the same order), so the two can alternate.
template <class Op, class = void>
struct HasFast { static constexpr bool value = false; };
template <class Op>
struct HasFast<Op, decltype((void)Op::HAS_FAST)> { static constexpr bool value = Op::HAS_FAST; };
template <class Op, class = void>
struct FastNullOk { static constexpr bool value = false; }; // true: the op never looks at null flags (N-B family)
template <class Op>
struct FastNullOk<Op, decltype((void)Op::FAST_NULL_OK)> { static constexpr bool value = Op::FAST_NULL_OK; };
// An op may take the rows of the fast path in batches of N = FAST_UNROLL rows:
//   template <int N> __device__ void steps_fast(int64_t t, const double (&x)[N][NIN], double (&y)[N][NOUT]);
// which lets it issue all LDS traffic that depends only on the inputs (window pops) ahead of the dependent arithmetic and
// exposes the row-independent output arithmetic (divisions, square roots) of N rows to the scheduler at once.
template <class Op, class = void>
struct HasFastBatch { static constexpr bool value = false; };
template <class Op>
struct HasFastBatch<Op, decltype((void)Op::FAST_BATCH)> { static constexpr bool value = Op::FAST_BATCH; };
template <class Op, int N>
__device__ __forceinline__ void fast_rows(Op &op, int64_t t, const double (&x)[N][Op::NIN], double (&y)[N][Op::NOUT]) {
    if constexpr (HasFastBatch<Op>::value) op.template steps_fast<N>(t, x, y);
    else {
#pragma unroll
        for (int u = 0; u < N; u++) op.step_fast(t + u, x[u], y[u]);
    }
}
// keeps a speculatively computed value where it is: without it the compiler turns `cond ? expensive : other` into a branch
// around the expensive part (a division), which splits the straight-line fast path into basic blocks
__device__ __forceinline__ double pq_keep(double v) { asm volatile("" : "+v"(v)); return v; }
template <class Op, class = void>
struct FastUnroll { static constexpr int value = 2; };      // rows per unrolled fast-loop iteration
template <class Op>
struct FastUnroll<Op, decltype((void)Op::FAST_UNROLL)> { static constexpr int value = Op::FAST_UNROLL; };

// Scheduling traits of a recordable op (suite.hip orders and groups jobs by them; nothing here is tied to a SEQ_ID):
//   static constexpr int COST_NS = ...;        solo cost of one row of the tiled body in ns (measured on MI355X at 5000 x 2520,
//                                              scripts/exp_solo.py); or, where the cost depends on a parameter,
//   __host__ int cost_ns() const;              the same from the op's parameters
//   static constexpr bool HEAVY = true;        needs more than the light job kernel's 192 VGPRs: runs in the 256-VGPR kernel
template <class Op, class = void>
struct OpCostStatic { static constexpr int value = 60 + 40 * (Op::NIN + Op::NOUT); };
template <class Op>
struct OpCostStatic<Op, decltype((void)Op::COST_NS)> { static constexpr int value = Op::COST_NS; };
template <class Op, class = void>
struct OpCost { static int get(const Op &) { return OpCostStatic<Op>::value; } };
template <class Op>
struct OpCost<Op, decltype((void)&Op::cost_ns)> { static int get(const Op &op) { return op.cost_ns(); } };
template <class Op, class = void>
struct IsHeavy { static constexpr bool value = false; };
template <class Op>
struct IsHeavy<Op, decltype((void)Op::HEAVY)> { static constexpr bool value = Op::HEAVY; };

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
    const int64_t sbase = dims_base(d, s);
    r.len = dims_len(d, s);
#pragma unroll
    for (int k = 0; k < NIN; k++) r.in[k] = inp[k] + sbase;
    double *o[NOUT];
#pragma unroll
    for (int k = 0; k < NOUT; k++) o[k] = outp[k] + sbase;
    op.init(r);
    int64_t lag[NTA];
    if constexpr (NT > 0) op.tap_lags(lag);
    const int64_t T = r.len;
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
    if constexpr (HasFinish<Op>::value) op.finish(outp, d, s); // the lane re-reads only what it stored itself
}

// ------------------------------------------------------------------------------------------------
// SEQ body, LDS-staged (the fast path).  The per-lane gather above costs one L1 tag lookup per lane and
// 8-byte element: a 64-lane load touches 64 different cache lines and the kernel becomes bound by the
// texture-addresser / L1 rate, not by HBM.  Here the wavefront instead moves a whole tile
// [64 series][K rows] per column with coalesced 16-byte accesses (K*8 contiguous bytes per series; K = 16 for
// 1-in/1-out ops, 8 otherwise.  Measured: K = 32/16 moves ~25 % more bytes/s when the grid is bandwidth-bound, but
// the doubled LDS footprint costs occupancy and the full suite runs slower, so the smaller tiles are the default),
// transposes it through LDS (row pitch K*8+8 bytes: conflict-free for the per-lane b64 reads, 2-way conflicts on the
// cooperative 16-byte fills / pulls -- no pitch serves both patterns), and each lane then walks ITS OWN row of the tile in the reference's order.
// Outputs take the same route back.  The next tile's global loads are in flight (in registers) while
// the current tile is computed.  Rolling windows live in LDS rings ([slot][lane] layout), which also
// makes them null-proof: only valid values are pushed.
// One wavefront = one workgroup, so the barriers below only order this wave's own LDS traffic.
// One wavefront == one workgroup: cross-lane LDS hand-offs only need this wave's own LDS operations to have
// completed (the LDS queue is in order per wave).  __syncthreads() would also drain vmcnt, i.e. wait for the
// prefetched global loads and the output stores of the previous tile on every tile -- exactly the latency the
// prefetch is there to hide -- so the hand-off is an LDS-only wait that the compiler may not move memory
// operations across.
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct Ring { // per-lane circular buffer in LDS; slot k of lane l at base + (k*64 + l)*8
    double *base; // already offset by the lane
    int depth, pos;
    __device__ double get(int back) const { // value pushed `back` pushes ago (1 = newest); back <= depth
        int k = pos - back;
        if (k < 0) k += depth;
        return base[k * 64];
    }
    // eight window values at once: out[u] = value pushed (back0 + DIR*u) pushes ago.  The addresses are plain selects
    // (no branches), so the eight LDS reads go out back to back and are waited for once.  Backs outside [1, depth] read
    // an arbitrary valid slot; the caller masks them.
    template <int DIR>
    __device__ void get8(int back0, double (&out)[8]) const {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int k = pos - (back0 + DIR * u);
            k += (k < 0) ? depth : 0;
            k = ((unsigned)k < (unsigned)depth) ? k : 0;
            out[u] = base[k * 64];
        }
    }
    __device__ void push(double v) {
        base[pos * 64] = v;
        pos = (pos + 1 == depth) ? 0 : pos + 1;
    }
    // N swaps back to back: old[u] = value pushed `depth` pushes before v[u].  Only LDS traffic and integer address
    // arithmetic (LDS executes a wave's operations in order, so a depth below N is fine too); nothing waits on a result.
    template <int N>
    __device__ void swap_n(const double (&v)[N], double (&old)[N]) {
#pragma unroll
        for (int u = 0; u < N; u++) {
            old[u] = base[pos * 64];
            base[pos * 64] = v[u];
            pos = (pos + 1 == depth) ? 0 : pos + 1;
        }
    }
    // the values that the next N pushes will overwrite (oldest first); requires depth >= N
    template <int N>
    __device__ void peek_n(double (&old)[N]) const {
        int k = pos;
#pragma unroll
        for (int u = 0; u < N; u++) {
            old[u] = base[k * 64];
            k = (k + 1 == depth) ? 0 : k + 1;
        }
    }
    template <int N>
    __device__ void push_n(const double (&v)[N]) {
#pragma unroll
        for (int u = 0; u < N; u++) {
            base[pos * 64] = v[u];
            pos = (pos + 1 == depth) ? 0 : pos + 1;
        }
    }
    __device__ double swap(double v) { // store v, return the value pushed `depth` pushes ago
        double old = base[pos * 64];
        base[pos * 64] = v;
        pos = (pos + 1 == depth) ? 0 : pos + 1;
        return old;
    }
};
struct RingAlloc {
    double *next; // lane-offset base of the free region
    // n doubles shared by all lanes of the wave (returns the un-offset pointer); costs ceil(n/64) slots
    __device__ double *make_shared(int64_t n) {
        double *p = next - (threadIdx.x & 63);
        next += (size_t)((n + 63) / 64) * 64;
        return p;
    }
    __device__ Ring make(int64_t depth) {
        Ring r;
        r.base = next;
        r.depth = depth > 0 ? (int)depth : 1;
        r.pos = 0;
        next += (size_t)r.depth * 64;
        return r;
    }
};
// ops with rolling windows add:   int64_t ring_slots() const;   void init_lds(const Row<NIN>&, RingAlloc&);
//                                 void step_lds(int64_t t, const double (&x)[NIN], double (&y)[NOUT]);
template <class Op, class = void>
struct HasRings { static constexpr bool value = false; };
template <class Op>
struct HasRings<Op, decltype((void)&Op::ring_slots)> { static constexpr bool value = true; };

#ifndef PQ_K11
#define PQ_K11 16
#endif
#ifndef PQ_KXX
#define PQ_KXX 8
#endif
template <class Op, class = void>
struct TileK { static constexpr int value = (Op::NIN == 1 && Op::NOUT == 1) ? PQ_K11 : PQ_KXX; };
template <class Op>
struct TileK<Op, decltype((void)Op::TILE_K)> { static constexpr int value = Op::TILE_K; }; // an op may trade tile size for LDS
template <class Op>
struct SeqTile {
    static constexpr int K = TileK<Op>::value;   // rows per tile
    static constexpr bool DIRECT = IsMasked<Op>::value; // row-masked outputs: per-lane 8-byte stores, no output tiles
    static constexpr int NT = DIRECT ? Op::NIN : (Op::NIN > Op::NOUT ? Op::NIN : Op::NOUT);
    static constexpr int ROWB = K * 8 + 8;                                 // LDS row pitch in bytes
    static constexpr int TILE_BYTES = 64 * ROWB;
    static constexpr int BYTES = NT * TILE_BYTES;
};
template <class Op>
static inline size_t seq_lds_bytes(const Op &op) {
    size_t b = SeqTile<Op>::BYTES;
    if constexpr (HasRings<Op>::value) b += (size_t)op.ring_slots() * 512;
    return b;
}
constexpr size_t SEQ_LDS_LIMIT = 64 * 1024; // above this an op falls back to the gather body

// The LDS body runs in a 2-wavefront workgroup: wave 0 loads + computes, wave 1 only stores.  Reason: gfx950 counts loads
// and stores in ONE counter (vmcnt), loads retire in order but stores do not, so a wave with stores in flight cannot wait
// for a prefetched load without also waiting for every store it has issued (measured: a tile copy by one wave runs at
// ~1 TB/s, split across a loading and a storing wave at ~2 TB/s, scripts/ubench/).  With the stores in another wave,
// wave 0's counter holds loads only and the prefetch is waited for exactly; wave 1 never waits on vmcnt at all.
// Hand-off per tile: wave 0 finishes the out tile in LDS -> barrier A -> wave 1 pulls it into registers -> barrier B ->
// wave 1 issues the global stores while wave 0 already overwrites LDS with the next input tile.
constexpr int SEQ_LDS_BLOCK = 128;
// Output columns are written once and not read again by the same step: a non-temporal store streams them past L2 / MALL
// instead of allocating there, which keeps the (shared, re-read) input columns cached.  Tile-copy microbenchmark
// (scripts/ubench/tilecopy2): 2.9 -> 4.3 TB/s of stores.
typedef double pq_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store2(double *p, const double2 &v) {
    pq_d2v w = {v.x, v.y};
    PQ_HOOK_STORE2(w, reinterpret_cast<pq_d2v *>(p));
}
// UNAL: the same tiles for columns whose rows are only 8-byte aligned (an odd row pitch -- a dense odd len --, or columns that start 8
// bytes off).  The cooperative accesses move 8 bytes per lane: K lanes cover the K rows (K * 8 contiguous bytes, the same piece as in the
// aligned body) of one series, 64 / K series per instruction, twice the instructions for the same bytes; a lane's register pair
// [k][i] holds accesses 2i and 2i + 1.  No pair mode (its 128-byte pieces are assembled from 16-byte chunks).  Measured at 5 000 x
// 2 520 with a row pitch of 2 521: 5.8 ms per step (7.7 with non-temporal stores, see g_store) against 19.8 ms in the per-lane gather
// body these batches ran before (255 VGPRs, 92 of them spilled, one L1 tag lookup per lane and 8 bytes) and 3.9 ms at the 16-byte
// aligned pitch 2 528.  (A form with 16-byte accesses on the 8-byte aligned addresses -- legal in the queue's unaligned access mode --
// was built and measured the same 7.7 ms: the cost of an odd pitch is that every 64-byte piece straddles two cache lines.)
// LENS: the batch is a re-housed ragged one (launch_seq, rg_pack): every series is walked over d.len rows of its padded row, but the op
// is told the series' OWN length (init / init_lds read r.len for their short-series rules); rows beyond it are computed on padding
// and never leave the padded columns.
template <class Op, bool UNAL = false, bool LENS = false>
__device__ __forceinline__ void run_seq_lds(Op &op, const double *const *inp, double *const *outp, const Dims &d,
                                            int64_t tile_s0, unsigned char *lds, int skip = 0) {
    constexpr int NIN = Op::NIN, NOUT = Op::NOUT, K = SeqTile<Op>::K, ROWB = SeqTile<Op>::ROWB;
    constexpr int TB = SeqTile<Op>::TILE_BYTES;
    constexpr int CPL = UNAL ? K : K / 2; // lanes per series segment: 16-byte chunks (UNAL: 8-byte elements)
    constexpr int SPI = 64 / CPL;   // series covered by one wave-wide access
    constexpr int NI = K / 2;       // register pairs per lane and column tile = 16-byte accesses (UNAL: each pair holds two 8-byte accesses)
    constexpr int EB = UNAL ? 8 : 16;
    constexpr bool MASKED = SeqTile<Op>::DIRECT; // per-lane stores by wave 0 (row-masked outputs)
    // the store replica (an A/B build, experiments.h PQ_EXP_STOREONLY): the step's grids, addresses, piece sizes and store policy with the
    // compute wave gone -- no loads, no LDS traffic, no barriers; what the write pattern alone costs
    constexpr bool SO = PQ_EXP_STOREONLY_ON && !SeqTile<Op>::DIRECT && !HasFinish<Op>::value && !UNAL;
    static_assert(NTap<Op>::value == 0 || HasRings<Op>::value, "an op with lag taps needs a ring variant for the LDS body");
    const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6);
    const int64_t T = d.len, nt = T / K;
    const int64_t it_first = (TsOk<Op>::value && skip > 0) ? skip - 1 : 0; // the first tile that is handed to the storer (wave-uniform; 0 unless time-split)
    const int csym = lane / CPL, cchunk = lane % CPL;
    // Global addresses of the cooperative tile accesses: a wave-uniform 64-bit base (column + first series of the tile + first row of
    // the tile: scalar arithmetic) plus a 32-bit BYTE offset per lane -- the global_load / global_store form with a scalar base and a
    // 32-bit vector offset.  The offset of access i is this lane's part (one register) + a uniform step, with the dead series of the last
    // tile folded onto the last live one by a compare + select: four full-rate 32-bit instructions per 16-byte access, no 64-bit
    // multiply.  (Four 64-bit row offsets held across the row loop were the first values the allocator spilled under the 192-VGPR cap.)
    // The host launches the tiled body only when 64 * stride * 8 < 2^32 (seq_cols_aligned).
    const int64_t tile_left = d.n - 1 - tile_s0;                         // wave-uniform: index of the last live series within the tile
    const unsigned rel_max = tile_left < 63 ? (tile_left > 0 ? (unsigned)tile_left : 0u) : 63u;
    const int64_t tile_base = tile_s0 * d.stride;                        // wave-uniform
    const unsigned stride_b = (unsigned)d.stride * 8u;                   // wave-uniform
    const unsigned lane_part = (unsigned)csym * stride_b + (unsigned)cchunk * (unsigned)EB;
    auto live_i = [&](int i) -> bool { return (unsigned)(csym + i * SPI) <= rel_max; };           // the series of access i exists
    auto toff = [&](int i) -> unsigned {
        const unsigned o = lane_part + (unsigned)(i * SPI) * stride_b;
        return live_i(i) ? o : rel_max * stride_b + (unsigned)cchunk * (unsigned)EB;
    };
    auto at = [&](const double *col, int i, int64_t t0) -> const double * { // col, t0 wave-uniform
        return reinterpret_cast<const double *>(reinterpret_cast<const unsigned char *>(col + tile_base + t0) + toff(i));
    };
    auto at_w = [&](double *col, int i, int64_t t0) -> double * {
        return reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(col + tile_base + t0) + toff(i));
    };
    unsigned char *const co_base = lds + csym * ROWB + cchunk * EB; // this lane's 16-byte slot of a cooperative tile access; access i adds
    auto co_row = [&](int i) -> unsigned char * { return co_base + i * (SPI * ROWB); }; // a compile-time offset (the DS offset field)
    // register pair [k][i] <-> memory / LDS.  Aligned: the 16 bytes of access i.  UNAL: the doubles of accesses 2i (.x) and 2i + 1 (.y)
    auto g_load = [&](const double *col, int i, int64_t t0) -> double2 {
        if constexpr (UNAL) return make_double2(*at(col, 2 * i, t0), *at(col, 2 * i + 1, t0));
        else return PQ_HOOK_TILE_LOAD(at(col, i, t0), t0, i);
    };
    auto g_store = [&](double *col, int i, int64_t t0, const double2 &v) {
        if constexpr (UNAL) { // plain stores, not non-temporal ones: at an odd pitch every 64-byte piece of a tile straddles two cache
            // lines and writes both partially; left in L2 in the usual order the halves of a line meet there (the next tile's piece
            // brings the other one) more often than when they are marked evict-first: 7.07 against 8.37 ms per step at pitch 2 521
            if (live_i(2 * i)) *at_w(col, 2 * i, t0) = v.x;
            if (live_i(2 * i + 1)) *at_w(col, 2 * i + 1, t0) = v.y;
        } else if (live_i(i)) nt_store2(at_w(col, i, t0), v);
    };
    auto l_get = [&](int i, int koff) -> double2 { // (two b64 reads: LDS rows are only 8-byte aligned)
        if constexpr (UNAL) return make_double2(*reinterpret_cast<const double *>(co_row(2 * i) + koff), *reinterpret_cast<const double *>(co_row(2 * i + 1) + koff));
        else { const double *q = reinterpret_cast<const double *>(co_row(i) + koff); return make_double2(q[0], q[1]); }
    };
    auto l_put = [&](int i, int koff, const double2 &v) {
        if constexpr (UNAL) { *reinterpret_cast<double *>(co_row(2 * i) + koff) = v.x; *reinterpret_cast<double *>(co_row(2 * i + 1) + koff) = v.y; }
        else { double *q = reinterpret_cast<double *>(co_row(i) + koff); q[0] = v.x; q[1] = v.y; }
    };

    PQ_PROF_SIMD(wave);
    // hand-off of a finished out tile to the storer wave.  (Measured alternative: one-wave workgroups in which the compute
    // wave pulls the tile back and issues the stores itself, with the next tile's loads issued before them -- 8.2 vs 5.5 ms
    // per suite step: the wave's own stores hold up its vmcnt waits whatever the order.)
    auto hand_off = [&](int64_t) {
        if constexpr (MASKED) return;
        __builtin_amdgcn_s_barrier(); // A: out tile complete, wave 1 may read it
        __builtin_amdgcn_s_barrier(); // B: wave 1 holds the tile in registers, LDS is free again
    };
    // (`wave == 1` for the two-wave form, literally: with `wave >= 1` the compute branch learns that its wave index is 0 and the register
    //  allocation of the light job kernel shifts by two spilled registers under its 192 cap)
    if (wave == 1) { // -------------------------------------------------------------------------- storer
        if constexpr (!MASKED) {
            // Tunable: the storer can keep ACC consecutive out tiles in registers and issue their stores back to back (ACC * K * 8
            // contiguous bytes per series within a few cycles).  In a pure tile copy 64-byte pieces scattered over 64 series run
            // the write path at ~3.1 TB/s against ~3.7 (pairs) / ~4.3 TB/s (128-byte pieces and up), scripts/ubench/tilecopy3.hip.
            // (PQ_STORER_ACC, experiments.h: 1; 2 / 4 measured -1 % .. +4 % per suite step: the microbenchmark's gain does not carry over)
#ifndef PQ_STORER_PAIR
#define PQ_STORER_PAIR 1
#endif
#ifndef PQ_PAIR_CAP
#define PQ_PAIR_CAP 136
#endif
            if constexpr (K == 8 && PQ_STORER_PAIR && NDer<Op>::value == 0 && !UNAL) {
                // Pair mode: the storer re-maps its lanes to (series of a group of 8, one of the 8 chunks of TWO consecutive tiles):
                // lanes with chunk < 4 pull their 16 bytes out of the even tile, the others out of the odd tile into the same
                // registers, and one store instruction then writes 8 series x 128 contiguous bytes instead of 16 x 64 -- with a row
                // pitch that is a multiple of 128 B every piece is one full cache line (no partial-line write in L2): -8 % per suite
                // step at pitch 2528 (against 64-byte pieces at the dense pitch 2520; -4 % of it from the pitch alone, which also
                // aligns the 16-row tiles of the 1-in/1-out ops).  No extra LDS, the register count of two held tiles; columns
                // beyond the register cap go out per tile as before.
                constexpr int CAP = IsHeavy<Op>::value ? 208 : PQ_PAIR_CAP;
                constexpr int W0 = (CAP - NOUT * NI * 4) / (NI * 4);
                constexpr int W = (SO && PQ_EXP_SO_ALLPAIR_ON) ? NOUT : (W0 < 0 ? 0 : (W0 > NOUT ? NOUT : W0)), R = NOUT - W; // (replica variant: every column in 128-byte pieces)
                const int half = (lane >> 2) & 1, sub = lane >> 3;
                const unsigned char *pr_row = lds + sub * ROWB + (lane & 3) * 16;
                static_assert(!TsOk<Op>::value, "a time-split op stores per tile: its check tile and its first own tile go to different columns");
                for (int64_t it = 0; it < nt; it += 2) {
                    double2 w[W > 0 ? W : 1][8];
#pragma unroll
                    for (int a = 0; a < 2; a++) {
                        if (it + a < nt) {
                            double2 v[R > 0 ? R : 1][NI];
                            PQ_HOOK_STORER_BARRIER(); // A: out tile `it + a` is complete
                            lds_fence();
                            if (half == a) {
#pragma unroll
                                for (int k = 0; k < W; k++) {
                                    const int kk = k;
#pragma unroll
                                    for (int i = 0; i < 8; i++) {
                                        const double *q = reinterpret_cast<const double *>(pr_row + i * 8 * ROWB + kk * TB);
                                        w[k][i] = PQ_HOOK_STORER_PULL(q, kk, i);
                                    }
                                }
                            }
#pragma unroll
                            for (int k = 0; k < R; k++) {
                                const int kk = W + k;
#pragma unroll
                                for (int i = 0; i < NI; i++) {
                                    const double *q = reinterpret_cast<const double *>(co_row(i) + kk * TB);
                                    v[k][i] = PQ_HOOK_STORER_PULL(q, kk, i);
                                }
                            }
                            lds_fence();
                            PQ_HOOK_STORER_BARRIER(); // B: LDS may be overwritten
                            const int64_t t0 = (it + a) * K;
#pragma unroll
                            for (int k = 0; k < R; k++) {
                                const int kk = W + k;
#pragma unroll
                                for (int i = 0; i < NI; i++)
                                    if (live_i(i)) nt_store2(at_w(outp[kk], i, t0), v[k][i]);
                            }
                        }
                    }
                    const bool mine = it + 1 < nt || half == 0; // an odd tile count leaves the last tile alone
                    const unsigned pair_part = (unsigned)sub * stride_b + (unsigned)(lane & 7) * 16u; // series sub + 8 i, rows 2 (lane & 7) ..
#pragma unroll
                    for (int k = 0; k < W; k++) {
                        const int kk = k;
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            unsigned char *const pb = reinterpret_cast<unsigned char *>(outp[kk] + tile_base + it * K); // wave-uniform
                            if (mine && (unsigned)(i * 8 + sub) <= rel_max)
                                nt_store2(reinterpret_cast<double *>(pb + (pair_part + (unsigned)(i * 8) * stride_b)), w[k][i]);
                        }
                    }
                }
            } else {
            // (an op with derived columns takes this per-tile form: the 64-bit constants of derive() -- ~80 registers for an atan -- leave
            //  no room for the two held tiles of pair mode under the 192-VGPR cap; its columns go out in 64-byte pieces)
            constexpr int ACC = (NOUT * NI * 4 * PQ_STORER_ACC <= 136 && NDer<Op>::value == 0) ? PQ_STORER_ACC : 1; // registers of the (otherwise idle) storer wave
            static_assert(!TsOk<Op>::value || ACC == 1, "a time-split op stores tile by tile");
            for (int64_t it = it_first; it < nt; it += ACC) {
                double2 v[ACC][NOUT][NI];
#pragma unroll
                for (int a = 0; a < ACC; a++) {
                    if (it + a < nt) {
                        PQ_HOOK_STORER_BARRIER(); // A: out tile `it + a` is complete
                        lds_fence();
#pragma unroll
                        for (int k = 0; k < NOUT; k++)
#pragma unroll
                            for (int i = 0; i < NI; i++) v[a][k][i] = PQ_HOOK_STORER_PULL_G(l_get(i, k * TB), k, i);
                        lds_fence();
                        PQ_HOOK_STORER_BARRIER(); // B: LDS may be overwritten
                    }
                }
                const int64_t t0 = it * K;
                bool check_tile = false; // time-split: the last warm-up tile goes to the op's check columns, nothing is derived from it
                if constexpr (TsOk<Op>::value) check_tile = it < skip;
#pragma unroll
                for (int k = 0; k < NOUT; k++) {
                    double *oc = outp[k];
                    if constexpr (TsOk<Op>::value) oc = check_tile ? op.chk[k] : oc;
#pragma unroll
                    for (int i = 0; i < NI; i++)
#pragma unroll
                        for (int a = 0; a < ACC; a++)
                            if (it + a < nt) g_store(oc, i, t0 + a * K, v[a][k][i]);
                }
                if (check_tile) continue;
                if constexpr (NDer<Op>::value > 0) { // derived columns from the two rows per access this lane holds
                    constexpr int ND = NDer<Op>::value;
#pragma unroll
                    for (int i = 0; i < NI; i++) {
                        double ya[NOUT], yb[NOUT], za[ND], zb[ND];
#pragma unroll
                        for (int k = 0; k < NOUT; k++) { ya[k] = v[0][k][i].x; yb[k] = v[0][k][i].y; }
                        Op::derive(ya, za);
                        Op::derive(yb, zb);
#pragma unroll
                        for (int k = 0; k < ND; k++) g_store(op.der[k], i, t0, make_double2(za[k], zb[k]));
                    }
                }
            }
            }
            if constexpr (HasFinish<Op>::value) { // the epilogue of wave 0 reads what this wave stored
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every store acknowledged
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier(); // C
            }
        }
        return;
    }
    // -------------------------------------------------------------------------------- loader + compute
    if constexpr (SO) return; // the store replica: storer waves only
    const int64_t s = tile_s0 + lane;
    const bool live = s < d.n;
    const int64_t srow = live ? s : d.n - 1; // dead lanes shadow the last series (never stored)
    Row<NIN> r;
    if constexpr (LENS) r.len = d.lens[srow]; else r.len = d.len;
#pragma unroll
    for (int k = 0; k < NIN; k++) r.in[k] = inp[k] + srow * d.stride;
    if constexpr (HasRings<Op>::value) {
        RingAlloc ra{reinterpret_cast<double *>(lds + SeqTile<Op>::BYTES) + lane};
        op.init_lds(r, ra);
    } else {
        op.init(r);
    }
    unsigned char *my_row = lds + lane * ROWB;
    // wave 0 has no stores in flight (except for MASKED ops), so its prefetched loads can be waited for exactly
    constexpr int PF = (NIN * NI * 4 <= PQ_PF2_MAX) ? 2 : 1; // a second tile of prefetch costs NIN*NI*4 VGPRs
    double2 pre[PF][NIN][NI];
    auto prefetch = [&](double2 (&buf)[NIN][NI], int64_t t0) {
#pragma unroll
        for (int k = 0; k < NIN; k++)
#pragma unroll
            for (int i = 0; i < NI; i++) buf[k][i] = g_load(inp[k], i, t0);
    };
    auto do_tile = [&](double2 (&buf)[NIN][NI], int64_t it) {
        const int64_t t0 = it * K;
        PQ_PROF_T(c0);
        bool maybe_null = false; // some value moved by this lane has NULL's low word (exact enough: a false alarm only costs the general path)
#pragma unroll
        for (int k = 0; k < NIN; k++)
#pragma unroll
            for (int i = 0; i < NI; i++) { // two b64 stores: LDS rows are only 8-byte aligned
                l_put(i, k * TB, buf[k][i]);
                if constexpr (HasFast<Op>::value && !FastNullOk<Op>::value)
                    maybe_null |= (unsigned)__double2loint(buf[k][i].x) == (unsigned)(PQ_NULL_BITS & 0xffffffffu) ||
                                  (unsigned)__double2loint(buf[k][i].y) == (unsigned)(PQ_NULL_BITS & 0xffffffffu);
            }
        if (it + PF < nt) prefetch(buf, t0 + PF * K);
        lds_fence();
        PQ_PROF_T(c1);
        if constexpr (HasFast<Op>::value) {
            // steady state on the whole wave and a null-free tile: straight-line rows
            if (PQ_HOOK_FAST_OK(__builtin_amdgcn_ballot_w64(maybe_null || !op.steady(t0)) == 0)) {
                constexpr int FU = FastUnroll<Op>::value < K ? FastUnroll<Op>::value : K;
                static_assert(K % FU == 0, "FAST_UNROLL must divide the tile height");
#pragma unroll 1
                for (int j0 = 0; j0 < K; j0 += FU) {
                    double xs[FU][NIN], ys[FU][NOUT];
#pragma unroll
                    for (int u = 0; u < FU; u++)
#pragma unroll
                        for (int k = 0; k < NIN; k++) xs[u][k] = *reinterpret_cast<const double *>(my_row + k * TB + (j0 + u) * 8);
                    PQ_HOOK_FAST_ROWS(Op, FU, NOUT, op, t0 + j0, xs, ys);
#pragma unroll
                    for (int u = 0; u < FU; u++) {
                        if constexpr (MASKED) {
#pragma unroll
                            for (int k = 0; k < NOUT; k++)
                                if (live && !pq_isskip(ys[u][k])) outp[k][s * d.stride + t0 + j0 + u] = ys[u][k];
                        } else {
#pragma unroll
                            for (int k = 0; k < NOUT; k++) *reinterpret_cast<double *>(my_row + k * TB + (j0 + u) * 8) = ys[u][k];
                        }
                    }
                }
                lds_fence();
                PQ_PROF_T(f2);
                if (it >= it_first) hand_off(t0);
                PQ_PROF_T(f3);
                PQ_PROF_ADD(0, c1 - c0); PQ_PROF_ADD(1, f2 - c1); PQ_PROF_ADD(2, f3 - f2); PQ_PROF_ADD(3, 1);
                return;
            }
        }
        // one row at a time, NOT unrolled: every job of a suite grid runs different code, and K copies of each
        // op body would thrash the instruction cache; the next row's inputs are read from LDS ahead of the step
        double xn[NIN];
#pragma unroll
        for (int k = 0; k < NIN; k++) xn[k] = *reinterpret_cast<const double *>(my_row + k * TB);
#pragma unroll 1
        for (int j = 0; j < K; j++) {
            double x[NIN], y[NOUT];
#pragma unroll
            for (int k = 0; k < NIN; k++) x[k] = xn[k];
            if (j + 1 < K) {
#pragma unroll
                for (int k = 0; k < NIN; k++) xn[k] = *reinterpret_cast<const double *>(my_row + k * TB + (j + 1) * 8);
            }
            if constexpr (HasRings<Op>::value) op.step_lds(t0 + j, x, y);
            else op.step(r, t0 + j, x, y);
            if constexpr (MASKED) {
#pragma unroll
                for (int k = 0; k < NOUT; k++)
                    if (live && !pq_isskip(y[k])) outp[k][s * d.stride + t0 + j] = y[k];
            } else {
#pragma unroll
                for (int k = 0; k < NOUT; k++) *reinterpret_cast<double *>(my_row + k * TB + j * 8) = y[k];
            }
        }
        lds_fence();
        PQ_PROF_T(c2);
        if (it >= it_first) hand_off(t0);
        PQ_PROF_T(c3);
        PQ_PROF_ADD(0, c1 - c0); PQ_PROF_ADD(1, c2 - c1); PQ_PROF_ADD(2, c3 - c2); PQ_PROF_ADD(3, 1);
    };
#pragma unroll
    for (int f = 0; f < PF; f++)
        if (f < nt) prefetch(pre[f], (int64_t)f * K);
    for (int64_t it = 0; it < nt; it += PF) {
#pragma unroll
        for (int f = 0; f < PF; f++)
            if (it + f < nt) do_tile(pre[f], it + f);
    }
    // ragged tail: fewer than K rows left.  The series index and its row offset are formed again here (from the lane id as the
    // hardware counts it, which the compiler cannot tie to the value used above) rather than kept -- or spilled -- across the row loop.
    const int64_t s_tail = tile_s0 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (s_tail < d.n) {
        const int64_t row_tail = s_tail * d.stride;
        for (int64_t t = nt * K; t < T; t++) {
            double x[NIN], y[NOUT];
#pragma unroll
            for (int k = 0; k < NIN; k++) x[k] = inp[k][row_tail + t];
            if constexpr (HasRings<Op>::value) op.step_lds(t, x, y);
            else op.step(r, t, x, y);
#pragma unroll
            for (int k = 0; k < NOUT; k++)
                if (!MASKED || !pq_isskip(y[k])) outp[k][row_tail + t] = y[k];
            // (an op with derived columns is only launched on series of whole tiles -- its host entry checks len % K: evaluating
            // derive() here would add its temporaries to the compute wave's live state for a loop of at most K - 1 rows)
        }
    }
    if constexpr (HasFinish<Op>::value) {
        static_assert(!MASKED, "finish() is not supported for row-masked ops");
        __builtin_amdgcn_s_barrier(); // C: wave 1's stores have been acknowledged
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (live) op.finish(outp, d, s);
    }
}

// 16-byte accesses need 16-byte aligned rows
template <int NIN, int NOUT>
static inline bool seq_cols_aligned(const pq_batch *b, const double *const *in, double *const *out) {
    if (b->stride % 2 || b->offsets) return false; // (ragged series start at arbitrary rows: the per-lane body runs them)
    if (b->stride >= (1 << 22)) return false;     // the tiled body addresses a tile's 64 series with 32-bit byte offsets
    for (int k = 0; k < NIN; k++) if (reinterpret_cast<uintptr_t>(in[k]) % 16) return false;
    for (int k = 0; k < NOUT; k++) if (reinterpret_cast<uintptr_t>(out[k]) % 16) return false;
    return true;
}

// 0: 16-byte aligned rows; 1: rows only 8-byte aligned (the UNAL form of the tiled body); -1: neither (gather body)
template <int NIN, int NOUT>
static inline int seq_cols_tiling(const pq_batch *b, const double *const *in, double *const *out) {
    if (seq_cols_aligned<NIN, NOUT>(b, in, out)) return 0;
    if (b->offsets || b->stride >= (1 << 22)) return -1;
    for (int k = 0; k < NIN; k++) if (reinterpret_cast<uintptr_t>(in[k]) % 8) return -1;
    for (int k = 0; k < NOUT; k++) if (reinterpret_cast<uintptr_t>(out[k]) % 8) return -1;
    return 1;
}

template <class Op, bool LDS, bool UNAL = false, bool LENS = false>
#ifndef PQ_SEQ_MIN_WAVES
#define PQ_SEQ_MIN_WAVES 1 // analysis builds: 3 = compile every stand-alone op kernel under the light job kernel's register cap
#endif
PQ_HOOK_SEQ_KERNEL_ATTR __global__ __launch_bounds__(LDS ? SEQ_LDS_BLOCK : SEQ_BLOCK, PQ_SEQ_MIN_WAVES) void seq_kernel(Op op, InCols<Op::NIN> in, OutCols<Op::NOUT> out, Dims d,
                                                                                                                          unsigned *gate) {
    // gate (nullable): this launch stands behind a wave-per-symbol launch of the same function (ops_wt.h) and redoes only the
    // 64-symbol tiles that one has flagged (a symbol with a NULL / NaN input); it clears the flag it consumed
    if (gate) {
        const unsigned g = gate[blockIdx.x];
        if (!g) return;
        __syncthreads();
        if (threadIdx.x == 0) gate[blockIdx.x] = 0;
    }
    if constexpr (LDS) {
        extern __shared__ __attribute__((aligned(16))) unsigned char seq_lds[];
        run_seq_lds<Op, UNAL, LENS>(op, in.p, out.p, d, (int64_t)blockIdx.x * SEQ_BLOCK, seq_lds);
    } else {
        const int64_t s = (int64_t)blockIdx.x * SEQ_BLOCK + threadIdx.x;
        if (s >= d.n) return;
        run_seq(op, in.p, out.p, d, s);
    }
}

// ---- recording hooks (implemented in suite.hip)
// a recordable SEQ op carries `static constexpr int SEQ_ID` = its switch case in the job grid (suite.hip)
struct SeqTraits { // what the scheduler needs to know about a recorded job
    int kind;          // Op::SEQ_ID: the switch case of the job kernels
    int cost;          // estimated solo duration of the job, microseconds
    bool heavy;        // register-heavy kernel variant
    bool masked;       // writes only some rows of its output column(s): several such jobs may share a column
    size_t lds_bytes;  // 0 = run the gather body
    size_t tile_bytes;
    int alg_cols;      // f64 column transfers credited (per reference call)
    double summary_bytes_per_series; // extra algorithmic bytes per series (the backtest's summary row)
    const void *extra_reads[4];      // columns read outside the tile path (signal / benchmark columns): ordering hazards only
    bool unal = false;               // the tiled body in its 8-byte form (rows not 16-byte aligned)
    int tile_k = 0;                  // SeqTile<Op>::K
    int ts_len = 0, ts_skip = 0, ts_row0 = 0; // a time-split job (TsOk): rows of its range, leading tiles it keeps to itself, its first row
    double alg_frac = 1.0;           // share of the op's algorithmic bytes this job is credited with (a time-split job: its own rows)
};
template <class Op, class = void>
struct HasExtraReads { static constexpr bool value = false; };
template <class Op>
struct HasExtraReads<Op, decltype((void)&Op::extra_reads)> { static constexpr bool value = true; };
pq_status rec_add_seq(pq_ctx *ctx, const pq_batch *b, const SeqTraits &tr, const void *op, size_t op_bytes, const double *const *in,
                      int nin, double *const *out, int nout, void *const *extra_writes = nullptr); // extra_writes: 4 pointers or null
struct RowThunk { // type-erased ROW launch for replay
    void (*launch)(const void *blob, hipStream_t stream);
    int row_id;     // Op::ROW_ID if the suite may run it inside its fused ROW grid (row_jobs_kernel), else 0
    int long_launch = 0; // 1: takes a good part of a small-shard step (a wave-per-symbol form): the heads of the ROW chain that are short go first
    int blob_bytes; // sizeof(RowBlob<Op>)
    Dims dims;      // of the batch the call was made on
    unsigned char blob[1200];
    const void *reads[40];
    int n_reads;
    void *writes[64];
    int n_writes;
};
pq_status rec_add_row(pq_ctx *ctx, const RowThunk &t);
void rec_set_shared_out(pq_ctx *ctx, bool on); // jobs recorded while on may write disjoint rows of one column
// Records the enclosed calls into a suite and runs it once at finish(); a no-op inside an outer recording.
// PQ_FUSE_OK(ctx): record / launch a multi-output form as ONE job (shared input tiles: what pays on a full chip)
#define PQ_FUSE_OK(ctx) (!((ctx)->rec && (ctx)->rec_small))
struct SuiteScope {
    pq_ctx *ctx;
    bool owner;
    pq_status status;
    SuiteScope(pq_ctx *ctx, const pq_batch *b);
    ~SuiteScope();
    pq_status finish();
};

template <class Op, class = void>
struct IsLdsOnly { static constexpr bool value = false; };
template <class Op>
struct IsLdsOnly<Op, decltype((void)Op::LDS_ONLY)> { static constexpr bool value = Op::LDS_ONLY; };
// ---- ragged batches through the tiled bodies (round 5) -------------------------------------------------------------------------------
// The groups of `.over("symbol")` start at arbitrary rows of the long columns, which the tiled body cannot address; the per-lane gather
// body can, at one L1 tag lookup per lane and 8 bytes (and, recorded, 255 registers with 92 spilled).  Every function of this library
// is CAUSAL in time -- row t of a series depends on rows <= t of that series only -- so a ragged batch whose groups are of similar
// length is re-housed instead: rg_pack copies every group into a row of a regular [n][pitch] batch (pitch = pq_recommended_stride of
// the longest group, zero padding behind the group's rows), the tiled kernel of the SAME op walks it -- told every series' own length
// (Dims::lens), so its short-series rules see what the reference sees -- and rg_unpack copies the group's rows of every output back.
// Two streaming copies per column at the chip's copy rate against a walk at L1 rate; results bit-identical to the gather body
// (tests/test_ragged_gpu.py runs both).  Taken when the padded batch is at most 1.5 x the rows of the ragged one.
template <class Op, class = void>
struct RgGather { static constexpr bool value = false; }; // Op::RG_GATHER: compute-bound walks (SAR, STOCH, the Hilbert pipeline, OBV) are faster per-lane when called alone
template <class Op>
struct RgGather<Op, decltype((void)Op::RG_GATHER)> { static constexpr bool value = Op::RG_GATHER; };
// Op::DIRECT_LANE_MAX: for a DIRECT call (no recording) on a regular batch of at most that many series the per-lane form is the faster
// one -- a handful of lone workgroups are bound by their own dependency latency, and for compute-bound walks the tiled body's LDS rings,
// transposes and hand-offs lengthen exactly that path (STOCHF 0.54 against 1.35 ms at 5 000 x 2 520, the Hilbert functions 1.15 against
// 1.57 at any size).  Inside a suite, where the chip is full, the tiled forms win (section 4).
template <class Op, class = void>
struct DirectLaneMax { static constexpr int64_t value = 0; };
template <class Op>
struct DirectLaneMax<Op, decltype((void)Op::DIRECT_LANE_MAX)> { static constexpr int64_t value = Op::DIRECT_LANE_MAX; };
template <class Op>
static inline bool direct_lane(const pq_ctx *ctx, const pq_batch *b) {
    return DirectLaneMax<Op>::value > 0 && !ctx->rec && !b->offsets && b->n_series <= DirectLaneMax<Op>::value && !getenv("PQ_NO_DIRECT_LANE");
}
template <class Op>
struct IsPackable { static constexpr bool value = !IsMasked<Op>::value && !HasFinish<Op>::value && NDer<Op>::value == 0 && !RgGather<Op>::value; };
static inline bool rg_worth(const pq_batch *b) {
    if (!b->offsets || b->n_series < 16 || b->stride < 16384 || b->len <= 0) return false;
    const int64_t pitch = pq_recommended_stride(b->len);
    return pitch < (1 << 22) && (double)b->n_series * (double)pitch <= 1.5 * (double)b->stride;
}
pq_status rg_reserve(pq_ctx *ctx, size_t bytes);
// src: n_cols long columns of the ragged batch b -> dst: n_cols regular columns [n_series][pitch]; lens: [n_series] (device) receives the group lengths
pq_status rg_pack(pq_ctx *ctx, const pq_batch *b, int64_t pitch, const double *const *src, double *const *dst, int n_cols, int64_t *lens);
pq_status rg_unpack(pq_ctx *ctx, const pq_batch *b, int64_t pitch, const double *const *src, double *const *dst, int n_cols);

// can this op instance run in the LDS body on these columns?  (fused ops have no gather body: callers check first)
template <class Op>
static inline bool seq_can_lds(const pq_ctx *ctx, const pq_batch *b, const Op &op, const InCols<Op::NIN> &in, const OutCols<Op::NOUT> &out) {
    if (seq_lds_bytes(op) > SEQ_LDS_LIMIT) return false;
    if (direct_lane<Op>(ctx, b)) return false; // (an LDS-only fused op: its caller then runs the chain of basic per-lane kernels)
    if (b->offsets && !ctx->rec && IsPackable<Op>::value && rg_worth(b) && !getenv("PQ_NO_RG_PACK")) return true; // (launch_seq re-houses the batch)
    return seq_cols_tiling<Op::NIN, Op::NOUT>(b, in.p, out.p) >= 0;
}
template <class Op, class = void>
struct HasSeqId { static constexpr bool value = false; };
template <class Op>
struct HasSeqId<Op, decltype((void)Op::SEQ_ID)> { static constexpr bool value = true; };

template <class Op>
static inline pq_status launch_seq(pq_ctx *ctx, const pq_batch *b, const Op &op, const InCols<Op::NIN> &in,
                                   const OutCols<Op::NOUT> &out) {
    if (b->n_series == 0 || b->len == 0) return PQ_OK;
    size_t lds = seq_lds_bytes(op);
    if constexpr (IsPackable<Op>::value) {
        if (b->offsets && !ctx->rec && lds <= SEQ_LDS_LIMIT && rg_worth(b) && !getenv("PQ_NO_RG_PACK")) { // ragged -> regular, tiled kernel, back
            constexpr int NIN = Op::NIN, NOUT = Op::NOUT;
            const int64_t pitch = pq_recommended_stride(b->len);
            const size_t col = (size_t)b->n_series * (size_t)pitch * sizeof(double);
            PQ_TRY(rg_reserve(ctx, (NIN + NOUT) * col + (size_t)b->n_series * sizeof(int64_t)));
            double *cols[NIN + NOUT];
            for (int k = 0; k < NIN + NOUT; k++) cols[k] = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(ctx->rg_ws) + (size_t)k * col);
            int64_t *lens = reinterpret_cast<int64_t *>(reinterpret_cast<unsigned char *>(ctx->rg_ws) + (size_t)(NIN + NOUT) * col);
            PQ_TRY(rg_pack(ctx, b, pitch, in.p, cols, NIN, lens));
            ctx->rg_calls++;
            InCols<NIN> pin;
            OutCols<NOUT> pout;
            for (int k = 0; k < NIN; k++) pin.p[k] = cols[k];
            for (int k = 0; k < NOUT; k++) pout.p[k] = cols[NIN + k];
            Dims pd{b->n_series, b->len, pitch, nullptr, lens};
            dim3 grid((unsigned)((b->n_series + SEQ_BLOCK - 1) / SEQ_BLOCK));
            hipLaunchKernelGGL((seq_kernel<Op, true, false, true>), grid, dim3(SEQ_LDS_BLOCK), lds, ctx->stream, op, pin, pout, pd, (unsigned *)nullptr);
            PQ_HIP_TRY(hipGetLastError());
            return rg_unpack(ctx, b, pitch, pout.p, out.p, NOUT);
        }
    }
    const int tiling = seq_cols_tiling<Op::NIN, Op::NOUT>(b, in.p, out.p);
    bool use_lds = lds <= SEQ_LDS_LIMIT && tiling >= 0;
    // (the 8-byte form of the tiled body exists for the light job kernel only -- seq_jobs_kernel<3> -- so a recorded HEAVY op on 8-byte
    //  rows takes the gather class; no op is marked HEAVY at present)
    if (IsHeavy<Op>::value && tiling == 1 && ctx->rec && !IsLdsOnly<Op>::value) use_lds = false;
    if (!IsLdsOnly<Op>::value && direct_lane<Op>(ctx, b)) use_lds = false; // compute-bound walk called alone: the per-lane form
    static_assert(!(IsHeavy<Op>::value && IsLdsOnly<Op>::value), "a HEAVY op needs a gather body: it is what 8-byte-aligned rows run when recorded");
    if (IsLdsOnly<Op>::value && !use_lds) {
        pq_set_error("internal: fused op launched without checking seq_can_lds");
        return PQ_ERR_UNSUPPORTED;
    }
    if (ctx->rec) {
        if constexpr (HasSeqId<Op>::value) {
            static_assert(sizeof(Op) <= 1024, "SEQ op too large for a job slot");
            void *extra[4] = {nullptr, nullptr, nullptr, nullptr}; // columns the job writes besides `out` (hazard tracking of the recording)
            if constexpr (HasFinish<Op>::value) extra[0] = op.finish_writes();
            if constexpr (NDer<Op>::value > 0) {
                static_assert(NDer<Op>::value <= 3 && !HasFinish<Op>::value, "at most three derived columns, no epilogue");
                for (int k = 0; k < NDer<Op>::value; k++) extra[k] = op.der[k];
            }
            SeqTraits tr{Op::SEQ_ID, (int)((double)OpCost<Op>::get(op) * (double)b->len * 1e-3), IsHeavy<Op>::value, IsMasked<Op>::value,
                         use_lds ? lds : 0, (size_t)SeqTile<Op>::BYTES, AlgCols<Op>::value, HasFinish<Op>::value ? 64.0 : 0.0, {}};
            tr.unal = use_lds && tiling == 1;
            tr.tile_k = SeqTile<Op>::K;
            if constexpr (HasExtraReads<Op>::value) op.extra_reads(tr.extra_reads);
            return rec_add_seq(ctx, b, tr, &op, sizeof(Op), in.p, Op::NIN, out.p, Op::NOUT, extra);
        } else {
            pq_set_error("this SEQ op cannot be recorded into a suite");
            return PQ_ERR_UNSUPPORTED;
        }
    }
    dim3 grid((unsigned)((b->n_series + SEQ_BLOCK - 1) / SEQ_BLOCK));
    if (use_lds && tiling == 1) hipLaunchKernelGGL((seq_kernel<Op, true, true>), grid, dim3(SEQ_LDS_BLOCK), lds, ctx->stream, op, in, out, dims_of(b), (unsigned *)nullptr);
    else if (use_lds) hipLaunchKernelGGL((seq_kernel<Op, true>), grid, dim3(SEQ_LDS_BLOCK), lds, ctx->stream, op, in, out, dims_of(b), (unsigned *)nullptr);
    else hipLaunchKernelGGL((seq_kernel<Op, false>), grid, dim3(SEQ_BLOCK), 0, ctx->stream, op, in, out, dims_of(b), (unsigned *)nullptr);
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
    const int64_t sbase = dims_base(d, s);
    Row<Op::NIN> r;
    r.len = dims_len(d, s);
    if (t >= r.len) return;
#pragma unroll
    for (int k = 0; k < Op::NIN; k++) r.in[k] = in.p[k] + sbase;
    typename Op::OutT y[Op::NOUT];
    op.eval(r, t, y);
#pragma unroll
    for (int k = 0; k < Op::NOUT; k++) PQ_HOOK_ROW_STORE(y[k], &out.p[k][sbase + t]); // non-temporal: written once, not re-read
}
template <class Op, class = void> struct RowId { static constexpr int value = 0; };
template <class Op> struct RowId<Op, decltype((void)Op::ROW_ID)> { static constexpr int value = Op::ROW_ID; };
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
        if (!b->offsets) { // (a ragged slice keeps the column pointers: its offsets are absolute rows)
            for (int k = 0; k < Op::NIN; k++) in2.p[k] += s0 * b->stride;
            for (int k = 0; k < Op::NOUT; k++) out2.p[k] += s0 * b->stride;
        }
        dim3 grid((unsigned)((b->len + ROW_BLOCK - 1) / ROW_BLOCK), (unsigned)ns);
        Dims d{ns, b->len, b->stride, b->offsets ? b->offsets + s0 : nullptr};
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
        t.row_id = RowId<Op>::value;
        t.blob_bytes = (int)sizeof rb;
        t.dims = dims_of(b);
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
