// suite.hip -- record many pq_* calls, replay them as a few chip-filling grids.
//
// Why: a SEQ launch of one indicator has n_series/64 wavefronts (79 for 5000 symbols) -- 256 CUs cannot be
// filled, each wave is latency-bound on its own loads and the launch runs at a few hundred GB/s.  The symbols
// are the only parallel axis of one indicator (the time axis is a strict recurrence), but a DataFrame query
// asks for MANY indicators at once.  pq_suite_begin()/pq_suite_end() therefore turn the enclosed pq_* calls into
// a dependency-ordered list of phases; all SEQ jobs of a phase execute as ONE grid (blockIdx.y = job,
// blockIdx.x = 64-symbol tile), ROW launches replay as recorded.  pq_suite_run() replays with no host work
// besides the launches.  Composite functions (MAVP = one job per candidate period) use the same machinery
// internally through SuiteScope.
#include "suite_jobs.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <set>
#include <vector>

struct GridStat { // one SEQ grid launch site: algorithmic bytes + optional HIP-event timing
    double alg_bytes = 0;
    int n_jobs = 0;
    unsigned lds = 0;
    std::vector<hipEvent_t> ev; // start/stop pairs, one pair per timed run
    int runs = 0;
};
// How the SEQ jobs of a phase are grouped into grids.  Facts that shape this (measured, scripts/wg_residency.py and
// scripts/ubench/nstreams.hip):
//  * every workgroup walks its 64 series for the whole step, so the step ends when the last-started long workgroup ends:
//    the longest jobs must start at t = 0 and the tail should consist of short jobs;
//  * a launch has ONE dynamic-LDS size (every workgroup is charged the largest need in its grid) and one register
//    allocation, and a 40 KB workgroup loses every race for freed LDS against 10 KB ones;
//  * HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): the caller's stream + 3 side streams run
//    concurrently, any further stream is serialized behind another one.
// Hence four grids = four chains:
//   LONG   (caller's stream)  the light-register jobs in order of decreasing cost, as many as fill the chip at t = 0, then
//                             the "fat" ones (more LDS than THIN_LDS_MAX) whatever their cost
//   SHORT  the remaining light-register jobs (all thin), longest first: they fill the chip as LONG workgroups retire;
//          the lighter ROW launches follow on the same stream
//   HEAVY  the register-heavy jobs (seq_jobs_kernel<1>, 2 waves/SIMD) + the per-lane backtest scan
//   GATHER gather-body fallbacks (very long windows / unaligned columns), and the ROW launches
enum { CLS_LONG = 0, CLS_SHORT = 1, CLS_HEAVY = 2, CLS_GATHER = 3, NCLS = 4, NCHAIN = 4, ROW_CHAIN = 3 };
static const int k_chain_order[NCHAIN] = {CLS_HEAVY, CLS_LONG, CLS_SHORT, CLS_GATHER}; // enqueue order: hungriest first
static const int k_variant[NCLS] = {0, 0, 1, 2};                                       // seq_jobs_kernel<V> of each class
// "thin" = six workgroups fit a CU's LDS; the LONG grid places half a workgroup less per CU at t = 0 than LDS admits of its widest
// job.  Both follow from the device's LDS size and the recorded jobs (at 5 000 x 2 520 on MI355X: 26 KB and 3.5, the values round
// 2 had tuned as literals).
static unsigned thin_lds_max(const hipDeviceProp_t &prop) { return (unsigned)(prop.maxSharedMemoryPerMultiProcessor / 6 / 1024 * 1024); }
struct Phase {
    GridStat gs[NCLS];
    GridStat gs_row;           // the chain of ROW launches (timed as one unit)
    std::vector<SeqJob> seq;   // sorted at finalize by class, longest jobs first inside a class
    std::vector<RowThunk> rows;
    std::vector<char> row_late; // 0: on the ROW chain; 1 + c: behind the SEQ grid(s) of chain c
    std::set<const void *> rd, wr; // every column the phase's calls read / write (hazard bookkeeping of the recording)
    std::vector<char> row_fused; // the launch runs inside the fused ROW grid of its chain position
    struct RowJobDev *d_rows[NCHAIN + 1] = {}; // fused ROW grid per position (index = row_late value)
    int n_rows[NCHAIN + 1] = {};
    SeqJob *d_seq = nullptr;
    unsigned long long *d_dbg = nullptr; // PQ_SUITE_DEBUG: [job][first start, last end] device timestamps
    unsigned long long *d_wg = nullptr;  // PQ_SUITE_DEBUG=2: [job][tile][start, end, hw id] of every workgroup
    unsigned wg_tiles = 0;
    int first[NCLS + 1] = {}; // job index range of each class
    unsigned lds[NCLS] = {};
    // small-shard schedule (Recorder::small): what chain c of this phase waits for -- (phase, chain) pairs of earlier phases whose launches
    // wrote a column it touches or read a column it writes -- and the event recorded behind its launches
    // (slot NCHAIN = the HEADS of the phase's ROW chain: the ROW launches that feed later phases are issued first and get an event of
    //  their own, so that a link waiting for the fast-k column does not wait for the patterns and the backtest behind it)
    // (slot NCHAIN: the SHORT heads -- plain ROW kernels; slot NCHAIN + 1: every head, i.e. the long launches among them too -- a wave-per-
    //  symbol RSI in front of STOCHRSI: STOCH's averages wait for the fast-k kernel, not for it)
    std::vector<std::pair<int, int>> deps[NCHAIN][2]; // [chain][0: its job grids, 1: its ROW launches] -- waited for right in front of each part
    hipEvent_t ev_done[NCHAIN + 2] = {};
    int n_heads = 0, n_heads_short = 0; // leading ROW launches on the phase's ROW chain that feed a later phase; the short ones of them come first
    bool grid_first = false;            // later phases: the job grid is issued before the ROW launches of the chain (small_deps)
    bool work[NCHAIN] = {};
    int chain_of[NCLS] = {0, 1, 2, 3}; // small-shard schedule: the chain (stream) the grid of class c runs on in this phase
    int row_chain = ROW_CHAIN;         // ... and the chain of its ROW launches
};
struct Recorder {
    pq_batch b;
    hipStream_t aux[NCHAIN] = {};   // chains 1.. (chain 0 is the caller's stream): the context's side streams this suite has used (pq_ctx::suite_aux)
    hipEvent_t ev_fork = nullptr, ev_join[NCHAIN] = {}, ev_tail = nullptr;
    std::vector<Phase> phases;
    std::map<const void *, int> writer_phase, reader_phase;
    std::vector<void *> scratch;
    bool shared_out = false;
    bool small = false; // the recording covers a small shard (small_shard): chains ordered by data dependencies instead of phase barriers
    std::set<const void *> feeds_seq; // small: columns that a sequential job of a LATER phase reads, directly or through ROW launches
    bool grids_first = false;         // small: on the chain of the links, a phase's job grid is issued before its ROW launches (small_deps)
    bool timing = false;
    static constexpr int MAX_TIMED_RUNS = 64;
};
struct pq_suite {
    Recorder rec;
};

// the tiled body of one job of the light kernel; an op that may be split in time (pq_dev.h TsOk) walks the row range of its job
template <class OP, bool TS = TsOk<OP>::value>
struct TiledJob {
    __device__ static __forceinline__ void run(OP &op, const SeqJob &job, const Dims &d, int64_t s0, unsigned char *lds) { run_seq_lds<OP, false>(op, job.in, job.out, d, s0, lds); }
};
template <class OP>
struct TiledJob<OP, true> {
    __device__ static __forceinline__ void run(OP &op, const SeqJob &job, const Dims &d, int64_t s0, unsigned char *lds) {
        Dims dj = d;
        if (job.ts_len) dj.len = job.ts_len;
        const double *inj[OP::NIN];
        double *outj[OP::NOUT];
#pragma unroll
        for (int k = 0; k < OP::NIN; k++) inj[k] = job.in[k] + job.ts_row0;
#pragma unroll
        for (int k = 0; k < OP::NOUT; k++) outj[k] = job.out[k] + job.ts_row0;
        op.ts_shift(job.ts_row0);
        run_seq_lds<OP, false>(op, inj, outj, dj, s0, lds, job.ts_skip);
    }
};
template <int V>
__device__ __forceinline__ void seq_jobs_body(const SeqJob *jobs, Dims d, unsigned long long *dbg, unsigned long long *wg) {
    extern __shared__ __align__(16) unsigned char jobs_lds[];
    const SeqJob &job = jobs[blockIdx.y];
    if (dbg && threadIdx.x == 0) atomicMin(&dbg[2 * blockIdx.y], wall_clock64()); // PQ_SUITE_DEBUG: first start / last end per job
    if (wg && threadIdx.x == 0) {
        unsigned long long *q = wg + 3 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        q[0] = wall_clock64();
        q[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4); // XCC_ID, HW_ID
    }
    const int64_t s0 = (int64_t)blockIdx.x * SEQ_BLOCK;
    const int64_t s = s0 + threadIdx.x;
    if (s0 >= d.n) return; // grid.x is padded to a multiple of 8 (see pq_suite_run)
    // long jobs are the critical path of the step: their waves win the issue arbitration against short jobs on the same SIMD
    if (job.prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (job.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (job.prio == 1) __builtin_amdgcn_s_setprio(1);
#define X(OP)                                                                                                        \
    case OP::SEQ_ID: {                                                                                               \
        OP op;                                                                                                       \
        __builtin_memcpy(&op, job.op, sizeof(OP));                                                                   \
        if constexpr (V == 2) run_seq(op, job.in, job.out, d, s);                                                    \
        else if constexpr (V == 0) TiledJob<OP>::run(op, job, d, s0, jobs_lds);                                      \
        else                                                                                                         \
            run_seq_lds<OP, V == 3>(op, job.in, job.out, d, s0, jobs_lds); \
    } break;
    // the two lists must agree with the ops' HEAVY trait (which is what the class assignment in suite_finalize looks at)
#define XL(OP) static_assert(!IsHeavy<OP>::value, "light list holds an op marked HEAVY"); X(OP)
#define XH(OP) static_assert(IsHeavy<OP>::value, "heavy list holds an op not marked HEAVY"); X(OP)
    if constexpr (V == 2) {
        if (s >= d.n) return; // gather bodies are per-lane work
        switch (job.kind) {   // wave-uniform
            SEQ_OPS_LIGHT(X)
            SEQ_OPS_HEAVY(X)
        default: break;
        }
    } else if constexpr (V == 1) {
        switch (job.kind) {
            SEQ_OPS_HEAVY(XH)
        case SEQ_ID_BACKTEST: { // one series per lane of wave 0, no tiles
            BtArgs a;
            __builtin_memcpy(&a, job.op, sizeof(BtArgs));
            if (threadIdx.x < SEQ_BLOCK && s < d.n) backtest_body<false, false>(a, d, s);
        } break;
        case SEQ_ID_BACKTEST + 1: {
            BtArgs a;
            __builtin_memcpy(&a, job.op, sizeof(BtArgs));
            if (threadIdx.x < SEQ_BLOCK && s < d.n) backtest_body<true, false>(a, d, s);
        } break;
        default: break;
        }
    } else {
        switch (job.kind) {
            SEQ_OPS_LIGHT(XL)
        default: break;
        }
    }
#undef XL
#undef XH
#undef X
    if (dbg && threadIdx.x == 0) atomicMax(&dbg[2 * blockIdx.y + 1], wall_clock64());
    if (wg && threadIdx.x == 0) wg[3 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 1] = wall_clock64();
}
// 0: light ops, tiled; 1: heavy ops, tiled; 2: gather bodies; 3: light ops, tiled, rows only 8-byte aligned (UNAL)
template <int V>
__global__ __launch_bounds__(V == 2 ? SEQ_BLOCK : SEQ_LDS_BLOCK, V == 0 ? PQ_LB0 : V == 2 ? PQ_LB2 : 2) void seq_jobs_kernel(const SeqJob *jobs, Dims d, unsigned long long *dbg, unsigned long long *wg) {
    seq_jobs_body<V>(jobs, d, dbg, wg);
}
#if PQ_NV0 > 0
template <>
__global__ __attribute__((amdgpu_num_vgpr(PQ_NV0))) __launch_bounds__(SEQ_LDS_BLOCK, PQ_LB0) void seq_jobs_kernel<0>(const SeqJob *jobs, Dims d, unsigned long long *dbg, unsigned long long *wg) {
    seq_jobs_body<0>(jobs, d, dbg, wg);
}
#endif

// A recording covers a SMALL shard when one job is at most cus / 12 workgroups (21 tiles = 1 344 series on MI355X): with the ~45 jobs of
// an un-fused indicator suite that is at most ~4 workgroups per CU, i.e. about one compute wave per SIMD -- every wave then runs at its
// own instruction rate and the step ends with its longest job.  PQ_SMALL_SHARD_TILES=<n> moves the bound (0: never small).
static bool small_shard(const pq_ctx *ctx, const pq_batch *b) {
    if (b->offsets) return false;
    int64_t bound = (ctx->cus > 0 ? ctx->cus : 256) / 12; // (measured: the small-shard plan wins at 1 250 symbols, 1.78 against 1.93 ms, and loses at 1 875, 2.18 against 1.99)
    if (const char *e = getenv("PQ_SMALL_SHARD_TILES")) bound = atoll(e);
    return (b->n_series + SEQ_BLOCK - 1) / SEQ_BLOCK <= bound;
}
static int phase_for(Recorder &r, const void *const *reads, int nr, void *const *writes, int nw) {
    int ph = 0;
    for (int i = 0; i < nr; i++) {
        auto it = r.writer_phase.find(reads[i]);
        if (it != r.writer_phase.end()) ph = std::max(ph, it->second + 1);
    }
    for (int i = 0; i < nw; i++) {
        if (!writes[i]) continue;
        auto it = r.writer_phase.find(writes[i]);
        if (it != r.writer_phase.end()) ph = std::max(ph, r.shared_out ? it->second : it->second + 1);
        auto ir = r.reader_phase.find(writes[i]);
        if (ir != r.reader_phase.end()) ph = std::max(ph, ir->second + 1);
    }
    if ((int)r.phases.size() <= ph) r.phases.resize(ph + 1);
    for (int i = 0; i < nr; i++) {
        int &rp = r.reader_phase[reads[i]];
        rp = std::max(rp, ph);
        r.phases[ph].rd.insert(reads[i]);
    }
    for (int i = 0; i < nw; i++)
        if (writes[i]) {
            r.phases[ph].wr.insert(writes[i]);
            auto it = r.writer_phase.find(writes[i]);
            r.writer_phase[writes[i]] = (it == r.writer_phase.end()) ? ph : std::max(it->second, ph);
        }
    return ph;
}

static pq_status same_batch(Recorder &r, const pq_batch *b) {
    if (b->n_series != r.b.n_series || b->len != r.b.len || b->stride != r.b.stride || b->offsets != r.b.offsets) {
        pq_set_error("every call recorded into one suite must use the same batch shape");
        return PQ_ERR_ARG;
    }
    return PQ_OK;
}

pq_status rec_add_seq(pq_ctx *ctx, const pq_batch *b, const SeqTraits &tr, const void *op, size_t op_bytes, const double *const *in,
                      int nin, double *const *out, int nout, void *const *extra_writes) {
    Recorder &r = *ctx->rec;
    PQ_TRY(same_batch(r, b));
    if (nin > 6 || nout > 8) { pq_set_error("internal: SEQ job has too many columns"); return PQ_ERR_UNSUPPORTED; }
    SeqJob j;
    memset(&j, 0, sizeof j);
    j.kind = tr.kind; j.nin = nin; j.nout = nout; j.cost = tr.cost; j.heavy = tr.heavy; j.masked = tr.masked;
    j.lds_bytes = (unsigned)tr.lds_bytes; j.tile_bytes = (unsigned)tr.tile_bytes;
    j.unal = tr.unal ? 1 : 0;
    j.tile_k = tr.tile_k;
    j.alg_cols = tr.alg_cols > 0 ? tr.alg_cols : nin + nout;
    j.ts_len = tr.ts_len; j.ts_skip = tr.ts_skip; j.ts_row0 = tr.ts_row0; j.alg_frac = (float)tr.alg_frac;
    j.summary_bytes = tr.summary_bytes_per_series;
    for (int k = 0; k < nin; k++) j.in[k] = in[k];
    for (int k = 0; k < nout; k++) j.out[k] = out[k];
    memcpy(j.op, op, op_bytes);
    void *writes[12];
    for (int k = 0; k < nout; k++) writes[k] = out[k];
    for (int k = 0; k < 4; k++) writes[nout + k] = extra_writes ? extra_writes[k] : nullptr; // e.g. a summary table, derived columns: hazard tracking only
    for (int k = 0; k < 4; k++) { j.xr[k] = tr.extra_reads[k]; j.xw[k] = extra_writes ? extra_writes[k] : nullptr; }
    const void *reads[10];
    int nr = 0;
    for (int k = 0; k < nin; k++) reads[nr++] = in[k];
    for (int k = 0; k < 4; k++) if (tr.extra_reads[k]) reads[nr++] = tr.extra_reads[k];
    int ph = phase_for(r, reads, nr, writes, nout + 4);
    r.phases[ph].seq.push_back(j);
    return PQ_OK;
}
pq_status rec_add_backtest(pq_ctx *ctx, const pq_batch *b, int kind, const BtArgs &a) {
    Recorder &r = *ctx->rec;
    PQ_TRY(same_batch(r, b));
    SeqJob j;
    memset(&j, 0, sizeof j);
    j.kind = kind; j.heavy = 1;
    j.cost = (int)(650.0 * (double)b->len * 1e-3); // per-lane scan with 8-byte accesses: ~650 ns per row solo
    memcpy(j.op, &a, sizeof a);
    const void *reads[4] = {a.price, a.buy, a.sell, a.bench};
    void *writes[4] = {a.position, a.cash, a.equity, a.summary};
    const void *rd[4]; int nr = 0;
    for (int i = 0; i < 4; i++) if (reads[i]) rd[nr++] = reads[i];
    for (int i = 0; i < 4; i++) { j.xr[i] = reads[i]; j.xw[i] = writes[i]; }
    int ph = phase_for(r, rd, nr, writes, 4);
    r.phases[ph].seq.push_back(j);
    return PQ_OK;
}
pq_status rec_add_row(pq_ctx *ctx, const RowThunk &t) {
    Recorder &r = *ctx->rec;
    int ph = phase_for(r, t.reads, t.n_reads, t.writes, t.n_writes);
    r.phases[ph].rows.push_back(t);
    return PQ_OK;
}
void rec_set_shared_out(pq_ctx *ctx, bool on) {
    if (ctx->rec) ctx->rec->shared_out = on;
}

// ---- fused ROW grid ------------------------------------------------------------------------------------------------------------
// The window-bounded ROW ops of a phase (price transforms, BOP, TRANGE, MOM / ROC x 4, AROON, WILLR, HT_TRENDLINE / TRENDMODE) read
// the same few OHLC columns: as separate launches each streams its inputs from HBM again (32 column reads for 4 distinct columns)
// and the chain of small launches cannot fill the slots the SEQ grids leave.  One grid instead: a block takes 256 rows of one series
// through EVERY recorded ROW job, so the inputs come from HBM once and from L1/L2 afterwards.  Same eval(), same stores: identical
// results.  Ops without a ROW_ID (the 61-pattern kernel, signal rules...) keep their own launches.
constexpr int ROW_FUSE_BLOB = 176;
struct RowJobDev { int kind; int pad; unsigned char blob[ROW_FUSE_BLOB]; };
#define ROW_OPS(X)                                                                                                             \
    X(PriceOp<0>) X(PriceOp<1>) X(PriceOp<2>) X(PriceOp<3>) X(TrangeOp) X(TrendlineOp) X(TrendmodeOp) X(LagOp<0>) X(LagOp<1>) \
    X(LagOp<2>) X(LagOp<3>) X(LagOp<4>) X(BopOp) X(AroonOp<0>) X(AroonOp<1>) X(AroonOp<2>) X(WillrOp) X(MidpriceRowOp)
__global__ __launch_bounds__(ROW_BLOCK) void row_jobs_kernel(const RowJobDev *jobs, int n_jobs, Dims d, int64_t s_base) {
    const int64_t s = s_base + blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * ROW_BLOCK + threadIdx.x;
    const int64_t sbase = dims_base(d, s), slen = dims_len(d, s);
    if (t >= slen) return;
    for (int j = 0; j < n_jobs; j++) {
        const RowJobDev &job = jobs[j];
        switch (job.kind) {
#define X(OP)                                                                                                                  \
    case OP::ROW_ID: {                                                                                                         \
        static_assert(sizeof(RowBlob<OP>) <= ROW_FUSE_BLOB, "ROW blob too large for the fused grid");                           \
        const RowBlob<OP> &rb = *reinterpret_cast<const RowBlob<OP> *>(job.blob);                                               \
        OP op = rb.op;                                                                                                         \
        Row<OP::NIN> r;                                                                                                        \
        r.len = slen;                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < OP::NIN; k++) r.in[k] = rb.in.p[k] + sbase;                                     \
        typename OP::OutT y[OP::NOUT];                                                                                         \
        op.eval(r, t, y);                                                                                                      \
        _Pragma("unroll") for (int k = 0; k < OP::NOUT; k++) __builtin_nontemporal_store(y[k], &rb.out.p[k][sbase + t]);      \
    } break;
            ROW_OPS(X)
#undef X
        default: break;
        }
    }
}
static bool row_fusable(const RowThunk &t, const Dims &d) {
    return t.row_id > 0 && t.blob_bytes <= ROW_FUSE_BLOB && t.dims.n == d.n && t.dims.len == d.len && t.dims.stride == d.stride && t.dims.offs == d.offs;
}

// Small-shard schedule: which earlier (phase, chain) launches must have finished before chain c of phase q starts -- those that wrote a
// column it reads or writes, or read a column it writes.  The launches of one phase are independent of each other (phase_for), a chain
// is a stream (its own earlier phases are ordered before it), and per foreign chain only its latest phase needs a wait.
static int chain_of_row(const Phase &p, size_t k) { return p.row_late[k] ? p.row_late[k] - 1 : p.row_chain; }
static void small_deps(Recorder &r) {
    // Chains.  Phase 0: the LONG grid (jobs nobody waits for) on the caller's stream, the SHORT grid (jobs that feed later phases) on chain
    // 1, the ROW launches on chain 3.  The job grids of LATER phases run on chain 2 -- idle otherwise (no op is register-heavy): behind
    // chain 1 they would queue up behind the phase-0 producers they do not all depend on.
    bool heavy_any = false;
    for (const Phase &p : r.phases) for (const SeqJob &j : p.seq) heavy_any |= j.cls == CLS_HEAVY;
    for (size_t q = 0; q < r.phases.size(); q++) {
        Phase &p = r.phases[q];
        for (int c = 0; c < NCLS; c++) p.chain_of[c] = c;
        p.row_chain = ROW_CHAIN;
        // (the ROW launches of later phases too: they are links of the same dependency chains -- MACD = fast - slow under MACDEXT, fast-k
        //  of RSI under STOCHRSI -- and must not queue behind the patterns, the backtest and the fused ROW grid of phase 0)
        if (q > 0 && !heavy_any) { p.chain_of[CLS_SHORT] = CLS_HEAVY; p.row_chain = CLS_HEAVY; }
        // ROW chain: the launches whose columns a later phase reads go first (they are the heads of dependency chains), the others keep
        // their recorded order
        std::vector<size_t> idx(p.rows.size());
        for (size_t k = 0; k < idx.size(); k++) idx[k] = k;
        auto feeds = [&](const RowThunk &t) {
            for (int i = 0; i < t.n_writes; i++) if (r.feeds_seq.count(t.writes[i])) return true;
            return false;
        };
        auto rank = [&](const RowThunk &t) { return !feeds(t) ? 2 : t.long_launch ? 1 : 0; }; // short heads, long heads, the rest
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return rank(p.rows[a]) < rank(p.rows[b]); });
        std::vector<RowThunk> rows2; std::vector<char> late2, fused2;
        for (size_t k : idx) { rows2.push_back(p.rows[k]); late2.push_back(p.row_late[k]); fused2.push_back(p.row_fused[k]); }
        p.rows.swap(rows2); p.row_late.swap(late2); p.row_fused.swap(fused2);
        p.n_heads = 0;
        while ((size_t)p.n_heads < p.rows.size() && feeds(p.rows[(size_t)p.n_heads]) && !p.row_late[(size_t)p.n_heads] && !p.row_fused[(size_t)p.n_heads]) p.n_heads++;
        p.n_heads_short = 0;
        while (p.n_heads_short < p.n_heads && !p.rows[(size_t)p.n_heads_short].long_launch) p.n_heads_short++;
    }
    // The chain of the links runs a phase's ROW launches and its job grid one after the other.  The ROW launches of phase 1 wait for the
    // phase-0 PRODUCERS (MACD = fast - slow for MACDEXT's two averages, fast-k of RSI for RSI), its job grid for links that finished long
    // ago (STOCH's averages for the fast-k kernel): if the producers outlast a link, the grid goes first (1 250 symbols, RSI as a job: 1.58
    // against 1.79 ms per step); if they do not, the ROW launches do (625 symbols, RSI in its wave form: 1.25 against 1.28).
    {
        int producers = 0, links = 0;
        if (!r.phases.empty()) for (const SeqJob &j : r.phases[0].seq) if (j.cls == CLS_SHORT) producers = std::max(producers, j.cost);
        if (r.phases.size() > 1) for (const SeqJob &j : r.phases[1].seq) links = std::max(links, j.cost);
        r.grids_first = 2 * producers > 3 * links;
        if (const char *e = getenv("PQ_SMALL_GRIDS_FIRST")) r.grids_first = atoi(e) != 0;
    }
    std::map<const void *, std::vector<std::pair<int, int>>> last_w, readers; // (several jobs of ONE phase may write disjoint rows of a column: MAVP's blocks)
    for (size_t q = 0; q < r.phases.size(); q++) {
        Phase &p = r.phases[q];
        // R / W: what the launches of chain c touch (its waits); RP / WP: the same per SLOT that later phases wait for -- the chain, or
        // slot NCHAIN for the heads of the ROW chain
        // (the waits of a chain are split by part: its job grids [0] and its ROW launches [1] are independent of each other within a phase,
        //  and the grid of a phase must not wait for what only the ROW launches behind it need -- STOCH's averages for MACDEXT's producers)
        std::set<const void *> R2[NCHAIN][2], W2[NCHAIN][2], RP[NCHAIN + 2], WP[NCHAIN + 2];
        for (int c = 0; c < NCHAIN; c++) { p.deps[c][0].clear(); p.deps[c][1].clear(); p.work[c] = false; }
        for (const SeqJob &j : p.seq) {
            const int c = p.chain_of[j.cls];
            p.work[c] = true;
            for (int k = 0; k < j.nin; k++) { R2[c][0].insert(j.in[k]); RP[c].insert(j.in[k]); }
            for (int k = 0; k < j.nout; k++) { W2[c][0].insert(j.out[k]); WP[c].insert(j.out[k]); }
            for (int k = 0; k < 4; k++) {
                if (j.xr[k]) { R2[c][0].insert(j.xr[k]); RP[c].insert(j.xr[k]); }
                if (j.xw[k]) { W2[c][0].insert(j.xw[k]); WP[c].insert(j.xw[k]); }
            }
        }
        for (size_t k = 0; k < p.rows.size(); k++) {
            const int c = chain_of_row(p, k);
            p.work[c] = true;
            const int slot = (int)k < p.n_heads_short ? NCHAIN : (int)k < p.n_heads ? NCHAIN + 1 : c;
            for (int i = 0; i < p.rows[k].n_reads; i++) { R2[c][1].insert(p.rows[k].reads[i]); RP[slot].insert(p.rows[k].reads[i]); }
            for (int i = 0; i < p.rows[k].n_writes; i++) if (p.rows[k].writes[i]) { W2[c][1].insert(p.rows[k].writes[i]); WP[slot].insert(p.rows[k].writes[i]); }
        }
        for (int cp = 0; cp < 2 * NCHAIN; cp++) {
            const int c = cp / 2, part = cp % 2;
            const std::set<const void *> &R = R2[c][part], &W = W2[c][part];
            int latest[NCHAIN + 2];
            for (int x = 0; x <= NCHAIN + 1; x++) latest[x] = -1;
            auto need = [&](const std::pair<int, int> &d) { // (an earlier phase of the same stream is ordered by the stream)
                const int stream_of = d.second >= NCHAIN ? r.phases[(size_t)d.first].row_chain : d.second;
                if (stream_of != c && d.first > latest[d.second]) latest[d.second] = d.first;
            };
            for (const void *col : R) { auto it = last_w.find(col); if (it != last_w.end()) for (const auto &d : it->second) need(d); }
            for (const void *col : W) {
                auto it = last_w.find(col); if (it != last_w.end()) for (const auto &d : it->second) need(d);
                auto ir = readers.find(col); if (ir != readers.end()) for (const auto &d : ir->second) need(d);
            }
            // (a wait for the whole ROW chain of a phase covers its heads, one for all heads the short ones)
            for (int x = NCHAIN; x <= NCHAIN + 1; x++)
                if (latest[x] >= 0 && latest[r.phases[(size_t)latest[x]].row_chain] >= latest[x]) latest[x] = -1;
            if (latest[NCHAIN] >= 0 && latest[NCHAIN + 1] >= latest[NCHAIN]) latest[NCHAIN] = -1;
            for (int x = 0; x <= NCHAIN + 1; x++) if (latest[x] >= 0) p.deps[c][part].push_back({latest[x], x});
        }
        // The chain of the links runs a phase's job grid and its ROW launches one after the other: the part whose inputs are there first goes
        // first.  If everything the grid waits for the ROW launches wait for as well (STOCH's averages: the fast-k kernel; the ROW launches
        // of the phase: RSI in its wave form behind it), the grid does; else the cost rule above decides.
        if (q > 0) {
            const int c = p.row_chain;
            auto implied = [&](const std::pair<int, int> &a) {
                for (const auto &b : p.deps[c][1]) {
                    if (b.second == a.second && b.first >= a.first) return true;
                    if (b.first == a.first && a.second >= NCHAIN && (b.second == NCHAIN + 1 || b.second == r.phases[(size_t)a.first].row_chain)) return true;
                }
                return false;
            };
            bool subset = !p.deps[c][1].empty();
            for (const auto &a : p.deps[c][0]) subset &= implied(a);
            p.grid_first = r.grids_first || (subset && p.deps[c][0].size() < p.deps[c][1].size());
            if (const char *e = getenv("PQ_SMALL_GRIDS_FIRST")) p.grid_first = atoi(e) != 0;
        }
        for (int c = 0; c <= NCHAIN + 1; c++) {
            for (const void *col : WP[c]) {
                auto &lw = last_w[col];
                if (!lw.empty() && lw.front().first != (int)q) lw.clear();
                lw.push_back({(int)q, c});
                readers.erase(col);
            }
            for (const void *col : RP[c]) readers[col].push_back({(int)q, c});
        }
    }
    if (getenv("PQ_SUITE_PLAN")) { // debug: the schedule as text -- per phase and chain its launches and what each part waits for
        fprintf(stderr, "[pq plan] small-shard schedule: %zu phases, grids first:", r.phases.size());
        for (const Phase &p : r.phases) fprintf(stderr, " %d", (int)p.grid_first);
        fprintf(stderr, "\n");
        for (size_t q = 0; q < r.phases.size(); q++) {
            const Phase &p = r.phases[q];
            for (int c = 0; c < NCHAIN; c++) {
                if (!p.work[c]) continue;
                fprintf(stderr, "[pq plan] phase %zu chain %d%s:", q, c, p.row_chain == c ? " (ROW chain)" : "");
                for (int part = 0; part < 2; part++) {
                    fprintf(stderr, " %s waits {", part ? "rows" : "grid");
                    for (const auto &d : p.deps[c][part]) fprintf(stderr, " (phase %d, %s)", d.first, d.second == NCHAIN ? "short heads" : d.second == NCHAIN + 1 ? "heads" : std::to_string(d.second).c_str());
                    fprintf(stderr, " }");
                }
                fprintf(stderr, "\n[pq plan]     jobs:");
                for (const SeqJob &j : p.seq) if (p.chain_of[j.cls] == c) fprintf(stderr, " %d(cost %d)", j.kind, j.cost);
                fprintf(stderr, "\n[pq plan]     rows:");
                for (size_t k = 0; k < p.rows.size(); k++)
                    if (chain_of_row(p, k) == c)
                        fprintf(stderr, " [id %d r%d w%d%s%s%s]", p.rows[k].row_id, p.rows[k].n_reads, p.rows[k].n_writes, (int)k < p.n_heads ? " head" : "",
                                p.row_late[k] ? " late" : "", p.row_fused[k] ? " fused" : "");
                fprintf(stderr, "\n");
            }
        }
    }
}
static pq_status suite_finalize(pq_ctx *ctx, Recorder &r) {
    if (r.small) { // columns that feed a sequential job of a later phase: from the last phase back, through the ROW launches
        r.feeds_seq.clear();
        for (size_t q = r.phases.size(); q-- > 0;) {
            Phase &p = r.phases[q];
            if (q > 0)
                for (const SeqJob &j : p.seq) {
                    for (int k = 0; k < j.nin; k++) r.feeds_seq.insert(j.in[k]);
                    for (int k = 0; k < 4; k++) if (j.xr[k]) r.feeds_seq.insert(j.xr[k]);
                }
            for (const RowThunk &t : p.rows) {
                bool feeds = false;
                for (int i = 0; i < t.n_writes; i++) feeds |= r.feeds_seq.count(t.writes[i]) > 0;
                if (feeds) for (int i = 0; i < t.n_reads; i++) r.feeds_seq.insert(t.reads[i]);
            }
        }
    }
    hipDeviceProp_t prop;
    PQ_HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    const unsigned tiles = (unsigned)((r.b.n_series + SEQ_BLOCK - 1) / SEQ_BLOCK);
    for (size_t pi = 0; pi < r.phases.size(); pi++) {
        Phase &p = r.phases[pi];
        p.gs_row.n_jobs = (int)p.rows.size();
        // The ROW launches are cheap streaming kernels, but beside the SEQ grids they get few wave slots and one chain of
        // them becomes the critical path of the step.  Two chains: the heaviest launches (by columns moved) stay on the ROW
        // chain from t = 0; the lighter half runs behind the SHORT grid, the first SEQ grid to drain (-4 % per step against
        // behind the LONG grid, A/B in one session).  Measured alternatives: all ROW launches on one chain (that chain becomes the
        // critical path), the light half behind the HEAVY grid, early fractions of 0 / 0.3 / 0.7: 2 - 8 % slower per step.
        p.row_late.assign(p.rows.size(), 0);
        if (r.small && pi > 0) {
            // Small shard, later phases: a ROW launch that feeds a sequential job further on (MACD = fast - slow under MACDEXT) is a LINK and
            // runs on the chain of the links; one that feeds nobody (ADXR from the ADX column, the hand-over check of the time-split Hilbert
            // job, MACDEXT's histogram) is a TAIL: it goes behind the LONG grid on the caller's stream -- in front of the links, waiting for
            // a job of the LONG grid, it would hold up the whole chain.
            // (a tail of the links themselves -- MACDEXT's histogram reads the signal line, a job of phase 2 -- follows them on THEIR chain: on
            //  the caller's stream it would cost the step's end one more dependency between two streams)
            bool heavy_any = false;
            for (const Phase &q : r.phases) for (const SeqJob &j : q.seq) heavy_any |= j.heavy && (j.lds_bytes > 0 || j.kind == SEQ_ID_BACKTEST || j.kind == SEQ_ID_BACKTEST + 1);
            const int links_chain = heavy_any ? CLS_SHORT : CLS_HEAVY; // (as small_deps maps the later phases)
            for (size_t k = 0; k < p.rows.size(); k++) {
                bool feeds = false, of_links = false;
                for (int i = 0; i < p.rows[k].n_writes; i++) feeds |= r.feeds_seq.count(p.rows[k].writes[i]) > 0;
                for (int i = 0; i < p.rows[k].n_reads; i++) {
                    bool own = false; // (a column the launch also writes -- the Hilbert hand-over check repairs its columns in place -- says nothing)
                    for (int w = 0; w < p.rows[k].n_writes; w++) own |= p.rows[k].writes[w] == p.rows[k].reads[i];
                    auto it = r.writer_phase.find(p.rows[k].reads[i]);
                    of_links |= !own && it != r.writer_phase.end() && it->second >= 1;
                }
                if (!feeds) p.row_late[k] = (char)(1 + (of_links ? links_chain : CLS_LONG));
            }
        }
        if (!p.seq.empty() && !r.small) { // (a small shard: every ROW launch of phase 0 on the ROW chain from t = 0 -- nothing there is a tail)
            double total = 0, early = 0;
            auto weight = [](const RowThunk &t) { return (double)(t.n_reads + t.n_writes); };
            for (const RowThunk &t : p.rows) total += weight(t);
            std::vector<size_t> idx(p.rows.size());
            for (size_t k = 0; k < idx.size(); k++) idx[k] = k;
            std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return weight(p.rows[a]) > weight(p.rows[b]); });
            for (size_t k : idx) {
                if (early < 0.5 * total) { early += weight(p.rows[k]); continue; }
                p.row_late[k] = 1 + CLS_SHORT;
            }
        }
        // fusable launches of one chain position become one grid (at least two of them, or there is nothing to share)
        auto fuse_rows = [&]() -> pq_status {
            p.row_fused.assign(p.rows.size(), 0);
#ifndef PQ_NO_ROW_FUSION // A/B builds: every ROW launch on its own
            const Dims bd = dims_of(&r.b);
            for (int pos = 0; pos <= NCHAIN; pos++) {
                std::vector<RowJobDev> jobs;
                for (size_t k = 0; k < p.rows.size(); k++)
                    if (p.row_late[k] == pos && row_fusable(p.rows[k], bd)) {
                        RowJobDev j;
                        memset(&j, 0, sizeof j);
                        j.kind = p.rows[k].row_id;
                        memcpy(j.blob, p.rows[k].blob, (size_t)p.rows[k].blob_bytes);
                        jobs.push_back(j);
                    }
                if (jobs.size() < 2) continue;
                for (size_t k = 0; k < p.rows.size(); k++)
                    if (p.row_late[k] == pos && row_fusable(p.rows[k], bd)) p.row_fused[k] = 1;
                p.n_rows[pos] = (int)jobs.size();
                PQ_HIP_TRY(hipMalloc((void **)&p.d_rows[pos], sizeof(RowJobDev) * jobs.size()));
                PQ_HIP_TRY(hipMemcpy(p.d_rows[pos], jobs.data(), sizeof(RowJobDev) * jobs.size(), hipMemcpyHostToDevice));
            }
#endif
            return PQ_OK;
        };
        if (p.seq.empty()) { PQ_TRY(fuse_rows()); continue; }
        // class of every job (see the comment at CLS_*): heavy / gather by trait, the light tiled jobs by cost and LDS need
        std::stable_sort(p.seq.begin(), p.seq.end(), [](const SeqJob &a, const SeqJob &b) { return a.cost > b.cost; });
        double long_wgs = 0;
        const unsigned THIN_LDS_MAX = thin_lds_max(prop);
        unsigned widest = 1;
        int cmax = 1;
        for (const SeqJob &j : p.seq) {
            if (!j.heavy) widest = std::max(widest, j.lds_bytes);
            cmax = std::max(cmax, j.cost);
        }
        double long_fill = std::max(1.0, (double)(prop.maxSharedMemoryPerMultiProcessor / widest) - 0.5);
        if (const char *e = getenv("PQ_LONG_FILL")) long_fill = atof(e); // A/B runs: workgroups per CU placed in the LONG grid (large: one grid for every light job)
        const double long_budget = long_fill * (double)prop.multiProcessorCount;
        // the longest jobs are the critical path of a step: their waves win the issue arbitration against shorter jobs on the same
        // SIMD.  Relative to the longest job of the phase (>= 0.8 / 0.58 / 0.4 of it), not to absolute microseconds.
        for (SeqJob &j : p.seq) j.prio = j.cost >= 0.8 * cmax ? 3 : j.cost >= 0.58 * cmax ? 2 : j.cost >= 0.4 * cmax ? 1 : 0;
        // Small shard: the chip is mostly idle, a grid ends with its longest job, and a stream runs its grids one after the other.  So the
        // jobs whose columns a LATER phase reads (the first links of the chains a composite function was taken apart into: RSI under
        // STOCHRSI, the two averages of MACDEXT) form the SHORT grid of phase 0, every job of a later phase follows on the same chain, and
        // the jobs nobody waits for -- the long ones among them -- are the LONG grid, which nothing queues behind.
        // "feeds a later phase" = some SEQUENTIAL job of a later phase reads the column, directly or through ROW launches: a column that only
        // a trailing ROW launch reads (ADX under ADXR, the chunks of a time-split job under their hand-over check) costs that launch a few
        // microseconds behind the LONG grid and makes nobody else wait.
        auto feeds_later = [&](const SeqJob &j) {
            for (int k = 0; k < j.nout; k++) if (r.feeds_seq.count(j.out[k])) return true;
            return false;
        };
        for (SeqJob &j : p.seq) {
            const bool bt = j.kind == SEQ_ID_BACKTEST || j.kind == SEQ_ID_BACKTEST + 1; // the per-lane scan lives in the heavy kernel
            if (j.heavy && (j.lds_bytes > 0 || bt)) j.cls = CLS_HEAVY;
            else if (j.lds_bytes == 0) j.cls = CLS_GATHER;
            // (the chunks of a time-split job are read by their hand-over check only -- a few microseconds behind the LONG grid: no producer)
            else if (r.small) j.cls = (pi > 0 || feeds_later(j)) ? CLS_SHORT : CLS_LONG;
            else if (j.lds_bytes > THIN_LDS_MAX) j.cls = CLS_LONG;
            else if (long_wgs + tiles <= long_budget) { j.cls = CLS_LONG; long_wgs += tiles; }
            else j.cls = CLS_SHORT;
        }
        // A ROW launch of the NEXT phase whose inputs come only from sequential jobs of ONE class of this phase (e.g. dcphase / sine /
        // leadsine from the Hilbert job's phasor columns) needs no phase barrier: behind that class's grid on the same stream it is
        // ordered after its producers and overlaps the other chains' tails instead of extending the step.
        if (pi + 1 < r.phases.size() && !r.small) { // (a small shard orders its chains by data dependencies: suite_launch_small)
            Phase &q = r.phases[pi + 1];
            std::vector<RowThunk> stay;
            for (const RowThunk &t : q.rows) {
                int cls = -1;
                bool ok = true;
                for (int k = 0; k < t.n_reads && ok; k++) {
                    if (!p.wr.count(t.reads[k])) continue;      // not produced in this phase
                    int c = -1;
                    for (const SeqJob &j : p.seq)
                        for (int o = 0; o < j.nout; o++) if (j.out[o] == t.reads[k]) c = j.cls;
                    if (c < 0 || (cls >= 0 && c != cls)) ok = false; // written by a ROW launch, or by jobs of two classes
                    cls = c;
                }
                for (int k = 0; k < t.n_writes && ok; k++)
                    if (p.rd.count(t.writes[k]) || p.wr.count(t.writes[k])) ok = false; // its outputs are touched in this phase
                if (!ok || cls < 0) { stay.push_back(t); continue; }
                p.rows.push_back(t);
                p.row_late.push_back((char)(1 + cls));
                for (int k = 0; k < t.n_reads; k++) p.rd.insert(t.reads[k]);
                for (int k = 0; k < t.n_writes; k++) p.wr.insert(t.writes[k]);
            }
            q.rows.swap(stay);
            p.gs_row.n_jobs = (int)p.rows.size();
        }
        PQ_TRY(fuse_rows());
        std::stable_sort(p.seq.begin(), p.seq.end(), [](const SeqJob &a, const SeqJob &b) { return a.cls < b.cls; }); // cost order kept
        const double rows = (double)r.b.n_series * (double)r.b.len;
        std::map<const void *, int> masked_seen[NCLS];
        for (int c = 0; c <= NCLS; c++) p.first[c] = 0;
        for (int c = 0; c < NCLS; c++) { p.lds[c] = 0; p.gs[c].alg_bytes = 0; p.gs[c].n_jobs = 0; }
        for (const SeqJob &j : p.seq) {
            const int g = j.cls;
            p.first[g + 1]++;
            p.lds[g] = std::max(p.lds[g], j.lds_bytes);
            // algorithmic bytes: 8 B per f64 column and row; a column written row-disjointly by several masked
            // jobs counts once; the backtest job reads price and writes position/cash/equity (+ 64 B/symbol summary)
            GridStat &st = p.gs[g];
            st.n_jobs++;
            if (j.kind == SEQ_ID_BACKTEST || j.kind == SEQ_ID_BACKTEST + 1) { st.alg_bytes += 32.0 * rows + 64.0 * r.b.n_series; continue; }
            st.alg_bytes += j.summary_bytes * r.b.n_series; // + the summary row of an op with an epilogue
            if (!j.masked) { st.alg_bytes += 8.0 * rows * j.alg_cols * (double)j.alg_frac; continue; }
            st.alg_bytes += 8.0 * rows * j.nin; // jobs that share one output column row-disjointly: the column counts once
            for (int k = 0; k < j.nout; k++) {
                if (masked_seen[g][j.out[k]]++) continue;
                st.alg_bytes += 8.0 * rows;
            }
        }
        for (int c = 0; c < NCLS; c++) { p.first[c + 1] += p.first[c]; p.gs[c].lds = p.lds[c]; }
        if (getenv("PQ_SUITE_DEBUG"))
            for (const SeqJob &j : p.seq) fprintf(stderr, "[pq suite] job kind=%d nin=%d nout=%d lds=%u cost=%d class=%d\n", j.kind, j.nin, j.nout, j.lds_bytes, j.cost, j.cls);
        PQ_HIP_TRY(hipMalloc((void **)&p.d_seq, sizeof(SeqJob) * p.seq.size()));
        PQ_HIP_TRY(hipMemcpyAsync(p.d_seq, p.seq.data(), sizeof(SeqJob) * p.seq.size(), hipMemcpyHostToDevice, ctx->stream));
        if (getenv("PQ_SUITE_DEBUG")) PQ_HIP_TRY(hipMalloc((void **)&p.d_dbg, 16 * p.seq.size()));
    }
    if (r.small) small_deps(r);
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream)); // the host vectors are pageable
    for (size_t pi = r.phases.size(); pi-- > 0;) // a phase whose only launches were hoisted is gone
        if (r.phases[pi].seq.empty() && r.phases[pi].rows.empty()) r.phases.erase(r.phases.begin() + (long)pi);
    return PQ_OK;
}
// The replay of a SMALL-shard recording.  Same chains (streams), same grids; what differs is the ordering between them: no phase
// barrier -- a barrier makes every chain wait for the longest job of a phase, and on a mostly idle chip the step IS its longest chain --
// but, per chain and phase, a wait for exactly the earlier launches it depends on (small_deps; a dependency between two streams costs
// ~10 us on this runtime, scripts/ubench/xstream.hip).  A composite function taken apart into its basic calls (fused.hip: STOCH = the
// fast-k ROW kernel, then two moving-average jobs) then costs the sum of ITS links, beside everything else.
static pq_status suite_launch_small(pq_ctx *ctx, Recorder &r) {
    const Dims d = dims_of(&r.b);
    if (!r.ev_fork) PQ_HIP_TRY(hipEventCreateWithFlags(&r.ev_fork, hipEventDisableTiming));
    const unsigned tiles = (unsigned)((r.b.n_series + SEQ_BLOCK * 8 - 1) / (SEQ_BLOCK * 8)) * 8;
    hipStream_t chain_st[NCHAIN] = {ctx->stream, nullptr, nullptr, nullptr};
    bool started[NCHAIN] = {true, false, false, false};
    int last_phase[NCHAIN] = {-1, -1, -1, -1};
    bool any_side = false;
    for (Phase &p : r.phases) for (int c = 1; c < NCHAIN; c++) any_side |= p.work[c];
    if (any_side) PQ_HIP_TRY(hipEventRecord(r.ev_fork, ctx->stream));
    for (size_t q = 0; q < r.phases.size(); q++) {
        Phase &p = r.phases[q];
        if (p.d_dbg && atoi(getenv("PQ_SUITE_DEBUG")) >= 2 && !p.d_wg && !p.seq.empty()) {
            p.wg_tiles = tiles;
            PQ_HIP_TRY(hipMalloc((void **)&p.d_wg, 24 * (size_t)tiles * p.seq.size()));
        }
        if (p.d_wg) PQ_HIP_TRY(hipMemset(p.d_wg, 0, 24 * (size_t)tiles * p.seq.size()));
        if (p.d_dbg) { // min slots start at ~0, max slots at 0
            std::vector<unsigned long long> init(2 * p.seq.size());
            for (size_t i = 0; i < p.seq.size(); i++) { init[2 * i] = ~0ULL; init[2 * i + 1] = 0; }
            PQ_HIP_TRY(hipMemcpy(p.d_dbg, init.data(), 16 * p.seq.size(), hipMemcpyHostToDevice));
        }
        auto launch_row_grid = [&](int pos, hipStream_t st) { // the fused ROW grid of a position (0: early on the phase's ROW chain; 1 + c: late on chain c)
            if (!p.n_rows[pos]) return;
            for (int64_t sb = 0; sb < d.n; sb += 65535) { // grid.y is limited to 65535: slice the series axis
                const int64_t ns = d.n - sb < 65535 ? d.n - sb : 65535;
                hipLaunchKernelGGL(row_jobs_kernel, dim3((unsigned)((d.len + ROW_BLOCK - 1) / ROW_BLOCK), (unsigned)ns), dim3(ROW_BLOCK), 0, st,
                                   (const RowJobDev *)p.d_rows[pos], p.n_rows[pos], d, sb);
            }
        };
        for (int oi = 0; oi < NCHAIN; oi++) {
            const int c = k_chain_order[oi]; // a CHAIN (stream) here: its early ROW launches, the grids of the classes mapped to it, its late ROW launches
            if (!p.work[c]) continue;
            if (c != 0 && !chain_st[c]) {
                if (!ctx->suite_aux[c]) {
                    int prio_lo = 0, prio_hi = 0;
                    PQ_HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
                    PQ_HIP_TRY(hipStreamCreateWithPriority(&ctx->suite_aux[c], hipStreamNonBlocking, prio_hi));
                }
                chain_st[c] = r.aux[c] = ctx->suite_aux[c];
            }
            hipStream_t st = chain_st[c];
            if (!started[c]) { PQ_HIP_TRY(hipStreamWaitEvent(st, r.ev_fork, 0)); started[c] = true; }
            auto wait_part = [&](int part) -> hipError_t {
                for (const auto &dep : p.deps[c][part]) {
                    const hipError_t e = hipStreamWaitEvent(st, r.phases[(size_t)dep.first].ev_done[dep.second], 0);
                    if (e != hipSuccess) return e;
                }
                return hipSuccess;
            };
            // Phase 0: the ROW launches first (the heads of the chains are among them), then the grid.  Later phases (the chain of the
            // links): whichever part has its inputs first (small_deps) -- a stream runs in order.
            const bool grids_first = q > 0 && p.grid_first;
            auto launch_early_rows = [&]() -> pq_status {
                if (p.row_chain != c) return PQ_OK;
                PQ_HIP_TRY(wait_part(1));
                for (size_t k = 0; k < p.rows.size(); k++) {
                    if (!p.row_late[k] && !p.row_fused[k]) p.rows[k].launch(p.rows[k].blob, st);
                    // the (short) heads of the chain are out: what waits for them need not wait for the rest
                    for (int hs = 0; hs < 2; hs++)
                        if ((int)k + 1 == (hs ? p.n_heads : p.n_heads_short) && (hs == 0 || p.n_heads > p.n_heads_short)) {
                            if (!p.ev_done[NCHAIN + hs]) PQ_HIP_TRY(hipEventCreateWithFlags(&p.ev_done[NCHAIN + hs], hipEventDisableTiming));
                            PQ_HIP_TRY(hipEventRecord(p.ev_done[NCHAIN + hs], st));
                        }
                }
                launch_row_grid(0, st);
                return PQ_OK;
            };
            if (!grids_first) PQ_TRY(launch_early_rows());
            PQ_HIP_TRY(wait_part(0));
            for (int cls = 0; cls < NCLS; cls++) {
                const int nj = p.first[cls + 1] - p.first[cls];
                if (p.chain_of[cls] != c || nj <= 0) continue;
                GridStat &g = p.gs[cls];
                // timed runs (pq_suite_set_timing): the start / stop events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL) --
                // two hipEventRecord markers around every grid cost a 625-symbol step 0.1 ms of its 1.25
                const bool tm = r.timing && g.runs < Recorder::MAX_TIMED_RUNS;
                hipEvent_t ev0 = nullptr, ev1 = nullptr;
                if (tm) {
                    while (g.ev.size() < (size_t)g.runs * 2 + 2) { hipEvent_t e; PQ_HIP_TRY(hipEventCreate(&e)); g.ev.push_back(e); }
                    ev0 = g.ev[(size_t)g.runs * 2]; ev1 = g.ev[(size_t)g.runs * 2 + 1];
                    g.runs++;
                }
                unsigned long long *dbg = p.d_dbg ? p.d_dbg + 2 * p.first[cls] : nullptr, *wg = p.d_wg ? p.d_wg + 3 * (size_t)tiles * p.first[cls] : nullptr;
                const dim3 grid(tiles, (unsigned)nj);
                int v = k_variant[cls];
                if (v == 0)
                    for (int jx = p.first[cls]; jx < p.first[cls + 1]; jx++) if (p.seq[jx].unal) v = 3;
                const SeqJob *jobs = p.d_seq + p.first[cls];
                if (v == 3) hipExtLaunchKernelGGL(seq_jobs_kernel<3>, grid, dim3(SEQ_LDS_BLOCK), p.lds[cls], st, ev0, ev1, 0, jobs, d, dbg, wg);
                else if (v == 2) hipExtLaunchKernelGGL(seq_jobs_kernel<2>, grid, dim3(SEQ_BLOCK), 0, st, ev0, ev1, 0, jobs, d, dbg, wg);
                else if (v == 1) hipExtLaunchKernelGGL(seq_jobs_kernel<1>, grid, dim3(SEQ_LDS_BLOCK), p.lds[cls], st, ev0, ev1, 0, jobs, d, dbg, wg);
                else hipExtLaunchKernelGGL(seq_jobs_kernel<0>, grid, dim3(SEQ_LDS_BLOCK), p.lds[cls], st, ev0, ev1, 0, jobs, d, dbg, wg);
            }
            if (grids_first) PQ_TRY(launch_early_rows());
            bool late_any = false;
            for (size_t k = 0; k < p.rows.size(); k++) late_any |= p.row_late[k] == 1 + c;
            if (late_any && p.row_chain != c) PQ_HIP_TRY(wait_part(1)); // (tails on a chain that has no early ROW launches of this phase)
            for (size_t k = 0; k < p.rows.size(); k++)
                if (p.row_late[k] == 1 + c && !p.row_fused[k]) p.rows[k].launch(p.rows[k].blob, st);
            launch_row_grid(1 + c, st);
            if (!p.ev_done[c]) PQ_HIP_TRY(hipEventCreateWithFlags(&p.ev_done[c], hipEventDisableTiming));
            PQ_HIP_TRY(hipEventRecord(p.ev_done[c], st));
            last_phase[c] = (int)q;
        }
    }
    for (int c = 1; c < NCHAIN; c++)
        if (last_phase[c] >= 0) PQ_HIP_TRY(hipStreamWaitEvent(ctx->stream, r.phases[(size_t)last_phase[c]].ev_done[c], 0));
    bool dbg_any = false;
    for (Phase &p : r.phases) dbg_any |= p.d_dbg != nullptr;
    if (dbg_any) { // debug only: wait and print the per-job schedule (100 MHz device clock), every phase against the step's first start
        PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
        unsigned long long t0 = ~0ULL;
        std::vector<std::vector<unsigned long long>> ts(r.phases.size());
        for (size_t q = 0; q < r.phases.size(); q++) {
            Phase &p = r.phases[q];
            if (!p.d_dbg) continue;
            ts[q].resize(2 * p.seq.size());
            PQ_HIP_TRY(hipMemcpy(ts[q].data(), p.d_dbg, 16 * p.seq.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < p.seq.size(); i++) t0 = ts[q][2 * i] < t0 ? ts[q][2 * i] : t0;
        }
        for (size_t q = 0; q < r.phases.size(); q++)
            for (size_t i = 0; i * 2 < ts[q].size(); i++)
                fprintf(stderr, "[pq suite] job %2zu kind=%3d class=%d lds=%6u  start %8.1f us  end %8.1f us  phase %zu\n", i, r.phases[q].seq[i].kind,
                        r.phases[q].seq[i].cls, r.phases[q].seq[i].lds_bytes, (double)(ts[q][2 * i] - t0) / 100.0, (double)(ts[q][2 * i + 1] - t0) / 100.0, q);
        for (size_t q = 0; q < r.phases.size(); q++) {
            Phase &p = r.phases[q];
            if (!p.d_wg) continue;
            std::vector<unsigned long long> w(3 * (size_t)tiles * p.seq.size());
            PQ_HIP_TRY(hipMemcpy(w.data(), p.d_wg, 8 * w.size(), hipMemcpyDeviceToHost));
            for (size_t j = 0; j < p.seq.size(); j++)
                for (unsigned x = 0; x < tiles; x++) {
                    const unsigned long long *e = &w[3 * (j * tiles + x)];
                    if (e[1]) fprintf(stderr, "[pq wg] %zu.%zu %u %.2f %.2f %llu %llu %u\n", q, j, x, (double)(e[0] - t0) / 100.0, (double)(e[1] - t0) / 100.0,
                                      e[2] >> 32, e[2] & 0xffffffffULL, p.seq[j].lds_bytes);
                }
        }
    }
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}
static pq_status suite_launch(pq_ctx *ctx, Recorder &r) {
    if (r.small) return suite_launch_small(ctx, r);
    Dims d = dims_of(&r.b);
    if (!r.ev_fork) PQ_HIP_TRY(hipEventCreateWithFlags(&r.ev_fork, hipEventDisableTiming));
    // Side streams are created when a chain first has work (an unused stream would still take its turn in the runtime's queue
    // assignment) and BELONG TO THE CONTEXT: every suite replayed on a context uses the same three, so that two recordings of one step
    // (Suite.record(summaries=[a, b]), the double-buffered form of a multi-GPU run) do not put seven streams onto the runtime's four
    // hardware queues -- measured with per-suite streams: 5.04 instead of 3.9 ms per step when the two recordings alternate.
    auto side_stream = [&](int i) -> hipError_t {
        if (!ctx->suite_aux[i]) {
            int prio_lo = 0, prio_hi = 0;
            hipError_t e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi); // numerically lower = higher priority
            if (e != hipSuccess) return e;
            // (at low priority the short ROW kernels crawl behind the SEQ grids and end up as the critical path of the step)
            if ((e = hipStreamCreateWithPriority(&ctx->suite_aux[i], hipStreamNonBlocking, prio_hi)) != hipSuccess) return e;
        }
        r.aux[i] = ctx->suite_aux[i];
        if (r.ev_join[i]) return hipSuccess;
        return hipEventCreateWithFlags(&r.ev_join[i], hipEventDisableTiming);
    };
    // workgroups go to the 8 XCDs round-robin by linear id: with grid.x a multiple of 8, series tile x runs on XCD x % 8 for
    // EVERY job, so jobs that read the same input column share that XCD's L2 for it
    const unsigned tiles = (unsigned)((r.b.n_series + SEQ_BLOCK * 8 - 1) / (SEQ_BLOCK * 8)) * 8;
    for (Phase &p : r.phases) {
        // the launches of one phase are independent of each other: one chain (stream) per class, chain 0 = the caller's
        auto timed = [&](GridStat &g, hipStream_t st, bool begin) -> hipError_t { // HIP events on the launch stream
            if (!r.timing || g.runs >= Recorder::MAX_TIMED_RUNS) return hipSuccess;
            size_t idx = (size_t)g.runs * 2 + (begin ? 0 : 1);
            while (g.ev.size() <= idx) {
                hipEvent_t e;
                hipError_t er = hipEventCreate(&e);
                if (er != hipSuccess) return er;
                g.ev.push_back(e);
            }
            hipError_t er = hipEventRecord(g.ev[idx], st);
            if (!begin) g.runs++;
            return er;
        };
        auto njobs = [&](int c) { return p.first[c + 1] - p.first[c]; };
        bool side[NCHAIN] = {};
        for (int c = 0; c < NCLS; c++) side[c] |= njobs(c) > 0;
        side[ROW_CHAIN] |= !p.rows.empty();
        for (size_t k = 0; k < p.rows.size(); k++)
            if (p.row_late[k]) side[p.row_late[k] - 1] = true; // a chain with late ROW launches runs even without SEQ jobs of its own
        bool any_side = false;
        for (int i = 1; i < NCHAIN; i++) any_side |= side[i];
        if (p.d_dbg && atoi(getenv("PQ_SUITE_DEBUG")) >= 2 && !p.d_wg) {
            p.wg_tiles = tiles;
            PQ_HIP_TRY(hipMalloc((void **)&p.d_wg, 24 * (size_t)tiles * p.seq.size()));
        }
        if (p.d_wg) PQ_HIP_TRY(hipMemset(p.d_wg, 0, 24 * (size_t)tiles * p.seq.size()));
        if (p.d_dbg) { // min slots start at ~0, max slots at 0
            std::vector<unsigned long long> init(2 * p.seq.size());
            for (size_t i = 0; i < p.seq.size(); i++) { init[2 * i] = ~0ULL; init[2 * i + 1] = 0; }
            PQ_HIP_TRY(hipMemcpy(p.d_dbg, init.data(), 16 * p.seq.size(), hipMemcpyHostToDevice));
        }
        if (any_side) PQ_HIP_TRY(hipEventRecord(r.ev_fork, ctx->stream));
        auto launch_class = [&](int c, hipStream_t st) -> pq_status {
            const int nj = njobs(c);
            if (nj <= 0) return PQ_OK;
            // (timed runs: the events ride on the grid's own dispatch packet, no marker packets around it)
            GridStat &g = p.gs[c];
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            if (r.timing && g.runs < Recorder::MAX_TIMED_RUNS) {
                while (g.ev.size() < (size_t)g.runs * 2 + 2) { hipEvent_t e; PQ_HIP_TRY(hipEventCreate(&e)); g.ev.push_back(e); }
                ev0 = g.ev[(size_t)g.runs * 2]; ev1 = g.ev[(size_t)g.runs * 2 + 1];
                g.runs++;
            }
            unsigned long long *dbg = p.d_dbg ? p.d_dbg + 2 * p.first[c] : nullptr, *wg = p.d_wg ? p.d_wg + 3 * (size_t)tiles * p.first[c] : nullptr;
            const dim3 grid(tiles, (unsigned)nj);
            int v = k_variant[c];
            if (v == 0) // one job with 8-byte rows: the whole grid runs the 8-byte form (it handles aligned columns as well)
                for (int jx = p.first[c]; jx < p.first[c + 1]; jx++) if (p.seq[jx].unal) v = 3;
            const SeqJob *jobs = p.d_seq + p.first[c];
            if (v == 3) hipExtLaunchKernelGGL(seq_jobs_kernel<3>, grid, dim3(SEQ_LDS_BLOCK), p.lds[c], st, ev0, ev1, 0, jobs, d, dbg, wg);
            else if (v == 2) hipExtLaunchKernelGGL(seq_jobs_kernel<2>, grid, dim3(SEQ_BLOCK), 0, st, ev0, ev1, 0, jobs, d, dbg, wg);
            else if (v == 1) hipExtLaunchKernelGGL(seq_jobs_kernel<1>, grid, dim3(SEQ_LDS_BLOCK), p.lds[c], st, ev0, ev1, 0, jobs, d, dbg, wg);
            else hipExtLaunchKernelGGL(seq_jobs_kernel<0>, grid, dim3(SEQ_LDS_BLOCK), p.lds[c], st, ev0, ev1, 0, jobs, d, dbg, wg);
            return PQ_OK;
        };
        auto launch_rows = [&](int pos, hipStream_t st) { // the fused ROW grid of a chain position
            if (!p.n_rows[pos]) return;
            for (int64_t sb = 0; sb < d.n; sb += 65535) { // grid.y is limited to 65535: slice the series axis
                const int64_t ns = d.n - sb < 65535 ? d.n - sb : 65535;
                hipLaunchKernelGGL(row_jobs_kernel, dim3((unsigned)((d.len + ROW_BLOCK - 1) / ROW_BLOCK), (unsigned)ns), dim3(ROW_BLOCK), 0, st,
                                   (const RowJobDev *)p.d_rows[pos], p.n_rows[pos], d, sb);
            }
        };
        pq_status ps;
        for (int oi = 0; oi < NCHAIN; oi++) { // enqueue order: the longest / hungriest chains first
            const int i = k_chain_order[oi];
            if (!side[i] && i != 0) continue;
            if (i != 0) PQ_HIP_TRY(side_stream(i));
            hipStream_t st = i == 0 ? ctx->stream : r.aux[i];
            if (i != 0) PQ_HIP_TRY(hipStreamWaitEvent(st, r.ev_fork, 0));
            if (i == ROW_CHAIN && !p.rows.empty()) {
                PQ_HIP_TRY(timed(p.gs_row, st, true));
                for (size_t k = 0; k < p.rows.size(); k++)
                    if (!p.row_late[k] && !p.row_fused[k]) p.rows[k].launch(p.rows[k].blob, st);
                launch_rows(0, st);
                PQ_HIP_TRY(timed(p.gs_row, st, false));
            }
            if ((ps = launch_class(i, st)) != PQ_OK) return ps;
            bool own_launches = false;
            for (size_t k = 0; k < p.rows.size(); k++) // the lighter ROW launches follow an early-draining SEQ grid (suite_finalize)
                if (p.row_late[k] == 1 + i && !p.row_fused[k]) own_launches = true;
            // The tail of the SHORT chain is the wave-per-symbol backtest AND the fused ROW grid.  When no job of the phase needs the
            // register-heavy kernel, that chain's hardware queue is idle: the ROW grid goes there, gated on the SHORT grid by an event,
            // and runs beside the backtest instead of behind it (-2 % per step, and the step-to-step spread halves).
            if (i == CLS_SHORT && !side[CLS_HEAVY] && own_launches && p.n_rows[1 + i]) {
                if (!r.ev_tail) PQ_HIP_TRY(hipEventCreateWithFlags(&r.ev_tail, hipEventDisableTiming));
                PQ_HIP_TRY(hipEventRecord(r.ev_tail, st));
                PQ_HIP_TRY(side_stream(CLS_HEAVY));
                PQ_HIP_TRY(hipStreamWaitEvent(r.aux[CLS_HEAVY], r.ev_tail, 0));
                launch_rows(1 + i, r.aux[CLS_HEAVY]);
                PQ_HIP_TRY(hipEventRecord(r.ev_join[CLS_HEAVY], r.aux[CLS_HEAVY]));
                side[CLS_HEAVY] = true; // joined below
                for (size_t k = 0; k < p.rows.size(); k++)
                    if (p.row_late[k] == 1 + i && !p.row_fused[k]) p.rows[k].launch(p.rows[k].blob, st);
            } else {
                for (size_t k = 0; k < p.rows.size(); k++)
                    if (p.row_late[k] == 1 + i && !p.row_fused[k]) p.rows[k].launch(p.rows[k].blob, st);
                launch_rows(1 + i, st);
            }
            if (i != 0) PQ_HIP_TRY(hipEventRecord(r.ev_join[i], st));
        }
        for (int i = 1; i < NCHAIN; i++)
            if (side[i]) PQ_HIP_TRY(hipStreamWaitEvent(ctx->stream, r.ev_join[i], 0));
        if (p.d_dbg) { // debug only: wait and print the per-job schedule (100 MHz device clock)
            PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
            std::vector<unsigned long long> t(2 * p.seq.size());
            PQ_HIP_TRY(hipMemcpy(t.data(), p.d_dbg, 16 * p.seq.size(), hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ULL;
            for (size_t i = 0; i < p.seq.size(); i++) t0 = t[2 * i] < t0 ? t[2 * i] : t0;
            for (size_t i = 0; i < p.seq.size(); i++)
                fprintf(stderr, "[pq suite] job %2zu kind=%3d class=%d lds=%6u  start %8.1f us  end %8.1f us\n", i, p.seq[i].kind,
                        p.seq[i].cls, p.seq[i].lds_bytes, (double)(t[2 * i] - t0) / 100.0,
                        (double)(t[2 * i + 1] - t0) / 100.0);
            if (p.d_wg) {
                std::vector<unsigned long long> w(3 * (size_t)tiles * p.seq.size());
                PQ_HIP_TRY(hipMemcpy(w.data(), p.d_wg, 8 * w.size(), hipMemcpyDeviceToHost));
                for (size_t j = 0; j < p.seq.size(); j++)
                    for (unsigned x = 0; x < tiles; x++) {
                        const unsigned long long *q = &w[3 * (j * tiles + x)];
                        if (q[1]) fprintf(stderr, "[pq wg] %zu %u %.2f %.2f %llu %llu %u\n", j, x, (double)(q[0] - t0) / 100.0, (double)(q[1] - t0) / 100.0,
                                          q[2] >> 32, q[2] & 0xffffffffULL, p.seq[j].lds_bytes);
                    }
            }
#if defined(PQ_EXPERIMENTS) && defined(PQ_PROFILE_WAVES)
            {
                static unsigned long long prof[128][4], zero[128][4];
                PQ_HIP_TRY(hipMemcpyFromSymbol(prof, HIP_SYMBOL(pq_prof), sizeof(prof)));
                for (int y = 120; y < 124; y++)
                    fprintf(stderr, "[pq prof] SIMD %d: compute waves %llu, storer waves %llu\n", y - 120, prof[y][0], prof[y][1]);
                for (int y = 0; y < 120; y++)
                    if (prof[y][3])
                        fprintf(stderr, "[pq prof] kind=%3d  fill+loadwait %8.0f  rows %8.0f  handoff %8.0f  cycles per tile, tiles=%llu\n", y,
                                (double)prof[y][0] / prof[y][3], (double)prof[y][1] / prof[y][3], (double)prof[y][2] / prof[y][3], prof[y][3]);
                PQ_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(pq_prof), zero, sizeof(zero)));
            }
#endif
        }
    }
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}
static void suite_free(pq_ctx *ctx, Recorder &r) {
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < NCHAIN; i++) {
        if (r.aux[i]) { (void)hipStreamSynchronize(r.aux[i]); r.aux[i] = nullptr; } // (the stream is the context's: pq_ctx_destroy ends it)
        if (r.ev_join[i]) { (void)hipEventDestroy(r.ev_join[i]); r.ev_join[i] = nullptr; }
    }
    if (r.ev_fork) { (void)hipEventDestroy(r.ev_fork); r.ev_fork = nullptr; }
    if (r.ev_tail) { (void)hipEventDestroy(r.ev_tail); r.ev_tail = nullptr; }
    for (Phase &p : r.phases) {
        for (int c = 0; c <= NCHAIN + 1; c++) if (p.ev_done[c]) { (void)hipEventDestroy(p.ev_done[c]); p.ev_done[c] = nullptr; }
        if (p.d_seq) (void)hipFree(p.d_seq);
        for (int pos = 0; pos <= NCHAIN; pos++)
            if (p.d_rows[pos]) { (void)hipFree(p.d_rows[pos]); p.d_rows[pos] = nullptr; p.n_rows[pos] = 0; }
        if (p.d_dbg) (void)hipFree(p.d_dbg);
        if (p.d_wg) (void)hipFree(p.d_wg);
        for (GridStat &g : p.gs) { for (hipEvent_t e : g.ev) (void)hipEventDestroy(e); g.ev.clear(); g.runs = 0; }
        for (hipEvent_t e : p.gs_row.ev) (void)hipEventDestroy(e);
        p.gs_row.ev.clear(); p.gs_row.runs = 0;
    }
    for (void *s : r.scratch) (void)hipFree(s);
    r.phases.clear();
    r.scratch.clear();
}

void *rec_alloc_zero(pq_ctx *ctx, size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, bytes ? bytes : 4) != hipSuccess) { (void)hipFree(p); return nullptr; }
    ctx->rec->scratch.push_back(p);
    return p;
}
double *pq_ws_col(pq_ctx *ctx, const pq_batch *b, int k) {
    size_t col = batch_rows(b);
    if (ctx->rec) { // fresh column per request, owned by the suite
        void *p = nullptr;
        if (hipMalloc(&p, col * sizeof(double)) != hipSuccess) return nullptr;
        ctx->rec->scratch.push_back(p);
        return (double *)p;
    }
    return reinterpret_cast<double *>(ctx->ws) + (size_t)k * col;
}

SuiteScope::SuiteScope(pq_ctx *c, const pq_batch *b) : ctx(c), owner(false), status(PQ_OK) {
    if (ctx->rec) return;
    pq_suite *s = new pq_suite();
    s->rec.b = *b;
    ctx->rec = &s->rec;
    ctx->rec_small = s->rec.small = small_shard(ctx, b);
    owner = true;
}
SuiteScope::~SuiteScope() {
    if (owner && ctx->rec) { // error path: drop whatever was recorded
        Recorder *r = ctx->rec;
        ctx->rec = nullptr;
        suite_free(ctx, *r);
        delete reinterpret_cast<pq_suite *>(r);
    }
}
pq_status SuiteScope::finish() {
    if (!owner) return PQ_OK;
    Recorder *r = ctx->rec;
    ctx->rec = nullptr;
    owner = false;
    pq_status st = suite_finalize(ctx, *r);
    if (st == PQ_OK) st = suite_launch(ctx, *r);
    suite_free(ctx, *r); // drains the stream first
    delete reinterpret_cast<pq_suite *>(r);
    return st;
}

extern "C" {

pq_status pq_suite_begin(pq_ctx *ctx, const pq_batch *b) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(!ctx->rec, "pq_suite_begin: a suite is already being recorded on this context");
    pq_suite *s = new pq_suite();
    s->rec.b = *b;
    ctx->rec = &s->rec;
    ctx->rec_small = s->rec.small = small_shard(ctx, b);
    return PQ_OK;
}
pq_status pq_suite_end(pq_ctx *ctx, pq_suite **out) {
    PQ_REQUIRE(ctx && out, "pq_suite_end: null pointer");
    PQ_REQUIRE(ctx->rec, "pq_suite_end: no suite is being recorded");
    pq_suite *s = reinterpret_cast<pq_suite *>(ctx->rec);
    ctx->rec = nullptr;
    pq_status st = suite_finalize(ctx, s->rec);
    if (st != PQ_OK) { suite_free(ctx, s->rec); delete s; return st; }
    *out = s;
    return PQ_OK;
}
pq_status pq_suite_abort(pq_ctx *ctx) {
    PQ_REQUIRE(ctx, "pq_suite_abort: null pointer");
    if (!ctx->rec) return PQ_OK;
    pq_suite *s = reinterpret_cast<pq_suite *>(ctx->rec);
    ctx->rec = nullptr;
    suite_free(ctx, s->rec);
    delete s;
    return PQ_OK;
}
pq_status pq_suite_run(pq_ctx *ctx, pq_suite *s) {
    PQ_REQUIRE(ctx && s, "pq_suite_run: null pointer");
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    return suite_launch(ctx, s->rec);
}
pq_status pq_suite_destroy(pq_ctx *ctx, pq_suite *s) {
    PQ_REQUIRE(ctx, "pq_suite_destroy: null pointer");
    if (!s) return PQ_OK;
    suite_free(ctx, s->rec);
    delete s;
    return PQ_OK;
}
pq_status pq_suite_set_timing(pq_suite *s, int32_t on) {
    PQ_REQUIRE(s, "pq_suite_set_timing: null pointer");
    s->rec.timing = on != 0;
    for (Phase &p : s->rec.phases) for (GridStat &g : p.gs) g.runs = 0;
    return PQ_OK;
}
// grid k (in launch order: phase 0 small-LDS, phase 0 large-LDS, phase 1 small, ...; empty grids skipped).
// avg_ms = mean HIP-event time over the runs since pq_suite_set_timing(1) (0 if none); call after pq_ctx_sync.
pq_status pq_suite_grid_stats(pq_suite *s, int32_t k, double *avg_ms, double *alg_bytes, int32_t *n_jobs, int32_t *lds_bytes,
                              int32_t *runs) {
    PQ_REQUIRE(s, "pq_suite_grid_stats: null pointer");
    int idx = 0;
    for (Phase &p : s->rec.phases)
        for (int c = 0; c <= NCLS; c++) {
            GridStat &g = c < NCLS ? p.gs[c] : p.gs_row;
            if (g.n_jobs == 0) continue;
            if (idx++ != k) continue;
            double tot = 0;
            for (int i = 0; i < g.runs; i++) {
                float ms = 0;
                PQ_HIP_TRY(hipEventElapsedTime(&ms, g.ev[2 * i], g.ev[2 * i + 1]));
                tot += ms;
            }
            if (avg_ms) *avg_ms = g.runs ? tot / g.runs : 0.0;
            if (alg_bytes) *alg_bytes = g.alg_bytes;
            if (n_jobs) *n_jobs = g.n_jobs;
            if (lds_bytes) *lds_bytes = (int32_t)g.lds;
            if (runs) *runs = g.runs;
            return PQ_OK;
        }
    pq_set_error("pq_suite_grid_stats: grid index out of range");
    return PQ_ERR_ARG;
}
// The launches of one kernel overlap within a step: mean over the timed runs of (latest end - earliest start) over the grids
// that launch kernel `variant`, on the device clock (HIP events of different streams are comparable).
pq_status pq_suite_span_stats(pq_suite *s, int32_t variant, double *avg_span_ms, double *alg_bytes) {
    PQ_REQUIRE(s && avg_span_ms && alg_bytes, "pq_suite_span_stats: null pointer");
    std::vector<GridStat *> gs;
    for (Phase &p : s->rec.phases)
        for (int c = 0; c < NCLS; c++)
            if (p.gs[c].n_jobs > 0 && p.gs[c].runs > 0 && k_variant[c] == variant) gs.push_back(&p.gs[c]);
    *avg_span_ms = 0.0; *alg_bytes = 0.0;
    if (gs.empty()) return PQ_OK;
    int runs = gs[0]->runs;
    for (GridStat *g : gs) { runs = g->runs < runs ? g->runs : runs; *alg_bytes += g->alg_bytes; }
    double tot = 0.0;
    for (int i = 0; i < runs; i++) {
        float lo = 0.0f, hi = 0.0f;
        for (size_t k = 0; k < gs.size(); k++) {
            float a = 0.0f, b = 0.0f;
            if (k > 0) PQ_HIP_TRY(hipEventElapsedTime(&a, gs[0]->ev[2 * i], gs[k]->ev[2 * i]));
            PQ_HIP_TRY(hipEventElapsedTime(&b, gs[0]->ev[2 * i], gs[k]->ev[2 * i + 1]));
            lo = (k == 0 || a < lo) ? a : lo;
            hi = (k == 0 || b > hi) ? b : hi;
        }
        tot += (double)(hi - lo);
    }
    *avg_span_ms = runs ? tot / runs : 0.0;
    return PQ_OK;
}
pq_status pq_suite_grid_variant(pq_suite *s, int32_t k, int32_t *variant) {
    PQ_REQUIRE(s && variant, "pq_suite_grid_variant: null pointer");
    int idx = 0;
    for (Phase &p : s->rec.phases)
        for (int c = 0; c <= NCLS; c++) {
            if ((c < NCLS ? p.gs[c] : p.gs_row).n_jobs == 0) continue;
            if (idx++ != k) continue;
            *variant = c < NCLS ? k_variant[c] : 3; // 3 = the chain of ROW launches
            if (c < NCLS && k_variant[c] == 0)
                for (int jx = p.first[c]; jx < p.first[c + 1]; jx++) if (p.seq[jx].unal) *variant = 4; // 4 = seq_jobs_kernel<3>: the 8-byte form of the tiled body
            return PQ_OK;
        }
    pq_set_error("pq_suite_grid_variant: grid index out of range");
    return PQ_ERR_ARG;
}
pq_status pq_suite_info(const pq_suite *s, int32_t *n_phases, int32_t *n_seq_jobs, int32_t *n_row_launches) {
    PQ_REQUIRE(s, "pq_suite_info: null pointer");
    int32_t nj = 0, nr = 0;
    for (const Phase &p : s->rec.phases) { nj += (int32_t)p.seq.size(); nr += (int32_t)p.rows.size(); }
    if (n_phases) *n_phases = (int32_t)s->rec.phases.size();
    if (n_seq_jobs) *n_seq_jobs = nj;
    if (n_row_launches) *n_row_launches = nr;
    return PQ_OK;
}

} // extern "C"
