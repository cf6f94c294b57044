// fused.hip -- C ABI of the composite functions: one sequential job when the LDS body applies (ops_fused.h),
// the chained launches through scratch columns otherwise (very long windows, unaligned columns, MA types that are
// themselves multi-pass).  Both forms are bit-identical.
#include "ops_fused.h"
#include "wt_api.h"

extern "C" {
pq_status pq_trima_chain(pq_ctx *, const pq_batch *, const double *, int64_t, double *);
pq_status pq_cci_chain(pq_ctx *, const pq_batch *, const double *, const double *, const double *, int64_t, double *);
pq_status pq_adxr_chain(pq_ctx *, const pq_batch *, const double *, const double *, const double *, int64_t, double *);
pq_status pq_adxr_from_adx(pq_ctx *, const pq_batch *, const double *, int64_t, double *);
pq_status pq_apo_chain(pq_ctx *, const pq_batch *, const double *, int64_t, int64_t, int64_t, double *);
pq_status pq_ppo_chain(pq_ctx *, const pq_batch *, const double *, int64_t, int64_t, int64_t, double *);
pq_status pq_macdext_chain(pq_ctx *, const pq_batch *, const double *, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                           double *, double *, double *);
pq_status pq_stoch_chain(pq_ctx *, const pq_batch *, const double *, const double *, const double *, int64_t, int64_t,
                         int64_t, int64_t, int64_t, double *, double *);
pq_status pq_stochf_chain(pq_ctx *, const pq_batch *, const double *, const double *, const double *, int64_t, int64_t,
                          int64_t, double *, double *);
pq_status pq_stochrsi_chain(pq_ctx *, const pq_batch *, const double *, int64_t, int64_t, int64_t, int64_t, double *, double *);
}

#define CHK(name, cond) PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(cond, name ": null pointer")

// ---- the Hilbert job of a SMALL shard, split in time --------------------------------------------------------------------------------
// On a small shard the step ends with its longest job, and that is the Hilbert pipeline (cycle.rs:27-63): ~210 dependent f64
// instructions per row on a lone wavefront, 1.75 ms for 2 520 rows whether the chip holds 625 symbols or 2 500.  Its state is a fading
// memory (0.2 / 0.8 smoothing of I2 / Q2 / Re / Im and of the period, the period clamped against its own previous value): a walk that
// starts W = 640 rows early from zeros agrees with the full walk to a few ulp -- worst relative difference over 2 000 series x two
// starting rows: dcperiod 3.8e-15, phasor components 3.2e-16 of the price level (oracle, scripts/ht_warmup_error.py) -- which is the
// size of the device atan's own difference from the host's, for outputs that are tolerance-class (<= 1e-12) anyway.  (They do NOT
// merge bit for bit -- median 566 rows, worst 1 709 --, so this is not the bit-verified speculation of wt_dev.h.)  So pq_ht_all records C
// chunk jobs: chunk c owns rows [c * own, (c + 1) * own), starts min(W, c * own) rows early and keeps those rows to itself, except its
// last warm-up tile, which goes to check columns.  A check kernel behind the chunks compares that tile with what chunk c - 1 wrote for
// the same rows (|a - b| <= 1e-13 of the value, of the price level for the phasor components); a 64-series tile that fails anywhere --
// a series that has not converged, a NaN that a later chunk never saw -- is walked again in full by the ordinary job, gated.
constexpr int HT_TS_MAX_CHUNKS = 8;
struct HtTsBlob {
    HtAll6Op op;   // the full-length job of the gated fallback
    const double *real;
    double *out[3], *chk[3];
    pq_batch b;
    int n_chunks, tile_rows;
    int own_row[HT_TS_MAX_CHUNKS]; // first own row of chunk c (c >= 1 is checked)
    unsigned *gate;
    unsigned lds;
};
struct HtTsRows { int own_row[HT_TS_MAX_CHUNKS]; };
// (one launch for every hand-over of the batch: blockIdx.y + 1 = the chunk whose first own tile is compared)
__global__ __launch_bounds__(SEQ_BLOCK) void ht_ts_check_kernel(const double *real, const double *o0, const double *o1, const double *o2, const double *c0,
                                                               const double *c1, const double *c2, Dims d, HtTsRows rows, int tile_rows, unsigned *gate) {
    const int64_t s = (int64_t)blockIdx.x * SEQ_BLOCK + threadIdx.x;
    const int own_row = rows.own_row[blockIdx.y + 1];
    bool bad = false;
    if (s < d.n) {
        const double *o[3] = {o0, o1, o2}, *c[3] = {c0, c1, c2};
        const int64_t base = s * d.stride;
        for (int j = 0; j < tile_rows; j++) {
            const int64_t t = own_row - tile_rows + j;
            const double price = fabs(real[base + t]);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double a = o[k][base + t], e = c[k][base + t];
                const bool same = __double_as_longlong(a) == __double_as_longlong(e) || (a != a && e != e);
                const double scale = fmax(fmax(fabs(a), fabs(e)), k == 0 ? 0.0 : price);
                bad |= !(same || fabs(a - e) <= 1e-13 * scale);
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && threadIdx.x == 0) atomicOr(&gate[blockIdx.x], 1u);
}
static void ht_ts_launch(const void *blob, hipStream_t stream) {
    const HtTsBlob &w = *reinterpret_cast<const HtTsBlob *>(blob);
    const Dims d = dims_of(&w.b);
    const dim3 tiles((unsigned)((w.b.n_series + SEQ_BLOCK - 1) / SEQ_BLOCK));
    if (w.n_chunks > 1) {
        HtTsRows rows;
        for (int c = 0; c < HT_TS_MAX_CHUNKS; c++) rows.own_row[c] = w.own_row[c];
        hipLaunchKernelGGL(ht_ts_check_kernel, dim3(tiles.x, (unsigned)(w.n_chunks - 1)), dim3(SEQ_BLOCK), 0, stream, w.real, w.out[0], w.out[1], w.out[2],
                           w.chk[0], w.chk[1], w.chk[2], d, rows, w.tile_rows, w.gate);
    }
    InCols<1> in{{w.real}};
    OutCols<3> out{{w.out[0], w.out[1], w.out[2]}};
    hipLaunchKernelGGL((seq_kernel<HtAll6Op, true>), tiles, dim3(SEQ_LDS_BLOCK), w.lds, stream, w.op, in, out, d, w.gate);
}
static int64_t ht_ts_warm() { const char *e = getenv("PQ_HT_WARM"); const int64_t w = e ? atoll(e) : 640; return w < 64 ? 64 : (w + 15) / 16 * 16; }
// chunks for a series of `len` rows (0: not worth splitting): PQ_HT_CHUNKS (default 4), each owning at least 256 rows
static int ht_ts_chunks(int64_t len) {
    const char *e = getenv("PQ_HT_CHUNKS");
    int64_t c = e ? atoll(e) : 4;
    if (c > HT_TS_MAX_CHUNKS) c = HT_TS_MAX_CHUNKS;
    while (c > 1 && (len + c - 1) / c < 256) c--;
    return c < 2 ? 0 : (int)c;
}
static pq_status ht_all_time_split(pq_ctx *ctx, const pq_batch *b, const HtAll6Op &op6, const double *real, double *const (&o3)[3], int C) {
    constexpr int K = SeqTile<HtAll6Op>::K;
    const int64_t T = b->len, W = ht_ts_warm();
    const int64_t own = ((T + C - 1) / C + 15) / 16 * 16; // whole 128-byte pieces per chunk
    HtTsBlob w{};
    w.op = op6; w.real = real; w.b = *b; w.tile_rows = K; w.lds = (unsigned)seq_lds_bytes(op6);
    for (int k = 0; k < 3; k++) {
        w.out[k] = o3[k];
        w.chk[k] = pq_ws_col(ctx, b, k); // (recording: a fresh column owned by the suite; only the check tiles of it are ever touched)
        if (!w.chk[k]) { pq_set_error("out of device memory for a check column"); return PQ_ERR_NOMEM; }
    }
    const size_t tiles = (size_t)((b->n_series + SEQ_BLOCK - 1) / SEQ_BLOCK);
    w.gate = reinterpret_cast<unsigned *>(rec_alloc_zero(ctx, tiles * sizeof(unsigned)));
    if (!w.gate) { pq_set_error("out of device memory for a gate"); return PQ_ERR_NOMEM; }
    int nc = 0;
    rec_set_shared_out(ctx, true); // the chunk jobs write disjoint rows of the same columns: one phase
    for (int c = 0; c < C; c++) {
        const int64_t t_own = (int64_t)c * own;
        if (t_own >= T) break;
        const int64_t t_end = t_own + own < T ? t_own + own : T, t_b = t_own - (t_own < W ? t_own : W);
        HtAll6Op opc = op6;
        for (int k = 0; k < 3; k++) opc.chk[k] = w.chk[k]; // (base pointers: the job adds its first row, HtAll6Op::ts_shift)
        const double *in[1] = {real};
        void *extra[4] = {op6.der[0], op6.der[1], op6.der[2], c > 0 ? (void *)w.chk[0] : nullptr}; // hazards: the derived columns, the check columns (one key)
        SeqTraits tr{HtAll6Op::SEQ_ID, (int)((double)OpCost<HtAll6Op>::get(opc) * (double)(t_end - t_b) * 1e-3), false, false, seq_lds_bytes(opc),
                     (size_t)SeqTile<HtAll6Op>::BYTES, AlgCols<HtAll6Op>::value, 0.0, {}};
        tr.tile_k = K;
        tr.ts_len = (int)(t_end - t_b); tr.ts_skip = (int)((t_own - t_b) / K); tr.ts_row0 = (int)t_b;
        tr.alg_frac = (double)(t_end - t_own) / (double)T;
        const pq_status st = rec_add_seq(ctx, b, tr, &opc, sizeof opc, in, 1, o3, 3, extra);
        if (st != PQ_OK) { rec_set_shared_out(ctx, false); return st; }
        w.own_row[nc++] = (int)t_own;
    }
    rec_set_shared_out(ctx, false);
    w.n_chunks = nc;
    static_assert(sizeof(HtTsBlob) <= sizeof(RowThunk::blob), "blob too large for a recorded launch");
    RowThunk t{};
    t.launch = &ht_ts_launch;
    t.row_id = 0;
    t.blob_bytes = (int)sizeof w;
    t.dims = dims_of(b);
    memcpy(t.blob, &w, sizeof w);
    t.reads[t.n_reads++] = real;
    for (int k = 0; k < 3; k++) { t.reads[t.n_reads++] = w.out[k]; t.writes[t.n_writes++] = w.out[k]; t.writes[t.n_writes++] = op6.der[k]; }
    t.reads[t.n_reads++] = w.chk[0];
    return rec_add_row(ctx, t);
}

extern "C" {

pq_status pq_trima(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *out) {
    CHK("pq_trima", real && out);
    TrimaOp op{};
    if (p % 2 == 1) { op.k1 = p / 2 + 1; op.k2 = op.k1; } else { op.k1 = p / 2; op.k2 = op.k1 + 1; } // overlap.rs:1313-1326
    InCols<1> in{{real}}; OutCols<1> o{{out}};
    if (p > 0 && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_trima_chain(ctx, b, real, p, out);
}
pq_status pq_apo(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t matype, double *out) {
    CHK("pq_apo", real && out);
    MaDiffOp<0> op{}; op.fast = fast; op.slow = slow; op.matype = matype;
    InCols<1> in{{real}}; OutCols<1> o{{out}};
    if (Ma2::supports(matype) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_apo_chain(ctx, b, real, fast, slow, matype, out);
}
pq_status pq_ppo(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t matype, double *out) {
    CHK("pq_ppo", real && out);
    MaDiffOp<1> op{}; op.fast = fast; op.slow = slow; op.matype = matype;
    InCols<1> in{{real}}; OutCols<1> o{{out}};
    if (Ma2::supports(matype) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_ppo_chain(ctx, b, real, fast, slow, matype, out);
}
pq_status pq_macdext(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t fastmt, int64_t slow,
                     int64_t slowmt, int64_t sig, int64_t sigmt, double *macd, double *signal, double *hist) {
    CHK("pq_macdext", real && macd && signal && hist);
    MacdextOp op{}; op.fast = fast; op.fastmt = fastmt; op.slow = slow; op.slowmt = slowmt; op.sig = sig; op.sigmt = sigmt;
    InCols<1> in{{real}}; OutCols<3> o{{macd, signal, hist}};
    // (a recording of a SMALL shard takes the chain of basic calls: its links are short jobs that the schedule orders by their data
    //  dependencies, suite_launch_small -- 0.7 ms of chain against 1.1 ms of one lone-wave job; the same for the STOCH family below)
    if (PQ_FUSE_OK(ctx) && Ma2::supports(fastmt) && Ma2::supports(slowmt) && Ma2::supports(sigmt) && seq_can_lds(ctx, b, op, in, o))
        return launch_seq(ctx, b, op, in, o);
    return pq_macdext_chain(ctx, b, real, fast, fastmt, slow, slowmt, sig, sigmt, macd, signal, hist);
}
pq_status pq_stoch(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t fastk,
                   int64_t slowk, int64_t slowk_mt, int64_t slowd, int64_t slowd_mt, double *outk, double *outd) {
    CHK("pq_stoch", h && l && c && outk && outd);
    StochOp<0> op{}; op.fastk = fastk; op.p1 = slowk; op.mt1 = slowk_mt; op.p2 = slowd; op.mt2 = slowd_mt;
    InCols<3> in{{h, l, c}}; OutCols<2> o{{outk, outd}};
    if (PQ_FUSE_OK(ctx) && Ma2::supports(slowk_mt) && Ma2::supports(slowd_mt) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_stoch_chain(ctx, b, h, l, c, fastk, slowk, slowk_mt, slowd, slowd_mt, outk, outd);
}
pq_status pq_stochf(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t fastk,
                    int64_t fastd, int64_t fastd_mt, double *outk, double *outd) {
    CHK("pq_stochf", h && l && c && outk && outd);
    StochOp<1> op{}; op.fastk = fastk; op.p1 = fastd; op.mt1 = fastd_mt; op.p2 = 0; op.mt2 = 0;
    InCols<3> in{{h, l, c}}; OutCols<2> o{{outk, outd}};
    if (PQ_FUSE_OK(ctx) && Ma2::supports(fastd_mt) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_stochf_chain(ctx, b, h, l, c, fastk, fastd, fastd_mt, outk, outd);
}
pq_status pq_stochrsi(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, int64_t fastk, int64_t fastd,
                      int64_t fastd_mt, double *outk, double *outd) {
    CHK("pq_stochrsi", real && outk && outd);
    StochRsiOp op{}; op.p = p; op.fastk = fastk; op.fastd = fastd; op.fastd_mt = fastd_mt;
    InCols<1> in{{real}}; OutCols<2> o{{outk, outd}};
    if (PQ_FUSE_OK(ctx) && Ma2::supports(fastd_mt) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_stochrsi_chain(ctx, b, real, p, fastk, fastd, fastd_mt, outk, outd);
}
pq_status pq_cci(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *out) {
    CHK("pq_cci", h && l && c && out);
    CciOp op{}; op.p = p;
    InCols<3> in{{h, l, c}}; OutCols<1> o{{out}};
    if (seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_cci_chain(ctx, b, h, l, c, p, out);
}
pq_status pq_adxr(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *out) {
    CHK("pq_adxr", h && l && c && out);
    { pq_status st; if (wt_dmi(ctx, b, h, l, c, p, nullptr, nullptr, nullptr, nullptr, out, &st)) return st; }
    DmAllOp<false> op{}; op.p = p;
    InCols<3> in{{h, l, c}}; OutCols<1> o{{out}};
    if (seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    return pq_adxr_chain(ctx, b, h, l, c, p, out);
}
// calc_dm (momentum.rs:668-727) evaluated once for all five of its users
pq_status pq_dmi_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p,
                     double *dx, double *plus_di, double *minus_di, double *adx, double *adxr) {
    CHK("pq_dmi_all", h && l && c && dx && plus_di && minus_di && adx && adxr);
    { pq_status st; if (wt_dmi(ctx, b, h, l, c, p, dx, plus_di, minus_di, adx, adxr, &st)) return st; }
    DmAllOp<true> op{}; op.p = p;
    InCols<3> in{{h, l, c}}; OutCols<5> o{{dx, plus_di, minus_di, adx, adxr}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_dx(ctx, b, h, l, c, p, dx));
    PQ_TRY(pq_plus_di(ctx, b, h, l, c, p, plus_di));
    PQ_TRY(pq_minus_di(ctx, b, h, l, c, p, minus_di));
    PQ_TRY(pq_adx(ctx, b, h, l, c, p, adx));
    if (ctx->rec && ctx->rec_small) // ADXR is a row-wise function of the ADX column just written (momentum.rs:61): no second walk for it
        return pq_adxr_from_adx(ctx, b, adx, p, adxr);
    return pq_adxr_chain(ctx, b, h, l, c, p, adxr);
}
// the shared Hilbert pipeline (cycle.rs:27-63) evaluated once for ht_dcperiod / ht_dcphase / ht_phasor / ht_sine
pq_status pq_ht_all(pq_ctx *ctx, const pq_batch *b, const double *real, double *dcperiod, double *dcphase, double *inphase,
                    double *quadrature, double *sine, double *leadsine) {
    CHK("pq_ht_all", real && dcperiod && dcphase && inphase && quadrature && sine && leadsine);
    {   // tiled body: one job, the derived columns leave through its storer wave
        HtAll6Op op6{};
        op6.der[0] = dcphase; op6.der[1] = sine; op6.der[2] = leadsine;
        InCols<1> in{{real}}; OutCols<3> o3{{dcperiod, inphase, quadrature}};
        const double *nodep[1] = {real};
        double *ders[3] = {dcphase, sine, leadsine};
        if (b->len % SeqTile<HtAll6Op>::K == 0 && seq_can_lds(ctx, b, op6, in, o3) && seq_cols_aligned<1, 3>(b, nodep, ders)) {
            const int chunks = (ctx->rec && ctx->rec_small && !getenv("PQ_NO_HT_SPLIT") && seq_cols_tiling<1, 3>(b, in.p, o3.p) == 0) ? ht_ts_chunks(b->len) : 0;
            if (chunks) return ht_all_time_split(ctx, b, op6, real, o3.p, chunks); // a small shard: the job split in time (above)
            return launch_seq(ctx, b, op6, in, o3);
        }
    }
    PQ_TRY(launch_seq(ctx, b, HtAllOp{}, InCols<1>{{real}}, OutCols<3>{{dcperiod, inphase, quadrature}}));
    // (in a recorded suite the ROW launch reads what the job writes, so it lands in the next phase)
    return launch_row(ctx, b, HtPhaseSineOp{}, InCols<2>{{inphase, quadrature}}, OutColsT<HtPhaseSineOp, double>{{dcphase, sine, leadsine}});
}

// ---- multi-output forms: several reference functions over the same inputs as ONE job (bit-identical columns) ----
pq_status pq_ema_all(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *ema, double *dema, double *tema, double *trix) {
    CHK("pq_ema_all", real && ema && dema && tema && trix);
    { pq_status st; if (wt_ema_all(ctx, b, real, p, ema, dema, tema, trix, &st)) return st; } // one symbol per wavefront (ops_wt.h)
    EmaAllOp op{};
    op.a.a.p = p; op.a.b.p = p; op.b.a.p = p; op.b.b.p = p;
    InCols<1> in{{real}}; OutCols<4> o{{ema, dema, tema, trix}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_ema(ctx, b, real, p, ema)); PQ_TRY(pq_dema(ctx, b, real, p, dema)); PQ_TRY(pq_tema(ctx, b, real, p, tema));
    return pq_trix(ctx, b, real, p, trix);
}
pq_status pq_atr_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *atr, double *natr) {
    CHK("pq_atr_all", h && l && c && atr && natr);
    { pq_status st; if (wt_atr(ctx, b, h, l, c, p, atr, natr, &st)) return st; }
    AtrAllOp op{}; op.a.p = p; op.b.p = p;
    InCols<3> in{{h, l, c}}; OutCols<2> o{{atr, natr}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_atr(ctx, b, h, l, c, p, atr));
    return pq_natr(ctx, b, h, l, c, p, natr);
}
pq_status pq_dm_pair(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, int64_t p, double *plus_dm, double *minus_dm) {
    CHK("pq_dm_pair", h && l && plus_dm && minus_dm);
    { pq_status st; if (wt_dm_pair(ctx, b, h, l, p, plus_dm, minus_dm, &st)) return st; }
    DmPairOp op{}; op.a.p = p; op.b.p = p;
    InCols<2> in{{h, l}}; OutCols<2> o{{plus_dm, minus_dm}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_plus_dm(ctx, b, h, l, p, plus_dm));
    return pq_minus_dm(ctx, b, h, l, p, minus_dm);
}
pq_status pq_ad_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, const double *v, int64_t fast,
                    int64_t slow, double *ad, double *adosc) {
    CHK("pq_ad_all", h && l && c && v && ad && adosc);
    AdAllOp op{}; op.a.fast = op.a.slow = 0; op.b.fast = fast; op.b.slow = slow;
    InCols<4> in{{h, l, c, v}}; OutCols<2> o{{ad, adosc}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_ad(ctx, b, h, l, c, v, ad));
    return pq_adosc(ctx, b, h, l, c, v, fast, slow, adosc);
}
pq_status pq_macd_pair(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t sig, int64_t fix_sig,
                       double *macd, double *signal, double *hist, double *fmacd, double *fsignal, double *fhist) {
    CHK("pq_macd_pair", real && macd && signal && hist && fmacd && fsignal && fhist);
    { pq_status st; if (wt_macd(ctx, b, real, fast, slow, sig, fix_sig, macd, signal, hist, fmacd, fsignal, fhist, &st)) return st; }
    MacdPairOp op{};
    op.a.fast = fast; op.a.slow = slow; op.a.sig = sig; op.b.fast = 12; op.b.slow = 26; op.b.sig = fix_sig; // momentum.py:90-92
    InCols<1> in{{real}}; OutCols<6> o{{macd, signal, hist, fmacd, fsignal, fhist}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_macd(ctx, b, real, fast, slow, sig, macd, signal, hist));
    return pq_macdfix(ctx, b, real, fix_sig, fmacd, fsignal, fhist);
}
pq_status pq_sar_pair(pq_ctx *ctx, const pq_batch *b, const double *high, const double *low, double accel, double maxv, double startvalue,
                      double offsetonreverse, double ai_long, double a_long, double am_long, double ai_short, double a_short,
                      double am_short, double *sar, double *sarext) {
    CHK("pq_sar_pair", high && low && sar && sarext);
    SarPairOp op{};
    op.a.ext = false; op.a.startvalue = 0.0; op.a.offset = 0.0;
    op.a.ai_long = op.a.a_long = op.a.ai_short = op.a.a_short = accel; op.a.am_long = op.a.am_short = maxv;
    op.b.ext = true; op.b.startvalue = startvalue; op.b.offset = offsetonreverse;
    op.b.ai_long = ai_long; op.b.a_long = a_long; op.b.am_long = am_long; op.b.ai_short = ai_short; op.b.a_short = a_short; op.b.am_short = am_short;
    InCols<2> in{{high, low}}; OutCols<2> o{{sar, sarext}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_sar(ctx, b, high, low, accel, maxv, sar));
    return pq_sarext(ctx, b, high, low, startvalue, offsetonreverse, ai_long, a_long, am_long, ai_short, a_short, am_short, sarext);
}
pq_status pq_volume_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, const double *v, int64_t mfi_p,
                        int64_t fast, int64_t slow, double *mfi, double *ad, double *adosc, double *obv) {
    CHK("pq_volume_all", h && l && c && v && mfi && ad && adosc && obv);
    VolumeAllOp op{};
    op.a.p = mfi_p; op.b.a.a.fast = op.b.a.a.slow = 0; op.b.a.b.fast = fast; op.b.a.b.slow = slow;
    InCols<4> in{{h, l, c, v}}; OutCols<4> o{{mfi, ad, adosc, obv}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_mfi(ctx, b, h, l, c, v, mfi_p, mfi)); PQ_TRY(pq_ad(ctx, b, h, l, c, v, ad));
    PQ_TRY(pq_adosc(ctx, b, h, l, c, v, fast, slow, adosc));
    return pq_obv(ctx, b, c, v, obv);
}
pq_status pq_dm_system_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p, double *dx,
                           double *plus_di, double *minus_di, double *adx, double *adxr, double *atr, double *natr) {
    CHK("pq_dm_system_all", h && l && c && dx && plus_di && minus_di && adx && adxr && atr && natr);
    {   // wave-per-symbol form: two jobs (the DI / DX / ADX chains need three LDS columns, ATR / NATR one)
        pq_status st;
        if (wt_dmi(ctx, b, h, l, c, p, dx, plus_di, minus_di, adx, adxr, &st)) {
            PQ_TRY(st);
            return pq_atr_all(ctx, b, h, l, c, p, atr, natr);
        }
    }
    DmiAtrOp op{}; op.a.p = p; op.b.a.p = p; op.b.b.p = p;
    InCols<3> in{{h, l, c}}; OutCols<7> o{{dx, plus_di, minus_di, adx, adxr, atr, natr}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_dmi_all(ctx, b, h, l, c, p, dx, plus_di, minus_di, adx, adxr));
    return pq_atr_all(ctx, b, h, l, c, p, atr, natr);
}
pq_status pq_sma_ma(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *sma, double *ma) {
    CHK("pq_sma_ma", real && sma && ma);
    SmaDupOp op{}; op.a.p = p;
    InCols<1> in{{real}}; OutCols<2> o{{sma, ma}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_sma(ctx, b, real, p, sma));
    return pq_ma(ctx, b, real, p, 0, ma);
}
pq_status pq_cmo_rsi(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t p, double *cmo, double *rsi) {
    CHK("pq_cmo_rsi", real && cmo && rsi);
    {   // RSI's Wilder averages contract (wave-per-symbol form); CMO's rolling sums do not: it stays a lane-per-symbol job
        pq_status st;
        if (wt_rsi(ctx, b, real, p, rsi, &st)) { PQ_TRY(st); return pq_cmo(ctx, b, real, p, cmo); }
    }
    CmoRsiOp op{}; op.a.p = p; op.b.p = p;
    InCols<1> in{{real}}; OutCols<2> o{{cmo, rsi}};
    if (PQ_FUSE_OK(ctx) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_cmo(ctx, b, real, p, cmo));
    return pq_rsi(ctx, b, real, p, rsi);
}
pq_status pq_stoch_all(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t fastk, int64_t slowk,
                       int64_t slowk_mt, int64_t slowd, int64_t slowd_mt, int64_t fastd, int64_t fastd_mt, double *slowk_out,
                       double *slowd_out, double *fastk_out, double *fastd_out) {
    CHK("pq_stoch_all", h && l && c && slowk_out && slowd_out && fastk_out && fastd_out);
    StochAllOp op{};
    op.fastk = fastk; op.slowk = slowk; op.slowk_mt = slowk_mt; op.slowd = slowd; op.slowd_mt = slowd_mt; op.fastd = fastd; op.fastd_mt = fastd_mt;
    InCols<3> in{{h, l, c}}; OutCols<4> o{{slowk_out, slowd_out, fastk_out, fastd_out}};
    if (PQ_FUSE_OK(ctx) && Ma2::supports(slowk_mt) && Ma2::supports(slowd_mt) && Ma2::supports(fastd_mt) && seq_can_lds(ctx, b, op, in, o))
        return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_stoch(ctx, b, h, l, c, fastk, slowk, slowk_mt, slowd, slowd_mt, slowk_out, slowd_out));
    return pq_stochf(ctx, b, h, l, c, fastk, fastd, fastd_mt, fastk_out, fastd_out);
}
pq_status pq_apo_ppo(pq_ctx *ctx, const pq_batch *b, const double *real, int64_t fast, int64_t slow, int64_t matype, double *apo, double *ppo) {
    CHK("pq_apo_ppo", real && apo && ppo);
    ApoPpoOp op{};
    op.a.fast = fast; op.a.slow = slow; op.a.matype = matype; op.b.fast = fast; op.b.slow = slow; op.b.matype = matype;
    InCols<1> in{{real}}; OutCols<2> o{{apo, ppo}};
    if (PQ_FUSE_OK(ctx) && Ma2::supports(matype) && seq_can_lds(ctx, b, op, in, o)) return launch_seq(ctx, b, op, in, o);
    PQ_TRY(pq_apo(ctx, b, real, fast, slow, matype, apo));
    return pq_ppo(ctx, b, real, fast, slow, matype, ppo);
}

} // extern "C"
