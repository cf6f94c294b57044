// misc.hip -- kernels + C ABI for volatility / volume / price transforms and the Hilbert-transform
// cycle indicators (reference: src/talib/{volatility,volume,price,cycle}.rs) plus MAMA (D-4).
#include "ops_misc.h"
#include "wt_api.h"

// ---------------------------------------------------------------- MAMA at fastlimit == 0 (the reference wrapper's default)
// MAMA(real, fastlimit = 0.0, slowlimit = 0.0) is what `python/polars_quant/talib/overlap.py:66-76` calls by default, and by D-4
// (oracle/overlap.c pqo_mama) alpha = clamp(fastlimit / max(prev_phase - phase, 1), slowlimit, fastlimit).  With fastlimit == 0.0 that is
// +-0 on every row whose phase is not NaN, whatever slowlimit is, so
//     mama = (+-0) * x + 1 * mama = +0.0      fama = (+-0) * mama + 1 * fama = +0.0        (both start at +0.0)
// for as long as every x (nulls count as 0.0, overlap.rs:161-170) is finite and small enough for the Hilbert pipeline to stay finite:
// its largest intermediates are products of two terms below 40 |x| (re, im: cycle.rs:53-54), so |x| <= 1e140 keeps everything finite, hence
// the phase (an atan of a non-NaN ratio) finite and alpha = +-0.  Then the whole 2 520-row walk of the pipeline (~270 dependent f64
// instructions per row, the longest job of a suite step) decides nothing: one row-parallel kernel writes null (rows < 31, or a series
// shorter than 32 rows) / +0.0 and checks the bound; a 64-series tile that holds any larger, infinite or NaN value is flagged and walked
// by the general op behind it, gated by the flag (from the first such row on the reference's values are NaNs whose payloads the walk
// decides).  Bit-identical to the walk in every case; any other fastlimit takes the walk directly.
constexpr double MAMA_FINITE_BOUND = 1e140;
__global__ __launch_bounds__(ROW_BLOCK) void mama_zero_kernel(const double *real, double *mama, double *fama, Dims d, unsigned *gate) {
    const int64_t s = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * ROW_BLOCK + threadIdx.x;
    const int64_t sbase = dims_base(d, s), slen = dims_len(d, s);
    if (t >= slen) return;
    const double v = n0m(real[sbase + t]);
    if (!(fabs(v) <= MAMA_FINITE_BOUND)) atomicOr(&gate[s / SEQ_BLOCK], 1u); // (NaN fails the compare too)
    const double y = (slen < 32 || t < 31) ? pq_null() : 0.0;
    __builtin_nontemporal_store(y, &mama[sbase + t]);
    __builtin_nontemporal_store(y, &fama[sbase + t]);
}
struct MamaZeroBlob {
    HtOp<4> op;
    const double *real;
    double *mama, *fama;
    pq_batch b;
    unsigned *gate;
    unsigned lds;   // of the tiled general body; 0: the per-lane body stands behind (ragged batch, rows not 16-byte aligned, direct call)
};
static void mama_zero_launch(const void *blob, hipStream_t stream) {
    const MamaZeroBlob &w = *reinterpret_cast<const MamaZeroBlob *>(blob);
    const pq_batch *b = &w.b;
    const Dims d = dims_of(b);
    for (int64_t s0 = 0; s0 < b->n_series; s0 += 65472) { // grid.y is limited to 65535: slices of whole 64-series tiles
        const int64_t ns = b->n_series - s0 < 65472 ? b->n_series - s0 : 65472;
        Dims ds{ns, d.len, d.stride, d.offs ? d.offs + s0 : nullptr};
        const int64_t off = d.offs ? 0 : s0 * d.stride;
        hipLaunchKernelGGL(mama_zero_kernel, dim3((unsigned)((b->len + ROW_BLOCK - 1) / ROW_BLOCK), (unsigned)ns), dim3(ROW_BLOCK), 0, stream, w.real + off,
                           w.mama + off, w.fama + off, ds, w.gate + s0 / SEQ_BLOCK);
    }
    const dim3 tiles((unsigned)((b->n_series + SEQ_BLOCK - 1) / SEQ_BLOCK));
    InCols<1> in{{w.real}};
    OutCols<2> out{{w.mama, w.fama}};
    if (w.lds) hipLaunchKernelGGL((seq_kernel<HtOp<4>, true>), tiles, dim3(SEQ_LDS_BLOCK), w.lds, stream, w.op, in, out, d, w.gate);
    else hipLaunchKernelGGL((seq_kernel<HtOp<4>, false>), tiles, dim3(SEQ_BLOCK), 0, stream, w.op, in, out, d, w.gate);
}
static pq_status mama_zero(pq_ctx *ctx, const pq_batch *b, const HtOp<4> &op, const double *real, double *mama, double *fama) {
    if (b->n_series == 0 || b->len == 0) return PQ_OK;
    MamaZeroBlob w{};
    w.op = op; w.real = real; w.mama = mama; w.fama = fama; w.b = *b;
    const double *in[1] = {real};
    double *out[2] = {mama, fama};
    const size_t tiles = (size_t)((b->n_series + SEQ_BLOCK - 1) / SEQ_BLOCK);
    // the general body behind the gate is the per-lane one (register delay lines, no LDS; what a direct call of the Hilbert family runs
    // anyway, DIRECT_LANE_MAX): a workgroup of the tiled body asks for 20 KB of LDS just to look at its flag and return, and beside a
    // job grid that fills the chip's LDS it waited 0.16 ms for a CU to have them (1 250 symbols); a flagged tile is rare, its speed immaterial
    (void)in; (void)out;
    w.lds = 0u;
    if (ctx->rec) {
        static_assert(sizeof(MamaZeroBlob) <= sizeof(RowThunk::blob), "blob too large for a recorded launch");
        w.gate = reinterpret_cast<unsigned *>(rec_alloc_zero(ctx, tiles * sizeof(unsigned)));
        if (!w.gate) { pq_set_error("out of device memory for a gate"); return PQ_ERR_NOMEM; }
        RowThunk t{};
        t.launch = &mama_zero_launch;
        t.row_id = 0;
        t.blob_bytes = (int)sizeof w;
        t.dims = dims_of(b);
        memcpy(t.blob, &w, sizeof w);
        t.reads[t.n_reads++] = real;
        t.writes[t.n_writes++] = mama;
        t.writes[t.n_writes++] = fama;
        return rec_add_row(ctx, t);
    }
    PQ_TRY(ctx_gate(ctx, tiles, &w.gate));
    mama_zero_launch(&w, ctx->stream);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}

// ---------------------------------------------------------------- C ABI
#define CHK(name, cond) PQ_TRY(pq_check(ctx, b)); PQ_REQUIRE(cond, name ": null pointer")
extern "C" {

pq_status pq_avgprice(pq_ctx *ctx, const pq_batch *b, const double *o, const double *h, const double *l,
                      const double *c, double *out) {
    CHK("pq_avgprice", o && h && l && c && out);
    return launch_row(ctx, b, PriceOp<0>{}, InCols<4>{{o, h, l, c}}, OutColsT<PriceOp<0>, double>{{out}});
}
pq_status pq_medprice(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, double *out) {
    CHK("pq_medprice", h && l && out);
    return launch_row(ctx, b, PriceOp<1>{}, InCols<2>{{h, l}}, OutColsT<PriceOp<1>, double>{{out}});
}
pq_status pq_typprice(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, double *out) {
    CHK("pq_typprice", h && l && c && out);
    return launch_row(ctx, b, PriceOp<2>{}, InCols<3>{{h, l, c}}, OutColsT<PriceOp<2>, double>{{out}});
}
pq_status pq_wclprice(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, double *out) {
    CHK("pq_wclprice", h && l && c && out);
    return launch_row(ctx, b, PriceOp<3>{}, InCols<3>{{h, l, c}}, OutColsT<PriceOp<3>, double>{{out}});
}
pq_status pq_trange(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, double *out) {
    CHK("pq_trange", h && l && c && out);
    return launch_row(ctx, b, TrangeOp{}, InCols<3>{{h, l, c}}, OutColsT<TrangeOp, double>{{out}});
}
pq_status pq_atr(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p,
                 double *out) {
    CHK("pq_atr", h && l && c && out);
    { pq_status st; if (wt_atr(ctx, b, h, l, c, p, out, nullptr, &st)) return st; }
    AtrOp<false> op{}; op.p = p;
    return launch_seq(ctx, b, op, InCols<3>{{h, l, c}}, OutCols<1>{{out}});
}
pq_status pq_natr(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, int64_t p,
                  double *out) {
    CHK("pq_natr", h && l && c && out);
    { pq_status st; if (wt_atr(ctx, b, h, l, c, p, nullptr, out, &st)) return st; }
    AtrOp<true> op{}; op.p = p;
    return launch_seq(ctx, b, op, InCols<3>{{h, l, c}}, OutCols<1>{{out}});
}
pq_status pq_ad(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, const double *v,
                double *out) {
    CHK("pq_ad", h && l && c && v && out);
    AdOp<false> op{}; op.fast = op.slow = 0;
    return launch_seq(ctx, b, op, InCols<4>{{h, l, c, v}}, OutCols<1>{{out}});
}
pq_status pq_adosc(pq_ctx *ctx, const pq_batch *b, const double *h, const double *l, const double *c, const double *v,
                   int64_t fast, int64_t slow, double *out) {
    CHK("pq_adosc", h && l && c && v && out);
    AdOp<true> op{}; op.fast = fast; op.slow = slow;
    return launch_seq(ctx, b, op, InCols<4>{{h, l, c, v}}, OutCols<1>{{out}});
}
pq_status pq_obv(pq_ctx *ctx, const pq_batch *b, const double *c, const double *v, double *out) {
    CHK("pq_obv", c && v && out);
    return launch_seq(ctx, b, ObvOp{}, InCols<2>{{c, v}}, OutCols<1>{{out}});
}
pq_status pq_ht_dcperiod(pq_ctx *ctx, const pq_batch *b, const double *real, double *out) {
    CHK("pq_ht_dcperiod", real && out);
    return launch_seq(ctx, b, HtOp<0>{}, InCols<1>{{real}}, OutCols<1>{{out}});
}
pq_status pq_ht_dcphase(pq_ctx *ctx, const pq_batch *b, const double *real, double *out) {
    CHK("pq_ht_dcphase", real && out);
    return launch_seq(ctx, b, HtOp<1>{}, InCols<1>{{real}}, OutCols<1>{{out}});
}
pq_status pq_ht_phasor(pq_ctx *ctx, const pq_batch *b, const double *real, double *inphase, double *quadrature) {
    CHK("pq_ht_phasor", real && inphase && quadrature);
    return launch_seq(ctx, b, HtOp<2>{}, InCols<1>{{real}}, OutCols<2>{{inphase, quadrature}});
}
pq_status pq_ht_sine(pq_ctx *ctx, const pq_batch *b, const double *real, double *sine, double *leadsine) {
    CHK("pq_ht_sine", real && sine && leadsine);
    return launch_seq(ctx, b, HtOp<3>{}, InCols<1>{{real}}, OutCols<2>{{sine, leadsine}});
}
pq_status pq_mama(pq_ctx *ctx, const pq_batch *b, const double *real, double fastlimit, double slowlimit,
                  double *mama, double *fama) {
    CHK("pq_mama", real && mama && fama);
    HtOp<4> op{}; op.fastlimit = fastlimit; op.slowlimit = slowlimit;
    if (fastlimit == 0.0 && !getenv("PQ_MAMA_WALK")) return mama_zero(ctx, b, op, real, mama, fama); // alpha == +-0 on every row: see mama_zero_kernel
    return launch_seq(ctx, b, op, InCols<1>{{real}}, OutCols<2>{{mama, fama}});
}
pq_status pq_ht_trendline(pq_ctx *ctx, const pq_batch *b, const double *real, double *out) {
    CHK("pq_ht_trendline", real && out);
    return launch_row(ctx, b, TrendlineOp{}, InCols<1>{{real}}, OutColsT<TrendlineOp, double>{{out}});
}
pq_status pq_ht_trendmode(pq_ctx *ctx, const pq_batch *b, const double *real, int32_t *out) {
    CHK("pq_ht_trendmode", real && out);
    return launch_row(ctx, b, TrendmodeOp{}, InCols<1>{{real}}, OutColsT<TrendmodeOp, int32_t>{{out}});
}

} // extern "C"
