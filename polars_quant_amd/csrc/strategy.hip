// strategy.hip -- the comparisons that turn indicator columns into buy / sell signals for the README's `Strategy` generators
// (README.md:862-994; README-only in the reference, decisions D-11 / D-11b in DESIGN.md) beyond the three D-11 rules of backtest.hip:
// threshold / column gates on existing signals (MA trend / slope / distance filters, stochastic zones, ADX strength), the z-score
// and percentage-band columns, volume surges, opening gaps, pattern any-of, and the MA-stack trend with its first-bar edge.
// ROW shape: one (series, row) per thread, rows fastest; recordable into a suite in front of pq_backtest_vectorized (the signal
// columns then never leave the device between the indicator kernels and the backtest).
#include "pq_dev.h"

namespace {
constexpr int RULE_MAX = 16;
struct GateOp { // signals in -> gated signals out (in place allowed)
    const double *a, *c;
    const uint8_t *bin, *sin;
    uint8_t *bout, *sout;
    int mode;
    double k0, k1;
    __device__ void eval(int64_t o, int64_t i, int64_t) const {
        bool b = bin[o + i] != 0, s = sin[o + i] != 0;
        const double x = a[o + i];
        switch (mode) {
        case 0: b = b && (x < k0); s = s && (x > k1); break;                          // zones: buys below k0, sells above k1
        case 1: { const bool g = x > k0; b = b && g; s = s && g; } break;             // strength: both only while a > k0
        case 2: b = b && (x > c[o + i]); break;                                       // buys only while a > c (trend filter)
        case 3: b = b && i > 0 && (x > a[o + i - 1]); break;                          // buys only while a rises
        default: { const double y = c[o + i]; b = b && ((x - y) > fabs(y) * k0); } break; // buys only while a is k0 * |c| above c
        }
        bout[o + i] = b; sout[o + i] = s; // (a NULL is a NaN: every comparison with it is false)
    }
};
struct ZscoreOp { // z = (price - mid) / (upper - mid), NULL where the band is (upper - mid = 1 x the population deviation)
    const double *p, *u, *m;
    double *z;
    __device__ void eval(int64_t o, int64_t i, int64_t) const {
        const double uu = u[o + i];
        z[o + i] = pq_isnull(uu) ? uu : (p[o + i] - m[o + i]) / (uu - m[o + i]);
    }
};
struct ScaleBandOp { // lo = base * f_lo, hi = base * f_hi, NULL where base is
    const double *base;
    double f_lo, f_hi;
    double *lo, *hi;
    __device__ void eval(int64_t o, int64_t i, int64_t) const {
        const double b = base[o + i];
        const bool n = pq_isnull(b);
        lo[o + i] = n ? b : b * f_lo;
        hi[o + i] = n ? b : b * f_hi;
    }
};
struct VolumeSurgeOp { // volume above multiplier x its average on an up day buys, on a down day sells
    const double *v, *sv, *c;
    double mult;
    uint8_t *buy, *sell;
    __device__ void eval(int64_t o, int64_t i, int64_t) const {
        const bool surge = v[o + i] > mult * sv[o + i];
        const bool up = i > 0 && c[o + i] > c[o + i - 1], dn = i > 0 && c[o + i] < c[o + i - 1];
        buy[o + i] = surge && up; sell[o + i] = surge && dn;
    }
};
struct GapOp { // an open above the previous high x f_up buys, below the previous low x f_dn sells
    const double *op, *h, *l;
    double f_up, f_dn;
    uint8_t *buy, *sell;
    __device__ void eval(int64_t o, int64_t i, int64_t) const {
        buy[o + i] = i > 0 && op[o + i] > h[o + i - 1] * f_up;
        sell[o + i] = i > 0 && op[o + i] < l[o + i - 1] * f_dn;
    }
};
struct PatternAnyOp { // any of the bullish recognisers at +100 buys, any of the bearish ones at -100 sells
    const int32_t *bull[RULE_MAX], *bear[RULE_MAX];
    int nb, ns;
    uint8_t *buy, *sell;
    __device__ void eval(int64_t o, int64_t i, int64_t) const {
        bool b = false, s = false;
        for (int k = 0; k < nb; k++) b |= bull[k][o + i] == 100;
        for (int k = 0; k < ns; k++) s |= bear[k][o + i] == -100;
        buy[o + i] = b; sell[o + i] = s;
    }
};
struct MaStackOp { // buy on the first bar where ma[0] > ma[1] > ... holds, sell on the first bar of the reverse order
    const double *ma[RULE_MAX];
    int n;
    uint8_t *buy, *sell;
    __device__ void order(int64_t q, bool &bull, bool &bear) const {
        bull = true; bear = true;
        for (int k = 0; k + 1 < n; k++) { const double x = ma[k][q], y = ma[k + 1][q]; bull = bull && x > y; bear = bear && x < y; }
    }
    __device__ void eval(int64_t o, int64_t i, int64_t) const {
        bool b1, s1, b0 = true, s0 = true;
        order(o + i, b1, s1);
        if (i > 0) order(o + i - 1, b0, s0);
        buy[o + i] = i > 0 && b1 && !b0; sell[o + i] = i > 0 && s1 && !s0;
    }
};

template <class Op>
__global__ __launch_bounds__(ROW_BLOCK) void rule_kernel(Op op, Dims d) {
    const int64_t s = (int64_t)blockIdx.z * 65535 + blockIdx.y, t = (int64_t)blockIdx.x * ROW_BLOCK + threadIdx.x;
    if (s >= d.n) return;
    const int64_t len = dims_len(d, s);
    if (t < len) op.eval(dims_base(d, s), t, len);
}
template <class Op>
struct RuleBlob { Op op; pq_batch b; };
template <class Op>
void rule_launch_blob(const void *blob, hipStream_t stream) {
    const RuleBlob<Op> &rb = *reinterpret_cast<const RuleBlob<Op> *>(blob);
    const pq_batch *b = &rb.b;
    const unsigned gy = (unsigned)(b->n_series < 65535 ? b->n_series : 65535), gz = (unsigned)((b->n_series + 65534) / 65535);
    hipLaunchKernelGGL(rule_kernel<Op>, dim3((unsigned)((b->len + ROW_BLOCK - 1) / ROW_BLOCK), gy, gz), dim3(ROW_BLOCK), 0, stream, rb.op, dims_of(b));
}
template <class Op>
pq_status rule_launch(pq_ctx *ctx, const pq_batch *b, const Op &op, std::initializer_list<const void *> reads, std::initializer_list<void *> writes) {
    if (b->n_series == 0 || b->len == 0) return PQ_OK;
    RuleBlob<Op> rb{op, *b};
    if (ctx->rec) {
        static_assert(sizeof(RuleBlob<Op>) <= sizeof(RowThunk::blob), "rule blob too large");
        RowThunk t{};
        t.launch = &rule_launch_blob<Op>;
        t.blob_bytes = (int)sizeof rb;
        t.dims = dims_of(b);
        memcpy(t.blob, &rb, sizeof rb);
        for (const void *p : reads) if (p) t.reads[t.n_reads++] = p;
        for (void *p : writes) if (p) t.writes[t.n_writes++] = p;
        return rec_add_row(ctx, t);
    }
    rule_launch_blob<Op>(&rb, ctx->stream);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}
} // namespace

extern "C" {

pq_status pq_gate_signals(pq_ctx *ctx, const pq_batch *b, const double *a, const double *c, int32_t mode, double k0, double k1,
                          const uint8_t *buy_in, const uint8_t *sell_in, uint8_t *buy, uint8_t *sell) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(a && buy_in && sell_in && buy && sell, "pq_gate_signals: null pointer");
    PQ_REQUIRE(mode >= 0 && mode <= 4, "pq_gate_signals: mode must be 0 (zones), 1 (strength), 2 (above column), 3 (rising) or 4 (distance)");
    PQ_REQUIRE(c || (mode != 2 && mode != 4), "pq_gate_signals: this mode compares with a second column");
    GateOp op{a, c, buy_in, sell_in, buy, sell, mode, k0, k1};
    return rule_launch(ctx, b, op, {a, c, buy_in, sell_in}, {buy, sell});
}
pq_status pq_zscore(pq_ctx *ctx, const pq_batch *b, const double *price, const double *upper, const double *mid, double *z) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(price && upper && mid && z, "pq_zscore: null pointer");
    ZscoreOp op{price, upper, mid, z};
    return rule_launch(ctx, b, op, {price, upper, mid}, {z});
}
pq_status pq_scale_band(pq_ctx *ctx, const pq_batch *b, const double *base, double f_lo, double f_hi, double *lo, double *hi) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(base && lo && hi, "pq_scale_band: null pointer");
    ScaleBandOp op{base, f_lo, f_hi, lo, hi};
    return rule_launch(ctx, b, op, {base}, {lo, hi});
}
pq_status pq_volume_surge_signals(pq_ctx *ctx, const pq_batch *b, const double *volume, const double *avg_volume, const double *close,
                                  double multiplier, uint8_t *buy, uint8_t *sell) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(volume && avg_volume && close && buy && sell, "pq_volume_surge_signals: null pointer");
    VolumeSurgeOp op{volume, avg_volume, close, multiplier, buy, sell};
    return rule_launch(ctx, b, op, {volume, avg_volume, close}, {buy, sell});
}
pq_status pq_gap_signals(pq_ctx *ctx, const pq_batch *b, const double *open, const double *high, const double *low, double f_up, double f_dn,
                         uint8_t *buy, uint8_t *sell) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(open && high && low && buy && sell, "pq_gap_signals: null pointer");
    GapOp op{open, high, low, f_up, f_dn, buy, sell};
    return rule_launch(ctx, b, op, {open, high, low}, {buy, sell});
}
pq_status pq_pattern_any_signals(pq_ctx *ctx, const pq_batch *b, const int32_t *const *bullish, int32_t n_bullish, const int32_t *const *bearish,
                                 int32_t n_bearish, uint8_t *buy, uint8_t *sell) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(buy && sell && (bullish || n_bullish == 0) && (bearish || n_bearish == 0), "pq_pattern_any_signals: null pointer");
    PQ_REQUIRE(n_bullish >= 0 && n_bullish <= RULE_MAX && n_bearish >= 0 && n_bearish <= RULE_MAX, "pq_pattern_any_signals: at most 16 recognisers per side");
    PatternAnyOp op{};
    op.nb = n_bullish; op.ns = n_bearish; op.buy = buy; op.sell = sell;
    if (b->n_series == 0 || b->len == 0) return PQ_OK;
    RuleBlob<PatternAnyOp> rb{op, *b};
    for (int k = 0; k < n_bullish; k++) { PQ_REQUIRE(bullish[k], "pq_pattern_any_signals: null column"); rb.op.bull[k] = bullish[k]; }
    for (int k = 0; k < n_bearish; k++) { PQ_REQUIRE(bearish[k], "pq_pattern_any_signals: null column"); rb.op.bear[k] = bearish[k]; }
    if (ctx->rec) {
        RowThunk t{};
        t.launch = &rule_launch_blob<PatternAnyOp>;
        t.blob_bytes = (int)sizeof rb;
        t.dims = dims_of(b);
        memcpy(t.blob, &rb, sizeof rb);
        for (int k = 0; k < n_bullish; k++) t.reads[t.n_reads++] = bullish[k];
        for (int k = 0; k < n_bearish; k++) t.reads[t.n_reads++] = bearish[k];
        t.writes[t.n_writes++] = buy; t.writes[t.n_writes++] = sell;
        return rec_add_row(ctx, t);
    }
    rule_launch_blob<PatternAnyOp>(&rb, ctx->stream);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}
pq_status pq_ma_stack_signals(pq_ctx *ctx, const pq_batch *b, const double *const *mas, int32_t n, uint8_t *buy, uint8_t *sell) {
    PQ_TRY(pq_check(ctx, b));
    PQ_REQUIRE(mas && buy && sell, "pq_ma_stack_signals: null pointer");
    PQ_REQUIRE(n >= 2 && n <= RULE_MAX, "pq_ma_stack_signals: 2 .. 16 moving averages");
    if (b->n_series == 0 || b->len == 0) return PQ_OK;
    RuleBlob<MaStackOp> rb{};
    rb.b = *b; rb.op.n = n; rb.op.buy = buy; rb.op.sell = sell;
    for (int k = 0; k < n; k++) { PQ_REQUIRE(mas[k], "pq_ma_stack_signals: null column"); rb.op.ma[k] = mas[k]; }
    if (ctx->rec) {
        RowThunk t{};
        t.launch = &rule_launch_blob<MaStackOp>;
        t.blob_bytes = (int)sizeof rb;
        t.dims = dims_of(b);
        memcpy(t.blob, &rb, sizeof rb);
        for (int k = 0; k < n; k++) t.reads[t.n_reads++] = mas[k];
        t.writes[t.n_writes++] = buy; t.writes[t.n_writes++] = sell;
        return rec_add_row(ctx, t);
    }
    rule_launch_blob<MaStackOp>(&rb, ctx->stream);
    PQ_HIP_TRY(hipGetLastError());
    return PQ_OK;
}

} // extern "C"
