// plugin.hip -- Polars expression-plugin entry points (include/pq_polars_plugin.h) over the C ABI of this library: every reference
// function of the shape (1..4 Float64 columns[, timeperiod]) -> Float64.  Host code only: import the exported Series (Arrow C Data Interface, any number of chunks, validity + offset),
// run the batched HIP entry point with n_series = 1, export one Float64 chunk.
#include "../../include/pq_polars_plugin.h"
#include "pq_dev.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <sys/mman.h>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

static thread_local std::string g_plugin_err;
static void plugin_fail(const char *what) {
    g_plugin_err = what;
    const char *e = pq_last_error();
    if (e && *e) { g_plugin_err += ": "; g_plugin_err += e; }
}

// ---- a minimal pickle reader: a dict whose keys are str and whose values are int / float / bool / None (what
// polars.plugins passes for kwargs: pickle.dumps(dict, protocol 2..5))
namespace {
struct PVal { enum Kind { NONE, INT, FLOAT, STR, DICT, MARK } kind; int64_t i; double f; std::string s; };
bool pickle_scalars(const uint8_t *p, size_t n, std::vector<std::pair<std::string, PVal>> &items) {
    std::vector<PVal> st;
    size_t k = 0;
    auto need = [&](size_t m) { return k + m <= n; };
    while (k < n) {
        const uint8_t op = p[k++];
        switch (op) {
        case 0x80: if (!need(1)) return false; k += 1; break;                       // PROTO
        case 0x95: if (!need(8)) return false; k += 8; break;                       // FRAME
        case 0x94: break;                                                           // MEMOIZE
        case 'q': if (!need(1)) return false; k += 1; break;                        // BINPUT
        case 'r': if (!need(4)) return false; k += 4; break;                        // LONG_BINPUT
        case '}': st.push_back(PVal{PVal::DICT, 0, 0.0, {}}); break;                // EMPTY_DICT
        case '(': st.push_back(PVal{PVal::MARK, 0, 0.0, {}}); break;                // MARK
        case 'N': st.push_back(PVal{PVal::NONE, 0, 0.0, {}}); break;
        case 0x88: st.push_back(PVal{PVal::INT, 1, 0.0, {}}); break;                // NEWTRUE
        case 0x89: st.push_back(PVal{PVal::INT, 0, 0.0, {}}); break;                // NEWFALSE
        case 'K': if (!need(1)) return false; st.push_back(PVal{PVal::INT, p[k], 0.0, {}}); k += 1; break;
        case 'M': if (!need(2)) return false; st.push_back(PVal{PVal::INT, (int64_t)(p[k] | (p[k + 1] << 8)), 0.0, {}}); k += 2; break;
        case 'J': { if (!need(4)) return false; int32_t v; memcpy(&v, p + k, 4); st.push_back(PVal{PVal::INT, v, 0.0, {}}); k += 4; break; }
        case 0x8a: { if (!need(1)) return false; const size_t m = p[k++]; if (m > 8 || !need(m)) return false; // LONG1 (little endian, signed)
            int64_t v = 0; for (size_t j = 0; j < m; j++) v |= (int64_t)p[k + j] << (8 * j);
            if (m > 0 && m < 8 && (p[k + m - 1] & 0x80)) v |= -((int64_t)1 << (8 * m));
            st.push_back(PVal{PVal::INT, v, 0.0, {}}); k += m; break; }
        case 'G': { if (!need(8)) return false; uint64_t b = 0; for (int j = 0; j < 8; j++) b = (b << 8) | p[k + j]; // BINFLOAT big endian
            double d; memcpy(&d, &b, 8); st.push_back(PVal{PVal::FLOAT, 0, d, {}}); k += 8; break; }
        case 0x8c: { if (!need(1)) return false; const size_t m = p[k++]; if (!need(m)) return false;             // SHORT_BINUNICODE
            st.push_back(PVal{PVal::STR, 0, 0.0, std::string((const char *)p + k, m)}); k += m; break; }
        case 'X': { if (!need(4)) return false; uint32_t m; memcpy(&m, p + k, 4); k += 4; if (!need(m)) return false; // BINUNICODE
            st.push_back(PVal{PVal::STR, 0, 0.0, std::string((const char *)p + k, m)}); k += m; break; }
        case 's': { if (st.size() < 3) return false; PVal v = st.back(); st.pop_back(); PVal key = st.back(); st.pop_back(); // SETITEM
            if (key.kind != PVal::STR || st.back().kind != PVal::DICT) return false;
            items.emplace_back(key.s, v); break; }
        case 'u': { size_t m = st.size(); while (m > 0 && st[m - 1].kind != PVal::MARK) m--;                        // SETITEMS
            if (m == 0 || m < 2 || st[m - 2].kind != PVal::DICT || (st.size() - m) % 2) return false;
            for (size_t j = m; j + 1 < st.size(); j += 2) { if (st[j].kind != PVal::STR) return false; items.emplace_back(st[j].s, st[j + 1]); }
            st.resize(m - 1); break; }
        case '.': return true;                                                                                       // STOP
        default: return false;
        }
    }
    return false;
}
} // namespace

extern "C" int32_t pq_plugin_kwargs_i64(const uint8_t *pickle, size_t len, const char *key, int64_t *out) {
    if (!pickle || !len || !key || !out) return 0;
    std::vector<std::pair<std::string, PVal>> items;
    if (!pickle_scalars(pickle, len, items)) return -1;
    for (auto &kv : items)
        if (kv.first == key) {
            if (kv.second.kind == PVal::INT) { *out = kv.second.i; return 1; }
            if (kv.second.kind == PVal::FLOAT) { *out = (int64_t)kv.second.f; return 1; }
            return 0; // None
        }
    return 0;
}

// 1 found (value as double: ints up to 2^53 are exact), 0 absent or None, -1 malformed
static int kwargs_scalar(const uint8_t *pickle, size_t len, const char *key, double *out) {
    if (!pickle || !len) return 0;
    std::vector<std::pair<std::string, PVal>> items;
    if (!pickle_scalars(pickle, len, items)) return -1;
    for (auto &kv : items)
        if (kv.first == key) {
            if (kv.second.kind == PVal::INT) { *out = (double)kv.second.i; return 1; }
            if (kv.second.kind == PVal::FLOAT) { *out = kv.second.f; return 1; }
            return 0;
        }
    return 0;
}

// ---- export side: one Float64 chunk owning its two buffers
namespace {
// Host side of a result column.  std::vector would zero-fill it -- 100 MB written once by the kernel's page clearing and once more by the
// constructor: 17-20 ms of a 27 ms `ema_over` call on 5 000 x 2 520 rows (PQ_PLUGIN_TIMING=1) -- before the download overwrites every
// element, and the consumer's release unmaps it again (8-19 ms).  HostBuf takes uninitialised, 64-byte aligned memory, has a few threads
// touch its pages (a page fault per 4 KB is what is left of the cost; faults of one mapping run in parallel), and hands a released
// block of >= 1 MB to a small pool instead of the allocator: the next call of about that size finds its pages mapped.  The pool is
// bounded (PQ_PLUGIN_HOSTPOOL_MB, default 1 024, 0 = off); a block serves requests down to half its size.
static unsigned host_threads(size_t work_items, size_t per_thread) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt > 8 ? 8 : (nt < 1 ? 1 : nt);
    const size_t want = work_items / (per_thread ? per_thread : 1);
    return want < 2 ? 1 : (want < nt ? (unsigned)want : nt);
}
struct HostBlock { void *p; size_t bytes; };
static std::mutex g_pool_mu;
static std::vector<HostBlock> g_pool;
static size_t g_pool_bytes = 0;
static size_t host_pool_limit() {
    const char *e = getenv("PQ_PLUGIN_HOSTPOOL_MB");
    return (size_t)(e ? atoll(e) : 1024) << 20;
}
static void *host_block_take(size_t bytes, size_t *got) {
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        int best = -1;
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i].bytes >= bytes && g_pool[i].bytes <= 2 * bytes && (best < 0 || g_pool[i].bytes < g_pool[(size_t)best].bytes)) best = (int)i;
        if (best >= 0) {
            void *p = g_pool[(size_t)best].p;
            *got = g_pool[(size_t)best].bytes;
            g_pool_bytes -= *got;
            g_pool.erase(g_pool.begin() + best);
            return p;
        }
    }
    // fresh memory: large blocks on 2 MB boundaries with a request for transparent huge pages (one fault and one unmap per 2 MB instead of
    // per 4 KB where the system grants them -- `madvise` or `always` in /sys/kernel/mm/transparent_hugepage/enabled; harmless elsewhere)
    const size_t HUGE = (size_t)2 << 20;
    void *p = nullptr;
    if (bytes >= 2 * HUGE) {
        bytes = (bytes + HUGE - 1) / HUGE * HUGE;
        p = aligned_alloc(HUGE, bytes);
        if (p) (void)madvise(p, bytes, MADV_HUGEPAGE);
    } else p = aligned_alloc(64, bytes);
    if (!p) return nullptr;
    *got = bytes;
    const size_t PAGE = 4096, pages = (bytes + PAGE - 1) / PAGE;
    const unsigned nt = host_threads(pages, 2048); // >= 8 MB per thread
    auto touch = [p](size_t lo, size_t hi) { for (size_t g = lo; g < hi; g++) ((volatile unsigned char *)p)[g * 4096] = 0; };
    if (nt < 2) { touch(0, pages); return p; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(touch, pages * t / nt, pages * (t + 1) / nt);
    for (auto &x : th) x.join();
    return p;
}
static void host_block_give(void *p, size_t bytes) {
    if (!p) return;
    if (bytes >= ((size_t)1 << 20)) {
        std::lock_guard<std::mutex> g(g_pool_mu);
        if (g_pool_bytes + bytes <= host_pool_limit()) { g_pool.push_back({p, bytes}); g_pool_bytes += bytes; return; }
    }
    free(p);
}
template <typename T> struct HostBuf {
    T *p = nullptr;
    size_t n = 0, bytes = 0;
    HostBuf() = default;
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    ~HostBuf() { host_block_give(p, bytes); }
    bool resize(size_t count) {
        host_block_give(p, bytes); p = nullptr; n = 0; bytes = 0;
        const size_t want = (count * sizeof(T) + 63) / 64 * 64;
        p = (T *)host_block_take(want ? want : 64, &bytes);
        if (!p) return false;
        n = count;
        return true;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
};
struct OutPriv { std::vector<uint8_t> validity; HostBuf<double> values; HostBuf<int32_t> ivalues; const void *bufs[2]; std::string name; };
void release_array(ArrowArray *a) { if (a && a->release) { delete (OutPriv *)a->private_data; a->release = nullptr; } }
void release_schema(ArrowSchema *s) { if (s && s->release) { delete (std::string *)s->private_data; s->release = nullptr; } }
void fill_schema(ArrowSchema *s, const std::string &name, const char *format = "g") {
    std::string *keep = new std::string(name);
    memset(s, 0, sizeof *s);
    s->format = format; s->name = keep->c_str(); s->flags = 2 /* ARROW_FLAG_NULLABLE */; s->release = release_schema; s->private_data = keep;
}
void release_series(pq_series_export *e) {
    if (!e || !e->release) return;
    if (e->arrays) { for (size_t i = 0; i < e->len; i++) { if (e->arrays[i]) { if (e->arrays[i]->release) e->arrays[i]->release(e->arrays[i]); delete e->arrays[i]; } } delete[] e->arrays; }
    if (e->field) { if (e->field->release) e->field->release(e->field); delete e->field; }
    e->release = nullptr;
}
pq_ctx *plugin_ctx() { // one context per host thread (Polars calls plugins from its rayon workers)
    static thread_local pq_ctx *c = nullptr;
    if (!c && pq_ctx_create(0, nullptr, &c) != PQ_OK) c = nullptr;
    return c;
}
// one adapter per exported function: `in` = the NIN device columns in the reference's input order
// pv = the scalar parameters in the reference's order (integers as exact doubles)
typedef pq_status (*col_fn)(pq_ctx *, const pq_batch *, const double *const *in, const double *pv, void *out);
typedef pq_status (*cols_fn)(pq_ctx *, const pq_batch *, const double *const *in, const double *pv, double *const *out);
struct PParam { const char *name; bool is_float; double def; };
struct PlugFn { const char *name; int nin; int nparams; PParam params[8]; bool reject_nulls; bool out_i32; col_fn call;
                unsigned int_inputs; /* bit k: input k is an integer-valued column in the reference (MAVP periods, overlap.rs:407-414); documentation only: every numeric column is accepted */ };

// every reference function starts with `inputs[k].cast(&DataType::Float64)?` (overlap.rs:120,129, momentum.rs:12, volume.rs:19-31,
// pattern.rs:11-17, cycle.rs:11): any numeric column is accepted and converted on the host gather with the conversion Polars' cast uses
// (Rust `as f64`: exact for every integer up to 2^53 and for f32 / f16, round-to-nearest-even for wider 64-bit integers; Boolean ->
// 0.0 / 1.0).  Arrow formats: c C s S i I l L (int8 .. uint64), e f g (float16 / 32 / 64), b (bit-packed Boolean).
bool numeric_format(const char *fm) { return fm && fm[0] && !fm[1] && strchr("cCsSiIlLefgb", fm[0]) != nullptr; }
double half_to_double(uint16_t h) {
    const uint32_t sign = (uint32_t)(h >> 15), ex = (h >> 10) & 31u, man = h & 1023u;
    double v;
    if (ex == 0) v = ldexp((double)man, -24);                      // zero / subnormal
    else if (ex == 31) {                                            // inf / NaN (payload shifted like a hardware conversion)
        uint64_t b = 0x7FF0000000000000ull | ((uint64_t)man << 42);
        memcpy(&v, &b, 8);
    } else v = ldexp((double)(man | 1024u), (int)ex - 25);
    return sign ? -v : v;
}
template <typename T> void widen(const void *buf, int64_t off, int64_t len, double *dst) {
    const T *p = (const T *)buf + off;
    for (int64_t i = 0; i < len; i++) dst[i] = (double)p[i];
}
// gather the chunks of one exported numeric Series into a host f64 column + a validity bitmap (bit = 1: valid)
bool gather_f64(const pq_series_export &in, int64_t n, std::vector<double> &host, std::vector<uint8_t> &validity, bool &any_null) {
    const char fmt = in.field && in.field->format ? in.field->format[0] : 'g';
    host.resize((size_t)(n > 0 ? n : 1));
    validity.assign((size_t)((n + 7) / 8 + 1), 0xff);
    int64_t pos = 0;
    for (size_t c = 0; c < in.len; c++) {
        const ArrowArray *a = in.arrays[c];
        if (a->n_buffers < 2 || (!a->buffers[1] && a->length)) return false;
        double *dst = host.data() + pos;
        const void *buf = a->buffers[1];
        switch (a->length ? fmt : 0) {
        case 0: break;
        case 'g': memcpy(dst, (const double *)buf + a->offset, (size_t)a->length * 8); break;
        case 'f': widen<float>(buf, a->offset, a->length, dst); break;
        case 'e': for (int64_t i = 0; i < a->length; i++) dst[i] = half_to_double(((const uint16_t *)buf)[a->offset + i]); break;
        case 'l': widen<int64_t>(buf, a->offset, a->length, dst); break;
        case 'L': widen<uint64_t>(buf, a->offset, a->length, dst); break;
        case 'i': widen<int32_t>(buf, a->offset, a->length, dst); break;
        case 'I': widen<uint32_t>(buf, a->offset, a->length, dst); break;
        case 's': widen<int16_t>(buf, a->offset, a->length, dst); break;
        case 'S': widen<uint16_t>(buf, a->offset, a->length, dst); break;
        case 'c': widen<int8_t>(buf, a->offset, a->length, dst); break;
        case 'C': widen<uint8_t>(buf, a->offset, a->length, dst); break;
        case 'b': for (int64_t i = 0; i < a->length; i++) { const int64_t bi = a->offset + i; dst[i] = (double)((((const uint8_t *)buf)[bi >> 3] >> (bi & 7)) & 1); } break;
        default: return false;
        }
        const uint8_t *vb = (const uint8_t *)a->buffers[0];
        if (vb && a->null_count != 0)
            for (int64_t i = 0; i < a->length; i++) {
                const int64_t bi = a->offset + i;
                if (!((vb[bi >> 3] >> (bi & 7)) & 1)) { validity[(size_t)((pos + i) >> 3)] &= (uint8_t)~(1u << ((pos + i) & 7)); any_null = true; }
            }
        pos += a->length;
    }
    return true;
}
int64_t series_len(const pq_series_export &in) {
    int64_t n = 0;
    for (size_t c = 0; c < in.len; c++) n += in.arrays[c]->length;
    return n;
}


// ---- batched form: `<f>_over(columns..., key, params...)`.  Polars evaluates `<f>(...).over("symbol")` with one plugin call per group
// (python/polars_quant/talib/momentum.py:13-16, is_elementwise=False): one H2D, one launch and one D2H per ~20 KB group.  The
// `_over` symbols take the WHOLE columns of a frame in which equal keys are contiguous (sorted by symbol) plus the key column
// itself, derive the group offsets from it and run every group in one ragged launch (pq_batch.offsets); the result is the
// concatenation of the per-group results, i.e. exactly what `.over(key)` assembles.  The key may be any integer / float column, a
// string column (utf8, large_utf8, string view) or a dictionary-encoded one; null keys form their own group.
struct KeyRef { const uint8_t *p; int64_t len; bool null; };
static bool key_refs(const pq_series_export &key, int64_t n, std::vector<KeyRef> &out) {
    if (!key.field || !key.field->format) return false;
    const char *fm = key.field->format;
    int width = 0;
    switch (fm[0]) {
    case 'c': case 'C': case 'b': width = 1; break;
    case 's': case 'S': width = 2; break;
    case 'i': case 'I': case 'f': width = 4; break;
    case 'l': case 'L': case 'g': width = 8; break;
    default: break;
    }
    const bool small = !strcmp(fm, "u") || !strcmp(fm, "z"), large = !strcmp(fm, "U") || !strcmp(fm, "Z");
    const bool view = !strcmp(fm, "vu") || !strcmp(fm, "vz");
    if (fm[0] == 'b' || (fm[1] != 0 && !view)) width = 0; // bit-packed booleans and parametrised types are not keys
    if (!width && !small && !large && !view) return false;
    out.clear();
    out.reserve((size_t)n);
    for (size_t c = 0; c < key.len; c++) {
        const ArrowArray *a = key.arrays[c];
        const uint8_t *vb = a->n_buffers >= 1 ? (const uint8_t *)a->buffers[0] : nullptr;
        for (int64_t i = 0; i < a->length; i++) {
            const int64_t r = a->offset + i;
            KeyRef k{nullptr, 0, false};
            if (vb && a->null_count != 0 && !((vb[r >> 3] >> (r & 7)) & 1)) k.null = true;
            else if (width) { if (a->n_buffers < 2 || !a->buffers[1]) return false; k.p = (const uint8_t *)a->buffers[1] + r * width; k.len = width; }
            else if (small || large) {
                if (a->n_buffers < 3 || !a->buffers[1]) return false;
                const int64_t lo = small ? ((const int32_t *)a->buffers[1])[r] : ((const int64_t *)a->buffers[1])[r];
                const int64_t hi = small ? ((const int32_t *)a->buffers[1])[r + 1] : ((const int64_t *)a->buffers[1])[r + 1];
                k.p = (const uint8_t *)a->buffers[2] + lo; k.len = hi - lo;
            } else { // 16-byte views: length, then 12 inline bytes or (prefix, buffer index, offset)
                if (a->n_buffers < 2 || !a->buffers[1]) return false;
                const uint8_t *v = (const uint8_t *)a->buffers[1] + r * 16;
                int32_t len; memcpy(&len, v, 4);
                k.len = len;
                if (len <= 12) k.p = v + 4;
                else {
                    int32_t bi, bo; memcpy(&bi, v + 8, 4); memcpy(&bo, v + 12, 4);
                    if (2 + bi >= a->n_buffers || !a->buffers[2 + bi]) return false;
                    k.p = (const uint8_t *)a->buffers[2 + bi] + bo;
                }
            }
            out.push_back(k);
        }
    }
    return (int64_t)out.size() == n;
}
// How the columns of one call sit on the device.  A plain call: one series of n rows.  An `_over` call: the group offsets come from the
// key column (host); groups of one common length (a balanced panel) become a REGULAR batch -- the tiled bodies instead of the ragged
// forms -- and, since the device columns are this library's own copies, they are placed at the 128-byte row pitch
// (pq_recommended_stride): a panel of ODD length would otherwise run the 8-byte forms of every kernel (1.5x slower).
// Anything else is a ragged batch over the flat long columns.
struct Layout {
    pq_batch b{1, 0, 0, nullptr};
    int64_t n = 0;         // rows of the long columns
    int64_t groups = 1, glen = 0, pitch = 0;
    bool pitched = false;  // device columns are [groups][pitch], host columns stay flat [n]
    void *d_off = nullptr;
    size_t dev_elems() const { return pitched ? (size_t)groups * (size_t)pitch : (size_t)n; }
};
static pq_status plan_layout(pq_ctx *ctx, const pq_series_export *key, int64_t n, Layout *L) {
    L->n = n; L->groups = 1; L->glen = n; L->pitch = n; L->pitched = false; L->d_off = nullptr;
    L->b = pq_batch{1, n, n, nullptr};
    if (!key) return PQ_OK;
    std::vector<int64_t> off;
    off.push_back(0);
    int64_t longest = 0;
    // The usual key -- ONE chunk of a fixed-width type without nulls (an integer symbol id, a date) -- is scanned in place: neighbours
    // compared as raw words, 12.6 M rows in a few milliseconds.  (Through the general KeyRef list below the same scan builds a 300 MB
    // vector first and memcmp()s every row: ~100 ms of a 120 ms `_over` call on a 5 000 x 2 520 frame, scripts/bench_plugin.py.)
    bool scanned = false;
    if (key->len == 1 && key->field && key->field->format && key->field->format[0] && !key->field->format[1] && key->arrays && key->arrays[0]) {
        const ArrowArray *a = key->arrays[0];
        int width = 0;
        switch (key->field->format[0]) {
        case 'c': case 'C': width = 1; break;
        case 's': case 'S': width = 2; break;
        case 'i': case 'I': case 'f': width = 4; break;
        case 'l': case 'L': case 'g': width = 8; break;
        default: break;
        }
        if (width && a->length == n && a->n_buffers >= 2 && a->buffers[1] && (a->null_count == 0 || !a->buffers[0])) {
            auto scan = [&](auto *p) { // (several host threads, each over its own range of rows; the boundaries joined in order)
                const unsigned nt = host_threads((size_t)n, (size_t)1 << 20);
                std::vector<std::vector<int64_t>> part(nt);
                auto range = [&](unsigned t) {
                    const int64_t lo = std::max<int64_t>(1, n * t / nt), hi = n * (t + 1) / nt;
                    for (int64_t i = lo; i < hi; i++) if (p[i] != p[i - 1]) part[t].push_back(i);
                };
                if (nt < 2) range(0);
                else {
                    std::vector<std::thread> th;
                    for (unsigned t = 0; t < nt; t++) th.emplace_back(range, t);
                    for (auto &x : th) x.join();
                }
                for (const auto &v : part)
                    for (int64_t i : v) { longest = std::max<int64_t>(longest, i - off.back()); off.push_back(i); }
            };
            const uint8_t *base = (const uint8_t *)a->buffers[1] + (size_t)a->offset * (size_t)width;
            switch (width) { // (floats too are compared as words: the general path compares their bytes)
            case 1: scan((const uint8_t *)base); break;
            case 2: scan((const uint16_t *)base); break;
            case 4: scan((const uint32_t *)base); break;
            default: scan((const uint64_t *)base); break;
            }
            longest = std::max<int64_t>(longest, n - off.back());
            off.push_back(n);
            scanned = true;
        }
    }
    if (!scanned) {
        std::vector<KeyRef> k;
        if (!key_refs(*key, n, k)) { pq_set_error("plugin: the key column of an _over call must be an integer, float, string or dictionary column of the frame's length"); return PQ_ERR_ARG; }
        for (int64_t i = 1; i <= n; i++) {
            const bool same = i < n && k[(size_t)i].null == k[(size_t)i - 1].null &&
                              (k[(size_t)i].null || (k[(size_t)i].len == k[(size_t)i - 1].len && !memcmp(k[(size_t)i].p, k[(size_t)i - 1].p, (size_t)k[(size_t)i].len)));
            if (!same) { longest = std::max<int64_t>(longest, i - off.back()); off.push_back(i); }
        }
    }
    bool uniform = off.size() >= 2;
    for (size_t i = 1; uniform && i < off.size(); i++) uniform = off[i] - off[i - 1] == longest;
    if (uniform) {
        L->groups = (int64_t)off.size() - 1; L->glen = longest; L->pitch = pq_recommended_stride(longest); L->pitched = L->pitch != longest;
        L->b = pq_batch{L->groups, longest, L->pitch, nullptr};
        return PQ_OK;
    }
    PQ_TRY(pq_malloc(ctx, off.size() * 8, &L->d_off));
    PQ_TRY(pq_memcpy_h2d(ctx, L->d_off, off.data(), off.size() * 8));
    PQ_TRY(pq_ctx_sync(ctx)); // `off` is a pageable local
    L->groups = (int64_t)off.size() - 1; L->glen = longest;
    L->b = pq_batch{L->groups, longest, n, (const int64_t *)L->d_off};
    return PQ_OK;
}
// a flat host column [n] of `elem`-byte values -> a fresh device column in the call's layout (and back)
static pq_status upload_col(pq_ctx *ctx, const Layout &L, const void *host, size_t elem, void **d) {
    PQ_TRY(pq_malloc(ctx, L.dev_elems() * elem, d));
    // a pitched panel: the tiled bodies load whole 16-byte tile pieces, i.e. the padding behind every group's rows too -- zeros, not whatever
    // the allocation held (the results inside the rows never depend on it, but no kernel should consume indeterminate bits)
    if (L.pitched) PQ_HIP_TRY(hipMemsetAsync(*d, 0, L.dev_elems() * elem, ctx->stream));
    if (L.pitched) return pq_memcpy_h2d_pitched(ctx, *d, (size_t)L.pitch * elem, host, (size_t)L.glen * elem, (size_t)L.glen * elem, (size_t)L.groups);
    return pq_memcpy_h2d(ctx, *d, host, (size_t)L.n * elem);
}
// ---- input-column cache ------------------------------------------------------------------------------------------------------------
// Polars evaluates the sixty expressions of one `with_columns` one plugin call each, and every call that reads `close` hands over the
// SAME Arrow buffer (python/polars_quant/talib/momentum.py:13-16): gathered and uploaded again each time, 100 MB per column of a
// 5 000 x 2 520 frame.  A Float64 column that arrives as one chunk without nulls is therefore (i) uploaded straight from its Arrow
// buffer -- no host copy -- and (ii) kept: the device copy is remembered under (buffer address, rows, layout of the call) together with a
// 64-bit hash of the WHOLE buffer, and a later call with the same key and the same hash uses it.  The hash is recomputed on every call
// (memory-bound, ~10 ms per 100 MB against ~40 ms for gather + pageable upload), so a freed-and-reused address with other content can
// not alias: other bytes, other hash, a miss.  Shared by the host threads Polars calls from (a mutex; an entry is published after its
// creator's stream has been drained, is never freed while a call uses it, least recently used entries go when PQ_PLUGIN_CACHE_MB --
// default 2 048, 0 = no cache -- is exceeded).
struct CacheEntry {
    const void *addr; int64_t n, groups, glen, pitch; uint64_t hash;
    void *d; size_t bytes; int in_use; uint64_t tick;
};
static std::mutex g_cache_mu;
static std::vector<CacheEntry> g_cache;
static size_t g_cache_bytes = 0;
static uint64_t g_cache_tick = 0;
static int64_t g_cache_hits = 0, g_cache_misses = 0;
static size_t cache_limit() {
    const char *e = getenv("PQ_PLUGIN_CACHE_MB");
    return (size_t)(e ? atoll(e) : 2048) << 20;
}
static uint64_t hash_bytes(const void *p, size_t nbytes) { // four interleaved multiply-rotate lanes over 8-byte words, then a mix
    const uint64_t K = 0x9E3779B97F4A7C15ULL;
    uint64_t h[4] = {0x243F6A8885A308D3ULL, 0x13198A2E03707344ULL, 0xA4093822299F31D0ULL, 0x082EFA98EC4E6C89ULL};
    const uint8_t *b = (const uint8_t *)p;
    size_t i = 0;
    for (; i + 32 <= nbytes; i += 32) {
        uint64_t w[4];
        memcpy(w, b + i, 32);
        for (int k = 0; k < 4; k++) { h[k] = (h[k] ^ w[k]) * K; h[k] = (h[k] << 29) | (h[k] >> 35); }
    }
    uint64_t tail[4] = {0, 0, 0, 0};
    memcpy(tail, b + i, nbytes - i);
    for (int k = 0; k < 4; k++) { h[k] = (h[k] ^ tail[k]) * K; h[k] ^= h[k] >> 32; }
    uint64_t r = (uint64_t)nbytes * K;
    for (int k = 0; k < 4; k++) { r = (r ^ h[k]) * K; r ^= r >> 29; }
    return r;
}
// the hash of a large buffer on several host threads (one core hashes ~15 GB/s, about what the runtime's pageable upload moves -- a hit
// would cost what it saves; eight threads leave the validation at a fifth of the upload): chunk hashes combined in order
static uint64_t hash_buffer(const void *p, size_t nbytes) {
    const size_t CH = (size_t)8 << 20;
    const size_t nch = (nbytes + CH - 1) / CH;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt > 16 ? 16 : (nt < 1 ? 1 : nt);
    if (nch < 2 || nt < 2) return hash_bytes(p, nbytes);
    std::vector<uint64_t> hs(nch);
    std::vector<std::thread> th;
    const unsigned use = nt < nch ? nt : (unsigned)nch;
    for (unsigned t = 0; t < use; t++)
        th.emplace_back([&, t]() {
            for (size_t c = t; c < nch; c += use) {
                const size_t lo = c * CH, len = lo + CH <= nbytes ? CH : nbytes - lo;
                hs[c] = hash_bytes((const uint8_t *)p + lo, len);
            }
        });
    for (auto &x : th) x.join();
    return hash_bytes(hs.data(), hs.size() * 8) ^ (uint64_t)nbytes;
}
struct InCol { void *d = nullptr; bool owned = false, publish = false; int hit = 0; CacheEntry key{}; };
// one chunk of Float64 without nulls: its values buffer IS the flat host column
static const double *flat_f64(const pq_series_export &in, int64_t n) {
    if (in.len != 1 || !in.field || !in.field->format || strcmp(in.field->format, "g")) return nullptr;
    const ArrowArray *a = in.arrays[0];
    if (!a || a->length != n || a->n_buffers < 2 || !a->buffers[1]) return nullptr;
    if (a->null_count != 0 && a->buffers[0]) return nullptr;
    return (const double *)a->buffers[1] + a->offset;
}
// the hashes of a call's flat inputs on a helper thread, started before the key column is scanned (plan_layout) and joined behind it:
// the two passes over host memory are independent
struct HashAhead {
    uint64_t h[4] = {0, 0, 0, 0};
    bool have[4] = {false, false, false, false};
    std::thread th;
    void start(const double *const *flat, int nin, int64_t n) {
        if (!cache_limit() || (size_t)n * 8 < ((size_t)8 << 20)) return;
        const double *f[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < nin && k < 4; k++) f[k] = flat[k];
        th = std::thread([this, f, n]() { for (int k = 0; k < 4; k++) if (f[k]) { h[k] = hash_buffer(f[k], (size_t)n * 8); have[k] = true; } });
    }
    void join() { if (th.joinable()) th.join(); }
    const uint64_t *get(int k) const { return have[k] ? &h[k] : nullptr; }
    ~HashAhead() { join(); }
};
static pq_status cached_input(pq_ctx *ctx, const Layout &L, const double *src, int64_t n, InCol *ic, const uint64_t *hash = nullptr) {
    CacheEntry &k = ic->key;
    k.addr = src; k.n = n;
    k.groups = L.pitched ? L.groups : 1; k.glen = L.pitched ? L.glen : n; k.pitch = L.pitched ? L.pitch : n;
    const size_t limit = cache_limit();
    if (limit) {
        k.hash = hash ? *hash : hash_buffer(src, (size_t)n * 8);
        std::lock_guard<std::mutex> g(g_cache_mu);
        for (CacheEntry &e : g_cache)
            if (e.addr == k.addr && e.n == k.n && e.groups == k.groups && e.glen == k.glen && e.pitch == k.pitch && e.hash == k.hash) {
                e.in_use++; e.tick = ++g_cache_tick; g_cache_hits++;
                ic->d = e.d; ic->hit = 1;
                return PQ_OK;
            }
        g_cache_misses++;
    }
    PQ_TRY(upload_col(ctx, L, src, 8, &ic->d)); // straight from the Arrow buffer
    ic->owned = true;
    ic->publish = limit && L.dev_elems() * 8 <= limit;
    return PQ_OK;
}
// after the call's stream has been drained (`synced`): publish / release / free the device copies of the inputs
static void inputs_done(pq_ctx *ctx, InCol *ic, int n_in, const Layout &L, bool synced) {
    for (int k = 0; k < n_in; k++) {
        InCol &c = ic[k];
        if (c.hit) {
            std::lock_guard<std::mutex> g(g_cache_mu);
            for (CacheEntry &e : g_cache) if (e.d == c.d) { e.in_use--; break; }
        } else if (c.owned && c.publish && synced && c.d) {
            std::vector<void *> drop;
            {
                std::lock_guard<std::mutex> g(g_cache_mu);
                CacheEntry e = c.key;
                e.d = c.d; e.bytes = L.dev_elems() * 8; e.in_use = 0; e.tick = ++g_cache_tick;
                g_cache.push_back(e);
                g_cache_bytes += e.bytes;
                const size_t limit = cache_limit();
                while (g_cache_bytes > limit) { // least recently used entry that no call holds
                    int victim = -1;
                    for (size_t i = 0; i < g_cache.size(); i++)
                        if (!g_cache[i].in_use && g_cache[i].d != c.d && (victim < 0 || g_cache[i].tick < g_cache[(size_t)victim].tick)) victim = (int)i;
                    if (victim < 0) break;
                    drop.push_back(g_cache[(size_t)victim].d);
                    g_cache_bytes -= g_cache[(size_t)victim].bytes;
                    g_cache.erase(g_cache.begin() + victim);
                }
            }
            for (void *q : drop) (void)pq_free(ctx, q);
        } else if (c.owned && c.d) (void)pq_free(ctx, c.d);
        c = InCol{};
    }
}
static pq_status download_col(pq_ctx *ctx, const Layout &L, const void *d, size_t elem, void *host) {
    if (L.pitched) return pq_memcpy_d2h_pitched(ctx, host, (size_t)L.glen * elem, d, (size_t)L.pitch * elem, (size_t)L.glen * elem, (size_t)L.groups);
    return pq_memcpy_d2h(ctx, host, d, (size_t)L.n * elem);
}
// Arrow validity <-> the NULL bit pattern, on the host copies (the plugin owns them; the device entry points pq_nulls_from_arrow /
// pq_validity_to_arrow do the same for callers whose columns already live on the device)
static void nulls_into_host(std::vector<double> &host, const std::vector<uint8_t> &validity, int64_t n) {
    const uint64_t nb = PQ_NULL_BITS;
    for (int64_t i = 0; i < n; i++)
        if (!((validity[(size_t)(i >> 3)] >> (i & 7)) & 1)) memcpy(&host[(size_t)i], &nb, 8);
}
static int64_t validity_from_host(const double *values, int64_t n, std::vector<uint8_t> &validity) {
    auto scan = [&](int64_t lo, int64_t hi) { // [lo, hi): lo a multiple of 8, so no two threads share a bitmap byte
        int64_t nulls = 0;
        for (int64_t i = lo; i < hi; i++) {
            uint64_t bits; memcpy(&bits, &values[(size_t)i], 8);
            if (bits == PQ_NULL_BITS) { validity[(size_t)(i >> 3)] &= (uint8_t)~(1u << (i & 7)); nulls++; }
        }
        return nulls;
    };
    const unsigned nt = host_threads((size_t)n, (size_t)1 << 20);
    if (nt < 2) return scan(0, n);
    std::vector<int64_t> part(nt, 0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) {
        const int64_t lo = n * t / nt / 8 * 8, hi = t + 1 == nt ? n : n * (t + 1) / nt / 8 * 8;
        th.emplace_back([&, t, lo, hi]() { part[t] = scan(lo, hi); });
    }
    for (auto &x : th) x.join();
    int64_t nulls = 0;
    for (int64_t v : part) nulls += v;
    return nulls;
}
static const char *const k_not_numeric = "plugin: the input column is not numeric (int8 .. uint64, float16 / 32 / 64 and Boolean are cast to Float64 like the reference's inputs[k].cast(&DataType::Float64))";
bool read_params(const PParam *params, int nparams, int nin, pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs,
                 size_t kwargs_len, double (&pv)[8]);
// PQ_PLUGIN_TIMING=1: where a call's host time goes, one line per call on stderr
struct PhaseClock {
    bool on = getenv("PQ_PLUGIN_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    std::string line;
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        char buf[64];
        snprintf(buf, sizeof buf, " %s %.2f", what, std::chrono::duration<double, std::milli>(now - t).count());
        line += buf; t = now;
    }
    void done(const char *name) { if (on) fprintf(stderr, "[pq plugin] %s (ms):%s\n", name, line.c_str()); }
};
void run_cols(const PlugFn &f, pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len, pq_series_export *ret, bool over = false) {
    PhaseClock clk;
    if (ret) memset(ret, 0, sizeof *ret);
    g_plugin_err.clear();
    if (!inputs || (int)n_inputs < f.nin + (over ? 1 : 0) || !ret) { plugin_fail("plugin: bad arguments (too few input Series)"); return; }
    for (int k = 0; k < f.nin; k++) {
        if (!inputs[k].field || !inputs[k].field->format || (inputs[k].len && !inputs[k].arrays)) { plugin_fail("plugin: bad arguments"); return; }
        if (!numeric_format(inputs[k].field->format)) { plugin_fail(k_not_numeric); return; }
    }
    double pv[8];
    if (!read_params(f.params, f.nparams, f.nin + (over ? 1 : 0), inputs, n_inputs, kwargs, kwargs_len, pv)) return;
    const int64_t n = series_len(inputs[0]);
    for (int k = 1; k < f.nin; k++)
        if (series_len(inputs[k]) != n) { plugin_fail("plugin: the input Series differ in length"); return; }
    std::vector<double> host[4];
    std::vector<uint8_t> valid[4];
    bool nulls[4] = {false, false, false, false}, any_null = false;
    const double *flat[4] = {nullptr, nullptr, nullptr, nullptr}; // one Float64 chunk without nulls: used in place (and cached), no gather
    for (int k = 0; k < f.nin; k++) {
        if ((flat[k] = flat_f64(inputs[k], n))) continue;
        if (!gather_f64(inputs[k], n, host[k], valid[k], nulls[k])) { plugin_fail("plugin: malformed input chunk"); return; }
        any_null |= nulls[k];
    }
    // the momentum / cycle families go through rechunk().cont_slice()? in the reference (momentum.rs:12-13): a null is an error there
    if (any_null && f.reject_nulls) { plugin_fail("plugin: the input contains nulls (this function rejects them, as the reference's cont_slice() does)"); return; }
    clk.mark("gather");
    OutPriv *op = new OutPriv();
    if (!(f.out_i32 ? op->ivalues.resize((size_t)(n > 0 ? n : 1)) : op->values.resize((size_t)(n > 0 ? n : 1)))) { delete op; plugin_fail("plugin: out of host memory for the result column"); return; }
    op->validity.assign((size_t)((n + 7) / 8 + 1), 0xff);
    clk.mark("host-alloc");
    int64_t null_count = 0;
    if (n > 0) {
        pq_ctx *ctx = plugin_ctx();
        if (!ctx) { delete op; plugin_fail("plugin: no HIP device / context"); return; }
        void *d_in[4] = {nullptr, nullptr, nullptr, nullptr}, *d_out = nullptr;
        InCol ic[4];
        Layout lay;
        HashAhead ha;
        if (over) ha.start(flat, f.nin, n);
        pq_status st = plan_layout(ctx, over ? &inputs[f.nin] : nullptr, n, &lay);
        ha.join();
        clk.mark("layout");
        if (st == PQ_OK) st = pq_malloc(ctx, lay.dev_elems() * 8, &d_out);
        clk.mark("dev-alloc");
        for (int k = 0; k < f.nin && st == PQ_OK; k++) {
            if (flat[k]) st = cached_input(ctx, lay, flat[k], n, &ic[k], ha.get(k));
            else {
                if (nulls[k]) nulls_into_host(host[k], valid[k], n);
                st = upload_col(ctx, lay, host[k].data(), 8, &ic[k].d);
                ic[k].owned = true;
            }
            d_in[k] = ic[k].d;
        }
        clk.mark("inputs");
        const double *cols[4] = {(const double *)d_in[0], (const double *)d_in[1], (const double *)d_in[2], (const double *)d_in[3]};
        if (st == PQ_OK) st = f.call(ctx, &lay.b, cols, pv, d_out);
        clk.mark("launch");
        if (f.out_i32) { // Int32 results of this library are never null on non-null input rows beyond the warm-up: PQ_NULL_I32 marks the rest
            if (st == PQ_OK) st = download_col(ctx, lay, d_out, 4, op->ivalues.data());
            if (st == PQ_OK) st = pq_ctx_sync(ctx);
            if (st == PQ_OK)
                for (int64_t i = 0; i < n; i++)
                    if (op->ivalues[(size_t)i] == PQ_NULL_I32) { op->validity[(size_t)(i >> 3)] &= (uint8_t)~(1u << (i & 7)); null_count++; }
        } else {
            if (st == PQ_OK) st = download_col(ctx, lay, d_out, 8, op->values.data());
            if (st == PQ_OK) st = pq_ctx_sync(ctx);
            clk.mark("download+sync");
            if (st == PQ_OK) null_count = validity_from_host(op->values.data(), n, op->validity);
            clk.mark("validity");
        }
        if (st == PQ_OK) st = pq_ctx_sync(ctx);
        inputs_done(ctx, ic, 4, lay, st == PQ_OK);
        for (void *q : {d_out, lay.d_off}) if (q) (void)pq_free(ctx, q);
        clk.mark("free");
        if (st != PQ_OK) { delete op; plugin_fail(f.name); return; }
    }
    ArrowArray *arr = new ArrowArray();
    memset(arr, 0, sizeof *arr);
    op->bufs[0] = null_count ? op->validity.data() : nullptr;
    op->bufs[1] = f.out_i32 ? (const void *)op->ivalues.data() : (const void *)op->values.data();
    arr->length = n; arr->null_count = null_count; arr->n_buffers = 2; arr->buffers = op->bufs; arr->release = release_array; arr->private_data = op;
    ret->field = new ArrowSchema();
    fill_schema(ret->field, inputs[0].field->name ? inputs[0].field->name : "", f.out_i32 ? "i" : "g");
    ret->arrays = new ArrowArray *[1];
    ret->arrays[0] = arr;
    ret->len = 1;
    ret->release = release_series;
    clk.done(f.name);
}
// every scalar parameter of a function: pickled kwargs by name first (overlap.rs:18-22), else its trailing literal input
// (overlap.py:36-43; momentum.rs reads inputs[nin + k] in declaration order), else the reference's default
bool read_params(const PParam *params, int nparams, int nin, pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs,
                 size_t kwargs_len, double (&pv)[8]) {
    for (int k = 0; k < 8; k++) pv[k] = 0.0;
    for (int k = 0; k < nparams; k++) {
        pv[k] = params[k].def;
        double v = 0.0;
        const int kw = kwargs_scalar(kwargs, kwargs_len, params[k].name, &v);
        if (kw < 0) { plugin_fail("plugin: cannot parse the pickled kwargs"); return false; }
        const pq_series_export *lit = (int)n_inputs > nin + k ? &inputs[nin + k] : nullptr;
        if (kw == 1) pv[k] = v;
        else if (lit && lit->len >= 1 && lit->arrays && lit->arrays[0] && lit->arrays[0]->length >= 1 && lit->field && lit->field->format) {
            const ArrowArray *a = lit->arrays[0];
            const char *fm = lit->field->format;
            const void *data = a->n_buffers >= 2 ? a->buffers[1] : nullptr;
            const uint8_t *vb = a->n_buffers >= 1 ? (const uint8_t *)a->buffers[0] : nullptr;
            const bool is_null = vb && a->null_count != 0 && !((vb[a->offset >> 3] >> (a->offset & 7)) & 1); // a null literal = the default
            if (data && !is_null) {
                if (!strcmp(fm, "l")) pv[k] = (double)((const int64_t *)data)[a->offset];
                else if (!strcmp(fm, "i")) pv[k] = (double)((const int32_t *)data)[a->offset];
                else if (!strcmp(fm, "g")) pv[k] = ((const double *)data)[a->offset];
            }
        }
        if (!params[k].is_float) pv[k] = (double)(int64_t)pv[k];
    }
    return true;
}

// ---- Struct-valued functions (bbands, mama, aroon, macd, ht_phasor, ht_sine): a "+s" array whose children are Float64 columns
struct StructFn { const char *name; const char *struct_name; int nin; int nparams; PParam params[4]; bool reject_nulls; int nout;
                  const char *fields[3]; cols_fn call; };
struct StructPriv { ArrowArray *kids[3]; const void *bufs[1]; };
void release_struct_array(ArrowArray *a) {
    if (!a || !a->release) return;
    StructPriv *sp = (StructPriv *)a->private_data;
    for (int64_t k = 0; k < a->n_children; k++)
        if (sp->kids[k]) { if (sp->kids[k]->release) sp->kids[k]->release(sp->kids[k]); delete sp->kids[k]; }
    delete sp;
    a->release = nullptr;
}
struct SchemaPriv { std::string name; ArrowSchema *kids[3]; int n; };
void release_struct_schema(ArrowSchema *s) {
    if (!s || !s->release) return;
    SchemaPriv *sp = (SchemaPriv *)s->private_data;
    for (int k = 0; k < sp->n; k++)
        if (sp->kids[k]) { if (sp->kids[k]->release) sp->kids[k]->release(sp->kids[k]); delete sp->kids[k]; }
    delete sp;
    s->release = nullptr;
}
void fill_struct_schema(ArrowSchema *s, const StructFn &f) {
    SchemaPriv *sp = new SchemaPriv();
    sp->name = f.struct_name; sp->n = f.nout;
    for (int k = 0; k < f.nout; k++) { sp->kids[k] = new ArrowSchema(); fill_schema(sp->kids[k], f.fields[k], "g"); }
    memset(s, 0, sizeof *s);
    s->format = "+s"; s->name = sp->name.c_str(); s->flags = 2; s->n_children = f.nout; s->children = sp->kids;
    s->release = release_struct_schema; s->private_data = sp;
}
void run_struct(const StructFn &f, pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len, pq_series_export *ret, bool over = false) {
    if (ret) memset(ret, 0, sizeof *ret);
    g_plugin_err.clear();
    if (!inputs || (int)n_inputs < f.nin + (over ? 1 : 0) || !ret) { plugin_fail("plugin: bad arguments (too few input Series)"); return; }
    for (int k = 0; k < f.nin; k++) {
        if (!inputs[k].field || !inputs[k].field->format || (inputs[k].len && !inputs[k].arrays)) { plugin_fail("plugin: bad arguments"); return; }
        if (!numeric_format(inputs[k].field->format)) { plugin_fail(k_not_numeric); return; }
    }
    double pv[8];
    if (!read_params(f.params, f.nparams, f.nin + (over ? 1 : 0), inputs, n_inputs, kwargs, kwargs_len, pv)) return;
    const int64_t n = series_len(inputs[0]);
    std::vector<double> host[2];
    std::vector<uint8_t> valid[2];
    bool nulls[2] = {false, false}, any_null = false;
    const double *flat[2] = {nullptr, nullptr};
    for (int k = 0; k < f.nin; k++) {
        if (series_len(inputs[k]) != n) { plugin_fail("plugin: the input Series differ in length"); return; }
        if ((flat[k] = flat_f64(inputs[k], n))) continue;
        if (!gather_f64(inputs[k], n, host[k], valid[k], nulls[k])) { plugin_fail("plugin: malformed input chunk"); return; }
        any_null |= nulls[k];
    }
    if (any_null && f.reject_nulls) { plugin_fail("plugin: the input contains nulls (this function rejects them, as the reference's cont_slice() does)"); return; }
    OutPriv *op[3] = {nullptr, nullptr, nullptr};
    int64_t null_count[3] = {0, 0, 0};
    for (int k = 0; k < f.nout; k++) {
        op[k] = new OutPriv();
        if (!op[k]->values.resize((size_t)(n > 0 ? n : 1))) { for (int j = 0; j <= k; j++) delete op[j]; plugin_fail("plugin: out of host memory for the result columns"); return; }
        op[k]->validity.assign((size_t)((n + 7) / 8 + 1), 0xff);
    }
    auto drop = [&]() { for (int k = 0; k < f.nout; k++) delete op[k]; };
    if (n > 0) {
        pq_ctx *ctx = plugin_ctx();
        if (!ctx) { drop(); plugin_fail("plugin: no HIP device / context"); return; }
        void *d_in[2] = {nullptr, nullptr}, *d_out[3] = {nullptr, nullptr, nullptr};
        InCol ic[2];
        Layout lay;
        HashAhead ha;
        if (over) ha.start(flat, f.nin, n);
        pq_status st = plan_layout(ctx, over ? &inputs[f.nin] : nullptr, n, &lay);
        ha.join();
        for (int k = 0; k < f.nout && st == PQ_OK; k++) st = pq_malloc(ctx, lay.dev_elems() * 8, &d_out[k]);
        for (int k = 0; k < f.nin && st == PQ_OK; k++) {
            if (flat[k]) st = cached_input(ctx, lay, flat[k], n, &ic[k], ha.get(k));
            else {
                if (nulls[k]) nulls_into_host(host[k], valid[k], n);
                st = upload_col(ctx, lay, host[k].data(), 8, &ic[k].d);
                ic[k].owned = true;
            }
            d_in[k] = ic[k].d;
        }
        const double *cols[2] = {(const double *)d_in[0], (const double *)d_in[1]};
        double *outs[3] = {(double *)d_out[0], (double *)d_out[1], (double *)d_out[2]};
        if (st == PQ_OK) st = f.call(ctx, &lay.b, cols, pv, outs);
        for (int k = 0; k < f.nout && st == PQ_OK; k++) st = download_col(ctx, lay, d_out[k], 8, op[k]->values.data());
        if (st == PQ_OK) st = pq_ctx_sync(ctx);
        for (int k = 0; k < f.nout && st == PQ_OK; k++) null_count[k] = validity_from_host(op[k]->values.data(), n, op[k]->validity);
        inputs_done(ctx, ic, 2, lay, st == PQ_OK);
        for (void *q : {d_out[0], d_out[1], d_out[2], lay.d_off}) if (q) (void)pq_free(ctx, q);
        if (st != PQ_OK) { drop(); plugin_fail(f.name); return; }
    }
    StructPriv *sp = new StructPriv();
    memset(sp, 0, sizeof *sp);
    for (int k = 0; k < f.nout; k++) {
        ArrowArray *kid = new ArrowArray();
        memset(kid, 0, sizeof *kid);
        op[k]->bufs[0] = null_count[k] ? op[k]->validity.data() : nullptr;
        op[k]->bufs[1] = op[k]->values.data();
        kid->length = n; kid->null_count = null_count[k]; kid->n_buffers = 2; kid->buffers = op[k]->bufs; kid->release = release_array; kid->private_data = op[k];
        sp->kids[k] = kid;
    }
    ArrowArray *arr = new ArrowArray();
    memset(arr, 0, sizeof *arr);
    sp->bufs[0] = nullptr; // the struct itself is never null: the nulls live in its fields
    arr->length = n; arr->null_count = 0; arr->n_buffers = 1; arr->buffers = sp->bufs; arr->n_children = f.nout; arr->children = sp->kids;
    arr->release = release_struct_array; arr->private_data = sp;
    ret->field = new ArrowSchema();
    fill_struct_schema(ret->field, f);
    ret->arrays = new ArrowArray *[1];
    ret->arrays[0] = arr;
    ret->len = 1;
    ret->release = release_series;
}

// the 61 candlestick recognisers: (open, high, low, close[, penetration literal]) -> Int32, never null (pattern.rs:10-2062; inputs go
// through cont_slice(): a null is an error; penetration = inputs.get(4) as f64, default 0.3, pattern.rs:529-532)
void run_pattern(int32_t id, pq_series_export *inputs, size_t n_inputs, pq_series_export *ret, bool over = false) {
    if (ret) memset(ret, 0, sizeof *ret);
    g_plugin_err.clear();
    if (!inputs || n_inputs < 4 || !ret) { plugin_fail("plugin: bad arguments (open, high, low, close expected)"); return; }
    for (int k = 0; k < 4; k++) {
        if (!inputs[k].field || !inputs[k].field->format || (inputs[k].len && !inputs[k].arrays)) { plugin_fail("plugin: bad arguments"); return; }
        if (!numeric_format(inputs[k].field->format)) { plugin_fail(k_not_numeric); return; }
    }
    double pen = 0.3;
    const size_t pi = over ? 5 : 4; // the penetration literal follows the key column of an _over call
    if (over && n_inputs < 5) { plugin_fail("plugin: bad arguments (an _over call takes open, high, low, close, key)"); return; }
    if (n_inputs > pi && inputs[pi].len >= 1 && inputs[pi].arrays && inputs[pi].arrays[0] && inputs[pi].arrays[0]->length >= 1 && inputs[pi].field &&
        inputs[pi].field->format && !strcmp(inputs[pi].field->format, "g") && inputs[pi].arrays[0]->n_buffers >= 2 && inputs[pi].arrays[0]->buffers[1])
        pen = ((const double *)inputs[pi].arrays[0]->buffers[1])[inputs[pi].arrays[0]->offset];
    const int64_t n = series_len(inputs[0]);
    std::vector<double> host[4];
    std::vector<uint8_t> valid;
    bool any_null = false;
    const double *flat[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 4; k++) {
        if (series_len(inputs[k]) != n) { plugin_fail("plugin: the input Series differ in length"); return; }
        if ((flat[k] = flat_f64(inputs[k], n))) continue;
        if (!gather_f64(inputs[k], n, host[k], valid, any_null)) { plugin_fail("plugin: malformed input chunk"); return; }
    }
    if (any_null) { plugin_fail("plugin: the input contains nulls (this function rejects them, as the reference's cont_slice() does)"); return; }
    OutPriv *op = new OutPriv();
    if (!op->ivalues.resize((size_t)(n > 0 ? n : 1))) { delete op; plugin_fail("plugin: out of host memory for the result column"); return; }
    if (n > 0) {
        pq_ctx *ctx = plugin_ctx();
        if (!ctx) { delete op; plugin_fail("plugin: no HIP device / context"); return; }
        void *d_in[4] = {nullptr, nullptr, nullptr, nullptr}, *d_out = nullptr;
        InCol ic[4];
        Layout lay;
        HashAhead ha;
        if (over) ha.start(flat, 4, n);
        pq_status st = plan_layout(ctx, over ? &inputs[4] : nullptr, n, &lay);
        ha.join();
        if (st == PQ_OK) st = pq_malloc(ctx, lay.dev_elems() * 4, &d_out);
        for (int k = 0; k < 4 && st == PQ_OK; k++) {
            if (flat[k]) st = cached_input(ctx, lay, flat[k], n, &ic[k], ha.get(k));
            else { st = upload_col(ctx, lay, host[k].data(), 8, &ic[k].d); ic[k].owned = true; }
            d_in[k] = ic[k].d;
        }
        if (st == PQ_OK) st = pq_cdl(ctx, &lay.b, id, (const double *)d_in[0], (const double *)d_in[1], (const double *)d_in[2], (const double *)d_in[3], pen, (int32_t *)d_out);
        if (st == PQ_OK) st = download_col(ctx, lay, d_out, 4, op->ivalues.data());
        if (st == PQ_OK) st = pq_ctx_sync(ctx);
        inputs_done(ctx, ic, 4, lay, st == PQ_OK);
        for (void *q : {d_out, lay.d_off}) if (q) (void)pq_free(ctx, q);
        if (st != PQ_OK) { delete op; plugin_fail("pq_cdl"); return; }
    }
    ArrowArray *arr = new ArrowArray();
    memset(arr, 0, sizeof *arr);
    op->bufs[0] = nullptr;
    op->bufs[1] = op->ivalues.data();
    arr->length = n; arr->null_count = 0; arr->n_buffers = 2; arr->buffers = op->bufs; arr->release = release_array; arr->private_data = op;
    ret->field = new ArrowSchema();
    fill_schema(ret->field, inputs[0].field->name ? inputs[0].field->name : "", "i");
    ret->arrays = new ArrowArray *[1];
    ret->arrays[0] = arr;
    ret->len = 1;
    ret->release = release_series;
}
void field_i32(ArrowSchema *fields, size_t n_fields, ArrowSchema *ret) {
    if (!ret) return;
    fill_schema(ret, (fields && n_fields >= 1 && fields[0].name) ? fields[0].name : "", "i");
}
void field_f64(ArrowSchema *fields, size_t n_fields, ArrowSchema *ret) {
    if (!ret) return;
    fill_schema(ret, (fields && n_fields >= 1 && fields[0].name) ? fields[0].name : "");
}
} // namespace

extern "C" {
// the input-column cache (above): counters since the library was loaded / the last clear; clear = free every entry no call holds
void pq_plugin_cache_stats(int64_t *hits, int64_t *misses, int64_t *bytes, int64_t *entries) {
    std::lock_guard<std::mutex> g(g_cache_mu);
    if (hits) *hits = g_cache_hits;
    if (misses) *misses = g_cache_misses;
    if (bytes) *bytes = (int64_t)g_cache_bytes;
    if (entries) *entries = (int64_t)g_cache.size();
}
void pq_plugin_cache_clear(void) {
    pq_ctx *ctx = plugin_ctx();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> g(g_cache_mu);
        for (size_t i = g_cache.size(); i-- > 0;)
            if (!g_cache[i].in_use) { drop.push_back(g_cache[i].d); g_cache_bytes -= g_cache[i].bytes; g_cache.erase(g_cache.begin() + (long)i); }
        g_cache_hits = g_cache_misses = 0;
    }
    if (ctx) for (void *q : drop) (void)pq_free(ctx, q);
}
uint32_t _polars_plugin_get_version(void) { return (0u << 16) | 1u; }
const char *_polars_plugin_get_last_error_message(void) { return g_plugin_err.c_str(); }
// Every reference function of the shape (1..4 Float64 columns[, timeperiod]) -> Float64.  overlap.rs takes the period as pickled
// kwargs (MaKwargs, overlap.rs:11-28), momentum.rs / volatility.rs as a trailing literal input; run_cols accepts both.
// X(name, number of input columns, takes timeperiod (1/0), default timeperiod, rejects nulls (N-B family))
#define PQ_PLUGIN_FUNCS(X)                                                                                                     \
    X(sma, 1, 1, 30, false) X(ema, 1, 1, 30, false) X(wma, 1, 1, 30, false) X(dema, 1, 1, 30, false) X(tema, 1, 1, 30, false)    \
    X(trima, 1, 1, 30, false) X(kama, 1, 1, 30, false) X(midpoint, 1, 1, 14, false) X(rsi, 1, 1, 14, true) X(cmo, 1, 1, 14, true) \
    X(mom, 1, 1, 10, true) X(roc, 1, 1, 10, true) X(rocp, 1, 1, 10, true) X(rocr, 1, 1, 10, true) X(rocr100, 1, 1, 10, true)      \
    X(trix, 1, 1, 30, true) X(ht_dcperiod, 1, 0, 0, true) X(ht_dcphase, 1, 0, 0, true) X(ht_trendline, 1, 0, 0, true)             \
    X(midprice, 2, 1, 14, false) X(plus_dm, 2, 1, 14, true) X(minus_dm, 2, 1, 14, true) X(aroonosc, 2, 1, 14, true)              \
    X(medprice, 2, 0, 0, false) X(obv, 2, 0, 0, false)                                                                           \
    X(adx, 3, 1, 14, true) X(adxr, 3, 1, 14, true) X(dx, 3, 1, 14, true) X(plus_di, 3, 1, 14, true) X(minus_di, 3, 1, 14, true)   \
    X(cci, 3, 1, 14, true) X(willr, 3, 1, 14, true) X(atr, 3, 1, 14, false) X(natr, 3, 1, 14, false) X(trange, 3, 0, 0, false)    \
    X(typprice, 3, 0, 0, false) X(wclprice, 3, 0, 0, false)                                                                      \
    X(mfi, 4, 1, 14, true) X(bop, 4, 0, 0, true) X(ad, 4, 0, 0, false) X(avgprice, 4, 0, 0, false)
#define PQ_ARGS_1(in) in[0]
#define PQ_ARGS_2(in) in[0], in[1]
#define PQ_ARGS_3(in) in[0], in[1], in[2]
#define PQ_ARGS_4(in) in[0], in[1], in[2], in[3]
#define PQ_CALL_1(NAME, NIN) pq_##NAME(ctx, b, PQ_ARGS_##NIN(in), (int64_t)pv[0], (double *)out)
#define PQ_CALL_0(NAME, NIN) pq_##NAME(ctx, b, PQ_ARGS_##NIN(in), (double *)out)
#define X(NAME, NIN, HAS_TP, DEFAULT, NB)                                                                                      \
    static pq_status plug_call_##NAME(pq_ctx *ctx, const pq_batch *b, const double *const *in, const double *pv, void *out) {   \
        (void)pv;                                                                                                              \
        return PQ_CALL_##HAS_TP(NAME, NIN);                                                                                    \
    }                                                                                                                          \
    void _polars_plugin_##NAME(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len,            \
                               pq_series_export *ret, void *) {                                                                \
        static const PlugFn f = {"pq_" #NAME, NIN, HAS_TP, {{"timeperiod", false, (double)DEFAULT}}, NB, false, &plug_call_##NAME}; \
        run_cols(f, inputs, n_inputs, kwargs, kwargs_len, ret);                                                                \
    }                                                                                                                          \
    void _polars_plugin_##NAME##_over(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len,     \
                                      pq_series_export *ret, void *) {                                                         \
        static const PlugFn f = {"pq_" #NAME, NIN, HAS_TP, {{"timeperiod", false, (double)DEFAULT}}, NB, false, &plug_call_##NAME}; \
        run_cols(f, inputs, n_inputs, kwargs, kwargs_len, ret, true);                                                          \
    }                                                                                                                          \
    void _polars_plugin_field_##NAME(ArrowSchema *fields, size_t n, ArrowSchema *ret, const uint8_t *, size_t) { field_f64(fields, n, ret); } \
    void _polars_plugin_field_##NAME##_over(ArrowSchema *fields, size_t n, ArrowSchema *ret, const uint8_t *, size_t) { field_f64(fields, n, ret); }
PQ_PLUGIN_FUNCS(X)
#undef X
// functions with other scalar parameters (reference defaults; overlap.rs:146-151 ma, :503-507 t3, :437-443 sar, :457-469 sarext;
// momentum.rs:572 ultosc; volume.rs:34 adosc; cycle.rs:377 ht_trendmode -> Int32)
#define PQ_PLUGIN_DEFINE(NAME, NIN, NPARAMS, PARAMS, NB, I32, CALL, FIELD)                                                       \
    static pq_status plug_call_##NAME(pq_ctx *ctx, const pq_batch *b, const double *const *in, const double *pv, void *out) {   \
        (void)pv;                                                                                                              \
        return CALL;                                                                                                           \
    }                                                                                                                          \
    void _polars_plugin_##NAME(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len,            \
                               pq_series_export *ret, void *) {                                                                \
        static const PlugFn f = {"pq_" #NAME, NIN, NPARAMS, PARAMS, NB, I32, &plug_call_##NAME};                               \
        run_cols(f, inputs, n_inputs, kwargs, kwargs_len, ret);                                                                \
    }                                                                                                                          \
    void _polars_plugin_##NAME##_over(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len,     \
                                      pq_series_export *ret, void *) {                                                         \
        static const PlugFn f = {"pq_" #NAME, NIN, NPARAMS, PARAMS, NB, I32, &plug_call_##NAME};                               \
        run_cols(f, inputs, n_inputs, kwargs, kwargs_len, ret, true);                                                          \
    }                                                                                                                          \
    void _polars_plugin_field_##NAME(ArrowSchema *fields, size_t n, ArrowSchema *ret, const uint8_t *, size_t) { FIELD(fields, n, ret); } \
    void _polars_plugin_field_##NAME##_over(ArrowSchema *fields, size_t n, ArrowSchema *ret, const uint8_t *, size_t) { FIELD(fields, n, ret); }
#define PQ_P(...) {__VA_ARGS__}
PQ_PLUGIN_DEFINE(ma, 1, 2, PQ_P({"timeperiod", false, 30.0}, {"matype", false, 0.0}), false, false,
                 pq_ma(ctx, b, in[0], (int64_t)pv[0], (int64_t)pv[1], (double *)out), field_f64)
PQ_PLUGIN_DEFINE(t3, 1, 2, PQ_P({"timeperiod", false, 5.0}, {"vfactor", true, 0.0}), false, false,
                 pq_t3(ctx, b, in[0], (int64_t)pv[0], pv[1], (double *)out), field_f64)
PQ_PLUGIN_DEFINE(ultosc, 3, 3, PQ_P({"timeperiod1", false, 7.0}, {"timeperiod2", false, 14.0}, {"timeperiod3", false, 28.0}), true, false,
                 pq_ultosc(ctx, b, in[0], in[1], in[2], (int64_t)pv[0], (int64_t)pv[1], (int64_t)pv[2], (double *)out), field_f64)
PQ_PLUGIN_DEFINE(adosc, 4, 2, PQ_P({"fastperiod", false, 3.0}, {"slowperiod", false, 10.0}), false, false,
                 pq_adosc(ctx, b, in[0], in[1], in[2], in[3], (int64_t)pv[0], (int64_t)pv[1], (double *)out), field_f64)
PQ_PLUGIN_DEFINE(sar, 2, 2, PQ_P({"acceleration", true, 0.0}, {"maximum", true, 0.0}), false, false,
                 pq_sar(ctx, b, in[0], in[1], pv[0], pv[1], (double *)out), field_f64)
PQ_PLUGIN_DEFINE(sarext, 2, 8,
                 PQ_P({"startvalue", true, 0.0}, {"offsetonreverse", true, 0.0}, {"accelerationinitlong", true, 0.0}, {"accelerationlong", true, 0.0},
                      {"accelerationmaxlong", true, 0.0}, {"accelerationinitshort", true, 0.0}, {"accelerationshort", true, 0.0},
                      {"accelerationmaxshort", true, 0.0}), false, false,
                 pq_sarext(ctx, b, in[0], in[1], pv[0], pv[1], pv[2], pv[3], pv[4], pv[5], pv[6], pv[7], (double *)out), field_f64)
// mavp: the period column is cast to Int64 by the reference (overlap.rs:407-414): Int64 / Int32 / Float64 are accepted
static pq_status plug_call_mavp(pq_ctx *ctx, const pq_batch *b, const double *const *in, const double *pv, void *out) {
    return pq_mavp(ctx, b, in[0], in[1], (int64_t)pv[0], (int64_t)pv[1], (int64_t)pv[2], (double *)out);
}
void _polars_plugin_mavp(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len, pq_series_export *ret, void *) {
    static const PlugFn f = {"pq_mavp", 2, 3, {{"minperiod", false, 2.0}, {"maxperiod", false, 30.0}, {"matype", false, 0.0}}, false, false, &plug_call_mavp, 2u};
    run_cols(f, inputs, n_inputs, kwargs, kwargs_len, ret);
}
void _polars_plugin_mavp_over(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len, pq_series_export *ret, void *) {
    static const PlugFn f = {"pq_mavp", 2, 3, {{"minperiod", false, 2.0}, {"maxperiod", false, 30.0}, {"matype", false, 0.0}}, false, false, &plug_call_mavp, 2u};
    run_cols(f, inputs, n_inputs, kwargs, kwargs_len, ret, true);
}
void _polars_plugin_field_mavp(ArrowSchema *fields, size_t n, ArrowSchema *ret, const uint8_t *, size_t) { field_f64(fields, n, ret); }
void _polars_plugin_field_mavp_over(ArrowSchema *fields, size_t n, ArrowSchema *ret, const uint8_t *, size_t) { field_f64(fields, n, ret); }
// apo / ppo: registered by the reference's Python (momentum.py:25-30, :136-141: args real, fastperiod, slowperiod, matype) although
// no Rust function of that name exists; decision D-6 defines them on the reference's own MA family
PQ_PLUGIN_DEFINE(apo, 1, 3, PQ_P({"fastperiod", false, 12.0}, {"slowperiod", false, 26.0}, {"matype", false, 0.0}), false, false,
                 pq_apo(ctx, b, in[0], (int64_t)pv[0], (int64_t)pv[1], (int64_t)pv[2], (double *)out), field_f64)
PQ_PLUGIN_DEFINE(ppo, 1, 3, PQ_P({"fastperiod", false, 12.0}, {"slowperiod", false, 26.0}, {"matype", false, 0.0}), false, false,
                 pq_ppo(ctx, b, in[0], (int64_t)pv[0], (int64_t)pv[1], (int64_t)pv[2], (double *)out), field_f64)
PQ_PLUGIN_DEFINE(ht_trendmode, 1, 0, PQ_P({nullptr, false, 0.0}), true, true, pq_ht_trendmode(ctx, b, in[0], (int32_t *)out), field_i32)
// Struct-valued functions: the struct and field names are the reference's (overlap.rs:30-44 bbands / mama, momentum.rs:63-66 aroon,
// :239-246 macd -> "macd_res", cycle.rs:149-155 ht_phasor, :229-232 ht_sine)
#define PQ_PLUGIN_STRUCT(NAME, SNAME, NIN, NPARAMS, PARAMS, NB, NOUT, FIELDS, CALL)                                              \
    static pq_status plug_call_##NAME(pq_ctx *ctx, const pq_batch *b, const double *const *in, const double *pv, double *const *out) { \
        (void)pv;                                                                                                              \
        return CALL;                                                                                                           \
    }                                                                                                                          \
    static const StructFn k_struct_##NAME = {"pq_" #NAME, SNAME, NIN, NPARAMS, PARAMS, NB, NOUT, FIELDS, &plug_call_##NAME};   \
    void _polars_plugin_##NAME(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len,            \
                               pq_series_export *ret, void *) {                                                                \
        run_struct(k_struct_##NAME, inputs, n_inputs, kwargs, kwargs_len, ret);                                                \
    }                                                                                                                          \
    void _polars_plugin_##NAME##_over(pq_series_export *inputs, size_t n_inputs, const uint8_t *kwargs, size_t kwargs_len,     \
                                      pq_series_export *ret, void *) {                                                         \
        run_struct(k_struct_##NAME, inputs, n_inputs, kwargs, kwargs_len, ret, true);                                          \
    }                                                                                                                          \
    void _polars_plugin_field_##NAME(ArrowSchema *, size_t, ArrowSchema *ret, const uint8_t *, size_t) {                       \
        if (ret) fill_struct_schema(ret, k_struct_##NAME);                                                                     \
    }                                                                                                                          \
    void _polars_plugin_field_##NAME##_over(ArrowSchema *, size_t, ArrowSchema *ret, const uint8_t *, size_t) {                \
        if (ret) fill_struct_schema(ret, k_struct_##NAME);                                                                     \
    }
PQ_PLUGIN_STRUCT(bbands, "bbands", 1, 3, PQ_P({"timeperiod", false, 20.0}, {"nbdevup", true, 2.0}, {"nbdevdn", true, 2.0}), false, 3,
                 PQ_P("bb_upper", "bb_middle", "bb_lower"), pq_bbands(ctx, b, in[0], (int64_t)pv[0], pv[1], pv[2], out[0], out[1], out[2]))
PQ_PLUGIN_STRUCT(mama, "mama", 1, 2, PQ_P({"fastlimit", true, 0.0}, {"slowlimit", true, 0.0}), false, 2, PQ_P("mama", "fama"),
                 pq_mama(ctx, b, in[0], pv[0], pv[1], out[0], out[1]))
PQ_PLUGIN_STRUCT(aroon, "aroon", 2, 1, PQ_P({"timeperiod", false, 14.0}), true, 2, PQ_P("aroon_up", "aroon_down"),
                 pq_aroon(ctx, b, in[0], in[1], (int64_t)pv[0], out[0], out[1]))
PQ_PLUGIN_STRUCT(macd, "macd_res", 1, 3, PQ_P({"fastperiod", false, 12.0}, {"slowperiod", false, 26.0}, {"signalperiod", false, 9.0}), true, 3,
                 PQ_P("macd", "macd_signal", "macd_hist"), pq_macd(ctx, b, in[0], (int64_t)pv[0], (int64_t)pv[1], (int64_t)pv[2], out[0], out[1], out[2]))
PQ_PLUGIN_STRUCT(ht_phasor, "ht_phasor", 1, 0, PQ_P({nullptr, false, 0.0}), true, 2, PQ_P("inphase", "quadrature"),
                 pq_ht_phasor(ctx, b, in[0], out[0], out[1]))
PQ_PLUGIN_STRUCT(ht_sine, "ht_sine", 1, 0, PQ_P({nullptr, false, 0.0}), true, 2, PQ_P("sine", "leadsine"), pq_ht_sine(ctx, b, in[0], out[0], out[1]))
// the 61 recognisers in id order (pq_pattern_name)
#define PQ_PLUGIN_PATTERNS(X) \
    X(cdl2crows, 0) X(cdl3blackcrows, 1) X(cdl3inside, 2) X(cdl3linestrike, 3) X(cdl3outside, 4) X(cdl3starsinsouth, 5) \
    X(cdl3whitesoldiers, 6) X(cdlabandonedbaby, 7) X(cdladvanceblock, 8) X(cdlbelthold, 9) X(cdlbreakaway, 10) \
    X(cdlclosingmarubozu, 11) X(cdlconcealbabyswall, 12) X(cdlcounterattack, 13) X(cdldarkcloudcover, 14) \
    X(cdldoji, 15) X(cdldojistar, 16) X(cdldragonflydoji, 17) X(cdlengulfing, 18) X(cdleveningdojistar, 19) \
    X(cdleveningstar, 20) X(cdlgapsidesidewhite, 21) X(cdlgravestonedoji, 22) X(cdlhammer, 23) X(cdlhangingman, 24) \
    X(cdlharami, 25) X(cdlharamicross, 26) X(cdlhighwave, 27) X(cdlhikkake, 28) X(cdlhikkakemod, 29) \
    X(cdlhomingpigeon, 30) X(cdlidentical3crows, 31) X(cdlinneck, 32) X(cdlinvertedhammer, 33) X(cdlkicking, 34) \
    X(cdlkickingbylength, 35) X(cdlladderbottom, 36) X(cdllongleggeddoji, 37) X(cdllongline, 38) X(cdlmarubozu, 39) \
    X(cdlmatchinglow, 40) X(cdlmathold, 41) X(cdlmorningdojistar, 42) X(cdlmorningstar, 43) X(cdlonneck, 44) \
    X(cdlpiercing, 45) X(cdlrickshawman, 46) X(cdlrisefall3methods, 47) X(cdlseparatinglines, 48) \
    X(cdlshootingstar, 49) X(cdlshortline, 50) X(cdlspinningtop, 51) X(cdlstalledpattern, 52) X(cdlsticksandwich, 53) \
    X(cdltakuri, 54) X(cdltasukigap, 55) X(cdlthrusting, 56) X(cdltristar, 57) X(cdlunique3river, 58) \
    X(cdlupsidegap2crows, 59) X(cdlxsidegap3methods, 60)
#define X(NAME, ID)                                                                                                            \
    void _polars_plugin_##NAME(pq_series_export *inputs, size_t n_inputs, const uint8_t *, size_t, pq_series_export *ret, void *) { \
        run_pattern(ID, inputs, n_inputs, ret);                                                                                \
    }                                                                                                                          \
    void _polars_plugin_##NAME##_over(pq_series_export *inputs, size_t n_inputs, const uint8_t *, size_t, pq_series_export *ret, void *) { \
        run_pattern(ID, inputs, n_inputs, ret, true);                                                                          \
    }                                                                                                                          \
    void _polars_plugin_field_##NAME(ArrowSchema *fields, size_t n, ArrowSchema *ret, const uint8_t *, size_t) { field_i32(fields, n, ret); } \
    void _polars_plugin_field_##NAME##_over(ArrowSchema *fields, size_t n, ArrowSchema *ret, const uint8_t *, size_t) { field_i32(fields, n, ret); }
PQ_PLUGIN_PATTERNS(X)
#undef X
}
