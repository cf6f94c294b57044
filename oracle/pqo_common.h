/* pqo_common.h -- helpers shared by the oracle's translation units (test infrastructure). */
#ifndef PQO_COMMON_H
#define PQO_COMMON_H
#include "pq_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline double pqo_null(void) {
    uint64_t b = PQO_NULL_BITS;
    double d;
    memcpy(&d, &b, 8);
    return d;
}
static inline int pqo_isnull(double x) {
    uint64_t b;
    memcpy(&b, &x, 8);
    return b == PQO_NULL_BITS;
}
static inline void pqo_fill_null(double *out, int64_t n) {
    for (int64_t i = 0; i < n; i++) out[i] = pqo_null();
}

/* Rust f64::max / f64::min: if one operand is NaN the other is returned == C fmax/fmin */
#define RMAX(a, b) fmax((a), (b))
#define RMIN(a, b) fmin((a), (b))

/* std::collections::VecDeque<f64> restated as a grow-never ring over a flat array of capacity n+1 */
typedef struct {
    double *buf;
    int64_t head, tail; /* [head, tail) */
} pqo_deque;
static inline void dq_init(pqo_deque *d, int64_t cap) {
    d->buf = (double *)malloc(sizeof(double) * (size_t)(cap + 1));
    d->head = d->tail = 0;
}
static inline void dq_free(pqo_deque *d) { free(d->buf); }
static inline void dq_push_back(pqo_deque *d, double v) { d->buf[d->tail++] = v; }
static inline int dq_empty(const pqo_deque *d) { return d->head == d->tail; }
static inline double dq_pop_front(pqo_deque *d) { return d->buf[d->head++]; }
static inline double dq_front(const pqo_deque *d) { return d->buf[d->head]; }

/* VecDeque<(usize, f64)> used by the monotonic windows */
typedef struct {
    uint64_t *idx;
    double *val;
    int64_t head, tail;
} pqo_ideque;
static inline void idq_init(pqo_ideque *d, int64_t cap) {
    d->idx = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(cap + 1));
    d->val = (double *)malloc(sizeof(double) * (size_t)(cap + 1));
    d->head = d->tail = 0;
}
static inline void idq_free(pqo_ideque *d) {
    free(d->idx);
    free(d->val);
}
static inline int idq_empty(const pqo_ideque *d) { return d->head == d->tail; }

#endif
