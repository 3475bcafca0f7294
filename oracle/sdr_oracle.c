/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE SHIPPED ENGINE.  See sdr_oracle.h.
 *
 * Plain-C restatement of the reference's single-pair set algebra.  Build with
 *   gcc -O2 -std=c11 -fno-fast-math -ffp-contract=off
 * (deliberately not the reference's -Ofast, CMakeLists.txt:21): every payload op is a
 * single IEEE binary32 operation and must round exactly once.
 */
#define _POSIX_C_SOURCE 199309L
#include "sdr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ payloads */

/* UnitData.hpp:22-24: `value_ >= 0.1` promotes the float to double and compares with
 * the double literal. */
static int unit_relevant(float v) { return (double)v >= 0.1; }

/* relevant(): EmptyData.hpp:29-32 (always), UnitData.hpp:22-24, ArithData.hpp:22-24 (always). */
static int relevant(int kind, float v) { return kind == OSDR_UNIT ? unit_relevant(v) : 1; }

/* rm_relevant(): EmptyData.hpp:34-37 (never), UnitData.hpp:26-28, ArithData.hpp:26-28 (= relevant: always). */
static int rm_relevant(int kind, float v) { return kind == OSDR_EMPTY ? 0 : kind == OSDR_UNIT ? unit_relevant(v) : 1; }

/* andi: UnitData.hpp:37-46, ArithData.hpp:37-43 (multiply; identity against EmptyData). */
static float data_and(int tk, int ak, float tv, float av) {
    if (tk != OSDR_EMPTY && ak != OSDR_EMPTY) return tv * av;
    return tv;
}
/* ori: UnitData.hpp:60-69 (the larger), ArithData.hpp:57-63 (the sum). */
static float data_or(int tk, int ak, float tv, float av) {
    if (tk == OSDR_UNIT && ak == OSDR_UNIT) return tv > av ? tv : av;
    if (tk == OSDR_ARITH && ak == OSDR_ARITH) return tv + av;
    return tv;
}
/* xori: UnitData.hpp:83-92 (std::abs on float). */
static float data_xor(int tk, int ak, float tv, float av) {
    if (tk == OSDR_UNIT && ak == OSDR_UNIT) return fabsf(tv - av);
    return tv;
}
/* rmi: UnitData.hpp:106-115 (`1 - o.value()` is a float subtraction, then a multiply);
 * ArithData.hpp:77-83 (the difference).  ArithData has no xor (ArithData.hpp:110-114). */
static float data_rm(int tk, int ak, float tv, float av) {
    if (tk == OSDR_UNIT && ak == OSDR_UNIT) {
        float one_minus = 1 - av;
        return tv * one_minus;
    }
    if (tk == OSDR_ARITH && ak == OSDR_ARITH) return tv - av;
    return tv;
}
/* Converting ctor SDRElem.hpp:29-30: UnitData(EmptyData) is 1 (UnitData.hpp:13-14), ArithData(EmptyData)
 * is 0 (ArithData.hpp:15-16); EmptyData swallows anything (EmptyData.hpp:25-27). */
static float data_convert(int tk, int ak, float av) {
    if (tk == OSDR_UNIT) return ak == OSDR_UNIT ? av : 1.0f;
    if (tk == OSDR_ARITH) return ak == OSDR_ARITH ? av : 0.0f;
    return 0.0f;
}

static float vdata(const osdr_view* s, size_t i) { return s->kind != OSDR_EMPTY ? s->data[i] : 0.0f; }

/* ------------------------------------------------------------------ merges */

/* std::lower_bound over [first,last) of an ascending int array (libstdc++ halving). */
static size_t lower_bound_i32(const int32_t* a, size_t first, size_t last, int32_t val) {
    size_t len = last - first;
    while (len > 0) {
        size_t half = len >> 1;
        size_t mid = first + half;
        if (a[mid] < val) {
            first = mid + 1;
            len = len - half - 1;
        } else {
            len = half;
        }
    }
    return first;
}

/* SDR.hpp:958-997 */
void osdr_andv(const int32_t* t, size_t tn, const int32_t* a, size_t an,
               osdr_and_visitor visitor, void* ctx) {
    size_t tp = 0, ap = 0;
    if (tp == tn) return;                         /* :964 */
    for (;;) {
        int32_t te = t[tp];                       /* :967 */
        ap = lower_bound_i32(a, ap, an, te);      /* :972 */
        if (ap == an) return;                     /* :974 */
        if (a[ap] == te) {                        /* :976 */
            size_t tv = tp++, av = ap++;
            visitor(ctx, tv, av);                 /* :977 */
            if (ap == an) return;                 /* :978 */
        } else {
            ++tp;                                 /* :980 */
        }
        int32_t ae = a[ap];                       /* :983 */
        tp = lower_bound_i32(t, tp, tn, ae);      /* :987 */
        if (tp == tn) return;                     /* :989 */
        if (t[tp] == ae) {                        /* :990 */
            size_t tv = tp++, av = ap++;
            visitor(ctx, tv, av);                 /* :991 */
            if (tp == tn) return;                 /* :992 */
        } else {
            ++ap;                                 /* :994 */
        }
    }
}

/* SDR.hpp:1099-1150 with the default tail-draining positions of :1153-1166. */
void osdr_orv(const int32_t* t, size_t tn, const int32_t* a, size_t an,
              osdr_one_visitor this_visitor, osdr_one_visitor arg_visitor,
              osdr_and_visitor both_visitor, void* ctx) {
    size_t tp = 0, ap = 0;
    int32_t tv = 0, av = 0;
#define DRAIN_ARG()  do { while (ap != an) arg_visitor(ctx, ap++); } while (0)   /* :1160-1164 */
#define DRAIN_THIS() do { while (tp != tn) this_visitor(ctx, tp++); } while (0)  /* :1155-1159 */
    if (tp != tn) tv = t[tp]; else { DRAIN_ARG(); return; }       /* get_this :1106-1114 */
    if (ap != an) av = a[ap]; else { DRAIN_THIS(); return; }      /* get_arg  :1116-1124 */
    for (;;) {                                                    /* :1136-1148 */
        if (tv < av) {
            this_visitor(ctx, tp++);
            if (tp != tn) tv = t[tp]; else { DRAIN_ARG(); return; }
        } else if (tv > av) {
            arg_visitor(ctx, ap++);
            if (ap != an) av = a[ap]; else { DRAIN_THIS(); return; }
        } else {
            size_t tq = tp++, aq = ap++;
            both_visitor(ctx, tq, aq);
            if (tp != tn) tv = t[tp]; else { DRAIN_ARG(); return; }
            if (ap != an) av = a[ap]; else { DRAIN_THIS(); return; }
        }
    }
#undef DRAIN_ARG
#undef DRAIN_THIS
}

/* ------------------------------------------------------------------ size ops */

typedef struct {
    const osdr_view* t;
    const osdr_view* a;
    size_t acc;
} size_ctx;

/* SDR.hpp:1087-1091; payload ands: EmptyData.hpp:50-51, UnitData.hpp:48-51. */
static void ands_visit(void* p, size_t tp, size_t ap) {
    size_ctx* c = (size_ctx*)p;
    float d = data_and(c->t->kind, c->a->kind, vdata(c->t, tp), vdata(c->a, ap));
    if (relevant(c->t->kind, d)) ++c->acc;
}
/* SDR.hpp:1298-1301: every id match counts, payloads ignored. */
static void ors_visit(void* p, size_t tp, size_t ap) {
    (void)tp; (void)ap;
    ++((size_ctx*)p)->acc;
}
/* SDR.hpp:1428-1431; payload xors: EmptyData.hpp:70-71, UnitData.hpp:94-97. */
static void xors_visit(void* p, size_t tp, size_t ap) {
    size_ctx* c = (size_ctx*)p;
    float d = data_xor(c->t->kind, c->a->kind, vdata(c->t, tp), vdata(c->a, ap));
    if (!rm_relevant(c->t->kind, d)) ++c->acc;
    ++c->acc;
}
/* SDR.hpp:1572-1574; payload rms: EmptyData.hpp:79-80, UnitData.hpp:117-120. */
static void rms_visit(void* p, size_t tp, size_t ap) {
    size_ctx* c = (size_ctx*)p;
    float d = data_rm(c->t->kind, c->a->kind, vdata(c->t, tp), vdata(c->a, ap));
    if (!rm_relevant(c->t->kind, d)) ++c->acc;
}

size_t osdr_size_op(int op, const osdr_view* t, const osdr_view* a) {
    size_ctx c = {t, a, 0};
    switch (op) {
    case OSDR_AND: /* SDR.hpp:1083-1095 */
        osdr_andv(t->ids, t->n, a->ids, a->n, ands_visit, &c);
        return c.acc;
    case OSDR_OR: /* SDR.hpp:1294-1304 */
        osdr_andv(t->ids, t->n, a->ids, a->n, ors_visit, &c);
        return t->n + a->n - c.acc;
    case OSDR_XOR: /* SDR.hpp:1424-1434 */
        osdr_andv(t->ids, t->n, a->ids, a->n, xors_visit, &c);
        return t->n + a->n - c.acc;
    case OSDR_RM: /* SDR.hpp:1568-1577 */
        osdr_andv(t->ids, t->n, a->ids, a->n, rms_visit, &c);
        return t->n - c.acc;
    }
    return 0;
}

/* ------------------------------------------------------------------ element ops */

typedef struct {
    const osdr_view* t;
    const osdr_view* a;
    int32_t* out_ids;
    float* out_data;
    size_t n;         /* result length so far */
    size_t this_fill; /* rme's trailing cursor (SDR.hpp:1520) */
} elem_ctx;

static void push(elem_ctx* c, int32_t id, float d) {
    c->out_ids[c->n] = id;
    if (c->out_data) c->out_data[c->n] = d;
    ++c->n;
}

/* SDR.hpp:1015-1021 */
static void ande_visit(void* p, size_t tp, size_t ap) {
    elem_ctx* c = (elem_ctx*)p;
    float d = data_and(c->t->kind, c->a->kind, vdata(c->t, tp), vdata(c->a, ap));
    if (relevant(c->t->kind, d)) push(c, c->t->ids[tp], d);
}
/* SDR.hpp:1207-1209 / 1332-1334 */
static void copy_this_visit(void* p, size_t tp) {
    elem_ctx* c = (elem_ctx*)p;
    push(c, c->t->ids[tp], vdata(c->t, tp));
}
/* SDR.hpp:1210-1212 / 1335-1337: ret_t(*arg_pos) converts the element type. */
static void copy_arg_visit(void* p, size_t ap) {
    elem_ctx* c = (elem_ctx*)p;
    push(c, c->a->ids[ap], data_convert(c->t->kind, c->a->kind, vdata(c->a, ap)));
}
/* SDR.hpp:1213-1217: no relevance check on a union match. */
static void ore_both_visit(void* p, size_t tp, size_t ap) {
    elem_ctx* c = (elem_ctx*)p;
    push(c, c->t->ids[tp], data_or(c->t->kind, c->a->kind, vdata(c->t, tp), vdata(c->a, ap)));
}
/* SDR.hpp:1338-1344 */
static void xore_both_visit(void* p, size_t tp, size_t ap) {
    elem_ctx* c = (elem_ctx*)p;
    float d = data_xor(c->t->kind, c->a->kind, vdata(c->t, tp), vdata(c->a, ap));
    if (rm_relevant(c->t->kind, d)) push(c, c->t->ids[tp], d);
}
/* SDR.hpp:1540-1550 */
static void rme_both_visit(void* p, size_t tp, size_t ap) {
    elem_ctx* c = (elem_ctx*)p;
    while (c->this_fill != tp) {
        push(c, c->t->ids[c->this_fill], vdata(c->t, c->this_fill));
        ++c->this_fill;
    }
    float d = data_rm(c->t->kind, c->a->kind, vdata(c->t, tp), vdata(c->a, ap));
    if (rm_relevant(c->t->kind, d)) push(c, c->t->ids[tp], d);
    ++c->this_fill;
}

size_t osdr_elem_op(int op, const osdr_view* t, const osdr_view* a,
                    int32_t* out_ids, float* out_data) {
    elem_ctx c = {t, a, out_ids, t->kind != OSDR_EMPTY ? out_data : NULL, 0, 0};
    switch (op) {
    case OSDR_AND: /* SDR.hpp:999-1025 */
        osdr_andv(t->ids, t->n, a->ids, a->n, ande_visit, &c);
        break;
    case OSDR_OR: /* SDR.hpp:1183-1222 */
        osdr_orv(t->ids, t->n, a->ids, a->n, copy_this_visit, copy_arg_visit, ore_both_visit, &c);
        break;
    case OSDR_XOR: /* SDR.hpp:1307-1349 */
        osdr_orv(t->ids, t->n, a->ids, a->n, copy_this_visit, copy_arg_visit, xore_both_visit, &c);
        break;
    case OSDR_RM: /* SDR.hpp:1515-1566 */
        osdr_andv(t->ids, t->n, a->ids, a->n, rme_both_visit, &c);
        while (c.this_fill != t->n) { /* :1553-1560 */
            push(&c, t->ids[c.this_fill], vdata(t, c.this_fill));
            ++c.this_fill;
        }
        break;
    }
    return c.n;
}

/* ------------------------------------------------------------------ in-place ops */

typedef struct {
    int32_t* ids;
    float* data;
    int kind;
    const osdr_view* a;
    size_t pos;       /* andi write cursor (SDR.hpp:1031) / rmi insert_pos (:1453) */
    size_t this_fill; /* rmi (:1452) */
} inplace_ctx;

/* SDR.hpp:1032-1041 */
static void andi_visit(void* p, size_t tp, size_t ap) {
    inplace_ctx* c = (inplace_ctx*)p;
    float d = 0.0f;
    if (c->kind != OSDR_EMPTY) {
        d = data_and(c->kind, c->a->kind, c->data[tp], vdata(c->a, ap));
        c->data[tp] = d; /* andi mutates this's payload in place (:1033) */
    }
    if (relevant(c->kind, d)) {
        if (c->pos != tp) {
            c->ids[c->pos] = c->ids[tp];
            if (c->kind != OSDR_EMPTY) c->data[c->pos] = d;
        }
        ++c->pos;
    }
}
/* SDR.hpp:1454-1467 */
static void rmi_visit(void* p, size_t tp, size_t ap) {
    inplace_ctx* c = (inplace_ctx*)p;
    while (c->this_fill != tp) {
        c->ids[c->pos] = c->ids[c->this_fill];
        if (c->kind != OSDR_EMPTY) c->data[c->pos] = c->data[c->this_fill];
        ++c->pos;
        ++c->this_fill;
    }
    float d = 0.0f;
    if (c->kind != OSDR_EMPTY) {
        d = data_rm(c->kind, c->a->kind, c->data[tp], vdata(c->a, ap));
        c->data[tp] = d; /* rmi_combine mutates in place (:1439-1443) */
    }
    if (rm_relevant(c->kind, d)) {
        c->ids[c->pos] = c->ids[tp];
        if (c->kind != OSDR_EMPTY) c->data[c->pos] = d;
        ++c->pos;
    }
    ++c->this_fill;
}

size_t osdr_inplace_op(int op, int32_t* t_ids, float* t_data, size_t t_n, int t_kind,
                       const osdr_view* a) {
    inplace_ctx c = {t_ids, t_data, t_kind, a, 0, 0};
    switch (op) {
    case OSDR_AND: /* SDR.hpp:1030-1043 */
        osdr_andv(t_ids, t_n, a->ids, a->n, andi_visit, &c);
        return c.pos;
    case OSDR_RM: /* SDR.hpp:1450-1472 */
        osdr_andv(t_ids, t_n, a->ids, a->n, rmi_visit, &c);
        while (c.this_fill != t_n) {
            c.ids[c.pos] = c.ids[c.this_fill];
            if (c.kind != OSDR_EMPTY) c.data[c.pos] = c.data[c.this_fill];
            ++c.pos;
            ++c.this_fill;
        }
        return c.pos;
    case OSDR_OR:    /* SDR.hpp:1242-1244: r = ore(move(this), arg); this = move(r) */
    case OSDR_XOR: { /* SDR.hpp:1369-1371 */
        size_t cap = t_n + a->n;
        int32_t* r_ids = (int32_t*)malloc((cap ? cap : 1) * sizeof(int32_t));
        float* r_data = (float*)malloc((cap ? cap : 1) * sizeof(float));
        osdr_view t = {t_ids, t_data, t_n, t_kind};
        size_t n = osdr_elem_op(op, &t, a, r_ids, r_data);
        memcpy(t_ids, r_ids, n * sizeof(int32_t));
        if (t_kind != OSDR_EMPTY) memcpy(t_data, r_data, n * sizeof(float));
        free(r_ids);
        free(r_data);
        return n;
    }
    }
    return 0;
}

/* ------------------------------------------------------------------ batches */

static osdr_view row_view(const uint64_t* off, const int32_t* ids, const float* data, int kind,
                          uint64_t r) {
    osdr_view v;
    v.ids = ids + off[r];
    v.data = (kind != OSDR_EMPTY && data) ? data + off[r] : NULL;
    v.n = (size_t)(off[r + 1] - off[r]);
    v.kind = kind;
    return v;
}

/* ---- range and point queries (SDR.hpp:860-956) ------------------------------------------ */
static size_t lower_bound_id(const int32_t* ids, size_t n, int32_t key) {   /* std::lower_bound */
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (ids[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

void osdr_batch_range_size(const uint64_t* off, const int32_t* ids, uint64_t n, uint64_t n_ranges,
                           const int32_t* lo, const int32_t* hi, uint32_t* counts, uint32_t* firsts) {
    for (uint64_t r = 0; r < n_ranges; ++r)
        for (uint64_t j = 0; j < n; ++j) {
            const int32_t* row = ids + off[j];
            size_t len = (size_t)(off[j + 1] - off[j]);
            size_t pos = lower_bound_id(row, len, lo[r]);                       /* SDR.hpp:945 */
            size_t end = pos + lower_bound_id(row + pos, len - pos, hi[r]);     /* SDR.hpp:946 */
            counts[r * n + j] = (uint32_t)(end - pos);
            if (firsts) firsts[r * n + j] = (uint32_t)pos;
        }
}

void osdr_batch_point(const uint64_t* off, const int32_t* ids, uint64_t n, uint64_t n_ids,
                      const int32_t* qid, int32_t* position) {
    for (uint64_t q = 0; q < n_ids; ++q)
        for (uint64_t j = 0; j < n; ++j) {
            const int32_t* row = ids + off[j];
            size_t len = (size_t)(off[j + 1] - off[j]);
            size_t pos = lower_bound_id(row, len, qid[q]);                      /* SDR.hpp:866 */
            position[q * n + j] = (pos == len || row[pos] != qid[q]) ? -1 : (int32_t)pos;   /* SDR.hpp:868-872 */
        }
}

/* ---- inner (SDR.hpp:1682-1692) ---------------------------------------------------------- */
typedef struct {
    const float* t_data;
    const float* a_data;
    float acc;
} inner_ctx;

static void inner_visit(void* vctx, size_t this_pos, size_t arg_pos) {
    inner_ctx* c = (inner_ctx*)vctx;
    /* elem = this_pos->data().ande(arg_pos->data()): value_ *= o.value() unless o is EmptyData
     * (ArithData.hpp:30-43); one rounding */
    volatile float elem = c->t_data[this_pos];
    if (c->a_data) elem = elem * c->a_data[arg_pos];
    /* ret.ori(elem): value_ += o.value() (ArithData.hpp:53-63); a second rounding, never fused */
    volatile float sum = c->acc + elem;
    c->acc = sum;
}

void osdr_batch_inner(const uint64_t* t_off, const int32_t* t_ids, const float* t_data, uint64_t t_n,
                      const uint64_t* a_off, const int32_t* a_ids, const float* a_data, uint64_t a_n,
                      float* out) {
    for (uint64_t i = 0; i < t_n; ++i)
        for (uint64_t j = 0; j < a_n; ++j) {
            inner_ctx c;
            c.t_data = t_data + t_off[i];
            c.a_data = a_data ? a_data + a_off[j] : NULL;
            c.acc = 0.0f;   /* ArithData() (ArithData.hpp:16) */
            osdr_andv(t_ids + t_off[i], (size_t)(t_off[i + 1] - t_off[i]), a_ids + a_off[j],
                      (size_t)(a_off[j + 1] - a_off[j]), inner_visit, &c);
            out[i * a_n + j] = c.acc;
        }
}

/* ---- operator== (SDR.hpp:470-483) --------------------------------------------------------- */
/* payload != : float compare when both sides carry a payload of the same kind (UnitData.hpp:122-128,
 * ArithData.hpp:90-96), never unequal against EmptyData (UnitData.hpp:130-136, EmptyData.hpp:82-83). */
void osdr_batch_equal(const uint64_t* t_off, const int32_t* t_ids, const float* t_data, int t_kind, uint64_t t_n,
                      const uint64_t* a_off, const int32_t* a_ids, const float* a_data, int a_kind, uint64_t a_n,
                      uint32_t* out) {
    const int payloads = t_kind != OSDR_EMPTY && a_kind != OSDR_EMPTY;
    for (uint64_t i = 0; i < t_n; ++i)
        for (uint64_t j = 0; j < a_n; ++j) {
            uint64_t tp = t_off[i], te = t_off[i + 1], ap = a_off[j], ae = a_off[j + 1];
            uint32_t same;
            for (;;) {
                if (tp == te) { same = ap == ae; break; }                  /* :477 */
                if (ap == ae) { same = 0; break; }                         /* :478 */
                if (t_ids[tp] != a_ids[ap] || (payloads && t_data[tp] != a_data[ap])) { same = 0; break; }   /* :481 */
                ++tp;
                ++ap;
            }
            out[i * a_n + j] = same;
        }
}

/* ---- the visit sequences themselves (what andv / orv hand to their callbacks) ------------- */
typedef struct {
    uint32_t* this_pos;
    uint32_t* arg_pos;
    uint64_t n;
} visit_ctx;

static void rec_both(void* p, size_t tp, size_t ap) {
    visit_ctx* c = (visit_ctx*)p;
    if (c->this_pos) { c->this_pos[c->n] = (uint32_t)tp; c->arg_pos[c->n] = (uint32_t)ap; }
    c->n++;
}
static void rec_this(void* p, size_t tp) {
    visit_ctx* c = (visit_ctx*)p;
    if (c->this_pos) { c->this_pos[c->n] = (uint32_t)tp; c->arg_pos[c->n] = 0xFFFFFFFFu; }
    c->n++;
}
static void rec_arg(void* p, size_t ap) {
    visit_ctx* c = (visit_ctx*)p;
    if (c->this_pos) { c->this_pos[c->n] = 0xFFFFFFFFu; c->arg_pos[c->n] = (uint32_t)ap; }
    c->n++;
}

/* which = 0: andv (SDR.hpp:958-997), one record per callback (this position, arg position);
 * which = 1: orv (SDR.hpp:1099-1166), the absent side of a single is 0xFFFFFFFF.
 * out_off[t_n*a_n + 1]; this_pos / arg_pos may be NULL (count only).  Returns the total. */
uint64_t osdr_batch_visit(int which, const uint64_t* t_off, const int32_t* t_ids, uint64_t t_n,
                          const uint64_t* a_off, const int32_t* a_ids, uint64_t a_n,
                          uint64_t* out_off, uint32_t* this_pos, uint32_t* arg_pos) {
    visit_ctx c = {this_pos, arg_pos, 0};
    for (uint64_t i = 0; i < t_n; ++i)
        for (uint64_t j = 0; j < a_n; ++j) {
            out_off[i * a_n + j] = c.n;
            const int32_t* t = t_ids + t_off[i];
            const int32_t* a = a_ids + a_off[j];
            const size_t tn = (size_t)(t_off[i + 1] - t_off[i]), an = (size_t)(a_off[j + 1] - a_off[j]);
            if (which == 0) osdr_andv(t, tn, a, an, rec_both, &c);
            else osdr_orv(t, tn, a, an, rec_this, rec_arg, rec_both, &c);
        }
    out_off[t_n * a_n] = c.n;
    return c.n;
}

void osdr_batch_size_op(int op,
                        const uint64_t* t_off, const int32_t* t_ids, const float* t_data,
                        int t_kind, uint64_t t_n,
                        const uint64_t* a_off, const int32_t* a_ids, const float* a_data,
                        int a_kind, uint64_t a_n,
                        uint32_t* out) {
    for (uint64_t i = 0; i < t_n; ++i) {
        osdr_view t = row_view(t_off, t_ids, t_data, t_kind, i);
        for (uint64_t j = 0; j < a_n; ++j) {
            osdr_view a = row_view(a_off, a_ids, a_data, a_kind, j);
            out[i * a_n + j] = (uint32_t)osdr_size_op(op, &t, &a);
        }
    }
}

uint64_t osdr_batch_elem_op(int op,
                            const uint64_t* t_off, const int32_t* t_ids, const float* t_data,
                            int t_kind, uint64_t t_n,
                            const uint64_t* a_off, const int32_t* a_ids, const float* a_data,
                            int a_kind, uint64_t a_n,
                            uint64_t* out_off, int32_t* out_ids, float* out_data) {
    uint64_t total = 0;
    size_t cap = 1;
    for (uint64_t i = 0; i < t_n; ++i)
        for (uint64_t j = 0; j < a_n; ++j) {
            size_t need = (size_t)((t_off[i + 1] - t_off[i]) + (a_off[j + 1] - a_off[j]));
            if (need > cap) cap = need;
        }
    int32_t* tmp_ids = (int32_t*)malloc(cap * sizeof(int32_t));
    float* tmp_data = (float*)malloc(cap * sizeof(float));
    for (uint64_t i = 0; i < t_n; ++i) {
        osdr_view t = row_view(t_off, t_ids, t_data, t_kind, i);
        for (uint64_t j = 0; j < a_n; ++j) {
            osdr_view a = row_view(a_off, a_ids, a_data, a_kind, j);
            out_off[i * a_n + j] = total;
            size_t n = osdr_elem_op(op, &t, &a, tmp_ids, tmp_data);
            if (out_ids) {
                memcpy(out_ids + total, tmp_ids, n * sizeof(int32_t));
                if (out_data && t_kind != OSDR_EMPTY) memcpy(out_data + total, tmp_data, n * sizeof(float));
            }
            total += n;
        }
    }
    out_off[t_n * a_n] = total;
    free(tmp_ids);
    free(tmp_data);
    return total;
}

double osdr_time_size_op(int op,
                         const uint64_t* t_off, const int32_t* t_ids, const float* t_data,
                         int t_kind, uint64_t t_n,
                         const uint64_t* a_off, const int32_t* a_ids, const float* a_data,
                         int a_kind, uint64_t a_n,
                         int repeats, uint64_t* checksum) {
    double best = 1e300;
    uint64_t sum = 0;
    for (int r = 0; r < repeats; ++r) {
        struct timespec t0, t1;
        sum = 0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (uint64_t i = 0; i < t_n; ++i) {
            osdr_view t = row_view(t_off, t_ids, t_data, t_kind, i);
            for (uint64_t j = 0; j < a_n; ++j) {
                osdr_view a = row_view(a_off, a_ids, a_data, a_kind, j);
                sum += osdr_size_op(op, &t, &a);
            }
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (s < best) best = s;
    }
    if (checksum) *checksum = sum;
    return best;
}
