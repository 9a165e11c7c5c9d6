/* cs_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See cs_oracle.h for scope, users and the parity pin.  Every function cites
 * the reference lines whose behaviour it restates. */
#include "cs_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DMIN INT32_MIN
#define DMAX INT32_MAX

/* ---- scalar layer (own formulation, independent of cs_arith.h) ------------ */

/* arith.c:27-35 */
int32_t cso_neg(int32_t a) {
  if (a == DMIN) return DMAX;
  if (a == DMAX) return DMIN;
  return -a;
}

/* arith.c:38-51: -inf absorbs first, then +inf, then signed overflow saturates */
int32_t cso_add(int32_t a, int32_t b) {
  if (a == DMIN || b == DMIN) return DMIN;
  if (a == DMAX || b == DMAX) return DMAX;
  int32_t s;
  if (__builtin_add_overflow(a, b, &s)) return a < 0 ? DMIN : DMAX;
  return s;
}

/* arith.c:54-75 */
int32_t cso_mul(int32_t a, int32_t b) {
  if (a == DMIN) return b < 0 ? DMAX : DMIN;
  if (b == DMIN) return a < 0 ? DMAX : DMIN;
  if (a == DMAX) return b < 0 ? DMIN : DMAX;
  if (b == DMAX) return a < 0 ? DMIN : DMAX;
  int32_t p;
  if (__builtin_mul_overflow(a, b, &p)) return ((a < 0) != (b < 0)) ? DMIN : DMAX;
  return p;
}

int32_t cso_min(int32_t a, int32_t b) { return a < b ? a : b; } /* arith.c:78-80 */
int32_t cso_max(int32_t a, int32_t b) { return a > b ? a : b; } /* arith.c:83-85 */

static cs_val iv(int32_t lo, int32_t hi) { cs_val v; v.lo = lo; v.hi = hi; return v; }
static int v_is_value(cs_val v) { return v.lo == v.hi; }            /* csolve.h:57-59 */
static int v_is_true(cs_val v) { return v.lo > 0 || v.hi < 0; }      /* csolve.h:61-63 */
static int v_is_false(cs_val v) { return v.lo == v.hi && v.lo == 0; } /* csolve.h:65-67 */

/* ---- instance ------------------------------------------------------------- */

typedef struct {
  int32_t var;
  cs_val old;
} trail_ent;

typedef struct {
  int32_t var;
  cs_val val;
} log_ent;

struct cso {
  const cs_model *m;
  cs_val *dom;       /* [n_vars] */
  cs_val *cval;      /* [n_nodes] private values of CS_OP_CONST terminals */
  uint64_t *ctag;    /* [n_clauses] prop_tag of each clause (csolve.h:95) */
  uint64_t tag;      /* _prop_tag (propagate.c:490) */
  int root_phase, record_only;
  uint64_t props;
  trail_ent *trail;
  size_t trail_n, trail_cap;
  log_ent *log;
  size_t log_n, log_cap;
  /* search-driver state */
  int64_t *prio;     /* env_t.prio */
  int32_t *heap;     /* _var_order */
  int32_t *order;    /* env_t.order, -1 = not in heap */
  int32_t heap_n;
  int prefer_failing, order_kind;
  /* test hook: the variables whose priority a failing call bumped, in order (propagate.c:33-54) */
  int32_t bump_log[4096];
  int32_t bump_n;
};

static void note_bump(cso *o, int32_t var) {
  if (o->bump_n < 4096) o->bump_log[o->bump_n] = var;
  o->bump_n++;
}
int32_t cso_test_bumps(const cso *o, int32_t *out, int32_t cap) {
  const int32_t k = o->bump_n < 4096 ? o->bump_n : 4096;
  for (int32_t i = 0; i < k && i < cap; i++) out[i] = o->bump_log[i];
  return o->bump_n;
}

cso *cso_new(const cs_model *m) {
  cso *o = (cso *)calloc(1, sizeof *o);
  o->m = m;
  size_t nv = (size_t)(m->n_vars ? m->n_vars : 1);
  o->dom = (cs_val *)malloc(nv * sizeof(cs_val));
  memcpy(o->dom, m->dom, (size_t)m->n_vars * sizeof(cs_val));
  o->cval = (cs_val *)calloc((size_t)(m->n_nodes ? m->n_nodes : 1), sizeof(cs_val));
  for (int32_t i = 0; i < m->n_nodes; i++)
    if (m->nodes[i].op == CS_OP_CONST) o->cval[i] = iv(m->nodes[i].a, m->nodes[i].b);
  o->ctag = (uint64_t *)calloc((size_t)(m->n_clauses > 0 ? m->n_clauses : 1), sizeof(uint64_t));
  o->prio = (int64_t *)malloc(nv * sizeof(int64_t));
  for (int32_t v = 0; v < m->n_vars; v++) o->prio[v] = m->prio[v];
  o->heap = (int32_t *)malloc(nv * sizeof(int32_t));
  o->order = (int32_t *)malloc(nv * sizeof(int32_t));
  for (int32_t v = 0; v < m->n_vars; v++) o->order[v] = -1;
  o->prefer_failing = 1;
  return o;
}

void cso_free(cso *o) {
  if (o == NULL) return;
  free(o->dom); free(o->cval); free(o->ctag); free(o->trail); free(o->log);
  free(o->prio); free(o->heap); free(o->order);
  free(o);
}

void cso_set_root_phase(cso *o, int on) { o->root_phase = on; }
void cso_set_record_only(cso *o, int on) { o->record_only = on; }
cs_val *cso_domains(cso *o) { return o->dom; }
uint64_t cso_props(const cso *o) { return o->props; }
void cso_reset_stats(cso *o) { o->props = 0; }

size_t cso_bind_depth(const cso *o) { return o->trail_n; }
size_t cso_log_len(const cso *o) { return o->log_n; }
void cso_log_get(const cso *o, size_t i, int32_t *var, cs_val *val) { *var = o->log[i].var; *val = o->log[i].val; }
void cso_log_clear(cso *o) { o->log_n = 0; }

/* util.c:137-162 */
void cso_bind(cso *o, int32_t var, cs_val val, int32_t clause) {
  (void)clause;
  if (o->record_only) {
    if (o->log_n == o->log_cap) {
      o->log_cap = o->log_cap ? o->log_cap * 2 : 16;
      o->log = (log_ent *)realloc(o->log, o->log_cap * sizeof(log_ent));
    }
    o->log[o->log_n].var = var;
    o->log[o->log_n].val = val;
    o->log_n++;
    return;
  }
  if (o->trail_n == o->trail_cap) {
    o->trail_cap = o->trail_cap ? o->trail_cap * 2 : 1024;
    o->trail = (trail_ent *)realloc(o->trail, o->trail_cap * sizeof(trail_ent));
  }
  o->trail[o->trail_n].var = var;
  o->trail[o->trail_n].old = o->dom[var];
  o->trail_n++;
  o->dom[var] = val;
}

/* util.c:165-173 */
void cso_unbind(cso *o, size_t depth) {
  while (o->trail_n > depth) {
    o->trail_n--;
    o->dom[o->trail[o->trail_n].var] = o->trail[o->trail_n].old;
  }
}

/* ---- variable-order heap (strategy.c:79-246) ------------------------------ */

static int var_cmp(const cso *o, int32_t e1, int32_t e2) {
  cs_val v1 = o->dom[e1], v2 = o->dom[e2];
  int cmp = 0;
  switch (o->order_kind) {
  case 1: /* smallest domain, strategy.c:85-91 */
    cmp = cso_add(cso_add(v2.hi, cso_neg(v2.lo)), cso_add(v1.lo, cso_neg(v1.hi)));
    break;
  case 2: /* largest domain, 92-98 */
    cmp = cso_add(cso_add(v1.hi, cso_neg(v1.lo)), cso_add(v2.lo, cso_neg(v2.hi)));
    break;
  case 3: cmp = cso_add(v2.lo, cso_neg(v1.lo)); break; /* smallest value, 99-102 */
  case 4: cmp = cso_add(v1.hi, cso_neg(v2.hi)); break; /* largest value, 103-106 */
  default: cmp = 0; break;
  }
  if (o->prefer_failing && cmp == 0) cmp = (int)(o->prio[e1] - o->prio[e2]); /* 116-118 */
  return cmp;
}

static void heap_swap(cso *o, int32_t p, int32_t q) {
  int32_t t = o->heap[p];
  o->heap[p] = o->heap[q]; o->order[o->heap[p]] = p;
  o->heap[q] = t; o->order[t] = q;
}

static void heap_up(cso *o, int32_t pos) { /* strategy.c:173-179 */
  while (pos > 0 && var_cmp(o, o->heap[(pos - 1) / 2], o->heap[pos]) < 0) {
    heap_swap(o, pos, (pos - 1) / 2);
    pos = (pos - 1) / 2;
  }
}

static void heap_down(cso *o, int32_t pos) { /* strategy.c:182-206 */
  for (;;) {
    int32_t l = 2 * pos + 1, r = 2 * pos + 2, best = pos;
    if (l < o->heap_n && var_cmp(o, o->heap[l], o->heap[best]) > 0) best = l;
    if (r < o->heap_n && var_cmp(o, o->heap[r], o->heap[best]) > 0) best = r;
    if (best == pos) break;
    heap_swap(o, best, pos);
    pos = best;
  }
}

static void heap_push(cso *o, int32_t v) { /* strategy.c:209-216 */
  int32_t pos = o->heap_n++;
  o->heap[pos] = v;
  o->order[v] = pos;
  heap_up(o, pos);
}

static int32_t heap_pop(cso *o) { /* strategy.c:219-232 */
  int32_t v = o->heap[0];
  o->order[v] = -1;
  o->heap_n--;
  if (o->heap_n > 0) {
    o->heap[0] = o->heap[o->heap_n];
    o->order[o->heap[0]] = 0;
    heap_down(o, 0);
  }
  return v;
}

static void heap_update(cso *o, int32_t v) { /* strategy.c:240-246 */
  if (o->order[v] >= 0) {
    heap_up(o, o->order[v]);
    heap_down(o, o->order[v]);
  }
}

/* test hooks: the reference's own unit vectors of strategy.c (tests/golden/ref_unit_strategy.json) are run
 * against the comparison and the heap above.  op: 0 up(a), 1 down(a), 2 swap(a, b), 3 push(var a), 4 pop -> var,
 * 5 update(var a) */
void cso_test_set_strategy(cso *o, int order_kind, int prefer_failing) {
  o->order_kind = order_kind;
  o->prefer_failing = prefer_failing;
}
void cso_test_set_prio(cso *o, int32_t var, int64_t prio) { o->prio[var] = prio; }
int cso_test_var_cmp(const cso *o, int32_t v1, int32_t v2) { return var_cmp(o, v1, v2); }
void cso_test_heap_load(cso *o, const int32_t *vars, int32_t n) {
  for (int32_t v = 0; v < o->m->n_vars; v++) o->order[v] = -1;
  o->heap_n = n;
  for (int32_t i = 0; i < n; i++) {
    o->heap[i] = vars[i];
    o->order[vars[i]] = i;
  }
}
int32_t cso_test_heap_op(cso *o, int op, int32_t a, int32_t b) {
  switch (op) {
  case 0: heap_up(o, a); return 0;
  case 1: heap_down(o, a); return 0;
  case 2: heap_swap(o, a, b); return 0;
  case 3: heap_push(o, a); return 0;
  case 4: return heap_pop(o);
  case 5: heap_update(o, a); return 0;
  default: return -1;
  }
}
int32_t cso_test_heap_get(const cso *o, int32_t *out) {
  for (int32_t i = 0; i < o->heap_n; i++) out[i] = o->heap[i];
  return o->heap_n;
}
int32_t cso_test_heap_pos(const cso *o, int32_t var) { return o->order[var]; }

/* ---- evaluation (eval.c) --------------------------------------------------- */

static cs_val ev(cso *o, int32_t node);

static cs_val tv_unknown(void) { return iv(0, 1); }

static int any_unbounded(cs_val a, cs_val b) {
  return a.lo == DMIN || a.hi == DMAX || b.lo == DMIN || b.hi == DMAX;
}

static cs_val ev(cso *o, int32_t node) {
  const cs_node *n = &o->m->nodes[node];
  switch (n->op) {
  case CS_OP_VAR: return o->dom[n->a];   /* eval.c:27-29 */
  case CS_OP_CONST: return o->cval[node];
  case CS_OP_EQ: { /* eval.c:32-63 */
    cs_val a = ev(o, n->a), b = ev(o, n->b);
    if (any_unbounded(a, b)) return tv_unknown();
    if (a.hi == b.hi && a.lo == b.lo && a.hi == a.lo) return iv(1, 1);
    if (a.hi < b.lo || a.lo > b.hi) return iv(0, 0);
    return tv_unknown();
  }
  case CS_OP_LT: { /* eval.c:66-97 */
    cs_val a = ev(o, n->a), b = ev(o, n->b);
    if (any_unbounded(a, b)) return tv_unknown();
    if (a.hi < b.lo) return iv(1, 1);
    if (a.lo >= b.hi) return iv(0, 0);
    return tv_unknown();
  }
  case CS_OP_NEG: { /* eval.c:100-114 */
    cs_val a = ev(o, n->a);
    return iv(cso_neg(a.hi), cso_neg(a.lo));
  }
  case CS_OP_ADD: { /* eval.c:117-135 */
    cs_val a = ev(o, n->a), b = ev(o, n->b);
    return iv(cso_add(a.lo, b.lo), cso_add(a.hi, b.hi));
  }
  case CS_OP_MUL: { /* eval.c:138-160 */
    cs_val a = ev(o, n->a), b = ev(o, n->b);
    int32_t c[4] = { cso_mul(a.lo, b.lo), cso_mul(a.lo, b.hi), cso_mul(a.hi, b.lo), cso_mul(a.hi, b.hi) };
    int32_t lo = c[0], hi = c[0];
    for (int i = 1; i < 4; i++) { lo = cso_min(lo, c[i]); hi = cso_max(hi, c[i]); }
    return iv(lo, hi);
  }
  case CS_OP_NOT: { /* eval.c:163-180 */
    cs_val a = ev(o, n->a);
    if (v_is_true(a)) return iv(0, 0);
    if (v_is_false(a)) return iv(1, 1);
    return tv_unknown();
  }
  case CS_OP_AND: { /* eval.c:183-205 */
    cs_val l = ev(o, n->a);
    if (v_is_false(l)) return iv(0, 0);
    cs_val r = ev(o, n->b);
    if (v_is_false(r)) return iv(0, 0);
    if (v_is_true(l) && v_is_true(r)) return iv(1, 1);
    return tv_unknown();
  }
  case CS_OP_OR: { /* eval.c:208-230 */
    cs_val l = ev(o, n->a);
    if (v_is_true(l)) return iv(1, 1);
    cs_val r = ev(o, n->b);
    if (v_is_true(r)) return iv(1, 1);
    if (v_is_false(l) && v_is_false(r)) return iv(0, 0);
    return tv_unknown();
  }
  case CS_OP_WAND: { /* eval.c:233-255 */
    int all_true = 1;
    for (int32_t i = 0; i < n->b; i++) {
      cs_val v = ev(o, o->m->kids[n->a + i]);
      if (v_is_false(v)) return iv(0, 0);
      if (!v_is_true(v)) all_true = 0;
    }
    return all_true ? iv(1, 1) : tv_unknown();
  }
  case CS_OP_CONFL: /* eval.c:258-277: elements in order; the first one that is not a value decides "unknown", the
                     * first value different from its conflict value decides "true" */
    for (int32_t i = 0; i < n->b; i++) {
      cs_val v = ev(o, o->m->kids[n->a + 2 * i]);
      if (!v_is_value(v)) return tv_unknown();
      if (v.lo != o->m->kids[n->a + 2 * i + 1]) return iv(1, 1);
    }
    return tv_unknown();
  default:
    return tv_unknown();
  }
}

cs_val cso_eval(cso *o, int32_t node) { return ev(o, node); }

/* ---- propagation (propagate.c) --------------------------------------------- */

static int32_t prop(cso *o, int32_t node, cs_val val, int32_t clause);

#define TRY(x)                                                                 \
  do {                                                                         \
    if ((x) == CSO_ERROR) return CSO_ERROR;                                    \
  } while (0)

/* propagate.c:57-87 with 33-54 */
static int32_t prop_var(cso *o, int32_t var, cs_val val, int32_t clause) {
  cs_val t = o->dom[var];
  int has_env = !o->root_phase;
  if (t.lo > val.hi || t.hi < val.lo) {
    if (has_env) {
      o->prio[var]++;
      note_bump(o, var);
      heap_update(o, var);
    }
    return CSO_ERROR;
  }
  int32_t lo = cso_max(t.lo, val.lo), hi = cso_min(t.hi, val.hi);
  if (lo == t.lo && hi == t.hi) return 0;
  if (!has_env) {
    o->dom[var] = iv(lo, hi);
    return 1;
  }
  cso_bind(o, var, iv(lo, hi), clause);
  o->props++;
  if (o->record_only) return 1; /* mocked bind: the variable has no clauses */
  int32_t p = cso_propagate_clauses(o, var);
  if (p == CSO_ERROR) {
    o->prio[var]++;
    note_bump(o, var);
    heap_update(o, var);
    return CSO_ERROR;
  }
  return p + 1;
}

/* terminal without environment: propagate.c:57-87 with var == NULL */
static int32_t prop_const(cso *o, int32_t node, cs_val val) {
  cs_val t = o->cval[node];
  if (t.lo > val.hi || t.hi < val.lo) return CSO_ERROR;
  int32_t lo = cso_max(t.lo, val.lo), hi = cso_min(t.hi, val.hi);
  if (lo == t.lo && hi == t.hi) return 0;
  o->cval[node] = iv(lo, hi);
  return 1;
}

/* propagate.c:106-120 */
static int32_t eq_false_side(cso *o, int32_t p, cs_val pval, cs_val other, int32_t clause) {
  if (v_is_value(other) && other.lo != DMIN && other.lo != DMAX) {
    if (other.lo == pval.lo) return prop(o, p, iv(other.lo + 1, DMAX), clause);
    if (other.lo == pval.hi) return prop(o, p, iv(DMIN, other.lo - 1), clause);
  }
  return 0;
}

/* propagate.c:223-230 */
static int32_t add_side(cso *o, int32_t p, int32_t c, cs_val val, int32_t clause) {
  cs_val cv = ev(o, c);
  return prop(o, p, iv(cso_add(val.lo, cso_neg(cv.hi)), cso_add(val.hi, cso_neg(cv.lo))), clause);
}

/* propagate.c:249-270 */
static int32_t mul_side(cso *o, int32_t p, int32_t c, cs_val val, int32_t clause) {
  if (val.lo != DMIN && val.hi != DMIN) {
    cs_val cv = ev(o, c);
    if (v_is_value(cv)) {
      if (((val.lo > 0 || val.hi < 0) && cv.lo == 0) ||
          (v_is_value(val) && cv.lo != 0 && (val.lo % cv.lo) != 0))
        return CSO_ERROR;
      if (cv.lo != 0) {
        int32_t lo = val.lo / cv.lo, hi = val.hi / cv.lo;
        return prop(o, p, iv(cso_min(lo, hi), cso_max(lo, hi)), clause);
      }
    }
  }
  return 0;
}

/* propagate.c:305-316 */
static int32_t logic_both(cso *o, int32_t l, int32_t r, cs_val val, int32_t clause) {
  int32_t p = prop(o, r, val, clause);
  TRY(p);
  int32_t q = prop(o, l, val, clause);
  TRY(q);
  return p + q;
}

/* propagate.c:320-340; neutral_true selects is_true / is_false as the neutral test */
static int32_t logic_either(cso *o, int32_t l, int32_t r, cs_val val, int neutral_true, int32_t clause) {
  int32_t p = 0, q = 0;
  cs_val lv = ev(o, l);
  if (neutral_true ? v_is_true(lv) : v_is_false(lv)) {
    p = prop(o, r, val, clause);
    TRY(p);
  }
  cs_val rv = ev(o, r);
  if (neutral_true ? v_is_true(rv) : v_is_false(rv)) {
    q = prop(o, l, val, clause);
    TRY(q);
  }
  return p + q;
}

static int32_t prop(cso *o, int32_t node, cs_val val, int32_t clause) {
  const cs_node *n = &o->m->nodes[node];
  const int32_t l = n->a, r = n->b;
  switch (n->op) {
  case CS_OP_VAR:
    return prop_var(o, n->a, val, clause);
  case CS_OP_CONST:
    return prop_const(o, node, val);
  case CS_OP_EQ: /* propagate.c:139-152 */
    if (v_is_true(val)) { /* 90-103 */
      int32_t p = prop(o, r, ev(o, l), clause);
      TRY(p);
      int32_t q = prop(o, l, ev(o, r), clause);
      TRY(q);
      return p + q;
    }
    if (v_is_false(val)) { /* 123-136 */
      cs_val lv = ev(o, l), rv = ev(o, r);
      int32_t p = eq_false_side(o, r, rv, lv, clause);
      TRY(p);
      int32_t q = eq_false_side(o, l, lv, rv, clause);
      TRY(q);
      return p + q;
    }
    return 0;
  case CS_OP_LT: /* propagate.c:195-208 */
    if (v_is_true(val)) { /* 155-176 */
      int32_t p = 0, q = 0;
      cs_val lv = ev(o, l);
      if (lv.lo != DMIN && lv.lo != DMAX) {
        p = prop(o, r, iv(lv.lo + 1, DMAX), clause);
        TRY(p);
      }
      cs_val rv = ev(o, r);
      if (rv.hi != DMIN && rv.hi != DMAX) {
        q = prop(o, l, iv(DMIN, rv.hi - 1), clause);
        TRY(q);
      }
      return p + q;
    }
    if (v_is_false(val)) { /* 179-192 */
      cs_val lv = ev(o, l);
      int32_t p = prop(o, r, iv(DMIN, lv.hi), clause);
      TRY(p);
      cs_val rv = ev(o, r);
      int32_t q = prop(o, l, iv(rv.lo, DMAX), clause);
      TRY(q);
      return p + q;
    }
    return 0;
  case CS_OP_NEG: /* propagate.c:211-220 */
    return prop(o, l, iv(cso_neg(val.hi), cso_neg(val.lo)), clause);
  case CS_OP_ADD: { /* propagate.c:233-246 */
    int32_t p = add_side(o, r, l, val, clause);
    TRY(p);
    int32_t q = add_side(o, l, r, val, clause);
    TRY(q);
    return p + q;
  }
  case CS_OP_MUL: { /* propagate.c:273-286 */
    int32_t p = mul_side(o, r, l, val, clause);
    TRY(p);
    int32_t q = mul_side(o, l, r, val, clause);
    TRY(q);
    return p + q;
  }
  case CS_OP_NOT: /* propagate.c:289-301 */
    if (v_is_true(val)) return prop(o, l, iv(0, 0), clause);
    if (v_is_false(val)) return prop(o, l, iv(1, 1), clause);
    return 0;
  case CS_OP_AND: /* propagate.c:343-358 */
    if (v_is_true(val)) return logic_both(o, l, r, val, clause);
    if (v_is_false(val)) return logic_either(o, l, r, val, 1, clause);
    return 0;
  case CS_OP_OR: /* propagate.c:361-376 */
    if (v_is_false(val)) return logic_both(o, l, r, val, clause);
    if (v_is_true(val)) return logic_either(o, l, r, val, 0, clause);
    return 0;
  case CS_OP_WAND: { /* propagate.c:379-392 */
    int32_t sum = 0;
    if (v_is_true(val)) {
      for (int32_t i = 0; i < n->b; i++) {
        int32_t p = prop(o, o->m->kids[n->a + i], val, clause);
        TRY(p);
        sum += p;
        n = &o->m->nodes[node];
      }
    }
    return sum;
  }
  case CS_OP_CONFL: { /* propagate.c:395-471 */
    if (!v_is_true(val)) return 0;
    /* propagate_confl_find (405-440): the single element that is not a value, provided every other element has
     * its conflict value.  (The reference also moves the element(s) that stopped the scan to the front of the
     * array; that shortens its next scan and changes no result of propagate_confl.  eval_confl of a reordered
     * clause can answer [0,1] where the original order answers 1 -- a value only the in-search normalise tail
     * consumes, which is out of scope here.) */
    int32_t p = -1;
    for (int32_t i = 0; i < n->b; i++) {
      cs_val v = ev(o, o->m->kids[n->a + 2 * i]);
      if (v_is_value(v)) {
        if (v.lo != o->m->kids[n->a + 2 * i + 1]) return 0;
      } else if (p < 0) {
        p = i;
      } else {
        return 0;
      }
    }
    if (p < 0) return 0;
    /* propagate_confl_infer (443-457) */
    const int32_t term = o->m->kids[n->a + 2 * p], cv = o->m->kids[n->a + 2 * p + 1];
    cs_val v = ev(o, term);
    if (v.lo == cv && v.lo != CS_DOM_MIN && v.lo != CS_DOM_MAX) return prop(o, term, iv(v.lo + 1, CS_DOM_MAX), clause);
    if (v.hi == cv && v.hi != CS_DOM_MIN && v.hi != CS_DOM_MAX) return prop(o, term, iv(CS_DOM_MIN, v.hi - 1), clause);
    return 0;
  }
  default:
    return 0;
  }
}

int32_t cso_propagate_node(cso *o, int32_t node, cs_val val, int32_t clause) {
  return prop(o, node, val, clause);
}

/* propagate.c:474-485 */
int32_t cso_propagate(cso *o, int32_t node, size_t limit) {
  int32_t total = 0, p;
  size_t i = 0;
  do {
    p = prop(o, node, iv(1, 1), -1);
    TRY(p);
    total += p;
  } while (p != 0 && i++ < limit);
  return total;
}

/* propagate.c:488-538.  The normalise-and-patch tail (521-535) rewrites clauses
 * without changing what they compute and is not restated. */
int32_t cso_propagate_clauses(cso *o, int32_t var) {
  const cs_model *m = o->m;
  uint64_t tag = ++o->tag;
  int32_t total = 0;
  for (int32_t i = m->list_off[var], e = m->list_off[var + 1]; i < e; i++) {
    int32_t c = m->list[i];
    if (o->ctag[c] > tag) continue;
    o->ctag[c] = tag;
    int32_t p = prop(o, m->clause_node[c], iv(1, 1), c);
    TRY(p);
    total += p;
  }
  return total;
}

int64_t cso_instance(cso *o, const cs_val *dom_in, int32_t var, cs_val val, cs_val *dom_out) {
  const cs_model *m = o->m;
  memcpy(o->dom, dom_in, (size_t)m->n_vars * sizeof(cs_val));
  o->trail_n = 0;
  o->props = 0;
  o->bump_n = 0;
  int32_t r;
  if (var >= 0) {
    o->root_phase = 0;
    o->dom[var] = val;
    r = cso_propagate_clauses(o, var);
  } else {
    o->root_phase = 1;
    r = cso_propagate(o, m->root, (size_t)-2);
    o->root_phase = 0;
  }
  if (dom_out != NULL) memcpy(dom_out, o->dom, (size_t)m->n_vars * sizeof(cs_val));
  o->trail_n = 0;
  if (r == CSO_ERROR) return CSO_ERROR;
  return var >= 0 ? (int64_t)o->props : (int64_t)r;
}

uint64_t cso_instances(cso *o, const cs_val *states_in, const int32_t *nodes, int64_t count, cs_val *states_out,
                       int64_t *status) {
  const size_t n = (size_t)o->m->n_vars;
  uint64_t binds = 0;
  for (int64_t i = 0; i < count; i++) {
    const int32_t *nd = &nodes[4 * i];
    cs_val v;
    v.lo = nd[1];
    v.hi = nd[2];
    int64_t st = cso_instance(o, states_in + (size_t)nd[3] * n, nd[0], v, states_out ? states_out + (size_t)i * n : NULL);
    if (status) status[i] = st;
    binds += o->props;
  }
  return binds;
}

/* ---- search driver (csolve.c) ----------------------------------------------- */

typedef struct {
  size_t bind_depth;
  int32_t var;
  int active;
  uint32_t iter, seed;
  cs_val bounds;
} step;

typedef struct {
  cso *o;
  const cso_options *opt;
  cso_result *res;
  step *steps;
  uint32_t fail_count;                 /* csolve.c:44-49 */
  uint64_t fail_threshold, fail_counter;
  int32_t best;                        /* *_objective_best */
} search;

void cso_default_options(cso_options *opt) {
  opt->prefer_failing = 1;        /* csolve.h:405 */
  opt->restart_frequency = 100;   /* csolve.h:419 */
  opt->order = 0;                 /* csolve.h:426 */
  opt->max_calls = 0;
  opt->max_solutions = 0;
}

static int objective_better(const search *s) { /* objective.c:62-78 */
  const cs_model *m = s->o->m;
  if (m->objective == CS_OBJ_MIN) return s->o->dom[m->obj_var].lo < s->best;
  if (m->objective == CS_OBJ_MAX) return s->o->dom[m->obj_var].hi > s->best;
  return 1;
}

static void objective_update_val(search *s) { /* objective.c:101-126: untrailed */
  const cs_model *m = s->o->m;
  if (m->objective == CS_OBJ_MIN) {
    int32_t b = cso_add(s->best, cso_neg(1));
    if (s->o->dom[m->obj_var].hi > b) s->o->dom[m->obj_var].hi = b;
  } else if (m->objective == CS_OBJ_MAX) {
    int32_t b = cso_add(s->best, 1);
    if (s->o->dom[m->obj_var].lo < b) s->o->dom[m->obj_var].lo = b;
  }
}

static int restartable(const search *s) { /* csolve.c:212-214 */
  return s->o->m->objective == CS_OBJ_ANY && s->opt->restart_frequency > 0;
}

static void step_leave(search *s, step *st) { cso_unbind(s->o, st->bind_depth); } /* csolve.c:307-314 */

static void step_deactivate(search *s, step *st) { /* csolve.c:288-291 */
  heap_push(s->o, st->var);
  st->active = 0;
}

static void unwind(search *s, size_t level, size_t stop) { /* csolve.c:341-347 */
  for (size_t i = level; i != stop - 1; --i) {
    step_leave(s, &s->steps[i]);
    step_deactivate(s, &s->steps[i]);
  }
}

static int update_solution(search *s) { /* csolve.c:222-244 */
  cso *o = s->o;
  const cs_model *m = o->m;
  if (!v_is_true(ev(o, m->root))) return 0;
  int found_any = m->objective == CS_OBJ_ANY && s->res->solutions > 0;
  if (found_any || !objective_better(s)) return 0;
  if (m->objective == CS_OBJ_MIN) s->best = o->dom[m->obj_var].lo; /* objective.c:81-98 */
  if (m->objective == CS_OBJ_MAX) s->best = o->dom[m->obj_var].hi;
  if (s->res->solutions_stored < s->opt->max_solutions) {
    int32_t *dst = &s->res->solution_values[s->res->solutions_stored * (size_t)m->n_vars];
    for (int32_t v = 0; v < m->n_vars; v++) dst[v] = o->dom[v].lo;
    s->res->solutions_stored++;
  }
  s->res->solutions++;
  return 1;
}

int cso_solve(cso *o, const cso_options *opt, cso_result *res) {
  const cs_model *m = o->m;
  if (m->list_off == NULL) return -1;
  size_t size = (size_t)m->n_vars;
  memset(res, 0, sizeof *res);
  if (opt->max_solutions)
    res->solution_values = (int32_t *)malloc((size_t)opt->max_solutions * (size ? size : 1) * sizeof(int32_t));

  search s;
  memset(&s, 0, sizeof s);
  s.o = o; s.opt = opt; s.res = res;
  s.steps = (step *)calloc(size ? size : 1, sizeof(step));
  s.fail_threshold = 1; s.fail_counter = 1;
  s.best = m->objective == CS_OBJ_MIN ? DMAX : (m->objective == CS_OBJ_MAX ? DMIN : 0); /* objective.c:34-53 */

  o->root_phase = 0;
  o->record_only = 0;
  o->props = 0;
  o->trail_n = 0;
  o->prefer_failing = opt->prefer_failing;
  o->order_kind = opt->order;
  o->heap_n = 0;
  for (int32_t v = 0; v < m->n_vars; v++) heap_push(o, v); /* strategy.c:154-162 */
  srand(1); /* the reference never seeds rand(): glibc's default sequence */

  size_t level = 0;
  for (;;) {
    if (opt->max_calls && res->calls >= opt->max_calls) { res->stopped_early = 1; break; }
    if (m->objective == CS_OBJ_ANY && res->solutions > 0) break; /* csolve.c:413-416 */

    if (level == size) { /* csolve.c:418-427 */
      int updated = update_solution(&s);
      if (updated && m->objective != CS_OBJ_ALL) {
        level--;
        unwind(&s, level, 0);
        level = 0;
        continue;
      }
      if (level != 0) { level--; continue; }
      break;
    }

    step *st = &s.steps[level];
    if (!st->active) { /* csolve.c:429-439, 279-286 */
      st->var = heap_pop(o);
      st->active = 1;
      st->bounds = o->dom[st->var];
      st->iter = 0;
      st->seed = restartable(&s) ? (uint32_t)rand() : 0;
    } else {
      step_leave(&s, st);
      st->iter++;
    }

    if (!(st->iter <= (uint32_t)(st->bounds.hi - st->bounds.lo))) { /* csolve.c:323-328, 441-445 */
      step_deactivate(&s, st);
      if (level != 0) { level--; continue; }
      break;
    }

    /* csolve.c:331-338 value order, 294-304 step_enter */
    int32_t val = ((st->iter ^ st->seed) & 1u) ? st->bounds.hi - (int32_t)(st->iter >> 1)
                                               : st->bounds.lo + (int32_t)(st->iter >> 1);
    st->bind_depth = o->trail_n;
    if (!v_is_value(o->dom[st->var])) cso_bind(o, st->var, iv(val, val), -1);

    objective_update_val(&s);
    res->calls++;

    /* csolve.c:247-261 */
    int failed = cso_propagate_clauses(o, st->var) == CSO_ERROR ||
                 (m->obj_var >= 0 && cso_propagate_clauses(o, m->obj_var) == CSO_ERROR);
    if (failed) res->cuts++;

    if (!failed) { /* csolve.c:457-468 */
      o->prio[st->var]--;
      level++;
    } else {
      o->prio[st->var]++;
      if (restartable(&s)) { /* csolve.c:264-276 */
        s.fail_count++;
        if (s.fail_count > s.fail_threshold * opt->restart_frequency) {
          s.fail_count = 0;
          /* Luby sequence, csolve.c:76-83 */
          if ((s.fail_counter & -s.fail_counter) == s.fail_threshold) {
            s.fail_counter++;
            s.fail_threshold = 1;
          } else {
            s.fail_threshold <<= 1;
          }
          res->restarts++;
          unwind(&s, level, 0);
          level = 0;
          continue;
        }
      }
    }
  }
  res->props = o->props;
  res->best = s.best;
  free(s.steps);
  return 0;
}

void cso_result_free(cso_result *res) {
  free(res->solution_values);
  res->solution_values = NULL;
}
