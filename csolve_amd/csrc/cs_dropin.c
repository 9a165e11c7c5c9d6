/* cs_dropin.c -- the csolve.h-named entry points (include/csolve_dropin.h) as thin shims
 * over the C ABI of libcsolve_hip.so.  Host marshalling only: the driver's pointer trees are
 * flattened into a cs_model, every interval computation happens on the GPU.
 *
 * What the shim has to honour of the reference's contract:
 *   - PROP_ERROR (-1) on inconsistency, otherwise a count whose only use is ==/!= 0;
 *   - every narrowed variable is recorded through the DRIVER's bind(), so that the driver's
 *     unbind(step->bind_depth) restores the parent state (reference src/csolve.c:309);
 *   - the driver's `props` statistic is advanced by the number of narrowings;
 *   - conflict_reset() at the entry of propagate_clauses (reference src/propagate.c:496).
 * Not reproduced: prio++/strategy_var_order_update of the failing variable (heuristic only;
 * the deterministic mode -f false is unaffected), conflict-clause creation (run the driver
 * with -c false), the in-search normalise/patch tail.
 */
#include "../../include/csolve_dropin.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cs_internal.h"

/* ---- provided by the reference driver ------------------------------------------------ */
extern void bind(struct env_t *var, struct val_t val, const struct wand_expr_t *clause);
extern void conflict_reset(void);
extern void print_fatal(const char *fmt, ...);
extern uint64_t props;

#define PROP_ERROR (-1)

/* enum operator_t of the reference (csolve.h:133-162) */
static int op_of(const struct constr_t *c) {
  switch (c->type->op) {
  case ' ': return -1; /* terminal */
  case '=': return CS_OP_EQ;
  case '<': return CS_OP_LT;
  case '-': return CS_OP_NEG;
  case '+': return CS_OP_ADD;
  case '*': return CS_OP_MUL;
  case '!': return CS_OP_NOT;
  case '&': return CS_OP_AND;
  case '|': return CS_OP_OR;
  case 'A': return CS_OP_WAND;
  default:
    print_fatal("conflict clauses are not supported by the GPU propagator (run with -c false): %02x", c->type->op);
    return -1;
  }
}

/* ---- pointer -> int map ------------------------------------------------------------------ */

typedef struct {
  const void **key;
  int32_t *val;
  size_t cap, n;
} pmap;

static void pmap_init(pmap *p, size_t cap) {
  p->cap = cap;
  p->n = 0;
  p->key = (const void **)calloc(cap, sizeof *p->key);
  p->val = (int32_t *)malloc(cap * sizeof *p->val);
}

static void pmap_free(pmap *p) {
  free(p->key);
  free(p->val);
  p->key = NULL;
  p->val = NULL;
  p->cap = p->n = 0;
}

static size_t pmap_slot(const pmap *p, const void *k) {
  size_t i = (size_t)((((uintptr_t)k >> 3) * 11400714819323198485ull) % p->cap);
  while (p->key[i] != NULL && p->key[i] != k) i = (i + 1) % p->cap;
  return i;
}

static int32_t pmap_get(const pmap *p, const void *k) {
  size_t i = pmap_slot(p, k);
  return p->key[i] == k ? p->val[i] : -1;
}

static void pmap_put(pmap *p, const void *k, int32_t v) {
  if ((p->n + 1) * 2 > p->cap) {
    pmap q;
    pmap_init(&q, p->cap * 2);
    for (size_t i = 0; i < p->cap; i++)
      if (p->key[i] != NULL) pmap_put(&q, p->key[i], p->val[i]);
    pmap_free(p);
    *p = q;
  }
  size_t i = pmap_slot(p, k);
  if (p->key[i] == NULL) p->n++;
  p->key[i] = k;
  p->val[i] = v;
}

/* ---- flattening -------------------------------------------------------------------------- */

typedef struct {
  cs_model *m;
  pmap nodes;            /* constr_t* -> node id */
  /* slot mode (root phase / single trees): every terminal is a model variable */
  int slot_mode;
  struct constr_t **slot_term; /* variable -> terminal */
  size_t n_slots, cap_slots;
  /* attached mode: terminals with env are variables by env index */
  struct env_t *env;
  size_t env_n;
  /* clause slots met on the root path: wand_expr_t* -> clause id */
  pmap *clause_ids;
  int32_t n_clauses;
} flat;

static int32_t flat_node(flat *f, struct constr_t *c, int in_root_path) {
  int32_t id = pmap_get(&f->nodes, c);
  if (id >= 0) return id;
  int op = op_of(c);
  if (op < 0) { /* terminal */
    struct env_t *e = c->constr.term.env;
    if (!f->slot_mode && e != NULL && e >= f->env && e < f->env + f->env_n) {
      id = f->m->var_node[e - f->env];
    } else if (f->slot_mode && c->constr.term.val.lo != c->constr.term.val.hi) {
      /* an open terminal is a variable of the temporary model; a single value can only
       * conflict, never narrow, so it is a constant whatever it belongs to */
      char name[32];
      snprintf(name, sizeof name, "t%zu", f->n_slots);
      int32_t v = cs_model_add_var(f->m, name, cs_interval(c->constr.term.val.lo, c->constr.term.val.hi));
      if (f->n_slots == f->cap_slots) {
        f->cap_slots = f->cap_slots ? f->cap_slots * 2 : 256;
        f->slot_term = (struct constr_t **)realloc(f->slot_term, f->cap_slots * sizeof *f->slot_term);
      }
      f->slot_term[f->n_slots++] = c;
      id = f->m->var_node[v];
    } else {
      id = cs_model_add_node(f->m, CS_OP_CONST, c->constr.term.val.lo, c->constr.term.val.hi);
    }
  } else if (op == CS_OP_WAND) {
    size_t n = c->constr.wand.length;
    int32_t *kids = (int32_t *)malloc((n ? n : 1) * sizeof *kids);
    for (size_t i = 0; i < n; i++) {
      struct wand_expr_t *e = &c->constr.wand.elems[i];
      int is_wand = e->constr->type->op == 'A';
      kids[i] = flat_node(f, e->constr, in_root_path && is_wand);
      if (in_root_path && !is_wand) {
        if (f->clause_ids != NULL) pmap_put(f->clause_ids, e, f->n_clauses);
        f->n_clauses++;
      }
    }
    id = cs_model_add_wand(f->m, kids, (int32_t)n);
    free(kids);
  } else {
    int32_t l = flat_node(f, c->constr.expr.l, 0);
    int32_t r = c->constr.expr.r != NULL ? flat_node(f, c->constr.expr.r, 0) : -1;
    id = cs_model_add_node(f->m, op, l, r);
  }
  pmap_put(&f->nodes, c, id);
  return id;
}

/* a model whose variables are the terminals below `constr` (root phase, single trees) */
static void flat_slots(flat *f, struct constr_t *constr) {
  memset(f, 0, sizeof *f);
  f->m = cs_model_new();
  f->slot_mode = 1;
  pmap_init(&f->nodes, 1024);
  int32_t top = flat_node(f, constr, 1);
  if (f->m->nodes[top].op != CS_OP_WAND) top = cs_model_add_wand(f->m, &top, 1);
  f->m->root = top;
}

static void flat_done(flat *f) {
  pmap_free(&f->nodes);
  free(f->slot_term);
  f->slot_term = NULL;
}

/* ---- state -------------------------------------------------------------------------------- */

static struct constr_t *g_root;   /* last root passed to propagate() */
static struct env_t *g_env;
static size_t g_size;
static csgpu_model *g_model;      /* attached search model */
static int g_trace = -1;           /* CSOLVE_DROPIN_TRACE=1: entry points on stderr */
#define TRACE(...) do { if (g_trace < 0) g_trace = getenv("CSOLVE_DROPIN_TRACE") != NULL; if (g_trace) { fprintf(stderr, __VA_ARGS__); fflush(stderr); } } while (0)
static csgpu_val *g_state, *g_out;
static uint64_t g_calls[4];

static void fatal_gpu(const char *what) { print_fatal("%s: %s", what, csgpu_last_error()); }

void csolve_dropin_counters(uint64_t out[4]) { memcpy(out, g_calls, sizeof g_calls); }

void csolve_dropin_detach(void) {
  csgpu_model_free(g_model);
  g_model = NULL;
  free(g_state);
  free(g_out);
  g_state = g_out = NULL;
  g_env = NULL;
  g_size = 0;
}

static int attach(struct env_t *env, size_t size, struct constr_t *root, int at_root) {
  TRACE("[dropin] attach size=%zu at_root=%d\n", size, at_root);
  csolve_dropin_detach();
  flat f;
  memset(&f, 0, sizeof f);
  f.m = cs_model_new();
  f.env = env;
  f.env_n = size;
  pmap_init(&f.nodes, 1 << 12);
  pmap clause_ids;
  pmap_init(&clause_ids, 1 << 12);
  f.clause_ids = &clause_ids;
  for (size_t i = 0; i < size; i++) {
    struct val_t v = env[i].val->constr.term.val;
    int32_t id = cs_model_add_var(f.m, env[i].key ? env[i].key : "?", cs_interval(v.lo, v.hi));
    f.m->prio[id] = env[i].prio;
    /* the variable's own terminal maps to its VAR node */
    pmap_put(&f.nodes, env[i].val, f.m->var_node[id]);
  }
  int32_t top = flat_node(&f, root, 1);
  if (f.m->nodes[top].op != CS_OP_WAND) top = cs_model_add_wand(f.m, &top, 1);
  f.m->root = top;
  /* clause index from the trees, per-variable lists from the DRIVER's own lists */
  if (cs_model_index(f.m) != 0) print_fatal("%s", f.m->err);
  size_t total = 0;
  for (size_t i = 0; i < size; i++) total += env[i].clauses.length;
  free(f.m->list);
  f.m->list = (int32_t *)malloc((total ? total : 1) * sizeof(int32_t));
  total = 0;
  for (size_t i = 0; i < size; i++) {
    f.m->list_off[i] = (int32_t)total;
    for (size_t j = 0; j < env[i].clauses.length; j++) {
      int32_t c = pmap_get(&clause_ids, env[i].clauses.elems[j]);
      if (c < 0) print_fatal("clause of %s is not an element of the root", env[i].key);
      f.m->list[total++] = c;
    }
  }
  f.m->list_off[size] = (int32_t)total;
  pmap_free(&clause_ids);
  flat_done(&f);

  if (csgpu_model_from_host(f.m, 1, at_root, &g_model) != CSGPU_OK) fatal_gpu("attach");
  if (csgpu_model_finalize(g_model) != CSGPU_OK) fatal_gpu("attach");
  g_env = env;
  g_size = size;
  g_root = root;
  g_state = (csgpu_val *)malloc((size ? size : 1) * sizeof *g_state);
  g_out = (csgpu_val *)malloc((size ? size : 1) * sizeof *g_out);
  TRACE("[dropin] attached\n");
  return 0;
}

/* explicit attach: called by the driver between clauses_init() and solve(), i.e. at the root */
int csolve_dropin_attach(struct env_t *env, size_t size, struct constr_t *root) { return attach(env, size, root, 1); }

/* find the environment array from the terminals under the last root (zero-patch attach) */
static void collect_env(struct constr_t *c, pmap *seen, struct env_t **lo, struct env_t **hi) {
  if (pmap_get(seen, c) >= 0) return;
  pmap_put(seen, c, 1);
  int op = op_of(c);
  if (op < 0) {
    struct env_t *e = c->constr.term.env;
    if (e != NULL) {
      if (*lo == NULL || e < *lo) *lo = e;
      if (*hi == NULL || e > *hi) *hi = e;
    }
  } else if (op == CS_OP_WAND) {
    for (size_t i = 0; i < c->constr.wand.length; i++) collect_env(c->constr.wand.elems[i].constr, seen, lo, hi);
  } else {
    collect_env(c->constr.expr.l, seen, lo, hi);
    if (c->constr.expr.r != NULL) collect_env(c->constr.expr.r, seen, lo, hi);
  }
}

static void lazy_attach(const struct clause_list_t *clauses) {
  if (g_root == NULL) print_fatal("propagate_clauses before propagate: no root to attach to");
  pmap seen;
  pmap_init(&seen, 1 << 12);
  struct env_t *lo = NULL, *hi = NULL;
  collect_env(g_root, &seen, &lo, &hi);
  pmap_free(&seen);
  /* the list being propagated belongs to a variable too (e.g. "<obj>") */
  struct env_t *owner = (struct env_t *)((char *)clauses - offsetof(struct env_t, clauses));
  if (lo == NULL || owner < lo) lo = owner;
  if (hi == NULL || owner > hi) hi = owner;
  /* the driver has already bound the branching variable: not the root state */
  attach(lo, (size_t)(hi - lo) + 1, g_root, 0);
}

/* ---- arithmetic (replaces src/arith.c) -------------------------------------------------------
 * The scalar layer is an API of its own: strategy.c:87-105, objective.c:110-119, conflict.c:145 and
 * parser.y:220 call it with plain integers on the host.  It is the same source the kernels use
 * (cs_arith.h is __host__ __device__). */
domain_t neg(domain_t a) { return cs_neg(a); }
domain_t add(domain_t a, domain_t b) { return cs_add(a, b); }
domain_t mul(domain_t a, domain_t b) { return cs_mul(a, b); }
domain_t min(domain_t a, domain_t b) { return cs_min(a, b); }
domain_t max(domain_t a, domain_t b) { return cs_max(a, b); }

/* ---- propagate_clauses ------------------------------------------------------------------------ */

prop_result_t propagate_clauses(const struct clause_list_t *clauses) {
  conflict_reset();
  if (g_model == NULL) lazy_attach(clauses);
  struct env_t *owner = (struct env_t *)((char *)clauses - offsetof(struct env_t, clauses));
  if (owner < g_env || owner >= g_env + g_size) print_fatal("propagate_clauses: list of an unknown variable");
  const int32_t var = (int32_t)(owner - g_env);
  for (size_t i = 0; i < g_size; i++) {
    g_state[i].lo = g_env[i].val->constr.term.val.lo;
    g_state[i].hi = g_env[i].val->constr.term.val.hi;
  }
  csgpu_node node = { var, g_state[var].lo, g_state[var].hi, 0 };
  csgpu_result res;
  g_calls[0]++;
  TRACE("[dropin] propagate_clauses var=%d [%d,%d]\n", node.var, node.lo, node.hi);
  if (csgpu_propagate_one(g_model, g_state, node, g_out, &res) != CSGPU_OK) fatal_gpu("propagate_clauses");
  TRACE("[dropin]   -> status %d props %d rounds %d\n", res.status, res.props, res.rounds);
  props += (uint64_t)res.props; /* narrowings of an inconsistent node count too (propagate.c:78) */
  if (res.status < 0) return PROP_ERROR;
  for (size_t i = 0; i < g_size; i++)
    if (g_out[i].lo != g_state[i].lo || g_out[i].hi != g_state[i].hi) {
      struct val_t v = { g_out[i].lo, g_out[i].hi };
      bind(&g_env[i], v, NULL);
    }
  return res.props;
}

/* ---- trees whose terminals are the variables (root phase, single operators) ------------------ */

/* sweeps to the fixpoint of `constr` pushed with `want`; writes the narrowed terminals back
 * (bind() when the terminal has an environment, plain store otherwise: propagate.c:75-83) */
static prop_result_t propagate_tree(struct constr_t *constr, struct val_t want, const struct wand_expr_t *clause,
                                    int recurse) {
  flat f;
  flat_slots(&f, constr);
  csgpu_model *gm = NULL;
  cs_model *hm = f.m;
  if (csgpu_model_from_host(hm, 0, 0, &gm) != CSGPU_OK) fatal_gpu("propagate");
  if (want.lo != 1 || want.hi != 1) {
    /* a single tree pushed with an arbitrary value: one clause, its want */
    if (csgpu_model_num_clauses(gm) < 0) fatal_gpu("propagate");
    hm->clause_want = (cs_val *)malloc((size_t)(hm->n_clauses ? hm->n_clauses : 1) * sizeof(cs_val));
    for (int32_t c = 0; c < hm->n_clauses; c++) hm->clause_want[c] = cs_interval(want.lo, want.hi);
  }
  int32_t status = 0;
  TRACE("[dropin] tree propagate, %d clauses\n", hm->n_clauses);
  if (csgpu_model_root_propagate(gm, &status) != CSGPU_OK) fatal_gpu("propagate");
  TRACE("[dropin]   -> %d\n", status);
  prop_result_t total = status;
  struct env_t **changed = NULL;
  size_t n_changed = 0;
  if (status >= 0) {
    total = 0;
    changed = (struct env_t **)malloc((f.n_slots ? f.n_slots : 1) * sizeof *changed);
    for (size_t i = 0; i < f.n_slots; i++) {
      struct constr_t *t = f.slot_term[i];
      cs_val d = hm->dom[i];
      if (d.lo == t->constr.term.val.lo && d.hi == t->constr.term.val.hi) continue;
      struct val_t v = { d.lo, d.hi };
      total++;
      if (t->constr.term.env != NULL) {
        bind(t->constr.term.env, v, clause);
        props++;
        changed[n_changed++] = t->constr.term.env;
      } else {
        t->constr.term.val = v;
      }
    }
  }
  csgpu_model_free(gm);
  flat_done(&f);
  /* propagate.c:44-54: a bound variable propagates into its clause lists */
  for (size_t i = 0; recurse && total >= 0 && i < n_changed; i++) {
    if (changed[i]->clauses.length == 0) continue;
    prop_result_t p = propagate_clauses(&changed[i]->clauses);
    total = p == PROP_ERROR ? PROP_ERROR : total + p;
  }
  free(changed);
  return total;
}

prop_result_t propagate(struct constr_t *constr, size_t limit) {
  (void)limit; /* the device runs to the fixpoint */
  g_root = constr;
  g_calls[1]++;
  struct val_t t = { 1, 1 };
  return propagate_tree(constr, t, NULL, 0);
}

/* ---- eval ----------------------------------------------------------------------------------------- */

static struct val_t eval_tree(const struct constr_t *constr) {
  struct val_t out = { 0, 1 };
  g_calls[2]++;
  if (g_model != NULL && constr == g_root) {
    /* update_solution (csolve.c:226): the attached root on the driver's current domains */
    for (size_t i = 0; i < g_size; i++) {
      g_state[i].lo = g_env[i].val->constr.term.val.lo;
      g_state[i].hi = g_env[i].val->constr.term.val.hi;
    }
    if (csgpu_model_set_domains(g_model, g_state) != CSGPU_OK) fatal_gpu("eval");
  }
  flat f;
  if (g_model != NULL && constr == g_root) {
    int n = csgpu_model_num_clauses(g_model);
    csgpu_val *vals = (csgpu_val *)malloc((size_t)(n > 0 ? n : 1) * sizeof *vals);
    if (csgpu_model_eval_clauses_host(g_model, vals) != CSGPU_OK) fatal_gpu("eval");
    int any_false = 0, all_true = 1;
    for (int c = 0; c < n; c++) {
      cs_val v = cs_interval(vals[c].lo, vals[c].hi);
      any_false |= cs_is_false(v);
      all_true &= cs_is_true(v);
    }
    free(vals);
    cs_val r = cs_tv(all_true && !any_false, any_false);
    out.lo = r.lo;
    out.hi = r.hi;
    return out;
  }
  flat_slots(&f, (struct constr_t *)constr);
  csgpu_model *gm = NULL;
  if (csgpu_model_from_host(f.m, 0, 0, &gm) != CSGPU_OK) fatal_gpu("eval");
  int n = csgpu_model_num_clauses(gm);
  csgpu_val *vals = (csgpu_val *)malloc((size_t)(n > 0 ? n : 1) * sizeof *vals);
  if (n < 0 || csgpu_model_eval_clauses_host(gm, vals) != CSGPU_OK) fatal_gpu("eval");
  if (constr->type->op == 'A') { /* eval_wand over the clause values (eval.c:233-255) */
    int any_false = 0, all_true = 1;
    for (int c = 0; c < n; c++) {
      cs_val v = cs_interval(vals[c].lo, vals[c].hi);
      any_false |= cs_is_false(v);
      all_true &= cs_is_true(v);
    }
    cs_val r = cs_tv(all_true && !any_false, any_false);
    out.lo = r.lo;
    out.hi = r.hi;
  } else {
    out.lo = vals[0].lo;
    out.hi = vals[0].hi;
  }
  free(vals);
  csgpu_model_free(gm);
  flat_done(&f);
  return out;
}

struct val_t eval_term(const struct constr_t *constr) { return constr->constr.term.val; } /* a field read */

#define EVAL_VIA_GPU(NAME)                                                                         \
  struct val_t eval_##NAME(const struct constr_t *constr) { return eval_tree(constr); }
EVAL_VIA_GPU(eq) EVAL_VIA_GPU(lt) EVAL_VIA_GPU(neg) EVAL_VIA_GPU(add) EVAL_VIA_GPU(mul)
EVAL_VIA_GPU(not) EVAL_VIA_GPU(and) EVAL_VIA_GPU(or) EVAL_VIA_GPU(wand)

struct val_t eval_confl(const struct constr_t *constr) {
  (void)constr;
  print_fatal("conflict clauses are not supported by the GPU propagator (run with -c false)");
  struct val_t v = { 0, 1 };
  return v;
}

/* ---- single-operator propagate ------------------------------------------------------------------- */

#define PROP_VIA_GPU(NAME)                                                                         \
  prop_result_t propagate_##NAME(struct constr_t *constr, struct val_t val, const struct wand_expr_t *clause) { \
    g_calls[3]++;                                                                                  \
    return propagate_tree(constr, val, clause, 1);                                                 \
  }
PROP_VIA_GPU(term) PROP_VIA_GPU(eq) PROP_VIA_GPU(lt) PROP_VIA_GPU(neg) PROP_VIA_GPU(add) PROP_VIA_GPU(mul)
PROP_VIA_GPU(not) PROP_VIA_GPU(and) PROP_VIA_GPU(or)

/* propagate.c:379-392: only "true" is pushed into a wide-and */
prop_result_t propagate_wand(struct constr_t *constr, struct val_t val, const struct wand_expr_t *clause) {
  if (!(val.lo > 0 || val.hi < 0)) return 0;
  g_calls[3]++;
  return propagate_tree(constr, val, clause, 1);
}

prop_result_t propagate_confl(struct constr_t *constr, struct val_t val, const struct wand_expr_t *clause) {
  (void)constr; (void)val; (void)clause;
  print_fatal("conflict clauses are not supported by the GPU propagator (run with -c false)");
  return PROP_ERROR;
}
