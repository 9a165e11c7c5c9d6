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
 *   - on inconsistency the variable whose domain emptied gets prio++ and strategy_var_order_update()
 *     (propagate_term_confl, reference src/propagate.c:33-41): the device reports one such variable.  The
 *     reference additionally bumps every variable on its recursion stack (propagate.c:44-54): the chain of
 *     variables from the assignment to the failure.  The device's trail gives such a chain (bump_failure_chain /
 *     deliver_causes below); it need not be the one the depth-first order walks, so default-flag (-f true) runs
 *     are valid searches of about the reference's length but not call for call the reference's; -f false runs are.
 *   - propagate(root, limit): at most limit + 1 device sweeps (csgpu_model_root_propagate_limit).
 * Sibling batching (CSOLVE_DROPIN_SIBLINGS=1): the driver tries the values of a variable one after the other
 * (step_val, csolve.c:331-338), each through bind + propagate_clauses.  The first such call propagates the next
 * two values in the driver's order in one launch (csgpu_propagate_values), later ones twice as many; calls are
 * served from the batch as long as the other variables' domains are what they were (checked on every call).
 * Same search, call for call; off by default because it does not pay on the driver's depth-first descent.
 * Conflict learning (-c true): see learning() below.  Not reproduced: the in-search normalise/patch tail.
 */
#include "../../include/csolve_dropin.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "cs_device.h"
#include "cs_internal.h"

/* ---- provided by the reference driver ------------------------------------------------ */
extern void bind(struct env_t *var, struct val_t val, const struct wand_expr_t *clause);
extern void strategy_var_order_update(struct env_t *e);
extern void conflict_reset(void);
extern void conflict_create(struct env_t *var, const struct wand_expr_t *clause); /* conflict.c:319-361 */
extern int strategy_create_conflicts(void);                                      /* strategy.c (bool) */
extern _Bool strategy_prefer_failing(void);                                      /* strategy.c:49 */
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
  case 'C': return CS_OP_CONFL;
  default:
    print_fatal("constraint type without a device implementation: %02x", c->type->op);
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
  } else if (op == CS_OP_CONFL) { /* a learnt clause: terminals and the values they must not all have */
    size_t n = c->constr.confl.length;
    int32_t *terms = (int32_t *)malloc((n ? n : 1) * sizeof *terms);
    int32_t *vals = (int32_t *)malloc((n ? n : 1) * sizeof *vals);
    for (size_t i = 0; i < n; i++) {
      terms[i] = flat_node(f, c->constr.confl.elems[i].var, 0);
      vals[i] = c->constr.confl.elems[i].val.lo;
    }
    id = cs_model_add_confl(f->m, terms, vals, (int32_t)n);
    free(terms);
    free(vals);
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
static const struct wand_expr_t **g_clause_ptr; /* device clause id -> the driver's clause */
static int32_t g_n_clause_ptr;
static size_t g_attached_lists;                 /* entries of all clause lists at attach time: they only grow */
static int32_t *g_trail;
#define CS_TRAIL_CAP 16384
static uint64_t g_conflicts_offered, g_reattached, g_chain_bumps;
static int32_t *g_clause_vars; /* [2 * clauses]: the two variables of a binary clause, -1 otherwise */
static uint64_t g_calls[4];
static uint64_t g_sib_launches, g_sib_served; /* sibling batches launched, calls served from one */

/* layout of the driver's trail entry (reference src/csolve.h:73-79): env_t.binds points at the newest one */
struct binding_view {
  struct env_t *var;
  struct val_t val; /* the variable's value BEFORE the bind */
  size_t level;
  const struct wand_expr_t *clause;
  struct binding_view *prev;
};

/* the last sibling batch: the next `count` values of `var` in the driver's order (step_val, csolve.c:331-338:
 * from the edges of the interval inwards, alternating) on the parent `parent` (var's own slot is ignored).  The
 * first batch of a (variable, parent) holds two values, every further one twice as many: a launch takes as long as
 * its slowest node, and the driver usually descends after one or two values. */
static struct {
  int valid, var, lo, hi; /* the interval being iterated */
  int first_is_lo;        /* the driver's seed parity: value 0 of the iteration is lo (else hi) */
  int next_iter;          /* iteration index of the first value NOT in a batch yet */
  int count;              /* values in this batch */
  int32_t *values;
  csgpu_val *parent, *outs;
  csgpu_result *res;
  size_t cap_rows;
} g_sib;
#define CS_SIBLING_MAX 4096

void csolve_dropin_sibling_counters(uint64_t out[2]) { out[0] = g_sib_launches; out[1] = g_sib_served; }
void csolve_dropin_learning_counters(uint64_t out[2]) { out[0] = g_conflicts_offered; out[1] = g_reattached; }
uint64_t csolve_dropin_chain_bumps(void) { return g_chain_bumps; }

/* where the shim's time goes: [0] attach (flatten + finalize + upload), [1] inside the device calls of
 * propagate_clauses, [2] the rest of propagate_clauses (state marshalling, bind() replay) */
static double g_seconds[3];
static double g_eval_seconds; /* inside the eval entry points (update_solution's root evaluation, single operators) */
static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
void csolve_dropin_seconds(double out[3]) { memcpy(out, g_seconds, sizeof g_seconds); }
double csolve_dropin_eval_seconds(void) { return g_eval_seconds; }

/* per-call device times of propagate_clauses (the first 2^20 calls): where a search's time goes call by call */
#define CS_CALL_TIMES (1 << 20)
static float *g_call_us;
static size_t g_call_n;
static csgpu_result g_last_res; /* the device's record of the call being timed */
static struct { float us; csgpu_result res; } g_slowest[8];
static void note_call(double seconds) {
  if (g_call_us == NULL) g_call_us = (float *)malloc((size_t)CS_CALL_TIMES * sizeof(float));
  if (g_call_us != NULL && g_call_n < CS_CALL_TIMES) g_call_us[g_call_n++] = (float)(seconds * 1e6);
  int at = 0;
  for (int i = 1; i < 8; i++)
    if (g_slowest[i].us < g_slowest[at].us) at = i;
  if ((float)(seconds * 1e6) > g_slowest[at].us) { g_slowest[at].us = (float)(seconds * 1e6); g_slowest[at].res = g_last_res; }
}
static int cmp_float(const void *a, const void *b) { return (*(const float *)a > *(const float *)b) - (*(const float *)a < *(const float *)b); }
/* out = { first call, median, 90th percentile, maximum } in microseconds (0 without calls) */
void csolve_dropin_call_times(double out[4]) {
  out[0] = out[1] = out[2] = out[3] = 0.0;
  if (g_call_n == 0) return;
  out[0] = g_call_us[0];
  float *sorted = (float *)malloc(g_call_n * sizeof(float));
  memcpy(sorted, g_call_us, g_call_n * sizeof(float));
  qsort(sorted, g_call_n, sizeof(float), cmp_float);
  out[1] = sorted[g_call_n / 2];
  out[2] = sorted[(g_call_n * 9) / 10];
  out[3] = sorted[g_call_n - 1];
  free(sorted);
  if (getenv("CSOLVE_DROPIN_SLOW") != NULL) /* what the slowest calls were: status, PROPS, revisions, rounds / failing variable */
    for (int i = 0; i < 8; i++)
      fprintf(stderr, "[dropin] slow call: %.1f us, status %d, props %d, revisions %d, rounds %d\n", g_slowest[i].us,
              g_slowest[i].res.status, g_slowest[i].res.props, g_slowest[i].res.revisions, g_slowest[i].res.rounds);
}

static void fatal_gpu(const char *what) { print_fatal("%s: %s", what, csgpu_last_error()); }

/* Values of expression nodes, valid while no domain changes: the driver's normalize() (normalize.c:67-75,
 * 305-316) asks for the value of every node of the root tree, one eval_<op> call per node, between the two root
 * propagations.  The first such call evaluates EVERY node below the last root on the device in one launch; nodes
 * that normalisation creates afterwards are evaluated (with everything below them) when they are asked for.
 * Dropped at every entry point that can change a domain. */
static pmap g_eval_cache;          /* constr_t* -> index into g_eval_vals */
static csgpu_val *g_eval_vals;
static size_t g_eval_n, g_eval_cap;
static int g_eval_root_done;
static uint64_t g_eval_launches;

static void eval_cache_drop(void) {
  if (g_eval_cache.key != NULL) pmap_free(&g_eval_cache);
  g_eval_n = 0;
  g_eval_root_done = 0;
}

void csolve_dropin_counters(uint64_t out[4]) { memcpy(out, g_calls, sizeof g_calls); }

void csolve_dropin_detach(void) {
  csgpu_model_free(g_model);
  g_model = NULL;
  free(g_state);
  free(g_out);
  g_state = g_out = NULL;
  free(g_sib.parent); free(g_sib.outs); free(g_sib.res); free(g_sib.values);
  memset(&g_sib, 0, sizeof g_sib);
  g_env = NULL;
  g_size = 0;
}

/* The root state: the variables' domains when the last propagate(root, limit) returned (parser.y:64-68 calls it
 * before and after normalize(); nothing narrows a domain between that and solve()).  A lazy attach happens inside the
 * driver's first propagate_clauses, when the branching variable is already bound: the model is built on these
 * domains all the same, so that it is a ROOT model (entailed clauses dropped, forbidden-set and interval-shaving
 * kernels eligible) -- without them a queens search ran on the general kernel, ten times slower per heavy node. */
static pmap g_snap;              /* terminal -> index into g_snap_val */
static struct val_t *g_snap_val;
static size_t g_snap_n, g_snap_cap;

static void snapshot_terms(const struct constr_t *c, pmap *seen) {
  if (pmap_get(seen, c) >= 0) return;
  pmap_put(seen, c, 1);
  const int op = op_of(c);
  if (op < 0) {
    /* keyed by the terminal: the root phase runs before env_generate has given the variables their env_t
     * (parser.y:64-80), and env_t.val points at this very terminal afterwards */
    if (g_snap_n == g_snap_cap) {
      g_snap_cap = g_snap_cap ? g_snap_cap * 2 : 1024;
      g_snap_val = (struct val_t *)realloc(g_snap_val, g_snap_cap * sizeof *g_snap_val);
    }
    g_snap_val[g_snap_n] = c->constr.term.val;
    pmap_put(&g_snap, c, (int32_t)g_snap_n++);
  } else if (op == CS_OP_WAND) {
    for (size_t i = 0; i < c->constr.wand.length; i++) snapshot_terms(c->constr.wand.elems[i].constr, seen);
  } else if (op == CS_OP_CONFL) {
    for (size_t i = 0; i < c->constr.confl.length; i++) snapshot_terms(c->constr.confl.elems[i].var, seen);
  } else {
    snapshot_terms(c->constr.expr.l, seen);
    if (c->constr.expr.r != NULL) snapshot_terms(c->constr.expr.r, seen);
  }
}

static void snapshot_root(const struct constr_t *root) {
  if (g_snap.key != NULL) pmap_free(&g_snap);
  pmap_init(&g_snap, 1 << 12);
  g_snap_n = 0;
  pmap seen;
  pmap_init(&seen, 1 << 12);
  snapshot_terms(root, &seen);
  pmap_free(&seen);
}

static int attach(struct env_t *env, size_t size, struct constr_t *root, int at_root) {
  TRACE("[dropin] attach size=%zu at_root=%d\n", size, at_root);
  const double t_attach = now_s();
  eval_cache_drop();
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
  int snapped = !at_root && g_snap.key != NULL;
  for (size_t i = 0; i < size; i++) {
    struct val_t v = env[i].val->constr.term.val;
    if (!at_root && g_snap.key != NULL) {
      const int32_t k = pmap_get(&g_snap, env[i].val);
      if (k >= 0) v = g_snap_val[k];
      else if (env[i].binds != NULL) snapped = 0; /* a variable outside the root tree that has been bound: not the root state */
    }
    int32_t id = cs_model_add_var(f.m, env[i].key ? env[i].key : "?", cs_interval(v.lo, v.hi));
    f.m->prio[id] = env[i].prio;
    /* the variable's own terminal maps to its VAR node */
    pmap_put(&f.nodes, env[i].val, f.m->var_node[id]);
  }
  int32_t top = flat_node(&f, root, 1);
  if (f.m->nodes[top].op != CS_OP_WAND) top = cs_model_add_wand(f.m, &top, 1);
  f.m->root = top;
  /* learnt conflict clauses live in the variables' lists only (conflict.c:352-358): each becomes one more
   * top-level clause, in the order the lists show them */
  for (size_t i = 0; i < size; i++)
    for (size_t j = 0; j < env[i].clauses.length; j++) {
      struct wand_expr_t *w = env[i].clauses.elems[j];
      if (pmap_get(&clause_ids, w) >= 0 || w->constr->type->op != 'C') continue;
      if (w->constr->constr.confl.length + 1 > CS_MAX_TREE_NODES)
        print_fatal("a conflict clause of %zu elements exceeds the device's tree size", w->constr->constr.confl.length);
      if (cs_model_append_clause(f.m, flat_node(&f, w->constr, 0)) != 0) print_fatal("%s", f.m->err);
      pmap_put(&clause_ids, w, f.n_clauses++);
    }
  /* clause index from the trees, per-variable lists from the DRIVER's own lists */
  if (cs_model_index(f.m) != 0) print_fatal("%s", f.m->err);
  if (f.m->n_clauses != f.n_clauses) print_fatal("attach: %d clauses indexed, %d met", f.m->n_clauses, f.n_clauses);
  free(g_clause_ptr);
  g_clause_ptr = (const struct wand_expr_t **)calloc((size_t)(f.n_clauses ? f.n_clauses : 1), sizeof *g_clause_ptr);
  g_n_clause_ptr = f.n_clauses;
  for (size_t k = 0; k < clause_ids.cap; k++)
    if (clause_ids.key[k] != NULL) g_clause_ptr[clause_ids.val[k]] = (const struct wand_expr_t *)clause_ids.key[k];
  size_t total = 0;
  for (size_t i = 0; i < size; i++) total += env[i].clauses.length;
  g_attached_lists = total;
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
  /* the two variables of every binary clause (for the failure chain below), from the lists */
  free(g_clause_vars);
  g_clause_vars = (int32_t *)malloc((size_t)(f.m->n_clauses ? f.m->n_clauses : 1) * 2 * sizeof(int32_t));
  for (int32_t c = 0; c < 2 * f.m->n_clauses; c++) g_clause_vars[c] = -1;
  for (size_t i = 0; i < size; i++)
    for (int32_t k = f.m->list_off[i]; k < f.m->list_off[i + 1]; k++) {
      int32_t *cv = &g_clause_vars[2 * f.m->list[k]];
      if (cv[0] == -1) cv[0] = (int32_t)i;
      else if (cv[0] >= 0 && cv[1] == -1 && cv[0] != (int32_t)i) cv[1] = (int32_t)i;
      else if (cv[0] != (int32_t)i && cv[1] != (int32_t)i) cv[0] = cv[1] = -2; /* three or more variables */
    }
  pmap_free(&clause_ids);
  flat_done(&f);

  if (csgpu_model_from_host(f.m, 1, at_root || snapped, &g_model) != CSGPU_OK) fatal_gpu("attach");
  if (csgpu_model_finalize(g_model) != CSGPU_OK) fatal_gpu("attach");
  /* every propagate_clauses of the driver is a single-node call: have the resident server up before the first one */
  if (csgpu_internal_server_warm(g_model) != CSGPU_OK) fatal_gpu("attach");
  g_env = env;
  g_size = size;
  g_root = root;
  g_state = (csgpu_val *)malloc((size ? size : 1) * sizeof *g_state);
  g_out = (csgpu_val *)malloc((size ? size : 1) * sizeof *g_out);
  TRACE("[dropin] attached: root model %d (snapshot of %zu variables), interval-shaving kernel %s\n", at_root || snapped, g_snap_n,
        csgpu_model_qualifies(g_model, 7) == 1 ? "eligible" : "not eligible");
  g_seconds[0] += now_s() - t_attach;
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
  } else if (op == CS_OP_CONFL) {
    for (size_t i = 0; i < c->constr.confl.length; i++) collect_env(c->constr.confl.elems[i].var, seen, lo, hi);
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

/* what the device found for one node, handed to the driver: trail, PROPS, failure side effects */
static prop_result_t deliver(int32_t var, const csgpu_result *res, const csgpu_val *out) {
  props += (uint64_t)res->props; /* narrowings of an inconsistent node count too (propagate.c:78) */
  if (res->status < 0) {
    /* propagate_term_confl (propagate.c:33-41): the variable whose domain emptied */
    const int32_t fv = res->rounds;
    if (fv >= 0 && (size_t)fv < g_size) {
      /* (bumping it once more per narrowing it had received in the node -- one recursion frame of the reference
       * each -- was tried and derails the heuristic: queens-32 .. 128 then need 150,000+ calls instead of 1,000 ..
       * 4,000) */
      g_env[fv].prio++;
      strategy_var_order_update(&g_env[fv]);
    }
    return PROP_ERROR;
  }
  for (size_t i = 0; i < g_size; i++)
    if (out[i].lo != g_state[i].lo || out[i].hi != g_state[i].hi) {
      struct val_t v = { out[i].lo, out[i].hi };
      bind(&g_env[i], v, NULL);
    }
  (void)var;
  return res->props;
}

/* Conflict learning (the driver's -c true, its default): a failing node needs the trail -- which clause made
 * which narrowing -- because conflict_create (conflict.c:319-361) walks it.  The device records it
 * (csgpu_propagate_one_traced); the shim replays the narrowings through bind() WITH their clauses, and at the
 * point of failure does what propagate_term_confl does (propagate.c:33-41).  The narrowings come in the order of
 * the device's rounds, not of the reference's depth-first recursion: every conflict derived from them is a
 * valid consequence of the problem, but it need not be the one the reference learns, so CALLS / CONFL of a
 * learning run differ from the pure reference's.  CSOLVE_DROPIN_LEARN=0 switches the trail off (failing nodes
 * then create no conflicts, as before). */
static int learning(void) {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("CSOLVE_DROPIN_LEARN");
    on = e == NULL || atoi(e) != 0;
  }
  return on && strategy_create_conflicts();
}

/* The failure chain (the driver's -f true, its default).  When a node fails the reference bumps the priority of the
 * variable whose domain emptied (propagate_term_confl, propagate.c:33-41) AND of every variable on its recursion
 * stack (propagate_term_recurse, propagate.c:44-54): the variables whose narrowing led, one through the next, from
 * the assignment to the failure.  The device's trail names the clause behind every narrowing, so such a chain can
 * be read off it: from the failing variable to the other variable of the clause that failed it, to the clause that
 * had narrowed THAT one, and so on back to the assignment.  It is a causal chain of the same kind, not necessarily
 * the one the depth-first order would have walked.  CSOLVE_DROPIN_CHAIN=0 switches it off. */
/* CSOLVE_DROPIN_CHAIN=reference: the reference's OWN chain.  A failing node is walked once more in the reference's
 * depth-first order by one wavefront (csgpu_propagate_one_chain, cs_chain.hip.h), which names exactly the variables
 * propagate_term_confl / propagate_term_recurse would bump, in their order, and counts the narrowings made before the
 * failure: the driver's search is then call for call the all-CPU reference's (CALLS, CUTS, PROPS, RESTARTS), at tens
 * of microseconds per failing node instead of a few. */
static int reference_chain(void) {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("CSOLVE_DROPIN_CHAIN");
    on = e != NULL && strcmp(e, "reference") == 0;
  }
  return on && strategy_prefer_failing();
}
static uint64_t g_reference_walks;
uint64_t csolve_dropin_reference_walks(void) { return g_reference_walks; }

static int chain_mode(void) {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("CSOLVE_DROPIN_CHAIN");
    on = e == NULL || strcmp(e, "reference") == 0 || atoi(e) != 0;
  }
  return on && strategy_prefer_failing();
}

static void bump_failure_chain(int32_t assigned, int32_t fail_var, int32_t fail_clause, int32_t fail_index) {
  int32_t v = fail_var, c = fail_clause, at = fail_index;
  for (size_t steps = 0; steps < g_size; steps++) {
    if (c < 0 || c >= g_n_clause_ptr) return;
    const int32_t a = g_clause_vars[2 * c], b = g_clause_vars[2 * c + 1];
    if (a < 0 || b < 0) return; /* not a binary clause: no single predecessor */
    const int32_t u = a == v ? b : (b == v ? a : -1);
    if (u < 0 || u == assigned) return; /* the driver bumps the assigned variable itself (csolve.c:462) */
    int32_t found = -1;
    for (int32_t r = at - 1; r >= 0; r--)
      if (g_trail[4 * r] == u && g_trail[4 * r + 1] != 2) { found = r; break; }
    if (found < 0) return; /* u had its value before this node */
    g_env[u].prio++;
    strategy_var_order_update(&g_env[u]);
    g_chain_bumps++;
    v = u;
    c = g_trail[4 * found + 3];
    at = found;
  }
}

/* the same with the trail of kernel 7 (causes are variables, no clauses: for the chain alone, not for learning) */
static prop_result_t deliver_causes(int32_t var, const csgpu_result *res, const csgpu_val *out, int32_t count) {
  props += (uint64_t)res->props;
  if (res->status >= 0) {
    for (size_t i = 0; i < g_size; i++)
      if (out[i].lo != g_state[i].lo || out[i].hi != g_state[i].hi) {
        struct val_t v = { out[i].lo, out[i].hi };
        bind(&g_env[i], v, NULL);
      }
    return res->props;
  }
  /* the failure: the first record whose move empties its variable (a trail longer than the device keeps -- 2,048
   * records -- is not walked: the emptied variable alone is bumped then) */
  const int32_t have = count <= 2048 && count <= CS_TRAIL_CAP ? count : 0;
  int32_t fail_var = -1, fail_at = -1;
  for (int32_t r = 0; r < have && fail_var < 0; r++) {
    const int32_t v = g_trail[4 * r], kind = g_trail[4 * r + 1], bound = g_trail[4 * r + 2];
    if (v < 0 || (size_t)v >= g_size) continue;
    if (kind == 0 && bound > g_state[v].lo) g_state[v].lo = bound;
    else if (kind == 1 && bound < g_state[v].hi) g_state[v].hi = bound;
    if (g_state[v].lo > g_state[v].hi) { fail_var = v; fail_at = r; }
  }
  if (fail_var < 0 && res->rounds >= 0 && (size_t)res->rounds < g_size) fail_var = res->rounds;
  if (fail_var >= 0) {
    g_env[fail_var].prio++;
    strategy_var_order_update(&g_env[fail_var]);
    /* from the failing variable to the variable that moved its bound, to the one that had moved THAT one's, ... */
    int32_t v = fail_var, at = fail_at;
    for (size_t steps = 0; at >= 0 && steps < g_size; steps++) {
      const int32_t u = g_trail[4 * at + 3];
      if (u < 0 || (size_t)u >= g_size || u == var) break; /* the driver bumps the assigned variable itself */
      int32_t found = -1;
      for (int32_t r = at - 1; r >= 0; r--)
        if (g_trail[4 * r] == u) { found = r; break; }
      if (found < 0) break; /* u had its value before this node */
      g_env[u].prio++;
      strategy_var_order_update(&g_env[u]);
      g_chain_bumps++;
      v = u;
      at = found;
    }
    (void)v;
  }
  return PROP_ERROR;
}

static prop_result_t deliver_traced(int32_t var, const csgpu_result *res, const csgpu_val *out, int32_t count) {
  props += (uint64_t)res->props;
  const int32_t have = count < CS_TRAIL_CAP ? count : CS_TRAIL_CAP;
  int32_t fail_var = -1, fail_clause_id = -1, fail_index = -1;
  const struct wand_expr_t *fail_clause = NULL;
  for (int32_t r = 0; r < have; r++) {
    const int32_t v = g_trail[4 * r], kind = g_trail[4 * r + 1], bound = g_trail[4 * r + 2], c = g_trail[4 * r + 3];
    const struct wand_expr_t *w = c >= 0 && c < g_n_clause_ptr ? g_clause_ptr[c] : NULL;
    if (kind == 2) { /* the device saw the node fail here */
      if (v >= 0 && (size_t)v < g_size) { fail_var = v; fail_clause = w; fail_clause_id = c; fail_index = r; }
      break;
    }
    if (v < 0 || (size_t)v >= g_size) continue;
    struct val_t now = g_env[v].val->constr.term.val;
    if (kind == 0 && bound > now.lo) now.lo = bound;
    else if (kind == 1 && bound < now.hi) now.hi = bound;
    else continue; /* superseded by a record replayed before it */
    if (now.lo > now.hi) { /* propagate_term finds the intersection empty before it binds (propagate.c:64-70) */
      fail_var = v;
      fail_clause = w;
      fail_clause_id = c;
      fail_index = r;
      break;
    }
    bind(&g_env[v], now, w);
  }
  if (res->status < 0) {
    if (fail_var < 0 && res->rounds >= 0 && (size_t)res->rounds < g_size) fail_var = res->rounds;
    if (fail_var >= 0) { /* propagate_term_confl (propagate.c:33-41) */
      g_env[fail_var].prio++;
      strategy_var_order_update(&g_env[fail_var]);
      if (chain_mode() && fail_index >= 0 && count <= CS_TRAIL_CAP) bump_failure_chain(var, fail_var, fail_clause_id, fail_index);
      if (learning() && fail_clause != NULL && count <= CS_TRAIL_CAP) {
        g_conflicts_offered++;
        conflict_create(&g_env[fail_var], fail_clause);
      }
    }
    return PROP_ERROR;
  }
  /* whatever the trail did not cover (it overflowed): bound without a clause, like a decision */
  for (size_t i = 0; i < g_size; i++) {
    const struct val_t now = g_env[i].val->constr.term.val;
    if (out[i].lo != now.lo || out[i].hi != now.hi) {
      struct val_t v = { out[i].lo, out[i].hi };
      if (count <= CS_TRAIL_CAP) print_fatal("propagate_clauses: the trail of variable %zu does not end in its fixpoint", i);
      bind(&g_env[i], v, NULL);
    }
  }
  (void)var;
  return res->props;
}

static prop_result_t propagate_clauses_timed(const struct clause_list_t *clauses);
prop_result_t propagate_clauses(const struct clause_list_t *clauses) {
  eval_cache_drop();
  conflict_reset();
  if (g_model != NULL && learning()) {
    /* clause lists only grow (clause_list_append, util.c:267-271): a longer total means new learnt clauses */
    size_t total = 0;
    for (size_t i = 0; i < g_size; i++) total += g_env[i].clauses.length;
    if (total != g_attached_lists) {
      struct env_t *env = g_env;
      const size_t size = g_size;
      g_reattached++;
      attach(env, size, g_root, 0);
    }
  }
  if (g_model == NULL) lazy_attach(clauses);
  const double t0 = now_s(), dev0 = g_seconds[1];
  const prop_result_t r = propagate_clauses_timed(clauses);
  g_seconds[2] += (now_s() - t0) - (g_seconds[1] - dev0);
  note_call(g_seconds[1] - dev0);
  return r;
}

static prop_result_t propagate_clauses_timed(const struct clause_list_t *clauses) {
  struct env_t *owner = (struct env_t *)((char *)clauses - offsetof(struct env_t, clauses));
  if (owner < g_env || owner >= g_env + g_size) print_fatal("propagate_clauses: list of an unknown variable");
  const int32_t var = (int32_t)(owner - g_env);
  for (size_t i = 0; i < g_size; i++) {
    g_state[i].lo = g_env[i].val->constr.term.val.lo;
    g_state[i].hi = g_env[i].val->constr.term.val.hi;
  }
  g_calls[0]++;
  static int batching = -1;
  /* opt-in: a batch takes as long as its slowest node, and the reference's driver usually descends after the first
   * value, so on its depth-first searches speculation costs more than it saves */
  if (batching < 0) batching = getenv("CSOLVE_DROPIN_SIBLINGS") != NULL && atoi(getenv("CSOLVE_DROPIN_SIBLINGS")) != 0;
  const int32_t k = g_state[var].lo;
  if (batching && !learning() && !chain_mode() && g_state[var].lo == g_state[var].hi) {
    /* the iteration this call belongs to: same variable, every OTHER domain what it was */
    int same = g_sib.valid && g_sib.var == var && k >= g_sib.lo && k <= g_sib.hi;
    for (size_t i = 0; i < g_size && same; i++)
      same = (int32_t)i == var || (g_sib.parent[i].lo == g_state[i].lo && g_sib.parent[i].hi == g_state[i].hi);
    if (same) {
      for (int j = 0; j < g_sib.count; j++)
        if (g_sib.values[j] == k) {
          g_sib_served++;
          TRACE("[dropin] propagate_clauses var=%d value %d: from the sibling batch\n", var, k);
          return deliver(var, &g_sib.res[j], g_sib.outs + (size_t)j * g_size);
        }
    } else {
      /* a new iteration: the interval = the variable's value before step_enter's bind (csolve.c:294-304) */
      const struct binding_view *b = (const struct binding_view *)owner->binds;
      g_sib.valid = 0;
      if (b != NULL && b->var == owner && b->val.lo < b->val.hi && (k == b->val.lo || k == b->val.hi) &&
          (int64_t)b->val.hi - b->val.lo < CS_SIBLING_MAX) {
        if (g_sib.parent == NULL) g_sib.parent = (csgpu_val *)malloc((g_size ? g_size : 1) * sizeof(csgpu_val));
        memcpy(g_sib.parent, g_state, g_size * sizeof(csgpu_val));
        g_sib.valid = 1; g_sib.var = var; g_sib.lo = b->val.lo; g_sib.hi = b->val.hi;
        g_sib.first_is_lo = k == b->val.lo;
        g_sib.next_iter = 0;
        g_sib.count = 0;
      }
    }
    if (g_sib.valid) {
      /* the next values in the driver's order, starting with the one asked for now */
      const int width = g_sib.hi - g_sib.lo + 1;
      int want = g_sib.count == 0 ? 2 : 2 * g_sib.count;
      if (want > width - g_sib.next_iter) want = width - g_sib.next_iter;
      if ((size_t)want > g_sib.cap_rows) {
        free(g_sib.outs); free(g_sib.res); free(g_sib.values);
        g_sib.cap_rows = (size_t)want * 2;
        g_sib.outs = (csgpu_val *)malloc(g_sib.cap_rows * (g_size ? g_size : 1) * sizeof(csgpu_val));
        g_sib.res = (csgpu_result *)malloc(g_sib.cap_rows * sizeof(csgpu_result));
        g_sib.values = (int32_t *)malloc(g_sib.cap_rows * sizeof(int32_t));
      }
      int found = -1;
      for (int j = 0; j < want; j++) {
        const int it = g_sib.next_iter + j;
        /* the parity of the driver's seed is known from the first value it asked for */
        g_sib.values[j] = cs_step_val(cs_interval(g_sib.lo, g_sib.hi), (uint32_t)it, g_sib.first_is_lo ? 0u : 1u);
        if (g_sib.values[j] == k) found = j;
      }
      if (found >= 0) {
        g_sib.parent[var].lo = g_sib.lo;
        g_sib.parent[var].hi = g_sib.hi;
        TRACE("[dropin] propagate_clauses var=%d: sibling batch of %d from value %d\n", var, want, k);
        const double td = now_s();
        if (csgpu_propagate_values(g_model, g_sib.parent, var, g_sib.values, want, g_sib.outs, g_sib.res) != CSGPU_OK)
          fatal_gpu("propagate_clauses");
        g_seconds[1] += now_s() - td;
        g_sib.count = want;
        g_sib.next_iter += want;
        g_sib_launches++;
        return deliver(var, &g_sib.res[found], g_sib.outs + (size_t)found * g_size);
      }
      g_sib.valid = 0; /* not the order this shim predicts: single nodes from here on */
    }
  }
  csgpu_node node = { var, g_state[var].lo, g_state[var].hi, 0 };
  csgpu_result res;
  TRACE("[dropin] propagate_clauses var=%d [%d,%d]\n", node.var, node.lo, node.hi);
  const double td = now_s();
  static int no_causes; /* the model's pair table leaves no room for the trail in LDS: the clause trail instead */
  if (!learning() && chain_mode() && !no_causes && csgpu_model_qualifies(g_model, 7) == 1) {
    /* pure != network: kernel 7's trail, at the latency of the untraced call */
    if (g_trail == NULL) g_trail = (int32_t *)malloc((size_t)CS_TRAIL_CAP * 4 * sizeof(int32_t));
    int32_t count = 0;
    const int rc = csgpu_propagate_one_causes(g_model, g_state, node, g_out, &res, g_trail, CS_TRAIL_CAP, &count);
    if (rc == CSGPU_OK) {
      static int no_walk; /* the model has clauses the reference-order walk does not cover: the causal chain instead */
      if (res.status < 0 && reference_chain() && !no_walk) {
        int32_t st = 0, pr = 0, nb = 0;
        const int rcw = csgpu_propagate_one_chain(g_model, g_state, node, &st, &pr, g_trail, CS_TRAIL_CAP, &nb);
        if (rcw == CSGPU_OK && st < 0) {
          g_seconds[1] += now_s() - td;
          g_reference_walks++;
          props += (uint64_t)pr;
          for (int32_t i = 0; i < nb && i < CS_TRAIL_CAP; i++) {
            const int32_t b = g_trail[i];
            if (b < 0 || (size_t)b >= g_size) continue;
            g_env[b].prio++;
            strategy_var_order_update(&g_env[b]);
            g_chain_bumps++;
          }
          TRACE("[dropin]   -> failed; reference-order walk: %d narrowings, %d variables bumped\n", pr, nb);
          g_last_res = res;
          return PROP_ERROR;
        }
        if (rcw != CSGPU_OK && rcw != CSGPU_E_LIMIT) fatal_gpu("propagate_clauses");
        no_walk = 1;
      }
      g_seconds[1] += now_s() - td;
      TRACE("[dropin]   -> status %d props %d, %d cause records\n", res.status, res.props, count);
      g_last_res = res;
      return deliver_causes(var, &res, g_out, count);
    }
    if (rc != CSGPU_E_LIMIT) fatal_gpu("propagate_clauses");
    no_causes = 1;
  }
  if (learning() || chain_mode()) {
    if (g_trail == NULL) g_trail = (int32_t *)malloc((size_t)CS_TRAIL_CAP * 4 * sizeof(int32_t));
    int32_t count = 0;
    if (csgpu_propagate_one_traced(g_model, g_state, node, g_out, &res, g_trail, CS_TRAIL_CAP, &count) != CSGPU_OK)
      fatal_gpu("propagate_clauses");
    g_seconds[1] += now_s() - td;
    TRACE("[dropin]   -> status %d props %d, %d trail records\n", res.status, res.props, count);
    g_last_res = res;
    return deliver_traced(var, &res, g_out, count);
  }
  if (csgpu_propagate_one(g_model, g_state, node, g_out, &res) != CSGPU_OK) fatal_gpu("propagate_clauses");
  g_seconds[1] += now_s() - td;
  TRACE("[dropin]   -> status %d props %d rounds %d\n", res.status, res.props, res.rounds);
  g_last_res = res;
  return deliver(var, &res, g_out);
}

/* ---- trees whose terminals are the variables (root phase, single operators) ------------------ */

/* sweeps to the fixpoint of `constr` pushed with `want`; writes the narrowed terminals back
 * (bind() when the terminal has an environment, plain store otherwise: propagate.c:75-83) */
static prop_result_t propagate_tree(struct constr_t *constr, struct val_t want, const struct wand_expr_t *clause,
                                    int recurse, int64_t limit) {
  eval_cache_drop();
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
  if (csgpu_model_root_propagate_limit(gm, limit, &status, NULL) != CSGPU_OK) fatal_gpu("propagate");
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
  g_root = constr;
  g_calls[1]++;
  struct val_t t = { 1, 1 };
  const prop_result_t r = propagate_tree(constr, t, NULL, 0, limit > (size_t)0x7ffffff0 ? -1 : (int64_t)limit);
  if (r != PROP_ERROR) snapshot_root(constr);
  return r;
}

/* ---- eval ----------------------------------------------------------------------------------------- */

/* post-order list of the non-terminal nodes below c that are not valued yet */
static void collect_nodes(const struct constr_t *c, pmap *seen, const struct constr_t ***list, size_t *n, size_t *cap) {
  if (c->type->op == ' ' || pmap_get(seen, c) >= 0 || pmap_get(&g_eval_cache, c) >= 0) return;
  pmap_put(seen, c, 1);
  if (c->type->op == 'A') {
    for (size_t i = 0; i < c->constr.wand.length; i++) collect_nodes(c->constr.wand.elems[i].constr, seen, list, n, cap);
  } else if (c->type->op == 'C') {
    /* elements are terminals: nothing below to value */
  } else {
    collect_nodes(c->constr.expr.l, seen, list, n, cap);
    if (c->constr.expr.r != NULL) collect_nodes(c->constr.expr.r, seen, list, n, cap);
  }
  if (*n == *cap) {
    *cap = *cap ? *cap * 2 : 1024;
    *list = (const struct constr_t **)realloc(*list, *cap * sizeof **list);
  }
  (*list)[(*n)++] = c;
}

static void eval_subtree_into_cache(const struct constr_t *top) {
  pmap seen;
  pmap_init(&seen, 1 << 12);
  const struct constr_t **list = NULL;
  size_t n = 0, cap = 0;
  collect_nodes(top, &seen, &list, &n, &cap);
  pmap_free(&seen);
  if (n == 0) { free(list); return; }
  /* a temporary model whose root wide-and has one element per collected node (wide-ands are valued from their
   * elements afterwards: eval_wand, eval.c:233-255) */
  flat f;
  memset(&f, 0, sizeof f);
  f.m = cs_model_new();
  f.slot_mode = 1;
  pmap_init(&f.nodes, 2 * n + 1024);
  int32_t *kids = (int32_t *)malloc(n * sizeof *kids);
  size_t n_kids = 0;
  int32_t *kid_of = (int32_t *)malloc(n * sizeof *kid_of);
  for (size_t i = 0; i < n; i++) {
    kid_of[i] = -1;
    if (list[i]->type->op == 'A') continue;
    kid_of[i] = (int32_t)n_kids;
    kids[n_kids++] = flat_node(&f, (struct constr_t *)list[i], 0);
  }
  csgpu_val *vals = (csgpu_val *)malloc((n_kids ? n_kids : 1) * sizeof *vals);
  if (n_kids > 0) {
    f.m->root = cs_model_add_wand(f.m, kids, (int32_t)n_kids);
    csgpu_model *gm = NULL;
    if (csgpu_model_from_host(f.m, 0, 0, &gm) != CSGPU_OK) fatal_gpu("eval");
    if (csgpu_model_num_clauses(gm) != (int)n_kids) print_fatal("eval: %d clauses for %zu nodes", csgpu_model_num_clauses(gm), n_kids);
    if (csgpu_model_eval_clauses_host(gm, vals) != CSGPU_OK) fatal_gpu("eval");
    g_eval_launches++;
    csgpu_model_free(gm);
  } else {
    cs_model_free(f.m);
  }
  flat_done(&f);
  for (size_t i = 0; i < n; i++) { /* post-order: the elements of a wide-and are valued before it */
    csgpu_val v;
    if (kid_of[i] >= 0) {
      v = vals[kid_of[i]];
    } else {
      int any_false = 0, all_true = 1;
      for (size_t j = 0; j < list[i]->constr.wand.length; j++) {
        const struct constr_t *e = list[i]->constr.wand.elems[j].constr;
        cs_val ev;
        if (e->type->op == ' ') ev = cs_interval(e->constr.term.val.lo, e->constr.term.val.hi);
        else { const int32_t a = pmap_get(&g_eval_cache, e); ev = cs_interval(g_eval_vals[a].lo, g_eval_vals[a].hi); }
        any_false |= cs_is_false(ev);
        all_true &= cs_is_true(ev);
      }
      const cs_val r = cs_tv(all_true && !any_false, any_false);
      v.lo = r.lo; v.hi = r.hi;
    }
    if (g_eval_n == g_eval_cap) {
      g_eval_cap = g_eval_cap ? g_eval_cap * 2 : 4096;
      g_eval_vals = (csgpu_val *)realloc(g_eval_vals, g_eval_cap * sizeof *g_eval_vals);
    }
    g_eval_vals[g_eval_n] = v;
    pmap_put(&g_eval_cache, list[i], (int32_t)g_eval_n++);
  }
  free(vals); free(kids); free(kid_of); free(list);
}

static struct val_t eval_tree_timed(const struct constr_t *constr);
static struct val_t eval_tree(const struct constr_t *constr) {
  const double t0 = now_s();
  const struct val_t v = eval_tree_timed(constr);
  g_eval_seconds += now_s() - t0;
  return v;
}

static struct val_t eval_tree_timed(const struct constr_t *constr) {
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
  /* every non-terminal node below `top` becomes a clause of a temporary model: one launch values them all */
  if (g_eval_cache.key == NULL) pmap_init(&g_eval_cache, 1 << 12);
  int32_t at = pmap_get(&g_eval_cache, constr);
  for (int pass = 0; at < 0 && pass < 2; pass++) {
    const struct constr_t *top = constr;
    if (pass == 0) {
      if (g_eval_root_done || g_root == NULL) continue;
      top = g_root; /* first question of a normalisation pass: value the whole root tree */
      g_eval_root_done = 1;
    }
    eval_subtree_into_cache(top);
    at = pmap_get(&g_eval_cache, constr);
  }
  if (at < 0) print_fatal("eval: node was not valued");
  out.lo = g_eval_vals[at].lo;
  out.hi = g_eval_vals[at].hi;
  return out;
}

struct val_t eval_term(const struct constr_t *constr) { return constr->constr.term.val; } /* a field read */

#define EVAL_VIA_GPU(NAME)                                                                         \
  struct val_t eval_##NAME(const struct constr_t *constr) { return eval_tree(constr); }
EVAL_VIA_GPU(eq) EVAL_VIA_GPU(lt) EVAL_VIA_GPU(neg) EVAL_VIA_GPU(add) EVAL_VIA_GPU(mul)
EVAL_VIA_GPU(not) EVAL_VIA_GPU(and) EVAL_VIA_GPU(or) EVAL_VIA_GPU(wand) EVAL_VIA_GPU(confl)

/* ---- single-operator propagate ------------------------------------------------------------------- */

#define PROP_VIA_GPU(NAME)                                                                         \
  prop_result_t propagate_##NAME(struct constr_t *constr, struct val_t val, const struct wand_expr_t *clause) { \
    g_calls[3]++;                                                                                  \
    return propagate_tree(constr, val, clause, 1, -1);                                             \
  }
PROP_VIA_GPU(term) PROP_VIA_GPU(eq) PROP_VIA_GPU(lt) PROP_VIA_GPU(neg) PROP_VIA_GPU(add) PROP_VIA_GPU(mul)
PROP_VIA_GPU(not) PROP_VIA_GPU(and) PROP_VIA_GPU(or)

/* propagate.c:379-392: only "true" is pushed into a wide-and */
prop_result_t propagate_wand(struct constr_t *constr, struct val_t val, const struct wand_expr_t *clause) {
  if (!(val.lo > 0 || val.hi < 0)) return 0;
  g_calls[3]++;
  return propagate_tree(constr, val, clause, 1, -1);
}

/* propagate.c:461-471: only "true" is pushed into a conflict clause */
prop_result_t propagate_confl(struct constr_t *constr, struct val_t val, const struct wand_expr_t *clause) {
  if (!(val.lo > 0 || val.hi < 0)) return 0;
  g_calls[3]++;
  return propagate_tree(constr, val, clause, 1, -1);
}
