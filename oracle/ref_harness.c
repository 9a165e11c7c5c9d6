/* ref_harness.c -- drives the COMPILED REFERENCE.  TEST INFRASTRUCTURE.
 *
 * Built only in the authoring container by oracle/Makefile, which compiles the
 * reference's own sources from /root/reference/src where they lie (never copied)
 * and links them with this driver into oracle/_ref/csolve_ref.  The reference's
 * text front end is flex/bison output that is not in its tree, so this driver
 * feeds the reference through the repo's recursive-descent parser
 * (csolve_amd/csrc/cs_frontend.c) with a builder that performs the semantic
 * actions of reference src/parser.y on the reference's own data structures,
 * then replays the `Input` action (parser.y:55-92).  main_name() (reference
 * src/main.c:136) is the only symbol supplied on the reference's behalf.
 *
 * Commands (all print to stdout):
 *   csolve_ref solve  <file> [-c b] [-f b] [-r n] [-o order] [-w b] [-t secs]
 *        run the reference search; the reference's own output, then a line
 *        "@STATS {json}".
 *   csolve_ref model  <file> <out.model>
 *        dump the reference's post-root trees and clause lists as a cs_model file.
 *   csolve_ref bench  <file> <instances.in> <results.out> [-c false]
 *        replay given node instances through the reference's propagate_clauses(), timed
 *        (the CPU baseline of bench.py; prints "@BENCH {json}").
 *   csolve_ref walk   <file> <seed> <count> <out.bin> [-c false]
 *        seeded random assignment walks through the reference's
 *        propagate_clauses(); writes node instances (before, var, value,
 *        status/props, after).
 */
#include "csolve.h"
#include "parser_support.h"

#include "../csolve_amd/csrc/cs_frontend.h"
#include "../csolve_amd/csrc/cs_model.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

const char *main_name(void) { return "csolve_ref"; }

/* ---- builder: parser.y semantic actions on reference structures ------------ */

static struct constr_t *root_wand;

static struct constr_t *new_node(struct constr_t v) {
  struct constr_t *c = (struct constr_t *)alloc(sizeof(struct constr_t));
  *c = v;
  return c;
}

static void *r_num(void *ctx, int32_t value) {
  (void)ctx;
  return new_node(CONSTRAINT_TERM(VALUE(value)));
}

static void *r_ident(void *ctx, const char *name) {
  (void)ctx;
  struct env_t *var = vars_find_key(name);
  if (var != NULL) return var->val;
  struct constr_t *c = new_node(CONSTRAINT_TERM(INTERVAL(DOMAIN_MIN, DOMAIN_MAX)));
  vars_add(name, c);
  return c;
}

static const struct constr_type_t *type_of(int op) {
  switch (op) {
  case CS_OP_EQ: return &CONSTR_EQ;
  case CS_OP_LT: return &CONSTR_LT;
  case CS_OP_NEG: return &CONSTR_NEG;
  case CS_OP_ADD: return &CONSTR_ADD;
  case CS_OP_MUL: return &CONSTR_MUL;
  case CS_OP_NOT: return &CONSTR_NOT;
  case CS_OP_AND: return &CONSTR_AND;
  case CS_OP_OR: return &CONSTR_OR;
  default: fprintf(stderr, "csolve_ref: bad operator %d\n", op); exit(2);
  }
}

static void *r_unary(void *ctx, int op, void *child) {
  (void)ctx;
  struct constr_t *c = (struct constr_t *)alloc(sizeof(struct constr_t));
  c->type = type_of(op);
  c->constr.expr.l = (struct constr_t *)child;
  c->constr.expr.r = NULL;
  return c;
}

static void *r_binary(void *ctx, int op, void *l, void *r) {
  (void)ctx;
  struct constr_t *c = (struct constr_t *)alloc(sizeof(struct constr_t));
  c->type = type_of(op);
  c->constr.expr.l = (struct constr_t *)l;
  c->constr.expr.r = (struct constr_t *)r;
  return c;
}

static void *r_wand(void *ctx, void **elems, size_t n) {
  (void)ctx;
  struct wand_expr_t *e = (struct wand_expr_t *)malloc((n ? n : 1) * sizeof(struct wand_expr_t));
  for (size_t i = 0; i < n; i++)
    e[i] = (struct wand_expr_t){ .constr = (struct constr_t *)elems[i], .orig = (struct constr_t *)elems[i], .prop_tag = 0 };
  return new_node(CONSTRAINT_WAND(n, n ? e : NULL));
}

static void r_weigh(void *ctx, void *expr, int32_t weight) {
  (void)ctx;
  if (strategy_compute_weights())
    vars_weighten((struct constr_t *)expr, weight / max(1, vars_count((struct constr_t *)expr)));
}

static void *r_objective(void *ctx, int kind, void *expr) {
  (void)ctx;
  switch (kind) {
  case CS_OBJ_ANY:
    objective_init(OBJ_ANY, &shared()->objective_best);
    return new_node(CONSTRAINT_TERM(VALUE(1)));
  case CS_OBJ_ALL:
    objective_init(OBJ_ALL, &shared()->objective_best);
    return new_node(CONSTRAINT_TERM(VALUE(1)));
  case CS_OBJ_MIN:
    objective_init(OBJ_MIN, &shared()->objective_best);
    vars_add("<obj>", objective_val());
    return new_node(CONSTRAINT_EXPR(EQ, (struct constr_t *)expr, objective_val()));
  default:
    objective_init(OBJ_MAX, &shared()->objective_best);
    vars_add("<obj>", objective_val());
    return new_node(CONSTRAINT_EXPR(EQ, objective_val(), (struct constr_t *)expr));
  }
}

static void r_constraint(void *ctx, void *expr) {
  (void)ctx;
  if (root_wand == NULL) root_wand = new_node(CONSTRAINT_WAND(0, NULL));
  size_t n = ++root_wand->constr.wand.length;
  root_wand->constr.wand.elems =
      (struct wand_expr_t *)realloc(root_wand->constr.wand.elems, n * sizeof(struct wand_expr_t));
  root_wand->constr.wand.elems[n - 1] =
      (struct wand_expr_t){ .constr = (struct constr_t *)expr, .orig = (struct constr_t *)expr, .prop_tag = 0 };
}

/* ---- option defaults: reference src/main.c:51-130 -------------------------- */

struct options {
  bool conflicts, prefer_failing, weighten;
  uint64_t restart_freq;
  enum order_t order;
  uint32_t time_max;
};

static void reference_init(const struct options *o) {
  bind_init(BIND_STACK_SIZE_DEFAULT);
  strategy_create_conflicts_init(o->conflicts);
  strategy_prefer_failing_init(o->prefer_failing);
  shared_init(WORKERS_MAX_DEFAULT);
  alloc_init(ALLOC_STACK_SIZE_DEFAULT);
  conflict_alloc_init(CONFLICT_ALLOC_STACK_SIZE_DEFAULT);
  strategy_order_init(o->order);
  patch_init(PATCH_STACK_SIZE_DEFAULT);
  strategy_restart_frequency_init(o->restart_freq);
  stats_frequency_init(0);
  timeout_init(o->time_max);
  strategy_compute_weights_init(o->weighten);
}

static char *slurp(const char *path) {
  FILE *f = fopen(path, "rb");
  if (f == NULL) { fprintf(stderr, "csolve_ref: %s: %s\n", path, strerror(errno)); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *s = (char *)malloc((size_t)n + 1);
  if (fread(s, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "csolve_ref: read error\n"); exit(2); }
  s[n] = '\0';
  fclose(f);
  return s;
}

/* the first half of parser.y's Input action (55-84): root propagate, normalise,
 * commit, environment, clause lists, variable order.  Returns the normalised root
 * or NULL if the problem is infeasible at the root. */
static struct constr_t *root_phase(const char *path, size_t *size, struct env_t **env) {
  char *text = slurp(path);
  char err[200];
  cs_builder b = { NULL, r_num, r_ident, r_unary, r_binary, r_wand, r_weigh, r_objective, r_constraint };
  if (cs_parse_text(text, &b, err, sizeof err) != 0) {
    fprintf(stderr, "csolve_ref: error: %s\n", err);
    exit(1);
  }
  free(text);

  *size = var_count();
  prop_result_t prop = propagate(root_wand, *size);
  struct constr_t *norm = root_wand;
  if (prop != PROP_ERROR) {
    struct constr_t *prev;
    do {
      prev = norm;
      norm = normalize(norm);
      prop = propagate(norm, *size);
    } while (norm != prev && prop != PROP_ERROR);
  }
  if (prop == PROP_ERROR) fprintf(stdout, "INFEASIBLE PROBLEM\n");
  bind_commit();
  patch_commit();
  stats_init();
  if (prop == PROP_ERROR) return NULL;
  *env = env_generate();
  clauses_init(norm, NULL);
  strategy_var_order_init(*size, *env);
  return norm;
}

/* ---- model dump ------------------------------------------------------------ */

struct pmap {
  const void **key;
  int32_t *val;
  size_t cap, n;
};

static void pmap_init(struct pmap *p, size_t cap) {
  p->cap = cap; p->n = 0;
  p->key = (const void **)calloc(cap, sizeof *p->key);
  p->val = (int32_t *)malloc(cap * sizeof *p->val);
}

static size_t pmap_slot(const struct pmap *p, const void *k) {
  size_t i = ((uintptr_t)k >> 3) * 11400714819323198485ull % p->cap;
  while (p->key[i] != NULL && p->key[i] != k) i = (i + 1) % p->cap;
  return i;
}

static int32_t pmap_get(const struct pmap *p, const void *k) {
  size_t i = pmap_slot(p, k);
  return p->key[i] == k ? p->val[i] : -1;
}

static void pmap_put(struct pmap *p, const void *k, int32_t v) {
  if ((p->n + 1) * 2 > p->cap) {
    struct pmap q;
    pmap_init(&q, p->cap * 2);
    for (size_t i = 0; i < p->cap; i++)
      if (p->key[i] != NULL) pmap_put(&q, p->key[i], p->val[i]);
    free(p->key); free(p->val);
    *p = q;
  }
  size_t i = pmap_slot(p, k);
  if (p->key[i] == NULL) p->n++;
  p->key[i] = k;
  p->val[i] = v;
}

static struct pmap node_ids, clause_ids;
static struct env_t *env_base;
static int32_t n_clauses_seen;
static int32_t *clause_nodes;
static size_t clause_nodes_cap;

static int32_t op_of(const struct constr_t *c) {
  switch (c->type->op) {
  case OP_EQ: return CS_OP_EQ;
  case OP_LT: return CS_OP_LT;
  case OP_NEG: return CS_OP_NEG;
  case OP_ADD: return CS_OP_ADD;
  case OP_MUL: return CS_OP_MUL;
  case OP_NOT: return CS_OP_NOT;
  case OP_AND: return CS_OP_AND;
  case OP_OR: return CS_OP_OR;
  default: fprintf(stderr, "csolve_ref: cannot dump operator %c\n", c->type->op); exit(2);
  }
}

/* in_root_path: this node is reached from the root through wide-ands only, so its
 * non-wide-and elements are clauses (parser_support.c:351-363) */
static int32_t dump_node(cs_model *m, struct constr_t *c, int in_root_path) {
  int32_t id = pmap_get(&node_ids, c);
  if (id >= 0) return id;
  if (IS_TYPE(TERM, c)) {
    if (c->constr.term.env != NULL) id = m->var_node[c->constr.term.env - env_base];
    else id = cs_model_add_node(m, CS_OP_CONST, get_lo(c->constr.term.val), get_hi(c->constr.term.val));
  } else if (IS_TYPE(WAND, c)) {
    size_t n = c->constr.wand.length;
    int32_t *kids = (int32_t *)malloc((n ? n : 1) * sizeof *kids);
    for (size_t i = 0; i < n; i++) {
      struct wand_expr_t *e = &c->constr.wand.elems[i];
      int is_wand = IS_TYPE(WAND, e->constr);
      kids[i] = dump_node(m, e->constr, in_root_path && is_wand);
      if (in_root_path && !is_wand) {
        if ((size_t)n_clauses_seen == clause_nodes_cap) {
          clause_nodes_cap = clause_nodes_cap ? clause_nodes_cap * 2 : 1024;
          clause_nodes = (int32_t *)realloc(clause_nodes, clause_nodes_cap * sizeof *clause_nodes);
        }
        clause_nodes[n_clauses_seen] = kids[i];
        pmap_put(&clause_ids, e, n_clauses_seen++);
      }
    }
    id = cs_model_add_wand(m, kids, (int32_t)n);
    free(kids);
  } else {
    int32_t l = dump_node(m, c->constr.expr.l, 0);
    int32_t r = c->constr.expr.r != NULL ? dump_node(m, c->constr.expr.r, 0) : -1;
    id = cs_model_add_node(m, op_of(c), l, r);
  }
  pmap_put(&node_ids, c, id);
  return id;
}

static cs_model *dump_model(struct constr_t *norm, size_t size, struct env_t *env) {
  cs_model *m = cs_model_new();
  env_base = env;
  pmap_init(&node_ids, 1 << 16);
  pmap_init(&clause_ids, 1 << 16);
  n_clauses_seen = 0;
  for (size_t i = 0; i < size; i++) {
    struct val_t v = env[i].val->constr.term.val;
    int32_t id = cs_model_add_var(m, env[i].key, cs_interval(v.lo, v.hi));
    m->prio[id] = env[i].prio;
  }
  m->objective = (int32_t)objective();
  m->obj_var = -1;
  if (objective_val() != NULL && objective_val()->constr.term.env != NULL)
    m->obj_var = (int32_t)(objective_val()->constr.term.env - env);
  m->root = dump_node(m, norm, 1);

  /* the reference's own clause lists, in its order */
  m->n_clauses = n_clauses_seen;
  m->clause_node = (int32_t *)malloc((n_clauses_seen ? n_clauses_seen : 1) * sizeof(int32_t));
  memcpy(m->clause_node, clause_nodes, (size_t)n_clauses_seen * sizeof(int32_t));
  m->list_off = (int32_t *)malloc((size + 1) * sizeof(int32_t));
  size_t total = 0;
  for (size_t i = 0; i < size; i++) { m->list_off[i] = (int32_t)total; total += env[i].clauses.length; }
  m->list_off[size] = (int32_t)total;
  m->list = (int32_t *)malloc((total ? total : 1) * sizeof(int32_t));
  for (size_t i = 0, k = 0; i < size; i++)
    for (size_t j = 0; j < env[i].clauses.length; j++) {
      int32_t c = pmap_get(&clause_ids, env[i].clauses.elems[j]);
      if (c < 0) { fprintf(stderr, "csolve_ref: clause of %s not found under root\n", env[i].key); exit(2); }
      m->list[k++] = c;
    }
  return m;
}

/* ---- random walks ---------------------------------------------------------- */

static uint64_t lcg_state;
static uint32_t lcg_next(void) {
  lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(lcg_state >> 33);
}

/* file layout (int32 LE): magic 'CSWK', version 1, n_vars, count, then per instance:
 *   var, value, status (-1 fail, else PROPS of this node), before[n]{lo,hi}, after[n]{lo,hi} */
static int cmd_walk(const char *path, uint64_t seed, long count, const char *out) {
  size_t size;
  struct env_t *env;
  struct constr_t *norm = root_phase(path, &size, &env);
  if (norm == NULL) return 1;
  FILE *f = fopen(out, "wb");
  if (f == NULL) { fprintf(stderr, "csolve_ref: %s: %s\n", out, strerror(errno)); return 2; }
  int32_t hdr[4] = { 0x4b575343, 1, (int32_t)size, (int32_t)count };
  fwrite(hdr, 4, 4, f);
  lcg_state = seed;
  int32_t *buf = (int32_t *)malloc(2 * size * sizeof(int32_t));
  size_t *open = (size_t *)malloc((size ? size : 1) * sizeof(size_t));
  void *marker = alloc(0);
  long done = 0, fails = 0;
  bind_level_set(0);
  while (done < count) {
    size_t n_open = 0;
    for (size_t i = 0; i < size; i++)
      if (!is_value(env[i].val->constr.term.val)) open[n_open++] = i;
    if (n_open == 0) { /* walk complete: start over from the root state */
      if (bind_depth() == 0) break; /* nothing is open even at the root */
      unbind(0); unpatch(0); dealloc(marker);
      continue;
    }
    struct env_t *var = &env[open[lcg_next() % n_open]];
    struct val_t d = var->val->constr.term.val;
    int32_t value = d.lo + (int32_t)(lcg_next() % (uint32_t)(d.hi - d.lo + 1));
    for (size_t i = 0; i < size; i++) {
      buf[2 * i] = env[i].val->constr.term.val.lo;
      buf[2 * i + 1] = env[i].val->constr.term.val.hi;
    }
    uint64_t props_before = stat_get_props();
    bind(var, VALUE(value), NULL);
    prop_result_t p = propagate_clauses(&var->clauses);
    int32_t rec[3] = { (int32_t)(var - env), value, p == PROP_ERROR ? -1 : (int32_t)(stat_get_props() - props_before) };
    fwrite(rec, 4, 3, f);
    fwrite(buf, 4, 2 * size, f);
    for (size_t i = 0; i < size; i++) {
      buf[2 * i] = env[i].val->constr.term.val.lo;
      buf[2 * i + 1] = env[i].val->constr.term.val.hi;
    }
    fwrite(buf, 4, 2 * size, f);
    done++;
    if (p == PROP_ERROR) { /* failed: next walk starts from the root state */
      fails++;
      unbind(0); unpatch(0); dealloc(marker);
    }
  }
  fclose(f);
  printf("@WALK {\"vars\": %zu, \"instances\": %ld, \"fails\": %ld}\n", size, done, fails);
  return 0;
}

/* ---- timed replay of given instances (the CPU baseline of bench.py) ------------ */

#include <time.h>

/* in : int32 LE: magic 'CSIN', n_vars, count, then per instance: var, value, before[n]{lo,hi}
 * out: int32 LE: magic 'CSOU', n_vars, count, then per instance: status (-1 | PROPS), after[n]{lo,hi}
 * prints "@BENCH {json}" with the time spent inside bind + propagate_clauses only */
static int cmd_bench(const char *path, const char *in, const char *out) {
  size_t size;
  struct env_t *env;
  struct constr_t *norm = root_phase(path, &size, &env);
  if (norm == NULL) return 1;
  FILE *f = fopen(in, "rb");
  if (f == NULL) { fprintf(stderr, "csolve_ref: %s: %s\n", in, strerror(errno)); return 2; }
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3 || hdr[0] != 0x4e495343 || (size_t)hdr[1] != size) {
    fprintf(stderr, "csolve_ref: %s: bad instance file\n", in);
    return 2;
  }
  const size_t count = (size_t)hdr[2], rec = 2 + 2 * size;
  int32_t *inst = (int32_t *)malloc(count * rec * sizeof(int32_t));
  if (fread(inst, sizeof(int32_t), count * rec, f) != count * rec) { fprintf(stderr, "csolve_ref: short read\n"); return 2; }
  fclose(f);
  int32_t *res = (int32_t *)malloc(count * (1 + 2 * size) * sizeof(int32_t));
  void *marker = alloc(0);
  bind_level_set(0);
  uint64_t binds = 0, fails = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (size_t i = 0; i < count; i++) {
    const int32_t *r = &inst[i * rec];
    for (size_t v = 0; v < size; v++) {
      env[v].val->constr.term.val.lo = r[2 + 2 * v];
      env[v].val->constr.term.val.hi = r[3 + 2 * v];
    }
    struct env_t *var = &env[r[0]];
    uint64_t before = stat_get_props();
    bind(var, VALUE(r[1]), NULL);
    prop_result_t p = propagate_clauses(&var->clauses);
    int32_t *o = &res[i * (1 + 2 * size)];
    o[0] = p == PROP_ERROR ? -1 : (int32_t)(stat_get_props() - before);
    binds += stat_get_props() - before;
    fails += p == PROP_ERROR;
    for (size_t v = 0; v < size; v++) {
      o[1 + 2 * v] = env[v].val->constr.term.val.lo;
      o[2 + 2 * v] = env[v].val->constr.term.val.hi;
    }
    unbind(0); unpatch(0); dealloc(marker);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  f = fopen(out, "wb");
  if (f == NULL) { fprintf(stderr, "csolve_ref: %s: %s\n", out, strerror(errno)); return 2; }
  int32_t ohdr[3] = { 0x554f5343, (int32_t)size, (int32_t)count };
  fwrite(ohdr, 4, 3, f);
  fwrite(res, sizeof(int32_t), count * (1 + 2 * size), f);
  fclose(f);
  printf("@BENCH {\"instances\": %zu, \"seconds\": %.6f, \"binds\": %lu, \"fails\": %lu}\n", count, secs, binds, fails);
  return 0;
}

/* ---- commands --------------------------------------------------------------- */

static bool parse_bool(const char *s) { return strcmp(s, "true") == 0 || strcmp(s, "1") == 0; }

static enum order_t parse_order(const char *s) {
  if (strcmp(s, "smallest-domain") == 0) return ORDER_SMALLEST_DOMAIN;
  if (strcmp(s, "largest-domain") == 0) return ORDER_LARGEST_DOMAIN;
  if (strcmp(s, "smallest-value") == 0) return ORDER_SMALLEST_VALUE;
  if (strcmp(s, "largest-value") == 0) return ORDER_LARGEST_VALUE;
  return ORDER_NONE;
}

#ifdef CS_DROPIN_BUILD
/* The driver draws its restart seeds from rand() (csolve.c:284), unseeded.  In a process that has loaded the HIP
 * runtime other threads draw from the same libc generator, so the driver's sequence would differ from run to
 * run.  The driver objects of THIS build therefore bind to a private generator that reproduces glibc's default
 * sequence (TYPE_3 additive feedback, srand(1)); it is hidden, the runtime keeps using libc's. */
static int cs_rand_f;
__attribute__((visibility("hidden"))) int rand(void) {
  static int32_t r[34];
  static int ready;
  if (!ready) {
    int32_t t[344];
    t[0] = 1;
    for (int i = 1; i < 31; i++) {
      int64_t v = (16807LL * t[i - 1]) % 2147483647;
      if (v < 0) v += 2147483647;
      t[i] = (int32_t)v;
    }
    for (int i = 31; i < 34; i++) t[i] = t[i - 31];
    for (int i = 34; i < 344; i++) t[i] = (int32_t)((uint32_t)t[i - 31] + (uint32_t)t[i - 3]);
    for (int i = 0; i < 34; i++) r[i] = t[310 + i];
    ready = 1;
    cs_rand_f = 0;
  }
  /* r holds the last 34 outputs of the recurrence x[k] = x[k-31] + x[k-3] as a ring */
  const int k = cs_rand_f;
  const uint32_t v = (uint32_t)r[(k + 34 - 31) % 34] + (uint32_t)r[(k + 34 - 3) % 34];
  r[k] = (int32_t)v;
  cs_rand_f = (k + 1) % 34;
  return (int)(v >> 1);
}
#endif

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: csolve_ref solve|model|walk <file> ...\n");
    return 2;
  }
  struct options o = { STRATEGY_CREATE_CONFLICTS_DEFAULT, STRATEGY_PREFER_FAILING_DEFAULT,
                       STRATEGY_COMPUTE_WEIGHTS_DEFAULT, STRATEGY_RESTART_FREQUENCY_DEFAULT,
                       STRATEGY_ORDER_DEFAULT, TIME_MAX_DEFAULT };
  const char *cmd = argv[1], *path = argv[2];
  int first_opt = strcmp(cmd, "solve") == 0 ? 3 : (strcmp(cmd, "model") == 0 || strcmp(cmd, "rootlimit") == 0 ? 4 : (strcmp(cmd, "bench") == 0 ? 5 : 6));
  for (int i = first_opt; i + 1 < argc; i += 2) {
    if (strcmp(argv[i], "-c") == 0) o.conflicts = parse_bool(argv[i + 1]);
    else if (strcmp(argv[i], "-f") == 0) o.prefer_failing = parse_bool(argv[i + 1]);
    else if (strcmp(argv[i], "-w") == 0) o.weighten = parse_bool(argv[i + 1]);
    else if (strcmp(argv[i], "-r") == 0) o.restart_freq = strtoull(argv[i + 1], NULL, 10);
    else if (strcmp(argv[i], "-o") == 0) o.order = parse_order(argv[i + 1]);
    else if (strcmp(argv[i], "-t") == 0) o.time_max = (uint32_t)strtoul(argv[i + 1], NULL, 10);
    else { fprintf(stderr, "csolve_ref: unknown option %s\n", argv[i]); return 2; }
  }
  reference_init(&o);

  if (strcmp(cmd, "bench") == 0) {
    if (argc < 5) { fprintf(stderr, "usage: csolve_ref bench <file> <instances.in> <results.out>\n"); return 2; }
    return cmd_bench(path, argv[3], argv[4]);
  }

  if (strcmp(cmd, "walk") == 0) {
    if (argc < 6) { fprintf(stderr, "usage: csolve_ref walk <file> <seed> <count> <out>\n"); return 2; }
    return cmd_walk(path, strtoull(argv[3], NULL, 10), strtol(argv[4], NULL, 10), argv[5]);
  }

  if (strcmp(cmd, "rootlimit") == 0) {
    /* csolve_ref rootlimit <file> <limit>: the front end, then ONE call propagate(root, limit) (propagate.c:474-485) on
     * the raw root -- no normalisation -- and the variables' domains after it (golden vectors of the limit's semantics) */
    if (argc < 4) { fprintf(stderr, "usage: csolve_ref rootlimit <file> <limit>\n"); return 2; }
    char *text = slurp(path);
    char err[200];
    cs_builder b = { NULL, r_num, r_ident, r_unary, r_binary, r_wand, r_weigh, r_objective, r_constraint };
    if (cs_parse_text(text, &b, err, sizeof err) != 0) { fprintf(stderr, "csolve_ref: error: %s\n", err); return 1; }
    free(text);
    const size_t nv = var_count();
    const prop_result_t p = propagate(root_wand, (size_t)strtoull(argv[3], NULL, 10));
    struct env_t *ev = env_generate();
    printf("@ROOT {\"status\": %d, \"domains\": {", p == PROP_ERROR ? -1 : (int)p);
    for (size_t v = 0; v < nv; v++)
      printf("%s\"%s\": [%d, %d]", v ? ", " : "", ev[v].key, ev[v].val->constr.term.val.lo, ev[v].val->constr.term.val.hi);
    printf("}}\n");
    return 0;
  }

  size_t size;
  struct env_t *env;
  struct timespec ts0, ts1, ts2;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
  struct constr_t *norm = root_phase(path, &size, &env);
  clock_gettime(CLOCK_MONOTONIC, &ts1);

  if (strcmp(cmd, "model") == 0) {
    if (argc < 4 || norm == NULL) return 1;
    cs_model *m = dump_model(norm, size, env);
    if (cs_model_save(m, argv[3]) != 0) { fprintf(stderr, "csolve_ref: cannot write %s\n", argv[3]); return 2; }
    printf("@MODEL {\"vars\": %d, \"nodes\": %d, \"clauses\": %d, \"list_total\": %d}\n",
           m->n_vars, m->n_nodes, m->n_clauses, m->list_off[m->n_vars]);
    return 0;
  }

  if (strcmp(cmd, "solve") == 0) {
    if (norm != NULL) solve(size, env, norm); /* second half of the Input action (parser.y:86) */
    clock_gettime(CLOCK_MONOTONIC, &ts2);
    fflush(stdout);
#ifdef CS_DROPIN_BUILD
    {
      /* this build links the reference DRIVER against libcsolve_dropin.so: report how often the
       * driver went through the GPU entry points */
      extern void csolve_dropin_counters(uint64_t out[4]);
      uint64_t k[4];
      csolve_dropin_counters(k);
      extern void csolve_dropin_sibling_counters(uint64_t out[2]);
      uint64_t sb[2];
      csolve_dropin_sibling_counters(sb);
      extern void csolve_dropin_seconds(double out[3]);
      double sec[3];
      csolve_dropin_seconds(sec);
      extern void csolve_dropin_call_times(double out[4]);
      double ct[4];
      csolve_dropin_call_times(ct);
      extern void csolve_dropin_learning_counters(uint64_t out[2]);
      uint64_t lc[2];
      csolve_dropin_learning_counters(lc);
      extern double csolve_dropin_eval_seconds(void);
      printf("@DROPIN_EVAL {\"eval_seconds\": %.6f}\n", csolve_dropin_eval_seconds());
      printf("@DROPIN {\"propagate_clauses\": %lu, \"propagate\": %lu, \"eval\": %lu, \"single_op\": %lu, "
             "\"sibling_batches\": %lu, \"served_from_batch\": %lu, \"conflicts_offered\": %lu, \"reattached\": %lu, "
             "\"attach_seconds\": %.6f, \"device_call_seconds\": %.6f, \"shim_host_seconds\": %.6f, "
             "\"call_us_first\": %.1f, \"call_us_median\": %.1f, \"call_us_p90\": %.1f, \"call_us_max\": %.1f}\n", k[0], k[1], k[2],
             k[3], sb[0], sb[1], lc[0], lc[1], sec[0], sec[1], sec[2], ct[0], ct[1], ct[2], ct[3]);
    }
#endif
    printf("@STATS {\"feasible_root\": %s, \"calls\": %lu, \"cuts\": %lu, \"props\": %lu, \"confl\": %lu, "
           "\"restarts\": %lu, \"solutions\": %lu, \"best\": %d, \"timeout\": %s, \"root_seconds\": %.6f, "
           "\"solve_seconds\": %.6f}\n",
           norm != NULL ? "true" : "false", stat_get_calls(), stat_get_cuts(), stat_get_props(),
           stat_get_confl(), stat_get_restarts(), shared()->solutions, objective_best(),
           shared()->timeout ? "true" : "false",
           (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec),
           (double)(ts2.tv_sec - ts1.tv_sec) + 1e-9 * (double)(ts2.tv_nsec - ts1.tv_nsec));
    return 0;
  }
  fprintf(stderr, "csolve_ref: unknown command %s\n", cmd);
  return 2;
}
