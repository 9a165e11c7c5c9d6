/* cs_model.c -- host-side problem model: construction from text, clause
 * indexing, golden-model file I/O.  See cs_model.h for the reference mapping. */
#include "cs_model.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CS_MODEL_MAGIC 0x4d445343 /* "CSDM" */
#define CS_MODEL_VERSION 1

static void *xrealloc(void *p, size_t n) {
  void *q = realloc(p, n ? n : 1);
  if (q == NULL) {
    fprintf(stderr, "csolve_amd: error: out of memory\n");
    exit(EXIT_FAILURE);
  }
  return q;
}

#define GROW(ptr, cap, need)                                                   \
  do {                                                                         \
    if ((need) > (cap)) {                                                      \
      (cap) = (cap) ? (cap)*2 : 64;                                            \
      while ((cap) < (need)) (cap) *= 2;                                       \
      (ptr) = xrealloc((ptr), (size_t)(cap) * sizeof *(ptr));                  \
    }                                                                          \
  } while (0)

cs_model *cs_model_new(void) {
  cs_model *m = (cs_model *)calloc(1, sizeof *m);
  if (m == NULL) return NULL;
  m->root = -1;
  m->obj_var = -1;
  m->objective = CS_OBJ_ANY;
  m->weights_on = 1;
  return m;
}

void cs_model_free(cs_model *m) {
  if (m == NULL) return;
  for (int32_t i = 0; i < m->n_vars; i++) free(m->names[i]);
  free(m->dom); free(m->names); free(m->prio); free(m->var_node);
  free(m->nodes); free(m->kids); free(m->top);
  free(m->clause_node); free(m->list_off); free(m->list); free(m->clause_want);
  free(m->name_tab);
  free(m);
}

/* a deep copy WITHOUT the clause index (cs_model_index rebuilds it): the trees, domains, names and weights of `src`,
 * for a model that is to be rewritten on its own (csgpu_model_specialize) */
static void *copy_of(const void *p, size_t bytes, size_t alloc_bytes) {
  void *q = xrealloc(NULL, alloc_bytes > bytes ? alloc_bytes : (bytes ? bytes : 1));
  if (bytes && p != NULL) memcpy(q, p, bytes);
  return q;
}

cs_model *cs_model_clone(const cs_model *src) {
  cs_model *m = cs_model_new();
  const int32_t nv = src->n_vars;
  m->n_vars = nv;
  m->cap_vars = nv ? nv : 1;
  m->dom = (cs_val *)xrealloc(NULL, (size_t)m->cap_vars * sizeof(cs_val));
  m->prio = (int64_t *)xrealloc(NULL, (size_t)m->cap_vars * sizeof(int64_t));
  m->var_node = (int32_t *)xrealloc(NULL, (size_t)m->cap_vars * sizeof(int32_t));
  m->names = (char **)calloc((size_t)m->cap_vars, sizeof(char *));
  memcpy(m->dom, src->dom, (size_t)nv * sizeof(cs_val));
  memcpy(m->prio, src->prio, (size_t)nv * sizeof(int64_t));
  memcpy(m->var_node, src->var_node, (size_t)nv * sizeof(int32_t));
  for (int32_t v = 0; v < nv; v++) m->names[v] = (char *)copy_of(src->names[v], strlen(src->names[v]) + 1, 0);
  m->n_nodes = src->n_nodes;
  m->cap_nodes = src->n_nodes > 4 ? src->n_nodes : 4;
  free(m->nodes);
  m->nodes = (cs_node *)copy_of(src->nodes, (size_t)src->n_nodes * sizeof(cs_node), (size_t)m->cap_nodes * sizeof(cs_node));
  m->n_kids = src->n_kids;
  m->cap_kids = src->n_kids > 4 ? src->n_kids : 4;
  free(m->kids);
  m->kids = (int32_t *)copy_of(src->kids, (size_t)src->n_kids * sizeof(int32_t), (size_t)m->cap_kids * sizeof(int32_t));
  m->root = src->root;
  m->n_top = src->n_top;
  m->cap_top = src->n_top > 4 ? src->n_top : 4;
  free(m->top);
  m->top = (int32_t *)copy_of(src->top, (size_t)src->n_top * sizeof(int32_t), (size_t)m->cap_top * sizeof(int32_t));
  m->objective = src->objective;
  m->obj_var = src->obj_var;
  m->weights_on = src->weights_on;
  if (src->name_tab != NULL && src->name_cap > 0) {
    free(m->name_tab);
    m->name_cap = src->name_cap;
    m->name_tab = (int32_t *)copy_of(src->name_tab, (size_t)src->name_cap * sizeof(int32_t), 0);
  }
  return m;
}

/* ---- name table ---------------------------------------------------------- */

static uint32_t name_hash(const char *s) {
  uint32_t h = 2166136261u;
  for (; *s; s++) h = (h ^ (unsigned char)*s) * 16777619u;
  return h;
}

static void name_tab_insert(cs_model *m, int32_t var) {
  uint32_t mask = (uint32_t)m->name_cap - 1;
  uint32_t i = name_hash(m->names[var]) & mask;
  while (m->name_tab[i] >= 0) i = (i + 1) & mask;
  m->name_tab[i] = var;
}

static void name_tab_grow(cs_model *m) {
  int32_t cap = m->name_cap ? m->name_cap * 2 : 256;
  m->name_tab = (int32_t *)xrealloc(m->name_tab, (size_t)cap * sizeof(int32_t));
  m->name_cap = cap;
  for (int32_t i = 0; i < cap; i++) m->name_tab[i] = -1;
  for (int32_t v = 0; v < m->n_vars; v++) name_tab_insert(m, v);
}

int32_t cs_model_find_var(const cs_model *m, const char *name) {
  if (m->name_cap == 0) return -1;
  uint32_t mask = (uint32_t)m->name_cap - 1;
  uint32_t i = name_hash(name) & mask;
  while (m->name_tab[i] >= 0) {
    if (strcmp(m->names[m->name_tab[i]], name) == 0) return m->name_tab[i];
    i = (i + 1) & mask;
  }
  return -1;
}

/* ---- construction -------------------------------------------------------- */

int32_t cs_model_add_node(cs_model *m, int32_t op, int32_t a, int32_t b) {
  GROW(m->nodes, m->cap_nodes, m->n_nodes + 1);
  cs_node *n = &m->nodes[m->n_nodes];
  n->op = op; n->a = a; n->b = b;
  return m->n_nodes++;
}

int32_t cs_model_add_var(cs_model *m, const char *name, cs_val dom) {
  int32_t need = m->n_vars + 1;
  if (need > m->cap_vars) {
    int32_t cap = m->cap_vars ? m->cap_vars * 2 : 64;
    m->dom = (cs_val *)xrealloc(m->dom, (size_t)cap * sizeof *m->dom);
    m->names = (char **)xrealloc(m->names, (size_t)cap * sizeof *m->names);
    m->prio = (int64_t *)xrealloc(m->prio, (size_t)cap * sizeof *m->prio);
    m->var_node = (int32_t *)xrealloc(m->var_node, (size_t)cap * sizeof *m->var_node);
    m->cap_vars = cap;
  }
  int32_t v = m->n_vars;
  size_t len = strlen(name) + 1;
  m->names[v] = (char *)xrealloc(NULL, len);
  memcpy(m->names[v], name, len);
  m->dom[v] = dom;
  m->prio[v] = 0;
  m->n_vars++;
  m->var_node[v] = cs_model_add_node(m, CS_OP_VAR, v, -1);
  if ((m->n_vars + 1) * 2 > m->name_cap) name_tab_grow(m);
  else name_tab_insert(m, v);
  return v;
}

int32_t cs_model_add_wand(cs_model *m, const int32_t *elems, int32_t n) {
  GROW(m->kids, m->cap_kids, m->n_kids + n);
  int32_t off = m->n_kids;
  for (int32_t i = 0; i < n; i++) m->kids[off + i] = elems[i];
  m->n_kids += n;
  return cs_model_add_node(m, CS_OP_WAND, off, n);
}

int32_t cs_model_add_confl(cs_model *m, const int32_t *term_nodes, const int32_t *values, int32_t n) {
  GROW(m->kids, m->cap_kids, m->n_kids + 2 * n);
  int32_t off = m->n_kids;
  for (int32_t i = 0; i < n; i++) {
    m->kids[off + 2 * i] = term_nodes[i];
    m->kids[off + 2 * i + 1] = values[i];
  }
  m->n_kids += 2 * n;
  return cs_model_add_node(m, CS_OP_CONFL, off, n);
}

void cs_model_set_root_from_top(cs_model *m) {
  m->root = cs_model_add_wand(m, m->top, m->n_top);
}

/* ---- builder callbacks for cs_parse_text --------------------------------- */

#define H2N(h) ((int32_t)(intptr_t)(h)-1)
#define N2H(n) ((void *)(intptr_t)((n) + 1))

static void *bld_num(void *ctx, int32_t v) {
  return N2H(cs_model_add_node((cs_model *)ctx, CS_OP_CONST, v, v));
}

static void *bld_ident(void *ctx, const char *name) {
  cs_model *m = (cs_model *)ctx;
  int32_t v = cs_model_find_var(m, name);
  if (v < 0) v = cs_model_add_var(m, name, cs_interval(CS_DOM_MIN, CS_DOM_MAX));
  return N2H(m->var_node[v]);
}

static void *bld_unary(void *ctx, int op, void *c) {
  return N2H(cs_model_add_node((cs_model *)ctx, op, H2N(c), -1));
}

static void *bld_binary(void *ctx, int op, void *l, void *r) {
  return N2H(cs_model_add_node((cs_model *)ctx, op, H2N(l), H2N(r)));
}

static void *bld_wand(void *ctx, void **elems, size_t n) {
  cs_model *m = (cs_model *)ctx;
  int32_t *ids = (int32_t *)xrealloc(NULL, (n ? n : 1) * sizeof *ids);
  for (size_t i = 0; i < n; i++) ids[i] = H2N(elems[i]);
  int32_t w = cs_model_add_wand(m, ids, (int32_t)n);
  free(ids);
  return N2H(w);
}

/* vars_count / vars_weighten (reference src/parser_support.c:182-242): every
 * occurrence of a non-value terminal counts; a wide-and below a weighted
 * operator is a fatal "invalid operation" in the reference. */
static int32_t occurrences(cs_model *m, int32_t node, int64_t add) {
  const cs_node *n = &m->nodes[node];
  switch (n->op) {
  case CS_OP_VAR:
    if (!cs_is_value(m->dom[n->a])) { m->prio[n->a] += add; return 1; }
    return 0;
  case CS_OP_CONST:
    return 0;
  case CS_OP_NEG: case CS_OP_NOT:
    return occurrences(m, n->a, add);
  case CS_OP_EQ: case CS_OP_LT: case CS_OP_ADD: case CS_OP_MUL: case CS_OP_AND: case CS_OP_OR: {
    int32_t r = occurrences(m, n->b, add);
    int32_t l = r < 0 ? -1 : occurrences(m, n->a, add);
    return (r < 0 || l < 0) ? -1 : l + r;
  }
  default:
    snprintf(m->err, sizeof m->err, "invalid operation: %02x", 'A');
    return -1;
  }
}

static void bld_weigh(void *ctx, void *expr, int32_t weight) {
  cs_model *m = (cs_model *)ctx;
  if (!m->weights_on) return;
  int32_t cnt = occurrences(m, H2N(expr), 0);
  if (cnt < 0) return;
  occurrences(m, H2N(expr), weight / (cnt > 1 ? cnt : 1));
}

static void *bld_objective(void *ctx, int kind, void *expr) {
  cs_model *m = (cs_model *)ctx;
  m->objective = kind;
  if (kind == CS_OBJ_ANY || kind == CS_OBJ_ALL)
    return N2H(cs_model_add_node(m, CS_OP_CONST, 1, 1));
  /* "<obj>" is bounded away from the sentinels (reference src/objective.c:37) */
  m->obj_var = cs_model_add_var(m, "<obj>", cs_interval(CS_DOM_MIN + 1, CS_DOM_MAX - 1));
  int32_t o = m->var_node[m->obj_var], e = H2N(expr);
  return N2H(kind == CS_OBJ_MIN ? cs_model_add_node(m, CS_OP_EQ, e, o)
                                : cs_model_add_node(m, CS_OP_EQ, o, e));
}

static void bld_constraint(void *ctx, void *expr) {
  cs_model *m = (cs_model *)ctx;
  GROW(m->top, m->cap_top, m->n_top + 1);
  m->top[m->n_top++] = H2N(expr);
}

cs_model *cs_model_parse(const char *text, int weights_on, char *err, size_t errlen) {
  cs_model *m = cs_model_new();
  if (m == NULL) return NULL;
  m->weights_on = weights_on;
  cs_builder b = { m, bld_num, bld_ident, bld_unary, bld_binary, bld_wand,
                   bld_weigh, bld_objective, bld_constraint };
  if (cs_parse_text(text, &b, err, errlen) != 0) {
    cs_model_free(m);
    return NULL;
  }
  if (m->err[0] != '\0') {
    if (err != NULL && errlen > 0) snprintf(err, errlen, "%s", m->err);
    cs_model_free(m);
    return NULL;
  }
  cs_model_set_root_from_top(m);
  return m;
}

/* ---- clause index (clauses_init) ----------------------------------------- */

typedef struct {
  int32_t *v;
  int32_t n, cap;
} ivec;

typedef struct {
  cs_model *m;
  ivec *lists;
  ivec clause_node;
  int bad;
} index_ctx;

static void ivec_push(ivec *x, int32_t e) {
  GROW(x->v, x->cap, x->n + 1);
  x->v[x->n++] = e;
}

static void index_node(index_ctx *c, int32_t node, int32_t clause) {
  const cs_node *n = &c->m->nodes[node];
  switch (n->op) {
  case CS_OP_VAR:
    if (clause >= 0 && !cs_is_value(c->m->dom[n->a])) {
      ivec *l = &c->lists[n->a];
      /* clause ids only grow, so "already contained" means "is the last entry" */
      if (l->n == 0 || l->v[l->n - 1] != clause) ivec_push(l, clause);
    }
    break;
  case CS_OP_CONST:
    break;
  case CS_OP_WAND:
    for (int32_t i = 0; i < n->b; i++) {
      int32_t kid = c->m->kids[n->a + i];
      int32_t cl = clause;
      if (clause < 0 && c->m->nodes[kid].op != CS_OP_WAND) {
        cl = c->clause_node.n;
        ivec_push(&c->clause_node, kid);
      }
      index_node(c, kid, cl);
      n = &c->m->nodes[node];
    }
    break;
  case CS_OP_CONFL: /* conflict_create appends the clause to the list of every element's variable, conflict.c:356-358 */
    for (int32_t i = 0; i < n->b; i++) {
      index_node(c, c->m->kids[n->a + 2 * i], clause);
      n = &c->m->nodes[node];
    }
    break;
  case CS_OP_NEG: case CS_OP_NOT:
    index_node(c, n->a, clause);
    break;
  case CS_OP_EQ: case CS_OP_LT: case CS_OP_ADD: case CS_OP_MUL: case CS_OP_AND: case CS_OP_OR:
    index_node(c, n->b, clause);
    index_node(c, c->m->nodes[node].a, clause);
    break;
  default:
    c->bad = 1;
    break;
  }
}

int cs_model_index(cs_model *m) {
  if (m->root < 0) return -1;
  index_ctx c;
  memset(&c, 0, sizeof c);
  c.m = m;
  c.lists = (ivec *)calloc((size_t)(m->n_vars ? m->n_vars : 1), sizeof(ivec));
  index_node(&c, m->root, -1);

  free(m->clause_node); free(m->list_off); free(m->list);
  m->n_clauses = c.clause_node.n;
  m->clause_node = c.clause_node.v ? c.clause_node.v : (int32_t *)xrealloc(NULL, sizeof(int32_t));
  m->list_off = (int32_t *)xrealloc(NULL, (size_t)(m->n_vars + 1) * sizeof(int32_t));
  int32_t total = 0;
  for (int32_t v = 0; v < m->n_vars; v++) { m->list_off[v] = total; total += c.lists[v].n; }
  m->list_off[m->n_vars] = total;
  m->list = (int32_t *)xrealloc(NULL, (size_t)(total ? total : 1) * sizeof(int32_t));
  for (int32_t v = 0; v < m->n_vars; v++) {
    if (c.lists[v].n) memcpy(&m->list[m->list_off[v]], c.lists[v].v, (size_t)c.lists[v].n * sizeof(int32_t));
    free(c.lists[v].v);
  }
  free(c.lists);
  if (c.bad) {
    snprintf(m->err, sizeof m->err, "invalid operation in clause index");
    return -1;
  }
  return 0;
}

/* One more top-level clause (a learnt conflict): a new root wide-and with `node` as its last element and, if the
 * model carries a clause index, the clause appended to the lists of its variables -- where conflict_create
 * puts it (conflict.c:352-358: clause_list_append for every element). */
int cs_model_append_clause(cs_model *m, int32_t node) {
  if (m->root < 0) return -1;
  const int32_t cnt = m->nodes[m->root].b, off = m->nodes[m->root].a;
  int32_t *elems = (int32_t *)xrealloc(NULL, (size_t)(cnt + 1) * sizeof(int32_t));
  for (int32_t i = 0; i < cnt; i++) elems[i] = m->kids[off + i];
  elems[cnt] = node;
  m->root = cs_model_add_wand(m, elems, cnt + 1);
  free(elems);
  if (m->clause_node == NULL) return 0;
  index_ctx c;
  memset(&c, 0, sizeof c);
  c.m = m;
  c.lists = (ivec *)calloc((size_t)(m->n_vars ? m->n_vars : 1), sizeof(ivec));
  const int32_t clause = m->n_clauses;
  index_node(&c, node, clause);
  m->clause_node = (int32_t *)xrealloc(m->clause_node, (size_t)(clause + 1) * sizeof(int32_t));
  m->clause_node[clause] = node;
  m->n_clauses = clause + 1;
  if (m->clause_want != NULL) {
    m->clause_want = (cs_val *)xrealloc(m->clause_want, (size_t)(clause + 1) * sizeof(cs_val));
    m->clause_want[clause] = cs_interval(1, 1);
  }
  if (m->list_off != NULL) {
    int32_t extra = 0;
    for (int32_t v = 0; v < m->n_vars; v++) extra += c.lists[v].n;
    const int32_t total = m->list_off[m->n_vars];
    int32_t *list = (int32_t *)xrealloc(NULL, (size_t)(total + extra ? total + extra : 1) * sizeof(int32_t));
    int32_t *loff = (int32_t *)xrealloc(NULL, (size_t)(m->n_vars + 1) * sizeof(int32_t));
    int32_t w = 0;
    for (int32_t v = 0; v < m->n_vars; v++) {
      loff[v] = w;
      for (int32_t i = m->list_off[v]; i < m->list_off[v + 1]; i++) list[w++] = m->list[i];
      if (c.lists[v].n > 0) list[w++] = clause;
    }
    loff[m->n_vars] = w;
    free(m->list); free(m->list_off);
    m->list = list;
    m->list_off = loff;
  }
  for (int32_t v = 0; v < m->n_vars; v++) free(c.lists[v].v);
  free(c.lists);
  return c.bad ? -1 : 0;
}

int32_t cs_model_first_unbounded(const cs_model *m) {
  for (int32_t v = 0; v < m->n_vars; v++)
    if (m->dom[v].lo == CS_DOM_MIN || m->dom[v].hi == CS_DOM_MAX) return v;
  return -1;
}

int32_t cs_model_tree_size(const cs_model *m, int32_t node) {
  const cs_node *n = &m->nodes[node];
  switch (n->op) {
  case CS_OP_VAR: case CS_OP_CONST: return 1;
  case CS_OP_NEG: case CS_OP_NOT: return 1 + cs_model_tree_size(m, n->a);
  case CS_OP_CONFL: return 1 + n->b;
  case CS_OP_WAND: {
    int32_t s = 1;
    for (int32_t i = 0; i < n->b; i++) s += cs_model_tree_size(m, m->kids[n->a + i]);
    return s;
  }
  default: return 1 + cs_model_tree_size(m, n->a) + cs_model_tree_size(m, n->b);
  }
}

/* ---- golden-model files ---------------------------------------------------
 * int32 little-endian stream:
 *   magic, version, n_vars, n_nodes, n_kids, root, objective, obj_var,
 *   n_clauses (-1: no clause index stored), n_list
 *   dom[n_vars]{lo,hi}, prio[n_vars]{lo32,hi32}, nodes[n_nodes]{op,a,b}, kids[n_kids],
 *   clause_node[n_clauses], list_off[n_vars+1], list[n_list]   (if n_clauses >= 0)
 *   names: per variable { len, bytes padded to a multiple of 4 }
 */
static int put(FILE *f, const void *p, size_t n) { return fwrite(p, 1, n, f) == n ? 0 : -1; }
static int put32(FILE *f, int32_t v) { return put(f, &v, 4); }

int cs_model_save(const cs_model *m, const char *path) {
  FILE *f = fopen(path, "wb");
  if (f == NULL) return -1;
  int has_index = m->clause_node != NULL && m->list_off != NULL;
  int32_t n_list = has_index ? m->list_off[m->n_vars] : 0;
  int rc = 0;
  rc |= put32(f, CS_MODEL_MAGIC); rc |= put32(f, CS_MODEL_VERSION);
  rc |= put32(f, m->n_vars); rc |= put32(f, m->n_nodes); rc |= put32(f, m->n_kids);
  rc |= put32(f, m->root); rc |= put32(f, m->objective); rc |= put32(f, m->obj_var);
  rc |= put32(f, has_index ? m->n_clauses : -1); rc |= put32(f, n_list);
  rc |= put(f, m->dom, (size_t)m->n_vars * sizeof(cs_val));
  rc |= put(f, m->prio, (size_t)m->n_vars * sizeof(int64_t));
  rc |= put(f, m->nodes, (size_t)m->n_nodes * sizeof(cs_node));
  rc |= put(f, m->kids, (size_t)m->n_kids * sizeof(int32_t));
  if (has_index) {
    rc |= put(f, m->clause_node, (size_t)m->n_clauses * sizeof(int32_t));
    rc |= put(f, m->list_off, (size_t)(m->n_vars + 1) * sizeof(int32_t));
    rc |= put(f, m->list, (size_t)n_list * sizeof(int32_t));
  }
  for (int32_t v = 0; v < m->n_vars; v++) {
    int32_t len = (int32_t)strlen(m->names[v]);
    char pad[4] = { 0, 0, 0, 0 };
    rc |= put32(f, len);
    rc |= put(f, m->names[v], (size_t)len);
    rc |= put(f, pad, (size_t)((4 - len % 4) % 4));
  }
  if (fclose(f) != 0) rc = -1;
  return rc;
}

static int get(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

cs_model *cs_model_load(const char *path, char *err, size_t errlen) {
  FILE *f = fopen(path, "rb");
  if (f == NULL) {
    if (err) snprintf(err, errlen, "%s: cannot open", path);
    return NULL;
  }
  int32_t h[10];
  cs_model *m = NULL;
  if (get(f, h, sizeof h) != 0 || h[0] != CS_MODEL_MAGIC || h[1] != CS_MODEL_VERSION) goto bad;
  if (h[2] < 0 || h[3] < 0 || h[4] < 0 || h[9] < 0) goto bad;
  m = cs_model_new();
  int32_t nv = h[2];
  m->cap_vars = nv ? nv : 1;
  m->dom = (cs_val *)xrealloc(NULL, (size_t)m->cap_vars * sizeof(cs_val));
  m->names = (char **)calloc((size_t)m->cap_vars, sizeof(char *));
  m->prio = (int64_t *)xrealloc(NULL, (size_t)m->cap_vars * sizeof(int64_t));
  m->var_node = (int32_t *)xrealloc(NULL, (size_t)m->cap_vars * sizeof(int32_t));
  m->n_nodes = m->cap_nodes = h[3];
  m->nodes = (cs_node *)xrealloc(NULL, (size_t)h[3] * sizeof(cs_node));
  m->n_kids = m->cap_kids = h[4];
  m->kids = (int32_t *)xrealloc(NULL, (size_t)h[4] * sizeof(int32_t));
  m->root = h[5]; m->objective = h[6]; m->obj_var = h[7];
  if (get(f, m->dom, (size_t)nv * sizeof(cs_val)) || get(f, m->prio, (size_t)nv * sizeof(int64_t)) ||
      get(f, m->nodes, (size_t)h[3] * sizeof(cs_node)) || get(f, m->kids, (size_t)h[4] * sizeof(int32_t)))
    goto bad;
  if (h[8] >= 0) {
    m->n_clauses = h[8];
    m->clause_node = (int32_t *)xrealloc(NULL, (size_t)h[8] * sizeof(int32_t));
    m->list_off = (int32_t *)xrealloc(NULL, (size_t)(nv + 1) * sizeof(int32_t));
    m->list = (int32_t *)xrealloc(NULL, (size_t)h[9] * sizeof(int32_t));
    if (get(f, m->clause_node, (size_t)h[8] * sizeof(int32_t)) ||
        get(f, m->list_off, (size_t)(nv + 1) * sizeof(int32_t)) ||
        get(f, m->list, (size_t)h[9] * sizeof(int32_t)))
      goto bad;
  }
  for (int32_t v = 0; v < nv; v++) {
    int32_t len;
    if (get(f, &len, 4) || len < 0 || len > 4096) goto bad;
    int32_t padded = len + (4 - len % 4) % 4;
    char *s = (char *)xrealloc(NULL, (size_t)padded + 1);
    if (get(f, s, (size_t)padded)) { free(s); goto bad; }
    s[len] = '\0';
    m->names[v] = s;
    m->var_node[v] = -1;
    m->n_vars = v + 1;
  }
  for (int32_t i = 0; i < m->n_nodes; i++)
    if (m->nodes[i].op == CS_OP_VAR && m->nodes[i].a >= 0 && m->nodes[i].a < nv && m->var_node[m->nodes[i].a] < 0)
      m->var_node[m->nodes[i].a] = i;
  name_tab_grow(m);
  fclose(f);
  return m;
bad:
  if (err) snprintf(err, errlen, "%s: not a csolve model file", path);
  if (m) cs_model_free(m);
  fclose(f);
  return NULL;
}

/* ---- comparison ---------------------------------------------------------- */

/* compare the trees under two nodes structurally (node numbering may differ) */
static int tree_equal(const cs_model *x, int32_t nx, const cs_model *y, int32_t ny) {
  const cs_node *a = &x->nodes[nx], *b = &y->nodes[ny];
  if (a->op != b->op) return 0;
  switch (a->op) {
  case CS_OP_VAR: return a->a == b->a;
  case CS_OP_CONST: return a->a == b->a && a->b == b->b;
  case CS_OP_NEG: case CS_OP_NOT: return tree_equal(x, a->a, y, b->a);
  case CS_OP_WAND:
    if (a->b != b->b) return 0;
    for (int32_t i = 0; i < a->b; i++)
      if (!tree_equal(x, x->kids[a->a + i], y, y->kids[b->a + i])) return 0;
    return 1;
  case CS_OP_CONFL:
    if (a->b != b->b) return 0;
    for (int32_t i = 0; i < a->b; i++)
      if (x->kids[a->a + 2 * i + 1] != y->kids[b->a + 2 * i + 1] ||
          !tree_equal(x, x->kids[a->a + 2 * i], y, y->kids[b->a + 2 * i])) return 0;
    return 1;
  default: return tree_equal(x, a->a, y, b->a) && tree_equal(x, a->b, y, b->b);
  }
}

int cs_model_equal(const cs_model *x, const cs_model *y, char *why, size_t whylen) {
#define DIFF(...)                                                              \
  do {                                                                         \
    if (why) snprintf(why, whylen, __VA_ARGS__);                               \
    return 0;                                                                  \
  } while (0)
  if (x->n_vars != y->n_vars) DIFF("n_vars %d vs %d", x->n_vars, y->n_vars);
  if (x->objective != y->objective) DIFF("objective %d vs %d", x->objective, y->objective);
  if (x->obj_var != y->obj_var) DIFF("obj_var %d vs %d", x->obj_var, y->obj_var);
  for (int32_t v = 0; v < x->n_vars; v++) {
    if (strcmp(x->names[v], y->names[v]) != 0) DIFF("name[%d] %s vs %s", v, x->names[v], y->names[v]);
    if (x->dom[v].lo != y->dom[v].lo || x->dom[v].hi != y->dom[v].hi)
      DIFF("dom[%s] [%d,%d] vs [%d,%d]", x->names[v], x->dom[v].lo, x->dom[v].hi, y->dom[v].lo, y->dom[v].hi);
    if (x->prio[v] != y->prio[v]) DIFF("prio[%s] %lld vs %lld", x->names[v], (long long)x->prio[v], (long long)y->prio[v]);
  }
  if ((x->root < 0) != (y->root < 0)) DIFF("root presence");
  if (x->root >= 0 && !tree_equal(x, x->root, y, y->root)) DIFF("root trees differ");
  if ((x->clause_node != NULL) && (y->clause_node != NULL)) {
    if (x->n_clauses != y->n_clauses) DIFF("n_clauses %d vs %d", x->n_clauses, y->n_clauses);
    for (int32_t c = 0; c < x->n_clauses; c++)
      if (!tree_equal(x, x->clause_node[c], y, y->clause_node[c])) DIFF("clause %d differs", c);
    for (int32_t v = 0; v <= x->n_vars; v++)
      if (x->list_off[v] != y->list_off[v]) DIFF("list_off[%d] %d vs %d", v, x->list_off[v], y->list_off[v]);
    for (int32_t i = 0; i < x->list_off[x->n_vars]; i++)
      if (x->list[i] != y->list[i]) DIFF("list[%d] %d vs %d", i, x->list[i], y->list[i]);
  }
#undef DIFF
  return 1;
}
