/* cs_device.c -- host-side construction of the device image (see cs_device.h). */
#include "cs_device.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NE_LIMIT (1 << 30) /* magnitudes below this cannot saturate when two are added */

typedef struct {
  int32_t *v;
  int32_t n, cap;
} ibuf;

static void ibuf_push(ibuf *b, int32_t x) {
  if (b->n == b->cap) {
    b->cap = b->cap ? b->cap * 2 : 256;
    b->v = (int32_t *)realloc(b->v, (size_t)b->cap * sizeof(int32_t));
    if (b->v == NULL) { fprintf(stderr, "csolve_amd: error: out of memory\n"); exit(EXIT_FAILURE); }
  }
  b->v[b->n++] = x;
}

static int small(int64_t x) { return x > -(int64_t)NE_LIMIT && x < (int64_t)NE_LIMIT; }

/* a sub-tree without variables that is a single small constant */
static int const_value(const cs_model *m, int32_t node, int64_t *c) {
  const cs_node *n = &m->nodes[node];
  if (n->op == CS_OP_CONST && n->a == n->b && small(n->a)) { *c = n->a; return 1; }
  if (n->op == CS_OP_NEG && const_value(m, n->a, c)) { *c = -*c; return 1; }
  return 0;
}

/* node == X_var + k with everything small */
static int affine(const cs_model *m, int32_t node, int32_t *var, int64_t *k) {
  const cs_node *n = &m->nodes[node];
  if (n->op == CS_OP_VAR) {
    cs_val d = m->dom[n->a];
    if (!small(d.lo) || !small(d.hi)) return 0;
    *var = n->a; *k = 0;
    return 1;
  }
  if (n->op == CS_OP_ADD) {
    int64_t c;
    if (const_value(m, n->b, &c) && affine(m, n->a, var, k)) { *k += c; return small(*k); }
    if (const_value(m, n->a, &c) && affine(m, n->b, var, k)) { *k += c; return small(*k); }
  }
  return 0;
}

static int match_ne(const cs_model *m, int32_t root, int32_t *a, int32_t *b, int32_t *d) {
  const cs_node *n = &m->nodes[root];
  if (n->op != CS_OP_NOT) return 0;
  const cs_node *e = &m->nodes[n->a];
  if (e->op != CS_OP_EQ) return 0;
  int64_t ka, kb;
  if (!affine(m, e->a, a, &ka) || !affine(m, e->b, b, &kb) || *a == *b) return 0;
  if (!small(kb - ka)) return 0;
  *d = (int32_t)(kb - ka);
  return 1;
}

int cs_dev_linear_fast_paths = 1;

/* node == X_var (bounds anything but a sentinel: the linear paths compute in 64 bits, and a bare variable
 * is never summed by the reference) or X_var + k with everything small */
static int affine_wide(const cs_model *m, int32_t node, int32_t *var, int64_t *k) {
  const cs_node *n = &m->nodes[node];
  if (n->op == CS_OP_VAR) {
    cs_val d = m->dom[n->a];
    if (d.lo == CS_DOM_MIN || d.hi == CS_DOM_MAX) return 0;
    *var = n->a; *k = 0;
    return 1;
  }
  /* VAR + constant: as on the NE path everything must be small, so that the reference's saturating sum
   * (arith.c:38-51) is the plain sum */
  if (n->op == CS_OP_ADD) return affine(m, node, var, k);
  return 0;
}

/* a literal "X_a < X_b + d" from LT(L, R) (negated = 0) or NOT(LT(L, R)) (negated = 1: R <= L, i.e.
 * X_rb < X_la + (ka - kb + 1)) */
static int lt_literal(const cs_model *m, int32_t lt_node, int negated, int32_t *a, int32_t *b, int32_t *d) {
  const cs_node *e = &m->nodes[lt_node];
  if (e->op != CS_OP_LT) return 0;
  int32_t va, vb;
  int64_t ka, kb;
  if (!affine_wide(m, e->a, &va, &ka) || !affine_wide(m, e->b, &vb, &kb) || va == vb) return 0;
  const int64_t dd = negated ? ka - kb + 1 : kb - ka;
  if (!small(dd)) return 0;
  if (negated) { *a = vb; *b = va; } else { *a = va; *b = vb; }
  *d = (int32_t)dd;
  return 1;
}

/* LT / NOT(LT) with the polarity the clause is wanted in */
static int match_lt(const cs_model *m, int32_t root, int want_true, int32_t *a, int32_t *b, int32_t *d) {
  const cs_node *n = &m->nodes[root];
  if (n->op == CS_OP_LT) return lt_literal(m, root, !want_true, a, b, d);
  if (n->op == CS_OP_NOT) return lt_literal(m, n->a, want_true, a, b, d);
  return 0;
}

/* EQ(L, R) wanted true: X_a = X_b + d */
static int match_eq(const cs_model *m, int32_t root, int32_t *a, int32_t *b, int32_t *d) {
  const cs_node *e = &m->nodes[root];
  if (e->op != CS_OP_EQ) return 0;
  int64_t ka, kb;
  if (!affine_wide(m, e->a, a, &ka) || !affine_wide(m, e->b, b, &kb) || *a == *b) return 0;
  if (!small(kb - ka)) return 0;
  *d = (int32_t)(kb - ka);
  return 1;
}

typedef struct {
  const cs_model *m;
  ibuf tnode, tkid;
  int32_t *local;   /* node id -> local index in the tree being built, -1 = not yet */
  ibuf touched;
  int32_t base;     /* first tnode index of the current tree */
  int bad;
} tree_ctx;

static int32_t emit_tree(tree_ctx *t, int32_t node) {
  if (t->local[node] >= 0) return t->local[node];
  const cs_node *n = &t->m->nodes[node];
  int32_t a = n->a, b = n->b;
  switch (n->op) {
  case CS_OP_VAR: case CS_OP_CONST:
    break;
  case CS_OP_NEG: case CS_OP_NOT:
    a = emit_tree(t, n->a); b = -1;
    break;
  case CS_OP_EQ: case CS_OP_LT: case CS_OP_ADD: case CS_OP_MUL: case CS_OP_AND: case CS_OP_OR:
    a = emit_tree(t, n->a);
    b = emit_tree(t, t->m->nodes[node].b);
    break;
  case CS_OP_WAND: {
    int32_t cnt = n->b, off = n->a;
    int32_t *loc = (int32_t *)malloc((size_t)(cnt ? cnt : 1) * sizeof(int32_t));
    for (int32_t i = 0; i < cnt; i++) loc[i] = emit_tree(t, t->m->kids[off + i]);
    a = t->tkid.n;
    for (int32_t i = 0; i < cnt; i++) ibuf_push(&t->tkid, loc[i]);
    b = cnt;
    free(loc);
    break;
  }
  case CS_OP_CONFL: { /* tkid holds the pairs { local index of the terminal, conflict value } */
    int32_t cnt = n->b, off = n->a;
    int32_t *loc = (int32_t *)malloc((size_t)(cnt ? cnt : 1) * sizeof(int32_t));
    for (int32_t i = 0; i < cnt; i++) loc[i] = emit_tree(t, t->m->kids[off + 2 * i]);
    a = t->tkid.n;
    for (int32_t i = 0; i < cnt; i++) {
      ibuf_push(&t->tkid, loc[i]);
      ibuf_push(&t->tkid, t->m->kids[off + 2 * i + 1]);
    }
    b = cnt;
    free(loc);
    break;
  }
  default:
    t->bad = 1;
    break;
  }
  int32_t idx = t->tnode.n / 4 - t->base;
  ibuf_push(&t->tnode, t->m->nodes[node].op);
  ibuf_push(&t->tnode, a);
  ibuf_push(&t->tnode, b);
  ibuf_push(&t->tnode, 0);
  t->local[node] = idx;
  ibuf_push(&t->touched, node);
  return idx;
}

void cs_dev_image_free(cs_dev_image *g) {
  if (g == NULL) return;
  free(g->adj_off); free(g->adj); free(g->adj_clause); free(g->clause); free(g->tree_off); free(g->tnode); free(g->tkid); free(g->tree_want); free(g->lit); free(g->adj_packed); free(g->sym_off); free(g->sym_packed); free(g->dense_tab);
  free(g);
}

cs_dev_image *cs_dev_image_build(const cs_model *m, int with_lists, const unsigned char *entailed, char *err,
                                 size_t errlen) {
  if (m->clause_node == NULL || (with_lists && m->list_off == NULL)) {
    if (err) snprintf(err, errlen, "model has no clause index");
    return NULL;
  }
  {
    const char *e = getenv("CSGPU_LINEAR_FAST_PATHS"); /* "0" switches the linear fast paths off (debugging) */
    if (e != NULL && e[0] == '0') cs_dev_linear_fast_paths = 0;
  }
  cs_dev_image *g = (cs_dev_image *)calloc(1, sizeof *g);
  g->n_vars = m->n_vars;
  g->n_clauses = m->n_clauses;
  g->clause = (int32_t *)calloc((size_t)(m->n_clauses ? m->n_clauses : 1) * 4, sizeof(int32_t));

  tree_ctx t;
  memset(&t, 0, sizeof t);
  t.m = m;
  t.local = (int32_t *)malloc((size_t)(m->n_nodes ? m->n_nodes : 1) * sizeof(int32_t));
  for (int32_t i = 0; i < m->n_nodes; i++) t.local[i] = -1;
  ibuf tree_off = { 0 };
  ibuf want = { 0 };
  ibuf lits = { 0 };

  for (int32_t c = 0; c < m->n_clauses; c++) {
    int32_t root = m->clause_node[c];
    const cs_node *rn = &m->nodes[root];
    int32_t *rec = &g->clause[4 * c];
    int32_t a, b, d;
    const int want_true = m->clause_want == NULL || (m->clause_want[c].lo == 1 && m->clause_want[c].hi == 1);
    if (want_true && ((entailed != NULL && entailed[c]) || (rn->op == CS_OP_CONST && rn->a == 1 && rn->b == 1))) {
      rec[0] = CS_CL_SKIP;
      g->n_skip++;
    } else if (want_true && match_ne(m, root, &a, &b, &d)) {
      rec[0] = CS_CL_NE; rec[1] = a; rec[2] = b; rec[3] = d;
      g->n_ne++;
    } else if (cs_dev_linear_fast_paths && want_true && match_eq(m, root, &a, &b, &d)) {
      rec[0] = CS_CL_EQ; rec[1] = a; rec[2] = b; rec[3] = d;
      g->n_lin++;
    } else if (cs_dev_linear_fast_paths && (want_true || (m->clause_want[c].lo == 0 && m->clause_want[c].hi == 0)) &&
               match_lt(m, root, want_true, &a, &b, &d)) {
      rec[0] = CS_CL_LT; rec[1] = a; rec[2] = b; rec[3] = d;
      g->n_lin++;
    } else if (cs_dev_linear_fast_paths && want_true && rn->op == CS_OP_OR) {
      int32_t a2, b2, d2;
      if (match_lt(m, rn->a, 1, &a, &b, &d) && match_lt(m, rn->b, 1, &a2, &b2, &d2)) {
        rec[0] = CS_CL_OR2; rec[1] = lits.n / 4;
        ibuf_push(&lits, a); ibuf_push(&lits, b); ibuf_push(&lits, d); ibuf_push(&lits, 0);
        ibuf_push(&lits, a2); ibuf_push(&lits, b2); ibuf_push(&lits, d2); ibuf_push(&lits, 0);
        g->n_or2++;
      } else {
        goto as_tree;
      }
    } else {
    as_tree:
      ibuf_push(&tree_off, t.tnode.n / 4);
      t.base = t.tnode.n / 4;
      t.touched.n = 0;
      emit_tree(&t, root);
      int32_t len = t.tnode.n / 4 - t.base;
      for (int32_t i = 0; i < t.touched.n; i++) t.local[t.touched.v[i]] = -1;
      if (len > g->max_tree) g->max_tree = len;
      rec[0] = CS_CL_TREE; rec[1] = g->n_trees;
      ibuf_push(&want, m->clause_want ? m->clause_want[c].lo : 1);
      ibuf_push(&want, m->clause_want ? m->clause_want[c].hi : 1);
      g->n_trees++;
      g->n_tree_clauses++;
    }
  }
  ibuf_push(&tree_off, t.tnode.n / 4);
  g->tree_off = tree_off.v;
  g->tnode = t.tnode.v ? t.tnode.v : (int32_t *)calloc(4, sizeof(int32_t));
  g->n_tnodes = t.tnode.n / 4;
  g->tkid = t.tkid.v ? t.tkid.v : (int32_t *)calloc(1, sizeof(int32_t));
  g->n_tkids = t.tkid.n;
  g->tree_want = want.v ? want.v : (int32_t *)calloc(2, sizeof(int32_t));
  g->lit = lits.v ? lits.v : (int32_t *)calloc(4, sizeof(int32_t));
  g->n_lits = lits.n / 4;
  free(t.local);
  free(t.touched.v);

  if (t.bad) {
    if (err) snprintf(err, errlen, "constraint type without a device implementation");
    cs_dev_image_free(g);
    return NULL;
  }
  if (g->max_tree > CS_MAX_TREE_NODES) {
    if (err) snprintf(err, errlen, "a clause has %d nodes, device limit is %d", g->max_tree, CS_MAX_TREE_NODES);
    cs_dev_image_free(g);
    return NULL;
  }

  g->adj_off = (int32_t *)calloc((size_t)m->n_vars + 1, sizeof(int32_t));
  if (with_lists) {
    ibuf adj = { 0 };
    ibuf adjc = { 0 };
    for (int32_t v = 0; v < m->n_vars; v++) {
      g->adj_off[v] = adj.n / 2;
      for (int32_t i = m->list_off[v]; i < m->list_off[v + 1]; i++) {
        const int32_t *rec = &g->clause[4 * m->list[i]];
        if (rec[0] != CS_CL_SKIP) ibuf_push(&adjc, m->list[i]);
        if (rec[0] == CS_CL_NE) {
          if (rec[1] == v) { ibuf_push(&adj, rec[2]); ibuf_push(&adj, rec[3]); }
          else { ibuf_push(&adj, rec[1]); ibuf_push(&adj, -rec[3]); }
        } else if (rec[0] == CS_CL_EQ) { /* X_a = X_b + d  <=>  X_b = X_a - d */
          if (rec[1] == v) { ibuf_push(&adj, rec[2] | (CS_REL_EQ << 28)); ibuf_push(&adj, rec[3]); }
          else { ibuf_push(&adj, rec[1] | (CS_REL_EQ << 28)); ibuf_push(&adj, -rec[3]); }
        } else if (rec[0] == CS_CL_LT) { /* X_a < X_b + d  <=>  X_b > X_a - d */
          if (rec[1] == v) { ibuf_push(&adj, rec[2] | (CS_REL_LT << 28)); ibuf_push(&adj, rec[3]); }
          else { ibuf_push(&adj, rec[1] | (CS_REL_GT << 28)); ibuf_push(&adj, -rec[3]); }
        } else if (rec[0] == CS_CL_OR2) {
          ibuf_push(&adj, ~rec[1]); ibuf_push(&adj, 1);
        } else if (rec[0] == CS_CL_TREE) {
          ibuf_push(&adj, ~rec[1]); ibuf_push(&adj, 0);
        }
      }
      int32_t len = adj.n / 2 - g->adj_off[v];
      if (len > g->max_list) g->max_list = len;
    }
    g->adj_off[m->n_vars] = adj.n / 2;
    g->n_adj = adj.n / 2;
    g->adj = adj.v ? adj.v : (int32_t *)calloc(2, sizeof(int32_t));
    g->adj_clause = adjc.v ? adjc.v : (int32_t *)calloc(1, sizeof(int32_t));
    /* packed copy for the LDS-resident kernel: only for pure binary-NE adjacency */
    if (g->n_tree_clauses == 0 && g->n_lin == 0 && g->n_or2 == 0 && g->n_adj > 0) {
      int32_t dmin = g->adj[1], dmax = g->adj[1];
      for (int32_t i = 0; i < g->n_adj; i++) {
        if (g->adj[2 * i + 1] < dmin) dmin = g->adj[2 * i + 1];
        if (g->adj[2 * i + 1] > dmax) dmax = g->adj[2 * i + 1];
      }
      int obits = 1, dbits = 1;
      while ((1 << obits) < m->n_vars) obits++;
      while (((int64_t)1 << dbits) <= (int64_t)dmax - (int64_t)dmin) dbits++;
      const int width = obits + dbits <= 16 ? 2 : (obits + dbits <= 32 ? 4 : 0);
      if (width != 0) {
        g->packed_width = width;
        g->packed_obits = obits;
        g->packed_dmin = dmin;
        g->adj_packed = malloc((size_t)g->n_adj * (size_t)width);
        for (int32_t i = 0; i < g->n_adj; i++) {
          uint32_t e = (uint32_t)g->adj[2 * i] | ((uint32_t)(g->adj[2 * i + 1] - dmin) << obits);
          if (width == 2) ((uint16_t *)g->adj_packed)[i] = (uint16_t)e;
          else ((uint32_t *)g->adj_packed)[i] = e;
        }
      }
    }
  } else {
    g->adj = (int32_t *)calloc(2, sizeof(int32_t));
  }
  if (with_lists && g->n_tree_clauses == 0 && g->n_lin == 0 && g->n_or2 == 0 && g->n_ne > 0) {
    /* symmetric lists straight from the clause records */
    const int32_t n = m->n_vars;
    int32_t *cnt = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    /* the stored offset already contains the neighbour's root lower bound: an entry (w, dd) of u
     * says "when u is the value c, bit c - dd of w's forbidden set is set" (dd = d + root_lo[w]) */
    int64_t dmin64 = 0, dmax64 = 0;
    int first = 1;
    for (int32_t c = 0; c < g->n_clauses; c++) {
      const int32_t *rec = &g->clause[4 * c];
      if (rec[0] != CS_CL_NE) continue;
      cnt[rec[1] + 1]++;
      cnt[rec[2] + 1]++;
      const int64_t da = (int64_t)rec[3] + m->dom[rec[2]].lo, db = (int64_t)-rec[3] + m->dom[rec[1]].lo;
      if (first) { dmin64 = dmax64 = da; first = 0; }
      if (da < dmin64) dmin64 = da;
      if (db < dmin64) dmin64 = db;
      if (da > dmax64) dmax64 = da;
      if (db > dmax64) dmax64 = db;
    }
    const int32_t dmin = (int32_t)dmin64, dmax = (int32_t)dmax64;
    for (int32_t v = 0; v < n; v++) cnt[v + 1] += cnt[v];
    int obits = 1, dbits = 1;
    while ((1 << obits) < n) obits++;
    while (((int64_t)1 << dbits) <= (int64_t)dmax - (int64_t)dmin) dbits++;
    const int width = obits + dbits <= 16 ? 2 : (obits + dbits <= 32 ? 4 : 0);
    if (width != 0) {
      g->sym_n_adj = cnt[n];
      g->sym_width = width;
      g->sym_obits = obits;
      g->sym_dmin = dmin;
      g->sym_off = (int32_t *)malloc(((size_t)n + 1) * sizeof(int32_t));
      memcpy(g->sym_off, cnt, ((size_t)n + 1) * sizeof(int32_t));
      g->sym_packed = malloc((size_t)(cnt[n] ? cnt[n] : 1) * (size_t)width);
      int32_t *fill = (int32_t *)malloc(((size_t)n + 1) * sizeof(int32_t));
      memcpy(fill, cnt, ((size_t)n + 1) * sizeof(int32_t));
      for (int32_t c = 0; c < g->n_clauses; c++) {
        const int32_t *rec = &g->clause[4 * c];
        if (rec[0] != CS_CL_NE) continue;
        /* X_a != X_b + d: seen from a -> (b, d); seen from b -> (a, -d) */
        const uint32_t ea = (uint32_t)rec[2] | ((uint32_t)(rec[3] + m->dom[rec[2]].lo - dmin) << obits);
        const uint32_t eb = (uint32_t)rec[1] | ((uint32_t)(-rec[3] + m->dom[rec[1]].lo - dmin) << obits);
        const int32_t ia = fill[rec[1]]++, ib = fill[rec[2]]++;
        if (width == 2) { ((uint16_t *)g->sym_packed)[ia] = (uint16_t)ea; ((uint16_t *)g->sym_packed)[ib] = (uint16_t)eb; }
        else { ((uint32_t *)g->sym_packed)[ia] = ea; ((uint32_t *)g->sym_packed)[ib] = eb; }
      }
      free(fill);
    }
    free(cnt);
    /* dense [u][slot][w] table of the same relation (register-resident kernel) */
    if (width != 0 && n <= 256) {
      const int32_t cols = ((n + 63) / 64) * 64;
      uint8_t *mult = (uint8_t *)calloc((size_t)n * (size_t)n, 1);
      int slots = 0, ok = 1;
      for (int32_t c = 0; c < g->n_clauses && ok; c++) {
        const int32_t *rec = &g->clause[4 * c];
        if (rec[0] != CS_CL_NE) continue;
        if (rec[1] == rec[2]) { ok = 0; break; } /* x != x + d is not a pair relation */
        const int k = ++mult[(size_t)rec[1] * n + rec[2]];
        mult[(size_t)rec[2] * n + rec[1]] = (uint8_t)k;
        if (k > slots) slots = k;
        if (k >= 32) ok = 0;
      }
      int64_t cmax = 0;
      for (int32_t v = 0; v < n; v++) if (m->dom[v].hi > cmax) cmax = m->dom[v].hi;
      /* the sentinel S must never decode to a bit: c - dmin - S < 0 for every value c, and every real
       * offset must be below S */
      const int64_t span = (cmax - dmin64 > dmax64 - dmin64 ? cmax - dmin64 : dmax64 - dmin64);
      const int dw = span < 255 ? 1 : (span < 65535 ? 2 : 0);
      if (ok && slots > 0 && dw != 0 && (size_t)n * slots * cols * dw <= 144u * 1024u) {
        const size_t total = (size_t)n * slots * cols;
        g->dense_width = dw;
        g->dense_slots = slots;
        g->dense_cols = cols;
        g->dense_dmin = dmin;
        g->dense_tab = malloc(total * (size_t)dw);
        memset(g->dense_tab, 0xff, total * (size_t)dw);
        memset(mult, 0, (size_t)n * (size_t)n);
        for (int32_t c = 0; c < g->n_clauses; c++) {
          const int32_t *rec = &g->clause[4 * c];
          if (rec[0] != CS_CL_NE) continue;
          const int k = mult[(size_t)rec[1] * n + rec[2]];
          mult[(size_t)rec[1] * n + rec[2]] = mult[(size_t)rec[2] * n + rec[1]] = (uint8_t)(k + 1);
          const uint32_t ea = (uint32_t)(rec[3] + m->dom[rec[2]].lo - dmin);  /* pushed by a into b */
          const uint32_t eb = (uint32_t)(-rec[3] + m->dom[rec[1]].lo - dmin); /* pushed by b into a */
          const size_t ia = ((size_t)rec[1] * slots + k) * cols + rec[2];
          const size_t ib = ((size_t)rec[2] * slots + k) * cols + rec[1];
          if (dw == 1) { ((uint8_t *)g->dense_tab)[ia] = (uint8_t)ea; ((uint8_t *)g->dense_tab)[ib] = (uint8_t)eb; }
          else { ((uint16_t *)g->dense_tab)[ia] = (uint16_t)ea; ((uint16_t *)g->dense_tab)[ib] = (uint16_t)eb; }
        }
      }
      free(mult);
    }
  }
  return g;
}
