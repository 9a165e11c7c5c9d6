/* cs_normalize.c -- the root normalisation pass on the index-based model.
 *
 * Reference: src/normalize.c:67-316, called once between the two root propagations
 * (src/parser.y:64-68; the do-while there runs exactly once because normal_wand always returns
 * its argument).  It is a host-side tree REWRITER: it folds sub-expressions whose
 * interval evaluation is a single value into constants, drops neutral elements, moves constants
 * across `<`, removes double negations, applies De Morgan.  It never narrows a domain.  It is
 * restated here because the shape of the trees decides which clauses exist and in which order
 * they appear in the per-variable lists (tests compare the result with the reference's own
 * post-root dumps node for node).
 *
 * The evaluation it needs (normal_eval, normalize.c:67-75) uses the same `cs_arith.h` source
 * as the device kernels, on the host, exactly as the reference calls eval.c from normalize.c.
 */
#include "cs_model.h"

#include <stdlib.h>

static cs_val node_eval(const cs_model *m, int32_t node) {
  const cs_node *n = &m->nodes[node];
  switch (n->op) {
  case CS_OP_VAR: return m->dom[n->a];
  case CS_OP_CONST: return cs_interval(n->a, n->b);
  case CS_OP_EQ: return cs_ev_eq(node_eval(m, n->a), node_eval(m, n->b));
  case CS_OP_LT: return cs_ev_lt(node_eval(m, n->a), node_eval(m, n->b));
  case CS_OP_NEG: return cs_ev_neg(node_eval(m, n->a));
  case CS_OP_ADD: return cs_ev_add(node_eval(m, n->a), node_eval(m, n->b));
  case CS_OP_MUL: return cs_ev_mul(node_eval(m, n->a), node_eval(m, n->b));
  case CS_OP_NOT: return cs_ev_not(node_eval(m, n->a));
  case CS_OP_AND: return cs_ev_and(node_eval(m, n->a), node_eval(m, n->b));
  case CS_OP_OR: return cs_ev_or(node_eval(m, n->a), node_eval(m, n->b));
  case CS_OP_WAND: {
    int any_false = 0, all_true = 1;
    for (int32_t i = 0; i < n->b; i++) {
      cs_val v = node_eval(m, m->kids[m->nodes[node].a + i]);
      any_false |= cs_is_false(v);
      all_true &= cs_is_true(v);
    }
    return cs_tv(all_true && !any_false, any_false);
  }
  default: return cs_interval(0, 1);
  }
}

static int is_term(const cs_model *m, int32_t node) {
  return m->nodes[node].op == CS_OP_VAR || m->nodes[node].op == CS_OP_CONST;
}

static cs_val term_val(const cs_model *m, int32_t node) {
  const cs_node *n = &m->nodes[node];
  return n->op == CS_OP_VAR ? m->dom[n->a] : cs_interval(n->a, n->b);
}

/* is_const (csolve.h:182-184): a terminal -- variable or not -- that is a single value */
static int is_const(const cs_model *m, int32_t node) { return is_term(m, node) && cs_is_value(term_val(m, node)); }

/* update_expr / update_unary_expr (normalize.c:36-60): a fresh node only if a child changed */
static int32_t with_children(cs_model *m, int32_t node, int32_t l, int32_t r) {
  if (l == m->nodes[node].a && r == m->nodes[node].b) return node;
  return cs_model_add_node(m, m->nodes[node].op, l, r);
}

static int32_t with_child(cs_model *m, int32_t node, int32_t l) {
  if (l == m->nodes[node].a) return node;
  return cs_model_add_node(m, m->nodes[node].op, l, -1);
}

static int32_t norm(cs_model *m, int32_t node);

/* NORM_EVAL (normalize.c:28-34, 67-75) */
#define FOLD(NODE)                                                             \
  do {                                                                         \
    cs_val v_ = node_eval(m, (NODE));                                          \
    if (cs_is_value(v_)) return cs_model_add_node(m, CS_OP_CONST, v_.lo, v_.lo); \
  } while (0)

/* normal_eq, normalize.c:83-101 */
static int32_t norm_eq(cs_model *m, int32_t node) {
  FOLD(node);
  int32_t l = norm(m, m->nodes[node].a);
  int32_t r = norm(m, m->nodes[node].b);
  if (l == r) return cs_model_add_node(m, CS_OP_CONST, 1, 1);
  return with_children(m, node, l, r);
}

static int32_t norm_arith(cs_model *m, int32_t node, int32_t neutral);
static int32_t norm_unary(cs_model *m, int32_t node);

/* normal_lt, normalize.c:104-161 */
static int32_t norm_lt(cs_model *m, int32_t node) {
  FOLD(node);
  int32_t l = norm(m, m->nodes[node].a);
  int32_t r = norm(m, m->nodes[node].b);
  if (l == r) return cs_model_add_node(m, CS_OP_CONST, 0, 0);
  if (m->nodes[l].op == CS_OP_NEG && m->nodes[r].op == CS_OP_NEG)
    return with_children(m, node, m->nodes[r].a, m->nodes[l].a);
  if (is_const(m, l)) {
    if (m->nodes[r].op == CS_OP_ADD && is_const(m, m->nodes[r].b)) { /* c < x + k  ->  c + -k < x */
      int32_t c = cs_model_add_node(m, CS_OP_NEG, m->nodes[r].b, -1);
      c = norm_arith(m, with_children(m, r, l, c), 0);
      return with_children(m, node, c, m->nodes[r].a);
    }
    if (m->nodes[r].op == CS_OP_NEG) /* c < -x  ->  x < -c */
      return with_children(m, node, m->nodes[r].a, norm_unary(m, with_child(m, r, l)));
  }
  if (is_const(m, r)) {
    if (m->nodes[l].op == CS_OP_ADD && is_const(m, m->nodes[l].b)) { /* x + k < c  ->  x < c + -k */
      int32_t c = cs_model_add_node(m, CS_OP_NEG, m->nodes[l].b, -1);
      c = norm_arith(m, with_children(m, l, r, c), 0);
      return with_children(m, node, m->nodes[l].a, c);
    }
    if (m->nodes[l].op == CS_OP_NEG) /* -x < c  ->  -c < x */
      return with_children(m, node, norm_unary(m, with_child(m, l, r)), m->nodes[l].a);
  }
  return with_children(m, node, l, r);
}

/* normal_arith for ADD (neutral 0) and MUL (neutral 1), normalize.c:164-194 */
static int32_t norm_arith(cs_model *m, int32_t node, int32_t neutral) {
  FOLD(node);
  const int32_t op = m->nodes[node].op;
  int32_t l = norm(m, m->nodes[node].a);
  int32_t r = norm(m, m->nodes[node].b);
  if (is_const(m, l)) return with_children(m, node, r, l);
  if (is_const(m, r) && term_val(m, r).lo == neutral) return l;
  if (m->nodes[r].op == op && is_const(m, m->nodes[r].b))
    return with_children(m, node, with_children(m, r, l, m->nodes[r].a), m->nodes[r].b);
  if (m->nodes[l].op == op && is_const(m, m->nodes[l].b))
    return with_children(m, node, m->nodes[l].a, with_children(m, l, r, m->nodes[l].b));
  return with_children(m, node, l, r);
}

/* normal_unary for NEG and NOT, normalize.c:207-220 */
static int32_t norm_unary(cs_model *m, int32_t node) {
  FOLD(node);
  const int32_t op = m->nodes[node].op;
  int32_t l = norm(m, m->nodes[node].a);
  if (m->nodes[l].op == op) return m->nodes[l].a;
  return with_child(m, node, l);
}

/* normal_logic for AND (neutral: true, dual OR) and OR (neutral: false, dual AND), normalize.c:233-269 */
static int32_t norm_logic(cs_model *m, int32_t node) {
  FOLD(node);
  const int32_t op = m->nodes[node].op;
  int32_t l = norm(m, m->nodes[node].a);
  int32_t r = norm(m, m->nodes[node].b);
  if (l == r) return l;
  if (is_term(m, l) && (op == CS_OP_AND ? cs_is_true(term_val(m, l)) : cs_is_false(term_val(m, l)))) return r;
  if (is_term(m, r) && (op == CS_OP_AND ? cs_is_true(term_val(m, r)) : cs_is_false(term_val(m, r)))) return l;
  if (m->nodes[l].op == CS_OP_NOT && m->nodes[r].op == CS_OP_NOT) { /* De Morgan */
    int32_t c = cs_model_add_node(m, op == CS_OP_AND ? CS_OP_OR : CS_OP_AND, m->nodes[l].a, m->nodes[r].a);
    return with_child(m, l, c);
  }
  return with_children(m, node, l, r);
}

/* normal_wand, normalize.c:282-295: elements are replaced in place, the node itself is kept */
static int32_t norm_wand(cs_model *m, int32_t node) {
  for (int32_t i = 0; i < m->nodes[node].b; i++) {
    int32_t o = m->kids[m->nodes[node].a + i];
    int32_t c = norm(m, o);
    if (c != o) m->kids[m->nodes[node].a + i] = c;
  }
  return node;
}

static int32_t norm(cs_model *m, int32_t node) {
  switch (m->nodes[node].op) {
  case CS_OP_VAR: case CS_OP_CONST: return node; /* normal_term, normalize.c:78-80 */
  case CS_OP_EQ: return norm_eq(m, node);
  case CS_OP_LT: return norm_lt(m, node);
  case CS_OP_ADD: return norm_arith(m, node, 0);
  case CS_OP_MUL: return norm_arith(m, node, 1);
  case CS_OP_NEG: case CS_OP_NOT: return norm_unary(m, node);
  case CS_OP_AND: case CS_OP_OR: return norm_logic(m, node);
  case CS_OP_WAND: return norm_wand(m, node);
  default: return node;
  }
}

/* normalize(), normalize.c:305-316.  Returns the (unchanged) root. */
int32_t cs_model_normalize(cs_model *m) {
  if (m->root < 0) return -1;
  int32_t prev, cur = m->root;
  do {
    prev = cur;
    cur = norm(m, cur);
  } while (cur != prev);
  m->root = cur;
  /* the clause index refers to the old element nodes */
  free(m->clause_node); free(m->list_off); free(m->list);
  m->clause_node = NULL;
  m->list_off = NULL;
  m->list = NULL;
  m->n_clauses = 0;
  return cur;
}
