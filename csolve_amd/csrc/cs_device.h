/* cs_device.h -- the immutable device image of a problem: what the HIP kernels read.
 *
 * Built on the host from a cs_model (cs_device.c), uploaded once, shared by every
 * node instance of every launch.  All arrays are int32.
 *
 * Two views of the same clause set:
 *
 *  (1) clause-centric, for full sweeps (root phase, reference propagate.c:474-485 /
 *      379-392) and for evaluating the root (eval.c:233-255):
 *        clause[4*c .. 4*c+3] = { kind, a, b, d }
 *          CS_CL_SKIP  constant-true element, never narrows, never fails
 *          CS_CL_NE    X_a != X_b + d   (binary fast path, see below)
 *          CS_CL_TREE  general expression tree, a = tree id
 *          CS_CL_EQ    X_a  = X_b + d   (linear fast paths, see below)
 *          CS_CL_LT    X_a  < X_b + d
 *          CS_CL_OR2   lit[a] or lit[a+1], each literal {a, b, d, 0}: X_a < X_b + d
 *
 *  (2) variable-centric, for the event-driven fixpoint (propagate.c:488-538):
 *        adj_off[v] .. adj_off[v+1]   entries {x, y} of variable v, in the order of
 *        the reference's per-variable clause list (parser_support.c:338-396)
 *          x >= 0 : binary relation seen from v, other variable = x & 0x0fffffff, relation = x >> 28:
 *                   0  X_v != X_o + y     1  X_v = X_o + y     2  X_v < X_o + y     3  X_v > X_o + y
 *          x <  0 : y == 0: tree clause, tree id = ~x;  y == 1: two-literal disjunction, first literal = ~x
 *
 * Binary fast path.  A clause NOT(EQ(L, R)) where L and R are each `VAR` or
 * `VAR + constant` (constant on either side of the ADD, possibly written as a NEG of
 * a constant) over two different variables is stored as X_a != X_b + d.  It is only
 * chosen when every bound and constant involved is below 2^30 in magnitude, so that
 * none of the reference's saturating additions (arith.c:38-51) can saturate and the
 * record is exactly equivalent to revising the tree through propagate_not ->
 * propagate_eq(false) -> propagate_add -> propagate_term (propagate.c:289-301, 123-136,
 * 106-120, 223-246, 57-87).
 *
 * Linear fast paths (schedule.txt-style models: precedences `s + d <= t`, definitions `e = s + d`,
 * disjunctive resources `a > b | c > d`).  A clause EQ(L, R), LT(L, R), NOT(LT(L, R)) or
 * OR(literal, literal) whose L and R are `VAR` or `VAR + constant` over two different variables, under
 * the same magnitude limits as the NE path, is stored in the normal form above (NOT(LT(L,R)) becomes
 * R < L + 1) and revised by bound propagation: exactly what propagate_eq / propagate_lt / propagate_not /
 * propagate_or / propagate_add push down to the two variables (propagate.c:139-246, 289-340).  Tests
 * compare these paths with the tree interpreter on the same models (csgpu_model_set_fast_paths).
 *
 * Trees.  tree_off[t] .. tree_off[t+1] index tnode[], 4 ints per node { op, a, b, 0 } in
 * post-order (children before parents, the root last); child references are indices local
 * to the tree; VAR: a = variable; CONST: a = lo, b = hi; WAND: a = offset into tkid[],
 * b = count, tkid[] holds local indices; CONFL (a learnt conflict clause, csolve.h:98-128): a = offset into
 * tkid[], b = number of elements, tkid[] holds the pairs { local index of the terminal, conflict value }.
 */
#ifndef CS_DEVICE_H
#define CS_DEVICE_H

#include "cs_model.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { CS_CL_SKIP = 0, CS_CL_NE = 1, CS_CL_TREE = 2, CS_CL_EQ = 3, CS_CL_LT = 4, CS_CL_OR2 = 5 };
enum { CS_REL_NE = 0, CS_REL_EQ = 1, CS_REL_LT = 2, CS_REL_GT = 3 };
#define CS_ADJ_VAR_MASK 0x0fffffff

/* largest tree a device lane can revise (per-lane value scratch), and the deepest
 * pending-push stack it keeps */
#define CS_MAX_TREE_NODES 256

typedef struct cs_dev_image {
  int32_t n_vars, n_clauses;
  int32_t n_adj;        /* adjacency entries */
  int32_t n_ne, n_tree_clauses, n_skip;
  int32_t n_lin, n_or2;  /* EQ/LT clauses, two-literal disjunctions */
  int32_t n_lits;
  int32_t *lit;         /* [4*n_lits] {a, b, d, 0}: X_a < X_b + d */
  int32_t max_list;     /* longest per-variable list */
  int32_t *adj_off;     /* [n_vars+1] */
  int32_t *adj;         /* [2*n_adj] */
  int32_t *adj_clause;  /* [n_adj] the clause behind every adjacency entry (traces) */
  int32_t *clause;      /* [4*n_clauses] */
  int32_t n_trees, n_tnodes, n_tkids, max_tree;
  int32_t *tree_off;    /* [n_trees+1] */
  int32_t *tnode;       /* [4*n_tnodes] */
  int32_t *tkid;        /* [n_tkids] */
  int32_t *tree_want;   /* [2*n_trees] {lo,hi} pushed into each tree root, normally {1,1} */
  /* packed adjacency of pure binary-NE models for the LDS-resident kernel: entry =
   * other | (d - packed_dmin) << packed_obits, 2 or 4 bytes wide (0 = not available) */
  int32_t packed_width, packed_obits, packed_dmin;
  void *adj_packed;     /* [n_adj] uint16_t or uint32_t */
  /* symmetric adjacency for the forbidden-set kernel: every binary clause is listed under BOTH of
   * its variables, also under one that was already a single value at the root (the reference gives
   * such a variable no clause list, parser_support.c:341, because it can never change -- but its
   * value must still reach the neighbours' forbidden sets).  The packed offset of an entry (w, .)
   * is d + root_lo[w], so that the bit index is simply value - offset. */
  int32_t sym_n_adj, sym_width, sym_obits, sym_dmin;
  int32_t *sym_off;     /* [n_vars+1] */
  void *sym_packed;     /* [sym_n_adj] */
  /* the same symmetric relation as a dense table for models of at most 256 variables (register-resident
   * kernel): dense_tab[(u * dense_slots + k) * dense_cols + w] = (d + root_lo[w]) - dense_dmin of the k-th
   * clause between u and w, or the all-ones sentinel; dense_cols = n_vars rounded up to 64.
   * dense_width = bytes per entry (1 or 2), 0 = not available. */
  int32_t dense_width, dense_slots, dense_cols, dense_dmin;
  void *dense_tab;
} cs_dev_image;

/* with_lists = 0: clause-centric view only (root phase, lists not needed).
 * entailed (may be NULL): one byte per clause; a non-zero byte marks a clause whose root
 * evaluates to true in the root state.  Domains only shrink below the root, so such a clause
 * can never narrow or fail again; it is stored as CS_CL_SKIP and left out of the adjacency.
 * (The reference reaches the same effect by folding the clause to the constant 1 in its
 * root normalisation pass, reference src/normalize.c:67-75 via parser.y:66.) */
/* 1 (default): EQ / LT / two-literal OR clauses take the linear fast paths; 0: they stay trees (tests) */
extern int cs_dev_linear_fast_paths;

cs_dev_image *cs_dev_image_build(const cs_model *m, int with_lists, const unsigned char *entailed, char *err,
                                 size_t errlen);
void cs_dev_image_free(cs_dev_image *img);

#ifdef __cplusplus
}
#endif
#endif
