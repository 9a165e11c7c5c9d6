/* cs_model.h -- the host-side problem model: an index-based (pointer-free)
 * restatement of the reference's constraint trees, variable table and
 * per-variable clause lists.
 *
 * Reference data model being mirrored (reference src/csolve.h):
 *   struct constr_t  105-130   -> cs_node {op,a,b} in one array, children by index
 *   struct wand_t    120-123   -> CS_OP_WAND node, children in kids[a .. a+b)
 *   struct env_t     231-239   -> parallel arrays dom[], names[], prio[] by variable index
 *   clause_list_t    225-228   -> CSR list_off[] / list[] of clause ids
 *   wand_expr_t       92-96    -> a "clause" = one element slot of a wide-and that is
 *                                 reachable from the root through wide-ands only
 *                                 (reference src/parser_support.c:351-396)
 * The model is what the device tables are built from (cs_device.c) and what the
 * CPU oracle (oracle/) interprets; it is also the on-disk golden-model format.
 */
#ifndef CS_MODEL_H
#define CS_MODEL_H

#include <stddef.h>
#include <stdint.h>

#include "cs_arith.h"
#include "cs_frontend.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cs_node {
  int32_t op; /* enum cs_op */
  int32_t a;  /* VAR: variable; CONST: lo; unary/binary: left child; WAND: kids offset */
  int32_t b;  /* CONST: hi; binary: right child; WAND: child count; else -1 */
} cs_node;

typedef struct cs_model {
  /* variables */
  int32_t n_vars, cap_vars;
  cs_val *dom;       /* current root domains */
  char **names;
  int64_t *prio;     /* initial ordering weights (parser.y:219-265) */
  int32_t *var_node; /* the one CS_OP_VAR node of each variable */
  /* expression nodes */
  int32_t n_nodes, cap_nodes;
  cs_node *nodes;
  int32_t n_kids, cap_kids;
  int32_t *kids;
  int32_t root;      /* CS_OP_WAND node holding the top-level constraints, -1 if none yet */
  /* top-level elements are collected here while parsing, then frozen into root */
  int32_t n_top, cap_top;
  int32_t *top;
  /* objective (parser.y:109-131, objective.c:34-53) */
  int32_t objective; /* enum cs_objective */
  int32_t obj_var;   /* variable index of "<obj>", -1 for ANY/ALL */
  int32_t weights_on; /* strategy_compute_weights(), default true (csolve.h:412) */
  /* clause index (cs_model_index) */
  int32_t n_clauses;
  int32_t *clause_node; /* [n_clauses] root node of each clause, root DFS order */
  int32_t *list_off;    /* [n_vars+1] */
  int32_t *list;        /* clause ids per variable, reference clause-list order */
  cs_val *clause_want;  /* optional [n_clauses]: value pushed into each clause root; NULL = true
                           for all (used by the single-operator entry points of the drop-in) */
  /* hash of names -> variable, open addressing */
  int32_t *name_tab;
  int32_t name_cap;
  char err[160];
} cs_model;

cs_model *cs_model_new(void);
void cs_model_free(cs_model *m);
/* deep copy of trees, domains, names and weights; the clause index is left out (cs_model_index rebuilds it) */
cs_model *cs_model_clone(const cs_model *src);

/* low-level construction (used by the builder, the loaders and the drop-in shim) */
int32_t cs_model_add_var(cs_model *m, const char *name, cs_val dom);
int32_t cs_model_find_var(const cs_model *m, const char *name);
int32_t cs_model_add_node(cs_model *m, int32_t op, int32_t a, int32_t b);
int32_t cs_model_add_wand(cs_model *m, const int32_t *elems, int32_t n);
/* a learnt conflict clause "not all of term_i == value_i" (struct confl_t, csolve.h:98-128): CS_OP_CONFL node,
 * a = offset into kids[], b = number of elements; kids holds the pairs { terminal node, value } */
/* append a top-level clause after the fact (root wide-and, and the clause index if the model has one) */
int cs_model_append_clause(cs_model *m, int32_t node);
int32_t cs_model_add_confl(cs_model *m, const int32_t *term_nodes, const int32_t *values, int32_t n);
void cs_model_set_root_from_top(cs_model *m);

/* text -> model; returns NULL and fills err on a parse error */
cs_model *cs_model_parse(const char *text, int weights_on, char *err, size_t errlen);

/* rebuild clause_node / list_off / list from the current domains, replaying
 * clauses_init() (reference src/parser_support.c:338-396): a variable that is
 * already a single value gets no clause list. */
int cs_model_index(cs_model *m);

/* the root normalisation pass (cs_normalize.c; reference src/normalize.c:305-316).  Rewrites the
 * trees in place using the current domains, drops the clause index.  Returns the root node. */
int32_t cs_model_normalize(cs_model *m);

/* env_generate() check (parser_support.c:245-257): index of the first variable with
 * an infinite bound, or -1 */
int32_t cs_model_first_unbounded(const cs_model *m);

/* binary golden-model files (little-endian int32 stream, see cs_model.c) */
int cs_model_save(const cs_model *m, const char *path);
cs_model *cs_model_load(const char *path, char *err, size_t errlen);

/* structural equality of two models (nodes, kids, root, domains, names, clause index).
 * Returns 1 if equal, else 0 and a description in why. */
int cs_model_equal(const cs_model *x, const cs_model *y, char *why, size_t whylen);

/* number of nodes in the tree under `node` (shared sub-trees counted once per path) */
int32_t cs_model_tree_size(const cs_model *m, int32_t node);

#ifdef __cplusplus
}
#endif
#endif
