/* cs_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded restatement of the reference propagator
 * (reference src/arith.c, src/eval.c, src/propagate.c) and of the parts of the
 * search driver that feed it (reference src/csolve.c, src/strategy.c,
 * src/objective.c, src/util.c bind trail), interpreting the index-based model of
 * csolve_amd/csrc/cs_model.h.  It reproduces the reference's revision ORDER
 * (depth-first, Gauss-Seidel, prop_tag skipping), so it also reproduces the
 * reference's PROPS / CALLS / CUTS counters, not just its fixpoints.
 *
 * Who may use it: tests/, __graft_entry__.smoke(), and the cpu_baseline leg of
 * bench.py.  The product (csolve_amd/) never includes, links or calls it.
 *
 * Parity pin: checked against (1) the known-answer vectors of the reference's
 * own unit tests (tests/golden/ref_unit_*.json) and (2) outputs of the compiled
 * reference itself (oracle/_ref, tests/golden/walk_*.bin, solve_*.json).
 */
#ifndef CS_ORACLE_H
#define CS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../csolve_amd/csrc/cs_model.h" /* data format only */

#ifdef __cplusplus
extern "C" {
#endif

#define CSO_ERROR (-1) /* PROP_ERROR, reference csolve.h:84 */

typedef struct cso cso;

/* scalar layer, reference arith.c:27-85 */
int32_t cso_neg(int32_t a);
int32_t cso_add(int32_t a, int32_t b);
int32_t cso_mul(int32_t a, int32_t b);
int32_t cso_min(int32_t a, int32_t b);
int32_t cso_max(int32_t a, int32_t b);

/* An oracle instance owns a private copy of the model's domains and constants. */
cso *cso_new(const cs_model *m);
void cso_free(cso *o);

/* root_phase = 1: terminals have no environment yet (parser.y:55-70): a narrowing
 * is stored directly, nothing is trailed, counted or recursed (propagate.c:81-83).
 * root_phase = 0: search phase (env set): bind + props++ + recursion (propagate.c:75-79). */
void cso_set_root_phase(cso *o, int on);
/* record_only = 1 reproduces the reference unit tests' mocked bind(): a narrowing is
 * logged and counted as 1 but the domain is left unchanged (test/test_propagate.c:52-54). */
void cso_set_record_only(cso *o, int on);
/* test hooks for the reference's strategy.c unit vectors (see cs_oracle.c) */
void cso_test_set_strategy(cso *o, int order_kind, int prefer_failing);
void cso_test_set_prio(cso *o, int32_t var, int64_t prio);
int cso_test_var_cmp(const cso *o, int32_t v1, int32_t v2);
void cso_test_heap_load(cso *o, const int32_t *vars, int32_t n);
int32_t cso_test_heap_op(cso *o, int op, int32_t a, int32_t b);
int32_t cso_test_heap_get(const cso *o, int32_t *out);
int32_t cso_test_heap_pos(const cso *o, int32_t var);

cs_val *cso_domains(cso *o); /* [n_vars], live */
uint64_t cso_props(const cso *o);
void cso_reset_stats(cso *o);

/* eval_<op> by node (eval.c:27-277) */
cs_val cso_eval(cso *o, int32_t node);
/* propagate_<op> by node (propagate.c:57-471); clause may be -1 */
int32_t cso_propagate_node(cso *o, int32_t node, cs_val val, int32_t clause);
/* propagate(): repeated sweeps of propagate_<op>(node, [1,1]) (propagate.c:474-485) */
int32_t cso_propagate(cso *o, int32_t node, size_t limit);
/* propagate_clauses(&var->clauses) (propagate.c:488-538) */
int32_t cso_propagate_clauses(cso *o, int32_t var);

/* bind trail (util.c:122-173) */
size_t cso_bind_depth(const cso *o);
void cso_bind(cso *o, int32_t var, cs_val val, int32_t clause);
void cso_unbind(cso *o, size_t depth);
/* bind log of record_only mode */
size_t cso_log_len(const cso *o);
void cso_log_get(const cso *o, size_t i, int32_t *var, cs_val *val);
void cso_log_clear(cso *o);

/* One node instance, exactly what the GPU batch kernel computes:
 * domains := dom_in; if var >= 0: dom[var] := val (the untrailed-count assignment of
 * csolve.c:294-304) then propagate_clauses(var); if var < 0: root sweeps to fixpoint.
 * Returns CSO_ERROR or the reference's PROPS for this node; dom_out (may alias
 * dom_in) receives the domains (meaningful only on success). */
int64_t cso_instance(cso *o, const cs_val *dom_in, int32_t var, cs_val val, cs_val *dom_out);
/* test hook: the variables whose priority the last cso_instance bumped, in the reference's order (the variable whose
 * domain emptied, then the recursion stack innermost first: propagate.c:33-54); returns how many there were */
int32_t cso_test_bumps(const cso *o, int32_t *out, int32_t cap);

/* cso_instance over a batch with single-value assignments.  states_in holds the parent
 * states ([*][n_vars]); nodes[i] = {var, lo, hi, parent_row}; status[i] = CSO_ERROR or PROPS;
 * states_out row i (may be NULL).  Returns the total number of binds, failed nodes included. */
uint64_t cso_instances(cso *o, const cs_val *states_in, const int32_t *nodes, int64_t count, cs_val *states_out,
                       int64_t *status);

/* ---- search driver (csolve.c:398-476) ---- */
typedef struct cso_options {
  int prefer_failing;        /* -f, default 1 */
  uint64_t restart_frequency; /* -r, default 100 */
  int order;                 /* -o: 0 none, 1 smallest-domain, 2 largest-domain, 3 smallest-value, 4 largest-value */
  uint64_t max_calls;        /* stop after this many nodes (0 = unlimited) */
  uint64_t max_solutions;    /* capacity of the solution buffer */
} cso_options;

typedef struct cso_result {
  uint64_t calls, cuts, props, restarts, solutions;
  int32_t best;      /* incumbent objective value (objective.c:81-98) */
  int stopped_early; /* max_calls hit */
  int32_t *solution_values; /* [solutions_stored][n_vars] */
  uint64_t solutions_stored;
} cso_result;

void cso_default_options(cso_options *opt);
/* runs solve() on the oracle's current (post-root) domains; the model must be indexed.
 * Conflict-clause learning is not restated: use on problems where the reference learns
 * none (no 0/1 variable involved) or compare with the reference at `-c false`. */
int cso_solve(cso *o, const cso_options *opt, cso_result *res);
void cso_result_free(cso_result *res);

#ifdef __cplusplus
}
#endif
#endif
