/* csolve_gpu.h -- C ABI of libcsolve_hip.so: the MI355X (gfx950) implementation of
 * CSolve's constraint-propagation fixpoint.
 *
 * Plain C, plain pointers and sizes.  Device pointers are raw HIP device addresses
 * (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void* (NULL =
 * the null stream).  Every entry returns 0 on success or a negative CSGPU_E_* code;
 * csgpu_last_error() gives the message.  There is no CPU fallback anywhere behind
 * this interface: without a usable HIP device the calls fail with CSGPU_E_HIP.
 *
 * What each entry replaces in the reference (jeuneS2/csolve, paths under src/):
 *
 *   csgpu_model_from_text / _from_file    the text front end up to the point where the
 *                                         trees exist: lexer.l:36-102, parser.y:94-283
 *   csgpu_model_root_propagate            propagate(root, size) of the Input action,
 *                                         parser.y:59,67 -> propagate.c:474-485 (sweeps of
 *                                         propagate_wand 379-392 over all top-level clauses)
 *   csgpu_model_finalize                  env_generate + clauses_init, parser.y:81-83 ->
 *                                         parser_support.c:245-257, 338-396
 *   csgpu_propagate_batch                 check_assignment -> propagate_clauses(&var->clauses),
 *                                         csolve.c:247-261 -> propagate.c:488-538, for a whole
 *                                         batch of search nodes at once; each node is
 *                                         step_enter's bind(var, VALUE(v)) (csolve.c:294-304)
 *                                         followed by the event-driven fixpoint
 *   csgpu_eval_batch                      update_solution's eval of the root, csolve.c:226 ->
 *                                         eval.c:233-255 (and everything below it, eval.c:27-230)
 *
 * The csolve.h-named drop-in symbols (propagate, propagate_clauses, eval_*, ...) live in
 * libcsolve_dropin.so, declared in include/csolve_dropin.h, and are thin shims over this ABI.
 */
#ifndef CSOLVE_GPU_H
#define CSOLVE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSGPU_OK 0
#define CSGPU_E_ARG (-1)      /* bad argument */
#define CSGPU_E_PARSE (-2)    /* problem text rejected (message has the reference's wording) */
#define CSGPU_E_HIP (-3)      /* HIP runtime error / no device */
#define CSGPU_E_LIMIT (-4)    /* model exceeds a device-side limit */
#define CSGPU_E_STATE (-5)    /* call out of order (e.g. propagate before finalize) */
#define CSGPU_E_UNBOUNDED (-6) /* env_generate: "unbounded variable: %s" */

/* closed interval, layout of reference `struct val_t` (csolve.h:43-46) */
typedef struct csgpu_val {
  int32_t lo, hi;
} csgpu_val;

/* one search node of a batch: take state `parent` (or the identity index), set
 * variable `var` to [lo,hi] (a single value for step_enter, an interval for the
 * worker split of csolve.c:121-150), propagate.  var < 0: no assignment, every
 * variable counts as changed (full fixpoint). */
typedef struct csgpu_node {
  int32_t var, lo, hi, parent;
} csgpu_node;

/* per-node result.  status: -1 = PROP_ERROR (csolve.h:84); otherwise the node is consistent and
 * status = number of variables that are still open (not a single value), 0 = complete assignment.
 * props = narrowing events (the reference's PROPS counter, propagate.c:77-78),
 * revisions = clause revisions performed, rounds = worklist rounds -- and for an INCONSISTENT node (status -1)
 * the index of a variable whose domain became empty, or -1 if the kernel does not attribute the failure (kernels
 * 1, 6, 7 do; the reference bumps that variable's priority, propagate_term_confl, propagate.c:33-41; which of
 * several emptied variables is reported depends on the revision order). */
typedef struct csgpu_result {
  int32_t status, props, revisions, rounds;
} csgpu_result;

typedef struct csgpu_model csgpu_model;

const char *csgpu_last_error(void);
/* number of visible HIP devices (0 or negative error); selects `device` for this thread */
int csgpu_device_count(void);
int csgpu_set_device(int device);

/* ---- host model ---- */
int csgpu_model_from_text(const char *text, int weights_on, csgpu_model **out);
int csgpu_model_from_file(const char *path, int weights_on, csgpu_model **out);
/* a golden-model file (csolve_amd/csrc/cs_model.c) with domains and clause lists already final */
int csgpu_model_from_dump(const char *path, csgpu_model **out);
void csgpu_model_free(csgpu_model *m);

int csgpu_model_num_vars(const csgpu_model *m);
int csgpu_model_num_clauses(const csgpu_model *m);
int csgpu_model_objective(const csgpu_model *m);        /* 0 ANY 1 ALL 2 MIN 3 MAX */
int csgpu_model_objective_var(const csgpu_model *m);    /* -1 if none */
const char *csgpu_model_var_name(const csgpu_model *m, int var);
/* copy the current root domains (host memory, n_vars entries) */
int csgpu_model_get_domains(const csgpu_model *m, csgpu_val *out);
int csgpu_model_set_domains(csgpu_model *m, const csgpu_val *in);
/* table sizes of the uploaded device image: [0] adjacency entries, [1] binary-NE clauses,
 * [2] tree clauses, [3] tree nodes, [4] LDS bytes per node instance, [5] max list length */
int csgpu_model_device_info(const csgpu_model *m, int64_t info[8]);

/* Root phase on the device: full sweeps over every top-level clause until nothing
 * changes.  *status = -1 if the problem is infeasible, else the number of narrowings.
 * The model's root domains are updated in place.  (The reference stops after limit+1
 * Gauss-Seidel sweeps, propagate.c:483; the device runs Jacobi rounds to the fixpoint,
 * which is the same state whenever the reference reaches its fixpoint within the limit.) */
int csgpu_model_root_propagate(csgpu_model *m, int32_t *status);
/* The same with the reference's `limit` (propagate.c:479-483: at most limit + 1 sweeps; limit < 0: none).  The
 * device's sweeps revise all clauses of a round in parallel, the reference's one after the other (Gauss-Seidel),
 * so k device rounds never narrow more than k reference sweeps: whenever the device reaches its fixpoint within
 * limit + 1 rounds (*rounds, if wanted, says how many it took) the reference does too and the states are equal;
 * when the limit cuts the device short the result is a sound but possibly wider state than the reference's. */
int csgpu_model_root_propagate_limit(csgpu_model *m, int64_t limit, int32_t *status, int32_t *rounds);

/* The root normalisation pass between the two root propagations (normalize(), reference
 * src/normalize.c:305-316, parser.y:66): a host-side rewrite of the trees (constant folding,
 * neutral elements, constants moved across `<`, double negation, De Morgan); no domain changes. */
int csgpu_model_normalize(csgpu_model *m);
/* SURVEY 8f-1 (the normaliser as a pre-pass that specialises the tables per subtree prefix): a new, unfinalized model
 * for the subtree below `state` (one interval per variable, inside the model's root domains; normally a consistent
 * state a search has reached): the same trees with `state` as root domains.  csgpu_model_normalize on it folds what the
 * prefix has decided (normalize.c:67-316), csgpu_model_root_propagate re-establishes the fixpoint, csgpu_model_finalize
 * leaves entailed clauses out of the device tables: shorter clause lists, the same results for every state inside
 * `state` (normalisation never changes results, only cost).  The new model is independent of `m`. */
int csgpu_model_specialize(const csgpu_model *m, const csgpu_val *state, csgpu_model **out);
/* A learnt conflict clause (struct confl_t, csolve.h:98-128; conflict_create, conflict.c:319-361): "not all of
 * vars[i] == values[i]".  It is evaluated like eval_confl (eval.c:258-277) and propagated like propagate_confl
 * (propagate.c:395-471: when every element but one has its conflict value, that value is shaved off the bound of
 * the remaining variable it sits on).  The clause becomes the last top-level clause and the last entry of its
 * variables' clause lists.  May be called before or after csgpu_model_finalize; afterwards the device tables are
 * rebuilt (the call costs O(model)) -- which frees the old ones: while a csgpu_search built on the model exists (its
 * kernels' arguments and captured graphs point into them) the call is refused with CSGPU_E_STATE; free the engines
 * first.  At most 255 elements (CSGPU_E_LIMIT). */
int csgpu_model_add_conflict(csgpu_model *m, int32_t count, const int32_t *vars, const int32_t *values);

/* eval_<op> on the current root domains, host buffers, no finalize needed (variables may
 * still be unbounded): vals[c] = interval value of clause c (eval.c:27-255).  Synchronous. */
int csgpu_model_eval_clauses_host(csgpu_model *m, csgpu_val *vals);

/* Host-only half of finalize: env_generate + clauses_init + construction of the device
 * tables in host memory (no HIP call).  csgpu_model_device_info works afterwards. */
int csgpu_model_build_tables(csgpu_model *m);

/* env_generate + clauses_init + upload of the search tables.  Fails with
 * CSGPU_E_UNBOUNDED if a variable still has an infinite bound. */
int csgpu_model_finalize(csgpu_model *m);

/* ---- sets-only states: the domains as bit vectors -------------------------------------------------------
 * For models that qualify for kernel 4 a search state can be carried as its forbidden sets alone
 * ([rows][n_vars][FW] 64-bit words, FW = csgpu_model_forbidden_words): every value outside a variable's
 * interval is marked in the variable's own set, so the interval is [lowest unmarked, highest unmarked] and
 * need not be stored.  Half the bytes per node of the interval + sets layout; same fixpoints, verdicts and
 * PROPS (tests unpack and compare).
 *   pack:    interval states -> sets (forbidden values of valued neighbours + everything outside the interval)
 *   unpack:  sets -> interval states ({1,0} for a variable with no allowed value)
 *   propagate_batch_sets: as csgpu_propagate_batch_fb, nodes[i].parent indexes d_sets_in; rows of inconsistent
 *   nodes in d_sets_out are unspecified.  CSGPU_E_LIMIT if the model does not qualify. */
int csgpu_sets_pack(const csgpu_model *m, const csgpu_val *d_states, uint64_t *d_sets, int64_t count, void *stream);
int csgpu_sets_unpack(const csgpu_model *m, const uint64_t *d_sets, csgpu_val *d_states, int64_t count, void *stream);
int csgpu_propagate_batch_sets(const csgpu_model *m, const uint64_t *d_sets_in, const csgpu_node *d_nodes,
                               uint64_t *d_sets_out, csgpu_result *d_results, int64_t batch, void *stream);

/* Kernel selection for the batched fixpoint: 0 = automatic (default), 1 = the general kernel
 * (adjacency read through L2; handles tree clauses), 2 = the LDS-resident unit-shaving kernel
 * (pure binary-NE models whose packed adjacency fits in LDS), 3 = the forbidden-set kernel (same
 * models, root intervals of at most 256 values), 4 = its register-resident variant (additionally
 * at most 256 variables and a dense pair table that fits in LDS), 5 = kernel 4 with two or four
 * nodes per wavefront (additionally at most 32 variables of at most 64 values; 4 then means one
 * node per wavefront), 6 = the clause-resident kernel for small models (at most 512 clauses: every
 * lane keeps its clauses in registers and a round revises all of them), 7 = the interval-only shaving
 * kernel (models that qualify for 4: bounds are shaved by pushes and moved bounds verified against the
 * valued variables through the symmetric pair table; no forbidden sets anywhere);
 * CSGPU_E_LIMIT if the model does not qualify.  All compute the same results; tests run every parity
 * case through each of them.
 * Automatic: csgpu_propagate_batch_fb (sets passed) uses 5 when the model qualifies, else 4, else 3;
 * csgpu_propagate_batch (states only) uses 5 with the sets rebuilt for models of at most 32 variables,
 * else 7, else 2 (more than 256 variables: rebuilding forbidden sets costs a list scan per valued variable of
 * every incoming state, the event-driven kernels scan one list per narrowing), else 6 (at most 256 clauses;
 * with 257-512 for batches of at most 8192 nodes only, where its lower latency counts and its lower throughput
 * does not), else 1. */
int csgpu_model_set_kernel(csgpu_model *m, int which);
/* Process-wide switch for models finalized afterwards: 1 (default) = EQ / LT / two-literal OR clauses
 * over `VAR` or `VAR + constant` operands are revised by direct bound propagation (schedule.txt-style
 * models then need no expression-tree interpreter); 0 = they stay expression trees.  Same fixpoints
 * either way (tests compare the two). */
void csgpu_set_linear_fast_paths(int on);
/* 1 if the finalized model can run kernel `which` (1..6), else 0 */
int csgpu_model_qualifies(const csgpu_model *m, int which);
/* which kernel csgpu_propagate_batch will launch (see above) */
int csgpu_model_get_kernel(const csgpu_model *m);

/* ---- batched propagation (the hot path) ----
 * d_states_in : device, [*][n_vars] csgpu_val   parent states
 * d_nodes     : device, [batch] csgpu_node      (parent = row of d_states_in)
 * d_states_out: device, [batch][n_vars] csgpu_val, row i = fixpoint of node i
 *               (left unwritten when the node fails)
 * d_results   : device, [batch] csgpu_result
 * Asynchronous on `stream`; no host synchronisation inside. */
int csgpu_propagate_batch(const csgpu_model *m, const csgpu_val *d_states_in, const csgpu_node *d_nodes,
                          csgpu_val *d_states_out, csgpu_result *d_results, int64_t batch, void *stream);

/* Same, with the incumbent bound of an optimisation run: before a node is propagated the
 * objective variable "<obj>" is intersected with [obj_lo, obj_hi] (objective_update_val,
 * objective.c:101-126) and, if that moved a bound, its clauses are propagated as well
 * (check_assignment, csolve.c:251-252).  Pass INT32_MIN / INT32_MAX for "no bound". */
int csgpu_propagate_batch_obj(const csgpu_model *m, const csgpu_val *d_states_in, const csgpu_node *d_nodes,
                              csgpu_val *d_states_out, csgpu_result *d_results, int64_t batch, int32_t obj_lo,
                              int32_t obj_hi, void *stream);

/* Forbidden-set variant (pure binary-NE models with root intervals of at most 256 values).
 * Besides its interval every variable carries FW = csgpu_model_forbidden_words(m) 64-bit words:
 * bit k set <=> value root_lo+k is forbidden by a neighbour that is a single value.  The sets are
 * derived data that the search keeps next to each state so that a child inherits them:
 *   d_forb_in  [*][n_vars][FW] uint64, rows parallel to d_states_in  (NULL: rebuild from the state)
 *   d_forb_out [batch][n_vars][FW] uint64, rows parallel to d_states_out (NULL: not wanted)
 * Results (fixpoints, verdicts, PROPS of consistent nodes) are those of csgpu_propagate_batch.
 * Rows of inconsistent nodes (status -1) in d_states_out / d_forb_out are unspecified: the register-resident
 * kernel stores them unconditionally, the other kernels leave them untouched.  Bits of values outside a
 * variable's root domain are unspecified as well (a push may or may not be recorded there); they never
 * influence a result, and sets written by one kernel may be fed to another. */
int csgpu_model_forbidden_words(const csgpu_model *m); /* 0: the model does not qualify */
int csgpu_propagate_batch_fb(const csgpu_model *m, const csgpu_val *d_states_in, const uint64_t *d_forb_in,
                             const csgpu_node *d_nodes, csgpu_val *d_states_out, uint64_t *d_forb_out,
                             csgpu_result *d_results, int64_t batch, void *stream);

/* Three-valued evaluation of the root wide-and for a batch of states:
 * d_truth[i] = 1 (all clauses true), 0 (some clause false), 2 (undecided). */
int csgpu_eval_batch(const csgpu_model *m, const csgpu_val *d_states, int32_t *d_truth, int64_t batch,
                     void *stream);

/* Interval value of every clause (top-level constraint) for ONE state: the eval_<op>
 * entry points of the reference (eval.c:27-255) applied to each clause root.
 * d_vals: device, [n_clauses] csgpu_val. */
int csgpu_eval_clauses(const csgpu_model *m, const csgpu_val *d_state, csgpu_val *d_vals, void *stream);

/* ---- device-resident tree search (the batched analogue of solve(), csolve.c:398-476) ----
 *
 * A LIFO pool of open states lives in HBM.  One iteration pops the newest states, branches each
 * on its first open variable (every value of its interval becomes a child, as step_val
 * enumerates them, csolve.c:331-338), propagates all children in one csgpu_propagate_batch_obj
 * launch, counts the inconsistent ones as cuts, checks complete assignments with the root
 * evaluation (update_solution, csolve.c:222-244) and pushes the open survivors back.
 * The search tree is the reference's (same children, same propagation per child); the ORDER in
 * which it is walked is not, so CALLS/CUTS are engine-specific while the set of solutions and
 * the optimum are not.  Subtree sharding across GPUs moves whole states between the pools of
 * different ranks (csgpu_search_take / _put) and exchanges the incumbent (_set_best).
 * When the states carry one forbidden-set word per variable, a child whose value the parent's own
 * set already forbids (a valued neighbour rules it out, so its propagation can only fail) is counted
 * in `nodes` and `cuts` without being launched; CSGPU_SEARCH_HOLES=0 launches those too (same
 * counts, same solutions). */
typedef struct csgpu_search csgpu_search;

typedef struct csgpu_search_stats {
  uint64_t nodes;      /* children = CALLS */
  uint64_t cuts;       /* inconsistent children = CUTS */
  uint64_t props;      /* narrowing events = PROPS */
  uint64_t revisions;  /* clause revisions */
  uint64_t solutions;  /* accepted solutions (root evaluates to true) */
  uint64_t iterations; /* batched expand+propagate rounds */
  uint64_t restarts;   /* ANY only: Luby restarts (check_restart, csolve.c:264-276) */
  int64_t pool;        /* open states now in the pool */
  int64_t pool_peak;
  int32_t best;        /* incumbent (MIN/MAX), INT32_MAX / INT32_MIN if none yet */
  int32_t done;        /* 1: pool empty, or objective ANY and a solution was found */
} csgpu_search_stats;

/* pool_capacity: states the pool can hold; max_children: children propagated per iteration */
int csgpu_search_create(const csgpu_model *m, int64_t pool_capacity, int64_t max_children, csgpu_search **out);
void csgpu_search_free(csgpu_search *s);
/* forget everything (pool, statistics, incumbent, stored solutions, restart seeds) but keep the
 * buffers: the engine can run another search on the same model */
int csgpu_search_reset(csgpu_search *s);
/* append `count` states ([count][n_vars], device memory) to the pool */
int csgpu_search_put(csgpu_search *s, const csgpu_val *d_states, int64_t count);
/* the same from host memory (e.g. the root domains of csgpu_model_get_domains) */
int csgpu_search_put_host(csgpu_search *s, const csgpu_val *states, int64_t count);
/* remove up to `max` of the OLDEST states (the largest subtrees) into d_states; *count = how many */
int csgpu_search_take(csgpu_search *s, csgpu_val *d_states, int64_t max, int64_t *count);
/* cap on the open states expanded per iteration (default: as many as max_children allows for
 * ALL; 64 for ANY/MIN/MAX, which makes the walk depth-first enough to reach a first solution or a
 * good incumbent early with a small pool; MIN/MAX go up to 256 while the pool holds a backlog of
 * more than 16 x as many open states).  A value set here is taken literally. */
int csgpu_search_set_parents(csgpu_search *s, int64_t parents_per_iteration);
/* ANY only: restart from the seeded states after luby(i) x `iterations` iterations without a
 * solution (the reference restarts after luby(i) x restart_frequency failures, csolve.c:76-83,
 * 264-276, default 100); every restart tries the values in a different pseudo-random rotation.
 * Default 64; 0 disables restarts. */
int csgpu_search_set_restart(csgpu_search *s, int64_t iterations);
/* The reference's strategy options (src/main.c:51-130, src/strategy.c:79-121), to be set before the first state is put.
 * order: which open variable is branched on first -- 0 none, 1 smallest domain (default), 2 largest domain, 3 smallest
 * value, 4 largest value (-o); prefer_failing != 0: among equals the variable with the highest failure count (-f):
 * the branching variable's count goes down when an assignment holds and up when it fails, the variable whose domain
 * emptied goes up (csolve.c:455-465, propagate.c:33-41; the reference's further bumps along its recursion stack follow
 * its depth-first order and have no counterpart in a batch).  Ties: lowest index (the reference: heap order).
 * Anything but the default (1, 0) takes the separate-kernel path on interval rows; with failure counts the tree of
 * a search depends on the order in which batches finish. */
int csgpu_search_set_strategy(csgpu_search *s, int order, int prefer_failing);
/* MIN / MAX: a better solution restarts the search from the states it was seeded with, under the new bound
 * (update_solution + is_solution_restartable, csolve.c:216-219, 418-425).  Off by default. */
int csgpu_search_set_restart_on_improvement(csgpu_search *s, int on);
/* host time csgpu_search_put / put_host have taken since the last reset (device copy + rebuilding the forbidden
 * sets of the arriving states, which travel between ranks without them) and the states they brought */
int csgpu_search_put_cost(const csgpu_search *s, double *seconds, int64_t *states);
/* ---- helpers of the search driver -----------------------------------------------------------------------------
 * Host functions, no device needed.  The same code (csolve_amd/csrc/cs_arith.h) runs inside the kernels
 * (incumbent bound), the engine (restart schedule) and the drop-in (order of sibling values); exported so the
 * reference's unit vectors (test/test_objective.c, test/test_csolve.c:305-337,628-657) can be run against it.
 * `objective`: as returned by csgpu_model_objective (0 ANY 1 ALL 2 MIN 3 MAX). */
/* objective_better (objective.c:62-78): can objective value `value` still beat the incumbent `best`? */
int csgpu_objective_better(int objective, csgpu_val value, int32_t best);
/* objective_update_val (objective.c:101-126): `value` under the incumbent bound (MIN: hi <= best - 1) */
csgpu_val csgpu_objective_bound(int objective, csgpu_val value, int32_t best);
/* objective_update_best (objective.c:81-98): the incumbent after a solution with objective value `value` */
int32_t csgpu_objective_best(int objective, csgpu_val value, int32_t best);
/* fail_threshold_next (csolve.c:76-83): the next element of the Luby sequence 1 1 2 1 1 2 4 ... */
void csgpu_luby_next(uint64_t *threshold, uint64_t *counter);
/* step_check / step_val (csolve.c:323-338): is iteration `iter` over `bounds` valid, and the value it tries */
int csgpu_step_check(csgpu_val bounds, uint32_t iter);
int32_t csgpu_step_val(csgpu_val bounds, uint32_t iter, uint32_t seed);
/* strategy_var_cmp (strategy.c:79-121) as a key: the engine branches on the open variable with the SMALLEST key (what
 * the reference's heap has on top); order 0 none / 1 smallest domain / 2 largest domain / 3 smallest value / 4 largest
 * value, then (prefer_failing) the higher failure count, then the lower index.  sign(strategy_var_cmp(a, b)) =
 * sign(key(b) - key(a)) with the index bits masked (test/test_strategy.c VarCmp.*). */
uint64_t csgpu_branch_key(int order, int prefer_failing, csgpu_val value, int64_t prio, int32_t index);

/* merge an incumbent found elsewhere (objective_best of the shared page, objective.c:89-93) */
int csgpu_search_set_best(csgpu_search *s, int32_t best);
/* MIN / MAX engines of the same model on one device: from now on `s` keeps its incumbent in `with`'s word of
 * device memory (the analogue of the reference's shared page, csolve.c:86-97): what one engine accepts bounds
 * the very next fixpoints of the other.  If `with` borrows itself, `s` borrows from the same owner; an engine
 * that lends cannot borrow (CSGPU_E_STATE).  Lifetime: csgpu_search_free of a lender is deferred by the library
 * until its last borrower has been freed (the handle must not be used after the call all the same).
 * csgpu_search_reset of either resets the word.  Needs the device-driven iterations (CSGPU_E_STATE otherwise),
 * and while shared csgpu_search_set_parents refuses a setting that would leave them (CSGPU_E_STATE).
 * csgpu_search_best_solution answers 1 only on the engine whose stored row attains the current incumbent
 * (after csgpu_search_run / csgpu_search_set_best have brought its statistics up to date); the others answer 0.
 * With several engines accepting concurrently, node counts of a MIN / MAX run depend on timing; the optimum
 * does not.  A no-op for ANY / ALL. */
int csgpu_search_share_incumbent(csgpu_search *s, csgpu_search *with);
/* run up to max_iterations iterations (stops early when done).  ANY/MIN/MAX iterations are enqueued
 * sixteen at a time as one hipGraph with the bookkeeping between them on the device (pool top, child
 * counts, incumbent, stop conditions); the host reads the totals once per sixteen.  Environment, read
 * at csgpu_search_create: CSGPU_SEARCH_BURST=0 drives every iteration from the host,
 * CSGPU_SEARCH_GRAPH=0 enqueues the launches one by one instead of as a graph. */
int csgpu_search_run(csgpu_search *s, int64_t max_iterations, csgpu_search_stats *stats);
/* copy up to `max` stored solutions ([k][n_vars] values, host memory); returns k */
int64_t csgpu_search_solutions(const csgpu_search *s, int32_t *values, int64_t max);

/* MIN/MAX: the values ([n_vars], host memory) of a solution that attains the incumbent
 * (csgpu_search_stats.best); returns 1 if there is one, 0 if no solution was found yet */
int csgpu_search_best_solution(const csgpu_search *s, int32_t *values);

/* `count` values of variable `var` on ONE parent state, host buffers: node i assigns values[i].  results and
 * states_out ([count][n_vars]; rows of inconsistent nodes unspecified) are host memory.  What the reference
 * driver asks for one value at a time (step_val, csolve.c:331-338 -> check_assignment, csolve.c:247-261), in one
 * launch; the drop-in shim serves the driver's following calls from it.  Synchronous. */
int csgpu_propagate_values(const csgpu_model *m, const csgpu_val *state, int32_t var, const int32_t *values,
                           int32_t count, csgpu_val *states_out, csgpu_result *results);

/* Convenience for single nodes with host buffers (used by the drop-in shim):
 * uploads `state` (n_vars), runs one node, downloads the result.  Synchronous. */
int csgpu_propagate_one(const csgpu_model *m, const csgpu_val *state, csgpu_node node, csgpu_val *state_out,
                        csgpu_result *result);
/* The same with the node's trail: every narrowing the fixpoint made, as {variable, 0 = lower bound raised /
 * 1 = upper bound lowered / 2 = failure seen at this variable (-1: at a constant), new bound, clause}, `clause`
 * being the index of the top-level clause (csgpu_model_num_clauses order) whose revision made it -- what the
 * reference's bind() records as binding_t.clause (csolve.h:73-79) for its conflict analysis (conflict.c:290-316).
 * trace: host, [4 * cap] int32; *count = records made (may exceed cap: the rest is lost).  Records of one round
 * are in no particular order; replaying them in sequence (intersecting) gives the fixpoint.  Runs the general
 * kernel whatever csgpu_model_set_kernel says; the order of narrowings is the device's, not the reference's
 * depth-first one, so a conflict derived from the trail is valid but not necessarily the reference's. */
int csgpu_propagate_one_traced(const csgpu_model *m, const csgpu_val *state, csgpu_node node, csgpu_val *state_out,
                               csgpu_result *result, int32_t *trace, int32_t cap, int32_t *count);
/* The trail of one node of a pure != network with the CAUSE of every bound move as a variable: records
 * {variable, 0 = lower bound raised / 1 = upper bound lowered, new bound, the valued variable whose value forbade the
 * old bound}, in the order the moves were made (one wavefront: replaying them in sequence gives the fixpoint; the
 * first record that empties a domain is the failure).  *count = moves made; at most min(cap, 2048) records are kept
 * (a longer trail is cut off: the caller sees count above that).  Runs the interval-only shaving kernel (models that qualify
 * for kernel 7, CSGPU_E_LIMIT otherwise) -- the latency of csgpu_propagate_one.  What the drop-in needs to bump the
 * variables on the way from the assignment to a failure (propagate_term_recurse, propagate.c:44-54). */
int csgpu_propagate_one_causes(const csgpu_model *m, const csgpu_val *state, csgpu_node node, csgpu_val *state_out,
                               csgpu_result *result, int32_t *trace, int32_t cap, int32_t *count);
/* The reference's OWN failure chain of one node: which variables its depth-first propagation bumps when the node fails
 * -- the variable whose domain emptied (propagate_term_confl, propagate.c:33-41; none when the failure is found at the
 * constant of an `x + c` operand) and then every variable on its recursion stack, innermost first
 * (propagate_term_recurse, propagate.c:44-54) -- and how many narrowings it made before (its PROPS of the call).  One
 * wavefront walks the reference's recursion (cs_chain.hip.h); for models whose clauses are all NOT(EQ(l, r)) with l, r a
 * variable or `variable + constant` (CSGPU_E_LIMIT otherwise).  *status = -1 (failed) or 0; bumps[0 .. min(cap, *count))
 * in the reference's order.  Tens of microseconds per node: the drop-in uses it for failing nodes only, and only when
 * asked for the reference's exact trace (CSOLVE_DROPIN_CHAIN=reference). */
int csgpu_propagate_one_chain(const csgpu_model *m, const csgpu_val *state, csgpu_node node, int32_t *status,
                              int32_t *props, int32_t *bumps, int32_t cap, int32_t *count);
/* THE RESIDENT SERVER behind the two entries above.  One propagate_clauses of the reference's driver (csolve.c:247-261)
 * is one node; as a kernel launch it costs launch submit + dispatch + completion signal + the host's wait, 14 of the
 * 19 us of a call.  For the models of kernel 7 (pure != networks of at most 256 variables) csgpu_propagate_one and
 * csgpu_propagate_one_causes therefore talk to ONE resident wavefront through a mailbox in coherent host memory: the
 * host writes node record and state, then a request number; the wave polls it, runs the fixpoint, writes state, trail
 * and result back and acknowledges.  The wave leaves when the model is freed or after 2 ms without a request
 * (CSGPU_SERVER_IDLE_US) and is started again by the next call; CSGPU_SERVER=0 keeps the launch per call.  Results are
 * those of kernel 7 (same code).  csgpu_debug_one_timing: where the host's time of these calls went -- seconds[0..3] =
 * {copy in, ring + wait, copy out, server (re)starts} with the server, {copy in, launch submit, wait for completion,
 * copy out} without; calls made, servers started. */
int csgpu_debug_one_timing(const csgpu_model *m, double *seconds, uint64_t *calls, uint64_t *starts);

#ifdef __cplusplus
}
#endif
#endif
