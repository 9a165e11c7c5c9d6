/* csolve_dropin.h -- the csolve.h-named entry points, served by the GPU.
 *
 * libcsolve_dropin.so exports, with C linkage and the reference's prototypes
 * (reference src/csolve.h:275-284, 341-349, 360-362), the symbols the reference's search
 * driver and front end call into the propagation layer:
 *
 *     domain_t neg/add/mul/min/max                           (replaces src/arith.c)
 *     struct val_t eval_<op>(const struct constr_t *)          (replaces src/eval.c)
 *     prop_result_t propagate_<op>(struct constr_t *, struct val_t, const struct wand_expr_t *)
 *     prop_result_t propagate(struct constr_t *, size_t)       (replaces src/propagate.c)
 *     prop_result_t propagate_clauses(const struct clause_list_t *)
 *        <op> in { term eq lt neg add mul not and or wand confl }
 *
 * A maintainer links the reference's remaining objects (csolve.c strategy.c objective.c
 * util.c normalize.c conflict.c constr_types.c parser*.c print.c stats.c main.c) against this
 * library instead of arith.o eval.o propagate.o.  The types below are declared here only so
 * that this header is self-contained; they are layout-compatible with the reference's
 * (reference src/csolve.h:43-46, 92-96, 105-130, 165-170, 225-239) and the shim is compiled
 * against THIS header, never against the reference's.
 *
 * Symbols the shim needs from the driver (all exist in the reference):
 *     void bind(struct env_t *, struct val_t, const struct wand_expr_t *)   util.c:137
 *     void conflict_reset(void)                                             conflict.c:131
 *     extern uint64_t props                                                 stats.c (STAT_LIST)
 *     void print_fatal(const char *fmt, ...)                                print.c:88
 */
#ifndef CSOLVE_DROPIN_H
#define CSOLVE_DROPIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t domain_t;
typedef int32_t prop_result_t;
typedef uint64_t prop_tag_t;

struct val_t {
  domain_t lo, hi;
};

struct constr_t;
struct env_t;

struct wand_expr_t {
  struct constr_t *constr;
  struct constr_t *orig;
  prop_tag_t prop_tag;
};

struct confl_elem_t {
  struct val_t val;
  struct constr_t *var;
};

struct constr_type_t {
  struct val_t (*eval)(const struct constr_t *);
  prop_result_t (*prop)(struct constr_t *, const struct val_t, const struct wand_expr_t *);
  struct constr_t *(*norm)(struct constr_t *);
  int op; /* enum operator_t: ' ' '=' '<' '-' '+' '*' '!' '&' '|' 'A' 'C' (csolve.h:133-162) */
};

struct constr_t {
  const struct constr_type_t *type;
  union {
    struct { struct val_t val; struct env_t *env; } term;
    struct { struct constr_t *l, *r; } expr;
    struct { size_t length; struct wand_expr_t *elems; } wand;
    struct { size_t length; struct confl_elem_t *elems; } confl;
  } constr;
};

struct clause_list_t {
  size_t length;
  struct wand_expr_t **elems;
};

struct env_t {
  const char *key;
  struct constr_t *val;
  void *binds;
  struct clause_list_t clauses;
  size_t order;
  int64_t prio;
  size_t level;
};

/* replaces reference src/arith.c:27-85 */
domain_t neg(domain_t a);
domain_t add(domain_t a, domain_t b);
domain_t mul(domain_t a, domain_t b);
domain_t min(domain_t a, domain_t b);
domain_t max(domain_t a, domain_t b);

/* replaces reference src/propagate.c:474-485 and 488-538 */
prop_result_t propagate(struct constr_t *constr, size_t limit);
prop_result_t propagate_clauses(const struct clause_list_t *clauses);

#define CSOLVE_DROPIN_OPS(F) F(term) F(eq) F(lt) F(neg) F(add) F(mul) F(not) F(and) F(or) F(wand) F(confl)
/* replaces reference src/eval.c:27-277 and src/propagate.c:57-471 */
#define CSOLVE_DROPIN_DECL(NAME)                                                                   \
  struct val_t eval_##NAME(const struct constr_t *constr);                                         \
  prop_result_t propagate_##NAME(struct constr_t *constr, struct val_t val, const struct wand_expr_t *clause);
CSOLVE_DROPIN_OPS(CSOLVE_DROPIN_DECL)

/* Optional explicit attach (otherwise done lazily at the first propagate_clauses call from
 * the root last passed to propagate()): env = env_generate(), size = var_count(), root = the
 * normalised root wide-and, i.e. the arguments of solve() (parser.y:81-86). */
int csolve_dropin_attach(struct env_t *env, size_t size, struct constr_t *root);
void csolve_dropin_detach(void);
/* device calls made so far: [0] propagate_clauses, [1] propagate, [2] eval, [3] single-op propagate */
void csolve_dropin_counters(uint64_t out[4]);
/* sibling batching of propagate_clauses: [0] batches launched, [1] calls served from a batch already there */
void csolve_dropin_sibling_counters(uint64_t out[2]);
/* conflict learning (driver run with -c true): [0] failing nodes handed to the driver's conflict_create with
 * their trail, [1] re-attachments after the driver's clause lists had grown by learnt clauses */
void csolve_dropin_learning_counters(uint64_t out[2]);
/* seconds spent so far in [0] attach, [1] the device calls of propagate_clauses, [2] the rest of propagate_clauses */
void csolve_dropin_seconds(double out[3]);
/* seconds spent inside the eval entry points (update_solution evaluates the root once per complete assignment) */
double csolve_dropin_eval_seconds(void);
/* failing nodes walked in the reference's own order (CSOLVE_DROPIN_CHAIN=reference) */
uint64_t csolve_dropin_reference_walks(void);
/* device time of the propagate_clauses calls so far, in microseconds: { first call, median, 90th percentile, maximum } */
void csolve_dropin_call_times(double out[4]);

#ifdef __cplusplus
}
#endif
#endif
