/* cs_arith.h -- saturating int32 bound arithmetic and interval helpers.
 *
 * One source for host C (gcc), host C++ and gfx950 device code: every function
 * is `static inline` and carries CS_HD, which expands to `__host__ __device__`
 * under hipcc and to nothing under a plain C compiler.
 *
 * Semantics follow the reference's scalar layer (reference src/arith.c:27-85)
 * and the interval predicates of reference src/csolve.h:43-70:
 *   - CS_DOM_MIN / CS_DOM_MAX are absorbing -inf / +inf sentinels,
 *   - in cs_add -inf wins over +inf,
 *   - in cs_mul a sentinel times zero keeps the sentinel's sign rule
 *     (0 is treated as "not negative"),
 *   - everything else clamps to the int32 range by sign.
 * The code is a restatement through int64 clamping, not a transcription.
 */
#ifndef CS_ARITH_H
#define CS_ARITH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CS_HD __host__ __device__
#else
#define CS_HD
#endif

#define CS_DOM_MIN INT32_MIN
#define CS_DOM_MAX INT32_MAX

/* closed interval [lo,hi]; layout == reference `struct val_t` (csolve.h:43-46) */
typedef struct cs_val {
  int32_t lo;
  int32_t hi;
} cs_val;

CS_HD static inline int32_t cs_clamp64(int64_t x) {
  return x > (int64_t)CS_DOM_MAX ? CS_DOM_MAX : (x < (int64_t)CS_DOM_MIN ? CS_DOM_MIN : (int32_t)x);
}

/* reference arith.c:27-35 */
CS_HD static inline int32_t cs_neg(int32_t a) {
  return a == CS_DOM_MIN ? CS_DOM_MAX : (a == CS_DOM_MAX ? CS_DOM_MIN : -a);
}

/* reference arith.c:38-51 */
CS_HD static inline int32_t cs_add(int32_t a, int32_t b) {
  if (a == CS_DOM_MIN || b == CS_DOM_MIN) return CS_DOM_MIN;
  if (a == CS_DOM_MAX || b == CS_DOM_MAX) return CS_DOM_MAX;
  return cs_clamp64((int64_t)a + (int64_t)b);
}

/* reference arith.c:54-75 */
CS_HD static inline int32_t cs_mul(int32_t a, int32_t b) {
  if (a == CS_DOM_MIN) return b < 0 ? CS_DOM_MAX : CS_DOM_MIN;
  if (b == CS_DOM_MIN) return a < 0 ? CS_DOM_MAX : CS_DOM_MIN;
  if (a == CS_DOM_MAX) return b < 0 ? CS_DOM_MIN : CS_DOM_MAX;
  if (b == CS_DOM_MAX) return a < 0 ? CS_DOM_MIN : CS_DOM_MAX;
  return cs_clamp64((int64_t)a * (int64_t)b);
}

/* reference arith.c:78-85 */
CS_HD static inline int32_t cs_min(int32_t a, int32_t b) { return a < b ? a : b; }
CS_HD static inline int32_t cs_max(int32_t a, int32_t b) { return a > b ? a : b; }

CS_HD static inline cs_val cs_interval(int32_t lo, int32_t hi) {
  cs_val v;
  v.lo = lo;
  v.hi = hi;
  return v;
}
CS_HD static inline cs_val cs_value(int32_t x) { return cs_interval(x, x); }

/* reference csolve.h:57-67 */
CS_HD static inline int cs_is_value(cs_val v) { return v.lo == v.hi; }
CS_HD static inline int cs_is_true(cs_val v) { return v.lo > 0 || v.hi < 0; }
CS_HD static inline int cs_is_false(cs_val v) { return v.lo == 0 && v.hi == 0; }

/* any bound sitting on a sentinel: comparisons give up (eval.c:47-50, 81-84) */
CS_HD static inline int cs_unbounded2(cs_val a, cs_val b) {
  return a.lo == CS_DOM_MIN || a.hi == CS_DOM_MAX || b.lo == CS_DOM_MIN || b.hi == CS_DOM_MAX;
}
CS_HD static inline int cs_is_sentinel(int32_t x) { return x == CS_DOM_MIN || x == CS_DOM_MAX; }

/* three-valued truth as an interval */
CS_HD static inline cs_val cs_tv(int must_true, int must_false) {
  return must_true ? cs_value(1) : (must_false ? cs_value(0) : cs_interval(0, 1));
}

/* ---- interval evaluation of one operator from child intervals ---- */

/* reference eval.c:32-63 */
CS_HD static inline cs_val cs_ev_eq(cs_val a, cs_val b) {
  if (cs_unbounded2(a, b)) return cs_interval(0, 1);
  return cs_tv(a.lo == a.hi && a.lo == b.lo && a.hi == b.hi, a.hi < b.lo || a.lo > b.hi);
}
/* reference eval.c:66-97 */
CS_HD static inline cs_val cs_ev_lt(cs_val a, cs_val b) {
  if (cs_unbounded2(a, b)) return cs_interval(0, 1);
  return cs_tv(a.hi < b.lo, a.lo >= b.hi);
}
/* reference eval.c:100-114 */
CS_HD static inline cs_val cs_ev_neg(cs_val a) { return cs_interval(cs_neg(a.hi), cs_neg(a.lo)); }
/* reference eval.c:117-135 */
CS_HD static inline cs_val cs_ev_add(cs_val a, cs_val b) {
  return cs_interval(cs_add(a.lo, b.lo), cs_add(a.hi, b.hi));
}
/* reference eval.c:138-160 */
CS_HD static inline cs_val cs_ev_mul(cs_val a, cs_val b) {
  int32_t p0 = cs_mul(a.lo, b.lo), p1 = cs_mul(a.lo, b.hi);
  int32_t p2 = cs_mul(a.hi, b.lo), p3 = cs_mul(a.hi, b.hi);
  return cs_interval(cs_min(cs_min(p0, p1), cs_min(p2, p3)), cs_max(cs_max(p0, p1), cs_max(p2, p3)));
}
/* reference eval.c:163-180 */
CS_HD static inline cs_val cs_ev_not(cs_val a) { return cs_tv(cs_is_false(a), cs_is_true(a)); }
/* reference eval.c:183-205 (short-circuit order does not change the value) */
CS_HD static inline cs_val cs_ev_and(cs_val a, cs_val b) {
  return cs_tv(cs_is_true(a) && cs_is_true(b), cs_is_false(a) || cs_is_false(b));
}
/* reference eval.c:208-230 */
CS_HD static inline cs_val cs_ev_or(cs_val a, cs_val b) {
  return cs_tv(cs_is_true(a) || cs_is_true(b), cs_is_false(a) && cs_is_false(b));
}

/* ---- helpers of the search driver shared by the device kernels, the engine and the drop-in ------------------
 * (pinned by the reference's own unit vectors: tests/golden/ref_unit_objective.json, ref_unit_search.json) */

/* objective_better (objective.c:62-78): can a node whose objective value is `d` still beat the incumbent?
 * sense: 0 = ANY / ALL (always), 1 = minimise, 2 = maximise */
CS_HD static inline int cs_objective_better(int sense, cs_val d, int32_t best) {
  return sense == 1 ? d.lo < best : (sense == 2 ? d.hi > best : 1);
}

/* objective_update_val (objective.c:101-126): the objective value under the incumbent bound */
CS_HD static inline cs_val cs_objective_bound(int sense, cs_val d, int32_t best) {
  if (sense == 1) {
    const int32_t h = cs_add(best, cs_neg(1));
    if (d.hi > h) d.hi = h;
  } else if (sense == 2) {
    const int32_t l = cs_add(best, 1);
    if (d.lo < l) d.lo = l;
  }
  return d;
}

/* objective_update_best (objective.c:81-98): the incumbent after a solution with objective value `d` */
CS_HD static inline int32_t cs_objective_best(int sense, cs_val d, int32_t best) {
  return sense == 1 ? d.lo : (sense == 2 ? d.hi : best);
}

/* fail_threshold_next (csolve.c:76-83): Knuth's formulation of the Luby sequence 1 1 2 1 1 2 4 ... */
static inline void cs_luby_next(uint64_t *threshold, uint64_t *counter) {
  if ((*counter & (0 - *counter)) == *threshold) {
    (*counter)++;
    *threshold = 1;
  } else {
    *threshold <<= 1;
  }
}

/* step_check / step_val (csolve.c:323-338): iteration `iter` of a variable with the interval `bounds` is valid
 * while iter <= hi - lo; its value walks in from the edges, the parity of `seed` deciding which edge is first */
CS_HD static inline int cs_step_check(cs_val bounds, uint32_t iter) {
  return iter <= (uint32_t)bounds.hi - (uint32_t)bounds.lo;
}
CS_HD static inline int32_t cs_step_val(cs_val bounds, uint32_t iter, uint32_t seed) {
  return ((iter ^ seed) & 1u) ? (int32_t)((uint32_t)bounds.hi - (iter >> 1)) : (int32_t)((uint32_t)bounds.lo + (iter >> 1));
}

/* The branching rule's key (strategy_var_cmp, reference src/strategy.c:79-121): the variable with the SMALLEST key is
 * branched on first -- what the reference keeps at the top of its heap.  order: 0 none, 1 smallest domain, 2 largest
 * domain, 3 smallest value, 4 largest value; then, with prefer_failing, the higher failure count; the caller breaks
 * remaining ties by index.  Domain sizes saturate like the reference's add() does (arith.c:38-51): an interval with a
 * bound at DOMAIN_MIN / DOMAIN_MAX has the largest size there is. */
CS_HD static inline uint32_t cs_order_key(int order, cs_val d) {
  const int unbounded = d.lo == CS_DOM_MIN || d.hi == CS_DOM_MAX || d.lo == CS_DOM_MAX || d.hi == CS_DOM_MIN;
  const int64_t w = (int64_t)d.hi - (int64_t)d.lo;
  const uint32_t size = unbounded || w > 0x7ffffffell ? 0x7fffffffu : (w < 0 ? 0u : (uint32_t)w); /* width - 1, saturated */
  switch (order) {
  case 0: return 0u;
  case 1: return size;
  case 2: return 0x7fffffffu - size;
  case 3: return (uint32_t)d.lo ^ 0x80000000u;               /* lower bound, as an unsigned rank */
  case 4: return 0xffffffffu - ((uint32_t)d.hi ^ 0x80000000u); /* higher upper bound first */
  default: return size;
  }
}
CS_HD static inline uint64_t cs_branch_key_of(int order, int prefer_failing, cs_val d, int64_t prio, int index) {
  uint32_t pk = 0u;
  if (prefer_failing) {
    int64_t p = prio + 32768;
    p = p < 0 ? 0 : (p > 65535 ? 65535 : p);
    pk = 65535u - (uint32_t)p;
  }
  return ((uint64_t)cs_order_key(order, d) << 32) | ((uint64_t)pk << 16) | (uint64_t)((uint32_t)index & 0xffffu);
}

#endif /* CS_ARITH_H */
