/* cs_kernels.hip.h -- gfx950 kernels of the constraint-propagation fixpoint.
 *
 * Written for CDNA4 only: 64-lane wavefronts, LDS-resident node state, wave-level
 * ballots.  No MFMA: the work is integer compare / select / LDS atomics.
 *
 * Kernels in this file (all compute the batched propagate_clauses fixpoint with the same results):
 *   cs_propagate_events   (1) any model: binary relations inline, other clauses through the tree interpreter
 *   cs_propagate_ne_lds   (2) pure binary-!= models, adjacency in LDS, unit shaving
 *   cs_propagate_ne_bitset(3) the same models with forbidden sets per variable in LDS
 *   cs_propagate_ne_regs  (4) models of at most 256 variables: sets and bounds in registers (the bench kernel),
 *                             also with the states carried as the sets alone
 *   cs_propagate_ne_packed(5) kernel 4 for at most 32 variables: two or four nodes per wave
 *   cs_propagate_clause_rounds (6) at most 512 clauses, resident in registers: all revised per round
 *   cs_propagate_ne_shave (7, cs_shave.hip.h) kernel 4's models on interval states alone: the bench kernel
 *   cs_propagate_sweeps, cs_eval_root, cs_eval_clauses, cs_sets_unpack: root phase, evaluation, layout helper
 *
 * Execution model of the general kernel (cs_propagate_events):
 *   - one WAVEFRONT owns one search node; a 256-thread workgroup is four independent
 *     nodes, there is no __syncthreads() anywhere in the kernel;
 *   - the node's interval domains (n_vars x {lo,hi}, 8 B each, the reference's
 *     `struct val_t`) live in that wave's slice of LDS for the whole fixpoint and are
 *     read from / written to HBM exactly once, 8 B per lane, fully coalesced;
 *   - the worklist is a bit mask of changed variables in LDS (two buffers, swapped per
 *     round); the wave walks the set bits with scalar code, and for each changed
 *     variable its 64 lanes stride over that variable's adjacency list (the reference's
 *     per-variable clause list), revising 64 clauses per step;
 *   - narrowing is ds_max_rtn_i32 / ds_min_rtn_i32 on the bounds (LDS atomics), so several
 *     clauses may tighten one variable in the same step; the returned old value tells
 *     which lane actually narrowed (that lane counts the propagation and sets the
 *     variable's bit for the next round);
 *   - failure (an empty interval) is detected by the narrowing lane when it can see it
 *     and, for racing lo/hi updates, when the variable is popped in the next round.
 *
 * Reference semantics restated here (one clause revision):
 *   propagate_clauses   propagate.c:488-538   -> rounds over the changed-variable mask
 *   propagate_term      propagate.c:57-87     -> cs_ctx::narrow
 *   propagate_eq/lt/neg/add/mul/not/and/or/wand  propagate.c:139-392 -> cs_tree_revise
 *   eval_*              eval.c:27-255         -> cs_tree_eval (+ cs_arith.h)
 *   NOT(EQ(x+k, y+l))   propagate.c:289-301,123-136,106-120,223-246 -> cs_ne_revise
 *   x = y+c, x < y+c, not(x < y), (a < b+c) or (d < e+f)      propagate.c:139-246, 289-340 -> cs_lin_revise, cs_or2_revise
 * The revision ORDER differs from the reference (parallel rounds instead of depth-first
 * recursion); the fixpoint and the consistent/inconsistent verdict do not.
 */
#ifndef CS_KERNELS_HIP_H
#define CS_KERNELS_HIP_H

#include <hip/hip_runtime.h>

#include "cs_arith.h"
#include "cs_device.h"
#include "cs_frontend.h"

#define CS_WAVE 64
#define CS_WAVES_PER_BLOCK 4
#define CS_BLOCK (CS_WAVE * CS_WAVES_PER_BLOCK)

struct cs_tables {
  int n_vars, n_clauses, n_words; /* n_words = ceil(n_vars / 32) */
  const int *adj_off;
  const int2 *adj;
  const int4 *clause;
  const int4 *clause_by_kind; /* the same records sorted by kind (kernel 6: the 64 lanes of a slot then mostly run one path) */
  const int *tree_off;
  const int4 *tnode;
  const int *tkid;
  const int2 *tree_want;
  const int4 *lit; /* literals {a, b, d, 0} of the two-literal disjunctions: X_a < X_b + d */
  int n_lits;
  /* objective bound applied to every node before it is propagated (objective_update_val,
   * reference src/objective.c:101-126): dom[obj_var] is intersected with [obj_lo, obj_hi] */
  int obj_var, obj_lo, obj_hi;
  /* the same bound taken from the incumbent in device memory when the kernel starts (the search engine's
   * device-driven iterations): sense 1 = minimise (hi = best - 1), 2 = maximise (lo = best + 1) */
  const int *obj_best_dev;
  int obj_sense;
  /* trace of one node (csgpu_propagate_one_traced; kernel 1 with TRACE only): the clause behind every adjacency
   * entry, and a log of the narrowings {variable, 0 = lower / 1 = upper bound / 2 = failure, new bound, clause} */
  const int *adj_clause;
  int4 *trace_log;
  unsigned *trace_n;
  unsigned trace_cap;
};

struct cs_node_in {
  int var, lo, hi, parent;
};
struct cs_node_out {
  int status, props, revisions, rounds;
};

/* compiler-level ordering between LDS accesses of different lanes of one wave: the
 * hardware already executes a wave's DS instructions in order */
__device__ __forceinline__ void cs_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* per-lane view of the node a wave (or block) is working on */
struct cs_ctx {
  cs_val *dom;        /* LDS, [n_vars] */
  unsigned *mark;     /* LDS, next-round changed mask (events) or one flag word (sweeps) */
  int mark_is_flag;
  int fail, props, revisions;
  int fail_var; /* a variable whose domain this lane saw become empty (-1: none / not attributable to one variable):
                 * the reference bumps that variable's priority (propagate_term_confl, propagate.c:33-41) */
  /* trace (null unless the kernel was built with TRACE): which clause is being revised, and where narrowings are
   * recorded -- what the reference keeps as binding_t.clause on its trail (csolve.h:73-79) */
  int4 *log;
  unsigned *log_n;
  unsigned log_cap;
  int cur_clause;
  __device__ __forceinline__ void record(int v, int kind, int bound) {
    if (log != nullptr) {
      const unsigned i = atomicAdd(log_n, 1u);
      if (i < log_cap) log[i] = make_int4(v, kind, bound, cur_clause);
    }
  }
  __device__ __forceinline__ void failed_at(int v) {
    if (!fail) record(v, 2, 0);
    fail = 1;
    fail_var = v;
  }

  __device__ __forceinline__ void touched(int v) {
    if (mark_is_flag) mark[0] = 1u;
    else atomicOr(&mark[v >> 5], 1u << (v & 31));
  }
  __device__ __forceinline__ void raise_lo(int v, int lo) {
    int old = atomicMax(&dom[v].lo, lo);
    if (old < lo) {
      props++;
      record(v, 0, lo);
      touched(v);
      if (lo > dom[v].hi) failed_at(v);
    }
  }
  __device__ __forceinline__ void lower_hi(int v, int hi) {
    int old = atomicMin(&dom[v].hi, hi);
    if (old > hi) {
      props++;
      record(v, 1, hi);
      touched(v);
      if (hi < dom[v].lo) failed_at(v);
    }
  }
  /* propagate_term (propagate.c:57-87): intersect dom[v] with want */
  __device__ __forceinline__ void narrow(int v, cs_val want) {
    cs_val d = dom[v];
    if (d.lo > want.hi || d.hi < want.lo) { failed_at(v); return; }
    int before = props;
    if (want.lo > d.lo) raise_lo(v, want.lo);
    if (want.hi < d.hi) lower_hi(v, want.hi);
    /* the reference binds lo and hi in one step: count one propagation */
    if (props == before + 2) props = before + 1;
  }
};

/* the variable a failed node reports in csgpu_result.rounds: the first lane that attributed its failure to a
 * variable, else a variable whose bounds have crossed in `dom`, else -1 */
__device__ __forceinline__ int cs_wave_fail_var(const cs_ctx &cx, const cs_val *dom, int n, int lane) {
  const unsigned long long m = __ballot(cx.fail && cx.fail_var >= 0);
  if (m != 0ull) return __builtin_amdgcn_readlane(cx.fail_var, __builtin_ctzll(m));
  for (int v0 = 0; v0 < n; v0 += CS_WAVE) {
    const int v = v0 + lane;
    const unsigned long long c = __ballot(v < n && dom[v < n ? v : 0].lo > dom[v < n ? v : 0].hi);
    if (c != 0ull) return v0 + __builtin_ctzll(c);
  }
  return -1;
}

/* X_u != X_w + d seen from u (both directions of the clause, propagate.c:123-136) */
/* X_a < X_b + d pushed down to the two variables (propagate_lt true + propagate_add, propagate.c:139-200,
 * 223-246): hi_a <= hi_b + d - 1, lo_b >= lo_a - d + 1.  Bounds are not sentinels and |d| < 2^30
 * (cs_device.h), the arithmetic is done in 64 bits: a bound that would leave the int32 range cannot
 * narrow anything, which is also what the saturating tree path makes of it. */
__device__ __forceinline__ void cs_lt_enforce(cs_ctx &cx, int a, int b, int d) {
  const cs_val da = cx.dom[a], db = cx.dom[b];
  const long long ha = (long long)db.hi + d - 1, lb = (long long)da.lo - d + 1;
  if (ha < da.lo || lb > db.hi) { cx.failed_at(ha < da.lo ? a : b); return; }
  if (ha < da.hi) cx.lower_hi(a, (int)ha);
  if (lb > db.lo) cx.raise_lo(b, (int)lb);
}

__device__ __forceinline__ cs_val cs_shifted(cs_val v, int d) { /* [v.lo + d, v.hi + d] clamped to int32 */
  const long long lo = (long long)v.lo + d, hi = (long long)v.hi + d;
  return cs_interval(lo < CS_DOM_MIN ? CS_DOM_MIN : (lo > CS_DOM_MAX ? CS_DOM_MAX : (int)lo),
                     hi < CS_DOM_MIN ? CS_DOM_MIN : (hi > CS_DOM_MAX ? CS_DOM_MAX : (int)hi));
}

/* three-valued X_a < X_b + d and X_a = X_b + d in 64 bits (no bound is a sentinel on these paths; the shift by
 * d, which for a negated LT is an artefact of the normal form, must not create one) */
__device__ __forceinline__ cs_val cs_ev_lt_shifted(cs_val a, cs_val b, int d) {
  return cs_tv((long long)a.hi < (long long)b.lo + d, (long long)a.lo >= (long long)b.hi + d);
}
__device__ __forceinline__ cs_val cs_ev_eq_shifted(cs_val a, cs_val b, int d) {
  return cs_tv(a.lo == a.hi && (long long)a.lo == (long long)b.lo + d && (long long)a.hi == (long long)b.hi + d,
               (long long)a.hi < (long long)b.lo + d || (long long)a.lo > (long long)b.hi + d);
}

/* binary relation seen from u: rel 0  X_u != X_w + d, 1  X_u = X_w + d, 2  X_u < X_w + d, 3  X_u > X_w + d */
__device__ __forceinline__ void cs_ne_revise(cs_ctx &cx, int u, int w, int d);
__device__ __forceinline__ void cs_lin_revise(cs_ctx &cx, int u, int w, int d, int rel) {
  if (rel == CS_REL_NE) { cs_ne_revise(cx, u, w, d); return; }
  cx.revisions++;
  if (rel == CS_REL_EQ) { /* propagate_eq true: each side into the other's interval */
    const cs_val du = cx.dom[u], dw = cx.dom[w];
    cx.narrow(u, cs_shifted(dw, d));
    if (!cx.fail) cx.narrow(w, cs_shifted(du, -d));
  } else if (rel == CS_REL_LT) {
    cs_lt_enforce(cx, u, w, d);
  } else { /* X_u > X_w + d  <=>  X_w < X_u - d */
    cs_lt_enforce(cx, w, u, -d);
  }
}

/* lit[0] or lit[1] wanted true (propagate_or, propagate.c:318-340): a literal that cannot hold any more
 * forces the other one; a literal is X_a < X_b + d */
__device__ __forceinline__ void cs_or2_revise(cs_ctx &cx, const int4 *lit) {
  cx.revisions++;
  const int4 l0 = lit[0], l1 = lit[1];
  const bool f0 = (long long)cx.dom[l0.x].lo >= (long long)cx.dom[l0.y].hi + l0.z; /* X_a >= X_b + d for sure */
  const bool f1 = (long long)cx.dom[l1.x].lo >= (long long)cx.dom[l1.y].hi + l1.z;
  if (f0 && f1) cx.failed_at(l1.x);
  else if (f0) cs_lt_enforce(cx, l1.x, l1.y, l1.z);
  else if (f1) cs_lt_enforce(cx, l0.x, l0.y, l0.z);
}

__device__ __forceinline__ void cs_ne_revise(cs_ctx &cx, int u, int w, int d) {
  cs_val du = cx.dom[u], dw = cx.dom[w];
  cx.revisions++;
  if (du.lo == du.hi) { /* u is a value: w must avoid f */
    int f = du.lo - d;
    if (dw.lo == f) cx.raise_lo(w, f + 1);
    else if (dw.hi == f) cx.lower_hi(w, f - 1);
  }
  if (dw.lo == dw.hi) { /* w is a value: u must avoid f */
    int f = dw.lo + d;
    if (du.lo == f) cx.raise_lo(u, f + 1);
    else if (du.hi == f) cx.lower_hi(u, f - 1);
  }
}

/* ---- general expression trees ------------------------------------------------ */

/* bottom-up interval evaluation of every node of one tree (eval.c:27-255) */
__device__ inline void cs_tree_eval(const cs_tables &T, const int4 *nd, int len, const cs_val *dom, cs_val *val) {
  for (int k = 0; k < len; k++) {
    int4 n = nd[k];
    cs_val r;
    switch (n.x) {
    case CS_OP_VAR: r = dom[n.y]; break;
    case CS_OP_CONST: r = cs_interval(n.y, n.z); break;
    case CS_OP_EQ: r = cs_ev_eq(val[n.y], val[n.z]); break;
    case CS_OP_LT: r = cs_ev_lt(val[n.y], val[n.z]); break;
    case CS_OP_NEG: r = cs_ev_neg(val[n.y]); break;
    case CS_OP_ADD: r = cs_ev_add(val[n.y], val[n.z]); break;
    case CS_OP_MUL: r = cs_ev_mul(val[n.y], val[n.z]); break;
    case CS_OP_NOT: r = cs_ev_not(val[n.y]); break;
    case CS_OP_AND: r = cs_ev_and(val[n.y], val[n.z]); break;
    case CS_OP_OR: r = cs_ev_or(val[n.y], val[n.z]); break;
    case CS_OP_WAND: { /* eval.c:233-255 */
      int any_false = 0, all_true = 1;
      for (int i = 0; i < n.z; i++) {
        cs_val c = val[T.tkid[n.y + i]];
        any_false |= cs_is_false(c);
        all_true &= cs_is_true(c);
      }
      r = cs_tv(all_true && !any_false, any_false);
      break;
    }
    case CS_OP_CONFL: { /* eval.c:258-277: the first element that is not a value answers "unknown", the first value
                         * different from its conflict value answers "true" */
      r = cs_interval(0, 1);
      for (int i = 0; i < n.z; i++) {
        const cs_val c = val[T.tkid[n.y + 2 * i]];
        if (!cs_is_value(c)) break;
        if (c.lo != T.tkid[n.y + 2 * i + 1]) {
          r = cs_value(1);
          break;
        }
      }
      break;
    }
    default: r = cs_interval(0, 1); break;
    }
    val[k] = r;
  }
}

struct cs_tree_scratch {
  cs_val val[CS_MAX_TREE_NODES];
  cs_val want[CS_MAX_TREE_NODES];
  short node[CS_MAX_TREE_NODES];
};

/* One revision of a tree clause: evaluate every node, then push `true` down from the
 * root (propagate.c:139-392).  Sibling values are the bottom-up ones (not re-evaluated
 * after the first child narrowed, as propagate.c:98,167,187,242 do): a weaker single
 * revision with the same fixpoint, because the clause is revised again whenever one of
 * its variables changed. */
__device__ inline void cs_tree_revise(const cs_tables &T, int tree, cs_ctx &cx, cs_tree_scratch &S) {
  const int base = T.tree_off[tree];
  const int len = T.tree_off[tree + 1] - base;
  const int4 *nd = T.tnode + base;
  cx.revisions++;
  cs_tree_eval(T, nd, len, cx.dom, S.val);

  int sp = 0;
#define CS_PUSH(K, W)                                                          \
  do {                                                                         \
    if (sp < CS_MAX_TREE_NODES) { S.node[sp] = (short)(K); S.want[sp] = (W); sp++; } \
  } while (0)
  {
    const int2 rw = T.tree_want[tree];
    CS_PUSH(len - 1, cs_interval(rw.x, rw.y));
  }
  while (sp > 0 && !cx.fail) {
    sp--;
    const int k = S.node[sp];
    const cs_val w = S.want[sp];
    const int4 n = nd[k];
    switch (n.x) {
    case CS_OP_VAR:
      cx.narrow(n.y, w);
      break;
    case CS_OP_CONST: /* terminal without variable: only a conflict is observable */
      if (n.y > w.hi || n.z < w.lo) cx.fail = 1;
      break;
    case CS_OP_EQ: {
      cs_val lv = S.val[n.y], rv = S.val[n.z];
      if (cs_is_true(w)) { /* propagate.c:90-103 */
        CS_PUSH(n.z, lv);
        CS_PUSH(n.y, rv);
      } else if (cs_is_false(w)) { /* propagate.c:106-136 */
        if (cs_is_value(lv) && !cs_is_sentinel(lv.lo)) {
          if (lv.lo == rv.lo) CS_PUSH(n.z, cs_interval(lv.lo + 1, CS_DOM_MAX));
          else if (lv.lo == rv.hi) CS_PUSH(n.z, cs_interval(CS_DOM_MIN, lv.lo - 1));
        }
        if (cs_is_value(rv) && !cs_is_sentinel(rv.lo)) {
          if (rv.lo == lv.lo) CS_PUSH(n.y, cs_interval(rv.lo + 1, CS_DOM_MAX));
          else if (rv.lo == lv.hi) CS_PUSH(n.y, cs_interval(CS_DOM_MIN, rv.lo - 1));
        }
      }
      break;
    }
    case CS_OP_LT: {
      cs_val lv = S.val[n.y], rv = S.val[n.z];
      if (cs_is_true(w)) { /* propagate.c:155-176 */
        if (!cs_is_sentinel(lv.lo)) CS_PUSH(n.z, cs_interval(lv.lo + 1, CS_DOM_MAX));
        if (!cs_is_sentinel(rv.hi)) CS_PUSH(n.y, cs_interval(CS_DOM_MIN, rv.hi - 1));
      } else if (cs_is_false(w)) { /* propagate.c:179-192 */
        CS_PUSH(n.z, cs_interval(CS_DOM_MIN, lv.hi));
        CS_PUSH(n.y, cs_interval(rv.lo, CS_DOM_MAX));
      }
      break;
    }
    case CS_OP_NEG: /* propagate.c:211-220 */
      CS_PUSH(n.y, cs_interval(cs_neg(w.hi), cs_neg(w.lo)));
      break;
    case CS_OP_ADD: { /* propagate.c:223-246 */
      cs_val lv = S.val[n.y], rv = S.val[n.z];
      CS_PUSH(n.z, cs_interval(cs_add(w.lo, cs_neg(lv.hi)), cs_add(w.hi, cs_neg(lv.lo))));
      CS_PUSH(n.y, cs_interval(cs_add(w.lo, cs_neg(rv.hi)), cs_add(w.hi, cs_neg(rv.lo))));
      break;
    }
    case CS_OP_MUL: { /* propagate.c:249-286 */
      if (w.lo != CS_DOM_MIN && w.hi != CS_DOM_MIN) {
        for (int side = 0; side < 2; side++) {
          const int p = side == 0 ? n.z : n.y;
          const cs_val cv = S.val[side == 0 ? n.y : n.z];
          if (!cs_is_value(cv)) continue;
          const int c = cv.lo;
          if (((w.lo > 0 || w.hi < 0) && c == 0) || (cs_is_value(w) && c != 0 && (w.lo % c) != 0)) {
            cx.fail = 1;
            break;
          }
          if (c != 0) {
            int a = w.lo / c, b = w.hi / c;
            CS_PUSH(p, cs_interval(cs_min(a, b), cs_max(a, b)));
          }
        }
      }
      break;
    }
    case CS_OP_NOT: /* propagate.c:289-301 */
      if (cs_is_true(w)) CS_PUSH(n.y, cs_value(0));
      else if (cs_is_false(w)) CS_PUSH(n.y, cs_value(1));
      break;
    case CS_OP_AND: { /* propagate.c:343-358 */
      if (cs_is_true(w)) {
        CS_PUSH(n.z, w);
        CS_PUSH(n.y, w);
      } else if (cs_is_false(w)) {
        if (cs_is_true(S.val[n.y])) CS_PUSH(n.z, w);
        if (cs_is_true(S.val[n.z])) CS_PUSH(n.y, w);
      }
      break;
    }
    case CS_OP_OR: { /* propagate.c:361-376 */
      if (cs_is_false(w)) {
        CS_PUSH(n.z, w);
        CS_PUSH(n.y, w);
      } else if (cs_is_true(w)) {
        if (cs_is_false(S.val[n.y])) CS_PUSH(n.z, w);
        if (cs_is_false(S.val[n.z])) CS_PUSH(n.y, w);
      }
      break;
    }
    case CS_OP_WAND: /* propagate.c:379-392 */
      if (cs_is_true(w))
        for (int i = 0; i < n.z; i++) CS_PUSH(T.tkid[n.y + i], w);
      break;
    case CS_OP_CONFL: /* propagate.c:395-471 */
      if (cs_is_true(w)) {
        /* propagate_confl_find: the one element that is not a value while every other has its conflict value */
        int p = -1, stop = 0;
        for (int i = 0; i < n.z && !stop; i++) {
          const cs_val c = S.val[T.tkid[n.y + 2 * i]];
          if (cs_is_value(c)) stop = c.lo != T.tkid[n.y + 2 * i + 1];
          else if (p < 0) p = i;
          else stop = 1;
        }
        if (!stop && p >= 0) { /* propagate_confl_infer: shave the conflict value off the bound it sits on */
          const int term = T.tkid[n.y + 2 * p], cv = T.tkid[n.y + 2 * p + 1];
          const cs_val c = S.val[term];
          if (c.lo == cv && !cs_is_sentinel(c.lo)) CS_PUSH(term, cs_interval(c.lo + 1, CS_DOM_MAX));
          else if (c.hi == cv && !cs_is_sentinel(c.hi)) CS_PUSH(term, cs_interval(CS_DOM_MIN, c.hi - 1));
        }
      }
      break;
    default:
      break;
    }
  }
#undef CS_PUSH
}

/* ---- the hot kernel: event-driven fixpoint, one wave per node ------------------- */

/* TAB_LDS: adj_off, adj and lit are copied into LDS by every workgroup (small models: the inner loop then
 * never waits for L2); the per-wave slices follow the tables */
template <bool HAS_TREE, bool TAB_LDS, bool TRACE = false>
__global__ __launch_bounds__(CS_BLOCK) void cs_propagate_events(cs_tables T, const cs_val *__restrict__ states_in,
                                                                const cs_node_in *__restrict__ nodes,
                                                                cs_val *__restrict__ states_out,
                                                                cs_node_out *__restrict__ results, long long batch,
                                                                const unsigned long long *__restrict__ batch_dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int wave_in_block = threadIdx.x >> 6;
  const int n = T.n_vars, nw = T.n_words;
  /* the search engine launches for an upper bound and leaves the real count on the device */
  if (batch_dev != nullptr && (long long)*batch_dev < batch) batch = (long long)*batch_dev;
  /* a workgroup without a node (launch sized for the upper bound) must not pay for the table copy below */
  if ((long long)blockIdx.x * CS_WAVES_PER_BLOCK >= batch) return;
  /* tables (TAB_LDS): adj_off[n+1] | adj[n_adj] | lit[n_lits], each padded to 16 bytes */
  const int n_adj = TAB_LDS ? T.adj_off[n] : 0;
  const size_t off_bytes = TAB_LDS ? ((((size_t)n + 1) * sizeof(int) + 15) & ~(size_t)15) : 0;
  const size_t adj_bytes = TAB_LDS ? (((size_t)n_adj * sizeof(int2) + 15) & ~(size_t)15) : 0;
  const size_t lit_bytes = TAB_LDS ? (((size_t)T.n_lits * sizeof(int4) + 15) & ~(size_t)15) : 0;
  int *s_adj_off = (int *)cs_lds;
  int2 *s_adj = (int2 *)(cs_lds + off_bytes);
  int4 *s_lit = (int4 *)(cs_lds + off_bytes + adj_bytes);
  if (TAB_LDS) {
    for (int i = threadIdx.x; i <= n; i += blockDim.x) s_adj_off[i] = T.adj_off[i];
    for (int i = threadIdx.x; i < n_adj; i += blockDim.x) s_adj[i] = T.adj[i];
    for (int i = threadIdx.x; i < T.n_lits; i += blockDim.x) s_lit[i] = T.lit[i];
    __syncthreads();
  }
  /* per-wave LDS slice: domains, then the two changed masks */
  const size_t slice = (size_t)n * sizeof(cs_val) + 2 * (size_t)nw * sizeof(unsigned);
  const size_t slice_al = (slice + 15) & ~(size_t)15;
  cs_val *dom = (cs_val *)(cs_lds + off_bytes + adj_bytes + lit_bytes + wave_in_block * slice_al);
  unsigned *mask_a = (unsigned *)(dom + n);
  unsigned *mask_b = mask_a + nw;

  const long long waves_total = (long long)gridDim.x * CS_WAVES_PER_BLOCK;
  for (long long node = (long long)blockIdx.x * CS_WAVES_PER_BLOCK + wave_in_block; node < batch; node += waves_total) {
    const cs_node_in nin = nodes[node];
    const cs_val *src = states_in + (size_t)nin.parent * n;
    /* larger states: eight strides of loads in flight at a time (a plain copy loop waits for every stride) */
    if (n <= 2 * CS_WAVE) {
      for (int v = lane; v < n; v += CS_WAVE) dom[v] = src[v];
    } else for (int v0 = 0; v0 < n; v0 += 8 * CS_WAVE) {
      cs_val t[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int v = v0 + q * CS_WAVE + lane;
        t[q] = v < n ? src[v] : cs_value(0);
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int v = v0 + q * CS_WAVE + lane;
        if (v < n) dom[v] = t[q];
      }
    }
    for (int w = lane; w < nw; w += CS_WAVE) { mask_a[w] = 0u; mask_b[w] = 0u; }
    cs_wave_sync();

    cs_ctx cx;
    cx.log = nullptr;
    cx.cur_clause = -1;
    if (TRACE) {
      cx.log = T.trace_log;
      cx.log_n = T.trace_n;
      cx.log_cap = T.trace_cap;
    }
    cx.dom = dom;
    cx.mark_is_flag = 0;
    cx.fail = 0;
    cx.fail_var = -1;
    cx.props = 0;
    cx.revisions = 0;
    if (nin.var >= 0) {
      /* step_enter: bind(var, VALUE(v)) -- csolve.c:294-304; not counted in PROPS */
      if (lane == 0) {
        dom[nin.var] = cs_interval(nin.lo, nin.hi);
        mask_a[nin.var >> 5] = 1u << (nin.var & 31);
      }
    } else {
      for (int w = lane; w < nw; w += CS_WAVE) {
        int rem = n - w * 32;
        mask_a[w] = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
      }
    }
    cs_wave_sync();
    if (T.obj_var >= 0 && lane == 0) {
      /* untrailed tightening of "<obj>" by the incumbent; the variable counts as changed only
       * if the bound actually moved (csolve.c:251-252 then re-propagates its clauses) */
      cs_val d = dom[T.obj_var];
      int obj_lo = T.obj_lo, obj_hi = T.obj_hi;
      if (T.obj_best_dev != nullptr) {
        const cs_val lim = cs_objective_bound(T.obj_sense, cs_interval(obj_lo, obj_hi), *T.obj_best_dev);
        obj_lo = lim.lo;
        obj_hi = lim.hi;
      }
      const int nl = cs_max(d.lo, obj_lo), nh = cs_min(d.hi, obj_hi);
      if (nl != d.lo || nh != d.hi) {
        dom[T.obj_var] = cs_interval(nl, nh);
        atomicOr(&mask_a[T.obj_var >> 5], 1u << (T.obj_var & 31));
      }
    }
    cs_wave_sync();

    unsigned *cur = mask_a, *nxt = mask_b;
    int rounds = 0, failed = 0;
    for (;;) {
      cx.mark = nxt;
      int any = 0;
      /* the round's changed mask: lane w holds word w (one LDS read for up to 2048 variables), the non-empty words are
       * walked through a ballot; one failure exit per round, not one per variable */
      unsigned my_word = 0u;
      unsigned long long nonempty = 0ull;
      if (nw <= CS_WAVE) {
        my_word = lane < nw ? cur[lane] : 0u;
        nonempty = __ballot(my_word != 0u);
      }
      for (int w = 0; w < nw; w++) {
        unsigned bits;
        if (nw <= CS_WAVE) {
          if (nonempty == 0ull) break;
          w = __builtin_ctzll(nonempty);
          nonempty &= nonempty - 1ull;
          bits = __builtin_amdgcn_readlane(my_word, w);
        } else {
          bits = __builtin_amdgcn_readfirstlane(cur[w]);
        }
        any |= bits != 0u;
        while (bits != 0u) {
          const int u = w * 32 + __builtin_ctz(bits);
          bits &= bits - 1u;
          /* a variable whose bounds crossed through racing lo/hi updates */
          const cs_val du = dom[u];
          if (du.lo > du.hi) {
            if (TRACE && lane != 0) { cx.fail = 1; cx.fail_var = u; } /* one record for the wave */
            else cx.failed_at(u);
          }
          const int beg = TAB_LDS ? s_adj_off[u] : T.adj_off[u], end = TAB_LDS ? s_adj_off[u + 1] : T.adj_off[u + 1];
          for (int i = beg + lane; i < end && !cx.fail; i += CS_WAVE) {
            const int2 e = TAB_LDS ? s_adj[i] : T.adj[i];
            if (TRACE) cx.cur_clause = T.adj_clause[i];
            if (e.x >= 0) {
              cs_lin_revise(cx, u, e.x & CS_ADJ_VAR_MASK, e.y, e.x >> 28);
            } else if (e.y != 0) {
              if (TAB_LDS) cs_or2_revise(cx, s_lit + ~e.x);
              else cs_or2_revise(cx, T.lit + ~e.x);
            } else if (HAS_TREE) {
              cs_tree_scratch S;
              cs_tree_revise(T, ~e.x, cx, S);
            }
          }
        }
      }
      if (__any(cx.fail)) failed = 1;
      if (failed || !any) break;
      rounds++;
      cs_wave_sync();
      for (int w = lane; w < nw; w += CS_WAVE) cur[w] = 0u;
      cs_wave_sync();
      unsigned *t = cur; cur = nxt; nxt = t;
    }
    cs_wave_sync();

    /* wave totals */
    int props = cx.props, revs = cx.revisions;
    for (int off = 32; off > 0; off >>= 1) {
      props += __shfl_xor(props, off);
      revs += __shfl_xor(revs, off);
    }
    int open_vars = 0;
    if (!failed) {
      cs_val *dst = states_out + (size_t)node * n;
      for (int v = lane; v < n; v += CS_WAVE) {
        const cs_val d = dom[v];
        dst[v] = d;
        open_vars += __popcll(__ballot(d.lo != d.hi));
      }
    } else {
      rounds = cs_wave_fail_var(cx, dom, n, lane); /* an inconsistent node reports the failing variable here */
    }
    if (lane == 0) {
      cs_node_out r;
      r.status = failed ? -1 : open_vars;
      r.props = props;
      r.revisions = revs;
      r.rounds = rounds;
      results[node] = r;
    }
    cs_wave_sync();
  }
}

/* ---- kernel 6: small models of binary / two-literal clauses, the clauses resident in registers ------
 *
 * The event-driven kernel walks the changed variables of a round one after the other (a stride of the wave per
 * variable, each a chain of dependent LDS accesses): on a model of a few dozen clauses that walk IS the latency
 * of a node -- 22 us per launch on schedule-10 however small the batch, which is what bounds an iteration of a
 * MIN / MAX search.  Here every lane keeps up to CPL clause records (and the literals of its disjunctions) in
 * registers for the whole kernel and a round revises ALL clauses at once; a round is one chain of LDS
 * accesses.  Expression-tree clauses (HAS_TREE) take one lane each: once per round instead of once per changed
 * variable of the tree.  Same fixpoints and verdicts (the revisions are monotone narrowings; the parent is a fixpoint, so
 * revising a clause nothing touched is a no-op), PROPS counted the same way, order of narrowings different. */
template <int CPL, bool HAS_TREE>
__global__ __launch_bounds__(CS_BLOCK) void cs_propagate_clause_rounds(cs_tables T, const cs_val *__restrict__ states_in,
                                                                       const cs_node_in *__restrict__ nodes,
                                                                       cs_val *__restrict__ states_out,
                                                                       cs_node_out *__restrict__ results, long long batch,
                                                                       const unsigned long long *__restrict__ batch_dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int wave_in_block = threadIdx.x >> 6;
  const int n = T.n_vars;
  if (batch_dev != nullptr && (long long)*batch_dev < batch) batch = (long long)*batch_dev;
  if ((long long)blockIdx.x * CS_WAVES_PER_BLOCK >= batch) return;
  const size_t slice_al = ((size_t)n * sizeof(cs_val) + 16 + 15) & ~(size_t)15;
  cs_val *dom = (cs_val *)(cs_lds + wave_in_block * slice_al);
  unsigned *flag = (unsigned *)(dom + n); /* [0]: something changed this round */

  int4 rec[CPL], lit0[CPL], lit1[CPL];
#pragma unroll
  for (int q = 0; q < CPL; q++) {
    const int c = lane + q * CS_WAVE;
    rec[q] = c < T.n_clauses ? T.clause_by_kind[c] : make_int4(CS_CL_SKIP, 0, 0, 0);
    lit0[q] = rec[q].x == CS_CL_OR2 ? T.lit[rec[q].y] : make_int4(0, 0, 0, 0);
    lit1[q] = rec[q].x == CS_CL_OR2 ? T.lit[rec[q].y + 1] : make_int4(0, 0, 0, 0);
  }

  const long long waves_total = (long long)gridDim.x * CS_WAVES_PER_BLOCK;
  for (long long node = (long long)blockIdx.x * CS_WAVES_PER_BLOCK + wave_in_block; node < batch; node += waves_total) {
    const cs_node_in nin = nodes[node];
    const cs_val *src = states_in + (size_t)nin.parent * n;
    for (int v = lane; v < n; v += CS_WAVE) dom[v] = src[v];
    cs_wave_sync();
    if (lane == 0) {
      /* step_enter: bind(var, VALUE(v)) -- csolve.c:294-304; not counted in PROPS */
      if (nin.var >= 0) dom[nin.var] = cs_interval(nin.lo, nin.hi);
      if (T.obj_var >= 0) { /* untrailed tightening of "<obj>" by the incumbent (objective.c:101-126) */
        int obj_lo = T.obj_lo, obj_hi = T.obj_hi;
        if (T.obj_best_dev != nullptr) {
          const cs_val lim = cs_objective_bound(T.obj_sense, cs_interval(obj_lo, obj_hi), *T.obj_best_dev);
          obj_lo = lim.lo;
          obj_hi = lim.hi;
        }
        const cs_val d = dom[T.obj_var];
        dom[T.obj_var] = cs_interval(cs_max(d.lo, obj_lo), cs_min(d.hi, obj_hi));
      }
    }
    cs_wave_sync();

    cs_ctx cx;
    cx.log = nullptr;
    cx.cur_clause = -1;
    cx.dom = dom;
    cx.mark = flag;
    cx.mark_is_flag = 1;
    cx.fail = 0;
    cx.fail_var = -1;
    cx.props = 0;
    cx.revisions = 0;
    int rounds = 0, failed = 0;
    for (;;) {
      if (lane == 0) flag[0] = 0u;
      /* an interval emptied by the assignment, the incumbent or racing updates of lo and hi */
      for (int v = lane; v < n; v += CS_WAVE)
        if (dom[v].lo > dom[v].hi) cx.failed_at(v);
      cs_wave_sync();
#pragma unroll
      for (int q = 0; q < CPL; q++) {
        /* a node that has failed in the slots so far is done (uniform); the relations sit in the first slots, `=` before
         * `<` before the disjunctions in the order the diverged lanes run, so the bounds a round's disjunctions see are
         * those the relations have just moved (schedule-12 MIN 1.49 -> 1.43 s; an exit between the relations and the
         * disjunctions of ONE slot costs more than it saves: 1.49 s) */
        if (q > 0 && __any(cx.fail)) break;
        if (cx.fail) break;
        const int4 r = rec[q];
        if (r.x == CS_CL_NE) {
          cs_ne_revise(cx, r.y, r.z, r.w);
        } else if (r.x == CS_CL_EQ || r.x == CS_CL_LT) {
          cs_lin_revise(cx, r.y, r.z, r.w, r.x == CS_CL_EQ ? CS_REL_EQ : CS_REL_LT);
        } else if (r.x == CS_CL_OR2) {
          const int4 lits[2] = { lit0[q], lit1[q] };
          cs_or2_revise(cx, lits);
        } else if (HAS_TREE && r.x == CS_CL_TREE) {
          cs_tree_scratch S; /* one lane interprets one expression tree: all trees of the model side by side */
          cs_tree_revise(T, r.y, cx, S);
        }
      }
      cs_wave_sync();
      if (__any(cx.fail)) { failed = 1; break; }
      if (flag[0] == 0u) break;
      rounds++;
    }
    cs_wave_sync();

    int props = cx.props, revs = cx.revisions;
    for (int off = 32; off > 0; off >>= 1) {
      props += __shfl_xor(props, off);
      revs += __shfl_xor(revs, off);
    }
    int open_vars = 0;
    if (!failed) {
      cs_val *dst = states_out + (size_t)node * n;
      for (int v = lane; v < n; v += CS_WAVE) {
        const cs_val d = dom[v];
        dst[v] = d;
        open_vars += __popcll(__ballot(d.lo != d.hi));
      }
    } else {
      rounds = cs_wave_fail_var(cx, dom, n, lane); /* an inconsistent node reports the failing variable here */
    }
    if (lane == 0) {
      cs_node_out r;
      r.status = failed ? -1 : open_vars;
      r.props = props;
      r.revisions = revs;
      r.rounds = rounds;
      results[node] = r;
    }
    cs_wave_sync();
  }
}

/* ---- the hot kernel, LDS-resident variant for pure binary-NE networks ---------------
 *
 * Same fixpoint as cs_propagate_events<false>, restructured around what the PMC profile of
 * that kernel showed (latency-bound on adjacency loads from L2, 1.5x more scalar than vector
 * instructions):
 *   - the whole adjacency (packed to 2 or 4 bytes per entry) and adj_off[] are copied into LDS
 *     once per workgroup; workgroups are persistent (grid = resident workgroups) and up to 16
 *     waves share one copy, so a clause revision touches LDS only;
 *   - the changed variable's own interval is read once per list scan into scalars; the scan is
 *     specialised on it: a VALUE pushes its forbidden value into every neighbour bound
 *     (propagate_eq_false_lr towards the other side, propagate.c:106-120), an OPEN interval only
 *     looks for valued neighbours sitting on one of its two bounds.  Each revision therefore
 *     does one direction, the only one that can fire;
 *   - revisions are counted per scan (scalar), not per lane.
 */
#define CS_CHUNK 16 /* nodes a wave takes at a time: their records sit in lanes 0..15 */
__device__ __forceinline__ int cs_wave_sum(int x); /* below */
__device__ __forceinline__ unsigned cs_wave_or(unsigned x); /* below */

/* ADJ_LDS = false: the adjacency stays in device memory (it is a few tens of KB that every workgroup reads: L2-resident)
 * and LDS holds the per-wave slices only -- for models whose lists would leave room for a handful of waves (a 25x25
 * sudoku: 90 KB of lists, 5 KB per node), where latency, not LDS bandwidth, is what needs hiding.
 * csz: nodes a wave takes at a time (1..16; small for small batches, so that every wave gets a share). */
/* R strides of 64 variables travel through registers; the host picks R with (R - 1) * 64 <= n, so that only the LAST of
 * them asks "v < n" */
template <typename E, int R, int U, bool ADJ_LDS = true>
__global__ __launch_bounds__(1024, (R <= 2 ? 8 : 4)) void cs_propagate_ne_lds(cs_tables T, const E *__restrict__ adj_packed, int n_adj,
                                                            int obits, int dmin,
                                                            const cs_val *__restrict__ states_in,
                                                            const cs_node_in *__restrict__ nodes,
                                                            cs_val *__restrict__ states_out,
                                                            cs_node_out *__restrict__ results, long long batch, int csz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const int n = T.n_vars, nw = T.n_words;
  /* LDS: {begin,end} of every list [n] | packed adjacency | one slice per wave
   * (domains, two masks, the propagation counter) */
  int2 *s_off2 = (int2 *)cs_lds;
  const size_t off_bytes = ADJ_LDS ? (((size_t)n * sizeof(int2)) + 15) & ~(size_t)15 : 0;
  E *s_adj = (E *)(cs_lds + off_bytes);
  const size_t adj_bytes = ADJ_LDS ? (((size_t)n_adj * sizeof(E)) + 15) & ~(size_t)15 : 0;
  const size_t slice = (size_t)n * sizeof(cs_val) + (2 * (size_t)nw + 1) * sizeof(unsigned);
  const size_t slice_al = (slice + 15) & ~(size_t)15;
  cs_val *dom = (cs_val *)(cs_lds + off_bytes + adj_bytes + wave_in_block * slice_al);
  unsigned *mask_a = (unsigned *)(dom + n);
  unsigned *mask_b = mask_a + nw;
  unsigned *pcount = mask_b + nw;

  if (ADJ_LDS) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_off2[i] = make_int2(T.adj_off[i], T.adj_off[i + 1]);
    for (int i = threadIdx.x; i < n_adj; i += blockDim.x) s_adj[i] = adj_packed[i];
    __syncthreads();
  }

  const unsigned omask = (1u << obits) - 1u;
  const long long chunks = (batch + csz - 1) / csz;
  const long long waves_total = (long long)gridDim.x * waves_per_block;
  for (long long chunk = (long long)blockIdx.x * waves_per_block + wave_in_block; chunk < chunks; chunk += waves_total) {
    const long long base = chunk * csz;
    const int cnt = (int)(batch - base < csz ? batch - base : csz);
    /* the chunk's node records: one coalesced 16-byte load per lane, then register reads only */
    cs_node_in rec;
    rec.var = -1; rec.lo = 0; rec.hi = 0; rec.parent = 0;
    if (lane < cnt) rec = nodes[base + lane];
    cs_node_out my_result;
    my_result.status = 0; my_result.props = 0; my_result.revisions = 0; my_result.rounds = 0;

    /* software pipeline: the state of node j+1 is in flight while node j is propagated */
    cs_val pre[R];
    {
      const cs_val *src = states_in + (size_t)__builtin_amdgcn_readlane(rec.parent, 0) * n;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int v = lane + r * CS_WAVE;
        pre[r] = r < R - 1 || v < n ? src[v] : cs_value(0);
      }
    }
    for (int j = 0; j < cnt; j++) {
      const int nvar = __builtin_amdgcn_readlane(rec.var, j);
      const int nlo = __builtin_amdgcn_readlane(rec.lo, j), nhi = __builtin_amdgcn_readlane(rec.hi, j);
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int v = lane + r * CS_WAVE;
        if (r < R - 1 || v < n) dom[v] = pre[r];
      }
      for (int v = lane + R * CS_WAVE; v < n; v += CS_WAVE) /* n > 64*R: the tail is loaded in place */
        dom[v] = states_in[(size_t)__builtin_amdgcn_readlane(rec.parent, j) * n + v];
      if (j + 1 < cnt) {
        const cs_val *src = states_in + (size_t)__builtin_amdgcn_readlane(rec.parent, j + 1) * n;
#pragma unroll
        for (int r = 0; r < R; r++) {
          const int v = lane + r * CS_WAVE;
          pre[r] = r < R - 1 || v < n ? src[v] : cs_value(0);
        }
      }
      for (int w = lane; w < nw; w += CS_WAVE) { mask_a[w] = 0u; mask_b[w] = 0u; }
      if (lane == 0) *pcount = 0u;
      cs_wave_sync();
      if (nvar >= 0) {
        if (lane == 0) {
          dom[nvar] = cs_interval(nlo, nhi);
          mask_a[nvar >> 5] = 1u << (nvar & 31);
        }
      } else {
        for (int w = lane; w < nw; w += CS_WAVE) {
          int rem = n - w * 32;
          mask_a[w] = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
        }
      }
      cs_wave_sync();

      unsigned *cur = mask_a, *nxt = mask_b;
      int rounds = 0, failed = 0, revisions = 0, fail = 0;
      /* narrowing: LDS atomic on the bound; the lane whose atomic moved it books the propagation
       * in the node's LDS counter and marks the variable for the next round */
#define CS_RAISE(V, LO)                                                        \
  do {                                                                         \
    const int v_ = (V), lo_ = (LO);                                            \
    if (atomicMax(&dom[v_].lo, lo_) < lo_) {                                   \
      atomicAdd(pcount, 1u);                                                   \
      atomicOr(&nxt[v_ >> 5], 1u << (v_ & 31));                                \
      if (lo_ > dom[v_].hi) fail = 1;                                          \
    }                                                                          \
  } while (0)
#define CS_LOWER(V, HI)                                                        \
  do {                                                                         \
    const int v_ = (V), hi_ = (HI);                                            \
    if (atomicMin(&dom[v_].hi, hi_) > hi_) {                                   \
      atomicAdd(pcount, 1u);                                                   \
      atomicOr(&nxt[v_ >> 5], 1u << (v_ & 31));                                \
      if (hi_ < dom[v_].lo) fail = 1;                                          \
    }                                                                          \
  } while (0)
      for (;;) {
        int any = 0;
        /* the round's changed mask: lane w holds word w (one LDS read for up to 2048 variables), the words that are
         * not empty are walked through a ballot */
        unsigned my_word = 0u;
        unsigned long long nonempty = 0ull;
        if (nw <= CS_WAVE) {
          my_word = lane < nw ? cur[lane] : 0u;
          nonempty = __ballot(my_word != 0u);
        }
        for (int w = 0; w < nw; w++) {
          unsigned bits;
          if (nw <= CS_WAVE) {
            if (nonempty == 0ull) break;
            w = __builtin_ctzll(nonempty);
            nonempty &= nonempty - 1ull;
            bits = __builtin_amdgcn_readlane(my_word, w);
          } else {
            bits = __builtin_amdgcn_readfirstlane(cur[w]);
          }
          any |= bits != 0u;
          while (bits != 0u) {
            const int u = w * 32 + __builtin_ctz(bits);
            bits &= bits - 1u;
            const cs_val du = dom[u];
            const int2 range = ADJ_LDS ? s_off2[u] : make_int2(T.adj_off[u], T.adj_off[u + 1]);
            const int ulo = __builtin_amdgcn_readfirstlane(du.lo), uhi = __builtin_amdgcn_readfirstlane(du.hi);
            if (ulo > uhi) { fail = 1; continue; } /* bounds crossed by racing updates (the round is finished all the same:
                                                    * one exit from the loops, not one per variable, keeps the scalar
                                                    * bookkeeping of the control flow small) */
            const int beg = __builtin_amdgcn_readfirstlane(range.x), end = __builtin_amdgcn_readfirstlane(range.y);
            revisions += end - beg;
            const int is_value = ulo == uhi;
            /* u open: which of the 32 values from each bound inwards does a valued neighbour forbid?  (bit p of flo:
             * ulo + p, of fhi: uhi - p) */
            unsigned flo = 0u, fhi = 0u;
            /* U strides at a time: all entry reads, then all domain reads, then the compares */
            for (int i0 = beg + lane; i0 < end; i0 += U * CS_WAVE) {
              unsigned e[U];
              cs_val dw[U];
#pragma unroll
              for (int k = 0; k < U; k++) {
                const int i = i0 + k * CS_WAVE;
                e[k] = i < end ? (unsigned)(ADJ_LDS ? s_adj[i] : adj_packed[i]) : 0xffffffffu;
              }
#pragma unroll
              for (int k = 0; k < U; k++) dw[k] = e[k] != 0xffffffffu ? dom[e[k] & omask] : cs_interval(1, 0);
#pragma unroll
              for (int k = 0; k < U; k++) {
                if (e[k] == 0xffffffffu) continue;
                const int wv = (int)(e[k] & omask);
                const int d = (int)(e[k] >> obits) + dmin;
                if (is_value) { /* u is a value: the neighbour must avoid ulo - d */
                  const int f = ulo - d;
                  if (dw[k].lo == f) CS_RAISE(wv, f + 1);
                  else if (dw[k].hi == f) CS_LOWER(wv, f - 1);
                } else if (dw[k].lo == dw[k].hi) { /* u is open: a valued neighbour near one of its bounds */
                  const int f = dw[k].lo + d;
                  const unsigned pl = (unsigned)(f - ulo), ph = (unsigned)(uhi - f);
                  if (pl < 32u) flo |= 1u << pl;
                  if (ph < 32u) fhi |= 1u << ph;
                }
              }
            }
            if (!is_value) {
              /* The whole list is seen: the bound moves past EVERY forbidden value in one step -- as many reference
               * narrowings as values passed (each is one unit shave of propagate_eq_false_lr, propagate.c:106-120) --
               * instead of one value per round with a scan of this list each.  The new bound is free of valued
               * neighbours, so u comes back only when it has become a value (it then pushes) or 32 values were not enough. */
              flo = cs_wave_or(flo);
              fhi = cs_wave_or(fhi);
              const int up = flo == 0xffffffffu ? 32 : __builtin_ctz(~flo), down = fhi == 0xffffffffu ? 32 : __builtin_ctz(~fhi);
              if (lane == 0) {
                if (up > 0) {
                  const int lo_ = ulo + up, old = atomicMax(&dom[u].lo, lo_);
                  if (old < lo_) {
                    atomicAdd(pcount, (unsigned)(lo_ - old));
                    const int hi_now = dom[u].hi;
                    if (lo_ > hi_now) fail = 1;
                    else if (lo_ == hi_now || up == 32) atomicOr(&nxt[u >> 5], 1u << (u & 31));
                  }
                }
                if (down > 0 && !fail) {
                  const int hi_ = uhi - down, old = atomicMin(&dom[u].hi, hi_);
                  if (old > hi_) {
                    atomicAdd(pcount, (unsigned)(old - hi_));
                    const int lo_now = dom[u].lo;
                    if (hi_ < lo_now) fail = 1;
                    else if (hi_ == lo_now || down == 32) atomicOr(&nxt[u >> 5], 1u << (u & 31));
                  }
                }
              }
            }
          }
        }
        if (__any(fail)) failed = 1;
        if (failed || !any) break;
        rounds++;
        cs_wave_sync();
        for (int w = lane; w < nw; w += CS_WAVE) cur[w] = 0u;
        cs_wave_sync();
        unsigned *t = cur; cur = nxt; nxt = t;
      }
#undef CS_RAISE
#undef CS_LOWER
      cs_wave_sync();

      const int props = (int)*pcount;
      int open_vars = 0;
      if (!failed) {
        cs_val *dst = states_out + (size_t)(base + j) * n;
        /* R strides straight through (all LDS reads in flight, then the stores), the tail of a larger state after them;
         * open variables are counted per lane and summed once */
        int open_l = 0;
        cs_val dd[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
          const int v = lane + r * CS_WAVE;
          dd[r] = r < R - 1 || v < n ? dom[v] : cs_value(0);
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
          const int v = lane + r * CS_WAVE;
          if (r < R - 1 || v < n) dst[v] = dd[r];
          open_l += dd[r].lo != dd[r].hi ? 1 : 0;
        }
        for (int v = lane + R * CS_WAVE; v < n; v += CS_WAVE) {
          const cs_val d = dom[v];
          dst[v] = d;
          open_l += d.lo != d.hi ? 1 : 0;
        }
        open_vars = cs_wave_sum(open_l);
      }
      if (lane == j) { /* lane j keeps node j's result; one coalesced store per chunk */
        my_result.status = failed ? -1 : open_vars;
        my_result.props = props;
        my_result.revisions = revisions;
        my_result.rounds = rounds;
      }
      cs_wave_sync();
    }
    if (lane < cnt) results[base + lane] = my_result;
  }
}

/* ---- the hot kernel, forbidden-set variant for pure binary-NE networks --------------
 *
 * What the reference's propagators compute on a != network is, for every open variable, the
 * interval between the lowest and the highest value that no VALUED neighbour forbids
 * (propagate_eq_false_lr only fires when the other side is a single value sitting on a bound,
 * propagate.c:106-120, and keeps firing until the bound is free).  This kernel keeps that
 * "forbidden by a valued neighbour" set explicitly: FW 64-bit words per variable, bit k <=>
 * value root_lo+k.  A round is
 *   (1) every variable that has just become a value pushes its forbidden value into the sets of
 *       all its neighbours (one LDS atomic OR per adjacency entry, no domain reads), then
 *   (2) every variable (one lane each) re-derives its bounds from its set with two bit scans;
 *       an empty set of allowed values is the inconsistency, a variable left with one value
 *       is pushed in the next round.
 * The fixpoint, the verdict and -- because every reference propagation on such a network removes
 * exactly one value from one bound -- the PROPS count of a consistent node are those of the
 * unit-shaving kernels above; the work per node drops from one list scan per narrowing to one
 * list scan per NEWLY VALUED variable.
 * The sets travel with the state (forb_in / forb_out, [rows][n][FW] u64): a child inherits its
 * parent's sets, so nothing is rebuilt down a search path.  With forb_in == NULL the sets are
 * rebuilt from the valued variables of the incoming state (one list scan per valued variable).
 */
template <typename E, int FW, int R> /* R = ceil(n_vars/64) for the prefetch pipeline, 0 = none */
__global__ __launch_bounds__(1024, (FW <= 1 && R <= 1 ? 8 : 4)) void cs_propagate_ne_bitset(
    cs_tables T, const E *__restrict__ adj_packed, int n_adj, int obits, int dmin, const int *__restrict__ root_lo,
    const cs_val *__restrict__ states_in, const unsigned long long *__restrict__ forb_in,
    const cs_node_in *__restrict__ nodes, cs_val *__restrict__ states_out, unsigned long long *__restrict__ forb_out,
    cs_node_out *__restrict__ results, long long batch, const unsigned long long *__restrict__ batch_dev,
    int csz /* nodes a wave takes at a time, 1..16: small for small batches, to spread them over the machine */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  typedef unsigned long long u64;
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const int n = T.n_vars, nw = T.n_words;
  if (batch_dev != nullptr && (long long)*batch_dev < batch) batch = (long long)*batch_dev;
  if ((long long)blockIdx.x * waves_per_block * csz >= batch) return; /* no node for this workgroup: skip the table copy */
  /* LDS: {begin,end}[n] | root_lo[n] | packed adjacency | per wave: domains, sets, two masks, counter */
  int2 *s_off2 = (int2 *)cs_lds;
  const size_t off_bytes = (((size_t)n * sizeof(int2)) + 15) & ~(size_t)15;
  int *s_base = (int *)(cs_lds + off_bytes);
  const size_t base_bytes = (((size_t)n * sizeof(int)) + 15) & ~(size_t)15;
  E *s_adj = (E *)(cs_lds + off_bytes + base_bytes);
  const size_t adj_bytes = (((size_t)n_adj * sizeof(E)) + 15) & ~(size_t)15;
  const size_t slice = (size_t)n * sizeof(cs_val) + (size_t)n * FW * sizeof(u64) + (2 * (size_t)nw + 2) * sizeof(unsigned);
  const size_t slice_al = (slice + 15) & ~(size_t)15;
  unsigned char *mine = cs_lds + off_bytes + base_bytes + adj_bytes + wave_in_block * slice_al;
  cs_val *dom = (cs_val *)mine;
  u64 *forb = (u64 *)(dom + n);
  unsigned *forb32 = (unsigned *)forb; /* the same sets as 32-bit words */
  unsigned *mask_a = (unsigned *)(forb + (size_t)n * FW);
  unsigned *mask_b = mask_a + nw;
  unsigned *pcount = mask_b + nw; /* propagations of the current node */

  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    s_off2[i] = make_int2(T.adj_off[i], T.adj_off[i + 1]);
    s_base[i] = root_lo[i];
  }
  for (int i = threadIdx.x; i < n_adj; i += blockDim.x) s_adj[i] = adj_packed[i];
  __syncthreads();

  const unsigned omask = (1u << obits) - 1u;
  const long long chunks = (batch + csz - 1) / csz;
  const long long waves_total = (long long)gridDim.x * waves_per_block;
  for (long long chunk = (long long)blockIdx.x * waves_per_block + wave_in_block; chunk < chunks; chunk += waves_total) {
    const long long base = chunk * csz;
    const int cnt = (int)(batch - base < csz ? batch - base : csz);
    cs_node_in rec;
    rec.var = -1; rec.lo = 0; rec.hi = 0; rec.parent = 0;
    if (lane < cnt) rec = nodes[base + lane];
    cs_node_out my_result;
    my_result.status = 0; my_result.props = 0; my_result.revisions = 0; my_result.rounds = 0;

    /* software pipeline (R > 0): the state and the sets of node j+1 are in flight while node j
     * is propagated */
    constexpr int RP = R > 0 ? R : 1;
    cs_val pre_dom[RP];
    u64 pre_forb[RP * FW];
    if (R > 0) {
      const size_t prow0 = (size_t)__builtin_amdgcn_readlane(rec.parent, 0);
#pragma unroll
      for (int r = 0; r < RP; r++) {
        const int v = lane + r * CS_WAVE;
        pre_dom[r] = v < n ? states_in[prow0 * n + v] : cs_value(0);
      }
      if (forb_in != nullptr) {
#pragma unroll
        for (int r = 0; r < RP * FW; r++) {
          const int k = lane + r * CS_WAVE;
          pre_forb[r] = k < n * FW ? forb_in[prow0 * n * FW + k] : 0ull;
        }
      }
    }
    for (int j = 0; j < cnt; j++) {
      const int nvar = __builtin_amdgcn_readlane(rec.var, j);
      const int nlo = __builtin_amdgcn_readlane(rec.lo, j), nhi = __builtin_amdgcn_readlane(rec.hi, j);
      const size_t prow = (size_t)__builtin_amdgcn_readlane(rec.parent, j);
      if (R > 0) {
#pragma unroll
        for (int r = 0; r < RP; r++) {
          const int v = lane + r * CS_WAVE;
          if (v < n) dom[v] = pre_dom[r];
        }
#pragma unroll
        for (int r = 0; r < RP * FW; r++) {
          const int k = lane + r * CS_WAVE;
          if (k < n * FW) forb[k] = forb_in != nullptr ? pre_forb[r] : 0ull;
        }
        if (j + 1 < cnt) {
          const size_t pnext = (size_t)__builtin_amdgcn_readlane(rec.parent, j + 1);
#pragma unroll
          for (int r = 0; r < RP; r++) {
            const int v = lane + r * CS_WAVE;
            pre_dom[r] = v < n ? states_in[pnext * n + v] : cs_value(0);
          }
          if (forb_in != nullptr) {
#pragma unroll
            for (int r = 0; r < RP * FW; r++) {
              const int k = lane + r * CS_WAVE;
              pre_forb[r] = k < n * FW ? forb_in[pnext * n * FW + k] : 0ull;
            }
          }
        }
      } else {
        /* large states: eight strides of loads in flight at a time (a plain copy loop waits for every
         * stride before it issues the next one) */
        const cs_val *src = states_in + prow * n;
        for (int v0 = 0; v0 < n; v0 += 8 * CS_WAVE) {
          cs_val t[8];
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const int v = v0 + q * CS_WAVE + lane;
            t[q] = v < n ? src[v] : cs_value(0);
          }
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const int v = v0 + q * CS_WAVE + lane;
            if (v < n) dom[v] = t[q];
          }
        }
        if (forb_in != nullptr) {
          const u64 *fsrc = forb_in + prow * n * FW;
          for (int k0 = 0; k0 < n * FW; k0 += 8 * CS_WAVE) {
            u64 t[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
              const int k = k0 + q * CS_WAVE + lane;
              t[q] = k < n * FW ? fsrc[k] : 0ull;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
              const int k = k0 + q * CS_WAVE + lane;
              if (k < n * FW) forb[k] = t[q];
            }
          }
        } else {
          for (int k = lane; k < n * FW; k += CS_WAVE) forb[k] = 0ull;
        }
      }
      if (n > CS_WAVE)
        for (int w = lane; w < nw; w += CS_WAVE) { mask_a[w] = 0u; mask_b[w] = 0u; }
      if (lane == 0) *pcount = 0u;
      cs_wave_sync();
      /* the assignment (step_enter, csolve.c:294-304) and the first set of variables to push */
      if (lane == 0 && nvar >= 0) dom[nvar] = cs_interval(nlo, nhi);
      cs_wave_sync();
      const bool small = n <= CS_WAVE; /* one lane per variable: the push set lives in a scalar */
      u64 cur64 = 0ull;
      if (forb_in == nullptr || nvar < 0) {
        /* no inherited sets: every valued variable of the incoming state pushes */
        for (int base_v = 0; base_v < n; base_v += CS_WAVE) {
          const int v = base_v + lane;
          const cs_val d = v < n ? dom[v] : cs_interval(0, 1);
          const u64 b = __ballot(d.lo == d.hi);
          cur64 = b;
          if (!small && lane == 0) {
            mask_a[base_v >> 5] = (unsigned)b;
            if ((base_v >> 5) + 1 < nw) mask_a[(base_v >> 5) + 1] = (unsigned)(b >> 32);
          }
        }
      } else if (nlo == nhi) {
        cur64 = 1ull << (nvar & 63);
        if (!small && lane == 0) mask_a[nvar >> 5] = 1u << (nvar & 31);
      }
      cs_wave_sync();

      unsigned *cur = mask_a, *nxt = mask_b;
      int rounds = 0, failed = 0, revisions = 0;
      for (;;) {
        /* (1) newly valued variables push their forbidden value into the neighbours' sets */
        /* the round's push set: with more than 64 variables lane w holds 32-bit word w of the mask (one LDS read for up
         * to 2048 variables) and the non-empty words are walked through a ballot */
        const bool by_lane = !small && nw <= CS_WAVE;
        unsigned my_word = 0u;
        u64 nonempty = 0ull;
        if (by_lane) {
          my_word = lane < nw ? cur[lane] : 0u;
          nonempty = __ballot(my_word != 0u);
        }
        const int words = small ? 1 : (by_lane ? nw : (nw + 1) / 2);
        for (int w = 0; w < words; w++) {
          u64 bits;
          int ubase;
          if (small) {
            bits = cur64;
            ubase = 0;
          } else if (by_lane) {
            if (nonempty == 0ull) break;
            w = __builtin_ctzll(nonempty);
            nonempty &= nonempty - 1ull;
            bits = (u64)(unsigned)__builtin_amdgcn_readlane(my_word, w);
            ubase = w * 32;
          } else {
            bits = (u64)(unsigned)__builtin_amdgcn_readfirstlane(cur[2 * w]) |
                   ((u64)(2 * w + 1 < nw ? (unsigned)__builtin_amdgcn_readfirstlane(cur[2 * w + 1]) : 0u) << 32);
            ubase = w * 64;
          }
          while (bits != 0ull) {
            const int u = ubase + __builtin_ctzll(bits);
            bits &= bits - 1ull;
            const int c = __builtin_amdgcn_readfirstlane(dom[u].lo);
            const int2 range = s_off2[u];
            const int beg = __builtin_amdgcn_readfirstlane(range.x), end = __builtin_amdgcn_readfirstlane(range.y);
            revisions += end - beg;
            for (int i = beg + lane; i < end; i += CS_WAVE) {
              const unsigned e = s_adj[i];
              const int wv = (int)(e & omask);
              const int bit = c - ((int)(e >> obits) + dmin); /* the offset includes root_lo[wv] */
              /* 32-bit words (no variable 64-bit shifts in this file, see cs_propagate_ne_regs) */
              if ((unsigned)bit < (unsigned)(64 * FW)) atomicOr(&forb32[wv * (2 * FW) + (bit >> 5)], 1u << (bit & 31));
            }
          }
        }
        cs_wave_sync();
        if (!small)
          for (int w = lane; w < nw; w += CS_WAVE) cur[w] = 0u;
        /* (2) every variable re-derives its bounds from its set */
        int fail = 0;
        u64 any_new = 0ull;
        for (int base_v = 0; base_v < n; base_v += CS_WAVE) {
          const int v = base_v + lane;
          int newly = 0;
          if (v < n) {
            const cs_val d = dom[v];
            const int b0 = s_base[v];
            int lo = d.lo, hi = d.hi;
            if (lo > hi) {
              fail = 1;
            } else {
              /* allowed = values of [lo,hi] that are not forbidden; its lowest / highest member,
               * word by word (32 values per word) */
              const int from = lo - b0, to = hi - b0;
              int first = 0x7fffffff, last = -1;
#pragma unroll
              for (int q = 0; q < 2 * FW; q++) {
                const int f = from - 32 * q, t = to - 32 * q;
                const unsigned mlo = f <= 0 ? ~0u : (f > 31 ? 0u : ~0u << (f & 31));
                const unsigned mhi = t >= 31 ? ~0u : (t < 0 ? 0u : ~0u >> ((31 - t) & 31));
                const unsigned a = ~forb32[v * (2 * FW) + q] & mlo & mhi;
                const int lo_q = a != 0u ? 32 * q + __builtin_ctz(a) : 0x7fffffff;
                const int hi_q = a != 0u ? 32 * q + 31 - __builtin_clz(a) : -1;
                first = lo_q < first ? lo_q : first;
                last = hi_q > last ? hi_q : last;
              }
              const int nlo2 = last < 0 ? 0x7fffffff : b0 + first, nhi2 = last < 0 ? (int)0x80000000 : b0 + last;
              if (nlo2 > nhi2) {
                fail = 1;
              } else if (nlo2 != lo || nhi2 != hi) {
                atomicAdd(pcount, (unsigned)((nlo2 - lo) + (hi - nhi2))); /* rare: a few lanes per node */
                dom[v] = cs_interval(nlo2, nhi2);
                newly = nlo2 == nhi2;
              }
            }
          }
          const u64 nb = __ballot(newly);
          any_new |= nb;
          if (!small && lane == 0 && nb != 0ull) {
            nxt[base_v >> 5] |= (unsigned)nb;
            if ((base_v >> 5) + 1 < nw) nxt[(base_v >> 5) + 1] |= (unsigned)(nb >> 32);
          }
        }
        if (__any(fail)) { failed = 1; break; }
        cs_wave_sync();
        if (any_new == 0ull) break; /* ballots are wave-uniform: no LDS read needed */
        rounds++;
        cur64 = any_new;
        unsigned *t = cur; cur = nxt; nxt = t;
      }
      cs_wave_sync();

      const int props = (int)*pcount;
      int open_vars = 0;
      if (!failed) {
        cs_val *dst = states_out + (size_t)(base + j) * n;
        for (int v = lane; v < n; v += CS_WAVE) {
          const cs_val d = dom[v];
          dst[v] = d;
          open_vars += __popcll(__ballot(d.lo != d.hi));
        }
        if (forb_out != nullptr) {
          u64 *fdst = forb_out + (size_t)(base + j) * n * FW;
          for (int k = lane; k < n * FW; k += CS_WAVE) fdst[k] = forb[k];
        }
      }
      open_vars = __builtin_amdgcn_readfirstlane(open_vars); /* lanes >= n_vars never ran the loop */
      if (lane == j) {
        my_result.status = failed ? -1 : open_vars;
        my_result.props = props;
        my_result.revisions = revisions;
        my_result.rounds = rounds;
      }
      cs_wave_sync();
    }
    if (lane < cnt) results[base + lane] = my_result;
  }
}

/* ---- forbidden sets in registers: models of at most 256 variables ---------------------------
 *
 * Same algorithm, results and buffers as cs_propagate_ne_bitset, for models whose symmetric
 * relation fits a dense table tab[u][slot][w] in LDS (cs_device.h).  Lane l owns variables
 * l, l+64, ... (R of them): bounds and forbidden words stay in VGPRs from the global load to the
 * global store.  A push by variable u is a conflict-free LDS read of row (u, slot) -- lane w reads
 * ITS OWN entry, so the sets need no atomics -- and step (2) is pure register arithmetic.  LDS holds
 * only the table, shared by the workgroup; PROPS of a node is the wave sum of the units shaved
 * from every variable.  D nodes are in flight per wave (register prefetch).
 */
__device__ __forceinline__ int cs_wave_sum(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); /* row_shr:1 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); /* row_shr:2 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); /* row_shr:4 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); /* row_shr:8: lane 15 of a row = row sum */
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true); /* row_bcast:15 into rows 1 and 3 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true); /* row_bcast:31 into rows 2 and 3 */
  return __builtin_amdgcn_readlane(x, 63);
}

__device__ __forceinline__ unsigned cs_wave_or(unsigned x) { /* the OR over the 64 lanes, the same in every lane */
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

/* FAST: n_vars == 64 * R and both set buffers are there -- every load and store of the node loop is
 * then unconditional straight-line code (a load or store under a branch makes the compiler's s_waitcnt
 * counts at the join conservative).  FAST stores the (meaningless) row of an inconsistent node too
 * instead of branching around it.
 *
 * No variable 64-bit shifts: the sets are handled as 32-bit words (two per u64).  A first version built
 * the bit with `1ull << bit`; the compiler put the shift amount into the last vector register the kernel
 * owns (`v_lshlrev_b64 v[44:45], v47, 1` with 48 registers), and on gfx950 a 64-bit shift whose amount is
 * in the last register of the wave's allocation intermittently shifts by v0 instead when several waves
 * share a SIMD (tools/k4_fault_repro.md, tools/shift64_last_vgpr.hip).  tools/check_isa_shift64.py
 * (tests/test_isa_lint.py) fails the CPU suite if any kernel of the library ever gets such a shift. */
/* 32-bit words of a forbidden set, word q = values 32 q .. 32 q + 31 relative to the variable's root lower bound:
 * lowest and highest allowed value within [from, to] (first > last: none) */
template <int NW>
__device__ __forceinline__ void cs_set_bounds(const unsigned *w, int from, int to, int *first_out, int *last_out) {
  int first = 0x7fffffff, last = -1;
#pragma unroll
  for (int q = 0; q < NW; q++) {
    const int f = from - 32 * q, t = to - 32 * q; /* the interval in this word's coordinates */
    const unsigned mlo = f <= 0 ? ~0u : (f > 31 ? 0u : ~0u << (f & 31));
    const unsigned mhi = t >= 31 ? ~0u : (t < 0 ? 0u : ~0u >> ((31 - t) & 31));
    const unsigned a = ~w[q] & mlo & mhi;
    const int lo_q = a != 0u ? 32 * q + __builtin_ctz(a) : 0x7fffffff;
    const int hi_q = a != 0u ? 32 * q + 31 - __builtin_clz(a) : -1;
    first = lo_q < first ? lo_q : first;
    last = hi_q > last ? hi_q : last;
  }
  *first_out = first;
  *last_out = last;
}

/* the same over the whole window -- for sets that already mark everything outside the variable's interval
 * (sets-only states), where the range masks would be redundant */
template <int NW>
__device__ __forceinline__ void cs_set_bounds_all(const unsigned *w, int *first_out, int *last_out) {
  int first = 0x7fffffff, last = -1;
#pragma unroll
  for (int q = NW - 1; q >= 0; q--) {
    const unsigned a = ~w[q];
    first = a != 0u ? 32 * q + __builtin_ctz(a) : first; /* lower words last: they win */
  }
#pragma unroll
  for (int q = 0; q < NW; q++) {
    const unsigned a = ~w[q];
    last = a != 0u ? 32 * q + 31 - __builtin_clz(a) : last; /* higher words last: they win */
  }
  *first_out = first;
  *last_out = last;
}

/* mark every value outside [from, to] (relative) as forbidden */
template <int NW>
__device__ __forceinline__ void cs_set_restrict(unsigned *w, int from, int to) {
#pragma unroll
  for (int q = 0; q < NW; q++) {
    const int f = from - 32 * q, t = to - 32 * q;
    const unsigned mlo = f <= 0 ? ~0u : (f > 31 ? 0u : ~0u << (f & 31));
    const unsigned mhi = t >= 31 ? ~0u : (t < 0 ? 0u : ~0u >> ((31 - t) & 31));
    w[q] |= ~(mlo & mhi);
  }
}
#define CS_K4_OUT_RESTRICT 1

template <typename E, int FW, int R, int D, bool FAST, bool SO>
__global__ __launch_bounds__(1024, (FW * R <= 1 ? 8 : 4)) void cs_propagate_ne_regs(
    int n, const E *__restrict__ tab_g, int slots, int dmin, const int *__restrict__ root_lo,
    const int *__restrict__ sym_off, const cs_val *__restrict__ states_in,
    const unsigned long long *__restrict__ forb_in, const cs_node_in *__restrict__ nodes,
    cs_val *__restrict__ states_out, unsigned long long *__restrict__ forb_out, cs_node_out *__restrict__ results,
    long long batch, const unsigned long long *__restrict__ batch_dev,
    int csz /* nodes a wave takes at a time, 1..16: small for small batches, to spread them over the machine */,
    int flags /* CS_K4_OUT_RESTRICT: the stored sets also mark the values outside the stored interval */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  constexpr int W = CS_WAVE * R; /* columns of the table */
  if (batch_dev != nullptr && (long long)*batch_dev < batch) batch = (long long)*batch_dev;
  constexpr int NW = 2 * FW;     /* 32-bit words of forbidden set per variable */
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); /* keeps row addresses scalar */
  const int waves_per_block = blockDim.x >> 6;
  if ((long long)blockIdx.x * waves_per_block * csz >= batch) return; /* no node for this workgroup: skip the table copy */
  E *s_tab = (E *)cs_lds;
  {
    const int vecs = (int)(((size_t)n * slots * W * sizeof(E)) / 16); /* W is a multiple of 64 */
    const uint4 *src = (const uint4 *)tab_g;
    uint4 *dst = (uint4 *)cs_lds;
    for (int i = threadIdx.x; i < vecs; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();

  int b0[R], deg[R], vcl[R];
  bool live[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int v = lane + r * CS_WAVE;
    live[r] = FAST || v < n;
    vcl[r] = live[r] ? v : n - 1; /* lanes past the end re-read the last variable and ignore it */
    b0[r] = live[r] ? root_lo[vcl[r]] : 0;
    deg[r] = live[r] ? sym_off[vcl[r] + 1] - sym_off[vcl[r]] : 0;
  }
  const bool have_in = FAST || SO || forb_in != nullptr;
  const uint2 *forb_in2 = (const uint2 *)forb_in; /* one u64 = {low word, high word} */
  uint2 *forb_out2 = (uint2 *)forb_out;

  const long long chunks = (batch + csz - 1) / csz;
  const long long waves_total = (long long)gridDim.x * waves_per_block;
  for (long long chunk = (long long)blockIdx.x * waves_per_block + wave_in_block; chunk < chunks; chunk += waves_total) {
    const long long base = chunk * csz;
    const int cnt = (int)(batch - base < csz ? batch - base : csz);
    cs_node_in rec;
    rec.var = -1; rec.lo = 0; rec.hi = 0; rec.parent = 0;
    if (lane < cnt) rec = nodes[base + lane];
    cs_node_out my_result;
    my_result.status = 0; my_result.props = 0; my_result.revisions = 0; my_result.rounds = 0;

    /* D nodes in flight; a node index past the end of the chunk re-reads the chunk's last node */
    cs_val pd[D][R];
    uint2 pf[D][R * FW];
#pragma unroll
    for (int d = 0; d < D; d++) {
      const size_t prow = (size_t)__builtin_amdgcn_readlane(rec.parent, d < cnt ? d : cnt - 1) * n;
#pragma unroll
      for (int r = 0; r < R; r++) pd[d][r] = SO ? cs_value(0) : states_in[prow + vcl[r]];
#pragma unroll
      for (int q = 0; q < R * FW; q++)
        pf[d][q] = have_in ? forb_in2[(prow + vcl[q / FW]) * FW + (q % FW)] : make_uint2(0u, 0u);
    }

    /* the node loop is unrolled D times so that the prefetch registers keep their roles: slot dd is
     * consumed by node j and refilled for node j + D at the end of that node, nothing is moved */
    for (int j0 = 0; j0 < cnt; j0 += D) {
#pragma unroll
      for (int dd = 0; dd < D; dd++) {
        const int j = j0 + dd;
        if (j >= cnt) continue;
        const int nvar = __builtin_amdgcn_readlane(rec.var, j);
        const int nlo = __builtin_amdgcn_readlane(rec.lo, j), nhi = __builtin_amdgcn_readlane(rec.hi, j);
        int lo[R], hi[R];
        unsigned fb[R][NW]; /* word q of variable r: values b0 + 32 q ... b0 + 32 q + 31 */
#pragma unroll
        for (int r = 0; r < R; r++) {
          lo[r] = live[r] ? pd[dd][r].lo : 0;
          hi[r] = live[r] ? pd[dd][r].hi : 0;
#pragma unroll
          for (int k = 0; k < FW; k++) {
            /* a lane without a variable behaves like a variable fixed at its window's first value */
            fb[r][2 * k] = live[r] ? pf[dd][r * FW + k].x : (k == 0 ? 0xfffffffeu : 0xffffffffu);
            fb[r][2 * k + 1] = live[r] ? pf[dd][r * FW + k].y : 0xffffffffu;
          }
        }

        if (SO) {
          /* the state IS the sets: every value outside a variable's interval is marked in its own set, so the
           * interval is [lowest allowed, highest allowed] of the whole window */
#pragma unroll
          for (int r = 0; r < R; r++) {
            int first, last;
            cs_set_bounds_all<NW>(fb[r], &first, &last);
            lo[r] = live[r] ? b0[r] + first : 0; /* a state with an empty set of allowed values is not a valid input */
            hi[r] = live[r] ? b0[r] + last : 0;
          }
        }

        /* the assignment (step_enter, csolve.c:294-304) and the first set of variables to push */
        unsigned long long push[R];
        const bool all_push = !have_in || nvar < 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (SO || (flags & CS_K4_OUT_RESTRICT)) { /* uniform: the interval layout skips this altogether */
            if (nvar >= 0 && (nvar >> 6) == r && lane == (nvar & 63))
              cs_set_restrict<NW>(fb[r], nlo - b0[r], nhi - b0[r]); /* the assignment becomes part of the set */
          }
          if (nvar >= 0 && (nvar >> 6) == r && lane == (nvar & 63)) { lo[r] = nlo; hi[r] = nhi; }
          if (all_push) push[r] = __ballot(lo[r] == hi[r] && live[r]);
          else push[r] = (nlo == nhi && (nvar >> 6) == r) ? 1ull << (nvar & 63) : 0ull; /* scalar */
        }
        int lo0[R], hi0[R];
#pragma unroll
        for (int r = 0; r < R; r++) { lo0[r] = lo[r]; hi0[r] = hi[r]; }

        int rounds = 0, failed = 0, revisions = 0;
        for (;;) {
          /* (1) newly valued variables push their forbidden value into the neighbours' sets */
#pragma unroll
          for (int r = 0; r < R; r++) {
            unsigned long long bits = push[r];
            while (bits != 0ull) {
              const int ul = __builtin_ctzll(bits);
              bits &= bits - 1ull;
              const int cd = __builtin_amdgcn_readlane(lo[r], ul) - dmin;
              revisions += __builtin_amdgcn_readlane(deg[r], ul);
              const E *row = s_tab + (size_t)(ul + r * CS_WAVE) * slots * W + lane;
              /* slots are read three at a time (queens: three clauses per pair); a slot index past the
               * end re-reads the last one, which is harmless because OR is idempotent */
              for (int k0 = 0; k0 < slots; k0 += 3) {
                const int k1 = k0 + 1 < slots ? k0 + 1 : slots - 1, k2 = k0 + 2 < slots ? k0 + 2 : slots - 1;
                E e[3][R];
#pragma unroll
                for (int r2 = 0; r2 < R; r2++) {
                  e[0][r2] = row[k0 * W + r2 * CS_WAVE];
                  e[1][r2] = row[k1 * W + r2 * CS_WAVE];
                  e[2][r2] = row[k2 * W + r2 * CS_WAVE];
                }
#pragma unroll
                for (int g = 0; g < 3; g++) {
#pragma unroll
                  for (int r2 = 0; r2 < R; r2++) {
                    /* the offset includes root_lo of the column; a bit outside [0, 64 FW) (also the
                     * sentinel) selects no word */
                    const unsigned bit = (unsigned)(cd - (int)e[g][r2]);
                    const unsigned sel = bit >> 5, m = 1u << (bit & 31u);
#pragma unroll
                    for (int q = 0; q < NW; q++) fb[r2][q] |= sel == (unsigned)q ? m : 0u;
                  }
                }
              }
            }
          }
          /* (2) every variable re-derives its bounds from its set: the lowest allowed value >= lo and
           * the highest allowed value <= hi, word by word */
          int fail = 0;
          unsigned long long any = 0ull;
#pragma unroll
          for (int r = 0; r < R; r++) {
            int first, last;
            if (SO) cs_set_bounds_all<NW>(fb[r], &first, &last); /* everything outside [lo, hi] is marked already */
            else cs_set_bounds<NW>(fb[r], lo[r] - b0[r], hi[r] - b0[r], &first, &last);
            const bool bad = last < 0; /* no allowed value in [lo, hi] (also when lo > hi) */
            const int nlo2 = b0[r] + first, nhi2 = b0[r] + last;
            const bool newly = !bad && (nlo2 != lo[r] || nhi2 != hi[r]) && nlo2 == nhi2;
            fail |= bad;
            lo[r] = bad ? lo[r] : nlo2;
            hi[r] = bad ? hi[r] : nhi2;
            push[r] = __ballot(newly);
            any |= push[r];
          }
          if (__any(fail)) { failed = 1; break; }
          if (any == 0ull) break;
          rounds++;
        }

        /* refill slot dd for node j + D (after the fixpoint, ahead of the stores) */
        {
          const int jn = j + D < cnt ? j + D : cnt - 1;
          const size_t prow = (size_t)__builtin_amdgcn_readlane(rec.parent, jn) * n;
#pragma unroll
          for (int r = 0; r < R; r++) pd[dd][r] = SO ? cs_value(0) : states_in[prow + vcl[r]];
#pragma unroll
          for (int q = 0; q < R * FW; q++)
            pf[dd][q] = have_in ? forb_in2[(prow + vcl[q / FW]) * FW + (q % FW)] : make_uint2(0u, 0u);
        }

        int open_vars = 0, shaved = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
          open_vars += __popcll(__ballot(lo[r] != hi[r]));
          shaved += (lo[r] - lo0[r]) + (hi0[r] - hi[r]);
        }
        const int props = cs_wave_sum(shaved);
        const size_t orow = (size_t)(base + j) * n;
        if (flags & CS_K4_OUT_RESTRICT) { /* packing: the interval goes into the set */
#pragma unroll
          for (int r = 0; r < R; r++) cs_set_restrict<NW>(fb[r], lo[r] - b0[r], hi[r] - b0[r]);
        }
        if (FAST) {
#pragma unroll
          for (int r = 0; r < R; r++)
            if (!SO) states_out[orow + lane + r * CS_WAVE] = cs_interval(lo[r], hi[r]);
#pragma unroll
          for (int q = 0; q < R * FW; q++)
            forb_out2[(orow + lane + (q / FW) * CS_WAVE) * FW + (q % FW)] =
                make_uint2(fb[q / FW][2 * (q % FW)], fb[q / FW][2 * (q % FW) + 1]);
        } else if (!failed) {
#pragma unroll
          for (int r = 0; r < R; r++)
            if (live[r] && !SO && states_out != nullptr) states_out[orow + lane + r * CS_WAVE] = cs_interval(lo[r], hi[r]);
          if (forb_out != nullptr) {
#pragma unroll
            for (int q = 0; q < R * FW; q++)
              if (live[q / FW])
                forb_out2[(orow + lane + (q / FW) * CS_WAVE) * FW + (q % FW)] =
                    make_uint2(fb[q / FW][2 * (q % FW)], fb[q / FW][2 * (q % FW) + 1]);
          }
        }
        if (lane == j) {
          my_result.status = failed ? -1 : open_vars;
          my_result.props = props;
          my_result.revisions = revisions;
          my_result.rounds = rounds;
        }
      }
    }
    if (lane < cnt) results[base + lane] = my_result;
  }
}

/* ---- kernel 5: kernel 4 for models of at most 32 variables, G nodes per wave ---------------------
 *
 * A model of n <= 64 / G variables with 64-value windows leaves most lanes of kernel 4 idle (queens-16: 16 of
 * 64).  Here a wave carries G = 2 or 4 nodes at once, node g in lanes g S .. g S + S - 1 (S = 64 / G).  Every
 * segment runs its own event queue: the ballot of newly valued variables is cut into G segments on the scalar
 * unit, each lane follows the lowest set bit of its own segment, fetches that variable's value with
 * ds_bpermute and reads that variable's table row.  A segment whose node failed stops pushing; the wave leaves
 * the loop when no segment has anything left.  Results per node are those of kernel 4 (status, props,
 * revisions, rounds).
 *
 * The kernel is VALU-bound (the search engine's children are mostly cut, so little is stored), hence:
 *  - per-node flags (failed, pushed, rounds, open variables) are wave masks / packed fields in SGPRs, turned
 *    into lane predicates with the inverse ballot; only bounds and sets are per-lane arithmetic;
 *  - bounds are kept relative to the variable's root lower bound (= bit positions of its set);
 *  - NW = 1: every root domain has at most 32 values, one set word per variable (the high word of the stored
 *    u64 is passed through); S3: exactly three table slots per pair (queens), read with immediate offsets.
 * Set bits of values outside a variable's root domain are unspecified (kernels 3 and 4 mark pushes that land
 * there, NW = 1 does not); they never influence a result. */
template <int S>
__device__ __forceinline__ int cs_segment_sum(int x) { /* valid in the last lane of every S-lane segment */
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); /* row_shr:1 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); /* row_shr:2 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); /* row_shr:4 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); /* row_shr:8 */
  if (S == 32) x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true); /* row_bcast:15 into rows 1 and 3 */
  return x;
}

/* wave mask -> bit 0 of every segment that has a bit set (uniform: scalar unit, no loops over segments) */
template <int G>
__device__ __forceinline__ unsigned long long cs_segments_low_bit(unsigned long long m) {
  constexpr int S = CS_WAVE / G;
  constexpr unsigned long long HI = G == 4 ? 0x8000800080008000ull : 0x8000000080000000ull;
  return ((((m & ~HI) + ~HI) | m) & HI) >> (S - 1);
}

/* ... -> every segment that has a bit set, filled */
template <int G>
__device__ __forceinline__ unsigned long long cs_segments_any(unsigned long long m) {
  constexpr int S = CS_WAVE / G;
  const unsigned long long low = cs_segments_low_bit<G>(m), hi = low << (S - 1);
  return (hi - low) | hi;
}

/* this lane's S-bit field of a uniform 64-bit word (32-bit operations only) */
template <int G>
__device__ __forceinline__ unsigned cs_segment_field(unsigned long long x, int g) {
  const unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
  if (G == 2) return g ? hi : lo;
  const unsigned h = (g & 2) ? hi : lo;
  return (g & 1) ? h >> 16 : h & 0xffffu;
}

/* minimum over the S lanes of a segment, in every lane of it: DPP row rotations (a row = 16 lanes), plus one
 * swizzle that swaps the two rows of a 32-lane segment */
template <int S>
__device__ __forceinline__ unsigned cs_segment_min(unsigned x) {
  /* v_min_u32 with the rotated operand read through DPP (the compiler emits mov_dpp + min + copy for the
   * same thing); a DPP read needs two wait states after the VALU write of its source */
  asm("s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
      : "+v"(x));
  if (S == 32) {
    const unsigned y = (unsigned)__builtin_amdgcn_ds_swizzle((int)x, 0x401f); /* lane ^ 16 */
    x = y < x ? y : x;
  }
  return x;
}

/* This kernel's table (built at finalize from the dense table) has 16-bit entries relative to the PUSHING
 * variable's root lower bound: with u at relative value r, bit r + bias - t[u][slot][w] of w's set is forbidden
 * (bias = the largest root_lo - dmin keeps every entry non-negative; 0xffff = no clause).  The pushed variable and its value then
 * travel together in one key (variable in the top bits, r + bias below), and the segment minimum of the keys of
 * the pending lanes IS the next event: no ds_bpermute, no scalar bookkeeping per event. */
template <int G, int NW, bool S3>
__global__ __launch_bounds__(1024, 8) void cs_propagate_ne_packed(
    int n, const unsigned short *__restrict__ tab_g /* the relative table */, int slots, int dmin /* unused */,
    const int *__restrict__ root_lo, const int *__restrict__ sym_off, const cs_val *__restrict__ states_in,
    const unsigned long long *__restrict__ forb_in, const cs_node_in *__restrict__ nodes,
    cs_val *__restrict__ states_out, unsigned long long *__restrict__ forb_out, cs_node_out *__restrict__ results,
    long long batch, const unsigned long long *__restrict__ batch_dev, int bias, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  constexpr int S = CS_WAVE / G;
  constexpr int W = CS_WAVE; /* columns of the table */
  constexpr int TOP = 32 * NW - 1; /* highest relative value */
  constexpr unsigned KEY_VALUE = (1u << 26) - 1u; /* key = variable << 26 | relative value + bias */
  if (batch_dev != nullptr && (long long)*batch_dev < batch) batch = (long long)*batch_dev;
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves_per_block = blockDim.x >> 6;
  if ((long long)blockIdx.x * waves_per_block * G >= batch) return; /* no node for this workgroup: skip the table copy */
  unsigned short *s_tab = (unsigned short *)cs_lds;
  {
    const int vecs = (int)(((size_t)n * slots * W * 2) / 16);
    const uint4 *src = (const uint4 *)tab_g;
    uint4 *dst = (uint4 *)cs_lds;
    for (int i = threadIdx.x; i < vecs; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();

  const int g = lane / S, v = lane & (S - 1);
  const bool live = v < n;
  const int vcl = live ? v : n - 1;
  const int b0 = live ? root_lo[vcl] : 0;
  const int deg = live ? sym_off[vcl + 1] - sym_off[vcl] : 0;
  const unsigned key_base = ((unsigned)v << 26) + (unsigned)bias;
  /* one segment sum for propagations and revisions when both stay below 2^15: S variables x 64 values, and the
   * sum of all degrees (uniform) */
  const bool pack_sums = sym_off[n] < 32768;
  const int row_stride = slots * W * 2;
  const bool have_in = forb_in != nullptr;
  const uint2 *forb_in2 = (const uint2 *)forb_in;
  uint2 *forb_out2 = (uint2 *)forb_out;

  const long long groups = (batch + G - 1) / G;
  const long long waves_total = (long long)gridDim.x * waves_per_block;
  long long q = (long long)blockIdx.x * waves_per_block + wave_in_block;
  if (q >= groups) return;

  /* one group of G nodes ahead; a node past the end of the batch re-reads the last node and stores nothing */
  cs_node_in rec_n;
  cs_val pd_n;
  uint2 pf_n;
  {
    const long long node = q * G + g < batch ? q * G + g : batch - 1;
    rec_n = nodes[node];
    const size_t prow = (size_t)rec_n.parent * n + vcl;
    pd_n = states_in[prow];
    pf_n = have_in ? forb_in2[prow] : make_uint2(0u, 0u);
  }
  for (; q < groups; q += waves_total) {
    const long long node = q * G + g;
    const bool valid = node < batch;
    const int nvar = rec_n.var, nlo = rec_n.lo, nhi = rec_n.hi;
    /* bounds relative to the root lower bound = bit positions; a lane without a variable behaves like a
     * variable fixed at its window's first value */
    int rl = live ? pd_n.lo - b0 : 0, rh = live ? pd_n.hi - b0 : 0;
    rl = rl < 0 ? 0 : (rl > TOP ? TOP : rl); /* states outside the root domain are not valid input: stay defined */
    rh = rh < 0 ? 0 : (rh > TOP ? TOP : rh);
    unsigned fb[2];
    fb[0] = live ? pf_n.x : 0xfffffffeu;
    fb[1] = live ? pf_n.y : 0xffffffffu;
    {
      const long long qn = q + waves_total < groups ? q + waves_total : q;
      const long long nn = qn * G + g < batch ? qn * G + g : batch - 1;
      rec_n = nodes[nn];
      const size_t prow = (size_t)rec_n.parent * n + vcl;
      pd_n = states_in[prow];
      pf_n = have_in ? forb_in2[prow] : make_uint2(0u, 0u);
    }

    /* the assignment (step_enter, csolve.c:294-304) and the first variables to push; relative to the root lower
     * bound (saturating: an assignment may carry the +-infinity sentinels) */
    const bool mine = nvar >= 0 && v == nvar;
    const int from = __builtin_elementwise_sub_sat(nlo, b0), to = __builtin_elementwise_sub_sat(nhi, b0);
    const bool gone = from > TOP || to < 0 || from > to; /* nothing of the root domain left */
    if (mine) {
      if ((flags & CS_K4_OUT_RESTRICT) && !gone) cs_set_restrict<2>(fb, from, to);
      fb[0] = gone ? 0xffffffffu : fb[0];
      fb[1] = gone ? 0xffffffffu : fb[1];
      rl = gone ? rl : (from < 0 ? 0 : from);
      rh = gone ? rh : (to > TOP ? TOP : to);
    }
    /* kernel 4's reference point for the count of propagations: the assigned interval itself (relative) */
    const int rl0 = mine ? from : rl, rh0 = mine ? to : rh;
    bool pending = live && rl == rh && (!have_in || nvar < 0 || mine);
    unsigned long long failedm = 0ull, pushedm = __ballot(pending);
    unsigned long long rounds_fields = 0ull; /* one S-bit counter per segment */
    for (;;) {
      /* (1) newly valued variables push their forbidden value into the sets of their own node */
      while (__ballot(pending) != 0ull) {
        const unsigned key = cs_segment_min<S>(pending ? key_base + (unsigned)rl : 0xffffffffu);
        const int ul = (int)(key >> 26);           /* 63: nothing pending in this segment */
        const int cd = (int)(key & KEY_VALUE);     /* then 2^26 - 1: selects no bit below */
        pending = pending && v != ul;
        const int ulc = ul < n ? ul : n - 1;
        const unsigned char *row = (const unsigned char *)s_tab + ulc * row_stride + v * 2;
        if (S3) {
          const int e0 = (int)*(const unsigned short *)(row), e1 = (int)*(const unsigned short *)(row + W * 2),
                    e2 = (int)*(const unsigned short *)(row + 2 * W * 2);
          const unsigned b0_ = (unsigned)(cd - e0), b1_ = (unsigned)(cd - e1), b2_ = (unsigned)(cd - e2);
          fb[0] |= (b0_ < 32u ? 1u << b0_ : 0u) | (b1_ < 32u ? 1u << b1_ : 0u) | (b2_ < 32u ? 1u << b2_ : 0u);
          if (NW == 2)
            fb[1] |= ((b0_ >> 5) == 1u ? 1u << (b0_ & 31u) : 0u) | ((b1_ >> 5) == 1u ? 1u << (b1_ & 31u) : 0u) |
                     ((b2_ >> 5) == 1u ? 1u << (b2_ & 31u) : 0u);
        } else {
          for (int k = 0; k < slots; k++) {
            const unsigned bit = (unsigned)(cd - (int)*(const unsigned short *)(row + (size_t)k * W * 2));
            fb[0] |= bit < 32u ? 1u << bit : 0u;
            if (NW == 2) fb[1] |= (bit >> 5) == 1u ? 1u << (bit & 31u) : 0u;
          }
        }
      }
      /* (2) bounds from the sets: lowest and highest allowed position within [rl, rh] */
      int first, last;
      bool bad;
      if (NW == 1) {
        const unsigned a = ~fb[0] & (~0u << rl) & (~0u >> (31 - rh));
        bad = a == 0u;
        first = __builtin_ctz(a | 0x80000000u); /* defined for a == 0 as well */
        last = 31 - __builtin_clz(a | 1u);
      } else {
        cs_set_bounds<2>(fb, rl, rh, &first, &last);
        bad = last < 0;
      }
      const bool newly = !bad && first == last && rl != rh;
      rl = bad ? rl : first;
      rh = bad ? rh : last;
      const unsigned long long badm = __ballot(bad);
      if (badm != 0ull) failedm |= cs_segments_any<G>(badm);
      const unsigned long long newm = __ballot(newly) & ~failedm;
      if (newm == 0ull) break;
      pending = __builtin_amdgcn_inverse_ballot_w64(newm);
      pushedm |= newm;
      rounds_fields += cs_segments_low_bit<G>(newm);
    }

    const int open_vars = __popc(cs_segment_field<G>(__ballot(rl != rh), g));
    const int lo = b0 + rl, hi = b0 + rh;
    /* propagations (at most 63 per variable, 32 variables) and revisions (at most the sum of all degrees) share
     * one segment sum, 16 bits each, when the latter fits */
    const int shaved = (rl - rl0) + (rh0 - rh);
    const int pushed_deg = __builtin_amdgcn_inverse_ballot_w64(pushedm) ? deg : 0;
    int props, revisions;
    if (pack_sums) {
      const int both = cs_segment_sum<S>((shaved << 16) + pushed_deg);
      props = both >> 16; /* shaved is non-negative for every input the kernel is specified for */
      revisions = both & 0xffff;
    } else {
      props = cs_segment_sum<S>(shaved);
      revisions = cs_segment_sum<S>(pushed_deg);
    }
    const bool failed = __builtin_amdgcn_inverse_ballot_w64(failedm);
    if (flags & CS_K4_OUT_RESTRICT) cs_set_restrict<2>(fb, rl, rh);
    const bool st = valid && live && !failed;
    const size_t orow = (size_t)node * n + v;
    if (st && states_out != nullptr) states_out[orow] = cs_interval(lo, hi);
    if (st && forb_out != nullptr) forb_out2[orow] = make_uint2(fb[0], fb[1]);
    if (valid && v == S - 1) {
      cs_node_out r;
      r.status = failed ? -1 : open_vars;
      r.props = props;
      r.revisions = revisions;
      r.rounds = (int)cs_segment_field<G>(rounds_fields, g);
      results[node] = r;
    }
  }
}

/* ---- full sweeps (root phase): one workgroup per instance ------------------------ */

/* sequential != 0: the reference's own order -- ONE thread revises the clauses one after the other, in root-element
 * order, every revision seeing what the ones before it left (propagate_wand, propagate.c:379-392, under the do-while
 * of propagate(), 474-485): a sweep is then exactly a sweep of the reference, which matters when `max_rounds` cuts
 * the iteration short of the fixpoint.  *converged (nullable) = the last sweep changed nothing. */
__global__ __launch_bounds__(CS_BLOCK) void cs_propagate_sweeps(cs_tables T, const cs_val *__restrict__ states_in,
                                                                cs_val *__restrict__ states_out,
                                                                cs_node_out *__restrict__ results, int max_rounds,
                                                                int sequential, int *__restrict__ converged) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  const int n = T.n_vars;
  cs_val *dom = (cs_val *)cs_lds;
  unsigned *flags = (unsigned *)(dom + n); /* [0] changed, [1] failed, [2] props, [3] revisions */
  const int inst = blockIdx.x;
  const cs_val *src = states_in + (size_t)inst * n;
  for (int v = threadIdx.x; v < n; v += blockDim.x) dom[v] = src[v];
  if (threadIdx.x < 4) flags[threadIdx.x] = 0u;
  __syncthreads();

  cs_ctx cx;
    cx.log = nullptr;
    cx.cur_clause = -1;
  cx.dom = dom;
  cx.mark = flags;
  cx.mark_is_flag = 1;
  cx.fail = 0;
  cx.fail_var = -1;
  cx.props = 0;
  cx.revisions = 0;
  int rounds = 0;
  for (;;) {
    const int c_first = sequential ? (threadIdx.x == 0 ? 0 : T.n_clauses) : (int)threadIdx.x;
    const int c_step = sequential ? 1 : (int)blockDim.x;
    for (int c = c_first; c < T.n_clauses && !cx.fail; c += c_step) {
      const int4 rec = T.clause[c];
      if (rec.x == CS_CL_NE) {
        cs_ne_revise(cx, rec.y, rec.z, rec.w);
      } else if (rec.x == CS_CL_EQ || rec.x == CS_CL_LT) {
        cs_lin_revise(cx, rec.y, rec.z, rec.w, rec.x == CS_CL_EQ ? CS_REL_EQ : CS_REL_LT);
      } else if (rec.x == CS_CL_OR2) {
        cs_or2_revise(cx, T.lit + rec.y);
      } else if (rec.x == CS_CL_TREE) {
        cs_tree_scratch S;
        cs_tree_revise(T, rec.y, cx, S);
      }
    }
    if (cx.fail) flags[1] = 1u;
    __syncthreads();
    /* an interval emptied by racing updates of lo and hi */
    for (int v = threadIdx.x; v < n; v += blockDim.x)
      if (dom[v].lo > dom[v].hi) flags[1] = 1u;
    __syncthreads();
    const unsigned changed = flags[0], failed = flags[1];
    __syncthreads();
    rounds++;
    if (failed || !changed || rounds >= max_rounds) break;
    if (threadIdx.x == 0) flags[0] = 0u;
    __syncthreads();
  }
  atomicAdd(&flags[2], (unsigned)cx.props);
  atomicAdd(&flags[3], (unsigned)cx.revisions);
  __syncthreads();
  cs_val *dst = states_out + (size_t)inst * n;
  for (int v = threadIdx.x; v < n; v += blockDim.x) dst[v] = dom[v];
  if (threadIdx.x == 0) {
    cs_node_out r;
    r.status = flags[1] ? -1 : 0;
    r.props = (int)flags[2];
    r.revisions = (int)flags[3];
    r.rounds = rounds;
    results[inst] = r;
    if (converged != nullptr) converged[inst] = flags[1] == 0u && flags[0] == 0u;
  }
}

/* ---- three-valued evaluation of the root wide-and (eval.c:233-255) ---------------- */

/* interval value of clause c on the domains `dom` (eval_<op> of the clause's root) */
template <bool HAS_TREE = true>
__device__ __forceinline__ cs_val cs_eval_record(const cs_tables &T, const int4 rec, const cs_val *dom) {
  cs_val v = cs_value(1);
  if (rec.x == CS_CL_NE) {
    /* NOT(EQ(X_a, X_b + d)); no saturation possible on this path (cs_device.h) */
    cs_val a = dom[rec.y], b = dom[rec.z];
    b.lo += rec.w;
    b.hi += rec.w;
    v = cs_ev_not(cs_ev_eq(a, b));
  } else if (rec.x == CS_CL_EQ || rec.x == CS_CL_LT) {
    v = rec.x == CS_CL_EQ ? cs_ev_eq_shifted(dom[rec.y], dom[rec.z], rec.w) : cs_ev_lt_shifted(dom[rec.y], dom[rec.z], rec.w);
  } else if (rec.x == CS_CL_OR2) {
    cs_val t[2];
    for (int k = 0; k < 2; k++) {
      const int4 l = T.lit[rec.y + k];
      t[k] = cs_ev_lt_shifted(dom[l.x], dom[l.y], l.z);
    }
    v = cs_ev_or(t[0], t[1]);
  } else if (HAS_TREE && rec.x == CS_CL_TREE) { /* the interpreter's scratch costs 4.6 KB of private memory per lane */
    cs_tree_scratch S;
    const int base = T.tree_off[rec.y], len = T.tree_off[rec.y + 1] - base;
    cs_tree_eval(T, T.tnode + base, len, dom, S.val);
    v = S.val[len - 1];
  }
  return v;
}

__device__ __forceinline__ cs_val cs_eval_clause(const cs_tables &T, int c, const cs_val *dom) {
  return cs_eval_record<true>(T, T.clause[c], dom);
}

/* one WAVE per state at a time (models whose domains fit a quarter of the LDS budget): no workgroup barrier, four
 * states per workgroup in flight, grid-stride over the states.  CPL > 0: the wave keeps its CPL clause records
 * per lane in registers across its states (otherwise every state streams the clause table from L2 again: 27,000
 * states x 5.7 KB on queens-16 were the whole cost of this kernel).  list / count_dev as below. */
template <int CPL, bool HAS_TREE>
__global__ __launch_bounds__(CS_BLOCK) void cs_eval_root_waves(cs_tables T, const cs_val *__restrict__ states,
                                                               int *__restrict__ truth, const int *__restrict__ list,
                                                               const unsigned long long *__restrict__ count_dev,
                                                               long long count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  const int n = T.n_vars;
  const int lane = threadIdx.x & (CS_WAVE - 1), wave_in_block = threadIdx.x >> 6;
  if (count_dev != nullptr && (long long)*count_dev < count) count = (long long)*count_dev;
  if ((long long)blockIdx.x * CS_WAVES_PER_BLOCK >= count) return;
  cs_val *dom = (cs_val *)cs_lds + (size_t)wave_in_block * n;
  constexpr int NR = CPL > 0 ? CPL : 1;
  int4 rec[NR];
  if (CPL > 0) {
#pragma unroll
    for (int q = 0; q < NR; q++) {
      const int c = lane + q * CS_WAVE;
      rec[q] = c < T.n_clauses ? T.clause[c] : make_int4(CS_CL_SKIP, 0, 0, 0);
    }
  }
  const long long waves_total = (long long)gridDim.x * CS_WAVES_PER_BLOCK;
  for (long long inst = (long long)blockIdx.x * CS_WAVES_PER_BLOCK + wave_in_block; inst < count; inst += waves_total) {
    const cs_val *src = states + (size_t)(list != nullptr ? list[inst] : inst) * n;
    for (int v = lane; v < n; v += CS_WAVE) dom[v] = src[v];
    cs_wave_sync();
    int any_false = 0, any_open = 0;
    if (CPL > 0) {
#pragma unroll
      for (int q = 0; q < NR; q++) {
        const cs_val v = cs_eval_record<HAS_TREE>(T, rec[q], dom);
        any_false |= cs_is_false(v);
        any_open |= !cs_is_false(v) && !cs_is_true(v);
      }
    } else {
      for (int c = lane; c < T.n_clauses; c += CS_WAVE) {
        const cs_val v = cs_eval_record<HAS_TREE>(T, T.clause[c], dom);
        any_false |= cs_is_false(v);
        any_open |= !cs_is_false(v) && !cs_is_true(v);
      }
    }
    const bool f = __any(any_false), o = __any(any_open);
    if (lane == 0) truth[inst] = f ? 0 : (o ? 2 : 1);
    cs_wave_sync();
  }
}

/* list (nullable): instance i is row list[i] of states; count_dev (nullable): the number of instances, on the
 * device (the grid is then sized for an upper bound) */
__global__ __launch_bounds__(CS_BLOCK) void cs_eval_root(cs_tables T, const cs_val *__restrict__ states,
                                                         int *__restrict__ truth, const int *__restrict__ list,
                                                         const unsigned long long *__restrict__ count_dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  const int n = T.n_vars;
  cs_val *dom = (cs_val *)cs_lds;
  unsigned *flags = (unsigned *)(dom + n); /* [0] some clause false, [1] some clause undecided */
  const int inst = blockIdx.x;
  if (count_dev != nullptr && (unsigned long long)inst >= *count_dev) return;
  const cs_val *src = states + (size_t)(list != nullptr ? list[inst] : inst) * n;
  for (int v = threadIdx.x; v < n; v += blockDim.x) dom[v] = src[v];
  if (threadIdx.x < 2) flags[threadIdx.x] = 0u;
  __syncthreads();
  int any_false = 0, any_open = 0;
  for (int c = threadIdx.x; c < T.n_clauses; c += blockDim.x) {
    const cs_val v = cs_eval_clause(T, c, dom);
    any_false |= cs_is_false(v);
    any_open |= !cs_is_false(v) && !cs_is_true(v);
  }
  if (any_false) flags[0] = 1u;
  if (any_open) flags[1] = 1u;
  __syncthreads();
  if (threadIdx.x == 0) truth[inst] = flags[0] ? 0 : (flags[1] ? 2 : 1);
}

/* interval value of every clause for one state (eval_<op> per clause root) */
__global__ __launch_bounds__(CS_BLOCK) void cs_eval_clauses(cs_tables T, const cs_val *__restrict__ state,
                                                            cs_val *__restrict__ vals) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  const int n = T.n_vars;
  cs_val *dom = (cs_val *)cs_lds;
  for (int v = threadIdx.x; v < n; v += blockDim.x) dom[v] = state[v];
  __syncthreads();
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < T.n_clauses; c += gridDim.x * blockDim.x) {
    vals[c] = cs_eval_clause(T, c, dom);
  }
}

/* sets-only states -> intervals: [lowest allowed, highest allowed] per variable ({1, 0} when nothing is allowed) */
template <int FW>
__global__ void cs_sets_unpack(int n, const int *__restrict__ root_lo, const unsigned long long *__restrict__ sets,
                               cs_val *__restrict__ states, long long count) {
  const long long total = count * n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % n);
    unsigned w[2 * FW];
#pragma unroll
    for (int k = 0; k < FW; k++) {
      const unsigned long long x = sets[i * FW + k];
      w[2 * k] = (unsigned)x;
      w[2 * k + 1] = (unsigned)(x >> 32);
    }
    int first, last;
    cs_set_bounds<2 * FW>(w, 0, 64 * FW - 1, &first, &last);
    states[i] = last < 0 ? cs_interval(1, 0) : cs_interval(root_lo[v] + first, root_lo[v] + last);
  }
}

#endif /* CS_KERNELS_HIP_H */
