/* cs_chain.hip.h -- the reference's OWN failure chain of one node, for the drop-in's exact mode.
 *
 * With -f true (its default) the reference's driver prefers variables that failed: a failing propagate_clauses bumps
 * the priority of the variable whose domain emptied (propagate_term_confl, reference src/propagate.c:33-41) and of
 * every variable on its recursion stack (propagate_term_recurse, 44-54).  Which variables those are depends on the
 * exact depth-first order of the reference's propagation (propagate_clauses, 488-538):
 *   - a variable's clause list is revised in order; a clause a nested call has revised meanwhile is skipped (prop_tag);
 *   - every narrowing recurses AT ONCE into the narrowed variable's list (propagate_term, 57-87);
 *   - a revision of NOT(EQ(l, r)) evaluates l and r ONCE, pushes onto r, and after that recursion has returned pushes
 *     onto l with the values it evaluated before (propagate_eq_false, 123-136; propagate_eq_false_lr, 106-120);
 *   - pushing onto `x + c` checks the constant first (propagate_add, 234-246): a push that empties `x + c` fails at the
 *     constant, which has no variable to bump; pushing onto a bare `x` fails at x and bumps it.
 * The parallel fixpoint kernels find the same fixpoints and verdicts but not this sequence, so the drop-in normally
 * bumps a causal chain read off the device's trail (cs_dropin.c).  This kernel IS the sequence, for pure != networks
 * whose clauses are NOT(EQ(l, r)) with l, r a variable or `variable + constant`: ONE wavefront walks the reference's
 * recursion with an explicit stack of frames; the only parallel step is the scan of a clause list, 64 clauses at a
 * time, for the first clause that does anything -- clauses that do nothing change no domain, so looking at 64 of them
 * against the same state is what the sequential loop would have seen.  It is run for FAILING nodes only (the verdict
 * comes from the fast path), and only when the drop-in is asked for the reference's exact trace
 * (CSOLVE_DROPIN_CHAIN=reference): a node costs tens of microseconds here, a few in the fast path.
 */
#ifndef CS_CHAIN_HIP_H
#define CS_CHAIN_HIP_H

#include "cs_kernels.hip.h"

/* one clause NOT(EQ(l, r)): x = variable | add flag << 30; the constants of `variable + constant` operands */
struct cs_chain_clause {
  int lx, lc, rx, rc;
};
#define CS_CHAIN_ADD (1 << 30)
#define CS_CHAIN_VAR(x) ((x) & ~CS_CHAIN_ADD)

struct cs_chain_frame {
  int var, pos, tag, phase; /* phase 0: scanning from pos; 1 / 2: the clause at pos, before its first / second push; 3: after it */
  int clause, llo, lhi, rlo; /* the values evaluated when the clause was entered (propagate_eq_false takes them once) */
  int rhi, pad0, pad1, pad2;
};

/* out[0] = status (-1 failed, 0 consistent), out[1] = narrowings made (the reference's PROPS of the call),
 * out[2] = variables bumped, out[3] = 1 if the frame stack or the bump list overflowed;
 * bumps[0 .. out[2]) = the variables in the order the reference bumps them */
__global__ __launch_bounds__(64) void cs_ne_chain(int n, int n_clauses, const cs_chain_clause *__restrict__ cl,
                                                  const int *__restrict__ list_off, const int *__restrict__ list,
                                                  const cs_val *__restrict__ state_in, cs_node_in node,
                                                  cs_chain_frame *__restrict__ frames, int frame_cap,
                                                  int *__restrict__ out, int *__restrict__ bumps, int bump_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  cs_val *dom = (cs_val *)cs_lds;
  unsigned short *tags = (unsigned short *)(dom + n);
  const int lane = threadIdx.x;
  for (int v = lane; v < n; v += 64) dom[v] = state_in[v];
  for (int c = lane; c < n_clauses; c += 64) tags[c] = 0;
  cs_wave_sync();
  if (lane == 0 && node.var >= 0) dom[node.var] = cs_interval(node.lo, node.hi);
  cs_wave_sync();

  int depth = 0, tagc = 0, props = 0, nb = 0, overflow = 0, failed = 0;
  /* the frame on top of the stack, in (scalar) registers */
  int f_var = node.var, f_pos = 0, f_tag = ++tagc, f_phase = 0, f_clause = -1, f_llo = 0, f_lhi = 0, f_rlo = 0, f_rhi = 0;

  auto bump = [&](int v) {
    if (nb < bump_cap) { if (lane == 0) bumps[nb] = v; }
    else overflow = 1;
    nb++;
  };
  /* evaluate an operand: a variable's domain, shifted when the operand is `variable + constant` (eval_add, eval.c:117-135) */
  auto operand = [&](int x, int c) -> cs_val {
    const cs_val d = dom[CS_CHAIN_VAR(x)];
    return (x & CS_CHAIN_ADD) ? cs_ev_add(d, cs_value(c)) : d;
  };

  for (;;) {
    f_var = __builtin_amdgcn_readfirstlane(f_var); f_pos = __builtin_amdgcn_readfirstlane(f_pos);
    f_tag = __builtin_amdgcn_readfirstlane(f_tag); f_phase = __builtin_amdgcn_readfirstlane(f_phase);
    f_clause = __builtin_amdgcn_readfirstlane(f_clause);
    depth = __builtin_amdgcn_readfirstlane(depth); tagc = __builtin_amdgcn_readfirstlane(tagc);
    if (f_phase == 0) {
      const int beg = list_off[f_var], len = list_off[f_var + 1] - beg;
      if (f_pos >= len) {
        /* this propagate_clauses is done: back to the revision that recursed into it */
        if (depth == 0) break;
        depth--;
        const cs_chain_frame fr = frames[depth];
        f_var = fr.var; f_pos = fr.pos; f_tag = fr.tag; f_phase = fr.phase; f_clause = fr.clause;
        f_llo = fr.llo; f_lhi = fr.lhi; f_rlo = fr.rlo; f_rhi = fr.rhi;
        continue;
      }
      /* 64 clauses of the list at once: which is the first that does anything? */
      const int i = f_pos + lane;
      const bool valid = i < len;
      const int c = valid ? list[beg + i] : 0;
      const bool fresh = valid && (int)tags[c] <= f_tag; /* not revised by a nested (later) call */
      bool acts = false;
      cs_val lval = cs_value(0), rval = cs_value(0);
      if (fresh && cl[c].lx >= 0) {
        const cs_chain_clause k = cl[c];
        lval = operand(k.lx, k.lc);
        rval = operand(k.rx, k.rc);
        const bool a1 = lval.lo == lval.hi && lval.lo != CS_DOM_MIN && lval.lo != CS_DOM_MAX && (lval.lo == rval.lo || lval.lo == rval.hi);
        const bool a2 = rval.lo == rval.hi && rval.lo != CS_DOM_MIN && rval.lo != CS_DOM_MAX && (rval.lo == lval.lo || rval.lo == lval.hi);
        acts = a1 || a2;
      }
      const unsigned long long m = __ballot(acts);
      const int first = m != 0ull ? __builtin_ctzll(m) : 64;
      if (fresh && lane <= first) tags[c] = (unsigned short)f_tag; /* revised (the acting one included) */
      cs_wave_sync();
      if (m == 0ull) { f_pos += 64; continue; }
      f_pos += first;
      f_clause = __builtin_amdgcn_readlane(c, first);
      f_llo = __builtin_amdgcn_readlane(lval.lo, first); f_lhi = __builtin_amdgcn_readlane(lval.hi, first);
      f_rlo = __builtin_amdgcn_readlane(rval.lo, first); f_rhi = __builtin_amdgcn_readlane(rval.hi, first);
      f_phase = 1;
      continue;
    }
    if (f_phase == 1 || f_phase == 2) {
      /* phase 1: propagate_eq_false_lr(r, rval, lval): the left value onto the right side; phase 2: the right value
       * (as evaluated BEFORE phase 1) onto the left side */
      const cs_chain_clause k = cl[f_clause];
      const int olo = f_phase == 1 ? f_llo : f_rlo, ohi = f_phase == 1 ? f_lhi : f_rhi; /* the other side's value */
      const int plo = f_phase == 1 ? f_rlo : f_llo, phi = f_phase == 1 ? f_rhi : f_lhi; /* this side's, as evaluated */
      const int px = f_phase == 1 ? k.rx : k.lx, pc = f_phase == 1 ? k.rc : k.lc;
      int narrowed = -1;
      if (olo == ohi && olo != CS_DOM_MIN && olo != CS_DOM_MAX && (olo == plo || olo == phi)) {
        cs_val want = olo == plo ? cs_interval(olo + 1, CS_DOM_MAX) : cs_interval(CS_DOM_MIN, olo - 1);
        const int y = CS_CHAIN_VAR(px);
        const cs_val d = dom[y];
        if (px & CS_CHAIN_ADD) {
          /* propagate_add (propagate.c:234-246): first the constant -- it must lie in want - eval(y) -- then y */
          const int lo = cs_add(want.lo, cs_neg(d.hi)), hi = cs_add(want.hi, cs_neg(d.lo));
          if (pc < lo || pc > hi) { failed = 1; break; } /* a terminal without a variable: nobody to bump */
          want = cs_interval(cs_add(want.lo, cs_neg(pc)), cs_add(want.hi, cs_neg(pc)));
        }
        /* propagate_term (propagate.c:57-87) */
        if (d.lo > want.hi || d.hi < want.lo) {
          bump(y); /* propagate_term_confl */
          failed = 1;
          break;
        }
        const int lo = d.lo > want.lo ? d.lo : want.lo, hi = d.hi < want.hi ? d.hi : want.hi;
        if (lo != d.lo || hi != d.hi) {
          if (lane == 0) dom[y] = cs_interval(lo, hi);
          cs_wave_sync();
          props++;
          narrowed = y;
        }
      }
      f_phase++;
      if (narrowed >= 0) {
        /* propagate_term_recurse: the narrowed variable's clause list, now */
        if (depth >= frame_cap) { overflow = 1; failed = 1; break; }
        if (lane == 0) {
          cs_chain_frame fr;
          fr.var = f_var; fr.pos = f_pos; fr.tag = f_tag; fr.phase = f_phase; fr.clause = f_clause;
          fr.llo = f_llo; fr.lhi = f_lhi; fr.rlo = f_rlo; fr.rhi = f_rhi; fr.pad0 = 0; fr.pad1 = 0; fr.pad2 = 0;
          frames[depth] = fr;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        depth++;
        f_var = narrowed; f_pos = 0; f_tag = ++tagc; f_phase = 0; f_clause = -1;
        if (tagc >= 65535) { overflow = 1; failed = 1; break; }
      }
      continue;
    }
    /* phase 3: the clause is done, on with the list */
    f_pos++;
    f_phase = 0;
  }
  if (failed && !overflow) {
    /* every propagate_clauses on the stack returns the error: propagate_term_recurse bumps its variable, innermost
     * first (the outermost call is check_assignment's own: the driver bumps that variable itself, csolve.c:462) */
    bump(f_var);
    for (int d = depth - 1; d >= 1; d--) bump(frames[d].var);
    if (depth == 0) nb--; /* the top frame WAS the outermost call: nothing recursed into it */
  }
  if (lane == 0) {
    out[0] = failed ? -1 : 0;
    out[1] = props;
    out[2] = nb < 0 ? 0 : nb;
    out[3] = overflow;
  }
}

#endif
