/* cs_shave.hip.h -- kernel 7: the batched propagate_clauses fixpoint on interval states alone.
 *
 * Pure binary-!= models of at most 256 variables whose symmetric relation fits the dense table
 * tab[u][slot][w] in LDS (cs_device.h; the table of kernel 4).  A node reads its parent's intervals
 * (8 B per variable), writes its own, and carries nothing else: 16 n + 32 bytes over HBM, the bytes
 * propagate_clauses itself is defined on (`struct val_t` per variable in, per variable out).
 *
 * What the reference computes on such a network (propagate_not -> propagate_eq(false) ->
 * propagate_eq_false_lr, propagate.c:289-301, 123-136, 106-120): a variable that is a single value
 * removes that value (shifted by the clause's constants) from a neighbour's domain ONLY when it sits on
 * one of the neighbour's bounds -- the bound moves by one, which is one PROPS, and the recursion
 * (propagate_term -> propagate_clauses, propagate.c:44-54) re-examines the moved bound against the
 * neighbour's other valued neighbours.  Kernels 3 to 5 keep the set of forbidden values per variable to
 * answer "where does the bound stop"; this kernel asks the question when it arises:
 *
 *   PUSH(u)    u has just become a value c.  Lane w reads ITS entries tab[u][k][w] (one conflict-free row
 *              read per slot), gets the value f of w that u = c forbids, and moves a bound that equals f
 *              by one.  Lanes whose bound moved are "dirty".
 *   VERIFY(w)  is w's moved bound v supported?  By symmetry of the table, lane u reads tab[w][k][u] and
 *              knows which value of w ITS OWN value forbids; a ballot over the valued lanes answers for
 *              all neighbours at once.  While some valued lane forbids v, the bound moves on (one PROPS
 *              each, as in the reference's recursion).  A bound that passes the other one is the failure.
 *
 * Fixpoint, verdict and PROPS of consistent nodes are those of kernels 1 to 5 (the fixpoint of these
 * monotone propagators is unique; every bound move is one reference narrowing) -- the parity tests run
 * every case through this kernel too.  Like the reference (and kernels 1, 2) it is event-driven: only
 * consequences of the node's assignment are drawn, the parent is taken to be a fixpoint.  A node with
 * var < 0 ("propagate everything") pushes every valued variable.
 *
 * One wavefront per node, lane l owns variables l, l + 64, ... (R of them); bounds live in VGPRs relative
 * to the variable's root lower bound; masks of valued / pushed / dirty variables are 64-bit scalars; D
 * parent rows are in flight per wave (register prefetch).  LDS holds the table only.
 */
#ifndef CS_SHAVE_HIP_H
#define CS_SHAVE_HIP_H

#include "cs_kernels.hip.h"

/* SL: slots per pair known at compile time (1 = alldiff-style, 3 = queens), 0 = run-time loop.
 * FULL: n == 64 R, every lane register holds a variable: loads and stores are unconditional. */
template <typename E, int R, int D, int SL, bool FULL>
__global__ __launch_bounds__(1024, (R <= 2 ? 8 : 4)) void cs_propagate_ne_shave(
    int n, const E *__restrict__ tab_g, int slots, int dmin, const int *__restrict__ root_lo,
    const int *__restrict__ sym_off, const cs_val *__restrict__ states_in, const cs_node_in *__restrict__ nodes,
    cs_val *__restrict__ states_out, cs_node_out *__restrict__ results, long long batch,
    const unsigned long long *__restrict__ batch_dev, int csz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  typedef unsigned long long u64;
  constexpr int W = CS_WAVE * R; /* columns of the table */
  if (batch_dev != nullptr && (long long)*batch_dev < batch) batch = (long long)*batch_dev;
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves_per_block = blockDim.x >> 6;
  if ((long long)blockIdx.x * waves_per_block * csz >= batch) return; /* no node for this workgroup */
  if (SL != 0) slots = SL;
  const E *s_tab = (const E *)cs_lds;
  {
    const int vecs = (int)(((size_t)n * slots * W * sizeof(E)) / 16);
    const uint4 *src = (const uint4 *)tab_g;
    uint4 *dst = (uint4 *)cs_lds;
    for (int i = threadIdx.x; i < vecs; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();

  int b0[R], kb[R], deg[R], vcl[R];
  bool live[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int v = lane + r * CS_WAVE;
    live[r] = FULL || v < n;
    vcl[r] = live[r] ? v : n - 1; /* lanes past the end re-read the last variable and ignore it */
    b0[r] = live[r] ? root_lo[vcl[r]] : 0;
    kb[r] = b0[r] - dmin;
    deg[r] = live[r] ? sym_off[vcl[r] + 1] - sym_off[vcl[r]] : 0;
  }
  u64 livemask[R]; /* lanes that hold a variable: the others look like values and must never push or forbid */
#pragma unroll
  for (int r = 0; r < R; r++) livemask[r] = FULL ? ~0ull : __ballot(live[r]);

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

    cs_val pd[D][R]; /* D parent rows in flight; an index past the end of the chunk re-reads its last node */
#pragma unroll
    for (int d = 0; d < D; d++) {
      const size_t prow = (size_t)__builtin_amdgcn_readlane(rec.parent, d < cnt ? d : cnt - 1) * n;
#pragma unroll
      for (int r = 0; r < R; r++) pd[d][r] = states_in[prow + vcl[r]];
    }

    for (int j0 = 0; j0 < cnt; j0 += D) {
#pragma unroll
      for (int dd = 0; dd < D; dd++) {
        const int j = j0 + dd;
        if (j >= cnt) continue;
        const int nvar = __builtin_amdgcn_readlane(rec.var, j);
        const int nlo = __builtin_amdgcn_readlane(rec.lo, j), nhi = __builtin_amdgcn_readlane(rec.hi, j);
        int rlo[R], rhi[R]; /* bounds relative to the root lower bound; a lane without a variable is the value 0 */
#pragma unroll
        for (int r = 0; r < R; r++) {
          rlo[r] = live[r] ? pd[dd][r].lo - b0[r] : 0;
          rhi[r] = live[r] ? pd[dd][r].hi - b0[r] : 0;
        }
        /* refill slot dd for node j + D right away: the loads fly while this node is propagated */
        {
          const int jn = j + D < cnt ? j + D : cnt - 1;
          const size_t prow = (size_t)__builtin_amdgcn_readlane(rec.parent, jn) * n;
#pragma unroll
          for (int r = 0; r < R; r++) pd[dd][r] = states_in[prow + vcl[r]];
        }

        /* variables that are values in the parent have been pushed there (the parent is a fixpoint);
         * var < 0: nothing is taken for granted, every valued variable pushes */
        u64 pushed[R], push[R], dl[R], dh[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
          pushed[r] = nvar < 0 ? ~livemask[r] : __ballot(rlo[r] == rhi[r]);
          dl[r] = 0ull; dh[r] = 0ull;
        }
        /* the assignment (step_enter, csolve.c:294-304; an interval for the worker split, csolve.c:121-150) */
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (nvar >= 0 && (nvar >> 6) == r) {
            const u64 bit = 1ull << (nvar & 63);
            if (lane == (nvar & 63)) { rlo[r] = nlo - b0[r]; rhi[r] = nhi - b0[r]; }
            pushed[r] &= ~bit;
            if (nlo != nhi) { dl[r] |= bit; dh[r] |= bit; } /* new bounds of an open variable: to be verified */
          }
        }
        int lo0[R], hi0[R];
#pragma unroll
        for (int r = 0; r < R; r++) { lo0[r] = rlo[r]; hi0[r] = rhi[r]; }
        u64 val[R]; /* variables that are single values now */
        u64 any_push = 0ull;
#pragma unroll
        for (int r = 0; r < R; r++) {
          val[r] = __ballot(rlo[r] == rhi[r]) & livemask[r];
          push[r] = val[r] & ~pushed[r];
          any_push |= push[r];
        }

        int rounds = 0, failed = 0, revisions = 0;
        for (;;) {
          /* (1) PUSH: every newly valued variable moves the bounds it sits on */
#pragma unroll
          for (int r = 0; r < R; r++) {
            u64 bits = push[r];
            pushed[r] |= bits;
            while (bits != 0ull) {
              const int ul = __builtin_ctzll(bits);
              bits &= bits - 1ull;
              const int cd = __builtin_amdgcn_readlane(rlo[r] + kb[r], ul); /* value - dmin */
              revisions += __builtin_amdgcn_readlane(deg[r], ul);
              const E *row = s_tab + (size_t)(ul + r * CS_WAVE) * slots * W + lane;
              bool hl[R], hh[R];
#pragma unroll
              for (int r2 = 0; r2 < R; r2++) { hl[r2] = false; hh[r2] = false; }
              if (SL != 0) {
#pragma unroll
                for (int k = 0; k < (SL ? SL : 1); k++) {
#pragma unroll
                  for (int r2 = 0; r2 < R; r2++) {
                    const int f = cd - (int)row[k * W + r2 * CS_WAVE]; /* the value of w that u forbids (sentinel: < 0) */
                    hl[r2] |= f == rlo[r2];
                    hh[r2] |= f == rhi[r2];
                  }
                }
              } else {
                for (int k = 0; k < slots; k++) {
#pragma unroll
                  for (int r2 = 0; r2 < R; r2++) {
                    const int f = cd - (int)row[k * W + r2 * CS_WAVE];
                    hl[r2] |= f == rlo[r2];
                    hh[r2] |= f == rhi[r2];
                  }
                }
              }
#pragma unroll
              for (int r2 = 0; r2 < R; r2++) {
                rlo[r2] += hl[r2] ? 1 : 0;
                rhi[r2] -= hh[r2] ? 1 : 0;
                dl[r2] |= __ballot(hl[r2]);
                dh[r2] |= __ballot(hh[r2]);
              }
            }
          }
          /* (2) VERIFY: a moved bound stops at the first value no valued variable forbids */
          u64 crossed = 0ull;
#pragma unroll
          for (int r = 0; r < R; r++) {
            crossed |= __ballot(rlo[r] > rhi[r]);
            val[r] = __ballot(rlo[r] == rhi[r]) & livemask[r];
          }
          if (crossed != 0ull) { failed = 1; break; }
#pragma unroll
          for (int r = 0; r < R; r++) {
            u64 bits = dl[r] | dh[r];
            while (bits != 0ull && !failed) {
              const int wl = __builtin_ctzll(bits);
              const u64 wbit = 1ull << wl;
              bits &= ~wbit;
              const bool do_lo = (dl[r] & wbit) != 0ull, do_hi = (dh[r] & wbit) != 0ull;
              int cl = __builtin_amdgcn_readlane(rlo[r], wl), ch = __builtin_amdgcn_readlane(rhi[r], wl);
              const int nkw = -__builtin_amdgcn_readlane(kb[r], wl);
              revisions += __builtin_amdgcn_readlane(deg[r], wl);
              const E *row = s_tab + (size_t)(wl + r * CS_WAVE) * slots * W + lane;
              if (SL != 0) {
                int rx[SL ? SL : 1][R]; /* the value of w that THIS lane's variable forbids when it is a value */
#pragma unroll
                for (int k = 0; k < (SL ? SL : 1); k++)
#pragma unroll
                  for (int r2 = 0; r2 < R; r2++) rx[k][r2] = rlo[r2] + (int)row[k * W + r2 * CS_WAVE] + nkw;
                if (do_lo) {
                  for (;;) {
                    u64 hit = 0ull;
#pragma unroll
                    for (int r2 = 0; r2 < R; r2++) {
                      bool h = false;
#pragma unroll
                      for (int k = 0; k < (SL ? SL : 1); k++) h |= rx[k][r2] == cl;
                      hit |= __ballot(h) & val[r2];
                    }
                    if (hit == 0ull) break;
                    cl++;
                    if (cl > ch) break;
                  }
                }
                if (do_hi && cl <= ch) {
                  for (;;) {
                    u64 hit = 0ull;
#pragma unroll
                    for (int r2 = 0; r2 < R; r2++) {
                      bool h = false;
#pragma unroll
                      for (int k = 0; k < (SL ? SL : 1); k++) h |= rx[k][r2] == ch;
                      hit |= __ballot(h) & val[r2];
                    }
                    if (hit == 0ull) break;
                    ch--;
                    if (cl > ch) break;
                  }
                }
              } else {
                /* run-time slot count: one candidate at a time, the row re-read per candidate */
                for (int side = 0; side < 2; side++) {
                  if (side == 0 ? !do_lo : (!do_hi || cl > ch)) continue;
                  for (;;) {
                    const int cand = side == 0 ? cl : ch;
                    u64 hit = 0ull;
                    for (int k = 0; k < slots; k++) {
#pragma unroll
                      for (int r2 = 0; r2 < R; r2++)
                        hit |= __ballot(rlo[r2] + (int)row[k * W + r2 * CS_WAVE] + nkw == cand) & val[r2];
                    }
                    if (hit == 0ull) break;
                    if (side == 0) cl++; else ch--;
                    if (cl > ch) break;
                  }
                }
              }
              if (lane == wl) { rlo[r] = cl; rhi[r] = ch; }
              if (cl > ch) failed = 1;
              else if (cl == ch) val[r] |= wbit; /* counts for the verifications that follow */
            }
            dl[r] = 0ull; dh[r] = 0ull;
          }
          if (failed) break;
          /* (3) variables that became values (and are supported) push next */
          any_push = 0ull;
#pragma unroll
          for (int r = 0; r < R; r++) {
            push[r] = val[r] & ~pushed[r];
            any_push |= push[r];
          }
          if (any_push == 0ull) break;
          rounds++;
        }

        int open_vars = 0, shaved = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
          open_vars += __popcll(__ballot(rlo[r] != rhi[r]));
          shaved += (rlo[r] - lo0[r]) + (hi0[r] - rhi[r]);
        }
        const int props = cs_wave_sum(shaved);
        const size_t orow = (size_t)(base + j) * n;
        if (FULL) {
#pragma unroll
          for (int r = 0; r < R; r++) states_out[orow + lane + r * CS_WAVE] = cs_interval(rlo[r] + b0[r], rhi[r] + b0[r]);
        } else if (!failed) {
#pragma unroll
          for (int r = 0; r < R; r++)
            if (live[r]) states_out[orow + lane + r * CS_WAVE] = cs_interval(rlo[r] + b0[r], rhi[r] + b0[r]);
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

#endif
