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
 * to the variable's root lower bound; masks of valued / pushed / dirty variables are 64-bit scalars.  LDS holds
 * the table only.
 *
 * Work distribution.  Nodes cost very different amounts (a node is 1 to 60 table-row operations), and with
 * static shares the waves of a launch finished anywhere between 30 and 119 us of a 119 us launch (mean
 * residency 0.5, tools/shave_timeline.py).  So the waves are persistent (one resident grid) and take their
 * work in chunks of two nodes from ticket counters in device memory: chunk c belongs to shard c mod S
 * (S <= 64 counters, each on its own 64-byte line, a workgroup uses shard blockIdx mod S; a single word
 * sustains about 88 atomics per microsecond, MI355X_MICROARCH.md), a wave's first chunk is static, the next
 * ones are ticket numbers.  Every wave stops at its first ticket past the end, so a launch draws exactly
 * "chunks of the shard" tickets per counter, and the wave that draws the last one resets the counter: no
 * host-side clearing, safe under hipGraph replay.  The pipeline of a wave, one chunk per step: ticket for
 * chunk k + 2 (atomic in flight), node records of chunk k + 1 (load in flight), parent rows of chunk k + 1
 * (loads in flight), fixpoints of chunk k.  Without a ticket buffer (tickets == NULL) or with fewer chunks
 * than waves the shares are static (chunk index strides by the number of waves of the shard).
 */
#ifndef CS_SHAVE_HIP_H
#define CS_SHAVE_HIP_H

#include "cs_kernels.hip.h"

#ifdef CS_SHAVE_TIMELINE
/* diagnostic build only (tools/shave_timeline.py): start, end (100 MHz real-time counter) and node count per wave */
__device__ unsigned long long cs_shave_tl[3 * 65536];
#endif

/* clear bit `i` of a wave-uniform 64-bit mask: one scalar instruction (the compiler's `m &= m - 1` is three) */
__device__ __forceinline__ unsigned long long cs_bitset0(unsigned long long m, int i) {
  asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(i));
  return m;
}

/* x + (x == f) and x - (x == f): a compare into VCC and an add / subtract with carry-in */
__device__ __forceinline__ int cs_inc_if_eq(int x, int f) {
  int r;
  asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %1, vcc" : "=v"(r) : "v"(x), "v"(f) : "vcc");
  return r;
}
__device__ __forceinline__ int cs_dec_if_eq(int x, int f) {
  int r;
  asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_subb_co_u32 %0, vcc, %1, 0, vcc" : "=v"(r) : "v"(x), "v"(f) : "vcc");
  return r;
}

/* v with lane L replaced by the wave-uniform x */
template <int L>
__device__ __forceinline__ int cs_writelane(int v, int x) {
  x = __builtin_amdgcn_readfirstlane(x); /* free when the compiler already knows x to be uniform */
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(x), "n"(L));
  return v;
}

/* the lane is a constant after unrolling: the branches fold */
__device__ __forceinline__ int cs_writelane_at(int lane, int v, int x) {
  switch (lane) {
  case 0: return cs_writelane<0>(v, x);
  case 1: return cs_writelane<1>(v, x);
  case 2: return cs_writelane<2>(v, x);
  default: return cs_writelane<3>(v, x);
  }
}

#define CS_SHAVE_TRACE_LDS 2048u /* trace records kept in LDS (32 KB) by the tracing variant */
#define CS_SHAVE_CHUNK 2        /* nodes per chunk = parent rows in flight per wave */
#define CS_SHAVE_SHARDS 64      /* ticket counters per launch at most */
#define CS_SHAVE_TICKET_STRIDE 16 /* unsigned words between two counters: one 64-byte line each */

/* The fixpoint of ONE node on a wave's registers: PUSH / VERIFY as described at the top of this file.  Shared by
 * kernel 7 (cs_propagate_ne_shave: a batch of nodes) and by the search's level kernel (cs_step_shave, cs_step.hip.h:
 * the children of a parent are run from the parent's registers).  Every function is inlined into its caller. */
template <typename E, int R, int SL, bool TRACE>
struct cs_shave_core {
  typedef unsigned long long u64;
  static constexpr int W = CS_WAVE * R; /* columns of the table */
  const E *s_tab;
  int slots, lane;
  int kb[R], deg[R], b0[R];
  u64 livemask[R];
  int4 *s_trace;   /* TRACE: the records of the node, in LDS */
  unsigned tcount; /* TRACE: records made so far (one wave, one node: a scalar) */

  /* the lanes of `mask` (register r2) record their new bound, moved because of variable `cause` */
  __device__ __forceinline__ void trace_lanes(u64 mask, int r2, int kind, int value, int cause) {
    if (mask == 0ull) return;
    const unsigned at = tcount + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    /* into LDS: a store to the caller's (host-mapped) buffer would be waited for before the registers are reused,
     * a PCIe round trip per row operation (measured: 3.4 us per operation, 1 ms for a cascade of 300) */
    if (__builtin_amdgcn_inverse_ballot_w64(mask) && at < CS_SHAVE_TRACE_LDS) /* (no 64-bit shift by a lane number: tools/k4_fault_repro.md) */
      s_trace[at] = make_int4(lane + r2 * CS_WAVE, kind, value + b0[r2], cause);
    tcount += (unsigned)__builtin_popcountll(mask);
  }

  /* PUSH(u): the variable `ul` of register r is the value cd + dmin; every lane moves a bound that equals the
   * value u forbids for it by one and is dirty then */
  __device__ __forceinline__ void push_var(int r, int ul, int cd, int (&rlo)[R], int (&rhi)[R], u64 (&dl)[R], u64 (&dh)[R],
                                           int &revisions) {
    revisions += __builtin_amdgcn_readlane(deg[r], ul);
    const E *row = s_tab + (size_t)(ul + r * CS_WAVE) * slots * W + lane;
    /* The slots are applied one after the other to the bounds as they stand (a bound moved by slot k may be moved
     * again by slot k + 1: both values are forbidden by u, each move is one narrowing); `bound += (f == bound)` is
     * a compare into VCC and an add with carry-in, and which bounds moved is ONE compare per bound at the end --
     * the scalar unit merges nothing. */
    int plo[R], phi[R];
#pragma unroll
    for (int r2 = 0; r2 < R; r2++) { plo[r2] = rlo[r2]; phi[r2] = rhi[r2]; }
    if (SL != 0) {
#pragma unroll
      for (int k = 0; k < (SL ? SL : 1); k++) {
#pragma unroll
        for (int r2 = 0; r2 < R; r2++) {
          const int f = cd - (int)row[k * W + r2 * CS_WAVE]; /* the value of w that u forbids (sentinel: < 0) */
          rlo[r2] = cs_inc_if_eq(rlo[r2], f);
          rhi[r2] = cs_dec_if_eq(rhi[r2], f);
        }
      }
    } else {
      for (int k = 0; k < slots; k++) {
#pragma unroll
        for (int r2 = 0; r2 < R; r2++) {
          const int f = cd - (int)row[k * W + r2 * CS_WAVE];
          rlo[r2] = cs_inc_if_eq(rlo[r2], f);
          rhi[r2] = cs_dec_if_eq(rhi[r2], f);
        }
      }
    }
#pragma unroll
    for (int r2 = 0; r2 < R; r2++) {
      const u64 ml = __ballot(rlo[r2] != plo[r2]), mh = __ballot(rhi[r2] != phi[r2]);
      dl[r2] |= ml;
      dh[r2] |= mh;
      if (TRACE) {
        trace_lanes(ml, r2, 0, rlo[r2], ul + r * CS_WAVE);
        trace_lanes(mh, r2, 1, rhi[r2], ul + r * CS_WAVE);
      }
    }
  }

  /* the valued variables after a push phase; -1, or a variable whose bounds have crossed */
  __device__ __forceinline__ int settle(const int (&rlo)[R], const int (&rhi)[R], u64 (&val)[R]) const {
    int bad = -1;
#pragma unroll
    for (int r = R - 1; r >= 0; r--) {
      const u64 c = __ballot(rlo[r] > rhi[r]);
      if (c != 0ull) bad = __builtin_ctzll(c) + r * CS_WAVE;
      val[r] = __ballot(rlo[r] == rhi[r]) & livemask[r];
    }
    return bad;
  }

  /* the fixpoint; returns -1, or a variable whose domain has become empty (what the reference's
   * propagate_term_confl would be called for, propagate.c:33-41).  push: the variables that push first; pushed: those
   * that have pushed already; dl / dh: bounds to be verified; val: the variables that are single values now. */
  __device__ __forceinline__ int fixpoint(int (&rlo)[R], int (&rhi)[R], u64 (&pushed)[R], u64 (&push)[R], u64 (&dl)[R],
                                          u64 (&dh)[R], u64 (&val)[R], int &rounds, int &revisions) {
    int fail_v = -1;
    for (;;) {
      /* (1) PUSH: every newly valued variable moves the bounds it sits on.  A valued variable's bounds do
       * not change any more, so "value - dmin" of every pusher of this round is taken from one register. */
      int cdv[R];
#pragma unroll
      for (int r = 0; r < R; r++) cdv[r] = rlo[r] + kb[r];
#pragma unroll
      for (int r = 0; r < R; r++) {
        u64 bits = push[r];
        pushed[r] |= bits;
        while (bits != 0ull) {
          const int ul = __builtin_ctzll(bits);
          bits = cs_bitset0(bits, ul);
          push_var(r, ul, __builtin_amdgcn_readlane(cdv[r], ul), rlo, rhi, dl, dh, revisions);
        }
      }
      fail_v = settle(rlo, rhi, val);
      if (fail_v >= 0) return fail_v;
      /* (2a) Many moved bounds, few valued variables (an assignment on a bound of the root domains moves a bound
       * of every other queen): cheaper than verifying each moved bound against all valued variables is to let
       * every valued variable push again -- each sweep moves the bounds that are still forbidden by one more
       * value -- until the moved bounds are fewer than the valued variables. */
      for (;;) {
        int n_dirty = 0, n_val = 0;
#pragma unroll
        for (int r = 0; r < R; r++) { n_dirty += __popcll(dl[r] | dh[r]); n_val += __popcll(val[r]); }
        if (n_dirty <= n_val) break;
#pragma unroll
        for (int r = 0; r < R; r++) { cdv[r] = rlo[r] + kb[r]; dl[r] = 0ull; dh[r] = 0ull; }
#pragma unroll
        for (int r = 0; r < R; r++) {
          u64 bits = val[r];
          pushed[r] |= bits;
          while (bits != 0ull) {
            const int ul = __builtin_ctzll(bits);
            bits = cs_bitset0(bits, ul);
            push_var(r, ul, __builtin_amdgcn_readlane(cdv[r], ul), rlo, rhi, dl, dh, revisions);
          }
        }
        fail_v = settle(rlo, rhi, val);
        if (fail_v >= 0) return fail_v;
      }
      /* (2) VERIFY: a moved bound stops at the first value no valued variable forbids.  Lane u holds the
       * table entry e of (w, k, u); its own value forbids the value rlo[u] + e - kb[w] of w, i.e. the
       * candidate c is forbidden iff e == c + kb[w] - rlo[u]: one subtraction per register, the compares run
       * on the raw table bytes.  Mostly the candidate is supported and nothing is written. */
#pragma unroll
      for (int side = 0; side < 2; side++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          u64 bits = side == 0 ? dl[r] : dh[r];
          if (side == 0) dl[r] = 0ull; else dh[r] = 0ull;
          while (bits != 0ull) {
            const int wl = __builtin_ctzll(bits);
            bits = cs_bitset0(bits, wl);
            int cand = __builtin_amdgcn_readlane(side == 0 ? rlo[r] : rhi[r], wl);
            const int ckw = cand + __builtin_amdgcn_readlane(kb[r], wl);
            revisions += __builtin_amdgcn_readlane(deg[r], wl);
            const E *row = s_tab + (size_t)(wl + r * CS_WAVE) * slots * W + lane;
            u64 hit = 0ull;
            if (SL != 0) {
              int e[SL ? SL : 1][R];
#pragma unroll
              for (int k = 0; k < (SL ? SL : 1); k++)
#pragma unroll
                for (int r2 = 0; r2 < R; r2++) e[k][r2] = (int)row[k * W + r2 * CS_WAVE];
              int cause = -1; /* TRACE: a valued variable that forbids the candidate */
#pragma unroll
              for (int r2 = 0; r2 < R; r2++) {
                const int t = ckw - rlo[r2];
                u64 h = 0ull;
#pragma unroll
                for (int k = 0; k < (SL ? SL : 1); k++) h |= __ballot(e[k][r2] == t);
                hit |= h & val[r2];
                if (TRACE && cause < 0 && (h & val[r2]) != 0ull) cause = __builtin_ctzll(h & val[r2]) + r2 * CS_WAVE;
              }
              if (hit == 0ull) continue; /* supported: the common case */
              /* the bound moves on, one value (one PROPS) at a time, until it is supported or passes the other */
              const int other = __builtin_amdgcn_readlane(side == 0 ? rhi[r] : rlo[r], wl);
              int step = 0;
              for (;;) {
                step++;
                if (side == 0 ? cand + step > other : cand - step < other) break;
                hit = 0ull;
#pragma unroll
                for (int r2 = 0; r2 < R; r2++) {
                  const int t = ckw + (side == 0 ? step : -step) - rlo[r2];
                  u64 h = 0ull;
#pragma unroll
                  for (int k = 0; k < (SL ? SL : 1); k++) h |= __ballot(e[k][r2] == t);
                  hit |= h & val[r2];
                }
                if (hit == 0ull) break;
              }
              cand += side == 0 ? step : -step;
              if (lane == wl) { if (side == 0) rlo[r] = cand; else rhi[r] = cand; }
              if (TRACE) trace_lanes(1ull << wl, r, side, cand, cause);
              /* a bound that passed the other one is the failure: noticed after the loops (no exit from in here) */
              if (side == 0 ? cand > other : cand < other) fail_v = wl + r * CS_WAVE;
              if (cand == other) val[r] |= 1ull << wl; /* counts for the verifications that follow */
            } else {
              /* run-time slot count: the row is re-read per candidate */
              const int other = __builtin_amdgcn_readlane(side == 0 ? rhi[r] : rlo[r], wl);
              int step = 0;
              int cause = -1;
              for (;;) {
                hit = 0ull;
                for (int k = 0; k < slots; k++) {
#pragma unroll
                  for (int r2 = 0; r2 < R; r2++) {
                    const u64 h = __ballot((int)row[k * W + r2 * CS_WAVE] == ckw + (side == 0 ? step : -step) - rlo[r2]) & val[r2];
                    hit |= h;
                    if (TRACE && cause < 0 && h != 0ull) cause = __builtin_ctzll(h) + r2 * CS_WAVE;
                  }
                }
                if (hit == 0ull) break;
                step++;
                if (side == 0 ? cand + step > other : cand - step < other) break;
              }
              if (step != 0) {
                cand += side == 0 ? step : -step;
                if (lane == wl) { if (side == 0) rlo[r] = cand; else rhi[r] = cand; }
                if (TRACE) trace_lanes(1ull << wl, r, side, cand, cause);
                if (side == 0 ? cand > other : cand < other) fail_v = wl + r * CS_WAVE;
                if (cand == other) val[r] |= 1ull << wl;
              }
            }
          }
        }
      }
      if (fail_v >= 0) return fail_v;
      /* (3) variables that became values (and are supported) push next */
      u64 any_push = 0ull;
#pragma unroll
      for (int r = 0; r < R; r++) {
        push[r] = val[r] & ~pushed[r];
        any_push |= push[r];
      }
      if (any_push == 0ull) return -1;
      /* a node that goes on cascading is a candidate for the tail of the launch (a few nodes cost 50 times the
       * average): from its second round on its wave is served first by the SIMD's arbiter */
      if (rounds == 0) __builtin_amdgcn_s_setprio(3);
      rounds++;
    }
  }
};

/* SL: slots per pair known at compile time (1 = alldiff-style, 3 = queens), 0 = run-time loop.
 * FULL: n == 64 R, every lane register holds a variable: loads and stores are unconditional. */
/* TRACE (single-node launches only, csgpu_propagate_one_causes): every bound move is recorded as
 * {variable, 0 = lower / 1 = upper bound, new bound, the valued variable that forbade the old one} in `trace`,
 * in the order the wave made them -- the trail the reference keeps through bind() (csolve.h:73-79), with the cause
 * as a variable instead of a clause (on a != network the clause is the pair). */
template <typename E, int R, int SL, bool FULL, bool TRACE = false>
__global__ __launch_bounds__(1024, (R <= 2 ? 8 : 4)) void cs_propagate_ne_shave(
    int n, const E *__restrict__ tab_g, int slots, int dmin, const int *__restrict__ root_lo,
    const int *__restrict__ sym_off, const cs_val *__restrict__ states_in, const cs_node_in *__restrict__ nodes,
    cs_val *__restrict__ states_out, cs_node_out *__restrict__ results, long long batch,
    const unsigned long long *__restrict__ batch_dev, int csz /* nodes per chunk: 1 .. D */, unsigned *tickets,
    int4 *__restrict__ trace, unsigned *__restrict__ trace_n, unsigned trace_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  typedef unsigned long long u64;
  constexpr int W = CS_WAVE * R; /* columns of the table */
  constexpr int D = R == 1 && !TRACE ? 2 * CS_SHAVE_CHUNK : CS_SHAVE_CHUNK; /* 64-variable models: four nodes per ticket (with two registers per lane the rows in flight spill) */
  if (batch_dev != nullptr && (long long)*batch_dev < batch) batch = (long long)*batch_dev;
  /* the ABI's batches stay below 2^31 nodes: node numbers are 32-bit scalars from here on (the scalar unit, which
   * bounds this kernel -- 119 scalar against 105 vector instructions per node on queens-64,
   * profiles/r02_f_sq_counters_shave_queens64.txt -- multiplies 64-bit numbers in six instructions) */
  const int nbatch = __builtin_amdgcn_readfirstlane((int)batch);
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves_per_block = blockDim.x >> 6;
  const int chunks = (nbatch + csz - 1) / csz;
  /* shards: chunk c belongs to shard c mod nsh, local index c / nsh; workgroup b works on shard b mod nsh */
  const int nsh = (int)gridDim.x < CS_SHAVE_SHARDS ? (int)gridDim.x : CS_SHAVE_SHARDS;
  const int shard = (int)(blockIdx.x % nsh);
  const int count_x = chunks > shard ? (chunks - 1 - shard) / nsh + 1 : 0; /* chunks of this shard */
  const int waves_x = (((int)gridDim.x - 1 - shard) / nsh + 1) * waves_per_block;
  const int wave_local = (int)(blockIdx.x / nsh) * waves_per_block + wave_in_block;
  const bool dynamic = tickets != nullptr && count_x > waves_x;
  if ((int)(blockIdx.x / nsh) * waves_per_block >= count_x) return; /* no chunk for any wave of this workgroup */
  if (SL != 0) slots = SL;
  const E *s_tab = (const E *)cs_lds;
  /* TRACE: the records of the node are collected behind the table */
  int4 *s_trace = (int4 *)(cs_lds + ((((size_t)n * slots * W * sizeof(E)) + 15) & ~(size_t)15));
  {
    const int vecs = (int)(((size_t)n * slots * W * sizeof(E)) / 16);
    const uint4 *src = (const uint4 *)tab_g;
    uint4 *dst = (uint4 *)cs_lds;
    for (int i = threadIdx.x; i < vecs; i += blockDim.x) dst[i] = src[i];
  }
#ifdef CS_SHAVE_TIMELINE
  const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long tl_nodes = 0;
#endif
  __syncthreads();
  if (wave_local >= count_x) return;

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
  cs_shave_core<E, R, SL, TRACE> C;
  C.s_tab = s_tab; C.slots = slots; C.lane = lane; C.s_trace = s_trace; C.tcount = 0u;
#pragma unroll
  for (int r = 0; r < R; r++) { C.kb[r] = kb[r]; C.deg[r] = deg[r]; C.b0[r] = b0[r]; C.livemask[r] = livemask[r]; }
  unsigned *my_ticket = tickets + (size_t)shard * CS_SHAVE_TICKET_STRIDE;

  /* the record of chunk i of this shard (lanes 0 .. csz-1); lanes past the end of the batch hold a no-op */
  auto load_rec = [&](int i) {
    cs_node_in rec;
    rec.var = -1; rec.lo = 0; rec.hi = 0; rec.parent = 0;
    const int node = (i * nsh + shard) * csz + lane;
    if (lane < csz && node < nbatch) rec = nodes[node];
    return rec;
  };
  /* ticket -> local chunk index (a scalar); the wave that draws the shard's last ticket resets the counter */
  auto next_index = [&](int i_prev, unsigned tk_v) -> int {
    if (!dynamic) return i_prev + waves_x;
    const int t = __builtin_amdgcn_readfirstlane((int)tk_v);
    if (t == count_x - 1 && lane == 0) __hip_atomic_store(my_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return waves_x + t;
  };
  auto draw = [&]() -> unsigned { /* the ticket arrives in lane 0 */
    unsigned tk_v = 0;
    if (lane == 0) tk_v = __hip_atomic_fetch_add(my_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return tk_v;
  };

  /* prologue of the pipeline: chunk 0 is the wave's own, the ticket of chunk 1 is drawn at once */
  int i_cur = wave_local;
  cs_node_in rec_cur = load_rec(i_cur);
  unsigned tk = dynamic ? draw() : 0u;
  cs_val pd[D][R];
#pragma unroll
  for (int d = 0; d < D; d++) {
    const size_t prow = (size_t)__builtin_amdgcn_readlane(rec_cur.parent, d) * n;
#pragma unroll
    for (int r = 0; r < R; r++) pd[d][r] = states_in[prow + vcl[r]];
  }
  int i_next = next_index(i_cur, tk);
  int have_next = i_next < count_x;
  cs_node_in rec_next = load_rec(have_next ? i_next : i_cur);

  for (;;) {
    /* the indices are wave-uniform by construction; say so (the compiler otherwise carries them in vector registers) */
    i_cur = __builtin_amdgcn_readfirstlane(i_cur);
    i_next = __builtin_amdgcn_readfirstlane(i_next);
    have_next = i_next < count_x; /* from the scalar: a flag carried through a vector register costs four instructions */
    if (have_next && dynamic) tk = draw(); /* for the chunk after the next one */
    cs_val pn[D][R]; /* parent rows of the next chunk (its records arrived during the previous step) */
#pragma unroll
    for (int d = 0; d < D; d++)
#pragma unroll
      for (int r = 0; r < R; r++) pn[d][r] = pd[d][r];
    if (have_next) {
#pragma unroll
      for (int d = 0; d < D; d++) {
        const size_t prow = (size_t)__builtin_amdgcn_readlane(rec_next.parent, d) * n;
#pragma unroll
        for (int r = 0; r < R; r++) pn[d][r] = states_in[prow + vcl[r]];
      }
    }
    const int base = (i_cur * nsh + shard) * csz;
    const int cnt = nbatch - base < csz ? nbatch - base : csz;
#ifdef CS_SHAVE_TIMELINE
    tl_nodes += cnt;
#endif
    cs_node_out my_result;
    my_result.status = 0; my_result.props = 0; my_result.revisions = 0; my_result.rounds = 0;
#pragma unroll
    for (int dd = 0; dd < D; dd++) {
      const int j = dd;
      if (j >= cnt) continue;
      const int nvar = __builtin_amdgcn_readlane(rec_cur.var, j);
      const int nlo = __builtin_amdgcn_readlane(rec_cur.lo, j), nhi = __builtin_amdgcn_readlane(rec_cur.hi, j);
      int rlo[R], rhi[R]; /* bounds relative to the root lower bound; a lane without a variable is the value 0 */
#pragma unroll
      for (int r = 0; r < R; r++) {
        rlo[r] = live[r] ? pd[dd][r].lo - b0[r] : 0;
        rhi[r] = live[r] ? pd[dd][r].hi - b0[r] : 0;
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
#pragma unroll
      for (int r = 0; r < R; r++) {
        val[r] = __ballot(rlo[r] == rhi[r]) & livemask[r];
        push[r] = val[r] & ~pushed[r];
      }

      C.tcount = 0u;
      int rounds = 0, revisions = 0;
      const int fail_var = C.fixpoint(rlo, rhi, pushed, push, dl, dh, val, rounds, revisions);
      const int failed = fail_var >= 0;
      if (TRACE) { /* the records leave LDS in one coalesced burst; a count beyond CS_SHAVE_TRACE_LDS tells the caller that only that many were kept */
        const unsigned keep = C.tcount < CS_SHAVE_TRACE_LDS ? C.tcount : CS_SHAVE_TRACE_LDS;
        for (unsigned i = lane; i < keep && i < trace_cap; i += CS_WAVE) trace[i] = s_trace[i];
        if (lane == 0) *trace_n = C.tcount;
      }
      if (rounds != 0) __builtin_amdgcn_s_setprio(0);

      int open_vars = 0, shaved = 0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        open_vars += __popcll(__ballot(rlo[r] != rhi[r]));
        shaved += (rlo[r] - lo0[r]) + (hi0[r] - rhi[r]);
      }
      const int props = cs_wave_sum(shaved);
      const size_t orow = (size_t)(unsigned)(base + j) * (unsigned)n;
#ifndef CS_SHAVE_STORE_FAILED
/* 1: with FULL an inconsistent node stores its meaningless row too (no branch around the store: round 2's form).  0: it
 * stores nothing -- 12 % of the bench's queens-64 nodes are inconsistent, their rows were 6 % of the launch's traffic:
 * queens-64 66.7 -> 65.1 us per 2^18 nodes, queens-128 60.4 -> 58.8 us (same box, tools/ab_bench.sh) */
#define CS_SHAVE_STORE_FAILED 0
#endif
      if (FULL && (CS_SHAVE_STORE_FAILED || !failed)) {
#pragma unroll
        for (int r = 0; r < R; r++) states_out[orow + lane + r * CS_WAVE] = cs_interval(rlo[r] + b0[r], rhi[r] + b0[r]);
      } else if (!FULL && !failed) {
#pragma unroll
        for (int r = 0; r < R; r++)
          if (live[r]) states_out[orow + lane + r * CS_WAVE] = cs_interval(rlo[r] + b0[r], rhi[r] + b0[r]);
      }
      /* lane j of the four result registers, one v_writelane each (the lane number is a constant of the unrolled copy) */
      my_result.status = cs_writelane_at(dd, my_result.status, failed ? -1 : open_vars);
      my_result.props = cs_writelane_at(dd, my_result.props, props);
      my_result.revisions = cs_writelane_at(dd, my_result.revisions, revisions);
      my_result.rounds = cs_writelane_at(dd, my_result.rounds, failed ? fail_var : rounds); /* an inconsistent node reports the failing variable here */
    }
    if (lane < cnt) results[base + lane] = my_result;

    if (!have_next) break;
    /* step the pipeline: the next chunk becomes the current one, the ticket drawn above names the one after */
    const int i_nn = next_index(i_next, tk);
    const int have_nn = i_nn < count_x;
    i_cur = i_next;
    rec_cur = rec_next;
#pragma unroll
    for (int d = 0; d < D; d++)
#pragma unroll
      for (int r = 0; r < R; r++) pd[d][r] = pn[d][r];
    rec_next = load_rec(have_nn ? i_nn : i_next);
    i_next = i_nn;
    have_next = have_nn;
  }
#ifdef CS_SHAVE_TIMELINE
  if (lane == 0) {
    const size_t w = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % 65536;
    cs_shave_tl[3 * w] = tl_start;
    cs_shave_tl[3 * w + 1] = __builtin_amdgcn_s_memrealtime();
    cs_shave_tl[3 * w + 2] = tl_nodes;
  }
#endif
}

/* ---- the resident single-node server (the drop-in's calls) --------------------------------------------------------
 * One propagate_clauses of the reference's driver is one node.  As a launch it costs what a launch costs: submit,
 * dispatch, completion signal and the host's wait were 14 of the 19 us of a call.  The server is
 * kernel 7's tracing variant as ONE resident wave on a mailbox in coherent host memory: the host writes the node record
 * and the state, then the request number; the wave polls that word, runs the fixpoint, writes state, trail and result
 * back and acknowledges with the same number.  No launch, no stream, no completion signal per call.
 * The wave leaves when the host says so (`stop`) or after `idle_ticks` of the 100 MHz clock without a request (a
 * process that exits without freeing its model must not leave a wave behind); `alive` tells the host, which starts
 * a new one with the next call. */
struct cs_mailbox_head {
  unsigned req_seq; unsigned pad0[15];            /* written by the host, last */
  unsigned ack_seq; unsigned alive; unsigned pad1[14]; /* written by the device */
  unsigned stop; unsigned pad2[15];               /* written by the host */
  cs_node_in node; unsigned want_trace; unsigned pad3[11];
  cs_node_out result; unsigned trace_n; unsigned pad4[11];
};

template <typename E, int R, int SL>
__global__ __launch_bounds__(1024) void cs_shave_server(int n, const E *__restrict__ tab_g, int slots, int dmin,
                                                        const int *__restrict__ root_lo, const int *__restrict__ sym_off,
                                                        cs_mailbox_head *box, unsigned long long *state_in /* [n] host */,
                                                        unsigned long long *state_out /* [n] host */, int4 *trace /* host */,
                                                        unsigned trace_cap, unsigned long long idle_ticks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  typedef unsigned long long u64;
  constexpr int W = CS_WAVE * R;
  const int lane = threadIdx.x & (CS_WAVE - 1);
  if (SL != 0) slots = SL;
  {
    const int vecs = (int)(((size_t)n * slots * W * sizeof(E)) / 16);
    const uint4 *src = (const uint4 *)tab_g;
    uint4 *dst = (uint4 *)cs_lds;
    for (int i = threadIdx.x; i < vecs; i += blockDim.x) dst[i] = src[i];
  }
  int4 *s_trace = (int4 *)(cs_lds + ((((size_t)n * slots * W * sizeof(E)) + 15) & ~(size_t)15));
  __syncthreads();
  if (threadIdx.x >= CS_WAVE) return; /* the other waves only helped with the table */

  cs_shave_core<E, R, SL, true> C;
  C.s_tab = (const E *)cs_lds; C.slots = slots; C.lane = lane; C.s_trace = s_trace; C.tcount = 0u;
  int b0[R];
  bool live[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int v = lane + r * CS_WAVE;
    live[r] = v < n;
    const int vc = live[r] ? v : n - 1;
    b0[r] = live[r] ? root_lo[vc] : 0;
    C.b0[r] = b0[r];
    C.kb[r] = b0[r] - dmin;
    C.deg[r] = live[r] ? sym_off[vc + 1] - sym_off[vc] : 0;
    C.livemask[r] = __ballot(live[r]);
  }
  unsigned last = __hip_atomic_load(&box->ack_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (lane == 0) __hip_atomic_store(&box->alive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  u64 idle_since = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned seq = __hip_atomic_load(&box->req_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (seq == last) {
      if (__hip_atomic_load(&box->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
      if (__builtin_amdgcn_s_memrealtime() - idle_since > idle_ticks) break;
      __builtin_amdgcn_s_sleep(2);
      continue;
    }
    /* the request: node record and state (coherent host memory: system-scope loads) */
    const int nvar = (int)__hip_atomic_load((unsigned *)&box->node.var, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int nlo = (int)__hip_atomic_load((unsigned *)&box->node.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int nhi = (int)__hip_atomic_load((unsigned *)&box->node.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned want_trace = __hip_atomic_load(&box->want_trace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int rlo[R], rhi[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int v = lane + r * CS_WAVE;
      const u64 e = __hip_atomic_load(&state_in[live[r] ? v : n - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      rlo[r] = live[r] ? (int)(unsigned)e - b0[r] : 0;
      rhi[r] = live[r] ? (int)(unsigned)(e >> 32) - b0[r] : 0;
    }
    u64 pushed[R], push[R], dl[R], dh[R], val[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      pushed[r] = nvar < 0 ? ~C.livemask[r] : __ballot(rlo[r] == rhi[r]);
      dl[r] = 0ull; dh[r] = 0ull;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (nvar >= 0 && (nvar >> 6) == r) {
        const u64 bit = 1ull << (nvar & 63); /* a scalar shift */
        if (lane == (nvar & 63)) { rlo[r] = nlo - b0[r]; rhi[r] = nhi - b0[r]; }
        pushed[r] &= ~bit;
        if (nlo != nhi) { dl[r] |= bit; dh[r] |= bit; }
      }
    }
    int lo0[R], hi0[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      lo0[r] = rlo[r]; hi0[r] = rhi[r];
      val[r] = __ballot(rlo[r] == rhi[r]) & C.livemask[r];
      push[r] = val[r] & ~pushed[r];
    }
    C.tcount = 0u;
    int rounds = 0, revisions = 0;
    const int fail_var = C.fixpoint(rlo, rhi, pushed, push, dl, dh, val, rounds, revisions);
    if (rounds != 0) __builtin_amdgcn_s_setprio(0);
    const int failed = fail_var >= 0;
    int open_vars = 0, shaved = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      open_vars += __popcll(__ballot(rlo[r] != rhi[r]));
      shaved += (rlo[r] - lo0[r]) + (hi0[r] - rhi[r]);
    }
    const int props = cs_wave_sum(shaved);
    /* the answer: state, trail, result -- then the acknowledgement behind a system-scope release */
    if (!failed) {
#pragma unroll
      for (int r = 0; r < R; r++)
        if (live[r]) {
          const u64 e = (u64)(unsigned)(rlo[r] + b0[r]) | ((u64)(unsigned)(rhi[r] + b0[r]) << 32);
          __hip_atomic_store(&state_out[lane + r * CS_WAVE], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (want_trace != 0u) {
      const unsigned keep = C.tcount < CS_SHAVE_TRACE_LDS ? C.tcount : CS_SHAVE_TRACE_LDS;
      for (unsigned i = lane; i < keep && i < trace_cap; i += CS_WAVE) trace[i] = s_trace[i];
    }
    if (lane == 0) {
      box->result.status = failed ? -1 : open_vars;
      box->result.props = props;
      box->result.revisions = revisions;
      box->result.rounds = failed ? fail_var : rounds;
      box->trace_n = C.tcount;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(&box->ack_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    last = seq;
    idle_since = __builtin_amdgcn_s_memrealtime();
  }
  if (lane == 0) __hip_atomic_store(&box->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#endif
