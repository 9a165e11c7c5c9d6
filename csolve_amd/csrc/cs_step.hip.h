/* cs_step.hip.h -- one level of the search tree in ONE launch: branch + propagate_clauses of every child + store.
 *
 * What the reference does per open node (solve(), reference src/csolve.c:398-476): pick a variable, try every
 * value of its interval (step_val 331-338), bind + propagate_clauses per value (check_assignment 247-261), count
 * CALLS / CUTS (65-73, 255-258), descend into the consistent ones.  The engine of round 1/2 did this as eight
 * launches per frontier (branch, scan, emit, fixpoint, classify x 3, scatter) with every intermediate -- node
 * records, child states, results, survivor lists -- written to and read back from HBM; the fixpoint was half of the
 * device time (profiles/r02_h_kernel_stats_search_q16.csv).  Here a frontier is one launch of persistent waves:
 *
 *   parents   rows of the pool (`struct val_t` per variable, nothing else), handed out by tickets: one atomic per
 *             `chunk` parents, ticket 0 = the newest rows, so that whatever is left undrawn is a prefix of the pool;
 *   branch    the open variable with the smallest interval, ties lowest index (cs_branch_seg's rule);
 *   holes     a value a valued neighbour of the branching variable forbids is counted as a node and a cut without a
 *             fixpoint (the child "x = that value" fails at its first revision);
 *   children  their fixpoints are computed from the parent's registers / LDS copy, never from HBM;
 *   survivors consistent children with open variables go to the wave's PRIVATE region of the staging buffer -- no
 *             allocation atomics, no lists; complete ones are solutions (on a pure != network a complete consistent
 *             node satisfies every clause: each clause between two valued variables was revised when the second became
 *             a value, propagate_eq_false_lr, propagate.c:106-120);
 *   counters  per wave, written once at exit; `cs_collect` (one more launch) appends the regions to the pool and adds
 *             the counters up.
 * A wave stops drawing parents when its region could overflow; the parents nobody drew stay in the pool.
 *
 * cs_step_packed<G, NW, S3>: models of at most 32 variables (G = 4 segments of 16 lanes, or 2 of 32), the forbidden-set
 * fixpoint of kernel 5 (cs_kernels.hip.h).  A parent's bounds and sets go into a slot of LDS when it is loaded; its
 * children (descriptors {slot, variable, value} in a queue of the wave's own in LDS) start from there.  Branching works
 * on G parents at a time, the fixpoints on G children at a time, whichever parents they belong to: the lanes stay
 * full although parents have different numbers of children.
 *
 * ENGINE ROWS.  The pool of this path holds 8 bytes per variable like `struct val_t`, but in the kernel's own terms,
 * so that a parent needs no preparation (rebuilding a parent's sets from its valued variables, one push each, was a
 * quarter of the first version's time):
 *   NW = 1 (every root domain within 32 values): {set word, rl | rh << 8}  -- bounds relative to the root lower bound;
 *   NW = 2 (within 64 values):                   {set word 0, set word 1}  -- every value outside the interval marked,
 *                                                                             the interval is [first, last unmarked].
 * The set of a fixpoint is exactly what its valued variables forbid, and a child's final registers hold it.
 * cs_step_import turns interval rows (states put from outside: the root, stolen states) into engine rows,
 * cs_step_export does the reverse (states taken away); both in place.
 */
#ifndef CS_STEP_HIP_H
#define CS_STEP_HIP_H

#include "cs_kernels.hip.h"
#include "cs_shave.hip.h"

#define CS_STEP_QN 256      /* child descriptors a wave can hold (power of two) */
#define CS_STEP_SHARDS 16   /* ticket counters of cs_step_shave (one parent per ticket) */
#define CS_STEP_STATS 8     /* words per wave in wstat: nodes, cuts, props, revisions, solutions, parents, 0, 0 */

struct cs_step_io {
  const uint2 *pool;         /* parents (engine rows): rows first_row .. first_row + parents - 1; ticket order is from the top down */
  long long first_row;
  int parents;
  int chunk;                 /* parents per ticket (a multiple of G) */
  int maxw;                  /* widest root interval: children of one parent at most */
  uint2 *stage;              /* survivors (engine rows): wave w owns rows w * K .. w * K + K - 1 */
  int K;
  unsigned *fill;            /* [waves] rows wave w has written */
  unsigned long long *wstat; /* [waves][CS_STEP_STATS] */
  unsigned *ticket;          /* zero at launch; left at the number of tickets drawn (cs_step_shave: CS_STEP_SHARDS counters,
                              * one per 64-byte line) */
  int32_t *solutions;        /* [max_solutions][n] */
  unsigned long long *stored;
  long long max_solutions;
  int store_open;            /* 0: the store is full, nobody asks for a slot */
};

/* segment-wide broadcast of lane `src` (segment-relative) through the LDS crossbar */
template <int S>
__device__ __forceinline__ unsigned cs_seg_bcast(unsigned x, int src, int lane) {
  return (unsigned)__builtin_amdgcn_ds_bpermute(((lane & ~(S - 1)) + src) << 2, (int)x);
}

/* what the packed kernels share: the table in LDS and the push of kernel 5 */
template <int G, int NW, bool S3>
struct cs_packed {
  static constexpr int S = CS_WAVE / G;
  static constexpr int W = CS_WAVE;
  static constexpr unsigned KEY_VALUE = (1u << 26) - 1u;
  const unsigned short *s_tab;
  int n, slots, v, row_stride;
  unsigned key_base;
  /* one push round: every pending lane's variable ORs the value it forbids into the sets of its own segment */
  __device__ __forceinline__ void push_pending(bool &pending, int rl, unsigned *fb) const {
    while (__ballot(pending) != 0ull) {
      const unsigned key = cs_segment_min<S>(pending ? key_base + (unsigned)rl : 0xffffffffu);
      const int ul = (int)(key >> 26);       /* 63: nothing pending in this segment */
      const int cd = (int)(key & KEY_VALUE); /* then 2^26 - 1: selects no bit below */
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
  }
  /* engine row element <-> bounds and sets (a lane without a variable: the value 0, everything else forbidden) */
  __device__ __forceinline__ void decode(uint2 e, bool live, int &rl, int &rh, unsigned *fb) const {
    if (NW == 1) {
      fb[0] = live ? e.x : 0xfffffffeu;
      fb[1] = 0xffffffffu;
      rl = live ? (int)(e.y & 0xffu) : 0;
      rh = live ? (int)((e.y >> 8) & 0xffu) : 0;
    } else {
      fb[0] = live ? e.x : 0xfffffffeu;
      fb[1] = live ? e.y : 0xffffffffu;
      cs_set_bounds_all<2>(fb, &rl, &rh);
      rl = rl > 63 ? 0 : rl; /* a row with nothing allowed is not valid input: stay defined */
      rh = rh < 0 ? 0 : rh;
    }
  }
  __device__ __forceinline__ uint2 encode(int rl, int rh, unsigned *fb) const {
    if (NW == 1) return make_uint2(fb[0], (unsigned)rl | ((unsigned)rh << 8));
    cs_set_restrict<2>(fb, rl, rh);
    return make_uint2(fb[0], fb[1]);
  }
};

template <int G, int NW, bool S3>
__global__ __launch_bounds__(1024, 8) void cs_step_packed(int n, const unsigned short *__restrict__ tab_g, int slots,
                                                          const int *__restrict__ root_lo, const int *__restrict__ sym_off,
                                                          int bias, size_t tab_bytes, cs_step_io io) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  typedef unsigned long long u64;
  constexpr int S = CS_WAVE / G;
  constexpr int W = CS_WAVE;          /* columns of the table */
  constexpr int LOG_S = G == 4 ? 4 : 5;
  constexpr int WORDS = 1 + NW;       /* per lane and parent slot: bounds, set word(s) */
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves_per_block = blockDim.x >> 6;
  const int wave_global = (int)blockIdx.x * waves_per_block + wave_in_block;
  {
    const int vecs = (int)(tab_bytes / 16);
    const uint4 *src = (const uint4 *)tab_g;
    uint4 *dst = (uint4 *)cs_lds;
    for (int i = threadIdx.x; i < vecs; i += blockDim.x) dst[i] = src[i];
  }
  /* the wave's own LDS: four groups of G parent slots (256 lanes' worth per word) and the child queue */
  unsigned *s_wave = (unsigned *)(cs_lds + ((tab_bytes + 15) & ~(size_t)15)) + (size_t)wave_in_block * (WORDS * 256 + CS_STEP_QN);
  unsigned *s_prl = s_wave;            /* [4][64] rl | rh << 8 */
  unsigned *s_pf = s_wave + 256;       /* [NW][4][64] set words */
  unsigned *s_q = s_wave + WORDS * 256; /* [CS_STEP_QN] slot | var << 8 | value << 16 */
  __syncthreads();

  const int g = lane >> LOG_S, v = lane & (S - 1);
  const bool live = v < n;
  const int vcl = live ? v : n - 1;
  const int b0 = live ? root_lo[vcl] : 0;
  const int deg = live ? sym_off[vcl + 1] - sym_off[vcl] : 0;
  cs_packed<G, NW, S3> P;
  P.s_tab = (const unsigned short *)cs_lds;
  P.n = n; P.slots = slots; P.v = v; P.row_stride = slots * W * 2;
  P.key_base = ((unsigned)v << 26) + (unsigned)bias;

  /* scalars of the wave */
  int c_cur = 0, c_end = 0;   /* the chunk of parents being worked on (ticket order) */
  int exhausted = 0;          /* no more tickets for this wave */
  int tk_pending = 0;         /* a ticket has been asked for; it arrives in lane 0 of tk_v */
  unsigned tk_v = 0u;
  int qhead = 0, qlen = 0;    /* the child queue */
  int grp_tail = 0;           /* next group of parent slots to fill (0 .. 3) */
  int fill = 0;               /* rows written to the wave's region */
  int store_open = io.store_open;
  int acc_fail = 0, acc_sol = 0, acc_parents = 0;
  int acc_nodes = 0, acc_skip = 0, acc_props = 0, acc_revs = 0; /* per lane, summed at the end */
  const size_t region = (size_t)wave_global * (size_t)io.K;
  uint2 pre = make_uint2(0u, 0u); /* the rows of the next G parents of the chunk, loaded one step ahead */

  /* rows of the G parents from ticket position `first` on (segment g: first + g, clamped to the chunk) */
  auto load_group = [&](int first, int end) -> uint2 {
    const int pidx = first + g < end ? first + g : end - 1;
    const long long prow = io.first_row + (long long)io.parents - 1 - (long long)pidx;
    return io.pool[(size_t)prow * n + vcl];
  };
  /* ask for a ticket unless the region could overflow with what is queued, what is left of the chunk and a new chunk
   * (a macro, not a lambda: captured by reference the flags end up in scratch memory) */
#define CS_STEP_ASK_TICKET()                                                                                        \
  do {                                                                                                              \
    if (!exhausted && !tk_pending) {                                                                                \
      if (fill + qlen + (c_end - c_cur + io.chunk) * io.maxw <= io.K) {                                             \
        tk_v = 0u;                                                                                                  \
        if (lane == 0) tk_v = __hip_atomic_fetch_add(io.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    \
        tk_pending = 1;                                                                                             \
      } else {                                                                                                      \
        exhausted = 1; /* this wave draws no more: the undrawn parents stay in the pool */                          \
      }                                                                                                             \
    }                                                                                                               \
  } while (0)
  CS_STEP_ASK_TICKET();

  for (;;) {
    /* the wave's bookkeeping is uniform by construction; say so (left to itself the compiler carries it in vector
     * registers, compares it under exec masks and spills some of it to scratch) */
    c_cur = __builtin_amdgcn_readfirstlane(c_cur);
    c_end = __builtin_amdgcn_readfirstlane(c_end);
    exhausted = __builtin_amdgcn_readfirstlane(exhausted);
    tk_pending = __builtin_amdgcn_readfirstlane(tk_pending);
    qhead = __builtin_amdgcn_readfirstlane(qhead);
    qlen = __builtin_amdgcn_readfirstlane(qlen);
    grp_tail = __builtin_amdgcn_readfirstlane(grp_tail);
    fill = __builtin_amdgcn_readfirstlane(fill);
    store_open = __builtin_amdgcn_readfirstlane(store_open);
    acc_fail = __builtin_amdgcn_readfirstlane(acc_fail);
    acc_sol = __builtin_amdgcn_readfirstlane(acc_sol);
    acc_parents = __builtin_amdgcn_readfirstlane(acc_parents);
    /* ---- groups of parent slots in use: those between the oldest queued child's and the last one filled ---- */
    int grp_live = 0;
    if (qlen > 0) {
      const unsigned d0 = s_q[qhead];
      const int grp_head = (int)((__builtin_amdgcn_readfirstlane((int)d0) & 0xff) >> (G == 4 ? 2 : 1));
      grp_live = (grp_tail - grp_head) & 3;
      if (grp_live == 0) grp_live = 4;
    }
    /* ---- A: G more parents, while there is room for their slots and their children ---- */
    if (grp_live < 4 && qlen + G * io.maxw <= CS_STEP_QN) {
      if (c_cur == c_end && tk_pending) { /* the chunk the ticket names */
        tk_pending = 0;
        const long long first = (long long)__builtin_amdgcn_readfirstlane((int)tk_v) * io.chunk;
        if (first < (long long)io.parents) {
          c_cur = (int)first;
          c_end = first + io.chunk < (long long)io.parents ? (int)first + io.chunk : io.parents;
          pre = load_group(c_cur, c_end);
        } else {
          exhausted = 1;
        }
      }
      if (c_cur < c_end) {
        const bool pvalid = c_cur + g < c_end;
        const int had = c_end - c_cur < G ? c_end - c_cur : G;
        const uint2 e = pre;
        c_cur += had;
        acc_parents += had;
        if (c_cur < c_end) pre = load_group(c_cur, c_end); /* the next group's rows: in flight during this one's work */
        else CS_STEP_ASK_TICKET();                         /* the next chunk's ticket likewise */
        int rl, rh;
        unsigned fb[2];
        P.decode(e, live, rl, rh, fb);
        /* the branching variable: smallest open interval, ties lowest index (cs_branch_seg) */
        const bool open = live && pvalid && rl != rh;
        const unsigned kmin = cs_segment_min<S>(open ? ((unsigned)(rh - rl) << 8) | (unsigned)v : 0xffffffffu);
        const int bv = (int)(kmin & 0xffu);
        const bool any_open = kmin != 0xffffffffu;
        const unsigned prl = (unsigned)rl | ((unsigned)rh << 8);
        const unsigned prl_b = cs_seg_bcast<S>(prl, bv & (S - 1), lane);
        const unsigned f0_b = cs_seg_bcast<S>(fb[0], bv & (S - 1), lane);
        const unsigned f1_b = NW == 2 ? cs_seg_bcast<S>(fb[1], bv & (S - 1), lane) : 0u;
        const int rlb = (int)(prl_b & 0xffu), rhb = (int)((prl_b >> 8) & 0xffu);
        /* values of [rlb, rhb] no valued neighbour forbids */
        unsigned a0, a1 = 0u;
        if (NW == 1) {
          a0 = ~f0_b & (~0u << rlb) & (~0u >> (31 - rhb));
        } else {
          const unsigned lo_m0 = rlb < 32 ? ~0u << rlb : 0u, lo_m1 = rlb < 32 ? ~0u : ~0u << (rlb - 32);
          const unsigned hi_m0 = rhb < 32 ? ~0u >> (31 - rhb) : ~0u, hi_m1 = rhb < 32 ? 0u : ~0u >> (63 - rhb);
          a0 = ~f0_b & lo_m0 & hi_m0;
          a1 = ~f1_b & lo_m1 & hi_m1;
        }
        if (!any_open) { a0 = 0u; a1 = 0u; }
        const int width = any_open ? rhb - rlb + 1 : 0;
        const int cnt = __popc(a0) + (NW == 2 ? __popc(a1) : 0);
        if (v == 0) { acc_nodes += width; acc_skip += width - cnt; }
        /* the parents' bounds and sets into their slots */
        const int slot_lane = grp_tail * 64 + lane;
        s_prl[slot_lane] = prl;
        s_pf[slot_lane] = fb[0];
        if (NW == 2) s_pf[256 + slot_lane] = fb[1];
        /* the children into the queue: segment g's after those of the segments before it */
        int off = 0, total = 0;
#pragma unroll
        for (int gg = 0; gg < G; gg++) {
          const int c = __builtin_amdgcn_readlane(cnt, gg * S);
          off += g > gg ? c : 0;
          total += c;
        }
        const unsigned desc_base = (unsigned)(grp_tail * G + g) | ((unsigned)bv << 8);
        const int qbase = qhead + qlen + off;
#pragma unroll
        for (int p = 0; p < (32 * NW) / S; p++) {
          const int bit = v + p * S;
          const unsigned word = (NW == 2 && bit >= 32) ? a1 : a0;
          const int b5 = bit & 31;
          if ((word >> b5) & 1u) {
            int rank = __popc(word & ((1u << b5) - 1u));
            if (NW == 2 && bit >= 32) rank += __popc(a0);
            s_q[(qbase + rank) & (CS_STEP_QN - 1)] = desc_base | ((unsigned)bit << 16);
          }
        }
        if (total > 0) {
          qlen += total;
          grp_tail = (grp_tail + 1) & 3;
        }
        continue; /* more parents if there is room, before the fixpoints */
      }
    }
    if (qlen == 0) break; /* no parent left for this wave and nothing queued */

    /* ---- B: the fixpoints of G children ---- */
    const int take = qlen < G ? qlen : G;
    const bool valid = g < take;
    const unsigned d = s_q[(qhead + (valid ? g : 0)) & (CS_STEP_QN - 1)];
    qhead = (qhead + take) & (CS_STEP_QN - 1);
    qlen -= take;
    const int slot = (int)(d & 0xffu), nvar = (int)((d >> 8) & 0xffu), nval = (int)(d >> 16);
    const int src = slot * S + v; /* slot = group * G + segment: the lanes of that parent */
    const unsigned prl = s_prl[src];
    unsigned fb[2];
    fb[0] = s_pf[src];
    fb[1] = NW == 2 ? s_pf[256 + src] : 0xffffffffu;
    int rl = (int)(prl & 0xffu), rh = (int)((prl >> 8) & 0xffu);
    const bool mine = v == nvar;
    if (mine) { rl = nval; rh = nval; }
    const int rl0 = rl, rh0 = rh; /* the assignment itself is no propagation (kernel 4's reference point) */
    bool pending = mine && valid;
    /* the segments that hold a child (those past the queue's end idle); 32-bit halves: no 64-bit shift by a variable */
    const unsigned valid_lo = G == 2 ? 0xffffffffu : (take >= 2 ? 0xffffffffu : 0x0000ffffu);
    const unsigned valid_hi = G == 2 ? (take >= 2 ? 0xffffffffu : 0u) : (take >= 4 ? 0xffffffffu : (take == 3 ? 0x0000ffffu : 0u));
    const u64 validm = ((u64)valid_hi << 32) | (u64)valid_lo;
    u64 failedm = 0ull, pushedm = __ballot(pending);
    for (;;) {
      P.push_pending(pending, rl, fb);
      int first, last;
      bool bad;
      if (NW == 1) {
        const unsigned a = ~fb[0] & (~0u << rl) & (~0u >> (31 - rh));
        bad = a == 0u;
        first = __builtin_ctz(a | 0x80000000u);
        last = 31 - __builtin_clz(a | 1u);
      } else {
        cs_set_bounds<2>(fb, rl, rh, &first, &last);
        bad = last < 0;
      }
      const bool newly = !bad && first == last && rl != rh;
      rl = bad ? rl : first;
      rh = bad ? rh : last;
      const u64 badm = __ballot(bad);
      if (badm != 0ull) failedm |= cs_segments_any<G>(badm);
      const u64 newm = __ballot(newly) & ~failedm & validm;
      if (newm == 0ull) break;
      pending = __builtin_amdgcn_inverse_ballot_w64(newm);
      pushedm |= newm;
    }
    failedm &= validm;
    const u64 openm = cs_segments_any<G>(__ballot(live && rl != rh)) & validm;
    const u64 survm = openm & ~failedm, complm = validm & ~failedm & ~openm;
    acc_fail += __popcll(failedm) >> LOG_S;
    /* propagations of consistent children only: there the count is the reference's PROPS (every bound move is one
     * narrowing whatever the order); what an inconsistent child did before it failed depends on the revision order */
    acc_props += __builtin_amdgcn_inverse_ballot_w64(validm & ~failedm) ? (rl - rl0) + (rh0 - rh) : 0;
    acc_revs += __builtin_amdgcn_inverse_ballot_w64(pushedm) ? deg : 0;
    if (survm != 0ull) {
      const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(survm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)survm, 0u)) >> LOG_S;
      const uint2 e = P.encode(rl, rh, fb);
      if (__builtin_amdgcn_inverse_ballot_w64(survm) && live) io.stage[(region + (size_t)(fill + rank)) * n + v] = e;
      fill += __popcll(survm) >> LOG_S;
    }
    if (complm != 0ull) {
      const int ns = __popcll(complm) >> LOG_S;
      acc_sol += ns;
      if (store_open) { /* which solutions are kept may vary from run to run; their count does not */
        unsigned long long s0 = 0ull;
        if (lane == 0) s0 = atomicAdd(io.stored, (unsigned long long)ns);
        const long long slot0 = (long long)(((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(s0 >> 32)) << 32) |
                                            (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)s0));
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(complm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)complm, 0u)) >> LOG_S;
        if (__builtin_amdgcn_inverse_ballot_w64(complm) && live && slot0 + rank < io.max_solutions)
          io.solutions[(size_t)(slot0 + rank) * n + v] = b0 + rl;
        if (slot0 + ns >= io.max_solutions) store_open = 0;
      }
    }
  }
  /* what the wave has to report: its fill and its counters */
  const int nodes = cs_wave_sum(acc_nodes), skip = cs_wave_sum(acc_skip), props = cs_wave_sum(acc_props),
            revs = cs_wave_sum(acc_revs);
  if (lane == 0) {
    io.fill[wave_global] = (unsigned)fill;
    unsigned long long *st = io.wstat + (size_t)wave_global * CS_STEP_STATS;
    st[0] = (unsigned long long)nodes;
    st[1] = (unsigned long long)(skip + acc_fail);
    st[2] = (unsigned long long)props;
    st[3] = (unsigned long long)revs;
    st[4] = (unsigned long long)acc_sol;
    st[5] = (unsigned long long)acc_parents;
  }
#undef CS_STEP_ASK_TICKET
}

/* interval rows -> engine rows, in place (rows [first_row, first_row + count) of `rows`): the sets of a fixpoint are
 * what its valued variables forbid; one segment per row, grid-stride */
template <int G, int NW, bool S3>
__global__ __launch_bounds__(256) void cs_step_import(int n, const unsigned short *__restrict__ tab_g, int slots,
                                                      const int *__restrict__ root_lo, int bias, size_t tab_bytes,
                                                      uint2 *__restrict__ rows, long long first_row, long long count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  constexpr int S = CS_WAVE / G;
  constexpr int TOP = 32 * NW - 1;
  {
    const int vecs = (int)(tab_bytes / 16);
    const uint4 *src = (const uint4 *)tab_g;
    uint4 *dst = (uint4 *)cs_lds;
    for (int i = threadIdx.x; i < vecs; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int v = lane & (S - 1);
  const bool live = v < n;
  const int vcl = live ? v : n - 1;
  const int b0 = live ? root_lo[vcl] : 0;
  cs_packed<G, NW, S3> P;
  P.s_tab = (const unsigned short *)cs_lds;
  P.n = n; P.slots = slots; P.v = v; P.row_stride = slots * CS_WAVE * 2;
  P.key_base = ((unsigned)v << 26) + (unsigned)bias;
  const long long segs = (long long)gridDim.x * (blockDim.x / S);
  const long long groups = (count + G - 1) / G; /* whole waves iterate together */
  for (long long q = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); q < groups; q += segs / G) {
    const long long r = q * G + lane / S;
    const bool valid = r < count;
    const size_t at = (size_t)(first_row + (valid ? r : count - 1)) * n + vcl;
    const cs_val d = ((const cs_val *)rows)[at];
    int rl = live ? d.lo - b0 : 0, rh = live ? d.hi - b0 : 0;
    rl = rl < 0 ? 0 : (rl > TOP ? TOP : rl); /* states outside the root domain are not valid input: stay defined */
    rh = rh < 0 ? 0 : (rh > TOP ? TOP : rh);
    unsigned fb[2];
    fb[0] = live ? 0u : 0xfffffffeu;
    fb[1] = live ? 0u : 0xffffffffu;
    bool pending = live && valid && rl == rh;
    P.push_pending(pending, rl, fb);
    const uint2 e = P.encode(rl, rh, fb);
    if (valid && live) rows[at] = e;
  }
}

/* engine rows -> interval rows, in place: one thread per element */
template <int NW>
__global__ __launch_bounds__(256) void cs_step_export(int n, const int *__restrict__ root_lo, uint2 *__restrict__ rows,
                                                      long long elements) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elements) return;
  const uint2 e = rows[i];
  const int b0 = root_lo[(int)(i % n)];
  int rl, rh;
  if (NW == 1) {
    rl = (int)(e.y & 0xffu);
    rh = (int)((e.y >> 8) & 0xffu);
  } else {
    unsigned fb[2] = { e.x, e.y };
    cs_set_bounds_all<2>(fb, &rl, &rh);
  }
  ((cs_val *)rows)[i] = cs_interval(b0 + rl, b0 + rh);
}

/* minimum over the wave, in a scalar */
__device__ __forceinline__ unsigned cs_wave_min_u32(unsigned x) {
  unsigned y;
  y = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)x, 0x111, 0xf, 0xf, false); x = y < x ? y : x; /* row_shr:1 */
  y = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)x, 0x112, 0xf, 0xf, false); x = y < x ? y : x;
  y = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)x, 0x114, 0xf, 0xf, false); x = y < x ? y : x;
  y = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)x, 0x118, 0xf, 0xf, false); x = y < x ? y : x; /* lane 15 of a row = row minimum */
  y = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)x, 0x142, 0xa, 0xf, false); x = y < x ? y : x; /* row_bcast:15 into rows 1 and 3 */
  y = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)x, 0x143, 0xc, 0xf, false); x = y < x ? y : x; /* row_bcast:31 into rows 2 and 3 */
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

/* cs_step_shave<E, R, SL, FULL>: the same level step for models of 33 to 256 variables, on kernel 7's terms
 * (cs_shave.hip.h): one wave per parent, lane l holds variables l, l + 64, ...; the pool holds plain interval rows.
 * A parent is loaded once; its branching variable is the wave minimum of (width - 1, index); the values of its
 * interval that a valued neighbour forbids are found with ONE row of the table -- lane u holds tab[x][k][u] and knows
 * which value of x its own value forbids, the lanes OR those bits into eight words of LDS -- and every other value
 * is a child whose fixpoint (PUSH / VERIFY from the assignment) runs from the parent's registers.
 * One parent per ticket, CS_STEP_SHARDS ticket counters (parent p belongs to shard p mod CS_STEP_SHARDS; a wave
 * starts at its own shard and goes on to the others when it is dry).  The caller sizes the frontier for the worst
 * case (every child survives), so every parent is drawn: a wave that has no room for another parent's children
 * stops, the others take over. */
template <typename E, int R, int SL, bool FULL>
__global__ __launch_bounds__(1024, (R <= 2 ? 8 : 4)) void cs_step_shave(int n, const E *__restrict__ tab_g, int slots, int dmin,
                                                                        const int *__restrict__ root_lo,
                                                                        const int *__restrict__ sym_off, size_t tab_bytes,
                                                                        cs_step_io io) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_lds[];
  typedef unsigned long long u64;
  const int lane = threadIdx.x & (CS_WAVE - 1);
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves_per_block = blockDim.x >> 6;
  const int wave_global = (int)blockIdx.x * waves_per_block + wave_in_block;
  if (SL != 0) slots = SL;
  {
    const int vecs = (int)(tab_bytes / 16);
    const uint4 *src = (const uint4 *)tab_g;
    uint4 *dst = (uint4 *)cs_lds;
    for (int i = threadIdx.x; i < vecs; i += blockDim.x) dst[i] = src[i];
  }
  unsigned *s_mask = (unsigned *)(cs_lds + ((tab_bytes + 15) & ~(size_t)15)) + (size_t)wave_in_block * 8;
  __syncthreads();
  const cs_val *pool = (const cs_val *)io.pool;
  cs_val *stage = (cs_val *)io.stage;

  cs_shave_core<E, R, SL, false> C;
  C.s_tab = (const E *)cs_lds; C.slots = slots; C.lane = lane; C.s_trace = nullptr; C.tcount = 0u;
  int b0[R], vcl[R];
  bool live[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int v = lane + r * CS_WAVE;
    live[r] = FULL || v < n;
    vcl[r] = live[r] ? v : n - 1; /* lanes past the end re-read the last variable and ignore it */
    b0[r] = live[r] ? root_lo[vcl[r]] : 0;
    C.b0[r] = b0[r];
    C.kb[r] = b0[r] - dmin;
    C.deg[r] = live[r] ? sym_off[vcl[r] + 1] - sym_off[vcl[r]] : 0;
    C.livemask[r] = FULL ? ~0ull : __ballot(live[r]);
  }
  constexpr int W = CS_WAVE * R;

  int fill = 0, store_open = io.store_open;
  int acc_nodes = 0, acc_cuts = 0, acc_sol = 0, acc_parents = 0, acc_revs = 0; /* scalars */
  int acc_props = 0;                                                           /* per lane */
  const size_t region = (size_t)wave_global * (size_t)io.K;
  int shard = wave_global % CS_STEP_SHARDS, tried = 0;

  /* OUT = the next parent (ticket order) of the wave's current shard, or -1 when every shard is dry (a macro: a lambda
   * that changes captured scalars leaves them in scratch memory) */
#define CS_STEP_DRAW(OUT)                                                                                            \
  do {                                                                                                               \
    (OUT) = -1;                                                                                                      \
    while (tried < CS_STEP_SHARDS) {                                                                                 \
      const int count_s_ = (io.parents - shard + CS_STEP_SHARDS - 1) / CS_STEP_SHARDS;                               \
      unsigned t_ = 0u;                                                                                              \
      if (count_s_ > 0) {                                                                                            \
        if (lane == 0) t_ = __hip_atomic_fetch_add(io.ticket + shard * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
        t_ = (unsigned)__builtin_amdgcn_readfirstlane((int)t_);                                                      \
      }                                                                                                              \
      if (count_s_ > 0 && t_ < (unsigned)count_s_) { (OUT) = (int)t_ * CS_STEP_SHARDS + shard; break; }              \
      shard = (shard + 1) % CS_STEP_SHARDS;                                                                          \
      tried++;                                                                                                       \
    }                                                                                                                \
  } while (0)
  auto load_parent = [&](int p, cs_val *pd) {
    const size_t prow = (size_t)(io.first_row + (long long)io.parents - 1 - (long long)p) * n;
#pragma unroll
    for (int r = 0; r < R; r++) pd[r] = pool[prow + vcl[r]];
  };

  int p_cur = -1;
  if (io.maxw <= io.K) CS_STEP_DRAW(p_cur);
  cs_val pd[R];
  if (p_cur >= 0) load_parent(p_cur, pd);
  while (p_cur >= 0) {
    p_cur = __builtin_amdgcn_readfirstlane(p_cur);
    fill = __builtin_amdgcn_readfirstlane(fill);
    shard = __builtin_amdgcn_readfirstlane(shard);
    tried = __builtin_amdgcn_readfirstlane(tried);
    store_open = __builtin_amdgcn_readfirstlane(store_open);
    /* the parent in registers (relative to the root lower bounds), the next one's ticket and row on their way */
    int plo[R], phi[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      plo[r] = live[r] ? pd[r].lo - b0[r] : 0;
      phi[r] = live[r] ? pd[r].hi - b0[r] : 0;
    }
    acc_parents++;
    /* room for one more parent's children behind this one's (worst case both)? */
    int p_next = -1;
    if (fill + 2 * io.maxw <= io.K) CS_STEP_DRAW(p_next);
    if (p_next >= 0) load_parent(p_next, pd);
    u64 pval[R];
    unsigned key = 0xffffffffu;
#pragma unroll
    for (int r = R - 1; r >= 0; r--) {
      pval[r] = __ballot(plo[r] == phi[r]) & C.livemask[r];
      const bool open = live[r] && plo[r] != phi[r];
      const unsigned k = ((unsigned)(phi[r] - plo[r]) << 8) | (unsigned)(lane + r * CS_WAVE);
      key = open && k < key ? k : key;
    }
    const unsigned kmin = cs_wave_min_u32(key);
    if (kmin != 0xffffffffu) { /* (a pool row always has an open variable) */
      const int bv = (int)(kmin & 0xffu), br = bv >> 6, bl = bv & 63;
      int blo = 0, bhi = 0, kbw = 0;
#pragma unroll
      for (int r = 0; r < R; r++)
        if (r == br) {
          blo = __builtin_amdgcn_readlane(plo[r], bl);
          bhi = __builtin_amdgcn_readlane(phi[r], bl);
          kbw = __builtin_amdgcn_readlane(C.kb[r], bl);
        }
      const int width = bhi - blo + 1;
      /* holes: lane u's own value forbids the value e + plo[u] - kb[x] of x (cs_shave.hip.h, VERIFY) */
      if (lane < 8) s_mask[lane] = 0u;
      {
        const E *row = (const E *)cs_lds + (size_t)bv * slots * W + lane;
        for (int k = 0; k < slots; k++) {
#pragma unroll
          for (int r2 = 0; r2 < R; r2++) {
            const int c = (int)row[k * W + r2 * CS_WAVE] + plo[r2] - kbw - blo;
            if (__builtin_amdgcn_inverse_ballot_w64(pval[r2]) && c >= 0 && c < width) atomicOr(&s_mask[c >> 5], 1u << (c & 31));
          }
        }
      }
      acc_nodes += width;
      /* the children: every value of [blo, bhi] whose bit is clear */
      for (int wd = 0; wd * 32 < width; wd++) {
        unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)s_mask[wd]);
        const int bits = width - wd * 32 < 32 ? width - wd * 32 : 32;
        unsigned todo = ~m & (bits == 32 ? 0xffffffffu : (1u << bits) - 1u);
        acc_cuts += bits - __builtin_popcount(todo);
        while (todo != 0u) {
          const int j = __builtin_ctz(todo);
          todo &= todo - 1u;
          const int value = blo + wd * 32 + j;
          int rlo[R], rhi[R], lo0[R], hi0[R];
          u64 pushed[R], push[R], dl[R], dh[R], val[R];
#pragma unroll
          for (int r = 0; r < R; r++) {
            rlo[r] = plo[r]; rhi[r] = phi[r];
            pushed[r] = pval[r] | ~C.livemask[r]; /* lanes without a variable look like values: they never push */
            push[r] = 0ull; dl[r] = 0ull; dh[r] = 0ull;
            val[r] = pval[r];
            if (r == br) {
              if (lane == bl) { rlo[r] = value; rhi[r] = value; }
              push[r] = 1ull << bl; /* a scalar shift */
              val[r] |= push[r];
            }
            lo0[r] = rlo[r]; hi0[r] = rhi[r];
          }
          int rounds = 0, revisions = 0;
          const int fail_var = C.fixpoint(rlo, rhi, pushed, push, dl, dh, val, rounds, revisions);
          if (rounds != 0) __builtin_amdgcn_s_setprio(0);
          acc_revs += revisions;
          int open_vars = 0;
#pragma unroll
          for (int r = 0; r < R; r++) {
            open_vars += __popcll(__ballot(rlo[r] != rhi[r]));
            if (fail_var < 0) acc_props += (rlo[r] - lo0[r]) + (hi0[r] - rhi[r]); /* consistent children only: the reference's PROPS */
          }
          if (fail_var >= 0) {
            acc_cuts++;
          } else if (open_vars > 0) {
            const size_t orow = (region + (size_t)fill) * n;
#pragma unroll
            for (int r = 0; r < R; r++)
              if (live[r]) stage[orow + lane + r * CS_WAVE] = cs_interval(rlo[r] + b0[r], rhi[r] + b0[r]);
            fill++;
          } else {
            acc_sol++;
            if (store_open) {
              unsigned long long s0 = 0ull;
              if (lane == 0) s0 = atomicAdd(io.stored, 1ull);
              const long long slot0 = (long long)(((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(s0 >> 32)) << 32) |
                                                  (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)s0));
              if (slot0 < io.max_solutions) {
#pragma unroll
                for (int r = 0; r < R; r++)
                  if (live[r]) io.solutions[(size_t)slot0 * n + lane + r * CS_WAVE] = rlo[r] + b0[r];
              }
              if (slot0 + 1 >= io.max_solutions) store_open = 0;
            }
          }
        }
      }
    }
    p_cur = p_next;
  }
  const int props = cs_wave_sum(acc_props);
  if (lane == 0) {
    io.fill[wave_global] = (unsigned)fill;
    unsigned long long *st = io.wstat + (size_t)wave_global * CS_STEP_STATS;
    st[0] = (unsigned long long)acc_nodes;
    st[1] = (unsigned long long)acc_cuts;
    st[2] = (unsigned long long)props;
    st[3] = (unsigned long long)acc_revs;
    st[4] = (unsigned long long)acc_sol;
    st[5] = (unsigned long long)acc_parents;
  }
#undef CS_STEP_DRAW
}

/* Appends the waves' regions to the pool (after the parents nobody drew) and adds the waves' counters up.
 * Workgroup w: rows before region w's = sum of fill[0 .. w), then a flat copy of its fill[w] * n elements.
 * out[0] = parents consumed, out[1] = survivors, out[2 ..] = nodes, cuts, props, revisions, solutions, out[7] = rows in
 * the solution store (workgroup 0). */
__global__ __launch_bounds__(256) void cs_collect(const unsigned *__restrict__ fill, int waves, const uint2 *__restrict__ stage,
                                                  int K, int n, uint2 *__restrict__ pool, long long first_row, int parents,
                                                  int chunk, const unsigned *__restrict__ ticket,
                                                  const unsigned long long *__restrict__ wstat,
                                                  unsigned long long *__restrict__ out,
                                                  const unsigned long long *__restrict__ stored) {
  __shared__ unsigned long long s_red[256];
  const int w = blockIdx.x, t = threadIdx.x;
  /* chunk == 0: the launch drew every parent (cs_step_shave, sized for the worst case) */
  const unsigned long long drawn = chunk == 0 ? (unsigned long long)parents : (unsigned long long)*ticket * (unsigned long long)chunk;
  const long long consumed = drawn < (unsigned long long)parents ? (long long)drawn : (long long)parents;
  unsigned long long before = 0ull;
  for (int i = t; i < w; i += 256) before += fill[i];
  s_red[t] = before;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (t < d) s_red[t] += s_red[t + d];
    __syncthreads();
  }
  before = s_red[0];
  __syncthreads();
  const size_t count = (size_t)fill[w] * n;
  const uint2 *src = stage + (size_t)w * K * n;
  uint2 *dst = pool + (size_t)(first_row + parents - consumed + (long long)before) * n;
  size_t e = t;
  for (; e + 768 < count; e += 1024) { /* four loads in flight per thread */
    const uint2 a = src[e], b = src[e + 256], c = src[e + 512], d = src[e + 768];
    dst[e] = a; dst[e + 256] = b; dst[e + 512] = c; dst[e + 768] = d;
  }
  for (; e < count; e += 256) dst[e] = src[e];
  if (w == 0) {
    for (int k = 0; k < 6; k++) {
      unsigned long long x = 0ull;
      for (int i = t; i < waves; i += 256) x += k < 5 ? wstat[(size_t)i * CS_STEP_STATS + k] : (unsigned long long)fill[i];
      s_red[t] = x;
      __syncthreads();
      for (int d = 128; d > 0; d >>= 1) {
        if (t < d) s_red[t] += s_red[t + d];
        __syncthreads();
      }
      if (t == 0) out[k < 5 ? 2 + k : 1] = s_red[0];
      __syncthreads();
    }
    if (t == 0) {
      out[0] = (unsigned long long)consumed;
      out[7] = *stored;
    }
  }
}

#endif
