/* cs_search.hip -- device-resident tree search on top of the batched propagation ABI.
 * Reference behaviour being mirrored: solve() (reference src/csolve.c:398-476) -- branch on a
 * variable, try every value of its interval (step_val 331-338), propagate each (check_assignment
 * 247-261), count CALLS/CUTS (65-73, 255-258), accept complete assignments whose root evaluates to
 * true (update_solution 222-244), keep the incumbent (objective.c:81-126).  The reference walks
 * this tree depth-first one node at a time; here whole frontiers are expanded per launch. */
#include <hip/hip_runtime.h>
#include <chrono>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/csolve_gpu.h"
#include "cs_arith.h"
#include "cs_frontend.h"
#include "cs_internal.h"

#define SB 256 /* threads per block of the bookkeeping kernels */

/* device counters.  [C_SURVIVORS, C_PER_ITERATION) are zeroed at the start of every iteration (C_SKIPPED: children
 * cut without a launch, see cs_holes); C_SOLUTIONS and C_STORED run over the whole search; C_BEST holds the
 * incumbent (an int in the low half) */
enum { C_SURVIVORS = 0, C_COMPLETE, C_CUTS, C_PROPS, C_REVS, C_TOTAL_CHILDREN, C_SKIPPED, C_PER_ITERATION,
       C_SOLUTIONS = C_PER_ITERATION, C_STORED, C_BEST, C_COUNT };

/* Values that the parent's own forbidden set already rules out (models whose states carry one set word per
 * variable): the child "variable = such a value" violates a != clause with a valued neighbour, so its fixpoint
 * can only fail.  Such children are counted as nodes and cuts but never launched: the tree, CALLS and CUTS are
 * those of enumerating every value of the interval, the batches are a fraction of it. */
struct cs_holes {
  const unsigned long long *pool_forb; /* nullptr: every value of the interval becomes a launched child */
  const int *root_lo;
  /* the branching rule (strategy_var_cmp, reference src/strategy.c:79-121): which open variable comes first --
   * order 0 none, 1 smallest domain (the default), 2 largest domain, 3 smallest value, 4 largest value -- and, with
   * `prio` != nullptr (-f true: prefer failing), among equals the one with the highest failure count; then the
   * lowest index.  What the reference keeps in a heap is the minimum of this key over the open variables. */
  int order;
  const int *prio;
};

/* the whole key: state part, then failure count (higher first), then index (cs_arith.h: the same function is
 * exported as csgpu_branch_key and pinned by the reference's VarCmp vectors) */
__device__ __forceinline__ unsigned long long cs_branch_key(const cs_holes &H, cs_val d, int v) {
  return cs_branch_key_of(H.order, H.prio != nullptr, d, H.prio != nullptr ? (long long)H.prio[v] : 0ll, v);
}

/* what the branching step decides for a parent and the emitting step needs (32 bytes per parent) */
struct cs_choice {
  int var;             /* -1: no open variable */
  int lo, hi;          /* the branching variable's interval */
  int count;           /* children that are launched */
  unsigned a_lo, a_hi; /* holes != 0: bit j <=> value lo + j is a child */
  int holes;           /* the parent's set was consulted: only the values it allows become children */
  int skipped;         /* values of the interval cut without a launch */
};

/* state of the device-driven iterations, in device memory between the kernels of a burst */
enum { B_TOP = 0, B_BUDGET, B_LIMIT, B_LIMIT_MAX, B_ITER_BASE, B_ITERS, B_NODES, B_CUTS, B_PROPS, B_REVS, B_PEAK, B_ERROR,
       B_SCATTER_BASE, B_IMPROVED, B_D_PARENTS, B_D_FIRST, B_D_ITER /* the iteration cs_burst_branch decided on */,
       B_BACKLOG_DIV /* parents = pool / this, within [B_LIMIT, B_LIMIT_MAX] */, B_COUNT };
#define BURST_ITERATIONS 16
/* a MIN / MAX iteration's bookkeeping is spread over this many workgroups (one workgroup is bound by what ONE CU
 * reads, ~25 GB/s: 1,024 parent rows took it 24 us, 10,000 results 18 us) */
#define BURST_PPW 64        /* parents per workgroup: sixteen lanes each, one pass of 1,024 threads */
#define BURST_WGS_MAX 256   /* one wave adds up the workgroups' child counts, four each: at most 16,384 parents per iteration */
#define BURST_PARENTS_MAX (BURST_PPW * BURST_WGS_MAX)
#define BURST_CLASS_WGS 128 /* at most 1,024: a workgroup adds up the others' counts one per thread */

struct csgpu_search {
  const csgpu_model *m;
  int n, objective, obj_var;
  int64_t cap, max_children, max_parents, max_width, parents_limit;
  int64_t parents_max; /* device-driven iterations take up to this many parents when the pool has a backlog */
  double avg_children; /* children per parent of the recent iterations (ALL sizes its batches by it) */
  cs_val *pool;
  int64_t top, peak;
  /* forbidden sets travelling with the states (pure binary-NE models, csgpu_propagate_batch_fb) */
  int fw;
  unsigned long long *pool_forb, *d_child_forb;
  csgpu_node *d_rebuild_nodes;
  cs_choice *d_choice; /* per parent: what cs_branch decided */
  int *d_child_off /* per workgroup of cs_branch */, *d_block_sum, *d_block_skip;
  csgpu_node *d_nodes;
  cs_val *d_child_states, *d_complete_states;
  csgpu_result *d_results;
  int *d_dest /* survivor k = child d_dest[k] */, *d_complete_list, *d_truth;
  int *d_block_surv, *d_block_comp, *d_block_cuts, *d_block_props, *d_block_revs, *d_surv_off, *d_comp_off;
  unsigned long long *d_counters; /* [C_COUNT] */
  int *d_best;                    /* = (int *)&d_counters[C_BEST] */
  int64_t pending_complete;       /* complete children of the last iteration whose accept results are unread */
  int32_t *d_solutions;           /* [max_solutions][n] */
  int32_t *d_best_solution;       /* [n] a solution attaining the incumbent (MIN/MAX) */
  int have_best_solution;
  int32_t best_solution_value;    /* the objective value d_best_solution attains (it may have been overtaken by an engine
                                   * that shares the incumbent word: then this engine has no best row to show) */
  /* a shared incumbent (csgpu_search_share_incumbent): the engine whose word this one uses, and how many engines
   * use this one's.  A lender is kept alive (csgpu_search_free deferred) until its last borrower is gone. */
  csgpu_search *lender;
  int borrowers, free_pending;
  double put_seconds;     /* host time spent in csgpu_search_put / put_host (copy + rebuilding the forbidden sets) */
  int64_t put_states;
  int device; /* the device the engine was created on; made current in the calling thread by every entry point
               * (a fresh host thread starts on device 0) */
  int64_t max_solutions;
  csgpu_search_stats st;
  /* restarts (ANY): the states put from outside are kept to restart from */
  cs_val *seed;
  int64_t seed_count, seed_cap, restart_base, since_restart;
  uint64_t luby_threshold, luby_counter;
  /* device-driven iterations (ANY / MIN / MAX): BURST_ITERATIONS iterations per host round trip, as one hipGraph */
  unsigned long long *d_burst, *h_burst; /* [B_COUNT] device / pinned host; h_burst[B_COUNT ...] = copy of the counters,
                                          * then the incumbent */
  hipStream_t burst_stream;
  hipGraphExec_t burst_exec;
  int64_t burst_limit; /* parents per iteration the graph was built for (0: none) */
  cs_holes holes;      /* values a parent's own set forbids are cut without a launch (one set word per variable) */
  /* fused levels (cs_step.hip.h): ALL on models with a step kernel -- the pool holds interval rows only, a frontier is
   * one launch (branch + fixpoints + store) plus cs_collect */
  int counted; /* this engine is in its model's engine count */
  int fused;
  int64_t stage_rows;           /* rows of d_child_states, the staging buffer of the survivors */
  uint32_t *d_fill, *d_ticket;
  uint64_t *d_wstat, *d_step_out, *h_step_out; /* h: pinned */
  double surv_per_parent;       /* recent survivors per parent (sizes the next frontier) */
  uint64_t stored_seen;         /* rows in the solution store after the last iteration */
  /* the reference's strategy options (main.c:51-130): -o order, -f prefer failing, restart on a better solution */
  int order, prefer_failing, restart_on_improvement, fail_var_known;
  int *d_prio; /* [n] failure counts (prefer failing) */
  int burst_off;       /* CSGPU_SEARCH_BURST=0: every iteration driven from the host */
  int eval_always;     /* CSGPU_SEARCH_EVAL=1: complete children of pure != networks are evaluated all the same (tests) */
  int graph_off;       /* CSGPU_SEARCH_GRAPH=0: the launches of a burst enqueued one by one */
  int burst_no_eval;   /* device-driven iterations launch no root evaluation: see enqueue_burst */
  int burst_split;     /* MIN / MAX: expansion and classification of a device-driven iteration by several workgroups
                        * (CSGPU_SEARCH_BURST_SPLIT=0: by one, as ANY) */
};

extern "C" int csgpu_internal_set_error(int code, const char *msg); /* cs_capi.hip */
static int fail(int code, const char *msg) { return csgpu_internal_set_error(code, msg); }
static int flush_accept_results(csgpu_search *s);
static int burst_applicable(const csgpu_search *s);

#define SPLIT_WIDTH 256 /* wider intervals are halved instead of enumerated (csolve.c:121-150 style) */
#define HIP_OK(expr)                                                           \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) return fail(CSGPU_E_HIP, hipGetErrorString(e_));     \
  } while (0)

/* the values lo .. lo + width - 1 of a variable whose set word is `forb` (bit k = value root_lo + k):
 * bit j of the result <=> value lo + j is not forbidden.  32-bit halves (no variable 64-bit shifts). */
__device__ __forceinline__ void cs_allowed_values(unsigned long long forb, int rel_lo, int width, unsigned *a_lo,
                                                  unsigned *a_hi) {
  const unsigned lo = ~(unsigned)forb, hi = ~(unsigned)(forb >> 32);
  unsigned x_lo, x_hi;
  if (rel_lo >= 32) { x_lo = hi >> (rel_lo - 32); x_hi = 0u; }
  else if (rel_lo == 0) { x_lo = lo; x_hi = hi; }
  else { x_lo = (lo >> rel_lo) | (hi << (32 - rel_lo)); x_hi = hi >> rel_lo; }
  if (width < 32) { x_lo &= (1u << width) - 1u; x_hi = 0u; }
  else if (width == 32) x_hi = 0u;
  else if (width < 64) x_hi &= (1u << (width - 32)) - 1u;
  *a_lo = x_lo;
  *a_hi = x_hi;
}

/* What a lane found among ITS variables (v = sl, sl + S, ...): the smallest key and that variable's interval.  Scan and
 * pick are separate so that a caller can have the rows of several parents in flight before it reduces any of them. */
struct cs_branch_part {
  unsigned long long best;
  cs_val d;
};

template <int S>
__device__ __forceinline__ cs_branch_part cs_branch_scan(const cs_val *__restrict__ row, int n, int sl, const cs_holes &H) {
  cs_branch_part p;
  p.best = ~0ull;
  p.d = cs_value(0);
  for (int v = sl; v < n; v += S) {
    const cs_val d = row[v];
    if (d.lo != d.hi) {
      const unsigned long long key = cs_branch_key(H, d, v);
      if (key < p.best) { p.best = key; p.d = d; }
    }
  }
  return p;
}

template <int S>
__device__ __forceinline__ cs_choice cs_branch_pick(cs_branch_part p, long long row_index, int n, const cs_holes &H) {
  unsigned long long best = p.best;
  for (int o = S / 2; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(best, o);
    best = other < best ? other : best;
  }
  cs_choice c;
  c.var = -1; c.lo = 0; c.hi = 0; c.count = 0; c.a_lo = 0u; c.a_hi = 0u; c.holes = 0; c.skipped = 0;
  if (best == ~0ull) return c;
  const int var = (int)(best & 0xffffu);
  /* the lane that scanned the variable (v = sl + k S, so sl = var mod S) holds its interval: no second look at the row */
  cs_val d;
  d.lo = __shfl(p.d.lo, var & (S - 1), S);
  d.hi = __shfl(p.d.hi, var & (S - 1), S);
  unsigned long long forb = 0ull;
  int root = 0;
  if (H.pool_forb != nullptr) {
    forb = H.pool_forb[(size_t)row_index * n + var];
    root = H.root_lo[var];
  }
  const long long width = (long long)d.hi - (long long)d.lo + 1;
  c.var = var;
  c.lo = d.lo;
  c.hi = d.hi;
  c.count = width > SPLIT_WIDTH ? 2 : (int)width;
  const long long rel_lo = (long long)d.lo - (long long)root;
  if (H.pool_forb != nullptr && width <= 64 && rel_lo >= 0 && rel_lo + width <= 64) {
    cs_allowed_values(forb, (int)rel_lo, (int)width, &c.a_lo, &c.a_hi);
    const int allowed = __popc(c.a_lo) + __popc(c.a_hi);
    c.holes = 1;
    c.skipped = c.count - allowed;
    c.count = allowed;
  }
  return c;
}

/* S lanes (a whole wave, or a half or a quarter of one for small models) per parent: the open variable the branching
 * rule puts first (cs_branch_key: by default the smallest interval, ties lowest index -- the reference's
 * "-o smallest-domain" idea, strategy.c:85-91, as a pure function of the state).  Intervals wider than SPLIT_WIDTH are
 * halved (two children) instead of enumerated.  The same choice in every lane of the segment. */
template <int S>
__device__ __forceinline__ cs_choice cs_branch_seg(const cs_val *__restrict__ row, long long row_index, int n, int sl,
                                                   const cs_holes &H) {
  return cs_branch_pick<S>(cs_branch_scan<S>(row, n, sl, H), row_index, n, H);
}

/* a workgroup takes SB / S consecutive parents; besides var and count per parent it leaves the number of
 * children of its parents in block_sum, so that the scan that follows runs over workgroups, not parents */
template <int S>
__global__ __launch_bounds__(SB) void cs_branch(const cs_val *__restrict__ pool, long long first_row, int parents,
                                                int n, cs_choice *__restrict__ choice,
                                                int *__restrict__ block_sum, cs_holes H,
                                                int *__restrict__ block_skip) {
  constexpr int PPB = SB / S;
  __shared__ int s_cnt[PPB], s_skip[PPB];
  const int seg = threadIdx.x / S, sl = threadIdx.x & (S - 1);
  const int p = blockIdx.x * PPB + seg;
  const int pc = p < parents ? p : parents - 1; /* segments past the end redo the last parent and drop it */
  const cs_choice c = cs_branch_seg<S>(pool + (size_t)(first_row + pc) * n, first_row + pc, n, sl, H);
  if (sl == 0) {
    if (p < parents) choice[p] = c;
    s_cnt[seg] = p < parents ? c.count : 0;
    s_skip[seg] = p < parents ? c.skipped : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0, skip = 0;
    for (int i = 0; i < PPB; i++) { total += s_cnt[i]; skip += s_skip[i]; }
    block_skip[blockIdx.x] = skip; /* summed by cs_scan */
    block_sum[blockIdx.x] = total;
  }
}

/* block-wide exclusive scan of one value per thread (1024 threads): wave scan with shuffles, the 16 wave
 * totals through LDS.  Returns the exclusive prefix; *total = the sum over the block. */
__device__ __forceinline__ long long cs_block_excl_scan(long long x, long long *s_part, long long *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long incl = x;
  for (int d = 1; d < 64; d <<= 1) {
    const long long up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  long long before = 0, all = 0;
  for (int w = 0; w < 16; w++) {
    const long long p = s_part[w];
    before += w < wave ? p : 0;
    all += p;
  }
  __syncthreads();
  *total = all;
  return before + incl - x;
}

/* exclusive scan of count[0..items) by one block; every thread takes SCAN_PER consecutive elements of a
 * tile (vector loads), the per-thread sums go through the block scan; total -> off[items] and
 * counters[total_slot].  count and off are 16-byte aligned (hipMalloc), the tail is handled one by one. */
#define SCAN_PER 16
__global__ __launch_bounds__(1024) void cs_scan(const int *__restrict__ count, int items, int *__restrict__ off,
                                                unsigned long long *__restrict__ counters, int total_slot,
                                                const int *__restrict__ extra /* nullable: summed into extra_slot */,
                                                int extra_slot) {
  __shared__ long long s_part[16];
  long long carry = 0, extra_sum = 0;
  for (int base = 0; base < items; base += 1024 * SCAN_PER) {
    const int first = base + (int)threadIdx.x * SCAN_PER;
    int x[SCAN_PER];
    if (first + SCAN_PER <= items) {
#pragma unroll
      for (int q = 0; q < SCAN_PER / 4; q++) {
        const int4 v = ((const int4 *)(count + first))[q];
        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < SCAN_PER; q++) x[q] = first + q < items ? count[first + q] : 0;
    }
    if (extra != nullptr) {
      if (first + SCAN_PER <= items) {
#pragma unroll
        for (int q = 0; q < SCAN_PER / 4; q++) {
          const int4 v = ((const int4 *)(extra + first))[q];
          extra_sum += (long long)v.x + v.y + v.z + v.w;
        }
      } else {
        for (int q = 0; q < SCAN_PER; q++) extra_sum += first + q < items ? extra[first + q] : 0;
      }
    }
    int sum = 0;
#pragma unroll
    for (int q = 0; q < SCAN_PER; q++) { const int v = x[q]; x[q] = sum; sum += v; } /* exclusive within the thread */
    long long total;
    const long long ex = carry + cs_block_excl_scan((long long)sum, s_part, &total);
    if (first + SCAN_PER <= items) {
#pragma unroll
      for (int q = 0; q < SCAN_PER / 4; q++)
        ((int4 *)(off + first))[q] = make_int4((int)ex + x[4 * q], (int)ex + x[4 * q + 1], (int)ex + x[4 * q + 2], (int)ex + x[4 * q + 3]);
    } else {
#pragma unroll
      for (int q = 0; q < SCAN_PER; q++)
        if (first + q < items) off[first + q] = (int)ex + x[q];
    }
    carry += total;
  }
  long long extra_total = 0;
  if (extra != nullptr) (void)cs_block_excl_scan(extra_sum, s_part, &extra_total);
  if (threadIdx.x == 0) {
    off[items] = (int)carry;
    counters[total_slot] = (unsigned long long)carry;
    if (extra != nullptr) counters[extra_slot] = (unsigned long long)extra_total;
  }
}

/* the per-block class counts of cs_classify_count: exclusive scans of the survivors and of the complete
 * children (one scan: survivors in the low half of a 64-bit word, complete children in the high half), sums
 * of cuts / propagations / revisions -> counters */
__global__ __launch_bounds__(1024) void cs_scan_classes(const int *__restrict__ block_surv, const int *__restrict__ block_comp,
                                                        const int *__restrict__ block_cuts, const int *__restrict__ block_props,
                                                        const int *__restrict__ block_revs, int blocks,
                                                        int *__restrict__ surv_off, int *__restrict__ comp_off,
                                                        unsigned long long *__restrict__ counters) {
  __shared__ long long s_part[16];
  constexpr int PER = 4;
  long long carry = 0, cuts = 0, props = 0, revs = 0;
  for (int base = 0; base < blocks; base += 1024 * PER) {
    const int first = base + (int)threadIdx.x * PER;
    long long x[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int i = first + q;
      const bool in = i < blocks;
      const long long v = in ? (long long)block_surv[i] | ((long long)block_comp[i] << 32) : 0;
      x[q] = sum;
      sum += v;
      cuts += in ? block_cuts[i] : 0;
      props += in ? block_props[i] : 0;
      revs += in ? block_revs[i] : 0;
    }
    long long total;
    const long long ex = carry + cs_block_excl_scan(sum, s_part, &total);
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int i = first + q;
      if (i < blocks) {
        surv_off[i] = (int)((ex + x[q]) & 0xffffffffll);
        comp_off[i] = (int)((ex + x[q]) >> 32);
      }
    }
    carry += total;
  }
  long long t_cuts, t_props, t_revs;
  (void)cs_block_excl_scan(cuts, s_part, &t_cuts);
  (void)cs_block_excl_scan(props, s_part, &t_props);
  (void)cs_block_excl_scan(revs, s_part, &t_revs);
  if (threadIdx.x == 0) {
    surv_off[blocks] = (int)(carry & 0xffffffffll);
    comp_off[blocks] = (int)(carry >> 32);
    counters[C_SURVIVORS] = (unsigned long long)(carry & 0xffffffffll);
    counters[C_COMPLETE] = (unsigned long long)(carry >> 32);
    counters[C_CUTS] = (unsigned long long)t_cuts;
    counters[C_PROPS] = (unsigned long long)t_props;
    counters[C_REVS] = (unsigned long long)t_revs;
  }
}

/* S lanes write the children {var, value, value, parent_row} of one parent at nodes[beg, beg + c.count) */
template <int S>
__device__ __forceinline__ void cs_emit_seg(const cs_choice &c, long long row, int beg, csgpu_node *__restrict__ nodes,
                                            int low_values_last, unsigned scramble, int sl) {
  const int var = c.var, cnt = c.count;
  if (var < 0) return;
  const long long width = (long long)c.hi - (long long)c.lo + 1;
  unsigned h = 0u;
  if (scramble != 0u && cnt > 0) {
    /* ANY: the values are tried from a pseudo-random starting point (the reference randomises its
     * value order too: the seed of step_val, csolve.c:284,331-338).  Deterministic: a function of
     * the variable, the row and the iteration only. */
    h = (scramble ^ (unsigned)var * 2654435761u ^ (unsigned)row * 40503u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    h %= (unsigned)cnt;
  }
  if (c.holes) {
    /* only the values the parent's set allows (cnt of them): value lo + j is the r-th allowed one from below
     * and takes the place the r-th value has in the full enumeration below */
    for (int j = sl; j < (int)width; j += S) {
      const unsigned bit = j < 32 ? (c.a_lo >> j) & 1u : (c.a_hi >> (j - 32)) & 1u;
      if (bit == 0u) continue;
      const int r = j < 32 ? __popc(c.a_lo & ((1u << j) - 1u)) : __popc(c.a_lo) + __popc(c.a_hi & ((1u << (j - 32)) - 1u));
      const int k = scramble != 0u ? (int)(((unsigned)r + (unsigned)cnt - h) % (unsigned)cnt) : (low_values_last ? cnt - 1 - r : r);
      csgpu_node nd;
      nd.var = var;
      nd.lo = c.lo + j;
      nd.hi = c.lo + j;
      nd.parent = (int)row;
      nodes[beg + k] = nd;
    }
    return;
  }
  if (width > SPLIT_WIDTH) { /* two halves, lower half first */
    const int mid = (int)(((long long)c.lo + (long long)c.hi) >> 1);
    if (sl < 2) {
      /* the pool is LIFO and later children land higher: the half written last is explored first */
      const int lower = low_values_last ? sl == 1 : sl == 0;
      csgpu_node nd;
      nd.var = var;
      nd.lo = lower ? c.lo : mid + 1;
      nd.hi = lower ? mid : c.hi;
      nd.parent = (int)row;
      nodes[beg + sl] = nd;
    }
    return;
  }
  for (int k = sl; k < cnt; k += S) {
    csgpu_node nd;
    int value = low_values_last ? c.hi - k : c.lo + k;
    if (scramble != 0u) value = c.lo + (int)(((unsigned)k + h) % (unsigned)cnt);
    nd.var = var;
    nd.lo = value;
    nd.hi = value;
    nd.parent = (int)row;
    nodes[beg + k] = nd;
  }
}

/* same geometry as cs_branch<S>: block_off[b] = children before this workgroup's parents (the scan of
 * cs_branch's block sums), the few parents in front within the workgroup are added up directly */
template <int S>
__global__ __launch_bounds__(SB) void cs_emit(long long first_row, int parents, const cs_choice *__restrict__ choice,
                                              const int *__restrict__ block_off, csgpu_node *__restrict__ nodes,
                                              int low_values_last, unsigned scramble) {
  constexpr int PPB = SB / S;
  __shared__ cs_choice s_choice[PPB];
  const int seg = threadIdx.x / S, sl = threadIdx.x & (S - 1);
  const int p0 = blockIdx.x * PPB, p = p0 + seg;
  if ((int)threadIdx.x < PPB) {
    cs_choice c;
    c.var = -1; c.lo = 0; c.hi = 0; c.count = 0; c.a_lo = 0u; c.a_hi = 0u; c.holes = 0; c.skipped = 0;
    if (p0 + (int)threadIdx.x < parents) c = choice[p0 + threadIdx.x];
    s_choice[threadIdx.x] = c;
  }
  __syncthreads();
  if (p >= parents) return;
  int beg = block_off[blockIdx.x];
  for (int j = 0; j < seg; j++) beg += s_choice[j].count;
  cs_emit_seg<S>(s_choice[seg], first_row + p, beg, nodes, low_values_last, scramble, sl);
}

/* ---- small iterations (at most SMALL_PARENTS parents: always for ANY / MIN / MAX): one workgroup does what
 * cs_branch + cs_scan + cs_emit do, and leaves the number of children on the device, so that the host need not
 * read anything before it launches the fixpoint ---- */
#define SMALL_PARENTS 1024 /* one workgroup of 1,024 threads scans their child counts */
__global__ __launch_bounds__(1024) void cs_expand_small(const cs_val *__restrict__ pool, long long first_row, int parents,
                                                        int n, csgpu_node *__restrict__ nodes,
                                                        unsigned long long *__restrict__ counters, int low_values_last,
                                                        unsigned scramble, cs_holes H) {
  __shared__ cs_choice s_choice[SMALL_PARENTS];
  __shared__ int s_off[SMALL_PARENTS];
  __shared__ long long s_part[16];
  if (threadIdx.x < C_PER_ITERATION) counters[threadIdx.x] = 0ull;
  /* sixteen lanes per parent, 64 parents per pass of the workgroup, four passes' rows in flight at a time (measured
   * no faster than one pass at a time: 1,024 parent rows are 300 KB through ONE CU, ~25 GB/s -- 12 us whatever the order) */
  for (int p0 = (int)threadIdx.x >> 4; p0 < parents; p0 += 256) {
    cs_branch_part part[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int p = p0 + 64 * k < parents ? p0 + 64 * k : parents - 1;
      part[k] = cs_branch_scan<16>(pool + (size_t)(first_row + p) * n, n, (int)threadIdx.x & 15, H);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int p = p0 + 64 * k;
      const cs_choice c = cs_branch_pick<16>(part[k], first_row + (p < parents ? p : parents - 1), n, H);
      if ((threadIdx.x & 15) == 0 && p < parents) s_choice[p] = c;
    }
  }
  __syncthreads();
  long long total, skipped_total;
  const int t = (int)threadIdx.x;
  (void)cs_block_excl_scan(t < parents ? (long long)s_choice[t].skipped : 0, s_part, &skipped_total);
  const long long ex = cs_block_excl_scan(t < parents ? (long long)s_choice[t].count : 0, s_part, &total);
  if (t < parents) s_off[t] = (int)ex;
  if (t == 0) {
    counters[C_TOTAL_CHILDREN] = (unsigned long long)total;
    counters[C_SKIPPED] = (unsigned long long)skipped_total;
  }
  __syncthreads();
  for (int p = (int)threadIdx.x >> 4; p < parents; p += 64)
    cs_emit_seg<16>(s_choice[p], first_row + p, s_off[p], nodes, low_values_last, scramble, (int)threadIdx.x & 15);
}


/* Classification of the children, deterministic: pool rows and the order of the solution check
 * depend on the child index only (block counts -> exclusive scan -> rows), never on which
 * workgroup finished first, so a search is reproducible run to run. */
__global__ __launch_bounds__(SB) void cs_classify_count(const csgpu_result *__restrict__ res, int children,
                                                        int *__restrict__ block_surv, int *__restrict__ block_comp,
                                                        int *__restrict__ block_cuts, int *__restrict__ block_props,
                                                        int *__restrict__ block_revs) {
  __shared__ int s_sum[5][SB / 64];
  const int t = threadIdx.x, i = blockIdx.x * SB + t;
  int status = -2, props = 0, revs = 0;
  if (i < children) {
    status = res[i].status;
    props = status >= 0 ? res[i].props : 0; /* propagations of consistent children only: on a != network those are the
                                             * reference's PROPS; an inconsistent child's count depends on the revision order */
    revs = res[i].revisions;
  }
  int ps = status > 0, pk = status == 0, pc = status == -1, pp = props, pr = revs;
  for (int o = 32; o > 0; o >>= 1) {
    ps += __shfl_xor(ps, o);
    pk += __shfl_xor(pk, o);
    pc += __shfl_xor(pc, o);
    pp += __shfl_xor(pp, o);
    pr += __shfl_xor(pr, o);
  }
  if ((t & 63) == 0) {
    s_sum[0][t >> 6] = ps; s_sum[1][t >> 6] = pk; s_sum[2][t >> 6] = pc; s_sum[3][t >> 6] = pp; s_sum[4][t >> 6] = pr;
  }
  __syncthreads();
  if (t < 5) {
    int v = 0;
    for (int w = 0; w < SB / 64; w++) v += s_sum[t][w];
    int *dst = t == 0 ? block_surv : (t == 1 ? block_comp : (t == 2 ? block_cuts : (t == 3 ? block_props : block_revs)));
    dst[blockIdx.x] = v; /* folded by cs_scan_classes: sums do not depend on any order */
  }
}

__global__ __launch_bounds__(SB) void cs_classify_assign(const csgpu_result *__restrict__ res, int children,
                                                         const int *__restrict__ surv_off,
                                                         const int *__restrict__ comp_off, int *__restrict__ surv_list,
                                                         int *__restrict__ complete_list) {
  __shared__ int s_surv[SB], s_comp[SB];
  const int t = threadIdx.x, i = blockIdx.x * SB + t;
  const int status = i < children ? res[i].status : -2;
  const int surv = status > 0, comp = status == 0;
  s_surv[t] = surv;
  s_comp[t] = comp;
  __syncthreads();
  for (int d = 1; d < SB; d <<= 1) {
    int a = t >= d ? s_surv[t - d] : 0, b = t >= d ? s_comp[t - d] : 0;
    __syncthreads();
    s_surv[t] += a;
    s_comp[t] += b;
    __syncthreads();
  }
  if (i < children) {
    if (surv) surv_list[surv_off[blockIdx.x] + s_surv[t] - 1] = i; /* survivor k goes to pool row new_top + k */
    if (comp) complete_list[comp_off[blockIdx.x] + s_comp[t] - 1] = i;
  }
}

/* -f true (prefer failing): the failure counts the branching rule looks at.  What the reference does per node
 * (csolve.c:455-465: the branching variable's prio-- when its assignment holds, prio++ when it fails;
 * propagate_term_confl, propagate.c:33-41: prio++ of the variable whose domain emptied), for a whole batch of children.
 * The reference's further bumps along its recursion stack (propagate.c:44-54) follow its depth-first order and have
 * no counterpart in a batch.  fail_var_known: the fixpoint kernel reports the emptied variable in result.rounds. */
/* prio[var] += delta for the lanes with var >= 0, ONE atomic per distinct variable of the wave: the children of a parent
 * share their variable and most failures empty the same one, so a lane each was 130,000 atomics on one word per
 * iteration of schedule-12 -- 1.5 ms at the ~88 atomics per microsecond a word sustains (2.3 ms per iteration with -f
 * true against 0.14 without) */
__device__ __forceinline__ void cs_wave_bump(int *__restrict__ prio, int var, int delta) {
  const int lane = (int)(threadIdx.x & 63);
  unsigned long long todo = __ballot(var >= 0);
  while (todo != 0ull) {
    const int leader = __builtin_ctzll(todo);
    const int lv = __builtin_amdgcn_readlane(var, leader);
    const unsigned long long same = __ballot(var == lv);
    const int sum = __popcll(__ballot(var == lv && delta > 0)) - __popcll(__ballot(var == lv && delta < 0));
    if (lane == leader && sum != 0) atomicAdd(&prio[lv], sum);
    todo &= ~same;
  }
}

__global__ __launch_bounds__(SB) void cs_prio_update(const csgpu_result *__restrict__ res, const csgpu_node *__restrict__ nodes,
                                                     int children, const unsigned long long *__restrict__ children_dev,
                                                     int n, int fail_var_known, int *__restrict__ prio) {
  if (children_dev != nullptr && (long long)*children_dev < (long long)children) children = (int)*children_dev;
  const int i = blockIdx.x * SB + threadIdx.x;
  if ((int)(blockIdx.x * SB) >= children) return; /* uniform over the workgroup; the waves below stay whole */
  int v = -1, failed_on = -1, delta = 0;
  if (i < children) {
    const csgpu_result r = res[i];
    v = nodes[i].var;
    if (v < 0 || v >= n) v = -1;
    delta = r.status >= 0 ? -1 : 1;
    if (v >= 0 && r.status < 0 && fail_var_known && r.rounds >= 0 && r.rounds < n && r.rounds != v) failed_on = r.rounds;
  }
  cs_wave_bump(prio, v, delta);
  cs_wave_bump(prio, failed_on, 1);
}

/* small iterations: cs_classify_count + cs_scan_classes + cs_classify_assign in one workgroup, the number of
 * children read from the device.  Same rows and the same order as the large path (tiles in child order). */
__global__ __launch_bounds__(1024) void cs_classify_small(const csgpu_result *__restrict__ res,
                                                          int *__restrict__ surv_list, int *__restrict__ complete_list,
                                                          unsigned long long *__restrict__ counters,
                                                          unsigned long long *__restrict__ burst) {
  __shared__ long long s_part[16];
  const int children = (int)counters[C_TOTAL_CHILDREN];
  long long carry = 0; /* survivors in the low half, complete children in the high half: one scan for both */
  long long cuts = 0, props = 0, revs = 0; /* per thread, reduced once at the end */
  /* consecutive children per thread and tile (sixteen, so that a MIN iteration of 10,000 children is one tile, was
   * measured no faster: this single workgroup is bound by what ONE CU reads, ~25 GB/s -- 160 KB of results are 7 us) */
  constexpr int PER = 4;
  for (int base = 0; base < children; base += 1024 * PER) {
    const int first = base + (int)threadIdx.x * PER;
    int status[PER];
    long long sum = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int i = first + q;
      status[q] = -2;
      if (i < children) {
        const csgpu_result r = res[i];
        status[q] = r.status;
        props += r.status >= 0 ? r.props : 0;
        revs += r.revisions;
        cuts += r.status == -1;
      }
      sum += (long long)(status[q] > 0) | ((long long)(status[q] == 0) << 32);
    }
    long long total;
    long long ex = carry + cs_block_excl_scan(sum, s_part, &total);
#pragma unroll
    for (int q = 0; q < PER; q++) {
      if (status[q] > 0) surv_list[ex & 0xffffffffll] = first + q;
      if (status[q] == 0) complete_list[ex >> 32] = first + q;
      ex += (long long)(status[q] > 0) | ((long long)(status[q] == 0) << 32);
    }
    carry += total;
  }
  long long t_cuts, t_props, t_revs;
  (void)cs_block_excl_scan(cuts, s_part, &t_cuts);
  (void)cs_block_excl_scan(props, s_part, &t_props);
  (void)cs_block_excl_scan(revs, s_part, &t_revs);
  if (threadIdx.x == 0) {
    counters[C_SURVIVORS] = (unsigned long long)(carry & 0xffffffffll);
    counters[C_COMPLETE] = (unsigned long long)(carry >> 32);
    counters[C_CUTS] = (unsigned long long)t_cuts;
    counters[C_PROPS] = (unsigned long long)t_props;
    counters[C_REVS] = (unsigned long long)t_revs;
    if (burst != nullptr) { /* device-driven iterations: the pool top and the running totals live on the device */
      const unsigned long long base = burst[B_TOP], top = base + (unsigned long long)(carry & 0xffffffffll);
      burst[B_SCATTER_BASE] = base;
      burst[B_TOP] = top;
      if (top > burst[B_PEAK]) burst[B_PEAK] = top;
      burst[B_CUTS] += (unsigned long long)t_cuts;
      burst[B_PROPS] += (unsigned long long)t_props;
      burst[B_REVS] += (unsigned long long)t_revs;
    }
  }
}

/* cs_accept + cs_pick_best for the complete children of a small iteration, by one workgroup, nothing read by the
 * host: counts the solutions, moves the incumbent, keeps a state that attains it (and, ANY: the first one).
 * Called by every thread of a workgroup of at least 256 threads (the first 256 work; uniform control flow). */
__device__ __forceinline__ void cs_accept_block(const cs_val *__restrict__ child_states, const int *__restrict__ list,
                                                const int *__restrict__ truth, int n, int objective, int obj_var,
                                                unsigned long long *__restrict__ counters,
                                                unsigned long long *__restrict__ burst, int32_t *__restrict__ solutions,
                                                long long max_solutions, int32_t *__restrict__ best_solution,
                                                int *__restrict__ best /* the incumbent: may be shared between engines */) {
  __shared__ long long s_key[256];
  __shared__ int s_cnt[256];
  __shared__ int s_pick;
  const int count = (int)counters[C_COMPLETE];
  if (count == 0) return; /* uniform */
  const int t = (int)threadIdx.x;
  const bool opt = objective == CS_OBJ_MIN || objective == CS_OBJ_MAX;
  /* key: (objective value, made "smaller is better") << 32 | child index: the minimum is the best value and,
   * among equals, the first child */
  long long key = 0x7fffffffffffffffll;
  int cnt = 0;
  if (t < 256)
    for (int i = t; i < count; i += 256) {
      if (truth != nullptr && truth[i] != 1) continue; /* truth == NULL: every complete child is a solution */
      cnt++;
      long long val = 0;
      if (opt) {
        const cs_val o = child_states[(size_t)list[i] * n + obj_var];
        val = objective == CS_OBJ_MIN ? (long long)o.lo : -(long long)o.hi;
      }
      const long long k = val * 4294967296ll + (long long)i;
      key = k < key ? k : key;
      if (objective != CS_OBJ_ANY && counters[C_STORED] < (unsigned long long)max_solutions) {
        const unsigned long long slot = atomicAdd(&counters[C_STORED], 1ull);
        if (slot < (unsigned long long)max_solutions)
          for (int v = 0; v < n; v++) solutions[(size_t)slot * n + v] = child_states[(size_t)list[i] * n + v].lo;
      }
    }
  if (t < 256) {
    s_key[t] = key;
    s_cnt[t] = cnt;
  }
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (t < d) {
      s_key[t] = s_key[t + d] < s_key[t] ? s_key[t + d] : s_key[t];
      s_cnt[t] += s_cnt[t + d];
    }
    __syncthreads();
  }
  if (t == 0) {
    s_pick = -1;
    const int accepted = s_cnt[0];
    if (accepted > 0) {
      const long long best_key = s_key[0];
      const int idx = (int)(best_key & 0xffffffffll);
      if (objective == CS_OBJ_ANY) {
        /* found_any (csolve.c:207-209): exactly one solution is accepted, the first in child order */
        if (counters[C_STORED] == 0ull) {
          counters[C_SOLUTIONS] += 1ull;
          counters[C_STORED] = 1ull;
          s_pick = idx;
        }
      } else {
        counters[C_SOLUTIONS] += (unsigned long long)accepted;
        if (opt) {
          const long long v = (best_key - (long long)idx) / 4294967296ll;
          const int val = objective == CS_OBJ_MIN ? (int)v : (int)-v;
          /* atomic: engines that share the incumbent accept concurrently */
          const int old = objective == CS_OBJ_MIN ? atomicMin(best, val) : atomicMax(best, val);
          if (objective == CS_OBJ_MIN ? val < old : val > old) {
            burst[B_IMPROVED] = 0x100000000ull | (unsigned)val; /* flag | the value the stored row attains */
            s_pick = idx;
          }
        }
      }
    }
  }
  __syncthreads();
  const int pick = s_pick;
  if (pick >= 0 && t < 256) {
    int32_t *out = objective == CS_OBJ_ANY ? solutions : best_solution;
    for (int v = t; v < n; v += 256) out[v] = child_states[(size_t)list[pick] * n + v].lo;
  }
  __syncthreads();
}

/* the accept of the LAST iteration of a burst (the others run at the head of the next cs_expand_burst) */
__global__ __launch_bounds__(256) void cs_accept_burst(const cs_val *__restrict__ child_states, const int *__restrict__ list,
                                                       const int *__restrict__ truth, int n, int objective, int obj_var,
                                                       unsigned long long *__restrict__ counters,
                                                       unsigned long long *__restrict__ burst,
                                                       int32_t *__restrict__ solutions, long long max_solutions,
                                                       int32_t *__restrict__ best_solution, int *__restrict__ best) {
  cs_accept_block(child_states, list, truth, n, objective, obj_var, counters, burst, solutions, max_solutions, best_solution,
                  best);
  if (threadIdx.x == 0) counters[C_COMPLETE] = 0ull; /* accepted: the next burst's first expansion must not do it again */
}

/* the head of a device-driven iteration -- how many parents, from which row -- as a function of `burst` alone, so
 * that every workgroup of a split expansion can decide it for itself */
struct cs_burst_head {
  int parents, error;
  long long first_row, iter;
};
__device__ __forceinline__ cs_burst_head cs_burst_decide(const unsigned long long *__restrict__ burst, bool done,
                                                         long long max_width, long long cap, long long room_limit) {
  cs_burst_head h;
  h.error = 0;
  const long long top = (long long)burst[B_TOP];
  /* a few parents while the pool is small (dive for a solution / an incumbent first), more once there is a
   * backlog of open states: a share of the pool (B_BACKLOG_DIV), within [B_LIMIT, B_LIMIT_MAX] (schedule-10: 0.7 s
   * instead of 1.6 s with 64 throughout; small searches lose a few ms) */
  long long limit = top / (long long)burst[B_BACKLOG_DIV];
  limit = limit < (long long)burst[B_LIMIT] ? (long long)burst[B_LIMIT] : limit;
  limit = limit > (long long)burst[B_LIMIT_MAX] ? (long long)burst[B_LIMIT_MAX] : limit;
  long long parents = top < limit ? top : limit;
  if (burst[B_BUDGET] == 0ull || burst[B_ERROR] != 0ull || done) parents = 0;
  if (parents > 0 && top - parents + parents * max_width > room_limit) { /* as one_iteration */
    const long long fit = max_width > 1 ? (room_limit - top) / (max_width - 1) : parents;
    parents = fit < 1 ? 1 : (fit < parents ? fit : parents);
    if (top - parents + parents * max_width > cap) {
      h.error = 1;
      parents = 0;
    }
  }
  h.parents = (int)parents;
  h.first_row = top - parents;
  h.iter = (long long)(burst[B_ITER_BASE] + burst[B_ITERS]);
  return h;
}

/* ---- device-driven iterations: what the host does around a small iteration, on the device ----
 * cs_expand_burst = the head of one_iteration (how many parents, does it fit) + cs_expand_small; the pool top,
 * the iteration budget and the running totals are in `burst`.  An iteration with nothing to do (pool empty,
 * budget used up, ANY already solved, error) leaves zero children, and every later kernel of it returns at once. */
__global__ __launch_bounds__(1024) void cs_expand_burst(const cs_val *__restrict__ pool, int n, csgpu_node *__restrict__ nodes,
                                                        unsigned long long *__restrict__ counters,
                                                        unsigned long long *__restrict__ burst, int objective,
                                                        long long max_width, long long cap, long long room_limit,
                                                        cs_holes H, const cs_val *__restrict__ child_states,
                                                        const int *__restrict__ complete_list,
                                                        const int *__restrict__ truth, int obj_var,
                                                        int32_t *__restrict__ solutions, long long max_solutions,
                                                        int32_t *__restrict__ best_solution, int *__restrict__ best) {
  __shared__ cs_choice s_choice[SMALL_PARENTS];
  __shared__ int s_off[SMALL_PARENTS];
  __shared__ long long s_part[16];
  __shared__ long long s_first, s_iter;
  __shared__ int s_parents;
  /* first the accept of the previous iteration's complete children (their root evaluation has run): it decides
   * whether ANY is done and moves the incumbent this iteration's fixpoints will see */
  cs_accept_block(child_states, complete_list, truth, n, objective, obj_var, counters, burst, solutions, max_solutions,
                  best_solution, best);
  if (threadIdx.x == 0) {
    const cs_burst_head h = cs_burst_decide(burst, objective == CS_OBJ_ANY && counters[C_STORED] != 0ull, max_width, cap, room_limit);
    if (h.error) burst[B_ERROR] = 1ull;
    s_parents = h.parents;
    s_first = h.first_row;
    s_iter = h.iter;
  }
  if (threadIdx.x < C_PER_ITERATION) counters[threadIdx.x] = 0ull;
  __syncthreads();
  const int parents = s_parents;
  if (parents == 0) return;
  const long long first_row = s_first;
  const int low_values_last = objective == CS_OBJ_MAX ? 0 : 1;
  const unsigned scramble =
      objective == CS_OBJ_ANY ? (unsigned)((unsigned long long)s_iter * 2654435761ull + 0x9e3779b9u) | 1u : 0u;
  /* sixteen lanes per parent, 64 parents per pass of the workgroup, four passes' rows in flight at a time (measured
   * no faster than one pass at a time: 1,024 parent rows are 300 KB through ONE CU, ~25 GB/s -- 12 us whatever the order) */
  for (int p0 = (int)threadIdx.x >> 4; p0 < parents; p0 += 256) {
    cs_branch_part part[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int p = p0 + 64 * k < parents ? p0 + 64 * k : parents - 1;
      part[k] = cs_branch_scan<16>(pool + (size_t)(first_row + p) * n, n, (int)threadIdx.x & 15, H);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int p = p0 + 64 * k;
      const cs_choice c = cs_branch_pick<16>(part[k], first_row + (p < parents ? p : parents - 1), n, H);
      if ((threadIdx.x & 15) == 0 && p < parents) s_choice[p] = c;
    }
  }
  __syncthreads();
  long long total, skipped_total;
  const int t = (int)threadIdx.x;
  (void)cs_block_excl_scan(t < parents ? (long long)s_choice[t].skipped : 0, s_part, &skipped_total);
  const long long ex = cs_block_excl_scan(t < parents ? (long long)s_choice[t].count : 0, s_part, &total);
  if (t < parents) s_off[t] = (int)ex;
  if (t == 0) {
    counters[C_TOTAL_CHILDREN] = (unsigned long long)total;
    counters[C_SKIPPED] = (unsigned long long)skipped_total;
    burst[B_TOP] = (unsigned long long)first_row;
    burst[B_ITERS] += 1ull;
    burst[B_BUDGET] -= 1ull;
    burst[B_NODES] += (unsigned long long)(total + skipped_total);
    burst[B_CUTS] += (unsigned long long)skipped_total;
  }
  __syncthreads();
  for (int p = (int)threadIdx.x >> 4; p < parents; p += 64)
    cs_emit_seg<16>(s_choice[p], first_row + p, s_off[p], nodes, low_values_last, scramble, (int)threadIdx.x & 15);
}

/* ---- the same iteration by up to BURST_WGS_MAX workgroups (MIN / MAX, whose iterations take 1,024 parents and more) ----
 * cs_burst_branch: workgroup g chooses for parents [64 g, 64 g + 64) and leaves their child counts' sum; workgroup 0
 * also runs the previous iteration's accept and publishes the head.  Nothing any workgroup READS to decide the head
 * is written here (the accept touches the incumbent, the solution counters and B_IMPROVED only), so all of them
 * decide alike without waiting for one another.
 * cs_burst_emit: workgroup g adds up its predecessors' sums (sixteen numbers) and writes its parents' children;
 * workgroup 0 moves the pool top and the running totals.  Same nodes in the same places as cs_expand_burst. */
__global__ __launch_bounds__(1024) void cs_burst_branch(const cs_val *__restrict__ pool, int n,
                                                        unsigned long long *__restrict__ counters,
                                                        unsigned long long *__restrict__ burst, int objective,
                                                        long long max_width, long long cap, long long room_limit,
                                                        cs_holes H, cs_choice *__restrict__ choice,
                                                        int *__restrict__ wg_sum, int *__restrict__ wg_skip,
                                                        const cs_val *__restrict__ child_states,
                                                        const int *__restrict__ complete_list,
                                                        const int *__restrict__ truth, int obj_var,
                                                        int32_t *__restrict__ solutions, long long max_solutions,
                                                        int32_t *__restrict__ best_solution, int *__restrict__ best) {
  __shared__ int s_cnt[BURST_PPW], s_skip[BURST_PPW];
  __shared__ long long s_first;
  __shared__ int s_parents;
  const int g = (int)blockIdx.x;
  if (g == 0) /* uniform within the workgroup */
    cs_accept_block(child_states, complete_list, truth, n, objective, obj_var, counters, burst, solutions, max_solutions,
                    best_solution, best);
  if (threadIdx.x == 0) {
    const cs_burst_head h = cs_burst_decide(burst, false, max_width, cap, room_limit);
    s_parents = h.parents;
    s_first = h.first_row;
    if (g == 0) {
      if (h.error) burst[B_ERROR] = 1ull;
      burst[B_D_PARENTS] = (unsigned long long)h.parents;
      burst[B_D_FIRST] = (unsigned long long)h.first_row;
      burst[B_D_ITER] = (unsigned long long)h.iter;
    }
  }
  if (g == 0 && threadIdx.x < C_PER_ITERATION) counters[threadIdx.x] = 0ull;
  __syncthreads();
  const int parents = s_parents;
  const int p = g * BURST_PPW + ((int)threadIdx.x >> 4);
  if (g * BURST_PPW >= parents) { /* uniform */
    if (threadIdx.x == 0) { wg_sum[g] = 0; wg_skip[g] = 0; }
    return;
  }
  const long long first_row = s_first;
  const int pc = p < parents ? p : parents - 1;
  const cs_choice c = cs_branch_seg<16>(pool + (size_t)(first_row + pc) * n, first_row + pc, n, (int)threadIdx.x & 15, H);
  if ((threadIdx.x & 15) == 0) {
    if (p < parents) choice[p] = c;
    s_cnt[threadIdx.x >> 4] = p < parents ? c.count : 0;
    s_skip[threadIdx.x >> 4] = p < parents ? c.skipped : 0;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    int cnt = s_cnt[threadIdx.x], skip = s_skip[threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o); skip += __shfl_xor(skip, o); }
    if (threadIdx.x == 0) { wg_sum[g] = cnt; wg_skip[g] = skip; }
  }
}

__global__ __launch_bounds__(1024) void cs_burst_emit(csgpu_node *__restrict__ nodes,
                                                      unsigned long long *__restrict__ counters,
                                                      unsigned long long *__restrict__ burst, int objective,
                                                      const cs_choice *__restrict__ choice,
                                                      const int *__restrict__ wg_sum, const int *__restrict__ wg_skip) {
  const int wgs = (int)gridDim.x; /* <= BURST_WGS_MAX */
  __shared__ cs_choice s_choice[BURST_PPW];
  __shared__ int s_off[BURST_PPW];
  const int g = (int)blockIdx.x;
  const int parents = (int)burst[B_D_PARENTS];
  if (g * BURST_PPW >= parents) return; /* uniform; parents == 0: the counters are zero already, nothing moves */
  const long long first_row = (long long)burst[B_D_FIRST];
  const long long iter = (long long)burst[B_D_ITER];
  if (threadIdx.x < 64) {
    const int t = (int)threadIdx.x, p = g * BURST_PPW + t;
    cs_choice c;
    c.var = -1; c.lo = 0; c.hi = 0; c.count = 0; c.a_lo = 0u; c.a_hi = 0u; c.holes = 0; c.skipped = 0;
    if (p < parents) c = choice[p];
    s_choice[t] = c;
    /* the children before this workgroup's parents, then before this parent */
    int before = 0, all = 0, all_skip = 0;
    for (int h = t; h < wgs; h += 64) {
      const int sum = wg_sum[h];
      before += h < g ? sum : 0;
      all += sum;
      all_skip += wg_skip[h];
    }
    int incl = c.count;
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (t >= d) incl += up;
    }
    for (int o = 32; o > 0; o >>= 1) {
      before += __shfl_xor(before, o);
      all += __shfl_xor(all, o);
      all_skip += __shfl_xor(all_skip, o);
    }
    s_off[t] = before + incl - c.count;
    if (g == 0 && t == 0) {
      counters[C_TOTAL_CHILDREN] = (unsigned long long)all;
      counters[C_SKIPPED] = (unsigned long long)all_skip;
      burst[B_TOP] = (unsigned long long)first_row;
      burst[B_ITERS] += 1ull;
      burst[B_BUDGET] -= 1ull;
      burst[B_NODES] += (unsigned long long)((long long)all + all_skip);
      burst[B_CUTS] += (unsigned long long)all_skip;
    }
  }
  __syncthreads();
  const int low_values_last = objective == CS_OBJ_MAX ? 0 : 1;
  const unsigned scramble =
      objective == CS_OBJ_ANY ? (unsigned)((unsigned long long)iter * 2654435761ull + 0x9e3779b9u) | 1u : 0u;
  const int q = (int)threadIdx.x >> 4;
  if (g * BURST_PPW + q < parents)
    cs_emit_seg<16>(s_choice[q], first_row + g * BURST_PPW + q, s_off[q], nodes, low_values_last, scramble, (int)threadIdx.x & 15);
}

/* cs_classify_small by BURST_CLASS_WGS workgroups: cs_burst_count leaves each workgroup's class counts (its share is
 * children / BURST_CLASS_WGS consecutive children), cs_burst_assign adds up its predecessors' and writes the lists in
 * child order, copies ITS survivors into the pool (no cs_scatter launch); its workgroup 0 moves the pool top and the
 * totals. */
__device__ __forceinline__ void cs_burst_share(int children, int g, int *beg, int *end) {
  const int chunk = (children + BURST_CLASS_WGS - 1) / BURST_CLASS_WGS;
  const long long b = (long long)g * chunk, e = b + chunk;
  *beg = b < children ? (int)b : children;
  *end = e < children ? (int)e : children;
}

__global__ __launch_bounds__(1024) void cs_burst_count(const csgpu_result *__restrict__ res,
                                                       const unsigned long long *__restrict__ counters,
                                                       int *__restrict__ wg_surv, int *__restrict__ wg_comp,
                                                       int *__restrict__ wg_cuts, int *__restrict__ wg_props,
                                                       int *__restrict__ wg_revs) {
  __shared__ long long s_part[16];
  int beg, end;
  cs_burst_share((int)counters[C_TOTAL_CHILDREN], (int)blockIdx.x, &beg, &end);
  long long classes = 0, cuts = 0, props = 0, revs = 0;
  for (int i = beg + (int)threadIdx.x; i < end; i += 1024) {
    const csgpu_result r = res[i];
    classes += (long long)(r.status > 0) | ((long long)(r.status == 0) << 32);
    props += r.status >= 0 ? r.props : 0;
    revs += r.revisions;
    cuts += r.status == -1;
  }
  long long t_classes, t_cuts, t_props, t_revs;
  (void)cs_block_excl_scan(classes, s_part, &t_classes);
  (void)cs_block_excl_scan(cuts, s_part, &t_cuts);
  (void)cs_block_excl_scan(props, s_part, &t_props);
  (void)cs_block_excl_scan(revs, s_part, &t_revs);
  if (threadIdx.x == 0) {
    wg_surv[blockIdx.x] = (int)(t_classes & 0xffffffffll);
    wg_comp[blockIdx.x] = (int)(t_classes >> 32);
    wg_cuts[blockIdx.x] = (int)t_cuts;
    wg_props[blockIdx.x] = (int)t_props;
    wg_revs[blockIdx.x] = (int)t_revs;
  }
}

__global__ __launch_bounds__(1024) void cs_burst_assign(const csgpu_result *__restrict__ res,
                                                        int *__restrict__ surv_list, int *__restrict__ complete_list,
                                                        unsigned long long *__restrict__ counters,
                                                        unsigned long long *__restrict__ burst,
                                                        const int *__restrict__ wg_surv, const int *__restrict__ wg_comp,
                                                        const int *__restrict__ wg_cuts, const int *__restrict__ wg_props,
                                                        const int *__restrict__ wg_revs,
                                                        const cs_val *__restrict__ child_states, cs_val *__restrict__ pool,
                                                        int n, const unsigned long long *__restrict__ child_forb,
                                                        unsigned long long *__restrict__ pool_forb, int fw) {
  __shared__ long long s_part[16];
  const int g = (int)blockIdx.x;
  int beg, end;
  cs_burst_share((int)counters[C_TOTAL_CHILDREN], g, &beg, &end);
  /* survivors | complete children << 32 of every workgroup, one per thread: the sum of the predecessors' and of all */
  const int t = (int)threadIdx.x;
  const long long mine = t < BURST_CLASS_WGS ? (long long)wg_surv[t] | ((long long)wg_comp[t] << 32) : 0ll;
  long long carry, classes_all;
  (void)cs_block_excl_scan(t < g ? mine : 0ll, s_part, &carry);
  (void)cs_block_excl_scan(mine, s_part, &classes_all);
  const int first_surv = (int)(carry & 0xffffffffll);
  for (int base = beg; base < end; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int status = i < end ? res[i].status : -2;
    const long long x = (long long)(status > 0) | ((long long)(status == 0) << 32);
    long long total;
    const long long ex = carry + cs_block_excl_scan(x, s_part, &total);
    if (status > 0) surv_list[ex & 0xffffffffll] = i;
    if (status == 0) complete_list[ex >> 32] = i;
    carry += total;
  }
  /* cs_scatter for this workgroup's survivors: rows first_surv .. of the new pool top (the rows of the iteration's
   * parents, B_D_FIRST, are the first to be overwritten: LIFO), walked flat so that small models fill the lanes */
  const int here = (int)(carry & 0xffffffffll) - first_surv;
  if (here > 0) { /* uniform */
    __syncthreads(); /* this workgroup's part of surv_list is written */
    const long long row0 = (long long)burst[B_D_FIRST] + first_surv;
    const int *src = surv_list + first_surv;
    {
      const int total = here * n;
      int c = (int)threadIdx.x / n, v = (int)threadIdx.x - c * n;
      const int dc = 1024 / n, dv = 1024 - dc * n;
      cs_val *dst = pool + (size_t)row0 * n;
      for (int e = (int)threadIdx.x; e < total; e += 1024) {
        dst[e] = child_states[(size_t)src[c] * n + v];
        c += dc; v += dv;
        if (v >= n) { v -= n; c++; }
      }
    }
    if (fw > 0) {
      const int nf = n * fw, total = here * nf;
      int c = (int)threadIdx.x / nf, k = (int)threadIdx.x - c * nf;
      const int dc = 1024 / nf, dk = 1024 - dc * nf;
      unsigned long long *dst = pool_forb + (size_t)row0 * nf;
      for (int e = (int)threadIdx.x; e < total; e += 1024) {
        dst[e] = child_forb[(size_t)src[c] * nf + k];
        c += dc; k += dk;
        if (k >= nf) { k -= nf; c++; }
      }
    }
  }
  if (g != 0) return; /* uniform */
  long long cuts, props, revs;
  (void)cs_block_excl_scan(t < BURST_CLASS_WGS ? (long long)wg_cuts[t] : 0ll, s_part, &cuts);
  (void)cs_block_excl_scan(t < BURST_CLASS_WGS ? (long long)wg_props[t] : 0ll, s_part, &props);
  (void)cs_block_excl_scan(t < BURST_CLASS_WGS ? (long long)wg_revs[t] : 0ll, s_part, &revs);
  if (threadIdx.x == 0) {
    const long long surv = classes_all & 0xffffffffll, comp = classes_all >> 32;
    counters[C_SURVIVORS] = (unsigned long long)surv;
    counters[C_COMPLETE] = (unsigned long long)comp;
    counters[C_CUTS] = (unsigned long long)cuts;
    counters[C_PROPS] = (unsigned long long)props;
    counters[C_REVS] = (unsigned long long)revs;
    const unsigned long long base = burst[B_TOP], top = base + (unsigned long long)surv;
    burst[B_SCATTER_BASE] = base;
    burst[B_TOP] = top;
    if (top > burst[B_PEAK]) burst[B_PEAK] = top;
    burst[B_CUTS] += (unsigned long long)cuts;
    burst[B_PROPS] += (unsigned long long)props;
    burst[B_REVS] += (unsigned long long)revs;
  }
}

/* copy survivor k (child surv_list[k]) into pool row new_top + k: a workgroup takes cpb (at most SB) consecutive
 * survivors and walks their cpb * n elements flat, so that small models fill the lanes too.  The number of
 * survivors is on the device; the grid is sized for the number of children. */
__global__ __launch_bounds__(SB) void cs_scatter(const cs_val *__restrict__ child_states, const int *__restrict__ surv_list,
                                                 const unsigned long long *__restrict__ counters, long long new_top, int n,
                                                 cs_val *__restrict__ pool,
                                                 const unsigned long long *__restrict__ child_forb,
                                                 unsigned long long *__restrict__ pool_forb, int fw, int cpb,
                                                 const unsigned long long *__restrict__ new_top_dev) {
  __shared__ int s_src[SB];
  if (new_top_dev != nullptr) new_top = (long long)*new_top_dev;
  const long long survivors = (long long)counters[C_SURVIVORS];
  const long long base = (long long)blockIdx.x * cpb;
  if (base >= survivors) return;
  const int here = survivors - base < cpb ? (int)(survivors - base) : cpb;
  if ((int)threadIdx.x < here) s_src[threadIdx.x] = surv_list[base + threadIdx.x];
  __syncthreads();
  const unsigned total = (unsigned)here * (unsigned)n;
  cs_val *dst = pool + (size_t)(new_top + base) * n;
  for (unsigned e = threadIdx.x; e < total; e += SB) {
    const unsigned c = e / (unsigned)n, v = e - c * (unsigned)n;
    dst[e] = child_states[(size_t)s_src[c] * n + v];
  }
  if (fw > 0) {
    const unsigned nf = (unsigned)n * (unsigned)fw, total_f = (unsigned)here * nf;
    unsigned long long *fd = pool_forb + (size_t)(new_top + base) * nf;
    for (unsigned e = threadIdx.x; e < total_f; e += SB) {
      const unsigned c = e / nf, k = e - c * nf;
      fd[e] = child_forb[(size_t)s_src[c] * nf + k];
    }
  }
}

/* nodes {-1,0,0,row}: "rebuild the forbidden sets of this state" */
__global__ void cs_fill_rebuild(csgpu_node *__restrict__ nodes, long long first_row, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  csgpu_node nd;
  nd.var = -1; nd.lo = 0; nd.hi = 0; nd.parent = (int)(first_row + i);
  nodes[i] = nd;
}

/* one wave per complete child: gather it for the root evaluation */
__global__ __launch_bounds__(SB) void cs_gather_complete(const cs_val *__restrict__ child_states,
                                                         const int *__restrict__ list, int count, int n,
                                                         cs_val *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (SB / 64) + (threadIdx.x >> 6);
  if (i >= count) return;
  const cs_val *src = child_states + (size_t)list[i] * n;
  cs_val *dst = out + (size_t)i * n;
  for (int v = lane; v < n; v += 64) dst[v] = src[v];
}

/* accept the complete children whose root evaluated to true: count, incumbent, store some.
 * One thread per complete child; one atomic per wave for the count and the incumbent. */
__global__ __launch_bounds__(SB) void cs_accept(const cs_val *__restrict__ complete, const int *__restrict__ truth,
                                                int count, int n, int objective, int obj_var,
                                                unsigned long long *__restrict__ counters,
                                                int32_t *__restrict__ solutions, long long max_solutions,
                                                const int *__restrict__ list /* nullable: child i is row list[i] */) {
  const int lane = threadIdx.x & 63;
  int i = blockIdx.x * SB + threadIdx.x;
  if (objective == CS_OBJ_ANY) {
    /* found_any (csolve.c:207-209): exactly one solution is accepted -- the first complete child,
     * in child order, whose root evaluates to true; one thread does the scan */
    if (i != 0) return;
    int first = -1;
    for (int k = 0; k < count && first < 0; k++)
      if (truth == nullptr || truth[k] == 1) first = k;
    if (first < 0 || counters[C_STORED] != 0ull) return;
    counters[C_SOLUTIONS] += 1ull;
    counters[C_STORED] = 1ull;
    for (int v = 0; v < n; v++) solutions[v] = complete[(size_t)(list != nullptr ? list[first] : first) * n + v].lo;
    return;
  }
  /* the count goes through LDS: one device atomic per workgroup (a word takes about 88 atomics per microsecond, and an
   * ALL iteration of queens-16 accepts 80,000 children: one atomic per wave was 14 of the kernel's 17 us) */
  __shared__ unsigned s_accepted;
  if (threadIdx.x == 0) s_accepted = 0u;
  __syncthreads();
  const bool ok = i < count && (truth == nullptr || truth[i] == 1); /* truth == NULL: every complete child is a solution */
  const size_t row = ok ? (size_t)(list != nullptr ? list[i] : i) * n : 0;
  const unsigned long long mask = __ballot(ok);
  const int accepted = __popcll(mask), leader = mask != 0ull ? __builtin_ctzll(mask) : 0;
  if (mask != 0ull && lane == leader) atomicAdd(&s_accepted, (unsigned)accepted);
  __syncthreads();
  if (threadIdx.x == 0 && s_accepted != 0u) atomicAdd(&counters[C_SOLUTIONS], (unsigned long long)s_accepted);
  if (mask == 0ull) return;
  if (objective == CS_OBJ_MIN || objective == CS_OBJ_MAX) {
    int val = objective == CS_OBJ_MIN ? 0x7fffffff : (int)0x80000000;
    if (ok) val = objective == CS_OBJ_MIN ? complete[row + obj_var].lo : complete[row + obj_var].hi;
    for (int o = 32; o > 0; o >>= 1) {
      const int other = __shfl_xor(val, o);
      val = objective == CS_OBJ_MIN ? (other < val ? other : val) : (other > val ? other : val);
    }
    if (lane == leader) {
      if (objective == CS_OBJ_MIN) atomicMin((int *)&counters[C_BEST], val);
      else atomicMax((int *)&counters[C_BEST], val);
    }
  }
  long long slot0 = max_solutions;
  if (lane == leader) {
    /* which solutions are kept may vary; their count does not.  Once the store is full nobody asks for a slot */
    if (counters[C_STORED] < (unsigned long long)max_solutions)
      slot0 = (long long)atomicAdd(&counters[C_STORED], (unsigned long long)accepted);
  }
  slot0 = __shfl(slot0, leader);
  if (ok) {
    /* rank among the accepted lanes below this one: mbcnt, not a 64-bit shift by the lane number (tools/k4_fault_repro.md) */
    const long long slot = slot0 + (long long)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    if (slot < max_solutions)
      for (int v = 0; v < n; v++) solutions[(size_t)slot * n + v] = complete[row + v].lo;
  }
}

/* one wave: the first accepted complete child whose objective value equals the incumbent */
__global__ void cs_pick_best(const cs_val *__restrict__ complete, const int *__restrict__ truth, int count, int n,
                             int objective, int obj_var, int best, int32_t *__restrict__ out) {
  const int lane = threadIdx.x;
  int pick = -1;
  for (int k = 0; k < count && pick < 0; k++) {
    const cs_val o = complete[(size_t)k * n + obj_var];
    if (truth[k] == 1 && (objective == CS_OBJ_MIN ? o.lo : o.hi) == best) pick = k;
  }
  if (pick < 0) return;
  for (int v = lane; v < n; v += 64) out[v] = complete[(size_t)pick * n + v].lo;
}

/* move the newest `count` rows into the hole left by taking the oldest ones */
__global__ __launch_bounds__(SB) void cs_move_rows(unsigned long long *__restrict__ pool, long long src_row,
                                                   long long dst_row, int count, int words_per_row) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (SB / 64) + (threadIdx.x >> 6);
  if (i >= count) return;
  const unsigned long long *src = pool + (size_t)(src_row + i) * words_per_row;
  unsigned long long *dst = pool + (size_t)(dst_row + i) * words_per_row;
  for (int v = lane; v < words_per_row; v += 64) dst[v] = src[v];
}

extern "C" void csgpu_search_free(csgpu_search *s) {
  if (s == NULL) return;
  if (s->borrowers > 0) { /* other engines' kernels and graphs still write the incumbent word in s->d_counters */
    s->free_pending = 1;
    return;
  }
  if (s->lender != NULL) {
    csgpu_search *l = s->lender;
    s->lender = NULL;
    if (--l->borrowers == 0 && l->free_pending) csgpu_search_free(l);
  }
  (void)hipSetDevice(s->device);
  if (s->counted) csgpu_internal_engine_ref(s->m, -1);
  (void)hipFree(s->pool_forb); (void)hipFree(s->d_child_forb); (void)hipFree(s->d_rebuild_nodes);
  (void)hipFree(s->pool); (void)hipFree(s->d_choice); (void)hipFree(s->d_child_off); (void)hipFree(s->d_block_sum); (void)hipFree(s->d_block_skip);
  (void)hipFree(s->d_nodes); (void)hipFree(s->d_child_states); (void)hipFree(s->d_complete_states);
  (void)hipFree(s->d_results); (void)hipFree(s->d_dest); (void)hipFree(s->d_complete_list); (void)hipFree(s->d_truth);
  (void)hipFree(s->d_block_surv); (void)hipFree(s->d_block_comp); (void)hipFree(s->d_surv_off); (void)hipFree(s->d_comp_off);
  (void)hipFree(s->d_block_cuts); (void)hipFree(s->d_block_props); (void)hipFree(s->d_block_revs);
  (void)hipFree(s->d_counters); (void)hipFree(s->d_solutions);
  (void)hipFree(s->seed);
  (void)hipFree(s->d_best_solution);
  (void)hipFree(s->d_burst);
  (void)hipFree(s->d_prio);
  (void)hipFree(s->d_fill); (void)hipFree(s->d_ticket); (void)hipFree(s->d_wstat); (void)hipFree(s->d_step_out);
  if (s->h_step_out) (void)hipHostFree(s->h_step_out);
  if (s->h_burst) (void)hipHostFree(s->h_burst);
  if (s->burst_exec) (void)hipGraphExecDestroy(s->burst_exec);
  if (s->burst_stream) (void)hipStreamDestroy(s->burst_stream);
  free(s);
}

extern "C" int csgpu_search_create(const csgpu_model *m, int64_t pool_capacity, int64_t max_children,
                                   csgpu_search **out) {
  if (m == NULL || out == NULL || pool_capacity < 1 || max_children < 1) return fail(CSGPU_E_ARG, "bad argument");
  const int n = csgpu_model_num_vars(m);
  if (n <= 0) return fail(CSGPU_E_ARG, "model without variables");
  csgpu_search *s = (csgpu_search *)calloc(1, sizeof *s);
  s->m = m;
  s->n = n;
  if (hipGetDevice(&s->device) != hipSuccess) { free(s); return fail(CSGPU_E_HIP, "hipGetDevice"); }
  csgpu_internal_engine_ref(m, 1);
  s->counted = 1;
  s->objective = csgpu_model_objective(m);
  s->obj_var = csgpu_model_objective_var(m);
  /* widest root interval bounds the branching factor (domains only shrink below the root) */
  csgpu_val *dom = (csgpu_val *)malloc((size_t)n * sizeof *dom);
  csgpu_model_get_domains(m, dom);
  /* children of one parent: the interval's values if it has at most SPLIT_WIDTH of them, else 2.
   * A wide root interval gets narrow by halving, so SPLIT_WIDTH bounds every variable that
   * starts wider than that. */
  s->max_width = 2;
  for (int v = 0; v < n; v++) {
    int64_t w = (int64_t)dom[v].hi - (int64_t)dom[v].lo + 1;
    if (w > SPLIT_WIDTH) w = SPLIT_WIDTH;
    if (w > s->max_width) s->max_width = w;
  }
  free(dom);
  if (max_children < s->max_width) max_children = s->max_width;
  if (max_children > 0x3fffffff) return fail(CSGPU_E_LIMIT, "max_children too large");
  s->max_children = max_children;
  s->max_parents = max_children / s->max_width;
  /* ALL walks the whole tree anyway: widest batches.  ANY/MIN/MAX profit from going deep first
   * (a first solution / a good incumbent early prunes everything else), so only the newest 64
   * open states are expanded per iteration. */
  s->avg_children = (double)s->max_width;
  s->parents_limit = csgpu_model_objective(m) == CS_OBJ_ALL ? s->max_parents : 64;
  if (s->parents_limit > s->max_parents) s->parents_limit = s->max_parents;
  s->parents_max = s->parents_limit;
  if (csgpu_model_objective(m) == CS_OBJ_MIN || csgpu_model_objective(m) == CS_OBJ_MAX) { /* ANY stays depth-first */
    /* parents of a device-driven MIN / MAX iteration once the pool holds a backlog (tuning: CSGPU_SEARCH_PARENTS_MAX) */
    /* schedule-12 MIN: 9.7 s with 256, 7.3 s with 512, 6.3 s with 1,024 (round 3, single-workgroup bookkeeping); with the
     * bookkeeping over many workgroups 3.84 s with 1,024 and 3.37 s with 2,048 (10 % more nodes in 20 % fewer, fuller
     * iterations).  What holds an iteration down after that is the share of the pool it takes -- a sixteenth of it was
     * rarely 2,048 states -- and the child buffer (parents x widest interval must fit): with a QUARTER of the pool per
     * iteration schedule-12 takes 2.30 s at 2,048 parents, 1.80 s at 4,096 (2^20 children), 1.68 s at 8,192 (2^21;
     * 1.53e9 nodes in 11,001 iterations of 150 us: the fixpoint kernel at its throughput) and 1.87 s at 16,384 -- from
     * there on the nodes the breadth costs (2.1e9) outweigh the launches it saves */
    int64_t want = 8192; /* see the next paragraph of this comment, and B_BACKLOG_DIV in run_burst */
    {
      const char *e = getenv("CSGPU_SEARCH_PARENTS_MAX");
      if (e != NULL && atoll(e) > 0) want = atoll(e);
      const char *es = getenv("CSGPU_SEARCH_BURST_SPLIT");
      const int64_t most = es != NULL && es[0] == '0' ? SMALL_PARENTS : BURST_PARENTS_MAX; /* what one workgroup scans */
      if (want > most) want = most;
    }
    s->parents_max = want < s->max_parents ? want : s->max_parents;
    if (s->parents_max < s->parents_limit) s->parents_max = s->parents_limit;
  }
  if (pool_capacity < max_children + 1) pool_capacity = max_children + 1;
  if (pool_capacity > 0x7fffffff) return fail(CSGPU_E_LIMIT, "pool_capacity too large");
  s->cap = pool_capacity;
  s->max_solutions = 1024;
  s->restart_base = s->objective == CS_OBJ_ANY ? 64 : 0;
  s->luby_threshold = 1;
  s->luby_counter = 1;
  s->st.best = s->objective == CS_OBJ_MIN ? CS_DOM_MAX : (s->objective == CS_OBJ_MAX ? CS_DOM_MIN : 0);
  const size_t row = (size_t)n * sizeof(cs_val);
  hipError_t e;
#define ALLOC(ptr, bytes)                                                      \
  if ((e = hipMalloc((void **)&(ptr), (bytes))) != hipSuccess) {               \
    csgpu_search_free(s);                                                      \
    return fail(CSGPU_E_HIP, hipGetErrorString(e));                            \
  }
  ALLOC(s->pool, row * (size_t)s->cap);
  s->fw = s->obj_var < 0 ? csgpu_model_forbidden_words(m) : 0;
  {
    /* CSGPU_SEARCH_SETS=0: interval rows only in the pool of the separate-kernel path too (the fixpoints then run on
     * kernel 7 / kernel 5's rebuild entry, and no child is cut without a launch) */
    const char *es = getenv("CSGPU_SEARCH_SETS");
    if (es != NULL && es[0] == '0') s->fw = 0;
  }
  {
    const char *ef = getenv("CSGPU_SEARCH_FUSED"), *ev = getenv("CSGPU_SEARCH_EVAL");
    s->fused = s->objective == CS_OBJ_ALL && csgpu_internal_step_kind(m) != 0 && !(ef != NULL && ef[0] == '0') &&
               !(ev != NULL && ev[0] == '1');
  }
  s->stage_rows = max_children;
  if (s->fused) {
    /* twice the rows a frontier's children may have: a wave's region then takes the worst case of a ticket of 64 parents
     * (fewer parents per ticket and the ticket counter limits the launch, cs_capi.hip) */
    s->stage_rows = 2 * max_children;
    const char *e = getenv("CSGPU_STEP_STAGE_MULT"); /* tuning: staging rows per max_children */
    if (e != NULL && atoi(e) >= 1) s->stage_rows = max_children * atoi(e);
    /* one wave per parent (33 to 256 variables): room for the children of four parents in every wave's region */
    if (s->stage_rows < csgpu_internal_step_stage_rows(m)) s->stage_rows = csgpu_internal_step_stage_rows(m);
    if (s->stage_rows > s->cap) s->stage_rows = s->cap;
  }
  s->surv_per_parent = (double)s->max_width;
  if (s->fused) {
    s->fw = 0; /* interval rows only: no sets in the pool, nothing to rebuild for states put from outside */
    const int64_t waves = csgpu_internal_step_waves(m);
    ALLOC(s->d_fill, sizeof(uint32_t) * (size_t)waves);
    ALLOC(s->d_wstat, sizeof(uint64_t) * 8 * (size_t)waves);
    ALLOC(s->d_ticket, 1024); /* sixteen counters on their own 64-byte lines */
    ALLOC(s->d_step_out, sizeof(uint64_t) * 8);
    if ((e = hipHostMalloc((void **)&s->h_step_out, sizeof(uint64_t) * 8, 0)) != hipSuccess) {
      csgpu_search_free(s);
      return fail(CSGPU_E_HIP, hipGetErrorString(e));
    }
  }
  if (s->fw > 0) {
    ALLOC(s->pool_forb, (size_t)n * s->fw * 8 * (size_t)s->cap);
    ALLOC(s->d_child_forb, (size_t)n * s->fw * 8 * (size_t)max_children);
    ALLOC(s->d_rebuild_nodes, sizeof(csgpu_node) * (size_t)max_children);
  }
  /* up to max_children / 2 parents when every parent has two children */
  ALLOC(s->d_choice, sizeof(cs_choice) * (size_t)max_children);
  ALLOC(s->d_child_off, sizeof(int) * ((size_t)max_children + 1));
  ALLOC(s->d_block_sum, sizeof(int) * ((size_t)max_children + 1));
  ALLOC(s->d_block_skip, sizeof(int) * ((size_t)max_children + 1));
  ALLOC(s->d_nodes, sizeof(csgpu_node) * (size_t)max_children);
  ALLOC(s->d_child_states, row * (size_t)(s->fused ? s->stage_rows : max_children));
  if (!s->fused) ALLOC(s->d_complete_states, row * (size_t)max_children);
  ALLOC(s->d_results, sizeof(csgpu_result) * (size_t)max_children);
  ALLOC(s->d_dest, sizeof(int) * (size_t)max_children);
  ALLOC(s->d_complete_list, sizeof(int) * (size_t)max_children);
  ALLOC(s->d_truth, sizeof(int) * (size_t)max_children);
  {
    size_t blocks = ((size_t)max_children + SB - 1) / SB + 1;
    if (blocks < BURST_CLASS_WGS) blocks = BURST_CLASS_WGS;
    ALLOC(s->d_block_surv, sizeof(int) * blocks);
    ALLOC(s->d_block_comp, sizeof(int) * blocks);
    ALLOC(s->d_block_cuts, sizeof(int) * blocks);
    ALLOC(s->d_block_props, sizeof(int) * blocks);
    ALLOC(s->d_block_revs, sizeof(int) * blocks);
    ALLOC(s->d_surv_off, sizeof(int) * (blocks + 1));
    ALLOC(s->d_comp_off, sizeof(int) * (blocks + 1));
  }
  ALLOC(s->d_counters, sizeof(unsigned long long) * C_COUNT);
  s->d_best = (int *)(s->d_counters + C_BEST);
  ALLOC(s->d_solutions, sizeof(int32_t) * (size_t)n * (size_t)s->max_solutions);
  ALLOC(s->d_best_solution, sizeof(int32_t) * (size_t)n);
#undef ALLOC
  HIP_OK(hipMemset(s->d_counters, 0, sizeof(unsigned long long) * C_COUNT));
  HIP_OK(hipMemcpy(s->d_best, &s->st.best, sizeof(int), hipMemcpyHostToDevice));
  s->order = 1;
  s->holes.order = 1;
  s->holes.prio = NULL;
  s->holes.pool_forb = NULL;
  s->holes.root_lo = csgpu_internal_root_lo(m);
  {
    const char *e = getenv("CSGPU_SEARCH_HOLES");
    if (s->fw == 1 && s->holes.root_lo != NULL && !(e != NULL && e[0] == '0')) s->holes.pool_forb = s->pool_forb;
  }
  HIP_OK(hipMalloc((void **)&s->d_burst, sizeof(unsigned long long) * B_COUNT));
  HIP_OK(hipHostMalloc((void **)&s->h_burst, sizeof(unsigned long long) * (B_COUNT + C_COUNT + 1), 0));
  HIP_OK(hipStreamCreate(&s->burst_stream));
  {
    const char *e = getenv("CSGPU_SEARCH_BURST");
    s->burst_off = e != NULL && e[0] == '0';
    const char *es = getenv("CSGPU_SEARCH_BURST_SPLIT");
    s->burst_split = !(es != NULL && es[0] == '0');
    const char *ev = getenv("CSGPU_SEARCH_EVAL");
    s->eval_always = ev != NULL && ev[0] == '1';
    e = getenv("CSGPU_SEARCH_GRAPH");
    s->graph_off = e != NULL && e[0] == '0';
    int64_t info[8];
    s->burst_no_eval = !s->eval_always && csgpu_model_device_info(m, info) == CSGPU_OK && info[2] == 0;
  }
  *out = s;
  return CSGPU_OK;
}

static int search_put(csgpu_search *s, const csgpu_val *d_states, int64_t count);

extern "C" int csgpu_search_put(csgpu_search *s, const csgpu_val *d_states, int64_t count) {
  if (s == NULL || (count > 0 && d_states == NULL) || count < 0) return fail(CSGPU_E_ARG, "bad argument");
  if (hipSetDevice(s->device) != hipSuccess) return fail(CSGPU_E_HIP, "hipSetDevice");
  const auto t0 = std::chrono::steady_clock::now();
  int rc = search_put(s, d_states, count);
  if (rc == CSGPU_OK && count > 0) {
    HIP_OK(hipDeviceSynchronize()); /* the rebuild launches are part of the cost */
    s->put_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    s->put_states += count;
  }
  return rc;
}

extern "C" int csgpu_search_put_cost(const csgpu_search *s, double *seconds, int64_t *states) {
  if (s == NULL || seconds == NULL || states == NULL) return fail(CSGPU_E_ARG, "bad argument");
  *seconds = s->put_seconds;
  *states = s->put_states;
  return CSGPU_OK;
}

static int search_put(csgpu_search *s, const csgpu_val *d_states, int64_t count) {
  if (s->top + count > s->cap) return fail(CSGPU_E_LIMIT, "state pool is full");
  if (count > 0)
    HIP_OK(hipMemcpy(s->pool + (size_t)s->top * s->n, d_states, (size_t)count * s->n * sizeof(cs_val),
                     hipMemcpyDeviceToDevice));
  if (count > 0 && s->fused) { /* the pool of the fused path holds engine rows (cs_step.hip.h) */
    const int rc = csgpu_internal_step_import(s->m, (csgpu_val *)s->pool, s->top, count, NULL);
    if (rc != CSGPU_OK) return rc;
  }
  if (count > 0 && s->fw > 0) {
    /* states arriving from outside (the root, another rank) carry no sets: rebuild them in place,
     * max_children rows at a time (the batch buffers are free between iterations) */
    for (int64_t done = 0; done < count; done += s->max_children) {
      const int64_t k = count - done < s->max_children ? count - done : s->max_children;
      hipLaunchKernelGGL(cs_fill_rebuild, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, 0, s->d_rebuild_nodes,
                         (long long)(s->top + done), (int)k);
      int rc = csgpu_propagate_batch_fb(s->m, (const csgpu_val *)s->pool, NULL, s->d_rebuild_nodes,
                                        (csgpu_val *)s->d_child_states, (uint64_t *)s->d_child_forb, s->d_results, k,
                                        NULL);
      if (rc != CSGPU_OK) return rc;
      HIP_OK(hipMemcpy(s->pool_forb + (size_t)(s->top + done) * s->n * s->fw, s->d_child_forb,
                       (size_t)k * s->n * s->fw * 8, hipMemcpyDeviceToDevice));
    }
  }
  if (count > 0 && (s->restart_base > 0 || s->restart_on_improvement) && !s->st.iterations) {
    /* remember what the search was started from (only states put before the first iteration) */
    if (s->seed_count + count > s->seed_cap) {
      const int64_t cap = (s->seed_count + count) * 2;
      cs_val *grown = NULL;
      HIP_OK(hipMalloc((void **)&grown, (size_t)cap * s->n * sizeof(cs_val)));
      if (s->seed_count > 0)
        HIP_OK(hipMemcpy(grown, s->seed, (size_t)s->seed_count * s->n * sizeof(cs_val), hipMemcpyDeviceToDevice));
      (void)hipFree(s->seed);
      s->seed = grown;
      s->seed_cap = cap;
    }
    HIP_OK(hipMemcpy(s->seed + (size_t)s->seed_count * s->n, d_states, (size_t)count * s->n * sizeof(cs_val),
                     hipMemcpyDeviceToDevice));
    s->seed_count += count;
  }
  s->top += count;
  if (s->top > s->peak) s->peak = s->top;
  return CSGPU_OK;
}

extern "C" int csgpu_search_reset(csgpu_search *s) {
  if (s == NULL) return fail(CSGPU_E_ARG, "bad argument");
  if (hipSetDevice(s->device) != hipSuccess) return fail(CSGPU_E_HIP, "hipSetDevice");
  s->top = 0;
  s->peak = 0;
  memset(&s->st, 0, sizeof s->st);
  s->st.best = s->objective == CS_OBJ_MIN ? CS_DOM_MAX : (s->objective == CS_OBJ_MAX ? CS_DOM_MIN : 0);
  s->seed_count = 0;
  s->since_restart = 0;
  s->luby_threshold = 1;
  s->luby_counter = 1;
  s->put_seconds = 0.0;
  s->put_states = 0;
  s->have_best_solution = 0;
  s->pending_complete = 0;
  s->surv_per_parent = (double)s->max_width;
  s->stored_seen = 0;
  if (s->d_prio != NULL) HIP_OK(hipMemset(s->d_prio, 0, sizeof(int) * (size_t)s->n));
  HIP_OK(hipMemset(s->d_counters, 0, sizeof(unsigned long long) * C_COUNT));
  HIP_OK(hipMemcpy(s->d_best, &s->st.best, sizeof(int), hipMemcpyHostToDevice));
  return CSGPU_OK;
}

extern "C" int csgpu_search_put_host(csgpu_search *s, const csgpu_val *states, int64_t count) {
  if (s == NULL || (count > 0 && states == NULL) || count < 0) return fail(CSGPU_E_ARG, "bad argument");
  if (hipSetDevice(s->device) != hipSuccess) return fail(CSGPU_E_HIP, "hipSetDevice");
  if (count == 0) return CSGPU_OK;
  cs_val *tmp = NULL;
  const size_t bytes = (size_t)count * s->n * sizeof(cs_val);
  HIP_OK(hipMalloc((void **)&tmp, bytes));
  hipError_t e = hipMemcpy(tmp, states, bytes, hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? csgpu_search_put(s, (const csgpu_val *)tmp, count) : fail(CSGPU_E_HIP, hipGetErrorString(e));
  (void)hipFree(tmp);
  return rc;
}

extern "C" int csgpu_search_set_restart(csgpu_search *s, int64_t iterations) {
  if (s == NULL || iterations < 0) return fail(CSGPU_E_ARG, "bad argument");
  s->restart_base = s->objective == CS_OBJ_ANY ? iterations : 0;
  return CSGPU_OK;
}

extern "C" int csgpu_search_set_strategy(csgpu_search *s, int order, int prefer_failing) {
  if (s == NULL || order < 0 || order > 4) return fail(CSGPU_E_ARG, "bad argument");
  if (hipSetDevice(s->device) != hipSuccess) return fail(CSGPU_E_HIP, "hipSetDevice");
  if (s->top != 0 || s->st.iterations != 0) return fail(CSGPU_E_STATE, "the strategy is set before the first state is put");
  const int is_default = order == 1 && !prefer_failing;
  if (!is_default) {
    /* the level kernels and the cut of children by the parent's own set implement the default rule only; a count of
     * failures needs the emptied variable of a failing child, which the interval kernels report */
    s->fused = 0;
    s->fw = 0;
    s->holes.pool_forb = NULL;
  }
  s->order = order;
  s->prefer_failing = prefer_failing != 0;
  s->holes.order = order;
  s->holes.prio = NULL;
  if (s->prefer_failing) {
    if (s->d_prio == NULL) HIP_OK(hipMalloc((void **)&s->d_prio, sizeof(int) * (size_t)s->n));
    HIP_OK(hipMemset(s->d_prio, 0, sizeof(int) * (size_t)s->n));
    s->holes.prio = s->d_prio;
    const int k = csgpu_model_get_kernel(s->m);
    s->fail_var_known = k == 1 || k == 6 || k == 7;
  }
  if (s->burst_exec != NULL) { /* the graph holds the old rule */
    (void)hipGraphExecDestroy(s->burst_exec);
    s->burst_exec = NULL;
  }
  return CSGPU_OK;
}

extern "C" int csgpu_search_set_restart_on_improvement(csgpu_search *s, int on) {
  if (s == NULL) return fail(CSGPU_E_ARG, "bad argument");
  if (s->top != 0 || s->st.iterations != 0) return fail(CSGPU_E_STATE, "set before the first state is put");
  s->restart_on_improvement = on != 0 && (s->objective == CS_OBJ_MIN || s->objective == CS_OBJ_MAX);
  return CSGPU_OK;
}

extern "C" int csgpu_search_take(csgpu_search *s, csgpu_val *d_states, int64_t max, int64_t *count) {
  if (s == NULL || d_states == NULL || count == NULL || max < 0) return fail(CSGPU_E_ARG, "bad argument");
  if (hipSetDevice(s->device) != hipSuccess) return fail(CSGPU_E_HIP, "hipSetDevice");
  int64_t k = max < s->top ? max : s->top;
  *count = k;
  if (k == 0) return CSGPU_OK;
  HIP_OK(hipMemcpy(d_states, s->pool, (size_t)k * s->n * sizeof(cs_val), hipMemcpyDeviceToDevice));
  if (s->fused) { /* engine rows -> interval rows, in the caller's buffer */
    const int rc = csgpu_internal_step_export(s->m, d_states, 0, k, NULL);
    if (rc != CSGPU_OK) return rc;
    HIP_OK(hipDeviceSynchronize());
  }
  /* fill the hole at the bottom with the newest rows */
  const int64_t rest = s->top - k, mv = rest < k ? rest : k;
  if (mv > 0) {
    hipLaunchKernelGGL(cs_move_rows, dim3((unsigned)((mv + 3) / 4)), dim3(SB), 0, 0, (unsigned long long *)s->pool,
                       (long long)(s->top - mv), 0ll, (int)mv, s->n);
    if (s->fw > 0)
      hipLaunchKernelGGL(cs_move_rows, dim3((unsigned)((mv + 3) / 4)), dim3(SB), 0, 0, s->pool_forb,
                         (long long)(s->top - mv), 0ll, (int)mv, s->n * s->fw);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
  }
  s->top -= k;
  return CSGPU_OK;
}

extern "C" int csgpu_search_set_parents(csgpu_search *s, int64_t parents_per_iteration) {
  if (s == NULL || parents_per_iteration < 1) return fail(CSGPU_E_ARG, "bad argument");
  const int64_t limit = parents_per_iteration < s->max_parents ? parents_per_iteration : s->max_parents;
  if ((s->lender != NULL || s->borrowers > 0) && (limit > SMALL_PARENTS || limit * s->max_width > s->max_children))
    return fail(CSGPU_E_STATE, "engines that share an incumbent run device-driven iterations: at most 256 parents per iteration");
  s->parents_limit = limit;
  s->parents_max = s->parents_limit; /* an explicit setting is taken literally */
  return CSGPU_OK;
}

extern "C" int csgpu_search_set_best(csgpu_search *s, int32_t best) {
  if (s == NULL) return fail(CSGPU_E_ARG, "bad argument");
  if (hipSetDevice(s->device) != hipSuccess) return fail(CSGPU_E_HIP, "hipSetDevice");
  const int rcf = flush_accept_results(s); /* an unread incumbent of the last iteration must not be overwritten */
  if (rcf != CSGPU_OK) return rcf;
  int better = (s->objective == CS_OBJ_MIN && best < s->st.best) || (s->objective == CS_OBJ_MAX && best > s->st.best);
  if (better) {
    s->st.best = best;
    HIP_OK(hipMemcpy(s->d_best, &best, sizeof(int), hipMemcpyHostToDevice));
  }
  return CSGPU_OK;
}

extern "C" int csgpu_search_share_incumbent(csgpu_search *s, csgpu_search *with) {
  if (s == NULL || with == NULL || s->objective != with->objective || s->obj_var != with->obj_var)
    return fail(CSGPU_E_ARG, "bad argument");
  if (hipSetDevice(s->device) != hipSuccess) return fail(CSGPU_E_HIP, "hipSetDevice");
  if (s->objective != CS_OBJ_MIN && s->objective != CS_OBJ_MAX) return CSGPU_OK; /* nothing to share */
  if (!burst_applicable(s) || !burst_applicable(with))
    return fail(CSGPU_E_STATE, "a shared incumbent needs the device-driven iterations");
  const int rcf = flush_accept_results(s);
  if (rcf != CSGPU_OK) return rcf;
  /* the better of the two goes into the shared word */
  int mine = 0, theirs = 0;
  HIP_OK(hipMemcpy(&mine, s->d_best, sizeof(int), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&theirs, with->d_best, sizeof(int), hipMemcpyDeviceToHost));
  const int best = s->objective == CS_OBJ_MIN ? (mine < theirs ? mine : theirs) : (mine > theirs ? mine : theirs);
  HIP_OK(hipMemcpy(with->d_best, &best, sizeof(int), hipMemcpyHostToDevice));
  if (s == with || s->lender == with) return CSGPU_OK;
  if (s->borrowers > 0) return fail(CSGPU_E_STATE, "an engine whose incumbent word is shared by others cannot borrow one itself");
  csgpu_search *owner = with->lender != NULL ? with->lender : with; /* chains collapse onto the engine that owns the word */
  if (s->lender != NULL && --s->lender->borrowers == 0 && s->lender->free_pending) csgpu_search_free(s->lender);
  s->lender = owner;
  owner->borrowers++;
  s->d_best = owner->d_best;
  s->st.best = best;
  if (s->burst_exec != NULL) { /* the graph holds the old pointer */
    (void)hipGraphExecDestroy(s->burst_exec);
    s->burst_exec = NULL;
  }
  return CSGPU_OK;
}

/* the accept kernel's results, read one host round trip later than they were produced */
static int apply_accept_results(csgpu_search *s, unsigned long long solutions_total, int best) {
  if (s->pending_complete == 0) return CSGPU_OK;
  const int improved = (s->objective == CS_OBJ_MIN || s->objective == CS_OBJ_MAX) && solutions_total > s->st.solutions &&
                       best != s->st.best;
  s->st.solutions = solutions_total;
  if (improved) {
    /* the complete children and their truth values of that iteration are still in place */
    hipLaunchKernelGGL(cs_pick_best, dim3(1), dim3(64), 0, 0, s->d_complete_states, s->d_truth, (int)s->pending_complete,
                       s->n, s->objective, s->obj_var, best, s->d_best_solution);
    s->have_best_solution = 1;
    s->best_solution_value = best;
  }
  if (s->objective == CS_OBJ_MIN || s->objective == CS_OBJ_MAX) s->st.best = best;
  s->pending_complete = 0;
  HIP_OK(hipGetLastError());
  return CSGPU_OK;
}

static int flush_accept_results(csgpu_search *s) {
  if (s->pending_complete == 0) return CSGPU_OK;
  unsigned long long tail[C_COUNT - C_SOLUTIONS];
  HIP_OK(hipMemcpy(tail, s->d_counters + C_SOLUTIONS, sizeof tail, hipMemcpyDeviceToHost));
  return apply_accept_results(s, tail[0], (int)(unsigned)tail[C_BEST - C_SOLUTIONS]);
}

/* ALL on a model with a step kernel: the newest `parents` rows of the pool are one frontier, expanded by one launch.
 * How many: as many as the pool and the staging buffer are likely to have room for the survivors of (the recent
 * survivors per parent size the attempt; a wave that could overflow its region stops drawing parents and the
 * undrawn ones stay where they are, so a wrong guess costs time, never a state). */
static int one_iteration_fused(csgpu_search *s) {
  const int64_t reserve = (int64_t)s->n * s->max_width;
  const double spp = s->surv_per_parent < 0.25 ? 0.25 : s->surv_per_parent;
  int64_t parents = s->top;
  /* staging: expect spp survivors per parent, keep a factor of two in hand */
  const int64_t by_stage = (int64_t)((double)s->stage_rows / (2.0 * spp));
  if (parents > by_stage) parents = by_stage;
  /* pool: a depth-first walk in batches of P holds about P * spp rows per level that is still open below the
   * frontier (at most n levels); leave that much room, shrink P as the pool fills */
  const int64_t room = s->cap - s->top - reserve;
  const int64_t by_pool = room > 0 ? (int64_t)((double)room / (spp * (double)s->n)) : 0;
  if (parents > by_pool) parents = by_pool;
  if (parents > 0x3fffffff) parents = 0x3fffffff;
  if (parents < 1) parents = 1;
  /* (no worst-case test "every child of every parent survives": the staging rows handed to the launch below are
   * capped by what the pool has room for, and a wave whose region could overflow stops drawing parents) */
  if (s->top - 1 + s->max_width > s->cap) return fail(CSGPU_E_LIMIT, "state pool is full");
  csgpu_step_launch L;
  L.pool = (const csgpu_val *)s->pool;
  L.first_row = s->top - parents;
  L.parents = (int32_t)parents;
  L.stage = (csgpu_val *)s->d_child_states;
  L.stage_rows = s->stage_rows;
  /* the survivors land behind the undrawn parents: never more than the pool has room for above its top (the rows the
   * drawn parents free come on top of that) */
  if (L.stage_rows > s->cap - s->top) L.stage_rows = s->cap - s->top;
  {
    const int64_t limit = csgpu_internal_step_parents_limit(s->m, L.stage_rows);
    if (limit < 1) return fail(CSGPU_E_LIMIT, "state pool is full");
    if (parents > limit) {
      parents = limit;
      L.first_row = s->top - parents;
      L.parents = (int32_t)parents;
    }
  }
  L.fill = s->d_fill;
  L.wstat = s->d_wstat;
  L.ticket = s->d_ticket;
  L.out = s->d_step_out;
  L.solutions = s->d_solutions;
  L.stored = (uint64_t *)(s->d_counters + C_STORED);
  L.max_solutions = s->max_solutions;
  L.store_open = s->stored_seen < (uint64_t)s->max_solutions;
  HIP_OK(hipMemsetAsync(s->d_ticket, 0, 1024, 0));
  const int rc = csgpu_internal_step(s->m, &L, NULL);
  if (rc != CSGPU_OK) return rc;
  HIP_OK(hipMemcpyAsync(s->h_step_out, s->d_step_out, sizeof(uint64_t) * 8, hipMemcpyDeviceToHost, 0));
  HIP_OK(hipStreamSynchronize(0));
  const uint64_t *h = s->h_step_out;
  const int64_t consumed = (int64_t)h[0], survivors = (int64_t)h[1];
  if (consumed < 1) return fail(CSGPU_E_LIMIT, "internal: a frontier of the fused search consumed no parent");
  s->top += survivors - consumed;
  if (s->top > s->peak) s->peak = s->top;
  s->st.iterations++;
  s->st.nodes += h[2];
  s->st.cuts += h[3];
  s->st.props += h[4];
  s->st.revisions += h[5];
  s->st.solutions += h[6];
  s->stored_seen = h[7];
  s->surv_per_parent = 0.5 * s->surv_per_parent + 0.5 * ((double)survivors / (double)consumed);
  if (getenv("CSGPU_SEARCH_TRACE") != NULL)
    fprintf(stderr, "fused: parents %lld consumed %lld survivors %lld top %lld nodes %llu cuts %llu solutions %llu\n",
            (long long)parents, (long long)consumed, (long long)survivors, (long long)s->top, (unsigned long long)h[2],
            (unsigned long long)h[3], (unsigned long long)h[6]);
  return CSGPU_OK;
}

static int one_iteration(csgpu_search *s) {
  if (s->fused) return one_iteration_fused(s);
  const int n = s->n;
  int64_t parents = s->top < s->parents_limit ? s->top : s->parents_limit;
  const int64_t reserve = (int64_t)s->n * s->max_width;
  const int64_t room_limit = s->cap > reserve ? s->cap - reserve : s->cap;
  /* ALL walks the whole tree: batches as large as the child buffers allow.  How many parents that is depends on
   * how wide they branch, which is only known after cs_branch; the recent average sizes the attempt (the exact
   * count is checked below and the attempt halved if it does not fit). */
  const int adaptive = s->objective == CS_OBJ_ALL && s->parents_limit == s->max_parents;
  if (adaptive) {
    int64_t guess = (int64_t)((double)s->max_children / (s->avg_children * 1.25));
    if (guess > s->max_children / 2) guess = s->max_children / 2;
    /* and as many as the pool is likely to have room for */
    const double per_parent = s->avg_children > 1.5 ? s->avg_children - 1.0 : 0.5;
    const int64_t room = (int64_t)((double)(room_limit - s->top) / (per_parent * 1.25));
    if (guess > room) guess = room;
    if (guess > parents) parents = guess < s->top ? guess : s->top;
  }
  /* the children of p parents need at most p * max_width rows above the top - p that stay.  A nearly full pool
   * takes as many parents as are guaranteed to fit below a reserve of n_vars * max_width rows, and one parent
   * (strict depth-first, which cannot grow the pool by more than that reserve) once the reserve is reached */
  if (!adaptive || parents <= s->max_parents) {
    if (s->top - parents + parents * s->max_width > room_limit) {
      const int64_t fit = s->max_width > 1 ? (room_limit - s->top) / (s->max_width - 1) : parents;
      parents = s->top < 1 ? 0 : (fit < 1 ? 1 : (fit < parents ? fit : parents));
      if (s->top - parents + parents * s->max_width > s->cap) return fail(CSGPU_E_LIMIT, "state pool is full");
    }
  }
  long long first_row = s->top - parents;
  if (parents == 0) return CSGPU_OK;
  const int small = parents <= SMALL_PARENTS && parents <= s->max_parents;
  const int low_last = s->objective == CS_OBJ_MAX ? 0 : 1;
  const unsigned scramble =
      s->objective == CS_OBJ_ANY ? (unsigned)(s->st.iterations * 2654435761ull + 0x9e3779b9u) | 1u : 0u;
  /* the incumbent tightens "<obj>" for every child (objective.c:101-126) */
  const int sense = s->objective == CS_OBJ_MIN ? 1 : (s->objective == CS_OBJ_MAX ? 2 : 0);
  cs_val lim = cs_objective_bound(sense, cs_interval(CS_DOM_MIN, CS_DOM_MAX), s->st.best);
  int32_t obj_lo = lim.lo, obj_hi = lim.hi;
  unsigned long long skipped_now = 0; /* large path: children cut without a launch, known with the child count */
  int64_t children;          /* what the launches are sized for */
  const uint64_t *d_children; /* where the real count is, when the host does not know it yet */
  if (small) {
    /* one workgroup expands; nothing is read back before the fixpoint is launched */
    hipLaunchKernelGGL(cs_expand_small, dim3(1), dim3(1024), 0, 0, s->pool, first_row, (int)parents, n, s->d_nodes,
                       s->d_counters, low_last, scramble, s->holes);
    children = parents * s->max_width;
    d_children = (const uint64_t *)(s->d_counters + C_TOTAL_CHILDREN);
  } else {
    unsigned pb;
    for (;;) {
      HIP_OK(hipMemsetAsync(s->d_counters, 0, sizeof(unsigned long long) * C_PER_ITERATION, 0));
      /* 16, 32 or 64 lanes per parent */
      const int ppb = n <= 16 ? SB / 16 : (n <= 32 ? SB / 32 : SB / 64);
      pb = (unsigned)((parents + ppb - 1) / ppb);
      if (n <= 16)
        hipLaunchKernelGGL(cs_branch<16>, dim3(pb), dim3(SB), 0, 0, s->pool, first_row, (int)parents, n, s->d_choice,
                           s->d_block_sum, s->holes, s->d_block_skip);
      else if (n <= 32)
        hipLaunchKernelGGL(cs_branch<32>, dim3(pb), dim3(SB), 0, 0, s->pool, first_row, (int)parents, n, s->d_choice,
                           s->d_block_sum, s->holes, s->d_block_skip);
      else
        hipLaunchKernelGGL(cs_branch<64>, dim3(pb), dim3(SB), 0, 0, s->pool, first_row, (int)parents, n, s->d_choice,
                           s->d_block_sum, s->holes, s->d_block_skip);
      hipLaunchKernelGGL(cs_scan, dim3(1), dim3(1024), 0, 0, s->d_block_sum, (int)pb, s->d_child_off, s->d_counters,
                         (int)C_TOTAL_CHILDREN, (const int *)s->d_block_skip, (int)C_SKIPPED);
      /* first host read of the iteration: the number of children, and with it what the previous
       * iteration's accept left behind (solutions so far, incumbent) */
      unsigned long long head[C_COUNT - C_TOTAL_CHILDREN];
      HIP_OK(hipMemcpy(head, s->d_counters + C_TOTAL_CHILDREN, sizeof head, hipMemcpyDeviceToHost));
      children = (int64_t)head[0];
      skipped_now = head[C_SKIPPED - C_TOTAL_CHILDREN];
      int rc0 = apply_accept_results(s, head[C_SOLUTIONS - C_TOTAL_CHILDREN], (int)(unsigned)head[C_BEST - C_TOTAL_CHILDREN]);
      if (rc0 != CSGPU_OK) return rc0;
      if (adaptive) s->avg_children = 0.5 * s->avg_children + 0.5 * ((double)children / (double)parents);
      /* the exact fit: the child buffers, and the pool rows above the parents that stay */
      const int fits = children <= s->max_children &&
                       (s->top - parents + children <= room_limit || (parents == 1 && s->top - 1 + children <= s->cap));
      if (fits) break;
      if (parents == 1) return fail(CSGPU_E_LIMIT, "state pool is full");
      parents = parents / 2 > 0 ? parents / 2 : 1;
      first_row = s->top - parents;
    }
    d_children = NULL;
    if (children > s->max_children) return fail(CSGPU_E_LIMIT, "internal: more children than the batch buffers hold");
    if (n <= 16)
      hipLaunchKernelGGL(cs_emit<16>, dim3(pb), dim3(SB), 0, 0, first_row, (int)parents, (const cs_choice *)s->d_choice,
                         (const int *)s->d_child_off, s->d_nodes, low_last, scramble);
    else if (n <= 32)
      hipLaunchKernelGGL(cs_emit<32>, dim3(pb), dim3(SB), 0, 0, first_row, (int)parents, (const cs_choice *)s->d_choice,
                         (const int *)s->d_child_off, s->d_nodes, low_last, scramble);
    else
      hipLaunchKernelGGL(cs_emit<64>, dim3(pb), dim3(SB), 0, 0, first_row, (int)parents, (const cs_choice *)s->d_choice,
                         (const int *)s->d_child_off, s->d_nodes, low_last, scramble);
    /* the incumbent may just have improved */
    lim = cs_objective_bound(sense, cs_interval(CS_DOM_MIN, CS_DOM_MAX), s->st.best);
    obj_lo = lim.lo;
    obj_hi = lim.hi;
  }
  s->top -= parents;
  s->st.iterations++;
  if (children == 0) { /* (large path only) every child was cut by its parent's own set */
    s->st.nodes += skipped_now;
    s->st.cuts += skipped_now;
    return CSGPU_OK;
  }

  int rc;
  if (s->fw > 0)
    rc = csgpu_internal_propagate_fb(s->m, (const csgpu_val *)s->pool, (const uint64_t *)s->pool_forb, s->d_nodes,
                                     (csgpu_val *)s->d_child_states, (uint64_t *)s->d_child_forb, s->d_results, children,
                                     d_children, NULL);
  else
    rc = csgpu_internal_propagate_obj(s->m, (const csgpu_val *)s->pool, s->d_nodes, (csgpu_val *)s->d_child_states,
                                      s->d_results, children, d_children, obj_lo, obj_hi, NULL);
  if (rc != CSGPU_OK) return rc;
  const unsigned cb = (unsigned)((children + SB - 1) / SB);
  if (s->prefer_failing)
    hipLaunchKernelGGL(cs_prio_update, dim3(cb), dim3(SB), 0, 0, (const csgpu_result *)s->d_results, (const csgpu_node *)s->d_nodes,
                       (int)children, (const unsigned long long *)d_children, n, s->fail_var_known, s->d_prio);
  if (small) {
    hipLaunchKernelGGL(cs_classify_small, dim3(1), dim3(1024), 0, 0, s->d_results, s->d_dest, s->d_complete_list,
                       s->d_counters, (unsigned long long *)NULL);
  } else {
    hipLaunchKernelGGL(cs_classify_count, dim3(cb), dim3(SB), 0, 0, s->d_results, (int)children, s->d_block_surv,
                       s->d_block_comp, s->d_block_cuts, s->d_block_props, s->d_block_revs);
    hipLaunchKernelGGL(cs_scan_classes, dim3(1), dim3(1024), 0, 0, s->d_block_surv, s->d_block_comp, s->d_block_cuts,
                       s->d_block_props, s->d_block_revs, (int)cb, s->d_surv_off, s->d_comp_off, s->d_counters);
    hipLaunchKernelGGL(cs_classify_assign, dim3(cb), dim3(SB), 0, 0, s->d_results, (int)children, s->d_surv_off,
                       s->d_comp_off, s->d_dest, s->d_complete_list);
  }
  {
    /* about 4096 state elements per workgroup, fewer when that would leave most of the machine idle */
    int cpb = 4096 / n;
    cpb = cpb < 4 ? 4 : (cpb > SB ? SB : cpb);
    while (cpb > 4 && children / cpb < 2048) cpb >>= 1;
    hipLaunchKernelGGL(cs_scatter, dim3((unsigned)((children + cpb - 1) / cpb)), dim3(SB), 0, 0, s->d_child_states, s->d_dest,
                       s->d_counters, (long long)s->top, n, s->pool, s->d_child_forb, s->pool_forb, s->fw, cpb,
                       (const unsigned long long *)NULL);
  }
  /* the (only, for a small iteration) host read: class counts, the real number of children, and what the
   * previous iteration's accept left behind */
  unsigned long long c[C_COUNT];
  HIP_OK(hipMemcpy(c, s->d_counters, sizeof c, hipMemcpyDeviceToHost));
  if (small) {
    children = (int64_t)c[C_TOTAL_CHILDREN];
    rc = apply_accept_results(s, c[C_SOLUTIONS], (int)(unsigned)c[C_BEST]);
    if (rc != CSGPU_OK) return rc;
  }
  s->top += (int64_t)c[C_SURVIVORS];
  if (s->top > s->peak) s->peak = s->top;
  s->st.nodes += (uint64_t)children + c[C_SKIPPED];
  s->st.cuts += c[C_CUTS] + c[C_SKIPPED];
  s->st.props += c[C_PROPS];
  s->st.revisions += c[C_REVS];

  const int64_t complete = (int64_t)c[C_COMPLETE];
  if (complete > 0) {
    if (s->objective == CS_OBJ_ALL) {
      /* evaluated and accepted where they lie, through the list of their child indices.  On a pure != network
       * (the models with forbidden sets, fw > 0) a complete consistent node IS a solution: a clause between two
       * valued variables was revised when the second of them became a value and would have emptied a domain
       * (propagate_eq_false_lr, propagate.c:106-120), and the root's own valued pairs were checked by the root
       * phase -- evaluating the root (eval_wand over every clause, eval.c:233-255) can only say "true" */
      const int *truth = s->fw > 0 && !s->eval_always ? (const int *)NULL : (const int *)s->d_truth;
      if (truth != NULL) {
        rc = csgpu_internal_eval_list(s->m, (const csgpu_val *)s->d_child_states, s->d_complete_list,
                                      (const uint64_t *)(s->d_counters + C_COMPLETE), complete, s->d_truth, NULL);
        if (rc != CSGPU_OK) return rc;
      }
      hipLaunchKernelGGL(cs_accept, dim3((unsigned)((complete + SB - 1) / SB)), dim3(SB), 0, 0, s->d_child_states,
                         truth, (int)complete, n, s->objective, s->obj_var, s->d_counters, s->d_solutions,
                         (long long)s->max_solutions, (const int *)s->d_complete_list);
    } else {
      /* MIN / MAX keep the complete children of the iteration together: cs_pick_best looks at them one host
       * round trip later */
      const unsigned gw = (unsigned)((complete + 3) / 4);
      hipLaunchKernelGGL(cs_gather_complete, dim3(gw), dim3(SB), 0, 0, s->d_child_states, s->d_complete_list,
                         (int)complete, n, s->d_complete_states);
      rc = csgpu_eval_batch(s->m, (const csgpu_val *)s->d_complete_states, s->d_truth, complete, NULL);
      if (rc != CSGPU_OK) return rc;
      hipLaunchKernelGGL(cs_accept, dim3((unsigned)((complete + SB - 1) / SB)), dim3(SB), 0, 0, s->d_complete_states,
                         s->d_truth, (int)complete, n, s->objective, s->obj_var, s->d_counters, s->d_solutions,
                         (long long)s->max_solutions, (const int *)NULL);
    }
    /* what accept found is read together with the next iteration's child count (or at the end of the
     * run); ANY stops on the first solution, so it looks at once */
    s->pending_complete = complete;
    if (s->objective == CS_OBJ_ANY) {
      rc = flush_accept_results(s);
      if (rc != CSGPU_OK) return rc;
    }
  }
  HIP_OK(hipGetLastError());
  return CSGPU_OK;
}

/* ---- device-driven iterations (ANY / MIN / MAX) ----
 * These searches expand a few parents per iteration (depth first towards a solution / a better incumbent), so an
 * iteration is a handful of small launches and, driven from the host, mostly the round trip for its counts.
 * Here BURST_ITERATIONS iterations are enqueued at once -- as one hipGraph, built once -- with everything the host
 * would decide in between (how many parents, where the survivors go, the incumbent, whether to stop) decided by
 * single-workgroup kernels from state in device memory; the host reads the totals once per burst. */
static int burst_applicable(const csgpu_search *s) {
  const int64_t most = s->burst_split && s->objective != CS_OBJ_ANY ? BURST_PARENTS_MAX : SMALL_PARENTS;
  return !s->burst_off && s->objective != CS_OBJ_ALL && s->parents_max <= most &&
         s->parents_max * s->max_width <= s->max_children;
}

static int enqueue_burst(csgpu_search *s, hipStream_t st) {
  const int n = s->n;
  const int64_t bound = s->parents_max * s->max_width; /* children of one iteration at most */
  const int64_t reserve = (int64_t)s->n * s->max_width;
  const long long room_limit = s->cap > reserve ? s->cap - reserve : s->cap;
  const uint64_t *d_children = (const uint64_t *)(s->d_counters + C_TOTAL_CHILDREN);
  const int sense = s->objective == CS_OBJ_MIN ? 1 : (s->objective == CS_OBJ_MAX ? 2 : 0);
  int cpb = 4096 / n;
  cpb = cpb < 4 ? 4 : (cpb > SB ? SB : cpb);
  while (cpb > 4 && bound / cpb < 2048) cpb >>= 1;
  /* ANY dives with few parents and must see the accept before it decides: one workgroup */
  const int split = s->burst_split && s->objective != CS_OBJ_ANY;
  const unsigned burst_wgs = (unsigned)((s->parents_max + BURST_PPW - 1) / BURST_PPW); /* <= BURST_WGS_MAX: burst_applicable */
  /* Without expression-tree clauses a complete consistent child IS a solution, and evaluating the root (eval_wand over
   * every clause, eval.c:233-255) can only say "true": every clause is a binary relation or a two-literal disjunction
   * whose revision on valued operands fails exactly when it is violated, and each was revised after the last of its
   * variables became a value (in this node or in the ancestor that valued it; the root's own valued clauses by the root
   * phase).  The launch that would say so is left out (CSGPU_SEARCH_EVAL=1 keeps it: tests compare the two). */
  const int *truth = s->burst_no_eval ? (const int *)NULL : (const int *)s->d_truth;
  for (int it = 0; it < BURST_ITERATIONS; it++) {
    if (split) {
      hipLaunchKernelGGL(cs_burst_branch, dim3(burst_wgs), dim3(1024), 0, st, s->pool, n, s->d_counters, s->d_burst,
                         s->objective, (long long)s->max_width, (long long)s->cap, room_limit, s->holes, s->d_choice,
                         s->d_block_sum, s->d_block_skip, (const cs_val *)s->d_child_states,
                         (const int *)s->d_complete_list, truth, s->obj_var, s->d_solutions,
                         (long long)s->max_solutions, s->d_best_solution, s->d_best);
      hipLaunchKernelGGL(cs_burst_emit, dim3(burst_wgs), dim3(1024), 0, st, s->d_nodes, s->d_counters, s->d_burst,
                         s->objective, (const cs_choice *)s->d_choice, (const int *)s->d_block_sum,
                         (const int *)s->d_block_skip);
    } else
      hipLaunchKernelGGL(cs_expand_burst, dim3(1), dim3(1024), 0, st, s->pool, n, s->d_nodes, s->d_counters, s->d_burst,
                         s->objective, (long long)s->max_width, (long long)s->cap, room_limit, s->holes,
                         (const cs_val *)s->d_child_states, (const int *)s->d_complete_list, truth,
                         s->obj_var, s->d_solutions, (long long)s->max_solutions, s->d_best_solution, s->d_best);
    int rc;
    if (s->fw > 0)
      rc = csgpu_internal_propagate_fb(s->m, (const csgpu_val *)s->pool, (const uint64_t *)s->pool_forb, s->d_nodes,
                                       (csgpu_val *)s->d_child_states, (uint64_t *)s->d_child_forb, s->d_results, bound,
                                       d_children, st);
    else
      rc = csgpu_internal_propagate_objdev(s->m, (const csgpu_val *)s->pool, s->d_nodes, (csgpu_val *)s->d_child_states,
                                           s->d_results, bound, d_children, CS_DOM_MIN, CS_DOM_MAX,
                                           sense ? (const int32_t *)s->d_best : NULL, sense, st);
    if (rc != CSGPU_OK) return rc;
    if (s->prefer_failing)
      hipLaunchKernelGGL(cs_prio_update, dim3((unsigned)((bound + SB - 1) / SB)), dim3(SB), 0, st, (const csgpu_result *)s->d_results,
                         (const csgpu_node *)s->d_nodes, (int)bound, (const unsigned long long *)d_children, n, s->fail_var_known,
                         s->d_prio);
    if (split) {
      hipLaunchKernelGGL(cs_burst_count, dim3(BURST_CLASS_WGS), dim3(1024), 0, st, (const csgpu_result *)s->d_results,
                         (const unsigned long long *)s->d_counters, s->d_block_surv, s->d_block_comp, s->d_block_cuts,
                         s->d_block_props, s->d_block_revs);
      hipLaunchKernelGGL(cs_burst_assign, dim3(BURST_CLASS_WGS), dim3(1024), 0, st, (const csgpu_result *)s->d_results,
                         s->d_dest, s->d_complete_list, s->d_counters, s->d_burst, (const int *)s->d_block_surv,
                         (const int *)s->d_block_comp, (const int *)s->d_block_cuts, (const int *)s->d_block_props,
                         (const int *)s->d_block_revs, (const cs_val *)s->d_child_states, s->pool, n,
                         (const unsigned long long *)s->d_child_forb, s->pool_forb, s->fw);
    } else {
      hipLaunchKernelGGL(cs_classify_small, dim3(1), dim3(1024), 0, st, s->d_results, s->d_dest, s->d_complete_list,
                         s->d_counters, s->d_burst);
      hipLaunchKernelGGL(cs_scatter, dim3((unsigned)((bound + cpb - 1) / cpb)), dim3(SB), 0, st, s->d_child_states, s->d_dest,
                         s->d_counters, 0ll, n, s->pool, s->d_child_forb, s->pool_forb, s->fw, cpb,
                         (const unsigned long long *)(s->d_burst + B_SCATTER_BASE));
    }
    if (truth != NULL) {
      rc = csgpu_internal_eval_list(s->m, (const csgpu_val *)s->d_child_states, s->d_complete_list,
                                    (const uint64_t *)(s->d_counters + C_COMPLETE), bound, s->d_truth, st);
      if (rc != CSGPU_OK) return rc;
    }
  }
  /* the last iteration's accept (the others ran at the head of the following expansion) */
  hipLaunchKernelGGL(cs_accept_burst, dim3(1), dim3(256), 0, st, s->d_child_states, s->d_complete_list, truth, n,
                     s->objective, s->obj_var, s->d_counters, s->d_burst, s->d_solutions, (long long)s->max_solutions,
                     s->d_best_solution, s->d_best);
  HIP_OK(hipGetLastError());
  return CSGPU_OK;
}

/* up to `budget` iterations; *done = how many had parents */
static int run_burst(csgpu_search *s, int64_t budget, int64_t *done) {
  int rc = flush_accept_results(s);
  if (rc != CSGPU_OK) return rc;
  if (!s->graph_off && (s->burst_exec == NULL || s->burst_limit != s->parents_max)) {
    if (s->burst_exec != NULL) {
      (void)hipGraphExecDestroy(s->burst_exec);
      s->burst_exec = NULL;
    }
    hipGraph_t graph = NULL;
    if (hipStreamBeginCapture(s->burst_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      rc = enqueue_burst(s, s->burst_stream);
      const hipError_t e = hipStreamEndCapture(s->burst_stream, &graph);
      if (rc != CSGPU_OK) {
        if (graph != NULL) (void)hipGraphDestroy(graph);
        return rc;
      }
      if (e == hipSuccess && graph != NULL && hipGraphInstantiate(&s->burst_exec, graph, NULL, NULL, 0) != hipSuccess)
        s->burst_exec = NULL;
      if (graph != NULL) (void)hipGraphDestroy(graph);
    }
    (void)hipGetLastError();
    s->burst_limit = s->parents_max;
  }
  unsigned long long *h = s->h_burst;
  memset(h, 0, sizeof(unsigned long long) * B_COUNT);
  h[B_TOP] = (unsigned long long)s->top;
  h[B_BUDGET] = (unsigned long long)(budget < BURST_ITERATIONS ? budget : BURST_ITERATIONS);
  h[B_LIMIT] = (unsigned long long)s->parents_limit;
  h[B_LIMIT_MAX] = (unsigned long long)s->parents_max;
  h[B_BACKLOG_DIV] = 4ull; /* a quarter of the pool per iteration, within [B_LIMIT, B_LIMIT_MAX] (was a sixteenth: see csgpu_search_create) */
  { const char *e = getenv("CSGPU_SEARCH_BACKLOG_DIV"); if (e != NULL && atoi(e) >= 1) h[B_BACKLOG_DIV] = (unsigned long long)atoi(e); } /* tuning */
  h[B_ITER_BASE] = (unsigned long long)s->st.iterations;
  h[B_PEAK] = (unsigned long long)s->peak;
  HIP_OK(hipMemcpyAsync(s->d_burst, h, sizeof(unsigned long long) * B_COUNT, hipMemcpyHostToDevice, s->burst_stream));
  if (s->burst_exec != NULL) {
    HIP_OK(hipGraphLaunch(s->burst_exec, s->burst_stream));
  } else {
    rc = enqueue_burst(s, s->burst_stream);
    if (rc != CSGPU_OK) return rc;
  }
  HIP_OK(hipMemcpyAsync(h, s->d_burst, sizeof(unsigned long long) * B_COUNT, hipMemcpyDeviceToHost, s->burst_stream));
  HIP_OK(hipMemcpyAsync(h + B_COUNT, s->d_counters, sizeof(unsigned long long) * C_COUNT, hipMemcpyDeviceToHost,
                        s->burst_stream));
  HIP_OK(hipMemcpyAsync(h + B_COUNT + C_COUNT, s->d_best, sizeof(int), hipMemcpyDeviceToHost, s->burst_stream));
  HIP_OK(hipStreamSynchronize(s->burst_stream));
  if (getenv("CSGPU_SEARCH_TRACE") != NULL)
    fprintf(stderr, "burst: iters %llu top %llu nodes %llu cuts %llu | surv %llu complete %llu children %llu solutions %llu stored %llu best %d\n",
            h[B_ITERS], h[B_TOP], h[B_NODES], h[B_CUTS], h[B_COUNT + C_SURVIVORS], h[B_COUNT + C_COMPLETE],
            h[B_COUNT + C_TOTAL_CHILDREN], h[B_COUNT + C_SOLUTIONS], h[B_COUNT + C_STORED], (int)(unsigned)h[B_COUNT + C_BEST]);
  if (h[B_ERROR] != 0ull) return fail(CSGPU_E_LIMIT, "state pool is full");
  *done = (int64_t)h[B_ITERS];
  s->top = (int64_t)h[B_TOP];
  s->peak = (int64_t)h[B_PEAK];
  s->st.iterations += h[B_ITERS];
  s->st.nodes += h[B_NODES];
  s->st.cuts += h[B_CUTS];
  s->st.props += h[B_PROPS];
  s->st.revisions += h[B_REVS];
  s->st.solutions = h[B_COUNT + C_SOLUTIONS];
  if (s->objective == CS_OBJ_MIN || s->objective == CS_OBJ_MAX) s->st.best = *(const int *)(h + B_COUNT + C_COUNT);
  if (h[B_IMPROVED] != 0ull) {
    s->have_best_solution = 1;
    s->best_solution_value = (int32_t)(uint32_t)h[B_IMPROVED];
  }
  return CSGPU_OK;
}

extern "C" int csgpu_search_run(csgpu_search *s, int64_t max_iterations, csgpu_search_stats *stats) {
  if (s == NULL || stats == NULL) return fail(CSGPU_E_ARG, "bad argument");
  if (hipSetDevice(s->device) != hipSuccess) return fail(CSGPU_E_HIP, "hipSetDevice");
  for (int64_t it = 0; it < max_iterations; it++) {
    if (s->top == 0) break;
    if (s->objective == CS_OBJ_ANY && s->st.solutions > 0) break;
    int rc;
    int64_t steps = 1; /* iterations this pass of the loop made */
    const int32_t best_before = s->st.best;
    const int restarts_on = s->restart_base > 0 && s->seed_count > 0 && s->st.solutions == 0;
    if (burst_applicable(s)) {
      int64_t budget = max_iterations - it;
      if (restarts_on) { /* stop where check_restart would fire */
        const int64_t until = (int64_t)s->luby_threshold * s->restart_base + 1 - s->since_restart;
        if (until < budget) budget = until < 1 ? 1 : until;
      }
      rc = run_burst(s, budget, &steps);
      if (rc != CSGPU_OK) return rc;
      if (steps == 0) break; /* nothing left to expand (or ANY solved) */
      it += steps - 1;
    } else {
      rc = one_iteration(s);
      if (rc != CSGPU_OK) return rc;
    }
    /* a better solution restarts a MIN / MAX search from its seeds with the new bound (update_solution +
     * is_solution_restartable, csolve.c:216-219, 418-425) */
    if (s->restart_on_improvement && s->seed_count > 0 && s->top > 0) {
      const int rcf = flush_accept_results(s);
      if (rcf != CSGPU_OK) return rcf;
      if (s->st.best != best_before) {
        s->st.restarts++;
        s->top = 0;
        const int keep_flag = s->restart_on_improvement;
        s->restart_on_improvement = 0; /* do not record the re-seeding as new seeds */
        const int64_t keep = s->restart_base;
        s->restart_base = 0;
        rc = csgpu_search_put(s, (const csgpu_val *)s->seed, s->seed_count);
        s->restart_base = keep;
        s->restart_on_improvement = keep_flag;
        if (rc != CSGPU_OK) return rc;
      }
    }
    /* check_restart (csolve.c:264-276) with Knuth's Luby sequence (csolve.c:76-83) */
    if (restarts_on && s->st.solutions == 0 &&
        (s->since_restart += steps) > (int64_t)s->luby_threshold * s->restart_base) {
      s->since_restart = 0;
      cs_luby_next(&s->luby_threshold, &s->luby_counter);
      s->st.restarts++;
      s->top = 0;
      const int64_t keep = s->restart_base; /* do not record the re-seeding as new seeds */
      s->restart_base = 0;
      rc = csgpu_search_put(s, (const csgpu_val *)s->seed, s->seed_count);
      s->restart_base = keep;
      if (rc != CSGPU_OK) return rc;
    }
  }
  {
    const int rcf = flush_accept_results(s);
    if (rcf != CSGPU_OK) return rcf;
  }
  s->st.pool = s->top;
  s->st.pool_peak = s->peak;
  s->st.done = s->top == 0 || (s->objective == CS_OBJ_ANY && s->st.solutions > 0);
  *stats = s->st;
  return CSGPU_OK;
}

extern "C" int64_t csgpu_search_solutions(const csgpu_search *s, int32_t *values, int64_t max) {
  if (s == NULL || values == NULL || max < 0) return CSGPU_E_ARG;
  if (hipSetDevice(s->device) != hipSuccess) return fail(CSGPU_E_HIP, "hipSetDevice");
  unsigned long long stored = 0;
  if (hipMemcpy(&stored, s->d_counters + C_STORED, sizeof stored, hipMemcpyDeviceToHost) != hipSuccess) return CSGPU_E_HIP;
  int64_t k = (int64_t)stored;
  if (k > s->max_solutions) k = s->max_solutions;
  if (k > max) k = max;
  if (k > 0 && hipMemcpy(values, s->d_solutions, (size_t)k * s->n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
    return CSGPU_E_HIP;
  return k;
}

extern "C" int csgpu_search_best_solution(const csgpu_search *s, int32_t *values) {
  if (s == NULL || values == NULL) return CSGPU_E_ARG;
  if (hipSetDevice(s->device) != hipSuccess) return fail(CSGPU_E_HIP, "hipSetDevice");
  /* with a shared incumbent another engine may hold the row that attains it */
  if (!s->have_best_solution || s->best_solution_value != s->st.best) return 0;
  if (hipMemcpy(values, s->d_best_solution, (size_t)s->n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
    return CSGPU_E_HIP;
  return 1;
}
