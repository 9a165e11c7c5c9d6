/* cs_capi.hip -- implementation of the C ABI declared in include/csolve_gpu.h.
 * Host orchestration only: parsing / indexing / table building happen in the C files
 * next to this one, every interval computation happens in the kernels of
 * cs_kernels.hip.h.  Nothing here evaluates or narrows a domain on the CPU. */
#include <hip/hip_runtime.h>
#include <chrono>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/csolve_gpu.h"
#include "cs_kernels.hip.h"
#include "cs_shave.hip.h"
#include "cs_step.hip.h"
#include "cs_chain.hip.h"
#include "cs_internal.h"

static thread_local char g_err[512] = "";

static int set_err(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

/* shared with cs_search.hip */
extern "C" int csgpu_internal_set_error(int code, const char *msg) { return set_err(code, "%s", msg); }

#define HIP_TRY(expr)                                                          \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess)                                                      \
      return set_err(CSGPU_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));     \
  } while (0)

#define CS_TICKET_SLOTS 1024
#define CS_TICKET_STREAMS 64
#define CS_TICKET_SLOT_WORDS (CS_SHAVE_SHARDS * CS_SHAVE_TICKET_STRIDE)

struct csgpu_model {
  cs_model *host;
  int from_dump;     /* clause lists came with the file: keep them */
  int not_root;      /* host domains are not the root state: no entailed-clause elimination */
  int finalized;
  cs_dev_image *img; /* search image */
  /* device copies of the image */
  int *d_adj_off, *d_adj, *d_clause, *d_tree_off, *d_tnode, *d_tkid, *d_tree_want, *d_lit, *d_clause_by_kind;
  cs_tables tab;
  size_t slice;      /* LDS bytes per node instance (16-byte aligned) */
  int has_tree_adj;  /* some adjacency entry is a tree clause */
  int packed_bias;   /* kernel 5: largest root_lo - dense_dmin */
  int packed_nw;     /* kernel 5: set words per variable it works on (1: all root domains within 32 values) */
  int kernel_choice; /* 0 auto, 1 general, 2 LDS-resident, 3 forbidden sets, 4 forbidden sets in registers, 5 = 4 with
                        several nodes per wave */
  void *d_adj_packed;
  int lds_waves;     /* waves per workgroup of the LDS-resident kernel, 0 = not eligible */
  size_t lds_bytes;  /* its dynamic LDS size */
  int lds_adj_global; /* kernel 2 reads its adjacency through L2 (the lists would leave LDS for fewer than 16 waves) */
  size_t k1_tab_bytes; /* general kernel: adj_off + adj + lit copied into LDS by every workgroup (0: read through L2) */
  int fb_words;      /* forbidden-set words per variable (0 = not eligible) */
  int fb_waves;
  size_t fb_bytes;
  int dense_waves;    /* register-resident forbidden-set kernel: waves per workgroup, 0 = not eligible */
  size_t dense_bytes; /* its LDS table */
  void *d_dense_tab;
  void *d_packed_tab; /* kernel 5: 16-bit table relative to the pushing variable */
  int *d_root_lo;
  int *d_sym_off;
  void *d_sym_packed;
  int n_cus;
  /* kernel 7: ticket counters of its work distribution (cs_shave.hip.h).  One slot (CS_SHAVE_SHARDS counters on
   * their own 64-byte lines) per stream that launches it -- launches of one stream are ordered, and the kernel
   * leaves its counters at zero -- and one slot of its own for every launch that is being captured into a
   * hipGraph (graphs captured on one stream may be replayed on different ones at the same time). */
  unsigned *d_tickets;
  int ticket_slots, ticket_streams, ticket_reserved;
  void *ticket_stream[CS_TICKET_STREAMS];
  /* staging of csgpu_propagate_one */
  /* pinned host memory mapped into the device's address space: the kernel reads the state and the node record
   * from it and writes the result and the new state back, no staging copies */
  unsigned char *h_one;          /* [state | node | result | state_out] */
  /* csgpu_propagate_one_traced: clause of every adjacency entry (device), the log and its counter (mapped host
   * memory), allocated at the first traced call */
  int *d_adj_clause;
  int32_t *h_trace;
  unsigned *h_trace_n;
  int trace_cap;
  cs_val *d_one_in, *d_one_out;  /* device views of the four parts */
  cs_node_in *d_one_node;
  cs_node_out *d_one_res;
  int engines; /* csgpu_search objects built on this model (they hold pointers into its device tables) */
  /* csgpu_propagate_one_chain (cs_chain.hip.h): clause shapes and lists on the device, the frame stack, the mapped
   * host buffers; chain_state: 0 not built yet, 1 ready, -1 the model does not qualify */
  int chain_state;
  cs_chain_clause *d_chain_cl;
  int *d_chain_off, *d_chain_list;
  cs_chain_frame *d_chain_frames;
  int *h_chain_out; /* mapped: [4] out, then the bumps */
  /* the resident single-node server (cs_shave_server): its mailbox in coherent host memory, its stream, the last
   * request number; srv_off: not used for this model (it does not qualify, or CSGPU_SERVER=0) */
  unsigned char *h_box;   /* [cs_mailbox_head | state_in | state_out | trace] */
  size_t box_state_off, box_out_off, box_trace_off;
  hipStream_t srv_stream;
  unsigned srv_seq;
  int srv_off, srv_launched;
  double srv_seconds[4]; /* host copy in, ring + wait, copy out, (re)starts -- csgpu_debug_one_timing */
  uint64_t srv_calls, srv_starts;
  /* staging of csgpu_propagate_values (mapped pinned memory, grown on demand) */
  unsigned char *h_values, *d_values;
  size_t values_cap;
};

extern "C" const char *csgpu_last_error(void) { return g_err; }

/* Wait for the null stream by polling.  The single-node entries (the drop-in's calls) are a launch of a few
 * microseconds followed by a wait: hipStreamSynchronize in a process that has not asked for spinning blocks in the
 * kernel driver and is woken 50-100 us later (the reference's driver linked on this library measured 75 us per call
 * where a python loop, whose runtime spins, measured 19 us for the same calls). */
static int wait_null_stream(void) {
  for (;;) {
    const hipError_t e = hipStreamQuery(NULL);
    if (e == hipSuccess) return CSGPU_OK;
    if (e != hipErrorNotReady) return set_err(CSGPU_E_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
  }
}

extern "C" int csgpu_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return set_err(CSGPU_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

extern "C" int csgpu_set_device(int device) {
  HIP_TRY(hipSetDevice(device));
  return CSGPU_OK;
}

/* ---- helpers of the search driver (cs_arith.h), exported so that the reference's unit vectors run against the
 * very functions the kernels, the engine and the drop-in use ---------------------------------------------------- */

static int objective_sense(int objective) { return objective == CS_OBJ_MIN ? 1 : (objective == CS_OBJ_MAX ? 2 : 0); }

extern "C" int csgpu_objective_better(int objective, csgpu_val value, int32_t best) {
  return cs_objective_better(objective_sense(objective), cs_interval(value.lo, value.hi), best);
}
extern "C" csgpu_val csgpu_objective_bound(int objective, csgpu_val value, int32_t best) {
  const cs_val v = cs_objective_bound(objective_sense(objective), cs_interval(value.lo, value.hi), best);
  csgpu_val out;
  out.lo = v.lo;
  out.hi = v.hi;
  return out;
}
extern "C" int32_t csgpu_objective_best(int objective, csgpu_val value, int32_t best) {
  return cs_objective_best(objective_sense(objective), cs_interval(value.lo, value.hi), best);
}
extern "C" void csgpu_luby_next(uint64_t *threshold, uint64_t *counter) {
  if (threshold != NULL && counter != NULL) cs_luby_next(threshold, counter);
}
extern "C" int csgpu_step_check(csgpu_val bounds, uint32_t iter) {
  return cs_step_check(cs_interval(bounds.lo, bounds.hi), iter);
}
extern "C" uint64_t csgpu_branch_key(int order, int prefer_failing, csgpu_val value, int64_t prio, int32_t index) {
  return cs_branch_key_of(order, prefer_failing, cs_interval(value.lo, value.hi), prio, index);
}
extern "C" int32_t csgpu_step_val(csgpu_val bounds, uint32_t iter, uint32_t seed) {
  return cs_step_val(cs_interval(bounds.lo, bounds.hi), iter, seed);
}

/* ---- host model ------------------------------------------------------------------ */

static int wrap_model(cs_model *host, int from_dump, csgpu_model **out) {
  csgpu_model *m = (csgpu_model *)calloc(1, sizeof *m);
  if (m == NULL) { cs_model_free(host); return set_err(CSGPU_E_ARG, "out of memory"); }
  m->host = host;
  m->from_dump = from_dump;
  *out = m;
  return CSGPU_OK;
}

extern "C" int csgpu_model_from_text(const char *text, int weights_on, csgpu_model **out) {
  if (text == NULL || out == NULL) return set_err(CSGPU_E_ARG, "null argument");
  char err[256];
  cs_model *host = cs_model_parse(text, weights_on, err, sizeof err);
  if (host == NULL) return set_err(CSGPU_E_PARSE, "%s", err);
  return wrap_model(host, 0, out);
}

extern "C" int csgpu_model_from_file(const char *path, int weights_on, csgpu_model **out) {
  if (path == NULL || out == NULL) return set_err(CSGPU_E_ARG, "null argument");
  FILE *f = fopen(path, "rb");
  if (f == NULL) return set_err(CSGPU_E_ARG, "%s: cannot open", path);
  fseek(f, 0, SEEK_END);
  long len = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *text = (char *)malloc((size_t)len + 1);
  size_t got = fread(text, 1, (size_t)len, f);
  fclose(f);
  text[got] = '\0';
  int rc = csgpu_model_from_text(text, weights_on, out);
  free(text);
  return rc;
}

extern "C" int csgpu_model_from_dump(const char *path, csgpu_model **out) {
  if (path == NULL || out == NULL) return set_err(CSGPU_E_ARG, "null argument");
  char err[256];
  cs_model *host = cs_model_load(path, err, sizeof err);
  if (host == NULL) return set_err(CSGPU_E_ARG, "%s", err);
  if (host->clause_node == NULL) {
    cs_model_free(host);
    return set_err(CSGPU_E_ARG, "%s: model file has no clause index", path);
  }
  return wrap_model(host, 1, out);
}

/* internal (cs_internal.h): wrap a host model built by other host code of this package
 * (the drop-in shim); takes ownership.  lists_final: keep the clause index it carries. */
extern "C" int csgpu_model_from_host(cs_model *host, int lists_final, int domains_are_root, csgpu_model **out) {
  if (host == NULL || out == NULL) return set_err(CSGPU_E_ARG, "null argument");
  int rc = wrap_model(host, lists_final, out);
  if (rc == CSGPU_OK) (*out)->not_root = !domains_are_root;
  return rc;
}

extern "C" cs_model *csgpu_model_host(csgpu_model *m) { return m ? m->host : NULL; }

static void server_stop(csgpu_model *m);
static void server_reap(csgpu_model *m);
/* Models whose resident server may be running.  hipFree, hipDeviceSynchronize and friends wait for EVERY stream of the
 * device, the server's included -- they would sit out its idle time-out (2 ms) each.  Entry points that allocate, free
 * or synchronise device-wide therefore ask the servers of this process to leave first (a flag in the mailbox: the wave
 * is gone within microseconds); the next single-node call starts one again. */
static csgpu_model *g_srv_models[16];
static void quiesce_servers(void) {
  for (int i = 0; i < 16; i++)
    if (g_srv_models[i] != NULL) server_stop(g_srv_models[i]);
}

static void free_device(csgpu_model *m) {
  (void)hipFree(m->d_chain_cl); (void)hipFree(m->d_chain_off); (void)hipFree(m->d_chain_list); (void)hipFree(m->d_chain_frames);
  if (m->h_chain_out != NULL) (void)hipHostFree(m->h_chain_out);
  m->d_chain_cl = NULL; m->d_chain_off = NULL; m->d_chain_list = NULL; m->d_chain_frames = NULL; m->h_chain_out = NULL;
  m->chain_state = 0;
  quiesce_servers(); /* before anything is freed: this model's resident wave reads the tables, and hipFree waits for all of them */
  for (int i = 0; i < 16; i++)
    if (g_srv_models[i] == m) g_srv_models[i] = NULL;
  if (m->h_box != NULL) (void)hipHostFree(m->h_box);
  m->h_box = NULL;
  if (m->srv_stream != NULL) (void)hipStreamDestroy(m->srv_stream);
  m->srv_stream = NULL;
  m->srv_off = 0;
  (void)hipFree(m->d_adj_off); (void)hipFree(m->d_adj); (void)hipFree(m->d_clause); (void)hipFree(m->d_clause_by_kind);
  (void)hipFree(m->d_tree_off); (void)hipFree(m->d_tnode); (void)hipFree(m->d_tkid); (void)hipFree(m->d_tree_want);
  (void)hipFree(m->d_lit);
  m->d_tree_want = NULL;
  m->d_lit = NULL;
  (void)hipFree(m->d_adj_packed);
  (void)hipFree(m->d_root_lo);
  (void)hipFree(m->d_sym_off);
  (void)hipFree(m->d_sym_packed);
  (void)hipFree(m->d_dense_tab);
  (void)hipFree(m->d_packed_tab);
  m->d_packed_tab = NULL;
  (void)hipFree(m->d_tickets);
  m->d_tickets = NULL;
  m->ticket_slots = m->ticket_streams = m->ticket_reserved = 0;
  m->d_sym_off = NULL;
  m->d_sym_packed = NULL;
  m->d_dense_tab = NULL;
  m->dense_waves = 0;
  m->d_adj_packed = NULL;
  m->d_root_lo = NULL;
  m->lds_waves = 0;
  m->fb_words = 0;
  (void)hipHostFree(m->h_one);
  m->h_one = NULL;
  (void)hipFree(m->d_adj_clause);
  m->d_adj_clause = NULL;
  if (m->h_trace != NULL) (void)hipHostFree(m->h_trace);
  m->h_trace = NULL;
  m->h_trace_n = NULL;
  m->trace_cap = 0;
  if (m->h_values != NULL) (void)hipHostFree(m->h_values);
  m->h_values = m->d_values = NULL;
  m->values_cap = 0;
  m->d_adj_off = m->d_adj = m->d_clause = m->d_tree_off = m->d_tnode = m->d_tkid = m->d_clause_by_kind = NULL;
  m->d_one_in = m->d_one_out = NULL;
  m->d_one_node = NULL;
  m->d_one_res = NULL;
  cs_dev_image_free(m->img);
  m->img = NULL;
}

extern "C" void csgpu_model_free(csgpu_model *m) {
  if (m == NULL) return;
  free_device(m);
  cs_model_free(m->host);
  free(m);
}

extern "C" int csgpu_model_num_vars(const csgpu_model *m) { return m ? m->host->n_vars : CSGPU_E_ARG; }
extern "C" int csgpu_model_num_clauses(const csgpu_model *m) {
  if (m == NULL) return CSGPU_E_ARG;
  /* the clause index is built on demand (it only depends on the trees and the domains) */
  if (m->host->clause_node == NULL && m->host->root >= 0 && cs_model_index(m->host) != 0) return CSGPU_E_ARG;
  return m->host->n_clauses;
}
extern "C" int csgpu_model_objective(const csgpu_model *m) { return m ? m->host->objective : CSGPU_E_ARG; }
extern "C" int csgpu_model_objective_var(const csgpu_model *m) { return m ? m->host->obj_var : CSGPU_E_ARG; }

extern "C" const char *csgpu_model_var_name(const csgpu_model *m, int var) {
  if (m == NULL || var < 0 || var >= m->host->n_vars) return NULL;
  return m->host->names[var];
}

extern "C" int csgpu_model_get_domains(const csgpu_model *m, csgpu_val *out) {
  if (m == NULL || out == NULL) return set_err(CSGPU_E_ARG, "null argument");
  memcpy(out, m->host->dom, (size_t)m->host->n_vars * sizeof(cs_val));
  return CSGPU_OK;
}

extern "C" int csgpu_model_set_domains(csgpu_model *m, const csgpu_val *in) {
  if (m == NULL || in == NULL) return set_err(CSGPU_E_ARG, "null argument");
  memcpy(m->host->dom, in, (size_t)m->host->n_vars * sizeof(cs_val));
  return CSGPU_OK;
}

extern "C" void csgpu_set_linear_fast_paths(int on) { cs_dev_linear_fast_paths = on != 0; }

extern "C" int csgpu_model_device_info(const csgpu_model *m, int64_t info[8]) {
  if (m == NULL || info == NULL) return set_err(CSGPU_E_ARG, "null argument");
  if (m->img == NULL) return set_err(CSGPU_E_STATE, "device tables are not built");
  info[0] = m->img->n_adj;
  info[1] = m->img->n_ne;
  info[2] = m->img->n_tree_clauses;
  info[3] = m->img->n_tnodes;
  info[4] = (int64_t)m->slice;
  info[5] = m->img->max_list;
  info[6] = m->img->n_skip;
  info[7] = m->img->max_tree;
  return CSGPU_OK;
}

/* ---- device image ------------------------------------------------------------------ */

struct dev_tables_owner {
  int *adj_off, *adj, *clause, *tree_off, *tnode, *tkid, *tree_want, *lit, *clause_by_kind;
};

static int upload(const void *src, size_t bytes, int **dst) {
  HIP_TRY(hipMalloc((void **)dst, bytes ? bytes : 16));
  if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return CSGPU_OK;
}

static int upload_image(const cs_dev_image *g, dev_tables_owner *o, cs_tables *t) {
  int rc;
  memset(o, 0, sizeof *o);
  if ((rc = upload(g->adj_off, ((size_t)g->n_vars + 1) * 4, &o->adj_off))) return rc;
  if ((rc = upload(g->adj, (size_t)(g->n_adj ? g->n_adj : 1) * 8, &o->adj))) return rc;
  if ((rc = upload(g->clause, (size_t)(g->n_clauses ? g->n_clauses : 1) * 16, &o->clause))) return rc;
  {
    /* the clause records once more, sorted by kind -- =, <, !=, then disjunctions, trees, constants -- for the
     * kernel that gives every lane its own clauses (kernel 6): a round revises ALL clauses whatever their order, and 64
     * lanes of one kind run one path instead of all of them one after the other */
    const int32_t nc = g->n_clauses;
    int32_t *sorted = (int32_t *)malloc((size_t)(nc ? nc : 1) * 16);
    if (sorted == NULL) return set_err(CSGPU_E_LIMIT, "out of memory");
    static const int order[] = { CS_CL_EQ, CS_CL_LT, CS_CL_NE, CS_CL_OR2, CS_CL_TREE, CS_CL_SKIP };
    int32_t k = 0;
    for (size_t q = 0; q < sizeof order / sizeof order[0]; q++)
      for (int32_t c = 0; c < nc; c++)
        if (g->clause[4 * c] == order[q]) { memcpy(sorted + 4 * k, g->clause + 4 * c, 16); k++; }
    rc = k == nc ? upload(sorted, (size_t)(nc ? nc : 1) * 16, &o->clause_by_kind) : set_err(CSGPU_E_STATE, "clause of unknown kind");
    free(sorted);
    if (rc) return rc;
  }
  if ((rc = upload(g->tree_off, ((size_t)g->n_trees + 1) * 4, &o->tree_off))) return rc;
  if ((rc = upload(g->tnode, (size_t)(g->n_tnodes ? g->n_tnodes : 1) * 16, &o->tnode))) return rc;
  if ((rc = upload(g->tkid, (size_t)(g->n_tkids ? g->n_tkids : 1) * 4, &o->tkid))) return rc;
  if ((rc = upload(g->tree_want, (size_t)(g->n_trees ? g->n_trees : 1) * 8, &o->tree_want))) return rc;
  if ((rc = upload(g->lit, (size_t)(g->n_lits ? g->n_lits : 1) * 16, &o->lit))) return rc;
  t->n_vars = g->n_vars;
  t->n_clauses = g->n_clauses;
  t->n_words = (g->n_vars + 31) / 32;
  t->adj_off = o->adj_off;
  t->adj = (const int2 *)o->adj;
  t->clause = (const int4 *)o->clause;
  t->clause_by_kind = (const int4 *)o->clause_by_kind;
  t->tree_off = o->tree_off;
  t->tnode = (const int4 *)o->tnode;
  t->tkid = o->tkid;
  t->tree_want = (const int2 *)o->tree_want;
  t->lit = (const int4 *)o->lit;
  t->n_lits = g->n_lits;
  t->obj_var = -1;
  t->obj_lo = CS_DOM_MIN;
  t->obj_hi = CS_DOM_MAX;
  t->obj_best_dev = NULL;
  t->obj_sense = 0;
  return CSGPU_OK;
}

static void free_tables(dev_tables_owner *o) {
  (void)hipFree(o->adj_off); (void)hipFree(o->adj); (void)hipFree(o->clause); (void)hipFree(o->clause_by_kind);
  (void)hipFree(o->tree_off); (void)hipFree(o->tnode); (void)hipFree(o->tkid); (void)hipFree(o->tree_want);
  (void)hipFree(o->lit);
}

static int lds_limit(size_t bytes, const void *func) {
  if (bytes > 160u * 1024u) return set_err(CSGPU_E_LIMIT, "%zu bytes of LDS per workgroup exceed the 160 KiB of a CU", bytes);
  if (bytes > 48u * 1024u)
    HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return CSGPU_OK;
}

/* ---- root phase -------------------------------------------------------------------- */

extern "C" int csgpu_model_root_propagate(csgpu_model *m, int32_t *status) {
  return csgpu_model_root_propagate_limit(m, -1, status, NULL);
}

extern "C" int csgpu_model_root_propagate_limit(csgpu_model *m, int64_t limit, int32_t *status, int32_t *rounds) {
  if (m == NULL || status == NULL) return set_err(CSGPU_E_ARG, "null argument");
  quiesce_servers();
  cs_model *h = m->host;
  if (h->root < 0) return set_err(CSGPU_E_STATE, "model has no root");
  if (!m->from_dump && cs_model_index(h) != 0) return set_err(CSGPU_E_ARG, "%s", h->err);
  char err[200];
  cs_dev_image *g = cs_dev_image_build(h, 0, NULL, err, sizeof err);
  if (g == NULL) return set_err(CSGPU_E_LIMIT, "%s", err);

  dev_tables_owner own;
  cs_tables tab;
  int rc = upload_image(g, &own, &tab);
  cs_val *d_in = NULL, *d_out = NULL;
  cs_node_out *d_res = NULL;
  int *d_conv = NULL, conv = 1;
  cs_node_out res;
  const size_t nbytes = (size_t)(h->n_vars ? h->n_vars : 1) * sizeof(cs_val);
  const size_t lds = (size_t)h->n_vars * sizeof(cs_val) + 4 * sizeof(unsigned);
  hipError_t e = hipSuccess;
  if (rc == CSGPU_OK) rc = lds_limit(lds, (const void *)cs_propagate_sweeps);
  if (rc == CSGPU_OK) {
    if ((e = hipMalloc((void **)&d_in, nbytes)) != hipSuccess || (e = hipMalloc((void **)&d_out, nbytes)) != hipSuccess ||
        (e = hipMalloc((void **)&d_res, sizeof res)) != hipSuccess || (e = hipMalloc((void **)&d_conv, sizeof(int))) != hipSuccess ||
        (e = hipMemcpy(d_in, h->dom, (size_t)h->n_vars * sizeof(cs_val), hipMemcpyHostToDevice)) != hipSuccess)
      rc = set_err(CSGPU_E_HIP, "root propagate setup: %s", hipGetErrorString(e));
  }
  if (rc == CSGPU_OK) {
    /* propagate(root, limit) stops after limit + 1 sweeps (propagate.c:479-483) */
    const int max_rounds = limit < 0 || limit >= 0x7ffffffe ? 0x7fffffff : (int)limit + 1;
    /* All clauses of a round in parallel first: a round never narrows more than a sweep of the reference, so a
     * fixpoint (or an inconsistency) reached within the limit is the reference's.  Only when the limit cuts the
     * iteration short do the states depend on the order: then the same number of sweeps is run again from the
     * start in the reference's own order, one clause after the other (propagate.c:379-392, 474-485). */
    hipLaunchKernelGGL(cs_propagate_sweeps, dim3(1), dim3(CS_BLOCK), lds, 0, tab, d_in, d_out, d_res, max_rounds, 0, d_conv);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess ||
        (e = hipMemcpy(&res, d_res, sizeof res, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(&conv, d_conv, sizeof conv, hipMemcpyDeviceToHost)) != hipSuccess)
      rc = set_err(CSGPU_E_HIP, "root propagate: %s", hipGetErrorString(e));
    if (rc == CSGPU_OK && res.status >= 0 && !conv) {
      hipLaunchKernelGGL(cs_propagate_sweeps, dim3(1), dim3(CS_BLOCK), lds, 0, tab, d_in, d_out, d_res, max_rounds, 1, d_conv);
      if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess ||
          (e = hipMemcpy(&res, d_res, sizeof res, hipMemcpyDeviceToHost)) != hipSuccess)
        rc = set_err(CSGPU_E_HIP, "root propagate (sequential sweeps): %s", hipGetErrorString(e));
    }
    if (rc == CSGPU_OK && (e = hipMemcpy(h->dom, d_out, (size_t)h->n_vars * sizeof(cs_val), hipMemcpyDeviceToHost)) != hipSuccess)
      rc = set_err(CSGPU_E_HIP, "root propagate: %s", hipGetErrorString(e));
    if (rc != CSGPU_OK) { /* reported below */ }
    else {
      *status = res.status < 0 ? -1 : res.props;
      if (rounds != NULL) *rounds = res.rounds;
    }
  }
  (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_res); (void)hipFree(d_conv);
  free_tables(&own);
  cs_dev_image_free(g);
  return rc;
}

extern "C" int csgpu_model_normalize(csgpu_model *m) {
  if (m == NULL) return set_err(CSGPU_E_ARG, "null argument");
  if (m->from_dump) return set_err(CSGPU_E_STATE, "a dumped model is already normalised");
  if (m->finalized) return set_err(CSGPU_E_STATE, "model is already finalized");
  if (cs_model_normalize(m->host) < 0) return set_err(CSGPU_E_STATE, "model has no root");
  return CSGPU_OK;
}

/* the model rewritten for the subtree below `state` -- a copy whose root domains are `state`, ready for
 * csgpu_model_normalize (which folds what the prefix has decided: normalize.c:67-75 replaces every subtree that now
 * evaluates to a value by that value, neutral elements and satisfied clauses drop out) and csgpu_model_finalize (which
 * leaves clauses that are true on the whole of `state` out of the device tables).  The reference does this inside the
 * search, clause by clause (normalize + patch at the end of propagate_clauses, propagate.c:521-535, undone on
 * backtracking); here it is a host-side pre-pass per subtree, as the survey puts it. */
extern "C" int csgpu_model_specialize(const csgpu_model *m, const csgpu_val *state, csgpu_model **out) {
  if (m == NULL || state == NULL || out == NULL) return set_err(CSGPU_E_ARG, "null argument");
  const cs_model *h = m->host;
  if (h->root < 0) return set_err(CSGPU_E_STATE, "model has no root");
  for (int32_t v = 0; v < h->n_vars; v++) {
    if (state[v].lo > state[v].hi) return set_err(CSGPU_E_ARG, "variable %d: empty interval", v);
    if (state[v].lo < h->dom[v].lo || state[v].hi > h->dom[v].hi)
      return set_err(CSGPU_E_ARG, "variable %d: [%d, %d] is not inside the model's [%d, %d]", v, state[v].lo, state[v].hi,
                     h->dom[v].lo, h->dom[v].hi);
  }
  cs_model *copy = cs_model_clone(h);
  if (copy == NULL) return set_err(CSGPU_E_LIMIT, "out of memory");
  memcpy(copy->dom, state, (size_t)h->n_vars * sizeof(cs_val));
  return wrap_model(copy, 0, out);
}

extern "C" int csgpu_model_add_conflict(csgpu_model *m, int32_t count, const int32_t *vars, const int32_t *values) {
  if (m == NULL || count < 1 || vars == NULL || values == NULL) return set_err(CSGPU_E_ARG, "bad argument");
  cs_model *h = m->host;
  if (h->root < 0) return set_err(CSGPU_E_STATE, "model has no root");
  if (count + 1 > CS_MAX_TREE_NODES) return set_err(CSGPU_E_LIMIT, "a conflict of %d elements, device limit is %d", count, CS_MAX_TREE_NODES - 1);
  /* a finalized model is finalized again below, which frees every device table: a search engine built on the model
   * (its captured hipGraphs, its kernels' arguments) would go on using the freed pointers */
  if (m->finalized && m->engines > 0)
    return set_err(CSGPU_E_STATE, "%d search engine(s) hold the model's device tables: free them before adding a clause", m->engines);
  int32_t *terms = (int32_t *)malloc((size_t)count * sizeof(int32_t));
  if (terms == NULL) return set_err(CSGPU_E_LIMIT, "out of memory");
  for (int32_t i = 0; i < count; i++) {
    if (vars[i] < 0 || vars[i] >= h->n_vars) {
      free(terms);
      return set_err(CSGPU_E_ARG, "conflict element %d: no such variable", i);
    }
    terms[i] = h->var_node[vars[i]];
  }
  const int32_t node = cs_model_add_confl(h, terms, values, count);
  free(terms);
  if (cs_model_append_clause(h, node) != 0) return set_err(CSGPU_E_ARG, "conflict clause could not be indexed");
  /* the device tables are immutable images: a finalized model is finalized again (O(model) per clause; learnt
   * clauses arrive once per failed node of a host-driven search, which costs a launch and a round trip anyway) */
  return m->finalized ? csgpu_model_finalize(m) : CSGPU_OK;
}

extern "C" int csgpu_model_eval_clauses_host(csgpu_model *m, csgpu_val *vals) {
  if (m == NULL || vals == NULL) return set_err(CSGPU_E_ARG, "null argument");
  quiesce_servers();
  cs_model *h = m->host;
  if (h->root < 0) return set_err(CSGPU_E_STATE, "model has no root");
  if (!m->from_dump && cs_model_index(h) != 0) return set_err(CSGPU_E_ARG, "%s", h->err);
  if (h->n_clauses == 0) return CSGPU_OK;
  char err[200];
  cs_dev_image *g = cs_dev_image_build(h, 0, NULL, err, sizeof err);
  if (g == NULL) return set_err(CSGPU_E_LIMIT, "%s", err);
  dev_tables_owner own;
  cs_tables tab;
  int rc = upload_image(g, &own, &tab);
  cs_val *d_state = NULL, *d_vals = NULL;
  const size_t lds = (size_t)h->n_vars * sizeof(cs_val) + 16;
  hipError_t e = hipSuccess;
  if (rc == CSGPU_OK) rc = lds_limit(lds, (const void *)cs_eval_clauses);
  if (rc == CSGPU_OK &&
      ((e = hipMalloc((void **)&d_state, (size_t)(h->n_vars ? h->n_vars : 1) * sizeof(cs_val))) != hipSuccess ||
       (e = hipMalloc((void **)&d_vals, (size_t)h->n_clauses * sizeof(cs_val))) != hipSuccess ||
       (e = hipMemcpy(d_state, h->dom, (size_t)h->n_vars * sizeof(cs_val), hipMemcpyHostToDevice)) != hipSuccess))
    rc = set_err(CSGPU_E_HIP, "eval: %s", hipGetErrorString(e));
  if (rc == CSGPU_OK) {
    unsigned blocks = (unsigned)((h->n_clauses + CS_BLOCK - 1) / CS_BLOCK);
    if (blocks > 1024u) blocks = 1024u;
    hipLaunchKernelGGL(cs_eval_clauses, dim3(blocks), dim3(CS_BLOCK), lds, 0, tab, d_state, d_vals);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess ||
        (e = hipMemcpy(vals, d_vals, (size_t)h->n_clauses * sizeof(cs_val), hipMemcpyDeviceToHost)) != hipSuccess)
      rc = set_err(CSGPU_E_HIP, "eval: %s", hipGetErrorString(e));
  }
  (void)hipFree(d_state); (void)hipFree(d_vals);
  free_tables(&own);
  cs_dev_image_free(g);
  return rc;
}

/* instantiation of the LDS-resident kernel for an entry width and a prefetch depth
 * R = ceil(n_vars / 64) rounded up to a power of two (at most 16: larger states load their tail in place) */
static const void *ne_lds_kernel(int width, int n_vars, int adj_global) {
  /* the largest instantiated depth that is at most ceil(n_vars / 64): all its strides but the last are full (the kernel
   * relies on that), a longer state loads its tail in place */
  const int strides = (n_vars + CS_WAVE - 1) / CS_WAVE;
  int r = strides >= 16 ? 16 : (strides >= 8 ? 8 : (strides >= 4 ? 4 : (strides >= 2 ? 2 : 1)));
  /* 577 to 1023 variables (a 25x25 sudoku has 625): ten strides in registers */
  if (strides >= 10 && strides < 16 && !adj_global)
    return width == 2 ? (const void *)cs_propagate_ne_lds<unsigned short, 10, 1, true>
                      : (const void *)cs_propagate_ne_lds<unsigned int, 10, 1, true>;
#define CS_PICK_U(E, RR) return adj_global ? (const void *)cs_propagate_ne_lds<E, RR, 1, false> : (const void *)cs_propagate_ne_lds<E, RR, 1, true>;
#define CS_PICK(E)                                                                                 \
  switch (r) {                                                                                     \
  case 1: CS_PICK_U(E, 1)                                                                          \
  case 2: CS_PICK_U(E, 2)                                                                          \
  case 4: CS_PICK_U(E, 4)                                                                          \
  case 8: CS_PICK_U(E, 8)                                                                          \
  default: CS_PICK_U(E, 16)                                                                        \
  }
  if (width == 2) { CS_PICK(unsigned short) }
  CS_PICK(unsigned int)
#undef CS_PICK
#undef CS_PICK_U
}

static const void *ne_bitset_kernel(int width, int fw, int n_vars) {
  /* prefetch depth R = ceil(n_vars/64) when that is 1, 2 or 4 (states of up to 256 variables), or 10 with
   * one set word per variable (up to 640 variables: a 25x25 sudoku's next state travels in 40 VGPRs) */
  const int chunks = (n_vars + CS_WAVE - 1) / CS_WAVE;
  int r = chunks <= 1 ? 1 : (chunks <= 2 ? 2 : (chunks <= 4 ? 4 : (chunks <= 10 && fw == 1 ? 10 : 0)));
#define CS_PICK_R(E, F)                                                                            \
  switch (r) {                                                                                     \
  case 1: return (const void *)cs_propagate_ne_bitset<E, F, 1>;                                    \
  case 2: return (const void *)cs_propagate_ne_bitset<E, F, 2>;                                    \
  case 4: return (const void *)cs_propagate_ne_bitset<E, F, 4>;                                    \
  case 10: return (const void *)cs_propagate_ne_bitset<E, 1, 10>;                                  \
  default: return (const void *)cs_propagate_ne_bitset<E, F, 0>;                                   \
  }
#define CS_PICK(E)                                                                                 \
  switch (fw) {                                                                                    \
  case 1: CS_PICK_R(E, 1)                                                                          \
  case 2: CS_PICK_R(E, 2)                                                                          \
  default: CS_PICK_R(E, 4)                                                                         \
  }
  if (width == 2) { CS_PICK(unsigned short) }
  CS_PICK(unsigned int)
#undef CS_PICK
#undef CS_PICK_R
}

/* register-resident forbidden-set kernel: entry width, set words, variables per lane; the FAST
 * instantiation (n_vars a multiple of 64 that fills the lanes, both set buffers) keeps D nodes in flight */
static const void *ne_regs_kernel(int width, int fw, int n_vars, int fast, int sets_only) {
  const int chunks = (n_vars + CS_WAVE - 1) / CS_WAVE;
  const int r = chunks <= 1 ? 1 : (chunks <= 2 ? 2 : 4);
#define CS_PICK_D(E, F, RR)                                                                        \
  if (sets_only)                                                                                   \
    return fast ? (const void *)cs_propagate_ne_regs<E, F, RR, 2, true, true>                       \
                : (const void *)cs_propagate_ne_regs<E, F, RR, 1, false, true>;                     \
  return fast ? (const void *)cs_propagate_ne_regs<E, F, RR, 2, true, false>                        \
              : (const void *)cs_propagate_ne_regs<E, F, RR, 1, false, false>;
#define CS_PICK_R(E, F)                                                                            \
  switch (r) {                                                                                     \
  case 1: CS_PICK_D(E, F, 1)                                                                       \
  case 2: CS_PICK_D(E, F, 2)                                                                       \
  default: CS_PICK_D(E, F, 4)                                                                      \
  }
#define CS_PICK(E)                                                                                 \
  switch (fw) {                                                                                    \
  case 1: CS_PICK_R(E, 1)                                                                          \
  case 2: CS_PICK_R(E, 2)                                                                          \
  default: CS_PICK_R(E, 4)                                                                         \
  }
  if (width == 1) { CS_PICK(unsigned char) }
  CS_PICK(unsigned short)
#undef CS_PICK
#undef CS_PICK_R
#undef CS_PICK_D
}

/* kernel 7 (cs_shave.hip.h): interval states only; entry width, variables per lane, slots per pair known at
 * compile time when 1 or 3, FULL when the variables fill the lanes */
static const void *ne_shave_kernel(int width, int n_vars, int slots, int full) {
  const int chunks = (n_vars + CS_WAVE - 1) / CS_WAVE;
  const int r = chunks <= 1 ? 1 : (chunks <= 2 ? 2 : 4);
  const int sl = slots == 1 ? 1 : (slots == 3 ? 3 : 0);
#define CS_PICK_F(E, RR, SS)                                                                       \
  return full ? (const void *)cs_propagate_ne_shave<E, RR, SS, true>                                \
              : (const void *)cs_propagate_ne_shave<E, RR, SS, false>;
#define CS_PICK_S(E, RR)                                                                           \
  switch (sl) {                                                                                    \
  case 1: CS_PICK_F(E, RR, 1)                                                                      \
  case 3: CS_PICK_F(E, RR, 3)                                                                      \
  default: CS_PICK_F(E, RR, 0)                                                                     \
  }
#define CS_PICK(E)                                                                                 \
  switch (r) {                                                                                     \
  case 1: CS_PICK_S(E, 1)                                                                          \
  case 2: CS_PICK_S(E, 2)                                                                          \
  default: CS_PICK_S(E, 4)                                                                         \
  }
  if (width == 1) { CS_PICK(unsigned char) }
  CS_PICK(unsigned short)
#undef CS_PICK
#undef CS_PICK_S
#undef CS_PICK_F
}

/* its tracing variant for single nodes (guarded lanes; the slot count a constant when 1 or 3: with the run-time loop
 * the six LDS reads of a row operation are waited for one by one, which is the latency of a single wavefront) */
static const void *ne_shave_trace_kernel(int width, int n_vars, int slots) {
  const int chunks = (n_vars + CS_WAVE - 1) / CS_WAVE;
  const int r = chunks <= 1 ? 1 : (chunks <= 2 ? 2 : 4);
  const int sl = slots == 1 ? 1 : (slots == 3 ? 3 : 0);
#define CS_PICK_S(E, RR)                                                                           \
  switch (sl) {                                                                                    \
  case 1: return (const void *)cs_propagate_ne_shave<E, RR, 1, false, true>;                        \
  case 3: return (const void *)cs_propagate_ne_shave<E, RR, 3, false, true>;                        \
  default: return (const void *)cs_propagate_ne_shave<E, RR, 0, false, true>;                       \
  }
#define CS_PICK(E)                                                                                 \
  switch (r) {                                                                                     \
  case 1: CS_PICK_S(E, 1)                                                                          \
  case 2: CS_PICK_S(E, 2)                                                                          \
  default: CS_PICK_S(E, 4)                                                                         \
  }
  if (width == 1) { CS_PICK(unsigned char) }
  CS_PICK(unsigned short)
#undef CS_PICK
#undef CS_PICK_S
}

/* kernel 5: kernel 4 for small models, 64 / n_vars (2 or 4) nodes per wave; needs the 8-bit dense table (its
 * 16-bit LDS copy is relative to the pushing variable, cs_kernels.hip.h) */
static int packed_nodes_per_wave(int fw, int n_vars, int dense_width) {
  return fw == 1 && dense_width == 1 && n_vars <= 32 ? (n_vars <= 16 ? 4 : 2) : 0;
}

static const void *ne_packed_kernel(int n_vars, int nw, int s3) {
#define CS_PICK_S(G, NW)                                                                           \
  return s3 ? (const void *)cs_propagate_ne_packed<G, NW, true> : (const void *)cs_propagate_ne_packed<G, NW, false>;
#define CS_PICK_NW(G)                                                                              \
  if (nw == 1) { CS_PICK_S(G, 1) }                                                                 \
  CS_PICK_S(G, 2)
  if (n_vars <= 16) { CS_PICK_NW(4) }
  CS_PICK_NW(2)
#undef CS_PICK_NW
#undef CS_PICK_S
}

static const void *step_packed_kernel(int n_vars, int nw, int s3); /* below, with the step launcher */
static const void *step_import_kernel(int n_vars, int nw, int s3);
static const void *step_shave_kernel(int width, int n_vars, int slots, int full);

/* kernel 6: clauses per lane (1, 2, 4, 8) if the model has at most 512 clauses, else 0 */
static int clause_rounds_cpl(const csgpu_model *m) {
  if (m->img == NULL || m->img->n_clauses <= 0 || m->img->n_clauses > 8 * CS_WAVE) return 0;
  const int per = (m->img->n_clauses + CS_WAVE - 1) / CS_WAVE;
  return per <= 1 ? 1 : (per <= 2 ? 2 : (per <= 4 ? 4 : 8));
}

/* ---- finalize ---------------------------------------------------------------------- */

extern "C" int csgpu_model_build_tables(csgpu_model *m) {
  if (m == NULL) return set_err(CSGPU_E_ARG, "null argument");
  cs_model *h = m->host;
  int32_t ub = cs_model_first_unbounded(h);
  if (ub >= 0) return set_err(CSGPU_E_UNBOUNDED, "unbounded variable: %s", h->names[ub]);
  if (!m->from_dump && cs_model_index(h) != 0) return set_err(CSGPU_E_ARG, "%s", h->err);
  cs_dev_image_free(m->img);
  m->img = NULL;
  m->finalized = 0;
  char err[200];
  m->img = cs_dev_image_build(h, 1, NULL, err, sizeof err);
  if (m->img == NULL) return set_err(CSGPU_E_LIMIT, "%s", err);
  const size_t slice = (size_t)h->n_vars * sizeof(cs_val) + 2 * (size_t)((h->n_vars + 31) / 32) * sizeof(unsigned);
  m->slice = (slice + 15) & ~(size_t)15;
  return CSGPU_OK;
}

extern "C" int csgpu_model_finalize(csgpu_model *m) {
  if (m == NULL) return set_err(CSGPU_E_ARG, "null argument");
  cs_model *h = m->host;
  free_device(m);
  int rc0 = csgpu_model_build_tables(m);
  if (rc0 != CSGPU_OK) return rc0;

  dev_tables_owner own;
  int rc = upload_image(m->img, &own, &m->tab);
  if (rc == CSGPU_OK && h->n_clauses > 0 && !m->not_root) {
    /* Clauses that already evaluate to true in the root state are entailed for the whole
     * search: evaluate every clause once on the device and drop those from the tables. */
    cs_val *d_state = NULL, *d_vals = NULL;
    const size_t lds0 = (size_t)h->n_vars * sizeof(cs_val) + 16;
    cs_val *vals = (cs_val *)malloc((size_t)h->n_clauses * sizeof(cs_val));
    unsigned char *entailed = (unsigned char *)calloc((size_t)h->n_clauses, 1);
    hipError_t e = hipSuccess;
    rc = lds_limit(lds0, (const void *)cs_eval_clauses);
    if (rc == CSGPU_OK &&
        ((e = hipMalloc((void **)&d_state, (size_t)(h->n_vars ? h->n_vars : 1) * sizeof(cs_val))) != hipSuccess ||
         (e = hipMalloc((void **)&d_vals, (size_t)h->n_clauses * sizeof(cs_val))) != hipSuccess ||
         (e = hipMemcpy(d_state, h->dom, (size_t)h->n_vars * sizeof(cs_val), hipMemcpyHostToDevice)) != hipSuccess))
      rc = set_err(CSGPU_E_HIP, "finalize: %s", hipGetErrorString(e));
    if (rc == CSGPU_OK) {
      unsigned blocks = (unsigned)((h->n_clauses + CS_BLOCK - 1) / CS_BLOCK);
      if (blocks > 1024u) blocks = 1024u;
      hipLaunchKernelGGL(cs_eval_clauses, dim3(blocks), dim3(CS_BLOCK), lds0, 0, m->tab, d_state, d_vals);
      if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess ||
          (e = hipMemcpy(vals, d_vals, (size_t)h->n_clauses * sizeof(cs_val), hipMemcpyDeviceToHost)) != hipSuccess)
        rc = set_err(CSGPU_E_HIP, "finalize: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_state); (void)hipFree(d_vals);
    if (rc == CSGPU_OK) {
      int any = 0;
      for (int32_t c = 0; c < h->n_clauses; c++)
        if (m->img->clause[4 * c] != CS_CL_SKIP && cs_is_true(vals[c])) { entailed[c] = 1; any = 1; }
      if (any) {
        free_tables(&own);
        memset(&own, 0, sizeof own);
        cs_dev_image_free(m->img);
        char err2[200];
        m->img = cs_dev_image_build(h, 1, entailed, err2, sizeof err2);
        if (m->img == NULL) rc = set_err(CSGPU_E_LIMIT, "%s", err2);
        else rc = upload_image(m->img, &own, &m->tab);
      }
    }
    free(vals); free(entailed);
  }
  m->d_adj_off = own.adj_off; m->d_adj = own.adj; m->d_clause = own.clause; m->d_clause_by_kind = own.clause_by_kind;
  m->d_tree_off = own.tree_off; m->d_tnode = own.tnode; m->d_tkid = own.tkid; m->d_tree_want = own.tree_want;
  m->d_lit = own.lit;
  if (rc != CSGPU_OK) return rc;

  const size_t slice = (size_t)h->n_vars * sizeof(cs_val) + 2 * (size_t)m->tab.n_words * sizeof(unsigned);
  m->slice = (slice + 15) & ~(size_t)15;
  m->has_tree_adj = 0;
  for (int32_t i = 0; i < m->img->n_adj; i++)
    if (m->img->adj[2 * i] < 0 && m->img->adj[2 * i + 1] == 0) { m->has_tree_adj = 1; break; }
  /* small tables travel into LDS with every workgroup of the general kernel */
  {
    const size_t tb = ((((size_t)h->n_vars + 1) * 4 + 15) & ~(size_t)15) + (((size_t)m->img->n_adj * 8 + 15) & ~(size_t)15) +
                      (((size_t)m->img->n_lits * 16 + 15) & ~(size_t)15);
    m->k1_tab_bytes = (tb <= 32u * 1024u && m->img->n_adj > 0) ? tb : 0;
  }
  const size_t lds = m->slice * CS_WAVES_PER_BLOCK + m->k1_tab_bytes;
  if ((rc = lds_limit(lds, (const void *)cs_propagate_events<false, false>))) return rc;
  if ((rc = lds_limit(lds, (const void *)cs_propagate_events<true, false>))) return rc;
  if ((rc = lds_limit(lds, (const void *)cs_propagate_events<false, true>))) return rc;
  if ((rc = lds_limit(lds, (const void *)cs_propagate_events<true, true>))) return rc;
  if ((rc = lds_limit((size_t)h->n_vars * sizeof(cs_val) + 16, (const void *)cs_eval_root))) return rc;
  if ((rc = lds_limit((size_t)h->n_vars * sizeof(cs_val) + 16, (const void *)cs_eval_clauses))) return rc;

  /* LDS-resident kernel: adj_off + packed adjacency + one slice per wave must fit in a CU's LDS */
  m->lds_waves = 0;
  if (m->img->packed_width != 0) {
    const size_t off_bytes = (((size_t)h->n_vars * 2 * sizeof(int)) + 15) & ~(size_t)15;
    const size_t adj_bytes = (((size_t)m->img->n_adj * (size_t)m->img->packed_width) + 15) & ~(size_t)15;
    const size_t lds_slice = ((size_t)h->n_vars * sizeof(cs_val) + (2 * (size_t)m->tab.n_words + 1) * sizeof(unsigned) + 15) & ~(size_t)15;
    m->lds_adj_global = 0;
    /* as many waves as fit next to the lists (not only 16, 8 or 4: the 25x25 sudoku's lists leave room for fifteen
     * slices, and with eight its nodes' chains of dependent steps had two waves per SIMD to hide behind) */
    int waves_max = 16;
    { const char *e = getenv("CSGPU_K2_WAVES"); if (e != NULL && atoi(e) >= 4 && atoi(e) <= 16) waves_max = atoi(e); } /* measurement override */
    for (int waves = waves_max; waves >= 4; waves--) {
      const size_t need = off_bytes + adj_bytes + (size_t)waves * lds_slice;
      if (need <= 160u * 1024u) {
        m->lds_waves = waves;
        m->lds_bytes = need;
        break;
      }
    }
    const char *force = getenv("CSGPU_K2_ADJ"); /* "global": measurement override */
    const int want_global = force != NULL && force[0] == 'g';
    if (want_global && 8 * lds_slice <= 160u * 1024u) {
      /* the lists in device memory (L2-resident) and LDS for the node slices only: 24 waves per CU instead of 16 on
       * the 25x25 sudoku network -- and 1.23 ms instead of 0.75 ms for 2^18 nodes: the two-byte gathers through L2
       * cost more than the extra waves hide.  Kept for measurements; never chosen automatically. */
      m->lds_adj_global = 1;
      m->lds_waves = 8;
      m->lds_bytes = 8 * lds_slice;
    }
    if (m->lds_waves) {
      if ((rc = upload(m->img->adj_packed, (size_t)m->img->n_adj * (size_t)m->img->packed_width,
                       (int **)&m->d_adj_packed)))
        return rc;
      if ((rc = lds_limit(m->lds_bytes, ne_lds_kernel(m->img->packed_width, h->n_vars, m->lds_adj_global)))) return rc;
    }
  }

  /* forbidden-set kernel: additionally root intervals of at most 256 values, and only when the
   * host domains really are the root state (the bit windows are anchored at the root bounds) */
  m->fb_words = 0;
  if (m->img->sym_width != 0 && !m->not_root) {
    int64_t width = 1;
    for (int32_t v = 0; v < h->n_vars; v++) {
      const int64_t w = (int64_t)h->dom[v].hi - (int64_t)h->dom[v].lo + 1;
      if (w > width) width = w;
    }
    const int fw = width <= 64 ? 1 : (width <= 128 ? 2 : (width <= 256 ? 4 : 0));
    if (fw) {
      const size_t off_bytes = (((size_t)h->n_vars * 2 * sizeof(int)) + 15) & ~(size_t)15;
      const size_t base_bytes = (((size_t)h->n_vars * sizeof(int)) + 15) & ~(size_t)15;
      const size_t adj_bytes = (((size_t)m->img->sym_n_adj * (size_t)m->img->sym_width) + 15) & ~(size_t)15;
      const size_t sl = ((size_t)h->n_vars * sizeof(cs_val) + (size_t)h->n_vars * fw * 8 +
                         (2 * (size_t)m->tab.n_words + 2) * sizeof(unsigned) + 15) & ~(size_t)15;
      for (int waves = 16; waves >= 1; waves--) { /* as many waves as fit next to the tables */
        const size_t need = off_bytes + base_bytes + adj_bytes + (size_t)waves * sl;
        if (need <= 160u * 1024u) {
          m->fb_words = fw;
          m->fb_waves = waves;
          m->fb_bytes = need;
          break;
        }
      }
    }
    if (m->fb_words) {
      int *lo = (int *)malloc((size_t)(h->n_vars ? h->n_vars : 1) * sizeof(int));
      for (int32_t v = 0; v < h->n_vars; v++) lo[v] = h->dom[v].lo;
      rc = upload(lo, (size_t)h->n_vars * sizeof(int), &m->d_root_lo);
      free(lo);
      if (rc) return rc;
      if ((rc = upload(m->img->sym_off, ((size_t)h->n_vars + 1) * sizeof(int), &m->d_sym_off))) return rc;
      if ((rc = upload(m->img->sym_packed, (size_t)m->img->sym_n_adj * (size_t)m->img->sym_width,
                       (int **)&m->d_sym_packed)))
        return rc;
      if ((rc = lds_limit(m->fb_bytes, ne_bitset_kernel(m->img->sym_width, m->fb_words, m->host->n_vars)))) return rc;
      /* the register-resident variant: the dense table must fit a CU's LDS; the workgroup size that keeps
       * the most waves resident (32 per CU at most), smaller workgroups on ties */
      if (m->img->dense_width != 0) {
        const size_t bytes = (size_t)h->n_vars * m->img->dense_slots * m->img->dense_cols * m->img->dense_width;
        int best = 0, best_waves = 0;
        for (int waves = 4; waves <= 16; waves <<= 1) {
          size_t wgs = (160u * 1024u) / bytes;
          if (wgs > (size_t)(32 / waves)) wgs = (size_t)(32 / waves);
          if ((int)wgs * waves > best) { best = (int)wgs * waves; best_waves = waves; }
        }
        if (best_waves) {
          m->dense_waves = best_waves;
          m->dense_bytes = bytes;
          if ((rc = upload(m->img->dense_tab, bytes, (int **)&m->d_dense_tab))) return rc;
          for (int variant = 0; variant < 4; variant++)
            if ((rc = lds_limit(bytes, ne_regs_kernel(m->img->dense_width, m->fb_words, h->n_vars, variant & 1, variant >> 1))))
              return rc;
          for (int full = 0; full < 2; full++)
            if ((rc = lds_limit(bytes, ne_shave_kernel(m->img->dense_width, h->n_vars, m->img->dense_slots, full)))) return rc;
          if (((bytes + 15) & ~(size_t)15) + (size_t)best_waves * 32 <= 160u * 1024u)
            for (int full = 0; full < 2; full++)
              if ((rc = lds_limit(((bytes + 15) & ~(size_t)15) + (size_t)best_waves * 32,
                                  step_shave_kernel(m->img->dense_width, h->n_vars, m->img->dense_slots, full))))
                return rc;
          if (((bytes + 15) & ~(size_t)15) + CS_SHAVE_TRACE_LDS * 16 <= 160u * 1024u &&
              (rc = lds_limit(((bytes + 15) & ~(size_t)15) + CS_SHAVE_TRACE_LDS * 16,
                              ne_shave_trace_kernel(m->img->dense_width, h->n_vars, m->img->dense_slots))))
            return rc;
          HIP_TRY(hipMalloc((void **)&m->d_tickets, (size_t)CS_TICKET_SLOTS * CS_TICKET_SLOT_WORDS * sizeof(unsigned)));
          HIP_TRY(hipMemset(m->d_tickets, 0, (size_t)CS_TICKET_SLOTS * CS_TICKET_SLOT_WORDS * sizeof(unsigned)));
          m->ticket_slots = CS_TICKET_SLOTS;
          m->packed_nw = 1; /* one set word per variable when every root domain has at most 32 values */
          for (int32_t v = 0; v < h->n_vars; v++)
            if ((int64_t)h->dom[v].hi - (int64_t)h->dom[v].lo + 1 > 32) m->packed_nw = 2;
          m->packed_bias = 0;
          for (int32_t v = 0; v < h->n_vars; v++)
            if (h->dom[v].lo - m->img->dense_dmin > m->packed_bias) m->packed_bias = h->dom[v].lo - m->img->dense_dmin;
          if (packed_nodes_per_wave(m->fb_words, h->n_vars, m->img->dense_width)) {
            if ((rc = lds_limit(2 * bytes, ne_packed_kernel(h->n_vars, m->packed_nw, m->img->dense_slots == 3)))) return rc;
            /* entry e of row u becomes e - (root_lo[u] - dmin) + bias (cs_kernels.hip.h, kernel 5) */
            const size_t per_row = (size_t)m->img->dense_slots * m->img->dense_cols, total = (size_t)h->n_vars * per_row;
            uint16_t *rel = (uint16_t *)malloc(total * sizeof(uint16_t));
            const uint8_t *src = (const uint8_t *)m->img->dense_tab;
            for (size_t i = 0; i < total; i++) {
              const int32_t u = (int32_t)(i / per_row);
              rel[i] = src[i] == 0xffu ? (uint16_t)0xffffu
                                       : (uint16_t)((int)src[i] - (h->dom[u].lo - m->img->dense_dmin) + m->packed_bias);
            }
            rc = upload(rel, total * sizeof(uint16_t), (int **)&m->d_packed_tab);
            free(rel);
            if (rc) return rc;
            {
              const size_t step_lds = ((2 * bytes + 15) & ~(size_t)15) + (size_t)16 * ((1 + m->packed_nw) * 256 + CS_STEP_QN) * sizeof(unsigned);
              if (step_lds <= 80u * 1024u &&
                  ((rc = lds_limit(step_lds, step_packed_kernel(h->n_vars, m->packed_nw, m->img->dense_slots == 3))) ||
                   (rc = lds_limit((2 * bytes + 15) & ~(size_t)15, step_import_kernel(h->n_vars, m->packed_nw, m->img->dense_slots == 3)))))
                return rc;
            }
          }
        }
      }
    }
  }

  int dev = 0;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  m->n_cus = prop.multiProcessorCount;

  const size_t nbytes = (size_t)(h->n_vars ? h->n_vars : 1) * sizeof(cs_val);
  {
    const size_t part = (nbytes + 63) & ~(size_t)63;
    unsigned char *dev = NULL;
    HIP_TRY(hipHostMalloc((void **)&m->h_one, 2 * part + 128, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&dev, m->h_one, 0));
    m->d_one_in = (cs_val *)dev;
    m->d_one_node = (cs_node_in *)(dev + part);
    m->d_one_res = (cs_node_out *)(dev + part + 64);
    m->d_one_out = (cs_val *)(dev + part + 128);
  }
  m->finalized = 1;
  return CSGPU_OK;
}

extern "C" int csgpu_model_set_kernel(csgpu_model *m, int which) {
  if (m == NULL || which < 0 || which > 7) return set_err(CSGPU_E_ARG, "bad argument");
  if (which >= 2) {
    if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
    if (which == 2 && !m->lds_waves) return set_err(CSGPU_E_LIMIT, "model does not qualify for the LDS-resident kernel");
    if (which == 3 && !m->fb_words) return set_err(CSGPU_E_LIMIT, "model does not qualify for the forbidden-set kernel");
    if (which == 4 && !m->dense_waves)
      return set_err(CSGPU_E_LIMIT, "model does not qualify for the register-resident forbidden-set kernel");
    if (which == 7 && !m->dense_waves)
      return set_err(CSGPU_E_LIMIT, "model does not qualify for the interval-only shaving kernel (dense pair table in LDS)");
    if (which == 6 && !clause_rounds_cpl(m))
      return set_err(CSGPU_E_LIMIT, "model does not qualify for the clause-resident kernel (at most 512 clauses)");
    if (which == 5 && !(m->dense_waves && packed_nodes_per_wave(m->fb_words, m->host->n_vars, m->img->dense_width)))
      return set_err(CSGPU_E_LIMIT, "model does not qualify for the several-nodes-per-wave kernel (at most 32 variables, 64 values)");
  }
  m->kernel_choice = which;
  return CSGPU_OK;
}

extern "C" int csgpu_model_qualifies(const csgpu_model *m, int which) {
  if (m == NULL || !m->finalized) return 0;
  switch (which) {
  case 1: return 1;
  case 2: return m->lds_waves != 0;
  case 3: return m->fb_words != 0;
  case 4: return m->dense_waves != 0;
  case 7: return m->dense_waves != 0;
  case 6: return clause_rounds_cpl(m) != 0;
  case 5: return m->dense_waves != 0 && packed_nodes_per_wave(m->fb_words, m->host->n_vars, m->img->dense_width) != 0;
  default: return 0;
  }
}

extern "C" void csgpu_internal_engine_ref(const csgpu_model *m, int delta) {
  if (m != NULL) const_cast<csgpu_model *>(m)->engines += delta;
}

extern "C" const int32_t *csgpu_internal_root_lo(const csgpu_model *m) {
  return m != NULL && m->finalized && m->fb_words ? m->d_root_lo : NULL;
}

extern "C" int csgpu_model_forbidden_words(const csgpu_model *m) { return m && m->finalized ? m->fb_words : 0; }

/* ---- cs_step.hip.h: one level of the search tree per launch ---- */
static const void *step_packed_kernel(int n_vars, int nw, int s3) {
#define CS_PICK_S(G, NW)                                                                           \
  return s3 ? (const void *)cs_step_packed<G, NW, true> : (const void *)cs_step_packed<G, NW, false>;
#define CS_PICK_NW(G)                                                                              \
  if (nw == 1) { CS_PICK_S(G, 1) }                                                                 \
  CS_PICK_S(G, 2)
  if (n_vars <= 16) { CS_PICK_NW(4) }
  CS_PICK_NW(2)
#undef CS_PICK_NW
#undef CS_PICK_S
}

static const void *step_import_kernel(int n_vars, int nw, int s3) {
#define CS_PICK_S(G, NW)                                                                           \
  return s3 ? (const void *)cs_step_import<G, NW, true> : (const void *)cs_step_import<G, NW, false>;
#define CS_PICK_NW(G)                                                                              \
  if (nw == 1) { CS_PICK_S(G, 1) }                                                                 \
  CS_PICK_S(G, 2)
  if (n_vars <= 16) { CS_PICK_NW(4) }
  CS_PICK_NW(2)
#undef CS_PICK_NW
#undef CS_PICK_S
}

/* LDS of a step workgroup of `waves` waves: the 16-bit table, then per wave the parent slots and the child queue */
static size_t step_packed_lds(const csgpu_model *m, int waves) {
  return ((2 * m->dense_bytes + 15) & ~(size_t)15) + (size_t)waves * ((1 + m->packed_nw) * 256 + CS_STEP_QN) * sizeof(unsigned);
}

static const void *step_shave_kernel(int width, int n_vars, int slots, int full) {
  const int chunks = (n_vars + CS_WAVE - 1) / CS_WAVE;
#define CS_PICK_S(E, R)                                                                                   \
  if (slots == 1) return full ? (const void *)cs_step_shave<E, R, 1, true> : (const void *)cs_step_shave<E, R, 1, false>; \
  if (slots == 3) return full ? (const void *)cs_step_shave<E, R, 3, true> : (const void *)cs_step_shave<E, R, 3, false>; \
  return full ? (const void *)cs_step_shave<E, R, 0, true> : (const void *)cs_step_shave<E, R, 0, false>;
#define CS_PICK(E)                                                                                 \
  switch (chunks <= 1 ? 1 : (chunks <= 2 ? 2 : 4)) {                                               \
  case 1: CS_PICK_S(E, 1)                                                                          \
  case 2: CS_PICK_S(E, 2)                                                                          \
  default: CS_PICK_S(E, 4)                                                                         \
  }
  if (width == 1) { CS_PICK(unsigned char) }
  CS_PICK(unsigned short)
#undef CS_PICK
#undef CS_PICK_S
}

static int model_max_width(const csgpu_model *m) {
  int maxw = 2;
  for (int32_t v = 0; v < m->host->n_vars; v++) {
    const int64_t w = (int64_t)m->host->dom[v].hi - (int64_t)m->host->dom[v].lo + 1;
    if (w > maxw) maxw = w > 0x7fffffff ? 0x7fffffff : (int)w;
  }
  return maxw;
}

/* LDS of a cs_step_shave workgroup: the dense table, then eight mask words per wave */
static size_t step_shave_lds(const csgpu_model *m) {
  return ((m->dense_bytes + 15) & ~(size_t)15) + (size_t)m->dense_waves * 8 * sizeof(unsigned);
}

extern "C" int csgpu_internal_step_kind(const csgpu_model *m) {
  if (m == NULL || !m->finalized || m->img == NULL || !m->dense_waves) return 0;
  if (m->d_packed_tab != NULL && packed_nodes_per_wave(m->fb_words, m->host->n_vars, m->img->dense_width) &&
      step_packed_lds(m, 16) <= 80u * 1024u) /* two workgroups of sixteen waves per CU */
    return 1;
  if (m->host->n_vars <= 256 && model_max_width(m) <= 256 && step_shave_lds(m) <= 160u * 1024u) return 2;
  return 0;
}

/* waves of a cs_step_shave launch with `stage_rows` staging rows: the resident grid, fewer when a wave's region would
 * not hold the children of four parents */
static int64_t step_shave_waves(const csgpu_model *m, int64_t stage_rows) {
  size_t wgs = (160u * 1024u) / step_shave_lds(m);
  if (wgs > (size_t)(32 / m->dense_waves)) wgs = (size_t)(32 / m->dense_waves);
  int64_t waves = (int64_t)m->n_cus * (int64_t)wgs * m->dense_waves;
  const int64_t by_stage = stage_rows / (4 * (int64_t)model_max_width(m));
  if (waves > by_stage) waves = by_stage / m->dense_waves * m->dense_waves;
  return waves;
}

extern "C" int64_t csgpu_internal_step_parents_limit(const csgpu_model *m, int64_t stage_rows) {
  const int kind = csgpu_internal_step_kind(m);
  if (kind == 1) return 0x3fffffff;
  if (kind != 2) return 0;
  const int maxw = model_max_width(m);
  const int64_t waves = step_shave_waves(m, stage_rows);
  if (waves < 1) return 0;
  return (stage_rows - waves * 2 * maxw) / maxw; /* every child of every parent may survive, and a wave stops two parents short of its region's end */
}

extern "C" int64_t csgpu_internal_step_stage_rows(const csgpu_model *m) {
  if (csgpu_internal_step_kind(m) != 2) return 0;
  return (int64_t)m->n_cus * 32 * 4 * model_max_width(m);
}

extern "C" int64_t csgpu_internal_step_waves(const csgpu_model *m) { return m == NULL ? 0 : (int64_t)m->n_cus * 32; }

static int launch_step_shave(const csgpu_model *m, const csgpu_step_launch *L, void *stream) {
  const int n = m->host->n_vars, maxw = model_max_width(m);
  int64_t waves = step_shave_waves(m, L->stage_rows);
  if (waves < m->dense_waves || (int64_t)L->parents > csgpu_internal_step_parents_limit(m, L->stage_rows))
    return set_err(CSGPU_E_LIMIT, "step kernel: staging buffer too small for the frontier");
  const int64_t by_parents = ((int64_t)L->parents + m->dense_waves - 1) / m->dense_waves * m->dense_waves;
  if (waves > by_parents) waves = by_parents;
  const int64_t grid = waves / m->dense_waves;
  const int64_t K = L->stage_rows / waves;
  cs_step_io io;
  io.pool = (const uint2 *)L->pool;
  io.first_row = (long long)L->first_row;
  io.parents = L->parents;
  io.chunk = 1;
  io.maxw = maxw;
  io.stage = (uint2 *)L->stage;
  io.K = (int)(K > 0x7fffffff ? 0x7fffffff : K);
  io.fill = L->fill;
  io.wstat = (unsigned long long *)L->wstat;
  io.ticket = L->ticket;
  io.solutions = L->solutions;
  io.stored = (unsigned long long *)L->stored;
  io.max_solutions = (long long)L->max_solutions;
  io.store_open = L->store_open;
  int nn = n, slots = m->img->dense_slots, dmin_d = m->img->dense_dmin;
  const void *tab_d = m->d_dense_tab;
  const int *root_lo_d = m->d_root_lo, *sym_off = m->d_sym_off;
  size_t tab_bytes = m->dense_bytes;
  void *args[] = { &nn, &tab_d, &slots, &dmin_d, &root_lo_d, &sym_off, &tab_bytes, &io };
  const int chunks_v = (n + CS_WAVE - 1) / CS_WAVE;
  const int lanes = (chunks_v <= 1 ? 1 : (chunks_v <= 2 ? 2 : 4)) * CS_WAVE;
  HIP_TRY(hipLaunchKernel(step_shave_kernel(m->img->dense_width, n, slots, n == lanes), dim3((unsigned)grid),
                          dim3((unsigned)(m->dense_waves * CS_WAVE)), args, step_shave_lds(m), (hipStream_t)stream));
  hipLaunchKernelGGL(cs_collect, dim3((unsigned)waves), dim3(256), 0, (hipStream_t)stream, (const unsigned *)L->fill, (int)waves,
                     (const uint2 *)L->stage, io.K, n, (uint2 *)L->pool, (long long)L->first_row, (int)L->parents,
                     0 /* every parent is drawn */, (const unsigned *)L->ticket, (const unsigned long long *)L->wstat,
                     (unsigned long long *)L->out, (const unsigned long long *)L->stored);
  HIP_TRY(hipGetLastError());
  return CSGPU_OK;
}

extern "C" int csgpu_internal_step(const csgpu_model *m, const csgpu_step_launch *L, void *stream) {
  const int kind = csgpu_internal_step_kind(m);
  if (kind == 0 || L == NULL || L->parents < 1) return set_err(CSGPU_E_ARG, "bad argument");
  if (kind == 2) return launch_step_shave(m, L, stream);
  const int n = m->host->n_vars;
  const int G = n <= 16 ? 4 : 2;
  int maxw = 2;
  for (int32_t v = 0; v < n; v++) {
    const int64_t w = (int64_t)m->host->dom[v].hi - (int64_t)m->host->dom[v].lo + 1;
    if (w > maxw) maxw = (int)w;
  }
  if (G * maxw > CS_STEP_QN) return set_err(CSGPU_E_LIMIT, "step kernel: interval too wide for the child queue");
  /* waves: the machine's, fewer when the staging buffer would leave a wave less than two rounds of worst-case children;
   * and no more than the parents need */
  int64_t waves = (int64_t)m->n_cus * 32;
  const int64_t by_stage = L->stage_rows / (2 * (int64_t)G * maxw);
  if (waves > by_stage) waves = by_stage;
  const int64_t by_parents = ((int64_t)L->parents + G - 1) / G;
  if (waves > by_parents) waves = by_parents;
  if (waves < 1) return set_err(CSGPU_E_LIMIT, "step kernel: staging buffer too small");
  int wg_waves = 16;
  while (wg_waves > waves) wg_waves >>= 1;
  const int64_t grid = waves / wg_waves;
  waves = grid * wg_waves;
  const int64_t K = L->stage_rows / waves;
  /* parents per ticket: about eight tickets per wave, at most 32 parents, and a chunk's worst case fits half a region */
  int64_t chunk = (int64_t)L->parents / (waves * 8);
  /* one word takes about 88 atomics per microsecond (MI355X_MICROARCH.md): with 32 parents per ticket a queens-16 frontier
   * of three million parents drew 78 tickets per microsecond and the whole launch waited for them (61 ms per search; 51 ms
   * with 64 and with 128 per ticket, 181 ms with 8) */
  int64_t chunk_max = 128;
  {
    const char *e = getenv("CSGPU_STEP_CHUNK_MAX"); /* tuning */
    if (e != NULL && atoi(e) > 0) chunk_max = atoi(e);
  }
  if (chunk > chunk_max) chunk = chunk_max;
  if (chunk * maxw > K / 2) chunk = K / (2 * maxw);
  chunk = chunk / G * G;
  if (chunk < G) chunk = G;
  cs_step_io io;
  io.pool = (const uint2 *)L->pool;
  io.first_row = (long long)L->first_row;
  io.parents = L->parents;
  io.chunk = (int)chunk;
  io.maxw = maxw;
  io.stage = (uint2 *)L->stage;
  io.K = (int)(K > 0x7fffffff ? 0x7fffffff : K);
  io.fill = L->fill;
  io.wstat = (unsigned long long *)L->wstat;
  io.ticket = L->ticket;
  io.solutions = L->solutions;
  io.stored = (unsigned long long *)L->stored;
  io.max_solutions = (long long)L->max_solutions;
  io.store_open = L->store_open;
  int nn = n, slots = m->img->dense_slots, bias = m->packed_bias;
  const void *tab_d = m->d_packed_tab;
  const int *root_lo_d = m->d_root_lo, *sym_off = m->d_sym_off;
  size_t tab_bytes = 2 * m->dense_bytes;
  void *args[] = { &nn, &tab_d, &slots, &root_lo_d, &sym_off, &bias, &tab_bytes, &io };
  HIP_TRY(hipLaunchKernel(step_packed_kernel(n, m->packed_nw, slots == 3), dim3((unsigned)grid),
                          dim3((unsigned)(wg_waves * CS_WAVE)), args, step_packed_lds(m, wg_waves), (hipStream_t)stream));
  hipLaunchKernelGGL(cs_collect, dim3((unsigned)waves), dim3(256), 0, (hipStream_t)stream, (const unsigned *)L->fill, (int)waves,
                     (const uint2 *)L->stage, io.K, n, (uint2 *)L->pool, (long long)L->first_row, (int)L->parents,
                     (int)chunk, (const unsigned *)L->ticket, (const unsigned long long *)L->wstat,
                     (unsigned long long *)L->out, (const unsigned long long *)L->stored);
  HIP_TRY(hipGetLastError());
  return CSGPU_OK;
}

extern "C" int csgpu_internal_step_import(const csgpu_model *m, csgpu_val *d_rows, int64_t first_row, int64_t count,
                                          void *stream) {
  if (csgpu_internal_step_kind(m) == 2) return CSGPU_OK; /* that path's pool holds interval rows */
  if (csgpu_internal_step_kind(m) != 1 || d_rows == NULL || count < 0) return set_err(CSGPU_E_ARG, "bad argument");
  if (count == 0) return CSGPU_OK;
  const int n = m->host->n_vars, G = n <= 16 ? 4 : 2;
  int nn = n, slots = m->img->dense_slots, bias = m->packed_bias;
  const void *tab_d = m->d_packed_tab;
  const int *root_lo_d = m->d_root_lo;
  size_t tab_bytes = 2 * m->dense_bytes;
  long long fr = first_row, cnt = count;
  int64_t grid = (count + (int64_t)G * 4 - 1) / ((int64_t)G * 4); /* four waves per workgroup, G rows per wave and step */
  if (grid > (int64_t)m->n_cus * 8) grid = (int64_t)m->n_cus * 8;
  void *args[] = { &nn, &tab_d, &slots, &root_lo_d, &bias, &tab_bytes, &d_rows, &fr, &cnt };
  HIP_TRY(hipLaunchKernel(step_import_kernel(n, m->packed_nw, slots == 3), dim3((unsigned)grid), dim3(256), args,
                          (tab_bytes + 15) & ~(size_t)15, (hipStream_t)stream));
  return CSGPU_OK;
}

extern "C" int csgpu_internal_step_export(const csgpu_model *m, csgpu_val *d_rows, int64_t first_row, int64_t count,
                                          void *stream) {
  if (csgpu_internal_step_kind(m) == 2) return CSGPU_OK;
  if (csgpu_internal_step_kind(m) != 1 || d_rows == NULL || count < 0) return set_err(CSGPU_E_ARG, "bad argument");
  if (count == 0) return CSGPU_OK;
  const int n = m->host->n_vars;
  const long long elements = (long long)count * n;
  uint2 *rows = (uint2 *)d_rows + (size_t)first_row * n;
  const unsigned grid = (unsigned)((elements + 255) / 256);
  if (m->packed_nw == 1)
    hipLaunchKernelGGL(cs_step_export<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, (const int *)m->d_root_lo, rows, elements);
  else
    hipLaunchKernelGGL(cs_step_export<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, (const int *)m->d_root_lo, rows, elements);
  HIP_TRY(hipGetLastError());
  return CSGPU_OK;
}

/* the register-resident forbidden-set kernel (kernel 4); sets_only: the states are the sets alone */
static int launch_regs(const csgpu_model *m, const csgpu_val *d_states_in, const uint64_t *d_forb_in,
                       const csgpu_node *d_nodes, csgpu_val *d_states_out, uint64_t *d_forb_out, csgpu_result *d_results,
                       int64_t batch, const uint64_t *d_batch, int sets_only, int flags, void *stream) {
  int csz = CS_CHUNK;
  {
    const int64_t machine_waves = (int64_t)m->n_cus * 32;
    while (csz > 1 && (batch + csz - 1) / csz < machine_waves) csz >>= 1;
  }
  const int64_t chunks = (batch + csz - 1) / csz;
  size_t wgs = (160u * 1024u) / m->dense_bytes;
  if (wgs > (size_t)(32 / m->dense_waves)) wgs = (size_t)(32 / m->dense_waves);
  int64_t g = (int64_t)m->n_cus * (int64_t)wgs;
  const int64_t need_wg = (chunks + m->dense_waves - 1) / m->dense_waves;
  g *= 2; /* twice the resident grid: the dispatcher backfills CUs whose waves finish early (measured +1.5 %) */
  if (g > need_wg) g = need_wg;
  int n = m->host->n_vars, slots = m->img->dense_slots, dmin_d = m->img->dense_dmin;
  const void *tab_d = m->d_dense_tab;
  const int *root_lo_d = m->d_root_lo, *sym_off = m->d_sym_off;
  long long nb_d = (long long)batch;
  void *args_d[] = { &n, &tab_d, &slots, &dmin_d, &root_lo_d, &sym_off, &d_states_in, &d_forb_in, &d_nodes,
                     &d_states_out, &d_forb_out, &d_results, &nb_d, &d_batch, &csz, &flags };
  const int per_wave = sets_only || m->kernel_choice == 4 ? 0 : packed_nodes_per_wave(m->fb_words, n, m->img->dense_width);
  if (per_wave) { /* kernel 5: one wave per group of nodes, grid-stride */
    const int64_t groups = (batch + per_wave - 1) / per_wave;
    int64_t gp = (int64_t)m->n_cus * (int64_t)wgs * 2;
    const int64_t need_gp = (groups + m->dense_waves - 1) / m->dense_waves;
    if (gp > need_gp) gp = need_gp;
    csz = m->packed_bias; /* this kernel's arguments in those positions */
    tab_d = m->d_packed_tab;
    HIP_TRY(hipLaunchKernel(ne_packed_kernel(n, m->packed_nw, slots == 3), dim3((unsigned)gp),
                            dim3((unsigned)(m->dense_waves * CS_WAVE)), args_d, 2 * m->dense_bytes, (hipStream_t)stream));
    return CSGPU_OK;
  }
  const int chunks_v = (n + CS_WAVE - 1) / CS_WAVE;
  const int lanes = (chunks_v <= 1 ? 1 : (chunks_v <= 2 ? 2 : 4)) * CS_WAVE;
  const int fast = n == lanes && d_forb_in != NULL && d_forb_out != NULL && flags == 0;
  HIP_TRY(hipLaunchKernel(ne_regs_kernel(m->img->dense_width, m->fb_words, n, fast, sets_only), dim3((unsigned)g),
                          dim3((unsigned)(m->dense_waves * CS_WAVE)), args_d, m->dense_bytes, (hipStream_t)stream));
  return CSGPU_OK;
}

/* kernel 7: intervals in, intervals out */
static std::mutex g_ticket_mutex;

/* the ticket slot of a launch on `stream` (NULL: none left, the kernel then uses static shares) */
static unsigned *ticket_slot(csgpu_model *m, void *stream) {
  if (m->d_tickets == NULL) return NULL;
  std::lock_guard<std::mutex> lock(g_ticket_mutex);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (stream != NULL && hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
  int slot = -1;
  if (cap == hipStreamCaptureStatusActive) {
    /* a captured launch keeps a slot of its own for good (taken from the top) */
    if (m->ticket_reserved < m->ticket_slots - CS_TICKET_STREAMS) slot = m->ticket_slots - 1 - m->ticket_reserved++;
  } else {
    for (int i = 0; i < m->ticket_streams; i++)
      if (m->ticket_stream[i] == stream) slot = i;
    if (slot < 0 && m->ticket_streams < CS_TICKET_STREAMS) {
      slot = m->ticket_streams++;
      m->ticket_stream[slot] = stream;
    }
  }
  return slot < 0 ? NULL : m->d_tickets + (size_t)slot * CS_TICKET_SLOT_WORDS;
}

static int launch_shave(const csgpu_model *m, const csgpu_val *d_states_in, const csgpu_node *d_nodes,
                        csgpu_val *d_states_out, csgpu_result *d_results, int64_t batch, const uint64_t *d_batch,
                        void *stream) {
  /* four or two nodes per chunk; fewer while that would leave a wave with less than four chunks */
  int csz = m->host->n_vars <= CS_WAVE ? 2 * CS_SHAVE_CHUNK : CS_SHAVE_CHUNK; /* four per ticket when a node is one register per lane */
  const int64_t machine_waves = (int64_t)m->n_cus * 32;
  while (csz > 1 && (batch + csz - 1) / csz < 4 * machine_waves) csz >>= 1; /* every wave gets several chunks */
  const int64_t chunks = (batch + csz - 1) / csz;
  size_t wgs = (160u * 1024u) / m->dense_bytes;
  if (wgs > (size_t)(32 / m->dense_waves)) wgs = (size_t)(32 / m->dense_waves);
  int64_t g = (int64_t)m->n_cus * (int64_t)wgs; /* the resident grid: persistent waves, work drawn by ticket */
  const int64_t need_wg = (chunks + m->dense_waves - 1) / m->dense_waves;
  if (g > need_wg) g = need_wg;
  int n = m->host->n_vars, slots = m->img->dense_slots, dmin_d = m->img->dense_dmin;
  const void *tab_d = m->d_dense_tab;
  const int *root_lo_d = m->d_root_lo, *sym_off = m->d_sym_off;
  long long nb_d = (long long)batch;
  unsigned *tickets = getenv("CSGPU_SHAVE_STATIC") != NULL ? NULL : ticket_slot((csgpu_model *)m, stream);
  int4 *no_trace = NULL;
  unsigned *no_trace_n = NULL;
  unsigned no_trace_cap = 0u;
  void *args[] = { &n, &tab_d, &slots, &dmin_d, &root_lo_d, &sym_off, &d_states_in, &d_nodes, &d_states_out, &d_results,
                   &nb_d, &d_batch, &csz, &tickets, &no_trace, &no_trace_n, &no_trace_cap };
  const int chunks_v = (n + CS_WAVE - 1) / CS_WAVE;
  const int lanes = (chunks_v <= 1 ? 1 : (chunks_v <= 2 ? 2 : 4)) * CS_WAVE;
  HIP_TRY(hipLaunchKernel(ne_shave_kernel(m->img->dense_width, n, slots, n == lanes), dim3((unsigned)g),
                          dim3((unsigned)(m->dense_waves * CS_WAVE)), args, m->dense_bytes, (hipStream_t)stream));
  return CSGPU_OK;
}

#ifdef CS_SHAVE_TIMELINE
extern "C" int csgpu_debug_shave_timeline(unsigned long long *out, int waves) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(cs_shave_tl), (size_t)waves * 3 * sizeof(unsigned long long)));
  return CSGPU_OK;
}
#endif

extern "C" int csgpu_propagate_batch_fb(const csgpu_model *m, const csgpu_val *d_states_in, const uint64_t *d_forb_in,
                                        const csgpu_node *d_nodes, csgpu_val *d_states_out, uint64_t *d_forb_out,
                                        csgpu_result *d_results, int64_t batch, void *stream) {
  return csgpu_internal_propagate_fb(m, d_states_in, d_forb_in, d_nodes, d_states_out, d_forb_out, d_results, batch, NULL,
                                     stream);
}

/* batch = upper bound the launch is sized for; d_batch (nullable) = the real count, on the device */
extern "C" int csgpu_internal_propagate_fb(const csgpu_model *m, const csgpu_val *d_states_in, const uint64_t *d_forb_in,
                                           const csgpu_node *d_nodes, csgpu_val *d_states_out, uint64_t *d_forb_out,
                                           csgpu_result *d_results, int64_t batch, const uint64_t *d_batch,
                                           void *stream) {
  if (m == NULL || batch < 0) return set_err(CSGPU_E_ARG, "null argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  if (!m->fb_words) return set_err(CSGPU_E_LIMIT, "model does not qualify for the forbidden-set kernel");
  if (batch == 0) return CSGPU_OK;
  if (d_states_in == NULL || d_nodes == NULL || d_states_out == NULL || d_results == NULL)
    return set_err(CSGPU_E_ARG, "null argument");
  /* nodes per wave at a time: CS_CHUNK for large batches (one coalesced record load per 16 nodes); a batch
   * smaller than the machine is spread thinner so that it does not run 16 nodes deep on a few waves */
  int csz = CS_CHUNK;
  {
    const int64_t machine_waves = (int64_t)m->n_cus * 32;
    while (csz > 1 && (batch + csz - 1) / csz < machine_waves) csz >>= 1;
  }
  const int64_t chunks = (batch + csz - 1) / csz;
  if (m->dense_waves && m->kernel_choice != 3)
    return launch_regs(m, d_states_in, d_forb_in, d_nodes, d_states_out, d_forb_out, d_results, batch, d_batch, 0, 0, stream);
  size_t wg_per_cu = (160u * 1024u) / m->fb_bytes;
  if (wg_per_cu > (size_t)(32 / m->fb_waves)) wg_per_cu = (size_t)(32 / m->fb_waves);
  if (wg_per_cu < 1) wg_per_cu = 1;
  int64_t grid = (int64_t)m->n_cus * (int64_t)wg_per_cu;
  const int64_t need = (chunks + m->fb_waves - 1) / m->fb_waves;
  if (grid > need) grid = need;
  cs_tables tab = m->tab;
  tab.adj_off = m->d_sym_off; /* the symmetric lists */
  const void *packed = m->d_sym_packed;
  int n_adj = m->img->sym_n_adj, obits = m->img->sym_obits, dmin = m->img->sym_dmin;
  const int *root_lo = m->d_root_lo;
  long long nb = (long long)batch;
  void *args[] = { &tab, &packed, &n_adj, &obits, &dmin, &root_lo, &d_states_in, &d_forb_in, &d_nodes,
                   &d_states_out, &d_forb_out, &d_results, &nb, &d_batch, &csz };
  HIP_TRY(hipLaunchKernel(ne_bitset_kernel(m->img->sym_width, m->fb_words, m->host->n_vars), dim3((unsigned)grid),
                          dim3((unsigned)(m->fb_waves * CS_WAVE)), args, m->fb_bytes, (hipStream_t)stream));
  return CSGPU_OK;
}

/* ---- sets-only states (include/csolve_gpu.h) ------------------------------------------------------------ */

__global__ void cs_fill_identity_nodes(cs_node_in *nodes, long long count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) {
    cs_node_in nd;
    nd.var = -1; nd.lo = 0; nd.hi = 0; nd.parent = (int)i;
    nodes[i] = nd;
  }
}

static int sets_ready(const csgpu_model *m) {
  if (m == NULL) return set_err(CSGPU_E_ARG, "null argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  if (!m->dense_waves) return set_err(CSGPU_E_LIMIT, "model does not qualify for the register-resident forbidden-set kernel");
  return CSGPU_OK;
}

extern "C" int csgpu_sets_pack(const csgpu_model *m, const csgpu_val *d_states, uint64_t *d_sets, int64_t count, void *stream) {
  int rc = sets_ready(m);
  if (rc != CSGPU_OK) return rc;
  if (count < 0) return set_err(CSGPU_E_ARG, "bad argument");
  if (count == 0) return CSGPU_OK;
  if (d_states == NULL || d_sets == NULL) return set_err(CSGPU_E_ARG, "null argument");
  if (count > 0x7fffffff) return set_err(CSGPU_E_LIMIT, "too many states");
  cs_node_in *d_nodes = NULL;
  cs_node_out *d_res = NULL;
  HIP_TRY(hipMalloc((void **)&d_nodes, (size_t)count * sizeof(cs_node_in)));
  if (hipMalloc((void **)&d_res, (size_t)count * sizeof(cs_node_out)) != hipSuccess) {
    (void)hipFree(d_nodes);
    return set_err(CSGPU_E_HIP, "out of device memory");
  }
  hipLaunchKernelGGL(cs_fill_identity_nodes, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_nodes, (long long)count);
  /* rebuild mode (every valued variable pushes), the intervals folded into the sets on the way out */
  rc = launch_regs(m, d_states, NULL, (const csgpu_node *)d_nodes, NULL, d_sets, (csgpu_result *)d_res, count, NULL, 0,
                   CS_K4_OUT_RESTRICT, stream);
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(d_nodes);
  (void)hipFree(d_res);
  if (rc != CSGPU_OK) return rc;
  if (e != hipSuccess) return set_err(CSGPU_E_HIP, "%s", hipGetErrorString(e));
  return CSGPU_OK;
}

extern "C" int csgpu_sets_unpack(const csgpu_model *m, const uint64_t *d_sets, csgpu_val *d_states, int64_t count, void *stream) {
  int rc = sets_ready(m);
  if (rc != CSGPU_OK) return rc;
  if (count < 0) return set_err(CSGPU_E_ARG, "bad argument");
  if (count == 0) return CSGPU_OK;
  if (d_states == NULL || d_sets == NULL) return set_err(CSGPU_E_ARG, "null argument");
  const int n = m->host->n_vars;
  int64_t blocks = (count * n + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  const unsigned long long *sets = (const unsigned long long *)d_sets;
  cs_val *states = (cs_val *)d_states;
  switch (m->fb_words) {
  case 1: hipLaunchKernelGGL(cs_sets_unpack<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, m->d_root_lo, sets, states, (long long)count); break;
  case 2: hipLaunchKernelGGL(cs_sets_unpack<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, m->d_root_lo, sets, states, (long long)count); break;
  default: hipLaunchKernelGGL(cs_sets_unpack<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, m->d_root_lo, sets, states, (long long)count); break;
  }
  HIP_TRY(hipGetLastError());
  return CSGPU_OK;
}

extern "C" int csgpu_propagate_batch_sets(const csgpu_model *m, const uint64_t *d_sets_in, const csgpu_node *d_nodes,
                                          uint64_t *d_sets_out, csgpu_result *d_results, int64_t batch, void *stream) {
  int rc = sets_ready(m);
  if (rc != CSGPU_OK) return rc;
  if (batch < 0) return set_err(CSGPU_E_ARG, "bad argument");
  if (batch == 0) return CSGPU_OK;
  if (d_sets_in == NULL || d_nodes == NULL || d_sets_out == NULL || d_results == NULL)
    return set_err(CSGPU_E_ARG, "null argument");
  return launch_regs(m, NULL, d_sets_in, d_nodes, NULL, d_sets_out, d_results, batch, NULL, 1, 0, stream);
}

extern "C" int csgpu_model_get_kernel(const csgpu_model *m) {
  if (m == NULL) return CSGPU_E_ARG;
  if (m->kernel_choice) return m->kernel_choice;
  /* pure != networks: the forbidden-set kernels, the sets rebuilt from the incoming state when the caller
   * carries none (queens-64, 2^18 nodes: 0.14 ms against 0.51 ms of kernel 2; queens-16: 0.06 against 0.51) */
  if (m->fb_words) {
    /* no sets passed: small models several nodes per wave with the sets rebuilt (5), otherwise the
     * interval-only shaving kernel (7), which needs no sets at all */
    if (m->dense_waves) return packed_nodes_per_wave(m->fb_words, m->host->n_vars, m->img->dense_width) ? 5 : 7;
    /* more than 256 variables and no sets to inherit: rebuilding them costs one list scan per VALUED variable of
     * the incoming state (sudoku-25, 2^18 nodes: 6.2 ms), the event-driven kernels scan one list per narrowing
     * (kernel 2: 1.16 ms, kernel 1: 1.18 ms).  Callers that carry the sets use csgpu_propagate_batch_fb (kernel 3:
     * 0.91 ms on that batch). */
    return m->lds_waves ? 2 : 1;
  }
  if (m->lds_waves) return 2;
  const int cpl = clause_rounds_cpl(m);
  return cpl >= 1 && cpl <= 4 ? 6 : 1; /* 8 clauses per lane: kernel 6 for small batches only, see below */
}

/* ---- batched propagation ----------------------------------------------------------- */

extern "C" int csgpu_propagate_batch(const csgpu_model *m, const csgpu_val *d_states_in, const csgpu_node *d_nodes,
                                     csgpu_val *d_states_out, csgpu_result *d_results, int64_t batch, void *stream) {
  return csgpu_propagate_batch_obj(m, d_states_in, d_nodes, d_states_out, d_results, batch, CS_DOM_MIN, CS_DOM_MAX,
                                   stream);
}

extern "C" int csgpu_propagate_batch_obj(const csgpu_model *m, const csgpu_val *d_states_in, const csgpu_node *d_nodes,
                                         csgpu_val *d_states_out, csgpu_result *d_results, int64_t batch,
                                         int32_t obj_lo, int32_t obj_hi, void *stream) {
  return csgpu_internal_propagate_obj(m, d_states_in, d_nodes, d_states_out, d_results, batch, NULL, obj_lo, obj_hi, stream);
}

extern "C" int csgpu_internal_propagate_obj(const csgpu_model *m, const csgpu_val *d_states_in, const csgpu_node *d_nodes,
                                            csgpu_val *d_states_out, csgpu_result *d_results, int64_t batch,
                                            const uint64_t *d_batch, int32_t obj_lo, int32_t obj_hi, void *stream) {
  return csgpu_internal_propagate_objdev(m, d_states_in, d_nodes, d_states_out, d_results, batch, d_batch, obj_lo, obj_hi,
                                         NULL, 0, stream);
}

/* d_best (nullable) / sense: the incumbent is read from device memory when the kernel starts */
extern "C" int csgpu_internal_propagate_objdev(const csgpu_model *m, const csgpu_val *d_states_in,
                                               const csgpu_node *d_nodes, csgpu_val *d_states_out,
                                               csgpu_result *d_results, int64_t batch, const uint64_t *d_batch,
                                               int32_t obj_lo, int32_t obj_hi, const int32_t *d_best, int sense,
                                               void *stream) {
  if (m == NULL || batch < 0) return set_err(CSGPU_E_ARG, "null argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  if (batch == 0) return CSGPU_OK; /* an empty batch needs no buffers */
  if (d_states_in == NULL || d_nodes == NULL || d_states_out == NULL || d_results == NULL)
    return set_err(CSGPU_E_ARG, "null argument");
  const size_t lds = m->slice * CS_WAVES_PER_BLOCK + m->k1_tab_bytes;
  /* resident workgroups per CU: LDS- and wave-slot-limited (32 waves per CU) */
  size_t per_cu = (160u * 1024u) / lds;
  if (per_cu > 32 / CS_WAVES_PER_BLOCK) per_cu = 32 / CS_WAVES_PER_BLOCK;
  if (per_cu < 1) per_cu = 1;
  int64_t blocks = (batch + CS_WAVES_PER_BLOCK - 1) / CS_WAVES_PER_BLOCK;
  const int64_t resident = (int64_t)m->n_cus * (int64_t)per_cu;
  /* enough workgroups to fill the chip several times over, the rest by grid stride */
  if (blocks > resident * 4) blocks = resident * 4;
  hipStream_t s = (hipStream_t)stream;
  const cs_val *in = (const cs_val *)d_states_in;
  const cs_node_in *nodes = (const cs_node_in *)d_nodes;
  cs_val *out = (cs_val *)d_states_out;
  cs_node_out *res = (cs_node_out *)d_results;
  cs_tables tab = m->tab;
  if (m->host->obj_var >= 0 && (obj_lo != CS_DOM_MIN || obj_hi != CS_DOM_MAX || (d_best != NULL && sense != 0))) {
    tab.obj_var = m->host->obj_var;
    tab.obj_lo = obj_lo;
    tab.obj_hi = obj_hi;
    if (d_best != NULL && sense != 0) {
      tab.obj_best_dev = d_best;
      tab.obj_sense = sense;
    }
  }
  const int auto_kernel = csgpu_model_get_kernel(m);
  if (auto_kernel == 7 && tab.obj_var < 0)
    return launch_shave(m, d_states_in, d_nodes, d_states_out, d_results, batch, d_batch, stream);
  if (auto_kernel >= 3 && auto_kernel <= 5 && tab.obj_var < 0)
    return csgpu_internal_propagate_fb(m, d_states_in, NULL, d_nodes, d_states_out, NULL, d_results, batch, d_batch, stream);
  const unsigned long long *bdev = (const unsigned long long *)d_batch;
  if (csgpu_model_get_kernel(m) == 2 && tab.obj_var < 0 && d_batch == NULL) {
    /* persistent workgroups: as many as stay resident (LDS- and wave-slot-limited) */
    size_t wg_per_cu = (160u * 1024u) / m->lds_bytes;
    if (wg_per_cu > (size_t)(32 / m->lds_waves)) wg_per_cu = (size_t)(32 / m->lds_waves);
    if (wg_per_cu < 1) wg_per_cu = 1;
    int64_t grid = (int64_t)m->n_cus * (int64_t)wg_per_cu;
    /* nodes a wave takes at a time: 16 when every wave gets several such chunks, fewer for smaller batches (a wave
     * with two chunks next to one with one is a launch twice as long as it needs to be) */
    int csz = (int)(batch / (grid * m->lds_waves * 4));
    csz = csz < 1 ? 1 : (csz > CS_CHUNK ? CS_CHUNK : csz);
    { const char *e = getenv("CSGPU_K2_CSZ"); if (e != NULL && atoi(e) >= 1 && atoi(e) <= CS_CHUNK) csz = atoi(e); } /* measurement override */
    const int64_t chunks = (batch + csz - 1) / csz;
    const int64_t need = (chunks + m->lds_waves - 1) / m->lds_waves;
    if (grid > need) grid = need;
    const dim3 blk((unsigned)(m->lds_waves * CS_WAVE));
    int n_adj = m->img->n_adj, obits = m->img->packed_obits, dmin = m->img->packed_dmin;
    long long nb = (long long)batch;
    const void *packed = m->d_adj_packed;
    void *args[] = { &tab, &packed, &n_adj, &obits, &dmin, &in, &nodes, &out, &res, &nb, &csz };
    HIP_TRY(hipLaunchKernel(ne_lds_kernel(m->img->packed_width, m->host->n_vars, m->lds_adj_global), dim3((unsigned)grid),
                            blk, args, m->lds_bytes, s));
    return CSGPU_OK;
  }
  /* kernel 6 when asked for; automatically for models of at most 256 clauses (faster at every batch size
   * measured), and with 257-512 clauses for small batches only (lower latency, 14 us against 21 us per launch
   * on schedule-20, but less throughput: 215 us against 122 us for 65,536 nodes) -- the search engine's
   * device-counted batches are small by construction */
  const int cpl6 = clause_rounds_cpl(m);
  const int use6 = m->kernel_choice == 6 ||
                   (m->kernel_choice == 0 && !m->lds_waves && !m->fb_words && cpl6 != 0 && (cpl6 <= 4 || d_batch != NULL || batch <= 8192));
  if (use6) {
    const size_t lds6 = ((((size_t)m->host->n_vars * sizeof(cs_val) + 16 + 15) & ~(size_t)15)) * CS_WAVES_PER_BLOCK;
    int64_t blocks6 = (batch + CS_WAVES_PER_BLOCK - 1) / CS_WAVES_PER_BLOCK;
    if (blocks6 > (int64_t)m->n_cus * 8 * 4) blocks6 = (int64_t)m->n_cus * 8 * 4;
#define CS_LAUNCH6(CPL, TREE)                                                                       \
  hipLaunchKernelGGL((cs_propagate_clause_rounds<CPL, TREE>), dim3((unsigned)blocks6), dim3(CS_BLOCK), lds6, s, tab, in, \
                     nodes, out, res, (long long)batch, bdev)
    switch (clause_rounds_cpl(m)) {
    case 1: if (m->has_tree_adj) CS_LAUNCH6(1, true); else CS_LAUNCH6(1, false); break;
    case 2: if (m->has_tree_adj) CS_LAUNCH6(2, true); else CS_LAUNCH6(2, false); break;
    case 4: if (m->has_tree_adj) CS_LAUNCH6(4, true); else CS_LAUNCH6(4, false); break;
    default: if (m->has_tree_adj) CS_LAUNCH6(8, true); else CS_LAUNCH6(8, false); break;
    }
#undef CS_LAUNCH6
    HIP_TRY(hipGetLastError());
    return CSGPU_OK;
  }
  if (m->has_tree_adj && m->k1_tab_bytes)
    hipLaunchKernelGGL((cs_propagate_events<true, true>), dim3((unsigned)blocks), dim3(CS_BLOCK), lds, s, tab, in, nodes,
                       out, res, (long long)batch, bdev);
  else if (m->has_tree_adj)
    hipLaunchKernelGGL((cs_propagate_events<true, false>), dim3((unsigned)blocks), dim3(CS_BLOCK), lds, s, tab, in, nodes,
                       out, res, (long long)batch, bdev);
  else if (m->k1_tab_bytes)
    hipLaunchKernelGGL((cs_propagate_events<false, true>), dim3((unsigned)blocks), dim3(CS_BLOCK), lds, s, tab, in, nodes,
                       out, res, (long long)batch, bdev);
  else
    hipLaunchKernelGGL((cs_propagate_events<false, false>), dim3((unsigned)blocks), dim3(CS_BLOCK), lds, s, tab, in, nodes,
                       out, res, (long long)batch, bdev);
  HIP_TRY(hipGetLastError());
  return CSGPU_OK;
}

/* one wave per state while four domain arrays fit 64 KB of LDS, else one workgroup per state */
static int launch_eval_root(const csgpu_model *m, const csgpu_val *d_states, const int32_t *d_list, const uint64_t *d_count,
                            int64_t count, int32_t *d_truth, void *stream) {
  const size_t row = (size_t)m->host->n_vars * sizeof(cs_val);
  if (row * CS_WAVES_PER_BLOCK <= 64u * 1024u) {
    /* grid-stride: enough workgroups to fill the machine, every wave keeps its share of the clause table */
    int64_t blocks = (count + CS_WAVES_PER_BLOCK - 1) / CS_WAVES_PER_BLOCK;
    if (blocks > (int64_t)m->n_cus * 16) blocks = (int64_t)m->n_cus * 16;
    /* a count that stays on the device is the handful of complete children of a search iteration, `count` only their
     * bound: one workgroup per CU (the waves stride over more), not thousands that start to find nothing to do */
    if (d_count != NULL && blocks > (int64_t)m->n_cus) blocks = (int64_t)m->n_cus;
    const int per = (m->img->n_clauses + CS_WAVE - 1) / CS_WAVE;
#define CS_LAUNCH_EVAL_T(CPL, TREE)                                                                \
  hipLaunchKernelGGL((cs_eval_root_waves<CPL, TREE>), dim3((unsigned)blocks), dim3(CS_BLOCK), row * CS_WAVES_PER_BLOCK, \
                     (hipStream_t)stream, m->tab, (const cs_val *)d_states, d_truth, (const int *)d_list,                \
                     (const unsigned long long *)d_count, (long long)count)
#define CS_LAUNCH_EVAL(CPL)                                                                        \
  do {                                                                                             \
    if (m->img->n_trees > 0) CS_LAUNCH_EVAL_T(CPL, true);                                          \
    else CS_LAUNCH_EVAL_T(CPL, false);                                                             \
  } while (0)
    if (per <= 1) CS_LAUNCH_EVAL(1);
    else if (per <= 2) CS_LAUNCH_EVAL(2);
    else if (per <= 4) CS_LAUNCH_EVAL(4);
    else if (per <= 8) CS_LAUNCH_EVAL(8);
    else CS_LAUNCH_EVAL(0);
#undef CS_LAUNCH_EVAL
#undef CS_LAUNCH_EVAL_T
  } else {
    hipLaunchKernelGGL(cs_eval_root, dim3((unsigned)count), dim3(CS_BLOCK), row + 16, (hipStream_t)stream, m->tab,
                       (const cs_val *)d_states, d_truth, (const int *)d_list, (const unsigned long long *)d_count);
  }
  HIP_TRY(hipGetLastError());
  return CSGPU_OK;
}

extern "C" int csgpu_eval_batch(const csgpu_model *m, const csgpu_val *d_states, int32_t *d_truth, int64_t batch,
                                void *stream) {
  if (m == NULL || d_states == NULL || d_truth == NULL || batch < 0) return set_err(CSGPU_E_ARG, "null argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  if (batch == 0) return CSGPU_OK;
  if (batch > 0x7fffffff) return set_err(CSGPU_E_LIMIT, "batch too large");
  return launch_eval_root(m, d_states, NULL, NULL, batch, d_truth, stream);
}

/* instance i = row d_list[i] of d_states, i < *d_count <= bound (the count stays on the device) */
extern "C" int csgpu_internal_eval_list(const csgpu_model *m, const csgpu_val *d_states, const int32_t *d_list,
                                        const uint64_t *d_count, int64_t bound, int32_t *d_truth, void *stream) {
  if (m == NULL || d_states == NULL || d_list == NULL || d_count == NULL || d_truth == NULL || bound < 0)
    return set_err(CSGPU_E_ARG, "null argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  if (bound == 0) return CSGPU_OK;
  if (bound > 0x7fffffff) return set_err(CSGPU_E_LIMIT, "batch too large");
  return launch_eval_root(m, d_states, d_list, d_count, bound, d_truth, stream);
}

extern "C" int csgpu_eval_clauses(const csgpu_model *m, const csgpu_val *d_state, csgpu_val *d_vals, void *stream) {
  if (m == NULL || d_state == NULL || d_vals == NULL) return set_err(CSGPU_E_ARG, "null argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  if (m->host->n_clauses == 0) return CSGPU_OK;
  const size_t lds = (size_t)m->host->n_vars * sizeof(cs_val) + 16;
  unsigned blocks = (unsigned)((m->host->n_clauses + CS_BLOCK - 1) / CS_BLOCK);
  if (blocks > 1024u) blocks = 1024u;
  hipLaunchKernelGGL(cs_eval_clauses, dim3(blocks), dim3(CS_BLOCK), lds, (hipStream_t)stream, m->tab,
                     (const cs_val *)d_state, (cs_val *)d_vals);
  HIP_TRY(hipGetLastError());
  return CSGPU_OK;
}

/* `count` values of one variable on one parent state, host buffers: the sibling batch of the drop-in shim (the
 * driver tries them one after the other, csolve.c:331-338; the device takes them at once).  The parent, the node
 * records, the results and the new states live in mapped pinned memory: one launch, one wait. */
extern "C" int csgpu_propagate_values(const csgpu_model *cm, const csgpu_val *state, int32_t var, const int32_t *values,
                                      int32_t count, csgpu_val *states_out, csgpu_result *results) {
  csgpu_model *m = (csgpu_model *)cm;
  if (m == NULL || state == NULL || values == NULL || states_out == NULL || results == NULL)
    return set_err(CSGPU_E_ARG, "null argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  const int n = m->host->n_vars;
  if (var < 0 || var >= n || count < 0 || count > (1 << 20)) return set_err(CSGPU_E_ARG, "bad argument");
  if (count == 0) return CSGPU_OK;
  const size_t row = (size_t)n * sizeof(cs_val);
  const size_t off_nodes = (row + 255) & ~(size_t)255;
  const size_t off_res = off_nodes + (((size_t)count * sizeof(cs_node_in) + 255) & ~(size_t)255);
  const size_t off_out = off_res + (((size_t)count * sizeof(cs_node_out) + 255) & ~(size_t)255);
  const size_t need = off_out + (size_t)count * row;
  if (need > m->values_cap) {
    if (m->h_values != NULL) (void)hipHostFree(m->h_values);
    m->h_values = NULL;
    m->values_cap = 0;
    HIP_TRY(hipHostMalloc((void **)&m->h_values, need * 2, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&m->d_values, m->h_values, 0));
    m->values_cap = need * 2;
  }
  memcpy(m->h_values, state, row);
  cs_node_in *nd = (cs_node_in *)(m->h_values + off_nodes);
  for (int32_t i = 0; i < count; i++) { nd[i].var = var; nd[i].lo = nd[i].hi = values[i]; nd[i].parent = 0; }
  int rc = csgpu_propagate_batch(m, (const csgpu_val *)m->d_values, (const csgpu_node *)(m->d_values + off_nodes),
                                 (csgpu_val *)(m->d_values + off_out), (csgpu_result *)(m->d_values + off_res), count, NULL);
  if (rc != CSGPU_OK) return rc;
  { const int rcw = wait_null_stream(); if (rcw != CSGPU_OK) return rcw; }
  memcpy(results, m->h_values + off_res, (size_t)count * sizeof *results);
  memcpy(states_out, m->h_values + off_out, (size_t)count * row);
  return CSGPU_OK;
}

/* One node with its trail.  The general kernel (one wave) records every narrowing with the clause that made
 * it -- what the reference's bind() keeps as binding_t.clause (csolve.h:73-79) and its conflict analysis walks
 * (conflict.c:290-316) -- and the point of failure. */
extern "C" int csgpu_propagate_one_traced(const csgpu_model *cm, const csgpu_val *state, csgpu_node node,
                                          csgpu_val *state_out, csgpu_result *result, int32_t *trace, int32_t cap,
                                          int32_t *count) {
  csgpu_model *m = const_cast<csgpu_model *>(cm);
  if (m == NULL || state == NULL || state_out == NULL || result == NULL || trace == NULL || cap < 1 || count == NULL)
    return set_err(CSGPU_E_ARG, "bad argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  if (m->d_adj_clause == NULL) {
    int rc = upload(m->img->adj_clause, (size_t)(m->img->n_adj ? m->img->n_adj : 1) * 4, &m->d_adj_clause);
    if (rc != CSGPU_OK) return rc;
  }
  if (m->trace_cap < cap) {
    if (m->h_trace != NULL) (void)hipHostFree(m->h_trace);
    m->h_trace = NULL;
    HIP_TRY(hipHostMalloc((void **)&m->h_trace, (size_t)cap * 16 + 16, hipHostMallocMapped));
    m->h_trace_n = (unsigned *)(m->h_trace + (size_t)cap * 4);
    m->trace_cap = cap;
  }
  const size_t nbytes = (size_t)m->host->n_vars * sizeof(cs_val);
  const size_t part = (nbytes + 63) & ~(size_t)63;
  node.parent = 0;
  memcpy(m->h_one, state, nbytes);
  memcpy(m->h_one + part, &node, sizeof node);
  *m->h_trace_n = 0u;
  cs_tables tab = m->tab;
  tab.adj_clause = m->d_adj_clause;
  void *dev = NULL;
  HIP_TRY(hipHostGetDevicePointer(&dev, m->h_trace, 0));
  tab.trace_log = (int4 *)dev;
  tab.trace_n = (unsigned *)((int32_t *)dev + (size_t)m->trace_cap * 4); /* where h_trace_n points */
  tab.trace_cap = (unsigned)cap;
  const size_t lds = m->slice * CS_WAVES_PER_BLOCK;
  int rc = lds_limit(lds, (const void *)cs_propagate_events<true, false, true>);
  if (rc != CSGPU_OK) return rc;
  hipLaunchKernelGGL((cs_propagate_events<true, false, true>), dim3(1), dim3(CS_BLOCK), lds, 0, tab,
                     (const cs_val *)m->d_one_in, (const cs_node_in *)m->d_one_node, (cs_val *)m->d_one_out,
                     (cs_node_out *)m->d_one_res, 1ll, (const unsigned long long *)NULL);
  HIP_TRY(hipGetLastError());
  { const int rcw = wait_null_stream(); if (rcw != CSGPU_OK) return rcw; }
  memcpy(result, m->h_one + part + 64, sizeof *result);
  if (result->status >= 0) memcpy(state_out, m->h_one + part + 128, nbytes);
  const unsigned n = *m->h_trace_n;
  *count = (int32_t)n;
  memcpy(trace, m->h_trace, (size_t)(n < (unsigned)cap ? n : (unsigned)cap) * 16);
  return CSGPU_OK;
}

/* One node of a pure != network with its trail as causes: kernel 7's tracing variant, one wave. */
/* ---- the resident single-node server (cs_shave.hip.h) ---- */
static const void *shave_server_kernel(int width, int n_vars, int slots) {
  const int chunks = (n_vars + CS_WAVE - 1) / CS_WAVE;
  const int r = chunks <= 1 ? 1 : (chunks <= 2 ? 2 : 4);
  const int sl = slots == 1 ? 1 : (slots == 3 ? 3 : 0);
#define CS_PICK_S(E, RR)                                                                           \
  switch (sl) {                                                                                    \
  case 1: return (const void *)cs_shave_server<E, RR, 1>;                                           \
  case 3: return (const void *)cs_shave_server<E, RR, 3>;                                           \
  default: return (const void *)cs_shave_server<E, RR, 0>;                                          \
  }
#define CS_PICK(E)                                                                                 \
  switch (r) {                                                                                     \
  case 1: CS_PICK_S(E, 1)                                                                          \
  case 2: CS_PICK_S(E, 2)                                                                          \
  default: CS_PICK_S(E, 4)                                                                         \
  }
  if (width == 1) { CS_PICK(unsigned char) }
  CS_PICK(unsigned short)
#undef CS_PICK
#undef CS_PICK_S
}

static double srv_now(void) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static cs_mailbox_head *srv_box(const csgpu_model *m) { return (cs_mailbox_head *)m->h_box; }

/* does this model use the server?  (kernel 7's models whose table leaves room for the trail in LDS) */
static int server_usable(csgpu_model *m) {
  if (m->srv_off) return 0;
  if (m->h_box != NULL) return 1;
  const char *e = getenv("CSGPU_SERVER");
  const size_t lds = ((m->dense_bytes + 15) & ~(size_t)15) + CS_SHAVE_TRACE_LDS * 16;
  if ((e != NULL && e[0] == '0') || !m->dense_waves || lds > 160u * 1024u) { m->srv_off = 1; return 0; }
  const size_t nbytes = ((size_t)m->host->n_vars * sizeof(cs_val) + 63) & ~(size_t)63;
  m->box_state_off = (sizeof(cs_mailbox_head) + 63) & ~(size_t)63;
  m->box_out_off = m->box_state_off + nbytes;
  m->box_trace_off = m->box_out_off + nbytes;
  const size_t total = m->box_trace_off + (size_t)CS_SHAVE_TRACE_LDS * 16;
  if (hipHostMalloc((void **)&m->h_box, total, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipStreamCreateWithFlags(&m->srv_stream, hipStreamNonBlocking) != hipSuccess ||
      hipFuncSetAttribute(shave_server_kernel(m->img->dense_width, m->host->n_vars, m->img->dense_slots),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    if (m->h_box != NULL) (void)hipHostFree(m->h_box);
    m->h_box = NULL;
    m->srv_off = 1;
    return 0;
  }
  memset(m->h_box, 0, total);
  m->srv_seq = 0;
  for (int i = 0; i < 16; i++)
    if (g_srv_models[i] == NULL) { g_srv_models[i] = m; break; }
  return 1;
}

/* start a resident wave (none is alive) and wait until it says so */
static int server_start(csgpu_model *m) {
  cs_mailbox_head *box = srv_box(m);
  if (m->srv_launched) server_reap(m); /* the previous one has left (alive == 0): reap it */
  void *dev = NULL;
  HIP_TRY(hipHostGetDevicePointer(&dev, m->h_box, 0));
  cs_mailbox_head *d_box = (cs_mailbox_head *)dev;
  unsigned long long *d_in = (unsigned long long *)((unsigned char *)dev + m->box_state_off);
  unsigned long long *d_out = (unsigned long long *)((unsigned char *)dev + m->box_out_off);
  int4 *d_trace = (int4 *)((unsigned char *)dev + m->box_trace_off);
  int n = m->host->n_vars, slots = m->img->dense_slots, dmin_d = m->img->dense_dmin;
  const void *tab_d = m->d_dense_tab;
  const int *root_lo_d = m->d_root_lo, *sym_off = m->d_sym_off;
  unsigned cap = CS_SHAVE_TRACE_LDS;
  unsigned long long idle = 200000ull; /* 2 ms of the 100 MHz clock: a search calls every few microseconds */
  { const char *e = getenv("CSGPU_SERVER_IDLE_US"); if (e != NULL && atoll(e) > 0) idle = (unsigned long long)atoll(e) * 100ull; }
  void *args[] = { &n, &tab_d, &slots, &dmin_d, &root_lo_d, &sym_off, &d_box, &d_in, &d_out, &d_trace, &cap, &idle };
  const size_t lds = ((m->dense_bytes + 15) & ~(size_t)15) + CS_SHAVE_TRACE_LDS * 16;
  HIP_TRY(hipLaunchKernel(shave_server_kernel(m->img->dense_width, n, slots), dim3(1), dim3((unsigned)(m->dense_waves * CS_WAVE)),
                          args, lds, m->srv_stream));
  m->srv_launched = 1;
  m->srv_starts++;
  const double t0 = srv_now();
  while (__atomic_load_n(&box->alive, __ATOMIC_ACQUIRE) == 0u) {
    /* a wave that has already served the pending request and left again also counts */
    if (__atomic_load_n(&box->ack_seq, __ATOMIC_ACQUIRE) == m->srv_seq && hipStreamQuery(m->srv_stream) == hipSuccess) break;
    if (srv_now() - t0 > 10.0) return set_err(CSGPU_E_HIP, "the single-node server did not start");
  }
  return CSGPU_OK;
}

/* wait for the server's stream by polling (hipStreamSynchronize on a kernel that has run for long sleeps in the kernel
 * driver and is woken milliseconds late: the solution check of a queens-64 search took 9 ms instead of 2) */
static void server_reap(csgpu_model *m) {
  const double t0 = srv_now();
  for (;;) {
    const hipError_t e = hipStreamQuery(m->srv_stream);
    if (e != hipErrorNotReady) break;
    if (srv_now() - t0 > 5.0) { (void)hipStreamSynchronize(m->srv_stream); break; }
  }
}

static void server_stop(csgpu_model *m) {
  if (m->h_box == NULL || !m->srv_launched) return;
  cs_mailbox_head *box = srv_box(m);
  __atomic_store_n(&box->stop, 1u, __ATOMIC_RELEASE);
  server_reap(m);
  __atomic_store_n(&box->stop, 0u, __ATOMIC_RELEASE);
  m->srv_launched = 0;
}

/* one node through the mailbox; trace == NULL: no trail wanted */
static int server_call(csgpu_model *m, const csgpu_val *state, csgpu_node node, csgpu_val *state_out, csgpu_result *result,
                       int32_t *trace, int32_t cap, int32_t *count) {
  cs_mailbox_head *box = srv_box(m);
  const size_t nbytes = (size_t)m->host->n_vars * sizeof(cs_val);
  const double t0 = srv_now();
  memcpy(m->h_box + m->box_state_off, state, nbytes);
  box->node.var = node.var; box->node.lo = node.lo; box->node.hi = node.hi; box->node.parent = 0;
  box->want_trace = trace != NULL ? 1u : 0u;
  const unsigned seq = ++m->srv_seq;
  const double t1 = srv_now();
  __atomic_store_n(&box->req_seq, seq, __ATOMIC_RELEASE);
  double t_start = 0.0;
  if (__atomic_load_n(&box->alive, __ATOMIC_ACQUIRE) == 0u) {
    const double ts = srv_now();
    const int rc = server_start(m);
    if (rc != CSGPU_OK) return rc;
    t_start = srv_now() - ts;
  }
  unsigned spins = 0;
  while (__atomic_load_n(&box->ack_seq, __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 0xfffu) == 0u) {
      /* the wave may have left (idle for too long) just before the request arrived: start another, which finds it */
      if (__atomic_load_n(&box->alive, __ATOMIC_ACQUIRE) == 0u && __atomic_load_n(&box->ack_seq, __ATOMIC_ACQUIRE) != seq) {
        const double ts = srv_now();
        const int rc = server_start(m);
        if (rc != CSGPU_OK) return rc;
        t_start += srv_now() - ts;
      }
      if (srv_now() - t1 > 30.0) return set_err(CSGPU_E_HIP, "the single-node server does not answer");
    }
  }
  const double t2 = srv_now();
  result->status = box->result.status; result->props = box->result.props;
  result->revisions = box->result.revisions; result->rounds = box->result.rounds;
  if (result->status >= 0) memcpy(state_out, m->h_box + m->box_out_off, nbytes);
  if (trace != NULL) {
    const unsigned made = box->trace_n;
    *count = (int32_t)made;
    unsigned kept = made < CS_SHAVE_TRACE_LDS ? made : CS_SHAVE_TRACE_LDS;
    if (kept > (unsigned)cap) kept = (unsigned)cap;
    memcpy(trace, m->h_box + m->box_trace_off, (size_t)kept * 16);
  }
  const double t3 = srv_now();
  m->srv_seconds[0] += t1 - t0;
  m->srv_seconds[1] += t2 - t1 - t_start;
  m->srv_seconds[2] += t3 - t2;
  m->srv_seconds[3] += t_start;
  m->srv_calls++;
  return CSGPU_OK;
}

extern "C" int csgpu_internal_server_warm(csgpu_model *m) {
  if (m == NULL || !m->finalized || !server_usable(m)) return CSGPU_OK;
  if (__atomic_load_n(&srv_box(m)->alive, __ATOMIC_ACQUIRE) != 0u) return CSGPU_OK;
  return server_start(m);
}

/* where the time of the single-node calls went: seconds[0..3] = host copy in, ring + wait, copy out, server (re)starts;
 * calls, starts.  For the launch path (CSGPU_SERVER=0): copy in, launch submit, wait, copy out (starts = 0). */
extern "C" int csgpu_debug_one_timing(const csgpu_model *m, double *seconds, uint64_t *calls, uint64_t *starts) {
  if (m == NULL || seconds == NULL || calls == NULL || starts == NULL) return set_err(CSGPU_E_ARG, "null argument");
  for (int i = 0; i < 4; i++) seconds[i] = m->srv_seconds[i];
  *calls = m->srv_calls;
  *starts = m->srv_starts;
  return CSGPU_OK;
}

extern "C" int csgpu_propagate_one_causes(const csgpu_model *cm, const csgpu_val *state, csgpu_node node,
                                          csgpu_val *state_out, csgpu_result *result, int32_t *trace, int32_t cap,
                                          int32_t *count) {
  csgpu_model *m = const_cast<csgpu_model *>(cm);
  if (m == NULL || state == NULL || state_out == NULL || result == NULL || trace == NULL || cap < 1 || count == NULL)
    return set_err(CSGPU_E_ARG, "bad argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  if (!m->dense_waves) return set_err(CSGPU_E_LIMIT, "model does not qualify for the interval-only shaving kernel");
  if (server_usable(m)) return server_call(m, state, node, state_out, result, trace, cap, count);
  if (m->trace_cap < cap) {
    if (m->h_trace != NULL) (void)hipHostFree(m->h_trace);
    m->h_trace = NULL;
    HIP_TRY(hipHostMalloc((void **)&m->h_trace, (size_t)cap * 16 + 16, hipHostMallocMapped));
    m->h_trace_n = (unsigned *)(m->h_trace + (size_t)cap * 4);
    m->trace_cap = cap;
  }
  const size_t nbytes = (size_t)m->host->n_vars * sizeof(cs_val);
  const size_t part = (nbytes + 63) & ~(size_t)63;
  node.parent = 0;
  const double tl0 = srv_now();
  memcpy(m->h_one, state, nbytes);
  memcpy(m->h_one + part, &node, sizeof node);
  *m->h_trace_n = 0u;
  void *dev = NULL;
  HIP_TRY(hipHostGetDevicePointer(&dev, m->h_trace, 0));
  int4 *d_trace = (int4 *)dev;
  unsigned *d_trace_n = (unsigned *)((int32_t *)dev + (size_t)m->trace_cap * 4);
  unsigned ucap = (unsigned)cap;
  int n = m->host->n_vars, slots = m->img->dense_slots, dmin_d = m->img->dense_dmin, csz = 1;
  const void *tab_d = m->d_dense_tab;
  const int *root_lo_d = m->d_root_lo, *sym_off = m->d_sym_off;
  long long nb_d = 1;
  const uint64_t *d_batch = NULL;
  unsigned *tickets = NULL;
  const csgpu_val *d_in = (const csgpu_val *)m->d_one_in;
  const csgpu_node *d_node = (const csgpu_node *)m->d_one_node;
  csgpu_val *d_out = (csgpu_val *)m->d_one_out;
  csgpu_result *d_res = (csgpu_result *)m->d_one_res;
  void *args[] = { &n, &tab_d, &slots, &dmin_d, &root_lo_d, &sym_off, &d_in, &d_node, &d_out, &d_res,
                   &nb_d, &d_batch, &csz, &tickets, &d_trace, &d_trace_n, &ucap };
  const size_t lds_trace = ((m->dense_bytes + 15) & ~(size_t)15) + CS_SHAVE_TRACE_LDS * 16;
  if (lds_trace > 160u * 1024u) return set_err(CSGPU_E_LIMIT, "no room in LDS for the trail next to the pair table");
  const double tl1 = srv_now();
  HIP_TRY(hipLaunchKernel(ne_shave_trace_kernel(m->img->dense_width, n, slots), dim3(1), dim3((unsigned)(m->dense_waves * CS_WAVE)),
                          args, lds_trace, (hipStream_t)NULL));
  const double tl2 = srv_now();
  { const int rcw = wait_null_stream(); if (rcw != CSGPU_OK) return rcw; }
  const double tl3 = srv_now();
  memcpy(result, m->h_one + part + 64, sizeof *result);
  if (result->status >= 0) memcpy(state_out, m->h_one + part + 128, nbytes);
  const unsigned made = *m->h_trace_n;
  *count = (int32_t)made;
  unsigned kept = made < CS_SHAVE_TRACE_LDS ? made : CS_SHAVE_TRACE_LDS; /* what the kernel's LDS buffer holds */
  if (kept > (unsigned)cap) kept = (unsigned)cap;
  memcpy(trace, m->h_trace, (size_t)kept * 16);
  m->srv_seconds[0] += tl1 - tl0; m->srv_seconds[1] += tl2 - tl1; m->srv_seconds[2] += tl3 - tl2; m->srv_seconds[3] += srv_now() - tl3;
  m->srv_calls++;
  return CSGPU_OK;
}

/* ---- the reference's own failure chain of one node (cs_chain.hip.h) ---- */
#define CS_CHAIN_FRAMES (1 << 16)
#define CS_CHAIN_BUMPS 4096

/* operand of NOT(EQ(l, r)): a variable, or `variable + constant` with the constant on the right */
static int chain_operand(const cs_model *h, int32_t node, int *x, int *c) {
  const cs_node *nd = &h->nodes[node];
  if (nd->op == CS_OP_VAR) { *x = nd->a; *c = 0; return 1; }
  if (nd->op == CS_OP_ADD) {
    const cs_node *l = &h->nodes[nd->a], *r = &h->nodes[nd->b];
    if (l->op == CS_OP_VAR && r->op == CS_OP_CONST && r->a == r->b) { *x = l->a | CS_CHAIN_ADD; *c = r->a; return 1; }
  }
  return 0;
}

static int chain_build(csgpu_model *m) {
  if (m->chain_state != 0) return m->chain_state;
  const cs_model *h = m->host;
  m->chain_state = -1;
  if (!m->finalized || h->n_clauses < 1 || h->n_clauses > 60000 || h->n_vars < 1 || h->list_off == NULL) return -1;
  if ((size_t)h->n_vars * sizeof(cs_val) + (size_t)h->n_clauses * 2 + 64 > 160u * 1024u) return -1;
  cs_chain_clause *cl = (cs_chain_clause *)malloc((size_t)h->n_clauses * sizeof *cl);
  if (cl == NULL) return -1;
  for (int32_t c = 0; c < h->n_clauses; c++) {
    const cs_node *top = &h->nodes[h->clause_node[c]];
    if (top->op == CS_OP_CONST) { /* a constant element of the root (a folded bound): in no variable's list */
      cl[c].lx = -1; cl[c].lc = 0; cl[c].rx = -1; cl[c].rc = 0;
      continue;
    }
    int ok = top->op == CS_OP_NOT && h->nodes[top->a].op == CS_OP_EQ;
    if (ok) {
      const cs_node *eq = &h->nodes[top->a];
      ok = chain_operand(h, eq->a, &cl[c].lx, &cl[c].lc) && chain_operand(h, eq->b, &cl[c].rx, &cl[c].rc);
    }
    if (!ok) { free(cl); return -1; } /* another clause shape: the walk of cs_chain.hip.h does not cover it */
  }
  int rc = upload(cl, (size_t)h->n_clauses * sizeof *cl, (int **)&m->d_chain_cl);
  free(cl);
  if (rc == CSGPU_OK) rc = upload(h->list_off, ((size_t)h->n_vars + 1) * sizeof(int32_t), &m->d_chain_off);
  if (rc == CSGPU_OK) rc = upload(h->list, (size_t)(h->list_off[h->n_vars] ? h->list_off[h->n_vars] : 1) * sizeof(int32_t), &m->d_chain_list);
  if (rc != CSGPU_OK) return -1;
  if (hipMalloc((void **)&m->d_chain_frames, (size_t)CS_CHAIN_FRAMES * sizeof(cs_chain_frame)) != hipSuccess ||
      hipHostMalloc((void **)&m->h_chain_out, (4 + CS_CHAIN_BUMPS) * sizeof(int), hipHostMallocMapped) != hipSuccess ||
      hipFuncSetAttribute((const void *)cs_ne_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  m->chain_state = 1;
  return 1;
}

extern "C" int csgpu_propagate_one_chain(const csgpu_model *cm, const csgpu_val *state, csgpu_node node, int32_t *status,
                                         int32_t *props_out, int32_t *bumps, int32_t cap, int32_t *count) {
  csgpu_model *m = const_cast<csgpu_model *>(cm);
  if (m == NULL || state == NULL || status == NULL || props_out == NULL || bumps == NULL || count == NULL || cap < 0)
    return set_err(CSGPU_E_ARG, "bad argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  if (node.var < 0 || node.var >= m->host->n_vars) return set_err(CSGPU_E_ARG, "no such variable");
  if (chain_build(m) != 1)
    return set_err(CSGPU_E_LIMIT, "the model is not a network of NOT(EQ(x [+ c], y [+ d])) clauses: no reference-order walk");
  const size_t nbytes = (size_t)m->host->n_vars * sizeof(cs_val);
  memcpy(m->h_one, state, nbytes); /* the mapped staging area of the single-node calls */
  void *dev = NULL;
  HIP_TRY(hipHostGetDevicePointer(&dev, m->h_chain_out, 0));
  int *d_out = (int *)dev, *d_bumps = (int *)dev + 4;
  cs_node_in nd;
  nd.var = node.var; nd.lo = node.lo; nd.hi = node.hi; nd.parent = 0;
  const size_t lds = ((nbytes + 15) & ~(size_t)15) + (((size_t)m->host->n_clauses * 2 + 15) & ~(size_t)15);
  hipLaunchKernelGGL(cs_ne_chain, dim3(1), dim3(64), lds, (hipStream_t)NULL, m->host->n_vars, m->host->n_clauses,
                     (const cs_chain_clause *)m->d_chain_cl, (const int *)m->d_chain_off, (const int *)m->d_chain_list,
                     (const cs_val *)m->d_one_in, nd, m->d_chain_frames, (int)CS_CHAIN_FRAMES, d_out, d_bumps, (int)CS_CHAIN_BUMPS);
  HIP_TRY(hipGetLastError());
  { const int rcw = wait_null_stream(); if (rcw != CSGPU_OK) return rcw; }
  if (m->h_chain_out[3] != 0) return set_err(CSGPU_E_LIMIT, "the reference-order walk overflowed its frame stack or its bump list");
  *status = m->h_chain_out[0];
  *props_out = m->h_chain_out[1];
  const int made = m->h_chain_out[2];
  *count = made;
  memcpy(bumps, m->h_chain_out + 4, (size_t)(made < cap ? made : cap) * sizeof(int32_t));
  return CSGPU_OK;
}

extern "C" int csgpu_propagate_one(const csgpu_model *m, const csgpu_val *state, csgpu_node node, csgpu_val *state_out,
                                   csgpu_result *result) {
  if (m == NULL || state == NULL || state_out == NULL || result == NULL) return set_err(CSGPU_E_ARG, "null argument");
  if (!m->finalized) return set_err(CSGPU_E_STATE, "model is not finalized");
  if ((m->kernel_choice == 0 || m->kernel_choice == 7) && m->dense_waves && server_usable(const_cast<csgpu_model *>(m)))
    return server_call(const_cast<csgpu_model *>(m), state, node, state_out, result, NULL, 0, NULL);
  const size_t nbytes = (size_t)m->host->n_vars * sizeof(cs_val);
  const size_t part = (nbytes + 63) & ~(size_t)63;
  node.parent = 0;
  memcpy(m->h_one, state, nbytes);
  memcpy(m->h_one + part, &node, sizeof node);
  int rc = csgpu_propagate_batch(m, (const csgpu_val *)m->d_one_in, (const csgpu_node *)m->d_one_node,
                                 (csgpu_val *)m->d_one_out, (csgpu_result *)m->d_one_res, 1, NULL);
  if (rc != CSGPU_OK) return rc;
  { const int rcw = wait_null_stream(); if (rcw != CSGPU_OK) return rcw; } /* one launch, one wait: the kernel worked on the mapped host buffers */
  memcpy(result, m->h_one + part + 64, sizeof *result);
  if (result->status >= 0) memcpy(state_out, m->h_one + part + 128, nbytes);
  return CSGPU_OK;
}
