/* cs_internal.h -- entry points shared between the objects of this package (the drop-in shim
 * builds its host model from the driver's trees and hands it over).  Not part of the public ABI. */
#ifndef CS_INTERNAL_H
#define CS_INTERNAL_H

#include "../../include/csolve_gpu.h"
#include "cs_model.h"

#ifdef __cplusplus
extern "C" {
#endif

/* domains_are_root = 0: the domains the model carries belong to some search node, so clauses
 * entailed by them must NOT be dropped at finalize */
int csgpu_model_from_host(cs_model *host, int lists_final, int domains_are_root, csgpu_model **out);
cs_model *csgpu_model_host(csgpu_model *m);

/* the batched fixpoint launched for an upper bound `batch` with the real node count left in device memory
 * (`d_batch`, nullable): the search engine's small iterations need no host round trip for the count */
int csgpu_internal_propagate_fb(const csgpu_model *m, const csgpu_val *d_states_in, const uint64_t *d_forb_in,
                                const csgpu_node *d_nodes, csgpu_val *d_states_out, uint64_t *d_forb_out,
                                csgpu_result *d_results, int64_t batch, const uint64_t *d_batch, void *stream);
int csgpu_internal_propagate_obj(const csgpu_model *m, const csgpu_val *d_states_in, const csgpu_node *d_nodes,
                                 csgpu_val *d_states_out, csgpu_result *d_results, int64_t batch,
                                 const uint64_t *d_batch, int32_t obj_lo, int32_t obj_hi, void *stream);

/* ... with the incumbent bound read from device memory (d_best, nullable; sense 1 = minimise, 2 = maximise) */
int csgpu_internal_propagate_objdev(const csgpu_model *m, const csgpu_val *d_states_in, const csgpu_node *d_nodes,
                                    csgpu_val *d_states_out, csgpu_result *d_results, int64_t batch,
                                    const uint64_t *d_batch, int32_t obj_lo, int32_t obj_hi, const int32_t *d_best,
                                    int sense, void *stream);
/* root lower bounds of the variables in device memory (NULL unless the model qualifies for the forbidden-set
 * kernels): bit k of a set word = value root_lo + k */
const int32_t *csgpu_internal_root_lo(const csgpu_model *m);
/* a search engine is built on / freed from the model: while any exists csgpu_model_add_conflict on the finalized
 * model is refused (it would free the device tables the engine's kernels and graphs point into) */
void csgpu_internal_engine_ref(const csgpu_model *m, int delta);
/* csgpu_eval_batch over the rows d_list[0 .. *d_count) of d_states, *d_count <= bound */
int csgpu_internal_eval_list(const csgpu_model *m, const csgpu_val *d_states, const int32_t *d_list,
                             const uint64_t *d_count, int64_t bound, int32_t *d_truth, void *stream);

/* start the resident single-node server of this model now (csolve_gpu.h, csgpu_debug_one_timing) instead of with the
 * first single-node call: the drop-in does it when it attaches, so that the start (stream, mailbox, code load, launch)
 * is part of the set-up and not of the driver's first propagate_clauses.  No-op for models without a server. */
int csgpu_internal_server_warm(csgpu_model *m);

/* ---- one level of the search tree in one launch (cs_step.hip.h): branch + fixpoints of the children + store ---- */
typedef struct csgpu_step_launch {
  const csgpu_val *pool;  /* parents (engine rows): rows first_row .. first_row + parents - 1, drawn from the top down */
  int64_t first_row;
  int32_t parents;
  csgpu_val *stage;       /* staging rows for the survivors (private regions per wave) */
  int64_t stage_rows;
  uint32_t *fill;         /* [csgpu_internal_step_waves] */
  uint64_t *wstat;        /* [csgpu_internal_step_waves][8] */
  uint32_t *ticket;       /* one word, zero at launch */
  uint64_t *out;          /* [8]: consumed parents, survivors, nodes, cuts, props, revisions, solutions, stored */
  int32_t *solutions;     /* [max_solutions][n_vars] */
  uint64_t *stored;
  int64_t max_solutions;
  int32_t store_open;
} csgpu_step_launch;
/* 0: the model has no step kernel; 1: cs_step_packed (pure != network of at most 32 variables, engine rows in the pool);
 * 2: cs_step_shave (33 to 256 variables, plain interval rows) */
int csgpu_internal_step_kind(const csgpu_model *m);
/* parents one launch may be given with `stage_rows` staging rows (kind 2 sizes for the worst case: every child survives) */
int64_t csgpu_internal_step_parents_limit(const csgpu_model *m, int64_t stage_rows);
/* staging rows that let a launch of kind 2 use the whole machine (0 for the others) */
int64_t csgpu_internal_step_stage_rows(const csgpu_model *m);
/* waves of the largest grid a step launch uses (sizes fill / wstat) */
int64_t csgpu_internal_step_waves(const csgpu_model *m);
/* the step kernel over the parents, then cs_collect: survivors appended to the pool behind the parents nobody drew,
 * totals in out[] */
int csgpu_internal_step(const csgpu_model *m, const csgpu_step_launch *L, void *stream);
/* The pool of that path holds ENGINE ROWS (8 bytes per variable like csgpu_val, in the step kernel's terms: bounds
 * relative to the root and the forbidden-set word, cs_step.hip.h).  Conversion in place, rows [first_row, first_row +
 * count) of d_rows: import = interval rows -> engine rows (states put from outside), export = the reverse (states
 * taken away). */
int csgpu_internal_step_import(const csgpu_model *m, csgpu_val *d_rows, int64_t first_row, int64_t count, void *stream);
int csgpu_internal_step_export(const csgpu_model *m, csgpu_val *d_rows, int64_t first_row, int64_t count, void *stream);

#ifdef __cplusplus
}
#endif
#endif
