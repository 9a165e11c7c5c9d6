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
/* csgpu_eval_batch over the rows d_list[0 .. *d_count) of d_states, *d_count <= bound */
int csgpu_internal_eval_list(const csgpu_model *m, const csgpu_val *d_states, const int32_t *d_list,
                             const uint64_t *d_count, int64_t bound, int32_t *d_truth, void *stream);

#ifdef __cplusplus
}
#endif
#endif
