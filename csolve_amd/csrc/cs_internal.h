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

#ifdef __cplusplus
}
#endif
#endif
